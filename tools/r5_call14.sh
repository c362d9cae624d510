#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5p; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_qkv_attention.py tests/test_gpu_parity.py -x -q -m gpu -k "qkv or full_size or chains or gemm or width or rowlin or splitk or 768 or 1024" 2>&1 | tail -3
for W in celeba imagenet64 imagenet256; do bash tools/ab_lib.sh prev $W 2 2>&1 | tee -a $out/ab_saddr.txt; done
