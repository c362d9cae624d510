#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5o; rm -rf $out; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_mlp_fused.py -x -q -m gpu 2>&1 | tail -2
for v in stampsold stamps stampsold stamps; do
  echo "== $v" | tee -a $out/saddr.txt
  DUODIFF_LIB=$R/duodiff_amd/libduodiff_$v.so timeout -k 10 200 python3 tools/power_probe.py --iters 2000 --tiles 256 128 2>> $out/err.txt | grep "random" | head -2 | tee -a $out/saddr.txt
done
