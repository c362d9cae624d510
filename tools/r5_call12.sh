#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5m; rm -rf $out; mkdir -p $out
timeout -k 10 300 python3 tools/widened_bench.py --out $out/widened_bench.json > $out/widened.log 2>&1; tail -1 $out/widened.log | cut -c1-900
timeout -k 10 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/parity_numbers.txt 2>&1; tail -4 $out/parity_numbers.txt
