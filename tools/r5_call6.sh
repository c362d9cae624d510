#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5f; rm -rf $out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_early_exit.py -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_ee.txt
timeout -k 10 300 python3 tools/ddim_probe.py 2>&1 | tee $out/ddim_probe.txt
timeout -k 10 400 python3 tools/widened_bench.py --out $out/widened_bench.json 2>&1 | tail -3 | cut -c1-600
