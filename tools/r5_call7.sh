#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5g; rm -rf $out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_early_exit.py tests/test_mlp_fused.py -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_ee.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_parity.txt
timeout -k 10 300 python3 tools/ddim_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/loops_probe.txt
for i in 1 2; do timeout -k 10 300 python3 bench.py --steps 100 --warmup 5 --no_cpu_baseline > $out/bench_$i.json 2> $out/bench_$i.err; python3 -c "import json; d=json.load(open('$out/bench_$i.json')); print('bench %d: %.2f img/s full step %.3f ms shallow %.3f ms tail %.1f us' % ($i, d['value'], d['config']['gpu_ms_late_backbone']/70, d['config']['gpu_ms_first_backbone']/30, d['roofline']['ms_per_launch']*1e3))"; done
