#!/bin/bash
# Same-box A/B of two dd_dev_set_flags settings on bench.py: bash tools/ab_flags.sh [workload] [flagsA] [flagsB] [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-imagenet64}; A=${2:-0}; B=${3:-1024}; R=${4:-1}
out=gpurun_out/abf_$W; rm -rf $out; mkdir -p $out
for r in $(seq 1 $R); do
  for F in $A $B; do
    timeout -k 10 400 python3 bench.py --workload $W --steps 100 --warmup 5 --no_cpu_baseline --dev_flags $F > $out/bench_${F}_$r.json 2> $out/bench_${F}_$r.err || { echo "bench flags $F failed"; tail -5 $out/bench_${F}_$r.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/bench_${F}_$r.json')); print('$W flags $F round $r: %.2f img/s  chains %d  full step %.3f ms  shallow %.3f ms' % (d['value'], d['config']['chains_in_timed_region'], d['config']['gpu_ms_late_backbone']/70, d['config']['gpu_ms_first_backbone']/30))"
  done
done
