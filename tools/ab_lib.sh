#!/bin/bash
# Same-box A/B of the product library against a variant build (tools/build_variant.py NAME): bash tools/ab_lib.sh NAME [workload] [rounds] [dev_flags]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$1; W=${2:-celeba}; R=${3:-2}; F=${4:-0}
out=gpurun_out/abl_${V}_$W; rm -rf $out; mkdir -p $out
for r in $(seq 1 $R); do
  for L in product $V; do
    if [ $L = product ]; then unset DUODIFF_LIB; else export DUODIFF_LIB=$PWD/duodiff_amd/libduodiff_$V.so; fi
    timeout -k 10 400 python3 bench.py --workload $W --steps 100 --warmup 5 --no_cpu_baseline --dev_flags $F > $out/bench_${L}_$r.json 2> $out/bench_${L}_$r.err || { echo "bench $L failed"; tail -5 $out/bench_${L}_$r.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/bench_${L}_$r.json')); print('$W $L round $r: %.2f img/s  chains %d  full step %.3f ms  shallow %.3f ms' % (d['value'], d['config']['chains_in_timed_region'], d['config']['gpu_ms_late_backbone']/70, d['config']['gpu_ms_first_backbone']/30))"
  done
done
