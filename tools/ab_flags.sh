#!/bin/bash
# Same-box A/B of two dd_dev_set_flags settings of ONE build on the headline bench (interleaved rounds) + a kernel trace of each.
#   bash tools/ab_flags.sh FLAGS_A FLAGS_B [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/abf; rm -rf $out; mkdir -p $out
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for n in A B; do
    F=$A; [ $n = B ] && F=$B
    timeout -k 10 300 python3 bench.py --steps 100 --warmup 5 --no_cpu_baseline --dev_flags $F > $out/bench_${n}_$r.json 2> $out/bench_${n}_$r.err || { echo "bench $n failed"; tail -5 $out/bench_${n}_$r.err; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$out/bench_${n}_$r.json')); print('$n (flags $F) round $r: %.2f img/s  full step %.3f ms  shallow %.3f ms  fused tail %.1f us' % (d['value'], d['config']['gpu_ms_late_backbone']/70, d['config']['gpu_ms_first_backbone']/30, d['roofline']['ms_per_launch']*1e3))"
  done
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for n in A B; do
  F=$A; [ $n = B ] && F=$B
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$n -o kt -- python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline --dev_flags $F > $out/kt_$n.json 2> $out/kt_$n.err
  python3 tools/prof_summary.py $out/kt_$n 0 | head -17 > $out/kernel_stats_$n.txt
  echo "== $n (flags $F)"; cat $out/kernel_stats_$n.txt
  rm -rf $out/kt_$n
done
