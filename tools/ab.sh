#!/bin/bash
# Same-box A/B of two builds of the library on the headline bench (interleaved rounds) + a kernel trace of each.
#   bash tools/ab.sh NAME_A NAME_B [rounds]     (NAME = "" for the product build, else duodiff_amd/libduodiff_NAME.so)
# Outputs under gpurun_out/ab/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab; rm -rf $out; mkdir -p $out
lib() { if [ -z "$1" ] || [ "$1" = "product" ]; then echo "$PWD/duodiff_amd/libduodiff.so"; else echo "$PWD/duodiff_amd/libduodiff_$1.so"; fi; }
A=$(lib "$1"); B=$(lib "$2"); R=${3:-3}
for r in $(seq 1 $R); do
  for n in A B; do
    L=$A; [ $n = B ] && L=$B
    DUODIFF_LIB=$L timeout -k 10 300 python3 bench.py --steps 100 --warmup 5 --no_cpu_baseline > $out/bench_${n}_$r.json 2> $out/bench_${n}_$r.err || { echo "bench $n failed"; tail -5 $out/bench_${n}_$r.err; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$out/bench_${n}_$r.json')); print('$n round $r: %.2f img/s  full step %.3f ms  shallow %.3f ms  fused tail %.1f us' % (d['value'], d['config']['gpu_ms_late_backbone']/70, d['config']['gpu_ms_first_backbone']/30, d['roofline']['ms_per_launch']*1e3))"
  done
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for n in A B; do
  L=$A; [ $n = B ] && L=$B
  export DUODIFF_LIB=$L
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$n -o kt -- python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/kt_$n.json 2> $out/kt_$n.err
  python3 tools/prof_summary.py $out/kt_$n 0 | head -16 > $out/kernel_stats_$n.txt
  echo "== $n"; cat $out/kernel_stats_$n.txt
  rm -rf $out/kt_$n
done
