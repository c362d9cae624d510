#!/bin/bash
# Kernel trace + bench line of one bench.py workload on the GPU box: bash tools/collect_workload_profile.sh WORKLOAD [outdir]
set -e
W=$1
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${2:-gpurun_out/profiles_$W}
rm -rf $out && mkdir -p $out/kt
timeout -k 10 500 python3 bench.py --workload $W --steps 20 --warmup 5 --no_cpu_baseline > $out/bench_$W.json 2> $out/bench_$W.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --workload $W --steps 20 --warmup 5 --no_cpu_baseline > $out/kt/bench.json 2> $out/kt/bench.err
python3 tools/prof_summary.py $out/kt 120 > $out/kernel_trace_summary_$W.txt
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats_$W.csv
rm -rf $out/kt
head -c 700 $out/bench_$W.json; echo; head -18 $out/kernel_trace_summary_$W.txt
