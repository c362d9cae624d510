#!/bin/bash
# Collect everything profiles/rNN holds for the headline workload on the GPU box (outputs under gpurun_out/profiles/):
#   bench lines (driver command and the full 1000-step run), rocprofv3 --kernel-trace --stats of the driver command,
#   PMC traffic and SQ utilisation of the full-model forward (tools/collect_pmc.sh).
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles
rm -rf $out && mkdir -p $out/kt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd_steps20.json 2> $out/bench_steps20.err
timeout -k 10 600 python3 bench.py --steps 1000 --warmup 5 --no_cpu_baseline > $out/bench_steps1000.json 2> $out/bench_steps1000.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 20 --warmup 5 > $out/kt/bench.json 2> $out/kt/bench.err
python3 tools/prof_summary.py $out/kt 90 > $out/kernel_trace_summary.txt
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/kt
bash tools/collect_pmc.sh > $out/pmc.log 2>&1
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc/pmc_sq.json gpurun_out/pmc/pmc_traffic_summary.txt gpurun_out/pmc/pmc_sq_utilisation.txt $out/
head -c 600 $out/bench_driver_cmd_steps20.json; echo; head -c 400 $out/bench_steps1000.json; echo; head -20 $out/kernel_trace_summary.txt
