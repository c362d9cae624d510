#!/bin/bash
# Quick per-kernel table of the headline bench on the GPU box: bash tools/kt_quick.sh NAME  -> gpurun_out/kt_NAME.txt (+ the bench line)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; N=${1:-x}; shift
rm -rf $R/gpurun_out/kt_$N && mkdir -p $R/gpurun_out/kt_$N
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$N -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline "$@" > $R/gpurun_out/kt_$N/bench.json 2> $R/gpurun_out/kt_$N/bench.err
cd $R && python3 tools/prof_summary.py gpurun_out/kt_$N 100 > gpurun_out/kt_$N.txt; cp gpurun_out/kt_$N/kt_kernel_stats.csv gpurun_out/kt_$N.csv; rm -rf gpurun_out/kt_$N/*.csv
head -30 gpurun_out/kt_$N.txt
