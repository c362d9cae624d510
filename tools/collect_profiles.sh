#!/bin/bash
# Collect everything profiles/rNN holds, on the GPU box (outputs under gpurun_out/profiles/):
#   headline workload: bench lines (the driver's command and the full 1000-step run), rocprofv3 --kernel-trace --stats of the driver's
#   command (per-kernel and per (kernel, grid) tables), PMC traffic + SQ utilisation + clock of the full-model forward (tools/collect_pmc.sh);
#   ImageNet-64 / ImageNet-256 workloads: bench line, kernel trace of the bench, the same PMC summaries.
#   bash tools/collect_profiles.sh [celeba] [imagenet64] [imagenet256]     (default: all three)
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
WL="${@:-celeba imagenet64 imagenet256}"
for W in $WL; do
  sfx=""; [ $W != celeba ] && sfx="_$W"
  # PMC first: bench.py quotes HBM traffic / MFMA-busy / clock of the dominant kernel from profiles/rNN/pmc_*.json when they are of THIS build
  bash tools/collect_pmc.sh $W $R/gpurun_out/pmc$sfx > $out/pmc$sfx.log 2>&1
  for f in pmc_traffic.json pmc_sq.json; do cp gpurun_out/pmc$sfx/$f $out/${f%.json}$sfx.json; cp gpurun_out/pmc$sfx/$f $R/profiles/${ROUND:-r05}/${f%.json}$sfx.json; done
  cp gpurun_out/pmc$sfx/pmc_traffic_summary.txt $out/pmc_traffic_summary$sfx.txt
  cp gpurun_out/pmc$sfx/pmc_sq_utilisation.txt $out/pmc_sq_utilisation$sfx.txt
  rm -rf $out/kt$sfx && mkdir -p $out/kt$sfx
  if [ $W = celeba ]; then
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd_steps20.json 2> $out/bench_steps20.err
    timeout -k 10 600 python3 bench.py --steps 1000 --warmup 5 --no_cpu_baseline > $out/bench_steps1000.json 2> $out/bench_steps1000.err
  else
    timeout -k 10 500 python3 bench.py --workload $W --steps 20 --warmup 5 --no_cpu_baseline > $out/bench$sfx.json 2> $out/bench$sfx.err
  fi
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$sfx -o kt -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no_cpu_baseline > $out/kt$sfx/bench.json 2> $out/kt$sfx/bench.err)
  python3 tools/prof_summary.py $out/kt$sfx 100 > $out/kernel_trace_summary$sfx.txt
  cp $out/kt$sfx/kt_kernel_stats.csv $out/kernel_stats$sfx.csv
  rm -rf $out/kt$sfx
  echo "== $W"; head -c 400 $out/bench$( [ $W = celeba ] && echo _driver_cmd_steps20 || echo $sfx ).json; echo; head -12 $out/kernel_trace_summary$sfx.txt
done
