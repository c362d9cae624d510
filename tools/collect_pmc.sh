#!/bin/bash
# Collect the per-kernel PMC profiles of one workload's full-backbone forward (tools/fwd_few.py, bf16, BASELINE batch) on the GPU box:
# HBM traffic in two passes (FETCH_SIZE, WRITE_SIZE) and SQ utilisation + the clock in a third (SQ counters that fit one pass,
# plus GRBM_GUI_ACTIVE: the GRBM block has its own slots).  Counters only (--kernel-trace for the timestamps), no sys/hip traces.
#   bash tools/collect_pmc.sh [celeba|imagenet64|imagenet256] [outdir] [dev_flags]
set -e
W=${1:-celeba}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${2:-gpurun_out/pmc}
F=${3:-0}
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- python3 tools/fwd_few.py $W $F > $out/$c.log 2>&1
done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/SQ -o p -- python3 tools/fwd_few.py $W $F > $out/SQ.log 2>&1
python3 tools/pmc_summary.py $out/FETCH_SIZE $out/WRITE_SIZE $out/pmc_traffic.json > $out/pmc_traffic_summary.txt
python3 tools/pmc_sq_summary.py $out/SQ $out/pmc_sq.json > $out/pmc_sq_utilisation.txt
cat $out/pmc_traffic_summary.txt $out/pmc_sq_utilisation.txt
# the raw counter csv files are large; keep the summaries
find $out -name "*.csv" -size +8M -delete
