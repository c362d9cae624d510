#!/bin/bash
# Collect the per-kernel PMC profiles of the headline forward (tools/fwd_few.py: CelebA U-ViT, B=128, bf16) on the GPU box:
# HBM traffic in two passes (FETCH_SIZE, WRITE_SIZE) and SQ utilisation in two more (counter groups that fit one pass each).
# Counters only (--kernel-trace for the timestamps), no sys/hip traces.  Outputs under gpurun_out/pmc/.
set -e
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- python3 tools/fwd_few.py > $out/$c.log 2>&1
done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/SQ -o p -- python3 tools/fwd_few.py > $out/SQ.log 2>&1
python3 tools/pmc_summary.py $out/FETCH_SIZE $out/WRITE_SIZE $out/pmc_traffic.json > $out/pmc_traffic_summary.txt
python3 tools/pmc_sq_summary.py $out/SQ $out/pmc_sq.json > $out/pmc_sq_utilisation.txt
cat $out/pmc_traffic_summary.txt $out/pmc_sq_utilisation.txt
# the raw counter csv files are large; keep the summaries
find $out -name "*.csv" -size +8M -delete
