#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5i; rm -rf $out; mkdir -p $out
# early-exit loop: which kernels run
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt_ee -o kt -- python3 $R/tools/ee_trace.py 6 > $R/$out/kt_ee.log 2>&1)
python3 tools/prof_summary.py $out/kt_ee 70 > $out/ee_trace_summary.txt; head -24 $out/ee_trace_summary.txt
rm -rf $out/kt_ee
# attention kernel: extras' operands read at the block's start (product) vs behind the mid-block barrier (prevatt), interleaved
timeout -k 10 300 python3 -m pytest tests/test_qkv_attention.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab.sh product prevatt 2 2>&1 | tee $out/ab_attention.txt | grep -v "^$" | head -40
