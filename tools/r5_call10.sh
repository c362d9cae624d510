#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
out=gpurun_out/r5j; rm -rf $out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_early_exit.py -x -q -m gpu 2>&1 | tail -6 | tee $out/pytest_ee.txt
timeout -k 10 300 python3 tools/ddim_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/loops_probe.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt_ee -o kt -- python3 $R/tools/ee_trace.py 6 > $R/$out/kt_ee.log 2>&1)
python3 tools/prof_summary.py $out/kt_ee 70 > $out/ee_trace_summary.txt; head -16 $out/ee_trace_summary.txt
rm -rf $out/kt_ee
