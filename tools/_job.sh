cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
rm -rf gpurun_out/at; mkdir -p gpurun_out/at
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/at -o kt -- python3 tools/fwd_few.py > gpurun_out/at/log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/at/kt_kernel_stats.csv")):
    if 'attention' in r['Name'] or 'gemm256_kernel<0>' in r['Name']:
        print(r['Name'][27:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
rm -f gpurun_out/at/kt_kernel_trace.csv
