#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5c; rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 tools/poison_probe.py > $out/poison_probe.txt 2> $out/poison_probe.err || { echo "poison probe failed"; tail -5 $out/poison_probe.err; }
cat $out/poison_probe.txt
timeout -k 10 600 python3 -m pytest tests/test_qkv_attention.py tests/test_mlp_fused.py -x -q -m gpu 2>&1 | tail -5 | tee $out/pytest_units.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/lds -o p -- python3 tools/fwd_few.py celeba > $out/lds.log 2>&1 || { echo "pmc lds failed"; tail -5 $out/lds.log; }
python3 - $out/lds <<'PY' | tee $out/lds_conflicts.txt
import csv,glob,sys,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=re.sub(r"dd::|\(anonymous namespace\)::|unsigned short|void ","",r["Kernel_Name"])[:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in sorted(acc.items()):
    up=lambda v: (lambda s: sum(s[len(s)//2:])/max(1,len(s[len(s)//2:])))(sorted(v))
    if up(c.get("SQ_INSTS_LDS",[0]))>0: print(f"{k:44s}", {n: round(up(v)) for n,v in sorted(c.items())})
PY
find $out -name "*.csv" -size +4M -delete
