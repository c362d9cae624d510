#!/bin/bash
# round-5 GPU call 1: power / clock probe of the fused block tail (stamped variant) + LDS conflict counters per kernel (baseline)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5a; rm -rf $out; mkdir -p $out
DUODIFF_LIB=$PWD/duodiff_amd/libduodiff_stamps.so timeout -k 10 300 python3 tools/power_probe.py --iters 3000 > $out/power_probe.txt 2> $out/power_probe.err || { echo "power probe failed"; tail -5 $out/power_probe.err; exit 1; }
cat $out/power_probe.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u > $out/lds_counters.txt; cat $out/lds_counters.txt | tr '\n' ' '; echo
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/lds -o p -- python3 tools/fwd_few.py celeba > $out/lds.log 2>&1 || { echo "pmc lds failed"; tail -5 $out/lds.log; }
python3 - $out/lds <<'PY'
import csv,glob,sys,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=re.sub(r"dd::|\(anonymous namespace\)::|unsigned short|void ","",r["Kernel_Name"])[:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in sorted(acc.items()):
    up=lambda v: (lambda s: sum(s[len(s)//2:])/max(1,len(s[len(s)//2:])))(sorted(v))
    print(f"{k:44s}", {n: round(up(v)) for n,v in sorted(c.items())})
PY
find $out -name "*.csv" -size +4M -delete
