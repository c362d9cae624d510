cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc2; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $out/sq_counters.txt
wc -l $out/sq_counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -o p -- python3 tools/fwd_few.py > $out/$n.log 2>&1
  python3 - "$out/$n" <<'PY'
import csv,glob,sys,collections,re
d=sys.argv[1]
f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=re.sub(r"dd::|\(anonymous namespace\)::|unsigned short","",r["Kernel_Name"])[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    if not any(s in k for s in ("gemm256","attention","mlp_fused")): continue
    wc=sorted(c.get("SQ_WAVE_CYCLES",[1]))
    up=lambda v: (lambda s: sum(s[len(s)//2:])/max(1,len(s[len(s)//2:])))(sorted(v))
    w=up(c["SQ_WAVE_CYCLES"])
    print(k, {n: round(up(v)/w,3) if n not in ("SQ_WAVES","SQ_WAVE_CYCLES") else round(up(v)) for n,v in c.items()})
PY
done
