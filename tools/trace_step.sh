#!/bin/bash
# Kernel-trace timeline of one full-model step of the headline bench (gaps between launches included).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/trace; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline > $out/bench.json 2> $out/bench.err
python3 tools/prof_summary.py $out/kt 90 > $out/kernel_trace_summary.txt
python3 - $out/kt <<'PY'
import csv,glob,sys,os,re
d=sys.argv[1]
tr=list(csv.DictReader(open(glob.glob(os.path.join(d,"*kernel_trace.csv"))[0])))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(tr) if "embed_" in r["Kernel_Name"]]
seq=tr[idx[-2]:idx[-1]]
clean=lambda n: re.sub(r"dd::|\(anonymous namespace\)::|unsigned short|void ","",n)[:34]
prev=None
for r in seq[:40]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    print(f"gap {gap:6.1f}  dur {(e-s)/1e3:7.1f}  {clean(r['Kernel_Name'])}  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')} lds {r.get('LDS_Block_Size','?')} scr {r.get('Scratch_Size', r.get('Private_Segment_Size','?'))} vgpr {r.get('VGPR_Count','?')}")
    prev=e
PY
rm -rf $out/kt
