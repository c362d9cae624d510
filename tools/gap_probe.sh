#!/bin/bash
# Where do the ~6 us holes around the fused block-tail launch come from?  Kernel traces of back-to-back fused + reduce
# launches (dd_dev_mlp, iters = 6) for several kernel shapes: gaps in front of / behind each launch.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/gap; rm -rf $out; mkdir -p $out
probe() {   # name, args
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/$1 -o kt -- python3 tools/mlp_unit.py $2 --iters 6 > $out/$1.log 2>&1
  python3 - $out/$1 "$1" <<'PY'
import csv,glob,sys,os,re
d=sys.argv[1]
tr=list(csv.DictReader(open(glob.glob(os.path.join(d,"**/*kernel_trace.csv"),recursive=True)[0])))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
tr=[r for r in tr if "mlp_" in r["Kernel_Name"] or "proj_rows" in r["Kernel_Name"]][-10:]
prev=None; out=[]
for r in tr:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    nm=re.sub(r"dd::|\(anonymous namespace\)::|void ","",r["Kernel_Name"])[:28]
    out.append(f"[gap {((s-prev)/1e3 if prev else 0):5.1f}] {nm} {(e-s)/1e3:6.1f}us lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')} acc={r.get('Accum_VGPR_Count','?')}")
    prev=e
print(sys.argv[2]); print("   "+"\n   ".join(out))
PY
}
probe d512_proj_x "--M 4096 --D 512 --proj --extras 1"
probe d512_plain_x "--M 4096 --D 512 --extras 1"
probe d256_ln_x "--M 4096 --D 256 --ln --extras 1"
probe d128_ln_x "--M 4096 --D 128 --ln --extras 1"
