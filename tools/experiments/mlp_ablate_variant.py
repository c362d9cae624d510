#!/usr/bin/env python3
"""Energy ablations of the fused block tail's chunk loop (diagnostic builds on top of the stamped variant; results are WRONG by
construction -- only time per launch, cycles and the clock the chip holds are read, tools/power_probe.py):

    python tools/experiments/mlp_ablate_variant.py MODE      MODE in nogelu | noread | nodma | nomfma
    python tools/build_variant.py abl_MODE --csrc build/var_abl_MODE/pkg/csrc

  nogelu  the chunk loop's gaps carry no GELU piece (the hidden activations stay zero)
  noread  ... no LDS fragment read (the MFMAs re-use whatever the fragment registers hold)
  nodma   ... no LDS-DMA request (the ring keeps its first blocks)
  nomfma  ... no MFMA (fragment reads, GELU pieces, DMA requests and waits remain)
"""
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
mode = sys.argv[1]
subprocess.run([sys.executable, str(REPO / "tools" / "experiments" / "mlp_stamps_variant.py")], check=True)
src = REPO / "build" / "var_stamps" / "pkg"
dst = REPO / "build" / f"var_abl_{mode}" / "pkg"
if dst.parent.exists():
    shutil.rmtree(dst.parent)
dst.parent.mkdir(parents=True)
shutil.copytree(src, dst)
(dst.parent / "include").symlink_to(REPO / "include")
p = dst / "csrc" / "mlp_fused.hip"
s = p.read_text()


def rep(old, new, count=1):
    global s
    assert s.count(old) >= count, (old, s.count(old))
    s = s.replace(old, new)


loop_a = "gap_stmt<1, LG, true, LOA, K>(Y[g >> 1], wq[g % PD], p_prev[g & 1], la, s_cur[2 * pair], s_cur[2 * pair + 1], gk, gr, pw[pair]);"
loop_b = "gap_stmt<0, LG, true, LOA, K>(s_next, wq[g % PD], xf[g - C::F], la, s_cur[2 * pair], s_cur[2 * pair + 1], gk, gr, pw[pair]);"
if mode == "nogelu":
    rep(loop_a, loop_a.replace("LOA, K>", "LOA, -1>"))
    rep(loop_b, loop_b.replace("LOA, K>", "LOA, -1>"))
    rep("        unsigned pw[8];\n", "        unsigned pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n")
elif mode == "noread":
    rep(loop_a, loop_a.replace("LG, true, LOA", "-1, false, LOA"))
    rep(loop_b, loop_b.replace("LG, true, LOA", "-1, false, LOA"))
elif mode == "nodma":
    rep("                    if constexpr (g < C::F) glds16u_j<j>(src_e, dma_voff, dst_e);\n                    else glds16u_j<j>(src_m, dma_voff, dst_m);",
        "                    (void)src_e; (void)src_m; (void)dst_e; (void)dst_m;")
elif mode == "nomfma":
    # the loop's two statements get their own MFMA-less strings: a flag template parameter would touch every instantiation
    rep('#define DD_S_MFMA_W "s_waitcnt lgkmcnt(%[lg])\\n\\tv_mfma_f32_32x32x16_bf16 %[acc], %[wa], %[xb], %[acc]"',
        '#define DD_S_MFMA_W "s_waitcnt lgkmcnt(%[lg])\\n\\ts_nop 0"')
    rep('#define DD_S_MFMA_N "v_mfma_f32_32x32x16_bf16 %[acc], %[wa], %[xb], %[acc]"     // LG < 0: the gap in front already waited for this fragment',
        '#define DD_S_MFMA_N "s_nop 0"')
else:
    raise SystemExit(__doc__)
p.write_text(s)
print("wrote", p)
