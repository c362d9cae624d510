#!/usr/bin/env python3
"""ISA audit of mlp_fused_kernel<512>: the hot loop must contain only what was hand-placed.

hipcc neither counts the asm ds_reads nor pads hazards around the asm MFMAs (mlp_fused.hip header), so a
compiler-generated copy, AGPR access or spill inside the loop would read registers whose producer is still in flight.
Checks, inside the innermost loop: no scratch_*, no v_accvgpr_*, no v_mov*, MFMA / ds_read / LDS-DMA counts as designed.
"""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
SRC = Path(sys.argv[1]).resolve() if len(sys.argv) > 1 else REPO / "duodiff_amd" / "csrc" / "mlp_fused.hip"   # (a variant copy: tools/build_variant.py --csrc)


def main():
    out = "/tmp/mlp_fused_audit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{REPO / 'include'}",
                    "-S", "--cuda-device-only", str(SRC), "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    rc = 0
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2dd\S*mlp_fused_kernelILi512E\S*:", l)]
    assert len(starts) == 7, "expected the plain, LayerNorm-in, + proj, + skip, + skip + y-tap (early exit), + qkv and + skip + qkv instantiations"
    for start in starts:
        end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))   # (a kernel may have several s_endpgm)
        body = lines[start:end]
        loops = []
        for head in (i for i, l in enumerate(body) if "Loop Header" in l):
            label = body[head].split(":")[0]
            back = next((i for i in range(head, len(body)) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", body[i])), None)
            if back is not None and sum("v_mfma" in l for l in body[head:back + 1]) >= 64:
                loops.append(body[head:back + 1])          # the chunk loops of the main and of the hidden-split body
        ok = len(loops) == 2
        for loop in loops:
            cnt = lambda pat: sum(bool(re.search(pat, l)) for l in loop)
            rep = {"lines": len(loop), "mfma": cnt(r"v_mfma"), "ds_read_b128": cnt(r"ds_read_b128"), "lds_dma": cnt(r"global_load_lds"),
                   "barrier": cnt(r"s_barrier"), "scratch": cnt(r"scratch_"), "accvgpr": cnt(r"v_accvgpr"), "v_mov": cnt(r"\bv_mov"),
                   "ds_write": cnt(r"ds_write"), "waitcnt_vm0": cnt(r"s_waitcnt vmcnt\(0\)"), "s_nop": cnt(r"s_nop")}
            good = (rep["scratch"] == 0 and rep["accvgpr"] == 0 and rep["v_mov"] == 0 and rep["ds_write"] == 0 and rep["waitcnt_vm0"] == 0
                    and rep["mfma"] == 128 and rep["ds_read_b128"] == 128 + 8 and rep["lds_dma"] == 32 and rep["barrier"] == 4)
            print("  loop:", rep, "OK" if good else "FAILED")
            ok = ok and good
        if "ELb1ELb1ELb1E" in lines[start] or "ELb1ELb1ELb0ELb1E" in lines[start]:
            # SKIP / QKV instantiations: the MFMAs of the skip / qkv phases sit behind the chunk loop -- straight-line
            # code; no compiler-generated register traffic (copies, AGPR moves, spills) may sit between its MFMAs
            mf = [i for i, l in enumerate(body) if "v_mfma" in l]
            last_loop_end = max(i for i, l in enumerate(body) if "s_cbranch" in l and i < mf[-1] - 1000) if any("s_cbranch" in l for l in body) else 0
            region = [i for i in mf if i > last_loop_end]
            tail = body[region[0]:region[-1] + 1]
            cnt = lambda pat: sum(bool(re.search(pat, l)) for l in tail)
            rep = {"mfma": cnt(r"v_mfma"), "ds_read_b128": cnt(r"ds_read_b128"), "lds_dma": cnt(r"global_load_lds"), "barrier": cnt(r"s_barrier"),
                   "scratch": cnt(r"scratch_"), "accvgpr": cnt(r"v_accvgpr"), "waitcnt_vm0": cnt(r"s_waitcnt vmcnt\(0\)")}
            print("  straight-line MFMA region after the last loop:", rep)
        total_scratch = sum("scratch_" in l for l in body)
        print(lines[start].split(":")[0][-40:], "| instructions:", sum(l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;") for l in body),
              "| scratch ops in whole kernel:", total_scratch, "| AUDIT", "OK" if ok else "FAILED")
        rc |= 0 if ok else 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
