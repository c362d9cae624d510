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
SRC = REPO / "duodiff_amd" / "csrc" / "mlp_fused.hip"


def main():
    out = "/tmp/mlp_fused_audit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{REPO / 'include'}",
                    "-S", "--cuda-device-only", str(SRC), "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2dd\S*mlp_fused_kernelILi512E\S*:", l))
    end = next(j for j in range(start, len(lines)) if "s_endpgm" in lines[j])
    body = lines[start:end]
    loop = []
    for head in (i for i, l in enumerate(body) if "Loop Header" in l):       # the hot loop = the longest loop body
        label = body[head].split(":")[0]
        back = next(i for i in range(head, len(body)) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", body[i]))
        if back - head + 1 > len(loop):
            loop = body[head:back + 1]
    cnt = lambda pat: sum(bool(re.search(pat, l)) for l in loop)
    rep = {"lines": len(loop), "mfma": cnt(r"v_mfma"), "ds_read_b128": cnt(r"ds_read_b128"), "lds_dma": cnt(r"global_load_lds"),
           "barrier": cnt(r"s_barrier"), "scratch": cnt(r"scratch_"), "accvgpr": cnt(r"v_accvgpr"), "v_mov": cnt(r"\bv_mov"),
           "ds_write": cnt(r"ds_write"), "waitcnt_vm0": cnt(r"s_waitcnt vmcnt\(0\)")}
    total_scratch = sum("scratch_" in l for l in body)
    print(rep, "| scratch ops in whole kernel:", total_scratch)
    ok = (rep["scratch"] == 0 and rep["accvgpr"] == 0 and rep["v_mov"] == 0 and rep["ds_write"] == 0 and rep["waitcnt_vm0"] == 0
          and rep["mfma"] == 128 and rep["ds_read_b128"] == 128 + 8 and rep["lds_dma"] == 32 and rep["barrier"] == 4)
    print("AUDIT", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
