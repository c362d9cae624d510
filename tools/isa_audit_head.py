#!/usr/bin/env python3
"""ISA audit of head_dec_kernel's hand-counted row loads (duodiff_amd/csrc/rowops.hip, embed_dim <= 512).

The row quads and the Wg share of those instantiations are asm loads hipcc does not count: every wait is a hand-written
`s_waitcnt vmcnt(N)`.  That is only correct if the compiler never touches a destination register between its load and the wait
that covers it (a copy for a phi, a spill, an AGPR park would read bytes that have not arrived).  This script checks exactly that
on the compiled ISA (a forward dataflow over the kernel's basic blocks: the set of loads in flight at a join is the union over its predecessors):

  * no scratch_* and no v_accvgpr_* anywhere in the kernel;
  * for every asm `global_load_dwordx4 v[a:b], ...`: no instruction in front of the first hand-written wait that covers the load
    (vector memory returns in order: a wait `vmcnt(N)` covers a load that has at least N asm loads issued behind it) names a
    register of v[a:b] -- other than a later asm load re-using it as its own destination after such a wait;
  * the MFMAs start before the last row quad's wait (the point of the exercise), and no compiler-made `s_waitcnt vmcnt(0)` sits
    between the first row-quad wait and the last one.

    python tools/isa_audit_head.py            (exit code 1 on a finding)
"""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
SRC = REPO / "duodiff_amd" / "csrc" / "rowops.hip"
HAND = [f"ILi{d}ELi{nt}ELi8ELb{pr}ELb{sp}E" for d in (256, 512) for nt in (1, 2, 3, 4) for pr in (0, 1) for sp in (0, 1)]      # <D, NT, 8, PROBE, SPLIT>


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return set()
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", parts[1]):
        out |= regs(tok)
    return out


def audit(name, body):
    findings = []
    if any("scratch_" in l for l in body):
        findings.append("scratch ops")
    if any("v_accvgpr" in l for l in body):
        findings.append("AGPR copies")
    # ---- basic blocks: a label starts one, a branch ends one
    blocks, cur, in_asm = [], {"label": "entry", "ev": [], "succ": None}, False
    for i, l in enumerate(body[1:], 1):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ev": [], "succ": None}
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        if in_asm and t.startswith("global_load_dwordx4"):
            cur["ev"].append(("load", i, frozenset(regs(t.split()[1].rstrip(",")))))
        elif in_asm and t.startswith("s_waitcnt vmcnt("):
            cur["ev"].append(("wait", i, int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
        else:
            cur["ev"].append(("insn", i, t))
            mb = re.match(r"^s_(branch|cbranch_\w+)\s+(\.LBB\d+_\d+)", t)
            if mb or t.startswith("s_endpgm"):
                cur["succ"] = [] if not mb else ([mb.group(2)] if mb.group(1) == "branch" else [mb.group(2), "<next>"])
                blocks.append(cur)
                cur = {"label": None, "ev": [], "succ": None}
    blocks.append(cur)
    index = {b["label"]: n for n, b in enumerate(blocks) if b["label"]}
    succ = []
    for n, b in enumerate(blocks):
        out = b["succ"] if b["succ"] is not None else ["<next>"]
        succ.append([n + 1 if x == "<next>" else index[x] for x in out if x != "<next>" or n + 1 < len(blocks)])

    def run(b, state, report):        # state: frozenset of (dest registers, asm loads issued behind it (capped), line of the load)
        st = set(state)
        for kind, i, p in b["ev"]:
            if kind == "load":
                st = {(d, min(k + 1, 64), at) for d, k, at in st} | {(p, 0, i)}
            elif kind == "wait":
                st = {(d, k, at) for d, k, at in st if k < p}          # still in flight: among the youngest p loads
            else:
                if report is not None and not p.startswith("s_"):
                    ops = operands(p)
                    for d, _, at in st:
                        if ops & d:
                            report.add(f"line {i}: `{p[:70]}` touches v{min(d)}..v{max(d)} whose load (line {at}) no wait has covered yet")
                if report is not None and st and re.match(r"s_waitcnt vmcnt\(0\)", p):
                    report.add(f"line {i}: compiler-made s_waitcnt vmcnt(0) with {len(st)} hand-counted loads in flight")
                if re.match(r"s_waitcnt vmcnt\(0\)", p):
                    st = set()
        return frozenset(st)

    state_in = [frozenset() for _ in blocks]
    work = [0]
    while work:                        # forward dataflow to a fixpoint: state at a join = union over the predecessors
        n = work.pop()
        out = run(blocks[n], state_in[n], None)
        for m in succ[n]:
            if not out <= state_in[m]:
                state_in[m] = state_in[m] | out
                work.append(m)
    report = set()
    for n, b in enumerate(blocks):
        run(b, state_in[n], report)
    findings += sorted(report, key=lambda x: int(x.split()[1].rstrip(":")))
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    waits = [i for b in blocks for k, i, p in b["ev"] if k == "wait"]
    if not mf or not waits or mf[0] > waits[-1]:
        findings.append("the first MFMA sits behind the last hand-written wait: nothing streams")
    return findings


def main():
    out = "/tmp/rowops_audit.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{REPO / 'include'}", "-S", "--cuda-device-only",
                    str(SRC), "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    rc = 0
    seen = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN2dd15head_dec_kernel(\w+?)EEvNS_11HeadDecArgsE):", l)
        if not m or m.group(2) not in HAND:
            continue
        seen += 1
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i:end]
        f = audit(m.group(1), body)
        n_load = sum(1 for j, b in enumerate(body) if b.strip().startswith("global_load_dwordx4") and j and "ASMSTART" in body[j - 1])
        print(f"head_dec_kernel<{m.group(2)}>: {len(body)} lines, {n_load} asm loads, {sum('v_mfma' in b for b in body)} MFMAs:", "OK" if not f else "FAILED")
        for x in f[:8]:
            print("   ", x)
        rc |= 1 if f else 0
    if seen < 32:
        print(f"only {seen} hand-counted instantiations found (expected 32: D = 256 / 512 x NT 1..4 x PROBE x SPLIT)")
        rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
