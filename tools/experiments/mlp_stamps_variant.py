#!/usr/bin/env python3
"""Stamped variant of the fused block tail for tools/power_probe.py (a diagnostic build, never the product):

    python tools/experiments/mlp_stamps_variant.py && python tools/build_variant.py stamps --csrc build/var_stamps/pkg/csrc

copies duodiff_amd/csrc to build/var_stamps/pkg/csrc and inserts s_memtime / s_memrealtime stamps (each followed by its own
s_waitcnt lgkmcnt(0): the loop's counted waits never see them) at the phase boundaries of mlp_body: entry, in front of the first
chunk, behind the chunk loop, at the epilogue's start, at the end (after s_waitcnt vmcnt(0)).  Wave 0 of every workgroup writes them
to a __device__ array of its own (read back through dd_dev_read_stamps); no output of the kernel depends on them.
"""
import shutil
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
dst = REPO / "build" / "var_stamps" / "pkg" / "csrc"
if dst.exists():
    shutil.rmtree(dst)
dst.parent.mkdir(parents=True, exist_ok=True)
shutil.copytree(REPO / "duodiff_amd" / "csrc", dst)
inc = REPO / "build" / "var_stamps" / "include"          # capi.hip includes ../../include/duodiff.h
if not inc.exists():
    inc.symlink_to(REPO / "include")
p = dst / "mlp_fused.hip"
s = p.read_text()


def rep(old, new):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, 1)


rep("namespace dd {\nnamespace {\n\nconstexpr int kMaxHidden",
    "namespace dd {\nnamespace {\n__device__ unsigned long long g_stamps[2048 * 8];\n"
    "#define DD_ST(v) unsigned long long v; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(v) :: \"memory\")\n"
    "#define DD_RT(v) unsigned long long v; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(v) :: \"memory\")\n\nconstexpr int kMaxHidden")
rep("    using C = MlpCfg<D>;\n    static_assert(!PROJ || (LNIN", "    DD_ST(st0); DD_RT(rt0);\n    using C = MlpCfg<D>;\n    static_assert(!PROJ || (LNIN")
rep("    bias_init(c0, sA);\n    // S of the first chunk", "    DD_ST(st1);\n    bias_init(c0, sA);\n    // S of the first chunk")
rep("    // SKIP: the first half of the long-skip operand's rows (k-steps 0 .. F/2-1, natural k order) is requested here",
    "    DD_ST(st2);\n    // SKIP: the first half of the long-skip operand's rows (k-steps 0 .. F/2-1, natural k order) is requested here")
rep("    const int he = half_of();\n    if constexpr (PARTIAL) {", "    DD_ST(st3);\n    const int he = half_of();\n    if constexpr (PARTIAL) {")
rep("        if constexpr (QKV) {\n            // ---- the qkv phases.",
    "        if constexpr (!QKV && !PARTIAL) {\n            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            DD_ST(st4); DD_RT(rt4);\n"
    "            if (threadIdx.x == 0 && blockIdx.x < 2048) {\n                unsigned long long* o = g_stamps + blockIdx.x * 8;\n"
    "                o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = rt0; o[6] = rt4;\n            }\n        }\n"
    "        if constexpr (QKV) {\n            // ---- the qkv phases.")
s = s.rstrip() + ("\n\nextern \"C\" int dd_dev_read_stamps(unsigned long long* out, int n) {\n"
                  "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dd::g_stamps), (size_t)n * 8);\n}\n")
p.write_text(s)
print("wrote", p)
