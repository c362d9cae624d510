#!/usr/bin/env python3
"""Stamped variant of qkv_attention_kernel for tools/attn_probe.py (a diagnostic build, never the product):

    python tools/experiments/attn_stamps_variant.py && bash tools/build_one_variant.sh attention astamps build/var_astamps/pkg/csrc

s_memtime / s_memrealtime (each with its own s_waitcnt lgkmcnt(0)) at the phase boundaries of every wave: entry, first MFMA of phase A, end of phase A, K / V images complete,
end of phase B, end of the extras' chunk.  Written to a __device__ array of their own (dd_dev_read_attn_stamps); no output depends on them.
"""
import shutil
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
dst = REPO / "build" / "var_astamps" / "pkg" / "csrc"
if dst.parent.parent.exists():
    shutil.rmtree(dst.parent.parent)
dst.parent.mkdir(parents=True)
shutil.copytree(REPO / "duodiff_amd" / "csrc", dst)
(dst.parent.parent / "include").symlink_to(REPO / "include")
p = dst / "attention.hip"
s = p.read_text()


def rep(old, new):
    global s
    assert s.count(old) == 1, (old[:70], s.count(old))
    s = s.replace(old, new)


rep("namespace dd {\nnamespace {\n\nconstexpr int kMaxKeyTiles = 9;",
    "namespace dd {\nnamespace {\n__device__ unsigned long long g_astamps[8192 * 8];\n"
    "#define DD_ST(v) unsigned long long v; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(v) :: \"memory\")\n"
    "#define DD_RT(v) unsigned long long v; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(v) :: \"memory\")\n\nconstexpr int kMaxKeyTiles = 9;")
rep("    using Lay = AttnLayout<bf16_t>;\n    constexpr int KS = D / 16, NS = KS / kQaKQ, NB = 3 * NS;", "    DD_ST(st0); DD_RT(rt0);\n    using Lay = AttnLayout<bf16_t>;\n    constexpr int KS = D / 16, NS = KS / kQaKQ, NB = 3 * NS;")
rep("#pragma unroll\n    for (int i = 0; i < 4; ++i) wq[i] = *reinterpret_cast<const bf16x8*>(ring + lane * 16 + i * 1024);", "    DD_ST(st1);\n#pragma unroll\n    for (int i = 0; i < 4; ++i) wq[i] = *reinterpret_cast<const bf16x8*>(ring + lane * 16 + i * 1024);")
rep("    // the extras' partial sums of this wave: lanes (l & 15) < E hold", "    DD_ST(st2);\n    // the extras' partial sums of this wave: lanes (l & 15) < E hold")
rep("    __syncthreads();     // the K / V images are complete\n", "    __syncthreads();     // the K / V images are complete\n    DD_ST(st3);\n")
rep("    // ---- the extra tokens' queries: all 8 waves together, wave w against key tile w (wave 7: and the 9th), merged through LDS\n", "    DD_ST(st4);\n    // ---- the extra tokens' queries: all 8 waves together, wave w against key tile w (wave 7: and the 9th), merged through LDS\n")
# end of kernel: after the final block of the extras chunk; find the closing of the kernel: the line '}\n\n}  // namespace\n\ntemplate <typename T>\nhipError_t launch_attention'
rep("            a.out[((long long)b * L + qi) * D + hh * kHD + d] = f2bf(num / den);\n        }\n    }\n}",
    "            a.out[((long long)b * L + qi) * D + hh * kHD + d] = f2bf(num / den);\n        }\n    }\n"
    "    DD_ST(st5); DD_RT(rt5);\n    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {\n        unsigned long long* o = g_astamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;\n"
    "        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = st5; o[6] = rt0; o[7] = rt5;\n    }\n}")
s = s.rstrip() + ("\n\nextern \"C\" int dd_dev_read_attn_stamps(unsigned long long* out, int n) {\n"
                  "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dd::g_astamps), (size_t)n * 8);\n}\n")
p.write_text(s)
print("wrote", p)
