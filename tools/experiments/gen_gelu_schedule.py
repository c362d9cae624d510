#!/usr/bin/env python3
"""Prints the 16 GELU slot macros of duodiff_amd/csrc/mlp_fused.hip (DD_S_Kn / DD_O_Kn / DD_I_Kn).

Two register pairs of the S accumulator (A, B: 4 values) are evaluated together over 16 MFMA gaps: 12 packed-fp32 VALU instructions per
pair, 24 in all, dealt 2, 1, 2, 1 ... over the gaps so that the two instructions of a gap belong to different chains (a lone wave issues a
dependent VALU instruction ~9.5 clocks after its producer, an independent one after ~5: tools/experiments/pk_rate.hip).
"""
COEF = {"c6": "0x331d7172u", "c4": "0x387e87acu", "c3": "0xba743309u", "c2": "0x3c18a4c9u", "c1": "0xbd869818u", "c0": "0x3ecc1f5cu"}


def op(name, X):
    r = f"r{X}"
    S = (f"s{X}", f'"v"(f32x2{{{r}.sa, {r}.sb}})')
    if name == "m1":
        return f"v_med3_f32 %[sa{X}], %[va{X}], %[kn], %[kh]", [(f"sa{X}", f'"=&v"({r}.sa)')], [(f"va{X}", f'"v"(va{X})'), ("kn", '"s"(-3.8f)'), ("kh", '"v"(kh)')]
    if name == "m2":
        return f"v_med3_f32 %[sb{X}], %[vb{X}], %[kn], %[kh]", [(f"sb{X}", f'"=&v"({r}.sb)')], [(f"vb{X}", f'"v"(vb{X})'), ("kn", '"s"(-3.8f)'), ("kh", '"v"(kh)')]
    if name == "sq":
        return f"v_pk_mul_f32 %[s2{X}], %[s{X}], %[s{X}]", [(f"s2{X}", f'"=&v"({r}.s2)')], [S]
    if name == "f6":
        return (f"v_pk_fma_f32 %[p{X}], %[s2{X}], %[c6], %[kc] op_sel:[0,0,1] op_sel_hi:[1,1,1]", [(f"p{X}", f'"=&v"({r}.p)')],
                [(f"s2{X}", f'"v"({r}.s2)'), ("c6", f'"s"(splat2({COEF["c6"]}))'), ("kc", '"v"(k.hc)')])
    if name in ("f4", "f3", "f2", "f1", "f0"):
        c = "c" + name[1]
        return f"v_pk_fma_f32 %[p{X}], %[p{X}], %[s2{X}], %[{c}]", [(f"p{X}", f'"+v"({r}.p)')], [(f"s2{X}", f'"v"({r}.s2)'), (c, f'"s"(splat2({COEF[c]}))')]
    if name == "hh":
        return f"v_pk_fma_f32 %[h{X}], %[s{X}], %[p{X}], 0.5 op_sel_hi:[1,1,0]", [(f"h{X}", f'"=&v"({r}.h)')], [S, (f"p{X}", f'"v"({r}.p)')]
    if name == "hv":
        return f"v_pk_mul_f32 %[h{X}], %[h{X}], %[vv{X}]", [(f"h{X}", f'"+v"({r}.h)')], [(f"vv{X}", f'"v"(vv{X})')]
    if name == "cv":
        return f"v_cvt_pk_bf16_f32 %[out{X}], %[ha{X}], %[hb{X}]", [(f"out{X}", f'"=&v"(out{X})')], [(f"ha{X}", f'"v"(ha{X})'), (f"hb{X}", f'"v"(hb{X})')]
    raise KeyError(name)


chain = ["m1", "m2", "sq", "f6", "f4", "f3", "f2", "f1", "f0", "hh", "hv", "cv"]
order = [("A", "m1"), ("A", "m2"), ("B", "m1"), ("B", "m2")]
for n in chain[2:]:
    order += [("A", n), ("B", n)]
assert len(order) == 24
slots = [[] for _ in range(16)]
for i, o in enumerate(order):
    slots[i * 16 // 24].append(o)
for k, sl in enumerate(slots):
    strs, outs, ins = [], [], []
    for X, n in sl:
        s, o, i = op(n, X)
        strs.append(s)
        outs += [x for x in o if x not in outs]
        ins += [x for x in i if x not in ins]
    assert len({n for n, _ in outs} & {n for n, _ in ins}) == 0
    print(f'#define DD_S_K{k} "' + "".join("\\n\\t" + s for s in strs) + f'"     // ' + ", ".join(f"{X}.{n}" for X, n in sl))
    print(f"#define DD_O_K{k} " + ", ".join(f"[{n}] {c}" for n, c in outs))
    print(f"#define DD_I_K{k} " + ", ".join(f"[{n}] {c}" for n, c in ins))
