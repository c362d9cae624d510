"""Per-kernel SQ utilisation from one rocprofv3 --pmc pass
(SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS).

    python tools/pmc_sq_summary.py <rocprofv3 output dir> [out.json] > table.txt

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * kernel cycles), kernel cycles from the launch duration at the clock
the counters imply (SQ_BUSY_CYCLES is per-SE; we use wave lifetime instead): per launch,
    wave_clk  = SQ_WAVE_CYCLES * 4 / SQ_WAVES      (mean lifetime of a wave in clocks; the counter ticks per 4 clocks)
    mfma_clk  = SQ_VALU_MFMA_BUSY_CYCLES / 1024    (MFMA-pipe busy clocks per SIMD; 256 CUs x 4 SIMDs)
    mfma_busy = mfma_clk / (wave_clk * rounds)     rounds = waves per SIMD slot actually used = SQ_WAVES / (1024 * waves resident per SIMD)
For the one-wave-per-SIMD kernels (fused MLP: 1028+ waves on 1024 SIMDs) the denominator is the kernel's own duration in
clocks, which is what the table prints as `mfma/dur` using the measured duration and the clock implied by wave_clk.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
clean = lambda n: re.sub(r"dd::|\(anonymous namespace\)::|unsigned short", "", n)
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for fn in files:
    for r in csv.DictReader(open(fn)):
        k = clean(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (k, r.get("Dispatch_Id"))
        if key not in seen and "Start_Timestamp" in r:
            seen.add(key)
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def upper_mean(v):       # the full-size launches: upper half by value
    v = sorted(v)
    v = v[len(v) // 2:]
    return sum(v) / len(v)


out = {}
print(f"{'kernel':60s} {'waves':>7s} {'wave_clk':>9s} {'mfma_clk':>9s} {'mfma/wave':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'lds_conf':>8s} {'us(pmc)':>8s}")
for k, c in sorted(acc.items(), key=lambda kv: -upper_mean(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if "SQ_WAVES" not in c:
        continue
    g = lambda n: upper_mean(c[n]) if n in c else float("nan")
    waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
    wave_clk = wc * 4 / max(waves, 1)
    mfma_clk = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024
    row = dict(waves=waves, wave_clk=wave_clk, mfma_clk_per_simd=mfma_clk, mfma_busy_frac=mfma_clk / wave_clk if wave_clk else 0.0,
               wait_any=g("SQ_WAIT_ANY") / wc if wc else 0.0, wait_inst=g("SQ_WAIT_INST_ANY") / wc if wc else 0.0,
               active=g("SQ_ACTIVE_INST_ANY") / wc if wc else 0.0,
               lds_conflict_per_lds_active=(g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")) if c.get("SQ_ACTIVE_INST_LDS") and g("SQ_ACTIVE_INST_LDS") else None,
               avg_us_under_pmc=(upper_mean(dur[k]) / 1e3 if dur[k] else None))
    out[k] = row
    lc = row["lds_conflict_per_lds_active"]
    print(f"{k[:60]:60s} {waves:7.0f} {wave_clk:9.0f} {mfma_clk:9.0f} {row['mfma_busy_frac']:9.2f} {row['wait_any']:8.2f} {row['wait_inst']:9.2f} "
          f"{row['active']:7.2f} {(lc if lc is not None else float('nan')):8.3f} {(row['avg_us_under_pmc'] or 0):8.1f}")
if len(sys.argv) > 2:
    # the build the counters were collected on (bench.py quotes them only for that build)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from duodiff_amd import _lib
    out["_build_id"] = _lib.load().dd_build_id().decode()
    json.dump(out, open(sys.argv[2], "w"), indent=1)
