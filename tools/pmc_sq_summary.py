"""Per-kernel SQ utilisation and the clock under load from one rocprofv3 --pmc pass
(SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
 SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE).

    python tools/pmc_sq_summary.py <rocprofv3 output dir> [out.json] > table.txt

Per launch (means over the upper half of a kernel's launches by value = the full-size ones):
    mfma_clk      = SQ_VALU_MFMA_BUSY_CYCLES / (CUs x 4)    MFMA-pipe busy clocks per SIMD (the counter is in clocks: 32 per 32x32x16 bf16 MFMA)
    gui_clk       = GRBM_GUI_ACTIVE / 8                       clocks the chip was busy for this dispatch (rocprofv3 sums the 8 XCDs)
    sclk_mhz      = gui_clk / duration                        the clock the part held under this kernel (MI355X_MICROARCH.md, DVFS give-back:
                                                              reads up to a few % high on dispatches shorter than 0.3 ms, and profiled passes run
                                                              2-5 % slower than un-profiled ones)
    mfma_busy_launch = mfma_clk / gui_clk                     LAUNCH-level utilisation of the matrix pipe: busy clocks / clocks of the launch
    mfma_busy_wave   = mfma_clk / mean wave lifetime          (the round-3 figure: per resident wave, not per launch -- it exceeds the launch-level
                                                              one whenever short-lived waves pull the mean lifetime down; > 1 with 2 waves per SIMD)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
NSIMD = 256 * 4
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
print(f"{'kernel':60s} {'waves':>7s} {'us(pmc)':>8s} {'sclk MHz':>8s} {'mfma_clk':>9s} {'busy/launch':>11s} {'busy/wave':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'lds_conf':>8s}")
for k, c in sorted(acc.items(), key=lambda kv: -upper_mean(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if "SQ_WAVES" not in c:
        continue
    g = lambda n: upper_mean(c[n]) if n in c else float("nan")
    waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
    wave_clk = wc * 4 / max(waves, 1)
    mfma_clk = g("SQ_VALU_MFMA_BUSY_CYCLES") / NSIMD
    us = upper_mean(dur[k]) / 1e3 if dur[k] else None
    gui_clk = g("GRBM_GUI_ACTIVE") / 8 if "GRBM_GUI_ACTIVE" in c else None
    row = dict(waves=waves, wave_clk=wave_clk, mfma_clk_per_simd=mfma_clk,
               mfma_busy_frac_launch=(mfma_clk / gui_clk) if gui_clk else None,
               sclk_mhz_under_load=(gui_clk / us) if (gui_clk and us) else None, gui_active_clk=gui_clk,
               mfma_busy_frac_wave_lifetime=mfma_clk / wave_clk if wave_clk else 0.0,
               wait_any=g("SQ_WAIT_ANY") / wc if wc else 0.0, wait_inst=g("SQ_WAIT_INST_ANY") / wc if wc else 0.0,
               active=g("SQ_ACTIVE_INST_ANY") / wc if wc else 0.0,
               lds_conflict_per_lds_active=(g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")) if c.get("SQ_ACTIVE_INST_LDS") and g("SQ_ACTIVE_INST_LDS") else None,
               avg_us_under_pmc=us)
    out[k] = row
    lc = row["lds_conflict_per_lds_active"]
    nan = float("nan")
    print(f"{k[:60]:60s} {waves:7.0f} {(us or 0):8.1f} {(row['sclk_mhz_under_load'] or nan):8.0f} {mfma_clk:9.0f} {(row['mfma_busy_frac_launch'] if row['mfma_busy_frac_launch'] is not None else nan):11.3f} "
          f"{row['mfma_busy_frac_wave_lifetime']:9.2f} {row['wait_any']:8.2f} {row['wait_inst']:9.2f} {row['active']:7.2f} {(lc if lc is not None else nan):8.3f}")
if len(sys.argv) > 2:
    # the build the counters were collected on (bench.py quotes them only for that build)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from duodiff_amd import _lib
    out["_build_id"] = _lib.load().dd_build_id().decode()
    json.dump(out, open(sys.argv[2], "w"), indent=1)
