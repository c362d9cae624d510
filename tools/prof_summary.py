"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel stats and one full-model step timeline."""
import csv, re, sys, glob, os
d = sys.argv[1]
stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
trace = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
clean = lambda n: re.sub(r"dd::|\(anonymous namespace\)::|unsigned short", "", n)
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== kernel stats (%s)" % os.path.basename(stats))
for r in rows[:14]:
    print(f"{clean(r['Name'])[:86]:86s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
tr = list(csv.DictReader(open(trace)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
# the same kernel runs at two grid sizes when dd_sample splits the batch into two half-batch chains (the timed loop: half-batch launches,
# two chains side by side) next to bench.py's stand-alone roofline leg (full-batch launches): one line per (kernel, workgroups)
import collections
by = collections.defaultdict(list)
for r in tr:
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
    by[(clean(r["Kernel_Name"])[:70], wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("== per (kernel, workgroups per launch)")
for (n, wg), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:20]:
    print(f"{n:70s} wgs={wg:6d} calls={len(v):6d} avg_us={sum(v) / len(v) / 1e3:8.1f} total_ms={sum(v) / 1e6:8.2f}")
idx = [i for i, r in enumerate(tr) if "embed_" in r["Kernel_Name"]]
seq = tr[idx[-2]:idx[-1]]
t0 = int(seq[0]["Start_Timestamp"])
print(f"== last full step: {len(seq)} kernels, span {(int(seq[-1]['End_Timestamp'])-t0)/1e3:.1f} us, busy {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seq)/1e3:.1f} us")
for r in seq[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {clean(r['Kernel_Name'])[:70]}")
