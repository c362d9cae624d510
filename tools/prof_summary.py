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
idx = [i for i, r in enumerate(tr) if "embed_" in r["Kernel_Name"]]
seq = tr[idx[-2]:idx[-1]]
t0 = int(seq[0]["Start_Timestamp"])
print(f"== last full step: {len(seq)} kernels, span {(int(seq[-1]['End_Timestamp'])-t0)/1e3:.1f} us, busy {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seq)/1e3:.1f} us")
for r in seq[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {clean(r['Kernel_Name'])[:70]}")
