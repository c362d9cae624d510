"""Summarise a rocprofv3 --pmc csv directory: mean counter value per kernel name (substring filter argv[2])."""
import csv, glob, sys, collections
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm256")
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
