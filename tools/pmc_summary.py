"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), gfx950 corrections applied:
counters are in KiB; FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM)."""
import csv, re, sys, glob, os, collections, json
fd, wd = sys.argv[1], sys.argv[2]
clean = lambda n: re.sub(r"dd::|\(anonymous namespace\)::|unsigned short", "", n)
def load(d, counter):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[clean(r["Kernel_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {}
print(f"{'kernel':64s} {'calls':>6s} {'fetch MB (x2 corr.)':>20s} {'write MB':>10s} {'avg us':>8s}")
for k in sorted(F, key=lambda k: -sum(v for v, _ in F[k])):
    if k not in W: continue
    f = [v for v, _ in F[k]]; w = [v for v, _ in W[k]]; t = [d for _, d in F[k]]
    # per-launch medians of the big launches (B=128 full-size): take the upper half by value
    f.sort(); w.sort()
    fm = f[len(f)//2:] ; wm = w[len(w)//2:]
    fetch_mb = 2.0 * (sum(fm)/len(fm)) * 1024 / 1e6
    write_mb = (sum(wm)/len(wm)) * 1024 / 1e6
    out[k] = dict(calls=len(f), fetch_MB_corrected=fetch_mb, write_MB=write_mb, avg_us=sum(t)/len(t)/1e3)
    print(f"{k[:64]:64s} {len(f):6d} {fetch_mb:20.1f} {write_mb:10.1f} {sum(t)/len(t)/1e3:8.1f}")
if len(sys.argv) > 3:
    # the build the counters were collected on (bench.py quotes them only for that build)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from duodiff_amd import _lib
    out["_build_id"] = _lib.load().dd_build_id().decode()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
