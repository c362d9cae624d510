#!/usr/bin/env python3
"""Run tools/bin/fetch_calib under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE in a second pass) and print, per access
pattern, counter bytes / true bytes.  Usage (on the GPU box):  python tools/fetch_calib.py [out.txt]"""
import csv
import glob
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
BIN = REPO / "tools" / "bin" / "fetch_calib"


def run(counter, outdir):
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--", str(BIN), "1024"],
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)[0]
    res = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            res.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return res


def main():
    true_bytes = 1024 << 20
    lines = ["FETCH_SIZE / WRITE_SIZE calibration on MI355X (gfx950), 1 GiB read once per kernel (buffer evicted from the",
             "Infinity Cache between kernels); counters are in KiB.  ratio = counter * 1024 / bytes actually read.", ""]
    fetch = run("FETCH_SIZE", "/tmp/fetch_calib_f")
    for k in ("ldsdma_linear", "ldsdma_8x128", "vgpr_dwordx4"):
        v = [x for n, xs in fetch.items() if k in n for x in xs]
        lines.append(f"{k:16s} FETCH_SIZE = {v[0]:12.0f} KiB   ratio to true bytes = {v[0] * 1024 / true_bytes:.3f}")
    wr = run("WRITE_SIZE", "/tmp/fetch_calib_w")
    v = [x for n, xs in wr.items() if "fill" in n for x in xs]
    lines.append(f"{'fill (4-B stores)':16s} WRITE_SIZE = {v[0]:12.0f} KiB   ratio to true bytes = {v[0] * 1024 / true_bytes:.3f} (1 GiB fill)")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(text + "\n")


if __name__ == "__main__":
    main()
