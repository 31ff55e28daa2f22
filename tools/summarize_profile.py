#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof) into profiles/<tag>_*: the kernel-trace stats CSV as is,
and per-kernel means of the PMC passes with the HBM traffic corrected as MI355X_MICROARCH.md §HBM prescribes
(FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads -> doubled)."""
import collections
import os
import csv
import glob
import json
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
what = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary` (G1, 2^20, c=16)"
def newest(pattern):   # gpurun merges runs into the same directory: take the latest file
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(f"{src}/stats/runc/*kernel_stats.csv"), f"profiles/{tag}_kernel_stats.csv")
out = collections.defaultdict(dict)
for name in ("fetch", "write", "sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{src}/{name}/runc/*counter_collection.csv"))):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, xs in v.items():
            out[k][c] = sum(xs) / len(xs)
            out[k][c + "_max"] = max(xs)
            out[k]["launches"] = len(xs)
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch_corrected"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        v["hbm_bytes_per_launch_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in v:
        v["gpu_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
json.dump({"source": "rocprofv3 --pmc passes of `" + what + " (means over all launches of a kernel in the run, warm-up launches included)",
           "kernels": out}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("wrote profiles/%s_kernel_stats.csv, profiles/%s_pmc_summary.json" % (tag, tag))
