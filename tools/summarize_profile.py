#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof_<tag>) into profiles/<tag>_*: the kernel-trace stats CSV as is,
the bench line printed under rocprof, and per-kernel means of the PMC passes with the HBM traffic corrected as
MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of
16-B-per-lane reads -> doubled).  The summary names its workload so that bench.py's roofline.traffic picks the matching one.
    python tools/summarize_profile.py <dir> <tag> <group> <log_n> [precomputed]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, group, log_n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
precomputed = len(sys.argv) > 5 and sys.argv[5] == "precomputed"
cmd = open(f"{src}/command.txt").read().strip() if os.path.exists(f"{src}/command.txt") else "bench.py"


def newest(pattern):   # gpurun merges runs into the same directory: take the latest file
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


shutil.copy(newest(f"{src}/stats/**/*kernel_stats.csv"), f"profiles/{tag}_kernel_stats.csv")
line = [l for l in open(f"{src}/bench_line_under_rocprof.json") if l.startswith("{")]
if line:
    open(f"profiles/{tag}_bench_line_under_rocprof.json", "w").write(line[-1])
out = collections.defaultdict(dict)
for name in ("fetch", "write", "sq", "stall", "stall2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        path = newest(f"{src}/{name}/**/*counter_collection.csv")
    except ValueError:
        continue   # optional pass (stall counters) not collected for this workload
    for r in csv.DictReader(open(path)):
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
        # the largest launch of the run (runs that also launch the kernel on small warm-up / sample inputs: the pairing bench)
        v["hbm_bytes_largest_launch_corrected"] = (2.0 * v["FETCH_SIZE_max"] + v["WRITE_SIZE_max"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in v:
        v["gpu_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
# shader clock of each kernel under its own load: cycles per launch (PMC pass) / average duration (kernel trace of the stats pass)
try:
    dur = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv"))}
    for k, v in out.items():
        if "gpu_cycles_per_launch" in v and k in dur and dur[k] > 0:
            v["avg_ns_in_kernel_trace"] = dur[k]
            v["shader_clock_ghz"] = v["gpu_cycles_per_launch"] / dur[k]
except Exception as e:
    print("no shader clock:", e)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # source_hash(): the tree this summary was measured on (the profile ran on a snapshot of it)
sha_file = f"{src}/source.sha256"
source_sha = open(sha_file).read().split()[0] if os.path.exists(sha_file) else bench.source_hash()
extra = {}
if "bench_rows_f.py" in cmd and "--once" in cmd:
    extra["calls_profiled"] = 2   # tools/bench_rows_f.py --once: one warm call sizes the buffers, one more of the same size follows (both full size)
json.dump({**extra, "source_sha256": source_sha, "source": f"rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*, one pass each) of `python3 {cmd}` "
                     "(means over all launches of a kernel in the run, warm-up launches included)",
           "workload": {"group": group, "log_n": log_n, "precomputed": precomputed},
           "kernels": out}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("wrote profiles/%s_kernel_stats.csv, profiles/%s_pmc_summary.json" % (tag, tag))
