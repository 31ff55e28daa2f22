#!/usr/bin/env python3
"""Same-box A/B of the window-group pipeline (mi_msm_set_pipeline): ms per MSM call (resident bases, device scalars) for a list of
group weightings at a list of sizes, every result checked against the closed form.
    python tools/pipe_scan.py g1 18,20,22 "off;auto;1,1;1,1,1;1,1,1,1;3,5,5,3" [--validated] [--c=LO,HI] [--threads2]
"off" = one group (the pipeline of rounds 1-5), "auto" = the built-in choice.  --c scans forced window sizes as well (0 = the plan's).
--threads2 additionally times two host threads calling concurrently (ms per MSM)."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
g = args[0] if args else "g1"
sizes = [int(x) for x in (args[1] if len(args) > 1 else "20").split(",")]
confs = (args[2] if len(args) > 2 else "off;auto").split(";")
validated = "--validated" in flags
cs = [0]
for a in flags:
    if a.startswith("--c="):
        lo, hi = (int(x) for x in a.split("=")[1].split(","))
        cs = [0] + list(range(lo, hi + 1))
nmax = 1 << max(sizes)
bases = co.gen_bases(g, 77, nmax, 16)
scalars = co.gen_scalars(78, nmax)
d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
wants = {ln: co.dlog_expected(g, scalars[:32 * (1 << ln)], 77, 1 << ln) for ln in sizes}


def weights(conf):
    return None if conf == "auto" else [1] if conf == "off" else [int(x) for x in conf.split(",")]


with pkg.Context([0]) as ctx:
    ctx.set_bases(g, bases, nmax)
    if "--level2" in flags:
        ctx.set_profile_level(2)
    if validated:
        assert ctx.validate_bases(g) == 0
    for ln in sizes:
        n = 1 << ln
        for c in cs:
            for conf in confs:
                try:
                    ctx.set_window_bits(c)
                    ctx.set_pipeline(weights(conf))
                    for _ in range(3):
                        r = ctx.msm_device(g, d.data_ptr(), n, 0)
                except Exception as e:
                    print(json.dumps({"group": g, "log_n": ln, "forced_c": c, "pipeline": conf, "error": str(e)[:100]}), flush=True)
                    continue
                reps = 15 if ln <= 20 else 7 if ln <= 22 else 4
                times, acc = [], []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    r = ctx.msm_device(g, d.data_ptr(), n, 0)
                    times.append((time.perf_counter() - t0) * 1e3)
                    acc.append(ctx.profile()["accumulate_ms"])
                p = ctx.profile()
                row = {"group": g, "log_n": ln, "forced_c": c, "pipeline": conf, "groups": p["window_groups"], "c": p["window_bits"],
                       "windows": p["num_windows"], "ms": round(sum(times) / reps, 3), "ms_min": round(min(times), 3),
                       "acc_span_ms": round(sum(acc) / reps, 3), "host_fold_ms": round(p["host_fold_ms"], 3),
                       "ok": co.to_affine(g, r) == wants[ln], "validated": validated,
                       "items": p["work_items"], "max_items": p["max_items_per_bucket"], "adds": p["accumulate_adds"]}
                if "--threads2" in flags:
                    k = max(4, reps)
                    def worker():
                        for _ in range(k):
                            ctx.msm_device(g, d.data_ptr(), n, 0)
                    ts = [threading.Thread(target=worker) for _ in range(2)]
                    t0 = time.perf_counter()
                    for t in ts: t.start()
                    for t in ts: t.join()
                    row["ms_two_threads"] = round((time.perf_counter() - t0) * 1e3 / (2 * k), 3)
                print(json.dumps(row), flush=True)
        ctx.set_window_bits(0)
        ctx.set_pipeline(None)
