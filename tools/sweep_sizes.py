#!/usr/bin/env python3
"""Size sweep on one MI355X: ms per MSM call (resident bases, device scalars) for G1 / G2 at 2^lo..2^hi, with the plan's
window size and phase times, and — with --scan-c — the same for every forced window size (to check the time model).
    python tools/sweep_sizes.py g1 12 24 [--scan-c | --scan-c=LO,HI]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
g = sys.argv[1] if len(sys.argv) > 1 else "g1"
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 22
scan = any(a.startswith("--scan-c") for a in sys.argv)   # --scan-c or --scan-c=LO,HI
c_lo, c_hi = 8, 22
for a in sys.argv:
    if a.startswith("--scan-c="):
        c_lo, c_hi = (int(x) for x in a.split("=")[1].split(","))
nmax = 1 << hi
bases = co.gen_bases(g, 77, nmax, 16)
scalars = co.gen_scalars(78, nmax)
d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
with pkg.Context([0]) as ctx:
    ctx.set_profile_level(2)   # every phase's events (the default records the accumulate kernel's interval only)
    ctx.set_bases(g, bases, nmax)
    for ln in range(lo, hi + 1):
        n = 1 << ln
        want = co.dlog_expected(g, scalars[:32 * n], 77, n)
        for c in ([0] + list(range(c_lo, c_hi + 1)) if scan else [0]):
            try:
                ctx.set_window_bits(c)
                r = ctx.msm_device(g, d.data_ptr(), n, 0)
            except Exception as e:
                continue
            reps = 5 if ln <= 20 else 2
            t0 = time.perf_counter()
            for _ in range(reps):
                r = ctx.msm_device(g, d.data_ptr(), n, 0)
            ms = (time.perf_counter() - t0) / reps * 1e3
            p = ctx.profile()
            ok = co.to_affine(g, r) == want
            print(json.dumps({"group": g, "log_n": ln, "forced_c": c, "c": p["window_bits"], "ms": round(ms, 3), "points_per_s": n / ms * 1e3, "ok": ok,
                              "sort": round(p["digits_ms"] + p["scatter_ms"], 3), "sched": round(p["scan_ms"], 3), "acc": round(p["accumulate_ms"], 3),
                              "reduce": round(p["reduce_ms"], 3), "combine": round(p["combine_ms"], 3), "host": round(p["host_fold_ms"], 3)}), flush=True)
        ctx.set_window_bits(0)
