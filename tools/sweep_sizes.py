#!/usr/bin/env python3
"""Size sweep on one MI355X: ms per MSM call (resident bases, device scalars) for G1 / G2 at 2^lo..2^hi, with the plan's
window size and phase times, and — with --scan-c — the same for every forced window size (to check the time model).
    python tools/sweep_sizes.py g1 12 24 [--scan-c | --scan-c=LO,HI] [--validated] [--test-hooks] [--no-phases]

Round 5: every (n, c) gets THREE warm-up calls before it is timed, and the plan's own pick (forced_c = 0) is
measured twice — before AND after the forced rows of its size — so that a cold first row can no longer make the plan look worse than
the same window size forced (2^21: 6.2-6.27 ms "auto" against 5.71 forced in the round-4 files).  --validated runs
mi_msm_g1_validate_bases first (the sign fold of a validated resident set); --no-phases times with the default profile level (the
per-phase events of level 2 idle the device ~6 us each: the `ms` of a small size is then slightly high, the phase columns are the
point of level 2); --test-hooks loads the test build (MI_TEST_* experiment switches of csrc/msm_sort.hip)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
g = args[0] if len(args) > 0 else "g1"
lo = int(args[1]) if len(args) > 1 else 12
hi = int(args[2]) if len(args) > 2 else 22
scan = any(a.startswith("--scan-c") for a in flags)   # --scan-c or --scan-c=LO,HI
c_lo, c_hi = 8, 22
for a in flags:
    if a.startswith("--scan-c="):
        c_lo, c_hi = (int(x) for x in a.split("=")[1].split(","))
validated, hooks, phases = "--validated" in flags, "--test-hooks" in flags, "--no-phases" not in flags
nmax = 1 << hi
bases = co.gen_bases(g, 77, nmax, 16)
scalars = co.gen_scalars(78, nmax)
d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
wants = {ln: co.dlog_expected(g, scalars[:32 * (1 << ln)], 77, 1 << ln) for ln in range(lo, hi + 1)}   # CPU work BEFORE the timed rows
with pkg.Context([0], test_hooks=hooks) as ctx:
    if phases:
        ctx.set_profile_level(2)   # every phase's events (the default records the accumulate kernel's interval only)
    ctx.set_bases(g, bases, nmax)
    if validated:
        assert ctx.validate_bases(g) == 0
    for ln in range(lo, hi + 1):
        n = 1 << ln
        rows = [0] + (list(range(c_lo, c_hi + 1)) + [0] if scan else [])
        for k, c in enumerate(rows):
            try:
                ctx.set_window_bits(c)
                for _ in range(3):
                    r = ctx.msm_device(g, d.data_ptr(), n, 0)
            except Exception as e:
                continue
            reps = 7 if ln <= 18 else 5 if ln <= 21 else 3
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = ctx.msm_device(g, d.data_ptr(), n, 0)
                times.append((time.perf_counter() - t0) * 1e3)
            ms = sum(times) / reps
            p = ctx.profile()
            ok = co.to_affine(g, r) == wants[ln]
            print(json.dumps({"group": g, "log_n": ln, "forced_c": c, "auto_row": ("first" if k == 0 else "last") if c == 0 else None,
                              "c": p["window_bits"], "windows": p["num_windows"], "ms": round(ms, 3), "ms_min": round(min(times), 3),
                              "points_per_s": n / ms * 1e3, "ok": ok, "validated": validated,
                              "sort": round(p["digits_ms"] + p["scatter_ms"], 3), "sched": round(p["scan_ms"], 3), "acc": round(p["accumulate_ms"], 3),
                              "reduce": round(p["reduce_ms"], 3), "combine": round(p["combine_ms"], 3), "host": round(p["host_fold_ms"], 3)}), flush=True)
        ctx.set_window_bits(0)
