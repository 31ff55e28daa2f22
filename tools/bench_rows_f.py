#!/usr/bin/env python3
"""Rows (f) as the caller sees them: wall time of the C call (mi_profile.total_ms: host pointers in, host pointers out) next to the kernels'
time, for normalize_batch, deserialize_batch (validate on), check_batch, and the two one-upload SRS loaders.
    python tools/bench_rows_f.py [g1|g2] [log_n]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
g = _args[0] if _args else "g1"
ln = int(_args[1]) if len(_args) > 1 else 20
ONLY = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("--only=")]
ONLY = ONLY[0] if ONLY else None
n = 1 << ln
aff, jb, unit = (96, 144, 48) if g == "g1" else (192, 288, 96)
bases = co.gen_bases(g, 4242, n, 16)
one = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615")
one = one if g == "g1" else one + bytes(48)
m = 1 << 12
jac = b"".join(co.sum_jac(g, bases[aff * i:aff * (i + 1)] + one + bases[aff * (i + 1):aff * (i + 2)] + one, 2) for i in range(m)) * (n // m)
rows = {}
with pkg.Context([0]) as c:
    def timed(name, fn, reps=3):
        if ONLY and not any(name.startswith(o) for o in ONLY):
            return
        if "--once" in sys.argv:   # profiler runs: one warm call sizes the buffers, ONE profiled-size call follows
            reps = 1
        fn()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            wall = (time.perf_counter() - t0) * 1e3
            p = c.profile()
            if best is None or p["total_ms"] < best["call_ms"]:
                best = {"call_ms": round(p["total_ms"], 3), "kernels_ms": round(p["accumulate_ms"], 3), "h2d_ms": round(p["h2d_ms"], 3), "python_wall_ms": round(wall, 2)}
        rows[name] = best
    timed("normalize_batch", lambda: c.normalize_batch(g, jac))
    enc = c.serialize_batch(g, bases, True)
    timed("serialize_batch", lambda: c.serialize_batch(g, bases, True))
    timed("deserialize_batch_validate", lambda: c.deserialize_batch(g, enc, True, True), reps=2)
    timed("deserialize_batch_no_validate", lambda: c.deserialize_batch(g, enc, True, False), reps=2)
    timed("check_batch", lambda: c.check_batch(g, bases), reps=2)
    timed("set_bases", lambda: c.set_bases(g, bases, n))
    timed("set_bases_from_jacobian", lambda: c.set_bases_from_jacobian(g, jac, n))
    timed("set_bases_from_compressed_validate", lambda: c.set_bases_from_compressed(g, enc, n, True, True), reps=2)
    timed("set_bases_from_compressed_no_validate", lambda: c.set_bases_from_compressed(g, enc, n, True, False), reps=2)
print(json.dumps({"group": g, "log_n": ln, "rows": rows}, indent=None if "--oneline" in sys.argv else 1))
