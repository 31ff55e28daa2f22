#!/usr/bin/env python3
"""Looks for performance cliffs of mi_msm_g1 around the headline shape (resident bases, scalars in HBM): odd sizes,
many infinity bases, narrow scalars, repeated points.  Prints ms per call for each case (all results checked)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
N = (1 << 20) + 12345
bases = co.gen_bases("g1", 5, N, 16)
scalars = co.gen_scalars(6, N)
res = {}
with pkg.Context([0]) as ctx:
    ctx.set_profile_level(2)   # every phase's events (the default records the accumulate kernel's interval only)
    def run(name, b, s, n, check=True):
        ctx.set_bases("g1", b, n)
        d = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
        r = ctx.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = ctx.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL); best = min(best, time.perf_counter() - t0)
        if check:
            assert co.to_affine("g1", r) == co.to_affine("g1", co.msm("g1", b, s, n, 0, 16)), name
        p = ctx.profile()
        res[name] = {"ms": round(best * 1e3, 3), "c": p["window_bits"], "acc": round(p["accumulate_ms"], 3), "red": round(p["reduce_ms"], 3),
                     "sort": round(p["digits_ms"] + p["scatter_ms"], 3), "items": p["work_items"], "max_items_per_bucket": p["max_items_per_bucket"]}
    run("2^20 + 12345 points", bases, scalars, N)
    n = 1 << 20
    run("2^20 baseline", bases[:96 * n], scalars[:32 * n], n)
    b = bytearray(bases[:96 * n]); a = np.frombuffer(b, dtype=np.uint8).reshape(n, 96); a[::2] = 0
    run("half the bases at infinity", bytes(b), scalars[:32 * n], n)
    s = np.frombuffer(scalars[:32 * n], dtype=np.uint8).reshape(n, 32).copy(); s[:, 4:] = 0
    run("32-bit scalars", bases[:96 * n], s.tobytes(), n)
    s = np.frombuffer(scalars[:32 * n], dtype=np.uint8).reshape(n, 32).copy(); s[:, 16:] = 0
    run("128-bit scalars", bases[:96 * n], s.tobytes(), n)
    rep = bases[:96 * 1024] * 1024
    run("1024 distinct points repeated", rep, scalars[:32 * n], n)
    s = np.frombuffer(scalars[:32 * n], dtype=np.uint8).reshape(n, 32).copy(); s[: n // 2] = s[0]
    run("half the scalars identical", bases[:96 * n], s.tobytes(), n)
print(json.dumps(res, indent=1))
