#!/usr/bin/env python3
"""Timing of the drop-in call shapes of mi_msm_g1 at 2^logn points: (a) bases + scalars from host memory on every call
(what the reference's driver does, /root/reference/src/gpu.rs:149-150), (b) resident bases + host scalars,
(c) resident bases + scalars already in HBM (bench.py's headline)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
bases = co.gen_bases("g1", 5, n, 16)
scalars = co.gen_scalars(6, n)
want = co.dlog_expected("g1", scalars, 5, n)
out = {"n": n}
with pkg.Context([0]) as ctx:
    ctx.set_profile_level(2)   # every phase's events
    def timeit(fn, reps=5):
        fn(); best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
        assert co.to_affine("g1", r) == want
        return best * 1e3, ctx.profile()
    ms, p = timeit(lambda: ctx.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL))
    out["host_bases_host_scalars_ms"] = ms; out["h2d_ms"] = p["h2d_ms"]; out["ingest_ms"] = p["ingest_ms"]
    ctx.set_bases("g1", bases, n)
    ms, p = timeit(lambda: ctx.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL))
    out["resident_bases_host_scalars_ms"] = ms; out["h2d_scalars_ms"] = p["h2d_ms"]
    d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
    ms, p = timeit(lambda: ctx.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL))
    out["resident_bases_device_scalars_ms"] = ms
print(json.dumps(out))
