#!/usr/bin/env python3
"""Timing of mi_g1_normalize_batch (row (f)-2 of SURVEY §8) next to the oracle's per-point conversion."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
bases = co.gen_bases("g1", 5, n, 16)
one = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615")
# Jacobian inputs with non-trivial Z: P_i + P_{i+1} computed by the oracle for a sample, tiled
m = 4096
jac = b"".join(co.sum_jac("g1", bases[96 * i:96 * i + 96] + one + bases[96 * (i + 1):96 * (i + 2)] + one, 2) for i in range(m))
blob = jac * (n // m)
with pkg.Context([0]) as ctx:
    ctx.normalize_batch("g1", blob[:144 * 1000])
    t0 = time.perf_counter(); out = ctx.normalize_batch("g1", blob); wall = time.perf_counter() - t0
    p = ctx.profile()
    t0 = time.perf_counter(); ref = b"".join(co.to_affine("g1", blob[144 * i:144 * (i + 1)]) for i in range(2000)); cpu = (time.perf_counter() - t0) / 2000
    assert out[:96 * 2000] == ref
    print({"n": n, "gpu_kernels_ms": round(p["accumulate_ms"], 3), "h2d_ms": round(p["h2d_ms"], 3), "wall_ms": round(wall * 1e3, 2),
           "points_per_s_kernels": round(n / (p["accumulate_ms"] * 1e-3)), "oracle_cpu_us_per_point_1thread": round(cpu * 1e6, 2)})
