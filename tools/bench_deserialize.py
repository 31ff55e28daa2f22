#!/usr/bin/env python3
"""Timing of mi_g1_deserialize_batch (row (f)-4 of SURVEY §8): decompression + subgroup check for 2^logn points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import coracle as co, bls12_381 as o
pkg = ge.load_package()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
bases = co.gen_bases("g1", 5, n, 16)
with pkg.Context([0]) as ctx:
    enc = ctx.g1_serialize_batch(bases, True)
    for validate in (False, True):
        ctx.g1_deserialize_batch(enc[:48 * 1000], True, validate)
        t0 = time.perf_counter(); dec, st = ctx.g1_deserialize_batch(enc, True, validate); wall = time.perf_counter() - t0
        p = ctx.profile()
        assert st == bytes(n) and dec == bases
        print({"n": n, "validate": validate, "kernel_ms": round(p["accumulate_ms"], 2), "wall_ms": round(wall * 1e3, 1),
               "points_per_s_kernel": round(n / (p["accumulate_ms"] * 1e-3))})
    t0 = time.perf_counter()
    for i in range(20):
        o.g1_deserialize(enc[48 * i:48 * (i + 1)], True, True)
    print({"python_oracle_ms_per_point": round((time.perf_counter() - t0) / 20 * 1e3, 2)})
