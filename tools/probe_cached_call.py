#!/usr/bin/env python3
"""Where do the ~0.2 ms between the resident-bases call and the cached host-bases call go?  Alternating calls, medians."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
n = 1 << 20
bases = co.gen_bases("g1", 5, n, 16)
sc = co.gen_scalars(6, n)
with pkg.Context([0]) as c:
    c.set_bases("g1", bases, n)
    c.set_base_cache(2)
    c.msm("g1", bases, sc, n, 0)
    c.msm("g1", None, sc, n, 0)
    t = {"resident": [], "cached": []}
    for rep in range(15):
        for name, b in (("resident", None), ("cached", bases)):
            t0 = time.perf_counter(); c.msm("g1", b, sc, n, 0); t[name].append((time.perf_counter() - t0) * 1e3)
    for k, v in t.items():
        print(k, "min %.3f median %.3f" % (min(v), statistics.median(v)), c.profile()["total_ms"])
    print(c.base_cache_stats())
