#!/usr/bin/env python3
"""Pairs per accumulator (m) of k_miller_accumulate, scanned through the test build's hook: ms of the two Miller kernels at
2^logn pairs for every m.   python tools/scan_pairing_share.py [logn=16]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << logn
g1 = co.gen_bases("g1", 21, n, 16)
g2 = co.gen_bases("g2", 22, n, 16)
with pkg.Context([0], test_hooks=True) as ctx:
    ref = None
    for m in (0,) + tuple(int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '1,2,4,8'.split(','))):
        ctx.test_set_pairing(share=m)
        ctx.multi_pairing(g1[:96 * 256], g2[:192 * 256])
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); gt = ctx.multi_pairing(g1, g2); dt = time.perf_counter() - t0
            p = ctx.pairing_profile()
            if best is None or dt < best[0]: best = (dt, p)
        ref = ref or gt
        print(json.dumps({"m": m, "ms": best[0] * 1e3, "lines_ms": best[1]["lines_ms"], "accumulate_ms": best[1]["accumulate_ms"], "same": gt == ref}), flush=True)
