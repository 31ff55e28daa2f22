#!/usr/bin/env python3
"""One-off parity check beyond one pass of the pipeline: G1 MSM over n = 2^26 + 12345 points (two passes in production, no test hook),
resident bases, host scalars, against the closed form; then the first 2^26 points in one pass.  ~2 minutes, 9 GB of host memory.
    python tools/big_call.py [extra_points=12345]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 12345
n = (1 << 26) + extra
t0 = time.time()
bases = co.gen_bases("g1", 2626, n, 16)
scalars = co.gen_scalars(2627, n)
gen_s = time.time() - t0
out = {"n": n, "gen_s": round(gen_s, 1)}
with pkg.Context([0]) as c:
    t0 = time.time(); c.set_bases("g1", bases, n); out["set_bases_s"] = round(time.time() - t0, 2)
    for m, tag in ((n, "two_passes"), (1 << 26, "one_pass")):
        t0 = time.time()
        got = c.msm("g1", None, scalars, m, pkg.SCALAR_CANONICAL)
        dt = time.time() - t0
        p = c.profile()
        ok = co.to_affine("g1", got) == co.dlog_expected("g1", scalars[:32 * m], 2626, m)
        out[tag] = {"points": m, "ms": round(dt * 1e3, 1), "points_per_s": m / dt, "window_bits": p["window_bits"], "bit_exact": ok}
        print(tag, out[tag], flush=True)
print(json.dumps(out))
sys.exit(0 if all(out[t]["bit_exact"] for t in ("two_passes", "one_pass")) else 1)
