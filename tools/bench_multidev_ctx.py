import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
n = 1 << 21
bases = co.gen_bases("g1", 5, n, 16); sc = co.gen_scalars(6, n)
want = co.dlog_expected("g1", sc, 5, n)
for devs in ([0], [0, 0], [0, 0, 0, 0]):
    with pkg.Context(devs) as ctx:
        ctx.set_bases("g1", bases, n)
        r = ctx.msm("g1", None, sc, n); best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = ctx.msm("g1", None, sc, n); best = min(best, time.perf_counter() - t0)
        assert co.to_affine("g1", r) == want
        print(len(devs), "device slots:", round(best * 1e3, 2), "ms", ctx.profile()["host_fold_ms"])
