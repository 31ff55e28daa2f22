#!/usr/bin/env python3
"""What the pipeline's own event records cost a call: ms per MSM call (resident bases, device scalars) at profile level 0 / 1 / 2.
    python tools/level_cost.py [g1|g2] [log_n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
g = sys.argv[1] if len(sys.argv) > 1 else "g1"
ln = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << ln
bases = co.gen_bases(g, 77, n, 16)
scalars = co.gen_scalars(78, n)
d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
with pkg.Context([0]) as ctx:
    ctx.set_bases(g, bases, n)
    for rep in range(2):
        for lvl in (0, 1, 2, 0, 1, 2):
            ctx.set_profile_level(lvl)
            for _ in range(5):
                ctx.msm_device(g, d.data_ptr(), n)
            t0 = time.perf_counter()
            for _ in range(40):
                ctx.msm_device(g, d.data_ptr(), n)
            print(f"level {lvl}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms per call")
