#!/usr/bin/env python3
"""GPU side of the call-ABI bisect: for every variant test library under ab_libs/call_abi/ (tools/call_abi/build_variants.sh)
run the single-lane Miller kernel against the production path — whole batches and pair by pair — one subprocess per library.

    python tools/call_abi/run_variants.py [variant ...]  > gpurun_out/call_abi.jsonl
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = os.path.join(ROOT, "ab_libs", "call_abi")

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import coracle as co
import ark_blst_amd.binding as b
b.lib_path = lambda test_hooks=False: %(lib)r if test_hooks else os.path.join(%(root)r, "ark-blst_amd", "lib", "libarkblst_amd.so")
n = 70
g1 = co.gen_bases("g1", 79, n, 2); g2 = co.gen_bases("g2", 80, n, 2)
out = {"variant": %(name)r, "batches": {}, "bad_pairs": []}
with pkg.Context([0]) as ctx, pkg.Context([0], test_hooks=True) as t:
    for m in (1, 2, 7, 63, 64, 65, 70):
        want = ctx.multi_pairing(g1[:96 * m], g2[:192 * m])
        t.test_set_pairing(single_lane=True)
        got = t.multi_pairing(g1[:96 * m], g2[:192 * m])
        t.test_set_pairing()
        out["batches"][m] = got == want
    for i in range(n):
        p, q = g1[96 * i:96 * i + 96], g2[192 * i:192 * i + 192]
        want = ctx.multi_pairing(p, q)
        t.test_set_pairing(single_lane=True)
        got = t.multi_pairing(p, q)
        t.test_set_pairing()
        if got != want:
            out["bad_pairs"].append(i)
print(json.dumps(out))
'''

def main():
    names = sys.argv[1:] or sorted(f[4:-3] for f in os.listdir(LIBS) if f.startswith("lib_") and f.endswith(".so"))
    for name in names:
        code = CHILD % {"root": ROOT, "lib": os.path.join(LIBS, f"lib_{name}.so"), "name": name}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"variant": name, "error": r.stderr[-400:]})
        print(line, flush=True)

if __name__ == "__main__":
    main()
