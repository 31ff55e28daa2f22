#!/usr/bin/env python3
"""Big-integer check of the values tools/ubench_fp52 leaves behind: per line  bits iters  x-limbs y-limbs out-limbs  (hex), the kernel
computed x <- x * y / R mod p `iters` times with R = 2^(bits * limbs).  Every output must be congruent and below 2p."""
import sys

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
ok = bad = 0
for line in open(sys.argv[1]):
    f = line.split()
    bits, iters = int(f[0]), int(f[1])
    nl = 8 if bits == 52 else 14
    v = [int(h, 16) for h in f[2:]]
    val = lambda limbs: sum(l << (bits * k) for k, l in enumerate(limbs))
    x, y, out = val(v[:nl]), val(v[nl:2 * nl]), val(v[2 * nl:3 * nl])
    rinv = pow(1 << (bits * nl), -1, P)
    want = x * pow(y * rinv, iters, P) % P
    if out % P == want and out < 2 * P and all(l < (1 << bits) + 64 for l in v[2 * nl:3 * nl - 1]):
        ok += 1
    else:
        bad += 1
        print("MISMATCH bits", bits, hex(out % P), hex(want))
print(f"fp52 / fp28 chains checked against big integers: {ok} ok, {bad} bad")
sys.exit(1 if bad or not ok else 0)
