"""CPU ORACLE (test infrastructure only) — textbook optimal-ate pairing on BLS12-381 with Python big ints.

NOT PRODUCT CODE (same rule as oracle/bls12_381.py: only tests/, smoke() and bench tools may import it).

Restates what `<Bls12 as Pairing>::multi_miller_loop` + `final_exponentiation` compute
(/root/reference/src/pairing.rs:49-74, 76-80 — both forwarded to blstrs/blst, which are absent from
/root/reference), deliberately in a DIFFERENT way from the shipped code so that agreement means something:

  * Fp12 is the flat extension Fp2[w]/(w^6 - xi), xi = 1 + u, as a list of six Fp2 coefficients (the shipped code
    uses the 2-3-2 tower with Karatsuba and lazy reduction);
  * the Miller loop walks the G2 point in AFFINE coordinates on the twist, with the exact untwisted line
        l(P) = yP - lambda' xP w^-1 + (lambda' xT - yT) w^-3         (psi(x', y') = (x' w^-2, y' w^-3), M-type twist)
    (the shipped code uses homogeneous projective coordinates and lines scaled by subfield elements);
  * the final exponentiation is ONE big-integer power f^(3 (p^12 - 1) / r) (the shipped code uses the
    Frobenius / cyclotomic chain of Hayashida-Hayasaka-Teruya, eprint 2020/875).

The factor 3: blst's hard part is (z-1)^2 (z+p) (z^2+p^2-1) + 3 = 3 (p^4 - p^2 + 1)/r (checked in selfcheck()), i.e.
blst / blstrs / arkworks' Gt is the cube of the "plain" reduced ate pairing; this oracle follows that convention.

PINNING STATUS: **parity unpinned**.  The reference's only pairing test is bilinearity e(sP, Q) = e(P, sQ)
(/root/reference/src/pairing.rs:92-101); it holds no Gt known-answer vector.  selfcheck() proves bilinearity,
non-degeneracy and e(P, Q)^r = 1 for this oracle.

Coefficient order of the reference's Fp12 (blst_fp12 = fp6[2], fp6 = fp2[3], fp2 = fp[2]; /root/reference/src/fp12.rs):
tower element (c0 + c1 W) with c_i = (a + b V + c V^2), V = W^2  <->  flat  a0 + a1 w + ... + a5 w^5 with
c0 = (a0, a2, a4), c1 = (a1, a3, a5).
"""

from __future__ import annotations

from . import bls12_381 as o

P = o.P
R = o.R_ORDER
Z = o.X_PARAM                      # negative
F2 = o.F2
XI = (1, 1)
XI_INV = F2.inv(XI)

FP12_ONE = [(1, 0)] + [(0, 0)] * 5

# exponent of the final exponentiation in blst's convention
HARD3 = (Z - 1) ** 2 * (Z + P) * (Z * Z + P * P - 1) + 3
FINAL_EXP = (P ** 6 - 1) * (P ** 2 + 1) * HARD3


def fp12_mul(a, b):
    """schoolbook product of two degree-5 polynomials in w over Fp2, reduced by w^6 = xi"""
    t = [(0, 0)] * 11
    for i in range(6):
        if a[i] == (0, 0):
            continue
        for j in range(6):
            t[i + j] = F2.add(t[i + j], F2.mul(a[i], b[j]))
    return [F2.add(t[k], F2.mul(XI, t[k + 6])) if k < 5 else t[k] for k in range(6)]


def fp12_pow(a, e: int):
    r = list(FP12_ONE)
    for bit in bin(e)[2:]:
        r = fp12_mul(r, r)
        if bit == "1":
            r = fp12_mul(r, a)
    return r


def fp12_eq(a, b) -> bool:
    return all(F2.eq(x, y) for x, y in zip(a, b))


def _line(lam, xt, yt, p1):
    """exact line through the untwisted point with twist-slope lam, evaluated at P = (xP, yP) in G1"""
    xp, yp = p1
    c = [(0, 0)] * 6
    c[0] = (yp % P, 0)
    # w^-1 = w^5 / xi, w^-3 = w^3 / xi
    c[5] = F2.mul(F2.neg(F2.mul(lam, (xp, 0))), XI_INV)
    c[3] = F2.mul(F2.sub(F2.mul(lam, xt), yt), XI_INV)
    return c


def miller_loop(p1, q2):
    """f_{|z|, Q}(P), conjugated for the negative parameter; P in G1 affine, Q in G2 affine (twist); None = infinity"""
    if p1 is None or q2 is None:
        return list(FP12_ONE)
    f = list(FP12_ONE)
    xt, yt = q2
    n = -Z
    for bit in bin(n)[3:]:
        lam = F2.mul(F2.mul((3, 0), F2.mul(xt, xt)), F2.inv(F2.mul((2, 0), yt)))
        f = fp12_mul(fp12_mul(f, f), _line(lam, xt, yt, p1))
        x3 = F2.sub(F2.mul(lam, lam), F2.add(xt, xt))
        yt = F2.sub(F2.mul(lam, F2.sub(xt, x3)), yt)
        xt = x3
        if bit == "1":
            lam = F2.mul(F2.sub(yt, q2[1]), F2.inv(F2.sub(xt, q2[0])))
            f = fp12_mul(f, _line(lam, xt, yt, p1))
            x3 = F2.sub(F2.sub(F2.mul(lam, lam), xt), q2[0])
            yt = F2.sub(F2.mul(lam, F2.sub(xt, x3)), yt)
            xt = x3
    # z < 0: f_{z} = 1 / f_{|z|} up to a vertical line; after the easy part the inverse is the conjugate
    # (w -> -w, i.e. odd coefficients negated)
    return [c if k % 2 == 0 else F2.neg(c) for k, c in enumerate(f)]


def final_exponentiation(f):
    return fp12_pow(f, FINAL_EXP)


def multi_miller_loop(ps, qs):
    f = list(FP12_ONE)
    for p1, q2 in zip(ps, qs):
        f = fp12_mul(f, miller_loop(p1, q2))
    return f


def pairing(p1, q2):
    return final_exponentiation(miller_loop(p1, q2))


# ---- the reference's in-memory form (12 x blst_fp, Montgomery) --------------------------------------------------
_FLAT_OF_TOWER = [0, 2, 4, 1, 3, 5]   # tower slot (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) -> flat power of w


def fp12_to_bytes(a) -> bytes:
    out = b""
    for k in _FLAT_OF_TOWER:
        out += o.fp_to_mont_bytes(a[k][0]) + o.fp_to_mont_bytes(a[k][1])
    return out


def fp12_from_bytes(b: bytes):
    a = [None] * 6
    for slot, k in enumerate(_FLAT_OF_TOWER):
        a[k] = (o.fp_from_mont_bytes(b[96 * slot:96 * slot + 48]), o.fp_from_mont_bytes(b[96 * slot + 48:96 * slot + 96]))
    return a


def selfcheck() -> None:
    assert HARD3 == 3 * ((P ** 4 - P ** 2 + 1) // R) and (P ** 4 - P ** 2 + 1) % R == 0
    g1, g2 = o.G1_GEN, o.G2_GEN
    e = pairing(g1, g2)
    assert not fp12_eq(e, FP12_ONE), "degenerate"
    assert fp12_eq(fp12_pow(e, R), FP12_ONE), "order"
    s = 0x1234567890ABCDEF1234567
    sp = o.scalar_mul(o.F1, g1, s)
    sq = o.scalar_mul(o.F2, g2, s)
    left, right = pairing(sp, g2), pairing(g1, sq)
    assert fp12_eq(left, right), "bilinearity"          # the reference's own test, src/pairing.rs:92-101
    assert fp12_eq(left, fp12_pow(e, s)), "e(sP, Q) = e(P, Q)^s"
    assert fp12_eq(fp12_from_bytes(fp12_to_bytes(e)), e)


if __name__ == "__main__":
    selfcheck()
    print("pairing oracle selfcheck OK")
