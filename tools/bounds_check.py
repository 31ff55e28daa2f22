#!/usr/bin/env python3
"""Static bound check of the lazily reduced device formulas (ark-blst_amd/csrc/ec.cuh).

Every device field value is kept in N-form (14 limbs <= 2^28+15 after fp_norm1) and only its VALUE is lazy.
This script replays each formula with value bounds (in units of p) and asserts the two preconditions:
  mul(a, b):  a*b < 2^392 * p   (Montgomery reduction then returns < 2p)
  sub<K>(a, b):  b <= (K-1) p   (the spread constant S_K dominates b limb-wise)
Run:  python tools/bounds_check.py
"""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
LIM = (1 << 392) / P  # ~2521


class V:
    def __init__(self, v, name=""):
        self.v = v
        self.name = name


def mul(a, b):
    assert a.v * b.v < LIM, f"mul overflow: {a.v} * {b.v} >= {LIM:.0f}"
    return V(2)


def mul2add(a, b, c, d):
    assert a.v * b.v + c.v * d.v < LIM, f"mul2add overflow: {a.v}*{b.v} + {c.v}*{d.v}"
    return V(2)


def add(a, b):
    r = V(a.v + b.v)
    assert r.v < LIM
    return r


def sub(K, a, b):
    assert b.v <= K - 1, f"sub<{K}>: subtrahend bound {b.v} > {K - 1}"
    return V(a.v + K)


def scale(k, a):
    r = V(k * a.v)
    assert r.v < LIM
    return r


def check_madd(X=10, Y=6, x2=4, y2=4):
    """xyzz_madd state machine; returns output bounds."""
    X, Y, ZZ, ZZZ, x2, y2 = V(X), V(Y), V(2), V(2), V(x2), V(y2)
    t0 = sub(16, mul(x2, ZZ), X)      # P
    t1 = sub(8, mul(y2, ZZZ), Y)      # R
    t2 = mul(t0, t0)                  # PP
    t0 = mul(t0, t2)                  # PPP
    ZZ = mul(ZZ, t2)
    t2 = mul(X, t2)                   # Q
    ZZZ = mul(ZZZ, t0)
    ny = sub(8, V(0), Y)              # 8p - Y1
    t3 = add(add(t0, t2), t2)         # PPP + 2Q
    X = sub(8, mul(t1, t1), t3)       # X3
    t2 = sub(16, t2, X)               # Q - X3
    Y = mul2add(t1, t2, ny, t0)       # Y3 = R (Q - X3) + (8p - Y1) PPP
    assert X.v <= 10 and Y.v <= 6 and ZZ.v <= 2 and ZZZ.v <= 2, (X.v, Y.v)
    return X.v, Y.v


def check_mdbl(x=4, y=4):
    x, y = V(x), V(y)
    U = add(y, y)
    Vv = mul(U, U)
    W = mul(U, Vv)
    S = mul(x, Vv)
    xx = mul(x, x)
    M = add(add(xx, xx), xx)
    X3 = sub(8, mul(M, M), add(S, S))
    v = sub(16, S, X3)
    Y3 = sub(4, mul(M, v), mul(W, y))
    assert X3.v <= 10 and Y3.v <= 6


def check_xyzz_to_proj(X=10, Y=6):
    X, Y, ZZ, ZZZ = V(X), V(Y), V(2), V(2)
    return mul(X, ZZZ).v, mul(Y, ZZ).v, mul(ZZ, ZZZ).v


def check_rcb_add(B=8):
    """Renes-Costello-Batina 2016 Alg. 7 (a = 0, b3 = 12), scheduled as 12 multiplications, the three outputs as fused two-product reductions."""
    X1, Y1, Z1, X2, Y2, Z2 = (V(B) for _ in range(6))
    t0 = mul(X1, X2)
    t1 = mul(Y1, Y2)
    t2 = mul(Z1, Z2)
    t3 = mul(add(X1, Y1), add(X2, Y2))
    t4 = mul(add(Y1, Z1), add(Y2, Z2))
    t5 = mul(add(X1, Z1), add(X2, Z2))
    t3 = sub(8, t3, add(t0, t1))
    t4 = sub(8, t4, add(t1, t2))
    t5 = sub(8, t5, add(t0, t2))
    t0 = scale(3, t0)
    t2 = scale(12, t2)
    Z3 = add(t1, t2)
    t1 = sub(32, t1, t2)
    t5 = scale(12, t5)
    u = Z3
    X3 = mul2add(t1, t3, t5, sub(16, V(0), t4))   # t1 t3 - t5 t4, fused: one reduction per output
    Y3 = mul2add(t1, u, t5, t0)
    Z3 = mul2add(u, t4, t0, t3)
    assert max(X3.v, Y3.v, Z3.v) <= B, (X3.v, Y3.v, Z3.v)
    return X3.v, Y3.v, Z3.v


# ------------------------------------------------------------------------------------------------
# limb-level check of the LAZY hot-loop mixed addition (FpOpsInline: no carry pass after add / sub)
E_LIMB = 1 << 28                     # exact multiplier output: limbs < 2^28
N_LIMB = (1 << 28) + 15              # after fp_norm1
SPREAD_LO = (1 << 28) + 64           # floor of S_K limbs
SPREAD_HI = SPREAD_LO + (1 << 28) + 4
SPREAD_B_LO = 3 * (1 << 28) + 64     # floor of S8B limbs
SPREAD_B_HI = SPREAD_B_LO + (1 << 28) + 4
COL_MAX = (1 << 64) - 14 * (1 << 56) - (1 << 37)   # room left in a 64-bit column for the a*b terms


class L:
    def __init__(self, v, l):
        self.v, self.l = v, l


def lmul(a, b):
    assert a.v * b.v < LIM
    assert 14 * a.l * b.l < COL_MAX, f"column overflow {a.l:#x} * {b.l:#x}"
    return L(2, E_LIMB)


def lmul2add(a, b, c, d):
    assert a.v * b.v + c.v * d.v < LIM
    assert 14 * (a.l * b.l + c.l * d.l) < COL_MAX, "column overflow in mul2add"
    return L(2, E_LIMB)


def ladd(a, b):
    assert a.l + b.l < 1 << 32
    return L(a.v + b.v, a.l + b.l)


def lsub(K, a, b, wide=False):
    lo, hi = (SPREAD_B_LO, SPREAD_B_HI) if wide else (SPREAD_LO, SPREAD_HI)
    assert b.l <= lo, f"subtrahend limbs {b.l:#x} above the spread floor {lo:#x}"
    assert b.v <= K - 1
    assert a.l + hi < 1 << 32
    return L(a.v + K, a.l + hi)


def lnorm(a):
    return L(a.v, (1 << 28) - 1 + (a.l >> 28) + 1)


def check_madd_lazy():
    """ec::xyzz_madd<FpOpsInline>: returns the limb bounds of the stored accumulator."""
    X, Y, ZZ, ZZZ = L(10, N_LIMB), L(6, N_LIMB), L(2, E_LIMB), L(2, E_LIMB)
    x2, y2 = L(2, E_LIMB), L(4, SPREAD_HI)            # y2 possibly negated lazily: S4 - y
    for _ in range(3):                                 # iterate to a fixed point of the invariants
        t0 = lsub(16, lmul(x2, ZZ), X)
        t2 = lmul(t0, t0)
        t1 = lsub(8, lmul(y2, ZZZ), Y)
        t0 = lmul(t0, t2)
        ZZ = lmul(ZZ, t2)
        t2 = lmul(X, t2)
        ZZZ = lmul(ZZZ, t0)
        ny = lsub(8, L(0, 0), Y)
        t3 = ladd(ladd(t0, t2), t2)
        X = lnorm(lsub(8, lmul(t1, t1), t3, wide=True))
        t2 = lsub(16, t2, X)
        Y = lmul2add(t1, t2, ny, t0)
        assert X.v <= 10 and X.l <= SPREAD_LO and Y.l <= SPREAD_LO and Y.v <= 6
    return X.l, Y.l


if __name__ == "__main__":
    print("limit a*b <", LIM)
    print("madd out bounds", check_madd())
    check_mdbl()
    print("xyzz->proj", check_xyzz_to_proj())
    print("rcb add out bounds", check_rcb_add())
    print("lazy madd limb bounds (X, Y): %#x %#x" % check_madd_lazy())
    print("all bounds OK")
