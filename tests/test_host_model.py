"""CPU-only check of the DEVICE arithmetic: ark-blst_amd/csrc/{fp28,ec}.cuh compile as plain C++ (g++), so the
radix-2^28 lazy field and the XYZZ / complete-projective formulas are pinned against the oracle without a GPU."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def h28():
    src = os.path.join(HERE, "host", "fp28_host_check.cpp")
    so = os.path.join(HERE, "host", "libfp28_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def test_bounds_checker():
    subprocess.check_call(["python3", os.path.join(HERE, "..", "tools", "bounds_check.py")])


def test_bounds_machine_checked_on_the_real_formulas():
    """ec.cuh's xyzz_madd / xyzz_to_proj / proj_add instantiated with a field class that carries value and limb bounds and
    enforces fp28.cuh's contracts at every call, iterated to a fixed point (tests/host/msm_bounds.cpp)"""
    exe = os.path.join(HERE, "host", "msm_bounds")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "host", "msm_bounds.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert "msm bounds OK" in out


def test_field_ops(h28, co, o):
    rnd = random.Random(1)
    n = 3000
    va = [rnd.randrange(o.P) for _ in range(n)]
    vb = [rnd.randrange(o.P) for _ in range(n)]
    edge = [0, 1, o.P - 1, (o.P - 1) // 2, 2, o.P - 2]
    va[:6] = edge
    vb[:6] = edge[::-1]
    a = b"".join(o.fp_to_mont_bytes(v) for v in va)
    b = b"".join(o.fp_to_mont_bytes(v) for v in vb)
    out = C.create_string_buffer(48 * n)
    h28.h28_roundtrip(a, out, C.c_size_t(n))
    assert out.raw == a
    h28.h28_fp_mul(a, b, out, C.c_size_t(n), 0)
    assert out.raw == co.fp_mul(a, b)
    h28.h28_fp_mul(a, b, out, C.c_size_t(n), 1)
    assert out.raw == co.fp_mul(a, a)
    oa, os_ = C.create_string_buffer(48 * n), C.create_string_buffer(48 * n)
    h28.h28_fp_addsub(a, b, oa, os_, C.c_size_t(n))
    assert oa.raw == b"".join(o.fp_to_mont_bytes((x + y) % o.P) for x, y in zip(va, vb))
    assert os_.raw == b"".join(o.fp_to_mont_bytes((x - y) % o.P) for x, y in zip(va, vb))
    h28.h28_fp_mul_small(a, oa, os_, C.c_size_t(n))
    assert oa.raw == b"".join(o.fp_to_mont_bytes(3 * x % o.P) for x in va)
    assert os_.raw == b"".join(o.fp_to_mont_bytes(12 * x % o.P) for x in va)


def _sum(o, pts):
    acc = None
    for p in pts:
        acc = o.aff_add(o.F1, acc, p)
    return acc


def test_bucket_hot_and_cold_paths(h28, co, o):
    rnd = random.Random(2)
    m = 40
    bases = co.gen_bases("g1", 7, m)
    pts = [o.affine_from_bytes(o.F1, bases[96 * i:96 * i + 96]) for i in range(m)]
    neg = bytes(rnd.randrange(2) for _ in range(m))
    oj = C.create_string_buffer(144)
    cold = C.c_int(0)
    h28.h28_g1_bucket(bases, neg, C.c_size_t(m), oj, C.byref(cold))
    want = _sum(o, [o.aff_neg(o.F1, p) if s else p for p, s in zip(pts, neg)])
    assert cold.value == 0
    assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, want)
    # exceptional pairs: duplicate (doubling), opposite (cancellation), repeated after cancellation
    P_, Q = pts[0], pts[1]
    nP, nQ = o.aff_neg(o.F1, P_), o.aff_neg(o.F1, Q)
    for seq in ([P_, P_], [P_, nP], [P_, Q, P_, P_, nQ, nP, nP, nP, Q, Q, Q], [P_, P_, P_, P_], [Q, P_, nP, nQ]):
        sb = b"".join(o.affine_to_bytes(o.F1, x) for x in seq)
        h28.h28_g1_bucket(sb, None, C.c_size_t(len(seq)), oj, C.byref(cold))
        assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, _sum(o, seq)), seq
    assert cold.value == 1


def test_complete_addition_tree_and_doubling(h28, co, o):
    m = 37
    bases = co.gen_bases("g1", 9, m)
    pts = [o.affine_from_bytes(o.F1, bases[96 * i:96 * i + 96]) for i in range(m)]
    oj = C.create_string_buffer(144)
    h28.h28_g1_add_tree(bases, C.c_size_t(m), oj)
    assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, _sum(o, pts))
    P_, Q = pts[0], pts[1]
    seq = [P_, P_, P_, P_, None, None, Q, o.aff_neg(o.F1, Q), P_, o.aff_neg(o.F1, P_), Q, None]
    sb = b"".join(o.affine_to_bytes(o.F1, x) for x in seq)
    h28.h28_g1_add_tree(sb, C.c_size_t(len(seq)), oj)
    assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, _sum(o, seq))
    for k in (0, 1, 5, 9):
        h28.h28_g1_dbl_n(o.affine_to_bytes(o.F1, P_), k, oj)
        assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, o.scalar_mul(o.F1, P_, 1 << k))
    h28.h28_g1_dbl_n(bytes(96), 3, oj)
    assert oj.raw == bytes(144)


def test_karatsuba_column_and_bias_bounds():
    """The pairing accumulate kernel collects Fp2 products as Karatsuba column sets (pairing_kernels.cuh: KaraCols, fp2_kara_reduce).
    Restated with big integers: the bias p 2^388 covers every V1 its two users reach, the reduction input stays below the 2^392 p of
    the multiplier's contract with outputs < 2p, and both recombined (signed) column sets stay inside +-2^63 for the worst operands
    the documented value bounds permit — with at most four terms per reduction; six (the Fp12 tree) would not fit, which is why
    k_fp12_prod keeps the four-product form with unsigned columns."""
    from oracle import bls12_381 as o

    p = o.P
    R = 1 << 392
    bias = p << 388
    limb = (1 << 28) + 64          # N-form limb bound (fp28.cuh: SPREAD_LO)
    prod = limb * limb             # one limb product
    # (terms, bound of a0', a1', g0, g1 in units of p): line multiplication (a line's c0 <= 6p), squaring (a <= 4p doubled
    # coefficient, xi a <= (12p, 8p))
    users = {"line": (3, 10, 4, 6, 6), "square": (4, 12, 8, 2, 2)}
    for name, (terms, a0, a1, g0, g1) in users.items():
        v0, v1 = terms * a0 * g0 * p * p, terms * a1 * g1 * p * p
        cross = terms * (a0 * g1 + a1 * g0) * p * p
        assert bias >= v1, name                                   # c0 = V0 - V1 + bias >= 0
        assert v0 + bias < R * p and cross < R * p, name          # the multiplier's contract: input < 2^392 p
        assert (v0 + bias) // R + p < 2 * p and cross // R + p < 2 * p, name   # outputs < 2p, as every consumer assumes
        col = terms * 14 * prod                                   # one column of one column set
        assert col + (1 << 52) + 14 * prod < 1 << 63, name        # |V0 - V1| + bias column + the reduction's 14 m_i p_j
        # |V2 - (V0 + V1)| + reduction terms + carry.  Round 4: g0 + g1 enters V2 WITHOUT a carry pass (limbs <= 2 limb), so a V2
        # column is <= terms * 14 * limb * 2 limb = 2 col, the same bound as the V0 + V1 column; both are non-negative, so the
        # difference is bounded by the larger one
        col2 = terms * 14 * limb * (2 * limb)
        assert col2 == 2 * col
        assert max(col2, 2 * col) + 14 * prod + (1 << 36) < 1 << 63, name
    assert 2 * 6 * 14 * prod > 1 << 63                            # six terms: why the tree does not use this form


def test_small_reduction_and_jacobian_subgroup_ladder(h28, co, o):
    """Round 6: the G1 subgroup test runs on a Jacobian ladder (ec.cuh jac_dbl / jac_add / g1_torsion_free) whose doubling subtracts 8 X B and
    comes back under the multiplier's contract with fp_reduce_small.  Pinned here on the host build of the same headers: the reduction on
    multiples up to 126 (value in [p, 2.01 p), same residue), [z^2] P against the oracle's scalar multiplication on subgroup and off-subgroup
    points, and the verdict on points of SMALL order, where the incomplete addition meets its exceptional cases (P + P, P - P, infinity + P)
    and must still answer "not in the subgroup"."""
    import tests.cofactor_util as cu

    rnd = random.Random(61)
    lim = (C.c_uint32 * 14)()
    lout = (C.c_uint32 * 14)()
    val = lambda l: sum(int(v) << (28 * k) for k, v in enumerate(l))
    for m in list(range(0, 12)) + [31, 32, 33, 63, 64, 65, 100, 125, 126]:
        for a in (0, 1, o.P - 1, rnd.randrange(o.P), rnd.randrange(o.P)):
            h28.h28_fp_reduce_small(o.fp_to_mont_bytes(a), m, lim, lout)
            vi, vo = val(lim), val(lout)
            assert (vi - vo) % o.P == 0 and o.P <= vo < 2 * o.P + o.P // 100, (m, a)
            assert all(int(v) <= (1 << 28) + 15 for v in lout[:13])

    z2 = 0xd201000000010000 ** 2
    bases = co.gen_bases("g1", 61, 6)
    good = [o.affine_from_bytes(o.F1, bases[96 * i:96 * i + 96]) for i in range(6)]
    off = cu.off_subgroup_points(o, "g1", 4)
    oj = C.create_string_buffer(144)
    for pt, want in [(p, 1) for p in good] + [(p, 0) for p in off]:
        b = o.affine_to_bytes(o.F1, pt)
        h28.h28_g1_mul_z2(b, oj)
        assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, o.scalar_mul(o.F1, pt, z2))
        assert h28.h28_g1_torsion_free(b) == want
    # points of small order: n = h1 * r is the group order, h1 = 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2
    n = cu.H1 * o.R_ORDER
    seen = 0
    for q in (3, 11, 10177):
        assert cu.H1 % q == 0
        cof = n
        while cof % q == 0:
            cof //= q
        for base in off:
            t = o.scalar_mul(o.F1, base, cof)               # in the q-primary part
            if t is o.INF:
                continue
            while o.scalar_mul(o.F1, t, q) is not o.INF:
                t = o.scalar_mul(o.F1, t, q)                # order exactly q
            assert h28.h28_g1_torsion_free(o.affine_to_bytes(o.F1, t)) == 0
            h28.h28_g1_mul_z2(o.affine_to_bytes(o.F1, t), oj)
            want = o.scalar_mul(o.F1, t, z2 % q)
            if oj.raw != bytes(144):                        # no exceptional case met on the way: then the value is the right one
                assert co.to_affine("g1", oj.raw) == o.affine_to_bytes(o.F1, want)
            seen += 1
    assert seen >= 6
