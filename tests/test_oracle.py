"""CPU tests (-m "not gpu"): the oracle against the reference's embedded known-answer constants and against the
committed golden vectors; the C restatement against the independent Python big-int restatement."""
import json
import os
import random

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "msm_vectors.json")) as f:
        return json.load(f)


def test_python_oracle_selfcheck(o):
    o.selfcheck()  # reference KATs: src/fp.rs:25-32,714-721; src/scalar.rs:476-481; src/g1.rs:42-51; src/g2.rs:45-63


def test_c_oracle_selfcheck(co):
    assert co.selfcheck() == 0


def test_c_oracle_fp_and_fr_golden(co, golden):
    a = bytes.fromhex("".join(v["a"] for v in golden["fp_mul"]))
    b = bytes.fromhex("".join(v["b"] for v in golden["fp_mul"]))
    assert co.fp_mul(a, b) == bytes.fromhex("".join(v["ab"] for v in golden["fp_mul"]))
    canon = bytes.fromhex("".join(v["canon"] for v in golden["fr_mont"]))
    mont = bytes.fromhex("".join(v["mont"] for v in golden["fr_mont"]))
    assert co.fr_from_mont(mont) == canon       # Scalar::into_bigint, src/scalar.rs:503-505
    assert co.fr_to_mont(canon) == mont


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_c_oracle_msm_golden(co, golden, group):
    for case in golden[group]:
        n = case["n"]
        bases, sc, scm = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"]), bytes.fromhex(case["scalars_mont"])
        want = bytes.fromhex(case["expected_affine"])
        assert co.to_affine(group, co.msm(group, bases, sc, n, 0, 1)) == want, case["name"]
        assert co.to_affine(group, co.msm(group, bases, scm, n, 1, 3)) == want, case["name"]
        assert co.to_affine(group, co.msm_naive(group, bases, sc, n)) == want, case["name"]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_c_oracle_generators_match_python(co, o, group):
    F, gen = (o.F1, o.G1_GEN) if group == "g1" else (o.F2, o.G2_GEN)
    aff = 96 if group == "g1" else 192
    n = 6
    bases = co.gen_bases(group, 0xA55E7, n, 2)
    for i in range(n):
        k = o.gen_dlog(0xA55E7, i)
        assert bases[aff * i:aff * (i + 1)] == o.affine_to_bytes(F, o.scalar_mul(F, gen, k))
    assert co.gen_scalars(0x5CA1A5, 50) == b"".join(o.fr_to_canon_bytes(s) for s in o.rand_scalars(50, 0x5CA1A5))


@pytest.mark.parametrize("group,n", [("g1", 1000), ("g1", 1 << 14), ("g2", 300)])
def test_c_oracle_pippenger_vs_closed_form(co, group, n):
    """Closed form (sum s_i k_i) G for known-discrete-log bases: independent of the Pippenger code."""
    bases = co.gen_bases(group, 77, n, 8)
    sc = co.gen_scalars(78, n)
    assert co.to_affine(group, co.msm(group, bases, sc, n, 0, 8)) == co.dlog_expected(group, sc, 77, n)


def test_c_oracle_fold_and_sum(co, o):
    rnd = random.Random(3)
    ks = [rnd.randrange(1, 1 << 40) for _ in range(5)]
    wins = b"".join(o.jac_to_bytes(o.F1, o.scalar_mul(o.F1, o.G1_GEN, k)) for k in ks)
    c = 13
    want = o.scalar_mul(o.F1, o.G1_GEN, sum(k << (c * i) for i, k in enumerate(ks)) % o.R_ORDER)
    assert co.to_affine("g1", co.fold_windows("g1", wins, 5, c)) == o.affine_to_bytes(o.F1, want)
    want = o.scalar_mul(o.F1, o.G1_GEN, sum(ks) % o.R_ORDER)
    assert co.to_affine("g1", co.sum_jac("g1", wins, 5)) == o.affine_to_bytes(o.F1, want)


def test_g1_encoding_oracle_vs_fixtures(o, golden):
    """ZCash/IETF G1 encoding restatement (src/g1.rs:358-431) against the frozen fixtures and the published
    compressed generator (97f1d3a7...c6bb)."""
    assert o.g1_compress(o.G1_GEN).hex().startswith("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905")
    for case in golden["g1_encoding"]:
        pt, st = o.g1_deserialize(bytes.fromhex(case["bytes"]), case["compressed"], case["validate"])
        assert st == case["status"], case["name"]
        if st == 0:
            assert o.affine_to_bytes(o.F1, pt).hex() == case["affine"]
            enc = o.g1_compress(pt) if case["compressed"] else o.g1_uncompressed(pt)
            if case["name"].startswith(("valid", "infinity", "generator")):
                assert enc == bytes.fromhex(case["bytes"])
    # the endomorphism constant used by the GPU subgroup test: (beta x, y) = -[z^2] P on G1
    beta = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE
    z = 0xD201000000010000
    q = o.scalar_mul(o.F1, o.G1_GEN, (z * z) % o.R_ORDER)
    assert (beta * o.G1_X % o.P, o.G1_Y) == o.aff_neg(o.F1, q)


PUBLIC_PK_SK1 = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
PUBLIC_PK_SK2 = "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e"


def test_public_known_answers_for_the_group_law(o, co):
    """An anchor from OUTSIDE this repository and outside /root/reference (which holds no MSM vector): the BLS12-381 public keys of the
    secret keys 1 and 2 in the ZCash / IETF compressed encoding — the generator and [2]G, constants that circulate in the BLS signature test
    vectors of every implementation (pk = [sk]G on G1).  They pin doubling, the encoding and — through sums that must land on [2]G — the MSM
    oracle to a published value, not only to itself."""
    assert o.g1_compress(o.G1_GEN).hex() == PUBLIC_PK_SK1
    two_g = o.aff_add(o.F1, o.G1_GEN, o.G1_GEN)
    assert o.g1_compress(two_g).hex() == PUBLIC_PK_SK2
    assert o.g1_compress(o.scalar_mul(o.F1, o.G1_GEN, 2)).hex() == PUBLIC_PK_SK2
    # the C oracle's Pippenger: 300 copies of G with scalars that sum to 2 modulo r
    import random

    rnd = random.Random(2)
    n = 300
    ks = [rnd.randrange(o.R_ORDER) for _ in range(n - 1)]
    ks.append((2 - sum(ks)) % o.R_ORDER)
    bases = o.affine_to_bytes(o.F1, o.G1_GEN) * n
    sc = b"".join(k.to_bytes(32, "little") for k in ks)
    got = co.to_affine("g1", co.msm("g1", bases, sc, n, 0, 4))
    assert o.g1_compress(o.affine_from_bytes(o.F1, got)).hex() == PUBLIC_PK_SK2


def test_g2_encoding_oracle_vs_fixtures(o, golden):
    assert o.g2_compress(o.G2_GEN).hex().startswith("93e02b6052719f607dacd3a088274f65596bd0d09920b61a")
    for case in golden["g2_encoding"]:
        pt, st = o.g2_deserialize(bytes.fromhex(case["bytes"]), case["compressed"], case["validate"])
        assert st == case["status"], case["name"]
        if st == 0:
            assert o.affine_to_bytes(o.F2, pt).hex() == case["affine"]
    # psi(P) = [z] P on G2 with the constants baked into the GPU subgroup test
    cx = (0, 0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAD)
    cy = (0x135203E60180A68EE2E9C448D77A2CD91C3DEDD930B1CF60EF396489F61EB45E304466CF3E67FA0AF1EE7B04121BDEA2,
          0x06AF0E0437FF400B6831E36D6BD17FFE48395DABC2D3435E77F76E17009241C5EE67992F72EC05F4C81084FBEDE3CC09)
    x, y = o.G2_GEN
    psi = (o.F2.mul((x[0], (-x[1]) % o.P), cx), o.F2.mul((y[0], (-y[1]) % o.P), cy))
    assert psi == o.scalar_mul(o.F2, o.G2_GEN, (-0xD201000000010000) % o.R_ORDER)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_cofactor_clearing_pins_the_group_law(co, o, group):
    """Points OUTSIDE the prime-order subgroup times the reference-held cofactor (src/g1.rs:42, src/g2.rs:45-54) land INSIDE it, in
    both oracles: the group law and the scalar path are tied to a constant the reference holds (the generators alone would not
    notice a formula that is only right on the r-torsion)."""
    from cofactor_util import H1, H2, off_subgroup_points, h2_pieces, SPLIT_BITS

    F = o.F1 if group == "g1" else o.F2
    h = H1 if group == "g1" else H2
    assert h == (o.H1 if group == "g1" else h)
    for pt in off_subgroup_points(o, group, 2):
        q = o.scalar_mul(F, pt, h)
        assert q is not o.INF and o.on_curve(F, q)
        assert o.scalar_mul(F, q, o.R_ORDER) is o.INF                          # h * P is in the r-torsion ...
        assert o.scalar_mul(F, o.scalar_mul(F, pt, h - 1), o.R_ORDER) is not o.INF   # ... and it takes exactly h
        want = o.affine_to_bytes(F, q)
        base = o.affine_to_bytes(F, pt)
        if group == "g1":
            sc = h.to_bytes(32, "little")
            assert co.to_affine(group, co.msm(group, base, sc, 1, 0, 1)) == want       # C oracle, Pippenger path
            assert co.to_affine(group, co.msm_naive(group, base, sc, 1)) == want       # C oracle, double-and-add
        else:
            # h2 has 507 bits: three pieces below r over the bases P, 2^250 P, 2^500 P
            bases = b"".join(o.affine_to_bytes(F, o.scalar_mul(F, pt, 1 << (SPLIT_BITS * i))) for i in range(3))
            sc = b"".join(a.to_bytes(32, "little") for a in h2_pieces())
            assert co.to_affine(group, co.msm(group, bases, sc, 3, 0, 1)) == want
            assert co.to_affine(group, co.msm_naive(group, bases, sc, 3)) == want
        # r * (h P) = infinity in the C oracle as well: r = (r - 1) + 1 over the base h P twice (scalars stay below r)
        two = want + want
        sc = (o.R_ORDER - 1).to_bytes(32, "little") + (1).to_bytes(32, "little")
        assert co.to_affine(group, co.msm(group, two, sc, 2, 0, 1)) == bytes(len(want))


def test_c_oracle_normalize_batch(co, o):
    """orc_g{1,2}_normalize_batch (the restatement behind bench.py's normalize cpu_baseline; CurveGroup::normalize_batch,
    src/g1.rs:537-543) against the per-point conversion and the Python oracle, infinity included, for several thread splits."""
    for group, F, gen, jsz in (("g1", o.F1, o.G1_GEN, 144), ("g2", o.F2, o.G2_GEN, 288)):
        pts = [o.scalar_mul(F, gen, k) for k in (1, 2, 3, 12345, o.R_ORDER - 1)]
        jac = []
        for i, p in enumerate(pts):   # non-trivial Z: the projective sum of two multiples, as a Jacobian triple from the oracle
            X, Y, Z = o.jac_add(F, o.jac_from_aff(F, p), o.jac_double(F, o.jac_from_aff(F, pts[(i + 1) % len(pts)])))
            assert not F.eq(Z, F.one)
            jac.append(o._felt_bytes(F, X) + o._felt_bytes(F, Y) + o._felt_bytes(F, Z))
        jac.insert(2, bytes(jsz))                                    # infinity (Z = 0)
        blob = b"".join(jac)
        want = b"".join(co.to_affine(group, j) for j in jac)
        for threads in (1, 2, 7):
            assert co.normalize_batch(group, blob, threads) == want
        assert want[2 * (jsz * 2 // 3):3 * (jsz * 2 // 3)] == bytes(jsz * 2 // 3)
        k0 = (1 + 2 * 2) % o.R_ORDER
        assert want[:jsz * 2 // 3] == o.affine_to_bytes(F, o.scalar_mul(F, gen, k0))
    assert co.normalize_batch("g1", b"", 4) == b""


def test_c_oracle_g1_deserialize(co, o, golden):
    """orc_g1_deserialize_batch (bench.py's deserialize cpu_baseline; src/g1.rs:386-431) against the frozen encoding fixtures and
    the Python oracle: both subgroup tests (the definition [r] P and the endomorphism form) agree on points inside AND outside the
    subgroup, malformed encodings and off-curve points get the reference's statuses."""
    from cofactor_util import off_subgroup_points

    for case in golden["g1_encoding"]:
        for mode in (0, 1):
            out, st = co.g1_deserialize_batch(bytes.fromhex(case["bytes"]), case["compressed"], case["validate"], mode, 1)
            assert st[0] == case["status"], (case["name"], mode)
            assert out.hex() == (case["affine"] if st[0] == 0 and case["affine"] else bytes(96).hex()), case["name"]
    # points on the curve but outside the prime-order subgroup: status 3 with validation, accepted without
    off = off_subgroup_points(o, "g1", 3)
    enc = b"".join(o.g1_compress(p) for p in off) + o.g1_compress(o.G1_GEN)
    for mode in (0, 1):
        out, st = co.g1_deserialize_batch(enc, True, True, mode, 2)
        assert st == bytes([3, 3, 3, 0]) and out[:288] == bytes(288) and out[288:] == o.affine_to_bytes(o.F1, o.G1_GEN)
    out, st = co.g1_deserialize_batch(enc, True, False, 0, 1)
    assert st == bytes(4) and out == b"".join(o.affine_to_bytes(o.F1, p) for p in off + [o.G1_GEN])
    # a batch against the Python oracle, compressed and uncompressed
    pts = [o.scalar_mul(o.F1, o.G1_GEN, k) for k in range(1, 30)]
    for compressed in (True, False):
        blob = b"".join((o.g1_compress(p) if compressed else o.g1_uncompressed(p)) for p in pts)
        out, st = co.g1_deserialize_batch(blob, compressed, True, 1, 3)
        assert st == bytes(len(pts)) and out == b"".join(o.affine_to_bytes(o.F1, p) for p in pts)


def test_c_oracle_g2_deserialize(co, o, golden):
    """orc_g2_deserialize_batch (bench.py's G2 deserialize cpu_baseline; src/g2.rs:366-411) against the frozen G2 encoding fixtures and
    the Python oracle: the definition [r] Q and the psi(Q) == [z] Q form of the subgroup test agree on points inside AND outside the
    subgroup; malformed encodings (no square root in Fp2, unreduced coordinate, flipped flags) and off-curve points get the
    reference's statuses."""
    from cofactor_util import off_subgroup_points

    for case in golden["g2_encoding"]:
        for mode in (0, 1):
            out, st = co.g2_deserialize_batch(bytes.fromhex(case["bytes"]), case["compressed"], case["validate"], mode, 1)
            assert st[0] == case["status"], (case["name"], mode)
            assert out.hex() == (case["affine"] if st[0] == 0 and case["affine"] else bytes(192).hex()), case["name"]
    off = off_subgroup_points(o, "g2", 2)
    enc = b"".join(o.g2_compress(p) for p in off) + o.g2_compress(o.G2_GEN)
    for mode in (0, 1):
        out, st = co.g2_deserialize_batch(enc, True, True, mode, 2)
        assert st == bytes([3, 3, 0]) and out[:384] == bytes(384) and out[384:] == o.affine_to_bytes(o.F2, o.G2_GEN)
    out, st = co.g2_deserialize_batch(enc, True, False, 0, 1)
    assert st == bytes(3) and out == b"".join(o.affine_to_bytes(o.F2, p) for p in off + [o.G2_GEN])
    bases = co.gen_bases("g2", 4711, 12, 2)
    pts = [o.affine_from_bytes(o.F2, bases[192 * i:192 * (i + 1)]) for i in range(12)]
    for compressed in (True, False):
        blob = b"".join((o.g2_compress(p) if compressed else o.g2_uncompressed(p)) for p in pts)
        out, st = co.g2_deserialize_batch(blob, compressed, True, 1, 3)
        assert st == bytes(len(pts)) and out == bases
