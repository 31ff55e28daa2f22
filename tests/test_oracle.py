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
