"""GPU parity tests (-m gpu) for the pairing row: `mi_multi_miller_loop` / `mi_final_exponentiation` /
`mi_multi_pairing` through the C ABI against the textbook oracle (oracle/pairing.py: flat Fp12, affine Miller loop,
one big power — a different algorithm from the shipped projective loop + Frobenius chain).

The reference's own test is bilinearity, e(sP, Q) == e(P, sQ) (/root/reference/src/pairing.rs:92-101); it is
test_bilinearity below.  Bar: bit-exact on the 576-byte blst_fp12 encoding of the Gt value (after the final
exponentiation — Miller values of different implementations differ by factors the exponentiation removes).
"""
import pytest

pytestmark = pytest.mark.gpu

SEED_P, SEED_Q = 0xB11 + 7, 0xB12 + 7


@pytest.fixture(scope="module")
def pr():
    from oracle import pairing

    return pairing


def _pts(o, F, blob, size):
    return [o.affine_from_bytes(F, blob[i:i + size]) for i in range(0, len(blob), size)]


def _neg_g1(o, blob):
    out = b""
    for i in range(0, len(blob), 96):
        pt = o.affine_from_bytes(o.F1, blob[i:i + 96])
        out += o.affine_to_bytes(o.F1, o.aff_neg(o.F1, pt))
    return out


@pytest.mark.parametrize("n", [1, 2, 5])
def test_multi_pairing_vs_oracle(ctx, co, o, pr, n):
    g1 = co.gen_bases("g1", SEED_P, n, 1)
    g2 = co.gen_bases("g2", SEED_Q, n, 1)
    got = ctx.multi_pairing(g1, g2)
    want = pr.final_exponentiation(pr.multi_miller_loop(_pts(o, o.F1, g1, 96), _pts(o, o.F2, g2, 192)))
    assert got == pr.fp12_to_bytes(want)


@pytest.mark.parametrize("n", [63, 700])
def test_multi_pairing_vs_c_oracle(ctx, co, n):
    """hundreds of random pairs, bit-exact against the C restatement of the textbook algorithm (oracle/pairing_oracle.c)"""
    g1 = co.gen_bases("g1", SEED_P + 9, n, 8)
    g2 = co.gen_bases("g2", SEED_Q + 9, n, 8)
    assert ctx.multi_pairing(g1, g2) == co.multi_pairing(g1, g2, 8)


def test_miller_then_final_exp_equals_multi_pairing(ctx, co, pkg):
    n = 7
    g1 = co.gen_bases("g1", SEED_P + 1, n, 1)
    g2 = co.gen_bases("g2", SEED_Q + 1, n, 1)
    ml = ctx.multi_miller_loop(g1, g2)
    assert pkg.final_exponentiation(ml) == ctx.multi_pairing(g1, g2)


def test_bilinearity(ctx, o, pr):
    """the reference's test: e(s G1, G2) == e(G1, s G2); here also == e(G1, G2)^s by the oracle"""
    s = 0x1F3C5A7E9B2D4F60718293A4B5C6D7E8F9012345
    sp = o.scalar_mul(o.F1, o.G1_GEN, s)
    sq = o.scalar_mul(o.F2, o.G2_GEN, s)
    a1, a2 = o.affine_to_bytes(o.F1, sp), o.affine_to_bytes(o.F2, o.G2_GEN)
    b1, b2 = o.affine_to_bytes(o.F1, o.G1_GEN), o.affine_to_bytes(o.F2, sq)
    left, right = ctx.multi_pairing(a1, a2), ctx.multi_pairing(b1, b2)
    assert left == right
    e = pr.fp12_from_bytes(ctx.multi_pairing(b1, a2))
    assert not pr.fp12_eq(e, pr.FP12_ONE)
    assert pr.fp12_eq(pr.fp12_from_bytes(left), pr.fp12_pow(e, s))


def test_infinity_pairs_and_empty(ctx, co, o, pr):
    one = pr.fp12_to_bytes(pr.FP12_ONE)
    assert ctx.multi_pairing(b"", b"") == one
    n = 4
    g1 = bytearray(co.gen_bases("g1", SEED_P + 2, n, 1))
    g2 = bytearray(co.gen_bases("g2", SEED_Q + 2, n, 1))
    full = ctx.multi_pairing(bytes(g1), bytes(g2))
    # pair 1: P at infinity, pair 2: Q at infinity -> both contribute 1 (src/pairing.rs:58-60)
    g1[96:192] = bytes(96)
    g2[2 * 192:3 * 192] = bytes(192)
    got = ctx.multi_pairing(bytes(g1), bytes(g2))
    keep = [0, 3]
    want = pr.final_exponentiation(pr.multi_miller_loop(
        [o.affine_from_bytes(o.F1, bytes(g1[96 * i:96 * i + 96])) for i in keep],
        [o.affine_from_bytes(o.F2, bytes(g2[192 * i:192 * i + 192])) for i in keep]))
    assert got == pr.fp12_to_bytes(want)
    assert got != full
    assert ctx.multi_pairing(bytes(96), bytes(192)) == one


@pytest.mark.parametrize("n", [64, 65, 1000, 4096])
def test_product_cancels_at_size(ctx, co, o, pr, n):
    """size-independent property: prod_i e(P_i, Q_i) * e(-P_i, Q_i) == 1, through every level of the multiplication tree"""
    g1 = co.gen_bases("g1", SEED_P + 3, n, 4)
    g2 = co.gen_bases("g2", SEED_Q + 3, n, 4)
    got = ctx.multi_pairing(g1 + _neg_g1(o, g1), g2 + g2)
    assert got == pr.fp12_to_bytes(pr.FP12_ONE)
    # and the product itself is not trivially 1
    assert ctx.multi_pairing(g1, g2) != got


def test_golden_vectors(ctx):
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairing_vectors.json")))
    for c in gold["multi_pairing"]:
        assert ctx.multi_pairing(bytes.fromhex(c["g1"]), bytes.fromhex(c["g2"])).hex() == c["gt"], c["name"]


def test_pairs_sharded_over_devices(pkg, ctx, co):
    """a context with several devices shards the pairs and multiplies the per-device products (here: device 0 twice)"""
    n = 131
    g1 = co.gen_bases("g1", SEED_P + 5, n, 2)
    g2 = co.gen_bases("g2", SEED_Q + 5, n, 2)
    with pkg.Context([0, 0, 0]) as c3:
        assert c3.multi_pairing(g1, g2) == ctx.multi_pairing(g1, g2)
        assert c3.multi_pairing(g1[:96], g2[:192]) == ctx.multi_pairing(g1[:96], g2[:192])


def test_line_buffer_batches(ctx, tctx, co):
    """more pairs than one line-buffer batch (2^17 in production; forced to 100 through the test build's hook)"""
    g1 = co.gen_bases("g1", 77, 257, 2)
    g2 = co.gen_bases("g2", 78, 257, 2)
    want = ctx.multi_pairing(g1, g2)
    for batch, share in ((100, 0), (100, 7), (64, 3)):   # batches that are not a multiple of the accumulator width either
        tctx.test_set_pairing(batch=batch, share=share)
        try:
            got = tctx.multi_pairing(g1, g2)
        finally:
            tctx.test_set_pairing()
        assert got == want and len(got) == 576, (batch, share)


def test_single_lane_first_version_agrees(ctx, tctx, co):
    """the one-lane-per-pair first version of the Miller loop (test builds only) as a second implementation"""
    g1 = co.gen_bases("g1", 79, 70, 2)
    g2 = co.gen_bases("g2", 80, 70, 2)
    tctx.test_set_pairing(single_lane=True)
    try:
        got = tctx.multi_pairing(g1, g2)
    finally:
        tctx.test_set_pairing()
    assert got == ctx.multi_pairing(g1, g2)


@pytest.mark.parametrize("share", [2, 3, 5, 7, 8])
def test_shared_squaring_accumulators(tctx, co, share):
    """m pairs per accumulator (one Fp12 squaring per Miller step for all of them; production picks the m <= 8 that leaves
    one accumulate wave per SIMD: 7 at 2^16 pairs): forced through the test hook on sizes that do and do not divide by m, with
    infinity pairs, against the C oracle.  The line buffer is laid out in blocks of 10 m pairs: 997 pairs leave a ragged block."""
    for n in (1, share, share + 1, 997):
        g1 = bytearray(co.gen_bases("g1", SEED_P + 11, n, 8))
        g2 = bytearray(co.gen_bases("g2", SEED_Q + 11, n, 8))
        if n > 10:
            g1[96 * 5:96 * 6] = bytes(96)
            g2[192 * (n - 1):192 * n] = bytes(192)
        tctx.test_set_pairing(share=share)
        try:
            got = tctx.multi_pairing(bytes(g1), bytes(g2))
        finally:
            tctx.test_set_pairing()
        assert got == co.multi_pairing(bytes(g1), bytes(g2), 8), (share, n)


def test_pairing_2_16_pairs_cancellation_and_sample(pkg, co, o):
    """BASELINE config #5's size: 2^16 G1 x G2 pairs.  Size-independent property at full size — the second half is the first
    with P negated, so prod e(P_i, Q_i) e(-P_i, Q_i) == 1 — and 1024 of the pairs bit-exact against the C oracle."""
    from oracle import pairing as pr

    half = 1 << 15
    p1 = co.gen_bases("g1", SEED_P + 160, half, 16)
    q2 = co.gen_bases("g2", SEED_Q + 160, half, 16)
    neg = bytearray(p1)
    for i in range(half):
        y = int.from_bytes(p1[96 * i + 48:96 * i + 96], "little")
        neg[96 * i + 48:96 * i + 96] = (o.P - y).to_bytes(48, "little")
    with pkg.Context([0]) as c:
        assert c.multi_pairing(p1 + bytes(neg), q2 + q2) == pr.fp12_to_bytes(pr.FP12_ONE)
        half_gt = c.multi_pairing(p1, q2)
        assert half_gt != pr.fp12_to_bytes(pr.FP12_ONE)
        m = 1024
        assert c.multi_pairing(p1[:96 * m], q2[:192 * m]) == co.multi_pairing(p1[:96 * m], q2[:192 * m], 16)
