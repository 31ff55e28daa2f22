"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Mirrors the reference's own MSM test — msm(normalize_batch(bases), scalars) == sum b_i * s_i
(/root/reference/src/tests.rs:50-67, run for G1 at src/g1.rs:677-680) — with fixed seeds, plus the edge cases the
reference documents (infinity among the bases, src/g1.rs:682-709) and the ones incomplete formulas would miss.
Bar: bit-exact on the canonical affine encoding (fully reduced Montgomery limbs of x, y).
"""
import random

import pytest

pytestmark = pytest.mark.gpu

SEED_B, SEED_S = 0xA55E7 + 2, 0x5CA1A5 + 2


def _canon(co, group, jac):
    return co.to_affine(group, jac)


def test_fp_ops_match_oracle(tctx, co, o):
    rnd = random.Random(7)
    n = 4096
    va = [rnd.randrange(o.P) for _ in range(n)]
    vb = [rnd.randrange(o.P) for _ in range(n)]
    edge = [0, 1, o.P - 1, (o.P - 1) // 2, 2, o.P - 2]
    va[:6] = edge
    vb[:6] = edge[::-1]
    a = b"".join(o.fp_to_mont_bytes(v) for v in va)
    b = b"".join(o.fp_to_mont_bytes(v) for v in vb)
    assert tctx.test_fp_op(0, a, b) == co.fp_mul(a, b)
    assert tctx.test_fp_op(1, a, b) == co.fp_mul(a, a)
    assert tctx.test_fp_op(2, a, b) == b"".join(o.fp_to_mont_bytes((x + y) % o.P) for x, y in zip(va, vb))
    assert tctx.test_fp_op(3, a, b) == b"".join(o.fp_to_mont_bytes((x - y) % o.P) for x, y in zip(va, vb))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 10, 500, 1024, 5000])
def test_g1_msm_small_vs_oracle(ctx, co, pkg, n):
    bases = co.gen_bases("g1", SEED_B, n, 4)
    scalars = co.gen_scalars(SEED_S, n)
    got = ctx.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
    want = co.msm_naive("g1", bases, scalars, n) if n <= 10 else co.msm("g1", bases, scalars, n, 0, 4)
    assert _canon(co, "g1", got) == _canon(co, "g1", want)
    if n:
        assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B, n)


@pytest.mark.parametrize("n", [3, 4096, 150000])
def test_g1_msm_lands_on_a_published_point(ctx, co, o, pkg, n):
    """An anchor from outside this repository and outside the reference (which holds no MSM vector): an MSM whose scalars are chosen so that
    sum(s_i k_i) = 2 modulo r over the suite's bases [k_i]G must give [2]G, whose compressed encoding a572cbea...0f4e is the published BLS12-381
    public key of the secret key 2 (the generator's, 97f1d3a7...c6bb, that of the secret key 1).  All scalars but the last are random: every
    window, sign fold and bucket of the HIP path takes part, and the result is compared with a constant, not with another implementation."""
    import ctypes as C
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle import PUBLIC_PK_SK1, PUBLIC_PK_SK2

    bases = co.gen_bases("g1", SEED_B, n, 4)                   # [k_i]G with the generator's k_i (oracle/msm_oracle.c orc_g1_gen_bases)
    sc = bytearray(co.gen_scalars(SEED_S, n))
    sc[32 * (n - 1):] = bytes(32)                              # last scalar 0: acc = sum over the others

    def dot(v):
        out = C.create_string_buffer(32)
        co.lib().orc_dot_mod_r(bytes(v), SEED_B, n, out)
        return int.from_bytes(out.raw, "little")

    acc = dot(sc)
    unit = bytearray(32 * n)
    unit[32 * (n - 1)] = 1
    k_last = dot(unit)
    sc[32 * (n - 1):] = ((2 - acc) * pow(k_last, -1, o.R_ORDER) % o.R_ORDER).to_bytes(32, "little")
    assert dot(sc) == 2
    got = _canon(co, "g1", ctx.msm("g1", bases, bytes(sc), n, pkg.SCALAR_CANONICAL))
    assert o.g1_compress(o.affine_from_bytes(o.F1, got)).hex() == PUBLIC_PK_SK2
    assert o.g1_compress(o.G1_GEN).hex() == PUBLIC_PK_SK1


def test_g1_msm_montgomery_scalars(ctx, co, pkg):
    n = 777
    bases = co.gen_bases("g1", SEED_B, n, 4)
    canon = co.gen_scalars(SEED_S, n)
    mont = co.gen_scalars(SEED_S, n, True)
    a = ctx.msm("g1", bases, canon, n, pkg.SCALAR_CANONICAL)
    b = ctx.msm("g1", bases, mont, n, pkg.SCALAR_MONTGOMERY)
    assert _canon(co, "g1", a) == _canon(co, "g1", b) == co.dlog_expected("g1", canon, SEED_B, n)


@pytest.mark.parametrize("c", list(range(7, 17)))
def test_g1_msm_all_window_sizes(ctx, co, pkg, c):
    """every window size of the two-level sort: each has its own reduce geometry (buckets per lane L = 1, 2, 3, 5, 9, 16, ...: the reduce
    wave takes any L and ragged last chunks since round 4)"""
    n = 3000
    bases = co.gen_bases("g1", SEED_B + 1, n, 4)
    scalars = co.gen_scalars(SEED_S + 1, n)
    ctx.set_window_bits(c)
    try:
        got = ctx.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
        assert ctx.profile()["window_bits"] == c
    finally:
        ctx.set_window_bits(0)
    assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 1, n)


@pytest.mark.parametrize("c", [8, 11, 14, 15])
def test_g2_msm_window_sizes(ctx, co, pkg, c):
    """the same for G2 (32 logical lanes per reduce wave: other L, other ragged chunks)"""
    n = 700
    bases = co.gen_bases("g2", SEED_B + 2, n, 4)
    scalars = co.gen_scalars(SEED_S + 2, n)
    ctx.set_window_bits(c)
    try:
        got = ctx.msm("g2", bases, scalars, n, pkg.SCALAR_CANONICAL)
        assert ctx.profile()["window_bits"] == c
    finally:
        ctx.set_window_bits(0)
    assert _canon(co, "g2", got) == co.dlog_expected("g2", scalars, SEED_B + 2, n)


def test_g1_msm_edge_cases(ctx, co, o, pkg):
    """Infinity bases (src/g1.rs:682-709), duplicate bases in one bucket (forces doubling), P and -P with equal
    digits (cancellation), scalars 0 / 1 / r-1, all-equal scalars."""
    n = 64
    raw = co.gen_bases("g1", SEED_B + 3, n, 1)
    pts = [raw[96 * i:96 * i + 96] for i in range(n)]
    neg0 = o.affine_to_bytes(o.F1, o.aff_neg(o.F1, o.affine_from_bytes(o.F1, pts[0])))
    bases = [pts[0], pts[0], pts[0], neg0, pts[1], bytes(96), pts[2], bytes(96), pts[3], pts[3], neg0, pts[4]]
    rnd = random.Random(11)
    s = rnd.randrange(o.R_ORDER)
    scal = [s, s, s, s, 0, 12345, 1, s, o.R_ORDER - 1, o.R_ORDER - 1, 5, s]
    bases += pts[5:40]
    scal += [s] * 35  # all-equal scalars: every point lands in the same bucket of every window
    n2 = len(bases)
    bb, ss = b"".join(bases), b"".join(o.fr_to_canon_bytes(x) for x in scal)
    got = ctx.msm("g1", bb, ss, n2, pkg.SCALAR_CANONICAL)
    want = co.msm_naive("g1", bb, ss, n2)
    assert _canon(co, "g1", got) == _canon(co, "g1", want)
    # everything cancels -> infinity
    bb2, ss2 = pts[0] + neg0, o.fr_to_canon_bytes(s) * 2
    got = ctx.msm("g1", bb2, ss2, 2, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", got) == bytes(96)


@pytest.mark.parametrize("kind", ["all_equal", "zero_one", "small64", "two_values"])
def test_g1_msm_skewed_scalars(ctx, co, o, pkg, kind):
    """Non-uniform scalar distributions (R1CS-like witnesses): heavy buckets are split into work items and merged."""
    n = 6000
    rnd = random.Random(5)
    bases = co.gen_bases("g1", SEED_B + 4, n, 4)
    if kind == "all_equal":
        sc = [rnd.randrange(o.R_ORDER)] * n
    elif kind == "zero_one":
        sc = [rnd.randrange(2) for _ in range(n)]
    elif kind == "small64":
        sc = [rnd.randrange(1 << 64) for _ in range(n)]
    else:
        a, b = rnd.randrange(o.R_ORDER), o.R_ORDER - 1
        sc = [a if rnd.randrange(2) else b for _ in range(n)]
    ss = b"".join(o.fr_to_canon_bytes(x) for x in sc)
    got = ctx.msm("g1", bases, ss, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", got) == co.dlog_expected("g1", ss, SEED_B + 4, n)
    prof = ctx.profile()
    if kind != "small64":
        assert prof["max_items_per_bucket"] > 1  # the split/merge path really ran


@pytest.mark.parametrize("group,n,c,vals", [("g1", 8192, 10, 240), ("g1", 40000, 12, 500), ("g2", 4096, 9, 100)])
def test_many_slightly_overfull_buckets(ctx, co, pkg, group, n, c, vals):
    """Every window's digit drawn from `vals` of the 2^(c-1) values: the used buckets hold about twice the mean the plan expects —
    just over its item size T — so that HUNDREDS of buckets are each cut into three or four short items.  That is the dense-list
    path of the merge tree with many small split buckets (the skewed tests above have a few giant ones) and the case that decides
    how long the schedule's merge list can get."""
    rnd = random.Random(77)
    W = -(-255 // c)
    top_bits = 254 - c * (W - 1)
    sc = []
    for _ in range(n):
        s = 0
        for w in range(W - 1):
            s |= rnd.randrange(1, vals + 1) << (c * w)       # raw window value <= 2^(c-1): no negative digit, no carry
        s |= rnd.randrange(1, min(vals, (1 << max(top_bits - 1, 1)) - 1) + 1) << (c * (W - 1))
        sc.append(s)
    ss = b"".join(x.to_bytes(32, "little") for x in sc)
    bases = co.gen_bases(group, SEED_B + 9, n, 4)
    ctx.set_window_bits(c)
    try:
        got = ctx.msm(group, bases, ss, n, pkg.SCALAR_CANONICAL)
        prof = ctx.profile()
    finally:
        ctx.set_window_bits(0)
    assert _canon(co, group, got) == co.dlog_expected(group, ss, SEED_B + 9, n)
    assert prof["window_bits"] == c and prof["max_items_per_bucket"] >= 3   # buckets were split, into few items each
    assert prof["work_items"] > 2 * vals * (W - 1)                          # ... and many of them were


def test_g1_resident_bases_and_prefix(ctx, co, pkg):
    n = 4096
    bases = co.gen_bases("g1", SEED_B, n, 4)
    scalars = co.gen_scalars(SEED_S, n)
    ctx.set_bases("g1", bases, n)
    for m in (n, 1000):
        got = ctx.msm("g1", None, scalars, m, pkg.SCALAR_CANONICAL)
        assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B, m)
    prof = ctx.profile()
    assert prof["accumulate_ms"] > 0 and prof["n"] == 1000


def test_g1_msm_2_16_and_2_20_closed_form(ctx, co, pkg):
    """BASELINE configs at full size via the size-independent closed form (sum s_i k_i) * G and linearity."""
    for logn in (16, 20):
        n = 1 << logn
        bases = co.gen_bases("g1", SEED_B, n, 16)
        scalars = co.gen_scalars(SEED_S, n)
        got = ctx.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
        assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B, n)
        if logn == 16:
            want = co.msm("g1", bases, scalars, n, 0, 16)
            assert _canon(co, "g1", got) == _canon(co, "g1", want)
        # linearity: msm(first half) + msm(second half) == msm(all)
        h = n // 2
        p1 = ctx.msm("g1", bases[:96 * h], scalars[:32 * h], h, pkg.SCALAR_CANONICAL)
        p2 = ctx.msm("g1", bases[96 * h:], scalars[32 * h:], h, pkg.SCALAR_CANONICAL)
        assert _canon(co, "g1", pkg.g1_sum([p1, p2])) == _canon(co, "g1", got)
        print(f"n=2^{logn}", ctx.profile())


# ------------------------------------------------------------------------------------------------ golden fixtures, G2
import json
import os


def _golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_golden_vectors(ctx, co, pkg, group):
    """Committed fixtures (tests/golden/msm_vectors.json): random sizes + every edge case, both scalar formats."""
    for case in _golden()[group]:
        n = case["n"]
        bases, sc, scm = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"]), bytes.fromhex(case["scalars_mont"])
        want = bytes.fromhex(case["expected_affine"])
        assert _canon(co, group, ctx.msm(group, bases, sc, n, pkg.SCALAR_CANONICAL)) == want, case["name"]
        assert _canon(co, group, ctx.msm(group, bases, scm, n, pkg.SCALAR_MONTGOMERY)) == want, case["name"]


@pytest.mark.parametrize("n", [1, 7, 300, 4096])
def test_g2_msm_vs_oracle(ctx, co, pkg, n):
    """G2 (Fp2) MSM — /root/reference/src/g2.rs:582-612 — against the oracle and the closed form."""
    bases = co.gen_bases("g2", SEED_B + 7, n, 8)
    scalars = co.gen_scalars(SEED_S + 7, n)
    got = ctx.msm("g2", bases, scalars, n, pkg.SCALAR_CANONICAL)
    want = co.msm("g2", bases, scalars, n, 0, 8)
    assert _canon(co, "g2", got) == _canon(co, "g2", want) == co.dlog_expected("g2", scalars, SEED_B + 7, n)


def test_g2_msm_skew_and_resident(ctx, co, o, pkg):
    n = 3000
    rnd = random.Random(9)
    bases = co.gen_bases("g2", SEED_B + 8, n, 8)
    s = rnd.randrange(o.R_ORDER)
    ss = b"".join(o.fr_to_canon_bytes(s if rnd.randrange(3) else rnd.randrange(2)) for _ in range(n))
    ctx.set_bases("g2", bases, n)
    got = ctx.msm("g2", None, ss, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g2", got) == co.dlog_expected("g2", ss, SEED_B + 8, n)
    assert ctx.profile()["max_items_per_bucket"] > 1
    h = n // 2
    p1 = ctx.msm("g2", bases[:192 * h], ss[:32 * h], h, pkg.SCALAR_CANONICAL)
    p2 = ctx.msm("g2", bases[192 * h:], ss[32 * h:], n - h, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g2", pkg.g2_sum([p1, p2])) == _canon(co, "g2", got)


def test_g2_msm_2_16_closed_form(ctx, co, pkg):
    n = 1 << 16
    bases = co.gen_bases("g2", SEED_B + 9, n, 16)
    scalars = co.gen_scalars(SEED_S + 9, n)
    got = ctx.msm("g2", bases, scalars, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g2", got) == co.dlog_expected("g2", scalars, SEED_B + 9, n)
    print("g2 n=2^16", ctx.profile())


def test_in_library_multi_device_sharding(pkg, co):
    """mi_msm_init with several device ids: contiguous shards per device, one host thread each, partials folded in
    device order.  A one-GPU box exercises the same code with device 0 listed twice (two streams, two shards)."""
    n = 5001
    bases = co.gen_bases("g1", SEED_B + 11, n, 8)
    scalars = co.gen_scalars(SEED_S + 11, n)
    with pkg.Context([0, 0, 0]) as c3:
        assert c3.num_devices() == 3
        got = c3.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
        assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 11, n)
        c3.set_bases("g1", bases, n)
        for m in (n, 1700, 3):   # prefixes that end inside the first / second / third shard
            got = c3.msm("g1", None, scalars, m, pkg.SCALAR_CANONICAL)
            assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 11, m)
        b2 = co.gen_bases("g2", SEED_B + 12, 700, 8)
        got = c3.msm("g2", b2, scalars, 700, pkg.SCALAR_CANONICAL)
        assert _canon(co, "g2", got) == co.dlog_expected("g2", scalars, SEED_B + 12, 700)


def test_error_paths(ctx, pkg):
    with pytest.raises(pkg.MsmError) as e:
        ctx2 = pkg.Context([0])
        try:
            ctx2.msm("g1", None, bytes(32), 1, pkg.SCALAR_CANONICAL)   # no resident bases
        finally:
            ctx2.close()
    assert e.value.code == -5
    with pytest.raises(pkg.MsmError):
        ctx.msm("g1", bytes(96), bytes(32), 1, 7)                      # unknown scalar format
    with pytest.raises(pkg.MsmError):
        ctx.set_window_bits(3)


@pytest.mark.parametrize("group,n", [("g1", 1), ("g1", 33), ("g1", 64), ("g1", 65), ("g1", 5000), ("g2", 7), ("g2", 2100)])
def test_normalize_batch_vs_oracle(ctx, co, o, group, n):
    """CurveGroup::normalize_batch (/root/reference/src/g1.rs:537-543, src/g2.rs:517-523): Jacobian -> affine with one
    inversion; every element against the oracle's per-point conversion, infinity (Z = 0) -> all-zero affine."""
    rnd = random.Random(1234 + n)
    F = o.F1 if group == "g1" else o.F2
    aff = 96 if group == "g1" else 192
    raw = co.gen_bases(group, SEED_B + 20, n, 8)
    jac, want = [], []
    for i in range(n):
        pt = o.affine_from_bytes(F, raw[aff * i:aff * (i + 1)])
        if i % 11 == 3:  # infinity with garbage X, Y
            garbage = F.mul(pt[0], pt[1])
            jac.append(o._felt_bytes(F, garbage) + o._felt_bytes(F, pt[1]) + o._felt_bytes(F, F.zero))
            want.append(bytes(aff))
            continue
        lam = rnd.randrange(1, o.P) if group == "g1" else (rnd.randrange(1, o.P), rnd.randrange(o.P))
        l2 = F.mul(lam, lam)
        jac.append(o._felt_bytes(F, F.mul(pt[0], l2)) + o._felt_bytes(F, F.mul(pt[1], F.mul(l2, lam))) + o._felt_bytes(F, lam))
        want.append(raw[aff * i:aff * (i + 1)])
    blob = b"".join(jac)
    got = ctx.normalize_batch(group, blob)
    assert got == b"".join(want)
    # and the C oracle agrees element by element on a sample
    jb = 144 if group == "g1" else 288
    for i in range(0, n, max(1, n // 20)):
        assert co.to_affine(group, blob[jb * i:jb * (i + 1)]) == got[aff * i:aff * (i + 1)]


def test_normalize_batch_feeds_msm(ctx, co, pkg):
    """The prover sequence the reference's own test uses: msm(normalize_batch(bases), scalars) (src/tests.rs:63-66)."""
    n = 512
    bases = co.gen_bases("g1", SEED_B + 21, n, 8)
    scalars = co.gen_scalars(SEED_S + 21, n)
    # partial MSM results are genuine Jacobian points with non-trivial Z
    parts = [ctx.msm("g1", bases[96 * k:96 * (k + 64)], scalars[32 * k:32 * (k + 64)], 64, pkg.SCALAR_CANONICAL) for k in range(0, n, 64)]
    affs = ctx.normalize_batch("g1", b"".join(parts))
    for k, p in enumerate(parts):
        assert affs[96 * k:96 * (k + 1)] == co.to_affine("g1", p)
    ones = b"".join((1).to_bytes(32, "little") for _ in parts)
    total = ctx.msm("g1", affs, ones, len(parts), pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", total) == co.dlog_expected("g1", scalars, SEED_B + 21, n)


def test_cpp_host_mirror_group_test(tmp_path, co, o):
    """The compiled-language host side (ark-blst_amd/host/ark_blst_amd.hpp) running the reference's own MSM test shape
    (src/tests.rs:50-67): msm(normalize_batch(bases), scalars) == sum b_i * s_i, for G1 and G2, plus the
    Err(min(len)) convention and iter::Sum.  Built with g++ against the C-ABI library and run as a child process."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = os.path.join(here, "host", "group_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(here, "host", "group_test.cpp"),
                           "-L" + os.path.join(root, "ark-blst_amd", "lib"), "-larkblst_amd",
                           "-Wl,-rpath," + os.path.join(root, "ark-blst_amd", "lib")])
    rnd = random.Random(77)
    for group, F, n in (("g1", o.F1, 300), ("g2", o.F2, 120)):
        aff = 96 if group == "g1" else 192
        raw = co.gen_bases(group, SEED_B + 30, n, 8)
        canon = co.gen_scalars(SEED_S + 30, n)
        jac = []
        for i in range(n):
            pt = o.affine_from_bytes(F, raw[aff * i:aff * (i + 1)])
            lam = rnd.randrange(1, o.P) if group == "g1" else (rnd.randrange(1, o.P), rnd.randrange(o.P))
            l2 = F.mul(lam, lam)
            jac.append(o._felt_bytes(F, F.mul(pt[0], l2)) + o._felt_bytes(F, F.mul(pt[1], F.mul(l2, lam))) + o._felt_bytes(F, lam))
        (tmp_path / f"{group}_bases_jac.bin").write_bytes(b"".join(jac))
        (tmp_path / f"{group}_scalars_canon.bin").write_bytes(canon)
        (tmp_path / f"{group}_scalars_mont.bin").write_bytes(co.fr_to_mont(canon))
        (tmp_path / f"{group}_expected_affine.bin").write_bytes(co.dlog_expected(group, canon, SEED_B + 30, n))
    s = 0x6A09E667F3BCC908B2FB1366EA957D3E3ADEC17512775099DA2F590B0667322A % o.R_ORDER
    (tmp_path / "pair_P.bin").write_bytes(o.affine_to_bytes(o.F1, o.G1_GEN))
    (tmp_path / "pair_Q.bin").write_bytes(o.affine_to_bytes(o.F2, o.G2_GEN))
    (tmp_path / "pair_sP.bin").write_bytes(o.affine_to_bytes(o.F1, o.scalar_mul(o.F1, o.G1_GEN, s)))
    (tmp_path / "pair_sQ.bin").write_bytes(o.affine_to_bytes(o.F2, o.scalar_mul(o.F2, o.G2_GEN, s)))
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "g1 group_test OK" in out.stdout and "g2 group_test OK" in out.stdout and "pairing test OK" in out.stdout


def test_g1_deserialize_golden(ctx):
    """Every encoding class of tests/golden (valid, infinity, not in subgroup, no square root, not on curve, x >= p,
    bad flags) through mi_g1_deserialize_batch: status and decoded point."""
    for case in _golden()["g1_encoding"]:
        pts, st = ctx.g1_deserialize_batch(bytes.fromhex(case["bytes"]), case["compressed"], case["validate"])
        assert st[0] == case["status"], case["name"]
        assert pts == (bytes.fromhex(case["affine"]) if case["status"] == 0 else bytes(96)), case["name"]


@pytest.mark.parametrize("compressed", [True, False])
def test_g1_serialize_roundtrip_and_oracle(ctx, co, o, compressed):
    """encode -> decode round trip on 3000 points (with infinity sprinkled in) and the encoder against the oracle's
    to_compressed / to_uncompressed restatement (src/g1.rs:358-384)."""
    n = 3000
    raw = bytearray(co.gen_bases("g1", SEED_B + 40, n, 8))
    for i in range(0, n, 97):
        raw[96 * i:96 * (i + 1)] = bytes(96)
    raw = bytes(raw)
    enc = ctx.g1_serialize_batch(raw, compressed)
    size = 48 if compressed else 96
    for i in list(range(0, n, 131)) + [0, 97]:
        pt = o.affine_from_bytes(o.F1, raw[96 * i:96 * (i + 1)])
        want = o.g1_compress(pt) if compressed else o.g1_uncompressed(pt)
        assert enc[size * i:size * (i + 1)] == want, i
    dec, st = ctx.g1_deserialize_batch(enc, compressed, True)
    assert st == bytes(n) and dec == raw
    # flipping the sort flag of a compressed point yields the negated point
    if compressed:
        flipped = bytes([enc[48] ^ 0x20]) + enc[49:96]      # point 1 (point 0 was replaced by infinity)
        dec1, st1 = ctx.g1_deserialize_batch(flipped, True, True)
        assert st1 == b"\0"
        assert o.affine_from_bytes(o.F1, dec1) == o.aff_neg(o.F1, o.affine_from_bytes(o.F1, raw[96:192]))
        # and a sort flag on the infinity encoding is malformed
        assert ctx.g1_deserialize_batch(bytes([enc[0] ^ 0x20]) + enc[1:48], True, True)[1] == b"\x01"


def test_g1_deserialize_rejects_mixed_batch(ctx, co, o):
    """A batch mixing valid points with each rejection class keeps per-element status (SRS loading must not abort)."""
    g = _golden()["g1_encoding"]
    comp = [c for c in g if c["compressed"] and c["validate"]]
    blob = b"".join(bytes.fromhex(c["bytes"]) for c in comp)
    pts, st = ctx.g1_deserialize_batch(blob, True, True)
    assert list(st) == [c["status"] for c in comp]
    for k, c in enumerate(comp):
        assert pts[96 * k:96 * (k + 1)] == (bytes.fromhex(c["affine"]) if c["status"] == 0 else bytes(96))


def test_g1_msm_unreduced_canonical_scalars(ctx, co, o, pkg):
    """msm_bigint accepts any 256-bit integer: values >= r (up to 2^256 - 1) act as their residue mod r."""
    n = 40
    bases = co.gen_bases("g1", SEED_B + 50, n, 2)
    rnd = random.Random(50)
    vals = [(1 << 256) - 1, o.R_ORDER, o.R_ORDER + 1, 2 * o.R_ORDER + 5, (1 << 255) + 12345] + [rnd.randrange(1 << 256) for _ in range(n - 5)]
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    red = b"".join((v % o.R_ORDER).to_bytes(32, "little") for v in vals)
    got = ctx.msm("g1", bases, raw, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", got) == co.dlog_expected("g1", red, SEED_B + 50, n)


def test_g1_msm_2_22_closed_form_and_mixed_skew(ctx, co, pkg):
    """Scale check inside the suite (2^24 is exercised by bench.py's bit_exact flag): 2^22 points, uniform scalars with a
    quarter of them replaced by small / repeated values, against the closed form."""
    import numpy as np

    n = 1 << 22
    bases = co.gen_bases("g1", SEED_B + 60, n, 16)
    a = np.frombuffer(co.gen_scalars(SEED_S + 60, n), dtype=np.uint8).reshape(n, 32).copy()
    a[0:n // 8, 1:] = 0            # byte-sized scalars
    a[n // 8:n // 4] = a[n // 8]   # one value repeated 2^19 times
    scalars = a.tobytes()
    ctx.set_bases("g1", bases, n)
    got = ctx.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 60, n)


def test_g1_g2_msm_fuzz_configs(ctx, co, o, pkg):
    """Seeded fuzz over sizes, window sizes, scalar formats and scalar distributions (with infinity bases and repeated
    bases mixed in), G1 and G2, each against the oracle's Pippenger on the same inputs."""
    rnd = random.Random(2026)
    pool_n = 4000
    pools = {"g1": co.gen_bases("g1", SEED_B + 70, pool_n, 8), "g2": co.gen_bases("g2", SEED_B + 71, 1500, 8)}
    for it in range(40):
        group = "g1" if it % 4 else "g2"
        aff = 96 if group == "g1" else 192
        limit = pool_n if group == "g1" else 1500
        n = rnd.choice([1, 2, 5, 31, 64, 65, 100, 257, 1000, rnd.randrange(1, limit)])
        idx = [rnd.randrange(limit) for _ in range(n)]
        if rnd.random() < 0.5:                      # repeated bases
            idx = [idx[rnd.randrange(max(1, n // 4))] for _ in range(n)]
        bases = bytearray(b"".join(pools[group][aff * i:aff * (i + 1)] for i in idx))
        for k in range(n):
            if rnd.random() < 0.05:
                bases[aff * k:aff * (k + 1)] = bytes(aff)   # infinity
        kind = rnd.choice(["uniform", "bits", "small", "equal", "edge"])
        if kind == "uniform":
            sc = [rnd.randrange(o.R_ORDER) for _ in range(n)]
        elif kind == "bits":
            sc = [rnd.randrange(2) for _ in range(n)]
        elif kind == "small":
            sc = [rnd.randrange(1 << rnd.choice([8, 16, 40, 64])) for _ in range(n)]
        elif kind == "equal":
            sc = [rnd.randrange(o.R_ORDER)] * n
        else:
            sc = [rnd.choice([0, 1, o.R_ORDER - 1, o.R_ORDER - 2, (1 << 255) - 19 if False else 2]) for _ in range(n)]
        canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
        fmt = rnd.choice([pkg.SCALAR_CANONICAL, pkg.SCALAR_MONTGOMERY])
        data = canon if fmt == pkg.SCALAR_CANONICAL else co.fr_to_mont(canon)
        c = rnd.choice([0, 0, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
        ctx.set_window_bits(c)
        try:
            got = ctx.msm(group, bytes(bases), data, n, fmt)
        finally:
            ctx.set_window_bits(0)
        want = co.msm(group, bytes(bases), canon, n, 0, 4)
        assert _canon(co, group, got) == _canon(co, group, want), (it, group, n, kind, c, fmt)


def test_g2_deserialize_golden_and_roundtrip(ctx, co, o):
    """G2 encodings (src/g2.rs:338-411): every fixture class, then encode -> decode of 1500 points in both forms and the
    encoder against the oracle."""
    for case in _golden()["g2_encoding"]:
        pts, st = ctx.deserialize_batch("g2", bytes.fromhex(case["bytes"]), case["compressed"], case["validate"])
        assert st[0] == case["status"], case["name"]
        assert pts == (bytes.fromhex(case["affine"]) if case["status"] == 0 else bytes(192)), case["name"]
    n = 1500
    raw = bytearray(co.gen_bases("g2", SEED_B + 80, n, 8))
    for i in range(5, n, 113):
        raw[192 * i:192 * (i + 1)] = bytes(192)
    raw = bytes(raw)
    for compressed in (True, False):
        enc = ctx.serialize_batch("g2", raw, compressed)
        size = 96 if compressed else 192
        for i in range(0, n, 97):
            pt = o.affine_from_bytes(o.F2, raw[192 * i:192 * (i + 1)])
            assert enc[size * i:size * (i + 1)] == (o.g2_compress(pt) if compressed else o.g2_uncompressed(pt)), i
        dec, st = ctx.deserialize_batch("g2", enc, compressed, True)
        assert st == bytes(n) and dec == raw


def test_concurrent_calls_on_one_context(ctx, co, pkg):
    """a context runs two MSM calls at a time (two lanes sharing the resident bases): results from four host threads,
    each with its own scalars, interleaved with calls that need the context exclusively"""
    import threading

    n = 20000
    bases = co.gen_bases("g1", SEED_B + 90, n, 8)
    ctx.set_bases("g1", bases, n)
    scal = [co.gen_scalars(SEED_S + 90 + t, n) for t in range(4)]
    want = [co.dlog_expected("g1", s, SEED_B + 90, n) for s in scal]
    errs = []

    def worker(t):
        try:
            for it in range(6):
                got = ctx.msm("g1", None, scal[t], n, pkg.SCALAR_CANONICAL)
                if co.to_affine("g1", got) != want[t]:
                    errs.append((t, it, "mismatch"))
                if t == 3 and it % 2 == 0:   # exclusive entry points in between
                    jac = ctx.msm("g1", bases[:96 * 50], scal[t][:32 * 50], 50, pkg.SCALAR_CANONICAL)
                    if ctx.normalize_batch("g1", jac) != co.to_affine("g1", jac):
                        errs.append((t, it, "normalize"))
        except Exception as e:   # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_batch_over_resident_bases(ctx, co, pkg, group):
    """mi_msm_g{1,2}_batch: k scalar vectors over one resident base set, two in flight; every result against the closed form"""
    n = 6000 if group == "g1" else 2500
    bases = co.gen_bases(group, SEED_B + 95, n, 8)
    ctx.set_bases(group, bases, n)
    vecs = [co.gen_scalars(SEED_S + 95 + j, n) for j in range(5)]
    got = ctx.msm_batch(group, vecs, n, pkg.SCALAR_CANONICAL)
    assert len(got) == 5
    for j in range(5):
        assert co.to_affine(group, got[j]) == co.dlog_expected(group, vecs[j], SEED_B + 95, n)
    assert ctx.msm_batch(group, [], n) == []
    one = ctx.msm_batch(group, vecs[:1], n, pkg.SCALAR_CANONICAL)
    assert co.to_affine(group, one[0]) == co.to_affine(group, got[0])
    mont = ctx.msm_batch(group, [co.fr_to_mont(v) for v in vecs[:2]], n, pkg.SCALAR_MONTGOMERY)
    assert [co.to_affine(group, x) for x in mont] == [co.to_affine(group, x) for x in got[:2]]
    # the same vectors from device memory
    import torch

    dv = [torch.frombuffer(bytearray(v), dtype=torch.uint8).cuda() for v in vecs]
    torch.cuda.synchronize()
    dev = ctx.msm_batch_device(group, [t.data_ptr() for t in dv], n, pkg.SCALAR_CANONICAL)
    assert [co.to_affine(group, x) for x in dev] == [co.to_affine(group, x) for x in got]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_cofactor_clearing_through_the_hip_path(ctx, co, o, group):
    """The same property on the GPU (tests/test_oracle.py::test_cofactor_clearing_pins_the_group_law holds it for the two oracles):
    a curve point OUTSIDE the prime-order subgroup times the reference-held cofactor (src/g1.rs:42, src/g2.rs:45-54), computed by
    mi_msm_g{1,2}, equals both oracles' value, is not infinity, and r times it is infinity."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cofactor_util import H1, H2, off_subgroup_points, h2_pieces, SPLIT_BITS

    F = o.F1 if group == "g1" else o.F2
    h = H1 if group == "g1" else H2
    for pt in off_subgroup_points(o, group, 3):
        want = o.affine_to_bytes(F, o.scalar_mul(F, pt, h))
        if group == "g1":
            bases, sc, n = o.affine_to_bytes(F, pt), h.to_bytes(32, "little"), 1
        else:
            bases = b"".join(o.affine_to_bytes(F, o.scalar_mul(F, pt, 1 << (SPLIT_BITS * i))) for i in range(3))
            sc, n = b"".join(a.to_bytes(32, "little") for a in h2_pieces()), 3
        got = _canon(co, group, ctx.msm(group, bases, sc, n, 0))
        assert got == want and got != bytes(len(want))
        assert got == co.to_affine(group, co.msm(group, bases, sc, n, 0, 1))
        # r = (r - 1) + 1: a single scalar ABOVE (r - 1) / 2 on a point outside the subgroup — the library multiplies by the integer, as blst does
        two = (o.R_ORDER - 1).to_bytes(32, "little") + (1).to_bytes(32, "little")
        rq = ctx.msm(group, got + got, two, 2, 0)
        assert _canon(co, group, rq) == bytes(len(want))               # r * (h P) = infinity
        # ... and r * P itself is NOT: the point really was outside the subgroup (so the property above is not vacuous)
        raw = o.affine_to_bytes(F, pt)
        rp = ctx.msm(group, raw + raw, two, 2, 0)
        assert _canon(co, group, rp) == o.affine_to_bytes(F, o.scalar_mul(F, pt, o.R_ORDER))
        assert _canon(co, group, rp) != bytes(len(want))
        # (r - 1) P and (h k) P for a 254-bit k, each ONE scalar on the off-subgroup point, against the big-int oracle
        k254 = (1 << 253) + 0x1234567 * (1 << 100) + 12345
        for s in (o.R_ORDER - 1, (o.R_ORDER + 1) // 2, k254 | 1):
            one = ctx.msm(group, raw, s.to_bytes(32, "little"), 1, 0)
            assert _canon(co, group, one) == o.affine_to_bytes(F, o.scalar_mul(F, pt, s)), hex(s)
        hk = ctx.msm(group, got, (k254).to_bytes(32, "little"), 1, 0)   # (h P) is in the subgroup: k (h P) has order r
        assert _canon(co, group, hk) == o.affine_to_bytes(F, o.scalar_mul(F, o.affine_from_bytes(F, got), k254))


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_unvalidated_resident_bases_keep_integer_semantics(pkg, co, o, group):
    """the reference multiplies whatever G1Affine it holds by the integer s (blst's Pippenger, src/g1.rs:614-617; points
    that skipped Valid::check exist: Validate::No, src/g1.rs:425).  A resident set with off-subgroup points in it: validate_bases counts
    them and leaves the sign fold OFF, the MSM over it equals the big-int oracle's sum for scalars above (r - 1) / 2, at a window size
    where the fold would change the window count (c = 15) and one where it would not (c = 16)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cofactor_util import off_subgroup_points

    F = o.F1 if group == "g1" else o.F2
    aff = 96 if group == "g1" else 192
    n = 64
    good = co.gen_bases(group, 4242, n - 3, 2)
    offp = off_subgroup_points(o, group, 3)
    bases = good[:aff * 20] + o.affine_to_bytes(F, offp[0]) + good[aff * 20:aff * 40] + o.affine_to_bytes(F, offp[1]) + good[aff * 40:] + o.affine_to_bytes(F, offp[2])
    rng = random.Random(77)
    sc = [rng.getrandbits(256) % o.R_ORDER for _ in range(n)]
    sc[20], sc[41], sc[63] = o.R_ORDER - 1, o.R_ORDER - 5, (o.R_ORDER + 3) // 2     # the off-subgroup points get scalars above (r - 1) / 2
    scb = b"".join(x.to_bytes(32, "little") for x in sc)
    want = o.INF
    for i in range(n):
        want = o.aff_add(F, want, o.scalar_mul(F, o.affine_from_bytes(F, bases[aff * i:aff * (i + 1)]), sc[i]))
    want = o.affine_to_bytes(F, want)
    with pkg.Context([0]) as c:
        c.set_bases(group, bases, n)
        assert c.validate_bases(group) == 3
        for cb, windows in ((15, 18), (16, 16), (17, 16), (0, None)):
            c.set_window_bits(cb)
            assert _canon(co, group, c.msm(group, None, scb, n, 0)) == want, cb
            assert windows is None or c.profile()["num_windows"] == windows
        assert _canon(co, group, c.msm(group, bases, scb, n, 0)) == want             # bases passed with the call: never folded


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_validated_bases_fold_signs_with_identical_results(pkg, co, group):
    """A resident set that passes validate_bases: the MSM recodes min(s, r - s) (one window fewer at c = 15 / 17) and returns the same
    point as the unvalidated context and as the C oracle; a new set_bases clears the record."""
    n = 3000
    bases = co.gen_bases(group, 991, n, 2)
    sc = co.gen_scalars(992, n)
    want = co.to_affine(group, co.msm(group, bases, sc, n, 0, 2))
    with pkg.Context([0]) as c:
        c.set_bases(group, bases, n)
        for cb, w_plain, w_fold in ((15, 18, 17), (17, 16, 15), (16, 16, 16), (13, 20, 20)):
            c.set_window_bits(cb)
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want
            assert c.profile()["num_windows"] == w_plain
        assert c.validate_bases(group) == 0
        for cb, w_plain, w_fold in ((15, 18, 17), (17, 16, 15), (16, 16, 16), (13, 20, 20)):
            c.set_window_bits(cb)
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want
            assert c.profile()["num_windows"] == w_fold
            assert _canon(co, group, c.msm(group, None, sc, n, 1)) == co.to_affine(group, co.msm(group, bases, sc, n, 1, 2))   # Montgomery scalars too
        c.set_bases(group, bases, n)            # a new set: the record is gone
        c.set_window_bits(15)
        assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want
        assert c.profile()["num_windows"] == 18


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_base_set_cache_for_the_stateless_call(pkg, co, group):
    """mi_msm_g{1,2} with host bases and the base-set cache on.  Second call with the same slice = a hit, same point as the
    uncached context and the C oracle; a slice REWRITTEN in place misses and gives the new value (single-point edits anywhere:
    test_base_set_cache_sees_every_byte); three
    slices through a two-entry cache evict the least recently used; invalidate empties it; below 4096 points nothing is cached;
    ARKBLST_AMD_BASE_CACHE=0 overrides the call."""
    import numpy as np

    aff = 96 if group == "g1" else 192
    n = 6000
    sets = [np.frombuffer(co.gen_bases(group, 555 + k, n, 4), dtype=np.uint8).copy() for k in range(3)]
    sc = co.gen_scalars(556, n)
    want = [co.to_affine(group, co.msm(group, b.tobytes(), sc, n, 0, 4)) for b in sets]
    with pkg.Context([0]) as c:
        c.set_base_cache(2)
        for rep in range(3):
            assert _canon(co, group, c.msm(group, sets[0], sc, n, 0)) == want[0]
        st = c.base_cache_stats()
        assert (st["hits"], st["misses"], st["entries"]) == (2, 1, 1), st
        # rewrite point 0 in place: same pointer, same length, different content
        sets[0][:aff] = sets[1][:aff]
        changed = co.to_affine(group, co.msm(group, sets[0].tobytes(), sc, n, 0, 4))
        assert changed != want[0]
        assert _canon(co, group, c.msm(group, sets[0], sc, n, 0)) == changed
        assert c.base_cache_stats()["misses"] == 2
        # LRU: sets[1], sets[2] push the two entries of sets[0] out; sets[1] stays warm
        for k in (1, 2, 1, 2):
            assert _canon(co, group, c.msm(group, sets[k], sc, n, 0)) == want[k]
        st = c.base_cache_stats()
        assert st["entries"] == 2 and st["misses"] == 4 and st["hits"] == 4, st
        assert _canon(co, group, c.msm(group, sets[0], sc, n, 0)) == changed and c.base_cache_stats()["misses"] == 5
        # Montgomery scalars and a shorter prefix of a cached vector (a different key: n is part of it)
        assert _canon(co, group, c.msm(group, sets[0], co.fr_to_mont(sc), n, 1)) == changed
        assert _canon(co, group, c.msm(group, sets[0], sc, 5000, 0)) == co.to_affine(group, co.msm(group, sets[0].tobytes(), sc, 5000, 0, 4))
        c.invalidate_base_cache()
        assert c.base_cache_stats()["entries"] == 0
        before = c.base_cache_stats()
        assert _canon(co, group, c.msm(group, sets[0][:aff * 1000], sc, 1000, 0)) == co.to_affine(group, co.msm(group, sets[0].tobytes(), sc, 1000, 0, 4))
        assert c.base_cache_stats() == before          # under 4096 points: not cached, not counted
        c.set_base_cache(0)
        assert _canon(co, group, c.msm(group, sets[1], sc, n, 0)) == want[1] and c.base_cache_stats() == before
    os.environ["ARKBLST_AMD_BASE_CACHE"] = "0"
    try:
        with pkg.Context([0]) as c:
            c.set_base_cache(2)                         # ignored: the environment decides
            for rep in range(2):
                assert _canon(co, group, c.msm(group, sets[2], sc, n, 0)) == want[2]
            assert c.base_cache_stats() == {"hits": 0, "misses": 0, "entries": 0}
    finally:
        del os.environ["ARKBLST_AMD_BASE_CACHE"]
    # two host threads on one context, the same slice: both lanes may miss at once, one entry survives, both results right
    import threading

    with pkg.Context([0]) as c:
        c.set_base_cache(2)
        res = [None, None]

        def work(t):
            for _ in range(3):
                res[t] = c.msm(group, sets[1], sc, n, 0)

        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        assert _canon(co, group, res[0]) == want[1] and _canon(co, group, res[1]) == want[1]
        assert c.base_cache_stats()["entries"] == 1


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_base_set_cache_sees_every_byte(pkg, co, group):
    """VERDICT r05 #4 / ADVICE r05 (high): the cache's fingerprint covers EVERY byte of the base vector (rounds 4-5 sampled 1024 points, and an
    edit outside the sample was a silent hit on stale device data).  One limb of a point the old sample skipped (index 1 of 6000; one deep
    inside a 2^16-point vector that is hashed in several slices) is changed in place between two calls: the call returns the oracle's NEW sum
    and counts a miss; restoring the bytes is a hit on the first entry again.  Also through the trait mirror's default context (msm.py), whose
    cache is on by default."""
    import numpy as np

    aff = 96 if group == "g1" else 192
    for n, idx in ((6000, 1), (1 << 16, 40001)):
        a = np.frombuffer(co.gen_bases(group, 911, n, 8), dtype=np.uint8).copy()
        other = np.frombuffer(co.gen_bases(group, 912, n, 8), dtype=np.uint8)
        sc = co.gen_scalars(913, n)
        want = co.to_affine(group, co.msm(group, a.tobytes(), sc, n, 0, 8))
        with pkg.Context([0]) as c:
            c.set_base_cache(4)                                                  # room for the three versions of the vector below
            for _ in range(2):
                assert _canon(co, group, c.msm(group, a, sc, n, 0)) == want
            assert c.base_cache_stats()["hits"] == 1
            keep = a[idx * aff:(idx + 1) * aff].copy()
            a[idx * aff:(idx + 1) * aff] = other[idx * aff:(idx + 1) * aff]     # another valid point, same address, same length
            changed = co.to_affine(group, co.msm(group, a.tobytes(), sc, n, 0, 8))
            assert changed != want
            assert _canon(co, group, c.msm(group, a, sc, n, 0)) == changed
            st = c.base_cache_stats()
            assert (st["hits"], st["misses"]) == (1, 2), st
            # ONE limb (8 bytes of x) of that point restored, the rest not: not a curve point any more, but the call must not serve either cached set —
            # compare against the oracle on exactly these bytes
            a[idx * aff:idx * aff + 8] = keep[:8]
            if bytes(a[idx * aff:idx * aff + 8]) != bytes(other[idx * aff:idx * aff + 8]):
                assert c.base_cache_stats()["misses"] == 2
                c.msm(group, a, sc, n, 0)                                        # value undefined (off-curve input), the bookkeeping is not
                assert c.base_cache_stats()["misses"] == 3
            a[idx * aff:(idx + 1) * aff] = keep
            assert _canon(co, group, c.msm(group, a, sc, n, 0)) == want
            assert c.base_cache_stats()["hits"] == 2                              # the first entry, found again by content
    # the trait mirror: msm(bases, scalars) is a pure function of its arguments, cache or not
    from ark_blst_amd import msm as trait

    cls = trait.G1Projective if group == "g1" else trait.G2Projective
    n = 6000
    a = np.frombuffer(co.gen_bases(group, 921, n, 8), dtype=np.uint8).copy()
    other = np.frombuffer(co.gen_bases(group, 922, n, 8), dtype=np.uint8)
    sc = co.gen_scalars(923, n)
    for step in range(3):
        got = cls.msm(a, sc, scalar_fmt=0)
        assert _canon(co, group, got) == co.to_affine(group, co.msm(group, a.tobytes(), sc, n, 0, 8)), step
        j = 1 + 37 * step
        a[j * aff:(j + 1) * aff] = other[j * aff:(j + 1) * aff]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_check_batch_is_valid_check(ctx, co, o, group):
    """mi_g{1,2}_check_batch = Valid::check per point (src/g1.rs:386-396, src/g2.rs:366-376): subgroup points and infinity pass, curve
    points outside the subgroup get 3, points off the curve get 2; against the big-int oracle's on_curve / in_subgroup on every point,
    and the input is left untouched."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cofactor_util import off_subgroup_points

    F = o.F1 if group == "g1" else o.F2
    aff = 96 if group == "g1" else 192
    in_sub = o.g1_in_subgroup if group == "g1" else o.g2_in_subgroup
    good = co.gen_bases(group, 6061, 500, 4)
    pts = [good[aff * i:aff * (i + 1)] for i in range(500)]
    off = [o.affine_to_bytes(F, p) for p in off_subgroup_points(o, group, 3)]
    bad = []
    for i in range(3):   # y + 1: not on the curve
        p = o.affine_from_bytes(F, pts[i])
        y1 = (p[1] + 1) % o.P if group == "g1" else ((p[1][0] + 1) % o.P, p[1][1])
        assert not o.on_curve(F, (p[0], y1))
        bad.append(o._felt_bytes(F, p[0]) + o._felt_bytes(F, y1))
    blob = pts[:100] + [off[0], bytes(aff)] + pts[100:300] + [bad[0], off[1], bad[1]] + pts[300:] + [off[2], bad[2], bytes(aff)]
    want = bytearray()
    for b in blob:
        if b == bytes(aff):
            want.append(0)
            continue
        p = o.affine_from_bytes(F, b)
        want.append(2 if not o.on_curve(F, p) else (0 if in_sub(p) else 3))
    data = b"".join(blob)
    st = ctx.check_batch(group, data)
    assert st == bytes(want)
    assert st.count(3) == 3 and st.count(2) == 3 and len(st) == 508
    assert ctx.check_batch(group, b"") == b""


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_subgroup_ladder_on_small_order_points(ctx, co, o, group):
    """Round 6: the subgroup tests run on Jacobian ladders whose additions are not complete (ec.cuh jac_add).  Points of SMALL order are the
    inputs that reach the exceptional cases ([k]P == +-P, [k]P == infinity in mid-ladder): orders 3, 11, 10177 on E(Fp) and 13, 23, 2713 on
    E'(Fp2) (the prime factors of the cofactors the reference holds, src/g1.rs:42, src/g2.rs:45-54).  All of them are curve points outside
    the subgroup: Valid::check must say 3 and the validating decoder must reject them, next to subgroup points that pass."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cofactor_util import H1, H2, off_subgroup_points

    F = o.F1 if group == "g1" else o.F2
    aff = 96 if group == "g1" else 192
    h, primes = (H1, (3, 11, 10177)) if group == "g1" else (H2, (13, 23, 2713))
    n_curve = h * o.R_ORDER
    small = []
    for q in primes:
        assert h % q == 0
        cof = n_curve
        while cof % q == 0:
            cof //= q
        for base in off_subgroup_points(o, group, 3):
            t = o.scalar_mul(F, base, cof)
            if t is o.INF:
                continue
            while o.scalar_mul(F, t, q) is not o.INF:
                t = o.scalar_mul(F, t, q)
            assert o.on_curve(F, t)
            small.append(o.affine_to_bytes(F, t))
            small.append(o.affine_to_bytes(F, o.aff_neg(F, t)))
    assert len(small) >= 6
    good = co.gen_bases(group, 6262, 64, 4)
    blob = []
    for i, b in enumerate(small):
        blob += [good[aff * (2 * i % 64):aff * (2 * i % 64 + 1)], b]
    st = ctx.check_batch(group, b"".join(blob))
    assert st == bytes([0, 3] * len(small))
    ser = ctx.serialize_batch(group, b"".join(blob), compressed=True)
    pts, st2 = ctx.deserialize_batch(group, ser, compressed=True, validate=True)
    assert bytes(st2) == bytes([0, 3] * len(small))
    for i, b in enumerate(blob):
        assert bytes(pts[aff * i:aff * (i + 1)]) == (b if i % 2 == 0 else bytes(aff))


def test_plain_cpp_harness_of_the_exchange(tmp_path, co):
    """examples/multi_gpu_msm.cpp: the one-process-per-GPU deployment in plain C++ against the two C ABIs — no Python host, no torch, the
    system's RCCL — built with g++ and run as ONE rank on this box's GPU (rank 0 writes the ncclUniqueId file, creates the communicator,
    runs mi_msm_g1_allgather_fold over device-resident scalars): the 144-byte result equals the C oracle's."""
    import json
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = os.path.join(here, "host", "multi_gpu_msm")
    lib = os.path.join(root, "ark-blst_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe,
                           os.path.join(root, "examples", "multi_gpu_msm.cpp"), "-L" + lib, "-larkblst_amd_rccl", "-larkblst_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    n = 70000
    bases = co.gen_bases("g1", 9911, n, 8)
    sc = co.gen_scalars(9912, n)
    (tmp_path / "bases.bin").write_bytes(bases)
    (tmp_path / "scalars.bin").write_bytes(sc)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([exe, "0", "1", str(tmp_path / "id"), str(tmp_path / "bases.bin"), str(tmp_path / "scalars.bin"), str(tmp_path / "out.bin"), "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_ranks"] == 1 and line["points_total"] == n and line["ms"] > 0 and line["exchange_ms"] > 0
    got = (tmp_path / "out.bin.0").read_bytes()
    assert len(got) == 144 and co.to_affine("g1", got) == co.dlog_expected("g1", sc, 9911, n)


def test_misaligned_device_pointer_is_an_error_not_a_fault(pkg, co):
    """device scalars are read as 16-byte vectors: a device pointer off that alignment comes back as MI_E_INVALID (the GPU is never asked)"""
    import torch

    n = 1000
    bases = co.gen_bases("g1", 1212, n, 2)
    d = torch.zeros(32 * n + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pkg.Context([0]) as c:
        c.set_bases("g1", bases, n)
        for off in (4, 8, 1):
            with pytest.raises(pkg.MsmError) as ei:
                c.msm_device("g1", d.data_ptr() + off, n, pkg.SCALAR_CANONICAL)
            assert ei.value.code == -1 and "aligned" in str(ei.value)
        assert c.msm_device("g1", d.data_ptr() + 16, n, pkg.SCALAR_CANONICAL) == bytes(144)   # all-zero scalars: infinity


def test_call_abi_reproducer():
    """The compiler issue behind round 3's "codegen-dependent miscompares" (DESIGN_HISTORY.md §9, csrc/Makefile): tools/call_abi/repro_tower.hip
    — the test-only single-lane Miller loop's shape: a 512-register kernel that keeps the point and the line state across ~40 calls
    of out-of-line tower functions per round — built twice from the shipped headers.  With the library's flags (VGPR spill slots are
    NOT turned into AGPRs) it must agree with the host run of the same source on all 64 lanes; with the compiler's default it is
    EXPECTED to differ while the compiler issue exists.  Should a later compiler fix it, this test says so instead of failing."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "ark-blst_amd", "lib")
    good = subprocess.run([os.path.join(lib, "repro_call_abi")], capture_output=True, text=True, timeout=300)
    assert good.returncode == 0 and "miller loop, 63 rounds: 0 of 64" in good.stdout, good.stdout[-1500:] + good.stderr[-500:]
    # The expected-to-FAIL half (the same source with the compiler's default flags) is opt-in: a miscompiled 512-register kernel has only
    # ever corrupted data values, but nothing says the next compiler will not clobber an address register on a shared box.
    # `make -C ark-blst_amd/csrc repro-default` builds it into tools/call_abi/; ARKBLST_RUN_BROKEN_REPRO=1 runs it.
    broken = os.path.join(root, "tools", "call_abi", "repro_call_abi_compiler_default")
    if os.environ.get("ARKBLST_RUN_BROKEN_REPRO") != "1" or not os.path.exists(broken):
        return
    dflt = subprocess.run([broken], capture_output=True, text=True, timeout=300)
    # the isolated tower functions are right either way; only the loop kernel is affected
    for piece in ("conj12", "sqr12", "mul_by_014", "chain"):
        assert f"{piece}: 0 of 64" in dflt.stdout, dflt.stdout[-1500:]
    if dflt.returncode == 0:
        print("NOTE: the reproducer passes with the compiler's default flags: the compiler issue is gone on this toolchain")
    else:
        assert "miller loop, 1 rounds: 64 of 64" in dflt.stdout, dflt.stdout[-1500:]


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (small size, as a child process)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--log-n", "14", "--steps", "2", "--warmup", "1", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "bit_exact"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["bit_exact"] is True and d["vs_baseline"] is None
    assert d["unit"] == "points/s" and d["higher_is_better"] is True and "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    # the roofline key names the BINDING bound of this path (integer multiply-add issue); the HBM line the north star asks for sits beside it
    assert d["roofline"]["bound"] == "valu_int_mad" and d["roofline"]["unit"] == "T MAD/s" and 0.05 < d["roofline"]["frac"] < 1.0
    assert d["hbm_roofline"]["bound"] == "hbm" and d["hbm_roofline"]["unit"] == "GB/s" and "valu_roofline" not in d
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port"
    assert d["summary"]["g1_2p14"]["bit_exact"] is True and 0.0 < d["step_frac"] < d["roofline"]["frac"]
    assert r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192   # the record is the LAST stdout line and small


def test_bench_default_invocation_is_one_small_line(tmp_path):
    """the DEFAULT invocation — every secondary leg on, the form the driver runs — prints exactly one JSON line, the last
    line of stdout, under 8 KB, with `roofline`, `cpu_baseline` and `step_frac` in it; the legs' full records are in the sidecar file."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    side = str(tmp_path / "secondary.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--secondary-out", side],
                       capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0]) < 8192, len(lines[0])
    d = json.loads(lines[0])
    assert d["bit_exact"] is True and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["workload"].startswith("G1 MSM, 2^20") and d["dtype"] == "u32" and d["vs_baseline"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert 0.0 < d["step_frac"] <= d["roofline"]["frac"] < 1.0
    assert "secondary" not in d and "pairing_2p16" not in d and "valu_roofline" not in d
    sec = json.load(open(side))["secondary"]
    for leg in ("g1_2p16", "g1_2p24", "g2_2p20", "pairing_2p16", "normalize_2p20", "deserialize_2p20", "normalize_g2_2p20", "deserialize_g2_2p18",
                "call_shapes", "two_host_threads"):
        assert leg in sec and "error" not in sec[leg], (leg, sec.get(leg))
    for leg in ("g1_2p16", "g1_2p24", "g2_2p20", "pairing_2p16", "normalize_2p20", "deserialize_2p20", "normalize_g2_2p20", "deserialize_g2_2p18"):
        assert sec[leg]["bit_exact"] is True, leg
        assert d["summary"][leg]["bit_exact"] is True


def test_bench_in_process_leg_as_child_process():
    """bench.py runs its in-library multi-device leg in a child process when several GPUs are visible (the first run on distinct
    physical devices happens on the driver's node: a fault there must not cost the headline line).  The child's entry, on this box's
    one GPU: device 0 listed twice."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--in-process-child", "--in-process", "2"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["device_scalars"]["bit_exact"] is True and d["host_scalars"]["bit_exact"] is True
    assert d["device_slots"] == [0, 0] and d["plumbing_only"] is True


def test_bench_two_ranks_exchange_from_device_memory():
    """The N > 1 path of bench.py end to end on ONE GPU (2 ranks share device 0, gloo collective): every rank leaves its window
    sums in device memory (mi_msm_g1_device_windows), they are all-gathered and folded (mi_g1_fold_windows); the line carries
    msm_ms / exchange_ms and the result is bit-exact against the closed form over both shards."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300),
           os.path.join(root, "bench.py"), "--gpus", "2", "--total-log-n", "15", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--share-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["bit_exact"] is True and d["scaling"] == "strong" and d["config"]["total_points"] == 1 << 15
    assert d["exchange_ms"] >= 0 and d["msm_ms"] > 0 and d["exchange"]["windows"] == d["config"]["num_windows"]
    assert d["exchange"]["backend"] == "gloo" and len(d["exchange"]["rank_msm_ms"]) == 2 and "expected_ms_per_rank" in d["config"]


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with NO launcher in the command and no WORLD_SIZE in the environment (the form the driver uses for
    N = 1): bench.py starts its two ranks itself; here they share GPU 0 and meet over gloo (RCCL refuses two ranks per device)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-log-n", "15", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--share-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["bit_exact"] is True and d["scaling"] == "strong" and d["exchange"]["world_size"] == 2
    assert d["exchange"]["exchange_incl_wait_ms"] > 0 and d["wait_ms"] >= 0   # split from the ranks' local times: no barrier inside the timed steps


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_bench_exchange_runs_under_one_rank_rccl(group):
    """The exchange of bench.py's N > 1 path on a one-GPU box: --force-exchange runs it at world size 1 — mi_msm_g{1,2}_allgather_fold of
    libarkblst_amd_rccl.so (window sums in device memory, ncclAllGather on the device buffer, the pinned D2H copy, the fold), under a
    one-rank RCCL communicator created from an ncclUniqueId — and the result is bit-exact.  (The gloo rehearsals never execute it.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-exchange", "--group", group, "--log-n", "16", "--steps", "3", "--warmup", "1",
           "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["bit_exact"] is True and d["metric"].startswith(group.upper())
    assert d["exchange"]["backend"].startswith("rccl") and d["exchange"]["world_size"] == 1 and d["exchange"]["windows"] == d["config"]["num_windows"]
    assert d["exchange_ms"] > 0 and d["msm_ms"] > 0 and d["exchange"]["window_size_repeats_in_timed_steps"] == 0
    assert "fallback_reason" not in d["exchange"] and d["exchange"]["backend"] == "rccl (in-library)"
    if group == "g1":
        # round 6: the in-library exchange has never met a second physical GPU; when any rank cannot set it up, ALL ranks switch (a collective
        # decision) to torch.distributed's all-gather on the device window sums and the line says why.  Forced here through the environment.
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, ARKBLST_AMD_BENCH_EXCHANGE="torch"))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["bit_exact"] is True and d["exchange"]["backend"].startswith("rccl (torch.distributed")
        assert "ARKBLST_AMD_BENCH_EXCHANGE" in d["exchange"]["fallback_reason"] and d["exchange"]["path"].startswith("FALLBACK")


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_rccl_allgather_fold_world_size_one(pkg, co, group):
    """libarkblst_amd_rccl.so through its C ABI at world size 1: the collective result equals mi_msm_{g}_device on the same inputs and the
    C oracle; n = 0 gives infinity; a shorter call after a longer one (stale window slots) is right; the timing record is filled."""
    import torch

    n = 5000
    jac = 144 if group == "g1" else 288
    bases = co.gen_bases(group, 31337, n, 4)
    sc = co.gen_scalars(31338, n)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with pkg.Context([0]) as c:
        c.set_bases(group, bases, n)
        with pkg.RcclComm(c, pkg.rccl_unique_id(), 1, 0) as comm:
            assert comm.size() == 1 and comm.rank() == 0
            got = comm.allgather_fold(group, d_sc.data_ptr(), n, pkg.SCALAR_CANONICAL)
            assert _canon(co, group, got) == _canon(co, group, c.msm_device(group, d_sc.data_ptr(), n, pkg.SCALAR_CANONICAL))
            assert _canon(co, group, got) == _canon(co, group, co.msm(group, bases, sc, n, 0, 4))
            t = comm.timing()
            assert t["msm_ms"] > 0 and t["exchange_ms"] > 0 and t["repeats"] == 0 and t["num_windows"] > 0 and t["bytes_per_rank"] == 38 * jac
            assert comm.allgather_fold(group, 0, 0, pkg.SCALAR_CANONICAL) == bytes(jac)      # nobody has a point: infinity
            # a rank whose local part fails (n beyond its resident shard) still joins the collective and gets ITS error; the communicator lives on
            with pytest.raises(pkg.MsmError) as ei:
                comm.allgather_fold(group, d_sc.data_ptr(), n + 1, pkg.SCALAR_CANONICAL)
            assert ei.value.code == -1 and "resident" in str(ei.value)
            short = comm.allgather_fold(group, d_sc.data_ptr(), 100, pkg.SCALAR_CANONICAL)
            assert _canon(co, group, short) == _canon(co, group, co.msm(group, bases[:100 * (jac * 2 // 3)], sc[:3200], 100, 0, 1))
            # round 6: the call leaves the context's window-size setting as it found it — pinned by the caller ...
            comm.set_timeout_ms(5000)
            c.set_window_bits(11)
            pinned = comm.allgather_fold(group, d_sc.data_ptr(), n, pkg.SCALAR_CANONICAL)
            assert _canon(co, group, pinned) == _canon(co, group, got) and comm.timing()["window_bits"] == 11 and c.get_window_bits() == 11
            c.set_window_bits(0)
            # ... or not at all
            comm.allgather_fold(group, d_sc.data_ptr(), n, pkg.SCALAR_CANONICAL)
            assert c.get_window_bits() == 0
        # a multi-device context is refused (one context per rank)
        with pkg.Context([0, 0]) as c2:
            with pytest.raises(pkg.MsmError):
                pkg.RcclComm(c2, pkg.rccl_unique_id(), 1, 0)


# ------------------------------------------------------------------------------------------------ BASELINE sizes
def test_g1_msm_2_24_closed_form(pkg, co):
    """BASELINE config #3's size on ONE device (the north-star target): 2^24 random bases + scalars, resident bases, scalars
    from host memory, against the closed form (sum s_i k_i) G; then the same through device-resident scalars."""
    import torch

    n = 1 << 24
    bases = co.gen_bases("g1", SEED_B + 24, n, 16)
    scalars = co.gen_scalars(SEED_S + 24, n)
    want = co.dlog_expected("g1", scalars, SEED_B + 24, n)
    with pkg.Context([0]) as c:
        c.set_bases("g1", bases, n)
        del bases
        assert _canon(co, "g1", c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        prof = c.profile()
        assert prof["n"] == n and prof["window_bits"] >= 16
        d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        assert _canon(co, "g1", c.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL)) == want
        # a prefix of the resident set (2^24 - 12345 points): the shard shapes an 8-way split produces are not powers of two
        m = n - 12345
        assert _canon(co, "g1", c.msm("g1", None, scalars[:32 * m], m, pkg.SCALAR_CANONICAL)) == co.dlog_expected("g1", scalars[:32 * m], SEED_B + 24, m)


def test_g2_msm_2_20_closed_form(pkg, co):
    """BASELINE config #4: G2, 2^20 random bases + scalars on one device, against the closed form."""
    n = 1 << 20
    bases = co.gen_bases("g2", SEED_B + 4, n, 16)
    scalars = co.gen_scalars(SEED_S + 4, n)
    with pkg.Context([0]) as c:
        c.set_bases("g2", bases, n)
        got = c.msm("g2", None, scalars, n, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g2", got) == co.dlog_expected("g2", scalars, SEED_B + 4, n)


def _mixed_scalars(o, rnd, n):
    """uniform, 0/1, small, equal and edge values mixed in one vector"""
    out = []
    same = rnd.randrange(o.R_ORDER)
    for i in range(n):
        k = i % 7
        out.append(rnd.randrange(o.R_ORDER) if k < 2 else rnd.randrange(2) if k == 2 else rnd.randrange(1 << 40) if k == 3 else same if k == 4
                   else rnd.choice([0, 1, 2, o.R_ORDER - 1, o.R_ORDER - 2]) if k == 5 else rnd.randrange(1 << 200))
    return out


@pytest.mark.parametrize("c", [17, 18, 19, 20, 21, 22])
def test_large_window_sizes_small_inputs(ctx, co, o, pkg, c):
    """window sizes 17..22 (static k_coarse instantiations with their own top-window shapes; the plan only picks them for
    millions of points) forced on small inputs, both groups, with edge / skewed scalars, infinity and repeated bases"""
    rnd = random.Random(1700 + c)
    for group, n in (("g1", 777), ("g1", 5000), ("g2", 300)):
        aff = 96 if group == "g1" else 192
        bases = bytearray(co.gen_bases(group, SEED_B + 170 + c, n, 8))
        for k in range(0, n, 97):
            bases[aff * k:aff * (k + 1)] = bytes(aff)                          # infinity
        for k in range(5, n, 53):
            bases[aff * k:aff * (k + 1)] = bases[aff * 1:aff * 2]              # repeated point
        sc = _mixed_scalars(o, rnd, n)
        canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
        ctx.set_window_bits(c)
        try:
            got = ctx.msm(group, bytes(bases), canon, n, pkg.SCALAR_CANONICAL)
            got_m = ctx.msm(group, bytes(bases), co.fr_to_mont(canon), n, pkg.SCALAR_MONTGOMERY)
            assert ctx.profile()["window_bits"] == c
        finally:
            ctx.set_window_bits(0)
        want = _canon(co, group, co.msm(group, bytes(bases), canon, n, 0, 4))
        assert _canon(co, group, got) == want, (group, n, c)
        assert _canon(co, group, got_m) == want, (group, n, c, "montgomery")


# ------------------------------------------------------------------------------------------------ precomputed tables (opt-in)
@pytest.mark.parametrize("group,n,c", [("g1", 1, 0), ("g1", 300, 8), ("g1", 5000, 0), ("g1", 5000, 13), ("g1", 40000, 16), ("g2", 900, 0), ("g2", 900, 11)])
def test_precomputed_tables_match_plain(pkg, co, o, group, n, c):
    """mi_msm_g{1,2}_set_bases_precomputed: resident 2^(c j) P tables, all windows share one bucket set.  Same results as the
    plain resident set and the oracle: uniform and skewed scalars, both scalar formats, infinity and repeated bases,
    a prefix of the resident set."""
    rnd = random.Random(4242 + n + c)
    aff = 96 if group == "g1" else 192
    bases = bytearray(co.gen_bases(group, SEED_B + 200 + n, n, 8))
    for k in range(3, n, 61):
        bases[aff * k:aff * (k + 1)] = bytes(aff)
    for k in range(7, n, 41):
        bases[aff * k:aff * (k + 1)] = bases[aff * 2:aff * 3]
    bases = bytes(bases)
    with pkg.Context([0]) as cp:
        cp.set_bases_precomputed(group, bases, n, c)
        for kind in ("uniform", "mixed", "ones"):
            sc = ([rnd.randrange(o.R_ORDER) for _ in range(n)] if kind == "uniform" else _mixed_scalars(o, rnd, n) if kind == "mixed" else [1] * n)
            canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
            want = _canon(co, group, co.msm(group, bases, canon, n, 0, 4))
            got = cp.msm(group, None, canon, n, pkg.SCALAR_CANONICAL)
            assert _canon(co, group, got) == want, (group, n, c, kind)
            prof = cp.profile()
            if c:
                assert prof["window_bits"] == c
            assert _canon(co, group, cp.msm(group, None, co.fr_to_mont(canon), n, pkg.SCALAR_MONTGOMERY)) == want
            m = max(1, n // 3)
            assert _canon(co, group, cp.msm(group, None, canon[:32 * m], m, pkg.SCALAR_CANONICAL)) == \
                _canon(co, group, co.msm(group, bases, canon, m, 0, 4))
        # a call that brings its own bases ignores the tables; plain set_bases afterwards replaces them
        sc = b"".join(o.fr_to_canon_bytes(rnd.randrange(o.R_ORDER)) for _ in range(n))
        want = _canon(co, group, co.msm(group, bases, sc, n, 0, 4))
        assert _canon(co, group, cp.msm(group, bases, sc, n, pkg.SCALAR_CANONICAL)) == want
        cp.set_bases(group, bases, n)
        assert _canon(co, group, cp.msm(group, None, sc, n, pkg.SCALAR_CANONICAL)) == want


def test_precomputed_tables_2_20(pkg, co):
    """the opt-in mode at BASELINE config #2's size against the closed form"""
    n = 1 << 20
    bases = co.gen_bases("g1", SEED_B + 220, n, 16)
    scalars = co.gen_scalars(SEED_S + 220, n)
    with pkg.Context([0]) as cp:
        cp.set_bases_precomputed("g1", bases, n, 0)
        got = cp.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)
        assert cp.profile()["window_bits"] >= 16
    assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 220, n)


# ------------------------------------------------------------------------------------------------ long inputs, failure injection
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_calls_longer_than_one_pass_are_split(pkg, co, group):
    """more points than one pass of the pipeline takes (2^26 per device in production; 1000 through the test build's hook):
    the parts are processed one after the other and added — host bases, resident bases, device scalars, precomputed tables"""
    import torch

    n = 4321
    aff = 96 if group == "g1" else 192
    bases = co.gen_bases(group, SEED_B + 230, n, 8)
    scalars = co.gen_scalars(SEED_S + 230, n)
    want = co.dlog_expected(group, scalars, SEED_B + 230, n)
    with pkg.Context([0], test_hooks=True) as c:
        c.test_set_max_part(1000)
        assert _canon(co, group, c.msm(group, bases, scalars, n, pkg.SCALAR_CANONICAL)) == want
        c.set_bases(group, bases, n)
        assert _canon(co, group, c.msm(group, None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        assert _canon(co, group, c.msm_device(group, d.data_ptr(), n, pkg.SCALAR_CANONICAL)) == want
        c.set_bases_precomputed(group, bases, n, 9)
        assert _canon(co, group, c.msm(group, None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        m = 2500
        assert _canon(co, group, c.msm(group, None, scalars[:32 * m], m, pkg.SCALAR_CANONICAL)) == co.dlog_expected(group, scalars[:32 * m], SEED_B + 230, m)
    with pkg.Context([0, 0, 0], test_hooks=True) as c3:   # three device slots, each splitting its shard
        c3.test_set_max_part(700)
        c3.set_bases(group, bases, n)
        assert _canon(co, group, c3.msm(group, None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        assert _canon(co, group, c3.msm(group, bases, scalars, n, pkg.SCALAR_CANONICAL)) == want
        c3.set_bases_precomputed(group, bases, n, 0)     # every slot builds the tables of its own shard
        assert _canon(co, group, c3.msm(group, None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        m = 3000                                         # a prefix that ends inside the last slot's shard
        assert _canon(co, group, c3.msm(group, None, scalars[:32 * m], m, pkg.SCALAR_CANONICAL)) == co.dlog_expected(group, scalars[:32 * m], SEED_B + 230, m)
        got = c3.msm_batch(group, [scalars, scalars], n, pkg.SCALAR_CANONICAL)
        assert [_canon(co, group, x) for x in got] == [want, want]


def test_allocation_failure_is_an_error_code_not_an_abort(pkg, co):
    """a failing device allocation — on the calling thread or inside a per-device worker of a multi-device context — comes
    back as MI_E_NOMEM across the C ABI (nothing throws or terminates), and the context works afterwards"""
    n = 3000
    bases = co.gen_bases("g1", SEED_B + 240, n, 8)
    scalars = co.gen_scalars(SEED_S + 240, n)
    want = co.dlog_expected("g1", scalars, SEED_B + 240, n)
    for ids in ([0], [0, 0]):
        with pkg.Context(ids, test_hooks=True) as c:
            for k in (1, 2, 5):
                c.test_fail_allocs(k)
                with pytest.raises(pkg.MsmError) as e:
                    c.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
                assert e.value.code == -4, e.value        # MI_E_NOMEM
                assert "out of memory" in str(e.value)
            c.test_fail_allocs(0)
            assert _canon(co, "g1", c.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)) == want
            c.test_fail_allocs(1)
            with pytest.raises(pkg.MsmError) as e:
                c.set_bases("g1", bases, n)
            assert e.value.code == -4
            c.test_fail_allocs(0)
            c.set_bases("g1", bases, n)
            assert _canon(co, "g1", c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)) == want
            c.test_fail_allocs(1)
            with pytest.raises(pkg.MsmError) as e:
                c.multi_pairing(bases[:96 * 40], co.gen_bases("g2", 3, 40, 2))
            assert e.value.code == -4
            c.test_fail_allocs(0)
    # no memory for the base-set cache's copy of the vector is NOT a failed call — it runs uncached, as it would with the cache off
    n2 = 6000
    b2 = co.gen_bases("g1", SEED_B + 241, n2, 8)
    s2 = co.gen_scalars(SEED_S + 241, n2)
    w2 = co.dlog_expected("g1", s2, SEED_B + 241, n2)
    with pkg.Context([0], test_hooks=True) as c:
        c.set_base_cache(2)
        c.msm("g1", b2[:96 * 5000], s2[:32 * 5000], 5000, pkg.SCALAR_CANONICAL)   # sizes the lane's scratch (uncached size class: its own entry)
        c.test_fail_allocs(1)                                                      # the next growing allocation: the new entry's point buffer
        assert _canon(co, "g1", c.msm("g1", b2, s2, n2, pkg.SCALAR_CANONICAL)) == w2
        st = c.base_cache_stats()
        assert st["entries"] == 1 and st["misses"] == 2                            # the 6000-point vector was not cached ...
        c.test_fail_allocs(0)
        assert _canon(co, "g1", c.msm("g1", b2, s2, n2, pkg.SCALAR_CANONICAL)) == w2
        assert c.base_cache_stats()["entries"] == 2                                # ... and is on the next call


def test_multi_device_context_device_resident_scalars(pkg, co):
    """a context over several device slots reads its shard of ONE device-resident scalar vector per slot"""
    import torch

    n = 10001
    bases = co.gen_bases("g1", SEED_B + 250, n, 8)
    scalars = co.gen_scalars(SEED_S + 250, n)
    d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with pkg.Context([0, 0, 0]) as c3:
        c3.set_bases("g1", bases, n)
        got = c3.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL)
        m = 7000
        got_m = c3.msm_device("g1", d.data_ptr(), m, pkg.SCALAR_CANONICAL)
    assert _canon(co, "g1", got) == co.dlog_expected("g1", scalars, SEED_B + 250, n)
    assert _canon(co, "g1", got_m) == co.dlog_expected("g1", scalars[:32 * m], SEED_B + 250, m)


def test_multi_device_context_without_peer_access_stages_the_shards(pkg, co):
    """The 'no peer access' branch (a node or pair without it must not dereference another device's pointer): forced through the
    test hook on one GPU, every slot but the owner's then copies its shard with hipMemcpyPeerAsync instead of reading in place."""
    import torch

    n = 9001
    bases = co.gen_bases("g1", SEED_B + 251, n, 8)
    scalars = co.gen_scalars(SEED_S + 251, n)
    d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    want = co.dlog_expected("g1", scalars, SEED_B + 251, n)
    with pkg.Context([0, 0, 0], test_hooks=True) as c3:
        c3.set_bases("g1", bases, n)
        assert _canon(co, "g1", c3.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL)) == want
        c3.test_set_no_peer(True)
        assert _canon(co, "g1", c3.msm_device("g1", d.data_ptr(), n, pkg.SCALAR_CANONICAL)) == want
        m = 4000   # a prefix that leaves the last slot without work
        assert _canon(co, "g1", c3.msm_device("g1", d.data_ptr(), m, pkg.SCALAR_CANONICAL)) == co.dlog_expected("g1", scalars[:32 * m], SEED_B + 251, m)


def test_host_pointer_as_device_scalars_is_an_error_code(ctx, pkg, co):
    """mi_msm_g1_device with memory the HIP runtime does not know (plain host memory here; memory of a second HIP runtime in the
    process looks the same) is MI_E_INVALID with a message, not a GPU fault"""
    import ctypes as C

    n = 64
    ctx.set_bases("g1", co.gen_bases("g1", 5, n, 1), n)
    host = C.create_string_buffer(co.gen_scalars(6, n), 32 * n)
    with pytest.raises(pkg.MsmError) as e:
        ctx.msm_device("g1", C.addressof(host), n, pkg.SCALAR_CANONICAL)
    assert e.value.code == -1 and "HIP runtime" in str(e.value)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_device_windows_and_fold_equal_msm(pkg, co, group):
    """mi_msm_g{1,2}_device_windows + mi_g{1,2}_fold_windows (the exchange step of the one-process-per-GPU deployment, BASELINE
    config #3): three 'ranks' (contexts) each own a contiguous shard of the base set and leave their per-window sums in DEVICE
    memory; the gathered windows folded on the host equal the MSM over the whole set (oracle) and mi_msm_device of each shard
    equals the one-rank fold of its own windows.  Ragged shard sizes, so the window size is pinned on every rank."""
    import torch

    aff, size = (96, 144) if group == "g1" else (192, 288)
    n = 20000 if group == "g1" else 6000
    cuts = [0, n // 3, n // 3 + n // 4, n]
    seed = 260 if group == "g1" else 261
    bases = co.gen_bases(group, SEED_B + seed, n, 8)
    scalars = co.gen_scalars(SEED_S + seed, n)
    d = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
    wins = [torch.zeros(pkg.MAX_WINDOWS * size, dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    infos, gathered = [], b""
    for r in range(3):
        lo, hi = cuts[r], cuts[r + 1]
        with pkg.Context([0]) as c:
            c.set_window_bits(13)
            c.set_bases(group, bases[aff * lo:aff * hi], hi - lo)
            info = c.msm_device_windows(group, d.data_ptr() + 32 * lo, hi - lo, pkg.SCALAR_CANONICAL, wins[r].data_ptr())
            infos.append(info)
            w = wins[r].cpu().numpy().tobytes()
            one = pkg.fold_windows(group, w, 1, pkg.MAX_WINDOWS, *info)
            assert _canon(co, group, one) == _canon(co, group, c.msm_device(group, d.data_ptr() + 32 * lo, hi - lo, pkg.SCALAR_CANONICAL))
            gathered += w[:info[1] * size]
            # an empty call reports no windows
            assert c.msm_device_windows(group, d.data_ptr(), 0, pkg.SCALAR_CANONICAL, wins[r].data_ptr()) == (0, 0)
    assert infos[0] == infos[1] == infos[2] == (13, 20)
    total = pkg.fold_windows(group, gathered, 3, infos[0][1], *infos[0])
    assert _canon(co, group, total) == co.dlog_expected(group, scalars, SEED_B + seed, n)
    # precomputed tables: every window feeds one bucket set, so there is ONE window sum and the fold is the sum over the ranks
    if group == "g1":
        with pkg.Context([0]) as c:
            c.set_bases_precomputed(group, bases[:aff * 3000], 3000, 11)
            info = c.msm_device_windows(group, d.data_ptr(), 3000, pkg.SCALAR_CANONICAL, wins[0].data_ptr())
            assert info == (11, 1)
            got = pkg.fold_windows(group, wins[0].cpu().numpy().tobytes(), 1, pkg.MAX_WINDOWS, *info)
            assert _canon(co, group, got) == co.dlog_expected(group, scalars[:32 * 3000], SEED_B + seed, 3000)
        # a multi-device context is refused (one context per rank)
        with pkg.Context([0, 0]) as c2:
            c2.set_bases(group, bases[:aff * 100], 100)
            with pytest.raises(pkg.MsmError) as e:
                c2.msm_device_windows(group, d.data_ptr(), 100, pkg.SCALAR_CANONICAL, wins[0].data_ptr())
            assert e.value.code == -1


@pytest.mark.parametrize("n", [(1 << 18) + 12345, (1 << 17) - 1, 65537])
def test_host_slices_cross_pcie_in_chunks(ctx, co, pkg, n):
    """The trait's call shape — host slices (/root/reference/src/g1.rs:604,623; uploaded per call at src/gpu.rs:149-150): above 2^16
    points the bases and the scalars are copied in several chunks on a copy stream, each ingested / counted as it lands.  Odd sizes,
    so the chunks are ragged; both scalar formats; host bases and resident bases."""
    bases = co.gen_bases("g1", SEED_B + 270, n, 8)
    scalars = co.gen_scalars(SEED_S + 270, n)
    want = co.dlog_expected("g1", scalars, SEED_B + 270, n)
    assert _canon(co, "g1", ctx.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)) == want
    ctx.set_bases("g1", bases, n)
    assert _canon(co, "g1", ctx.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)) == want
    mont = co.fr_to_mont(scalars)
    assert _canon(co, "g1", ctx.msm("g1", None, mont, n, pkg.SCALAR_MONTGOMERY)) == want
    m = n - 4097   # a prefix of the resident set
    assert _canon(co, "g1", ctx.msm("g1", None, scalars[:32 * m], m, pkg.SCALAR_CANONICAL)) == co.dlog_expected("g1", scalars[:32 * m], SEED_B + 270, m)


def test_host_slices_in_chunks_g2_and_large_window(pkg, co):
    """the same for G2 and for a window size of the three-level sort (c = 18 forced), whose count pass is the other chunked kernel"""
    n = (1 << 17) + 77
    for group, c in (("g2", 0), ("g1", 18)):
        bases = co.gen_bases(group, SEED_B + 271, n, 8)
        scalars = co.gen_scalars(SEED_S + 271, n)
        with pkg.Context([0]) as cx:
            cx.set_window_bits(c)
            assert _canon(co, group, cx.msm(group, bases, scalars, n, pkg.SCALAR_CANONICAL)) == co.dlog_expected(group, scalars, SEED_B + 271, n)


def test_plain_set_bases_gives_back_the_table_allocation(pkg, co):
    """mi_msm_g1_set_bases after mi_msm_g1_set_bases_precomputed must not keep the W x table allocation for the life of the context
    (DevBuf::ensure_fit): the device's free memory comes back."""
    import torch

    n = 1 << 18
    bases = co.gen_bases("g1", SEED_B + 280, n, 8)
    with pkg.Context([0]) as c:
        torch.cuda.synchronize()
        c.set_bases_precomputed("g1", bases, n, 13)           # 20 tables of 32 MiB
        free_tables = torch.cuda.mem_get_info()[0]
        c.set_bases("g1", bases[:96 * 1000], 1000)
        free_plain = torch.cuda.mem_get_info()[0]
        assert free_plain - free_tables > 400 << 20, (free_tables, free_plain)
        sc = co.gen_scalars(SEED_S + 280, 1000)
        assert _canon(co, "g1", c.msm("g1", None, sc, 1000, pkg.SCALAR_CANONICAL)) == co.dlog_expected("g1", sc, SEED_B + 280, 1000)


@pytest.mark.parametrize("group,n,c,weights", [
    ("g1", 70000, 0, [1, 3]), ("g1", 70000, 0, [1, 1]), ("g1", 70000, 0, [2, 1, 1]), ("g1", 70000, 0, [1, 2, 2, 1]), ("g1", 70000, 0, [1, 100]),
    ("g1", 9000, 11, [1, 3]), ("g1", 33000, 15, [3, 1]), ("g1", 150000, 18, [1, 2]), ("g1", 150000, 20, [1, 1, 1]),
    ("g2", 20000, 0, [1, 3]), ("g2", 20000, 13, [1, 1, 1, 1])])
def test_window_group_pipeline_is_bit_exact(pkg, co, group, n, c, weights):
    """Round 6: a call processed as window groups — per-group scratch, the sort of group g + 1 under the accumulate
    kernel of group g on a second stream, reductions together at the end — returns the point of the one-group call and of the oracle, for
    every weighting (a group of one window included), window size (two- and three-level sort), resident or host bases, canonical or
    Montgomery scalars, host or device scalars, a validated set (sign fold: one window fewer at c = 15), and scalars that overfill buckets
    (merge launches inside the pipeline).  mi_profile.window_groups reports the groups."""
    import torch

    bases = co.gen_bases(group, 0x6A0 + n, n, 8)
    sc = co.gen_scalars(0x6A1 + n, n)
    want = co.to_affine(group, co.msm(group, bases, sc, n, 0, 8))
    skew = bytearray(sc)                       # a quarter of the scalars share four values: buckets beyond T entries, split and merged
    for i in range(0, n, 4):
        skew[32 * i:32 * i + 32] = sc[32 * (i % 16):32 * (i % 16) + 32]
    skew = bytes(skew)
    want_skew = co.to_affine(group, co.msm(group, bases, skew, n, 0, 8))
    with pkg.Context([0]) as cx:
        cx.set_window_bits(c)
        cx.set_bases(group, bases, n)
        for w in ([1], weights):
            cx.set_pipeline(w)
            assert _canon(co, group, cx.msm(group, None, sc, n, 0)) == want                    # resident bases, host scalars (chunked H2D)
            assert cx.profile()["window_groups"] == (1 if w == [1] else min(len(w), cx.profile()["num_windows"]))
            assert _canon(co, group, cx.msm(group, bases, co.fr_to_mont(sc), n, 1)) == want     # host bases, Montgomery scalars
            d = torch.frombuffer(bytearray(skew), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            assert _canon(co, group, cx.msm_device(group, d.data_ptr(), n, 0)) == want_skew     # device scalars, overfull buckets
            assert cx.profile()["max_items_per_bucket"] > 1
        assert cx.validate_bases(group) == 0
        assert _canon(co, group, cx.msm(group, None, sc, n, 0)) == want                        # pipelined + sign fold
        cx.set_pipeline(None)
        cx.set_window_bits(0)


def test_window_group_pipeline_at_2_20_and_device_windows(pkg, co):
    """The built-in choice is one group (the pipelined form is opt-in: its gain depends on the runtime's queue assignment); with
    two groups switched on, 2^20 points and the exchange path (device_windows: window sums left in device memory, per group at its window
    offset) give the same point."""
    import torch

    n = 1 << 20
    bases = co.gen_bases("g1", 0x6B0, n, 16)
    sc = co.gen_scalars(0x6B1, n)
    want = co.dlog_expected("g1", sc, 0x6B0, n)
    d = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    out = torch.zeros(37 * 144, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pkg.Context([0]) as cx:
        cx.set_bases("g1", bases, n)
        assert _canon(co, "g1", cx.msm_device("g1", d.data_ptr(), n, 0)) == want and cx.profile()["window_groups"] == 1
        cx.set_pipeline([1, 3])
        assert _canon(co, "g1", cx.msm_device("g1", d.data_ptr(), n, 0)) == want and cx.profile()["window_groups"] == 2
        assert 1.0 < cx.profile()["accumulate_clock_ghz"] < 3.0          # the in-kernel clock of the accumulate launches
        info = cx.msm_device_windows("g1", d.data_ptr(), n, 0, out.data_ptr())
        torch.cuda.synchronize()
        assert cx.profile()["window_groups"] == 2
        folded = pkg.fold_windows("g1", out.cpu().numpy().tobytes(), 1, pkg.MAX_WINDOWS, *info)
        assert _canon(co, "g1", folded) == want
        cx.set_pipeline([1])
        assert _canon(co, "g1", cx.msm_device("g1", d.data_ptr(), n, 0)) == want and cx.profile()["window_groups"] == 1


def _scaled_jacobian(o, F, group, raw, n, seed):
    """affine points (reference bytes) -> Jacobian with random non-trivial Z, every 13th one infinity with garbage X, Y"""
    rnd = random.Random(seed)
    aff = 96 if group == "g1" else 192
    jac, want = [], []
    for i in range(n):
        pt = o.affine_from_bytes(F, raw[aff * i:aff * (i + 1)])
        if i % 13 == 5:
            jac.append(o._felt_bytes(F, F.mul(pt[0], pt[1])) + o._felt_bytes(F, pt[1]) + o._felt_bytes(F, F.zero))
            want.append(bytes(aff))
            continue
        lam = rnd.randrange(1, o.P) if group == "g1" else (rnd.randrange(1, o.P), rnd.randrange(o.P))
        l2 = F.mul(lam, lam)
        jac.append(o._felt_bytes(F, F.mul(pt[0], l2)) + o._felt_bytes(F, F.mul(pt[1], F.mul(l2, lam))) + o._felt_bytes(F, lam))
        want.append(raw[aff * i:aff * (i + 1)])
    return b"".join(jac), b"".join(want)


@pytest.mark.parametrize("group,n", [("g1", 70000), ("g1", 1024 * 33 + 5), ("g2", 40000)])
def test_rows_f_chunked_and_device_forms(pkg, co, o, group, n):
    """Round 6 (VERDICT r05 #2 / missing #3): the host-pointer forms of normalize / deserialize / serialize / check cross PCIe in chunks (sizes
    that are not multiples of the chunk or of the product tree's fan-out), the *_device forms take and leave device memory, and both agree
    with the C oracle.  Decoded points feed mi_msm_set_bases_device without touching the host; a second call reuses the context's buffers."""
    import numpy as np
    import torch

    F = o.F1 if group == "g1" else o.F2
    aff, jb, unit = (96, 144, 48) if group == "g1" else (192, 288, 96)
    raw = co.gen_bases(group, 0x7F0 + n, n, 8)
    small = 1500                                            # the big-int part of the fixture stays small; the rest comes from the C oracle
    jac_s, want_s = _scaled_jacobian(o, F, group, raw, small, n)
    # Jacobian inputs for all n: the first `small` with random Z, the others with Z = 1 (the affine point itself)
    one = o._felt_bytes(F, F.one)
    jac = bytearray(jac_s)
    for i in range(small, n):
        jac += raw[aff * i:aff * (i + 1)] + one
    jac = bytes(jac)
    want = want_s + raw[aff * small:]
    with pkg.Context([0]) as c:
        for rep in range(2):
            assert c.normalize_batch(group, jac) == want
        d_in = torch.frombuffer(bytearray(jac), dtype=torch.uint8).cuda()
        d_out = torch.zeros(n * aff, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.normalize_batch_device(group, d_in.data_ptr(), n, d_out.data_ptr())
        assert d_out.cpu().numpy().tobytes() == want
        # serialize -> deserialize round trip through host memory, then the device form on the same encodings
        enc = c.serialize_batch(group, want, True)
        pts, st = c.deserialize_batch(group, enc, True, True)
        assert pts == want and st == bytes(n)
        wpts, wst = (co.g1_deserialize_batch if group == "g1" else co.g2_deserialize_batch)(enc[:unit * 3000], True, True, 0, 8)
        assert pts[:aff * 3000] == wpts and st[:3000] == wst
        bad = bytearray(enc)
        bad[unit * 777] ^= 0x40                              # infinity flag on a finite encoding: malformed
        bad[unit * (n - 1) + unit - 1] ^= 1                  # another x: almost surely not on the curve
        pts2, st2 = c.deserialize_batch(group, bytes(bad), True, True)
        assert st2[777] == 1 and st2[n - 1] != 0 and sum(1 for x in st2 if x) == 2
        d_enc = torch.frombuffer(bytearray(bad), dtype=torch.uint8).cuda()
        d_pts = torch.zeros(n * aff, dtype=torch.uint8, device="cuda")
        d_st = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.deserialize_batch_device(group, d_enc.data_ptr(), n, True, True, d_pts.data_ptr(), d_st.data_ptr())
        assert d_pts.cpu().numpy().tobytes() == pts2 and d_st.cpu().numpy().tobytes() == st2
        # Valid::batch_check, host and device
        chk = c.check_batch(group, pts2)
        assert chk == bytes(n)                               # rejected points were zeroed = infinity = valid
        c.check_batch_device(group, d_pts.data_ptr(), n, d_st.data_ptr())
        assert d_st.cpu().numpy().tobytes() == chk
        # decode -> resident set without the host: same MSM as set_bases over the decoded bytes
        sc = co.gen_scalars(0x7F1, n)
        c.set_bases_device(group, d_pts.data_ptr(), n)
        got = c.msm(group, None, sc, n, 0)
        assert _canon(co, group, got) == co.to_affine(group, co.msm(group, pts2, sc, n, 0, 8))
        # a device pointer that is not one
        with pytest.raises(pkg.MsmError) as e:
            c.normalize_batch_device(group, 4096, n, d_out.data_ptr())
        assert e.value.code == -1


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_set_bases_from_jacobian_and_from_compressed(pkg, co, o, group):
    """normalize -> msm and decode -> msm as ONE upload each (src/g1.rs:597-599 feeding 604; SRS loading): the resident set equals set_bases over
    the oracle's affine points; a validating load marks the set validated (sign fold: 17 windows instead of 18 at c = 15); a single
    rejected encoding installs nothing and leaves the previous set in place; two 'devices' shard the work."""
    F = o.F1 if group == "g1" else o.F2
    aff, unit = (96, 48) if group == "g1" else (192, 96)
    n = 9000 if group == "g1" else 5000
    raw = co.gen_bases(group, 0x801, n, 8)
    sc = co.gen_scalars(0x802, n)
    want = co.to_affine(group, co.msm(group, raw, sc, n, 0, 8))
    jac_s, want_s = _scaled_jacobian(o, F, group, raw, 600, 5)
    one = o._felt_bytes(F, F.one)
    jac = jac_s + b"".join(raw[aff * i:aff * (i + 1)] + one for i in range(600, n))
    affine = want_s + raw[aff * 600:]
    want_j = co.to_affine(group, co.msm(group, affine, sc, n, 0, 8))     # some points became infinity
    for devs in ([0], [0, 0]):
        with pkg.Context(devs) as c:
            c.set_bases_from_jacobian(group, jac, n)
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want_j
            enc = c.serialize_batch(group, raw, True)
            c.set_window_bits(15)
            assert c.set_bases_from_compressed(group, enc, n, True, True) == 0
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want and c.profile()["num_windows"] == 17   # validated: sign fold
            assert c.set_bases_from_compressed(group, enc, n, True, False) == 0
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want and c.profile()["num_windows"] == 18   # not validated
            unc = c.serialize_batch(group, raw, False)
            assert c.set_bases_from_compressed(group, unc, n, False, True) == 0
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want
            bad = bytearray(enc)
            bad[unit * (n // 2) + 3] ^= 0x55
            bad[unit * 7] ^= 0x40
            rej = c.set_bases_from_compressed(group, bytes(bad), n, True, True)
            assert rej in (1, 2) and rej >= 1
            assert _canon(co, group, c.msm(group, None, sc, n, 0)) == want                                      # the previous set is still there
            c.set_window_bits(0)


def test_abort_check_is_the_reference_drivers_maybe_abort(pkg, co):
    """VERDICT r05 missing #7: the cooperative abort hook of the reference's driver (src/gpu.rs:55-58,133-137).  The check is asked at the start of a
    call and between the passes of a long one (1000-point passes through the test build's hook); True ends the call with MI_E_ABORTED (-8),
    the context stays usable, and removing the check restores normal service."""
    n = 4321
    bases = co.gen_bases("g1", SEED_B + 231, n, 8)
    scalars = co.gen_scalars(SEED_S + 231, n)
    want = co.dlog_expected("g1", scalars, SEED_B + 231, n)
    with pkg.Context([0], test_hooks=True) as c:
        c.test_set_max_part(1000)
        c.set_bases("g1", bases, n)
        asked = []
        c.set_abort_check(lambda: (asked.append(1), False)[1])
        assert _canon(co, "g1", c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)) == want
        assert len(asked) == 5                               # at the start + between the five passes
        asked.clear()
        c.set_abort_check(lambda: (asked.append(1), len(asked) >= 3)[1])   # stop before the third pass
        with pytest.raises(pkg.MsmError) as e:
            c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)
        assert e.value.code == -8 and len(asked) == 3
        c.set_abort_check(lambda: True)
        with pytest.raises(pkg.MsmError) as e:
            c.msm("g1", bases, scalars, n, pkg.SCALAR_CANONICAL)
        assert e.value.code == -8
        c.set_abort_check(None)
        assert _canon(co, "g1", c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL)) == want
