// Machine check of the lazy-reduction bookkeeping of the MSM formulas in ark-blst_amd/csrc/ec.cuh ON THE REAL CODE:
// xyzz_madd (the accumulate hot loop, including its carry-free "lazy" linear operations), xyzz_to_proj and the
// complete proj_add are instantiated with a field class that carries, instead of a value, a VALUE bound (multiples of
// p) and a LIMB bound (largest 32-bit limb), and enforces the contracts of fp28.cuh at every call:
//   product (one reduction)   sum a_i b_i < 2^392 / p (2520 p^2);  14 * sum la_i lb_i < 2^64 - 2^60;   output exact, < 2p
//   squaring                  as product, and limb < 2^31 (the doubled operand)
//   fp_add / fp_sub<K>        limb sums < 2^32; subtrahend limbs <= 2^28 + 64 and value <= (K-1) p;   output N-form
//   lazy add / sub            the same without the carry pass: limbs add up
//   fp_sub8_lazy_wide         subtrahend limbs <= 3 * 2^28 + 64, value <= 7p
// The formulas are iterated to a fixed point from the kernel's initial state.  Complements tools/bounds_check.py (a
// transcript of the formulas); test helper only — not part of the shipped library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../ark-blst_amd/csrc/ec.cuh"

static const double LIMIT = 2520.0, TWO28 = 268435456.0, TWO32 = 4294967296.0;
static const double EXACT = TWO28 - 1, NFORM = TWO28 + 15, SPREAD_LO = TWO28 + 64, SPREAD_HI = 2 * TWO28 + 68;
static const double COLMAX = 18446744073709551616.0 - 1152921504606846976.0;   // 2^64 - 2^60
static double g_max_prod = 0, g_max_col = 0, g_max_limb = 0;

static void fail(const char* what, double v, double lim) {
    fprintf(stderr, "BOUND VIOLATION: %s: %.6g > %.6g\n", what, v, lim);
    exit(1);
}
struct BFp {
    double v, l;   // value bound in units of p, limb bound
};
static BFp reduce(double prod, double col) {
    if (prod >= LIMIT) fail("product value", prod, LIMIT);
    if (14.0 * col >= COLMAX) fail("column sum", 14.0 * col, COLMAX);
    g_max_prod = std::max(g_max_prod, prod);
    g_max_col = std::max(g_max_col, 14.0 * col);
    return {2.0, EXACT};
}
static double limb(double l) {
    if (l >= TWO32) fail("limb overflow", l, TWO32);
    g_max_limb = std::max(g_max_limb, l);
    return l;
}
static void need_subtrahend(const BFp& b, int K, double lo) {
    if (b.v > K - 1) fail("subtrahend value", b.v, K - 1);
    if (b.l > lo) fail("subtrahend limb", b.l, lo);
}

struct BoundFp {
    using E = BFp;
    static constexpr int B3 = 12;
    static E zero() { return {0, 0}; }
    static E one() { return {1, EXACT}; }
    static E mul(const E& a, const E& b) { return reduce(a.v * b.v, a.l * b.l); }
    static E sqr(const E& a) {
        if (a.l >= TWO32 / 2) fail("squaring operand limb", a.l, TWO32 / 2);
        return reduce(a.v * a.v, a.l * a.l);
    }
    static E mul2add(const E& a, const E& b, const E& c, const E& d) { return reduce(a.v * b.v + c.v * d.v, a.l * b.l + c.l * d.l); }
    static E add(const E& a, const E& b) { limb(a.l + b.l); return {a.v + b.v, NFORM}; }
    template <int K> static E sub(const E& a, const E& b) { need_subtrahend(b, K, SPREAD_LO); limb(a.l + SPREAD_HI); return {a.v + K, NFORM}; }
    template <int K> static E neg(const E& a) { return sub<K>(zero(), a); }
    static E mul3(const E& a) { limb(3 * a.l); return {3 * a.v, NFORM}; }
    template <int K> static E mul_small(const E& a) { limb(K * a.l); return {K * a.v, NFORM}; }
    static E reduce_small(const E& a) {                                                   // fp_reduce_small: N-form in, value < 127p; [p, 2.01p) out
        if (a.v > 126) fail("reduce_small operand value", a.v, 126);
        if (a.l > NFORM) fail("reduce_small operand limb", a.l, NFORM);
        return {2.01, NFORM};
    }
    static E mul_b3(const E& a) { limb(12 * a.l); return {12 * a.v, NFORM}; }
    static E mul_b3_red(const E& a) { return mul(a, E{1, EXACT}); }                       // times the constant 12 (< p) as a field product
    static E sqr_sub12sqr(const E& s, const E& e) {                                       // one fused reduction (FpOpsInlinePS)
        E w = neg<4>(e);
        limb(12 * w.l);
        return reduce(s.v * s.v + e.v * 12 * w.v, s.l * s.l + e.l * NFORM);
    }
    static bool is_zero_2p(const E& a) { if (a.v > 2) fail("is_zero_2p operand", a.v, 2); return false; }
    static E select(bool, const E& a, const E& b) { return {std::max(a.v, b.v), std::max(a.l, b.l)}; }
    static bool limbs_all_zero(const E&) { return false; }
    // the hot loop's carry-free operations (FpOpsInline)
    static E add_l(const E& a, const E& b) { return {a.v + b.v, limb(a.l + b.l)}; }
    template <int K> static E sub_l(const E& a, const E& b) { need_subtrahend(b, K, SPREAD_LO); return {a.v + K, limb(a.l + SPREAD_HI)}; }
    template <int K> static E neg_l(const E& a) { return sub_l<K>(zero(), a); }
    static E sub8_wide(const E& a, const E& b) { need_subtrahend(b, 8, 3 * TWO28 + 64); return {a.v + 8, limb(a.l + 4 * TWO28 + 68)}; }
    static E norm(const E& a) { limb(a.l); return {a.v, NFORM}; }
};

static BFp mx(const BFp& a, const BFp& b) { return {std::max(a.v, b.v), std::max(a.l, b.l)}; }

// G2: ec::Fp2OpsT builds an Fp2 product from fused two- (four-) product reductions of Fp and normalises after every
// linear operation, so limbs are always N-form and only VALUE bounds matter.  This mirrors its formulas component-wise:
//   c0 = a0 b0 + a1 (32p - b1)     needs b1 <= 31p          c1 = a0 b1 + a1 b0
// INLINE = the hot loop's variant (mul2add is ONE reduction of four products per component).
struct B2 {
    double c0, c1;
};
static double red2(double prod) {
    if (prod >= LIMIT) fail("Fp2 product value", prod, LIMIT);
    g_max_prod = std::max(g_max_prod, prod);
    return 2.0;
}
static double val(double v) {
    if (v >= LIMIT) fail("value does not fit 2^392", v, LIMIT);
    return v;
}
template <bool INLINE>
struct BoundFp2 {
    using E = B2;
    static E zero() { return {0, 0}; }
    static E one() { return {1, 0}; }
    static double fsub(int K, double a, double b) {
        if (b > K - 1) fail("Fp2 subtrahend value", b, K - 1);
        return val(a + K);
    }
    static E mul(const E& a, const E& b) {
        if (b.c1 > 31) fail("Fp2 mul: b.c1", b.c1, 31);
        return {red2(a.c0 * b.c0 + a.c1 * 32.0), red2(a.c0 * b.c1 + a.c1 * b.c0)};
    }
    static E sqr(const E& a) { return {red2((a.c0 + a.c1) * fsub(32, a.c0, a.c1)), red2(2 * a.c0 * a.c1)}; }
    static E mul2add(const E& a, const E& b, const E& c, const E& d) {
        if (b.c1 > 31 || d.c1 > 31) fail("Fp2 mul2add: c1", std::max(b.c1, d.c1), 31);
        if (INLINE)
            return {red2(a.c0 * b.c0 + a.c1 * 32.0 + c.c0 * d.c0 + c.c1 * 32.0), red2(a.c0 * b.c1 + a.c1 * b.c0 + c.c0 * d.c1 + c.c1 * d.c0)};
        return {val(red2(a.c0 * b.c0 + a.c1 * 32.0) + red2(c.c0 * d.c0 + c.c1 * 32.0)),
                val(red2(a.c0 * b.c1 + a.c1 * b.c0) + red2(c.c0 * d.c1 + c.c1 * d.c0))};
    }
    static E add(const E& a, const E& b) { return {val(a.c0 + b.c0), val(a.c1 + b.c1)}; }
    template <int K> static E sub(const E& a, const E& b) { return {fsub(K, a.c0, b.c0), fsub(K, a.c1, b.c1)}; }
    template <int K> static E neg(const E& a) { return sub<K>(zero(), a); }
    static E add_l(const E& a, const E& b) { return add(a, b); }
    template <int K> static E sub_l(const E& a, const E& b) { return sub<K>(a, b); }
    template <int K> static E neg_l(const E& a) { return neg<K>(a); }
    static E sub8_wide(const E& a, const E& b) { return sub<8>(a, b); }
    static E norm(const E& a) { return a; }
    static E mul3(const E& a) { return {val(3 * a.c0), val(3 * a.c1)}; }
    static E mul_b3(const E& a) { return mul(a, E{1, 1}); }   // by the constant (12, 12), each component < p
    static E mul_b3_red(const E& a) { return mul_b3(a); }
    static E sqr_sub12sqr(const E& s, const E& e) {            // ec::Fp2OpsT::sqr_sub12sqr: one fused two-product reduction per component
        double d = fsub(32, s.c0, s.c1), w0 = val(12 * fsub(4, e.c1, e.c0)), w1 = val(12 * fsub(4, 0, e.c1));
        return {red2((s.c0 + s.c1) * d + (e.c0 + e.c1) * w0), red2(2 * s.c0 * s.c1 + 2 * e.c0 * w1)};
    }
    static bool is_zero_2p(const E& a) { if (a.c0 > 2 || a.c1 > 2) fail("is_zero_2p operand", std::max(a.c0, a.c1), 2); return false; }
    static E select(bool, const E& a, const E& b) { return {std::max(a.c0, b.c0), std::max(a.c1, b.c1)}; }
    static bool limbs_all_zero(const E&) { return false; }
};
static B2 mx(const B2& a, const B2& b) { return {std::max(a.c0, b.c0), std::max(a.c1, b.c1)}; }
static bool same(const B2& a, const B2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }

template <class FA, class F>
static void check_g2(const char* tag) {
    B2 x2{2, 2}, y2 = mx(B2{2, 2}, FA::template neg_l<4>(B2{2, 2}));
    ec::Xyzz<FA> inv;
    inv.x = x2; inv.y = FA::norm(y2); inv.zz = FA::one(); inv.zzz = FA::one();
    for (int it = 0;; it++) {
        ec::Xyzz<FA> a = inv;
        ec::xyzz_madd<FA>(a, x2, y2);
        ec::Xyzz<FA> nx{mx(inv.x, a.x), mx(inv.y, a.y), mx(inv.zz, a.zz), mx(inv.zzz, a.zzz)};
        bool fix = same(nx.x, inv.x) && same(nx.y, inv.y) && same(nx.zz, inv.zz) && same(nx.zzz, inv.zzz);
        inv = nx;
        if (fix) break;
        if (it == 63) fail("G2 madd invariant did not converge", it, 63);
    }
    ec::Xyzz<F> as_f{inv.x, inv.y, inv.zz, inv.zzz};
    ec::Proj<F> pj = ec::xyzz_to_proj<F>(as_f);
    ec::Proj<F> q = ec::proj_from_affine<F>(x2, F::template neg<4>(B2{2, 2}));
    ec::Proj<F> pinv{mx(pj.x, q.x), mx(pj.y, q.y), mx(pj.z, q.z)};
    for (int it = 0;; it++) {
        ec::Proj<F> a = pinv;
        ec::proj_add<F>(a, pinv);
        ec::Proj<F> nx{mx(pinv.x, a.x), mx(pinv.y, a.y), mx(pinv.z, a.z)};
        bool fix = same(nx.x, pinv.x) && same(nx.y, pinv.y) && same(nx.z, pinv.z);
        pinv = nx;
        if (fix) break;
        if (it == 63) fail("G2 proj_add invariant did not converge", it, 63);
    }
    {   // ec::proj_dbl (round 4: the subgroup-test ladders of the point decoder alternate it with proj_add): closed under both
        ec::Proj<F> d = pinv;
        for (int it = 0; it < 4; it++) {
            ec::proj_dbl<F>(d);
            ec::Proj<F> t = d;
            ec::proj_add<F>(t, pinv);
            d = ec::Proj<F>{mx(d.x, t.x), mx(d.y, t.y), mx(d.z, t.z)};
        }
    }
    printf("%s: xyzz_madd X < %.0fp, Y < %.0fp; proj_add X < %.0fp, Y < %.0fp, Z < %.0fp\n", tag, std::max(inv.x.c0, inv.x.c1),
           std::max(inv.y.c0, inv.y.c1), std::max(pinv.x.c0, pinv.x.c1), std::max(pinv.y.c0, pinv.y.c1), std::max(pinv.z.c0, pinv.z.c1));
}

// One COMPONENT of the lane-pair Fp2 field (coop_fp2.cuh CoopF2; both lanes of a pair hold values of the same bounds):
//   mul   a s1 + a' s2 with s1 = b or b', s2 = 32p - b' (carry-free: limbs < 2^29 + 68) or b       needs b <= 31p
//   sqr   (a + a')(a + 32p - a')  |  (2 a') a                                                        needs a <= 31p
struct BoundCoop : BoundFp {
    static E mul(const E& a, const E& b) {
        if (b.v > 31) fail("coop mul: b", b.v, 31);
        return reduce(a.v * b.v + a.v * 32.0, a.l * b.l + a.l * SPREAD_HI);
    }
    static E sqr(const E& a) {
        const E u = add(a, a), v = sub<32>(a, a);
        return reduce(u.v * v.v, u.l * v.l);
    }
};

// The Jacobian subgroup ladders (ec.cuh jac_dbl / jac_add, round 6): the invariant under "double, then maybe add the base point or any state".
template <class F, bool REDUCE_Y, bool GENERAL_ADD>
static void check_jac(const char* tag) {
    using J = ec::JacE<BFp>;
    const J base{BFp{4, NFORM}, BFp{4, NFORM}, F::one()};     // affine inputs x, y < 4p
    J inv = base;
    auto mxj = [](const J& a, const J& b) { return J{mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z)}; };
    for (int it = 0;; it++) {
        J d = inv;
        ec::jac_dbl<F, REDUCE_Y>(d);
        J t = d;
        ec::jac_add<F, true>(t, base);
        J nx = mxj(mxj(inv, d), t);
        if (GENERAL_ADD) {
            J g = d;
            ec::jac_add<F, false>(g, inv);
            nx = mxj(nx, g);
        }
        const bool fix = nx.x.v == inv.x.v && nx.y.v == inv.y.v && nx.z.v == inv.z.v && nx.x.l == inv.x.l && nx.y.l == inv.y.l && nx.z.l == inv.z.l;
        inv = nx;
        if (fix) break;
        if (it == 63) fail("Jacobian ladder invariant did not converge", it, 63);
    }
    // the verdict (ec::g1_torsion_free / k_validate_g2_coop): X vs (beta x) Z^2 through fp_sub<16>, Y + y Z^3, both below the ~50p of fp_is_zero_any
    const BFp zz = F::sqr(inv.z);
    const BFp ex = F::template sub<16>(inv.x, F::mul(base.x, zz)), ey = F::add(inv.y, F::mul(base.y, F::mul(zz, inv.z)));
    if (ex.v > 50 || ey.v > 50) fail("verdict operand", std::max(ex.v, ey.v), 50);
    printf("%s: Jacobian ladder X < %.2fp, Y < %.2fp, Z < %.2fp\n", tag, inv.x.v, inv.y.v, inv.z.v);
}

int main() {
    using F = BoundFp;
    // --- accumulate hot loop (k_accumulate): x2 exact < 2p (ingest output), y2 = y or its lazy negation 4p - y
    BFp x2{2, EXACT}, y2 = mx(BFp{2, EXACT}, F::neg_l<4>(BFp{2, EXACT}));
    ec::Xyzz<F> acc;
    acc.x = x2; acc.y = F::norm(y2); acc.zz = F::one(); acc.zzz = F::one();      // the kernel's first-point initialisation
    ec::Xyzz<F> inv = acc;
    for (int it = 0; it < 64; it++) {
        ec::Xyzz<F> a = inv;
        ec::xyzz_madd<F>(a, x2, y2);
        ec::Xyzz<F> nx{mx(inv.x, a.x), mx(inv.y, a.y), mx(inv.zz, a.zz), mx(inv.zzz, a.zzz)};
        bool same = nx.x.v == inv.x.v && nx.y.v == inv.y.v && nx.x.l == inv.x.l && nx.y.l == inv.y.l && nx.zz.v == inv.zz.v && nx.zzz.v == inv.zzz.v;
        inv = nx;
        if (same) break;
        if (it == 63) fail("madd invariant did not converge", it, 63);
    }
    printf("xyzz_madd invariant: X < %.0fp (limb %.3g), Y < %.0fp, ZZ < %.0fp, ZZZ < %.0fp\n", inv.x.v, inv.x.l, inv.y.v, inv.zz.v, inv.zzz.v);
    // --- bucket leaves the hot loop: XYZZ -> projective, then the complete additions of merge / reduce / cold path
    ec::Proj<F> pj = ec::xyzz_to_proj<F>(inv);
    ec::Proj<F> q = ec::proj_from_affine<F>(x2, F::neg<4>(BFp{2, EXACT}));
    ec::Proj<F> pinv{mx(pj.x, q.x), mx(pj.y, q.y), mx(pj.z, q.z)};
    for (int it = 0; it < 64; it++) {
        ec::Proj<F> a = pinv;
        ec::proj_add<F>(a, pinv);
        ec::Proj<F> nx{mx(pinv.x, a.x), mx(pinv.y, a.y), mx(pinv.z, a.z)};
        bool same = nx.x.v == pinv.x.v && nx.y.v == pinv.y.v && nx.z.v == pinv.z.v;
        pinv = nx;
        if (same) break;
        if (it == 63) fail("proj_add invariant did not converge", it, 63);
    }
    printf("proj_add invariant: X < %.0fp, Y < %.0fp, Z < %.0fp\n", pinv.x.v, pinv.y.v, pinv.z.v);
    {   // ec::proj_dbl alternating with proj_add (the point decoder's subgroup-test ladders, round 4)
        ec::Proj<F> d = pinv;
        for (int it = 0; it < 4; it++) {
            ec::proj_dbl<F>(d);
            ec::Proj<F> t = d;
            ec::proj_add<F>(t, pinv);
            d = ec::Proj<F>{mx(d.x, t.x), mx(d.y, t.y), mx(d.z, t.z)};
        }
        printf("proj_dbl / proj_add ladder: X < %.0fp, Y < %.0fp, Z < %.0fp\n", d.x.v, d.y.v, d.z.v);
    }
    // --- the subgroup ladders: G1 (both ladders: the second adds a Jacobian point), G2 on the lane-pair field (affine additions only)
    check_jac<BoundFp, false, true>("G1");
    check_jac<BoundCoop, true, false>("G2 (lane pair)");
    // --- G2: hot loop on the inlined Fp2 variant, everything else on the shared-call variant (msm_kernels.cuh: G2C)
    check_g2<BoundFp2<true>, BoundFp2<false>>("G2");
    printf("msm bounds OK: largest product %.0f p^2 (limit %.0f), largest column sum 2^%.2f (limit 2^%.2f), largest limb 2^%.2f\n",
           g_max_prod, LIMIT, std::log2(g_max_col), std::log2(COLMAX), std::log2(g_max_limb));
    return 0;
}
