// Machine check of the lazy-reduction bookkeeping of ark-blst_amd/csrc/pairing.cuh: the generic tower / Miller loop /
// final exponentiation is instantiated with an Fp2 class that carries VALUE BOUNDS (multiples of p) instead of values
// and enforces, at every call, the input contract of the real operations (fp28.cuh / ec.cuh):
//   Montgomery product      sum of a_i * b_i < 2^392 / p  (= LIMIT, ~2520 p^2 in units of p^2);   output < 2p
//   fp_sub<K>(a, b)         b <= (K-1) p;                                                          output < a + K p
//   Fp2 product             negates b.c1 as 32p - b.c1: b.c1 <= 31p
//   any sum                 < 2^392 (fits 14 limbs)
// Test helper only — not part of the shipped library.  Prints the largest bounds seen; exits non-zero on a violation.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "../../ark-blst_amd/csrc/pairing.cuh"

static const double LIMIT = 2520.0;   // floor(2^392 / p) = 2520.7...
static double g_max_mul = 0, g_max_val = 0;

static void fail(const char* what, double v, double lim) {
    fprintf(stderr, "BOUND VIOLATION: %s: %.1f > %.1f\n", what, v, lim);
    exit(1);
}
static double chk_val(double v) {
    if (v >= LIMIT) fail("value does not fit 2^392", v, LIMIT);
    g_max_val = std::max(g_max_val, v);
    return v;
}
static double mont(double sum_of_products) {   // one Montgomery reduction of an accumulated sum
    if (sum_of_products >= LIMIT) fail("multiplier input", sum_of_products, LIMIT);
    g_max_mul = std::max(g_max_mul, sum_of_products);
    return 2.0;
}

struct B2 {
    double c0, c1;
};
struct BoundF2 {
    using E = B2;
    using Fp = double;
    static E zero() { return {0, 0}; }
    static E one() { return {1, 0}; }
    static double fsub(int K, double a, double b, const char* what) {
        if (b > K - 1) fail(what, b, K - 1);
        return chk_val(a + K);
    }
    static E mul(const E& a, const E& b) {
        if (b.c1 > 31) fail("Fp2 mul: b.c1", b.c1, 31);
        return {mont(a.c0 * b.c0 + a.c1 * 32.0), mont(a.c0 * b.c1 + a.c1 * b.c0)};
    }
    static E sqr(const E& a) {
        double d = fsub(32, a.c0, a.c1, "Fp2 sqr: a.c1");
        return {mont((a.c0 + a.c1) * d), mont(2 * a.c0 * a.c1)};
    }
    static E sqr_sub12sqr(const E& s, const E& e) {   // one fused two-product reduction per component (ec.cuh)
        double d = fsub(32, s.c0, s.c1, "sqr_sub12sqr: s.c1");
        double w0 = chk_val(12 * fsub(4, e.c1, e.c0, "sqr_sub12sqr: e.c0")), w1 = chk_val(12 * fsub(4, 0, e.c1, "sqr_sub12sqr: e.c1"));
        return {mont((s.c0 + s.c1) * d + (e.c0 + e.c1) * w0), mont(2 * s.c0 * s.c1 + 2 * e.c0 * w1)};
    }
    static E mul2add(const E& a, const E& b, const E& c, const E& d) {   // the shared (non-inlined) variant: two reductions per component
        if (b.c1 > 31 || d.c1 > 31) fail("Fp2 mul2add: c1", std::max(b.c1, d.c1), 31);
        return {chk_val(mont(a.c0 * b.c0 + a.c1 * 32.0) + mont(c.c0 * d.c0 + c.c1 * 32.0)),
                chk_val(mont(a.c0 * b.c1 + a.c1 * b.c0) + mont(c.c0 * d.c1 + c.c1 * d.c0))};
    }
    static E add(const E& a, const E& b) { return {chk_val(a.c0 + b.c0), chk_val(a.c1 + b.c1)}; }
    template <int K>
    static E sub(const E& a, const E& b) { return {fsub(K, a.c0, b.c0, "sub"), fsub(K, a.c1, b.c1, "sub")}; }
    template <int K>
    static E neg(const E& a) { return {fsub(K, 0, a.c0, "neg"), fsub(K, 0, a.c1, "neg")}; }
    static E mul3(const E& a) { return {chk_val(3 * a.c0), chk_val(3 * a.c1)}; }
    static E mul_b3(const E& a) { return mul(a, E{1, 1}); }
    template <int K>
    static E mul_xi(const E& a) { return {fsub(K, a.c0, a.c1, "mul_xi"), chk_val(a.c0 + a.c1)}; }
    static E mul_fp(const E& a, const Fp& s) { return {mont(a.c0 * s), mont(a.c1 * s)}; }
    static E norm2(const E& a) { return {mont(a.c0), mont(a.c1)}; }
    static E dbl(const E& a) { return add(a, a); }
    static Fp fp_neg4(const Fp& a) { return fsub(4, 0, a, "fp_neg4"); }
    static E inv(const E& a) {
        double n = chk_val(mont(a.c0 * a.c0) + mont(a.c1 * a.c1));
        mont(n * 2);                       // the power ladder multiplies values < 2p by n
        double ni = 2;
        return {mont(a.c0 * ni), mont(fsub(32, 0, a.c1, "inv") * ni)};
    }
    static E frob_const(int) { return {1, 1}; }
    static E select(bool, const E& a, const E& b) { return {std::max(a.c0, b.c0), std::max(a.c1, b.c1)}; }
};

int main() {
    using T = pairing::Tower<BoundF2>;
    // inputs as the kernel prepares them: x, y of both points are outputs of the ingest multiplication (< 2p); -xP = 4p - xP
    T::G1Pt p{4.0, 2.0};
    B2 xq{2, 2}, yq{2, 2};
    T::E12 f = T::miller_loop(p, xq, yq);
    // product tree / final exponentiation take normalised values
    T::E12 g = T::mul12(f, f);
    T::E12 e = T::final_exp(g);
    double m = 0;
    const B2* c[6] = {&e.c0.c0, &e.c0.c1, &e.c0.c2, &e.c1.c0, &e.c1.c1, &e.c1.c2};
    for (int i = 0; i < 6; i++) m = std::max(m, std::max(c[i]->c0, c[i]->c1));
    printf("pairing bounds OK: largest multiplier input %.0f p^2 (limit %.0f), largest value %.0f p, outputs < %.0f p\n", g_max_mul,
           LIMIT, g_max_val, m);
    return 0;
}
