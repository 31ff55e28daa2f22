// Short-Weierstrass (a = 0) group law over a lazily reduced field.  Every formula is straight-line code around
// ONE shared, out-of-line field multiplication (F::mul is a real device function call, see fp28::fp_mul_call).
//
// Why: with the multiplier inlined, a mixed addition is ~80 KB of straight-line code.  MI355X shares a 64 KB
// instruction cache between two CUs and the waves of a bucket kernel are NOT in lockstep, so that body thrashes
// it: measured 962 ms for 10 M additions in the first version of this file (~1900 cycles per instruction), while
// a lockstep micro-benchmark of the same size runs at full rate (tools/ubench_icache.hip).  With one ~4.5 KB
// multiplier body per kernel the hot loop stays cache resident; the call costs ~42 register moves (~4 %).
//
// Two coordinate systems:
//   * XYZZ (x = X/ZZ, y = Y/ZZZ) for the bucket ACCUMULATION hot loop: mixed addition madd-2008-s, 10 M.
//     Its exceptional inputs (same x: doubling / cancellation) are only DETECTED there; the lane then leaves
//     the hot loop and finishes its bucket on the complete path below.
//   * homogeneous projective (X : Y : Z) with the COMPLETE addition of Renes-Costello-Batina 2016, Alg. 7
//     (a = 0, b3 = 3b), 12 M, no exceptional cases at all (doubling, inverses, infinity (0:1:0)) — used for
//     bucket reduction, the wave scan, doublings and the cold path.
//
// This is the bucket arithmetic of the MSM that replaces the generated POINT_multiexp kernels
// (/root/reference/build.rs:9-11, driven by /root/reference/src/gpu.rs:165-183); unlike those it handles
// infinity among the bases (the reference's known-bad case, /root/reference/src/g1.rs:682-688).
//
// Value bounds (multiples of p; all values N-form) are proved by tools/bounds_check.py:
//   XYZZ accumulator: X < 10p, Y < 6p, ZZ, ZZZ < 2p; affine inputs x, y < 4p; projective coordinates < 8p.
#pragma once
#include "fp28.cuh"

namespace ec {

struct FpOps {
    using E = fp28::Fp;
    static constexpr int B3 = 12;  // 3b for y^2 = x^3 + 4
    static FP_HD E zero() { return fp28::fp_zero(); }
    static FP_HD E one() { return fp28::fp_one(); }
    static FP_HD E mul(const E& a, const E& b) { return fp28::fp_mul_call(a, b); }
    static FP_HD E sqr(const E& a) { return fp28::fp_sqr_call(a); }
    static FP_HD E mul2add(const E& a, const E& b, const E& c, const E& d) { return fp28::fp_add(fp28::fp_mul_call(a, b), fp28::fp_mul_call(c, d)); }
    static FP_HD E add(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static FP_HD E sub(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static FP_HD E neg(const E& a) { return fp28::fp_neg<K>(a); }
    static FP_HD E mul3(const E& a) { return fp28::fp_mul_small<3>(a); }
    template <int K>
    static FP_HD E mul_small(const E& a) { return fp28::fp_mul_small<K>(a); }
    static FP_HD E reduce_small(const E& a) { return fp28::fp_reduce_small(a); }   // value < 127p -> [p, 2.01p) (the Jacobian ladder below)
    static FP_HD E mul_b3(const E& a) { return fp28::fp_mul_small<12>(a); }
    static FP_HD E mul_b3_red(const E& a) { return fp28::fp_mul_call(a, fp28::fp_const(fp28::TWELVE)); }   // 12 a as a field product: < 2p
    // s^2 - 12 e^2 (proj_dbl): s <= 22p, e < 3p
    static FP_HD E sqr_sub12sqr(const E& s, const E& e) { return fp28::fp_add(fp28::fp_mul_call(s, s), fp28::fp_mul_call(e, fp28::fp_mul_small<12>(fp28::fp_neg<4>(e)))); }
    static FP_HD bool is_zero_2p(const E& a) { return fp28::fp_is_zero_2p(a); }
    static FP_HD E select(bool take_b, const E& a, const E& b) { return fp28::fp_select(take_b, a, b); }
    // "lazy" hooks used by xyzz_madd: here they are the normalising operations (safe everywhere)
    static FP_HD E add_l(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static FP_HD E sub_l(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static FP_HD E neg_l(const E& a) { return fp28::fp_neg<K>(a); }
    static FP_HD E sub8_wide(const E& a, const E& b) { return fp28::fp_sub<8>(a, b); }
    static FP_HD E norm(const E& a) { return a; }
    static FP_HD bool limbs_all_zero(const E& a) {
        uint32_t z = 0;
#pragma unroll
        for (int k = 0; k < fp28::NL; k++) z |= a.l[k];
        return z == 0;
    }
};

// Same field with the multiplier INLINED at every use: for the accumulate hot loop only.  Measured on MI355X
// (2^20 points): 2.87 ms inlined vs 3.43 ms through the shared call — the ~45 KB loop body still streams from the
// instruction cache, and the compiler schedules across multiplication boundaries.  Everything that is not the hot
// loop keeps the shared call (code size: a complete addition is 12 multiplications).
struct FpOpsInline : FpOps {
    // operand-scanning form of the multiplier (fp28.cuh): 2 % faster than product scanning in this one loop
    static FP_HD E mul(const E& a, const E& b) { return fp28::fp_mul_os(a, b); }
    static FP_HD E sqr(const E& a) { return fp28::fp_sqr_os(a); }
    static FP_HD E mul2add(const E& a, const E& b, const E& c, const E& d) { return fp28::fp_mul2add_os(a, b, c, d); }
    static FP_HD bool is_zero_2p(const E& a) { return fp28::fp_is_zero_2p_exact(a); }   // only ever asked of a multiplier output
    // truly lazy linear operations (no carry pass); limb bounds proved in tools/bounds_check.py check_madd_lazy()
    static FP_HD E add_l(const E& a, const E& b) { return fp28::fp_add_lazy(a, b); }
    template <int K>
    static FP_HD E sub_l(const E& a, const E& b) { return fp28::fp_sub_lazy<K>(a, b); }
    template <int K>
    static FP_HD E neg_l(const E& a) { return fp28::fp_sub_lazy<K>(fp28::fp_zero(), a); }
    static FP_HD E sub8_wide(const E& a, const E& b) { return fp28::fp_sub8_lazy_wide(a, b); }
    static FP_HD E norm(const E& a) { return fp28::fp_norm(a); }
};

// The same with the product-scanning multiplier: the serial bucket reduction (k_reduce_serial)
struct FpOpsInlinePS : FpOpsInline {
    static FP_HD E mul(const E& a, const E& b) { return fp28::fp_mul(a, b); }
    static FP_HD E sqr(const E& a) { return fp28::fp_sqr(a); }
    static FP_HD E mul2add(const E& a, const E& b, const E& c, const E& d) { return fp28::fp_mul2add(a, b, c, d); }
    static FP_HD E mul_b3_red(const E& a) { return fp28::fp_mul(a, fp28::fp_const(fp28::TWELVE)); }
    static FP_HD E sqr_sub12sqr(const E& s, const E& e) { return fp28::fp_mul2add(s, s, e, fp28::fp_mul_small<12>(fp28::fp_neg<4>(e))); }   // one reduction: < 2p
};

// ------------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2 + 1) for G2 (layout (c0, c1): /root/reference/src/fp2.rs:228).  A product is two fused
// two-term multiply-adds, each with ONE Montgomery reduction, so every component of a product is < 2p and the
// bound analysis of the Fp formulas carries over component-wise (tools/bounds_check.py, fp2 mode):
//   c0 = a0 b0 + a1 (32p - b1)      needs b1 <= 31p and a (b + 32) < 2^392/p ~ 2520
//   c1 = a0 b1 + a1 b0              needs 2 a b < 2520
// b3 = 3 * 4(1 + u) = 12 + 12u is applied as a field multiplication (keeps the result < 2p).
struct Fp2 {
    fp28::Fp c0, c1;
};

template <bool INLINE>
struct Fp2OpsT {
    using E = Fp2;
    using Fp = fp28::Fp;
    static FP_HD Fp m1(const Fp& a, const Fp& b) {
        if constexpr (INLINE) return fp28::fp_mul(a, b); else return fp28::fp_mul_call(a, b);
    }
    static FP_HD Fp m2(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
        if constexpr (INLINE) return fp28::fp_mul2add(a, b, c, d); else return fp28::fp_mul2add_call(a, b, c, d);
    }
    static FP_HD E zero() { return E{fp28::fp_zero(), fp28::fp_zero()}; }
    static FP_HD E one() { return E{fp28::fp_one(), fp28::fp_zero()}; }
    static FP_HD E mul(const E& a, const E& b) {
        Fp nb1 = fp28::fp_neg<32>(b.c1);
        E r;
        r.c0 = m2(a.c0, b.c0, a.c1, nb1);
        r.c1 = m2(a.c0, b.c1, a.c1, b.c0);
        return r;
    }
    static FP_HD E sqr(const E& a) {  // (a0 + a1)(a0 - a1) + 2 a0 a1 u ; a < 22p
        E r;
        r.c0 = m1(fp28::fp_add(a.c0, a.c1), fp28::fp_sub<32>(a.c0, a.c1));
        r.c1 = m1(fp28::fp_add(a.c0, a.c0), a.c1);
        return r;
    }
    static FP_HD E mul2add(const E& a, const E& b, const E& c, const E& d) {  // a b + c d ; b, d <= 31p
        Fp nb1 = fp28::fp_neg<32>(b.c1), nd1 = fp28::fp_neg<32>(d.c1);
        E r;
        if constexpr (INLINE) {
            r.c0 = fp28::fp_mul4add(a.c0, b.c0, a.c1, nb1, c.c0, d.c0, c.c1, nd1);
            r.c1 = fp28::fp_mul4add(a.c0, b.c1, a.c1, b.c0, c.c0, d.c1, c.c1, d.c0);
        } else {
            r.c0 = fp28::fp_add(m2(a.c0, b.c0, a.c1, nb1), m2(c.c0, d.c0, c.c1, nd1));
            r.c1 = fp28::fp_add(m2(a.c0, b.c1, a.c1, b.c0), m2(c.c0, d.c1, c.c1, d.c0));
        }
        return r;
    }
    // s^2 - 12 e^2 with ONE reduction per component (the doubling step of the Miller loop: Y3 = (B + 3E)^2 - 12 E^2):
    //   c0 = (s0 + s1)(s0 - s1) + (e0 + e1) * 12 (e1 - e0),   c1 = (2 s0) s1 + (2 e0) * 12 (-e1);   s <= 22p, e < 3p; result < 2p
    static FP_HD E sqr_sub12sqr(const E& s, const E& e) {
        E r;
        r.c0 = m2(fp28::fp_add(s.c0, s.c1), fp28::fp_sub<32>(s.c0, s.c1), fp28::fp_add(e.c0, e.c1), fp28::fp_mul_small<12>(fp28::fp_sub<4>(e.c1, e.c0)));
        r.c1 = m2(fp28::fp_add(s.c0, s.c0), s.c1, fp28::fp_add(e.c0, e.c0), fp28::fp_mul_small<12>(fp28::fp_neg<4>(e.c1)));
        return r;
    }
    static FP_HD E add(const E& a, const E& b) { return E{fp28::fp_add(a.c0, b.c0), fp28::fp_add(a.c1, b.c1)}; }
    template <int K>
    static FP_HD E sub(const E& a, const E& b) { return E{fp28::fp_sub<K>(a.c0, b.c0), fp28::fp_sub<K>(a.c1, b.c1)}; }
    template <int K>
    static FP_HD E neg(const E& a) { return E{fp28::fp_neg<K>(a.c0), fp28::fp_neg<K>(a.c1)}; }
    static FP_HD E add_l(const E& a, const E& b) { return add(a, b); }
    template <int K>
    static FP_HD E sub_l(const E& a, const E& b) { return sub<K>(a, b); }
    template <int K>
    static FP_HD E neg_l(const E& a) { return neg<K>(a); }
    static FP_HD E sub8_wide(const E& a, const E& b) { return sub<8>(a, b); }
    static FP_HD E norm(const E& a) { return a; }
    static FP_HD E mul3(const E& a) { return E{fp28::fp_mul_small<3>(a.c0), fp28::fp_mul_small<3>(a.c1)}; }
    static FP_HD E mul_b3(const E& a) {
        E b3{fp28::fp_const(fp28::TWELVE), fp28::fp_const(fp28::TWELVE)};
        return mul(a, b3);
    }
    static FP_HD E mul_b3_red(const E& a) { return mul_b3(a); }   // already a field product: < 2p
    static FP_HD bool is_zero_2p(const E& a) { return fp28::fp_is_zero_2p(a.c0) && fp28::fp_is_zero_2p(a.c1); }
    static FP_HD E select(bool take_b, const E& a, const E& b) {
        return E{fp28::fp_select(take_b, a.c0, b.c0), fp28::fp_select(take_b, a.c1, b.c1)};
    }
    static FP_HD bool limbs_all_zero(const E& a) { return FpOps::limbs_all_zero(a.c0) && FpOps::limbs_all_zero(a.c1); }
};
using Fp2Ops = Fp2OpsT<false>;
using Fp2OpsInline = Fp2OpsT<true>;

// Points are structs over the ELEMENT type, and Xyzz<F> / Proj<F> are aliases: every operation class with the same element
// (FpOps / FpOpsInline / FpOpsInlinePS; CoopF2 / CoopF2A) names ONE struct type, so a value computed with one multiplier form
// is handed to code instantiated for another without a cast (round 3 cast between distinct instantiations: undefined behaviour
// under the strict-aliasing rules the build uses).  F is not deducible through the alias: callers name it, proj_add<F>(a, b).
template <class E>
struct XyzzE {
    E x, y, zz, zzz;
};
template <class E>
struct ProjE {
    E x, y, z;
};
template <class F>
using Xyzz = XyzzE<typename F::E>;
template <class F>
using Proj = ProjE<typename F::E>;

template <class F>
FP_HD Proj<F> proj_inf() {
    Proj<F> r;
    r.x = F::zero(); r.y = F::one(); r.z = F::zero();
    return r;
}
template <class F>
FP_HD Proj<F> proj_from_affine(const typename F::E& x, const typename F::E& y) {
    Proj<F> r;
    r.x = x; r.y = y; r.z = F::one();
    return r;
}
template <class F>
FP_HD Proj<F> proj_select(bool take_b, const Proj<F>& a, const Proj<F>& b) {
    Proj<F> r;
    r.x = F::select(take_b, a.x, b.x);
    r.y = F::select(take_b, a.y, b.y);
    r.z = F::select(take_b, a.z, b.z);
    return r;
}

// a <- a + b, complete (RCB16 Alg. 7): 12 multiplications (9 reductions with the fused form), no exceptional cases.
template <class F>
FP_HD void proj_add(Proj<F>& a, const Proj<F>& b) {
    using E = typename F::E;
    E t0 = F::mul(a.x, b.x);
    E t1 = F::mul(a.y, b.y);
    E t2 = F::mul(a.z, b.z);
    E t3 = F::mul(F::add(a.x, a.y), F::add(b.x, b.y));
    E t4 = F::mul(F::add(a.y, a.z), F::add(b.y, b.z));
    E t5 = F::mul(F::add(a.x, a.z), F::add(b.x, b.z));
    t3 = F::template sub<8>(t3, F::add(t0, t1));   // X1Y2 + X2Y1
    t4 = F::template sub<8>(t4, F::add(t1, t2));   // Y1Z2 + Y2Z1
    t5 = F::template sub<8>(t5, F::add(t0, t2));   // X1Z2 + X2Z1
    t0 = F::mul3(t0);                              // 3 X1X2
    t2 = F::mul_b3(t2);                            // b3 Z1Z2
    E u = F::add(t1, t2);                          // Y1Y2 + b3 Z1Z2
    t1 = F::template sub<32>(t1, t2);              // Y1Y2 - b3 Z1Z2
    t5 = F::mul_b3(t5);                            // b3 (X1Z2 + X2Z1)
    // each output is ONE fused two-product reduction where the field has one (fp_mul2add): three reductions fewer
    a.x = F::mul2add(t1, t3, t5, F::template neg<16>(t4));     // t1 t3 - t5 t4  (t4 <= 10p; Fp2: second operands <= 31p)
    a.y = F::mul2add(t1, u, t5, t0);
    a.z = F::mul2add(u, t4, t0, t3);
}

// a <- 2 a, exception-free for a = 0: the three values of Renes-Costello-Batina 2016 Alg. 9 rearranged so that squarings and one
// fused difference of squares replace products (the Miller loop's doubling step, pairing.cuh line_dbl) — with B = Y^2, C = Z^2,
// E = b3 C, H = (Y + Z)^2 - B - C = 2YZ:   X3 = 2 XY (B - 3E),   Y3 = (B + 3E)^2 - 12 E^2,   Z3 = 4 H B.
// 3 S + 4 M + one fused pair (about 6.8 multiplications over Fp) where the complete addition of a point to itself takes 12.
// Coordinates <= 8p in; x < 4p, y < 4p, z < 2p out.
template <class F>
FP_HD void proj_dbl(Proj<F>& a) {
    using E = typename F::E;
    E B = F::sqr(a.y), C = F::sqr(a.z);
    E H = F::template sub<8>(F::sqr(F::add(a.y, a.z)), F::add(B, C));    // 2 Y Z                  < 10p
    E Eb = F::mul_b3_red(C);                                              // b3 Z^2                 < 2p
    E E3 = F::mul3(Eb);                                                   //                        < 6p
    E xy = F::mul(a.x, a.y);
    E d = F::mul(F::template sub<8>(B, E3), xy);
    a.x = F::add(d, d);                                                   // 2 XY (B - 3E)          < 4p
    E H2 = F::add(H, H);
    a.z = F::mul(F::add(H2, H2), B);                                      // 4 H B                  < 2p
    a.y = F::sqr_sub12sqr(F::add(B, E3), Eb);                             // (B + 3E)^2 - 12 E^2    < 4p
}

// r = 2^k * a
template <class F>
FP_HD void proj_dbl_n(Proj<F>& a, int k) {
#pragma unroll 1
    for (int i = 0; i < k; i++) {
        Proj<F> c = a;
        proj_add<F>(a, c);
    }
}

// ------------------------------------------------------------------------------------------------
// Jacobian arithmetic over Fp for the G1 subgroup ladder (is_torsion_free, /root/reference/src/g1.rs:386-431 -> blst_p1_affine_in_g1):
// x = X / Z^2, y = Y / Z^3, infinity is Z == 0 (mod p).  The ladder is 126 doublings and 10 additions, so the doubling is the cost: the
// a = 0 Jacobian doubling is 4 S + 3 M + one small reduction (2478 multiply-adds + ~120 plain instructions) where proj_dbl takes 3178.
//   The formulas are NOT complete, and need not be: an addition whose operands agree up to sign, or with an operand at infinity, leaves
//   Z3 == 0, and Z == 0 survives every later doubling and addition — the caller reads that as "not in the subgroup".  That answer is right:
//   with P of prime order r the ladder's partial multiples [k]P, 1 < k < 2^64, are never 0 or +-P, so only points outside G1 can reach
//   those cases (small-order points do, tests/test_host_model.py and the GPU suite hold them).
// Bounds: coordinates in X < 10p, Y < 34p, Z < 4p; jac_dbl leaves X < 2.01p, Y < 34p, Z < 4p; jac_add leaves X, Y < 10p, Z < 2p.
// ------------------------------------------------------------------------------------------------
template <class E>
struct JacE {
    E x, y, z;
};
using JacFp = JacE<fp28::Fp>;

// a <- 2a (dbl-2009-l with X B as a product): A = X^2, B = Y^2, C = B^2, S = X B, E = 3A, X3 = E^2 - 8S, Y3 = E (4S - X3) - 8C, Z3 = 2YZ
// REDUCE_Y: Y3 is brought below 2.01p as well — for the lane-pair Fp2 field (coop_fp2.cuh), whose squaring forms (a + a')(a + 32p - a') and
// takes components below 20p only.  Every operation goes through F, so tests/host/msm_bounds.cpp runs this very code on a field of bounds.
template <class F, bool REDUCE_Y = false>
FP_HD void jac_dbl(JacE<typename F::E>& a) {
    using E = typename F::E;
    const E A = F::sqr(a.x), B = F::sqr(a.y), C = F::sqr(B);
    const E S = F::mul(a.x, B);
    const E M = F::template mul_small<3>(A);                                                          // 3 X^2 < 6p
    const E yz = F::mul(a.y, a.z);                                                                    // 34p * 4p
    a.x = F::reduce_small(F::template sub<32>(F::sqr(M), F::template mul_small<8>(S)));               // 8S < 16p; < 34p before the reduction, < 2.01p after
    a.y = F::template sub<32>(F::mul(M, F::template sub<4>(F::template mul_small<4>(S), a.x)), F::template mul_small<8>(C));   // 6p * 12p; < 34p
    if (REDUCE_Y) a.y = F::reduce_small(a.y);
    a.z = F::add(yz, yz);                                                                             // < 4p
}

// a <- a + b (add-2007-bl; Z2_ONE: b is affine, b.z is not read — madd-2007-bl)
template <class F, bool Z2_ONE>
FP_HD void jac_add(JacE<typename F::E>& a, const JacE<typename F::E>& b) {
    using E = typename F::E;
    const E z1z1 = F::sqr(a.z);
    const E U2 = F::mul(b.x, z1z1), S2 = F::mul(F::mul(b.y, a.z), z1z1);
    E U1, S1, zsum;
    if (Z2_ONE) {
        U1 = F::reduce_small(a.x);                                                                    // < 2.01p
        S1 = F::reduce_small(a.y);
        zsum = a.z;
    } else {
        const E z2z2 = F::sqr(b.z);
        U1 = F::mul(a.x, z2z2);
        S1 = F::mul(F::mul(a.y, b.z), z2z2);
        zsum = F::template sub<8>(F::sqr(F::add(a.z, b.z)), F::add(z1z1, z2z2));                      // 2 Z1 Z2  < 10p
    }
    const E H = F::template sub<4>(U2, U1);                                                           // < 6p
    const E H2 = F::add(H, H);
    const E I = F::sqr(H2), J = F::mul(H, I), V = F::mul(U1, I);
    const E d = F::template sub<4>(S2, S1);
    const E r = F::add(d, d);                                                                         // < 12p
    const E S1J = F::mul(S1, J);
    a.x = F::template sub<8>(F::sqr(r), F::add(J, F::add(V, V)));                                     // J + 2V < 6p; < 10p
    a.y = F::template sub<8>(F::mul(r, F::template sub<16>(V, a.x)), F::add(S1J, S1J));               // 12p * 18p; < 10p
    a.z = Z2_ONE ? F::mul(F::add(zsum, zsum), H) : F::mul(zsum, H);                                   // < 2p
}

// [|z|] p for the BLS parameter |z| = 0xd201000000010000: 63 doublings with the multiplier of FD (inlined in the kernels), 5 additions
// with FA's (the shared call), on a copy so that the loop variable's address is never taken (codec_kernels.cuh, round 6)
template <class FD, class FA, bool AFFINE, bool REDUCE_Y = false>
FP_HD JacE<typename FD::E> jac_mul_z(const JacE<typename FD::E>& p) {
    JacE<typename FD::E> r = p;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int bit = 62; bit >= 0; bit--) {
        jac_dbl<FD, REDUCE_Y>(r);
        if ((fp28c::Z_ABS >> bit) & 1) {
            JacE<typename FD::E> t = r;
            jac_add<FA, AFFINE>(t, p);
            r = t;
        }
    }
    return r;
}

// is_torsion_free of an affine point of the curve (not infinity): (beta x, y) == -[z^2] P (Scott 2021; what blst_p1_affine_in_g1 tests), in
// Jacobian coordinates X == beta x Z^2, Y == -y Z^3, Z != 0.  x, y < 4p.
template <class FD, class FA>
FP_HD bool g1_torsion_free(const fp28::Fp& x, const fp28::Fp& y) {
    using namespace fp28;
    JacFp p;
    p.x = x; p.y = y; p.z = fp_one();
    const JacFp q1 = jac_mul_z<FD, FA, true>(p);
    const JacFp q2 = jac_mul_z<FD, FA, false>(q1);                                 // [z^2] P
    const Fp zz = FA::sqr(q2.z);
    const Fp bx = FA::mul(x, fp_const(fp28c::BETA));
    bool ok = !fp_is_zero_any(q2.z);
    ok = ok && fp_is_zero_any(fp_sub<16>(q2.x, FA::mul(bx, zz)));                  // q2.x < 10p
    ok = ok && fp_is_zero_any(fp_add(q2.y, FA::mul(y, FA::mul(zz, q2.z))));        // q2.y < 34p
    return ok;
}

template <class F>
FP_HD bool proj_is_inf_exact(const Proj<F>& p) { return F::limbs_all_zero(p.z); }

// Hot-loop mixed addition acc <- acc + (x2, y2) in XYZZ, acc NOT infinity.  Returns true ("special") without
// touching acc when the x-coordinates agree (P == 0 mod p): the caller finishes that bucket on the complete path.
// Scheduled for few live values: three temporaries besides the accumulator and the point.
template <class F>
FP_HD bool xyzz_madd(Xyzz<F>& acc, const typename F::E& x2, const typename F::E& y2) {
    using E = typename F::E;
    // limb classes: E exact (< 2^28, every multiplier output), N (<= 2^28+15 after norm), U (< 2^29.6, one lazy sub).
    // Invariant: acc.x is N, acc.y / acc.zz / acc.zzz are E; x2 is E, y2 is E or U (negated).
    E t0 = F::template sub_l<16>(F::mul(x2, acc.zz), acc.x);  // P = U2 - X1          < 18p   U
    E t2 = F::sqr(t0);                                         // PP                           E
    if (F::is_zero_2p(t2)) return true;
    E t1 = F::template sub_l<8>(F::mul(y2, acc.zzz), acc.y);  // R = S2 - Y1          < 10p   U
    t0 = F::mul(t0, t2);                                       // PPP                          E
    acc.zz = F::mul(acc.zz, t2);                               // ZZ3 = ZZ1 PP
    t2 = F::mul(acc.x, t2);                                    // Q = X1 PP                    E
    acc.zzz = F::mul(acc.zzz, t0);                             // ZZZ3 = ZZZ1 PPP
    E ny = F::template neg_l<8>(acc.y);                        // 8p - Y1              < 8p    U
    E t3 = F::add_l(F::add_l(t0, t2), t2);                     // PPP + 2Q             < 6p    limbs < 3 * 2^28
    acc.x = F::norm(F::sub8_wide(F::sqr(t1), t3));             // X3 = R^2 - PPP - 2Q  < 10p   N
    t2 = F::template sub_l<16>(t2, acc.x);                     // Q - X3               < 18p   U
    acc.y = F::mul2add(t1, t2, ny, t0);                        // Y3 = R (Q - X3) + (8p - Y1) PPP, one reduction  < 2p  E
    return false;
}

// XYZZ -> projective: (X ZZZ : Y ZZ : ZZ ZZZ)
template <class F>
FP_HD Proj<F> xyzz_to_proj(const Xyzz<F>& a) {
    Proj<F> r;
    r.x = F::mul(a.x, a.zzz);
    r.y = F::mul(a.y, a.zz);
    r.z = F::mul(a.zz, a.zzz);
    return r;
}

}  // namespace ec
