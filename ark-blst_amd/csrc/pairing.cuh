// Optimal-ate pairing on BLS12-381 over the 28-bit lazily reduced field: what sits under
// `<Bls12 as Pairing>::multi_miller_loop` and `final_exponentiation` (/root/reference/src/pairing.rs:49-74, 76-80;
// the reference forwards both to blstrs/blst on ONE CPU thread, pair after pair).
//
// Everything here is generic over an Fp2 operation class F2 (interface of ec::Fp2OpsT plus the few extras of PF2
// below), for two reasons:
//   * the same source is the GPU Miller loop (one lane per pair) and the host-side final exponentiation;
//   * tests/host/pairing_bounds.cpp instantiates it with a class that tracks VALUE BOUNDS instead of values and
//     asserts the multiplier's input contract at every call, so the lazy-reduction bookkeeping below is machine
//     checked on the real code, not on a transcript of it.
//
// Tower (the reference's: src/fp2.rs, fp6.rs, fp12.rs):  Fp2 = Fp[u]/(u^2+1),  Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u,
// Fp12 = Fp6[w]/(w^2 - v).  Memory order of blst_fp12 = c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (each an Fp2).
//
// The Fp6 / Fp12 functions are out of line (FP_HD_NOINLINE): on the GPU their arguments travel through scratch, a few
// percent of the ~10^4 instructions each of them executes, and the Miller-loop kernel stays a few KB of code around
// the shared multiplier bodies instead of ~200 call sites' worth of register marshalling.
//
// Bound contract (units of p, per Fp component): every Fp6/Fp12 function takes components <= 4p and returns
// components < 2p ("normalised": each component is the output of a Montgomery reduction).  Inside, Karatsuba sums
// and differences grow to < 92p; they are brought back by one multiplication by the internal one (F2::norm).
#pragma once
#include "ec.cuh"

namespace pairing {

// Fp2 operations the tower needs on top of ec::Fp2Ops
struct PF2 : ec::Fp2Ops {
    using Fp = fp28::Fp;
    // a * xi = (a0 - a1) + (a0 + a1) u ; needs a.c1 <= (K-1)p
    template <int K>
    static FP_HD E mul_xi(const E& a) { return E{fp28::fp_sub<K>(a.c0, a.c1), fp28::fp_add(a.c0, a.c1)}; }
    static FP_HD E mul_fp(const E& a, const Fp& s) { return E{fp28::fp_mul_call(a.c0, s), fp28::fp_mul_call(a.c1, s)}; }
    static FP_HD E norm2(const E& a) {   // any value < 2520p -> < 2p
        Fp one = fp28::fp_one();
        return E{fp28::fp_mul_call(a.c0, one), fp28::fp_mul_call(a.c1, one)};
    }
    static FP_HD E dbl(const E& a) { return add(a, a); }
    static FP_HD Fp fp_neg4(const Fp& a) { return fp28::fp_neg<4>(a); }
    // a^(p-2) in Fp; inv(0) = 0
    static FP_HD Fp fp_inv(const Fp& a) {
        Fp acc = fp28::fp_one();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int bit = 380; bit >= 0; bit--) {
            acc = fp28::fp_sqr_call(acc);
            if ((fp28c::EXP_P_2_32[bit >> 5] >> (bit & 31)) & 1) acc = fp28::fp_mul_call(acc, a);
        }
        return acc;
    }
    static FP_HD E inv(const E& a) {   // conj(a) / (a0^2 + a1^2); a <= 22p
        Fp n = fp28::fp_add(fp28::fp_sqr_call(a.c0), fp28::fp_sqr_call(a.c1));
        Fp ni = fp_inv(n);
        return E{fp28::fp_mul_call(a.c0, ni), fp28::fp_mul_call(fp28::fp_neg<32>(a.c1), ni)};
    }
    static FP_HD E frob_const(int i) {   // g^i, g = xi^((p-1)/6), i = 1..5
        E r;
#pragma unroll
        for (int k = 0; k < fp28::NL; k++) {
            r.c0.l[k] = fp28c::FROB_G[(i - 1) * 28 + k];
            r.c1.l[k] = fp28c::FROB_G[(i - 1) * 28 + 14 + k];
        }
        return r;
    }
};

template <class F2>
struct Fp6T {
    typename F2::E c0, c1, c2;
};
template <class F2>
struct Fp12T {
    Fp6T<F2> c0, c1;
};

template <class F2>
struct Tower {
    using E2 = typename F2::E;
    using E6 = Fp6T<F2>;
    using E12 = Fp12T<F2>;

    static FP_HD E6 add6(const E6& a, const E6& b) { return E6{F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
    static FP_HD E6 norm6(const E6& a) { return E6{F2::norm2(a.c0), F2::norm2(a.c1), F2::norm2(a.c2)}; }
    static FP_HD E6 zero6() { return E6{F2::zero(), F2::zero(), F2::zero()}; }
    static FP_HD E12 one12() {
        E12 r;
        r.c0 = E6{F2::one(), F2::zero(), F2::zero()};
        r.c1 = zero6();
        return r;
    }

    // Karatsuba over Fp2 (6 products).  Inputs <= 16p; RAW output: c0 < 28p, c1 < 16p, c2 < 12p.
    static FP_HD_NOINLINE E6 mul6_raw(const E6& a, const E6& b) {
        E2 t0 = F2::mul(a.c0, b.c0), t1 = F2::mul(a.c1, b.c1), t2 = F2::mul(a.c2, b.c2);
        E2 s12 = F2::mul(F2::add(a.c1, a.c2), F2::add(b.c1, b.c2));
        E2 s01 = F2::mul(F2::add(a.c0, a.c1), F2::add(b.c0, b.c1));
        E2 s02 = F2::mul(F2::add(a.c0, a.c2), F2::add(b.c0, b.c2));
        E6 r;
        E2 x = F2::template sub<8>(s12, F2::add(t1, t2));                        // a1 b2 + a2 b1           < 10p
        r.c0 = F2::add(t0, F2::template mul_xi<16>(x));                          //                         < 28p
        r.c1 = F2::add(F2::template sub<8>(s01, F2::add(t0, t1)), F2::template mul_xi<4>(t2));   //        < 16p
        r.c2 = F2::add(F2::template sub<8>(s02, F2::add(t0, t2)), t1);           //                         < 12p
        return r;
    }
    // a * v : (c0, c1, c2) -> (xi c2, c0, c1); K bounds c2
    template <int K>
    static FP_HD E6 mul_v(const E6& a) { return E6{F2::template mul_xi<K>(a.c2), a.c0, a.c1}; }

    // (a0 + a1 w)(b0 + b1 w): 3 Fp6 products + 12 normalisations.  Inputs <= 4p, output < 2p.
    static FP_HD_NOINLINE E12 mul12(const E12& a, const E12& b) {
        E6 t0 = mul6_raw(a.c0, b.c0), t1 = mul6_raw(a.c1, b.c1);
        E6 m = mul6_raw(add6(a.c0, a.c1), add6(b.c0, b.c1));
        E12 r;
        E6 s = add6(t0, t1);                                                     // < 56p
        r.c1 = norm6(E6{F2::template sub<64>(m.c0, s.c0), F2::template sub<64>(m.c1, s.c1), F2::template sub<64>(m.c2, s.c2)});
        r.c0 = norm6(add6(t0, mul_v<16>(t1)));
        return r;
    }
    // complex squaring: 2 Fp6 products
    static FP_HD_NOINLINE E12 sqr12(const E12& a) {
        E6 t = mul6_raw(a.c0, a.c1);
        E6 m = mul6_raw(add6(a.c0, a.c1), add6(a.c0, mul_v<8>(a.c1)));          // (a0 + a1)(a0 + v a1)
        E6 s = add6(t, mul_v<16>(t));                                            // t + v t               < 56p
        E12 r;
        r.c0 = norm6(E6{F2::template sub<64>(m.c0, s.c0), F2::template sub<64>(m.c1, s.c1), F2::template sub<64>(m.c2, s.c2)});
        r.c1 = norm6(add6(t, t));
        return r;
    }
    // a * (d0 + d1 v), a <= 8p, d <= 8p.  RAW output < 8p
    static FP_HD_NOINLINE E6 mul6_by_01(const E6& a, const E2& d0, const E2& d1) {
        E6 r;
        r.c0 = F2::add(F2::mul(a.c0, d0), F2::template mul_xi<4>(F2::mul(a.c2, d1)));
        r.c1 = F2::mul2add(a.c0, d1, a.c1, d0);
        r.c2 = F2::mul2add(a.c1, d1, a.c2, d0);
        return r;
    }
    // a * d1 v.  RAW output < 6p
    static FP_HD_NOINLINE E6 mul6_by_1(const E6& a, const E2& d1) {
        return E6{F2::template mul_xi<4>(F2::mul(a.c2, d1)), F2::mul(a.c0, d1), F2::mul(a.c1, d1)};
    }
    // f * (c0 + c1 v + c4 v w): the sparse line of an M-type twist.  f <= 4p, c0 <= 6p, c1, c4 <= 2p; output < 2p
    static FP_HD_NOINLINE E12 mul_by_014(const E12& f, const E2& c0, const E2& c1, const E2& c4) {
        E6 t0 = mul6_by_01(f.c0, c0, c1);
        E6 t1 = mul6_by_1(f.c1, c4);
        E6 m = mul6_by_01(add6(f.c0, f.c1), c0, F2::add(c1, c4));
        E6 s = add6(t0, t1);                                                     // < 14p
        E12 r;
        r.c1 = norm6(E6{F2::template sub<16>(m.c0, s.c0), F2::template sub<16>(m.c1, s.c1), F2::template sub<16>(m.c2, s.c2)});
        r.c0 = norm6(add6(t0, mul_v<4>(t1)));
        return r;
    }
    static FP_HD E6 neg6n(const E6& a) {   // -a, normalised; a < 2p... (<= 3p)
        return norm6(E6{F2::template neg<4>(a.c0), F2::template neg<4>(a.c1), F2::template neg<4>(a.c2)});
    }
    // conjugation over Fp6 (w -> -w): the inverse on the cyclotomic subgroup
    static FP_HD_NOINLINE E12 conj12(const E12& a) {
        E12 r;
        r.c0 = a.c0;
        r.c1 = neg6n(a.c1);
        return r;
    }
    // 1/a in Fp6 (a <= 4p), output < 2p
    static FP_HD_NOINLINE E6 inv6(const E6& a) {
        E2 A = F2::template sub<8>(F2::sqr(a.c0), F2::template mul_xi<4>(F2::mul(a.c1, a.c2)));   // a0^2 - xi a1 a2      < 10p
        E2 B = F2::template sub<4>(F2::template mul_xi<4>(F2::sqr(a.c2)), F2::mul(a.c0, a.c1));   // xi a2^2 - a0 a1      < 10p
        E2 C = F2::template sub<4>(F2::sqr(a.c1), F2::mul(a.c0, a.c2));                           // a1^2 - a0 a2         < 6p
        E2 Fd = F2::add(F2::mul(a.c0, A), F2::template mul_xi<8>(F2::add(F2::mul(a.c2, B), F2::mul(a.c1, C))));   // < 14p
        E2 Fi = F2::inv(Fd);
        return E6{F2::mul(A, Fi), F2::mul(B, Fi), F2::mul(C, Fi)};
    }
    // 1/a in Fp12
    static FP_HD_NOINLINE E12 inv12(const E12& a) {
        E6 s0 = mul6_raw(a.c0, a.c0), s1 = mul6_raw(a.c1, a.c1);
        E6 vs = mul_v<16>(s1);                                                   // < 28p
        E6 D = norm6(E6{F2::template sub<32>(s0.c0, vs.c0), F2::template sub<32>(s0.c1, vs.c1), F2::template sub<32>(s0.c2, vs.c2)});
        E6 Di = inv6(D);
        E12 r;
        r.c0 = norm6(mul6_raw(a.c0, Di));
        r.c1 = neg6n(norm6(mul6_raw(a.c1, Di)));
        return r;
    }
    // a^p: the coefficient of w^i is conjugated and multiplied by g^i; w-powers of the tower slots:
    // c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)
    static FP_HD E2 conj2n(const E2& a) {   // a0 - a1 u, normalised
        E2 t = a;
        t.c1 = F2::fp_neg4(a.c1);
        return F2::norm2(t);
    }
    static FP_HD_NOINLINE E12 frob12(const E12& a) {
        E12 r;
        r.c0.c0 = conj2n(a.c0.c0);
        r.c0.c1 = F2::mul(conj2n(a.c0.c1), F2::frob_const(2));
        r.c0.c2 = F2::mul(conj2n(a.c0.c2), F2::frob_const(4));
        r.c1.c0 = F2::mul(conj2n(a.c1.c0), F2::frob_const(1));
        r.c1.c1 = F2::mul(conj2n(a.c1.c1), F2::frob_const(3));
        r.c1.c2 = F2::mul(conj2n(a.c1.c2), F2::frob_const(5));
        return r;
    }

    // ---------------------------------------------------------------------------------------- Miller loop
    using PT = ec::Proj<F2>;
    struct G1Pt {   // affine G1 point prepared for line evaluation: (-xP, yP)
        typename F2::Fp nx, y;
    };
    // tangent line at T (scaled by 2YZ) evaluated at P, then T <- 2T.  Round 4: the doubling of Renes-Costello-Batina 2016, Alg. 9
    // (a = 0) rearranged so that squarings replace products — with B = Y^2, C = Z^2, E = b3 C = 3b' Z^2, H = (Y + Z)^2 - B - C = 2YZ:
    //   X3 = 2 XY (B - 3E),   Y3 = (B + 3E)^2 - 12 E^2   (= 8EB + (B - 3E)(B + E)),   Z3 = 4 H B = 8 Y^3 Z
    // the same three VALUES as before (so every Miller value is unchanged), 4 S + 3 M + one fused difference of squares instead of
    // 3 S + 6 M: 5096 multiply-adds per lane of the two-lane kernel instead of 5880.  The line is (B - E) + (-3 X^2 xP) v + (H yP) v w.
    // T <= 6p in, < 4p out.
    static FP_HD void line_dbl(PT& T, const G1Pt& p, E2& c0, E2& c1, E2& c4) {
        E2 B = F2::sqr(T.y);                                                     // Y^2
        E2 C = F2::sqr(T.z);                                                     // Z^2
        E2 H = F2::template sub<8>(F2::sqr(F2::add(T.y, T.z)), F2::add(B, C));   // 2 Y Z                  < 10p
        E2 E = F2::mul_b3(C);                                                    // b3 Z^2 = 3b' Z^2
        c0 = F2::template sub<4>(B, E);                                          // Y^2 - 3b' Z^2          < 6p
        c4 = F2::mul_fp(H, p.y);                                                 // 2 Y Z yP
        c1 = F2::mul_fp(F2::mul3(F2::sqr(T.x)), p.nx);                           // -3 X^2 xP
        E2 xy = F2::mul(T.x, T.y);
        E2 E3 = F2::mul3(E);                                                     //                        < 6p
        T.x = F2::dbl(F2::mul(F2::template sub<8>(B, E3), xy));                  // 2 XY (B - 3E)          < 4p
        T.z = F2::mul(F2::dbl(F2::dbl(H)), B);                                   // 4 H B                  < 2p
        T.y = F2::sqr_sub12sqr(F2::add(B, E3), E);                               // (B + 3E)^2 - 12 E^2    < 2p
    }
    // line through T and Q (scaled by X - xQ Z) evaluated at P, then T <- T + Q
    static FP_HD void line_add(PT& T, const E2& xq, const E2& yq, const G1Pt& p, E2& c0, E2& c1, E2& c4) {
        E2 N = F2::template sub<4>(T.y, F2::mul(yq, T.z));                       // < 10p
        E2 D = F2::template sub<4>(T.x, F2::mul(xq, T.z));
        c0 = F2::template sub<4>(F2::mul(N, xq), F2::mul(D, yq));                // N xQ - D yQ            < 6p
        c1 = F2::mul_fp(N, p.nx);
        c4 = F2::mul_fp(D, p.y);
        PT q = ec::proj_from_affine<F2>(xq, yq);
        ec::proj_add<F2>(T, q);
    }
    // f_{z,Q}(P) up to factors in proper subfields; P, Q affine and not infinity
    static FP_HD E12 miller_loop(const G1Pt& p, const E2& xq, const E2& yq) {
        E12 f = one12();
        PT T = ec::proj_from_affine<F2>(xq, yq);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int i = 62; i >= 0; i--) {
            E2 c0, c1, c4;
            f = sqr12(f);
            line_dbl(T, p, c0, c1, c4);
            f = mul_by_014(f, c0, c1, c4);
            if ((fp28c::Z_ABS >> i) & 1) {
                line_add(T, xq, yq, p, c0, c1, c4);
                f = mul_by_014(f, c0, c1, c4);
            }
        }
        return conj12(f);   // z < 0
    }

    // ---------------------------------------------------------------------------------------- final exponentiation
    // (a + b s)^2 in Fp4 = Fp2[s]/(s^2 - xi): (a^2 + xi b^2, 2ab).  a, b < 2p in; c0 < 8p, c1 < 10p out.
    static FP_HD void fp4_sqr(const E2& a, const E2& b, E2& c0, E2& c1) {
        E2 t0 = F2::sqr(a), t1 = F2::sqr(b);
        c0 = F2::add(F2::template mul_xi<4>(t1), t0);
        E2 s = F2::sqr(F2::add(a, b));
        c1 = F2::template sub<4>(F2::template sub<4>(s, t0), t1);
    }
    // Squaring on the cyclotomic subgroup (Granger-Scott 2010: three Fp4 squarings, 9 Fp2 squarings instead of the 12 Fp2
    // products of sqr12) — what CyclotomicMultSubgroup::cyclotomic_square_in_place forwards to blst for
    // (/root/reference/src/pairing.rs:25-31).  Only valid after the easy part of the final exponentiation.
    static FP_HD_NOINLINE E12 cyclotomic_sqr12(const E12& f) {
        E2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
        E2 t0, t1, t2, t3;
        fp4_sqr(z0, z1, t0, t1);
        z0 = F2::add(F2::dbl(F2::template sub<4>(t0, z0)), t0);                  // 3 t0 - 2 z0
        z1 = F2::add(F2::dbl(F2::add(t1, z1)), t1);                              // 3 t1 + 2 z1
        fp4_sqr(z2, z3, t0, t1);
        fp4_sqr(z4, z5, t2, t3);
        z4 = F2::add(F2::dbl(F2::template sub<4>(t0, z4)), t0);
        z5 = F2::add(F2::dbl(F2::add(t1, z5)), t1);
        t0 = F2::template mul_xi<16>(t3);                                        // xi t3
        z2 = F2::add(F2::dbl(F2::add(t0, z2)), t0);
        z3 = F2::add(F2::dbl(F2::template sub<4>(t2, z3)), t2);
        E12 r;
        r.c0 = norm6(E6{z0, z4, z3});
        r.c1 = norm6(E6{z2, z1, z5});
        return r;
    }

    // x^|z| for x in the cyclotomic subgroup, then conjugated (z < 0): x^z.  half: x^(z/2)
    static FP_HD_NOINLINE E12 raise_to_z(const E12& x, bool half) {
        uint64_t e = half ? (fp28c::Z_ABS >> 1) : fp28c::Z_ABS;
        int top = half ? 62 : 63;
        E12 r = x;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int i = top - 1; i >= 0; i--) {
            r = cyclotomic_sqr12(r);
            if ((e >> i) & 1) r = mul12(r, x);
        }
        return conj12(r);
    }
    // f^((p^6-1)(p^2+1) * ((z-1)^2 (z+p)(z^2+p^2-1) + 3)): the hard part of Hayashida-Hayasaka-Teruya
    // (eprint 2020/875), = 3 (p^4-p^2+1)/r — the convention of blst's final_exp, hence of the reference's Gt.
    static FP_HD E12 final_exp(const E12& f) {
        E12 r = mul12(conj12(f), inv12(f));                  // f^(p^6 - 1)
        r = mul12(frob12(frob12(r)), r);                     // ^(p^2 + 1)
        E12 y0 = cyclotomic_sqr12(r);                        // 2
        E12 y1 = raise_to_z(y0, false);                      // 2z
        E12 y2 = raise_to_z(y1, true);                       // z^2
        y1 = mul12(y1, conj12(r));                           // 2z - 1
        y1 = mul12(conj12(y1), y2);                          // z^2 - 2z + 1 = (z-1)^2
        y2 = raise_to_z(y1, false);                          // z (z-1)^2
        E12 y3 = raise_to_z(y2, false);                      // z^2 (z-1)^2
        y3 = mul12(y3, conj12(y1));                          // (z^2 - 1)(z-1)^2
        y1 = frob12(frob12(frob12(y1)));                     // p^3 (z-1)^2
        y2 = frob12(frob12(y2));                             // p^2 z (z-1)^2
        y1 = mul12(y1, y2);
        y2 = raise_to_z(y3, false);                          // z (z^2-1)(z-1)^2
        y2 = mul12(mul12(y2, y0), r);                        // ... + 3
        y1 = mul12(y1, y2);
        y2 = frob12(y3);                                     // p (z^2-1)(z-1)^2
        return mul12(y1, y2);
    }
};

}  // namespace pairing
