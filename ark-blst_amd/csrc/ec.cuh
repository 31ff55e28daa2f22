// Short-Weierstrass (a = 0) group law in extended Jacobian "XYZZ" coordinates over a lazily reduced field.
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ has all-zero limbs (exact zero, never produced by a
//   regular operation).  Formulas: EFD madd-2008-s / add-2008-s / dbl-2008-s-1 / mdbl-2008-s-1.
//
// This is the bucket arithmetic of the MSM that replaces the generated POINT_multiexp kernels
// (/root/reference/build.rs:9-11, driven by /root/reference/src/gpu.rs:165-183); unlike those it is complete:
// P + P, P + (-P) and infinity operands are all handled (cf. the reference's known-bad case
// /root/reference/src/g1.rs:682-688).
//
// Value-bound invariants of a stored point (multiples of p, all coordinates N-form):
//   X < 10p, Y < 6p, ZZ < 2p, ZZZ < 2p.     (affine inputs: x < 4p, y < 4p)
// The template parameter F supplies the field: F::E element type and mul/sqr/add/sub<K>/is_zero_2p/...
#pragma once
#include "fp28.cuh"

namespace ec {

struct FpOps {
    using E = fp28::Fp;
    static FP_HD E zero() { return fp28::fp_zero(); }
    static FP_HD E one() { return fp28::fp_one(); }
    static FP_HD E mul(const E& a, const E& b) { return fp28::fp_mul(a, b); }
    static FP_HD E sqr(const E& a) { return fp28::fp_sqr(a); }
    static FP_HD E add(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static FP_HD E sub(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static FP_HD E neg(const E& a) { return fp28::fp_neg<K>(a); }
    static FP_HD bool is_zero_2p(const E& a) { return fp28::fp_is_zero_2p(a); }
    static FP_HD bool is_zero_any(const E& a) { return fp28::fp_is_zero_any(a); }
    static FP_HD E select(bool take_b, const E& a, const E& b) { return fp28::fp_select(take_b, a, b); }
    static FP_HD bool limbs_all_zero(const E& a) {
        uint32_t z = 0;
#pragma unroll
        for (int k = 0; k < fp28::NL; k++) z |= a.l[k];
        return z == 0;
    }
};

template <class F>
struct Xyzz {
    typename F::E x, y, zz, zzz;
};
template <class F>
struct Affine {
    typename F::E x, y;
};

template <class F>
FP_HD Xyzz<F> xyzz_inf() {
    Xyzz<F> r;
    r.x = F::zero(); r.y = F::zero(); r.zz = F::zero(); r.zzz = F::zero();
    return r;
}
template <class F>
FP_HD bool xyzz_is_inf(const Xyzz<F>& p) { return F::limbs_all_zero(p.zz); }

template <class F>
FP_HD Xyzz<F> xyzz_from_affine(const typename F::E& x, const typename F::E& y) {
    Xyzz<F> r;
    r.x = x; r.y = y; r.zz = F::one(); r.zzz = F::one();
    return r;
}
template <class F>
FP_HD Xyzz<F> xyzz_select(bool take_b, const Xyzz<F>& a, const Xyzz<F>& b) {
    Xyzz<F> r;
    r.x = F::select(take_b, a.x, b.x);
    r.y = F::select(take_b, a.y, b.y);
    r.zz = F::select(take_b, a.zz, b.zz);
    r.zzz = F::select(take_b, a.zzz, b.zzz);
    return r;
}

// Regular-case mixed addition acc + (x2, y2), acc not infinity.  Sets p_is_zero when the x-coordinates agree
// (P == 0 mod p): the result is then meaningless and the caller must take xyzz_madd_special().
template <class F>
FP_HD Xyzz<F> xyzz_madd_core(const Xyzz<F>& a, const typename F::E& x2, const typename F::E& y2, bool& p_is_zero) {
    using E = typename F::E;
    E U2 = F::mul(x2, a.zz);                  // < 2p
    E S2 = F::mul(y2, a.zzz);                 // < 2p
    E Pp = F::template sub<16>(U2, a.x);      // X < 10p  -> < 18p
    E Rr = F::template sub<8>(S2, a.y);       // Y < 6p   -> < 10p
    E PP = F::sqr(Pp);
    p_is_zero = F::is_zero_2p(PP);
    E PPP = F::mul(Pp, PP);
    E Q = F::mul(a.x, PP);
    E t = F::add(F::add(PPP, Q), Q);          // < 6p
    Xyzz<F> r;
    r.x = F::template sub<8>(F::sqr(Rr), t);  // < 10p
    E v = F::template sub<16>(Q, r.x);        // < 18p
    r.y = F::template sub<4>(F::mul(Rr, v), F::mul(a.y, PPP));  // < 6p
    r.zz = F::mul(a.zz, PP);
    r.zzz = F::mul(a.zzz, PPP);
    return r;
}

// Doubling of an affine point (mdbl-2008-s-1).  x, y < 4p.  y == 0 cannot happen on a prime-order subgroup;
// it degrades to ZZ == 0 (mod p), which downstream code treats as a (non-canonical) infinity only via
// xyzz_fix_inf(); callers that may see 2-torsion call that.
template <class F>
FP_HD Xyzz<F> xyzz_mdbl(const typename F::E& x, const typename F::E& y) {
    using E = typename F::E;
    E U = F::add(y, y);                        // < 8p
    E V = F::sqr(U);
    E Wq = F::mul(U, V);
    E S = F::mul(x, V);
    E xx = F::sqr(x);
    E M = F::add(F::add(xx, xx), xx);          // < 6p
    Xyzz<F> r;
    r.x = F::template sub<8>(F::sqr(M), F::add(S, S));   // < 10p
    E v = F::template sub<16>(S, r.x);                     // < 18p
    r.y = F::template sub<4>(F::mul(M, v), F::mul(Wq, y));  // < 6p
    r.zz = V;
    r.zzz = Wq;
    return r;
}

// Slow path of the mixed addition, taken when P == 0: either the same point (double) or opposite points (inf).
template <class F>
FP_HD Xyzz<F> xyzz_madd_special(const Xyzz<F>& a, const typename F::E& x2, const typename F::E& y2) {
    using E = typename F::E;
    E S2 = F::mul(y2, a.zzz);
    E Rr = F::template sub<8>(S2, a.y);
    if (F::is_zero_any(Rr)) return xyzz_mdbl<F>(x2, y2);
    return xyzz_inf<F>();
}

// Doubling of an XYZZ point (dbl-2008-s-1), a not infinity.
template <class F>
FP_HD Xyzz<F> xyzz_dbl(const Xyzz<F>& a) {
    using E = typename F::E;
    E U = F::add(a.y, a.y);                    // < 12p
    E V = F::sqr(U);
    E Wq = F::mul(U, V);
    E S = F::mul(a.x, V);
    E xx = F::sqr(a.x);
    E M = F::add(F::add(xx, xx), xx);          // < 6p
    Xyzz<F> r;
    r.x = F::template sub<8>(F::sqr(M), F::add(S, S));
    E v = F::template sub<16>(S, r.x);
    r.y = F::template sub<4>(F::mul(M, v), F::mul(Wq, a.y));
    r.zz = F::mul(V, a.zz);
    r.zzz = F::mul(Wq, a.zzz);
    return r;
}

// Complete addition of two XYZZ points (add-2008-s + the exceptional cases).
template <class F>
FP_HD Xyzz<F> xyzz_add(const Xyzz<F>& a, const Xyzz<F>& b) {
    using E = typename F::E;
    bool ainf = xyzz_is_inf(a), binf = xyzz_is_inf(b);
    E U1 = F::mul(a.x, b.zz);
    E U2 = F::mul(b.x, a.zz);
    E S1 = F::mul(a.y, b.zzz);
    E S2 = F::mul(b.y, a.zzz);
    E Pp = F::template sub<4>(U2, U1);         // < 6p
    E Rr = F::template sub<4>(S2, S1);         // < 6p
    E PP = F::sqr(Pp);
    bool pz = F::is_zero_2p(PP);
    E PPP = F::mul(Pp, PP);
    E Q = F::mul(U1, PP);
    E t = F::add(F::add(PPP, Q), Q);           // < 6p
    Xyzz<F> r;
    r.x = F::template sub<8>(F::sqr(Rr), t);   // < 10p
    E v = F::template sub<16>(Q, r.x);         // < 18p
    r.y = F::template sub<4>(F::mul(Rr, v), F::mul(S1, PPP));  // < 6p
    r.zz = F::mul(F::mul(a.zz, b.zz), PP);
    r.zzz = F::mul(F::mul(a.zzz, b.zzz), PPP);
    if (!ainf && !binf && pz) {  // rare: same x
        if (F::is_zero_any(Rr)) r = xyzz_dbl<F>(a);
        else r = xyzz_inf<F>();
    }
    r = xyzz_select<F>(ainf, r, b);
    r = xyzz_select<F>(binf && !ainf, r, a);
    return r;
}

// k doublings (k small), infinity-safe
template <class F>
FP_HD Xyzz<F> xyzz_dbl_n(Xyzz<F> a, int k) {
    bool inf = xyzz_is_inf(a);
    for (int i = 0; i < k; i++) a = xyzz_dbl<F>(a);
    return xyzz_select<F>(inf, a, xyzz_inf<F>());
}

}  // namespace ec
