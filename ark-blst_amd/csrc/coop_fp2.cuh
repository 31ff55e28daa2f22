// Fp2 arithmetic split over a lane pair (used by the pairing line kernel and the G2 bucket reduction).
#pragma once
#include <hip/hip_runtime.h>
#include "ec.cuh"

namespace msmk {

using fp28::Fp;

// Fp2 arithmetic split over a LANE PAIR: the even lane holds c0 and the odd lane c1 of every Fp2 value; a product
// exchanges the partner's components by DPP (quad_perm [1,0,3,2]) and each lane does ONE fused two-product reduction:
//   even: a0 b0 + a1 (32p - b1)        odd: a0 b1 + a1 b0
// Same static interface as pairing::PF2, so the generic line functions of pairing.cuh run on it unchanged (the value
// bounds checked by tests/host/pairing_bounds.cpp are per component and carry over).  Halves the registers and the
// dependent chain per lane.  The multiplier is inlined here: through the shared out-of-line bodies the values that live
// across the calls spill (measured 12.5 vs 14.5 ms for the whole Miller phase at 2^16 pairs).
struct CoopF2 {
    using E = Fp;
    using Fp = fp28::Fp;
    static __device__ __forceinline__ bool hi() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ Fp partner(const Fp& a) {
        Fp r;
#pragma unroll
        for (int k = 0; k < fp28::NL; k++) r.l[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[k], 0xB1, 0xF, 0xF, true);
        return r;
    }
    static __device__ __forceinline__ E zero() { return fp28::fp_zero(); }
    static __device__ __forceinline__ E one() { return fp28::fp_select(hi(), fp28::fp_one(), fp28::fp_zero()); }
    // 32p - b without a carry pass (b in N-form, < 31p): limbs < 2^29, still a legal multiplier operand next to N-form
    // partners (two products + the reduction stay below 2^62 per column; four below 2^62.6)
    static __device__ __forceinline__ Fp neg32_lazy(const Fp& b) { return fp28::fp_sub_lazy<32>(fp28::fp_zero(), b); }
    static __device__ __forceinline__ E mul(const E& a, const E& b) {
        // own a times s1 plus partner's a times s2:  even lane  a0 b0 + a1 (32p - b1),  odd lane  a1 b0 + a0 b1
        Fp pa = partner(a), pb = partner(b);
        Fp s1 = fp28::fp_select(hi(), b, pb);
        Fp s2 = fp28::fp_select(hi(), neg32_lazy(pb), b);
        return fp28::fp_mul2add(a, s1, pa, s2);
    }
    static __device__ __forceinline__ E sqr(const E& a) {          // (a0 + a1)(a0 - a1) | (2 a0) a1
        Fp pa = partner(a);
        Fp u = fp28::fp_select(hi(), fp28::fp_add(a, pa), fp28::fp_add(pa, pa));
        Fp v = fp28::fp_select(hi(), fp28::fp_sub<32>(a, pa), a);
        return fp28::fp_mul(u, v);
    }
    // s^2 - 12 e^2 as ONE fused two-product reduction per lane (see ec::Fp2OpsT::sqr_sub12sqr): 588 multiply-adds where two
    // squarings and the subtraction's normalisation would take 784 + a multiplication by one
    static __device__ __forceinline__ E sqr_sub12sqr(const E& s, const E& e) {
        Fp ps = partner(s), pe = partner(e);
        Fp u = fp28::fp_select(hi(), fp28::fp_add(s, ps), fp28::fp_add(ps, ps));               // s0 + s1 | 2 s0
        Fp v = fp28::fp_select(hi(), fp28::fp_sub<32>(s, ps), s);                               // s0 - s1 | s1
        Fp ue = fp28::fp_select(hi(), fp28::fp_add(e, pe), fp28::fp_add(pe, pe));              // e0 + e1 | 2 e0
        Fp we = fp28::fp_mul_small<12>(fp28::fp_select(hi(), fp28::fp_sub<4>(pe, e), fp28::fp_neg<4>(e)));   // 12 (e1 - e0) | -12 e1
        return fp28::fp_mul2add(u, v, ue, we);
    }
    static __device__ __forceinline__ E mul2add(const E& a, const E& b, const E& c, const E& d) { return fp28::fp_add(mul(a, b), mul(c, d)); }
    static __device__ __forceinline__ E add(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static __device__ __forceinline__ E sub(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static __device__ __forceinline__ E neg(const E& a) { return fp28::fp_neg<K>(a); }
    static __device__ __forceinline__ E mul3(const E& a) { return fp28::fp_mul_small<3>(a); }
    template <int K>
    static __device__ __forceinline__ E mul_small(const E& a) { return fp28::fp_mul_small<K>(a); }
    static __device__ __forceinline__ E reduce_small(const E& a) { return fp28::fp_reduce_small(a); }   // component-wise (ec.cuh jac_dbl)
    // b3 a, b3 = 12 (1 + u):  12 (a0 - a1) + 12 (a0 + a1) u — one linear combination with the partner and ONE single-product
    // multiplication by the constant 12 per lane (406 multiply-adds instead of the 602 of a full Fp2 product); a <= 3p, result < 2p
    static __device__ __forceinline__ E mul_b3(const E& a) {
        Fp pa = partner(a);
        Fp lin = fp28::fp_select(hi(), fp28::fp_sub<4>(a, pa), fp28::fp_add(a, pa));
        return fp28::fp_mul(lin, fp28::fp_const(fp28::TWELVE));
    }
    static __device__ __forceinline__ E mul_b3_red(const E& a) { return mul_b3(a); }   // already a field product: < 2p (ec::proj_dbl asks for it)
    static __device__ __forceinline__ E mul_fp(const E& a, const Fp& s) { return fp28::fp_mul(a, s); }
    static __device__ __forceinline__ E norm2(const E& a) { return fp28::fp_mul(a, fp28::fp_one()); }
    static __device__ __forceinline__ E dbl(const E& a) { return fp28::fp_add(a, a); }
    static __device__ __forceinline__ Fp fp_neg4(const Fp& a) { return fp28::fp_neg<4>(a); }
    static __device__ __forceinline__ E select(bool take_b, const E& a, const E& b) { return fp28::fp_select(take_b, a, b); }
};

// The same lane-pair field with the hooks ec::xyzz_madd needs, for the G2 accumulate hot loop: every value is ONE Fp per lane
// (half the registers of an Fp2 per lane: the one-lane-per-item G2 loop kept 1.3 KB of scratch per lane), the two-product
// sum is one fused four-product reduction per lane, zero tests are pair-wide.  The linear hooks are the normalising ones, as
// in ec::Fp2OpsT, so the value bounds machine-checked for G2 (tests/host/msm_bounds.cpp) carry over component-wise.
struct CoopF2A : CoopF2 {
    static __device__ __forceinline__ bool pair_and(bool v) {
        int mine = v ? 1 : 0;
        return (mine & __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xF, 0xF, true)) != 0;
    }
    static __device__ __forceinline__ E mul2add(const E& a, const E& b, const E& c, const E& d) {   // a b + c d ; b, d <= 31p
        Fp pa = partner(a), pb = partner(b), pc = partner(c), pd = partner(d);
        Fp b1 = fp28::fp_select(hi(), b, pb), b2 = fp28::fp_select(hi(), neg32_lazy(pb), b);
        Fp d1 = fp28::fp_select(hi(), d, pd), d2 = fp28::fp_select(hi(), neg32_lazy(pd), d);
        return fp28::fp_mul4add(a, b1, pa, b2, c, d1, pc, d2);      // even: a0 b0 - a1 b1 + c0 d0 - c1 d1 ; odd: a1 b0 + a0 b1 + c1 d0 + c0 d1
    }
    static __device__ __forceinline__ E add_l(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static __device__ __forceinline__ E sub_l(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static __device__ __forceinline__ E neg_l(const E& a) { return fp28::fp_neg<K>(a); }
    static __device__ __forceinline__ E sub8_wide(const E& a, const E& b) { return fp28::fp_sub<8>(a, b); }
    static __device__ __forceinline__ E norm(const E& a) { return a; }
    static __device__ __forceinline__ bool is_zero_2p(const E& a) {   // a: a multiplier output (exact limbs)
        const bool maybe = a.l[0] == 0 || a.l[0] == fp28::P[0];        // this lane's component could be 0 or p
        if (!pair_and(maybe)) return false;                              // pair-wide and practically always the answer
        return pair_and(fp28::fp_is_zero_2p(a));
    }
    static __device__ __forceinline__ bool limbs_all_zero(const E& a) { return pair_and(ec::FpOps::limbs_all_zero(a)); }
};

}  // namespace msmk
