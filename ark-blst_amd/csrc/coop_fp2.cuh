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
    static __device__ __forceinline__ E mul(const E& a, const E& b) {
        Fp pa = partner(a), pb = partner(b);
        Fp x = fp28::fp_select(hi(), a, pa);                       // a0
        Fp z = fp28::fp_select(hi(), pa, a);                       // a1
        Fp w = fp28::fp_select(hi(), fp28::fp_neg<32>(pb), pb);    // even: 32p - b1, odd: b0
        return fp28::fp_mul2add(x, b, z, w);
    }
    static __device__ __forceinline__ E sqr(const E& a) {          // (a0 + a1)(a0 - a1) | (2 a0) a1
        Fp pa = partner(a);
        Fp u = fp28::fp_select(hi(), fp28::fp_add(a, pa), fp28::fp_add(pa, pa));
        Fp v = fp28::fp_select(hi(), fp28::fp_sub<32>(a, pa), a);
        return fp28::fp_mul(u, v);
    }
    static __device__ __forceinline__ E mul2add(const E& a, const E& b, const E& c, const E& d) { return fp28::fp_add(mul(a, b), mul(c, d)); }
    static __device__ __forceinline__ E add(const E& a, const E& b) { return fp28::fp_add(a, b); }
    template <int K>
    static __device__ __forceinline__ E sub(const E& a, const E& b) { return fp28::fp_sub<K>(a, b); }
    template <int K>
    static __device__ __forceinline__ E neg(const E& a) { return fp28::fp_neg<K>(a); }
    static __device__ __forceinline__ E mul3(const E& a) { return fp28::fp_mul_small<3>(a); }
    static __device__ __forceinline__ E mul_b3(const E& a) { return mul(a, fp28::fp_const(fp28::TWELVE)); }   // b3 = 12 + 12u
    static __device__ __forceinline__ E mul_fp(const E& a, const Fp& s) { return fp28::fp_mul(a, s); }
    static __device__ __forceinline__ E norm2(const E& a) { return fp28::fp_mul(a, fp28::fp_one()); }
    static __device__ __forceinline__ E dbl(const E& a) { return fp28::fp_add(a, a); }
    static __device__ __forceinline__ Fp fp_neg4(const Fp& a) { return fp28::fp_neg<4>(a); }
    static __device__ __forceinline__ E select(bool take_b, const E& a, const E& b) { return fp28::fp_select(take_b, a, b); }
};

}  // namespace msmk
