// BLS12-381 base-field arithmetic for gfx950 (MI355X): radix 2^28, 14 limbs, Montgomery R' = 2^392.
//
// Why this shape (measured with tools/ubench_valu.hip on MI355X):
//   v_mad_u64_u32 issues at HALF rate (~4.7 cyc/wave) — the same cost as ONE v_add_co/v_addc — so the
//   cheapest 384-bit multiply is the one with the fewest carry instructions, not the fewest products.
//   28-bit limbs leave 8 bits of headroom in a 64-bit column: 14 a*b products + 14 m*p products
//   (each < 2^56) accumulate with plain v_mad_u64_u32 and NO carry handling; add/sub are 14 carry-less
//   v_add_u32; modular reduction is lazy (R' = 2^392 leaves 2^11 of slack over p).
//
// Replaces: the ec-gpu-gen generated FIELD_mul/add/sub templates that /root/reference/build.rs:9-11
// instantiates for blstrs::Fp (Montgomery, 12 x u32 limbs on CUDA), driven from /root/reference/src/gpu.rs.
// Field constants: /root/reference/src/fp.rs:25-32 (modulus), src/gpu.rs:253-273 (one / r2 / modulus).
//
// Representation invariants
//   N-form  : limbs l[0..12] <= 2^28 + 15 (what fp_norm1 leaves), l[13] small; value = sum l[k] 2^(28k)   (not canonical)
//   value   : every function documents the bound (multiples of p) it needs / produces.
//   Montgomery multiplication accepts any a, b with a*b < 2^392 * p (i.e. a, b < ~50p) and returns < 2p.
#pragma once
#include <cstdint>
#include <type_traits>
#include "fp28_consts.h"

#if defined(__HIPCC__)
#define FP_HD __host__ __device__ __forceinline__
#define FP_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define FP_HD inline
#define FP_HD_NOINLINE inline
#endif

namespace fp28 {

using namespace fp28c;

struct Fp {
    uint32_t l[NL];
};

FP_HD Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = 0;
    return r;
}
FP_HD Fp fp_const(const uint32_t (&c)[NL]) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = c[k];
    return r;
}
FP_HD Fp fp_one() { return fp_const(ONE); }

// One parallel carry pass: any limbs < 2^32  ->  N-form (l[k] <= 2^28 - 1 + 15 < SPREAD_LO = 2^28 + 64).
FP_HD void fp_norm1(Fp& r) {
    uint32_t cprev = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        uint32_t c = r.l[k] >> W;
        r.l[k] = (r.l[k] & MASK) + cprev;
        cprev = c;
    }
    r.l[NL - 1] += cprev;
}

// r = a + b   (value bound adds)
FP_HD Fp fp_add(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = a.l[k] + b.l[k];
    fp_norm1(r);
    return r;
}

// r = a - b + K*p, K in {2,4,8,16,32,64}; requires b in N-form with value < (K-1)p. Result < a + K*p.
template <int K>
FP_HD Fp fp_sub(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t s = K == 2 ? S2[k] : K == 4 ? S4[k] : K == 8 ? S8[k] : K == 16 ? S16[k] : K == 32 ? S32[k] : S64[k];
        r.l[k] = a.l[k] + s - b.l[k];
    }
    fp_norm1(r);
    return r;
}

// Lazy variants for the accumulate hot loop: NO carry pass.  a + b: limbs add up.  a - b + K p: b must have limbs
// <= SPREAD_LO (exact / N-form) and value < (K-1) p; result limbs < a + 2^29 + 68.  Callers track limb sizes
// (tools/bounds_check.py): a multiplication takes operands with 14 * la * lb < 2^64 - 2^60.
FP_HD Fp fp_add_lazy(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = a.l[k] + b.l[k];
    return r;
}
template <int K>
FP_HD Fp fp_sub_lazy(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t s = K == 2 ? S2[k] : K == 4 ? S4[k] : K == 8 ? S8[k] : K == 16 ? S16[k] : K == 32 ? S32[k] : S64[k];
        r.l[k] = a.l[k] + s - b.l[k];
    }
    return r;
}
// a - b + 8p where b is a lazy sum of up to three exact values (limbs < 3 * 2^28): spread constant S8B has every low
// limb >= 3 * 2^28 + 64
FP_HD Fp fp_sub8_lazy_wide(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = a.l[k] + S8B[k] - b.l[k];
    return r;
}
FP_HD Fp fp_norm(const Fp& a) {
    Fp r = a;
    fp_norm1(r);
    return r;
}

// r = K*p - b
template <int K>
FP_HD Fp fp_neg(const Fp& b) {
    Fp z = fp_zero();
    return fp_sub<K>(z, b);
}

// r = K*a for a small constant K (K * limb must stay below 2^32: K <= 15)
template <int K>
FP_HD Fp fp_mul_small(const Fp& a) {
    static_assert(K >= 1 && K <= 15, "small multiplier");
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = a.l[k] * (uint32_t)K;
    fp_norm1(r);
    return r;
}

// v + 2p - (q + 1) p with q = floor(l[13] * 40323 / 2^32), for an N-form value v < 127p: result N-form, p <= result < 2.01p.  About 100
// plain instructions where a multiplication by the internal one (the other way below 2p) is 406 multiply-adds: what lets the Jacobian
// doubling of the G1 subgroup ladder (ec.cuh jac_dbl) subtract 8 X B and stay inside the multiplier's contract.
//   l[13] <= v / 2^364 < l[13] + 1.01 (N-form low limbs), 106513 < p / 2^364 < 106514, 40323 / 2^32 = 1 / 106514.82: q <= v / p and
//   v / p - q < 1 + 1.36e7 * 1.6e-10 + 1e-5 < 1.003.  (q + 1) p is produced with EXACT limbs (one 64-bit multiply-add per limb, the carry
//   is the addend of the next), so the subtraction is fp_sub<2>'s: the spread 2p keeps every low limb non-negative, and its top limb
//   (2 * 106513 - 1) covers the top limb of (q + 1) p, which exceeds v's by at most 1 + 106514.
FP_HD Fp fp_reduce_small(const Fp& v) {
    const uint32_t q1 = (uint32_t)(((uint64_t)v.l[NL - 1] * 40323u) >> 32) + 1u;   // <= 128
    Fp qp;
    uint64_t t = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        t = (uint64_t)q1 * P[k] + (t >> W);
        qp.l[k] = k < NL - 1 ? ((uint32_t)t & MASK) : (uint32_t)t;
    }
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = v.l[k] + S2[k] - qp.l[k];
    fp_norm1(r);
    return r;
}

// lane-wise select without branches
FP_HD Fp fp_select(bool take_b, const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = take_b ? b.l[k] : a.l[k];
    return r;
}

// ------------------------------------------------------------------------------------------------
// Montgomery multiplication, two forms with the same multiply-adds and the same column sums (bit-identical results):
//
//   product scanning (FIPS; fp_mul, fp_sqr, fp_mul2add, fp_mul4add — the default): the columns of the product are produced one
//     after the other in ONE 64-bit accumulator, the reduction's m_i follows from column i the moment it is complete and the carry
//     into the next column is a shift of the accumulator itself.  2 live registers of column state instead of 56: the G2
//     accumulate loop spilled 35 registers with the wide form (2.8 GB of scratch writes per launch at 2^20 points), it runs
//     without scratch and 5 % faster with this one; the lane-pair and lone-wave kernels are unchanged or slightly faster.
//   operand scanning (fp_mul_os, fp_sqr_os, fp_mul2add_os): 28 64-bit columns live, every product row and every reduction row
//     a run of independent multiply-adds.  Kept for the G1 accumulate hot loop only, where it measures 2 % faster (same-box A/B,
//     2^20 points: 2.34 vs 2.39 ms) — that loop has the registers, and the compiler interleaves the rows of neighbouring
//     multiplications.
//
// Measured and not kept (round 3, profiles/r03_ab_multiplier.txt): the product-scanning chain written with inline-assembly
// v_mad_u64_u32 whose addend IS the accumulator.  From plain C++ the compiler re-associates every column into (fresh chain
// from zero) + (carry of the previous column) and spends a v_lshl_add_u64 per column on the last addition (235 per mixed
// addition); the assembly form removes those, but the hazard recogniser then puts an s_nop behind every inline-assembly
// instruction whose result is consumed next (it must assume a dst_sel forwarding hazard): 3378 s_nop per mixed addition, the
// same 2.33 ms for the two-wave accumulate kernel and 4-30 % SLOWER lone-wave kernels (reduce, combine, G2 reduce).
// ------------------------------------------------------------------------------------------------

// Montgomery reduction of 28 64-bit columns (each < 2^63) holding a product < 2^392 * p.  Returns exact limbs, < 2p.
FP_HD Fp fp_mont_reduce(uint64_t (&c)[2 * NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t m = ((uint32_t)c[i] * PINV) & MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)m * P[j];
        c[i + 1] += c[i] >> W;  // low 28 bits of c[i] are now zero
    }
    // columns c[14..27] -> EXACT 28-bit limbs by one sequential 64-bit carry chain (3 instructions per limb; measured
    // 8 % faster on the accumulate kernel than splitting every column into three pieces plus a parallel carry pass)
    Fp r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint64_t v = c[NL + k] + carry;
        r.l[k] = (uint32_t)v & MASK;
        carry = v >> W;
    }
    r.l[NL - 1] |= (uint32_t)carry << W;  // value < 2p < 2^382: nothing is carried out of limb 13
    return r;
}

template <int I, int N, class Fn>
FP_HD void fp_static_for(Fn&& f) {   // f(integral_constant<int, I>) for I .. N-1: every index below is a compile-time constant
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fp_static_for<I + 1, N>(f);
    }
}
// col(K, acc) adds every product term of column K (K = 0 .. 26) into acc; terms + 14 reduction terms + carry stay below 2^63.
template <class ColFn>
FP_HD Fp fp_fips(ColFn col) {
    uint32_t m[NL];
    uint64_t acc = 0;
    Fp r;
    fp_static_for<0, 2 * NL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 2 * NL - 1) col(ic, acc);
        constexpr int j0 = i < NL ? 0 : i - NL + 1, j1 = i < NL ? i : NL;   // reduction terms m_j P_{i-j}, j in [j0, j1)
#pragma unroll
        for (int j = j0; j < j1; j++) acc += (uint64_t)m[j] * P[i - j];
        if constexpr (i < NL) {
            m[i] = ((uint32_t)acc * PINV) & MASK;
            acc += (uint64_t)m[i] * P[0];   // the low 28 bits are zero now
        } else {
            r.l[i - NL] = (uint32_t)acc & MASK;
        }
        acc >>= W;
    });
    r.l[NL - 1] |= (uint32_t)acc << W;  // value < 2p < 2^382: nothing is carried out of limb 13
    return r;
}
// the terms a_i b_j, i + j = K, of one product
template <int K>
FP_HD void fp_col(uint64_t& acc, const Fp& a, const Fp& b) {
    constexpr int i0 = K < NL ? 0 : K - NL + 1, i1 = K < NL ? K + 1 : NL;
#pragma unroll
    for (int i = i0; i < i1; i++) acc += (uint64_t)a.l[i] * b.l[K - i];
}

// r = a*b / 2^392 mod p.  a, b in N-form (limbs < 2^29.5 are still safe), a*b < 2^392 p.  Result N-form, < 2p.
FP_HD Fp fp_mul(const Fp& a, const Fp& b) {
    return fp_fips([&](auto kc, uint64_t& acc) { fp_col<decltype(kc)::value>(acc, a, b); });
}
FP_HD Fp fp_mul_os(const Fp& a, const Fp& b) {
    uint64_t c[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    return fp_mont_reduce(c);
}

// r = (a*b + c*d) / 2^392 mod p with ONE Montgomery reduction (the two products share the 64-bit columns:
// 28 + 14 terms of < 2^56 stay below 2^62).  Needs a*b + c*d < 2^392 p.  Saves a 210-MAD reduction per use.
FP_HD Fp fp_mul2add(const Fp& a, const Fp& b, const Fp& c2, const Fp& d) {
    return fp_fips([&](auto kc, uint64_t& acc) {
        constexpr int k = decltype(kc)::value;
        fp_col<k>(acc, a, b);
        fp_col<k>(acc, c2, d);
    });
}
FP_HD Fp fp_mul2add_os(const Fp& a, const Fp& b, const Fp& c2, const Fp& d) {
    uint64_t c[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)c2.l[i] * d.l[j];
    }
    return fp_mont_reduce(c);
}

// r = (a*b + c*d + e*f + g*h) / 2^392 mod p, one reduction (Fp2 fused multiply-add): 56 + 14 terms < 2^62.2.
FP_HD Fp fp_mul4add(const Fp& a, const Fp& b, const Fp& c2, const Fp& d, const Fp& e, const Fp& f, const Fp& g, const Fp& h) {
    return fp_fips([&](auto kc, uint64_t& acc) {
        constexpr int k = decltype(kc)::value;
        fp_col<k>(acc, a, b);
        fp_col<k>(acc, c2, d);
        fp_col<k>(acc, e, f);
        fp_col<k>(acc, g, h);
    });
}

// The shared multiplier instance.  The out-of-line bodies keep the operand-scanning form (register pressure is no concern inside a
// function of its own).  Round 3 saw the test-only single-lane Miller kernel disagree with the production path when this body was the
// product-scanning one; round 4 pinned that on a code-generation defect of the compiler that has nothing to do with this function's body
// (the same kernel shape fails with the operand-scanning body too; tools/call_abi/README.md) and fenced it off in the build
// (csrc/Makefile: VGPR spill slots are not turned into AGPRs, kernels with calls are capped at 256 registers).
// On the device this is a REAL function: the first operand travels in v0..v13, the second by reference to a scratch copy (the AMDGPU
// ABI passes at most 16 registers of aggregates directly), the result in v0..v13: one ~4.5 KB body per kernel instead of one per
// use keeps bucket kernels inside the 64 KB instruction cache (see ec.cuh).  On the host it is plain inline code.
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(MI_CALL_PS)   // reproducer switch (tools/call_abi/): the product-scanning body behind the by-value signature
static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul(a, b); }
#else
static __device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fp_mul_os(a, b); }
#endif
#else
FP_HD_NOINLINE Fp fp_mul_call(const Fp& a, const Fp& b) { return fp_mul_os(a, b); }
#endif

// r = a^2 / 2^392 mod p (105 products instead of 196)
FP_HD Fp fp_sqr(const Fp& a) {
    uint32_t a2[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) a2[k] = a.l[k] << 1;
    return fp_fips([&](auto kc, uint64_t& acc) {
        constexpr int k = decltype(kc)::value;
        constexpr int i0 = k < NL ? 0 : k - NL + 1;   // pairs i < j = k - i, j < NL; the square a_{k/2}^2 for even k
#pragma unroll
        for (int i = i0; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
        if constexpr (k % 2 == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
    });
}
FP_HD Fp fp_sqr_os(const Fp& a) {
    uint32_t a2[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) a2[k] = a.l[k] << 1;
    uint64_t c[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        c[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < NL; j++) c[i + j] += (uint64_t)a.l[i] * a2[j];
    }
    return fp_mont_reduce(c);
}

// shared fused two-product instance (Fp2 arithmetic outside the hot loop)
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __noinline__ Fp fp_mul2add_call(Fp a, Fp b, Fp c, Fp d) { return fp_mul2add_os(a, b, c, d); }
#else
FP_HD_NOINLINE Fp fp_mul2add_call(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return fp_mul2add_os(a, b, c, d); }
#endif

// shared squaring instance (see fp_mul_call)
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __noinline__ Fp fp_sqr_call(Fp a) { return fp_sqr_os(a); }
#else
FP_HD_NOINLINE Fp fp_sqr_call(const Fp& a) { return fp_sqr_os(a); }
#endif

// Exact carry propagation of an N-form value that fits 392 bits: limbs -> [0, 2^28), l[13] holds the rest.
FP_HD void fp_carry_exact(Fp& r) {
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
        uint32_t v = r.l[k] + carry;
        r.l[k] = v & MASK;
        carry = v >> W;
    }
    r.l[NL - 1] += carry;
}

// a == 0 (mod p) for an N-form value < 2p (any Montgomery product).
FP_HD bool fp_is_zero_2p(const Fp& a) {
    Fp t = a;
    fp_carry_exact(t);
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        z |= t.l[k];
        e |= t.l[k] ^ P[k];
    }
    return z == 0 || e == 0;
}

// The same for a value with EXACT limbs (every multiplier output): 0 has l[0] = 0 and p has l[0] = P[0] (non-zero), so one look at the
// low limb settles all but 2^-27 of the cases; the full comparison runs behind a branch that is practically never taken.  The
// accumulate hot loop asks this once per mixed addition: ~70 instructions less per iteration than the unconditional form.
FP_HD bool fp_is_zero_2p_exact(const Fp& a) {
    if (a.l[0] != 0 && a.l[0] != P[0]) return false;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        z |= a.l[k];
        e |= a.l[k] ^ P[k];
    }
    return z == 0 || e == 0;
}

// a == 0 (mod p) for any N-form value < ~50p: one multiplication by the internal one brings it below 2p.
FP_HD bool fp_is_zero_any(const Fp& a) { return fp_is_zero_2p(fp_mul_call(a, fp_one())); }

// Canonical representative in [0, p) with exact 28-bit limbs, for an N-form value < 2p.
FP_HD Fp fp_canon_2p(const Fp& a) {
    Fp t = a;
    fp_carry_exact(t);
    Fp d;
    uint32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint32_t v = t.l[k] - P[k] - borrow;
        borrow = v >> 31;
        d.l[k] = v & MASK;
    }
    return fp_select(borrow != 0, d, t);  // borrow => t < p, keep t
}

// ------------------------------------------------------------------------------------------------
// Conversions to / from the reference's in-memory form: blst_fp = 12 x u32 LE words of x*2^384 mod p
// (the raw bytes /root/reference/src/gpu.rs:149 uploads and :185-186 downloads).
// ------------------------------------------------------------------------------------------------
FP_HD Fp fp_unpack384(const uint32_t (&w)[12]) {  // plain re-limbing, no domain change
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        int bit = W * k, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = w[wi] >> sh;
        if (sh > 4 && wi + 1 < 12) lo |= w[wi + 1] << (32 - sh);
        r.l[k] = lo & MASK;
    }
    return r;
}
FP_HD void fp_pack384(uint32_t (&w)[12], const Fp& a) {  // a: exact limbs, value < 2^384
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        int bit = W * k, wi = bit >> 5, sh = bit & 31;
        if (wi < 12) w[wi] |= a.l[k] << sh;
        if (sh > 4 && wi + 1 < 12) w[wi + 1] |= a.l[k] >> (32 - sh);
    }
}
// blst Montgomery words (x*2^384 mod p)  ->  internal (x*2^392 mod p), N-form < 2p
FP_HD Fp fp_from_blst(const uint32_t (&w)[12]) { return fp_mul_call(fp_unpack384(w), fp_const(C_IN)); }
// internal (any value < ~50p) -> blst Montgomery words, canonical
FP_HD void fp_to_blst(uint32_t (&w)[12], const Fp& a) { fp_pack384(w, fp_canon_2p(fp_mul_call(a, fp_const(C_OUT)))); }

}  // namespace fp28
