// HIP kernels of the Pippenger MSM pipeline for gfx950 (MI355X).  Hand-written; no MFMA (384-bit carry-chain
// integer work), no hipify, no CUDA dual paths.
//
// Replaces the single generated kernel `POINT_multiexp` the reference launches at /root/reference/src/gpu.rs:172-183
// (one thread = one (base-group, window) with 2^w private Jacobian buckets in global memory, unsigned digits,
// host-side fold of ~32k partials at gpu.rs:193-209) with a sort-based pipeline:
//
//   k_ingest<C>   bases: blst_p{1,2}_affine (R = 2^384)  ->  device form (14 x 28-bit limbs per Fp, R' = 2^392)
//   k_coarse<0/1>, k_colscan, k_binscan, k_fine_sort: scalars -> signed c-bit digits (NEGATION_IS_CHEAP,
//                 src/g1.rs:595) -> two-level LDS-staged bucket sort -> sorted (index|sign) entries + histogram
//   k_sched1-3    prefix sums of the histogram -> bucket offsets, work items (heavy buckets split), length-sorted order
//   k_accumulate  one lane per work item (bucket, chunk<=T): XYZZ mixed additions over its run   <- dominant kernel
//                 (exceptional pairs finish on the complete projective formulas); bucket stored projective
//   k_merge       (only if a bucket was split) binary-tree merge of a bucket's partial sums
//   k_reduce      one wave per 64*L consecutive buckets: lane-serial running sums + wavefront suffix scan
//                 -> (S, T) = (sum B_b, sum (b-b0+1) B_b) per chunk, written as blst_p1 Jacobian
//   host          per-window chunk combine + Horner fold (hostec), as the reference folds on the host too.
#pragma once
#include <hip/hip_runtime.h>
#include "ec.cuh"
#include "coop_fp2.cuh"

namespace msmk {

using fp28::Fp;
using fp28::NL;

// ---------------------------------------------------------------------------------------------- layouts
// Every field element occupies a 16-word (64 B) aligned slot per Fp component (14 limbs + 2 spare words), so it moves
// as 4 x dwordx4.  Curve descriptor C:
//   G1: affine point = 2 slots  (128 B: x, y; word 31 = infinity flag), projective bucket = 3 slots (192 B)
//   G2: affine point = 4 slots  (256 B: x.c0, x.c1, y.c0, y.c1; word 63 = flag),       bucket = 6 slots (384 B)
// Raw (reference) forms: blst_p1_affine 24 words, blst_p1 36 words; blst_p2_affine 48 words, blst_p2 72 words.
__device__ __forceinline__ void load_fp16(Fp& r, const uint32_t* p) {  // 16-word aligned slot, 14 used
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
}
__device__ __forceinline__ void store_fp16(uint32_t* p, const Fp& r, uint32_t w14 = 0, uint32_t w15 = 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
    q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
    q[2] = make_uint4(r.l[8], r.l[9], r.l[10], r.l[11]);
    q[3] = make_uint4(r.l[12], r.l[13], w14, w15);
}
__device__ __forceinline__ Fp shfl_down_fp(const Fp& a, int d) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = __shfl_down(a.l[k], d, 64);
    return r;
}
__device__ __forceinline__ void fp_from_raw(Fp& r, const uint32_t* raw) {  // 12 raw words -> device form
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 12; k++) w[k] = raw[k];
    r = fp28::fp_from_blst(w);
}
__device__ __forceinline__ uint32_t fp_to_raw(uint32_t* out, const Fp& a, bool keep) {  // returns OR of the words
    uint32_t w[12], any = 0;
    fp28::fp_to_blst(w, a);
#pragma unroll
    for (int k = 0; k < 12; k++) { out[k] = keep ? w[k] : 0u; any |= w[k]; }
    return any;
}

// The reference's words WITHOUT the change of Montgomery radix, for values that meet an internal-form factor exactly once: the product of
// x 2^384 (the caller's form, re-limbed) and w 2^392 (internal) under the internal multiplication is (x w) 2^384 — the caller's form again.
// normalize_batch's x = X / Z^2, y = Y / Z^3 are such products: four conversions (multiplications) per point less (round 6).
__device__ __forceinline__ void fp_unpack_raw(Fp& r, const uint32_t* raw) {
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 12; k++) w[k] = raw[k];
    r = fp28::fp_unpack384(w);
}
__device__ __forceinline__ void fp_pack_raw(uint32_t* out, const Fp& a, bool keep) {   // a < 2p (a multiplier output)
    uint32_t w[12];
    fp28::fp_pack384(w, fp28::fp_canon_2p(a));
#pragma unroll
    for (int k = 0; k < 12; k++) out[k] = keep ? w[k] : 0u;
}

// element I/O, generic over Fp / Fp2
template <class E> struct ElemIO;
template <> struct ElemIO<Fp> {
    static constexpr int SLOT = 16, RAW = 12;
    static __device__ __forceinline__ void load(Fp& r, const uint32_t* p) { load_fp16(r, p); }
    static __device__ __forceinline__ void store(uint32_t* p, const Fp& r, uint32_t flag = 0) { store_fp16(p, r, 0, flag); }
    static __device__ __forceinline__ Fp shfl_down(const Fp& a, int d) { return shfl_down_fp(a, d); }
    static __device__ __forceinline__ void from_raw(Fp& r, const uint32_t* raw) { fp_from_raw(r, raw); }
    static __device__ __forceinline__ uint32_t to_raw(uint32_t* out, const Fp& a, bool keep) { return fp_to_raw(out, a, keep); }
    static __device__ __forceinline__ void unpack_raw(Fp& r, const uint32_t* raw) { fp_unpack_raw(r, raw); }
    static __device__ __forceinline__ void pack_raw(uint32_t* out, const Fp& a, bool keep) { fp_pack_raw(out, a, keep); }
};
template <> struct ElemIO<ec::Fp2> {
    static constexpr int SLOT = 32, RAW = 24;
    static __device__ __forceinline__ void load(ec::Fp2& r, const uint32_t* p) { load_fp16(r.c0, p); load_fp16(r.c1, p + 16); }
    static __device__ __forceinline__ void store(uint32_t* p, const ec::Fp2& r, uint32_t flag = 0) {
        store_fp16(p, r.c0);
        store_fp16(p + 16, r.c1, 0, flag);
    }
    static __device__ __forceinline__ ec::Fp2 shfl_down(const ec::Fp2& a, int d) {
        return ec::Fp2{shfl_down_fp(a.c0, d), shfl_down_fp(a.c1, d)};
    }
    static __device__ __forceinline__ void from_raw(ec::Fp2& r, const uint32_t* raw) { fp_from_raw(r.c0, raw); fp_from_raw(r.c1, raw + 12); }
    static __device__ __forceinline__ uint32_t to_raw(uint32_t* out, const ec::Fp2& a, bool keep) {
        return fp_to_raw(out, a.c0, keep) | fp_to_raw(out + 12, a.c1, keep);
    }
    static __device__ __forceinline__ void unpack_raw(ec::Fp2& r, const uint32_t* raw) { fp_unpack_raw(r.c0, raw); fp_unpack_raw(r.c1, raw + 12); }
    static __device__ __forceinline__ void pack_raw(uint32_t* out, const ec::Fp2& a, bool keep) { fp_pack_raw(out, a.c0, keep); fp_pack_raw(out + 12, a.c1, keep); }
};

struct G1C {                         // /root/reference/src/g1.rs: G1Affine / G1Projective over Fp
    using F = ec::FpOps;             // shared-call multiplier: everything outside the hot loop
    using FA = ec::FpOpsInline;      // accumulate hot loop
    using FR = ec::FpOpsInlinePS;    // the single addition site of the serial reduce loop: product-scanning multiplier (2.99 vs 3.13 ms at 2^24)
    static constexpr int OCC = 2;    // waves per SIMD the accumulate kernel is built for
};
struct G2C {                         // /root/reference/src/g2.rs: G2Affine / G2Projective over Fp2
    using F = ec::Fp2Ops;
    using FA = ec::Fp2OpsInline;     // measured at 2^18: accumulate 2.25 vs 5.05 ms against the shared bodies (despite spills)
    using FR = ec::Fp2OpsInline;     // reduce 2.28 vs 3.93 ms
    static constexpr int OCC = 2;
};
template <class C> struct Geo {
    using E = typename C::F::E;
    static constexpr int SLOT = ElemIO<E>::SLOT;
    static constexpr int PT_WORDS = 2 * SLOT, BK_WORDS = 3 * SLOT;
    static constexpr int RAW_AFF = 2 * ElemIO<E>::RAW, RAW_JAC = 3 * ElemIO<E>::RAW;
};
constexpr int G1_PT_WORDS = Geo<G1C>::PT_WORDS, G1_BK_WORDS = Geo<G1C>::BK_WORDS;
constexpr int G2_PT_WORDS = Geo<G2C>::PT_WORDS, G2_BK_WORDS = Geo<G2C>::BK_WORDS;

template <class C>
__device__ __forceinline__ ec::Proj<typename C::F> load_bucket(const uint32_t* p) {
    using E = typename C::F::E;
    ec::Proj<typename C::F> r;
    ElemIO<E>::load(r.x, p); ElemIO<E>::load(r.y, p + Geo<C>::SLOT); ElemIO<E>::load(r.z, p + 2 * Geo<C>::SLOT);
    return r;
}
template <class C>
__device__ __forceinline__ void store_bucket(uint32_t* p, const ec::Proj<typename C::F>& r) {
    using E = typename C::F::E;
    ElemIO<E>::store(p, r.x); ElemIO<E>::store(p + Geo<C>::SLOT, r.y); ElemIO<E>::store(p + 2 * Geo<C>::SLOT, r.z);
}

// Complete addition as ONE out-of-line body per curve: the reduce / merge / cold paths call it from several sites.
// G1 inlines its twelve multiplications inside it (latency-bound callers; ~16 % faster than the shared multiplier).
__device__ __noinline__ void add_inplace(ec::Proj<ec::FpOps>& a, const ec::Proj<ec::FpOps>& b) { ec::proj_add<ec::FpOpsInline>(a, b); }
__device__ __noinline__ void add_inplace(ec::Proj<ec::Fp2Ops>& a, const ec::Proj<ec::Fp2Ops>& b) { ec::proj_add<ec::Fp2Ops>(a, b); }

// ---------------------------------------------------------------------------------------------- scalars
// blst_fr (Montgomery, R = 2^256) -> canonical integer: one Montgomery reduction (multiply by 1).
// Device-side replacement for Scalar::into_bigint (/root/reference/src/scalar.rs:450-463,503-505).
__device__ __forceinline__ void fr_from_mont(uint32_t (&s)[8]) {
    using namespace fp28c;
    uint32_t t[9];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = s[k];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t m = t[0] * FR_INV32;
        uint64_t c = (uint64_t)m * FR_MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * FR_MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    // t < 2r: conditional subtract
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = (uint64_t)t[k] - FR_MOD[k] - borrow;
        d[k] = (uint32_t)v;
        borrow = (v >> 32) & 1;
    }
    bool ge = t[8] != 0 || borrow == 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = ge ? d[k] : t[k];
}

// fmt bit 0: the scalars are blst_fr Montgomery values (converted here); bit 1: sign fold (Plan::fold, host side).
// Without the fold the canonical integer s < r is recoded as it is — s P for ANY point P, the reference's semantics
// (blst's Pippenger, src/g1.rs:614-617, also for points that skipped Valid::check: Validate::No, src/g1.rs:425) — and `false` is returned.
// With the fold — only for resident bases that passed mi_msm_g{1,2}_validate_bases — a scalar above (r - 1) / 2 is replaced by
// r - s and `true` is returned: the caller inverts the sign of every digit (s P = (r - s)(-P) on the prime-order subgroup, and
// negating a point is free: NEGATION_IS_CHEAP, src/g1.rs:595).  The recoded values are then below 2^254, ceil(255 / c) windows
// always suffice and the signed recoding never carries out of the top window — for c = 15 and c = 17 (the divisors of 255) that
// removes the extra window whose single bucket collects a carry from 45 % of the points.
__device__ __forceinline__ bool load_scalar(uint32_t (&s)[8], const uint32_t* scalars, uint32_t i, unsigned fmt) {
    const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    if (fmt & 1u) {
        fr_from_mont(s);
    } else {
        // canonical integers are expected below r, but any 256-bit value is accepted: subtract r up to twice
        // (2^256 < 2.3 r) so the signed recoding never carries out of the top window
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            uint32_t d[8];
            uint64_t borrow = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint64_t v = (uint64_t)s[k] - fp28c::FR_MOD[k] - borrow;
                d[k] = (uint32_t)v;
                borrow = (v >> 32) & 1;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = borrow ? s[k] : d[k];
        }
    }
    if (!(fmt & 2u)) return false;
    // s > (r - 1) / 2  <=>  2 s > r - 1  <=>  r - s < s ... decided on d = r - s: flip when d < s
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = (uint64_t)fp28c::FR_MOD[k] - s[k] - borrow;
        d[k] = (uint32_t)v;
        borrow = (v >> 32) & 1;
    }
    uint64_t lt = 0;   // d < s ?
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = (uint64_t)d[k] - s[k] - lt;
        lt = (v >> 32) & 1;
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) any |= s[k];
    const bool flip = lt != 0 && any != 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = flip ? d[k] : s[k];
    return flip;
}

// Wave priority of the latency-bound kernels of the pipeline (sort, schedule, merge, reduce, combine).  In a pipelined call (run_msm,
// msm_curve.hpp) they run UNDER an accumulate kernel whose two waves per SIMD keep the VALU issue port busy: at equal priority the arbiter
// serves the older accumulate waves first and a sort kernel's handful of instructions wait behind thousands of multiply-adds (a 15-us
// k_colscan took 355 us).  s_setprio 3 lets these short waves issue whenever they are ready; the accumulate waves stay at 0.
#ifndef MI_AUX_PRIO
#define MI_AUX_PRIO 3
#endif
__device__ __forceinline__ void aux_priority() {
#if MI_AUX_PRIO
    __builtin_amdgcn_s_setprio(MI_AUX_PRIO);
#endif
}

constexpr uint32_t MERGE_FAN = 4;   // fan-in of one level of the merge of split buckets (k_merge, curve_kernels.cuh; the plan counts its levels)
// the schedule's counters in device memory (words): [0] items, [1] max items of a bucket, [2] entries, [3] merge list length of level 0,
// [4] split buckets, [8 + l] merge list length of level l >= 1
constexpr uint32_t MERGE_META = 32;
// [CLK_META .. CLK_META + 3]: two 64-bit sums the accumulate kernel leaves behind — shader-clock ticks (s_memtime) and 100-MHz ticks
// (s_memrealtime) its waves spent inside it: their ratio x 0.1 GHz is the clock the kernel ACTUALLY ran at in this launch (bench.py prices
// the roofline at it instead of a clock replayed from an old profile).  Zeroed with the rest by k_sched2, copied out by the last k_combine.
constexpr uint32_t CLK_META = 24;

// Nothing of it stays in registers across the kernel's hot loop (two 64-bit stamps held there cost k_accumulate<G1C> a register granule:
// 217 instead of 215 VGPRs, 16 registers fewer left beside its two waves for the sort kernels of a pipelined call): lane 0 parks the
// start stamps in the first 16 bytes of ITS OWN output slot, which nothing reads before the kernel's last store overwrites it.
struct WaveClock {
    static __device__ __forceinline__ void start(uint32_t* own_slot) {
        if ((threadIdx.x & 63u) == 0) {
            const unsigned long long t0 = clock64(), w0 = wall_clock64();
            reinterpret_cast<unsigned long long*>(own_slot)[0] = t0;
            reinterpret_cast<unsigned long long*>(own_slot)[1] = w0;
        }
    }
    // call BEFORE the output slot is written
    static __device__ __forceinline__ void stop(const uint32_t* own_slot, uint32_t* meta) {
        if ((threadIdx.x & 63u) == 0) {
            const unsigned long long t0 = reinterpret_cast<const unsigned long long*>(own_slot)[0], w0 = reinterpret_cast<const unsigned long long*>(own_slot)[1];
            atomicAdd(reinterpret_cast<unsigned long long*>(meta + CLK_META), clock64() - t0);
            atomicAdd(reinterpret_cast<unsigned long long*>(meta + CLK_META + 2), wall_clock64() - w0);
        }
    }
};

// Item geometry of the schedule, ONE definition for the kernels that build the items (sort_kernels.cuh) and the ones that walk them
// (curve_kernels.cuh): a bucket of up to T = 2^logT entries is one item, a fuller one is cut into items of S = 2^logS entries.
// Launches pack logT | (class shift << 8) | (logS << 16) into one argument.
__device__ __forceinline__ uint32_t item_size_log(uint32_t cnt, uint32_t logT, uint32_t logS) { return cnt > (1u << logT) ? logS : logT; }

// c-bit field starting at bit `off` of a 256-bit little-endian integer (zero beyond bit 255)
__device__ __forceinline__ uint32_t scalar_bits(const uint32_t (&s)[8], uint32_t off, uint32_t c) {
    uint32_t w = off >> 5, sh = off & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {  // register-resident select instead of dynamic indexing
        lo = (w == (uint32_t)k) ? s[k] : lo;
        hi = (w + 1 == (uint32_t)k) ? s[k] : hi;
    }
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// Signed-digit recoding shared by the histogram and scatter passes.  Calls f(window, bucket, negative) for every
// non-zero digit; bucket = |d| - 1 in [0, 2^(c-1)).
template <class Fn>
__device__ __forceinline__ void for_each_digit(const uint32_t (&s)[8], bool flip, uint32_t c, uint32_t nwin, Fn f) {
    uint32_t carry = 0, half = 1u << (c - 1);
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t raw = scalar_bits(s, w * c, c) + carry;
        bool neg = raw > half;
        carry = neg ? 1u : 0u;
        uint32_t mag = neg ? (1u << c) - raw : raw;
        if (mag != 0) f(w, mag - 1, neg != flip);
    }
}

// Same recoding with the window size known at compile time: the window loop unrolls, every bit-field extraction
// becomes one or two shifts on statically indexed words (the runtime version spends 16 selects per window), and
// windows outside [w0, w1) cost only their carry.  k_coarse is instantiated for c = 7..22.
template <int CB, class Fn>
__device__ __forceinline__ void for_each_digit_static(const uint32_t (&s)[8], bool flip, uint32_t w0, uint32_t w1, Fn f) {
    // NW = the windows of the unfolded recoding (num_windows(CB, false), common.hpp); with the fold the value is below 2^254, the last
    // of them (CB = 15, 17) sees no bits and no carry and lies beyond w1
    constexpr uint32_t NW = (255 + CB - 1) / CB + (255 % CB == 0 ? 1 : 0), HALF = 1u << (CB - 1), MASKC = (1u << CB) - 1u;
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) {
        constexpr uint32_t dummy = 0; (void)dummy;
        const uint32_t off = w * CB, wi = off >> 5, sh = off & 31;
        uint32_t v = wi < 8 ? s[wi < 8 ? wi : 0] >> sh : 0u;
        if (sh + CB > 32 && wi + 1 < 8) v |= s[wi + 1 < 8 ? wi + 1 : 0] << (32 - sh);
        uint32_t raw = (v & MASKC) + carry;
        bool neg = raw > HALF;
        carry = neg ? 1u : 0u;
        uint32_t mag = neg ? (1u << CB) - raw : raw;
        if (w >= w0 && w < w1 && mag != 0) f(w, mag - 1, neg != flip);
    }
}

}  // namespace msmk
