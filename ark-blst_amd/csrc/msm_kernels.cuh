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

// element I/O, generic over Fp / Fp2
template <class E> struct ElemIO;
template <> struct ElemIO<Fp> {
    static constexpr int SLOT = 16, RAW = 12;
    static __device__ __forceinline__ void load(Fp& r, const uint32_t* p) { load_fp16(r, p); }
    static __device__ __forceinline__ void store(uint32_t* p, const Fp& r, uint32_t flag = 0) { store_fp16(p, r, 0, flag); }
    static __device__ __forceinline__ Fp shfl_down(const Fp& a, int d) { return shfl_down_fp(a, d); }
    static __device__ __forceinline__ void from_raw(Fp& r, const uint32_t* raw) { fp_from_raw(r, raw); }
    static __device__ __forceinline__ uint32_t to_raw(uint32_t* out, const Fp& a, bool keep) { return fp_to_raw(out, a, keep); }
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
};

struct G1C {                         // /root/reference/src/g1.rs: G1Affine / G1Projective over Fp
    using F = ec::FpOps;             // shared-call multiplier: everything outside the hot loop
    using FA = ec::FpOpsInline;      // accumulate hot loop
    using FR = ec::FpOpsInline;      // the single addition site of the reduce loop
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
__device__ __noinline__ void add_inplace(ec::Proj<ec::FpOps>& a, const ec::Proj<ec::FpOps>& b) {
    ec::Proj<ec::FpOpsInline>& ai = reinterpret_cast<ec::Proj<ec::FpOpsInline>&>(a);
    const ec::Proj<ec::FpOpsInline>& bi = reinterpret_cast<const ec::Proj<ec::FpOpsInline>&>(b);
    ec::proj_add<ec::FpOpsInline>(ai, bi);
}
__device__ __noinline__ void add_inplace(ec::Proj<ec::Fp2Ops>& a, const ec::Proj<ec::Fp2Ops>& b) { ec::proj_add<ec::Fp2Ops>(a, b); }

// ---------------------------------------------------------------------------------------------- ingest
// raw: n affine points in the reference's form.  One thread per point.
template <class C>
__global__ void __launch_bounds__(256) k_ingest(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out,
                                                uint8_t* __restrict__ inf_flags, uint32_t n) {
    using E = typename C::F::E;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = raw + (size_t)i * Geo<C>::RAW_AFF;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < Geo<C>::RAW_AFF; k++) any |= q[k];
    E x, y;
    ElemIO<E>::from_raw(x, q);
    ElemIO<E>::from_raw(y, q + ElemIO<E>::RAW);
    uint32_t* o = out + (size_t)i * Geo<C>::PT_WORDS;
    ElemIO<E>::store(o, x);
    ElemIO<E>::store(o + Geo<C>::SLOT, y, any == 0 ? 1u : 0u);
    inf_flags[i] = any == 0 ? 1 : 0;   // compact copy for the sort passes (a 4-byte read per 128-byte point costs a line)
}

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

__device__ __forceinline__ void load_scalar(uint32_t (&s)[8], const uint32_t* scalars, uint32_t i, unsigned fmt) {
    const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    if (fmt == 1) {
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
}

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
__device__ __forceinline__ void for_each_digit(const uint32_t (&s)[8], uint32_t c, uint32_t nwin, Fn f) {
    uint32_t carry = 0, half = 1u << (c - 1);
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t raw = scalar_bits(s, w * c, c) + carry;
        bool neg = raw > half;
        carry = neg ? 1u : 0u;
        uint32_t mag = neg ? (1u << c) - raw : raw;
        if (mag != 0) f(w, mag - 1, neg);
    }
}

// Same recoding with the window size known at compile time: the window loop unrolls, every bit-field extraction
// becomes one or two shifts on statically indexed words (the runtime version spends 16 selects per window), and
// windows outside [w0, w1) cost only their carry.  k_coarse is instantiated for c = 7..22.
template <int CB, class Fn>
__device__ __forceinline__ void for_each_digit_static(const uint32_t (&s)[8], uint32_t w0, uint32_t w1, Fn f) {
    constexpr uint32_t NW = (256 + CB - 1) / CB, HALF = 1u << (CB - 1), MASKC = (1u << CB) - 1u;
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
        if (w >= w0 && w < w1 && mag != 0) f(w, mag - 1, neg);
    }
}

// ---------------------------------------------------------------------------------------------- bucket sort
// Two-level sort of the N*W (window, bucket) keys, staged through LDS — replaces one global atomic per key in the
// histogram pass and one returning global atomic + 4-byte scatter per key in the scatter pass.
//   bucket id = (hi, lo): lo = low `lo_bits` (<= 8) bits -> F = 2^lo_bits fine buckets, H = 2^(c-1) / F coarse bins
//   level 1 (global, coarse):  k_coarse_count   per tile of points: LDS histogram over (window, hi)  -> tilecnt[tile][bin]
//                              k_colscan        per bin: exclusive scan over tiles, bin totals
//                              k_binscan        exclusive scan of the bin totals                     -> bin_base[bin]
//                              k_coarse_scatter per tile: LDS cursors seeded with the scanned bases; entries
//                                               (index | sign | lo) land in their coarse bin of `coarse`
//   level 2 (LDS, fine):       k_fine_count / k_fine_scan / k_fine_scatter over bin SEGMENTS (see below)
//                                               -> sorted[] in (window, bucket) order and hist[window][bucket]
// Windows are processed in groups of `wgroup` so that wgroup * H counters fit LDS (<= 16384 counters, 64 KB).
struct SortGeom {
    uint32_t n, fmt, c, nwin;
    uint32_t lo_bits, H;          // fine bits, coarse bins per window
    uint32_t tiles, tile_pts;     // point tiles (grid.x) and points per tile (multiple of 1024)
    uint32_t wgroup, ngroups;     // windows per group, groups (grid.y)
    uint32_t nbins;               // nwin * H
};
constexpr uint32_t SORT_MAX_COUNTERS = 16384;  // 64 KB of LDS counters per workgroup (2 workgroups per CU)

// entry in `coarse`: (point index << (lo_bits+1)) | (negative << lo_bits) | lo
template <bool SCATTER, int CB>
__global__ void __launch_bounds__(1024) k_coarse(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flags, SortGeom g,
                                                 uint32_t* __restrict__ tilecnt, const uint32_t* __restrict__ bin_base,
                                                 uint32_t* __restrict__ coarse) {
    __shared__ uint32_t cnt[SORT_MAX_COUNTERS];
    const uint32_t tile = blockIdx.x, grp = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
    uint32_t w0 = grp * g.wgroup, w1 = w0 + g.wgroup < g.nwin ? w0 + g.wgroup : g.nwin;
    uint32_t ncnt = (w1 - w0) * g.H;
    for (uint32_t k = t; k < ncnt; k += nt) {
        if (SCATTER) {
            uint32_t bin = w0 * g.H + k;
            cnt[k] = bin_base[bin] + tilecnt[(size_t)tile * g.nbins + bin];  // where this tile's run of the bin starts
        } else {
            cnt[k] = 0;
        }
    }
    __syncthreads();
    uint32_t lo_mask = (1u << g.lo_bits) - 1u;
    uint32_t p0 = tile * g.tile_pts, p1 = p0 + g.tile_pts < g.n ? p0 + g.tile_pts : g.n;
    for (uint32_t i = p0 + t; i < p1; i += nt) {
        if (inf_flags[i] != 0) continue;  // infinity base: contributes nothing
        uint32_t s[8];
        load_scalar(s, scalars, i, g.fmt);
        for_each_digit_static<CB>(s, w0, w1, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t k = (w - w0) * g.H + (b >> g.lo_bits);
            uint32_t pos = atomicAdd(&cnt[k], 1u);
            if (SCATTER) coarse[pos] = (i << (g.lo_bits + 1)) | ((neg ? 1u : 0u) << g.lo_bits) | (b & lo_mask);
        });
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t k = t; k < ncnt; k += nt) tilecnt[(size_t)tile * g.nbins + w0 * g.H + k] = cnt[k];
    }
}

// per bin: exclusive scan over the tiles (in place) and the bin total.  One lane per bin: coalesced across bins.
__global__ void __launch_bounds__(256) k_colscan(uint32_t* __restrict__ tilecnt, uint32_t nbins, uint32_t tiles,
                                                 uint32_t* __restrict__ bin_tot) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    uint32_t run = 0;
    for (uint32_t k = 0; k < tiles; k++) {
        uint32_t v = tilecnt[(size_t)k * nbins + b];
        tilecnt[(size_t)k * nbins + b] = run;
        run += v;
    }
    bin_tot[b] = run;
}

// exclusive scan of m <= ~100k values by one workgroup; out[m] = total
__global__ void __launch_bounds__(1024) k_binscan(const uint32_t* __restrict__ in, uint32_t m, uint32_t* __restrict__ out) {
    __shared__ uint32_t part[1024];
    uint32_t t = threadIdx.x;
    uint32_t per = (m + 1023) / 1024;
    uint32_t lo = t * per, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += in[k];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t v = in[k];
        out[k] = run;
        run += v;
    }
    if (t == 1023) out[m] = part[1023];
}

// ---- level 2: fine sort.  A coarse bin is cut into SEGMENTS of at most FINE_SEG entries, one workgroup each, so a
// heavy bin (skewed scalars; the short top window, whose n entries share a handful of buckets) is spread over
// the chip instead of being streamed by a single workgroup (measured: 30 ms for two 8 M-entry bins at n = 2^24).
//   k_seg_count   segments per bin                       -> seg_cnt[bin]      (then k_binscan -> seg_base[bin])
//   k_fine_count  per segment: LDS histogram of lo       -> segcnt[seg][lo]
//   k_fine_scan   per bin: scan over its segments and over lo -> segcnt becomes the start of (seg, lo) inside the
//                 bin; emits hist[window][bucket]
//   k_fine_scatter per segment: LDS cursors seeded from segcnt -> sorted[]
constexpr uint32_t FINE_SEG = 8192;

__global__ void __launch_bounds__(256) k_seg_count(const uint32_t* __restrict__ bin_base, uint32_t nbins, uint32_t* __restrict__ seg_cnt) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    uint32_t sz = bin_base[b + 1] - bin_base[b];
    seg_cnt[b] = sz == 0 ? 1u : (sz + FINE_SEG - 1) / FINE_SEG;
}

// segment id -> (bin, first entry, one-past-last entry); identity guess first (no bin split before it), else binary search
__device__ __forceinline__ bool seg_locate(uint32_t seg, const uint32_t* seg_base, const uint32_t* bin_base, uint32_t nbins,
                                           uint32_t& bin, uint32_t& beg, uint32_t& end) {
    if (seg >= seg_base[nbins]) return false;
    uint32_t b = seg < nbins ? seg : nbins - 1;
    if (!(seg_base[b] <= seg && seg < seg_base[b + 1])) {
        uint32_t lo = 0, hi = b;
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (seg_base[mid] <= seg) lo = mid; else hi = mid - 1;
        }
        b = lo;
    }
    uint32_t k = seg - seg_base[b];
    bin = b;
    beg = bin_base[b] + k * FINE_SEG;
    uint32_t bend = bin_base[b + 1];
    end = beg + FINE_SEG < bend ? beg + FINE_SEG : bend;
    return true;
}

__global__ void __launch_bounds__(256) k_fine_count(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ bin_base,
                                                    const uint32_t* __restrict__ seg_base, SortGeom g, uint32_t* __restrict__ segcnt) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t sb[3];
    uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, seg_base, bin_base, g.nbins, bin, beg, end);
        sb[0] = ok ? beg : 1u; sb[1] = ok ? end : 0u; sb[2] = ok ? 1u : 0u;
    }
    cnt[t] = 0;
    __syncthreads();
    if (!sb[2]) return;
    uint32_t beg = sb[0], end = sb[1];
    uint32_t F = 1u << g.lo_bits, lo_mask = F - 1u;
    for (uint32_t i = beg + t; i < end; i += 256) atomicAdd(&cnt[coarse[i] & lo_mask], 1u);
    __syncthreads();
    if (t < F) segcnt[(size_t)seg * F + t] = cnt[t];
}

// one workgroup per bin, lane = lo.  segcnt[seg][lo] <- offset of (seg, lo) relative to the bin start.
__global__ void __launch_bounds__(256) k_fine_scan(const uint32_t* __restrict__ seg_base, SortGeom g, uint32_t* __restrict__ segcnt,
                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t scan[256];
    uint32_t bin = blockIdx.x, t = threadIdx.x;
    uint32_t F = 1u << g.lo_bits;
    uint32_t s0 = seg_base[bin], s1 = seg_base[bin + 1];
    uint32_t tot = 0;
    if (t < F)
        for (uint32_t sg = s0; sg < s1; sg++) {
            uint32_t v = segcnt[(size_t)sg * F + t];
            segcnt[(size_t)sg * F + t] = tot;   // exclusive over the segments of this (bin, lo)
            tot += v;
        }
    scan[t] = t < F ? tot : 0;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    if (t < F) {
        hist[(size_t)bin * F + t] = tot;                 // bucket (window, hi, lo) has index bin * F + lo
        uint32_t lo_base = scan[t] - tot;                // start of fine bucket lo inside the bin
        for (uint32_t sg = s0; sg < s1; sg++) segcnt[(size_t)sg * F + t] += lo_base;
    }
}

__global__ void __launch_bounds__(256) k_fine_scatter(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ bin_base,
                                                      const uint32_t* __restrict__ seg_base, SortGeom g,
                                                      const uint32_t* __restrict__ segcnt, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cur[256];
    __shared__ uint32_t sb[4];
    uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, seg_base, bin_base, g.nbins, bin, beg, end);
        sb[0] = beg; sb[1] = end; sb[2] = ok ? 1u : 0u; sb[3] = ok ? bin_base[bin] : 0u;
    }
    __syncthreads();
    if (!sb[2]) return;
    uint32_t beg = sb[0], end = sb[1], bin_beg = sb[3];
    uint32_t F = 1u << g.lo_bits, lo_mask = F - 1u;
    cur[t] = t < F ? bin_beg + segcnt[(size_t)seg * F + t] : 0u;
    __syncthreads();
    uint32_t sh = g.lo_bits + 1;
    for (uint32_t i = beg + t; i < end; i += 256) {
        uint32_t e = coarse[i];
        uint32_t pos = atomicAdd(&cur[e & lo_mask], 1u);
        sorted[pos] = (e >> sh) | (((e >> g.lo_bits) & 1u) << 31);
    }
}

// ---------------------------------------------------------------------------------------------- scan / schedule
// Three small launches turn the histogram into (a) entry offsets for the scatter, (b) WORK ITEMS and (c) a
// length-sorted processing order for them:
//   * a bucket with cnt entries becomes max(1, ceil(cnt / T)) items of at most T = 2^logT entries, so a heavy
//     bucket (skewed scalars, or the short top window) is spread over many lanes instead of serialising one;
//   * order[] lists item ids by DESCENDING length class (65 classes): the 64 lanes of a wave then run the same
//     number of additions (bucket loads are Poisson distributed: unsorted, a wave waits for its longest lane,
//     ~70 % lane efficiency at a mean of 32) and the longest items start first.
// Layout: nblk <= 256 blocks of 1024 lanes; block k owns `per_blk` consecutive buckets, lane t owns
// per_blk/1024 consecutive ones.  item id of (bucket b, chunk k) = woff[b] + k.
constexpr int SCHED_CLASSES = 65;

__device__ __forceinline__ uint32_t items_of(uint32_t cnt, uint32_t logT) {
    return cnt == 0 ? 1u : (cnt + (1u << logT) - 1u) >> logT;
}
__device__ __forceinline__ uint32_t class_of(uint32_t len, uint32_t logT) { return (len << 6) >> logT; }  // 0..64

__global__ void __launch_bounds__(1024) k_sched1(const uint32_t* __restrict__ hist, uint32_t m, uint32_t per_blk, uint32_t logT,
                                                 uint32_t nblk, uint32_t* __restrict__ blk_e, uint32_t* __restrict__ blk_i,
                                                 uint32_t* __restrict__ blk_cls, uint32_t* __restrict__ blk_max) {
    __shared__ uint32_t cls[SCHED_CLASSES];
    __shared__ uint32_t se, si, smax;
    uint32_t t = threadIdx.x, blk = blockIdx.x;
    if (t < SCHED_CLASSES) cls[t] = 0;
    if (t == 0) { se = 0; si = 0; smax = 1; }
    __syncthreads();
    uint32_t per_t = per_blk >> 10;
    uint32_t lo = blk * per_blk + t * per_t, hi = lo + per_t < m ? lo + per_t : m;
    uint32_t sum_e = 0, sum_i = 0, mx = 1, T = 1u << logT;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k], it = items_of(h, logT);
        sum_e += h;
        sum_i += it;
        mx = it > mx ? it : mx;
        if (it > 1) atomicAdd(&cls[64], it - 1);
        atomicAdd(&cls[class_of(h - (it - 1) * T, logT)], 1u);
    }
    atomicAdd(&se, sum_e);
    atomicAdd(&si, sum_i);
    atomicMax(&smax, mx);
    __syncthreads();
    if (t < SCHED_CLASSES) blk_cls[t * nblk + blk] = cls[t];
    if (t == 0) { blk_e[blk] = se; blk_i[blk] = si; blk_max[blk] = smax; }
}

// one workgroup: exclusive scans over the <= 256 block sums, and per (class, block) the base position in order[]
__global__ void __launch_bounds__(1024) k_sched2(uint32_t nblk, uint32_t* __restrict__ blk_e, uint32_t* __restrict__ blk_i,
                                                 uint32_t* __restrict__ blk_cls, const uint32_t* __restrict__ blk_max,
                                                 uint32_t* __restrict__ meta) {
    __shared__ uint32_t a[256], b[256], ctot[SCHED_CLASSES], cbase[SCHED_CLASSES];
    uint32_t t = threadIdx.x;
    uint32_t ve = 0, vi = 0;
    if (t < 256) {
        ve = t < nblk ? blk_e[t] : 0;
        vi = t < nblk ? blk_i[t] : 0;
        a[t] = ve;
        b[t] = vi;
    }
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t xa = 0, xb = 0;
        if (t < 256 && t >= d) { xa = a[t - d]; xb = b[t - d]; }
        __syncthreads();
        if (t < 256) { a[t] += xa; b[t] += xb; }
        __syncthreads();
    }
    if (t < nblk) { blk_e[t] = a[t] - ve; blk_i[t] = b[t] - vi; }
    // class rows: wave w handles classes w, w+16, ...; exclusive scan of each row in chunks of 64 lanes
    uint32_t wave = t >> 6, lane = t & 63;
    for (uint32_t c = wave; c < SCHED_CLASSES; c += 16) {
        uint32_t run = 0;
        for (uint32_t base = 0; base < nblk; base += 64) {
            uint32_t idx = base + lane;
            uint32_t v = idx < nblk ? blk_cls[c * nblk + idx] : 0, incl = v;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
            if (idx < nblk) blk_cls[c * nblk + idx] = run + incl - v;
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) ctot[c] = run;
    }
    __syncthreads();
    if (t < SCHED_CLASSES) {  // descending class order: class c starts after all longer classes
        uint32_t above = 0;
        for (uint32_t c = t + 1; c < SCHED_CLASSES; c++) above += ctot[c];
        cbase[t] = above;
    }
    __syncthreads();
    for (uint32_t idx = t; idx < SCHED_CLASSES * nblk; idx += 1024) blk_cls[idx] += cbase[idx / nblk];
    if (t == 0) {
        uint32_t mx = 1;
        for (uint32_t k = 0; k < nblk; k++) mx = blk_max[k] > mx ? blk_max[k] : mx;
        meta[0] = b[255];  // total items   (a[], b[] are inclusive scans; entries past nblk are zero)
        meta[1] = mx;      // max items of any bucket
        meta[2] = a[255];  // total entries
        meta[3] = 0;       // merge-list length, filled by k_sched3
    }
}

// per block: bucket-level exclusive scans -> offsets / cursor / woff; every item gets its slot in order[]
__global__ void __launch_bounds__(1024) k_sched3(const uint32_t* __restrict__ hist, uint32_t m, uint32_t per_blk, uint32_t logT,
                                                 uint32_t nblk, const uint32_t* __restrict__ blk_e, const uint32_t* __restrict__ blk_i,
                                                 const uint32_t* __restrict__ blk_cls, uint32_t* __restrict__ offsets,
                                                 uint32_t* __restrict__ woff,
                                                 uint32_t* __restrict__ order, uint32_t* __restrict__ item_bucket,
                                                 uint32_t* __restrict__ merge_list, uint32_t* __restrict__ meta) {
    __shared__ uint32_t pe[1024], pi[1024], cur[SCHED_CLASSES];
    // buckets split into many items (skewed scalars: one bucket can hold all N entries) are written out by the whole
    // workgroup after the per-lane pass; one lane doing it alone cost 0.65 ms for a bucket of 2^20 entries
    constexpr uint32_t HV_CAP = 64, HV_MIN = 64;
    __shared__ uint32_t hv_k[HV_CAP], hv_run[HV_CAP], hv_it[HV_CAP], hv_pos[HV_CAP], hv_mp[HV_CAP], hv_n;
    uint32_t t = threadIdx.x, blk = blockIdx.x;
    if (t == 0) hv_n = 0;
    if (t < SCHED_CLASSES) cur[t] = blk_cls[t * nblk + blk];
    uint32_t per_t = per_blk >> 10;
    uint32_t lo = blk * per_blk + t * per_t, hi = lo + per_t < m ? lo + per_t : m;
    uint32_t sum_e = 0, sum_i = 0, T = 1u << logT;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k];
        sum_e += h;
        sum_i += items_of(h, logT);
    }
    pe[t] = sum_e;
    pi[t] = sum_i;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t xe = t >= d ? pe[t - d] : 0, xi = t >= d ? pi[t - d] : 0;
        __syncthreads();
        pe[t] += xe;
        pi[t] += xi;
        __syncthreads();
    }
    uint32_t run_e = blk_e[blk] + pe[t] - sum_e, run_i = blk_i[blk] + pi[t] - sum_i;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k], it = items_of(h, logT);
        offsets[k] = run_e;
        woff[k] = run_i;
        if (it > 1) {  // full-length chunks of a split bucket: one reservation in the longest class
            uint32_t pos = atomicAdd(&cur[64], it - 1);
            uint32_t mp = atomicAdd(&meta[3], it);     // the merge passes only visit the items of split buckets
            uint32_t slot = it >= HV_MIN ? atomicAdd(&hv_n, 1u) : HV_CAP;
            if (slot < HV_CAP) {
                hv_k[slot] = k; hv_run[slot] = run_i; hv_it[slot] = it; hv_pos[slot] = pos; hv_mp[slot] = mp;
            } else {
                for (uint32_t j = 0; j + 1 < it; j++) { order[pos + j] = run_i + j; item_bucket[run_i + j] = k; }
                for (uint32_t j = 0; j < it; j++) merge_list[mp + j] = run_i + j;
            }
        }
        uint32_t last = run_i + it - 1;
        uint32_t pos = atomicAdd(&cur[class_of(h - (it - 1) * T, logT)], 1u);
        order[pos] = last;
        item_bucket[last] = k;
        run_e += h;
        run_i += it;
    }
    if (blk == nblk - 1 && t == 1023) { offsets[m] = run_e; woff[m] = run_i; }
    __syncthreads();
    const uint32_t nh = hv_n < HV_CAP ? hv_n : HV_CAP;
    for (uint32_t s = 0; s < nh; s++) {
        const uint32_t k = hv_k[s], r0 = hv_run[s], it = hv_it[s], pos = hv_pos[s], mp = hv_mp[s];
        for (uint32_t j = t; j < it; j += 1024) {
            if (j + 1 < it) { order[pos + j] = r0 + j; item_bucket[r0 + j] = k; }
            merge_list[mp + j] = r0 + j;
        }
    }
}

// ---------------------------------------------------------------------------------------------- accumulate
template <class C>
__device__ __forceinline__ void load_point(typename C::F::E& x, typename C::F::E& y, const uint32_t* bases, uint32_t ent) {
    using E = typename C::F::E;
    const uint32_t* p = bases + (size_t)(ent & 0x7fffffffu) * Geo<C>::PT_WORDS;
    ElemIO<E>::load(x, p);
    ElemIO<E>::load(y, p + Geo<C>::SLOT);
}

// One lane per WORK ITEM = (bucket, chunk): item i of bucket b covers entries
// sorted[offsets[b] + k*T .. min(offsets[b] + (k+1)*T, offsets[b+1])), k = i - woff[b]; entries are (index | sign<<31).
// Hot loop: XYZZ mixed additions.  Register budget is the constraint (256 VGPRs at 2 waves/SIMD), so the next
// point is not staged in registers: its index is fetched one iteration ahead and its line(s) touched early so the
// real load hits L2; the other resident wave covers what latency is left (staging the next point in 28 more
// registers was measured slower: 3.06 vs 2.88 ms, it pushes the loop into scratch spills).
// A lane that meets an exceptional pair (same x) leaves the hot loop and finishes on the complete formulas.
// Output: partial[i] (projective), i = natural item id.
template <class C>
__global__ void __launch_bounds__(256, C::OCC) k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ woff,
                                                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ item_bucket,
                                                            uint32_t nitems, uint32_t logT, uint32_t* __restrict__ partial) {
    using F = typename C::F;
    using FA = typename C::FA;
    using E = typename F::E;
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nitems) return;
    uint32_t i = order[j];          // items are processed longest class first; partial[] keeps natural item order
    uint32_t b = item_bucket[i];
    uint32_t k = i - woff[b];
    uint32_t e = offsets[b] + (k << logT), bend = offsets[b + 1];
    uint32_t end = e + (1u << logT) < bend ? e + (1u << logT) : bend;
    ec::Xyzz<FA> acc;
    acc.x = F::zero(); acc.y = F::zero(); acc.zz = F::zero(); acc.zzz = F::zero();
    bool inf = true;
    uint32_t nent = e < end ? sorted[e] : 0u;
    while (e < end) {
        uint32_t ent = nent;
        E x, y;
        load_point<C>(x, y, bases, ent);
        if (e + 1 < end) {
            nent = sorted[e + 1];
            __builtin_prefetch(bases + (size_t)(nent & 0x7fffffffu) * Geo<C>::PT_WORDS, 0, 1);
        }
        y = F::select((ent >> 31) != 0, y, FA::template neg_l<4>(y));
        if (inf) {
            acc.x = x; acc.y = FA::norm(y); acc.zz = F::one(); acc.zzz = F::one();   // acc.y must be subtractable: N-form
            inf = false;
        } else if (ec::xyzz_madd<FA>(acc, x, y)) {
            break;  // exceptional pair at entry e: acc untouched
        }
        e++;
    }
    ec::Proj<F> out = ec::proj_inf<F>();
    if (!inf) out = ec::xyzz_to_proj<F>(reinterpret_cast<const ec::Xyzz<F>&>(acc));
    while (e < end) {  // cold path (never taken on random inputs): complete additions
        uint32_t ent = sorted[e];
        E x, y;
        load_point<C>(x, y, bases, ent);
        y = F::select((ent >> 31) != 0, y, F::template neg<4>(y));
        ec::Proj<F> q = ec::proj_from_affine<F>(x, y);
        ec::proj_add<F>(out, q);  // shared-call multiplier: keeps the cold path out of the hot loop's register budget
        e++;
    }
    store_bucket<C>(partial + (size_t)i * Geo<C>::BK_WORDS, out);
}

// One binary-tree level of the per-bucket merge of split buckets: partial[i] += partial[i + d] for the items whose
// chunk index is a multiple of 2d.  After ceil(log2(max items)) levels partial[woff[b]] is bucket b.  Launched only
// when some bucket was split (meta[1] > 1), over the items of split buckets only (merge_list, meta[3] entries).
template <class C>
__global__ void __launch_bounds__(256, 1) k_merge(uint32_t* __restrict__ partial, const uint32_t* __restrict__ item_bucket,
                                                       const uint32_t* __restrict__ woff, const uint32_t* __restrict__ merge_list,
                                                       uint32_t nlist, uint32_t d) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nlist) return;
    uint32_t i = merge_list[j];
    uint32_t b = item_bucket[i];
    uint32_t n = woff[b + 1] - woff[b];
    if (n <= d) return;
    uint32_t k = i - woff[b];
    if ((k & (2 * d - 1)) != 0 || k + d >= n) return;
    auto a = load_bucket<C>(partial + (size_t)i * Geo<C>::BK_WORDS);
    auto c = load_bucket<C>(partial + (size_t)(i + d) * Geo<C>::BK_WORDS);
    add_inplace(a, c);
    store_bucket<C>(partial + (size_t)i * Geo<C>::BK_WORDS, a);
}

// ---------------------------------------------------------------------------------------------- reduce
template <class C>
__device__ __forceinline__ ec::Proj<typename C::F> shfl_down_pt(const ec::Proj<typename C::F>& a, int d) {
    using E = typename C::F::E;
    ec::Proj<typename C::F> r;
    r.x = ElemIO<E>::shfl_down(a.x, d); r.y = ElemIO<E>::shfl_down(a.y, d); r.z = ElemIO<E>::shfl_down(a.z, d);
    return r;
}

// projective (X : Y : Z) -> Jacobian in the reference's form (X Z, Y Z^2, Z); infinity (Z == 0 mod p) -> all-zero
template <class C>
__device__ __forceinline__ void store_jac_raw(uint32_t* out, const ec::Proj<typename C::F>& p) {
    using F = typename C::F;
    using E = typename F::E;
    constexpr int R = ElemIO<E>::RAW;
    uint32_t any = ElemIO<E>::to_raw(out + 2 * R, p.z, true);
    E zz = F::mul(p.z, p.z);
    ElemIO<E>::to_raw(out, F::mul(p.x, p.z), any != 0);
    ElemIO<E>::to_raw(out + R, F::mul(p.y, zz), any != 0);
}

// One wave per chunk of 64*L consecutive buckets of one window (L = 2^logL).  Lane l owns buckets
// [l*L, l*L+L) of the chunk.  Output per chunk: S = sum B, T = sum (rel+1) B with rel = index inside the chunk,
// as two Jacobian points in the reference's form.  All additions are the complete projective formulas.
//
// The whole reduction is ONE loop with ONE inlined addition site: the operands of step s are selected by the
// (wave-uniform) step number.  An out-of-line addition would pass its 2 x 42 limbs through scratch (the AMDGPU
// calling convention puts large structs on the stack): measured 568 MB of scratch writes per launch and ~15 % of
// the kernel time; five inlined sites would be 5 x 55 KB of code.  Steps:
//   [0, 2L)            t = L-1 .. 0 :  run += B_t ;  acc += run          (lane-serial running sums)
//   [2L, 2L+6)         run += shfl_down(run, 1, 2, 4, .., 32)             (suffix scan: run_l = sum_{j>=l} S_j)
//   [.., +logL)        LP = 2 LP  (LP starts as run)                      (L * P_l)
//   one step           acc += (lane == 0 ? inf : LP)                      (V_l = T_l + L P_l)
//   six steps          acc += shfl_down(acc, 32, 16, .., 1)               (sum over lanes)
template <class C>
__global__ void __launch_bounds__(64, 1) k_reduce(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ woff,
                                                  uint32_t* __restrict__ pairs, uint32_t logL) {
    using F = typename C::F;
    using FR = typename C::FR;
    using PJ = ec::Proj<F>;
    using PR = ec::Proj<FR>;
    uint32_t chunk = blockIdx.x, lane = threadIdx.x;
    uint32_t L = 1u << logL;
    const uint32_t* wp = woff + (size_t)chunk * 64 * L + (size_t)lane * L;  // bucket b lives at partial[woff[b]]
    PJ run = ec::proj_inf<F>(), acc = ec::proj_inf<F>(), LP = ec::proj_inf<F>();
    const uint32_t s_scan = 2 * L, s_dbl = s_scan + 6, s_comb = s_dbl + logL, s_end = s_comb + 7;
#pragma unroll 1
    for (uint32_t s = 0; s < s_end; s++) {
        PJ A, B;
        uint32_t dst;  // 0 run, 1 acc, 2 LP
        if (s < s_scan) {
            if ((s & 1u) == 0) {
                A = run; B = load_bucket<C>(partial + (size_t)wp[L - 1 - (s >> 1)] * Geo<C>::BK_WORDS); dst = 0;
            } else {
                A = acc; B = run; dst = 1;
            }
        } else if (s < s_dbl) {
            int d = 1 << (s - s_scan);
            A = run;
            B = ec::proj_select<F>(lane + d < 64, ec::proj_inf<F>(), shfl_down_pt<C>(run, d));
            dst = 0;
        } else if (s < s_comb) {
            if (s == s_dbl) LP = run;
            A = LP; B = LP; dst = 2;
        } else if (s == s_comb) {
            if (logL == 0) LP = run;
            A = acc;
            B = ec::proj_select<F>(lane == 0, LP, ec::proj_inf<F>());
            dst = 1;
        } else {
            int d = 32 >> (s - s_comb - 1);
            A = acc;
            B = ec::proj_select<F>((int)lane < d, ec::proj_inf<F>(), shfl_down_pt<C>(acc, d));
            dst = 1;
        }
        ec::proj_add<FR>(reinterpret_cast<PR&>(A), reinterpret_cast<const PR&>(B));
        if (dst == 0) run = A;
        else if (dst == 1) acc = A;
        else LP = A;
    }
    if (lane == 0) {
        store_jac_raw<C>(pairs + (size_t)chunk * 2 * Geo<C>::RAW_JAC, run);
        store_jac_raw<C>(pairs + (size_t)chunk * 2 * Geo<C>::RAW_JAC + Geo<C>::RAW_JAC, acc);
    }
}


// G2 variant with TWO lanes per logical lane (CoopF2: the even lane holds c0 and the odd lane c1 of every Fp2 coordinate; a
// product is one fused two-term reduction per lane): 32 logical lanes per wave, each owning Lc = 2^logLc consecutive buckets
// of a chunk of 32 * Lc buckets.  Same steps and outputs as k_reduce<G2C>, half the multiplications per lane and step, and
// the per-lane state of the G1 kernel (no spills): 1.42 vs 2.18 ms at 2^20 points (the kernel is a latency chain).
__global__ void __launch_bounds__(64, 1) k_reduce_g2_coop(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ woff,
                                                          uint32_t* __restrict__ pairs, uint32_t logLc) {
    using F = CoopF2;
    using PJ = ec::Proj<F>;
    const uint32_t chunk = blockIdx.x, lane = threadIdx.x, h = lane & 1u, ll = lane >> 1;
    const uint32_t Lc = 1u << logLc;
    const uint32_t* wp = woff + (size_t)chunk * 32 * Lc + (size_t)ll * Lc;
    auto load_b = [&](uint32_t idx) {
        const uint32_t* p = partial + (size_t)idx * G2_BK_WORDS + 16 * h;   // x | y | z, each (c0, c1) in 16-word slots
        PJ r;
        load_fp16(r.x, p); load_fp16(r.y, p + 32); load_fp16(r.z, p + 64);
        return r;
    };
    auto shfl = [&](const PJ& a, int d) {   // logical lane ll + d
        PJ r;
        r.x = shfl_down_fp(a.x, 2 * d); r.y = shfl_down_fp(a.y, 2 * d); r.z = shfl_down_fp(a.z, 2 * d);
        return r;
    };
    PJ run = ec::proj_inf<F>(), acc = ec::proj_inf<F>(), LP = ec::proj_inf<F>();
    const uint32_t s_scan = 2 * Lc, s_dbl = s_scan + 5, s_comb = s_dbl + logLc, s_end = s_comb + 6;
#pragma unroll 1
    for (uint32_t s = 0; s < s_end; s++) {
        PJ A, B;
        uint32_t dst;  // 0 run, 1 acc, 2 LP
        if (s < s_scan) {
            if ((s & 1u) == 0) {
                A = run; B = load_b(wp[Lc - 1 - (s >> 1)]); dst = 0;
            } else {
                A = acc; B = run; dst = 1;
            }
        } else if (s < s_dbl) {
            int d = 1 << (s - s_scan);
            A = run;
            B = ec::proj_select<F>(ll + d < 32, ec::proj_inf<F>(), shfl(run, d));
            dst = 0;
        } else if (s < s_comb) {
            if (s == s_dbl) LP = run;
            A = LP; B = LP; dst = 2;
        } else if (s == s_comb) {
            if (logLc == 0) LP = run;
            A = acc;
            B = ec::proj_select<F>(ll == 0, LP, ec::proj_inf<F>());
            dst = 1;
        } else {
            int d = 16 >> (s - s_comb - 1);
            A = acc;
            B = ec::proj_select<F>((int)ll < d, ec::proj_inf<F>(), shfl(acc, d));
            dst = 1;
        }
        ec::proj_add<F>(A, B);
        if (dst == 0) run = A;
        else if (dst == 1) acc = A;
        else LP = A;
    }
    if (ll == 0) {   // lanes 0 and 1: (X Z, Y Z^2, Z) of S and T, this lane's component; infinity (Z == 0) -> all-zero
        for (int k = 0; k < 2; k++) {
            const PJ& p = k == 0 ? run : acc;
            uint32_t* out = pairs + (size_t)chunk * 2 * Geo<G2C>::RAW_JAC + (size_t)k * Geo<G2C>::RAW_JAC + 12 * h;
            uint32_t zw[12];
            fp28::fp_to_blst(zw, p.z);
            uint32_t any = 0;
#pragma unroll
            for (int t = 0; t < 12; t++) any |= zw[t];
            any |= (uint32_t)__builtin_amdgcn_mov_dpp((int)any, 0xB1, 0xF, 0xF, true);   // either component non-zero
            Fp zz = F::sqr(p.z);
            fp_to_raw(out, F::mul(p.x, p.z), any != 0);
            fp_to_raw(out + 24, F::mul(p.y, zz), any != 0);
#pragma unroll
            for (int t = 0; t < 12; t++) out[48 + t] = any != 0 ? zw[t] : 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------- normalize_batch
// Jacobian -> affine for n points with ONE field inversion (Montgomery's trick as a product tree of fan-out NORM_K):
// replaces blstrs::G{1,2}Projective::batch_normalize behind CurveGroup::normalize_batch
// (/root/reference/src/g1.rs:537-543, src/g2.rs:517-523; the step arkworks provers run right before an MSM,
// `batch_convert_to_mul_base`, src/g1.rs:597-599).  Infinity (Z = 0) maps to the all-zero affine point.
//   k_norm_load : Z_i (raw) -> device form, infinity replaced by 1
//   k_norm_up   : per group of K values: exclusive prefix products + group total (= value of the next level)
//   (top level <= 64 values: inverted on the host, one Fermat inversion)
//   k_norm_down : per group: inverse of each value from the inverse of the group total
//   k_norm_final: x = X / Z^2, y = Y / Z^3, back to the reference's form
constexpr uint32_t NORM_K = 32;

template <class C>
__global__ void __launch_bounds__(256) k_norm_load(const uint32_t* __restrict__ raw_jac, uint32_t n, uint32_t* __restrict__ vals) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* zr = raw_jac + (size_t)i * Geo<C>::RAW_JAC + 2 * ElemIO<E>::RAW;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < ElemIO<E>::RAW; k++) any |= zr[k];
    E z;
    ElemIO<E>::from_raw(z, zr);
    z = F::select(any == 0, z, F::one());
    ElemIO<E>::store(vals + (size_t)i * Geo<C>::SLOT, z);
}

template <class C>
__global__ void __launch_bounds__(256) k_norm_up(const uint32_t* __restrict__ vals, uint32_t m, uint32_t* __restrict__ pref,
                                                 uint32_t* __restrict__ tot) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo = g * NORM_K, hi = lo + NORM_K < m ? lo + NORM_K : m;
    if (lo >= m) return;
    E run = F::one();
    for (uint32_t k = lo; k < hi; k++) {
        ElemIO<E>::store(pref + (size_t)k * Geo<C>::SLOT, run);
        E v;
        ElemIO<E>::load(v, vals + (size_t)k * Geo<C>::SLOT);
        run = F::mul(run, v);
    }
    ElemIO<E>::store(tot + (size_t)g * Geo<C>::SLOT, run);
}

template <class C>
__global__ void __launch_bounds__(256) k_norm_down(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pref,
                                                   const uint32_t* __restrict__ inv_tot, uint32_t m, uint32_t* __restrict__ inv_vals) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo = g * NORM_K, hi = lo + NORM_K < m ? lo + NORM_K : m;
    if (lo >= m) return;
    E I;
    ElemIO<E>::load(I, inv_tot + (size_t)g * Geo<C>::SLOT);
    for (uint32_t k = hi; k-- > lo;) {
        E p, v;
        ElemIO<E>::load(p, pref + (size_t)k * Geo<C>::SLOT);
        ElemIO<E>::load(v, vals + (size_t)k * Geo<C>::SLOT);
        ElemIO<E>::store(inv_vals + (size_t)k * Geo<C>::SLOT, F::mul(I, p));
        I = F::mul(I, v);
    }
}

template <class C>
__global__ void __launch_bounds__(256) k_norm_final(const uint32_t* __restrict__ raw_jac, const uint32_t* __restrict__ zinv, uint32_t n,
                                                    uint32_t* __restrict__ raw_aff) {
    using F = typename C::F;
    using E = typename F::E;
    constexpr int R = ElemIO<E>::RAW;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = raw_jac + (size_t)i * Geo<C>::RAW_JAC;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < R; k++) any |= q[2 * R + k];
    E x, y, zi;
    ElemIO<E>::from_raw(x, q);
    ElemIO<E>::from_raw(y, q + R);
    ElemIO<E>::load(zi, zinv + (size_t)i * Geo<C>::SLOT);
    E zi2 = F::mul(zi, zi);
    E zi3 = F::mul(zi2, zi);
    uint32_t* o = raw_aff + (size_t)i * Geo<C>::RAW_AFF;
    ElemIO<E>::to_raw(o, F::mul(x, zi2), any != 0);
    ElemIO<E>::to_raw(o + R, F::mul(y, zi3), any != 0);
}

// device form <-> reference form for a short vector of field elements (top of the product tree, host inversion)
template <class C>
__global__ void __launch_bounds__(64) k_elems_to_raw(const uint32_t* __restrict__ dev, uint32_t m, uint32_t* __restrict__ raw) {
    using E = typename C::F::E;
    uint32_t i = threadIdx.x;
    if (i >= m) return;
    E v;
    ElemIO<E>::load(v, dev + (size_t)i * Geo<C>::SLOT);
    ElemIO<E>::to_raw(raw + (size_t)i * ElemIO<E>::RAW, v, true);
}
template <class C>
__global__ void __launch_bounds__(64) k_elems_from_raw(const uint32_t* __restrict__ raw, uint32_t m, uint32_t* __restrict__ dev) {
    using E = typename C::F::E;
    uint32_t i = threadIdx.x;
    if (i >= m) return;
    E v;
    ElemIO<E>::from_raw(v, raw + (size_t)i * ElemIO<E>::RAW);
    ElemIO<E>::store(dev + (size_t)i * Geo<C>::SLOT, v);
}

// ---------------------------------------------------------------------------------------------- G1 point decoding
// Bulk CanonicalDeserialize + Valid::check for G1 (/root/reference/src/g1.rs:386-431): ZCash/IETF encoding
// (48-byte compressed / 96-byte uncompressed, big-endian, flag bits 0x80 compressed, 0x40 infinity, 0x20 y is the
// lexicographically larger root) -> blst_p1_affine, with per-point status instead of the reference's unwrap():
//   0 ok, 1 malformed encoding (flags, x >= p, no square root), 2 not on the curve, 3 not in the prime-order subgroup.
// Decompression: y = (x^3 + 4)^((p+1)/4).  Subgroup check (blstrs is_torsion_free [ext]) by the endomorphism test
// (beta x, y) == -[z^2](x, y), z = 0xd201000000010000 (M. Scott, eprint 2021/1130): two 64-bit double-and-add ladders
// on the complete projective formulas instead of a 255-bit multiplication by r.
__device__ __forceinline__ bool words_lt_p(const uint32_t (&w)[12]) {
    uint64_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint64_t v = (uint64_t)w[k] - fp28c::P32[k] - borrow;
        borrow = (v >> 32) & 1;
    }
    return borrow != 0;
}
__device__ __forceinline__ void be48_to_words(uint32_t (&w)[12], const uint8_t* b, uint32_t top_mask) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint8_t* q = b + 44 - 4 * k;  // word k = bytes [44-4k, 48-4k) big-endian
        w[k] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    w[11] &= top_mask;
}
// canonical integer of an internal value (< 50p), exact 28-bit limbs
__device__ __forceinline__ Fp fp_to_canonical(const Fp& a) {
    Fp one_int = fp28::fp_zero();
    one_int.l[0] = 1;
    return fp28::fp_canon_2p(fp28::fp_mul_call(a, one_int));
}
__device__ __forceinline__ bool fp_equal(const Fp& a, const Fp& b) {  // a == b (mod p), a, b < 15p
    return fp28::fp_is_zero_any(fp28::fp_sub<16>(a, b));
}

// r = [|z|] p on the complete formulas (63 doublings + 5 additions)
__device__ __noinline__ void g1_mul_z(ec::Proj<ec::FpOps>& r, const ec::Proj<ec::FpOps>& p) {
    using F = ec::FpOps;
    r = p;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        ec::Proj<F> c = r;
        ec::proj_add<F>(r, c);
        if ((fp28c::Z_ABS >> bit) & 1) ec::proj_add<F>(r, p);
    }
}

__global__ void __launch_bounds__(256) k_deserialize_g1(const uint8_t* __restrict__ bytes, uint32_t n, int compressed, int validate,
                                                        uint32_t* __restrict__ out_aff, uint8_t* __restrict__ status) {
    using F = ec::FpOps;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t size = compressed ? 48u : 96u;
    const uint8_t* b = bytes + (size_t)i * size;
    uint32_t* o = out_aff + (size_t)i * 24;
    uint8_t b0 = b[0];
    uint32_t c_flag = b0 >> 7, i_flag = (b0 >> 6) & 1, s_flag = (b0 >> 5) & 1;
    uint32_t xw[12], yw[12];
    be48_to_words(xw, b, 0x1fffffffu);
#pragma unroll
    for (int k = 0; k < 12; k++) yw[k] = 0;
    if (!compressed) be48_to_words(yw, b + 48, 0xffffffffu);
    uint8_t st = 0;
    bool is_inf = false;
    if (c_flag != (uint32_t)(compressed ? 1 : 0)) st = 1;
    if (st == 0 && i_flag) {
        uint32_t any = s_flag;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= xw[k] | yw[k];
        if (any) st = 1;
        is_inf = true;
    }
    if (st == 0 && !is_inf) {
        if (!words_lt_p(xw) || (!compressed && (!words_lt_p(yw) || s_flag))) st = 1;
    }
    Fp x = fp28::fp_zero(), y = fp28::fp_zero();
    if (st == 0 && !is_inf) {
        const Fp r2 = fp28::fp_const(fp28c::R2);
        x = fp28::fp_mul_call(fp28::fp_unpack384(xw), r2);                       // canonical integer -> internal
        Fp rhs = fp28::fp_add(fp28::fp_mul_call(fp28::fp_sqr_call(x), x), fp28::fp_const(fp28c::FOUR));   // x^3 + 4  < 4p
        bool on_curve;
        if (compressed) {
            // y = rhs^((p+1)/4): left-to-right square and multiply over the 379-bit exponent
            Fp acc = rhs;
#pragma unroll 1
            for (int bit = 377; bit >= 0; bit--) {  // top set bit of (p+1)/4 is bit 378
                acc = fp28::fp_sqr_call(acc);
                if ((fp28c::SQRT_EXP32[bit >> 5] >> (bit & 31)) & 1) acc = fp28::fp_mul_call(acc, rhs);
            }
            y = acc;
            on_curve = fp_equal(fp28::fp_sqr_call(y), rhs);
            if (!on_curve) st = 1;  // no square root: malformed compressed encoding
            // pick the root the sort flag asks for
            Fp yc = fp_to_canonical(y);
            bool larger = false, decided = false;
#pragma unroll
            for (int k = NL - 1; k >= 0; k--) {
                if (!decided && yc.l[k] != fp28c::HALF_P[k]) { larger = yc.l[k] > fp28c::HALF_P[k]; decided = true; }
            }
            if (larger != (s_flag != 0)) y = fp28::fp_neg<4>(y);
        } else {
            y = fp28::fp_mul_call(fp28::fp_unpack384(yw), r2);
            on_curve = !validate || fp_equal(fp28::fp_sqr_call(y), rhs);
            if (!on_curve) st = 2;
        }
        if (st == 0 && validate) {
            ec::Proj<F> p1 = ec::proj_from_affine<F>(x, y), q, q2;
            g1_mul_z(q, p1);
            g1_mul_z(q2, q);                                                     // [z^2] P
            // membership: (beta x, y) == -[z^2] P, i.e. X == beta x Z, Y == -y Z, Z != 0
            Fp bx = fp28::fp_mul_call(x, fp28::fp_const(fp28c::BETA));
            bool ok = !fp28::fp_is_zero_any(q2.z);
            ok = ok && fp_equal(q2.x, fp28::fp_mul_call(bx, q2.z));
            ok = ok && fp28::fp_is_zero_any(fp28::fp_add(q2.y, fp28::fp_mul_call(y, q2.z)));
            if (!ok) st = 3;
        }
    }
    uint32_t w[12];
    bool keep = st == 0 && !is_inf;
    fp28::fp_to_blst(w, x);
#pragma unroll
    for (int k = 0; k < 12; k++) o[k] = keep ? w[k] : 0u;
    fp28::fp_to_blst(w, y);
#pragma unroll
    for (int k = 0; k < 12; k++) o[12 + k] = keep ? w[k] : 0u;
    status[i] = st;
}

// affine (blst form) -> ZCash encoding
__global__ void __launch_bounds__(256) k_serialize_g1(const uint32_t* __restrict__ aff, uint32_t n, int compressed, uint8_t* __restrict__ bytes) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = aff + (size_t)i * 24;
    const uint32_t size = compressed ? 48u : 96u;
    uint8_t* b = bytes + (size_t)i * size;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < 24; k++) any |= q[k];
    Fp x, y;
    fp_from_raw(x, q);
    fp_from_raw(y, q + 12);
    Fp xc = fp_to_canonical(x), yc = fp_to_canonical(y);
    uint32_t xw[12], yw[12];
    fp28::fp_pack384(xw, xc);
    fp28::fp_pack384(yw, yc);
    bool larger = false, decided = false;
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        if (!decided && yc.l[k] != fp28c::HALF_P[k]) { larger = yc.l[k] > fp28c::HALF_P[k]; decided = true; }
    }
    for (uint32_t k = 0; k < size; k++) b[k] = 0;
    if (any == 0) {
        b[0] = compressed ? 0xC0 : 0x40;
        return;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint8_t* d = b + 44 - 4 * k;
        d[0] = (uint8_t)(xw[k] >> 24); d[1] = (uint8_t)(xw[k] >> 16); d[2] = (uint8_t)(xw[k] >> 8); d[3] = (uint8_t)xw[k];
        if (!compressed) {
            uint8_t* e = b + 48 + 44 - 4 * k;
            e[0] = (uint8_t)(yw[k] >> 24); e[1] = (uint8_t)(yw[k] >> 16); e[2] = (uint8_t)(yw[k] >> 8); e[3] = (uint8_t)yw[k];
        }
    }
    if (compressed) b[0] |= (uint8_t)(0x80 | (larger ? 0x20 : 0));
}

// ---------------------------------------------------------------------------------------------- G2 point decoding
// Same for G2 (/root/reference/src/g2.rs:338-411): 96-byte compressed / 192-byte uncompressed, coordinates in Fp2
// serialised c1 first; y^2 = x^3 + 4(1 + u).  Square root in Fp2 for p = 3 mod 4 (Adj, Rodriguez-Henriquez Alg. 9):
// a1 = a^((p-3)/4), alpha = a1^2 a, x0 = a1 a; root = u x0 if alpha = -1 else (1 + alpha)^((p-1)/2) x0.
// Subgroup test: psi(P) == [z] P (z < 0), psi(x, y) = (conj(x) PSI_X, conj(y) PSI_Y)  (M. Scott, eprint 2021/1130).
using G2F = ec::Fp2Ops;
__device__ __forceinline__ bool fp2_equal(const ec::Fp2& a, const ec::Fp2& b) { return fp_equal(a.c0, b.c0) && fp_equal(a.c1, b.c1); }
__device__ __forceinline__ bool fp2_is_zero(const ec::Fp2& a) { return fp28::fp_is_zero_any(a.c0) && fp28::fp_is_zero_any(a.c1); }
__device__ __forceinline__ ec::Fp2 fp2_conj(const ec::Fp2& a) { return ec::Fp2{a.c0, fp28::fp_neg<16>(a.c1)}; }

__device__ __noinline__ ec::Fp2 fp2_pow(const ec::Fp2& a, const uint32_t (&e)[12], int top_bit) {
    ec::Fp2 acc = a;
#pragma unroll 1
    for (int bit = top_bit - 1; bit >= 0; bit--) {
        acc = G2F::sqr(acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) acc = G2F::mul(acc, a);
    }
    return acc;
}
__device__ __noinline__ void g2_mul_z(ec::Proj<G2F>& r, const ec::Proj<G2F>& p) {
    r = p;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        ec::Proj<G2F> c = r;
        ec::proj_add<G2F>(r, c);
        if ((fp28c::Z_ABS >> bit) & 1) ec::proj_add<G2F>(r, p);
    }
}
__device__ __forceinline__ bool canon_gt_half(const Fp& c) {
    bool larger = false, decided = false;
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        if (!decided && c.l[k] != fp28c::HALF_P[k]) { larger = c.l[k] > fp28c::HALF_P[k]; decided = true; }
    }
    return larger;
}
__device__ __forceinline__ bool fp2_lex_largest(const ec::Fp2& y) {  // c1 first, then c0
    Fp c1 = fp_to_canonical(y.c1);
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) z |= c1.l[k];
    if (z != 0) return canon_gt_half(c1);
    return canon_gt_half(fp_to_canonical(y.c0));
}

__global__ void __launch_bounds__(256) k_deserialize_g2(const uint8_t* __restrict__ bytes, uint32_t n, int compressed, int validate,
                                                        uint32_t* __restrict__ out_aff, uint8_t* __restrict__ status) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t size = compressed ? 96u : 192u;
    const uint8_t* b = bytes + (size_t)i * size;
    uint32_t* o = out_aff + (size_t)i * 48;
    uint8_t b0 = b[0];
    uint32_t c_flag = b0 >> 7, i_flag = (b0 >> 6) & 1, s_flag = (b0 >> 5) & 1;
    uint32_t x1w[12], x0w[12], y1w[12], y0w[12];
    be48_to_words(x1w, b, 0x1fffffffu);
    be48_to_words(x0w, b + 48, 0xffffffffu);
#pragma unroll
    for (int k = 0; k < 12; k++) { y1w[k] = 0; y0w[k] = 0; }
    if (!compressed) {
        be48_to_words(y1w, b + 96, 0xffffffffu);
        be48_to_words(y0w, b + 144, 0xffffffffu);
    }
    uint8_t st = 0;
    bool is_inf = false;
    if (c_flag != (uint32_t)(compressed ? 1 : 0)) st = 1;
    if (st == 0 && i_flag) {
        uint32_t any = s_flag;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= x1w[k] | x0w[k] | y1w[k] | y0w[k];
        if (any) st = 1;
        is_inf = true;
    }
    if (st == 0 && !is_inf) {
        bool ok = words_lt_p(x1w) && words_lt_p(x0w);
        if (!compressed) ok = ok && words_lt_p(y1w) && words_lt_p(y0w) && !s_flag;
        if (!ok) st = 1;
    }
    ec::Fp2 x = G2F::zero(), y = G2F::zero();
    if (st == 0 && !is_inf) {
        const Fp r2 = fp28::fp_const(fp28c::R2), four = fp28::fp_const(fp28c::FOUR);
        x.c0 = fp28::fp_mul_call(fp28::fp_unpack384(x0w), r2);
        x.c1 = fp28::fp_mul_call(fp28::fp_unpack384(x1w), r2);
        ec::Fp2 rhs = G2F::add(G2F::mul(G2F::sqr(x), x), ec::Fp2{four, four});     // x^3 + 4(1 + u)   < 4p
        if (compressed) {
            ec::Fp2 a1 = fp2_pow(rhs, fp28c::EXP_P3_4_32, 378);                      // (p-3)/4 has its top bit at 378
            ec::Fp2 x0 = G2F::mul(a1, rhs);
            ec::Fp2 alpha = G2F::mul(a1, x0);
            ec::Fp2 ap1 = G2F::add(alpha, G2F::one());
            if (fp2_is_zero(ap1)) {
                y = ec::Fp2{fp28::fp_neg<4>(x0.c1), x0.c0};                          // u * x0
            } else {
                ec::Fp2 bb = fp2_pow(ap1, fp28c::EXP_P1_2_32, 379);                  // (p-1)/2: top bit 379
                y = G2F::mul(bb, x0);
            }
            if (!fp2_equal(G2F::sqr(y), rhs)) st = 1;                                // not a square: malformed
            if (fp2_lex_largest(y) != (s_flag != 0)) y = G2F::neg<4>(y);
        } else {
            y.c0 = fp28::fp_mul_call(fp28::fp_unpack384(y0w), r2);
            y.c1 = fp28::fp_mul_call(fp28::fp_unpack384(y1w), r2);
            if (validate && !fp2_equal(G2F::sqr(y), rhs)) st = 2;
        }
        if (st == 0 && validate) {
            ec::Proj<G2F> p1 = ec::proj_from_affine<G2F>(x, y), q;
            g2_mul_z(q, p1);                                                          // [|z|] P
            ec::Fp2 px = G2F::mul(fp2_conj(x), ec::Fp2{fp28::fp_zero(), fp28::fp_const(fp28c::PSI_X1)});
            ec::Fp2 py = G2F::mul(fp2_conj(y), ec::Fp2{fp28::fp_const(fp28c::PSI_Y0), fp28::fp_const(fp28c::PSI_Y1)});
            // psi(P) == [z] P = -[|z|] P :  X_q == px Z_q,  Y_q == -py Z_q,  Z_q != 0
            bool ok = !fp2_is_zero(q.z);
            ok = ok && fp2_equal(q.x, G2F::mul(px, q.z));
            ok = ok && fp2_is_zero(G2F::add(q.y, G2F::mul(py, q.z)));
            if (!ok) st = 3;
        }
    }
    bool keep = st == 0 && !is_inf;
    ElemIO<ec::Fp2>::to_raw(o, x, keep);
    ElemIO<ec::Fp2>::to_raw(o + 24, y, keep);
    status[i] = st;
}

__device__ __forceinline__ void words_to_be48(uint8_t* d, const uint32_t (&w)[12]) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint8_t* q = d + 44 - 4 * k;
        q[0] = (uint8_t)(w[k] >> 24); q[1] = (uint8_t)(w[k] >> 16); q[2] = (uint8_t)(w[k] >> 8); q[3] = (uint8_t)w[k];
    }
}
__global__ void __launch_bounds__(256) k_serialize_g2(const uint32_t* __restrict__ aff, uint32_t n, int compressed, uint8_t* __restrict__ bytes) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = aff + (size_t)i * 48;
    const uint32_t size = compressed ? 96u : 192u;
    uint8_t* b = bytes + (size_t)i * size;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < 48; k++) any |= q[k];
    for (uint32_t k = 0; k < size; k++) b[k] = 0;
    if (any == 0) {
        b[0] = compressed ? 0xC0 : 0x40;
        return;
    }
    ec::Fp2 x, y;
    ElemIO<ec::Fp2>::from_raw(x, q);
    ElemIO<ec::Fp2>::from_raw(y, q + 24);
    uint32_t w[12];
    fp28::fp_pack384(w, fp_to_canonical(x.c1)); words_to_be48(b, w);
    fp28::fp_pack384(w, fp_to_canonical(x.c0)); words_to_be48(b + 48, w);
    if (!compressed) {
        fp28::fp_pack384(w, fp_to_canonical(y.c1)); words_to_be48(b + 96, w);
        fp28::fp_pack384(w, fp_to_canonical(y.c0)); words_to_be48(b + 144, w);
    } else {
        b[0] |= (uint8_t)(0x80 | (fp2_lex_largest(y) ? 0x20 : 0));
    }
}

// ---------------------------------------------------------------------------------------------- field test hook
__global__ void __launch_bounds__(256) k_test_fp_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                    uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t aw[12], bw[12], ow[12];
#pragma unroll
    for (int k = 0; k < 12; k++) { aw[k] = a[(size_t)i * 12 + k]; bw[k] = b[(size_t)i * 12 + k]; }
    Fp x = fp28::fp_from_blst(aw), y = fp28::fp_from_blst(bw), z;
    if (op == 0) z = fp28::fp_mul_call(x, y);
    else if (op == 1) z = fp28::fp_sqr(x);
    else if (op == 2) z = fp28::fp_add(x, y);
    else z = fp28::fp_sub<4>(x, y);
    
    fp28::fp_to_blst(ow, z);
#pragma unroll
    for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = ow[k];
}

}  // namespace msmk
