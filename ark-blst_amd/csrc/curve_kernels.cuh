// Curve-generic kernels of the MSM pipeline (ingest, accumulate, merge, reduce, combine) and normalize_batch.
//
// Launch-bounds rule (csrc/Makefile, DESIGN_HISTORY.md §9 "call-ABI miscompare"): a kernel whose body or callees contain a CALL of an
// out-of-line device function is declared with two waves per SIMD (second __launch_bounds__ argument 2): at most 256 registers per
// lane, hence no AGPRs — ROCm 7.2's hipcc miscompiles calls combined with VGPR spills into AGPRs.  Kernels built for one wave per
// SIMD (512 registers: PairG2 reduce, the Fp12 kernels) contain no call.  tests/test_cabi.py checks both on the shipped code objects.
#pragma once
#include "kernels_common.cuh"

namespace msmk {

// ---------------------------------------------------------------------------------------------- ingest
// raw: n affine points in the reference's form.  One thread per point.
template <class C>
__global__ void __launch_bounds__(256, 2) k_ingest(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out,
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
// point is not staged in registers (28 more registers spill: 3.06 vs 2.88 ms); its index is fetched one iteration ahead and
// the other resident wave covers the load latency — completely, as it turned out: routing the next point through LDS
// with global_load_lds_dwordx4 (no registers, load overlapped with the running addition; tools/probe/lds_load_probe.hip)
// left the kernel time unchanged (2.38 vs 2.38 ms), the SIMDs are issue-bound either way.
// A lane that meets an exceptional pair (same x) leaves the hot loop and finishes on the complete formulas.
// Output: partial[i] (projective), i = natural item id.
// Workgroups of ONE wave: a CU refills a wave slot the moment a wave retires instead of waiting for the four slots a 256-thread
// workgroup needs (same-box A/B at 2^20 points: 2.30 vs 2.33 ms; nothing at 2^24, nothing for G2).
// Entries [e, end) of item k of bucket b.  `packed` = log2 T | log2 S << 16 (the schedule's item geometry, sort_kernels.cuh items_of): a bucket
// of up to T entries is one item, a fuller one is cut into items of S entries.
__device__ __forceinline__ void item_range(const uint32_t* __restrict__ offsets, uint32_t b, uint32_t k, uint32_t packed, uint32_t& e, uint32_t& end) {
    const uint32_t logT = packed & 0xffu, logS = (packed >> 16) & 0xffu;
    const uint32_t beg = offsets[b], bend = offsets[b + 1];
    const uint32_t lg = item_size_log(bend - beg, logT, logS);
    e = beg + (k << lg);
    end = e + (1u << lg) < bend ? e + (1u << lg) : bend;
}

template <class C>
__global__ void __launch_bounds__(64, C::OCC) k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ woff,
                                                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ item_bucket,
                                                            uint32_t* __restrict__ meta, uint32_t logT, uint32_t* __restrict__ partial) {
    using F = typename C::F;
    using FA = typename C::FA;
    using E = typename F::E;
    uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= meta[0]) return;       // the item count stays on the device: the grid is the host's upper bound (run_msm), surplus waves leave here
    uint32_t i = order[j];          // items are processed longest class first; partial[] keeps natural item order
    WaveClock::start(partial + (size_t)i * Geo<C>::BK_WORDS);
    uint32_t b = item_bucket[i];
    uint32_t k = i - woff[b];
    uint32_t e, end;
    item_range(offsets, b, k, logT, e, end);
    // the first entry only initialises the running sum: peeled off the loop, which then has no "still empty" branch
    ec::Xyzz<FA> acc;
    const bool inf = e >= end;
    {
        uint32_t ent = inf ? 0u : sorted[e];
        E x, y;
        load_point<C>(x, y, bases, ent);
        y = F::select((ent >> 31) != 0, y, FA::template neg_l<4>(y));
        acc.x = x; acc.y = FA::norm(y); acc.zz = F::one(); acc.zzz = F::one();   // acc.y must be subtractable: N-form
        if (!inf) e++;
    }
    uint32_t nent = e < end ? sorted[e] : 0u;
    while (e < end) {
        uint32_t ent = nent;
        E x, y;
        load_point<C>(x, y, bases, ent);
        if (e + 1 < end) {
            nent = sorted[e + 1];
        }
        y = F::select((ent >> 31) != 0, y, FA::template neg_l<4>(y));
        if (ec::xyzz_madd<FA>(acc, x, y)) break;  // exceptional pair at entry e: acc untouched
        e++;
    }
    ec::Proj<F> out = ec::proj_inf<F>();
    if (!inf) out = ec::xyzz_to_proj<F>(acc);   // Xyzz<FA> and Xyzz<F> are one type (ec.cuh)
    while (e < end) {  // cold path (never taken on random inputs): complete additions
        uint32_t ent = sorted[e];
        E x, y;
        load_point<C>(x, y, bases, ent);
        y = F::select((ent >> 31) != 0, y, F::template neg<4>(y));
        ec::Proj<F> q = ec::proj_from_affine<F>(x, y);
        ec::proj_add<F>(out, q);  // shared-call multiplier: keeps the cold path out of the hot loop's register budget
        e++;
    }
    WaveClock::stop(partial + (size_t)i * Geo<C>::BK_WORDS, meta);
    store_bucket<C>(partial + (size_t)i * Geo<C>::BK_WORDS, out);
}

// G2 accumulate with TWO lanes per work item (CoopF2A: the even lane holds c0 and the odd lane c1 of every Fp2 value; a product
// exchanges the partner's components by DPP and is one fused reduction per lane).  Same schedule, same entries and the same
// formulas as k_accumulate<G2C>; the accumulator is 4 x 14 registers per lane instead of 8 x 14.  The hot loop runs without
// scratch since round 3 (product-scanning multiplier, first entry peeled, tail out of line: 35 spilled registers -> 0, which
// tests/test_cabi.py::test_hot_kernels_do_not_spill reads off the shipped code object).  Cold path (exceptional pairs):
// complete additions on the same lane pair through ONE out-of-line body.
static __device__ __noinline__ void coop_add_inplace(ec::Proj<CoopF2>& a, const ec::Proj<CoopF2>& b) { ec::proj_add<CoopF2>(a, b); }

// Everything after the hot loop of k_accumulate_g2_coop as ONE out-of-line body: XYZZ -> projective, the cold path (complete
// additions for the entries from an exceptional pair on; never taken on random inputs) and the store.  Out of line so that none
// of its values (a second point, the projective sum, the shared-call operands) is live inside the hot loop: with the tail
// inlined the loop spilled 35 registers (2.8 GB of scratch writes per launch at 2^20 points, profiles/r02c_g2_2p20_pmc_summary.json).
static __device__ __noinline__ void g2_coop_finish(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, uint32_t e,
                                                   uint32_t end, bool inf, const ec::Xyzz<CoopF2>& acc, uint32_t* __restrict__ o) {
    const uint32_t h = threadIdx.x & 1u;
    ec::Proj<CoopF2> out = ec::proj_inf<CoopF2>();
    if (!inf) out = ec::xyzz_to_proj<CoopF2>(acc);
    while (e < end) {
        uint32_t ent = sorted[e];
        Fp x, y;
        const uint32_t* p = bases + (size_t)(ent & 0x7fffffffu) * G2_PT_WORDS + 16 * h;
        load_fp16(x, p);
        load_fp16(y, p + 32);
        y = fp28::fp_select((ent >> 31) != 0, y, fp28::fp_neg<4>(y));
        ec::Proj<CoopF2> q = ec::proj_from_affine<CoopF2>(x, y);
        coop_add_inplace(out, q);
        e++;
    }
    store_fp16(o, out.x); store_fp16(o + 32, out.y); store_fp16(o + 64, out.z);   // x | y | z, each (c0, c1) in 16-word slots
}

template <class C>   // C = G2C (a template so that only the G2 translation unit instantiates it)
__global__ void __launch_bounds__(64, 2) k_accumulate_g2_coop(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ woff,
                                                               const uint32_t* __restrict__ order, const uint32_t* __restrict__ item_bucket,
                                                               uint32_t* __restrict__ meta, uint32_t logT, uint32_t* __restrict__ partial) {
    using FA = CoopF2A;
    const uint32_t h = threadIdx.x & 1u;
    uint32_t j = (blockIdx.x * 64 + threadIdx.x) >> 1;
    if (j >= meta[0]) return;       // pairs never straddle the bound (even block size); the grid is an upper bound, as for k_accumulate
    uint32_t i = order[j];
    WaveClock::start(partial + (size_t)i * G2_BK_WORDS);
    uint32_t b = item_bucket[i];
    uint32_t k = i - woff[b];
    uint32_t e, end;
    item_range(offsets, b, k, logT, e, end);
    auto load_comp = [&](Fp& x, Fp& y, uint32_t ent) {   // this lane's component of x and y: slots x.c0 | x.c1 | y.c0 | y.c1
        const uint32_t* p = bases + (size_t)(ent & 0x7fffffffu) * G2_PT_WORDS + 16 * h;
        load_fp16(x, p);
        load_fp16(y, p + 32);
    };
    // the first entry only initialises the running sum: peeled off the loop (inside it the lane-dependent constant of
    // FA::one() was hoisted into 14 loop-invariant registers, which the loop then spilled)
    ec::Xyzz<FA> acc;
    const bool inf = e >= end;
    {
        uint32_t ent = inf ? 0u : sorted[e];
        Fp x, y;
        load_comp(x, y, ent);
        y = fp28::fp_select((ent >> 31) != 0, y, fp28::fp_neg<4>(y));
        acc.x = x; acc.y = y; acc.zz = FA::one(); acc.zzz = FA::one();
        if (!inf) e++;
    }
    uint32_t nent = e < end ? sorted[e] : 0u;
    while (e < end) {
        uint32_t ent = nent;
        Fp x, y;
        load_comp(x, y, ent);
        if (e + 1 < end) {
            nent = sorted[e + 1];
        }
        y = fp28::fp_select((ent >> 31) != 0, y, fp28::fp_neg<4>(y));
        if (ec::xyzz_madd<FA>(acc, x, y)) break;  // exceptional pair at entry e (pair-wide decision): acc untouched
        e++;
    }
    // the tail takes a COPY made after the loop: passing `acc` itself by reference makes the loop variable address-taken, and the
    // compiler then keeps it in scratch and writes all 224 bytes back on every iteration (7.9 GB of scratch writes per launch at
    // 2^20 points, profiles/r03_g2_2p20_* first take) although the code object reports no spill
    ec::Xyzz<CoopF2> fin;
    fin.x = acc.x; fin.y = acc.y; fin.zz = acc.zz; fin.zzz = acc.zzz;
    WaveClock::stop(partial + (size_t)i * G2_BK_WORDS, meta);
    g2_coop_finish(bases, sorted, e, end, inf, fin, partial + (size_t)i * G2_BK_WORDS + 16 * h);
}

// One level of the per-bucket merge of split buckets, fan-in MERGE_FAN: partial[i] += partial[i + d] + partial[i + 2d] + ... for the listed
// items i, whose index k inside their bucket is a multiple of MERGE_FAN * d.  Levels d = 1, FAN, FAN^2, .. until d >= max items; then
// partial[woff[b]] is bucket b.  Launched only when some bucket was split (meta[1] > 1).  The list of level 0 comes from the schedule
// (k_sched3: every FAN-th item of every split bucket); each level appends the items that accumulate again at the next one (k a
// multiple of FAN^2 d with something left to add) to the next level's list, so every level's launch is dense whatever the bucket sizes
// are (a level that skipped through ONE list left one active lane per wave from the second level on: 0.9 ms per level for a bucket of
// 2^20 entries).  One LOGICAL lane of the combine's lane scheme per listed item (round 4: it was one lane per item and a binary tree —
// 14 launches of a 25-us single-lane addition for such a bucket; a quad-lane addition takes 4-5 us).  Every lane of a wave runs the
// same number of additions (the lane schemes exchange operands across lanes): idle ones add infinity.  grid = the host's bound of the
// list length; the length itself is read from *count_in.
template <class CS>
__global__ void __launch_bounds__(64, CS::MAX_OCC) k_merge(uint32_t* __restrict__ partial, const uint32_t* __restrict__ item_bucket,
                                                           const uint32_t* __restrict__ woff, const uint32_t* __restrict__ list_in,
                                                           const uint32_t* __restrict__ count_in, uint32_t* __restrict__ list_out,
                                                           uint32_t* __restrict__ count_out, uint32_t d) {
    aux_priority();
    using Pt = typename CS::Pt;
    constexpr int NLL = 1 << CS::LOG_LL, BK = Geo<typename CS::C>::BK_WORDS;
    const uint32_t slot = blockIdx.x * NLL + CS::ll();
    const bool have = slot < *count_in;
    if (__ballot(have) == 0) return;
    const uint32_t i = list_in[have ? slot : 0];
    const uint32_t b = item_bucket[i];
    const uint32_t n = woff[b + 1] - woff[b], k = i - woff[b];
    const uint32_t room = have && n - 1 - k >= d ? (n - 1 - k) / d : 0u;            // items k + d, k + 2 d, .. that exist
    const uint32_t m = room < MERGE_FAN - 1 ? room : MERGE_FAN - 1;                  // addends of this level
    Pt acc = CS::load(partial + (size_t)i * BK);
#pragma unroll 1
    for (uint32_t t = 1; t < MERGE_FAN; t++) {
        if (__ballot(t <= m) == 0) break;
        Pt nb = CS::select(t <= m, CS::inf(), CS::load(partial + (size_t)(i + (t <= m ? t * d : 0u)) * BK));
        CS::add(acc, nb);
    }
    if (m) CS::store(partial + (size_t)i * BK, acc);
    // accumulates again at the next level (stride FAN d): first lane of the logical lane appends the item
    const uint64_t fd = (uint64_t)MERGE_FAN * d;
    if (have && (threadIdx.x & (64 / NLL - 1)) == 0 && k % (MERGE_FAN * fd) == 0 && (uint64_t)k + fd < n) list_out[atomicAdd(count_out, 1u)] = i;
}

// ---------------------------------------------------------------------------------------------- precomputed tables
// T_j[i] = 2^c T_{j-1}[i] for a resident base set (mi_msm_g{1,2}_set_bases_precomputed): c complete doublings per point, then one
// batch inversion (the product tree of normalize_batch) back to affine.  Neither curve group has points of even order (both
// cofactors are odd), so a multiple of a finite point is finite; a Z of zero (input not on the curve) is kept out of the
// shared inversion and only spoils its own entry.
template <class C>
__global__ void __launch_bounds__(256, 2) k_table_dbl(const uint32_t* __restrict__ prev, const uint8_t* __restrict__ inf_flags, uint32_t n, uint32_t c,
                                                   uint32_t* __restrict__ proj, uint32_t* __restrict__ vals) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    E x, y;
    load_point<C>(x, y, prev, i);
    ec::Proj<F> p = ec::proj_from_affine<F>(x, y);
#pragma unroll 1
    for (uint32_t k = 0; k < c; k++) {
        ec::Proj<F> q = p;
        add_inplace(p, q);
    }
    store_bucket<C>(proj + (size_t)i * Geo<C>::BK_WORDS, p);
    bool skip = inf_flags[i] != 0 || F::is_zero_2p(p.z);
    ElemIO<E>::store(vals + (size_t)i * Geo<C>::SLOT, F::select(skip, p.z, F::one()));
}
template <class C>
__global__ void __launch_bounds__(256, 2) k_table_affine(const uint32_t* __restrict__ proj, const uint32_t* __restrict__ zinv,
                                                      const uint8_t* __restrict__ inf_flags, uint32_t n, uint32_t* __restrict__ out) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ec::Proj<F> p = load_bucket<C>(proj + (size_t)i * Geo<C>::BK_WORDS);
    E zi;
    ElemIO<E>::load(zi, zinv + (size_t)i * Geo<C>::SLOT);
    uint32_t* o = out + (size_t)i * Geo<C>::PT_WORDS;
    ElemIO<E>::store(o, F::mul(p.x, zi));
    ElemIO<E>::store(o + Geo<C>::SLOT, F::mul(p.y, zi), inf_flags[i] != 0 ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------- reduce + combine
// Bucket reduction with SEVERAL LANES PER LOGICAL LANE.  A complete projective addition is 12 multiplications on one
// lane (5200 instructions, 17 us for a lone wave); the reduction is a dependent chain of them, so the step time sets the
// kernel time.  A "coop scheme" CS splits one addition over the lanes of a small lane group:
//   QuadG1   (G1): the lanes 0..2 of a quad hold X, Y, Z of a point (lane 3 mirrors lane 2).  Step 1: lane q computes
//            own_q = a_q b_q and cross_q = (a_q + a_q')(b_q + b_q'), q' = q+1 mod 3 (partner coordinates by DPP quad_perm);
//            step 2: the six values are broadcast inside the quad and lane q computes output coordinate q as ONE fused
//            two-product reduction.  2 multiplications + 1 fused pair per lane instead of 6 + 3: ~1800 instructions per step,
//            a third of the registers (two waves per SIMD), 16 logical lanes per wave.
//   PairG2   (G2): the even lane holds c0, the odd lane c1 of every Fp2 coordinate (CoopF2, coop_fp2.cuh): 32 logical lanes.
// A scheme provides: LOG_LL (log2 logical lanes per wave), LPL (lanes per logical lane), Pt (per-lane state of a point),
// inf / add / select / load (device bucket layout) / store / shfl_down / bcast0 / store_jac_raw.
struct QuadG1 {
    using C = G1C;
    static constexpr int LOG_LL = 4, LPL = 4, MAX_OCC = 2;
    struct Pt { Fp c; };
    static __device__ __forceinline__ uint32_t q() { return threadIdx.x & 3u; }
    static __device__ __forceinline__ uint32_t ll() { return (threadIdx.x & 63u) >> 2; }
    template <int CTRL>
    static __device__ __forceinline__ Fp dpp(const Fp& a) {
        Fp r;
#pragma unroll
        for (int k = 0; k < NL; k++) r.l[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[k], CTRL, 0xF, 0xF, true);
        return r;
    }
    static constexpr int NEXT = 0xC9;                                    // quad_perm [1,2,0,3]: lane q reads coordinate q+1 mod 3
    static constexpr int B0 = 0x00, B1 = 0x55, B2 = 0xAA;                // broadcast of lane 0 / 1 / 2 inside the quad
    static __device__ __forceinline__ Pt inf() { return Pt{fp28::fp_select(q() == 1, fp28::fp_zero(), fp28::fp_one())}; }   // (0 : 1 : 0)
    static __device__ __forceinline__ Pt select(bool take_b, const Pt& a, const Pt& b) { return Pt{fp28::fp_select(take_b, a.c, b.c)}; }
    static __device__ __forceinline__ Pt load(const uint32_t* bucket) {   // 3 slots of 16 words: X | Y | Z
        uint32_t s = q() < 2 ? q() : 2;
        Pt r;
        load_fp16(r.c, bucket + 16 * s);
        return r;
    }
    static __device__ __forceinline__ void store(uint32_t* bucket, const Pt& p) {
        if (q() < 3) store_fp16(bucket + 16 * q(), p.c);
    }
    static __device__ __forceinline__ Pt shfl_down(const Pt& a, int d) { return Pt{shfl_down_fp(a.c, 4 * d)}; }   // logical lane ll + d
    static __device__ __forceinline__ Pt from_ll(const Pt& a, int src) {                                           // value of logical lane src
        Pt r;
#pragma unroll
        for (int k = 0; k < NL; k++) r.c.l[k] = __shfl(a.c.l[k], 4 * src + (int)q(), 64);
        return r;
    }
    // a <- a + b, complete (RCB16 Alg. 7, the formulas of ec::proj_add; the same values, hence the same value bounds).
    // Linear operations are lazy where the limb sizes allow (coordinates come in with exact limbs < 2^28, every multiplier
    // output is exact): a carry pass costs 42 instructions and the first version spent ten of them per addition, four remain.
    // Limb sizes (column sums of the 64-bit accumulators must stay below 2^63): sums of two exact values < 2^29; 3x < 2^29.6;
    // a - b + 32p with b in N-form < 2^29.6; 16p - b < 2^29; N-form <= 2^28 + 64.  Largest accumulation: t1m t3 + t5 (16p - t4)
    // = 14 (2^57.6 + 2^57) + 2^59.8 < 2^62.2.
    static __device__ __forceinline__ void add(Pt& a, const Pt& b) {
        using namespace fp28;
        const bool is2 = q() >= 2, is0 = q() == 0;
        Fp an = dpp<NEXT>(a.c), bn = dpp<NEXT>(b.c);
        Fp own = fp_mul(a.c, b.c);                                                       // t0 | t1 | t2                          exact
        Fp cross = fp_mul(fp_add_lazy(a.c, an), fp_add_lazy(b.c, bn));                   // t3 | t4 | t5    (2^29 x 2^29 limbs)   exact
        cross = fp_norm(fp_sub8_lazy_wide(cross, fp_add_lazy(own, dpp<NEXT>(own))));     // X1Y2+X2Y1 | Y1Z2+Y2Z1 | X1Z2+X2Z1     N-form, < 10p
        Fp t0 = dpp<B0>(own);
        t0 = fp_add_lazy(fp_add_lazy(t0, t0), t0);                                       // 3 X1X2                                < 2^29.6, < 6p
        Fp t1 = dpp<B1>(own);
        Fp t2 = fp_mul_small<12>(dpp<B2>(own));                                          // b3 Z1Z2                               N-form, < 24p
        Fp u = fp_add(t1, t2);                                                           // Y1Y2 + b3 Z1Z2                        N-form
        Fp t1m = fp_sub_lazy<32>(t1, t2);                                                // Y1Y2 - b3 Z1Z2                        < 2^29.6, < 34p
        Fp t3 = dpp<B0>(cross), t4 = dpp<B1>(cross);
        Fp t5 = fp_mul_small<12>(dpp<B2>(cross));                                        // b3 (X1Z2 + X2Z1)                      N-form
        // X3 = t1m t3 - t5 t4 ; Y3 = t1m u + t5 t0 ; Z3 = u t4 + t0 t3
        Fp A1 = fp_select(is2, t1m, u);
        Fp B1v = fp_select(is0, fp_select(is2, u, t4), t3);
        Fp A2 = fp_select(is2, t5, t0);
        Fp B2v = fp_select(is0, fp_select(is2, t0, t3), fp_sub_lazy<16>(fp_zero(), t4));
        a.c = fp_mul2add(A1, B1v, A2, B2v);
    }
    // logical lane's point -> Jacobian in the reference's form (X Z, Y Z^2, Z); infinity (Z == 0 mod p) -> all-zero
    static __device__ __forceinline__ void store_jac_raw(uint32_t* out, const Pt& p) {
        using namespace fp28;
        Fp z = dpp<B2>(p.c);
        Fp zz = fp_sqr(z);
        Fp v = fp_select(q() >= 2, fp_mul(p.c, fp_select(q() == 0, zz, z)), p.c);
        uint32_t w[12], any = 0;
        fp_to_blst(w, v);
#pragma unroll
        for (int k = 0; k < 12; k++) any |= w[k];
        any = (uint32_t)__builtin_amdgcn_mov_dpp((int)any, B2, 0xF, 0xF, true);   // Z's words
        if (q() < 3) {
#pragma unroll
            for (int k = 0; k < 12; k++) out[12 * q() + k] = any != 0 ? w[k] : 0u;
        }
    }
};

struct PairG2 {
    using C = G2C;
    using F = CoopF2;
    static constexpr int LOG_LL = 5, LPL = 2, MAX_OCC = 1;
    using Pt = ec::Proj<F>;
    static __device__ __forceinline__ uint32_t h() { return threadIdx.x & 1u; }
    static __device__ __forceinline__ uint32_t ll() { return (threadIdx.x & 63u) >> 1; }
    static __device__ __forceinline__ Pt inf() { return ec::proj_inf<F>(); }
    static __device__ __forceinline__ Pt select(bool take_b, const Pt& a, const Pt& b) { return ec::proj_select<F>(take_b, a, b); }
    static __device__ __forceinline__ Pt load(const uint32_t* bucket) {   // x | y | z, each (c0, c1) in 16-word slots
        const uint32_t* p = bucket + 16 * h();
        Pt r;
        load_fp16(r.x, p); load_fp16(r.y, p + 32); load_fp16(r.z, p + 64);
        return r;
    }
    static __device__ __forceinline__ void store(uint32_t* bucket, const Pt& p) {
        uint32_t* o = bucket + 16 * h();
        store_fp16(o, p.x); store_fp16(o + 32, p.y); store_fp16(o + 64, p.z);
    }
    static __device__ __forceinline__ Pt shfl_down(const Pt& a, int d) {
        Pt r;
        r.x = shfl_down_fp(a.x, 2 * d); r.y = shfl_down_fp(a.y, 2 * d); r.z = shfl_down_fp(a.z, 2 * d);
        return r;
    }
    static __device__ __forceinline__ Fp from_fp(const Fp& a, int src) {
        Fp r;
#pragma unroll
        for (int k = 0; k < NL; k++) r.l[k] = __shfl(a.l[k], 2 * src + (int)h(), 64);
        return r;
    }
    static __device__ __forceinline__ Pt from_ll(const Pt& a, int src) { return Pt{from_fp(a.x, src), from_fp(a.y, src), from_fp(a.z, src)}; }
    static __device__ __forceinline__ void add(Pt& a, const Pt& b) { ec::proj_add<F>(a, b); }
    static __device__ __forceinline__ void store_jac_raw(uint32_t* out, const Pt& p) {   // this lane's component of (X Z, Y Z^2, Z)
        uint32_t* o = out + 12 * h();
        uint32_t zw[12];
        fp28::fp_to_blst(zw, p.z);
        uint32_t any = 0;
#pragma unroll
        for (int t = 0; t < 12; t++) any |= zw[t];
        any |= (uint32_t)__builtin_amdgcn_mov_dpp((int)any, 0xB1, 0xF, 0xF, true);   // either component non-zero
        Fp zz = F::sqr(p.z);
        fp_to_raw(o, F::mul(p.x, p.z), any != 0);
        fp_to_raw(o + 24, F::mul(p.y, zz), any != 0);
#pragma unroll
        for (int t = 0; t < 12; t++) o[48 + t] = any != 0 ? zw[t] : 0u;
    }
};
// G2 with EIGHT lanes per logical lane: lane = 8 l + 2 q + h holds component h (c0 / c1) of coordinate q (X, Y, Z; q = 3 idles on
// bounded garbage).  The Fp2 products run on lane pairs (CoopF2), the coordinates are exchanged across the eight lanes by
// ds_bpermute.  Per step and lane: two Fp2 products + two multiplications by b3 + one fused pair, ~3800 instructions instead of
// the ~9500 of PairG2, and a third of the registers (two waves per SIMD); 8 logical lanes per wave.
struct OctG2 {
    using C = G2C;
    using F = CoopF2;
    static constexpr int LOG_LL = 3, LPL = 8, MAX_OCC = 2;
    struct Pt { Fp c; };
    static __device__ __forceinline__ uint32_t h() { return threadIdx.x & 1u; }
    static __device__ __forceinline__ uint32_t q() { return (threadIdx.x >> 1) & 3u; }
    static __device__ __forceinline__ uint32_t ll() { return (threadIdx.x & 63u) >> 3; }
    static __device__ __forceinline__ Fp from_lane(const Fp& a, int src) {
        Fp r;
#pragma unroll
        for (int k = 0; k < NL; k++) r.l[k] = __shfl(a.l[k], src, 64);
        return r;
    }
    static __device__ __forceinline__ int lane_of(uint32_t coord) { return (int)((threadIdx.x & 56u) | (coord << 1) | h()); }   // same group, same component
    static __device__ __forceinline__ Fp coord(const Fp& a, uint32_t k) { return from_lane(a, lane_of(k)); }
    static __device__ __forceinline__ Fp next(const Fp& a) { uint32_t qq = q() < 2 ? q() + 1 : 0; return from_lane(a, lane_of(qq)); }
    static __device__ __forceinline__ Pt inf() { return Pt{fp28::fp_select(q() == 1 && h() == 0, fp28::fp_zero(), fp28::fp_one())}; }   // (0 : 1 : 0)
    static __device__ __forceinline__ Pt select(bool take_b, const Pt& a, const Pt& b) { return Pt{fp28::fp_select(take_b, a.c, b.c)}; }
    static __device__ __forceinline__ Pt load(const uint32_t* bucket) {   // x | y | z, each (c0, c1) in 16-word slots
        uint32_t s = q() < 2 ? q() : 2;
        Pt r;
        load_fp16(r.c, bucket + 32 * s + 16 * h());
        return r;
    }
    static __device__ __forceinline__ void store(uint32_t* bucket, const Pt& p) {
        if (q() < 3) store_fp16(bucket + 32 * q() + 16 * h(), p.c);
    }
    static __device__ __forceinline__ Pt shfl_down(const Pt& a, int d) { return Pt{shfl_down_fp(a.c, 8 * d)}; }
    static __device__ __forceinline__ Pt from_ll(const Pt& a, int src) { return Pt{from_lane(a.c, 8 * src + (int)(threadIdx.x & 7u))}; }
    // a <- a + b, complete (RCB16 Alg. 7): the structure of QuadG1::add over Fp2 components
    static __device__ __forceinline__ void add(Pt& a, const Pt& b) {
        const bool is2 = q() >= 2, is0 = q() == 0;
        Fp an = next(a.c), bn = next(b.c);
        Fp own = F::mul(a.c, b.c);                                           // t0 | t1 | t2
        Fp cross = F::mul(F::add(a.c, an), F::add(b.c, bn));                 // t3 | t4 | t5
        cross = F::sub<8>(cross, F::add(own, next(own)));                    // X1Y2+X2Y1 | Y1Z2+Y2Z1 | X1Z2+X2Z1
        Fp t0 = F::mul3(coord(own, 0));                                      // 3 X1X2
        Fp t1 = coord(own, 1);
        Fp t2 = F::mul_b3(coord(own, 2));                                    // b3 Z1Z2
        Fp u = F::add(t1, t2);
        Fp t1m = F::sub<32>(t1, t2);
        Fp t3 = coord(cross, 0), t4 = coord(cross, 1);
        Fp t5 = F::mul_b3(coord(cross, 2));
        // X3 = t1m t3 - t5 t4 ; Y3 = t1m u + t5 t0 ; Z3 = u t4 + t0 t3
        Fp A1 = fp28::fp_select(is2, t1m, u);
        Fp B1v = fp28::fp_select(is0, fp28::fp_select(is2, u, t4), t3);
        Fp A2 = fp28::fp_select(is2, t5, t0);
        Fp B2v = fp28::fp_select(is0, fp28::fp_select(is2, t0, t3), F::neg<16>(t4));
        a.c = CoopF2A::mul2add(A1, B1v, A2, B2v);
    }
    static __device__ __forceinline__ void store_jac_raw(uint32_t* out, const Pt& p) {   // (X Z, Y Z^2, Z), this lane's component
        Fp z = coord(p.c, 2);
        Fp zz = F::sqr(z);
        Fp v = fp28::fp_select(q() >= 2, F::mul(p.c, fp28::fp_select(q() == 0, zz, z)), p.c);
        uint32_t w[12], any = 0;
        fp28::fp_to_blst(w, v);
#pragma unroll
        for (int k = 0; k < 12; k++) any |= w[k];
        any |= (uint32_t)__builtin_amdgcn_mov_dpp((int)any, 0xB1, 0xF, 0xF, true);   // either component
        any = (uint32_t)__shfl((int)any, lane_of(2), 64);                              // of Z
        if (q() < 3) {
#pragma unroll
            for (int k = 0; k < 12; k++) out[24 * q() + 12 * h() + k] = any != 0 ? w[k] : 0u;
        }
    }
};
template <class C> struct CoopOf;
template <> struct CoopOf<G1C> { using RS = QuadG1; using CS = QuadG1; };
template <> struct CoopOf<G2C> { using RS = PairG2; using CS = OctG2; };   // throughput-bound reduce, latency-bound combine

// One wave per chunk of K = NLL * L consecutive buckets of one window (NLL = 2^LOG_LL logical lanes; L is ANY value since round 4,
// it was a power of two: 2^16 points at c = 15 take L = 9, 1938 waves of 31 steps, where the power-of-two geometry needed L = 16 and
// 45 steps).  Chunk j of a window covers buckets [j K, j K + K); logical lane l owns [l L, l L + L) of the chunk; buckets beyond
// the window's nb (last chunk of a window, ragged) are infinity.  Output per chunk, in the device bucket layout:
// pairs[2 chunk] = K * S with S = sum B, pairs[2 chunk + 1] = T = sum (rel + 1) B, rel = index inside the chunk.
//
// The whole reduction is ONE loop with ONE inlined addition site: the operands of step s are selected by the (wave-uniform)
// step number (an out-of-line addition passes its operands through scratch; several inlined sites would multiply the code).
//   [0, 2L)            t = L-1 .. 0 :  run += B_t ;  acc += run          (lane-serial running sums)
//   LOG_LL steps       run += shfl_down(run, 1, 2, 4, ..)                 (suffix scan: run_l = sum_{j>=l} S_j)
//   chain steps        LP = L * run by double-and-add over L's bits       (one doubling per bit below the top one, one addition per set bit)
//   one step           acc += (l == 0 ? inf : LP)                         (V_l = T_l + L P_l)
//   LOG_LL steps       acc += shfl_down(acc, NLL/2, .., 1)                (sum over lanes)
// The last logical lane is idle in the final butterfly after its value has been read in the first step: it doubles
// LP_0 = L * S there, LOG_LL times, which gives K * S without a single extra step ("rider").
template <class CS>
__global__ void __launch_bounds__(64, CS::MAX_OCC) k_reduce_coop(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ woff,
                                                                 const uint32_t* __restrict__ offsets, uint32_t* __restrict__ pairs, uint32_t L,
                                                                 uint32_t nb, uint32_t cpw) {
    aux_priority();
    using Pt = typename CS::Pt;
    constexpr int NLL = 1 << CS::LOG_LL, BK = Geo<typename CS::C>::BK_WORDS;
    const uint32_t chunk = blockIdx.x, ll = CS::ll();
    const uint32_t w = chunk / cpw, j = chunk - w * cpw;
    const uint32_t first = (j * NLL + ll) * L;                                   // this logical lane's buckets: first .. first + cnt - 1 of window w
    const uint32_t cnt = first >= nb ? 0u : (nb - first < L ? nb - first : L);
    const uint32_t* wp = woff + (size_t)w * nb + (cnt ? first : 0u);            // bucket b lives at partial[woff[b]]; only rel < cnt is dereferenced as such
    Pt run = CS::inf(), acc = CS::inf(), LP = CS::inf();
    // A chunk without a single entry (offsets[] are the buckets' entry prefix sums) is done: both sums are infinity.  Scalars of a few
    // dozen bits leave most windows empty (64-bit values: 11 of 16 windows at c = 16; witness bits: 15), and their chunks' waves would
    // otherwise walk the whole latency chain next to the few that have work.
    {
        const uint32_t* op = offsets + (size_t)w * nb + (cnt ? first : 0u);
        const uint32_t mine = cnt ? op[cnt] - op[0] : 0u;
        if (__ballot(mine != 0) == 0) {
            if (ll == NLL - 1) CS::store(pairs + (size_t)(2 * chunk) * BK, acc);
            if (ll == 0) CS::store(pairs + (size_t)(2 * chunk + 1) * BK, acc);
            return;
        }
    }
    int top = 0;                                                                 // index of L's top bit
    while ((L >> top) > 1u) top++;
    uint32_t chain = 0;
    for (int b = top - 1; b >= 0; b--) chain += 1 + ((L >> b) & 1u);
    const uint32_t s_scan = 2 * L, s_dbl = s_scan + CS::LOG_LL, s_comb = s_dbl + chain, s_end = s_comb + 1 + CS::LOG_LL;
    const bool rider = ll == NLL - 1;
    int bit = top - 1;                                                           // wave-uniform state of the chain
    bool pend_add = false;
    // software pipeline of the bucket loads: the bucket of pair p + 1 and the index of pair p + 2 are requested while pair p
    // is being added (two dependent loads of ~2 us would otherwise sit in front of every other step of the chain).  A ragged
    // lane loads bucket 0 of its range in place of the missing ones and replaces the value by infinity.
    auto index_of = [&](uint32_t rel) { return wp[rel < cnt ? rel : 0u]; };
    auto bucket_of = [&](uint32_t idx, uint32_t rel) { return CS::select(rel < cnt, CS::inf(), CS::load(partial + (size_t)idx * BK)); };
    // Steps of the running sums whose second operand is infinity in EVERY logical lane of the wave are skipped (the sum is unchanged):
    // "bucket t holds an entry" comes from the entry offsets, "run may be non-zero" follows from it.  Dense chunks never skip; a window
    // with a handful of occupied buckets (witness bits: one) runs the 13 + LOG_LL closing steps and little else instead of all 2 L + ...
    const uint32_t* opw = offsets + (size_t)w * nb + (cnt ? first : 0u);
    auto full_of = [&](uint32_t rel) { return rel < cnt && opw[rel + 1] != opw[rel]; };
    Pt nbk = bucket_of(index_of(L - 1), L - 1);
    bool nbf = full_of(L - 1);
    uint32_t idx2 = L > 1 ? index_of(L - 2) : 0u;
    bool f2 = L > 1 && full_of(L - 2);
    bool fr = false;   // this lane's `run` may be non-zero
#pragma unroll 1
    for (uint32_t s = 0; s < s_end; s++) {
        Pt A, B;
        uint32_t dst;  // 0 run, 1 acc, 2 LP
        if (s < s_scan) {
            bool fB;
            if ((s & 1u) == 0) {
                const uint32_t p = s >> 1;
                A = run; B = nbk; dst = 0; fB = nbf;
                if (p + 1 < L) {
                    nbk = bucket_of(idx2, L - 2 - p);
                    nbf = f2;
                    if (p + 2 < L) { idx2 = index_of(L - 3 - p); f2 = full_of(L - 3 - p); }
                }
                fr = fr || fB;
            } else {
                A = acc; B = run; dst = 1; fB = fr;
            }
            if (__ballot(fB) == 0) continue;
        } else if (s < s_dbl) {
            int d = 1 << (s - s_scan);
            A = run;
            B = CS::select((int)ll + d < NLL, CS::inf(), CS::shfl_down(run, d));
            dst = 0;
        } else if (s < s_comb) {
            if (s == s_dbl) LP = run;
            A = LP;
            if (pend_add) {          // LP = 2 LP + run
                B = run;
                pend_add = false;
            } else {                 // LP = 2 LP, then + run if this bit of L is set
                B = LP;
                pend_add = ((L >> bit) & 1u) != 0;
                bit--;
            }
            dst = 2;
        } else if (s == s_comb) {
            if (chain == 0) LP = run;
            A = acc;
            B = CS::select(ll == 0, LP, CS::inf());
            dst = 1;
        } else {
            int k = s - s_comb - 1, d = (NLL / 2) >> k;
            Pt sh = CS::shfl_down(acc, d);
            A = acc;
            B = CS::select((int)ll < d, CS::inf(), sh);
            if (k == 0) A = CS::select(rider, A, CS::from_ll(LP, 0));   // rider: start from LP_0 = L * S (k is wave-uniform: the
            B = CS::select(rider, B, A);                                 // shuffle runs with every lane active, after acc was read)
            dst = 1;
        }
        CS::add(A, B);
        if (dst == 0) run = A;
        else if (dst == 1) acc = A;
        else LP = A;
    }
    if (rider) CS::store(pairs + (size_t)(2 * chunk) * BK, acc);
    if (ll == 0) CS::store(pairs + (size_t)(2 * chunk + 1) * BK, acc);
}

// Throughput form of the bucket reduction for millions of buckets (>= 2^21: 2^24 points at c = 20 have 6.8 M): ONE lane per L
// consecutive buckets of a window, single-lane complete additions (5200 instructions per addition instead of 4 x 1864 on a quad),
// two waves per SIMD.  Only the running sum lives in registers across a step: the weighted sum `acc` is parked in LDS (168 B per
// lane) while run += B is computed and fetched for acc += run, so the one inlined addition site fits 256 VGPRs without
// spilling (the round-1 kernel kept run, acc and a third point live: 512 VGPRs, one wave per SIMD, half the issue rate).
// L is ANY value <= 64 (round 3; it was fixed at 64): the plan picks the L whose lanes fill one round of wave slots — 53 at 2^24
// points, 2010 waves of 114 steps instead of 1664 waves of 134.  Lane g = (window w, j) covers buckets [j L, min(j L + L, nb)) of
// window w (the last lane of a window is ragged: its missing TOP buckets are infinity) and leaves the pair (L S, T) in the layout
// k_reduce_coop uses — k_combine takes it from there; its element size K may be any number, it only has to be the same for all
// pairs of a level.  L S comes from S by the double-and-add chain of L's bits (S parked in LDS once T has been written out).
template <class C>
__global__ void __launch_bounds__(64, 2) k_reduce_serial(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ woff,
                                                        const uint32_t* __restrict__ offsets, uint32_t nlanes, uint32_t L, uint32_t nb,
                                                        uint32_t lpw, uint32_t* __restrict__ pairs) {
    aux_priority();
    using F = typename C::F;
    using FR = typename C::FR;
    using E = typename F::E;
    using PJ = ec::Proj<F>;
    constexpr int BK = Geo<C>::BK_WORDS, SLOT = Geo<C>::SLOT;
    __shared__ uint32_t park[64 * BK];   // one point per lane, slot-interleaved: [coordinate][lane][SLOT words]; acc, later S
    const uint32_t lane = threadIdx.x;
    uint32_t g = blockIdx.x * 64 + lane;
    const bool live = g < nlanes;
    if (!live) g = nlanes - 1;             // idle lanes of the last wave repeat the last lane's work and store nothing
    const uint32_t w = g / lpw, j = g - w * lpw;
    const uint32_t first = j * L, cnt = nb - first < L ? nb - first : L;   // this lane's buckets: first .. first + cnt - 1 of window w
    const uint32_t* wp = woff + (size_t)w * nb + first;
    {   // a wave whose lanes' buckets hold no entry at all (empty windows of short scalars) writes infinity pairs and leaves (cf. k_reduce_coop)
        const uint32_t* op = offsets + (size_t)w * nb + first;
        if (__ballot(op[cnt] != op[0]) == 0) {
            if (live) {
                store_bucket<C>(pairs + (size_t)(2 * g) * BK, ec::proj_inf<F>());
                store_bucket<C>(pairs + (size_t)(2 * g + 1) * BK, ec::proj_inf<F>());
            }
            return;
        }
    }
    auto park_store = [&](const PJ& p) {
        ElemIO<E>::store(park + (0 * 64 + lane) * SLOT, p.x);
        ElemIO<E>::store(park + (1 * 64 + lane) * SLOT, p.y);
        ElemIO<E>::store(park + (2 * 64 + lane) * SLOT, p.z);
    };
    auto park_load = [&]() {
        PJ p;
        ElemIO<E>::load(p.x, park + (0 * 64 + lane) * SLOT);
        ElemIO<E>::load(p.y, park + (1 * 64 + lane) * SLOT);
        ElemIO<E>::load(p.z, park + (2 * 64 + lane) * SLOT);
        return p;
    };
    int top = 0;                           // index of L's top bit
    while ((L >> top) > 1u) top++;
    uint32_t chain = 0;                    // steps of the double-and-add chain: one doubling per bit below the top one, one addition more per set bit
    for (int b = top - 1; b >= 0; b--) chain += 1 + ((L >> b) & 1u);
    const uint32_t nsteps = 2 * L + chain;
    PJ run = ec::proj_inf<F>();
    park_store(ec::proj_inf<F>());
    uint32_t idx = wp[L - 1 < cnt ? L - 1 : 0];   // bucket of step pair t = 0 is rel = L - 1 (beyond a ragged lane's buckets: infinity)
    int bit = top - 1;                     // wave-uniform state of the chain
    bool pend_add = false;
    // as in k_reduce_coop: a running-sum step whose second operand is infinity in every lane of the wave is skipped ("bucket holds an
    // entry" from the entry offsets; `fr`: this lane's run may be non-zero)
    const uint32_t* opw = offsets + (size_t)w * nb + first;
    bool fr = false;
#pragma unroll 1
    for (uint32_t s = 0; s < nsteps; s++) {
        PJ A, B;
        const bool phase1 = s < 2 * L, even = (s & 1u) == 0;
        if (phase1 && even) {              // run += B_rel, rel = L - 1 - t
            const uint32_t t = s >> 1, rel = L - 1 - t;
            const bool full = rel < cnt && opw[rel + 1] != opw[rel];
            const uint32_t idx_now = idx;
            if (t + 1 < L) idx = wp[rel - 1 < cnt ? rel - 1 : 0];
            fr = fr || full;
            if (__ballot(full) == 0) continue;
            A = run;
            B = load_bucket<C>(partial + (size_t)idx_now * BK);
            // beyond a ragged lane's buckets: infinity = (0 : Y : 0) for ANY Y != 0 — clearing X and Z of whatever was loaded costs
            // two selects against the inline constant 0 (a select against (0 : 1 : 0) parked the 14 limbs of one in registers)
            B.x = F::select(rel < cnt, F::zero(), B.x);
            B.z = F::select(rel < cnt, F::zero(), B.z);
        } else if (phase1) {               // acc += run
            if (__ballot(fr) == 0) continue;
            A = park_load();
            B = run;
        } else {
            if (s == 2 * L) {              // T is complete: out it goes, the LDS slot now keeps S for the chain's additions
                if (live) store_bucket<C>(pairs + (size_t)(2 * g + 1) * BK, park_load());
                park_store(run);
            }
            A = run;
            if (pend_add) {                // run = 2 run + S
                B = park_load();
                pend_add = false;
            } else {                       // run = 2 run, then + S if this bit of L is set
                B = run;
                pend_add = ((L >> bit) & 1u) != 0;
                bit--;
            }
        }
        ec::proj_add<FR>(A, B);   // Proj<F> and Proj<FR> are one type (ec.cuh): FR only selects the multiplier form
        if (phase1 && !even) park_store(A);
        else run = A;
    }
    if (nsteps == 2 * L && live) store_bucket<C>(pairs + (size_t)(2 * g + 1) * BK, park_load());   // L = 1: no chain
    if (live) store_bucket<C>(pairs + (size_t)(2 * g) * BK, run);
}

// One level of the per-window combine: a wave takes NLL consecutive pairs (S'_j, T_j) of one window (S'_j = K S_j already
// scaled by the element size K of this level) and leaves ONE pair for the next level:
//   T_out = sum_j T_j + sum_{j >= 1} P'_j,  P'_j = sum_{i >= j} S'_i  (= sum_j (T_j + j K S_j)),   S'_out = NLL * P'_0.
//   LOG_LL steps  run += shfl_down(run, 1, 2, ..)     suffix scan of S'
//   one step      acc = T + (l == 0 ? inf : run)
//   LOG_LL steps  butterfly over the lanes; the last logical lane doubles P'_0 meanwhile (rider, as in k_reduce_coop)
// cpw_in pairs per window come in, cpw_out = ceil(cpw_in / NLL) go out.  With jac_out != nullptr (last level, cpw_out == 1) the
// window sum is written as a Jacobian point in the reference's form instead: the host does only the Horner fold over the
// windows (/root/reference/src/gpu.rs:193-209 does the whole tail on the host).  grid = nwin * cpw_out waves.
template <class CS>
__global__ void __launch_bounds__(64, CS::MAX_OCC) k_combine(const uint32_t* __restrict__ pairs_in, uint32_t cpw_in, uint32_t cpw_out,
                                                             uint32_t* __restrict__ pairs_out, uint32_t* __restrict__ jac_out,
                                                             const uint32_t* __restrict__ meta) {
    aux_priority();
    using Pt = typename CS::Pt;
    constexpr int NLL = 1 << CS::LOG_LL, BK = Geo<typename CS::C>::BK_WORDS, RJ = Geo<typename CS::C>::RAW_JAC;
    const uint32_t w = blockIdx.x / cpw_out, g = blockIdx.x % cpw_out, ll = CS::ll();
    const uint32_t j = g * NLL + ll;
    const bool have = j < cpw_in;
    const uint32_t* src = pairs_in + ((size_t)w * cpw_in + (have ? j : 0)) * 2 * BK;
    Pt run = CS::select(have, CS::inf(), CS::load(src));
    Pt acc = CS::select(have, CS::inf(), CS::load(src + BK));
    const bool rider = ll == NLL - 1;
    if (cpw_in > 1) {
        const uint32_t s_comb = CS::LOG_LL, s_end = s_comb + 1 + CS::LOG_LL;
#pragma unroll 1
        for (uint32_t s = 0; s < s_end; s++) {
            Pt A, B;
            uint32_t dst;  // 0 run, 1 acc
            if (s < s_comb) {
                int d = 1 << s;
                A = run;
                B = CS::select((int)ll + d < NLL, CS::inf(), CS::shfl_down(run, d));
                dst = 0;
            } else if (s == s_comb) {
                A = acc;
                B = CS::select(ll == 0, run, CS::inf());
                dst = 1;
            } else {
                int k = s - s_comb - 1, d = (NLL / 2) >> k;
                Pt sh = CS::shfl_down(acc, d);
                A = acc;
                B = CS::select((int)ll < d, CS::inf(), sh);
                if (k == 0) A = CS::select(rider, A, CS::from_ll(run, 0));
                B = CS::select(rider, B, A);
                dst = 1;
            }
            CS::add(A, B);
            if (dst == 0) run = A;
            else acc = A;
        }
    }
    if (jac_out) {
        if (ll == 0) CS::store_jac_raw(jac_out + (size_t)w * RJ, acc);
        // last level: the accumulate kernel's clock sums travel to the host behind the window sums (4 words after the gridDim.x Jacobian points)
        if (blockIdx.x == 0 && threadIdx.x < 4) jac_out[(size_t)gridDim.x * RJ + threadIdx.x] = meta[CLK_META + threadIdx.x];
    } else {
        uint32_t* o = pairs_out + ((size_t)w * cpw_out + g) * 2 * BK;
        if (rider) CS::store(o, acc);
        if (ll == 0) CS::store(o + BK, acc);
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
constexpr uint32_t NORM_K = 8;   // round 6: 32 -> 8 (one lane per group: 2^20 values at fan-out 32 are 512 waves, a quarter of the wave slots; at 8 they fill them: 1.17 -> 0.96 ms, G2 3.09 -> 2.20; 4 is slower again — seven levels)

template <class C>
__global__ void __launch_bounds__(256, 2) k_norm_load(const uint32_t* __restrict__ raw_jac, uint32_t n, uint32_t* __restrict__ vals) {
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
__global__ void __launch_bounds__(256, 2) k_norm_up(const uint32_t* __restrict__ vals, uint32_t m, uint32_t* __restrict__ pref,
                                                 uint32_t* __restrict__ tot) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo = g * NORM_K, hi = lo + NORM_K < m ? lo + NORM_K : m;
    if (lo >= m) return;
    E run = F::one(), nv;
    ElemIO<E>::load(nv, vals + (size_t)lo * Geo<C>::SLOT);   // the next value is in flight while the current product runs (a lane walks its group alone)
    for (uint32_t k = lo; k < hi; k++) {
        const E v = nv;
        if (k + 1 < hi) ElemIO<E>::load(nv, vals + (size_t)(k + 1) * Geo<C>::SLOT);
        ElemIO<E>::store(pref + (size_t)k * Geo<C>::SLOT, run);
        run = F::mul(run, v);
    }
    ElemIO<E>::store(tot + (size_t)g * Geo<C>::SLOT, run);
}

template <class C>
__global__ void __launch_bounds__(256, 2) k_norm_down(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pref,
                                                   const uint32_t* __restrict__ inv_tot, uint32_t m, uint32_t* __restrict__ inv_vals) {
    using F = typename C::F;
    using E = typename F::E;
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo = g * NORM_K, hi = lo + NORM_K < m ? lo + NORM_K : m;
    if (lo >= m) return;
    E I, np_, nv;
    ElemIO<E>::load(I, inv_tot + (size_t)g * Geo<C>::SLOT);
    ElemIO<E>::load(np_, pref + (size_t)(hi - 1) * Geo<C>::SLOT);   // as in k_norm_up: the next step's operands are in flight during this one's products
    ElemIO<E>::load(nv, vals + (size_t)(hi - 1) * Geo<C>::SLOT);
    for (uint32_t k = hi; k-- > lo;) {
        const E p = np_, v = nv;
        if (k > lo) {
            ElemIO<E>::load(np_, pref + (size_t)(k - 1) * Geo<C>::SLOT);
            ElemIO<E>::load(nv, vals + (size_t)(k - 1) * Geo<C>::SLOT);
        }
        ElemIO<E>::store(inv_vals + (size_t)k * Geo<C>::SLOT, F::mul(I, p));
        I = F::mul(I, v);
    }
}

template <class C>
__global__ void __launch_bounds__(256, 2) k_norm_final(const uint32_t* __restrict__ raw_jac, const uint32_t* __restrict__ zinv, uint32_t n,
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
    ElemIO<E>::unpack_raw(x, q);                // as in k_norm_down0_final: no change of radix for X and Y
    ElemIO<E>::unpack_raw(y, q + R);
    ElemIO<E>::load(zi, zinv + (size_t)i * Geo<C>::SLOT);
    E zi2 = F::mul(zi, zi);
    E zi3 = F::mul(zi2, zi);
    uint32_t* o = raw_aff + (size_t)i * Geo<C>::RAW_AFF;
    ElemIO<E>::pack_raw(o, F::mul(x, zi2), any != 0);
    ElemIO<E>::pack_raw(o + R, F::mul(y, zi3), any != 0);
}

// Level 0 of the tree FUSED with what surrounds it (round 6): the up-sweep reads Z straight from the caller's Jacobian points, the down-sweep
// produces the affine point the moment a Z inverse exists.  Level 0 then keeps only its prefix products: the separate form wrote and re-read
// the converted Z values and their inverses (2 x 64 B per point each way) and took two launches more per chunk — 1.65 GB of HBM traffic per
// 2^20-point call against 252 MB algorithmic.  Used when the tree has more than one level (n > 64).
//   k_norm_up0         : per group of K points: Z_i (raw, infinity -> 1) -> exclusive prefix products + group total (= level 1's value)
//   k_norm_down0_final : per group, last point first: 1 / Z_i = I * prefix_i, I *= Z_i (re-read, re-converted: one multiplication instead of
//                        128 B of traffic); x = X / Z^2, y = Y / Z^3 in the reference's form
template <class C>
__global__ void __launch_bounds__(256, 2) k_norm_up0(const uint32_t* __restrict__ raw_jac, uint32_t n, uint32_t* __restrict__ pref,
                                                  uint32_t* __restrict__ tot) {
    using F = typename C::F;
    using E = typename F::E;
    constexpr int R = ElemIO<E>::RAW;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lo = g * NORM_K, hi = lo + NORM_K < n ? lo + NORM_K : n;
    if (lo >= n) return;
    // the next point's Z words are fetched while the current product runs (one lane walks its group: without the prefetch every step
    // waits for its own gather — 104 -> ~60 us at 2^20 points)
    uint32_t zw[R], nw[R];
    {
        const uint32_t* zr = raw_jac + (size_t)lo * Geo<C>::RAW_JAC + 2 * R;
#pragma unroll
        for (int w = 0; w < R; w++) nw[w] = zr[w];
    }
    E run = F::one();
    for (uint32_t k = lo; k < hi; k++) {
#pragma unroll
        for (int w = 0; w < R; w++) zw[w] = nw[w];
        if (k + 1 < hi) {
            const uint32_t* zr = raw_jac + (size_t)(k + 1) * Geo<C>::RAW_JAC + 2 * R;
#pragma unroll
            for (int w = 0; w < R; w++) nw[w] = zr[w];
        }
        ElemIO<E>::store(pref + (size_t)k * Geo<C>::SLOT, run);
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < R; w++) any |= zw[w];
        E z;
        ElemIO<E>::from_raw(z, zw);
        run = F::mul(run, F::select(any == 0, z, F::one()));
    }
    ElemIO<E>::store(tot + (size_t)g * Geo<C>::SLOT, run);
}

template <class C>
__global__ void __launch_bounds__(256, 2) k_norm_down0_final(const uint32_t* __restrict__ raw_jac, const uint32_t* __restrict__ pref,
                                                          const uint32_t* __restrict__ inv_tot, uint32_t n, uint32_t* __restrict__ raw_aff) {
    using F = typename C::F;
    using E = typename F::E;
    constexpr int R = ElemIO<E>::RAW;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lo = g * NORM_K, hi = lo + NORM_K < n ? lo + NORM_K : n;
    if (lo >= n) return;
    E I;
    ElemIO<E>::load(I, inv_tot + (size_t)g * Geo<C>::SLOT);
    // as in k_norm_up0: the next point (3 R words) and its prefix product are in flight while the current point's seven products run.  For Fp
    // only: over Fp2 the two staged points are 144 registers and the loop spills 152 (the compiler hoists the plain loads as far as it can anyway).
    constexpr bool PF = R == 12;
    uint32_t cw[PF ? 3 * R : 1], nw[PF ? 3 * R : 1];
    E p, np_;
    if constexpr (PF) {
        const uint32_t* q = raw_jac + (size_t)(hi - 1) * Geo<C>::RAW_JAC;
#pragma unroll
        for (int w = 0; w < 3 * R; w++) nw[w] = q[w];
        ElemIO<E>::load(np_, pref + (size_t)(hi - 1) * Geo<C>::SLOT);
    }
    for (uint32_t k = hi; k-- > lo;) {
        if constexpr (PF) {
#pragma unroll
            for (int w = 0; w < 3 * R; w++) cw[w] = nw[w];
            p = np_;
            if (k > lo) {
                const uint32_t* q = raw_jac + (size_t)(k - 1) * Geo<C>::RAW_JAC;
#pragma unroll
                for (int w = 0; w < 3 * R; w++) nw[w] = q[w];
                ElemIO<E>::load(np_, pref + (size_t)(k - 1) * Geo<C>::SLOT);
            }
        } else {
            ElemIO<E>::load(p, pref + (size_t)k * Geo<C>::SLOT);
        }
        const uint32_t* src = PF ? cw : raw_jac + (size_t)k * Geo<C>::RAW_JAC;
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < R; w++) any |= src[2 * R + w];
        E x, y, z;
        ElemIO<E>::from_raw(z, src + 2 * R);
        z = F::select(any == 0, z, F::one());
        const E zi = F::mul(I, p);
        I = F::mul(I, z);
        ElemIO<E>::unpack_raw(x, src);          // X, Y stay in the caller's Montgomery form: times an internal-form factor they come out in it
        ElemIO<E>::unpack_raw(y, src + R);
        const E zi2 = F::mul(zi, zi);
        const E zi3 = F::mul(zi2, zi);
        uint32_t* o = raw_aff + (size_t)k * Geo<C>::RAW_AFF;
        ElemIO<E>::pack_raw(o, F::mul(x, zi2), any != 0);
        ElemIO<E>::pack_raw(o + R, F::mul(y, zi3), any != 0);
    }
}

// device form <-> reference form for a short vector of field elements (top of the product tree, host inversion)
template <class C>
__global__ void __launch_bounds__(64, 2) k_elems_to_raw(const uint32_t* __restrict__ dev, uint32_t m, uint32_t* __restrict__ raw) {
    using E = typename C::F::E;
    uint32_t i = threadIdx.x;
    if (i >= m) return;
    E v;
    ElemIO<E>::load(v, dev + (size_t)i * Geo<C>::SLOT);
    ElemIO<E>::to_raw(raw + (size_t)i * ElemIO<E>::RAW, v, true);
}
template <class C>
__global__ void __launch_bounds__(64, 2) k_elems_from_raw(const uint32_t* __restrict__ raw, uint32_t m, uint32_t* __restrict__ dev) {
    using E = typename C::F::E;
    uint32_t i = threadIdx.x;
    if (i >= m) return;
    E v;
    ElemIO<E>::from_raw(v, raw + (size_t)i * ElemIO<E>::RAW);
    ElemIO<E>::store(dev + (size_t)i * Geo<C>::SLOT, v);
}

}  // namespace msmk
