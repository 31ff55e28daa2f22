// Batched Miller loop + final exponentiation (BASELINE config #5): host driver of pairing_kernels.cuh.
// Replaces <Bls12 as Pairing>::multi_miller_loop / final_exponentiation (/root/reference/src/pairing.rs:49-80).
#include <cmath>
#include "internal.hpp"
#include "pairing_kernels.cuh"

namespace mi {
namespace {

// Host instance of the generic tower for the O(1) tail (products of the last few tree values, final exponentiation):
// the same pairing.cuh code over hostec's 64-bit-limb, fully reduced Fp2 (every bound hook is the identity), ~4x faster on
// a CPU core than the 28-bit representation the GPU uses.  blst_fp12 is exactly its memory layout.
struct HostF2 {
    using E = hostec::Fp2;
    using Fp = hostec::Fp;
    static E zero() { return E::zero(); }
    static E one() { return E::one(); }
    static E mul(const E& a, const E& b) { return a * b; }
    static E sqr(const E& a) { return a.sqr(); }
    static E mul2add(const E& a, const E& b, const E& c, const E& d) { return a * b + c * d; }
    static E add(const E& a, const E& b) { return a + b; }
    template <int K> static E sub(const E& a, const E& b) { return a - b; }
    template <int K> static E neg(const E& a) { return E::zero() - a; }
    template <int K> static E mul_xi(const E& a) { return E{a.c0 - a.c1, a.c0 + a.c1}; }
    static E norm2(const E& a) { return a; }
    static E dbl(const E& a) { return a + a; }
    static Fp fp_neg4(const Fp& a) { return Fp::zero() - a; }
    static E inv(const E& a) { return a.inv(); }
    static E frob_const(int i) {   // g^i from the generated table (internal 2^392 form) -> blst form, converted once
        static const std::array<E, 5> tab = [] {
            std::array<E, 5> t;
            for (int k = 0; k < 5; k++) {
                ec::Fp2 c = pairing::PF2::frob_const(k + 1);
                uint32_t w[12];
                fp28::fp_to_blst(w, c.c0);
                memcpy(t[k].c0.l, w, 48);
                fp28::fp_to_blst(w, c.c1);
                memcpy(t[k].c1.l, w, 48);
            }
            return t;
        }();
        return tab[i - 1];
    }
};
using HT = pairing::Tower<HostF2>;
static_assert(sizeof(HT::E12) == sizeof(mi_fp12), "host Fp12 must be the reference's blst_fp12 layout");

HT::E12 fp12_from_raw(const mi_fp12* f) {
    HT::E12 r;
    memcpy(&r, f, sizeof r);
    return r;
}
void fp12_to_raw(mi_fp12* out, const HT::E12& a) { memcpy(out, &a, sizeof a); }


// Miller loops of one shard on one device, multiplied down to <= 4 values on the GPU and to one on the host
HT::E12 device_miller(mi_ctx* ctx, DevState& d, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_pairing_profile& pp) {
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    DevBuf &dp = d.pr_p, &dq = d.pr_q, &raw = d.pr_raw, &dlines = d.pr_lines;   // kept across calls (no per-call hipMalloc)
    DevBuf* lvl = d.pr_lvl;
    HT::E12 acc = HT::one12();
    dp.ensure(n * sizeof(mi_g1_affine));
    dq.ensure(n * sizeof(mi_g2_affine));
    size_t fp12_bytes = (size_t)msmk::FP12_WORDS * 4;
    lvl[0].ensure(n * fp12_bytes);
    lvl[1].ensure(((n + msmk::FP12_TREE_K - 1) / msmk::FP12_TREE_K) * fp12_bytes);
    raw.ensure(64 * sizeof(mi_fp12));
    d.ensure_host(64 * sizeof(mi_fp12));
    HIP_TRY(hipEventRecord(d.ev[0], s));
    HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(mi_g1_affine), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dq.p, q, n * sizeof(mi_g2_affine), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(d.ev[1], s));
    bool single_lane = false;
    size_t batch_cap = (size_t)1 << 17;
    // Pairs per accumulator m: f <- f^2 l_1 ... l_m per step, so a larger m saves squarings (one squaring costs about two line
    // multiplications), but the accumulate kernel is a grid of lone waves (ten accumulators each, 254 VGPRs): its time is the time
    // of ONE wave as long as every wave has a SIMD to itself, and 1.84x that once a second wave shares the SIMD.  m is the minimum of
    //   (1 + 0.53 m) * g(ceil(waves(m) / SIMD slots)),  g(1) = 1, g(2) = 1.84, g(r) = 0.92 r,  m <= 8
    // Measured (tools/scan_pairing_share.py, accumulate kernel, ms): 2^16 pairs m = 4 / 6 / 7 / 8 / 16 -> 5.7 / 7.5 / 4.75 / 5.3 / 9.2
    // (937 waves at 7, 1093 at 6); 2^17: 7 / 8 / 13 / 16 -> 8.8 / 9.6 / 14.0 / 16.9; 2^15: 2 / 3 / 4 / 5 -> 3.6 / 4.4 / 3.0 / 3.6;
    // 2^14: 1 / 2 / 4 -> 2.4 / 1.8 / 2.8.  1009 waves (m = 13 at 2^17) already behave as doubled up: slots = 15/16 of the SIMDs.
    // A twelve-lane variant of the kernel (one Fp component per lane) was slower at every setting.
    uint32_t share = 0;
#if defined(MI_TEST_HOOKS)
    single_lane = ctx->test_pairing_single_lane;
    if (ctx->test_pairing_batch) batch_cap = ctx->test_pairing_batch;
    if (ctx->test_pairing_share) share = ctx->test_pairing_share;
#else
    (void)ctx;
#endif
    if (!share) {
        const size_t batch_pairs = std::min<size_t>(n, batch_cap);
        const double slots = d.simds * (15.0 / 16.0);
        double best = 0;
        for (uint32_t m = 1; m <= 8; m++) {
            const double waves = std::ceil(std::ceil((double)batch_pairs / m) / msmk::MILLER_GROUPS);
            const double r = std::max(1.0, std::ceil(waves / slots));
            const double cost = (1.0 + 0.53 * m) * r;   // the kernel is built for one wave per SIMD: a second round of waves waits for the first
            if (!share || cost < best) { best = cost; share = m; }
        }
    }
    size_t nvals = 0;   // Fp12 values the Miller kernels leave in lvl[0]
    bool split_timed = false;
    if (single_lane) {
#if defined(MI_TEST_HOOKS)
        hipLaunchKernelGGL(msmk::k_miller_loop, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t*)dp.p, (const uint32_t*)dq.p,
                           (uint32_t)n, (uint32_t*)lvl[0].p);
#endif
        nvals = n;
    } else {
        // line coefficients of a batch of pairs (26 KB per pair), then six lanes per accumulator fold them into f
        const size_t batch = std::min<size_t>(n, batch_cap);
        const uint32_t blk = share * msmk::MILLER_GROUPS;   // pairs per accumulate wave = one block of the line buffer
        dlines.ensure((batch + blk - 1) / blk * blk * msmk::MILLER_LINES * 3 * 32 * 4);
        for (size_t lo = 0; lo < n; lo += batch) {
            uint32_t mm = (uint32_t)std::min(batch, n - lo);
            uint32_t groups = (mm + share - 1) / share;
            hipLaunchKernelGGL(msmk::k_miller_lines2, dim3((2 * mm + 63) / 64), dim3(64), 0, s,
                               (const uint32_t*)dp.p + lo * msmk::Geo<msmk::G1C>::RAW_AFF, (const uint32_t*)dq.p + lo * msmk::Geo<msmk::G2C>::RAW_AFF,
                               mm, blk, (uint32_t*)dlines.p);
            if (lo == 0) HIP_TRY(hipEventRecord(d.ev[4], s));   // first batch: the two kernels timed separately (profile)
            hipLaunchKernelGGL(msmk::k_miller_accumulate, dim3((groups + msmk::MILLER_GROUPS - 1) / msmk::MILLER_GROUPS), dim3(64), 0, s,
                               (const uint32_t*)dlines.p, mm, share, blk, (uint32_t*)lvl[0].p + nvals * msmk::FP12_WORDS);
            if (lo == 0) HIP_TRY(hipEventRecord(d.ev[5], s));
            nvals += groups;
        }
        split_timed = true;
    }
    HIP_TRY(hipEventRecord(d.ev[2], s));
    size_t m = nvals;
    int cur = 0;
    while (m > 4) {   // a host Fp12 product costs ~13 us, a tree level ~50 us
        size_t g = (m + msmk::FP12_TREE_K - 1) / msmk::FP12_TREE_K;
        hipLaunchKernelGGL(msmk::k_fp12_prod, dim3((uint32_t)((g + msmk::MILLER_GROUPS - 1) / msmk::MILLER_GROUPS)), dim3(64), 0, s, (const uint32_t*)lvl[cur].p, (uint32_t)m,
                           (uint32_t*)lvl[cur ^ 1].p);
        cur ^= 1;
        m = g;
    }
    hipLaunchKernelGGL(msmk::k_fp12_to_raw, dim3(1), dim3(64), 0, s, (const uint32_t*)lvl[cur].p, (uint32_t)m, (uint32_t*)raw.p);
    HIP_TRY(hipEventRecord(d.ev[3], s));
    HIP_TRY(hipMemcpyAsync(d.h_pairs, raw.p, m * sizeof(mi_fp12), hipMemcpyDeviceToHost, s));   // pinned staging
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    const mi_fp12* top = static_cast<const mi_fp12*>(d.h_pairs);
    for (size_t k = 0; k < m; k++) acc = HT::mul12(acc, fp12_from_raw(&top[k]));
    pp = mi_pairing_profile{};
    pp.n = n;
    pp.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
    pp.miller_ms = ev_ms(d.ev[1], d.ev[2]);       // Miller loops (line kernel + accumulate kernel, all batches)
    pp.tree_ms = ev_ms(d.ev[2], d.ev[3]);         // multiplication tree
    if (split_timed) {                            // first line batch (the whole call up to 2^17 pairs): the two kernels apart
        pp.lines_ms = ev_ms(d.ev[1], d.ev[4]);
        pp.accumulate_ms = ev_ms(d.ev[4], d.ev[5]);
        pp.pairs_per_accumulator = share;
    }
    return acc;
}

}  // namespace

int miller(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out, bool final_exp) {
    if (!ctx || !out || (n && (!p || !q))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        size_t g = ctx->devs.size();
        if (n < 2 * g) g = 1;   // a handful of pairs: one device
        std::vector<HT::E12> part(g, HT::one12());
        std::vector<PartErr> errs(g);
        std::vector<mi_pairing_profile> pps(g);
        auto t0 = std::chrono::steady_clock::now();
        auto work = [&](size_t k) {
            if (k >= g) return;
            guarded_part(errs[k], [&] {
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                if (hi > lo) part[k] = device_miller(ctx, ctx->devs[k], p + lo, q + lo, hi - lo, pps[k]);
            });
        };
        if (g == 1) work(0);
        else for_each_device(lk, ctx->devs.size(), work);
        for (size_t k = 0; k < g; k++)
            if (errs[k].code != MI_OK) return fail(ctx, errs[k].code, errs[k].msg);
        HT::E12 f = part[0];
        for (size_t k = 1; k < g; k++) f = HT::mul12(f, part[k]);
        auto t1 = std::chrono::steady_clock::now();
        if (final_exp) f = HT::final_exp(f);
        fp12_to_raw(out, f);
        mi_pairing_profile pr = pps[0];
        pr.n = n;
        pr.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk2(ctx->info_mu);
            ctx->pprof = pr;
        }
        return MI_OK;
    });
}

int final_exponentiation(const mi_fp12* f, mi_fp12* out) {
    if (!f || !out) return MI_E_INVALID;
    try {
        fp12_to_raw(out, HT::final_exp(fp12_from_raw(f)));
    } catch (...) {
        return MI_E_HIP;
    }
    return MI_OK;
}

}  // namespace mi
