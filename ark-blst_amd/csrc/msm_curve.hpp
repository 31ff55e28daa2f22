// The MSM pipeline of one curve on the host side: templates over the curve descriptor (msmk::G1C / msmk::G2C), instantiated
// in msm_g1.hip and msm_g2.hip.  Replaces crate::gpu::msm + SingleMultiexpKernel::multiexp of the reference
// (/root/reference/src/gpu.rs:126-241) behind <G{1,2}Projective as VariableBaseMSM>::msm (src/g1.rs:602-632, src/g2.rs:582-612).
#pragma once
#include <exception>
#include "internal.hpp"
#include "io_chunks.hpp"
#include "curve_kernels.cuh"

namespace mi {

// host-side view of a curve: the reference's raw sizes and the CPU Jacobian type used for the O(windows) tail
template <class C> struct HostCurve;
template <> struct HostCurve<msmk::G1C> {
    using J = hostec::G1;
    static constexpr int IDX = 0;
    // accumulate 7.1e9 additions/s, 11 us per addition and lane; quad-lane complete addition ~6 us per step, two waves per SIMD
    static CurveCost cost() { return CurveCost{msmk::QuadG1::LOG_LL, msmk::QuadG1::LOG_LL, 2048, 7400.0, 14.0, 0.68, 64.0, 9.5, 5.9, 4.4, 40.0, 1ull << 21, 25.0}; }
};
template <> struct HostCurve<msmk::G2C> {
    using J = hostec::G2;
    static constexpr int IDX = 1;
    // lane pairs for the reduce (throughput), eight lanes per logical lane for the combine (latency)
    static CurveCost cost() { return CurveCost{msmk::PairG2::LOG_LL, msmk::OctG2::LOG_LL, 1024, 2300.0, 36.0, 0.45, 32.0, 28.0, 24.5, 14.0, 80.0, 0, 0.0}; }
};
template <class C> constexpr size_t aff_bytes() { return (size_t)msmk::Geo<C>::RAW_AFF * 4; }
template <class C> constexpr size_t jac_bytes() { return (size_t)msmk::Geo<C>::RAW_JAC * 4; }

// One call handles at most this many points per device in one pass of the pipeline (32-bit entry offsets: n * windows < 2^32);
// longer inputs are cut into parts whose sums are added (cf. the unfinished calc_chunk_size path, /root/reference/src/gpu.rs:64-85,238-239).
constexpr size_t MAX_PART_POINTS = (size_t)1 << 26;
inline size_t max_part(mi_ctx* ctx) {
#if defined(MI_TEST_HOOKS)
    if (ctx->test_max_part) return ctx->test_max_part;
#endif
    (void)ctx;
    return MAX_PART_POINTS;
}

// Does device slot k of the context read its shard of a scalar vector that lives on device `owner` IN PLACE?  Only when it is the
// same device.  A remote shard is copied once with hipMemcpyPeerAsync (a DMA over xGMI where mi_msm_init could enable peer access,
// staged through the host by the runtime where it could not) instead of being read twice by the sort's count and scatter passes
// through a peer mapping: half the link traffic, and no kernel ever dereferences a pointer its device may not be able to map.
inline bool can_read(const mi_ctx* ctx, size_t k, int owner) {
#if defined(MI_TEST_HOOKS)
    if (ctx->test_no_peer) return false;
#endif
    return ctx->devs[k].dev == owner;
}

// what mi_msm_g{1,2}_device_windows leaves behind instead of a folded result
struct WinOut {
    uint32_t* d_out = nullptr;   // caller's device buffer: num_windows Jacobian points in the reference's form
    mi_window_info info{};
};

// bases raw (host or device) -> device form in `dst`
template <class C>
void ingest(DevState& d, const void* bases, bool bases_on_device, size_t n, uint32_t* dst, uint8_t* flags) {
    const void* src = bases;
    if (!bases_on_device) {
        d.raw.ensure(n * aff_bytes<C>());
        HIP_TRY(hipMemcpyAsync(d.raw.p, bases, n * aff_bytes<C>(), hipMemcpyHostToDevice, d.stream));
        src = d.raw.p;
    }
    uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(msmk::k_ingest<C>, dim3(grid), dim3(256), 0, d.stream, (const uint32_t*)src, dst, flags, (uint32_t)n);
    HIP_TRY(hipGetLastError());
}

// ---- batch inversion of n device-form field elements (Montgomery's trick as a product tree of fan-out NORM_K; the <= 64 values
// at the top are inverted on the host with one Fermat inversion: a single GPU lane would need ~1 ms for it)
struct InvTree {
    std::vector<size_t> sz, off;
    size_t total = 0;
    explicit InvTree(size_t n) {
        sz.push_back(n);
        while (sz.back() > 64) sz.push_back((sz.back() + msmk::NORM_K - 1) / msmk::NORM_K);
        off.resize(sz.size());
        for (size_t l = 0; l < sz.size(); total += sz[l], l++) off[l] = total;
    }
};
// vals level 0 must be filled; on return inv level 0 holds the inverses.  Synchronises the stream once (host inversion).
template <class C>
void invert_tree(DevState& d, const InvTree& t, DevBuf& vals, DevBuf& pref, DevBuf& inv, DevBuf& top_raw) {
    using J = typename HostCurve<C>::J;
    using FE = decltype(J::inf().x);
    constexpr size_t SLOTB = (size_t)msmk::Geo<C>::SLOT * 4;
    hipStream_t s = d.stream;
    auto at = [&](DevBuf& b, size_t l) { return (uint32_t*)((char*)b.p + t.off[l] * SLOTB); };
    for (size_t l = 0; l + 1 < t.sz.size(); l++) {
        uint32_t groups = (uint32_t)t.sz[l + 1];
        hipLaunchKernelGGL(msmk::k_norm_up<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(vals, l), (uint32_t)t.sz[l],
                           at(pref, l), at(vals, l + 1));
    }
    size_t top = t.sz.size() - 1, m = t.sz[top];
    hipLaunchKernelGGL(msmk::k_elems_to_raw<C>, dim3(1), dim3(64), 0, s, (const uint32_t*)at(vals, top), (uint32_t)m, (uint32_t*)top_raw.p);
    std::vector<FE> v(m), pre(m), iv(m);
    HIP_TRY(hipMemcpyAsync(v.data(), top_raw.p, m * sizeof(FE), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    FE run = FE::one();
    for (size_t k = 0; k < m; k++) { pre[k] = run; run = run * v[k]; }
    FE I = run.inv();
    for (size_t k = m; k-- > 0;) { iv[k] = I * pre[k]; I = I * v[k]; }
    HIP_TRY(hipMemcpyAsync(top_raw.p, iv.data(), m * sizeof(FE), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(msmk::k_elems_from_raw<C>, dim3(1), dim3(64), 0, s, (const uint32_t*)top_raw.p, (uint32_t)m, at(inv, top));
    HIP_TRY(hipStreamSynchronize(s));   // iv is a local: the copy must have read it before it goes away
    for (size_t l = top; l-- > 0;) {
        uint32_t groups = (uint32_t)t.sz[l + 1];
        hipLaunchKernelGGL(msmk::k_norm_down<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(vals, l),
                           (const uint32_t*)at(pref, l), (const uint32_t*)at(inv, l + 1), (uint32_t)t.sz[l], at(inv, l));
    }
}

struct ScopedBufs {   // per-call device buffers of the cold entry points (normalize, table build): freed on every path
    std::vector<DevBuf*> v;
    ~ScopedBufs() { for (DevBuf* b : v) b->release(); }
};

// Host tail: Horner fold over the window sums (the chunk combine ran on the GPU).  256 doublings: ~0.1 ms on one core; a
// GPU lane would need ~2 ms for the same serial chain (cf. /root/reference/src/gpu.rs:193-209, which folds ~32k partials here).
template <class J>
J horner(const J* win, const Plan& pl) {
    J r = J::inf();
    for (int w = (int)pl.bwin - 1; w >= 0; w--) r = r.dbl_n(pl.c).add(win[w]);
    return r;
}

// the dominant kernel: one lane per work item for G1, a lane pair per work item for G2 (k_accumulate_g2_coop)
template <class C>
void launch_accumulate(hipStream_t s, const uint32_t* bases, const uint32_t* sorted, const uint32_t* offsets, const uint32_t* woff,
                       const uint32_t* order, const uint32_t* item_bucket, size_t items_cap, uint32_t* meta, uint32_t logT, uint32_t* partial) {
    // grid = the host's upper bound of the item count; the kernels read the count itself from meta[0]
    if constexpr (std::is_same<C, msmk::G2C>::value) {
        hipLaunchKernelGGL(msmk::k_accumulate_g2_coop<C>, dim3((uint32_t)((2 * items_cap + 63) / 64)), dim3(64), 0, s, bases, sorted, offsets, woff, order,
                           item_bucket, meta, logT, partial);
    } else {
        hipLaunchKernelGGL(msmk::k_accumulate<C>, dim3((uint32_t)((items_cap + 63) / 64)), dim3(64), 0, s, bases, sorted, offsets, woff, order, item_bucket,
                           meta, logT, partial);
    }
}

// The pipeline on one device.  d_bases: device-form points (tables of `stride` points when shared); d_scalars: n x 32 B on device.
//
// A call is ONE window group on the lane's stream (small inputs, shared bucket sets), or — from 2^17 points on — TWO groups of digit
// windows (top windows first; up to MAX_GROUPS on request), each with its own scratch set, with the SORT of the second group hidden
// under the accumulate kernel of the first:
//     aux stream (high priority)   sort(0) sort(1)
//     lane stream                  [sort(0) done] accumulate(0)   [accumulate(1) done] merge / reduce / combine(0), window sums out
//     second accumulate stream     [sort(1) done] accumulate(1)   [accumulate(0) done] merge / reduce / combine(1), window sums out
// What overlaps and what does not was measured, not assumed (tools/ubench_coresidency.hip, tools/pipe_scan.py):
//   * the accumulate kernel keeps two waves of 216 registers on every SIMD and the multiply-add port busy; a workgroup fits BESIDE them when
//     its waves need at most 80 registers per lane and SIMD — the sort and schedule kernels do (256-lane workgroups of <= 56 registers,
//     dynamic LDS so that the compiler does not pad the allocation), their work is LDS atomics and memory, and they run at wave priority 3:
//     the sort of group 1 costs the accumulate kernel of group 0 nothing measurable;
//   * the bucket reduction does NOT overlap usefully: its waves (184 registers) each take the place of an accumulate wave and its additions
//     compete for the same multiply-add port — run under accumulate(1), reduce(0) lengthened that kernel by what it takes alone.  The
//     groups' reductions therefore start together after the last accumulate kernel, on the two accumulate streams, with the whole call's
//     geometry: the tail is the one-group call's;
//   * the accumulate kernels of the two groups overlap at their ends (the waves of group 1 take the slots the last, short items of group 0
//     give back): two launches on two streams run as fast as one (tools/ubench_two_queues.hip).
// The reference runs one launch and folds on the host (/root/reference/src/gpu.rs:172-209).
template <class C>
typename HostCurve<C>::J run_msm(mi_ctx* ctx, DevState& d, const uint32_t* d_bases, const uint8_t* d_flags, const uint32_t* d_scalars,
                                 size_t n, unsigned fmt, bool shared, unsigned table_c, size_t stride, bool fold, WinOut* wo = nullptr,
                                 const uint8_t* host_scalars = nullptr) {
    using J = typename HostCurve<C>::J;
    using RS = typename msmk::CoopOf<C>::RS;   // lane scheme of the reduce kernel
    using CS = typename msmk::CoopOf<C>::CS;   // lane scheme of the combine levels
    constexpr int BK = msmk::Geo<C>::BK_WORDS;
    constexpr size_t CLK_BYTES = 16;   // the accumulate kernel's clock sums (kernels_common.cuh CLK_META), behind every group's window sums
    Plan pl = make_plan(n, shared ? table_c : ctx->forced_c, HostCurve<C>::cost(), shared, stride, fold);
    if (pl.c == 0) throw HipFail{"window_bits not usable for this n (sort geometry)"};
    const std::vector<Plan> groups = split_plan(pl, HostCurve<C>::cost(), n, shared, ctx->pipe_weights);
    const size_t G = groups.size();
    const bool piped = G > 1;
    d.prof.window_bits = pl.c;
    d.prof.num_windows = pl.nwin;
    d.prof.n = n;
    d.prof.window_groups = (uint32_t)G;
    d.ensure_host((size_t)pl.bwin * jac_bytes<C>() + G * CLK_BYTES);
    // host image of the window sums: group g's windows (ascending) followed by its 16 clock bytes, groups in ascending window order
    auto host_at = [&](size_t g) { return (shared ? 0 : (size_t)groups[g].win0 * jac_bytes<C>()) + (G - 1 - g) * CLK_BYTES; };

    hipStream_t sS = d.stream, sAcc[2] = {d.stream, d.stream};
    const bool phases = d.prof_level >= 2;   // see sort_and_schedule
    const bool trace = ctx->trace && phases;
    if (piped) {
        d.ensure_pipeline_streams();
        sS = d.aux_stream;
        if (ctx->pipe_two_acc_streams) sAcc[1] = d.acc2_stream;
    }
    if (piped || trace) {
        // the aux stream starts behind what the caller queued on the lane's stream (base conversion, a staged scalar shard)
        HIP_TRY(hipEventRecord(d.ev[10], d.stream));
        if (piped) HIP_TRY(hipStreamWaitEvent(sS, d.ev[10], 0));
    }
    std::vector<SortOut> so(G);
    // ---- phase 1: every group's sort + schedule (aux stream, in group order) and, behind its schedule, its accumulate kernel.  The
    // accumulate kernel is queued BEFORE the host learns the item count (only the merge launches need it): the read-back's latency is
    // hidden behind the kernel instead of idling the device between the two
    for (size_t g = 0; g < G; g++) {
        Scratch& sc = d.sc[g];
        const Plan& gp = groups[g];
        sc.pairs.ensure(gp.nchunks * 2 * BK * 4);
        sc.pairs2.ensure(((gp.nchunks >> CS::LOG_LL) + gp.bwin) * 2 * BK * 4 + (size_t)gp.bwin * jac_bytes<C>() + CLK_BYTES);
        sort_and_schedule(d, sc, sS, gp, d_scalars, d_flags, n, fmt, shared, stride, so[g], g == 0 ? host_scalars : nullptr, piped && g > 0);
        sc.partial.ensure(so[g].items_cap * BK * 4);
        hipStream_t sa = sAcc[g & 1];
        if (piped) HIP_TRY(hipStreamWaitEvent(sa, sc.ev[3], 0));
        launch_accumulate<C>(sa, d_bases, (const uint32_t*)sc.sorted.p, (const uint32_t*)sc.offsets.p, (const uint32_t*)sc.woff.p,
                             (const uint32_t*)sc.order.p, (const uint32_t*)sc.item_bucket.p, so[g].items_cap, (uint32_t*)sc.meta.p, gp.logT | (gp.logS << 16),
                             (uint32_t*)sc.partial.p);
        if (piped) HIP_TRY(hipEventRecord(sc.ev[4], sa));
    }
    // ---- phase 2: per group, on ITS accumulate stream and behind EVERY group's accumulate kernel: merges of split buckets, bucket
    // reduction, combine, window sums out.  (The host needs the schedule's counts only for the merge launches.)
    for (size_t g = 0; g < G; g++) {
        Scratch& sc = d.sc[g];
        const Plan& gp = groups[g];
        hipStream_t sr = sAcc[g & 1];
        read_schedule(sc, so[g]);
        if (piped)
            for (size_t q = 0; q < G; q++)
                if (sAcc[q & 1] != sr) HIP_TRY(hipStreamWaitEvent(sr, d.sc[q].ev[4], 0));   // the same stream's kernels precede anyway
        const uint32_t max_items = so[g].max_items;
        if (max_items > 1 && so[g].nlist) {
            // the fan-in tree over the items of split buckets: one launch per level; level l reads list l and appends list l + 1 (device
            // counters meta[3], meta[8 + l]); the grid is the host's bound of the list length: ceil(items / FAN^(l+1)) summed over the split buckets
            uint32_t* meta = (uint32_t*)sc.meta.p;
            uint32_t* lists[2] = {(uint32_t*)sc.merge_list.p, (uint32_t*)sc.merge_list2.p};
            uint64_t dd = 1, shrink = 1;
            for (uint32_t l = 0; dd < max_items; l++, dd *= msmk::MERGE_FAN, shrink *= msmk::MERGE_FAN) {
                if (8 + l + 1 >= msmk::CLK_META) throw HipFail{"merge tree deeper than its counters"};
                const uint64_t bound = std::min<uint64_t>(so[g].nlist, so[g].nlist / shrink + so[g].nsplit);
                hipLaunchKernelGGL(msmk::k_merge<CS>, dim3((uint32_t)((bound + (1u << CS::LOG_LL) - 1) >> CS::LOG_LL)), dim3(64), 0, sr, (uint32_t*)sc.partial.p,
                                   (const uint32_t*)sc.item_bucket.p, (const uint32_t*)sc.woff.p, (const uint32_t*)lists[l & 1],
                                   (const uint32_t*)(l == 0 ? meta + 3 : meta + 8 + l), lists[(l + 1) & 1], meta + 8 + l + 1, (uint32_t)dd);
            }
        }
        if (!piped && d.prof_level >= 1) HIP_TRY(hipEventRecord(sc.ev[4], sr));   // one group: the interval covers the merges as well
        // bucket reduction: one wave per chunk of chunk_buckets buckets -> (K S, T) pairs; then the per-window combine, 2^LOG_LL pairs per
        // wave and level, down to one Jacobian point per window
        bool reduced = false;
        if constexpr (std::is_same<C, msmk::G1C>::value) {   // the throughput form exists for G1 only (HostCurve<G2C>::cost() never asks for it)
            if (gp.serial_reduce) {
                hipLaunchKernelGGL(msmk::k_reduce_serial<C>, dim3((uint32_t)((gp.nchunks + 63) / 64)), dim3(64), 0, sr, (const uint32_t*)sc.partial.p,
                                   (const uint32_t*)sc.woff.p, (const uint32_t*)sc.offsets.p, (uint32_t)gp.nchunks, gp.serial_L, gp.nb, gp.chunks_per_win,
                                   (uint32_t*)sc.pairs.p);
                reduced = true;
            }
        }
        if (!reduced) {
            if (gp.serial_reduce) throw HipFail{"serial reduce requested for a curve without it"};
            hipLaunchKernelGGL(msmk::k_reduce_coop<RS>, dim3((uint32_t)gp.nchunks), dim3(64), 0, sr, (const uint32_t*)sc.partial.p,
                               (const uint32_t*)sc.woff.p, (const uint32_t*)sc.offsets.p, (uint32_t*)sc.pairs.p, gp.coop_L, gp.nb, gp.chunks_per_win);
        }
        if (phases) HIP_TRY(hipEventRecord(sc.ev[5], sr));
        uint32_t* jac_dev = (uint32_t*)((char*)sc.pairs2.p + sc.pairs2.cap - (size_t)gp.bwin * jac_bytes<C>() - CLK_BYTES);   // window sums, then the clock sums
        {
            uint32_t cpw = gp.chunks_per_win;
            uint32_t* in = (uint32_t*)sc.pairs.p;
            uint32_t* out = (uint32_t*)sc.pairs2.p;
            for (;;) {
                uint32_t cpw_out = (cpw + (1u << CS::LOG_LL) - 1) >> CS::LOG_LL;
                const bool last = cpw_out == 1;
                hipLaunchKernelGGL(msmk::k_combine<CS>, dim3(gp.bwin * cpw_out), dim3(64), 0, sr, (const uint32_t*)in, cpw, cpw_out, out,
                                   last ? jac_dev : (uint32_t*)nullptr, (const uint32_t*)sc.meta.p);
                if (last) break;
                std::swap(in, out);   // the levels shrink by 2^LOG_LL: ping-pong between the two pair buffers
                cpw = cpw_out;
            }
        }
        if (phases) HIP_TRY(hipEventRecord(sc.ev[6], sr));
        // window sums of the group, in their place among the call's windows (window 0 first; shared bucket sets: the one sum)
        const size_t at = shared ? 0 : (size_t)gp.win0 * jac_bytes<C>();
        if (wo) {   // ... they stay on the device (the caller exchanges them: mi_msm_g1_device_windows); no fold here, only the clock sums come back
            HIP_TRY(hipMemcpyAsync((char*)wo->d_out + at, jac_dev, (size_t)gp.bwin * jac_bytes<C>(), hipMemcpyDeviceToDevice, sr));
            HIP_TRY(hipMemcpyAsync((char*)d.h_pairs + host_at(g) + (size_t)gp.bwin * jac_bytes<C>(), (const char*)jac_dev + (size_t)gp.bwin * jac_bytes<C>(), CLK_BYTES,
                                   hipMemcpyDeviceToHost, sr));
        } else
            HIP_TRY(hipMemcpyAsync((char*)d.h_pairs + host_at(g), jac_dev, (size_t)gp.bwin * jac_bytes<C>() + CLK_BYTES, hipMemcpyDeviceToHost, sr));
        if (phases || piped) HIP_TRY(hipEventRecord(sc.ev[7], sr));
    }
    if (wo) {
        wo->info.window_bits = pl.c;
        wo->info.num_windows = pl.bwin;
    }
    // ---- phase 3: the host tail.  Horner fold over the window sums (top window first), group by group as they arrive
    J r = J::inf();
    double fold_ms = 0;
    unsigned long long clk_ticks = 0, ref_ticks = 0;   // the accumulate kernels' s_memtime / s_memrealtime sums
    for (size_t g = 0; g < G; g++) {
        const Plan& gp = groups[g];
        if (piped) HIP_TRY(hipEventSynchronize(d.sc[g].ev[7]));
        else HIP_TRY(hipStreamSynchronize(sS));
        const char* img = (const char*)d.h_pairs + host_at(g);   // pinned bytes the D2H copy wrote
        auto t0 = std::chrono::steady_clock::now();
        for (int w = (int)gp.bwin - 1; w >= 0 && !wo; w--) {
            J wsum;
            memcpy(&wsum, img + (size_t)w * jac_bytes<C>(), sizeof wsum);
            r = r.dbl_n(pl.c).add(wsum);
        }
        fold_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long ck[2];
        memcpy(ck, img + (size_t)gp.bwin * jac_bytes<C>(), sizeof ck);
        clk_ticks += ck[0];
        ref_ticks += ck[1];
    }
    if (ref_ticks) {   // shader clock the accumulate kernels ran at: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
        d.prof.accumulate_clock_ticks += clk_ticks;
        d.prof.accumulate_ref_ticks += ref_ticks;
        d.prof.accumulate_clock_ghz = 0.1 * (double)d.prof.accumulate_clock_ticks / (double)d.prof.accumulate_ref_ticks;
    }
    if (piped) HIP_TRY(hipStreamSynchronize(sS));   // nothing of this call is left on any stream (the aux stream ended with the last schedule)
    HIP_TRY(hipGetLastError());

    for (size_t g = 0; g < G; g++) {
        Scratch& sc = d.sc[g];
        if (phases) {
            d.prof.digits_ms += ev_ms(sc.ev[0], sc.ev[1]);      // digits + coarse partition (4 kernels)
            d.prof.scatter_ms += ev_ms(sc.ev[1], sc.ev[2]);     // fine sort in LDS
            d.prof.scan_ms += ev_ms(sc.ev[2], sc.ev[3]);        // schedule (3 kernels)
            d.prof.combine_ms += ev_ms(sc.ev[5], sc.ev[6]);
            d.prof.d2h_ms += ev_ms(sc.ev[6], sc.ev[7]);
        }
        d.prof.accumulate_adds += so[g].entries;
        d.prof.work_items += so[g].nitems;
        d.prof.max_items_per_bucket = std::max(d.prof.max_items_per_bucket, so[g].max_items);
    }
    // accumulate_ms — one group: the kernel and the merges of split buckets; pipelined: from the first group's schedule (its accumulate
    // kernel starts there) to the end of the LAST accumulate kernel to finish (the launches overlap: their sum would count twice; the sort
    // of the later groups runs inside this span).  reduce_ms — pipelined: from there to the last reduce kernel's end (they run together).
    if (piped) {
        double span = 0, red = 0;
        size_t last = 0;
        for (size_t g = 0; g < G; g++) {
            const double e = ev_ms(d.sc[0].ev[3], d.sc[g].ev[4]);
            if (e > span) { span = e; last = g; }
        }
        d.prof.accumulate_ms += span;
        if (phases) {
            for (size_t g = 0; g < G; g++) red = std::max(red, ev_ms(d.sc[last].ev[4], d.sc[g].ev[5]));
            d.prof.reduce_ms += red;
        }
    } else {
        if (d.prof_level >= 1) d.prof.accumulate_ms += ev_ms(d.sc[0].ev[3], d.sc[0].ev[4]);
        if (phases) d.prof.reduce_ms += ev_ms(d.sc[0].ev[4], d.sc[0].ev[5]);
    }
    d.prof.host_fold_ms += fold_ms;
    if (trace) {   // ARKBLST_AMD_TRACE=1 with profile level 2: the call's phase boundaries on the GPU, microseconds from the call's entry
        std::string line = "[arkblst trace] n=" + std::to_string(n) + " c=" + std::to_string(pl.c) + " groups=" + std::to_string(G) + "\n";
        for (size_t g = 0; g < G; g++) {
            char b[256];
            Scratch& sc = d.sc[g];
            auto at = [&](int k) { return ev_ms(d.ev[10], sc.ev[k]) * 1e3; };
            snprintf(b, sizeof b, "  group %zu windows [%u,%u): sort %.0f..%.0f..%.0f sched %.0f | acc ..%.0f | reduce ..%.0f combine ..%.0f out %.0f\n", g, groups[g].win0,
                     groups[g].win0 + groups[g].nwin, at(0), at(1), at(2), at(3), at(4), at(5), at(6), at(7));
            line += b;
        }
        fputs(line.c_str(), stderr);
    }
    return r;
}

// One device's share of an MSM call. bases: host raw pointer for this shard or nullptr (= resident, starting at resident index r0).
// cached (base-set cache, mi_msm_set_base_cache): with bases == nullptr the resident set to read instead of the context's; with
// bases != nullptr the entry being FILLED — the call's converted points are written into it instead of the lane's scratch, so that a
// miss costs exactly what the uncached call costs.
template <class C>
typename HostCurve<C>::J device_msm(mi_ctx* ctx, DevState& d, const uint8_t* bases, size_t r0, const uint8_t* scalars, bool scalars_on_device,
                                    size_t n, unsigned fmt, int stage_from_dev = -1, WinOut* wo = nullptr, Resident* cached = nullptr) {
    using J = typename HostCurve<C>::J;
    HIP_TRY(hipSetDevice(d.dev));
    d.prof = mi_profile{};
    d.prof_level = ctx->profile_level;
    auto t0 = std::chrono::steady_clock::now();
    J total = J::inf();
    if (n == 0) return total;
    hipStream_t s = d.stream;
    Resident& res = (cached && !bases) ? *cached : d.res[HostCurve<C>::IDX];
    const bool shared = !bases && res.tables > 1;
    const size_t part_max = max_part(ctx);
    if (cached && bases) {   // the whole shard in one buffer, whatever the number of passes
        try {
            cached->buf.ensure(n * msmk::Geo<C>::PT_WORDS * 4);
            cached->flags.ensure(n);
        } catch (const HipFail& f) {
            // no room for another cached copy of the shard: the call itself needs only one pass's worth in the lane's scratch, which it
            // has had all along — give the entry up (msm_impl sees n = 0 and does not publish it) and run uncached
            if (!f.oom && f.msg.find("out of memory") == std::string::npos) throw;
            (void)hipGetLastError();
            cached->buf.release();
            cached->flags.release();
            cached->n = 0;
            cached = nullptr;
        }
    }
    DevBuf& conv_bases = cached && bases ? cached->buf : d.call_bases;   // where the call's converted points go
    DevBuf& conv_flags = cached && bases ? cached->flags : d.call_flags;
    if (wo && n > part_max) throw HipFail{"device_windows: n exceeds one pass of the pipeline (2^26 points per device)", false, true};
    // A failure between enqueueing the chunked H2D copies (copy stream) and the call's final synchronisation must not return while
    // copies still read the caller's host buffers and write this lane's scratch: drain both streams before the error leaves
    struct DrainOnFailure {
        DevState& d;
        int live = std::uncaught_exceptions();
        ~DrainOnFailure() {
            if (std::uncaught_exceptions() > live) {
                (void)hipStreamSynchronize(d.copy_stream);
                (void)hipStreamSynchronize(d.stream);
                if (d.acc2_stream) (void)hipStreamSynchronize(d.acc2_stream);   // a pipelined call has work on all of them
                if (d.aux_stream) (void)hipStreamSynchronize(d.aux_stream);
                (void)hipGetLastError();
            }
        }
    } drain{d};
    for (size_t lo = 0; lo < n; lo += part_max) {   // one pass unless n exceeds the per-pass limit
        // between two passes nothing of this call is on the GPU (run_msm returned): the place where the reference's driver asks maybe_abort
        if (lo > 0 && abort_requested(ctx)) throw HipFail{"aborted by the caller's abort check between two passes", false, false, true};
        const size_t m = std::min(part_max, n - lo);
        if (d.prof_level >= 2) HIP_TRY(hipEventRecord(d.ev[0], s));
        // Host slices (the trait's call shape) cross PCIe in chunks on the lane's copy stream, each consumed as it lands: bases first
        // (k_ingest per chunk), then the scalars (count pass of the sort per chunk, inside sort_and_schedule).
        const uint32_t* d_scalars;
        const uint8_t* host_scalars = nullptr;
        if (scalars_on_device && stage_from_dev < 0) {
            d_scalars = reinterpret_cast<const uint32_t*>(scalars + lo * 32);
        } else if (scalars_on_device) {   // the vector lives on a device this one cannot read: peer copy of the shard
            d.scalars.ensure(m * 32);
            HIP_TRY(hipMemcpyPeerAsync(d.scalars.p, d.dev, scalars + lo * 32, stage_from_dev, m * 32, s));
            d_scalars = reinterpret_cast<const uint32_t*>(d.scalars.p);
        } else {
            d.scalars.ensure(m * 32);
            d_scalars = reinterpret_cast<const uint32_t*>(d.scalars.p);
            host_scalars = scalars + lo * 32;
        }
        const uint32_t* d_bases;
        const uint8_t* d_flags;
        if (bases) {
            d.raw.ensure(m * aff_bytes<C>());
            const size_t base0 = cached ? lo : 0;   // a cache entry keeps every pass; the lane's scratch is reused per pass
            if (!cached) {
                conv_bases.ensure(m * msmk::Geo<C>::PT_WORDS * 4);
                conv_flags.ensure(m);
            }
            const size_t K = std::min<size_t>(8, std::max<size_t>(1, m >> 16));
            for (size_t j = 0; j < K; j++) {
                const size_t p0 = m * j / K, p1 = m * (j + 1) / K;
                HIP_TRY(hipMemcpyAsync((char*)d.raw.p + p0 * aff_bytes<C>(), bases + (lo + p0) * aff_bytes<C>(), (p1 - p0) * aff_bytes<C>(),
                                       hipMemcpyHostToDevice, d.copy_stream));
                HIP_TRY(hipEventRecord(d.cev[1 + j], d.copy_stream));
                HIP_TRY(hipStreamWaitEvent(s, d.cev[1 + j], 0));
                if (p1 > p0)
                    ingest<C>(d, (const char*)d.raw.p + p0 * aff_bytes<C>(), true, p1 - p0, (uint32_t*)conv_bases.p + (base0 + p0) * msmk::Geo<C>::PT_WORDS,
                              (uint8_t*)conv_flags.p + base0 + p0);
            }
            if (d.prof_level >= 2) HIP_TRY(hipEventRecord(d.ev[1], s));
            d_bases = reinterpret_cast<const uint32_t*>(conv_bases.p) + base0 * msmk::Geo<C>::PT_WORDS;
            d_flags = reinterpret_cast<const uint8_t*>(conv_flags.p) + base0;
        } else {
            if (d.prof_level >= 2) HIP_TRY(hipEventRecord(d.ev[1], s));
            d_bases = reinterpret_cast<const uint32_t*>(res.buf.p) + (r0 + lo) * msmk::Geo<C>::PT_WORDS;
            d_flags = reinterpret_cast<const uint8_t*>(res.flags.p) + r0 + lo;
        }
        // sign fold (one window fewer at c = 15 / 17) only over a resident set whose every point passed the subgroup check on this
        // context; bases handed over with the call are multiplied by the integer s, as the reference does (src/g1.rs:614-617)
        J r = run_msm<C>(ctx, d, d_bases, d_flags, d_scalars, m, fmt, shared, res.table_c, res.n, !bases && res.validated, wo, host_scalars);
        total = lo == 0 ? r : total.add(r);
        // h2d_ms: bases (their chunks interleave with k_ingest on the main stream) + scalars (copy stream, first to last chunk; the
        // sort's count pass runs underneath, so digits_ms of a host-scalar call includes waiting for the chunks)
        if (d.prof_level >= 2) d.prof.h2d_ms += ev_ms(d.ev[0], d.ev[1]) + (host_scalars ? ev_ms(d.cev[0], d.cev[9]) : 0.0);
    }
    d.prof.n = n;
    d.prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return total;
}

// Resident base set.  precompute_c == 0: plain bases.  Otherwise W = num_windows(c, false) tables T_j[i] = 2^(c j) P_i (affine, device
// form): every window of a later MSM then shares ONE bucket set (no per-window reduce, no Horner doublings) and c can be larger.
template <class C>
void build_resident(mi_ctx* ctx, DevState& d, Resident& res, const uint8_t* bases, size_t n, unsigned precompute_c, bool bases_on_device = false) {
    using F = typename C::F;
    constexpr size_t PTB = (size_t)msmk::Geo<C>::PT_WORDS * 4, SLOTB = (size_t)msmk::Geo<C>::SLOT * 4, BKB = (size_t)msmk::Geo<C>::BK_WORDS * 4;
    (void)sizeof(F);
    HIP_TRY(hipSetDevice(d.dev));
    unsigned c = 0, W = 1;
    if (precompute_c) {
        c = precompute_c == 1 ? make_plan(n, 0, HostCurve<C>::cost(), true, n).c : precompute_c;
        // c == 0: no window size fits the entry encoding (n x windows > 2^30 entries, i.e. more than ~9e7 points per device)
        if (c < 7 || c > 22 || make_plan(n, c, HostCurve<C>::cost(), true, n).c == 0)
            throw HipFail{"window_bits not usable for precomputed tables of this size", false, true};
        W = num_windows(c, false);   // one more than a validated set's calls use at c = 15 / 17
    }
    res.buf.ensure_fit(n * W * PTB);
    res.flags.ensure(n);
    res.tables = 1;
    res.table_c = 0;
    res.validated = false;
    ingest<C>(d, bases, bases_on_device, n, (uint32_t*)res.buf.p, (uint8_t*)res.flags.p);
    if (W > 1) {
        InvTree t(n);
        DevBuf proj, vals, pref, inv, top_raw;
        ScopedBufs guard{{&proj, &vals, &pref, &inv, &top_raw}};
        proj.ensure(n * BKB);
        vals.ensure(t.total * SLOTB); pref.ensure(t.total * SLOTB); inv.ensure(t.total * SLOTB);
        top_raw.ensure(64 * sizeof(decltype(HostCurve<C>::J::inf().x)));
        const uint32_t grid = (uint32_t)((n + 255) / 256);
        for (unsigned j = 1; j < W; j++) {
            const uint32_t* prev = (const uint32_t*)((const char*)res.buf.p + (size_t)(j - 1) * n * PTB);
            uint32_t* next = (uint32_t*)((char*)res.buf.p + (size_t)j * n * PTB);
            hipLaunchKernelGGL(msmk::k_table_dbl<C>, dim3(grid), dim3(256), 0, d.stream, prev, (const uint8_t*)res.flags.p, (uint32_t)n, c,
                               (uint32_t*)proj.p, (uint32_t*)vals.p);
            invert_tree<C>(d, t, vals, pref, inv, top_raw);
            hipLaunchKernelGGL(msmk::k_table_affine<C>, dim3(grid), dim3(256), 0, d.stream, (const uint32_t*)proj.p, (const uint32_t*)inv.p,
                               (const uint8_t*)res.flags.p, (uint32_t)n, next);
        }
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(hipGetLastError());
        res.tables = W;
        res.table_c = c;
    }
    HIP_TRY(hipStreamSynchronize(d.stream));
    (void)ctx;
}

// where a new resident base set comes from
enum class BaseSrc {
    HostAffine,     // mi_msm_g1_set_bases: blst_p1_affine in host memory
    DeviceAffine,   // mi_msm_g1_set_bases_device: the same form, already in device memory (single-device contexts)
    HostJacobian    // mi_msm_g1_set_bases_from_jacobian: blst_p1 in host memory, normalised on the GPU on the way (normalize_batch feeding msm,
                    // /root/reference/src/g1.rs:597-599 -> 604, without returning to the host in between)
};
template <class C> void normalize_run(mi_ctx* ctx, DevState& d, const void* h_in, const void* d_in_user, size_t n, void* h_out, void* d_out_user, bool publish_profile);

template <class C>
int set_bases_impl(mi_ctx* ctx, const void* bases, size_t n, unsigned precompute_c, BaseSrc src = BaseSrc::HostAffine) {
    if (!ctx || (n && !bases)) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (precompute_c > 1 && (precompute_c < 7 || precompute_c > 22)) return fail(ctx, MI_E_INVALID, "window_bits must be 0 (choose) or 7..22");
    if ((n + ctx->devs.size() - 1) / ctx->devs.size() > (1ull << 31)) return fail(ctx, MI_E_INVALID, "more than 2^31 points per device");
    if (src == BaseSrc::DeviceAffine && ctx->devs.size() != 1) return fail(ctx, MI_E_INVALID, "set_bases_device needs a single-device context");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        size_t g = ctx->devs.size();
        if (src == BaseSrc::DeviceAffine && n && device_of_ptr(bases, "d_bases", 4) != ctx->devs[0].dev)
            return fail(ctx, MI_E_INVALID, "d_bases must live on the context's device");
        const size_t in_bytes = src == BaseSrc::HostJacobian ? jac_bytes<C>() : aff_bytes<C>();
        std::vector<PartErr> errs(g);
        for_each_device(lk, g, [&](size_t k) {
            guarded_part(errs[k], [&] {
                DevState& d = ctx->devs[k];
                auto& res = d.res[HostCurve<C>::IDX];
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                res.lo = lo;
                res.n = hi - lo;
                res.tables = 1;
                res.table_c = 0;
                res.validated = false;
                if (hi == lo) return;
                const uint8_t* shard = (const uint8_t*)bases + lo * in_bytes;
                if (src == BaseSrc::HostJacobian) {
                    normalize_run<C>(ctx, d, shard, nullptr, hi - lo, nullptr, nullptr, false);   // affine points (reference form) in d.io_out
                    build_resident<C>(ctx, d, res, (const uint8_t*)d.io_out.p, hi - lo, precompute_c, true);
                } else {
                    build_resident<C>(ctx, d, res, shard, hi - lo, precompute_c, src == BaseSrc::DeviceAffine);
                }
            });
        });
        for (size_t k = 0; k < g; k++)
            if (errs[k].code != MI_OK) {
                for (size_t q = 0; q < g; q++) { auto& r = ctx->devs[q].res[HostCurve<C>::IDX]; r.n = 0; r.tables = 1; }   // all or nothing
                return fail(ctx, errs[k].code, errs[k].msg);
            }
        return MI_OK;
    });
}

// device slot k's shard of a new resident set from points already decoded into DEVICE memory in the reference's affine form (the caller —
// mi_msm_g1_set_bases_from_compressed, points.hip — holds the context exclusively and runs this on the slot's worker thread).  validated:
// the decoder ran Valid::check on every point.
template <class C>
void install_resident(mi_ctx* ctx, size_t k, const void* d_affine, size_t lo, size_t n, bool validated) {
    DevState& d = ctx->devs[k];
    auto& res = d.res[HostCurve<C>::IDX];
    res.lo = lo;
    res.n = n;
    res.tables = 1;
    res.table_c = 0;
    res.validated = false;
    if (n == 0) return;
    build_resident<C>(ctx, d, res, (const uint8_t*)d_affine, n, 0, true);
    res.validated = validated;
}

template <class C>
int msm_impl(mi_ctx* ctx, const void* bases_v, const uint8_t* scalars, bool scalars_on_device, size_t n, unsigned fmt, void* out) {
    using J = typename HostCurve<C>::J;
    const uint8_t* bases = static_cast<const uint8_t*>(bases_v);
    if (!ctx || !out || (n && !scalars) || fmt > 1) return fail(ctx, MI_E_INVALID, "invalid argument");
    LaneLock lane(ctx, false);
    if (abort_requested(ctx)) return fail(ctx, MI_E_ABORTED, "aborted by the caller's abort check");   // src/gpu.rs:133-137
    std::vector<DevState>& devs = lane.devs();
    return guarded(ctx, [&]() -> int {
        size_t g = devs.size();
        std::vector<J> part(g, J::inf());
        std::vector<PartErr> errs(g);
        // resident path: each device covers the overlap of [0, n) with its resident shard
        if (!bases) {
            size_t have = 0;
            for (auto& d : devs) have += d.res[HostCurve<C>::IDX].n;
            if (have == 0 && n) return fail(ctx, MI_E_NO_BASES, "no resident base set for this group");
            if (n > have) return fail(ctx, MI_E_INVALID, "n exceeds the resident base set");
        }
        // base-set cache (opt-in): a host base vector this context has converted before is read from HBM instead of crossing PCIe again.
        // The fingerprint covers EVERY byte of the vector (round 6) and runs on the lane's persistent helper threads UNDER the GPU work: an
        // entry with the same (pointer, n) is used speculatively and confirmed before the result leaves; a miss fills its entry first and
        // learns its key at the end.
        constexpr size_t CACHE_MIN_POINTS = 1u << 12;   // below that the upload is cheaper than the bookkeeping is worth
        std::shared_ptr<BaseCacheEntry> hit, fill;
        std::shared_ptr<HashPool> pool;
        if (bases && n >= CACHE_MIN_POINTS && cache_begin(ctx, HostCurve<C>::IDX, bases, n, hit)) {
            std::shared_ptr<HashPool>& slot = ctx->hash_pool[lane.lane == 1 ? 1 : 0];   // this call holds the lane: the pool is its own
            if (!slot) slot = std::make_shared<HashPool>(hash_pool_threads(), HashKey{ctx->hash_seed, ctx->hash_mult});
            pool = slot;
            pool->start(bases, n * aff_bytes<C>());
        }
        struct JoinFp {   // the helper threads read the caller's bases: the job is waited for on every path out of this call
            std::shared_ptr<HashPool>& p;
            ~JoinFp() { if (p && p->busy()) (void)p->finish(); }
        } join_fp{pool};
        // device-resident scalars: the shard of device k is read by device k — in place when the vector lives there, through ONE peer
        // copy of the shard otherwise (can_read)
        int owner = -1;
        if (scalars_on_device && n) owner = device_of_ptr(scalars, "d_scalars");
        auto t0 = std::chrono::steady_clock::now();
        auto run = [&]() -> int {   // one pass over the devices with the current (hit, fill)
            for (auto& e : errs) e = PartErr{};
            for_each_device(lane, g, [&](size_t k) {
                guarded_part(errs[k], [&] {
                    DevState& d = devs[k];
                    size_t lo, hi;
                    if (bases) {
                        shard_range(n, g, k, lo, hi);
                    } else {
                        auto& res = d.res[HostCurve<C>::IDX];
                        lo = std::min(n, res.lo);
                        hi = std::min(n, res.lo + res.n);
                    }
                    const int stage = scalars_on_device && hi > lo && !can_read(ctx, k, owner) ? owner : -1;
                    if (hit) {
                        part[k] = device_msm<C>(ctx, d, nullptr, 0, scalars + lo * 32, scalars_on_device, hi - lo, fmt, stage, nullptr, &hit->shard[k]);
                    } else {
                        if (fill) { fill->shard[k].lo = lo; fill->shard[k].n = hi - lo; }
                        part[k] = device_msm<C>(ctx, d, bases ? bases + lo * aff_bytes<C>() : nullptr, 0, scalars + lo * 32, scalars_on_device, hi - lo, fmt, stage,
                                                nullptr, fill ? &fill->shard[k] : nullptr);
                    }
                });
            });
            for (size_t k = 0; k < g; k++)
                if (errs[k].code != MI_OK) return fail(ctx, errs[k].code, errs[k].msg);
            return MI_OK;
        };
        auto new_entry = [&]() {
            auto e = std::make_shared<BaseCacheEntry>();
            e->ptr = bases; e->n = n;
            e->shard.resize(g);
            for (auto& d : devs) e->devs.push_back(d.dev);
            return e;
        };
        if (pool && !hit) fill = new_entry();
        int rc = run();
        if (rc != MI_OK) return rc;
        if (pool) {
            const Fp128 fp = pool->finish();
            if (hit && hit->fp != fp) {
                // mis-speculation: the vector under this pointer changed since the candidate was built.  Another entry may hold the
                // new content (a buffer that alternates between two sets); otherwise convert it now.
                hit = cache_find(ctx, HostCurve<C>::IDX, bases, n, fp);
                if (!hit) fill = new_entry();
                rc = run();
                if (rc != MI_OK) return rc;
            }
            if (fill)   // a shard that found no memory for its cached copy ran uncached (device_msm): the entry is incomplete, not published
                for (size_t k = 0; k < g; k++) {
                    size_t lo, hi;
                    shard_range(n, g, k, lo, hi);
                    if (fill && fill->shard[k].n != hi - lo) fill.reset();
                }
            cache_finish(ctx, HostCurve<C>::IDX, hit, fill, fp);
        }
        J r = J::inf();
        for (size_t k = 0; k < g; k++) r = r.add(part[k]);
        memcpy(out, &r, sizeof r);
        // report the slowest device's profile
        size_t slow = 0;
        for (size_t k = 1; k < g; k++)
            if (devs[k].prof.total_ms > devs[slow].prof.total_ms) slow = k;
        mi_profile pr = devs[slow].prof;
        pr.n = n;
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

// The pipeline of mi_msm_g{1,2}_device up to the per-window sums, which stay in the caller's DEVICE buffer (no D2H, no host fold):
// the exchange step of a one-process-per-GPU deployment starts from device memory (RCCL all-gather of the window sums, one D2H
// of the gathered block, mi_g{1,2}_fold_windows on the host).  Single-device contexts, resident bases, one pass.
template <class C>
int msm_windows_impl(mi_ctx* ctx, const uint8_t* d_scalars, size_t n, unsigned fmt, void* d_out, mi_window_info* info) {
    if (!ctx || !d_out || !info || (n && !d_scalars) || fmt > 1) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (ctx->devs.size() != 1) return fail(ctx, MI_E_INVALID, "device_windows needs a single-device context (one context per rank)");
    LaneLock lane(ctx, false);
    std::vector<DevState>& devs = lane.devs();
    return guarded(ctx, [&]() -> int {
        DevState& d = devs[0];
        auto& res = d.res[HostCurve<C>::IDX];
        if (res.n == 0 && n) return fail(ctx, MI_E_NO_BASES, "no resident base set for this group");
        if (n > res.n) return fail(ctx, MI_E_INVALID, "n exceeds the resident base set");
        info->window_bits = 0;
        info->num_windows = 0;
        if (n == 0) return MI_OK;
        // the window sums are written by a plain device-to-device copy on this context's device: the buffer must live there; a scalar
        // vector on another GPU is staged by ONE peer copy, exactly as mi_msm_g1_device does, never read in place
        const int owner = device_of_ptr(d_scalars, "d_scalars");
        if (device_of_ptr(d_out, "d_out_windows") != d.dev)
            return fail(ctx, MI_E_INVALID, "d_out_windows must live on the context's device");
        auto t0 = std::chrono::steady_clock::now();
        WinOut wo;
        wo.d_out = static_cast<uint32_t*>(d_out);
        (void)device_msm<C>(ctx, d, nullptr, 0, d_scalars, true, n, fmt, can_read(ctx, 0, owner) ? -1 : owner, &wo);
        *info = wo.info;
        mi_profile pr = d.prof;
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

// k MSMs over the resident base set, two in flight (one per lane): two persistent job pullers share the k jobs
template <class C, class Out>
int msm_batch_impl(mi_ctx* ctx, const uint8_t* const* scalars, bool scalars_on_device, size_t k, size_t n, unsigned fmt, Out* out) {
    if (!ctx || (k && (!scalars || !out))) return fail(ctx, MI_E_INVALID, "invalid argument");
    for (size_t j = 0; j < k; j++)
        if (n && !scalars[j]) return fail(ctx, MI_E_INVALID, "null scalar vector");
    return guarded(ctx, [&]() -> int {
        std::atomic<size_t> next{0};
        std::atomic<int> first_err{MI_OK};
        std::string first_msg;
        std::mutex msg_mu;
        auto worker = [&]() noexcept {
            for (;;) {
                size_t j = next.fetch_add(1);
                if (j >= k || first_err.load() != MI_OK) break;
                int rc = msm_impl<C>(ctx, nullptr, scalars[j], scalars_on_device, n, fmt, &out[j]);   // never throws (guarded)
                int ok = MI_OK;
                if (rc != MI_OK && first_err.compare_exchange_strong(ok, rc)) {
                    std::lock_guard<std::mutex> lk(msg_mu);
                    first_msg = tls_error();
                }
            }
        };
        if (k > 1) {   // two persistent job pullers (one per lane), created on the first batch call: nothing is spawned per call
            std::lock_guard<std::mutex> lk(ctx->batch_mu);
            if (!ctx->batch_workers) ctx->batch_workers.reset(new DeviceWorkers(NLANES));
            std::function<void(size_t)> f = [&](size_t) { worker(); };
            ctx->batch_workers->run(f);
        } else {
            worker();
        }
        if (first_err.load() != MI_OK) return fail(ctx, first_err.load(), first_msg);
        return MI_OK;
    });
}

// normalize_batch on one device.  Host pointers cross PCIe in chunks (io_chunks.hpp):
//   per chunk, as it lands     k_norm_up0: Z straight from the caller's points -> prefix products of level 0, group totals = level 1
//   once                       levels >= 1 of the product tree up, the <= 64 top values inverted on the host, the tree down to level 1 — at FULL
//                              size: a tree kernel has one lane per group of NORM_K values, and cut into chunks its launches left most of the
//                              machine idle (measured: 10 ms instead of 1 ms of kernels for 2^18 G2 points)
//   per chunk                  k_norm_down0_final: level 0's inverses and the affine points in one pass; the chunk's points leave on the d2h
//                              stream while the next chunk is converted
// (n <= 64: one level, k_norm_load / host inversion / k_norm_final.)  A NULL host pointer with a device pointer: the
// data is / stays in device memory (mi_g1_normalize_batch_device, mi_msm_g1_set_bases_from_jacobian: d_out_user == nullptr leaves the
// affine points in d.io_out).  Scratch lives in the lane's DevState: nothing is allocated in steady state (round 5: six hipMalloc + hipFree per call).
template <class C>
void normalize_run(mi_ctx* ctx, DevState& d, const void* h_in, const void* d_in_user, size_t n, void* h_out, void* d_out_user, bool publish_profile) {
    using J = typename HostCurve<C>::J;
    using FE = decltype(J::inf().x);
    constexpr size_t SLOTB = (size_t)msmk::Geo<C>::SLOT * 4, K2 = (size_t)msmk::NORM_K * msmk::NORM_K;
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    const InvTree t(n);
    const size_t top = t.sz.size() - 1;
    const uint32_t* raw_in = (const uint32_t*)d_in_user;
    if (!raw_in) { d.io_in.ensure(n * jac_bytes<C>()); raw_in = (const uint32_t*)d.io_in.p; }
    uint32_t* raw_out = (uint32_t*)d_out_user;
    if (!raw_out) { d.io_out.ensure(n * aff_bytes<C>()); raw_out = (uint32_t*)d.io_out.p; }
    // more than one level (n > 64): level 0 is fused with the conversions around it (k_norm_up0, k_norm_down0_final) and keeps only its prefix
    // products — the value and inverse arrays start at level 1
    const bool fused = top >= 1;
    const size_t skip = fused ? t.sz[0] : 0;
    d.nv_vals.ensure((t.total - skip) * SLOTB);
    d.nv_pref.ensure(t.total * SLOTB);
    d.nv_inv.ensure((t.total - skip) * SLOTB);
    d.nv_top.ensure(64 * sizeof(FE));
    auto at = [&](DevBuf& b, size_t l, size_t idx) {
        const size_t base = (&b == &d.nv_pref) ? t.off[l] : t.off[l] - skip;   // level 0 of vals / inv does not exist when fused (never asked for)
        return (uint32_t*)((char*)b.p + (base + idx) * SLOTB);
    };
    // range [lo, lo + cnt) of level 0 seen from level l
    auto level_range = [&](size_t l, size_t lo, size_t cnt, size_t& llo, size_t& lcnt) {
        size_t div = 1;
        for (size_t q = 0; q < l; q++) div *= msmk::NORM_K;
        llo = lo / div;
        lcnt = (lo + cnt + div - 1) / div - llo;
    };
    auto up = [&](size_t l, size_t lo, size_t cnt) {       // level l -> prefix products of its groups, group totals = level l + 1
        size_t llo, lcnt;
        level_range(l, lo, cnt, llo, lcnt);
        const uint32_t groups = (uint32_t)((lcnt + msmk::NORM_K - 1) / msmk::NORM_K);
        hipLaunchKernelGGL(msmk::k_norm_up<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(d.nv_vals, l, llo), (uint32_t)lcnt,
                           at(d.nv_pref, l, llo), at(d.nv_vals, l + 1, llo / msmk::NORM_K));
    };
    auto down = [&](size_t l, size_t lo, size_t cnt) {     // inverses of level l from the inverses of level l + 1
        size_t llo, lcnt;
        level_range(l, lo, cnt, llo, lcnt);
        const uint32_t groups = (uint32_t)((lcnt + msmk::NORM_K - 1) / msmk::NORM_K);
        hipLaunchKernelGGL(msmk::k_norm_down<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(d.nv_vals, l, llo),
                           (const uint32_t*)at(d.nv_pref, l, llo), (const uint32_t*)at(d.nv_inv, l + 1, llo / msmk::NORM_K), (uint32_t)lcnt,
                           at(d.nv_inv, l, llo));
    };
    const size_t local = fused ? 1 : 0;                     // tree levels handled per chunk: level 0 when it is fused with the conversions
    const IoPlan p = io_plan(n, h_in != nullptr || h_out != nullptr, K2);
    const IoOut outs[1] = {{h_out, raw_out, aff_bytes<C>()}};
    IoDrain drain(d);
    // ---- per chunk, as it lands: Z values and the chunk-local up-sweep
    for (size_t j = 0; j < p.K; j++) {
        size_t lo, cnt;
        io_range(p, n, j, lo, cnt);
        io_feed(d, p, n, j, h_in, const_cast<uint32_t*>(raw_in), jac_bytes<C>());
        HIP_TRY(hipEventRecord(d.iev[0][j], s));
        if (fused)
            hipLaunchKernelGGL(msmk::k_norm_up0<C>, dim3((uint32_t)(((cnt + msmk::NORM_K - 1) / msmk::NORM_K + 255) / 256)), dim3(256), 0, s,
                               raw_in + lo * msmk::Geo<C>::RAW_JAC, (uint32_t)cnt, at(d.nv_pref, 0, lo), at(d.nv_vals, 1, lo / msmk::NORM_K));
        else
            hipLaunchKernelGGL(msmk::k_norm_load<C>, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, raw_in + lo * msmk::Geo<C>::RAW_JAC, (uint32_t)cnt,
                               at(d.nv_vals, 0, lo));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(d.iev[1][j], s));
    }
    // ---- once: upper up-sweep, top inversion on the host (one Fermat inversion; a single GPU lane would need ~1 ms), upper down-sweep
    HIP_TRY(hipEventRecord(d.ev[0], s));
    for (size_t l = local; l < top; l++) up(l, 0, n);
    const size_t m = t.sz[top];
    hipLaunchKernelGGL(msmk::k_elems_to_raw<C>, dim3(1), dim3(64), 0, s, (const uint32_t*)at(d.nv_vals, top, 0), (uint32_t)m, (uint32_t*)d.nv_top.p);
    d.ensure_host(2 * 64 * sizeof(FE));
    FE* hv = static_cast<FE*>(d.h_pairs);                   // pinned: [0, 64) the top values, [64, 128) their inverses
    HIP_TRY(hipMemcpyAsync(hv, d.nv_top.p, m * sizeof(FE), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    {
        FE pre[64], run = FE::one();
        for (size_t k = 0; k < m; k++) { pre[k] = run; run = run * hv[k]; }
        FE I = run.inv();
        for (size_t k = m; k-- > 0;) { hv[64 + k] = I * pre[k]; I = I * hv[k]; }
    }
    HIP_TRY(hipMemcpyAsync(d.nv_top.p, hv + 64, m * sizeof(FE), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(msmk::k_elems_from_raw<C>, dim3(1), dim3(64), 0, s, (const uint32_t*)d.nv_top.p, (uint32_t)m, at(d.nv_inv, top, 0));
    for (size_t l = top; l-- > local;) down(l, 0, n);
    HIP_TRY(hipEventRecord(d.ev[1], s));
    // ---- per chunk: chunk-local down-sweep, affine coordinates, results out
    for (size_t j = 0; j < p.K; j++) {
        size_t lo, cnt;
        io_range(p, n, j, lo, cnt);
        HIP_TRY(hipEventRecord(d.iev[2][j], s));
        if (fused)
            hipLaunchKernelGGL(msmk::k_norm_down0_final<C>, dim3((uint32_t)(((cnt + msmk::NORM_K - 1) / msmk::NORM_K + 255) / 256)), dim3(256), 0, s,
                               raw_in + lo * msmk::Geo<C>::RAW_JAC, (const uint32_t*)at(d.nv_pref, 0, lo), (const uint32_t*)at(d.nv_inv, 1, lo / msmk::NORM_K),
                               (uint32_t)cnt, raw_out + lo * msmk::Geo<C>::RAW_AFF);
        else
            hipLaunchKernelGGL(msmk::k_norm_final<C>, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, raw_in + lo * msmk::Geo<C>::RAW_JAC,
                               (const uint32_t*)at(d.nv_inv, 0, lo), (uint32_t)cnt, raw_out + lo * msmk::Geo<C>::RAW_AFF);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(d.iev[3][j], s));
    }
    for (size_t j = 0; j < p.K; j++) io_drain_chunk(d, p, n, j, d.iev[3][j], outs, 1);   // after every kernel is enqueued (io_chunks.hpp io_stream_pass)
    io_finish(d);
    if (publish_profile) {
        mi_profile pr{};
        pr.n = n;
        pr.h2d_ms = h_in ? ev_ms(d.cev[0], d.cev[p.K]) : 0.0;
        pr.accumulate_ms = ev_ms(d.ev[0], d.ev[1]);          // all normalize kernels incl. the host inversion round trip ...
        for (size_t j = 0; j < p.K; j++) pr.accumulate_ms += ev_ms(d.iev[0][j], d.iev[1][j]) + ev_ms(d.iev[2][j], d.iev[3][j]);   // ... per chunk, without the waits for its input
        set_prof(ctx, pr);
    }
}

template <class C, class In, class Out>
int normalize_impl(mi_ctx* ctx, const In* in, bool on_device, size_t n, Out* out) {
    if (!ctx || (n && (!in || !out))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    if (on_device && ctx->devs.size() != 1) return fail(ctx, MI_E_INVALID, "the *_device entry points need a single-device context");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        auto t0 = std::chrono::steady_clock::now();
        if (on_device) {
            if (device_of_ptr(in, "d_in", 4) != d.dev || device_of_ptr(out, "d_out", 4) != d.dev)
                return fail(ctx, MI_E_INVALID, "device buffers must live on the context's device");
            normalize_run<C>(ctx, d, nullptr, in, n, nullptr, out, true);
        } else {
            normalize_run<C>(ctx, d, in, nullptr, n, out, nullptr, true);
        }
        std::lock_guard<std::mutex> g(ctx->info_mu);
        ctx->prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();   // the call as a C caller sees it
        return MI_OK;
    });
}

}  // namespace mi
