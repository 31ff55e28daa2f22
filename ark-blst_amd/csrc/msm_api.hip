// C-ABI host driver of the MI355X MSM backend (see include/arkblst_amd.h for the contract and the
// reference interfaces each entry point replaces: /root/reference/src/gpu.rs:101-241, src/g1.rs:602-632,
// src/g2.rs:582-612).
//
// One mi_ctx owns, per device: a HIP stream, a resident base set in device form, and reusable scratch
// (histogram / offsets / sorted indices / buckets / chunk sums) sized for the largest call seen so far —
// the reference rebuilds its program and re-allocates every buffer on every call (src/gpu.rs:148-156,235).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arkblst_amd.h"
#include "host_curve.hpp"
#include "host_pool.hpp"
#include "msm_kernels.cuh"
#include "pairing_kernels.cuh"

namespace {

using hostec::G1;
using hostec::G2;

struct HipFail {
    std::string msg;
};
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            char _b[512];                                                                                 \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw HipFail{_b};                                                                            \
        }                                                                                                 \
    } while (0)

struct Plan {
    uint32_t c, nwin, nb, logL, chunks_per_win, logT, lo_bits;
    uint64_t nbuckets, nchunks;
};

// Window size c by a time model of the pipeline on one MI355X (microseconds; constants measured on G1):
//   accumulate  max(throughput: N W mixed additions at 7.1e9 /s,  latency: one lane walks an item of T entries at 11 us each)
//   merge       one ~60 us launch per binary-tree level when the short top window overfills its buckets
//   reduce      a latency chain of 2L + 13 + log2 L complete additions of 17 us per wave, one wave per SIMD per round
//   sort        N W entries at 4.1e10 /s;  schedule ~ buckets / 1e4;  host tail ~ 100 + 0.16 per chunk pair
// Plays the role of calc_window_size (/root/reference/src/gpu.rs:218-223).  G2 costs ~3x more in accumulate and reduce
// alike, which leaves the optimum where it is.
Plan make_plan(size_t n, unsigned forced_c) {
    Plan best{};
    double best_cost = 1e300;
    uint32_t idx_bits = 1;
    while ((1ull << idx_bits) < n) idx_bits++;
    for (unsigned c = 7; c <= 22; c++) {
        if (forced_c && c != forced_c) continue;
        // sort geometry: lo bits share a 32-bit entry with the point index and the sign; the coarse bins of one
        // window must fit the LDS counter array
        uint32_t lo_bits = std::min<uint32_t>(std::min<uint32_t>(8, c - 1), 31 - idx_bits);
        if (((1u << (c - 1)) >> lo_bits) > msmk::SORT_MAX_COUNTERS) continue;
        Plan p{};
        p.c = c;
        p.lo_bits = lo_bits;
        p.nwin = (256 + c - 1) / c;
        if ((uint64_t)n * p.nwin >= (1ull << 32)) continue;   // entry offsets are 32-bit
        p.nb = 1u << (c - 1);
        p.nbuckets = (uint64_t)p.nb * p.nwin;
        // reduce geometry: the smallest L = 2^logL buckets per lane that still gives every wave its own SIMD (<= 1024
        // chunks of 64 L buckets), capped at L = 64: at 2^20 points (c = 16) that is L = 8
        p.logL = 0;
        while (p.logL < 6 && p.logL < c - 7 && (p.nbuckets >> (6 + p.logL)) > 1024) p.logL++;
        p.chunks_per_win = p.nb >> (6 + p.logL);
        p.nchunks = (uint64_t)p.chunks_per_win * p.nwin;
        // work-item size: twice the mean bucket load (uniform scalars then never split), at least 32 entries
        double mean = (double)n / p.nb;
        p.logT = 5;
        while ((double)(1u << p.logT) < 2.0 * mean && p.logT < 20) p.logT++;
        const double T = (double)(1u << p.logT), entries = (double)n * p.nwin;
        // The top window holds only 255 - c (nwin - 1) significant bits: its n entries share 2^top_bits buckets, and
        // buckets beyond T entries are split and merged by a binary tree
        int top_bits = std::max(0, std::min<int>(255 - (int)c * ((int)p.nwin - 1), (int)c - 1));
        double per_bucket = (double)n / (double)(1u << top_bits);
        int merge_levels = 0;
        for (double x = per_bucket; x > T; x *= 0.5) merge_levels++;
        const double item_len = std::min(T, std::max(mean, std::min(per_bucket, T)));   // entries a lane walks serially
        const double rounds = (double)((p.nchunks + 1023) / 1024);
        double cost = std::max(entries / 7100.0, item_len * 11.0) + merge_levels * 60.0 +
                      rounds * (2.0 * (1u << p.logL) + 13.0 + p.logL) * 17.0 + entries / 41000.0 + (double)p.nbuckets / 1e4 + 100.0 +
                      0.16 * (double)p.nchunks;
        if (cost < best_cost) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// resident bases (device form) of one device, per group: [0] = G1, [1] = G2.  Shared by the context's lanes.
struct Resident {
    DevBuf buf, flags;   // device-form points; one byte per point: 1 = point at infinity
    size_t n = 0;    // points resident on this device
    size_t lo = 0;   // global index of the first resident point
};

// Per-device state of ONE LANE of a context: stream, events and scratch.  A context has two lanes per device so that two
// host threads (arkworks calls the trait method from rayon workers) overlap: one call's sort / reduce / host tail runs
// under the other's accumulate kernel (measured +19 % at 2^20).  The resident bases are shared.
struct DevState {
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[10] = {};
    Resident* res = nullptr;   // -> mi_ctx::residents[device][2]
    // scratch
    DevBuf raw, call_bases, call_flags, scalars, hist, offsets, woff, meta, sched, sorted, partial, order, item_bucket, pairs;
    DevBuf tilecnt, bin_tot, bin_base, coarse, seg_cnt, seg_base, segcnt, merge_list;
    DevBuf pr_p, pr_q, pr_lvl[2], pr_raw, pr_lines;   // pairing: inputs, tree levels, top values, line coefficients
    void* h_pairs = nullptr;
    size_t h_pairs_cap = 0;
    mi_profile prof{};
};

}  // namespace

constexpr int NLANES = 2;

struct mi_ctx {
    std::vector<DevState> devs;                              // lane 0 (also used by every non-MSM entry point)
    std::vector<DevState> devs_b;                            // lane 1
    std::vector<std::array<Resident, 2>> residents;          // per device
    std::unique_ptr<hostpool::Pool> pool;
    std::mutex pool_mu;                                      // the pool runs one parallel_for at a time
    std::mutex lane_mu;                                      // lane bookkeeping
    std::condition_variable lane_cv;
    bool lane_busy[NLANES] = {false, false};
    mutable std::mutex info_mu;                              // prof / err
    unsigned forced_c = 0;
    mi_profile prof{};
    std::string err;
};

namespace {

// An MSM call takes ONE free lane (two calls run concurrently); everything that touches the resident bases or the
// shared settings takes BOTH (exclusive).
struct LaneLock {
    mi_ctx* c;
    int lane;   // 0 / 1, or -1 = both
    LaneLock(mi_ctx* ctx, bool exclusive) : c(ctx), lane(-1) {
        std::unique_lock<std::mutex> lk(c->lane_mu);
        if (exclusive) {
            c->lane_cv.wait(lk, [&] { return !c->lane_busy[0] && !c->lane_busy[1]; });
            c->lane_busy[0] = c->lane_busy[1] = true;
        } else {
            c->lane_cv.wait(lk, [&] { return !c->lane_busy[0] || !c->lane_busy[1]; });
            lane = c->lane_busy[0] ? 1 : 0;
            c->lane_busy[lane] = true;
        }
    }
    ~LaneLock() {
        {
            std::lock_guard<std::mutex> lk(c->lane_mu);
            if (lane < 0) c->lane_busy[0] = c->lane_busy[1] = false;
            else c->lane_busy[lane] = false;
        }
        c->lane_cv.notify_all();
    }
    LaneLock(const LaneLock&) = delete;
    LaneLock& operator=(const LaneLock&) = delete;
    std::vector<DevState>& devs() { return lane == 1 ? c->devs_b : c->devs; }
};

void set_prof(mi_ctx* ctx, const mi_profile& p) {
    std::lock_guard<std::mutex> lk(ctx->info_mu);
    ctx->prof = p;
}

void ensure_host(DevState& d, size_t bytes) {
    if (bytes <= d.h_pairs_cap) return;
    if (d.h_pairs) HIP_TRY(hipHostFree(d.h_pairs));
    d.h_pairs = nullptr;
    HIP_TRY(hipHostMalloc(&d.h_pairs, bytes, hipHostMallocDefault));
    d.h_pairs_cap = bytes;
}

double ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    return ms;
}

// host-side view of a curve: the reference's raw sizes and the CPU Jacobian type used for the O(windows) tail
template <class C> struct HostCurve;
template <> struct HostCurve<msmk::G1C> {
    using J = hostec::G1;
    static constexpr int IDX = 0;
};
template <> struct HostCurve<msmk::G2C> {
    using J = hostec::G2;
    static constexpr int IDX = 1;
};
template <class C> constexpr size_t aff_bytes() { return (size_t)msmk::Geo<C>::RAW_AFF * 4; }
template <class C> constexpr size_t jac_bytes() { return (size_t)msmk::Geo<C>::RAW_JAC * 4; }

// bases raw (host or device) -> device form in `dst`
template <class C>
void ingest(DevState& d, const void* bases, bool bases_on_device, size_t n, DevBuf& dst, DevBuf& flags) {
    dst.ensure(n * msmk::Geo<C>::PT_WORDS * 4);
    flags.ensure(n);
    const void* src = bases;
    if (!bases_on_device) {
        d.raw.ensure(n * aff_bytes<C>());
        HIP_TRY(hipMemcpyAsync(d.raw.p, bases, n * aff_bytes<C>(), hipMemcpyHostToDevice, d.stream));
        src = d.raw.p;
    }
    uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(msmk::k_ingest<C>, dim3(grid), dim3(256), 0, d.stream, (const uint32_t*)src, (uint32_t*)dst.p, (uint8_t*)flags.p, (uint32_t)n);
    HIP_TRY(hipGetLastError());
}

// Host tail: combine chunk sums per window and Horner-fold the windows (cf. /root/reference/src/gpu.rs:193-209).
template <class J>
J host_fold(mi_ctx* ctx, const J* pairs, const Plan& pl) {
    // window sum = sum_j T_j + (64 L) * sum_j j * S_j over its chunks j.  The chunk range of a window is cut into
    // `parts` sub-ranges so that windows x parts work units keep the pool busy: for a sub-range [a, b)
    //   sum_{j in [a,b)} j S_j = W_ab + a * S_ab,  W_ab = sum (j - a) S_j (running sums),  S_ab = sum S_j.
    const uint32_t cpw = pl.chunks_per_win;
    uint32_t parts = 1;
    while (parts < 64 && cpw / (parts * 2) >= 16 && pl.nwin * parts < 192) parts *= 2;
    const uint32_t seg = (cpw + parts - 1) / parts;
    struct Part { J w, s, t; };
    std::vector<Part> part((size_t)pl.nwin * parts);
    auto do_part = [&](unsigned u) {
        uint32_t w = u / parts, q = u % parts;
        const J* p = pairs + (size_t)w * cpw * 2;
        uint32_t a = q * seg, b = std::min(cpw, a + seg);
        J run = J::inf(), acc = J::inf(), tsum = J::inf();
        for (int j = (int)b - 1; j >= (int)a; j--) {
            tsum = tsum.add(p[2 * j + 1]);
            if (j > (int)a) {
                run = run.add(p[2 * j]);
                acc = acc.add(run);
            }
        }
        J sab = a < b ? run.add(p[2 * a]) : J::inf();
        part[u] = Part{acc, sab, tsum};
    };
    // seg is a power of two whenever parts > 1 (chunks_per_win is): a * S = (q * seg) * S by doublings
    unsigned log_seg = 0;
    while ((1u << log_seg) < seg) log_seg++;
    std::vector<J> win(pl.nwin);
    auto do_window = [&](unsigned w) {
        // sum_q [W_q + q * seg * S_q] = sum_q W_q + seg * sum_q q S_q
        J wsum = J::inf(), tsum = J::inf(), run = J::inf(), qs = J::inf();
        for (int q = (int)parts - 1; q >= 0; q--) {
            const Part& pt = part[(size_t)w * parts + q];
            wsum = wsum.add(pt.w);
            tsum = tsum.add(pt.t);
            if (q >= 1) {
                run = run.add(pt.s);
                qs = qs.add(run);
            }
        }
        J x = parts > 1 ? wsum.add(qs.dbl_n(log_seg)) : wsum;
        win[w] = x.dbl_n(6 + pl.logL).add(tsum);
    };
    size_t work = (size_t)pl.nwin * cpw;
    const uint32_t units = pl.nwin * parts;
    const bool use_pool = work >= 128 && ctx->pool && ctx->devs.size() == 1;
    if (use_pool) {
        // Pipelined with the Horner fold: the workers take the units of the TOP window first; whoever finishes the last part
        // of a window combines it and flags it; this thread folds the windows top-down as they become ready, so the 240
        // serial doublings run beside the chunk sums instead of after them.
        std::lock_guard<std::mutex> pool_lock(ctx->pool_mu);   // the other lane's fold (~0.2 ms) may be running
        std::vector<std::atomic<int>> left(pl.nwin), ready(pl.nwin);
        for (uint32_t w = 0; w < pl.nwin; w++) { left[w].store((int)parts); ready[w].store(0); }
        std::function<void(unsigned)> job = [&](unsigned k) {
            uint32_t w = pl.nwin - 1 - k / parts, q = k % parts;
            do_part(w * parts + q);
            if (left[w].fetch_sub(1, std::memory_order_acq_rel) == 1) {
                do_window(w);
                ready[w].store(1, std::memory_order_release);
            }
        };
        if (ctx->pool->start(units, job)) {
            J r = J::inf();
            for (int w = (int)pl.nwin - 1; w >= 0; w--) {
                while (!ready[w].load(std::memory_order_acquire)) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                r = r.dbl_n(pl.c).add(win[w]);
            }
            ctx->pool->finish(units);
            return r;
        }
        for (uint32_t k = 0; k < units; k++) job(k);   // no workers
    } else {
        // multi-device context: every device thread folds its own pairs; it may take a few helper threads of its own (the
        // shared pool runs one loop at a time and would serialise the devices)
        const unsigned helpers = work >= 512 && ctx->devs.size() > 1
                                     ? std::min<unsigned>(4, std::max<unsigned>(1, std::thread::hardware_concurrency() / (unsigned)ctx->devs.size()))
                                     : 1;
        if (helpers > 1) {
            std::atomic<uint32_t> next{0};
            auto loop = [&]() {
                for (uint32_t u = next.fetch_add(1); u < units; u = next.fetch_add(1)) do_part(u);
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < helpers; t++) th.emplace_back(loop);
            loop();
            for (auto& t : th) t.join();
        } else {
            for (uint32_t u = 0; u < units; u++) do_part(u);
        }
        for (uint32_t w = 0; w < pl.nwin; w++) do_window(w);
    }
    J r = J::inf();
    for (int w = (int)pl.nwin - 1; w >= 0; w--) r = r.dbl_n(pl.c).add(win[w]);  // Horner: the only serial part
    return r;
}

// k_coarse is compiled per window size (static digit extraction): dispatch on c = 7..22
template <bool SCATTER, int CB = 7>
void launch_coarse(uint32_t c, dim3 grid, dim3 block, hipStream_t s, const uint32_t* scalars, const uint8_t* flags, const msmk::SortGeom& g,
                   uint32_t* tilecnt, const uint32_t* bin_base, uint32_t* coarse) {
    if constexpr (CB > 22) {
        throw HipFail{"window_bits out of range"};
    } else {
        if (c == CB)
            hipLaunchKernelGGL((msmk::k_coarse<SCATTER, CB>), grid, block, 0, s, scalars, flags, g, tilecnt, bin_base, coarse);
        else
            launch_coarse<SCATTER, CB + 1>(c, grid, block, s, scalars, flags, g, tilecnt, bin_base, coarse);
    }
}

// The pipeline on one device.  d_bases: device-form points for indices [0, n); d_scalars: n x 32 B on device.
template <class C>
typename HostCurve<C>::J run_msm(mi_ctx* ctx, DevState& d, const uint32_t* d_bases, const uint8_t* d_flags, const uint32_t* d_scalars,
                                 size_t n, unsigned fmt, int ev0) {
    using J = typename HostCurve<C>::J;
    Plan pl = make_plan(n, ctx->forced_c);
    if (pl.c == 0) throw HipFail{"window_bits not usable for this n (sort geometry)"};
    d.prof.window_bits = pl.c;
    d.prof.num_windows = pl.nwin;
    d.prof.n = n;
    const size_t pair_bytes = 2 * jac_bytes<C>();
    d.hist.ensure(pl.nbuckets * 4);
    d.offsets.ensure((pl.nbuckets + 1) * 4);
    d.woff.ensure((pl.nbuckets + 1) * 4);
    d.meta.ensure(16);
    d.sorted.ensure((size_t)n * pl.nwin * 4);
    d.pairs.ensure(pl.nchunks * pair_bytes);
    ensure_host(d, pl.nchunks * pair_bytes + 16);

    hipStream_t s = d.stream;
    HIP_TRY(hipEventRecord(d.ev[ev0], s));
    // ---- two-level LDS-staged bucket sort (geometry in msmk::SortGeom)
    msmk::SortGeom g{};
    g.n = (uint32_t)n; g.fmt = fmt; g.c = pl.c; g.nwin = pl.nwin;
    g.lo_bits = pl.lo_bits;
    g.H = pl.nb >> g.lo_bits;
    // A tile contributes tile_pts / H entries to each coarse bin of a window, written as one contiguous run: keep
    // runs >= 64 entries (256 B) or the 4-byte scatter is write-amplified (9.5 ms at n = 2^24 with 16-entry runs).
    size_t want = std::max<size_t>(std::max<size_t>(4096, n / 512), (size_t)64 * g.H);
    g.tile_pts = (uint32_t)((std::min(want, n) + 1023) / 1024 * 1024);
    g.tiles = (uint32_t)((n + g.tile_pts - 1) / g.tile_pts);
    const uint32_t coarse_block = g.tile_pts >= 16384 ? 1024 : 256;
    g.wgroup = std::max<uint32_t>(1, std::min<uint32_t>(pl.nwin, msmk::SORT_MAX_COUNTERS / g.H));
    g.ngroups = (pl.nwin + g.wgroup - 1) / g.wgroup;
    g.nbins = pl.nwin * g.H;
    d.tilecnt.ensure((size_t)g.tiles * g.nbins * 4);
    d.bin_tot.ensure((size_t)g.nbins * 4);
    d.bin_base.ensure((size_t)(g.nbins + 1) * 4);
    d.coarse.ensure((size_t)n * pl.nwin * 4);
    launch_coarse<false>(pl.c, dim3(g.tiles, g.ngroups), dim3(coarse_block), s, d_scalars, d_flags, g, (uint32_t*)d.tilecnt.p,
                         (const uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(msmk::k_colscan, dim3((g.nbins + 255) / 256), dim3(256), 0, s, (uint32_t*)d.tilecnt.p, g.nbins, g.tiles,
                       (uint32_t*)d.bin_tot.p);
    hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)d.bin_tot.p, g.nbins, (uint32_t*)d.bin_base.p);
    launch_coarse<true>(pl.c, dim3(g.tiles, g.ngroups), dim3(coarse_block), s, d_scalars, d_flags, g, (uint32_t*)d.tilecnt.p,
                        (const uint32_t*)d.bin_base.p, (uint32_t*)d.coarse.p);
    HIP_TRY(hipEventRecord(d.ev[ev0 + 1], s));
    // fine sort over bin segments (upper bound on the segment count: one per bin plus one per FINE_SEG entries)
    uint32_t segs_cap = g.nbins + (uint32_t)(((size_t)n * pl.nwin) / msmk::FINE_SEG) + 1;
    d.seg_cnt.ensure((size_t)g.nbins * 4);
    d.seg_base.ensure((size_t)(g.nbins + 1) * 4);
    d.segcnt.ensure((size_t)segs_cap * (1u << g.lo_bits) * 4);
    hipLaunchKernelGGL(msmk::k_seg_count, dim3((g.nbins + 255) / 256), dim3(256), 0, s, (const uint32_t*)d.bin_base.p, g.nbins,
                       (uint32_t*)d.seg_cnt.p);
    hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)d.seg_cnt.p, g.nbins, (uint32_t*)d.seg_base.p);
    hipLaunchKernelGGL(msmk::k_fine_count, dim3(segs_cap), dim3(256), 0, s, (const uint32_t*)d.coarse.p, (const uint32_t*)d.bin_base.p,
                       (const uint32_t*)d.seg_base.p, g, (uint32_t*)d.segcnt.p);
    hipLaunchKernelGGL(msmk::k_fine_scan, dim3(g.nbins), dim3(256), 0, s, (const uint32_t*)d.seg_base.p, g, (uint32_t*)d.segcnt.p,
                       (uint32_t*)d.hist.p);
    hipLaunchKernelGGL(msmk::k_fine_scatter, dim3(segs_cap), dim3(256), 0, s, (const uint32_t*)d.coarse.p, (const uint32_t*)d.bin_base.p,
                       (const uint32_t*)d.seg_base.p, g, (const uint32_t*)d.segcnt.p, (uint32_t*)d.sorted.p);
    HIP_TRY(hipEventRecord(d.ev[ev0 + 2], s));
    // ---- schedule: <= 256 blocks of 1024 lanes, each lane owning per_blk/1024 consecutive buckets
    uint32_t per_blk = 4096;
    while ((pl.nbuckets + per_blk - 1) / per_blk > 256) per_blk <<= 1;
    uint32_t nblk = (uint32_t)((pl.nbuckets + per_blk - 1) / per_blk);
    size_t items_cap = pl.nbuckets + (((size_t)n * pl.nwin) >> pl.logT) + 1;  // one per bucket plus one per T entries
    d.sched.ensure((size_t)(3 + msmk::SCHED_CLASSES) * nblk * 4);
    d.order.ensure(items_cap * 4);
    d.item_bucket.ensure(items_cap * 4);
    d.merge_list.ensure(items_cap * 4);
    uint32_t* blk_e = (uint32_t*)d.sched.p;
    uint32_t* blk_i = blk_e + nblk;
    uint32_t* blk_max = blk_i + nblk;
    uint32_t* blk_cls = blk_max + nblk;
    hipLaunchKernelGGL(msmk::k_sched1, dim3(nblk), dim3(1024), 0, s, (const uint32_t*)d.hist.p, (uint32_t)pl.nbuckets, per_blk, pl.logT,
                       nblk, blk_e, blk_i, blk_cls, blk_max);
    hipLaunchKernelGGL(msmk::k_sched2, dim3(1), dim3(1024), 0, s, nblk, blk_e, blk_i, blk_cls, (const uint32_t*)blk_max,
                       (uint32_t*)d.meta.p);
    hipLaunchKernelGGL(msmk::k_sched3, dim3(nblk), dim3(1024), 0, s, (const uint32_t*)d.hist.p, (uint32_t)pl.nbuckets, per_blk, pl.logT,
                       nblk, (const uint32_t*)blk_e, (const uint32_t*)blk_i, (const uint32_t*)blk_cls, (uint32_t*)d.offsets.p,
                       (uint32_t*)d.woff.p, (uint32_t*)d.order.p, (uint32_t*)d.item_bucket.p, (uint32_t*)d.merge_list.p,
                       (uint32_t*)d.meta.p);
    // the item count sizes the next launches: one small read-back (the only mid-pipeline sync)
    uint32_t meta[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(meta, d.meta.p, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(d.ev[ev0 + 3], s));
    HIP_TRY(hipStreamSynchronize(s));
    uint32_t nitems = meta[0], max_items = meta[1];
    d.partial.ensure((size_t)nitems * msmk::Geo<C>::BK_WORDS * 4);
    uint32_t grid_items = (nitems + 255) / 256;
    hipLaunchKernelGGL(msmk::k_accumulate<C>, dim3(grid_items), dim3(256), 0, s, d_bases, (const uint32_t*)d.sorted.p,
                       (const uint32_t*)d.offsets.p, (const uint32_t*)d.woff.p, (const uint32_t*)d.order.p,
                       (const uint32_t*)d.item_bucket.p, nitems, pl.logT, (uint32_t*)d.partial.p);
    uint32_t nlist = meta[3];
    for (uint32_t dd = 1; dd < max_items && nlist; dd <<= 1)
        hipLaunchKernelGGL(msmk::k_merge<C>, dim3((nlist + 255) / 256), dim3(256), 0, s, (uint32_t*)d.partial.p,
                           (const uint32_t*)d.item_bucket.p, (const uint32_t*)d.woff.p, (const uint32_t*)d.merge_list.p, nlist, dd);
    HIP_TRY(hipEventRecord(d.ev[ev0 + 4], s));
    static const bool g2_one_lane = getenv("MI_G2_REDUCE_ONE_LANE") != nullptr;   // A/B hook: the generic kernel for G2
    if (std::is_same<C, msmk::G2C>::value && !g2_one_lane) {
        // two lanes per logical lane: a chunk of 64 << logL buckets = 32 logical lanes x 2^(logL+1) buckets
        hipLaunchKernelGGL(msmk::k_reduce_g2_coop, dim3((uint32_t)pl.nchunks), dim3(64), 0, s, (const uint32_t*)d.partial.p,
                           (const uint32_t*)d.woff.p, (uint32_t*)d.pairs.p, pl.logL + 1);
    } else {
        hipLaunchKernelGGL(msmk::k_reduce<C>, dim3((uint32_t)pl.nchunks), dim3(64), 0, s, (const uint32_t*)d.partial.p,
                           (const uint32_t*)d.woff.p, (uint32_t*)d.pairs.p, pl.logL);
    }
    HIP_TRY(hipEventRecord(d.ev[ev0 + 5], s));
    HIP_TRY(hipMemcpyAsync(d.h_pairs, d.pairs.p, pl.nchunks * pair_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(d.ev[ev0 + 6], s));
    HIP_TRY(hipEventSynchronize(d.ev[ev0 + 4]));  // accumulate done: ~0.6 ms of reduce left
    if (ctx->pool && ctx->devs.size() == 1) ctx->pool->prewake();
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());

    d.prof.digits_ms = ev_ms(d.ev[ev0], d.ev[ev0 + 1]);      // digits + coarse partition (4 kernels)
    d.prof.scatter_ms = ev_ms(d.ev[ev0 + 1], d.ev[ev0 + 2]);  // fine sort in LDS
    d.prof.scan_ms = ev_ms(d.ev[ev0 + 2], d.ev[ev0 + 3]);     // schedule (3 kernels)
    d.prof.accumulate_ms = ev_ms(d.ev[ev0 + 3], d.ev[ev0 + 4]);
    d.prof.reduce_ms = ev_ms(d.ev[ev0 + 4], d.ev[ev0 + 5]);
    d.prof.d2h_ms = ev_ms(d.ev[ev0 + 5], d.ev[ev0 + 6]);
    d.prof.accumulate_adds = meta[2];
    d.prof.work_items = nitems;
    d.prof.max_items_per_bucket = max_items;

    auto t0 = std::chrono::steady_clock::now();
    J r = host_fold<J>(ctx, reinterpret_cast<const J*>(d.h_pairs), pl);
    d.prof.host_fold_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return r;
}

// One device's share of an MSM call. bases: host raw pointer for this shard or nullptr (= resident).
template <class C>
typename HostCurve<C>::J device_msm(mi_ctx* ctx, DevState& d, const uint8_t* bases, const uint8_t* scalars, bool scalars_on_device,
                                    size_t n, unsigned fmt) {
    using J = typename HostCurve<C>::J;
    HIP_TRY(hipSetDevice(d.dev));
    d.prof = mi_profile{};
    auto t0 = std::chrono::steady_clock::now();
    if (n == 0) return J::inf();
    hipStream_t s = d.stream;
    HIP_TRY(hipEventRecord(d.ev[0], s));
    const uint32_t* d_scalars;
    if (scalars_on_device) {
        d_scalars = reinterpret_cast<const uint32_t*>(scalars);
    } else {
        d.scalars.ensure(n * 32);
        HIP_TRY(hipMemcpyAsync(d.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
        d_scalars = reinterpret_cast<const uint32_t*>(d.scalars.p);
    }
    const uint32_t* d_bases;
    const uint8_t* d_flags;
    if (bases) {
        d.raw.ensure(n * aff_bytes<C>());
        HIP_TRY(hipMemcpyAsync(d.raw.p, bases, n * aff_bytes<C>(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(d.ev[1], s));
        ingest<C>(d, d.raw.p, true, n, d.call_bases, d.call_flags);
        d_bases = reinterpret_cast<const uint32_t*>(d.call_bases.p);
        d_flags = reinterpret_cast<const uint8_t*>(d.call_flags.p);
    } else {
        HIP_TRY(hipEventRecord(d.ev[1], s));
        d_bases = reinterpret_cast<const uint32_t*>(d.res[HostCurve<C>::IDX].buf.p);
        d_flags = reinterpret_cast<const uint8_t*>(d.res[HostCurve<C>::IDX].flags.p);
    }
    J r = run_msm<C>(ctx, d, d_bases, d_flags, d_scalars, n, fmt, 2);
    d.prof.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
    d.prof.ingest_ms = ev_ms(d.ev[1], d.ev[2]);
    d.prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return r;
}

// normalize_batch on device 0 of the context (host pointers in the reference's forms)
template <class C>
void normalize_batch_dev(mi_ctx* ctx, DevState& d, const void* in, size_t n, void* out) {
    using J = typename HostCurve<C>::J;
    using FE = decltype(J::inf().x);
    constexpr size_t SLOTB = (size_t)msmk::Geo<C>::SLOT * 4;
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    // level sizes: n, ceil(n/K), ... until <= 64
    std::vector<size_t> sz{n};
    while (sz.back() > 64) sz.push_back((sz.back() + msmk::NORM_K - 1) / msmk::NORM_K);
    size_t total = 0;
    for (size_t v : sz) total += v;
    DevBuf raw_in, raw_out, vals, pref, inv, top_raw;
    raw_in.ensure(n * jac_bytes<C>());
    raw_out.ensure(n * aff_bytes<C>());
    vals.ensure(total * SLOTB);
    pref.ensure(total * SLOTB);
    inv.ensure(total * SLOTB);
    top_raw.ensure(64 * sizeof(FE));
    std::vector<size_t> off(sz.size());
    for (size_t l = 0, acc = 0; l < sz.size(); acc += sz[l], l++) off[l] = acc;
    auto at = [&](DevBuf& b, size_t l) { return (uint32_t*)((char*)b.p + off[l] * SLOTB); };
    try {
        HIP_TRY(hipEventRecord(d.ev[0], s));
        HIP_TRY(hipMemcpyAsync(raw_in.p, in, n * jac_bytes<C>(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(d.ev[1], s));
        hipLaunchKernelGGL(msmk::k_norm_load<C>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)raw_in.p, (uint32_t)n,
                           at(vals, 0));
        for (size_t l = 0; l + 1 < sz.size(); l++) {
            uint32_t groups = (uint32_t)sz[l + 1];
            hipLaunchKernelGGL(msmk::k_norm_up<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(vals, l), (uint32_t)sz[l],
                               at(pref, l), at(vals, l + 1));
        }
        // top level on the host: Montgomery's trick over <= 64 values, one inversion
        size_t top = sz.size() - 1, m = sz[top];
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
        for (size_t l = top; l-- > 0;) {
            uint32_t groups = (uint32_t)sz[l + 1];
            hipLaunchKernelGGL(msmk::k_norm_down<C>, dim3((groups + 255) / 256), dim3(256), 0, s, (const uint32_t*)at(vals, l),
                               (const uint32_t*)at(pref, l), (const uint32_t*)at(inv, l + 1), (uint32_t)sz[l], at(inv, l));
        }
        hipLaunchKernelGGL(msmk::k_norm_final<C>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)raw_in.p,
                           (const uint32_t*)at(inv, 0), (uint32_t)n, (uint32_t*)raw_out.p);
        HIP_TRY(hipEventRecord(d.ev[2], s));
        HIP_TRY(hipMemcpyAsync(out, raw_out.p, n * aff_bytes<C>(), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        mi_profile pr{};
        pr.n = n;
        pr.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
        pr.accumulate_ms = ev_ms(d.ev[1], d.ev[2]);   // all normalize kernels incl. the host inversion round trip
        set_prof(ctx, pr);
    } catch (...) {
        for (DevBuf* b : {&raw_in, &raw_out, &vals, &pref, &inv, &top_raw}) b->release();
        throw;
    }
    for (DevBuf* b : {&raw_in, &raw_out, &vals, &pref, &inv, &top_raw}) b->release();
}

int fail(mi_ctx* ctx, int code, const std::string& msg) {
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->info_mu);
        ctx->err = msg;
    }
    return code;
}

template <class Fn>
int guarded(mi_ctx* ctx, Fn fn) {
    try {
        return fn();
    } catch (const HipFail& e) {
        bool oom = e.msg.find("out of memory") != std::string::npos;
        return fail(ctx, oom ? MI_E_NOMEM : MI_E_HIP, e.msg);
    } catch (const std::bad_alloc&) {
        return fail(ctx, MI_E_NOMEM, "host allocation failed");
    } catch (...) {
        return fail(ctx, MI_E_HIP, "unexpected exception");
    }
}

// contiguous shard [lo, hi) of n items for device k of g
void shard_range(size_t n, size_t g, size_t k, size_t& lo, size_t& hi) {
    size_t per = (n + g - 1) / g;
    lo = std::min(n, k * per);
    hi = std::min(n, lo + per);
}

template <class C>
int set_bases_impl(mi_ctx* ctx, const void* bases, size_t n) {
    if (!ctx || (n && !bases)) return fail(ctx, MI_E_INVALID, "invalid argument");
    if ((n + ctx->devs.size() - 1) / ctx->devs.size() > (1ull << 26)) return fail(ctx, MI_E_INVALID, "more than 2^26 points per device");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        size_t g = ctx->devs.size();
        for (size_t k = 0; k < g; k++) {
            DevState& d = ctx->devs[k];
            auto& res = d.res[HostCurve<C>::IDX];
            size_t lo, hi;
            shard_range(n, g, k, lo, hi);
            HIP_TRY(hipSetDevice(d.dev));
            res.lo = lo;
            res.n = hi - lo;
            if (hi > lo) {
                ingest<C>(d, (const uint8_t*)bases + lo * aff_bytes<C>(), false, hi - lo, res.buf, res.flags);
                HIP_TRY(hipStreamSynchronize(d.stream));
            }
        }
        return MI_OK;
    });
}

template <class C>
int msm_impl(mi_ctx* ctx, const void* bases_v, const uint8_t* scalars, bool scalars_on_device, size_t n, unsigned fmt, void* out) {
    using J = typename HostCurve<C>::J;
    const uint8_t* bases = static_cast<const uint8_t*>(bases_v);
    if (!ctx || !out || (n && !scalars) || fmt > 1) return fail(ctx, MI_E_INVALID, "invalid argument");
    // 32-bit entry offsets: n * windows must stay below 2^32 on every device (2^26 points leave room for c >= 8)
    if ((n + ctx->devs.size() - 1) / ctx->devs.size() > (1ull << 26))
        return fail(ctx, MI_E_INVALID, "more than 2^26 points per device in one call: split the MSM and add the results (mi_g1_sum)");
    LaneLock lane(ctx, false);
    std::vector<DevState>& devs = lane.devs();
    return guarded(ctx, [&]() -> int {
        size_t g = devs.size();
        if (scalars_on_device && g != 1) return fail(ctx, MI_E_INVALID, "device-resident scalars need a single-device context");
        std::vector<J> part(g, J::inf());
        std::vector<std::string> errs(g);
        // resident path: each device covers the overlap of [0, n) with its resident shard
        if (!bases) {
            size_t have = 0;
            for (auto& d : devs) have += d.res[HostCurve<C>::IDX].n;
            if (have == 0 && n) return fail(ctx, MI_E_NO_BASES, "no resident base set for this group");
            if (n > have) return fail(ctx, MI_E_INVALID, "n exceeds the resident base set");
        }
        auto work = [&](size_t k) {
            DevState& d = devs[k];
            try {
                size_t lo, hi;
                if (bases) {
                    shard_range(n, g, k, lo, hi);
                } else {
                    auto& res = d.res[HostCurve<C>::IDX];
                    lo = std::min(n, res.lo);
                    hi = std::min(n, res.lo + res.n);
                }
                const uint8_t* sc = scalars_on_device ? scalars : scalars + lo * 32;
                part[k] = device_msm<C>(ctx, d, bases ? bases + lo * aff_bytes<C>() : nullptr, sc, scalars_on_device, hi - lo, fmt);
            } catch (const HipFail& e) {
                errs[k] = e.msg;
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        if (g == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (size_t k = 0; k < g; k++) th.emplace_back(work, k);
            for (auto& t : th) t.join();
        }
        for (size_t k = 0; k < g; k++)
            if (!errs[k].empty()) return fail(ctx, MI_E_HIP, errs[k]);
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

// k MSMs over the resident base set, two in flight (one per lane): the calling thread and one helper pull jobs
template <class C, class Out>
int msm_batch_impl(mi_ctx* ctx, const uint8_t* const* scalars, size_t k, size_t n, unsigned fmt, Out* out) {
    if (!ctx || (k && (!scalars || !out))) return fail(ctx, MI_E_INVALID, "invalid argument");
    for (size_t j = 0; j < k; j++)
        if (n && !scalars[j]) return fail(ctx, MI_E_INVALID, "null scalar vector");
    std::atomic<size_t> next{0};
    std::atomic<int> first_err{MI_OK};
    auto worker = [&]() {
        for (;;) {
            size_t j = next.fetch_add(1);
            if (j >= k || first_err.load() != MI_OK) break;
            int rc = msm_impl<C>(ctx, nullptr, scalars[j], false, n, fmt, &out[j]);
            int ok = MI_OK;
            if (rc != MI_OK) first_err.compare_exchange_strong(ok, rc);
        }
    };
    if (k > 1) {
        std::thread helper(worker);
        worker();
        helper.join();
    } else {
        worker();
    }
    return first_err.load();
}

// shared driver of the (de)serialisation entry points: `unit` = compressed size in bytes (48 G1, 96 G2)
template <class KDe>
int deserialize_impl(mi_ctx* ctx, KDe kernel, size_t unit, const uint8_t* bytes, size_t n, int compressed, int validate, void* out,
                     uint8_t* status) {
    if (!ctx || (n && (!bytes || !out || !status))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        HIP_TRY(hipSetDevice(d.dev));
        size_t sz = compressed ? unit : 2 * unit, aff = 2 * unit;
        DevBuf din, dout, dst;
        try {
            din.ensure(n * sz); dout.ensure(n * aff); dst.ensure(n);
            HIP_TRY(hipEventRecord(d.ev[0], d.stream));
            HIP_TRY(hipMemcpyAsync(din.p, bytes, n * sz, hipMemcpyHostToDevice, d.stream));
            HIP_TRY(hipEventRecord(d.ev[1], d.stream));
            hipLaunchKernelGGL(kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, (const uint8_t*)din.p, (uint32_t)n,
                               compressed ? 1 : 0, validate ? 1 : 0, (uint32_t*)dout.p, (uint8_t*)dst.p);
            HIP_TRY(hipEventRecord(d.ev[2], d.stream));
            HIP_TRY(hipMemcpyAsync(out, dout.p, n * aff, hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipMemcpyAsync(status, dst.p, n, hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipStreamSynchronize(d.stream));
            HIP_TRY(hipGetLastError());
            mi_profile pr{};
            pr.n = n;
            pr.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
            pr.accumulate_ms = ev_ms(d.ev[1], d.ev[2]);
            set_prof(ctx, pr);
        } catch (...) {
            din.release(); dout.release(); dst.release();
            throw;
        }
        din.release(); dout.release(); dst.release();
        return MI_OK;
    });
}

template <class KSer>
int serialize_impl(mi_ctx* ctx, KSer kernel, size_t unit, const void* points, size_t n, int compressed, uint8_t* bytes) {
    if (!ctx || (n && (!bytes || !points))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        HIP_TRY(hipSetDevice(d.dev));
        size_t sz = compressed ? unit : 2 * unit, aff = 2 * unit;
        DevBuf din, dout;
        try {
            din.ensure(n * aff); dout.ensure(n * sz);
            HIP_TRY(hipMemcpyAsync(din.p, points, n * aff, hipMemcpyHostToDevice, d.stream));
            hipLaunchKernelGGL(kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, (const uint32_t*)din.p, (uint32_t)n,
                               compressed ? 1 : 0, (uint8_t*)dout.p);
            HIP_TRY(hipMemcpyAsync(bytes, dout.p, n * sz, hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipStreamSynchronize(d.stream));
            HIP_TRY(hipGetLastError());
        } catch (...) {
            din.release(); dout.release();
            throw;
        }
        din.release(); dout.release();
        return MI_OK;
    });
}


// ------------------------------------------------------------------------------------------------ pairing
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

// Miller loops of one shard on one device, multiplied down to <= 64 values on the GPU and to one on the host
HT::E12 device_miller(DevState& d, const mi_g1_affine* p, const mi_g2_affine* q, size_t n) {
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    DevBuf &dp = d.pr_p, &dq = d.pr_q, &raw = d.pr_raw, &dlines = d.pr_lines;   // kept across calls (no per-call hipMalloc)
    DevBuf* lvl = d.pr_lvl;
    HT::E12 acc = HT::one12();
    {
        dp.ensure(n * sizeof(mi_g1_affine));
        dq.ensure(n * sizeof(mi_g2_affine));
        size_t fp12_bytes = (size_t)msmk::FP12_WORDS * 4;
        lvl[0].ensure(n * fp12_bytes);
        lvl[1].ensure(((n + msmk::FP12_TREE_K - 1) / msmk::FP12_TREE_K) * fp12_bytes);
        raw.ensure(64 * sizeof(mi_fp12));
        HIP_TRY(hipEventRecord(d.ev[0], s));
        HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(mi_g1_affine), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dq.p, q, n * sizeof(mi_g2_affine), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(d.ev[1], s));
        static const bool single_lane = getenv("MI_PAIRING_SINGLE_LANE") != nullptr;   // the first kernel, kept for cross-checks
        uint32_t share = 1;
        size_t nvals = 0;   // Fp12 values the Miller kernels leave in lvl[0]
        if (single_lane) {
            hipLaunchKernelGGL(msmk::k_miller_loop, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t*)dp.p, (const uint32_t*)dq.p,
                               (uint32_t)n, (uint32_t*)lvl[0].p);
        } else {
            // line coefficients of a batch of pairs (26 KB per pair), then six lanes per accumulator fold them into f.
            // share = pairs per accumulator (one squaring per step for all of them): as many as still leave ~1600 waves
            static const size_t batch_cap = getenv("MI_PAIRING_BATCH") ? std::max(1, atoi(getenv("MI_PAIRING_BATCH"))) : (1u << 17);   // test hook
            const size_t batch = std::min<size_t>(n, batch_cap);
            share = (uint32_t)std::min<size_t>(8, std::max<size_t>(1, n >> 14));
            if (const char* e = getenv("MI_PAIRING_SHARE")) share = (uint32_t)std::max(1, atoi(e));   // test hook
            while (share > 1 && batch % share) share >>= 1;   // accumulators must not straddle line batches
            dlines.ensure(batch * msmk::MILLER_LINES * 3 * 32 * 4);
            for (size_t lo = 0; lo < n; lo += batch) {
                uint32_t mm = (uint32_t)std::min(batch, n - lo);
                uint32_t groups = (mm + share - 1) / share;
                hipLaunchKernelGGL(msmk::k_miller_lines2, dim3((2 * mm + 63) / 64), dim3(64), 0, s,
                                   (const uint32_t*)dp.p + lo * msmk::Geo<msmk::G1C>::RAW_AFF, (const uint32_t*)dq.p + lo * msmk::Geo<msmk::G2C>::RAW_AFF,
                                   mm, (uint32_t*)dlines.p);
                hipLaunchKernelGGL(msmk::k_miller_accumulate, dim3((groups + msmk::MILLER_GROUPS - 1) / msmk::MILLER_GROUPS), dim3(64), 0, s,
                                   (const uint32_t*)dlines.p, mm, share, (uint32_t*)lvl[0].p + nvals * msmk::FP12_WORDS);
                nvals += groups;
            }
        }
        HIP_TRY(hipEventRecord(d.ev[2], s));
        size_t m = single_lane ? n : nvals;
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
        std::vector<mi_fp12> top(m);
        HIP_TRY(hipMemcpyAsync(top.data(), raw.p, m * sizeof(mi_fp12), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipGetLastError());
        for (size_t k = 0; k < m; k++) acc = HT::mul12(acc, fp12_from_raw(&top[k]));
        d.prof = mi_profile{};
        d.prof.n = n;
        d.prof.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
        d.prof.accumulate_ms = ev_ms(d.ev[1], d.ev[2]);   // Miller loops
        d.prof.reduce_ms = ev_ms(d.ev[2], d.ev[3]);       // multiplication tree
    }
    return acc;
}

int miller_impl(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out, bool final_exp) {
    if (!ctx || !out || (n && (!p || !q))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        size_t g = ctx->devs.size();
        std::vector<HT::E12> part(g, HT::one12());
        std::vector<std::string> errs(g);
        auto work = [&](size_t k) {
            try {
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                if (hi > lo) part[k] = device_miller(ctx->devs[k], p + lo, q + lo, hi - lo);
            } catch (const HipFail& e) {
                errs[k] = e.msg;
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        if (g == 1 || n < 2 * g) {
            for (size_t k = 0; k < g; k++) work(k);
        } else {
            std::vector<std::thread> th;
            for (size_t k = 0; k < g; k++) th.emplace_back(work, k);
            for (auto& t : th) t.join();
        }
        for (size_t k = 0; k < g; k++)
            if (!errs[k].empty()) return fail(ctx, MI_E_HIP, errs[k]);
        HT::E12 f = part[0];
        for (size_t k = 1; k < g; k++) f = HT::mul12(f, part[k]);
        auto t1 = std::chrono::steady_clock::now();
        if (final_exp) f = HT::final_exp(f);
        fp12_to_raw(out, f);
        mi_profile pr = ctx->devs[0].prof;
        pr.n = n;
        pr.host_fold_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

}  // namespace

extern "C" {

int mi_msm_init(mi_ctx** out, const int* device_ids, int n_devices) {
    if (!out || n_devices < 0) return MI_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MI_E_NO_DEVICE;
    if (n_devices == 0) n_devices = device_ids ? 0 : count;
    if (n_devices <= 0 || (!device_ids && n_devices > count) || n_devices > 64) return MI_E_NO_DEVICE;
    mi_ctx* ctx = new (std::nothrow) mi_ctx();
    if (!ctx) return MI_E_NOMEM;
    int rc = guarded(ctx, [&]() -> int {
        ctx->devs.resize(n_devices);
        ctx->devs_b.resize(n_devices);
        ctx->residents.resize(n_devices);
        for (int k = 0; k < n_devices; k++) {
            int id = device_ids ? device_ids[k] : k;
            if (id < 0 || id >= count) return MI_E_NO_DEVICE;
            HIP_TRY(hipSetDevice(id));
            for (DevState* d : {&ctx->devs[k], &ctx->devs_b[k]}) {
                d->dev = id;
                d->res = ctx->residents[k].data();
                HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
                for (auto& e : d->ev) HIP_TRY(hipEventCreate(&e));
            }
        }
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        ctx->pool.reset(new hostpool::Pool(std::min(hw, 16u) - 1));
        return MI_OK;
    });
    if (rc != MI_OK) {
        mi_msm_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MI_OK;
}

void mi_msm_destroy(mi_ctx* ctx) {
    if (!ctx) return;
    for (std::vector<DevState>* lane : {&ctx->devs, &ctx->devs_b})
    for (auto& d : *lane) {
        (void)hipSetDevice(d.dev);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        for (DevBuf* b : {&d.raw, &d.call_bases, &d.call_flags, &d.scalars, &d.hist, &d.offsets, &d.woff, &d.meta, &d.sched, &d.sorted,
                          &d.partial, &d.order, &d.item_bucket, &d.pairs, &d.tilecnt, &d.bin_tot, &d.bin_base, &d.coarse, &d.seg_cnt, &d.seg_base, &d.segcnt, &d.merge_list,
                          &d.pr_p, &d.pr_q, &d.pr_lvl[0], &d.pr_lvl[1], &d.pr_raw, &d.pr_lines})
            b->release();
        if (d.h_pairs) (void)hipHostFree(d.h_pairs);
        for (auto& e : d.ev)
            if (e) (void)hipEventDestroy(e);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    for (size_t k = 0; k < ctx->residents.size() && k < ctx->devs.size(); k++) {
        (void)hipSetDevice(ctx->devs[k].dev);
        for (Resident& x : ctx->residents[k]) { x.buf.release(); x.flags.release(); }
    }
    delete ctx;
}

int mi_msm_num_devices(const mi_ctx* ctx) { return ctx ? (int)ctx->devs.size() : 0; }

int mi_msm_g1_set_bases(mi_ctx* ctx, const mi_g1_affine* bases, size_t n) { return set_bases_impl<msmk::G1C>(ctx, bases, n); }
int mi_msm_g2_set_bases(mi_ctx* ctx, const mi_g2_affine* bases, size_t n) { return set_bases_impl<msmk::G2C>(ctx, bases, n); }

int mi_msm_g1(mi_ctx* ctx, const mi_g1_affine* bases, const uint8_t* scalars, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return msm_impl<msmk::G1C>(ctx, bases, scalars, false, n, scalar_fmt, out);
}
int mi_msm_g2(mi_ctx* ctx, const mi_g2_affine* bases, const uint8_t* scalars, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return msm_impl<msmk::G2C>(ctx, bases, scalars, false, n, scalar_fmt, out);
}
int mi_msm_g1_device(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return msm_impl<msmk::G1C>(ctx, nullptr, static_cast<const uint8_t*>(d_scalars), true, n, scalar_fmt, out);
}
int mi_msm_g2_device(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return msm_impl<msmk::G2C>(ctx, nullptr, static_cast<const uint8_t*>(d_scalars), true, n, scalar_fmt, out);
}

int mi_msm_g1_batch(mi_ctx* ctx, const uint8_t* const* scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return msm_batch_impl<msmk::G1C>(ctx, scalars, k, n, scalar_fmt, out);
}
int mi_msm_g2_batch(mi_ctx* ctx, const uint8_t* const* scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return msm_batch_impl<msmk::G2C>(ctx, scalars, k, n, scalar_fmt, out);
}

int mi_g1_normalize_batch(mi_ctx* ctx, const mi_g1* in, size_t n, mi_g1_affine* out) {
    if (!ctx || (n && (!in || !out))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int { normalize_batch_dev<msmk::G1C>(ctx, ctx->devs[0], in, n, out); return MI_OK; });
}
int mi_g2_normalize_batch(mi_ctx* ctx, const mi_g2* in, size_t n, mi_g2_affine* out) {
    if (!ctx || (n && (!in || !out))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int { normalize_batch_dev<msmk::G2C>(ctx, ctx->devs[0], in, n, out); return MI_OK; });
}

int mi_g1_deserialize_batch(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g1_affine* out,
                            uint8_t* status) {
    return deserialize_impl(ctx, msmk::k_deserialize_g1, 48, bytes, n, compressed, validate, out, status);
}
int mi_g1_serialize_batch(mi_ctx* ctx, const mi_g1_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl(ctx, msmk::k_serialize_g1, 48, points, n, compressed, bytes);
}
int mi_g2_deserialize_batch(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g2_affine* out,
                            uint8_t* status) {
    return deserialize_impl(ctx, msmk::k_deserialize_g2, 96, bytes, n, compressed, validate, out, status);
}
int mi_g2_serialize_batch(mi_ctx* ctx, const mi_g2_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl(ctx, msmk::k_serialize_g2, 96, points, n, compressed, bytes);
}

int mi_multi_miller_loop(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out) {
    return miller_impl(ctx, p, q, n, out, false);
}
int mi_multi_pairing(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out) {
    return miller_impl(ctx, p, q, n, out, true);
}
int mi_final_exponentiation(const mi_fp12* f, mi_fp12* out) {
    if (!f || !out) return MI_E_INVALID;
    fp12_to_raw(out, HT::final_exp(fp12_from_raw(f)));
    return MI_OK;
}

int mi_g1_sum(const mi_g1* partials, size_t n, mi_g1* out) {
    if (!out || (n && !partials)) return MI_E_INVALID;
    G1 r = G1::inf();
    for (size_t i = 0; i < n; i++) {
        G1 p;
        memcpy(&p, &partials[i], sizeof p);
        r = r.add(p);
    }
    memcpy(out, &r, sizeof r);
    return MI_OK;
}

int mi_g2_sum(const mi_g2* partials, size_t n, mi_g2* out) {
    if (!out || (n && !partials)) return MI_E_INVALID;
    hostec::G2 r = hostec::G2::inf();
    for (size_t i = 0; i < n; i++) {
        hostec::G2 p;
        memcpy(&p, &partials[i], sizeof p);
        r = r.add(p);
    }
    memcpy(out, &r, sizeof r);
    return MI_OK;
}

int mi_msm_set_window_bits(mi_ctx* ctx, unsigned window_bits) {
    if (!ctx || (window_bits != 0 && (window_bits < 7 || window_bits > 22))) return fail(ctx, MI_E_INVALID, "window_bits must be 0 or 7..22");
    LaneLock lk(ctx, true);
    ctx->forced_c = window_bits;
    return MI_OK;
}

int mi_msm_last_profile(const mi_ctx* ctx, mi_profile* out) {
    if (!ctx || !out) return MI_E_INVALID;
    std::lock_guard<std::mutex> lk(ctx->info_mu);
    *out = ctx->prof;
    return MI_OK;
}

// text of the most recent failure on this context (valid until the next failing call)
const char* mi_msm_last_error(const mi_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char* mi_msm_strerror(int code) {
    switch (code) {
        case MI_OK: return "ok";
        case MI_E_INVALID: return "invalid argument";
        case MI_E_NO_DEVICE: return "no usable HIP device";
        case MI_E_HIP: return "HIP runtime error";
        case MI_E_NOMEM: return "out of memory";
        case MI_E_NO_BASES: return "no resident base set";
        case MI_E_UNSUPPORTED: return "not supported in this build";
        default: return "unknown error";
    }
}

int mi_test_fp_op(mi_ctx* ctx, int op, const mi_fp* a, const mi_fp* b, mi_fp* out, size_t n) {
    if (!ctx || !a || !b || !out || op < 0 || op > 3) return fail(ctx, MI_E_INVALID, "invalid argument");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        HIP_TRY(hipSetDevice(d.dev));
        DevBuf da, db, dout;
        da.ensure(n * 48); db.ensure(n * 48); dout.ensure(n * 48);
        HIP_TRY(hipMemcpyAsync(da.p, a, n * 48, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(db.p, b, n * 48, hipMemcpyHostToDevice, d.stream));
        hipLaunchKernelGGL(msmk::k_test_fp_op, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, op, (const uint32_t*)da.p,
                           (const uint32_t*)db.p, (uint32_t*)dout.p, (uint32_t)n);
        HIP_TRY(hipMemcpyAsync(out, dout.p, n * 48, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        da.release(); db.release(); dout.release();
        return MI_OK;
    });
}

}  // extern "C"
