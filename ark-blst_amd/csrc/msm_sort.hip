// Window-size plan, two-level LDS-staged bucket sort and work-item schedule of one MSM call (curve independent).
// Plays the role of calc_window_size / work_units / calc_chunk_size of the reference driver (/root/reference/src/gpu.rs:37-92,218-223).
#include <cmath>
#include "internal.hpp"
#include "sort_kernels.cuh"

namespace mi {

namespace {
struct ReduceCost { double coop_steps = 0, coop_rounds = 0, serial_steps = 0, serial_rounds = 0; };

// Reduce geometry of a plan whose c, nb, bwin and nbuckets are set (a whole call, or one window group of a pipelined call: fewer bucket
// windows, hence fewer buckets per logical lane and a shorter latency chain).  false: c is too small for the reduce's lane scheme.
bool reduce_geometry(Plan& p, const CurveCost& cc, ReduceCost& rc) {
    const uint32_t c = p.c;
    // L buckets per logical lane, ANY value up to 64 (round 4; a power of two before): the L that costs the fewest
    // (rounds of max_chunks wave slots) x (steps of the wave's latency chain).  2^16 points at c = 15 (17 windows of 2^14 buckets):
    // L = 9, 1938 waves of 31 steps; the power-of-two geometry needed L = 16 (L = 8 gives 2176 waves, a second round) and 45 steps.
    const uint32_t log_ll = (uint32_t)cc.log_ll;
    if (c - 1 < log_ll) return false;
    {
        double best = 1e300;
        const uint32_t max_L = std::min<uint32_t>(64, std::max<uint32_t>(1, p.nb >> log_ll));
        for (uint32_t L = 1; L <= max_L; L++) {
            const uint64_t chunks = (uint64_t)((p.nb + (L << log_ll) - 1) / (L << log_ll)) * p.bwin;
            const double rounds = (double)((chunks + cc.max_chunks - 1) / cc.max_chunks);
            uint32_t bits = 0, ones = 0;
            for (uint32_t v = L; v; v >>= 1) { bits++; ones += v & 1u; }
            const double steps = 2.0 * L + 2.0 * log_ll + (bits - 1) + (ones - 1) + 1.0;
            if (rounds * steps < best) { best = rounds * steps; p.coop_L = L; rc.coop_steps = steps; rc.coop_rounds = rounds; }
        }
    }
    p.chunk_buckets = p.coop_L << log_ll;
    p.serial_reduce = cc.serial_buckets != 0 && p.nbuckets >= cc.serial_buckets && c - 1 >= 6;
    uint32_t serial_L_lo = 8, serial_L_hi = 64;
#if defined(MI_TEST_HOOKS)
    // experiment switches of the test build (tools/sweep_sizes.py --test-hooks): force the single-lane reduce from a bucket count on, pin its L
    if (const char* e = getenv("MI_TEST_SERIAL_MIN_BUCKETS")) p.serial_reduce = cc.serial_buckets != 0 && p.nbuckets >= (uint64_t)atoll(e) && c - 1 >= 6;
    if (const char* e = getenv("MI_TEST_SERIAL_L")) serial_L_lo = serial_L_hi = (uint32_t)std::min(64, std::max(1, atoi(e)));
#endif
    p.chunks_per_win = (p.nb + p.chunk_buckets - 1) / p.chunk_buckets;
    p.serial_L = 0;
    if (p.serial_reduce) {
        // one lane per L consecutive buckets of a window, L <= 64 and NOT necessarily a power of two: the L that costs the fewest
        // (rounds of wave slots) x (2 L running-sum steps + the double-and-add chain that turns S into L S).  2^24 points at c = 20:
        // L = 53 fills 2010 of the 2048 slots with 114 steps, where L = 64 left 1664 waves walking 134 (3.0 -> 2.6 ms).
        double best = 1e300;
        for (uint32_t L = serial_L_lo; L <= serial_L_hi; L++) {
            const uint64_t lanes = (uint64_t)((p.nb + L - 1) / L) * p.bwin;
            const double rounds = std::ceil(std::ceil((double)lanes / 64.0) / (double)cc.max_chunks);
            uint32_t bits = 0, ones = 0;
            for (uint32_t v = L; v; v >>= 1) { bits++; ones += v & 1u; }
            const double steps = 2.0 * L + (bits - 1) + (ones - 1);
            if (rounds * steps < best) { best = rounds * steps; p.serial_L = L; rc.serial_steps = steps; rc.serial_rounds = rounds; }
        }
        p.chunk_buckets = p.serial_L; p.coop_L = 0;
        p.chunks_per_win = (p.nb + p.serial_L - 1) / p.serial_L;
    }
    p.nchunks = (uint64_t)p.chunks_per_win * p.bwin;
    return true;
}
}  // namespace

// Window size c by a time model of the pipeline on one MI355X (microseconds; constants measured, see DESIGN_HISTORY.md §8):
//   accumulate  max(throughput: N W mixed additions at cc.add_per_us,  latency: one lane walks an item of T entries)
//   merge       one launch per binary-tree level when the short top window overfills its buckets
//   reduce      a latency chain of 2L + 2 LOG_LL + chain(L) + 1 complete additions per wave, max_chunks waves per round
//   combine     2 LOG_LL + 1 additions per level;  sort  N W entries at 4.1e10 /s;  schedule ~ buckets / 1e4;  host Horner ~ 100 us
Plan make_plan(size_t n, unsigned forced_c, const CurveCost& cc, bool shared, size_t stride, bool fold) {
    Plan best{};
    double best_cost = 1e300;
    for (unsigned c = 7; c <= 22; c++) {
        if (forced_c && c != forced_c) continue;
        Plan p{};
        p.c = c;
        p.win0 = 0;
        p.fold = fold;
        p.nwin = num_windows(c, fold);
        // sort geometry: lo bits share a 32-bit entry with the point index and the sign; the coarse bins of one
        // window must fit the LDS counter array
        uint32_t idx_bits = 1;
        const uint64_t nidx = shared ? (uint64_t)std::max(stride, n) * p.nwin : n;
        while ((1ull << idx_bits) < nidx) idx_bits++;
        if (idx_bits > 30) continue;
        uint32_t lo_bits = std::min<uint32_t>(std::min<uint32_t>(8, c - 1), 31 - idx_bits);
        if (((1u << (c - 1)) >> lo_bits) > msmk::SORT_MAX_COUNTERS) continue;
        p.lo_bits = lo_bits;
        if ((uint64_t)n * p.nwin >= (1ull << 32)) continue;   // entry offsets are 32-bit
        p.nb = 1u << (c - 1);
        p.bwin = shared ? 1 : p.nwin;
        p.nbuckets = (uint64_t)p.nb * p.bwin;
        ReduceCost rc;
        if (!reduce_geometry(p, cc, rc)) continue;
        const double coop_steps = rc.coop_steps, coop_rounds = rc.coop_rounds, serial_steps = rc.serial_steps, serial_rounds = rc.serial_rounds;
        // The top window holds only 254 (fold) or 255 - c (nwin - 1) significant bits (0: the carry-only window of c = 15 / 17 without the fold): its n entries share 2^top_bits buckets (on top of the
        // others' entries when all windows share one bucket set)
        const double entries = (double)n * p.nwin;
        double mean = entries / (double)p.nbuckets;
        int top_bits = std::max(0, std::min<int>((fold ? 254 : 255) - (int)c * ((int)p.nwin - 1), (int)c - 1));
        double per_bucket = (double)n / (double)(1u << top_bits) + (shared ? mean : 0.0);
        // work-item size: twice the mean bucket load (uniform scalars then never split), at least 32 entries; the fuller buckets
        // of the top window count as the mean while they are within 4x of it (splitting them would cost a merge launch for
        // nothing); buckets beyond T entries are split and merged by a binary tree
        const double typical = per_bucket > mean && per_bucket <= 4.0 * mean ? per_bucket : mean;
        p.logT = 5;
        while ((double)(1u << p.logT) < 2.0 * typical && p.logT < 20) p.logT++;
        // ... and an item may be as long as a lane walks in a fifth of the kernel's time anyway: at 2^24 points (c = 20) the top window's
        // 512-entry buckets then stay whole (8 items each and three merge launches, 0.5 ms, for nothing: the kernel runs 30 ms)
        p.cls_shift = p.logT > 6 ? p.logT - 6 : 0;
        const double walk = 0.2 * ((entries + 0.2 * (double)p.nbuckets) / cc.add_per_us) / cc.lane_add_us;
        while (p.logT < 20 && (double)(2u << p.logT) <= walk) p.logT++;
        const double T = (double)(1u << p.logT);
        p.logS = std::max<uint32_t>(4, p.logT - 2);   // buckets beyond T entries are cut into items of S = max(16, T / 4) (sort_kernels.cuh items_of; logT >= 5)
        // the fan-in tree over the items of the fullest expected buckets (k_merge): the short top window's 2^top_bits buckets, n / 2^top_bits
        // entries each.  A level costs a launch, or — many items — rounds of (FAN - 1) complete additions on every wave slot's logical lanes
        double merge_total_us = 0;
        if (per_bucket > T) {
            const double lanes_at_once = (double)cc.max_chunks * (double)(1u << cc.comb_log_ll);
            double per = per_bucket / (double)(1u << p.logS), all = per * (double)(1u << top_bits);
            for (; per > 1.0; per /= (double)msmk::MERGE_FAN, all /= (double)msmk::MERGE_FAN)
                merge_total_us += std::max(cc.merge_us, std::ceil(all / (double)msmk::MERGE_FAN / lanes_at_once) * (msmk::MERGE_FAN - 1) * cc.step_us);
            // ... and n entries aimed at a handful of buckets contend for the same LDS counters in every pass of the sort and make short
            // items of the accumulate kernel's first rounds (2^23 points at c = 19, a 7-bit top window: sort + 0.6 ms, accumulate + 0.7 ms
            // against the plain model; profiles/r04_scan_c_g1_2p21_2p24_merge_tree.jsonl)
            merge_total_us += 1.6e-4 * (double)n;
        }
        // entries the longest ordinary lane walks serially: bucket loads are Poisson distributed, the kernel ends with the tail
        // (round 4: the kernel ends with the LONGEST of ~10^5 items — mean + 4.5 sigma, not + 3 — and a lone lane's addition costs 14 us
        // with its dependent gather, not 11; with the old figures the plan chose c = 13 at 2^14 points, 0.86 ms where c = 16 takes 0.79:
        // profiles/r04_scan_c_g1_2p14_2p18.jsonl)
        const double tail = mean + 4.5 * std::sqrt(mean);
        // (a bucket beyond T entries is cut into items of S: the fullest expected bucket then costs S entries per lane plus its merge levels)
        // the top window's buckets have a tail of their own; a top window of under 4 bits also has a carry-only bucket that fills to just below T
        // (one unsplit item of ~T entries: 2^10 points at c = 9, 12, 14 measure 0.33-0.36 ms of accumulate where c = 10 takes 0.16)
        double item_len = std::min(T, std::max(tail, per_bucket > T ? (double)(1u << p.logS) : per_bucket + 3.0 * std::sqrt(per_bucket)));
        if (per_bucket > T && top_bits < 4) item_len = T;
        int levels = 0;
        for (uint32_t m = p.chunks_per_win; m > 1; m = (m + (1u << cc.comb_log_ll) - 1) >> cc.comb_log_ll) levels++;
        const double reduce_us = p.serial_reduce ? serial_rounds * serial_steps * cc.serial_step_us
                                                 : coop_rounds * coop_steps * (p.nchunks <= 1000 ? cc.lone_step_us : cc.step_us);   // lone waves step faster
        // combine: one latency chain per level; the first level of a long pair list runs in several rounds of 2048 waves
        const double comb_chain = (2.0 * cc.comb_log_ll + 1.0) * cc.comb_step_us;
        const double comb_us = std::max(1, levels) * (comb_chain + 8.0) +
                               std::max(0.0, std::ceil((double)(p.nchunks >> cc.comb_log_ll) / 2048.0) - 1.0) * comb_chain;
        // accumulate: the first entry of an item only initialises the running sum (no addition), every item pays ~1.2 additions' worth of
        // dependent index loads, XYZZ -> projective conversion and store; with few work items the kernel's time is a whole number of
        // rounds of 2048 wave slots and the ragged last round counts (round 4: refitted on profiles/r04_scan_c_g1_2p14_2p23_signfold.jsonl —
        // the old "+ 1.4 per bucket" made c = 15 look 15 % cheaper than c = 16 at 2^16 points where the two measure the same)
        const double nonempty = (double)p.nbuckets * (1.0 - std::exp(-mean));
        double acc_adds = entries - nonempty + 1.2 * (double)p.nbuckets;
        const double wave_rounds = (double)p.nbuckets / 64.0 / (double)cc.max_chunks;
        if (wave_rounds < 4.0) acc_adds *= 1.0 + 0.5 * (std::ceil(wave_rounds) - wave_rounds) / std::max(wave_rounds, 0.25);
        // a lane's addition takes lane_add_us with every wave slot taken; with at most one wave per SIMD (few items) two thirds of it (G1) / under half (G2's lane pairs)
        const double lane_us = cc.lane_add_us * ((double)p.nbuckets / cc.acc_wave_items <= 1024.0 ? cc.lone_lane : 1.0);
        // the kernel drains over the life of its SHORTEST items (the schedule starts the longest first): half a wave generation of
        // mean - 3 sigma entries is lost at the end — fuller buckets, longer drain (c = 15 vs 16 at 2^20 points: 10 % fewer additions per us)
        const double drain_us = 0.5 * std::max(0.0, mean - 3.0 * std::sqrt(mean)) * cc.lane_add_us;
        double cost = std::max(acc_adds / cc.add_per_us + drain_us, item_len * lane_us) + merge_total_us + reduce_us + comb_us +
                      entries / 41000.0 + (double)p.nbuckets / 1e4 + (shared ? 20.0 : 100.0);
        if (cost < best_cost) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

// Window groups of a pipelined call (run_msm), TOP windows first: group plans share c, the item geometry (logT / logS / class width), the
// sort's bit split and the reduce geometry with the whole-call plan `pl`; each has its own bucket windows.  weights: relative sizes of the
// groups (empty = the built-in choice, {1} = one group); a weight list longer than the window count is cut.
std::vector<Plan> split_plan(const Plan& pl, const CurveCost& cc, size_t n, bool shared, const std::vector<unsigned>& weights) {
    (void)cc;
    std::vector<unsigned> w = weights;
    if (w.empty()) {
        // built-in: ONE group.  The pipelined form (mi_msm_set_pipeline / ARKBLST_AMD_PIPELINE) hides the second group's sort, but it needs
        // its streams on separate hardware queues and the HIP runtime assigns those by its own bookkeeping (DevState::ensure_pipeline_streams):
        // same-box A/B in two processes, 2^20 / 2^22 / 2^24 points — tools/pipe_scan.py -2 % / -3 % / -4 %, bench.py +0 .. +30 % (profiles/
        // r06_pipeline_ab.txt).  A gain that depends on which queue a stream lands on is not a default.
        w = {1};
    }
    if (w.size() > (size_t)MAX_GROUPS) w.resize(MAX_GROUPS);
    if (w.size() > pl.nwin) w.resize(pl.nwin);
    if (shared || w.size() <= 1 || pl.bwin != pl.nwin) return {pl};
    // window counts by largest remainder, at least one window per group
    const size_t G = w.size();
    uint64_t sum = 0;
    for (unsigned& x : w) { x = std::max(1u, x); sum += x; }
    std::vector<uint32_t> cnt(G);
    std::vector<double> frac(G);
    uint32_t used = 0;
    for (size_t g = 0; g < G; g++) {
        const double exact = (double)pl.nwin * w[g] / (double)sum;
        cnt[g] = std::max<uint32_t>(1, (uint32_t)exact);
        frac[g] = exact - (double)cnt[g];
        used += cnt[g];
    }
    while (used < pl.nwin) {
        size_t b = 0;
        for (size_t g = 1; g < G; g++) if (frac[g] > frac[b]) b = g;
        cnt[b]++; frac[b] -= 1.0; used++;
    }
    while (used > pl.nwin) {
        size_t b = 0;
        for (size_t g = 1; g < G; g++) if (cnt[g] > cnt[b]) b = g;
        cnt[b]--; used--;
    }
    std::vector<Plan> out;
    uint32_t top = pl.nwin;
    for (size_t g = 0; g < G; g++) {
        Plan p = pl;
        p.nwin = p.bwin = cnt[g];
        top -= cnt[g];
        p.win0 = top;
        p.nbuckets = (uint64_t)p.nb * p.bwin;
        // the reduce geometry (buckets per lane, chunk size) stays the whole call's: the groups' reductions run TOGETHER after the last
        // accumulate kernel (run_msm), and all their waves together fill one round of wave slots, as the single launch of a one-group call does
        p.nchunks = (uint64_t)p.chunks_per_win * p.bwin;
        out.push_back(p);
    }
    return out;
}

namespace {

// k_coarse is compiled per window size (static digit extraction): dispatch on c = 7..22
template <bool SCATTER, int CB = 7>
void launch_coarse(uint32_t c, dim3 grid, dim3 block, hipStream_t s, const uint32_t* scalars, const uint8_t* flags, const msmk::SortGeom& g,
                   uint32_t* tilecnt, const uint32_t* tileoff, const uint32_t* bin_base, uint32_t* coarse) {
    if constexpr (CB > 22) {
        throw HipFail{"window_bits out of range"};
    } else {
        if (c == CB)
            hipLaunchKernelGGL((msmk::k_coarse<SCATTER, CB>), grid, block, 0, s, scalars, flags, g, tilecnt, tileoff, bin_base, coarse);
        else
            launch_coarse<SCATTER, CB + 1>(c, grid, block, s, scalars, flags, g, tilecnt, tileoff, bin_base, coarse);
    }
}
// level A of the three-level sort (c = 17..22)
template <bool SCATTER, int CB = 17>
void launch_coarseA(uint32_t c, dim3 grid, dim3 block, hipStream_t s, const uint32_t* scalars, const uint8_t* flags, const msmk::SortGeom& g,
                    uint32_t* tilecnt, const uint32_t* tileoff, const uint32_t* binA_base, uint2* coarseA) {
    if constexpr (CB > 22) {
        throw HipFail{"three-level sort: window_bits out of range"};
    } else {
        if (c == CB)
            hipLaunchKernelGGL((msmk::k_coarseA<SCATTER, CB>), grid, block, 0, s, scalars, flags, g, tilecnt, tileoff, binA_base, coarseA);
        else
            launch_coarseA<SCATTER, CB + 1>(c, grid, block, s, scalars, flags, g, tilecnt, tileoff, binA_base, coarseA);
    }
}
// the LDS-staged scatter exists for the window sizes whose coarse bins allow it (H <= 128: c <= 16)
template <int CB = 7>
void launch_coarse_staged(uint32_t c, bool co, dim3 grid, hipStream_t s, const uint32_t* scalars, const uint8_t* flags, const msmk::SortGeom& g,
                          const uint32_t* tilecnt, const uint32_t* tileoff, const uint32_t* bin_base, uint32_t* coarse) {
    if constexpr (CB > 16) {
        throw HipFail{"staged coarse scatter: window_bits out of range"};
    } else {
        if (c == CB)
            if (co) hipLaunchKernelGGL((msmk::k_coarse_staged_co<CB>), grid, dim3(256), msmk::COARSE_STAGED_CO_LDS, s, scalars, flags, g, tilecnt, tileoff, bin_base, coarse);
            else hipLaunchKernelGGL((msmk::k_coarse_staged<CB>), grid, dim3(512), 0, s, scalars, flags, g, tilecnt, tileoff, bin_base, coarse);
        else
            launch_coarse_staged<CB + 1>(c, co, grid, s, scalars, flags, g, tilecnt, tileoff, bin_base, coarse);
    }
}

}  // namespace

void sort_and_schedule(DevState& d, Scratch& sc, hipStream_t s, const Plan& pl, const uint32_t* d_scalars, const uint8_t* d_flags, size_t n, unsigned fmt,
                       bool shared_buckets, size_t stride, SortOut& out, const uint8_t* host_scalars, bool under_accumulate) {
    // Workgroups of a pipelined call's sort must find room BESIDE the two resident waves of an accumulate kernel (2 x 216 of a SIMD's 512
    // registers per lane: 80 are left): 256 lanes = one wave per SIMD of <= 80 registers does, the 1024-lane blocks of the digit passes
    // (4 x 56) would wait until a whole compute unit drains — which an accumulate kernel with queued waves never lets happen
    const dim3 digit_block_big(under_accumulate ? 256 : 1024);
    const size_t entries_cap = (size_t)n * pl.nwin;
    sc.hist.ensure(pl.nbuckets * 4);
    sc.offsets.ensure((pl.nbuckets + 1) * 4);
    sc.woff.ensure((pl.nbuckets + 1) * 4);
    sc.meta.ensure(msmk::MERGE_META * 4);
    sc.sorted.ensure(entries_cap * 4);
    if (!sc.h_meta) HIP_TRY(hipHostMalloc((void**)&sc.h_meta, 32, hipHostMallocDefault));

    // phase events only on request (profile level 2): each record leaves the device idle for ~6 us between two kernels
    const bool phases = d.prof_level >= 2;
    if (phases) HIP_TRY(hipEventRecord(sc.ev[0], s));
    // ---- two-level LDS-staged bucket sort (geometry in msmk::SortGeom)
    // The count pass, per chunk of tiles: with host scalars chunk j is copied on the copy stream and counted as soon as it has
    // landed, while chunk j + 1 is still crossing PCIe; device-resident scalars are one chunk.
    auto feed_and_count = [&](msmk::SortGeom& gg, auto&& count_tiles) {
        const uint32_t K = host_scalars ? std::min<uint32_t>(8, std::max<uint32_t>(1, std::min<uint32_t>(gg.tiles, (uint32_t)(n >> 16)))) : 1;
        if (host_scalars) HIP_TRY(hipEventRecord(d.cev[0], d.copy_stream));
        for (uint32_t j = 0; j < K; j++) {
            const uint32_t t0 = (uint32_t)((uint64_t)gg.tiles * j / K), t1 = (uint32_t)((uint64_t)gg.tiles * (j + 1) / K);
            if (host_scalars) {
                const size_t p0 = (size_t)t0 * gg.tile_pts, p1 = std::min<size_t>(n, (size_t)t1 * gg.tile_pts);
                HIP_TRY(hipMemcpyAsync((char*)const_cast<uint32_t*>(d_scalars) + p0 * 32, host_scalars + p0 * 32, (p1 - p0) * 32, hipMemcpyHostToDevice,
                                       d.copy_stream));
                HIP_TRY(hipEventRecord(d.cev[1 + j], d.copy_stream));
                HIP_TRY(hipStreamWaitEvent(s, d.cev[1 + j], 0));
            }
            gg.tile0 = t0;
            if (t1 > t0) count_tiles(t1 - t0);
        }
        gg.tile0 = 0;
        if (host_scalars) {
            HIP_TRY(hipEventRecord(d.cev[9], d.copy_stream));
            HIP_TRY(hipStreamWaitEvent(s, d.cev[9], 0));   // the call's final synchronisation of `s` then covers the event read for h2d_ms
        }
    };
    msmk::SortGeom g{};
    g.n = (uint32_t)n; g.fmt = fmt | (pl.fold ? 2u : 0u); g.c = pl.c; g.nwin = pl.nwin; g.win0 = pl.win0;
    g.shared = shared_buckets ? 1u : 0u;
    g.stride = (uint32_t)stride;
    g.lo_bits = pl.lo_bits;
    g.H = pl.nb >> g.lo_bits;
    g.nbins = pl.bwin * g.H;
    sc.bin_base.ensure((size_t)(g.nbins + 1) * 4);
    sc.coarse.ensure(entries_cap * 4);
    // fine-level scratch (upper bound on the segment count: one per bin plus one per FINE_SEG entries); sized here, with the needs of
    // level B below, so that no buffer grows between two launches of a call
    const uint32_t segs_cap = g.nbins + (uint32_t)(entries_cap / msmk::FINE_SEG) + 1;
    const bool three_level = !shared_buckets && pl.c >= 17;
    {
        size_t seg_words = (size_t)segs_cap * (1u << g.lo_bits), bins = g.nbins;
        if (three_level) {
            const uint32_t nbinsA = pl.nwin * msmk::A_BINS;
            const size_t segs_capA = nbinsA + entries_cap / msmk::FINE_SEG + 1;
            seg_words = std::max(seg_words, segs_capA << ((pl.c - 1 - msmk::A_BITS) - pl.lo_bits));
            bins = std::max<size_t>(bins, nbinsA);
        }
        sc.seg_cnt.ensure(bins * 4);
        sc.seg_base.ensure((bins + 1) * 4);
        sc.segcnt.ensure(seg_words * 4);
        sc.segoff.ensure(seg_words * 4);
    }
    if (three_level) {
        // ---- c >= 17: level A (window, top 5 bits: all windows in one pass, 8-byte entries), level B (LDS-staged split by the next
        // MID bits into the 4-byte entries and (window, hi) bins of the one-step pass); see sort_kernels.cuh
        const uint32_t nbinsA = pl.nwin * msmk::A_BINS;
        const uint32_t mid_bits = (pl.c - 1 - msmk::A_BITS) - pl.lo_bits, M = 1u << mid_bits;
        if (nbinsA > 512 || M > msmk::MID_MAX || nbinsA * M != g.nbins) throw HipFail{"three-level sort: geometry"};
        g.wgroup = pl.nwin; g.ngroups = 1;
        g.tile_pts = (uint32_t)std::min<size_t>(8192, (n + 1023) / 1024 * 1024);
        g.tiles = (uint32_t)((n + g.tile_pts - 1) / g.tile_pts);
        const uint32_t segs_capA = nbinsA + (uint32_t)(entries_cap / msmk::FINE_SEG) + 1;
        sc.tilecnt.ensure((size_t)g.tiles * nbinsA * 4);
        sc.tileoff.ensure((size_t)g.tiles * nbinsA * 4);
        sc.bin_tot.ensure((size_t)nbinsA * 4);
        sc.binA_base.ensure((size_t)(nbinsA + 1) * 4);
        sc.coarseA.ensure(entries_cap * 8);
        feed_and_count(g, [&](uint32_t ntiles) {
            launch_coarseA<false>(pl.c, dim3(ntiles), digit_block_big, s, d_scalars, d_flags, g, (uint32_t*)sc.tilecnt.p, (const uint32_t*)nullptr,
                                  (const uint32_t*)nullptr, (uint2*)nullptr);
        });
        hipLaunchKernelGGL(msmk::k_colscan, dim3((nbinsA + 255) / 256), dim3(256), 0, s, (const uint32_t*)sc.tilecnt.p, nbinsA, g.tiles,
                           (uint32_t*)sc.tileoff.p, (uint32_t*)sc.bin_tot.p);
        hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)sc.bin_tot.p, nbinsA, (uint32_t*)sc.binA_base.p);
        launch_coarseA<true>(pl.c, dim3(g.tiles), digit_block_big, s, d_scalars, d_flags, g, (uint32_t*)sc.tilecnt.p, (const uint32_t*)sc.tileoff.p,
                             (const uint32_t*)sc.binA_base.p, (uint2*)sc.coarseA.p);
        hipLaunchKernelGGL(msmk::k_seg_count, dim3((nbinsA + 255) / 256), dim3(256), 0, s, (const uint32_t*)sc.binA_base.p, nbinsA,
                           (uint32_t*)sc.seg_cnt.p);
        hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)sc.seg_cnt.p, nbinsA, (uint32_t*)sc.seg_base.p);
        hipLaunchKernelGGL(msmk::k_mid_count, dim3(segs_capA), dim3(256), 0, s, (const uint2*)sc.coarseA.p, (const uint32_t*)sc.binA_base.p,
                           (const uint32_t*)sc.seg_base.p, nbinsA, pl.lo_bits, mid_bits, (uint32_t*)sc.segcnt.p);
        hipLaunchKernelGGL(msmk::k_mid_scan, dim3(nbinsA), dim3(512), 0, s, (const uint32_t*)sc.binA_base.p, (const uint32_t*)sc.seg_base.p, mid_bits,
                           (const uint32_t*)sc.segcnt.p, (uint32_t*)sc.segoff.p, (uint32_t*)sc.bin_base.p, nbinsA);
        hipLaunchKernelGGL(msmk::k_mid_scatter, dim3(segs_capA), dim3(256), msmk::MID_SCATTER_LDS, s, (const uint2*)sc.coarseA.p, (const uint32_t*)sc.binA_base.p,
                           (const uint32_t*)sc.seg_base.p, nbinsA, pl.lo_bits, mid_bits, (const uint32_t*)sc.segcnt.p, (const uint32_t*)sc.segoff.p,
                           (uint32_t*)sc.coarse.p);
    } else {
    // A tile contributes tile_pts / H entries to each coarse bin of a window, written as one contiguous run: keep
    // runs >= 64 entries (256 B) or the 4-byte scatter is write-amplified (9.5 ms at n = 2^24 with 16-entry runs).
    // With few coarse bins per window (H <= 128, i.e. c <= 16) the scatter is staged through LDS instead: a workgroup takes
    // as many windows as keep wgroup * H <= 512 bins and as many points as keep its entries within 64 KB of LDS.
    const bool staged = !shared_buckets && g.H <= 128 && pl.c <= 16 && n >= 4096;
    uint32_t coarse_block;
    if (staged) {
        g.wgroup = std::min<uint32_t>(std::min<uint32_t>(pl.nwin, 16), msmk::COARSE_STAGE_BINS / g.H);
        // tile_pts * wgroup entries fit the staging buffer of the kernel that will run: k_coarse_staged (512 lanes, 16384 entries) or, beside an
        // accumulate kernel, k_coarse_staged_co (256 lanes, 14336 entries)
        g.tile_pts = under_accumulate ? (msmk::COARSE_STAGE_CO / g.wgroup) / 256 * 256 : (msmk::COARSE_STAGE / g.wgroup) / 512 * 512;
        coarse_block = under_accumulate ? 256 : 512;
    } else {
        size_t want = std::max<size_t>(std::max<size_t>(4096, n / 512), (size_t)64 * g.H / (shared_buckets ? pl.nwin : 1));
        g.tile_pts = (uint32_t)((std::min(want, n) + 1023) / 1024 * 1024);
        coarse_block = g.tile_pts >= 16384 && !under_accumulate ? 1024 : 256;
        g.wgroup = shared_buckets ? pl.nwin : std::max<uint32_t>(1, std::min<uint32_t>(pl.nwin, msmk::SORT_MAX_COUNTERS / g.H));
    }
    g.tiles = (uint32_t)((n + g.tile_pts - 1) / g.tile_pts);
    g.ngroups = (pl.nwin + g.wgroup - 1) / g.wgroup;
    sc.tilecnt.ensure((size_t)g.tiles * g.nbins * 4);
    sc.tileoff.ensure((size_t)g.tiles * g.nbins * 4);
    sc.bin_tot.ensure((size_t)g.nbins * 4);
    feed_and_count(g, [&](uint32_t ntiles) {
        launch_coarse<false>(pl.c, dim3(ntiles, g.ngroups), dim3(coarse_block), s, d_scalars, d_flags, g, (uint32_t*)sc.tilecnt.p,
                             (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    });
    hipLaunchKernelGGL(msmk::k_colscan, dim3((g.nbins + 255) / 256), dim3(256), 0, s, (const uint32_t*)sc.tilecnt.p, g.nbins, g.tiles,
                       (uint32_t*)sc.tileoff.p, (uint32_t*)sc.bin_tot.p);
    hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)sc.bin_tot.p, g.nbins, (uint32_t*)sc.bin_base.p);
    if (staged)
        launch_coarse_staged(pl.c, under_accumulate, dim3(g.tiles, g.ngroups), s, d_scalars, d_flags, g, (const uint32_t*)sc.tilecnt.p, (const uint32_t*)sc.tileoff.p,
                             (const uint32_t*)sc.bin_base.p, (uint32_t*)sc.coarse.p);
    else
        launch_coarse<true>(pl.c, dim3(g.tiles, g.ngroups), dim3(coarse_block), s, d_scalars, d_flags, g, (uint32_t*)sc.tilecnt.p,
                            (const uint32_t*)sc.tileoff.p, (const uint32_t*)sc.bin_base.p, (uint32_t*)sc.coarse.p);
    }
    if (phases) HIP_TRY(hipEventRecord(sc.ev[1], s));
    // fine sort over bin segments
    hipLaunchKernelGGL(msmk::k_seg_count, dim3((g.nbins + 255) / 256), dim3(256), 0, s, (const uint32_t*)sc.bin_base.p, g.nbins,
                       (uint32_t*)sc.seg_cnt.p);
    hipLaunchKernelGGL(msmk::k_binscan, dim3(1), dim3(1024), 0, s, (const uint32_t*)sc.seg_cnt.p, g.nbins, (uint32_t*)sc.seg_base.p);
    hipLaunchKernelGGL(msmk::k_fine_count, dim3(segs_cap), dim3(256), 0, s, (const uint32_t*)sc.coarse.p, (const uint32_t*)sc.bin_base.p,
                       (const uint32_t*)sc.seg_base.p, g, (uint32_t*)sc.segcnt.p);
    hipLaunchKernelGGL(msmk::k_fine_scan, dim3(g.nbins), dim3(256), 0, s, (const uint32_t*)sc.seg_base.p, g, (const uint32_t*)sc.segcnt.p,
                       (uint32_t*)sc.segoff.p, (uint32_t*)sc.hist.p);
    hipLaunchKernelGGL(msmk::k_fine_scatter, dim3(segs_cap), dim3(256), msmk::FINE_SCATTER_LDS, s, (const uint32_t*)sc.coarse.p, (const uint32_t*)sc.bin_base.p,
                       (const uint32_t*)sc.seg_base.p, g, (const uint32_t*)sc.segcnt.p, (const uint32_t*)sc.segoff.p, (uint32_t*)sc.sorted.p);
    if (phases) HIP_TRY(hipEventRecord(sc.ev[2], s));
    // ---- schedule: <= SCHED_MAX_BLK blocks of 1024 (beside an accumulate kernel: 512) lanes, each lane owning per_blk / lanes consecutive buckets
    const uint32_t sched_nt = under_accumulate ? 512 : 1024;
    uint32_t per_blk = 4 * sched_nt;   // four buckets per lane (one per lane measured slower below 2^23 points: 0.073 against 0.043 ms)
    while ((pl.nbuckets + per_blk - 1) / per_blk > msmk::SCHED_MAX_BLK) per_blk <<= 1;
    uint32_t nblk = (uint32_t)((pl.nbuckets + per_blk - 1) / per_blk);
    // every bucket has at least one item (an empty bucket's item leaves infinity for the reduce), plus one per S entries of the split ones
    const size_t items_cap = pl.nbuckets + (entries_cap >> pl.logS) + 1;
    out.items_cap = items_cap;
    sc.sched.ensure((size_t)(3 + msmk::SCHED_CLASSES) * nblk * 4);
    sc.order.ensure(items_cap * 4);
    sc.item_bucket.ensure(items_cap * 4);
    // level 0 of the merge tree lists every FAN-th item of the split buckets: sum ceil(items_b / FAN) <= items / FAN + 3/4 per split bucket; a split
    // bucket has at least 3 items (more than T >= 2 S entries), so the list never exceeds items / 2.  The levels ping-pong between two lists.
    sc.merge_list.ensure((items_cap / 2 + 1) * 4);
    sc.merge_list2.ensure((items_cap / 2 + 1) * 4);
    uint32_t* blk_e = (uint32_t*)sc.sched.p;
    uint32_t* blk_i = blk_e + nblk;
    uint32_t* blk_max = blk_i + nblk;
    uint32_t* blk_cls = blk_max + nblk;
    const uint32_t packed = pl.logT | (pl.cls_shift << 8) | (pl.logS << 16);
    auto schedule = [&](auto nt_tag) {
        constexpr uint32_t NT = decltype(nt_tag)::value;
        hipLaunchKernelGGL(msmk::k_sched1<NT>, dim3(nblk), dim3(NT), 0, s, (const uint32_t*)sc.hist.p, (uint32_t)pl.nbuckets, per_blk, packed, nblk, blk_e, blk_i, blk_cls,
                           blk_max);
        hipLaunchKernelGGL(msmk::k_sched2<NT>, dim3(1), dim3(NT), 0, s, nblk, blk_e, blk_i, blk_cls, (const uint32_t*)blk_max, (uint32_t*)sc.meta.p);
        hipLaunchKernelGGL(msmk::k_sched3<NT>, dim3(nblk), dim3(NT), 0, s, (const uint32_t*)sc.hist.p, (uint32_t)pl.nbuckets, per_blk, packed, nblk, (const uint32_t*)blk_e,
                           (const uint32_t*)blk_i, (const uint32_t*)blk_cls, (uint32_t*)sc.offsets.p, (uint32_t*)sc.woff.p, (uint32_t*)sc.order.p,
                           (uint32_t*)sc.item_bucket.p, (uint32_t*)sc.merge_list.p, (uint32_t*)sc.meta.p);
    };
    if (under_accumulate) schedule(std::integral_constant<uint32_t, 512>{});
    else schedule(std::integral_constant<uint32_t, 1024>{});
    // the merge launches are sized by the schedule's counts: one small read-back into pinned memory, waited for by read_schedule
    HIP_TRY(hipMemcpyAsync(sc.h_meta, sc.meta.p, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(sc.ev[3], s));
}

void read_schedule(Scratch& sc, SortOut& out) {
    HIP_TRY(hipEventSynchronize(sc.ev[3]));
    HIP_TRY(hipGetLastError());
    out.nitems = sc.h_meta[0];
    out.max_items = sc.h_meta[1];
    out.entries = sc.h_meta[2];
    out.nlist = sc.h_meta[3];
    out.nsplit = sc.h_meta[4];
    if ((uint64_t)out.nlist > out.items_cap / 2 + 1) throw HipFail{"schedule produced a longer merge list than its bound"};
    if (out.nitems > out.items_cap) throw HipFail{"schedule produced more work items than its bound"};
}

}  // namespace mi
