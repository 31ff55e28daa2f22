// Host-side state shared by the translation units of the C-ABI library (see include/arkblst_amd.h).
//
// One mi_ctx owns, per device: streams, a resident base set in device form, and reusable scratch (histogram / offsets /
// sorted indices / buckets / chunk sums) sized for the largest call seen so far — the reference rebuilds its program
// and re-allocates every buffer on every call (/root/reference/src/gpu.rs:148-156,235).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arkblst_amd.h"
#include "host_curve.hpp"

namespace mi {

struct HipFail {
    std::string msg;
    bool oom = false;
    bool invalid = false;   // the caller's argument is at fault (MI_E_INVALID), not the runtime
    bool aborted = false;   // the caller's abort check said stop (MI_E_ABORTED)
};
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            char _b[512];                                                                                 \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw ::mi::HipFail{_b, _e == hipErrorOutOfMemory};                                           \
        }                                                                                                 \
    } while (0)

// Test builds (-DMI_TEST_HOOKS, libarkblst_amd_test.so) can make the next device allocations fail, to prove that an
// allocation failure anywhere (worker threads included) comes back as MI_E_NOMEM instead of an abort.
#if defined(MI_TEST_HOOKS)
extern std::atomic<int> g_fail_allocs;   // > 0: that many upcoming DevBuf::ensure calls that need to grow will throw
#endif

// Digit windows of a scalar below r < 2^255 recoded into signed c-bit digits.  fold = the digit kernels recode min(s, r - s) < 2^254 with
// the sign folded into the digits (exact only on the prime-order subgroup: resident bases that passed mi_msm_g{1,2}_validate_bases):
// ceil(255 / c) windows always suffice.  Otherwise the integer s itself is recoded and a top window of a full c bits can carry out:
// one more window when c divides 255 (c = 15, 17).
inline uint32_t num_windows(unsigned c, bool fold) { return (255 + c - 1) / c + ((!fold && 255 % c == 0) ? 1u : 0u); }

struct Plan {
    uint32_t c, nwin;     // window bits; digit windows of this plan: num_windows(c, fold) for a whole call, fewer for one window group of it
    uint32_t win0;        // first digit window (0 for a whole call; a group of a pipelined call covers [win0, win0 + nwin))
    bool fold;            // sign fold (see num_windows): travels to the digit kernels as bit 1 of their fmt argument
    uint32_t bwin;        // bucket sets: nwin, or 1 when all windows share one (precomputed tables)
    uint32_t nb, coop_L, chunks_per_win, logT, lo_bits;   // coop_L: buckets per logical lane of k_reduce_coop (any value 1..64)
    uint32_t logS;        // log2 of the item size of buckets that hold more than T = 2^logT entries (<= logT)
    uint32_t cls_shift;   // log2 of the width of a length class of the schedule (follows the typical item, not T)
    uint64_t nbuckets, nchunks;
    uint32_t chunk_buckets;   // buckets one reduce wave (or, serial form, one reduce lane) covers; the last chunk of a window may be ragged
    bool serial_reduce;   // throughput form: one lane per serial_L buckets (k_reduce_serial)
    uint32_t serial_L;    // buckets per lane of the serial form (<= 64, any value: chosen so that the lanes fill one round of wave slots)
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
#if defined(MI_TEST_HOOKS)
        if (g_fail_allocs.load() > 0 && g_fail_allocs.fetch_sub(1) > 0) throw HipFail{"injected allocation failure (test hook): out of memory", true};
#endif
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
    // ensure(), but an allocation more than twice the need (and > 64 MiB larger) is given back first: a plain set_bases after
    // set_bases_precomputed must not keep the W x table allocation for the life of the context
    void ensure_fit(size_t bytes) {
        if (cap > 2 * bytes && cap - bytes > ((size_t)64 << 20)) release();
        ensure(bytes);
    }
};

// resident bases (device form) of one device, per group: [0] = G1, [1] = G2.  Shared by the context's lanes.
struct Resident {
    DevBuf buf, flags;   // device-form points; one byte per point: 1 = point at infinity
    size_t n = 0;        // points resident on this device
    size_t lo = 0;       // global index of the first resident point
    uint32_t tables = 1; // 1 = plain bases; W > 1 = precomputed 2^(c j) P_i tables, j < W (see mi_msm_g1_set_bases_precomputed)
    uint32_t table_c = 0;// window size the tables were built for
    bool validated = false;   // every point passed Valid::check on the GPU (mi_msm_g{1,2}_validate_bases): MSMs over this set may fold signs
};

// 128-bit fingerprint of EVERY byte of a host base vector (content_fingerprint below)
struct Fp128 {
    uint64_t a = 0, b = 0;
    bool operator==(const Fp128& o) const { return a == o.a && b == o.b; }
    bool operator!=(const Fp128& o) const { return !(*this == o); }
};

// One cached base set of the stateless call shape (mi_msm_set_base_cache): the device-form shards of a host base vector the context has
// seen, keyed by (host pointer, length, fingerprint of its whole content).  Lanes hold a shared_ptr while they use it, so an
// eviction by the other lane frees the memory only when the last user is done.
struct BaseCacheEntry {
    const void* ptr = nullptr;
    size_t n = 0;
    Fp128 fp;
    uint64_t stamp = 0;             // LRU clock
    std::vector<Resident> shard;    // one per device of the context
    std::vector<int> devs;          // their HIP ordinals (the buffers are freed on the right device)
    ~BaseCacheEntry() {
        for (size_t k = 0; k < shard.size(); k++) {
            if (!shard[k].buf.p && !shard[k].flags.p) continue;
            (void)hipSetDevice(devs[k]);
            shard[k].buf.release();
            shard[k].flags.release();
        }
    }
};

// Scratch of ONE WINDOW GROUP of an MSM call: everything the sort, the schedule, the accumulate kernel and the reduction of a set of
// digit windows write.  A call is either one group (all windows, one stream: small inputs, shared bucket sets) or up to MAX_GROUPS
// groups whose phases overlap on the lane's streams (run_msm, msm_curve.hpp): sort(g + 1) and reduce(g - 1) run under accumulate(g).
constexpr int MAX_GROUPS = 4;
struct Scratch {
    DevBuf hist, offsets, woff, meta, sched, sorted, partial, order, item_bucket, pairs, pairs2;
    DevBuf tilecnt, tileoff, bin_tot, bin_base, binA_base, coarse, coarseA, seg_cnt, seg_base, segcnt, segoff, merge_list, merge_list2;
    uint32_t* h_meta = nullptr;   // pinned, 32 B: the schedule's item counts
    // [0] sort start, [1] coarse partition done, [2] fine sort done, [3] schedule done (always recorded: read_schedule waits on it and the
    // accumulate stream of a pipelined call does), [4] accumulate done, [5] reduce done, [6] combine done, [7] window sums copied out
    hipEvent_t ev[8] = {};
    template <class Fn> void for_each_buf(Fn fn) {
        for (DevBuf* b : {&hist, &offsets, &woff, &meta, &sched, &sorted, &partial, &order, &item_bucket, &pairs, &pairs2, &tilecnt, &tileoff, &bin_tot,
                          &bin_base, &binA_base, &coarse, &coarseA, &seg_cnt, &seg_base, &segcnt, &segoff, &merge_list, &merge_list2})
            fn(*b);
    }
};

// Per-device state of ONE LANE of a context: streams, events and scratch.  A context has two lanes per device so that two
// host threads (arkworks calls the trait method from rayon workers) overlap: one call's sort / reduce / host tail runs
// under the other's accumulate kernel.  The resident bases are shared.
struct DevState {
    int dev = 0;
    uint32_t simds = 1024;   // 4 per compute unit (256 CUs on MI355X)
    hipStream_t stream = nullptr;        // everything of a one-group call; accumulate kernels of the even groups of a pipelined call
    hipStream_t copy_stream = nullptr;   // chunked H2D of host-slice calls, overlapped with the kernels that consume the chunks
    // pipelined calls only, created on the lane's first one (ensure_pipeline_streams): the accumulate kernels of the odd groups, and the
    // sorts + reductions of every group (high priority: short latency-bound launches that must find wave slots under the accumulate kernels)
    hipStream_t acc2_stream = nullptr, aux_stream = nullptr;
    hipStream_t d2h_stream = nullptr;    // rows (f): results of chunk j leave while chunk j + 1 is computed (created on first use, ensure_d2h_stream)
    hipEvent_t ev[12] = {};
    hipEvent_t cev[10] = {};             // copy stream: [0] first copy issued, [1..8] chunk landed, [9] last copy done
    hipEvent_t iev[4][8] = {};           // rows (f), per chunk: [0] kernels start, [1] kernels done, [2] second-phase start, [3] second-phase done
    Resident* res = nullptr;   // -> mi_ctx::residents[device][2]
    // scratch
    DevBuf raw, call_bases, call_flags, scalars;
    Scratch sc[MAX_GROUPS];
    DevBuf pr_p, pr_q, pr_lvl[2], pr_raw, pr_lines;   // pairing: inputs, tree levels, top values, line coefficients
    DevBuf io_in, io_out, io_status, nv_vals, nv_pref, nv_inv, nv_top;   // rows (f): staging of normalize / (de)serialize / check, kept across calls
    void* h_pairs = nullptr;   // pinned host staging: window sums (D2H), pairing top values
    size_t h_pairs_cap = 0;
    mi_profile prof{};
    int prof_level = 1;   // copied from the context at the start of a call (mi_msm_set_profile_level): which events the pipeline records

    void ensure_host(size_t bytes) {
        if (bytes <= h_pairs_cap) return;
        if (h_pairs) (void)hipHostFree(h_pairs);
        h_pairs = nullptr;
        h_pairs_cap = 0;
        HIP_TRY(hipHostMalloc(&h_pairs, bytes, hipHostMallocDefault));
        h_pairs_cap = bytes;
    }
    // Stream priorities decide which HARDWARE QUEUE a stream gets: the HIP runtime keeps one pool of queues per priority level
    // (GPU_MAX_HW_QUEUES = 4 each) and streams beyond a pool's size SHARE a queue, which serialises their kernels.  Measured (round 6): with
    // the second accumulate stream as the sixth normal-priority stream of the process it shared a queue with the lane's own stream whenever
    // the runtime's bookkeeping fell that way — the pipelined bench.py step took 3.62 ms against 3.27 unpipelined while tools/pipe_scan.py,
    // one stream fewer, showed a gain; on the LOW-priority pool its kernel starved behind the first group's (3.64 against 3.31).  So the
    // NORMAL pool is kept for the streams whose kernels must overlap — the process's null stream, the two lanes' streams and their second
    // accumulate streams (the fifth, lane 1's, exists only once two callers pipeline at the same time) — and everything short and latency-bound
    // goes to the HIGH pool: copies (mi_msm_init creates the copy streams there), the sorts of a pipelined call, the D2H stream of rows (f).
    static int high_priority() {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = greatest priority (numerically lower)
        return hi;
    }
    void ensure_pipeline_streams() {
        if (aux_stream) return;
        HIP_TRY(hipStreamCreateWithFlags(&acc2_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, high_priority()));
    }
    void ensure_d2h_stream() {
        if (!d2h_stream) HIP_TRY(hipStreamCreateWithPriority(&d2h_stream, hipStreamNonBlocking, high_priority()));
    }
    template <class Fn> void for_each_buf(Fn fn) {
        for (DevBuf* b : {&raw, &call_bases, &call_flags, &scalars, &pr_p, &pr_q, &pr_lvl[0], &pr_lvl[1], &pr_raw, &pr_lines, &io_in, &io_out, &io_status,
                          &nv_vals, &nv_pref, &nv_inv, &nv_top})
            fn(*b);
        for (Scratch& s : sc) s.for_each_buf(fn);
    }
};

constexpr int NLANES = 2;
class HashPool;   // defined below mi_ctx

// One persistent host thread per device of a multi-device context: the calling thread posts one job per device and
// waits; nothing is spawned per call (a std::thread per device per call cost 1.4 ms for 0.2 ms of work).
class DeviceWorkers {
public:
    explicit DeviceWorkers(size_t n) : slots_(n) {
        for (size_t k = 0; k < n; k++) threads_.emplace_back([this, k] { loop(k); });
    }
    ~DeviceWorkers() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    DeviceWorkers(const DeviceWorkers&) = delete;
    DeviceWorkers& operator=(const DeviceWorkers&) = delete;
    size_t size() const { return threads_.size(); }
    // Runs fn(k) on worker k for every k and returns when all are done.  fn must not throw (callers wrap it).
    // One run at a time per worker set (callers hold a lane of the context).
    void run(const std::function<void(size_t)>& fn) {
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        pending_ = slots_.size();
        for (auto& s : slots_) s = true;
        cv_.notify_all();
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop(size_t k) {
        for (;;) {
            const std::function<void(size_t)>* fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || slots_[k]; });
                if (stop_) return;
                slots_[k] = false;
                fn = fn_;
            }
            (*fn)(k);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::vector<char> slots_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t pending_ = 0;
    bool stop_ = false;
};

}  // namespace mi

struct mi_ctx {
    std::vector<mi::DevState> devs;                              // lane 0 (also used by every non-MSM entry point)
    std::vector<mi::DevState> devs_b;                            // lane 1
    std::vector<std::array<mi::Resident, 2>> residents;          // per device
    std::unique_ptr<mi::DeviceWorkers> workers[mi::NLANES];      // multi-device contexts only: one thread per device and lane
    std::unique_ptr<mi::DeviceWorkers> batch_workers;            // mi_msm_*_batch: two persistent job pullers, created on first use
    std::mutex batch_mu;                                         // one batch call at a time drives them
    std::mutex lane_mu;                                          // lane bookkeeping
    std::condition_variable lane_cv;
    bool lane_busy[mi::NLANES] = {false, false};
    mutable std::mutex info_mu;                                  // prof / err
    unsigned forced_c = 0;
    // window groups of a pipelined call (run_msm): 0 entries = the built-in choice; {1} = never pipeline; otherwise relative weights of the
    // groups, top windows first (mi_msm_set_pipeline / ARKBLST_AMD_PIPELINE)
    std::vector<unsigned> pipe_weights;
    // the caller's abort check (mi_msm_set_abort_check): the reference driver's maybe_abort, src/gpu.rs:58,133-137
    int (*abort_check)(void*) = nullptr;
    void* abort_user = nullptr;
    bool pipe_two_acc_streams = false;   // ARKBLST_AMD_PIPELINE_ACC2=1: accumulate kernels of odd groups on a second normal-priority stream (see DevState::ensure_pipeline_streams)
    bool trace = false;   // ARKBLST_AMD_TRACE=1: MSM calls at profile level 2 print their phase boundaries to stderr (run_msm)
    // base-set cache of the stateless call shape (api.hip mi_msm_set_base_cache); [0] = G1, [1] = G2
    std::mutex cache_mu;
    unsigned cache_entries = 0;       // 0 = off
    bool cache_env = false;           // ARKBLST_AMD_BASE_CACHE was set: it overrides mi_msm_set_base_cache
    uint64_t cache_clock = 0, cache_hits = 0, cache_misses = 0;
    std::vector<std::shared_ptr<mi::BaseCacheEntry>> cache[2];
    uint64_t hash_seed = 0, hash_mult = 0xD6E8FEB86659FD93ull;   // per-context key of the fingerprint (mi_msm_init: std::random_device)
    std::shared_ptr<mi::HashPool> hash_pool[mi::NLANES];   // fingerprint helpers of the base-set cache, one pool per lane, created on first use
    int profile_level = 1;   // 0: no timing events beyond the one the pipeline waits on; 1: + the accumulate kernel's interval; 2: every phase
    mi_profile prof{};
    mi_pairing_profile pprof{};
    std::string err;
#if defined(MI_TEST_HOOKS)
    unsigned test_pairing_share = 0, test_pairing_batch = 0;
    bool test_pairing_single_lane = false;
    size_t test_max_part = 0;   // points per pass of the pipeline (0 = the built-in limit)
    bool test_no_peer = false;  // pretend no device can read another's memory (exercises the staging path on one GPU)
#endif
};

namespace mi {

// Fingerprint of a host base vector: EVERY byte (round 6; rounds 4-5 sampled 1024 points, and an in-place edit of a point outside the
// sample was a silent hit on stale device data).  The vector is cut into slices hashed in parallel (HashPool, below); a slice is four
// independent 64-bit chains h <- xorshift((h ^ w) * K) over interleaved 8-byte words.  Every step is a bijection of the chain's state
// for a fixed word and of the word for a fixed state, so a change confined to ONE 8-byte word (one limb of one coordinate) always changes
// the chain it feeds, and the folds below are bijections of each state they take in: such an edit is detected with certainty, any other
// edit with probability 1 - 2^-64 or better.  ~10 GB/s per core (memory-bound): 96 MiB (2^20 G1 points) on five helper threads in ~2 ms,
// under the ~3 ms of GPU work the call queues first.
// The chains start from, and multiply by, values drawn per context from std::random_device (HashKey): the fingerprint is not a
// cryptographic hash, but whoever fills a base vector cannot aim for a collision without knowing them.
struct HashKey { uint64_t seed = 0, mult = 0xD6E8FEB86659FD93ull; };   // mult is odd
struct SliceHash { uint64_t h[4]; };
inline SliceHash hash_slice(const uint8_t* p, size_t bytes, uint64_t slice, const HashKey& key) {
    const uint64_t K = key.mult | 1u, seed = key.seed + slice * 0x9E3779B97F4A7C15ull;
    SliceHash s{{0x9E3779B97F4A7C15ull ^ seed, 0xC2B2AE3D27D4EB4Full + seed, 0x165667B19E3779F9ull ^ (seed << 1), 0x27D4EB2F165667C5ull + (seed << 2)}};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        for (int t = 0; t < 4; t++) {
            s.h[t] = (s.h[t] ^ w[t]) * K;
            s.h[t] ^= s.h[t] >> 32;
        }
    }
    if (i < bytes) {   // tail of under 32 bytes (never for point vectors: 96 / 192 B per point), zero-padded, its length mixed in
        uint64_t w[4] = {0, 0, 0, 0};
        memcpy(w, p + i, bytes - i);
        for (int t = 0; t < 4; t++) {
            s.h[t] = (s.h[t] ^ w[t] ^ (uint64_t)(bytes - i)) * K;
            s.h[t] ^= s.h[t] >> 32;
        }
    }
    return s;
}
// slices folded in order: two different bijective folds of the same states give the two halves of the fingerprint
inline Fp128 fold_slices(const SliceHash* s, size_t count, uint64_t total_bytes) {
    Fp128 r{0x243F6A8885A308D3ull ^ total_bytes, 0x13198A2E03707344ull + total_bytes};
    for (size_t k = 0; k < count; k++)
        for (int t = 0; t < 4; t++) {
            r.a = (r.a ^ s[k].h[t]) * 0xD6E8FEB86659FD93ull;
            r.a ^= r.a >> 29;
            r.b = (r.b + s[k].h[t]) * 0x9FB21C651E98DF25ull;
            r.b ^= r.b >> 31;
        }
    return r;
}
constexpr size_t HASH_SLICE_MIN = (size_t)1 << 20;   // a slice is at least 1 MiB (smaller vectors are one slice)
inline size_t hash_slices(size_t bytes, size_t threads) { return std::max<size_t>(1, std::min(threads, bytes / HASH_SLICE_MIN)); }
// the fingerprint computed on the calling thread (what HashPool computes in parallel: same slicing, same value)
inline Fp128 content_fingerprint(const uint8_t* p, size_t bytes, size_t threads, const HashKey& key = HashKey{}) {
    const size_t S = hash_slices(bytes, threads);
    std::vector<SliceHash> part(S);
    for (size_t k = 0; k < S; k++) {
        const size_t lo = (bytes / 32 * k / S) * 32, hi = k + 1 == S ? bytes : (bytes / 32 * (k + 1) / S) * 32;
        part[k] = hash_slice(p + lo, hi - lo, (uint64_t)k, key);
    }
    return fold_slices(part.data(), S, bytes);
}

// Persistent helper threads that fingerprint a host base vector WHILE the calling thread queues the GPU work of the call (one pool per
// lane of the context, created on the lane's first cached call; nothing is spawned per call — rounds 4-5 used std::async).  One job at a
// time: start() hands out the job, finish() waits for it.  The threads only READ the caller's bases; finish() is called on every path
// out of the MSM call, so none of them touches the vector after the call returned.
class HashPool {
public:
    explicit HashPool(size_t threads, const HashKey& key = HashKey{}) : key_(key), part_(std::max<size_t>(1, threads)) {
        for (size_t k = 0; k < part_.size(); k++) threads_.emplace_back([this, k] { loop(k); });
    }
    ~HashPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    HashPool(const HashPool&) = delete;
    HashPool& operator=(const HashPool&) = delete;
    size_t threads() const { return part_.size(); }
    bool busy() const { return busy_; }
    void start(const uint8_t* p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu_);
        p_ = p; bytes_ = bytes;
        slices_ = hash_slices(bytes, part_.size());
        pending_ = slices_;
        gen_++;
        busy_ = true;
        cv_.notify_all();
    }
    Fp128 finish() {   // start() must have been called; idempotent until the next start()
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        busy_ = false;
        return fold_slices(part_.data(), slices_, bytes_);
    }

private:
    void loop(size_t k) {
        uint64_t seen = 0;
        for (;;) {
            const uint8_t* p;
            size_t bytes, S;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                p = p_; bytes = bytes_; S = slices_;
            }
            if (k >= S) continue;   // fewer slices than threads: nothing for this one in this job
            const size_t lo = (bytes / 32 * k / S) * 32, hi = k + 1 == S ? bytes : (bytes / 32 * (k + 1) / S) * 32;
            const SliceHash h = hash_slice(p + lo, hi - lo, (uint64_t)k, key_);
            {
                std::lock_guard<std::mutex> lk(mu_);
                part_[k] = h;
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    const HashKey key_;
    std::vector<SliceHash> part_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const uint8_t* p_ = nullptr;
    size_t bytes_ = 0, slices_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false, busy_ = false;
};
// helper threads per pool: a third of the host's hardware threads, 1..6 (ARKBLST_AMD_HASH_THREADS overrides, 1..16)
inline size_t hash_pool_threads() {
    if (const char* e = getenv("ARKBLST_AMD_HASH_THREADS")) return (size_t)std::min(16l, std::max(1l, atol(e)));
    return std::min<size_t>(6, std::max<size_t>(1, std::thread::hardware_concurrency() / 3));
}

// ---- bookkeeping of the base-set cache (mi_ctx::cache, guarded by cache_mu).  Host-only and free of HIP calls, so that
// tests/host/workers_test.cpp can run it under ThreadSanitizer / AddressSanitizer.
// cache_begin: false when the cache is off; otherwise `candidate` = the most recently used entry with the same (pointer, n), if any —
// the caller uses it speculatively while the fingerprint is computed, then confirms it.
inline bool cache_begin(mi_ctx* ctx, int idx, const void* ptr, size_t n, std::shared_ptr<BaseCacheEntry>& candidate) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    if (!ctx->cache_entries) return false;
    for (auto& e : ctx->cache[idx])
        if (e->ptr == ptr && e->n == n && (!candidate || e->stamp > candidate->stamp)) candidate = e;
    return true;
}
inline std::shared_ptr<BaseCacheEntry> cache_find(mi_ctx* ctx, int idx, const void* ptr, size_t n, const Fp128& fp) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    for (auto& e : ctx->cache[idx])
        if (e->ptr == ptr && e->n == n && e->fp == fp) return e;
    return nullptr;
}
// cache_finish: a confirmed hit is stamped; a filled entry gets its key and is published (unless the other lane published the same
// key first, or the cache was switched off meanwhile), the least recently used entries beyond the limit leave the list — their
// memory is freed when the last lane that still reads them drops its reference.
inline void cache_finish(mi_ctx* ctx, int idx, const std::shared_ptr<BaseCacheEntry>& hit, const std::shared_ptr<BaseCacheEntry>& fill, const Fp128& fp) {
    std::vector<std::shared_ptr<BaseCacheEntry>> dropped;   // destroyed after the lock is released
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    auto& v = ctx->cache[idx];
    if (hit) {
        hit->stamp = ++ctx->cache_clock;
        ctx->cache_hits++;
        return;
    }
    ctx->cache_misses++;
    if (!fill) return;
    fill->fp = fp;
    bool dup = false;
    for (auto& e : v) dup = dup || (e->ptr == fill->ptr && e->n == fill->n && e->fp == fill->fp);
    if (dup || !ctx->cache_entries) return;
    fill->stamp = ++ctx->cache_clock;
    v.push_back(fill);
    while (v.size() > ctx->cache_entries) {
        size_t old = 0;
        for (size_t q = 1; q < v.size(); q++)
            if (v[q]->stamp < v[old]->stamp) old = q;
        dropped.push_back(v[old]);
        v.erase(v.begin() + (long)old);
    }
}

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
    DeviceWorkers* workers() { return c->workers[lane == 1 ? 1 : 0].get(); }
};

// true: the caller asked to stop (mi_msm_set_abort_check)
inline bool abort_requested(const mi_ctx* ctx) { return ctx->abort_check && ctx->abort_check(ctx->abort_user) != 0; }

inline void set_prof(mi_ctx* ctx, const mi_profile& p) {
    std::lock_guard<std::mutex> lk(ctx->info_mu);
    ctx->prof = p;
}

inline double ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    return ms;
}

// The text of the calling thread's most recent failure: mi_msm_last_error() hands out a pointer that stays valid until
// the same thread fails again, whatever other threads do on the context (two lanes can fail concurrently).
inline std::string& tls_error() {
    static thread_local std::string e;
    return e;
}

inline int fail(mi_ctx* ctx, int code, const std::string& msg) {
    tls_error() = msg;
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->info_mu);
        ctx->err = msg;
    }
    return code;
}

// The single-device entry points run on the caller's thread and select the context's device there: the caller's current device
// (torch's, say) is put back when the call returns.
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() {
        if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); }
    }
    ~DeviceRestore() {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
    DeviceRestore(const DeviceRestore&) = delete;
    DeviceRestore& operator=(const DeviceRestore&) = delete;
};

// Nothing may cross the C ABI as an exception (include/arkblst_amd.h): every entry point body runs inside guarded().
template <class Fn>
int guarded(mi_ctx* ctx, Fn fn) {
    DeviceRestore restore;
    try {
        return fn();
    } catch (const HipFail& e) {
        bool oom = e.oom || e.msg.find("out of memory") != std::string::npos;
        return fail(ctx, e.aborted ? MI_E_ABORTED : e.invalid ? MI_E_INVALID : oom ? MI_E_NOMEM : MI_E_HIP, e.msg);
    } catch (const std::bad_alloc&) {
        return fail(ctx, MI_E_NOMEM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(ctx, MI_E_HIP, std::string("unexpected exception: ") + e.what());
    } catch (...) {
        return fail(ctx, MI_E_HIP, "unexpected exception");
    }
}

// Per-device share of a call: what a worker reports back instead of throwing (a throw inside a std::thread is std::terminate).
struct PartErr {
    int code = MI_OK;
    std::string msg;
};
template <class Fn>
void guarded_part(PartErr& e, Fn fn) noexcept {
    try {
        fn();
    } catch (const HipFail& f) {
        e.code = f.aborted ? MI_E_ABORTED : f.invalid ? MI_E_INVALID : (f.oom || f.msg.find("out of memory") != std::string::npos) ? MI_E_NOMEM : MI_E_HIP;
        e.msg = f.msg;
    } catch (const std::bad_alloc&) {
        e.code = MI_E_NOMEM;
        e.msg = "host allocation failed";
    } catch (const std::exception& x) {
        e.code = MI_E_HIP;
        e.msg = std::string("unexpected exception: ") + x.what();
    } catch (...) {
        e.code = MI_E_HIP;
        e.msg = "unexpected exception";
    }
}

// fn(k) for every device k of the lane: inline for one device, on the lane's persistent workers otherwise
template <class Fn>
void for_each_device(LaneLock& lane, size_t g, Fn fn) {
    DeviceWorkers* w = lane.workers();
    if (g == 1 || !w) {
        for (size_t k = 0; k < g; k++) fn(k);
    } else {
        std::function<void(size_t)> f = fn;
        w->run(f);
    }
}

// Device of a caller-supplied device pointer.  A pointer this runtime does not know — plain host memory, or memory of a second HIP
// runtime loaded into the process (load order) — is the caller's error (MI_E_INVALID), not an opaque fault later.
inline int device_of_ptr(const void* p, const char* what, size_t align = 16) {
    // the kernels read scalars and write window sums as 16-byte vectors (points: 4-byte words): a misaligned device pointer would fault on the GPU
    if (reinterpret_cast<uintptr_t>(p) & (align - 1))
        throw HipFail{std::string(what) + " must be " + std::to_string(align) + "-byte aligned", false, true};
    hipPointerAttribute_t a{};
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess || a.type == hipMemoryTypeUnregistered)
        throw HipFail{std::string(what) + " is not device memory known to this HIP runtime (a host pointer, or memory allocated through a "
                      "second HIP runtime in this process: see INTEGRATION.md, load order)", false, true};
    return a.device;
}
// contiguous shard [lo, hi) of n items for device k of g
inline void shard_range(size_t n, size_t g, size_t k, size_t& lo, size_t& hi) {
    size_t per = (n + g - 1) / g;
    lo = std::min(n, k * per);
    hi = std::min(n, lo + per);
}

}  // namespace mi
