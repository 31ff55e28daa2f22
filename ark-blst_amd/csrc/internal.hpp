// Functions that cross the translation units of the library (all host code; kernels never do).
//   msm_sort.hip     window-size plan, bucket sort + work-item schedule (curve independent)
//   msm_g1.hip / msm_g2.hip   the curve pipelines (explicit instantiations of msm_curve.hpp), normalize_batch
//   points.hip       bulk (de)serialisation
//   pairing_api.hip  batched Miller loop + final exponentiation
//   api.hip          the extern "C" entry points of include/arkblst_amd.h
#pragma once
#include "common.hpp"

namespace mi {

// ---- msm_sort.hip
struct CurveCost {        // what the plan needs to know about the curve's kernels (microseconds, measured; DESIGN_HISTORY.md §8)
    int log_ll;           // log2 logical lanes per reduce wave (coop scheme of k_reduce_coop)
    int comb_log_ll;      // log2 logical lanes per combine wave (k_combine may use a narrower, lower-latency scheme)
    uint32_t max_chunks;  // reduce waves that run at once (1024 SIMDs x occupancy)
    double add_per_us;    // mixed additions per microsecond of the accumulate kernel at full occupancy
    double lane_add_us;   // one lane's time per mixed addition (latency view), every wave slot taken
    double lone_lane;     // ... as a fraction of it when at most one accumulate wave runs per SIMD (few items)
    double acc_wave_items; // work items per accumulate wave (64 lanes; G2: a lane pair per item)
    double step_us;       // one complete addition of the reduce chain, every wave slot taken
    double lone_step_us;  // the same with at most one reduce wave per SIMD
    double comb_step_us;  // one complete addition of the combine chain
    double merge_us;      // one level of the split-bucket merge (fan-in MERGE_FAN, k_merge)
    uint64_t serial_buckets;  // bucket count from which the one-lane-per-64-buckets reduce is used (0 = never)
    double serial_step_us;    // one single-lane complete addition at two waves per SIMD
};
// shared = false: one bucket set per window (plain bases).  shared = true: resident 2^(c j) P tables — every window feeds ONE
// bucket set; `stride` = points per table (entry index = w * stride + i).  forced_c = 0 lets the time model choose.
// fold: see num_windows (common.hpp).
Plan make_plan(size_t n, unsigned forced_c, const CurveCost& cc, bool shared, size_t stride, bool fold = false);

struct SortOut {
    uint32_t nitems = 0, max_items = 0, nlist = 0, nsplit = 0;   // nlist: list length of merge level 0; nsplit: split buckets
    uint64_t entries = 0;
    size_t items_cap = 0;   // upper bound of nitems known before the schedule has run (sizes the accumulate launch and its output)
};
// scalars -> signed digits -> sorted (index | sign) entries, bucket offsets, work items, processing order.  Records
// sc.ev[0] .. sc.ev[3] (start, coarse done, fine done, schedule done).  Only ENQUEUES: the item counts (sc.meta on the device, copied to
// sc.h_meta before ev[3]) are read by read_schedule, which the caller runs after it has queued the accumulate kernel behind the
// schedule — the host's wait for the counts then overlaps that kernel instead of leaving the device idle (round 4).
// host_scalars != nullptr: the scalars are still in host memory.  They are copied into d_scalars (device scratch of n x 32 B) in
// chunks of whole point tiles on the lane's copy stream, and the count pass of a chunk starts when that chunk has landed
// (the trait's actual call shape: host slices, /root/reference/src/g1.rs:604,623, uploaded per call at src/gpu.rs:149-150).
// under_accumulate: the launches will run beside an accumulate kernel (pipelined call): digit passes in 256-lane workgroups.
// sc / s: the scratch set and the stream of this window group (a one-group call: d.sc[0], d.stream); pl.win0 / pl.nwin: its digit windows.
void sort_and_schedule(DevState& d, Scratch& sc, hipStream_t s, const Plan& pl, const uint32_t* d_scalars, const uint8_t* d_flags, size_t n, unsigned fmt,
                       bool shared_buckets, size_t stride, SortOut& out, const uint8_t* host_scalars = nullptr, bool under_accumulate = false);
void read_schedule(Scratch& sc, SortOut& out);   // waits for sc.ev[3], fills nitems / max_items / entries / nlist
// window groups of a pipelined call, top windows first ({pl} = one group: no pipelining)
std::vector<Plan> split_plan(const Plan& pl, const CurveCost& cc, size_t n, bool shared, const std::vector<unsigned>& weights);

// ---- msm_g1.hip / msm_g2.hip
int g1_set_bases(mi_ctx* ctx, const void* bases, size_t n, unsigned precompute_c);
int g2_set_bases(mi_ctx* ctx, const void* bases, size_t n, unsigned precompute_c);
int g1_set_bases_device(mi_ctx* ctx, const void* d_bases, size_t n);
int g2_set_bases_device(mi_ctx* ctx, const void* d_bases, size_t n);
int g1_set_bases_from_jacobian(mi_ctx* ctx, const void* jac, size_t n);
int g2_set_bases_from_jacobian(mi_ctx* ctx, const void* jac, size_t n);
// device slot k's shard of a new resident set from decoded points in device memory (points.hip, set_bases_from_compressed); may throw HipFail
void g1_install_resident(mi_ctx* ctx, size_t k, const void* d_affine, size_t lo, size_t n, bool validated);
void g2_install_resident(mi_ctx* ctx, size_t k, const void* d_affine, size_t lo, size_t n, bool validated);
int g1_msm(mi_ctx* ctx, const void* bases, const uint8_t* scalars, bool scalars_on_device, size_t n, unsigned fmt, void* out);
int g2_msm(mi_ctx* ctx, const void* bases, const uint8_t* scalars, bool scalars_on_device, size_t n, unsigned fmt, void* out);
int g1_msm_windows(mi_ctx* ctx, const uint8_t* d_scalars, size_t n, unsigned fmt, void* d_out, mi_window_info* info);
int g2_msm_windows(mi_ctx* ctx, const uint8_t* d_scalars, size_t n, unsigned fmt, void* d_out, mi_window_info* info);
int g1_msm_batch(mi_ctx* ctx, const uint8_t* const* scalars, bool scalars_on_device, size_t k, size_t n, unsigned fmt, mi_g1* out);
int g2_msm_batch(mi_ctx* ctx, const uint8_t* const* scalars, bool scalars_on_device, size_t k, size_t n, unsigned fmt, mi_g2* out);
int g1_normalize(mi_ctx* ctx, const mi_g1* in, bool on_device, size_t n, mi_g1_affine* out);
CurveCost g1_cost();
CurveCost g2_cost();
int g2_normalize(mi_ctx* ctx, const mi_g2* in, bool on_device, size_t n, mi_g2_affine* out);

// ---- points.hip
int g1_deserialize(mi_ctx* ctx, const uint8_t* bytes, bool on_device, size_t n, int compressed, int validate, mi_g1_affine* out, uint8_t* status);
int g1_serialize(mi_ctx* ctx, const mi_g1_affine* points, size_t n, int compressed, uint8_t* bytes);
int g2_deserialize(mi_ctx* ctx, const uint8_t* bytes, bool on_device, size_t n, int compressed, int validate, mi_g2_affine* out, uint8_t* status);
int g2_serialize(mi_ctx* ctx, const mi_g2_affine* points, size_t n, int compressed, uint8_t* bytes);
int g1_check_batch(mi_ctx* ctx, const mi_g1_affine* points, bool on_device, size_t n, uint8_t* status);
int g2_check_batch(mi_ctx* ctx, const mi_g2_affine* points, bool on_device, size_t n, uint8_t* status);
int g1_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected);
int g2_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected);
int g1_validate_bases(mi_ctx* ctx, size_t* n_invalid);
int g2_validate_bases(mi_ctx* ctx, size_t* n_invalid);
#if defined(MI_TEST_HOOKS)
int test_fp_op(mi_ctx* ctx, int op, const mi_fp* a, const mi_fp* b, mi_fp* out, size_t n);
#endif

// ---- pairing_api.hip
int miller(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out, bool final_exp);
int final_exponentiation(const mi_fp12* f, mi_fp12* out);

}  // namespace mi
