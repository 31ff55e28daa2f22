// extern "C" entry points of include/arkblst_amd.h: context lifetime, argument checks, dispatch into the curve / pairing /
// codec translation units.  Nothing throws across this boundary (every body below either cannot throw or runs in guarded()).
#include <random>
#include "internal.hpp"
#if defined(MI_TEST_HOOKS)
#include "test_hooks.h"
namespace mi { std::atomic<int> g_fail_allocs{0}; }
#endif

using namespace mi;

namespace {
// The host tail (Horner fold, partial-sum fold, final exponentiation: the functions of host_curve.hpp) is compiled for BMI2 + ADX
// (Intel since Broadwell, AMD since Zen): on anything older the entry points that run it return MI_E_UNSUPPORTED instead of
// faulting.  This function and everything else outside host_curve.hpp is baseline x86-64 code.
bool cpu_ok() {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
    return ok;
#else
    return true;
#endif
}

// Host fold of gathered window sums: out = sum_w 2^(c w) sum_r windows[r * rank_stride + w] (Horner over the windows; the
// doubling chain is the one thing the GPU cannot do in time, DESIGN_HISTORY.md §2.6).  Deterministic: ranks are added in index order.
template <class J, class Raw>
int fold_windows(const Raw* windows, size_t n_ranks, size_t rank_stride, const mi_window_info* info, Raw* out) {
    if (!out || !info || (n_ranks && info->num_windows && !windows)) return MI_E_INVALID;
    if (info->num_windows > MI_MAX_WINDOWS || rank_stride < info->num_windows ||
        (info->num_windows && (info->window_bits < 1 || info->window_bits > 32)))
        return MI_E_INVALID;
    if (!cpu_ok()) return MI_E_UNSUPPORTED;
    static_assert(sizeof(J) == sizeof(Raw), "host Jacobian type must be the reference's layout");
    J r = J::inf();
    for (int w = (int)info->num_windows - 1; w >= 0; w--) {
        r = r.dbl_n(info->window_bits);
        for (size_t k = 0; k < n_ranks; k++) {
            J p;
            memcpy(&p, &windows[k * rank_stride + (size_t)w], sizeof p);
            r = r.add(p);
        }
    }
    memcpy(out, &r, sizeof r);
    return MI_OK;
}
}  // namespace

extern "C" {

int mi_msm_init(mi_ctx** out, const int* device_ids, int n_devices) {
    if (!out || n_devices < 0) return MI_E_INVALID;
    *out = nullptr;
    if (!cpu_ok()) return MI_E_UNSUPPORTED;
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
                int cus = 0;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, id) == hipSuccess && cus > 0) d->simds = 4u * (uint32_t)cus;
                HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
                HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
                for (auto& e : d->ev) HIP_TRY(hipEventCreate(&e));
                for (auto& e : d->cev) HIP_TRY(hipEventCreate(&e));
                for (Scratch& sc : d->sc)
                    for (auto& e : sc.ev) HIP_TRY(hipEventCreate(&e));
                for (auto& row : d->iev)
                    for (auto& e : row) HIP_TRY(hipEventCreate(&e));
            }
        }
        if (const char* e = getenv("ARKBLST_AMD_TRACE")) ctx->trace = atoi(e) != 0;
        if (const char* e = getenv("ARKBLST_AMD_PIPELINE_ACC2")) ctx->pipe_two_acc_streams = atoi(e) != 0;
        if (const char* e = getenv("ARKBLST_AMD_PIPELINE")) {   // window groups of a pipelined call (mi_msm_set_pipeline): "0" / "off", "auto", or weights "3,5,5,3"
            std::vector<unsigned> w;
            if (!strcmp(e, "0") || !strcmp(e, "off") || !strcmp(e, "1")) w = {1};
            else if (strcmp(e, "auto"))
                for (const char* q = e; *q;) {
                    char* end = nullptr;
                    unsigned long v = strtoul(q, &end, 10);
                    if (end == q) break;
                    w.push_back((unsigned)std::min<unsigned long>(std::max<unsigned long>(v, 1), 1000));
                    q = *end == ',' ? end + 1 : end;
                    if (*end && *end != ',') break;
                }
            if (w.size() > (size_t)MAX_GROUPS) w.resize(MAX_GROUPS);
            ctx->pipe_weights = w;
        }
        {   // key of the base-set cache's fingerprint (common.hpp HashKey): per context, from the system's entropy source
            std::random_device rd;
            ctx->hash_seed = ((uint64_t)rd() << 32) | rd();
            ctx->hash_mult = (((uint64_t)rd() << 32) | rd()) | 0x8000000000000001ull;   // odd, top bit set
        }
        if (const char* e = getenv("ARKBLST_AMD_BASE_CACHE")) {   // the operator's switch: overrides mi_msm_set_base_cache
            long v = strtol(e, nullptr, 10);
            ctx->cache_entries = (unsigned)std::min<long>(std::max<long>(v, 0), MI_BASE_CACHE_MAX);
            ctx->cache_env = true;
        }
        if (n_devices > 1) {
            // one persistent host thread per device and lane (nothing is spawned per call); peer access so that a device can read
            // its shard of a scalar vector that lives on another device of the context (xGMI)
            for (int l = 0; l < NLANES; l++) ctx->workers[l].reset(new DeviceWorkers((size_t)n_devices));
            // Peer access is ENABLED where the runtime offers it, so that the one hipMemcpyPeerAsync that stages a remote shard of a
            // device-resident scalar vector (msm_impl) is a direct xGMI DMA; where it is not offered the runtime stages that copy
            // through the host by itself.  No kernel of this library reads remote memory in place, so nothing else depends on it.
            for (int a = 0; a < n_devices; a++)
                for (int b = 0; b < n_devices; b++) {
                    int da = ctx->devs[a].dev, db = ctx->devs[b].dev, can = 0;
                    if (da == db) continue;
                    if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                        (void)hipSetDevice(da);
                        if (hipDeviceEnablePeerAccess(db, 0) != hipSuccess) (void)hipGetLastError();   // already enabled (by torch, or a second context): fine
                    } else {
                        (void)hipGetLastError();
                    }
                }
        }
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
    ctx->batch_workers.reset();
    for (int l = 0; l < NLANES; l++) ctx->workers[l].reset();
    for (std::vector<DevState>* lane : {&ctx->devs, &ctx->devs_b})
        for (auto& d : *lane) {
            (void)hipSetDevice(d.dev);
            if (d.stream) (void)hipStreamSynchronize(d.stream);
            if (d.acc2_stream) (void)hipStreamSynchronize(d.acc2_stream);
            if (d.aux_stream) (void)hipStreamSynchronize(d.aux_stream);
            d.for_each_buf([](DevBuf& b) { b.release(); });
            if (d.h_pairs) (void)hipHostFree(d.h_pairs);
            for (auto& e : d.ev)
                if (e) (void)hipEventDestroy(e);
            for (auto& e : d.cev)
                if (e) (void)hipEventDestroy(e);
            for (Scratch& sc : d.sc) {
                if (sc.h_meta) (void)hipHostFree(sc.h_meta);
                for (auto& e : sc.ev)
                    if (e) (void)hipEventDestroy(e);
            }
            for (auto& row : d.iev)
                for (auto& e : row)
                    if (e) (void)hipEventDestroy(e);
            if (d.d2h_stream) { (void)hipStreamSynchronize(d.d2h_stream); (void)hipStreamDestroy(d.d2h_stream); }
            if (d.aux_stream) (void)hipStreamDestroy(d.aux_stream);
            if (d.acc2_stream) (void)hipStreamDestroy(d.acc2_stream);
            if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
            if (d.stream) (void)hipStreamDestroy(d.stream);
        }
    for (auto& v : ctx->cache) v.clear();   // entries free their shards on their own devices
    for (size_t k = 0; k < ctx->residents.size() && k < ctx->devs.size(); k++) {
        (void)hipSetDevice(ctx->devs[k].dev);
        for (Resident& x : ctx->residents[k]) { x.buf.release(); x.flags.release(); }
    }
    delete ctx;
}

int mi_msm_num_devices(const mi_ctx* ctx) { return ctx ? (int)ctx->devs.size() : 0; }
int mi_msm_device_id(const mi_ctx* ctx, int slot) { return ctx && slot >= 0 && (size_t)slot < ctx->devs.size() ? ctx->devs[(size_t)slot].dev : -1; }

int mi_msm_g1_set_bases(mi_ctx* ctx, const mi_g1_affine* bases, size_t n) { return g1_set_bases(ctx, bases, n, 0); }
int mi_msm_g2_set_bases(mi_ctx* ctx, const mi_g2_affine* bases, size_t n) { return g2_set_bases(ctx, bases, n, 0); }
int mi_msm_g1_set_bases_precomputed(mi_ctx* ctx, const mi_g1_affine* bases, size_t n, unsigned window_bits) {
    return g1_set_bases(ctx, bases, n, window_bits ? window_bits : 1);
}
int mi_msm_g2_set_bases_precomputed(mi_ctx* ctx, const mi_g2_affine* bases, size_t n, unsigned window_bits) {
    return g2_set_bases(ctx, bases, n, window_bits ? window_bits : 1);
}

int mi_msm_g1_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return g1_validate_bases(ctx, n_invalid); }
int mi_msm_g2_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return g2_validate_bases(ctx, n_invalid); }

int mi_msm_g1(mi_ctx* ctx, const mi_g1_affine* bases, const uint8_t* scalars, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return g1_msm(ctx, bases, scalars, false, n, scalar_fmt, out);
}
int mi_msm_g2(mi_ctx* ctx, const mi_g2_affine* bases, const uint8_t* scalars, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return g2_msm(ctx, bases, scalars, false, n, scalar_fmt, out);
}
int mi_msm_g1_device(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return g1_msm(ctx, nullptr, static_cast<const uint8_t*>(d_scalars), true, n, scalar_fmt, out);
}
int mi_msm_g2_device(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return g2_msm(ctx, nullptr, static_cast<const uint8_t*>(d_scalars), true, n, scalar_fmt, out);
}

int mi_msm_g1_device_windows(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, void* d_out_windows, mi_window_info* info) {
    return g1_msm_windows(ctx, static_cast<const uint8_t*>(d_scalars), n, scalar_fmt, d_out_windows, info);
}
int mi_msm_g2_device_windows(mi_ctx* ctx, const void* d_scalars, size_t n, unsigned scalar_fmt, void* d_out_windows, mi_window_info* info) {
    return g2_msm_windows(ctx, static_cast<const uint8_t*>(d_scalars), n, scalar_fmt, d_out_windows, info);
}
int mi_g1_fold_windows(const mi_g1* windows, size_t n_ranks, size_t rank_stride, const mi_window_info* info, mi_g1* out) {
    return fold_windows<hostec::G1>(windows, n_ranks, rank_stride, info, out);
}
int mi_g2_fold_windows(const mi_g2* windows, size_t n_ranks, size_t rank_stride, const mi_window_info* info, mi_g2* out) {
    return fold_windows<hostec::G2>(windows, n_ranks, rank_stride, info, out);
}

int mi_msm_g1_batch(mi_ctx* ctx, const uint8_t* const* scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return g1_msm_batch(ctx, scalars, false, k, n, scalar_fmt, out);
}
int mi_msm_g2_batch(mi_ctx* ctx, const uint8_t* const* scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return g2_msm_batch(ctx, scalars, false, k, n, scalar_fmt, out);
}
int mi_msm_g1_batch_device(mi_ctx* ctx, const void* const* d_scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return g1_msm_batch(ctx, reinterpret_cast<const uint8_t* const*>(d_scalars), true, k, n, scalar_fmt, out);
}
int mi_msm_g2_batch_device(mi_ctx* ctx, const void* const* d_scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return g2_msm_batch(ctx, reinterpret_cast<const uint8_t* const*>(d_scalars), true, k, n, scalar_fmt, out);
}

int mi_g1_normalize_batch(mi_ctx* ctx, const mi_g1* in, size_t n, mi_g1_affine* out) { return g1_normalize(ctx, in, false, n, out); }
int mi_g2_normalize_batch(mi_ctx* ctx, const mi_g2* in, size_t n, mi_g2_affine* out) { return g2_normalize(ctx, in, false, n, out); }
int mi_g1_normalize_batch_device(mi_ctx* ctx, const void* d_in, size_t n, void* d_out) {
    return g1_normalize(ctx, static_cast<const mi_g1*>(d_in), true, n, static_cast<mi_g1_affine*>(d_out));
}
int mi_g2_normalize_batch_device(mi_ctx* ctx, const void* d_in, size_t n, void* d_out) {
    return g2_normalize(ctx, static_cast<const mi_g2*>(d_in), true, n, static_cast<mi_g2_affine*>(d_out));
}
int mi_msm_g1_set_bases_device(mi_ctx* ctx, const void* d_bases, size_t n) { return g1_set_bases_device(ctx, d_bases, n); }
int mi_msm_g2_set_bases_device(mi_ctx* ctx, const void* d_bases, size_t n) { return g2_set_bases_device(ctx, d_bases, n); }
int mi_msm_g1_set_bases_from_jacobian(mi_ctx* ctx, const mi_g1* points, size_t n) { return g1_set_bases_from_jacobian(ctx, points, n); }
int mi_msm_g2_set_bases_from_jacobian(mi_ctx* ctx, const mi_g2* points, size_t n) { return g2_set_bases_from_jacobian(ctx, points, n); }
int mi_msm_g1_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected) {
    return g1_set_bases_from_compressed(ctx, bytes, n, compressed, validate, n_rejected);
}
int mi_msm_g2_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected) {
    return g2_set_bases_from_compressed(ctx, bytes, n, compressed, validate, n_rejected);
}

int mi_g1_deserialize_batch(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g1_affine* out,
                            uint8_t* status) {
    return g1_deserialize(ctx, bytes, false, n, compressed, validate, out, status);
}
int mi_g1_deserialize_batch_device(mi_ctx* ctx, const void* d_bytes, size_t n, int compressed, int validate, void* d_out, void* d_status) {
    return g1_deserialize(ctx, static_cast<const uint8_t*>(d_bytes), true, n, compressed, validate, static_cast<mi_g1_affine*>(d_out), static_cast<uint8_t*>(d_status));
}
int mi_g2_deserialize_batch_device(mi_ctx* ctx, const void* d_bytes, size_t n, int compressed, int validate, void* d_out, void* d_status) {
    return g2_deserialize(ctx, static_cast<const uint8_t*>(d_bytes), true, n, compressed, validate, static_cast<mi_g2_affine*>(d_out), static_cast<uint8_t*>(d_status));
}
int mi_g1_serialize_batch(mi_ctx* ctx, const mi_g1_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return g1_serialize(ctx, points, n, compressed, bytes);
}
int mi_g2_deserialize_batch(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g2_affine* out,
                            uint8_t* status) {
    return g2_deserialize(ctx, bytes, false, n, compressed, validate, out, status);
}
int mi_g2_serialize_batch(mi_ctx* ctx, const mi_g2_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return g2_serialize(ctx, points, n, compressed, bytes);
}

int mi_g1_check_batch(mi_ctx* ctx, const mi_g1_affine* points, size_t n, uint8_t* status) { return g1_check_batch(ctx, points, false, n, status); }
int mi_g2_check_batch(mi_ctx* ctx, const mi_g2_affine* points, size_t n, uint8_t* status) { return g2_check_batch(ctx, points, false, n, status); }
int mi_g1_check_batch_device(mi_ctx* ctx, const void* d_points, size_t n, void* d_status) {
    return g1_check_batch(ctx, static_cast<const mi_g1_affine*>(d_points), true, n, static_cast<uint8_t*>(d_status));
}
int mi_g2_check_batch_device(mi_ctx* ctx, const void* d_points, size_t n, void* d_status) {
    return g2_check_batch(ctx, static_cast<const mi_g2_affine*>(d_points), true, n, static_cast<uint8_t*>(d_status));
}

int mi_multi_miller_loop(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out) {
    return miller(ctx, p, q, n, out, false);
}
int mi_multi_pairing(mi_ctx* ctx, const mi_g1_affine* p, const mi_g2_affine* q, size_t n, mi_fp12* out) {
    return miller(ctx, p, q, n, out, true);
}
int mi_final_exponentiation(const mi_fp12* f, mi_fp12* out) { return cpu_ok() ? final_exponentiation(f, out) : MI_E_UNSUPPORTED; }

int mi_g1_sum(const mi_g1* partials, size_t n, mi_g1* out) {
    if (!out || (n && !partials)) return MI_E_INVALID;
    if (!cpu_ok()) return MI_E_UNSUPPORTED;
    hostec::G1 r = hostec::G1::inf();
    for (size_t i = 0; i < n; i++) {
        hostec::G1 p;
        memcpy(&p, &partials[i], sizeof p);
        r = r.add(p);
    }
    memcpy(out, &r, sizeof r);
    return MI_OK;
}

int mi_g2_sum(const mi_g2* partials, size_t n, mi_g2* out) {
    if (!out || (n && !partials)) return MI_E_INVALID;
    if (!cpu_ok()) return MI_E_UNSUPPORTED;
    hostec::G2 r = hostec::G2::inf();
    for (size_t i = 0; i < n; i++) {
        hostec::G2 p;
        memcpy(&p, &partials[i], sizeof p);
        r = r.add(p);
    }
    memcpy(out, &r, sizeof r);
    return MI_OK;
}

int mi_msm_set_base_cache(mi_ctx* ctx, unsigned entries) {
    if (!ctx || entries > MI_BASE_CACHE_MAX) return fail(ctx, MI_E_INVALID, "entries must be 0..MI_BASE_CACHE_MAX");
    DeviceRestore restore;
    std::vector<std::shared_ptr<BaseCacheEntry>> dropped;   // freed outside the lock
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        if (ctx->cache_env) return MI_OK;   // ARKBLST_AMD_BASE_CACHE decides
        ctx->cache_entries = entries;
        for (auto& v : ctx->cache)
            while (v.size() > entries) { dropped.push_back(v.back()); v.pop_back(); }
    }
    return MI_OK;
}

int mi_msm_invalidate_base_cache(mi_ctx* ctx) {
    if (!ctx) return MI_E_INVALID;
    DeviceRestore restore;
    std::vector<std::shared_ptr<BaseCacheEntry>> dropped;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        for (auto& v : ctx->cache) { for (auto& e : v) dropped.push_back(e); v.clear(); }
    }
    return MI_OK;
}

int mi_msm_base_cache_stats(const mi_ctx* ctx, uint64_t* hits, uint64_t* misses, unsigned* entries_in_use) {
    if (!ctx) return MI_E_INVALID;
    mi_ctx* c = const_cast<mi_ctx*>(ctx);
    std::lock_guard<std::mutex> lk(c->cache_mu);
    if (hits) *hits = c->cache_hits;
    if (misses) *misses = c->cache_misses;
    if (entries_in_use) *entries_in_use = (unsigned)(c->cache[0].size() + c->cache[1].size());
    return MI_OK;
}

int mi_msm_set_window_bits(mi_ctx* ctx, unsigned window_bits) {
    if (!ctx || (window_bits != 0 && (window_bits < 7 || window_bits > 22))) return fail(ctx, MI_E_INVALID, "window_bits must be 0 or 7..22");
    LaneLock lk(ctx, true);
    ctx->forced_c = window_bits;
    return MI_OK;
}

int mi_msm_set_abort_check(mi_ctx* ctx, int (*check)(void*), void* user) {
    if (!ctx) return MI_E_INVALID;
    LaneLock lk(ctx, true);   // no call is in flight while the pair changes
    ctx->abort_check = check;
    ctx->abort_user = user;
    return MI_OK;
}

int mi_msm_get_window_bits(const mi_ctx* ctx, unsigned* window_bits) {
    if (!ctx || !window_bits) return MI_E_INVALID;
    LaneLock lk(const_cast<mi_ctx*>(ctx), true);
    *window_bits = ctx->forced_c;
    return MI_OK;
}

int mi_msm_set_pipeline(mi_ctx* ctx, const unsigned* weights, unsigned n_groups) {
    if (!ctx || n_groups > (unsigned)MAX_GROUPS || (n_groups > 1 && !weights)) return fail(ctx, MI_E_INVALID, "at most 4 window groups; weights must be given for more than one");
    LaneLock lk(ctx, true);
    ctx->pipe_weights.clear();
    if (n_groups == 1) ctx->pipe_weights = {1};
    for (unsigned g = 0; n_groups > 1 && g < n_groups; g++) ctx->pipe_weights.push_back(std::min(std::max(weights[g], 1u), 1000u));
    return MI_OK;
}

int mi_msm_set_profile_level(mi_ctx* ctx, int level) {
    if (!ctx || level < 0 || level > 2) return fail(ctx, MI_E_INVALID, "profile level must be 0, 1 or 2");
    LaneLock lk(ctx, true);
    ctx->profile_level = level;
    return MI_OK;
}

int mi_msm_last_profile(const mi_ctx* ctx, mi_profile* out) {
    if (!ctx || !out) return MI_E_INVALID;
    std::lock_guard<std::mutex> lk(ctx->info_mu);
    *out = ctx->prof;
    return MI_OK;
}

int mi_pairing_last_profile(const mi_ctx* ctx, mi_pairing_profile* out) {
    if (!ctx || !out) return MI_E_INVALID;
    std::lock_guard<std::mutex> lk(ctx->info_mu);
    *out = ctx->pprof;
    return MI_OK;
}

// text of the calling thread's most recent failure (thread-local: valid until the same thread fails again)
const char* mi_msm_last_error(const mi_ctx* ctx) {
    (void)ctx;
    return tls_error().c_str();
}

const char* mi_msm_strerror(int code) {
    switch (code) {
        case MI_OK: return "ok";
        case MI_E_INVALID: return "invalid argument";
        case MI_E_NO_DEVICE: return "no usable HIP device";
        case MI_E_HIP: return "HIP runtime error";
        case MI_E_NOMEM: return "out of memory";
        case MI_E_NO_BASES: return "no resident base set";
        case MI_E_UNSUPPORTED: return "not supported on this host (the library needs an x86-64 CPU with BMI2 and ADX)";
        case MI_E_COMM: return "RCCL communication error";
        case MI_E_ABORTED: return "aborted by the caller's abort check";
        default: return "unknown error";
    }
}

#if defined(MI_TEST_HOOKS)
int mi_test_fp_op(mi_ctx* ctx, int op, const mi_fp* a, const mi_fp* b, mi_fp* out, size_t n) { return test_fp_op(ctx, op, a, b, out, n); }
int mi_test_set_pairing(mi_ctx* ctx, unsigned share, unsigned batch, int single_lane) {
    if (!ctx) return MI_E_INVALID;
    LaneLock lk(ctx, true);
    ctx->test_pairing_share = share;
    ctx->test_pairing_batch = batch;
    ctx->test_pairing_single_lane = single_lane != 0;
    return MI_OK;
}
int mi_test_set_max_part(mi_ctx* ctx, size_t points) {
    if (!ctx) return MI_E_INVALID;
    LaneLock lk(ctx, true);
    ctx->test_max_part = points;
    return MI_OK;
}
void mi_test_fail_allocs(int count) { g_fail_allocs.store(count); }
int mi_test_set_no_peer(mi_ctx* ctx, int no_peer) {
    if (!ctx) return MI_E_INVALID;
    LaneLock lk(ctx, true);
    ctx->test_no_peer = no_peer != 0;
    return MI_OK;
}
int mi_test_plan(size_t n, unsigned forced_c, int group, int shared, size_t stride, uint32_t* out) {
    if (!out || n == 0) return MI_E_INVALID;
    Plan p = make_plan(n, forced_c, group == 0 ? g1_cost() : g2_cost(), (shared & 1) != 0, stride, (shared & 2) != 0);
    out[0] = p.c; out[1] = p.nwin; out[2] = p.bwin; out[3] = p.coop_L; out[4] = p.chunk_buckets; out[5] = p.logT; out[6] = p.lo_bits;
    out[7] = p.serial_reduce ? 1u : 0u; out[8] = p.chunks_per_win; out[9] = (uint32_t)(p.nbuckets >> 32); out[10] = (uint32_t)p.nbuckets;
    out[11] = (uint32_t)p.nchunks; out[12] = p.serial_reduce ? p.serial_L : 0u;
    return MI_OK;
}
#endif

}  // extern "C"
