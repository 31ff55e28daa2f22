// Bulk point (de)serialisation entry points (/root/reference/src/g1.rs:358-431, src/g2.rs:338-411) and,
// in test builds, the field-level hook that pins the device arithmetic against the oracle.
#include "internal.hpp"
#include "io_chunks.hpp"
#include "codec_kernels.cuh"

namespace mi {
namespace {

// what the decode of one curve consists of: `unit` = compressed size in bytes (48 G1, 96 G2); the G2 decoder's subgroup test is a second
// kernel over the decoded points (k_validate<G2C, 1>: fused, the decoder kept 86 registers in scratch)
struct G1Codec {
    using C = msmk::G1C;
    static constexpr size_t UNIT = 48;
    static void decode(hipStream_t s, const uint8_t* in, size_t cnt, int compressed, int validate, uint32_t* out, uint8_t* st) {
        hipLaunchKernelGGL(msmk::k_deserialize_g1, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, in, (uint32_t)cnt, compressed ? 1 : 0, validate ? 1 : 0, out, st);
    }
    static void encode(hipStream_t s, const uint32_t* in, size_t cnt, int compressed, uint8_t* out) {
        hipLaunchKernelGGL(msmk::k_serialize_g1, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, in, (uint32_t)cnt, compressed ? 1 : 0, out);
    }
};
struct G2Codec {
    using C = msmk::G2C;
    static constexpr size_t UNIT = 96;
    static void decode(hipStream_t s, const uint8_t* in, size_t cnt, int compressed, int validate, uint32_t* out, uint8_t* st) {
        hipLaunchKernelGGL(msmk::k_deserialize_g2, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, in, (uint32_t)cnt, compressed ? 1 : 0, validate ? 1 : 0, out, st);
        if (validate)   // the subgroup test on lane pairs (codec_kernels.cuh k_validate_g2_coop): two lanes per point
            hipLaunchKernelGGL((msmk::k_validate_g2_coop<1>), dim3((uint32_t)((2 * cnt + 255) / 256)), dim3(256), 0, s, out, (uint32_t)cnt, st, (uint32_t*)nullptr);
    }
    static void encode(hipStream_t s, const uint32_t* in, size_t cnt, int compressed, uint8_t* out) {
        hipLaunchKernelGGL(msmk::k_serialize_g2, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, in, (uint32_t)cnt, compressed ? 1 : 0, out);
    }
};

// decode of n encodings on one device.  Host pointers cross PCIe in chunks (io_chunks.hpp); a NULL host pointer with a device pointer
// means the data is / stays in device memory.  Results: d_out (n affine points in the reference's form), d_st (n status bytes).
template <class K>
void decode_run(mi_ctx* ctx, DevState& d, const uint8_t* h_bytes, const uint8_t* d_bytes_user, size_t n, int compressed, int validate, void* h_out,
                uint8_t* h_status, void* d_out_user, uint8_t* d_status_user, bool publish_profile) {
    HIP_TRY(hipSetDevice(d.dev));
    const size_t sz = compressed ? K::UNIT : 2 * K::UNIT, aff = 2 * K::UNIT;
    uint8_t* d_in = const_cast<uint8_t*>(d_bytes_user);
    if (!d_in) { d.io_in.ensure(n * sz); d_in = (uint8_t*)d.io_in.p; }
    uint32_t* d_out = (uint32_t*)d_out_user;
    if (!d_out) { d.io_out.ensure(n * aff); d_out = (uint32_t*)d.io_out.p; }
    uint8_t* d_st = d_status_user;
    if (!d_st) { d.io_status.ensure(n + 16); d_st = (uint8_t*)d.io_status.p; }
    const IoOut outs[2] = {{h_out, d_out, aff}, {h_status, d_st, 1}};
    double h2d = 0;
    const double k_ms = io_stream_pass(d, n, h_bytes, d_in, sz, outs, 2, [&](size_t lo, size_t cnt) {
        K::decode(d.stream, d_in + lo * sz, cnt, compressed, validate, d_out + lo * (aff / 4), d_st + lo);
    }, &h2d, io_heavy_chunk(std::is_same<typename K::C, msmk::G2C>::value));
    if (publish_profile) {
        mi_profile pr{};
        pr.n = n;
        pr.h2d_ms = h2d;
        pr.accumulate_ms = k_ms;   // the decode kernels (sum over the chunks)
        set_prof(ctx, pr);
    }
}

template <class K>
int deserialize_impl(mi_ctx* ctx, const uint8_t* bytes, bool on_device, size_t n, int compressed, int validate, void* out, uint8_t* status) {
    if (!ctx || (n && (!bytes || !out || !status))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    if (on_device && ctx->devs.size() != 1) return fail(ctx, MI_E_INVALID, "the *_device entry points need a single-device context");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        auto t0 = std::chrono::steady_clock::now();
        if (on_device) {
            if (device_of_ptr(bytes, "d_bytes", 1) != d.dev || device_of_ptr(out, "d_out", 4) != d.dev || device_of_ptr(status, "d_status", 1) != d.dev)
                return fail(ctx, MI_E_INVALID, "device buffers must live on the context's device");
            decode_run<K>(ctx, d, nullptr, bytes, n, compressed, validate, nullptr, nullptr, out, status, true);
        } else {
            decode_run<K>(ctx, d, bytes, nullptr, n, compressed, validate, out, status, nullptr, nullptr, true);
        }
        std::lock_guard<std::mutex> g(ctx->info_mu);
        ctx->prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return MI_OK;
    });
}

template <class K>
int serialize_impl(mi_ctx* ctx, const void* points, size_t n, int compressed, uint8_t* bytes) {
    if (!ctx || (n && (!bytes || !points))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipSetDevice(d.dev));
        const size_t sz = compressed ? K::UNIT : 2 * K::UNIT, aff = 2 * K::UNIT;
        d.io_in.ensure(n * aff);
        d.io_out.ensure(n * sz);
        const IoOut outs[1] = {{bytes, d.io_out.p, sz}};
        double h2d = 0;
        const double k_ms = io_stream_pass(d, n, points, d.io_in.p, aff, outs, 1, [&](size_t lo, size_t cnt) {
            K::encode(d.stream, (const uint32_t*)d.io_in.p + lo * (aff / 4), cnt, compressed, (uint8_t*)d.io_out.p + lo * sz);
        }, &h2d);
        mi_profile pr{};
        pr.n = n; pr.h2d_ms = h2d; pr.accumulate_ms = k_ms;
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

// Valid::check of the resident base set, every device over its shard; a clean set is recorded (Resident::validated): later MSMs over it
// may fold the scalars' signs (common.hpp num_windows)
template <class C>
int validate_bases_impl(mi_ctx* ctx, int idx, size_t* n_invalid) {
    if (!ctx || !n_invalid) return fail(ctx, MI_E_INVALID, "invalid argument");
    *n_invalid = 0;
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        size_t have = 0, bad = 0;
        for (auto& d : ctx->devs) have += d.res[idx].n;
        if (have == 0) return fail(ctx, MI_E_NO_BASES, "no resident base set for this group");
        auto t0 = std::chrono::steady_clock::now();
        // all devices first (the kernels run side by side), then one wait per device; the counter lives in the lane's staging buffer
        for (size_t k = 0; k < ctx->devs.size(); k++) {
            DevState& d = ctx->devs[k];
            Resident& res = d.res[idx];
            res.validated = false;
            if (res.n == 0) continue;
            if (res.n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "resident shard too large");
            HIP_TRY(hipSetDevice(d.dev));
            d.io_status.ensure(16);
            HIP_TRY(hipMemsetAsync(d.io_status.p, 0, 4, d.stream));
            if constexpr (std::is_same<C, msmk::G2C>::value)
                hipLaunchKernelGGL((msmk::k_validate_g2_coop<0>), dim3((uint32_t)((2 * res.n + 255) / 256)), dim3(256), 0, d.stream, (uint32_t*)res.buf.p,
                                   (uint32_t)res.n, (uint8_t*)nullptr, (uint32_t*)d.io_status.p);
            else
                hipLaunchKernelGGL((msmk::k_validate<C, 0>), dim3((uint32_t)((res.n + 255) / 256)), dim3(256), 0, d.stream, (uint32_t*)res.buf.p,
                                   (uint32_t)res.n, (uint8_t*)nullptr, (uint32_t*)d.io_status.p);
            HIP_TRY(hipGetLastError());
        }
        for (size_t k = 0; k < ctx->devs.size(); k++) {
            DevState& d = ctx->devs[k];
            if (d.res[idx].n == 0) continue;
            HIP_TRY(hipSetDevice(d.dev));
            uint32_t cnt = 0;
            HIP_TRY(hipMemcpyAsync(&cnt, d.io_status.p, 4, hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipStreamSynchronize(d.stream));
            bad += cnt;
        }
        if (bad == 0)
            for (auto& d : ctx->devs) d.res[idx].validated = true;
        *n_invalid = bad;
        mi_profile pr{};
        pr.n = have;
        pr.total_ms = pr.accumulate_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

// Valid::batch_check over affine points (src/g1.rs:386-396 per element; the projective form, src/g1.rs:570-579, is normalize_batch
// followed by this)
template <class C>
int check_batch_impl(mi_ctx* ctx, const void* points, bool on_device, size_t n, uint8_t* status) {
    if (!ctx || (n && (!points || !status))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    if (on_device && ctx->devs.size() != 1) return fail(ctx, MI_E_INVALID, "the *_device entry points need a single-device context");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipSetDevice(d.dev));
        const size_t aff = (size_t)msmk::Geo<C>::RAW_AFF * 4;
        uint32_t* d_pts;
        uint8_t* d_st;
        if (on_device) {
            if (device_of_ptr(points, "d_points", 4) != d.dev || device_of_ptr(status, "d_status", 1) != d.dev)
                return fail(ctx, MI_E_INVALID, "device buffers must live on the context's device");
            d_pts = (uint32_t*)const_cast<void*>(points);
            d_st = status;
        } else {
            d.io_in.ensure(n * aff);
            d.io_status.ensure(n + 16);
            d_pts = (uint32_t*)d.io_in.p;
            d_st = (uint8_t*)d.io_status.p;
        }
        const IoOut outs[1] = {{on_device ? nullptr : (void*)status, d_st, 1}};
        double h2d = 0;
        const double k_ms = io_stream_pass(d, n, on_device ? nullptr : points, d_pts, aff, outs, 1, [&](size_t lo, size_t cnt) {
            if constexpr (std::is_same<C, msmk::G2C>::value)
                hipLaunchKernelGGL((msmk::k_validate_g2_coop<2>), dim3((uint32_t)((2 * cnt + 255) / 256)), dim3(256), 0, d.stream, d_pts + lo * (aff / 4), (uint32_t)cnt,
                                   d_st + lo, (uint32_t*)nullptr);
            else
                hipLaunchKernelGGL((msmk::k_validate<C, 2>), dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, d.stream, d_pts + lo * (aff / 4), (uint32_t)cnt,
                                   d_st + lo, (uint32_t*)nullptr);
        }, &h2d, io_heavy_chunk(std::is_same<C, msmk::G2C>::value));
        mi_profile pr{};
        pr.n = n; pr.h2d_ms = h2d; pr.accumulate_ms = k_ms;
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

// SRS loading without a round trip through the host: decode (+ Valid::check) the encodings on the GPU, every device over its shard, and
// make the decoded points the resident base set.  All or nothing: when a single encoding is rejected nothing is installed (the previous
// resident set stays) and the count comes back; with `validate` a clean set is recorded as validated right away (the decoder ran
// Valid::check on every point), so no separate mi_msm_g1_validate_bases pass is needed.
template <class K, class Install>
int set_bases_from_compressed_impl(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected, Install install) {
    if (!ctx || (n && !bytes) || !n_rejected) return fail(ctx, MI_E_INVALID, "invalid argument");
    *n_rejected = 0;
    if ((n + ctx->devs.size() - 1) / ctx->devs.size() > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "more than 2^31 points per device");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        const size_t g = ctx->devs.size(), sz = compressed ? K::UNIT : 2 * K::UNIT;
        std::vector<PartErr> errs(g);
        std::vector<uint32_t> bad(g, 0);
        auto t0 = std::chrono::steady_clock::now();
        for_each_device(lk, g, [&](size_t k) {
            guarded_part(errs[k], [&] {
                DevState& d = ctx->devs[k];
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                if (hi == lo) return;
                decode_run<K>(ctx, d, bytes + lo * sz, nullptr, hi - lo, compressed, validate, nullptr, nullptr, nullptr, nullptr, false);
                uint32_t* counter = (uint32_t*)((uint8_t*)d.io_status.p + ((hi - lo + 3) & ~(size_t)3));   // io_status has 16 spare bytes
                HIP_TRY(hipMemsetAsync(counter, 0, 4, d.stream));
                hipLaunchKernelGGL(msmk::k_count_rejected, dim3((uint32_t)((hi - lo + 255) / 256)), dim3(256), 0, d.stream, (const uint8_t*)d.io_status.p,
                                   (uint32_t)(hi - lo), counter);
                HIP_TRY(hipMemcpyAsync(&bad[k], counter, 4, hipMemcpyDeviceToHost, d.stream));
                HIP_TRY(hipStreamSynchronize(d.stream));
            });
        });
        size_t total_bad = 0;
        for (size_t k = 0; k < g; k++) {
            if (errs[k].code != MI_OK) return fail(ctx, errs[k].code, errs[k].msg);
            total_bad += bad[k];
        }
        *n_rejected = total_bad;
        if (total_bad) return fail(ctx, MI_E_INVALID, "rejected encodings: the resident base set was not changed");
        for_each_device(lk, g, [&](size_t k) {
            guarded_part(errs[k], [&] {
                size_t lo, hi;
                shard_range(n, g, k, lo, hi);
                install(ctx, k, hi > lo ? ctx->devs[k].io_out.p : nullptr, lo, hi - lo, validate != 0);
            });
        });
        for (size_t k = 0; k < g; k++)
            if (errs[k].code != MI_OK) return fail(ctx, errs[k].code, errs[k].msg);
        mi_profile pr{};
        pr.n = n;
        pr.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        set_prof(ctx, pr);
        return MI_OK;
    });
}

}  // namespace

int g1_check_batch(mi_ctx* ctx, const mi_g1_affine* points, bool on_device, size_t n, uint8_t* status) { return check_batch_impl<msmk::G1C>(ctx, points, on_device, n, status); }
int g2_check_batch(mi_ctx* ctx, const mi_g2_affine* points, bool on_device, size_t n, uint8_t* status) { return check_batch_impl<msmk::G2C>(ctx, points, on_device, n, status); }

int g1_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return validate_bases_impl<msmk::G1C>(ctx, 0, n_invalid); }
int g2_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return validate_bases_impl<msmk::G2C>(ctx, 1, n_invalid); }

int g1_deserialize(mi_ctx* ctx, const uint8_t* bytes, bool on_device, size_t n, int compressed, int validate, mi_g1_affine* out, uint8_t* status) {
    return deserialize_impl<G1Codec>(ctx, bytes, on_device, n, compressed, validate, out, status);
}
int g1_serialize(mi_ctx* ctx, const mi_g1_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl<G1Codec>(ctx, points, n, compressed, bytes);
}
int g2_deserialize(mi_ctx* ctx, const uint8_t* bytes, bool on_device, size_t n, int compressed, int validate, mi_g2_affine* out, uint8_t* status) {
    return deserialize_impl<G2Codec>(ctx, bytes, on_device, n, compressed, validate, out, status);
}
int g2_serialize(mi_ctx* ctx, const mi_g2_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl<G2Codec>(ctx, points, n, compressed, bytes);
}
int g1_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected) {
    return set_bases_from_compressed_impl<G1Codec>(ctx, bytes, n, compressed, validate, n_rejected, g1_install_resident);
}
int g2_set_bases_from_compressed(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, size_t* n_rejected) {
    return set_bases_from_compressed_impl<G2Codec>(ctx, bytes, n, compressed, validate, n_rejected, g2_install_resident);
}

#if defined(MI_TEST_HOOKS)
int test_fp_op(mi_ctx* ctx, int op, const mi_fp* a, const mi_fp* b, mi_fp* out, size_t n) {
    if (!ctx || !a || !b || !out || op < 0 || op > 3) return fail(ctx, MI_E_INVALID, "invalid argument");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        HIP_TRY(hipSetDevice(d.dev));
        DevBuf da, db, dout;
        struct Rel { DevBuf &a, &b, &c; ~Rel() { a.release(); b.release(); c.release(); } } rel{da, db, dout};
        da.ensure(n * 48); db.ensure(n * 48); dout.ensure(n * 48);
        HIP_TRY(hipMemcpyAsync(da.p, a, n * 48, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(db.p, b, n * 48, hipMemcpyHostToDevice, d.stream));
        hipLaunchKernelGGL(msmk::k_test_fp_op, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, op, (const uint32_t*)da.p,
                           (const uint32_t*)db.p, (uint32_t*)dout.p, (uint32_t)n);
        HIP_TRY(hipMemcpyAsync(out, dout.p, n * 48, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        return MI_OK;
    });
}
#endif

}  // namespace mi
