// Bulk point (de)serialisation entry points (/root/reference/src/g1.rs:358-431, src/g2.rs:338-411) and,
// in test builds, the field-level hook that pins the device arithmetic against the oracle.
#include "internal.hpp"
#include "codec_kernels.cuh"

namespace mi {
namespace {

// shared driver of the (de)serialisation entry points: `unit` = compressed size in bytes (48 G1, 96 G2)
// kernel2: an optional second pass over the decoded points (the subgroup test of the G2 decoder, k_validate<G2C, true>)
template <class KDe, class KVal>
int deserialize_impl(mi_ctx* ctx, KDe kernel, KVal kernel2, size_t unit, const uint8_t* bytes, size_t n, int compressed, int validate, void* out,
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
            if constexpr (!std::is_same<KVal, std::nullptr_t>::value) {
                if (validate)
                    hipLaunchKernelGGL(kernel2, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, (uint32_t*)dout.p, (uint32_t)n,
                                       (uint8_t*)dst.p, (uint32_t*)nullptr);
            }
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
        // all devices first (the kernels run side by side), then one wait per device
        std::vector<uint32_t*> counters(ctx->devs.size(), nullptr);
        struct Free { std::vector<uint32_t*>& v; mi_ctx* c; ~Free() { for (size_t k = 0; k < v.size(); k++) if (v[k]) { (void)hipSetDevice(c->devs[k].dev); (void)hipFree(v[k]); } } } fr{counters, ctx};
        for (size_t k = 0; k < ctx->devs.size(); k++) {
            DevState& d = ctx->devs[k];
            Resident& res = d.res[idx];
            res.validated = false;
            if (res.n == 0) continue;
            if (res.n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "resident shard too large");
            HIP_TRY(hipSetDevice(d.dev));
            HIP_TRY(hipMalloc((void**)&counters[k], 4));
            HIP_TRY(hipMemsetAsync(counters[k], 0, 4, d.stream));
            hipLaunchKernelGGL((msmk::k_validate<C, 0>), dim3((uint32_t)((res.n + 255) / 256)), dim3(256), 0, d.stream, (uint32_t*)res.buf.p,
                               (uint32_t)res.n, (uint8_t*)nullptr, counters[k]);
            HIP_TRY(hipGetLastError());
        }
        for (size_t k = 0; k < ctx->devs.size(); k++) {
            if (!counters[k]) continue;
            DevState& d = ctx->devs[k];
            HIP_TRY(hipSetDevice(d.dev));
            uint32_t cnt = 0;
            HIP_TRY(hipMemcpyAsync(&cnt, counters[k], 4, hipMemcpyDeviceToHost, d.stream));
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

// Valid::batch_check over affine points in host memory (src/g1.rs:386-396 per element; the projective form, src/g1.rs:570-579, is
// normalize_batch followed by this)
template <class C>
int check_batch_impl(mi_ctx* ctx, const void* points, size_t n, uint8_t* status) {
    if (!ctx || (n && (!points || !status))) return fail(ctx, MI_E_INVALID, "invalid argument");
    if (n == 0) return MI_OK;
    if (n > 0x7fffffffull) return fail(ctx, MI_E_INVALID, "n too large");
    LaneLock lk(ctx, true);
    return guarded(ctx, [&]() -> int {
        DevState& d = ctx->devs[0];
        HIP_TRY(hipSetDevice(d.dev));
        const size_t aff = (size_t)msmk::Geo<C>::RAW_AFF * 4;
        DevBuf din, dst;
        struct Rel { DevBuf &a, &b; ~Rel() { a.release(); b.release(); } } rel{din, dst};
        din.ensure(n * aff); dst.ensure(n);
        HIP_TRY(hipEventRecord(d.ev[0], d.stream));
        HIP_TRY(hipMemcpyAsync(din.p, points, n * aff, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipEventRecord(d.ev[1], d.stream));
        hipLaunchKernelGGL((msmk::k_validate<C, 2>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, d.stream, (uint32_t*)din.p, (uint32_t)n,
                           (uint8_t*)dst.p, (uint32_t*)nullptr);
        HIP_TRY(hipEventRecord(d.ev[2], d.stream));
        HIP_TRY(hipMemcpyAsync(status, dst.p, n, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(hipGetLastError());
        mi_profile pr{};
        pr.n = n;
        pr.h2d_ms = ev_ms(d.ev[0], d.ev[1]);
        pr.accumulate_ms = ev_ms(d.ev[1], d.ev[2]);
        set_prof(ctx, pr);
        return MI_OK;
    });
}

}  // namespace

int g1_check_batch(mi_ctx* ctx, const mi_g1_affine* points, size_t n, uint8_t* status) { return check_batch_impl<msmk::G1C>(ctx, points, n, status); }
int g2_check_batch(mi_ctx* ctx, const mi_g2_affine* points, size_t n, uint8_t* status) { return check_batch_impl<msmk::G2C>(ctx, points, n, status); }

int g1_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return validate_bases_impl<msmk::G1C>(ctx, 0, n_invalid); }
int g2_validate_bases(mi_ctx* ctx, size_t* n_invalid) { return validate_bases_impl<msmk::G2C>(ctx, 1, n_invalid); }

int g1_deserialize(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g1_affine* out, uint8_t* status) {
    return deserialize_impl(ctx, msmk::k_deserialize_g1, nullptr, 48, bytes, n, compressed, validate, out, status);
}
int g1_serialize(mi_ctx* ctx, const mi_g1_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl(ctx, msmk::k_serialize_g1, 48, points, n, compressed, bytes);
}
int g2_deserialize(mi_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int validate, mi_g2_affine* out, uint8_t* status) {
    return deserialize_impl(ctx, msmk::k_deserialize_g2, msmk::k_validate<msmk::G2C, 1>, 96, bytes, n, compressed, validate, out, status);
}
int g2_serialize(mi_ctx* ctx, const mi_g2_affine* points, size_t n, int compressed, uint8_t* bytes) {
    return serialize_impl(ctx, msmk::k_serialize_g2, 96, points, n, compressed, bytes);
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
