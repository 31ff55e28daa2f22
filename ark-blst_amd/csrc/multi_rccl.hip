// libarkblst_amd_rccl.so — the exchange step of a one-process-per-GPU deployment (include/arkblst_amd_rccl.h), on top of the PUBLIC
// entry points of libarkblst_amd.so: per-window sums in device memory -> ncclAllGather over xGMI -> one D2H -> host fold.
// Host code only (no kernel in this translation unit).  The reference has nothing here (/root/reference/src/gpu.rs:233-239).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/arkblst_amd_rccl.h"
#include "deadline.hpp"

static_assert(MI_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

std::string& tls_err() {
    static thread_local std::string e;
    return e;
}
int fail(int code, const std::string& msg) {
    tls_err() = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? MI_E_NOMEM : MI_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
int nccl_fail(ncclResult_t r, const char* what) { return fail(MI_E_COMM, std::string(what) + ": " + ncclGetErrorString(r)); }

#define HIP_RC(expr)                                     \
    do {                                                 \
        hipError_t e_ = (expr);                          \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)
#define NCCL_RC(expr)                                      \
    do {                                                   \
        ncclResult_t r_ = (expr);                          \
        if (r_ != ncclSuccess) return nccl_fail(r_, #expr); \
    } while (0)

// One rank's block of the all-gather: SLOTS points of the group's Jacobian size; slot 0 is the header, slots 1.. the window sums
// (window 0 first).  A fixed size, so that ranks with different window counts still meet in ONE collective.
constexpr size_t SLOTS = 1 + MI_MAX_WINDOWS;
constexpr size_t MAX_JAC = sizeof(mi_g2);
struct Header {
    uint32_t magic, window_bits, num_windows, reserved;
    uint64_t n;
};
constexpr uint32_t MAGIC = 0x4d495243u;   // "MIRC"
constexpr uint32_t FAILED = 0xffffffffu;  // Header::window_bits of a rank whose local part failed: it still joins the collective, so that nobody hangs

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

struct DeviceGuard {   // the caller's current device is restored on every path
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct mi_rccl_comm {
    mi_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    int device = -1, n_ranks = 0, rank = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_send = nullptr;   // SLOTS x MAX_JAC
    uint8_t* d_recv = nullptr;   // n_ranks x SLOTS x MAX_JAC
    uint8_t* h_recv = nullptr;   // pinned, same size
    Header* h_hdr = nullptr;     // pinned staging of this rank's header
    std::mutex mu;
    mi_rccl_timing timing{};
    double timeout_ms = 60000.0;   // mi_rccl_comm_set_timeout_ms: how long a rank waits inside the exchange for its peers
    bool aborted = false;          // the communicator was aborted after a timeout / an asynchronous error: every later call fails at once
    // window size the ranks agreed on after a disagreement, and the n of THIS rank it was for: pinned for the duration of later calls with
    // the same n, so that they agree at the first attempt (the caller's context is left as it was: ADVICE r05)
    unsigned agreed_c = 0;
    size_t agreed_for_n = 0;
};

namespace {

int finish_create(mi_rccl_comm* c) {
    HIP_RC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_RC(hipMalloc((void**)&c->d_send, SLOTS * MAX_JAC));
    HIP_RC(hipMalloc((void**)&c->d_recv, (size_t)c->n_ranks * SLOTS * MAX_JAC));
    HIP_RC(hipHostMalloc((void**)&c->h_recv, (size_t)c->n_ranks * SLOTS * MAX_JAC, hipHostMallocDefault));
    HIP_RC(hipHostMalloc((void**)&c->h_hdr, sizeof(Header), hipHostMallocDefault));
    HIP_RC(hipMemset(c->d_send, 0, SLOTS * MAX_JAC));
    return MI_OK;
}

// the caller's window-size setting is restored on every path out of the call
struct WindowBitsGuard {
    mi_ctx* ctx;
    unsigned saved = 0;
    bool armed = false;
    explicit WindowBitsGuard(mi_ctx* c) : ctx(c) { armed = mi_msm_get_window_bits(ctx, &saved) == MI_OK; }
    ~WindowBitsGuard() { if (armed) (void)mi_msm_set_window_bits(ctx, saved); }
};

// After a timeout or an asynchronous error the communicator is unusable: abort it (peers blocked in the same collective then see an
// error instead of waiting for ever) and fail every later call.  A communicator the host program attached stays the program's to destroy.
void abort_comm(mi_rccl_comm* c) {
    if (c->aborted) return;
    c->aborted = true;
    if (c->comm) (void)ncclCommAbort(c->comm);
    if (c->owns_comm) c->comm = nullptr;
    (void)hipGetLastError();
}

template <class Jac, class Windows, class Fold>
int allgather_fold(mi_rccl_comm* c, const void* d_scalars, size_t n, unsigned fmt, Jac* out, Windows device_windows, Fold fold_windows) {
    if (!c || !out || (n && !d_scalars)) return fail(MI_E_INVALID, "invalid argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->aborted || !c->comm) return fail(MI_E_COMM, "the communicator was aborted by an earlier failure: create a new one");
    DeviceGuard dg(c->device);
    WindowBitsGuard restore(c->ctx);
    constexpr size_t JAC = sizeof(Jac);
    const size_t block = SLOTS * JAC;
    mi_rccl_timing tm{};
    tm.bytes_per_rank = (uint32_t)block;
    if (c->agreed_c && c->agreed_for_n == n && restore.saved == 0) (void)mi_msm_set_window_bits(c->ctx, c->agreed_c);   // what this shard size agreed on before
    for (uint32_t attempt = 0;; attempt++) {
        // ---- local part: this rank's shard through the single-GPU pipeline; the window sums stay in device memory.  From here to the
        // all-gather NOTHING returns: a failure is recorded and travels as a FAILED header, so that no peer is left waiting for this rank
        auto t0 = std::chrono::steady_clock::now();
        mi_window_info info{};
        int local_rc = MI_OK;
        std::string local_msg;
        auto local_fail = [&](int code, const std::string& msg) {
            if (local_rc == MI_OK) { local_rc = code; local_msg = msg; }
            info = mi_window_info{FAILED, 0};
        };
        if (n) {
            int rc = device_windows(c->ctx, d_scalars, n, fmt, c->d_send + JAC, &info);
            if (rc != MI_OK) local_fail(rc, std::string("device_windows: ") + mi_msm_last_error(c->ctx));   // e.g. n beyond this rank's resident shard
        } else {
            hipError_t e = hipMemsetAsync(c->d_send, 0, block, c->stream);   // Z = 0 everywhere: the point at infinity per window
            if (e != hipSuccess) { (void)hipGetLastError(); local_fail(MI_E_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e)); }
        }
        tm.msm_ms += ms_since(t0);
        // ---- exchange: header + window sums of every rank in ONE all-gather, one D2H of the gathered block, fold on every rank
        auto t1 = std::chrono::steady_clock::now();
        *c->h_hdr = Header{MAGIC, info.window_bits, info.num_windows, 0, (uint64_t)n};
        hipError_t he = hipMemcpyAsync(c->d_send, c->h_hdr, sizeof(Header), hipMemcpyHostToDevice, c->stream);
        if (he != hipSuccess) {
            // not even the failure mark can be sent: abort the communicator, the peers' collective ends with an error instead of hanging
            (void)hipGetLastError();
            abort_comm(c);
            return fail(local_rc != MI_OK ? local_rc : MI_E_HIP, local_rc != MI_OK ? local_msg : std::string("hipMemcpyAsync(header): ") + hipGetErrorString(he));
        }
        ncclResult_t nr = ncclAllGather(c->d_send, c->d_recv, block, ncclUint8, c->comm, c->stream);
        if (nr != ncclSuccess) {
            abort_comm(c);
            return nccl_fail(nr, "ncclAllGather");
        }
        he = hipMemcpyAsync(c->h_recv, c->d_recv, (size_t)c->n_ranks * block, hipMemcpyDeviceToHost, c->stream);
        if (he != hipSuccess) {   // the collective itself is complete-able by the peers: no abort needed, this rank just has no result
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->stream);
            return hip_fail(he, "hipMemcpyAsync(gathered block)");
        }
        // wait with a deadline: a peer that never arrives must not hang this rank for ever
        hipError_t stream_err = hipSuccess;
        ncclResult_t async_err = ncclSuccess;
        double waited = 0;
        const mi::WaitResult wr = mi::wait_deadline(
            [&] {
                hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) return true;
                if (q != hipErrorNotReady) stream_err = q;
                return false;
            },
            [&] {
                if (stream_err != hipSuccess) return true;
                ncclResult_t a = ncclSuccess;
                if (ncclCommGetAsyncError(c->comm, &a) != ncclSuccess) return false;
                if (a != ncclSuccess && a != ncclInProgress) { async_err = a; return true; }
                return false;
            },
            c->timeout_ms, &waited);
        if (wr != mi::WaitResult::Done) {
            (void)hipGetLastError();
            abort_comm(c);
            tm.exchange_ms += ms_since(t1);
            c->timing = tm;
            if (wr == mi::WaitResult::TimedOut)
                return fail(MI_E_COMM, "the all-gather did not complete within " + std::to_string((long)c->timeout_ms) + " ms (a peer never arrived?): communicator aborted");
            return fail(MI_E_COMM, stream_err != hipSuccess ? std::string("stream error during the all-gather: ") + hipGetErrorString(stream_err)
                                                            : std::string("asynchronous RCCL error: ") + ncclGetErrorString(async_err));
        }
        // every rank reads the same headers and therefore takes the same decision below
        mi_window_info agreed{};
        bool same = true;
        uint32_t cmax = 0;
        if (local_rc != MI_OK) return fail(local_rc, local_msg);   // after the collective: every rank has seen this rank's FAILED header
        for (int r = 0; r < c->n_ranks; r++) {
            Header h;
            memcpy(&h, c->h_recv + (size_t)r * block, sizeof h);
            if (h.magic != MAGIC || (h.window_bits != FAILED && h.num_windows > MI_MAX_WINDOWS)) return fail(MI_E_COMM, "all-gather returned a malformed block");
            if (h.window_bits == FAILED) return fail(MI_E_COMM, "the local part of rank " + std::to_string(r) + " failed: no result on any rank");
            if (h.num_windows == 0) continue;   // a rank without points: its slots are infinity
            if (agreed.num_windows == 0) agreed = mi_window_info{h.window_bits, h.num_windows};
            same = same && h.window_bits == agreed.window_bits && h.num_windows == agreed.num_windows;
            cmax = h.window_bits > cmax ? h.window_bits : cmax;
        }
        if (!same) {
            tm.exchange_ms += ms_since(t1);
            if (attempt >= 1)
                return fail(MI_E_INVALID, "ranks disagree on the window geometry with the window size pinned (validated and unvalidated base sets mixed?)");
            // pin the largest window size FOR THIS CALL (WindowBitsGuard puts the caller's setting back) and remember it for this shard size.
            // A failure to pin it does not leave the loop: the repeated local part then disagrees again and every rank fails together above
            c->agreed_c = cmax;
            c->agreed_for_n = n;
            (void)mi_msm_set_window_bits(c->ctx, cmax);
            tm.repeats++;
            continue;
        }
        int rc = MI_OK;
        if (agreed.num_windows == 0) {
            memset(out, 0, sizeof *out);   // nobody had a point: infinity
        } else {
            // a rank without points wrote num_windows = 0 but zeroed slots: Z = 0, the fold adds infinity
            rc = fold_windows(reinterpret_cast<const Jac*>(c->h_recv) + 1, (size_t)c->n_ranks, SLOTS, &agreed, out);
            if (rc != MI_OK) return fail(rc, "fold_windows");
        }
        tm.exchange_ms += ms_since(t1);
        tm.window_bits = agreed.window_bits;
        tm.num_windows = agreed.num_windows;
        c->timing = tm;
        return MI_OK;
    }
}

}  // namespace

extern "C" {

int mi_rccl_get_unique_id(uint8_t id[MI_RCCL_UNIQUE_ID_BYTES]) {
    if (!id) return fail(MI_E_INVALID, "invalid argument");
    ncclUniqueId u;
    NCCL_RC(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MI_OK;
}

static int create_common(mi_rccl_comm** out, mi_ctx* ctx, mi_rccl_comm*& c) {
    if (!out) return fail(MI_E_INVALID, "invalid argument");
    *out = nullptr;
    if (!ctx || mi_msm_num_devices(ctx) != 1) return fail(MI_E_INVALID, "the exchange needs a single-device context (one context per rank)");
    c = new (std::nothrow) mi_rccl_comm();
    if (!c) return fail(MI_E_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->device = mi_msm_device_id(ctx, 0);
    return MI_OK;
}

int mi_rccl_comm_create(mi_rccl_comm** out, mi_ctx* ctx, const uint8_t id[MI_RCCL_UNIQUE_ID_BYTES], int n_ranks, int rank) {
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MI_E_INVALID, "invalid argument");
    mi_rccl_comm* c = nullptr;
    int rc = create_common(out, ctx, c);
    if (rc != MI_OK) return rc;
    DeviceGuard dg(c->device);
    c->n_ranks = n_ranks;
    c->rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail(r, "ncclCommInitRank");
    }
    c->owns_comm = true;
    rc = finish_create(c);
    if (rc != MI_OK) {
        mi_rccl_comm_destroy(c);
        return rc;
    }
    *out = c;
    return MI_OK;
}

int mi_rccl_comm_attach(mi_rccl_comm** out, mi_ctx* ctx, void* nccl_comm) {
    if (!nccl_comm) return fail(MI_E_INVALID, "invalid argument");
    mi_rccl_comm* c = nullptr;
    int rc = create_common(out, ctx, c);
    if (rc != MI_OK) return rc;
    DeviceGuard dg(c->device);
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    int dev = -1;
    ncclResult_t r = ncclCommCount(c->comm, &c->n_ranks);
    if (r == ncclSuccess) r = ncclCommUserRank(c->comm, &c->rank);
    if (r == ncclSuccess) r = ncclCommCuDevice(c->comm, &dev);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail(r, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
    }
    if (dev != c->device) {
        delete c;
        return fail(MI_E_INVALID, "the communicator's device is not the context's");
    }
    rc = finish_create(c);
    if (rc != MI_OK) {
        mi_rccl_comm_destroy(c);
        return rc;
    }
    *out = c;
    return MI_OK;
}

void mi_rccl_comm_destroy(mi_rccl_comm* c) {
    if (!c) return;
    {
        DeviceGuard dg(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->owns_comm && c->comm) (void)(c->aborted ? ncclCommAbort(c->comm) : ncclCommDestroy(c->comm));
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        if (c->h_recv) (void)hipHostFree(c->h_recv);
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int mi_rccl_comm_set_timeout_ms(mi_rccl_comm* c, double timeout_ms) {
    if (!c || !(timeout_ms >= 0)) return fail(MI_E_INVALID, "invalid argument");
    std::lock_guard<std::mutex> lk(c->mu);
    c->timeout_ms = timeout_ms;
    return MI_OK;
}

int mi_rccl_comm_size(const mi_rccl_comm* c) { return c ? c->n_ranks : 0; }
int mi_rccl_comm_rank(const mi_rccl_comm* c) { return c ? c->rank : -1; }

int mi_msm_g1_allgather_fold(mi_rccl_comm* c, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g1* out) {
    return allgather_fold<mi_g1>(c, d_scalars, n, scalar_fmt, out, mi_msm_g1_device_windows, mi_g1_fold_windows);
}
int mi_msm_g2_allgather_fold(mi_rccl_comm* c, const void* d_scalars, size_t n, unsigned scalar_fmt, mi_g2* out) {
    return allgather_fold<mi_g2>(c, d_scalars, n, scalar_fmt, out, mi_msm_g2_device_windows, mi_g2_fold_windows);
}

int mi_rccl_last_timing(const mi_rccl_comm* c, mi_rccl_timing* out) {
    if (!c || !out) return fail(MI_E_INVALID, "invalid argument");
    std::lock_guard<std::mutex> lk(const_cast<mi_rccl_comm*>(c)->mu);
    *out = c->timing;
    return MI_OK;
}

const char* mi_rccl_last_error(void) { return tls_err().c_str(); }

}  // extern "C"
