/* arkblst_amd_rccl.h — the multi-GPU exchange step of the MSM backend, as a C ABI (libarkblst_amd_rccl.so).
 *
 * BASELINE config #3 / SURVEY.md §8(e): the base set is sharded contiguously over the GPUs of one node, ONE PROCESS PER GPU; every rank
 * runs the single-GPU pipeline over its shard and the ranks' partial sums are combined by an RCCL collective over xGMI — "all-reduce
 * under the curve group law".  Point addition is not an ncclRedOp_t, so the all-reduce is
 *     per-window sums in device memory (mi_msm_g{1,2}_device_windows)  ->  ncclAllGather  ->  ONE device-to-host copy of the gathered
 *     block  ->  mi_g{1,2}_fold_windows on every rank (ranks added per window in rank order, then the Horner fold over the windows),
 * which leaves the identical point on every rank.  This library issues those calls; it links librccl.so.1 and sits on top of the
 * public entry points of arkblst_amd.h (a single-GPU user never loads RCCL).  The reference has no counterpart: its driver takes
 * Device::all()[0] and leaves chunking as a TODO (/root/reference/src/gpu.rs:233-239).
 *
 * Deployment: rank r creates `mi_msm_init(&ctx, &device_r, 1)`, uploads ITS shard with mi_msm_g1_set_bases, and joins the communicator.
 * Launching the ranks and handing the 128-byte id from rank 0 to the others is the host program's business (torch.distributed, MPI, a
 * file, a socket).  Error model as arkblst_amd.h: 0 = ok, negative = MI_E_* (MI_E_COMM: an RCCL call failed); mi_rccl_last_error() has the
 * calling thread's text.  One call at a time per communicator (internal mutex); collective calls must be made by every rank.
 */
#ifndef ARKBLST_AMD_RCCL_H
#define ARKBLST_AMD_RCCL_H

#include "arkblst_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_rccl_comm mi_rccl_comm;

#define MI_RCCL_UNIQUE_ID_BYTES 128   /* sizeof(ncclUniqueId) */

/* ncclGetUniqueId: called by ONE rank, the bytes are handed to every rank of the communicator. */
int mi_rccl_get_unique_id(uint8_t id[MI_RCCL_UNIQUE_ID_BYTES]);

/* Collective (ncclCommInitRank on the context's device).  ctx must be a single-device context; it stays owned by the caller and must
 * outlive the communicator.  Allocates the exchange buffers once: (1 + MI_MAX_WINDOWS) x 288 B per rank on the device and pinned on the host. */
int mi_rccl_comm_create(mi_rccl_comm **out, mi_ctx *ctx, const uint8_t id[MI_RCCL_UNIQUE_ID_BYTES], int n_ranks, int rank);

/* The same around a communicator the host program already has (an ncclComm_t, passed as void *); it is NOT destroyed by
 * mi_rccl_comm_destroy.  Its device must be the context's. */
int mi_rccl_comm_attach(mi_rccl_comm **out, mi_ctx *ctx, void *nccl_comm);

/* How long a rank waits inside the exchange for its peers (milliseconds; default 60000; 0 = for ever).  When the deadline passes, or RCCL
 * reports an asynchronous error, the call aborts the communicator (ncclCommAbort: peers blocked in the same collective get an error instead
 * of waiting), returns MI_E_COMM, and every later call on this communicator fails at once with MI_E_COMM — the process is expected to exit
 * or to build a new communicator.  Round 5 waited in hipStreamSynchronize for ever. */
int mi_rccl_comm_set_timeout_ms(mi_rccl_comm *comm, double timeout_ms);

void mi_rccl_comm_destroy(mi_rccl_comm *comm);
int mi_rccl_comm_size(const mi_rccl_comm *comm);
int mi_rccl_comm_rank(const mi_rccl_comm *comm);

/* Collective.  out = sum over ALL ranks of sum_i scalars_r[i] * bases_r[i], i < n_r: rank r passes ITS n_r scalars (device memory on the
 * context's device, synchronised by the caller as for mi_msm_g1_device) over the first n_r points of ITS resident shard; every rank
 * receives the same point.  Ranks may pass different n (0 included).  All ranks must arrive at the same window size: equal shard sizes
 * do; when they do not, the call pins the largest one FOR ITS OWN DURATION (the context's mi_msm_set_window_bits setting is put back on every
 * path out) and repeats the local part once; the communicator remembers the agreed size for this shard size, so later calls agree at once.  A rank whose local part fails (say n beyond its resident shard) still joins the all-gather with a
 * failure mark: it returns its own error, every other rank MI_E_COMM naming it — nobody is left waiting inside the collective.  Blocking. */
int mi_msm_g1_allgather_fold(mi_rccl_comm *comm, const void *d_scalars, size_t n, unsigned scalar_fmt, mi_g1 *out);
int mi_msm_g2_allgather_fold(mi_rccl_comm *comm, const void *d_scalars, size_t n, unsigned scalar_fmt, mi_g2 *out);

/* Timing of the last allgather_fold on this communicator (milliseconds, host clock): the local pipeline up to the window sums in
 * device memory; everything after it (header, all-gather incl. waiting for the slowest rank, the one D2H, the host fold); and how often
 * the local part had to be repeated to agree on a window size (0 in steady state). */
typedef struct {
    double msm_ms;
    double exchange_ms;
    uint32_t window_bits, num_windows;
    uint32_t repeats;
    uint32_t bytes_per_rank;   /* size of one rank's block in the all-gather */
} mi_rccl_timing;
int mi_rccl_last_timing(const mi_rccl_comm *comm, mi_rccl_timing *out);

const char *mi_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ARKBLST_AMD_RCCL_H */
