/* arkblst_amd.h — C ABI of the MI355X-native BLS12-381 MSM backend.
 *
 * Drop-in boundary for ONE path of nikkolasg/ark-blst: everything below
 *     <G1Projective as VariableBaseMSM>::msm      /root/reference/src/g1.rs:602-619 (CPU), 621-632 (GPU)
 *     <G2Projective as VariableBaseMSM>::msm      /root/reference/src/g2.rs:582-599 (CPU), 601-612 (GPU)
 * i.e. it replaces crate::gpu::msm + SingleMultiexpKernel (/root/reference/src/gpu.rs:101-241) and the
 * ec-gpu-gen generated kernels (/root/reference/build.rs:5-13).  The Rust binding a maintainer adds is in
 * INTEGRATION.md.  Plain pointers and sizes only; no exceptions or aborts cross this boundary.
 *
 * Data layouts are the reference's own in-memory types, passed zero-copy exactly as src/gpu.rs:149-150
 * uploads them and :185-186 reads them back (all #[repr(transparent)] newtypes, src/g1.rs:55-56,436-437):
 *     mi_fp         blst_fp        6 x u64 LE limbs, Montgomery R = 2^384, fully reduced (< p)
 *     mi_g1_affine  blst_p1_affine 96 B  (x, y); all-zero = point at infinity
 *     mi_g1         blst_p1        144 B Jacobian (X, Y, Z); Z == 0 = infinity
 *     mi_g2_affine  blst_p2_affine 192 B, coordinates in Fp2 = (c0, c1)        (src/fp2.rs:228)
 *     mi_g2         blst_p2        288 B
 *     scalars       n x 32 B little-endian: MI_SCALAR_CANONICAL = BigInteger256 integer < r (what
 *                   src/g1.rs:624-627 builds via Scalar::into_bigint), or MI_SCALAR_MONTGOMERY = blst_fr
 *                   (4 x u64, R = 2^256: the raw `Scalar` slice, src/scalar.rs:23-25) which is converted on
 *                   the GPU, removing the per-element heap round trip of src/scalar.rs:450-463,503-505.
 *
 * Error model: 0 = success, negative = MI_E_* (mi_msm_strerror).  The reference maps every GPU failure to
 * Err(0) (src/g1.rs:628-630) and panics on length mismatch (src/gpu.rs:131); here the caller passes one n.
 * Thread safety: every entry point may be called from any thread.  A context runs up to TWO MSM calls at a time (two
 * lanes per device: stream + scratch each, resident bases shared), so concurrent callers — arkworks calls the trait method
 * from rayon workers — overlap one call's sort / reduce / host tail with the other's bucket accumulation (+15-19 % points/s
 * at 2^20); further callers wait.  set_bases, normalize, (de)serialize, pairing and set_window_bits take the context
 * exclusively.  Results are deterministic (as curve points) for identical inputs and independent of the number of devices.
 */
#ifndef ARKBLST_AMD_H
#define ARKBLST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[6]; } mi_fp;
typedef struct { mi_fp x, y; } mi_g1_affine;
typedef struct { mi_fp x, y, z; } mi_g1;
typedef struct { mi_fp c[2]; } mi_fp2;
typedef struct { mi_fp2 x, y; } mi_g2_affine;
typedef struct { mi_fp2 x, y, z; } mi_g2;
typedef struct { mi_fp2 c[3]; } mi_fp6;     /* blst_fp6:  c0 + c1 v + c2 v^2, v^3 = 1 + u            (src/fp6.rs)  */
typedef struct { mi_fp6 c[2]; } mi_fp12;    /* blst_fp12: c0 + c1 w, w^2 = v; 576 B = the reference's Gt (src/fp12.rs) */

typedef struct mi_ctx mi_ctx;

enum { MI_SCALAR_CANONICAL = 0, MI_SCALAR_MONTGOMERY = 1 };

enum {
    MI_OK = 0,
    MI_E_INVALID = -1,      /* bad argument (NULL pointer, unknown scalar_fmt, n too large for resident set, a "device" pointer
                               this HIP runtime does not know) */
    MI_E_NO_DEVICE = -2,    /* no usable HIP device / bad device id */
    MI_E_HIP = -3,          /* a HIP runtime call failed; mi_msm_last_error() has the text */
    MI_E_NOMEM = -4,        /* device or host allocation failed */
    MI_E_NO_BASES = -5,     /* bases == NULL but no resident base set was uploaded */
    MI_E_UNSUPPORTED = -6,  /* host CPU lacks BMI2 / ADX (the host tail is built for them) */
    MI_E_COMM = -7,         /* an RCCL call of the multi-process exchange failed (libarkblst_amd_rccl.so, arkblst_amd_rccl.h) */
    MI_E_ABORTED = -8       /* the caller's abort check (mi_msm_set_abort_check) asked to stop: EcError::Aborted of the reference's driver */
};

/* Per-call timing of the last MSM on this context, milliseconds, measured with HIP events on the
 * library's own stream (bench.py reads these for the roofline line). */
typedef struct {
    double h2d_ms;          /* host->device copies (0 for the device-resident entry points): host bases in chunks with their conversion
                               to the device representation interleaved, host scalars in chunks on a copy stream */
    double ingest_ms;       /* unused since the base conversion runs per chunk inside h2d_ms (0) */
    double digits_ms;       /* signed-window digit extraction + histogram (host scalars: includes waiting for their chunks) */
    double scan_ms;         /* bucket offsets (prefix sum) */
    double scatter_ms;      /* bucket scatter (sort by bucket) */
    double accumulate_ms;   /* bucket accumulation (dominant kernel) + merge of split buckets */
    double reduce_ms;       /* per-chunk weighted bucket reduction */
    double combine_ms;      /* per-window combine of the chunk sums on the GPU */
    double d2h_ms;          /* window sums device->host (one Jacobian point per window) */
    double host_fold_ms;    /* CPU tail: Horner fold over the window sums, one thread */
    double total_ms;        /* wall time of the call */
    uint32_t window_bits;   /* c */
    uint32_t num_windows;   /* ceil(255 / c), one more when c divides 255 (c = 15, 17) unless the base set passed mi_msm_g{1,2}_validate_bases */
    uint64_t n;             /* points in the call */
    uint64_t accumulate_adds; /* mixed additions executed by the accumulate kernel */
    uint32_t work_items;      /* lanes of the accumulate kernel: buckets, heavy ones split into chunks */
    uint32_t max_items_per_bucket; /* 1 = no bucket was split */
    uint32_t window_groups;   /* 1 = one pass of the pipeline on one stream; > 1 = the digit windows were processed as that many groups whose sort /
                                 accumulate / reduce phases overlap on the lane's streams (mi_msm_set_pipeline).  The phase times above are then
                                 sums over the groups of intervals that overlap in time; accumulate_ms is the span from the first accumulate kernel's
                                 start to the last one's end */
    uint32_t reserved;
    /* Shader clock of the accumulate kernel(s) of THIS call, measured inside the kernel: every wave sums its s_memtime (shader cycles) and
       s_memrealtime (constant 100 MHz) ticks; GHz = 0.1 x the ratio of the two sums.  bench.py prices the roofline at this clock (round 5 replayed one from a committed profile of another box). */
    double accumulate_clock_ghz;
    uint64_t accumulate_clock_ticks, accumulate_ref_ticks;
} mi_profile;

/* Replaces Device::all()[0] + ec_gpu_gen::program! + SingleMultiexpKernel::create (src/gpu.rs:233-237,101-119),
 * which the reference repeats on EVERY call.  device_ids == NULL selects devices 0..n_devices-1;
 * n_devices == 0 selects all visible devices.  With several devices the base set is split into contiguous
 * shards, one per device, and the partial sums are folded in device order. */
int mi_msm_init(mi_ctx **out, const int *device_ids, int n_devices);
void mi_msm_destroy(mi_ctx *ctx);
int mi_msm_num_devices(const mi_ctx *ctx);
/* HIP device ordinal of device slot `slot` of the context (0 <= slot < mi_msm_num_devices), -1 otherwise. */
int mi_msm_device_id(const mi_ctx *ctx, int slot);

/* Optional: keep a base set resident in HBM across calls (an SRS).  The reference re-uploads the bases on
 * every call (src/gpu.rs:149).  Copies; no pointer is retained. */
int mi_msm_g1_set_bases(mi_ctx *ctx, const mi_g1_affine *bases, size_t n);
int mi_msm_g2_set_bases(mi_ctx *ctx, const mi_g2_affine *bases, size_t n);

/* The same from points that are ALREADY in device memory in the same form (hipMalloc'd, a torch CUDA tensor's data_ptr, the output of
 * mi_g1_deserialize_batch_device / mi_g1_normalize_batch_device): converted in place on the GPU, nothing crosses PCIe.  Synchronise the stream
 * that produced them first.  Single-device contexts (the pointer lives on ONE device). */
int mi_msm_g1_set_bases_device(mi_ctx *ctx, const void *d_bases, size_t n);
int mi_msm_g2_set_bases_device(mi_ctx *ctx, const void *d_bases, size_t n);
/* ... from Jacobian points in host memory (blst_p1 / blst_p2): normalize_batch on the GPU on the way into the resident set — the reference's
 * `batch_convert_to_mul_base` feeding `msm` (src/g1.rs:597-599 -> 604) without the affine points returning to the host in between.  Every
 * device of the context normalises its own shard. */
int mi_msm_g1_set_bases_from_jacobian(mi_ctx *ctx, const mi_g1 *points, size_t n);
int mi_msm_g2_set_bases_from_jacobian(mi_ctx *ctx, const mi_g2 *points, size_t n);
/* ... from serialized points (the encodings of mi_g1_deserialize_batch below; `compressed`: 48 / 96-byte or 96 / 192-byte units): SRS loading
 * as ONE call — decode, with `validate` Valid::check, and conversion run on the GPU, every device over its shard, and only the encodings cross
 * PCIe (48 B per G1 point instead of 48 in + 96 out + 96 in again).  All or nothing: if any encoding is rejected (malformed, off the curve or,
 * with `validate`, outside the prime-order subgroup) the call returns MI_E_INVALID, *n_rejected says how many, and the previous resident set is
 * unchanged (mi_g1_deserialize_batch gives the per-point status).  With `validate` a set that was installed is recorded as validated: MSMs over
 * it may fold signs as after mi_msm_g1_validate_bases, without that second pass.  The reference decodes point by point on the CPU
 * (src/g1.rs:398-431). */
int mi_msm_g1_set_bases_from_compressed(mi_ctx *ctx, const uint8_t *bytes, size_t n, int compressed, int validate, size_t *n_rejected);
int mi_msm_g2_set_bases_from_compressed(mi_ctx *ctx, const uint8_t *bytes, size_t n, int compressed, int validate, size_t *n_rejected);

/* Opt-in variant for a long-lived SRS: besides the bases, keep W = ceil(255 / c) tables T_j[i] = 2^(c j) * bases[i] resident
 * (W x the memory, built once on the GPU: c doublings per point and table plus one batch inversion).  Every window of a later
 * MSM over the resident set then feeds ONE bucket set: no per-window bucket reduction, no Horner doublings, and c can be
 * larger (c = 20 at 2^20 points: 13 additions per point instead of 16).  window_bits = 0 lets the time model choose c.
 * The reference has no counterpart (it re-uploads plain bases per call, src/gpu.rs:149); results are identical. */
int mi_msm_g1_set_bases_precomputed(mi_ctx *ctx, const mi_g1_affine *bases, size_t n, unsigned window_bits);
int mi_msm_g2_set_bases_precomputed(mi_ctx *ctx, const mi_g2_affine *bases, size_t n, unsigned window_bits);

/* Valid::check (is_on_curve && is_torsion_free, src/g1.rs:419-431, src/g2.rs:399-411) of the RESIDENT base set, on the GPU, every device
 * over its shard (the endomorphism tests of the bulk decoders below; ~30 ms per 2^20 G1 points, once per SRS).  *n_invalid = points that
 * are not on the curve or not in the prime-order subgroup (infinity passes).  When it is 0 the context records it, and later MSMs over the
 * resident set may recode a scalar above (r - 1) / 2 as -(r - s) — identical results on the subgroup, one digit window fewer at
 * c = 15 and c = 17.  A new set_bases clears the record.  No counterpart in the reference (its MSM takes whatever G1Affine it is given). */
int mi_msm_g1_validate_bases(mi_ctx *ctx, size_t *n_invalid);
int mi_msm_g2_validate_bases(mi_ctx *ctx, size_t *n_invalid);

/* out = sum_i scalars[i] * bases[i], i < n.   Replaces gpu::msm::<G1Affine> (src/gpu.rs:226-241) and the CPU
 * multi_exp (src/g1.rs:614-617).  bases == NULL uses the first n resident bases.  Infinity bases contribute
 * nothing (the reference's blst path fails on them, src/g1.rs:682-688).  n == 0 returns infinity.  Blocking.
 * Integer semantics, as blst's Pippenger: every base — also a curve point OUTSIDE the prime-order subgroup, which the reference hands
 * out under Validate::No (src/g1.rs:425) — is multiplied by the integer s_i (0 <= s_i < r; a canonical value >= r is reduced modulo r
 * first).  The one exception is a resident set that passed mi_msm_g{1,2}_validate_bases (above), where it makes no difference.
 * Any n: more than 2^26 points per device are processed in several passes whose sums are added (the reference's
 * calc_chunk_size path, src/gpu.rs:64-85,238-239, is unfinished). */
int mi_msm_g1(mi_ctx *ctx, const mi_g1_affine *bases, const uint8_t *scalars, size_t n, unsigned scalar_fmt,
              mi_g1 *out);
int mi_msm_g2(mi_ctx *ctx, const mi_g2_affine *bases, const uint8_t *scalars, size_t n, unsigned scalar_fmt,
              mi_g2 *out);

/* Base-set cache for the stateless call shape.  The trait method is `msm(&[G1Affine], &[Scalar])` (src/g1.rs:604): no handle, so a
 * prover with a fixed SRS passes the same host base vector on every call, and the call above converts and uploads it every time (as the
 * reference does, src/gpu.rs:149: 6.2 ms instead of 4.1 ms at 2^20 points).  With `entries` > 0 the context keeps the device form of the
 * last `entries` base vectors per group it was given (mi_msm_g{1,2} with bases != NULL, n >= 4096; least recently used out first), keyed
 * by (host pointer, n, a 128-bit fingerprint of EVERY byte of the vector).  The fingerprint runs on a few persistent helper threads of the
 * context (a third of the host's hardware threads, at most 6; ARKBLST_AMD_HASH_THREADS overrides) UNDER the GPU work of the call: an entry with
 * the same pointer and n is used speculatively and confirmed before the result leaves — 96 MiB (2^20 G1 points) take ~2 ms on five threads, the
 * call ~3 ms.  A hit runs the resident path: the bases do not cross PCIe.  A miss costs what the uncached call costs (the conversion writes
 * into the new entry).  EXACTNESS: an in-place change confined to one 8-byte word (one limb of one coordinate) changes the fingerprint with
 * certainty, any other change with probability 1 - 2^-64 or better; the chains are keyed per context from std::random_device, so whoever
 * supplies base vectors cannot aim for a collision (rounds 4-5 fingerprinted a 1024-point sample and could serve stale bases after an edit
 * outside it).  The call therefore stays a function of its arguments, as the reference's is (src/g1.rs:604).  Default: off (entries = 0); the
 * trait shims (host/ark_blst_amd.hpp, msm.py) switch it on.  The environment variable ARKBLST_AMD_BASE_CACHE=<entries>
 * (0 = off), read by mi_msm_init, overrides this call: an operator can switch the cache of a shim off without rebuilding anything.
 * Memory: 128 B (G1) / 256 B (G2) of HBM per cached point.  No counterpart in the reference. */
#define MI_BASE_CACHE_MAX 4
int mi_msm_set_base_cache(mi_ctx *ctx, unsigned entries);
int mi_msm_invalidate_base_cache(mi_ctx *ctx);
int mi_msm_base_cache_stats(const mi_ctx *ctx, uint64_t *hits, uint64_t *misses, unsigned *entries_in_use);

/* Same computation with the scalars ALREADY in device memory (hipMalloc'd or a torch CUDA tensor's data_ptr) and the bases
 * resident: nothing crosses PCIe except one Jacobian point per window.  The library reads d_scalars on its OWN stream: the
 * caller must have synchronised the stream that produced them (hipStreamSynchronize / torch.cuda.synchronize) before the
 * call.  With a multi-device context device k reads its shard [lo_k, hi_k) of the one vector: in place when the vector lives on
 * that device, through one peer copy of the shard (hipMemcpyPeerAsync) otherwise.  A pointer the runtime does not know
 * (host memory; memory of a second HIP runtime in the process) or one that is not 16-byte aligned is MI_E_INVALID. */
int mi_msm_g1_device(mi_ctx *ctx, const void *d_scalars, size_t n, unsigned scalar_fmt, mi_g1 *out);
int mi_msm_g2_device(mi_ctx *ctx, const void *d_scalars, size_t n, unsigned scalar_fmt, mi_g2 *out);

/* k MSMs over the resident base set in one call: out[j] = sum_i scalars[j][i] * B_i, i < n (host scalar vectors, all in
 * scalar_fmt).  Two run at a time on the context's lanes, so the sort / reduce / host tail of one overlap the bucket
 * accumulation of the other — the shape of a KZG prover committing to many polynomials over one SRS (the reference would
 * call msm once per polynomial and re-upload the bases each time, src/gpu.rs:149).  Stops at the first error. */
int mi_msm_g1_batch(mi_ctx *ctx, const uint8_t *const *scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g1 *out);
int mi_msm_g2_batch(mi_ctx *ctx, const uint8_t *const *scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g2 *out);
/* The same with the k scalar vectors already in device memory (an array of k device pointers, each n x 32 B; synchronise their
 * producer first, as for mi_msm_g1_device): nothing but the window sums crosses PCIe. */
int mi_msm_g1_batch_device(mi_ctx *ctx, const void *const *d_scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g1 *out);
int mi_msm_g2_batch_device(mi_ctx *ctx, const void *const *d_scalars, size_t k, size_t n, unsigned scalar_fmt, mi_g2 *out);

/* Jacobian -> affine for n points with one field inversion (product tree on the GPU).  Replaces
 * CurveGroup::normalize_batch = blstrs::G{1,2}Projective::batch_normalize (src/g1.rs:537-543, src/g2.rs:517-523), the step
 * arkworks provers run right before an MSM (ScalarMul::batch_convert_to_mul_base, src/g1.rs:597-599).  Infinity inputs
 * (Z == 0) give the all-zero affine point.  Host pointers; runs on the context's first device. */
int mi_g1_normalize_batch(mi_ctx *ctx, const mi_g1 *in, size_t n, mi_g1_affine *out);
int mi_g2_normalize_batch(mi_ctx *ctx, const mi_g2 *in, size_t n, mi_g2_affine *out);
/* Rows (f) as the caller sees them (round 6).  The host-pointer forms above and below cross PCIe in up to eight chunks: the kernels of a
 * chunk start when it has landed and its results leave while the next chunk is computed; the staging buffers belong to the context, nothing
 * is allocated per call in steady state (round 5: 52.8 ms around 1.05 ms of kernels for normalize_batch of 2^20 G1 points; now the time of
 * the two copies).  The *_device forms take and leave everything in DEVICE memory (4-byte aligned pointers on the context's device,
 * single-device contexts; synchronise the producing stream first): decode -> check -> normalize -> mi_msm_g1_set_bases_device chains never
 * touch the host.  d_in and d_out may not overlap. */
int mi_g1_normalize_batch_device(mi_ctx *ctx, const void *d_in, size_t n, void *d_out);
int mi_g2_normalize_batch_device(mi_ctx *ctx, const void *d_in, size_t n, void *d_out);

/* Bulk point (de)serialisation for G1 in the ZCash / IETF format the reference uses (src/g1.rs:358-431:
 * to_compressed / to_uncompressed, from_*_unchecked, Valid::check = is_on_curve && is_torsion_free): the SRS-loading
 * step that feeds set_bases.  `compressed`: 48-byte (1) or 96-byte (0) encodings.  `validate` = ark_serialize::Validate.
 * status[i]: 0 ok, 1 malformed encoding (the reference would unwrap() a None), 2 not on the curve, 3 not in the
 * prime-order subgroup (2 and 3 = Err(InvalidData) of check()); rejected points are written as all-zero. */
int mi_g1_deserialize_batch(mi_ctx *ctx, const uint8_t *bytes, size_t n, int compressed, int validate,
                            mi_g1_affine *out, uint8_t *status);
int mi_g1_serialize_batch(mi_ctx *ctx, const mi_g1_affine *points, size_t n, int compressed, uint8_t *bytes);
/* Same for G2 (src/g2.rs:338-411): 96-byte compressed / 192-byte uncompressed encodings, Fp2 coordinates c1 first. */
int mi_g2_deserialize_batch(mi_ctx *ctx, const uint8_t *bytes, size_t n, int compressed, int validate,
                            mi_g2_affine *out, uint8_t *status);
int mi_g2_serialize_batch(mi_ctx *ctx, const mi_g2_affine *points, size_t n, int compressed, uint8_t *bytes);
/* device-memory forms (see mi_g1_normalize_batch_device): d_out n affine points, d_status n bytes */
int mi_g1_deserialize_batch_device(mi_ctx *ctx, const void *d_bytes, size_t n, int compressed, int validate, void *d_out, void *d_status);
int mi_g2_deserialize_batch_device(mi_ctx *ctx, const void *d_bytes, size_t n, int compressed, int validate, void *d_out, void *d_status);

/* Valid::check = is_on_curve && is_torsion_free (src/g1.rs:386-396, src/g2.rs:366-376) for n affine points in host memory: what
 * ark_serialize::Valid::batch_check runs per element on the CPU (the projective form, src/g1.rs:570-579, is normalize_batch followed by this).
 * status[i]: 0 valid (infinity included), 2 not on the curve, 3 on the curve but not in the prime-order subgroup.  The endomorphism tests of
 * the decoders above, without the decoding. */
int mi_g1_check_batch(mi_ctx *ctx, const mi_g1_affine *points, size_t n, uint8_t *status);
int mi_g2_check_batch(mi_ctx *ctx, const mi_g2_affine *points, size_t n, uint8_t *status);
int mi_g1_check_batch_device(mi_ctx *ctx, const void *d_points, size_t n, void *d_status);
int mi_g2_check_batch_device(mi_ctx *ctx, const void *d_points, size_t n, void *d_status);

/* Pairing (BASELINE config #5).  Replaces <Bls12 as Pairing>::multi_miller_loop (src/pairing.rs:49-74:
 * a serial loop of blstrs::miller_loop_lines + blst_fp12_mul on one CPU thread) and final_exponentiation
 * (src/pairing.rs:76-80).  On the GPU: two lanes per pair compute the 68 line evaluations, six lanes per accumulator
 * fold them into f (several pairs share one accumulator and its squarings), then a multiplication tree; pairs are sharded
 * over the context's devices.  A pair with p[i] or q[i] at infinity (all-zero) contributes 1, as pairing.rs:58-60.  n == 0 gives 1.
 * q[i] are plain G2 affine points: the reference's G2Prepared (68 precomputed line coefficients, 19.6 KB per point) is
 * not materialised, lines are computed on the fly.  The Miller value agrees with blst's up to factors from proper
 * subfields, which the final exponentiation removes; compare Gt values, i.e. after mi_final_exponentiation.
 * mi_final_exponentiation is the O(1) host tail (one Fp12 element): f^(3 (p^12-1)/r), blst's convention. */
int mi_multi_miller_loop(mi_ctx *ctx, const mi_g1_affine *p, const mi_g2_affine *q, size_t n, mi_fp12 *out);
int mi_final_exponentiation(const mi_fp12 *f, mi_fp12 *out);
/* both steps: out = prod_i e(p[i], q[i])  (ark_ec::pairing::Pairing::multi_pairing) */
int mi_multi_pairing(mi_ctx *ctx, const mi_g1_affine *p, const mi_g2_affine *q, size_t n, mi_fp12 *out);
/* Timing of the last pairing call on this context (milliseconds, HIP events on the library's stream; first device). */
typedef struct {
    double h2d_ms;          /* points host -> device */
    double lines_ms;        /* k_miller_lines2 of the first line batch (the whole call up to 2^17 pairs) */
    double accumulate_ms;   /* k_miller_accumulate of the first line batch */
    double miller_ms;       /* all Miller kernels of the call (every batch) */
    double tree_ms;         /* Fp12 multiplication tree on the GPU */
    double host_ms;         /* last <= 4 products and, for mi_multi_pairing, the final exponentiation on the host */
    double total_ms;        /* wall time of the call */
    uint64_t n;             /* pairs */
    uint32_t pairs_per_accumulator;   /* m: pairs that share one accumulator and its squarings */
    uint32_t reserved;
} mi_pairing_profile;
int mi_pairing_last_profile(const mi_ctx *ctx, mi_pairing_profile *out);

/* Deterministic fold of partial sums (one per GPU / rank), in index order: the "all-reduce under the curve
 * group law" that follows the RCCL all-gather in the multi-process harness.  Host only. */
int mi_g1_sum(const mi_g1 *partials, size_t n, mi_g1 *out);
int mi_g2_sum(const mi_g2 *partials, size_t n, mi_g2 *out);

/* Exchange step of a one-process-per-GPU deployment (BASELINE config #3: the base set sharded over the GPUs of a node, partial
 * sums combined by an RCCL collective over xGMI) starting from DEVICE memory.  mi_msm_g1_device_windows runs the pipeline of
 * mi_msm_g1_device on this rank's shard but stops before the host tail: the num_windows per-window sums (Jacobian points in the
 * reference's form, 144 B / 288 B each, window 0 first) are left in the caller's device buffer d_out_windows (room for
 * MI_MAX_WINDOWS points, on the context's device) and nothing crosses PCIe.  The caller all-gathers the buffers of all ranks
 * with its own communicator (ncclAllGather / torch.distributed.all_gather_into_tensor: point addition is not an ncclRedOp_t, so
 * the all-reduce is all-gather + fold), copies the gathered block to the host ONCE and calls mi_g1_fold_windows, which adds the
 * ranks' sums per window in rank order and runs the Horner fold over the windows: the result equals mi_msm_g1 over the
 * concatenated shards.  All ranks must report the same mi_window_info — equal shard sizes do; otherwise fix the window size with
 * mi_msm_set_window_bits on every rank.  Single-device contexts, resident bases, at most 2^26 points per call.  The reference has
 * no counterpart (it uses Device::all()[0] only, src/gpu.rs:233-239).  Blocking: the buffer is complete when the call returns. */
#define MI_MAX_WINDOWS 37   /* ceil(255 / 7) */
typedef struct { uint32_t window_bits, num_windows; } mi_window_info;
int mi_msm_g1_device_windows(mi_ctx *ctx, const void *d_scalars, size_t n, unsigned scalar_fmt, void *d_out_windows,
                             mi_window_info *info);
int mi_msm_g2_device_windows(mi_ctx *ctx, const void *d_scalars, size_t n, unsigned scalar_fmt, void *d_out_windows,
                             mi_window_info *info);
/* windows[r * rank_stride + w] = window sum w of rank r (rank_stride >= info->num_windows).  Host only. */
int mi_g1_fold_windows(const mi_g1 *windows, size_t n_ranks, size_t rank_stride, const mi_window_info *info, mi_g1 *out);
int mi_g2_fold_windows(const mi_g2 *windows, size_t n_ranks, size_t rank_stride, const mi_window_info *info, mi_g2 *out);

/* Tuning / introspection. window_bits = 0 restores the built-in heuristic (cf. calc_window_size, src/gpu.rs:218-223). */
int mi_msm_set_window_bits(mi_ctx *ctx, unsigned window_bits);
int mi_msm_get_window_bits(const mi_ctx *ctx, unsigned *window_bits);   /* the current setting (0 = built-in heuristic) */
/* Cooperative abort, the reference driver's `maybe_abort` (src/gpu.rs:55-58,133-137: "an optional function which will be called at places where
 * it is possible to abort the multiexp calculations").  check(user) is called on the calling thread at the start of every MSM call and between the
 * passes of a call that is longer than one pass of the pipeline (more than 2^26 points per device); a non-zero return ends the call with
 * MI_E_ABORTED (EcError::Aborted) before the next pass is queued — work already on the GPU is waited for, nothing is left running.  check = NULL
 * removes it.  The function must be callable from any thread that calls into the context and must not call back into the library. */
int mi_msm_set_abort_check(mi_ctx *ctx, int (*check)(void *user), void *user);

/* Window groups of a pipelined call.  From 2^17 points on, a call over plain (not precomputed) bases processes its digit windows in groups, top
 * windows first, each group with its own scratch: the sort of group g + 1 and the bucket reduction of group g - 1 run under the accumulate kernel
 * of group g on separate streams, and the host folds the window sums of a group while the GPU works on the next.  n_groups = 0
 * restores the built-in choice; n_groups = 1 switches the pipelining off (one pass on one stream, the behaviour before round 6); otherwise
 * weights[0..n_groups) are the relative sizes of the groups, top windows first (at most 4 groups; every group gets at least one window).
 * Results do not depend on it.  The environment variable ARKBLST_AMD_PIPELINE ("0" / "off", "auto", or a comma-separated weight list such as
 * "3,5,5,3"), read by mi_msm_init, sets the initial value.  No counterpart in the reference (one launch, src/gpu.rs:172-183). */
int mi_msm_set_pipeline(mi_ctx *ctx, const unsigned *weights, unsigned n_groups);
/* Which timing events an MSM call records (every record leaves the device idle for ~6 us between two kernels — 5 % of a 2^16-point call):
 * 0 = none beyond the one the pipeline itself waits on; 1 (default) = the accumulate kernel's interval (mi_profile.accumulate_ms, total_ms,
 * host_fold_ms and the counters are filled, the other phase times are 0); 2 = every phase (digits, scatter, scan, reduce, combine, d2h, h2d).
 * No counterpart in the reference (its driver has no instrumentation, src/gpu.rs:101-241). */
int mi_msm_set_profile_level(mi_ctx *ctx, int level);
int mi_msm_last_profile(const mi_ctx *ctx, mi_profile *out);
/* Text of the CALLING THREAD's most recent failure (thread-local storage: the pointer stays valid until the same thread
 * fails again, whatever other threads do on the context). */
const char *mi_msm_last_error(const mi_ctx *ctx);
const char *mi_msm_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* ARKBLST_AMD_H */
