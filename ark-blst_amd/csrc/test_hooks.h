/* Test-only entry points of libarkblst_amd_test.so (built with -DMI_TEST_HOOKS; the product library exports none of them
 * and compiles none of the code behind them).  Used by tests/ to pin the device field arithmetic against the oracle, to
 * force the pairing kernels through every sharing / batching shape at small sizes, to exercise the multi-pass split of very
 * long inputs at small sizes, and to inject allocation failures. */
#ifndef ARKBLST_AMD_TEST_HOOKS_H
#define ARKBLST_AMD_TEST_HOOKS_H
#include "../../include/arkblst_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* batch Montgomery multiply / square / add / sub on the device representation, I/O in blst_fp form. op: 0 mul, 1 sqr(a), 2 add, 3 sub */
int mi_test_fp_op(mi_ctx *ctx, int op, const mi_fp *a, const mi_fp *b, mi_fp *out, size_t n);
/* pairing: pairs per accumulator (0 = heuristic), pairs per line batch (0 = 2^17), the one-lane-per-pair first version */
int mi_test_set_pairing(mi_ctx *ctx, unsigned share, unsigned batch, int single_lane);
/* points per pass of the MSM pipeline (0 = built-in 2^26): longer calls are split and the parts added */
int mi_test_set_max_part(mi_ctx *ctx, size_t points);
/* the next `count` growing device allocations of the process fail (as hipErrorOutOfMemory would) */
void mi_test_fail_allocs(int count);
/* treat every device slot as remote from the scalar vector's device: device-resident scalars of a multi-device context are then
 * staged by peer copies even on one GPU (the path every slot but the owner's takes on a real multi-GPU node) */
int mi_test_set_no_peer(mi_ctx *ctx, int no_peer);
/* the window-size plan of an n-point call (host only, no device): out[13] = c, windows, bucket sets, buckets per logical lane of the cooperative
 * reduce (coop_L), buckets per reduce chunk, logT, lo_bits, serial reduce, chunks per window, buckets (hi, lo), chunks, buckets per lane of the
 * serial reduce.  group 0 = G1, 1 = G2; shared bit 0 = precomputed tables, bit 1 = sign fold (a validated resident set); c = 0 in out[0]: no usable plan */
int mi_test_plan(size_t n, unsigned forced_c, int group, int shared, size_t stride, uint32_t *out);
#ifdef __cplusplus
}
#endif
#endif
