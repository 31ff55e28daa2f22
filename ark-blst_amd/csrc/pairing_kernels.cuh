// GPU side of `Bls12::multi_miller_loop` (/root/reference/src/pairing.rs:49-74): the reference walks the pairs one
// after the other on one CPU thread (blstrs::miller_loop_lines, then blst_fp12_mul into a running product); the pairs
// are independent, so here every pair is one lane and the running product is a multiplication tree.
//   k_miller_loop   one lane per (P, Q): f_{z,Q}(P) by the generic loop of pairing.cuh; a pair with P or Q at infinity
//                   contributes 1 (pairing.rs:58-60)
//   k_fp12_prod     one tree level: out[g] = prod in[g*K .. g*K+K)
//   k_fp12_to_raw   internal form -> the reference's blst_fp12 (12 x blst_fp, Montgomery R = 2^384)
// Device form of an Fp12: 12 slots of 16 words (14 limbs used), order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... as blst_fp12.
#pragma once
#include "msm_kernels.cuh"
#include "pairing.cuh"

namespace msmk {

using PTower = pairing::Tower<pairing::PF2>;
constexpr int FP12_WORDS = 12 * 16;
constexpr uint32_t FP12_TREE_K = 4;

__device__ __forceinline__ PTower::E12 load_fp12(const uint32_t* p) {
    PTower::E12 r;
    ec::Fp2* c = &r.c0.c0;
#pragma unroll
    for (int i = 0; i < 6; i++) ElemIO<ec::Fp2>::load(c[i], p + 32 * i);
    return r;
}
__device__ __forceinline__ void store_fp12(uint32_t* p, const PTower::E12& a) {
    const ec::Fp2* c = &a.c0.c0;
#pragma unroll
    for (int i = 0; i < 6; i++) ElemIO<ec::Fp2>::store(p + 32 * i, c[i]);
}

__global__ void __launch_bounds__(64, 1) k_miller_loop(const uint32_t* __restrict__ g1_raw, const uint32_t* __restrict__ g2_raw, uint32_t n,
                                                       uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* pr = g1_raw + (size_t)i * Geo<G1C>::RAW_AFF;
    const uint32_t* qr = g2_raw + (size_t)i * Geo<G2C>::RAW_AFF;
    uint32_t anyp = 0, anyq = 0;
#pragma unroll 4
    for (int k = 0; k < Geo<G1C>::RAW_AFF; k++) anyp |= pr[k];
#pragma unroll 4
    for (int k = 0; k < Geo<G2C>::RAW_AFF; k++) anyq |= qr[k];
    PTower::E12 f = PTower::one12();
    if (anyp != 0 && anyq != 0) {
        Fp x, y;
        fp_from_raw(x, pr);
        fp_from_raw(y, pr + 12);
        PTower::G1Pt p{fp28::fp_neg<4>(x), y};
        ec::Fp2 xq, yq;
        ElemIO<ec::Fp2>::from_raw(xq, qr);
        ElemIO<ec::Fp2>::from_raw(yq, qr + 24);
        f = PTower::miller_loop(p, xq, yq);
    }
    store_fp12(out + (size_t)i * FP12_WORDS, f);
}

__global__ void __launch_bounds__(64, 1) k_fp12_prod(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t g = blockIdx.x * 64 + threadIdx.x;
    uint32_t lo = g * FP12_TREE_K;
    if (lo >= n) return;
    uint32_t hi = lo + FP12_TREE_K < n ? lo + FP12_TREE_K : n;
    PTower::E12 acc = load_fp12(in + (size_t)lo * FP12_WORDS);
#pragma unroll 1
    for (uint32_t k = lo + 1; k < hi; k++) acc = PTower::mul12(acc, load_fp12(in + (size_t)k * FP12_WORDS));
    store_fp12(out + (size_t)g * FP12_WORDS, acc);
}

__global__ void __launch_bounds__(64) k_fp12_to_raw(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ raw) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
        ec::Fp2 c;
        ElemIO<ec::Fp2>::load(c, in + (size_t)i * FP12_WORDS + 32 * k);
        ElemIO<ec::Fp2>::to_raw(raw + (size_t)i * 144 + 24 * k, c, true);
    }
}

}  // namespace msmk
