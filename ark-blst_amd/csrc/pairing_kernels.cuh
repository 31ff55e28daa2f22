// GPU side of `Bls12::multi_miller_loop` (/root/reference/src/pairing.rs:49-74): the reference walks the pairs one
// after the other on one CPU thread (blstrs::miller_loop_lines, then blst_fp12_mul into a running product); the pairs
// are independent, so here the work is spread by data shape:
//   k_miller_lines2      two lanes per pair: the G2 point walk in Fp2 and the 68 evaluated lines (default path)
//   k_miller_accumulate  six lanes per accumulator: f <- f^2 * l_1 ... l_m per step for m pairs, coefficients in LDS
//   k_fp12_prod          one multiplication-tree level, six lanes per output: out[g] = prod in[g*K .. g*K+K)
//   k_fp12_to_raw        internal form -> the reference's blst_fp12 (12 x blst_fp, Montgomery R = 2^384)
//   k_miller_loop        the first version, one lane per pair for the whole loop (generic pairing.cuh code; 4 KB of
//                        scratch per lane) — compiled into test builds only (MI_TEST_HOOKS) as a second implementation to cross-check
// A pair with P or Q at infinity contributes 1 (pairing.rs:58-60).
// Device form of an Fp12: 12 slots of 16 words (14 limbs used), order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... as blst_fp12.
#pragma once
#include "kernels_common.cuh"
#include "coop_fp2.cuh"
#include "pairing.cuh"

namespace msmk {

using PTower = pairing::Tower<pairing::PF2>;
constexpr int FP12_WORDS = 12 * 16;
constexpr uint32_t FP12_TREE_K = 2;   // a tree level is one latency chain of K - 1 products per wave: (K - 1) log_K(n) is smallest at K = 2
                                      // (2^16 pairs: 0.25 ms for 12 levels of 2 against 0.31 ms for 6 levels of 4)

__device__ __forceinline__ PTower::E12 load_fp12(const uint32_t* p) {   // member by member: no pointer walk across struct members
    using IO = ElemIO<ec::Fp2>;
    PTower::E12 r;
    IO::load(r.c0.c0, p); IO::load(r.c0.c1, p + 32); IO::load(r.c0.c2, p + 64);
    IO::load(r.c1.c0, p + 96); IO::load(r.c1.c1, p + 128); IO::load(r.c1.c2, p + 160);
    return r;
}
__device__ __forceinline__ void store_fp12(uint32_t* p, const PTower::E12& a) {
    using IO = ElemIO<ec::Fp2>;
    IO::store(p, a.c0.c0); IO::store(p + 32, a.c0.c1); IO::store(p + 64, a.c0.c2);
    IO::store(p + 96, a.c1.c0); IO::store(p + 128, a.c1.c1); IO::store(p + 160, a.c1.c2);
}

#if defined(MI_TEST_HOOKS)   // the first version, one lane per pair: a second implementation for cross-checks (test builds only)
__global__ void __launch_bounds__(64, 2) k_miller_loop(const uint32_t* __restrict__ g1_raw, const uint32_t* __restrict__ g2_raw, uint32_t n,
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
#endif


// ------------------------------------------------------------------------------------------------------------
// Two-kernel Miller loop (the default).  The one-lane-per-pair kernel above keeps an Fp12 (168 words) plus its
// Karatsuba temporaries per lane: 4 KB of scratch per lane, 120 GB of HBM traffic for 2^16 pairs
// (profiles/r01_e_pairing_pmc_summary.json) — it is bound by scratch traffic, not by arithmetic.  Split instead:
//   k_miller_lines2      TWO lanes per pair walk T = [.]Q in Fp2 only (state: 6 + 4 + 2 field elements, one Fp2 component
//                        per lane, see CoopF2) and write the 68 evaluated line coefficients (c0, c1, c4) — what the
//                        reference keeps in a G2Prepared, already multiplied by xP / yP.  Layout [line][pair][3] Fp2
//                        slots of 32 words.
//   k_miller_accumulate  SIX lanes per pair: lane k owns coefficient k of f in the flat basis f = sum f_k w^k
//                        (Fp12 = Fp2[w]/(w^6 - xi); tower slot of w^k: c_{k&1}.c_{k>>1}).  The six coefficients of a
//                        pair sit in LDS; a product h = f g is h_k = sum_i xi^[i>k] f_i g_{(k-i) mod 6}: every lane
//                        accumulates its six Fp2 products into two sets of 64-bit columns and reduces ONCE per
//                        component (12 products per reduction).  No scratch, ~200 registers, two waves per SIMD.
//                        A line is sparse (w^0, w^2, w^3): three terms.
// Ten accumulators per wave (lanes 60..63 idle), each folding m consecutive pairs with ONE squaring per step.  A pair with P
// or Q at infinity gets the lines (1, 0, 0): f stays 1.
constexpr int MILLER_LINES = 68;          // 63 doublings + 5 additions for |z| = 0xd201000000010000
constexpr int MILLER_GROUPS = 10;         // pairs per wave in k_miller_accumulate
constexpr int LDS_COEFF_WORDS = 36;       // one Fp2 coefficient in LDS (k_fp12_prod): 2 x 16 words + 4 words of padding (bank spread)
// k_miller_accumulate keeps FIVE values per coefficient f_j = (c0, c1) (round 4): c0 | c1 | s = c0 + c1 | d = c0 - c1 + 4p | d + s.
// A consumer's Karatsuba operands (a0', a1', a0' + a1') for the term f_j g are (c0, c1, s), or (d, s, d + s) when the index wrapped
// (times xi = 1 + u: (c0 - c1, c0 + c1)): three LDS reads at computed addresses instead of a subtraction, two additions, their
// carry passes and two 14-limb selects PER TERM in each of the three lanes that read f_j — the owner forms s, d, d + s once per
// product (about 420 of the 1415 non-multiply instructions of a line product).
constexpr int ACC_COEFF_WORDS = 5 * 16 + 4;

using CTower = pairing::Tower<CoopF2>;

// Two lanes per pair (see CoopF2).  Lines of a pair with P or Q at infinity are overwritten with (1, 0, 0) at store time,
// so every lane runs the same instruction stream (the DPP exchange needs both lanes of a pair anyway).
//
// Register diet (the first version of this kernel spilled 442 registers: 1.4 KB of scratch per lane, half of its HBM traffic):
// the loop-invariant values (-xP, yP of the pair; this lane's component of xQ, yQ) live in LDS and are read where a formula
// uses them, every line coefficient is stored the moment it is computed, and the formulas (the same as
// pairing::Tower::line_dbl / line_add, whose bounds tests/host/pairing_bounds.cpp checks) are ordered so that T's old
// coordinates die early.  What stays in registers across a step is T (3 x 14 limbs per lane).
constexpr int LINES_LDS_WORDS = 64 * 2 * 16 + 32 * 2 * 16;   // per block of 64 lanes: (xQ, yQ) component per lane, (-xP, yP) per pair

struct LineSink {   // where the three coefficients of one line go: [line][pair][3] Fp2 slots of 32 words, this lane's half
    uint32_t* o;
    bool valid, inf;
    uint32_t h;
    __device__ __forceinline__ void put(int slot, const Fp& c) const {
        if (!valid) return;
        Fp dflt = slot == 0 ? CoopF2::one() : fp28::fp_zero();
        store_fp16(o + 32 * slot, fp28::fp_select(inf, c, dflt));
    }
};

// tangent line at T (scaled by 2YZ) evaluated at P, then T <- 2T: the formulas of pairing::Tower::line_dbl (round 4: 4 S + 3 M + one
// fused difference of squares), ordered so that a coefficient is stored the moment it is computed.  T <= 6p in, < 4p out.
__device__ __forceinline__ void coop_line_dbl(ec::Proj<CoopF2>& T, const uint32_t* lds_p, const LineSink& out) {
    using F2 = CoopF2;
    Fp B = F2::sqr(T.y);                                                     // Y^2
    Fp C = F2::sqr(T.z);                                                     // Z^2
    Fp H = F2::sub<8>(F2::sqr(F2::add(T.y, T.z)), F2::add(B, C));            // 2 Y Z                  < 10p
    {
        Fp yp;
        load_fp16(yp, lds_p + 16);
        out.put(2, F2::mul_fp(H, yp));                                       // c4 = 2 Y Z yP
    }
    Fp E = F2::mul_b3(C);                                                    // b3 Z^2 = 3b' Z^2
    out.put(0, F2::sub<4>(B, E));                                            // c0 = Y^2 - 3b' Z^2     < 6p
    {
        Fp nx;
        load_fp16(nx, lds_p);
        out.put(1, F2::mul_fp(F2::mul3(F2::sqr(T.x)), nx));                  // c1 = -3 X^2 xP
    }
    Fp xy = F2::mul(T.x, T.y);
    Fp E3 = F2::mul3(E);                                                     //                        < 6p
    T.x = F2::dbl(F2::mul(F2::sub<8>(B, E3), xy));                           // 2 XY (B - 3E)          < 4p
    T.z = F2::mul(F2::dbl(F2::dbl(H)), B);                                   // 4 H B = 8 Y^3 Z        < 2p
    T.y = F2::sqr_sub12sqr(F2::add(B, E3), E);                               // (B + 3E)^2 - 12 E^2    < 2p
}
// line through T and Q (scaled by X - xQ Z) evaluated at P, then T <- T + Q
__device__ __forceinline__ void coop_line_add(ec::Proj<CoopF2>& T, const uint32_t* lds_q, const uint32_t* lds_p, const LineSink& out) {
    using F2 = CoopF2;
    Fp xq, yq;
    load_fp16(xq, lds_q);
    load_fp16(yq, lds_q + 16);
    Fp N = F2::sub<4>(T.y, F2::mul(yq, T.z));                                // < 10p
    Fp D = F2::sub<4>(T.x, F2::mul(xq, T.z));
    out.put(0, F2::sub<4>(F2::mul(N, xq), F2::mul(D, yq)));                  // N xQ - D yQ            < 6p
    {
        Fp nx, yp;
        load_fp16(nx, lds_p);
        load_fp16(yp, lds_p + 16);
        out.put(1, F2::mul_fp(N, nx));
        out.put(2, F2::mul_fp(D, yp));
    }
    ec::Proj<CoopF2> q = ec::proj_from_affine<CoopF2>(xq, yq);
    ec::proj_add<CoopF2>(T, q);
}

// Line buffer layout: [block of `blk` pairs][line][pair in block][3 Fp2 slots of 32 words].  blk = the pairs one wave of
// k_miller_accumulate folds (ten accumulators of m pairs), so a wave's 68 reads walk ONE contiguous region (1.8 MB at m = 7)
// instead of 68 regions n * 384 B apart: with [line][pair] over all pairs the 3.4 GB of 2^17 pairs fell off the TLB reach and
// every line cost twice what it costs at 2^16 pairs.
__device__ __forceinline__ size_t line_base_words(uint32_t pair, uint32_t blk) {   // line 0 of `pair`; line l is l * blk * 96 words on
    const uint32_t b = pair / blk, i = pair - b * blk;
    return ((size_t)b * MILLER_LINES * blk + i) * 96;
}

__global__ void __launch_bounds__(64, 2) k_miller_lines2(const uint32_t* __restrict__ g1_raw, const uint32_t* __restrict__ g2_raw, uint32_t n,
                                                         uint32_t blk, uint32_t* __restrict__ lines) {
    __shared__ uint32_t cst[LINES_LDS_WORDS];
    const uint32_t lane = threadIdx.x, h = lane & 1u;
    const uint32_t pair = (blockIdx.x * 64 + lane) >> 1;
    const bool valid = pair < n;
    const uint32_t i = valid ? pair : n - 1;
    const uint32_t* pr = g1_raw + (size_t)i * Geo<G1C>::RAW_AFF;
    const uint32_t* qr = g2_raw + (size_t)i * Geo<G2C>::RAW_AFF;
    uint32_t anyp = 0, anyq = 0;
#pragma unroll 4
    for (int k = 0; k < Geo<G1C>::RAW_AFF; k++) anyp |= pr[k];
#pragma unroll 4
    for (int k = 0; k < Geo<G2C>::RAW_AFF; k++) anyq |= qr[k];
    uint32_t* lds_q = cst + lane * 32;                       // this lane's component of xQ | yQ
    uint32_t* lds_p = cst + 64 * 32 + (lane >> 1) * 32;      // -xP | yP of the pair (both lanes write the same values)
    ec::Proj<CoopF2> T;
    {
        Fp x, y, xq, yq;
        fp_from_raw(x, pr);
        fp_from_raw(y, pr + 12);
        fp_from_raw(xq, qr + 12 * h);              // this lane's component of x_Q, y_Q
        fp_from_raw(yq, qr + 24 + 12 * h);
        store_fp16(lds_p, fp28::fp_neg<4>(x));
        store_fp16(lds_p + 16, y);
        store_fp16(lds_q, xq);
        store_fp16(lds_q + 16, yq);
        T = ec::proj_from_affine<CoopF2>(xq, yq);
    }
    __syncthreads();
    LineSink out{nullptr, valid, anyp == 0 || anyq == 0, h};
    uint32_t* const line0 = lines + line_base_words(i, blk) + 16 * h;
    const size_t line_stride = (size_t)blk * 96;
    int line = 0;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        out.o = line0 + line * line_stride;
        coop_line_dbl(T, lds_p, out);
        line++;
        if ((fp28c::Z_ABS >> b) & 1) {
            out.o = line0 + line * line_stride;
            coop_line_add(T, lds_q, lds_p, out);
            line++;
        }
    }
}

// columns += a * b (196 multiply-adds, no carries: see fp28.cuh)
__device__ __forceinline__ void fp_acc(uint64_t (&c)[2 * fp28::NL], const Fp& a, const Fp& b) {
#pragma unroll
    for (int i = 0; i < fp28::NL; i++) {
#pragma unroll
        for (int j = 0; j < fp28::NL; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
}
__device__ __forceinline__ ec::Fp2 lds_load_fp2(const uint32_t* p) {
    ec::Fp2 r;
    load_fp16(r.c0, p);
    load_fp16(r.c1, p + 16);
    return r;
}
__device__ __forceinline__ void lds_store_fp2(uint32_t* p, const ec::Fp2& a) {
    store_fp16(p, a.c0);
    store_fp16(p + 16, a.c1);
}
// Karatsuba accumulation of Fp2 products (round 3).  A lane's coefficient h_k = sum_t xi^[wrapped] a_t g_t is collected in THREE
// column sets  V0 += a0' g0,  V1 += a1' g1,  V2 += (a0' + a1')(g0 + g1)   (a' = a or xi a = (a0 - a1, a0 + a1))  — three products
// per term instead of four — and recombined ONCE per coefficient:  c0 = V0 - V1 (+ bias),  c1 = V2 - V0 - V1.  The recombined
// columns are signed (a column of V1 may exceed the same column of V0): the reduction below carries with arithmetic shifts.
// c1 = a0' g1 + a1' g0 is non-negative as a number; c0 gets the bias p 2^388 (a multiple of p: P[j] << 24 added to column 13 + j),
// which exceeds every V1 reached here (<= 144 p^2 ~ 2^387.9 p) and keeps the reduction's input below 373 p^2 of the 2520 p^2 it
// may take, so every output is < 2p as before.  Operand bounds: a' <= (12p, 8p) in N-form, g <= 6p in N-form (a line's c0; exact
// coefficients are < 2p), sums a0' + a1' and g0 + g1 normalised: <= 4 terms x 14 products of < 2^56.01 per column set (see fp2_kara_reduce).
// 168 registers of columns: the kernels below are built for ONE wave per SIMD (512 registers) — which costs nothing, a lone wave
// of a 64-thread workgroup issues at the SIMD's full rate (tools/ubench_fp52.hip: 2484 vs 2447 cycles per multiplication).
struct KaraCols {
    uint64_t v0[2 * fp28::NL], v1[2 * fp28::NL], v2[2 * fp28::NL];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int t = 0; t < 2 * fp28::NL; t++) { v0[t] = 0; v1[t] = 0; v2[t] = 0; }
    }
};
__device__ __forceinline__ void fp2_acc_term(KaraCols& c, const ec::Fp2& a, bool wrapped, const ec::Fp2& g) {
    Fp xa0 = fp28::fp_sub<8>(a.c0, a.c1), xa1 = fp28::fp_add(a.c0, a.c1);
    Fp a0 = fp28::fp_select(wrapped, a.c0, xa0), a1 = fp28::fp_select(wrapped, a.c1, xa1);
    fp_acc(c.v0, a0, g.c0);
    fp_acc(c.v1, a1, g.c1);
    fp_acc(c.v2, fp28::fp_add(a0, a1), fp28::fp_add_lazy(g.c0, g.c1));   // g0 + g1 without a carry pass (see fp2_acc_term_pre)
}
// the five values of one coefficient (see ACC_COEFF_WORDS); f exact (< 2p per component): s < 4p, d < 6p, d + s < 10p, all N-form
__device__ __forceinline__ void lds_store_variants(uint32_t* p, const ec::Fp2& f) {
    const Fp sum = fp28::fp_add(f.c0, f.c1), dif = fp28::fp_sub<4>(f.c0, f.c1);
    store_fp16(p, f.c0);
    store_fp16(p + 16, f.c1);
    store_fp16(p + 32, sum);
    store_fp16(p + 48, dif);
    store_fp16(p + 64, fp28::fp_add(dif, sum));
}
// cols += f_j * xi^[wrapped] * g with the operands of f_j read ready-made from its five-value LDS record
__device__ __forceinline__ void fp2_acc_term_pre(KaraCols& c, const uint32_t* rec, bool wrapped, const ec::Fp2& g) {
    Fp a0, a1, as;
    load_fp16(a0, rec + (wrapped ? 48 : 0));
    load_fp16(a1, rec + (wrapped ? 32 : 16));
    load_fp16(as, rec + (wrapped ? 64 : 32));
    fp_acc(c.v0, a0, g.c0);
    fp_acc(c.v1, a1, g.c1);
    fp_acc(c.v2, as, fp28::fp_add_lazy(g.c0, g.c1));   // no carry pass: limbs <= 2 (2^28 + 64), a V2 column <= the V0 + V1 column bound
}
// Montgomery reduction of SIGNED 64-bit columns (|column| < 2^62, value in [0, 2^392 p)): fp28::fp_mont_reduce with arithmetic carries
__device__ __forceinline__ Fp fp_mont_reduce_signed(int64_t (&c)[2 * fp28::NL]) {
    using namespace fp28;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t m = ((uint32_t)c[i] * PINV) & MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (int64_t)((uint64_t)m * P[j]);
        c[i + 1] += c[i] >> W;   // exact: the low 28 bits are zero
    }
    Fp r;
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        int64_t v = c[NL + k] + carry;
        r.l[k] = (uint32_t)v & MASK;
        carry = v >> W;
    }
    r.l[NL - 1] |= (uint32_t)carry << W;
    return r;
}
// Column ranges (tests/test_host_model.py::test_karatsuba_column_and_bias_bounds restates them with big integers).  The sums
// a0' + a1' and g0 + g1 are carry-normalised, so the Karatsuba identity holds for the VALUE, not column by column: both recombined
// column sets are signed.  With at most FOUR terms per reduction: |c0 column| <= max(V0, V1) column <= 4 x 14 x 2^56.01 = 2^61.8,
// |c1 column| <= V0 + V1 column <= 2^62.8, plus the bias (< 2^52) and the reduction's own 14 x 2^56: inside +-2^63.  (Six terms —
// the Fp12 tree — would reach 2^63.4: k_fp12_prod keeps the four-product form with unsigned columns, fp2_acc_term4 below.)
__device__ __forceinline__ ec::Fp2 fp2_kara_reduce(const KaraCols& c) {
    int64_t c0[2 * fp28::NL], c1[2 * fp28::NL];
#pragma unroll
    for (int t = 0; t < 2 * fp28::NL; t++) {
        const uint64_t w = c.v0[t] + c.v1[t];
        c0[t] = (int64_t)(c.v0[t] - c.v1[t]);
        c1[t] = (int64_t)(c.v2[t] - w);
    }
#pragma unroll
    for (int j = 0; j < fp28::NL; j++) c0[fp28::NL - 1 + j] += (int64_t)((uint64_t)fp28::P[j] << 24);   // + p 2^388
    return ec::Fp2{fp_mont_reduce_signed(c0), fp_mont_reduce_signed(c1)};
}
// (c0, c1) += a * g * xi^[wrapped] as FOUR products into two unsigned column sets (the form of rounds 1-2; the Fp12 tree's six terms):
// a <= 4p, g exact (< 2p): xi a <= (12p, 8p) in N-form, 4p - g1 <= 4p: a term adds <= 56 p^2 to a component, six terms per reduction
// (limit 2520 p^2); <= 168 column terms of < 2^56.01 plus the reduction's 2^59.9 stay below 2^64.
__device__ __forceinline__ void fp2_acc_term4(uint64_t (&c0)[2 * fp28::NL], uint64_t (&c1)[2 * fp28::NL], const ec::Fp2& a, bool wrapped,
                                              const ec::Fp2& g) {
    Fp xa0 = fp28::fp_sub<8>(a.c0, a.c1), xa1 = fp28::fp_add(a.c0, a.c1);
    Fp a0 = fp28::fp_select(wrapped, a.c0, xa0), a1 = fp28::fp_select(wrapped, a.c1, xa1);
    Fp ng1 = fp28::fp_neg<4>(g.c1);
    fp_acc(c0, a0, g.c0);
    fp_acc(c0, a1, ng1);
    fp_acc(c1, a0, g.c1);
    fp_acc(c1, a1, g.c0);
}

// `m` consecutive pairs share one accumulator: f <- f^2 * l_1 * ... * l_m per step (the multi-Miller-loop trick: one squaring
// for m pairs, as blst's miller_loop_n does); out[g] = product of the Miller values of pairs [g m, g m + m).
__global__ void __launch_bounds__(64, 1) k_miller_accumulate(const uint32_t* __restrict__ lines, uint32_t n, uint32_t m, uint32_t blk,
                                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t fs[(MILLER_GROUPS + 1) * 6 * ACC_COEFF_WORDS];   // + one dummy group for the idle lanes
    const uint32_t lane = threadIdx.x;
    const uint32_t grp = lane / 6, k = lane - grp * 6;                   // lanes 60..63: group 10 (dummy)
    const uint32_t ngroups = (n + m - 1) / m;
    const uint32_t g_idx = blockIdx.x * MILLER_GROUPS + grp;
    const bool valid = grp < MILLER_GROUPS && g_idx < ngroups;
    const uint32_t first = (valid ? g_idx : ngroups - 1) * m;            // idle lanes shadow a real group, write nothing
    uint32_t* fg = fs + grp * 6 * ACC_COEFF_WORDS;
    ec::Fp2 own = k == 0 ? ec::Fp2Ops::one() : ec::Fp2Ops::zero();
    lds_store_variants(fg + k * ACC_COEFF_WORDS, own);
    __syncthreads();
    int line = 0;
    // squaring terms of this lane: the i <= j of the pairs i + j = k (mod 6): four of them for even k, three for odd k
    uint32_t sq_tab = 0, sq_cnt = 0;
    for (uint32_t i = 0; i < 6; i++) {
        uint32_t j = (k + 6 - i) % 6;
        if (i <= j) sq_tab |= i << (3 * sq_cnt++);
    }
    // h_k = sum over the three non-zero line coefficients at w^0, w^2, w^3.  The three coefficients of the NEXT pair's line are
    // requested before the current pair's products are computed (84 more registers, which one wave per SIMD has): a lone wave has
    // nobody to cover a load's latency, and with the loads inside a rolled term loop every term waited for its own.
    auto line_ptr = [&](uint32_t q) {
        const bool live = first + q < n;
        return lines + line_base_words(live ? first + q : n - 1, blk) + (size_t)line * blk * 96;
    };
    auto mul_line = [&]() {   // f <- f * (line `line` of pair first + q) for q < m; a pair beyond n multiplies by one
        ec::Fp2 g0, g1, g2;
        {
            const uint32_t* lp = line_ptr(0);
            ElemIO<ec::Fp2>::load(g0, lp); ElemIO<ec::Fp2>::load(g1, lp + 32); ElemIO<ec::Fp2>::load(g2, lp + 64);
        }
#pragma unroll 1
        for (uint32_t q = 0; q < m; q++) {
            const bool live = first + q < n;
            ec::Fp2 n0 = g0, n1 = g1, n2 = g2;
            if (q + 1 < m) {
                const uint32_t* lp = line_ptr(q + 1);
                ElemIO<ec::Fp2>::load(n0, lp); ElemIO<ec::Fp2>::load(n1, lp + 32); ElemIO<ec::Fp2>::load(n2, lp + 64);
            }
            KaraCols cols;
            cols.clear();
            {   // terms at w^0, w^2, w^3: f_j with j = k, k - 2, k - 3 (mod 6), times xi when the index wrapped
                int j = (int)k;
                fp2_acc_term_pre(cols, fg + j * ACC_COEFF_WORDS, false, g0);
                j = (int)k - 2;
                bool wrapped = j < 0;
                if (wrapped) j += 6;
                fp2_acc_term_pre(cols, fg + j * ACC_COEFF_WORDS, wrapped, g1);
                j = (int)k - 3;
                wrapped = j < 0;
                if (wrapped) j += 6;
                fp2_acc_term_pre(cols, fg + j * ACC_COEFF_WORDS, wrapped, g2);
            }
            ec::Fp2 r = fp2_kara_reduce(cols);
            own = ec::Fp2Ops::select(live, own, r);
            __syncthreads();                                             // every lane has read the old f
            lds_store_variants(fg + k * ACC_COEFF_WORDS, own);
            __syncthreads();
            g0 = n0; g1 = n1; g2 = n2;
        }
        line++;
    };
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        {   // f <- f^2 : h_k = sum over unordered {i, j}, i + j = k (mod 6), of (2 - [i == j]) xi^[i + j >= 6] f_i f_j
            KaraCols cols;
            cols.clear();
#pragma unroll 1
            for (uint32_t t = 0; t < 4; t++) {
                uint32_t i = (sq_tab >> (3 * t)) & 7u;
                int j = (int)k - (int)i;
                if (j < 0) j += 6;
                ec::Fp2 a = lds_load_fp2(fg + i * ACC_COEFF_WORDS);
                ec::Fp2 a2 = ec::Fp2Ops::add(a, a);
                a = ec::Fp2Ops::select((int)i != j, a, a2);
                a = ec::Fp2Ops::select(t >= sq_cnt, a, ec::Fp2Ops::zero());          // odd k has three terms only
                fp2_acc_term(cols, a, i + (uint32_t)j >= 6, lds_load_fp2(fg + j * ACC_COEFF_WORDS));
            }
            own = fp2_kara_reduce(cols);
            __syncthreads();
            lds_store_variants(fg + k * ACC_COEFF_WORDS, own);
            __syncthreads();
        }
        mul_line();
        if ((fp28c::Z_ABS >> b) & 1) mul_line();
    }
    // z < 0: conjugate (w -> -w: odd coefficients negated); flat w^k -> tower slot c_{k&1}.c_{k>>1}
    if (k & 1) {   // exact again (the tree takes coefficients < 2p): multiplied by one INLINE — this kernel uses AGPRs, so it must not call
        const ec::Fp2 neg = ec::Fp2Ops::neg<4>(own);
        own = ec::Fp2{fp28::fp_mul(neg.c0, fp28::fp_one()), fp28::fp_mul(neg.c1, fp28::fp_one())};
    }
    if (valid) ElemIO<ec::Fp2>::store(out + (size_t)g_idx * FP12_WORDS + ((k & 1) * 3 + (k >> 1)) * 32, own);
}

// One level of the multiplication tree, six lanes per output: out[g] = prod in[g*K .. g*K+K).  Same scheme as
// k_miller_accumulate: the running product's coefficients in LDS, the factor's coefficients read from HBM.  Built for one wave per
// SIMD: a level has at most a few hundred waves (468 for the 9363 Miller values of 2^16 pairs), and capped at 256 registers the
// kernel spilled 29 of them.
__global__ void __launch_bounds__(64, 1) k_fp12_prod(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t fs[(MILLER_GROUPS + 1) * 6 * LDS_COEFF_WORDS];
    const uint32_t lane = threadIdx.x;
    const uint32_t grp = lane / 6, k = lane - grp * 6;
    const uint32_t g = blockIdx.x * MILLER_GROUPS + grp;
    const uint32_t groups = (n + FP12_TREE_K - 1) / FP12_TREE_K;
    const bool valid = grp < MILLER_GROUPS && g < groups;
    const uint32_t gc = valid ? g : groups - 1;
    const uint32_t lo = gc * FP12_TREE_K, hi = lo + FP12_TREE_K < n ? lo + FP12_TREE_K : n;
    auto slot = [](uint32_t coeff) { return ((coeff & 1u) * 3u + (coeff >> 1)) * 32u; };   // flat w^coeff -> tower slot
    uint32_t* fg = fs + grp * 6 * LDS_COEFF_WORDS;
    ec::Fp2 own;
    ElemIO<ec::Fp2>::load(own, in + (size_t)lo * FP12_WORDS + slot(k));
    lds_store_fp2(fg + k * LDS_COEFF_WORDS, own);
    __syncthreads();
#pragma unroll 1
    for (uint32_t e = 1; e < FP12_TREE_K; e++) {
        // every group runs all steps (barriers are block-wide); a group that is out of elements multiplies by in[hi-1]
        // again and discards the result
        const bool live = lo + e < hi;
        const uint32_t* gp = in + (size_t)(live ? lo + e : hi - 1) * FP12_WORDS;
        uint64_t c0[2 * fp28::NL], c1[2 * fp28::NL];
#pragma unroll
        for (int t = 0; t < 2 * fp28::NL; t++) { c0[t] = 0; c1[t] = 0; }
#pragma unroll 1
        for (int i = 0; i < 6; i++) {
            int j = (int)k - i;
            bool wrapped = j < 0;
            if (wrapped) j += 6;
            ec::Fp2 gj;
            ElemIO<ec::Fp2>::load(gj, gp + slot((uint32_t)j));
            fp2_acc_term4(c0, c1, lds_load_fp2(fg + i * LDS_COEFF_WORDS), wrapped, gj);
        }
        ec::Fp2 r{fp28::fp_mont_reduce(c0), fp28::fp_mont_reduce(c1)};
        own = ec::Fp2Ops::select(live, own, r);
        __syncthreads();
        lds_store_fp2(fg + k * LDS_COEFF_WORDS, own);
        __syncthreads();
    }
    if (valid) ElemIO<ec::Fp2>::store(out + (size_t)g * FP12_WORDS + slot(k), own);
}

__global__ void __launch_bounds__(64, 2) k_fp12_to_raw(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ raw) {
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
