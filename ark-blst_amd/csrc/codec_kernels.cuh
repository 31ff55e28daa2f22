// Bulk point (de)serialisation kernels and the field-level test hook.  Compiled in points.hip only.
#pragma once
#include "kernels_common.cuh"

namespace msmk {

// ---------------------------------------------------------------------------------------------- G1 point decoding
// Bulk CanonicalDeserialize + Valid::check for G1 (/root/reference/src/g1.rs:386-431): ZCash/IETF encoding
// (48-byte compressed / 96-byte uncompressed, big-endian, flag bits 0x80 compressed, 0x40 infinity, 0x20 y is the
// lexicographically larger root) -> blst_p1_affine, with per-point status instead of the reference's unwrap():
//   0 ok, 1 malformed encoding (flags, x >= p, no square root), 2 not on the curve, 3 not in the prime-order subgroup.
// Decompression: y = (x^3 + 4)^((p+1)/4).  Subgroup check (blstrs is_torsion_free [ext]) by the endomorphism test
// (beta x, y) == -[z^2](x, y), z = 0xd201000000010000 (M. Scott, eprint 2021/1130): two 64-bit double-and-add ladders
// on the complete projective formulas instead of a 255-bit multiplication by r.
__device__ __forceinline__ bool words_lt_p(const uint32_t (&w)[12]) {
    uint64_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint64_t v = (uint64_t)w[k] - fp28c::P32[k] - borrow;
        borrow = (v >> 32) & 1;
    }
    return borrow != 0;
}
__device__ __forceinline__ void be48_to_words(uint32_t (&w)[12], const uint8_t* b, uint32_t top_mask) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint8_t* q = b + 44 - 4 * k;  // word k = bytes [44-4k, 48-4k) big-endian
        w[k] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    w[11] &= top_mask;
}
// canonical integer of an internal value (< 50p), exact 28-bit limbs
__device__ __forceinline__ Fp fp_to_canonical(const Fp& a) {
    Fp one_int = fp28::fp_zero();
    one_int.l[0] = 1;
    return fp28::fp_canon_2p(fp28::fp_mul_call(a, one_int));
}
__device__ __forceinline__ bool fp_equal(const Fp& a, const Fp& b) {  // a == b (mod p), a, b < 15p
    return fp28::fp_is_zero_any(fp28::fp_sub<16>(a, b));
}

// a^e for one of the two constant square-root exponents, by a sliding window of four bits (round 6): the schedule (tools/gen_fp28_consts.py sw4)
// is a list of (squarings << 4 | table index) over the odd powers a, a^3 .. a^15 — 378 squarings + 79 multiplications + 8 for the table where
// square-and-multiply over the 229 set bits took 378 + 228: 27 % fewer multiply-adds per exponentiation.  The table is 112 registers; its entry is
// picked by the (wave-uniform) index with selects, so there is ONE inlined multiplication site and one squaring site, as before.
template <int N>
__device__ __forceinline__ Fp fp_pow_sw4(const Fp& a, const uint16_t (&sched)[N]) {
    Fp t[8];
    const Fp a2 = fp28::fp_sqr_call(a);
    t[0] = a;
#pragma unroll
    for (int k = 1; k < 8; k++) t[k] = fp28::fp_mul_call(t[k - 1], a2);
    auto pick = [&](uint32_t idx) {
        Fp m = t[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; k++) m = fp28::fp_select(idx == k, m, t[k]);
        return m;
    };
    Fp acc = pick(sched[0] & 15u);
#pragma unroll 1
    for (int i = 1; i < N; i++) {
        const uint32_t v = sched[i];
#pragma unroll 1
        for (uint32_t sq = v >> 4; sq; sq--) acc = fp28::fp_sqr(acc);
        if ((v & 15u) != 15u) acc = fp28::fp_mul(acc, pick(v & 15u));
    }
    return acc;
}

// is_torsion_free of an affine point in the internal form: ec::g1_torsion_free — two 63-step Jacobian ladders (round 6: 4 S + 3 M + one
// small reduction per doubling where the homogeneous doubling of rounds 4-5 took 3 S + 5.5 M), doublings with the multiplier inlined,
// the ten additions through the shared body on a COPY.  BY VALUE and inlined since round 6: as an out-of-line function over references
// the running point lived in scratch memory and every doubling loaded and stored it — rocprofv3's FETCH_SIZE / WRITE_SIZE of the callers
// showed 26 KB of HBM traffic per point (k_validate<G1C>: 6.85 GB per 2^18-point launch, 185 x the 144 algorithmic bytes).
__device__ __forceinline__ bool g1_in_subgroup(const Fp& x, const Fp& y) { return ec::g1_torsion_free<ec::FpOpsInlinePS, ec::FpOps>(x, y); }
__device__ __forceinline__ bool g1_on_curve(const Fp& x, const Fp& y) {   // y^2 == x^3 + 4
    Fp rhs = fp28::fp_add(fp28::fp_mul_call(fp28::fp_sqr_call(x), x), fp28::fp_const(fp28c::FOUR));
    return fp_equal(fp28::fp_sqr_call(y), rhs);
}

__global__ void __launch_bounds__(256, 2) k_deserialize_g1(const uint8_t* __restrict__ bytes, uint32_t n, int compressed, int validate,
                                                        uint32_t* __restrict__ out_aff, uint8_t* __restrict__ status) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t size = compressed ? 48u : 96u;
    const uint8_t* b = bytes + (size_t)i * size;
    uint32_t* o = out_aff + (size_t)i * 24;
    uint8_t b0 = b[0];
    uint32_t c_flag = b0 >> 7, i_flag = (b0 >> 6) & 1, s_flag = (b0 >> 5) & 1;
    uint32_t xw[12], yw[12];
    be48_to_words(xw, b, 0x1fffffffu);
#pragma unroll
    for (int k = 0; k < 12; k++) yw[k] = 0;
    if (!compressed) be48_to_words(yw, b + 48, 0xffffffffu);
    uint8_t st = 0;
    bool is_inf = false;
    if (c_flag != (uint32_t)(compressed ? 1 : 0)) st = 1;
    if (st == 0 && i_flag) {
        uint32_t any = s_flag;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= xw[k] | yw[k];
        if (any) st = 1;
        is_inf = true;
    }
    if (st == 0 && !is_inf) {
        if (!words_lt_p(xw) || (!compressed && (!words_lt_p(yw) || s_flag))) st = 1;
    }
    Fp x = fp28::fp_zero(), y = fp28::fp_zero();
    if (st == 0 && !is_inf) {
        const Fp r2 = fp28::fp_const(fp28c::R2);
        x = fp28::fp_mul_call(fp28::fp_unpack384(xw), r2);                       // canonical integer -> internal
        Fp rhs = fp28::fp_add(fp28::fp_mul_call(fp28::fp_sqr_call(x), x), fp28::fp_const(fp28c::FOUR));   // x^3 + 4  < 4p
        bool on_curve;
        if (compressed) {
            y = fp_pow_sw4(rhs, fp28c::SQRT_SW4);   // rhs^((p+1)/4)
            on_curve = fp_equal(fp28::fp_sqr_call(y), rhs);
            if (!on_curve) st = 1;  // no square root: malformed compressed encoding
            // pick the root the sort flag asks for
            Fp yc = fp_to_canonical(y);
            bool larger = false, decided = false;
#pragma unroll
            for (int k = NL - 1; k >= 0; k--) {
                if (!decided && yc.l[k] != fp28c::HALF_P[k]) { larger = yc.l[k] > fp28c::HALF_P[k]; decided = true; }
            }
            if (larger != (s_flag != 0)) y = fp28::fp_neg<4>(y);
        } else {
            y = fp28::fp_mul_call(fp28::fp_unpack384(yw), r2);
            on_curve = !validate || fp_equal(fp28::fp_sqr_call(y), rhs);
            if (!on_curve) st = 2;
        }
        if (st == 0 && validate && !g1_in_subgroup(x, y)) st = 3;
    }
    uint32_t w[12];
    bool keep = st == 0 && !is_inf;
    fp28::fp_to_blst(w, x);
#pragma unroll
    for (int k = 0; k < 12; k++) o[k] = keep ? w[k] : 0u;
    fp28::fp_to_blst(w, y);
#pragma unroll
    for (int k = 0; k < 12; k++) o[12 + k] = keep ? w[k] : 0u;
    status[i] = st;
}

// affine (blst form) -> ZCash encoding
__global__ void __launch_bounds__(256, 2) k_serialize_g1(const uint32_t* __restrict__ aff, uint32_t n, int compressed, uint8_t* __restrict__ bytes) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = aff + (size_t)i * 24;
    const uint32_t size = compressed ? 48u : 96u;
    uint8_t* b = bytes + (size_t)i * size;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < 24; k++) any |= q[k];
    Fp x, y;
    fp_from_raw(x, q);
    fp_from_raw(y, q + 12);
    Fp xc = fp_to_canonical(x), yc = fp_to_canonical(y);
    uint32_t xw[12], yw[12];
    fp28::fp_pack384(xw, xc);
    fp28::fp_pack384(yw, yc);
    bool larger = false, decided = false;
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        if (!decided && yc.l[k] != fp28c::HALF_P[k]) { larger = yc.l[k] > fp28c::HALF_P[k]; decided = true; }
    }
    for (uint32_t k = 0; k < size; k++) b[k] = 0;
    if (any == 0) {
        b[0] = compressed ? 0xC0 : 0x40;
        return;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint8_t* d = b + 44 - 4 * k;
        d[0] = (uint8_t)(xw[k] >> 24); d[1] = (uint8_t)(xw[k] >> 16); d[2] = (uint8_t)(xw[k] >> 8); d[3] = (uint8_t)xw[k];
        if (!compressed) {
            uint8_t* e = b + 48 + 44 - 4 * k;
            e[0] = (uint8_t)(yw[k] >> 24); e[1] = (uint8_t)(yw[k] >> 16); e[2] = (uint8_t)(yw[k] >> 8); e[3] = (uint8_t)yw[k];
        }
    }
    if (compressed) b[0] |= (uint8_t)(0x80 | (larger ? 0x20 : 0));
}

// ---------------------------------------------------------------------------------------------- G2 point decoding
// Same for G2 (/root/reference/src/g2.rs:338-411): 96-byte compressed / 192-byte uncompressed, coordinates in Fp2
// serialised c1 first; y^2 = x^3 + 4(1 + u).  Square root in Fp2 by the complex method (fp2_sqrt_complex below; rounds 2-5: Adj,
// Rodriguez-Henriquez Alg. 9 with two exponentiations in Fp2).
// Subgroup test: psi(P) == [z] P (z < 0), psi(x, y) = (conj(x) PSI_X, conj(y) PSI_Y)  (M. Scott, eprint 2021/1130).
using G2F = ec::Fp2Ops;
constexpr int G2_RAW_AFF_WORDS = Geo<G2C>::RAW_AFF;   // 48
__device__ __forceinline__ bool fp2_equal(const ec::Fp2& a, const ec::Fp2& b) { return fp_equal(a.c0, b.c0) && fp_equal(a.c1, b.c1); }
__device__ __forceinline__ bool fp2_is_zero(const ec::Fp2& a) { return fp28::fp_is_zero_any(a.c0) && fp28::fp_is_zero_any(a.c1); }
__device__ __forceinline__ ec::Fp2 fp2_conj(const ec::Fp2& a) { return ec::Fp2{a.c0, fp28::fp_neg<16>(a.c1)}; }

__device__ __noinline__ ec::Fp2 fp2_pow(const ec::Fp2& a, const uint32_t (&e)[12], int top_bit) {
    ec::Fp2 acc = a;
#pragma unroll 1
    for (int bit = top_bit - 1; bit >= 0; bit--) {
        acc = G2F::sqr(acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) acc = G2F::mul(acc, a);
    }
    return acc;
}
// a^((p-3)/4) in Fp: for a square a, a * a^((p-3)/4) is a square root of a and a^((p-3)/4) itself is that root's INVERSE
// (their product is a^((p-1)/2) = 1); for a non-square the product of the two is -1 and (a * a^((p-3)/4))^2 = -a.
__device__ __noinline__ Fp fp_pow_p3_4(const Fp& a) { return fp_pow_sw4(a, fp28c::P3_4_SW4); }
// Square root in Fp2 = Fp[u] / (u^2 + 1) by the complex method: TWO exponentiations in Fp (2 x 570 field multiplications) where rounds 2-5
// ran two in Fp2 (Adj / Rodriguez-Henriquez: 2 x 1330) — VERDICT r05 #7.  For a = a0 + a1 u with norm n = a0^2 + a1^2:
//   s = sqrt(n) (a square whenever a is one);  t = (a0 + s) / 2;  r = t^((p-3)/4);  c = r t   (c^2 = t if t is a square, -t if not)
//   t a square:      y = c + (a1 r / 2) u          (r = 1 / c, so a1 r / 2 = a1 / (2 c))
//   t a non-square:  y = -(a1 r / 2) + c u         ((a0 - s) / 2 = -a1^2 / (4 t) is the square then; -r = 1 / c)
//   a1 = 0:          t = a0;  y = c  or  c u       (a0 a square or not)
// Either root serves: the caller picks the sign the encoding's sort flag asks for and verifies y^2 = a (which also rejects a non-square a).
__device__ __forceinline__ ec::Fp2 fp2_sqrt_complex(const ec::Fp2& a) {
    const Fp inv2 = fp28::fp_const(fp28c::INV2);
    const bool a1z = fp28::fp_is_zero_any(a.c1);
    const Fp n = fp28::fp_add(fp28::fp_sqr_call(a.c0), fp28::fp_sqr_call(a.c1));
    const Fp s = fp28::fp_mul_call(fp_pow_p3_4(n), n);
    const Fp t = fp28::fp_select(a1z, fp28::fp_mul_call(fp28::fp_add(a.c0, s), inv2), a.c0);
    const Fp r = fp_pow_p3_4(t);
    const Fp c = fp28::fp_mul_call(r, t);
    const bool qr = fp_equal(fp28::fp_sqr_call(c), t);
    const Fp h = fp28::fp_select(a1z, fp28::fp_mul_call(fp28::fp_mul_call(a.c1, r), inv2), fp28::fp_zero());   // a1 r / 2 (0 when a1 = 0)
    ec::Fp2 y;
    y.c0 = fp28::fp_select(qr, fp28::fp_neg<4>(h), c);
    y.c1 = fp28::fp_select(qr, c, h);
    return y;
}

__device__ __noinline__ void g2_mul_z(ec::Proj<G2F>& r, const ec::Proj<G2F>& p) {
    r = p;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        ec::proj_dbl<G2F>(r);   // 3 S + 4 M + one fused pair over Fp2 (22 field multiplications) instead of the 36 of a complete addition
        if ((fp28c::Z_ABS >> bit) & 1) ec::proj_add<G2F>(r, p);
    }
}
__device__ __forceinline__ bool canon_gt_half(const Fp& c) {
    bool larger = false, decided = false;
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        if (!decided && c.l[k] != fp28c::HALF_P[k]) { larger = c.l[k] > fp28c::HALF_P[k]; decided = true; }
    }
    return larger;
}
__device__ __forceinline__ bool fp2_lex_largest(const ec::Fp2& y) {  // c1 first, then c0
    Fp c1 = fp_to_canonical(y.c1);
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) z |= c1.l[k];
    if (z != 0) return canon_gt_half(c1);
    return canon_gt_half(fp_to_canonical(y.c0));
}

// is_torsion_free of an affine point in the internal form: psi(P) == [z] P = -[|z|] P, i.e. X_q == px Z_q, Y_q == -py Z_q, Z_q != 0
__device__ __noinline__ bool g2_in_subgroup(const ec::Fp2& x, const ec::Fp2& y) {
    ec::Proj<G2F> p1 = ec::proj_from_affine<G2F>(x, y), q;
    g2_mul_z(q, p1);                                                          // [|z|] P
    ec::Fp2 px = G2F::mul(fp2_conj(x), ec::Fp2{fp28::fp_zero(), fp28::fp_const(fp28c::PSI_X1)});
    ec::Fp2 py = G2F::mul(fp2_conj(y), ec::Fp2{fp28::fp_const(fp28c::PSI_Y0), fp28::fp_const(fp28c::PSI_Y1)});
    bool ok = !fp2_is_zero(q.z);
    ok = ok && fp2_equal(q.x, G2F::mul(px, q.z));
    ok = ok && fp2_is_zero(G2F::add(q.y, G2F::mul(py, q.z)));
    return ok;
}
__device__ __forceinline__ bool g2_on_curve(const ec::Fp2& x, const ec::Fp2& y) {   // y^2 == x^3 + 4 (1 + u)
    const Fp four = fp28::fp_const(fp28c::FOUR);
    ec::Fp2 rhs = G2F::add(G2F::mul(G2F::sqr(x), x), ec::Fp2{four, four});
    return fp2_equal(G2F::sqr(y), rhs);
}

// The decoder proper: encoding checks, decompression (square root in Fp2) and, for uncompressed input with `validate`, the curve equation.  The
// subgroup test of Valid::check is the SECOND kernel of mi_g2_deserialize_batch (k_validate<G2C, true> below): fused, the straight-line
// code between the two out-of-line loops (fp2_pow, g2_mul_z) kept 86 registers in scratch (round 4's finding); apart, neither kernel spills.
__global__ void __launch_bounds__(256, 2) k_deserialize_g2(const uint8_t* __restrict__ bytes, uint32_t n, int compressed, int validate,
                                                        uint32_t* __restrict__ out_aff, uint8_t* __restrict__ status) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t size = compressed ? 96u : 192u;
    const uint8_t* b = bytes + (size_t)i * size;
    uint32_t* o = out_aff + (size_t)i * 48;
    uint8_t b0 = b[0];
    uint32_t c_flag = b0 >> 7, i_flag = (b0 >> 6) & 1, s_flag = (b0 >> 5) & 1;
    uint32_t x1w[12], x0w[12], y1w[12], y0w[12];
    be48_to_words(x1w, b, 0x1fffffffu);
    be48_to_words(x0w, b + 48, 0xffffffffu);
#pragma unroll
    for (int k = 0; k < 12; k++) { y1w[k] = 0; y0w[k] = 0; }
    if (!compressed) {
        be48_to_words(y1w, b + 96, 0xffffffffu);
        be48_to_words(y0w, b + 144, 0xffffffffu);
    }
    uint8_t st = 0;
    bool is_inf = false;
    if (c_flag != (uint32_t)(compressed ? 1 : 0)) st = 1;
    if (st == 0 && i_flag) {
        uint32_t any = s_flag;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= x1w[k] | x0w[k] | y1w[k] | y0w[k];
        if (any) st = 1;
        is_inf = true;
    }
    if (st == 0 && !is_inf) {
        bool ok = words_lt_p(x1w) && words_lt_p(x0w);
        if (!compressed) ok = ok && words_lt_p(y1w) && words_lt_p(y0w) && !s_flag;
        if (!ok) st = 1;
    }
    ec::Fp2 x = G2F::zero(), y = G2F::zero();
    if (st == 0 && !is_inf) {
        const Fp r2 = fp28::fp_const(fp28c::R2), four = fp28::fp_const(fp28c::FOUR);
        x.c0 = fp28::fp_mul_call(fp28::fp_unpack384(x0w), r2);
        x.c1 = fp28::fp_mul_call(fp28::fp_unpack384(x1w), r2);
        ec::Fp2 rhs = G2F::add(G2F::mul(G2F::sqr(x), x), ec::Fp2{four, four});     // x^3 + 4(1 + u)   < 4p
        if (compressed) {
            y = fp2_sqrt_complex(rhs);
            if (!fp2_equal(G2F::sqr(y), rhs)) st = 1;                                // not a square: malformed
            if (fp2_lex_largest(y) != (s_flag != 0)) y = G2F::neg<4>(y);
        } else {
            y.c0 = fp28::fp_mul_call(fp28::fp_unpack384(y0w), r2);
            y.c1 = fp28::fp_mul_call(fp28::fp_unpack384(y1w), r2);
            if (validate && !fp2_equal(G2F::sqr(y), rhs)) st = 2;
        }
    }
    bool keep = st == 0 && !is_inf;
    ElemIO<ec::Fp2>::to_raw(o, x, keep);
    ElemIO<ec::Fp2>::to_raw(o + 24, y, keep);
    status[i] = st;
}

template <class C> struct PointCheck;
template <> struct PointCheck<G1C> {
    static __device__ __forceinline__ bool on_curve(const Fp& x, const Fp& y) { return g1_on_curve(x, y); }
    static __device__ __forceinline__ bool in_subgroup(const Fp& x, const Fp& y) { return g1_in_subgroup(x, y); }
};
template <> struct PointCheck<G2C> {
    static __device__ __forceinline__ bool on_curve(const ec::Fp2& x, const ec::Fp2& y) { return g2_on_curve(x, y); }
    static __device__ __forceinline__ bool in_subgroup(const ec::Fp2& x, const ec::Fp2& y) { return g2_in_subgroup(x, y); }
};

// Valid::check (is_on_curve && is_torsion_free, /root/reference/src/g1.rs:386-396, src/g2.rs:366-376) of n affine points in device memory.
//   MODE 0  mi_msm_g{1,2}_validate_bases: the RESIDENT base set in the device form (infinity flag in the point's last word);
//           both halves of the check; *n_bad counts the points that fail.
//   MODE 1  second pass of mi_g2_deserialize_batch: points in the reference's form as the decoder wrote them (all-zero = infinity or
//           rejected: skipped), already known to be on the curve; a point outside the subgroup gets status 3 and is zeroed.
//   MODE 2  mi_g{1,2}_check_batch (Valid::batch_check over affine points): points in the reference's form, both halves of the check,
//           status[i] = 0 / 2 (not on the curve) / 3 (not in the subgroup); the points are not touched.
template <class C, int MODE>
__global__ void __launch_bounds__(256, 2) k_validate(uint32_t* __restrict__ pts, uint32_t n, uint8_t* __restrict__ status, uint32_t* __restrict__ n_bad) {
    using E = typename C::F::E;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    E x, y;
    if constexpr (MODE != 0) {
        if (MODE == 1 && status[i] != 0) return;
        uint32_t* q = pts + (size_t)i * Geo<C>::RAW_AFF;
        uint32_t any = 0;
#pragma unroll 4
        for (int k = 0; k < Geo<C>::RAW_AFF; k++) any |= q[k];
        if (any == 0) {   // infinity: a member of every subgroup
            if (MODE == 2) status[i] = 0;
            return;
        }
        ElemIO<E>::from_raw(x, q);
        ElemIO<E>::from_raw(y, q + ElemIO<E>::RAW);
        if constexpr (MODE == 1) {
            if (!PointCheck<C>::in_subgroup(x, y)) {
                status[i] = 3;
#pragma unroll 4
                for (int k = 0; k < Geo<C>::RAW_AFF; k++) q[k] = 0u;
            }
        } else {
            uint8_t st = 0;
            if (!PointCheck<C>::on_curve(x, y)) st = 2;
            else if (!PointCheck<C>::in_subgroup(x, y)) st = 3;
            status[i] = st;
        }
    } else {
        const uint32_t* q = pts + (size_t)i * Geo<C>::PT_WORDS;
        if (q[Geo<C>::PT_WORDS - 1] != 0) return;   // infinity: a member of every subgroup
        ElemIO<E>::load(x, q);
        ElemIO<E>::load(y, q + Geo<C>::SLOT);
        uint8_t st = 0;
        if (!PointCheck<C>::on_curve(x, y)) st = 2;
        else if (!PointCheck<C>::in_subgroup(x, y)) st = 3;
        if (status) status[i] = st;
        if (st) atomicAdd(n_bad, 1u);
    }
}

__device__ __forceinline__ void words_to_be48(uint8_t* d, const uint32_t (&w)[12]) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        uint8_t* q = d + 44 - 4 * k;
        q[0] = (uint8_t)(w[k] >> 24); q[1] = (uint8_t)(w[k] >> 16); q[2] = (uint8_t)(w[k] >> 8); q[3] = (uint8_t)w[k];
    }
}
__global__ void __launch_bounds__(256, 2) k_serialize_g2(const uint32_t* __restrict__ aff, uint32_t n, int compressed, uint8_t* __restrict__ bytes) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = aff + (size_t)i * 48;
    const uint32_t size = compressed ? 96u : 192u;
    uint8_t* b = bytes + (size_t)i * size;
    uint32_t any = 0;
#pragma unroll 4
    for (int k = 0; k < 48; k++) any |= q[k];
    for (uint32_t k = 0; k < size; k++) b[k] = 0;
    if (any == 0) {
        b[0] = compressed ? 0xC0 : 0x40;
        return;
    }
    ec::Fp2 x, y;
    ElemIO<ec::Fp2>::from_raw(x, q);
    ElemIO<ec::Fp2>::from_raw(y, q + 24);
    uint32_t w[12];
    fp28::fp_pack384(w, fp_to_canonical(x.c1)); words_to_be48(b, w);
    fp28::fp_pack384(w, fp_to_canonical(x.c0)); words_to_be48(b + 48, w);
    if (!compressed) {
        fp28::fp_pack384(w, fp_to_canonical(y.c1)); words_to_be48(b + 96, w);
        fp28::fp_pack384(w, fp_to_canonical(y.c0)); words_to_be48(b + 144, w);
    } else {
        b[0] |= (uint8_t)(0x80 | (fp2_lex_largest(y) ? 0x20 : 0));
    }
}

// Valid::check for G2 on LANE PAIRS (round 6): the even lane holds c0 and the odd lane c1 of every Fp2 value (CoopF2, coop_fp2.cuh — the
// scheme of the G2 accumulate kernel), two lanes per point.  k_validate<G2C> above keeps a whole Fp2 per lane and runs its 64-bit ladder
// through the shared out-of-line multiplier: the point lives in scratch, 180 scratch loads / stores per doubling — rocprofv3 counted
// 61 GB of HBM traffic per 2^18-point launch (4 TB/s: the kernel was bound by its own spills, profiles/r06_rows_f_g2_2p18_pmc_summary.json).
// Here a projective point is 42 registers per lane, the multiplier is inlined, nothing is called and nothing spills; a lane-pair product is
// one fused two-product reduction per lane (600 instructions against 3 x 616 for the Karatsuba product of one lane).  Same MODEs, same
// outputs as k_validate<G2C, MODE>; grid = ceil(2 n / 256) workgroups of 256 lanes.
__device__ __forceinline__ bool pair_all(bool v) {
    const int m = v ? 1 : 0;
    return (m & __builtin_amdgcn_mov_dpp(m, 0xB1, 0xF, 0xF, true)) != 0;
}
__device__ __forceinline__ uint32_t pair_or(uint32_t v) { return v | (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_validate_g2_coop(uint32_t* __restrict__ pts, uint32_t n, uint8_t* __restrict__ status, uint32_t* __restrict__ n_bad) {
    using F = CoopF2;
    const uint32_t i = (blockIdx.x * 256 + threadIdx.x) >> 1, h = threadIdx.x & 1u;
    if (i >= n) return;                                   // both lanes of a pair share i
    Fp x, y;                                               // this lane's component of x and y
    if constexpr (MODE != 0) {
        if (MODE == 1 && status[i] != 0) return;
        uint32_t* q = pts + (size_t)i * G2_RAW_AFF_WORDS;   // x.c0 | x.c1 | y.c0 | y.c1, 12 words each
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= q[12 * h + k] | q[24 + 12 * h + k];
        if (pair_or(any) == 0) {                           // infinity: a member of every subgroup
            if (MODE == 2 && h == 0) status[i] = 0;
            return;
        }
        fp_from_raw(x, q + 12 * h);
        fp_from_raw(y, q + 24 + 12 * h);
    } else {
        const uint32_t* q = pts + (size_t)i * G2_PT_WORDS;   // device form: x.c0 | x.c1 | y.c0 | y.c1 in 16-word slots, word 63 = infinity flag
        if (q[G2_PT_WORDS - 1] != 0) return;
        load_fp16(x, q + 16 * h);
        load_fp16(y, q + 32 + 16 * h);
    }
    uint8_t st = 0;
    if constexpr (MODE != 1) {                             // y^2 == x^3 + 4 (1 + u)
        const Fp four = fp28::fp_const(fp28c::FOUR);
        const Fp rhs = F::add(F::mul(F::sqr(x), x), four);
        if (!pair_all(fp_equal(F::sqr(y), rhs))) st = 2;
    }
    if (st == 0) {                                         // psi(P) == [z] P = -[|z|] P:  X_q == px Z_q^2, Y_q == -py Z_q^3, Z_q != 0
        // Jacobian ladder (ec.cuh jac_dbl / jac_add over the lane-pair field, round 6): 4 S + 3 M per doubling where the homogeneous doubling
        // took 4 S + 4 M; its exceptional cases leave Z == 0, i.e. "not in the subgroup", and only points outside G2 reach them
        ec::JacFp p1;
        p1.x = x; p1.y = y; p1.z = F::one();
        const ec::JacFp q = ec::jac_mul_z<F, F, true, true>(p1);   // five of the 63 steps add; wave-uniform
        const Fp cx = F::select(h != 0, x, fp28::fp_neg<16>(x)), cy = F::select(h != 0, y, fp28::fp_neg<16>(y));   // conj: the odd lane's component negated
        const Fp px = F::mul(cx, F::select(h != 0, fp28::fp_zero(), fp28::fp_const(fp28c::PSI_X1)));
        const Fp py = F::mul(cy, F::select(h != 0, fp28::fp_const(fp28c::PSI_Y0), fp28::fp_const(fp28c::PSI_Y1)));
        const Fp zz = F::sqr(q.z);
        bool ok = !pair_all(fp28::fp_is_zero_any(q.z));
        ok = ok && pair_all(fp_equal(q.x, F::mul(px, zz)));
        ok = ok && pair_all(fp28::fp_is_zero_any(F::add(q.y, F::mul(py, F::mul(zz, q.z)))));
        if (!ok) st = 3;
    }
    if constexpr (MODE == 1) {
        if (st) {
            if (h == 0) status[i] = st;
            uint32_t* q = pts + (size_t)i * G2_RAW_AFF_WORDS;
#pragma unroll
            for (int k = 0; k < 12; k++) { q[12 * h + k] = 0u; q[24 + 12 * h + k] = 0u; }
        }
    } else if constexpr (MODE == 2) {
        if (h == 0) status[i] = st;
    } else {
        if (h == 0) {
            if (status) status[i] = st;
            if (st) atomicAdd(n_bad, 1u);
        }
    }
}

// rejected points of a decode pass (status != 0), for mi_msm_g{1,2}_set_bases_from_compressed: one counter per call
__global__ void __launch_bounds__(256) k_count_rejected(const uint8_t* __restrict__ status, uint32_t n, uint32_t* __restrict__ counter) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool bad = i < n && status[i] != 0;
    const uint64_t m = __ballot(bad);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(counter, (uint32_t)__popcll(m));
}

// ---------------------------------------------------------------------------------------------- field test hook
#if defined(MI_TEST_HOOKS)
__global__ void __launch_bounds__(256, 2) k_test_fp_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                    uint32_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t aw[12], bw[12], ow[12];
#pragma unroll
    for (int k = 0; k < 12; k++) { aw[k] = a[(size_t)i * 12 + k]; bw[k] = b[(size_t)i * 12 + k]; }
    Fp x = fp28::fp_from_blst(aw), y = fp28::fp_from_blst(bw), z;
    if (op == 0) z = fp28::fp_mul_call(x, y);
    else if (op == 1) z = fp28::fp_sqr(x);
    else if (op == 2) z = fp28::fp_add(x, y);
    else z = fp28::fp_sub<4>(x, y);
    
    fp28::fp_to_blst(ow, z);
#pragma unroll
    for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = ow[k];
}

#endif

}  // namespace msmk
