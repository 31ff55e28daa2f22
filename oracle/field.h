/* CPU ORACLE (test infrastructure only) — NOT PRODUCT CODE.
 *
 * Plain-C restatement of the base-field / scalar-field arithmetic under the reference's MSM path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this;
 * the shipped library (ark-blst_amd/csrc) has its own, independent arithmetic.
 *
 * Reference anchors (paths relative to /root/reference):
 *   - blst_fp  = 6 x u64 little-endian limbs, Montgomery R = 2^384, fully reduced; modulus src/fp.rs:25-32.
 *   - blst_fr  = 4 x u64, Montgomery R = 2^256; modulus src/scalar.rs:476-481.
 *   - Fp2 = Fp[u]/(u^2+1), layout (c0, c1) 96 B, src/fp2.rs:228,246-261.
 * The arithmetic itself is blst =0.3.10 (Cargo.toml:22), a dependency that is ABSENT from /root/reference;
 * this file restates the textbook algorithms (CIOS Montgomery multiplication, Fermat inversion).
 * Pinned by the reference's known-answer constants in oracle_selfcheck() (src/fp.rs:714-721 etc.).
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { uint64_t l[4]; } fr;

static const fp FP_P = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                         0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const fp FP_ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                           0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}}; /* R mod p */
static const fp FP_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                          0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};  /* R^2 mod p */
#define FP_PINV 0x89f3fffcfffcfffdULL /* -p^-1 mod 2^64 */

static const fr FR_R = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const fr FR_ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const fr FR_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
#define FR_RINV 0xfffffffeffffffffULL /* -r^-1 mod 2^64 */

/* ---------------------------------------------------------------- Fp */
static inline int fp_is_zero(const fp *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0;
}
static inline int fp_eq(const fp *a, const fp *b) { return memcmp(a, b, sizeof(fp)) == 0; }
static inline int fp_geq_p(const uint64_t *t) {
    for (int i = 5; i >= 0; i--) {
        if (t[i] > FP_P.l[i]) return 1;
        if (t[i] < FP_P.l[i]) return 0;
    }
    return 1;
}
static inline void fp_sub_p(uint64_t *t) {
    u128 b = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)t[i] - FP_P.l[i] - (uint64_t)b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
/* branch-free: the sum (difference) and its correction by p are both computed, one is selected by the final borrow */
static inline void fp_add(fp *r, const fp *a, const fp *b) {
    uint64_t t[6], d[6];
    u128 c = 0;
    for (int i = 0; i < 6; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    uint64_t bw = 0;   /* a+b < 2p < 2^382: no carry out of limb 5 */
    for (int i = 0; i < 6; i++) {
        u128 x = (u128)t[i] - FP_P.l[i] - bw;
        d[i] = (uint64_t)x;
        bw = (uint64_t)(x >> 64) & 1;
    }
    const uint64_t keep = (uint64_t)0 - bw;
    for (int i = 0; i < 6; i++) r->l[i] = (t[i] & keep) | (d[i] & ~keep);
}
static inline void fp_sub(fp *r, const fp *a, const fp *b) {
    uint64_t t[6];
    uint64_t bw = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - bw;
        t[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
    }
    const uint64_t fix = (uint64_t)0 - bw;
    u128 c = 0;
    for (int i = 0; i < 6; i++) { c += (u128)t[i] + (FP_P.l[i] & fix); r->l[i] = (uint64_t)c; c >>= 64; }
}
static inline void fp_neg(fp *r, const fp *a) {
    fp z; memset(&z, 0, sizeof z);
    fp_sub(r, &z, a);
}
/* CIOS Montgomery multiplication: r = a*b/2^384 mod p.
 * With clang and BMI2 + ADX (oracle/Makefile picks ROCm's clang and -mbmi2 -madx when both are there; the GPU box's EPYC has
 * them) every pass is six mulx feeding two independent carry chains, which is how blst's assembly is organised as well:
 * ~80 cycles per multiplication instead of ~125 for the portable form — the CPU baseline should not be a strawman. */
#if defined(__clang__) && defined(__ADX__) && defined(__BMI2__)
#include <immintrin.h>
#define ORACLE_FP_MUL_KIND "mulx/adcx/adox (clang intrinsics)"
static inline void fp_mul(fp *r, const fp *a, const fp *b) {
    typedef unsigned long long u64;
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
    const u64 *x = (const u64 *)a->l, *p = (const u64 *)FP_P.l;
#pragma clang loop unroll(full)
    for (int i = 0; i < 6; i++) {
        const u64 bi = b->l[i];
        u64 lo, h0, h1, h2, h3, h4, h5, drop;
        unsigned char c1 = 0, c2 = 0;
        lo = _mulx_u64(x[0], bi, &h0); c1 = _addcarryx_u64(c1, t0, lo, &t0);
        lo = _mulx_u64(x[1], bi, &h1); c1 = _addcarryx_u64(c1, t1, lo, &t1); c2 = _addcarryx_u64(c2, t1, h0, &t1);
        lo = _mulx_u64(x[2], bi, &h2); c1 = _addcarryx_u64(c1, t2, lo, &t2); c2 = _addcarryx_u64(c2, t2, h1, &t2);
        lo = _mulx_u64(x[3], bi, &h3); c1 = _addcarryx_u64(c1, t3, lo, &t3); c2 = _addcarryx_u64(c2, t3, h2, &t3);
        lo = _mulx_u64(x[4], bi, &h4); c1 = _addcarryx_u64(c1, t4, lo, &t4); c2 = _addcarryx_u64(c2, t4, h3, &t4);
        lo = _mulx_u64(x[5], bi, &h5); c1 = _addcarryx_u64(c1, t5, lo, &t5); c2 = _addcarryx_u64(c2, t5, h4, &t5);
        c1 = _addcarryx_u64(c1, t6, 0, &t6); c2 = _addcarryx_u64(c2, t6, h5, &t6);
        const u64 m = t0 * FP_PINV;
        c1 = 0; c2 = 0;
        lo = _mulx_u64(m, p[0], &h0); c1 = _addcarryx_u64(c1, t0, lo, &drop);
        lo = _mulx_u64(m, p[1], &h1); c1 = _addcarryx_u64(c1, t1, lo, &t0); c2 = _addcarryx_u64(c2, t0, h0, &t0);
        lo = _mulx_u64(m, p[2], &h2); c1 = _addcarryx_u64(c1, t2, lo, &t1); c2 = _addcarryx_u64(c2, t1, h1, &t1);
        lo = _mulx_u64(m, p[3], &h3); c1 = _addcarryx_u64(c1, t3, lo, &t2); c2 = _addcarryx_u64(c2, t2, h2, &t2);
        lo = _mulx_u64(m, p[4], &h4); c1 = _addcarryx_u64(c1, t4, lo, &t3); c2 = _addcarryx_u64(c2, t3, h3, &t3);
        lo = _mulx_u64(m, p[5], &h5); c1 = _addcarryx_u64(c1, t5, lo, &t4); c2 = _addcarryx_u64(c2, t4, h4, &t4);
        c1 = _addcarryx_u64(c1, t6, 0, &t5); c2 = _addcarryx_u64(c2, t5, h5, &t5);
        t6 = (u64)c1 + (u64)c2;
    }
    uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
    if (t6 || fp_geq_p(t)) fp_sub_p(t);
    memcpy(r->l, t, sizeof t);
}
#else
#define ORACLE_FP_MUL_KIND "unsigned __int128 CIOS (portable C)"
static inline void fp_mul(fp *r, const fp *a, const fp *b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FP_PINV;
        c = (u128)m * FP_P.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 6; j++) { c += (u128)m * FP_P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
    }
    if (t[6] || fp_geq_p(t)) fp_sub_p(t);
    memcpy(r->l, t, 6 * sizeof(uint64_t));
}
#endif
static inline void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }
static inline void fp_from_mont(fp *r, const fp *a) { /* -> canonical integer limbs */
    fp one; memset(&one, 0, sizeof one); one.l[0] = 1;
    fp_mul(r, a, &one);
}
static inline void fp_to_mont(fp *r, const fp *a) { fp_mul(r, a, &FP_R2); }
/* r = a^(p-2) (Fermat); a = 0 -> 0 */
static inline void fp_inv(fp *r, const fp *a) {
    uint64_t e[6];
    memcpy(e, FP_P.l, sizeof e);
    e[0] -= 2; /* p-2, no borrow: low limb ends in ...aaab */
    fp acc = FP_ONE, base = *a;
    for (int i = 0; i < 384; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
        fp_sqr(&base, &base);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1) */
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
    fp t0, t1, s0, s1, m;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1);
    fp_add(&s1, &b->c0, &b->c1);
    fp_mul(&m, &s0, &s1);
    fp_sub(&m, &m, &t0);
    fp_sub(&r->c1, &m, &t1);
    fp_sub(&r->c0, &t0, &t1);
}
static inline void fp2_sqr(fp2 *r, const fp2 *a) { fp2_mul(r, a, a); }
static inline void fp2_inv(fp2 *r, const fp2 *a) {
    fp n, t;
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n);
    fp_mul(&t, &a->c1, &n);
    fp_neg(&r->c1, &t);
}

/* ---------------------------------------------------------------- Fr (4 x u64, Montgomery R = 2^256) */
static inline int fr_geq_r(const uint64_t *t) {
    for (int i = 3; i >= 0; i--) {
        if (t[i] > FR_R.l[i]) return 1;
        if (t[i] < FR_R.l[i]) return 0;
    }
    return 1;
}
static inline void fr_sub_r(uint64_t *t) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - FR_R.l[i] - (uint64_t)b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline void fr_mul(fr *r, const fr *a, const fr *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_RINV;
        c = (u128)m * FR_R.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * FR_R.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || fr_geq_r(t)) fr_sub_r(t);
    memcpy(r->l, t, 4 * sizeof(uint64_t));
}
static inline void fr_add(fr *r, const fr *a, const fr *b) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (fr_geq_r(t)) fr_sub_r(t); /* r < 2^255: a+b < 2^256 */
    memcpy(r->l, t, sizeof t);
}
/* Scalar::into_bigint (src/scalar.rs:450-463,503-505): Montgomery -> canonical */
static inline void fr_from_mont(fr *r, const fr *a) {
    fr one = {{1, 0, 0, 0}};
    fr_mul(r, a, &one);
}
static inline void fr_to_mont(fr *r, const fr *a) { fr_mul(r, a, &FR_R2); }

#endif
