/* CPU ORACLE (test infrastructure only) — plain C restatement of the textbook optimal-ate pairing of
 * oracle/pairing.py, so that pairing parity can be checked on thousands of pairs and a CPU figure can be timed.
 *
 * NOT PRODUCT CODE: only tests/, smoke() and the bench tools' cpu_baseline leg may load it (through oracle/coracle.py).
 * Restates what /root/reference/src/pairing.rs:49-80 computes (it forwards to blstrs / blst, absent from
 * /root/reference).  Same algorithm as the Python oracle and deliberately unlike the shipped code: flat
 * Fp12 = Fp2[w]/(w^6 - xi) with schoolbook products, AFFINE Miller loop on the twist (one Fp2 inversion per step) with
 * the exact untwisted line  l(P) = yP - lambda xP w^-1 + (lambda xT - yT) w^-3,  final exponentiation as ONE
 * square-and-multiply by the integer 3 (p^12 - 1) / r handed over by the caller (blst's convention, see pairing.py).
 * PINNING: parity unpinned (the reference holds no Gt known answer); checked against oracle/pairing.py and its golden
 * vectors in tests/test_pairing_cpu.py.
 * As a timed CPU baseline it is a "port": portable C on 64-bit limbs, ~4x slower per Miller loop than blst's assembly.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "field.h"

typedef struct { fp2 a[6]; } fp12;   /* a[0] + a[1] w + ... + a[5] w^5 */

static void fp2_mul_xi(fp2 *r, const fp2 *a) {   /* (a0 + a1 u)(1 + u) */
    fp t0, t1;
    fp_sub(&t0, &a->c0, &a->c1);
    fp_add(&t1, &a->c0, &a->c1);
    r->c0 = t0; r->c1 = t1;
}
static void fp12_one(fp12 *r) { memset(r, 0, sizeof *r); r->a[0].c0 = FP_ONE; }
static void fp12_mul(fp12 *r, const fp12 *x, const fp12 *y) {
    fp2 t[11];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 6; i++) {
        if (fp2_is_zero(&x->a[i])) continue;
        for (int j = 0; j < 6; j++) {
            fp2 m;
            fp2_mul(&m, &x->a[i], &y->a[j]);
            fp2_add(&t[i + j], &t[i + j], &m);
        }
    }
    for (int k = 0; k < 5; k++) {
        fp2 m;
        fp2_mul_xi(&m, &t[k + 6]);
        fp2_add(&r->a[k], &t[k], &m);
    }
    r->a[5] = t[5];
}

static uint64_t Z_ABS = 0xd201000000010000ULL;

typedef struct { fp x, y; } g1a;
typedef struct { fp2 x, y; } g2a;

static void line(fp12 *l, const fp2 *lam, const fp2 *xt, const fp2 *yt, const g1a *p, const fp2 *xi_inv) {
    memset(l, 0, sizeof *l);
    l->a[0].c0 = p->y;
    fp2 t, xp;
    memset(&xp, 0, sizeof xp);
    xp.c0 = p->x;
    fp2_mul(&t, lam, &xp);
    fp2_neg(&t, &t);
    fp2_mul(&l->a[5], &t, xi_inv);               /* -lambda xP w^-1 = (-lambda xP / xi) w^5 */
    fp2_mul(&t, lam, xt);
    fp2_sub(&t, &t, yt);
    fp2_mul(&l->a[3], &t, xi_inv);               /* (lambda xT - yT) w^-3 = (.. / xi) w^3 */
}

static void miller_loop(fp12 *f, const g1a *p, const g2a *q) {
    fp2 xi, xi_inv;
    xi.c0 = FP_ONE; xi.c1 = FP_ONE;
    fp2_inv(&xi_inv, &xi);
    fp12_one(f);
    fp2 xt = q->x, yt = q->y;
    for (int b = 62; b >= 0; b--) {
        fp2 lam, n, d, x3, t;
        fp12 l, s;
        fp2_sqr(&n, &xt);                         /* 3 x^2 / (2 y) */
        fp2_add(&t, &n, &n); fp2_add(&n, &t, &n);
        fp2_add(&d, &yt, &yt);
        fp2_inv(&d, &d);
        fp2_mul(&lam, &n, &d);
        line(&l, &lam, &xt, &yt, p, &xi_inv);
        fp12_mul(&s, f, f);
        fp12_mul(f, &s, &l);
        fp2_sqr(&x3, &lam); fp2_sub(&x3, &x3, &xt); fp2_sub(&x3, &x3, &xt);
        fp2_sub(&t, &xt, &x3); fp2_mul(&t, &lam, &t); fp2_sub(&yt, &t, &yt);
        xt = x3;
        if ((Z_ABS >> b) & 1) {
            fp2_sub(&n, &yt, &q->y);
            fp2_sub(&d, &xt, &q->x);
            fp2_inv(&d, &d);
            fp2_mul(&lam, &n, &d);
            line(&l, &lam, &xt, &yt, p, &xi_inv);
            fp12_mul(&s, f, &l);
            *f = s;
            fp2_sqr(&x3, &lam); fp2_sub(&x3, &x3, &xt); fp2_sub(&x3, &x3, &q->x);
            fp2_sub(&t, &xt, &x3); fp2_mul(&t, &lam, &t); fp2_sub(&yt, &t, &yt);
            xt = x3;
        }
    }
    for (int k = 1; k < 6; k += 2) fp2_neg(&f->a[k], &f->a[k]);   /* z < 0: conjugate (w -> -w) */
}

/* blst_fp12 order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) <-> flat powers (0, 2, 4, 1, 3, 5) */
static const int FLAT_OF_SLOT[6] = {0, 2, 4, 1, 3, 5};
static void fp12_to_bytes(uint8_t *out, const fp12 *f) {
    for (int s = 0; s < 6; s++) memcpy(out + 96 * s, &f->a[FLAT_OF_SLOT[s]], 96);
}
static void fp12_from_bytes(fp12 *f, const uint8_t *in) {
    for (int s = 0; s < 6; s++) memcpy(&f->a[FLAT_OF_SLOT[s]], in + 96 * s, 96);
}
static int all_zero(const uint8_t *p, size_t n) {
    uint8_t v = 0;
    for (size_t i = 0; i < n; i++) v |= p[i];
    return v == 0;
}

typedef struct { const uint8_t *g1, *g2; size_t n, per; fp12 *part; } ml_ctx;
static void *ml_worker(void *arg_) {
    void **arg = (void **)arg_;
    ml_ctx *c = (ml_ctx *)arg[0];
    size_t t = (size_t)arg[1];
    size_t lo = t * c->per, hi = lo + c->per < c->n ? lo + c->per : c->n;
    fp12 acc;
    fp12_one(&acc);
    for (size_t i = lo; i < hi; i++) {
        const uint8_t *p = c->g1 + 96 * i, *q = c->g2 + 192 * i;
        if (all_zero(p, 96) || all_zero(q, 192)) continue;        /* pairing with infinity is one (pairing.rs:58-60) */
        g1a P; g2a Q; fp12 f, m;
        memcpy(&P, p, 96); memcpy(&Q, q, 192);
        miller_loop(&f, &P, &Q);
        fp12_mul(&m, &acc, &f);
        acc = m;
    }
    c->part[t] = acc;
    return NULL;
}

/* prod_i f_{z,Q_i}(P_i) over packed blst_p1_affine / blst_p2_affine arrays -> blst_fp12 bytes */
void orc_multi_miller_loop(const uint8_t *g1, const uint8_t *g2, size_t n, int nthreads, uint8_t *out) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    ml_ctx c = {g1, g2, n, (n + nthreads - 1) / (nthreads ? nthreads : 1), NULL};
    c.part = (fp12 *)malloc(sizeof(fp12) * nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    void **args = (void **)malloc(sizeof(void *) * 2 * nthreads);
    for (int t = 0; t < nthreads; t++) {
        args[2 * t] = &c; args[2 * t + 1] = (void *)(size_t)t;
        if (nthreads == 1) ml_worker(&args[0]);
        else pthread_create(&th[t], NULL, ml_worker, &args[2 * t]);
    }
    fp12 acc, m;
    fp12_one(&acc);
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        fp12_mul(&m, &acc, &c.part[t]);
        acc = m;
    }
    fp12_to_bytes(out, &acc);
    free(c.part); free(th); free(args);
}

/* f^e for a little-endian integer e of `nbytes` bytes (the caller passes 3 (p^12 - 1) / r) */
void orc_fp12_pow(const uint8_t *in, const uint8_t *e, size_t nbytes, uint8_t *out) {
    fp12 f, r, t;
    fp12_from_bytes(&f, in);
    fp12_one(&r);
    for (size_t i = nbytes * 8; i-- > 0;) {
        fp12_mul(&t, &r, &r);
        r = t;
        if ((e[i >> 3] >> (i & 7)) & 1) { fp12_mul(&t, &r, &f); r = t; }
    }
    fp12_to_bytes(out, &r);
}

void orc_fp12_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) {
    fp12 x, y, r;
    fp12_from_bytes(&x, a); fp12_from_bytes(&y, b);
    fp12_mul(&r, &x, &y);
    fp12_to_bytes(out, &r);
}
