/* CPU ORACLE (test infrastructure only) — NOT PRODUCT CODE.
 *
 * Group-law + Pippenger template, instantiated twice by msm_oracle.c:
 *   G1 over Fp  (FE = fp,  F(x) = fp_##x,  G(x) = g1_##x)   — restates src/g1.rs:602-619
 *   G2 over Fp2 (FE = fp2, F(x) = fp2_##x, G(x) = g2_##x)   — restates src/g2.rs:582-599
 * (paths relative to /root/reference).  The reference delegates to blstrs::G{1,2}Projective::multi_exp
 * -> blst =0.3.10 `blst_p{1,2}s_mult_pippenger` (absent from /root/reference); this restates that
 * published algorithm's shape: signed (Booth) digits, window(N) = blst's heuristic, XYZZ buckets,
 * running-sum bucket integration, (point-slice x window) tiling over threads, Horner fold.
 *
 * Layouts: affine (x, y) with all-zero = infinity; Jacobian (X, Y, Z) with Z = 0 = infinity
 * (blst_p1_affine / blst_p1 / blst_p2_affine / blst_p2: src/gpu.rs:69-71,149-156,185-186).
 */

typedef struct { FE x, y; } G(affine);
typedef struct { FE x, y, z; } G(jac);
typedef struct { FE x, y, zz, zzz; } G(xyzz); /* zz == 0 <=> infinity */

static inline int G(aff_is_inf)(const G(affine) *p) { return F(is_zero)(&p->x) && F(is_zero)(&p->y); }
static inline int G(jac_is_inf)(const G(jac) *p) { return F(is_zero)(&p->z); }
static inline void G(jac_set_inf)(G(jac) *p) { memset(p, 0, sizeof *p); }
static inline void G(xyzz_set_inf)(G(xyzz) *p) { memset(p, 0, sizeof *p); }

static void G(jac_double)(G(jac) *r, const G(jac) *p) { /* dbl-2009-l, a = 0 */
    if (G(jac_is_inf)(p) || F(is_zero)(&p->y)) { G(jac_set_inf)(r); return; }
    FE A, B, C, D, E, Fq, t, X3, Y3, Z3;
    F(sqr)(&A, &p->x);
    F(sqr)(&B, &p->y);
    F(sqr)(&C, &B);
    F(add)(&t, &p->x, &B);
    F(sqr)(&t, &t);
    F(sub)(&t, &t, &A);
    F(sub)(&t, &t, &C);
    F(add)(&D, &t, &t);
    F(add)(&E, &A, &A);
    F(add)(&E, &E, &A);
    F(sqr)(&Fq, &E);
    F(add)(&t, &D, &D);
    F(sub)(&X3, &Fq, &t);
    F(sub)(&t, &D, &X3);
    F(mul)(&Y3, &E, &t);
    F(add)(&C, &C, &C); F(add)(&C, &C, &C); F(add)(&C, &C, &C);
    F(sub)(&Y3, &Y3, &C);
    F(mul)(&Z3, &p->y, &p->z);
    F(add)(&Z3, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void G(jac_add)(G(jac) *r, const G(jac) *p, const G(jac) *q) { /* add-2007-bl, complete via branches */
    if (G(jac_is_inf)(p)) { *r = *q; return; }
    if (G(jac_is_inf)(q)) { *r = *p; return; }
    FE Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
    F(sqr)(&Z1Z1, &p->z);
    F(sqr)(&Z2Z2, &q->z);
    F(mul)(&U1, &p->x, &Z2Z2);
    F(mul)(&U2, &q->x, &Z1Z1);
    F(mul)(&t, &q->z, &Z2Z2);
    F(mul)(&S1, &p->y, &t);
    F(mul)(&t, &p->z, &Z1Z1);
    F(mul)(&S2, &q->y, &t);
    if (F(eq)(&U1, &U2)) {
        if (F(eq)(&S1, &S2)) { G(jac_double)(r, p); return; }
        G(jac_set_inf)(r);
        return;
    }
    F(sub)(&H, &U2, &U1);
    F(sub)(&Rr, &S2, &S1);
    F(sqr)(&HH, &H);
    F(mul)(&HHH, &H, &HH);
    F(mul)(&V, &U1, &HH);
    F(sqr)(&X3, &Rr);
    F(sub)(&X3, &X3, &HHH);
    F(sub)(&X3, &X3, &V);
    F(sub)(&X3, &X3, &V);
    F(sub)(&t, &V, &X3);
    F(mul)(&Y3, &Rr, &t);
    F(mul)(&t, &S1, &HHH);
    F(sub)(&Y3, &Y3, &t);
    F(mul)(&Z3, &p->z, &q->z);
    F(mul)(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void G(jac_from_affine)(G(jac) *r, const G(affine) *p) {
    if (G(aff_is_inf)(p)) { G(jac_set_inf)(r); return; }
    r->x = p->x; r->y = p->y; r->z = F(one)();
}

static void G(jac_to_affine)(G(affine) *r, const G(jac) *p) {
    if (G(jac_is_inf)(p)) { memset(r, 0, sizeof *r); return; }
    FE zi, zi2, zi3;
    F(inv)(&zi, &p->z);
    F(sqr)(&zi2, &zi);
    F(mul)(&zi3, &zi2, &zi);
    F(mul)(&r->x, &p->x, &zi2);
    F(mul)(&r->y, &p->y, &zi3);
}

static void G(jac_add_affine)(G(jac) *r, const G(jac) *p, const G(affine) *q) {
    G(jac) t;
    G(jac_from_affine)(&t, q);
    G(jac_add)(r, p, &t);
}

/* XYZZ <- XYZZ + affine (madd-2008-s), complete: handles inf / equal / opposite. neg != 0 adds -q. */
static void G(xyzz_add_affine)(G(xyzz) *r, const G(affine) *q, int neg) {
    if (G(aff_is_inf)(q)) return;
    FE qy = q->y;
    if (neg) F(neg)(&qy, &qy);
    if (F(is_zero)(&r->zz)) { r->x = q->x; r->y = qy; r->zz = F(one)(); r->zzz = F(one)(); return; }
    FE U2, S2, Pp, Rr, PP, PPP, Q, t, X3, Y3;
    F(mul)(&U2, &q->x, &r->zz);
    F(mul)(&S2, &qy, &r->zzz);
    F(sub)(&Pp, &U2, &r->x);
    F(sub)(&Rr, &S2, &r->y);
    if (F(is_zero)(&Pp)) {
        if (F(is_zero)(&Rr)) { /* doubling of the affine point (mdbl-2008-s-1) */
            FE U, V, W, S, M;
            F(add)(&U, &qy, &qy);
            F(sqr)(&V, &U);
            F(mul)(&W, &U, &V);
            F(mul)(&S, &q->x, &V);
            F(sqr)(&M, &q->x);
            F(add)(&t, &M, &M);
            F(add)(&M, &t, &M);
            F(sqr)(&X3, &M);
            F(sub)(&X3, &X3, &S);
            F(sub)(&X3, &X3, &S);
            F(sub)(&t, &S, &X3);
            F(mul)(&Y3, &M, &t);
            F(mul)(&t, &W, &qy);
            F(sub)(&Y3, &Y3, &t);
            r->x = X3; r->y = Y3; r->zz = V; r->zzz = W;
            return;
        }
        G(xyzz_set_inf)(r);
        return;
    }
    F(sqr)(&PP, &Pp);
    F(mul)(&PPP, &Pp, &PP);
    F(mul)(&Q, &r->x, &PP);
    F(sqr)(&X3, &Rr);
    F(sub)(&X3, &X3, &PPP);
    F(sub)(&X3, &X3, &Q);
    F(sub)(&X3, &X3, &Q);
    F(sub)(&t, &Q, &X3);
    F(mul)(&Y3, &Rr, &t);
    F(mul)(&t, &r->y, &PPP);
    F(sub)(&Y3, &Y3, &t);
    F(mul)(&r->zz, &r->zz, &PP);
    F(mul)(&r->zzz, &r->zzz, &PPP);
    r->x = X3; r->y = Y3;
}

/* x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2.  Take Z' := ZZ: Z'^2 = ZZ^2, Z'^3 = ZZ^3 = ZZZ^2, so
 * (X*ZZ, Y*ZZZ, ZZ) is a Jacobian triple of the same point (2 multiplications, no inversion). */
static void G(xyzz_to_jac)(G(jac) *r, const G(xyzz) *p) {
    if (F(is_zero)(&p->zz)) { G(jac_set_inf)(r); return; }
    F(mul)(&r->x, &p->x, &p->zz);
    F(mul)(&r->y, &p->y, &p->zzz);
    r->z = p->zz;
}

/* scalar helpers: scalars are 4 x u64 canonical little-endian integers < r */
static inline unsigned G(get_bits)(const uint64_t *s, unsigned off, unsigned n) {
    if (off >= 256) return 0;
    unsigned w = off >> 6, b = off & 63;
    uint64_t v = s[w] >> b;
    if (b + n > 64 && w + 1 < 4) v |= s[w + 1] << (64 - b);
    return (unsigned)(v & ((1ULL << n) - 1));
}

/* naive double-and-add (the definition sum_i s_i*P_i asserted by src/tests.rs:57-67) */
static void G(mul_naive)(G(jac) *r, const G(affine) *p, const uint64_t *s) {
    G(jac) acc; G(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        G(jac_double)(&acc, &acc);
        if ((s[i >> 6] >> (i & 63)) & 1) G(jac_add_affine)(&acc, &acc, p);
    }
    *r = acc;
}

/* blst window heuristic (SURVEY Appendix A [ext]) */
static unsigned G(pippenger_window)(size_t n) {
    unsigned wbits = 0;
    while ((n >> (wbits + 1)) != 0) wbits++; /* floor(log2 n) */
    if (n == 0) return 1;
    return wbits > 12 ? wbits - 3 : (wbits > 4 ? wbits - 2 : (wbits ? 2 : 1));
}

/* One tile: points [lo, hi), window index win (bits [win*w, win*w + w)), signed digits with the carry
 * convention d = raw + carry_in; if d > 2^(w-1): d -= 2^w, carry_out = 1.  The carry into window `win`
 * is recomputed from the lower bits (it only depends on them).  Result = sum_b b * bucket[b]. */
static void G(pippenger_tile)(G(jac) *out, const G(affine) *bases, const uint64_t *scalars, size_t lo, size_t hi,
                              unsigned win, unsigned w, G(xyzz) *buckets) {
    size_t nb = (size_t)1 << (w - 1);
    for (size_t b = 0; b <= nb; b++) G(xyzz_set_inf)(&buckets[b]);
    for (size_t i = lo; i < hi; i++) {
        const uint64_t *s = scalars + 4 * i;
        /* carry into this window: 1 iff the (win*w)-bit suffix, recoded, overflowed. Walk the lower windows. */
        unsigned carry = 0;
        for (unsigned k = 0; k < win; k++) {
            unsigned raw = G(get_bits)(s, k * w, w) + carry;
            carry = raw > (1u << (w - 1));
        }
        unsigned raw = G(get_bits)(s, win * w, w) + carry;
        int neg = 0;
        if (raw > (1u << (w - 1))) { raw = (1u << w) - raw; neg = 1; }
        if (raw) G(xyzz_add_affine)(&buckets[raw], &bases[i], neg);
    }
    /* running-sum integration: sum_b b*B_b */
    G(jac) run, acc, t;
    G(jac_set_inf)(&run); G(jac_set_inf)(&acc);
    for (size_t b = nb; b >= 1; b--) {
        G(xyzz_to_jac)(&t, &buckets[b]);
        G(jac_add)(&run, &run, &t);
        G(jac_add)(&acc, &acc, &run);
    }
    *out = acc;
}
