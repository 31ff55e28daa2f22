// Host-side (CPU) BLS12-381 arithmetic used by the shipped library for the O(windows) tail of an MSM:
// combining the per-chunk bucket sums the GPU returns and the Horner fold over windows — the step the
// reference also performs on the host (/root/reference/src/gpu.rs:193-209) — plus folding per-GPU partials.
// Operates directly on the reference's in-memory form: blst_fp = 6 x u64 LE limbs, Montgomery R = 2^384
// (/root/reference/src/fp.rs:25-32 modulus), Jacobian (X, Y, Z) with Z = 0 <=> infinity.
//
// Independent of oracle/ (which is test infrastructure): nothing here includes or links it.
#pragma once
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

// BMI2 + ADX are enabled for the functions of THIS header only (clang's per-function target attribute), not for the translation
// units that include it: everything outside — mi_msm_init and its cpu_ok() check, std::string / vector code, static initialisers —
// is compiled for the baseline x86-64, so a CPU without the extensions gets MI_E_UNSUPPORTED from the check instead of a SIGILL
// from some shlx / mulx the compiler placed in front of it (the -mbmi2 -madx flags of rounds 1-2 applied to all host code).
#if defined(__clang__) && defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define HOSTEC_MULX_ADX 1
#pragma clang attribute push(__attribute__((target("bmi2,adx"))), apply_to = function)
#define HOSTEC_FLATTEN __attribute__((flatten))   // the point formulas inline every field operation (callers outside this header cannot:
                                                   // they are compiled without the extensions and call these as functions)
#else
#define HOSTEC_FLATTEN
#endif

namespace hostec {

typedef unsigned __int128 u128;

struct Fp {
    uint64_t l[6];
    static constexpr uint64_t MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                        0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static constexpr uint64_t NINV = 0x89f3fffcfffcfffdULL;  // -p^-1 mod 2^64
    static Fp zero() { Fp r; memset(r.l, 0, sizeof r.l); return r; }
    static Fp one() {  // 2^384 mod p
        Fp r = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
                 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
        return r;
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3] | l[4] | l[5]) == 0; }
    bool operator==(const Fp& o) const { return memcmp(l, o.l, sizeof l) == 0; }

    static bool geq_mod(const uint64_t* t) {
        for (int i = 5; i >= 0; i--) {
            if (t[i] != MOD[i]) return t[i] > MOD[i];
        }
        return true;
    }
    static void sub_mod(uint64_t* t) {
        uint64_t borrow = 0;
        for (int i = 0; i < 6; i++) {
            u128 d = (u128)t[i] - MOD[i] - borrow;
            t[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
    }
    // branch-free: the sum (difference) and its correction by p are both computed, one is selected by the final borrow
    Fp operator+(const Fp& o) const {
        uint64_t r[6], d[6];
        u128 c = 0;
        for (int i = 0; i < 6; i++) { c += (u128)l[i] + o.l[i]; r[i] = (uint64_t)c; c >>= 64; }
        uint64_t borrow = 0;
        for (int i = 0; i < 6; i++) {
            u128 x = (u128)r[i] - MOD[i] - borrow;
            d[i] = (uint64_t)x;
            borrow = (uint64_t)(x >> 64) & 1;
        }
        const uint64_t keep = (uint64_t)0 - borrow;   // all ones: r < p, keep r (a + b < 2p < 2^384: no carry out of r)
        Fp out;
        for (int i = 0; i < 6; i++) out.l[i] = (r[i] & keep) | (d[i] & ~keep);
        return out;
    }
    Fp operator-(const Fp& o) const {
        uint64_t r[6];
        uint64_t borrow = 0;
        for (int i = 0; i < 6; i++) {
            u128 x = (u128)l[i] - o.l[i] - borrow;
            r[i] = (uint64_t)x;
            borrow = (uint64_t)(x >> 64) & 1;
        }
        const uint64_t fix = (uint64_t)0 - borrow;    // all ones: went negative, add p back
        Fp out;
        u128 c = 0;
        for (int i = 0; i < 6; i++) { c += (u128)r[i] + (MOD[i] & fix); out.l[i] = (uint64_t)c; c >>= 64; }
        return out;
    }
    // Montgomery multiplication, coarsely integrated operand scanning (CIOS), fully unrolled: one pass of a * b_i and one
    // reduction round per word of b.  With BMI2 + ADX (enabled for this header's functions, see the top; mi_msm_init refuses
    // CPUs without them) each pass is six mulx feeding two independent carry chains (adcx / adox): 79 cycles on a 2.1 GHz
    // Xeon against 124 for the portable unsigned __int128 form below (kept for other compilers: gcc serialises the chains).
#if defined(HOSTEC_MULX_ADX)
    Fp operator*(const Fp& o) const {
        typedef unsigned long long u64;
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#pragma clang loop unroll(full)
        for (int i = 0; i < 6; i++) {
            const u64 bi = o.l[i];
            u64 lo, h0, h1, h2, h3, h4, h5, drop;
            unsigned char c1 = 0, c2 = 0;
            lo = _mulx_u64(l[0], bi, &h0); c1 = _addcarryx_u64(c1, t0, lo, &t0);
            lo = _mulx_u64(l[1], bi, &h1); c1 = _addcarryx_u64(c1, t1, lo, &t1); c2 = _addcarryx_u64(c2, t1, h0, &t1);
            lo = _mulx_u64(l[2], bi, &h2); c1 = _addcarryx_u64(c1, t2, lo, &t2); c2 = _addcarryx_u64(c2, t2, h1, &t2);
            lo = _mulx_u64(l[3], bi, &h3); c1 = _addcarryx_u64(c1, t3, lo, &t3); c2 = _addcarryx_u64(c2, t3, h2, &t3);
            lo = _mulx_u64(l[4], bi, &h4); c1 = _addcarryx_u64(c1, t4, lo, &t4); c2 = _addcarryx_u64(c2, t4, h3, &t4);
            lo = _mulx_u64(l[5], bi, &h5); c1 = _addcarryx_u64(c1, t5, lo, &t5); c2 = _addcarryx_u64(c2, t5, h4, &t5);
            c1 = _addcarryx_u64(c1, t6, 0, &t6); c2 = _addcarryx_u64(c2, t6, h5, &t6);   // the running value stays below 2p 2^64
            const u64 m = t0 * NINV;
            c1 = 0; c2 = 0;
            lo = _mulx_u64(m, MOD[0], &h0); c1 = _addcarryx_u64(c1, t0, lo, &drop);
            lo = _mulx_u64(m, MOD[1], &h1); c1 = _addcarryx_u64(c1, t1, lo, &t0); c2 = _addcarryx_u64(c2, t0, h0, &t0);
            lo = _mulx_u64(m, MOD[2], &h2); c1 = _addcarryx_u64(c1, t2, lo, &t1); c2 = _addcarryx_u64(c2, t1, h1, &t1);
            lo = _mulx_u64(m, MOD[3], &h3); c1 = _addcarryx_u64(c1, t3, lo, &t2); c2 = _addcarryx_u64(c2, t2, h2, &t2);
            lo = _mulx_u64(m, MOD[4], &h4); c1 = _addcarryx_u64(c1, t4, lo, &t3); c2 = _addcarryx_u64(c2, t3, h3, &t3);
            lo = _mulx_u64(m, MOD[5], &h5); c1 = _addcarryx_u64(c1, t5, lo, &t4); c2 = _addcarryx_u64(c2, t4, h4, &t4);
            c1 = _addcarryx_u64(c1, t6, 0, &t5); c2 = _addcarryx_u64(c2, t5, h5, &t5);
            t6 = (u64)c1 + (u64)c2;
        }
        Fp r = {{t0, t1, t2, t3, t4, t5}};
        if (t6 || geq_mod(r.l)) sub_mod(r.l);
        return r;
    }
#else
    Fp operator*(const Fp& o) const {
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#if defined(__clang__)
#pragma clang loop unroll(full)
#endif
        for (int i = 0; i < 6; i++) {
            const uint64_t bi = o.l[i];
            u128 c;
            c = (u128)l[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
            c += (u128)l[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
            c += (u128)l[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
            c += (u128)l[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
            c += (u128)l[4] * bi + t4; t4 = (uint64_t)c; c >>= 64;
            c += (u128)l[5] * bi + t5; t5 = (uint64_t)c; c >>= 64;
            t6 += (uint64_t)c;   // the running value stays below 2p 2^64: t6 never overflows
            const uint64_t m = t0 * NINV;
            c = (u128)m * MOD[0] + t0; c >>= 64;
            c += (u128)m * MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;
            c += (u128)m * MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;
            c += (u128)m * MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;
            c += (u128)m * MOD[4] + t4; t3 = (uint64_t)c; c >>= 64;
            c += (u128)m * MOD[5] + t5; t4 = (uint64_t)c; c >>= 64;
            c += t6; t5 = (uint64_t)c; t6 = (uint64_t)(c >> 64);
        }
        Fp r = {{t0, t1, t2, t3, t4, t5}};
        if (t6 || geq_mod(r.l)) sub_mod(r.l);
        return r;
    }
#endif
    Fp sqr() const { return *this * *this; }
    Fp dbl() const { return *this + *this; }
    // a^(p-2) (Fermat); inv(0) = 0.  Used once per normalize_batch call on the top of the product tree.
    Fp inv() const {
        uint64_t e[6];
        memcpy(e, MOD, sizeof e);
        e[0] -= 2;
        Fp acc = one(), base = *this;
        for (int i = 0; i < 384; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * base;
            base = base.sqr();
        }
        return acc;
    }
};

struct Fp2 {
    Fp c0, c1;  // c0 + c1*u, u^2 = -1  (/root/reference/src/fp2.rs:228)
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    HOSTEC_FLATTEN Fp2 operator*(const Fp2& o) const {
        Fp a = c0 * o.c0, b = c1 * o.c1;
        Fp m = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, m - a - b};
    }
    HOSTEC_FLATTEN Fp2 sqr() const {
        Fp s = c0 + c1, d = c0 - c1, m = c0 * c1;
        return {s * d, m + m};
    }
    Fp2 dbl() const { return *this + *this; }
    Fp2 inv() const {  // conj(a) / (c0^2 + c1^2)
        Fp n = (c0.sqr() + c1.sqr()).inv();
        return {c0 * n, Fp::zero() - c1 * n};
    }
};

template <class FE>
struct Jac {
    FE x, y, z;
    static Jac inf() { return {FE::zero(), FE::zero(), FE::zero()}; }
    bool is_inf() const { return z.is_zero(); }

    HOSTEC_FLATTEN Jac dbl() const {  // a = 0 short Weierstrass
        if (is_inf() || y.is_zero()) return inf();
        FE xx = x.sqr(), yy = y.sqr(), yyyy = yy.sqr();
        FE s = ((x + yy).sqr() - xx - yyyy).dbl();
        FE m = xx.dbl() + xx;
        FE x3 = m.sqr() - s.dbl();
        FE y3 = m * (s - x3) - yyyy.dbl().dbl().dbl();
        FE z3 = (y * z).dbl();
        return {x3, y3, z3};
    }
    HOSTEC_FLATTEN Jac add(const Jac& q) const {
        if (is_inf()) return q;
        if (q.is_inf()) return *this;
        FE z1z1 = z.sqr(), z2z2 = q.z.sqr();
        FE u1 = x * z2z2, u2 = q.x * z1z1;
        FE s1 = y * q.z * z2z2, s2 = q.y * z * z1z1;
        if (u1 == u2) return s1 == s2 ? dbl() : inf();
        FE h = u2 - u1, r = s2 - s1;
        FE hh = h.sqr(), hhh = h * hh, v = u1 * hh;
        FE x3 = r.sqr() - hhh - v.dbl();
        FE y3 = r * (v - x3) - s1 * hhh;
        FE z3 = z * q.z * h;
        return {x3, y3, z3};
    }
    Jac dbl_n(unsigned k) const {
        Jac r = *this;
        for (unsigned i = 0; i < k; i++) r = r.dbl();
        return r;
    }
};

using G1 = Jac<Fp>;    // 144 B == blst_p1
using G2 = Jac<Fp2>;   // 288 B == blst_p2
static_assert(sizeof(G1) == 144 && sizeof(G2) == 288, "layout must match blst_p1 / blst_p2");

}  // namespace hostec

#if defined(HOSTEC_MULX_ADX)
#pragma clang attribute pop
#endif
