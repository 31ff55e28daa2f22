// Host build (g++) of the device field/curve headers, so their arithmetic can be checked on a CPU-only box
// against the oracle before any GPU run.  Test helper only — not part of the shipped library.
#include <cstdint>
#include <cstring>
#include "../../ark-blst_amd/csrc/ec.cuh"

using namespace fp28;
using F = ec::FpOps;
using X = ec::Xyzz<F>;

static Fp load_blst(const uint8_t* p) {
    uint32_t w[12];
    memcpy(w, p, 48);
    return fp_from_blst(w);
}
static void store_blst(uint8_t* p, const Fp& a) {
    uint32_t w[12];
    fp_to_blst(w, a);
    memcpy(p, w, 48);
}

extern "C" {

// out = a*b in blst Montgomery form (n elements)
void h28_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int use_sqr) {
    for (size_t i = 0; i < n; i++) {
        Fp x = load_blst(a + 48 * i), y = load_blst(b + 48 * i);
        Fp z = use_sqr ? fp_sqr(x) : fp_mul(x, y);
        store_blst(out + 48 * i, z);
    }
}
// out = a + b, a - b (mod p) canonical blst form
void h28_fp_addsub(const uint8_t* a, const uint8_t* b, uint8_t* out_add, uint8_t* out_sub, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fp x = load_blst(a + 48 * i), y = load_blst(b + 48 * i);
        store_blst(out_add + 48 * i, fp_add(x, y));
        store_blst(out_sub + 48 * i, fp_sub<4>(x, y));
    }
}
// pack/unpack round trip
void h28_roundtrip(const uint8_t* a, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) store_blst(out + 48 * i, load_blst(a + 48 * i));
}
static void store_xyzz_as_jac(uint8_t* out, const X& acc) {
    if (ec::xyzz_is_inf(acc)) { memset(out, 0, 144); return; }
    store_blst(out, fp_mul(acc.x, acc.zz));
    store_blst(out + 48, fp_mul(acc.y, acc.zzz));
    store_blst(out + 96, acc.zz);
}
// signed sum of affine points with the mixed-add path: out (Jacobian blst bytes) = sum (+/-) P_i
void h28_g1_madd_chain(const uint8_t* bases, const uint8_t* neg, size_t n, uint8_t* out) {
    X acc = ec::xyzz_inf<F>();
    bool inf = true;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = bases + 96 * i;
        bool zero = true;
        for (int k = 0; k < 96; k++) zero &= p[k] == 0;
        if (zero) continue;
        Fp x = load_blst(p), y = load_blst(p + 48);
        if (neg && neg[i]) y = fp_neg<4>(y);
        if (inf) { acc = ec::xyzz_from_affine<F>(x, y); inf = false; continue; }
        bool pz;
        X r = ec::xyzz_madd_core<F>(acc, x, y, pz);
        if (pz) { r = ec::xyzz_madd_special<F>(acc, x, y); inf = ec::xyzz_is_inf(r); }
        acc = r;
    }
    if (inf) acc = ec::xyzz_inf<F>();
    store_xyzz_as_jac(out, acc);
}
// tree sum with the complete XYZZ+XYZZ addition (exercises inf / doubling / cancellation)
void h28_g1_add_tree(const uint8_t* bases, size_t n, uint8_t* out) {
    X* v = new X[n + 1];
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = bases + 96 * i;
        bool zero = true;
        for (int k = 0; k < 96; k++) zero &= p[k] == 0;
        v[i] = zero ? ec::xyzz_inf<F>() : ec::xyzz_from_affine<F>(load_blst(p), load_blst(p + 48));
    }
    size_t m = n;
    while (m > 1) {
        size_t h = (m + 1) / 2;
        for (size_t i = 0; i + h < m; i++) v[i] = ec::xyzz_add<F>(v[i], v[i + h]);
        m = h;
    }
    X r = n ? v[0] : ec::xyzz_inf<F>();
    store_xyzz_as_jac(out, r);
    delete[] v;
}
// out = 2^k * P via xyzz_dbl_n
void h28_g1_dbl_n(const uint8_t* base, int k, uint8_t* out) {
    X a = ec::xyzz_from_affine<F>(load_blst(base), load_blst(base + 48));
    store_xyzz_as_jac(out, ec::xyzz_dbl_n<F>(a, k));
}
}
