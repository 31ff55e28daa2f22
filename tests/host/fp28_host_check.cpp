// Host build (g++) of the device field/curve headers, so their arithmetic can be checked on a CPU-only box
// against the oracle before any GPU run.  Test helper only — not part of the shipped library.
#include <cstdint>
#include <cstring>
#include "../../ark-blst_amd/csrc/ec.cuh"

using namespace fp28;
using F = ec::FpOps;
using FI = ec::FpOpsInline;  // the accumulate hot loop's field (inlined multiplier, mul2add)
using X = ec::Xyzz<FI>;
using Pj = ec::Proj<F>;

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
static bool is_zero96(const uint8_t* p) {
    bool zero = true;
    for (int k = 0; k < 96; k++) zero &= p[k] == 0;
    return zero;
}
// projective -> blst Jacobian bytes (X Z, Y Z^2, Z); infinity -> zeros
static void store_proj_as_jac(uint8_t* out, const Pj& p) {
    uint8_t z[48];
    store_blst(z, p.z);
    bool any = false;
    for (int k = 0; k < 48; k++) any |= z[k] != 0;
    if (!any) { memset(out, 0, 144); return; }
    Fp zz = fp_mul(p.z, p.z);
    store_blst(out, fp_mul(p.x, p.z));
    store_blst(out + 48, fp_mul(p.y, zz));
    memcpy(out + 96, z, 48);
}

extern "C" {

void h28_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int use_sqr) {
    for (size_t i = 0; i < n; i++) {
        Fp x = load_blst(a + 48 * i), y = load_blst(b + 48 * i);
        store_blst(out + 48 * i, use_sqr ? fp_sqr(x) : fp_mul(x, y));
    }
}
void h28_fp_addsub(const uint8_t* a, const uint8_t* b, uint8_t* out_add, uint8_t* out_sub, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fp x = load_blst(a + 48 * i), y = load_blst(b + 48 * i);
        store_blst(out_add + 48 * i, fp_add(x, y));
        store_blst(out_sub + 48 * i, fp_sub<4>(x, y));
    }
}
void h28_fp_mul_small(const uint8_t* a, uint8_t* out3, uint8_t* out12, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fp x = load_blst(a + 48 * i);
        store_blst(out3 + 48 * i, fp_mul_small<3>(x));
        store_blst(out12 + 48 * i, fp_mul_small<12>(x));
    }
}
void h28_roundtrip(const uint8_t* a, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) store_blst(out + 48 * i, load_blst(a + 48 * i));
}
// The accumulate kernel's per-bucket logic: XYZZ hot loop, then the complete cold path from the first
// exceptional pair on.  *took_cold reports whether the cold path ran.
void h28_g1_bucket(const uint8_t* bases, const uint8_t* neg, size_t n, uint8_t* out, int* took_cold) {
    X acc;
    acc.x = fp_zero(); acc.y = fp_zero(); acc.zz = fp_zero(); acc.zzz = fp_zero();
    bool inf = true;
    size_t e = 0;
    for (; e < n; e++) {
        const uint8_t* p = bases + 96 * e;
        Fp x = load_blst(p), y = load_blst(p + 48);
        if (neg && neg[e]) y = FI::neg_l<4>(y);
        if (inf) { acc.x = x; acc.y = FI::norm(y); acc.zz = fp_one(); acc.zzz = fp_one(); inf = false; }
        else if (ec::xyzz_madd<FI>(acc, x, y)) break;
    }
    Pj o = ec::proj_inf<F>();
    if (!inf) { ec::Xyzz<F> a2; a2.x = acc.x; a2.y = acc.y; a2.zz = acc.zz; a2.zzz = acc.zzz; o = ec::xyzz_to_proj<F>(a2); }
    *took_cold = e < n;
    for (; e < n; e++) {
        const uint8_t* p = bases + 96 * e;
        Fp x = load_blst(p), y = load_blst(p + 48);
        if (neg && neg[e]) y = fp_neg<4>(y);
        Pj q = ec::proj_from_affine<F>(x, y);
        ec::proj_add<F>(o, q);
    }
    store_proj_as_jac(out, o);
}
// tree sum with the complete projective addition (infinity inputs = all-zero affine)
void h28_g1_add_tree(const uint8_t* bases, size_t n, uint8_t* out) {
    Pj* v = new Pj[n + 1];
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = bases + 96 * i;
        v[i] = is_zero96(p) ? ec::proj_inf<F>() : ec::proj_from_affine<F>(load_blst(p), load_blst(p + 48));
    }
    size_t m = n;
    while (m > 1) {
        size_t h = (m + 1) / 2;
        for (size_t i = 0; i + h < m; i++) ec::proj_add<F>(v[i], v[i + h]);
        m = h;
    }
    Pj r = n ? v[0] : ec::proj_inf<F>();
    store_proj_as_jac(out, r);
    delete[] v;
}
void h28_g1_dbl_n(const uint8_t* base, int k, uint8_t* out) {
    Pj a = is_zero96(base) ? ec::proj_inf<F>() : ec::proj_from_affine<F>(load_blst(base), load_blst(base + 48));
    ec::proj_dbl_n<F>(a, k);
    store_proj_as_jac(out, a);
}
// fp_reduce_small on m * a (m <= 127 copies of a value < 2p added up): raw limbs in and out, the caller does the integer arithmetic
void h28_fp_reduce_small(const uint8_t* a, int m, uint32_t* in_limbs, uint32_t* out_limbs) {
    Fp x = load_blst(a), v = fp_zero();
    for (int k = 0; k < m; k++) v = fp_add(v, x);
    Fp r = fp_reduce_small(v);
    memcpy(in_limbs, v.l, sizeof v.l);
    memcpy(out_limbs, r.l, sizeof r.l);
}
// the G1 subgroup test of the decoders and of Valid::check (codec_kernels.cuh g1_in_subgroup), and its ladder's result [z^2] P as
// Jacobian bytes (zeros when Z == 0)
int h28_g1_torsion_free(const uint8_t* base) { return ec::g1_torsion_free<ec::FpOpsInlinePS, F>(load_blst(base), load_blst(base + 48)) ? 1 : 0; }
void h28_g1_mul_z2(const uint8_t* base, uint8_t* out) {
    ec::JacFp p;
    p.x = load_blst(base); p.y = load_blst(base + 48); p.z = fp_one();
    ec::JacFp q = ec::jac_mul_z<ec::FpOpsInlinePS, F, false>(ec::jac_mul_z<ec::FpOpsInlinePS, F, true>(p));
    if (fp_is_zero_any(q.z)) { memset(out, 0, 144); return; }
    store_blst(out, q.x); store_blst(out + 48, q.y); store_blst(out + 96, q.z);
}
}
