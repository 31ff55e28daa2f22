// Host build (g++) of the pairing arithmetic in ark-blst_amd/csrc/pairing.cuh with the REAL 28-bit field, so the
// tower, the Miller loop and the final exponentiation can be checked against the oracle on a CPU-only box.
// Test helper only — not part of the shipped library.
#include <cstdint>
#include <cstring>
#include "../../ark-blst_amd/csrc/pairing.cuh"

using namespace fp28;
using T = pairing::Tower<pairing::PF2>;

static Fp load_fp(const uint8_t* p) {
    uint32_t w[12];
    memcpy(w, p, 48);
    return fp_from_blst(w);
}
static void store_fp(uint8_t* p, const Fp& a) {
    uint32_t w[12];
    fp_to_blst(w, a);
    memcpy(p, w, 48);
}
static ec::Fp2 load_fp2(const uint8_t* p) { return ec::Fp2{load_fp(p), load_fp(p + 48)}; }
static T::E12 load_fp12(const uint8_t* p) {   // member by member (no pointer walk across struct members)
    T::E12 r;
    ec::Fp2* c[6] = {&r.c0.c0, &r.c0.c1, &r.c0.c2, &r.c1.c0, &r.c1.c1, &r.c1.c2};
    for (int i = 0; i < 6; i++) *c[i] = load_fp2(p + 96 * i);
    return r;
}
static void store_fp12(uint8_t* p, const T::E12& a) {
    const ec::Fp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) {
        store_fp(p + 96 * i, c[i]->c0);
        store_fp(p + 96 * i + 48, c[i]->c1);
    }
}

extern "C" {

void hp_fp12_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { store_fp12(out, T::mul12(load_fp12(a), load_fp12(b))); }
void hp_fp12_sqr(const uint8_t* a, uint8_t* out) { store_fp12(out, T::sqr12(load_fp12(a))); }
void hp_fp12_inv(const uint8_t* a, uint8_t* out) { store_fp12(out, T::inv12(load_fp12(a))); }
void hp_fp12_frob(const uint8_t* a, uint8_t* out) { store_fp12(out, T::frob12(load_fp12(a))); }
void hp_final_exp(const uint8_t* a, uint8_t* out) { store_fp12(out, T::final_exp(load_fp12(a))); }
// p: blst_p1_affine (96 B), q: blst_p2_affine (192 B), both not infinity
void hp_miller_loop(const uint8_t* p, const uint8_t* q, uint8_t* out) {
    T::G1Pt g{fp_neg<4>(load_fp(p)), load_fp(p + 48)};
    store_fp12(out, T::miller_loop(g, load_fp2(q), load_fp2(q + 96)));
}

}  // extern "C"
