// Reproducer for the round-3 miscompare: the out-of-line tower functions of pairing.cuh (sqr12, mul_by_014, conj12: each ends in
// norm6 = six calls of the shared multiplier with the constant one as second operand), one call per kernel, 64 lanes with
// different inputs.  Built twice — fp_mul_call with the operand-scanning body (shipped) and with the product-scanning body
// (-DMI_CALL_PS) — the two binaries must print identical checksums; the host runs the same source as the reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMI_CALL_PS] [-mllvm -amdgpu-spill-vgpr-to-agpr=0] -I ark-blst_amd/csrc -o repro tools/call_abi/repro_tower.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "pairing.cuh"
using T = pairing::Tower<pairing::PF2>;
using E12 = T::E12;
static __host__ __device__ E12 make(uint32_t seed) {   // pseudo-random N-form limbs < 2^28 (values < 2^392: legal multiplier input < ~50p? no: keep < 2p)
    E12 r;   // the twelve Fp components member by member (no pointer walk across struct members: the source is to be beyond reproach)
    fp28::Fp* c[12] = {&r.c0.c0.c0, &r.c0.c0.c1, &r.c0.c1.c0, &r.c0.c1.c1, &r.c0.c2.c0, &r.c0.c2.c1,
                       &r.c1.c0.c0, &r.c1.c0.c1, &r.c1.c1.c0, &r.c1.c1.c1, &r.c1.c2.c0, &r.c1.c2.c1};
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < fp28::NL; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            c[i]->l[k] = (uint32_t)(s >> 36) & (k == fp28::NL - 1 ? 0xFFFFu : fp28::MASK);   // top limb small: value < 2^380 < p
        }
    return r;
}
static __host__ __device__ uint64_t sum(const E12& a) {
    const fp28::Fp* c[12] = {&a.c0.c0.c0, &a.c0.c0.c1, &a.c0.c1.c0, &a.c0.c1.c1, &a.c0.c2.c0, &a.c0.c2.c1,
                             &a.c1.c0.c0, &a.c1.c0.c1, &a.c1.c1.c0, &a.c1.c1.c1, &a.c1.c2.c0, &a.c1.c2.c1};
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < fp28::NL; k++) h = (h ^ c[i]->l[k]) * 1099511628211ull;
    return h;
}
__global__ void k_conj(uint64_t* out) { E12 f = make(threadIdx.x); out[threadIdx.x] = sum(T::conj12(f)); }
__global__ void k_sqr(uint64_t* out) { E12 f = make(threadIdx.x); out[threadIdx.x] = sum(T::sqr12(f)); }
__global__ void k_014(uint64_t* out) { E12 f = make(threadIdx.x); out[threadIdx.x] = sum(T::mul_by_014(f, f.c0.c1, f.c1.c0, f.c1.c2)); }
__global__ void k_chain(uint64_t* out) {   // the shape of the Miller loop body, a few rounds
    E12 f = make(threadIdx.x), g = make(threadIdx.x + 1000);
    for (int i = 0; i < 5; i++) { f = T::sqr12(f); f = T::mul_by_014(f, g.c0.c1, g.c1.c0, g.c1.c2); }
    out[threadIdx.x] = sum(T::conj12(f));
}
// the Miller loop of pairing.cuh (Tower::miller_loop) with a run-time number of rounds: the shape of the failing kernel
static __host__ __device__ E12 miller_rounds(uint32_t seed, int rounds) {
    E12 g = make(seed + 2000);
    T::G1Pt p{g.c0.c0.c0, g.c0.c0.c1};
    ec::Fp2 xq = g.c0.c1, yq = g.c0.c2;
    E12 f = T::one12();
    T::PT Tq = ec::proj_from_affine<pairing::PF2>(xq, yq);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i > 62 - rounds; i--) {
        ec::Fp2 c0, c1, c4;
        f = T::sqr12(f);
        T::line_dbl(Tq, p, c0, c1, c4);
        f = T::mul_by_014(f, c0, c1, c4);
        if ((fp28c::Z_ABS >> i) & 1) {
            T::line_add(Tq, xq, yq, p, c0, c1, c4);
            f = T::mul_by_014(f, c0, c1, c4);
        }
    }
    return T::conj12(f);
}
#if defined(STACK_SLACK)
// never called at run time (rounds >= 0): its only effect is that the kernel's private segment is STACK_SLACK bytes larger than the
// assembler's own sum of frame sizes says it needs to be
static __device__ __noinline__ uint64_t slack_frame(uint64_t* p) {
    volatile uint64_t buf[STACK_SLACK / 8];
    for (int i = 0; i < STACK_SLACK / 8; i++) buf[i] = p[i & 63] + i;
    uint64_t h = 0;
    for (int i = 0; i < STACK_SLACK / 8; i += 97) h ^= buf[i];
    return h;
}
#endif
__global__ void __launch_bounds__(64, 1) k_miller(uint64_t* out, int rounds) {
#if defined(STACK_SLACK)
    if (rounds < 0) { out[threadIdx.x] = slack_frame(out); return; }
#endif
    out[threadIdx.x] = sum(miller_rounds(threadIdx.x, rounds));
}

static __host__ __device__ uint64_t sum2(uint64_t h, const ec::Fp2& a) {
    for (int k = 0; k < fp28::NL; k++) h = (h ^ a.c0.l[k]) * 1099511628211ull;
    for (int k = 0; k < fp28::NL; k++) h = (h ^ a.c1.l[k]) * 1099511628211ull;
    return h;
}
// pieces of one round: which = 0 line_dbl only; 1 sqr12 + line_dbl; 2 line_dbl + mul_by_014; 3 line_dbl + conj12 of the input
static __host__ __device__ uint64_t piece(uint32_t seed, int which) {
    E12 g = make(seed + 2000), f = make(seed + 3000);
    T::G1Pt p{g.c0.c0.c0, g.c0.c0.c1};
    ec::Fp2 xq = g.c0.c1, yq = g.c0.c2;
    T::PT Tq = ec::proj_from_affine<pairing::PF2>(xq, yq);
    ec::Fp2 c0, c1, c4;
    if (which == 1) f = T::sqr12(f);
    T::line_dbl(Tq, p, c0, c1, c4);
    if (which == 2) f = T::mul_by_014(f, c0, c1, c4);
    if (which == 3) f = T::conj12(f);
    uint64_t h = sum(f);
    h = sum2(h, c0); h = sum2(h, c1); h = sum2(h, c4); h = sum2(h, Tq.x); h = sum2(h, Tq.y); h = sum2(h, Tq.z);
    return h;
}
template <int W>
__global__ void __launch_bounds__(64, 1) k_piece(uint64_t* out) { out[threadIdx.x] = piece(threadIdx.x, W); }

int main() {
    uint64_t* d;
    if (hipMalloc(&d, 64 * 8) != hipSuccess) return 2;
    const char* names[4] = {"conj12", "sqr12", "mul_by_014", "chain"};
    int bad = 0;
    for (int t = 0; t < 4; t++) {
        if (t == 0) hipLaunchKernelGGL(k_conj, dim3(1), dim3(64), 0, 0, d);
        if (t == 1) hipLaunchKernelGGL(k_sqr, dim3(1), dim3(64), 0, 0, d);
        if (t == 2) hipLaunchKernelGGL(k_014, dim3(1), dim3(64), 0, 0, d);
        if (t == 3) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d);
        std::vector<uint64_t> h(64);
        if (hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        int wrong = 0;
        for (uint32_t l = 0; l < 64; l++) {
            E12 f = make(l), g = make(l + 1000), r;
            if (t == 0) r = T::conj12(f);
            if (t == 1) r = T::sqr12(f);
            if (t == 2) r = T::mul_by_014(f, f.c0.c1, f.c1.c0, f.c1.c2);
            if (t == 3) { for (int i = 0; i < 5; i++) { f = T::sqr12(f); f = T::mul_by_014(f, g.c0.c1, g.c1.c0, g.c1.c2); } r = T::conj12(f); }
            if (sum(r) != h[l]) wrong++;
        }
        printf("%s: %d of 64 lanes differ from the host\n", names[t], wrong);
        bad += wrong;
    }
    for (int w = 0; w < 4; w++) {
        if (w == 0) hipLaunchKernelGGL(k_piece<0>, dim3(1), dim3(64), 0, 0, d);
        if (w == 1) hipLaunchKernelGGL(k_piece<1>, dim3(1), dim3(64), 0, 0, d);
        if (w == 2) hipLaunchKernelGGL(k_piece<2>, dim3(1), dim3(64), 0, 0, d);
        if (w == 3) hipLaunchKernelGGL(k_piece<3>, dim3(1), dim3(64), 0, 0, d);
        std::vector<uint64_t> h(64);
        if (hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        int wrong = 0;
        for (uint32_t l = 0; l < 64; l++) wrong += piece(l, w) != h[l];
        printf("piece %d (0 line_dbl; 1 sqr12 + line_dbl; 2 line_dbl + mul_by_014; 3 line_dbl + conj12): %d of 64 lanes differ from the host\n", w, wrong);
        bad += wrong;
    }
    for (int rounds : {1, 2, 3, 8, 63}) {
        hipLaunchKernelGGL(k_miller, dim3(1), dim3(64), 0, 0, d, rounds);
        std::vector<uint64_t> h(64);
        if (hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        int wrong = 0;
        for (uint32_t l = 0; l < 64; l++) wrong += sum(miller_rounds(l, rounds)) != h[l];
        printf("miller loop, %d rounds: %d of 64 lanes differ from the host\n", rounds, wrong);
        bad += wrong;
    }
    return bad ? 1 : 0;
}
