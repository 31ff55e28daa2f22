// Micro-benchmark: is there a cheaper 381-bit Montgomery multiplication on gfx950 than the 14 x 28-bit v_mad_u64_u32 form?
// Candidate: 8 limbs of 52 bits held as doubles, products split into high and low halves by fused multiply-adds in round-toward-
// zero mode (the "DPFP" scheme of Emmart, Zheng, Weems, "Faster modular exponentiation using double precision floating point
// arithmetic on the GPU", ARITH 2018): v_fma_f64 issues at the rate of v_mad_u64_u32 (profiles/r01_ubench_valu.txt), and an
// 8 x 8 product needs 64 limb products where the 28-bit form needs 196.
//
// What a limb product costs in that scheme (a, b < 2^52 integers in doubles; C1 = 2^104, C2 = 2^104 + 2^52; MODE.fp_round(f64) = RZ):
//     hi  = fma(a, b, C1)        = 2^104 + floor(ab / 2^52) 2^52      bit pattern 0x467.. | H        v_fma_f64
//     sub = C2 - hi              = 2^52 - H 2^52  (exact)                                             v_add_f64
//     lo  = fma(a, b, sub)       = 2^52 + (ab mod 2^52)  (exact)       bit pattern 0x433.. | L        v_fma_f64
//     col[i+j+1] += bits(hi);  col[i+j] += bits(lo)                    64-bit integer adds            2 x v_lshl_add_u64
// i.e. FIVE instructions per limb product (the exponent patterns are taken out of the columns once, as constants), against ONE
// v_mad_u64_u32 whose 64-bit accumulate is free.  The count says 64 x 5 = 320 against 196 for the product and the same again for
// the reduction; this file measures it.  fp52_mul below is a complete Montgomery multiplication (R = 2^416), checked against
// big-integer arithmetic by tools/check_fp52.py on the values the benchmark leaves behind.
//
// Build: hipcc -O3 --offload-arch=gfx950 -I ark-blst_amd/csrc tools/ubench_fp52.hip -o tools/ubench_fp52
// Run:   tools/ubench_fp52 gpurun_out/fp52_vectors.txt && python tools/check_fp52.py gpurun_out/fp52_vectors.txt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "fp28.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

namespace fp52 {
constexpr int NL = 8;
constexpr uint64_t MASK = (1ull << 52) - 1;
constexpr uint64_t P[NL] = {0xeffffffffaaabull, 0xfeb153ffffb9full, 0x6b0f6241eabffull, 0x12bf6730d2a0full,
                            0x764774b84f385ull, 0x1ba7b6434bacdull, 0x1ea397fe69a4bull, 0x1a011ull};
constexpr uint64_t PINV = 0x3fffcfffcfffdull;   // -p^-1 mod 2^52
constexpr uint64_t B_HI = 0x4670000000000000ull, B_LO = 0x4330000000000000ull;   // bit patterns of 2^104 and 2^52
// number of (i, j) in [0, 8)^2 with i + j = k
__host__ __device__ constexpr int cnt(int k) { return k < 0 || k > 14 ? 0 : (k < 14 - k ? k : 14 - k) + 1; }

struct F52 { double l[NL]; };   // integers < 2^52

__device__ __forceinline__ double u2d(uint64_t v) { return __longlong_as_double((long long)(v | B_LO)) - 4503599627370496.0; }   // v < 2^52, exact

// one limb product into the columns: the three floating-point instructions as assembly (the first one is rounding-sensitive: the
// kernel has switched MODE.fp_round for f64 to round-toward-zero, which the compiler does not know)
__device__ __forceinline__ void limb_product(uint64_t& c_lo, uint64_t& c_hi, double a, double b) {
    const double C1 = 20282409603651670423947251286016.0;                 // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52
    double hi, sub, lo;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(a), "v"(b), "v"(C1));
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(sub) : "v"(C2), "v"(hi));
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(a), "v"(b), "v"(sub));
    c_hi += (uint64_t)__double_as_longlong(hi);
    c_lo += (uint64_t)__double_as_longlong(lo);
}

// r = a b / 2^416 mod p, a, b < 2^416 with a b < 2^416 p; result < 2p, limbs exact
__device__ __forceinline__ F52 fp52_mul(const F52& a, const F52& b) {
    uint64_t c[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) c[k] = 0ull - (2ull * cnt(k) * B_LO + 2ull * cnt(k - 1) * B_HI);   // exponent patterns of everything column k will receive
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) limb_product(c[i + j], c[i + j + 1], a.l[i], b.l[j]);
    }
    double pd[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) pd[j] = (double)P[j];
    const double pinv = (double)PINV;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        // m = (c[i] mod 2^52) * PINV mod 2^52: the low half of one more limb product (the patterns have zero low bits)
        const double q = u2d(c[i] & MASK);
        uint64_t mlo = 0, mhi = 0;
        limb_product(mlo, mhi, q, pinv);
        const double m = u2d(mlo & MASK);
#pragma unroll
        for (int j = 0; j < NL; j++) limb_product(c[i + j], c[i + j + 1], m, pd[j]);
        c[i + 1] += c[i] >> 52;   // the low 52 bits of c[i] are zero now, every pattern it was due has arrived
    }
    F52 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint64_t v = c[NL + k] + carry;
        r.l[k] = u2d(v & MASK);
        carry = v >> 52;
    }
    return r;
}
}  // namespace fp52

// x <- x * y, `iters` times, per lane: fp52 (V = 0) or fp28::fp_mul (V = 1, the shipped product-scanning form; V = 2 operand scanning)
template <int V>
__global__ void __launch_bounds__(64) k_chain(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if constexpr (V == 0) {
        // MODE[3:2] (f64 / f16 rounding) = 3: round toward zero.  As volatile assembly: the builtin form is free to move, and the
        // compiler hoisted the restoring write above the loop (the products then ran in round-to-nearest and failed the check)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_nop 3");
        fp52::F52 x, y;
#pragma unroll
        for (int k = 0; k < 8; k++) { x.l[k] = (double)in[(size_t)t * 16 + k]; y.l[k] = (double)in[(size_t)t * 16 + 8 + k]; }
#pragma unroll 1
        for (int it = 0; it < iters; it++) x = fp52::fp52_mul(x, y);
#pragma unroll
        for (int k = 0; k < 8; k++) out[(size_t)t * 8 + k] = (uint64_t)x.l[k];
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 3");
    } else {
        fp28::Fp x, y;
#pragma unroll
        for (int k = 0; k < 14; k++) { x.l[k] = (uint32_t)in[(size_t)t * 28 + k]; y.l[k] = (uint32_t)in[(size_t)t * 28 + 14 + k]; }
#pragma unroll 1
        for (int it = 0; it < iters; it++) x = V == 1 ? fp28::fp_mul(x, y) : fp28::fp_mul_os(x, y);
#pragma unroll
        for (int k = 0; k < 14; k++) out[(size_t)t * 14 + k] = x.l[k];
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <int V>
int run(const char* name, int waves_per_simd, int iters, FILE* vec) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4, grid = simds * waves_per_simd, lanes = grid * 64;
    const int nin = V == 0 ? 16 : 28, nout = V == 0 ? 8 : 14, bits = V == 0 ? 52 : 28;
    std::vector<uint64_t> h_in((size_t)lanes * nin), h_out((size_t)lanes * nout);
    for (size_t i = 0; i < h_in.size(); i++) {
        const size_t k = i % (nin / 2);
        uint64_t v = rnd() & ((1ull << bits) - 1);
        if (k == (size_t)nin / 2 - 1) v &= V == 0 ? 0xffffull : 0x1ffffull;   // top limb: value < 2^380 < p
        h_in[i] = v;
    }
    uint64_t *d_in, *d_out;
    CK(hipMalloc(&d_in, h_in.size() * 8)); CK(hipMalloc(&d_out, h_out.size() * 8));
    CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<V>, dim3(grid), dim3(64), 0, 0, d_in, d_out, 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain<V>, dim3(grid), dim3(64), 0, 0, d_in, d_out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost));
    const double wave_muls_per_simd = (double)waves_per_simd * iters;
    const double cyc = best * 1e-3 * 2.4e9 / wave_muls_per_simd;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.0f cycles per multiplication (per wave and SIMD, 2.4 GHz)  %.2f T mul-lanes/s\n", name, waves_per_simd,
           best, cyc, (double)lanes * iters / (best * 1e-3) / 1e12);
    if (vec && waves_per_simd == 2) {   // 64 lanes spread over the launch: inputs, iterations, output
        for (int s = 0; s < 64; s++) {
            const size_t t = (size_t)s * (lanes / 64) + s;
            fprintf(vec, "%d %d", bits, iters);
            for (int k = 0; k < nin; k++) fprintf(vec, " %llx", (unsigned long long)h_in[t * nin + k]);
            for (int k = 0; k < nout; k++) fprintf(vec, " %llx", (unsigned long long)h_out[t * nout + k]);
            fprintf(vec, "\n");
        }
    }
    CK(hipFree(d_in)); CK(hipFree(d_out));
    return 0;
}

int main(int argc, char** argv) {
    FILE* vec = argc > 1 ? fopen(argv[1], "w") : nullptr;
    const int iters = 512;
    for (int w : {1, 2, 4}) {
        if (run<0>("fp52 (8 x 52-bit, v_fma_f64 DPFP)", w, iters, vec)) return 1;
        if (run<1>("fp28 fp_mul (product scanning)", w, iters, vec)) return 1;
        if (run<2>("fp28 fp_mul_os (operand scanning)", w, iters, vec)) return 1;
    }
    if (vec) fclose(vec);
    return 0;
}
