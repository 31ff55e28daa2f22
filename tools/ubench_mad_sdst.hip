// Micro-benchmark: does the carry-out SGPR pair of v_mad_u64_u32 (VOP3B always writes one) serialise a wave's multiply-adds?
// tools/ubench_valu.hip measured 9.9 cycles per v_mad_u64_u32 for a lone wave per SIMD (4.9 with two) while v_fma_f64 /
// v_mul_lo_u32, which write no SGPR, reach 5.7 alone.  Every MAD there (and every compiler-generated one) names the SAME carry-out
// pair.  Variants:
//   0  sdst = vcc for every instruction                (the r01 benchmark)
//   1  sdst = s[40:41] for every instruction           (what the compiler does: one dead pair)
//   2  sdst rotating over 4 pairs  s[40:41] .. s[46:47]
//   3  sdst rotating over 8 pairs
//   4  ONE dependent chain, sdst rotating over 8 pairs (latency of the multiply-add itself)
//   5  ONE dependent chain, sdst = s[40:41]
//   6  v_mul_lo_u32 + v_mul_hi_u32 pairs (no SGPR write) for reference
//   7  8 chains, constant multiplier in an SGPR, sdst rotating over 8 pairs
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mad_sdst.hip -o tools/ubench_mad_sdst
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int NACC = 8, UNROLL = 8;

#define MAD(SD, ACC) asm volatile("v_mad_u64_u32 %0, " SD ", %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc")
#define MADK(SD, ACC) asm volatile("v_mad_u64_u32 %0, " SD ", %1, %2, %0" : "+v"(ACC) : "v"(a), "s"(kc) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc")

template <int V>
__global__ void __launch_bounds__(256) k_bench(uint64_t* out, int iters, uint32_t seed, uint32_t kc) {
    uint64_t acc[NACC];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u;
    uint32_t b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (uint64_t)a * (i + 3) + b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if constexpr (V == 0) {
                MAD("vcc", acc[0]); MAD("vcc", acc[1]); MAD("vcc", acc[2]); MAD("vcc", acc[3]);
                MAD("vcc", acc[4]); MAD("vcc", acc[5]); MAD("vcc", acc[6]); MAD("vcc", acc[7]);
            } else if constexpr (V == 1) {
                MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[1]); MAD("s[40:41]", acc[2]); MAD("s[40:41]", acc[3]);
                MAD("s[40:41]", acc[4]); MAD("s[40:41]", acc[5]); MAD("s[40:41]", acc[6]); MAD("s[40:41]", acc[7]);
            } else if constexpr (V == 2) {
                MAD("s[40:41]", acc[0]); MAD("s[42:43]", acc[1]); MAD("s[44:45]", acc[2]); MAD("s[46:47]", acc[3]);
                MAD("s[40:41]", acc[4]); MAD("s[42:43]", acc[5]); MAD("s[44:45]", acc[6]); MAD("s[46:47]", acc[7]);
            } else if constexpr (V == 3) {
                MAD("s[40:41]", acc[0]); MAD("s[42:43]", acc[1]); MAD("s[44:45]", acc[2]); MAD("s[46:47]", acc[3]);
                MAD("s[48:49]", acc[4]); MAD("s[50:51]", acc[5]); MAD("s[52:53]", acc[6]); MAD("s[54:55]", acc[7]);
            } else if constexpr (V == 4) {
                MAD("s[40:41]", acc[0]); MAD("s[42:43]", acc[0]); MAD("s[44:45]", acc[0]); MAD("s[46:47]", acc[0]);
                MAD("s[48:49]", acc[0]); MAD("s[50:51]", acc[0]); MAD("s[52:53]", acc[0]); MAD("s[54:55]", acc[0]);
            } else if constexpr (V == 5) {
                MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]);
                MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]); MAD("s[40:41]", acc[0]);
            } else if constexpr (V == 6) {
#pragma unroll
                for (int i = 0; i < NACC; i += 2) {
                    uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)acc[i + 1];
                    asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_mul_hi_u32 %1, %1, %2" : "+v"(lo), "+v"(hi) : "v"(a));
                    acc[i] = lo | 1u; acc[i + 1] = hi | 0x80000001u;
                }
            } else {
                MADK("s[40:41]", acc[0]); MADK("s[42:43]", acc[1]); MADK("s[44:45]", acc[2]); MADK("s[46:47]", acc[3]);
                MADK("s[48:49]", acc[4]); MADK("s[50:51]", acc[5]); MADK("s[52:53]", acc[6]); MADK("s[54:55]", acc[7]);
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s == 0x1234567ull) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const char* names[] = {"8 chains, sdst=vcc", "8 chains, sdst=s[40:41]", "8 chains, sdst over 4 pairs", "8 chains, sdst over 8 pairs",
                              "1 dependent chain, sdst over 8 pairs", "1 dependent chain, sdst=s[40:41]", "mul_lo + mul_hi (no sdst)",
                              "8 chains, SGPR multiplier, sdst over 8 pairs"};

template <int V>
int run(int blocks_per_cu, int iters, uint64_t* d_out) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, grid = ncu * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<V>, dim3(grid), dim3(256), 0, 0, d_out, iters / 8, 1u, 0x0fffaaabu);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<V>, dim3(grid), dim3(256), 0, 0, d_out, iters, 2u + r, 0x0fffaaabu);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr = (double)grid * 4 * (double)iters * UNROLL * NACC;   // wave-instructions
    const double cyc = best * 1e-3 * 2.4e9 * ncu * 4 / instr;
    printf("%-46s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction and SIMD @2.4 GHz\n", names[V], blocks_per_cu, best, cyc);
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 256ull * 8 * 256 * 8));
    const int iters = 2000;
    for (int bpc : {1, 2, 4}) {
        run<0>(bpc, iters, d_out); run<1>(bpc, iters, d_out); run<2>(bpc, iters, d_out); run<3>(bpc, iters, d_out);
        run<4>(bpc, iters, d_out); run<5>(bpc, iters, d_out); run<6>(bpc, iters, d_out); run<7>(bpc, iters, d_out);
    }
    CK(hipFree(d_out));
    return 0;
}
