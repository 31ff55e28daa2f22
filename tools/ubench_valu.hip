// Micro-benchmark: VALU integer / f64 issue rates on gfx950 (MI355X).
// Decides the limb width and multiply primitive of the Fp Montgomery kernel
// ("measure v_mad_u64_u32 first").
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int NACC = 8;     // independent chains per lane
constexpr int UNROLL = 8;   // asm groups per loop iteration

enum Op { MAD64 = 0, MULLO, MULHI, MAD24, MULHI24, ADD64, ADD32, FMA64, FMA32, ADD3, MAD64_ADDC, NOPS };
static const char* op_names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                                 "v_add_co+v_addc(64b add)", "v_add_u32", "v_fma_f64", "v_fma_f32", "v_add3_u32", "mad_u64_u32+addc"};
// lane-ops counted per asm group (1 for everything; ADD64 and MAD64_ADDC are 2 instructions)

template <int OP>
__global__ void __launch_bounds__(256) k_bench(uint64_t* out, int iters, uint32_t seed) {
    uint64_t acc[NACC];
    uint32_t c2[NACC];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u;
    uint32_t b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < NACC; i++) { acc[i] = (uint64_t)a * (i + 3) + b; c2[i] = i; }
    double fa = 1.0000001 + a * 1e-12, fb = 0.9999999 + b * 1e-12;
    float ga = 1.0001f, gb = 0.9999f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if constexpr (OP == MAD64) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (OP == MULLO) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[i] = lo;
                } else if constexpr (OP == MULHI) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[i] = lo | 0x80000001u;
                } else if constexpr (OP == MAD24) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
                    acc[i] = lo;
                } else if constexpr (OP == MULHI24) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[i] = lo | 0x00800001u;
                } else if constexpr (OP == ADD64) {
                    uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                    acc[i] = ((uint64_t)hi << 32) | lo;
                } else if constexpr (OP == ADD32) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[i] = lo;
                } else if constexpr (OP == FMA64) {
                    double d = __longlong_as_double((long long)acc[i]);
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(fa), "v"(fb));
                    acc[i] = (uint64_t)__double_as_longlong(d);
                } else if constexpr (OP == FMA32) {
                    float d = __uint_as_float((uint32_t)acc[i]);
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(ga), "v"(gb));
                    acc[i] = __float_as_uint(d);
                } else if constexpr (OP == ADD3) {
                    uint32_t lo = (uint32_t)acc[i];
                    asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
                    acc[i] = lo;
                } else if constexpr (OP == MAD64_ADDC) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(c2[i]) : "v"(a), "v"(b) : "vcc");
                }
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i] + c2[i];
    if (s == 0x1234567ull) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // practically never; keeps results live
}

template <int OP>
int run(int blocks_per_cu, int iters, uint64_t* d_out) {
    int ncu = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
    int grid = ncu * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters / 8, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 2u + r);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double groups = (double)grid * 256 * (double)iters * UNROLL * NACC;  // lane-level asm groups
    double rate = groups / (best * 1e-3);
    // cycles per wave-instruction-group per SIMD at nominal 2.4 GHz: SIMDs = ncu*4
    double wave_groups_per_s = rate / 64.0;
    double cyc = (double)ncu * 4 * 2.4e9 / wave_groups_per_s;
    printf("%-28s blocks/CU=%d waves/SIMD=%d  %.3f ms  %.2f T lane-groups/s  ~%.2f cyc/wave-group/SIMD @2.4GHz\n",
           op_names[OP], blocks_per_cu, blocks_per_cu, best, rate * 1e-12, cyc);
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 256ull * 8 * 256 * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int iters = 2000;
    for (int bpc : {1, 2, 4, 8}) {
        run<MAD64>(bpc, iters, d_out);
        run<MAD64_ADDC>(bpc, iters, d_out);
        run<MULLO>(bpc, iters, d_out);
        run<MULHI>(bpc, iters, d_out);
        run<MAD24>(bpc, iters, d_out);
        run<MULHI24>(bpc, iters, d_out);
        run<ADD64>(bpc, iters, d_out);
        run<ADD32>(bpc, iters, d_out);
        run<ADD3>(bpc, iters, d_out);
        run<FMA64>(bpc, iters, d_out);
        run<FMA32>(bpc, iters, d_out);
    }
    CK(hipFree(d_out));
    return 0;
}
