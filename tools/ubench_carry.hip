// Micro-benchmark: what the carry instructions of the Montgomery reduction cost on gfx950, next to the multiply-add they sit between.
// A reduction row ends in  c[i+1] += c[i] >> 28  (v_lshrrev_b64 + v_lshl_add_u64 as the compiler emits it) and the final limbs in
// v_lshl_add_u64 + v_and_b32 + v_lshrrev_b64: 726 such instructions per mixed addition next to 3542 multiply-adds.  Eight independent
// registers per lane, 64-thread workgroups, cycles per wave-instruction and SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_carry.hip -o tools/ubench_carry
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
template <int V>
__global__ void __launch_bounds__(64) k_bench(uint64_t* out, int iters, uint32_t seed) {
    uint64_t r[8];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = ((uint64_t)a << 31) * (i + 3) + i;
    uint64_t k = 0x0123456789ull + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t lo = (uint32_t)r[i], hi = (uint32_t)(r[i] >> 32);
                if constexpr (V == 0) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(r[i]));
                else if constexpr (V == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(k));
                else if constexpr (V == 2) { asm volatile("v_alignbit_b32 %0, %1, %0, 1" : "+v"(lo) : "v"(hi)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 3) { asm volatile("v_and_b32 %0, 0xfffffff, %0" : "+v"(lo)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 4) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a) : "vcc"); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 5) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(r[i]));
                else if constexpr (V == 6) { asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(lo)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 7) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(lo) : "vcc");
                else if constexpr (V == 8) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 9) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 10) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(lo) : "v"(a)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 11) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(a)); r[i] = ((uint64_t)hi << 32) | lo; }
                else if constexpr (V == 12) { asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(hi)); r[i] = ((uint64_t)hi << 32) | lo; }
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= r[i];
    if (s == 0x1234567ull) out[blockIdx.x * 64 + threadIdx.x] = s;
}
static const char* names[] = {"v_lshrrev_b64", "v_lshl_add_u64", "v_alignbit_b32", "v_and_b32", "v_add_co_u32 + v_addc_co_u32 (pair)", "v_ashrrev_i64",
                              "v_lshrrev_b32", "v_mad_u64_u32", "v_mul_lo_u32", "v_add_u32", "v_lshl_add_u32", "v_cndmask_b32", "v_mov_b32"};
template <int V>
int run(int w, int iters, uint64_t* d_out) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 4 * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<V>, dim3(grid), dim3(64), 0, 0, d_out, iters / 8, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<V>, dim3(grid), dim3(64), 0, 0, d_out, iters, 2u + r);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-38s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction group and SIMD @2.4 GHz\n", names[V], w, best, best * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    return 0;
}
int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 1024 * 8 * 64 * 8));
    for (int w : {1, 2}) {
        if (run<0>(w, 2000, d_out) || run<1>(w, 2000, d_out) || run<2>(w, 2000, d_out) || run<3>(w, 2000, d_out) || run<4>(w, 2000, d_out) || run<5>(w, 2000, d_out) ||
            run<6>(w, 2000, d_out) || run<7>(w, 2000, d_out) || run<8>(w, 2000, d_out) || run<9>(w, 2000, d_out) || run<10>(w, 2000, d_out) || run<11>(w, 2000, d_out) ||
            run<12>(w, 2000, d_out)) return 1;
    }
    return 0;
}
