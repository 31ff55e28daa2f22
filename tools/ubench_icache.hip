// Micro-benchmark: throughput of a straight-line loop body of S bytes of v_mad_u64_u32 — finds the
// instruction-cache capacity that bounds how much of the bucket-add formula may be unrolled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NMAD>  // NMAD mads of 8 bytes each in the loop body
__global__ void __launch_bounds__(256) k_body(uint64_t* out, int iters, uint32_t seed) {
    uint64_t acc[8];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u, b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)a * (i + 3) + b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < NMAD / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i];
    if (s == 0x1234567ull) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NMAD>
int run(uint64_t* d_out, int bpc) {
    int grid = 256 * bpc;
    long total_mads_per_lane = 1 << 20;
    int iters = (int)(total_mads_per_lane / NMAD);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_body<NMAD>, dim3(grid), dim3(256), 0, 0, d_out, iters / 4 + 1, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_body<NMAD>, dim3(grid), dim3(256), 0, 0, d_out, iters, 2u + r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double mads = (double)grid * 256 * (double)iters * NMAD;
    double cyc = 1024.0 * 2.4e9 / (mads / 64.0 / (best * 1e-3));
    printf("body %6d B (%5d mads)  waves/SIMD=%d  %.3f ms  %.2f cyc/mad/SIMD\n", NMAD * 8, NMAD, bpc, best, cyc);
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 256ull * 8 * 256 * 8));
    for (int bpc : {1, 2}) {
        run<512>(d_out, bpc); run<1024>(d_out, bpc); run<2048>(d_out, bpc); run<3072>(d_out, bpc); run<3584>(d_out, bpc);
        run<4096>(d_out, bpc); run<5120>(d_out, bpc); run<6144>(d_out, bpc); run<7168>(d_out, bpc); run<8192>(d_out, bpc);
        run<10240>(d_out, bpc); run<12288>(d_out, bpc); run<16384>(d_out, bpc);
    }
    return 0;
}
