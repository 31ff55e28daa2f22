// Micro-benchmark (round 6): does a register-heavy kernel lose throughput when its work is cut into TWO launches that run concurrently on
// two streams (hardware queues)?  The window-group pipeline (csrc/msm_curve.hpp) runs the accumulate kernels of consecutive groups that way.
//   one launch of 16384 one-wave workgroups  vs  two launches of 8192 on two streams  vs  the same two launches back to back on one stream
// waves of four different lengths (as the accumulate kernel's items); 216 registers per lane, two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(64, 2) hog(uint64_t* out, int iters, uint32_t seed, int sorted) {
    uint64_t acc[8];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u, b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)a * (i + 3) + b;
    asm volatile("v_mov_b32 v215, 0" ::: "v215");
    // sorted: longest waves first (the schedule's order); otherwise lengths in pseudo-random order
    const int cls = sorted ? 3 - (int)((4ull * blockIdx.x) / gridDim.x) : (int)((blockIdx.x * 2654435761u >> 28) & 3u);
    const int mine = iters * (1 + cls);
    for (int it = 0; it < mine; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i];
    if (s == 0x1234567ull) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 1 << 24));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int N = 16384, IT = 40;
    for (int sorted = 0; sorted < 2; sorted++)
        for (int rep = 0; rep < 3; rep++) {
            float a, b, c;
            CK(hipEventRecord(e0, s0));
            hipLaunchKernelGGL(hog, dim3(N), dim3(64), 0, s0, d_out, IT, 7u, sorted);
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&a, e0, e1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            hipLaunchKernelGGL(hog, dim3(N / 2), dim3(64), 0, s0, d_out, IT, 7u, sorted);
            hipLaunchKernelGGL(hog, dim3(N / 2), dim3(64), 0, s0, d_out, IT, 8u, sorted);
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&b, e0, e1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            CK(hipStreamWaitEvent(s1, e0, 0));
            hipLaunchKernelGGL(hog, dim3(N / 2), dim3(64), 0, s0, d_out, IT, 7u, sorted);
            hipLaunchKernelGGL(hog, dim3(N / 2), dim3(64), 0, s1, d_out, IT, 8u, sorted);
            CK(hipEventRecord(e2, s1)); CK(hipStreamWaitEvent(s0, e2, 0));
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&c, e0, e1));
            CK(hipDeviceSynchronize());
            printf("%s lengths: one launch %.3f ms | two launches, one stream %.3f ms | two launches, two streams %.3f ms\n", sorted ? "sorted  " : "shuffled", a, b, c);
        }
    return 0;
}
