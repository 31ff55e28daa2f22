// Micro-benchmark (round 6): which workgroups find room on a compute unit BESIDE a register-heavy kernel that already holds two
// waves per SIMD?  The window-group pipeline of run_msm (csrc/msm_curve.hpp) runs the sort / schedule / reduce kernels of one group
// under the accumulate kernel of another (k_accumulate<G1C>: 215 VGPRs, one-wave workgroups, no LDS), and its gain depends on those
// workgroups being dispatched at once instead of waiting for accumulate waves to retire.
//   hog     one-wave workgroups, 216 registers (v215 is written), a long v_mad_u64_u32 loop: the stand-in for the accumulate kernel;
//           the grid is four times the machine's wave slots so that retiring waves are replaced from the hog's own queue
//   probe   workgroups of NT lanes with V registers and L bytes of LDS; lane 0 stamps the 100-MHz clock on entry, all lanes spin ~20 us
// Reported per probe shape: kernel time beside the hog and alone, first / median / last workgroup start relative to the first.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_coresidency tools/ubench_coresidency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64, 2) hog(uint64_t* out, int iters, uint32_t seed) {
    uint64_t acc[8];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u, b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)a * (i + 3) + b;
    asm volatile("v_mov_b32 v215, 0" ::: "v215");   // 216 registers, as k_accumulate<G1C>
    for (int it = 0; it < iters; it++) {
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

// hog with TURNOVER: waves of different lengths (60 .. 240 us) from a long queue, as the accumulate kernel's items of different lengths
__global__ void __launch_bounds__(64, 2) hog_turnover(uint64_t* out, int iters, uint32_t seed) {
    uint64_t acc[8];
    uint32_t a = seed * 2654435761u + threadIdx.x * 40503u + 1u, b = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)a * (i + 3) + b;
    asm volatile("v_mov_b32 v215, 0" ::: "v215");
    const int mine = iters * (1 + (int)((blockIdx.x * 2654435761u >> 28) & 3u));
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

#define PROBE(V)                                                                                                             \
    template <int PRIO> __global__ void probe_##V(unsigned long long* stamps, uint32_t lds_words, int spin) {               \
        extern __shared__ uint32_t lds[];                                                                                    \
        if (PRIO) __builtin_amdgcn_s_setprio(3);                                                                             \
        if (threadIdx.x == 0) stamps[blockIdx.x] = wall_clock64();                                                           \
        asm volatile("v_mov_b32 v" #V ", 0" ::: "v" #V);                                                                     \
        uint32_t x = threadIdx.x;                                                                                            \
        if (lds_words) lds[threadIdx.x % lds_words] = x;                                                                     \
        unsigned long long t0 = wall_clock64();                                                                              \
        while (wall_clock64() - t0 < (unsigned long long)spin) x = x * 1664525u + 1013904223u;                               \
        if (x == 0x12345u) stamps[blockIdx.x] = x;                                                                           \
    }
PROBE(23) PROBE(31) PROBE(39) PROBE(47) PROBE(55) PROBE(63) PROBE(79) PROBE(111) PROBE(183)

template <class K>
int run(K kern, const char* name, int vg, int nt, size_t lds, int nwg, hipStream_t hs, hipStream_t ps, uint64_t* d_out, unsigned long long* d_st,
        int hog_waves, bool turnover = false, int spin_ticks = 2000) {
    std::vector<unsigned long long> st(nwg);
    for (int beside = 1; beside >= 0; beside--) {
        CK(hipMemset(d_st, 0, nwg * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        if (beside) {
            if (turnover) hipLaunchKernelGGL(hog_turnover, dim3(hog_waves), dim3(64), 0, hs, d_out, 60, 7u);   // 60 .. 240 us per wave
            else hipLaunchKernelGGL(hog, dim3(hog_waves), dim3(64), 0, hs, d_out, 1200, 7u);   // ~1.3 ms per wave
            // let the hog fill the machine before the probe arrives
            hipEvent_t eh; CK(hipEventCreate(&eh)); CK(hipEventRecord(eh, hs));
            for (volatile int spin = 0; spin < 400000; spin++) {}
        }
        CK(hipEventRecord(e0, ps));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(nt), lds, ps, d_st, (uint32_t)(lds / 4), spin_ticks);   // 20 us of spinning per workgroup by default
        CK(hipEventRecord(e1, ps));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), d_st, nwg * 8, hipMemcpyDeviceToHost));
        std::sort(st.begin(), st.end());
        printf("%-9s V=%3d NT=%4d LDS=%6zu wgs=%4d %s: kernel %8.1f us   starts: median +%7.1f us  p90 +%7.1f  last +%7.1f us\n", name, vg, nt, lds, nwg,
               beside ? "beside hog" : "alone     ", ms * 1e3, (double)(st[nwg / 2] - st[0]) * 0.01, (double)(st[nwg * 9 / 10] - st[0]) * 0.01,
               (double)(st[nwg - 1] - st[0]) * 0.01);
    }
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 1 << 24));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 1 << 20));
    hipStream_t hs, ps, ps_hi;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&ps_hi, hipStreamNonBlocking, hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipFuncSetAttribute((const void*)probe_55<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe_23<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe_39<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int HOGQ = 8192;   // four generations of hog waves: retiring waves are replaced from the hog's own queue
    // (a) registers: 256-lane workgroups (one wave per SIMD), no LDS, high-priority stream
    run(probe_23<1>, "prio-hi", 24, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_31<1>, "prio-hi", 32, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_39<1>, "prio-hi", 40, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_47<1>, "prio-hi", 48, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_55<1>, "prio-hi", 56, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_63<1>, "prio-hi", 64, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_79<1>, "prio-hi", 80, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_111<1>, "prio-hi", 112, 256, 0, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_183<1>, "prio-hi", 184, 64, 0, 2048, hs, ps_hi, d_out, d_st, HOGQ);   // the reduce kernel's shape
    // (b) the same on a normal-priority stream
    run(probe_39<1>, "prio-norm", 40, 256, 0, 512, hs, ps, d_out, d_st, HOGQ);
    run(probe_55<1>, "prio-norm", 56, 256, 0, 512, hs, ps, d_out, d_st, HOGQ);
    run(probe_183<1>, "prio-norm", 184, 64, 0, 2048, hs, ps, d_out, d_st, HOGQ);
    // (c) LDS
    run(probe_23<1>, "lds", 24, 256, 36 * 1024, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_23<1>, "lds", 24, 256, 72 * 1024, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_55<1>, "lds", 56, 256, 72 * 1024, 512, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_39<1>, "lds", 40, 256, 72 * 1024, 512, hs, ps_hi, d_out, d_st, HOGQ);
    // (d) workgroup size: two and four waves per SIMD
    run(probe_23<1>, "wg-size", 24, 512, 0, 256, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_23<1>, "wg-size", 24, 1024, 0, 128, hs, ps_hi, d_out, d_st, HOGQ);
    run(probe_39<1>, "wg-size", 40, 512, 0, 256, hs, ps_hi, d_out, d_st, HOGQ);
    // (e) a hog that exactly fills the machine (no queue behind it): slots freed by nobody until the hog ends
    run(probe_55<1>, "hog-1gen", 56, 256, 0, 512, hs, ps_hi, d_out, d_st, 2048);
    run(probe_183<1>, "hog-1gen", 184, 64, 0, 2048, hs, ps_hi, d_out, d_st, 2048);
    // (f) a hog with turnover (waves of 60 .. 240 us from a queue of 16 generations): do freed accumulate slots fragment the register file?
    run(probe_23<1>, "turnover", 24, 256, 0, 512, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_39<1>, "turnover", 40, 256, 72 * 1024, 512, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_55<1>, "turnover", 56, 256, 72 * 1024, 512, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_55<1>, "turnover", 56, 256, 72 * 1024, 256, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_79<1>, "turnover", 80, 256, 0, 256, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_183<1>, "turnover", 184, 64, 0, 2048, hs, ps_hi, d_out, d_st, 32768, true, 6000);
    run(probe_183<1>, "turn-norm", 184, 64, 0, 2048, hs, ps, d_out, d_st, 32768, true, 6000);
    return 0;
}
