// What clock64() and wall_clock64() count on gfx950, and the shader clock under a sustained integer multiply-add load.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/clock_probe tools/probe/clock_probe.hip && tools/probe/clock_probe
// The kernel fills every SIMD (1024 workgroups of 64 lanes x 2) with a dependent v_mad_u64_u32 chain of known length and
// reads s_memtime (clock64) and s_memrealtime (wall_clock64) around it; the host times the same launch with HIP events.
//   realtime ticks / event time  -> the constant counter's frequency (100 MHz expected)
//   memtime ticks / event time   -> what clock64 counts
//   MADs per lane / event time   -> issue rate; with 4 cycles per wave64 multiply-add and two waves per SIMD: shader clock >= 8 * MADs / time
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void __launch_bounds__(64, 2) k_clock(uint64_t* out, uint32_t iters, uint32_t seed) {
    uint64_t t0 = clock64(), w0 = wall_clock64();
    uint64_t acc = seed + threadIdx.x;
    uint32_t a = seed * 2654435761u + threadIdx.x, b = seed ^ 0x9e3779b9u;
#pragma unroll 1
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) acc = (uint64_t)(a + k) * (b ^ (uint32_t)acc) + acc;   // dependent chain of 32 multiply-adds
    }
    uint64_t t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = t1 - t0;
        out[3 * blockIdx.x + 1] = w1 - w0;
        out[3 * blockIdx.x + 2] = acc;
    }
}

int main() {
    const int blocks = 2048;
    uint64_t* d;
    hipMalloc(&d, blocks * 3 * sizeof(uint64_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t iters : {2000u, 20000u, 100000u}) {
        hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(64), 0, 0, d, iters, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(64), 0, 0, d, iters, 777u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks * 3);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double st = 0, sw = 0;
        for (int i = 0; i < blocks; i++) { st += h[3 * i]; sw += h[3 * i + 1]; }
        st /= blocks; sw /= blocks;
        double mads = 32.0 * iters;   // per lane
        printf("iters %u: event %.3f ms | clock64 %.0f ticks = %.1f MHz | wall_clock64 %.0f ticks = %.1f MHz | %.0f MAD per lane: "
               "%.2f ns per wave-MAD per SIMD-slot, shader clock >= %.0f MHz if a wave64 multiply-add issues in 4 cycles\n",
               iters, ms, st, st / (ms * 1e3), sw, sw / (ms * 1e3), mads, ms * 1e6 / (mads * 2), 2 * mads * 4 / (ms * 1e3));
    }
    return 0;
}
