// Micro-benchmark: does the VGPR bank of the operands change the issue rate of v_mad_u64_u32 on gfx950?
// The multiply-add reads four dwords (src0, src1, the 64-bit addend) and measures 4.7-4.9 cycles per wave-instruction where a
// 4-cycle instruction would be the design rate; with the multiplier in an SGPR it measures 4.7 (tools/ubench_mad_sdst.hip).  If
// register-bank conflicts (bank = VGPR number mod 4) were the difference, the placement of a and b relative to the accumulator
// pair would show.  Everything is hand-placed: accumulators v[16:17], v[20:21], .. v[44:45] (banks 0/1), eight chains.
//   0  a = v2 (bank 2), b = v3 (bank 3)     no two operands of an instruction share a bank
//   1  a = v2 (bank 2), b = v6 (bank 2)     a and b share a bank
//   2  a = v4 (bank 0), b = v8 (bank 0)     a, b and the accumulator's low half share a bank
//   3  a = v1 (bank 1), b = v3 (bank 3)     a shares the bank of the accumulator's high half
//   4  a = v2, b = s40                      multiplier in an SGPR: three VGPR dwords read
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mad_banks.hip -o tools/ubench_mad_banks
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define MAD8(A, B)                                           \
    "v_mad_u64_u32 v[16:17], s[42:43], " A ", " B ", v[16:17]\n\t" \
    "v_mad_u64_u32 v[20:21], s[42:43], " A ", " B ", v[20:21]\n\t" \
    "v_mad_u64_u32 v[24:25], s[42:43], " A ", " B ", v[24:25]\n\t" \
    "v_mad_u64_u32 v[28:29], s[42:43], " A ", " B ", v[28:29]\n\t" \
    "v_mad_u64_u32 v[32:33], s[42:43], " A ", " B ", v[32:33]\n\t" \
    "v_mad_u64_u32 v[36:37], s[42:43], " A ", " B ", v[36:37]\n\t" \
    "v_mad_u64_u32 v[40:41], s[42:43], " A ", " B ", v[40:41]\n\t" \
    "v_mad_u64_u32 v[44:45], s[42:43], " A ", " B ", v[44:45]\n\t"
#define CLOB "v1", "v2", "v3", "v4", "v6", "v8", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45", "s40", "s42", "s43"

template <int V>
__global__ void __launch_bounds__(64) k_bench(uint32_t* out, int iters, uint32_t seed) {
    const uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + 1u, y = (seed ^ 0x9e3779b9u) + blockIdx.x * 7919u + 3u;
    asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v6, %1\n\tv_mov_b32 v8, %1\n\t"
                 "s_mov_b32 s40, 0x0fffaaab\n\t"
                 "v_mov_b32 v16, %0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v20, %1\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, 0\n\t"
                 "v_mov_b32 v28, %1\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v36, %1\n\tv_mov_b32 v37, 0\n\t"
                 "v_mov_b32 v40, %0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v45, 0"
                 : : "v"(x), "v"(y) : CLOB);
    for (int it = 0; it < iters; it++) {
        if constexpr (V == 0) asm volatile(MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") MAD8("v2", "v3") : : : CLOB);
        else if constexpr (V == 1) asm volatile(MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") MAD8("v2", "v6") : : : CLOB);
        else if constexpr (V == 2) asm volatile(MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") MAD8("v4", "v8") : : : CLOB);
        else if constexpr (V == 3) asm volatile(MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") MAD8("v1", "v3") : : : CLOB);
        else asm volatile(MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") MAD8("v2", "s40") : : : CLOB);
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v16, v20\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v28\n\tv_xor_b32 %0, %0, v32\n\tv_xor_b32 %0, %0, v36\n\t"
                 "v_xor_b32 %0, %0, v40\n\tv_xor_b32 %0, %0, v44" : "=v"(r) : : CLOB);
    if (r == 0x12345u) out[blockIdx.x * 64 + threadIdx.x] = r;
}

static const char* names[] = {"a bank 2, b bank 3 (no shared bank)", "a, b both bank 2", "a, b, acc.lo all bank 0", "a bank 1 = acc.hi, b bank 3", "b in an SGPR"};

template <int V>
int run(int waves_per_simd, int iters, uint32_t* d_out) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4, grid = simds * waves_per_simd;
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
    const double instr_per_wave = (double)iters * 64;
    printf("%-40s waves/SIMD=%d  %.3f ms  %.2f cycles per v_mad_u64_u32 and SIMD @2.4 GHz\n", names[V], waves_per_simd, best,
           best * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
    return 0;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 1024 * 8 * 64 * 4));
    const int iters = 4000;
    for (int w : {1, 2, 4}) {
        if (run<0>(w, iters, d_out) || run<1>(w, iters, d_out) || run<2>(w, iters, d_out) || run<3>(w, iters, d_out) || run<4>(w, iters, d_out)) return 1;
    }
    CK(hipFree(d_out));
    return 0;
}
