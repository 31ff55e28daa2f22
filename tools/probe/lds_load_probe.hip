// Probe: global_load_lds_dwordx4 on gfx950 — each lane fetches 16 B from ITS OWN global address straight into LDS at
// M0-base + lane * 16 (no VGPR staging).  Checks the layout assumption the accumulate kernel's point prefetch relies on.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out) {
    __shared__ uint32_t buf[256 * 32];   // [wave][k: 8][lane: 64][4 words]
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* wbase = buf + wave * (8 * 64 * 4);
    const uint32_t* p = src + (size_t)idx[blockIdx.x * 256 + threadIdx.x] * 32;
#pragma unroll
    for (int k = 0; k < 8; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 4 * k),
                                         (__attribute__((address_space(3))) void*)(wbase + k * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint4 v = *reinterpret_cast<const uint4*>(wbase + k * 256 + lane * 4);
        uint32_t* o = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * 32 + 4 * k;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
}
int main() {
    const int N = 1 << 16, T = 256 * 64;
    std::vector<uint32_t> src((size_t)N * 32), idx(T), out((size_t)T * 32);
    for (size_t i = 0; i < src.size(); i++) src[i] = (uint32_t)(i * 2654435761u);
    for (int i = 0; i < T; i++) idx[i] = (uint32_t)((i * 40503u + 7) % N);
    uint32_t *ds, *di, *dout;
    hipMalloc(&ds, src.size() * 4); hipMalloc(&di, idx.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(T / 256), dim3(256), 0, 0, ds, di, dout);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < T; i++)
        for (int w = 0; w < 32; w++)
            if (out[(size_t)i * 32 + w] != src[(size_t)idx[i] * 32 + w]) bad++;
    printf("global_load_lds_dwordx4 probe: %zu mismatching words of %zu\n", bad, out.size());
    return bad != 0;
}
