// multi_gpu_msm.cpp — the one-process-per-GPU deployment of the MSM backend in plain C++ against the two C ABIs
// (include/arkblst_amd.h, include/arkblst_amd_rccl.h): no Python, no torch.  BASELINE config #3's shape: the base set sharded
// contiguously over the ranks, every rank runs its shard, mi_msm_g1_allgather_fold combines the window sums with ncclAllGather over
// xGMI and leaves the same point on every rank.  The reference has no counterpart (/root/reference/src/gpu.rs:233-239: device 0 only).
//
//   multi_gpu_msm <rank> <n_ranks> <id_file> <bases.bin> <scalars.bin> <out.bin> [steps]
//
// Start it n_ranks times (rank r uses HIP device r mod the visible devices).  Rank 0 writes the 128-byte ncclUniqueId to <id_file>.tmp
// and renames it to <id_file>; the other ranks wait for the file — the side channel is the host program's business, any will do.
// <bases.bin>: N x 96-byte blst_p1_affine; <scalars.bin>: N x 32-byte canonical little-endian integers; rank r takes points
// [r N / n_ranks, (r + 1) N / n_ranks).  Every rank writes the 144-byte result to <out.bin>.<rank> and prints one line with its timings.
//
// Build:  g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/multi_gpu_msm.cpp -Lark-blst_amd/lib -larkblst_amd_rccl \
//             -larkblst_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/ark-blst_amd/lib -Wl,-rpath,/opt/rocm/lib -o multi_gpu_msm
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/arkblst_amd_rccl.h"

static std::vector<char> read_file(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define DIE(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } while (0)

int main(int argc, char** argv) {
    if (argc < 7) DIE("usage: %s <rank> <n_ranks> <id_file> <bases.bin> <scalars.bin> <out.bin> [steps]", argv[0]);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   // dmabuf IPC (hosts whose driver has no legacy IPC): before the runtime loads anything
    const int rank = std::atoi(argv[1]), n_ranks = std::atoi(argv[2]), steps = argc > 7 ? std::atoi(argv[7]) : 3;
    const std::string id_file = argv[3];
    if (rank < 0 || n_ranks < 1 || rank >= n_ranks) DIE("bad rank");
    std::vector<char> bases = read_file(argv[4]), scalars = read_file(argv[5]);
    const size_t N = bases.size() / 96;
    if (N == 0 || scalars.size() != N * 32) DIE("inputs: %zu bases, %zu scalar bytes", N, scalars.size());
    const size_t lo = N * (size_t)rank / (size_t)n_ranks, hi = N * (size_t)(rank + 1) / (size_t)n_ranks, n = hi - lo;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) DIE("no HIP device");
    const int dev = rank % ndev;
    mi_ctx* ctx = nullptr;
    int rc = mi_msm_init(&ctx, &dev, 1);
    if (rc != MI_OK) DIE("mi_msm_init: %s", mi_msm_strerror(rc));
    rc = mi_msm_g1_set_bases(ctx, reinterpret_cast<const mi_g1_affine*>(bases.data()) + lo, n);
    if (rc != MI_OK) DIE("set_bases: %s", mi_msm_last_error(ctx));
    size_t invalid = 0;
    if (n) (void)mi_msm_g1_validate_bases(ctx, &invalid);   // an SRS is validated once when it is loaded
    void* d_scalars = nullptr;
    if (hipSetDevice(dev) != hipSuccess || hipMalloc(&d_scalars, n ? n * 32 : 32) != hipSuccess ||
        hipMemcpy(d_scalars, scalars.data() + lo * 32, n * 32, hipMemcpyHostToDevice) != hipSuccess)
        DIE("scalar upload failed");

    // the 128-byte id: rank 0 makes it, the file system carries it
    uint8_t id[MI_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        rc = mi_rccl_get_unique_id(id);
        if (rc != MI_OK) DIE("get_unique_id: %s", mi_rccl_last_error());
        std::ofstream(id_file + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(id), sizeof id);
        std::rename((id_file + ".tmp").c_str(), id_file.c_str());
    } else {
        for (int tries = 0;; tries++) {
            std::vector<char> v = read_file(id_file);
            if (v.size() == sizeof id) { std::memcpy(id, v.data(), sizeof id); break; }
            if (tries > 6000) DIE("rank %d: no id file after 60 s", rank);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    mi_rccl_comm* comm = nullptr;
    rc = mi_rccl_comm_create(&comm, ctx, id, n_ranks, rank);
    if (rc != MI_OK) DIE("comm_create: %s", mi_rccl_last_error());

    mi_g1 out{};
    double best_ms = 1e30;
    mi_rccl_timing tm{};
    for (int s = 0; s < steps + 1; s++) {   // one warm-up (ragged shards agree on a window size there), then `steps` timed calls
        auto t0 = std::chrono::steady_clock::now();
        rc = mi_msm_g1_allgather_fold(comm, d_scalars, n, MI_SCALAR_CANONICAL, &out);
        if (rc != MI_OK) DIE("rank %d allgather_fold: %s", rank, mi_rccl_last_error());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s > 0 && ms < best_ms) { best_ms = ms; mi_rccl_last_timing(comm, &tm); }
    }
    std::ofstream(std::string(argv[6]) + "." + std::to_string(rank), std::ios::binary).write(reinterpret_cast<const char*>(&out), sizeof out);
    std::printf("{\"rank\": %d, \"n_ranks\": %d, \"device\": %d, \"points_total\": %zu, \"points_this_rank\": %zu, \"ms\": %.4f, \"msm_ms\": %.4f, "
                "\"exchange_ms\": %.4f, \"window_bits\": %u, \"num_windows\": %u, \"points_per_s\": %.4g}\n",
                rank, n_ranks, dev, N, n, best_ms, tm.msm_ms, tm.exchange_ms, tm.window_bits, tm.num_windows, (double)N / (best_ms * 1e-3));
    mi_rccl_comm_destroy(comm);
    (void)hipFree(d_scalars);
    mi_msm_destroy(ctx);
    return 0;
}
