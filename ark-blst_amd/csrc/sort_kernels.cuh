// Two-level LDS-staged bucket sort and the work-item schedule (curve independent).  Compiled in msm_sort.hip only.
#pragma once
#include "kernels_common.cuh"

namespace msmk {

// ---------------------------------------------------------------------------------------------- bucket sort
// Two-level sort of the N*W (window, bucket) keys, staged through LDS — replaces one global atomic per key in the
// histogram pass and one returning global atomic + 4-byte scatter per key in the scatter pass.
//   bucket id = (hi, lo): lo = low `lo_bits` (<= 8) bits -> F = 2^lo_bits fine buckets, H = 2^(c-1) / F coarse bins
//   level 1 (global, coarse):  k_coarse_count   per tile of points: LDS histogram over (window, hi)  -> tilecnt[tile][bin]
//                              k_colscan        per bin: exclusive scan over tiles, bin totals
//                              k_binscan        exclusive scan of the bin totals                     -> bin_base[bin]
//                              k_coarse_scatter per tile: LDS cursors seeded with the scanned bases; entries
//                                               (index | sign | lo) land in their coarse bin of `coarse`
//   level 2 (LDS, fine):       k_fine_count / k_fine_scan / k_fine_scatter over bin SEGMENTS (see below)
//                                               -> sorted[] in (window, bucket) order and hist[window][bucket]
// Windows are processed in groups of `wgroup` so that wgroup * H counters fit LDS (<= 16384 counters, 64 KB).
struct SortGeom {
    uint32_t n, fmt, c, nwin;     // nwin digit windows starting at window win0 (a whole call: win0 = 0, every window)
    uint32_t win0;
    uint32_t lo_bits, H;          // fine bits, coarse bins per window
    uint32_t tiles, tile_pts;     // point tiles (grid.x) and points per tile (multiple of 1024)
    uint32_t tile0;               // first tile of this launch (count passes run per chunk of tiles while host scalars still arrive)
    uint32_t wgroup, ngroups;     // windows per group, groups (grid.y)
    uint32_t nbins;               // bucket windows * H
    uint32_t shared, stride;      // shared = 1: every digit window feeds ONE bucket set (precomputed 2^(c j) P tables of
                                  // `stride` points each); entry index = w * stride + i
};
constexpr uint32_t SORT_MAX_COUNTERS = 16384;  // 64 KB of LDS counters per workgroup (2 workgroups per CU)

// entry in `coarse`: (point index << (lo_bits+1)) | (negative << lo_bits) | lo
template <bool SCATTER, int CB>
__global__ void __launch_bounds__(1024) k_coarse(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flags, SortGeom g,
                                                 uint32_t* __restrict__ tilecnt, const uint32_t* __restrict__ tileoff,
                                                 const uint32_t* __restrict__ bin_base, uint32_t* __restrict__ coarse) {
    aux_priority();
    __shared__ uint32_t cnt[SORT_MAX_COUNTERS];
    const uint32_t tile = blockIdx.x + g.tile0, grp = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
    const uint32_t wend = g.win0 + g.nwin;
    uint32_t w0 = g.win0 + grp * g.wgroup, w1 = w0 + g.wgroup < wend ? w0 + g.wgroup : wend;
    const uint32_t bin0 = g.shared ? 0u : (w0 - g.win0) * g.H;
    uint32_t ncnt = (g.shared ? 1u : (w1 - w0)) * g.H;
    for (uint32_t k = t; k < ncnt; k += nt) {
        if (SCATTER) {
            uint32_t bin = bin0 + k;
            cnt[k] = bin_base[bin] + tileoff[(size_t)tile * g.nbins + bin];  // where this tile's run of the bin starts
        } else {
            cnt[k] = 0;
        }
    }
    __syncthreads();
    uint32_t lo_mask = (1u << g.lo_bits) - 1u;
    uint32_t p0 = tile * g.tile_pts, p1 = p0 + g.tile_pts < g.n ? p0 + g.tile_pts : g.n;
    for (uint32_t i = p0 + t; i < p1; i += nt) {
        if (inf_flags[i] != 0) continue;  // infinity base: contributes nothing
        uint32_t s[8];
        const bool flip = load_scalar(s, scalars, i, g.fmt);
        for_each_digit_static<CB>(s, flip, w0, w1, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t k = (g.shared ? 0u : (w - w0) * g.H) + (b >> g.lo_bits);
            uint32_t pos = atomicAdd(&cnt[k], 1u);
            uint32_t idx = g.shared ? w * g.stride + i : i;
            if (SCATTER) coarse[pos] = (idx << (g.lo_bits + 1)) | ((neg ? 1u : 0u) << g.lo_bits) | (b & lo_mask);
        });
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t k = t; k < ncnt; k += nt) tilecnt[(size_t)tile * g.nbins + bin0 + k] = cnt[k];
    }
}

// LDS-staged form of the coarse scatter for H <= 128 coarse bins per window (c <= 16): a workgroup takes a tile of points and a
// group of windows small enough that ALL its entries fit in LDS (<= 16384 entries, 64 KB), counting-sorts them by bin there
// (positions from the tile's own counts) and writes them out in order: every (tile, bin) run — >= 32 entries — leaves as
// consecutive words instead of one 4-byte store per entry through an LDS cursor.
constexpr uint32_t COARSE_STAGE = 16384;
constexpr uint32_t COARSE_STAGE_BINS = 512;
template <int CB>
__global__ void __launch_bounds__(512) k_coarse_staged(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flags, SortGeom g,
                                                       const uint32_t* __restrict__ tilecnt, const uint32_t* __restrict__ tileoff,
                                                       const uint32_t* __restrict__ bin_base, uint32_t* __restrict__ coarse) {
        aux_priority();
    __shared__ uint32_t stage[COARSE_STAGE];
    __shared__ uint32_t lstart[COARSE_STAGE_BINS + 1], cur[COARSE_STAGE_BINS], goff[COARSE_STAGE_BINS];
    const uint32_t tile = blockIdx.x, grp = blockIdx.y, t = threadIdx.x, nt = 512;
    const uint32_t wend = g.win0 + g.nwin;
    const uint32_t w0 = g.win0 + grp * g.wgroup, w1 = w0 + g.wgroup < wend ? w0 + g.wgroup : wend;
    const uint32_t bin0 = (w0 - g.win0) * g.H, ncnt = (w1 - w0) * g.H;   // ncnt <= COARSE_STAGE_BINS
    const uint32_t mine = t < ncnt ? tilecnt[(size_t)tile * g.nbins + bin0 + t] : 0u;
    goff[t] = t < ncnt ? bin_base[bin0 + t] + tileoff[(size_t)tile * g.nbins + bin0 + t] : 0u;
    cur[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < COARSE_STAGE_BINS; d <<= 1) {   // inclusive scan of the counts
        uint32_t v = t >= d ? cur[t - d] : 0;
        __syncthreads();
        cur[t] += v;
        __syncthreads();
    }
    const uint32_t excl = cur[t] - mine;
    __syncthreads();
    lstart[t] = excl;
    cur[t] = excl;
    if (t == COARSE_STAGE_BINS - 1) lstart[COARSE_STAGE_BINS] = excl + mine;
    __syncthreads();
    const uint32_t total = lstart[COARSE_STAGE_BINS];
    const uint32_t lo_mask = (1u << g.lo_bits) - 1u;
    const uint32_t p0 = tile * g.tile_pts, p1 = p0 + g.tile_pts < g.n ? p0 + g.tile_pts : g.n;
    for (uint32_t i = p0 + t; i < p1; i += nt) {
        if (inf_flags[i] != 0) continue;
        uint32_t s[8];
        const bool flip = load_scalar(s, scalars, i, g.fmt);
        for_each_digit_static<CB>(s, flip, w0, w1, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t k = (w - w0) * g.H + (b >> g.lo_bits);
            uint32_t pos = atomicAdd(&cur[k], 1u);
            stage[pos] = (i << (g.lo_bits + 1)) | ((neg ? 1u : 0u) << g.lo_bits) | (b & lo_mask);
        });
    }
    __syncthreads();
    for (uint32_t j = t; j < total; j += nt) {
        uint32_t k = 0;   // the bin whose run holds position j
#pragma unroll
        for (uint32_t step = COARSE_STAGE_BINS / 2; step >= 1; step >>= 1)
            if (lstart[k + step] <= j) k += step;
        coarse[goff[k] + (j - lstart[k])] = stage[j];
    }
}

// ---- the same kernel for launches that run BESIDE an accumulate kernel (the later groups of a pipelined call, run_msm): 256 lanes, two
// bins per lane, and DYNAMIC LDS (extern __shared__, the size passed at launch; k_fine_scatter / k_mid_scatter likewise).  With a large static array the
// compiler derives the kernel's maximum occupancy from it and PADS the register allocation up to the most that occupancy allows (72 KB
// and 256 lanes: two waves per SIMD, so 176 registers are allocated for a kernel that uses 52; 36 KB: 104 for one that uses 20 — read off the
// kernel descriptors, tools/kernel_resources.py).  Alone that costs nothing; beside an accumulate kernel, which leaves 80 registers per
// lane and SIMD, such a workgroup waits for accumulate waves to retire on all four SIMDs of a compute unit (tools/ubench_coresidency.hip).
constexpr uint32_t COARSE_STAGE_CO = 14336;   // entries staged per workgroup; with the three bin arrays below 62 KB, under the 64 KB a launch may ask for without a function attribute
constexpr uint32_t COARSE_STAGED_CO_LDS = (COARSE_STAGE_CO + 3 * COARSE_STAGE_BINS + 1) * 4;
template <int CB>
__global__ void __launch_bounds__(256) k_coarse_staged_co(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flags, SortGeom g,
                                                       const uint32_t* __restrict__ tilecnt, const uint32_t* __restrict__ tileoff,
                                                       const uint32_t* __restrict__ bin_base, uint32_t* __restrict__ coarse) {
    aux_priority();
    // 256 lanes, two bins per lane (round 6; it was 512 lanes): one wave per SIMD with 56 registers finds room beside the two resident waves
    // of an accumulate kernel (2 x 216 of 512 registers per lane), two did not — in a pipelined call this kernel runs under one (see AUX_BLOCK)
    extern __shared__ uint32_t dyn_lds[];
    uint32_t* const stage = dyn_lds;                                  // [COARSE_STAGE_CO]
    uint32_t* const lstart = stage + COARSE_STAGE_CO;                    // [COARSE_STAGE_BINS + 1]
    uint32_t* const cur = lstart + COARSE_STAGE_BINS + 1;             // [COARSE_STAGE_BINS]
    uint32_t* const goff = cur + COARSE_STAGE_BINS;                   // [COARSE_STAGE_BINS]
    constexpr uint32_t NT = 256;
    static_assert(COARSE_STAGE_BINS == 2 * NT, "two bins per lane");
    const uint32_t tile = blockIdx.x, grp = blockIdx.y, t = threadIdx.x;
    const uint32_t wend = g.win0 + g.nwin;
    const uint32_t w0 = g.win0 + grp * g.wgroup, w1 = w0 + g.wgroup < wend ? w0 + g.wgroup : wend;
    const uint32_t bin0 = (w0 - g.win0) * g.H, ncnt = (w1 - w0) * g.H;   // ncnt <= COARSE_STAGE_BINS
    uint32_t mine[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t k = t + NT * r;
        mine[r] = k < ncnt ? tilecnt[(size_t)tile * g.nbins + bin0 + k] : 0u;
        goff[k] = k < ncnt ? bin_base[bin0 + k] + tileoff[(size_t)tile * g.nbins + bin0 + k] : 0u;
        cur[k] = mine[r];
    }
    __syncthreads();
    for (uint32_t d = 1; d < COARSE_STAGE_BINS; d <<= 1) {   // inclusive scan of the counts, two per lane
        uint32_t v0 = t >= d ? cur[t - d] : 0, v1 = cur[t + NT - d];
        __syncthreads();
        cur[t] += v0; cur[t + NT] += v1;
        __syncthreads();
    }
    const uint32_t e0 = cur[t] - mine[0], e1 = cur[t + NT] - mine[1];
    __syncthreads();
    lstart[t] = e0; lstart[t + NT] = e1;
    cur[t] = e0; cur[t + NT] = e1;
    if (t == NT - 1) lstart[COARSE_STAGE_BINS] = e1 + mine[1];
    __syncthreads();
    const uint32_t total = lstart[COARSE_STAGE_BINS];
    const uint32_t lo_mask = (1u << g.lo_bits) - 1u;
    const uint32_t p0 = tile * g.tile_pts, p1 = p0 + g.tile_pts < g.n ? p0 + g.tile_pts : g.n;
    for (uint32_t i = p0 + t; i < p1; i += NT) {
        if (inf_flags[i] != 0) continue;
        uint32_t s[8];
        const bool flip = load_scalar(s, scalars, i, g.fmt);
        for_each_digit_static<CB>(s, flip, w0, w1, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t k = (w - w0) * g.H + (b >> g.lo_bits);
            uint32_t pos = atomicAdd(&cur[k], 1u);
            stage[pos] = (i << (g.lo_bits + 1)) | ((neg ? 1u : 0u) << g.lo_bits) | (b & lo_mask);
        });
    }
    __syncthreads();
    for (uint32_t j = t; j < total; j += NT) {
        uint32_t k = 0;   // the bin whose run holds position j
#pragma unroll
        for (uint32_t step = COARSE_STAGE_BINS / 2; step >= 1; step >>= 1)
            if (lstart[k + step] <= j) k += step;
        coarse[goff[k] + (j - lstart[k])] = stage[j];
    }
}

// per bin: exclusive scan over the tiles (tilecnt keeps the counts, tileoff gets the offsets) and the bin total.  One lane per
// bin: coalesced across bins; eight tiles' loads in flight per lane.
__global__ void __launch_bounds__(256) k_colscan(const uint32_t* __restrict__ tilecnt, uint32_t nbins, uint32_t tiles,
                                                 uint32_t* __restrict__ tileoff, uint32_t* __restrict__ bin_tot) {
    aux_priority();
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    uint32_t run = 0, k = 0;
    for (; k + 8 <= tiles; k += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = tilecnt[(size_t)(k + u) * nbins + b];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            tileoff[(size_t)(k + u) * nbins + b] = run;
            run += v[u];
        }
    }
    for (; k < tiles; k++) {
        uint32_t v = tilecnt[(size_t)k * nbins + b];
        tileoff[(size_t)k * nbins + b] = run;
        run += v;
    }
    bin_tot[b] = run;
}

// exclusive scan of m <= ~100k values by one workgroup; out[m] = total
__global__ void __launch_bounds__(1024) k_binscan(const uint32_t* __restrict__ in, uint32_t m, uint32_t* __restrict__ out) {
    aux_priority();
    __shared__ uint32_t part[1024];
    uint32_t t = threadIdx.x;
    uint32_t per = (m + 1023) / 1024;
    uint32_t lo = t * per, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += in[k];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t v = in[k];
        out[k] = run;
        run += v;
    }
    if (t == 1023) out[m] = part[1023];
}

// ---- level 2: fine sort.  A coarse bin is cut into SEGMENTS of at most FINE_SEG entries, one workgroup each, so a
// heavy bin (skewed scalars; the short top window, whose n entries share a handful of buckets) is spread over
// the chip instead of being streamed by a single workgroup (measured: 30 ms for two 8 M-entry bins at n = 2^24).
//   k_seg_count   segments per bin                       -> seg_cnt[bin]      (then k_binscan -> seg_base[bin])
//   k_fine_count  per segment: LDS histogram of lo       -> segcnt[seg][lo]
//   k_fine_scan   per bin: scan over its segments and over lo -> segcnt becomes the start of (seg, lo) inside the
//                 bin; emits hist[window][bucket]
//   k_fine_scatter per segment: LDS cursors seeded from segcnt -> sorted[]
constexpr uint32_t FINE_SEG = 8192;
constexpr uint32_t FINE_SCATTER_LDS = (FINE_SEG + 257 + 256 + 256 + 4) * 4;

__global__ void __launch_bounds__(256) k_seg_count(const uint32_t* __restrict__ bin_base, uint32_t nbins, uint32_t* __restrict__ seg_cnt) {
    aux_priority();
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    uint32_t sz = bin_base[b + 1] - bin_base[b];
    seg_cnt[b] = sz == 0 ? 1u : (sz + FINE_SEG - 1) / FINE_SEG;
}

// segment id -> (bin, first entry, one-past-last entry); identity guess first (no bin split before it), else binary search
__device__ __forceinline__ bool seg_locate(uint32_t seg, const uint32_t* seg_base, const uint32_t* bin_base, uint32_t nbins,
                                           uint32_t& bin, uint32_t& beg, uint32_t& end) {
    if (seg >= seg_base[nbins]) return false;
    uint32_t b = seg < nbins ? seg : nbins - 1;
    if (!(seg_base[b] <= seg && seg < seg_base[b + 1])) {
        uint32_t lo = 0, hi = b;
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (seg_base[mid] <= seg) lo = mid; else hi = mid - 1;
        }
        b = lo;
    }
    uint32_t k = seg - seg_base[b];
    bin = b;
    beg = bin_base[b] + k * FINE_SEG;
    uint32_t bend = bin_base[b + 1];
    end = beg + FINE_SEG < bend ? beg + FINE_SEG : bend;
    return true;
}

__global__ void __launch_bounds__(256) k_fine_count(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ bin_base,
                                                    const uint32_t* __restrict__ seg_base, SortGeom g, uint32_t* __restrict__ segcnt) {
    aux_priority();
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t sb[3];
    uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, seg_base, bin_base, g.nbins, bin, beg, end);
        sb[0] = ok ? beg : 1u; sb[1] = ok ? end : 0u; sb[2] = ok ? 1u : 0u;
    }
    cnt[t] = 0;
    __syncthreads();
    if (!sb[2]) return;
    uint32_t beg = sb[0], end = sb[1];
    uint32_t F = 1u << g.lo_bits, lo_mask = F - 1u;
    for (uint32_t i = beg + t; i < end; i += 256) atomicAdd(&cnt[coarse[i] & lo_mask], 1u);
    __syncthreads();
    if (t < F) segcnt[(size_t)seg * F + t] = cnt[t];
}

// one workgroup per bin, lane = lo.  segoff[seg][lo] <- offset of (seg, lo) relative to the bin start.
__global__ void __launch_bounds__(256) k_fine_scan(const uint32_t* __restrict__ seg_base, SortGeom g, const uint32_t* __restrict__ segcnt,
                                                   uint32_t* __restrict__ segoff, uint32_t* __restrict__ hist) {
    aux_priority();
    __shared__ uint32_t scan[256];
    uint32_t bin = blockIdx.x, t = threadIdx.x;
    uint32_t F = 1u << g.lo_bits;
    uint32_t s0 = seg_base[bin], s1 = seg_base[bin + 1];
    uint32_t tot = 0;
    if (t < F)
        for (uint32_t sg = s0; sg < s1; sg++) {
            uint32_t v = segcnt[(size_t)sg * F + t];
            segoff[(size_t)sg * F + t] = tot;   // exclusive over the segments of this (bin, lo); the counts stay in segcnt
            tot += v;
        }
    scan[t] = t < F ? tot : 0;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    if (t < F) {
        hist[(size_t)bin * F + t] = tot;                 // bucket (window, hi, lo) has index bin * F + lo
        uint32_t lo_base = scan[t] - tot;                // start of fine bucket lo inside the bin
        for (uint32_t sg = s0; sg < s1; sg++) segoff[(size_t)sg * F + t] += lo_base;
    }
}

// LDS-staged: the segment's entries are first counting-sorted by `lo` inside LDS (positions from the segment's own counts),
// then written out in order, so consecutive lanes store consecutive words of a (segment, lo) run — a bin that fits one segment
// is written as ONE contiguous range.  (The first version scattered every 4-byte entry straight to its global position through
// an LDS cursor: 16.8 M single-word stores at 2^20.)
__global__ void __launch_bounds__(256) k_fine_scatter(const uint32_t* __restrict__ coarse, const uint32_t* __restrict__ bin_base,
                                                      const uint32_t* __restrict__ seg_base, SortGeom g,
                                                      const uint32_t* __restrict__ segcnt, const uint32_t* __restrict__ segoff,
                                                      uint32_t* __restrict__ sorted) {
    aux_priority();
    extern __shared__ uint32_t dyn_lds[];   // dynamic: see COARSE_STAGE
    uint32_t* const stage = dyn_lds;        // [FINE_SEG]
    uint32_t* const lstart = stage + FINE_SEG;   // [257]
    uint32_t* const cur = lstart + 257;     // [256]
    uint32_t* const goff = cur + 256;       // [256]
    uint32_t* const sb = goff + 256;        // [4]
    uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, seg_base, bin_base, g.nbins, bin, beg, end);
        sb[0] = beg; sb[1] = end; sb[2] = ok ? 1u : 0u; sb[3] = ok ? bin_base[bin] : 0u;
    }
    __syncthreads();
    if (!sb[2]) return;
    const uint32_t beg = sb[0], end = sb[1], bin_beg = sb[3];
    const uint32_t F = 1u << g.lo_bits, lo_mask = F - 1u;
    const uint32_t mine = t < F ? segcnt[(size_t)seg * F + t] : 0u;
    goff[t] = t < F ? bin_beg + segoff[(size_t)seg * F + t] : 0u;
    cur[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {   // inclusive scan of the 256 counts
        uint32_t v = t >= d ? cur[t - d] : 0;
        __syncthreads();
        cur[t] += v;
        __syncthreads();
    }
    const uint32_t excl = cur[t] - mine;
    __syncthreads();
    lstart[t] = excl;
    cur[t] = excl;
    if (t == 255) lstart[256] = excl + mine;
    __syncthreads();
    const uint32_t sh = g.lo_bits + 1;
    for (uint32_t i = beg + t; i < end; i += 256) {
        uint32_t e = coarse[i];
        uint32_t pos = atomicAdd(&cur[e & lo_mask], 1u);
        stage[pos] = (e >> sh) | (((e >> g.lo_bits) & 1u) << 31);
    }
    __syncthreads();
    const uint32_t len = end - beg;
    for (uint32_t j = t; j < len; j += 256) {
        uint32_t lo = 0;   // largest lo with lstart[lo] <= j (empty buckets share a start: take the last one that begins at or before j and is non-empty)
#pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1)
            if (lstart[lo + step] <= j) lo += step;
        sorted[goff[lo] + (j - lstart[lo])] = stage[j];
    }
}

// ---------------------------------------------------------------------------------------------- three-level sort (c >= 17)
// With c >= 17 a window has up to 4096 coarse bins: a workgroup of the one-step coarse scatter keeps up to 16384 runs open at
// once, far more than L2 can merge, so every 4-byte entry reaches HBM as a partial line (2.8 ms at 2^24 for 0.87 GB), and the
// window groups re-read the scalars 4 to 13 times.  The coarse level is therefore split in two when c >= 17:
//   level A   bins = (window, top 5 bucket bits): <= 512 bins for ALL windows together, so the scalars are read once per pass and a
//             workgroup keeps <= 512 runs open, few enough for L2 to merge its stores (staging this scatter through LDS as well
//             was measured: 2.55 vs 2.42 ms for the level at 2^24, not kept).  Entries are 8 bytes here — (point index, sign | remaining bucket bits): index +
//             sign + up to 16 remaining bits do not fit one word.
//   level B   per <= 8192-entry segment of an A bin: counting sort in LDS by the next MID bits (<= 512 sub-bins), written out as
//             the 4-byte entries and the (window, hi) bins the fine level expects — coarse[] and bin_base[] come out exactly as
//             the one-step coarse pass leaves them.
constexpr uint32_t A_BITS = 5, A_BINS = 1u << A_BITS;
constexpr uint32_t MID_MAX = 512;
constexpr uint32_t MID_SCATTER_LDS = (8192 + 3 * MID_MAX + 1 + 4) * 4;   // FINE_SEG entries + the three sub-bin arrays

template <bool SCATTER, int CB>
__global__ void __launch_bounds__(1024) k_coarseA(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flags, SortGeom g,
                                                  uint32_t* __restrict__ tilecnt, const uint32_t* __restrict__ tileoff,
                                                  const uint32_t* __restrict__ binA_base, uint2* __restrict__ coarseA) {
    aux_priority();
    __shared__ uint32_t cnt[512];
    const uint32_t tile = blockIdx.x + g.tile0, t = threadIdx.x, nt = blockDim.x;
    const uint32_t nbinsA = g.nwin * A_BINS;
    for (uint32_t k = t; k < nbinsA; k += nt) cnt[k] = SCATTER ? binA_base[k] + tileoff[(size_t)tile * nbinsA + k] : 0u;
    __syncthreads();
    constexpr uint32_t REM = CB - 1 - A_BITS;   // bucket bits left below the A bin
    const uint32_t p0 = tile * g.tile_pts, p1 = p0 + g.tile_pts < g.n ? p0 + g.tile_pts : g.n;
    for (uint32_t i = p0 + t; i < p1; i += nt) {
        if (inf_flags[i] != 0) continue;
        uint32_t s[8];
        const bool flip = load_scalar(s, scalars, i, g.fmt);
        for_each_digit_static<CB>(s, flip, g.win0, g.win0 + g.nwin, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t pos = atomicAdd(&cnt[(w - g.win0) * A_BINS + (b >> REM)], 1u);
            if (SCATTER) coarseA[pos] = make_uint2(i, ((neg ? 1u : 0u) << 31) | (b & ((1u << REM) - 1u)));
        });
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t k = t; k < nbinsA; k += nt) tilecnt[(size_t)tile * nbinsA + k] = cnt[k];
    }
}

// level B, per segment of an A bin: histogram of the MID bits
__global__ void __launch_bounds__(256) k_mid_count(const uint2* __restrict__ coarseA, const uint32_t* __restrict__ binA_base,
                                                   const uint32_t* __restrict__ segA_base, uint32_t nbinsA, uint32_t lo_bits, uint32_t mid_bits,
                                                   uint32_t* __restrict__ segcnt) {
    aux_priority();
    __shared__ uint32_t cnt[MID_MAX];
    __shared__ uint32_t sb[3];
    uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, segA_base, binA_base, nbinsA, bin, beg, end);
        sb[0] = ok ? beg : 1u; sb[1] = ok ? end : 0u; sb[2] = ok ? 1u : 0u;
    }
    cnt[t] = 0; cnt[t + 256] = 0;
    __syncthreads();
    if (!sb[2]) return;
    const uint32_t beg = sb[0], end = sb[1], M = 1u << mid_bits;
    for (uint32_t i = beg + t; i < end; i += 256) atomicAdd(&cnt[(coarseA[i].y >> lo_bits) & (M - 1u)], 1u);
    __syncthreads();
    for (uint32_t k = t; k < M; k += 256) segcnt[(size_t)seg * M + k] = cnt[k];
}

// level B, one workgroup per A bin: offsets of every (segment, sub-bin) run inside the A bin, and the bases of the final bins
// (window, top 5 bits, MID bits) = the bins of the one-step coarse pass: bin_base[a * M + sub].
__global__ void __launch_bounds__(512) k_mid_scan(const uint32_t* __restrict__ binA_base, const uint32_t* __restrict__ segA_base,
                                                  uint32_t mid_bits, const uint32_t* __restrict__ segcnt, uint32_t* __restrict__ segoff,
                                                  uint32_t* __restrict__ bin_base, uint32_t nbinsA) {
    aux_priority();
    __shared__ uint32_t scan[MID_MAX];
    const uint32_t a = blockIdx.x, t = threadIdx.x, M = 1u << mid_bits;
    const uint32_t s0 = segA_base[a], s1 = segA_base[a + 1];
    // lane = sub-bin (an A bin of 2^24 points has 64 segments: the loads of eight of them are kept in flight per lane)
    uint32_t tot = 0;
    if (t < M) {
        uint32_t sg = s0;
        for (; sg + 8 <= s1; sg += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = segcnt[(size_t)(sg + u) * M + t];
#pragma unroll
            for (int u = 0; u < 8; u++) tot += v[u];
        }
        for (; sg < s1; sg++) tot += segcnt[(size_t)sg * M + t];
    }
    scan[t] = t < M ? tot : 0;
    __syncthreads();
    for (uint32_t d = 1; d < MID_MAX; d <<= 1) {
        uint32_t v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    if (t < M) {
        uint32_t run = scan[t] - tot;   // start of sub-bin t inside the A bin; then exclusive over the segments
        bin_base[(size_t)a * M + t] = binA_base[a] + run;
        uint32_t sg = s0;
        for (; sg + 8 <= s1; sg += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = segcnt[(size_t)(sg + u) * M + t];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                segoff[(size_t)(sg + u) * M + t] = run;
                run += v[u];
            }
        }
        for (; sg < s1; sg++) {
            uint32_t v = segcnt[(size_t)sg * M + t];
            segoff[(size_t)sg * M + t] = run;
            run += v;
        }
    }
    if (a == nbinsA - 1 && t == 0) bin_base[(size_t)nbinsA * M] = binA_base[nbinsA];
}

// level B scatter, LDS-staged like k_fine_scatter: 8-byte entries in, 4-byte entries out
__global__ void __launch_bounds__(256) k_mid_scatter(const uint2* __restrict__ coarseA, const uint32_t* __restrict__ binA_base,
                                                     const uint32_t* __restrict__ segA_base, uint32_t nbinsA, uint32_t lo_bits, uint32_t mid_bits,
                                                     const uint32_t* __restrict__ segcnt, const uint32_t* __restrict__ segoff,
                                                     uint32_t* __restrict__ coarse) {
    aux_priority();
    extern __shared__ uint32_t dyn_lds[];   // dynamic: see COARSE_STAGE
    uint32_t* const stage = dyn_lds;                 // [FINE_SEG]
    uint32_t* const lstart = stage + FINE_SEG;       // [MID_MAX + 1]
    uint32_t* const cur = lstart + MID_MAX + 1;      // [MID_MAX]
    uint32_t* const goff = cur + MID_MAX;            // [MID_MAX]
    uint32_t* const sb = goff + MID_MAX;             // [4]
    const uint32_t t = threadIdx.x, seg = blockIdx.x;
    if (t == 0) {
        uint32_t bin = 0, beg = 0, end = 0;
        bool ok = seg_locate(seg, segA_base, binA_base, nbinsA, bin, beg, end);
        sb[0] = beg; sb[1] = end; sb[2] = ok ? 1u : 0u; sb[3] = ok ? binA_base[bin] : 0u;
    }
    __syncthreads();
    if (!sb[2]) return;
    const uint32_t beg = sb[0], end = sb[1], binA_beg = sb[3], M = 1u << mid_bits;
    uint32_t mine[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t k = t + 256 * r;
        mine[r] = k < M ? segcnt[(size_t)seg * M + k] : 0u;
        goff[k] = k < M ? binA_beg + segoff[(size_t)seg * M + k] : 0u;
        cur[k] = mine[r];
    }
    __syncthreads();
    for (uint32_t d = 1; d < MID_MAX; d <<= 1) {   // inclusive scan of the 512 counts, two per lane
        uint32_t v0 = t >= d ? cur[t - d] : 0, v1 = cur[t + 256 - d];
        __syncthreads();
        cur[t] += v0; cur[t + 256] += v1;
        __syncthreads();
    }
    const uint32_t e0 = cur[t] - mine[0], e1 = cur[t + 256] - mine[1];
    __syncthreads();
    lstart[t] = e0; lstart[t + 256] = e1;
    cur[t] = e0; cur[t + 256] = e1;
    if (t == 255) lstart[MID_MAX] = e1 + mine[1];
    __syncthreads();
    const uint32_t lo_mask = (1u << lo_bits) - 1u;
    for (uint32_t i = beg + t; i < end; i += 256) {
        uint2 e = coarseA[i];
        uint32_t pos = atomicAdd(&cur[(e.y >> lo_bits) & (M - 1u)], 1u);
        stage[pos] = (e.x << (lo_bits + 1)) | ((e.y >> 31) << lo_bits) | (e.y & lo_mask);
    }
    __syncthreads();
    const uint32_t len = end - beg;
    for (uint32_t j = t; j < len; j += 256) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = MID_MAX / 2; step >= 1; step >>= 1)
            if (lstart[k + step] <= j) k += step;
        coarse[goff[k] + (j - lstart[k])] = stage[j];
    }
}

// ---------------------------------------------------------------------------------------------- scan / schedule
// Three small launches turn the histogram into (a) entry offsets for the scatter, (b) WORK ITEMS and (c) a
// length-sorted processing order for them:
//   * a bucket with cnt entries becomes max(1, ceil(cnt / T)) items of at most T = 2^logT entries, so a heavy
//     bucket (skewed scalars, or the short top window) is spread over many lanes instead of serialising one;
//   * order[] lists item ids by DESCENDING length class (65 classes): the 64 lanes of a wave then run the same
//     number of additions (bucket loads are Poisson distributed: unsorted, a wave waits for its longest lane,
//     ~70 % lane efficiency at a mean of 32) and the longest items start first.
// Layout: nblk <= SCHED_MAX_BLK blocks of NT lanes; block k owns `per_blk` consecutive buckets, lane t owns per_blk / NT consecutive ones.  item id of (bucket b, chunk k) = woff[b] + k.
// The kernels are templates over the workgroup size NT: 1024 lanes for a one-group call; 512 for the later groups of a pipelined call (round 6),
// whose schedule runs under an accumulate kernel — two waves per SIMD of <= 32 registers find room beside its two resident waves (80 of
// 512 registers per lane are left), four do not.
constexpr int SCHED_CLASSES = 65;

// A bucket of up to T = 2^logT entries is ONE item; a fuller one is split into items of S = 2^logS entries, S = max(16, T / 4) (round 4: it
// was T).  Skewed scalars (witness bits: half the scalars are 0 or 1) put 10^5..10^6 entries into one bucket; with items of T = 64 a lane
// then walked 64 dependent additions (0.9 ms) while most of the machine idled; items of 16 finish in 0.2 ms and are merged by k_merge's
// fan-in tree.  The launch packs logT | class shift << 8 | logS << 16 into one argument; item_range (curve_kernels.cuh) applies the same rule.
__device__ __forceinline__ uint32_t items_of(uint32_t cnt, uint32_t logT, uint32_t logS) {
    const uint32_t lg = item_size_log(cnt, logT, logS);
    return cnt == 0 ? 1u : (cnt + (1u << lg) - 1u) >> lg;   // an empty bucket keeps one (empty) item: it leaves infinity for the reduce
}
// entries of a bucket's LAST item (the others are full: S entries)
__device__ __forceinline__ uint32_t last_len(uint32_t cnt, uint32_t it, uint32_t logS) { return it > 1 ? cnt - ((it - 1) << logS) : cnt; }
// length class 0..64 of an item: len >> cls_shift, clamped.  The class width follows the TYPICAL item (twice the mean bucket load
// spans the 64 classes), not T: when T is raised for a long kernel the ordinary buckets must still be sorted to a few entries,
// or the lanes of a wave walk items of visibly different lengths (measured +8 % on the accumulate kernel at 2^24).
__device__ __forceinline__ uint32_t class_of(uint32_t len, uint32_t cls_shift) { uint32_t c = len >> cls_shift; return c < 64 ? c : 64; }

template <uint32_t NT>
__global__ void __launch_bounds__(NT) k_sched1(const uint32_t* __restrict__ hist, uint32_t m, uint32_t per_blk, uint32_t logT,
                                                 uint32_t nblk, uint32_t* __restrict__ blk_e, uint32_t* __restrict__ blk_i,
                                                 uint32_t* __restrict__ blk_cls, uint32_t* __restrict__ blk_max) {
    aux_priority();
    __shared__ uint32_t cls[SCHED_CLASSES];
    __shared__ uint32_t se, si, smax;
    uint32_t t = threadIdx.x, blk = blockIdx.x;
    if (t < SCHED_CLASSES) cls[t] = 0;
    if (t == 0) { se = 0; si = 0; smax = 1; }
    __syncthreads();
    uint32_t per_t = per_blk / NT;
    uint32_t lo = blk * per_blk + t * per_t, hi = lo + per_t < m ? lo + per_t : m;
    const uint32_t cls_shift = (logT >> 8) & 0xffu, logS = (logT >> 16) & 0xffu;   // the launch packs log2 T | class shift << 8 | log2 S << 16
    logT &= 0xffu;
    uint32_t sum_e = 0, sum_i = 0, mx = 1;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k], it = items_of(h, logT, logS);
        sum_e += h;
        sum_i += it;
        mx = it > mx ? it : mx;
        if (it > 1) atomicAdd(&cls[64], it - 1);
        atomicAdd(&cls[class_of(last_len(h, it, logS), cls_shift)], 1u);
    }
    atomicAdd(&se, sum_e);
    atomicAdd(&si, sum_i);
    atomicMax(&smax, mx);
    __syncthreads();
    if (t < SCHED_CLASSES) blk_cls[t * nblk + blk] = cls[t];
    if (t == 0) { blk_e[blk] = se; blk_i[blk] = si; blk_max[blk] = smax; }
}

// one workgroup: exclusive scans over the <= SCHED_MAX_BLK block sums, and per (class, block) the base position in order[]
// (1024 blocks since round 5: at 2^24 points the 6.8 M buckets were spread over 208 blocks, 32 consecutive buckets per lane — one lane per
// 128-byte line, one block per CU — and k_sched1 / k_sched3 took 0.12 + 0.50 ms)
constexpr uint32_t SCHED_MAX_BLK = 1024;
template <uint32_t NT>
__global__ void __launch_bounds__(NT) k_sched2(uint32_t nblk, uint32_t* __restrict__ blk_e, uint32_t* __restrict__ blk_i,
                                                 uint32_t* __restrict__ blk_cls, const uint32_t* __restrict__ blk_max,
                                                 uint32_t* __restrict__ meta) {
    aux_priority();
    __shared__ uint32_t a[SCHED_MAX_BLK], b[SCHED_MAX_BLK], ctot[SCHED_CLASSES], cbase[SCHED_CLASSES];
    uint32_t t = threadIdx.x;
    constexpr uint32_t PER = SCHED_MAX_BLK / NT;   // block sums per lane (1 or 2)
    static_assert(PER == 1 || PER == 2, "k_sched2: 512 or 1024 lanes");
    uint32_t ve[PER], vi[PER];
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
        const uint32_t k = t + NT * r;
        ve[r] = k < nblk ? blk_e[k] : 0;
        vi[r] = k < nblk ? blk_i[k] : 0;
        a[k] = ve[r];
        b[k] = vi[r];
    }
    __syncthreads();
    for (uint32_t d = 1; d < SCHED_MAX_BLK; d <<= 1) {
        uint32_t xa[PER], xb[PER];
#pragma unroll
        for (uint32_t r = 0; r < PER; r++) {
            const uint32_t k = t + NT * r;
            xa[r] = k >= d ? a[k - d] : 0;
            xb[r] = k >= d ? b[k - d] : 0;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < PER; r++) {
            const uint32_t k = t + NT * r;
            a[k] += xa[r];
            b[k] += xb[r];
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
        const uint32_t k = t + NT * r;
        if (k < nblk) { blk_e[k] = a[k] - ve[r]; blk_i[k] = b[k] - vi[r]; }
    }
    // class rows: wave w handles classes w, w + NT / 64, ...; exclusive scan of each row in chunks of 64 lanes
    uint32_t wave = t >> 6, lane = t & 63;
    for (uint32_t c = wave; c < SCHED_CLASSES; c += NT / 64) {
        uint32_t run = 0;
        for (uint32_t base = 0; base < nblk; base += 64) {
            uint32_t idx = base + lane;
            uint32_t v = idx < nblk ? blk_cls[c * nblk + idx] : 0, incl = v;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
            if (idx < nblk) blk_cls[c * nblk + idx] = run + incl - v;
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) ctot[c] = run;
    }
    __syncthreads();
    if (t < SCHED_CLASSES) {  // descending class order: class c starts after all longer classes
        uint32_t above = 0;
        for (uint32_t c = t + 1; c < SCHED_CLASSES; c++) above += ctot[c];
        cbase[t] = above;
    }
    __syncthreads();
    for (uint32_t idx = t; idx < SCHED_CLASSES * nblk; idx += NT) blk_cls[idx] += cbase[idx / nblk];
    if (t == 0) {
        uint32_t mx = 1;
        for (uint32_t k = 0; k < nblk; k++) mx = blk_max[k] > mx ? blk_max[k] : mx;
        meta[0] = b[SCHED_MAX_BLK - 1];  // total items   (a[], b[] are inclusive scans; entries past nblk are zero)
        meta[1] = mx;      // max items of any bucket
        meta[2] = a[SCHED_MAX_BLK - 1];  // total entries
        meta[3] = 0;       // merge-list length (level 0 of the merge tree), filled by k_sched3
        meta[4] = 0;       // number of split buckets, filled by k_sched3
        for (uint32_t k = 5; k < MERGE_META; k++) meta[k] = 0;   // [8 + l]: list length of merge level l >= 1 (k_merge appends)
    }
}

// per block: bucket-level exclusive scans -> offsets / cursor / woff; every item gets its slot in order[]
template <uint32_t NT>
__global__ void __launch_bounds__(NT) k_sched3(const uint32_t* __restrict__ hist, uint32_t m, uint32_t per_blk, uint32_t logT,
                                                 uint32_t nblk, const uint32_t* __restrict__ blk_e, const uint32_t* __restrict__ blk_i,
                                                 const uint32_t* __restrict__ blk_cls, uint32_t* __restrict__ offsets,
                                                 uint32_t* __restrict__ woff,
                                                 uint32_t* __restrict__ order, uint32_t* __restrict__ item_bucket,
                                                 uint32_t* __restrict__ merge_list, uint32_t* __restrict__ meta) {
    aux_priority();
    __shared__ uint32_t pe[NT], pi[NT], cur[SCHED_CLASSES];
    // buckets split into many items (skewed scalars: one bucket can hold all N entries) are written out by the whole
    // workgroup after the per-lane pass; one lane doing it alone cost 0.65 ms for a bucket of 2^20 entries
    constexpr uint32_t HV_CAP = 512, HV_MIN = 64;   // (64 slots until round 4: the 128 buckets of a 7-bit top window overflowed them, and a lone lane wrote 4096 items each — 1.2 ms)
    __shared__ uint32_t hv_k[HV_CAP], hv_run[HV_CAP], hv_it[HV_CAP], hv_pos[HV_CAP], hv_mp[HV_CAP], hv_n;
    uint32_t t = threadIdx.x, blk = blockIdx.x;
    if (t == 0) hv_n = 0;
    if (t < SCHED_CLASSES) cur[t] = blk_cls[t * nblk + blk];
    uint32_t per_t = per_blk / NT;
    uint32_t lo = blk * per_blk + t * per_t, hi = lo + per_t < m ? lo + per_t : m;
    const uint32_t cls_shift = (logT >> 8) & 0xffu, logS = (logT >> 16) & 0xffu;
    logT &= 0xffu;
    uint32_t sum_e = 0, sum_i = 0;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k];
        sum_e += h;
        sum_i += items_of(h, logT, logS);
    }
    pe[t] = sum_e;
    pi[t] = sum_i;
    __syncthreads();
    for (uint32_t d = 1; d < NT; d <<= 1) {
        uint32_t xe = t >= d ? pe[t - d] : 0, xi = t >= d ? pi[t - d] : 0;
        __syncthreads();
        pe[t] += xe;
        pi[t] += xi;
        __syncthreads();
    }
    uint32_t run_e = blk_e[blk] + pe[t] - sum_e, run_i = blk_i[blk] + pi[t] - sum_i;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t h = hist[k], it = items_of(h, logT, logS);
        offsets[k] = run_e;
        woff[k] = run_i;
        if (it > 1) {  // full-length chunks of a split bucket: one reservation in the longest class
            uint32_t pos = atomicAdd(&cur[64], it - 1);
            // the merge tree only visits the items of split buckets: level 0 lists every MERGE_FAN-th item of the bucket
            uint32_t mp = atomicAdd(&meta[3], (it + MERGE_FAN - 1) / MERGE_FAN);
            atomicAdd(&meta[4], 1u);
            uint32_t slot = it >= HV_MIN ? atomicAdd(&hv_n, 1u) : HV_CAP;
            if (slot < HV_CAP) {
                hv_k[slot] = k; hv_run[slot] = run_i; hv_it[slot] = it; hv_pos[slot] = pos; hv_mp[slot] = mp;
            } else {
                for (uint32_t j = 0; j + 1 < it; j++) { order[pos + j] = run_i + j; item_bucket[run_i + j] = k; }
                for (uint32_t j = 0; j < it; j += MERGE_FAN) merge_list[mp + j / MERGE_FAN] = run_i + j;
            }
        }
        uint32_t last = run_i + it - 1;
        uint32_t pos = atomicAdd(&cur[class_of(last_len(h, it, logS), cls_shift)], 1u);
        order[pos] = last;
        item_bucket[last] = k;
        run_e += h;
        run_i += it;
    }
    if (blk == nblk - 1 && t == NT - 1) { offsets[m] = run_e; woff[m] = run_i; }
    __syncthreads();
    const uint32_t nh = hv_n < HV_CAP ? hv_n : HV_CAP;
    for (uint32_t s = 0; s < nh; s++) {
        const uint32_t k = hv_k[s], r0 = hv_run[s], it = hv_it[s], pos = hv_pos[s], mp = hv_mp[s];
        for (uint32_t j = t; j < it; j += NT) {
            if (j + 1 < it) { order[pos + j] = r0 + j; item_bucket[r0 + j] = k; }
            if (j % MERGE_FAN == 0) merge_list[mp + j / MERGE_FAN] = r0 + j;
        }
    }
}

}  // namespace msmk
