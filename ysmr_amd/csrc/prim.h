// Device-wide primitives written for the tables of this path (gfx950, wave64): an inclusive prefix sum over
// u32 and a stable least-significant-digit radix sort of u32 / u64 keys with an optional u32 / u64 payload.
// Used by rows.hip (general fallback of ysmr_rows_sort) and select.hip (segment boundaries, stream
// compaction, per-track medians, quantiles).  Tables here hold 1e5 - 3e7 entries and the calls run once per
// video, so the kernels are sized for clarity, not for the last GB/s: every pass is a coalesced stream.
//
// Layout of a radix pass (8-bit digits): tiles of RADIX_TILE consecutive keys, one block per tile;
//   k_radix_hist    per-tile digit histogram -> hist[digit][tile]                (digit-major)
//   scan            exclusive prefix over hist                                   (= first output slot of
//                                                                                 every (digit, tile))
//   k_radix_scatter re-reads the tile IN ORDER, ranks equal digits by position (ballot match within a
//                   wave, per-wave counts across the waves of a round, running counts across rounds) and
//                   writes key and payload to hist[digit][tile] + rank           (stable)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "common.h"

namespace ysmr {
namespace prim {

constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;
constexpr int RADIX_ROUNDS = 8;                     // keys per thread
constexpr int RADIX_TILE = 256 * RADIX_ROUNDS;

// ---- inclusive scan ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    // Hillis-Steele over the 64 lanes by DPP row shifts / broadcasts would save a few cycles; __shfl_up is
    // a ds_bpermute each and this kernel streams 4 B per element -- it is not the bottleneck of anything
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}

// out[i] = in[0] + .. + in[i] within each tile of SCAN_TILE; tile_sum[tile] = the tile's total
static __global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                           uint32_t *__restrict__ tile_sum, size_t n)
{
    __shared__ uint32_t s_wave[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0u;
        run += v[k];
        v[k] = run;
    }
    const uint32_t incl = wave_inclusive_sum(run);
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s_wave[w];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] = v[k] + before;
    if (threadIdx.x == 255 && tile_sum) tile_sum[blockIdx.x] = before + run;
}

// out[i] += (inclusive) prefix of the tiles before tile(i)
static __global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ tile_incl, size_t n)
{
    if (blockIdx.x == 0) return;
    const uint32_t add = tile_incl[blockIdx.x - 1];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] += add;
}

inline size_t scan_tiles(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// scratch (in u32 words) for an inclusive scan of n values: the tile sums of every level
inline size_t scan_temp_words(size_t n)
{
    size_t words = 0;
    while (n > (size_t)SCAN_TILE) {
        n = scan_tiles(n);
        words += align_up(n, 64);
    }
    return words + 64;
}

// out[i] = in[0] + ... + in[i]  (in == out allowed); temp: scan_temp_words(n) u32
inline void inclusive_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, size_t n, uint32_t *temp)
{
    if (n == 0) return;
    const size_t tiles = scan_tiles(n);
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, st, in, out, (uint32_t *)nullptr, n);
        return;
    }
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(256), 0, st, in, out, temp, n);
    inclusive_scan_u32(st, temp, temp, tiles, temp + align_up(tiles, 64));
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(256), 0, st, out, temp, n);
}

// ---- radix sort ----------------------------------------------------------------------------------------
struct NoValue {};

template <typename KeyT>
static __global__ __launch_bounds__(256) void k_radix_hist(const KeyT *__restrict__ keys, size_t n, int shift,
                                                           uint32_t *__restrict__ hist, size_t tiles)
{
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RADIX_TILE;
#pragma unroll
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = s_hist[threadIdx.x];
}

// hist_incl: INCLUSIVE scan of hist (digit-major), so the first slot of (digit d, tile t) is the entry before it
template <typename KeyT, typename ValT>
static __global__ __launch_bounds__(256) void k_radix_scatter(const KeyT *__restrict__ keys, const ValT *__restrict__ vals,
                                                              KeyT *__restrict__ keys_out, ValT *__restrict__ vals_out, size_t n,
                                                              int shift, const uint32_t *__restrict__ hist_incl, size_t tiles)
{
    __shared__ uint32_t s_run[256];        // output slot of the next key of each digit
    __shared__ uint32_t s_wave[4][256];    // keys of each digit held by each wave in this round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const size_t cell = (size_t)threadIdx.x * tiles + blockIdx.x;
        s_run[threadIdx.x] = cell ? hist_incl[cell - 1] : 0u;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    const size_t base = (size_t)blockIdx.x * RADIX_TILE;
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
        for (int k = threadIdx.x; k < 4 * 256; k += 256) (&s_wave[0][0])[k] = 0;
        __syncthreads();
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        const bool live = i < n;
        KeyT key = 0;
        if (live) key = keys[i];
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        // lanes of this wave that hold the same digit (dead lanes match nobody)
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(same & below);
        if (live && rank_in_wave == 0) s_wave[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (live) {
            uint32_t slot = s_run[digit] + rank_in_wave;
            for (int w = 0; w < wave; ++w) slot += s_wave[w][digit];
            keys_out[slot] = key;
            if constexpr (!std::is_same<ValT, NoValue>::value) vals_out[slot] = vals[i];
        }
        __syncthreads();
        s_run[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] + s_wave[3][threadIdx.x];
        __syncthreads();
    }
}

inline size_t radix_tiles(size_t n) { return (n + RADIX_TILE - 1) / RADIX_TILE; }

// scratch in bytes: the histogram matrix and its scan levels (the ping-pong buffers are the caller's)
inline size_t radix_temp_bytes(size_t n)
{
    const size_t cells = 256 * radix_tiles(n);
    return sizeof(uint32_t) * (align_up(cells, 64) + scan_temp_words(cells));
}

// Stable sort by bits [0, bits) of the keys (bits a multiple of 8).  The result ends up in keys_b / vals_b if
// the number of passes is odd, else back in keys_a / vals_a: the return value says which (0 = a, 1 = b).
// Both buffers are overwritten.
template <typename KeyT, typename ValT>
inline int radix_sort(hipStream_t st, KeyT *keys_a, KeyT *keys_b, ValT *vals_a, ValT *vals_b, size_t n, int bits, void *temp)
{
    if (n == 0) return 0;
    const size_t tiles = radix_tiles(n), cells = 256 * tiles;
    uint32_t *hist = (uint32_t *)temp, *scan_tmp = hist + align_up(cells, 64);
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        KeyT *kin = cur ? keys_b : keys_a, *kout = cur ? keys_a : keys_b;
        ValT *vin = cur ? vals_b : vals_a, *vout = cur ? vals_a : vals_b;
        hipLaunchKernelGGL((k_radix_hist<KeyT>), dim3((unsigned)tiles), dim3(256), 0, st, kin, n, shift, hist, tiles);
        inclusive_scan_u32(st, hist, hist, cells, scan_tmp);
        hipLaunchKernelGGL((k_radix_scatter<KeyT, ValT>), dim3((unsigned)tiles), dim3(256), 0, st, kin, vin, kout, vout, n, shift,
                           hist, tiles);
        cur ^= 1;
    }
    return cur;
}

}  // namespace prim
}  // namespace ysmr
