// gm_compact.hpp -- order-preserving stream compaction (count -> scan -> scatter).
//
// Used for the crop box (CropBox keeps survivors in input order,
// /root/reference src/tunnel_processing.cpp:42-46), the NaN-normal removal
// (src/tunnel_processing.cpp:74-85) and the voxel segment heads.  Output order
// is part of the observable result (/choppedCloud point order), so the scan is
// exact and deterministic: wave ballot + popcount for the rank inside a wave,
// a 4-entry LDS table for the rank of the wave inside the block, one global
// exclusive scan of per-block counts.
#pragma once

#include "gm_device.hpp"

namespace gm {

constexpr int kCpThreads = 256;
constexpr int kCpItems = 8;
constexpr int kCpTile = kCpThreads * kCpItems;  // points per block

inline uint32_t compact_blocks(uint32_t n) { return (n + kCpTile - 1) / kCpTile; }

// Pred: __device__ bool operator()(uint32_t i) const
template <class Pred>
__global__ __launch_bounds__(kCpThreads) void k_compact_count(Pred pred, const uint32_t *__restrict__ n_ptr,
                                                               uint32_t n_host, uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wsum[kCpThreads / kWave];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t base = blockIdx.x * kCpTile;
    uint32_t cnt = 0;
    if (base < n) {
#pragma unroll
        for (int j = 0; j < kCpItems; ++j) {
            uint32_t i = base + j * kCpThreads + threadIdx.x;
            bool v = (i < n) && pred(i);
            cnt += (uint32_t)__popcll(__ballot(v));
        }
    }
    if (lane_id() == 0) wsum[threadIdx.x / kWave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < kCpThreads / kWave; ++w) s += wsum[w];
        block_counts[blockIdx.x] = s;
    }
}

// Emit: __device__ void operator()(uint32_t src, uint32_t dst) const
template <class Pred, class Emit>
__global__ __launch_bounds__(kCpThreads) void k_compact_scatter(Pred pred, Emit emit,
                                                                 const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                                 const uint32_t *__restrict__ block_offsets)
{
    __shared__ uint32_t wcnt[2][kCpThreads / kWave];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t base = blockIdx.x * kCpTile;
    if (base >= n) return;  // uniform per block
    const int w = threadIdx.x / kWave;
    uint32_t running = block_offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kCpItems; ++j) {
        uint32_t i = base + j * kCpThreads + threadIdx.x;
        bool v = (i < n) && pred(i);
        uint64_t mask = __ballot(v);
        uint32_t lane_rank = (uint32_t)__popcll(mask & lanemask_lt());
        if (lane_id() == 0) wcnt[j & 1][w] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < kCpThreads / kWave; ++k) {
            uint32_t c = wcnt[j & 1][k];
            if (k < w) woff += c;
            tot += c;
        }
        if (v) emit(i, running + woff + lane_rank);
        running += tot;
    }
}

}  // namespace gm
