// gm_compact.hpp -- order-preserving stream compaction (count -> scan -> scatter).
//
// Used for the crop box (CropBox keeps survivors in input order,
// /root/reference src/tunnel_processing.cpp:42-46), the NaN-normal removal
// (src/tunnel_processing.cpp:74-85) and the voxel segment heads.  Output order
// is part of the observable result (/choppedCloud point order), so the scan is
// exact and deterministic: wave ballot + popcount for the rank inside a wave,
// a 4-entry LDS table for the rank of the wave inside the block, and for the
// rank of the block the sum of the per-block counts before it, which every
// scatter block adds up for itself (coalesced reads of a few KB from L2: there
// is no scan kernel between the two launches; block 0 also publishes the total).
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
// block_counts[nblocks] come from k_compact_count over the same grid (blocks past n wrote 0).
template <class Pred, class Emit>
__global__ __launch_bounds__(kCpThreads) void k_compact_scatter(Pred pred, Emit emit,
                                                                 const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                                 const uint32_t *__restrict__ block_counts,
                                                                 uint32_t nblocks, uint32_t *__restrict__ total_out,
                                                                 uint32_t *__restrict__ total_out2)
{
    __shared__ uint32_t wcnt[2][kCpThreads / kWave];
    __shared__ uint32_t wpre[2][kCpThreads / kWave];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t base = blockIdx.x * kCpTile;
    if (base >= n && blockIdx.x != 0) return;  // uniform per block (block 0 stays: it publishes the total)
    const int w = threadIdx.x / kWave;
    // this block's first output slot = number of survivors in the blocks before it; block 0 sums ALL blocks
    uint32_t before = 0, all = 0;
    {
        const uint32_t lim = blockIdx.x == 0 ? nblocks : blockIdx.x;
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < lim; i += kCpThreads) part += block_counts[i];
        const uint32_t ws = (uint32_t)wave_sum((unsigned long long)part);
        if (lane_id() == 0) wpre[0][w] = ws;
        __syncthreads();
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < kCpThreads / kWave; ++k) t += wpre[0][k];
        if (blockIdx.x == 0) all = t; else before = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (total_out) *total_out = all;
        if (total_out2) *total_out2 = all;
    }
    if (base >= n) return;
    uint32_t running = before;
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
