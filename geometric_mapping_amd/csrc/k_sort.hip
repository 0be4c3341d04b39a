// k_sort.hip -- stable LSD radix sort of (key, value) pairs, 8 bits per pass.
//
// Utility for the two binning steps of the path: search-grid cell keys (stands
// in for the FLANN kd-tree build, /root/reference src/tunnel_processing.cpp:62-70)
// and voxel keys (pcl::VoxelGrid's std::sort, src/tunnel_processing.cpp:217-220).
// Stable on purpose: the order of points inside a cell fixes the fp32 summation
// order downstream, so results are bit-reproducible run to run.
//
// Per pass: per-block digit histogram -> one global exclusive scan (digit-major)
// -> stable scatter.  Inside a block the rank of an item among equal digits is
// wave-ballot "match" (8 ballots) + a [waves][256] LDS table.  The element
// count is device-resident; blocks past the end exit after publishing zeros.
#include "gm_compact.hpp"
#include "gm_internal.hpp"

namespace gm {

constexpr int kRsThreads = 256;
constexpr int kRsItems = 16;
constexpr int kRsTile = kRsThreads * kRsItems;
constexpr int kRsBits = 8;
constexpr int kRsBins = 1 << kRsBits;
constexpr int kRsWaves = kRsThreads / kWave;

static inline uint32_t rs_blocks(uint32_t n) { return (n + kRsTile - 1) / kRsTile; }
uint32_t radix_hist_entries(uint32_t n_cap) { return kRsBins * (rs_blocks(n_cap) + 1); }

__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const uint32_t *__restrict__ keys,
                                                        const uint32_t *__restrict__ n_ptr, int shift,
                                                        uint32_t *__restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t h[kRsBins];
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * kRsTile;
    h[threadIdx.x] = 0;
    __syncthreads();
    if (base < n) {
#pragma unroll 4
        for (int j = 0; j < kRsItems; ++j) {
            uint32_t i = base + j * kRsThreads + threadIdx.x;
            if (i < n) atomicAdd(&h[(keys[i] >> shift) & (kRsBins - 1)], 1u);
        }
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in,
                                                           uint32_t *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out,
                                                           const uint32_t *__restrict__ n_ptr, int shift,
                                                           const uint32_t *__restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t base_of[kRsBins];
    __shared__ uint32_t wtab[kRsWaves][kRsBins];
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x * kRsTile;
    if (tile >= n) return;  // uniform per block
    const int w = threadIdx.x / kWave;
    base_of[threadIdx.x] = hist[threadIdx.x * nblocks + blockIdx.x];
    for (int j = 0; j < kRsItems; ++j) {
        const uint32_t i = tile + j * kRsThreads + threadIdx.x;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0xFFFFFFFFu;
        const uint32_t d = (key >> shift) & (kRsBins - 1);
        // lanes of this wave holding the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRsBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
#pragma unroll
        for (int k = 0; k < kRsBins / kWave; ++k) wtab[w][k * kWave + lane_id()] = 0;
        wave_lds_fence();
        if (valid && rank == 0) wtab[w][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        {   // thread t owns digit t: turn per-wave counts into per-wave global offsets
            uint32_t run = base_of[threadIdx.x];
#pragma unroll
            for (int k = 0; k < kRsWaves; ++k) {
                uint32_t c = wtab[k][threadIdx.x];
                wtab[k][threadIdx.x] = run;
                run += c;
            }
            base_of[threadIdx.x] = run;
        }
        __syncthreads();
        if (valid) {
            const uint32_t dst = wtab[w][d] + rank;
            keys_out[dst] = key;
            vals_out[dst] = vals_in ? vals_in[i] : i;
        }
        __syncthreads();
    }
}

int launch_radix_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b,
                      const uint32_t *n_ptr, uint32_t n_cap, int key_bits, SortScratch &sc, hipStream_t s)
{
    const uint32_t nb = rs_blocks(n_cap);
    if (nb == 0) return 0;
    int passes = (key_bits + kRsBits - 1) / kRsBits;
    if (passes < 1) passes = 1;
    uint32_t *kin = keys_a, *vin = nullptr /* first pass: value = index */, *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kRsBits;
        hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(kRsThreads), 0, s, (const uint32_t *)kin, n_ptr, shift, sc.hist, nb);
        launch_exclusive_scan(sc.hist, kRsBins * nb, nullptr, nullptr, s);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(kRsThreads), 0, s, (const uint32_t *)kin,
                           (const uint32_t *)vin, kout, vout, n_ptr, shift, (const uint32_t *)sc.hist, nb);
        // ping-pong
        if (p == 0) { kin = keys_b; vin = vals_b; kout = keys_a; vout = vals_a; }
        else { uint32_t *t = kin; kin = kout; kout = t; t = vin; vin = vout; vout = t; }
    }
    return (passes & 1) ? 1 : 0;
}

}  // namespace gm
