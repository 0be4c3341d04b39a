// k_sort.hip -- stable LSD radix sort of (key, value) pairs, up to 11 bits per pass.
//
// Utility for the binning steps of the path: search-grid cell keys (stands in
// for the FLANN kd-tree build, /root/reference src/tunnel_processing.cpp:62-70)
// and, on the fallback path, voxel keys (pcl::VoxelGrid's std::sort,
// src/tunnel_processing.cpp:217-220).  Stable on purpose: the order of points
// inside a cell fixes the fp32 summation order downstream, so results are
// bit-reproducible run to run.
//
// Per pass, two launches:
//   k_rs_hist    per-block digit histogram -> hist[block][digit] (plain stores) and
//                digit totals (one atomicAdd per block and digit, integer: order-free)
//   k_rs_scatter each block derives its own base offsets -- exclusive scan of the
//                digit totals (block-local, 2^bits entries) plus the sum of the
//                histogram rows of the blocks before it (coalesced row reads from
//                L2) -- then ranks its items stably: wave-ballot "match" for the rank
//                among equal digits inside a wave, a [waves][bins] LDS table across
//                waves, running per-digit bases across rounds.
//   (k_rs_scatter_staged: the same pass with the items put into digit order in LDS first -- frames beyond ~1 M points)
// There is no global scan kernel: the block count is kept <= ~512 by growing the
// per-block tile with n, so the row sums stay a few tens of MB of L2 traffic.
// The element count is device-resident; blocks past the end publish zero rows.
#include "gm_internal.hpp"

namespace gm {

#ifndef GM_RSTHREADS
#define GM_RSTHREADS 512
#endif
constexpr int kRsThreads = GM_RSTHREADS;   // 512-thread blocks slot in beside other frames' k_normals blocks sooner than 1024-thread ones: -2 % step time
constexpr int kRsWaves = kRsThreads / kWave;  // 8
constexpr int kRsMaxBits = 11;
constexpr int kRsMaxPasses = 4;
constexpr int kRsBatch = 8;  // keys a lane keeps in registers at a time

static inline uint32_t rs_items(uint32_t n_cap)
{
    // items per thread (multiple of kRsBatch): tile = kRsThreads * items; aim for <= 256 blocks
    uint32_t items = (n_cap + 256u * kRsThreads - 1) / (256u * kRsThreads);
    items = (items + kRsBatch - 1) / kRsBatch * kRsBatch;
    return items < (uint32_t)kRsBatch ? (uint32_t)kRsBatch : items;
}
static inline uint32_t rs_blocks(uint32_t n_cap)
{
    const uint32_t tile = rs_items(n_cap) * kRsThreads;
    return (n_cap + tile - 1) / tile;
}
uint32_t radix_hist_entries(uint32_t n_cap)
{
    // Sized for EVERY element count n <= n_cap, not just for n_cap: rs_blocks() is not monotonic (the tile doubles
    // when items/thread steps up, so a 2.2 M-point capacity lays out 135 blocks where a 1.2 M-point frame lays out
    // 147).  rs_blocks(n) <= ceil(n / (kRsBatch * kRsThreads)) and <= 256 for every n.
    const uint64_t by_min_tile = ((uint64_t)n_cap + (uint64_t)kRsBatch * kRsThreads - 1) / ((uint64_t)kRsBatch * kRsThreads);
    const uint32_t nb_max = (uint32_t)(by_min_tile < 256u ? by_min_tile : 256u);
    return (1u << kRsMaxBits) * (nb_max + 1) + kRsMaxPasses * (1u << kRsMaxBits);
}

template <int BITS>
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const uint32_t *__restrict__ keys,
                                                        const uint32_t *__restrict__ n_ptr, int shift,
                                                        uint32_t items, uint32_t *__restrict__ hist,
                                                        uint32_t *__restrict__ totals)
{
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t h[BINS];
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * items * kRsThreads;
    for (int k = threadIdx.x; k < BINS; k += kRsThreads) h[k] = 0;
    __syncthreads();
    if (base < n) {
        for (uint32_t j0 = 0; j0 < items; j0 += kRsBatch) {
            uint32_t kk[kRsBatch];
#pragma unroll
            for (int u = 0; u < kRsBatch; ++u) {  // kRsBatch independent loads in flight
                const uint32_t i = base + (j0 + u) * kRsThreads + threadIdx.x;
                kk[u] = (i < n) ? keys[i] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < kRsBatch; ++u) {
                const uint32_t i = base + (j0 + u) * kRsThreads + threadIdx.x;
                if (i < n) atomicAdd(&h[(kk[u] >> shift) & (BINS - 1)], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < BINS; k += kRsThreads) {
        const uint32_t c = h[k];
        hist[(size_t)blockIdx.x * BINS + k] = c;
        if (c) atomicAdd(&totals[k], c);
    }
}

template <int BITS>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in,
                                                           uint32_t *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out,
                                                           const uint32_t *__restrict__ n_ptr, int shift,
                                                           uint32_t items, const uint32_t *__restrict__ hist,
                                                           const uint32_t *__restrict__ totals)
{
    constexpr int BINS = 1 << BITS;
    constexpr int PER = (BINS + kRsThreads - 1) / kRsThreads;  // digits owned by one thread (blocked)
    __shared__ uint32_t wtab[kRsWaves][BINS];  // per wave: digit count, then next output slot
    __shared__ uint32_t wsum[kRsWaves];
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x * items * kRsThreads;
    if (tile >= n) return;  // uniform per block
    const int w = threadIdx.x / kWave;
    const int lane = lane_id();
    // wave w owns the contiguous chunk [wbase, wbase + items*64) of the tile, so every
    // item of wave w precedes every item of wave w+1: ranks need no per-round barrier
    const uint32_t wbase = tile + (uint32_t)w * items * kWave;

    for (int k = threadIdx.x; k < kRsWaves * BINS; k += kRsThreads) (&wtab[0][0])[k] = 0;
    __syncthreads();
    for (uint32_t j0 = 0; j0 < items; j0 += kRsBatch) {
        uint32_t kk[kRsBatch];
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (j0 + u) * kWave + lane;
            kk[u] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (j0 + u) * kWave + lane;
            if (i < n) atomicAdd(&wtab[w][(kk[u] >> shift) & (BINS - 1)], 1u);
        }
    }
    // ---- start of this block's span for every digit: exclusive scan of the digit
    //      totals + histogram rows of the blocks before this one.  The rows are summed by ALL threads: thread t
    //      takes column t % BINS and every RG-th row starting at t / BINS (RG = threads / BINS row groups, 8 rows
    //      in flight each), the row groups are joined through LDS -- the serial chain of the last block is
    //      blocks / (8 RG) dependent steps instead of blocks / 8.
    __shared__ uint32_t colsum[kRsThreads];
    constexpr int RG = BINS < kRsThreads ? kRsThreads / BINS : 1;  // row groups (BINS is a power of two)
    if (BINS < kRsThreads) {
        const int col = threadIdx.x % BINS, rg = threadIdx.x / BINS;
        uint32_t acc = 0;
        uint32_t b = (uint32_t)rg;
        for (; b + 15u * RG < blockIdx.x; b += 16u * RG) {
            uint32_t r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = hist[(size_t)(b + (uint32_t)u * RG) * BINS + col];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += r[u];
        }
        for (; b < blockIdx.x; b += RG) acc += hist[(size_t)b * BINS + col];
        colsum[threadIdx.x] = acc;
    }
    __syncthreads();  // (also orders the wtab zeroing / counting above; harmless extra barrier)
    uint32_t tot[PER], before[PER];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int dig = threadIdx.x * PER + k;  // thread t owns digits [t*PER, t*PER+PER)
        tot[k] = dig < BINS ? totals[dig] : 0u;
        tsum += tot[k];
        before[k] = 0;
    }
    if (BINS < kRsThreads) {
        if ((int)threadIdx.x < BINS) {
            uint32_t a = 0;
#pragma unroll
            for (int g = 0; g < RG; ++g) a += colsum[g * BINS + threadIdx.x];
            before[0] = a;  // PER == 1 here
        }
    } else if (threadIdx.x * PER < BINS) {
        const uint32_t *col = hist + threadIdx.x * PER;
        uint32_t b = 0;
        // (the last block's chain of dependent L2 round trips is the launch's critical path: 32 rows in flight per step)
        constexpr int kInFlight = 32 / PER;
        for (; b + kInFlight <= blockIdx.x; b += kInFlight) {
            uint32_t r[kInFlight][PER];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u)
#pragma unroll
                for (int k = 0; k < PER; ++k) r[u][k] = col[(size_t)(b + u) * BINS + k];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u)
#pragma unroll
                for (int k = 0; k < PER; ++k) before[k] += r[u][k];
        }
        for (; b < blockIdx.x; ++b)
#pragma unroll
            for (int k = 0; k < PER; ++k) before[k] += col[(size_t)b * BINS + k];
    }
    const uint32_t inc = wave_inclusive_scan(tsum);
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();  // also: all per-wave digit counts are in wtab
    uint32_t run = inc - tsum;
#pragma unroll
    for (int k = 0; k < kRsWaves; ++k) if (k < w) run += wsum[k];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int dig = threadIdx.x * PER + k;
        if (dig < BINS) {
            uint32_t slot = run + before[k];
#pragma unroll
            for (int ww = 0; ww < kRsWaves; ++ww) {  // counts -> first output slot of each wave
                const uint32_t c = wtab[ww][dig];
                wtab[ww][dig] = slot;
                slot += c;
            }
        }
        run += tot[k];
    }
    __syncthreads();

    for (uint32_t j0 = 0; j0 < items; j0 += kRsBatch) {
        if (wbase + j0 * kWave >= n) break;  // wave-uniform: the rest of the chunk is past the end
        uint32_t kk[kRsBatch], vv[kRsBatch];
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (j0 + u) * kWave + lane;
            kk[u] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
            vv[u] = (i < n && vals_in) ? vals_in[i] : i;
        }
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (j0 + u) * kWave + lane;
            const bool valid = i < n;
            const uint32_t key = kk[u];
            const uint32_t d = (key >> shift) & (BINS - 1);
            // lanes of this wave holding the same digit
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
            uint32_t slot = 0;
            if (valid && rank == 0) {  // lowest lane of each digit group claims the group's slots
                slot = wtab[w][d];
                wtab[w][d] = slot + (uint32_t)__popcll(peers);
            }
            wave_lds_fence();
            slot = __shfl(slot, valid ? (int)__builtin_ctzll(peers) : lane, kWave);
            if (valid) {
                const uint32_t dst = slot + rank;
                keys_out[dst] = key;
                vals_out[dst] = vv[u];
            }
        }
    }
}

// The same pass with the block's items STAGED through LDS in digit order before they are written out.  Scattering keys
// straight from registers writes 4 bytes to 64 different places per wave: each costs a 32-byte sector (a 1 M-point pass
// moves 13 MB and takes the time of 107).  Here a block works through its tile in batches of 4096 consecutive positions:
// per-wave digit counts of the batch -> local slot of every (wave, digit) -> items ranked (stable: wave, round, lane =
// position order) into an LDS image sorted by digit -> written out with consecutive threads on consecutive slots, so
// that the items of one digit (8 on average with 512 bins) leave as one run.  BITS <= 9: one digit per thread.
template <int BITS>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter_staged(const uint32_t *__restrict__ keys_in,
                                                                  const uint32_t *__restrict__ vals_in,
                                                                  uint32_t *__restrict__ keys_out,
                                                                  uint32_t *__restrict__ vals_out,
                                                                  const uint32_t *__restrict__ n_ptr, int shift,
                                                                  uint32_t items, const uint32_t *__restrict__ hist,
                                                                  const uint32_t *__restrict__ totals)
{
    constexpr int BINS = 1 << BITS;
    static_assert(BINS <= kRsThreads, "one digit per thread");
    constexpr int BATCH = kRsBatch * kRsThreads;      // positions per batch
    constexpr int WCHUNK = kRsBatch * kWave;          // consecutive positions of one wave inside a batch
    constexpr int RG = kRsThreads / BINS;             // row groups of the histogram-row sum
    __shared__ uint32_t cw[kRsWaves][BINS];           // per wave: digit count of the batch, then its next local slot
    __shared__ uint32_t skey[BATCH], sval[BATCH];     // the batch in digit order
    __shared__ uint32_t lstart[BINS], gbase[BINS];    // per digit: first slot in the image / next slot in the output
    __shared__ uint32_t colsum[kRsThreads], wsum[kRsWaves];
    const uint32_t n = *n_ptr;
    const uint32_t tile = blockIdx.x * items * kRsThreads;
    if (tile >= n) return;  // uniform per block
    const int w = threadIdx.x / kWave, lane = lane_id();
    // ---- where this block's items of every digit start in the output: exclusive scan of the digit totals + the
    //      histogram rows of the blocks before this one (thread t: column t % BINS, every RG-th row from t / BINS)
    {
        const int col = threadIdx.x % BINS, rg = threadIdx.x / BINS;
        uint32_t acc = 0, b = (uint32_t)rg;
        for (; b + 31u * RG < blockIdx.x; b += 32u * RG) {   // 32 rows in flight per step
            uint32_t r[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) r[u] = hist[(size_t)(b + (uint32_t)u * RG) * BINS + col];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += r[u];
        }
        for (; b < blockIdx.x; b += RG) acc += hist[(size_t)b * BINS + col];
        colsum[threadIdx.x] = acc;
    }
    __syncthreads();
    {
        uint32_t tot = 0, before = 0;
        if ((int)threadIdx.x < BINS) {
            tot = totals[threadIdx.x];
#pragma unroll
            for (int g = 0; g < RG; ++g) before += colsum[g * BINS + threadIdx.x];
        }
        const uint32_t inc = wave_inclusive_scan(tot);
        if (lane == kWave - 1) wsum[w] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
#pragma unroll
        for (int k = 0; k < kRsWaves; ++k) if (k < w) run += wsum[k];
        if ((int)threadIdx.x < BINS) gbase[threadIdx.x] = run + before;
    }
    const uint32_t tile_end = tile + items * kRsThreads;
    for (uint32_t b0 = tile; b0 < tile_end && b0 < n; b0 += (uint32_t)BATCH) {   // uniform per block
        __syncthreads();   // the previous batch is written out (and gbase is set)
        for (int k = threadIdx.x; k < kRsWaves * BINS; k += kRsThreads) (&cw[0][0])[k] = 0;
        __syncthreads();
        // wave w owns positions [b0 + w * WCHUNK, + WCHUNK): round u, lane l <-> position + u * 64 + l
        uint32_t kk[kRsBatch], vv[kRsBatch];
        const uint32_t wbase = b0 + (uint32_t)w * WCHUNK;
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (uint32_t)u * kWave + lane;
            kk[u] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
            vv[u] = (i < n && vals_in) ? vals_in[i] : i;
        }
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (uint32_t)u * kWave + lane;
            if (i < n) atomicAdd(&cw[w][(kk[u] >> shift) & (BINS - 1)], 1u);
        }
        __syncthreads();
        // per digit: counts of the waves -> first local slot of every (wave, digit); the batch's count of the digit
        uint32_t bcount = 0;
        {
            uint32_t c[kRsWaves];
            if ((int)threadIdx.x < BINS) {
#pragma unroll
                for (int ww = 0; ww < kRsWaves; ++ww) { c[ww] = cw[ww][threadIdx.x]; bcount += c[ww]; }
            }
            const uint32_t inc = wave_inclusive_scan(bcount);
            if (lane == kWave - 1) wsum[w] = inc;
            __syncthreads();
            uint32_t ls = inc - bcount;
#pragma unroll
            for (int k = 0; k < kRsWaves; ++k) if (k < w) ls += wsum[k];
            if ((int)threadIdx.x < BINS) {
                lstart[threadIdx.x] = ls;
                uint32_t slot = ls;
#pragma unroll
                for (int ww = 0; ww < kRsWaves; ++ww) { cw[ww][threadIdx.x] = slot; slot += c[ww]; }
            }
        }
        __syncthreads();
        // rank (stable) and stage
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = wbase + (uint32_t)u * kWave + lane;
            const bool valid = i < n;
            const uint32_t d = (kk[u] >> shift) & (BINS - 1);
            uint64_t peers = __ballot(valid);   // lanes of this wave holding the same digit
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
            uint32_t slot = 0;
            if (valid && rank == 0) {  // lowest lane of each digit group claims the group's slots
                slot = cw[w][d];
                cw[w][d] = slot + (uint32_t)__popcll(peers);
            }
            wave_lds_fence();
            slot = __shfl(slot, valid ? (int)__builtin_ctzll(peers) : lane, kWave);
            if (valid) { skey[slot + rank] = kk[u]; sval[slot + rank] = vv[u]; }
        }
        __syncthreads();
        // write out: consecutive threads, consecutive image slots
        const uint32_t staged = n - b0 < (uint32_t)BATCH ? n - b0 : (uint32_t)BATCH;
#pragma unroll
        for (int u = 0; u < kRsBatch; ++u) {
            const uint32_t i = (uint32_t)u * kRsThreads + threadIdx.x;
            if (i < staged) {
                const uint32_t key = skey[i];
                const uint32_t d = (key >> shift) & (BINS - 1);
                const uint32_t dst = gbase[d] + (i - lstart[d]);
                keys_out[dst] = key;
                vals_out[dst] = sval[i];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < BINS) gbase[threadIdx.x] += bcount;
    }
}

template <int BITS>
static void rs_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, const uint32_t *n_ptr,
                    int shift, uint32_t items, uint32_t nb, uint32_t *hist, uint32_t *totals, hipStream_t s, bool staged)
{
    hipLaunchKernelGGL(k_rs_hist<BITS>, dim3(nb), dim3(kRsThreads), 0, s, kin, n_ptr, shift, items, hist, totals);
    if constexpr (BITS <= 9) {
        if (staged) {
            hipLaunchKernelGGL(k_rs_scatter_staged<BITS>, dim3(nb), dim3(kRsThreads), 0, s, kin, vin, kout, vout, n_ptr, shift,
                               items, (const uint32_t *)hist, (const uint32_t *)totals);
            return;
        }
    }
    hipLaunchKernelGGL(k_rs_scatter<BITS>, dim3(nb), dim3(kRsThreads), 0, s, kin, vin, kout, vout, n_ptr, shift, items,
                       (const uint32_t *)hist, (const uint32_t *)totals);
}

// scratch layout: [digit totals: kRsMaxPasses x 2^kRsMaxBits words][histogram rows: blocks x bins]
size_t radix_totals_bytes() { return sizeof(uint32_t) * kRsMaxPasses * (1u << kRsMaxBits); }

int launch_radix_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b,
                      const uint32_t *n_ptr, uint32_t n_cap, int key_bits, SortScratch &sc, bool totals_cleared,
                      hipStream_t s)
{
    const uint32_t nb = rs_blocks(n_cap);
    if (nb == 0) return 0;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    // The staged scatter pays when a block's tile is several batches long (frames beyond ~1 M points: the 10 M-point
    // frame sorts in 0.61 ms instead of 0.81, with four 8-bit passes instead of three 11-bit ones); on a one-batch tile its
    // barriers cost what the coalescing saves (1 M points: 0.099 against 0.102 ms alone, 1 % slower with frames in
    // flight).  GM_SORT_STAGED=0|1 forces either (tests).
    static const char *st = getenv("GM_SORT_STAGED");
    const bool staged = st ? atoi(st) != 0 : rs_items(n_cap) > (uint32_t)kRsBatch;
    const int max_bits = staged ? 9 : kRsMaxBits;   // (the staged scatter has one digit per thread)
    int passes = (key_bits + max_bits - 1) / max_bits;
    int bits = (key_bits + passes - 1) / passes;  // spread the bits evenly
    if (bits < 8) bits = 8;
    const uint32_t items = rs_items(n_cap);
    uint32_t *totals = sc.hist;  // fixed place: the frame's opening zero-fill clears it (totals_cleared)
    uint32_t *hist = sc.hist + (size_t)kRsMaxPasses * (1u << kRsMaxBits);
    if (!totals_cleared) hipMemsetAsync(totals, 0, radix_totals_bytes(), s);
    const uint32_t *kin = keys_a, *vin = nullptr /* first pass: value = index */;
    uint32_t *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * bits;
        uint32_t *tot = totals + (size_t)p * (1u << kRsMaxBits);
        switch (bits) {
        case 8: rs_pass<8>(kin, vin, kout, vout, n_ptr, shift, items, nb, hist, tot, s, staged); break;
        case 9: rs_pass<9>(kin, vin, kout, vout, n_ptr, shift, items, nb, hist, tot, s, staged); break;
        case 10: rs_pass<10>(kin, vin, kout, vout, n_ptr, shift, items, nb, hist, tot, s, staged); break;
        default: rs_pass<11>(kin, vin, kout, vout, n_ptr, shift, items, nb, hist, tot, s, staged); break;
        }
        if (p == 0) { kin = keys_b; vin = vals_b; kout = keys_a; vout = vals_a; }
        else {
            const uint32_t *t = kin; kin = kout; kout = (uint32_t *)t;
            t = vin; vin = vout; vout = (uint32_t *)t;
        }
    }
    return (passes & 1) ? 1 : 0;
}

}  // namespace gm
