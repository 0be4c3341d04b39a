// gm_compact.hpp -- order-preserving stream compaction in ONE launch (chained scan, decoupled look-back).
//
// Used for the crop box (CropBox keeps survivors in input order,
// /root/reference src/tunnel_processing.cpp:42-46), the NaN-normal removal
// (src/tunnel_processing.cpp:74-85) and the voxel segment heads.  Output order
// is part of the observable result (/choppedCloud point order), so the ranks
// are exact and deterministic: wave ballot + popcount inside a wave, a small
// LDS table across the waves of a block, and across tiles a chained scan --
// a tile's block publishes the number of survivors of the tile as soon as it
// knows it (AGGREGATE), looks back over the records of the tiles before it until
// it meets one that already holds an inclusive prefix, and then publishes its
// own (INCLUSIVE).  The input is read once; there is no count pass and no scan
// kernel.
//
// Progress: tiles are handed out by an atomic ticket, so a block only ever waits for tiles whose blocks are already
// running -- whatever else shares the chip.  (Taking the tile from blockIdx instead saves the ticket's ~4 us of
// same-address atomics per launch and was measured; but then a resident block can wait for one that has no slot yet, and
// with several compactions in flight -- frames on other streams, other processes -- the waiting blocks of different
// launches could fill an XCD between them and keep each other's missing blocks out for good.  A hang is not a price.)
// The blocks are 512 threads wide to halve the number of tickets.
// A record is one 64-bit word [epoch:30 | state:2 | value:32] written and read with single agent-scope atomics; the
// epoch (a per-slot launch counter, never 0) makes the records of earlier launches read as "not there yet", so the array
// is never cleared (zeroed once at allocation).  The block that takes the last ticket resets the ticket word for the
// next launch on the stream.
#pragma once

#include "gm_device.hpp"

namespace gm {

struct NoPayload {};   // predicates that load nothing worth keeping

#ifndef GM_CPTHREADS
#define GM_CPTHREADS 512
#endif
constexpr int kCpThreads = GM_CPTHREADS;
constexpr int kCpItems = 8;
constexpr int kCpTile = kCpThreads * kCpItems;  // points per tile
constexpr int kCpWaves = kCpThreads / kWave;

inline uint32_t compact_blocks(uint32_t n) { return (n + kCpTile - 1) / kCpTile; }  // tiles of n points
inline uint32_t compact_grid(uint32_t n) { return compact_blocks(n); }               // one block per tile

// Pred: typedef ... Payload (what the predicate has loaded and the emit step needs again: kept in registers, not re-read);
//       __device__ bool operator()(uint32_t i, Payload &p) const
// Emit: __device__ void operator()(uint32_t src, uint32_t dst, const Payload &p); static constexpr bool kHasFinish; when true,
//       __device__ void finish(uint32_t tile) is called by every thread of the block once per tile that held input,
//       after the tile's last emit (per-tile state lives in the functor; finish() resets it)
// The element count is *n_ptr (device-resident) or n_host; the grid is compact_grid(capacity): a block per tile.  The number of survivors
// goes to total_out / total_out2 (either may be null) -- also when it is 0.
template <class Pred, class Emit>
__global__ __launch_bounds__(kCpThreads) void k_compact(Pred pred, Emit emit, const uint32_t *__restrict__ n_ptr,
                                                            uint32_t n_host, ScanState st,
                                                            uint32_t *__restrict__ total_out,
                                                            uint32_t *__restrict__ total_out2)
{
    constexpr uint64_t kAggregate = 1ull << 32, kInclusive = 2ull << 32;
    __shared__ uint32_t lb_sum[kCpWaves], lb_state[kCpWaves];
    __shared__ uint32_t wcnt[2][kCpItems][kCpWaves];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t ntiles = (n + (uint32_t)kCpTile - 1u) / (uint32_t)kCpTile;
    const int w = threadIdx.x / kWave, lane = lane_id();
    const uint32_t epoch = scan_epoch(st);
    const uint64_t tag = (uint64_t)epoch << 34;
    if (n == 0) {  // nothing to do but to say so (no ticket is taken: the word stays 0)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (total_out) *total_out = 0;
            if (total_out2) *total_out2 = 0;
        }
        return;
    }
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(st.ticket, 1u);
        if (t == gridDim.x - 1) atomicExch(st.ticket, 0u);  // every ticket of this launch has been taken
        s_tile = t;
    }
    __syncthreads();
    const int buf = 0;
    const uint32_t tile = s_tile;
    if (tile < ntiles) {   // uniform per block; nothing after the end is ever looked at
        const uint32_t base = tile * (uint32_t)kCpTile;
        uint64_t mask[kCpItems];  // wave-uniform
        typename Pred::Payload pay[kCpItems];
#pragma unroll
        for (int j = 0; j < kCpItems; ++j) {
            const uint32_t i = base + j * kCpThreads + threadIdx.x;
            const bool v = (i < n) && pred(i, pay[j]);
            mask[j] = __ballot(v);
            if (lane == 0) wcnt[buf][j][w] = (uint32_t)__popcll(mask[j]);
        }
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < kCpItems; ++j)
#pragma unroll
            for (int k = 0; k < kCpWaves; ++k) total += wcnt[buf][j][k];
        if (threadIdx.x == 0)
            __hip_atomic_store(&st.status[tile], tag | (tile == 0 ? kInclusive : kAggregate) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // ---- look back.  The whole block does, one record per thread (tile-1-t): the blocks of a 1 M-point frame all
        // run at once, so inclusive prefixes are rare when the walk starts and a 64-wide window would need several
        // dependent round trips through the fabric; 512 records per trip cover a 2 M-point span.
        uint32_t before = 0;
        if (tile > 0) {
            int32_t top = (int32_t)tile - 1;
            for (;;) {
                const int32_t idx = top - (int32_t)threadIdx.x;
                // (tiles "before tile 0" read as an inclusive prefix of 0: the walk always ends there at the latest)
                const uint64_t sv = idx >= 0 ? __hip_atomic_load(&st.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : (tag | kInclusive);
                const bool ready = (sv >> 34) == (uint64_t)epoch && ((sv >> 32) & 3u) != 0u;
                const bool incl = ready && ((sv >> 32) & 3u) == 2u;
                const uint64_t im = __ballot(incl), rm = __ballot(ready);
                // per wave: lanes up to and including its first inclusive record (all lanes if it has none)
                const int first = im ? (int)__builtin_ctzll(im) : kWave;
                const uint64_t need = first >= kWave - 1 ? ~0ull : ((2ull << first) - 1ull);
                const uint32_t part = wave_sum(lane <= first ? (uint32_t)sv : 0u);
                __syncthreads();  // (the previous trip's readers of lb_* are done)
                if (lane == 0) { lb_sum[w] = part; lb_state[w] = ((rm & need) != need) ? 2u : (im ? 1u : 0u); }
                __syncthreads();
                // walk the waves in order: stop at the first with an inclusive record; a wave before that point that
                // still misses a record means: look again
                uint32_t acc = 0;
                bool retry = false, done = false;
#pragma unroll
                for (int k = 0; k < kCpWaves; ++k) {
                    if (!done && !retry) {
                        if (lb_state[k] == 2u) retry = true;
                        else { acc += lb_sum[k]; done = lb_state[k] == 1u; }
                    }
                }
                if (retry) { __builtin_amdgcn_s_sleep(2); continue; }
                before += acc;
                if (done) break;
                top -= kCpThreads;
            }
            if (threadIdx.x == 0)
                __hip_atomic_store(&st.status[tile], tag | kInclusive | (uint64_t)(before + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0 && tile == ntiles - 1) {  // the last tile knows the number of survivors
            if (total_out) *total_out = before + total;
            if (total_out2) *total_out2 = before + total;
        }
        uint32_t running = before;
#pragma unroll
        for (int j = 0; j < kCpItems; ++j) {
            const uint32_t i = base + j * kCpThreads + threadIdx.x;
            uint32_t woff = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < kCpWaves; ++k) {
                const uint32_t c = wcnt[buf][j][k];
                if (k < w) woff += c;
                tot += c;
            }
            if ((mask[j] >> lane) & 1ull) emit(i, running + woff + (uint32_t)__popcll(mask[j] & lanemask_lt()), pay[j]);
            running += tot;
        }
        if constexpr (Emit::kHasFinish) emit.finish(tile);
    }
}

}  // namespace gm
