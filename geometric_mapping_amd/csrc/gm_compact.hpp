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
#ifndef GM_CPITEMS
#define GM_CPITEMS 8
#endif
constexpr int kCpItems = GM_CPITEMS;
constexpr int kCpTile = kCpThreads * kCpItems;  // points per tile
constexpr int kCpWaves = kCpThreads / kWave;

inline uint32_t compact_blocks(uint32_t n) { return (n + kCpTile - 1) / kCpTile; }  // tiles of n points
constexpr int kCpMinTile = 2048;   // no instantiation cuts finer: the records of a slot are sized by it
inline uint32_t compact_records(uint32_t n) { return (n + kCpMinTile - 1) / kCpMinTile; }
inline uint32_t compact_grid(uint32_t n) { return compact_blocks(n); }               // one block per tile

// Pred: typedef ... Payload (what the predicate has loaded and the emit step needs again: kept in registers, not re-read);
//       __device__ bool operator()(uint32_t i, Payload &p) const
// Emit: __device__ void operator()(uint32_t src, uint32_t dst, const Payload &p); static constexpr bool kHasFinish; when true,
//       __device__ void finish(uint32_t tile) is called by every thread of the block once per tile that held input,
//       after the tile's last emit (per-tile state lives in the functor; finish() resets it); static constexpr bool
//       kHasPrepare; when true, __device__ void prepare() is called by every thread as the block starts (block-shared
//       state of the emit step; a block barrier follows before the first emit)
// The element count is *n_ptr (device-resident) or n_host; the grid is compact_grid(capacity): a block per tile.  The number of survivors
// goes to total_out / total_out2 (either may be null) -- also when it is 0.
template <class Pred, class Emit, int THREADS = kCpThreads, int ITEMS = kCpItems>
__global__ __launch_bounds__(THREADS) void k_compact(Pred pred, Emit emit, const uint32_t *__restrict__ n_ptr,
                                                            uint32_t n_host, ScanState st,
                                                            uint32_t *__restrict__ total_out,
                                                            uint32_t *__restrict__ total_out2)
{
    static_assert(THREADS * ITEMS >= kCpMinTile, "tile below the size the record arrays are laid out for");
    constexpr uint64_t kAggregate = 1ull << 32, kInclusive = 2ull << 32;
    __shared__ uint32_t lb_sum[(THREADS / kWave)], lb_state[(THREADS / kWave)];
    __shared__ uint32_t wcnt[(THREADS / kWave)];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t ntiles = (n + (uint32_t)(THREADS * ITEMS) - 1u) / (uint32_t)(THREADS * ITEMS);
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
    if constexpr (Emit::kHasPrepare) emit.prepare();
#ifdef GM_CP_DIAG_NOTICKET   // timing diagnostic only (tools/build_variants.sh): what the ticket costs
    if (threadIdx.x == 0) s_tile = blockIdx.x;
#else
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(st.ticket, 1u);
        if (t == gridDim.x - 1) atomicExch(st.ticket, 0u);  // every ticket of this launch has been taken
        s_tile = t;
    }
#endif
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile < ntiles) {   // uniform per block; nothing after the end is ever looked at
        const uint32_t base = tile * (uint32_t)(THREADS * ITEMS);
        // A wave owns ITEMS * 64 CONSECUTIVE positions of the tile (item j of wave w: base + w * 64 ITEMS + 64 j + lane), so the
        // survivors of everything before item (w, j) are those of the waves before w -- one LDS word per wave -- plus those
        // of the wave's own earlier items, which it holds in scalar registers: the ranks need one barrier and (waves) LDS
        // reads per thread.  (Items striding the whole block, position = base + 512 j + thread, needed an
        // [items][waves] table and 2 x items x waves LDS reads per thread: 1 M-point crop 17.5 -> 15.9 us, 10 M-point
        // crop with 1024 x 8 tiles 88 -> 80 us.)
        uint64_t mask[ITEMS];  // wave-uniform
        typename Pred::Payload pay[ITEMS];
        const uint32_t wbase = base + (uint32_t)w * (uint32_t)(kWave * ITEMS);
        uint32_t wtotal = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t i = wbase + j * kWave + lane;
            const bool v = (i < n) && pred(i, pay[j]);
            mask[j] = __ballot(v);
            wtotal += (uint32_t)__popcll(mask[j]);
        }
        if (lane == 0) wcnt[w] = wtotal;
        __syncthreads();
        uint32_t total = 0, woff = 0;
#pragma unroll
        for (int k = 0; k < (THREADS / kWave); ++k) {
            const uint32_t c = wcnt[k];
            if (k < w) woff += c;
            total += c;
        }
        if (threadIdx.x == 0)
            __hip_atomic_store(&st.status[tile], tag | (tile == 0 ? kInclusive : kAggregate) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // ---- look back.  The whole block does, one record per thread (tile-1-t): the blocks of a 1 M-point frame all
        // run at once, so inclusive prefixes are rare when the walk starts and a 64-wide window would need several
        // dependent round trips through the fabric; 512 records per trip cover a 2 M-point span.
        uint32_t before = 0;
#ifdef GM_CP_DIAG_NOLOOKBACK   // timing diagnostic only, WRONG output (survivors of a tile written at the tile's own base): what the chain costs
        before = base;
        if (false) {
#else
        if (tile > 0) {
#endif
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
                for (int k = 0; k < (THREADS / kWave); ++k) {
                    if (!done && !retry) {
                        if (lb_state[k] == 2u) retry = true;
                        else { acc += lb_sum[k]; done = lb_state[k] == 1u; }
                    }
                }
                if (retry) { __builtin_amdgcn_s_sleep(2); continue; }
                before += acc;
                if (done) break;
                top -= THREADS;
            }
            if (threadIdx.x == 0)
                __hip_atomic_store(&st.status[tile], tag | kInclusive | (uint64_t)(before + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0 && tile == ntiles - 1) {  // the last tile knows the number of survivors
#ifdef GM_CP_DIAG_NOLOOKBACK
            if (total_out) *total_out = 0;    // (nothing downstream may look at the holes)
            if (total_out2) *total_out2 = 0;
#else
            if (total_out) *total_out = before + total;
            if (total_out2) *total_out2 = before + total;
#endif
        }
        uint32_t running = before + woff;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t i = wbase + j * kWave + lane;
            if ((mask[j] >> lane) & 1ull) emit(i, running + (uint32_t)__popcll(mask[j] & lanemask_lt()), pay[j]);
            running += (uint32_t)__popcll(mask[j]);
        }
        if constexpr (Emit::kHasFinish) emit.finish(tile);
    }
}

}  // namespace gm
