// k_sort.hip -- stable LSD radix sort of (key, value) pairs, ONE launch per pass.
//
// Utility for the binning steps of the path: search-grid cell keys (stands in
// for the FLANN kd-tree build, /root/reference src/tunnel_processing.cpp:62-70)
// and, on the fallback path, voxel keys (pcl::VoxelGrid's std::sort,
// src/tunnel_processing.cpp:217-220).  Stable on purpose: the order of points
// inside a cell fixes the fp32 summation order downstream, so results are
// bit-reproducible run to run.
//
// A pass is one kernel (k_rs_pass), a block per tile of 8192 consecutive positions:
//   * the digit TOTALS of every pass are known before the first pass starts -- they do not depend on the order of
//     the keys: the crop's emit step counts the first pass's as it produces the keys (k_crop.hip), the first pass those
//     of the passes after it as it reads the keys; any other caller runs k_rs_hist_all (one read for all passes);
//   * a block counts its tile's digits per wave in LDS, publishes the tile's count of every digit as one 64-bit
//     record [epoch:30 | state:2 | count:32] and looks back over the records of the tiles before it, one digit per
//     thread, 16 records in flight (decoupled look-back, as gm_compact.hpp does for one counter): the tile's items of
//     digit d start at (totals scanned over the digits)[d] + (tiles before)[d];
//   * the items are ranked stably (wave ballot "match" inside a wave, the per-wave counts across waves: position
//     order), put into digit order in LDS and written out as runs -- consecutive threads, consecutive slots;
//   * the LAST pass of the cell sort does not write (key, value) pairs: the value is the cropped index, the pass
//     fetches that row and writes the sorted cloud itself (GATHER), so no gather pass follows the sort.
// Tiles are handed out by ticket, so a block only waits for tiles whose blocks are already running, whatever else
// shares the chip (see gm_compact.hpp).  Records are never cleared: the epoch of a launch makes older records read as
// "not there yet".
//
// Round 3 ran two launches per pass (per-block histogram rows, then a scatter whose blocks summed the rows of the blocks
// before them: 204 thin blocks, up to 25 dependent L2 round trips for the last one) and three 11-bit passes + a gather
// launch: 3 x (6.6 + 19.5) + 28.5 us on the 1 M-point frame.  See DESIGN.md par. 4 for what this one measures.
#include <stdlib.h>

#include "gm_internal.hpp"

namespace gm {

// Block shape: 1024 threads (tile 8192, 115 KB of LDS) or 512 (tile 4096, 60 KB).  The large one is the faster pass when a
// frame runs alone (half the tiles: every tile's look-back fits one window); the small one slots in beside the blocks of
// other frames' kernels (k_normals: 256 threads, 40 KB) when several frames are in flight -- launch_radix_sort chooses.
constexpr int kRsItems = 8;                       // keys a lane holds
constexpr int kRsMinTile = 512 * kRsItems;        // the record arrays are laid out for the small tile
constexpr int kRsMaxBits = 9;                     // two threads per digit (the halves of the look-back)
constexpr int kRsMaxPasses = 4;
constexpr int kRsTotalsStride = 2048;             // words per pass in the totals array (gm_api.hip zero-fills them)
constexpr int kRsWindow = 112;                    // tiles (at least) whose 16-bit rows a block sums directly; one load round covers threads / (bins / 8) tiles

#ifdef GM_SORT_TIMELINE   // diagnostic builds (tools/sort_timeline.py): 100 MHz ticks of every block's phases, [pass][tile][8]
__device__ unsigned long long gm_sort_tl[4][256][8];
#define GM_ST_STAMP(k) do { if (threadIdx.x == 0 && tl_tile < 256u) gm_sort_tl[tl_pass][tl_tile][k] = wall_clock64(); } while (0)
#else
#define GM_ST_STAMP(k) do {} while (0)
#endif

SortPlan radix_plan(int key_bits)
{
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    SortPlan p;
    p.passes = (key_bits + kRsMaxBits - 1) / kRsMaxBits;
    p.bits = (key_bits + p.passes - 1) / p.passes;   // spread the bits evenly
    if (p.bits < 8) p.bits = 8;
    return p;
}

uint32_t radix_tiles(uint32_t n_cap, uint32_t tile = kRsMinTile) { return (n_cap + tile - 1) / tile; }
// record words of one pass: per tile a row of 16-bit counts and a row of 32-bit inclusive prefixes
static size_t radix_pass_words(uint32_t n_cap, int bits) { return (size_t)radix_tiles(n_cap) * ((size_t)1 << bits) * 3 / 2; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 16-byte loads / stores that other XCDs' blocks see / have made (sc1: served by / written through to memory, not this
// XCD's L2).  Records travel 16 bytes at a time: MI355X_MICROARCH.md prices a dword sc1 store at 6x a dwordx4 store per
// byte, and handed-off bytes arrive at 60-70 GB/s per block whatever the access size.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rec_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load16_agent(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void store16_agent(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset, u32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_offset, 0, 16 /* sc1 */);
}
size_t radix_totals_bytes() { return sizeof(uint32_t) * kRsMaxPasses * kRsTotalsStride; }
// record words [pass][tile][digit] of a sort over n_cap positions (cleared before its first pass), and of a slot
size_t radix_record_words(uint32_t n_cap, int key_bits)
{
    const SortPlan p = radix_plan(key_bits);
    return (size_t)p.passes * radix_pass_words(n_cap, p.bits);
}
size_t radix_record_words_max(uint32_t n_cap) { return (size_t)kRsMaxPasses * (radix_pass_words(n_cap, kRsMaxBits) + 1024); }

// digit totals of every pass in one read of the keys (callers whose keys do not come out of the crop)
__global__ __launch_bounds__(1024) void k_rs_hist_all(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ n_ptr,
                                                      SortPlan plan, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t h[kRsMaxPasses << kRsMaxBits];
    const uint32_t n = *n_ptr;
    const uint32_t bins = 1u << plan.bits;
    for (uint32_t k = threadIdx.x; k < (uint32_t)plan.passes * bins; k += blockDim.x) h[k] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t key = keys[i];
        for (int p = 0; p < plan.passes; ++p) atomicAdd(&h[(uint32_t)p * bins + ((key >> (p * plan.bits)) & (bins - 1u))], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < (uint32_t)plan.passes * bins; k += blockDim.x) {
        const uint32_t c = h[k];
        if (c) atomicAdd(&totals[(k / bins) * (uint32_t)kRsTotalsStride + (k % bins)], c);
    }
}

// One pass.  GATHER: the values are indices into rows_in; instead of (keys_out, vals_out) the pass writes keys_out and
// rows_out[dst] = {rows_in[value].xyz, bits(value)} -- the cropped cloud in cell-sorted order.
//
// What the phases of a block cost was measured per block (tools/sort_timeline.py, diagnostic build) on the 1 M-point
// frame (102 tiles, all resident at once).  First version: ticket 1.3 us | keys loaded + counted 1.2 | published 0.8 |
// look-back 4.5 | ranked + staged 6.0 (nine ballots per item: ~115 vector instructions, the four waves of a SIMD share
// its issue) | written 1.2, + 8 when the pass gathers the rows.  Three things follow:
//  (1) ranking: the lanes of a wave that hold the same digit find each other through a 64-bit lane mask per (wave,
//      digit) in LDS -- one ds_or, one read, two mbcnt -- instead of BITS ballots (6.0 -> 4.1 us for rank + stage);
//  (2) the look-back is not bound by its trips or its volume but by the HOP: a record stored at agent scope is visible
//      to a load of another XCD ~3 us later (7 trips of 16 records, 1 trip of 80 and a two-level scheme with 15 + 6 rows
//      measured 4.5, 4.5 and 6.3 us after the last needed record was out).  So the tile's counts go out as early as
//      possible -- counted with plain LDS atomics right after the keys arrive -- and everything that needs only LOCAL
//      offsets (ranking, staging the tile in digit order) runs while they travel; the look-back comes last, in front of
//      the write-out, when the records of the other tiles have long arrived;
//  (3) what a block reads is what the look-back costs once the records are there (handed-off bytes arrive at 60-70 GB/s
//      per block: 44 dword records per thread, 180 KB for the last tiles, took 5 us).  A tile's counts are <= 8192: they
//      travel as 16-bit records [ready:1 | count:14], eight digits per 16-byte load, the 16 x 64 threads of the block
//      taking 16 tiles per load round -- the 101 tiles before the last one of a 1 M-point frame are 101 KB, one trip.
//      Sorts of more than 112 tiles add, per tile, a row of 32-bit INCLUSIVE prefixes: a tile sums the 16-bit rows of the
//      112 tiles before it and the inclusive row of the tile before those.  All records are cleared when the frame opens.
template <int BITS, bool GATHER, int kRsThreads>
__global__ __launch_bounds__(kRsThreads) void k_rs_pass(const uint32_t *__restrict__ keys_in,
                                                        const uint32_t *__restrict__ vals_in,
                                                        uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                        const uint32_t *__restrict__ n_ptr, int shift,
                                                        const uint32_t *__restrict__ totals, uint32_t *__restrict__ rec,
                                                        uint32_t *__restrict__ ticket,
                                                        const float4 *__restrict__ rows_in, float4 *__restrict__ rows_out,
                                                        uint32_t *__restrict__ totals_later, int count_later)
{
    constexpr int BINS = 1 << BITS;
    constexpr int kRsWaves = kRsThreads / kWave, kRsTile = kRsThreads * kRsItems;
    static_assert(BITS == 8 || BITS == 9, "8 digits per 16-byte record load");
    static_assert(BINS <= kRsThreads, "a thread per digit");
    __shared__ uint32_t cw[kRsWaves][BINS];            // per wave: digit count of the tile, then its next local slot
    // the tile in digit order (keys, then values); before that, the lane masks of the ranking: [wave][digit] 64-bit
    __shared__ __attribute__((aligned(16))) uint32_t stage[2 * kRsTile];
    static_assert(sizeof(unsigned long long) * kRsWaves * BINS <= sizeof(uint32_t) * 2 * kRsTile, "lane masks fit the staging image");
    __shared__ uint32_t lstart[BINS], gpos[BINS];      // per digit: first slot in the image / in the output
    constexpr int OCTS = BINS / 8, GROUPS = kRsThreads / OCTS, LOADS = (kRsWindow + GROUPS - 1) / GROUPS;   // (16-byte record loads a thread holds in flight)
    constexpr uint32_t WINDOW = GROUPS * LOADS;        // tiles whose 16-bit rows a block sums directly (112 or 128)
    __shared__ __attribute__((aligned(16))) uint32_t lbq[GROUPS][OCTS][4];   // look-back: the groups' packed partial sums; also the rows being published
    __shared__ uint32_t wsum_b[kRsWaves], wsum_t[kRsWaves];
    __shared__ uint32_t s_tile, s_more;
    // count_later > 0 (the first pass of a sort whose pass-0 totals came from the crop): this pass reads every key anyway
    // and counts the digits of the count_later passes after it -- their totals do not depend on the order of the keys --
    // so the crop's emit step counts one digit per survivor instead of all of them
    __shared__ uint32_t later[kRsMaxPasses - 1][BINS];
    uint32_t *const skey = stage, *const sval = stage + kRsTile;
    unsigned long long *const lmask = reinterpret_cast<unsigned long long *>(stage);
#ifdef GM_SORT_TIMELINE
    const unsigned long long tl_entry = wall_clock64();
    uint32_t tl_tile = 0xFFFFFFFFu;
    const uint32_t tl_pass = (uint32_t)(shift / BITS) & 3u;
#endif
    const uint32_t n = *n_ptr;
    if (n == 0) return;   // (no ticket is taken: the word stays 0)
    const uint32_t ntiles = (n + (uint32_t)kRsTile - 1u) / (uint32_t)kRsTile;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(ticket, 1u);
        if (t == gridDim.x - 1) atomicExch(ticket, 0u);   // every ticket of this launch has been taken
        s_tile = t;
    }
    const int dig = threadIdx.x & (BINS - 1), half = threadIdx.x >> BITS;   // half 0 owns the digit; halves 0 and 1 look back
    const bool has_digit = half == 0;
    const uint32_t tot = has_digit ? totals[dig] : 0u;   // (does not depend on the tile: in flight behind the ticket)
    for (int k = threadIdx.x; k < kRsWaves * BINS; k += kRsThreads) { (&cw[0][0])[k] = 0; lmask[k] = 0ull; }
    for (int k = threadIdx.x; k < count_later * BINS; k += kRsThreads) (&later[0][0])[k] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= ntiles) return;   // uniform per block; nothing after the end is ever looked at
#ifdef GM_SORT_TIMELINE
    tl_tile = tile;
    if (threadIdx.x == 0 && tile < 256u) gm_sort_tl[tl_pass][tile][0] = tl_entry;
#endif
    GM_ST_STAMP(1);   // ticket in hand
    const int w = threadIdx.x / kWave, lane = lane_id();
    const uint32_t base = tile * (uint32_t)kRsTile;
    // wave w owns positions [wbase, wbase + 512): item u, lane l <-> wbase + 64 u + l (position order = wave, item, lane)
    const uint32_t wbase = base + (uint32_t)w * (uint32_t)(kRsItems * kWave);
    uint32_t kk[kRsItems], vv[kRsItems];
#pragma unroll
    for (int u = 0; u < kRsItems; ++u) {
        const uint32_t i = wbase + (uint32_t)u * kWave + lane;
        kk[u] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
        vv[u] = (i < n && vals_in) ? vals_in[i] : i;   // first pass: value = index
    }
#pragma unroll
    for (int u = 0; u < kRsItems; ++u) {
        const uint32_t i = wbase + (uint32_t)u * kWave + lane;
        if (i < n) atomicAdd(&cw[w][(kk[u] >> shift) & (BINS - 1)], 1u);
    }
    if (count_later) {   // uniform
#pragma unroll
        for (int u = 0; u < kRsItems; ++u) {
            const uint32_t i = wbase + (uint32_t)u * kWave + lane;
            if (i < n)
                for (int q = 0; q < count_later; ++q) atomicAdd(&later[q][(kk[u] >> (shift + (q + 1) * BITS)) & (BINS - 1)], 1u);
        }
    }
    if (threadIdx.x == 0) s_more = 0u;
    __syncthreads();
    GM_ST_STAMP(2);   // keys loaded and counted
    // ---- per digit (thread d of half 0): the tile's count; the row of 16-bit records [ready | count] goes out at once,
    // 16 bytes per lane
    const uint32_t tiles_all = gridDim.x;
    const __amdgpu_buffer_rsrc_t r_agg = rec_rsrc(rec, tiles_all * (uint32_t)BINS * 2u);
    uint32_t *const inc_rows = rec + (size_t)tiles_all * (BINS / 2);
    uint32_t bcount = 0;
    if (has_digit) {
#pragma unroll
        for (int ww = 0; ww < kRsWaves; ++ww) bcount += cw[ww][dig];
        reinterpret_cast<uint16_t *>(&lbq[0][0][0])[dig] = (uint16_t)(0x8000u | bcount);
    }
    const uint32_t inc_b = wave_inclusive_scan(bcount), inc_t = wave_inclusive_scan(tot);
    if (lane == kWave - 1) { wsum_b[w] = inc_b; wsum_t[w] = inc_t; }
    __syncthreads();
    if ((int)threadIdx.x < OCTS)
        store16_agent(r_agg, (tile * (uint32_t)BINS + 8u * threadIdx.x) * 2u, reinterpret_cast<const u32x4 *>(&lbq[0][0][0])[threadIdx.x]);
    GM_ST_STAMP(3);   // published
    uint32_t gb = inc_t - tot;   // exclusive scan of the digit totals: where the digit starts in the whole output
    {
        uint32_t ls = inc_b - bcount;   // ... and in the tile's image
#pragma unroll
        for (int k = 0; k < kRsWaves; ++k) if (k < w) { ls += wsum_b[k]; gb += wsum_t[k]; }
        if (has_digit) {
            lstart[dig] = ls;
            uint32_t slot = ls;
#pragma unroll
            for (int ww = 0; ww < kRsWaves; ++ww) { const uint32_t cq = cw[ww][dig]; cw[ww][dig] = slot; slot += cq; }
        }
    }
    // ---- rank inside the wave (no barrier needed: the lane masks are the wave's own).  Item u of a lane: its digit's
    // lane mask collects the lanes of this round that hold the digit (ds_or), rank = lanes below mine in it; the lowest
    // lane clears the mask for the next round.  (LDS operations of one wave execute in order.)
    uint32_t info[kRsItems / 2];   // per item 16 bits: rank | group size << 8
#pragma unroll
    for (int u = 0; u < kRsItems / 2; ++u) info[u] = 0u;
    const unsigned long long my_bit = 1ull << lane;
#pragma unroll
    for (int u = 0; u < kRsItems; ++u) {
        const uint32_t i = wbase + (uint32_t)u * kWave + lane;
        const bool valid = i < n;
        const uint32_t d = (kk[u] >> shift) & (BINS - 1);
        unsigned long long *mk = &lmask[(uint32_t)w * BINS + d];
        if (valid) atomicOr(mk, my_bit);
        wave_lds_fence();
        const unsigned long long peers = valid ? *mk : my_bit;
        wave_lds_fence();
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        const uint32_t cnt = (uint32_t)__popcll(peers);
        if (valid && rank == 0) *mk = 0ull;
        info[u >> 1] |= (rank | (cnt << 8)) << (16 * (u & 1));
    }
    __syncthreads();   // the lane masks are done with (the image takes their place); the local slots above are written
    GM_ST_STAMP(4);   // ranked
    // ---- stage: the group's lowest lane takes the group's slots off the wave's counter; everybody reads the counter back
#pragma unroll
    for (int u = 0; u < kRsItems; ++u) {
        const uint32_t i = wbase + (uint32_t)u * kWave + lane;
        const bool valid = i < n;
        const uint32_t d = (kk[u] >> shift) & (BINS - 1);
        const uint32_t inf = (info[u >> 1] >> (16 * (u & 1))) & 0xFFFFu, rank = inf & 0xFFu, cnt = inf >> 8;
        if (valid && rank == 0) atomicAdd(&cw[w][d], cnt);
        wave_lds_fence();
        const uint32_t after = cw[w][d];
        wave_lds_fence();
        if (valid) { const uint32_t slot = after - cnt + rank; skey[slot] = kk[u]; sval[slot] = vv[u]; }
    }
    GM_ST_STAMP(5);   // staged
    // ---- the tiles before this one (their records have been travelling since before the ranking): thread (g, o) loads the
    // eight 16-bit counts of digits 8 o .. 8 o + 7 of the tiles at distances 1 + g, 1 + g + GROUPS, ...; tiles further than
    // WINDOW back come in through the inclusive row of tile - WINDOW - 1
    uint32_t before = 0;
    if (tile > 0) {   // uniform per block
        const int g = threadIdx.x / OCTS, o = threadIdx.x % OCTS;
        const bool anchored = tile > WINDOW;
        const uint32_t anchor = anchored ? tile - WINDOW - 1u : 0u;
        for (;;) {   // block-uniform trip count (s_more is 0 here)
            u32x4 a[LOADS];
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                const uint32_t dist = 1u + (uint32_t)g + (uint32_t)(GROUPS * j);
                // (tiles "before tile 0" count nothing and are always there)
                const u32x4 none = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                a[j] = dist <= tile ? load16_agent(r_agg, ((tile - dist) * (uint32_t)BINS + 8u * (uint32_t)o) * 2u) : none;
            }
            uint32_t inc_word = 0x80000000u;
            if (anchored && has_digit)
                inc_word = __hip_atomic_load(&inc_rows[(size_t)anchor * BINS + dig], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t all = 0x80008000u;
            u32x4 acc = {0u, 0u, 0u, 0u};   // packed 16-bit sums: <= 7 x 8192 each
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                all &= a[j].x & a[j].y & a[j].z & a[j].w;
                acc += a[j] & 0x3FFF3FFFu;
            }
            if (all != 0x80008000u || !(inc_word >> 31)) s_more = 1u;   // a record is not there yet: everybody looks again
            __syncthreads();
            const bool more = s_more != 0u;
            if (!more) {
                lbq[g][o][0] = acc.x; lbq[g][o][1] = acc.y; lbq[g][o][2] = acc.z; lbq[g][o][3] = acc.w;
                if (has_digit) before = inc_word & 0x7FFFFFFFu;
            }
            __syncthreads();   // (everybody has read s_more; the partial sums are written)
            if (threadIdx.x == 0) s_more = 0u;
            if (!more) break;
            __builtin_amdgcn_s_sleep(2);
            __syncthreads();   // (the reset precedes the next trip's sets)
        }
        if (has_digit) {
            const int oo = dig >> 3, kx = dig & 7;
#pragma unroll
            for (int gg = 0; gg < GROUPS; ++gg) before += (lbq[gg][oo][kx >> 1] >> (16 * (kx & 1))) & 0xFFFFu;
        }
    }
    // a later tile that lies more than WINDOW tiles on needs this tile's inclusive prefix
    if (tile + WINDOW + 1u < ntiles) {   // uniform per block
        __syncthreads();   // (the partial sums have been read)
        uint32_t *row = &lbq[0][0][0];
        if (has_digit) row[dig] = 0x80000000u | ((before + bcount) & 0x7FFFFFFFu);
        __syncthreads();
        if ((int)threadIdx.x < BINS / 4)
            store16_agent(rec_rsrc(inc_rows, tiles_all * (uint32_t)BINS * 4u), (tile * (uint32_t)BINS + 4u * threadIdx.x) * 4u,
                          reinterpret_cast<const u32x4 *>(row)[threadIdx.x]);
    }
    if (has_digit) gpos[dig] = gb + before;
    __syncthreads();   // the image and gpos are complete
    GM_ST_STAMP(6);   // looked back
    // ---- write out: consecutive threads, consecutive image slots
    const uint32_t staged = n - base < (uint32_t)kRsTile ? n - base : (uint32_t)kRsTile;
    if (GATHER) {
        uint32_t key[kRsItems], val[kRsItems];
        float4 p[kRsItems];
#pragma unroll
        for (int u = 0; u < kRsItems; ++u) {
            const uint32_t i = (uint32_t)u * kRsThreads + threadIdx.x;
            key[u] = i < staged ? skey[i] : 0u;
            val[u] = i < staged ? sval[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kRsItems; ++u) p[u] = rows_in[val[u] < n ? val[u] : 0u];   // all in flight together (index 0 is readable: n > 0)
#pragma unroll
        for (int u = 0; u < kRsItems; ++u) {
            const uint32_t i = (uint32_t)u * kRsThreads + threadIdx.x;
            if (i < staged) {
                const uint32_t d = (key[u] >> shift) & (BINS - 1);
                const uint32_t dst = gpos[d] + (i - lstart[d]);
                if (dst < n) {   // (always: a wrong prefix must never become an out-of-bounds store)
                    keys_out[dst] = key[u];
                    rows_out[dst] = make_float4(p[u].x, p[u].y, p[u].z, __uint_as_float(val[u]));   // cropped index rides in the pad lane
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < kRsItems; ++u) {
            const uint32_t i = (uint32_t)u * kRsThreads + threadIdx.x;
            if (i < staged) {
                const uint32_t key = skey[i];
                const uint32_t d = (key >> shift) & (BINS - 1);
                const uint32_t dst = gpos[d] + (i - lstart[d]);
                if (dst < n) { keys_out[dst] = key; vals_out[dst] = sval[i]; }   // (always, see above)
            }
        }
    }
    if (count_later) {   // (every thread passed the barriers above: the counts are complete)
        for (int k = threadIdx.x; k < count_later * BINS; k += kRsThreads) {
            const uint32_t cq = (&later[0][0])[k];
            if (cq) atomicAdd(&totals_later[(uint32_t)(k / BINS) * (uint32_t)kRsTotalsStride + (uint32_t)(k % BINS)], cq);
        }
    }
#ifdef GM_SORT_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GM_ST_STAMP(7);   // stores acknowledged
#endif
}

template <int BITS, int THREADS>
static void rs_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, const uint32_t *n_ptr,
                    int shift, uint32_t nb, const uint32_t *totals, uint32_t *rec, uint32_t *ticket,
                    const float4 *rows_in, float4 *rows_out, uint32_t *totals_later, int count_later, hipStream_t s)
{
    if (rows_out)
        hipLaunchKernelGGL((k_rs_pass<BITS, true, THREADS>), dim3(nb), dim3(THREADS), 0, s, kin, vin, kout, vout, n_ptr, shift, totals,
                           rec, ticket, rows_in, rows_out, totals_later, count_later);
    else
        hipLaunchKernelGGL((k_rs_pass<BITS, false, THREADS>), dim3(nb), dim3(THREADS), 0, s, kin, vin, kout, vout, n_ptr, shift, totals,
                           rec, ticket, rows_in, rows_out, totals_later, count_later);
}

// Sorts (keys_a, index) by the low key_bits bits of the keys.  Returns 0 if the sorted keys (and values) end in
// (keys_a, vals_a), 1 if in (keys_b, vals_b).  prepared: the digit totals of radix_plan(key_bits) are in sl.sort.totals
// (the crop counted them) and the records [0, radix_record_words(n_cap, key_bits)) are clear (the frame's opening
// zero-fill); otherwise both are done here.  rows_out != nullptr: the last pass writes rows_out[sorted position] =
// {rows_in[value].xyz, bits(value)} instead of the values.
int launch_radix_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, const uint32_t *n_ptr,
                      uint32_t n_cap, int key_bits, Slot &sl, bool prepared, hipStream_t s, const float4 *rows_in,
                      float4 *rows_out)
{
    if (n_cap == 0) return 0;
    const SortPlan plan = radix_plan(key_bits);
    // block shape (see the top of the file).  GM_SORT_THREADS=512|1024: experiments
    static const char *te = getenv("GM_SORT_THREADS");
    const int threads = te ? (atoi(te) >= 1024 ? 1024 : 512) : (sl.pipelined ? 512 : 1024);
    const uint32_t nb = radix_tiles(n_cap, (uint32_t)threads * kRsItems);
    uint32_t *totals = sl.sort.totals;
    if (!prepared) {
        (void)hipMemsetAsync(totals, 0, radix_totals_bytes(), s);
        (void)hipMemsetAsync(sl.sort.rec, 0, sizeof(uint32_t) * radix_record_words(n_cap, key_bits), s);
        uint32_t hb = (n_cap + 8191u) / 8192u;
        if (hb > 512u) hb = 512u;
        hipLaunchKernelGGL(k_rs_hist_all, dim3(hb ? hb : 1u), dim3(1024), 0, s, (const uint32_t *)keys_a, n_ptr, plan, totals);
    }
    const uint32_t *kin = keys_a, *vin = nullptr /* first pass: value = index */;
    uint32_t *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < plan.passes; ++p) {
        const int shift = p * plan.bits;
        const uint32_t *tot = totals + (size_t)p * kRsTotalsStride;
        uint32_t *rec = sl.sort.rec + (size_t)p * radix_pass_words(n_cap, plan.bits);
        const bool last = p + 1 == plan.passes;
        const float4 *ri = last ? rows_in : nullptr;
        float4 *ro = last ? rows_out : nullptr;
        // (prepared: the crop counted pass 0's digits only; pass 0 counts those of the passes after it)
        uint32_t *tl = totals + (size_t)(p + 1) * kRsTotalsStride;
        const int cl = (prepared && p == 0) ? plan.passes - 1 : 0;
        if (threads == 1024) {
            if (plan.bits == 8) rs_pass<8, 1024>(kin, vin, kout, vout, n_ptr, shift, nb, tot, rec, sl.sort.ticket, ri, ro, tl, cl, s);
            else rs_pass<9, 1024>(kin, vin, kout, vout, n_ptr, shift, nb, tot, rec, sl.sort.ticket, ri, ro, tl, cl, s);
        } else {
            if (plan.bits == 8) rs_pass<8, 512>(kin, vin, kout, vout, n_ptr, shift, nb, tot, rec, sl.sort.ticket, ri, ro, tl, cl, s);
            else rs_pass<9, 512>(kin, vin, kout, vout, n_ptr, shift, nb, tot, rec, sl.sort.ticket, ri, ro, tl, cl, s);
        }
        if (p == 0) { kin = keys_b; vin = vals_b; kout = keys_a; vout = vals_a; }
        else {
            const uint32_t *t = kin; kin = kout; kout = (uint32_t *)t;
            t = vin; vin = vout; vout = (uint32_t *)t;
        }
    }
    return (plan.passes & 1) ? 1 : 0;
}

}  // namespace gm

#ifdef GM_SORT_TIMELINE
// diagnostic builds only: [pass][tile][8] ticks (100 MHz) of the last cell sort
extern "C" int gm_debug_sort_timeline(unsigned long long *out)
{
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gm::gm_sort_tl), sizeof(unsigned long long) * 4 * 256 * 8);
}
#endif
