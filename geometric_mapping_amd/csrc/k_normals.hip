// k_normals.hip -- fixed-radius neighbourhood covariance -> surface normal +
// curvature for every cropped point.  THE dominant kernel of the path.
//
// Replaces pcl::NormalEstimation::compute with a radius search
// (/root/reference src/tunnel_processing.cpp:58-70; PCL semantics restated in
// oracle/gm_oracle.c): neighbour set { j : fl32((dx*dx+dy*dy)+dz*dz) < fl32(r*r) }
// including the point itself, biased covariance, smallest eigenpair,
// curvature = |lambda0 / trace|, normal flipped towards the origin, NaN when
// fewer than 3 neighbours.
//
// MI355X design (not a kd-tree walk):
//  * points are binned into a uniform grid (y/z cell edge >= r, or r/D for frames with very many
//    neighbours per point; x binned 64x finer) by a stable radix sort of cell keys, x fastest, so the
//    neighbourhood of a run of cells in one x-row is (2D+1)^2 CONTIGUOUS, x-sorted ranges of the
//    sorted array (k_rows_and_tiles publishes every row's range and cuts the work items);
//  * a work item ("tile") is <= 64 consecutive sorted points of one x-row, one wave per tile, no
//    block barrier anywhere: the wave finds the tile's candidate windows (all 64 lanes probing one
//    range at a time: two rounds of loads per search), then streams them in chunks of 128
//    candidates through a private 10 KB LDS slice, the rows of the next chunk in flight in
//    registers while the current one is worked on;
//  * a chunk is staged as a feature-major bf16 image: every fp32 monomial {1, u, u u^T, |u|^2} of
//    the candidate's offset u from a tile origin cut exactly into three bf16 terms.  One image
//    feeds BOTH matrix products of the pair loop (v_mfma_f32_32x32x16_bf16): the squared distances
//    |u|^2 + |v|^2 - 2 u.v of 32 candidates x 32 queries (through gfx950's transposed LDS read),
//    turned into exact 0/1 weights by one clamped bf16 conversion -- pairs closer to the threshold
//    than the product's error bound are re-evaluated with FLANN's own fp32 chain, so the neighbour
//    sets are bit-exact -- and the ten moments sum_c G[f][c] W[c][q], whose weight operand is the
//    first product's result as it stands in the registers;
//  * thin neighbourhoods (sparse clouds) skip the image: offsets from the query itself, fp64 sums;
//  * the 3x3 solve runs in fp64 per query (closed-form root + two Newton steps, eigenvector =
//    largest cross product as pcl::eigen33 picks it); the epilogue also adds the point to the dense
//    voxel table (exact fixed-point sums: pcl::VoxelGrid, src/tunnel_processing.cpp:217-220);
//  * blocks retire after their four tiles: the hardware block scheduler balances the uneven
//    candidate counts and lets other frames' kernels interleave; consecutive runs of blocks are
//    dealt to one XCD so that a sorted row is fetched into one L2.
// Bound: instruction issue and the latency of a wave's own dependent steps (VALU staging + MFMA
// pair loop), not HBM: every candidate byte is read once per tile and reused by 64 queries
// (DESIGN.md par. 4 has the measured breakdown).  k_normals_valu / k_normals_m are the earlier
// all-VALU / moments-only formulations, kept for A/B measurements and cross-checks in tests.
#include <stdlib.h>
#include <string.h>

#include "gm_internal.hpp"

namespace gm {

typedef float v2f __attribute__((ext_vector_type(2)));

// products / sums that must NOT be contracted into an fma (the neighbour predicate)
__device__ __forceinline__ v2f pk_mul_rn(v2f a, v2f b)
{
    v2f r;
    r.x = __fmul_rn(a.x, b.x);
    r.y = __fmul_rn(a.y, b.y);
    return r;
}
__device__ __forceinline__ v2f pk_add_rn(v2f a, v2f b)
{
    v2f r;
    r.x = __fadd_rn(a.x, b.x);
    r.y = __fadd_rn(a.y, b.y);
    return r;
}

// 1.0 where d2 < r2 (strictly), else 0.0, for both halves in ONE VALU op: clamp01(fma(d2, -s, r2 * s)) with
// s = GridParams::r2_scale, a power of two with r2 * s ~ 2^100.  The fma rounds s * (r2 - d2) once, so its sign is
// exact and it is zero only when d2 == r2; any positive value is >= s * ulp(r2) / 2 >= 2^75 and clamps to 1, negative
// values and the overflow to -inf of far padding clamp to 0.  (v_cmp + v_cndmask per half costs
// four issue slots and a VALU->SGPR hazard.)
__device__ __forceinline__ v2f pk_within(v2f d2, v2f neg_big, v2f r2_big)
{
    v2f w;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(w) : "v"(d2), "v"(neg_big), "v"(r2_big));
    return w;
}

#ifndef GM_NRTHREADS
#define GM_NRTHREADS 256   // four tiles per block.  Measured on the 1 M frame with three frames in flight (tools/sweep_bench.sh),
                           // this kernel as it stands at the end of round 2: 64 / 128 / 256 / 512 threads = 3.73 / 3.83 / 3.99 /
                           // 3.89 G points/s (the kernel alone takes the same 0.165 ms with all but 512).  Its first matrix-core
                           // version, with longer dependent chains per wave, wanted 128 (0.326 ms per step against 0.374).
#endif
constexpr int kNrThreads = GM_NRTHREADS;
constexpr int kTileQ = kWave;             // queries per tile: one per lane
constexpr int kTileClasses = kTileListClasses;   // cost classes of the tile list (gm_device.hpp)
constexpr int kGroups = 4;                // lane groups with their own candidate window
constexpr int kGroupLanes = kWave / kGroups;
#ifndef GM_FOLD_TRIPS
#define GM_FOLD_TRIPS 16  // groups of four candidates between two folds of the fp32 partial sums into fp64
#endif
#ifndef GM_WINCAP
#define GM_WINCAP 320
#endif
constexpr int kWinCap = GM_WINCAP;              // candidates staged per chunk of one row range (4 x 16 B per lane in flight)
constexpr int kWinPad = 16;               // far-away padding behind a chunk (alignment + last partial group)
constexpr int kTileSpan = 3;              // max x extent of one tile, in (coarse) cell edges
constexpr int kNrWaves = kNrThreads / kWave;

// ---- per-row table and tiles ---------------------------------------------------------
// (The sorted cloud itself is written by the cell sort's last pass, k_sort.hip.)  A block covers kTbSpan sorted positions:
//  * it publishes the first / one-past-last sorted position of every occupied x-row (only occupied rows are ever read),
//  * it cuts tiles: <= 64 consecutive sorted points of one x-row spanning <= span+1 cells.  Tiles are cut at every 64th
//    point counted from the start of the x-row (so dense rows give full tiles: ~90 % lane fill) and a 64-chunk that
//    spans more than `span` cell steps (sparse rows) is cut again at aligned (span+1)-cell groups, which bounds the
//    candidate count of a tile: without the bound a sparse row yields tiles 20x the mean that the whole grid waits for.
// The start of a point's row is the nearest row start at or before it: inside the block a running maximum over the
// start flags; for the row that is already running when the block begins, the block walks the sorted keys backwards
// (kTbSpan positions per step) -- no table written by another block is read.
// A thread owns kTbPer CONSECUTIVE positions (one 16 B key load): the block's chain of barrier-separated steps
// (~10 us of latency) is paid once per 4096 positions and a 1 M-point frame is one round of resident blocks.
//
// Filing: every tile goes to the list of its cost class (below), and inside a class the tiles stand in POSITION order --
// whatever order the blocks run in: the rank of a tile among the block's tiles of its class comes from wave ballots and a
// per-wave table (thread order = position order), the number of tiles of the class in the blocks before from a chained
// scan over the blocks (one record per block and class, decoupled look-back as in gm_compact.hpp, a wave per class, 64
// records per trip; blocks take their span by ticket).  Round 3 filed by atomic arrival (rank from an LDS counter, base
// from one global atomic per block and class): k_normals deals runs of 128 consecutive list entries to one XCD so that
// neighbouring tiles share an L2, and with arrival order inside a class those runs were made of pieces from all over the
// frame: 124.7 -> 178.1 MB of HBM traffic per launch (profiles/r03_normals_pmc.json).  Position order is also the same
// from run to run.  A class other than the last holds tiles of >= 2 points, i.e. at most n / 2 of them: no list can
// overflow (Slot::tile_seg).
#ifndef GM_TBTHREADS
#define GM_TBTHREADS 1024
#endif
constexpr int kTbThreads = GM_TBTHREADS;
constexpr int kTbPer = 4;
constexpr int kTbSpan = kTbThreads * kTbPer;
static_assert(kTileClasses <= kTbThreads / kWave, "a wave of the cutter per cost class");
uint32_t tile_cutter_blocks(uint32_t n_cap) { return (n_cap + kTbSpan - 1) / kTbSpan; }
#ifdef GM_SORT_TIMELINE   // diagnostic builds (tools/sort_timeline.py): 100 MHz ticks of every cutter block's phases
__device__ unsigned long long gm_cut_tl[512][8];
#define GM_CT_STAMP(k) do { if (threadIdx.x == 0 && blk < 512u) gm_cut_tl[blk][k] = wall_clock64(); } while (0)
#else
#define GM_CT_STAMP(k) do {} while (0)
#endif
// x / d for a launch-invariant divisor: inv_d = a double just below 1 / d, so the truncated product is the quotient or one
// less (x < 2^32: the product's rounding error is far below 1 / d); ~1/4 of the instructions of the integer division
__device__ __forceinline__ uint32_t div_inv(uint32_t x, uint32_t d, double inv_d)
{
    uint32_t q = (uint32_t)((double)x * inv_d);
    if (x - q * d >= d) ++q;
    return q;
}

__global__ __launch_bounds__(kTbThreads) void k_rows_and_tiles(const uint32_t *__restrict__ skeys,
                                                               DevCounters *__restrict__ ctr, uint32_t nx, uint32_t span,
                                                               double inv_nx, double inv_group,
                                                               uint2 *__restrict__ row_bounds /* [ny*nz]: begin, end */, uint32_t nrows,
                                                               uint2 *__restrict__ tiles, uint32_t tiles_cap, uint32_t tile_seg,
                                                               ScanState st, uint32_t single_class)
{
    constexpr uint64_t kAggregate = 1ull << 32, kInclusive = 2ull << 32;
    __shared__ uint32_t wstart[kTbThreads / kWave];
    __shared__ uint32_t s_start0, s_tile;
    // the block's keys and the kTileQ keys on either side of them: every key a tile cut of this block looks at (the start
    // of a tile's 64-chunk lies < 64 positions before the tile, its end < 64 behind) -- the binary searches below run on
    // LDS.  (On the sorted array in memory they were chains of dependent L2 round trips, two searches of six steps for a
    // tile of a sparse row: the blocks with such rows finished 6 us after the others, and every block behind them in the
    // chained scan waited for their counts.)
    __shared__ uint32_t lk[kTbSpan + 2 * kTileQ];
    __shared__ uint32_t wcls[kTbThreads / kWave][kTileClasses];   // per wave: tiles of each class, then the wave's first rank
#ifdef GM_SORT_TIMELINE
    const unsigned long long tl_entry = wall_clock64();
#endif
    const uint32_t n = ctr->n_cropped;
    if (n == 0) return;   // (no ticket is taken: the word stays 0)
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(st.ticket, 1u);
        if (t == gridDim.x - 1) atomicExch(st.ticket, 0u);   // every ticket of this launch has been taken
        s_tile = t;
    }
    __syncthreads();
    const uint32_t blk = s_tile;
    const uint32_t base = blk * (uint32_t)kTbSpan;
    if (base >= n) return;  // uniform per block; nothing after the end is ever looked at
#ifdef GM_SORT_TIMELINE
    if (threadIdx.x == 0 && blk < 512u) gm_cut_tl[blk][0] = tl_entry;
#endif
    GM_CT_STAMP(1);   // ticket in hand
    const uint32_t nblk = (n + (uint32_t)kTbSpan - 1u) / (uint32_t)kTbSpan;
    const int w = threadIdx.x / kWave;
    const uint32_t s0 = base + threadIdx.x * (uint32_t)kTbPer;   // this thread's first position
    // keys of the thread's positions and of the two around them (the buffers hold whole 16 B groups: capacity is
    // allocated in multiples of 4 and the first position of a thread is a multiple of 4)
    uint32_t key[kTbPer + 2];
    {
        // (positions at or past n hold whatever the buffer holds: every use below is guarded by a position < n)
        const uint4 k4 = s0 < n ? *reinterpret_cast<const uint4 *>(skeys + s0) : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(&lk[kTileQ + threadIdx.x * kTbPer]) = k4;
        if (threadIdx.x < 2 * kTileQ / kTbPer) {   // the halos: 16 threads each
            const bool left = threadIdx.x < kTileQ / kTbPer;
            const uint32_t h = left ? threadIdx.x : threadIdx.x - kTileQ / kTbPer;
            const uint32_t pos = left ? base - (uint32_t)kTileQ + h * kTbPer : base + (uint32_t)kTbSpan + h * kTbPer;
            const bool ok = left ? base >= (uint32_t)kTileQ : pos < n;   // (base is a multiple of kTbSpan: all of the left halo or none)
            const uint4 h4 = ok ? *reinterpret_cast<const uint4 *>(skeys + pos) : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(&lk[left ? h * kTbPer : kTileQ + kTbSpan + h * kTbPer]) = h4;
        }
        __syncthreads();
        key[1] = k4.x; key[2] = k4.y; key[3] = k4.z; key[4] = k4.w;
        key[0] = (s0 > 0 && s0 < n) ? lk[kTileQ + threadIdx.x * kTbPer - 1] : 0u;
        key[kTbPer + 1] = s0 + kTbPer < n ? lk[kTileQ + threadIdx.x * kTbPer + kTbPer] : 0u;
    }
    const uint32_t *const lkeys = lk + kTileQ - base;   // lkeys[position] for positions in [base - 64, base + span + 64)
    // (The keys ascend.  Whether two of them lie in the same x-row, or in the same cell group of a row, is a comparison with
    // the row's / group's first key -- the cutter used to divide for each such question, inside the binary searches too, and
    // was bound by those instructions: the blocks of sparse rows, with many short tiles, published 6 us after the others.)
    uint32_t row[kTbPer], rbeg[kTbPer];   // x-row of the position, first key of that row
    bool starts[kTbPer];
    uint32_t m = 0;  // nearest row start at or before the thread's last position, +1; 0 = none in this thread
#pragma unroll
    for (int j = 0; j < kTbPer; ++j) {
        const uint32_t s = s0 + j;
        if (j == 0 || key[j + 1] >= rbeg[j - 1] + nx) { row[j] = div_inv(key[j + 1], nx, inv_nx); rbeg[j] = row[j] * nx; }
        else { row[j] = row[j - 1]; rbeg[j] = rbeg[j - 1]; }
        starts[j] = false;
        if (s < n && row[j] < nrows) {   // (a key is always inside the grid: no stray store, whatever the sort left)
            starts[j] = s == 0 || key[j] < rbeg[j];
            if (starts[j]) { row_bounds[row[j]].x = s; m = s + 1u; }
            if (s + 1 == n || key[j + 2] >= rbeg[j] + nx) row_bounds[row[j]].y = s + 1;
        }
    }
    GM_CT_STAMP(2);   // keys loaded, row table written
    // ---- start of the row that runs into this block
    if (threadIdx.x == 0) s_start0 = starts[0] ? base : 0xFFFFFFFFu;
    __syncthreads();
    {
        const uint32_t rbeg0 = div_inv(lkeys[base], nx, inv_nx) * nx;   // first key of the row that runs into the block
        for (uint32_t back = 0; s_start0 == 0xFFFFFFFFu; back += (uint32_t)kTbSpan) {  // block-uniform
            // positions base-back-4(tid+1) .. +3: the row start is the one position of row0 whose predecessor is not
            const uint32_t off = back + (threadIdx.x + 1u) * (uint32_t)kTbPer;
            uint32_t hit = 0xFFFFFFFFu;
            if (off <= base) {   // (base and off are multiples of 4: the group lies inside [0, base))
                const uint32_t t0 = base - off;
                const uint4 k4 = *reinterpret_cast<const uint4 *>(skeys + t0);
                const uint32_t kk[5] = {t0 ? skeys[t0 - 1] : 0u, k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int j = 0; j < kTbPer; ++j)   // (every key before `base` is <= the block's first: inside the row iff >= its first key)
                    if (kk[j + 1] >= rbeg0 && (t0 + j == 0 || kk[j] < rbeg0)) hit = t0 + j;
            }
            __syncthreads();            // everyone has read s_start0 for the loop test
            if (hit != 0xFFFFFFFFu) s_start0 = hit;   // at most one thread over the whole walk
            __syncthreads();
        }
    }
    GM_CT_STAMP(3);   // start of the row running into the block known
    // ---- exclusive running maximum of m over the threads before this one
    uint32_t mi = m;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = __shfl_up(mi, o, kWave);
        if (lane_id() >= o) mi = t > mi ? t : mi;
    }
    if (lane_id() == kWave - 1) wstart[w] = mi;
    uint32_t before = __shfl_up(mi, 1, kWave);
    if (lane_id() == 0) before = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTbThreads / kWave; ++k)
        if (k < w) before = wstart[k] > before ? wstart[k] : before;
    uint32_t row_start = before ? before - 1u : s_start0;
    // a point starts a tile iff it starts a 64-chunk of its x-row, or its chunk is "sparse"
    // (spans more than `span` cell steps) and it is the first point of an aligned cell group
    uint32_t cnt[kTbPer], tend[kTbPer], tcls[kTbPer];
    const uint32_t group = span + 1u;
#pragma unroll
    for (int j = 0; j < kTbPer; ++j) {
        const uint32_t s = s0 + j;
        cnt[j] = 0; tend[j] = 0; tcls[j] = 0;
        if (s < n) {
            if (starts[j]) row_start = s;
            const uint32_t kj = key[j + 1], rb = rbeg[j], rend = rb + nx;   // the row's keys: [rb, rend)
            const uint32_t cstart = row_start + ((s - row_start) / (uint32_t)kTileQ) * (uint32_t)kTileQ;
            // a tile can only start at the chunk start or where the cell group changes: everything else is skipped
            // before any further load
            const bool at_cstart = s == cstart;
            const uint32_t gbeg = rb + div_inv(kj - rb, group, inv_group) * group;   // first key of the point's cell group
            const bool group_edge = !at_cstart && key[j] < gbeg;
            if (at_cstart || group_edge) {
                // the chunk ends 64 points on or with the row (keys ascend: a short binary search in the row's last chunk)
                uint32_t cend = cstart + kTileQ;
                if (cend > n || lkeys[cend - 1] >= rend) {
                    uint32_t lo = s + 1, hi = cend < n ? cend : n;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (lkeys[mid] < rend) lo = mid + 1; else hi = mid;
                    }
                    cend = lo;
                }
                const uint32_t chunk_ext = lkeys[cend - 1] - lkeys[cstart];  // same row: key difference = cell steps
                const bool sparse = chunk_ext > span;
                if (at_cstart || sparse) {
                    cnt[j] = 1;
                    // the tile ends with its 64-chunk or, in a sparse chunk, where the next cell group begins
                    tend[j] = cend;
                    uint32_t ext = chunk_ext;
                    if (sparse) {
                        const uint32_t gend = gbeg + group;
                        uint32_t lo = s + 1, hi = cend;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (lkeys[mid] < gend) lo = mid + 1; else hi = mid;
                        }
                        tend[j] = lo;
                        ext = lkeys[lo - 1] - kj;
                    }
                    // cost class: a tile's candidates are those of windows 2 r + its own x extent long in every row around
                    // it -- their number follows the extent (correlation 0.99 on the 1 M-point frame), so the extent orders
                    // the tiles by cost: class 0 = longest
                    const uint32_t c8 = div_inv(ext * (uint32_t)kTileClasses, group, inv_group);
                    tcls[j] = (uint32_t)(kTileClasses - 1) - (c8 < (uint32_t)kTileClasses - 1u ? c8 : (uint32_t)kTileClasses - 1u);
                    // single_class: every tile in the last list (the one that can hold them all), i.e. plain position order --
                    // what a context that keeps several frames in flight wants (launch_grid_and_normals)
                    if (single_class) tcls[j] = (uint32_t)(kTileClasses - 1);
                }
            }
        }
    }
    GM_CT_STAMP(4);   // tiles cut and classified (thread 0's; the barrier below waits for everybody's)
    // ---- rank of every tile among the block's tiles of its class, in position order (thread, then item)
    uint32_t rank[kTbPer];
#pragma unroll
    for (int j = 0; j < kTbPer; ++j) rank[j] = 0;
    for (int c = 0; c < kTileClasses; ++c) {   // (wave-uniform loop: four ballots per class)
        uint32_t lt = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < kTbPer; ++j) {
            const uint64_t mk = __ballot(cnt[j] && tcls[j] == (uint32_t)c);
            lt += (uint32_t)__popcll(mk & lanemask_lt());
            tot += (uint32_t)__popcll(mk);
        }
        uint32_t in_thread = 0;
#pragma unroll
        for (int j = 0; j < kTbPer; ++j)
            if (cnt[j] && tcls[j] == (uint32_t)c) { rank[j] = lt + in_thread; ++in_thread; }
        if (lane_id() == 0) wcls[w][c] = tot;
    }
    __syncthreads();
    GM_CT_STAMP(5);   // ranked inside the block
    // ---- a wave per class: the waves' counts -> first rank of every wave, the block's count; then the blocks before
    if (w < kTileClasses) {
        const int c = w, lane = lane_id();
        const uint32_t mine = lane < kTbThreads / kWave ? wcls[lane][c] : 0u;
        const uint32_t inc = wave_inclusive_scan(mine);
        const uint32_t total = __shfl(inc, kWave - 1, kWave);
        const uint32_t epoch = scan_epoch(st);
        const uint64_t tag = (uint64_t)epoch << 34;
        unsigned long long *rec = st.status + c;   // record of (block b, this class): rec[b * kTileClasses]
        if (lane == 0)
            __hip_atomic_store(&rec[(size_t)blk * kTileClasses], tag | (blk == 0 ? kInclusive : kAggregate) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t prev = 0;
        if (blk > 0) {
            int32_t top = (int32_t)blk - 1;
            for (;;) {
                const int32_t idx = top - lane;
                // (blocks "before block 0" read as an inclusive prefix of 0: the walk always ends there at the latest)
                const uint64_t sv = idx >= 0 ? __hip_atomic_load(&rec[(size_t)idx * kTileClasses], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : (tag | kInclusive);
                const bool ready = (sv >> 34) == (uint64_t)epoch && ((sv >> 32) & 3u) != 0u;
                const bool incl = ready && ((sv >> 32) & 3u) == 2u;
                const uint64_t im = __ballot(incl), rm = __ballot(ready);
                const int first = im ? (int)__builtin_ctzll(im) : kWave;   // lanes up to and including the first inclusive record
                const uint64_t need = first >= kWave - 1 ? ~0ull : ((2ull << first) - 1ull);
                if ((rm & need) != need) { __builtin_amdgcn_s_sleep(1); continue; }
                prev += wave_sum(lane <= first ? (uint32_t)sv : 0u);
                if (im) break;
                top -= kWave;
            }
            if (lane == 0)
                __hip_atomic_store(&rec[(size_t)blk * kTileClasses], tag | kInclusive | (uint64_t)(prev + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < kTbThreads / kWave) wcls[lane][c] = prev + inc - mine;   // first list index of wave `lane` in class c
        if (lane == 0 && blk == nblk - 1) ctr->n_tiles_c[c][0] = prev + total;   // the last block knows the class's length
    }
    __syncthreads();
    GM_CT_STAMP(6);   // looked back
#pragma unroll
    for (int j = 0; j < kTbPer; ++j) {
        if (cnt[j]) {
            const uint2 t = make_uint2(s0 + j, (tend[j] - (s0 + j)) | (row[j] << 8));  // first query; number of queries (<= 64) | x-row << 8
            const uint32_t c = tcls[j], idx = wcls[w][c] + rank[j];
            if (idx < (c + 1u < (uint32_t)kTileClasses ? tile_seg : tiles_cap)) tiles[(size_t)c * tile_seg + idx] = t;   // (always: see Slot::tile_seg)
        }
    }
#ifdef GM_SORT_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GM_CT_STAMP(7);   // stores acknowledged
#endif
}

// fp64 reciprocal / reciprocal square root from the hardware estimate (v_rcp_f64 / v_rsq_f64) and one Newton step:
// ~2^-50 relative, 3-4 instructions where the IEEE division and sqrt sequences take ~30 each.  The epilogue below had five
// divisions and three square roots per query; its results are rounded to float (normals, curvature) or feed a Newton
// iteration, so correctly rounded quotients buy nothing.  x > 0 and normal everywhere they are used.
__device__ __forceinline__ double rcp_nr(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double rsq_nr(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x * r;
    return fma(fma(-h, r, 0.5), r, r);
}

// ---- smallest eigenpair of a symmetric PSD 3x3, fp64 ----------------------------
// c = {xx,xy,xz,yy,yz,zz}.  Root of the characteristic cubic: closed form with
// fp32 trigonometry as the starting point, two Newton steps in fp64; eigenvector
// = largest cross product of two rows of (C - lambda I) (pcl::eigen33's choice).
// Returns false when every cross product is exactly zero (PCL divides 0/0 there
// and the point is then removed as a NaN normal).
__device__ __forceinline__ bool smallest_eigpair(const double c[6], double &lam, double v[3])
{
    const double tr = c[0] + c[3] + c[5];
    const double m = tr * (1.0 / 3.0);
    const double k0 = c[0] - m, k3 = c[3] - m, k5 = c[5] - m;
    const double p = (k0 * k0 + k3 * k3 + k5 * k5 + 2.0 * (c[1] * c[1] + c[2] * c[2] + c[4] * c[4])) * (1.0 / 6.0);
    double l0 = m;
    if (p > 0.0) {
        const double q = 0.5 * (k0 * (k3 * k5 - c[4] * c[4]) - c[1] * (c[1] * k5 - c[4] * c[2]) +
                                c[2] * (c[1] * c[4] - k3 * c[2]));
        const double sp = p * rsq_nr(p);   // sqrt(p)
        double disc = p * p * p - q * q;
        if (disc < 0.0) disc = 0.0;
        const float phi = atan2f(disc > 0.0 ? (float)(disc * rsq_nr(disc)) : 0.0f, (float)q) * (1.0f / 3.0f);
        float sn, cs;
        __sincosf(phi, &sn, &cs);
        // smallest root of the three (phi in [0, pi/3])
        l0 = m - sp * ((double)cs + 1.7320508075688772 * (double)sn);
        // Newton on f(l) = det(C - l I)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double a = c[0] - l0, b = c[3] - l0, d = c[5] - l0;
            const double f = a * (b * d - c[4] * c[4]) - c[1] * (c[1] * d - c[4] * c[2]) + c[2] * (c[1] * c[4] - b * c[2]);
            const double fp = -(b * d + a * d + a * b - c[1] * c[1] - c[2] * c[2] - c[4] * c[4]);
            if (fabs(fp) > 1e-300) {
                const double step = f * rcp_nr(fp);
                // never step past the neighbouring root: |step| is bounded by the gap scale
                if (fabs(step) < sp) l0 -= step;
            }
        }
    }
    if (l0 < 0.0) l0 = 0.0;  // PSD matrix: a negative root is rounding (pcl::computeRoots does the same)
    lam = l0;
    const double r0[3] = {c[0] - l0, c[1], c[2]};
    const double r1[3] = {c[1], c[3] - l0, c[4]};
    const double r2[3] = {c[2], c[4], c[5] - l0};
    double v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    double v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    double v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double n1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const double n2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    const double n3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    double nn;
    if (n1 >= n2 && n1 >= n3) { v[0] = v1[0]; v[1] = v1[1]; v[2] = v1[2]; nn = n1; }
    else if (n2 >= n1 && n2 >= n3) { v[0] = v2[0]; v[1] = v2[1]; v[2] = v2[2]; nn = n2; }
    else { v[0] = v3[0]; v[1] = v3[1]; v[2] = v3[2]; nn = n3; }
    if (!(nn > 0.0)) return false;
    const double inv = rsq_nr(nn);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
    return true;
}

// ---- per-query epilogue shared by both formulations of the neighbourhood kernel ----
// mom = {n, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz}: moments of the neighbours' offsets from ANY fixed point (the
// covariance does not depend on it).  NormalEstimation::computePointNormal + flipNormalTowardsViewpoint(p, 0,0,0).
// Returns whether the point enters the voxel grid (finite normal and owned by this rank's slab).
__device__ __forceinline__ bool emit_normal(bool active, const float4 q, const double mom[10], const VoxDense &vd,
                                            float4 *__restrict__ normals4, int32_t *__restrict__ counts, uint32_t qn,
                                            unsigned long long stat_t0, unsigned long long stat_entry = 0ull)
{
    bool vox_ok = false;
    const int cnt = (int)mom[0];
    if (active) {
        const uint32_t dst = __float_as_uint(q.w);
        float4 out = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        if (cnt >= 3) {
            const double inv_n = rcp_nr((double)cnt);
            const double mx = mom[1] * inv_n, my = mom[2] * inv_n, mz = mom[3] * inv_n;
            double c[6];
            c[0] = mom[4] * inv_n - mx * mx; c[1] = mom[5] * inv_n - mx * my; c[2] = mom[6] * inv_n - mx * mz;
            c[3] = mom[7] * inv_n - my * my; c[4] = mom[8] * inv_n - my * mz; c[5] = mom[9] * inv_n - mz * mz;
            double lam, nv[3];
            if (smallest_eigpair(c, lam, nv)) {
                const double trc = c[0] + c[3] + c[5];
                const double curv = (trc != 0.0) ? fabs(lam * rcp_nr(trc)) : 0.0;
                const double ct = -((double)q.x * nv[0] + (double)q.y * nv[1] + (double)q.z * nv[2]);
                const double sgn = (ct < 0.0) ? -1.0 : 1.0;
                out = make_float4((float)(sgn * nv[0]), (float)(sgn * nv[1]), (float)(sgn * nv[2]), (float)curv);
                vox_ok = finite3(out.x, out.y, out.z) && q.x >= vd.own_lo && q.x < vd.own_hi;
            }
        }
        // The NaN-normal compaction keeps a point iff its stored normal is finite: a point this rank does not own (halo of a
        // slab: a neighbour, never an output) is therefore stored without one -- no separate validity array to scatter into.
        if (!vox_ok) out = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        normals4[dst] = out;
        if (counts) counts[dst] = cnt;
    }
    return vox_ok;
}

// ---- VoxelGrid fast path: points of a tile are spatial neighbours, so they fall into one to three voxels;
// reduce by voxel inside the wave, then one set of integer atomics per (wave, voxel).
// (pcl::VoxelGrid, src/tunnel_processing.cpp:217-220)
// wave-wide sum of a 64-bit value, delivered as a wave-uniform value: data-parallel-primitive adds (row shifts inside the
// 16-lane rows, then the row totals handed on) -- 12 moves + 6 add pairs on the vector pipe, no trip through the LDS crossbar
// as the lane-permute formulation of wave_sum takes (12 permutes per sum, three sums per voxel of a tile)
__device__ __forceinline__ unsigned long long wave_total_u64(unsigned long long v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto step = [&](auto dpp) {
        const uint32_t tl = dpp(lo), th = dpp(hi);
        const uint32_t nl = lo + tl;
        hi = hi + th + (nl < lo ? 1u : 0u);
        lo = nl;
    };
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true); });
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true); });
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true); });
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true); });
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false); });
    step([](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false); });
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63), rh = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)rh << 32) | rl;
}

__device__ __forceinline__ void voxel_sums(bool vox_ok, const float4 q, const VoxDense &vd, VoxCell *__restrict__ vox_table)
{
    const int lane = lane_id();
    uint32_t vkey = 0;
    unsigned long long fx = 0, fy = 0, fz = 0;
    if (vox_ok) {
        const int ix = (int)floorf(q.x * vd.inv_leaf) - vd.i_lo;
        const int iy = (int)floorf(q.y * vd.inv_leaf) - vd.i_lo;
        const int iz = (int)floorf(q.z * vd.inv_leaf) - vd.i_lo;
        vkey = (uint32_t)((iz * vd.dim + iy) * vd.dim + ix);
        fx = (unsigned long long)(((double)q.x - (double)vd.lo) * vd.scale + 0.5);
        fy = (unsigned long long)(((double)q.y - (double)vd.lo) * vd.scale + 0.5);
        fz = (unsigned long long)(((double)q.z - (double)vd.lo) * vd.scale + 0.5);
    }
    uint64_t remaining = __ballot(vox_ok);
    while (remaining) {  // wave-uniform loop
        const int leader = (int)__builtin_ctzll(remaining);
        const uint32_t k = __shfl(vkey, leader, kWave);
        const bool mine = vox_ok && vkey == k;
        const uint64_t same = __ballot(mine);
        const unsigned long long ax = wave_total_u64(mine ? fx : 0ull);
        const unsigned long long ay = wave_total_u64(mine ? fy : 0ull);
        const unsigned long long az = wave_total_u64(mine ? fz : 0ull);
        if (lane == leader) {
            VoxCell *cell = vox_table + k;
            atomicAdd(&cell->sx, ax);
            atomicAdd(&cell->sy, ay);
            atomicAdd(&cell->sz, az);
            atomicAdd(&cell->cnt, (uint32_t)__popcll(same));
        }
        remaining &= ~same;
    }
}

// Everything the tile bodies need (one struct: the kernel takes it by value)
struct NormalsArgs {
    const float4 *__restrict__ spts4;
    const uint32_t *__restrict__ skeys;
    const uint2 *__restrict__ tiles;
    DevCounters *__restrict__ ctr;
    GridParams g;
    uint32_t tiles_cap;   // entries of the last class's segment (any number of tiles a frame can have)
    uint32_t tile_seg;    // entries of every other class's segment
    const uint2 *__restrict__ row_bounds;
    float4 *__restrict__ normals4;
    int32_t *__restrict__ counts;
    VoxDense vd;
    VoxCell *__restrict__ vox_table;
};

// ---- one tile, all-VALU formulation ------------------------------------------------
// `lds` is THIS WAVE's slice of the block's LDS (the two formulations never share a slice across waves, so the waves of
// a block choose their formulation independently and no block barrier exists anywhere).
constexpr int kValuWaveLdsBytes = 3 * (kWinCap + kWinPad) * 4 + 10 * kWave * 8;
__device__ __forceinline__ void normals_tile_valu(const NormalsArgs &A, unsigned char *lds, const uint2 tile)
{
    const float4 *__restrict__ spts4 = A.spts4;
    const uint32_t *__restrict__ skeys = A.skeys;
    DevCounters *__restrict__ ctr = A.ctr;
    const GridParams &g = A.g;
    const uint2 *__restrict__ row_bounds = A.row_bounds;
    float4 *__restrict__ normals4 = A.normals4;
    int32_t *__restrict__ counts = A.counts;
    const VoxDense &vd = A.vd;
    VoxCell *__restrict__ vox_table = A.vox_table;
    (void)ctr;
    // candidate window, one per wave, SoA so that one broadcast ds_read_b128 feeds
    // the x (or y, z) of FOUR candidates to every lane; then the per-lane fp64 moment totals of the tile
    typedef float WinT[3][kWinCap + kWinPad];
    typedef double TotT[10][kWave];
    WinT *win = reinterpret_cast<WinT *>(lds);            // indexed [0] below: the slice is already this wave's
    TotT *tot = reinterpret_cast<TotT *>(lds + sizeof(WinT));
    const int lane = lane_id();
    const int w = 0;
    const uint32_t n = ctr->n_cropped;
    {
        const uint32_t qs = tile.x, qn = tile.y & 0xFFu;  // first query (sorted position), number of queries (1..64); the x-row rides above
#ifdef GM_NORMALS_TIMELINE
        const unsigned long long stat_t0 = wall_clock64();
#endif

        // this lane's query and its fine x cell (keys inside a tile are ascending: lanes are x-sorted)
        const int ql = lane;
        const bool active = (uint32_t)ql < qn;
        const uint32_t qidx = qs + (active ? (uint32_t)ql : qn - 1u);
        const float4 q = spts4[qidx];
        const uint32_t kl = skeys[qidx];
        // tile geometry: one x-row (lane 0 holds the tile's first query)
        const uint32_t ka = __builtin_amdgcn_readfirstlane(kl);
        const uint32_t row = ka / (uint32_t)g.nx;
        const int cy = (int)(row % (uint32_t)g.ny), cz = (int)(row / (uint32_t)g.ny);
        const int fxl = (int)(kl - row * (uint32_t)g.nx);
        // slab sharding: a tile made only of halo points (outside this rank's x range) produces no output
        if (__ballot(active && q.x >= vd.own_lo && q.x < vd.own_hi) == 0) {
            if (active) {
                const float nanv = __builtin_nanf("");
                normals4[__float_as_uint(q.w)] = make_float4(nanv, nanv, nanv, nanv);
                if (counts) counts[__float_as_uint(q.w)] = 0;
            }
            return;
        }

        // ---- candidate windows.  The 3x3 neighbouring x-rows are x-sorted runs of the sorted cloud.  The
        // wave is cut into kGroups lane groups (x-sorted, so each covers a short x interval); group gi only
        // needs the candidates of a row whose fine x cell lies within xreach of ITS interval.  All groups
        // walk their own window in lock-step (different LDS addresses, broadcast inside a group), so the
        // loop runs for the longest group window (~2.4 cell edges) instead of the tile's (~3.4).
        const int gi = lane / kGroupLanes;  // (inactive lanes repeat the tile's last query, so group intervals stay valid)
        // 9 rows x kGroups groups: lane i < 36 finds BOTH ends of the window of (row i / 4, group i % 4) by two
        // binary searches that run in the same loop (two independent loads in flight per step) and only inside
        // the row's own range [row_bounds.x, row_bounds.y): ~log2(points of the row) dependent steps per tile
        // instead of 2 x log2(n).  Unoccupied rows have row_bounds = (0, 0) (cleared per frame): empty window.
        uint32_t sb = 0, se = 0;
        {
            const int slot = lane < 9 * kGroups ? lane : 0;
            const int r = slot / kGroups, gg = slot % kGroups;
            // x interval of lane group gg (shuffles run with every lane active)
            const int lo_fx = __shfl(fxl, gg * kGroupLanes, kWave);
            const int hi_fx = __shfl(fxl, gg * kGroupLanes + kGroupLanes - 1, kWave);
            const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
            if (lane < 9 * kGroups && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                const uint32_t nrow = (uint32_t)(zz * g.ny + yy);
                const uint2 rb = row_bounds[nrow];
                const uint32_t rbk = nrow * (uint32_t)g.nx;
                const int xa = lo_fx > g.xreach ? lo_fx - g.xreach : 0;
                const int xb = hi_fx + g.xreach < g.nx - 1 ? hi_fx + g.xreach : g.nx - 1;
                const uint32_t key_b = rbk + (uint32_t)xa, key_e = rbk + (uint32_t)xb + 1u;
                uint32_t lo1 = rb.x, hi1 = rb.y, lo2 = rb.x, hi2 = rb.y;
                while (lo1 < hi1 || lo2 < hi2) {
                    const uint32_t m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                    const uint32_t k1 = skeys[m1 < n ? m1 : n - 1u], k2 = skeys[m2 < n ? m2 : n - 1u];
                    if (lo1 < hi1) { if (k1 < key_b) lo1 = m1 + 1u; else hi1 = m1; }
                    if (lo2 < hi2) { if (k2 < key_e) lo2 = m2 + 1u; else hi2 = m2; }
                }
                sb = lo1; se = lo2;
            }
        }

        // fp64 totals live in LDS (one 8-byte slot per lane and moment): 20 VGPRs less keeps 4 waves per SIMD
        double *T = &tot[w][0][lane];
#pragma unroll
        for (int k = 0; k < 10; ++k) T[k * kWave] = 0.0;
        const v2f qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
        const float big = g.r2_scale;  // power of two chosen by the host so that r2 * big ~ 2^100
        const v2f neg_big = {-big, -big}, r2_big = {g.r2 * big, g.r2 * big};  // exact scalings
        float *wx = &win[w][0][0], *wy = &win[w][1][0], *wz = &win[w][2][0];
        v2f sn = {0, 0}, sx = {0, 0}, sy = {0, 0}, sz = {0, 0}, sxx = {0, 0}, sxy = {0, 0}, sxz = {0, 0},
            syy = {0, 0}, syz = {0, 0}, szz = {0, 0};
        int since_fold = 0;
        auto fold = [&]() {  // fp32 partial sums -> fp64 totals
            // (the two halves are first joined in fp32: one more rounding of a <= 64-term partial sum, then one
            // conversion and one fp64 add per moment)
            T[0 * kWave] += (double)(sn.x + sn.y);
            T[1 * kWave] += (double)(sx.x + sx.y); T[2 * kWave] += (double)(sy.x + sy.y);
            T[3 * kWave] += (double)(sz.x + sz.y);
            T[4 * kWave] += (double)(sxx.x + sxx.y); T[5 * kWave] += (double)(sxy.x + sxy.y);
            T[6 * kWave] += (double)(sxz.x + sxz.y); T[7 * kWave] += (double)(syy.x + syy.y);
            T[8 * kWave] += (double)(syz.x + syz.y); T[9 * kWave] += (double)(szz.x + szz.y);
            sn = (v2f){0, 0}; sx = sn; sy = sn; sz = sn; sxx = sn; sxy = sn; sxz = sn; syy = sn; syz = sn; szz = sn;
            since_fold = 0;
        };

        // ---- chunk iterator over the 9 row ranges (row-wide range = begin of the lowest-x group .. end of
        // the highest-x group), software-pipelined: the global loads of chunk i+1 are in flight (in registers)
        // while chunk i is consumed from LDS
        auto row_begin = [&](int r) -> uint32_t { return __builtin_amdgcn_readlane(sb, r * kGroups); };
        auto row_end = [&](int r) -> uint32_t { return __builtin_amdgcn_readlane(se, r * kGroups + (kGroups - 1)); };
        int nr = 0;                       // next chunk: row index, start, length (0 = none left)
        uint32_t nc0 = 0, nlen = 0;
        auto seek = [&](int r, uint32_t c) {  // first non-empty chunk at or after (r, c)
            nlen = 0;
            while (r < 9) {
                const uint32_t e = row_end(r);
                if (c < e) { nr = r; nc0 = c; nlen = (e - c < (uint32_t)kWinCap) ? e - c : (uint32_t)kWinCap; return; }
                ++r;
                if (r < 9) c = row_begin(r);
            }
        };
        if (lane < kWinPad) { wx[kWinCap + lane] = 3.0e18f; wy[kWinCap + lane] = 3.0e18f; wz[kWinCap + lane] = 3.0e18f; }
        seek(0, row_begin(0));
        while (nlen) {
            const int r = nr;
            const uint32_t c0 = nc0, clen = nlen;
            // stage the chunk (coalesced 16 B loads) plus far-away padding right behind it
            wave_lds_fence();  // previous chunk fully consumed
#pragma unroll
            for (int k = 0; k < kWinCap / kWave; ++k) {
                const uint32_t i = (uint32_t)lane + (uint32_t)k * kWave;
                if (i < clen + kWinPad) {
                    float4 cp = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);  // never within r of anything
                    if (i < clen) cp = spts4[c0 + i];
                    wx[i] = cp.x; wy[i] = cp.y; wz[i] = cp.z;
                }
            }
            wave_lds_fence();
            if (c0 + clen < row_end(r)) seek(r, c0 + clen); else seek(r + 1, r + 1 < 9 ? row_begin(r + 1) : 0u);
            // my group's window inside this chunk (clamped), start aligned down to 4 for ds_read_b128
            const uint32_t mb = __shfl(sb, r * kGroups + gi, kWave), me = __shfl(se, r * kGroups + gi, kWave);
            uint32_t ob = mb > c0 ? mb - c0 : 0u, oe = me > c0 ? me - c0 : 0u;
            if (ob > clen) ob = clen;
            if (oe > clen) oe = clen;
            ob &= ~3u;
            const uint32_t mylen = oe > ob ? oe - ob : 0u;
            uint32_t maxlen = mylen;  // longest group window of the wave (wave-uniform trip count)
#pragma unroll
            for (int o = kGroupLanes; o < kWave; o <<= 1) {
                const uint32_t t2 = __shfl_xor(maxlen, o, kWave);
                maxlen = maxlen > t2 ? maxlen : t2;
            }
            maxlen = __builtin_amdgcn_readfirstlane(maxlen);
            const int iters = (int)((maxlen + 3u) >> 2);
#ifdef GM_NORMALS_STATS  // diagnostic build only (tools/normals_stats.py): candidate-stream accounting
            {
                const uint32_t sum_len = (uint32_t)wave_sum((unsigned long long)mylen) / kGroupLanes;
                if (lane == 0) {
                    atomicAdd(&ctr->pad[0], (uint32_t)(((iters + 1) & ~1) * 4));  // wave-candidates streamed
                    atomicAdd(&ctr->pad[1], sum_len);                              // sum of the 4 group windows
                    atomicAdd(&ctr->pad[2], clen);                                 // candidates staged
                    atomicAdd(&ctr->pad[3], 1u);                                   // chunks
                }
            }
#endif
            // a group whose window is shorter than the longest one keeps reading: first real candidates of
            // the row beyond its window (they fail the distance test), then the far padding behind the chunk
            const uint32_t lim = (clen + 3u) & ~3u;  // first all-padding group of four
            // offsets of four candidates from this lane's query: after this the candidate registers are dead
            struct Off4 { v2f dx[2], dy[2], dz[2]; };
            auto offsets = [&](const float4 cx4, const float4 cy4, const float4 cz4) {
                Off4 o;
                o.dx[0] = (v2f){cx4.x, cx4.y} - qx; o.dx[1] = (v2f){cx4.z, cx4.w} - qx;
                o.dy[0] = (v2f){cy4.x, cy4.y} - qy; o.dy[1] = (v2f){cy4.z, cy4.w} - qy;
                o.dz[0] = (v2f){cz4.x, cz4.y} - qz; o.dz[1] = (v2f){cz4.z, cz4.w} - qz;
                return o;
            };
            auto accumulate = [&](const Off4 &o) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f dx = o.dx[h], dy = o.dy[h], dz = o.dz[h];
                    // FLANN L2_Simple: every product and sum rounded, in this order
                    const v2f xx = pk_mul_rn(dx, dx), yy = pk_mul_rn(dy, dy), zz = pk_mul_rn(dz, dz);
                    const v2f d2 = pk_add_rn(pk_add_rn(xx, yy), zz);
                    // RadiusResultSet::addPoint: strict d2 < r2
                    const v2f wgt = pk_within(d2, neg_big, r2_big);
                    const v2f mx = wgt * dx, my = wgt * dy;
                    sn += wgt;
                    sx += mx; sy += my; sz += wgt * dz;
                    sxx += wgt * xx; syy += wgt * yy; szz += wgt * zz;  // the rounded squares of the predicate
                    sxy += mx * dy; sxz += mx * dz; syz += my * dz;
                }
            };
            // Two groups of four candidates per trip, software-pipelined with NO extra registers: a group's candidate
            // registers are dead once its six offset subtractions are done, so its LDS reads for the NEXT trip are
            // re-issued right there and have the rest of the trip (~300 cycles of arithmetic) to land, instead of
            // stalling the wave at the top of every trip.  (Reads past the window are clamped to `lim`: far-away
            // padding or real non-neighbours, harmless.)
            auto rd = [&](const float *base, uint32_t a) { return *reinterpret_cast<const float4 *>(base + (a < lim ? a : lim)); };
            uint32_t a = ob;
            float4 x0 = rd(wx, a), y0 = rd(wy, a), z0 = rd(wz, a);
            float4 x1 = rd(wx, a + 4u), y1 = rd(wy, a + 4u), z1 = rd(wz, a + 4u);
            for (int it = 0; it < iters; it += 2) {
                a += 8u;
                const Off4 oa = offsets(x0, y0, z0);
                x0 = rd(wx, a); y0 = rd(wy, a); z0 = rd(wz, a);
                __builtin_amdgcn_sched_barrier(0);  // keep the re-issue here
                accumulate(oa);
                const Off4 ob4 = offsets(x1, y1, z1);
                x1 = rd(wx, a + 4u); y1 = rd(wy, a + 4u); z1 = rd(wz, a + 4u);
                __builtin_amdgcn_sched_barrier(0);
                accumulate(ob4);
                since_fold += 2;
                if (since_fold >= GM_FOLD_TRIPS) fold();  // 16 trips = every 64 candidates (32 per packed half)
            }
        }
        fold();
        const double Sn = T[0 * kWave], Sx = T[1 * kWave], Sy = T[2 * kWave], Sz = T[3 * kWave], Sxx = T[4 * kWave],
                     Sxy = T[5 * kWave], Sxz = T[6 * kWave], Syy = T[7 * kWave], Syz = T[8 * kWave], Szz = T[9 * kWave];
        const double mom[10] = {Sn, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz};
#ifdef GM_NORMALS_TIMELINE
        const bool vox_ok = emit_normal(active, q, mom, vd, normals4, counts, qn, stat_t0);
#else
        const bool vox_ok = emit_normal(active, q, mom, vd, normals4, counts, qn, 0ull);
#endif
        if (vd.enabled) voxel_sums(vox_ok, q, vd, vox_table);
    }
}

// ---- the neighbourhood kernel, matrix-core formulation ---------------------------
// Same tiles, same grid, same neighbour predicate -- but the ten moments of a neighbourhood are a matrix product:
//     S[feature f][query q] = sum over candidates c of  G[f][c] * W[c][q],   W[c][q] = 1 if |c - q|^2 < r^2 else 0,
// with G the monomials {1, u, u u^T} of the candidate's offset u = c - o from ONE origin o per tile (the covariance
// does not depend on the origin; |u| < ~2r, so no cancellation).  W is exact in bf16; every fp32 feature value is cut
// into three bf16 terms (8 + 8 + 8 mantissa bits: the cut is exact), which makes 1 + 9 x 3 = 28 rows of a
// 32-row A operand, and v_mfma_f32_32x32x16_bf16 accumulates them in fp32 (measured on gfx950,
// tools/microbench/mfma_probe.hip: no truncation bias, error of a 640-term sum below that of an fp32 fma chain).
// The VALU keeps only the predicate (9 packed operations per two pairs, exactly FLANN's rounding) and one v_perm to
// pack two weights: 10 operations per two pairs instead of 21, the twelve moment operations run on the matrix pipe.
//   * a tile is two groups of 32 queries; in a group's loop lane l stands for query (l & 31) and candidate octet
//     (l >> 5): per MFMA a lane tests its query against 8 candidates and hands the 8 weights over as its B fragment;
//   * the candidates of a row range are staged once per tile: x, y, z as fp32 SoA (predicate) and the 32 feature
//     rows as [octet][row][8 candidates] bf16, 464 B per octet, so a lane's A fragment is one ds_read_b128;
//   * D comes out with the query on the lane: lanes l and l ^ 32 hold complementary feature rows of one query.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef GM_MXCHUNK
#define GM_MXCHUNK 128
#endif
#ifndef GM_MX_WAVES
#define GM_MX_WAVES 4   // waves per SIMD the register allocation aims for (with 128-candidate chunks LDS allows four blocks per CU)
#endif
constexpr int kMxChunk = GM_MXCHUNK;      // candidates staged per chunk of a row range (multiple of 16)
constexpr int kMxPad = 16;                // far-away padding behind a chunk (the last 16-candidate block of a group)
#ifndef GM_MXMINCAND
#define GM_MXMINCAND 64
#endif
constexpr int kMxMinCandidates = GM_MXMINCAND;  // tiles whose first group sees fewer candidates than this take the direct path
constexpr int kMxGroups = 2;              // query groups of a tile: 32 queries each (the N of the 32x32x16 MFMA)
constexpr int kMxGroupLanes = kWave / kMxGroups;
constexpr int kMxOctets = (kMxChunk + kMxPad) / 8;
[[maybe_unused]] constexpr int kMxRows = 28;  // feature rows per octet: 1 + 9 x 3.  The A operand's rows 28..31 read the pad row and the
                                          // next octet's first rows: whatever is there only reaches result rows nobody reads
constexpr int kMxOctetWords = 29 * 4;     // 464 B per octet: 116 dwords = 20 mod 32 banks, so the staging stores of the 16
                                          // octets a wave writes at once spread over all banks (448 B would be 8-way conflicts)
static_assert(kMxChunk % 16 == 0, "chunks are cut into 16-candidate MFMA steps");

__device__ __forceinline__ uint32_t pack_hi16(uint32_t hi_src, uint32_t lo_src)
{
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);  // (lo_src >> 16) | (hi_src & 0xFFFF0000)
}

// per wave: the candidate window as fp32 SoA, then the feature rows [octet][29 rows][8 bf16] (+ the last octet's
// over-read).  The prefetch of a group's last step reads 16 slots past its array: x into y, y into z, z into the
// feature rows -- inside this struct, values never used.
struct MxWaveLds {
    float w[3][kMxChunk + kMxPad];
    uint32_t f[kMxOctets * kMxOctetWords + 16];
};
constexpr int kMxWaveLdsBytes = (int)sizeof(MxWaveLds);

__device__ __forceinline__ void normals_tile_mx(const NormalsArgs &A, unsigned char *lds, const uint2 tile, uint32_t min_candidates)
{
    const float4 *__restrict__ spts4 = A.spts4;
    const uint32_t *__restrict__ skeys = A.skeys;
    DevCounters *__restrict__ ctr = A.ctr;
    const GridParams &g = A.g;
    const uint2 *__restrict__ row_bounds = A.row_bounds;
    float4 *__restrict__ normals4 = A.normals4;
    int32_t *__restrict__ counts = A.counts;
    const VoxDense &vd = A.vd;
    VoxCell *__restrict__ vox_table = A.vox_table;
    (void)ctr;
    MxWaveLds *mxl = reinterpret_cast<MxWaveLds *>(lds);   // this wave's slice
    const int lane = lane_id();
    const int w = 0;
    const uint32_t n = ctr->n_cropped;
    float *wx = &mxl[w].w[0][0], *wy = &mxl[w].w[1][0], *wz = &mxl[w].w[2][0];
    uint32_t *feat = &mxl[w].f[0];
    const int qsel = lane & 31, half = lane >> 5;
    {
        const uint32_t qs = tile.x, qn = tile.y & 0xFFu;
#ifdef GM_NORMALS_TIMELINE
        const unsigned long long stat_t0 = wall_clock64();
#else
        const unsigned long long stat_t0 = 0ull;
#endif
        const bool active = (uint32_t)lane < qn;
        const uint32_t qidx = qs + (active ? (uint32_t)lane : qn - 1u);
        const float4 q = spts4[qidx];
        const uint32_t kl = skeys[qidx];
        const uint32_t ka = __builtin_amdgcn_readfirstlane(kl);
        const uint32_t row = ka / (uint32_t)g.nx;
        const int cy = (int)(row % (uint32_t)g.ny), cz = (int)(row / (uint32_t)g.ny);
        const int fxl = (int)(kl - row * (uint32_t)g.nx);
        if (__ballot(active && q.x >= vd.own_lo && q.x < vd.own_hi) == 0) {  // halo-only tile (slab sharding)
            if (active) {
                const float nanv = __builtin_nanf("");
                normals4[__float_as_uint(q.w)] = make_float4(nanv, nanv, nanv, nanv);
                if (counts) counts[__float_as_uint(q.w)] = 0;
            }
            return;
        }
        // ---- candidate windows: lane i < 18 finds both ends of the window of (row i / 2, group i % 2)
        uint32_t sb = 0, se = 0;
        {
            const int slot = lane < 9 * kMxGroups ? lane : 0;
            const int r = slot / kMxGroups, gg = slot % kMxGroups;
            const int lo_fx = __shfl(fxl, gg * kMxGroupLanes, kWave);
            const int hi_fx = __shfl(fxl, gg * kMxGroupLanes + kMxGroupLanes - 1, kWave);
            const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
            if (lane < 9 * kMxGroups && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                const uint32_t nrow = (uint32_t)(zz * g.ny + yy);
                const uint2 rb = row_bounds[nrow];
                const uint32_t rbk = nrow * (uint32_t)g.nx;
                const int xa = lo_fx > g.xreach ? lo_fx - g.xreach : 0;
                const int xb = hi_fx + g.xreach < g.nx - 1 ? hi_fx + g.xreach : g.nx - 1;
                const uint32_t key_b = rbk + (uint32_t)xa, key_e = rbk + (uint32_t)xb + 1u;
                uint32_t lo1 = rb.x, hi1 = rb.y, lo2 = rb.x, hi2 = rb.y;
                while (lo1 < hi1 || lo2 < hi2) {
                    const uint32_t m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                    const uint32_t k1 = skeys[m1 < n ? m1 : n - 1u], k2 = skeys[m2 < n ? m2 : n - 1u];
                    if (lo1 < hi1) { if (k1 < key_b) lo1 = m1 + 1u; else hi1 = m1; }
                    if (lo2 < hi2) { if (k2 < key_e) lo2 = m2 + 1u; else hi2 = m2; }
                }
                sb = lo1; se = lo2;
            }
        }
        // Thin neighbourhoods (fewer than min_candidates candidates in the first group's windows) take the direct path
        // below: nothing to amortise the feature staging over, and the few-point covariances of a sparse cloud are
        // near-degenerate, where offsets from the query itself (|offset| < r) in fp64 keep more than offsets from a
        // tile origin through bf16 features do.
        bool thin;
        {
            uint32_t wl = (lane < 9 * kMxGroups && (lane % kMxGroups) == 0) ? se - sb : 0u;
            wl = wave_sum(wl);
            thin = wl < min_candidates;
        }
        // one origin per tile for the moment features: the tile's middle query (any point near the tile will do)
        const int mid = (int)(qn >> 1);
        // (readlane on the BIT PATTERN: the builtin is integer-typed, a float argument would be converted, i.e. truncated)
        const float ox = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.x), mid)),
                    oy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.y), mid)),
                    oz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.z), mid));
        const float big = g.r2_scale;
        const v2f neg_big = {-big, -big}, r2_big = {g.r2 * big, g.r2 * big};
        const int ngroups = qn > (uint32_t)kMxGroupLanes ? 2 : 1;  // (a second group of repeated queries is skipped)
        f32x16 acc[kMxGroups];

#pragma unroll
        for (int gi = 0; gi < kMxGroups; ++gi)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[gi][k] = 0.f;

        auto row_begin = [&](int r) -> uint32_t { return __builtin_amdgcn_readlane(sb, r * kMxGroups); };
        auto row_end = [&](int r) -> uint32_t { return __builtin_amdgcn_readlane(se, r * kMxGroups + ngroups - 1); };
        int nr = 0;
        uint32_t nc0 = 0, nlen = 0;
        auto seek = [&](int r, uint32_t c) {
            nlen = 0;
            while (r < 9) {
                const uint32_t e = row_end(r);
                if (c < e) { nr = r; nc0 = c; nlen = (e - c < (uint32_t)kMxChunk) ? e - c : (uint32_t)kMxChunk; return; }
                ++r;
                if (r < 9) c = row_begin(r);
            }
        };
        if (thin) {
            // direct path, self-contained: every lane tests its own query against every candidate of the tile's row
            // ranges (wave-uniform loads straight from the sorted cloud) and sums the offsets from the query in fp64
            double tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < 9; ++r) {
                const uint32_t e = row_end(r);
                for (uint32_t i = row_begin(r); i < e; ++i) {
                    const float4 c = spts4[i];
                    const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
                    // FLANN L2_Simple: every product and sum rounded, in this order; RadiusResultSet: strict d2 < r2
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    if (d2 < g.r2) {
                        const double ex = dx, ey = dy, ez = dz;
                        tm[0] += 1.0; tm[1] += ex; tm[2] += ey; tm[3] += ez;
                        tm[4] += ex * ex; tm[5] += ex * ey; tm[6] += ex * ez; tm[7] += ey * ey; tm[8] += ey * ez; tm[9] += ez * ez;
                    }
                }
            }
            const bool vox_ok = emit_normal(active, q, tm, vd, normals4, counts, qn, stat_t0);
            if (vd.enabled) voxel_sums(vox_ok, q, vd, vox_table);
            return;
        }
        seek(0, row_begin(0));
        while (nlen) {
            const int r = nr;
            const uint32_t c0 = nc0, clen = nlen;
            wave_lds_fence();  // previous chunk fully consumed
            // ---- stage the chunk: each lane takes candidate PAIRS (two bf16 of a feature row make one dword)
#pragma unroll
            for (int k = 0; k < (kMxChunk + kMxPad + 2 * kWave - 1) / (2 * kWave); ++k) {
                const uint32_t i = 2u * (uint32_t)lane + (uint32_t)k * 2u * kWave;
                if (i < clen + kMxPad && i < (uint32_t)(kMxChunk + kMxPad)) {
                    const bool va = i < clen, vb = i + 1u < clen;
                    float4 ca = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f), cb = ca;  // never within r of anything
                    if (va) ca = spts4[c0 + i];
                    if (vb) cb = spts4[c0 + i + 1u];
                    *reinterpret_cast<float2 *>(wx + i) = make_float2(ca.x, cb.x);
                    *reinterpret_cast<float2 *>(wy + i) = make_float2(ca.y, cb.y);
                    *reinterpret_cast<float2 *>(wz + i) = make_float2(ca.z, cb.z);
                    // features of the offsets from the tile origin (padding: all zero, never NaN/Inf in the A operand)
                    const float uxa = va ? ca.x - ox : 0.f, uya = va ? ca.y - oy : 0.f, uza = va ? ca.z - oz : 0.f;
                    const float uxb = vb ? cb.x - ox : 0.f, uyb = vb ? cb.y - oy : 0.f, uzb = vb ? cb.z - oz : 0.f;
                    const uint32_t oct = i >> 3;
                    uint32_t *fo = feat + oct * (uint32_t)kMxOctetWords + ((i & 7u) >> 1);
                    auto st = [&](int frow, uint32_t v) { fo[frow * 4] = v; };
                    auto split3 = [&](float a, float b, int frow) {  // exact: 8 + 8 + 8 mantissa bits
                        const uint32_t ah = __float_as_uint(a) & 0xFFFF0000u, bh = __float_as_uint(b) & 0xFFFF0000u;
                        st(frow, pack_hi16(bh, ah));
                        const float ra = a - __uint_as_float(ah), rb = b - __uint_as_float(bh);
                        const uint32_t am = __float_as_uint(ra) & 0xFFFF0000u, bm = __float_as_uint(rb) & 0xFFFF0000u;
                        st(frow + 1, pack_hi16(bm, am));
                        const float sa = ra - __uint_as_float(am), sb2 = rb - __uint_as_float(bm);
                        st(frow + 2, pack_hi16(__float_as_uint(sb2), __float_as_uint(sa)));
                    };
                    st(0, 0x3F803F80u);  // the count row: 1.0 | 1.0
                    split3(uxa, uxb, 1); split3(uya, uyb, 4); split3(uza, uzb, 7);
                    split3(uxa * uxa, uxb * uxb, 10); split3(uxa * uya, uxb * uyb, 13); split3(uxa * uza, uxb * uzb, 16);
                    split3(uya * uya, uyb * uyb, 19); split3(uya * uza, uyb * uzb, 22); split3(uza * uza, uzb * uzb, 25);
                }
            }
            wave_lds_fence();
            if (c0 + clen < row_end(r)) seek(r, c0 + clen); else seek(r + 1, r + 1 < 9 ? row_begin(r + 1) : 0u);
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) {
                if (gi >= ngroups) break;  // wave-uniform
                // this group's window inside the chunk, start aligned down to an octet
                const uint32_t mb = __builtin_amdgcn_readlane(sb, r * kMxGroups + gi),
                               me = __builtin_amdgcn_readlane(se, r * kMxGroups + gi);
                uint32_t ob = mb > c0 ? mb - c0 : 0u, oe = me > c0 ? me - c0 : 0u;
                if (ob > clen) ob = clen;
                if (oe > clen) oe = clen;
                ob &= ~7u;
                const int steps = oe > ob ? (int)((oe - ob + 15u) >> 4) : 0;  // wave-uniform: the MFMA needs every lane
#if defined(GM_NORMALS_STATS) && !defined(GM_MD_DEBUG)
                if (lane == 0) {
                    atomicAdd(&ctr->pad[0], (uint32_t)(steps * 16) / 2u);  // candidates streamed per query, in 64-query tile units
                    atomicAdd(&ctr->pad[1], (oe > ob ? oe - ob : 0u) * 2u);
                    if (gi == 0) { atomicAdd(&ctr->pad[2], clen); atomicAdd(&ctr->pad[3], 1u); }
                }
#endif
                // the query this lane stands for in this group's loop
                const float gqa = __shfl(q.x, gi * kMxGroupLanes + qsel, kWave), gqb = __shfl(q.y, gi * kMxGroupLanes + qsel, kWave),
                            gqc = __shfl(q.z, gi * kMxGroupLanes + qsel, kWave);
                const v2f qx = {gqa, gqa}, qy = {gqb, gqb}, qz = {gqc, gqc};
                // this lane's octet of the current step: candidates pw[0..7], feature rows at fa
                const float *pw = wx + ob + 8u * (uint32_t)half;
                const uint32_t *fa = feat + ((ob >> 3) + (uint32_t)half) * (uint32_t)kMxOctetWords + (uint32_t)qsel * 4u;
                constexpr int W = kMxChunk + kMxPad;  // x -> y -> z stride of the SoA window
                auto ld4 = [](const float *q4) { return *reinterpret_cast<const float4 *>(q4); };
                float4 x0 = ld4(pw), x1 = ld4(pw + 4), y0 = ld4(pw + W), y1 = ld4(pw + W + 4), z0 = ld4(pw + 2 * W),
                       z1 = ld4(pw + 2 * W + 4);
                for (int it = 0; it < steps; ++it) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(fa);  // consumed by the MFMA at the end of the step
                    // offsets from this lane's query: after these the candidate registers are dead, and the NEXT step's
                    // reads are issued right here, a whole step of arithmetic ahead of their use (the last step reads
                    // 16 slots past the window: inside the slack behind the padding, never used)
                    const v2f dx0 = (v2f){x0.x, x0.y} - qx, dx1 = (v2f){x0.z, x0.w} - qx, dx2 = (v2f){x1.x, x1.y} - qx, dx3 = (v2f){x1.z, x1.w} - qx;
                    const v2f dy0 = (v2f){y0.x, y0.y} - qy, dy1 = (v2f){y0.z, y0.w} - qy, dy2 = (v2f){y1.x, y1.y} - qy, dy3 = (v2f){y1.z, y1.w} - qy;
                    const v2f dz0 = (v2f){z0.x, z0.y} - qz, dz1 = (v2f){z0.z, z0.w} - qz, dz2 = (v2f){z1.x, z1.y} - qz, dz3 = (v2f){z1.z, z1.w} - qz;
                    pw += 16; fa += 2 * kMxOctetWords;
                    x0 = ld4(pw); x1 = ld4(pw + 4); y0 = ld4(pw + W); y1 = ld4(pw + W + 4); z0 = ld4(pw + 2 * W); z1 = ld4(pw + 2 * W + 4);
                    __builtin_amdgcn_sched_barrier(0);  // keep the re-issue here
                    auto within2 = [&](const v2f dx, const v2f dy, const v2f dz) -> uint32_t {
                        // FLANN L2_Simple: every product and sum rounded, in this order; RadiusResultSet: strict d2 < r2
                        const v2f d2 = pk_add_rn(pk_add_rn(pk_mul_rn(dx, dx), pk_mul_rn(dy, dy)), pk_mul_rn(dz, dz));
                        const v2f wgt = pk_within(d2, neg_big, r2_big);  // 1.0f / 0.0f: exact in bf16
                        return pack_hi16(__float_as_uint(wgt.y), __float_as_uint(wgt.x));
                    };
                    union { bf16x8 v; uint32_t u[4]; } b;
                    b.u[0] = within2(dx0, dy0, dz0);
                    b.u[1] = within2(dx1, dy1, dz1);
                    b.u[2] = within2(dx2, dy2, dz2);
                    b.u[3] = within2(dx3, dy3, dz3);
                    acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.v, acc[gi], 0, 0, 0);
                }
            }
        }
        // ---- D[row][query]: lane (q, h) of group g holds rows (k & 3) + 8 (k >> 2) + 4 h in acc[g][k].  A lane's home
        // query is query (lane & 31) of group (lane >> 5): it keeps its own half of that group's rows and swaps the
        // other group's registers with lane ^ 32 for the missing half.
        float own[16], got[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            own[k] = half ? acc[1][k] : acc[0][k];
            const float snd = half ? acc[0][k] : acc[1][k];
            got[k] = __shfl_xor(snd, 32, kWave);
        }
        auto frow = [&](int rr) -> double {  // feature row rr of the home query
            const int k = 4 * (rr >> 3) + (rr & 3), hh = (rr >> 2) & 1;
            return (double)((hh == half) ? own[k] : got[k]);
        };
        double mom[10];
        mom[0] = frow(0);
#pragma unroll
        for (int m = 0; m < 9; ++m) mom[1 + m] = (frow(1 + 3 * m) + frow(2 + 3 * m)) + frow(3 + 3 * m);

        const bool vox_ok = emit_normal(active, q, mom, vd, normals4, counts, qn, stat_t0);
        if (vd.enabled) voxel_sums(vox_ok, q, vd, vox_table);
    }
}

// ---- matrix-core formulation, distances too ("S1") ----------------------------------
// The neighbour predicate itself moves onto the matrix cores: for a block of 32 candidates and a group of 32 queries
//     D1[c][q] = |u_c|^2 + |v_q|^2 - 2 u_c . v_q = |c - q|^2,      u = c - o, v = q - o,
// is one more bf16 product over 24 k-slots (the bf16x3 terms of |u|^2, 1, u on the candidate side against 1, |v|^2,
// -2 v on the query side; cross terms below 2^-24 of the largest product are dropped).  D1 differs from FLANN's fp32
// value by ~1e-6 r^2 (tools/microbench/mfma_probe.hip); a pair closer than `band` to the threshold is re-evaluated
// exactly as FLANN does, everything else is decided by one packed multiply with clamp.  D1 comes out with the query on
// the lane and 16 candidates in its registers -- exactly the B fragment order the moment MFMA needs, so the weights
// never leave the registers.  The candidate-side operand of the distance product needs candidate rows with the
// FEATURE index along k, the moment product feature rows with the CANDIDATE index along k: one feature-major image
// serves both, the first through gfx950's transposed LDS read (ds_read_b64_tr_b16, tools/microbench/tr_probe.hip).
// VALU work per pair: ~2.7 operations instead of 5 (and 10.5 in the all-VALU kernel).
typedef short s4 __attribute__((ext_vector_type(4)));
#ifndef GM_MDCHUNK
#define GM_MDCHUNK 128
#endif
constexpr int kMdChunk = GM_MDCHUNK;                // candidates staged per chunk (multiple of 8)
constexpr int kMdSlots = kMdChunk + 24;             // a 32-candidate block may start at slot 120: reads reach slot 151
constexpr int kMdOctets = kMdSlots / 8;
constexpr int kMdOctetWords = 33 * 4;               // 31 feature rows (1 + 9 + 18 + 3) + 2 pad rows = 528 B: 132 dwords = 4 mod 32 banks
constexpr int kMdWaveLdsBytes = (kMdOctets * kMdOctetWords + 16) * 4;

// Feature rows of the image (one 16-byte row = eight candidates' bf16 values of one feature), ordered so that the FOUR
// transposed reads a lane issues per 32-candidate block for the distance product hit four CONSECUTIVE rows: one address
// register and three immediate offsets.  A lane's reads serve the k-slots b, b+4, b+16, b+20 of the product
// (b = 8 (lane >> 5) + ((lane & 15) >> 2)); slot b + {0,4,16,20} <-> row w(b) + {0,1,2,3}:
//     row  0: 1          1: u_x hi    2: u_y hi    3: u_z hi        window 0 (b = 0, 1, 2): against r2-|v|^2 and 2v, hi / mid / lo
//     row  4: u_x mid    5: u_y mid   6: u_z mid   7: |u|^2 mid     window 4 (b = 3: against 2v hi, -1;  b = 8: 2v mid, 0)
//     row  8: u_x lo     9: u_y lo   10: u_z lo   11: |u|^2 hi      window 8 (b = 9: against 2v hi, -1)
//     row 12: |u|^2 lo  13..30: the six products u_a u_b, hi / mid / lo each                window 12 (b = 10: -1, 0, 0, 0)
// (b = 11 carries zeros.)  The 24 products kept are the ones of round 2: (hi,hi) (mid,hi) (lo,hi) (hi,mid) (mid,mid)
// (hi,lo) per coordinate, the three terms of |u|^2 against -1 and of r2 - |v|^2 against 1.
constexpr int kRowX[3] = {1, 4, 8}, kRowY[3] = {2, 5, 9}, kRowZ[3] = {3, 6, 10}, kRowQ[3] = {11, 7, 12};   // hi, mid, lo
constexpr int kRowProd0 = 13;
__device__ __forceinline__ uint32_t md_window_of_base(int b)   // first row of the four a lane with slot base b reads
{
    return b < 3 ? 0u : ((b == 3 || b == 8) ? 4u : (b == 9 ? 8u : (b == 10 ? 12u : 0u)));
}
// fp32 -> three bf16 bit patterns (exact: 8 + 8 + 8 mantissa bits, truncation)
__device__ __forceinline__ void md_split3(float a, uint32_t out[3])
{
    const uint32_t ah = __float_as_uint(a) & 0xFFFF0000u;
    const float ra = a - __uint_as_float(ah);
    const uint32_t am = __float_as_uint(ra) & 0xFFFF0000u;
    const float sa = ra - __uint_as_float(am);
    out[0] = ah >> 16; out[1] = am >> 16; out[2] = __float_as_uint(sa) >> 16;
}
__device__ __forceinline__ s4 md_tr_read(const unsigned char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)p);
}

#ifndef GM_NORMALS_PREFETCH
#define GM_NORMALS_PREFETCH 1   // 0: a chunk's rows are loaded when the chunk is staged (A/B measurements)
#endif
#ifdef GM_NORMALS_TIMELINE  // diagnostic build (tools/tile_timeline.py): per wave of the straight-line copy, 100 MHz ticks of its first
                            // instruction, of the moment it knows its tile, and of its end
__device__ unsigned long long gm_tl_buf[3 * 65536];
#endif
#ifdef GM_NORMALS_PHASES   // diagnostic build (tools/normals_phases.py): shader-clock ticks a wave spends in each part of a tile
__device__ unsigned long long gm_phase_ticks[16];
#define GM_PH_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define GM_PH_ADD(slot, expr) do { ph_acc[slot] += (uint32_t)(expr); } while (0)
#else
#define GM_PH_STAMP(var) do {} while (0)
#define GM_PH_ADD(slot, expr) do {} while (0)
#endif

// FINE = false: the usual grid (cells one radius wide, 3 x 3 rows: everything about rows is a compile-time constant);
// FINE = true: y/z rows finer than the radius (GridParams::D > 1), chosen for frames with very many neighbours per point.
template <bool FINE>
__device__ __forceinline__ void normals_tile_mxd(const NormalsArgs &A, unsigned char *lds, const uint2 tile, uint32_t min_candidates,
                                                 unsigned long long stat_entry = 0ull)
{
    const float4 *__restrict__ spts4 = A.spts4;
    const uint32_t *__restrict__ skeys = A.skeys;
    DevCounters *__restrict__ ctr = A.ctr;
    const GridParams &g = A.g;
    const uint2 *__restrict__ row_bounds = A.row_bounds;
    float4 *__restrict__ normals4 = A.normals4;
    int32_t *__restrict__ counts = A.counts;
    const VoxDense &vd = A.vd;
    VoxCell *__restrict__ vox_table = A.vox_table;
    (void)ctr;
    uint32_t *feat = reinterpret_cast<uint32_t *>(lds);    // this wave's slice: feature rows only
    // (the lane id goes through an empty asm: each inlined copy of this function then derives its lane patterns from a value
    // of its own, and the compiler cannot keep one copy's patterns alive in registers across the other copy's loops)
    int lane = lane_id();
    asm volatile("" : "+v"(lane));
    const uint32_t n = ctr->n_cropped;
    const int qsel = lane & 31, half = lane >> 5;
    // lane roles of the transposed reads that build the distance MFMA's A fragment (see the header of this section)
    const int tq = (lane & 15) >> 2, tp = lane & 3, tr0 = 16 * ((lane >> 4) & 1);
    // byte offset of this lane's transposed reads inside a 32-candidate block: first row of its window + candidate quad
    const uint32_t tr_off = md_window_of_base(8 * half + tq) * 16u +
                            (uint32_t)((tr0 + 4 * tp) >> 3) * (uint32_t)(kMdOctetWords * 4) + (uint32_t)((tr0 + 4 * tp) & 7) * 2u;
    const uint32_t mom_off = (uint32_t)qsel * 16u + 8u * (uint32_t)half;   // ... and of its moment-MFMA operand reads
    {
        const uint32_t qs = tile.x, qn = tile.y & 0xFFu, tile_row = tile.y >> 8;   // first query, queries (1 .. 64), x-row of the tile
#ifdef GM_NORMALS_TIMELINE
        const unsigned long long stat_t0 = wall_clock64();
#else
        const unsigned long long stat_t0 = 0ull;
#endif
#ifdef GM_NORMALS_PHASES
        uint32_t ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (wave-uniform: scalar registers)
#endif
        GM_PH_STAMP(ph_t0);
        const bool active = (uint32_t)lane < qn;
        const uint32_t qidx = qs + (active ? (uint32_t)lane : qn - 1u);
        const float4 q = spts4[qidx];
        const uint32_t kl = skeys[qidx];
        // (the tile's x-row comes with the tile: the row table below is asked before anything waits for the keys)
        const uint32_t row = tile_row;
        const int cy = (int)(row % (uint32_t)g.ny), cz = (int)(row / (uint32_t)g.ny);
        // ---- candidate windows.  The tile's candidates lie in the (2D+1)^2 rows around its own; they are taken in passes
        // of up to 32 rows: lane i finds both ends of the window of (row i / 2 of the pass, group i % 2).  D = 1 (the
        // usual grid): 9 rows, one pass.
        const int gD = FINE ? g.D : 1;
        const int side = 2 * gD + 1, nrows_all = side * side;
        int nrows = 0;   // rows of the current pass
        // which row of the grid a lane searches in a pass, and that row's entry of the row table (first / one-past-last sorted
        // position) -- asked for here, while the query and key loads above are still in flight
        struct LaneRow { int gg, reach; uint32_t nrow; bool ok; uint2 rb; };
        auto lane_row = [&](int row0) -> LaneRow {
            nrows = nrows_all - row0 < 32 ? nrows_all - row0 : 32;
            const int k = FINE ? lane : (lane >> 1);     // (row, group) this lane works for
            const int r = row0 + (k >> 1);
            const int a = (r % side) - gD, b = (r / side) - gD;
            const int yy = cy + a, zz = cz + b;
            LaneRow R;
            R.gg = k & 1;
            R.reach = (k >> 1) < nrows ? (FINE ? (int)g.reach[a < 0 ? -a : a][b < 0 ? -b : b] : g.xreach) : 0;
            R.ok = R.reach > 0 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz;
            R.nrow = R.ok ? (uint32_t)(zz * g.ny + yy) : 0u;
            R.rb = R.ok ? row_bounds[R.nrow] : make_uint2(0u, 0u);
            return R;
        };
        const LaneRow lr0 = lane_row(0);
        const int fxl = (int)(kl - row * (uint32_t)g.nx);
        if (__ballot(active && q.x >= vd.own_lo && q.x < vd.own_hi) == 0) {  // halo-only tile (slab sharding)
            if (active) {
                const float nanv = __builtin_nanf("");
                normals4[__float_as_uint(q.w)] = make_float4(nanv, nanv, nanv, nanv);
                if (counts) counts[__float_as_uint(q.w)] = 0;
            }
            return;
        }
        // Window ends per (row, group).  FINE: lane i holds both ends of (row i / 2 of the pass, group i % 2) in sb / se.  The
        // usual grid: ONE end per lane -- lane 2 k holds the begin, lane 2 k + 1 the end of (row k / 2, group k % 2) in sb --
        // so the 36 searches of a tile run on 36 lanes instead of two apiece on 18: half the instructions per round.
        uint32_t sb = 0, se = 0;
        auto win_begin = [&](int r, int gg) -> uint32_t {
            return FINE ? __builtin_amdgcn_readlane(sb, r * kMxGroups + gg) : __builtin_amdgcn_readlane(sb, 2 * (r * kMxGroups + gg));
        };
        auto win_end = [&](int r, int gg) -> uint32_t {
            return FINE ? __builtin_amdgcn_readlane(se, r * kMxGroups + gg) : __builtin_amdgcn_readlane(sb, 2 * (r * kMxGroups + gg) + 1);
        };
        auto find_windows = [&](const LaneRow &R) {
            sb = 0; se = 0;
            const int gg = R.gg, reach = R.reach;
            const int lo_fx = __shfl(fxl, gg * kMxGroupLanes, kWave);
            const int hi_fx = __shfl(fxl, gg * kMxGroupLanes + kMxGroupLanes - 1, kWave);
            uint32_t key_b = 0, key_e = 0, lo1 = 0, hi1 = 0, row_key0 = 0;   // row_key0: the row's first key
            if (R.ok) {
                const uint32_t rbk = R.nrow * (uint32_t)g.nx;
                const int xa = lo_fx > reach ? lo_fx - reach : 0;
                const int xb = hi_fx + reach < g.nx - 1 ? hi_fx + reach : g.nx - 1;
                key_b = rbk + (uint32_t)xa; key_e = rbk + (uint32_t)xb + 1u;
                lo1 = R.rb.x; hi1 = R.rb.y; row_key0 = rbk;
            }
            if (FINE) {
                uint32_t lo2 = lo1, hi2 = hi1;
                while (lo1 < hi1 || lo2 < hi2) {
                    const uint32_t m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                    const uint32_t k1 = skeys[m1 < n ? m1 : n - 1u], k2 = skeys[m2 < n ? m2 : n - 1u];
                    if (lo1 < hi1) { if (k1 < key_b) lo1 = m1 + 1u; else hi1 = m1; }
                    if (lo2 < hi2) { if (k2 < key_e) lo2 = m2 + 1u; else hi2 = m2; }
                }
                sb = lo1; se = lo2;
            } else {
                // (Searching 5, 9 or 17 ways per round -- a third of the dependent round trips -- was measured in round 3: no
                // change.  Round 4 again: 8 ways, then 15 probes at once = 4 round trips and 36 probe loads instead of 12 and 12:
                // 151.5 us against 150.  What costs is the number of these loads, each lane its own cache line, not the depth of
                // the chain: with the interpolated guess alone -- two loads, wrong windows -- the kernel runs 4.6 % faster,
                // which is all a perfect search could give.  And once more behind the guided round below: finishing a range of
                // <= 128 positions with 15 probes at once and then 8 -- three rounds after the row table instead of eight, the
                // probes inside a few cache lines -- is 1.5 % slower alone and 2 % slower per pipelined step.  A coarse level
                // first (every 64th key, packed by the tile cutter into 52 KB that stay in cache: six halvings there, six
                // between two samples) makes the head of a tile LONGER, 19.9 k -> 22.2 k clocks: a round costs its ~1 500
                // clocks whether the line is cached or not -- it is the trip through the vector-memory pipeline under this
                // kernel's load, not a miss, and only fewer trips help.  Fewer trips by finishing all 36 searches at once --
                // the wave loads the <= 64 keys left of each with one coalesced instruction, ballot + population count --
                // costs ~500 instructions: 133 -> 147 us.)
                const uint32_t key = (lane & 1) ? key_e : key_b;   // first position of the row whose key is >= key
#ifndef GM_NORMALS_NO_GUIDED_PROBE
                // A guided first round: two probes kGuide positions either side of where the key would sit in a row that
                // covers the grid's x-range evenly.  In such a row (the tunnel's) they bracket the answer and seven
                // halvings are left of twelve; in any other row they are two ordinary probes of the search -- the range
                // still shrinks to one of three parts, one round is lost at worst.
                constexpr uint32_t kGuide = 64;
                if (hi1 - lo1 > 8u * kGuide) {
                    const uint32_t xq = key - row_key0;   // the key's x cell (0 .. nx)
                    const uint32_t gpos = lo1 + (uint32_t)((float)xq * (float)(hi1 - lo1) / (float)g.nx);
                    const uint32_t pa0 = gpos > lo1 + kGuide ? gpos - kGuide : lo1;
                    const uint32_t pb0 = gpos + kGuide < hi1 ? gpos + kGuide : hi1 - 1u;
                    const uint32_t ka0 = skeys[pa0], kb0 = skeys[pb0];   // (lo1 <= pa0 <= pb0 < hi1 <= n)
                    if (ka0 < key) lo1 = pa0 + 1u; else hi1 = pa0;
                    if (pb0 >= lo1 && pb0 < hi1) { if (kb0 < key) lo1 = pb0 + 1u; else hi1 = pb0; }
                }
#endif
                while (lo1 < hi1) {
                    const uint32_t m1 = (lo1 + hi1) >> 1;
                    if (skeys[m1] < key) lo1 = m1 + 1u; else hi1 = m1;   // (m1 < hi1 <= n)
                }
                sb = lo1;
            }
        };
        find_windows(lr0);
        GM_PH_STAMP(ph_t1);
        GM_PH_ADD(0, ph_t1 - ph_t0);   // tile head: query + key loads, window search
        // Thin neighbourhoods (fewer than min_candidates candidates in the first group's windows) take the direct path
        // below: nothing to amortise the feature staging over, and the few-point covariances of a sparse cloud are
        // near-degenerate, where offsets from the query itself (|offset| < r) in fp64 keep more than offsets from a
        // tile origin through bf16 features do.  (Only on the usual grid: finer rows are chosen for dense frames.)
        bool thin = false;
        if (!FINE) {
            // (summed on the scalar side: a wave_sum here would leave its six shuffle patterns in vector registers all the
            // way to the epilogue's reductions, across the candidate stream)
            uint32_t wl = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r)
                wl += win_end(r, 0) - win_begin(r, 0);
            thin = wl < min_candidates;
        }
        // one origin per tile: the tile's middle query snapped to a multiple of g.snap (a power of two >= one ulp of the
        // largest coordinate): o is then a multiple of every point's ulp, so the offsets c - o and q - o below carry at
        // most the rounding of a number of size ~r, not of size ~coordinate
        const int mid = (int)(qn >> 1);
        const float inv_snap = 1.0f / g.snap;   // exact: powers of two
        // (readlane on the BIT PATTERN: the builtin is integer-typed, a float argument would be converted, i.e. truncated)
        const float ox = rintf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.x), mid)) * inv_snap) * g.snap,
                    oy = rintf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.y), mid)) * inv_snap) * g.snap,
                    oz = rintf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.z), mid)) * inv_snap) * g.snap;
#ifdef GM_MD_DEBUG
        const float band = g.band;
#endif
        const int ngroups = qn > (uint32_t)kMxGroupLanes ? 2 : 1;  // (a second group of repeated queries is skipped)
        f32x16 acc[kMxGroups];

#pragma unroll
        for (int gi = 0; gi < kMxGroups; ++gi)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[gi][k] = 0.f;

        // the distance MFMA's B operand (queries): per group and K-step, this lane's 8 coefficients of its query
        //   d2(c, q) = |u|^2 + |v|^2 - 2 u.v,  u = c - o,  v = q - o,   over 24 k-slots (md_row_of_slot)
        bf16x8 qb[kMxGroups][2];
        float gq[kMxGroups][3];
#pragma unroll
        for (int gi = 0; gi < kMxGroups; ++gi) {
            gq[gi][0] = __shfl(q.x, gi * kMxGroupLanes + qsel, kWave);
            gq[gi][1] = __shfl(q.y, gi * kMxGroupLanes + qsel, kWave);
            gq[gi][2] = __shfl(q.z, gi * kMxGroupLanes + qsel, kWave);
            const float vx = gq[gi][0] - ox, vy = gq[gi][1] - oy, vz = gq[gi][2] - oz;
            const float vv = __fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz));
            uint32_t sv[3], sx[3], sy[3], sz[3];
            // The product is t = r2 - d2 itself: the |u|^2 rows meet -1, the constant row meets r2 - |v|^2 (one more
            // rounding of a number of size r2: 6e-8 r2, far inside the band), the u rows meet +2v -- so the weights
            // below need no subtraction.
            // Everything times dscale (a power of two: exact), see GridParams.
            const float S = g.dscale;
            md_split3(__fsub_rn(g.r2, vv) * S, sv); md_split3(2.0f * S * vx, sx); md_split3(2.0f * S * vy, sy); md_split3(2.0f * S * vz, sz);
            const uint32_t one = __float_as_uint(-S) >> 16;   // bf16(-dscale)
            // 16-bit patterns of this lane's eight k-slots per K-step: slot 16 t + 8 half + j <-> base 8 half + (j & 3),
            // row offset 2 t + (j >> 2) of that base's window (table in front of md_window_of_base)
            const uint32_t lo0[8] = {sv[0], sv[1], sv[2], sx[0], sx[0], sx[1], sx[2], sy[0]};   // half 0, K-step 0
            const uint32_t lo1[8] = {sy[0], sy[1], sy[2], sz[0], sz[0], sz[1], sz[2], one};     // half 0, K-step 1
            const uint32_t hi0[8] = {sx[1], sx[0], one, 0u, sy[1], sy[0], 0u, 0u};              // half 1, K-step 0
            const uint32_t hi1[8] = {sz[1], sz[0], 0u, 0u, 0u, one, 0u, 0u};                    // half 1, K-step 1
            union { bf16x8 v; uint32_t u[4]; } f0, f1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f0.u[j] = half ? (hi0[2 * j] | (hi0[2 * j + 1] << 16)) : (lo0[2 * j] | (lo0[2 * j + 1] << 16));
                f1.u[j] = half ? (hi1[2 * j] | (hi1[2 * j + 1] << 16)) : (lo1[2 * j] | (lo1[2 * j + 1] << 16));
            }
            qb[gi][0] = f0.v; qb[gi][1] = f1.v;
        }
        auto row_begin = [&](int r) -> uint32_t { return win_begin(r, 0); };
        auto row_end = [&](int r) -> uint32_t { return win_end(r, ngroups - 1); };
        if (thin) {
            // direct path, self-contained: every lane tests its own query against every candidate of the tile's row
            // ranges (wave-uniform loads straight from the sorted cloud) and sums the offsets from the query in fp64
            double tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < 9; ++r) {
                const uint32_t e = row_end(r);
                for (uint32_t i = row_begin(r); i < e; ++i) {
                    const float4 c = spts4[i];
                    const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
                    // FLANN L2_Simple: every product and sum rounded, in this order; RadiusResultSet: strict d2 < r2
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    if (d2 < g.r2) {
                        const double ex = dx, ey = dy, ez = dz;
                        tm[0] += 1.0; tm[1] += ex; tm[2] += ey; tm[3] += ez;
                        tm[4] += ex * ex; tm[5] += ex * ey; tm[6] += ex * ez; tm[7] += ey * ey; tm[8] += ey * ez; tm[9] += ez * ez;
                    }
                }
            }
            const bool vox_ok = emit_normal(active, q, tm, vd, normals4, counts, qn, stat_t0);
            if (vd.enabled) voxel_sums(vox_ok, q, vd, vox_table);
            return;
        }
        // ---- The usual grid: the windows in STREAM coordinates.  The tile's candidates form one stream of image slots: the
        // windows of its 9 rows one after the other, each starting on an octet of slots.  S_r = first slot of row r.  Every
        // window end (lane 4r + 2g: begin, + 1: end of (row r, group g)) becomes its position in that stream; the register's
        // spare lanes take S_{r+1} (lanes 40 + r) and rb_r - S_r (lanes 50 + r: sorted position = slot + that).  What a chunk
        // holds is then arithmetic on its first slot -- no walk over rows and pieces (that walk, ~70 scalar and readlane
        // instructions per piece in a dependent chain, was a fifth of a tile's time: GM_PH_ASSEMBLE).
        uint32_t gs = 0, stream_total = 0, nxt_c = 0;
        const int endl = 2 * (ngroups - 1) + 1;   // lane offset of the end of a row's last group
        if (!FINE) {
            const int rl = lane & ~3;
            const uint32_t rb = __shfl(sb, rl, kWave), re = __shfl(sb, rl + endl, kWave);   // the row's window, all groups
            const bool rowlane = lane < 36;
            const uint32_t len = rowlane ? re - rb : 0u;
            const uint32_t padded = (len + 7u) & ~7u;
            // S_{r+1} on the row's four lanes: a scan over the rows, four lanes apiece with the same value -- lane shifts by 4
            // and 8 inside each 16-lane DPP row, then the row totals handed on (lane 15 / 31 hold their row's total)
            uint32_t incl = padded;
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
            const uint32_t srow = incl - padded;                                                     // S_r
            const uint32_t s_next = __shfl(incl, 4 * (lane - 40), kWave);        // (lanes 40 .. 48)
            const uint32_t off_r = __shfl(rb - srow, 4 * (lane - 50), kWave);    // (lanes 50 .. 58)
            gs = rowlane ? srow + (sb - rb) : ((lane >= 40 && lane < 49) ? s_next : ((lane >= 50 && lane < 59) ? off_r : 0u));
            stream_total = __builtin_amdgcn_readlane(gs, 48);
        }
        GM_PH_STAMP(ph_t2);
        GM_PH_ADD(1, ph_t2 - ph_t1);   // origin, query-side operand fragments
        // ---- the candidate stream.  The tile's candidates -- the windows of its (2D+1)^2 rows, one after the other -- are
        // cut into chunks of up to kMdChunk image slots.  A chunk holds up to kMdPieces pieces, each a run of one row's window
        // starting on an octet of slots, so that chunks are full: the rest of one row's window and the head of the next
        // share a chunk instead of leaving most of its lanes idle when it is staged.
        // Per group the pieces' clipped windows are covered by ONE run of 32-candidate blocks [lo, hi): what lies between
        // two windows is not a neighbour of any query of the group (a window is a superset of its neighbours), and every
        // slot that holds no candidate is marked far.  Chunks are assembled one AHEAD: the rows of chunk i+1 are loaded
        // right after chunk i has been staged and stay in flight, in registers, while the wave runs chunk i's pair loops.
        // Lane l stages slots 2l and 2l+1 (two bf16 of a feature row make one dword).
        static_assert(kMdChunk == 2 * kWave, "one slot pair per lane and chunk");
        constexpr int kMdPieces = 4;
        int cur_r = 0, row0 = 0;                  // cursor: row of the pass (first row of the pass), ...
        uint32_t cur_c = row_begin(0);            // ... next unread position of its window
        uint32_t n_lo[kMxGroups], n_hi[kMxGroups], n_slots = 0;   // the chunk being fetched: block range per group, slots used
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;    // its rows (this lane's slot pair)
        uint32_t pidx = qs, pv = 0;               // sorted position of the pair's first candidate; bit 0 / 1: first / second slot holds one
        auto assemble_stream = [&]() {   // the usual grid
            n_slots = 0; pv = 0; pidx = qs;
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) { n_lo[gi] = 0xFFFFFFFFu; n_hi[gi] = 0u; }
            if (nxt_c >= stream_total) return;   // wave-uniform
            const uint32_t c0 = nxt_c, c1 = nxt_c + (uint32_t)kMdChunk;
            nxt_c = c1;
            n_slots = stream_total - c0 < (uint32_t)kMdChunk ? stream_total - c0 : (uint32_t)kMdChunk;
            // this lane's slot pair: its row = the number of row ends at or before its first slot (empty rows end where they begin)
            const uint32_t sl0 = c0 + 2u * (uint32_t)lane;
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) r += sl0 >= (uint32_t)__builtin_amdgcn_readlane(gs, 40 + k) ? 1u : 0u;
            r = r > 8u ? 8u : r;   // (past the stream: the last row's end says so below)
            const uint32_t off = __shfl(gs, 50 + (int)r, kWave), lim = __shfl(gs, 4 * (int)r + endl, kWave);
            const bool v0 = sl0 < lim, v1 = sl0 + 1u < lim;
            pidx = v0 ? sl0 + off : qs;
            pv = v0 ? (v1 ? 3u : 1u) : 0u;
            // per group: the chunk's part of its windows.  Windows follow each other in the stream, so the run of blocks that
            // covers them goes from the first one that reaches into the chunk to the last one
            const uint32_t other = __shfl_xor(gs, 1, kWave);   // (a begin lane reads its end)
            const uint32_t cb = gs > c0 ? gs : c0, ce = other < c1 ? other : c1;
            const uint64_t m = __ballot((lane & 1) == 0 && lane < 36 && ce > cb);
            const uint32_t cb8 = (cb & ~7u) - c0, ce0 = ce - c0;
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) {
                if (gi >= ngroups) break;
                const uint64_t mg = m & (gi ? 0x444444444ull : 0x111111111ull);
                if (mg) {   // wave-uniform
                    n_lo[gi] = __builtin_amdgcn_readlane(cb8, (int)__builtin_ctzll(mg));
                    n_hi[gi] = __builtin_amdgcn_readlane(ce0, 63 - (int)__builtin_clzll(mg));
                }
            }
            pa = spts4[pidx];
            pb = spts4[pidx + (pv >> 1)];
        };
        auto assemble_walk = [&]() {     // finer grids: rows in passes of 32
            n_slots = 0; pv = 0; pidx = qs;
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) { n_lo[gi] = 0xFFFFFFFFu; n_hi[gi] = 0u; }
            int pieces = 0;
            for (;;) {   // wave-uniform
                if (cur_r >= nrows) {   // the pass has no rows left
                    if (!FINE || row0 + 32 >= nrows_all) break;
                    row0 += 32;
                    find_windows(lane_row(row0));   // the next pass of rows (finer grids only)
                    cur_r = 0; cur_c = row_begin(0);
                    continue;
                }
                const uint32_t e = row_end(cur_r);
                if (cur_c >= e) {
                    ++cur_r;
                    if (cur_r < nrows) cur_c = row_begin(cur_r);
                    continue;
                }
                const uint32_t len = e - cur_c, room = (uint32_t)kMdChunk - n_slots;
                const uint32_t take = len < room ? len : room;
#pragma unroll
                for (int gi = 0; gi < kMxGroups; ++gi) {
                    if (gi >= ngroups) break;
                    const uint32_t mb = win_begin(cur_r, gi), me = win_end(cur_r, gi);
                    uint32_t ob = mb > cur_c ? mb - cur_c : 0u, oe = me > cur_c ? me - cur_c : 0u;
                    if (ob > take) ob = take;
                    if (oe > take) oe = take;
                    if (oe > ob) {
                        const uint32_t l = n_slots + (ob & ~7u), h = n_slots + oe;
                        n_lo[gi] = l < n_lo[gi] ? l : n_lo[gi];
                        n_hi[gi] = h > n_hi[gi] ? h : n_hi[gi];
                    }
                }
                {
                    const uint32_t i = 2u * (uint32_t)lane;
                    const bool in = i >= n_slots && i < n_slots + take;
                    pidx = in ? cur_c + (i - n_slots) : pidx;
                    pv = in ? (i + 1u < n_slots + take ? 3u : 1u) : pv;
                }
                n_slots += (take + 7u) & ~7u;
                cur_c += take;
                ++pieces;
                if (n_slots >= (uint32_t)kMdChunk || pieces == kMdPieces) break;
            }
            if (n_slots) {   // (lanes without a candidate re-read the tile's first query: every lane's loads are unconditional)
                pa = spts4[pidx];
                pb = spts4[pidx + (pv >> 1)];
            }
        };
        auto assemble = [&]() { if (FINE) assemble_walk(); else assemble_stream(); };
        assemble();
        while (n_slots) {
            uint32_t c_lo[kMxGroups], c_hi[kMxGroups];
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) { c_lo[gi] = n_lo[gi]; c_hi[gi] = n_hi[gi]; }
#ifdef GM_NORMALS_STATS
            const uint32_t c_slots = n_slots;
#endif
            wave_lds_fence();  // previous chunk fully consumed
            GM_PH_STAMP(ph_c0);
#ifdef GM_NORMALS_PHASES
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GM_PH_STAMP(ph_c1);
            GM_PH_ADD(2, ph_c1 - ph_c0);   // waiting for the chunk's rows
#endif
            // ---- stage the chunk
            {
                const uint32_t i = 2u * (uint32_t)lane;
                const uint32_t oct = i >> 3;
                uint32_t *fo = feat + oct * (uint32_t)kMdOctetWords + ((i & 7u) >> 1);
                auto st = [&](int frow, uint32_t v) { fo[frow * 4] = v; };
                if (pv & 1u) {
                    const bool vb = (pv & 2u) != 0u;
                    const float4 ca = pa, cb = pb;
                    const float uxa = ca.x - ox, uya = ca.y - oy, uza = ca.z - oz;
                    const float uxb = vb ? cb.x - ox : 0.f, uyb = vb ? cb.y - oy : 0.f, uzb = vb ? cb.z - oz : 0.f;
                    auto split3 = [&](float a, float b, int rh, int rm, int rl) {  // exact: 8 + 8 + 8 mantissa bits
                        const uint32_t ah = __float_as_uint(a) & 0xFFFF0000u, bh = __float_as_uint(b) & 0xFFFF0000u;
                        st(rh, pack_hi16(bh, ah));
                        const float ra = a - __uint_as_float(ah), rb = b - __uint_as_float(bh);
                        const uint32_t am = __float_as_uint(ra) & 0xFFFF0000u, bm = __float_as_uint(rb) & 0xFFFF0000u;
                        st(rm, pack_hi16(bm, am));
                        const float sa = ra - __uint_as_float(am), sb2 = rb - __uint_as_float(bm);
                        st(rl, pack_hi16(__float_as_uint(sb2), __float_as_uint(sa)));
                    };
                    auto prod3 = [&](float a, float b, int k) { split3(a, b, kRowProd0 + 3 * k, kRowProd0 + 3 * k + 1, kRowProd0 + 3 * k + 2); };
                    st(0, 0x3F803F80u);  // the count row: 1.0 | 1.0
                    split3(uxa, uxb, kRowX[0], kRowX[1], kRowX[2]); split3(uya, uyb, kRowY[0], kRowY[1], kRowY[2]);
                    split3(uza, uzb, kRowZ[0], kRowZ[1], kRowZ[2]);
                    const float xxa = uxa * uxa, yya = uya * uya, zza = uza * uza, xxb = uxb * uxb, yyb = uyb * uyb, zzb = uzb * uzb;
                    prod3(xxa, xxb, 0); prod3(uxa * uya, uxb * uyb, 1); prod3(uxa * uza, uxb * uzb, 2);
                    prod3(yya, yyb, 3); prod3(uya * uza, uyb * uzb, 4); prod3(zza, zzb, 5);
                    // |u|^2; a missing second candidate of the pair is "far" (2^100: d2 = 2^100 + ... never within r2)
                    split3(__fadd_rn(__fadd_rn(xxa, yya), zza), vb ? __fadd_rn(__fadd_rn(xxb, yyb), zzb) : 0x1p100f, kRowQ[0], kRowQ[1], kRowQ[2]);
                    // the candidates' sorted positions, in the octet's two spare rows: the rare exact re-evaluation of a pair
                    // finds its candidate through them
                    *reinterpret_cast<uint2 *>(feat + oct * (uint32_t)kMdOctetWords + 31u * 4u + (i & 7u)) = make_uint2(pidx, pidx + 1u);
                } else {
                    // a slot pair without candidates (between two pieces, behind the last one): far (the other rows are stale
                    // but finite, the weights come out 0 like any other far candidate's: no index masks in the pair loop)
                    st(kRowQ[0], 0x71807180u);   // |u|^2 hi := bf16(2^100) | bf16(2^100)
                }
                // slots behind the chunk that a group's last 32-candidate block may still read
                if (lane < (kMdSlots - kMdChunk) / 2)
                    feat[((uint32_t)kMdChunk / 8u + ((uint32_t)lane >> 2)) * (uint32_t)kMdOctetWords + ((uint32_t)lane & 3u) + (uint32_t)kRowQ[0] * 4u] = 0x71807180u;
            }
            wave_lds_fence();
#ifdef GM_PH_ASSEMBLE   // (diagnostic: the next chunk's assembly on its own, in the slot of the -- empty -- wait for the rows)
            GM_PH_STAMP(ph_ca);
#endif
#if GM_NORMALS_PREFETCH
            assemble();   // the next chunk: its rows are in flight during the pair loops below
#endif
            GM_PH_STAMP(ph_c2);
#ifdef GM_NORMALS_PHASES
#ifdef GM_PH_ASSEMBLE
            GM_PH_ADD(2, ph_c2 - ph_ca);
            GM_PH_ADD(3, ph_ca - ph_c1);
#else
            GM_PH_ADD(3, ph_c2 - ph_c1);   // feature image of the chunk
#endif
#endif
#pragma unroll
            for (int gi = 0; gi < kMxGroups; ++gi) {
                if (gi >= ngroups) break;  // wave-uniform
                const uint32_t ob8 = c_lo[gi], oe = c_hi[gi];
#if defined(GM_NORMALS_STATS) && !defined(GM_MD_DEBUG)
                if (lane == 0) {
                    atomicAdd(&ctr->pad[0], (oe > ob8 ? ((oe - ob8 + 31u) >> 5) * 32u : 0u) / 2u);  // candidates streamed per query, in 64-query tile units
                    atomicAdd(&ctr->pad[1], (oe > ob8 ? oe - ob8 : 0u) * 2u);
                    if (gi == 0) { atomicAdd(&ctr->pad[2], c_slots); atomicAdd(&ctr->pad[3], 1u); }
                }
#endif
                const int nblk = oe > ob8 ? (int)((oe - ob8 + 31u) >> 5) : 0;   // 32-candidate blocks, wave-uniform
                const unsigned char *fbytes = reinterpret_cast<const unsigned char *>(feat);
                uint32_t p = ob8;
                for (int it = 0; it < nblk; ++it, p += 32u) {
                    const unsigned char *blk = fbytes + (p >> 3) * (uint32_t)(kMdOctetWords * 4);
                    // ---- distance MFMA: D1[candidate p + r][query] over the 24 k-slots.  A fragment (candidate rows,
                    // feature k) out of the feature-major image by transposed reads: 4 rows x 16 candidates per read
                    const unsigned char *ta = blk + tr_off;
                    union { bf16x8 v; s4 h[2]; } a0, a1;
                    a0.h[0] = md_tr_read(ta); a0.h[1] = md_tr_read(ta + 16);          // k-slots b, b + 4
                    a1.h[0] = md_tr_read(ta + 32); a1.h[1] = md_tr_read(ta + 48);    // k-slots b + 16, b + 20
                    // moment MFMA's A fragments (feature row qsel, the two quads of k-step s of this lane half)
                    const unsigned char *ma = blk + mom_off;
                    union { bf16x8 v; uint2 q2[2]; } m0, m1;
                    m0.q2[0] = *reinterpret_cast<const uint2 *>(ma);
                    m0.q2[1] = *reinterpret_cast<const uint2 *>(ma + kMdOctetWords * 4);
                    m1.q2[0] = *reinterpret_cast<const uint2 *>(ma + 2 * kMdOctetWords * 4);
                    m1.q2[1] = *reinterpret_cast<const uint2 *>(ma + 3 * kMdOctetWords * 4);
                    f32x16 d1;
#pragma unroll
                    for (int k = 0; k < 16; ++k) d1[k] = 0.f;
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, qb[gi][0], d1, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, qb[gi][1], d1, 0, 0, 0);
                    // ---- weights: 1 where d2 < r2.  d1 is within `band` of FLANN's fp32 value (mfma_probe.hip): outside the
                    // band the decision is certain; inside it the pair is re-evaluated exactly as FLANN does
                    uint32_t pw[8];   // two bf16 weights each: candidates 2k (low half) and 2k + 1 of this lane's 16
                    float amin = 3.0e38f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) amin = fminf(fminf(fabsf(d1[2 * k]), fabsf(d1[2 * k + 1])), amin);
                    // clamp01 of T = (r2 - d2) * dscale, rounded to bf16: exactly 1 or 0 outside the band.  (The asm takes
                    // amin as an operand it does not use: that orders it behind the compiler's own reads of d1, for which
                    // the MFMA -> VALU wait states are inserted; nothing inserts them for an asm statement.)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(pw[k]) : "v"(d1[2 * k]), "v"(d1[2 * k + 1]), "v"(amin));
#ifdef GM_MD_DEBUG   // diagnostic build: error of the distance MFMA against FLANN's fp32 value, every pair
                    {
                        const float vx = gq[gi][0] - ox, vy = gq[gi][1] - oy, vz = gq[gi][2] - oz;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const uint32_t rk = (uint32_t)((k & 3) + 8 * (k >> 2) + 4 * half);
                            if (p + rk < oe) {
                                const float4 c = spts4[feat[((p + rk) >> 3) * (uint32_t)kMdOctetWords + 31u * 4u + ((p + rk) & 7u)]];
                                const float dx = c.x - gq[gi][0], dy = c.y - gq[gi][1], dz = c.z - gq[gi][2];
                                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                                const float d1k = g.r2 - d1[k] / g.dscale;
                                const float err = fabsf(d1k - d2) / g.r2;
                                if (d2 < 4.0f * g.r2) {
                                    atomicMax(&ctr->pad[1], __float_as_uint(err));
                                    if (err > 1e-5f) atomicAdd(&ctr->pad[2], 1u);
                                    if (fabsf(d1k - d2) > band) {
                                        atomicAdd(&ctr->pad[0], 1u);
                                        atomicMax(&ctr->pad[3], (uint32_t)(sqrtf(vx * vx + vy * vy + vz * vz) / sqrtf(g.r2) * 1000.f));
                                        const float ux = c.x - ox, uy = c.y - oy, uz = c.z - oz;
                                        atomicMax(&ctr->pad[4], (uint32_t)(sqrtf(ux * ux + uy * uy + uz * uz) / sqrtf(g.r2) * 1000.f));
                                    }
                                }
                            }
                        }
                    }
#endif
                    if (__ballot(amin < g.dband)) {   // rare: some pair of this block lies inside the band
                        // (the lane's query of this group is read again, an L2 hit, rather than kept in three registers per
                        // group across the whole stream: those registers hold the next chunk's rows)
                        uint32_t gqi = (uint32_t)(gi * kMxGroupLanes + qsel);
                        gqi = qs + (gqi < qn ? gqi : qn - 1u);
                        asm volatile("" : "+v"(gqi));   // (keeps the address arithmetic here, out of the loop's live registers)
                        const float4 gqv = spts4[gqi];
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const uint32_t rk = (uint32_t)((k & 3) + 8 * (k >> 2) + 4 * half);
                            const bool need = fabsf(d1[k]) < g.dband;   // (far slots are never inside the band: a real candidate)
                            if (__ballot(need)) {
                                if (need) {
                                    const float4 c = spts4[feat[((p + rk) >> 3) * (uint32_t)kMdOctetWords + 31u * 4u + ((p + rk) & 7u)]];
                                    const float dx = c.x - gqv.x, dy = c.y - gqv.y, dz = c.z - gqv.z;
                                    // FLANN L2_Simple: every product and sum rounded, in this order; strict d2 < r2
                                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                                    const uint32_t wk = (d2 < g.r2 ? 1u : 0u) * 0x3F80u;   // (a multiply: no constant held in a register across the loop)
                                    pw[k >> 1] = (k & 1) ? ((pw[k >> 1] & 0x0000FFFFu) | (wk << 16)) : ((pw[k >> 1] & 0xFFFF0000u) | wk);
                                }
                            }
                        }
                    }
                    union { bf16x8 v; uint32_t u[4]; } b0, b1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { b0.u[k] = pw[k]; b1.u[k] = pw[4 + k]; }
                    // ---- moment MFMA: k order of step s = the candidate order of d1's registers 8s .. 8s+7
                    acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m0.v, b0.v, acc[gi], 0, 0, 0);
                    acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m1.v, b1.v, acc[gi], 0, 0, 0);
                }
            }
#if !GM_NORMALS_PREFETCH
            assemble();
#endif
            GM_PH_STAMP(ph_c3);
            GM_PH_ADD(4, ph_c3 - ph_c2);   // pair loops of the chunk
        }
        // ---- D[row][query]: lane (q, h) of group g holds rows (k & 3) + 8 (k >> 2) + 4 h in acc[g][k].  A lane's home
        // query is query (lane & 31) of group (lane >> 5): it keeps its own half of that group's rows and swaps the
        // other group's registers with lane ^ 32 for the missing half.
        GM_PH_STAMP(ph_t3);
        // (the lane id is taken afresh, behind an empty asm: carried across the candidate stream it was the one register the
        // kernel spilled)
        int lane_e = lane_id();
        asm volatile("" : "+v"(lane_e));
        const int half_e = lane_e >> 5;
        const bool active_e = (uint32_t)lane_e < qn;
        float own[16], got[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            own[k] = half_e ? acc[1][k] : acc[0][k];
            const float snd = half_e ? acc[0][k] : acc[1][k];
            got[k] = __shfl_xor(snd, 32, kWave);
        }
        auto frow = [&](int rr) -> double {  // feature row rr of the home query
            const int k = 4 * (rr >> 3) + (rr & 3), hh = (rr >> 2) & 1;
            return (double)((hh == half_e) ? own[k] : got[k]);
        };
        double mom[10];
        mom[0] = frow(0);
        mom[1] = (frow(kRowX[0]) + frow(kRowX[1])) + frow(kRowX[2]);
        mom[2] = (frow(kRowY[0]) + frow(kRowY[1])) + frow(kRowY[2]);
        mom[3] = (frow(kRowZ[0]) + frow(kRowZ[1])) + frow(kRowZ[2]);
#pragma unroll
        for (int m = 0; m < 6; ++m) mom[4 + m] = (frow(kRowProd0 + 3 * m) + frow(kRowProd0 + 3 * m + 1)) + frow(kRowProd0 + 3 * m + 2);

        // The query is read again here (an L2 hit) rather than kept in registers across the whole candidate stream: the
        // kernel sits at its 128-VGPR budget and everything live across the loop that the loop does not use was being
        // spilled once per tile -- 50 MB of scratch writes per launch.  (The index goes through an empty asm so that the
        // compiler cannot tell it is the load it already has.)
        uint32_t qidx_e = qs + (active_e ? (uint32_t)lane_e : qn - 1u);
        asm volatile("" : "+v"(qidx_e));
        const float4 qe = spts4[qidx_e];
        const bool vox_ok = emit_normal(active_e, qe, mom, vd, normals4, counts, qn, stat_t0, stat_entry);
        if (vd.enabled) voxel_sums(vox_ok, qe, vd, vox_table);
        GM_PH_STAMP(ph_t4);
        GM_PH_ADD(5, ph_t4 - ph_t3);   // moments -> normal, stores, voxel sums
        GM_PH_ADD(6, ph_t4 - ph_t0);   // the whole tile
        GM_PH_ADD(7, 1);
#ifdef GM_NORMALS_PHASES
        if (lane < 8) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v = lane == k ? ph_acc[k] : v;
            atomicAdd(&gm_phase_ticks[lane], (unsigned long long)v);
        }
#endif
    }
}

// ---- the neighbourhood kernel: one launch, one wave per tile, two formulations ------
// Tiles are assigned by wave id (grid-stride; with the default grid every wave gets at most one tile and its block
// retires right after).  No block is long-lived: the hardware block scheduler balances the uneven candidate counts and
// lets the kernels of other frames in flight interleave (a persistent work-queue grid measured 8 % slower alone, 4 %
// slower with three frames in flight).
// XCD-aware tile mapping: the dispatcher deals consecutive blocks round-robin to the 8 XCDs, each with its own L2; the
// tile list is in (nearly) sorted order, so neighbouring tiles share candidate rows.  Blocks are re-labelled bijectively
// so that runs of xcd_chunk consecutive blocks land on ONE XCD (a sorted row is then fetched into one L2 instead of
// all eight), runs are dealt round-robin; the tail that does not fill 8 runs keeps the plain order.
constexpr int kWaveLdsBytes = (kMxWaveLdsBytes + 15) / 16 * 16;

// The tile list as k_normals walks it: class 0 (the costliest tiles) first.  A frame's tiles take 22-52 us each and the
// chip holds 4 096 of them at a time; in the order they were cut -- position order -- the long ones that happen to come
// last drain for 40 us over a mostly empty chip.  pre[c] = tiles in classes before c (scalar registers: the counters are
// read with scalar loads).
struct TileList {
    uint32_t n;
    uint32_t pre[kTileClasses];
};
__device__ __forceinline__ TileList tile_list(const NormalsArgs &A)
{
    TileList L;
    uint32_t run = 0;
#pragma unroll
    for (int c = 0; c < kTileClasses; ++c) {
        uint32_t k = A.ctr->n_tiles_c[c][0];
        const uint32_t cap = c + 1 < kTileClasses ? A.tile_seg : A.tiles_cap;
        if (k > cap) k = cap;   // (a full segment: its further tiles were filed in the last class)
        L.pre[c] = run;
        run += k;
    }
    L.n = run;
    return L;
}
__device__ __forceinline__ uint2 tile_at(const NormalsArgs &A, const TileList &L, uint32_t v)   // v < L.n, wave-uniform
{
    uint32_t c = 0, first = 0;
#pragma unroll
    for (int k = 1; k < kTileClasses; ++k)
        if (v >= L.pre[k]) { c = (uint32_t)k; first = L.pre[k]; }
    return A.tiles[(size_t)c * A.tile_seg + (v - first)];
}

__device__ __forceinline__ uint32_t normals_wave_id(const NormalsArgs &A, uint32_t ntiles)
{
    uint32_t vblock = blockIdx.x;
    const uint32_t nblk = (ntiles + kNrWaves - 1) / kNrWaves;
    if (gridDim.x >= nblk && A.vd.xcd_chunk) {
        if (blockIdx.x >= nblk) return 0xFFFFFFFFu;  // uniform per block: no tile for this block
        const uint32_t cb = A.vd.xcd_chunk, full = nblk / (8u * cb) * (8u * cb);
        if (blockIdx.x < full) {
            const uint32_t xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u;
            vblock = ((slot / cb) * 8u + xcd) * cb + slot % cb;
        }
    }
    // (wave-uniform by construction; saying so keeps it, and everything derived from it, in scalar registers)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(vblock * kNrWaves + threadIdx.x / kWave));
}

__global__ __launch_bounds__(kNrThreads, GM_MX_WAVES) void k_normals_m(NormalsArgs A, uint32_t mx_min_candidates)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kNrWaves][kWaveLdsBytes];
    const TileList L = tile_list(A);   // final before the launch (k_rows_and_tiles)
    const uint32_t ntiles = L.n;
    const uint32_t wave_id = normals_wave_id(A, ntiles), n_waves = gridDim.x * kNrWaves;
    for (uint32_t t = wave_id; t < ntiles; t += n_waves)   // every wave reaches the end: the list is final
        normals_tile_mx(A, lds[threadIdx.x / kWave], tile_at(A, L, t), mx_min_candidates);
}

// THE production kernel: distances AND moments on the matrix cores (k_normals_m = GM_NORMALS_IMPL=auto0 keeps the
// predicate on the VALU: 13 % slower alone, 4 % slower per step with three frames in flight; DESIGN.md par. 4)
// A wave takes the tile of its id; tiles beyond the grid (the grid is capped at 65 536 blocks: frames beyond ~8 M points,
// or GM_NORMALS_BLOCKS in tests) are walked by a second, looped copy of the tile code.  Two copies on purpose: inside a
// loop the compiler hoists every tile-invariant (lane patterns, constants) out of it and, at the 128-VGPR budget, then
// spills them -- once per wave, i.e. once per tile: 80 MB of scratch writes per 1 M-point launch when the loop was the
// only copy.  The straight-line copy spills nothing; the looped one is only entered by the frames that need it.
template <bool FINE>
__global__ __launch_bounds__(kNrThreads, GM_MX_WAVES) void k_normals(NormalsArgs A, uint32_t mx_min_candidates)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kNrWaves][(kMdWaveLdsBytes + 15) / 16 * 16];
#ifdef GM_NORMALS_TIMELINE
    // (the very first instruction of the wave: an asm the compiler may not move the kernel-argument loads above)
    unsigned long long stat_entry;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stat_entry) : : "memory");
#else
    const unsigned long long stat_entry = 0ull;
#endif
    const TileList L = tile_list(A);
    const uint32_t ntiles = L.n;
    const uint32_t wave_id = normals_wave_id(A, ntiles), n_waves = gridDim.x * kNrWaves;
    // (the grid is sized for the most tiles a frame of this size can have: most of its waves find no tile and leave here)
    if (wave_id >= ntiles) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));   // (scalar: the slice's address costs no vector register)
    // slots the staging never writes (past a chunk's end, the two pad rows) are read as MFMA operands whose products are
    // masked or land in unused result rows: they only have to be FINITE
    {
        uint4 *z = reinterpret_cast<uint4 *>(lds[wv]);
        for (int i = lane_id(); i < kMdWaveLdsBytes / 16; i += kWave) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#ifdef GM_NORMALS_TIMELINE
    const uint2 tile_tl = tile_at(A, L, wave_id);
    unsigned long long stat_tile;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stat_tile) : "v"(tile_tl.x) : "memory");   // (after the tile's load has returned)
    normals_tile_mxd<FINE>(A, lds[wv], tile_tl, mx_min_candidates, stat_entry);
    if (wave_id < 65536u && lane_id() == 0) {
        gm_tl_buf[3 * wave_id + 0] = stat_entry;
        gm_tl_buf[3 * wave_id + 1] = stat_tile;
        gm_tl_buf[3 * wave_id + 2] = (unsigned long long)wall_clock64();
    }
#else
    normals_tile_mxd<FINE>(A, lds[wv], tile_at(A, L, wave_id), mx_min_candidates, stat_entry);
#endif
#ifndef GM_NORMALS_NO_LOOP   // (experiment: the kernel without its looped copy -- frames with more tiles than waves unsupported)
    if (ntiles > n_waves && wave_id != 0xFFFFFFFFu)
        for (uint32_t t = wave_id + n_waves; t < ntiles; t += n_waves)
            normals_tile_mxd<FINE>(A, lds[wv], tile_at(A, L, t), mx_min_candidates);
#endif
}

// the all-VALU formulation of every tile (GM_NORMALS_IMPL=valu: A/B measurements and the cross-check in tests)
__global__ __launch_bounds__(kNrThreads) void k_normals_valu(NormalsArgs A)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kNrWaves][(kValuWaveLdsBytes + 15) / 16 * 16];
    const TileList L = tile_list(A);
    const uint32_t ntiles = L.n;
    const uint32_t wave_id = normals_wave_id(A, ntiles), n_waves = gridDim.x * kNrWaves;
    for (uint32_t t = wave_id; t < ntiles; t += n_waves) normals_tile_valu(A, lds[threadIdx.x / kWave], tile_at(A, L, t));
}

// bits needed by the largest cell key
int cell_key_bits(const GridParams &g)
{
    const uint64_t ncell = (uint64_t)g.nx * g.ny * g.nz;
    int bits = 1;
    while ((1ull << bits) < ncell) ++bits;
    return bits;
}

// a double just below 1 / d (k_rows_and_tiles' div_inv)
static double inv_below(uint32_t d) { return nextafter(1.0 / (double)(d ? d : 1u), 0.0); }

uint32_t max_tiles(uint32_t n_cap, const GridParams &g)
{
    // every aligned cell group adds at most one partially filled tile, every 64-chunk at most one more
    const uint64_t span = (uint64_t)kTileSpan * (uint64_t)(g.xreach - 1);
    const uint64_t groups = (uint64_t)g.ny * (uint64_t)g.nz * ((uint64_t)g.nx / (span + 1) + 1);
    const uint64_t extra = (groups < n_cap ? groups : n_cap) + n_cap / kTileQ;
    return (uint32_t)(n_cap / kTileQ + extra + 1);
}

void launch_grid_and_normals(const GridParams &g, const VoxDense &vd, Slot &sl, uint32_t n_cap, bool keep_counts,
                             bool scratch_cleared, hipStream_t s)
{
    if (n_cap == 0) return;
    // cell sort; its last pass writes the sorted cloud (spts4) itself.  scratch_cleared: the crop has counted the digit totals.
    const int where = launch_radix_sort(sl.keys_a, sl.vals_a, sl.keys_b, sl.vals_b, &sl.ctr->n_cropped, n_cap, cell_key_bits(g),
                                        sl, scratch_cleared, s, sl.crop4, sl.spts4);
    uint32_t *skeys = where ? sl.keys_b : sl.keys_a;
    sl.skeys = skeys;
    // rows the frame leaves unoccupied must read as empty ranges in k_normals' window searches
    if (!scratch_cleared) hipMemsetAsync(sl.row_bounds, 0, sizeof(uint2) * (size_t)g.ny * (size_t)g.nz, s);
    {
        ScanState st = next_scan(sl);
        st.status = sl.tile_rec;
        st.ticket = sl.sort.ticket + 1;
        // Tile order.  Cost classes (longest tiles first) shorten the kernel's drain: 147.5 us against 157 in position order,
        // alone on the chip.  They also spread what runs at any one time over the whole frame: 146.6 MB of HBM traffic per
        // launch against 125, and with four frames in flight that costs more than the drain -- which the other frames'
        // kernels fill anyway -- saves: 0.231 ms per step against 0.219 in position order (tools/sweep_classes.sh, one
        // box).  So a context that keeps several frames in flight (n_slots > 1) cuts in position order, a context that runs
        // one frame at a time by cost class.  GM_TILE_ORDER=classes|position overrides.
        static const char *to = getenv("GM_TILE_ORDER");
        const uint32_t single_class = to ? (to[0] == 'p' ? 1u : 0u) : (sl.pipelined ? 1u : 0u);
        hipLaunchKernelGGL(k_rows_and_tiles, dim3(tile_cutter_blocks(n_cap)), dim3(kTbThreads), 0, s, (const uint32_t *)skeys, sl.ctr,
                           (uint32_t)g.nx, (uint32_t)(kTileSpan * (g.xreach - 1)), inv_below((uint32_t)g.nx),
                           inv_below((uint32_t)(kTileSpan * (g.xreach - 1)) + 1u), sl.row_bounds, (uint32_t)g.ny * (uint32_t)g.nz, sl.tiles,
                           sl.tiles_cap, sl.tile_seg, st, single_class);
    }
    // one wave per tile: four tiles per block
    // A wave per tile for twice the tiles of a dense frame (n / 64: full 64-point tiles); a frame with more -- sparse rows
    // cut into many short tiles, at most max_tiles() -- has the rest walked by the kernel's second, looped copy of the
    // tile code.  (Sizing the grid for max_tiles() put 58 000 tile-less blocks behind the 7 400 working ones of the
    // 1 M-point frame, each zeroing its LDS slice before finding out: 0.168 -> 0.165 ms for the kernel alone, 0.279 ->
    // 0.269 ms per step with three frames in flight, once they leave first and most of them are not launched.)
    const uint32_t mt = max_tiles(n_cap, g);
    uint32_t nb = (mt + kNrWaves - 1) / kNrWaves;
    {
        const uint32_t usual = (n_cap / 32u + kNrWaves) / kNrWaves;
        if (nb > usual) nb = usual < 64u ? 64u : usual;
    }
    {
        // one wave per tile up to 65 536 blocks (262 144 tiles: a ~16 M-point frame), grid-stride beyond that.
        // GM_NORMALS_BLOCKS lowers the cap: tests use it to force the grid-stride path on a small frame.
        static const char *e = getenv("GM_NORMALS_BLOCKS");
        const uint32_t cap = e ? (uint32_t)atoi(e) : 65536u;
        if (nb > cap) nb = cap;
    }
    // blocks per XCD chunk (0 = plain round-robin).  Measured on the 1 M frame: 32 keeps the kernel time of the plain
    // mapping with 29 % less L2 fill traffic; one contiguous eighth per XCD fetches 36 % less but runs 4 % longer
    // (the eighths are not equally expensive).
    static const char *xc = getenv("GM_NORMALS_XCD");
    VoxDense vdx = vd;
    vdx.xcd_chunk = xc ? (uint32_t)atoi(xc) : 32u;
    NormalsArgs na;
    na.spts4 = sl.spts4; na.skeys = skeys; na.tiles = sl.tiles; na.ctr = sl.ctr; na.g = g; na.tiles_cap = sl.tiles_cap; na.tile_seg = sl.tile_seg;
    na.row_bounds = sl.row_bounds; na.normals4 = sl.normals4; na.counts = keep_counts ? sl.counts : (int32_t *)nullptr;
    na.vd = vdx; na.vox_table = sl.vox_table;
    // GM_NORMALS_IMPL: auto (default) = moments on the matrix cores except for thin neighbourhoods (fewer than
    // kMxMinCandidates candidates in a tile's windows), which take the kernel's direct fp64 path; mfma = matrix cores for
    // every tile; valu = the all-VALU kernel (A/B measurements, cross-checks in tests)
    static const char *impl = getenv("GM_NORMALS_IMPL");
    const uint32_t mx_min = !impl ? (uint32_t)kMxMinCandidates : (impl[0] == 'v' ? 0xFFFFFFFFu : (impl[0] == 'm' ? 0u : (uint32_t)kMxMinCandidates));
    if (!sl.capturing) hipEventRecord(sl.ev_k0, s);
    // a trailing 0 (auto0 / mfma0) keeps the neighbour predicate on the VALU (k_normals_m: moments only on the matrix cores)
    const bool dist_on_mx = !(impl && strchr(impl, '0'));
    if (mx_min == 0xFFFFFFFFu) hipLaunchKernelGGL(k_normals_valu, dim3(nb), dim3(kNrThreads), 0, s, na);
    else if (dist_on_mx) {
        if (g.D > 1) hipLaunchKernelGGL(k_normals<true>, dim3(nb), dim3(kNrThreads), 0, s, na, mx_min);
        else hipLaunchKernelGGL(k_normals<false>, dim3(nb), dim3(kNrThreads), 0, s, na, mx_min);
    }
    else hipLaunchKernelGGL(k_normals_m, dim3(nb), dim3(kNrThreads), 0, s, na, mx_min);
    if (!sl.capturing) hipEventRecord(sl.ev_k1, s);
}

}  // namespace gm

#ifdef GM_NORMALS_STATS
// diagnostic builds only: raw device counters of a slot (16 words)
extern "C" int gm_debug_counters(gm_ctx *ctx, uint32_t slot, uint32_t *out)
{
    hipDeviceSynchronize();
    gm::DevCounters c;
    const int rc = (int)hipMemcpy(&c, ctx->slots[slot].ctr, sizeof(c), hipMemcpyDeviceToHost);
    // out[0..18): the counters up to pad[]; out[2] = number of tiles (sum over the cost classes; full segments overflow into
    // the last class and are counted there a second time -- none in the frames the diagnostics run on)
    memcpy(out, &c, 18 * sizeof(uint32_t));
    out[2] = 0;
    for (int k = 0; k < gm::kTileListClasses; ++k) out[2] += c.n_tiles_c[k][0];
    return rc;
}
#endif

#ifdef GM_NORMALS_TIMELINE
// diagnostic builds only: (entry, tile known, end) ticks of the first n waves of the last k_normals launch
extern "C" int gm_debug_timeline(unsigned long long *out, uint32_t n)
{
    hipDeviceSynchronize();
    if (n > 65536u) n = 65536u;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gm::gm_tl_buf), sizeof(unsigned long long) * 3 * (size_t)n);
}
#endif

#ifdef GM_NORMALS_PHASES
// diagnostic builds only: read (and clear) the phase tick sums of k_normals
extern "C" int gm_debug_phases(unsigned long long *out16)
{
    hipDeviceSynchronize();
    int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(gm::gm_phase_ticks), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {};
    if (rc == 0) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(gm::gm_phase_ticks), z, sizeof(z));
    return rc;
}
#endif

#ifdef GM_SORT_TIMELINE
// diagnostic builds only: [block][8] ticks (100 MHz) of the last tile cutter launch
extern "C" int gm_debug_cutter_timeline(unsigned long long *out)
{
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gm::gm_cut_tl), sizeof(unsigned long long) * 512 * 8);
}
#endif
