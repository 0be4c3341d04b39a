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
//  * points are binned into a uniform grid (cell edge >= r) by a stable radix
//    sort of cell keys, x fastest, so the 3x3x3 neighbourhood of a run of cells
//    in one x-row is 9 CONTIGUOUS ranges of the sorted array;
//  * a work item ("tile") is <= 64 consecutive sorted points of one x-row: one
//    query per lane, the tile's candidates are streamed through a 4 KiB
//    wave-private SoA LDS window (coalesced 16 B/lane loads in, broadcast
//    ds_read_b128 out: four candidates' x, y or z per read), so no block
//    barrier exists in the hot loop; the wave's four 16-lane groups walk
//    their own x-windows of a row in lock-step;
//  * each lane keeps 10 fp32 accumulators of offsets FROM ITS OWN QUERY POINT
//    (|offset| < r: no cancellation), folded into fp64 once per 64 candidates;
//  * the 3x3 solve runs in fp64 (MI355X fp64 vector rate is half the fp32 rate;
//    ~250 instructions against ~15 000 in the neighbour loop);
//  * one wave per tile, blocks retire after their tile: the hardware block
//    scheduler balances the uneven candidate counts and lets other frames'
//    kernels interleave.
// Bound: fp32 VALU issue (~21 ops per query-candidate pair), not HBM: every
// candidate byte is read once per tile and reused by 64 lanes.
#include <stdlib.h>

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
#define GM_NRTHREADS 256
#endif
constexpr int kNrThreads = GM_NRTHREADS;
constexpr int kTileQ = kWave;             // queries per tile: one per lane
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

// ---- gather the cropped cloud into cell-sorted order ---------------------------
__global__ __launch_bounds__(256) void k_gather_sorted(const float4 *__restrict__ crop4,
                                                       const uint32_t *__restrict__ perm,
                                                       const uint32_t *__restrict__ skeys,
                                                       const uint32_t *__restrict__ n_ptr, uint32_t nx,
                                                       float4 *__restrict__ spts4,
                                                       uint2 *__restrict__ row_bounds /* [ny*nz]: begin, end */)
{
    const uint32_t n = *n_ptr;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const uint32_t i = perm[s];
        float4 p = crop4[i];
        p.w = __uint_as_float(i);  // cropped index rides in the pad lane
        spts4[s] = p;
        // first / one-past-last sorted position of every occupied x-row (only occupied rows are ever read)
        const uint32_t row = skeys[s] / nx;
        if (s == 0 || skeys[s - 1] / nx != row) row_bounds[row].x = s;
        if (s + 1 == n || skeys[s + 1] / nx != row) row_bounds[row].y = s + 1;
    }
}

// ---- tiles: <=64 consecutive sorted points of one x-row spanning <= span+1 cells ----
// Tiles are cut at every 64th point counted from the start of the x-row (so dense rows give
// full tiles: ~90 % lane fill) and a 64-chunk that spans more than `span` cell steps (sparse
// rows) is cut again at aligned (span+1)-cell groups, which bounds the candidate count of a
// tile: without the bound a sparse row yields tiles 20x the mean that the whole grid waits for.
__global__ __launch_bounds__(1024) void k_build_tiles(const uint32_t *__restrict__ skeys,
                                                      DevCounters *__restrict__ ctr, uint32_t nx, uint32_t span,
                                                      const uint2 *__restrict__ row_bounds,
                                                      uint2 *__restrict__ tiles, uint32_t tiles_cap)
{
    __shared__ uint32_t wtot[1024 / kWave];
    __shared__ uint32_t block_base;
    const uint32_t n = ctr->n_cropped;
    const uint32_t s = blockIdx.x * 1024u + threadIdx.x;
    if (blockIdx.x * 1024u >= n) return;  // uniform per block
    const int w = threadIdx.x / kWave;
    // a point starts a tile iff it starts a 64-chunk of its x-row, or its chunk is "sparse"
    // (spans more than `span` cell steps) and it is the first point of an aligned cell group
    uint32_t cnt = 0, tend = 0;
    if (s < n) {
        const uint32_t key = skeys[s];
        const uint32_t row = key / nx;
        const uint2 rb = row_bounds[row];  // written by k_gather_sorted
        const uint32_t cstart = rb.x + ((s - rb.x) / (uint32_t)kTileQ) * (uint32_t)kTileQ;
        const uint32_t cend = (cstart + kTileQ < rb.y) ? cstart + kTileQ : rb.y;
        const bool sparse = skeys[cend - 1] - skeys[cstart] > span;  // same row: key difference = cell steps
        const uint32_t group = span + 1u;
        const uint32_t my_group = (key - row * nx) / group;
        if (s == cstart) cnt = 1;
        else if (sparse && my_group != (skeys[s - 1] - row * nx) / group) cnt = 1;
        if (cnt) {
            // the tile ends with its 64-chunk or, in a sparse chunk, where the next cell group begins
            // (keys ascend inside a row: a short binary search, on the rare sparse chunks only)
            tend = cend;
            if (sparse) {
                uint32_t lo = s + 1, hi = cend;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((skeys[mid] - row * nx) / group == my_group) lo = mid + 1; else hi = mid;
                }
                tend = lo;
            }
        }
    }
    // block-wide exclusive prefix of the flags, ONE atomic per block for the base
    const uint32_t inc = wave_inclusive_scan(cnt);
    if (lane_id() == kWave - 1) wtot[w] = inc;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 1024 / kWave; ++k) {
        const uint32_t c = wtot[k];
        if (k < w) woff += c;
        total += c;
    }
    if (threadIdx.x == 0) block_base = total ? atomicAdd(&ctr->n_tiles, total) : 0u;
    __syncthreads();
    const uint32_t t_out = block_base + woff + inc - cnt;
    if (cnt && t_out < tiles_cap) tiles[t_out] = make_uint2(s, tend - s);  // first query, number of queries
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
        const double sp = sqrt(p);
        double disc = p * p * p - q * q;
        if (disc < 0.0) disc = 0.0;
        const float phi = atan2f((float)sqrt(disc), (float)q) * (1.0f / 3.0f);
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
                const double step = f / fp;
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
    const double inv = 1.0 / sqrt(nn);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
    return true;
}

// ---- the neighbourhood kernel ----------------------------------------------------
__global__ __launch_bounds__(kNrThreads) void k_normals(const float4 *__restrict__ spts4,
                                                        const uint32_t *__restrict__ skeys,
                                                        const uint2 *__restrict__ tiles,
                                                        DevCounters *__restrict__ ctr, GridParams g,
                                                        uint32_t tiles_cap, const uint2 *__restrict__ row_bounds,
                                                        float4 *__restrict__ normals4,
                                                        int32_t *__restrict__ counts, VoxDense vd,
                                                        VoxCell *__restrict__ vox_table)
{
    // candidate window, one per wave, SoA so that one broadcast ds_read_b128 feeds
    // the x (or y, z) of FOUR candidates to every lane
    __shared__ __attribute__((aligned(16))) float win[kNrWaves][3][kWinCap + kWinPad];
    __shared__ double tot[kNrWaves][10][kWave];  // per-lane fp64 moment totals of the current tile
    const int lane = lane_id();
    const int w = threadIdx.x / kWave;
    const uint32_t n = ctr->n_cropped;
    uint32_t ntiles = ctr->n_tiles;
    if (ntiles > tiles_cap) ntiles = tiles_cap;

    // Tiles are assigned by wave id (grid-stride; with the default grid every wave gets at most one tile and
    // its block retires right after).  The hardware block scheduler does the load balancing, and because no
    // block is long-lived the kernels of other frames in flight get wave slots as blocks retire -- a
    // persistent work-queue grid measured 8 % slower alone and 4 % slower with three frames in flight.
    // XCD-aware tile mapping.  The dispatcher deals consecutive blocks round-robin to the 8 XCDs, each with its own
    // L2; the tile list is in (nearly) sorted order, so neighbouring tiles share candidate rows.  With one wave per
    // tile the blocks that have work (the first ceil(ntiles / 4)) are re-labelled bijectively so that runs of
    // consecutive tiles land on ONE XCD: a sorted row is then fetched into one L2 instead of all eight.
    uint32_t vblock = blockIdx.x;
    const uint32_t nblk = (ntiles + kNrWaves - 1) / kNrWaves;
    if (gridDim.x >= nblk && vd.xcd_chunk) {
        if (blockIdx.x >= nblk) return;  // uniform per block: no tile for this block
        // chunks of xcd_chunk consecutive blocks (4 tiles each) go to one XCD, chunks are dealt round-robin: locality
        // inside a chunk, balance across the XCDs; the tail that does not fill 8 chunks keeps the plain mapping
        const uint32_t cb = vd.xcd_chunk, full = nblk / (8u * cb) * (8u * cb);
        if (blockIdx.x < full) {
            const uint32_t xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u;
            vblock = ((slot / cb) * 8u + xcd) * cb + slot % cb;
        }
    }
    const uint32_t wave_id = vblock * kNrWaves + (uint32_t)w, n_waves = gridDim.x * kNrWaves;
    for (uint32_t iter = 0;; ++iter) {
        const uint32_t t = wave_id + iter * n_waves;
        if (t >= ntiles) break;  // every wave reaches this: the tile list is final before the launch
        const uint2 tile = tiles[t];
        const uint32_t qs = tile.x, qn = tile.y;  // first query (sorted position), number of queries (1..64)
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
            continue;
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
                    atomicAdd(&ctr->reserved0, 1u);                                // chunks
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
        const int cnt = (int)Sn;

        bool vox_ok = false;
        if (active) {
            const uint32_t dst = __float_as_uint(q.w);
            float4 out = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            if (cnt >= 3) {  // NormalEstimation::computePointNormal
                const double inv_n = 1.0 / (double)cnt;
                const double mx = Sx * inv_n, my = Sy * inv_n, mz = Sz * inv_n;
                double c[6];
                c[0] = Sxx * inv_n - mx * mx; c[1] = Sxy * inv_n - mx * my; c[2] = Sxz * inv_n - mx * mz;
                c[3] = Syy * inv_n - my * my; c[4] = Syz * inv_n - my * mz; c[5] = Szz * inv_n - mz * mz;
                double lam, nv[3];
                if (smallest_eigpair(c, lam, nv)) {
                    const double trc = c[0] + c[3] + c[5];
                    const double curv = (trc != 0.0) ? fabs(lam / trc) : 0.0;
                    // flipNormalTowardsViewpoint(p, 0,0,0)
                    const double ct = -((double)q.x * nv[0] + (double)q.y * nv[1] + (double)q.z * nv[2]);
                    const double sgn = (ct < 0.0) ? -1.0 : 1.0;
                    out = make_float4((float)(sgn * nv[0]), (float)(sgn * nv[1]), (float)(sgn * nv[2]), (float)curv);
                    vox_ok = finite3(out.x, out.y, out.z) && q.x >= vd.own_lo && q.x < vd.own_hi;
                }
            }
            normals4[dst] = out;
            if (counts) counts[dst] = cnt;
#ifdef GM_NORMALS_TIMELINE  // diagnostic build (tools/tile_timeline.py): lanes 0 / 1 of a tile report its end tick / duration
            {
                const unsigned long long t1 = wall_clock64();
                if (counts && qn >= 2 && lane == 0) counts[dst] = -(int)((t1 & 0x1FFFFFFFull) | 0x20000000ull);  // end tick, bit 29 set
                if (counts && qn >= 2 && lane == 1) counts[dst] = -(int)((t1 - stat_t0) & 0xFFFFFull) - 1;        // duration < 2^20 ticks
            }
#endif
        }

        // ---- VoxelGrid fast path: points of a tile are spatial neighbours, so they fall
        // into one to three voxels; reduce by voxel inside the wave, then one set of
        // integer atomics per (wave, voxel).  (pcl::VoxelGrid, src/tunnel_processing.cpp:217-220)
        if (vd.enabled) {
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
                const unsigned long long ax = wave_sum(mine ? fx : 0ull);
                const unsigned long long ay = wave_sum(mine ? fy : 0ull);
                const unsigned long long az = wave_sum(mine ? fz : 0ull);
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
    }
}

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
    // bits needed by the largest cell key
    const uint64_t ncell = (uint64_t)g.nx * g.ny * g.nz;
    int bits = 1;
    while ((1ull << bits) < ncell) ++bits;
    const int where = launch_radix_sort(sl.keys_a, sl.vals_a, sl.keys_b, sl.vals_b, &sl.ctr->n_cropped, n_cap, bits,
                                        sl.sort, scratch_cleared, s);
    uint32_t *skeys = where ? sl.keys_b : sl.keys_a;
    uint32_t *perm = where ? sl.vals_b : sl.vals_a;
    sl.skeys = skeys;
    const uint32_t gb = (n_cap + 255) / 256 < 2048 ? (n_cap + 255) / 256 : 2048;
    // rows the frame leaves unoccupied must read as empty ranges in k_normals' window searches
    if (!scratch_cleared) hipMemsetAsync(sl.row_bounds, 0, sizeof(uint2) * (size_t)g.ny * (size_t)g.nz, s);
    hipLaunchKernelGGL(k_gather_sorted, dim3(gb), dim3(256), 0, s, (const float4 *)sl.crop4, (const uint32_t *)perm,
                       (const uint32_t *)skeys, (const uint32_t *)&sl.ctr->n_cropped, (uint32_t)g.nx, sl.spts4,
                       sl.row_bounds);
    hipLaunchKernelGGL(k_build_tiles, dim3((n_cap + 1023) / 1024), dim3(1024), 0, s, (const uint32_t *)skeys, sl.ctr, (uint32_t)g.nx,
                       (uint32_t)(kTileSpan * (g.xreach - 1)), (const uint2 *)sl.row_bounds, sl.tiles, sl.tiles_cap);
    // one wave per tile: four tiles per block
    const uint32_t mt = max_tiles(n_cap, g);
    uint32_t nb = (mt + kNrWaves - 1) / kNrWaves;
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
    hipEventRecord(sl.ev_k0, s);
    hipLaunchKernelGGL(k_normals, dim3(nb), dim3(kNrThreads), 0, s, (const float4 *)sl.spts4, (const uint32_t *)skeys,
                       (const uint2 *)sl.tiles, sl.ctr, g, sl.tiles_cap, (const uint2 *)sl.row_bounds, sl.normals4,
                       keep_counts ? sl.counts : (int32_t *)nullptr, vdx, sl.vox_table);
    hipEventRecord(sl.ev_k1, s);
}

}  // namespace gm
