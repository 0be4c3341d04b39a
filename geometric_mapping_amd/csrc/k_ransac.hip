// k_ransac.hip -- BUILD-DEFINED EXTENSIONS: batched RANSAC plane / cylinder
// hypothesis scoring, inlier labelling, per-segment moment accumulation, refits.
//
// The reference has NO counterpart: getCylinder is an empty commented stub
// (/root/reference src/tunnel_processing.cpp:149-154, tunnel_processing.hpp:56-59),
// its publisher and parameter are commented out (src/geometric_mapping.cpp:41,
// 119-122,163-166).  Conventions follow PCL's SampleConsensusModelPlane /
// SampleConsensusModelCylinder (SURVEY.md par. 8a-ext); the CPU restatement is
// oracle/gm_oracle_ext.c.  Parity for everything here is "vs the build's own
// restatement + analytic truth", never "vs reference".
//
// Shape of the scorers (k_score / k_score_sel, the only ones launched):
//   lane <-> hypothesis: a lane keeps ITS one or two hypotheses in registers with a private integer
//   inlier counter; a block's points are staged once in LDS (coalesced 16 B loads, masked points
//   become NaN) as groups of four, SoA inside a group, and every lane walks them through broadcast
//   ds_read_b128 -- the inner loop is pure VALU (plane: 3 fma + compare + add-with-carry), no
//   ballots, no SALU, no atomics; one integer atomicAdd per (block, hypothesis) closes a block.
//   Counts are integers and integer adds commute: results are run-to-run identical.
//   In-frame scoring is preemptive (launch_score_preemptive): all H hypotheses on every 64th point
//   -> the 128 best -> those on every 16th point -> the 8 best -> those on every point; the top-K
//   selection of a stage runs in the last block of the scoring launch before it (a done-counter and a
//   4-bit radix select over keys held in registers), so a model costs five launches.
//   Bound: fp32 VALU for the scorers, HBM for the label pass (which also sums the segment's moments).
#include "gm_internal.hpp"

namespace gm {

constexpr int kScThreads = 256;
constexpr int kScP = 1;                       // points staged per thread (small blocks: short tail)
constexpr int kScTile = kScThreads * kScP;    // points per block
constexpr int kMaxDraws = 64;

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ inline uint32_t draw_index(uint64_t seed, uint32_t h, uint32_t t, uint32_t n)
{
    const uint64_t u = mix64(seed ^ mix64(((uint64_t)h << 8) | t));
    return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
}

__device__ inline uint32_t next_sample(uint64_t seed, uint32_t h, uint32_t &t, uint32_t n,
                                       const uint8_t *__restrict__ labels, uint32_t want, uint32_t a, uint32_t b)
{
    while (t < (uint32_t)kMaxDraws) {
        const uint32_t i = draw_index(seed, h, t++, n);
        if (i == a || i == b) continue;
        if (labels && labels[i] != want) continue;
        return i;
    }
    return 0xFFFFFFFFu;
}

__device__ inline void store_nan(float *o, int k)
{
    for (int i = 0; i < k; ++i) o[i] = __builtin_nanf("");
}

// inlier band of a cylinder hypothesis: (r-tau)^2 < dist_axis^2 < (r+tau)^2, both ends in fp32
__host__ __device__ inline void cyl_band(float r, double tau, float &lo2, float &hi2)
{
    const double lo = (double)r - tau, hi = (double)r + tau;
    lo2 = lo > 0 ? (float)(lo * lo) : -1.0f;
    hi2 = (float)(hi * hi);
}

// hyp8 rows: plane a,b,c,d,-,-,-,-   cylinder px,py,pz,dx,dy,dz,r,-
// (zero_counts != nullptr: the kernel also clears the score counter of its hypothesis, so the frame pipeline needs no memset)
__global__ __launch_bounds__(256) void k_plane_hypotheses(const float4 *__restrict__ pts,
                                                          const uint8_t *__restrict__ labels, uint32_t want,
                                                          const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                          uint64_t seed, uint32_t H, float *__restrict__ hyp8,
                                                          int32_t *__restrict__ zero_counts)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    if (zero_counts) zero_counts[h] = 0;
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    float *o = hyp8 + 8 * (size_t)h;
    for (int k = 4; k < 8; ++k) o[k] = 0.f;
    if (n < 3) { store_nan(o, 4); return; }
    uint32_t t = 0;
    const uint32_t none = 0xFFFFFFFFu;
    const uint32_t i0 = next_sample(seed, h, t, n, labels, want, none, none);
    const uint32_t i1 = i0 == none ? none : next_sample(seed, h, t, n, labels, want, i0, none);
    const uint32_t i2 = i1 == none ? none : next_sample(seed, h, t, n, labels, want, i0, i1);
    if (i2 == none) { store_nan(o, 4); return; }
    const float4 p0 = pts[i0], p1 = pts[i1], p2 = pts[i2];
    const double ax = (double)p1.x - p0.x, ay = (double)p1.y - p0.y, az = (double)p1.z - p0.z;
    const double bx = (double)p2.x - p0.x, by = (double)p2.y - p0.y, bz = (double)p2.z - p0.z;
    double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    if (!(len > 1e-12)) { store_nan(o, 4); return; }
    nx /= len; ny /= len; nz /= len;
    const double d = -(nx * p0.x + ny * p0.y + nz * p0.z);
    o[0] = (float)nx; o[1] = (float)ny; o[2] = (float)nz; o[3] = (float)d;
}

__device__ inline void cylinder_hypothesis(const float4 *__restrict__ pts, const float4 *__restrict__ nrm,
                                           const uint8_t *__restrict__ labels, uint32_t want, uint32_t n, uint64_t seed,
                                           uint32_t h, float *o /* [8] */)
{
    o[7] = 0.f;
    if (n < 2) { store_nan(o, 7); return; }
    uint32_t t = 0;
    const uint32_t none = 0xFFFFFFFFu;
    const uint32_t i0 = next_sample(seed, h, t, n, labels, want, none, none);
    const uint32_t i1 = i0 == none ? none : next_sample(seed, h, t, n, labels, want, i0, none);
    if (i1 == none) { store_nan(o, 7); return; }
    const float4 P1 = pts[i0], P2 = pts[i1], N1 = nrm[i0], N2 = nrm[i1];
    const double p1[3] = {P1.x, P1.y, P1.z}, p2[3] = {P2.x, P2.y, P2.z};
    const double n1[3] = {N1.x, N1.y, N1.z}, n2[3] = {N2.x, N2.y, N2.z};
    // closest points of the two normal lines (p1+n1)+s*n1 and p2+t*n2
    const double w[3] = {n1[0] + p1[0] - p2[0], n1[1] + p1[1] - p2[1], n1[2] + p1[2] - p2[2]};
    const double a = n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2];
    const double b = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
    const double c = n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2];
    const double d = n1[0] * w[0] + n1[1] * w[1] + n1[2] * w[2];
    const double e = n2[0] * w[0] + n2[1] * w[1] + n2[2] * w[2];
    const double den = a * c - b * b;
    double sc, tc;
    if (den < 1e-8) { sc = 0.0; tc = (b > c ? d / b : e / c); }
    else { sc = (b * e - c * d) / den; tc = (a * e - b * d) / den; }
    double lp[3], ld[3];
    for (int k = 0; k < 3; ++k) lp[k] = p1[k] + n1[k] + sc * n1[k];
    for (int k = 0; k < 3; ++k) ld[k] = p2[k] + tc * n2[k] - lp[k];
    const double len = sqrt(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
    if (!(len > 1e-9) || !isfinite(len)) { store_nan(o, 7); return; }
    for (int k = 0; k < 3; ++k) ld[k] /= len;
    const double v[3] = {p1[0] - lp[0], p1[1] - lp[1], p1[2] - lp[2]};
    const double tt = v[0] * ld[0] + v[1] * ld[1] + v[2] * ld[2];
    const double q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - tt * tt;
    const double r = sqrt(q > 0 ? q : 0);
    o[0] = (float)lp[0]; o[1] = (float)lp[1]; o[2] = (float)lp[2];
    o[3] = (float)ld[0]; o[4] = (float)ld[1]; o[5] = (float)ld[2];
    o[6] = (float)r;
}

__global__ __launch_bounds__(256) void k_cylinder_hypotheses(const float4 *__restrict__ pts,
                                                             const float4 *__restrict__ nrm,
                                                             const uint8_t *__restrict__ labels, uint32_t want,
                                                             const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                             uint64_t seed, uint32_t H, float *__restrict__ hyp8,
                                                             int32_t *__restrict__ zero_counts, float2 *__restrict__ band,
                                                             double tau)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    float o[8];
    cylinder_hypothesis(pts, nrm, labels, want, n, seed, h, o);
    float4 *dst = reinterpret_cast<float4 *>(hyp8 + 8 * (size_t)h);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    if (zero_counts) zero_counts[h] = 0;
    if (band) {  // the scorer's inlier band of this hypothesis (k_cyl_bands for caller-supplied hypotheses)
        float lo2, hi2;
        cyl_band(o[6], tau, lo2, hi2);
        band[h] = make_float2(lo2, hi2);
    }
}

__global__ __launch_bounds__(256) void k_cyl_bands(const float *__restrict__ hyp8, uint32_t H, double tau,
                                                   float2 *__restrict__ band)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    float lo2, hi2;
    cyl_band(hyp8[8 * (size_t)h + 6], tau, lo2, hi2);
    band[h] = make_float2(lo2, hi2);
}

__device__ __forceinline__ bool plane_inlier(float x, float y, float z, float a, float b, float c, float d, float tau)
{
    const float dist = __fmaf_rn(a, x, __fmaf_rn(b, y, __fmaf_rn(c, z, d)));
    return fabsf(dist) < tau;  // NaN hypothesis or masked (NaN) point -> false
}

__device__ __forceinline__ bool cyl_inlier(float x, float y, float z, float px, float py, float pz, float dx, float dy,
                                           float dz, float lo2, float hi2)
{
    const float vx = __fsub_rn(x, px), vy = __fsub_rn(y, py), vz = __fsub_rn(z, pz);
    const float t = __fmaf_rn(vx, dx, __fmaf_rn(vy, dy, __fmul_rn(vz, dz)));
    const float vv = __fmaf_rn(vx, vx, __fmaf_rn(vy, vy, __fmul_rn(vz, vz)));
    const float q = __fmaf_rn(-t, t, vv);
    return q > lo2 && q < hi2;
}

// ---- top-K selection by the last block of a scoring launch -----------------------------------------------------------
// A scoring kernel's blocks add their inlier counts with integer atomics; the block that finishes last (a done-counter)
// picks the K best of the M scored candidates for the next stage, so no selection kernel sits between two scoring
// stages.  Order = count descending, hypothesis index ascending, i.e. descending 64-bit keys
// (count << 13 | 8191 - index) -- all different.  Only the SET matters downstream (the final winner is taken by count
// and index, not by position), so nothing is sorted: a bitwise radix select finds the K-th largest key (one block-wide
// count per key bit, ~27 of them) and the candidates at or above it are written in candidate order.
// Candidate i has count counts[i] and hypothesis index ids ? ids[i] : i (< kMaxHypotheses = 8192);
// sel[0..K) = the selected hypothesis indices, counts_out[0..K) = 0 (the next stage's counters).
struct SelectNext {
    uint32_t *done;            // zero between launches (the last block resets it)
    const uint32_t *ids;
    uint32_t M, K;
    uint32_t *sel;             // nullptr: no selection folded into this launch
    int32_t *counts_out;
    uint32_t zero_n;           // words of counts_out to clear (0: K) -- the streaming last stage keeps replicated counters
};
inline uint32_t select_lds_bytes(uint32_t) { return 0u; }   // (the keys live in registers)
constexpr uint32_t kSelectFoldMax = 2048;  // candidates (8 keys per thread of a 256-thread block); above that a k_select_topk launch does it
static_assert(kMaxHypotheses <= 8192, "select keys keep the hypothesis index in 13 bits");

__device__ __forceinline__ void select_by_last_block(const int32_t *__restrict__ counts, const SelectNext nx,
                                                     unsigned long long * /* unused */)
{
    __shared__ uint32_t s_last;
    __shared__ unsigned long long s_or;
    // This thread's count atomics must be performed before the block reports in.  They are agent-scope RMWs (done at
    // the coherence point, acknowledged through vmcnt) and the last block reads the counters behind an agent-scope
    // acquire, so waiting for the acknowledgements is all that is needed -- a __threadfence() here would also write back
    // the whole L2 from every wave of the grid (measured: +30 us on a 20 us launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = gridDim.x * gridDim.y;
        const uint32_t t = atomicAdd(nx.done, 1u);
        s_last = t == total - 1u ? 1u : 0u;
        if (s_last) atomicExch(nx.done, 0u);
        s_or = 0ull;
    }
    __syncthreads();
    if (!s_last) return;
    // acquire at agent scope (drops this XCD's stale L2 lines -- one block does this, once): the counters can then be
    // read with plain loads, all in flight together, instead of one agent-scope atomic load after the other
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // From here on the block is alone on the chip and every dependent step (an LDS round trip, a barrier) costs its
    // full latency -- measured ~0.5 us per barrier-separated step -- so the search takes 4 key bits per step (a 16-bin
    // histogram in LDS, one barrier), the keys stay in registers, and the output ranks need one barrier in all.
    constexpr int kSelKeys = kSelectFoldMax / 256;   // keys per thread (blocks of >= 256 threads)
    constexpr int kMaxWaves = 16;
    __shared__ uint32_t s_hist[16][16], s_cnt[kSelKeys][kMaxWaves];
    const uint32_t T = blockDim.x, nw = T / kWave;
    const int w = threadIdx.x / kWave;
    unsigned long long key[kSelKeys];
    uint32_t cnt_raw[kSelKeys];
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {   // all loads in flight together
        const uint32_t i = (uint32_t)q * T + threadIdx.x;
        cnt_raw[q] = i < nx.M ? (uint32_t)counts[i] : 0u;
    }
    unsigned long long my_or = 0ull;
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {
        const uint32_t i = (uint32_t)q * T + threadIdx.x;
        key[q] = 0ull;
        if (i < nx.M) {
            const uint32_t h = nx.ids ? nx.ids[i] : i;
            key[q] = ((unsigned long long)cnt_raw[q] << 13) | (unsigned long long)(8191u - (h & 8191u));
        }
        my_or |= key[q];
    }
    if (my_or) atomicOr(&s_or, my_or);
    for (uint32_t t = threadIdx.x; t < 256u; t += T) s_hist[t >> 4][t & 15u] = 0u;
    for (uint32_t r = threadIdx.x; r < (nx.zero_n ? nx.zero_n : nx.K); r += T) nx.counts_out[r] = 0;
    __syncthreads();
    unsigned long long kth = 0ull;  // K >= M: everything is selected
    if (nx.K < nx.M) {
        uint32_t need = nx.K;
        for (int d = (63 - __builtin_clzll(s_or | 1ull)) >> 2; d >= 0; --d) {
            const int shift = 4 * d;
#pragma unroll
            for (int q = 0; q < kSelKeys; ++q) {
                if ((uint32_t)q * T >= nx.M) break;   // block-uniform
                const bool in = (uint32_t)q * T + threadIdx.x < nx.M;
                // keys that share the digits found so far
                const bool match = shift + 4 >= 64 || (key[q] >> (shift + 4)) == (kth >> (shift + 4));
                if (in && match) atomicAdd(&s_hist[d][(uint32_t)(key[q] >> shift) & 15u], 1u);
            }
            __syncthreads();
            uint32_t bins[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) bins[v] = s_hist[d][v];
            // the K-th largest has the highest digit v for which (#keys with a digit above v) < need
            uint32_t above = 0, digit = 0, sub = 0;
            bool found = false;
#pragma unroll
            for (int v = 15; v >= 0; --v) {
                if (!found && above + bins[v] >= need) { digit = (uint32_t)v; sub = above; found = true; }
                above += bins[v];
            }
            need -= sub;
            kth |= (unsigned long long)digit << shift;
        }
    }
    // keys >= kth, in candidate order: candidate q*T + t comes after every candidate of the chunks before q
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {
        const bool take = (uint32_t)q * T + threadIdx.x < nx.M && key[q] >= kth;
        const uint64_t m = __ballot(take);
        if (lane_id() == 0) s_cnt[q][w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t running = 0;
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {
        if ((uint32_t)q * T >= nx.M) break;   // block-uniform
        const bool take = (uint32_t)q * T + threadIdx.x < nx.M && key[q] >= kth;
        const uint64_t m = __ballot(take);
        uint32_t woff = 0, tot = 0;
        for (uint32_t k = 0; k < nw; ++k) {
            const uint32_t c = s_cnt[q][k];
            if ((int)k < w) woff += c;
            tot += c;
        }
        const uint32_t dst = running + woff + (uint32_t)__popcll(m & lanemask_lt());
        if (take && dst < nx.K) nx.sel[dst] = 8191u - (uint32_t)(key[q] & 8191ull);
        running += tot;
    }
}

// MODEL 0: plane, 1: cylinder.  Block (x, y) scores hypotheses [y*256, y*256+256) against points
// [x*kScTile, +kScTile).  lane <-> hypothesis: each lane keeps ITS hypothesis in registers and a
// private inlier counter; the block's points are staged once in LDS (coalesced 16 B loads, masked
// points become NaN) and every lane walks them through broadcast ds_read_b128.  The inner loop is
// pure VALU (3 fma + cmp + add-carry for a plane): no ballots, no SALU, no atomics; one integer
// atomicAdd per (block, hypothesis) at the end (integer adds commute: exact, run-to-run identical).
constexpr int kScHPL = 2;                      // hypotheses per lane: amortises each LDS point read
constexpr int kScHC = kScThreads * kScHPL;     // hypotheses per block

template <int MODEL>
__global__ __launch_bounds__(kScThreads) void k_score(const float4 *__restrict__ pts,
                                                      const uint8_t *__restrict__ labels, uint32_t want,
                                                      const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                      const float *__restrict__ hyp8,
                                                      const float2 *__restrict__ band, uint32_t H, float tau,
                                                      uint32_t stride, uint32_t tile, int32_t *__restrict__ counts, SelectNext nx)
{
    extern __shared__ unsigned long long sel_keys[];
    // points in LDS as groups of four, SoA inside a group: x0..x3 | y0..y3 | z0..z3, so three
    // broadcast ds_read_b128 deliver four points
    __shared__ float4 lp[kScTile / 4][3];
    // with stride > 1 only every stride-th point is scored (the pre-selection stage of the in-frame RANSAC)
    // tile (<= kScTile, a multiple of 4): points per block.  A pre-selection stage sees a few thousand points: small tiles
    // spread them over every CU instead of a long loop on a fifth of them.
    const uint32_t n_full = n_ptr ? *n_ptr : n_host;
    const uint32_t n = (n_full + stride - 1) / stride;
    const uint32_t base = blockIdx.x * tile;
    if (base < n) {  // uniform per block
    const uint32_t m = (n - base < tile) ? n - base : tile;
    const uint32_t groups = (m + 3u) >> 2;
#pragma unroll
    for (int p = 0; p < kScP; ++p) {
        const uint32_t j = p * kScThreads + threadIdx.x;
        const uint32_t i = (base + j) * stride;
        bool ok = j < m;
        if (ok && labels) ok = labels[i] == want;
        float4 v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), 0.f);
        if (ok) v = pts[i];  // masked / out-of-range points are NaN: never an inlier
        float *g = reinterpret_cast<float *>(&lp[j >> 2][0]);
        if (j < ((m + 3u) & ~3u)) { g[(j & 3)] = v.x; g[4 + (j & 3)] = v.y; g[8 + (j & 3)] = v.z; }
    }
    float hp[kScHPL][8];
    float c[kScHPL];  // float counters (exact: <= kScTile per block): keeps the compiler on cmp+cndmask+add
#pragma unroll
    for (int k = 0; k < kScHPL; ++k) {
        const uint32_t h = blockIdx.y * kScHC + k * kScThreads + threadIdx.x;
        c[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) hp[k][q] = 0.f;
        hp[k][0] = __builtin_nanf("");  // lanes past H score nothing
        if (h < H) {
            const float4 a = *reinterpret_cast<const float4 *>(hyp8 + 8 * (size_t)h);
            hp[k][0] = a.x; hp[k][1] = a.y; hp[k][2] = a.z; hp[k][3] = a.w;
            if (MODEL == 1) {
                const float4 b = *reinterpret_cast<const float4 *>(hyp8 + 8 * (size_t)h + 4);
                const float2 bd = band[h];
                hp[k][4] = b.x; hp[k][5] = b.y; hp[k][6] = bd.x; hp[k][7] = bd.y;
            }
        }
    }
    __syncthreads();
    for (uint32_t gi = 0; gi < groups; ++gi) {
        const float4 X = lp[gi][0], Y = lp[gi][1], Z = lp[gi][2];  // broadcast reads: 4 points
        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int k = 0; k < kScHPL; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODEL == 0)
                    c[k] += plane_inlier(xs[q], ys[q], zs[q], hp[k][0], hp[k][1], hp[k][2], hp[k][3], tau) ? 1.0f : 0.0f;
                else
                    c[k] += cyl_inlier(xs[q], ys[q], zs[q], hp[k][0], hp[k][1], hp[k][2], hp[k][3], hp[k][4], hp[k][5],
                                       hp[k][6], hp[k][7]) ? 1.0f : 0.0f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kScHPL; ++k) {
        const uint32_t h = blockIdx.y * kScHC + k * kScThreads + threadIdx.x;
        if (h < H && c[k] > 0.f) atomicAdd(&counts[h], (int32_t)c[k]);
    }
    }
    if (nx.sel) select_by_last_block(counts, nx, sel_keys);
}

// arg-max over the hypothesis counts (largest count, lowest index on ties)
__global__ __launch_bounds__(1024) void k_best_hypothesis(const int32_t *__restrict__ counts, uint32_t H,
                                                          uint32_t *__restrict__ best /* [2]: index, count */)
{
    __shared__ uint32_t bc[1024], bi[1024];
    uint32_t my_c = 0, my_i = 0xFFFFFFFFu;
    for (uint32_t h = threadIdx.x; h < H; h += 1024) {
        const uint32_t s = (uint32_t)counts[h];
        if (my_i == 0xFFFFFFFFu || s > my_c) { my_c = s; my_i = h; }
    }
    bc[threadIdx.x] = my_c; bi[threadIdx.x] = my_i;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const uint32_t oc = bc[threadIdx.x + st], oi = bi[threadIdx.x + st];
            if (oi != 0xFFFFFFFFu && (bi[threadIdx.x] == 0xFFFFFFFFu || oc > bc[threadIdx.x] ||
                                      (oc == bc[threadIdx.x] && oi < bi[threadIdx.x]))) {
                bc[threadIdx.x] = oc; bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best[0] = bi[0]; best[1] = bc[0]; }
}

// ---- preemptive scoring (successive thirds): every stage costs about the same number of point-hypothesis tests ----
//   stage 1: all H hypotheses on every 64th point            -> the 128 best (count desc, hypothesis index asc)
//   stage 2: those 128 on every 16th point                    -> the 8 best
//   stage 3: those 8 on every point                           -> the winner (taken by the label kernel)
// (H <= 128 starts at stage 2, H <= 8 is scored exhaustively.)  H = 1024: 16 n + 8 n + 8 n point-hypothesis tests for a
// frame of n points; the first version of the scheme (every 32nd / 8th point, 256 / 32 kept) spent 96 n -- a fifth of
// the instructions of k_normals, which it shares the chip with when frames are in flight.  A sample of n / 64 points
// (13 000 of the 1 M-point frame) ranks hypotheses whose inlier counts differ by a few per cent reliably; the kept sets
// are deep enough (128, 8) that the eventual winner is not lost to sampling noise (tests/test_gpu_ext.py).
constexpr int kPre1Stride = 64, kPre1Keep = 128;
constexpr int kPre2Stride = 16, kPre2Keep = 8;
// scratch layout (uint32 words): selA[256] cntA[256] selB[32] cntB[32]
constexpr int kPreSelA = 0, kPreCntA = 256, kPreSelB = 512, kPreCntB = 544;  // 576 words, then
constexpr int kPreDone = 576;  // the done-counter of select_by_last_block: zero between launches (gm_ensure_ext allocates 4096 zeroed words)
constexpr int kPreCntStream = 1024;  // the streaming last stage's replicated counters: kStReplicas x kStReplicaStride words

// Top-K of M scored candidates.  Candidate i has count counts[i] and hypothesis index ids ? ids[i] : i; order =
// count descending, hypothesis index ascending.  sel[rank] = hypothesis index; block 0 also clears counts_out[0..K).
__global__ __launch_bounds__(256) void k_select_topk(const int32_t *__restrict__ counts, const uint32_t *__restrict__ ids,
                                                     uint32_t M, uint32_t K, uint32_t *__restrict__ sel,
                                                     int32_t *__restrict__ counts_out)
{
    // 16 candidates per block, 16 lanes per candidate: every lane ranks its candidate against 1/16 of the
    // list (staged in LDS), a 4-step shuffle sum joins the partial ranks
    extern __shared__ int32_t lc[];  // [M] counts, then [M] hypothesis indices
    uint32_t *li = reinterpret_cast<uint32_t *>(lc + M);
    for (uint32_t i = threadIdx.x; i < M; i += 256) { lc[i] = counts[i]; li[i] = ids ? ids[i] : i; }
    if (blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < K; k += 256) counts_out[k] = 0;  // the next stage's counters start at zero (not the replicated ones: see launch_score_preemptive)
    __syncthreads();
    const uint32_t i = blockIdx.x * 16u + threadIdx.x / 16u, part = threadIdx.x & 15u;
    const int32_t c = i < M ? lc[i] : 0;
    const uint32_t h = i < M ? li[i] : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (uint32_t o = part; o < M; o += 16u) {
        const int32_t co = lc[o];
        rank += (co > c || (co == c && li[o] < h)) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) rank += __shfl_xor(rank, o, kWave);
    if (i < M && part == 0 && rank < K) sel[rank] = h;  // ranks are a permutation: slots < min(K,M) are written once
}

// Later stages: K selected hypotheses against every stride-th point.  Same shape as k_score: lane <-> selected
// hypothesis (in registers; blockIdx.y picks the chunk of 64), each of the block's 4 waves owns 256 points staged
// in LDS as SoA groups of four, pure-VALU inner loop, LDS reduce over the waves, one integer atomicAdd per
// (block, hypothesis).
template <int MODEL, int PTS /* points per wave: 256, or 64 to spread a small stage over the chip */>
__global__ __launch_bounds__(256) void k_score_sel(const float4 *__restrict__ pts, const uint8_t *__restrict__ labels,
                                                   uint32_t want, const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                   const float *__restrict__ hyp8, const float2 *__restrict__ band,
                                                   const uint32_t *__restrict__ sel, uint32_t K, float tau,
                                                   uint32_t stride, int32_t *__restrict__ counts_k, SelectNext nx)
{
    extern __shared__ unsigned long long sel_keys[];
    __shared__ float4 lp[4][PTS / 4][3];
    __shared__ float red[4][64];
    const uint32_t n_full = n_ptr ? *n_ptr : n_host;
    const uint32_t n = (n_full + stride - 1) / stride;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // K <= 32 (the last stage): the wave is cut into 64 / P parts (P = the power of two >= K) that score the SAME K
    // hypotheses on different parts of the wave's points, so no lane idles in the stage that sees every point;
    // otherwise one hypothesis per lane, 64 per blockIdx.y
    const bool split = K <= 32u;
    uint32_t P = 64u;
    if (split) { P = 1u; while (P < K) P <<= 1; }
    const uint32_t parts = 64u / P;
    const uint32_t slot = split ? (lane & (P - 1u)) : blockIdx.y * 64u + lane;  // which selected hypothesis this lane scores
    const uint32_t base = (blockIdx.x * 4u + wave) * PTS;
    const uint32_t m = base >= n ? 0u : ((n - base < (uint32_t)PTS) ? n - base : (uint32_t)PTS);
    const uint32_t groups = (m + 3u) >> 2;
    if (blockIdx.x * 4u * PTS < n) {  // uniform per block
#pragma unroll
    for (int p = 0; p < PTS / 64; ++p) {
        const uint32_t j = p * 64 + lane;
        const uint32_t i = (base + j) * stride;
        bool ok = base + j < n;
        if (ok && labels) ok = labels[i] == want;
        float4 v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), 0.f);
        if (ok) v = pts[i];  // masked / out-of-range points are NaN: never an inlier
        float *gp = reinterpret_cast<float *>(&lp[wave][j >> 2][0]);
        gp[(j & 3)] = v.x; gp[4 + (j & 3)] = v.y; gp[8 + (j & 3)] = v.z;
    }
    float h0 = __builtin_nanf(""), h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, lo2 = 0, hi2 = 0;
    if (slot < K) {
        const uint32_t h = sel[slot];
        const float4 a = *reinterpret_cast<const float4 *>(hyp8 + 8 * (size_t)h);
        h0 = a.x; h1 = a.y; h2 = a.z; h3 = a.w;
        if (MODEL == 1) {
            const float4 b = *reinterpret_cast<const float4 *>(hyp8 + 8 * (size_t)h + 4);
            const float2 bd = band[h];
            h4 = b.x; h5 = b.y; lo2 = bd.x; hi2 = bd.y;
        }
    }
    wave_lds_fence();  // every wave reads only the points it staged itself
    float c = 0.f;
    // (every group of the wave's 64 is staged, points past the end as NaN, so a part may read its groups blind)
    const uint32_t gpp = (uint32_t)(PTS / 4) / parts;   // groups of four points per part
    const uint32_t g0 = split ? (lane / P) * gpp : 0u, g1 = split ? g0 + gpp : groups;
    for (uint32_t gi = g0; gi < g1; ++gi) {
        const float4 X = lp[wave][gi][0], Y = lp[wave][gi][1], Z = lp[wave][gi][2];  // broadcast reads: 4 points
        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODEL == 0) c += plane_inlier(xs[q], ys[q], zs[q], h0, h1, h2, h3, tau) ? 1.0f : 0.0f;
            else c += cyl_inlier(xs[q], ys[q], zs[q], h0, h1, h2, h3, h4, h5, lo2, hi2) ? 1.0f : 0.0f;
        }
    }
    red[wave][lane] = c;
    __syncthreads();
    if (wave == 0 && slot < K && (!split || lane < P)) {
        float t = 0.f;  // exact: integers <= 1024
        for (uint32_t part = 0; part < (split ? parts : 1u); ++part)
            t += red[0][lane + P * part] + red[1][lane + P * part] + red[2][lane + P * part] + red[3][lane + P * part];
        if (t > 0.f) atomicAdd(&counts_k[slot], (int32_t)t);
    }
    }
    if (nx.sel) select_by_last_block(counts_k, nx, sel_keys);
}


// two HYPOTHESES per vector instruction (v_pk_add / v_pk_mul / v_pk_fma_f32; a hypothesis pair's parameters sit in one
// scalar register pair, the point's coordinate is duplicated into a vector pair): every component rounds exactly like
// the scalar operation of plane_inlier / cyl_inlier, so the decisions are theirs bit for bit
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_sub_rn(v2f a, v2f b) { v2f r; r.x = __fsub_rn(a.x, b.x); r.y = __fsub_rn(a.y, b.y); return r; }
__device__ __forceinline__ v2f pk_mul_rn(v2f a, v2f b) { v2f r; r.x = __fmul_rn(a.x, b.x); r.y = __fmul_rn(a.y, b.y); return r; }
__device__ __forceinline__ v2f pk_fma_rn(v2f a, v2f b, v2f c) { v2f r; r.x = __fmaf_rn(a.x, b.x, c.x); r.y = __fmaf_rn(a.y, b.y, c.y); return r; }

// ---- the LAST stage: K <= 8 hypotheses on EVERY point, as a stream -------------------------------------------------------
// lane <-> point (one coalesced 16-byte load per lane, kStPer points per lane in flight), the K hypotheses wave-uniform
// in scalar registers, per hypothesis one compare chain, one ballot and one popcount into a scalar counter; a block adds
// its K counters with one integer atomic each.  HBM-bound for the plane (16 B per point), vector-bound for the cylinder
// (~17 instructions per hypothesis and point).  The stage can also leave, per point, the K-bit mask of the hypotheses it
// is an inlier of (1 byte), for a label pass that reads masks and touches the rows of inliers only: built and measured on
// one box (tools/ab_times.sh) -- with masks the cylinder label of the 10 M-point frame takes 62 us instead of 71.5 (nearly
// every point of a tunnel is an inlier: the rows are read anyway, the distance arithmetic is what is saved), the plane's
// 40.6 instead of 38.0 (every row load now waits for its mask), the 1 M-point frame's 21.4 instead of 20.6 -- a wash
// against a 2.8 ms frame, so the hand-off is off (GM_LABEL_MASKS=1 switches it on for A/B).
constexpr int kStThreads = 256, kStPer = 4, kStK = 8;
// A block ends with one integer atomic per hypothesis; on ONE word per hypothesis they are served one after the other
// (~12 ns each): 814 blocks made the 1 M-point launch 16.7 us, and a grid capped at 256 blocks left too few loads in
// flight on the 10 M-point frame (155 us for 133 MB).  The counters are kept in kStReplicas copies, 128 bytes apart --
// block b adds to copy b % kStReplicas -- and the label pass, the only reader, sums the copies.
constexpr int kStReplicas = 32, kStReplicaStride = 32;   // words (kStReplicas * kStK <= 256: the label pass reads them with one load per thread)
template <int MODEL, bool MASKS>
__global__ __launch_bounds__(kStThreads) void k_score_stream(const float4 *__restrict__ pts, const uint8_t *__restrict__ labels,
                                                             uint32_t want, const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                             const float *__restrict__ hyp8, const float2 *__restrict__ band,
                                                             const uint32_t *__restrict__ sel, uint32_t K, float tau,
                                                             int32_t *__restrict__ counts_k, uint8_t *__restrict__ masks)
{
    __shared__ uint32_t red[kStThreads / kWave][kStK];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    // the hypotheses, in pairs (uniform addresses: scalar loads; a pair's values share a scalar register pair); slots past K
    // score nothing
    static_assert(kStK % 2 == 0, "hypotheses are scored in pairs");
    v2f hp[kStK / 2][8];
#pragma unroll
    for (int k = 0; k < kStK; ++k) {
        float hv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) hv[q] = 0.f;
        hv[0] = __builtin_nanf("");
        if ((uint32_t)k < K) {
            const uint32_t hi = sel[k];
            const float *hy = hyp8 + 8 * (size_t)hi;
            hv[0] = hy[0]; hv[1] = hy[1]; hv[2] = hy[2]; hv[3] = hy[3];
            if (MODEL == 1) { const float2 bd = band[hi]; hv[4] = hy[4]; hv[5] = hy[5]; hv[6] = bd.x; hv[7] = bd.y; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { if (k & 1) hp[k >> 1][q].y = hv[q]; else hp[k >> 1][q].x = hv[q]; }
    }
    uint32_t cnt[kStK];
#pragma unroll
    for (int k = 0; k < kStK; ++k) cnt[k] = 0u;
    const uint32_t span = (uint32_t)(kStThreads * kStPer);
    // (loading the block's next span before scoring the current one was measured on the 10 M-point frame: no gain)
    for (uint32_t base = blockIdx.x * span; base < n; base += gridDim.x * span) {   // uniform per block
        float4 p[kStPer];
        bool ok[kStPer];
#pragma unroll
        for (int u = 0; u < kStPer; ++u) {
            const uint32_t i = base + (uint32_t)u * kStThreads + threadIdx.x;
            ok[u] = i < n;
            if (ok[u] && labels) ok[u] = labels[i] == want;
            p[u] = pts[i < n ? i : n - 1u];
        }
        uint32_t m[kStPer];
#pragma unroll
        for (int u = 0; u < kStPer; ++u) {
            m[u] = 0u;
            const v2f X = {p[u].x, p[u].x}, Y = {p[u].y, p[u].y}, Z = {p[u].z, p[u].z};
#pragma unroll
            for (int kp = 0; kp < kStK / 2; ++kp) {
                bool in0, in1;
                if (MODEL == 0) {   // plane_inlier: |fma(a, x, fma(b, y, fma(c, z, d)))| < tau
                    const v2f dist = pk_fma_rn(hp[kp][0], X, pk_fma_rn(hp[kp][1], Y, pk_fma_rn(hp[kp][2], Z, hp[kp][3])));
                    in0 = fabsf(dist.x) < tau; in1 = fabsf(dist.y) < tau;
                } else {            // cyl_inlier: lo2 < |v|^2 - (v.d)^2 < hi2, v = p - point on axis
                    const v2f vx = pk_sub_rn(X, hp[kp][0]), vy = pk_sub_rn(Y, hp[kp][1]), vz = pk_sub_rn(Z, hp[kp][2]);
                    const v2f t = pk_fma_rn(vx, hp[kp][3], pk_fma_rn(vy, hp[kp][4], pk_mul_rn(vz, hp[kp][5])));
                    const v2f vv = pk_fma_rn(vx, vx, pk_fma_rn(vy, vy, pk_mul_rn(vz, vz)));
                    const v2f q = pk_fma_rn(-t, t, vv);
                    in0 = (q.x > hp[kp][6].x) & (q.x < hp[kp][7].x); in1 = (q.y > hp[kp][6].y) & (q.y < hp[kp][7].y);
                }
                in0 = in0 & ok[u]; in1 = in1 & ok[u];
                cnt[2 * kp] += (uint32_t)__popcll(__ballot(in0));       // wave-uniform
                cnt[2 * kp + 1] += (uint32_t)__popcll(__ballot(in1));
                if (MASKS) m[u] |= (in0 ? (1u << (2 * kp)) : 0u) | (in1 ? (2u << (2 * kp)) : 0u);
            }
        }
        if (MASKS) {
#pragma unroll
            for (int u = 0; u < kStPer; ++u) {
                const uint32_t i = base + (uint32_t)u * kStThreads + threadIdx.x;
                if (i < n) masks[i] = (uint8_t)m[u];
            }
        }
    }
    const int w = threadIdx.x / kWave;
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < kStK; ++k) red[w][k] = cnt[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        uint32_t t = 0;
#pragma unroll
        for (int j = 0; j < kStThreads / kWave; ++j) t += red[j][threadIdx.x];
        if (t) atomicAdd(&counts_k[(blockIdx.x % (uint32_t)kStReplicas) * (uint32_t)kStReplicaStride + threadIdx.x], (int32_t)t);
    }
}

// init != 0: this is the first model of the frame -- every point is eligible and labels are WRITTEN for all
// points (no memset of the label array is needed); otherwise only points with labels == want are touched.
// counts_k != nullptr: the winner is taken from the K re-scored hypotheses (largest count, lowest hypothesis index
// on ties) by every block for itself, and block 0 publishes it in best[0..1].
// masks != nullptr (with counts_k): the last scoring stage left every point's K-bit inlier mask (k_score_stream): the pass
// reads the winner's bit instead of the row, and loads rows (and normals) of inliers only.
// MOM (frame pipeline): the pass also sums the moments of the segment it labels -- count, sum p, sum pp^T and, for the
// cylinder, sum nn^T, in fp64 -- into mom_partial[MODEL][block][16]: the points are in registers anyway, so the
// separate pass over labels + points + normals (33 B per point for 13 algorithmic) is gone; normals are only loaded
// for inliers.
template <int MODEL, int MOM>
__global__ __launch_bounds__(256) void k_label(const float4 *__restrict__ pts, uint8_t *__restrict__ labels,
                                               uint32_t want, uint32_t label, const uint32_t *__restrict__ n_ptr,
                                               uint32_t n_host, const float *__restrict__ hyp8,
                                               const float2 *__restrict__ band, uint32_t *__restrict__ best,
                                               float tau, int init, const int32_t *__restrict__ counts_k,
                                               const uint32_t *__restrict__ sel, uint32_t K,
                                               const float4 *__restrict__ nrm, double *__restrict__ mom_partial,
                                               const uint8_t *__restrict__ masks, uint32_t replicas)
{
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    uint32_t h, wbit = 0;
    if (counts_k) {
        __shared__ uint32_t win[3];
        __shared__ uint32_t csum[kStK];
        if (replicas > 1u) {   // (uniform) the streaming last stage's counters: kStReplicas copies, one load per thread, summed in LDS
            if (threadIdx.x < (uint32_t)kStK) csum[threadIdx.x] = 0u;
            __syncthreads();
            const uint32_t r = threadIdx.x / (uint32_t)kStK, k = threadIdx.x % (uint32_t)kStK;
            if (r < replicas && k < K) {
                const uint32_t v = (uint32_t)counts_k[r * (uint32_t)kStReplicaStride + k];
                if (v) atomicAdd(&csum[k], v);
            }
            __syncthreads();
        }
        if (threadIdx.x < kWave) {
            uint32_t c = 0, hi = 0xFFFFFFFFu, sl = threadIdx.x;
            if (threadIdx.x < K) {
                c = replicas > 1u ? csum[threadIdx.x] : (uint32_t)counts_k[threadIdx.x];
                hi = sel[threadIdx.x];
            }
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t oc = __shfl_xor(c, o, kWave), oh = __shfl_xor(hi, o, kWave), os = __shfl_xor(sl, o, kWave);
                if (oh != 0xFFFFFFFFu && (hi == 0xFFFFFFFFu || oc > c || (oc == c && oh < hi))) { c = oc; hi = oh; sl = os; }
            }
            if (threadIdx.x == 0) { win[0] = hi; win[1] = c; win[2] = sl; }
        }
        __syncthreads();
        h = win[0];
        wbit = win[2] & 7u;
        if (blockIdx.x == 0 && threadIdx.x == 0) { best[0] = win[0]; best[1] = win[1]; }
    } else {
        h = best[0];
    }
    constexpr int NM = MODEL == 0 ? 10 : 16;
    if (h == 0xFFFFFFFFu) {
        if (init)
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) labels[i] = 0;
        if (MOM && threadIdx.x < 16) mom_partial[((size_t)MODEL * kScatterBlocks + blockIdx.x) * 16 + threadIdx.x] = 0.0;
        return;
    }
    const float *hy = hyp8 + 8 * (size_t)h;
    const float a = hy[0], b = hy[1], c = hy[2], d = hy[3], e = hy[4], f = hy[5];
    float lo2 = 0, hi2 = 0;
    if (MODEL == 1) { const float2 bd = band[h]; lo2 = bd.x; hi2 = bd.y; }
    double m[NM];
#pragma unroll
    for (int k = 0; k < NM; ++k) m[k] = 0.0;
    const bool by_mask = masks != nullptr && counts_k != nullptr;   // uniform
    auto add_moments = [&](const float4 p, uint32_t i) {
        const double x = p.x, y = p.y, z = p.z;
        m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
        m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
        if (MODEL == 1) {
            const float4 q = nrm[i];
            const double u = q.x, v = q.y, w = q.z;
            m[10] += u * u; m[11] += u * v; m[12] += u * w; m[13] += v * v; m[14] += v * w; m[15] += w * w;
        }
    };
    // (four points per thread and trip -- all masks, then all rows -- was measured on the 10 M-point frame: 56.7 -> 67.6 us
    // for the cylinder label, 35.5 -> 37.1 for the plane's; one point per trip it stays)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        bool in;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (by_mask) {
            // (the mask was taken over the points eligible at scoring time -- labels == want -- and no label has changed since)
            in = ((masks[i] >> wbit) & 1u) != 0u;
            if (MOM && in) p = pts[i];
        } else {
            const bool take = init || labels[i] == want;
            p = pts[i];
            in = take && (MODEL == 0 ? plane_inlier(p.x, p.y, p.z, a, b, c, d, tau)
                                     : cyl_inlier(p.x, p.y, p.z, a, b, c, d, e, f, lo2, hi2));
        }
        if (init) labels[i] = in ? (uint8_t)label : (uint8_t)0;
        else if (in) labels[i] = (uint8_t)label;
        if (MOM && in) add_moments(p, i);
    }
    if (MOM) {
        __shared__ double red[256 / kWave][16];
        const int w = threadIdx.x / kWave;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            const double r = wave_sum(m[k]);
            if (lane_id() == 0) red[w][k] = r;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            double r = 0;
            if ((int)threadIdx.x < NM)
#pragma unroll
                for (int j = 0; j < 256 / kWave; ++j) r += red[j][threadIdx.x];
            mom_partial[((size_t)MODEL * kScatterBlocks + blockIdx.x) * 16 + threadIdx.x] = r;
        }
    }
}

// mom16 = count, sum p (3), sum pp^T (6), sum nn^T (6) over points with labels == label (fp64)
__global__ __launch_bounds__(256) void k_segment_moments(const float4 *__restrict__ pts,
                                                         const float4 *__restrict__ nrm,
                                                         const uint8_t *__restrict__ labels, uint32_t label,
                                                         const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                         double *__restrict__ partial /* [gridDim.x][16] */)
{
    __shared__ double red[256 / kWave][16];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    double m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (labels && labels[i] != label) continue;
        const float4 p = pts[i];
        const double x = p.x, y = p.y, z = p.z;
        m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
        m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
        if (nrm) {
            const float4 q = nrm[i];
            const double a = q.x, b = q.y, c = q.z;
            m[10] += a * a; m[11] += a * b; m[12] += a * c; m[13] += b * b; m[14] += b * c; m[15] += c * c;
        }
    }
    const int w = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double r = wave_sum(m[k]);
        if (lane_id() == 0) red[w][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double r = 0;
#pragma unroll
        for (int j = 0; j < 256 / kWave; ++j) r += red[j][threadIdx.x];
        partial[(size_t)blockIdx.x * 16 + threadIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void k_moments_finalize(const double *__restrict__ partial, uint32_t nblocks,
                                                          double *__restrict__ mom16)
{
    // fixed-order: thread t sums rows t, t+16, ... of column (t & 15); then a 16-way tree per column
    __shared__ double red[16][16];
    const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
    double r = 0;
    for (uint32_t b = part; b < nblocks; b += 16) r += partial[(size_t)b * 16 + col];
    red[part][col] = r;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        mom16[threadIdx.x] = t;
    }
}

// refits + copy of the winning hypotheses into the frame record
// mom_rows > 0: first reduce the label passes' partial rows (fixed order) into mom_plane / mom_cyl, then finalize.
__global__ __launch_bounds__(256) void k_ext_finalize(const float *__restrict__ hyp_plane, const uint32_t *__restrict__ best_plane,
                               const float *__restrict__ hyp_cyl, const uint32_t *__restrict__ best_cyl,
                               double *__restrict__ mom_plane, double *__restrict__ mom_cyl,
                               FrameExt *__restrict__ ext, const double *__restrict__ partial, uint32_t mom_rows,
                               const double *__restrict__ scatter_partials, uint32_t scatter_rows, uint32_t row_tile,
                               const DevCounters *__restrict__ ctr, const VoxelParams *__restrict__ voxp,
                               FrameOut *__restrict__ frame_out)
{
    // the frame's own closing step rides along (frame pipeline only) as a second block: one launch instead of two
    // single-block ones, and the two latency chains (reduce + 3x3 solves each) run side by side
    if (blockIdx.x == 1) {
        __shared__ double fred[256 * 6];
        if (frame_out) frame_finalize_block(scatter_partials, scatter_rows, ctr, voxp, frame_out, fred, row_tile);
        return;
    }
    if (mom_rows) {   // fixed-order reduction of the label passes' partial rows: partial = [model][kScatterBlocks][16]
        __shared__ double red[8][32];
        const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
        const double *src = partial + (size_t)(col >> 4) * kScatterBlocks * 16 + (col & 15);
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0;  // four independent chains: the loads of a trip are all in flight together
        uint32_t b = part;
        for (; b + 24 < mom_rows; b += 32) {
            r0 += src[(size_t)b * 16];
            r1 += src[(size_t)(b + 8) * 16];
            r2 += src[(size_t)(b + 16) * 16];
            r3 += src[(size_t)(b + 24) * 16];
        }
        for (; b < mom_rows; b += 8) r0 += src[(size_t)b * 16];
        red[part][col] = (r0 + r1) + (r2 + r3);
        __syncthreads();
        if (threadIdx.x < 32) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
            if (threadIdx.x < 16) mom_plane[threadIdx.x] = t; else mom_cyl[threadIdx.x - 16] = t;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    FrameExt e;
    for (int k = 0; k < 4; ++k) { e.plane[k] = __builtin_nanf(""); e.plane_refit[k] = __builtin_nan(""); }
    for (int k = 0; k < 7; ++k) e.cylinder[k] = __builtin_nanf("");
    for (int k = 0; k < 3; ++k) e.cyl_axis_refit[k] = __builtin_nan("");
    e.plane_inliers = 0; e.cylinder_inliers = 0;
    if (best_plane && best_plane[0] != 0xFFFFFFFFu) {
        const float *hy = hyp_plane + 8 * (size_t)best_plane[0];
        for (int k = 0; k < 4; ++k) e.plane[k] = hy[k];
        e.plane_inliers = best_plane[1];
        const double *m = mom_plane;
        const double n = m[0];
        if (n >= 3.0) {
            const double cx = m[1] / n, cy = m[2] / n, cz = m[3] / n;
            const double C[6] = {m[4] / n - cx * cx, m[5] / n - cx * cy, m[6] / n - cx * cz,
                                 m[7] / n - cy * cy, m[8] / n - cy * cz, m[9] / n - cz * cz};
            double w[3], V[9];
            jacobi_eig3(C, w, V);
            e.plane_refit[0] = V[0]; e.plane_refit[1] = V[1]; e.plane_refit[2] = V[2];
            e.plane_refit[3] = -(V[0] * cx + V[1] * cy + V[2] * cz);
        }
    }
    if (best_cyl && best_cyl[0] != 0xFFFFFFFFu) {
        const float *hy = hyp_cyl + 8 * (size_t)best_cyl[0];
        for (int k = 0; k < 7; ++k) e.cylinder[k] = hy[k];
        e.cylinder_inliers = best_cyl[1];
        const double *m = mom_cyl;
        const double S[6] = {m[10], m[11], m[12], m[13], m[14], m[15]};
        double w[3], V[9];
        jacobi_eig3(S, w, V);
        e.cyl_axis_refit[0] = V[0]; e.cyl_axis_refit[1] = V[1]; e.cyl_axis_refit[2] = V[2];
    }
    *ext = e;
}

// ---- launchers -------------------------------------------------------------------

uint32_t score_blocks(uint32_t n_cap) { return (n_cap + kScTile - 1) / kScTile; }

void launch_plane_hypotheses(const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr,
                             uint32_t n_host, uint64_t seed, uint32_t H, float *hyp8, int32_t *zero_counts,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_plane_hypotheses, dim3((H + 255) / 256), dim3(256), 0, s, pts, labels, want, n_ptr, n_host,
                       seed, H, hyp8, zero_counts);
}

void launch_cylinder_hypotheses(const float4 *pts, const float4 *nrm, const uint8_t *labels, uint32_t want,
                                const uint32_t *n_ptr, uint32_t n_host, uint64_t seed, uint32_t H, float *hyp8,
                                int32_t *zero_counts, float2 *band, double tau, hipStream_t s)
{
    hipLaunchKernelGGL(k_cylinder_hypotheses, dim3((H + 255) / 256), dim3(256), 0, s, pts, nrm, labels, want, n_ptr,
                       n_host, seed, H, hyp8, zero_counts, band, tau);
}

void launch_score(int model, const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr,
                  uint32_t n_cap, const float *hyp8, float2 *band, uint32_t H, double tau, uint32_t *partial,
                  int32_t *counts, uint32_t *best, hipStream_t s)
{
    (void)partial;
    const uint32_t nb = score_blocks(n_cap) ? score_blocks(n_cap) : 1;
    const dim3 grid(nb, (H + kScHC - 1) / kScHC);
    hipMemsetAsync(counts, 0, sizeof(int32_t) * H, s);
    if (model == 0) {
        hipLaunchKernelGGL(k_score<0>, grid, dim3(kScThreads), 0, s, pts, labels, want, n_ptr, n_cap, hyp8,
                           (const float2 *)band, H, (float)tau, 1u, (uint32_t)kScTile, counts, SelectNext{});
    } else {
        hipLaunchKernelGGL(k_cyl_bands, dim3((H + 255) / 256), dim3(256), 0, s, hyp8, H, tau, band);
        hipLaunchKernelGGL(k_score<1>, grid, dim3(kScThreads), 0, s, pts, labels, want, n_ptr, n_cap, hyp8,
                           (const float2 *)band, H, (float)tau, 1u, (uint32_t)kScTile, counts, SelectNext{});
    }
    hipLaunchKernelGGL(k_best_hypothesis, dim3(1), dim3(1024), 0, s, (const int32_t *)counts, H, best);
}

// In-frame RANSAC: preemptive scoring in three stages of equal cost (see kPre* above).  `scratch` holds
// kPreScratchWords words.  Returns false when H was small enough for the exhaustive scorer (the winner is then in
// `best`); otherwise the winner is still to be taken from (*sel_out, *cnt_out, *k_out) -- the label kernel does that.
// prepared: the hypothesis kernel already cleared `counts` and wrote the cylinder bands.
// next.sel != nullptr: the launch's last block also selects the next stage's candidates (select_by_last_block)
template <int MODEL>
static void score_stage_all(const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr, uint32_t n_cap,
                            const float *hyp8, const float2 *band, uint32_t H, double tau, uint32_t stride,
                            int32_t *counts, const SelectNext &next, hipStream_t s)
{
    const uint32_t n_sub = (n_cap + stride - 1) / stride;
    // points per block: as few as keep ~2 blocks per CU busy (at least 32: a block's fixed costs), at most kScTile
    const uint32_t hy = (H + kScHC - 1) / kScHC;
    uint32_t tile = (uint32_t)kScTile;
    while (tile > 32u && (uint64_t)((n_sub + tile - 1) / tile) * hy < 512u) tile >>= 1;
    static const char *te = getenv("GM_RANSAC_TILE");   // experiments
    if (te) {   // (a multiple of 4 in [32, kScTile]: anything else is not a tile the kernel can walk)
        const int tv = atoi(te);
        if (tv >= 32 && tv <= kScTile && (tv & 3) == 0) tile = (uint32_t)tv;
    }
    const uint32_t nb = (n_sub + tile - 1) / tile ? (n_sub + tile - 1) / tile : 1;
    hipLaunchKernelGGL(k_score<MODEL>, dim3(nb, hy), dim3(kScThreads),
                       next.sel ? select_lds_bytes(next.M) : 0u, s, pts, labels, want, n_ptr, n_cap, hyp8, band, H,
                       (float)tau, stride, tile, counts, next);
}
template <int MODEL>
static void score_stage_sel(const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr, uint32_t n_cap,
                            const float *hyp8, const float2 *band, const uint32_t *sel, uint32_t K, double tau,
                            uint32_t stride, int32_t *counts_k, const SelectNext &next, hipStream_t s)
{
    const uint32_t n_sub = (n_cap + stride - 1) / stride;
    const uint32_t ky = (K + 63) / 64;
    // (K <= 32 splits a wave's points over lane parts and wants the full 256; otherwise 64 points per wave when the stage
    // would not fill the chip with 256)
    const bool small = K > 32u && (uint64_t)((n_sub + 1023) / 1024) * ky < 512u;
    const uint32_t per_block = small ? 256u : 1024u;
    const uint32_t nb = (n_sub + per_block - 1) / per_block ? (n_sub + per_block - 1) / per_block : 1;
    if (small)
        hipLaunchKernelGGL((k_score_sel<MODEL, 64>), dim3(nb, ky), dim3(256), next.sel ? select_lds_bytes(next.M) : 0u, s,
                           pts, labels, want, n_ptr, n_cap, hyp8, band, sel, K, (float)tau, stride, counts_k, next);
    else
        hipLaunchKernelGGL((k_score_sel<MODEL, 256>), dim3(nb, ky), dim3(256), next.sel ? select_lds_bytes(next.M) : 0u, s,
                           pts, labels, want, n_ptr, n_cap, hyp8, band, sel, K, (float)tau, stride, counts_k, next);
}
static void select_topk(const int32_t *counts, const uint32_t *ids, uint32_t M, uint32_t K, uint32_t *sel,
                        int32_t *counts_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_select_topk, dim3((M + 15) / 16), dim3(256), 2 * sizeof(int32_t) * M, s, counts, ids, M, K, sel,
                       counts_out);
}

bool launch_score_preemptive(int model, const float4 *pts, const uint8_t *labels, uint32_t want,
                             const uint32_t *n_ptr, uint32_t n_cap, const float *hyp8, float2 *band, uint32_t H,
                             double tau, uint32_t *scratch, int32_t *counts, uint32_t *best, bool prepared,
                             const uint32_t **sel_out, const int32_t **cnt_out, uint32_t *k_out, hipStream_t s, uint8_t *masks,
                             bool *masks_written, bool *replicated_counts)
{
    if (masks_written) *masks_written = false;
    if (replicated_counts) *replicated_counts = false;
    if (H <= (uint32_t)kPre2Keep) {  // nothing to pre-select
        launch_score(model, pts, labels, want, n_ptr, n_cap, hyp8, band, H, tau, nullptr, counts, best, s);
        return false;
    }
    uint32_t *selA = scratch + kPreSelA, *selB = scratch + kPreSelB;
    int32_t *cntA = (int32_t *)(scratch + kPreCntA), *cntB = (int32_t *)(scratch + kPreCntB);
    if (!prepared) {
        hipMemsetAsync(counts, 0, sizeof(int32_t) * H, s);
        if (model == 1) hipLaunchKernelGGL(k_cyl_bands, dim3((H + 255) / 256), dim3(256), 0, s, hyp8, H, tau, band);
    }
    const float2 *cb = band;
    const uint32_t K2 = kPre2Keep;
    uint32_t *done = scratch + kPreDone;
    static const char *fs = getenv("GM_RANSAC_FINAL");   // "sel": the last stage in the lane <-> hypothesis shape (A/B timing)
    const bool stream = masks && K2 <= (uint32_t)kStK && !(fs && fs[0] == 's');
    static const char *lm = getenv("GM_LABEL_MASKS");   // "1": the streaming stage leaves inlier masks for the label pass (A/B)
    const bool write_masks = stream && lm && lm[0] == '1';
    if (stream) cntB = (int32_t *)(scratch + kPreCntStream);
    const uint32_t zeroB = stream ? (uint32_t)(kStReplicas * kStReplicaStride) : 0u;
    const SelectNext none{};
    // (a selection over more than kSelectFoldMax candidates keeps its own launch: the LDS sort would not fit)
    if (H > (uint32_t)kPre1Keep) {
        const uint32_t K1 = kPre1Keep;
        static const char *own = getenv("GM_RANSAC_SELECT");   // "kernel": selections keep their own launches (A/B timing)
        const bool fold = H <= kSelectFoldMax && !(own && own[0] == 'k');
        const SelectNext n1{done, nullptr, H, K1, selA, cntA, 0u}, n2{done, selA, K1, K2, selB, cntB, zeroB};
        if (model == 0) score_stage_all<0>(pts, labels, want, n_ptr, n_cap, hyp8, cb, H, tau, kPre1Stride, counts, fold ? n1 : none, s);
        else score_stage_all<1>(pts, labels, want, n_ptr, n_cap, hyp8, cb, H, tau, kPre1Stride, counts, fold ? n1 : none, s);
        if (!fold) select_topk(counts, nullptr, H, K1, selA, cntA, s);
        if (model == 0) score_stage_sel<0>(pts, labels, want, n_ptr, n_cap, hyp8, cb, selA, K1, tau, kPre2Stride, cntA, fold ? n2 : none, s);
        else score_stage_sel<1>(pts, labels, want, n_ptr, n_cap, hyp8, cb, selA, K1, tau, kPre2Stride, cntA, fold ? n2 : none, s);
        if (!fold) {
            select_topk(cntA, selA, K1, K2, selB, cntB, s);
            if (stream) (void)hipMemsetAsync(cntB, 0, sizeof(int32_t) * zeroB, s);
        }
    } else {
        const SelectNext n2{done, nullptr, H, K2, selB, cntB, zeroB};
        if (model == 0) score_stage_all<0>(pts, labels, want, n_ptr, n_cap, hyp8, cb, H, tau, kPre2Stride, counts, n2, s);
        else score_stage_all<1>(pts, labels, want, n_ptr, n_cap, hyp8, cb, H, tau, kPre2Stride, counts, n2, s);
    }
    if (stream) {
        // the last stage streams: lane <-> point, hypotheses in scalar registers (k_score_stream); also leaves the inlier masks
        uint32_t nb = (n_cap + kStThreads * kStPer - 1) / (kStThreads * kStPer);
        if (nb > 2048u) nb = 2048u;   // (8 waves per CU: 32 MB of loads in flight)
        if (nb == 0) nb = 1;
#define GM_STREAM(M, MK)                                                                                                   \
    hipLaunchKernelGGL((k_score_stream<M, MK>), dim3(nb), dim3(kStThreads), 0, s, pts, labels, want, n_ptr, n_cap, hyp8, cb, \
                       (const uint32_t *)selB, K2, (float)tau, cntB, MK ? masks : (uint8_t *)nullptr)
        if (model == 0) { if (write_masks) GM_STREAM(0, true); else GM_STREAM(0, false); }
        else { if (write_masks) GM_STREAM(1, true); else GM_STREAM(1, false); }
#undef GM_STREAM
        *sel_out = selB; *cnt_out = cntB; *k_out = K2;
        if (masks_written) *masks_written = write_masks;
        if (replicated_counts) *replicated_counts = true;
        return true;
    }
    if (model == 0) score_stage_sel<0>(pts, labels, want, n_ptr, n_cap, hyp8, cb, selB, K2, tau, 1u, cntB, none, s);
    else score_stage_sel<1>(pts, labels, want, n_ptr, n_cap, hyp8, cb, selB, K2, tau, 1u, cntB, none, s);
    *sel_out = selB; *cnt_out = cntB; *k_out = K2;
    return true;
}

uint32_t launch_label(int model, const float4 *pts, uint8_t *labels, uint32_t want, uint32_t label, const uint32_t *n_ptr,
                      uint32_t n_cap, const float *hyp8, const float2 *band, uint32_t *best, double tau, int init,
                      const uint32_t *sel, const int32_t *counts_k, uint32_t K, hipStream_t s, const float4 *nrm,
                      double *mom_partial, const uint8_t *masks, bool replicated_counts)
{
    // counts_k != nullptr: winner = best of the K (<= 64) finally re-scored hypotheses in (sel[K], counts_k[K])
    uint32_t nb = (n_cap + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (mom_partial) {   // one partial row per block: ~4096 points per block, at most kScatterBlocks rows
#ifndef GM_LABEL_PTS
#define GM_LABEL_PTS 4096
#endif
        nb = (n_cap + GM_LABEL_PTS - 1) / GM_LABEL_PTS;
        if (nb < 64u) nb = (n_cap + 255) / 256 < 64u ? (n_cap + 255) / 256 : 64u;
        if (nb > (uint32_t)kScatterBlocks) nb = kScatterBlocks;
    }
    if (nb == 0) nb = 1;
#define GM_LABEL(M, MO)                                                                                                 \
    hipLaunchKernelGGL((k_label<M, MO>), dim3(nb), dim3(256), 0, s, pts, labels, want, label, n_ptr, n_cap, hyp8, band, \
                       best, (float)tau, init, counts_k, sel, K, nrm, mom_partial, masks, replicated_counts ? (uint32_t)kStReplicas : 1u)
    if (model == 0) { if (mom_partial) GM_LABEL(0, 1); else GM_LABEL(0, 0); }
    else { if (mom_partial) GM_LABEL(1, 1); else GM_LABEL(1, 0); }
#undef GM_LABEL
    return nb;
}

void launch_segment_moments(const float4 *pts, const float4 *nrm, const uint8_t *labels, uint32_t label,
                            const uint32_t *n_ptr, uint32_t n_cap, double *partial, double *mom16, hipStream_t s)
{
    uint32_t nb = (n_cap + 255) / 256;
    if (nb > (uint32_t)kScatterBlocks) nb = kScatterBlocks;  // 4 blocks per CU: enough loads in flight to stream
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_segment_moments, dim3(nb), dim3(256), 0, s, pts, nrm, labels, label, n_ptr, n_cap, partial);
    hipLaunchKernelGGL(k_moments_finalize, dim3(1), dim3(256), 0, s, (const double *)partial, nb, mom16);
}

void launch_ext_finalize(const float *hyp_plane, const uint32_t *best_plane, const float *hyp_cyl,
                         const uint32_t *best_cyl, double *mom_plane, double *mom_cyl, FrameExt *ext,
                         const double *partial32, uint32_t mom_rows, hipStream_t s, const double *scatter_partials,
                         uint32_t scatter_rows, uint32_t row_tile, const DevCounters *ctr, const VoxelParams *voxp,
                         FrameOut *frame_out)
{
    hipLaunchKernelGGL(k_ext_finalize, dim3(frame_out ? 2 : 1), dim3(256), 0, s, hyp_plane, best_plane, hyp_cyl, best_cyl, mom_plane,
                       mom_cyl, ext, partial32, mom_rows, scatter_partials, scatter_rows, row_tile, ctr, voxp, frame_out);
}

}  // namespace gm
