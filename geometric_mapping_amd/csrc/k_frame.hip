// k_frame.hip -- NaN-normal compaction, the curvature-weighted normal scatter
// matrix and its 3x3 eigen-decomposition (the "centre axis" fit).
//
// Replaces
//   removeNaNNormalsFromPointCloud + ExtractIndices
//       (/root/reference src/tunnel_processing.cpp:74-85)
//   getLocalFrame (/root/reference src/tunnel_processing.cpp:92-148):
//       w_i = exp((c_i + .001/wf)^2) evaluated in double, stored to float (:106),
//       M = (W N)^T (W N) = sum_i w_i^2 n_i n_i^T (:119-124), SelfAdjointEigenSolver (:129).
// The reference materialises W as a dense n x n matrix (4 n^2 bytes); here M is
// six fp64 running sums per lane over one streaming read of 16 B per point --
// the HBM-streaming kernel of the path.  fp64 accumulation makes M independent
// of the reduction order to ~1e-16, so partials from any number of blocks (or
// GPUs) merge bit-stably.
#include "gm_compact.hpp"
#include "gm_internal.hpp"

namespace gm {

// ---- compaction of points with a finite normal ----------------------------------
// removeNaNNormalsFromPointCloud's predicate: all three components finite.  (k_normals stores no normal for a point the
// rank does not own, so slab ownership needs no test of its own.)  The normal rides to the emit step in registers.
struct ValidPred {
    const float4 *__restrict__ normals4;
    uint32_t *__restrict__ first_drop_enc;   // DevCounters::first_drop_enc: up to the first dropped point the valid cloud IS the cropped cloud
    typedef float4 Payload;
    __device__ __forceinline__ bool operator()(uint32_t i, Payload &p) const
    {
        p = normals4[i];
        const bool ok = finite3(p.x, p.y, p.z);
        if (!ok) {   // (rare on dense frames; a lane only adds what can still raise the maximum it last saw)
            const uint32_t enc = 0xFFFFFFFFu - i;
            if (*first_drop_enc < enc) atomicMax(first_drop_enc, enc);
        }
        return ok;
    }
};

// rows [begin, *end_ptr) of src -> dst, where dst is page-locked HOST memory mapped into the device's address space
// (/choppedCloud, gm_set_cloud_output): 16-byte stores, consecutive lanes on consecutive rows, over PCIe.  A few blocks are
// enough to fill the link and leave the chip to the frame's kernels.  begin_enc != nullptr: begin = 0xFFFFFFFF - *begin_enc
// (DevCounters::first_drop_enc; 0 = nothing to copy).
__global__ __launch_bounds__(256) void k_rows_to_host(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                      const uint32_t *__restrict__ begin_enc, const uint32_t *__restrict__ end_ptr)
{
    const uint32_t end = *end_ptr;
    uint32_t begin = 0;
    if (begin_enc) {
        const uint32_t e = *begin_enc;
        if (e == 0u) return;
        begin = 0xFFFFFFFFu - e;
    }
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
        dst[i] = src[i];
}

void launch_rows_to_host(const float4 *src, float4 *dst_mapped, const uint32_t *begin_enc, const uint32_t *end_ptr, hipStream_t s)
{
    hipLaunchKernelGGL(k_rows_to_host, dim3(64), dim3(256), 0, s, src, dst_mapped, begin_enc, end_ptr);
}

// one term of getLocalFrame's scatter matrix (src/tunnel_processing.cpp:100-124) added to m[6]
__device__ __forceinline__ void scatter_term(const float4 v /* nx,ny,nz,curvature */, double k_wf, double m[6])
{
    // :106: float + double -> double, pow(.,2), exp, store to float
    const double t = (double)v.w + k_wf;
    const float wgt = (float)exp(t * t);
    // :119 weights*normals: one fp32 product per component (zeros add exactly)
    const double a = (double)(wgt * v.x), b = (double)(wgt * v.y), c = (double)(wgt * v.z);
    // :124 exact fp64 products of those fp32 values
    m[0] += a * a; m[1] += a * b; m[2] += a * c; m[3] += b * b; m[4] += b * c; m[5] += c * c;
}

// block sum of m[6] in a fixed order -> row[0..6)
__device__ __forceinline__ void scatter_row_store(const double m[6], double *__restrict__ row)
{
    __shared__ double red[16][6];   // (blocks of up to 1024 threads)
    const int w = threadIdx.x / kWave, nw = blockDim.x / kWave;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double r = wave_sum(m[k]);
        if (lane_id() == 0) red[w][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = 0;
#pragma unroll
        for (int j = 0; j < nw; ++j) r += red[j][threadIdx.x];
        row[threadIdx.x] = r;
    }
}

// The compaction's emit step also accumulates the survivors' scatter terms -- the normals are in registers anyway -- and
// leaves one partial row per tile (index = tile, i.e. position order: the rows do not depend on which block ran which
// tile), so getLocalFrame needs no pass of its own over the compacted normals.
struct ValidEmit {
    static constexpr bool kHasFinish = true, kHasPrepare = false;
    const float4 *__restrict__ crop4;
    float4 *__restrict__ valid4;
    float4 *__restrict__ vnorm4;
    double k_wf;                     // .001 / weightingFactor
    double *__restrict__ partials;   // [tiles][6]
    double m[6];
    __device__ __forceinline__ void operator()(uint32_t src, uint32_t dst, const float4 &v)
    {
        valid4[dst] = crop4[src];
        vnorm4[dst] = v;
        scatter_term(v, k_wf, m);
    }
    __device__ __forceinline__ void finish(uint32_t tile)
    {
        scatter_row_store(m, partials + (size_t)tile * 6);
#pragma unroll
        for (int k = 0; k < 6; ++k) m[k] = 0;   // the block's next tile starts afresh
    }
};

// min/max of x,y,z over pts[0..n) (getMinMax3D of pcl::VoxelGrid) -> ordered uints
__global__ __launch_bounds__(256) void k_minmax(const float4 *__restrict__ pts, const uint32_t *__restrict__ n_ptr,
                                                uint32_t n_host, DevCounters *__restrict__ ctr)
{
    __shared__ float red[6][256 / kWave];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    const int w = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane_id() == 0) { red[k][w] = a; red[3 + k][w] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[k][0];
        for (int j = 1; j < 256 / kWave; ++j) v = (k < 3) ? fminf(v, red[k][j]) : fmaxf(v, red[k][j]);
        if (blockIdx.x * blockDim.x < n) {
            if (k < 3) atomicMin(&ctr->mm[k], float_to_ordered(v));
            else atomicMax(&ctr->mm[k], float_to_ordered(v));
        }
    }
}

__global__ void k_init_minmax(DevCounters *ctr)
{
    if (threadIdx.x < 3) ctr->mm[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) ctr->mm[threadIdx.x] = 0u;
}

void launch_minmax(const float4 *pts, const uint32_t *n_ptr, uint32_t n_cap, DevCounters *ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_minmax, dim3(1), dim3(64), 0, s, ctr);
    if (n_cap == 0) return;
    uint32_t nb = (n_cap + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, s, pts, n_ptr, n_cap, ctr);
}

uint32_t launch_compact_valid(Slot &sl, uint32_t n_cap, double wf, hipStream_t s, uint32_t *row_tile)
{
    // Tile: 4096 points (512 threads x 8); 8192 (1024 x 8) for frames of several million points, as for the crop
    // (k_crop.hip: fewer tickets and look-backs per byte).  GM_VALID_TILE=<threads>x<items>: experiments.
    static const char *e = getenv("GM_VALID_TILE");
    int th = 512, it = 8;
    if (e) sscanf(e, "%dx%d", &th, &it);
    else if (n_cap > 2000000u) { th = 1024; it = 8; }   // (10 M points: 122 -> 109 us; 3 M: 44 -> 38)
    const bool big = th == 1024 && it == 8;
    const uint32_t tile = big ? 8192u : (uint32_t)kCpTile;
    if (row_tile) *row_tile = tile;
    const uint32_t nb = (n_cap + tile - 1) / tile;
    if (nb == 0) return 0;
    ValidPred pred{sl.normals4, &sl.ctr->first_drop_enc};
    ValidEmit emit{sl.crop4, sl.valid4, sl.vnorm4, .001 / wf, sl.tile_partials, {0, 0, 0, 0, 0, 0}};
    if (big)
        hipLaunchKernelGGL((k_compact<ValidPred, ValidEmit, 1024, 8>), dim3(nb), dim3(1024), 0, s, pred, emit,
                           (const uint32_t *)&sl.ctr->n_cropped, 0u, next_scan(sl), &sl.ctr->n_valid, &sl.ctr->vox_n);
    else
        hipLaunchKernelGGL((k_compact<ValidPred, ValidEmit>), dim3(nb), dim3(kCpThreads), 0, s, pred, emit,
                           (const uint32_t *)&sl.ctr->n_cropped, 0u, next_scan(sl), &sl.ctr->n_valid, &sl.ctr->vox_n);
    return nb;  // partial rows: one per tile that held input (the finalizer derives how many from n_cropped)
}

// ---- scatter matrix: streaming pass ---------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_partials(const float4 *__restrict__ vnorm4,
                                                          const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                          double k_wf /* .001 / weightingFactor */,
                                                          double *__restrict__ partials)
{
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        scatter_term(vnorm4[i], k_wf, m);  // one coalesced 16 B load per lane: nx,ny,nz,curvature
    scatter_row_store(m, partials + (size_t)blockIdx.x * 6);
}

// ---- fixed-order reduction of the block partials + 3x3 eigen (one block) ---------
__global__ __launch_bounds__(256) void k_frame_finalize(const double *__restrict__ partials, uint32_t nblocks,
                                                        const DevCounters *__restrict__ ctr,
                                                        const VoxelParams *__restrict__ voxp,
                                                        FrameOut *__restrict__ out, uint32_t row_tile)
{
    __shared__ double red[256 * 6];
    frame_finalize_block(partials, nblocks, ctr, voxp, out, red, row_tile);
}

uint32_t launch_scatter_partials(const float4 *vnorm4, const uint32_t *n_ptr, uint32_t n_cap, double wf, Slot &sl,
                                 hipStream_t s)
{
    // ~4096 points per block (16 per thread in flight), at least 64 and at most kScatterBlocks blocks: a 1 M-point frame
    // leaves ~200 partial rows for the single-block finalizer instead of 1024 (its reduction is pure latency)
    uint32_t nb = (n_cap + 4095) / 4096;
    if (nb < 64u) nb = (n_cap + 255) / 256 < 64u ? (n_cap + 255) / 256 : 64u;
    if (nb > (uint32_t)kScatterBlocks) nb = kScatterBlocks;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_scatter_partials, dim3(nb), dim3(256), 0, s, vnorm4, n_ptr, n_cap, .001 / wf, sl.partials);
    return nb;
}

void launch_frame_finalize(const double *partials, uint32_t n_partials, uint32_t row_tile, Slot &sl, hipStream_t s)
{
    hipLaunchKernelGGL(k_frame_finalize, dim3(1), dim3(256), 0, s, partials, n_partials,
                       (const DevCounters *)sl.ctr, (const VoxelParams *)sl.voxp, sl.d_out, row_tile);
}

}  // namespace gm
