// k_voxel.hip -- voxel binning: pcl::VoxelGrid centroid down-sampling.
//
// Replaces the VoxelGrid call inside rvizNormals
// (/root/reference src/tunnel_processing.cpp:214-220; PCL semantics restated in
// oracle/gm_oracle.c gmo_voxel_grid): min/max of the cloud -> min_b = floor(min *
// inv_leaf) -> key = (ix-min_bx) + (iy-min_by)*divx + (iz-min_bz)*divx*divy ->
// sort by key -> one centroid per run of equal keys, ascending key order.
// The reference runs this every frame (geometric_mapping.cpp:70-75 is not gated
// by displayNormals), so it is part of the timed path.
//
// Device shape: keys (1 streaming pass) -> stable radix sort (k_sort.hip) ->
// segment heads by order-preserving compaction -> one wave per voxel sums its
// points in fp64 and divides once.
#include "gm_compact.hpp"
#include "gm_internal.hpp"

namespace gm {

__global__ void k_voxel_setup(const DevCounters *__restrict__ ctr, float leaf, VoxelParams *__restrict__ vp)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // setLeafSize(float...) ; inverse_leaf_size_ = 1 / leaf in float
    const float inv = 1.0f / leaf;
    vp->inv_leaf = inv;
    vp->passthrough = 0;
    if (ctr->vox_n == 0) {
        for (int k = 0; k < 3; ++k) { vp->min_b[k] = 0; vp->div_b[k] = 1; }
        vp->mul1 = 1; vp->mul2 = 1;
        return;
    }
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = ordered_to_float(ctr->mm[k]); mx[k] = ordered_to_float(ctr->mm[3 + k]); }
    // "Leaf size is too small" guard: int64 product of per-axis extents
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1;
    const long long dy = (long long)((mx[1] - mn[1]) * inv) + 1;
    const long long dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > 2147483647ll) vp->passthrough = 1;
    for (int k = 0; k < 3; ++k) {
        const int lo = (int)floorf(mn[k] * inv), hi = (int)floorf(mx[k] * inv);
        vp->min_b[k] = lo;
        vp->div_b[k] = hi - lo + 1;
    }
    vp->mul1 = vp->div_b[0];
    vp->mul2 = vp->div_b[0] * vp->div_b[1];
}

__global__ __launch_bounds__(256) void k_voxel_keys(const float4 *__restrict__ pts, const DevCounters *__restrict__ ctr,
                                                    const VoxelParams *__restrict__ vp, uint32_t *__restrict__ keys)
{
    const uint32_t n = ctr->vox_n;
    const float inv = vp->inv_leaf;
    const float bx = (float)vp->min_b[0], by = (float)vp->min_b[1], bz = (float)vp->min_b[2];
    const int m1 = vp->mul1, m2 = vp->mul2;
    const bool pass = vp->passthrough != 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        const int i0 = (int)(floorf(p.x * inv) - bx);
        const int i1 = (int)(floorf(p.y * inv) - by);
        const int i2 = (int)(floorf(p.z * inv) - bz);
        keys[i] = pass ? i : (uint32_t)(i0 + i1 * m1 + i2 * m2);
    }
}

struct HeadPred {
    const uint32_t *__restrict__ skeys;
    typedef NoPayload Payload;
    __device__ __forceinline__ bool operator()(uint32_t s, Payload &) const { return s == 0 || skeys[s] != skeys[s - 1]; }
};
struct HeadEmit {
    static constexpr bool kHasFinish = false, kHasPrepare = false;
    uint32_t *__restrict__ seg_start;
    __device__ __forceinline__ void operator()(uint32_t src, uint32_t dst, const NoPayload &) const { seg_start[dst] = src; }
};

__global__ __launch_bounds__(256) void k_voxel_centroids(const float4 *__restrict__ pts,
                                                         const uint32_t *__restrict__ perm,
                                                         const uint32_t *__restrict__ seg_start,
                                                         const DevCounters *__restrict__ ctr,
                                                         float4 *__restrict__ vox4)
{
    const uint32_t n = ctr->vox_n, V = ctr->n_voxels;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t nwaves = gridDim.x * blockDim.x / kWave;
    for (uint32_t v = wave; v < V; v += nwaves) {
        const uint32_t b = seg_start[v], e = (v + 1 < V) ? seg_start[v + 1] : n;
        double sx = 0, sy = 0, sz = 0;
        for (uint32_t s = b + lane_id(); s < e; s += kWave) {
            const float4 p = pts[perm[s]];
            sx += p.x; sy += p.y; sz += p.z;
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane_id() == 0) {
            const double c = (double)(e - b);
            vox4[v] = make_float4((float)(sx / c), (float)(sy / c), (float)(sz / c), (float)(e - b));
        }
    }
}

// ---- dense-table fast path: ascending table order IS ascending pcl key order ----
struct DensePred {
    const VoxCell *__restrict__ table;
    typedef NoPayload Payload;
    __device__ __forceinline__ bool operator()(uint32_t i, Payload &) const { return table[i].cnt != 0; }
};
struct DenseEmit {
    static constexpr bool kHasFinish = false, kHasPrepare = false;
    const VoxCell *__restrict__ table;
    float4 *__restrict__ vox4;
    double lo, inv_scale;
    __device__ __forceinline__ void operator()(uint32_t src, uint32_t dst, const NoPayload &) const
    {
        const VoxCell c = table[src];
        const double inv = inv_scale / (double)c.cnt;
        vox4[dst] = make_float4((float)(lo + (double)c.sx * inv), (float)(lo + (double)c.sy * inv),
                                (float)(lo + (double)c.sz * inv), (float)c.cnt);
    }
};

void launch_voxel_dense_finalize(const VoxDense &vd, Slot &sl, hipStream_t s)
{
    const uint32_t cells = (uint32_t)vd.dim * vd.dim * vd.dim;
    DensePred pred{sl.vox_table};
    DenseEmit emit{sl.vox_table, sl.vox4, (double)vd.lo, vd.inv_scale};
    hipLaunchKernelGGL((k_compact<DensePred, DenseEmit>), dim3(compact_grid(cells)), dim3(kCpThreads), 0, s, pred, emit,
                       (const uint32_t *)nullptr, cells, next_scan(sl), &sl.ctr->n_voxels, (uint32_t *)nullptr);
}

void launch_voxel_grid(Slot &sl, uint32_t n_cap, float leaf, int key_bits, hipStream_t s)
{
    hipLaunchKernelGGL(k_voxel_setup, dim3(1), dim3(64), 0, s, (const DevCounters *)sl.ctr, leaf, sl.voxp);
    if (n_cap == 0) return;
    uint32_t gb = (n_cap + 255) / 256;
    if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(k_voxel_keys, dim3(gb), dim3(256), 0, s, (const float4 *)sl.valid4, (const DevCounters *)sl.ctr,
                       (const VoxelParams *)sl.voxp, sl.keys_a);
    const int where = launch_radix_sort(sl.keys_a, sl.vals_a, sl.keys_b, sl.vals_b, &sl.ctr->vox_n, n_cap, key_bits,
                                        sl, false, s);
    const uint32_t *skeys = where ? sl.keys_b : sl.keys_a;
    const uint32_t *perm = where ? sl.vals_b : sl.vals_a;
    HeadPred pred{skeys};
    HeadEmit emit{sl.seg_start};
    hipLaunchKernelGGL((k_compact<HeadPred, HeadEmit>), dim3(compact_grid(n_cap)), dim3(kCpThreads), 0, s, pred, emit,
                       (const uint32_t *)&sl.ctr->vox_n, 0u, next_scan(sl), &sl.ctr->n_voxels, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_voxel_centroids, dim3(gb), dim3(256), 0, s, (const float4 *)sl.valid4, perm,
                       (const uint32_t *)sl.seg_start, (const DevCounters *)sl.ctr, sl.vox4);
}

}  // namespace gm
