// gm_device.hpp -- device-side helpers shared by the gfx950 kernels.
// Wave = 64 lanes everywhere (CDNA4); nothing here is written for 32-wide warps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gm {

constexpr int kWave = 64;

// Per-frame device counters.  Zeroed by one hipMemsetAsync at the head of a
// frame; every data-dependent size of the pipeline lives here so that no stage
// needs a host round-trip.
struct DevCounters {
    uint32_t n_cropped;   // points surviving the crop box
    uint32_t n_valid;     // points with a finite normal (and owned, when sharded)
    uint32_t n_tiles;     // query tiles built from the sorted cell keys
    uint32_t reserved0;
    uint32_t n_voxels;    // occupied voxels
    uint32_t vox_n;       // points entering the voxel grid (= n_valid)
    uint32_t mm[6];       // ordered-uint encodings: min x,y,z then max x,y,z of the valid cloud
    uint32_t scratch_total;
    uint32_t pad[3];
};

// Voxel-grid parameters derived on the device from the min/max of the cloud
// (pcl::VoxelGrid::applyFilter: min_b_, div_b_, divb_mul_).
struct VoxelParams {
    float inv_leaf;
    int32_t min_b[3];
    int32_t div_b[3];
    int32_t mul1, mul2;
    uint32_t passthrough;
};

// Uniform cell grid used for the fixed-radius search (stands in for the FLANN
// kd-tree of src/tunnel_processing.cpp:62-63: same neighbour sets).
struct GridParams {
    float ox, oy, oz;   // origin (lower corner)
    float inv_h;        // 1 / cell edge in y and z, edge >= 1.001 * radius
    float inv_hx;       // 1 / cell edge in x = fine * inv_h: x is binned `fine` times finer, so that
                        // the points of one x-row are sorted by x and candidate ranges can be cut to
                        // [xmin - r, xmax + r] of a tile instead of whole cells
    int32_t nx, ny, nz; // cells per axis (nx counts the fine x cells)
    int32_t xreach;     // fine x cells that cover the radius (fine + 1)
    float r2;           // (float)(radius*radius): KdTreeFLANN::radiusSearch's cast
    float r2_scale;     // power of two s with r2 * s ~ 2^100: k_normals evaluates d2 < r2 as clamp01(fma(d2, -s, r2 * s)),
                        // exact because one ulp of a d2 next to r2, times s, is >= 2^76
};

// Dense voxel table (fast path of the VoxelGrid stage): when the crop box bounds
// the voxel lattice to a small table, per-voxel sums are accumulated as exact
// 64-bit fixed-point integers (order-free, so bit-reproducible) right where the
// normals are produced, and the sorted output is a compaction of the table.
struct VoxCell {
    unsigned long long sx, sy, sz;  // sum of (coord - lo) * scale, rounded to integer per point
    uint32_t cnt, pad;
};
struct VoxDense {
    uint32_t enabled;
    int32_t i_lo;        // floor(lo * inv_leaf): lattice index of the box's lower face
    int32_t dim;         // lattice cells per axis covered by the box
    float inv_leaf;      // 1 / (float)leaf, as pcl::VoxelGrid computes it
    float lo;            // lower face of the crop box
    float own_lo, own_hi;// slab ownership (multi-GPU), +-inf otherwise
    double scale, inv_scale;
    uint32_t xcd_chunk;  // k_normals launch option riding along: blocks per XCD chunk of the tile mapping (0 = round-robin)
    uint32_t pad_;
};

// extension outputs (RANSAC plane / cylinder; no reference counterpart)
struct FrameExt {
    float plane[4];
    float cylinder[7];
    uint32_t plane_inliers, cylinder_inliers;
    uint32_t pad;
    double plane_refit[4];
    double cyl_axis_refit[3];
};

struct FrameOut {        // device -> host result record (one small D2H per frame)
    DevCounters ctr;
    float evals[3];
    float evecs[9];      // column-major
    double scatter[6];
    VoxelParams vox;
    FrameExt ext;
};

// ---- wave-level primitives ---------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

template <class T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

// inclusive scan of one value per lane across the wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t t = __shfl_up(v, o, kWave);
        if (lane_id() >= o) v += t;
    }
    return v;
}

// LDS traffic of ONE wave is issued and serviced in order; this only stops the
// compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// total order on floats as unsigned ints (for atomicMin/atomicMax)
__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// first index in sorted keys[0..n) whose key is >= key
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ keys, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h, int n)
{
    int c = (int)floorf((x - o) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__device__ __forceinline__ uint32_t cell_key(const GridParams &g, float x, float y, float z)
{
    int cx = cell_coord(x, g.ox, g.inv_hx, g.nx);
    int cy = cell_coord(y, g.oy, g.inv_h, g.ny);
    int cz = cell_coord(z, g.oz, g.inv_h, g.nz);
    return (uint32_t)((cz * g.ny + cy) * g.nx + cx);
}

// ---- symmetric 3x3 eigen-solvers (fp64) --------------------------------------

// Cyclic Jacobi; a = {xx,xy,xz,yy,yz,zz}.  w ascending, V column-major.
// One thread, once per frame (the K11 epilogue) -- clarity over speed.
__host__ __device__ inline void jacobi_eig3(const double a6[6], double w[3], double V[9])
{
    double a[3][3] = {{a6[0], a6[1], a6[2]}, {a6[1], a6[3], a6[4]}, {a6[2], a6[4], a6[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * dg || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (a[ord[j]][ord[j]] < a[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int c = 0; c < 3; ++c) {
        w[c] = a[ord[c]][ord[c]];
        for (int r = 0; r < 3; ++r) V[3 * c + r] = v[r][ord[c]];
    }
}

}  // namespace gm
