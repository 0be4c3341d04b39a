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
// state of one chained-scan launch (gm_compact.hpp)
struct ScanState {
    unsigned long long *status;  // [>= grid size] tile records
    uint32_t *ticket;            // 0 between launches
    uint32_t epoch;              // 1 .. 2^30-1, different from the previous launches' on this array
    const uint32_t *frame_ptr;   // graph replay (kernel arguments are frozen): epoch = f(*frame_ptr, epoch), see scan_epoch()
};

// Epoch of a chained-scan launch.  Launches enqueued one by one take it from the host (a per-slot counter, range
// [1, 2^29)); launches replayed from a captured graph cannot -- their arguments are frozen -- and derive it from the
// slot's device-side frame counter (bumped by the frame's opening kernel) and the launch's index in the frame, range
// [2^29, 2^30): the two ranges never meet, so records of the one kind are never taken for records of the other.
__device__ __forceinline__ uint32_t scan_epoch(const ScanState &st)
{
    return st.frame_ptr ? 0x20000000u + (((*st.frame_ptr) * 32u + st.epoch) & 0x1FFFFFFFu) : st.epoch;
}

#ifndef GM_TILE_CLASSES
#define GM_TILE_CLASSES 8
#endif
constexpr int kTileListClasses = GM_TILE_CLASSES;   // cost classes of k_normals' tile list (k_normals.hip)

struct DevCounters {
    uint32_t n_cropped;   // points surviving the crop box
    uint32_t n_valid;     // points with a finite normal (and owned, when sharded)
    uint32_t n_tiles;     // (unused since the tile list has cost classes: n_tiles_c below)
    uint32_t first_drop_enc;   // 0xFFFFFFFF - (first cropped index without a finite normal); 0 = none (NaN-normal compaction)
    uint32_t n_voxels;    // occupied voxels
    uint32_t vox_n;       // points entering the voxel grid (= n_valid)
    uint32_t mm[6];       // ordered-uint encodings: min x,y,z then max x,y,z of the valid cloud
    uint32_t scratch_total;
    uint32_t pad[5];      // (diagnostic builds count candidate streams here)
    // tiles per cost class (k_rows_and_tiles files a tile by its x extent; k_normals runs class 0 first), one counter per
    // 128-byte line: every cutter block adds to every class, and returning atomics on one line are served one after the
    // other (~88 per us) -- eight counters in one line cost the tile builder 4.5 us
    uint32_t n_tiles_c[kTileListClasses][32];
};
static_assert(sizeof(DevCounters) % 8 == 0, "zero-filled in 8-byte words");

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
    float snap;         // power of two >= one ulp of the largest coordinate of the grid box: tile origins of the
                        // matrix-core kernel are multiples of it, which makes point - origin exact up to the rounding of a
                        // radius-sized number
    int32_t D;          // y/z rows per radius (1 .. 4): the cells are 1.001 r / D wide in y and z, a tile's candidates lie
                        // in the (2D+1)^2 rows around its own.  D > 1 pays when a tile is short against the radius (many
                        // neighbours per point): the far rows need a shorter x-window, sqrt(r^2 - gap_y^2 - gap_z^2).
    int16_t reach[5][5];// [|row offset y|][|row offset z|]: fine x cells that cover that window half-width; 0 = no row
    float dscale;       // power of two >= 1 / band: the distance MFMA of k_normals computes T = (r2 - d2) * dscale, so that
                        // one v_cvt_pk_bf16_f32 with clamp turns two of them into two 0/1 weights (every T in (0, 1) lies
                        // inside the band and is re-evaluated exactly)
    float dband;        // band * dscale, in [1, 2)
    float band;         // half-width of the band around r2 inside which the distance MFMA's value does not decide a pair
                        // (2e-5 * (1.001 r)^2, whatever D is: analytic worst case of the distance product 1.7e-5 of that
                        // square, largest error measured by the diagnostic build over every tested pair of the 1 M-point frame
                        // 4.2e-6 -- profiles/r02_distance_mfma_error.json, tools/microbench/mfma_probe.hip)
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

// Symmetric 3x3 eigen-decomposition the way Eigen::SelfAdjointEigenSolver<MatrixXf>::compute reaches it for a
// dynamic-size matrix (/root/reference src/tunnel_processing.cpp:129): scale by max|a_ij|, one Householder reflection
// to tridiagonal form, implicit symmetric QR steps with Wilkinson shift and Givens rotations accumulated into the
// eigenvector matrix, ascending sort -- in the scalar type T with T's convergence thresholds.  The SIGNS of the
// returned eigenvectors (arbitrary mathematically, visible on /eigenBasisOutput as the arrow directions) depend on the
// number of QR sweeps, i.e. on the precision the iteration runs in: T = float reproduces the sign pattern of the
// reference's MatrixXf solve.  a = {xx,xy,xz,yy,yz,zz}; w ascending, V column-major.
template <class T>
__host__ __device__ inline void eigen_tridiag_qr3(const T a6[6], T w[3], T V[9])
{
    typedef T double_;   // (the body below is written once for both scalar types)
    double_ scale = 0.0;
    for (int k = 0; k < 6; ++k) { const double_ t = a6[k] < 0 ? -a6[k] : a6[k]; if (t > scale) scale = t; }
    if (!(scale > 0.0)) scale = 1.0;
    const double_ a00 = a6[0] / scale, a10 = a6[1] / scale, a20 = a6[2] / scale, a11 = a6[3] / scale, a21 = a6[4] / scale,
                 a22 = a6[5] / scale;
    double_ Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double_ diag[3], sub[2];
    const double_ tiny = sizeof(T) == 4 ? (T)1.17549435e-38 : (T)2.2250738585072014e-308;
    if (a20 * a20 <= tiny) {              // column already tridiagonal: no reflection
        diag[0] = a00; diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
    } else {
        double_ beta = (T)sqrt((double)(a10 * a10 + a20 * a20));
        if (a10 >= 0) beta = -beta;
        const double_ ess = a20 / (a10 - beta), tau = (beta - a10) / beta;
        const double_ v[2] = {1.0, ess};
        double_ B[2][2] = {{a11, a21}, {a21, a22}};
        const double_ p0 = tau * (B[0][0] * v[0] + B[0][1] * v[1]), p1 = tau * (B[1][0] * v[0] + B[1][1] * v[1]);
        const double_ K = -0.5 * tau * (p0 * v[0] + p1 * v[1]);
        const double_ ww[2] = {p0 + K * v[0], p1 + K * v[1]};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) B[i][j] -= v[i] * ww[j] + ww[i] * v[j];
        diag[0] = a00; diag[1] = B[0][0]; diag[2] = B[1][1]; sub[0] = beta; sub[1] = B[1][0];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) Q[1 + i][1 + j] = (i == j ? 1.0 : 0.0) - tau * v[i] * v[j];
    }
    int end = 2, start = 0, iter = 0;
    const double_ prec = sizeof(T) == 4 ? (T)(2.0 * 1.1920928955078125e-7) : (T)(2.0 * 2.220446049250313e-16);
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            const double_ s = sub[i] < 0 ? -sub[i] : sub[i];
            const double_ d = (diag[i] < 0 ? -diag[i] : diag[i]) + (diag[i + 1] < 0 ? -diag[i + 1] : diag[i + 1]);
            if (s <= d * prec || s <= tiny) sub[i] = 0.0;
        }
        while (end > 0 && sub[end - 1] == 0.0) --end;
        if (end <= 0) break;
        if (++iter > 90) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.0) --start;
        const double_ td = (diag[end - 1] - diag[end]) * 0.5, e = sub[end - 1];
        double_ mu = diag[end];
        if (td == 0.0) mu -= (e < 0 ? -e : e);
        else {
            const double_ h = (T)sqrt((double)(td * td + e * e));
            mu -= (e * e) / (td + (td > 0 ? h : -h));
        }
        double_ x = diag[start] - mu, z = sub[start];
        for (int k = start; k < end; ++k) {
            double_ c, s;   // Givens rotation taking (x, z) to (r, 0), Eigen's JacobiRotation::makeGivens conventions
            if (z == 0.0) { c = x < 0 ? -1.0 : 1.0; s = 0.0; }
            else if (x == 0.0) { c = 0.0; s = z < 0 ? 1.0 : -1.0; }
            else if ((x < 0 ? -x : x) > (z < 0 ? -z : z)) {
                const double_ t = z / x; double_ u = (T)sqrt((double)((T)1 + t * t));
                if (x < 0) u = -u;
                c = 1.0 / u; s = -t * c;
            } else {
                const double_ t = x / z; double_ u = (T)sqrt((double)((T)1 + t * t));
                if (z < 0) u = -u;
                s = -1.0 / u; c = -t * s;
            }
            const double_ sdk = s * diag[k] + c * sub[k], dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
            for (int r = 0; r < 3; ++r) {
                const double_ xi = Q[r][k], yi = Q[r][k + 1];
                Q[r][k] = c * xi - s * yi;
                Q[r][k + 1] = s * xi + c * yi;
            }
        }
    }
    for (int i = 0; i < 2; ++i) {          // ascending selection sort, columns follow
        int k = i;
        for (int j = i + 1; j < 3; ++j) if (diag[j] < diag[k]) k = j;
        if (k != i) {
            const double_ t = diag[i]; diag[i] = diag[k]; diag[k] = t;
            for (int r = 0; r < 3; ++r) { const double_ u = Q[r][i]; Q[r][i] = Q[r][k]; Q[r][k] = u; }
        }
    }
    for (int c = 0; c < 3; ++c) {
        w[c] = diag[c] * scale;
        for (int r = 0; r < 3; ++r) V[3 * c + r] = Q[r][c];
    }
}


// Accurate eigenpairs (fp64 Jacobi) with the column signs of the reference's float solve (above, T = float).
__host__ __device__ inline void eig3_sym_eigen_signs(const double a6[6], double w[3], double V[9])
{
    jacobi_eig3(a6, w, V);
    float af[6], wf[3], Vf[9];
    for (int k = 0; k < 6; ++k) af[k] = (float)a6[k];
    eigen_tridiag_qr3<float>(af, wf, Vf);
    for (int c = 0; c < 3; ++c) {
        const double d = V[3 * c] * (double)Vf[3 * c] + V[3 * c + 1] * (double)Vf[3 * c + 1] + V[3 * c + 2] * (double)Vf[3 * c + 2];
        if (d < 0.0)
            for (int r = 0; r < 3; ++r) V[3 * c + r] = -V[3 * c + r];
    }
}

// ---- the frame's closing step: fixed-order reduction of the scatter partials + 3x3 eigen + result record -------------
// (SelfAdjointEigenSolver, /root/reference src/tunnel_processing.cpp:129-137.)  One 256-thread block; shared by
// k_frame_finalize and, when a RANSAC model is part of the frame, the tail of k_ext_finalize (one launch less).
// red: __shared__ double[256 * 6].
__device__ __forceinline__ void frame_finalize_block(const double *__restrict__ partials, uint32_t nblocks,
                                                     const DevCounters *__restrict__ ctr,
                                                     const VoxelParams *__restrict__ voxp, FrameOut *__restrict__ out,
                                                     double *red, uint32_t row_tile = 0)
{
    // row_tile != 0: the rows are the NaN-normal compaction's, one per row_tile cropped points (gm_compact.hpp); tiles
    // past the end of the cropped cloud wrote none
    if (row_tile) {
        const uint32_t rows = (ctr->n_cropped + row_tile - 1u) / row_tile;
        nblocks = rows < nblocks ? rows : nblocks;
    }
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 6; ++k) m[k] += partials[b * 6 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x * 6 + k] = m[k];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[threadIdx.x * 6 + k] += red[(threadIdx.x + stride) * 6 + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double M[6], w[3], V[9];
        for (int k = 0; k < 6; ++k) { M[k] = red[k]; out->scatter[k] = M[k]; }
        eig3_sym_eigen_signs(M, w, V);   // fp64 Jacobi pairs, column signs of Eigen's float tridiagonal-QR solve
        for (int k = 0; k < 3; ++k) out->evals[k] = (float)w[k];
        for (int k = 0; k < 9; ++k) out->evecs[k] = (float)V[k];
        if (ctr) out->ctr = *ctr;
        if (voxp) out->vox = *voxp;
    }
}

}  // namespace gm
