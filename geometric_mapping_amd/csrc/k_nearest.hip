// k_nearest.hip -- nearest cloud point of every voxel centroid.
//
// Replaces kdtree->nearestKSearch(centroid, 1, ...) in the marker loop of
// rvizNormals (/root/reference src/tunnel_processing.cpp:237-239).  Only the
// /surfaceNormals display needs it (displayNormals=false in the launch file).
//
// Queries are few (one per occupied voxel), points are many: one lane per query,
// the cloud is cut into chunks over blockIdx.y, each (query tile, chunk) keeps its
// best (d2, index) and merges with a 64-bit atomicMin on (d2 bits << 32 | index):
// non-negative float bits order like unsigned ints, so the minimum is the nearest
// point and, on equal distance, the lowest index -- deterministic.
#include "gm_internal.hpp"

namespace gm {

constexpr int kNnChunk = 4096;

__global__ __launch_bounds__(256) void k_nn_init(unsigned long long *__restrict__ best, const uint32_t *__restrict__ nq_ptr,
                                                 uint32_t nq_host)
{
    const uint32_t nq = nq_ptr ? *nq_ptr : nq_host;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x)
        best[i] = 0xFFFFFFFFFFFFFFFFull;
}

__global__ __launch_bounds__(256) void k_nn_scan(const float4 *__restrict__ pts, const uint32_t *__restrict__ n_ptr,
                                                 uint32_t n_host, const float4 *__restrict__ queries,
                                                 const uint32_t *__restrict__ nq_ptr, uint32_t nq_host,
                                                 unsigned long long *__restrict__ best)
{
    __shared__ float4 win[256];
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t nq = nq_ptr ? *nq_ptr : nq_host;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nq) return;
    const uint32_t c0 = blockIdx.y * kNnChunk;
    if (c0 >= n) return;
    const uint32_t c1 = (c0 + kNnChunk < n) ? c0 + kNnChunk : n;
    const float4 qq = queries[q < nq ? q : 0];
    float bd = 3.0e38f;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t b = c0; b < c1; b += 256) {
        const uint32_t i = b + threadIdx.x;
        __syncthreads();
        win[threadIdx.x] = (i < c1) ? pts[i] : make_float4(3e18f, 3e18f, 3e18f, 0.f);
        __syncthreads();
        const int m = (c1 - b < 256u) ? (int)(c1 - b) : 256;
        for (int j = 0; j < m; ++j) {
            const float4 p = win[j];
            const float dx = qq.x - p.x, dy = qq.y - p.y, dz = qq.z - p.z;
            // FLANN L2_Simple order, every operation rounded
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d2 < bd) { bd = d2; bi = b + j; }  // ascending index: first minimum wins
        }
    }
    if (q < nq && bi != 0xFFFFFFFFu) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | bi;
        atomicMin(&best[q], key);
    }
}

__global__ __launch_bounds__(256) void k_nn_unpack(const unsigned long long *__restrict__ best,
                                                   const uint32_t *__restrict__ nq_ptr, uint32_t nq_host,
                                                   int32_t *__restrict__ idx, const float4 *__restrict__ attr,
                                                   float4 *__restrict__ attr_out)
{
    const uint32_t nq = nq_ptr ? *nq_ptr : nq_host;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        const unsigned long long k = best[i];
        const int32_t j = (k == 0xFFFFFFFFFFFFFFFFull) ? -1 : (int32_t)(uint32_t)(k & 0xFFFFFFFFull);
        idx[i] = j;
        // normals->at(kIndices[0]) of the marker loop (src/tunnel_processing.cpp:247-249), gathered where the index is made
        if (attr_out) {
            const float nanv = __builtin_nanf("");
            attr_out[i] = j >= 0 ? attr[j] : make_float4(nanv, nanv, nanv, nanv);
        }
    }
}

void launch_nearest(const float4 *pts, const uint32_t *n_ptr, uint32_t n_cap, const float4 *queries,
                    const uint32_t *nq_ptr, uint32_t nq_cap, unsigned long long *best, int32_t *idx, hipStream_t s,
                    const float4 *attr, float4 *attr_out)
{
    if (nq_cap == 0) return;
    uint32_t gq = (nq_cap + 255) / 256;
    uint32_t gi = gq < 1024 ? gq : 1024;
    hipLaunchKernelGGL(k_nn_init, dim3(gi), dim3(256), 0, s, best, nq_ptr, nq_cap);
    if (n_cap) {
        const uint32_t chunks = (n_cap + kNnChunk - 1) / kNnChunk;
        hipLaunchKernelGGL(k_nn_scan, dim3(gq, chunks), dim3(256), 0, s, pts, n_ptr, n_cap, queries, nq_ptr, nq_cap,
                           best);
    }
    hipLaunchKernelGGL(k_nn_unpack, dim3(gi), dim3(256), 0, s, (const unsigned long long *)best, nq_ptr, nq_cap, idx, attr, attr_out);
}

}  // namespace gm
