// stream_probe.hip -- what bounds a "read 16-byte rows, write the survivors" stream on this chip?
// Standalone (hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_probe.hip -o build/stream_probe); run on the GPU box.
// Variants of one 10 M-row pass (160 MB in), timed with hipEvents, best and median of 20:
//   copy        grid-stride float4 copy (the guide's 6.29 TB/s shape)
//   tile        512-thread blocks, ITEMS rows per thread loaded first, then stored to the same index (the compaction's phases)
//   tile_shift  the same, stored 3 rows further (stores not aligned to 128-byte lines)
//   tile_rank   the same, rows with (i % 6 == 5) dropped: survivors stored at tile base + rank inside the wave's 64 (ballot),
//               i.e. the compaction's store pattern without any scan
//   tile_rank_key  + a 4-byte key per survivor into a second array
//   read_only   tile loads only (sum into one word per block)
//   write_only  tile stores only
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <int ITEMS, int MODE>
__global__ __launch_bounds__(512) void k_tile(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t *__restrict__ keys,
                                              uint32_t n, float *__restrict__ sink)
{
    const uint32_t base = blockIdx.x * (512u * ITEMS);
    float4 v[ITEMS];
    if (MODE != 5) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t i = base + j * 512u + threadIdx.x;
            v[j] = i < n ? in[i] : make_float4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j] = make_float4((float)threadIdx.x, (float)j, 1.f, 2.f);
    }
    if (MODE == 4) {   // read only
        float s = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) s += v[j].x + v[j].y + v[j].z;
        if (s == 12345.678f) sink[blockIdx.x] = s;
        return;
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = base + j * 512u + threadIdx.x;
        if (i >= n) continue;
        if (MODE == 0 || MODE == 5) out[i] = v[j];
        else if (MODE == 1) out[i + 3] = v[j];
        else {
            const bool keep = (i % 6u) != 5u;
            const unsigned long long m = __ballot(keep);
            const uint32_t dst = (i - lane) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                out[dst] = v[j];
                if (MODE == 3) keys[dst] = i * 2654435761u;
            }
        }
    }
}

template <class F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int r = 0; r < 23; ++r) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (r >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-22s best %7.1f us  median %7.1f us   %6.2f TB/s (median)\n", name, ms[0] * 1e3, ms[ms.size() / 2] * 1e3,
           bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
}

template <int ITEMS>
static void run_tiles(const float4 *in, float4 *out, uint32_t *keys, uint32_t n, float *sink)
{
    const uint32_t nb = (n + 512u * ITEMS - 1) / (512u * ITEMS);
    char nm[64];
    const double rd = 16.0 * n, wr = 16.0 * n;
    snprintf(nm, sizeof nm, "tile x%d", ITEMS);
    timeit(nm, rd + wr, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 0>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
    snprintf(nm, sizeof nm, "tile_shift x%d", ITEMS);
    timeit(nm, rd + wr, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 1>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
    snprintf(nm, sizeof nm, "tile_rank x%d", ITEMS);
    timeit(nm, rd + wr * 5 / 6, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 2>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
    snprintf(nm, sizeof nm, "tile_rank_key x%d", ITEMS);
    timeit(nm, rd + (wr + 4.0 * n) * 5 / 6, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 3>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
    snprintf(nm, sizeof nm, "read_only x%d", ITEMS);
    timeit(nm, rd, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 4>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
    snprintf(nm, sizeof nm, "write_only x%d", ITEMS);
    timeit(nm, wr, [&] { hipLaunchKernelGGL((k_tile<ITEMS, 5>), dim3(nb), dim3(512), 0, 0, in, out, keys, n, sink); });
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    float4 *in, *out; uint32_t *keys; float *sink;
    CK(hipMalloc(&in, 16ull * (n + 64))); CK(hipMalloc(&out, 16ull * (n + 64))); CK(hipMalloc(&keys, 4ull * (n + 64)));
    CK(hipMalloc(&sink, 4u << 20));
    CK(hipMemset(in, 0x3c, 16ull * (n + 64))); CK(hipMemset(out, 0, 16ull * (n + 64)));
    printf("rows %u (%.0f MB in)\n", n, 16.0 * n / 1e6);
    timeit("copy 2048x256", 32.0 * n, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, in, out, (size_t)n); });
    timeit("copy 8192x256", 32.0 * n, [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, in, out, (size_t)n); });
    run_tiles<4>(in, out, keys, n, sink);
    run_tiles<8>(in, out, keys, n, sink);
    run_tiles<16>(in, out, keys, n, sink);
    return 0;
}
