// valu_rate.hip -- fp32 VALU issue-rate probe for gfx950: lane-FMAs per second of scalar v_fma_f32 against packed
// v_pk_fma_f32 at 1, 2 and 4 waves per SIMD.  Evidence for DESIGN.md's "packed fp32 does not raise the lane rate".
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (PACKED) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f x = {r[i], r[i + 1]}, aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(aa), "v"(bb));
                r[i] = x.x; r[i + 1] = x.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2)   // 256-thread block = 1 wave per SIMD
        for (int p = 0; p < 2; ++p) {
            const int nb = 256 * blocks_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (p) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, d, iters, 0.999f, 1e-3f);
                else hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, d, iters, 0.999f, 1e-3f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)nb * 256 * 16.0 * iters;
            std::printf("%s waves/SIMD=%d  %.3f ms  %.1f T lane-FMA/s  (%.1f TFLOP/s)\n", p ? "v_pk_fma_f32" : "v_fma_f32   ",
                        blocks_per_cu, ms, fma / ms * 1e-9, 2 * fma / ms * 1e-9);
        }
    return 0;
}
