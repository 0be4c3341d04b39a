// cvt_probe.hip -- does v_cvt_pk_bf16_f32 honour the VOP3 clamp bit on gfx950 (result clamped to [0, 1])?
// build: hipcc --offload-arch=gfx950 -O2 cvt_probe.hip -o cvt_probe ; prints input -> packed result with and without clamp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
__global__ void k(const float *a, unsigned *o, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = a[(i + 1) % n];
    unsigned r0, r1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r0) : "v"(x), "v"(y));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r1) : "v"(x), "v"(y));
    o[2 * i] = r0;
    o[2 * i + 1] = r1;
}
int main()
{
    const float vals[] = {-5.0f, -0.0f, 0.0f, 1e-30f, 0.3f, 0.5f, 0.999f, 1.0f, 1.5f, 2.0f, 1e30f, INFINITY, -INFINITY, NAN,
                          -0x1p122f, 0x1p-20f, 3.0e-39f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *da; unsigned *dout; unsigned h[2 * 32];
    hipMalloc(&da, sizeof(vals)); hipMalloc(&dout, sizeof(h));
    hipMemcpy(da, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, n);
    hipMemcpy(h, dout, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("x=%-14g y=%-14g  plain=%08x  clamp=%08x\n", vals[i], vals[(i + 1) % n], h[2 * i], h[2 * i + 1]);
    return 0;
}
