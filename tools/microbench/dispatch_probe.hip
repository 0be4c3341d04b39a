// dispatch_probe.hip -- how fast does the chip fill with waves of a given shape?
// Standalone (hipcc --offload-arch=gfx950 -O3 tools/microbench/dispatch_probe.hip -o build/dispatch_probe); run on the GPU box.
// Every wave stamps the 100 MHz wall clock at its first instruction, then spins for ~SPIN us so that the first
// 4 096 waves (256 CUs x 16) are all resident while the fill is observed.  Shapes: registers per lane (64 / 128, by
// launch bounds and a live array), LDS per block (0 / 10 / 40 KB), scratch (a runtime-indexed private array), block
// size (64 / 256 / 512 threads).  Printed: waves started 5, 10, 15, 20, 30, 40 us after the first one, and the rate
// between the 10 % and 90 % marks of the first 4 096.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS, int MINW, int LDS_KB, int SCRATCH>
__global__ __launch_bounds__(THREADS, MINW) void k_probe(unsigned long long *__restrict__ t_start, int spin_ticks, float *__restrict__ sink,
                                                        int idx)
{
    const unsigned long long t0 = wall_clock64();
    __shared__ unsigned char lds[LDS_KB > 0 ? LDS_KB * 1024 : 16];
    if (threadIdx.x == 0) lds[(LDS_KB > 0 ? LDS_KB * 1024 : 16) - 1] = (unsigned char)idx;
    float keep[SCRATCH > 0 ? SCRATCH : 1];
    if (SCRATCH > 0) {
#pragma unroll 1
        for (int i = 0; i < SCRATCH; ++i) keep[i] = (float)(i + idx);
    }
    // registers: MINW decides the budget; make the compiler use it with a live array that stays in registers
    constexpr int NR = MINW <= 4 ? 96 : 40;
    float r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = (float)(threadIdx.x + i);
    const int wave = (blockIdx.x * THREADS + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) t_start[wave] = t0;
    while ((long long)(wall_clock64() - t0) < spin_ticks) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += r[i];
    if (SCRATCH > 0) s += keep[(idx + threadIdx.x) % SCRATCH];
    if (s == 123.456f) sink[0] = s + lds[threadIdx.x % 16];
}

template <int THREADS, int MINW, int LDS_KB, int SCRATCH>
static void run(const char *name, int blocks, unsigned long long *d_t, float *sink)
{
    const int waves = blocks * THREADS / 64;
    std::vector<unsigned long long> h(waves);
    double best_rate = 0;
    int at[6] = {0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_t, 0, sizeof(unsigned long long) * waves));
        hipLaunchKernelGGL((k_probe<THREADS, MINW, LDS_KB, SCRATCH>), dim3(blocks), dim3(THREADS), 0, 0, d_t, 6000 /* 60 us */, sink, rep);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const unsigned long long t0 = h[0];
        const int marks[6] = {5, 10, 15, 20, 30, 40};
        for (int m = 0; m < 6; ++m)
            at[m] = (int)(std::lower_bound(h.begin(), h.end(), t0 + (unsigned long long)marks[m] * 100ull) - h.begin());
        const int n = waves < 4096 ? waves : 4096;
        const double dt = (double)(h[n * 9 / 10] - h[n / 10]) / 100.0;   // us
        best_rate = dt > 0 ? (n * 0.8) / dt : 0;
    }
    printf("%-44s started after 5/10/15/20/30/40 us: %5d %5d %5d %5d %5d %5d   fill rate %6.0f waves/us\n", name, at[0], at[1], at[2],
           at[3], at[4], at[5], best_rate);
}

int main()
{
    unsigned long long *d_t; float *sink;
    CK(hipMalloc(&d_t, sizeof(unsigned long long) * 65536)); CK(hipMalloc(&sink, 64));
    const int B = 3713;
    run<256, 4, 40, 0>("256 thr, 128 regs, 40 KB LDS (k_normals)", B, d_t, sink);
    run<256, 4, 40, 256>("256 thr, 128 regs, 40 KB LDS, scratch", B, d_t, sink);
    run<256, 4, 10, 0>("256 thr, 128 regs, 10 KB LDS", B, d_t, sink);
    run<256, 4, 0, 0>("256 thr, 128 regs, no LDS", B, d_t, sink);
    run<256, 8, 40, 0>("256 thr, 64 regs, 40 KB LDS", B, d_t, sink);
    run<256, 8, 0, 0>("256 thr, 64 regs, no LDS", B, d_t, sink);
    run<64, 4, 10, 0>("64 thr, 128 regs, 10 KB LDS", B * 4, d_t, sink);
    run<512, 4, 80, 0>("512 thr, 128 regs, 80 KB LDS", B / 2, d_t, sink);
    run<1024, 4, 160, 0>("1024 thr, 128 regs, 160 KB LDS", B / 4, d_t, sink);
    return 0;
}
