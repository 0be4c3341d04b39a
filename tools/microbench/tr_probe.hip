// tr_probe.hip -- semantics of gfx950's ds_read_b64_tr_b16 as k_normals' distance MFMA uses it: within a group of 16
// consecutive lanes, lane 4q+p supplies the address of 4 consecutive 16-bit elements (piece p of row q); lane i receives
// element (i & 3) of piece (i >> 2) of rows q = 0..3, i.e. "column i of the 4 rows".  Rows and pieces are placed at
// unrelated LDS addresses here to check that nothing but the per-lane addresses matters.
//   hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void k(short *out)
{
    __shared__ __attribute__((aligned(16))) short m[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) m[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    // piece (row q, piece p) of group g lives at a scrambled offset (multiple of 4 elements = 8 bytes)
    const int off = 4 * ((g * 16 + q * 4 + p) * 37 % 1021);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(m + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main()
{
    short *d, h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15;
            // expected: row q = j, piece p = i >> 2, element i & 3
            const int off = 4 * ((g * 16 + j * 4 + (i >> 2)) * 37 % 1021) + (i & 3);
            if (h[l * 4 + j] != (short)off) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h[l * 4 + j], off); ++bad; }
        }
    printf("ds_read_b64_tr_b16 semantics: %s (%d wrong of 256)\n", bad ? "DIFFERENT" : "as assumed", bad);
    return 0;
}
