// mfma_probe.hip -- what k_normals' matrix-core formulation needs to know about v_mfma_f32_32x32x16_bf16 on gfx950:
//   1. the A / B / D lane maps (exact integer data, asymmetric operands);
//   2. how the products of one instruction and a chain of instructions are accumulated (error against fp64 of a
//      long positive sum: round-to-nearest fp32 chain, truncation, or wider);
//   3. the error of a squared distance evaluated as a bf16x3-split bilinear form |u|^2 + |v|^2 - 2 u.v, relative
//      to r^2 (sets the width of the "uncertain" band of an MFMA neighbour predicate).
//   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static uint16_t bf16_rne(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float v, uint16_t s[3])
{
    s[0] = bf16_rne(v); float r = v - bf16_to_f(s[0]);
    s[1] = bf16_rne(r); r -= bf16_to_f(s[1]);
    s[2] = bf16_rne(r);
}

// D[32x32] = sum over `steps` of A_t[32x16] * B_t[16x32]; A as [steps][32][16], B as [steps][16][32], D as [32][32]
__global__ void k_mfma(const uint16_t *A, const uint16_t *B, float *D, int steps)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int t = 0; t < steps; ++t) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (short)A[(t * 32 + r) * 16 + 8 * h + j];   // A[row r][k = 8h + j]
            b[j] = (short)B[(t * 16 + 8 * h + j) * 32 + r];   // B[k = 8h + j][col r]
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;   // D[row][col r]
        D[row * 32 + r] = acc[reg];
    }
}

int main()
{
    std::mt19937 rng(7);
    uint16_t *dA, *dB; float *dD;
    const int max_steps = 64;
    hipMalloc(&dA, max_steps * 32 * 16 * 2); hipMalloc(&dB, max_steps * 16 * 32 * 2); hipMalloc(&dD, 32 * 32 * 4);
    std::vector<uint16_t> A(max_steps * 512), B(max_steps * 512);
    std::vector<float> D(1024);
    auto run = [&](int steps) {
        hipMemcpy(dA, A.data(), steps * 512 * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), steps * 512 * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    };
    // ---- 1. lane maps
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bf16_rne((float)((i * 7 + 3 * k) % 13 - 6));
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bf16_rne((float)((5 * k + j * 3) % 11 - 5));
    run(1);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        int s = 0;
        for (int k = 0; k < 16; ++k) s += ((i * 7 + 3 * k) % 13 - 6) * ((5 * k + j * 3) % 11 - 5);
        if ((float)s != D[i * 32 + j]) ++bad;
    }
    printf("lane_maps_32x32x16_bf16: %s (%d wrong of 1024)\n", bad ? "WRONG" : "ok", bad);

    // ---- 2. accumulation: positive terms, 1 step (16 products) and 40 steps (640 products)
    for (int steps : {1, 4, 40}) {
        std::uniform_real_distribution<float> U(0.5f, 1.0f);
        for (auto &x : A) x = bf16_rne(U(rng));
        for (auto &x : B) x = bf16_rne(U(rng));
        run(steps);
        double max_rel = 0, mean_signed = 0, max_chain = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double e = 0; float chain = 0.f;
            for (int t = 0; t < steps; ++t) for (int k = 0; k < 16; ++k) {
                const float p = bf16_to_f(A[(t * 32 + i) * 16 + k]) * bf16_to_f(B[(t * 16 + k) * 32 + j]);  // exact in fp32
                e += (double)p; chain += p;   // fp32 round-to-nearest chain in k order
            }
            const double rel = ((double)D[i * 32 + j] - e) / e;
            max_rel = fmax(max_rel, fabs(rel)); mean_signed += rel / 1024.0;
            max_chain = fmax(max_chain, fabs(((double)chain - e) / e));
        }
        printf("accumulate steps=%d terms=%d: max|rel err| %.3e  mean signed rel err %+.3e  (fp32 RN chain: %.3e; 2^-24 = 5.96e-8)\n",
               steps, steps * 16, max_rel, mean_signed, max_chain);
    }

    // ---- 3. squared distance as a split bilinear form: K slots (2 steps of 16):
    // cand row:  |u|^2 h,m,l | 1,1,1 | per coord: uh,uh,uh,um,um,ul      (6 + 18 = 24, rest 0)
    // query col: 1,1,1 | |v|^2 h,m,l | per coord: -2vh,-2vm,-2vl,-2vh,-2vm,-2vh
    {
        const float r = 0.1118034f, r2 = r * r;
        std::uniform_real_distribution<float> Uo(-4.5f, 4.5f), Uq(-0.8f, 0.8f), Uc(-1.8f, 1.8f);
        double worst = 0, worst_near = 0; long n_near = 0, n_tot = 0;
        for (int rep = 0; rep < 200; ++rep) {
            const float o[3] = {Uo(rng), Uo(rng), Uo(rng)};
            float q[32][3], c[32][3];
            for (int i = 0; i < 32; ++i) for (int d = 0; d < 3; ++d) {
                q[i][d] = o[d] + r * Uq(rng);
                // half of the candidates close to the sphere surface of some query
                c[i][d] = (i & 1) ? o[d] + r * Uc(rng) : q[i][d] + r * (d == (rep + i) % 3 ? 0.99999f + 2e-5f * Uq(rng) : 0.f);
            }
            std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
            auto putA = [&](int cand, int k, uint16_t v) { A[((k >> 4) * 32 + cand) * 16 + (k & 15)] = v; };
            auto putB = [&](int k, int qi, uint16_t v) { B[((k >> 4) * 16 + (k & 15)) * 32 + qi] = v; };
            const uint16_t one = bf16_rne(1.f);
            for (int i = 0; i < 32; ++i) {
                float u[3], v[3];
                for (int d = 0; d < 3; ++d) { u[d] = c[i][d] - o[d]; v[d] = q[i][d] - o[d]; }
                const float uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2], vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                uint16_t s[3], t[3];
                split3(uu, s); split3(vv, t);
                for (int k = 0; k < 3; ++k) { putA(i, k, s[k]); putA(i, 3 + k, one); putB(k, i, one); putB(3 + k, i, t[k]); }
                for (int d = 0; d < 3; ++d) {
                    split3(u[d], s); split3(-2.f * v[d], t);
                    const int kb = 6 + 6 * d;
                    const int ua[6] = {0, 0, 0, 1, 1, 2}, va[6] = {0, 1, 2, 0, 1, 0};
                    for (int k = 0; k < 6; ++k) { putA(i, kb + k, s[ua[k]]); putB(kb + k, i, t[va[k]]); }
                }
            }
            run(2);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {   // D[cand i][query j]
                const double dx = (double)c[i][0] - q[j][0], dy = (double)c[i][1] - q[j][1], dz = (double)c[i][2] - q[j][2];
                const double d2 = dx * dx + dy * dy + dz * dz;
                const double err = fabs((double)D[i * 32 + j] - d2) / r2;
                ++n_tot;
                if (d2 < 9.0 * r2) worst = fmax(worst, err);
                if (fabs(d2 - r2) < 0.01 * r2) { worst_near = fmax(worst_near, err); ++n_near; }
            }
        }
        printf("distance_bilinear_bf16x3: max |d2_mfma - d2| / r^2 = %.3e over pairs within 3r; %.3e over %ld pairs within 1%% of r^2 (of %ld)\n",
               worst, worst_near, n_near, n_tot);
    }
    return 0;
}
