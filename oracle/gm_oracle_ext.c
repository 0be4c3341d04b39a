/*
 * gm_oracle_ext.c -- CPU restatement of the BUILD-DEFINED extensions
 * (batched RANSAC plane / cylinder scoring, per-segment moments, refits).
 *
 * TEST INFRASTRUCTURE ONLY (see gm_oracle.h).
 *
 * The reference has no counterpart: getCylinder is an empty commented stub
 * (/root/reference src/tunnel_processing.cpp:149-154).  Model conventions
 * follow PCL's SampleConsensusModelPlane / SampleConsensusModelCylinder (the
 * library the reference already depends on); everything here is pinned by
 * analytic truth of the synthetic generator only.
 *
 * Scoring arithmetic is spelled with explicit fmaf() so that the device
 * kernels (which use __fmaf_rn in the same order) produce bit-identical
 * inlier decisions.
 */
#include "gm_oracle.h"

#include <math.h>
#include <string.h>

uint64_t gmo_mix64(uint64_t x)
{
    /* splitmix64 finaliser */
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* t-th draw of hypothesis h: index in [0,n) */
static uint32_t draw_index(uint64_t seed, uint32_t h, uint32_t t, uint32_t n)
{
    uint64_t u = gmo_mix64(seed ^ gmo_mix64(((uint64_t)h << 8) | t));
    return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
}

static void set_nan(float *p, int k) { for (int i = 0; i < k; i++) p[i] = NAN; }

#define GMO_MAX_DRAWS 64u

/* next index of hypothesis h that lies in the wanted segment (labels==NULL: any)
 * and differs from a and b; advances *t; returns 0xFFFFFFFF when the draw budget
 * is spent */
static uint32_t next_sample(uint64_t seed, uint32_t h, uint32_t *t, uint32_t n, const uint8_t *labels, int want,
                            uint32_t a, uint32_t b)
{
    while (*t < GMO_MAX_DRAWS) {
        uint32_t i = draw_index(seed, h, (*t)++, n);
        if (i == a || i == b) continue;
        if (labels && labels[i] != want) continue;
        return i;
    }
    return 0xFFFFFFFFu;
}

void gmo_plane_hypotheses(const float *xyz, int n, const uint8_t *labels, int want, uint64_t seed, int H,
                          float *hyp4)
{
    for (int h = 0; h < H; h++) {
        float *o = hyp4 + 4 * (size_t)h;
        if (n < 3) { set_nan(o, 4); continue; }
        uint32_t t = 0;
        const uint32_t none = 0xFFFFFFFFu;
        uint32_t i0 = next_sample(seed, (uint32_t)h, &t, (uint32_t)n, labels, want, none, none);
        uint32_t i1 = i0 == none ? none : next_sample(seed, (uint32_t)h, &t, (uint32_t)n, labels, want, i0, none);
        uint32_t i2 = i1 == none ? none : next_sample(seed, (uint32_t)h, &t, (uint32_t)n, labels, want, i0, i1);
        if (i2 == none) { set_nan(o, 4); continue; }
        const float *p0 = xyz + 3 * (size_t)i0, *p1 = xyz + 3 * (size_t)i1, *p2 = xyz + 3 * (size_t)i2;
        double ax = (double)p1[0] - p0[0], ay = (double)p1[1] - p0[1], az = (double)p1[2] - p0[2];
        double bx = (double)p2[0] - p0[0], by = (double)p2[1] - p0[1], bz = (double)p2[2] - p0[2];
        double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
        double len = sqrt(nx * nx + ny * ny + nz * nz);
        if (!(len > 1e-12)) { set_nan(o, 4); continue; }
        nx /= len; ny /= len; nz /= len;
        double d = -(nx * p0[0] + ny * p0[1] + nz * p0[2]);
        o[0] = (float)nx; o[1] = (float)ny; o[2] = (float)nz; o[3] = (float)d;
    }
}

void gmo_cylinder_hypotheses(const float *xyz, const float *normals, int n, const uint8_t *labels, int want,
                             uint64_t seed, int H, float *hyp7)
{
    for (int h = 0; h < H; h++) {
        float *o = hyp7 + 7 * (size_t)h;
        if (n < 2) { set_nan(o, 7); continue; }
        uint32_t t = 0;
        const uint32_t none = 0xFFFFFFFFu;
        uint32_t i0 = next_sample(seed, (uint32_t)h, &t, (uint32_t)n, labels, want, none, none);
        uint32_t i1 = i0 == none ? none : next_sample(seed, (uint32_t)h, &t, (uint32_t)n, labels, want, i0, none);
        if (i1 == none) { set_nan(o, 7); continue; }
        const float *p1 = xyz + 3 * (size_t)i0, *p2 = xyz + 3 * (size_t)i1;
        const float *n1 = normals + 4 * (size_t)i0, *n2 = normals + 4 * (size_t)i1;
        /* closest points of the two normal lines (p1+n1)+s*n1 and p2+t*n2 */
        double w[3] = {(double)n1[0] + p1[0] - p2[0], (double)n1[1] + p1[1] - p2[1], (double)n1[2] + p1[2] - p2[2]};
        double a = (double)n1[0] * n1[0] + (double)n1[1] * n1[1] + (double)n1[2] * n1[2];
        double b = (double)n1[0] * n2[0] + (double)n1[1] * n2[1] + (double)n1[2] * n2[2];
        double c = (double)n2[0] * n2[0] + (double)n2[1] * n2[1] + (double)n2[2] * n2[2];
        double d = n1[0] * w[0] + n1[1] * w[1] + n1[2] * w[2];
        double e = n2[0] * w[0] + n2[1] * w[1] + n2[2] * w[2];
        double den = a * c - b * b, sc, tc;
        if (den < 1e-8) { sc = 0.0; tc = (b > c ? d / b : e / c); }
        else { sc = (b * e - c * d) / den; tc = (a * e - b * d) / den; }
        double lp[3], ld[3];
        for (int k = 0; k < 3; k++) lp[k] = (double)p1[k] + n1[k] + sc * n1[k];
        for (int k = 0; k < 3; k++) ld[k] = (double)p2[k] + tc * n2[k] - lp[k];
        double len = sqrt(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
        if (!(len > 1e-9) || !isfinite(len)) { set_nan(o, 7); continue; }
        for (int k = 0; k < 3; k++) ld[k] /= len;
        double v[3] = {(double)p1[0] - lp[0], (double)p1[1] - lp[1], (double)p1[2] - lp[2]};
        double tt = v[0] * ld[0] + v[1] * ld[1] + v[2] * ld[2];
        double q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - tt * tt;
        double r = sqrt(q > 0 ? q : 0);
        o[0] = (float)lp[0]; o[1] = (float)lp[1]; o[2] = (float)lp[2];
        o[3] = (float)ld[0]; o[4] = (float)ld[1]; o[5] = (float)ld[2];
        o[6] = (float)r;
    }
}

static inline int plane_inlier(const float *p, const float *hy, float tau)
{
    float d = fmaf(hy[0], p[0], fmaf(hy[1], p[1], fmaf(hy[2], p[2], hy[3])));
    return fabsf(d) < tau; /* NaN hypothesis -> false */
}

static inline void cyl_band(const float *hy, double tau, float *lo2, float *hi2)
{
    double r = hy[6];
    double lo = r - tau, hi = r + tau;
    *lo2 = lo > 0 ? (float)(lo * lo) : -1.0f;
    *hi2 = (float)(hi * hi);
}

static inline int cyl_inlier(const float *p, const float *hy, float lo2, float hi2)
{
    float vx = p[0] - hy[0], vy = p[1] - hy[1], vz = p[2] - hy[2];
    float t = fmaf(vx, hy[3], fmaf(vy, hy[4], vz * hy[5]));
    float vv = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
    float q = fmaf(-t, t, vv);
    return q > lo2 && q < hi2;
}

void gmo_score_planes(const float *xyz, int n, const uint8_t *mask, int want,
                      const float *hyp4, int H, double tau, int32_t *counts)
{
    const float tf = (float)tau;
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++) {
        const float *hy = hyp4 + 4 * (size_t)h;
        int32_t c = 0;
        for (int i = 0; i < n; i++) {
            if (mask && mask[i] != want) continue;
            c += plane_inlier(xyz + 3 * (size_t)i, hy, tf);
        }
        counts[h] = c;
    }
}

void gmo_score_cylinders(const float *xyz, int n, const uint8_t *mask, int want,
                         const float *hyp7, int H, double tau, int32_t *counts)
{
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; h++) {
        const float *hy = hyp7 + 7 * (size_t)h;
        float lo2, hi2;
        cyl_band(hy, tau, &lo2, &hi2);
        int32_t c = 0;
        for (int i = 0; i < n; i++) {
            if (mask && mask[i] != want) continue;
            c += cyl_inlier(xyz + 3 * (size_t)i, hy, lo2, hi2);
        }
        counts[h] = c;
    }
}

int gmo_label_plane(const float *xyz, int n, uint8_t *labels, int want, int label,
                    const float *hyp4, double tau)
{
    const float tf = (float)tau;
    int c = 0;
    for (int i = 0; i < n; i++)
        if (labels[i] == want && plane_inlier(xyz + 3 * (size_t)i, hyp4, tf)) { labels[i] = (uint8_t)label; c++; }
    return c;
}

int gmo_label_cylinder(const float *xyz, int n, uint8_t *labels, int want, int label,
                       const float *hyp7, double tau)
{
    float lo2, hi2;
    cyl_band(hyp7, tau, &lo2, &hi2);
    int c = 0;
    for (int i = 0; i < n; i++)
        if (labels[i] == want && cyl_inlier(xyz + 3 * (size_t)i, hyp7, lo2, hi2)) { labels[i] = (uint8_t)label; c++; }
    return c;
}

void gmo_segment_moments(const float *xyz, const float *normals, int n, const uint8_t *labels,
                         int label, double *m)
{
    memset(m, 0, sizeof(double) * 16);
    for (int i = 0; i < n; i++) {
        if (labels && labels[i] != label) continue;
        const float *p = xyz + 3 * (size_t)i;
        double x = p[0], y = p[1], z = p[2];
        m[0] += 1.0;
        m[1] += x; m[2] += y; m[3] += z;
        m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
        if (normals) {
            const float *q = normals + 4 * (size_t)i;
            double a = q[0], b = q[1], c = q[2];
            m[10] += a * a; m[11] += a * b; m[12] += a * c; m[13] += b * b; m[14] += b * c; m[15] += c * c;
        }
    }
}

void gmo_refit_plane(const double *m, double *plane4)
{
    double n = m[0];
    if (!(n >= 3)) { plane4[0] = plane4[1] = plane4[2] = plane4[3] = NAN; return; }
    double cx = m[1] / n, cy = m[2] / n, cz = m[3] / n;
    double C[9];
    C[0] = m[4] / n - cx * cx; C[1] = m[5] / n - cx * cy; C[2] = m[6] / n - cx * cz;
    C[4] = m[7] / n - cy * cy; C[5] = m[8] / n - cy * cz; C[8] = m[9] / n - cz * cz;
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    double w[3], V[9];
    gmo_eig3(C, w, V);
    plane4[0] = V[0]; plane4[1] = V[1]; plane4[2] = V[2];
    plane4[3] = -(V[0] * cx + V[1] * cy + V[2] * cz);
}

void gmo_refit_axis(const double *m, double *axis3)
{
    double S[9] = {m[10], m[11], m[12], m[11], m[13], m[14], m[12], m[14], m[15]};
    double w[3], V[9];
    gmo_eig3(S, w, V);
    axis3[0] = V[0]; axis3[1] = V[1]; axis3[2] = V[2];
}
