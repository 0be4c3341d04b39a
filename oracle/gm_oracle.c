/*
 * gm_oracle.c -- CPU restatement of the geometric_mapping per-frame path.
 *
 * TEST INFRASTRUCTURE ONLY (see gm_oracle.h).  PARITY UNPINNED: no reference
 * fixture exists; pinned by analytic cases + oracle/oracle_np.py only.
 *
 * Build: see oracle/Makefile.  -ffp-contract=off is REQUIRED: the fp32
 * neighbour predicate and the fp32-faithful sums must round every product.
 */
#include "gm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

static int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* FLANN 1.9.1 L2_Simple<float>: result += diff*diff, dims in order, float. */
static inline float l2_simple(const float *a, const float *b)
{
    float r = 0.0f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

/* cyclic Jacobi, symmetric 3x3, double.  A row-major in; w ascending,
 * V column-major (V[3*c+r]). */
void gmo_eig3(const double *Ain, double *w, double *V)
{
    double a[3][3], v[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            a[i][j] = 0.5 * (Ain[3 * i + j] + Ain[3 * j + i]);
            v[i][j] = (i == j);
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * dg || off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { /* A <- A J */
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A <- J^T A */
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (a[ord[j]][ord[j]] < a[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int c = 0; c < 3; c++) {
        w[c] = a[ord[c]][ord[c]];
        for (int r = 0; r < 3; r++) V[3 * c + r] = v[r][ord[c]];
    }
}

/* ------------------------------------------------------------------ */
/* crop                                                                */
/* ------------------------------------------------------------------ */

int gmo_crop_box(const float *xyz, int n, double bound, int *idx_out)
{
    /* Eigen::Vector4f(-bound, ...) at tunnel_processing.cpp:43-44: double -> float */
    const float lo = (float)(-bound), hi = (float)bound;
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float *p = xyz + 3 * (size_t)i;
        if (!finite3(p)) {
            /* +-Inf fails the box test anyway; NaN would pass it (all compares
             * false) -- dropped here, see header. */
            continue;
        }
        if (p[0] < lo || p[1] < lo || p[2] < lo || p[0] > hi || p[1] > hi || p[2] > hi) continue;
        idx_out[m++] = i;
    }
    return m;
}

/* ------------------------------------------------------------------ */
/* uniform grid (stands in for the FLANN kd-tree: identical neighbour  */
/* SETS, different traversal)                                          */
/* ------------------------------------------------------------------ */

typedef struct {
    float ox, oy, oz, inv_h, h;
    int nx, ny, nz;
    int *start; /* ncell+1 */
    int *order; /* n, ascending point index inside each cell */
    int *cell;  /* n */
} grid_t;

static inline int cell_coord(float x, float o, float inv_h, int n)
{
    int c = (int)floorf((x - o) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

static int grid_build(grid_t *g, const float *xyz, int n, float h_min, long max_cells)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float v = xyz[3 * (size_t)i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    if (n == 0) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
    float h = h_min;
    for (;;) {
        double ex = ((double)mx[0] - mn[0]) / h + 2, ey = ((double)mx[1] - mn[1]) / h + 2,
               ez = ((double)mx[2] - mn[2]) / h + 2;
        if (ex * ey * ez <= (double)max_cells) break;
        h *= 1.26f;
    }
    g->h = h; g->inv_h = 1.0f / h;
    g->ox = mn[0]; g->oy = mn[1]; g->oz = mn[2];
    g->nx = (int)floorf((mx[0] - mn[0]) * g->inv_h) + 1;
    g->ny = (int)floorf((mx[1] - mn[1]) * g->inv_h) + 1;
    g->nz = (int)floorf((mx[2] - mn[2]) * g->inv_h) + 1;
    size_t ncell = (size_t)g->nx * g->ny * g->nz;
    g->start = (int *)calloc(ncell + 1, sizeof(int));
    g->order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    g->cell = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (!g->start || !g->order || !g->cell) return -1;
    for (int i = 0; i < n; i++) {
        const float *p = xyz + 3 * (size_t)i;
        int cx = cell_coord(p[0], g->ox, g->inv_h, g->nx);
        int cy = cell_coord(p[1], g->oy, g->inv_h, g->ny);
        int cz = cell_coord(p[2], g->oz, g->inv_h, g->nz);
        int c = (cz * g->ny + cy) * g->nx + cx;
        g->cell[i] = c;
        g->start[c + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
    int *fill = (int *)malloc(sizeof(int) * ncell);
    if (!fill) return -1;
    memcpy(fill, g->start, sizeof(int) * ncell);
    for (int i = 0; i < n; i++) g->order[fill[g->cell[i]]++] = i;
    free(fill);
    return 0;
}

static void grid_free(grid_t *g) { free(g->start); free(g->order); free(g->cell); }

/* ------------------------------------------------------------------ */
/* PCL eigen33 (smallest eigenpair), fp32 -- pcl/common/impl/eigen.hpp */
/* ------------------------------------------------------------------ */

static void pcl_compute_roots2(float b, float c, float *roots)
{
    roots[0] = 0.0f;
    float d = (float)(b * b - 4.0 * c); /* Scalar(b*b - 4.0*c): double expr, rounded to float */
    if (d < 0.0) d = 0.0f;
    float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}

static void pcl_compute_roots(const float m[3][3], float *roots)
{
    float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2]
             - m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2]
             - m[2][2] * m[0][1] * m[0][1];
    float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2]
             - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    float c2 = m[0][0] + m[1][1] + m[2][2];

    if (fabsf(c0) < FLT_EPSILON) {
        pcl_compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = cosf(theta), sin_theta = sinf(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
        t = roots[1]; roots[1] = roots[2]; roots[2] = t;
        if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0.0f) pcl_compute_roots2(c2, c1, roots);
}

static void pcl_eigen33_smallest(const float C[3][3], float *eigenvalue, float *vec)
{
    float scale = 0.0f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (fabsf(C[i][j]) > scale) scale = fabsf(C[i][j]);
    if (scale <= FLT_MIN) scale = 1.0f;
    float s[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) s[i][j] = C[i][j] / scale;
    float roots[3];
    pcl_compute_roots(s, roots);
    *eigenvalue = roots[0] * scale;
    s[0][0] -= roots[0]; s[1][1] -= roots[0]; s[2][2] -= roots[0];
    float v1[3], v2[3], v3[3];
#define CROSS(o, a, b) do { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; } while (0)
    CROSS(v1, s[0], s[1]);
    CROSS(v2, s[0], s[2]);
    CROSS(v3, s[1], s[2]);
#undef CROSS
    float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const float *pick; float len;
    if (l1 >= l2 && l1 >= l3) { pick = v1; len = l1; }
    else if (l2 >= l1 && l2 >= l3) { pick = v2; len = l2; }
    else { pick = v3; len = l3; }
    float nrm = sqrtf(len);
    vec[0] = pick[0] / nrm; vec[1] = pick[1] / nrm; vec[2] = pick[2] / nrm;
}

/* ------------------------------------------------------------------ */
/* normals                                                             */
/* ------------------------------------------------------------------ */

typedef struct { float d2; int j; } nb_t;

static int nb_cmp(const void *a, const void *b)
{
    /* FLANN DistanceIndex::operator< : by distance, then by index (sorted=true
     * is pcl::search::KdTree's default, so neighbours arrive in this order) */
    const nb_t *x = (const nb_t *)a, *y = (const nb_t *)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->j > y->j) - (x->j < y->j);
}

static void normal_from_neighbours_f64(const float *xyz, const float *q, const nb_t *nb, int m, float *out)
{
    /* double sums of offsets from the query point: the mathematical value of
     * PCL's covariance (centroid.hpp computeMeanAndCovarianceMatrix) */
    double s[9] = {0};
    for (int t = 0; t < m; t++) {
        const float *p = xyz + 3 * (size_t)nb[t].j;
        double dx = (double)p[0] - q[0], dy = (double)p[1] - q[1], dz = (double)p[2] - q[2];
        s[0] += dx * dx; s[1] += dx * dy; s[2] += dx * dz;
        s[3] += dy * dy; s[4] += dy * dz; s[5] += dz * dz;
        s[6] += dx; s[7] += dy; s[8] += dz;
    }
    for (int k = 0; k < 9; k++) s[k] /= (double)m;
    double C[9];
    C[0] = s[0] - s[6] * s[6]; C[1] = s[1] - s[6] * s[7]; C[2] = s[2] - s[6] * s[8];
    C[4] = s[3] - s[7] * s[7]; C[5] = s[4] - s[7] * s[8]; C[8] = s[5] - s[8] * s[8];
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    if (C[0] == 0.0 && C[1] == 0.0 && C[2] == 0.0 && C[4] == 0.0 && C[5] == 0.0 && C[8] == 0.0) {
        /* all neighbours coincide: pcl::eigen33 normalises a zero cross product
         * (0/0) -> NaN normal, and the point is then removed as NaN */
        out[0] = out[1] = out[2] = out[3] = NAN;
        return;
    }
    double w[3], V[9];
    gmo_eig3(C, w, V);
    double nx = V[0], ny = V[1], nz = V[2];
    double tr = C[0] + C[4] + C[8];
    double curv = (tr != 0.0) ? fabs(w[0] / tr) : 0.0;
    /* flipNormalTowardsViewpoint(point, 0,0,0, nx,ny,nz) */
    double ct = (0.0 - q[0]) * nx + (0.0 - q[1]) * ny + (0.0 - q[2]) * nz;
    if (ct < 0) { nx = -nx; ny = -ny; nz = -nz; }
    out[0] = (float)nx; out[1] = (float)ny; out[2] = (float)nz; out[3] = (float)curv;
}

static void normal_from_neighbours_f32(const float *xyz, const float *q, nb_t *nb, int m, float *out, int shifted)
{
    qsort(nb, (size_t)m, sizeof(nb_t), nb_cmp);
    /* PCL 1.8 centroid.hpp: un-shifted single pass, 9 float accumulators.
     * PCL >= 1.10 centroid.hpp (shifted != 0): the same pass over p - K with K = the neighbourhood's first point
     * ("shift the data to avoid catastrophic cancellation"); the covariance does not depend on K, only its rounding. */
    float a[9] = {0};
    float K[3] = {0.0f, 0.0f, 0.0f};
    if (shifted) { const float *p0 = xyz + 3 * (size_t)nb[0].j; K[0] = p0[0]; K[1] = p0[1]; K[2] = p0[2]; }
    for (int t = 0; t < m; t++) {
        const float *pp = xyz + 3 * (size_t)nb[t].j;
        const float p[3] = {pp[0] - K[0], pp[1] - K[1], pp[2] - K[2]};
        a[0] += p[0] * p[0]; a[1] += p[0] * p[1]; a[2] += p[0] * p[2];
        a[3] += p[1] * p[1]; a[4] += p[1] * p[2]; a[5] += p[2] * p[2];
        a[6] += p[0]; a[7] += p[1]; a[8] += p[2];
    }
    for (int k = 0; k < 9; k++) a[k] /= (float)m;
    float C[3][3];
    C[0][0] = a[0] - a[6] * a[6]; C[0][1] = a[1] - a[6] * a[7]; C[0][2] = a[2] - a[6] * a[8];
    C[1][1] = a[3] - a[7] * a[7]; C[1][2] = a[4] - a[7] * a[8]; C[2][2] = a[5] - a[8] * a[8];
    C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
    float ev, v[3];
    pcl_eigen33_smallest(C, &ev, v);
    float eig_sum = C[0][0] + C[1][1] + C[2][2];
    float curv = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
    float vx = 0.0f - q[0], vy = 0.0f - q[1], vz = 0.0f - q[2];
    float ct = vx * v[0] + vy * v[1] + vz * v[2];
    if (ct < 0) { v[0] *= -1; v[1] *= -1; v[2] *= -1; }
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = curv;
}

int gmo_normals(const float *xyz, int n, double radius, int mode, int nthreads,
                float *normals_out, int *counts_out)
{
    if (n <= 0) return 0;
    /* KdTreeFLANN::radiusSearch passes static_cast<float>(radius*radius) */
    const float r2 = (float)(radius * radius);
    grid_t g;
    if (grid_build(&g, xyz, n, (float)radius * 1.001f, 1L << 26) != 0) return -1;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        int cap = 1024;
        nb_t *nb = (nb_t *)malloc(sizeof(nb_t) * (size_t)cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int i = 0; i < n; i++) {
            const float *q = xyz + 3 * (size_t)i;
            int cx = cell_coord(q[0], g.ox, g.inv_h, g.nx);
            int cy = cell_coord(q[1], g.oy, g.inv_h, g.ny);
            int cz = cell_coord(q[2], g.oz, g.inv_h, g.nz);
            int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < g.nx - 1 ? cx + 1 : g.nx - 1;
            int m = 0;
            for (int z = cz - 1; z <= cz + 1; z++) {
                if (z < 0 || z >= g.nz) continue;
                for (int y = cy - 1; y <= cy + 1; y++) {
                    if (y < 0 || y >= g.ny) continue;
                    int row = (z * g.ny + y) * g.nx;
                    int b = g.start[row + x0], e = g.start[row + x1 + 1];
                    for (int s = b; s < e; s++) {
                        int j = g.order[s];
                        float d2 = l2_simple(q, xyz + 3 * (size_t)j);
                        if (d2 < r2) { /* FLANN RadiusResultSet::addPoint: dist < radius */
                            if (m == cap) {
                                cap *= 2;
                                nb = (nb_t *)realloc(nb, sizeof(nb_t) * (size_t)cap);
                                if (!nb) { fail = 1; break; }
                            }
                            nb[m].d2 = d2; nb[m].j = j; m++;
                        }
                    }
                }
            }
            float *o = normals_out + 4 * (size_t)i;
            if (counts_out) counts_out[i] = m;
            if (m < 3 || fail) { /* NormalEstimation::computePointNormal: indices.size()<3 */
                o[0] = o[1] = o[2] = o[3] = NAN;
            } else if (mode == GMO_F32_FAITHFUL || mode == GMO_F32_SHIFTED) {
                normal_from_neighbours_f32(xyz, q, nb, m, o, mode == GMO_F32_SHIFTED);
            } else {
                normal_from_neighbours_f64(xyz, q, nb, m, o);
            }
        }
        free(nb);
    }
    grid_free(&g);
    return fail ? -1 : n;
}

int gmo_finite_normals(const float *normals, int n, int *idx_out)
{
    int m = 0;
    for (int i = 0; i < n; i++)
        if (finite3(normals + 4 * (size_t)i)) idx_out[m++] = i;
    return m;
}

/* ------------------------------------------------------------------ */
/* voxel grid                                                          */
/* ------------------------------------------------------------------ */

typedef struct { uint32_t key; int idx; } vk_t;

static int vk_cmp(const void *a, const void *b)
{
    const vk_t *x = (const vk_t *)a, *y = (const vk_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    /* PCL's std::sort compares the key only (order inside a voxel is
     * unspecified there); ascending index makes the oracle deterministic */
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int gmo_voxel_grid(const float *xyz, int n, double leaf, int mode,
                   float *out_xyz, int32_t *out_key, int32_t *out_count, int *passthrough)
{
    if (passthrough) *passthrough = 0;
    if (n <= 0) return 0;
    /* setLeafSize(float,float,float): double -> float at the call, inverse in float */
    const float leaf_f = (float)leaf;
    const float inv = 1.0f / leaf_f;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float v = xyz[3 * (size_t)i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {
        if (passthrough) *passthrough = 1;
        memcpy(out_xyz, xyz, sizeof(float) * 3 * (size_t)n);
        if (out_key) for (int i = 0; i < n; i++) out_key[i] = i;
        if (out_count) for (int i = 0; i < n; i++) out_count[i] = 1;
        return n;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int k = 0; k < 3; k++) {
        min_b[k] = (int)floorf(mn[k] * inv);
        max_b[k] = (int)floorf(mx[k] * inv);
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    vk_t *v = (vk_t *)malloc(sizeof(vk_t) * (size_t)n);
    if (!v) return -1;
    for (int i = 0; i < n; i++) {
        const float *p = xyz + 3 * (size_t)i;
        int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        v[i].key = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
        v[i].idx = i;
    }
    qsort(v, (size_t)n, sizeof(vk_t), vk_cmp);
    int V = 0, a = 0;
    while (a < n) {
        int b = a + 1;
        while (b < n && v[b].key == v[a].key) b++;
        if (mode == GMO_F32_FAITHFUL) {
            float sx = 0, sy = 0, sz = 0; /* CentroidPoint / AccumulatorXYZ: float sums */
            for (int t = a; t < b; t++) {
                const float *p = xyz + 3 * (size_t)v[t].idx;
                sx += p[0]; sy += p[1]; sz += p[2];
            }
            float c = (float)(b - a);
            out_xyz[3 * (size_t)V + 0] = sx / c;
            out_xyz[3 * (size_t)V + 1] = sy / c;
            out_xyz[3 * (size_t)V + 2] = sz / c;
        } else {
            double sx = 0, sy = 0, sz = 0;
            for (int t = a; t < b; t++) {
                const float *p = xyz + 3 * (size_t)v[t].idx;
                sx += p[0]; sy += p[1]; sz += p[2];
            }
            double c = (double)(b - a);
            out_xyz[3 * (size_t)V + 0] = (float)(sx / c);
            out_xyz[3 * (size_t)V + 1] = (float)(sy / c);
            out_xyz[3 * (size_t)V + 2] = (float)(sz / c);
        }
        if (out_key) out_key[V] = (int32_t)v[a].key;
        if (out_count) out_count[V] = b - a;
        V++;
        a = b;
    }
    free(v);
    return V;
}

/* ------------------------------------------------------------------ */
/* local frame                                                         */
/* ------------------------------------------------------------------ */

/* Eigen 3.3 SelfAdjointEigenSolver<MatrixXf>::compute restated in float:
 * scale by max|a_ij|, Householder tridiagonalisation, implicit symmetric QR
 * with Wilkinson shift (Golub & Van Loan alg. 8.3.2/8.3.3), ascending sort.
 * Not claimed bit-identical to Eigen's blocked kernels. */
static void givens_f(float p, float q, float *c, float *s)
{
    if (q == 0.0f) { *c = p < 0 ? -1.0f : 1.0f; *s = 0.0f; }
    else if (p == 0.0f) { *c = 0.0f; *s = q < 0 ? 1.0f : -1.0f; }
    else if (fabsf(p) > fabsf(q)) {
        float t = q / p, u = sqrtf(1.0f + t * t);
        if (p < 0) u = -u;
        *c = 1.0f / u; *s = -t * (*c);
    } else {
        float t = p / q, u = sqrtf(1.0f + t * t);
        if (q < 0) u = -u;
        *s = -1.0f / u; *c = -t * (*s);
    }
}

static void eigen_selfadjoint3_f32(const float Ain[3][3], float *evals, float *evecs_cm)
{
    float A[3][3];
    float scale = 0.0f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j <= i; j++)
            if (fabsf(Ain[i][j]) > scale) scale = fabsf(Ain[i][j]);
    if (scale == 0.0f) scale = 1.0f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[i][j] = Ain[i][j] / scale;
    /* Householder on column 0: x = (a10, a20) */
    float Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    float diag[3], sub[2];
    float c0 = A[1][0], tail = A[2][0];
    float tailsq = tail * tail;
    if (tailsq <= FLT_MIN) {
        diag[0] = A[0][0]; diag[1] = A[1][1]; diag[2] = A[2][2];
        sub[0] = c0; sub[1] = A[2][1];
    } else {
        float beta = sqrtf(c0 * c0 + tailsq);
        if (c0 >= 0) beta = -beta;
        float ess = tail / (c0 - beta);
        float tau = (beta - c0) / beta;
        float v[2] = {1.0f, ess};
        float B[2][2] = {{A[1][1], A[2][1]}, {A[2][1], A[2][2]}};
        float p[2] = {tau * (B[0][0] * v[0] + B[0][1] * v[1]), tau * (B[1][0] * v[0] + B[1][1] * v[1])};
        float K = -0.5f * tau * (p[0] * v[0] + p[1] * v[1]);
        float w[2] = {p[0] + K * v[0], p[1] + K * v[1]};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) B[i][j] -= v[i] * w[j] + w[i] * v[j];
        diag[0] = A[0][0]; diag[1] = B[0][0]; diag[2] = B[1][1];
        sub[0] = beta; sub[1] = B[1][0];
        /* Q = I - tau v v^T on rows/cols 1..2 */
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) Q[1 + i][1 + j] = (i == j ? 1.0f : 0.0f) - tau * v[i] * v[j];
    }
    const int n = 3;
    int end = n - 1, start = 0, iter = 0;
    const float prec = 2.0f * FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; i++)
            if (fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * prec || fabsf(sub[i]) <= FLT_MIN)
                sub[i] = 0.0f;
        while (end > 0 && sub[end - 1] == 0.0f) end--;
        if (end <= 0) break;
        if (++iter > 30 * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.0f) start--;
        float td = (diag[end - 1] - diag[end]) * 0.5f;
        float e = sub[end - 1];
        float mu = diag[end];
        if (td == 0.0f) mu -= fabsf(e);
        else {
            float e2 = e * e, h = hypotf(td, e);
            if (e2 == 0.0f) mu -= (e / (td + (td > 0 ? 1.0f : -1.0f))) * (e / h);
            else mu -= e2 / (td + (td > 0 ? h : -h));
        }
        float x = diag[start] - mu, z = sub[start];
        for (int k = start; k < end; k++) {
            float c, s;
            givens_f(x, z, &c, &s);
            float sdk = s * diag[k] + c * sub[k];
            float dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
            for (int r = 0; r < 3; r++) { /* Q <- Q G */
                float xi = Q[r][k], yi = Q[r][k + 1];
                Q[r][k] = c * xi - s * yi;
                Q[r][k + 1] = s * xi + c * yi;
            }
        }
    }
    for (int i = 0; i < n - 1; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (diag[j] < diag[k]) k = j;
        if (k != i) {
            float t = diag[i]; diag[i] = diag[k]; diag[k] = t;
            for (int r = 0; r < 3; r++) { float u = Q[r][i]; Q[r][i] = Q[r][k]; Q[r][k] = u; }
        }
    }
    for (int c = 0; c < 3; c++) {
        evals[c] = diag[c] * scale;
        for (int r = 0; r < 3; r++) evecs_cm[3 * c + r] = Q[r][c];
    }
}

void gmo_local_frame(const float *normals, int n, double wf, int mode,
                     double *M_out, float *evals, float *evecs)
{
    if (mode == GMO_F32_FAITHFUL || mode == GMO_F32_SHIFTED) {
        float M[3][3] = {{0}};
        for (int i = 0; i < n; i++) {
            const float *nr = normals + 4 * (size_t)i;
            /* tunnel_processing.cpp:106: float + (double/double), pow in double,
             * exp in double, stored into a MatrixXf element */
            float w = (float)exp(pow(nr[3] + .001 / wf, 2));
            /* :119 weights*normals: row i = w_i * n_i (+ exact zeros) */
            float a = w * nr[0], b = w * nr[1], c = w * nr[2];
            /* :124 transpose product, float accumulation */
            M[0][0] += a * a; M[0][1] += a * b; M[0][2] += a * c;
            M[1][1] += b * b; M[1][2] += b * c; M[2][2] += c * c;
        }
        M[1][0] = M[0][1]; M[2][0] = M[0][2]; M[2][1] = M[1][2];
        if (M_out) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_out[3 * i + j] = M[i][j];
        eigen_selfadjoint3_f32(M, evals, evecs);
    } else {
        double M[9] = {0};
        for (int i = 0; i < n; i++) {
            const float *nr = normals + 4 * (size_t)i;
            double t = (double)nr[3] + .001 / wf;
            double w = exp(t * t);
            double a = w * nr[0], b = w * nr[1], c = w * nr[2];
            M[0] += a * a; M[1] += a * b; M[2] += a * c;
            M[4] += b * b; M[5] += b * c; M[8] += c * c;
        }
        M[3] = M[1]; M[6] = M[2]; M[7] = M[5];
        if (M_out) memcpy(M_out, M, sizeof(M));
        double w[3], V[9];
        gmo_eig3(M, w, V);
        for (int k = 0; k < 3; k++) evals[k] = (float)w[k];
        for (int k = 0; k < 9; k++) evecs[k] = (float)V[k];
    }
}

/* ------------------------------------------------------------------ */
/* 1-NN                                                                */
/* ------------------------------------------------------------------ */

void gmo_nearest(const float *xyz, int n, const float *queries, int nq, int *idx_out)
{
    if (n <= 0) { for (int i = 0; i < nq; i++) idx_out[i] = -1; return; }
    grid_t g;
    /* ~4 points per cell on average, capped */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float v = xyz[3 * (size_t)i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    double vol = ((double)mx[0] - mn[0] + 1e-3) * ((double)mx[1] - mn[1] + 1e-3) * ((double)mx[2] - mn[2] + 1e-3);
    float h = (float)cbrt(vol * 4.0 / (double)n);
    if (!(h > 1e-6f)) h = 1e-3f;
    grid_build(&g, xyz, n, h, 1L << 24);
    for (int qi = 0; qi < nq; qi++) {
        const float *q = queries + 3 * (size_t)qi;
        int cx = (int)floorf((q[0] - g.ox) * g.inv_h);
        int cy = (int)floorf((q[1] - g.oy) * g.inv_h);
        int cz = (int)floorf((q[2] - g.oz) * g.inv_h);
        float best = FLT_MAX; int bi = -1;
        int maxring = g.nx + g.ny + g.nz + abs(cx) + abs(cy) + abs(cz);
        for (int ring = 0; ring <= maxring; ring++) {
            for (int z = cz - ring; z <= cz + ring; z++) {
                if (z < 0 || z >= g.nz) continue;
                for (int y = cy - ring; y <= cy + ring; y++) {
                    if (y < 0 || y >= g.ny) continue;
                    int shell = (abs(z - cz) == ring) || (abs(y - cy) == ring);
                    for (int x = cx - ring; x <= cx + ring; x += (shell ? 1 : (ring > 0 ? 2 * ring : 1))) {
                        if (x < 0 || x >= g.nx) continue;
                        int c = (z * g.ny + y) * g.nx + x;
                        for (int s = g.start[c]; s < g.start[c + 1]; s++) {
                            int j = g.order[s];
                            float d2 = l2_simple(q, xyz + 3 * (size_t)j);
                            if (d2 < best || (d2 == best && j < bi)) { best = d2; bi = j; }
                        }
                    }
                }
            }
            if (bi >= 0) {
                float reach = (float)ring * g.h; /* unexamined cells are at least this far */
                if (best <= reach * reach) break;
            }
        }
        idx_out[qi] = bi;
    }
    grid_free(&g);
}

/* ------------------------------------------------------------------ */
/* whole frame                                                         */
/* ------------------------------------------------------------------ */

int gmo_process_frame(const float *xyz, int n, double bound, double radius, double leaf,
                      double wf, int mode, int nthreads,
                      float *out_xyz, float *out_normals, float *out_vox,
                      gmo_frame_result *res)
{
    memset(res, 0, sizeof(*res));
    res->n_in = n;
    size_t cap = (size_t)(n > 0 ? n : 1);
    int *idx = (int *)malloc(sizeof(int) * cap);
    float *c1 = (float *)malloc(sizeof(float) * 3 * cap);
    float *nr = (float *)malloc(sizeof(float) * 4 * cap);
    float *c2 = (float *)malloc(sizeof(float) * 3 * cap);
    float *n2 = (float *)malloc(sizeof(float) * 4 * cap);
    float *vx = (float *)malloc(sizeof(float) * 3 * cap);
    if (!idx || !c1 || !nr || !c2 || !n2 || !vx) return -1;
    /* geometric_mapping.cpp:57 */
    int n1 = gmo_crop_box(xyz, n, bound, idx);
    for (int i = 0; i < n1; i++) memcpy(c1 + 3 * (size_t)i, xyz + 3 * (size_t)idx[i], 12);
    res->n_cropped = n1;
    /* :63 getNormals (normals, NaN removal, in-place compaction of the cloud) */
    gmo_normals(c1, n1, radius, mode, nthreads, nr, NULL);
    int nv = gmo_finite_normals(nr, n1, idx);
    for (int i = 0; i < nv; i++) {
        memcpy(c2 + 3 * (size_t)i, c1 + 3 * (size_t)idx[i], 12);
        memcpy(n2 + 4 * (size_t)i, nr + 4 * (size_t)idx[i], 16);
    }
    res->n_valid = nv;
    /* :70-75 rvizNormals -> VoxelGrid on the compacted cloud (always runs) */
    int pt = 0;
    res->n_voxels = gmo_voxel_grid(c2, nv, leaf, mode, vx, NULL, NULL, &pt);
    /* :82-88 getLocalFrame(cloudChopped->points.size(), ...) */
    gmo_local_frame(n2, nv, wf, mode, res->M, res->evals, res->evecs);
    if (out_xyz) memcpy(out_xyz, c2, sizeof(float) * 3 * (size_t)nv);
    if (out_normals) memcpy(out_normals, n2, sizeof(float) * 4 * (size_t)nv);
    if (out_vox) memcpy(out_vox, vx, sizeof(float) * 3 * (size_t)res->n_voxels);
    free(idx); free(c1); free(nr); free(c2); free(n2); free(vx);
    return 0;
}
