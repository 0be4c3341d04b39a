// gm_ext.hip -- C ABI entry points of the build-defined extensions (RANSAC
// scoring, segment moments, 1-NN, labels, compressed map).  Host logic only;
// kernels are in k_ransac.hip / k_nearest.hip.
#include <math.h>
#include <string.h>

#include <vector>

#include "gm_internal.hpp"

using namespace gm;

namespace {

#define GMX_HIP(ctx, call)                                                           \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);         \
            return (e__ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_DEVICE;        \
        }                                                                            \
    } while (0)

// rows of 3 floats -> device float4 array `dst` (through the slot's pinned staging)
gm_status upload_xyz(gm_ctx *ctx, Slot &sl, const float *xyz, uint32_t n, float4 *dst)
{
    if (!n) return GM_OK;
    float4 *stage = (float4 *)sl.h_raw;
    for (uint32_t i = 0; i < n; ++i) stage[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f);
    GMX_HIP(ctx, hipMemcpyAsync(dst, stage, (size_t)n * 16, hipMemcpyHostToDevice, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));  // staging is reused by the caller right away
    return GM_OK;
}

gm_status prepare(gm_ctx *ctx, Slot *&sl, uint32_t n, uint32_t H)
{
    gm_status st = gm_begin_stage(ctx, sl);
    if (st != GM_OK) return st;
    st = gm_ensure_capacity(ctx, *sl, n, (size_t)n * 16, true);
    if (st != GM_OK) return st;
    return gm_ensure_ext(ctx, *sl, H ? H : 1);
}

gm_status upload_labels(gm_ctx *ctx, Slot &sl, const uint8_t *labels, uint32_t n)
{
    if (labels && n) GMX_HIP(ctx, hipMemcpyAsync(sl.labels, labels, n, hipMemcpyHostToDevice, sl.stream));
    return GM_OK;
}

gm_status score(gm_ctx *ctx, int model, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                const float *hyp, uint32_t H, double tau, int32_t *counts)
{
    Slot *slp;
    if (!ctx) return GM_ERR_INVALID_ARG;
    if ((n && !xyz) || !hyp || !counts) return gm_fail(ctx, GM_ERR_INVALID_ARG, "NULL argument");
    gm_status st = prepare(ctx, slp, n, H);
    if (st != GM_OK) return st;
    Slot &sl = *slp;
    st = upload_xyz(ctx, sl, xyz, n, sl.valid4);
    if (st != GM_OK) return st;
    st = upload_labels(ctx, sl, labels, n);
    if (st != GM_OK) return st;
    // caller rows (4 or 7 floats) -> internal rows of 8
    const int w = model == 0 ? 4 : 7;
    float *stage = (float *)sl.h_raw;  // capacity >= n*16 bytes; H*32 must fit too
    if ((size_t)H * 32 > sl.raw_cap) {
        st = gm_ensure_capacity(ctx, sl, n, (size_t)H * 32, true);
        if (st != GM_OK) return st;
        stage = (float *)sl.h_raw;
    }
    memset(stage, 0, (size_t)H * 32);
    for (uint32_t h = 0; h < H; ++h)
        for (int k = 0; k < w; ++k) stage[8 * (size_t)h + k] = hyp[(size_t)w * h + k];
    float *dh = model == 0 ? sl.hyp_plane : sl.hyp_cyl;
    GMX_HIP(ctx, hipMemcpyAsync(dh, stage, (size_t)H * 32, hipMemcpyHostToDevice, sl.stream));
    launch_score(model, sl.valid4, labels ? sl.labels : nullptr, want, nullptr, n, dh, sl.band, H, tau, sl.score_partial,
                 sl.cnt_plane, sl.best_plane, sl.stream);
    GMX_HIP(ctx, hipMemcpyAsync(counts, sl.cnt_plane, (size_t)H * 4, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    return GM_OK;
}

}  // namespace

extern "C" {

int gm_ext_available(void) { return 1; }

gm_status gm_score_planes(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                          const float *hyp4, uint32_t H, double tau, int32_t *counts)
{
    return score(ctx, 0, xyz, n, labels, want, hyp4, H, tau, counts);
}

gm_status gm_score_cylinders(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                             const float *hyp7, uint32_t H, double tau, int32_t *counts)
{
    return score(ctx, 1, xyz, n, labels, want, hyp7, H, tau, counts);
}

gm_status gm_score_frame(gm_ctx *ctx, uint32_t slot, int model, const float *hyp, uint32_t H, double tau,
                         uint32_t unlabelled_only, int32_t *counts)
{
    gm_status st = gm_check_slot(ctx, slot);
    if (st != GM_OK) return st;
    if (!hyp || !counts || (model != 0 && model != 1)) return gm_fail(ctx, GM_ERR_INVALID_ARG, "gm_score_frame: bad argument");
    if (unlabelled_only && !(ctx->cfg.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)))
        return gm_fail(ctx, GM_ERR_NOT_READY, "unlabelled_only needs a context created with a GM_CFG_RANSAC_* flag");
    Slot &sl = ctx->slots[slot];
    st = gm_ensure_ext(ctx, sl, H);
    if (st != GM_OK) return st;
    // caller rows (4 or 7 floats) -> internal rows of 8
    const int w = model == 0 ? 4 : 7;
    std::vector<float> rows((size_t)H * 8, 0.f);
    for (uint32_t h = 0; h < H; ++h)
        for (int k = 0; k < w; ++k) rows[8 * (size_t)h + k] = hyp[(size_t)w * h + k];
    float *dh = model == 0 ? sl.hyp_plane : sl.hyp_cyl;
    GMX_HIP(ctx, hipMemcpyAsync(dh, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));  // `rows` is pageable and about to go away
    // exhaustive scorer over the slot's resident valid cloud (owned points only: halo rows never reach it)
    launch_score(model, sl.valid4, unlabelled_only ? sl.labels : nullptr, 0, &sl.ctr->n_valid, sl.last.n_valid, dh, sl.band,
                 H, tau, nullptr, sl.cnt_plane, sl.score_partial + 600, sl.stream);
    GMX_HIP(ctx, hipMemcpyAsync(counts, sl.cnt_plane, (size_t)H * 4, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    return GM_OK;
}

gm_status gm_plane_hypotheses(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                              uint64_t seed, uint32_t H, float *hyp4)
{
    Slot *slp;
    if (!ctx) return GM_ERR_INVALID_ARG;
    if ((n && !xyz) || !hyp4) return gm_fail(ctx, GM_ERR_INVALID_ARG, "NULL argument");
    gm_status st = prepare(ctx, slp, n, H);
    if (st != GM_OK) return st;
    Slot &sl = *slp;
    st = upload_xyz(ctx, sl, xyz, n, sl.valid4);
    if (st != GM_OK) return st;
    st = upload_labels(ctx, sl, labels, n);
    if (st != GM_OK) return st;
    launch_plane_hypotheses(sl.valid4, labels ? sl.labels : nullptr, want, nullptr, n, seed, H, sl.hyp_plane, nullptr,
                            sl.stream);
    if ((size_t)H * 32 > sl.raw_cap) {
        st = gm_ensure_capacity(ctx, sl, n, (size_t)H * 32, true);
        if (st != GM_OK) return st;
    }
    float *stage = (float *)sl.h_raw;
    GMX_HIP(ctx, hipMemcpyAsync(stage, sl.hyp_plane, (size_t)H * 32, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    for (uint32_t h = 0; h < H; ++h)
        for (int k = 0; k < 4; ++k) hyp4[4 * (size_t)h + k] = stage[8 * (size_t)h + k];
    return GM_OK;
}

gm_status gm_cylinder_hypotheses(gm_ctx *ctx, const float *xyz, const float *nxyzc, uint32_t n, const uint8_t *labels,
                                 uint32_t want, uint64_t seed, uint32_t H, float *hyp7)
{
    Slot *slp;
    if (!ctx) return GM_ERR_INVALID_ARG;
    if ((n && (!xyz || !nxyzc)) || !hyp7) return gm_fail(ctx, GM_ERR_INVALID_ARG, "NULL argument");
    gm_status st = prepare(ctx, slp, n, H);
    if (st != GM_OK) return st;
    Slot &sl = *slp;
    st = upload_xyz(ctx, sl, xyz, n, sl.valid4);
    if (st != GM_OK) return st;
    if (n) GMX_HIP(ctx, hipMemcpyAsync(sl.vnorm4, nxyzc, (size_t)n * 16, hipMemcpyHostToDevice, sl.stream));
    st = upload_labels(ctx, sl, labels, n);
    if (st != GM_OK) return st;
    launch_cylinder_hypotheses(sl.valid4, sl.vnorm4, labels ? sl.labels : nullptr, want, nullptr, n, seed, H, sl.hyp_cyl,
                               nullptr, nullptr, 0.0, sl.stream);
    if ((size_t)H * 32 > sl.raw_cap) {
        st = gm_ensure_capacity(ctx, sl, n, (size_t)H * 32, true);
        if (st != GM_OK) return st;
    }
    float *stage = (float *)sl.h_raw;
    GMX_HIP(ctx, hipMemcpyAsync(stage, sl.hyp_cyl, (size_t)H * 32, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    for (uint32_t h = 0; h < H; ++h)
        for (int k = 0; k < 7; ++k) hyp7[7 * (size_t)h + k] = stage[8 * (size_t)h + k];
    return GM_OK;
}

gm_status gm_segment_moments(gm_ctx *ctx, const float *xyz, const float *nxyzc, const uint8_t *labels, uint32_t n,
                             uint32_t label, double mom16[16])
{
    Slot *slp;
    if (!ctx) return GM_ERR_INVALID_ARG;
    if ((n && !xyz) || !mom16) return gm_fail(ctx, GM_ERR_INVALID_ARG, "NULL argument");
    gm_status st = prepare(ctx, slp, n, 1);
    if (st != GM_OK) return st;
    Slot &sl = *slp;
    st = upload_xyz(ctx, sl, xyz, n, sl.valid4);
    if (st != GM_OK) return st;
    if (n && nxyzc) GMX_HIP(ctx, hipMemcpyAsync(sl.vnorm4, nxyzc, (size_t)n * 16, hipMemcpyHostToDevice, sl.stream));
    st = upload_labels(ctx, sl, labels, n);
    if (st != GM_OK) return st;
    launch_segment_moments(sl.valid4, nxyzc ? sl.vnorm4 : nullptr, labels ? sl.labels : nullptr, label, nullptr, n,
                           sl.mom_partial, sl.mom_plane, sl.stream);
    GMX_HIP(ctx, hipMemcpyAsync(mom16, sl.mom_plane, 16 * sizeof(double), hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    return GM_OK;
}

gm_status gm_nearest(gm_ctx *ctx, const float *xyz, uint32_t n, const float *queries, uint32_t nq, int32_t *idx_out)
{
    Slot *slp;
    if (!ctx) return GM_ERR_INVALID_ARG;
    if ((n && !xyz) || (nq && (!queries || !idx_out))) return gm_fail(ctx, GM_ERR_INVALID_ARG, "NULL argument");
    const uint32_t m = n > nq ? n : nq;
    gm_status st = prepare(ctx, slp, m, 1);
    if (st != GM_OK) return st;
    Slot &sl = *slp;
    st = upload_xyz(ctx, sl, xyz, n, sl.valid4);
    if (st != GM_OK) return st;
    st = upload_xyz(ctx, sl, queries, nq, sl.vox4);
    if (st != GM_OK) return st;
    launch_nearest(sl.valid4, nullptr, n, sl.vox4, nullptr, nq, sl.nn_best, sl.vox_nn, sl.stream);
    if (nq) GMX_HIP(ctx, hipMemcpyAsync(idx_out, sl.vox_nn, (size_t)nq * 4, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    GMX_HIP(ctx, hipGetLastError());
    return GM_OK;
}

static gm_status fetch_bytes(gm_ctx *ctx, Slot &sl, const void *dev, uint32_t avail, size_t elem, void *out,
                             uint32_t capacity, uint32_t *n_out)
{
    if (n_out) *n_out = avail;
    if (avail > capacity) return gm_fail(ctx, GM_ERR_CAPACITY, "output buffer too small");
    if (!avail) return GM_OK;
    if (!out) return gm_fail(ctx, GM_ERR_INVALID_ARG, "output pointer is NULL");
    GMX_HIP(ctx, hipMemcpyAsync(out, dev, (size_t)avail * elem, hipMemcpyDeviceToHost, sl.stream));
    GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    return GM_OK;
}

gm_status gm_get_voxel_nearest(gm_ctx *ctx, uint32_t slot, int32_t *idx, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = gm_check_slot(ctx, slot);
    if (st != GM_OK) return st;
    if ((ctx->cfg.flags & (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID)) != (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID))
        return gm_fail(ctx, GM_ERR_NOT_READY, "context created without GM_CFG_NEAREST | GM_CFG_VOXEL_GRID");
    Slot &sl = ctx->slots[slot];
    return fetch_bytes(ctx, sl, sl.vox_nn, sl.last.n_voxels, 4, idx, capacity, n_out);
}

gm_status gm_get_voxel_normals(gm_ctx *ctx, uint32_t slot, float *nxyzc, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = gm_check_slot(ctx, slot);
    if (st != GM_OK) return st;
    if ((ctx->cfg.flags & (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID)) != (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID))
        return gm_fail(ctx, GM_ERR_NOT_READY, "context created without GM_CFG_NEAREST | GM_CFG_VOXEL_GRID");
    Slot &sl = ctx->slots[slot];
    return fetch_bytes(ctx, sl, sl.vox_nrm4, sl.last.n_voxels, 16, nxyzc, capacity, n_out);
}

gm_status gm_get_labels(gm_ctx *ctx, uint32_t slot, uint8_t *labels, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = gm_check_slot(ctx, slot);
    if (st != GM_OK) return st;
    if (!(ctx->cfg.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)))
        return gm_fail(ctx, GM_ERR_NOT_READY, "context created without a GM_CFG_RANSAC_* flag");
    Slot &sl = ctx->slots[slot];
    return fetch_bytes(ctx, sl, sl.labels, sl.last.n_valid, 1, labels, capacity, n_out);
}

gm_status gm_get_compressed_map(gm_ctx *ctx, uint32_t slot, void *buf, size_t capacity, size_t *n_bytes)
{
    gm_status st = gm_check_slot(ctx, slot);
    if (st != GM_OK) return st;
    Slot &sl = ctx->slots[slot];
    const gm_frame_result &r = sl.last;
    uint32_t nprim = 0;
    const bool has_plane = (ctx->cfg.flags & GM_CFG_RANSAC_PLANE) && r.plane_inliers > 0 && isfinite(r.plane[0]);
    const bool has_cyl = (ctx->cfg.flags & GM_CFG_RANSAC_CYLINDER) && r.cylinder_inliers > 0 && isfinite(r.cylinder[0]);
    nprim = (has_plane ? 1u : 0u) + (has_cyl ? 1u : 0u);
    const uint32_t nvox = (ctx->cfg.flags & GM_CFG_VOXEL_GRID) ? r.n_voxels : 0;
    const size_t need = sizeof(gm_map_header) + (size_t)nprim * sizeof(gm_map_primitive) + (size_t)nvox * 16;
    if (n_bytes) *n_bytes = need;
    if (need > capacity) return gm_fail(ctx, GM_ERR_CAPACITY, "map buffer too small");
    if (!buf) return gm_fail(ctx, GM_ERR_INVALID_ARG, "map buffer is NULL");
    uint8_t *p = (uint8_t *)buf;
    gm_map_header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "GMAP", 4);
    h.version = 1;
    h.n_primitives = nprim;
    h.n_voxels = nvox;
    h.leaf = (float)ctx->cfg.voxelGridLeafSize;
    h.bound = (float)ctx->cfg.boxFilterBound;
    h.n_points = r.n_valid;
    for (int k = 0; k < 3; ++k) { h.eigenvalues[k] = r.eigenvalues[k]; h.center_axis[k] = r.center_axis[k]; }
    memcpy(p, &h, sizeof(h)); p += sizeof(h);
    if (has_plane) {
        gm_map_primitive q;
        memset(&q, 0, sizeof(q));
        q.type = 1; q.inliers = r.plane_inliers;
        for (int k = 0; k < 4; ++k) q.params[k] = (float)r.plane_refit[k];
        memcpy(p, &q, sizeof(q)); p += sizeof(q);
    }
    if (has_cyl) {
        gm_map_primitive q;
        memset(&q, 0, sizeof(q));
        q.type = 2; q.inliers = r.cylinder_inliers;
        for (int k = 0; k < 7; ++k) q.params[k] = r.cylinder[k];
        memcpy(p, &q, sizeof(q)); p += sizeof(q);
    }
    if (nvox) {
        GMX_HIP(ctx, hipMemcpyAsync(p, sl.vox4, (size_t)nvox * 16, hipMemcpyDeviceToHost, sl.stream));
        GMX_HIP(ctx, hipStreamSynchronize(sl.stream));
    }
    return GM_OK;
}

}  // extern "C"
