// gm_ext.hip -- extension entry points (placeholder until k_ransac.hip lands)
#include "gm_internal.hpp"

extern "C" {

static gm_status unsupported(gm_ctx *ctx, const char *what)
{
    if (ctx) ctx->err = std::string(what) + ": not implemented yet";
    return GM_ERR_UNSUPPORTED;
}

gm_status gm_get_voxel_nearest(gm_ctx *ctx, uint32_t, int32_t *, uint32_t, uint32_t *) { return unsupported(ctx, "gm_get_voxel_nearest"); }
gm_status gm_get_labels(gm_ctx *ctx, uint32_t, uint8_t *, uint32_t, uint32_t *) { return unsupported(ctx, "gm_get_labels"); }
gm_status gm_nearest(gm_ctx *ctx, const float *, uint32_t, const float *, uint32_t, int32_t *) { return unsupported(ctx, "gm_nearest"); }
gm_status gm_score_planes(gm_ctx *ctx, const float *, uint32_t, const float *, uint32_t, double, int32_t *) { return unsupported(ctx, "gm_score_planes"); }
gm_status gm_score_cylinders(gm_ctx *ctx, const float *, uint32_t, const float *, uint32_t, double, int32_t *) { return unsupported(ctx, "gm_score_cylinders"); }
gm_status gm_plane_hypotheses(gm_ctx *ctx, const float *, uint32_t, uint64_t, uint32_t, float *) { return unsupported(ctx, "gm_plane_hypotheses"); }
gm_status gm_cylinder_hypotheses(gm_ctx *ctx, const float *, const float *, uint32_t, uint64_t, uint32_t, float *) { return unsupported(ctx, "gm_cylinder_hypotheses"); }
gm_status gm_segment_moments(gm_ctx *ctx, const float *, const float *, const uint8_t *, uint32_t, uint32_t, double *) { return unsupported(ctx, "gm_segment_moments"); }
gm_status gm_get_compressed_map(gm_ctx *ctx, uint32_t, void *, size_t, size_t *) { return unsupported(ctx, "gm_get_compressed_map"); }

}
