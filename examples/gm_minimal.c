/* gm_minimal.c -- the whole drop-in boundary from plain C99: what a maintainer of the reference links against
 * (INTEGRATION.md par. 2c).  Builds with a bare C compiler, no HIP headers:
 *   gcc -std=c99 -O2 -I include examples/gm_minimal.c -L geometric_mapping_amd -lgm_hip \
 *       -Wl,-rpath,$PWD/geometric_mapping_amd -Wl,-rpath,/opt/rocm/lib -lm -o examples/gm_minimal
 * Runs one synthetic tunnel frame (R = 2 m along x, PointCloud2-style 32-byte rows) through gm_process_frame and
 * prints the centre axis, which must come out as +-x. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gm_hip.h"

static unsigned long long s = 88172645463325252ull;
static double urand(void) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 100000u, step = 32;
    unsigned char *rows = (unsigned char *)malloc((size_t)n * step);
    unsigned i;
    gm_config cfg;
    gm_ctx *ctx = NULL;
    gm_cloud cloud;
    gm_frame_result r;
    if (!rows) return 3;
    memset(rows, 0xAB, (size_t)n * step);                     /* intensity / ring / time bytes: ignored by the path */
    for (i = 0; i < n; ++i) {
        const double t = -6.0 + 12.0 * urand(), th = 6.283185307179586 * urand(), rr = 2.0 + 0.01 * (urand() - 0.5);
        const float p[3] = {(float)t, (float)(rr * cos(th)), (float)(rr * sin(th))};
        memcpy(rows + (size_t)i * step, p, sizeof p);
    }
    gm_default_config(&cfg);                                  /* launch/mapping.launch values */
    if (gm_create(&cfg, &ctx) != GM_OK) { fprintf(stderr, "gm_create: %s\n", gm_last_error(NULL)); return 1; }
    memset(&cloud, 0, sizeof cloud);
    cloud.data = rows; cloud.n_points = n; cloud.point_step = step;
    cloud.off_x = 0; cloud.off_y = 4; cloud.off_z = 8; cloud.flags = 0;
    if (gm_process_frame(ctx, &cloud, &r) != GM_OK) { fprintf(stderr, "gm_process_frame: %s\n", gm_last_error(ctx)); return 2; }
    printf("n_in=%u n_cropped=%u n_valid=%u n_voxels=%u axis=(%.6f %.6f %.6f) eigenvalues=(%.4g %.4g %.4g)\n", r.n_in,
           r.n_cropped, r.n_valid, r.n_voxels, r.center_axis[0], r.center_axis[1], r.center_axis[2], r.eigenvalues[0],
           r.eigenvalues[1], r.eigenvalues[2]);
    gm_destroy(ctx);
    free(rows);
    if (fabs(fabs(r.center_axis[0]) - 1.0) > 1e-3) { fprintf(stderr, "axis is not +-x\n"); return 4; }
    puts("gm_minimal ok");
    return 0;
}
