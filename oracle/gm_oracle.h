/*
 * gm_oracle.h -- CPU restatement of the geometric_mapping per-frame path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (/root/reference) ships no tests, fixtures
 * or golden vectors and cannot be built here (ROS, PCL, FLANN, Eigen absent),
 * so this restatement is pinned only by analytic known-answer cases and by an
 * independent numpy/scipy twin (oracle/oracle_np.py), never by reference output.
 *
 * The arithmetic lives in third-party code the reference calls:
 *   PCL 1.8.1 (inferred, versions are not pinned: package.xml:51-63,
 *   CMakeLists.txt:18-19), FLANN 1.9.1, Eigen 3.3.4.
 * Each function below cites the reference call site it restates
 * (paths relative to /root/reference).
 *
 * Clouds are row-major float32 [n][3]; normals are float32 [n][4] =
 * (nx, ny, nz, curvature), i.e. the meaningful fields of pcl::Normal.
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numeric modes */
#define GMO_F64 0          /* mathematical truth: same neighbour sets, double sums, Jacobi */
#define GMO_F32_FAITHFUL 1 /* PCL<=1.9 / Eigen 3.3 fp32 operation order as far as restatable */
#define GMO_F32_SHIFTED 2  /* as 1, but the PCL >= 1.10 computeMeanAndCovarianceMatrix: fp32 accumulators of the offsets
                              from the neighbourhood's FIRST point (package.xml pins no PCL version: SURVEY.md par. 8c) */

/* src/tunnel_processing.cpp:39-49 chopCloud -> pcl::CropBox::applyFilter.
 * Keeps i iff !(x<-b || y<-b || z<-b || x>b || y>b || z>b) with b cast to
 * float (:43-44); NaN rows are dropped (they are dropped by PCL when the cloud
 * is not dense; for a dense-flagged cloud holding NaNs the reference is
 * undefined).  Survivor indices in input order -> idx_out.  Returns n'. */
int gmo_crop_box(const float *xyz, int n, double bound, int *idx_out);

/* src/tunnel_processing.cpp:58-70 getNormals -> pcl::NormalEstimation::compute
 * with a radius search (FLANN L2_Simple, strict d2 < (float)(r*r), self
 * included), computeMeanAndCovarianceMatrix, solvePlaneParameters/eigen33,
 * flipNormalTowardsViewpoint(0,0,0).  <3 neighbours -> NaN normal+curvature.
 * counts_out (may be NULL) receives the neighbour count per point.
 * nthreads<=1 runs serially (how the reference runs). */
int gmo_normals(const float *xyz, int n, double radius, int mode, int nthreads,
                float *normals_out, int *counts_out);

/* src/tunnel_processing.cpp:74-85 removeNaNNormalsFromPointCloud +
 * ExtractIndices: indices of rows whose nx,ny,nz are all finite, ascending. */
int gmo_finite_normals(const float *normals, int n, int *idx_out);

/* src/tunnel_processing.cpp:214-220 pcl::VoxelGrid::applyFilter, cubic leaf.
 * out_xyz [n][3] capacity; out_key/out_count may be NULL.  Returns V, the
 * number of occupied voxels (ascending key order).  *passthrough=1 when PCL's
 * "leaf size too small" guard fires (output = input). */
int gmo_voxel_grid(const float *xyz, int n, double leaf, int mode,
                   float *out_xyz, int32_t *out_key, int32_t *out_count, int *passthrough);

/* src/tunnel_processing.cpp:92-148 getLocalFrame: w_i = exp((c_i+.001/wf)^2),
 * M = sum w_i^2 n_i n_i^T, symmetric eigen-decomposition, eigenvalues
 * ascending, eigenvectors as columns (evecs is column-major like
 * Eigen::Matrix3f).  M_out (may be NULL) row-major 3x3 double. */
void gmo_local_frame(const float *normals, int n, double weighting_factor, int mode,
                     double *M_out, float *evals, float *evecs);

/* src/tunnel_processing.cpp:237-239 kdtree->nearestKSearch(q,1): index of the
 * nearest cloud point to each query (fp32 L2_Simple distance, lowest index on
 * ties). */
void gmo_nearest(const float *xyz, int n, const float *queries, int nq, int *idx_out);

/* Whole callback, src/geometric_mapping.cpp:48-125 (processing half only). */
typedef struct {
    int n_in, n_cropped, n_valid, n_voxels;
    float evals[3];
    float evecs[9];     /* column-major */
    double M[9];        /* row-major */
} gmo_frame_result;

/* out_xyz [n][3], out_normals [n][4], out_vox [n][3] may each be NULL. */
int gmo_process_frame(const float *xyz, int n, double bound, double radius, double leaf,
                      double weighting_factor, int mode, int nthreads,
                      float *out_xyz, float *out_normals, float *out_vox,
                      gmo_frame_result *res);

/* ---- build-defined extensions (no reference counterpart) ---- */

/* counter-based PRNG shared by the oracle and the device hypothesis generator */
uint64_t gmo_mix64(uint64_t x);

/* Hypotheses from seeded minimal samples of the (valid) cloud.
 * plane: 3 points -> (a,b,c,d), unit normal, a*x+b*y+c*z+d=0
 *        (PCL SampleConsensusModelPlane::computeModelCoefficients).
 * cylinder: 2 points + their normals -> (px,py,pz, dx,dy,dz, r)
 *        (PCL SampleConsensusModelCylinder::computeModelCoefficients).
 * A degenerate sample yields a hypothesis of NaNs (scores 0 inliers). */
/* labels (may be NULL): samples are drawn only from points with labels[i]==want. */
void gmo_plane_hypotheses(const float *xyz, int n, const uint8_t *labels, int want, uint64_t seed, int H,
                          float *hyp4);
void gmo_cylinder_hypotheses(const float *xyz, const float *normals, int n, const uint8_t *labels, int want,
                             uint64_t seed, int H, float *hyp7);

/* inlier counts: plane |n.p+d| < tau ; cylinder |dist(p,axis)-r| < tau.
 * mask (may be NULL): only points with mask[i]==want participate. */
void gmo_score_planes(const float *xyz, int n, const uint8_t *mask, int want,
                      const float *hyp4, int H, double tau, int32_t *counts);
void gmo_score_cylinders(const float *xyz, int n, const uint8_t *mask, int want,
                         const float *hyp7, int H, double tau, int32_t *counts);

/* label inliers of one model: labels[i]=label where mask matches and the point
 * is within tau.  Returns the inlier count. */
int gmo_label_plane(const float *xyz, int n, uint8_t *labels, int want, int label,
                    const float *hyp4, double tau);
int gmo_label_cylinder(const float *xyz, int n, uint8_t *labels, int want, int label,
                       const float *hyp7, double tau);

/* per-segment moments over points with labels[i]==label:
 * mom[0]=count, mom[1..3]=sum p, mom[4..9]=sum pp^T (xx,xy,xz,yy,yz,zz),
 * mom[10..15]=sum nn^T (same order); all double, unshifted. */
void gmo_segment_moments(const float *xyz, const float *normals, int n, const uint8_t *labels,
                         int label, double *mom16);

/* refits from moments.  plane: centroid + min-eigenvector of the covariance
 * -> (a,b,c,d).  cylinder axis: min-eigenvector of sum nn^T. */
void gmo_refit_plane(const double *mom16, double *plane4);
void gmo_refit_axis(const double *mom16, double *axis3);

/* symmetric 3x3 eigen (double Jacobi), ascending, evecs column-major */
void gmo_eig3(const double *A_rowmajor, double *evals, double *evecs_colmajor);

#ifdef __cplusplus
}
#endif
#endif
