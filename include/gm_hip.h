/*
 * gm_hip.h -- C ABI of libgm_hip.so: the MI355X (gfx950) implementation of the
 * geometric_mapping per-frame point-cloud path.
 *
 * The reference exposes this path as plain C++ free functions linked into one
 * ROS executable (there is no plugin / FFI layer to bind to):
 *   /root/reference include/geometric_mapping/tunnel_processing.hpp:38-54,77-82
 *   called only from cloud_cb, /root/reference src/geometric_mapping.cpp:48-125.
 * This header is the boundary a catkin host links instead; every entry point
 * cites the reference interface it replaces.  No PCL / Eigen / ROS / torch
 * types cross it: plain pointers, sizes and POD structs only.  Nothing throws.
 *
 * Threading: a gm_ctx is NOT thread-safe (the reference's callback never
 * re-enters either: ros::spin(), src/geometric_mapping.cpp:169).  Use one
 * context per calling thread.  All device work of a context runs on HIP
 * streams it owns (one per slot).
 *
 * There is no CPU fallback behind this ABI: if no gfx950 device is usable,
 * gm_create fails with GM_ERR_DEVICE.
 */
#ifndef GM_HIP_H
#define GM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_ABI_VERSION 3u

typedef struct gm_ctx gm_ctx; /* opaque */

typedef enum gm_status {
    GM_OK = 0,
    GM_ERR_INVALID_ARG = 1,
    GM_ERR_TOO_FEW_POINTS = 2, /* reserved: stages accept empty clouds like the reference does */
    GM_ERR_DEVICE = 3,         /* a HIP call failed; gm_last_error() has the hipError string */
    GM_ERR_OOM = 4,
    GM_ERR_CAPACITY = 5,       /* caller buffer too small; *n_out holds the needed count */
    GM_ERR_NOT_READY = 6,      /* slot has no submitted / completed frame */
    GM_ERR_UNSUPPORTED = 7,
    GM_ERR_COMM = 8            /* RCCL could not be loaded or a collective failed (gm_group_*) */
} gm_status;

/* gm_config.flags */
#define GM_CFG_VOXEL_GRID      (1u << 0) /* run VoxelGrid every frame, as the reference does
                                            (src/geometric_mapping.cpp:70-75 is not gated) */
#define GM_CFG_NEAREST         (1u << 1) /* also 1-NN of every voxel centroid
                                            (src/tunnel_processing.cpp:237-239; only /surfaceNormals needs it) */
#define GM_CFG_RANSAC_PLANE    (1u << 2) /* extension, no reference counterpart */
#define GM_CFG_RANSAC_CYLINDER (1u << 3) /* extension, no reference counterpart */
#define GM_CFG_STAGE_TIMING    (1u << 4) /* bracket stages with hipEvents -> gm_frame_result.stage_ms */
#define GM_CFG_KEEP_COUNTS     (1u << 5) /* keep per-point neighbour counts (tests) */
#define GM_CFG_GRAPH           (1u << 6) /* capture the frame's launch chain as a hipGraph once and replay it for every
                                            frame of about the same size (launch-bound small frames; results are the
                                            same bit for bit; normals_kernel_ms / stage_ms are not measured) */
#define GM_CFG_DEFAULT         (GM_CFG_VOXEL_GRID)

/* The four numeric parameters are the reference's, with its types:
 * include/geometric_mapping/paramHandler.hpp:26-29 (all double). */
typedef struct gm_config {
    uint32_t struct_size;        /* = sizeof(gm_config) */
    uint32_t flags;              /* GM_CFG_* */
    double   boxFilterBound;     /* launch/mapping.launch:7  */
    double   voxelGridLeafSize;  /* launch/mapping.launch:8  */
    double   neighborRadius;     /* launch/mapping.launch:9  */
    double   weightingFactor;    /* launch/mapping.launch:10 */
    int32_t  device;             /* HIP device ordinal */
    uint32_t n_slots;            /* frames in flight (>=1); 2 overlaps H2D of frame i+1 with compute of i.  Every slot has its own
                                    HIP stream; ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), so
                                    for more than 3 slots export GPU_MAX_HW_QUEUES=8 before the process's first HIP call (measured:
                                    4 slots 0.211 ms per 1 M-point frame with 8 queues, 0.271 with 4; 3 slots 0.227) */
    uint32_t max_points;         /* capacity hint; buffers grow on demand */
    uint32_t ransac_hypotheses;  /* H per model per frame (extension) */
    double   ransac_threshold;   /* tau, metres (extension) */
    uint64_t ransac_seed;        /* extension */
} gm_config;

/* gm_cloud.flags */
#define GM_CLOUD_DEVICE    (1u << 0) /* data is a device pointer valid on the context's device */
#define GM_CLOUD_BIGENDIAN (1u << 1) /* sensor_msgs/PointCloud2.is_bigendian */
#define GM_CLOUD_PINNED    (1u << 2) /* data is page-locked host memory from gm_host_alloc: it is copied to the device
                                        straight from there (no staging copy on the calling thread) and must stay
                                        untouched until gm_wait_frame / the blocking call returns */

/* One sensor_msgs/PointCloud2 worth of rows: what pcl::fromROSMsg reads at
 * src/geometric_mapping.cpp:55.  x,y,z are float32 at byte offsets off_* of
 * each point_step-byte row. */
typedef struct gm_cloud {
    const void *data;
    uint32_t n_points;
    uint32_t point_step;
    uint32_t off_x, off_y, off_z;
    uint32_t flags;
} gm_cloud;

enum { GM_STAGE_UPLOAD = 0, GM_STAGE_CROP, GM_STAGE_GRID, GM_STAGE_NORMALS, GM_STAGE_COMPACT,
       GM_STAGE_FRAME, GM_STAGE_VOXEL, GM_STAGE_RANSAC, GM_STAGE_TOTAL, GM_N_STAGES };

/* gm_frame_result.status_flags */
#define GM_RES_VOXEL_PASSTHROUGH (1u << 0) /* PCL "leaf size too small" guard: voxel output = input */

typedef struct gm_frame_result {
    uint32_t n_in;         /* points received */
    uint32_t n_cropped;    /* after chopCloud                          (src/geometric_mapping.cpp:57) */
    uint32_t n_valid;      /* after NaN-normal removal inside getNormals (:63)                       */
    uint32_t n_voxels;     /* VoxelGrid output size                    (src/tunnel_processing.cpp:220) */
    float    eigenvalues[3];   /* ascending, = *eigenVals              (src/tunnel_processing.cpp:132) */
    float    eigenvectors[9];  /* column-major like Eigen::Matrix3f    (src/tunnel_processing.cpp:136) */
    float    center_axis[3];   /* eigenvectors column 0                (src/geometric_mapping.cpp:92)  */
    uint32_t status_flags;
    double   scatter[6];       /* M = sum w^2 n n^T: xx,xy,xz,yy,yz,zz in fp64 (multi-GPU merge unit) */
    /* extensions (valid only with the GM_CFG_RANSAC_* flags) */
    uint32_t plane_inliers, cylinder_inliers;
    float    plane[4];         /* a,b,c,d  (unit normal)                */
    float    cylinder[7];      /* point on axis, unit axis direction, radius */
    double   plane_refit[4];   /* least-squares refit over the plane segment */
    double   cylinder_axis_refit[3]; /* min-eigenvector of sum nn^T over the cylinder segment */
    float    stage_ms[GM_N_STAGES];  /* device time per stage when GM_CFG_STAGE_TIMING; else 0 */
    float    normals_kernel_ms;      /* the neighbourhood-normals kernel alone (hipEvent bracketed) */
} gm_frame_result;

/* ---- lifetime ------------------------------------------------------------ */

/* Replaces the start-up half of main(): src/geometric_mapping.cpp:128-161
 * (parameters are read once; there is no dynamic reconfigure).  Picks the
 * device, creates one stream + pinned staging + device buffers per slot. */
gm_status gm_create(const gm_config *cfg, gm_ctx **out);
void gm_destroy(gm_ctx *ctx);

/* Page-locked host memory for GM_CLOUD_PINNED input (e.g. the buffer a driver or a custom ROS allocator fills with
 * PointCloud2 rows).  Owned by the caller until gm_host_free; outlives nothing: free it before gm_destroy. */
gm_status gm_host_alloc(gm_ctx *ctx, size_t bytes, void **out);
gm_status gm_host_free(gm_ctx *ctx, void *ptr);
/* The same for memory the caller already owns (e.g. the data vector of a pre-allocated sensor_msgs/PointCloud2 that a
 * node publishes from: gm_set_cloud_output then delivers /choppedCloud straight into the message): page-locks
 * [ptr, ptr + bytes) until gm_host_unregister.  Registered memory may be used wherever gm_host_alloc memory may. */
gm_status gm_host_register(gm_ctx *ctx, void *ptr, size_t bytes);
gm_status gm_host_unregister(gm_ctx *ctx, void *ptr);

/* Defaults = the launch file's values (launch/mapping.launch:7-10). */
void gm_default_config(gm_config *cfg);

uint32_t gm_abi_version(void);
const char *gm_status_string(gm_status s);
/* Message of the last failure on this context ("" if none).  ctx may be NULL
 * for a failure inside gm_create. */
const char *gm_last_error(const gm_ctx *ctx);

/* ---- the per-frame callback ------------------------------------------------ */

/* Replaces the processing half of cloud_cb, src/geometric_mapping.cpp:55-92:
 * fromROSMsg -> chopCloud -> getNormals -> VoxelGrid (inside rvizNormals) ->
 * getLocalFrame -> centerAxis.  Blocking; uses slot 0. */
gm_status gm_process_frame(gm_ctx *ctx, const gm_cloud *cloud, gm_frame_result *res);

/* Same work split for streaming (BASELINE config 5): submit returns once the
 * host buffer has been staged and all device work is enqueued on the slot's
 * stream; wait blocks until that frame's result is on the host.  The host
 * buffer may be reused as soon as submit returns. */
gm_status gm_submit_frame(gm_ctx *ctx, uint32_t slot, const gm_cloud *cloud);
gm_status gm_wait_frame(gm_ctx *ctx, uint32_t slot, gm_frame_result *res);
/* Completion query, never blocks: GM_OK when the slot's submitted frame has finished (gm_wait_frame will return at once),
 * GM_ERR_NOT_READY while it is still running or when the slot holds no submitted frame.  The reference publishes every
 * frame inside its own callback (src/geometric_mapping.cpp:100-117); a host that keeps frames in flight polls at the top
 * of each callback (and on a timer) and publishes what has finished, instead of waiting for the pipeline to fill. */
gm_status gm_poll_frame(gm_ctx *ctx, uint32_t slot);

/* /choppedCloud straight into host memory (the default launch has displayCloud = true: launch/mapping.launch:12,
 * src/geometric_mapping.cpp:100-107).  xyzw = page-locked memory from gm_host_alloc with room for capacity rows of
 * 4 floats (x, y, z, pad = input row index), owned by the caller, or NULL to switch the output off for the slot.  Every
 * later frame of the slot copies its valid cloud there -- the copy starts right behind the NaN-normal compaction, on a
 * stream of its own, and runs while the rest of the frame (voxel grid, RANSAC, eigen solve) executes -- and is complete
 * when gm_wait_frame returns; rows [0, n_valid) are the frame's, identical to gm_get_cropped_xyz's.  A frame with more
 * points than capacity is refused with GM_ERR_CAPACITY at submit. */
gm_status gm_set_cloud_output(gm_ctx *ctx, uint32_t slot, float *xyzw, uint32_t capacity);

/* Bulky per-frame outputs of a completed slot, fetched only when a display
 * flag needs them (src/geometric_mapping.cpp:100-117).  `capacity` counts
 * points; *n_out receives the available count (also on GM_ERR_CAPACITY).
 *   cropped xyz : /choppedCloud = cloudChopped AFTER the in-place NaN compaction
 *                 (src/tunnel_processing.cpp:81-85), rows of 4 floats x,y,z,pad
 *                 = pcl::PointXYZ; pad holds the point's row index in the input
 *                 cloud as int32 bits.
 *   normals     : rows nx,ny,nz,curvature (the meaningful fields of pcl::Normal)
 *   voxels      : VoxelGrid centroids in ascending voxel-key order, rows x,y,z,count
 *   nearest     : for each voxel centroid, index (into the cropped cloud) of its
 *                 nearest point (needs GM_CFG_NEAREST)
 *   voxel normals: normals->at(kIndices[0]) of rvizNormals' marker loop
 *                 (src/tunnel_processing.cpp:237-249): the normal (nx,ny,nz,curvature) of that nearest
 *                 point, one row per voxel centroid, gathered on the device (needs GM_CFG_NEAREST) --
 *                 with the centroids this is everything /surfaceNormals needs, a few KB instead of
 *                 the whole normals cloud */
gm_status gm_get_cropped_xyz(gm_ctx *ctx, uint32_t slot, float *xyzw, uint32_t capacity, uint32_t *n_out);
gm_status gm_get_normals(gm_ctx *ctx, uint32_t slot, float *nxyzc, uint32_t capacity, uint32_t *n_out);
gm_status gm_get_voxel_centroids(gm_ctx *ctx, uint32_t slot, float *xyzc, uint32_t capacity, uint32_t *n_out);
gm_status gm_get_voxel_nearest(gm_ctx *ctx, uint32_t slot, int32_t *idx, uint32_t capacity, uint32_t *n_out);
gm_status gm_get_voxel_normals(gm_ctx *ctx, uint32_t slot, float *nxyzc, uint32_t capacity, uint32_t *n_out);
/* per-point neighbour counts of the cropped cloud, pre-compaction order (GM_CFG_KEEP_COUNTS) */
gm_status gm_get_neighbor_counts(gm_ctx *ctx, uint32_t slot, int32_t *counts, uint32_t capacity, uint32_t *n_out);
/* extension: per-point segment label of the valid cloud: 0 none, 1 plane, 2 cylinder */
gm_status gm_get_labels(gm_ctx *ctx, uint32_t slot, uint8_t *labels, uint32_t capacity, uint32_t *n_out);

/* ---- the reference's stage functions, one call each ------------------------ */
/* Host buffers in, host buffers out (blocking, slot 0).  These exist so the
 * four signatures of tunnel_processing.hpp can be re-implemented one-to-one. */

/* chopCloud(bound, cloud): tunnel_processing.hpp:38, src/tunnel_processing.cpp:39-49.
 * Order-preserving.  out rows x,y,z,pad(=input row index). */
gm_status gm_chop_cloud(gm_ctx *ctx, const gm_cloud *cloud, double bound,
                        float *xyzw_out, uint32_t capacity, uint32_t *n_out);

/* getNormals(radius, cloud&, kdtree&): tunnel_processing.hpp:41-45,
 * src/tunnel_processing.cpp:52-89.  xyz rows of 3 floats in; compacted cloud
 * (rows x,y,z,pad(=input row)) and normals (nx,ny,nz,curvature) out, same
 * length and order, NaN-normal rows removed. */
gm_status gm_get_normals_stage(gm_ctx *ctx, const float *xyz, uint32_t n, double radius,
                               float *xyzw_out, float *nxyzc_out, uint32_t capacity, uint32_t *n_out);

/* getLocalFrame(n, wf, normals, vals*&, vecs*&): tunnel_processing.hpp:48-54,
 * src/tunnel_processing.cpp:92-148.  eigenvectors column-major. */
gm_status gm_get_local_frame(gm_ctx *ctx, const float *nxyzc, uint32_t n, double weighting_factor,
                             float eigenvalues[3], float eigenvectors[9], double scatter6[6]);

/* The pcl::VoxelGrid half of rvizNormals: tunnel_processing.hpp:77-82,
 * src/tunnel_processing.cpp:214-220.  out rows x,y,z,count. */
gm_status gm_voxel_grid(gm_ctx *ctx, const float *xyz, uint32_t n, double leaf,
                        float *xyzc_out, uint32_t capacity, uint32_t *n_out, uint32_t *status_flags);

/* kdtree->nearestKSearch(q, 1): src/tunnel_processing.cpp:237-239. */
gm_status gm_nearest(gm_ctx *ctx, const float *xyz, uint32_t n, const float *queries, uint32_t nq,
                     int32_t *idx_out);

/* ---- multi-GPU merge unit (SURVEY.md par. 8e) ------------------------------- */

/* Eigen-solve a merged scatter matrix (sum over shards of gm_frame_result.scatter).
 * Pure host arithmetic on 6 doubles; same Jacobi as the device epilogue. */
gm_status gm_solve_local_frame(const double scatter6[6], float eigenvalues[3], float eigenvectors[9]);

/* Restrict which cropped points act as QUERY points: only points whose x lies in
 * [own_lo, own_hi) get a normal (the rest are halo: neighbours only, dropped
 * from the outputs).  Defaults to (-inf, +inf).  Used by slab sharding. */
gm_status gm_set_owned_range(gm_ctx *ctx, double own_lo, double own_hi);

/* ---- extensions without a reference counterpart (SURVEY.md par. 8a-ext) ----- */
/* getCylinder is an empty stub in the reference (src/tunnel_processing.cpp:149-154);
 * these follow PCL's SampleConsensusModelPlane / SampleConsensusModelCylinder
 * conventions and are checked against oracle/gm_oracle_ext.c + analytic truth only.
 * With GM_CFG_RANSAC_PLANE / _CYLINDER a frame additionally runs, on its valid cloud:
 * seeded hypotheses -> batched scoring (preemptive, three stages: all H hypotheses on
 * every 64th point, the 128 best of them on every 16th point, the 8 best of those on
 * every point; order = count descending, hypothesis index ascending; H <= 128 starts at
 * the second stage, H <= 8 is exhaustive) -> best model -> inlier labels (1 plane,
 * 2 cylinder; the cylinder samples and scores only points the plane left) ->
 * per-segment moments -> refits, reported in gm_frame_result. */

int gm_ext_available(void); /* 1 when the extension kernels are built in */

/* labels (may be NULL): only points with labels[i]==want take part / are sampled. */

/* Score caller-supplied hypotheses against a host cloud (xyz rows of 3 floats).
 * plane rows a,b,c,d: inlier iff |a x + b y + c z + d| < tau  (fp32, fma chain).
 * cylinder rows px,py,pz,dx,dy,dz,r: inlier iff (r-tau)^2 < dist_axis^2 < (r+tau)^2. */
gm_status gm_score_planes(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                          const float *hyp4, uint32_t H, double tau, int32_t *counts);
gm_status gm_score_cylinders(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                             const float *hyp7, uint32_t H, double tau, int32_t *counts);
/* Seeded minimal-sample hypotheses generated on the device (splitmix64 counter PRNG). */
/* Score caller-supplied hypotheses against the valid cloud a completed frame left in `slot` (no upload of points).
 * model 0: plane rows a,b,c,d; 1: cylinder rows px,py,pz,dx,dy,dz,r.  unlabelled_only != 0 counts only points the
 * frame's own RANSAC left unlabelled.  With gm_set_owned_range the valid cloud holds owned points only, so the counts
 * of the ranks of a sharded frame add up to the count on the whole frame: the building block of the multi-GPU
 * primitive vote (geometric_mapping_amd/sharding.py, DESIGN.md par. 6). */
gm_status gm_score_frame(gm_ctx *ctx, uint32_t slot, int model, const float *hyp, uint32_t H, double tau,
                         uint32_t unlabelled_only, int32_t *counts);
gm_status gm_plane_hypotheses(gm_ctx *ctx, const float *xyz, uint32_t n, const uint8_t *labels, uint32_t want,
                              uint64_t seed, uint32_t H, float *hyp4);
gm_status gm_cylinder_hypotheses(gm_ctx *ctx, const float *xyz, const float *nxyzc, uint32_t n,
                                 const uint8_t *labels, uint32_t want, uint64_t seed, uint32_t H, float *hyp7);
/* Per-segment moments over points with labels[i]==label (labels NULL: all points):
 * mom16 = count, sum p (3), sum pp^T (xx,xy,xz,yy,yz,zz), sum nn^T (same order); fp64. */
gm_status gm_segment_moments(gm_ctx *ctx, const float *xyz, const float *nxyzc, const uint8_t *labels,
                             uint32_t n, uint32_t label, double mom16[16]);

/* "Compressed map" record of a completed slot.  The reference defines no such
 * output; this is a build-defined format: header, primitive records,
 * then n_voxels rows of x,y,z,count (float32).  Returns the bytes needed in
 * *n_bytes (also on GM_ERR_CAPACITY). */
typedef struct gm_map_header {
    char     magic[4];       /* "GMAP" */
    uint32_t version;        /* 1 */
    uint32_t n_primitives;
    uint32_t n_voxels;
    float    leaf, bound;
    uint32_t n_points;       /* valid points the map was built from */
    uint32_t reserved;
    float    eigenvalues[3];
    float    center_axis[3];
} gm_map_header;
typedef struct gm_map_primitive {
    uint32_t type;           /* 1 plane (a,b,c,d refit), 2 cylinder (point, axis, radius) */
    uint32_t inliers;
    float    params[7];
    float    pad;
} gm_map_primitive;
gm_status gm_get_compressed_map(gm_ctx *ctx, uint32_t slot, void *buf, size_t capacity, size_t *n_bytes);

/* ---- multi-device group: one host thread, every local GPU ---------------------
 * The reference is one single-threaded process (ros::spin(), src/geometric_mapping.cpp:146,169: subscriber queue 1, one
 * frame at a time) on one CPU core; it has no counterpart for this section.  north_star: "Frames shard spatially across
 * the 8 GPUs of one node with an RCCL all-gather of fitted primitives over xGMI only when a scan exceeds single-GPU
 * capacity"; BASELINE configs[3] (one frame over 4 GPUs) and configs[4] (frames streamed over 8 GPUs).
 *
 * A group owns one gm_ctx per rank (one rank per device) and, across distinct devices, one RCCL communicator per rank
 * (ncclCommInitAll, single process; librccl is loaded at run time, the copy the process already has if it has one).
 *
 * SHARDED FRAME.  gm_group_process_frame cuts ONE frame into x-slabs balanced by the count of in-box points, adds a
 * 1.01 * neighborRadius halo (neighbours only, never outputs: gm_set_owned_range) ON THE HOST before H2D -- one parallel
 * pass for a histogram of x over the VoxelGrid lattice, edges on lattice planes (exactly, by the kernels' own float
 * expression) so that no voxel straddles two ranks, one parallel pass that scatters the rows into per-rank page-locked
 * buffers -- runs the unchanged single-GPU pipeline on every rank asynchronously, and exchanges the results with ONE
 * ncclAllGather of a 24-double record per rank (scatter partials, counts, the rank's fitted plane / cylinder).  The
 * merged frame: scatter = sum of the partials (rank order), 3x3 solve, counts summed, voxel centroids of the ranks in
 * ascending pcl key order (bit for bit the unsharded frame's; a lattice too coarse to cut along is merged through the
 * ranks' exact fixed-point voxel sums instead); fitted primitives by vote -- every rank's fit is a candidate, every rank
 * counts every candidate's inliers on its own resident owned points (gm_score_frame), the largest total wins.
 *
 * STREAMING.  Frames that fit one GPU are independent: gm_group_submit_frame hands a whole frame to the next device in
 * turn (its next free slot; gm_config.n_slots frames in flight per device), gm_group_wait_frame returns the frames in
 * submission order.  No collective.  Results are those of gm_process_frame on that device, bit for bit. */
typedef struct gm_group gm_group; /* opaque */
#define GM_GROUP_LOOPBACK (1u << 0) /* ranks may share a device (tests on a 1-GPU box): the records travel by device
                                       copies instead of RCCL; everything else is the same code */
gm_status gm_group_create(const gm_config *cfg, const int32_t *devices, uint32_t n_ranks, uint32_t flags, gm_group **out);
void gm_group_destroy(gm_group *grp);
uint32_t gm_group_size(const gm_group *grp);
/* the rank's own context, for the per-rank accessors (gm_get_normals, gm_get_voxel_centroids, ...) */
gm_ctx *gm_group_ctx(gm_group *grp, uint32_t rank);
const char *gm_group_last_error(const gm_group *grp); /* grp may be NULL for a failure inside gm_group_create */
/* One sharded frame, blocking.  cloud must be host rows.  res: the merged frame (n_cropped counts every in-box point
 * once; eigen results from the summed scatter; plane / cylinder = the vote's winners with their global inlier counts).
 * After a failure nothing of the frame is in flight and the accessors below report GM_ERR_NOT_READY; after a failed
 * collective (GM_ERR_COMM) the group refuses further work. */
gm_status gm_group_process_frame(gm_group *grp, const gm_cloud *cloud, gm_frame_result *res);
/* /choppedCloud of the last sharded frame in the single-GPU order (ascending input row); rows x,y,z,pad(= input row) */
gm_status gm_group_get_cropped_xyz(gm_group *grp, float *xyzw, uint32_t capacity, uint32_t *n_out);
/* pcl::VoxelGrid output of the last sharded frame, ascending key order: rows x,y,z,count (src/tunnel_processing.cpp:217-220) */
gm_status gm_group_get_voxel_centroids(gm_group *grp, float *xyzc, uint32_t capacity, uint32_t *n_out);
/* per centroid: nx,ny,nz,curvature of its nearest valid point / that point's index in gm_group_get_cropped_xyz's order
 * (kdtree->nearestKSearch + normals->at of the marker loop, src/tunnel_processing.cpp:237-249); the search runs on every
 * rank, the closest point of all wins.  Needs GM_CFG_NEAREST | GM_CFG_VOXEL_GRID. */
gm_status gm_group_get_voxel_normals(gm_group *grp, float *nxyzc, uint32_t capacity, uint32_t *n_out);
gm_status gm_group_get_voxel_nearest(gm_group *grp, int32_t *idx, uint32_t capacity, uint32_t *n_out);
/* wall-clock split of the last gm_group_process_frame call, milliseconds */
enum { GM_GROUP_T_CUT = 0,    /* host: histogram, edges, rows scattered into the per-rank page-locked buffers */
       GM_GROUP_T_SUBMIT = 1, /* host: the ranks' frames enqueued (H2D + launch chains) */
       GM_GROUP_T_DEVICE = 2, /* waiting for H2D, kernels and the all-gather of every rank */
       GM_GROUP_T_MERGE = 3,  /* records merged, voxel lists merged, primitive vote */
       GM_GROUP_T_TOTAL = 4,
       GM_GROUP_N_TIMINGS = 5 };
gm_status gm_group_get_timing(const gm_group *grp, double *ms, uint32_t capacity);
/* the n_ranks + 1 slab edges of the last sharded frame (first -inf, last +inf); *on_lattice = 1 when they lie on planes
 * of the VoxelGrid lattice */
gm_status gm_group_get_edges(const gm_group *grp, double *edges, uint32_t capacity, uint32_t *on_lattice);
/* streaming: a whole frame to the next device in turn, asynchronously (cloud as for gm_submit_frame: host rows are
 * released on return unless GM_CLOUD_PINNED); GM_ERR_NOT_READY when every slot of every device holds a frame */
gm_status gm_group_submit_frame(gm_group *grp, const gm_cloud *cloud);
/* the oldest frame in flight (submission order); *rank / *slot (may be NULL) name where its bulky outputs can be fetched
 * (gm_get_cropped_xyz(gm_group_ctx(grp, rank), slot, ...)) until that slot is submitted to again */
gm_status gm_group_wait_frame(gm_group *grp, gm_frame_result *res, uint32_t *rank, uint32_t *slot);
/* never blocks: GM_OK when the oldest frame in flight has finished (gm_group_wait_frame returns at once),
 * GM_ERR_NOT_READY while it is running or when no frame is in flight */
gm_status gm_group_poll_frame(gm_group *grp);
uint32_t gm_group_in_flight(const gm_group *grp);

#ifdef __cplusplus
}
#endif
#endif /* GM_HIP_H */
