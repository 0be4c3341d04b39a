// gm_internal.hpp -- host-side context and the launch functions each kernel
// file exports.  Nothing here is visible through the C ABI (include/gm_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/gm_hip.h"
#include "gm_device.hpp"

namespace gm {

// How x,y,z are pulled out of sensor_msgs/PointCloud2 rows
// (pcl::fromROSMsg, /root/reference src/geometric_mapping.cpp:55).
struct RowLayout {
    const uint8_t *data;
    // When set, the rows start at *data_at instead (a device word): a captured launch chain reads device-resident rows through
    // it, so that a frame handed over at another address replays the same graph (the pointer travels like the point count).
    const uint8_t *const *data_at;
    uint32_t step, ox, oy, oz;
    uint32_t mode;  // 0: 16-byte rows x,y,z at 0/4/8 (one dwordx4 load)  1: 4-byte aligned  2: byte loads
    uint32_t bswap; // PointCloud2.is_bigendian
};

struct SortScratch {
    uint32_t *totals = nullptr;           // [kRsMaxPasses][2048] digit totals of every pass (zero-filled when a frame opens)
    uint32_t *rec = nullptr;              // [pass][tile][digit] records of the passes' chained scans (k_sort.hip), cleared per sort
    size_t rec_words = 0;
    uint32_t *ticket = nullptr;           // ticket word of the passes (0 between launches); [1]: the tile cutter's
};

// passes x bits of the radix sort for a key width (k_sort.hip); the crop counts the digit totals of exactly this plan
struct SortPlan { int passes, bits; };

// One in-flight frame: its stream, staging and device buffers (grow-only).
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[GM_N_STAGES + 1] = {};
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;  // around the normals kernel alone
    uint32_t cap = 0;          // point capacity of the buffers below
    size_t raw_cap = 0;        // bytes
    uint8_t *h_raw = nullptr;  // pinned staging for the incoming rows
    uint8_t *d_raw = nullptr;
    float4 *crop4 = nullptr;   // cropped cloud: x,y,z, bits(input row)
    uint32_t *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    float4 *spts4 = nullptr;   // cropped cloud in cell-sorted order: x,y,z, bits(cropped index)
    uint32_t *skeys = nullptr; // == whichever of keys_a/keys_b holds the sorted keys
    float4 *normals4 = nullptr;   // per cropped point: nx,ny,nz,curvature (NaN when <3 neighbours)
    int32_t *counts = nullptr;    // per cropped point neighbour count (GM_CFG_KEEP_COUNTS)
    float4 *valid4 = nullptr;     // compacted cloud (finite normals)
    float4 *vnorm4 = nullptr;     // compacted normals
    uint2 *tiles = nullptr;
    uint32_t tiles_cap = 0;   // entries of the tile list's last class (every tile of a frame fits)
    uint32_t tile_seg = 0;    // entries of each of its other kTileListClasses - 1 classes
    uint2 *row_bounds = nullptr;  // [1024*1024] first / one-past-last sorted position per x-row of the search grid
    unsigned long long *blk = nullptr;  // tile records of the single-pass compactions 
    uint32_t blk_cap = 0;
    uint32_t scan_epoch = 0;            // launches of k_compact on this slot so far (see gm_compact.hpp)
    uint32_t scan_seq = 0;              // index of the next k_compact launch inside the frame being captured
    uint32_t frames_enqueued = 0;       // host mirror of the device-side frame counter (frame_in[1]) that replayed scans take their epochs from
    // graph replay (GM_CFG_GRAPH): the frame's launch chain is captured once per (sizes, layout, configuration) and replayed
    bool capturing = false;             // the launches being enqueued go into a stream capture
    bool kernel_timed = false;          // ev_k0 / ev_k1 bracket the last frame's k_normals (not in a replayed frame)
    uint32_t *frame_in = nullptr;       // device: [0] = points of the frame, [2..3] = address of device-resident rows, [4] = frames replayed so far (epochs)
    uint32_t *h_frame_in = nullptr;     // pinned: words [0..3] of frame_in (copied by a node of the graph)
    static constexpr int kGraphs = 4;   // cached captures (a caller that rotates a few device buffers keeps them all)
    hipGraphExec_t graph_exec[kGraphs] = {};
    unsigned char graph_key[kGraphs][512] = {};  // everything the captured launches froze
    uint32_t graph_key_len[kGraphs] = {};
    uint64_t graph_used[kGraphs] = {};  // last use (least recently used is replaced)
    uint64_t graph_clock = 0;
    uint32_t graph_captures = 0;        // launch chains captured on this slot so far (gm_debug_graph_captures: tests)
    uint32_t alloc_gen = 0;             // bumped whenever a device buffer of the slot is (re)allocated
    SortScratch sort = {};
    unsigned long long *tile_rec = nullptr;   // [cutter blocks + 1][kTileListClasses] records of the tile cutter's chained scan
    size_t tile_rec_words = 0;
    uint32_t *seg_start = nullptr;
    float4 *vox4 = nullptr;       // voxel centroids: x,y,z,count
    VoxCell *vox_table = nullptr; // dense voxel table (fast path)
    uint32_t vox_table_cap = 0;   // cells
    int32_t *vox_nn = nullptr;
    double *partials = nullptr;   // [kScatterBlocks][6]  (gm_get_local_frame on caller-supplied normals)
    double *tile_partials = nullptr;  // [compact_blocks(cap)][6]  scatter rows left by the NaN-normal compaction
    uint8_t *labels = nullptr;
    uint8_t *inl_mask = nullptr;   // per valid point: which of the last RANSAC stage's hypotheses it is an inlier of
    DevCounters *ctr = nullptr;
    VoxelParams *voxp = nullptr;
    FrameOut *d_out = nullptr;
    FrameOut *h_out = nullptr;    // pinned
    // extension scratch (allocated on first use)
    uint32_t ext_H = 0, ext_cap = 0;
    float *hyp_plane = nullptr, *hyp_cyl = nullptr;   // [H][8]
    float2 *band = nullptr;                           // [H]
    uint32_t *score_partial = nullptr;                // [score_blocks][H]
    int32_t *cnt_plane = nullptr, *cnt_cyl = nullptr; // [H]
    uint32_t *best_plane = nullptr, *best_cyl = nullptr; // [2]
    double *mom_partial = nullptr;                    // [kScatterBlocks][16]
    double *mom_plane = nullptr, *mom_cyl = nullptr;  // [16]
    unsigned long long *nn_best = nullptr;            // [cap]
    float4 *vox_nrm4 = nullptr;                       // [cap] normal of each voxel centroid's nearest point (GM_CFG_NEAREST)
    // /choppedCloud output (gm_set_cloud_output): caller-owned page-locked rows, copied on a stream of their own
    float4 *cloud_out = nullptr;
    float4 *cloud_out_dev = nullptr;   // the same rows as the device sees them (mapped page-locked memory)
    uint32_t cloud_out_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_crop = nullptr, ev_valid = nullptr, ev_copied = nullptr;
    // state
    bool submitted = false, complete = false;
    bool pipelined = false;      // the context keeps several frames in flight (n_slots > 1): kernels choose block shapes that share the chip
    bool vox_sort_path = false;  // this frame's voxels came from the sort path (may report passthrough)
    uint32_t n_in = 0;
    gm_frame_result last = {};
};

// record array + ticket word + a fresh epoch for the next k_compact launch on this slot's stream
inline ScanState next_scan(Slot &sl)
{
    ScanState st;
    st.status = sl.blk;
    if (sl.capturing) {   // (replayed launches derive their epoch on the device)
        st.epoch = sl.scan_seq++ & 31u;
        st.frame_ptr = sl.frame_in + 4;
    } else {
        // 1 .. 2^29-2, never 0.  When the counter wraps, the records are cleared (on the slot's stream, behind every
        // launch that wrote them): a record left at a tile index that no launch of the last 2^29 has reached would
        // otherwise carry the epoch that is about to be reused and read as ready.
        if (sl.scan_epoch >= 0x1FFFFFFEu) {
            (void)hipMemsetAsync(sl.blk, 0, sizeof(unsigned long long) * (size_t)sl.blk_cap, sl.stream);
            if (sl.tile_rec) (void)hipMemsetAsync(sl.tile_rec, 0, sizeof(unsigned long long) * sl.tile_rec_words, sl.stream);
            sl.scan_epoch = 0;
        }
        sl.scan_epoch += 1u;
        st.epoch = sl.scan_epoch;
        st.frame_ptr = nullptr;
    }
    st.ticket = reinterpret_cast<uint32_t *>(sl.blk + sl.blk_cap);
    return st;
}

}  // namespace gm

struct gm_ctx {
    gm_config cfg;
    int device = 0;
    uint32_t n_slots = 1;
    gm::Slot *slots = nullptr;
    double own_lo, own_hi;
    bool force_voxel_sort = false;
    std::string err;
};

namespace gm {

constexpr int kScatterBlocks = 1024;

// ---- launchers (each enqueues on `s`, never synchronises) --------------------

// k_crop.hip
// count_digits: the emit step also counts the digit totals of the cell sort's passes into sl.sort.totals (which the
// caller has zeroed on this stream)
void launch_crop(const RowLayout &rows, uint32_t n, float lo, float hi, const GridParams &g, Slot &sl, hipStream_t s,
                 uint32_t n_size = 0, const uint32_t *n_dev = nullptr, bool count_digits = false);
// k_sort.hip : stable LSD radix sort of (key, index) pairs, one launch per pass; n is device-resident.
// Returns 0 if the sorted keys (and values) end in (keys_a, vals_a), 1 if in (keys_b, vals_b).
size_t radix_totals_bytes();
size_t radix_record_words(uint32_t n_cap, int key_bits);   // record words a sort of n_cap positions uses (cleared before it)
size_t radix_record_words_max(uint32_t n_cap);
SortPlan radix_plan(int key_bits);
int cell_key_bits(const GridParams &g);
int launch_radix_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b,
                      const uint32_t *n_ptr, uint32_t n_cap, int key_bits, Slot &sl, bool prepared,
                      hipStream_t s, const float4 *rows_in = nullptr, float4 *rows_out = nullptr);
// k_normals.hip
// scratch_cleared: the frame's opening zero-fill cleared the per-row table AND the crop counted the sort's digit totals
void launch_grid_and_normals(const GridParams &g, const VoxDense &vd, Slot &sl, uint32_t n_cap, bool keep_counts,
                             bool scratch_cleared, hipStream_t s);
// one launch that zero-fills up to four 8-byte-granular regions (the frame's counters and scratch tables)
struct ZeroJobs { void *ptr[6]; uint64_t words8[6]; uint32_t *frame_counter; };   // (+1 on the counter: one frame more)
void launch_zero_fill(const ZeroJobs &jobs, hipStream_t s);
uint32_t max_tiles(uint32_t n_cap, const GridParams &g);
uint32_t tile_cutter_blocks(uint32_t n_cap);   // blocks (= chained-scan records per class) of k_rows_and_tiles
// k_frame.hip
// NaN-normal compaction; also leaves the scatter-matrix partial rows of the survivors in sl.tile_partials (one per
// kCpTile cropped points); returns the number of rows launched
uint32_t launch_compact_valid(Slot &sl, uint32_t n_cap, double weightingFactor, hipStream_t s, uint32_t *row_tile = nullptr);
// scatter partials over vnorm4[0..n); returns the number of partial rows written
uint32_t launch_scatter_partials(const float4 *vnorm4, const uint32_t *n_ptr, uint32_t n_cap, double wf, Slot &sl,
                                 hipStream_t s);
void launch_rows_to_host(const float4 *src, float4 *dst_mapped, const uint32_t *begin_enc, const uint32_t *end_ptr, hipStream_t s);
void launch_frame_finalize(const double *partials, uint32_t n_partials, uint32_t row_tile, Slot &sl, hipStream_t s);
// k_voxel.hip
void launch_voxel_grid(Slot &sl, uint32_t n_cap, float leaf, int key_bits, hipStream_t s);
void launch_voxel_dense_finalize(const VoxDense &vd, Slot &sl, hipStream_t s);
constexpr uint32_t kVoxDenseMaxCells = 1u << 18;
// k_ransac.hip (extensions)
constexpr uint32_t kMaxHypotheses = 8192;
uint32_t score_blocks(uint32_t n_cap);
void launch_plane_hypotheses(const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr,
                             uint32_t n_host, uint64_t seed, uint32_t H, float *hyp8, int32_t *zero_counts,
                             hipStream_t s);
void launch_cylinder_hypotheses(const float4 *pts, const float4 *nrm, const uint8_t *labels, uint32_t want,
                                const uint32_t *n_ptr, uint32_t n_host, uint64_t seed, uint32_t H, float *hyp8,
                                int32_t *zero_counts, float2 *band, double tau, hipStream_t s);
void launch_score(int model, const float4 *pts, const uint8_t *labels, uint32_t want, const uint32_t *n_ptr,
                  uint32_t n_cap, const float *hyp8, float2 *band, uint32_t H, double tau, uint32_t *partial,
                  int32_t *counts, uint32_t *best, hipStream_t s);
bool launch_score_preemptive(int model, const float4 *pts, const uint8_t *labels, uint32_t want,
                             const uint32_t *n_ptr, uint32_t n_cap, const float *hyp8, float2 *band, uint32_t H,
                             double tau, uint32_t *scratch, int32_t *counts, uint32_t *best, bool prepared,
                             const uint32_t **sel_out, const int32_t **cnt_out, uint32_t *k_out, hipStream_t s,
                             uint8_t *masks = nullptr, bool *masks_written = nullptr, bool *replicated_counts = nullptr);
                             // masks != nullptr: the last stage streams (k_score_stream; its counters are then kept in copies) and may leave inlier masks
uint32_t launch_label(int model, const float4 *pts, uint8_t *labels, uint32_t want, uint32_t label, const uint32_t *n_ptr,
                      uint32_t n_cap, const float *hyp8, const float2 *band, uint32_t *best, double tau, int init,
                      const uint32_t *sel, const int32_t *counts_k, uint32_t K, hipStream_t s,
                      const float4 *nrm = nullptr, double *mom_partial = nullptr,
                      const uint8_t *masks = nullptr, bool replicated_counts = false);  // returns the grid size (= partial rows); masks: of the K hypotheses in sel
void launch_segment_moments(const float4 *pts, const float4 *nrm, const uint8_t *labels, uint32_t label,
                            const uint32_t *n_ptr, uint32_t n_cap, double *partial, double *mom16, hipStream_t s);
void launch_ext_finalize(const float *hyp_plane, const uint32_t *best_plane, const float *hyp_cyl,
                         const uint32_t *best_cyl, double *mom_plane, double *mom_cyl, FrameExt *ext,
                         const double *partial32, uint32_t mom_rows, hipStream_t s,
                         const double *scatter_partials = nullptr, uint32_t scatter_rows = 0, uint32_t row_tile = 0,
                         const DevCounters *ctr = nullptr, const VoxelParams *voxp = nullptr, FrameOut *frame_out = nullptr);
gm_status gm_enqueue_ransac(gm_ctx *ctx, Slot &sl, uint32_t n_cap, uint32_t scatter_rows, uint32_t row_tile);
// k_nearest.hip
void launch_nearest(const float4 *pts, const uint32_t *n_ptr, uint32_t n_cap, const float4 *queries,
                    const uint32_t *nq_ptr, uint32_t nq_cap, unsigned long long *best, int32_t *idx, hipStream_t s,
                    const float4 *attr = nullptr, float4 *attr_out = nullptr);  // attr_out[q] = attr[idx[q]]

// gm_api.hip helpers shared with gm_ext.hip
gm_status gm_fail(gm_ctx *ctx, gm_status st, const char *msg);
VoxDense gm_make_vox_dense(const gm_ctx *ctx, uint32_t n_cap);   // the dense voxel table a frame of this context uses (enabled = 0: sort path)
gm_status gm_ensure_capacity(gm_ctx *ctx, Slot &sl, uint32_t n, size_t raw_bytes, bool need_raw);
gm_status gm_ensure_ext(gm_ctx *ctx, Slot &sl, uint32_t H);
gm_status gm_begin_stage(gm_ctx *ctx, Slot *&sl);
gm_status gm_check_slot(gm_ctx *ctx, uint32_t slot);
void launch_minmax(const float4 *pts, const uint32_t *n_ptr, uint32_t n_cap, DevCounters *ctr, hipStream_t s);

}  // namespace gm
