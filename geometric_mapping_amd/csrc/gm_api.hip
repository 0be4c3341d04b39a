// gm_api.hip -- the C ABI of include/gm_hip.h: context, slots, the per-frame
// pipeline and the stage-level entry points.  Host logic only; kernels live in
// the k_*.hip files.  No exception crosses the ABI.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <limits>
#include <new>

#include "gm_compact.hpp"
#include "gm_internal.hpp"

using namespace gm;

namespace {

thread_local std::string g_create_err;

#define GM_HIP(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);              \
            return (e__ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_DEVICE;             \
        }                                                                                 \
    } while (0)

gm_status fail(gm_ctx *ctx, gm_status st, const char *msg)
{
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return st;
}

template <class T>
hipError_t dmalloc(T *&p, size_t count)
{
    p = nullptr;
    return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
}

void free_slot_buffers(Slot &sl)
{
    hipFree(sl.d_raw); hipFree(sl.crop4); hipFree(sl.keys_a); hipFree(sl.keys_b); hipFree(sl.vals_a);
    hipFree(sl.vals_b); hipFree(sl.spts4); hipFree(sl.normals4); hipFree(sl.counts); hipFree(sl.valid4);
    hipFree(sl.vnorm4); hipFree(sl.tiles); hipFree(sl.row_bounds); hipFree(sl.blk); hipFree(sl.tile_partials); hipFree(sl.sort.totals); hipFree(sl.sort.rec);
    hipFree(sl.sort.ticket); hipFree(sl.tile_rec); hipFree(sl.seg_start);
    hipFree(sl.vox4); hipFree(sl.vox_nn); hipFree(sl.labels); hipFree(sl.inl_mask);
    if (sl.h_raw) hipHostFree(sl.h_raw);
    sl.d_raw = nullptr; sl.h_raw = nullptr; sl.crop4 = nullptr; sl.keys_a = sl.keys_b = sl.vals_a = sl.vals_b = nullptr;
    sl.spts4 = sl.normals4 = sl.valid4 = sl.vnorm4 = sl.vox4 = nullptr; sl.counts = nullptr; sl.tiles = nullptr; sl.row_bounds = nullptr;
    sl.blk = nullptr; sl.tile_partials = nullptr; sl.sort = SortScratch{}; sl.tile_rec = nullptr; sl.tile_rec_words = 0; sl.seg_start = nullptr; sl.vox_nn = nullptr; sl.labels = nullptr; sl.inl_mask = nullptr;
    sl.cap = 0; sl.raw_cap = 0; sl.tiles_cap = 0; sl.tile_seg = 0;
}

// search grid over the crop box: cell edge >= 1.001 r in y and z (<= 1024 cells per axis), x binned
// `fine` times finer (a power of two chosen so that the cell key still fits 31 bits)
// The neighbour predicate of k_normals needs one ulp of r^2, scaled by a power of two <= 2^126, to reach 1: true for
// every radius >= 1e-15 m (and the radius 0, "no neighbours", is handled exactly as well).
static bool radius_ok(double r) { return std::isfinite(r) && (r == 0.0 || (r >= 1e-15 && r <= 1e15)); }

// Rows per radius.  Finer y/z rows cut the candidate stream when a tile (64 consecutive points of one row) is short
// against the radius -- many neighbours per point -- and lengthen it otherwise (tools/sim_normals_grid.py: at k ~ 256 the
// streamed pair slots grow 14 % with D = 2, at k ~ 5 100 they shrink 19 / 27 / 31 % with D = 2 / 3 / 4).  The neighbour
// count is not known before the frame is processed; the estimate is the count a uniform fill of the box would give,
// times ten (a surface scan concentrates its points).  GM_NORMALS_ROWS=1..4
// overrides; only the matrix-core kernel knows D > 1.
static int rows_per_radius(double radius, float ex, float ey, float ez, uint32_t n_points)
{
    const char *e = getenv("GM_NORMALS_ROWS");          // (read per context: tests switch it between cases)
    static const char *impl = getenv("GM_NORMALS_IMPL");
    if (impl && !(impl[0] == 'a' || impl[0] == 'm') ) return 1;
    if (impl && strchr(impl, '0')) return 1;
    if (e) { const int d = atoi(e); return d < 1 ? 1 : (d > 4 ? 4 : d); }
    const double vol = (double)ex * ey * ez;
    if (!(vol > 0.0) || n_points == 0) return 1;
    const double k_est = 10.0 * (double)n_points * (4.18879 * radius * radius * radius) / vol;
    // measured on the 1 M-point tunnel frame (tools/sweep_rows.sh): r = 0.25 (estimate 650, true k 1 280): D = 1 / 2 / 3 / 4
    // = 0.446 / 0.422 / 0.477 / 0.536 ms; r = 0.5 (estimate 5 200, true k 5 100): 1.47 / 1.24 / 1.25 / 1.21 ms.
    // Round 4, on a lidar-shaped frame (synth.velodyne_tunnel, 64 rings x 1 800, r = 0.5: estimate 600, k from ~100 to 7 600
    // along a row, median 2 350; tools/lidar_probe.py): D = 1 / 2 / 3 / 4 = 0.213 / 0.242 / 0.240 / 0.279 ms -- the estimate
    // is weighted by points (dense patches beside the sensor), a tile's length by rows, and most rows of such a frame are
    // thin: their 64-point tiles are several radii long, where finer rows only lengthen the candidate stream.  The band
    // in which D = 2 was chosen (estimate 400 .. 2 000) bought 5 % on the uniform tunnel and cost 13 % on the lidar frame --
    // the reference's actual input (launch/mapping.launch:22) -- so it is gone: D = 1 below 2 000, D = 4 from there.
    return k_est < 2000.0 ? 1 : 4;
}

GridParams make_grid(float lo_x, float lo_y, float lo_z, float ex, float ey, float ez, double radius, uint32_t n_points)
{
    GridParams g;
    memset(&g, 0, sizeof(g));   // (also the padding: the struct is compared bytewise as part of a graph key)
    float ext = fmaxf(ex, fmaxf(ey, ez));
    float hr = (float)radius * 1.001f;     // the radius-sized edge: x binning and the error band are relative to it
    if (!(hr > 1e-9f)) hr = 1e-9f;
    if (hr < ext / 1023.0f) hr = ext / 1023.0f;
    int D = rows_per_radius(radius, ex, ey, ez, n_points);
    while (D > 1 && hr / (float)D < ext / 1023.0f) --D;   // <= 1024 rows per axis
    const float h = hr / (float)D;         // y/z cell edge
    g.D = D;
    g.ox = lo_x; g.oy = lo_y; g.oz = lo_z;
    g.inv_h = 1.0f / h;
    auto dim = [&](float e, float inv, int cap) { int n = (int)floorf(e * inv) + 1; return n < 1 ? 1 : (n > cap ? cap : n); };
    g.ny = dim(ey, g.inv_h, 1024); g.nz = dim(ez, g.inv_h, 1024);
    const float inv_hr = 1.0f / hr;
    const int nxc = dim(ex, inv_hr, 1024);
    // x is binned 64x finer than y/z (measured on the 1 M frame: 8x 0.370 ms, 16x 0.345, 32x 0.337, 64x 0.328,
    // 128x 0.329 for k_normals; beyond that the extra key bits cost the sort more than the windows gain)
    int fine = 64;
    while (fine > 1 && (uint64_t)g.ny * g.nz * (uint64_t)(nxc * fine + fine) >= (1ull << 31)) fine >>= 1;
    g.inv_hx = inv_hr * (float)fine;
    g.nx = dim(ex, g.inv_hx, 1024 * fine);
    g.xreach = fine + 1;
    // per-row window half-widths.  A candidate in the row (a, b) cells away from the query's row is further than
    // gap = (|a| - 1) h resp. (|b| - 1) h from it in y resp. z, so |dx| < sqrt(r^2 - gap_y^2 - gap_z^2); two fine cells
    // whose points are closer than w in x differ by at most floor(w * inv_hx) + 1 in their index (+1 spare, as before).
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            g.reach[a][b] = 0;
            if (a > D || b > D) continue;
            const double gy = a > 1 ? (a - 1) * (double)h : 0.0, gz = b > 1 ? (b - 1) * (double)h : 0.0;
            const double w2 = radius * radius - gy * gy - gz * gz;
            if (w2 <= 0.0) continue;
            int rc = (int)floor(sqrt(w2) * (double)g.inv_hx * (1.0 + 1e-6)) + 2;
            if (rc > g.xreach) rc = g.xreach;
            g.reach[a][b] = (int16_t)rc;
        }
    g.r2 = (float)(radius * radius);  // KdTreeFLANN::radiusSearch: static_cast<float>(radius*radius)
    {
        int e = 0;
        (void)frexpf(g.r2 > 0.f ? g.r2 : 1.f, &e);  // r2 = m * 2^e, m in [0.5, 1)
        int k = 100 - e;
        k = k > 126 ? 126 : (k < -100 ? -100 : k);  // never clamps for the radii radius_ok() admits
        g.r2_scale = ldexpf(1.0f, k);
    }
    {
        const float cmax = fmaxf(fmaxf(fabsf(lo_x), fabsf(lo_x + ex)), fmaxf(fmaxf(fabsf(lo_y), fabsf(lo_y + ey)), fmaxf(fabsf(lo_z), fabsf(lo_z + ez))));
        int e = 0;
        (void)frexpf(cmax > 1e-30f ? cmax : 1e-30f, &e);   // cmax = m * 2^e, m in [0.5, 1): ulp(cmax) = 2^(e - 24)
        g.snap = ldexpf(1.0f, e - 24 < -120 ? -120 : e - 24);
        static const char *bs = getenv("GM_MX_BAND_SCALE");   // experiments only
        g.band = 2.0e-5f * hr * hr * (bs ? (float)atof(bs) : 1.0f);
        int be = 0;
        (void)frexpf(1.0f / g.band, &be);                    // 1/band = m * 2^be, m in [0.5, 1)  =>  2^be >= 1/band
        g.dscale = ldexpf(1.0f, be > 60 ? 60 : be);
        g.dband = g.band * g.dscale;
    }
    return g;
}

// dense voxel table over the crop box, when it is small enough (see VoxDense)
VoxDense make_vox_dense(float lo, float hi, double leaf, uint32_t n_cap, double own_lo, double own_hi)
{
    VoxDense vd;
    memset(&vd, 0, sizeof(vd));
    vd.inv_leaf = 1.0f / (float)leaf;
    vd.lo = lo;
    vd.own_lo = (float)own_lo;
    vd.own_hi = (float)own_hi;
    const double a = floor((double)(lo * vd.inv_leaf)), b = floor((double)(hi * vd.inv_leaf));
    const double dim = b - a + 1.0;
    if (std::isfinite(dim) && dim >= 1.0 && dim * dim * dim <= (double)kVoxDenseMaxCells && fabs(a) < 1e9) {
        vd.enabled = 1;
        vd.i_lo = (int32_t)a;
        vd.dim = (int32_t)dim;
        // power-of-two scale such that 2^28 points cannot overflow 63 bits; independent of the
        // buffer capacity so that results do not depend on allocation history
        const double ext = fmax((double)hi - (double)lo, 1e-30);
        int k = (int)floor(log2(9.0e18 / (ext * 268435456.0)));
        if (k > 40) k = 40;
        if (k < 8 || n_cap > 268435456u) { vd.enabled = 0; k = 0; }
        vd.scale = ldexp(1.0, k);
        vd.inv_scale = ldexp(1.0, -k);
    }
    return vd;
}

int bits_for(uint64_t count)
{
    int b = 1;
    while ((1ull << b) < count && b < 32) ++b;
    return b;
}

// upper bound of VoxelGrid's dx*dy*dz for points inside [lo,hi]^3 -> sort key bits
int voxel_key_bits(double lo, double hi, double leaf)
{
    const float inv = 1.0f / (float)leaf;
    double d = floor((double)((float)hi * inv)) - floor((double)((float)lo * inv)) + 2.0;
    if (!(d > 0) || !std::isfinite(d)) return 32;
    double prod = d * d * d;
    if (prod > 2147483647.0) return 32;  // may hit PCL's overflow guard: keys become row indices
    return bits_for((uint64_t)prod);
}

gm_status ensure_capacity(gm_ctx *ctx, Slot &sl, uint32_t n, size_t raw_bytes, bool need_raw)
{
    if (need_raw && raw_bytes > sl.raw_cap) {
        if (sl.h_raw) hipHostFree(sl.h_raw);
        hipFree(sl.d_raw);
        sl.h_raw = nullptr; sl.d_raw = nullptr;
        size_t cap = raw_bytes + raw_bytes / 4 + 4096;
        GM_HIP(ctx, hipHostMalloc((void **)&sl.h_raw, cap, hipHostMallocDefault));
        GM_HIP(ctx, hipMalloc((void **)&sl.d_raw, cap));
        sl.raw_cap = cap;
    }
    if (n <= sl.cap) return GM_OK;
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    uint8_t *h_raw = sl.h_raw, *d_raw = sl.d_raw; size_t raw_cap = sl.raw_cap;
    sl.h_raw = nullptr; sl.d_raw = nullptr;
    free_slot_buffers(sl);
    sl.h_raw = h_raw; sl.d_raw = d_raw; sl.raw_cap = raw_cap;
    uint32_t cap = n + n / 4 + 1024;
    if (cap < ctx->cfg.max_points) cap = ctx->cfg.max_points;
    cap = (cap + 3u) & ~3u;  // whole 16 B groups of keys (k_rows_and_tiles reads them as uint4)
    GM_HIP(ctx, dmalloc(sl.crop4, cap));
    GM_HIP(ctx, dmalloc(sl.keys_a, cap)); GM_HIP(ctx, dmalloc(sl.keys_b, cap));
    GM_HIP(ctx, dmalloc(sl.vals_a, cap)); GM_HIP(ctx, dmalloc(sl.vals_b, cap));
    GM_HIP(ctx, dmalloc(sl.spts4, cap)); GM_HIP(ctx, dmalloc(sl.normals4, cap));
    GM_HIP(ctx, dmalloc(sl.counts, cap));
    GM_HIP(ctx, dmalloc(sl.valid4, cap)); GM_HIP(ctx, dmalloc(sl.vnorm4, cap));
    GM_HIP(ctx, dmalloc(sl.seg_start, cap)); GM_HIP(ctx, dmalloc(sl.vox4, cap));
    GM_HIP(ctx, dmalloc(sl.vox_nn, cap)); GM_HIP(ctx, dmalloc(sl.labels, cap)); GM_HIP(ctx, dmalloc(sl.inl_mask, cap));
    // tile list: kTileListClasses - 1 segments for the tiles with an x extent (>= 2 points each: none of them can hold more
    // than cap / 2 tiles) + one that holds every tile a frame can have (>= 1 point each).  No list can overflow.
    sl.tiles_cap = cap + 2u;
    sl.tile_seg = cap / 2u + 2u;
    GM_HIP(ctx, dmalloc(sl.tiles, (size_t)(kTileListClasses - 1) * sl.tile_seg + sl.tiles_cap));
    GM_HIP(ctx, dmalloc(sl.row_bounds, (size_t)1024 * 1024));  // make_grid caps every axis at 1024 cells
    sl.blk_cap = compact_records(cap > kVoxDenseMaxCells ? cap : kVoxDenseMaxCells) + 1;
    GM_HIP(ctx, dmalloc(sl.tile_partials, (size_t)compact_blocks(cap) * 6));
    GM_HIP(ctx, dmalloc(sl.blk, (size_t)sl.blk_cap + 1));  // + the ticket word
    GM_HIP(ctx, hipMemsetAsync(sl.blk, 0, sizeof(unsigned long long) * ((size_t)sl.blk_cap + 1), sl.stream));
    {
        // tests only: start the chained scans' epoch counter just below its wrap (tests/test_gpu_parity.py)
        static const char *e = getenv("GM_TEST_SCAN_EPOCH");
        sl.scan_epoch = e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
    }
    // radix sort scratch: digit totals, the passes' record array (both cleared per sort) and the ticket words (zeroed once)
    GM_HIP(ctx, dmalloc(sl.sort.totals, radix_totals_bytes() / sizeof(uint32_t)));
    sl.sort.rec_words = (radix_record_words_max(cap) + 1u) & ~(size_t)1u;
    GM_HIP(ctx, dmalloc(sl.sort.rec, sl.sort.rec_words));
    GM_HIP(ctx, dmalloc(sl.sort.ticket, 4));
    GM_HIP(ctx, hipMemsetAsync(sl.sort.ticket, 0, 16, sl.stream));
    sl.tile_rec_words = (size_t)(tile_cutter_blocks(cap) + 1) * (size_t)kTileListClasses;
    GM_HIP(ctx, dmalloc(sl.tile_rec, sl.tile_rec_words));
    GM_HIP(ctx, hipMemsetAsync(sl.tile_rec, 0, sizeof(unsigned long long) * sl.tile_rec_words, sl.stream));
    sl.cap = cap;
    ++sl.alloc_gen;
    return GM_OK;
}

// x/y/z must lie inside a row; 64-bit arithmetic so that an offset near 2^32 cannot wrap past the check.  Called
// before anything of the frame is enqueued (no DMA may be in flight from the caller's buffer on an error return).
gm_status check_layout(gm_ctx *ctx, const gm_cloud *c)
{
    const uint64_t step = c->point_step;
    if (c->n_points && (step < 12 || (uint64_t)c->off_x + 4 > step || (uint64_t)c->off_y + 4 > step ||
                        (uint64_t)c->off_z + 4 > step))
        return fail(ctx, GM_ERR_INVALID_ARG, "gm_cloud: x/y/z offsets do not fit in point_step");
    return GM_OK;
}

gm_status make_rows(gm_ctx *ctx, const gm_cloud *c, const uint8_t *dev_data, RowLayout &rows)
{
    const gm_status lst = check_layout(ctx, c);
    if (lst != GM_OK) return lst;
    memset(&rows, 0, sizeof(rows));   // (also the padding: compared bytewise as part of a graph key)
    rows.data = dev_data;
    rows.step = c->point_step; rows.ox = c->off_x; rows.oy = c->off_y; rows.oz = c->off_z;
    rows.bswap = (c->flags & GM_CLOUD_BIGENDIAN) ? 1u : 0u;
    const bool al4 = ((c->point_step | c->off_x | c->off_y | c->off_z) & 3u) == 0 && (((uintptr_t)dev_data) & 3u) == 0;
    const bool al16 = al4 && c->point_step == 16 && c->off_x == 0 && c->off_y == 4 && c->off_z == 8 &&
                      (((uintptr_t)dev_data) & 15u) == 0;
    rows.mode = al16 ? 0u : (al4 ? 1u : 2u);
    return GM_OK;
}

void record(gm_ctx *ctx, Slot &sl, int idx)
{
    if (ctx->cfg.flags & GM_CFG_STAGE_TIMING) hipEventRecord(sl.ev[idx], sl.stream);
}

gm_status reset_counters(gm_ctx *ctx, Slot &sl)
{
    GM_HIP(ctx, hipMemsetAsync(sl.ctr, 0, sizeof(DevCounters), sl.stream));
    return GM_OK;
}

// Host rows -> device.  Page-locked input (GM_CLOUD_PINNED) is one DMA.  Pageable input of a BLOCKING call
// (gm_process_frame, the stage calls: the caller cannot touch its buffer before we return anyway) is handed to the
// runtime as it is -- ROCm pins it on the fly: 0.23 ms for a 12 MB frame, the rate of page-locked input.  Pageable
// input of gm_submit_frame must be released on return, so it goes through the slot's pinned staging buffer in 2 MiB
// pieces: the calling thread copies piece k+1 while the DMA of piece k is in flight (0.31 ms instead of 0.47).
static hipError_t upload_rows(Slot &sl, const gm_cloud *cloud, size_t raw_bytes, bool blocking_call, hipStream_t s)
{
    if ((cloud->flags & GM_CLOUD_PINNED) || blocking_call)
        return hipMemcpyAsync(sl.d_raw, cloud->data, raw_bytes, hipMemcpyHostToDevice, s);
    const size_t piece = (size_t)2 << 20;
    const uint8_t *src = (const uint8_t *)cloud->data;
    for (size_t off = 0; off < raw_bytes; off += piece) {
        const size_t len = raw_bytes - off < piece ? raw_bytes - off : piece;
        memcpy(sl.h_raw + off, src + off, len);
        const hipError_t e = hipMemcpyAsync(sl.d_raw + off, sl.h_raw + off, len, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Launch sizes follow a bucketed point count (at most 1/16 above the frame's own): frames of about the same size then
// share their launch geometry -- the condition for replaying a captured graph -- and kernels read the true counts from
// device memory as they always did.
static uint32_t size_bucket(uint32_t n)
{
    if (n == 0) return 0;
    uint32_t p2 = 1024;
    while (p2 < n && p2 < 0x80000000u) p2 <<= 1;
    const uint32_t step = p2 / 16 < 1024u ? 1024u : p2 / 16;
    const uint64_t b = ((uint64_t)n + step - 1) / step * step;
    return b > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b;
}

// Everything between the upload of the rows and the download of the result record: the launches of one frame.  With
// sl.capturing set they go into a stream capture: no events, the point count comes from device memory (n_dev).
static gm_status enqueue_launches(gm_ctx *ctx, Slot &sl, const RowLayout &rows, uint32_t n, uint32_t ns, const GridParams &g,
                                  const VoxDense &vd, float lo, float hi)
{
    const gm_config &cf = ctx->cfg;
    hipStream_t s = sl.stream;
    const uint32_t *n_dev = nullptr;
    if (sl.capturing) {
        sl.scan_seq = 0;
        // the frame's point count -- and the address of device-resident rows -- travel through pinned words (the node's
        // addresses are fixed, the values are not)
        GM_HIP(ctx, hipMemcpyAsync(sl.frame_in, sl.h_frame_in, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        n_dev = sl.frame_in;
    }
    {
        // one zero-fill launch opens the frame: device counters, dense voxel table, per-row table of the search
        // grid, digit totals of the sort (all only touched later in this frame, on this stream)
        ZeroJobs z;
        memset(&z, 0, sizeof(z));
        z.ptr[0] = sl.ctr; z.words8[0] = sizeof(DevCounters) / 8;
        if (vd.enabled) { z.ptr[1] = sl.vox_table; z.words8[1] = (uint64_t)vd.dim * vd.dim * vd.dim * sizeof(VoxCell) / 8; }
        if (sl.row_bounds) { z.ptr[2] = sl.row_bounds; z.words8[2] = (uint64_t)g.ny * (uint64_t)g.nz; }  // (no buffers yet
        if (sl.sort.totals) {                                                                            //  before the first non-empty frame)
            z.ptr[3] = sl.sort.totals; z.words8[3] = radix_totals_bytes() / 8;
            z.ptr[4] = sl.sort.rec; z.words8[4] = (radix_record_words(ns, cell_key_bits(g)) + 1) / 2;   // records of the cell sort's chained scans
        }
        z.frame_counter = sl.frame_in + 4;
        launch_zero_fill(z, s);
    }
    launch_crop(rows, n, lo, hi, g, sl, s, ns, n_dev, true);   // (also counts the digit totals of the cell sort)
    // /choppedCloud to the caller's page-locked rows (gm_set_cloud_output), on a stream of its own, in two steps.  Up to the
    // first point that loses its normal the valid cloud IS the cropped cloud: the cropped rows leave right here, with
    // nearly the whole frame still ahead of them to hide the copy behind; what the NaN-normal compaction moved is sent
    // again below (nothing at all on a dense frame).  The first copy is a DMA of the frame's n rows (>= n_cropped, whose
    // value only the device knows; a kernel writing the mapped rows with the exact count was measured: 64 blocks of
    // stores over PCIe take 0.6 ms for 13 MB where the DMA engine takes 0.25); the second is a kernel that reads the
    // counts on the device and writes the mapped host rows.  (Inside a captured graph: one in-line copy further down.)
    const bool cloud_overlap = sl.cloud_out && n && !sl.capturing && sl.cloud_out_dev;
    if (cloud_overlap) {
        GM_HIP(ctx, hipEventRecord(sl.ev_crop, s));
        GM_HIP(ctx, hipStreamWaitEvent(sl.copy_stream, sl.ev_crop, 0));
        GM_HIP(ctx, hipMemcpyAsync(sl.cloud_out, sl.crop4, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, sl.copy_stream));
    }
    record(ctx, sl, 2);
    launch_grid_and_normals(g, vd, sl, ns, (cf.flags & GM_CFG_KEEP_COUNTS) != 0, true, s);
    record(ctx, sl, 3);  // end of grid+normals; the kernel alone is bracketed by ev_k0/ev_k1
    // (getLocalFrame's scatter terms are summed by the compaction: one partial row per kCpTile cropped points)
    uint32_t row_tile = kCpTile;
    const uint32_t nparts = launch_compact_valid(sl, ns, cf.weightingFactor, s, &row_tile);
    bool cloud_copy_pending = false;
    if (cloud_overlap) {
        GM_HIP(ctx, hipEventRecord(sl.ev_valid, s));
        GM_HIP(ctx, hipStreamWaitEvent(sl.copy_stream, sl.ev_valid, 0));
        launch_rows_to_host(sl.valid4, sl.cloud_out_dev, &sl.ctr->first_drop_enc, &sl.ctr->n_valid, sl.copy_stream);
        GM_HIP(ctx, hipEventRecord(sl.ev_copied, sl.copy_stream));
        cloud_copy_pending = true;
    } else if (sl.cloud_out && n) {
        // (a captured copy's size is frozen: the bucketed size, as far as the buffer has rows -- both >= n_valid)
        const size_t frozen = (size_t)(ns < sl.cloud_out_cap ? ns : sl.cloud_out_cap) * sizeof(float4);
        GM_HIP(ctx, hipMemcpyAsync(sl.cloud_out, sl.valid4, frozen, hipMemcpyDeviceToHost, s));
    }
    record(ctx, sl, 4);
    record(ctx, sl, 5);
    sl.vox_sort_path = (cf.flags & GM_CFG_VOXEL_GRID) && !vd.enabled;
    if (cf.flags & GM_CFG_VOXEL_GRID) {
        if (vd.enabled) {
            launch_voxel_dense_finalize(vd, sl, s);
        } else {  // lattice too large for a table: min/max -> keys -> sort -> segmented mean
            launch_minmax(sl.valid4, &sl.ctr->n_valid, ns, sl.ctr, s);
            launch_voxel_grid(sl, ns, (float)cf.voxelGridLeafSize, voxel_key_bits(lo, hi, cf.voxelGridLeafSize), s);
        }
    }
    record(ctx, sl, 6);
    if (cf.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)) {
        const gm_status st = gm_enqueue_ransac(ctx, sl, ns, nparts, row_tile);   // its closing launch also finalizes the frame
        if (st != GM_OK) return st;
    }
    if ((cf.flags & GM_CFG_NEAREST) && (cf.flags & GM_CFG_VOXEL_GRID)) {
        const uint32_t nq_cap = vd.enabled ? (uint32_t)vd.dim * vd.dim * vd.dim : ns;
        launch_nearest(sl.valid4, &sl.ctr->n_valid, ns, sl.vox4, &sl.ctr->n_voxels, nq_cap < ns ? nq_cap : ns, sl.nn_best,
                       sl.vox_nn, s, sl.vnorm4, sl.vox_nrm4);
    }
    record(ctx, sl, 7);
    if (!(cf.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)))
        launch_frame_finalize(sl.tile_partials, nparts, row_tile, sl, s);
    if (cloud_copy_pending) GM_HIP(ctx, hipStreamWaitEvent(s, sl.ev_copied, 0));   // the slot's stream ends behind the cloud copy
    GM_HIP(ctx, hipMemcpyAsync(sl.h_out, sl.d_out, sizeof(FrameOut), hipMemcpyDeviceToHost, s));
    return GM_OK;
}

gm_status enqueue_frame(gm_ctx *ctx, Slot &sl, const gm_cloud *cloud, bool blocking_call)
{
    const uint32_t n = cloud->n_points;
    const uint32_t ns = size_bucket(n);
    const size_t raw_bytes = (size_t)n * cloud->point_step;
    const bool on_dev = (cloud->flags & GM_CLOUD_DEVICE) != 0;
    if (n && !cloud->data) return fail(ctx, GM_ERR_INVALID_ARG, "gm_cloud.data is NULL");
    if (sl.cloud_out && n > sl.cloud_out_cap) return fail(ctx, GM_ERR_CAPACITY, "gm_set_cloud_output: the frame has more points than the cloud buffer has rows");
    gm_status st = check_layout(ctx, cloud);
    if (st != GM_OK) return st;
    // n == 0 still sizes the buffers for one point: every stage below may then assume non-null scratch (an empty
    // cloud as the very first frame of a context used to reach the compaction with a null scratch array)
    st = ensure_capacity(ctx, sl, ns ? ns : 1u, raw_bytes, !on_dev);
    if (st != GM_OK) return st;
    if (ctx->cfg.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER | GM_CFG_NEAREST)) {
        st = gm_ensure_ext(ctx, sl, ctx->cfg.ransac_hypotheses ? ctx->cfg.ransac_hypotheses : 1);
        if (st != GM_OK) return st;
    }
    hipStream_t s = sl.stream;
    sl.n_in = n;
    // Epochs of replayed (graph) scans are (frame counter * 32 + launch index) mod 2^29: they repeat every 2^24 frames of
    // the slot.  Twice per period the records are cleared between two frames (all of them are stale there), so that a
    // record can never be as old as a period.
    if ((++sl.frames_enqueued & 0x7FFFFFu) == 0u && sl.blk) {
        GM_HIP(ctx, hipMemsetAsync(sl.blk, 0, sizeof(unsigned long long) * (size_t)sl.blk_cap, s));
        GM_HIP(ctx, hipMemsetAsync(sl.tile_rec, 0, sizeof(unsigned long long) * sl.tile_rec_words, s));
    }
    record(ctx, sl, 0);
    const uint8_t *dev_rows = (const uint8_t *)cloud->data;
    if (!on_dev && n) {
        // Page-locked rows (gm_host_alloc) are mapped into the device's address space, so the crop kernel COULD read them
        // where they are, over PCIe, instead of waiting for a copy (GM_PINNED_ZERO_COPY=1: blocking calls only, =2: every
        // call).  Measured on the 1 M-point frame: a blocking frame 0.697 -> 0.679 ms, but with frames in flight a crop
        // kernel that waits on PCIe holds the CUs the other frames' kernels want (0.35 -> 0.52 ms per step), and once a
        // kernel has read a page-locked buffer later copies FROM that buffer run a quarter slower (0.35 -> 0.49 ms per
        // step).  Off by default.
        static const char *zc = getenv("GM_PINNED_ZERO_COPY");
        const int zmode = zc ? atoi(zc) : 0;
        void *mapped = nullptr;
        if ((cloud->flags & GM_CLOUD_PINNED) && (zmode == 2 || (zmode == 1 && blocking_call)) &&
            hipHostGetDevicePointer(&mapped, const_cast<void *>(cloud->data), 0) == hipSuccess && mapped) {
            dev_rows = (const uint8_t *)mapped;
        } else {
            (void)hipGetLastError();
            GM_HIP(ctx, upload_rows(sl, cloud, raw_bytes, blocking_call, s));
            dev_rows = sl.d_raw;
        }
    }
    record(ctx, sl, 1);
    RowLayout rows;
    st = make_rows(ctx, cloud, dev_rows, rows);
    if (st != GM_OK) return st;
    const gm_config &cf = ctx->cfg;
    // Eigen::Vector4f(-bound, ...) : double -> float (src/tunnel_processing.cpp:43-44)
    const float lo = (float)(-cf.boxFilterBound), hi = (float)cf.boxFilterBound;
    const float ext = hi - lo;
    const GridParams g = make_grid(lo, lo, lo, ext, ext, ext, cf.neighborRadius, ns);
    VoxDense vd;
    memset(&vd, 0, sizeof(vd));
    vd.own_lo = (float)ctx->own_lo;
    vd.own_hi = (float)ctx->own_hi;
    if ((cf.flags & GM_CFG_VOXEL_GRID) && !ctx->force_voxel_sort)
        vd = make_vox_dense(lo, hi, cf.voxelGridLeafSize, sl.cap, ctx->own_lo, ctx->own_hi);
    if (vd.enabled) {
        const uint32_t cells = (uint32_t)vd.dim * vd.dim * vd.dim;
        if (cells > sl.vox_table_cap) {
            GM_HIP(ctx, hipStreamSynchronize(s));
            hipFree(sl.vox_table);
            sl.vox_table = nullptr;
            GM_HIP(ctx, dmalloc(sl.vox_table, cells));
            sl.vox_table_cap = cells;
            ++sl.alloc_gen;
        }
    }
    // GM_CFG_GRAPH: the launch chain is captured once and replayed for every frame whose bucketed size, row layout,
    // grid and buffers are those of the capture (all counts the kernels work with are device-resident; the one the host
    // knows, the number of points, travels through a pinned word).  Not with stage timing (events between launches).
    const bool graph = (cf.flags & GM_CFG_GRAPH) && !(cf.flags & GM_CFG_STAGE_TIMING) && n > 0;
    if (!graph) {
        st = enqueue_launches(ctx, sl, rows, n, ns, g, vd, lo, hi);
        if (st != GM_OK) return st;
    } else {
        // device-resident rows are read through a device word: their address is not part of what a capture freezes
        // (the row mode, which depends on the address's alignment, is)
        const uint8_t *rows_at = rows.data;
        if (on_dev) { rows.data = nullptr; rows.data_at = reinterpret_cast<const uint8_t *const *>(sl.frame_in + 2); }
        unsigned char key[sizeof(sl.graph_key[0])];
        uint32_t klen = 0;
        auto put = [&](const void *p, size_t len) { memcpy(key + klen, p, len); klen += (uint32_t)len; };
        static_assert(sizeof(RowLayout) + sizeof(GridParams) + sizeof(VoxDense) + 80 <= sizeof(sl.graph_key[0]), "graph key");
        memset(key, 0, sizeof(key));
        put(&ns, 4); put(&sl.alloc_gen, 4); put(&rows, sizeof(rows)); put(&g, sizeof(g)); put(&vd, sizeof(vd));
        put(&cf.flags, 4); put(&ctx->own_lo, 8); put(&ctx->own_hi, 8); put(&sl.cloud_out, sizeof(sl.cloud_out)); put(&sl.cloud_out_cap, 4);
        int gi = -1, victim = -1;   // victim: an empty entry if there is one, else the least recently used
        for (int k = 0; k < Slot::kGraphs; ++k) {
            if (sl.graph_exec[k] && sl.graph_key_len[k] == klen && memcmp(key, sl.graph_key[k], klen) == 0) gi = k;
            if (victim >= 0 && !sl.graph_exec[victim]) continue;
            if (victim < 0 || !sl.graph_exec[k] || sl.graph_used[k] < sl.graph_used[victim]) victim = k;
        }
        if (gi < 0) {
            gi = victim;
            if (sl.graph_exec[gi]) { hipGraphExecDestroy(sl.graph_exec[gi]); sl.graph_exec[gi] = nullptr; }
            hipGraph_t graph_obj = nullptr;
            GM_HIP(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            sl.capturing = true;
            st = enqueue_launches(ctx, sl, rows, n, ns, g, vd, lo, hi);
            sl.capturing = false;
            const hipError_t ce = hipStreamEndCapture(s, &graph_obj);
            if (st != GM_OK) { if (graph_obj) hipGraphDestroy(graph_obj); return st; }
            GM_HIP(ctx, ce);
            const hipError_t ie = hipGraphInstantiate(&sl.graph_exec[gi], graph_obj, nullptr, nullptr, 0);
            hipGraphDestroy(graph_obj);
            GM_HIP(ctx, ie);
            memcpy(sl.graph_key[gi], key, sizeof(key));
            sl.graph_key_len[gi] = klen;
            ++sl.graph_captures;
        }
        sl.graph_used[gi] = ++sl.graph_clock;
        sl.h_frame_in[0] = n;
        memcpy(sl.h_frame_in + 2, &rows_at, sizeof(rows_at));
        GM_HIP(ctx, hipGraphLaunch(sl.graph_exec[gi], s));
    }
    record(ctx, sl, 8);
    GM_HIP(ctx, hipGetLastError());
    sl.kernel_timed = !graph;
    sl.submitted = true;
    sl.complete = false;
    return GM_OK;
}

void fill_result(gm_ctx *ctx, Slot &sl, gm_frame_result *r)
{
    memset(r, 0, sizeof(*r));
    const FrameOut &o = *sl.h_out;
    r->n_in = sl.n_in;
    r->n_cropped = o.ctr.n_cropped;
    r->n_valid = o.ctr.n_valid;
    r->n_voxels = (ctx->cfg.flags & GM_CFG_VOXEL_GRID) ? o.ctr.n_voxels : 0;
    for (int k = 0; k < 3; ++k) r->eigenvalues[k] = o.evals[k];
    for (int k = 0; k < 9; ++k) r->eigenvectors[k] = o.evecs[k];
    for (int k = 0; k < 3; ++k) r->center_axis[k] = o.evecs[k];  // block<3,1>(0,0)
    for (int k = 0; k < 6; ++k) r->scatter[k] = o.scatter[k];
    // (the dense-table path never passes through; voxp may still hold the flag of an earlier gm_voxel_grid stage call)
    if (sl.vox_sort_path && o.vox.passthrough) r->status_flags |= GM_RES_VOXEL_PASSTHROUGH;
    if (ctx->cfg.flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)) {
        r->plane_inliers = o.ext.plane_inliers;
        r->cylinder_inliers = o.ext.cylinder_inliers;
        for (int k = 0; k < 4; ++k) { r->plane[k] = o.ext.plane[k]; r->plane_refit[k] = o.ext.plane_refit[k]; }
        for (int k = 0; k < 7; ++k) r->cylinder[k] = o.ext.cylinder[k];
        for (int k = 0; k < 3; ++k) r->cylinder_axis_refit[k] = o.ext.cyl_axis_refit[k];
    }
    float ms = 0;
    if (sl.n_in && sl.kernel_timed && hipEventElapsedTime(&ms, sl.ev_k0, sl.ev_k1) == hipSuccess) r->normals_kernel_ms = ms;
    if (ctx->cfg.flags & GM_CFG_STAGE_TIMING) {
        static const int stage_of[8] = {GM_STAGE_UPLOAD, GM_STAGE_CROP, GM_STAGE_NORMALS, GM_STAGE_COMPACT,
                                        GM_STAGE_FRAME, GM_STAGE_VOXEL, GM_STAGE_RANSAC, GM_STAGE_FRAME};
        for (int i = 0; i < 8; ++i)
            if (hipEventElapsedTime(&ms, sl.ev[i], sl.ev[i + 1]) == hipSuccess) r->stage_ms[stage_of[i]] += ms;
        if (hipEventElapsedTime(&ms, sl.ev[0], sl.ev[8]) == hipSuccess) r->stage_ms[GM_STAGE_TOTAL] = ms;
        r->stage_ms[GM_STAGE_GRID] = r->stage_ms[GM_STAGE_NORMALS] - r->normals_kernel_ms;
        r->stage_ms[GM_STAGE_NORMALS] = r->normals_kernel_ms;
    }
}

gm_status wait_slot(gm_ctx *ctx, Slot &sl)
{
    if (!sl.submitted) return fail(ctx, GM_ERR_NOT_READY, "slot has no submitted frame");
    if (!sl.complete) {
        GM_HIP(ctx, hipStreamSynchronize(sl.stream));
        fill_result(ctx, sl, &sl.last);
        sl.complete = true;
    }
    return GM_OK;
}

template <class T>
gm_status fetch(gm_ctx *ctx, uint32_t slot, const T *dev, uint32_t avail, T *out, uint32_t capacity, uint32_t *n_out)
{
    if (n_out) *n_out = avail;
    if (avail > capacity) return fail(ctx, GM_ERR_CAPACITY, "output buffer too small");
    if (avail == 0) return GM_OK;
    if (!out) return fail(ctx, GM_ERR_INVALID_ARG, "output pointer is NULL");
    Slot &sl = ctx->slots[slot];
    GM_HIP(ctx, hipMemcpyAsync(out, dev, (size_t)avail * sizeof(T), hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    return GM_OK;
}

gm_status check_slot(gm_ctx *ctx, uint32_t slot)
{
    if (!ctx) return GM_ERR_INVALID_ARG;
    if (slot >= ctx->n_slots) return fail(ctx, GM_ERR_INVALID_ARG, "slot out of range");
    gm_status st = hipSetDevice(ctx->device) == hipSuccess ? GM_OK : GM_ERR_DEVICE;
    if (st != GM_OK) return fail(ctx, st, "hipSetDevice failed");
    return wait_slot(ctx, ctx->slots[slot]);
}

// stage calls reuse slot 0 and leave it "not submitted".  A frame submitted to slot 0 and not yet waited for owns the slot's
// buffers: the stage call is refused (it used to overwrite them and drop the frame's result).
gm_status begin_stage(gm_ctx *ctx, Slot *&sl)
{
    if (!ctx) return GM_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    sl = &ctx->slots[0];
    if (sl->submitted && !sl->complete)
        return fail(ctx, GM_ERR_NOT_READY, "slot 0 holds a submitted frame that has not been waited for (gm_wait_frame first)");
    GM_HIP(ctx, hipStreamSynchronize(sl->stream));
    sl->submitted = false;
    sl->complete = false;
    return GM_OK;
}

}  // namespace

namespace gm {

gm_status gm_fail(gm_ctx *ctx, gm_status st, const char *msg) { return fail(ctx, st, msg); }
VoxDense gm_make_vox_dense(const gm_ctx *ctx, uint32_t n_cap)
{
    const float lo = (float)(-ctx->cfg.boxFilterBound), hi = (float)ctx->cfg.boxFilterBound;
    VoxDense vd;
    memset(&vd, 0, sizeof(vd));
    if ((ctx->cfg.flags & GM_CFG_VOXEL_GRID) && !ctx->force_voxel_sort)
        vd = make_vox_dense(lo, hi, ctx->cfg.voxelGridLeafSize, n_cap, ctx->own_lo, ctx->own_hi);
    return vd;
}
gm_status gm_ensure_capacity(gm_ctx *ctx, Slot &sl, uint32_t n, size_t raw_bytes, bool need_raw)
{
    return ensure_capacity(ctx, sl, n, raw_bytes, need_raw);
}
gm_status gm_begin_stage(gm_ctx *ctx, Slot *&sl) { return begin_stage(ctx, sl); }
gm_status gm_check_slot(gm_ctx *ctx, uint32_t slot) { return check_slot(ctx, slot); }

gm_status gm_ensure_ext(gm_ctx *ctx, Slot &sl, uint32_t H)
{
    if (H == 0 || H > kMaxHypotheses) return fail(ctx, GM_ERR_INVALID_ARG, "ransac_hypotheses must be in [1, 8192]");
    if (H <= sl.ext_H && sl.cap <= sl.ext_cap) return GM_OK;
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    hipFree(sl.hyp_plane); hipFree(sl.hyp_cyl); hipFree(sl.band); hipFree(sl.score_partial); hipFree(sl.cnt_plane);
    hipFree(sl.cnt_cyl); hipFree(sl.best_plane); hipFree(sl.best_cyl); hipFree(sl.mom_partial); hipFree(sl.mom_plane);
    hipFree(sl.mom_cyl); hipFree(sl.nn_best); hipFree(sl.vox_nrm4);
    // (a failed allocation below must not leave dangling pointers for gm_destroy to free again)
    sl.hyp_plane = sl.hyp_cyl = nullptr; sl.band = nullptr; sl.score_partial = nullptr; sl.cnt_plane = sl.cnt_cyl = nullptr;
    sl.best_plane = sl.best_cyl = nullptr; sl.mom_partial = sl.mom_plane = sl.mom_cyl = nullptr; sl.nn_best = nullptr; sl.vox_nrm4 = nullptr;
    const uint32_t HH = H > sl.ext_H ? H : sl.ext_H;
    sl.ext_H = 0; sl.ext_cap = 0;
    GM_HIP(ctx, dmalloc(sl.hyp_plane, (size_t)HH * 8)); GM_HIP(ctx, dmalloc(sl.hyp_cyl, (size_t)HH * 8));
    GM_HIP(ctx, dmalloc(sl.band, HH));
    GM_HIP(ctx, dmalloc(sl.score_partial, (size_t)4096));  // pre-selection scratch (k_ransac.hip kPre*: 577 words, replicated counters at 1024)
    GM_HIP(ctx, dmalloc(sl.cnt_plane, HH)); GM_HIP(ctx, dmalloc(sl.cnt_cyl, HH));
    GM_HIP(ctx, dmalloc(sl.best_plane, 2)); GM_HIP(ctx, dmalloc(sl.best_cyl, 2));
    GM_HIP(ctx, dmalloc(sl.mom_partial, (size_t)kScatterBlocks * 16 * 2));  // [model][row][16]
    GM_HIP(ctx, dmalloc(sl.mom_plane, 16)); GM_HIP(ctx, dmalloc(sl.mom_cyl, 16));
    GM_HIP(ctx, dmalloc(sl.nn_best, sl.cap));
    GM_HIP(ctx, dmalloc(sl.vox_nrm4, sl.cap));
    GM_HIP(ctx, hipMemset(sl.score_partial, 0, sizeof(uint32_t) * 4096));  // (holds the scoring launches' done-counter)
    GM_HIP(ctx, hipMemset(sl.best_plane, 0xFF, 8));
    GM_HIP(ctx, hipMemset(sl.best_cyl, 0xFF, 8));
    sl.ext_H = HH; sl.ext_cap = sl.cap;
    ++sl.alloc_gen;
    return GM_OK;
}

// sequential multi-model RANSAC over the valid cloud of a frame: plane first (label 1),
// then cylinder on what is left (label 2), moments + refits per segment
gm_status gm_enqueue_ransac(gm_ctx *ctx, Slot &sl, uint32_t n_cap, uint32_t scatter_rows, uint32_t row_tile)
{
    const gm_config &cf = ctx->cfg;
    const bool do_plane = (cf.flags & GM_CFG_RANSAC_PLANE) != 0, do_cyl = (cf.flags & GM_CFG_RANSAC_CYLINDER) != 0;
    if (!do_plane && !do_cyl) return GM_OK;
    const uint32_t H = cf.ransac_hypotheses;
    hipStream_t s = sl.stream;
    const uint32_t *n_ptr = &sl.ctr->n_valid;
    // the first model of the frame sees every point (labels == nullptr) and its label pass writes the whole
    // label array, so no memset is needed; best_* are always written by the arg-max kernels
    bool first = true;
    uint32_t mom_rows = 0;
    if (do_plane) {
        uint8_t *lab = first ? nullptr : sl.labels;
        launch_plane_hypotheses(sl.valid4, lab, 0, n_ptr, n_cap, cf.ransac_seed, H, sl.hyp_plane, sl.cnt_plane, s);
        const uint32_t *fsel = nullptr; const int32_t *fcnt = nullptr; uint32_t fk = 0; bool fmask = false, frep = false;
        launch_score_preemptive(0, sl.valid4, lab, 0, n_ptr, n_cap, sl.hyp_plane, sl.band, H, cf.ransac_threshold,
                                sl.score_partial, sl.cnt_plane, sl.best_plane, true, &fsel, &fcnt, &fk, s, sl.inl_mask, &fmask, &frep);
        mom_rows = launch_label(0, sl.valid4, sl.labels, 0, 1, n_ptr, n_cap, sl.hyp_plane, sl.band, sl.best_plane,
                                cf.ransac_threshold, first ? 1 : 0, fsel, fcnt, fk, s, sl.vnorm4, sl.mom_partial, fmask ? sl.inl_mask : nullptr, frep);
        first = false;
    }
    if (do_cyl) {
        uint8_t *lab = first ? nullptr : sl.labels;
        launch_cylinder_hypotheses(sl.valid4, sl.vnorm4, lab, 0, n_ptr, n_cap, cf.ransac_seed + 1, H, sl.hyp_cyl,
                                   sl.cnt_cyl, sl.band, cf.ransac_threshold, s);
        const uint32_t *fsel = nullptr; const int32_t *fcnt = nullptr; uint32_t fk = 0; bool fmask = false, frep = false;
        launch_score_preemptive(1, sl.valid4, lab, 0, n_ptr, n_cap, sl.hyp_cyl, sl.band, H, cf.ransac_threshold,
                                sl.score_partial, sl.cnt_cyl, sl.best_cyl, true, &fsel, &fcnt, &fk, s, sl.inl_mask, &fmask, &frep);
        mom_rows = launch_label(1, sl.valid4, sl.labels, 0, 2, n_ptr, n_cap, sl.hyp_cyl, sl.band, sl.best_cyl,
                                cf.ransac_threshold, first ? 1 : 0, fsel, fcnt, fk, s, sl.vnorm4, sl.mom_partial, fmask ? sl.inl_mask : nullptr, frep);
        first = false;
    }
    // the label passes left the moments of their segments in sl.mom_partial (one row per block, the same grid for both
    // models); the finalizer reduces them
    launch_ext_finalize(sl.hyp_plane, do_plane ? sl.best_plane : nullptr, sl.hyp_cyl, do_cyl ? sl.best_cyl : nullptr,
                        sl.mom_plane, sl.mom_cyl, &sl.d_out->ext, sl.mom_partial, mom_rows, s, sl.tile_partials, scatter_rows, row_tile,
                        sl.ctr, sl.voxp, sl.d_out);
    return GM_OK;
}

}  // namespace gm

extern "C" {

// diagnostic (tests): launch chains captured on a slot so far -- a replayed frame does not add to it
int gm_debug_graph_captures(gm_ctx *ctx, uint32_t slot)
{
    if (!ctx || slot >= ctx->n_slots) return -1;
    return (int)ctx->slots[slot].graph_captures;
}

gm_status gm_host_alloc(gm_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return fail(ctx, GM_ERR_INVALID_ARG, "gm_host_alloc: NULL argument");
    *out = nullptr;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        *out = nullptr;
        return fail(ctx, GM_ERR_OOM, "gm_host_alloc: hipHostMalloc failed");
    }
    return GM_OK;
}

gm_status gm_host_free(gm_ctx *ctx, void *ptr)
{
    if (!ptr) return GM_OK;
    if (hipHostFree(ptr) != hipSuccess) return fail(ctx, GM_ERR_INVALID_ARG, "gm_host_free: not a gm_host_alloc pointer");
    return GM_OK;
}

gm_status gm_host_register(gm_ctx *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || !bytes) return fail(ctx, GM_ERR_INVALID_ARG, "gm_host_register: NULL argument");
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GM_ERR_INVALID_ARG, "gm_host_register: hipHostRegister failed (already registered, or not host memory)");
    }
    return GM_OK;
}

gm_status gm_host_unregister(gm_ctx *ctx, void *ptr)
{
    if (!ptr) return GM_OK;
    if (hipHostUnregister(ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GM_ERR_INVALID_ARG, "gm_host_unregister: not a registered pointer");
    }
    return GM_OK;
}

uint32_t gm_abi_version(void) { return GM_ABI_VERSION; }


const char *gm_status_string(gm_status s)
{
    switch (s) {
    case GM_OK: return "ok";
    case GM_ERR_INVALID_ARG: return "invalid argument";
    case GM_ERR_TOO_FEW_POINTS: return "too few points";
    case GM_ERR_DEVICE: return "device error";
    case GM_ERR_OOM: return "out of memory";
    case GM_ERR_CAPACITY: return "output capacity too small";
    case GM_ERR_NOT_READY: return "not ready";
    case GM_ERR_UNSUPPORTED: return "unsupported";
    case GM_ERR_COMM: return "communication (RCCL) error";
    }
    return "unknown";
}

const char *gm_last_error(const gm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

void gm_default_config(gm_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(gm_config);
    cfg->flags = GM_CFG_DEFAULT;
    cfg->boxFilterBound = 5.0;     // launch/mapping.launch:7
    cfg->voxelGridLeafSize = 0.5;  // :8
    cfg->neighborRadius = 0.5;     // :9
    cfg->weightingFactor = 0.2;    // :10
    cfg->device = 0;
    cfg->n_slots = 1;
    cfg->max_points = 0;
    cfg->ransac_hypotheses = 1024;
    cfg->ransac_threshold = 0.03;
    cfg->ransac_seed = 1;
}

gm_status gm_create(const gm_config *cfg, gm_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, GM_ERR_INVALID_ARG, "gm_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(gm_config)) return fail(nullptr, GM_ERR_INVALID_ARG, "gm_config.struct_size mismatch");
    if (!(cfg->weightingFactor != 0.0) || !std::isfinite(cfg->boxFilterBound) || !(cfg->boxFilterBound >= 0.0) ||
        !(cfg->voxelGridLeafSize > 0.0) || !radius_ok(cfg->neighborRadius))
        return fail(nullptr, GM_ERR_INVALID_ARG, "gm_config: bad numeric parameter");
    if ((cfg->flags & (GM_CFG_RANSAC_PLANE | GM_CFG_RANSAC_CYLINDER)) &&
        (cfg->ransac_hypotheses == 0 || cfg->ransac_hypotheses > kMaxHypotheses || !(cfg->ransac_threshold > 0.0) ||
         !std::isfinite(cfg->ransac_threshold)))
        return fail(nullptr, GM_ERR_INVALID_ARG, "gm_config: ransac_hypotheses must be in [1, 8192] and ransac_threshold > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, GM_ERR_DEVICE, "no HIP device visible (libgm_hip has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GM_ERR_INVALID_ARG, "gm_config.device out of range");
    gm_ctx *ctx = new (std::nothrow) gm_ctx();
    if (!ctx) return fail(nullptr, GM_ERR_OOM, "host allocation failed");
    ctx->cfg = *cfg;
    ctx->device = cfg->device;
    ctx->n_slots = cfg->n_slots ? cfg->n_slots : 1;
    if (ctx->n_slots > 16) ctx->n_slots = 16;
    ctx->own_lo = -std::numeric_limits<double>::infinity();
    ctx->own_hi = std::numeric_limits<double>::infinity();
    ctx->force_voxel_sort = getenv("GM_VOXEL_SORT_PATH") != nullptr;  // tests: exercise the general path
    gm_status st = GM_OK;
    auto body = [&]() -> gm_status {
        GM_HIP(ctx, hipSetDevice(ctx->device));
        hipDeviceProp_t prop;
        GM_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            ctx->err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
            return GM_ERR_DEVICE;
        }
        ctx->slots = new (std::nothrow) Slot[ctx->n_slots];
        if (!ctx->slots) return GM_ERR_OOM;
        for (uint32_t i = 0; i < ctx->n_slots; ++i) {
            Slot &sl = ctx->slots[i];
            sl.pipelined = ctx->n_slots > 1;
            GM_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            for (int k = 0; k <= GM_N_STAGES; ++k) GM_HIP(ctx, hipEventCreate(&sl.ev[k]));
            GM_HIP(ctx, hipEventCreate(&sl.ev_k0));
            GM_HIP(ctx, hipEventCreate(&sl.ev_k1));
            // (the /choppedCloud copy stream is created by gm_set_cloud_output: every stream of the process takes part in
            // the runtime's mapping of streams onto hardware queues, and an idle one can push two slots onto one queue)
            GM_HIP(ctx, dmalloc(sl.ctr, 1));
            GM_HIP(ctx, dmalloc(sl.voxp, 1));
            GM_HIP(ctx, dmalloc(sl.d_out, 1));
            GM_HIP(ctx, dmalloc(sl.partials, (size_t)kScatterBlocks * 6));
            GM_HIP(ctx, dmalloc(sl.frame_in, 8));
            GM_HIP(ctx, hipMemset(sl.frame_in, 0, 32));
            if (const char *e = getenv("GM_TEST_FRAME_COUNTER")) {   // tests only: start near the replayed epochs' wrap
                const uint32_t v = (uint32_t)strtoul(e, nullptr, 0);
                GM_HIP(ctx, hipMemcpy(sl.frame_in + 4, &v, 4, hipMemcpyHostToDevice));
                sl.frames_enqueued = v;
            }
            GM_HIP(ctx, hipHostMalloc((void **)&sl.h_frame_in, 16, hipHostMallocDefault));
            GM_HIP(ctx, hipHostMalloc((void **)&sl.h_out, sizeof(FrameOut), hipHostMallocDefault));
            GM_HIP(ctx, hipMemset(sl.voxp, 0, sizeof(VoxelParams)));
            GM_HIP(ctx, hipMemset(sl.d_out, 0, sizeof(FrameOut)));
            memset(sl.h_out, 0, sizeof(FrameOut));
            if (cfg->max_points) {
                gm_status s2 = ensure_capacity(ctx, sl, cfg->max_points, 0, false);
                if (s2 != GM_OK) return s2;
            }
        }
        return GM_OK;
    };
    st = body();
    if (st != GM_OK) {
        g_create_err = ctx->err;
        gm_destroy(ctx);
        return st;
    }
    *out = ctx;
    return GM_OK;
}

void gm_destroy(gm_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->slots) {
        for (uint32_t i = 0; i < ctx->n_slots; ++i) {
            Slot &sl = ctx->slots[i];
            if (sl.stream) hipStreamSynchronize(sl.stream);
            free_slot_buffers(sl);
            hipFree(sl.ctr); hipFree(sl.voxp); hipFree(sl.d_out); hipFree(sl.partials); hipFree(sl.vox_table);
            hipFree(sl.frame_in);
            if (sl.h_frame_in) hipHostFree(sl.h_frame_in);
            for (int k = 0; k < Slot::kGraphs; ++k) if (sl.graph_exec[k]) hipGraphExecDestroy(sl.graph_exec[k]);
            hipFree(sl.hyp_plane); hipFree(sl.hyp_cyl); hipFree(sl.band); hipFree(sl.score_partial);
            hipFree(sl.cnt_plane); hipFree(sl.cnt_cyl); hipFree(sl.best_plane); hipFree(sl.best_cyl);
            hipFree(sl.mom_partial); hipFree(sl.mom_plane); hipFree(sl.mom_cyl); hipFree(sl.nn_best); hipFree(sl.vox_nrm4);
            if (sl.h_out) hipHostFree(sl.h_out);
            for (int k = 0; k <= GM_N_STAGES; ++k) if (sl.ev[k]) hipEventDestroy(sl.ev[k]);
            if (sl.ev_k0) hipEventDestroy(sl.ev_k0);
            if (sl.ev_k1) hipEventDestroy(sl.ev_k1);
            if (sl.copy_stream) { hipStreamSynchronize(sl.copy_stream); hipStreamDestroy(sl.copy_stream); }
            if (sl.ev_crop) hipEventDestroy(sl.ev_crop);
            if (sl.ev_valid) hipEventDestroy(sl.ev_valid);
            if (sl.ev_copied) hipEventDestroy(sl.ev_copied);
            if (sl.stream) hipStreamDestroy(sl.stream);
        }
        delete[] ctx->slots;
    }
    delete ctx;
}

gm_status gm_set_owned_range(gm_ctx *ctx, double own_lo, double own_hi)
{
    if (!ctx) return GM_ERR_INVALID_ARG;
    if (!(own_lo <= own_hi)) return fail(ctx, GM_ERR_INVALID_ARG, "owned range is empty or NaN");
    ctx->own_lo = own_lo;
    ctx->own_hi = own_hi;
    return GM_OK;
}

static gm_status submit(gm_ctx *ctx, uint32_t slot, const gm_cloud *cloud, bool blocking_call)
{
    if (!ctx || !cloud) return GM_ERR_INVALID_ARG;
    if (slot >= ctx->n_slots) return fail(ctx, GM_ERR_INVALID_ARG, "slot out of range");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    Slot &sl = ctx->slots[slot];
    if (sl.submitted && !sl.complete) {  // previous frame of this slot still owns the buffers
        GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    }
    sl.submitted = false;
    return enqueue_frame(ctx, sl, cloud, blocking_call);
}

gm_status gm_submit_frame(gm_ctx *ctx, uint32_t slot, const gm_cloud *cloud) { return submit(ctx, slot, cloud, false); }

gm_status gm_wait_frame(gm_ctx *ctx, uint32_t slot, gm_frame_result *res)
{
    gm_status st = check_slot(ctx, slot);
    if (st != GM_OK) return st;
    if (res) *res = ctx->slots[slot].last;
    return GM_OK;
}

gm_status gm_poll_frame(gm_ctx *ctx, uint32_t slot)
{
    if (!ctx) return GM_ERR_INVALID_ARG;
    if (slot >= ctx->n_slots) return fail(ctx, GM_ERR_INVALID_ARG, "slot out of range");
    Slot &sl = ctx->slots[slot];
    if (!sl.submitted) return GM_ERR_NOT_READY;   // (not an error worth a message: the caller is asking)
    if (sl.complete) return GM_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    const hipError_t e = hipStreamQuery(sl.stream);   // (the stream ends behind the cloud copy, if there is one)
    if (e == hipSuccess) return GM_OK;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return GM_ERR_NOT_READY; }
    ctx->err = std::string("hipStreamQuery: ") + hipGetErrorString(e);
    return GM_ERR_DEVICE;
}

gm_status gm_set_cloud_output(gm_ctx *ctx, uint32_t slot, float *xyzw, uint32_t capacity)
{
    if (!ctx) return GM_ERR_INVALID_ARG;
    if (slot >= ctx->n_slots) return fail(ctx, GM_ERR_INVALID_ARG, "slot out of range");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, GM_ERR_DEVICE, "hipSetDevice failed");
    Slot &sl = ctx->slots[slot];
    if (sl.submitted && !sl.complete) GM_HIP(ctx, hipStreamSynchronize(sl.stream));   // a frame in flight still writes the old buffer
    if (xyzw) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, xyzw) != hipSuccess || a.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            return fail(ctx, GM_ERR_INVALID_ARG, "gm_set_cloud_output: the buffer must be page-locked (gm_host_alloc / gm_host_register)");
        }
    }
    if (xyzw && !sl.copy_stream) {
        GM_HIP(ctx, hipStreamCreateWithFlags(&sl.copy_stream, hipStreamNonBlocking));
        GM_HIP(ctx, hipEventCreateWithFlags(&sl.ev_crop, hipEventDisableTiming));
        GM_HIP(ctx, hipEventCreateWithFlags(&sl.ev_valid, hipEventDisableTiming));
        GM_HIP(ctx, hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming));
    }
    sl.cloud_out_dev = nullptr;
    if (xyzw) {   // the rows as the device addresses them (page-locked memory is mapped; without a mapping the copy stays a DMA)
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, xyzw, 0) == hipSuccess) sl.cloud_out_dev = reinterpret_cast<float4 *>(dp);
        else (void)hipGetLastError();
    }
    sl.cloud_out = reinterpret_cast<float4 *>(xyzw);
    sl.cloud_out_cap = xyzw ? capacity : 0u;
    return GM_OK;
}

gm_status gm_process_frame(gm_ctx *ctx, const gm_cloud *cloud, gm_frame_result *res)
{
    gm_status st = submit(ctx, 0, cloud, true);
    if (st != GM_OK) return st;
    return gm_wait_frame(ctx, 0, res);
}

gm_status gm_get_cropped_xyz(gm_ctx *ctx, uint32_t slot, float *xyzw, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = check_slot(ctx, slot);
    if (st != GM_OK) return st;
    Slot &sl = ctx->slots[slot];
    return fetch(ctx, slot, (const float4 *)sl.valid4, sl.last.n_valid, (float4 *)xyzw, capacity, n_out);
}

gm_status gm_get_normals(gm_ctx *ctx, uint32_t slot, float *nxyzc, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = check_slot(ctx, slot);
    if (st != GM_OK) return st;
    Slot &sl = ctx->slots[slot];
    return fetch(ctx, slot, (const float4 *)sl.vnorm4, sl.last.n_valid, (float4 *)nxyzc, capacity, n_out);
}

gm_status gm_get_voxel_centroids(gm_ctx *ctx, uint32_t slot, float *xyzc, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = check_slot(ctx, slot);
    if (st != GM_OK) return st;
    Slot &sl = ctx->slots[slot];
    if (!(ctx->cfg.flags & GM_CFG_VOXEL_GRID)) return fail(ctx, GM_ERR_NOT_READY, "context created without GM_CFG_VOXEL_GRID");
    return fetch(ctx, slot, (const float4 *)sl.vox4, sl.last.n_voxels, (float4 *)xyzc, capacity, n_out);
}

gm_status gm_get_neighbor_counts(gm_ctx *ctx, uint32_t slot, int32_t *counts, uint32_t capacity, uint32_t *n_out)
{
    gm_status st = check_slot(ctx, slot);
    if (st != GM_OK) return st;
    Slot &sl = ctx->slots[slot];
    if (!(ctx->cfg.flags & GM_CFG_KEEP_COUNTS)) return fail(ctx, GM_ERR_NOT_READY, "context created without GM_CFG_KEEP_COUNTS");
    return fetch(ctx, slot, (const int32_t *)sl.counts, sl.last.n_cropped, counts, capacity, n_out);
}

// ---- stage-level entry points -----------------------------------------------------

gm_status gm_chop_cloud(gm_ctx *ctx, const gm_cloud *cloud, double bound, float *xyzw_out, uint32_t capacity,
                        uint32_t *n_out)
{
    Slot *slp;
    gm_status st = begin_stage(ctx, slp);
    if (st != GM_OK) return st;
    if (!cloud) return fail(ctx, GM_ERR_INVALID_ARG, "cloud is NULL");
    if (!std::isfinite(bound) || bound < 0) return fail(ctx, GM_ERR_INVALID_ARG, "bound must be finite and >= 0");
    Slot &sl = *slp;
    const uint32_t n = cloud->n_points;
    const size_t raw_bytes = (size_t)n * cloud->point_step;
    const bool on_dev = (cloud->flags & GM_CLOUD_DEVICE) != 0;
    if (n && !cloud->data) return fail(ctx, GM_ERR_INVALID_ARG, "gm_cloud.data is NULL");
    st = check_layout(ctx, cloud);
    if (st != GM_OK) return st;
    st = ensure_capacity(ctx, sl, n ? n : 1u, raw_bytes, !on_dev);
    if (st != GM_OK) return st;
    st = reset_counters(ctx, sl);
    if (st != GM_OK) return st;
    const uint8_t *dev_rows = (const uint8_t *)cloud->data;
    if (!on_dev && n) {
        GM_HIP(ctx, upload_rows(sl, cloud, raw_bytes, true, sl.stream));
        dev_rows = sl.d_raw;
    }
    RowLayout rows;
    st = make_rows(ctx, cloud, dev_rows, rows);
    if (st != GM_OK) return st;
    const float lo = (float)(-bound), hi = (float)bound;
    const GridParams g = make_grid(lo, lo, lo, hi - lo, hi - lo, hi - lo, ctx->cfg.neighborRadius, n);
    launch_crop(rows, n, lo, hi, g, sl, sl.stream);
    uint32_t m = 0;
    GM_HIP(ctx, hipMemcpyAsync(&m, &sl.ctr->n_cropped, 4, hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    return fetch(ctx, 0, (const float4 *)sl.crop4, m, (float4 *)xyzw_out, capacity, n_out);
}

gm_status gm_get_normals_stage(gm_ctx *ctx, const float *xyz, uint32_t n, double radius, float *xyzw_out,
                               float *nxyzc_out, uint32_t capacity, uint32_t *n_out)
{
    Slot *slp;
    gm_status st = begin_stage(ctx, slp);
    if (st != GM_OK) return st;
    if (n && !xyz) return fail(ctx, GM_ERR_INVALID_ARG, "xyz is NULL");
    if (!radius_ok(radius)) return fail(ctx, GM_ERR_INVALID_ARG, "radius must be 0 or within [1e-15, 1e15]");
    Slot &sl = *slp;
    const size_t raw_bytes = (size_t)n * 12;
    st = ensure_capacity(ctx, sl, n, raw_bytes, true);
    if (st != GM_OK) return st;
    st = reset_counters(ctx, sl);
    if (st != GM_OK) return st;
    // extents of the finite rows fix the search grid (the frame path knows them from the crop box)
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], p[k]); mx[k] = fmaxf(mx[k], p[k]); }
    }
    if (mn[0] > mx[0]) { for (int k = 0; k < 3; ++k) { mn[k] = 0; mx[k] = 0; } }
    if (n) {
        memcpy(sl.h_raw, xyz, raw_bytes);
        GM_HIP(ctx, hipMemcpyAsync(sl.d_raw, sl.h_raw, raw_bytes, hipMemcpyHostToDevice, sl.stream));
    }
    gm_cloud c = {sl.d_raw, n, 12, 0, 4, 8, GM_CLOUD_DEVICE};
    RowLayout rows;
    st = make_rows(ctx, &c, sl.d_raw, rows);
    if (st != GM_OK) return st;
    const GridParams g = make_grid(mn[0], mn[1], mn[2], mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], radius, n);
    const float big = std::numeric_limits<float>::max();
    launch_crop(rows, n, -big, big, g, sl, sl.stream);  // drops non-finite rows only
    VoxDense vd_off;
    memset(&vd_off, 0, sizeof(vd_off));
    vd_off.own_lo = -std::numeric_limits<float>::infinity();
    vd_off.own_hi = std::numeric_limits<float>::infinity();
    launch_grid_and_normals(g, vd_off, sl, n, (ctx->cfg.flags & GM_CFG_KEEP_COUNTS) != 0, false, sl.stream);
    launch_compact_valid(sl, n, 1.0, sl.stream);  // (the scatter rows it leaves are not used here)
    uint32_t m[2] = {0, 0};
    GM_HIP(ctx, hipMemcpyAsync(m, &sl.ctr->n_cropped, 8, hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    GM_HIP(ctx, hipGetLastError());
    sl.last.n_cropped = m[0];
    sl.last.n_valid = m[1];
    sl.submitted = true;  // accessors (neighbour counts) may read this slot
    sl.complete = true;
    st = fetch(ctx, 0, (const float4 *)sl.valid4, m[1], (float4 *)xyzw_out, capacity, n_out);
    if (st != GM_OK) return st;
    return fetch(ctx, 0, (const float4 *)sl.vnorm4, m[1], (float4 *)nxyzc_out, capacity, n_out);
}

gm_status gm_get_local_frame(gm_ctx *ctx, const float *nxyzc, uint32_t n, double wf, float eigenvalues[3],
                             float eigenvectors[9], double scatter6[6])
{
    Slot *slp;
    gm_status st = begin_stage(ctx, slp);
    if (st != GM_OK) return st;
    if (n && !nxyzc) return fail(ctx, GM_ERR_INVALID_ARG, "normals pointer is NULL");
    if (!(wf != 0.0)) return fail(ctx, GM_ERR_INVALID_ARG, "weighting factor must be non-zero");
    Slot &sl = *slp;
    st = ensure_capacity(ctx, sl, n, 0, false);
    if (st != GM_OK) return st;
    st = reset_counters(ctx, sl);
    if (st != GM_OK) return st;
    if (n) GM_HIP(ctx, hipMemcpyAsync(sl.vnorm4, nxyzc, (size_t)n * 16, hipMemcpyHostToDevice, sl.stream));
    const uint32_t np = launch_scatter_partials(sl.vnorm4, nullptr, n, wf, sl, sl.stream);
    launch_frame_finalize(sl.partials, np, 0, sl, sl.stream);
    GM_HIP(ctx, hipMemcpyAsync(sl.h_out, sl.d_out, sizeof(FrameOut), hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    GM_HIP(ctx, hipGetLastError());
    if (eigenvalues) for (int k = 0; k < 3; ++k) eigenvalues[k] = sl.h_out->evals[k];
    if (eigenvectors) for (int k = 0; k < 9; ++k) eigenvectors[k] = sl.h_out->evecs[k];
    if (scatter6) for (int k = 0; k < 6; ++k) scatter6[k] = sl.h_out->scatter[k];
    return GM_OK;
}

gm_status gm_voxel_grid(gm_ctx *ctx, const float *xyz, uint32_t n, double leaf, float *xyzc_out, uint32_t capacity,
                        uint32_t *n_out, uint32_t *status_flags)
{
    Slot *slp;
    gm_status st = begin_stage(ctx, slp);
    if (st != GM_OK) return st;
    if (n && !xyz) return fail(ctx, GM_ERR_INVALID_ARG, "xyz is NULL");
    if (!(leaf > 0.0) || !std::isfinite(leaf)) return fail(ctx, GM_ERR_INVALID_ARG, "leaf must be finite and > 0");
    Slot &sl = *slp;
    st = ensure_capacity(ctx, sl, n, (size_t)n * 16, true);
    if (st != GM_OK) return st;
    st = reset_counters(ctx, sl);
    if (st != GM_OK) return st;
    float mn = 3e38f, mx = -3e38f;
    float4 *stage = (float4 *)sl.h_raw;
    for (uint32_t i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        stage[i] = make_float4(p[0], p[1], p[2], 0.f);
        for (int k = 0; k < 3; ++k) { mn = fminf(mn, p[k]); mx = fmaxf(mx, p[k]); }
    }
    if (n) GM_HIP(ctx, hipMemcpyAsync(sl.valid4, stage, (size_t)n * 16, hipMemcpyHostToDevice, sl.stream));
    GM_HIP(ctx, hipMemcpyAsync(&sl.ctr->vox_n, &n, 4, hipMemcpyHostToDevice, sl.stream));
    launch_minmax(sl.valid4, nullptr, n, sl.ctr, sl.stream);
    launch_voxel_grid(sl, n, (float)leaf, n ? voxel_key_bits(mn, mx, leaf) : 1, sl.stream);
    uint32_t V = 0, pass = 0;
    GM_HIP(ctx, hipMemcpyAsync(&V, &sl.ctr->n_voxels, 4, hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipMemcpyAsync(&pass, &sl.voxp->passthrough, 4, hipMemcpyDeviceToHost, sl.stream));
    GM_HIP(ctx, hipStreamSynchronize(sl.stream));
    GM_HIP(ctx, hipGetLastError());
    if (status_flags) *status_flags = pass ? GM_RES_VOXEL_PASSTHROUGH : 0u;
    return fetch(ctx, 0, (const float4 *)sl.vox4, V, (float4 *)xyzc_out, capacity, n_out);
}

gm_status gm_solve_local_frame(const double scatter6[6], float eigenvalues[3], float eigenvectors[9])
{
    if (!scatter6 || !eigenvalues || !eigenvectors) return GM_ERR_INVALID_ARG;
    double w[3], V[9];
    eig3_sym_eigen_signs(scatter6, w, V);
    for (int k = 0; k < 3; ++k) eigenvalues[k] = (float)w[k];
    for (int k = 0; k < 9; ++k) eigenvectors[k] = (float)V[k];
    return GM_OK;
}

}  // extern "C"
