// gm_group.hip -- one host thread driving every local GPU: spatial sharding of ONE frame across the devices of a node
// with an in-library RCCL all-gather of the fitted records, and round-robin streaming of whole frames over the same
// devices (SURVEY.md par. 8b "Threading", par. 8e; north_star: "Frames shard spatially across the 8 GPUs of one node
// with an RCCL all-gather of fitted primitives over xGMI only when a scan exceeds single-GPU capacity"; BASELINE
// configs[3] and configs[4]).
//
// Replaces the single-threaded ros::spin() consumer of /root/reference src/geometric_mapping.cpp:146,169; the reference
// has no counterpart (no parallelism of any kind, SURVEY.md par. 2).
//
// Sharded frame.  The path shards because every per-point stage depends only on points within neighborRadius and the
// final fit is a SUM (M = sum w^2 n n^T) followed by a 3x3 solve.  Rank g owns the points with x in [edge[g], edge[g+1])
// and also receives halo points within 1.01 r of its edges -- neighbours only, never outputs (gm_set_owned_range).
//   * The cut is made on the host before H2D (no device-to-device halo exchange): ONE parallel pass histograms the x of
//     the in-box rows over the VoxelGrid lattice, the edges are put on lattice planes -- exactly, through the device's
//     own float expression floorf(x * inv_leaf) -- so that no voxel of pcl::VoxelGrid (src/tunnel_processing.cpp:217-220)
//     straddles two ranks, and a second parallel pass scatters every row once into per-rank page-locked buffers.
//   * The only exchange step of the path is one ncclAllGather of a 24-double record per rank (scatter partials, counts,
//     the rank's fitted plane / cylinder): latency bound, a few hundred bytes over xGMI.  RCCL is loaded at run time
//     (dlopen) so that a single-GPU host needs no librccl.
//   * Voxel outputs of the group (centroids, and the normal of every centroid's nearest point: the /surfaceNormals
//     markers, src/tunnel_processing.cpp:237-252) are the ranks' own, merged into ascending pcl key order; the nearest
//     point of a centroid is searched on every rank (it may lie across an edge) and the closest wins.
// Streaming.  Frames that fit one GPU are independent: gm_group_submit_frame hands each to the next device's next free
// slot, gm_group_wait_frame returns them in submission order; no collective.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gm_internal.hpp"

using namespace gm;

namespace {

constexpr int kRecLen = 24;  // doubles per rank: scatter[6] | n_in n_cropped n_valid n_voxels | plane[4] cyl[7] inliers[2] | pad

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err)
    {
        // ONE RCCL per process.  A Python process with PyTorch-ROCm must use the copy PyTorch bundles (it has no SONAME in
        // common with /opt/rocm's, so the loader would happily map both, and the two tear each other down at exit):
        // geometric_mapping_amd/_lib.py names it in GM_RCCL_PATH.  A host that already has an RCCL mapped (it links one,
        // or loaded one before us) gets THAT copy: RTLD_NOLOAD finds it without mapping a second one.
        const char *names[] = {getenv("GM_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (handle) break;
        }
        for (const char *n : names) {
            if (handle) break;
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!handle) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};

// FrameOut of a finished frame -> the rank's record, on the device, on the frame's own stream
__global__ void k_pack_record(const FrameOut *__restrict__ o, uint32_t n_in, double *__restrict__ rec)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 6; ++k) rec[k] = o->scatter[k];
    rec[6] = (double)n_in; rec[7] = (double)o->ctr.n_cropped; rec[8] = (double)o->ctr.n_valid; rec[9] = (double)o->ctr.n_voxels;
    for (int k = 0; k < 4; ++k) rec[10 + k] = (double)o->ext.plane[k];
    for (int k = 0; k < 7; ++k) rec[14 + k] = (double)o->ext.cylinder[k];
    rec[21] = (double)o->ext.plane_inliers; rec[22] = (double)o->ext.cylinder_inliers; rec[23] = 0.0;
}

// the normal AND the point (x, y, z, bits(local input row)) behind every nearest-point index
__global__ __launch_bounds__(256) void k_gather_nearest(const int32_t *__restrict__ idx, uint32_t nq, const float4 *__restrict__ pts,
                                                        float4 *__restrict__ pts_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int32_t j = idx[i];
    pts_out[i] = j >= 0 ? pts[j] : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
}

inline float load_f32(const uint8_t *p, bool bswap)
{
    uint32_t u;
    memcpy(&u, p, 4);
    if (bswap) u = __builtin_bswap32(u);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fn(t) for t in [0, T) on T threads (the caller's included)
template <class F>
void parallel_for(uint32_t T, F fn)
{
    if (T <= 1) { fn(0u); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (uint32_t t = 1; t < T; ++t) th.emplace_back([&fn, t]() { fn(t); });
    fn(0u);
    for (auto &x : th) x.join();
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// pcl::VoxelGrid's lattice index of a coordinate, as every kernel computes it (k_normals.hip voxel_sums, k_voxel.hip)
inline long long lattice_cell(float x, float inv_leaf) { return (long long)floorf(x * inv_leaf); }

// smallest float t with lattice_cell(t) >= k: the plane between cells k-1 and k in float space (the map is monotone)
float lattice_plane(long long k, float inv_leaf)
{
    float t = (float)((double)k / (double)inv_leaf);
    for (int it = 0; it < 64 && lattice_cell(t, inv_leaf) >= k; ++it) t = nextafterf(t, -std::numeric_limits<float>::infinity());
    for (int it = 0; it < 128 && lattice_cell(t, inv_leaf) < k; ++it) t = nextafterf(t, std::numeric_limits<float>::infinity());
    return t;
}

}  // namespace

struct gm_group {
    gm_config cfg;
    uint32_t flags = 0;
    uint32_t n = 0;
    bool loopback = false;           // ranks share a device: records travel by device copies, no communicator
    bool dead = false;               // a collective failed half-way: the ranks' streams can no longer be trusted
    std::vector<int> devices;
    std::vector<gm_ctx *> ctx;
    std::vector<double *> d_rec;     // [n] device: this rank's record
    std::vector<double *> d_all;     // [n] device: every rank's record (the all-gather's receive buffer)
    std::vector<hipEvent_t> ev;      // loopback: "record packed" per rank
    double *h_all = nullptr;         // pinned host copy of rank 0's gathered buffer
    std::vector<ncclComm_t> comms;
    Rccl rccl;
    std::vector<uint8_t *> prow;     // per rank: page-locked row buffer of the slab cut (grow-only)
    std::vector<size_t> prow_cap;
    std::vector<std::vector<uint32_t>> row_ids;  // per-rank: input row of every row sent
    std::vector<float> xs;           // x of every input row (scratch of the cut)
    std::vector<double> edges;
    bool edges_on_lattice = false;
    gm_frame_result last = {};
    bool have_frame = false;         // `last`, row_ids and the ranks' slot 0 hold one sharded frame
    double timing[GM_GROUP_N_TIMINGS] = {};
    std::vector<float> vox_cen;      // merged voxel centroids of the last sharded frame: x, y, z, count per voxel
    std::vector<float> vox_nrm;      // ... and the normal of each centroid's nearest point (filled on demand)
    std::vector<int32_t> vox_near;   // ... and that point's index in the merged /choppedCloud
    bool vox_nrm_valid = false;
    // streaming: frames in flight in submission order
    struct Ticket { uint32_t rank, slot; };
    std::deque<Ticket> inflight;
    uint32_t next_rank = 0;
    std::vector<uint32_t> next_slot;
    std::string err;
};

namespace {
thread_local std::string g_group_create_err;

gm_status gfail(gm_group *g, gm_status st, const std::string &msg)
{
    if (g) g->err = msg; else g_group_create_err = msg;
    return st;
}

// after a failure behind the first submit: nothing of the frame may still be in flight when we return
void drain(gm_group &G)
{
    for (uint32_t r = 0; r < G.n; ++r) {
        if (!G.ctx[r]) continue;
        hipSetDevice(G.devices[r]);
        for (uint32_t s = 0; s < G.ctx[r]->n_slots; ++s) hipStreamSynchronize(G.ctx[r]->slots[s].stream);
    }
    (void)hipGetLastError();
}

struct CutPlan {
    std::vector<double> edges;
    bool on_lattice = false;
    uint32_t n_inside = 0;
};

// ---- the host side of the cut (see the header of this file).  Returns per-rank row counts in total[].
gm_status cut_rows(gm_group &G, const gm_cloud *cloud, CutPlan &plan, std::vector<uint32_t> &total)
{
    const uint32_t n = cloud->n_points, R = G.n;
    const uint64_t step = cloud->point_step;
    const bool bswap = (cloud->flags & GM_CLOUD_BIGENDIAN) != 0;
    const uint8_t *base = (const uint8_t *)cloud->data;
    const float lo = (float)(-G.cfg.boxFilterBound), hi = (float)G.cfg.boxFilterBound;
    const float inv_leaf = 1.0f / (float)G.cfg.voxelGridLeafSize;
    uint32_t T = std::thread::hardware_concurrency();
    if (const char *e = getenv("GM_GROUP_THREADS")) T = (uint32_t)atoi(e);
    if (T > 16) T = 16;
    if (T < 1 || n < 262144u) T = 1;
    // histogram bins: lattice cells of the box along x (several cells per bin when the lattice is very fine); when the
    // lattice is too coarse to balance the ranks, 4096 uniform bins instead (a voxel may then straddle an edge: the
    // dense tables of the ranks are merged exactly, see merge_voxels)
    const long long c_lo = lattice_cell(lo, inv_leaf), c_hi = lattice_cell(hi, inv_leaf);
    const long long ncell = c_hi - c_lo + 1;
    const bool want_voxels = (G.cfg.flags & GM_CFG_VOXEL_GRID) != 0;
    plan.on_lattice = want_voxels && ncell >= (long long)4 * R;
    int shift = 0;
    while (plan.on_lattice && (ncell >> shift) > (1ll << 20)) ++shift;
    const uint32_t nbins = plan.on_lattice ? (uint32_t)((ncell + (1ll << shift) - 1) >> shift) : 4096u;
    const double ubin = (double)nbins / ((double)hi - (double)lo > 0 ? (double)hi - (double)lo : 1.0);
    G.xs.resize(n);
    std::vector<std::vector<uint32_t>> hist(T, std::vector<uint32_t>(nbins, 0u));
    std::vector<uint32_t> inside(T, 0u);
    parallel_for(T, [&](uint32_t t) {
        const uint32_t i0 = (uint32_t)((uint64_t)n * t / T), i1 = (uint32_t)((uint64_t)n * (t + 1) / T);
        std::vector<uint32_t> &h = hist[t];
        uint32_t in = 0;
        for (uint32_t i = i0; i < i1; ++i) {
            const uint8_t *row = base + (size_t)i * step;
            const float x = load_f32(row + cloud->off_x, bswap), y = load_f32(row + cloud->off_y, bswap), z = load_f32(row + cloud->off_z, bswap);
            G.xs[i] = x;
            if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z) && !(x < lo || y < lo || z < lo || x > hi || y > hi || z > hi)) {
                uint32_t b;
                if (plan.on_lattice) b = (uint32_t)((lattice_cell(x, inv_leaf) - c_lo) >> shift);
                else { const double u = ((double)x - (double)lo) * ubin; b = u < 0 ? 0u : (uint32_t)u; }
                ++h[b < nbins ? b : nbins - 1];
                ++in;
            }
        }
        inside[t] = in;
    });
    uint64_t n_in = 0;
    for (uint32_t t = 0; t < T; ++t) n_in += inside[t];
    plan.n_inside = (uint32_t)n_in;
    // edges: the bin boundary at which the running count first reaches g / R of the in-box rows
    plan.edges.assign(R + 1, 0.0);
    plan.edges[0] = -std::numeric_limits<double>::infinity();
    plan.edges[R] = std::numeric_limits<double>::infinity();
    {
        uint64_t run = 0;
        uint32_t g = 1;
        for (uint32_t b = 0; b < nbins && g < R; ++b) {
            uint64_t c = 0;
            for (uint32_t t = 0; t < T; ++t) c += hist[t][b];
            run += c;
            while (g < R && run * R >= n_in * (uint64_t)g && n_in) {   // the edge behind bin b
                if (plan.on_lattice) plan.edges[g] = (double)lattice_plane(c_lo + ((long long)(b + 1) << shift), inv_leaf);
                else plan.edges[g] = (double)(float)((double)lo + (double)(b + 1) / ubin);
                ++g;
            }
        }
        for (; g < R; ++g) plan.edges[g] = n_in ? plan.edges[g - 1] : 0.0;   // (an empty frame: every edge at 0)
        for (uint32_t k = 1; k < R; ++k) if (plan.edges[k] < plan.edges[k - 1]) plan.edges[k] = plan.edges[k - 1];
    }
    // membership: rank r takes the rows with x in [edge[r] - halo, edge[r+1] + halo)   (NaN rows fail both comparisons:
    // every rank's crop would drop them anyway)
    const double halo = 1.01 * G.cfg.neighborRadius;
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>(R, 0u));
    parallel_for(T, [&](uint32_t t) {
        const uint32_t i0 = (uint32_t)((uint64_t)n * t / T), i1 = (uint32_t)((uint64_t)n * (t + 1) / T);
        std::vector<uint32_t> &c = cnt[t];
        for (uint32_t i = i0; i < i1; ++i) {
            const double x = (double)G.xs[i];
            for (uint32_t r = 0; r < R; ++r) c[r] += (x >= plan.edges[r] - halo && x < plan.edges[r + 1] + halo) ? 1u : 0u;
        }
    });
    total.assign(R, 0u);
    std::vector<std::vector<uint32_t>> off(T, std::vector<uint32_t>(R, 0u));
    for (uint32_t r = 0; r < R; ++r)
        for (uint32_t t = 0; t < T; ++t) { off[t][r] = total[r]; total[r] += cnt[t][r]; }
    for (uint32_t r = 0; r < R; ++r) {
        const size_t need = (size_t)total[r] * step;
        if (need > G.prow_cap[r]) {
            if (G.prow[r]) hipHostFree(G.prow[r]);
            G.prow[r] = nullptr; G.prow_cap[r] = 0;
            const size_t cap = need + need / 8 + 4096;
            if (hipSetDevice(G.devices[r]) != hipSuccess || hipHostMalloc((void **)&G.prow[r], cap, hipHostMallocDefault) != hipSuccess)
                return gfail(&G, GM_ERR_OOM, "gm_group: page-locked row buffer allocation failed");
            G.prow_cap[r] = cap;
        }
        G.row_ids[r].resize(total[r]);
    }
    parallel_for(T, [&](uint32_t t) {
        const uint32_t i0 = (uint32_t)((uint64_t)n * t / T), i1 = (uint32_t)((uint64_t)n * (t + 1) / T);
        std::vector<uint32_t> at(off[t]);
        for (uint32_t i = i0; i < i1; ++i) {
            const double x = (double)G.xs[i];
            for (uint32_t r = 0; r < R; ++r)
                if (x >= plan.edges[r] - halo && x < plan.edges[r + 1] + halo) {
                    memcpy(G.prow[r] + (size_t)at[r] * step, base + (size_t)i * step, step);
                    G.row_ids[r][at[r]] = i;
                    ++at[r];
                }
        }
    });
    return GM_OK;
}

struct VoxRow { long long kz, ky, kx; float v[4]; };

// Voxel centroids of the group in ascending pcl key order (z slowest, x fastest: the key's order for any box).  Edges on
// the lattice: every voxel belongs to one rank, the ranks' lists are concatenated and ordered.  Otherwise (a lattice too
// coarse to cut along, which always means the dense-table path) a voxel may hold points of two ranks: the ranks' exact
// fixed-point sums are added per cell and divided once, as one rank would have.
gm_status merge_voxels(gm_group &G, const std::vector<gm_frame_result> &rr)
{
    const float inv = 1.0f / (float)G.cfg.voxelGridLeafSize;
    std::vector<VoxRow> rows;
    if (G.edges_on_lattice || G.n == 1) {
        std::vector<float> cen;
        for (uint32_t r = 0; r < G.n; ++r) {
            uint32_t v = rr[r].n_voxels, got = 0;
            cen.resize((size_t)(v ? v : 1) * 4);
            const gm_status st = gm_get_voxel_centroids(G.ctx[r], 0, cen.data(), v ? v : 1, &got);
            if (st != GM_OK) return gfail(&G, st, gm_last_error(G.ctx[r]));
            for (uint32_t i = 0; i < got; ++i) {
                VoxRow w;
                w.kx = lattice_cell(cen[4 * i], inv); w.ky = lattice_cell(cen[4 * i + 1], inv); w.kz = lattice_cell(cen[4 * i + 2], inv);
                memcpy(w.v, &cen[4 * (size_t)i], 16);
                rows.push_back(w);
            }
        }
    } else {
        const VoxDense vd = gm_make_vox_dense(G.ctx[0], G.ctx[0]->slots[0].cap);
        if (!vd.enabled) return gfail(&G, GM_ERR_UNSUPPORTED, "gm_group: voxel lattice neither cuttable nor dense");
        const size_t cells = (size_t)vd.dim * vd.dim * vd.dim;
        std::vector<VoxCell> sum(cells), part(cells);
        memset(sum.data(), 0, cells * sizeof(VoxCell));
        for (uint32_t r = 0; r < G.n; ++r) {
            Slot &sl = G.ctx[r]->slots[0];
            if (hipSetDevice(G.devices[r]) != hipSuccess ||
                hipMemcpy(part.data(), sl.vox_table, cells * sizeof(VoxCell), hipMemcpyDeviceToHost) != hipSuccess)
                return gfail(&G, GM_ERR_DEVICE, "gm_group: D2H of a rank's voxel table failed");
            for (size_t c = 0; c < cells; ++c) { sum[c].sx += part[c].sx; sum[c].sy += part[c].sy; sum[c].sz += part[c].sz; sum[c].cnt += part[c].cnt; }
        }
        for (size_t c = 0; c < cells; ++c) {
            if (!sum[c].cnt) continue;
            // DenseEmit of k_voxel.hip, operation for operation (the device contracts lo + s * inv into one fma)
            const double q = vd.inv_scale / (double)sum[c].cnt, l = (double)vd.lo;
            VoxRow w;
            w.kx = (long long)(c % vd.dim); w.ky = (long long)((c / vd.dim) % vd.dim); w.kz = (long long)(c / ((size_t)vd.dim * vd.dim));
            w.v[0] = (float)fma((double)sum[c].sx, q, l); w.v[1] = (float)fma((double)sum[c].sy, q, l);
            w.v[2] = (float)fma((double)sum[c].sz, q, l); w.v[3] = (float)sum[c].cnt;
            rows.push_back(w);
        }
    }
    std::stable_sort(rows.begin(), rows.end(), [](const VoxRow &a, const VoxRow &b) {
        return a.kz != b.kz ? a.kz < b.kz : (a.ky != b.ky ? a.ky < b.ky : a.kx < b.kx);
    });
    G.vox_cen.resize(rows.size() * 4);
    for (size_t i = 0; i < rows.size(); ++i) memcpy(&G.vox_cen[4 * i], rows[i].v, 16);
    G.last.n_voxels = (uint32_t)rows.size();
    return GM_OK;
}

}  // namespace

extern "C" {

const char *gm_group_last_error(const gm_group *grp) { return grp ? grp->err.c_str() : g_group_create_err.c_str(); }
uint32_t gm_group_size(const gm_group *grp) { return grp ? grp->n : 0u; }
gm_ctx *gm_group_ctx(gm_group *grp, uint32_t rank) { return (grp && rank < grp->n) ? grp->ctx[rank] : nullptr; }

void gm_group_destroy(gm_group *grp)
{
    if (!grp) return;
    for (uint32_t r = 0; r < grp->n; ++r) {
        if (r < grp->ctx.size() && grp->ctx[r]) hipSetDevice(grp->devices[r]);
        if (r < grp->comms.size() && grp->comms[r] && grp->rccl.CommDestroy) grp->rccl.CommDestroy(grp->comms[r]);
        if (r < grp->d_rec.size()) hipFree(grp->d_rec[r]);
        if (r < grp->d_all.size()) hipFree(grp->d_all[r]);
        if (r < grp->ev.size() && grp->ev[r]) hipEventDestroy(grp->ev[r]);
        if (r < grp->ctx.size() && grp->ctx[r]) gm_destroy(grp->ctx[r]);   // (drains the rank's streams first)
        if (r < grp->prow.size() && grp->prow[r]) hipHostFree(grp->prow[r]);
    }
    if (grp->h_all) hipHostFree(grp->h_all);
    delete grp;
}

gm_status gm_group_create(const gm_config *cfg, const int32_t *devices, uint32_t n_ranks, uint32_t flags, gm_group **out)
{
    if (!cfg || !out || !devices || n_ranks == 0 || n_ranks > 64)
        return gfail(nullptr, GM_ERR_INVALID_ARG, "gm_group_create: bad argument (1..64 ranks)");
    *out = nullptr;
    gm_group *g = new (std::nothrow) gm_group();
    if (!g) return gfail(nullptr, GM_ERR_OOM, "host allocation failed");
    g->cfg = *cfg; g->flags = flags; g->n = n_ranks;
    g->devices.assign(devices, devices + n_ranks);
    bool repeats = false;
    for (uint32_t a = 0; a < n_ranks; ++a)
        for (uint32_t b = a + 1; b < n_ranks; ++b) repeats |= devices[a] == devices[b];
    if (repeats && !(flags & GM_GROUP_LOOPBACK)) {
        delete g;
        return gfail(nullptr, GM_ERR_INVALID_ARG, "gm_group_create: a device is listed twice (pass GM_GROUP_LOOPBACK to allow it)");
    }
    g->loopback = repeats;
    g->ctx.assign(n_ranks, nullptr); g->d_rec.assign(n_ranks, nullptr); g->d_all.assign(n_ranks, nullptr);
    g->ev.assign(n_ranks, nullptr); g->comms.assign(n_ranks, nullptr);
    g->prow.assign(n_ranks, nullptr); g->prow_cap.assign(n_ranks, 0); g->row_ids.resize(n_ranks);
    g->next_slot.assign(n_ranks, 0u);
    auto bail = [&](gm_status st, const std::string &msg) { g_group_create_err = msg; gm_group_destroy(g); return st; };
    for (uint32_t r = 0; r < n_ranks; ++r) {
        gm_config c = *cfg;
        c.device = devices[r];
        c.ransac_seed = cfg->ransac_seed + r;   // ranks draw different hypotheses: more candidates for the vote
        const gm_status st = gm_create(&c, &g->ctx[r]);
        if (st != GM_OK) return bail(st, std::string("gm_group_create: rank ") + std::to_string(r) + ": " + gm_last_error(nullptr));
        if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc((void **)&g->d_rec[r], sizeof(double) * kRecLen) != hipSuccess ||
            hipMalloc((void **)&g->d_all[r], sizeof(double) * kRecLen * n_ranks) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming) != hipSuccess)
            return bail(GM_ERR_DEVICE, "gm_group_create: device allocation failed");
    }
    if (hipHostMalloc((void **)&g->h_all, sizeof(double) * kRecLen * n_ranks, hipHostMallocDefault) != hipSuccess)
        return bail(GM_ERR_OOM, "gm_group_create: hipHostMalloc failed");
    if (!g->loopback) {
        std::string e;
        if (!g->rccl.load(e)) return bail(GM_ERR_COMM, e);
        const ncclResult_t rc = g->rccl.CommInitAll(g->comms.data(), (int)n_ranks, g->devices.data());
        if (rc != ncclSuccess) return bail(GM_ERR_COMM, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(rc));
    }
    *out = g;
    return GM_OK;
}

gm_status gm_group_process_frame(gm_group *grp, const gm_cloud *cloud, gm_frame_result *res)
{
    if (!grp || !cloud) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (G.dead) return gfail(grp, GM_ERR_COMM, "gm_group: an earlier collective failed; destroy the group");
    if (!G.inflight.empty()) return gfail(grp, GM_ERR_NOT_READY, "gm_group_process_frame: streamed frames are still in flight (gm_group_wait_frame)");
    // whatever happens below, the accessors must not pair this frame's rows with an older frame's results
    G.have_frame = false;
    G.last = gm_frame_result{};
    G.vox_cen.clear(); G.vox_nrm.clear(); G.vox_near.clear(); G.vox_nrm_valid = false;
    if (cloud->flags & GM_CLOUD_DEVICE) return gfail(grp, GM_ERR_UNSUPPORTED, "gm_group_process_frame cuts the slabs on the host: pass host rows");
    const uint32_t n = cloud->n_points;
    const uint64_t step = cloud->point_step;
    if (n && !cloud->data) return gfail(grp, GM_ERR_INVALID_ARG, "gm_cloud.data is NULL");
    if (n && (step < 12 || (uint64_t)cloud->off_x + 4 > step || (uint64_t)cloud->off_y + 4 > step || (uint64_t)cloud->off_z + 4 > step))
        return gfail(grp, GM_ERR_INVALID_ARG, "gm_cloud: x/y/z offsets do not fit in point_step");
    for (int k = 0; k < GM_GROUP_N_TIMINGS; ++k) G.timing[k] = 0.0;
    const double t0 = now_ms();
    CutPlan plan;
    std::vector<uint32_t> total;
    gm_status st;
    try {   // (the cut allocates and starts threads: nothing may leave through the C ABI)
        st = cut_rows(G, cloud, plan, total);
    } catch (const std::bad_alloc &) {
        return gfail(grp, GM_ERR_OOM, "gm_group_process_frame: host allocation failed in the slab cut");
    } catch (const std::exception &e) {
        return gfail(grp, GM_ERR_DEVICE, std::string("gm_group_process_frame: the slab cut failed: ") + e.what());
    }
    if (st != GM_OK) return st;
    G.edges = plan.edges;
    G.edges_on_lattice = plan.on_lattice;
    const double t1 = now_ms();
    // ---- every rank runs the unchanged single-GPU pipeline on its slab, asynchronously
    auto fail_after_submit = [&](gm_status s, const std::string &msg) { drain(G); return gfail(grp, s, msg); };
    for (uint32_t r = 0; r < G.n; ++r) {
        st = gm_set_owned_range(G.ctx[r], G.edges[r], G.edges[r + 1]);
        if (st != GM_OK) return fail_after_submit(st, gm_last_error(G.ctx[r]));
        gm_cloud c = *cloud;
        c.data = total[r] ? G.prow[r] : nullptr;
        c.n_points = total[r];
        c.flags = (cloud->flags & GM_CLOUD_BIGENDIAN) | GM_CLOUD_PINNED;
        st = gm_submit_frame(G.ctx[r], 0, &c);
        if (st != GM_OK) return fail_after_submit(st, std::string("rank ") + std::to_string(r) + ": " + gm_last_error(G.ctx[r]));
        Slot &sl = G.ctx[r]->slots[0];
        hipLaunchKernelGGL(k_pack_record, dim3(1), dim3(64), 0, sl.stream, (const FrameOut *)sl.d_out, c.n_points, G.d_rec[r]);
        if (G.loopback) hipEventRecord(G.ev[r], sl.stream);
    }
    const double t2 = now_ms();
    // ---- the one exchange step: all-gather of the per-rank records
    if (!G.loopback) {
        ncclResult_t rc = G.rccl.GroupStart();
        for (uint32_t r = 0; r < G.n && rc == ncclSuccess; ++r) {
            hipSetDevice(G.devices[r]);
            rc = G.rccl.AllGather(G.d_rec[r], G.d_all[r], kRecLen, ncclDouble, G.comms[r], G.ctx[r]->slots[0].stream);
        }
        const ncclResult_t rc2 = G.rccl.GroupEnd();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) {
            // some ranks may have posted their half of the collective: their streams can block for good.  No drain (it
            // could hang with them); the group refuses further work.
            G.dead = true;
            return gfail(grp, GM_ERR_COMM, std::string("ncclAllGather: ") + G.rccl.GetErrorString(rc));
        }
    } else {
        // ranks on one device (tests on a 1-GPU box): the same gather by device copies, ordered by events
        for (uint32_t r = 0; r < G.n; ++r) {
            hipStream_t s = G.ctx[r]->slots[0].stream;
            for (uint32_t q = 0; q < G.n; ++q) {
                if (q != r) hipStreamWaitEvent(s, G.ev[q], 0);
                hipMemcpyAsync(G.d_all[r] + (size_t)q * kRecLen, G.d_rec[q], sizeof(double) * kRecLen, hipMemcpyDeviceToDevice, s);
            }
        }
    }
    hipSetDevice(G.devices[0]);
    if (hipMemcpyAsync(G.h_all, G.d_all[0], sizeof(double) * kRecLen * G.n, hipMemcpyDeviceToHost, G.ctx[0]->slots[0].stream) != hipSuccess)
        return fail_after_submit(GM_ERR_DEVICE, "D2H of the gathered records failed");
    std::vector<gm_frame_result> rr(G.n);
    for (uint32_t r = 0; r < G.n; ++r) {
        st = gm_wait_frame(G.ctx[r], 0, &rr[r]);
        if (st != GM_OK) return fail_after_submit(st, std::string("rank ") + std::to_string(r) + ": " + gm_last_error(G.ctx[r]));
    }
    hipSetDevice(G.devices[0]);
    if (hipStreamSynchronize(G.ctx[0]->slots[0].stream) != hipSuccess) return fail_after_submit(GM_ERR_DEVICE, "stream sync failed");
    const double t3 = now_ms();
    // ---- merge (every rank holds the same gathered records; rank 0's copy is read)
    gm_frame_result out;
    memset(&out, 0, sizeof(out));
    out.n_in = n;
    out.n_cropped = plan.n_inside;
    double M[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t r = 0; r < G.n; ++r) {   // rank order: deterministic
        const double *rec = G.h_all + (size_t)r * kRecLen;
        for (int k = 0; k < 6; ++k) M[k] += rec[k];
        out.n_valid += (uint32_t)rec[8];
    }
    for (int k = 0; k < 6; ++k) out.scatter[k] = M[k];
    double w[3], V[9];
    eig3_sym_eigen_signs(M, w, V);
    for (int k = 0; k < 3; ++k) { out.eigenvalues[k] = (float)w[k]; out.center_axis[k] = (float)V[k]; }
    for (int k = 0; k < 9; ++k) out.eigenvectors[k] = (float)V[k];
    G.last = out;
    if (G.cfg.flags & GM_CFG_VOXEL_GRID) {
        st = merge_voxels(G, rr);
        if (st != GM_OK) { G.last = gm_frame_result{}; return st; }
        out.n_voxels = G.last.n_voxels;
    }
    // fitted primitives: every rank's fit is a candidate for the whole frame; each rank counts every candidate's
    // inliers on its own resident valid cloud (owned points only: the counts add up to the count on the unsharded
    // frame); the largest total wins, lowest rank on ties.  One process sees every rank: the counts are summed here.
    for (int model = 0; model < 2; ++model) {
        const uint32_t flag = model == 0 ? GM_CFG_RANSAC_PLANE : GM_CFG_RANSAC_CYLINDER;
        if (!(G.cfg.flags & flag)) continue;
        const int w0 = model == 0 ? 10 : 14, wl = model == 0 ? 4 : 7;
        std::vector<float> cand;
        std::vector<uint32_t> owner;
        for (uint32_t r = 0; r < G.n; ++r) {
            const double *rec = G.h_all + (size_t)r * kRecLen;
            bool ok = rec[21 + model] > 0;
            for (int k = 0; k < wl; ++k) ok = ok && std::isfinite(rec[w0 + k]);
            if (!ok) continue;
            for (int k = 0; k < wl; ++k) cand.push_back((float)rec[w0 + k]);
            owner.push_back(r);
        }
        if (owner.empty()) continue;
        std::vector<long long> totalc(owner.size(), 0);
        std::vector<int32_t> cnt(owner.size());
        for (uint32_t r = 0; r < G.n; ++r) {
            st = gm_score_frame(G.ctx[r], 0, model, cand.data(), (uint32_t)owner.size(), G.cfg.ransac_threshold, 0, cnt.data());
            if (st != GM_OK) { G.last = gm_frame_result{}; return gfail(grp, st, gm_last_error(G.ctx[r])); }
            for (size_t k = 0; k < owner.size(); ++k) totalc[k] += cnt[k];
        }
        size_t best = 0;
        for (size_t k = 1; k < owner.size(); ++k) if (totalc[k] > totalc[best]) best = k;
        const gm_frame_result &src = rr[owner[best]];
        if (model == 0) {
            out.plane_inliers = (uint32_t)totalc[best];
            for (int k = 0; k < 4; ++k) { out.plane[k] = src.plane[k]; out.plane_refit[k] = src.plane_refit[k]; }
        } else {
            out.cylinder_inliers = (uint32_t)totalc[best];
            for (int k = 0; k < 7; ++k) out.cylinder[k] = src.cylinder[k];
            for (int k = 0; k < 3; ++k) out.cylinder_axis_refit[k] = src.cylinder_axis_refit[k];
        }
    }
    float km = 0.f;
    for (uint32_t r = 0; r < G.n; ++r) km = fmaxf(km, rr[r].normals_kernel_ms);
    out.normals_kernel_ms = km;
    const double t4 = now_ms();
    G.timing[GM_GROUP_T_CUT] = t1 - t0; G.timing[GM_GROUP_T_SUBMIT] = t2 - t1; G.timing[GM_GROUP_T_DEVICE] = t3 - t2;
    G.timing[GM_GROUP_T_MERGE] = t4 - t3; G.timing[GM_GROUP_T_TOTAL] = t4 - t0;
    G.last = out;
    G.have_frame = true;
    if (res) *res = out;
    return GM_OK;
}

gm_status gm_group_get_timing(const gm_group *grp, double *ms, uint32_t capacity)
{
    if (!grp || !ms) return GM_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < capacity && k < (uint32_t)GM_GROUP_N_TIMINGS; ++k) ms[k] = grp->timing[k];
    return GM_OK;
}

gm_status gm_group_get_edges(const gm_group *grp, double *edges, uint32_t capacity, uint32_t *on_lattice)
{
    if (!grp || !edges) return GM_ERR_INVALID_ARG;
    if (capacity < grp->n + 1 || grp->edges.size() != (size_t)grp->n + 1) return GM_ERR_CAPACITY;
    for (uint32_t k = 0; k <= grp->n; ++k) edges[k] = grp->edges[k];
    if (on_lattice) *on_lattice = grp->edges_on_lattice ? 1u : 0u;
    return GM_OK;
}

// /choppedCloud of a sharded frame: every rank's valid cloud, merged back into the single-GPU order (ascending input
// row).  Rows x,y,z,pad with pad = the row index in the ORIGINAL cloud.
gm_status gm_group_get_cropped_xyz(gm_group *grp, float *xyzw, uint32_t capacity, uint32_t *n_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (!G.have_frame) { if (n_out) *n_out = 0; return gfail(grp, GM_ERR_NOT_READY, "gm_group: no completed sharded frame"); }
    const uint32_t total = G.last.n_valid;
    if (n_out) *n_out = total;
    if (total > capacity) return gfail(grp, GM_ERR_CAPACITY, "output buffer too small");
    if (!total) return GM_OK;
    if (!xyzw) return gfail(grp, GM_ERR_INVALID_ARG, "output pointer is NULL");
    struct Row { uint32_t id; float x, y, z; };
    std::vector<Row> all;
    all.reserve(total);
    std::vector<float> buf;
    for (uint32_t r = 0; r < G.n; ++r) {
        uint32_t m = 0;
        gm_status st = gm_get_cropped_xyz(G.ctx[r], 0, nullptr, 0, &m);
        if (st != GM_OK && st != GM_ERR_CAPACITY) return gfail(grp, st, gm_last_error(G.ctx[r]));
        buf.resize((size_t)(m ? m : 1) * 4);
        st = gm_get_cropped_xyz(G.ctx[r], 0, buf.data(), m ? m : 1, &m);
        if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
        for (uint32_t i = 0; i < m; ++i) {
            uint32_t local;
            memcpy(&local, &buf[4 * (size_t)i + 3], 4);
            if (local >= G.row_ids[r].size()) return gfail(grp, GM_ERR_DEVICE, "gm_group: a rank's row index is out of range");
            all.push_back(Row{G.row_ids[r][local], buf[4 * (size_t)i], buf[4 * (size_t)i + 1], buf[4 * (size_t)i + 2]});
        }
    }
    std::stable_sort(all.begin(), all.end(), [](const Row &a, const Row &b) { return a.id < b.id; });
    for (size_t i = 0; i < all.size() && i < capacity; ++i) {
        xyzw[4 * i] = all[i].x; xyzw[4 * i + 1] = all[i].y; xyzw[4 * i + 2] = all[i].z;
        memcpy(&xyzw[4 * i + 3], &all[i].id, 4);
    }
    return GM_OK;
}

gm_status gm_group_get_voxel_centroids(gm_group *grp, float *xyzc, uint32_t capacity, uint32_t *n_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (!G.have_frame) { if (n_out) *n_out = 0; return gfail(grp, GM_ERR_NOT_READY, "gm_group: no completed sharded frame"); }
    if (!(G.cfg.flags & GM_CFG_VOXEL_GRID)) return gfail(grp, GM_ERR_NOT_READY, "group created without GM_CFG_VOXEL_GRID");
    const uint32_t V = (uint32_t)(G.vox_cen.size() / 4);
    if (n_out) *n_out = V;
    if (V > capacity) return gfail(grp, GM_ERR_CAPACITY, "output buffer too small");
    if (V && !xyzc) return gfail(grp, GM_ERR_INVALID_ARG, "output pointer is NULL");
    if (V) memcpy(xyzc, G.vox_cen.data(), (size_t)V * 16);
    return GM_OK;
}

// The normal of every centroid's nearest valid point (normals->at(kIndices[0]) of the marker loop, src/tunnel_processing.cpp:
// 239-249).  The nearest point of a centroid next to a slab edge may belong to the neighbouring rank: every rank searches
// every centroid in its own valid cloud, the smallest (distance, input row) wins -- the 1-NN of the unsharded frame.
gm_status gm_group_get_voxel_normals(gm_group *grp, float *nxyzc, uint32_t capacity, uint32_t *n_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (!G.have_frame) { if (n_out) *n_out = 0; return gfail(grp, GM_ERR_NOT_READY, "gm_group: no completed sharded frame"); }
    if ((G.cfg.flags & (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID)) != (GM_CFG_NEAREST | GM_CFG_VOXEL_GRID))
        return gfail(grp, GM_ERR_NOT_READY, "group created without GM_CFG_NEAREST | GM_CFG_VOXEL_GRID");
    const uint32_t V = (uint32_t)(G.vox_cen.size() / 4);
    if (n_out) *n_out = V;
    if (V > capacity) return gfail(grp, GM_ERR_CAPACITY, "output buffer too small");
    if (!V) return GM_OK;
    if (!nxyzc) return gfail(grp, GM_ERR_INVALID_ARG, "output pointer is NULL");
    if (!G.vox_nrm_valid) {
        G.vox_nrm.assign((size_t)V * 4, std::numeric_limits<float>::quiet_NaN());
        std::vector<unsigned long long> best_key(V, ~0ull);   // (distance bits << 32 | input row) of the winner so far
        std::vector<unsigned long long> keys;
        std::vector<float> nrm, pts;
        for (uint32_t r = 0; r < G.n; ++r) {
            gm_ctx *c = G.ctx[r];
            Slot &sl = c->slots[0];
            if (hipSetDevice(G.devices[r]) != hipSuccess) return gfail(grp, GM_ERR_DEVICE, "hipSetDevice failed");
            if (sl.last.n_valid == 0) continue;
            // queries go through the rank's own centroid buffer in pieces of at most its capacity (its own centroids
            // were fetched by the merge already); results through its nearest-point scratch
            for (uint32_t q0 = 0; q0 < V; q0 += sl.cap) {
                const uint32_t nq = V - q0 < sl.cap ? V - q0 : sl.cap;
                if (hipMemcpyAsync(sl.vox4, &G.vox_cen[(size_t)q0 * 4], (size_t)nq * 16, hipMemcpyHostToDevice, sl.stream) != hipSuccess)
                    return gfail(grp, GM_ERR_DEVICE, "gm_group: upload of the merged centroids failed");
                launch_nearest(sl.valid4, &sl.ctr->n_valid, sl.cap, sl.vox4, nullptr, nq, sl.nn_best, sl.vox_nn, sl.stream, sl.vnorm4, sl.vox_nrm4);
                hipLaunchKernelGGL(k_gather_nearest, dim3((nq + 255) / 256), dim3(256), 0, sl.stream, (const int32_t *)sl.vox_nn, nq,
                                   (const float4 *)sl.valid4, sl.spts4);
                keys.resize(nq); nrm.resize((size_t)nq * 4); pts.resize((size_t)nq * 4);
                if (hipMemcpyAsync(keys.data(), sl.nn_best, (size_t)nq * 8, hipMemcpyDeviceToHost, sl.stream) != hipSuccess ||
                    hipMemcpyAsync(nrm.data(), sl.vox_nrm4, (size_t)nq * 16, hipMemcpyDeviceToHost, sl.stream) != hipSuccess ||
                    hipMemcpyAsync(pts.data(), sl.spts4, (size_t)nq * 16, hipMemcpyDeviceToHost, sl.stream) != hipSuccess ||
                    hipStreamSynchronize(sl.stream) != hipSuccess)
                    return gfail(grp, GM_ERR_DEVICE, "gm_group: nearest-point search failed");
                for (uint32_t i = 0; i < nq; ++i) {
                    if (keys[i] == ~0ull) continue;
                    uint32_t local;
                    memcpy(&local, &pts[4 * (size_t)i + 3], 4);
                    if (local >= G.row_ids[r].size()) return gfail(grp, GM_ERR_DEVICE, "gm_group: a rank's row index is out of range");
                    const unsigned long long k = (keys[i] & 0xFFFFFFFF00000000ull) | G.row_ids[r][local];
                    if (k < best_key[q0 + i]) {
                        best_key[q0 + i] = k;
                        memcpy(&G.vox_nrm[4 * (size_t)(q0 + i)], &nrm[4 * (size_t)i], 16);
                    }
                }
            }
        }
        // the winners' positions in the merged /choppedCloud (gm_group_get_voxel_nearest): rank of their input row
        // among the valid rows
        G.vox_near.assign(V, -1);
        {
            std::vector<uint32_t> valid_rows;
            valid_rows.reserve(G.last.n_valid);
            std::vector<float> buf;
            for (uint32_t r = 0; r < G.n; ++r) {
                uint32_t m = G.ctx[r]->slots[0].last.n_valid;
                buf.resize((size_t)(m ? m : 1) * 4);
                const gm_status st = gm_get_cropped_xyz(G.ctx[r], 0, buf.data(), m ? m : 1, &m);
                if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
                for (uint32_t i = 0; i < m; ++i) {
                    uint32_t local;
                    memcpy(&local, &buf[4 * (size_t)i + 3], 4);
                    if (local < G.row_ids[r].size()) valid_rows.push_back(G.row_ids[r][local]);
                }
            }
            std::sort(valid_rows.begin(), valid_rows.end());
            for (uint32_t i = 0; i < V; ++i) {
                if (best_key[i] == ~0ull) continue;
                const uint32_t row = (uint32_t)(best_key[i] & 0xFFFFFFFFull);
                G.vox_near[i] = (int32_t)(std::lower_bound(valid_rows.begin(), valid_rows.end(), row) - valid_rows.begin());
            }
        }
        G.vox_nrm_valid = true;
    }
    memcpy(nxyzc, G.vox_nrm.data(), (size_t)V * 16);
    return GM_OK;
}

gm_status gm_group_get_voxel_nearest(gm_group *grp, int32_t *idx, uint32_t capacity, uint32_t *n_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (!G.have_frame) { if (n_out) *n_out = 0; return gfail(grp, GM_ERR_NOT_READY, "gm_group: no completed sharded frame"); }
    const uint32_t V = (uint32_t)(G.vox_cen.size() / 4);
    if (n_out) *n_out = V;
    if (V > capacity) return gfail(grp, GM_ERR_CAPACITY, "output buffer too small");
    if (!V) return GM_OK;
    if (!idx) return gfail(grp, GM_ERR_INVALID_ARG, "output pointer is NULL");
    if (!G.vox_nrm_valid) {
        std::vector<float> tmp((size_t)V * 4);
        const gm_status st = gm_group_get_voxel_normals(grp, tmp.data(), V, nullptr);
        if (st != GM_OK) return st;
    }
    memcpy(idx, G.vox_near.data(), (size_t)V * 4);
    return GM_OK;
}

// ---- streaming: whole frames, round-robin over the group's devices ---------------------------------------------------
// (BASELINE configs[4]; replaces the one-frame-at-a-time consumer of src/geometric_mapping.cpp:146,169.)

uint32_t gm_group_in_flight(const gm_group *grp) { return grp ? (uint32_t)grp->inflight.size() : 0u; }

gm_status gm_group_submit_frame(gm_group *grp, const gm_cloud *cloud)
{
    if (!grp || !cloud) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (G.dead) return gfail(grp, GM_ERR_COMM, "gm_group: an earlier collective failed; destroy the group");
    uint32_t cap = 0;
    for (uint32_t r = 0; r < G.n; ++r) cap += G.ctx[r]->n_slots;
    if (G.inflight.size() >= cap) return gfail(grp, GM_ERR_NOT_READY, "gm_group_submit_frame: every slot holds a frame (gm_group_wait_frame first)");
    // the next device in turn that has a free slot (slots of a rank are used in ring order, so the oldest frees first)
    for (uint32_t tries = 0; tries < G.n; ++tries) {
        const uint32_t r = (G.next_rank + tries) % G.n;
        uint32_t used = 0;
        for (const gm_group::Ticket &t : G.inflight) used += t.rank == r ? 1u : 0u;
        if (used >= G.ctx[r]->n_slots) continue;
        const uint32_t slot = G.next_slot[r];
        // a sharded frame (also one that failed half-way) leaves the ranks with slab ranges: rank 0 owns (-inf, edge[1]),
        // so BOTH ends have to be looked at -- a streamed frame is a whole frame, every point of it is the rank's own
        if (G.ctx[r]->own_lo != -std::numeric_limits<double>::infinity() ||
            G.ctx[r]->own_hi != std::numeric_limits<double>::infinity()) {
            const gm_status st = gm_set_owned_range(G.ctx[r], -std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity());
            if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
        }
        const gm_status st = gm_submit_frame(G.ctx[r], slot, cloud);
        if (st != GM_OK) return gfail(grp, st, std::string("rank ") + std::to_string(r) + ": " + gm_last_error(G.ctx[r]));
        G.have_frame = false;   // slot 0 of the ranks no longer holds a sharded frame
        G.inflight.push_back(gm_group::Ticket{r, slot});
        G.next_slot[r] = (slot + 1) % G.ctx[r]->n_slots;
        G.next_rank = (r + 1) % G.n;
        return GM_OK;
    }
    return gfail(grp, GM_ERR_NOT_READY, "gm_group_submit_frame: no free slot");
}

gm_status gm_group_poll_frame(gm_group *grp)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (G.inflight.empty()) return GM_ERR_NOT_READY;
    const gm_group::Ticket t = G.inflight.front();
    const gm_status st = gm_poll_frame(G.ctx[t.rank], t.slot);
    if (st != GM_OK && st != GM_ERR_NOT_READY) return gfail(grp, st, std::string("rank ") + std::to_string(t.rank) + ": " + gm_last_error(G.ctx[t.rank]));
    return st;
}

gm_status gm_group_wait_frame(gm_group *grp, gm_frame_result *res, uint32_t *rank_out, uint32_t *slot_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (G.inflight.empty()) return gfail(grp, GM_ERR_NOT_READY, "gm_group_wait_frame: no frame in flight");
    const gm_group::Ticket t = G.inflight.front();
    G.inflight.pop_front();
    if (rank_out) *rank_out = t.rank;
    if (slot_out) *slot_out = t.slot;
    const gm_status st = gm_wait_frame(G.ctx[t.rank], t.slot, res);
    if (st != GM_OK) return gfail(grp, st, std::string("rank ") + std::to_string(t.rank) + ": " + gm_last_error(G.ctx[t.rank]));
    return GM_OK;
}

}  // extern "C"
