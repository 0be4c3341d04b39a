// gm_group.hip -- one host thread driving every local GPU: spatial sharding of ONE frame across the devices of a node
// with an in-library RCCL all-gather of the fitted records (SURVEY.md par. 8b "Threading", par. 8e; north_star:
// "Frames shard spatially across the 8 GPUs of one node with an RCCL all-gather of fitted primitives over xGMI only
// when a scan exceeds single-GPU capacity").
//
// Replaces the single-threaded ros::spin() design of /root/reference src/geometric_mapping.cpp:169 for scans beyond one
// GPU; the reference has no counterpart (no parallelism of any kind, SURVEY.md par. 2).
//
// The path shards because every per-point stage depends only on points within neighborRadius and the final fit is a
// SUM (M = sum w^2 n n^T) followed by a 3x3 solve.  Rank g owns the points with x in [edge[g], edge[g+1]) and also
// receives halo points within 1.01 r of its edges -- neighbours only, never outputs (gm_set_owned_range).  The cut is
// made on the host before H2D, so no device-to-device halo exchange exists.  The only exchange step of the path is one
// ncclAllGather of a 24-double record per rank (scatter partials, counts, the rank's fitted plane / cylinder): latency
// bound, a few hundred bytes over xGMI.  RCCL is loaded at run time (dlopen) so that a single-GPU host needs no librccl.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <new>
#include <string>
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
        // geometric_mapping_amd/_lib.py names it in GM_RCCL_PATH.  A C++ host takes the system library.
        const char *names[] = {getenv("GM_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
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

inline float load_f32(const uint8_t *p, bool bswap)
{
    uint32_t u;
    memcpy(&u, p, 4);
    if (bswap) u = __builtin_bswap32(u);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

struct gm_group {
    gm_config cfg;
    uint32_t flags = 0;
    uint32_t n = 0;
    bool loopback = false;           // ranks share a device: records travel by device copies, no communicator
    std::vector<int> devices;
    std::vector<gm_ctx *> ctx;
    std::vector<double *> d_rec;     // [n] device: this rank's record
    std::vector<double *> d_all;     // [n] device: every rank's record (the all-gather's receive buffer)
    std::vector<hipEvent_t> ev;      // loopback: "record packed" per rank
    double *h_all = nullptr;         // pinned host copy of rank 0's gathered buffer
    std::vector<ncclComm_t> comms;
    Rccl rccl;
    std::vector<std::vector<uint8_t>> rows;      // per-rank row buffers (host side of the slab cut)
    std::vector<std::vector<uint32_t>> row_ids;  // per-rank: input row of every row sent
    std::vector<double> edges;
    gm_frame_result last = {};
    std::string err;
};

namespace {
thread_local std::string g_group_create_err;

gm_status gfail(gm_group *g, gm_status st, const std::string &msg)
{
    if (g) g->err = msg; else g_group_create_err = msg;
    return st;
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
        if (r < grp->ctx.size()) gm_destroy(grp->ctx[r]);
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
    g->rows.resize(n_ranks); g->row_ids.resize(n_ranks);
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

// x-slab edges balanced by the count of in-box points; first = -inf, last = +inf; float32-representable (ownership is
// tested in fp32 on the device)
static void slab_edges(const std::vector<float> &x_inside, uint32_t n_slabs, std::vector<double> &edges)
{
    std::vector<float> xs(x_inside);
    std::sort(xs.begin(), xs.end());
    edges.assign(n_slabs + 1, 0.0);
    edges[0] = -std::numeric_limits<double>::infinity();
    edges[n_slabs] = std::numeric_limits<double>::infinity();
    for (uint32_t g = 1; g < n_slabs; ++g) {
        if (xs.empty()) { edges[g] = 0.0; continue; }
        const size_t k = std::min(xs.size() - 1, (xs.size() * (size_t)g) / n_slabs);
        edges[g] = (double)xs[k];
    }
}

gm_status gm_group_process_frame(gm_group *grp, const gm_cloud *cloud, gm_frame_result *res)
{
    if (!grp || !cloud) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
    if (cloud->flags & GM_CLOUD_DEVICE) return gfail(grp, GM_ERR_UNSUPPORTED, "gm_group_process_frame cuts the slabs on the host: pass host rows");
    const uint32_t n = cloud->n_points;
    const uint64_t step = cloud->point_step;
    if (n && !cloud->data) return gfail(grp, GM_ERR_INVALID_ARG, "gm_cloud.data is NULL");
    if (n && (step < 12 || (uint64_t)cloud->off_x + 4 > step || (uint64_t)cloud->off_y + 4 > step || (uint64_t)cloud->off_z + 4 > step))
        return gfail(grp, GM_ERR_INVALID_ARG, "gm_cloud: x/y/z offsets do not fit in point_step");
    const bool bswap = (cloud->flags & GM_CLOUD_BIGENDIAN) != 0;
    const uint8_t *base = (const uint8_t *)cloud->data;
    // ---- the host side of the cut: x of every row, the in-box rows fix the edges
    const float lo = (float)(-G.cfg.boxFilterBound), hi = (float)G.cfg.boxFilterBound;
    std::vector<float> xs(n), x_in;
    x_in.reserve(n);
    uint32_t n_inside = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *row = base + (size_t)i * step;
        const float x = load_f32(row + cloud->off_x, bswap), y = load_f32(row + cloud->off_y, bswap), z = load_f32(row + cloud->off_z, bswap);
        xs[i] = x;
        if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z) && !(x < lo || y < lo || z < lo || x > hi || y > hi || z > hi)) {
            x_in.push_back(x);
            ++n_inside;
        }
    }
    slab_edges(x_in, G.n, G.edges);
    const double halo = 1.01 * G.cfg.neighborRadius;
    for (uint32_t r = 0; r < G.n; ++r) {
        std::vector<uint8_t> &buf = G.rows[r];
        std::vector<uint32_t> &ids = G.row_ids[r];
        buf.clear(); ids.clear();
        const double a = G.edges[r] - halo, b = G.edges[r + 1] + halo;
        for (uint32_t i = 0; i < n; ++i) {
            const double x = (double)xs[i];
            if (x >= a && x < b) ids.push_back(i);   // (NaN rows fail both comparisons: they are dropped by every rank's crop anyway)
        }
        buf.resize(ids.size() * (size_t)step);
        for (size_t k = 0; k < ids.size(); ++k) memcpy(&buf[k * step], base + (size_t)ids[k] * step, step);
    }
    // ---- every rank runs the unchanged single-GPU pipeline on its slab, asynchronously
    for (uint32_t r = 0; r < G.n; ++r) {
        gm_status st = gm_set_owned_range(G.ctx[r], G.edges[r], G.edges[r + 1]);
        if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
        gm_cloud c = *cloud;
        c.data = G.rows[r].empty() ? nullptr : G.rows[r].data();
        c.n_points = (uint32_t)G.row_ids[r].size();
        c.flags = cloud->flags & GM_CLOUD_BIGENDIAN;
        st = gm_submit_frame(G.ctx[r], 0, &c);
        if (st != GM_OK) return gfail(grp, st, std::string("rank ") + std::to_string(r) + ": " + gm_last_error(G.ctx[r]));
        Slot &sl = G.ctx[r]->slots[0];
        hipLaunchKernelGGL(k_pack_record, dim3(1), dim3(64), 0, sl.stream, (const FrameOut *)sl.d_out, c.n_points, G.d_rec[r]);
        if (G.loopback) hipEventRecord(G.ev[r], sl.stream);
    }
    // ---- the one exchange step: all-gather of the per-rank records
    if (!G.loopback) {
        ncclResult_t rc = G.rccl.GroupStart();
        for (uint32_t r = 0; r < G.n && rc == ncclSuccess; ++r) {
            hipSetDevice(G.devices[r]);
            rc = G.rccl.AllGather(G.d_rec[r], G.d_all[r], kRecLen, ncclDouble, G.comms[r], G.ctx[r]->slots[0].stream);
        }
        const ncclResult_t rc2 = G.rccl.GroupEnd();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) return gfail(grp, GM_ERR_COMM, std::string("ncclAllGather: ") + G.rccl.GetErrorString(rc));
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
        return gfail(grp, GM_ERR_DEVICE, "D2H of the gathered records failed");
    std::vector<gm_frame_result> rr(G.n);
    for (uint32_t r = 0; r < G.n; ++r) {
        const gm_status st = gm_wait_frame(G.ctx[r], 0, &rr[r]);
        if (st != GM_OK) return gfail(grp, st, std::string("rank ") + std::to_string(r) + ": " + gm_last_error(G.ctx[r]));
    }
    hipSetDevice(G.devices[0]);
    if (hipStreamSynchronize(G.ctx[0]->slots[0].stream) != hipSuccess) return gfail(grp, GM_ERR_DEVICE, "stream sync failed");
    // ---- merge (every rank holds the same gathered records; rank 0's copy is read)
    gm_frame_result out;
    memset(&out, 0, sizeof(out));
    out.n_in = n;
    out.n_cropped = n_inside;
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
    // voxels cut by a slab edge appear in two ranks: count distinct lattice cells (centroids are a few KB per rank)
    if (G.cfg.flags & GM_CFG_VOXEL_GRID) {
        std::vector<long long> keys;
        std::vector<float> cen;
        const float inv = 1.0f / (float)G.cfg.voxelGridLeafSize;
        for (uint32_t r = 0; r < G.n; ++r) {
            uint32_t v = rr[r].n_voxels, got = 0;
            cen.resize((size_t)(v ? v : 1) * 4);
            const gm_status st = gm_get_voxel_centroids(G.ctx[r], 0, cen.data(), v ? v : 1, &got);
            if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
            for (uint32_t i = 0; i < got; ++i) {
                const long long ix = (long long)floorf(cen[4 * i] * inv), iy = (long long)floorf(cen[4 * i + 1] * inv),
                                iz = (long long)floorf(cen[4 * i + 2] * inv);
                keys.push_back(((iz & 0x1FFFFF) << 42) | ((iy & 0x1FFFFF) << 21) | (ix & 0x1FFFFF));
            }
        }
        std::sort(keys.begin(), keys.end());
        out.n_voxels = (uint32_t)(std::unique(keys.begin(), keys.end()) - keys.begin());
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
        std::vector<long long> total(owner.size(), 0);
        std::vector<int32_t> cnt(owner.size());
        for (uint32_t r = 0; r < G.n; ++r) {
            const gm_status st = gm_score_frame(G.ctx[r], 0, model, cand.data(), (uint32_t)owner.size(), G.cfg.ransac_threshold, 0, cnt.data());
            if (st != GM_OK) return gfail(grp, st, gm_last_error(G.ctx[r]));
            for (size_t k = 0; k < owner.size(); ++k) total[k] += cnt[k];
        }
        size_t best = 0;
        for (size_t k = 1; k < owner.size(); ++k) if (total[k] > total[best]) best = k;
        const gm_frame_result &src = rr[owner[best]];
        if (model == 0) {
            out.plane_inliers = (uint32_t)total[best];
            for (int k = 0; k < 4; ++k) { out.plane[k] = src.plane[k]; out.plane_refit[k] = src.plane_refit[k]; }
        } else {
            out.cylinder_inliers = (uint32_t)total[best];
            for (int k = 0; k < 7; ++k) out.cylinder[k] = src.cylinder[k];
            for (int k = 0; k < 3; ++k) out.cylinder_axis_refit[k] = src.cylinder_axis_refit[k];
        }
    }
    float km = 0.f;
    for (uint32_t r = 0; r < G.n; ++r) km = fmaxf(km, rr[r].normals_kernel_ms);
    out.normals_kernel_ms = km;
    G.last = out;
    if (res) *res = out;
    return GM_OK;
}

// /choppedCloud of a sharded frame: every rank's valid cloud, merged back into the single-GPU order (ascending input
// row).  Rows x,y,z,pad with pad = the row index in the ORIGINAL cloud.
gm_status gm_group_get_cropped_xyz(gm_group *grp, float *xyzw, uint32_t capacity, uint32_t *n_out)
{
    if (!grp) return GM_ERR_INVALID_ARG;
    gm_group &G = *grp;
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

}  // extern "C"
