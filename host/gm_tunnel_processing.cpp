// gm_tunnel_processing.cpp -- see gm_tunnel_processing.hpp.  Plain C++11; links libgm_hip.so.
#include "gm_tunnel_processing.hpp"

#include <cmath>
#include <cstring>

namespace gm_host {

void Processor::check(gm_status s, const char *what)
{
    if (s != GM_OK) throw Error(s, std::string(what) + ": " + gm_status_string(s) + ": " + gm_last_error(ctx_));
}

Processor::Processor(double b, double leaf, double r, double wf, int device, unsigned flags)
    : ctx_(0), grp_(0), cur_(0), cur_slot_(0), n_slots_(1), next_slot_(0), pending_(0)
{
    gm_config cfg;
    gm_default_config(&cfg);
    cfg.boxFilterBound = b; cfg.voxelGridLeafSize = leaf; cfg.neighborRadius = r; cfg.weightingFactor = wf;
    cfg.device = device; cfg.flags = flags;
    gm_status s = gm_create(&cfg, &ctx_);
    if (s != GM_OK) throw Error(s, std::string("gm_create: ") + gm_last_error(0));
    cur_ = ctx_;
    std::memset(&last_, 0, sizeof(last_));
}

Processor::Processor(double b, double leaf, double r, double wf, const std::vector<int> &devices, unsigned flags,
                     unsigned slots_per_device)
    : ctx_(0), grp_(0), cur_(0), cur_slot_(0), n_slots_(slots_per_device ? slots_per_device : 1), next_slot_(0), pending_(0)
{
    if (devices.empty()) throw Error(GM_ERR_INVALID_ARG, "Processor: empty device list");
    gm_config cfg;
    gm_default_config(&cfg);
    cfg.boxFilterBound = b; cfg.voxelGridLeafSize = leaf; cfg.neighborRadius = r; cfg.weightingFactor = wf;
    cfg.flags = flags; cfg.n_slots = n_slots_;
    std::vector<int32_t> dev(devices.begin(), devices.end());
    gm_status s = gm_group_create(&cfg, &dev[0], (uint32_t)dev.size(), 0u, &grp_);
    if (s != GM_OK) throw Error(s, std::string("gm_group_create: ") + gm_group_last_error(0));
    ctx_ = gm_group_ctx(grp_, 0);
    cur_ = ctx_;
    std::memset(&last_, 0, sizeof(last_));
}

Processor::~Processor()
{
    for (size_t i = 0; i < cloud_bufs_.size(); ++i) {   // (gm_set_cloud_output waits for a frame that still writes the rows)
        gm_set_cloud_output(cloud_bufs_[i].ctx, cloud_bufs_[i].slot, 0, 0);
        gm_host_free(cloud_bufs_[i].ctx, cloud_bufs_[i].rows);
    }
    if (grp_) gm_group_destroy(grp_);   // (owns the contexts)
    else gm_destroy(ctx_);
}

void Processor::enableCloudOutput(unsigned max_points)
{
    if (cloud_bufs_.empty()) {
        const unsigned ranks = grp_ ? gm_group_size(grp_) : 1u;
        for (unsigned r = 0; r < ranks; ++r)
            for (unsigned s = 0; s < n_slots_; ++s) {
                CloudBuf b = {grp_ ? gm_group_ctx(grp_, r) : ctx_, s, 0, 0u};
                cloud_bufs_.push_back(b);
            }
    }
    growCloudOutput(max_points ? max_points : 1u);
}

void Processor::growCloudOutput(unsigned n_points)
{
    for (size_t i = 0; i < cloud_bufs_.size(); ++i) {
        CloudBuf &b = cloud_bufs_[i];
        if (b.cap >= n_points) continue;
        const unsigned cap = n_points + n_points / 4u;
        void *rows = 0;
        gm_status s = gm_host_alloc(b.ctx, (size_t)cap * 16u, &rows);
        if (s != GM_OK) throw Error(s, std::string("enableCloudOutput: ") + gm_last_error(b.ctx));
        s = gm_set_cloud_output(b.ctx, b.slot, (float *)rows, cap);   // (waits for the slot's frame in flight, if any)
        if (s != GM_OK) { gm_host_free(b.ctx, rows); throw Error(s, std::string("enableCloudOutput: ") + gm_last_error(b.ctx)); }
        if (b.rows) gm_host_free(b.ctx, b.rows);
        b.rows = (float *)rows; b.cap = cap;
    }
}

unsigned Processor::capacity() const { return grp_ ? gm_group_size(grp_) * n_slots_ : n_slots_; }
unsigned Processor::inFlight() const { return grp_ ? gm_group_in_flight(grp_) : pending_; }

void Processor::submitFrame(const void *rows, unsigned n, unsigned step, unsigned ox, unsigned oy, unsigned oz, bool bigendian)
{
    gm_cloud c = {rows, n, step, ox, oy, oz, bigendian ? GM_CLOUD_BIGENDIAN : 0u};
    if (!cloud_bufs_.empty()) growCloudOutput(n);
    if (grp_) {
        gm_status s = gm_group_submit_frame(grp_, &c);
        if (s != GM_OK) throw Error(s, std::string("submitFrame: ") + gm_group_last_error(grp_));
        return;
    }
    if (pending_ >= n_slots_) throw Error(GM_ERR_NOT_READY, "submitFrame: every slot holds a frame (waitFrame first)");
    check(gm_submit_frame(ctx_, next_slot_, &c), "submitFrame");
    next_slot_ = (next_slot_ + 1) % n_slots_;
    ++pending_;
}

gm_frame_result Processor::waitFrame()
{
    gm_frame_result r;
    if (grp_) {
        uint32_t rank = 0, slot = 0;
        gm_status s = gm_group_wait_frame(grp_, &r, &rank, &slot);
        if (s != GM_OK) throw Error(s, std::string("waitFrame: ") + gm_group_last_error(grp_));
        cur_ = gm_group_ctx(grp_, rank);
        cur_slot_ = slot;
        last_ = r;
        return r;
    }
    if (!pending_) throw Error(GM_ERR_NOT_READY, "waitFrame: no frame in flight");
    const unsigned slot = (next_slot_ + n_slots_ - pending_) % n_slots_;   // the oldest
    --pending_;   // (the frame leaves the queue whatever the wait returns: a failed frame must not be waited for again)
    check(gm_wait_frame(ctx_, slot, &r), "waitFrame");
    cur_ = ctx_; cur_slot_ = slot;
    last_ = r;
    return r;
}

bool Processor::tryWaitFrame(gm_frame_result &result)
{
    gm_status s;
    if (grp_) {
        if (!gm_group_in_flight(grp_)) return false;
        s = gm_group_poll_frame(grp_);
        if (s != GM_OK && s != GM_ERR_NOT_READY) throw Error(s, std::string("tryWaitFrame: ") + gm_group_last_error(grp_));
    } else {
        if (!pending_) return false;
        s = gm_poll_frame(ctx_, (next_slot_ + n_slots_ - pending_) % n_slots_);
        if (s != GM_OK && s != GM_ERR_NOT_READY) check(s, "tryWaitFrame");
    }
    if (s != GM_OK) return false;
    result = waitFrame();
    return true;
}

PointCloud Processor::chopCloud(const double &bound, const PointCloud &cloud)
{
    gm_cloud c = {cloud.empty() ? 0 : &cloud[0], (uint32_t)cloud.size(), 16, 0, 4, 8, 0};
    PointCloud out(cloud.size() ? cloud.size() : 1);
    uint32_t n = 0;
    check(gm_chop_cloud(ctx_, &c, bound, &out[0].x, (uint32_t)out.size(), &n), "chopCloud");
    out.resize(n);
    return out;
}

NormalCloud Processor::getNormals(const double &radius, PointCloud &cloud)
{
    const size_t n0 = cloud.size();
    std::vector<float> xyz(3 * (n0 ? n0 : 1));
    for (size_t i = 0; i < n0; ++i) { xyz[3 * i] = cloud[i].x; xyz[3 * i + 1] = cloud[i].y; xyz[3 * i + 2] = cloud[i].z; }
    PointCloud oc(n0 ? n0 : 1);
    NormalCloud on(n0 ? n0 : 1);
    uint32_t n = 0;
    check(gm_get_normals_stage(ctx_, &xyz[0], (uint32_t)n0, radius, &oc[0].x, &on[0].normal[0], (uint32_t)oc.size(), &n),
          "getNormals");
    oc.resize(n); on.resize(n);
    cloud.swap(oc);  // ExtractIndices::filter(*cloud): the caller's cloud is compacted in place
    return on;
}

void Processor::getLocalFrame(const int &cloudSize, const double &wf, const NormalCloud &nrm, Vector3f &vals, Matrix3f &vecs)
{
    if (cloudSize < 0 || (size_t)cloudSize > nrm.size())
        throw std::out_of_range("getLocalFrame: cloudSize exceeds the normals cloud");  // the reference's .at() (:106)
    check(gm_get_local_frame(ctx_, nrm.empty() ? 0 : &nrm[0].normal[0], (uint32_t)cloudSize, wf, vals.v, vecs.m, 0),
          "getLocalFrame");
}

PointCloud Processor::voxelGrid(const double &leaf, const PointCloud &cloud)
{
    const size_t n0 = cloud.size();
    std::vector<float> xyz(3 * (n0 ? n0 : 1));
    for (size_t i = 0; i < n0; ++i) { xyz[3 * i] = cloud[i].x; xyz[3 * i + 1] = cloud[i].y; xyz[3 * i + 2] = cloud[i].z; }
    PointCloud out(n0 ? n0 : 1);
    uint32_t n = 0, fl = 0;
    check(gm_voxel_grid(ctx_, &xyz[0], (uint32_t)n0, leaf, &out[0].x, (uint32_t)out.size(), &n, &fl), "voxelGrid");
    out.resize(n);
    return out;
}

std::vector<int> Processor::nearest(const PointCloud &cloud, const PointCloud &queries)
{
    std::vector<float> a(3 * (cloud.size() ? cloud.size() : 1)), q(3 * (queries.size() ? queries.size() : 1));
    for (size_t i = 0; i < cloud.size(); ++i) { a[3 * i] = cloud[i].x; a[3 * i + 1] = cloud[i].y; a[3 * i + 2] = cloud[i].z; }
    for (size_t i = 0; i < queries.size(); ++i) { q[3 * i] = queries[i].x; q[3 * i + 1] = queries[i].y; q[3 * i + 2] = queries[i].z; }
    std::vector<int> idx(queries.size() ? queries.size() : 1);
    check(gm_nearest(ctx_, &a[0], (uint32_t)cloud.size(), &q[0], (uint32_t)queries.size(), &idx[0]), "nearest");
    idx.resize(queries.size());
    return idx;
}

gm_frame_result Processor::processFrame(const void *rows, unsigned n, unsigned step, unsigned ox, unsigned oy, unsigned oz,
                                        bool bigendian)
{
    if (grp_ || pending_) {   // (frames already in flight finish first: results come back in submission order)
        while (inFlight()) waitFrame();
        submitFrame(rows, n, step, ox, oy, oz, bigendian);
        return waitFrame();
    }
    gm_cloud c = {rows, n, step, ox, oy, oz, bigendian ? GM_CLOUD_BIGENDIAN : 0u};
    if (!cloud_bufs_.empty()) growCloudOutput(n);
    gm_frame_result r;
    check(gm_process_frame(ctx_, &c, &r), "processFrame");
    cur_ = ctx_; cur_slot_ = 0;
    last_ = r;
    return r;
}

PointCloud Processor::choppedCloud()
{
    for (size_t i = 0; i < cloud_bufs_.size(); ++i) {   // the rows are already on the host (enableCloudOutput)
        const CloudBuf &b = cloud_bufs_[i];
        if (b.ctx != cur_ || b.slot != cur_slot_ || !b.rows) continue;
        PointCloud out(last_.n_valid);
        if (last_.n_valid) std::memcpy(&out[0].x, b.rows, (size_t)last_.n_valid * sizeof(PointXYZ));
        return out;
    }
    uint32_t n = 0;
    gm_status s = gm_get_cropped_xyz(cur_, cur_slot_, 0, 0, &n);
    if (s != GM_OK && s != GM_ERR_CAPACITY) check(s, "choppedCloud");
    PointCloud out(n ? n : 1);
    check(gm_get_cropped_xyz(cur_, cur_slot_, &out[0].x, (uint32_t)out.size(), &n), "choppedCloud");
    out.resize(n);
    return out;
}

NormalCloud Processor::normals()
{
    uint32_t n = 0;
    gm_status s = gm_get_normals(cur_, cur_slot_, 0, 0, &n);
    if (s != GM_OK && s != GM_ERR_CAPACITY) check(s, "normals");
    NormalCloud out(n ? n : 1);
    check(gm_get_normals(cur_, cur_slot_, &out[0].normal[0], (uint32_t)out.size(), &n), "normals");
    out.resize(n);
    return out;
}

PointCloud Processor::voxelCentroids()
{
    uint32_t n = 0;
    gm_status s = gm_get_voxel_centroids(cur_, cur_slot_, 0, 0, &n);
    if (s != GM_OK && s != GM_ERR_CAPACITY) check(s, "voxelCentroids");
    PointCloud out(n ? n : 1);
    check(gm_get_voxel_centroids(cur_, cur_slot_, &out[0].x, (uint32_t)out.size(), &n), "voxelCentroids");
    out.resize(n);
    return out;
}

NormalCloud Processor::voxelNormals()
{
    uint32_t n = 0;
    gm_status s = gm_get_voxel_normals(cur_, cur_slot_, 0, 0, &n);
    if (s != GM_OK && s != GM_ERR_CAPACITY) check(s, "voxelNormals");
    NormalCloud out(n ? n : 1);
    check(gm_get_voxel_normals(cur_, cur_slot_, &out[0].normal[0], (uint32_t)out.size(), &n), "voxelNormals");
    out.resize(n);
    return out;
}

// ---- marker formatting: src/tunnel_processing.cpp:161-205, 225-256, 260-300 ----

Marker Processor::rvizArrow(const Vector3f &start, const Vector3f &end, const Vector3f &scale, const Vector4f &color,
                            const std::string &ns, const int &id, const std::string &frame)
{
    Marker m;
    m.frame_id = frame;      // :174 (stamp = ros::Time::now() and seq = 0 are set by the ROS shim)
    m.ns = ns; m.id = id;    // :177-178
    m.type = MARKER_ARROW; m.action = MARKER_ADD;  // :179-180
    for (int k = 0; k < 3; ++k) { m.points[0][k] = start(k); m.points[1][k] = end(k); m.scale[k] = scale(k); }
    m.color_a = color(0); m.color_r = color(1); m.color_g = color(2); m.color_b = color(3);  // :199-202: A,R,G,B
    m.position[0] = m.position[1] = m.position[2] = 0.0;   // the reference leaves the pose default-constructed
    m.orientation[0] = m.orientation[1] = m.orientation[2] = 0.0; m.orientation[3] = 1.0;
    return m;
}

bool Processor::rvizCylinder(const gm_frame_result &r, const double &length, Marker &m, const std::string &frame)
{
    const double p[3] = {r.cylinder[0], r.cylinder[1], r.cylinder[2]};
    double d[3] = {r.cylinder[3], r.cylinder[4], r.cylinder[5]};
    const double rad = r.cylinder[6];
    const double dn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(r.cylinder_inliers > 0) || !(dn > 0.0) || !(rad == rad)) return false;   // no cylinder this frame
    for (int k = 0; k < 3; ++k) d[k] /= dn;
    m.frame_id = frame; m.ns = "cylinder"; m.id = 0;
    m.type = MARKER_CYLINDER; m.action = MARKER_ADD;
    const double t = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
    for (int k = 0; k < 3; ++k) { m.position[k] = p[k] - t * d[k]; m.points[0][k] = m.points[1][k] = 0.0; }
    // shortest-arc rotation taking the marker's z axis onto d: q = (z x d, 1 + z.d), normalised
    double q[4] = {-d[1], d[0], 0.0, 1.0 + d[2]};
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (qn < 1e-12) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }   // d = -z: half turn about x
    else for (int k = 0; k < 4; ++k) q[k] /= qn;
    for (int k = 0; k < 4; ++k) m.orientation[k] = q[k];
    m.scale[0] = m.scale[1] = 2.0 * rad; m.scale[2] = length;
    m.color_a = 0.3f; m.color_r = 0.0f; m.color_g = 1.0f; m.color_b = 1.0f;
    return true;
}

MarkerArray Processor::rvizNormals(const double &leafSize, const PointCloud &cloud, const NormalCloud &nrm)
{
    const PointCloud vox = voxelGrid(leafSize, cloud);        // :217-220
    const std::vector<int> idx = nearest(cloud, vox);         // :239 (searched in the compacted cloud)
    MarkerArray out(vox.size());
    const Vector3f scale = {{0.025f, 0.075f, 0.0625f}};       // :230
    const Vector4f color = {{1, 0, 0, 1}};                    // :231
    for (size_t i = 0; i < vox.size(); ++i) {
        const Vector3f s = {{vox[i].x, vox[i].y, vox[i].z}};
        const Normal &nn = nrm.at((size_t)idx[i]);
        const Vector3f e = {{nn.normal[0], nn.normal[1], nn.normal[2]}};   // :247-249: the arrow END is the normal itself
        out[i] = rvizArrow(s, e, scale, color, "normals", (int)i);
    }
    return out;
}

MarkerArray Processor::rvizNormalsFromFrame()
{
    const PointCloud vox = voxelCentroids();                  // :217-220, computed inside the frame
    const NormalCloud vn = voxelNormals();                    // :239 + :247-249, gathered on the device
    MarkerArray out(vox.size());
    const Vector3f scale = {{0.025f, 0.075f, 0.0625f}};       // :230
    const Vector4f color = {{1, 0, 0, 1}};                    // :231
    for (size_t i = 0; i < vox.size(); ++i) {
        const Vector3f s = {{vox[i].x, vox[i].y, vox[i].z}};
        const Normal &nn = vn.at(i);
        const Vector3f e = {{nn.normal[0], nn.normal[1], nn.normal[2]}};
        out[i] = rvizArrow(s, e, scale, color, "normals", (int)i);
    }
    return out;
}

MarkerArray Processor::rvizEigens(const Vector3f &vals, const Matrix3f &vecs)
{
    MarkerArray out(3);
    const float nrm = std::sqrt(vals(0) * vals(0) + vals(1) * vals(1) + vals(2) * vals(2));
    for (int i = 0; i < 3; ++i) {
        const float e = (1.0f / nrm) * std::fabs(vals(i));    // :265
        const Vector3f scale = {{(float)(0.1 - (0.05 * e)), (float)(0.3 - (0.15 * e)), (float)(0.25 - (0.125 * e))}};  // :274-278
        const Vector4f color = {{1.0f, i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f, i == 2 ? 1.0f : 0.0f}};          // :280-287
        const Vector3f zero = {{0, 0, 0}};
        out[i] = rvizArrow(zero, vecs.col(i), scale, color, "eigenBasis", i);   // :289-296
    }
    return out;
}

}  // namespace gm_host
