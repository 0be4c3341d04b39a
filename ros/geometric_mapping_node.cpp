// geometric_mapping_node.cpp -- catkin/ROS host for libgm_hip.so.
//
// Keeps the reference node's surface exactly (so launch/mapping.launch and
// rviz_config/mapping.rviz of the reference work unchanged):
//   node name  geometric_mapping_node                       (src/geometric_mapping.cpp:130)
//   sub        "input"            sensor_msgs/PointCloud2, queue 1       (:146)
//   pub        "cloudOutput"      sensor_msgs/PointCloud2, queue 10, if displayCloud      (:149-151)
//   pub        "normalsOutput"    visualization_msgs/MarkerArray, queue 10, if displayNormals    (:154-156)
//   pub        "eigenBasisOutput" visualization_msgs/MarkerArray, queue 10, if displayCenterAxis (:159-161)
//   params     boxFilterBound voxelGridLeafSize neighborRadius weightingFactor
//              displayCloud displayNormals displayCenterAxis usePCLViz   (src/paramHandler.cpp:12-66)
//   pub        "centerAxisOutput" visualization_msgs/Marker, queue 10, if displayCylinder -- the reference declares
//              this publisher and parameter but leaves them commented out (:41,119-121,163-165,
//              src/paramHandler.cpp:55-58, launch/mapping.launch:15 sets displayCylinder=false); here the
//              parameter is honoured: it switches the RANSAC cylinder fit on and publishes one CYLINDER marker
// and replaces the PCL/Eigen arithmetic of cloud_cb (:48-125) with ONE call into
// the C ABI (gm_process_frame reads the PointCloud2 rows directly: no pcl::fromROSMsg).
//
// ROS is absent from the build image: here the node is only TYPE-CHECKED against minimal
// stand-in headers (tests/stubs/, tests/test_ros_node_compiles.py).  Build it for real in a catkin workspace with
// ros/CMakeLists.txt.
// Deliberate differences from the reference, all on non-default paths:
//   * usePCLViz is accepted and ignored with a warning (the reference stores the
//     address of a block-local PCLVisualizer: dangling, src/geometric_mapping.cpp:140-143);
//   * nothing is heap-allocated per frame (the reference leaks 5 objects per callback);
//   * 1-NN for /surfaceNormals searches the compacted cloud (the reference's kd-tree
//     still indexes the pre-compaction cloud, src/tunnel_processing.cpp:65,85,239);
//   * with displayNormals the context is created with GM_CFG_NEAREST: the voxel grid, the 1-NN and the gather of
//     the nearest points' normals all happen inside the one frame pass, the node fetches V centroids + V normals.
// Two parameters of its own, both off by default (the reference's launch file then behaves as before):
//   gmGraphReplay  bool    replay the frame's launch chain from a captured hipGraph (GM_CFG_GRAPH).  Pays for clouds of
//                          a steady size; a lidar whose point count wanders across 1/16-wide size buckets re-captures
//                          (hipStreamBeginCapture + hipGraphInstantiate) inside the callback, so it is a choice
//   gmDevices      string  "0,1,2,...": more than one device streams the frames round-robin over them
//                          (gm_group_submit_frame / gm_group_poll_frame / gm_group_wait_frame) -- a pipeline a few frames
//                          deep instead of the one-frame-at-a-time consumer of src/geometric_mapping.cpp:146,169.  Every
//                          frame is still published, in order, as soon as it has finished: the callback first publishes
//                          whatever is done (a completion query, no waiting), a 2 ms timer does the same between
//                          callbacks, and the frames still in flight at shutdown are waited for and published.  A
//                          frame's header travels with it; a frame that fails is dropped WITH its header.
// With displayCloud (the launch file's default, launch/mapping.launch:12) /choppedCloud is copied into page-locked host
// rows while the rest of the frame still runs (gm_set_cloud_output), not fetched afterwards.
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/PointField.h>
#include <visualization_msgs/MarkerArray.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "../host/gm_tunnel_processing.hpp"

namespace {

struct Parameters {  // mirrors class Parameters, include/geometric_mapping/paramHandler.hpp:9-37
    double boxFilterBound = 5.0, leafSize = .1, neighborRadius = .03, weightingFactor = .2;
    bool rvizCloud = true, rvizNormals = true, rvizCenterAxis = true, pclviz = false, rvizCylinder = false;
    bool graphReplay = false;          // gmGraphReplay
    std::vector<int> devices;          // gmDevices (empty: device 0, one frame at a time)
    explicit Parameters(ros::NodeHandle &node)
    {
        // global names, read once (src/paramHandler.cpp:13-65); log text kept
        if (node.getParam("boxFilterBound", boxFilterBound)) ROS_INFO("boxFilterBound set to: \t %f", boxFilterBound);
        else ROS_INFO("ERROR: boxFilterBound set to default");
        if (node.getParam("voxelGridLeafSize", leafSize)) ROS_INFO("voxelGridLeafSize set to: \t %f", leafSize);
        else ROS_INFO("ERROR: voxelGridLeafSize set to default");
        if (node.getParam("neighborRadius", neighborRadius)) ROS_INFO("neighborRadius set to: \t %f", neighborRadius);
        else ROS_INFO("ERROR: neighborRadius set to default");
        if (node.getParam("weightingFactor", weightingFactor)) ROS_INFO("weightingFactor set to: \t %f", weightingFactor);
        else ROS_INFO("ERROR: weightingFactor set to default");
        node.getParam("displayCloud", rvizCloud);
        node.getParam("displayNormals", rvizNormals);
        node.getParam("displayCenterAxis", rvizCenterAxis);
        node.getParam("usePCLViz", pclviz);
        node.getParam("displayCylinder", rvizCylinder);
        node.getParam("gmGraphReplay", graphReplay);
        std::string dev;
        if (node.getParam("gmDevices", dev)) {
            for (size_t i = 0; i < dev.size();) {
                size_t j = dev.find(',', i);
                if (j == std::string::npos) j = dev.size();
                if (j > i) devices.push_back(std::atoi(dev.substr(i, j - i).c_str()));
                i = j + 1;
            }
        }
    }
};

std::unique_ptr<Parameters> params;
std::unique_ptr<gm_host::Processor> proc;
std::deque<std_msgs::Header> headers;   // of the frames in flight (streaming over several devices)
ros::Publisher cloudPub, normalsPub, centerAxisPub, cylinderPub;

bool find_xyz(const sensor_msgs::PointCloud2 &m, unsigned &ox, unsigned &oy, unsigned &oz)
{
    bool fx = false, fy = false, fz = false;
    for (const sensor_msgs::PointField &f : m.fields) {
        if (f.datatype != sensor_msgs::PointField::FLOAT32) continue;
        if (f.name == "x") { ox = f.offset; fx = true; }
        else if (f.name == "y") { oy = f.offset; fy = true; }
        else if (f.name == "z") { oz = f.offset; fz = true; }
    }
    return fx && fy && fz;
}

visualization_msgs::Marker to_ros(const gm_host::Marker &s)
{
    visualization_msgs::Marker m;
    m.header.frame_id = s.frame_id;
    m.header.stamp = ros::Time::now();  // src/tunnel_processing.cpp:175 (not the input stamp)
    m.header.seq = 0;
    m.ns = s.ns; m.id = s.id;
    m.type = s.type;                    // ARROW = 0, CYLINDER = 3: same values as visualization_msgs::Marker
    m.action = visualization_msgs::Marker::ADD;
    if (s.type == gm_host::MARKER_ARROW) {
        m.points.resize(2);
        for (int k = 0; k < 2; ++k) { m.points[k].x = s.points[k][0]; m.points[k].y = s.points[k][1]; m.points[k].z = s.points[k][2]; }
    }
    m.pose.position.x = s.position[0]; m.pose.position.y = s.position[1]; m.pose.position.z = s.position[2];
    m.pose.orientation.x = s.orientation[0]; m.pose.orientation.y = s.orientation[1];
    m.pose.orientation.z = s.orientation[2]; m.pose.orientation.w = s.orientation[3];
    m.scale.x = s.scale[0]; m.scale.y = s.scale[1]; m.scale.z = s.scale[2];
    m.color.a = s.color_a; m.color.r = s.color_r; m.color.g = s.color_g; m.color.b = s.color_b;
    return m;
}

visualization_msgs::MarkerArray to_ros(const gm_host::MarkerArray &in)
{
    visualization_msgs::MarkerArray out;
    out.markers.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) out.markers[i] = to_ros(in[i]);
    return out;
}

// everything the reference publishes for one frame (src/geometric_mapping.cpp:94-124); the processor's accessors refer
// to the frame `r` came from
void publish_frame(const gm_frame_result &r, const std_msgs::Header &header)
{
    gm_host::Vector3f vals = {{r.eigenvalues[0], r.eigenvalues[1], r.eigenvalues[2]}};
    gm_host::Matrix3f vecs;
    std::memcpy(vecs.m, r.eigenvectors, sizeof(vecs.m));
    ROS_INFO("Center Axis found...");
    if (params->rvizCloud) {  // :100-107, xyz-only PointCloud2 as pcl::toROSMsg lays it out (16-byte points)
        const gm_host::PointCloud cloud = proc->choppedCloud();   // (already on the host: enableCloudOutput)
        sensor_msgs::PointCloud2 out;
        out.header = header;
        out.height = 1; out.width = (uint32_t)cloud.size();
        out.is_bigendian = false; out.is_dense = true;
        out.point_step = 16; out.row_step = 16 * out.width;
        out.fields.resize(3);
        const char *names[3] = {"x", "y", "z"};
        for (int k = 0; k < 3; ++k) {
            out.fields[k].name = names[k]; out.fields[k].offset = 4 * k;
            out.fields[k].datatype = sensor_msgs::PointField::FLOAT32; out.fields[k].count = 1;
        }
        out.data.resize((size_t)out.row_step);
        if (!cloud.empty()) std::memcpy(&out.data[0], &cloud[0], out.data.size());
        cloudPub.publish(out);
    }
    if (params->rvizNormals)  // :109-112: centroids + nearest normals were produced inside the frame (GM_CFG_NEAREST)
        normalsPub.publish(to_ros(proc->rvizNormalsFromFrame()));
    if (params->rvizCenterAxis)  // :114-117
        centerAxisPub.publish(to_ros(gm_host::Processor::rvizEigens(vals, vecs)));
    if (params->rvizCylinder) {  // the reference's commented-out block :119-121
        gm_host::Marker cyl;
        if (gm_host::Processor::rvizCylinder(r, 2.0 * params->boxFilterBound, cyl)) cylinderPub.publish(to_ros(cyl));
    }
    ROS_INFO("Published...");
}

// Streaming: publishes the frames in flight that have finished, oldest first; with `block` every frame in flight.  A frame
// leaves the queue together with its header whether it succeeded or not (the library has already dropped it).
void publish_finished(bool block)
{
    while (proc && proc->inFlight() && !headers.empty()) {
        gm_frame_result r;
        const std_msgs::Header h = headers.front();
        try {
            if (block) r = proc->waitFrame();
            else if (!proc->tryWaitFrame(r)) return;      // the oldest is still running: nothing was taken off the queue
        } catch (const gm_host::Error &e) {
            headers.pop_front();
            ROS_ERROR("libgm_hip: %s", e.what());
            continue;
        }
        headers.pop_front();
        publish_frame(r, h);
    }
}

void timer_cb(const ros::TimerEvent &) { publish_finished(false); }

void cloud_cb(const sensor_msgs::PointCloud2ConstPtr &input)
{
    ROS_INFO("Callback started...");
    unsigned ox = 0, oy = 4, oz = 8;
    if (!find_xyz(*input, ox, oy, oz)) { ROS_ERROR("input cloud has no float32 x/y/z fields"); return; }
    const unsigned n = input->width * input->height;
    // fromROSMsg + chopCloud + getNormals + VoxelGrid + getLocalFrame: one device pass (:55-92)
    const unsigned char *rows = n ? &input->data[0] : nullptr;
    std::vector<unsigned char> packed;
    if (n && input->height > 1 && input->row_step != input->width * input->point_step) {
        // organised cloud with padded rows: pcl::fromROSMsg honours row_step, the C ABI takes point_step-strided
        // rows -- drop the padding once on the host
        packed.resize((size_t)n * input->point_step);
        for (uint32_t y = 0; y < input->height; ++y)
            std::memcpy(&packed[(size_t)y * input->width * input->point_step], &input->data[(size_t)y * input->row_step],
                        (size_t)input->width * input->point_step);
        rows = &packed[0];
    }
    if (params->devices.size() > 1) {
        // streaming: publish what has finished, make room if every slot is taken, hand the frame to the next device
        publish_finished(false);
        try {
            if (proc->inFlight() >= proc->capacity() && !headers.empty()) {
                const std_msgs::Header h = headers.front();
                headers.pop_front();                       // (waitFrame takes the frame off the queue, also when it throws)
                const gm_frame_result r = proc->waitFrame();
                publish_frame(r, h);
            }
            proc->submitFrame(rows, n, input->point_step, ox, oy, oz, input->is_bigendian);
            headers.push_back(input->header);
        } catch (const gm_host::Error &e) {
            ROS_ERROR("libgm_hip: %s", e.what());
        }
        return;
    }
    gm_frame_result r;
    try {
        r = proc->processFrame(rows, n, input->point_step, ox, oy, oz, input->is_bigendian);
    } catch (const gm_host::Error &e) {
        ROS_ERROR("libgm_hip: %s", e.what());
        return;
    }
    ROS_INFO("Box filter applied...");
    ROS_INFO("Surface normals found...");
    publish_frame(r, input->header);
}

}  // namespace

int main(int argc, char **argv)
{
    ros::init(argc, argv, "geometric_mapping_node");
    ros::NodeHandle node;
    params.reset(new Parameters(node));
    ROS_INFO("Launched geometric_mapping_node...");
    if (params->pclviz) ROS_WARN("usePCLViz is ignored by the MI355X host (no PCL in this build)");
    try {
        const unsigned flags = GM_CFG_DEFAULT | (params->rvizCylinder ? GM_CFG_RANSAC_CYLINDER : 0u) |
                               (params->rvizNormals ? GM_CFG_NEAREST : 0u) | (params->graphReplay ? GM_CFG_GRAPH : 0u);
        if (params->devices.size() > 1)
            proc.reset(new gm_host::Processor(params->boxFilterBound, params->leafSize, params->neighborRadius,
                                              params->weightingFactor, params->devices, flags, 2));
        else
            proc.reset(new gm_host::Processor(params->boxFilterBound, params->leafSize, params->neighborRadius,
                                              params->weightingFactor, params->devices.empty() ? 0 : params->devices[0], flags));
    } catch (const gm_host::Error &e) {
        ROS_FATAL("libgm_hip: %s", e.what());
        return 1;
    }
    if (params->rvizCloud) {
        try { proc->enableCloudOutput(1u << 20); }   // grown by the first larger frame
        catch (const gm_host::Error &e) { ROS_WARN("libgm_hip: %s (the cloud is fetched after the frame instead)", e.what()); }
    }
    ros::Subscriber sub = node.subscribe("input", 1, cloud_cb);
    ros::Timer timer;
    if (params->devices.size() > 1) timer = node.createTimer(ros::Duration(0.002), timer_cb);
    if (params->rvizCloud) cloudPub = node.advertise<sensor_msgs::PointCloud2>("cloudOutput", 10);
    if (params->rvizNormals) normalsPub = node.advertise<visualization_msgs::MarkerArray>("normalsOutput", 10);
    if (params->rvizCenterAxis) centerAxisPub = node.advertise<visualization_msgs::MarkerArray>("eigenBasisOutput", 10);
    if (params->rvizCylinder) cylinderPub = node.advertise<visualization_msgs::Marker>("centerAxisOutput", 10);  // :165
    ros::spin();
    publish_finished(true);   // the frames still in flight at shutdown
    proc.reset();
    return 0;
}
