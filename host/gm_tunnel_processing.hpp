// gm_tunnel_processing.hpp -- ROS-free C++ host mirror of the reference's
// processing + marker functions, implemented on the C ABI of libgm_hip.so.
//
// Same names, argument order and meaning as
//   /root/reference include/geometric_mapping/tunnel_processing.hpp:38-54,66-85
// with PCL / Eigen / ROS message types replaced by plain structs of identical
// layout, so a catkin node can convert at its edges (ros/geometric_mapping_node.cpp)
// and everything below compiles with a bare C++11 compiler -- no HIP headers.
// Unlike the reference nothing here is heap-allocated-and-leaked
// (src/tunnel_processing.cpp:131,135,171,225,261): results are values.
#ifndef GM_TUNNEL_PROCESSING_HPP
#define GM_TUNNEL_PROCESSING_HPP

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/gm_hip.h"

namespace gm_host {

struct PointXYZ { float x, y, z, pad; };                 // = pcl::PointXYZ (16 B)
struct Normal { float normal[3]; float curvature; };      // meaningful fields of pcl::Normal
typedef std::vector<PointXYZ> PointCloud;
typedef std::vector<Normal> NormalCloud;

struct Vector3f { float v[3]; float operator()(int i) const { return v[i]; } float &operator()(int i) { return v[i]; } };
struct Vector4f { float v[4]; float operator()(int i) const { return v[i]; } };
struct Matrix3f {                                         // column-major, like Eigen::Matrix3f
    float m[9];
    float operator()(int r, int c) const { return m[3 * c + r]; }
    Vector3f col(int c) const { Vector3f o = {{m[3 * c], m[3 * c + 1], m[3 * c + 2]}}; return o; }
};

// visualization_msgs/Marker fields the reference fills (src/tunnel_processing.cpp:161-205)
struct Marker {
    std::string frame_id, ns;
    int id, type, action;
    double points[2][3];
    double scale[3];
    float color_a, color_r, color_g, color_b;
    double position[3], orientation[4];                   // pose (x,y,z,w); identity for the reference's arrows
};
typedef std::vector<Marker> MarkerArray;
enum { MARKER_ARROW = 0, MARKER_CYLINDER = 3, MARKER_ADD = 0 };   // visualization_msgs::Marker::ARROW / CYLINDER / ADD

class Error : public std::runtime_error {
public:
    Error(gm_status s, const std::string &what) : std::runtime_error(what), status(s) {}
    gm_status status;
};

// Owns the gm_ctx; the reference's globals `params` + per-call PCL objects collapse into this.
class Processor {
public:
    // names/defaults: include/geometric_mapping/paramHandler.hpp:26-29, launch/mapping.launch:7-10
    Processor(double boxFilterBound = 5.0, double voxelGridLeafSize = 0.5, double neighborRadius = 0.5,
              double weightingFactor = 0.2, int device = 0, unsigned flags = GM_CFG_DEFAULT);
    // Several devices: frames are streamed round-robin over them (gm_group_submit_frame / gm_group_wait_frame,
    // `slots_per_device` frames in flight on each) -- the reference's one-frame-at-a-time consumer
    // (src/geometric_mapping.cpp:146,169) becomes a pipeline as deep as capacity().  processFrame() keeps working
    // (submit + wait), the stage calls use the first device.
    Processor(double boxFilterBound, double voxelGridLeafSize, double neighborRadius, double weightingFactor,
              const std::vector<int> &devices, unsigned flags, unsigned slots_per_device = 2);
    ~Processor();

    // ---- streaming (any constructor; with one device and one slot it degenerates to processFrame) ----
    unsigned capacity() const;        // frames that can be in flight
    unsigned inFlight() const;
    void submitFrame(const void *rows, unsigned n_points, unsigned point_step, unsigned off_x, unsigned off_y, unsigned off_z,
                     bool bigendian = false);   // the rows may be released on return
    gm_frame_result waitFrame();      // the oldest frame in flight; the accessors below then refer to it
    // the same without blocking: false (and `result` untouched) while the oldest frame in flight is still running or when
    // nothing is in flight (gm_poll_frame / gm_group_poll_frame).  The reference publishes each frame inside its own
    // callback (src/geometric_mapping.cpp:100-117); a host with frames in flight calls this at the top of every
    // callback and from a timer, and publishes whatever has finished.
    bool tryWaitFrame(gm_frame_result &result);
    // /choppedCloud straight into page-locked host rows (gm_set_cloud_output): one buffer per slot of every device, each
    // with room for max_points rows (grown when a larger frame is submitted).  The copy overlaps the tail of the frame;
    // choppedCloud() then reads those rows instead of fetching the cloud from the device.
    void enableCloudOutput(unsigned max_points);

    // ---- tunnel_processing.hpp:38-54 ----
    PointCloud chopCloud(const double &bound, const PointCloud &cloud);
    // compacts `cloud` in place (NaN-normal rows removed), like the reference (:81-85)
    NormalCloud getNormals(const double &neighborRadius, PointCloud &cloud);
    void getLocalFrame(const int &cloudSize, const double &weightingFactor, const NormalCloud &cloud_normals,
                       Vector3f &eigenVals, Matrix3f &eigenVecs);
    // the pcl::VoxelGrid half of rvizNormals (:214-220)
    PointCloud voxelGrid(const double &leafSize, const PointCloud &cloud);
    // kdtree->nearestKSearch(q, 1) (:237-239)
    std::vector<int> nearest(const PointCloud &cloud, const PointCloud &queries);

    // ---- the whole callback (src/geometric_mapping.cpp:55-92) in one device pass ----
    gm_frame_result processFrame(const void *rows, unsigned n_points, unsigned point_step, unsigned off_x, unsigned off_y,
                                 unsigned off_z, bool bigendian = false);
    PointCloud choppedCloud();        // /choppedCloud of the last frame
    NormalCloud normals();
    PointCloud voxelCentroids();
    NormalCloud voxelNormals();       // normals->at(kIndices[0]) per voxel centroid (needs GM_CFG_NEAREST)

    // ---- tunnel_processing.hpp:66-85 (pure host formatting, no device work) ----
    static Marker rvizArrow(const Vector3f &start, const Vector3f &end, const Vector3f &scale, const Vector4f &color,
                            const std::string &ns, const int &id = 0, const std::string &frame = "/velodyne");
    MarkerArray rvizNormals(const double &leafSize, const PointCloud &cloud, const NormalCloud &normals);
    // the same markers from what the last processFrame already left on the device (context created with
    // GM_CFG_NEAREST): voxel centroids + the normal of each centroid's nearest point -- a few KB of D2H, no second
    // voxel grid / 1-NN pass, no upload of the cloud
    MarkerArray rvizNormalsFromFrame();
    static MarkerArray rvizEigens(const Vector3f &eigenVals, const Matrix3f &eigenVecs);
    // The reference's unfinished cylinder output (getCylinder stub src/tunnel_processing.cpp:149-154, publisher
    // "centerAxisOutput" of type Marker src/geometric_mapping.cpp:41,119-121,163-165, param displayCylinder
    // launch/mapping.launch:15): one CYLINDER marker from gm_frame_result.cylinder (needs GM_CFG_RANSAC_CYLINDER),
    // centred on the axis point nearest the sensor origin, z axis along the fitted axis, `length` metres long.
    // Returns false (marker untouched) when the frame produced no cylinder.
    static bool rvizCylinder(const gm_frame_result &result, const double &length, Marker &marker,
                             const std::string &frame = "/velodyne");

    gm_ctx *ctx() { return ctx_; }

private:
    void check(gm_status s, const char *what);
    gm_ctx *ctx_;          // the context the stage calls run on (the group's first rank when there is a group)
    gm_group *grp_;        // several devices: owns the contexts
    gm_ctx *cur_;          // where the last completed frame's bulky outputs live ...
    unsigned cur_slot_;    // ... and in which slot
    unsigned n_slots_, next_slot_, pending_;   // single device: ring of slots
    gm_frame_result last_;                     // of the frame the accessors refer to
    struct CloudBuf { gm_ctx *ctx; unsigned slot; float *rows; unsigned cap; };
    std::vector<CloudBuf> cloud_bufs_;         // enableCloudOutput: one per (device, slot)
    void growCloudOutput(unsigned n_points);
    Processor(const Processor &);
    Processor &operator=(const Processor &);
};

}  // namespace gm_host
#endif
