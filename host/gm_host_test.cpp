// gm_host_test.cpp -- ROS-free C++ harness: drives the host mirror exactly as cloud_cb does
// (/root/reference src/geometric_mapping.cpp:48-125) on a synthetic PointCloud2-shaped
// buffer and checks invariants + analytic truth.  Built with g++ (no HIP headers), linked
// against libgm_hip.so: proves the C ABI is usable from the reference's own language.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gm_tunnel_processing.hpp"

using namespace gm_host;

static unsigned long long lcg = 88172645463325252ull;
static double urand() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-300, v = urand(); return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); }

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 50000;
    // Velodyne-like rows: 32 bytes, x,y,z at 0,4,8 then intensity/ring/time garbage
    const unsigned step = 32;
    std::vector<unsigned char> rows((size_t)n * step, 0xAB);
    for (int i = 0; i < n; ++i) {
        const double t = -6 + 12 * urand(), th = 6.283185307179586 * urand(), rr = 2.0 + 0.01 * nrand();
        const float p[3] = {(float)t, (float)(rr * std::cos(th)), (float)(rr * std::sin(th))};
        std::memcpy(&rows[(size_t)i * step], p, 12);
    }
    try {
        Processor proc;   // launch-file parameters
        // --- one device pass, as the node runs it
        gm_frame_result r = proc.processFrame(&rows[0], n, step, 0, 4, 8);
        PointCloud chopped = proc.choppedCloud();
        NormalCloud nrm = proc.normals();
        PointCloud vox = proc.voxelCentroids();
        std::printf("n_in=%u n_cropped=%u n_valid=%u n_voxels=%u axis=(%.6f %.6f %.6f) evals=(%.4g %.4g %.4g)\n", r.n_in,
                    r.n_cropped, r.n_valid, r.n_voxels, r.center_axis[0], r.center_axis[1], r.center_axis[2],
                    r.eigenvalues[0], r.eigenvalues[1], r.eigenvalues[2]);
        REQUIRE(r.n_in == (unsigned)n && r.n_cropped > 0.8 * n && r.n_cropped < 0.87 * n);
        REQUIRE(chopped.size() == r.n_valid && nrm.size() == r.n_valid && vox.size() == r.n_voxels);
        REQUIRE(std::fabs(std::fabs(r.center_axis[0]) - 1.0f) < 1e-3f);   // tunnel along x
        REQUIRE(r.eigenvalues[0] <= r.eigenvalues[1] && r.eigenvalues[1] <= r.eigenvalues[2]);
        for (size_t i = 0; i < chopped.size(); ++i)
            REQUIRE(std::fabs(chopped[i].x) <= 5 && std::fabs(chopped[i].y) <= 5 && std::fabs(chopped[i].z) <= 5);
        // --- the same through the four stage functions, in cloud_cb's order
        PointCloud cloud(n);
        for (int i = 0; i < n; ++i) { std::memcpy(&cloud[i].x, &rows[(size_t)i * step], 12); cloud[i].pad = 1.0f; }
        PointCloud cloudChopped = proc.chopCloud(5.0, cloud);                         // :57
        NormalCloud cloudNormals = proc.getNormals(0.5, cloudChopped);                 // :63
        MarkerArray normalsDisp = proc.rvizNormals(0.5, cloudChopped, cloudNormals);   // :70-75
        Vector3f eigenVals; Matrix3f eigenVecs;
        proc.getLocalFrame((int)cloudChopped.size(), 0.2, cloudNormals, eigenVals, eigenVecs);  // :82-88
        Vector3f centerAxis = eigenVecs.col(0);                                        // :91-92
        MarkerArray eigenBasis = Processor::rvizEigens(eigenVals, eigenVecs);          // :96
        REQUIRE(cloudChopped.size() == r.n_valid && cloudNormals.size() == r.n_valid);
        REQUIRE(normalsDisp.size() == r.n_voxels && eigenBasis.size() == 3);
        // stage path vs frame path: same neighbour sets, but the stage call bins the cloud on a grid anchored
        // at the data's min corner (the frame path anchors at the crop box), so fp32 sums run in another order
        for (int k = 1; k < 3; ++k) REQUIRE(std::fabs(eigenVals(k) - r.eigenvalues[k]) <= 1e-6f * r.eigenvalues[2]);
        REQUIRE(std::fabs(eigenVals(0) - r.eigenvalues[0]) <= 1e-5f * r.eigenvalues[2]);
        {
            float dot = 0;
            for (int k = 0; k < 3; ++k) dot += eigenVecs.m[k] * r.eigenvectors[k];
            REQUIRE(std::fabs(std::fabs(dot) - 1.0f) < 1e-6f);
        }
        REQUIRE(std::memcmp(&cloudChopped[0], &chopped[0], 12) == 0);
        REQUIRE(normalsDisp[0].ns == "normals" && normalsDisp[0].frame_id == "/velodyne" && normalsDisp[0].color_b == 1.0f);
        REQUIRE(eigenBasis[2].ns == "eigenBasis" && eigenBasis[2].id == 2 && eigenBasis[0].color_r == 1.0f);
        REQUIRE(std::fabs(eigenBasis[0].points[1][0] - centerAxis(0)) < 1e-7);
        // --- the cylinder marker (reference's displayCylinder stub): RANSAC cylinder on, tunnel R = 2 along x
        {
            Processor pc(5.0, 0.5, 0.5, 0.2, 0, GM_CFG_DEFAULT | GM_CFG_RANSAC_CYLINDER);
            gm_frame_result rc = pc.processFrame(&rows[0], n, step, 0, 4, 8);
            Marker cyl;
            REQUIRE(Processor::rvizCylinder(rc, 10.0, cyl));
            REQUIRE(cyl.type == MARKER_CYLINDER && cyl.ns == "cylinder" && cyl.frame_id == "/velodyne");
            REQUIRE(std::fabs(cyl.scale[0] - 4.0) < 0.1 && cyl.scale[1] == cyl.scale[0] && cyl.scale[2] == 10.0);
            // rotate (0,0,1) by the marker quaternion: must give the fitted axis, which is +-x here
            const double *q = cyl.orientation;
            const double zx = 2 * (q[0] * q[2] + q[3] * q[1]), zy = 2 * (q[1] * q[2] - q[3] * q[0]),
                         zz = 1 - 2 * (q[0] * q[0] + q[1] * q[1]);
            REQUIRE(std::fabs(zx - rc.cylinder[3]) < 1e-6 && std::fabs(zy - rc.cylinder[4]) < 1e-6 && std::fabs(zz - rc.cylinder[5]) < 1e-6);
            REQUIRE(std::fabs(std::fabs(zx) - 1.0) < 0.02);
            REQUIRE(std::sqrt(cyl.position[1] * cyl.position[1] + cyl.position[2] * cyl.position[2]) < 0.1);  // axis through the origin
            gm_frame_result none = r;   // a frame processed without the RANSAC flag carries no cylinder
            Marker untouched;
            REQUIRE(!Processor::rvizCylinder(none, 10.0, untouched));
        }
        // --- streaming through the device-list constructor (gm_group_submit_frame / gm_group_wait_frame): frames come
        // back in submission order, each the frame processFrame computes, and the accessors follow the returned frame
        {
            std::vector<int> devs(1, 0);
            Processor ps(5.0, 0.5, 0.5, 0.2, devs, GM_CFG_DEFAULT, 2);
            REQUIRE(ps.capacity() == 2 && ps.inFlight() == 0);
            const unsigned sizes[4] = {(unsigned)n, (unsigned)(n / 2), (unsigned)(n / 3), (unsigned)n};
            unsigned done = 0;
            for (int k = 0; k < 4 || ps.inFlight(); ) {
                while (k < 4 && ps.inFlight() < ps.capacity()) { ps.submitFrame(&rows[0], sizes[k], step, 0, 4, 8); ++k; }
                gm_frame_result rs = ps.waitFrame();
                REQUIRE(rs.n_in == sizes[done]);
                REQUIRE(ps.choppedCloud().size() == rs.n_valid);
                if (sizes[done] == (unsigned)n) {
                    REQUIRE(rs.n_valid == r.n_valid && std::memcmp(rs.scatter, r.scatter, sizeof(r.scatter)) == 0);
                    REQUIRE(std::memcmp(&ps.choppedCloud()[0], &chopped[0], chopped.size() * sizeof(PointXYZ)) == 0);
                }
                ++done;
            }
            REQUIRE(done == 4);
        }
        // --- the node's streaming loop (ros/geometric_mapping_node.cpp): page-locked /choppedCloud rows, and every frame
        // taken off the pipeline by a completion query -- at the top of each "callback", then until the pipeline is
        // empty -- never by waiting for it to fill.  Frames come back in order, each the blocking frame bit for bit.
        {
            std::vector<int> devs(1, 0);
            Processor ps(5.0, 0.5, 0.5, 0.2, devs, GM_CFG_DEFAULT, 2);
            ps.enableCloudOutput((unsigned)n / 4);            // too small on purpose: grown by the first frame
            const unsigned total = 12;
            unsigned submitted = 0, done = 0, polls = 0;
            while (done < total) {
                gm_frame_result rs;
                while (ps.tryWaitFrame(rs)) {                 // publish what has finished
                    const unsigned want = done % 2 ? (unsigned)(n / 2) : (unsigned)n;
                    REQUIRE(rs.n_in == want);
                    const PointCloud pc = ps.choppedCloud();  // (from the page-locked rows)
                    REQUIRE(pc.size() == rs.n_valid);
                    if (want == (unsigned)n) {
                        REQUIRE(rs.n_valid == r.n_valid && std::memcmp(rs.scatter, r.scatter, sizeof(r.scatter)) == 0);
                        REQUIRE(std::memcmp(&pc[0], &chopped[0], chopped.size() * sizeof(PointXYZ)) == 0);
                    }
                    ++done;
                }
                if (submitted < total && ps.inFlight() < ps.capacity()) {
                    ps.submitFrame(&rows[0], submitted % 2 ? (unsigned)(n / 2) : (unsigned)n, step, 0, 4, 8);
                    ++submitted;
                }
                REQUIRE(++polls < 100000000u);                // (a frame that never finishes)
            }
            REQUIRE(ps.inFlight() == 0 && !ps.tryWaitFrame(r) && r.n_in == (unsigned)n);   // nothing in flight: false, result untouched
        }
        // error conventions: a status, never a crash
        bool threw = false;
        try { proc.getLocalFrame((int)cloudNormals.size() + 1, 0.2, cloudNormals, eigenVals, eigenVecs); } catch (const std::out_of_range &) { threw = true; }
        REQUIRE(threw);
        PointCloud empty;
        REQUIRE(proc.chopCloud(5.0, empty).empty());
    } catch (const Error &e) {
        std::fprintf(stderr, "gm error %d: %s\n", (int)e.status, e.what());
        return 2;
    }
    std::printf("gm_host_test ok\n");
    return 0;
}
