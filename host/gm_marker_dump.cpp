// gm_marker_dump.cpp -- prints the host mirror's markers as JSON, every field the reference fills
// (/root/reference src/tunnel_processing.cpp:161-205), for comparison with tests/golden/markers_*.json.
//   gm_marker_dump eigens <12 hex words: 3 eigenvalues, 9 eigenvector entries column-major, fp32 bit patterns>
//       rvizEigens only: static host formatting, runs without a GPU
//   gm_marker_dump frame <file of n*3 float32> <n> <bound> <leaf> <radius> <wf>
//       the whole callback on the GPU (one gm_process_frame with GM_CFG_NEAREST): eigen-basis arrows and the
//       /surfaceNormals arrows built from what the frame left on the device
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gm_tunnel_processing.hpp"

using namespace gm_host;

static void print_marker(const Marker &m, bool last)
{
    std::printf("  {\"frame_id\": \"%s\", \"seq\": 0, \"ns\": \"%s\", \"id\": %d, \"type\": %d, \"action\": %d, "
                "\"points\": [[%.17g, %.17g, %.17g], [%.17g, %.17g, %.17g]], \"scale\": [%.17g, %.17g, %.17g], "
                "\"color\": {\"a\": %.9g, \"r\": %.9g, \"g\": %.9g, \"b\": %.9g}}%s\n",
                m.frame_id.c_str(), m.ns.c_str(), m.id, m.type, m.action, m.points[0][0], m.points[0][1], m.points[0][2],
                m.points[1][0], m.points[1][1], m.points[1][2], m.scale[0], m.scale[1], m.scale[2], (double)m.color_a,
                (double)m.color_r, (double)m.color_g, (double)m.color_b, last ? "" : ",");
}

static void print_array(const char *name, const MarkerArray &a, bool last)
{
    std::printf(" \"%s\": [\n", name);
    for (size_t i = 0; i < a.size(); ++i) print_marker(a[i], i + 1 == a.size());
    std::printf(" ]%s\n", last ? "" : ",");
}

int main(int argc, char **argv)
{
    if (argc >= 14 && !std::strcmp(argv[1], "eigens")) {
        float f[12];
        for (int k = 0; k < 12; ++k) { const unsigned u = (unsigned)std::strtoul(argv[2 + k], 0, 16); std::memcpy(&f[k], &u, 4); }
        Vector3f vals = {{f[0], f[1], f[2]}};
        Matrix3f vecs;
        std::memcpy(vecs.m, f + 3, sizeof(vecs.m));
        std::printf("{\n");
        print_array("eigenBasis", Processor::rvizEigens(vals, vecs), true);
        std::printf("}\n");
        return 0;
    }
    if (argc >= 8 && !std::strcmp(argv[1], "frame")) {
        const unsigned n = (unsigned)std::atoi(argv[3]);
        std::vector<float> xyz(3 * (size_t)(n ? n : 1));
        FILE *fp = std::fopen(argv[2], "rb");
        if (!fp || std::fread(&xyz[0], 12, n, fp) != n) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
        std::fclose(fp);
        try {
            Processor proc(std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7]), 0,
                           GM_CFG_DEFAULT | GM_CFG_NEAREST);
            const gm_frame_result r = proc.processFrame(&xyz[0], n, 12, 0, 4, 8);
            Vector3f vals = {{r.eigenvalues[0], r.eigenvalues[1], r.eigenvalues[2]}};
            Matrix3f vecs;
            std::memcpy(vecs.m, r.eigenvectors, sizeof(vecs.m));
            std::printf("{\n \"n_valid\": %u, \"n_voxels\": %u,\n", r.n_valid, r.n_voxels);
            print_array("eigenBasis", Processor::rvizEigens(vals, vecs), false);
            print_array("normals", proc.rvizNormalsFromFrame(), true);
            std::printf("}\n");
        } catch (const Error &e) {
            std::fprintf(stderr, "gm error %d: %s\n", (int)e.status, e.what());
            return 3;
        }
        return 0;
    }
    std::fprintf(stderr, "usage: gm_marker_dump eigens <12 hex words> | frame <xyz.f32> <n> <bound> <leaf> <radius> <wf>\n");
    return 1;
}
