"""CPU restatement of the reference's marker formatting (test infrastructure; parity unpinned).

Field-for-field restatement of
  rvizArrow    /root/reference src/tunnel_processing.cpp:161-205
  rvizNormals  /root/reference src/tunnel_processing.cpp:208-256 (marker loop :237-252)
  rvizEigens   /root/reference src/tunnel_processing.cpp:260-300
in numpy float32 arithmetic (Eigen::Vector3f / Vector4f are float; Marker point / scale
fields are float64, colours float32).  header.stamp = ros::Time::now() (:175) is not
restated: it is wall-clock.  Only tests/ may import this module.
"""
import numpy as np

ARROW, ADD = 0, 0          # visualization_msgs::Marker::ARROW / ADD (:179-180)
F = np.float32


def rviz_arrow(start, end, scale, color, ns, id=0, frame="/velodyne"):
    """:161-205.  start/end/scale: Eigen::Vector3f, color: Eigen::Vector4f read as A,R,G,B (:199-202).
    Defaults id = 0, frame = "/velodyne": include/geometric_mapping/tunnel_processing.hpp:72-73."""
    s, e, sc, c = (np.asarray(a, dtype=F) for a in (start, end, scale, color))
    return {
        "frame_id": frame, "seq": 0,                                   # :174, :176
        "ns": ns, "id": int(id), "type": ARROW, "action": ADD,         # :177-180
        "points": [[float(s[0]), float(s[1]), float(s[2])],            # :185-187 (float -> float64 fields)
                   [float(e[0]), float(e[1]), float(e[2])]],           # :189-191
        "scale": [float(sc[0]), float(sc[1]), float(sc[2])],           # :194-196
        "color": {"a": float(c[0]), "r": float(c[1]), "g": float(c[2]), "b": float(c[3])},   # :199-202
        # pose is left default-constructed by the reference (all zeros; rviz treats the zero quaternion as identity)
    }


def rviz_normals(voxel_centroids, nearest_idx, normals):
    """:225-256 after the VoxelGrid filter (:214-220): one arrow per voxel centroid, in VoxelGrid output order, from the
    centroid to THE NORMAL VECTOR ITSELF (not centroid + normal, :247-249) of the point nearestKSearch(…, 1) returns."""
    scale = np.array([0.025, 0.075, 0.0625], dtype=F)                  # :230 (double literals -> float)
    color = np.array([1, 0, 0, 1], dtype=F)                            # :231: a=1 r=0 g=0 b=1
    out = []
    for i in range(len(voxel_centroids)):                              # :236
        start = np.asarray(voxel_centroids[i][:3], dtype=F)            # :242-244
        end = np.asarray(normals[int(nearest_idx[i])][:3], dtype=F)    # :247-249
        out.append(rviz_arrow(start, end, scale, color, "normals", i))  # :251
    return out


def rviz_eigens(eigen_vals, eigen_vecs):
    """:260-300.  eigen_vals: Vector3f ascending; eigen_vecs: Matrix3f, eigenvectors in COLUMNS ([row, col])."""
    vals = np.asarray(eigen_vals, dtype=F)
    vecs = np.asarray(eigen_vecs, dtype=F)
    nrm = F(np.sqrt(F(F(F(vals[0] * vals[0]) + F(vals[1] * vals[1])) + F(vals[2] * vals[2]))))   # eigenVals.norm()
    norms = (F(1) / nrm) * np.abs(vals)                                # :265: (1 / norm) * cwiseAbs, float
    out = []
    for i in range(3):                                                 # :272
        e = float(norms[i])                                            # float promoted to double in the expressions below
        scale = np.array([0.1 - (0.05 * e), 0.3 - (0.15 * e), 0.25 - (0.125 * e)], dtype=np.float64).astype(F)   # :274-278
        color = np.array([1.0, 1.0 if i == 0 else 0.0, 1.0 if i == 1 else 0.0, 1.0 if i == 2 else 0.0], dtype=F)  # :280-287
        out.append(rviz_arrow(np.zeros(3, F), vecs[:, i], scale, color, "eigenBasis", i))   # :289-296
    return out
