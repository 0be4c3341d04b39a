"""Synthetic tunnel frames (build-defined; the reference ships no data).

The reference's only data hint is a commented-out Velodyne rosbag path
(/root/reference launch/mapping.launch:29-32), so every test and benchmark
frame is produced here: a noisy cylinder of radius R around an axis through
the sensor origin, optionally with a floor plane and uniform outliers.
Seeded with numpy's PCG64 so the same (n, seed) gives the same float32 cloud
on every machine.
"""
from __future__ import annotations

import numpy as np

__all__ = ["tunnel_frame", "cylinder_frame", "plane_patch", "fixed_k_radius", "to_pointcloud2"]


def _basis(axis):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = np.array([0.0, 0.0, 1.0]) if abs(a[2]) < 0.9 else np.array([0.0, 1.0, 0.0])
    u = np.cross(h, a)
    u /= np.linalg.norm(u)
    v = np.cross(a, u)
    return a, u, v


def cylinder_frame(n, seed=0, radius=2.0, length=12.0, sigma=0.01, axis=(1.0, 0.0, 0.0)):
    """n points on a noisy cylinder; float32 (n,3), generation order (unsorted)."""
    rng = np.random.default_rng(seed)
    a, u, v = _basis(axis)
    t = rng.uniform(-length / 2, length / 2, n)
    th = rng.uniform(0.0, 2 * np.pi, n)
    rr = radius + rng.normal(0.0, sigma, n)
    p = t[:, None] * a + (rr * np.cos(th))[:, None] * u + (rr * np.sin(th))[:, None] * v
    return np.ascontiguousarray(p, dtype=np.float32)


def tunnel_frame(n, seed=0, radius=2.0, length=12.0, sigma=0.01, axis=(1.0, 0.0, 0.0),
                 floor_z=None, outlier_frac=0.0, outlier_cube=6.0):
    """Cylinder, optionally with a floor (points below floor_z are projected
    onto the plane z=floor_z with N(0,sigma) noise) and a fraction of uniform
    outliers in a +-outlier_cube box (config 3 uses
    floor_z=-1.2, outlier_frac=0.01)."""
    rng = np.random.default_rng(seed)
    a, u, v = _basis(axis)
    t = rng.uniform(-length / 2, length / 2, n)
    th = rng.uniform(0.0, 2 * np.pi, n)
    rr = radius + rng.normal(0.0, sigma, n)
    p = t[:, None] * a + (rr * np.cos(th))[:, None] * u + (rr * np.sin(th))[:, None] * v
    if floor_z is not None:
        below = p[:, 2] < floor_z
        p[below, 2] = floor_z + rng.normal(0.0, sigma, int(below.sum()))
    if outlier_frac > 0.0:
        m = int(round(n * outlier_frac))
        idx = rng.choice(n, size=m, replace=False)
        p[idx] = rng.uniform(-outlier_cube, outlier_cube, (m, 3))
    return np.ascontiguousarray(p, dtype=np.float32)


def plane_patch(n, seed=0, normal=(0.0, 0.0, 1.0), offset=1.5, half=3.0, sigma=0.0):
    """n points on the plane normal.p = offset (|in-plane coords| <= half)."""
    rng = np.random.default_rng(seed)
    a, u, v = _basis(normal)
    s = rng.uniform(-half, half, n)
    t = rng.uniform(-half, half, n)
    e = rng.normal(0.0, sigma, n) if sigma > 0 else np.zeros(n)
    p = (offset + e)[:, None] * a + s[:, None] * u + t[:, None] * v
    return np.ascontiguousarray(p, dtype=np.float32)


def fixed_k_radius(n, r50k=0.5):
    """Neighbour radius that keeps k~256 as the frame grows (SURVEY.md §8d:
    r = 0.5*sqrt(50000/N); surface density scales with N, disc area with r^2)."""
    return float(r50k * np.sqrt(50000.0 / float(n)))


def to_pointcloud2(xyz, point_step=16, offsets=(0, 4, 8), fill=0):
    """Pack (n,3) float32 into sensor_msgs/PointCloud2 `data` rows: the byte
    layout pcl::fromROSMsg reads at /root/reference src/geometric_mapping.cpp:55."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    buf = np.full((n, point_step), fill, dtype=np.uint8)
    raw = xyz.view(np.uint8).reshape(n, 12)
    for k, off in enumerate(offsets):
        buf[:, off:off + 4] = raw[:, 4 * k:4 * k + 4]
    return buf.reshape(-1)
