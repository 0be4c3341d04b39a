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

__all__ = ["tunnel_frame", "cylinder_frame", "plane_patch", "fixed_k_radius", "to_pointcloud2", "velodyne_tunnel", "drop_row_padding"]


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


def velodyne_tunnel(rings=16, az=1800, seed=0, radius=2.0, axis_offset=(0.3, 0.5), floor_z=-1.2, sigma=0.01,
                    max_range=30.0, nan_frac=0.02, point_step=32, row_pad=0):
    """A spinning multi-beam lidar inside the tunnel: what /velodyne_points carries
    (/root/reference launch/mapping.launch:22; the reference's only data hint is a Velodyne rosbag, :29-32).

    The sensor sits at the origin; the tunnel is a cylinder of `radius` around an x-parallel axis through
    (0, axis_offset), cut by a floor plane z = floor_z.  `rings` beams at evenly spaced elevations (+-15 deg for 16 rings,
    -25..+15 deg otherwise, like the VLP-16 / HDL-32 / HDL-64) sweep `az` azimuth steps; every ray returns its nearest hit
    with N(0, sigma) range noise, or NaN when it hits nothing within max_range (rays along the tunnel) or drops out
    (nan_frac).  The cloud is ORGANISED: height = rings, width = az, one row of the message per ring, each row
    row_pad bytes longer than width * point_step -- row_step padding, which pcl::fromROSMsg honours
    (src/geometric_mapping.cpp:55).  Density falls off with range: a few centimetres between returns on the wall beside
    the sensor, decimetres at the far ends of the crop box -- the frame the uniform-density generator above is not.

    Returns dict(data=uint8 message bytes, height, width, point_step, row_step, offsets=(x, y, z),
                 xyz=float32 [height * width, 3] in message order, NaN rows included).
    point_step 32: x, y, z at 0 / 4 / 8, intensity float at 16, ring uint16 at 20 (velodyne_pointcloud's XYZIR);
    point_step 22: x, y, z at 0 / 4 / 8, intensity at 12, ring at 16, time float at 18 (XYZIRT, unaligned rows)."""
    if point_step not in (22, 32):
        raise ValueError("point_step must be 22 or 32")
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(-15.0, 15.0, rings) if rings <= 16 else np.linspace(-25.0, 15.0, rings))
    th = np.linspace(0.0, 2.0 * np.pi, az, endpoint=False) + rng.uniform(0, 2 * np.pi / az)
    E, T = np.meshgrid(el, th, indexing="ij")                  # [rings, az]
    d = np.stack([np.cos(E) * np.cos(T), np.cos(E) * np.sin(T), np.sin(E)], axis=-1).reshape(-1, 3)
    oy, oz = float(axis_offset[0]), float(axis_offset[1])
    # cylinder around the x-parallel axis through (0, oy, oz): |(t d)_yz - (oy, oz)|^2 = R^2, sensor inside
    a = d[:, 1] ** 2 + d[:, 2] ** 2
    b = -2.0 * (d[:, 1] * oy + d[:, 2] * oz)
    c = oy * oy + oz * oz - radius * radius
    disc = b * b - 4.0 * a * c
    with np.errstate(divide="ignore", invalid="ignore"):
        t_cyl = np.where(a > 1e-12, (-b + np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a), np.inf)
        t_floor = np.where(d[:, 2] < -1e-9, floor_z / d[:, 2], np.inf) if floor_z is not None else np.full(len(d), np.inf)
    t = np.minimum(t_cyl, t_floor)
    t = t + rng.normal(0.0, sigma, len(t))
    hit = np.isfinite(t) & (t > 0.3) & (t < max_range) & (rng.random(len(t)) >= nan_frac)
    xyz = np.where(hit[:, None], d * t[:, None], np.nan).astype(np.float32)
    row_step = az * point_step + int(row_pad)
    buf = np.full((rings, row_step), 0xC3, dtype=np.uint8)
    pts = buf[:, : az * point_step].reshape(rings, az, point_step)
    raw = xyz.view(np.uint8).reshape(rings, az, 12)
    pts[:, :, 0:12] = raw
    inten = rng.uniform(0, 255, rings * az).astype(np.float32).view(np.uint8).reshape(rings, az, 4)
    ring = np.repeat(np.arange(rings, dtype=np.uint16), az).view(np.uint8).reshape(rings, az, 2)
    if point_step == 32:
        pts[:, :, 16:20] = inten; pts[:, :, 20:22] = ring
    else:
        pts[:, :, 12:16] = inten; pts[:, :, 16:18] = ring
    return dict(data=np.ascontiguousarray(buf).reshape(-1), height=rings, width=az, point_step=point_step, row_step=row_step,
                offsets=(0, 4, 8), xyz=xyz)


def drop_row_padding(msg):
    """What the node does with an organised cloud whose rows are padded (ros/geometric_mapping_node.cpp): the C ABI takes
    point_step-strided rows, so the row_step padding is dropped once on the host.  Returns the packed uint8 rows."""
    h, w, ps, rs = msg["height"], msg["width"], msg["point_step"], msg["row_step"]
    return np.ascontiguousarray(msg["data"].reshape(h, rs)[:, : w * ps]).reshape(-1)
