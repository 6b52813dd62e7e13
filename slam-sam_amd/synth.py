"""Seeded, dependency-free synthetic clouds for the NDT configurations of BASELINE.json.

Nothing here reads the reference tree; the geometry follows SURVEY.md section 8(d):
  * two_planes(): the reference test's two-plane fixture
    (ref: extern/svn_ndt/test/test_svn_ndt.cpp:44-83) with a portable PRNG;
  * OusterSim: ray-cast OS-2-128-style scans (128 or 256 beams x 1024 columns,
    +-11.25 deg vertical FOV, range clipped to [0.5, 250] m as
    config/lidar_config_berlin.json does) inside a seeded block-world street scene;
  * config_c1 / config_c2 / config_c3: the workloads C1, C2 and C3/C4.
All functions return float32 N x 3 arrays and float64 4x4 poses.
"""
import numpy as np


# --------------------------------------------------------------------------- poses
def rot_xyz(roll, pitch, yaw):
    """R = Rx(roll) Ry(pitch) Rz(yaw) (the NDT pose convention)."""
    cx, sx, cy, sy, cz, sz = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def pose_matrix(x, y, z, roll, pitch, yaw):
    T = np.eye(4)
    T[:3, :3] = rot_xyz(roll, pitch, yaw)
    T[:3, 3] = [x, y, z]
    return T


def so3_log(R):
    """Rotation vector of R; atan2 form so tiny angles do not underflow to zero."""
    a = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(a)
    c = (np.trace(R) - 1.0) / 2.0
    th = np.arctan2(s, c)
    if s < 1e-300:
        return np.zeros(3)
    return a * (th / s)


def pose_error(T_est, T_ref):
    """(translation error in m, rotation error in rad) of inv(T_est) @ T_ref."""
    E = np.linalg.inv(np.asarray(T_est, dtype=np.float64)) @ np.asarray(T_ref, dtype=np.float64)
    return float(np.linalg.norm(E[:3, 3])), float(np.linalg.norm(so3_log(E[:3, :3])))


def se3_log_error(T_est, T_ref):
    """Norms of the rotation / translation parts of Log(inv(T_est) T_ref), the metric of
    the reference test (ref: test_svn_ndt.cpp:188-191)."""
    E = np.linalg.inv(np.asarray(T_est, dtype=np.float64)) @ np.asarray(T_ref, dtype=np.float64)
    w = so3_log(E[:3, :3])
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        Vinv = np.eye(3) - 0.5 * K
    else:
        Vinv = np.eye(3) - 0.5 * K + (1.0 / th ** 2) * (1.0 - th * np.sin(th) / (2.0 * (1.0 - np.cos(th)))) * (K @ K)
    u = Vinv @ E[:3, 3]
    return float(np.linalg.norm(u)), float(th)


def transform(T, pts):
    T = np.asarray(T, dtype=np.float64)
    return (pts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


# --------------------------------------------------------------------------- two planes
def two_planes(seed=1337, noise=0.02, step=0.15, half=10.0, max_points=None):
    """Source / target / ground truth / initial guess of the reference test geometry."""
    rng = np.random.default_rng(seed)
    ax = np.arange(-half, half + 1e-9, step)
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    p1 = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1)
    p2 = np.stack([gx.ravel(), np.zeros(gx.size), gy.ravel()], 1)
    src = np.concatenate([p1, p2]).astype(np.float32)
    if max_points is not None and len(src) > max_points:
        sel = np.sort(rng.choice(len(src), size=max_points, replace=False))
        src = src[sel]
    # ground truth Rz(0.2618) Ry(0.0873), t = (0.5, 0, 0.3)  (ref :104-106)
    cz, sz, cy, sy = np.cos(0.2618), np.sin(0.2618), np.cos(0.0873), np.sin(0.0873)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    gt = np.eye(4)
    gt[:3, :3] = Rz @ Ry
    gt[:3, 3] = [0.5, 0.0, 0.3]
    tgt = (src.astype(np.float64) @ gt[:3, :3].T + gt[:3, 3] + rng.normal(0.0, noise, src.shape)).astype(np.float32)
    # initial guess: gt * Exp(-delta), delta = [rot 0.05,-0.02,0.04 | trans 0.02,-0.01,0.03] (ref :110-111)
    w = -np.array([0.05, -0.02, 0.04])
    v = -np.array([0.02, -0.01, 0.03])
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
    V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    d = np.eye(4)
    d[:3, :3] = dR
    d[:3, 3] = V @ v
    guess = gt @ d
    return src, tgt, gt, guess


# --------------------------------------------------------------------------- lidar sim
class Scene:
    """Ground plane z = 0, axis-aligned boxes, and an enclosing wall box."""

    def __init__(self, seed=42, extent=(120.0, 80.0), n_boxes=60, wall_height=30.0):
        rng = np.random.default_rng(seed)
        ex, ey = extent
        boxes = []
        # two rows of buildings along a street (the x axis), plus scattered blocks
        x = -ex
        while x < ex:
            w = rng.uniform(8.0, 25.0)
            for side in (-1, 1):
                d = rng.uniform(8.0, 20.0)
                hgt = rng.uniform(6.0, 25.0)
                y0 = side * rng.uniform(7.0, 10.0)
                y1 = y0 + side * d
                boxes.append([x, min(y0, y1), 0.0, x + w - rng.uniform(0.5, 3.0), max(y0, y1), hgt])
            x += w
        for _ in range(n_boxes):
            cx, cy = rng.uniform(-ex, ex), rng.uniform(-ey, ey)
            if abs(cy) < 6.0:
                continue  # keep the street clear
            sx, sy, hgt = rng.uniform(1.0, 6.0), rng.uniform(1.0, 6.0), rng.uniform(1.0, 8.0)
            boxes.append([cx - sx, cy - sy, 0.0, cx + sx, cy + sy, hgt])
        self.boxes = np.array(boxes, dtype=np.float64)
        self.wall = np.array([-ex - 5, -ey - 5, -1.0, ex + 5, ey + 5, wall_height], dtype=np.float64)


class OusterSim:
    def __init__(self, scene, beams=128, cols=1024, fov_deg=22.5, range_min=0.5, range_max=250.0,
                 range_noise=0.02):
        self.scene = scene
        self.beams, self.cols = beams, cols
        alt = np.deg2rad(np.linspace(fov_deg / 2, -fov_deg / 2, beams))
        az = np.linspace(0.0, 2 * np.pi, cols, endpoint=False)
        A, B = np.meshgrid(alt, az, indexing="ij")
        self.dirs = np.stack([np.cos(A) * np.cos(B), np.cos(A) * np.sin(B), np.sin(A)], -1).reshape(-1, 3)
        self.rmin, self.rmax, self.noise = range_min, range_max, range_noise

    def scan(self, T_world_sensor, seed=0):
        """Points in the SENSOR frame (float32), in firing order, misses dropped."""
        rng = np.random.default_rng(seed)
        T = np.asarray(T_world_sensor, dtype=np.float64)
        o = T[:3, 3].astype(np.float32)
        out = []
        chunk = 32768
        bx = self.scene.boxes.astype(np.float32)
        for s in range(0, len(self.dirs), chunk):
            db = self.dirs[s:s + chunk]
            d = (db @ T[:3, :3].T).astype(np.float32)
            t = np.full(len(d), np.inf, dtype=np.float32)
            # ground
            dz = d[:, 2]
            tg = np.where(dz < -1e-9, -o[2] / np.where(dz < -1e-9, dz, -1.0), np.inf)
            t = np.minimum(t, np.where(tg > 0, tg, np.inf))
            inv = (1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)).astype(np.float32)
            # boxes: slab test per axis (2-D temporaries), nearest entry
            tn = np.full((len(d), len(bx)), -np.inf, dtype=np.float32)
            tf = np.full((len(d), len(bx)), np.inf, dtype=np.float32)
            for a in range(3):
                t1 = (bx[None, :, a] - o[a]) * inv[:, a, None]
                t2 = (bx[None, :, 3 + a] - o[a]) * inv[:, a, None]
                np.maximum(tn, np.minimum(t1, t2), out=tn)
                np.minimum(tf, np.maximum(t1, t2), out=tf)
            hit = (tf >= np.maximum(tn, 0.0)) & (tn > 0.0)
            t = np.minimum(t, np.where(hit, tn, np.inf).min(-1))
            # enclosing walls, seen from inside: exit distance; the open top is a miss
            w = self.scene.wall.astype(np.float32)
            t1 = (w[None, 0:3] - o[None, :]) * inv
            t2 = (w[None, 3:6] - o[None, :]) * inv
            tfar_axis = np.maximum(t1, t2)
            tw = tfar_axis.min(-1)
            top_exit = (np.argmin(tfar_axis, -1) == 2) & (d[:, 2] > 0)
            t = np.minimum(t, np.where(top_exit, np.inf, tw))
            r = t.astype(np.float64) + rng.normal(0.0, self.noise, len(t))
            ok = np.isfinite(t) & (r >= self.rmin) & (r <= self.rmax)
            out.append((db[ok] * r[ok, None]).astype(np.float32))
        return np.concatenate(out)


def voxel_downsample(pts, leaf):
    """First point per voxel (order-preserving), like a coarse pcl::VoxelGrid."""
    key = np.floor(pts.astype(np.float64) / leaf).astype(np.int64)
    key -= key.min(0)
    dims = key.max(0) + 1
    flat = (key[:, 2] * dims[1] + key[:, 1]) * dims[0] + key[:, 0]
    _, first = np.unique(flat, return_index=True)
    return pts[np.sort(first)]


# --------------------------------------------------------------------------- configs
def config_c1(max_points=10000):
    """C1: two 10k-point clouds, 1.0 m voxel."""
    src, tgt, gt, guess = two_planes(seed=1337, max_points=max_points)
    return dict(name="C1 two-plane 10k/10k, 1.0 m", source=src, target=tgt, gt=gt, guess=guess,
                resolution=1.0)


def config_c2(beams=128, cols=1024, seed=42):
    """C2: 128x1024 scan-to-scan, 1.0 m voxel; offset (0.5 m, 0.1 m, 2 deg yaw)."""
    scene = Scene(seed=seed)
    sim = OusterSim(scene, beams=beams, cols=cols)
    Ta = pose_matrix(0.0, 0.0, 2.0, 0.0, 0.0, 0.0)
    Tb = Ta @ pose_matrix(0.5, 0.1, 0.0, 0.0, 0.0, np.deg2rad(2.0))
    tgt = sim.scan(Ta, seed=seed + 1)
    src = sim.scan(Tb, seed=seed + 2)
    gt = np.linalg.inv(Ta) @ Tb
    return dict(name="C2 %dx%d scan-to-scan, 1.0 m" % (beams, cols), source=src, target=tgt, gt=gt,
                guess=np.eye(4), resolution=1.0)


def _cached(name, builder):
    """Generated workloads are cached under $NDT_SYNTH_CACHE (default /tmp/ndt_synth)."""
    import os
    d = os.environ.get("NDT_SYNTH_CACHE", "/tmp/ndt_synth")
    path = os.path.join(d, name + ".npz")
    try:
        z = np.load(path, allow_pickle=False)
        return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    except Exception:
        pass
    cfg = builder()
    try:
        os.makedirs(d, exist_ok=True)
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, **cfg)
        os.replace(tmp, path)
    except Exception:
        pass
    return cfg


def config_c3(n_map=1000000, n_src=200000, seed=7, cols=1024):
    return _cached("c3_v2_%d_%d_%d_%d" % (n_map, n_src, seed, cols),
                   lambda: _config_c3(n_map, n_src, seed, cols))


def _config_c3(n_map=1000000, n_src=200000, seed=7, cols=1024):
    """C3/C4: 200k-point scan into a 1M-point submap, 0.5 m voxel.

    Map = 8 scans (256 beams) along a 40 m track, in the map frame, thinned with a
    0.03 m voxel filter and randomly subsampled to exactly n_map points.  Source = a
    fresh 256-beam scan subsampled (order kept) to n_src points; the initial guess is
    off by (0.3 m, 2 deg).
    """
    scene = Scene(seed=42, extent=(140.0, 90.0), n_boxes=90)
    sim = OusterSim(scene, beams=256, cols=cols)
    rng = np.random.default_rng(seed)
    clouds = []
    for k in range(8):
        Tk = pose_matrix(-20.0 + 40.0 * k / 7.0, 0.3 * np.sin(k), 2.0, 0.0, 0.0, 0.02 * k)
        clouds.append(transform(Tk, sim.scan(Tk, seed=100 + k)))
    m = voxel_downsample(np.concatenate(clouds), 0.03)
    if len(m) < n_map:
        raise RuntimeError("map has only %d points" % len(m))
    m = m[np.sort(rng.choice(len(m), size=n_map, replace=False))]
    Ts = pose_matrix(3.7, -0.4, 2.0, 0.004, -0.006, 0.11)
    s = sim.scan(Ts, seed=11)
    if len(s) < n_src:
        raise RuntimeError("scan has only %d points" % len(s))
    s = s[np.sort(np.random.default_rng(11).choice(len(s), size=n_src, replace=False))]
    guess = Ts @ pose_matrix(0.25, -0.15, 0.06, 0.0, 0.0, np.deg2rad(2.0))
    return dict(name="C3 %dk scan into %dk-pt map, 0.5 m" % (n_src // 1000, n_map // 1000), source=s,
                target=m, gt=Ts, guess=guess, resolution=0.5)


# --------------------------------------------------------------------------- C3-wide
def _terrain(x, y):
    """Gentle hills: a few metres of relief over tens of metres."""
    return 3.0 * np.sin(x / 37.0) * np.cos(y / 53.0) + 1.5 * np.sin((x + y) / 17.0) + 0.01 * x


def _wide_surfaces(rng, n_ground, n_wall, half, boxes):
    """Random points on the terrain and on the vertical faces of `boxes` (cx, cy, sx, sy, h)."""
    gx, gy = rng.uniform(-half, half, n_ground), rng.uniform(-half, half, n_ground)
    ground = np.stack([gx, gy, _terrain(gx, gy)], 1)
    b = boxes[rng.integers(0, len(boxes), n_wall)]
    face = rng.integers(0, 4, n_wall)
    u = rng.uniform(-1.0, 1.0, n_wall)
    wx = np.where(face < 2, b[:, 0] + np.where(face == 0, -1.0, 1.0) * b[:, 2], b[:, 0] + u * b[:, 2])
    wy = np.where(face < 2, b[:, 1] + u * b[:, 3], b[:, 1] + np.where(face == 2, -1.0, 1.0) * b[:, 3])
    wz = _terrain(b[:, 0], b[:, 1]) + rng.uniform(0.0, 1.0, n_wall) * b[:, 4]
    return np.concatenate([ground, np.stack([wx, wy, wz], 1)])


def config_c3_wide(n_map=4000000, n_src=200000, seed=21):
    return _cached("c3wide_v3_%d_%d_%d" % (n_map, n_src, seed), lambda: _config_c3_wide(n_map, n_src, seed))


def _config_c3_wide(n_map, n_src, seed):
    """C3-wide: the C3 shape (200k-point scan into a voxelised map, 0.5 m voxel) on a map whose
    voxel table does NOT fit a cache level that C3's does: open terrain of 260 m x 260 m with 60
    buildings, 4M map points -> ~2.9e5 valid leaves (23 MB of records, 70 MB dense index grid)
    against C3's 2.1e4 (1.7 MB).  The map is in random order (no scan coherence); the source is a
    fresh sample of the same surfaces within 100 m of the sensor, in the sensor frame."""
    rng = np.random.default_rng(seed)
    half = 130.0
    nb = 60
    boxes = np.stack([rng.uniform(-half + 10, half - 10, nb), rng.uniform(-half + 10, half - 10, nb),
                      rng.uniform(2.0, 9.0, nb), rng.uniform(2.0, 9.0, nb), rng.uniform(4.0, 12.0, nb)], 1)
    n_wall = n_map // 5
    m = _wide_surfaces(rng, n_map - n_wall, n_wall, half, boxes)
    m = (m + rng.normal(0.0, 0.02, m.shape)).astype(np.float32)
    m = m[rng.permutation(len(m))]
    z0 = float(_terrain(5.0, -3.0)) + 2.0
    Ts = pose_matrix(5.0, -3.0, z0, 0.01, -0.015, 0.35)
    rs = np.random.default_rng(seed + 1)
    cand = _wide_surfaces(rs, 6 * n_src, 2 * n_src, half, boxes)
    d = np.hypot(cand[:, 0] - 5.0, cand[:, 1] + 3.0)
    cand = cand[d < 100.0]
    if len(cand) < n_src:
        raise RuntimeError("wide source has only %d points" % len(cand))
    cand = cand[np.sort(rs.choice(len(cand), size=n_src, replace=False))]
    cand = cand + rs.normal(0.0, 0.02, cand.shape)
    s = transform(np.linalg.inv(Ts), cand).astype(np.float32)
    guess = Ts @ pose_matrix(0.22, -0.12, 0.05, 0.0, 0.0, np.deg2rad(1.5))
    return dict(name="C3-wide %dk scan into %dk-pt open-terrain map, 0.5 m" % (n_src // 1000, n_map // 1000),
                source=s, target=m, gt=Ts, guess=guess, resolution=0.5)
