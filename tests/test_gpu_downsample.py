"""pcl::VoxelGrid on the device (second half; ref: run/pipeline_ins_map_distribution.cpp:324-340 filters the
accumulated map at `mapvoxelsize` = 0.5 m before the NDT export).  Checked against a NumPy restatement of PCL's
published algorithm (f32 index arithmetic, all-field float centroid per occupied voxel, ascending voxel index): voxel
set, counts-as-implied and order exact; centroids exact too, because the restatement adds a voxel's points in the
same (input) order -- PCL's own order within a voxel is unspecified, agreement with it is to float rounding (1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def voxelgrid_numpy(pts, leaf, intensity=None):
    """PCL VoxelGrid::applyFilter restated: returns (centroids [m,3] f32, intensity [m] f32 or None, counts)."""
    p = np.asarray(pts, np.float32)
    fin = np.isfinite(p).all(axis=1)
    idx_in = np.nonzero(fin)[0]
    q = p[fin]
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = q.min(0), q.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    assert int(div_b[0]) * int(div_b[1]) * int(div_b[2]) <= 2**31 - 1
    ijk = (np.floor(q * inv) - min_b.astype(np.float32)).astype(np.int64)     # f32 subtraction, as PCL writes it
    flat = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    order = np.argsort(flat, kind="stable")
    fs = flat[order]
    heads = np.nonzero(np.r_[True, fs[1:] != fs[:-1]])[0]
    counts = np.diff(np.r_[heads, len(fs)])
    cols = [q[order, 0], q[order, 1], q[order, 2]]
    if intensity is not None:
        cols.append(np.asarray(intensity, np.float32)[idx_in][order])
    sums = [np.zeros(len(heads), np.float32) for _ in cols]
    for j in range(int(counts.max())):                   # sequential float sums, vectorised over the voxels
        live = counts > j
        for s, c in zip(sums, cols):
            s[live] = s[live] + c[heads[live] + j]
    nf = counts.astype(np.float32)
    out = [s / nf for s in sums]
    return np.stack(out[:3], axis=1), (out[3] if intensity is not None else None), counts


def test_downsample_matches_voxelgrid_on_the_c3_map(pkg, S, hipmem):
    cfg = S.config_c3()
    m = cfg["target"]                                     # the 8-scan map union, 1 M points
    rng = np.random.default_rng(2)
    inten = rng.uniform(0, 255, len(m)).astype(np.float32)
    ref_xyz, ref_i, counts = voxelgrid_numpy(m, 0.5, inten)
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5)
    d = [hipmem.upload(m[:, a]) for a in range(3)] + [hipmem.upload(inten)]
    o = [hipmem.upload(np.zeros(len(m), np.float32)) for _ in range(4)]
    n_out = ndt.voxelDownsampleDevice(d[0], d[1], d[2], len(m), 0.5, o[0], o[1], o[2], len(m), d_intensity=d[3], o_intensity=o[3])
    assert n_out == len(ref_xyz)
    import ctypes as C
    back = np.zeros((4, n_out), np.float32)
    for a in range(4):
        assert hipmem.rt.hipMemcpy(back[a].ctypes.data, C.c_void_p(o[a]), 4 * n_out, 2) == 0
    assert np.array_equal(back[:3].T, ref_xyz) and np.array_equal(back[3], ref_i)
    # ... and straight into the NDT target without a host round trip: same leaves as building from the host copy
    ndt.setInputTargetDevice(o[0], o[1], o[2], n_out)
    L_dev = ndt.getLeaves()
    ndt.setInputTarget(ref_xyz)
    L_host = ndt.getLeaves()
    for f in ("cell", "count", "mean", "cov"):
        assert np.array_equal(L_dev[f], L_host[f]), f
    # the host form, pcl::PointXYZI layout (stride 32, intensity at byte 16)
    aos = np.zeros((len(m), 8), np.float32)
    aos[:, :3] = m
    aos[:, 3] = 1.0
    aos[:, 4] = inten
    out = ndt.voxelDownsample(aos, 0.5, intensity_column=4)
    assert out.shape == (n_out, 8) and np.array_equal(out[:, :3], ref_xyz) and np.array_equal(out[:, 4], ref_i)
    # a capacity that is too small is refused with the size that is needed
    with pytest.raises(pkg.NdtError) as ei:
        ndt.voxelDownsampleDevice(d[0], d[1], d[2], len(m), 0.5, o[0], o[1], o[2], 100)
    assert ei.value.code == -1 and str(n_out) in str(ei.value)


def test_downsample_edge_cases_and_target_untouched(pkg, S, hipmem):
    cfg = S.config_c1()
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4)
    ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
    T0 = ndt.align(cfg["guess"])
    L0 = ndt.getLeaves()
    pts = cfg["target"].copy()
    pts[::13] = np.nan                                    # non-finite points are dropped
    pts[5::31, 2] = np.inf
    for leaf in (0.25, 1.0, 3.0):
        ref, _, _ = voxelgrid_numpy(pts, leaf)
        out = ndt.voxelDownsample(pts, leaf)
        assert np.array_equal(out, ref), leaf
    one = ndt.voxelDownsample(np.array([[1.0, 2.0, 3.0]], np.float32), 0.5)
    assert np.array_equal(one, [[1.0, 2.0, 3.0]])
    assert len(ndt.voxelDownsample(np.zeros((0, 3), np.float32), 0.5)) == 0
    assert len(ndt.voxelDownsample(np.full((7, 3), np.nan, np.float32), 0.5)) == 0
    with pytest.raises(pkg.NdtError) as ei:              # PCL refuses a grid of more than INT32_MAX cells
        ndt.voxelDownsample(np.array([[0, 0, 0], [4e6, 4e6, 4e6]], np.float32), 0.01)
    assert ei.value.code == -6
    # the engine's target and source are as they were
    assert np.array_equal(ndt.align(cfg["guess"]), T0)
    L1 = ndt.getLeaves()
    assert np.array_equal(L1["cell"], L0["cell"]) and np.array_equal(L1["cov"], L0["cov"])
    # between an asynchronous hand-off and its first consumer: the pending build is completed, not lost
    ndt.setInputTarget(cfg["target"][::2]); ndt.setInputSource(cfg["source"])
    ndt.voxelDownsample(pts, 1.0)
    assert ndt.getGridInfo()["n_target_points"] == len(cfg["target"][::2])
