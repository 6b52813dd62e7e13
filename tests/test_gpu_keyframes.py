"""device-resident keyframe archive and sliding-window target assembly
(ref: run/pipeline_ligo_tc.cpp:519-529) against the host-assembled target."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def host_transform_f64(T, pts):
    """pcl::transformPointCloud with a double matrix: f64 products summed left to right, one
    rounding to f32 (elementwise NumPy ops round like the scalar code, no FMA)."""
    x, y, z = (pts[:, k].astype(np.float64) for k in range(3))
    out = np.empty_like(pts)
    for r in range(3):
        out[:, r] = (((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]).astype(np.float32)
    return out


def test_sliding_window_target_matches_host_assembly(pkg, O, S):
    from slam_sam_amd import replay
    stream = replay.make_stream(n_frames=6, beams=64, cols=512)
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4,
                                           max_iterations=35)
    for k, (scan, _) in enumerate(stream[:5]):
        ndt.putKeyframe(100 + k, scan)
    assert ndt.keyframeCount() == 5
    ids = [100, 101, 102, 103, 104]
    poses = [gt for _, gt in stream[:5]]
    ndt.setInputTargetFromKeyframes(ids, poses)
    dev_leaves = ndt.getLeaves()
    host_target = np.concatenate([host_transform_f64(T, scan) for (scan, _), T in zip(stream[:5], poses)])
    assert ndt.getGridInfo()["n_target_points"] == len(host_target)
    ref = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4,
                                           max_iterations=35)
    ref.setInputTarget(host_target)
    host_leaves = ref.getLeaves()
    for k in ("cell", "count", "mean", "cov", "icov"):
        assert np.array_equal(dev_leaves[k], host_leaves[k]), k     # bit-identical target
    # and against the oracle on the host-assembled cloud
    grid = O.Grid(host_target, O.default_params(resolution=1.0))
    assert np.array_equal(dev_leaves["cell"], grid.export()["cell"])
    # register the 6th scan into the window
    scan, gt = stream[5]
    ndt.setInputSource(scan)
    guess = poses[-1] @ np.linalg.inv(poses[-2]) @ poses[-1]   # constant-velocity prediction
    T = ndt.align(guess)
    dt, dr = S.pose_error(T, gt)
    assert dt < 0.03 and dr < 0.005
    # window slides: drop the oldest, add the newest
    ndt.eraseKeyframe(100)
    ndt.putKeyframe(105, scan)
    ndt.setInputTargetFromKeyframes([101, 102, 103, 104, 105], poses[1:] + [T])
    assert ndt.keyframeCount() == 5 and ndt.getGridInfo()["n_leaves"] > 100
    with pytest.raises(pkg.NdtError):
        ndt.setInputTargetFromKeyframes([100], [np.eye(4)])        # erased id


def test_source_from_keyframe_is_a_view_of_the_archive(pkg, S):
    """setInputSourceFromKeyframe registers the archived scan in place (no copy): same result as handing the scan over
    from the host; erasing or replacing that keyframe unsets the source instead of leaving a dangling view."""
    cfg = S.config_c2()
    kw = dict(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    ref = pkg.NormalDistributionsTransform(**kw)
    ref.setInputTarget(cfg["target"]); ref.setInputSource(cfg["source"])
    T0 = ref.align(cfg["guess"]); r0 = ref.getResult()
    ndt = pkg.NormalDistributionsTransform(**kw)
    ndt.setInputTarget(cfg["target"])
    src = cfg["source"].copy()
    ndt.putKeyframe(7, src); src[:] = np.nan          # (consumed when the call returns)
    ndt.setInputSourceFromKeyframe(7)
    for _ in range(2):
        T = ndt.align(cfg["guess"]); r = ndt.getResult()
        assert np.array_equal(T, T0) and r["score"] == r0["score"] and np.array_equal(r["hessian"], r0["hessian"])
    ndt.putKeyframe(8, cfg["source"][::2])              # another keyframe: the view stays
    assert np.array_equal(ndt.align(cfg["guess"]), T0)
    ndt.eraseKeyframe(7)
    with pytest.raises(pkg.NdtError) as ei:
        ndt.align(cfg["guess"])
    assert ei.value.code == -5                          # NDT_ERR_NO_SOURCE
    ndt.putKeyframe(9, cfg["source"]); ndt.setInputSourceFromKeyframe(9)
    assert np.array_equal(ndt.align(cfg["guess"]), T0)
    ndt.putKeyframe(9, cfg["source"][::3])              # replaced under the view: unset, not stale
    with pytest.raises(pkg.NdtError) as ei:
        ndt.align(cfg["guess"])
    assert ei.value.code == -5
