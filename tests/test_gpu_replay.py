"""BASELINE config C5: sequential scan-to-scan odometry over a synthetic OS-2-128 stream with
the GPU engine inside the registration slot (slam-sam_amd/replay.py mirrors the per-keyframe
body of run/pipeline.cpp:494-610 and run/pipeline_lo_svn.cpp:376-388)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class OracleEngine:
    """The CPU oracle behind the same four calls (test-side checker only)."""

    def __init__(self, O, **kw):
        self.O, self.prm = O, O.default_params(num_threads=8, **kw)

    def setInputTarget(self, c): self.grid = self.O.Grid(c, self.prm)
    def setInputSource(self, c): self.src = c
    def align(self, guess):
        self.r = self.grid.align(self.src, guess)
        return self.r["T"]
    def getFinalNumIteration(self): return self.r["iterations"]


def test_replay_ndt_odometry(pkg, O, S):
    from slam_sam_amd import replay
    stream = replay.make_stream(n_frames=8)
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
    replay.run_lidar_odometry(ndt, stream[:3])  # warm-up (allocations)
    out = replay.run_lidar_odometry(ndt, stream)
    err = replay.trajectory_errors(out["poses"], stream)
    print("C5 NDT replay: %.1f Hz end to end, %.2f ms/frame, iterations %s, final drift %.4f m %.5f rad"
          % (out["hz"], out["ms"].mean(), out["iterations"], err[-1][0], err[-1][1]))
    assert err[-1][0] < 0.05 and err[-1][1] < 0.01          # 7 chained registrations
    assert max(e[0] for e in err) < 0.05
    # frame-by-frame agreement with the oracle run through the same loop
    ref = replay.run_lidar_odometry(OracleEngine(O, **kw), stream[:4])
    for a, b in zip(out["poses"][:4], ref["poses"]):
        dt, dr = S.pose_error(a, b)
        assert dt < 1e-3 and dr < 1e-4


def test_replay_ndt_odometry_device_resident(pkg, S):
    """Same odometry with the scans archived on the device (keyframe API): one upload per scan,
    the target assembled and the source taken from the archive.  Must track the host-path result."""
    from slam_sam_amd import replay
    stream = replay.make_stream(n_frames=8)
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    host = pkg.NormalDistributionsTransform(device_id=0, **kw)
    ref = replay.run_lidar_odometry(host, stream)
    dev = pkg.NormalDistributionsTransform(device_id=0, **kw)
    replay.run_lidar_odometry(dev, stream[:3], mode="ndt_keyframes")  # warm-up (allocations)
    out = replay.run_lidar_odometry(dev, stream, mode="ndt_keyframes")
    print("C5 NDT replay, device-resident keyframes: %.1f Hz end to end, %.2f ms/frame, iterations %s"
          % (out["hz"], out["ms"].mean(), out["iterations"]))
    assert dev.keyframeCount() == 2
    for a, b in zip(out["poses"], ref["poses"]):
        dt, dr = S.pose_error(a, b)       # the device moves the target with a double matrix (pcl semantics),
        assert dt < 1e-3 and dr < 1e-4    # the harness with NumPy: identical up to an f32 ulp per point
    err = replay.trajectory_errors(out["poses"], stream)
    assert max(e[0] for e in err) < 0.05


def test_replay_svn_odometry(pkg, S):
    from slam_sam_amd import replay
    stream = replay.make_stream(n_frames=5)
    svn = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0)
    svn.setParticleCount(20); svn.setMaxIterations(100); svn.setKernelBandwidth(5.0)   # config/register_config.json:13-19
    svn.setStepSize(0.05); svn.setEarlyStopThreshold(1e-4); svn.setOutlierRatio(0.55)
    # the lo_svn driver hands align() the INS pose as prior: ground truth + a few cm / mrad
    rng = np.random.default_rng(3)
    priors = [gt @ S.pose_matrix(*(rng.normal(0, 0.03, 3)), *(rng.normal(0, 0.003, 3))) for _, gt in stream]
    out = replay.run_lidar_odometry(svn, stream, mode="svn", priors=priors)
    err = replay.trajectory_errors(out["poses"], stream)
    perr = [S.pose_error(p, gt)[0] for p, (_, gt) in zip(priors, stream)]
    print("C5 SVN replay (K=20): %.2f Hz end to end, %.1f ms/frame, iterations %s, errors %s (priors %s)"
          % (out["hz"], out["ms"].mean(), out["iterations"], np.round([e[0] for e in err], 3), np.round(perr, 3)))
    # 100 iterations at step 0.05 do not converge in the reference either (output/output.txt:104);
    # the particle mean must still end closer to the truth than the prior it started from
    assert np.mean([e[0] for e in err[1:]]) < np.mean(perr[1:])
