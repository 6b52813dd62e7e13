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


C5_SVN = dict(particle_count=20, max_iterations=100, kernel_bandwidth=5.0, step_size=0.05, stop_threshold=1e-4,
              outlier_ratio=0.55)   # config/register_config.json:13-19


def _c5_svn(pkg, max_iterations=None):
    svn = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0)
    svn.setParticleCount(C5_SVN["particle_count"]); svn.setMaxIterations(max_iterations or C5_SVN["max_iterations"])
    svn.setKernelBandwidth(C5_SVN["kernel_bandwidth"]); svn.setStepSize(C5_SVN["step_size"])
    svn.setEarlyStopThreshold(C5_SVN["stop_threshold"]); svn.setOutlierRatio(C5_SVN["outlier_ratio"])
    return svn


def _c5_oracle_svn(O, target, scan, prior, particles, max_iterations):
    prm = O.default_params(resolution=1.0, outlier_ratio=C5_SVN["outlier_ratio"], hessian_mode=O.HESSIAN_GAUSS_NEWTON,
                           add_ridge=1, num_threads=16)
    return O.svn_align(O.Grid(target, prm), scan, prior, particles, prm, max_iterations=max_iterations,
                       kernel_bandwidth=C5_SVN["kernel_bandwidth"], step_size=C5_SVN["step_size"],
                       stop_threshold=C5_SVN["stop_threshold"])


def test_svn_on_a_c5_frame_pair_follows_the_oracle(pkg, O, S):
    """SVN-NDT at the workload that matters (ref: svn_ndt_impl.hpp:740-949): one C5 frame pair --
    a 128 x 1024 scan (131 072 points) against the previous scan as target, K = 20 particles, the settings of
    config/register_config.json -- through the HIP engine and through the oracle FROM THE SAME PARTICLES.  After 1, 5
    and 20 iterations every particle agrees to 2 mm / 0.2 mrad, and so do the mean pose and the covariance estimate."""
    from slam_sam_amd import replay
    (scan0, T0), (scan1, T1) = replay.make_stream(n_frames=2)
    target = S.transform(T0, scan0)                                   # ref: run/pipeline_lo_svn.cpp:376-386
    prior = T1 @ S.pose_matrix(0.03, -0.02, 0.01, 0.003, -0.002, 0.002)   # an INS prior: ground truth + cm / mrad
    K = C5_SVN["particle_count"]
    particles = pkg.svn_sample_particles(prior, K, seed=11)
    np.testing.assert_allclose(particles, O.svn_sample_particles(prior, K, 11), atol=1e-12)
    assert len(scan1) == 131072
    for iters in (1, 5, 20):
        ref = _c5_oracle_svn(O, target, scan1, prior, particles, iters)
        svn = _c5_svn(pkg, iters)
        svn.setInputTarget(target)
        got = svn.align(scan1, prior, particles=particles)
        assert got["iterations"] == ref["iterations"] == iters
        worst_t = worst_r = 0.0
        for a, b in zip(got["particles"], ref["particles"]):
            pt, pr = S.pose_error(a, b)
            worst_t, worst_r = max(worst_t, pt), max(worst_r, pr)
        dt, dr = S.pose_error(got["final_pose"], ref["pose"])
        print("C5 SVN vs oracle after %2d iterations: particles within %.2e m %.2e rad, mean pose %.2e m %.2e rad"
              % (iters, worst_t, worst_r, dt, dr))
        # (asked for: 2 mm / 0.2 mrad; measured on MI355X: 2e-11 m / 1e-11 rad after 20 iterations)
        assert worst_t < 1e-6 and worst_r < 1e-7, (iters, worst_t, worst_r)
        assert dt < 1e-6 and dr < 1e-7
        np.testing.assert_allclose(got["final_covariance"], ref["covariance"], rtol=0.02,
                                   atol=0.01 * np.abs(ref["covariance"]).max())
        svn.close() if hasattr(svn, "close") else None


def test_replay_svn_odometry(pkg, O, S):
    from slam_sam_amd import replay
    stream = replay.make_stream(n_frames=5)
    svn = _c5_svn(pkg)
    # the lo_svn driver hands align() the INS pose as prior: ground truth + a few cm / mrad
    rng = np.random.default_rng(3)
    priors = [gt @ S.pose_matrix(*(rng.normal(0, 0.03, 3)), *(rng.normal(0, 0.003, 3))) for _, gt in stream]
    out = replay.run_lidar_odometry(svn, stream, mode="svn", priors=priors)
    err = replay.trajectory_errors(out["poses"], stream)
    perr = [S.pose_error(p, gt)[0] for p, (_, gt) in zip(priors, stream)]
    print("C5 SVN replay (K=20): %.2f Hz end to end, %.1f ms/frame, iterations %s, errors %s (priors %s)"
          % (out["hz"], out["ms"].mean(), out["iterations"], np.round([e[0] for e in err], 3), np.round(perr, 3)))
    # 100 iterations at step 0.05 do not converge in the reference either (output/output.txt:104): what the replay is
    # held to is the ORACLE's run of the same frames -- same target (the engine's own previous estimate), same prior,
    # same particles (the replay's per-frame seed), the full 100 iterations: the same pose to a micrometre, same iteration count
    for k in (1, 2):
        target = S.transform(out["poses"][k - 1], stream[k - 1][0])
        particles = O.svn_sample_particles(priors[k], C5_SVN["particle_count"], k)   # run_lidar_odometry: seed = svn_seed + k
        ref = _c5_oracle_svn(O, target, stream[k][0], priors[k], particles, C5_SVN["max_iterations"])
        dt, dr = S.pose_error(out["poses"][k], ref["pose"])
        print("C5 SVN replay frame %d vs oracle (%d / %d iterations): %.2e m %.2e rad" % (k, out["iterations"][k - 1], ref["iterations"], dt, dr))
        assert out["iterations"][k - 1] == ref["iterations"]
        assert dt < 1e-6 and dr < 1e-7, (k, dt, dr)   # (measured 2.5e-10 m / 3e-11 rad after the full 100 iterations)
    assert np.mean([e[0] for e in err[1:]]) < np.mean(perr[1:])   # ... and still ends closer to the truth than the priors
