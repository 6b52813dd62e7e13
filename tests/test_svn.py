"""SVN-NDT: the oracle's restatement of svn_ndt::align is pinned against the
reference's own SVN test, and the HIP engine's batched-kernel SVN is checked against it."""
import numpy as np
import pytest


# --------------------------------------------------------------------------------- CPU
def test_se3_exp_log_roundtrip(O):
    rng = np.random.default_rng(0)
    for _ in range(100):
        w = rng.normal(size=3)
        w *= rng.uniform(0, 3.0) / np.linalg.norm(w)  # |omega| < pi: the principal branch
        xi = np.concatenate([w, rng.normal(size=3) * 3])
        T = O.se3_expmap(xi)
        assert np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-12
        np.testing.assert_allclose(O.se3_logmap(T), xi, atol=1e-9)
    np.testing.assert_allclose(O.se3_logmap(np.eye(4)), 0, atol=1e-15)
    # first-order behaviour: Exp([0, v]) is a pure translation
    np.testing.assert_allclose(O.se3_expmap([0, 0, 0, 1, 2, 3])[:3, 3], [1, 2, 3])


def test_reference_svn_k10_test_pins_oracle(O, S):
    """ConvergenceComparison.SvnNdtK10 (ref: test_svn_ndt.cpp:205-257): resolution 1.0,
    min 3 points per voxel, DIRECT7, K = 10, <= 100 iterations, h = 1.0, stop 1e-4, step 1.0,
    Gauss-Newton Hessian; must converge in < 100 iterations within 0.05 m / 0.035 rad.  The
    reference draws its particles from a wall-clock seed (svn_ndt_impl.hpp:712); here fixed."""
    src, tgt, gt, guess = O.two_plane_fixture()
    prm = O.default_params(resolution=1.0, min_points_per_voxel=3, search_method=O.DIRECT7,
                           hessian_mode=O.HESSIAN_GAUSS_NEWTON, add_ridge=1, num_threads=4)
    grid = O.Grid(tgt, prm)
    particles = O.svn_sample_particles(guess, 10, 2)
    r = O.svn_align(grid, src, guess, particles, prm, max_iterations=100, kernel_bandwidth=1.0,
                    step_size=1.0, stop_threshold=1e-4)
    assert r["converged"] and r["iterations"] < 100
    trans_err, rot_err = S.se3_log_error(r["pose"], gt)
    assert trans_err < 0.05 and rot_err < 0.035
    ev = np.linalg.eigvalsh(r["covariance"])
    assert (ev >= 1e-9 * (1 - 1e-6)).all()  # eigenvalue floor, ref :932-949


def test_svn_single_particle_is_newton_with_gn_hessian(O, S):
    """K = 1: no kernel interaction, the update is -(H + 1e-6 I)^-1 g.  (The reference's
    SvnNdtK1_Newton test asks for the full analytic Hessian with step 1.0; with the vendored
    math that first step has norm 0.64 from the test's initial guess and the iteration diverges,
    so that test cannot be reproduced -- recorded in DESIGN.md.)"""
    src, tgt, gt, guess = O.two_plane_fixture()
    prm = O.default_params(resolution=1.0, min_points_per_voxel=3, hessian_mode=O.HESSIAN_GAUSS_NEWTON,
                           add_ridge=1, num_threads=4)
    grid = O.Grid(tgt, prm)
    r = O.svn_align(grid, src[::4], guess, guess[None], prm, max_iterations=100, step_size=1.0)
    assert r["converged"]
    t, rr = S.se3_log_error(r["pose"], gt)
    assert t < 0.05 and rr < 0.035
    # K = 1: 1e-6 * sigma^2 on the diagonal (ref :921-925), then the 1e-9 eigenvalue floor (:932-949)
    np.testing.assert_allclose(np.diag(r["covariance"]),
                               np.maximum(1e-6 * np.array([.01, .01, .02, .05, .05, .05]) ** 2, 1e-9), rtol=1e-9)


# --------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_svn_matches_oracle(pkg, O, S):
    src, tgt, gt, guess = O.two_plane_fixture()
    src = src[::3].copy()
    K = 10
    prm = O.default_params(resolution=1.0, min_points_per_voxel=3, hessian_mode=O.HESSIAN_GAUSS_NEWTON,
                           add_ridge=1, num_threads=8)
    grid = O.Grid(tgt, prm)
    particles = pkg.svn_sample_particles(guess, K, seed=7)
    # the product's sampler and the oracle's draw the same particles from the same seed
    np.testing.assert_allclose(particles, O.svn_sample_particles(guess, K, 7), atol=1e-12)
    ref = O.svn_align(grid, src, guess, particles, prm, max_iterations=60, kernel_bandwidth=1.0,
                      step_size=1.0, stop_threshold=1e-4)
    svn = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0, min_points_per_voxel=3)
    svn.setParticleCount(K); svn.setMaxIterations(60); svn.setKernelBandwidth(1.0)
    svn.setStepSize(1.0); svn.setEarlyStopThreshold(1e-4)
    svn.setNeighborhoodSearchMethod(pkg.DIRECT7)
    svn.setInputTarget(tgt)
    got = svn.align(src, guess, particles=particles)
    assert got["converged"] == ref["converged"]
    assert abs(got["iterations"] - ref["iterations"]) <= 2
    dt, dr = S.pose_error(got["final_pose"], ref["pose"])
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    # particle clouds agree, hence so does the covariance estimate
    for a, b in zip(got["particles"], ref["particles"]):
        pt, pr = S.pose_error(a, b)
        assert pt < 2e-3 and pr < 2e-4
    np.testing.assert_allclose(got["final_covariance"], ref["covariance"], rtol=0.05,
                               atol=0.02 * np.abs(ref["covariance"]).max())
    t, r = S.se3_log_error(got["final_pose"], gt)
    assert t < 0.05 and r < 0.035
    # one iteration = one launch for all K particles
    one = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0, min_points_per_voxel=3)
    one.setParticleCount(K); one.setMaxIterations(1)
    one.setInputTarget(tgt)
    n0 = one.getTiming()["n_eval_launches"]
    one.align(src, guess, particles=particles)
    assert one.getTiming()["n_eval_launches"] - n0 == 1


@pytest.mark.gpu
def test_hip_svn_failure_conventions(pkg, S):
    """No target: the prior comes back, not converged, identity covariance (ref :682-702)."""
    svn = pkg.SvnNormalDistributionsTransform(device_id=0)
    svn.setParticleCount(4)
    with pytest.raises(pkg.NdtError):
        svn.align(np.zeros((10, 3), np.float32), np.eye(4))
