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


def _k1_config(O):
    """ConvergenceComparison.SvnNdtK1_Newton (ref: test_svn_ndt.cpp:262-316): resolution 1.0,
    min 3 points per voxel, DIRECT7, K = 1, <= 100 iterations, stop 1e-4, step 1.0 and
    setUseGaussNewtonHessian(false) -- the full analytic Hessian of svn_ndt_impl.hpp:472-494."""
    src, tgt, gt, guess = O.two_plane_fixture()
    prm = O.default_params(resolution=1.0, min_points_per_voxel=3, search_method=O.DIRECT7,
                           hessian_mode=O.HESSIAN_FULL, add_ridge=1, num_threads=4)
    return src, tgt, gt, guess, prm


def _passes_reference_assertions(r, gt, S):
    t, rot = S.se3_log_error(r["pose"], gt)
    return bool(r["converged"]) and r["iterations"] < 100 and t < 0.05 and rot < 0.035


@pytest.mark.xfail(strict=True, reason="the reference's SvnNdtK1_Newton test cannot pass with the vendored "
                   "math: the score is not concave at the test's initial guess (see the next two tests and "
                   "DESIGN.md section 2)")
def test_reference_svn_k1_newton_test(O, S):
    """The reference's third test, as written, through the oracle.  The reference draws even
    the single particle around the prior from a wall-clock seed (svn_ndt_impl.hpp:708-716);
    here the particle is the prior itself (the noise-free case)."""
    src, tgt, gt, guess, prm = _k1_config(O)
    grid = O.Grid(tgt, prm)
    r = O.svn_align(grid, src, guess, guess[None], prm, max_iterations=100, kernel_bandwidth=1.0,
                    step_size=1.0, stop_threshold=1e-4)
    assert _passes_reference_assertions(r, gt, S)


def test_reference_svn_k1_newton_fails_for_every_sampled_particle(O, S):
    """... and no draw of the particle rescues it: 0 of 24 seeds of the reference's own
    sampling distribution (sigma 0.01/0.01/0.02 rad, 0.05 m, :709) meet the test's assertions."""
    src, tgt, gt, guess, prm = _k1_config(O)
    grid = O.Grid(tgt, prm)
    passed = 0
    for seed in range(24):
        part = O.svn_sample_particles(guess, 1, seed)
        r = O.svn_align(grid, src, guess, part, prm, max_iterations=100, kernel_bandwidth=1.0,
                        step_size=1.0, stop_threshold=1e-4)
        passed += _passes_reference_assertions(r, gt, S)
    assert passed == 0


def test_reference_svn_k1_newton_failure_is_the_objective_not_the_restatement(O, S):
    """Why it fails, from numbers alone.  K = 1 makes the SVN update the plain Newton step
    -(H + 1e-6 I)^-1 g (kernel value 1, kernel gradient 0, :797-830).  At the test's initial
    guess (3.7 cm / 67 mrad from ground truth; 67 mrad is 0.67 m at the planes' 10 m edge,
    against voxel sigmas of ~3 cm) the score is not concave:
      * the full analytic Hessian has an eigenvalue of +6e6 beside -4e6 ... -4e3;
      * central differences of the GRADIENT confirm it (same sign and size), so it is the
        objective's curvature, not an artefact of the angle tables, of the [x,y,z,r,p,y] ->
        [r,p,y,x,y,z] permutation (a symmetric relabelling), or of the ridge (1e-6 beside 1e3+);
      * the Gauss-Newton Hessian (svn default) is negative definite there and converges;
      * at ground truth the full Hessian is negative definite but its weakest eigenvalue is 1e-4
        of the strongest (both planes contain the x axis, so x translation is constrained only
        by the planes' edges) and it is lost within a tenth of the guess error."""
    src, tgt, gt, guess, prm = _k1_config(O)
    prm.add_ridge = 0
    grid = O.Grid(tgt, prm)
    p = O.matrix_to_pose(guess)
    d = grid.derivatives(src, p, T=guess)
    ev = np.linalg.eigvalsh(d["hessian"])
    assert ev[-1] > 1e6 and ev[0] < -1e6
    # finite differences of the gradient in the consistent parametrisation (T built from p)
    prm64 = O.default_params(resolution=1.0, min_points_per_voxel=3, search_method=O.DIRECT7,
                             hessian_mode=O.HESSIAN_FULL, add_ridge=0, num_threads=4, pair_mode=2)
    H = np.zeros((6, 6))
    h = 1e-5
    for i in range(6):
        e = np.zeros(6); e[i] = h
        gp = grid.derivatives(src, p + e, T=O.pose_to_matrix(p + e), params=prm64, compute_hessian=False)["gradient"]
        gm = grid.derivatives(src, p - e, T=O.pose_to_matrix(p - e), params=prm64, compute_hessian=False)["gradient"]
        H[:, i] = (gp - gm) / (2 * h)
    ev_fd = np.linalg.eigvalsh(0.5 * (H + H.T))
    assert ev_fd[-1] > 1e6 and abs(ev_fd[-1] - ev[-1]) < 0.1 * ev[-1]
    # Gauss-Newton: negative definite at the same pose
    prm.hessian_mode = O.HESSIAN_GAUSS_NEWTON
    ev_gn = np.linalg.eigvalsh(grid.derivatives(src, p, T=guess)["hessian"])
    assert ev_gn[-1] < -1e4
    # ground truth: full Hessian negative definite, conditioning ~1e4
    prm.hessian_mode = O.HESSIAN_FULL
    ev_gt = np.linalg.eigvalsh(grid.derivatives(src, O.matrix_to_pose(gt), T=gt)["hessian"])
    assert ev_gt[-1] < 0 and ev_gt[0] / ev_gt[-1] > 1e3
    # the Newton iteration with the full Hessian converges from ground truth itself but not
    # from a tenth of the test's guess error
    xi = -0.1 * np.array([0.05, -0.02, 0.04, 0.02, -0.01, 0.03])
    prm.add_ridge = 1
    near = gt @ O.se3_expmap(xi)
    r0 = O.svn_align(grid, src, gt, gt[None], prm, max_iterations=100, step_size=1.0)
    r1 = O.svn_align(grid, src, near, near[None], prm, max_iterations=100, step_size=1.0)
    assert _passes_reference_assertions(r0, gt, S)
    assert not _passes_reference_assertions(r1, gt, S)


def test_svn_single_particle_is_newton_with_gn_hessian(O, S):
    """K = 1 with the svn default (Gauss-Newton Hessian): no kernel interaction, the update is
    -(H + 1e-6 I)^-1 g, and it meets the tolerances the reference's K = 1 test asks for."""
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
def test_hip_svn_k1_full_hessian_follows_the_oracle(pkg, O, S):
    """The reference's SvnNdtK1_Newton configuration through the HIP engine: the iteration the
    oracle shows to diverge (tests above) is the one the kernel computes -- the first Newton steps
    agree with the oracle's (later ones are chaotic), and the HIP engine does not meet the
    reference test's assertions either."""
    src, tgt, gt, guess, prm = _k1_config(O)
    grid = O.Grid(tgt, prm)
    svn = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0, min_points_per_voxel=3)
    svn.setUseGaussNewtonHessian(False)
    svn.setNeighborhoodSearchMethod(pkg.DIRECT7)
    svn.setParticleCount(1); svn.setKernelBandwidth(1.0); svn.setStepSize(1.0); svn.setEarlyStopThreshold(1e-4)
    svn.setInputTarget(tgt)
    for n_it, tol_t, tol_r in ((1, 1e-4, 1e-5), (2, 2e-3, 2e-4)):
        ref = O.svn_align(grid, src, guess, guess[None], prm, max_iterations=n_it, step_size=1.0)
        svn.setMaxIterations(n_it)
        got = svn.align(src, guess, particles=guess[None])
        dt, dr = S.pose_error(got["particles"][0], ref["particles"][0])
        assert dt < tol_t and dr < tol_r, (n_it, dt, dr)
    # the first step is the 0.64-long one that leaves the basin
    one = O.svn_align(grid, src, guess, guess[None], prm, max_iterations=1, step_size=1.0)
    assert 0.5 < one["log_mean_update"][0] < 0.8
    svn.setMaxIterations(100)
    full = svn.align(src, guess, particles=guess[None])
    t, r = S.se3_log_error(full["final_pose"], gt)
    assert not (full["converged"] and full["iterations"] < 100 and t < 0.05 and r < 0.035)


@pytest.mark.gpu
def test_hip_svn_failure_conventions(pkg, S):
    """No target: the prior comes back, not converged, identity covariance (ref :682-702)."""
    svn = pkg.SvnNormalDistributionsTransform(device_id=0)
    svn.setParticleCount(4)
    with pytest.raises(pkg.NdtError):
        svn.align(np.zeros((10, 3), np.float32), np.eye(4))
