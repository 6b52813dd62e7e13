"""CPU tests of the oracle itself: pinned against the reference's own test for this path,
against committed golden vectors, an independent NumPy restatement and finite differences.
None of this touches the HIP product path."""
import hashlib
import os

import numpy as np
import pytest


# ---------------------------------------------------------------------------------------
# the reference's own test: extern/svn_ndt/test/test_svn_ndt.cpp:138-199
# ---------------------------------------------------------------------------------------
def test_reference_fixture_pins_oracle(O, S):
    """ConvergenceComparison.PclOmp restated: resolution 1.0, DIRECT7, 50 iterations max,
    epsilon 1e-4, step 0.1 (ref :147-151); converged, < 50 iterations, < 0.05 m and
    < 0.035 rad from ground truth in the Logmap metric (ref :185-198)."""
    src, tgt, gt, guess = O.two_plane_fixture()
    assert len(src) == 35912  # 134*134*2, ref :55-64
    prm = O.default_params(resolution=1.0, search_method=O.DIRECT7, max_iterations=50,
                           trans_epsilon=1e-4, step_size=0.1, num_threads=4)
    grid = O.Grid(tgt, prm)
    r = grid.align(src, guess)
    assert r["converged"]
    assert r["iterations"] < 50
    trans_err, rot_err = S.se3_log_error(r["T"], gt)
    assert trans_err < 0.05
    assert rot_err < 0.035


def test_reference_fixture_golden(O, golden_dir):
    z = np.load(os.path.join(golden_dir, "g2_reference_fixture_expect.npz"))
    src, tgt, gt, guess = O.two_plane_fixture()
    np.testing.assert_allclose(gt, z["gt"], atol=1e-15)
    np.testing.assert_allclose(guess, z["guess"], atol=1e-15)
    assert hashlib.sha256(src.tobytes()).hexdigest() == str(z["src_sha"])
    if hashlib.sha256(tgt.tobytes()).hexdigest() != str(z["tgt_sha"]):
        pytest.skip("std::normal_distribution stream differs from the image the fixture was made on")
    prm = O.default_params(resolution=1.0, max_iterations=50, trans_epsilon=1e-4, step_size=0.1)
    grid = O.Grid(tgt, prm)
    assert grid.n_leaves == int(z["n_leaves"])
    r = grid.align(src, guess)
    assert r["iterations"] == int(z["align_iterations"])
    assert r["n_evaluations"] == int(z["align_n_evaluations"])
    np.testing.assert_allclose(r["T"], z["align_T"], atol=1e-7)
    np.testing.assert_allclose(r["log_step"], z["align_log_step"], rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------
# committed golden vectors (oracle regression pin)
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "g1_two_plane_3k.npz"))


def test_golden_inputs_match_generator(S, g1):
    src, tgt, gt, guess = S.two_planes(seed=2024, max_points=3000)
    assert np.array_equal(src, g1["source"]) and np.array_equal(tgt, g1["target"])
    np.testing.assert_allclose(guess, g1["guess"], atol=1e-15)


def test_golden_leaves(O, g1):
    prm = O.default_params(resolution=1.0)
    grid = O.Grid(g1["target"], prm)
    L = grid.export()
    assert np.array_equal(L["cell"], g1["leaf_cell"])
    assert np.array_equal(L["count"], g1["leaf_count"])
    assert np.array_equal(grid.min_b, g1["min_b"]) and np.array_equal(grid.div_b, g1["div_b"])
    for k in ("mean", "cov", "icov", "evals"):
        np.testing.assert_allclose(L[k], g1["leaf_" + k], rtol=1e-12, atol=1e-14)


def test_golden_derivatives(O, g1):
    prm = O.default_params(resolution=1.0)
    grid = O.Grid(g1["target"], prm)
    for i, p in enumerate(g1["poses"]):
        d = grid.derivatives(g1["source"], p)
        assert d["n_pairs"] == g1["n_pairs"][i] and d["n_with_neighbors"] == g1["n_with"][i]
        np.testing.assert_allclose(d["score"], g1["score"][i], rtol=1e-12)
        np.testing.assert_allclose(d["gradient"], g1["gradient"][i], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(d["hessian"], g1["hessian"][i], rtol=1e-10, atol=1e-7)
    d = grid.derivatives(g1["source"], g1["poses"][0],
                         params=O.default_params(resolution=1.0, hessian_mode=O.HESSIAN_GAUSS_NEWTON, add_ridge=1))
    np.testing.assert_allclose(d["hessian"], g1["gn_hessian"], rtol=1e-10, atol=1e-7)
    d = grid.derivatives(g1["source"], g1["poses"][0], params=O.default_params(resolution=1.0, search_method=O.DIRECT1))
    assert d["n_pairs"] == int(g1["d1_n_pairs"])
    np.testing.assert_allclose(d["gradient"], g1["d1_gradient"], rtol=1e-10, atol=1e-9)


def test_golden_align(O, g1):
    prm = O.default_params(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50)
    grid = O.Grid(g1["target"], prm)
    r = grid.align(g1["source"], g1["guess"])
    assert r["iterations"] == int(g1["align_iterations"])
    assert r["n_evaluations"] == int(g1["align_n_evaluations"])
    np.testing.assert_allclose(r["log_pose"], g1["align_log_pose"], atol=1e-9)
    np.testing.assert_allclose(r["T"], g1["align_T"], atol=1e-7)


# ---------------------------------------------------------------------------------------
# independent NumPy restatement of the voxel statistics
# ---------------------------------------------------------------------------------------
def numpy_leaves(tgt, leaf, min_pts=6, ratio=0.01):
    """voxel_grid_covariance_impl.hpp:218-343 in NumPy (f32 index math, f64 statistics)."""
    inv = np.float32(1.0) / np.float32(leaf)
    fin = np.isfinite(tgt).all(1)
    pts = tgt[fin]
    mn, mx = pts.min(0), pts.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(pts * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    out = {}
    for c in np.unique(idx):
        P = pts[idx == c].astype(np.float64)
        n = len(P)
        if n < min_pts:
            continue
        mu = P.sum(0) / n
        cov = (P.T @ P) / n - np.outer(mu, mu)
        cov *= n / (n - 1.0)
        ev, V = np.linalg.eigh(cov)
        if ev[0] < 0 or ev[1] < 0 or ev[2] < 1e-12:
            continue
        fl = max(1e-12, ev[2] * ratio)
        if ev[0] < fl or ev[1] < fl:
            ev = np.maximum(ev, [fl, fl, -np.inf])
            cov = V @ np.diag(ev) @ V.T
        icov = np.linalg.inv(cov)
        if not np.isfinite(icov).all() or np.abs(icov).max() > 1e12:
            continue
        out[int(c)] = (n, mu, cov, icov, ev)
    return min_b, div, out


@pytest.mark.parametrize("shift", [0.0, 3000.0])
def test_numpy_mirror_leaves(O, S, shift):
    src, tgt, gt, guess = S.two_planes(seed=5, max_points=6000)
    tgt = (tgt + np.float32(shift)).astype(np.float32)  # +3 km: NED-scale coordinates
    prm = O.default_params(resolution=1.0)
    grid = O.Grid(tgt, prm)
    L = grid.export()
    min_b, div, ref = numpy_leaves(tgt, 1.0)
    assert np.array_equal(min_b, grid.min_b) and np.array_equal(div, grid.div_b)
    assert sorted(ref.keys()) == list(L["cell"])
    # single-pass covariance cancels ~ |mu|^2 * eps / sigma^2: allow for it at 3 km
    rtol = 1e-9 if shift == 0 else 2e-4
    for i, c in enumerate(L["cell"]):
        n, mu, cov, icov, ev = ref[int(c)]
        assert n == L["count"][i]
        np.testing.assert_allclose(L["mean"][i], mu, rtol=1e-14)
        np.testing.assert_allclose(L["cov"][i], cov, rtol=rtol, atol=rtol * np.abs(cov).max())
        np.testing.assert_allclose(L["icov"][i], icov, rtol=rtol * 10, atol=rtol * 10 * np.abs(icov).max())
        np.testing.assert_allclose(L["evals"][i], ev, rtol=rtol * 10, atol=rtol * np.abs(ev).max())


# ---------------------------------------------------------------------------------------
# derivatives: tables and finite differences on a frozen pair set
# ---------------------------------------------------------------------------------------
def test_angle_tables_vs_numeric(O, S):
    p = np.array([0.1, 0.2, 0.3, 0.4, -0.3, 0.7])
    j, h = O.angle_tables(p)
    a, e, E = p[3:], 1e-4, np.eye(3)
    R = lambda v: S.rot_xyz(*v)  # noqa: E731
    d1 = [(R(a + e * E[i]) - R(a - e * E[i])) / (2 * e) for i in range(3)]
    expect_j = [d1[0][1], d1[0][2], d1[1][0], d1[1][1], d1[1][2], d1[2][0], d1[2][1], d1[2][2]]
    np.testing.assert_allclose(j, np.array(expect_j), atol=1e-6)

    def d2(i, k):
        return (R(a + e * E[i] + e * E[k]) - R(a + e * E[i] - e * E[k]) - R(a - e * E[i] + e * E[k])
                + R(a - e * E[i] - e * E[k])) / (4 * e * e)
    expect_h = np.array([d2(0, 0)[1], d2(0, 0)[2], d2(0, 1)[1], d2(0, 1)[2], d2(0, 2)[1], d2(0, 2)[2],
                         d2(1, 1)[0], d2(1, 1)[1], d2(1, 1)[2], d2(1, 2)[0], d2(1, 2)[1], d2(1, 2)[2],
                         d2(2, 2)[0], d2(2, 2)[1], d2(2, 2)[2]])
    bad = np.argwhere(np.abs(h - expect_h) > 1e-4)
    # The reference's table (svn_ndt_impl.hpp:305, same as PCL's h_ang_d1) has +sin(pitch) where
    # d2x'/dpitch^2 has -sin(pitch); it is restated as is.  Nothing else may deviate.
    assert bad.tolist() == [[6, 2]]
    assert h[6, 2] == pytest.approx(-expect_h[6, 2], abs=1e-6)


def test_gradient_hessian_vs_finite_differences(O, S):
    src, tgt, gt, guess = S.two_planes(seed=9, max_points=8000)
    src = src[::3].copy()
    prm = O.default_params(resolution=1.0)
    grid = O.Grid(tgt, prm)
    L = grid.export()
    p0 = O.matrix_to_pose(guess)
    d1, d2, _ = O.gauss_constants(1.0, 0.55)
    T0 = O.pose_to_matrix(p0)
    xt = (src.astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]).astype(np.float32)
    pi, li = [], []
    for i, x in enumerate(xt):
        for r in grid.neighbors(x):
            pi.append(i)
            li.append(r)
    pi, li = np.array(pi), np.array(li)

    def score(p):  # smooth: the pair set is frozen
        X = src[pi].astype(np.float64) @ S.rot_xyz(*p[3:]).T + p[:3]
        xr = X - L["mean"][li]
        q = np.einsum("ni,nij,nj->n", xr, L["icov"][li], xr)
        return np.sum(-d1 * np.exp(-d2 * q / 2))

    d = grid.derivatives(src, p0)
    assert d["n_pairs"] == len(pi)
    assert d["score"] == pytest.approx(score(p0), rel=1e-7)
    E, h = np.eye(6), 1e-5
    g_fd = np.array([(score(p0 + h * e) - score(p0 - h * e)) / (2 * h) for e in E])
    assert np.linalg.norm(g_fd - d["gradient"]) / np.linalg.norm(g_fd) < 1e-5
    h = 1e-4
    H_fd = np.array([[(score(p0 + h * E[i] + h * E[j]) - score(p0 + h * E[i] - h * E[j])
                       - score(p0 - h * E[i] + h * E[j]) + score(p0 - h * E[i] - h * E[j])) / (4 * h * h)
                      for j in range(6)] for i in range(6)])
    diff = np.abs(H_fd - d["hessian"]) / np.linalg.norm(H_fd)
    diff[4, 4] = 0.0  # carries the reference's h_ang_d1 sign (see test_angle_tables_vs_numeric)
    assert diff.max() < 2e-4


def test_gauss_constants(O):
    for res, o in ((1.0, 0.55), (0.5, 0.55), (2.0, 0.3)):
        d1, d2, d3 = O.gauss_constants(res, o)
        c1, c2 = 10 * (1 - o), o / res ** 3
        e3 = -np.log(c2)
        e1 = -np.log(c1 + c2) - e3
        e2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - e3) / e1)
        assert (d1, d2, d3) == pytest.approx((e1, e2, e3), rel=1e-14)
        assert d1 < 0 < d2


def test_pose_matrix_roundtrip(O, S):
    rng = np.random.default_rng(0)
    for _ in range(50):
        p = np.concatenate([rng.uniform(-50, 50, 3), rng.uniform(-1.2, 1.2, 3)])
        T = O.pose_to_matrix(p)
        np.testing.assert_allclose(T[:3, :3], S.rot_xyz(*p[3:]), atol=3e-7)
        q = O.matrix_to_pose(T)
        # Euler extraction may return the equivalent (roll+pi, pi-pitch, yaw+pi) triple
        np.testing.assert_allclose(O.pose_to_matrix(q), T, atol=2e-6)


# ---------------------------------------------------------------------------------------
# edge cases the reference guards
# ---------------------------------------------------------------------------------------
def test_empty_and_degenerate_targets(O, S):
    prm = O.default_params(resolution=1.0)
    assert O.Grid(np.zeros((0, 3), np.float32), prm).n_leaves == 0
    # fewer than min_points_per_voxel everywhere -> no valid leaf (ref :270-273)
    assert O.Grid(np.random.default_rng(0).uniform(-50, 50, (200, 3)).astype(np.float32), prm).n_leaves == 0
    # all points identical: zero covariance -> largest eigenvalue < 1e-12 -> discarded (ref :303-309)
    assert O.Grid(np.ones((50, 3), np.float32), prm).n_leaves == 0
    # NaN / Inf points are skipped (ref :219)
    src, tgt, gt, guess = S.two_planes(seed=3, max_points=4000)
    dirty = tgt.copy()
    dirty[::17] = np.nan
    dirty[5::29, 1] = np.inf
    clean = dirty[np.isfinite(dirty).all(1)]
    a, b = O.Grid(dirty, prm).export(), O.Grid(clean, prm).export()
    assert np.array_equal(a["cell"], b["cell"]) and np.array_equal(a["mean"], b["mean"])


def test_no_overlap_gives_zero_score(O, S):
    src, tgt, gt, guess = S.two_planes(seed=3, max_points=4000)
    prm = O.default_params(resolution=1.0)
    grid = O.Grid(tgt, prm)
    far = S.pose_matrix(500.0, 0, 0, 0, 0, 0)
    d = grid.derivatives(src, O.matrix_to_pose(far))
    assert d["n_pairs"] == 0 and d["score"] == 0.0 and not d["gradient"].any()
    r = grid.align(src, far)
    assert r["iterations"] == 0  # zero Newton step: loop exits at once


def test_kdtree_radius_search_matches_brute_force(O, S):
    """KDTREE mode (ref: voxel_grid_covariance_impl.hpp:505-554): the oracle's 27-cell scan must
    return exactly the leaves whose f32 centroid is within one leaf size (f32 distance, strict <)."""
    cfg = S.config_c1()
    for res in (1.0, 0.7):
        prm = O.default_params(resolution=res, search_method=O.KDTREE)
        grid = O.Grid(cfg["target"], prm)
        cent = grid.export()["mean"].astype(np.float32)
        r2 = np.float32(np.float64(np.float32(res)) ** 2)
        xt = S.transform(cfg["guess"], cfg["source"])
        pts = np.concatenate([xt[::41], xt[:50] + np.float32(30.0)])  # incl. points outside the grid box
        for p in pts:
            d = (p[None] - cent) ** 2
            d2 = (d[:, 0] + d[:, 1]) + d[:, 2]
            assert sorted(grid.neighbors(p, O.KDTREE)) == np.nonzero(d2 < r2)[0].tolist()
    r = grid.align(cfg["source"], cfg["guess"])
    assert r["converged"] and S.pose_error(r["T"], cfg["gt"])[0] < 0.05


def test_km_scale_gap_is_the_references_f32_products(O, S):
    """SURVEY 8d asks for a +3 km (NED-scale) variant.  There the oracle's two arithmetics --
    the reference's f32 per-pair products (pair_mode 0, ref: svn_ndt_impl.hpp:412-415,449-494)
    and the same formulas with f64 products (pair_mode 2) -- end 12 mm apart on the two-plane
    fixture, although they agree to 0.1 mm at the origin and the voxel statistics do not depend
    on the summation order (<= 32 f32 points per voxel sum exactly in f64): x' is quantised to
    0.24 mm (f32, as the reference's transformPointCloud), the score becomes a staircase, and the
    fixture's unconstrained x direction turns a last-bit difference into another line-search
    decision.  Parity at km scale is therefore defined against the f64 trajectory (GPU test
    test_km_scale_coordinates), and this test pins the size of the reference's own ambiguity."""
    cfg = S.config_c1()
    gaps = {}
    for off in ([0.0, 0.0, 0.0], [3000.0, -2000.0, 100.0]):
        o = np.array(off, np.float32)
        tgt = (cfg["target"] + o).astype(np.float32)
        shift = np.eye(4)
        shift[:3, 3] = o
        guess, gt = shift @ cfg["guess"], shift @ cfg["gt"]
        res = []
        for order in (slice(None), slice(None, None, -1)):   # summation order of the voxel sums
            grid = O.Grid(tgt[order].copy(), O.default_params(resolution=1.0))
            for pm in (0, 2):
                prm = O.default_params(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50,
                                       pair_mode=pm)
                r = grid.align(cfg["source"], guess, params=prm)
                assert r["converged"] and S.pose_error(r["T"], gt)[0] < 0.05
                res.append(r["T"])
        assert S.pose_error(res[0], res[2])[0] < 1e-9 and S.pose_error(res[1], res[3])[0] < 1e-9  # order-free
        gaps[off[0]] = S.pose_error(res[0], res[1])[0]
    assert gaps[0.0] < 2e-4
    assert 5e-3 < gaps[3000.0] < 2e-2


def test_km_scale_gap_vanishes_on_non_degenerate_geometry(O, S):
    """The other half of the km-scale question: on the C2 street canyon, where no direction
    is free, the oracle's two arithmetics -- the reference's f32 per-pair products and the same formulas in f64 --
    end within 1e-9 m of each other at +3 km as at the origin, in the same number of iterations.  The 12 mm of
    test_km_scale_gap_is_the_references_f32_products is a property of the two-plane fixture (its unconstrained x
    direction), not of the drivers' NED map frame."""
    cfg = S.config_c2()
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35, num_threads=8)
    for off in ([0.0, 0.0, 0.0], [3000.0, -2000.0, 100.0]):
        o = np.array(off)
        tgt = (cfg["target"].astype(np.float64) + o).astype(np.float32)
        shift = np.eye(4)
        shift[:3, 3] = o
        guess, gt = shift @ cfg["guess"], shift @ cfg["gt"]
        grid = O.Grid(tgt, O.default_params(**kw))
        r0 = grid.align(cfg["source"], guess)
        r2 = grid.align(cfg["source"], guess, params=O.default_params(pair_mode=2, **kw))
        assert r0["converged"] and r0["iterations"] == r2["iterations"]
        assert S.pose_error(r0["T"], r2["T"])[0] < 1e-7
        assert S.pose_error(r0["T"], gt)[0] < 0.01
