"""Randomised GPU-vs-oracle parity sweep: many small seeded clouds of awkward shapes (tiny, ragged,
duplicated points, coplanar / collinear voxels, non-finite entries, negative and large offsets,
odd leaf sizes and min-point thresholds) through the C-ABI, in all three neighbourhood modes and
all three Hessian modes.  Same tolerances as tests/test_gpu_parity.py: voxel membership, point
counts and pair counts bit-exact; means 1e-12; covariances 1e-9 of their largest entry (looser
where the reference's own single-pass formula cancels); score 1e-8 (1e-9 on ordinary clouds; the
near-degenerate voxels of the `dupes` clouds carry the covariance tolerance into the inverse);
gradient / Hessian 5e-6 of
their norms (the oracle rounds the per-pair Jacobian products to f32 as the reference does, the
kernel keeps f64: 1e-8..1e-7 on ordinary clouds, up to 1.2e-6 on collinear voxels whose inflated
inverse covariances are ill-conditioned)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_cloud(rng, kind, n, offset):
    if kind == "blobs":      # a few Gaussian clusters
        c = rng.uniform(-8, 8, (max(1, n // 300), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.35, (n, 3))
    elif kind == "planes":   # thin intersecting planes (coplanar voxels: eigenvalue inflation at work)
        p = rng.uniform(-6, 6, (n, 3))
        k = rng.integers(0, 3, n)
        p[np.arange(n), k] = np.round(p[np.arange(n), k] / 3.0) * 3.0 + rng.normal(0, 0.004, n)
    elif kind == "lines":    # collinear voxels: two tiny eigenvalues
        t = rng.uniform(-10, 10, n)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        p = t[:, None] * d + rng.normal(0, 0.002, (n, 3)) + rng.integers(-1, 2, (n, 1)) * np.array([0.0, 1.7, 0.0])
    elif kind == "dupes":    # every point repeated: zero-covariance voxels are rejected, mixed ones inflated
        q = rng.uniform(-4, 4, (max(1, n // 8), 3))
        p = np.repeat(q, 8, axis=0)[:n]
        p[::3] += rng.normal(0, 0.05, p[::3].shape)
    else:                    # uniform box
        p = rng.uniform(-5, 5, (n, 3))
    return (p + offset).astype(np.float32)


CASES = []
_rng = np.random.default_rng(20241004)
for i, kind in enumerate(["blobs", "planes", "lines", "dupes", "box"] * 10):
    CASES.append(dict(
        seed=1000 + i, kind=kind,
        n=int(_rng.choice([7, 60, 700, 5000, 20000])),
        res=float(_rng.choice([0.3, 0.5, 1.0, 1.7, 3.0])),
        min_pts=int(_rng.choice([3, 6, 6, 10])),
        offset=_rng.choice([0.0, -37.5, 120.25, 1500.0]) * np.array([1.0, -0.5, 0.1]),
        bad=bool(_rng.integers(0, 2)),
    ))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-n%d-r%g-m%d%s" % (c["kind"], c["n"], c["res"], c["min_pts"], "-nan" if c["bad"] else ""))
def test_random_cloud_parity(pkg, O, S, case):
    rng = np.random.default_rng(case["seed"])
    tgt = make_cloud(rng, case["kind"], case["n"], case["offset"])
    # the source: the target seen from a slightly different pose, thinned, plus a few far outliers
    dT = S.pose_matrix(*(rng.normal(0, 0.05, 3)), *(rng.normal(0, 0.01, 3)))
    src = S.transform(np.linalg.inv(dT), tgt[rng.random(len(tgt)) < 0.7].astype(np.float64)).astype(np.float32)
    src = np.concatenate([src, (rng.uniform(-60, 60, (5, 3)) + case["offset"]).astype(np.float32)])
    if case["bad"]:
        tgt = tgt.copy(); src = src.copy()
        tgt[rng.integers(0, len(tgt), max(1, len(tgt) // 50))] = np.nan
        src[rng.integers(0, len(src), 2)] = [np.inf, 0.0, np.nan]
    kw = dict(resolution=case["res"], step_size=0.1, trans_epsilon=1e-4, max_iterations=20,
              min_points_per_voxel=case["min_pts"])
    grid = O.Grid(tgt, O.default_params(num_threads=4, **kw))
    n, info = pkg.backend_info()
    assert n > 0, info
    ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
    if grid.n_leaves == 0 and not np.isfinite(tgt).all(axis=1).any():
        with pytest.raises(pkg.NdtError):
            ndt.setInputTarget(tgt)
        return
    ndt.setInputTarget(tgt)
    gi = ndt.getGridInfo()
    assert np.array_equal(gi["min_b"], grid.min_b) and np.array_equal(gi["div_b"], grid.div_b)
    L, OL = ndt.getLeaves(), grid.export()
    assert np.array_equal(L["cell"], OL["cell"]) and np.array_equal(L["count"], OL["count"])
    if len(OL["cell"]):
        np.testing.assert_allclose(L["mean"], OL["mean"], rtol=1e-12, atol=0)
        # the single-pass covariance loses (|mean| / spread)^2 * eps in both implementations
        spread = np.sqrt(np.maximum(np.abs(OL["cov"]).max(axis=(1, 2)), 1e-30))
        loss = 1e-9 + 64 * np.finfo(np.float64).eps * (np.abs(OL["mean"]).max(axis=1) / spread) ** 2
        scale = np.abs(OL["cov"]).max(axis=(1, 2))
        assert ((np.abs(L["cov"] - OL["cov"]).max(axis=(1, 2)) / scale) < loss).all()
    ndt.setInputSource(src)
    if not (OL["count"] > 0).any():   # no voxel passed the eigenvalue checks: loud refusal, as for an empty target
        with pytest.raises(pkg.NdtError) as ei:
            ndt.evalDerivatives(np.zeros(6))
        assert ei.value.code == -4
        return
    p0 = O.matrix_to_pose(dT)
    poses = np.stack([p0, p0 + rng.normal(0, 0.02, 6), np.zeros(6)])
    for method, omethod in ((pkg.DIRECT7, O.DIRECT7), (pkg.DIRECT1, O.DIRECT1), (pkg.KDTREE, O.KDTREE)):
        for hmode in (pkg.HESSIAN_FULL, pkg.HESSIAN_GAUSS_NEWTON):
            ndt.setParams(search_method=method, hessian_mode=hmode)
            oprm = O.default_params(num_threads=4, search_method=omethod,
                                    hessian_mode=1 if hmode == pkg.HESSIAN_GAUSS_NEWTON else 0, **kw)
            got = ndt.evalDerivatives(poses)
            for p, e in zip(poses, got):
                d = grid.derivatives(src, p, params=oprm)
                assert e["n_pairs"] == d["n_pairs"] and e["n_with_neighbors"] == d["n_with_neighbors"]
                assert e["score"] == pytest.approx(d["score"], rel=1e-8, abs=1e-9)
                gn, hn = np.linalg.norm(d["gradient"]), np.linalg.norm(d["hessian"])
                assert np.linalg.norm(e["gradient"] - d["gradient"]) <= 5e-6 * gn + 1e-9
                assert np.linalg.norm(e["hessian"] - d["hessian"]) <= 5e-6 * hn + 1e-9
            # score + gradient only (the line search's cheap evaluation) agrees with the full one
            e0 = ndt.evalDerivatives(poses[0], compute_hessian=False)[0]
            assert e0["score"] == got[0]["score"] and np.array_equal(e0["gradient"], got[0]["gradient"])
