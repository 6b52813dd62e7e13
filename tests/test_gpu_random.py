"""Randomised GPU-vs-oracle parity sweep: many small seeded clouds of awkward shapes (tiny, ragged,
duplicated points, coplanar / collinear voxels, non-finite entries, negative and large offsets,
odd leaf sizes and min-point thresholds) through the C-ABI, in all three neighbourhood modes and
all three Hessian modes.  Same tolerances as tests/test_gpu_parity.py: voxel membership, point
counts and pair counts bit-exact; means 1e-12; covariances 1e-9 of their largest entry, looser by
the cancellation loss of the reference's own single-pass formula (`amp` below, which also widens
the derivative tolerances of a cloud whose thinnest voxel is ill-conditioned); score 1e-8.
Gradient / Hessian: the reference forms the per-pair products in f32 (float x_trans4, c_inv4,
point_gradient4: svn_ndt_impl.hpp:412-415), the kernel in f64.  So the kernel is compared (1) with
the oracle's f64 evaluation of the same formulas (`pair_mode=2`, a test seam): 1e-9 of the norms;
(2) with the oracle's reference arithmetic: no farther from it than that arithmetic is from f64
(measured 1e-8..1e-7 on ordinary clouds, up to 4e-5 on collinear / few-point voxels)."""
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
for i, kind in enumerate(["blobs", "planes", "lines", "dupes", "box"] * 20):
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
    amp = 0.0
    if len(OL["cell"]):
        np.testing.assert_allclose(L["mean"], OL["mean"], rtol=1e-12, atol=0)
        # The reference's single-pass covariance (sum xx^T / n - mean mean^T) loses
        # (|mean| / spread)^2 * eps to cancellation in BOTH implementations, which sum in different
        # orders: a millimetre-thin voxel 1.5 km from the origin keeps ~4 digits.  The inflated
        # inverse, and with it score / gradient / Hessian, inherit that: tolerances scale with it.
        spread = np.sqrt(np.maximum(np.abs(OL["cov"]).max(axis=(1, 2)), 1e-30))
        loss = 1e-9 + 64 * np.finfo(np.float64).eps * (np.abs(OL["mean"]).max(axis=1) / spread) ** 2
        scale = np.abs(OL["cov"]).max(axis=(1, 2))
        assert ((np.abs(L["cov"] - OL["cov"]).max(axis=(1, 2)) / scale) < loss).all()
        amp = float(loss.max())
    ndt.setInputSource(src)
    if not (OL["count"] > 0).any():   # no voxel passed the eigenvalue checks: loud refusal, as for an empty target
        with pytest.raises(pkg.NdtError) as ei:
            ndt.evalDerivatives(np.zeros(6))
        assert ei.value.code == -4
        return
    p0 = O.matrix_to_pose(dT)
    poses = np.stack([p0, p0 + rng.normal(0, 0.02, 6), np.zeros(6)])
    for method, omethod in ((pkg.DIRECT7, O.DIRECT7), (pkg.DIRECT1, O.DIRECT1), (pkg.KDTREE, O.KDTREE),
                            (pkg.DIRECT26, O.DIRECT26)):
        for hmode in (pkg.HESSIAN_FULL, pkg.HESSIAN_GAUSS_NEWTON):
            ndt.setParams(search_method=method, hessian_mode=hmode)
            okw = dict(num_threads=4, search_method=omethod,
                       hessian_mode=1 if hmode == pkg.HESSIAN_GAUSS_NEWTON else 0, **kw)
            oprm = O.default_params(**okw)                 # the reference's arithmetic (f32 products)
            oprm64 = O.default_params(pair_mode=2, **okw)  # the same formulas carried in f64 (test seam)
            got = ndt.evalDerivatives(poses)
            for p, e in zip(poses, got):
                d = grid.derivatives(src, p, params=oprm)
                x = grid.derivatives(src, p, params=oprm64)
                assert e["n_pairs"] == d["n_pairs"] and e["n_with_neighbors"] == d["n_with_neighbors"]
                assert e["score"] == pytest.approx(d["score"], rel=1e-8 + amp, abs=1e-9)
                gn, hn = np.linalg.norm(x["gradient"]), np.linalg.norm(x["hessian"])
                # (1) the kernel evaluates the reference's formulas: against their f64 evaluation
                #     it agrees to 1e-9 (+ what the two covariance tables differ by)
                assert np.linalg.norm(e["gradient"] - x["gradient"]) <= (1e-9 + 30 * amp) * gn + 1e-9
                assert np.linalg.norm(e["hessian"] - x["hessian"]) <= (1e-9 + 30 * amp) * hn + 1e-9
                # (2) against the reference's own f32 products it is as far off as they are from f64
                rg = np.linalg.norm(d["gradient"] - x["gradient"])
                rh = np.linalg.norm(d["hessian"] - x["hessian"])
                assert np.linalg.norm(e["gradient"] - d["gradient"]) <= 1.01 * rg + (1e-9 + 30 * amp) * gn + 1e-9
                assert np.linalg.norm(e["hessian"] - d["hessian"]) <= 1.01 * rh + (1e-9 + 30 * amp) * hn + 1e-9
                assert rg <= 2e-4 * gn + 1e-9 and rh <= 2e-4 * hn + 1e-9   # and that spread is f32-sized
            # score + gradient only (the line search's cheap evaluation) agrees with the full one
            e0 = ndt.evalDerivatives(poses[0], compute_hessian=False)[0]
            assert e0["score"] == got[0]["score"] and np.array_equal(e0["gradient"], got[0]["gradient"])
            # ... and so does the score-only kernel (ndt_score_transform), bit for bit
            sc = ndt.scoreTransform(O.pose_to_matrix(poses[0]))
            assert sc["score"] == got[0]["score"] and sc["n_pairs"] == got[0]["n_pairs"]
            if got[0]["n_with_neighbors"]:
                assert sc["nvtl"] == got[0]["nvtl_sum"] / got[0]["n_with_neighbors"]
