"""Multi-grid target [RECALLED: tier4 ndt_omp MultiGridNormalDistributionsTransform; the reference only
names it in its build, CMakeLists.txt:41-42] through the C-ABI: addTarget / removeTarget /
createVoxelKdtree.  The union's neighbourhood is the radius search over every grid's centroids, so
score, gradient, Hessian and the pair count of the union are the SUMS of the per-grid KDTREE
evaluations -- which the oracle provides grid by grid (1e-9 of the norms against its f64 products,
the usual spread against its f32 products)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)


def _tiles(cfg, overlap=3.0):
    """three tiles along x that overlap by `overlap` metres (voxels in the overlap exist in two grids)"""
    t = cfg["target"]
    lo, hi = t[:, 0].min(), t[:, 0].max()
    e = np.linspace(lo, hi, 4)
    return [t[(t[:, 0] >= e[i] - overlap) & (t[:, 0] <= e[i + 1] + overlap)] for i in range(3)]


def _sum(ds):
    return dict(score=sum(d["score"] for d in ds), gradient=sum(d["gradient"] for d in ds),
                hessian=sum(d["hessian"] for d in ds), n_pairs=sum(d["n_pairs"] for d in ds))


def test_union_is_the_sum_of_the_grids(pkg, O, S):
    cfg = S.config_c2()
    tiles = _tiles(cfg)
    ndt = pkg.NormalDistributionsTransform(device_id=0, **KW)
    for i, t in enumerate(tiles):
        ndt.addTarget(t, 100 + i)
    assert ndt.targetCount() == 3
    with pytest.raises(pkg.NdtError):          # stored, but not a target yet
        ndt.setInputSource(cfg["source"]); ndt.evalDerivatives(np.zeros(6))
    ndt.createVoxelKdtree()
    ndt.setInputSource(cfg["source"])
    grids = [O.Grid(t, O.default_params(num_threads=4, **KW)) for t in tiles]
    assert ndt.getGridInfo()["n_leaves"] == sum(g.n_leaves for g in grids)
    L = ndt.getLeaves()
    assert len(L["cell"]) == sum(g.n_leaves for g in grids) and len(np.unique(L["cell"])) < len(L["cell"])
    rng = np.random.default_rng(3)
    p0 = O.matrix_to_pose(cfg["guess"])
    poses = np.stack([p0, O.matrix_to_pose(cfg["gt"]), p0 + rng.normal(0, 0.02, 6)])
    for hmode, oh in ((pkg.HESSIAN_FULL, 0), (pkg.HESSIAN_GAUSS_NEWTON, 1)):
        ndt.setParams(hessian_mode=hmode, search_method=pkg.DIRECT7)   # search_method is ignored by a multi-grid target
        got = ndt.evalDerivatives(poses)
        for p, e in zip(poses, got):
            okw = dict(num_threads=4, search_method=O.KDTREE, hessian_mode=oh, **KW)
            d = _sum([g.derivatives(cfg["source"], p, params=O.default_params(**okw)) for g in grids])
            x = _sum([g.derivatives(cfg["source"], p, params=O.default_params(pair_mode=2, **okw)) for g in grids])
            assert e["n_pairs"] == d["n_pairs"]
            assert e["score"] == pytest.approx(d["score"], rel=1e-8, abs=1e-9)
            gn, hn = np.linalg.norm(x["gradient"]), np.linalg.norm(x["hessian"])
            assert np.linalg.norm(e["gradient"] - x["gradient"]) <= 1e-9 * gn + 1e-9
            assert np.linalg.norm(e["hessian"] - x["hessian"]) <= 1e-9 * hn + 1e-9
            rg, rh = np.linalg.norm(d["gradient"] - x["gradient"]), np.linalg.norm(d["hessian"] - x["hessian"])
            assert np.linalg.norm(e["gradient"] - d["gradient"]) <= 1.01 * rg + 1e-9 * gn + 1e-9
            assert np.linalg.norm(e["hessian"] - d["hessian"]) <= 1.01 * rh + 1e-9 * hn + 1e-9
        # the packed 48-byte record format does not apply to a union (its leaves are chained through the 80-byte
        # record): selecting it changes no bit
        ndt.setRecordFormat(pkg.RECORDS_PACKED48)
        for e, e48 in zip(got, ndt.evalDerivatives(poses)):
            assert e48["score"] == e["score"] and np.array_equal(e48["hessian"], e["hessian"])
        ndt.setRecordFormat(pkg.RECORDS_F64)
        sc = ndt.scoreTransform(O.pose_to_matrix(poses[0]))
        assert sc["score"] == got[0]["score"] and sc["n_pairs"] == got[0]["n_pairs"]
    # align on the union converges to the ground truth like the single-grid target does
    ndt.setParams(hessian_mode=pkg.HESSIAN_FULL)
    T = ndt.align(cfg["guess"])
    dt, dr = S.pose_error(T, cfg["gt"])
    assert ndt.hasConverged() and dt < 0.05 and dr < 0.035
    # removing a tile invalidates the union until it is re-created; the result then changes
    ndt.removeTarget(101)
    with pytest.raises(pkg.NdtError):
        ndt.evalDerivatives(poses[0])
    ndt.createVoxelKdtree()
    e2 = ndt.evalDerivatives(poses[1])[0]
    okw = dict(num_threads=4, search_method=O.KDTREE, hessian_mode=0, pair_mode=2, **KW)
    x2 = _sum([g.derivatives(cfg["source"], poses[1], params=O.default_params(**okw)) for g in (grids[0], grids[2])])
    assert e2["n_pairs"] == x2["n_pairs"]
    assert np.linalg.norm(e2["hessian"] - x2["hessian"]) <= 1e-9 * np.linalg.norm(x2["hessian"])
    with pytest.raises(pkg.NdtError):
        ndt.removeTarget(101)
    # a plain setInputTarget replaces the union (the stored grids stay)
    ndt.setInputTarget(tiles[0])
    ndt.setParams(search_method=pkg.KDTREE)
    e3 = ndt.evalDerivatives(poses[1])[0]
    x3 = grids[0].derivatives(cfg["source"], poses[1], params=O.default_params(**okw))
    assert e3["n_pairs"] == x3["n_pairs"] and ndt.targetCount() == 2


def test_one_grid_union_equals_kdtree_target_and_stacked_grids_add_up(pkg, O, S):
    """A union of ONE grid is the KDTREE evaluation of that cloud, bit for bit (same records, same
    27-cell order); the same cloud stored under five ids gives every cell a chain of five leaves
    (135 candidates per point: the chain walk, not the 27-entry list, carries them) and five times
    the sums."""
    cfg = S.config_c1()
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    ref = pkg.NormalDistributionsTransform(device_id=0, search_method=pkg.KDTREE, **kw)
    ref.setInputTarget(cfg["target"]); ref.setInputSource(cfg["source"])
    p = O.matrix_to_pose(cfg["guess"])
    e0 = ref.evalDerivatives(p)[0]
    ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
    ndt.addTarget(cfg["target"], 7)
    ndt.createVoxelKdtree()
    ndt.setInputSource(cfg["source"])
    e1 = ndt.evalDerivatives(p)[0]
    assert e1["score"] == e0["score"] and e1["n_pairs"] == e0["n_pairs"]
    assert np.array_equal(e1["gradient"], e0["gradient"]) and np.array_equal(e1["hessian"], e0["hessian"])
    assert np.array_equal(ndt.align(cfg["guess"]), ref.align(cfg["guess"]))
    for k in range(4):
        ndt.addTarget(cfg["target"], 20 + k)
    ndt.createVoxelKdtree()
    e5 = ndt.evalDerivatives(p)[0]
    assert e5["n_pairs"] == 5 * e0["n_pairs"] and e5["n_with_neighbors"] == e0["n_with_neighbors"]
    assert e5["score"] == pytest.approx(5 * e0["score"], rel=1e-12)
    np.testing.assert_allclose(e5["hessian"], 5 * e0["hessian"], rtol=1e-10, atol=1e-9 * np.abs(e0["hessian"]).max())
    # nine copies: more than 27 centroids within the radius of many points -- their waves leave the
    # filtered listing for the chain walk over all occupied cells
    for k in range(4):
        ndt.addTarget(cfg["target"], 40 + k)
    ndt.createVoxelKdtree()
    e9 = ndt.evalDerivatives(p)[0]
    assert e9["n_pairs"] == 9 * e0["n_pairs"]
    assert e9["score"] == pytest.approx(9 * e0["score"], rel=1e-12)
    np.testing.assert_allclose(e9["gradient"], 9 * e0["gradient"], rtol=1e-10, atol=1e-9 * np.abs(e0["gradient"]).max())
    for k in range(4):
        ndt.removeTarget(40 + k)
    # grid parameters changed after a tile was stored: refused until the tile is added again
    ndt.setResolution(2.0)
    with pytest.raises(pkg.NdtError):
        ndt.createVoxelKdtree()
    with pytest.raises(pkg.NdtError):
        pkg.NormalDistributionsTransform(device_id=0, **kw).createVoxelKdtree()   # nothing stored
