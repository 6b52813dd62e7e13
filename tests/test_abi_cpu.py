"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every
symbol include/ndt_hip.h declares, fails loudly without a GPU (no fallback), and the host
Newton / More-Thuente driver reproduces the oracle's trajectory when fed the same
evaluations (no compute kernels run here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "ndt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(ndt_[a-z0-9_]+)\s*\(", txt))
    # function-pointer typedefs are types, not exports
    return sorted(n for n in names if n not in ("ndt_eval_fn", "ndt_allreduce_fn"))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libndt_hip.so lacks %s" % name
    assert sorted(pkg.ABI_SYMBOLS) == declared  # the Python binding covers the whole ABI
    assert L.ndt_abi_version() == 3


def test_production_library_is_not_steered_by_the_environment(pkg):
    """the production library reads four documented OPERATIONAL variables and nothing else --
    every tuning / A-B switch is ndt_tuning (ndt_set_tuning), every fault-injection seam lives in the seams variant."""
    import subprocess
    out = subprocess.run(["strings", os.path.join(ROOT, "slam-sam_amd", "libndt_hip.so")], capture_output=True, text=True).stdout
    names = sorted({ln for ln in out.splitlines() if re.fullmatch(r"NDT_[A-Z0-9_]+", ln)})
    assert names == ["NDT_COMM_TIMEOUT_S", "NDT_HANDOFF", "NDT_PRELAUNCH", "NDT_UPLOAD_THREADS"], names
    for src in ("ndt_handle.hip", "ndt_handoff.hip", "ndt_evaluate.hip", "ndt_keyframes.hip", "ndt_keepwarm.hip", "ndt_derivs.hip", "ndt_target.hip", "ndt_comm.cpp", "ndt_repack_pool.h", "ndt_newton.cpp", "ndt_svn.cpp"):
        txt = open(os.path.join(ROOT, "slam-sam_amd", "csrc", src)).read()
        for m in re.finditer(r'getenv\("(NDT_[A-Z0-9_]+)"\)', txt):
            assert m.group(1) in names or m.group(1).startswith("NDT_DEBUG_"), (src, m.group(1))
    # (the NDT_DEBUG_* seams are compiled only with -DNDT_TEST_SEAMS)
    seams = subprocess.run(["strings", os.path.join(ROOT, "slam-sam_amd", "libndt_hip_seams.so")], capture_output=True, text=True).stdout
    assert "NDT_DEBUG_MUTE_ROW_AT" in seams and "NDT_DEBUG_MUTE_ROW_AT" not in out


def test_tuning_round_trip_and_validation(pkg):
    t0 = pkg.get_tuning()
    assert t0["deriv_block"] == 0 and t0["deriv_summer"] == 1 and t0["deriv_dedicated"] == 1 and t0["deriv_xcd"] == 1
    assert t0["bucket_build"] == 1 and t0["fused_sort"] == 1 and t0["speculate_first"] == 1 and t0["prelaunch_streams"] == 2
    try:
        assert pkg.set_tuning(deriv_block=512, fused_sort=0)["deriv_block"] == 512
        assert pkg.get_tuning()["fused_sort"] == 0
        for bad in (dict(deriv_block=100), dict(deriv_block=2048), dict(deriv_xcd=3), dict(bounds_unroll=5),
                    dict(finalize_threads=128), dict(prelaunch_streams=3), dict(bucket_tile=1000), dict(deriv_single_level_max=0)):
            with pytest.raises(pkg.NdtError):
                pkg.set_tuning(**bad)
        assert pkg.get_tuning()["deriv_block"] == 512       # a refused call changes nothing
        # the environment does NOT reach the production library ...
        os.environ["NDT_DERIV_BLOCK"] = "256"
        assert pkg.get_tuning()["deriv_block"] == 512
        # ... unless a harness asks the Python mirror to translate it
        assert pkg.apply_env_tuning()["deriv_block"] == 256
    finally:
        os.environ.pop("NDT_DERIV_BLOCK", None)
        pkg.set_tuning(**t0)
    assert pkg.get_tuning() == t0


def test_finishing_wave_tables_cover_every_block_shape(pkg):
    """k_derivatives' finishing waves (round 5): for every block shape the tables name one finishing wave per SIMD, every
    other wave's points are expanded by exactly one of them, and the 13-wave block of the 200 k-point scan gets the
    assignment DESIGN 4.1 describes (wave 12 itself only; waves 1, 2, 3 four items each)."""
    L = pkg.lib()
    for threads in range(64, 1025, 64):
        ow, fw = C.c_uint(), C.c_uint()
        assert L.ndt_debug_item_owners(threads, C.byref(ow), C.byref(fw)) == 0
        nw = threads // 64
        fins = [(fw.value >> (4 * q)) & 15 for q in range(4)]
        owners = [(ow.value >> (2 * i)) & 3 for i in range(nw)]
        nfin = min(4, nw)
        assert all(f < nw and f % 4 == q for q, f in enumerate(fins[:nfin])) and all(f == 15 for f in fins[nfin:])
        load = [0] * 4
        for i in range(nw):
            assert fins[owners[i]] != 15
            if i in fins:
                assert owners[i] == i % 4        # a finishing wave expands its own points
            load[owners[i]] += 1
        assert sum(load) == nw and (nw < 4 or min(load[:nfin]) >= 1)
        if nw == 13:
            assert fins == [12, 1, 2, 3] and load == [1, 4, 4, 4]
            assert [owners[i] for i in (0, 4, 8)] == [1, 2, 3]
        if nw in (8, 12, 16):
            assert fins == [0, 1, 2, 3] and load == [nw // 4] * 4
    assert L.ndt_debug_item_owners(100, C.byref(ow), C.byref(fw)) == -1


def test_abi_signatures_have_no_torch_types():
    txt = open(os.path.join(ROOT, "include", "ndt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # declarations only
    assert "torch" not in txt and "at::" not in txt and "std::" not in txt


def test_default_params_follow_reference_defaults(pkg):
    p = pkg.default_params()
    assert p.resolution == 1.0 and p.outlier_ratio == 0.55       # svn_ndt_impl.hpp:62
    assert p.min_points_per_voxel == 6 and p.eig_inflation_ratio == 0.01  # voxel_grid_covariance.h:153-154
    assert p.search_method == pkg.DIRECT7
    assert p.hessian_mode == pkg.HESSIAN_FULL and p.use_line_search == 1


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    n, info = pkg.backend_info()
    if n > 0:
        pytest.skip("a HIP device is present: " + info)
    with pytest.raises(pkg.NdtError) as ei:
        pkg.NormalDistributionsTransform()
    assert ei.value.code == -2  # NDT_ERR_NO_DEVICE


def test_invalid_arguments_are_rejected(pkg):
    L = pkg.lib()
    h = C.c_void_p()
    bad = pkg.default_params(resolution=0.0)
    assert L.ndt_create(C.byref(bad), C.byref(h)) == -1
    bad = pkg.default_params(search_method=17)
    assert L.ndt_create(C.byref(bad), C.byref(h)) == -1
    assert L.ndt_align(None, None, None) == -1
    assert L.ndt_get_grid_info(None, None) == -1
    assert L.ndt_destroy(None) == 0
    # multi-grid and view entry points: null handle / null cloud
    assert L.ndt_multigrid_add_target(None, 1, None, 0, 12) == -1
    assert L.ndt_multigrid_remove_target(None, 1) == -1
    assert L.ndt_multigrid_create_kdtree(None) == -1
    assert L.ndt_multigrid_count(None) == -1
    assert L.ndt_set_source_device_view(None, None, None, None, 0) == -1


def test_shard_range_partitions(pkg):
    for n in (0, 1, 7, 200000, 200003):
        for w in (1, 2, 3, 4, 8):
            covered = 0
            for r in range(w):
                b, c = pkg.shard_range(n, r, w)
                assert b == covered
                covered += c
                assert abs(c - n / w) < 1
            assert covered == n


def oracle_evaluator(pkg, O, grid, src, prm, log=None):
    def fn(pose, T, need_h):
        d = grid.derivatives(src, pose, T=T, compute_hessian=need_h, params=prm)
        if log is not None:
            log.append((pose.copy(), T.copy(), need_h))
        return pkg.pack_eval(d["score"], d["gradient"], d["hessian"], d["nvtl_sum"],
                             d["n_with_neighbors"], d["n_pairs"])
    return fn


@pytest.mark.parametrize("line_search", [1, 0])
def test_newton_driver_matches_oracle_trajectory(pkg, O, S, line_search):
    """Same evaluations in -> the product's host loop must take the oracle's steps."""
    src, tgt, gt, guess = S.two_planes(seed=2024, max_points=3000)
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50, use_line_search=line_search)
    # the packed evaluation carries the upper triangle of H only: give the oracle the same H
    oprm = O.default_params(symmetrize_hessian=1, **kw)
    grid = O.Grid(tgt, oprm)
    ref = grid.align(src, guess)
    log = []
    got = pkg.newton_align(pkg.default_params(**kw), len(src), guess,
                           oracle_evaluator(pkg, O, grid, src, oprm, log))
    assert got["converged"] == ref["converged"]
    assert got["iterations"] == ref["iterations"]
    assert got["n_evaluations"] == ref["n_evaluations"] == len(log)
    np.testing.assert_allclose(got["pose"], ref["pose"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(got["T"].astype(np.float32), ref["T"].astype(np.float32))
    np.testing.assert_allclose(got["hessian"], ref["hessian"], rtol=1e-12)
    assert got["score"] == pytest.approx(ref["score"], rel=1e-12)
    assert got["nvtl"] == pytest.approx(ref["nvtl"], rel=1e-12)
    # the f32 transform the product builds from a pose is bit-identical to the oracle's
    for pose, T, _ in log[1:]:
        np.testing.assert_array_equal(T.astype(np.float32), O.pose_to_matrix(pose).astype(np.float32))
    # first evaluation uses the guess matrix itself and Eigen-style Euler angles
    np.testing.assert_array_equal(log[0][1].astype(np.float32), np.asarray(guess, np.float32))
    np.testing.assert_allclose(log[0][0], O.matrix_to_pose(guess), atol=1e-12)


def test_newton_driver_regularization_and_ridge(pkg, O, S):
    src, tgt, gt, guess = S.two_planes(seed=2024, max_points=3000)
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=30)
    reg_pose = S.pose_matrix(0.45, 0.02, 0.3, 0, 0, 0.26)
    oprm = O.default_params(use_regularization=1, regularization_scale_factor=0.01,
                            regularization_pose=reg_pose, add_ridge=1, symmetrize_hessian=1, **kw)
    grid = O.Grid(tgt, oprm)
    ref = grid.align(src, guess)
    plain = O.default_params(**kw)  # the evaluator returns the un-regularised sums
    got = pkg.newton_align(pkg.default_params(regularization_scale_factor=0.01, add_ridge=1, **kw), len(src),
                           guess, oracle_evaluator(pkg, O, grid, src, plain), regularization_pose=reg_pose)
    assert got["iterations"] == ref["iterations"]
    np.testing.assert_allclose(got["pose"], ref["pose"], atol=1e-9)
    np.testing.assert_allclose(got["hessian"], ref["hessian"], rtol=1e-10)


def test_newton_driver_propagates_evaluator_failure(pkg, S):
    src, tgt, gt, guess = S.two_planes(seed=1, max_points=500)
    def boom(pose, T, need_h):
        raise RuntimeError("evaluator failed")
    with pytest.raises(pkg.NdtError):
        pkg.newton_align(pkg.default_params(), len(src), guess, boom)


def test_pack_unpack_roundtrip(pkg):
    rng = np.random.default_rng(0)
    g = rng.normal(size=6)
    H = rng.normal(size=(6, 6))
    H = H + H.T
    w = pkg.pack_eval(1.5, g, H, 2.5, 7, 11)
    u = pkg.unpack_eval(w)
    assert u["score"] == 1.5 and u["n_with_neighbors"] == 7 and u["n_pairs"] == 11
    np.testing.assert_array_equal(u["gradient"], g)
    np.testing.assert_array_equal(u["hessian"], H)


def test_result_covariance_matches_driver_recipe(pkg, O):
    """-(H + 1e-6 I)^-1 and the GTSAM block order (ref: run/pipeline.cpp:594-603,
    src/registercallback.cpp:170-186) -- host-only entry point, runs without a GPU."""
    rng = np.random.default_rng(3)
    for scale in (1.0, 1e4, 1e-3):
        A = rng.normal(size=(6, 6))
        H = -(A @ A.T + 0.1 * np.eye(6)) * scale        # Hessian of a maximised score: negative definite
        H[:3, 3:] *= 3.0                                 # make the cross blocks visibly asymmetric in size
        H[3:, :3] = H[:3, 3:].T
        for order in (True, False):
            got = pkg.result_covariance(H, 1e-6, order)
            want = O.covariance_for_gtsam(H, 1e-6, order)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    # the cross blocks are copied, not transposed (reference behaviour)
    H = -np.diag([1.0, 2, 3, 4, 5, 6]); H[0, 4] = H[4, 0] = 0.5
    c = pkg.result_covariance(H, 0.0, False); g = pkg.result_covariance(H, 0.0, True)
    assert np.array_equal(g[:3, 3:], c[:3, 3:]) and np.array_equal(g[:3, :3], c[3:, 3:]) and np.array_equal(g[3:, 3:], c[:3, :3])
    with pytest.raises(pkg.NdtError):
        pkg.result_covariance(np.zeros((6, 6)), 0.0)


def test_presets_and_new_parameter_validation(pkg):
    """ndt_params_preset: the two engines' switch sets; wait_mode / search-method checks."""
    d = pkg.default_params()
    assert (d.cov_mode, d.hessian_mode, d.add_ridge, d.wait_mode) == (pkg.COV_SVN, pkg.HESSIAN_FULL, 0, pkg.WAIT_SPIN)
    p = pkg.default_params(preset=pkg.PRESET_PCLOMP_RECALLED)
    assert (p.cov_mode, p.hessian_mode, p.add_ridge, p.use_line_search) == (pkg.COV_PCL_RECALLED, pkg.HESSIAN_FULL, 0, 1)
    s = pkg.default_params(preset=pkg.PRESET_SVN)
    assert (s.cov_mode, s.hessian_mode, s.add_ridge) == (pkg.COV_SVN, pkg.HESSIAN_GAUSS_NEWTON, 1)
    with pytest.raises(pkg.NdtError):
        pkg.default_params(preset=7)
    L = pkg.lib()
    h = C.c_void_p()
    assert L.ndt_create(C.byref(pkg.default_params(wait_mode=5)), C.byref(h)) == -1
    assert L.ndt_create(C.byref(pkg.default_params(search_method=9)), C.byref(h)) == -1
    # DIRECT26 is a known method now: without a GPU the failure is the missing device, not the argument
    rc = L.ndt_create(C.byref(pkg.default_params(search_method=pkg.DIRECT26)), C.byref(h))
    assert rc in (0, -2)
    if rc == 0:
        L.ndt_destroy(h)


def test_comm_info_names_the_collective_library(pkg):
    v, path = pkg.comm_info()
    assert v > 20000 and "rccl" in path


def test_xy_covariance_laplace_and_search_poses(pkg):
    """tier4 ndt_omp's estimate_covariance helpers [RECALLED], host-only parts:
    Laplace covariance = -(H_xy)^-1; search poses = offsets rotated onto the principal axis of
    the smaller eigenvalue of that covariance, added to the result's translation."""
    rng = np.random.default_rng(4)
    A = rng.normal(size=(6, 6))
    H = -(A @ A.T + 6 * np.eye(6))
    cov = pkg.xy_covariance_laplace(H)
    np.testing.assert_allclose(cov, -np.linalg.inv(H[:2, :2]), rtol=1e-12)
    with pytest.raises(pkg.NdtError):
        pkg.xy_covariance_laplace(np.zeros((6, 6)))
    r = pkg.Result()
    T = np.eye(4); T[:3, 3] = [10.0, -4.0, 1.5]
    r.final_transformation[:] = list(np.ascontiguousarray(T.T, dtype=np.float32).ravel())
    r.hessian[:] = list(H.ravel())
    ox, oy = np.array([0.0, 0.0, 0.3, -0.3, 1.0]), np.array([0.4, -0.4, 0.0, 0.0, 1.0])
    out = np.zeros((5, 16), np.float32)
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    assert pkg.lib().ndt_propose_poses_to_search(C.byref(r), ox.ctypes.data_as(dp), oy.ctypes.data_as(dp), 5,
                                                 out.ctypes.data_as(fp)) == 0
    w, V = np.linalg.eigh(cov)
    th = np.arctan2(V[1, 0], V[0, 0])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    for i in range(5):
        P = out[i].reshape(4, 4).T
        want = T[:2, 3] + R @ np.array([ox[i], oy[i]])
        # the eigenvector's sign is arbitrary: the offset may come out mirrored through the centre
        mirrored = T[:2, 3] - R @ np.array([ox[i], oy[i]])
        assert np.allclose(P[:2, 3], want, atol=1e-5) or np.allclose(P[:2, 3], mirrored, atol=1e-5)
        assert np.allclose(P[:3, :3], np.eye(3)) and P[2, 3] == np.float32(1.5)


def test_quoted_hbm_traffic_matches_the_newest_pmc_summary():
    """bench.py quotes profiles/traffic_k_derivatives.json as the kernel's HBM traffic (a static figure:
    the PMC pass is a separate rocprofv3 run).  It must equal FETCH_SIZE x 2 + WRITE_SIZE of the newest
    committed profiles/rNN_pmc_summary.txt (scripts/traffic_from_pmc.py regenerates it)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("traffic_from_pmc", os.path.join(ROOT, "scripts", "traffic_from_pmc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.traffic(mod.newest_summary())
    have = json.load(open(os.path.join(ROOT, "profiles", "traffic_k_derivatives.json")))
    assert have["kernel"] == want["kernel"]
    assert have["hbm_bytes_per_launch"] == pytest.approx(want["hbm_bytes_per_launch"], rel=2e-3)
    assert have["FETCH_SIZE_KB"] == pytest.approx(want["FETCH_SIZE_KB"], rel=2e-3)
    assert have["WRITE_SIZE_KB"] == pytest.approx(want["WRITE_SIZE_KB"], rel=2e-3)
    assert os.path.basename(mod.newest_summary()) in have["source"]


def test_angle_tables_are_the_derivatives_of_the_rotation(pkg):
    """ndt_angle_tables (host arithmetic, no device): the eight rows of j_ang are d(R x)/d(roll, pitch, yaw) and the fifteen
    rows of h_ang the second derivatives, in the reference's row order (svn_ndt_impl.hpp:270-331) -- checked against central
    differences of R = Rx(roll) Ry(pitch) Rz(yaw) (ref: the pose-to-matrix convention of :761); ndt_gauss_constants against
    the closed form of :90-130."""
    L = pkg.lib()
    L.ndt_angle_tables.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ndt_gauss_constants.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def rot(a):
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        return rx @ ry @ rz

    ang = np.array([0.31, -0.22, 0.47])
    pose = (C.c_double * 6)(1.0, 2.0, 3.0, *ang)
    j, h = (C.c_float * 24)(), (C.c_float * 45)()
    assert L.ndt_angle_tables(pose, j, h) == 0
    j, h = np.array(j, dtype=np.float64).reshape(8, 3), np.array(h, dtype=np.float64).reshape(15, 3)
    e = 1e-5

    def d1(k):
        dp = np.zeros(3); dp[k] = e
        return (rot(ang + dp) - rot(ang - dp)) / (2 * e)

    def d2(a, b):
        da, db = np.zeros(3), np.zeros(3); da[a] = e; db[b] = e
        return (rot(ang + da + db) - rot(ang + da - db) - rot(ang - da + db) + rot(ang - da - db)) / (4 * e * e)

    # rows of j_ang: (component of x', angle): (y, roll) (z, roll) (x, pitch) (y, pitch) (z, pitch) (x, yaw) (y, yaw) (z, yaw)
    for row, (comp, k) in enumerate([(1, 0), (2, 0), (0, 1), (1, 1), (2, 1), (0, 2), (1, 2), (2, 2)]):
        np.testing.assert_allclose(j[row], d1(k)[comp], atol=2e-7)
    assert np.abs(d1(0)[0]).max() < 1e-9   # x' does not depend on roll: no row for it
    hrows = [(1, 0, 0), (2, 0, 0), (1, 0, 1), (2, 0, 1), (1, 0, 2), (2, 0, 2), (0, 1, 1), (1, 1, 1), (2, 1, 1),
             (0, 1, 2), (1, 1, 2), (2, 1, 2), (0, 2, 2), (1, 2, 2), (2, 2, 2)]
    for row, (comp, a, b) in enumerate(hrows):
        want = d2(a, b)[comp]
        if row == 6:   # the reference's table has +sin(pitch) where d2 x'/dpitch^2 has -sin(pitch) (kept: svn_ndt_impl.hpp:312)
            want = want * np.array([1.0, 1.0, -1.0])
        np.testing.assert_allclose(h[row], want, atol=2e-4)
    d1c, d2c = C.c_double(), C.c_double()
    assert L.ndt_gauss_constants(1.0, 0.55, C.byref(d1c), C.byref(d2c)) == 0
    c1, c2 = 10.0 * (1 - 0.55), 0.55
    d3 = -np.log(c2)
    a = -np.log(c1 + c2) - d3
    b = -2.0 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / a)
    assert abs(d1c.value - a) < 1e-15 and abs(d2c.value - b) < 1e-15
    assert L.ndt_gauss_constants(0.0, 0.55, C.byref(d1c), C.byref(d2c)) != 0


def test_two_launch_build_partition_plan(pkg):
    """Host logic of the two-launch target build (round 5: per-tile partition): the tile is the smallest of 2048 / 4096 /
    8192 points that keeps the launch at <= 256 tiles -- the column table has 256 x 256 words and k_bucket_leaves scans a
    column with 256 threads -- for every cloud the build accepts (<= 1 310 720 points); ndt_tuning::bucket_tile forces a
    size where it fits.  No device needed."""
    L = pkg.lib()
    L.ndt_debug_bucket_plan.argtypes = [C.c_size_t, C.POINTER(C.c_longlong)]

    def plan(n):
        out = (C.c_longlong * 4)()
        assert L.ndt_debug_bucket_plan(n, out) == 0
        return tuple(out)

    assert plan(0)[0] == 0 and plan(1310721)[0] == 0
    for n, tile in ((1, 2048), (5000, 2048), (131072, 2048), (524288, 2048), (524289, 4096), (1000000, 4096),
                    (1048576, 4096), (1048577, 8192), (1310720, 8192)):
        fits, t, tiles, words = plan(n)
        assert fits == 1 and t == tile and tiles == (n + tile - 1) // tile and tiles <= 256 and words == 256 * 256, (n, plan(n))
    import random
    rnd = random.Random(7)
    for _ in range(2000):
        n = rnd.randint(1, 1310720)
        fits, t, tiles, _w = plan(n)
        assert fits == 1 and t in (2048, 4096, 8192) and tiles == (n + t - 1) // t and tiles <= 256
        assert t == 2048 or (n + t // 2 - 1) // (t // 2) > 256      # no smaller tile would have fitted
    before = pkg.get_tuning()
    try:
        for forced, n, want in ((8192, 131072, 8192), (1024, 131072, 1024), (1024, 1000000, 4096), (4096, 50000, 4096)):
            pkg.set_tuning(bucket_tile=forced)
            assert plan(n)[1] == want, (forced, n, plan(n))
    finally:
        pkg.set_tuning(**before)


def test_derivative_launch_shapes(pkg):
    """Host logic behind k_derivatives' launches (the partition of the scan decides the last bits of the sums, so it is
    pinned here): one source point per thread; single-pose scans of 131 k - 262 k points take one block per compute unit
    (one unit left to the summing side); FOUR summing blocks stand in front of the point blocks where the machine has
    the units (round 5), one where only one is spare, none where the point blocks fill it (the 128 x 1024 scan: rows
    0 .. 3's blocks share the final sum); batched launches keep one per pose.  No device needed."""
    L = pkg.lib()
    L.ndt_debug_launch_shape.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]

    def shape(n, K=1, cus=256):
        out = (C.c_int * 4)()
        assert L.ndt_debug_launch_shape(n, K, cus, out) == 0
        return tuple(out)

    assert shape(200000) == (832, 241, 4, 245)            # C3
    assert shape(131072) == (512, 256, 0, 256)            # C2 / C5: the point blocks fill the machine
    assert shape(131072, K=20) == (512, 256, 1, 257)      # SVN Stage 1: one summing block per pose
    assert shape(130000) == (512, 254, 1, 255)            # one unit spare: one summing block
    assert shape(25000) == (256, 98, 4, 102)              # a rank's share of the scan
    assert shape(1000) == (256, 4, 4, 8)
    assert shape(1000000) == (512, 1954, 1, 1955)          # several residency rounds: one block adds the rows directly (<= 2048 of them)
    assert shape(1200000) == (512, 2344, 0, 2344)          # ... beyond that the two-level sum, no dedicated block
    for n in (1, 63, 64, 65, 5000, 99999, 131071, 131073, 200001, 262144, 262145, 400000):
        t, pb, ns, grid = shape(n)
        assert t % 64 == 0 and 64 <= t <= 1024 and pb == max(1, -(-n // t)) and grid == pb + ns and ns in (0, 1, 4)
        if ns == 4:
            assert pb + 4 <= 256
    before = pkg.get_tuning()
    try:
        pkg.set_tuning(deriv_summer_split=0)
        assert shape(200000) == (832, 241, 1, 242) and shape(131072) == (512, 256, 0, 256)
        pkg.set_tuning(deriv_summer_split=8)
        assert shape(200000) == (832, 241, 8, 249)
    finally:
        pkg.set_tuning(**before)
