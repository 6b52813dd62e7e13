"""GPU tests of the round-2 surface: DIRECT26, the scoring-only entry point, wait modes, the
evaluation memo of the line search, parameter changes on a consumed device target, and the
RCCL reducer on a one-rank communicator."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEAMS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam-sam_amd",
                         "libndt_hip_seams.so")

KW = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)


def _ndt(pkg, **kw):
    base = dict(KW)
    base.update(kw)
    return pkg.NormalDistributionsTransform(device_id=0, **base)


def test_direct26_search_mode(pkg, O, S):
    """pclomp's DIRECT26 [RECALLED]: every valid voxel of the 3x3x3 block around the point's cell.
    Neighbour counts exact, score 1e-9, g / H 1e-6 against the oracle's statement of it; a
    superset of DIRECT7's pairs; align ends within the reference tolerance of ground truth."""
    for cfg in (S.config_c1(), S.config_c2()):
        grid = O.Grid(cfg["target"], O.default_params(search_method=O.DIRECT26, num_threads=8, **KW))
        ndt = _ndt(pkg, search_method=pkg.DIRECT26)
        ndt.setInputTarget(cfg["target"])
        ndt.setInputSource(cfg["source"])
        p0 = O.matrix_to_pose(cfg["guess"])
        poses = np.stack([p0, O.matrix_to_pose(cfg["gt"]), O.matrix_to_pose(S.pose_matrix(400, 0, 0, 0, 0, 0))])
        prm64 = O.default_params(search_method=O.DIRECT26, num_threads=8, pair_mode=2, **KW)
        for p, e in zip(poses, ndt.evalDerivatives(poses)):
            d = grid.derivatives(cfg["source"], p)
            assert e["n_pairs"] == d["n_pairs"] and e["n_with_neighbors"] == d["n_with_neighbors"]
            assert e["score"] == pytest.approx(d["score"], rel=1e-9, abs=1e-9)
            x = grid.derivatives(cfg["source"], p, params=prm64)
            assert np.linalg.norm(e["gradient"] - x["gradient"]) <= 1e-9 * np.linalg.norm(x["gradient"]) + 1e-9
            assert np.linalg.norm(e["hessian"] - x["hessian"]) <= 1e-9 * np.linalg.norm(x["hessian"]) + 1e-9
        n26 = ndt.evalDerivatives(p0)[0]["n_pairs"]
        ndt.setNeighborhoodSearchMethod(pkg.DIRECT7)
        assert ndt.evalDerivatives(p0)[0]["n_pairs"] <= n26
        ndt.setNeighborhoodSearchMethod(pkg.DIRECT26)
        T = ndt.align(cfg["guess"])
        ref = grid.align(cfg["source"], cfg["guess"])
        dt, dr = S.pose_error(T, ref["T"])
        assert dt < 1e-3 and dr < 1e-4, (dt, dr)
        assert S.pose_error(T, cfg["gt"])[0] < 0.05
        ndt.close()


def test_score_transform_is_the_evaluations_score(pkg, O, S):
    """ndt_score_transform (pclomp calculateTransformationProbability / NVTL): one score-only
    launch, the same score / NVTL / counts as a full evaluation at that transform, bit for bit."""
    cfg = S.config_c2()
    ndt = _ndt(pkg)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    for T in (cfg["guess"], cfg["gt"], np.eye(4)):
        p = O.matrix_to_pose(T)
        e = ndt.evalDerivatives(p, transforms=[T])[0]
        n0 = ndt.getTiming()["n_eval_launches"]
        sc = ndt.scoreTransform(T)
        assert ndt.getTiming()["n_eval_launches"] == n0 + 1
        assert sc["score"] == e["score"] and sc["n_pairs"] == e["n_pairs"]
        assert sc["n_points_with_neighbors"] == e["n_with_neighbors"]
        assert sc["transform_probability"] == e["score"] / len(cfg["source"])
        assert sc["nvtl"] == e["nvtl_sum"] / e["n_with_neighbors"]
    # against the oracle, and through pclomp's method names on a pre-transformed cloud
    grid = O.Grid(cfg["target"], O.default_params(num_threads=8, **KW))
    d = grid.derivatives(cfg["source"], O.matrix_to_pose(cfg["gt"]), T=cfg["gt"])
    moved = ndt.transformSource(cfg["gt"])
    tp = ndt.calculateTransformationProbability(moved)
    nv = ndt.calculateNearestVoxelTransformationLikelihood(moved)
    assert tp == pytest.approx(d["score"] / len(moved), rel=1e-6)
    assert nv == pytest.approx(d["nvtl_sum"] / d["n_with_neighbors"], rel=1e-6)
    # after align(): the result's TP / NVTL are those of the final transformation
    ndt.setInputSource(cfg["source"])
    T = ndt.align(cfg["guess"])
    r = ndt.getResult()
    sc = ndt.scoreTransform(T)
    assert sc["transform_probability"] == pytest.approx(r["transform_probability"], rel=1e-12)
    assert sc["nvtl"] == pytest.approx(r["nvtl"], rel=1e-12)


def test_wait_modes_agree_bit_for_bit(pkg, S):
    cfg = S.config_c2()
    out = []
    for mode in (pkg.WAIT_SPIN, pkg.WAIT_BLOCK):
        ndt = _ndt(pkg, wait_mode=mode)
        ndt.setInputTarget(cfg["target"])
        ndt.setInputSource(cfg["source"])
        T = ndt.align(cfg["guess"])
        r = ndt.getResult()
        out.append((T, r["hessian"], r["iterations"], r["n_evaluations"], r["score"]))
        ndt.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2:] == out[1][2:]


def test_line_search_memo_changes_no_number(pkg, O, S):
    """A More-Thuente step clamped to its lower bound is re-tried up to 10 times at the same pose
    (and the reference re-evaluates at the accepted step for the Hessian): the engine answers
    those from the last result.  Driving the same host loop with the engine as an EXTERNAL
    evaluator (no memo: ndt_newton_align) must give the identical result with more launches."""
    cfg = S.config_c2()
    ndt = _ndt(pkg)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    T = ndt.align(cfg["guess"])
    r = ndt.getResult()
    calls = []

    def ev(pose, Tm, need_h):
        e = ndt.evalDerivatives(pose, transforms=[Tm], compute_hessian=True)[0]
        calls.append(pose.copy())
        return pkg.pack_eval(e["score"], e["gradient"], e["hessian"], e["nvtl_sum"], e["n_with_neighbors"], e["n_pairs"])

    r2 = pkg.newton_align(pkg.default_params(**KW), len(cfg["source"]), cfg["guess"], ev)
    assert np.array_equal(r2["T"], T) and r2["iterations"] == r["iterations"]
    assert np.array_equal(r2["hessian"], r["hessian"]) and r2["score"] == r["score"]
    distinct = sum(1 for i, p in enumerate(calls) if i == 0 or not np.array_equal(p, calls[i - 1]))
    assert r["n_evaluations"] == distinct
    assert r["n_evaluations"] + r["n_evaluations_reused"] >= distinct
    assert len(calls) > r["n_evaluations"]   # C2 ends with a clamped step: launches were saved


def test_grid_parameter_change_on_a_consumed_device_target(pkg, S, hipmem):
    """ndt_set_target_device keeps no copy of the target; a later change of resolution (or of any
    other grid-affecting parameter) cannot re-voxelise it and must not evaluate the old grid with
    the new constants: the next align reports NDT_ERR_NO_TARGET (pclomp's setResolution
    re-voxelises; with a host target, ndt_set_target, this engine does too)."""
    cfg = S.config_c1()
    n = len(cfg["target"])
    t = [hipmem.upload(cfg["target"][:, a]) for a in range(3)]
    ndt = _ndt(pkg)
    ndt.setInputTargetDevice(t[0], t[1], t[2], n)
    ndt.setInputSource(cfg["source"])
    ndt.align(cfg["guess"])
    assert ndt.hasConverged()
    ndt.setStepSize(0.05)          # not a grid parameter: still fine
    ndt.align(cfg["guess"])
    ndt.setResolution(2.0)
    with pytest.raises(pkg.NdtError) as ei:
        ndt.align(cfg["guess"])
    assert ei.value.code == -4
    with pytest.raises(pkg.NdtError):
        ndt.getGridInfo()
    ndt.setInputTargetDevice(t[0], t[1], t[2], n)
    ndt.align(cfg["guess"])
    assert ndt.hasConverged() and ndt.getGridInfo()["leaf_size"] == 2.0
    # device-resident source arrays agree with the host hand-off
    s = [hipmem.upload(cfg["source"][:, a]) for a in range(3)]
    T_host = ndt.align(cfg["guess"])
    ndt.setInputSourceDevice(s[0], s[1], s[2], len(cfg["source"]))
    assert np.array_equal(ndt.align(cfg["guess"]), T_host)
    # ... and so does a view of them (no copy: setInputSource's shared_ptr contract)
    ndt.setInputSource(cfg["source"][:100])
    ndt.setInputSourceDeviceView(s[0], s[1], s[2], len(cfg["source"]))
    assert np.array_equal(ndt.align(cfg["guess"]), T_host)
    assert np.array_equal(ndt.align(cfg["guess"]), T_host)
    ndt.setInputSource(cfg["source"])          # replacing a view by an owned copy
    assert np.array_equal(ndt.align(cfg["guess"]), T_host)
    # a host target is kept and re-voxelised
    ndt2 = _ndt(pkg)
    ndt2.setInputTarget(cfg["target"])
    n1 = ndt2.getGridInfo()["n_leaves"]
    ndt2.setResolution(2.0)
    assert ndt2.getGridInfo()["n_leaves"] < n1


def test_rccl_reducer_on_a_one_rank_communicator(pkg, S):
    """The RCCL leg end to end on one GPU: ncclCommInitRank(nranks = 1) + one ncclAllReduce per
    evaluation on the engine's stream, with PyTorch (and its bundled librccl) loaded first, as in
    bench.py.  Must reproduce the reducer-free align bit for bit.  Prints which librccl serves it."""
    import torch  # noqa: F401  (torch/lib/librccl.so is on the process' library list before the reducer runs)
    version, path = pkg.comm_info()
    print("ncclGetVersion = %d from %s" % (version, path))
    assert version > 20000
    cfg = S.config_c2()
    plain = _ndt(pkg)
    plain.setInputTarget(cfg["target"])
    plain.setInputSource(cfg["source"])
    T0 = plain.align(cfg["guess"])
    r0 = plain.getResult()
    ndt = _ndt(pkg)
    ndt.commInitRccl(pkg.comm_unique_id(), 0, 1)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    T1 = ndt.align(cfg["guess"])
    r1 = ndt.getResult()
    assert np.array_equal(T0, T1) and np.array_equal(r0["hessian"], r1["hessian"])
    assert (r0["iterations"], r0["n_evaluations"], r0["score"]) == (r1["iterations"], r1["n_evaluations"], r1["score"])
    # batched evaluations go through the same reducer
    p = r1["pose"]
    a = plain.evalDerivatives(np.stack([p, p + 0.01]))
    b = ndt.evalDerivatives(np.stack([p, p + 0.01]))
    for x, y in zip(a, b):
        assert x["score"] == y["score"] and np.array_equal(x["hessian"], y["hessian"])
    ndt.commDestroy()
    T2 = ndt.align(cfg["guess"])
    assert np.array_equal(T0, T2)


def test_source_ordering_is_a_permutation_with_the_same_result(pkg, O, S):
    """NDT_SOURCE_ORDER_SORT: the source is evaluated in block order of the target grid (sorted
    once per (source, target) under the first transform).  Counts identical, score / g / H equal
    to the unsorted evaluation up to f64 association (1e-12), align lands on the same optimum;
    AUTO sorts on the wide map (26 MB table) and not on C3 (1.7 MB); the output cloud of
    transformSource keeps the caller's order."""
    for cfg, auto_sorts in ((S.config_c3(), False), (S.config_c3_wide(), True)):
        kw = dict(resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
        res = {}
        for mode in (pkg.SOURCE_ORDER_KEEP, pkg.SOURCE_ORDER_SORT, pkg.SOURCE_ORDER_AUTO):
            ndt = pkg.NormalDistributionsTransform(device_id=0, source_order=mode, **kw)
            ndt.setInputTarget(cfg["target"])
            ndt.setInputSource(cfg["source"])
            p = O.matrix_to_pose(cfg["guess"])
            e = ndt.evalDerivatives(p)[0]
            T = ndt.align(cfg["guess"])
            r = ndt.getResult()
            moved = ndt.transformSource(cfg["gt"])
            res[mode] = (e, T, r, moved)
            ndt.close()
        e0, T0, r0, m0 = res[pkg.SOURCE_ORDER_KEEP]
        e1, T1, r1, m1 = res[pkg.SOURCE_ORDER_SORT]
        assert e0["n_pairs"] == e1["n_pairs"] and e0["n_with_neighbors"] == e1["n_with_neighbors"]
        assert e1["score"] == pytest.approx(e0["score"], rel=1e-12)
        assert np.linalg.norm(e1["gradient"] - e0["gradient"]) <= 1e-11 * np.linalg.norm(e0["gradient"])
        assert np.linalg.norm(e1["hessian"] - e0["hessian"]) <= 1e-11 * np.linalg.norm(e0["hessian"])
        dt, dr = S.pose_error(T0, T1)
        assert dt < 1e-6 and dr < 1e-7 and r0["iterations"] == r1["iterations"]
        assert np.array_equal(m0, m1)                      # caller's order
        ea, Ta, ra, _ = res[pkg.SOURCE_ORDER_AUTO]
        same_as = e1 if auto_sorts else e0                 # bit-identical to the variant AUTO chose
        assert ea["score"] == same_as["score"] and np.array_equal(ea["hessian"], same_as["hessian"])


def test_prelaunched_evaluations_change_no_number(pkg, S):
    """NDT_PRELAUNCH_AUTO: inside align() the kernel of the next evaluation is already on the stream
    and waits on the device for its pose.  Same numbers as ordinary launches, bit for bit; every
    evaluation but the first of an align is served that way, one enqueued kernel per align is told
    to leave; other entry points in between are unaffected."""
    cfg = S.config_c2()
    res = {}
    for mode in (pkg.PRELAUNCH_OFF, pkg.PRELAUNCH_AUTO):
        ndt = _ndt(pkg, prelaunch=mode)
        ndt.setInputTarget(cfg["target"])
        ndt.setInputSource(cfg["source"])
        out = []
        for k in range(3):
            T = ndt.align(cfg["guess"])
            r = ndt.getResult()
            out.append((T, r["hessian"], r["iterations"], r["n_evaluations"], r["score"]))
            if k == 0:   # other launches between two aligns
                sc = ndt.scoreTransform(T)
                e = ndt.evalDerivatives(r["pose"])[0]
                assert sc["score"] == e["score"]
                ndt.setInputTarget(cfg["target"])
        res[mode] = (out, ndt.prelaunchCounters())
        ndt.close()
    off, (used0, quit0, to0) = res[pkg.PRELAUNCH_OFF]
    on, (used1, quit1, to1) = res[pkg.PRELAUNCH_AUTO]
    assert (used0, quit0, to0) == (0, 0, 0)
    for a, b in zip(off, on):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    n_ev = sum(a[3] for a in on)
    assert to1 == 0
    if used1 == 0:
        pytest.skip("this device exposes no BAR-mapped memory: pre-launch stays off")
    assert used1 == n_ev - 3 and quit1 == 3


def test_split_summing_blocks_change_no_bit(pkg, S):
    """Round 5: where compute units are spare a single-pose launch has FOUR summing blocks in front of its point blocks,
    each polling one 128-byte line (eight words) of every row and publishing those words itself (ndt_tuning::
    deriv_summer_split; eight blocks of four words as an A/B).  Columns, order of additions, pairing and tree of a word
    are the one summing block's: the evaluation, the aligned transform and the Hessian are the same BITS with one, four
    or eight of them, through ordinary and pre-launched launches."""
    cfg = S.config_c3()
    src = cfg["source"][:150000]            # 181 point blocks of 832 threads: 4 or 8 more fit the 256 compute units
    before = pkg.get_tuning()
    res = {}
    try:
        for split in (0, 1, 8):
            pkg.set_tuning(deriv_summer_split=split)
            ndt = _ndt(pkg, resolution=0.5)
            ndt.setInputTarget(cfg["target"])
            ndt.setInputSource(src)
            T = ndt.align(cfg["guess"])
            r = ndt.getResult()
            e = ndt.evalDerivatives(r["pose"])[0]
            res[split] = (T.copy(), r["hessian"].copy(), r["iterations"], r["n_evaluations"], r["score"], e["score"], e["gradient"].copy(),
                          e["hessian"].copy(), ndt.prelaunchCounters()[2])
            ndt.close()
    finally:
        pkg.set_tuning(**before)
    for split in (1, 8):
        a, b = res[0], res[split]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:6] == b[2:6], split
        assert np.array_equal(a[6], b[6]) and np.array_equal(a[7], b[7]), split
        assert b[8] == 0


def test_prelaunched_kernel_that_times_out_falls_back(pkg, S):
    """A pre-launched kernel that waited 20 ms for its pose gives up; the host sees it (word 31 of
    the result = 2), stops pre-launching on that handle and evaluates the pose through an ordinary
    launch: same result, no hang.  (Own process: the delay seam is read once from the environment.)"""
    import subprocess, sys, os
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c2()
kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ref = pkg.NormalDistributionsTransform(device_id=0, prelaunch=pkg.PRELAUNCH_OFF, **kw)
ref.setInputTarget(cfg["target"]); ref.setInputSource(cfg["source"])
T0 = ref.align(cfg["guess"])
ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
T1 = ndt.align(cfg["guess"])
used, quit, timeouts = ndt.prelaunchCounters()
print("counters", used, quit, timeouts, "equal", bool(np.array_equal(T0, T1)))
T2 = ndt.align(cfg["guess"])
print("again equal", bool(np.array_equal(T0, T2)))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the fault-injection seams are compiled into libndt_hip_seams.so only (make VARIANT=seams), never
    # into the production library
    env = dict(os.environ, NDT_DEBUG_PUBLISH_DELAY_MS="60", NDT_HIP_LIB=SEAMS_LIB)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("counters")][0].split()
    used, quit, timeouts = int(line[1]), int(line[2]), int(line[3])
    assert "equal True" in p.stdout and "again equal True" in p.stdout
    if used == 0:
        pytest.skip("no BAR-mapped memory on this device")
    assert timeouts == 1


def test_lost_row_is_re_evaluated_through_the_ticketed_sum(pkg, S):
    """An ordinary launch whose summing block gives up waiting for a partial row (seam: one block withholds its row;
    in the field: a device shared with other processes, gpurun_out/r03/bench_4on1.err) is not an error yet: the pose
    is evaluated once more, stream-synchronised, with the final sum made by the block that draws the last ticket.
    Same sums bit for bit (same rows, same order).  (Own process: the seam is read once from the environment.)"""
    import subprocess, sys, os
    code = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c2()
kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
p = np.array([0.4, 0.05, 0.0, 0.0, 0.0, 0.03])
e0 = ndt.evalDerivatives(p)[0]            # launch 0 (batched entry point: not the seam's)
T0 = ndt.align(cfg["guess"])              # launch 1 = the align's first evaluation: its block 2 withholds its row
r0 = ndt.getResult()
print("retries", ndt.lostRowRetries(), "converged", ndt.hasConverged())
T1 = ndt.align(cfg["guess"])
r1 = ndt.getResult()
print("equal", bool(np.array_equal(T0, T1)), r0["iterations"] == r1["iterations"], r0["score"] == r1["score"],
      bool(np.array_equal(r0["hessian"], r1["hessian"])))
print("retries after", ndt.lostRowRetries())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NDT_DEBUG_MUTE_ROW_AT="1", NDT_HIP_LIB=SEAMS_LIB)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "retries 1 converged True" in p.stdout, p.stdout
    assert "equal True True True True" in p.stdout and "retries after 1" in p.stdout, p.stdout


def test_pose_that_lands_at_the_deadline_with_two_streams(pkg, S):
    """The pose of a pre-launched kernel arrives right at its 20 ms deadline: some blocks have left, others compute
    (every block times out on its own clock).  Whatever mixture results, the host drains both streams before the
    re-launch reuses the rows: same transform as without pre-launching, and the next align is clean."""
    import subprocess, sys, os
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
kw = dict(resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ref = pkg.NormalDistributionsTransform(device_id=0, prelaunch=pkg.PRELAUNCH_OFF, **kw)
ref.setInputTarget(cfg["target"]); ref.setInputSource(cfg["source"])
T0 = ref.align(cfg["guess"])
ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
ok = True
for k in range(4):
    T = ndt.align(cfg["guess"])
    ok = ok and bool(np.array_equal(T0, T))
print("equal", ok, "counters", ndt.prelaunchCounters(), "overlapped", ndt.prelaunchOverlapped())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NDT_DEBUG_PUBLISH_DELAY_MS="20", NDT_HIP_LIB=SEAMS_LIB)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "equal True" in p.stdout, p.stdout


def test_batched_scoring_and_xy_covariance_estimators(pkg, O, S):
    """ndt_score_transforms: K poses in one launch == K single scoring launches, bit for bit.
    The 2-D covariance estimators of tier4 ndt_omp [RECALLED] on top of it: MULTI_NDT (re-align
    from every search pose) and MULTI_NDT_SCORE (NVTL softmax) against a NumPy restatement that
    uses the ORACLE for the re-alignments / scores."""
    cfg = S.config_c2()
    ndt = _ndt(pkg)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    T = ndt.align(cfg["guess"])
    main = ndt.getResult()
    lap = pkg.xy_covariance_laplace(main["hessian"])
    np.testing.assert_allclose(lap, -np.linalg.inv(main["hessian"][:2, :2]), rtol=1e-10)
    assert np.all(np.linalg.eigvalsh(lap) > 0)
    ox = [0.0, 0.0, 0.5, -0.5, 1.0, -1.0]
    oy = [0.5, -0.5, 0.0, 0.0, 0.0, 0.0]
    poses = ndt.proposePosesToSearch(ox, oy)
    n0 = ndt.getTiming()["n_eval_launches"]
    batch = ndt.scoreTransforms(poses)
    assert ndt.getTiming()["n_eval_launches"] == n0 + 1
    for P, b in zip(poses, batch):
        one = ndt.scoreTransform(P)
        assert one["score"] == b["score"] and one["nvtl"] == b["nvtl"] and one["n_pairs"] == b["n_pairs"]
    # MULTI_NDT_SCORE
    temperature = 0.05
    mean_s, cov_s = ndt.estimateXYCovarianceMultiNdtScore(poses, temperature)
    grid = O.Grid(cfg["target"], O.default_params(num_threads=8, **KW))
    pts = [T[:2, 3]] + [P[:2, 3] for P in poses]
    sc = [main["nvtl"]]
    for P in poses:
        d = grid.derivatives(cfg["source"], O.matrix_to_pose(P), T=P, compute_hessian=False)
        sc.append(d["nvtl_sum"] / d["n_with_neighbors"])
    w = np.exp((np.array(sc) - max(sc)) / temperature); w /= w.sum()
    m = sum(wi * p for wi, p in zip(w, pts))
    c = sum(wi * np.outer(p - m, p - m) for wi, p in zip(w, pts))
    np.testing.assert_allclose(mean_s, m, atol=1e-6)
    np.testing.assert_allclose(cov_s, c, rtol=1e-4, atol=1e-9)
    # MULTI_NDT
    mean_m, cov_m = ndt.estimateXYCovarianceMultiNdt(poses)
    pts = [T[:2, 3]]
    for P in poses:
        r = grid.align(cfg["source"], P)
        pts.append(r["T"][:2, 3])
    pts = np.array(pts)
    np.testing.assert_allclose(mean_m, pts.mean(0), atol=1e-3)
    cm = np.cov(pts.T, ddof=1)
    assert np.abs(cov_m - cm).max() < 1e-6 + 0.2 * np.abs(cm).max()   # re-alignments end within 1 mm of the oracle's
    assert np.all(np.linalg.eigvalsh(cov_m) >= -1e-12)


def _leaf_dump(env_extra, cfg_name="config_c2", res=1.0):
    import subprocess, sys, os
    code = r"""
import sys, numpy as np, hashlib
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
pkg.apply_env_tuning()   # (NDT_FUSED_SORT etc. -> ndt_set_tuning: the library itself does not read them)
cfg = getattr(S, %r)()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=%r, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
h = hashlib.sha256()
for rep in range(3):       # first build waits for the geometry, the later ones are optimistic
    ndt.setInputTarget(cfg["target"])
    L = ndt.getLeaves()
    for k in ("cell", "count", "mean", "cov", "icov", "evecs", "evals"):
        h.update(np.ascontiguousarray(L[k]).tobytes())
ndt.setInputSource(cfg["source"]); T = ndt.align(cfg["guess"])
h.update(np.ascontiguousarray(T).tobytes())
print("dump", h.hexdigest(), len(L["cell"]), ndt.buildCounters()[0])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cfg_name, res)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env_extra))
    assert p.returncode == 0, p.stderr[-2000:]
    w = [ln for ln in p.stdout.splitlines() if ln.startswith("dump")][0].split()
    return w[1], int(w[2]), int(w[3])


def test_fused_sort_passes_equal_classic_passes_and_fall_back():
    """The build's one-launch-per-digit sort passes (k_sort_pass) against the classic
    count / scan / scatter passes: every exported leaf field and the aligned transform, bit for bit
    (own processes: the switch is read once from the environment).  With one tile made to withhold
    its counts every block of the fused pass gives up after its time-out, the build is marked
    BG_SPIN, and the engine repeats it with the classic passes: same leaves, and the fallback is
    counted."""
    # (the two-launch build is switched off: since round 5 it waits for no sibling block, so every build of this test
    # takes the sort-based pipeline the seam sits in)
    fused = _leaf_dump({"NDT_FUSED_SORT": "1", "NDT_BUCKET_BUILD": "0"})
    classic = _leaf_dump({"NDT_FUSED_SORT": "0", "NDT_BUCKET_BUILD": "0"})
    assert fused[1] > 1000 and fused[2] == 0 and classic[2] == 0
    assert fused[:2] == classic[:2]
    assert _leaf_dump({})[:2] == classic[:2]      # ... and the two-launch build gives the same leaves, bit for bit
    muted = _leaf_dump({"NDT_FUSED_SORT": "1", "NDT_BUCKET_BUILD": "0", "NDT_DEBUG_FUSED_MUTE_TILE": "0", "NDT_HIP_LIB": SEAMS_LIB})
    # the production library carries no such seam: the variable changes nothing there
    inert = _leaf_dump({"NDT_FUSED_SORT": "1", "NDT_BUCKET_BUILD": "0", "NDT_DEBUG_FUSED_MUTE_TILE": "0"})
    assert inert == fused
    assert muted[:2] == classic[:2]
    assert muted[2] == 3          # every one of the three builds fell back


def test_two_launch_build_is_tile_size_independent():
    """k_bucket_pass partitions the cloud tile by tile (round 5: a tile's window + the column table, no cloud-wide
    offsets); k_bucket_leaves gathers a bucket's segments in tile order.  Whatever the tile size (1024 ... 8192 points:
    128 ... 16 tiles for this scan, the last one ragged for none of them -- the 100 003-point cloud below has one),
    the bucket holds the same points in input order: leaves and the aligned transform are bit-identical to the
    sort-based pipeline's."""
    ref = _leaf_dump({"NDT_BUCKET_BUILD": "0"})
    assert ref[1] > 1000
    for tile in ("1024", "2048", "4096", "8192"):
        got = _leaf_dump({"NDT_BUCKET_TILE": tile})
        assert got[:2] == ref[:2], tile
        assert got[2] == 0


def test_fused_build_launch_tags_wrap_around(pkg, hipmem):
    """The fused build kernels recognise this launch's words by a 16-bit tag; the tag counters wrap
    after 65535 launches (21845 builds for the three sort passes, 65535 for the run search) and the
    tag tables are cleared at that point.  70 000 builds of one small cloud: the leaves never change."""
    rng = np.random.default_rng(5)
    tgt = rng.uniform(-12, 12, (20000, 3)).astype(np.float32)
    t = [hipmem.upload(tgt[:, a]) for a in range(3)]
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=5)
    ndt.setInputTargetDevice(t[0], t[1], t[2], len(tgt))
    ref = ndt.getLeaves()
    assert len(ref["cell"]) > 1000
    for i in range(70000):
        ndt.setInputTargetDevice(t[0], t[1], t[2], len(tgt))
        if i % 7000 == 6999 or i in (21843, 21844, 21845, 21846, 65533, 65534, 65535, 65536):
            L = ndt.getLeaves()
            for k in ("cell", "count", "mean", "icov"):
                assert np.array_equal(L[k], ref[k]), (i, k)
    assert ndt.buildCounters()[0] == 0


def test_prelaunched_evaluations_survive_a_starved_host():
    """A host thread that is frozen for tens of milliseconds now and then (background threads keep the
    BLAS pool busy under the box's CPU quota): waiting kernels give up, their notices must not replace
    the unread result of their predecessor (they did, with one shared result buffer: the host then
    waited 5 s for tags that were gone and reported NDT_ERR_HIP).  Every align returns the reference
    result; give-ups are counted and handled.  tools/mbox_stress.py, own process."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "mbox_stress.py"), "2500"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if "aligns under host contention" in ln][0]
    assert " 0 failures, 0 mismatches" in line, p.stdout[-2000:]


def test_two_engines_on_two_threads(pkg, S):
    """Two engines of one process driven from two threads at once (rebuild + align, each on its own
    stream, mailbox and result buffers): pre-launched kernels of one wait on the device while the
    other builds (fused sort passes) and evaluates.  Every result equals the single-threaded one."""
    import threading
    cfgs = [S.config_c2(), S.config_c1()]
    kws = [dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)] * 2
    refs = []
    for cfg, kw in zip(cfgs, kws):
        ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
        ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
        refs.append((ndt.align(cfg["guess"]).copy(), ndt.getResult()["score"], ndt.getLeaves()["count"].sum()))
    errors = []

    def worker(k):
        try:
            cfg, kw = cfgs[k], kws[k]
            ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
            for i in range(150):
                ndt.setInputTarget(cfg["target"])
                ndt.setInputSource(cfg["source"])
                T = ndt.align(cfg["guess"])
                if not np.array_equal(T, refs[k][0]) or ndt.getResult()["score"] != refs[k][1]:
                    errors.append((k, i, "result"))
                if i % 50 == 0 and ndt.getLeaves()["count"].sum() != refs[k][2]:
                    errors.append((k, i, "leaves"))
        except Exception as e:   # noqa: BLE001 -- reported below
            errors.append((k, "exception", repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=300)
    assert not any(t.is_alive() for t in ts)
    assert errors == []


def test_viewed_source_rewritten_in_place(pkg, S, hipmem):
    """ndt_set_source_device_view keeps the caller's arrays; with source ordering on, the engine caches a
    block-ordered COPY of them.  A caller that re-fills the same device buffers (a reused scan buffer) says
    so with ndt_source_changed (or sets the view again): the next align then sees the new points, exactly
    as an engine that was handed the new scan from scratch."""
    cfg = S.config_c2()
    kw = dict(KW, source_order=pkg.SOURCE_ORDER_SORT)
    src_a = cfg["source"]
    rng = np.random.default_rng(3)
    src_b = (src_a[rng.permutation(len(src_a))] + rng.normal(0, 0.01, src_a.shape)).astype(np.float32)
    ndt = _ndt(pkg, **kw)
    ndt.setInputTarget(cfg["target"])
    ptr = [hipmem.upload(src_a[:, a]) for a in range(3)]
    ndt.setInputSourceDeviceView(ptr[0], ptr[1], ptr[2], len(src_a))
    Ta = ndt.align(cfg["guess"])
    for a in range(3):
        hipmem.write(ptr[a], src_b[:, a])
    ndt.sourceChanged()
    Tb = ndt.align(cfg["guess"])
    fresh = _ndt(pkg, **kw)
    fresh.setInputTarget(cfg["target"])
    fresh.setInputSource(src_b)
    assert np.array_equal(Tb, fresh.align(cfg["guess"]))
    assert not np.array_equal(Ta, Tb)
    # ... and setting the view again does the same
    for a in range(3):
        hipmem.write(ptr[a], src_a[:, a])
    ndt.setInputSourceDeviceView(ptr[0], ptr[1], ptr[2], len(src_a))
    assert np.array_equal(ndt.align(cfg["guess"]), Ta)


def test_packed_voxel_records(pkg, O, S):
    """ndt_set_record_format(NDT_RECORDS_PACKED48): 48-byte records (f64 mean, inverse covariance rounded to f32 --
    the reference's own c_inv4 for the per-pair gradient / Hessian products).  Same pairs (counts exact); score,
    gradient and Hessian within 1e-6 of the f64 records' (measured ~1e-7: the f32 rounding of the matrix) and of the
    oracle; align ends on the same pose within the convergence threshold and within 1 mm of the oracle's, iterations within one; batched ==
    single launches bit for bit; selecting the format before or after the build gives the same bits; the exported
    leaf statistics stay the f64 ones.  DIRECT7 / DIRECT1 only: KDTREE / DIRECT26 ignore the format."""
    for cfg, res in ((S.config_c1(), 1.0), (S.config_c2(), 1.0)):
        kw = dict(KW, resolution=res)
        p0, pg = O.matrix_to_pose(cfg["guess"]), O.matrix_to_pose(cfg["gt"])
        poses = np.stack([p0, pg])
        for method, omethod in ((pkg.DIRECT7, O.DIRECT7), (pkg.DIRECT1, O.DIRECT1), (pkg.KDTREE, O.KDTREE), (pkg.DIRECT26, O.DIRECT26)):
            grid = O.Grid(cfg["target"], O.default_params(search_method=omethod, num_threads=8, **kw))
            a = _ndt(pkg, search_method=method, resolution=res)
            b = _ndt(pkg, search_method=method, resolution=res)
            b.setRecordFormat(pkg.RECORDS_PACKED48)           # before the build: packed behind it
            assert b.getRecordFormat() == pkg.RECORDS_PACKED48 and a.getRecordFormat() == pkg.RECORDS_F64
            for n in (a, b):
                n.setInputTarget(cfg["target"])
                n.setInputSource(cfg["source"])
            la, lb = a.getLeaves(), b.getLeaves()
            for f in ("cell", "count", "mean", "cov", "icov"):
                assert np.array_equal(la[f], lb[f]), f
            ea, eb = a.evalDerivatives(poses), b.evalDerivatives(poses)
            if method in (pkg.KDTREE, pkg.DIRECT26):
                # the 27-cell neighbourhoods keep reading the 80-byte records: selecting the format changes no bit
                for x, y in zip(ea, eb):
                    assert x["score"] == y["score"] and np.array_equal(x["hessian"], y["hessian"])
                a.close(); b.close()
                continue
            for p, x, y in zip(poses, ea, eb):
                assert x["n_pairs"] == y["n_pairs"] and x["n_with_neighbors"] == y["n_with_neighbors"]
                assert y["score"] == pytest.approx(x["score"], rel=1e-6)
                assert y["score"] != x["score"]              # (the packed table is what was read)
                gn, hn = np.linalg.norm(x["gradient"]), np.linalg.norm(x["hessian"])
                assert np.linalg.norm(y["gradient"] - x["gradient"]) <= 1e-6 * gn + 1e-9
                assert np.linalg.norm(y["hessian"] - x["hessian"]) <= 1e-6 * hn + 1e-9
                d = grid.derivatives(cfg["source"], p)
                assert y["n_pairs"] == d["n_pairs"] and y["score"] == pytest.approx(d["score"], rel=1e-6)
                assert np.linalg.norm(y["gradient"] - d["gradient"]) <= 2e-6 * np.linalg.norm(d["gradient"]) + 1e-9
            # batched launch == single launches, bit for bit, on the packed table too
            for p, y in zip(poses, eb):
                one = b.evalDerivatives(p)[0]
                assert one["score"] == y["score"] and np.array_equal(one["hessian"], y["hessian"])
            # format selected AFTER the build: packed on demand, same bits
            a.setRecordFormat(pkg.RECORDS_PACKED48)
            for x, y in zip(a.evalDerivatives(poses), eb):
                assert x["score"] == y["score"] and np.array_equal(x["gradient"], y["gradient"]) and np.array_equal(x["hessian"], y["hessian"])
            a.setRecordFormat(pkg.RECORDS_F64)
            for x, y in zip(a.evalDerivatives(poses), ea):
                assert x["score"] == y["score"] and np.array_equal(x["hessian"], y["hessian"])
            Ta = a.align(cfg["guess"]); ra = a.getResult()
            Tb = b.align(cfg["guess"]); rb = b.getResult()
            dt, dr = S.pose_error(Ta, Tb)
            # (both stop once a step is shorter than trans_epsilon = 1e-4 m: the end poses agree to that, not better)
            assert dt < 2e-4 and dr < 1e-5 and abs(ra["iterations"] - rb["iterations"]) <= 1, (method, dt, dr)
            ref = grid.align(cfg["source"], cfg["guess"])
            dt, dr = S.pose_error(Tb, ref["T"])
            assert dt < 1e-3 and dr < 1e-4, (method, dt, dr)
            # a second target through the same engine: the packed copy follows the rebuild
            b.setInputTarget(cfg["target"][::2])
            a.setInputTarget(cfg["target"][::2])
            x, y = a.evalDerivatives(pg)[0], b.evalDerivatives(pg)[0]
            assert x["n_pairs"] == y["n_pairs"] and y["score"] == pytest.approx(x["score"], rel=1e-6) and y["score"] != x["score"]
            a.close(); b.close()


def test_iteration_history_of_an_align(pkg, S):
    """pclomp::NdtResult's per-iteration arrays [RECALLED: tier4 ndt_omp transformation_array / score arrays]: entry 0 is
    the initial guess, one entry per Newton iteration after it; the last entry is the result."""
    cfg = S.config_c2()
    ndt = _ndt(pkg)
    ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
    T = ndt.align(cfg["guess"])
    r = ndt.getResult()
    Ts, tp, nv = ndt.getIterationHistory()
    assert len(Ts) == r["iterations"] + 1 == len(tp) == len(nv)
    assert np.array_equal(Ts[0], np.asarray(cfg["guess"], np.float32).astype(np.float64)) and np.array_equal(Ts[-1], T)
    assert tp[-1] == r["transform_probability"] and nv[-1] == r["nvtl"]
    # the score climbs from the guess to the optimum, and every entry is the score of the source under ITS transform
    assert tp[-1] > tp[0] and nv[-1] > nv[0]
    for k in (0, len(Ts) // 2, len(Ts) - 1):
        sc = ndt.scoreTransform(Ts[k])
        assert abs(sc["transform_probability"] - tp[k]) <= 1e-9 * abs(tp[k]) and abs(sc["nvtl"] - nv[k]) <= 1e-9 * abs(nv[k])
    # a second align replaces the history
    ndt.setMaximumIterations(2)
    ndt.align(cfg["guess"])
    assert len(ndt.getIterationHistory()[1]) == ndt.getFinalNumIteration() + 1 <= 5   # (the loop stops when iters > max: max + 2 iterations)
