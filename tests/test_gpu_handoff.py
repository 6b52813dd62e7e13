"""The asynchronous hand-off of HOST clouds (ndt_set_target / ndt_set_source; the drivers hold host
pcl::PointCloud<PointXYZI>, ref: run/pipeline.cpp:554-561): the calls return once the caller's memory has been
consumed, copies and build finish behind them, the first call that needs them waits.  Nothing about the RESULTS may
depend on the mode: every number here is compared bit for bit with the blocking hand-off of rounds 1-3."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ndt(pkg, **kw):
    base = dict(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    base.update(kw)
    return pkg.NormalDistributionsTransform(**base)


def _xyzi(a, fill=3.5):
    out = np.full((len(a), 8), fill, np.float32)
    out[:, :3] = a
    return out


def _leaves_equal(a, b):
    for f in ("cell", "count", "mean", "cov", "icov", "evals"):
        assert np.array_equal(a[f], b[f]), f


def test_async_and_sync_handoff_agree_bit_for_bit(pkg, S):
    """Scan after scan (steady-state builds are the deferred ones), three layouts, the caller's arrays
    overwritten the moment the call returns."""
    ca, cb = S.config_c2(), S.config_c1()
    clouds = [(ca["target"], ca["source"], ca["guess"]), (cb["target"], cb["source"], cb["guess"]),
              (ca["target"][::2], ca["source"][::3], ca["guess"])]
    out = {}
    for mode in (pkg.HANDOFF_SYNC, pkg.HANDOFF_ASYNC):
        ndt = _ndt(pkg)
        ndt.setHandoffMode(mode)
        assert ndt.getHandoffMode() == mode
        res = []
        for rep in range(3):
            for tgt, src, guess in clouds:
                for layout in ("xyz", "xyzi", "soa"):
                    if layout == "xyz":
                        t, s = tgt.copy(), src.copy()
                        ndt.setInputTarget(t); t[:] = np.nan
                        ndt.setInputSource(s); s[:] = np.nan
                    elif layout == "xyzi":
                        t, s = _xyzi(tgt), _xyzi(src)
                        ndt.setInputTarget(t); t[:] = np.nan
                        ndt.setInputSource(s); s[:] = np.nan
                    else:
                        t = [np.ascontiguousarray(tgt[:, a]) for a in range(3)]
                        s = [np.ascontiguousarray(src[:, a]) for a in range(3)]
                        ndt.setInputTargetSoA(*t); [v.fill(np.nan) for v in t]
                        ndt.setInputSourceSoA(*s); [v.fill(np.nan) for v in s]
                    T = ndt.align(guess)
                    r = ndt.getResult()
                    res.append((T.copy(), r["iterations"], r["n_evaluations"], r["score"], r["hessian"].copy()))
            if rep == 0:
                res.append(ndt.getLeaves())
        out[mode] = res
        ndt.close()
    for a, b in zip(out[pkg.HANDOFF_SYNC], out[pkg.HANDOFF_ASYNC]):
        if isinstance(a, dict):
            _leaves_equal(a, b)
        else:
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and a[3] == b[3]
            assert np.array_equal(a[4], b[4])


def test_deferred_build_failure_is_reported_by_the_first_consumer(pkg, S):
    """A steady-state build that fails is reported by the first call that needs the grid (and by wait()), with the
    status the blocking hand-off returns from setInputTarget; a later good target clears it."""
    cfg = S.config_c1()
    bad = np.full((5000, 3), np.nan, np.float32)
    ndt = _ndt(pkg)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    ndt.align(cfg["guess"])
    assert ndt.hasConverged()
    # blocking hand-off: the error comes from setInputTarget itself
    ndt.setHandoffMode(pkg.HANDOFF_SYNC)
    with pytest.raises(pkg.NdtError) as ei:
        ndt.setInputTarget(bad)
    assert ei.value.code == -4
    ndt.setInputTarget(cfg["target"])
    ndt.align(cfg["guess"])
    # asynchronous: setInputTarget has nothing to report yet; align / getGridInfo / getLeaves / wait do
    ndt.setHandoffMode(pkg.HANDOFF_ASYNC)
    ndt.setInputTarget(bad)
    ndt.setStepSize(0.1)   # a setter in between neither reports nor loses the failure
    with pytest.raises(pkg.NdtError) as ei:
        ndt.align(cfg["guess"])
    assert ei.value.code == -4 and "finite" in str(ei.value)
    # reported once; from then on the handle is where a failed BLOCKING setInputTarget leaves it: no target for the
    # calls that need one, nothing in flight for wait()
    for call in (ndt.getGridInfo, ndt.getLeaves):
        with pytest.raises(pkg.NdtError) as ei:
            call()
        assert ei.value.code == -4
    ndt.wait()
    ndt.setInputTarget(cfg["target"]); ndt.wait()
    ndt.setInputTarget(bad)       # deferred again; this time wait() is the first to hear of it -- and the only one
    with pytest.raises(pkg.NdtError) as ei:
        ndt.wait()
    assert ei.value.code == -4 and "finite" in str(ei.value)
    ndt.wait()
    with pytest.raises(pkg.NdtError) as ei:
        ndt.align(cfg["guess"])
    assert ei.value.code == -4
    # the build after a failed one starts from a cleared index grid and waits for its geometry like a handle's
    # first build: it completes inside the call, so its failure is reported there
    with pytest.raises(pkg.NdtError) as ei:
        ndt.setInputTarget(bad)
    assert ei.value.code == -4
    ndt.setInputTarget(cfg["target"])
    ndt.wait()
    ndt.setInputTarget(bad)       # steady state again: deferred
    with pytest.raises(pkg.NdtError):
        ndt.wait()
    ndt.setInputTarget(cfg["target"])
    ndt.wait()
    T = ndt.align(cfg["guess"])
    dt, dr = S.pose_error(T, cfg["gt"])
    assert ndt.hasConverged() and dt < 0.05 and dr < 0.035


def test_handoff_overlaps_and_every_consumer_waits(pkg, O, S):
    """Target and source handed over back to back, then every kind of consumer first: each must see the NEW clouds."""
    a, b = S.config_c1(), S.config_c2()
    ndt = _ndt(pkg)
    ndt.setInputTarget(a["target"]); ndt.setInputSource(a["source"]); ndt.align(a["guess"])   # first build: blocking
    consumers = {
        "align": lambda: ndt.align(b["guess"]),
        "evalDerivatives": lambda: ndt.evalDerivatives(O.matrix_to_pose(b["guess"])),
        "scoreTransform": lambda: ndt.scoreTransform(b["guess"]),
        "getGridInfo": ndt.getGridInfo,
        "getLeaves": ndt.getLeaves,
        "setResolution": lambda: ndt.setResolution(1.0),
        "transformSource": lambda: ndt.transformSource(b["guess"]) if hasattr(ndt, "transformSource") else None,
    }
    grid_b = O.Grid(b["target"], O.default_params(resolution=1.0))
    ref = grid_b.derivatives(b["source"], O.matrix_to_pose(b["guess"]))
    for name, first in consumers.items():
        ndt.setInputTarget(a["target"]); ndt.setInputSource(a["source"]); ndt.align(a["guess"])
        ndt.setInputTarget(b["target"])
        ndt.setInputSource(b["source"])
        first()
        gi = ndt.getGridInfo()
        assert gi["n_leaves"] == grid_b.n_leaves, name
        e = ndt.evalDerivatives(O.matrix_to_pose(b["guess"]))[0]
        assert e["n_pairs"] == ref["n_pairs"], name
        np.testing.assert_allclose(e["score"], ref["score"], rtol=1e-9, err_msg=name)
    t = ndt.getHandoffTiming()
    assert t["mode"] == pkg.HANDOFF_ASYNC and t["target"]["n_points"] == len(b["target"])
    assert t["source"]["n_points"] == len(b["source"]) and t["target"]["threads"] >= 1 and t["cpu_budget"] >= 1
    assert t["target"]["bytes_dma"] == 12 * len(b["target"])


def test_handoff_timing_reports_dma_rate(pkg, S):
    cfg = S.config_c2()
    ndt = _ndt(pkg)
    t32, s32 = _xyzi(cfg["target"]), _xyzi(cfg["source"])
    for _ in range(2):
        ndt.setInputTarget(t32); ndt.setInputSource(s32); ndt.align(cfg["guess"])
    ndt.enableKernelTiming(True)
    ndt.setInputTarget(t32); ndt.setInputSource(s32); ndt.align(cfg["guess"])
    ndt.wait()
    t = ndt.getHandoffTiming()
    ndt.enableKernelTiming(False)
    assert t["target"]["bytes_in"] == 32 * len(t32) and t["target"]["ms_dma"] > 0 and t["target"]["dma_gb_per_s"] > 1.0
    assert t["source"]["ms_dma"] > 0 and t["source"]["ms_repack"] > 0


@pytest.mark.parametrize("seed", [11, 12, 13] + list(range(100, 100 + int(os.environ.get("NDT_FUZZ_EXTRA_SEEDS", "0")))))
def test_random_api_sequences_async_equals_sync(pkg, S, hipmem, seed):
    """Differential fuzz of the asynchronous hand-off's bookkeeping: two engines, one blocking, one asynchronous, are fed
    the SAME random sequence of calls (targets and sources through every entry point, consumers of every kind, keyframes,
    downsample, parameter changes in between) -- whatever is pending when a call arrives, every observable must be
    identical, errors included."""
    rng = np.random.default_rng(seed)
    a, b = S.config_c1(), S.config_c2()
    clouds = [a["target"], b["target"][::2], a["target"][::3] + np.float32(0.25), b["target"]]
    sources = [a["source"], b["source"][::4], a["source"][::2]]
    guesses = [a["guess"], b["guess"], a["guess"]]
    bad = np.full((3000, 3), np.nan, np.float32)
    dev = {k: [hipmem.upload(np.ascontiguousarray(c[:, ax])) for ax in range(3)] for k, c in enumerate(clouds)}
    dsrc = {k: [hipmem.upload(np.ascontiguousarray(c[:, ax])) for ax in range(3)] for k, c in enumerate(sources)}
    eng = {}
    for mode in (pkg.HANDOFF_SYNC, pkg.HANDOFF_ASYNC):
        # (the blocking engine is also the plainest one: every evaluation an ordinary launch; the asynchronous one has
        # everything on -- pre-launched kernels, both streams, the first evaluation behind a running build)
        e = _ndt(pkg, prelaunch=pkg.PRELAUNCH_OFF if mode == pkg.HANDOFF_SYNC else pkg.PRELAUNCH_AUTO)
        e.setHandoffMode(mode)
        e.setInputTarget(clouds[0]); e.setInputSource(sources[0]); e.align(guesses[0])
        eng[mode] = e

    def call(e, op, k):
        """One operation -> a comparable observation (or the error's status code)."""
        try:
            if op == "target":
                e.setInputTarget(clouds[k % 4]); return "ok"
            if op == "target_xyzi":
                e.setInputTarget(_xyzi(clouds[k % 4])); return "ok"
            if op == "target_soa":
                e.setInputTargetSoA(*[np.ascontiguousarray(clouds[k % 4][:, ax]) for ax in range(3)]); return "ok"
            if op == "target_dev":
                d = dev[k % 4]; e.setInputTargetDevice(d[0], d[1], d[2], len(clouds[k % 4])); return "ok"
            if op == "target_dev_deferred":   # (enqueued under the asynchronous hand-off; the arrays are never touched here)
                d = dev[k % 4]; e.setInputTargetDeviceDeferred(d[0], d[1], d[2], len(clouds[k % 4])); return "ok"
            if op == "target_bad":
                e.setInputTarget(bad); return "ok"
            if op == "source":
                e.setInputSource(sources[k % 3]); return "ok"
            if op == "source_soa":
                e.setInputSourceSoA(*[np.ascontiguousarray(sources[k % 3][:, ax]) for ax in range(3)]); return "ok"
            if op == "source_dev":
                d = dsrc[k % 3]; e.setInputSourceDevice(d[0], d[1], d[2], len(sources[k % 3])); return "ok"
            if op == "source_view":
                d = dsrc[k % 3]; e.setInputSourceDeviceView(d[0], d[1], d[2], len(sources[k % 3])); return "ok"
            if op == "align":
                T = e.align(guesses[k % 3]); r = e.getResult()
                return (T.tobytes(), r["iterations"], r["n_evaluations"], r["score"], r["hessian"].tobytes())
            if op == "eval":
                ev = e.evalDerivatives(np.array([0.3, 0.05, 0.0, 0.0, 0.01, 0.02 * (k % 5)]))[0]
                return (ev["score"], ev["n_pairs"], ev["hessian"].tobytes())
            if op == "score":
                sc = e.scoreTransform(guesses[k % 3]); return (sc["score"], sc["n_pairs"])
            if op == "grid":
                g = e.getGridInfo(); return (int(g["n_leaves"]), int(g["n_cells"]), int(g["n_target_points"]))
            if op == "leaves":
                L = e.getLeaves(); return (L["cell"].tobytes(), L["count"].tobytes(), L["cov"].tobytes())
            if op == "step":
                e.setStepSize(0.1 if k % 2 else 0.05); return "ok"
            if op == "resolution":
                e.setResolution(1.0 if k % 2 else 1.5); return "ok"
            if op == "search":
                e.setNeighborhoodSearchMethod((pkg.DIRECT7, pkg.DIRECT1, pkg.KDTREE, pkg.DIRECT26)[k % 4]); return "ok"
            if op == "hessian":
                e.setParams(hessian_mode=pkg.HESSIAN_GAUSS_NEWTON if k % 2 else pkg.HESSIAN_FULL); return "ok"
            if op == "maxit":
                e.setMaximumIterations((35, 2, 0, 12)[k % 4]); return "ok"
            if op == "evalbatch":
                P = np.array([[0.3, 0.05, 0.0, 0.0, 0.01, 0.02 * j] for j in range(1 + k % 5)])
                return tuple((ev["score"], ev["n_pairs"], ev["hessian"].tobytes()) for ev in e.evalDerivatives(P))
            if op == "keyframe":
                e.putKeyframe(7, sources[k % 3]); e.setInputSourceFromKeyframe(7); return "ok"
            if op == "downsample":
                return e.voxelDownsample(clouds[k % 4][::5], 0.75).tobytes()
            if op == "wait":
                e.wait(); return "ok"
        except pkg.NdtError as err:
            return ("error", err.code)
        raise AssertionError(op)

    ops = ["target", "target_xyzi", "target_soa", "target_dev", "source", "source_soa", "source_dev", "source_view",
           "align", "align", "eval", "score", "grid", "leaves", "step", "resolution", "keyframe", "downsample", "wait",
           "target_bad", "target_dev_deferred", "search", "hessian", "maxit", "evalbatch"]
    weights = np.array([4, 3, 2, 2, 4, 2, 1, 1, 5, 5, 3, 2, 2, 1, 2, 1, 1, 1, 1, 1, 3, 2, 1, 1, 2], float)
    weights /= weights.sum()
    history = []
    for i in range(400):
        op = ops[rng.choice(len(ops), p=weights)]
        k = int(rng.integers(0, 12))
        history.append((op, k))
        ra = call(eng[pkg.HANDOFF_SYNC], op, k)
        rb = call(eng[pkg.HANDOFF_ASYNC], op, k)
        if op.startswith("target") and isinstance(ra, tuple) and ra[0] == "error" and rb == "ok":
            # the blocking hand-off reports a failed build at once, the asynchronous one at the first consumer
            # (or, for the build after a failed one, at once as well): take the deferred verdict now
            rb = call(eng[pkg.HANDOFF_ASYNC], "wait", 0)
        assert ra == rb, "step %d, last calls %s: blocking %s, asynchronous %s" % (
            i, " ".join("%s(%d)" % h for h in history[-10:]), ra if not isinstance(ra, tuple) or ra[0] == "error" else "...",
            rb if not isinstance(rb, tuple) or rb[0] == "error" else "...")


# ---- the first evaluation of an align enqueued BEHIND a build that is still in flight ---------------------------------

def _upload3(hip, a):
    return [hip.upload(np.ascontiguousarray(a[:, k])) for k in range(3)]


def test_first_evaluation_behind_a_deferred_device_build_changes_no_number(pkg, hipmem, S):
    hip = hipmem
    """ndt_set_target_device_deferred + align against ndt_set_target_device + align, scan after scan: every answer
    bit for bit, and the deferred handle really took the short cut (counter)."""
    cfgs = [S.config_c2(), S.config_c1()]
    dev = [(_upload3(hip, c["target"]), len(c["target"]), _upload3(hip, c["source"]), len(c["source"]), c["guess"]) for c in cfgs]
    out = {}   # (hipMemcpy is synchronous: the uploads are complete)
    for deferred in (False, True):
        ndt = _ndt(pkg)
        res = []
        for rep in range(4):
            for tp, nt, sp, ns, guess in dev:
                (ndt.setInputTargetDeviceDeferred if deferred else ndt.setInputTargetDevice)(tp[0], tp[1], tp[2], nt)
                ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
                T = ndt.align(guess)
                r = ndt.getResult()
                res.append((T.copy(), r["iterations"], r["n_evaluations"], r["score"], r["hessian"].copy()))
        out[deferred] = res
        kept, discarded = ndt.speculationCounters()
        if deferred and pkg.get_tuning()["speculate_first"] != 0:
            assert kept >= 6 and discarded == 0, (kept, discarded)   # (the first build of a handle completes inside the call)
        else:
            assert kept == 0 and discarded == 0
        ndt.close()
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a[0], b[0]) and a[1:4] == b[1:4] and np.array_equal(a[4], b[4])
    # under the blocking hand-off the deferred call IS the blocking one: nothing is left in flight for the align
    ndt = _ndt(pkg)
    ndt.setHandoffMode(pkg.HANDOFF_SYNC)
    for rep in range(3):
        tp, nt, sp, ns, guess = dev[0]
        ndt.setInputTargetDeviceDeferred(tp[0], tp[1], tp[2], nt)
        ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
        T = ndt.align(guess)
        assert np.array_equal(T, out[False][0][0])
    assert ndt.speculationCounters() == (0, 0)
    ndt.close()


def test_first_evaluation_behind_a_build_that_is_refused_or_repeated(pkg, S):
    """The short cut must not survive a build that does not go through as enqueued: a cloud without a finite point
    (reported by the align, the prior returned), and a cloud the two-launch build declines (crowded voxels: repeated
    sort-based) -- same answers as the blocking hand-off."""
    c = S.config_c1()
    rng = np.random.default_rng(5)
    crowded = np.concatenate([c["target"], (rng.random((9000, 3)) * 0.9 + np.array([2.0, 2.0, 0.5])).astype(np.float32)])
    seq = [c["target"], crowded, c["target"], np.full((500, 3), np.nan, np.float32), c["target"]]
    out = {}
    for mode in (pkg.HANDOFF_SYNC, pkg.HANDOFF_ASYNC):
        ndt = _ndt(pkg)
        ndt.setHandoffMode(mode)
        res = []
        for t in seq:
            try:
                ndt.setInputTarget(t)
                ndt.setInputSource(c["source"])
                T = ndt.align(c["guess"])
                r = ndt.getResult()
                res.append(("ok", T.copy(), r["iterations"], r["score"]))
            except pkg.NdtError as e:
                res.append(("err", e.code))
        out[mode] = res
        if mode == pkg.HANDOFF_ASYNC and pkg.get_tuning()["speculate_first"] != 0:
            kept, discarded = ndt.speculationCounters()
            assert kept >= 1 and discarded >= 1, (kept, discarded)
        ndt.close()
    assert [r[0] for r in out[pkg.HANDOFF_SYNC]] == [r[0] for r in out[pkg.HANDOFF_ASYNC]] == ["ok", "ok", "ok", "err", "ok"]
    for a, b in zip(out[pkg.HANDOFF_SYNC], out[pkg.HANDOFF_ASYNC]):
        if a[0] == "ok":
            assert np.array_equal(a[1], b[1]) and a[2:] == b[2:]
        else:
            assert a[1] == b[1] == -4   # NDT_ERR_NO_TARGET


def test_partition_under_the_transfer_builds_the_same_grid(pkg, S):
    """Round 5: a steady-state asynchronous hand-off of a host target launches the two-launch
    build's partition chunk by chunk behind the pull kernels (the partition needs neither the grid geometry nor a
    sibling tile), so that only k_bucket_leaves is left behind the last chunk.  Same kernels over the same tiles: leaves,
    transform, iteration count and score are bit-identical to the hand-off that partitions behind the transfer
    (ndt_tuning::handoff_chunk_pass = 0) and to the blocking hand-off -- for clouds of several chunks, of one chunk, with
    a ragged last tile, and for one the two-launch build declines (a crowded voxel: the sort-based build repeats it)."""
    c3, c2 = S.config_c3(), S.config_c2()
    rng = np.random.default_rng(3)
    crowded = np.concatenate([c2["target"], (rng.uniform(0.0, 0.9, (9000, 3)) + np.float32(5.0)).astype(np.float32)])
    clouds = [(c3["target"][:700001], c3["source"][:60000], c3["guess"], 0.5), (c2["target"], c2["source"], c2["guess"], 1.0),
              (c2["target"][:5000], c2["source"][:4000], c2["guess"], 1.0), (crowded, c2["source"], c2["guess"], 1.0)]
    before = pkg.get_tuning()
    out = {}
    try:
        for variant in ("chunked", "behind", "sync"):
            pkg.set_tuning(handoff_chunk_pass=0 if variant == "behind" else 1)
            res = []
            for tgt, src, guess, leaf in clouds:
                ndt = _ndt(pkg, resolution=leaf)
                ndt.setHandoffMode(pkg.HANDOFF_SYNC if variant == "sync" else pkg.HANDOFF_ASYNC)
                for rep in range(3):   # (the first build of a handle waits for the geometry; the later ones are steady state)
                    t = _xyzi(tgt)
                    ndt.setInputTarget(t); t[:] = np.nan
                    ndt.setInputSource(src)
                    T = ndt.align(guess)
                r = ndt.getResult()
                res.append((ndt.getLeaves(), T.copy(), r["iterations"], r["score"], ndt.handoffCounters(), ndt.buildCounters()))
            out[variant] = res
    finally:
        pkg.set_tuning(**before)
    for k in range(len(clouds)):
        a = out["chunked"][k]
        for other in ("behind", "sync"):
            b = out[other][k]
            _leaves_equal(a[0], b[0])
            assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
            assert b[4] == (0, 0)
    # two steady-state hand-offs per cloud ran their partition under the transfer; the 700 001-point cloud in several launches
    assert out["chunked"][0][4][0] == 2 and out["chunked"][0][4][1] >= 2 * 3, out["chunked"][0][4]
    assert out["chunked"][1][4][0] == 2 and out["chunked"][2][4][0] == 2
    # the crowded cloud: its first steady-state build is declined (counted), the back-off then skips the two-launch build
    assert out["chunked"][3][5][1] >= 1 and out["chunked"][3][4][0] >= 1
