"""The other single-GPU configurations of BASELINE.json on the driver-run line (bench.py's `configs` block):

  C2  configs[1]: 128 x 1024 Ouster-style scan (131 072 points) scan-to-scan, 1.0 m voxels
  C5  configs[4]: pipeline_lo_svn-style replay -- sequential odometry over a synthetic OS-2-128 stream with the engine
      in the registration slot (slam-sam_amd/replay.py mirrors run/pipeline.cpp:494-610 and
      run/pipeline_lo_svn.cpp:376-388): NDT through host clouds, NDT with device-resident keyframes, SVN-NDT K = 20

Each is bounded to a few seconds.  Nothing here is `value`; nothing here touches the oracle.
"""
import time

import numpy as np

HBM_PEAK_GBS = 8000.0
ALGO_BYTES_PER_POINT = lambda nbar: 12.0 + 7 * 4.0 + nbar * 48.0  # noqa: E731  SURVEY.md section 8(d)


def c2_block(pkg, hip, steps=20, warmup=5):
    """C2 scan-to-scan: clouds resident in HBM (as the headline step), step = build + align."""
    S = pkg.synth
    cfg = S.config_c2()
    n_t, n_s = len(cfg["target"]), len(cfg["source"])
    tptr = [hip.upload(cfg["target"][:, a]) for a in range(3)]
    sptr = [hip.upload(cfg["source"][:, a]) for a in range(3)]
    hip.synchronize()
    ndt = pkg.NormalDistributionsTransform(device_id=-1, resolution=float(cfg["resolution"]), step_size=0.1,
                                           trans_epsilon=1e-4, max_iterations=35)
    guess = pkg.ColMajor4f(cfg["guess"])

    def step():
        t0 = time.perf_counter()
        ndt.setInputTargetDeviceDeferred(tptr[0], tptr[1], tptr[2], n_t)   # (enqueued; finished inside the step, as bench.py's)
        t1 = time.perf_counter()
        ndt.setInputSourceDeviceView(sptr[0], sptr[1], sptr[2], n_s)
        ndt.align(guess, return_transform=False)
        return t1 - t0, time.perf_counter() - t1

    for _ in range(40 + warmup):   # (40: device wake-up after the seconds of synthesis above, as bench.py's timed regions)
        step()
    hip.synchronize()
    iters = evals = 0
    t_build = t_align = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        tb, ta = step()
        iters += ndt.getFinalNumIteration()
        evals += ndt.getNumEvaluations()
        t_build += tb
        t_align += ta
    hip.synchronize()
    el = time.perf_counter() - t0
    err_t, err_r = S.pose_error(ndt.getResult()["T"], cfg["gt"])
    # instrumented repeat: the derivative kernel's own duration (events attached to the dispatch), ordinary launches
    ndt.enableKernelTiming(True)
    tm0 = ndt.getTiming()
    for _ in range(max(3, steps // 4)):
        step()
    r = ndt.getResult()
    tm1 = ndt.getTiming()
    ndt.enableKernelTiming(False)
    gi = ndt.getGridInfo()
    n_timed = tm1["n_timed_evals"] - tm0["n_timed_evals"]
    ms_kernel = (tm1["ms_eval_kernel_total"] - tm0["ms_eval_kernel_total"]) / max(n_timed, 1)
    nbar = r["n_pairs"] / float(n_s)
    algo = n_s * ALGO_BYTES_PER_POINT(nbar)
    achieved = algo / (ms_kernel * 1e-3) / 1e9 if ms_kernel > 0 else 0.0
    ndt.close()
    return {"workload": "C2 scan-to-scan: 128x1024 Ouster-style scan (%d pts) into the scan before it (%d pts), 1.0 m voxel, "
                        "DIRECT7; step = voxel-grid build + align, clouds resident in HBM" % (n_s, n_t),
            "value": iters / el, "unit": "iterations/s", "ms_scan": 1e3 * el / steps, "steps": steps,
            "ms_target_build": 1e3 * t_build / steps, "ms_align": 1e3 * t_align / steps,
            "iterations_per_align": iters / steps, "evaluations_per_align": evals / steps,
            # (the build is only enqueued by the set-target call and finishes inside the align: the scan without the build's
            # own duration, per evaluation)
            "us_per_evaluation": 1e6 * max(el - steps * 1e-3 * gi["ms_build"], 0.0) / max(evals, 1),
            "voxels": int(gi["n_leaves"]), "mean_neighbors": nbar,
            "roofline": {"kernel": "k_derivatives", "bound": "latency/valu", "roof": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": algo, "ms_per_launch": ms_kernel, "launches_timed": int(n_timed)},
            "ms_target_build_device": gi["ms_build"],
            "final_error_vs_ground_truth": {"m": err_t, "rad": err_r}}


def c5_block(pkg, n_frames=8):
    """C5 replay: Hz end to end for the three ways the drivers' loop can run on the engine."""
    from slam_sam_amd import replay
    S = pkg.synth
    stream = replay.make_stream(n_frames=n_frames)
    n_pts = len(stream[0][0])
    kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    out = {"workload": "C5 replay: scan-to-scan odometry over %d synthetic OS-2-128 frames (%d pts each), 1.0 m voxel; the "
                       "loop body of run/pipeline.cpp:494-610 / run/pipeline_lo_svn.cpp:376-388 around the engine "
                       "(the reference ships no recording)" % (n_frames, n_pts)}

    def summary(res, label):
        err = replay.trajectory_errors(res["poses"], stream[:len(res["poses"])])
        return {"what": label, "hz": res["hz"], "ms_per_frame": float(res["ms"].mean()), "frames": len(res["ms"]),
                # the registration calls alone (setInputTarget / setInputSource / align): a frame of the harness also holds
                # its own NumPy transform of the previous scan (0.7-1.2 ms for 131 k points, box dependent)
                "ms_engine_per_frame": float(res["ms_engine"].mean()), "hz_engine": 1e3 / float(res["ms_engine"].mean()),
                "iterations": [int(i) for i in res["iterations"]],
                "final_error_vs_ground_truth": {"m": err[-1][0], "rad": err[-1][1]},
                "max_error_m": max(e[0] for e in err)}

    ndt = pkg.NormalDistributionsTransform(device_id=-1, **kw)
    replay.run_lidar_odometry(ndt, stream[:3])   # warm-up (allocations)
    out["ndt_host_clouds"] = summary(replay.run_lidar_odometry(ndt, stream),
                                     "NDT, host clouds through setInputTarget / setInputSource per frame (the frame's time "
                                     "includes the loop's own NumPy transform of the previous scan, the drivers' "
                                     "pcl::transformPointCloud)")
    ndt.close()
    dev = pkg.NormalDistributionsTransform(device_id=-1, **kw)
    replay.run_lidar_odometry(dev, stream[:3], mode="ndt_keyframes")
    out["ndt_device_keyframes"] = summary(replay.run_lidar_odometry(dev, stream, mode="ndt_keyframes"),
                                          "NDT, scans archived on the device (keyframe API): one upload per scan, target "
                                          "assembled and source taken from the archive")
    dev.close()
    # SVN-NDT as config/register_config.json:13-19 sets it: K = 20, 100 iterations max, h = 5.0, step 0.05
    K = 20
    svn = pkg.SvnNormalDistributionsTransform(device_id=-1, resolution=1.0)
    svn.setParticleCount(K); svn.setMaxIterations(100); svn.setKernelBandwidth(5.0)
    svn.setStepSize(0.05); svn.setEarlyStopThreshold(1e-4); svn.setOutlierRatio(0.55)
    rng = np.random.default_rng(3)   # the lo_svn driver hands align() the INS pose as prior: truth + a few cm / mrad
    sv_stream = stream[:5]
    priors = [gt @ S.pose_matrix(*(rng.normal(0, 0.03, 3)), *(rng.normal(0, 0.003, 3))) for _, gt in sv_stream]
    replay.run_lidar_odometry(svn, sv_stream[:2], mode="svn", priors=priors)   # warm-up
    res = replay.run_lidar_odometry(svn, sv_stream, mode="svn", priors=priors)
    sv = summary(res, "SVN-NDT, K = 20 particles, <= 100 iterations, host clouds; Stage 1 of every iteration = ONE batched "
                      "launch of the derivative kernel for all particles")
    perr = [S.pose_error(p, gt)[0] for p, (_, gt) in zip(priors, sv_stream)]
    sv["mean_error_m"] = float(np.mean([e[0] for e in replay.trajectory_errors(res["poses"], sv_stream)][1:]))
    sv["mean_prior_error_m"] = float(np.mean(perr[1:]))
    # SVN iterations per second of the registration calls, beside the only timing the reference publishes (BASELINE.md 2:
    # its own pipeline_lo_svn log, output/output.txt -- other hardware, an older revision, <= 65 536 points per scan)
    sv["svn_iterations_per_sec"] = float(sum(res["iterations"]) / (res["ms_engine"].sum() * 1e-3))
    sv["reference_log"] = {"svn_iterations_per_sec": 13.6, "ms_per_iteration": 73.5,
                           "source": "BASELINE.md section 2 (output/output.txt: K = 20, CPU/OpenMP, unstated machine, scan <= 65 536 points)"}
    # Stage-1 launch: the batched kernel's own duration and ITS algorithmic fraction (K poses x B_eval per launch)
    svn.enableKernelTiming(True)
    tm0 = svn.getTiming()
    svn.setInputTarget(S.transform(sv_stream[0][1], sv_stream[0][0]))
    r1 = svn.align(sv_stream[1][0], priors[1], seed=1)
    tm1 = svn.getTiming()
    svn.enableKernelTiming(False)
    n_timed = tm1["n_timed_evals"] - tm0["n_timed_evals"]
    ms_launch = (tm1["ms_eval_kernel_total"] - tm0["ms_eval_kernel_total"]) / max(n_timed, 1)
    e = svn.evalDerivatives(np.zeros((1, 6)), transforms=[r1["final_pose"]])[0]
    nbar = e["n_pairs"] / float(n_pts)
    algo = K * n_pts * ALGO_BYTES_PER_POINT(nbar)
    ach = algo / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
    sv["stage1"] = {"kernel": "k_derivatives<batched>", "poses_per_launch": K, "ms_per_launch": ms_launch,
                    "launches_timed": int(n_timed), "mean_neighbors": nbar, "algorithmic_bytes_per_launch": algo,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "ms_stage1_per_align": r1["ms_stage1"], "ms_stage2_per_align": r1["ms_stage2"],
                    "ms_stage3_per_align": r1["ms_stage3"], "iterations": r1["iterations"]}
    out["svn_k20"] = sv
    svn.close()
    return out
