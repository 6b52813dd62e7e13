"""align() per search mode / Hessian mode on C2 and C3 (tuning aid, not collected by pytest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
for cname, cfg in (("C2", S.config_c2()), ("C3", S.config_c3())):
    for mode in ("DIRECT7", "DIRECT1", "KDTREE", "DIRECT26", "MULTIGRID"):
        ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=float(cfg["resolution"]), step_size=0.1,
                                               trans_epsilon=1e-4, max_iterations=35)
        if os.environ.get("NDT_MODES_PACKED") == "1": ndt.setRecordFormat(pkg.RECORDS_PACKED48)   # (not applied to MULTIGRID)
        if mode == "MULTIGRID":   # the map as 2 x 2 tiles that overlap by 2 m, united by createVoxelKdtree
            t = cfg["target"]; mx, my = np.median(t[:, 0]), np.median(t[:, 1])
            t0 = time.perf_counter()
            for k, (sx, sy) in enumerate(((-1, -1), (-1, 1), (1, -1), (1, 1))):
                ndt.addTarget(t[(sx * (t[:, 0] - mx) > -2.0) & (sy * (t[:, 1] - my) > -2.0)], k)
            t1 = time.perf_counter(); ndt.createVoxelKdtree(); t2 = time.perf_counter()
            print("%s MULTIGRID 4 tiles: addTarget x4 %.1f ms, createVoxelKdtree %.1f ms, %d leaves" %
                  (cname, 1e3 * (t1 - t0), 1e3 * (t2 - t1), ndt.getGridInfo()["n_leaves"]), flush=True)
            ndt.setInputSource(cfg["source"])
        else:
            ndt.setNeighborhoodSearchMethod(getattr(pkg, mode))
            ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
        for _ in range(3): ndt.align(cfg["guess"])
        ts = []
        for _ in range(10):
            t = time.perf_counter(); ndt.align(cfg["guess"]); ts.append(time.perf_counter() - t)
        r = ndt.getResult()
        ndt.enableKernelTiming(True); t0 = ndt.getTiming()
        for _ in range(3): ndt.align(cfg["guess"])
        t1 = ndt.getTiming(); ndt.enableKernelTiming(False)
        k_us = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
        et, er = S.pose_error(r["T"], cfg["gt"])
        print("%s %-8s align %.3f ms  it %2d ev %2d  %.1f us/eval  kernel %.2f us  nbar %.2f  err %.4f m %.5f rad" %
              (cname, mode, 1e3 * np.median(ts), r["iterations"], r["n_evaluations"], 1e6 * np.median(ts) / r["n_evaluations"],
               k_us, r["n_pairs"] / len(cfg["source"]), et, er), flush=True)
        ndt.close()
