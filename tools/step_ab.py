"""A/B numbers on the C3 workload (tuning aid, not collected by pytest): device-resident step (build + set
source + align) as bench.py times it, build time, per-evaluation wall time and HIP-event kernel time.
Usage: python tools/step_ab.py <tag>; knobs come from the environment (NDT_PRELAUNCH_STREAMS,
NDT_DERIV_BLOCK, NDT_BUCKET_BUILD, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
tag = sys.argv[1] if len(sys.argv) > 1 else ""
hip = pkg.ranks.Hip(0)
cfg = S.config_c3()
if os.environ.get("NDT_STEP_AB_NSRC"):   # a rank's share of the scan in a multi-GPU job (first points of the scan)
    cfg["source"] = cfg["source"][:int(os.environ["NDT_STEP_AB_NSRC"])]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tp = [hip.upload(cfg["target"][:, a]) for a in range(3)]; sp = [hip.upload(cfg["source"][:, a]) for a in range(3)]
hip.synchronize()
g = pkg.ColMajor4f(cfg["guess"])
nt, ns = len(cfg["target"]), len(cfg["source"])
set_target = ndt.setInputTargetDeviceDeferred if os.environ.get("NDT_STEP_AB_DEFERRED") == "1" else ndt.setInputTargetDevice
def step():
    t0 = time.perf_counter(); set_target(tp[0], tp[1], tp[2], nt)
    t1 = time.perf_counter(); ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
    t2 = time.perf_counter(); ndt.align(g, return_transform=False)
    t3 = time.perf_counter(); return t1 - t0, t2 - t1, t3 - t2
for _ in range(10): step()
B, Sx, A, dev_build = [], [], [], []
for _ in range(40):
    b, s, a = step(); B.append(b); Sx.append(s); A.append(a); dev_build.append(ndt.getGridInfo()["ms_build"])
r = ndt.getResult()
ndt.enableKernelTiming(True); t0 = ndt.getTiming()
for _ in range(10): ndt.align(g, return_transform=False)
t1 = ndt.getTiming(); ndt.enableKernelTiming(False)
k_us = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
med = lambda v: 1e3 * float(np.median(v))
pre = ndt.prelaunchCounters() + (ndt.prelaunchOverlapped(),)
print("%-26s step %.3f ms = build %.3f (device %.3f) + source %.3f + align %.3f | it %d ev %d reused %d -> %.2f us/eval wall, kernel %.2f us | %.0f it/s | score %.9f | prelaunch %s"
      % (tag, med(B) + med(Sx) + med(A), med(B), float(np.median(dev_build)), med(Sx), med(A), r["iterations"], r["n_evaluations"],
         r["n_evaluations_reused"], 1e3 * med(A) / r["n_evaluations"], k_us, r["iterations"] / (np.median(B) + np.median(Sx) + np.median(A)), r["score"], pre), flush=True)
