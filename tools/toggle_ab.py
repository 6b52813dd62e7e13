"""In-process interleaved A/B of a per-handle switch (tuning aid, not collected by pytest): steps alternate between
off and on on ONE handle, so box-to-box and run-to-run drift cancel; paired differences resolve < 1 us per step.
Usage: python tools/toggle_ab.py speculation.  C3 and C2 with device-resident clouds
(ndt_set_target_device_deferred), C2 with host clouds (asynchronous hand-off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
hip = pkg.ranks.Hip(0)
N = int(os.environ.get("NDT_AB_STEPS", "300"))
FEATURE = sys.argv[1] if len(sys.argv) > 1 else "speculation"
SETTER = {"speculation": "setSpeculation"}[FEATURE]   # (add a setter here for the next switch to be A/B-ed)

def run(name, cfg, res, host):
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=res, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    g = pkg.ColMajor4f(cfg["guess"])
    if not host:
        tp = [hip.upload(cfg["target"][:, a]) for a in range(3)]; sp = [hip.upload(cfg["source"][:, a]) for a in range(3)]
        hip.synchronize()
    nt, ns = len(cfg["target"]), len(cfg["source"])
    def step():
        t0 = time.perf_counter()
        if host:
            ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
        else:
            ndt.setInputTargetDeviceDeferred(tp[0], tp[1], tp[2], nt); ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
        ndt.align(g, return_transform=False)
        return time.perf_counter() - t0
    for _ in range(10): step()
    t = {0: [], 1: []}
    for i in range(2 * N):
        getattr(ndt, SETTER)(i & 1)
        t[i & 1].append(step())
    k = ndt.speculationCounters()
    a, b = 1e6 * np.array(t[0]), 1e6 * np.array(t[1])
    print("%-12s %-24s off: median %.1f mean %.1f us | on: median %.1f mean %.1f us | on - off: median %+.2f, mean of pairs %+.2f +- %.2f us | kept %d discarded %d"
          % (FEATURE, name, np.median(a), a.mean(), np.median(b), b.mean(), np.median(b) - np.median(a), (b - a).mean(),
             (b - a).std() / np.sqrt(len(a)), k[0], k[1]), flush=True)
    ndt.close()

run("C3 device clouds", S.config_c3(), 0.5, False)
run("C2 device clouds", S.config_c2(), 1.0, False)
run("C2 host clouds (async)", S.config_c2(), 1.0, True)
