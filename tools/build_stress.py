"""Stress of the target build (not collected by pytest): many builds of alternating clouds through the host
and the device entry points; every grid must hash like the first build of its cloud."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
clouds = [("c3", S.config_c3()["target"], 0.5), ("c2", S.config_c2()["target"], 0.5), ("c3/3", S.config_c3()["target"][::3], 0.5)]
hip = pkg.ranks.Hip(0)
dev = [[hip.upload(c[:, a]) for a in range(3)] for _, c, _ in clouds]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=5)

def digest():
    L = ndt.getLeaves()
    h = hashlib.sha256()
    for k in ("cell", "count", "mean", "cov", "icov", "evals"):
        h.update(np.ascontiguousarray(L[k]).tobytes())
    return h.hexdigest()[:16], len(L["cell"])

want, bad = {}, 0
t0 = time.time()
for it in range(reps):
    for k, (name, c, res) in enumerate(clouds):
        try:
            if it % 2:
                ndt.setInputTargetDevice(dev[k][0], dev[k][1], dev[k][2], len(c))
            else:
                ndt.setInputTarget(c)
            d = digest()
        except pkg.NdtError as e:
            bad += 1
            print("iteration", it, name, "ERROR", e, ndt.buildCounters(), flush=True)
            continue
        if name not in want:
            want[name] = d
        elif d != want[name]:
            bad += 1
            print("iteration", it, name, "MISMATCH", d, want[name], ndt.buildCounters(), flush=True)
print("builds %d, bad %d, counters %s, %.1f s" % (3 * reps, bad, ndt.buildCounters(), time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
