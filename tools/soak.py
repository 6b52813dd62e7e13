"""Determinism soak of the evaluation hand-off (tagged slots, tickets, host polling): many thousand
align() calls at several launch shapes must reproduce the first result bit for bit (tuning aid /
stress test, not collected by pytest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
gt_inv = np.linalg.inv(cfg["gt"])
big = (cfg["target"].astype(np.float64) @ gt_inv[:3, :3].T + gt_inv[:3, 3]).astype(np.float32)
shapes = [("200k (241 blocks of 832)", cfg["source"], reps),
          ("1k (4 blocks of 256)", cfg["source"][:1000], reps),
          ("25k (98 blocks of 256)", cfg["source"][:25000], reps),
          ("130k (254 blocks of 512)", cfg["source"][:130000], reps),
          ("1M (1954 rows)", big, max(50, reps // 20)),
          ("1.2M (two-level sum)", np.concatenate([big, big[:200000] + np.float32(0.01)]), max(50, reps // 20))]
g = pkg.ColMajor4f(cfg["guess"])
bad = 0
for name, src, n in shapes:
    ndt.setInputSource(np.ascontiguousarray(src))
    ndt.align(g); ref = ndt.getResult()
    t0 = time.perf_counter(); evals = 0
    for i in range(n):
        try:
            ndt.align(g, return_transform=False)
        except pkg.NdtError as e:
            bad += 1
            print("FAILED", name, "align", i, e, "counters (used, quit, timeouts)", ndt.prelaunchCounters(), flush=True)
            if bad > 5: sys.exit(1)
            continue
        evals += ndt.getNumEvaluations()
        if i % 97 == 0 or i == n - 1:
            r = ndt.getResult()
            if r["score"] != ref["score"] or not np.array_equal(r["T"], ref["T"]) or not np.array_equal(r["hessian"], ref["hessian"]):
                bad += 1
                print("MISMATCH", name, i, r["score"], ref["score"], flush=True)
        elif ndt.getFinalNumIteration() != ref["iterations"]:
            bad += 1
            print("MISMATCH (iterations)", name, i, flush=True)
    print("%-28s %6d aligns, %8d evaluations, %.1f s: %s" % (name, n, evals, time.perf_counter() - t0, "identical" if not bad else "MISMATCHES"), flush=True)
# the whole step: rebuild (fused launches, tagged tables, done word) + align
import hashlib
def leaf_hash():
    L = ndt.getLeaves()
    h = hashlib.sha256()
    for k in ("cell", "count", "mean", "icov"): h.update(np.ascontiguousarray(L[k]).tobytes())
    return h.hexdigest()
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
ndt.align(g); ref = ndt.getResult(); href = leaf_hash()
t0 = time.perf_counter(); nsteps = max(500, reps // 4)
for i in range(nsteps):
    ndt.setInputTarget(cfg["target"] if i % 2 == 0 else cfg["target"][:900000])   # two sizes in turn: 123 / 110 tiles
    if i % 2 == 0:
        ndt.align(g, return_transform=False)
        if ndt.getFinalNumIteration() != ref["iterations"]: bad += 1; print("MISMATCH (step, iterations)", i, flush=True)
        if i % 200 == 0:
            r = ndt.getResult()
            if r["score"] != ref["score"] or not np.array_equal(r["T"], ref["T"]) or leaf_hash() != href:
                bad += 1; print("MISMATCH (step)", i, flush=True)
print("%-28s %6d steps, %.1f s: %s (fused-sort fallbacks %d)" % ("rebuild + align", nsteps, time.perf_counter() - t0, "identical" if not bad else "MISMATCHES", ndt.buildCounters()[0]), flush=True)
# batched path
p = ref["pose"]
ndt.setInputSource(cfg["source"])
e0 = ndt.evalDerivatives(np.tile(p, (20, 1)))
for i in range(max(50, reps // 20)):
    e = ndt.evalDerivatives(np.tile(p, (20, 1)))
    for a, b in zip(e, e0):
        if a["score"] != b["score"] or not np.array_equal(a["hessian"], b["hessian"]):
            bad += 1
print("batched K=20: %s" % ("identical" if not bad else "MISMATCHES"))
sys.exit(1 if bad else 0)
