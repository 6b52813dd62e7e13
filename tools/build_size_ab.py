"""Build time vs target size, fused launches against classic passes (tuning aid)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
torch.cuda.init(); dev = torch.device("cuda:0")
full = S.config_c3()["target"]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
out = []
for n in (30000, 65000, 130000, 260000, 520000, 1000000):
    t = full[:: max(1, len(full) // n)][:n]
    d = [torch.from_numpy(np.ascontiguousarray(t[:, a])).to(dev) for a in range(3)]
    torch.cuda.synchronize()
    W = []
    for i in range(40):
        t0 = time.perf_counter(); ndt.setInputTargetDevice(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(t)); t1 = time.perf_counter()
        if i >= 8: W.append(t1 - t0)
    out.append("%%d: %%.1f" %% (len(t), 1e6 * float(np.median(W))))
print("%%-8s build wall us by target size  %%s" %% (sys.argv[1], "  ".join(out)), flush=True)
''' % ROOT
for rep in range(2):
    for name, env in (("fused", {}), ("classic", {"NDT_FUSED_SORT": "0"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD, name], env=e, capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if "build wall" in l]
        print(out[0] if out else "FAILED %s rc=%d %s" % (name, r.returncode, r.stderr[-600:]), flush=True)
