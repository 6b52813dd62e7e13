"""Cost of putting the source into block order on the C3-wide map (tuning aid): align() right after a
fresh setInputSource (orders it) against a repeated align() on the ordered source."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
torch.cuda.init(); dev = torch.device("cuda:0")
cfg = S.config_c3_wide()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
src = [torch.from_numpy(np.ascontiguousarray(cfg["source"][:, a])).to(dev) for a in range(3)]
torch.cuda.synchronize()
sp = [t.data_ptr() for t in src]; ns = len(cfg["source"])
g = pkg.ColMajor4f(cfg["guess"])
fresh, again = [], []
for i in range(40):
    ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
    t0 = time.perf_counter(); ndt.align(g, return_transform=False); t1 = time.perf_counter()
    ndt.align(g, return_transform=False); t2 = time.perf_counter()
    if i >= 5: fresh.append(t1 - t0); again.append(t2 - t1)
print("%s: align after a fresh source %.1f us, repeated %.1f us -> ordering the source costs %.1f us"
      % (os.environ.get("NDT_FUSED_SORT", "fused"), 1e6 * np.median(fresh), 1e6 * np.median(again), 1e6 * (np.median(fresh) - np.median(again))), flush=True)
