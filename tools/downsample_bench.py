"""pcl::VoxelGrid on the device: time of ndt_voxel_downsample_device on the C3 map (1 M points) at a few leaf sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3(); m = cfg["target"]
hip = pkg.ranks.Hip(0)
d = [hip.upload(np.ascontiguousarray(m[:, a])) for a in range(3)]
o = [hip.upload(np.zeros(len(m), np.float32)) for _ in range(3)]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5)
for leaf in (0.1, 0.5, 2.0):
    for _ in range(3): n_out = ndt.voxelDownsampleDevice(d[0], d[1], d[2], len(m), leaf, o[0], o[1], o[2], len(m))
    ts = []
    for _ in range(20):
        t = time.perf_counter(); ndt.voxelDownsampleDevice(d[0], d[1], d[2], len(m), leaf, o[0], o[1], o[2], len(m)); ts.append(time.perf_counter() - t)
    print("leaf %.2f m: %d -> %d points, %.3f ms per call (median of 20, wall incl. two host syncs)" % (leaf, len(m), n_out, 1e3 * np.median(ts)), flush=True)
