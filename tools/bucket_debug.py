"""debug aid: the two-launch build on the tile-boundary clouds (not collected by pytest)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for n in [int(a) for a in sys.argv[1:]] or [100003, 262144]:
    rng = np.random.default_rng(n)
    tgt = (rng.normal(0, 1, (n, 3)) * np.array([14.0, 9.0, 1.5])).astype(np.float32)
    tgt[rng.integers(0, n, 5)] = np.nan
    kw = dict(resolution=0.7, step_size=0.1, trans_epsilon=1e-4, max_iterations=5, min_points_per_voxel=6)
    ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
    for rep in range(3):
        try:
            ndt.setInputTarget(tgt)
            gi = ndt.getGridInfo()
            print(n, rep, "ok leaves", gi["n_leaves"], "cells", gi["n_cells"], "counters", ndt.buildCounters(), flush=True)
        except pkg.NdtError as e:
            print(n, rep, "ERR", e, ndt.buildCounters(), flush=True)
