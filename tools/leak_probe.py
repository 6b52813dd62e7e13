"""Resident set of a process that registers scans through the asynchronous hand-off for a long time (debugging aid, not
collected by pytest): no allocation on the Python side inside the loop, so growth would be the engine's.
Usage: python tools/leak_probe.py [scans]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
cfg = S.config_c2()
t, s, g = cfg["target"], cfg["source"], pkg.ColMajor4f(cfg["guess"])
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
def rss_mb(): return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2**20
marks = []
for i in range(n):
    ndt.setInputTarget(t); ndt.setInputSource(s); ndt.align(g, return_transform=False)
    if i in (200, n // 4, n // 2, 3 * n // 4, n - 1): marks.append((i, rss_mb()))
print("resident set (MB) after scan: " + "  ".join("%d: %.1f" % m for m in marks), flush=True)
