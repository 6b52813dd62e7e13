"""SVN-NDT iteration breakdown on the C5 stream (tuning aid, not collected by pytest)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
from slam_sam_amd import replay
stream = replay.make_stream(n_frames=3)
svn = pkg.SvnNormalDistributionsTransform(device_id=0, resolution=1.0)
svn.setParticleCount(20); svn.setMaxIterations(100); svn.setKernelBandwidth(5.0)
svn.setStepSize(0.05); svn.setEarlyStopThreshold(1e-4); svn.setOutlierRatio(0.55)
svn.setInputTarget(stream[0][0])
for k in range(3):
    r = svn.align(stream[1][0], stream[1][1], seed=3)
print("K=20, %d points: %d iterations, total %.2f ms = stage1 %.2f + stage2 %.2f + stage3 %.2f  (per iteration %.1f us: %.1f + %.1f + %.1f)"
      % (len(stream[1][0]), r["iterations"], r["ms_total"], r["ms_stage1"], r["ms_stage2"], r["ms_stage3"],
         1e3 * r["ms_total"] / r["iterations"], 1e3 * r["ms_stage1"] / r["iterations"], 1e3 * r["ms_stage2"] / r["iterations"], 1e3 * r["ms_stage3"] / r["iterations"]))
svn.enableKernelTiming(True); t0 = svn.getTiming()
r = svn.align(stream[1][0], stream[1][1], seed=3)
t1 = svn.getTiming()
print("batched kernel: %.1f us per launch (20 poses)" % (1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])))
