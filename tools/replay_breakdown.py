"""C5 replay through host clouds: where a frame's time goes (the loop's NumPy transform, setInputTarget, setInputSource, align)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
from slam_sam_amd import replay
stream = replay.make_stream(n_frames=8)
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
for rep in range(3):
    poses = [stream[0][1].copy()]
    rows = []
    for k in range(1, len(stream)):
        t0 = time.perf_counter()
        target = S.transform(poses[k - 1], stream[k - 1][0])
        t1 = time.perf_counter(); ndt.setInputTarget(target)
        t2 = time.perf_counter(); ndt.setInputSource(stream[k][0])
        t3 = time.perf_counter()
        guess = poses[k - 1] @ np.linalg.inv(poses[k - 2]) @ poses[k - 1] if k >= 2 else poses[k - 1].copy()
        t4 = time.perf_counter(); T = ndt.align(guess)
        t5 = time.perf_counter()
        poses.append(np.asarray(T, dtype=np.float64))
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, ndt.getNumEvaluations(), ndt.getHandoffTiming()["ms_build_wait"]))
    a = np.array(rows)
    print("rep %d: per frame (ms) transform %.3f  setInputTarget %.3f  setInputSource %.3f  guess %.3f  align %.3f | evals %s | build wait %s | auto %s bc %s"
          % (rep, *(1e3 * np.median(a[:, :5], axis=0)), a[:, 5].astype(int).tolist(), np.round(a[:, 6], 3).tolist(), ndt.autoStreamPlacement(), ndt.buildCounters()), flush=True)

print("-- device-resident keyframes")
dev = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
for rep in range(3):
    poses = [stream[0][1].copy()]
    rows = []
    for k in range(1, len(stream)):
        t0 = time.perf_counter()
        if k == 1: dev.putKeyframe(1000 * rep, stream[0][0])
        dev.putKeyframe(1000 * rep + k, stream[k][0])
        t1 = time.perf_counter(); dev.setInputTargetFromKeyframes([1000 * rep + k - 1], [poses[k - 1]])
        t2 = time.perf_counter(); dev.setInputSourceFromKeyframe(1000 * rep + k)
        t3 = time.perf_counter()
        guess = poses[k - 1] @ np.linalg.inv(poses[k - 2]) @ poses[k - 1] if k >= 2 else poses[k - 1].copy()
        t4 = time.perf_counter(); T = dev.align(guess)
        t5 = time.perf_counter()
        if k >= 2: dev.eraseKeyframe(1000 * rep + k - 2)
        t6 = time.perf_counter()
        poses.append(np.asarray(T, dtype=np.float64))
        rows.append((t1 - t0, t2 - t1, t3 - t2, t5 - t4, t6 - t5, dev.getHandoffTiming()["ms_build_wait"]))
    a = np.array(rows)
    print("rep %d: per frame (ms) putKeyframe %.3f  targetFromKeyframes %.3f  sourceFromKeyframe %.3f  align %.3f  erase %.3f | build wait %.3f"
          % (rep, *(1e3 * np.median(a[:, :5], axis=0)), np.median(a[:, 5])), flush=True)
    for k in (len(stream) - 2, len(stream) - 1): dev.eraseKeyframe(1000 * rep + k)
