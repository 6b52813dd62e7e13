"""Replay harness for BASELINE config C5: sequential lidar odometry over a synthetic
OS-2-128-style stream, with the GPU engine called exactly where the reference's drivers call
their registration object.

Mirrors the per-keyframe body of the reference's odometry threads (nothing else of them):
  * run/pipeline.cpp:494-610  -- target = previous scan moved to the map frame by the previous
    estimate (:554-556), registration->setInputTarget / setInputSource / align(guess) (:557-561),
    getFinalTransformation (:566), guess = constant-velocity prediction;
  * run/pipeline_lo_svn.cpp:376-388 -- target = previous map-frame cloud, svn align(source, prior).
The stream itself (decode, sync, GTSAM) is out of scope; scans come from synth.OusterSim.
"""
import time

import numpy as np

from . import synth


def make_stream(n_frames=10, beams=128, cols=1024, seed=42, step=0.8, yaw_step=0.01):
    """[(scan in the sensor frame, ground-truth T_map_sensor), ...] along a gently curving track."""
    scene = synth.Scene(seed=seed)
    sim = synth.OusterSim(scene, beams=beams, cols=cols)
    out = []
    for k in range(n_frames):
        T = synth.pose_matrix(-4.0 + step * k, 0.2 * np.sin(0.3 * k), 2.0, 0.0, 0.0, yaw_step * k)
        out.append((sim.scan(T, seed=seed + 10 + k), T))
    return out


def run_lidar_odometry(engine, stream, mode="ndt", svn_seed=0, priors=None):
    """Scan-to-scan odometry.  engine: an object with the pclomp (mode 'ndt') or svn_ndt
    (mode 'svn') method names.  priors: optional per-frame prior poses (the INS pose the
    reference's lo_svn driver passes, run/pipeline_lo_svn.cpp:388); default = constant-velocity
    prediction.  Returns dict(poses, ms, hz, iterations)."""
    poses = [stream[0][1].copy()]
    ms, iters, ms_engine = [], [], []   # ms_engine: the registration calls alone (set target / source, align), without the
    t_all = time.perf_counter()         # loop's own NumPy transform of the previous scan (the drivers' pcl::transformPointCloud)
    for k in range(1, len(stream)):
        scan, _ = stream[k]
        prev_scan = stream[k - 1][0]
        t0 = time.perf_counter()
        if mode != "ndt_keyframes":
            target = synth.transform(poses[k - 1], prev_scan)       # ref: run/pipeline.cpp:554-556
        if priors is not None:
            guess = np.asarray(priors[k], dtype=np.float64)
        elif k >= 2:                                                 # constant-velocity prediction
            guess = poses[k - 1] @ np.linalg.inv(poses[k - 2]) @ poses[k - 1]
        else:
            guess = poses[k - 1].copy()
        t_e = time.perf_counter()
        if mode == "ndt_keyframes":
            # device-resident variant: every scan crosses PCIe once, as a keyframe;
            # the target is keyframe k-1 moved by its pose ON the device, the source is keyframe k
            if k == 1:
                engine.putKeyframe(0, prev_scan)
            engine.putKeyframe(k, scan)
            engine.setInputTargetFromKeyframes([k - 1], [poses[k - 1]])
            engine.setInputSourceFromKeyframe(k)
            T = engine.align(guess)
            iters.append(engine.getFinalNumIteration())
            if k >= 2:
                engine.eraseKeyframe(k - 2)
            poses.append(np.asarray(T, dtype=np.float64))
            ms.append(1e3 * (time.perf_counter() - t0))
            ms_engine.append(1e3 * (time.perf_counter() - t_e))
            continue
        engine.setInputTarget(target)                                # :557
        if mode == "ndt":
            engine.setInputSource(scan)                              # :558
            T = engine.align(guess)                                  # :561
            iters.append(engine.getFinalNumIteration())
        else:
            r = engine.align(scan, guess, seed=svn_seed + k)         # ref: run/pipeline_lo_svn.cpp:387-388
            T = r["final_pose"]
            iters.append(r["iterations"])
        poses.append(np.asarray(T, dtype=np.float64))
        ms.append(1e3 * (time.perf_counter() - t0))
        ms_engine.append(1e3 * (time.perf_counter() - t_e))
    wall = time.perf_counter() - t_all
    return dict(poses=poses, ms=np.array(ms), ms_engine=np.array(ms_engine), hz=(len(stream) - 1) / wall, iterations=iters)


def trajectory_errors(poses, stream):
    """Per-frame (translation m, rotation rad) error against the stream's ground truth."""
    return [synth.pose_error(p, gt) for p, (_, gt) in zip(poses, stream)]
