#!/usr/bin/env python3
"""NDT align() benchmark on MI355X -- the metric of BASELINE.json.

A "step" is one scan registration of workload C3 (a 200 000-point scan into a
1 000 000-point voxelised submap, 0.5 m voxels, DIRECT7): setInputTarget (voxel-grid
build) + setInputSource + align(), with both clouds already resident in HBM when the
timed region starts.  `value` = Newton iterations / wall second over the K timed steps
(build time included), whole job.  With N > 1 (one process per GPU) the target is replicated,
the source is sharded and every derivative evaluation ends in one 256-byte sum over the ranks:
strong scaling.  Every transport of that sum is timed back to back -- shared memory (pinned-host
partials), the in-kernel xGMI peer-write exchange, and RCCL's all-reduce -- and reported under
config.reduce_variants; NDT_BENCH_REDUCE=shm|p2p|rccl pins one.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 runs either way: launched by `python -m torch.distributed.run --nproc-per-node N ...` (RANK /
LOCAL_RANK / WORLD_SIZE from the environment), or as a plain `python bench.py --gpus N`, which then
becomes a parent that touches no GPU and starts its N ranks itself (slam-sam_amd/ranks.py).  No torch
in either case: device buffers come from the HIP runtime the engine links, the ranks meet on a
shared-memory board, so libndt_hip.so is the only HIP / RCCL user in a rank's process.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (k_derivatives):
algorithmic bytes per launch = N_src * (12 + 7*4 + nbar*48) over its mean
HIP-event duration, measured in an instrumented repeat of the timed steps.  `cpu_baseline`
is the CPU oracle (OpenMP, all host cores) on a bounded sample of the same steps -- a
reported baseline, not the target.
"""
import argparse
import json
import os

# NumPy's BLAS keeps a pool of spinning worker threads alive for a while after every matmul; on the
# GPU box's 16-CPU share they compete with the engine's polling thread (measured: a 1M x 3 matmul
# right before a timed region cost it 2.4x).  The synthetic workloads need no threaded BLAS.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "BLIS_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
# A rank that stops answering inside an all-reduce is a failed reduce variant here, not something to sit out
os.environ.setdefault("NDT_COMM_TIMEOUT_S", "20")
import gc
import hashlib
import struct
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_POINT = lambda nbar: 12.0 + 7 * 4.0 + nbar * 48.0  # noqa: E731  SURVEY.md section 8(d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="per CPU-baseline thread setting")
    return ap.parse_args()


def build_fast_oracle():
    """The CPU baseline is timed on an -O3 -march=native build of the oracle, compiled HERE (the
    box whose cores are timed: -march=native of another machine could fault) into a scratch
    directory; the -O2 -ffp-contract=off build under oracle/ stays the parity checker.  Returns
    (module, flags) -- the exact build and its flags if the compile fails."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exact_flags = "-O2 -fopenmp -ffp-contract=off (exactness build)"
    try:
        d = tempfile.mkdtemp(prefix="ndt_oracle_fast_")
        for f in ("ndt_oracle.cpp", "ndt_oracle.h", "oracle.py", "Makefile"):
            shutil.copy(os.path.join(ROOT, "oracle", f), d)
        flags = "-O3 -march=native -std=c++17 -fPIC -fopenmp"
        subprocess.check_call(["g++"] + flags.split() + ["-shared", "-o", os.path.join(d, "libndt_oracle.so"),
                                                         os.path.join(d, "ndt_oracle.cpp")],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        spec = importlib.util.spec_from_file_location("ndt_oracle_fast", os.path.join(d, "oracle.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.lib()
        return mod, flags
    except Exception as e:  # noqa: BLE001
        print("bench: fast oracle build failed (%s); timing the exactness build" % e, file=sys.stderr, flush=True)
        O = ge.load_oracle()
        O.build()
        return O, exact_flags


def usable_cpus():
    """CPUs this job may actually use: the affinity mask, cut down to the cgroup CPU quota (a 1-GPU
    box shows all 256 host CPUs to a job that is allowed 16 of them; 256 OpenMP threads on a
    16-CPU quota ran the baseline 10x slower than 8 threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    if quota is None:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


WAKE_STEPS = int(os.environ.get("NDT_BENCH_WAKE_STEPS", "60"))


def cpu_baseline(cfg, params, seconds):
    """The oracle timed on this box's host cores; step = grid build + align, like the GPU step.
    Two thread settings: every core this job may use, and the 8 of config/register_config.json:3."""
    O, flags = build_fast_oracle()
    nproc = os.cpu_count() or 1
    avail = usable_cpus()

    def run(threads):
        prm = O.default_params(num_threads=threads, **params)
        iters, steps, t_build, t_align = 0, 0, 0.0, 0.0
        t0 = time.perf_counter()
        while True:
            ta = time.perf_counter()
            grid = O.Grid(cfg["target"], prm)
            tb = time.perf_counter()
            r = grid.align(cfg["source"], cfg["guess"])
            tc = time.perf_counter()
            iters += r["iterations"]
            steps += 1
            t_build += tb - ta
            t_align += tc - tb
            if (tc - t0 >= seconds and steps >= 2) or steps >= 200:
                break
        wall = time.perf_counter() - t0
        return {"value": iters / wall, "threads": threads, "steps": steps, "seconds": wall,
                "ms_per_step": 1e3 * wall / steps, "ms_build": 1e3 * t_build / steps,
                "ms_align": 1e3 * t_align / steps, "iterations_per_step": iters / steps}

    cores = max(1, min(avail, int(os.environ.get("NDT_BENCH_CPU_THREADS", str(avail)))))
    full = run(cores)
    if cores > 16:
        # no quota was visible, yet a 1-GPU box grants a 16-CPU share of the host: keep the faster of the two
        alt = run(16)
        if alt["value"] > full["value"]:
            full, cores = alt, 16
    eight = run(min(8, avail)) if cores != min(8, avail) else full
    return {"value": full["value"], "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": "%d full steps (oracle grid build + align) of the same C3 workload in %.1f s on %d OpenMP "
                      "threads (the grid build is single-threaded like the reference's)" % (full["steps"], full["seconds"], cores),
            "build_flags": flags, "nproc": nproc, "cpus_usable": avail,
            "ms_per_step": full["ms_per_step"], "ms_build": full["ms_build"], "ms_align": full["ms_align"],
            "iterations_per_step": full["iterations_per_step"],
            "threads_8": {k: eight[k] for k in ("value", "threads", "steps", "ms_per_step", "ms_build", "ms_align")}}


BUILD_BYTES = lambda n_tgt, v: n_tgt * 60.0 + v * 52.0  # noqa: E731  SURVEY.md section 8(d), B_build


def main():
    args = parse()
    pkg = ge.load_package()   # imports only: the HIP library is loaded on first use, in the ranks
    R = pkg.ranks
    # NDT_BENCH_FORCE_DIST=1: run the multi-rank code path (board, every reducer, watchdog) even with
    # one rank -- the only way to execute the RCCL leg end to end on a 1-GPU box
    force_dist = os.environ.get("NDT_BENCH_FORCE_DIST", "0") == "1"
    w = R.env_world()
    if w is None and (args.gpus > 1 or force_dist):
        # plain `python bench.py --gpus N`: this process is the parent of N ranks and touches no GPU
        limit = float(os.environ.get("NDT_BENCH_LAUNCH_TIMEOUT", "1500"))
        raise SystemExit(R.launch(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=limit))
    rank, local_rank, world = w if w is not None else (0, 0, 1)
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # rehearsal knob for a 1-GPU box: every rank on device 0 (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("NDT_BENCH_SINGLE_DEVICE", "0") == "1"
    rehearsal_tuning = {}
    if rehearsal:
        local_rank = 0
        if world > 2:
            # The build's fast paths wait INSIDE a kernel for sibling blocks (every tile of k_bucket_pass / k_sort_pass for
            # every other one) and need all of them resident: three or more ranks' builds on ONE device do not fit its 256
            # compute units together, each waits out its 50 ms time-out and falls back (0.2-0.6 s per step with four
            # ranks, profiles/r03_4on1_rehearsal.txt).  A rehearsal of that shape takes the launch-per-phase passes, which
            # never wait inside a kernel.  One rank per device -- the deployment -- is not affected.
            rehearsal_tuning = {"bucket_build": 0, "fused_sort": 0}   # (ndt_set_tuning below: the library reads no such variables)
    # stdout carries ONE JSON line and nothing else: native libraries write there too (librccl prints
    # "RCCL version : ..." on its first communicator), so file descriptor 1 is pointed at stderr for the
    # life of the rank and the line goes out through a private copy of the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    hip = R.Hip()
    # LOCAL_RANK is the device ordinal; a rank whose launcher narrowed the visible devices to one uses device 0
    # and says so in config.launch; LOCAL_RANK beyond the visible devices otherwise ends the rank with one clear line
    local_rank, device_note = R.pick_device(local_rank, world, hip.device_count(), rehearsal)
    hip.set_device(local_rank)
    multi = world > 1 or force_dist
    board = R.Board(R.board_path(), rank, world) if multi else None
    if board is not None:   # (any rank's note: rank 0 prints the line, the narrowed rank may be another one)
        device_note = next((b.decode() for b in board.allgather((device_note or "").encode()[:250]) if b), None)

    pkg = ge.load_package()
    if rehearsal_tuning:
        pkg.set_tuning(**rehearsal_tuning)
    S = pkg.synth
    cfg = S.config_c3()
    params = dict(resolution=float(cfg["resolution"]), step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    n_src_total = len(cfg["source"])
    b, c = pkg.shard_range(n_src_total, rank, world)

    # inputs resident in HBM before the timed region (SoA float32)
    tptr = [hip.upload(cfg["target"][:, a]) for a in range(3)]
    sptr = [hip.upload(cfg["source"][b:b + c, a]) for a in range(3)]
    hip.synchronize()

    # ranks that share one device (rehearsal): every engine keeps its pre-launched kernels on its own stream -- a
    # waiting kernel holds compute units the other rank's running kernel needs (include/ndt_hip.h, ndt_prelaunch)
    # (NDT_BENCH_REHEARSAL_AUTO=1 leaves the choice to NDT_PRELAUNCH_AUTO, which finds out by measurement)
    force_one = rehearsal and world > 1 and os.environ.get("NDT_BENCH_REHEARSAL_AUTO", "0") != "1"
    engine_kw = dict(params, prelaunch=pkg.PRELAUNCH_ONE_STREAM) if force_one else dict(params)
    ndt = pkg.NormalDistributionsTransform(device_id=local_rank, **engine_kw)

    # what a C++ caller hands over without any per-call work: raw pointers and guess.data()
    n_tgt = len(cfg["target"])
    guess_cm = pkg.ColMajor4f(cfg["guess"])

    current = {"mode": "none", "steps": 0}
    inject = os.environ.get("NDT_BENCH_INJECT_FAILURE")   # tests only: "<variant>" -> the last rank errors in its 3rd step

    step_opts = {"blocking_target": False}

    def step():
        if inject is not None and inject == current["mode"] and rank == world - 1:
            current["steps"] += 1
            if current["steps"] == 3:
                raise pkg.NdtError(-7, "injected failure (NDT_BENCH_INJECT_FAILURE)")
        t0 = time.perf_counter()
        # (the build is ENQUEUED -- the arrays stay put -- and the align's first evaluation goes onto the stream behind it;
        # t1 - t0 is the host's time in the call, the build itself is `ms_target_build_device`)
        if step_opts["blocking_target"]:   # (protocol_variants only: the blocking call of rounds 1-3)
            ndt.setInputTargetDevice(tptr[0], tptr[1], tptr[2], n_tgt)
        else:
            ndt.setInputTargetDeviceDeferred(tptr[0], tptr[1], tptr[2], n_tgt)
        t1 = time.perf_counter()
        ndt.setInputSourceDeviceView(sptr[0], sptr[1], sptr[2], c)
        ndt.align(guess_cm, return_transform=False)
        t2 = time.perf_counter()
        return ndt, t1 - t0, t2 - t1

    def fence():
        if multi:
            board.barrier()
        hip.synchronize()

    def timed_region(wake_steps=None):
        """W warm-up steps, then exactly K steps between two fences; max wall time over ranks.
        With several ranks an engine error (a reducer that stops answering) does not leave the other
        ranks stranded in a fence: the failing rank still walks through the same board calls and
        reports an infinite time, so every rank sees the variant as failed and moves on."""
        failed = None
        try:
            # (device wake-up, untimed, in front of the W warm-up steps: after an idle period -- the seconds of cloud
            # synthesis before this point -- an MI355X needs ~30 ms of work to be back at its clocks; W = 5 steps are
            # 2.6 ms.  tools/warmup_probe.py: steps 5..24 after 5 s of idling average 0.576 ms, steps 45.. 0.52.)
            for _ in range((WAKE_STEPS if wake_steps is None else wake_steps) + args.warmup):
                step()
        except pkg.NdtError as e:
            if not multi:
                raise
            failed = e
        fence()
        pre0 = ndt.prelaunchCounters()
        # (the K steps are ~14 ms in all: one cyclic garbage collection of the interpreter inside them -- the process holds the
        # synthetic clouds and every earlier leg's results -- would be a tenth of that; collected here, switched off until the fence)
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        iters = evals = reused = 0
        t_build = t_align = 0.0
        try:
            for _ in range(args.steps if failed is None else 0):
                e, tb, ta = step()
                iters += e.getFinalNumIteration()
                evals += e.getNumEvaluations()
                reused += e._raw.n_evaluations_reused
                t_build += tb
                t_align += ta
        except pkg.NdtError as e:
            if not multi:
                raise
            failed = e
        fence()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if failed is not None:
            print("rank %d: engine error inside the timed region (%s)" % (rank, failed), file=sys.stderr, flush=True)
            elapsed = float("inf")
        if multi:
            elapsed = board.allmax(elapsed)
        if elapsed == float("inf"):
            return None
        pre1 = ndt.prelaunchCounters()
        last = ndt.getResult()
        err_t, err_r = S.pose_error(last["T"], cfg["gt"])
        # the answer, bit for bit: final transform and Hessian of the last step, iterations and evaluations of all steps
        digest = hashlib.sha256(np.ascontiguousarray(last["T"]).tobytes() + np.ascontiguousarray(last["hessian"]).tobytes() +
                                struct.pack("<qq", iters, evals)).hexdigest()[:16]
        return dict(elapsed=elapsed, iters=iters, evals=evals, reused=reused, t_build=t_build, t_align=t_align,
                    prelaunched=pre1[0] - pre0[0], prelaunch_timeouts=pre1[2] - pre0[2], err_m=err_t, err_rad=err_r,
                    digest=digest, lost_row_retries=ndt.lostRowRetries(), auto_streams=ndt.autoStreamPlacement())

    unavailable = {}

    def init_reducer(mode):
        """Creates the engine's cross-rank reducer on every rank; False if any rank failed.  A transport the engine
        REFUSES because it cannot work here (NDT_ERR_UNSUPPORTED: no peer access, an area that cannot be opened) is
        recorded as unavailable with the engine's reason -- distinct from one that failed while running."""
        ok = 1.0
        try:
            if mode == "rccl":   # the engine's own RCCL communicator; rank 0's id travels over the board
                if rehearsal and world > 1:
                    raise pkg.NdtError(-9, "ranks share one device: RCCL refuses duplicate GPUs")
                ndt.commInitRccl(board.bcast(pkg.comm_unique_id() if rank == 0 else b""), rank, world)
            elif mode == "p2p":  # in-kernel peer-write exchange: every rank's slot array, opened by all
                handles = board.allgather(ndt.commP2pHandle())
                ndt.commInitP2p(b"".join(handles), rank, world)
            else:                # host-side sum through POSIX shared memory
                name = board.bcast(("/ndt_bench_%d" % os.getpid()).encode() if rank == 0 else b"").decode()
                shm_name["name"] = name
                ndt.commInitShm(name, rank, world)
        except (pkg.NdtError, AttributeError) as e:
            print("rank %d: %s reducer failed (%s)" % (rank, mode, e), file=sys.stderr, flush=True)
            ok = -1.0 if getattr(e, "code", 0) == -9 else 0.0
            unavailable[mode] = str(e)
        worst = board.allmin(ok)
        if worst <= 0.0:
            ndt.commDestroy()
            if worst < 0.0:
                # every rank reports the same variant the same way: the reason of the lowest rank that has one
                reasons = board.allgather(unavailable.get(mode, "").encode()[:250])
                unavailable[mode] = next((r.decode(errors="replace") for r in reasons if r), "unsupported")
            else:
                unavailable.pop(mode, None)
            return False
        ndt.setGlobalSourceSize(n_src_total)
        return True

    def instrumented(res, reduce_mode, variants):
        """Repeats the timed steps with HIP events around k_derivatives (engine's own stream)
        and assembles rank 0's JSON line for the timed result `res`."""
        ndt.enableKernelTiming(True)
        tm0 = ndt.getTiming()
        for _ in range(args.steps):
            step()
        r = ndt.getResult()
        tm1 = ndt.getTiming()
        ndt.enableKernelTiming(False)
        n_timed = tm1["n_timed_evals"] - tm0["n_timed_evals"]
        ms_kernel = (tm1["ms_eval_kernel_total"] - tm0["ms_eval_kernel_total"]) / max(n_timed, 1)
        ms_reduce = (tm1["ms_reduce_kernel_total"] - tm0["ms_reduce_kernel_total"]) / max(n_timed, 1)
        gi = ndt.getGridInfo()
        nbar = r["n_pairs"] / float(n_src_total)          # global pairs / global points (last evaluation)
        algo_bytes = c * ALGO_BYTES_PER_POINT(nbar)       # this rank's launch
        achieved = algo_bytes / (ms_kernel * 1e-3) / 1e9 if ms_kernel > 0 else 0.0
        # HBM bytes per launch from the PMC counters: NOT collected in this run (rocprofv3 --pmc is
        # a separate pass); the committed summary of that pass is quoted and labelled as such
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_k_derivatives.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # measured on the full 200k-point launch; a shard touches its share of it
                traffic = tj.get("hbm_bytes_per_launch") * (c / float(n_src_total))
                traffic_source = "static: profiles/traffic_k_derivatives.json (%s)" % tj.get("source", "rocprofv3 --pmc pass")
            except Exception:
                traffic = None
        final = ndt.getResult()
        err_t, err_r = S.pose_error(final["T"], cfg["gt"])
        if rank != 0:
            return None
        return {
            "metric": "ndt_align_iterations_per_sec", "value": 0.0, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 0.0, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3 scan-to-map: 200k-pt source into 1M-pt voxelised submap, 0.5 m voxel, DIRECT7, "
                                   "outlier 0.55, eps 1e-4, step 0.1, max 35 it; step = voxel-grid build + align",
                       "handoff": "clouds resident in HBM as SoA: the target's build enqueued by ndt_set_target_device_deferred "
                                  "and finished inside the step (the align's first evaluation is on the stream behind it), "
                                  "source viewed in place (ndt_set_source_device_view: setInputSource's shared_ptr contract)",
                       "n_source": n_src_total, "n_target": len(cfg["target"]), "voxels": int(gi["n_leaves"]),
                       "grid_cells": int(gi["n_cells"]), "mean_neighbors": nbar, "sharding": "source/%d" % world,
                       "reduce": reduce_mode, "reduce_variants": variants},
            "ms_target_build_device": gi["ms_build"],
            "device_wake_steps": WAKE_STEPS,   # untimed steps in front of the W warm-up steps of every timed region (clock ramp after idling)
            "final_error_vs_ground_truth": {"m": err_t, "rad": err_r},
            # `bound`: what the evidence shows limits this launch (DESIGN 4.1 / HISTORY 4.1: half of the 200 k-point launch is a
            # fixed latency chain, the rest VALU issue); `roof`: the roofline `achieved` / `peak` / `frac` are
            # priced against, as the contract defines them (algorithmic bytes over the HBM peak)
            "roofline": {"bound": "latency/valu", "roof": "hbm", "kernel": "k_derivatives", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "hbm_achieved_from_traffic": (traffic / (ms_kernel * 1e-3) / 1e9) if (traffic and ms_kernel > 0) else None,
                         "note": "achieved = algorithmic bytes / launch duration (SURVEY 8d) by HIP events ATTACHED TO THE DISPATCH "
                                 "(hipExtLaunchKernel: the kernel's own begin / end timestamps, the figure rocprofv3 reports; events "
                                 "recorded around the launch call, as until round 3, add ~2.4 us of dispatch), measured in an "
                                 "instrumented repeat that uses ordinary launches (k_derivatives<..., false>): the "
                                 "pre-launched variant (<..., true>) of the timed steps starts early and its duration "
                                 "includes waiting for the pose.  The table is cache-resident at this size, so real HBM "
                                 "traffic is far lower and the kernel is VALU/latency-bound, see DESIGN.md 4.1",
                         "algorithmic_bytes_per_launch": algo_bytes, "ms_per_launch": ms_kernel,
                         "ms_final_reduce": ms_reduce, "launches_timed": int(n_timed)},
            # the build as a group of kernels: SURVEY 8d's B_build over the device time of one build
            "roofline_build": {"bound": "hbm", "kernel": "target voxel-grid build (all launches of ndt_target.hip)",
                               "algorithmic_bytes": BUILD_BYTES(len(cfg["target"]), int(gi["n_leaves"])),
                               "ms_device": gi["ms_build"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "achieved": BUILD_BYTES(len(cfg["target"]), int(gi["n_leaves"])) / (gi["ms_build"] * 1e-3) / 1e9
                               if gi["ms_build"] > 0 else 0.0,
                               "frac": BUILD_BYTES(len(cfg["target"]), int(gi["n_leaves"])) / (gi["ms_build"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                               if gi["ms_build"] > 0 else 0.0},
        }

    def headline(out, res, reduce_mode):
        """Puts the timed result `res` into the JSON line (rank 0 only)."""
        if out is None:
            return
        elapsed, iters, evals = res["elapsed"], res["iters"], res["evals"]
        out.update({
            "value": iters / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_scan": 1e3 * elapsed / args.steps,
            # Key meanings as in rounds 1-3: ms_target_build is the BUILD (device time; the set-target call
            # only enqueues it since round 4 -- its host time is ms_target_build_enqueue), ms_align the align WITHOUT the
            # build that now finishes inside it (ms_align_incl_build is what the call takes)
            "ms_target_build": out.get("ms_target_build_device") or 0.0,
            "ms_target_build_enqueue": 1e3 * res["t_build"] / args.steps,
            "ms_align_incl_build": 1e3 * res["t_align"] / args.steps,
            "ms_align": max(1e3 * res["t_align"] / args.steps - (out.get("ms_target_build_device") or 0.0), 0.0),
            "iterations_per_align": iters / args.steps, "evaluations_per_align": evals / args.steps,
            # line-search requests at the pose of the evaluation before them, answered without a launch
            "evaluations_reused_per_align": res["reused"] / args.steps,
            # evaluations whose kernel was already waiting on the device for its pose (pre-launched while the
            # previous evaluation ran); the rest went through an ordinary launch
            "evaluations_prelaunched_per_align": res["prelaunched"] / args.steps,
            "prelaunch_timeouts": res["prelaunch_timeouts"],
            # (the build is enqueued by the set-target call and finishes inside the align: "align only" = the step
            # without the build's own duration)
            "align_only_iterations_per_sec": iters / max(elapsed - args.steps * 1e-3 * (out.get("ms_target_build_device") or 0.0), 1e-9),
            "evaluations_per_sec": evals / max(elapsed - args.steps * 1e-3 * (out.get("ms_target_build_device") or 0.0), 1e-9),
            "prelaunch_auto": {"one_stream": res["auto_streams"][0], "switches": res["auto_streams"][1]},
        })
        out["config"]["reduce"] = reduce_mode

    def host_cloud():
        """Not `value`: the same step with the clouds handed over as the drivers hold them -- host
        pcl::PointXYZI arrays (32 bytes per point) through ndt_set_target / ndt_set_source, i.e.
        the AoS->SoA repack, the PCIe upload, the build and the align (ref: run/pipeline.cpp:557-561)."""
        if world != 1 or os.environ.get("NDT_BENCH_HOST_CLOUD", "1") != "1":
            return None
        def xyzi(a):
            out = np.zeros((len(a), 8), np.float32)
            out[:, :3] = a
            out[:, 3] = 1.0
            return out
        tgt_h, src_h = xyzi(cfg["target"]), xyzi(cfg["source"])
        def hstep():
            t0 = time.perf_counter()
            ndt.setInputTarget(tgt_h)
            t1 = time.perf_counter()
            ndt.setInputSource(src_h)
            t2 = time.perf_counter()
            ndt.align(guess_cm, return_transform=False)
            return t1 - t0, t2 - t1, time.perf_counter() - t2
        for _ in range(3):
            hstep()
        k, iters, tt, ts, ta = 20, 0, 0.0, 0.0, 0.0
        rp_t = rp_s = bw = 0.0
        per_call = []
        t0 = time.perf_counter()
        for _ in range(k):
            a, b_, c_ = hstep()
            tt += a; ts += b_; ta += c_
            per_call.append((a, b_, c_))
            iters += ndt.getFinalNumIteration()
            ht = ndt.getHandoffTiming()
            rp_t += ht["target"]["ms_repack"]; rp_s += ht["source"]["ms_repack"]; bw += ht["ms_build_wait"]
        el = time.perf_counter() - t0
        pc = 1e3 * np.array(per_call)
        # one instrumented scan for what the engine can only time with events: the transfers and the build
        ndt.enableKernelTiming(True)
        hstep()
        ndt.wait()
        hi = ndt.getHandoffTiming()
        gi_h = ndt.getGridInfo()
        ndt.enableKernelTiming(False)
        # the same scan with the blocking hand-off of rounds 1-3 (ndt_set_handoff_mode), for the difference
        ndt.setHandoffMode(pkg.HANDOFF_SYNC)
        for _ in range(2):
            hstep()
        t1 = time.perf_counter()
        for _ in range(5):
            hstep()
        el_sync = time.perf_counter() - t1
        ndt.setHandoffMode(pkg.HANDOFF_ASYNC)
        # leave the engine as the timed steps use it
        ndt.setInputTargetDevice(tptr[0], tptr[1], tptr[2], n_tgt)
        ndt.setInputSourceDeviceView(sptr[0], sptr[1], sptr[2], c)
        # What the drivers pay on top when they call align() through pcl::Registration (RegisterCallback::registration;
        # align is not virtual): identity indices, output.resize, a per-point copy of the 32-byte source points and the
        # data[3] = 1 pass before computeTransformation -- reproduced in the API mock and timed by a C++ program at
        # the headline source size (tests/cpp/bench_registration_prework.cpp)
        pre = None
        exe = os.path.join(ROOT, "tests", "cpp", "bench_registration_prework")
        if os.path.exists(exe):
            try:
                import subprocess
                pr = subprocess.run([exe, str(n_src_total), str(n_tgt), "20"], capture_output=True, text=True, timeout=120)
                pre = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
                pre["what"] = ("host work of pcl::Registration::align around computeTransformation (API mock that repeats PCL "
                               "1.14's steps), same source size, synthetic room")
            except Exception as e:  # noqa: BLE001
                pre = {"error": str(e)}
        return {"what": "PCIe-inclusive: host PointXYZI (32 B/pt) clouds through ndt_set_target / ndt_set_source, then align; "
                        "asynchronous hand-off (the calls return once the caller's cloud is repacked into pinned staging; "
                        "transfer and build run behind, the target's under the source's repack; align waits for them)",
                "pcl_registration": pre,
                "value": iters / el, "unit": "iterations/s", "ms_scan": 1e3 * el / k, "ms_set_target": 1e3 * tt / k,
                "ms_set_source": 1e3 * ts / k, "ms_align": 1e3 * ta / k, "steps": k,
                # the repack runs on host threads of a shared box: a worker that is off its CPU when its piece is due shows
                # as a slow call and moves the MEANS above; the medians over the same steps say what an undisturbed scan costs
                "median": {"ms_scan": float(np.median(pc.sum(1))), "ms_set_target": float(np.median(pc[:, 0])),
                           "ms_set_source": float(np.median(pc[:, 1])), "ms_align": float(np.median(pc[:, 2])),
                           "ms_scan_max": float(pc.sum(1).max())},
                "ms_scan_blocking_handoff": 1e3 * el_sync / 5,
                "breakdown": {"ms_repack_target": rp_t / k, "ms_repack_source": rp_s / k,
                              "ms_align_waited_for_build": bw / k,
                              "ms_transfer_target": hi["target"]["ms_dma"], "ms_transfer_source": hi["source"]["ms_dma"],
                              "pcie_gb_per_s_target": hi["target"]["dma_gb_per_s"], "pcie_gb_per_s_source": hi["source"]["dma_gb_per_s"],
                              "bytes_read_target": hi["target"]["bytes_in"], "bytes_over_pcie_target": hi["target"]["bytes_dma"],
                              "ms_build_device": gi_h["ms_build"], "repack_threads": hi["target"]["threads"],
                              "cpu_budget": hi["cpu_budget"],
                              "note": "ms_transfer_*: device time from before a cloud's first chunk to behind its last (HIP "
                                      "events, one instrumented scan): chunks are pulled over PCIe as the repack finishes "
                                      "them, so this window contains the repack it overlaps"}}

    def scaling_probe():
        """Not part of `value`: the same engine on a source five times larger (the map's own
        1 M points seen from the scan pose, 2 cm noise), sharded like the headline workload --
        SURVEY 8e asks where sharding pays, and at 200 k points an evaluation is latency-bound."""
        if os.environ.get("NDT_BENCH_PROBE", "1") != "1":
            return None
        gt = cfg["gt"].astype(np.float64)
        Rinv, tinv = gt[:3, :3].T, -gt[:3, :3].T @ gt[:3, 3]
        t64 = cfg["target"].astype(np.float64)
        # element-wise on purpose: a BLAS matmul here leaves a pool of spinning worker threads
        # behind that competes with the engine's polling thread for the box's CPU share
        big = np.stack([t64[:, 0] * Rinv[r, 0] + t64[:, 1] * Rinv[r, 1] + t64[:, 2] * Rinv[r, 2] + tinv[r]
                        for r in range(3)], axis=1)
        big = (big + np.random.default_rng(5).normal(0.0, 0.02, big.shape)).astype(np.float32)
        nb = len(big)
        bb, cb = pkg.shard_range(nb, rank, world)
        bsrc = [hip.upload(big[bb:bb + cb, a]) for a in range(3)]
        hip.synchronize()
        if multi:
            ndt.setGlobalSourceSize(nb)

        def pstep():
            ndt.setInputTargetDevice(tptr[0], tptr[1], tptr[2], n_tgt)
            ndt.setInputSourceDevice(bsrc[0], bsrc[1], bsrc[2], cb)
            ndt.align(cfg["guess"])
            return ndt.getResult()

        for _ in range(3):   # first touches of the larger buffers (allocations, page mapping) stay outside
            pstep()
        fence()
        t0 = time.perf_counter()
        k, iters, evals = 5, 0, 0
        for _ in range(k):
            r = pstep()
            iters += r["iterations"]
            evals += r["n_evaluations"]
        fence()
        el = time.perf_counter() - t0
        if multi:
            el = board.allmax(el)
            ndt.setGlobalSourceSize(n_src_total)
        err_t, err_r = S.pose_error(r["T"], cfg["gt"])
        ndt.setInputSourceDeviceView(sptr[0], sptr[1], sptr[2], c)
        return {"workload": "same map, %d-point source (the map seen from the scan pose, 2 cm noise), sharded /%d" % (nb, world),
                "n_source": nb, "value": iters / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / k,
                "iterations_per_align": iters / k, "evaluations_per_align": evals / k,
                "final_error_vs_ground_truth": {"m": err_t, "rad": err_r}}

    variants = {}
    shm_name = {"name": None}

    def comm_record():
        v, path = pkg.comm_info()
        return {"version": v, "library": path}

    if not multi:
        res = timed_region()
        out = instrumented(res, "none", variants)
        headline(out, res, "none")
        # Not `value`: the same step on 48-byte voxel records (f64 mean, f32 inverse covariance; ndt_set_record_format):
        # three 16-byte fetches per neighbour instead of five.  The headline stays on the 80-byte f64 records, the
        # format the 1e-9 parity tests run on.
        # Everything below is reported BESIDE the headline and must never cost it: a side measurement that fails is
        # recorded as {"error": ...} under its own key and the line still goes out.
        def side(key, fn):
            try:
                v = fn()
                if out is not None and v is not None:
                    out[key] = v
            except Exception as e:  # noqa: BLE001
                print("bench: side measurement %s failed: %s: %s" % (key, type(e).__name__, e), file=sys.stderr, flush=True)
                if out is not None:
                    out[key] = {"error": "%s: %s" % (type(e).__name__, e)}

        def packed():
            if os.environ.get("NDT_BENCH_PACKED", "1") != "1":
                return None
            build_s = 1e-3 * ((out or {}).get("ms_target_build_device") or 0.0)
            ndt.setRecordFormat(pkg.RECORDS_PACKED48)
            try:
                pk = timed_region()
            finally:
                ndt.setRecordFormat(pkg.RECORDS_F64)
            return {"what": "same step, NDT_RECORDS_PACKED48 (48-byte voxel records: f64 mean + f32 inverse covariance)",
                    "value": pk["iters"] / pk["elapsed"], "unit": "iterations/s", "ms_per_step": 1e3 * pk["elapsed"] / args.steps,
                    "ms_target_build": 1e3 * pk["t_build"] / args.steps, "ms_align": 1e3 * pk["t_align"] / args.steps,
                    "iterations_per_align": pk["iters"] / args.steps, "evaluations_per_align": pk["evals"] / args.steps,
                    # compare THIS with the headline's ms_align / evaluations_per_align: the rounded table can move a
                    # line-search decision, and a step with fewer evaluations says nothing about the format
                    # (per evaluation: the step without the build's own duration)
                    "us_per_evaluation": 1e6 * max(pk["elapsed"] - args.steps * build_s, 0.0) / max(pk["evals"], 1),
                    "us_per_evaluation_f64_records": 1e6 * max(res["elapsed"] - args.steps * build_s, 0.0) / max(res["evals"], 1),
                    "final_error_vs_ground_truth": {"m": pk["err_m"], "rad": pk["err_rad"]}}

        def other_configs():
            # the other single-GPU configurations of BASELINE.json (C2 scan-to-scan, C5 replay), bounded to a few seconds
            if os.environ.get("NDT_BENCH_CONFIGS", "1") != "1":
                return None
            import bench_configs as BC
            cfgs = {}
            for name, fn in (("C2", lambda: BC.c2_block(pkg, hip)), ("C5", lambda: BC.c5_block(pkg))):
                try:
                    t_c = time.perf_counter()
                    cfgs[name] = fn()
                    cfgs[name]["seconds"] = time.perf_counter() - t_c
                except Exception as e:  # noqa: BLE001
                    cfgs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            return cfgs

        def protocol_variants():
            """Not `value`: the headline step under the measurement protocols of earlier rounds, so that a change of the
            headline between rounds splits into methodology and engine: without the untimed device wake-up
            steps in front of the warm-up (rounds 1-3; NDT_BENCH_WAKE_STEPS=0), and with the BLOCKING ndt_set_target_device
            in the step (rounds 1-3: the build awaited inside the set-target call), and both."""
            if os.environ.get("NDT_BENCH_PROTOCOLS", "1") != "1":
                return None
            v = {"what": "same C3 step, K = %d steps after W = %d warm-up steps" % (args.steps, args.warmup),
                 "headline": {"ms_per_step": 1e3 * res["elapsed"] / args.steps, "device_wake_steps": WAKE_STEPS, "set_target": "deferred"}}
            time.sleep(2.0)   # (an idle pause like the one in front of the headline's region: cloud synthesis took seconds)
            r0 = timed_region(wake_steps=0)
            v["no_wake_steps"] = {"ms_per_step": 1e3 * r0["elapsed"] / args.steps, "device_wake_steps": 0, "set_target": "deferred"}
            step_opts["blocking_target"] = True
            try:
                rb = timed_region()
                time.sleep(2.0)
                rb0 = timed_region(wake_steps=0)
            finally:
                step_opts["blocking_target"] = False
            v["blocking_set_target"] = {"ms_per_step": 1e3 * rb["elapsed"] / args.steps, "device_wake_steps": WAKE_STEPS, "set_target": "blocking",
                                        "ms_set_target": 1e3 * rb["t_build"] / args.steps, "ms_align": 1e3 * rb["t_align"] / args.steps}
            v["rounds_1_to_3_protocol"] = {"ms_per_step": 1e3 * rb0["elapsed"] / args.steps, "device_wake_steps": 0, "set_target": "blocking"}
            return v

        def cadence():
            """Not `value`: the headline step as a driver at the reference's keyframe rate sees it -- one step every 100 ms
            (10 Hz) and every 50 ms (20 Hz), the device idle in between, and the first step after 5 s of idling; then the
            same with the engine's idle-time heartbeat on (ndt_set_keepwarm, 1 ms period; default off).  Medians."""
            if os.environ.get("NDT_BENCH_CADENCE", "1") != "1":
                return None

            def clocks():
                # the clocks the driver reports right now (MHz: shader, memory, fabric, SoC), where the box lets an ordinary user read them
                out = {}
                try:
                    import glob
                    for name in ("sclk", "mclk", "fclk", "socclk"):
                        best = None
                        for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_" + name):
                            for ln in open(f).read().splitlines():
                                if ln.strip().endswith("*"):
                                    mhz = int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
                                    best = mhz if best is None else max(best, mhz)
                        if best is not None:
                            out[name] = best
                except Exception:  # noqa: BLE001
                    pass
                return out

            def sclk():
                return clocks().get("sclk")

            def one():
                t0 = time.perf_counter()
                step()
                return 1e3 * (time.perf_counter() - t0)

            def paced(period, n, spin=False):
                t, clk = [], []
                for _ in range(n):
                    if spin:   # the HOST thread stays awake (busy-wait), only the device idles
                        t_end = time.perf_counter() + period
                        while time.perf_counter() < t_end:
                            pass
                    else:
                        time.sleep(period)
                    clk.append(sclk())
                    t.append(one())
                clk = [c_ for c_ in clk if c_ is not None]
                return float(np.median(t)), (float(np.median(clk)) if clk else None)

            def block():
                for _ in range(10):
                    one()
                busy = float(np.median([one() for _ in range(20)]))
                m10, c10 = paced(0.1, 25)
                m20, c20 = paced(0.05, 30)
                m10s, _ = paced(0.1, 15, spin=True)
                time.sleep(5.0)
                c5 = sclk()
                call = clocks()
                first = one()
                return {"ms_step_back_to_back": busy, "ms_step_10hz": m10, "ms_step_20hz": m20, "ms_first_step_after_5s_idle": first,
                        # the same 10 Hz cadence with the calling thread busy-waiting instead of sleeping: what of the
                        # difference to back-to-back is the HOST core waking up, not the device
                        "ms_step_10hz_host_thread_spinning": m10s,
                        "sclk_mhz_before_a_10hz_step": c10, "sclk_mhz_before_a_20hz_step": c20, "sclk_mhz_after_5s_idle": c5,
                        "clocks_mhz_after_5s_idle": call}

            v = block()
            ndt.setKeepWarm(1000)
            try:
                time.sleep(0.05)
                v["keepwarm_1ms"] = block()
                v["keepwarm_1ms"]["beats"] = ndt.keepWarm()[1]
            finally:
                ndt.setKeepWarm(0)
            return v

        side("protocol_variants", protocol_variants)
        side("cadence", cadence)
        side("packed_records", packed)
        side("host_cloud", host_cloud)
        side("configs", other_configs)
        side("scaling_probe", scaling_probe)
        if out is not None:
            out["config"]["rccl"] = comm_record()
        if not args.no_cpu_baseline:
            side("cpu_baseline", lambda: cpu_baseline(cfg, params, args.cpu_seconds))
    else:
        # Every transport of the 256-byte evaluation sum is timed back to back on the same workload
        # (SURVEY 8e): pinned-host partials summed through shared memory, the in-kernel peer-write
        # exchange over xGMI, and an RCCL all-reduce on the engine's stream.  All are reported under
        # config.reduce_variants (RCCL always, with the communicator's own rank count); the fastest is
        # the headline and is named in config.reduce; NDT_BENCH_REDUCE=rccl|p2p|shm pins one.  RCCL
        # comes last and under a watchdog: every earlier pass is complete (JSON line assembled) before
        # it is touched, so a stuck communicator cannot cost the whole measurement -- the watchdog
        # prints that line and ends the rank with status 4.
        forced = os.environ.get("NDT_BENCH_REDUCE")
        modes = [forced] if forced in ("rccl", "shm", "p2p") else ["shm", "p2p", "rccl"]
        out, best = None, None
        state = {"out": None, "mode": None}
        digests = {}
        # Every variant runs under a watchdog with its share of the wall budget (ranks.variant_budget: the driver allows
        # 600 s for the whole command): a transport that hangs costs its own share, not the measurement -- the watchdog
        # prints the line assembled from the variants that did finish and ends the rank with status 4.
        budget = R.variant_budget(len(modes))

        def on_timeout():
            """A variant did not finish inside its budget: print what the earlier variants measured (this one marked as
            timed out) and end EVERY rank with a non-zero status, so the hang shows in the run record instead of
            passing as rc 0."""
            mode = state["mode"]
            print("rank %d: reduce variant %s exceeded its wall budget of %.0f s" % (rank, mode, budget), file=sys.stderr, flush=True)
            if rank == 0 and state["out"] is not None:
                state["out"]["config"]["reduce_variants"][mode] = "timed out"
                state["out"]["reduce_failed"] = mode
                os.write(json_fd, (json.dumps(state["out"]) + "\n").encode())
            if shm_name["name"]:
                try:
                    os.unlink("/dev/shm" + shm_name["name"])
                except OSError:
                    pass
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(4)

        for mode in modes:
            state["mode"] = mode
            dog = threading.Timer(float(os.environ.get("NDT_BENCH_RCCL_TIMEOUT", budget)) if mode == "rccl" else budget, on_timeout)
            dog.daemon = True
            dog.start()
            if not init_reducer(mode):
                if mode in unavailable:
                    variants[mode] = {"unavailable": unavailable[mode]}
                else:
                    variants[mode] = None
                    if out is not None:
                        out.setdefault("reduce_failed", mode)
            else:
                n_comm = ndt.commRankCount()
                current["mode"], current["steps"] = mode, 0
                integrity = None
                if mode == "p2p":
                    # Before anything is timed over the peer-write exchange: 10 000 lock-step rounds of patterned slots
                    # through the exchange areas (ndt_comm_p2p_selftest) -- the direct test of the 16-byte single-copy
                    # assumption the in-kernel exchange rests on, the first time it crosses a device boundary.
                    try:
                        it = ndt.commP2pSelftest(int(os.environ.get("NDT_BENCH_P2P_SELFTEST", "10000")))
                    except pkg.NdtError as e:
                        it = {"error": str(e), "torn": -1, "missed": -1, "rounds": 0, "longest_us": 0.0}
                    board.barrier()
                    worst = board.allmax(float(max(it["torn"], 0) + max(it["missed"], 0) + (1 if "error" in it else 0)))
                    integrity = dict(it, ranks_clean=bool(worst == 0.0))
                    if worst != 0.0:
                        variants[mode] = {"failed": "slot integrity pass", "slot_integrity": integrity}
                        if out is not None:
                            out.setdefault("reduce_failed", mode)
                        ndt.commDestroy()
                        board.barrier()
                        dog.cancel()
                        continue
                res = timed_region()
                if res is None:      # an engine error on some rank: the variant is reported as failed, the others stand
                    variants[mode] = "failed"
                    if out is not None:
                        out.setdefault("reduce_failed", mode)
                    ndt.commDestroy()
                else:
                    # A transport that sums wrongly shows up in the answer.  (1) the same scan must land on the same pose;
                    # (2) every rank runs the same host loop on the same sums: final transform, Hessian, iterations and
                    # evaluations must agree BIT FOR BIT across the ranks; (3) shm and p2p add the same rows in the same
                    # rank order: their answers must be identical too (RCCL's ring adds in another order: not required).
                    sane = bool(board.allmax(0.0 if res["err_m"] < 0.05 else 1.0) == 0.0)
                    all_digests = [d.decode() for d in board.allgather(res["digest"].encode())]
                    ranks_agree = len(set(all_digests)) == 1
                    digests[mode] = all_digests[0] if ranks_agree else None
                    variants[mode] = {"value": res["iters"] / res["elapsed"], "ms_per_step": 1e3 * res["elapsed"] / args.steps,
                                      "ranks": n_comm, "evaluations_prelaunched_per_align": res["prelaunched"] / args.steps,
                                      "iterations_per_align": res["iters"] / args.steps, "final_error_m": res["err_m"],
                                      "answer_digest": digests[mode], "ranks_bit_identical": ranks_agree,
                                      "lost_row_retries": res["lost_row_retries"],
                                      # (NDT_PRELAUNCH_AUTO's placement of waiting kernels on this rank: 1 = one stream; switches)
                                      "auto_one_stream": res["auto_streams"][0], "auto_switches": res["auto_streams"][1]}
                    if not sane:
                        variants[mode]["suspect"] = "final pose more than 5 cm from ground truth: not eligible as the headline"
                    elif not ranks_agree:
                        sane = False
                        variants[mode]["suspect"] = "the ranks' answers differ (%s): not eligible as the headline" % ",".join(all_digests)
                    elif mode in ("shm", "p2p") and all(digests.get(m) for m in ("shm", "p2p")) and digests["shm"] != digests["p2p"]:
                        sane = False
                        variants[mode]["suspect"] = ("answer differs from the shm variant's (%s vs %s): the two add the same rows in "
                                                     "the same order and must agree bit for bit" % (digests["p2p"], digests["shm"]))
                    if mode == "rccl":
                        variants[mode]["ncclCommCount"] = n_comm
                    if integrity is not None:
                        variants[mode]["slot_integrity"] = integrity
                    # What the sum itself costs under this transport and how long every rank's kernel runs -- a flat
                    # scaling curve then reads as "reduce-bound" or "fixed-chain-bound" from this one run:
                    # an instrumented repeat (ordinary launches, HIP events attached to the dispatch; host-side transports:
                    # wall time of the cross-rank sum; peer-write: the exchange's duration counted inside the kernel)
                    try:
                        if mode == "p2p":
                            ndt.commP2pStats(reset=True)
                        ndt.enableKernelTiming(True)
                        tq0 = ndt.getTiming()
                        for _ in range(max(2, args.steps // 4)):
                            step()
                        tq1 = ndt.getTiming()
                        ndt.enableKernelTiming(False)
                        nq = max(tq1["n_timed_evals"] - tq0["n_timed_evals"], 1)
                        k_us = 1e3 * (tq1["ms_eval_kernel_total"] - tq0["ms_eval_kernel_total"]) / nq
                        if mode == "p2p":
                            ps = ndt.commP2pStats()
                            r_us, r_max = ps["mean_us"], ps["max_us"]
                        else:
                            r_us, r_max = 1e3 * (tq1["ms_reduce_kernel_total"] - tq0["ms_reduce_kernel_total"]) / nq, None
                        per_rank = [json.loads(b.decode()) for b in board.allgather(json.dumps([k_us, r_us]).encode())]
                        variants[mode]["k_derivatives_us_per_rank"] = [round(p_[0], 2) for p_ in per_rank]
                        variants[mode]["reduce_us_per_evaluation_per_rank"] = [round(p_[1], 2) for p_ in per_rank]
                        variants[mode]["reduce_us_per_evaluation"] = max(p_[1] for p_ in per_rank)
                        if r_max is not None:
                            variants[mode]["reduce_us_longest_on_rank0"] = r_max
                        variants[mode]["reduce_note"] = ("local sum in hand -> global sum in hand: " +
                                                         {"shm": "host wall time of the shared-memory sum", "p2p": "own row published -> every rank's row read, inside the kernel's final sum",
                                                          "rccl": "ncclAllReduce + read-back on the engine's stream (host wall time)"}[mode])
                    except Exception as e:  # noqa: BLE001  (instrumentation must never cost the variant)
                        variants[mode]["reduce_probe_error"] = "%s: %s" % (type(e).__name__, e)
                        try:
                            ndt.enableKernelTiming(False)
                        except Exception:  # noqa: BLE001
                            pass
                    if out is None and sane:
                        out = instrumented(res, mode, variants)
                        if rank != 0:
                            out = {"config": {"reduce_variants": variants}}  # placeholder: only rank 0 prints
                        probe = scaling_probe()
                        if probe is not None:
                            probe["reduce"] = mode
                            out["scaling_probe"] = probe
                    if sane and (best is None or res["elapsed"] < best["elapsed"]):
                        best = res
                        headline(out, res, mode)
                    ndt.commDestroy()
            if rank == 0:
                state["out"] = out
            board.barrier()
            dog.cancel()
        if best is None:
            raise SystemExit("no cross-rank reducer could be created")
        if out is not None and "config" in out:
            out["config"]["rccl"] = comm_record()
            out["config"]["launch"] = ("self-launched ranks" if "NDT_RANKS_BOARD" in os.environ else "external launcher") + \
                                      ("; " + device_note if device_note else "")
            out["config"]["variant_wall_budget_s"] = budget
        if rank != 0:
            out = None
        board.barrier()
    ndt.close()
    hip.free_all()
    if board is not None:
        board.close()
    if out is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
