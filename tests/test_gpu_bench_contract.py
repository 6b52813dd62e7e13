"""bench.py keeps its contract: one JSON line on stdout with the BASELINE.json metric, the
whole-job value, and the roofline / cpu_baseline objects (short run: 2 steps, bounded CPU sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    env = dict(os.environ, NDT_BENCH_PROBE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "iterations" in d["metric"] and "iterations/sec" in base["metric"]
    assert d["unit"] == "iterations/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["iterations_per_align"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "latency/valu" and rf["roof"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["traffic"] is None or (rf["traffic"] > 0 and rf["traffic_source"].startswith("static: profiles/"))
    rb = d["roofline_build"]
    assert rb["bound"] == "hbm" and rb["algorithmic_bytes"] > 6e7 and 0 < rb["frac"] < 1
    assert abs(rb["achieved"] - rb["algorithmic_bytes"] / (rb["ms_device"] * 1e-3) / 1e9) < 1e-6 * rb["achieved"]
    hc = d["host_cloud"]
    assert hc["unit"] == "iterations/s" and 0 < hc["value"] < d["value"] and hc["ms_scan"] > d["ms_per_step"]
    assert 0 < hc["median"]["ms_scan"] <= hc["median"]["ms_scan_max"] and hc["median"]["ms_set_source"] > 0
    bd = hc["breakdown"]
    assert bd["ms_repack_target"] > 0 and bd["ms_repack_source"] > 0 and bd["ms_align_waited_for_build"] >= 0
    assert bd["ms_transfer_target"] > 0 and bd["pcie_gb_per_s_target"] > 1 and bd["bytes_over_pcie_target"] == 12 * d["config"]["n_target"]
    assert bd["bytes_read_target"] == 32 * d["config"]["n_target"] and bd["repack_threads"] >= 1 and bd["ms_build_device"] > 0
    # the hand-off the unchanged drivers use (run/pipeline.cpp:554-561): returns before the device has the cloud
    assert hc["ms_set_target"] < hc["ms_scan_blocking_handoff"] and hc["ms_scan_blocking_handoff"] > 0
    # the other single-GPU configurations on the same line: C2 scan-to-scan, C5 replay (BASELINE.json configs[1], [4])
    c2 = d["configs"]["C2"]
    assert "error" not in c2 and c2["unit"] == "iterations/s" and c2["value"] > 0 and c2["ms_scan"] > 0 and c2["us_per_evaluation"] > 0
    assert 0 < c2["roofline"]["frac"] < 1 and c2["roofline"]["kernel"] == "k_derivatives" and c2["final_error_vs_ground_truth"]["m"] < 0.05
    c5 = d["configs"]["C5"]
    assert "error" not in c5
    for leg in ("ndt_host_clouds", "ndt_device_keyframes", "svn_k20"):
        assert c5[leg]["hz"] > 0 and c5[leg]["ms_per_frame"] > 0 and "final_error_vs_ground_truth" in c5[leg], leg
        assert 0 < c5[leg]["ms_engine_per_frame"] <= c5[leg]["ms_per_frame"] and c5[leg]["hz_engine"] >= c5[leg]["hz"] * 0.99, leg
    assert c5["ndt_host_clouds"]["max_error_m"] < 0.05 and c5["ndt_device_keyframes"]["max_error_m"] < 0.05
    st = c5["svn_k20"]["stage1"]
    assert st["poses_per_launch"] == 20 and st["ms_per_launch"] > 0 and 0 < st["frac"] < 1 and st["launches_timed"] >= 1
    assert c5["svn_k20"]["mean_error_m"] < c5["svn_k20"]["mean_prior_error_m"]
    assert c5["svn_k20"]["svn_iterations_per_sec"] > 100 * c5["svn_k20"]["reference_log"]["svn_iterations_per_sec"]
    # key meanings of rounds 1-3 kept: the build's own time under ms_target_build, the call's under ..._enqueue
    assert d["ms_target_build"] == d["ms_target_build_device"] > 0.02 and 0 < d["ms_target_build_enqueue"] < d["ms_target_build"]
    assert abs(d["ms_align"] + d["ms_target_build"] - d["ms_align_incl_build"]) < 1e-9 and d["ms_align"] > 0
    # the headline under the protocols of earlier rounds, and at the drivers' cadence
    pv = d["protocol_variants"]
    for k in ("headline", "no_wake_steps", "blocking_set_target", "rounds_1_to_3_protocol"):
        assert pv[k]["ms_per_step"] > 0 and pv[k]["set_target"] in ("deferred", "blocking"), k
    assert pv["no_wake_steps"]["device_wake_steps"] == 0 and pv["headline"]["device_wake_steps"] == d["device_wake_steps"]
    cd = d["cadence"]
    for blk in (cd, cd["keepwarm_1ms"]):
        assert 0 < blk["ms_step_back_to_back"] and 0 < blk["ms_step_10hz"] and 0 < blk["ms_step_20hz"] and 0 < blk["ms_first_step_after_5s_idle"]
    assert cd["keepwarm_1ms"]["beats"] > 1000      # (one per millisecond of idling: > 9 s of pauses)
    assert d["evaluations_reused_per_align"] >= 0 and d["config"]["rccl"]["version"] > 20000
    assert 0 <= d["evaluations_prelaunched_per_align"] < d["evaluations_per_align"] and d["prelaunch_timeouts"] == 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "iterations/s" and cb["sample"]
    assert cb["nproc"] >= cb["cores"] and cb["cpus_usable"] >= 1 and "-O" in cb["build_flags"] and cb["threads_8"]["threads"] <= 8
    assert d["final_error_vs_ground_truth"]["m"] < 0.05


def test_bench_gpus_2_from_a_plain_invocation():
    """`python bench.py --gpus 2` exactly as the driver invokes N = 1 (no launcher, no WORLD_SIZE): the
    process starts its two ranks itself, torch-free; here both ranks share the box's one device
    (NDT_BENCH_SINGLE_DEVICE=1), so RCCL is skipped by name (it refuses duplicate GPUs) while the
    shared-memory and the peer-write transports are both timed.  One JSON line, rc 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env.update(NDT_BENCH_PROBE="0", NDT_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    v = d["config"]["reduce_variants"]
    # RCCL cannot work with two ranks on one device: reported as unavailable with the reason, not as a failure
    assert set(v) == {"shm", "p2p", "rccl"} and "duplicate" in v["rccl"]["unavailable"] and "reduce_failed" not in d
    assert v["shm"]["value"] > 0 and v["p2p"]["value"] > 0 and v["shm"]["ranks"] == 2 and v["p2p"]["ranks"] == 2
    # every rank runs the same host loop on the same sums, and shm / p2p add the same rows in the same order
    assert v["shm"]["ranks_bit_identical"] and v["p2p"]["ranks_bit_identical"] and "suspect" not in v["shm"] and "suspect" not in v["p2p"]
    assert v["shm"]["answer_digest"] == v["p2p"]["answer_digest"]
    assert d["config"]["reduce"] in ("shm", "p2p") and d["config"]["launch"].startswith("self-launched ranks; rehearsal")
    assert d["config"]["variant_wall_budget_s"] > 0
    assert d["config"]["sharding"] == "source/2" and d["final_error_vs_ground_truth"]["m"] < 0.05


def test_bench_four_ranks_on_one_device_rehearsal():
    """The shape that aborted in round 3 (gpurun_out/r03/bench_4on1.err: three ranks lost a partial row while four
    processes time-sliced one device): four ranks on the box's one device, launch-per-phase build, run ONCE.  rc 0,
    one line, every rank on the same pose bit for bit.  A lost row is no longer an error either (it is re-evaluated
    through the ticketed sum) -- the line says how often that happened."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env.update(NDT_BENCH_PROBE="0", NDT_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    v = d["config"]["reduce_variants"]
    assert d["n_gpus"] == 4 and d["config"]["sharding"] == "source/4" and d["final_error_vs_ground_truth"]["m"] < 0.05
    for m in ("shm", "p2p"):
        assert v[m]["ranks"] == 4 and v[m]["ranks_bit_identical"] and "suspect" not in v[m] and v[m]["lost_row_retries"] >= 0
    assert v["shm"]["answer_digest"] == v["p2p"]["answer_digest"]
    assert "hand-off lost" not in r.stderr


def test_prelaunch_auto_notices_a_second_engine_on_the_device():
    """Two ranks on ONE device with the stream placement of the waiting kernels left to NDT_PRELAUNCH_AUTO: on the other
    stream a waiting kernel holds its compute units for the whole evaluation of its predecessor -- units the other rank's
    kernel needs (5 ms per step measured, against 0.59 with one stream; profiles/r04_prelaunch_auto_placement.txt).  AUTO
    probes the other placement every 32nd align and must have switched both ranks to one stream by the timed steps."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env.update(NDT_BENCH_PROBE="0", NDT_BENCH_SINGLE_DEVICE="1", NDT_BENCH_REHEARSAL_AUTO="1", NDT_BENCH_REDUCE="shm")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "70",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    v = d["config"]["reduce_variants"]["shm"]
    assert v["auto_one_stream"] == 1 and v["auto_switches"] >= 1 and v["ranks_bit_identical"]
    assert d["ms_per_step"] < 2.0, d["ms_per_step"]


def test_bench_forced_distributed_one_rank_runs_rccl():
    """NDT_BENCH_FORCE_DIST=1: the whole multi-rank path with ONE rank -- the only way to execute the RCCL
    leg (ncclCommInitRank + one ncclAllReduce per evaluation) on a 1-GPU box.  The line reports the
    communicator's own rank count and which librccl served it: the process holds no torch, so it is ROCm's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env.update(NDT_BENCH_PROBE="0", NDT_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    v = d["config"]["reduce_variants"]
    assert v["rccl"]["ncclCommCount"] == 1 and v["rccl"]["value"] > 0 and v["shm"]["value"] > 0 and v["p2p"]["value"] > 0
    assert "/opt/rocm" in d["config"]["rccl"]["library"] and "torch" not in d["config"]["rccl"]["library"]


def test_bench_survives_a_reduce_variant_that_fails_mid_run():
    """One rank errors inside the peer-write variant's timed region (NDT_BENCH_INJECT_FAILURE): the other rank's
    all-reduce times out (NDT_COMM_TIMEOUT_S), both walk through the same fences, the variant is reported as
    failed and the shared-memory measurement taken before it stands -- one JSON line, rc 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env.update(NDT_BENCH_PROBE="0", NDT_BENCH_SINGLE_DEVICE="1", NDT_BENCH_INJECT_FAILURE="p2p", NDT_COMM_TIMEOUT_S="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    v = d["config"]["reduce_variants"]
    assert v["p2p"] == "failed" and d["reduce_failed"] == "p2p" and v["shm"]["value"] > 0 and d["config"]["reduce"] == "shm"
    assert v["shm"]["final_error_m"] < 0.05 and "suspect" not in v["shm"]
    assert "injected failure" in r.stderr and "timed out" in r.stderr
