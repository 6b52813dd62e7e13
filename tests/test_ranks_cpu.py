"""The torch-free rank plumbing of bench.py (slam-sam_amd/ranks.py), on the CPU: the parent launcher,
the shared-memory slot board (all-gather / broadcast / max, stale files, an externally launched job),
and bench.py's own `--gpus N` entry as the driver invokes it."""
import json
import os
import struct
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r'''
    import os, struct, sys
    sys.path.insert(0, %r)
    import __graft_entry__ as ge
    R = ge.load_package().ranks
    rank, local_rank, world = R.env_world()
    b = R.Board(R.board_path(), rank, world, timeout=60)
    got = b.allgather(("r%%d" %% rank).encode())
    assert got == [("r%%d" %% r).encode() for r in range(world)], got
    blob = b.bcast(os.urandom(128) if rank == 0 else b"")
    assert len(blob) == 128
    echo = b.allgather(blob)                      # every rank received the same 128 bytes
    assert all(e == blob for e in echo)
    for it in range(200):                         # many rounds: the two payload generations never mix
        vals = b.allgather(struct.pack("<ii", rank, it))
        assert [struct.unpack("<ii", v) for v in vals] == [(r, it) for r in range(world)]
    mx, mn = b.allmax(1.5 + rank), b.allmin(1.5 + rank)
    if os.environ.get("CHILD_FAIL_RANK") == str(rank):
        sys.exit(7)
    b.barrier()
    if rank == 0:
        print('{"ok": true, "world": %%d, "max": %%.1f, "min": %%.1f}' %% (world, mx, mn), flush=True)
    b.close()
''') % ROOT


def _pkg():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    return ge.load_package()


def test_parent_launches_ranks_and_relays_rank0(tmp_path, capfd):
    R = _pkg().ranks
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    p = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; R = ge.load_package().ranks; "
                        "sys.exit(R.launch(4, [sys.executable, %r], timeout=120))" % (ROOT, str(script))],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE line, rank 0's
    assert json.loads(lines[0]) == {"ok": True, "world": 4, "max": 4.5, "min": 1.5}
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("ndt_board_")]   # the parent cleaned up
    # a failing rank fails the job, and the survivors (who would wait for it for ever) are stopped
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; R = ge.load_package().ranks; "
                        "sys.exit(R.launch(3, [sys.executable, %r], timeout=120))" % (ROOT, str(script))],
                       env=dict(env, CHILD_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 7 and time.monotonic() - t0 < 60
    assert "rank 1 exited with 7" in p.stderr


def test_externally_launched_ranks_meet_on_a_derived_board_despite_a_stale_file(tmp_path):
    """What torch.distributed.run provides: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT and a common
    parent.  The ranks derive the same board name; a file of that name left by a crashed run (magic set,
    large sequence words) is not used: only a live rank 0 answers the attach handshake."""
    R = _pkg().ranks
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    port = "29%03d" % (os.getpid() % 1000)
    stale = "/dev/shm/ndt_board_%d_%s" % (os.getpid(), port)
    with open(stale, "wb") as f:
        blob = bytearray(64 + 64 * (64 + 512))
        struct.pack_into("<QII", blob, 0, 0x4E44545F424F4152, 3, 1)
        for r in range(3):
            struct.pack_into("<Q", blob, 64 + r * 576, 10 ** 9)   # sequence words far ahead
        f.write(blob)
    env = {k: v for k, v in os.environ.items() if k != "NDT_RANKS_BOARD"}
    procs = []
    for r in (2, 1, 0):   # rank 0 last: the others meet the stale file first
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_PORT=port)))
        time.sleep(0.3)
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], [o[1][-500:] for o in outs]
    assert json.loads(outs[2][0].strip()) == {"ok": True, "world": 3, "max": 3.5, "min": 1.5}
    assert not os.path.exists(stale)   # rank 0 unlinked its board at close


def test_bench_gpus_2_is_runnable_as_the_driver_runs_it():
    """`python bench.py --gpus 2` with no WORLD_SIZE (the shape of the driver's N = 1 command): the
    process must start its ranks itself instead of refusing (a guaranteed rc != 0).  Here,
    without a GPU, each rank fails LOUDLY (no CPU fallback) and the parent reports it; on the GPU box the
    same invocation prints one JSON line (tests/test_gpu_bench_contract.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    env["NDT_BENCH_SINGLE_DEVICE"] = "1"
    env["HIP_VISIBLE_DEVICES"] = "-1" if os.path.exists("/dev/kfd") else env.get("HIP_VISIBLE_DEVICES", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "launch N>1 with" not in p.stderr
    assert "needs an MI355X" in p.stderr and "ranks.launch: rank" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]   # no line from a run that measured nothing


def test_rank_device_choice_and_wall_budget():
    """First contact with a multi-GPU node: one rank per device is the rule; a rank whose launcher
    narrowed the visible devices to one uses device 0 and says so; LOCAL_RANK beyond the visible devices otherwise
    is a launch error with one clear line.  Every reduce variant gets a share of the wall budget."""
    R = _pkg().ranks
    assert R.pick_device(5, 8, 8) == (5, None)
    dev, note = R.pick_device(5, 8, 1)
    assert dev == 0 and "ONE device" in note and "LOCAL_RANK 5" in note
    dev, note = R.pick_device(3, 4, 1, rehearsal=True)
    assert dev == 0 and "rehearsal" in note
    with pytest.raises(SystemExit) as ei:
        R.pick_device(5, 8, 4)
    assert "LOCAL_RANK 5" in str(ei.value) and "4 device(s)" in str(ei.value)
    with pytest.raises(SystemExit) as ei:
        R.pick_device(0, 1, 0)
    assert "no CPU fallback" in str(ei.value)
    old = {k: os.environ.pop(k, None) for k in ("NDT_BENCH_WALL_BUDGET", "NDT_BENCH_VARIANT_TIMEOUT")}
    try:
        assert R.variant_budget(3) == pytest.approx(105.0)          # 420 s over three variants + one share
        assert 3 * R.variant_budget(3) + 60 < 600                   # ... ends well inside the driver's 600 s
        assert R.variant_budget(1, total=100.0) == pytest.approx(50.0)
        os.environ["NDT_BENCH_VARIANT_TIMEOUT"] = "7"
        assert R.variant_budget(3) == 7.0
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_signalled_parent_takes_its_ranks_along(tmp_path):
    """`timeout -k 10 400 python bench.py --gpus 2` signals only the parent: SIGTERM is passed on to the ranks (each in
    a session of its own), the board file is unlinked; a parent that is KILLED takes them along too (PDEATHSIG)."""
    import signal
    child = tmp_path / "sleeper.py"
    child.write_text("import os, sys, time\nopen(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    parent_code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; R = ge.load_package().ranks; "
                   "sys.exit(R.launch(2, [sys.executable, %r, %r], timeout=100))")

    def pids(prefix):
        out = []
        for r in range(2):
            f = "%s.%d" % (prefix, r)
            t0 = time.monotonic()
            while not (os.path.exists(f) and open(f).read().strip()):
                assert time.monotonic() - t0 < 60
                time.sleep(0.02)
            out.append(int(open(f).read()))
        return out

    def gone(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return True
        try:   # a zombie whose parent (this test's grandchild) is gone is reaped by init; treat 'Z' as gone
            return open("/proc/%d/stat" % pid).read().split()[2] == "Z"
        except OSError:
            return True

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NDT_RANKS_BOARD")}
    for how in (signal.SIGTERM, signal.SIGKILL):
        prefix = str(tmp_path / ("pid%d" % how))
        p = subprocess.Popen([sys.executable, "-c", parent_code % (ROOT, str(child), prefix)], env=env)
        kids = pids(prefix)
        boards = [f for f in os.listdir("/dev/shm") if f.startswith("ndt_board_%d_" % p.pid)]
        os.kill(p.pid, how)
        p.wait(timeout=60)
        t0 = time.monotonic()
        while not all(gone(k) for k in kids):
            assert time.monotonic() - t0 < 30, "ranks survived their parent (%s)" % how
            time.sleep(0.05)
        if how == signal.SIGTERM:
            assert p.returncode == 128 + signal.SIGTERM
            assert not [f for f in boards if os.path.exists("/dev/shm/" + f)]


def test_bench_imports_no_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.cuda" not in src
