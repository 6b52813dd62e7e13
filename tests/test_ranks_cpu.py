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


def test_bench_imports_no_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.cuda" not in src
