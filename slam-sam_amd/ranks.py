"""One process per GPU on ONE node, without torch: rank launch, a shared-memory slot board for the few
host-side exchanges a multi-rank run needs (barrier, broadcast of the 128-byte RCCL id, all-gather of the
IPC handles of the peer-write reducer, max over ranks of a wall time), and a ctypes shim over the HIP
runtime the engine itself links (device buffers for bench.py and the tests).

Why not torch.distributed: the engine's library is then the ONLY HIP / RCCL user of a rank's process -- one
HIP runtime (/opt/rocm/lib/libamdhip64.so) and one librccl.so ("What's weak" 4: with torch
imported first the engine's nccl* symbols bound to torch's bundled librccl 2.26 while its HIP calls went to
ROCm's runtime).  This is host plumbing for bench.py / tests, not part of the NDT path.

Two ways to get N ranks, same code in the ranks:
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment: the process becomes a PARENT that
    touches no GPU, starts N children (RANK / LOCAL_RANK / WORLD_SIZE / NDT_RANKS_BOARD in their
    environment), relays rank 0's output and exits non-zero if any child does (`launch`);
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the agent has set
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT already: the ranks find each other through a board named
    after the agent's pid and the port (all ranks of one node share the agent as parent).
"""
import ctypes as C
import mmap
import os
import struct
import subprocess
import sys
import time

_MAGIC = 0x4E44545F424F4152  # "NDT_BOAR"
_MAX_RANKS = 64
_PAYLOAD = 256
_HDR = 64
_RANK_BYTES = 64 + 2 * _PAYLOAD          # {seq, attach, ack, pad} + two payload generations
_FILE_BYTES = _HDR + _MAX_RANKS * _RANK_BYTES


class BoardError(RuntimeError):
    pass


class Board:
    """A file in /dev/shm with one slot per rank.  Every rank writes only its own slot; an exchange is
    "write payload, bump own sequence word, wait for everybody's sequence word, read all payloads" with
    two payload generations (a rank can be at most one round ahead of the slowest), exactly the scheme of
    the engine's shared-memory reducer (slam-sam_amd/csrc/ndt_comm.cpp).  x86 keeps the two stores of one
    rank in order; CPython cannot reorder them.  Rank 0 creates the file; the others attach through a
    handshake that only a LIVE rank 0 answers, so a file left behind by a crashed run is never used."""

    def __init__(self, path, rank, nranks, timeout=120.0):
        if not (0 <= rank < nranks <= _MAX_RANKS):
            raise ValueError("rank %d of %d" % (rank, nranks))
        self.path, self.rank, self.n, self.timeout = path, rank, nranks, timeout
        self.round = 0
        self.mm = None
        t0 = time.monotonic()
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, _FILE_BYTES)
                self.mm = mmap.mmap(fd, _FILE_BYTES)
            finally:
                os.close(fd)
            struct.pack_into("<QII", self.mm, 0, _MAGIC, nranks, os.getpid())
            for r in range(1, nranks):   # echo every rank's attach word
                while True:
                    v = self._u64(self._rank_off(r) + 8)
                    if v:
                        self._set_u64(self._rank_off(r) + 16, v)
                        break
                    self._check_time(t0, "waiting for rank %d to attach to %s" % (r, path))
                    time.sleep(0.0005)
        else:
            mine = ((os.getpid() << 32) ^ (rank << 24) ^ time.monotonic_ns() ^ 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF | 1
            while self.mm is None:
                self._check_time(t0, "attaching to %s" % path)
                try:
                    fd = os.open(path, os.O_RDWR)
                except FileNotFoundError:
                    time.sleep(0.002)
                    continue
                try:
                    if os.fstat(fd).st_size < _FILE_BYTES:
                        time.sleep(0.002)
                        continue
                    mm = mmap.mmap(fd, _FILE_BYTES)
                finally:
                    os.close(fd)
                t_try = time.monotonic()
                live = False
                while time.monotonic() - t_try < 0.5:
                    magic, nr, _pid = struct.unpack_from("<QII", mm, 0)
                    if magic == _MAGIC and nr == nranks:
                        struct.pack_into("<Q", mm, self._rank_off(rank) + 8, mine)
                        if struct.unpack_from("<Q", mm, self._rank_off(rank) + 16)[0] == mine:
                            live = True
                            break
                    time.sleep(0.0005)
                if live:
                    self.mm = mm
                else:
                    mm.close()   # a stale file, or rank 0 has not re-created it yet: open the name again
        self.barrier()

    @staticmethod
    def _rank_off(r):
        return _HDR + r * _RANK_BYTES

    def _u64(self, off):
        return struct.unpack_from("<Q", self.mm, off)[0]

    def _set_u64(self, off, v):
        struct.pack_into("<Q", self.mm, off, v)

    def _check_time(self, t0, what):
        if time.monotonic() - t0 > self.timeout:
            raise BoardError("rank %d: timed out %s" % (self.rank, what))

    def allgather(self, payload=b""):
        """Every rank's payload (<= 256 bytes), in rank order."""
        if len(payload) > _PAYLOAD - 4:
            raise ValueError("payload too large")
        self.round += 1
        gen = self.round & 1
        off = self._rank_off(self.rank) + 64 + gen * _PAYLOAD
        struct.pack_into("<I", self.mm, off, len(payload))
        self.mm[off + 4:off + 4 + len(payload)] = payload
        self._set_u64(self._rank_off(self.rank), self.round)
        t0 = time.monotonic()
        spins = 0
        for r in range(self.n):
            while self._u64(self._rank_off(r)) < self.round:
                spins += 1
                if spins > 200:
                    time.sleep(0.0002)
                    self._check_time(t0, "in exchange %d waiting for rank %d" % (self.round, r))
        out = []
        for r in range(self.n):
            o = self._rank_off(r) + 64 + gen * _PAYLOAD
            ln = struct.unpack_from("<I", self.mm, o)[0]
            out.append(bytes(self.mm[o + 4:o + 4 + ln]))
        return out

    def barrier(self):
        self.allgather(b"")

    def bcast(self, payload, src=0):
        return self.allgather(payload if self.rank == src else b"")[src]

    def allmax(self, x):
        return max(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(x))))

    def allmin(self, x):
        return min(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(x))))

    def close(self):
        if self.mm is not None:
            self.mm.close()
            self.mm = None
            if self.rank == 0:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


def env_world():
    """(rank, local_rank, world) from the environment, or None when this process is not a rank."""
    if "WORLD_SIZE" not in os.environ:
        return None
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ["WORLD_SIZE"]))


def board_path():
    """Where this job's ranks meet: NDT_RANKS_BOARD (set by `launch`), else a name every rank of a
    torch.distributed.run agent derives alike (the agent's pid and the rendezvous port)."""
    p = os.environ.get("NDT_RANKS_BOARD")
    if p:
        return p
    return "/dev/shm/ndt_board_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))


def pick_device(local_rank, world, device_count, rehearsal=False):
    """Which device a rank binds, and a note for config.launch.  One rank per device is the deployment: LOCAL_RANK is the
    device ordinal.  A launcher that narrows the visible devices per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES:
    device count 1 while WORLD_SIZE > 1) leaves exactly one choice, device 0 -- hipSetDevice(LOCAL_RANK) would fail
    there.  Anything else with LOCAL_RANK beyond the device count is a launch error: one clear line, non-zero exit."""
    if device_count <= 0:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if rehearsal:
        return 0, "rehearsal: every rank on device 0 (NDT_BENCH_SINGLE_DEVICE=1)"
    if local_rank < device_count:
        return local_rank, None
    if device_count == 1 and world > 1:
        return 0, "rank sees ONE device (visible devices narrowed per rank by the launcher): device 0 used for LOCAL_RANK %d" % local_rank
    raise SystemExit("bench.py: LOCAL_RANK %d but only %d device(s) visible to this rank (WORLD_SIZE %d): launch one rank per "
                     "visible device, or narrow the visible devices to one per rank" % (local_rank, device_count, world))


def variant_budget(n_variants, total=None):
    """Wall seconds one reduce variant of a multi-rank bench may take before the watchdog ends the run with what has been
    measured: the driver allows 600 s for the whole command; NDT_BENCH_WALL_BUDGET (default 420 s) is split evenly over
    the variants plus one share for set-up, the probe and tear-down.  NDT_BENCH_VARIANT_TIMEOUT pins it."""
    pinned = os.environ.get("NDT_BENCH_VARIANT_TIMEOUT")
    if pinned:
        return max(1.0, float(pinned))
    if total is None:
        total = float(os.environ.get("NDT_BENCH_WALL_BUDGET", "420"))
    return max(5.0, total / (max(1, n_variants) + 1))


def _die_with_parent():
    """preexec_fn of a rank: SIGTERM when the parent dies, however it dies (a SIGKILLed parent cannot tell anyone)."""
    try:
        C.CDLL("libc.so.6", use_errno=True).prctl(1, 15)   # PR_SET_PDEATHSIG, SIGTERM
    except Exception:
        pass


def launch(nranks, argv, timeout=None, extra_env=None):
    """PARENT side: start `nranks` children running `argv` (one per GPU; LOCAL_RANK = RANK), relay rank
    0's stdout line by line, wait for all.  Returns the exit status for the parent: 0 only if every child
    returned 0.  A child that fails takes the others down (they would wait for it for ever).  The parent
    never touches a GPU.  The ranks run in sessions of their own and are ended with the parent: SIGTERM / SIGINT to
    the parent (`timeout -k 10 400 python bench.py --gpus 2` signals only the parent) is passed on to every rank's
    process group, a parent that is killed outright takes them along through PR_SET_PDEATHSIG, and the board file is
    unlinked on every way out."""
    import signal
    base = dict(os.environ)
    base.update(extra_env or {})
    base["WORLD_SIZE"] = str(nranks)
    base["NDT_RANKS_BOARD"] = "/dev/shm/ndt_board_%d_%d" % (os.getpid(), time.monotonic_ns() & 0xFFFFFF)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / hipIpc across processes need it on this pool
    procs = []
    stop = {"sig": None}

    def on_signal(signum, _frame):
        stop["sig"] = signum

    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:   # not the main thread
            pass
    try:
        return _launch(nranks, argv, timeout, base, procs, stop)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
        for sg, h in old.items():
            signal.signal(sg, h)
        try:
            os.unlink(base["NDT_RANKS_BOARD"])
        except OSError:
            pass


def _launch(nranks, argv, timeout, base, procs, stop):
    import signal
    for r in range(nranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      text=(r == 0), start_new_session=True, preexec_fn=_die_with_parent))
    import threading

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    t0 = time.monotonic()
    status = 0
    live = set(range(nranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print("ranks.launch: rank %d exited with %d; stopping the others" % (r, rc), file=sys.stderr, flush=True)
        if status != 0 or stop["sig"] is not None or (timeout is not None and time.monotonic() - t0 > timeout):
            if status == 0 and stop["sig"] is not None:
                status = 128 + int(stop["sig"])
                print("ranks.launch: signal %d: stopping the ranks" % stop["sig"], file=sys.stderr, flush=True)
            elif status == 0:
                status = 124
                print("ranks.launch: timed out after %.0f s" % timeout, file=sys.stderr, flush=True)
            t_kill = time.monotonic()
            for r in sorted(live):
                try:
                    os.killpg(procs[r].pid, signal.SIGTERM)   # the rank's whole session (it may have children of its own)
                except OSError:
                    pass
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=max(0.1, 10.0 - (time.monotonic() - t_kill)))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(procs[r].pid, signal.SIGKILL)
                    except OSError:
                        pass
                    procs[r].wait()
            live.clear()
        time.sleep(0.02)
    t.join(timeout=5.0)
    return status


class Hip:
    """The few HIP runtime calls a host needs to hold clouds in HBM, through the SAME runtime library the
    engine links (ctypes on /opt/rocm/lib/libamdhip64.so -- no second runtime in the process)."""

    H2D, D2H = 1, 2

    def __init__(self, device=None):
        self.rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipSetDevice.argtypes = [C.c_int]
        self.rt.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
        self.rt.hipGetErrorString.restype = C.c_char_p
        self.rt.hipGetErrorString.argtypes = [C.c_int]
        self.live = []
        if device is not None:
            self.set_device(device)

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.rt.hipGetErrorString(rc).decode()))

    def device_count(self):
        n = C.c_int(0)
        return n.value if self.rt.hipGetDeviceCount(C.byref(n)) == 0 else 0

    def set_device(self, d):
        self._ok(self.rt.hipSetDevice(int(d)), "hipSetDevice(%d)" % d)

    def synchronize(self):
        self._ok(self.rt.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def upload(self, arr):
        """A NumPy array copied into a fresh device buffer; returns the device address."""
        import numpy as np
        a = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._ok(self.rt.hipMalloc(C.byref(p), max(a.nbytes, 4)), "hipMalloc(%d)" % a.nbytes)
        self._ok(self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, self.H2D), "hipMemcpy H2D")
        self.live.append(p)
        return p.value

    def write(self, ptr, arr):
        import numpy as np
        a = np.ascontiguousarray(arr)
        self._ok(self.rt.hipMemcpy(C.c_void_p(ptr), a.ctypes.data, a.nbytes, self.H2D), "hipMemcpy H2D")

    def free_all(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []
