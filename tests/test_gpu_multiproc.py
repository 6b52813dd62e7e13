"""N>1 path on the GPU: two processes share the one MI355X of the test box, each owns a
handle, builds the (replicated) voxel table, takes its shard of the source and sums the
32-double evaluation through the shared-memory reducer of the C-ABI.  The sharded align
must reproduce the single-process align.  (RCCL refuses two ranks on one device, so the
RCCL reducer is exercised only by bench.py on a multi-GPU node.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, name, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    S = pkg.synth
    cfg = S.config_c2()
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4,
                                           max_iterations=35)
    ndt.setInputTarget(cfg["target"])
    b, c = pkg.shard_range(len(cfg["source"]), rank, world)
    ndt.setInputSource(cfg["source"][b:b + c])
    ndt.setGlobalSourceSize(len(cfg["source"]))
    ndt.commInitShm(name, rank, world)
    T = ndt.align(cfg["guess"])
    r = ndt.getResult()
    e = ndt.evalDerivatives(r["pose"])[0]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=T, it=r["iterations"], ev=r["n_evaluations"],
             H=r["hessian"], tp=r["transform_probability"], n_pairs=e["n_pairs"], score=e["score"])
    ndt.commDestroy()
    ndt.close()


def test_two_processes_one_gpu_shm_reduction(pkg, S, tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world = 2
    name = "/ndt_test_%d" % os.getpid()
    procs = [ctx.Process(target=_worker, args=(r, world, name, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    cfg = S.config_c2()
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4,
                                           max_iterations=35)
    ndt.setInputTarget(cfg["target"])
    ndt.setInputSource(cfg["source"])
    T = ndt.align(cfg["guess"])
    r = ndt.getResult()
    z = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    # both ranks hold the same global result ...
    assert np.array_equal(z[0]["T"], z[1]["T"]) and np.array_equal(z[0]["H"], z[1]["H"])
    assert int(z[0]["it"]) == int(z[1]["it"]) and int(z[0]["ev"]) == int(z[1]["ev"])
    # ... which is the single-process result up to f64 association of the two partial sums
    dt, dr = S.pose_error(z[0]["T"], T)
    assert dt < 1e-5 and dr < 1e-6
    # pair count of the last evaluation: the two runs stop within 1e-5 m of each other, so only
    # points sitting on a voxel face can be counted differently
    assert abs(int(z[0]["n_pairs"]) - int(r["n_pairs"])) <= max(2, 1e-4 * r["n_pairs"])
    assert float(z[0]["tp"]) == pytest.approx(r["transform_probability"], rel=1e-6)
