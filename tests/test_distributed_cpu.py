"""World-size-2 gloo test (CPU) of the sharded N>1 path: the source cloud is split with
ndt_shard_range, every rank evaluates its shard, the 32-double partial is all-reduced and
the product's host Newton driver (ndt_newton_align) must take exactly the steps of the
unsharded run.  The per-shard evaluator here is the CPU oracle acting as the checker; the
HIP evaluator replaces it on the GPU (tests/test_gpu_multiproc.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg, O = ge.load_package(), ge.load_oracle()
    S = pkg.synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, tgt, gt, guess = S.two_planes(seed=77, max_points=4001)  # odd size: ragged shards
        kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=40)
        oprm = O.default_params(symmetrize_hessian=1, **kw)
        grid = O.Grid(tgt, oprm)  # target replicated on every rank
        b, c = pkg.shard_range(len(src), rank, world)
        shard = np.ascontiguousarray(src[b:b + c])
        n_calls = [0]

        def evaluator(pose, T, need_h):
            d = grid.derivatives(shard, pose, T=T, compute_hessian=need_h)
            w = torch.from_numpy(pkg.pack_eval(d["score"], d["gradient"], d["hessian"], d["nvtl_sum"],
                                               d["n_with_neighbors"], d["n_pairs"]))
            dist.all_reduce(w, op=dist.ReduceOp.SUM)  # the only exchange of the path
            n_calls[0] += 1
            return w.numpy()

        got = pkg.newton_align(pkg.default_params(**kw), len(src), guess, evaluator)
        # every rank must have taken the same decisions
        sig = torch.tensor([got["iterations"], got["n_evaluations"], n_calls[0]], dtype=torch.int64)
        gathered = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(gathered, sig)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        if rank == 0:
            ref = grid.align(src, guess)
            np.savez(os.path.join(out_dir, "result.npz"), got_T=got["T"], ref_T=ref["T"],
                     got_pose=got["pose"], ref_pose=ref["pose"], got_it=got["iterations"], ref_it=ref["iterations"],
                     got_ev=got["n_evaluations"], ref_ev=ref["n_evaluations"], got_H=got["hessian"],
                     ref_H=ref["hessian"], tp=got["transform_probability"], ref_tp=ref["transform_probability"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_align_matches_single(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "result.npz"))
    # sums over two shards differ from the single sum only by f64 association
    assert int(z["got_it"]) == int(z["ref_it"])
    assert int(z["got_ev"]) == int(z["ref_ev"])
    np.testing.assert_allclose(z["got_pose"], z["ref_pose"], atol=1e-8)
    np.testing.assert_allclose(z["got_T"], z["ref_T"], atol=1e-6)
    np.testing.assert_allclose(z["got_H"], z["ref_H"], rtol=1e-8)
    assert float(z["tp"]) == pytest.approx(float(z["ref_tp"]), rel=1e-9)
