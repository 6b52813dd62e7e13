"""Generates the committed golden fixtures from the CPU oracle (oracle/ndt_oracle.cpp).

    python tests/golden/make_golden.py

The reference holds no golden vectors for this path and cannot be built here
(SURVEY.md section 8c), so these vectors are ORACLE outputs: they pin the oracle (and the
HIP path) against regressions; parity with the reference itself is pinned only by the
reference test's own assertions (tests/test_oracle.py::test_reference_fixture_pins_oracle,
tests/test_svn.py).
Inputs and expected outputs only -- no reference source text.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    pkg = ge.load_package()
    O = ge.load_oracle()
    O.build(force=True)
    S = pkg.synth

    # ---- G1: small two-plane case (3000 pts), all levels --------------------------
    src, tgt, gt, guess = S.two_planes(seed=2024, max_points=3000)
    params = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50)
    prm = O.default_params(**params)
    grid = O.Grid(tgt, prm)
    L = grid.export()
    p0 = O.matrix_to_pose(guess)
    poses = np.stack([p0, p0 + np.array([0.05, -0.03, 0.02, 0.01, -0.005, 0.008]),
                      O.matrix_to_pose(gt)])
    d = [grid.derivatives(src, p) for p in poses]
    prm_gn = O.default_params(hessian_mode=O.HESSIAN_GAUSS_NEWTON, add_ridge=1, **params)
    d_gn = grid.derivatives(src, p0, params=prm_gn)
    prm_d1 = O.default_params(search_method=O.DIRECT1, **params)
    d_d1 = grid.derivatives(src, p0, params=prm_d1)
    r = grid.align(src, guess)
    np.savez_compressed(
        os.path.join(HERE, "g1_two_plane_3k.npz"),
        source=src, target=tgt, gt=gt, guess=guess, poses=poses,
        resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50,
        min_b=grid.min_b, div_b=grid.div_b,
        leaf_cell=L["cell"], leaf_count=L["count"], leaf_mean=L["mean"], leaf_cov=L["cov"],
        leaf_icov=L["icov"], leaf_evals=L["evals"],
        score=np.array([x["score"] for x in d]), gradient=np.stack([x["gradient"] for x in d]),
        hessian=np.stack([x["hessian"] for x in d]), n_pairs=np.array([x["n_pairs"] for x in d]),
        n_with=np.array([x["n_with_neighbors"] for x in d]), nvtl_sum=np.array([x["nvtl_sum"] for x in d]),
        gn_score=d_gn["score"], gn_gradient=d_gn["gradient"], gn_hessian=d_gn["hessian"],
        d1_score=d_d1["score"], d1_gradient=d_d1["gradient"], d1_hessian=d_d1["hessian"],
        d1_n_pairs=d_d1["n_pairs"],
        align_T=r["T"], align_pose=r["pose"], align_iterations=r["iterations"],
        align_n_evaluations=r["n_evaluations"], align_converged=int(r["converged"]),
        align_log_pose=r["log_pose"], align_log_step=r["log_step"], align_log_score=r["log_score"],
        align_hessian=r["hessian"])

    # ---- G2: the reference's own test fixture (regenerated at run time) ------------
    rs, rt, rgt, rguess = O.two_plane_fixture()
    prm = O.default_params(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=50)
    g2 = O.Grid(rt, prm)
    r2 = g2.align(rs, rguess)
    np.savez_compressed(
        os.path.join(HERE, "g2_reference_fixture_expect.npz"),
        n_points=len(rs), src_sha=hashlib.sha256(rs.tobytes()).hexdigest(),
        tgt_sha=hashlib.sha256(rt.tobytes()).hexdigest(), tgt_head=rt[:8], gt=rgt, guess=rguess,
        n_leaves=g2.n_leaves, align_T=r2["T"], align_iterations=r2["iterations"],
        align_n_evaluations=r2["n_evaluations"], align_log_step=r2["log_step"],
        align_log_score=r2["log_score"])
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
    print("G1 leaves", len(L["cell"]), "align iters", r["iterations"], "| G2 iters", r2["iterations"],
          "evals", r2["n_evaluations"])


if __name__ == "__main__":
    main()
