"""Ad-hoc GPU check used during bring-up (not collected by pytest)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle()
S = pkg.synth
print(pkg.backend_info(), flush=True)

def check(cfg, iters=35):
    print('==', cfg['name'], cfg['source'].shape, cfg['target'].shape, flush=True)
    prm = O.default_params(resolution=cfg['resolution'], step_size=0.1, trans_epsilon=1e-4, max_iterations=iters, num_threads=16)
    t = time.time(); g = O.Grid(cfg['target'], prm); t_og = time.time() - t
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=cfg['resolution'], step_size=0.1, trans_epsilon=1e-4, max_iterations=iters)
    t = time.time(); ndt.setInputTarget(cfg['target']); t_hg = time.time() - t
    gi = ndt.getGridInfo(); print('grid oracle %.3fs leaves %d | hip %.3fs (dev %.3f ms) leaves %d cells %d' % (t_og, g.n_leaves, t_hg, gi['ms_build'], gi['n_leaves'], gi['n_cells']), flush=True)
    L = ndt.getLeaves(); OL = g.export()
    same = np.array_equal(L['cell'], OL['cell']) and np.array_equal(L['count'], OL['count'])
    print('leaf cells/count equal:', same)
    if same and len(L['cell']):
        for k in ('mean', 'cov', 'icov', 'evals'):
            den = np.abs(OL[k]).max(axis=tuple(range(1, OL[k].ndim)), keepdims=True)
            print('  ', k, 'max rel err', float((np.abs(L[k] - OL[k]) / den).max()))
    ndt.setInputSource(cfg['source'])
    p0 = O.matrix_to_pose(cfg['guess'])
    ndt.enableKernelTiming(True)
    for p in (p0, p0 + np.array([0.05, -0.03, 0.02, 0.01, -0.005, 0.008])):
        d = g.derivatives(cfg['source'], p)
        e = ndt.evalDerivatives(p)[0]
        print('  score', d['score'], e['score'], 'pairs', d['n_pairs'], e['n_pairs'], 'with', d['n_with_neighbors'], e['n_with_neighbors'])
        print('  g rel', np.linalg.norm(d['gradient'] - e['gradient']) / np.linalg.norm(d['gradient']),
              'H rel', np.linalg.norm(d['hessian'] - e['hessian']) / np.linalg.norm(d['hessian']),
              'nvtl', d['nvtl_sum'], e['nvtl_sum'], 'kernel ms', ndt.getTiming()['ms_last_eval_kernel'])
    ndt.enableKernelTiming(False)
    t = time.time(); T = ndt.align(cfg['guess']); t_h = time.time() - t
    r = ndt.getResult()
    t = time.time(); ro = g.align(cfg['source'], cfg['guess']); t_o = time.time() - t
    print('align hip %.4fs it %d ev %d conv %s | oracle %.3fs it %d ev %d conv %s' % (t_h, r['iterations'], r['n_evaluations'], r['converged'], t_o, ro['iterations'], ro['n_evaluations'], ro['converged']))
    print('  vs oracle', S.pose_error(T, ro['T']), ' vs gt', S.pose_error(T, cfg['gt']), 'oracle vs gt', S.pose_error(ro['T'], cfg['gt']))
    for _ in range(3):
        t = time.time(); ndt.align(cfg['guess']); print('  align again %.2f ms' % ((time.time() - t) * 1e3), ndt.getResult()['ms_total'])
    sys.stdout.flush()

check(S.config_c1())
check(S.config_c2())
if len(sys.argv) > 1 and sys.argv[1] == 'c3':
    t = time.time(); c3 = S.config_c3(); print('c3 gen', time.time() - t)
    check(c3)
