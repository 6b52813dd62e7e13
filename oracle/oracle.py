"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (slam-sam_amd/) never does.  See
oracle/ndt_oracle.h for the parity status ("derivative-level parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libndt_oracle.so")

DIRECT1, DIRECT7, DIRECT26, KDTREE = 1, 7, 26, 27
HESSIAN_FULL, HESSIAN_GAUSS_NEWTON = 0, 1
COV_SVN, COV_PCL_RECALLED = 0, 1
PAIR_SVN, PAIR_PCLOMP_RECALLED = 0, 1


class Params(C.Structure):
    _fields_ = [
        ("resolution", C.c_float),
        ("outlier_ratio", C.c_double),
        ("step_size", C.c_double),
        ("trans_epsilon", C.c_double),
        ("max_iterations", C.c_int),
        ("search_method", C.c_int),
        ("min_points_per_voxel", C.c_int),
        ("eig_inflation_ratio", C.c_double),
        ("hessian_mode", C.c_int),
        ("cov_mode", C.c_int),
        ("pair_mode", C.c_int),
        ("add_ridge", C.c_int),
        ("use_line_search", C.c_int),
        ("num_threads", C.c_int),
        ("use_regularization", C.c_int),
        ("regularization_scale_factor", C.c_float),
        ("regularization_pose", C.c_float * 16),
        ("symmetrize_hessian", C.c_int),
    ]


class GridInfo(C.Structure):
    _fields_ = [
        ("min_b", C.c_int * 3), ("max_b", C.c_int * 3), ("div_b", C.c_int * 3),
        ("leaf", C.c_float), ("inv_leaf", C.c_float),
        ("n_leaves", C.c_int64), ("n_cells_hit", C.c_int64),
    ]


class Derivs(C.Structure):
    _fields_ = [
        ("score", C.c_double), ("gradient", C.c_double * 6), ("hessian", C.c_double * 36),
        ("nvtl_sum", C.c_double), ("n_with_neighbors", C.c_int64), ("n_pairs", C.c_int64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("final_transformation", C.c_float * 16), ("final_pose", C.c_double * 6),
        ("converged", C.c_int), ("iterations", C.c_int), ("n_evaluations", C.c_int),
        ("hessian", C.c_double * 36), ("score", C.c_double),
        ("transform_probability", C.c_double), ("nvtl", C.c_double),
        ("n_logged", C.c_int), ("log_pose", (C.c_double * 6) * 128),
        ("log_step", C.c_double * 128), ("log_score", C.c_double * 128),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("ndt_oracle.cpp", "ndt_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libndt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.oracle_default_params.argtypes = [C.POINTER(Params)]
        L.oracle_grid_build.restype = C.c_void_p
        L.oracle_grid_build.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(Params)]
        L.oracle_grid_free.argtypes = [C.c_void_p]
        L.oracle_grid_get_info.argtypes = [C.c_void_p, C.POINTER(GridInfo)]
        L.oracle_grid_export.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.oracle_grid_neighbors.restype = C.c_int
        L.oracle_grid_neighbors.argtypes = [C.c_void_p, fp, C.c_int, C.POINTER(C.c_int64)]
        L.oracle_gauss_constants.argtypes = [C.c_double, C.c_double, dp]
        L.oracle_angle_tables.argtypes = [dp, fp, fp]
        L.oracle_pose_to_matrix.argtypes = [dp, fp]
        L.oracle_matrix_to_pose.argtypes = [fp, dp]
        L.oracle_derivatives.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, fp, dp,
                                         C.POINTER(Params), C.c_int, C.POINTER(Derivs)]
        L.oracle_align.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, fp,
                                   C.POINTER(Params), C.POINTER(Result)]
        L.oracle_two_plane_fixture.restype = C.c_size_t
        L.oracle_two_plane_fixture.argtypes = [fp, fp, dp, dp]
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().oracle_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "regularization_pose":
            arr = np.asarray(v, dtype=np.float32).reshape(4, 4).T.ravel()  # -> column-major
            for i in range(16):
                p.regularization_pose[i] = float(arr[i])
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def _xyz(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def mat_to_colmajor(T):
    """4x4 numpy (row,col) -> 16 floats column-major (Eigen layout)."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).ravel()


def colmajor_to_mat(a):
    return np.asarray(a, dtype=np.float64).reshape(4, 4).T.copy()


class Grid:
    def __init__(self, xyz, params):
        self.xyz = _xyz(xyz)
        self.params = params
        self.h = lib().oracle_grid_build(self.xyz.ctypes.data, len(self.xyz), 12, C.byref(params))
        info = GridInfo()
        lib().oracle_grid_get_info(self.h, C.byref(info))
        self.info = info
        self.min_b = np.array(info.min_b[:]); self.max_b = np.array(info.max_b[:])
        self.div_b = np.array(info.div_b[:])
        self.n_leaves = info.n_leaves

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_grid_free(self.h)
            self.h = None

    def export(self):
        n = self.n_leaves
        out = dict(cell=np.zeros(n, np.int64), count=np.zeros(n, np.int32),
                   mean=np.zeros((n, 3)), cov=np.zeros((n, 3, 3)), icov=np.zeros((n, 3, 3)),
                   evecs=np.zeros((n, 3, 3)), evals=np.zeros((n, 3)))
        lib().oracle_grid_export(self.h, *[out[k].ctypes.data for k in
                                           ("cell", "count", "mean", "cov", "icov", "evecs", "evals")])
        return out

    def neighbors(self, p, method=DIRECT7):
        p = np.asarray(p, dtype=np.float32)
        out = (C.c_int64 * 27)()
        n = lib().oracle_grid_neighbors(self.h, _fp(p), method, out)
        return list(out[:n])

    def derivatives(self, src, pose6, T=None, compute_hessian=True, params=None):
        """score/g/H at pose6; T (4x4 numpy) defaults to the matrix built from pose6."""
        src = _xyz(src)
        pose6 = np.ascontiguousarray(pose6, dtype=np.float64)
        Tc = pose_to_matrix_colmajor(pose6) if T is None else mat_to_colmajor(T)
        d = Derivs()
        prm = params if params is not None else self.params
        lib().oracle_derivatives(self.h, src.ctypes.data, len(src), 12, _fp(Tc), _dp(pose6),
                                 C.byref(prm), int(compute_hessian), C.byref(d))
        return dict(score=d.score, gradient=np.array(d.gradient[:]),
                    hessian=np.array(d.hessian[:]).reshape(6, 6), nvtl_sum=d.nvtl_sum,
                    n_with_neighbors=d.n_with_neighbors, n_pairs=d.n_pairs)

    def align(self, src, guess, params=None):
        src = _xyz(src)
        g = mat_to_colmajor(guess)
        r = Result()
        prm = params if params is not None else self.params
        lib().oracle_align(self.h, src.ctypes.data, len(src), 12, _fp(g), C.byref(prm), C.byref(r))
        n = r.n_logged
        return dict(T=colmajor_to_mat(r.final_transformation[:]), pose=np.array(r.final_pose[:]),
                    converged=bool(r.converged), iterations=r.iterations,
                    n_evaluations=r.n_evaluations, hessian=np.array(r.hessian[:]).reshape(6, 6),
                    score=r.score, transform_probability=r.transform_probability, nvtl=r.nvtl,
                    log_pose=np.array([list(r.log_pose[i][:]) for i in range(n)]).reshape(n, 6),
                    log_step=np.array(r.log_step[:n]), log_score=np.array(r.log_score[:n]))


def gauss_constants(resolution, outlier_ratio):
    out = np.zeros(3)
    lib().oracle_gauss_constants(float(resolution), float(outlier_ratio), _dp(out))
    return out


def angle_tables(pose6):
    pose6 = np.ascontiguousarray(pose6, dtype=np.float64)
    j = np.zeros(24, np.float32); h = np.zeros(45, np.float32)
    lib().oracle_angle_tables(_dp(pose6), _fp(j), _fp(h))
    return j.reshape(8, 3), h.reshape(15, 3)


def pose_to_matrix_colmajor(pose6):
    pose6 = np.ascontiguousarray(pose6, dtype=np.float64)
    T = np.zeros(16, np.float32)
    lib().oracle_pose_to_matrix(_dp(pose6), _fp(T))
    return T


def pose_to_matrix(pose6):
    return colmajor_to_mat(pose_to_matrix_colmajor(pose6))


def matrix_to_pose(T):
    Tc = mat_to_colmajor(T)
    p = np.zeros(6)
    lib().oracle_matrix_to_pose(_fp(Tc), _dp(p))
    return p


def covariance_for_gtsam(hessian, eps=1e-6, gtsam_order=True):
    """NumPy restatement of what the drivers do with NdtResult::hessian: lidarCov =
    -(hessian + 1e-6 I)^-1 (ref: run/pipeline.cpp:594-596), then
    RegisterCallback::reorderCovarianceForGTSAM (ref: src/registercallback.cpp:170-186):
    C_tt -> bottom-right, C_rr -> top-left, the two cross blocks copied in place."""
    H = np.asarray(hessian, dtype=np.float64).reshape(6, 6)
    cov = -np.linalg.inv(H + eps * np.eye(6))
    if not gtsam_order:
        return cov
    out = np.empty((6, 6))
    out[3:, 3:] = cov[:3, :3]
    out[:3, :3] = cov[3:, 3:]
    out[:3, 3:] = cov[:3, 3:]
    out[3:, :3] = cov[3:, :3]
    return out


def two_plane_fixture():
    """(source Nx3 f32, target Nx3 f32, gt 4x4 f64, guess 4x4 f64) of the reference test."""
    n = 35912
    src = np.zeros((n, 3), np.float32); tgt = np.zeros((n, 3), np.float32)
    gt = np.zeros(16); guess = np.zeros(16)
    got = lib().oracle_two_plane_fixture(_fp(src), _fp(tgt), _dp(gt), _dp(guess))
    assert got == n, got
    return src, tgt, gt.reshape(4, 4).T.copy(), guess.reshape(4, 4).T.copy()


# ---- SVN-NDT outer loop (ref: extern/svn_ndt/include/svn_ndt_impl.hpp:675-964) -------------
class SvnParams(C.Structure):
    _fields_ = [("particle_count", C.c_int), ("max_iterations", C.c_int),
                ("kernel_bandwidth", C.c_double), ("step_size", C.c_double),
                ("stop_threshold", C.c_double)]


class SvnResult(C.Structure):
    _fields_ = [("final_pose", C.c_double * 16), ("final_covariance", C.c_double * 36),
                ("converged", C.c_int), ("iterations", C.c_int), ("n_logged", C.c_int),
                ("log_mean_update", C.c_double * 128)]


def _svn_protos():
    L = lib()
    if not getattr(L, "_svn_ready", False):
        dp = C.POINTER(C.c_double)
        L.oracle_svn_align.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, dp, dp,
                                       C.POINTER(Params), C.POINTER(SvnParams), C.POINTER(SvnResult)]
        L.oracle_svn_sample_particles.argtypes = [dp, C.c_int, C.c_uint64, dp]
        L.oracle_se3_expmap.argtypes = [dp, dp]
        L.oracle_se3_logmap.argtypes = [dp, dp]
        L._svn_ready = True
    return L


def _pose16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).ravel()


def svn_sample_particles(prior, K, seed):
    """K x 4 x 4 poses: prior.retract(sigma * N(0,1)), sigmas as svn_ndt_impl.hpp:709."""
    L = _svn_protos()
    out = np.zeros(16 * K)
    L.oracle_svn_sample_particles(_dp(_pose16(prior)), K, seed, _dp(out))
    return out.reshape(K, 4, 4).transpose(0, 2, 1).copy()


def svn_align(grid, src, prior, particles, params, K=None, max_iterations=50, kernel_bandwidth=1.0,
              step_size=1.0, stop_threshold=1e-4):
    L = _svn_protos()
    src = _xyz(src)
    particles = np.asarray(particles, dtype=np.float64)
    K = len(particles) if K is None else K
    part = np.ascontiguousarray(particles.transpose(0, 2, 1)).ravel().copy()
    sp = SvnParams(K, max_iterations, kernel_bandwidth, step_size, stop_threshold)
    r = SvnResult()
    L.oracle_svn_align(grid.h, src.ctypes.data, len(src), 12, _dp(_pose16(prior)), _dp(part),
                       C.byref(params), C.byref(sp), C.byref(r))
    return dict(pose=np.array(r.final_pose[:]).reshape(4, 4).T.copy(),
                covariance=np.array(r.final_covariance[:]).reshape(6, 6), converged=bool(r.converged),
                iterations=r.iterations, log_mean_update=np.array(r.log_mean_update[:r.n_logged]),
                particles=part.reshape(K, 4, 4).transpose(0, 2, 1).copy())


def se3_expmap(xi):
    L = _svn_protos()
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    T = np.zeros(16)
    L.oracle_se3_expmap(_dp(xi), _dp(T))
    return T.reshape(4, 4).T.copy()


def se3_logmap(T):
    L = _svn_protos()
    xi = np.zeros(6)
    L.oracle_se3_logmap(_dp(_pose16(T)), _dp(xi))
    return xi
