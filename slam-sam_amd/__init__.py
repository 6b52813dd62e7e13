"""slam-sam_amd -- MI355X-native NDT scan-matching engine (Python host binding).

Thin ctypes layer over the C-ABI in include/ndt_hip.h (built in-tree as
slam-sam_amd/libndt_hip.so from slam-sam_amd/csrc).  The class below mirrors the
`pclomp::NormalDistributionsTransform` surface the reference's drivers use
(ref: run/pipeline.cpp:464-481,557-568; extern/svn_ndt/test/test_svn_ndt.cpp:144-179)
so tests read like the reference's own.  There is NO CPU fallback: importing
works anywhere, but if the HIP library is missing or no gfx950 device is present
every compute call raises.

The directory name has a hyphen; load it with `__graft_entry__.load_package()`
(registers the module as `slam_sam_amd`).
"""
import ctypes as C
import os

import numpy as np

from . import ranks  # noqa: F401  (one process per GPU without torch: launcher, slot board, HIP shim)
from . import synth  # noqa: F401  (seeded synthetic cloud generators)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDT_HIP_LIB: tuning aid (A/B of two builds in one GPU session); the default is the in-tree build
LIB_PATH = os.environ.get("NDT_HIP_LIB") or os.path.join(_HERE, "libndt_hip.so")

EVAL_WORDS = 32
# pclomp::NeighborSearchMethod order
KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3
HESSIAN_FULL, HESSIAN_GAUSS_NEWTON = 0, 1
COV_SVN, COV_PCL_RECALLED = 0, 1
WAIT_SPIN, WAIT_BLOCK = 0, 1
SOURCE_ORDER_AUTO, SOURCE_ORDER_KEEP, SOURCE_ORDER_SORT = 0, 1, 2
PRELAUNCH_AUTO, PRELAUNCH_OFF, PRELAUNCH_ONE_STREAM = 0, 1, 2
RECORDS_F64, RECORDS_PACKED48 = 0, 1
HANDOFF_ASYNC, HANDOFF_SYNC = 0, 1
PRESET_DEFAULT, PRESET_PCLOMP_RECALLED, PRESET_SVN = 0, 1, 2

STATUS = {0: "NDT_OK", -1: "NDT_ERR_INVALID_ARG", -2: "NDT_ERR_NO_DEVICE", -3: "NDT_ERR_HIP",
          -4: "NDT_ERR_NO_TARGET", -5: "NDT_ERR_NO_SOURCE", -6: "NDT_ERR_GRID_OVERFLOW",
          -7: "NDT_ERR_ALLOC", -8: "NDT_ERR_COMM", -9: "NDT_ERR_UNSUPPORTED"}


class NdtError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))


class Params(C.Structure):
    _fields_ = [
        ("resolution", C.c_float), ("step_size", C.c_double), ("trans_epsilon", C.c_double),
        ("max_iterations", C.c_int), ("outlier_ratio", C.c_double), ("search_method", C.c_int),
        ("min_points_per_voxel", C.c_int), ("eig_inflation_ratio", C.c_double),
        ("hessian_mode", C.c_int), ("cov_mode", C.c_int), ("add_ridge", C.c_int),
        ("use_line_search", C.c_int), ("regularization_scale_factor", C.c_float),
        ("num_threads", C.c_int), ("device_id", C.c_int), ("wait_mode", C.c_int), ("source_order", C.c_int),
        ("prelaunch", C.c_int),
    ]


class Result(C.Structure):
    _fields_ = [
        ("final_transformation", C.c_float * 16), ("final_pose", C.c_double * 6),
        ("converged", C.c_int), ("iterations", C.c_int), ("n_evaluations", C.c_int),
        ("hessian", C.c_double * 36), ("score", C.c_double), ("transform_probability", C.c_double),
        ("nearest_voxel_transformation_likelihood", C.c_double), ("n_pairs", C.c_int64),
        ("n_points_with_neighbors", C.c_int64), ("ms_total", C.c_double), ("ms_device", C.c_double),
        ("n_evaluations_reused", C.c_int),
    ]


class Score(C.Structure):
    _fields_ = [
        ("score", C.c_double), ("transform_probability", C.c_double),
        ("nearest_voxel_transformation_likelihood", C.c_double), ("n_pairs", C.c_int64),
        ("n_points_with_neighbors", C.c_int64),
    ]


class Leaf(C.Structure):
    _fields_ = [
        ("index", C.c_int64), ("point_count", C.c_int32), ("center", C.c_float * 3),
        ("mean", C.c_double * 3), ("cov", C.c_double * 9), ("icov", C.c_double * 9),
        ("evecs", C.c_double * 9), ("evals", C.c_double * 3),
    ]


class GridInfo(C.Structure):
    _fields_ = [
        ("min_b", C.c_int * 3), ("max_b", C.c_int * 3), ("div_b", C.c_int * 3),
        ("leaf_size", C.c_float), ("inverse_leaf_size", C.c_float), ("n_leaves", C.c_int64),
        ("n_cells", C.c_int64), ("n_target_points", C.c_int64), ("ms_build", C.c_double),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("ms_last_eval_kernel", C.c_double), ("ms_last_reduce_kernel", C.c_double),
        ("ms_last_build", C.c_double), ("n_eval_launches", C.c_int64),
        ("ms_eval_kernel_total", C.c_double), ("ms_reduce_kernel_total", C.c_double),
        ("n_timed_evals", C.c_int64),
    ]


class HandoffLaneTiming(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("bytes_in", C.c_int64), ("bytes_dma", C.c_int64),
                ("ms_repack", C.c_double), ("ms_dma", C.c_double), ("dma_gb_per_s", C.c_double),
                ("threads", C.c_int)]


class HandoffTiming(C.Structure):
    _fields_ = [("target", HandoffLaneTiming), ("source", HandoffLaneTiming), ("ms_build_wait", C.c_double),
                ("mode", C.c_int), ("cpu_budget", C.c_int), ("repack_workers", C.c_int)]


class SvnParams(C.Structure):
    _fields_ = [("particle_count", C.c_int), ("max_iterations", C.c_int),
                ("kernel_bandwidth", C.c_double), ("step_size", C.c_double),
                ("stop_threshold", C.c_double)]


class SvnResult(C.Structure):
    _fields_ = [("final_pose", C.c_double * 16), ("final_covariance", C.c_double * 36),
                ("converged", C.c_int), ("iterations", C.c_int), ("last_mean_update", C.c_double),
                ("ms_total", C.c_double), ("ms_stage1", C.c_double), ("ms_stage2", C.c_double),
                ("ms_stage3", C.c_double)]


class Tuning(C.Structure):
    """ndt_tuning (include/ndt_hip.h): the engine's A/B switches.  The library reads none of them from the environment."""
    _fields_ = [(k, C.c_int) for k in (
        "deriv_block", "deriv_summer", "deriv_dedicated", "deriv_single_level_max", "deriv_xcd", "bucket_build",
        "bucket_tile", "fused_sort", "bounds_blocks", "bounds_unroll", "finalize_threads", "build_events",
        "build_wait_sync", "mbox_tagged", "mbox_preload", "prelaunch_streams", "prelaunch_probe", "speculate_first",
        "timing_bracket", "handoff_chunk_pass", "deriv_summer_split", "deriv_one_block_per_cu")] + [("reserved", C.c_int * 10)]


# the variables the tuning programs under tools/ (and bench.py's rehearsals) have always used, mapped onto ndt_tuning by
# apply_env_tuning() -- in Python, on request: the C library itself does not look at them
TUNING_ENV = {
    "NDT_DERIV_BLOCK": "deriv_block", "NDT_DERIV_SUMMER": "deriv_summer", "NDT_DERIV_DEDICATED": "deriv_dedicated",
    "NDT_DERIV_SINGLE_LEVEL_MAX": "deriv_single_level_max", "NDT_DERIV_XCD": "deriv_xcd",
    "NDT_BUCKET_BUILD": "bucket_build", "NDT_BUCKET_TILE": "bucket_tile", "NDT_FUSED_SORT": "fused_sort",
    "NDT_BOUNDS_BLOCKS": "bounds_blocks", "NDT_BOUNDS_UNROLL": "bounds_unroll",
    "NDT_FINALIZE_THREADS": "finalize_threads", "NDT_BUILD_EVENTS": "build_events",
    "NDT_MBOX_TAGGED": "mbox_tagged", "NDT_MBOX_PRELOAD": "mbox_preload",
    "NDT_PRELAUNCH_STREAMS": "prelaunch_streams", "NDT_PRELAUNCH_PROBE": "prelaunch_probe",
    "NDT_SPECULATE_FIRST": "speculate_first", "NDT_TIMING_BRACKET": "timing_bracket",
    "NDT_HANDOFF_CHUNK_PASS": "handoff_chunk_pass", "NDT_DERIV_SUMMER_SPLIT": "deriv_summer_split",
    "NDT_DERIV_ONE_BLOCK_PER_CU": "deriv_one_block_per_cu",
}


def get_tuning():
    t = Tuning()
    rc = lib().ndt_get_tuning(C.byref(t))
    if rc:
        raise NdtError(rc, "ndt_get_tuning")
    return {k: getattr(t, k) for k, _ in Tuning._fields_ if k != "reserved"}


def set_tuning(**fields):
    """ndt_set_tuning with the named fields changed; applies to the handles created and the launches made afterwards."""
    t = Tuning()
    rc = lib().ndt_get_tuning(C.byref(t))
    if rc:
        raise NdtError(rc, "ndt_get_tuning")
    for k, v in fields.items():
        if k == "reserved" or not hasattr(t, k):
            raise KeyError(k)
        setattr(t, k, int(v))
    rc = lib().ndt_set_tuning(C.byref(t))
    if rc:
        raise NdtError(rc, "ndt_set_tuning(%s)" % fields)
    return get_tuning()


def apply_env_tuning(environ=None):
    """For tuning programs and test harnesses: NDT_* variables of the historical names -> ndt_set_tuning."""
    env = os.environ if environ is None else environ
    fields = {f: int(env[k]) for k, f in TUNING_ENV.items() if env.get(k, "") != ""}
    if env.get("NDT_BUILD_WAIT", "") != "":
        fields["build_wait_sync"] = 1 if env["NDT_BUILD_WAIT"] == "sync" else 0
    return set_tuning(**fields) if fields else get_tuning()


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int,
                      C.POINTER(C.c_double))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

# every symbol include/ndt_hip.h declares
ABI_SYMBOLS = [
    "ndt_abi_version", "ndt_default_params", "ndt_create", "ndt_destroy", "ndt_set_params",
    "ndt_get_params", "ndt_last_error", "ndt_backend_info", "ndt_set_target", "ndt_set_target_soa",
    "ndt_set_target_device", "ndt_set_target_device_deferred", "ndt_set_source", "ndt_set_source_soa", "ndt_set_source_device", "ndt_set_source_device_view",
    "ndt_set_regularization_pose", "ndt_clear_regularization_pose", "ndt_align",
    "ndt_eval_derivatives", "ndt_unpack_eval", "ndt_transform_source", "ndt_get_grid_info",
    "ndt_export_leaves", "ndt_newton_align", "ndt_shard_range", "ndt_comm_unique_id",
    "ndt_comm_init_rccl", "ndt_comm_init_shm", "ndt_comm_init_hook", "ndt_comm_destroy",
    "ndt_set_global_source_size", "ndt_enable_kernel_timing", "ndt_get_timing",
    "ndt_svn_default_params", "ndt_svn_sample_particles", "ndt_svn_align",
    "ndt_multigrid_add_target", "ndt_multigrid_remove_target", "ndt_multigrid_count", "ndt_multigrid_create_kdtree",
    "ndt_keyframe_put", "ndt_keyframe_erase", "ndt_keyframe_count", "ndt_set_target_from_keyframes",
    "ndt_result_covariance", "ndt_set_source_from_keyframe",
    "ndt_params_preset", "ndt_score_transform", "ndt_comm_info", "ndt_score_transforms",
    "ndt_xy_covariance_laplace", "ndt_propose_poses_to_search", "ndt_xy_covariance_multi_ndt",
    "ndt_xy_covariance_multi_ndt_score", "ndt_source_changed", "ndt_comm_rank_count", "ndt_comm_p2p_handle", "ndt_comm_init_p2p",
    "ndt_set_record_format", "ndt_get_record_format",
    "ndt_set_handoff_mode", "ndt_get_handoff_mode", "ndt_wait", "ndt_get_handoff_timing",
    "ndt_voxel_downsample_device", "ndt_voxel_downsample", "ndt_get_iteration_history",
    "ndt_get_tuning", "ndt_set_tuning", "ndt_set_keepwarm", "ndt_get_keepwarm",
    "ndt_comm_p2p_selftest", "ndt_comm_p2p_stats", "ndt_angle_tables", "ndt_gauss_constants", "ndt_svn_rbf_kernel",
]

_lib = None


def lib():
    """The C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("slam-sam_amd: %s is missing -- run __graft_entry__.build() "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
        L.ndt_abi_version.restype = C.c_int
        L.ndt_comm_p2p_selftest.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
        L.ndt_comm_p2p_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
        L.ndt_set_keepwarm.argtypes = [vp, C.c_int]
        L.ndt_get_keepwarm.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.ndt_get_tuning.argtypes = [C.POINTER(Tuning)]
        L.ndt_set_tuning.argtypes = [C.POINTER(Tuning)]
        L.ndt_default_params.argtypes = [C.POINTER(Params)]
        L.ndt_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.ndt_destroy.argtypes = [vp]
        L.ndt_set_params.argtypes = [vp, C.POINTER(Params)]
        L.ndt_get_params.argtypes = [vp, C.POINTER(Params)]
        L.ndt_last_error.restype = C.c_char_p
        L.ndt_last_error.argtypes = [vp]
        L.ndt_backend_info.argtypes = [C.c_char_p, C.c_size_t]
        L.ndt_set_target.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.ndt_set_target_soa.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_set_target_device.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_set_target_device_deferred.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_set_source.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.ndt_set_source_soa.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_set_source_device.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_set_source_device_view.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.ndt_source_changed.argtypes = [vp]
        L.ndt_set_record_format.argtypes = [vp, C.c_int]
        L.ndt_get_record_format.argtypes = [vp]
        L.ndt_set_regularization_pose.argtypes = [vp, fp]
        L.ndt_clear_regularization_pose.argtypes = [vp]
        L.ndt_align.argtypes = [vp, fp, C.POINTER(Result)]
        L.ndt_eval_derivatives.argtypes = [vp, dp, fp, C.c_int, C.c_int, dp]
        L.ndt_unpack_eval.restype = None
        L.ndt_unpack_eval.argtypes = [dp, dp, dp, dp]
        L.ndt_transform_source.argtypes = [vp, fp, fp, C.c_size_t]
        L.ndt_get_grid_info.argtypes = [vp, C.POINTER(GridInfo)]
        L.ndt_export_leaves.restype = C.c_int64
        L.ndt_export_leaves.argtypes = [vp, C.POINTER(Leaf), C.c_size_t]
        L.ndt_newton_align.argtypes = [C.POINTER(Params), C.c_int64, fp, fp, EVAL_FN, vp,
                                       C.POINTER(Result)]
        L.ndt_shard_range.restype = None
        L.ndt_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]
        L.ndt_comm_unique_id.argtypes = [vp]
        L.ndt_comm_init_rccl.argtypes = [vp, vp, C.c_int, C.c_int]
        L.ndt_comm_init_shm.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.ndt_comm_init_hook.argtypes = [vp, ALLREDUCE_FN, vp, C.c_int, C.c_int]
        L.ndt_comm_destroy.argtypes = [vp]
        L.ndt_set_global_source_size.argtypes = [vp, C.c_int64]
        L.ndt_enable_kernel_timing.argtypes = [vp, C.c_int]
        L.ndt_get_timing.argtypes = [vp, C.POINTER(Timing)]
        L.ndt_multigrid_add_target.argtypes = [vp, C.c_int64, vp, C.c_size_t, C.c_size_t]
        L.ndt_multigrid_remove_target.argtypes = [vp, C.c_int64]
        L.ndt_multigrid_count.restype = C.c_int64
        L.ndt_multigrid_count.argtypes = [vp]
        L.ndt_multigrid_create_kdtree.argtypes = [vp]
        L.ndt_keyframe_put.argtypes = [vp, C.c_int64, vp, C.c_size_t, C.c_size_t]
        L.ndt_keyframe_erase.argtypes = [vp, C.c_int64]
        L.ndt_keyframe_count.restype = C.c_int64
        L.ndt_keyframe_count.argtypes = [vp]
        L.ndt_set_target_from_keyframes.argtypes = [vp, C.POINTER(C.c_int64), dp, C.c_int]
        L.ndt_set_source_from_keyframe.argtypes = [vp, C.c_int64]
        L.ndt_svn_default_params.restype = None
        L.ndt_svn_default_params.argtypes = [C.POINTER(SvnParams)]
        L.ndt_svn_sample_particles.argtypes = [dp, C.c_int, C.c_uint64, dp]
        L.ndt_svn_align.argtypes = [vp, C.POINTER(SvnParams), dp, dp, C.POINTER(SvnResult)]
        L.ndt_result_covariance.argtypes = [dp, C.c_double, C.c_int, dp]
        L.ndt_params_preset.argtypes = [C.POINTER(Params), C.c_int]
        L.ndt_score_transform.argtypes = [vp, fp, C.POINTER(Score)]
        L.ndt_comm_info.argtypes = [C.c_char_p, C.c_size_t]
        L.ndt_comm_rank_count.argtypes = [vp]
        L.ndt_comm_p2p_handle.argtypes = [vp, vp]
        L.ndt_comm_init_p2p.argtypes = [vp, vp, C.c_int, C.c_int]
        L.ndt_score_transforms.argtypes = [vp, fp, C.c_int, C.POINTER(Score)]
        L.ndt_xy_covariance_laplace.argtypes = [dp, dp]
        L.ndt_propose_poses_to_search.argtypes = [C.POINTER(Result), dp, dp, C.c_int, fp]
        L.ndt_xy_covariance_multi_ndt.argtypes = [vp, C.POINTER(Result), fp, C.c_int, dp, dp]
        L.ndt_xy_covariance_multi_ndt_score.argtypes = [vp, C.POINTER(Result), fp, C.c_int, C.c_double, dp, dp]
        L.ndt_voxel_downsample_device.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_float, vp, vp, vp, vp, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
        L.ndt_voxel_downsample.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_long, C.c_float, vp, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.ndt_get_iteration_history.argtypes = [vp, fp, dp, dp, C.c_int]
        L.ndt_set_handoff_mode.argtypes = [vp, C.c_int]
        L.ndt_get_handoff_mode.argtypes = [vp]
        L.ndt_wait.argtypes = [vp]
        L.ndt_get_handoff_timing.argtypes = [vp, C.POINTER(HandoffTiming)]
        L.ndt_debug_prelaunch_counters.argtypes = [vp, C.POINTER(C.c_int64)]  # test seam, not in the header
        L.ndt_debug_build_counters.argtypes = [vp, C.POINTER(C.c_int64)]  # test seam, not in the header
        L.ndt_debug_speculation_counters.argtypes = [vp, C.POINTER(C.c_int64)]  # test seam, not in the header
        if hasattr(L, "ndt_debug_handoff_counters"):   # (a library of an earlier round under NDT_HIP_LIB has none)
            L.ndt_debug_handoff_counters.argtypes = [vp, C.POINTER(C.c_int64)]  # test seam, not in the header
        L.ndt_debug_set_speculation.argtypes = [vp, C.c_int]  # tuning aid, not in the header
        L.ndt_debug_sort_pairs.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]  # test seam, not in the header
        _lib = L
    return _lib


def default_params(preset=None, **kw):
    p = Params()
    lib().ndt_default_params(C.byref(p))
    if preset is not None:
        rc = lib().ndt_params_preset(C.byref(p), int(preset))
        if rc != 0:
            raise NdtError(rc, "ndt_params_preset")
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def comm_info():
    """(ncclGetVersion code, path of the shared object serving the nccl* symbols of the engine)."""
    buf = C.create_string_buffer(512)
    v = lib().ndt_comm_info(buf, 512)
    return v, buf.value.decode()


def backend_info():
    buf = C.create_string_buffer(256)
    n = lib().ndt_backend_info(buf, 256)
    return n, buf.value.decode()


def shard_range(n, rank, nranks):
    b, c = C.c_size_t(), C.c_size_t()
    lib().ndt_shard_range(n, rank, nranks, C.byref(b), C.byref(c))
    return b.value, c.value


def _colmajor(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).ravel()


class ColMajor4f:
    """A 4x4 transform already converted to the ABI's 16 column-major floats (what
    Eigen::Matrix4f::data() is for the C++ callers): lets a loop convert its guess once."""

    def __init__(self, T):
        self.a = _colmajor(T)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def unpack_eval(words):
    """32 packed doubles -> dict(score, gradient[6], hessian[6,6], nvtl_sum, n_with, n_pairs)."""
    w = np.ascontiguousarray(words, dtype=np.float64)
    g = np.zeros(6)
    H = np.zeros(36)
    s = C.c_double()
    lib().ndt_unpack_eval(_dp(w), C.byref(s), _dp(g), _dp(H))
    return dict(score=s.value, gradient=g, hessian=H.reshape(6, 6), nvtl_sum=w[28],
                n_with_neighbors=int(w[29]), n_pairs=int(w[30]))


def result_to_dict(r):
    return dict(T=np.array(r.final_transformation[:], dtype=np.float64).reshape(4, 4).T.copy(),
                pose=np.array(r.final_pose[:]), converged=bool(r.converged),
                iterations=r.iterations, n_evaluations=r.n_evaluations,
                hessian=np.array(r.hessian[:]).reshape(6, 6), score=r.score,
                transform_probability=r.transform_probability,
                nvtl=r.nearest_voxel_transformation_likelihood, n_pairs=r.n_pairs,
                n_points_with_neighbors=r.n_points_with_neighbors, ms_total=r.ms_total,
                ms_device=r.ms_device, n_evaluations_reused=r.n_evaluations_reused)


class NormalDistributionsTransform:
    """pclomp::NormalDistributionsTransform-shaped engine on one MI355X."""

    def __init__(self, device_id=-1, **params):
        self._p = default_params(device_id=device_id, **params)
        self._h = C.c_void_p()
        rc = lib().ndt_create(C.byref(self._p), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise NdtError(rc, "ndt_create failed (a gfx950 device is required; no CPU fallback)")
        self._raw, self._cooked = None, None
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().ndt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise NdtError(rc, lib().ndt_last_error(self._h).decode())

    def _push(self):
        self._check(lib().ndt_set_params(self._h, C.byref(self._p)))

    # --- setters the drivers call (ref: run/pipeline.cpp:467-480) ---
    def setNumThreads(self, n): self._p.num_threads = int(n); self._push()
    def getNumThreads(self): return self._p.num_threads
    def setResolution(self, r): self._p.resolution = float(r); self._push()
    def getResolution(self): return self._p.resolution
    def setStepSize(self, s): self._p.step_size = float(s); self._push()
    def setTransformationEpsilon(self, e): self._p.trans_epsilon = float(e); self._push()
    def setMaximumIterations(self, n): self._p.max_iterations = int(n); self._push()
    def setOutlierRatio(self, o): self._p.outlier_ratio = float(o); self._push()
    def setNeighborhoodSearchMethod(self, m): self._p.search_method = int(m); self._push()
    def setMinPointPerVoxel(self, n): self._p.min_points_per_voxel = int(n); self._push()
    def setRegularizationScaleFactor(self, k): self._p.regularization_scale_factor = float(k); self._push()

    def setParams(self, **kw):
        for k, v in kw.items():
            if not hasattr(self._p, k):
                raise AttributeError(k)
            setattr(self._p, k, v)
        self._push()

    def setRegularizationPose(self, T):
        a = _colmajor(T)
        self._check(lib().ndt_set_regularization_pose(self._h, _fp(a)))

    def unsetRegularizationPose(self):
        self._check(lib().ndt_clear_regularization_pose(self._h))

    # --- clouds ---
    @staticmethod
    def _xyz(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("cloud must be N x >=3 float32")
        return a

    def debugSortPairs(self, keys, end_bit):
        """Test seam: the voxel build's stable radix sort on caller keys -> (sorted keys, permutation)."""
        k = np.ascontiguousarray(keys, dtype=np.uint32)
        ko, vo = np.empty_like(k), np.empty_like(k)
        self._check(lib().ndt_debug_sort_pairs(self._h, k.ctypes.data, len(k), int(end_bit), ko.ctypes.data,
                                               vo.ctypes.data))
        return ko, vo

    def setInputTarget(self, cloud):
        a = self._xyz(cloud)
        self._check(lib().ndt_set_target(self._h, a.ctypes.data, len(a), a.strides[0]))

    def setInputSource(self, cloud):
        a = self._xyz(cloud)
        self._check(lib().ndt_set_source(self._h, a.ctypes.data, len(a), a.strides[0]))
        self._n_src = len(a)

    def setInputTargetSoA(self, x, y, z):
        x, y, z = (np.ascontiguousarray(v, dtype=np.float32) for v in (x, y, z))
        self._check(lib().ndt_set_target_soa(self._h, x.ctypes.data, y.ctypes.data, z.ctypes.data, len(x)))

    def setInputSourceSoA(self, x, y, z):
        x, y, z = (np.ascontiguousarray(v, dtype=np.float32) for v in (x, y, z))
        self._check(lib().ndt_set_source_soa(self._h, x.ctypes.data, y.ctypes.data, z.ctypes.data, len(x)))
        self._n_src = len(x)

    def setInputTargetDevice(self, dx, dy, dz, n):
        """dx/dy/dz: integer device addresses of SoA float32 arrays already in HBM."""
        self._check(lib().ndt_set_target_device(self._h, dx, dy, dz, n))

    def setInputTargetDeviceDeferred(self, dx, dy, dz, n):
        """As setInputTargetDevice, but the voxel-grid build is only ENQUEUED (a steady-state build under the
        asynchronous hand-off): the arrays must stay valid and unchanged until the first call that needs the grid --
        align(), getGridInfo(), wait() ... -- has returned; a failed build is reported by that call."""
        self._check(lib().ndt_set_target_device_deferred(self._h, dx, dy, dz, n))

    def setInputSourceDevice(self, dx, dy, dz, n):
        self._check(lib().ndt_set_source_device(self._h, dx, dy, dz, n))

    def setInputSourceDeviceView(self, dx, dy, dz, n):
        """No copy: the device arrays must stay valid and unchanged until the source is replaced
        (pcl::Registration::setInputSource keeps the caller's shared_ptr the same way)."""
        self._check(lib().ndt_set_source_device_view(self._h, dx, dy, dz, n))
        self._n_src = int(n)

    def sourceChanged(self):
        """The arrays of a viewed source were rewritten in place: drop whatever the engine cached of them."""
        self._check(lib().ndt_source_changed(self._h))

    def setHandoffMode(self, mode):
        """HANDOFF_ASYNC (default): setInputTarget / setInputSource return once the host cloud has been consumed; the
        copies and the voxel-grid build finish behind, and a failing steady-state build is reported by the first call
        that needs the grid (or by wait()).  HANDOFF_SYNC: both block until the device has everything."""
        self._check(lib().ndt_set_handoff_mode(self._h, int(mode)))

    def getHandoffMode(self):
        return int(lib().ndt_get_handoff_mode(self._h))

    def wait(self):
        """Blocks until every hand-off in flight is complete; raises what a deferred build failed with."""
        self._check(lib().ndt_wait(self._h))

    def getHandoffTiming(self):
        t = HandoffTiming()
        self._check(lib().ndt_get_handoff_timing(self._h, C.byref(t)))
        lane = lambda l: {k: getattr(l, k) for k, _ in HandoffLaneTiming._fields_}  # noqa: E731
        return dict(target=lane(t.target), source=lane(t.source), ms_build_wait=t.ms_build_wait, mode=t.mode,
                    cpu_budget=t.cpu_budget, repack_workers=t.repack_workers)

    def setRecordFormat(self, fmt):
        """RECORDS_F64 (80-byte voxel records, default) or RECORDS_PACKED48 (f64 mean + f32 inverse covariance,
        three 16-byte loads per neighbour instead of five; ndt_hip.h)."""
        self._check(lib().ndt_set_record_format(self._h, int(fmt)))

    def getRecordFormat(self):
        return int(lib().ndt_get_record_format(self._h))

    # --- multi-grid target [RECALLED: tier4 MultiGridNormalDistributionsTransform] ---
    def addTarget(self, cloud, target_id):
        """Voxelises `cloud` on its own and keeps its leaves under `target_id`."""
        a = self._xyz(cloud)
        self._check(lib().ndt_multigrid_add_target(self._h, int(target_id), a.ctypes.data, len(a), a.strides[0]))

    def removeTarget(self, target_id):
        self._check(lib().ndt_multigrid_remove_target(self._h, int(target_id)))

    def targetCount(self):
        return int(lib().ndt_multigrid_count(self._h))

    def createVoxelKdtree(self):
        """The union of all stored grids becomes the target (radius search over every grid's leaves)."""
        self._check(lib().ndt_multigrid_create_kdtree(self._h))

    # --- device-resident keyframe archive (ref: run/pipeline.cpp:784, run/pipeline_ligo_tc.cpp:519-529) ---
    def putKeyframe(self, kf_id, cloud):
        a = self._xyz(cloud)
        self._check(lib().ndt_keyframe_put(self._h, int(kf_id), a.ctypes.data, len(a), a.strides[0]))

    def setInputSourceFromKeyframe(self, kf_id):
        self._check(lib().ndt_set_source_from_keyframe(self._h, int(kf_id)))

    def eraseKeyframe(self, kf_id):
        self._check(lib().ndt_keyframe_erase(self._h, int(kf_id)))

    def keyframeCount(self):
        return int(lib().ndt_keyframe_count(self._h))

    def setInputTargetFromKeyframes(self, ids, poses):
        """target = sum of archived scans, each moved by its 4x4 double pose, built on the device."""
        ids_a = (C.c_int64 * len(ids))(*[int(i) for i in ids])
        p = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T for T in poses])).ravel()
        self._check(lib().ndt_set_target_from_keyframes(self._h, ids_a, _dp(p), len(ids)))

    # --- pcl::VoxelGrid downsample (ref: run/pipeline_ins_map_distribution.cpp:324-340) ---
    def voxelDownsampleDevice(self, dx, dy, dz, n, leaf, ox, oy, oz, cap, d_intensity=None, o_intensity=None):
        """SoA device arrays in, centroids of the occupied voxels (ascending voxel index) out; returns their number.
        The output arrays can be handed to setInputTargetDevice as they are."""
        m = C.c_size_t(0)
        self._check(lib().ndt_voxel_downsample_device(self._h, dx, dy, dz, d_intensity, int(n), float(leaf), ox, oy, oz,
                                                      o_intensity, int(cap), C.byref(m)))
        return int(m.value)

    def voxelDownsample(self, cloud, leaf, intensity_column=None):
        """Host cloud (N x >= 3 float32; intensity_column: index of the intensity field, 4 for pcl::PointXYZI's
        8-float layout) -> the filtered cloud in the same layout (other columns zero)."""
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("cloud must be N x >=3 float32")
        out = np.zeros_like(a)
        if len(a) == 0:
            return out
        m = C.c_size_t(0)
        off = -1 if intensity_column is None else 4 * int(intensity_column)
        self._check(lib().ndt_voxel_downsample(self._h, a.ctypes.data, len(a), a.strides[0], off, float(leaf),
                                               out.ctypes.data, len(out), C.byref(m)))
        return out[:m.value]

    def setGlobalSourceSize(self, n):
        self._check(lib().ndt_set_global_source_size(self._h, int(n)))

    # --- registration ---
    def align(self, guess=None, return_transform=True):
        """align(guess) -> final 4x4.  return_transform=False skips building the NumPy result
        (a tight loop reads the few scalars it needs through the getters below)."""
        if guess is None:
            g = _colmajor(np.eye(4))
        elif isinstance(guess, ColMajor4f):
            g = guess.a
        else:
            g = _colmajor(guess)
        r = Result()
        self._check(lib().ndt_align(self._h, _fp(g), C.byref(r)))
        self._raw, self._cooked = r, None
        return self._result["T"] if return_transform else None

    computeTransformation = align

    @property
    def _result(self):
        if self._cooked is None and self._raw is not None:
            self._cooked = result_to_dict(self._raw)
        return self._cooked

    def getFinalTransformation(self): return self._result["T"]
    def hasConverged(self): return bool(self._raw.converged)
    def getFinalNumIteration(self): return self._raw.iterations
    def getNumEvaluations(self): return self._raw.n_evaluations
    def getTransformationProbability(self): return self._raw.transform_probability
    def getNearestVoxelTransformationLikelihood(self): return self._raw.nearest_voxel_transformation_likelihood

    def scoreTransform(self, T=None):
        """Scoring-only evaluation of the current source under T (default identity): dict(score,
        transform_probability, nvtl, n_pairs, n_points_with_neighbors)."""
        g = _colmajor(np.eye(4) if T is None else T)
        sc = Score()
        self._check(lib().ndt_score_transform(self._h, _fp(g), C.byref(sc)))
        return dict(score=sc.score, transform_probability=sc.transform_probability,
                    nvtl=sc.nearest_voxel_transformation_likelihood, n_pairs=sc.n_pairs,
                    n_points_with_neighbors=sc.n_points_with_neighbors)

    def scoreTransforms(self, transforms):
        """K transforms scored in one launch: list of dicts like scoreTransform()."""
        t = np.ascontiguousarray(np.stack([_colmajor(T) for T in transforms]), dtype=np.float32)
        K = len(t)
        out = (Score * K)()
        self._check(lib().ndt_score_transforms(self._h, _fp(t), K, out))
        return [dict(score=o.score, transform_probability=o.transform_probability,
                     nvtl=o.nearest_voxel_transformation_likelihood, n_pairs=o.n_pairs,
                     n_points_with_neighbors=o.n_points_with_neighbors) for o in out]

    # --- 2-D covariance estimators of tier4 ndt_omp [RECALLED] ---
    def proposePosesToSearch(self, offsets_x, offsets_y):
        """Offsets rotated onto the principal axes of the Laplace covariance of the LAST result."""
        ox = np.ascontiguousarray(offsets_x, dtype=np.float64)
        oy = np.ascontiguousarray(offsets_y, dtype=np.float64)
        out = np.zeros((len(ox), 16), np.float32)
        self._check(lib().ndt_propose_poses_to_search(C.byref(self._raw), _dp(ox), _dp(oy), len(ox), _fp(out)))
        return [out[i].reshape(4, 4).T.astype(np.float64) for i in range(len(ox))]

    def estimateXYCovarianceMultiNdt(self, poses):
        main = Result.from_buffer_copy(self._raw)
        t = np.ascontiguousarray(np.stack([_colmajor(T) for T in poses]), dtype=np.float32)
        mean, cov = np.zeros(2), np.zeros(4)
        self._check(lib().ndt_xy_covariance_multi_ndt(self._h, C.byref(main), _fp(t), len(t), _dp(mean), _dp(cov)))
        return mean, cov.reshape(2, 2)

    def estimateXYCovarianceMultiNdtScore(self, poses, temperature):
        main = Result.from_buffer_copy(self._raw)
        t = np.ascontiguousarray(np.stack([_colmajor(T) for T in poses]), dtype=np.float32)
        mean, cov = np.zeros(2), np.zeros(4)
        self._check(lib().ndt_xy_covariance_multi_ndt_score(self._h, C.byref(main), _fp(t), len(t), float(temperature),
                                                            _dp(mean), _dp(cov)))
        return mean, cov.reshape(2, 2)

    def calculateTransformationProbability(self, cloud, T=None):
        """pclomp's scoring-only call [RECALLED]: score / #points of `cloud` (already transformed,
        or moved by T) against the current target."""
        self.setInputSource(cloud)
        return self.scoreTransform(T)["transform_probability"]

    def calculateNearestVoxelTransformationLikelihood(self, cloud, T=None):
        self.setInputSource(cloud)
        return self.scoreTransform(T)["nvtl"]

    def getResult(self):
        """pclomp::NdtResult: iteration_num, hessian, pose, transform_probability, nvtl."""
        r = dict(self._result)
        r["iteration_num"] = r["iterations"]
        return r

    def getIterationHistory(self):
        """pclomp::NdtResult's per-iteration arrays [RECALLED] of the last align: (transformation_array [n,4,4],
        transform_probability_array [n], nearest_voxel_transformation_likelihood_array [n]); entry 0 = the guess."""
        n = lib().ndt_get_iteration_history(self._h, None, None, None, 0)
        if n < 0:
            raise NdtError(n, "ndt_get_iteration_history")
        T = np.zeros((n, 16), np.float32)
        tp, nv = np.zeros(n), np.zeros(n)
        if n:
            lib().ndt_get_iteration_history(self._h, _fp(T), _dp(tp), _dp(nv), n)
        return T.reshape(n, 4, 4).transpose(0, 2, 1).astype(np.float64), tp, nv

    def evalDerivatives(self, poses6, transforms=None, compute_hessian=True):
        """computeDerivatives at K poses (K x 6); returns a list of dicts."""
        p = np.ascontiguousarray(np.atleast_2d(poses6), dtype=np.float64)
        K = p.shape[0]
        t = None
        if transforms is not None:
            t = np.ascontiguousarray(np.stack([_colmajor(T) for T in transforms]), dtype=np.float32)
        out = np.zeros((K, EVAL_WORDS))
        self._check(lib().ndt_eval_derivatives(self._h, _dp(p), _fp(t) if t is not None else None, K,
                                               int(compute_hessian), _dp(out)))
        return [unpack_eval(out[k]) for k in range(K)]

    def transformSource(self, T):
        """The `output` cloud of align(): the source transformed by T on the device."""
        n = getattr(self, "_n_src", 0)
        out = np.zeros((n, 3), np.float32)
        a = _colmajor(T)
        self._check(lib().ndt_transform_source(self._h, _fp(a), _fp(out), n))
        return out

    # --- voxel grid accessors (ref: include/pipeline.hpp:175-206) ---
    def getGridInfo(self):
        gi = GridInfo()
        self._check(lib().ndt_get_grid_info(self._h, C.byref(gi)))
        return dict(min_b=np.array(gi.min_b[:]), max_b=np.array(gi.max_b[:]), div_b=np.array(gi.div_b[:]),
                    leaf_size=gi.leaf_size, n_leaves=gi.n_leaves, n_cells=gi.n_cells,
                    n_target_points=gi.n_target_points, ms_build=gi.ms_build)

    def getLeaves(self):
        """getTargetCells().getLeaves(): dict of arrays, valid leaves, ascending voxel index."""
        n = int(self.getGridInfo()["n_leaves"])
        buf = (Leaf * max(n, 1))()
        got = lib().ndt_export_leaves(self._h, buf, n)
        if got < 0:
            raise NdtError(int(got), lib().ndt_last_error(self._h).decode())
        a = np.frombuffer(buf, dtype=np.dtype(Leaf), count=got) if got else None
        if a is None:
            return dict(cell=np.zeros(0, np.int64), count=np.zeros(0, np.int32), center=np.zeros((0, 3), np.float32),
                        mean=np.zeros((0, 3)), cov=np.zeros((0, 3, 3)), icov=np.zeros((0, 3, 3)),
                        evecs=np.zeros((0, 3, 3)), evals=np.zeros((0, 3)))
        return dict(cell=a["index"].copy(), count=a["point_count"].copy(), center=a["center"].copy(),
                    mean=a["mean"].copy(), cov=a["cov"].reshape(-1, 3, 3).copy(),
                    icov=a["icov"].reshape(-1, 3, 3).copy(), evecs=a["evecs"].reshape(-1, 3, 3).copy(),
                    evals=a["evals"].copy())

    def getMinPointPerVoxel(self): return max(3, self._p.min_points_per_voxel)

    # --- multi-GPU ---
    def commInitRccl(self, id128, rank, nranks):
        buf = C.create_string_buffer(bytes(id128), 128)
        self._check(lib().ndt_comm_init_rccl(self._h, buf, rank, nranks))

    def commInitShm(self, name, rank, nranks):
        self._check(lib().ndt_comm_init_shm(self._h, name.encode(), rank, nranks))

    def commP2pHandle(self):
        """This rank's exchange area of the peer-write reducer as a 64-byte IPC handle (all-gather them)."""
        buf = C.create_string_buffer(64)
        self._check(lib().ndt_comm_p2p_handle(self._h, buf))
        return buf.raw

    def commInitP2p(self, handles, rank, nranks):
        """handles: the nranks x 64 bytes of every rank's commP2pHandle(), in rank order."""
        b = bytes(handles)
        if len(b) != 64 * nranks:
            raise ValueError("need %d handle bytes" % (64 * nranks))
        buf = C.create_string_buffer(b, len(b))
        self._check(lib().ndt_comm_init_p2p(self._h, buf, rank, nranks))

    def commInitHook(self, fn, rank, nranks):
        cb = ALLREDUCE_FN(fn)
        self._keep.append(cb)
        self._check(lib().ndt_comm_init_hook(self._h, cb, None, rank, nranks))

    def commRankCount(self):
        """Ranks of the live reducer as the transport reports them (RCCL: ncclCommCount)."""
        n = lib().ndt_comm_rank_count(self._h)
        if n < 0:
            raise NdtError(n, "ndt_comm_rank_count")
        return n

    def commP2pSelftest(self, rounds=10000):
        """COLLECTIVE slot-integrity pass of the peer-write reducer: dict(rounds, torn, missed, longest_us)."""
        out = (C.c_int64 * 4)()
        self._check(lib().ndt_comm_p2p_selftest(self._h, int(rounds), out))
        return dict(rounds=int(out[0]), torn=int(out[1]), missed=int(out[2]), longest_us=out[3] * 0.01)

    def commP2pStats(self, reset=False):
        """In-kernel exchanges since init / the last reset: dict(exchanges, mean_us, max_us, late)."""
        out = (C.c_int64 * 4)()
        self._check(lib().ndt_comm_p2p_stats(self._h, out, int(reset)))
        n = int(out[0])
        return dict(exchanges=n, mean_us=(out[1] * 0.01 / n) if n else 0.0, max_us=out[2] * 0.01, late=int(out[3]))

    def commDestroy(self):
        self._check(lib().ndt_comm_destroy(self._h))

    # --- instrumentation ---
    def setKeepWarm(self, period_us):
        """ndt_set_keepwarm: an idle-time heartbeat every period_us (0: off, the default)."""
        self._check(lib().ndt_set_keepwarm(self._h, int(period_us)))

    def keepWarm(self):
        """(period_us, beats launched so far)."""
        b = C.c_longlong(0)
        return lib().ndt_get_keepwarm(self._h, C.byref(b)), b.value

    def enableKernelTiming(self, on=True):
        self._check(lib().ndt_enable_kernel_timing(self._h, int(on)))

    def prelaunchCounters(self):
        """(evaluations served by a pre-launched kernel, pre-launched kernels told to leave, time-outs)."""
        out = (C.c_int64 * 8)()
        self._check(lib().ndt_debug_prelaunch_counters(self._h, out))
        return tuple(out)[:3]

    def autoStreamPlacement(self):
        """(1 if NDT_PRELAUNCH_AUTO has settled on the one-stream placement of waiting kernels, switches so far)."""
        out = (C.c_int64 * 8)()
        self._check(lib().ndt_debug_prelaunch_counters(self._h, out))
        return int(out[6]), int(out[7])

    def speculationCounters(self):
        """First evaluations of an align enqueued behind a build still in flight: (kept, discarded)."""
        out = (C.c_int64 * 2)()
        self._check(lib().ndt_debug_speculation_counters(self._h, out))
        return int(out[0]), int(out[1])

    def setSpeculation(self, on):
        """Tuning aid: the first-evaluation-behind-the-build short cut on / off for this handle."""
        self._check(lib().ndt_debug_set_speculation(self._h, 1 if on else 0))

    def lostRowRetries(self):
        """Evaluations repeated through the ticketed final sum after the summing block had given up waiting for a row."""
        out = (C.c_int64 * 8)()
        self._check(lib().ndt_debug_prelaunch_counters(self._h, out))
        return int(out[5])

    def prelaunchOverlapped(self):
        """Pre-launched kernels that were enqueued on the other stream (resident before their predecessor ended)."""
        out = (C.c_int64 * 8)()
        self._check(lib().ndt_debug_prelaunch_counters(self._h, out))
        return int(out[3])

    def p2pHostFinishes(self):
        """Peer-write evaluations whose cross-rank exchange the host had to finish (a peer's row was > 20 ms late)."""
        out = (C.c_int64 * 8)()
        self._check(lib().ndt_debug_prelaunch_counters(self._h, out))
        return int(out[4])

    def handoffCounters(self):
        """(host hand-offs of a target whose partition launch ran under the transfer, launches of tile ranges they made)."""
        out = (C.c_int64 * 2)()
        self._check(lib().ndt_debug_handoff_counters(self._h, out))
        return int(out[0]), int(out[1])

    def buildCounters(self):
        """(builds that fell back from the fused sort passes to the classic ones, two-launch bucketed builds
        that were declined and repeated sort-based, builds that went through in two launches)."""
        out = (C.c_int64 * 3)()
        self._check(lib().ndt_debug_build_counters(self._h, out))
        return tuple(out)

    def getTiming(self):
        t = Timing()
        self._check(lib().ndt_get_timing(self._h, C.byref(t)))
        return dict(ms_last_eval_kernel=t.ms_last_eval_kernel, ms_last_reduce_kernel=t.ms_last_reduce_kernel,
                    ms_last_build=t.ms_last_build, n_eval_launches=t.n_eval_launches,
                    ms_eval_kernel_total=t.ms_eval_kernel_total,
                    ms_reduce_kernel_total=t.ms_reduce_kernel_total, n_timed_evals=t.n_timed_evals)


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().ndt_comm_unique_id(buf)
    if rc != 0:
        raise NdtError(rc, "ndt_comm_unique_id")
    return buf.raw


def xy_covariance_laplace(hessian):
    """-(H[0:2,0:2])^-1 [RECALLED tier4 estimate_xy_covariance_by_Laplace_approximation]."""
    H = np.ascontiguousarray(hessian, dtype=np.float64).reshape(36)
    out = np.zeros(4)
    rc = lib().ndt_xy_covariance_laplace(_dp(H), _dp(out))
    if rc != 0:
        raise NdtError(rc, "xy block of the Hessian is singular or not finite")
    return out.reshape(2, 2)


def result_covariance(hessian, eps=1e-6, gtsam_order=True):
    """-(H + eps I)^-1 of an align() Hessian, optionally in the block order the drivers hand to
    GTSAM (ref: run/pipeline.cpp:594-603, src/registercallback.cpp:170-186)."""
    H = np.ascontiguousarray(hessian, dtype=np.float64).reshape(36)
    out = np.zeros(36)
    rc = lib().ndt_result_covariance(_dp(H), float(eps), 1 if gtsam_order else 0, _dp(out))
    if rc != 0:
        raise NdtError(rc, "Hessian + eps I is singular or not finite")
    return out.reshape(6, 6)


def newton_align(params, n_source_total, guess, eval_fn, regularization_pose=None):
    """Host Newton/More-Thuente driver with an external evaluator.

    eval_fn(pose6 ndarray, T 4x4 ndarray, compute_hessian bool) -> 32 packed doubles.
    """
    def _cb(_ctx, pose_p, T_p, need_h, out_p):
        try:
            pose = np.array([pose_p[i] for i in range(6)])
            T = np.array([T_p[i] for i in range(16)], dtype=np.float32).reshape(4, 4).T
            w = np.asarray(eval_fn(pose, T, bool(need_h)), dtype=np.float64)
            for i in range(EVAL_WORDS):
                out_p[i] = float(w[i])
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -1
    cb = EVAL_FN(_cb)
    g = _colmajor(guess)
    reg = _colmajor(regularization_pose) if regularization_pose is not None else None
    r = Result()
    rc = lib().ndt_newton_align(C.byref(params), int(n_source_total), _fp(g),
                                _fp(reg) if reg is not None else None, cb, None, C.byref(r))
    if rc != 0:
        raise NdtError(rc, "ndt_newton_align")
    return result_to_dict(r)


def pack_eval(score, gradient, hessian, nvtl_sum=0.0, n_with=0, n_pairs=0):
    """Inverse of unpack_eval (for external evaluators)."""
    w = np.zeros(EVAL_WORDS)
    w[0] = score
    w[1:7] = gradient
    H = np.asarray(hessian).reshape(6, 6)
    k = 7
    for i in range(6):
        for j in range(i, 6):
            w[k] = H[i, j]
            k += 1
    w[28], w[29], w[30] = nvtl_sum, n_with, n_pairs
    return w


def _pose16d(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).ravel()


def svn_sample_particles(prior, K, seed=0):
    """K x 4 x 4 initial particles around `prior` (ref: svn_ndt_impl.hpp:708-716, seed explicit)."""
    out = np.zeros(16 * K)
    rc = lib().ndt_svn_sample_particles(_dp(_pose16d(prior)), K, seed, _dp(out))
    if rc != 0:
        raise NdtError(rc, "ndt_svn_sample_particles")
    return out.reshape(K, 4, 4).transpose(0, 2, 1).copy()


class SvnNormalDistributionsTransform(NormalDistributionsTransform):
    """svn_ndt::SvnNormalDistributionsTransform-shaped engine (ref: extern/svn_ndt/include/
    svn_ndt.h:100-182; driver run/pipeline_lo_svn.cpp:301-319,387-388): K pose particles,
    Stage 1 as one batched kernel launch."""

    def __init__(self, device_id=-1, **params):
        base = dict(hessian_mode=HESSIAN_GAUSS_NEWTON, add_ridge=1)  # svn_ndt.h:314, impl :650-653
        base.update(params)
        super().__init__(device_id=device_id, **base)
        self._sp = SvnParams()
        lib().ndt_svn_default_params(C.byref(self._sp))

    def setParticleCount(self, k): self._sp.particle_count = int(k)
    def setMaxIterations(self, n): self._sp.max_iterations = int(n)
    def setKernelBandwidth(self, h): self._sp.kernel_bandwidth = float(h)
    def setStepSize(self, s): self._sp.step_size = float(s)
    def setEarlyStopThreshold(self, t): self._sp.stop_threshold = float(t)
    def setUseGaussNewtonHessian(self, on):
        self._p.hessian_mode = HESSIAN_GAUSS_NEWTON if on else HESSIAN_FULL
        self._push()

    def align(self, source_cloud, prior_pose, particles=None, seed=0):
        """SvnNdtResult as a dict: final_pose, final_covariance ([rot, trans] order), converged,
        iterations (+ the final particles and stage timings)."""
        self.setInputSource(source_cloud)
        K = self._sp.particle_count
        if particles is None:
            particles = svn_sample_particles(prior_pose, K, seed)
        part = np.ascontiguousarray(np.asarray(particles, dtype=np.float64).transpose(0, 2, 1)).ravel().copy()
        if len(part) != 16 * K:
            raise ValueError("need %d particles" % K)
        r = SvnResult()
        self._check(lib().ndt_svn_align(self._h, C.byref(self._sp), _dp(_pose16d(prior_pose)), _dp(part),
                                        C.byref(r)))
        return dict(final_pose=np.array(r.final_pose[:]).reshape(4, 4).T.copy(),
                    final_covariance=np.array(r.final_covariance[:]).reshape(6, 6),
                    converged=bool(r.converged), iterations=r.iterations,
                    last_mean_update=r.last_mean_update, ms_total=r.ms_total, ms_stage1=r.ms_stage1,
                    ms_stage2=r.ms_stage2, ms_stage3=r.ms_stage3,
                    particles=part.reshape(K, 4, 4).transpose(0, 2, 1).copy())
