"""ctypes binding of libmvkpconv.so (C ABI: include/mvkpconv.h).

There is NO fallback: if the HIP library is missing or stale this module raises. PyTorch is used
only as the owner of device memory and streams -- every signature is plain pointers and sizes.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvkpconv.so")
ABI_VERSION = 8

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float



class BnFwdProblem(C.Structure):
    """mvk_bn_fwd_problem (include/mvkpconv.h): the arguments of mvk_bn_lrelu_fwd for one of two paired problems."""
    _fields_ = [("x", _vp), ("n_valid", _vp), ("R", _i64), ("D", C.c_int32), ("gamma", _vp), ("beta", _vp), ("eps", _f),
                ("momentum", _f), ("slope", _f), ("running_mean", _vp), ("running_var", _vp), ("mean", _vp), ("invstd", _vp),
                ("scratch2D", _vp), ("y", _vp), ("num_batches_tracked", _vp), ("addend", _vp), ("ext_part", _vp),
                ("ext_rows", C.c_int32)]


class BnBwdProblem(C.Structure):
    """mvk_bn_bwd_problem: the arguments of mvk_bn_lrelu_bwd."""
    _fields_ = [("x", _vp), ("g", _vp), ("n_valid", _vp), ("R", _i64), ("D", C.c_int32), ("gamma", _vp), ("beta", _vp),
                ("mean", _vp), ("invstd", _vp), ("slope", _f), ("scratch", _vp), ("dgamma_dbeta", _vp), ("dx", _vp),
                ("y_out", _vp), ("d_addend", _vp)]


class RevList(C.Structure):
    """mvk_rev_list: one reverse list for mvk_reverse_finish_many."""
    _fields_ = [("rev", _vp), ("counts", _vp), ("status", _vp), ("rows", _i64), ("width", C.c_int32), ("shadow", C.c_int32)]


class RegLayer(C.Structure):
    """mvk_reg_layer: one deformable layer for mvk_deform_regularizer_many."""
    _fields_ = [("min_d2", _vp), ("deformed_kp", _vp), ("n_valid", _vp), ("d_min_d2", _vp), ("d_deformed_kp", _vp),
                ("N", _i64), ("extent", _f), ("repulse_extent", _f), ("power", _f)]


class BnFinish(C.Structure):
    """mvk_bn_finish: the producer's half of a folded BatchNorm (statistics finished inside the GEMM launch)."""
    _fields_ = [("counters", _vp), ("eps", _f), ("momentum", _f), ("mean", _vp), ("invstd", _vp), ("running_mean", _vp),
                ("running_var", _vp), ("num_batches_tracked", _vp)]


class ATransform(C.Structure):
    """mvk_a_transform: the consumer's half (BatchNorm + LeakyReLU applied while the A operand is staged)."""
    _fields_ = [("mean", _vp), ("invstd", _vp), ("gamma", _vp), ("beta", _vp), ("slope", _f), ("n_valid", _vp), ("out", _vp)]


_SIGNATURES = {
    "mvk_abi_version": (C.c_int, []),
    "mvk_last_error": (C.c_char_p, []),
    "mvk_kpconv_gather_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _i, _vp, _i, _f, _i, _i,
                                        _vp, _vp, _vp, _vp, _vp]),
    "mvk_kpconv_gather_fwd_ordered": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _i, _vp, _i, _f, _i, _i,
                                                _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvk_gemm_f32_stream_plan": (C.c_int, [_i64, _i, _i64, _vp]),
    "mvk_gemm_f32_stream": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i, _i64, _vp, _vp, _vp]),
    "mvk_bn_single_launch_rows": (C.c_int, [_i]),
    "mvk_kpconv_gather_plan": (C.c_int, [_i64, _i64, _i, _i, _i, _i, _vp]),
    "mvk_kpconv_scatter_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _vp, _i, _f, _i, _i,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvk_gemm_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _vp]),
    "mvk_gemm_f32_plan": (C.c_int, [_i64, _i64, _i64, _i, _i, _vp, _vp]),
    "mvk_gemm_f32_kp_transposed": (C.c_int, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "mvk_reverse_neighbors": (C.c_int, [_vp, _i, _i64, _i, _i64, _i64, _vp, _i, C.c_int32, _i, _vp, _vp, _vp]),
    "mvk_gather_sum_rows": (C.c_int, [_vp, _i64, _i64, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    "mvk_max_pool_bwd_gather": (C.c_int, [_vp, _vp, _vp, _i, _i, _i64, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    "mvk_gemm_f32_ldb": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mvk_gemm_split_arena": (C.c_int, [_vp, _i64, _vp, _i64]),
    "mvk_gemm_split_ordered": (C.c_int, []),
    "mvk_gemm_f32_bias_act": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _vp, _f, _vp]),
    "mvk_gemm_f32_scatter_cat": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i, _i64, _i64, _i, _vp, _vp, _vp]),
    "mvk_gemm_f32_dual_plan": (C.c_int, [_i64, _i64, _i64, _i64, _vp]),
    "mvk_gemm_f32_dual": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mvk_gemm_f32_pair_plan": (C.c_int, [_i64, _i64, _i64, _i64, _i, _vp]),
    "mvk_gemm_f32_pair": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "mvk_gemm_f32_ex": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mvk_gemm_f32_bn": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _vp, _vp, C.POINTER(BnFinish), C.POINTER(ATransform), _vp]),
    "mvk_gemm_f32_pair_bn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp,
                                       C.POINTER(BnFinish), C.POINTER(BnFinish), _vp]),
    "mvk_gemm_group_entry_bytes": (C.c_int64, []),
    "mvk_gemm_f32_tn_grouped_split": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "mvk_gemm_f32_tn_grouped_plan": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvk_gemm_f32_tn_grouped": (C.c_int, [_vp, _i, _i, _i64, _i64, _vp]),
    "mvk_kpconv_gather_rev_deform": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "mvk_kpconv_deform_doff": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp,
                                        _vp, _vp]),
    "mvk_deform_regularizer": (C.c_int, [_vp, _vp, _vp, _i64, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "mvk_deform_regularizer_ex": (C.c_int, [_vp, _vp, _vp, _i64, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "mvk_bias_act_nhwc": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mvk_sgd_chunk_elems": (C.c_int, []),
    "mvk_sgd_clip_step": (C.c_int, [_vp, _vp, _i64, _f, _f, _i, _vp]),
    "mvk_deform_operands_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "mvk_deform_operands_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "mvk_xent_workspace_floats": (C.c_int64, [_i64]),
    "mvk_xent_fwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "mvk_xent_bwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mvk_bn_lrelu_fwd": (C.c_int, [_vp, _vp, _i64, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _i, _vp]),
    "mvk_bn_lrelu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvk_bn_lrelu_fwd_pair": (C.c_int, [C.POINTER(BnFwdProblem), C.POINTER(BnFwdProblem), _vp]),
    "mvk_bn_lrelu_bwd_pair": (C.c_int, [C.POINTER(BnBwdProblem), C.POINTER(BnBwdProblem), _vp]),
    "mvk_bias_lrelu_fwd": (C.c_int, [_vp, _vp, _i64, _i, _f, _vp, _vp]),
    "mvk_bias_lrelu_bwd": (C.c_int, [_vp, _vp, _i64, _i, _f, _vp, _vp, _vp]),
    "mvk_add_lrelu_fwd": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp]),
    "mvk_add_lrelu_bwd": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp]),
    "mvk_max_pool_fwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    "mvk_max_pool_bwd": (C.c_int, [_vp, _vp, _vp, _i, _i64, _i, _i64, _i, _vp, _vp]),
    "mvk_gather_rows_fwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _i64, _i64, _vp, _vp]),
    "mvk_gather_rows_bwd": (C.c_int, [_vp, _vp, _i, _i64, _i64, _i64, _i, _vp, _vp]),
    "mvk_gather_rows_bwd_ld": (C.c_int, [_vp, _i64, _vp, _i, _i64, _i64, _i64, _i, _vp, _vp]),
    "mvk_gather_rows_cat_fwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _i64, _i64, _vp, _i, _vp, _vp]),
    "mvk_gather_rows_cat_bwd": (C.c_int, [_vp, _vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp]),
    "mvk_grid_subsample_workspace": (C.c_int64, [_i64, _i, _i, _i]),
    "mvk_grid_subsample_batch": (C.c_int, [_vp, _i64, _vp, _i, _vp, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _i64, _vp]),
    "mvk_grid_subsample_batch_oriented": (C.c_int, [_vp, _i64, _vp, _i, _vp, _vp, _i, _vp, _i, _f, _i, _vp, _vp,
                                                    _vp, _vp, _vp, _vp, _i64, _vp]),
    "mvk_grid_subsample_batch_dev": (C.c_int, [_vp, _i64, _vp, _i, _vp, _f, _vp, _i64, _f, _vp, _vp, _vp, _vp, _i64,
                                               _vp]),
    "mvk_radius_neighbors_dev": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _f, _vp, _i, _i, _vp, _i, _vp, _i64,
                                           _vp]),
    "mvk_radius_neighbors_dev_rev": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _f, _vp, _i, _i, _vp, _i, _vp, _i64,
                                               _vp, _i, _vp, _vp, _vp]),
    "mvk_reverse_finish_many": (C.c_int, [C.POINTER(RevList), _i, _vp]),
    "mvk_deform_regularizer_many": (C.c_int, [C.POINTER(RegLayer), _i, _i, _vp, _vp, _vp]),
    "mvk_neighbors_cell_order": (C.c_int, [_i64, _i, _vp, _vp, _i64, _vp, _i64, _vp]),
    "mvk_radius_neighbors_workspace": (C.c_int64, [_i64, _i64, _i]),
    "mvk_radius_neighbors_batch": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _f, _vp, _i, _vp,
                                             _vp, _i64, _vp]),
    "mvk_radius_neighbors_enqueue": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _f, _vp, _i, _vp, _i,
                                               _vp, _i64, _vp]),
    "mvk_pad_points": (C.c_int, [_vp, _i64, _vp, _i64, _i, _f, _vp, _vp]),
    "mvk_pad_index_rows": (C.c_int, [_vp, _i, _i64, _i, _i64, _vp, _i64, _i, _i64, _vp]),
    "mvk_unproject_depth": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mvk_knn_workspace": (C.c_int64, [_i64, _i64, _i]),
    "mvk_knn_f64": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "mvk_ball_query_workspace": (C.c_int64, [_i64]),
    "mvk_ball_query": (C.c_int, [_vp, _i64, _vp, C.c_double, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mvk_tukey_update": (C.c_int, [_vp, _i64, _vp, C.c_double, _vp, _vp]),
    "mvk_fa_gather_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
    "mvk_fa_gather_fwd_ex": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
    "mvk_group_points_fwd": (C.c_int, [_vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
    "mvk_group_points_bwd": (C.c_int, [_vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
    "mvk_group_points_fwd_f64": (C.c_int, [_vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
    "mvk_group_points_bwd_f64": (C.c_int, [_vp, _vp, _i, _i, _i64, _i64, _i, _vp, _vp]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None
_owner_pid = None

FORK_MESSAGE = ("the MV-KPConv HIP library was initialised in process %d and is being called from its forked child %d: "
                "a HIP context does not survive fork(). Start DataLoader workers with multiprocessing_context='spawn', "
                "or let the workers emit only stacked points / lengths and build the pyramid "
                "(datasets.common.segmentation_inputs_sphere) in the main process -- see INTEGRATION.md")


def _check_process():
    """Loud failure instead of a hang when a forked worker (the reference builds the pyramid in
    num_workers forked DataLoader processes, train_ScanNet_sphere.py:365-377) calls into the library."""
    pid = os.getpid()
    if _owner_pid is not None and pid != _owner_pid:
        raise RuntimeError(FORK_MESSAGE % (_owner_pid, pid))
    try:
        import torch
        if torch.cuda._is_in_bad_fork():
            raise RuntimeError(FORK_MESSAGE % (os.getppid(), pid))
    except (ImportError, AttributeError):
        pass


def lib():
    """The loaded library. Raises (never falls back) when it is absent or has the wrong ABI, or when the
    caller is a forked child of the process that initialised it."""
    global _lib, _owner_pid
    _check_process()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmvkpconv.so is not built (%s). Run `python __graft_entry__.py build` -- there is "
                "no CPU fallback for the MV-KPConv hot path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError = missing export = stale build: loud
            fn.restype = res
            fn.argtypes = args
        if l.mvk_abi_version() != ABI_VERSION:
            raise RuntimeError("libmvkpconv.so ABI %d != expected %d; rebuild" % (l.mvk_abi_version(), ABI_VERSION))
        _lib = l
        _owner_pid = os.getpid()
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().mvk_last_error().decode("utf-8", "replace"))
