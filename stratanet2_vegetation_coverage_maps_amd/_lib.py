"""ctypes binding of libstrata_hip.so (include/strata_hip.h).  No torch C++ extension, no fallback: if the library
is missing the product path raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_long, c_longlong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libstrata_hip.so")

SN2_VERSION = 100
MAX_NEIGHBORS = 2000  # model/point_net2.py:24
STAT_SLOTS = 1024     # SN2_STAT_SLOTS


class Block(Structure):  # sn2_block
    _fields_ = [("cin", c_int), ("cout", c_int), ("W", c_void_p), ("b", c_void_p), ("gamma", c_void_p),
                ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("a", c_void_p),
                ("c", c_void_p), ("mean", c_void_p), ("invstd", c_void_p), ("stat_slots", c_void_p),
                ("dW", c_void_p), ("db", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("grad_replicas", c_int), ("grad_replica_stride", c_int), ("mma_bf16", c_int),
                ("num_batches_tracked", c_void_p)]


class SA(Structure):  # sn2_sa
    _fields_ = [("B", c_int), ("Nsrc", c_int), ("M", c_int), ("cap", c_int), ("cf", c_int), ("nl", c_int),
                ("feat", c_void_p), ("feat_stride", c_int), ("spos", c_void_p), ("spos_stride", c_int),
                ("cpos", c_void_p), ("nbr", c_void_p), ("cnt", c_void_p), ("total", c_void_p), ("order", c_void_p),
                ("blk", Block * 2),
                ("ext", c_void_p), ("arg", c_void_p), ("out", c_void_p), ("dout", c_void_p), ("dfeat", c_void_p)]


class FP(Structure):  # sn2_fp
    _fields_ = [("B", c_int), ("R_per_plot", c_int), ("S_per_plot", c_int), ("ca", c_int), ("cb", c_int),
                ("src", c_void_p), ("src_stride", c_int), ("src_a", c_void_p), ("src_c", c_void_p),
                ("knn_idx", c_void_p), ("knn_w", c_void_p), ("skip", c_void_p), ("skip_stride", c_int),
                ("blk", Block), ("h", c_void_p), ("h_stride", c_int), ("dy", c_void_p), ("dsrc", c_void_p),
                ("dsrc_stride", c_int), ("dskip", c_void_p), ("dskip_stride", c_int), ("du_scratch", c_void_p),
                ("scatter_ws", c_void_p), ("scatter_ready", c_int), ("bn_sums_done", c_void_p), ("src_ws", c_void_p),
                ("act_bf16", c_int), ("row_perm", c_void_p)]


class Head(Structure):  # sn2_head
    _fields_ = [("R", c_int), ("cin", c_int), ("f_stride", c_int), ("f", c_void_p), ("fa", c_void_p),
                ("fc", c_void_p), ("W1", c_void_p), ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p),
                ("coverages", c_void_p), ("proba", c_void_p), ("dcoverages", c_void_p), ("dproba", c_void_p),
                ("dy", c_void_p), ("dW1", c_void_p), ("db1", c_void_p), ("dW2", c_void_p), ("db2", c_void_p),
                ("grad_replicas", c_int), ("grad_replica_stride", c_int), ("drop_mask", c_void_p), ("drop_scale", c_float),
                ("act_bf16", c_int)]


# name -> argtypes; every entry point returns int (0 ok, >0 hipError_t, <0 argument error)
SIGNATURES = {
    "sn2_version": [],
    "sn2_debug_mfma_chain": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "sn2_debug_spin": [c_int, c_longlong, c_void_p, c_void_p],
    "sn2_pack_rows": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_fps": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_fps_waves": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "sn2_fps_status": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "sn2_debug_fps_spin_limit": [ctypes.c_uint],
    "sn2_ball_query": [c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p],
    "sn2_three_nn": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_count_sum": [c_void_p, c_int, c_void_p, c_void_p],
    "sn2_three_nn_xy": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_prepare_plots": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p],
    "sn2_znorm": [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                  c_void_p, c_void_p],
    "sn2_sa_order": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sn2_sa_forward": [POINTER(SA), c_int, c_void_p],
    "sn2_sa_backward": [POINTER(SA), c_void_p],
    "sn2_grad_reduce": [c_void_p, c_int, c_int, c_int, c_void_p],
    "sn2_interp_index": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_interp_index_perm": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_fp_forward": [POINTER(FP), c_int, c_void_p],
    "sn2_fp_backward": [POINTER(FP), c_void_p],
    "sn2_plot_max_forward": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sn2_plot_max_backward": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_global_level_forward": [POINTER(FP), POINTER(FP), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_debug_global_spin_limit": [ctypes.c_uint],
    "sn2_head_forward": [POINTER(Head), c_void_p],
    "sn2_fp_head_eval": [POINTER(FP), POINTER(Head), c_void_p],
    "sn2_fp_bn_sums": [POINTER(FP), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_head_bn_sums": [POINTER(Head), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_head_backward": [POINTER(Head), c_void_p],
    "sn2_plot_project_forward": [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p],
    "sn2_plot_pixels": [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sn2_plot_project_forward_pix": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_plot_project_backward": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_raster_project": [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                           c_void_p],
    "sn2_mosaic_merge": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                         c_int, c_int, c_int, c_int, c_void_p],
    "sn2_mosaic_finalize": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_kde_lookup": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "sn2_loss_forward": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p, c_void_p,
                         c_void_p],
    "sn2_loss_backward": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p, c_void_p,
                          c_void_p, c_void_p],
    "sn2_adam_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float,
                      c_void_p, c_float, c_void_p],
}

_lib = None


class StrataHipError(RuntimeError):
    pass


def load():
    """Load libstrata_hip.so (built in-tree by `_build.build()`); raises if it is absent -- there is no CPU
    fallback in the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StrataHipError(
            f"{LIB_PATH} not found: build it with `python -m stratanet2_vegetation_coverage_maps_amd._build` "
            "(or __graft_entry__.build()); the HIP library is mandatory, there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    if lib.sn2_version() != SN2_VERSION:
        raise StrataHipError(f"libstrata_hip.so version {lib.sn2_version()} != binding {SN2_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = "hipError_t" if rc > 0 else {-1: "SN2_EINVAL", -2: "SN2_ELIMIT"}.get(rc, "error")
        raise StrataHipError(f"{what} failed: {kind} {rc}")
