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
                ("num_batches_tracked", c_void_p), ("frozen_stats", c_int)]


BN_FROZEN_KEEP = 2          # SN2_BN_FROZEN_KEEP: the `training` argument of an eval-mode forward whose backward will be asked for


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
                ("act_bf16", c_int), ("zero_fill", c_void_p), ("zero_fill_words", c_long)]


class NetLayer(Structure):  # sn2_net_layer
    _fields_ = [("cin", c_int), ("cout", c_int), ("W", c_void_p), ("b", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p),
                ("gW", c_int), ("gb", c_int), ("ggamma", c_int), ("gbeta", c_int), ("mma_bf16", c_int)]


class NetModel(Structure):  # sn2_net_model
    _fields_ = [("sa1", NetLayer * 2), ("sa2", NetLayer), ("sa3", NetLayer), ("fp3", NetLayer), ("fp2", NetLayer), ("fp1", NetLayer),
                ("lin1_W", c_void_p), ("lin1_b", c_void_p), ("lin2_W", c_void_p), ("lin2_b", c_void_p),
                ("g_lin1_W", c_int), ("g_lin1_b", c_int), ("g_lin2_W", c_int), ("g_lin2_b", c_int), ("n_flat", c_int),
                ("r1_sq", c_float), ("r2_sq", c_float), ("max_neighbors", c_int), ("drop_p", c_float),
                ("fuse_global_level", c_int), ("fuse_eval_head", c_int), ("source_side", c_int),
                ("fps_waves_shared", c_int), ("fps_waves_many", c_int)]


class NetDims(Structure):  # sn2_net_dims
    _fields_ = [("B", c_int), ("N", c_int), ("M1", c_int), ("M2", c_int), ("cap1", c_int), ("cap2", c_int), ("act_bf16", c_int),
                ("p2_diam_pix", c_int)]


class NetGeo(Structure):  # sn2_net_geo
    _fields_ = [(n, c_void_p) for n in (
        "xyz", "idx1", "pos1_soa", "pos1_aos", "ws1", "nbr1", "cnt1", "tot1", "ord1",
        "idx2", "pos2_soa", "pos2_aos", "ws2", "nbr2", "cnt2", "tot2", "ord2", "pos3",
        "knn3_idx", "knn3_w", "knn2_idx", "knn2_w", "knn1_idx", "knn1_w", "inv3", "inv2", "inv1", "nn_ws2", "nn_ws1",
        "rank1", "rows0", "p2_pix", "p2_mm")]


class NetAct(Structure):  # sn2_net_act
    _fields_ = [(n, c_void_p) for n in (
        "aux", "stats", "ext1", "arg1", "x1", "ext2", "arg2", "x2", "h_sa3", "h3", "x3", "arg3", "h2", "h1", "src_ws1", "src_ws2",
        "cov", "proba", "drop_mask", "bwd_arena")] + [("bwd_arena_words", c_long)]


class NetBwd(Structure):  # sn2_net_bwd
    _fields_ = [("dcov", c_void_p), ("dproba", c_void_p), ("arena", c_void_p), ("arena_words", c_long), ("images", c_int),
                ("image_stride", c_int)] + [(n, c_void_p) for n in (
                    "dy2", "dy3", "dx1", "dx2", "dx3", "dy_sa3", "dy1", "du1", "du2", "du3", "bn_ok", "src_ws1", "src_ws2")] + [
                        ("defer_grad_reduce", c_int), ("arena_is_zero", c_int), ("frozen_stats", c_int)]


class NetIO(Structure):  # sn2_net_io
    _fields_ = [("cloud", c_void_p), ("fps_start", c_void_p), ("fps_status", c_void_p), ("gl_xchg", c_void_p), ("gl_ctl", c_void_p),
                ("stream_b", c_void_p), ("stream_c", c_void_p), ("stream_pack", c_void_p), ("ctx", c_void_p), ("flags", c_int),
                ("training", c_int)]


NET_FORK, NET_SHARED, NET_INVERTED, NET_DEFER_JOIN, NET_INPUT_ONLY = 1, 2, 4, 8, 16          # SN2_NET_* of the header
NET_HAS_ROWS0, NET_JOIN_PENDING, NET_WITH_GEOMETRY, NET_HAS_INVERTED = 32, 64, 128, 256


# name -> argtypes; every entry point returns int (0 ok, >0 hipError_t, <0 argument error)
SIGNATURES = {
    "sn2_version": [],
    "sn2_debug_mfma_chain": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "sn2_debug_spin": [c_int, c_longlong, c_void_p, c_void_p],
    "sn2_debug_fp1_backward_parts": [c_int],
    "sn2_debug_fp_rows_form": [c_int],
    "sn2_debug_fp_table_form": [c_int],
    "sn2_debug_stream_probe": [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p, c_void_p],
    "sn2_debug_mfma_probe": [c_int, c_int, c_void_p, POINTER(c_double), c_void_p],
    "sn2_pack_rows": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_fps": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_fps_waves": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "sn2_fps_status": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "sn2_debug_fps_spin_limit": [ctypes.c_uint],
    "sn2_ball_query": [c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p],
    "sn2_three_nn": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_count_sum": [c_void_p, c_int, c_void_p, c_void_p],
    "sn2_count_sum_group": [c_void_p, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p],
    "sn2_three_nn_xy": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_prepare_plots": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p],
    "sn2_znorm": [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                  c_void_p, c_void_p],
    "sn2_sa_order": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "sn2_sa_order_group": [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p],
    "sn2_sa_forward": [POINTER(SA), c_int, c_void_p],
    "sn2_sa_backward": [POINTER(SA), c_void_p],
    "sn2_grad_reduce": [c_void_p, c_int, c_int, c_int, c_void_p],
    "sn2_interp_index": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_interp_index_perm": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_interp_index_group": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p],
    "sn2_fp_forward": [POINTER(FP), c_int, c_void_p],
    "sn2_fp_backward": [POINTER(FP), c_void_p],
    "sn2_plot_max_forward": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sn2_plot_max_backward": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sn2_global_pool_backward": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p],
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
    "sn2_projected_loss_forward": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_projected_loss_backward": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sn2_adam_step_images": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float,
                             c_void_p, c_float, c_void_p],
    "sn2_net_ctx_create": [POINTER(c_void_p)],
    "sn2_net_ctx_destroy": [c_void_p],
    "sn2_net_geo_carve": [POINTER(NetModel), POINTER(NetDims), c_void_p, POINTER(NetGeo), POINTER(ctypes.c_size_t)],
    "sn2_net_act_carve": [POINTER(NetModel), POINTER(NetDims), c_int, c_void_p, POINTER(NetAct), POINTER(ctypes.c_size_t)],
    "sn2_net_bwd_carve": [POINTER(NetModel), POINTER(NetDims), c_void_p, c_void_p, POINTER(NetBwd), POINTER(ctypes.c_size_t),
                          POINTER(ctypes.c_size_t)],
    "sn2_net_geometry": [POINTER(NetModel), POINTER(NetDims), POINTER(NetGeo), POINTER(NetIO), c_void_p],
    "sn2_net_forward": [POINTER(NetModel), POINTER(NetDims), POINTER(NetGeo), POINTER(NetAct), POINTER(NetIO), c_void_p],
    "sn2_net_backward": [POINTER(NetModel), POINTER(NetDims), POINTER(NetGeo), POINTER(NetAct), POINTER(NetBwd), c_void_p],
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
