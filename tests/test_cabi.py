"""The C-ABI library builds in-tree, loads, and exports every symbol include/strata_hip.h declares (CPU only: no
compute call is made)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "strata_hip.h")).read()
    return sorted(set(re.findall(r"^int\s+(sn2_\w+)\s*\(", txt, flags=re.M)))


def test_header_declares_the_hot_path():
    names = _declared()
    for must in ("sn2_fps", "sn2_ball_query", "sn2_three_nn", "sn2_sa_forward", "sn2_sa_backward", "sn2_fp_forward",
                 "sn2_fp_backward", "sn2_head_forward", "sn2_head_backward", "sn2_plot_project_forward",
                 "sn2_plot_project_backward", "sn2_raster_project", "sn2_version"):
        assert must in names


def test_library_builds_loads_and_exports_everything():
    from stratanet2_vegetation_coverage_maps_amd import _build, _lib
    path = _build.build(verbose=False)
    assert os.path.exists(path)
    raw = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(raw, name), f"{name} declared in strata_hip.h but not exported"
    lib = _lib.load()
    assert lib.sn2_version() == _lib.SN2_VERSION
    assert set(_lib.SIGNATURES) == set(_declared()), "ctypes binding and header disagree"


def test_struct_layouts_match_the_header_abi():
    """sizeof of the ctypes mirrors == what a C compiler computes for the header's structs."""
    import subprocess
    import tempfile
    from stratanet2_vegetation_coverage_maps_amd import _lib
    src = ('#include <stdio.h>\n#include "strata_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(sn2_block),sizeof(sn2_sa),sizeof(sn2_fp),sizeof(sn2_head),sizeof(sn2_net_layer),sizeof(sn2_net_model),'
           'sizeof(sn2_net_dims),sizeof(sn2_net_geo),sizeof(sn2_net_act),sizeof(sn2_net_bwd),sizeof(sn2_net_io));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(c) for c in (_lib.Block, _lib.SA, _lib.FP, _lib.Head, _lib.NetLayer, _lib.NetModel, _lib.NetDims,
                                                _lib.NetGeo, _lib.NetAct, _lib.NetBwd, _lib.NetIO)]


def test_argument_checks_return_before_any_device_work():
    """The entry points validate their descriptors first and return SN2_EINVAL (-1) / SN2_ELIMIT (-2) without touching the
    device: checked here for the newest one, the fused global level (shapes other than the reference architecture's, bfloat16
    operands and NULL workspaces are refused -- the host then uses the separate calls)."""
    from stratanet2_vegetation_coverage_maps_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH if os.path.exists(_lib.LIB_PATH) else __import__(
        "stratanet2_vegetation_coverage_maps_amd._build", fromlist=["build"]).build(verbose=False))
    fn = raw.sn2_global_level_forward
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(_lib.FP), ctypes.POINTER(_lib.FP)] + [ctypes.c_void_p] * 5
    sa3, fp3 = _lib.FP(), _lib.FP()
    assert fn(None, None, None, None, None, None, None) == -1
    fake = 0x1000                                            # never dereferenced: every call below fails a check first
    sa3.B = fp3.B = 4
    sa3.R_per_plot = fp3.R_per_plot = sa3.S_per_plot = 256
    fp3.S_per_plot = 1
    sa3.ca, sa3.cb, fp3.ca, fp3.cb = 32, 3, 64, 32
    sa3.blk.cin, sa3.blk.cout, fp3.blk.cin, fp3.blk.cout = 35, 64, 96, 48          # FP3 with 48 outputs: not the architecture
    assert fn(ctypes.byref(sa3), ctypes.byref(fp3), fake, fake, fake, fake, None) == -2
    fp3.blk.cout = 64
    fp3.blk.mma_bf16 = 1                                                            # bfloat16 operands: the separate kernels
    assert fn(ctypes.byref(sa3), ctypes.byref(fp3), fake, fake, fake, fake, None) == -2
    fp3.blk.mma_bf16 = 0
    fp3.S_per_plot = 256                                                            # FP3 must interpolate the plot's ONE source
    assert fn(ctypes.byref(sa3), ctypes.byref(fp3), fake, fake, fake, fake, None) == -1
    fp3.S_per_plot = 1
    assert fn(ctypes.byref(sa3), ctypes.byref(fp3), fake, fake, fake, fake, None) == -1   # no rows, tables or weights given


def _net_model_shell(max_neighbors=2000, n_flat=14997, source_side=1, fuse_eval_head=1):
    """A sn2_net_model with the reference architecture's layer shapes and no pointers: what the layout and the argument checks read."""
    from stratanet2_vegetation_coverage_maps_amd import _lib
    m = _lib.NetModel()
    for L, (ci, co) in zip([m.sa1[0], m.sa1[1], m.sa2, m.sa3, m.fp3, m.fp2, m.fp1],
                           [(11, 16), (16, 16), (19, 32), (35, 64), (96, 64), (80, 34), (42, 34)]):
        L.cin, L.cout = ci, co
    m.n_flat, m.max_neighbors, m.source_side, m.fuse_eval_head = n_flat, max_neighbors, source_side, fuse_eval_head
    return m


def _dims(B, N, M1, M2, maxn=2000, act_bf16=0, p2=0):
    from stratanet2_vegetation_coverage_maps_amd import _lib
    d = _lib.NetDims()
    d.B, d.N, d.M1, d.M2, d.cap1, d.cap2, d.act_bf16, d.p2_diam_pix = B, N, M1, M2, min(maxn, N), min(maxn, M1), act_bf16, p2
    return d


def test_network_executor_layouts_match_the_host_shape_tables():
    """sn2_net_geo_carve / sn2_net_act_carve (host-only arithmetic, no device): every buffer 256-byte aligned, no two overlap, each
    at least as large as the shape the host views it with (executor._geo_shapes / _act_shapes), everything inside the arena --
    for the metric's shape, the reference defaults, config 5's plots in bfloat16 and a tiny batch."""
    from ctypes import byref
    from stratanet2_vegetation_coverage_maps_amd import _lib
    from stratanet2_vegetation_coverage_maps_amd import executor as X
    lib = _lib.load()
    base = X._FAKE_BASE
    for (B, N, M1, M2, bf, p2) in [(16, 32768, 1024, 256, 0, 20), (20, 10000, 2500, 625, 0, 0), (8, 131072, 1024, 256, 1, 0),
                                   (2, 64, 16, 4, 0, 0), (512, 10000, 2500, 625, 0, 0)]:
        m, d = _net_model_shell(), _dims(B, N, M1, M2, act_bf16=bf, p2=p2)
        sz = ctypes.c_size_t()
        g = _lib.NetGeo()
        assert lib.sn2_net_geo_carve(byref(m), byref(d), base, byref(g), byref(sz)) == 0
        offs = X._offsets(g, X._geo_shapes(B, N, M1, M2, d.cap1, d.cap2))
        assert ("p2_pix" in offs) == (p2 > 0) and ("ws1" in offs) == (N > 2048 and not (N <= 4096 and B > 32))
        spans = sorted((o, o + X._ITEM[dt] * int(__import__("math").prod(sh))) for o, dt, sh in offs.values())
        assert all(o % 256 == 0 for o, _ in spans) and spans[-1][1] <= sz.value
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        for training in (0, 1):
            a = _lib.NetAct()
            assert lib.sn2_net_act_carve(byref(m), byref(d), training, base, byref(a), byref(sz)) == 0
            offs = X._offsets(a, X._act_shapes(B, N, M1, M2, bf))
            assert ("h1" in offs) == bool(training or bf)               # the fused eval pass keeps no per-point rows
            spans = sorted((o, o + X._ITEM[dt] * int(__import__("math").prod(sh))) for o, dt, sh in offs.values())
            assert all(o % 256 == 0 for o, _ in spans) and spans[-1][1] <= sz.value
            assert all(x[1] <= y[0] for x, y in zip(spans, spans[1:]))
        bw = _lib.NetBwd()
        sz2 = ctypes.c_size_t()
        assert lib.sn2_net_bwd_carve(byref(m), byref(d), base, base, byref(bw), byref(sz), byref(sz2)) == 0
        stride = (14997 + 63) // 64 * 64
        assert bw.images == 32 and bw.image_stride == stride and bw.arena_words * 4 == sz.value
        assert bw.dy2 - base == 32 * stride * 4 and bw.dy_sa3 - base + B * M2 * 64 * 4 <= sz.value


def test_network_executor_argument_checks_return_before_any_device_work():
    from ctypes import byref
    from stratanet2_vegetation_coverage_maps_amd import _lib
    lib = _lib.load()
    m, d = _net_model_shell(), _dims(4, 4096, 512, 128)
    g, a, b, io = _lib.NetGeo(), _lib.NetAct(), _lib.NetBwd(), _lib.NetIO()
    sz = ctypes.c_size_t()
    assert lib.sn2_net_geo_carve(None, byref(d), None, byref(g), byref(sz)) == -1
    assert lib.sn2_net_geometry(byref(m), byref(d), byref(g), byref(io), None) == -1          # no tables
    assert lib.sn2_net_forward(byref(m), byref(d), byref(g), byref(a), byref(io), None) == -1  # no parameters
    assert lib.sn2_net_backward(byref(m), byref(d), byref(g), byref(a), byref(b), None) == -1
    d.cap1 = 100                                                                                # not min(max_neighbors, N)
    assert lib.sn2_net_geo_carve(byref(m), byref(d), None, byref(g), byref(sz)) == -1
    d = _dims(4, 4096, 512, 128)
    m.fp2.cout = 48                                                                             # not the reference architecture
    assert lib.sn2_net_geo_carve(byref(m), byref(d), None, byref(g), byref(sz)) == -2
    assert lib.sn2_net_ctx_destroy(None) == -1
