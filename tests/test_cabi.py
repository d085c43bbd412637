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
    src = ('#include <stdio.h>\n#include "strata_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
           'sizeof(sn2_block),sizeof(sn2_sa),sizeof(sn2_fp),sizeof(sn2_head));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.Block), ctypes.sizeof(_lib.SA), ctypes.sizeof(_lib.FP), ctypes.sizeof(_lib.Head)]


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
