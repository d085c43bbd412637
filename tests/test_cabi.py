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
