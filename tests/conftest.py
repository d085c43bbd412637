import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box: start a forkserver NOW, before anything in this process initialises the GPU (device_count() does not).
    # tests/test_gpu_distributed.py forks its ranks from it, so every rank is a clean process -- never a fork or an exec of
    # a process that already holds the GPU.
    if torch.cuda.device_count() > 0:
        import multiprocessing
        from multiprocessing import forkserver
        multiprocessing.get_context("forkserver")
        forkserver.set_forkserver_preload([])
        forkserver.ensure_running()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_state_dict(g):
    return {k[3:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd/")}


def golden_args(name):
    from oracle.make_golden import CASES
    from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args
    c = CASES[name]
    return make_args(subsample_size=c["N"], ratio1=c["ratio1"], r1=c["r1"], ratio2=c["ratio2"], r2=c["r2"])


GOLDEN_CASES = ["c1_ref_defaults", "b2_ref_defaults", "b2_c2_style", "b4_well_conditioned"]
