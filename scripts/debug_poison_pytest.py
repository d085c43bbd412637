"""pytest with every CUDA torch.empty() poisoned (NaN floats, 1 ints): python scripts/debug_poison_pytest.py <pytest args>"""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
_empty = torch.empty
def poisoned_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel():
        if t.dtype in (torch.float32, torch.float64):
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(1)
    return t
torch.empty = poisoned_empty
import pytest                                        # noqa: E402
sys.exit(pytest.main(sys.argv[1:]))
