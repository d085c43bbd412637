"""The two row passes of the source-side forward (fp_fwd_rows2_kernel / fp_fwd_rows_kernel) against each other, and each against
itself at another launch size (the first PLOTS2 plots alone): where do they differ?   CA=64 CB=16: FP2's shape in the parcel loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stratanet2_vegetation_coverage_maps_amd import _lib, hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_batch
dev = torch.device("cuda:0")
B, N, M1 = int(os.environ.get("PLOTS", 3)), int(os.environ.get("POINTS", 24001)), int(os.environ.get("M1", 1024))
B2 = int(os.environ.get("PLOTS2", max(1, B // 2)))
CA, CB = int(os.environ.get("CA", 34)), int(os.environ.get("CB", 8))
CAs = (CA + 3) // 4 * 4
torch.manual_seed(B * N)
d = make_batch(B, N, first_plot=3)
xyz = d["xyz"].to(dev).float().contiguous()
_, pos1_soa, _ = ops.fps(xyz, M1, torch.zeros(B, dtype=torch.int32, device=dev))[:3]
knn = ops.three_nn(pos1_soa, xyz, 3)
src = torch.randn(B * M1, CAs, device=dev)
a2, c2 = torch.rand(CA, device=dev) + 0.5, torch.randn(CA, device=dev) * 0.1
skip = torch.randn(B * N, 16, device=dev)
lin, bn = torch.nn.Linear(CA + CB, 34).to(dev), torch.nn.BatchNorm1d(34).to(dev)
bn.eval()
out = {}


def run(form, nb, training):
    _lib.load().sn2_debug_fp_rows_form(form)
    blk = ops.BlockBuffers(lin, bn)
    h1 = torch.full((nb * N, 36), 7.0, device=dev)
    k = (knn[0][:nb * N].contiguous(), knn[1][:nb * N].contiguous())
    ops.fp_forward(ops.fp_desc(blk, nb, N, M1, CA, CB, src[:nb * M1].contiguous(), h1, src_affine=(a2, c2), knn=k,
                               skip=skip[:nb * N, 0:CB], force_src_ws=True), training)
    torch.cuda.synchronize()
    return h1


def diff(a, b, what):
    ne = (a != b)
    rows = ne.any(dim=1).nonzero().flatten()
    print(f"{what}: differing elements {int(ne.sum())} of {a.numel()}, rows {rows.numel()} {rows[:12].tolist()} ... {rows[-4:].tolist()}; "
          f"columns {ne.any(dim=0).nonzero().flatten().tolist()}; max abs diff {float((a - b).abs().max()):.3e}; nan {int(torch.isnan(a).sum())} {int(torch.isnan(b).sum())}")
    if rows.numel():
        print("   rows mod 14:", torch.bincount(rows % 14, minlength=14).tolist())


for training in (0, 1):
    print("training", training)
    f1, f1b, f0 = run(1, B, training), run(1, B, training), run(0, B, training)
    diff(f1, f1b, "  form 1 twice")
    diff(f1, f0, "  form 1 vs form 0")
    s1, s0 = run(1, B2, training), run(0, B2, training)
    diff(s1, f1[:B2 * N], f"  form 1: {B2} plots alone vs inside {B}")
    diff(s0, f0[:B2 * N], f"  form 0: {B2} plots alone vs inside {B}")
_lib.load().sn2_debug_fp_rows_form(1)
