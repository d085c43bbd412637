"""The position-only kernels of one parcel launch (256 plots x 10 000 points, reference ratios) alone on the chip, eager, one
stream: for a kernel trace (scripts/ktrace_cmd.sh) that shows what each costs without the other passes beside it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratanet2_vegetation_coverage_maps_amd import PointNet2  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch  # noqa: E402

B, N = int(os.environ.get("PLOTS", "256")), int(os.environ.get("POINTS", "10000"))
args = make_args(cuda=0, subsample_size=N)
model = PointNet2(args).eval()
d = make_batch(B, N)
xyz = d["xyz"].cuda()
fs = torch.zeros(2, B, dtype=torch.int32, device="cuda")
geo = model.alloc_geometry(B, N, xyz.device)
for _ in range(5):
    model._geometry(xyz, fs, out=geo, fork=False)
torch.cuda.synchronize()
