#!/usr/bin/env python3
"""Which block's bf16 path disagrees with the oracle's bf16 emulation?  One block at a time (3sa-arch)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import check, network
from stratanet2_vegetation_coverage_maps_amd import losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd.point_net2_3sa import PointNet2ThreeSA
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch

B, N = (int(v) for v in (sys.argv[1:3] + ["2", "4096"][len(sys.argv) - 1:]))
args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0, ratio3=0.25, r3=4.0, mma_dtype="bf16")
d = make_batch(B, N, first_plot=500)
fs = torch.zeros(3, B, dtype=torch.long); d["fps_start"] = fs
sd = network.init_state_dict_3sa(2)
for blocks in [()] + [(b,) for b in PointNet2ThreeSA.BF16_BLOCKS] + [PointNet2ThreeSA.BF16_BLOCKS]:
    m = PointNet2ThreeSA(args); m.set_mma_dtype("bf16"); m.load_state_dict(sd); m.train(); m.BF16_BLOCKS = blocks
    cov, proba = m(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args, model=m)
    loss, _ = losses.total_loss(pred, proba, d["coverages"].cuda(), d["pdf_all"].cuda(), args.m, args.e)
    loss.backward()
    ref = check.train_step(sd, d, args, fps_start=fs, arch="3sa", bf16_layers=blocks)
    fails, report = check.compare(m, cov, proba, loss.item(), ref, tol_out=1e-3, tol_grad=2e-2, pred=pred)
    worst = max(float(l.split("grad err ")[1].split()[0]) for l in report.split("\n") if "grad err" in l)
    print(f"{str(blocks)[:70]:70s} -> {len(fails):2d} violations; {report.split(chr(10))[0].strip()[:60]}; worst grad err {worst:.2e}", flush=True)
