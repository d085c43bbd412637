#!/usr/bin/env python3
"""Which block's bf16 path disagrees with the oracle's bf16 emulation?  One block at a time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_args, golden_state_dict, load_golden
from oracle import check
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages

g, args = load_golden("b4_well_conditioned"), golden_args("b4_well_conditioned")
d = {"cloud": torch.from_numpy(g["in/cloud"]), "xyz": torch.from_numpy(g["in/xyz"]),
     "coverages": torch.from_numpy(g["in/coverages"]), "pdf_all": torch.from_numpy(g["in/pdf_all"])}
fs = torch.from_numpy(g["in/fps_start"]); d["fps_start"] = fs
sd = golden_state_dict(g)
for blocks in (("sa3_module.nn",), ("fp3_module.nn",), ("fp2_module.nn",), PointNet2.BF16_BLOCKS):
    args.cuda, args.mma_dtype = 0, "bf16"
    m = PointNet2(args); m.load_state_dict(sd); m.train(); m.BF16_BLOCKS = blocks
    cov, proba = m(d)
    saved = cov.grad_fn.saved
    pred = project_to_plotwise_coverages(cov, d["cloud"], args, model=m)
    loss, _ = losses.total_loss(pred, proba, d["coverages"].cuda(), d["pdf_all"].cuda(), args.m, args.e)
    loss.backward()
    ref = check.train_step(sd, d, args, fps_start=fs, bf16_layers=blocks)
    fails, report = check.compare(m, cov, proba, loss.item(), ref, tol_out=1e-3, tol_grad=2e-2, pred=pred)
    print(blocks, "->", len(fails), "violations;", report.split("\n")[0].strip(), flush=True)
    for f in fails[:6]: print("    ", f)
