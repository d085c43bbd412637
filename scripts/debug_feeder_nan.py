#!/usr/bin/env python3
"""Diagnostic for the NaN loss `bench.py --host-inputs` recorded in round 1 (profiles/r01_bench_host_inputs.json).

Runs the pipelined loop at the metric's size in several variants and prints, for each, the per-step losses (cloned on the
main stream after every step, read back once at the end: no host synchronisation inside the loop) and whether the
parameters stayed finite:

    resident          slots hold the data (the headline mode)
    feeder            every batch copied from pinned host memory on the side streams
    resident+delay    resident, but every geometry pass starts with a long spin kernel: the feature passes must WAIT
    feeder+sync       feeder with a device synchronisation after every step (no overlap at all)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.pipeline import TrainPipeline  # noqa: E402
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch  # noqa: E402


def run(variant, B=16, N=32768, depth=3, steps=80, pair=True, only=None, pin_first=False, blocking=False, pinned_mem=True,
        from_device=False):
    dev = torch.device("cuda", 0)
    args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
    torch.manual_seed(0)
    model = PointNet2(args).train()
    flatten_parameters(model)
    opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
    n_slots = 2 * depth + 2 if pair else depth + 1
    slots = []
    for j in range(n_slots):
        h = make_batch(B, N, first_plot=j * B)
        slots.append({"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev),
                      "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
                      "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)})

    def feature_step(inp, geo=None):
        opt.zero_grad(set_to_none=True)
        cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"]}
        if geo is not None:
            cd["geometry"] = geo
        cov, proba = model(cd)
        pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
        loss, _ = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
        loss.backward()
        return loss

    names = only or ("cloud", "xyz", "gt", "pdf")

    def make_sources():
        if from_device:
            return [{k: v.clone() for k, v in sl.items() if k in names} for sl in slots]
        if pinned_mem:
            return [{k: v.cpu().pin_memory() for k, v in sl.items() if k in names} for sl in slots]
        return [{k: v.cpu() for k, v in sl.items() if k in names} for sl in slots]
    pinned = make_sources() if pin_first else None
    pipe = TrainPipeline(model, opt, feature_step, slots, depth=depth, use_graph=True)
    pipe.capture()
    if variant.startswith("feeder"):
        if pinned is None:
            pinned = make_sources()
        pipe.set_feeder(lambda i: pinned[i % len(pinned)])
        pipe.feeder_blocking = blocking
    if variant == "resident+delay":
        orig = pipe.issue_geometry

        def delayed(i=None):
            for st in pipe.side:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(4_000_000)       # ~2 ms at 2 GHz in front of whatever comes next on that stream
            return orig(i)
        pipe.issue_geometry = delayed
    pipe.prime()
    out = torch.zeros(steps, dtype=torch.float64, device=dev)
    fin = torch.zeros(steps, dtype=torch.int32, device=dev)
    for s in range(steps):
        loss = pipe.step()
        out[s] = loss.detach()
        fin[s] = torch.isfinite(model._flat_params).all().to(torch.int32)
        if variant.endswith("+sync"):
            torch.cuda.synchronize()
    pipe.drain()
    torch.cuda.synchronize()
    ls = out.cpu().tolist()
    fn = fin.cpu().tolist()
    bad = [i for i, v in enumerate(ls) if v != v]
    variant = f"{variant} only={only} pin_first={pin_first} blocking={blocking} pinned={pinned_mem} from_device={from_device} N={N}"
    print(f"{variant:16s} pair={pair} first NaN loss at step {bad[0] if bad else None}; params first non-finite at "
          f"{fn.index(0) if 0 in fn else None}; losses[:10] = {[round(v, 5) for v in ls[:10]]} last = {ls[-1]:.5f}", flush=True)
    return ls


def main():
    variants = sys.argv[1:] or ["resident", "feeder", "resident+delay", "feeder+sync"]
    ref = None
    for v in variants:
        ls = run(v)
        if ref is None:
            ref = ls
        else:
            d = max((abs(a - b) if a == a and b == b else float("inf")) for a, b in zip(ls, ref))
            print(f"   max |loss - resident| = {d:.3e}", flush=True)
    run("feeder", pair=False)


def probe(B=16, N=32768, depth=3):
    """After prime() with the feeder: are the slots bit-identical to their host sources, and which tensor of an eager
    feature pass is the first non-finite one?"""
    dev = torch.device("cuda", 0)
    args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
    torch.manual_seed(0)
    model = PointNet2(args).train()
    flatten_parameters(model)
    opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
    slots = []
    for j in range(2 * depth + 2):
        h = make_batch(B, N, first_plot=j * B)
        slots.append({"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev),
                      "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
                      "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)})
    pinned = [{k: v.cpu().pin_memory() for k, v in sl.items() if k in ("cloud", "xyz", "gt", "pdf")} for sl in slots]
    for k, p in enumerate(pinned):
        for name, t in p.items():
            print(f"slot {k} {name}: pinned={t.is_pinned()} dtype={t.dtype} shape={tuple(t.shape)} stride={t.stride()} "
                  f"equal_to_slot={bool((t == slots[k][name].cpu()).all())} finite={bool(torch.isfinite(t).all())}")
        break

    def feature_step(inp, geo=None):
        opt.zero_grad(set_to_none=True)
        cd = {"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"]}
        if geo is not None:
            cd["geometry"] = geo
        cov, proba = model(cd)
        pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
        loss, parts = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
        loss.backward()
        feature_step.last = dict(cov=cov, proba=proba, pred=pred, loss=loss, parts=torch.stack(parts))
        return loss

    pipe = TrainPipeline(model, opt, feature_step, slots, depth=depth, use_graph=False)
    pipe.capture()
    pipe.set_feeder(lambda i: pinned[i % len(pinned)])
    pipe.prime()
    torch.cuda.synchronize()
    for k in range(6):
        for name in ("cloud", "xyz", "gt", "pdf"):
            a, b = slots[k][name].cpu(), pinned[k][name]
            nbad = int((a != b).sum())
            if nbad or k == 0:
                print(f"after prime: slot {k} {name}: {nbad} of {a.numel()} elements differ from the host source; "
                      f"finite={bool(torch.isfinite(a).all())}")
    loss = pipe.step()
    torch.cuda.synchronize()
    for name, t in feature_step.last.items():
        print(f"eager step 0: {name}: finite={bool(torch.isfinite(t).all())} {t.flatten()[:4].tolist()}")


def matrix():
    kw = dict(steps=6)
    run("feeder", pin_first=True, **kw)
    for nm in ("gt", "pdf", "cloud", "xyz"):
        run("feeder", only=(nm,), **kw)
    run("feeder", blocking=True, **kw)
    run("feeder", pinned_mem=False, **kw)
    run("feeder", from_device=True, **kw)
    run("feeder", B=2, N=4096, **kw)
    run("feeder", B=16, N=8192, **kw)


if __name__ == "__main__":
    if "probe" in sys.argv:
        probe()
    elif "matrix" in sys.argv:
        matrix()
    else:
        main()
