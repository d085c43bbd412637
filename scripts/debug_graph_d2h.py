#!/usr/bin/env python3
"""A large device->host copy issued AFTER a hipGraph was captured makes later replays of the pipeline's feature graphs
return NaN (found with scripts/debug_feeder_nan.py: `bench.py --host-inputs` built its pinned sources with `.cpu()` after
`pipe.capture()`).  Which kernel's result changes, and does plain torch show it too?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def plain_torch():
    x = torch.randn(1 << 22, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            y = (x * 2 + 1).relu()
            z = y.sum()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf = torch.zeros(1 << 20, device=dev)          # a memset node
        y = (x * 2 + 1).relu()
        z = y.sum() + buf.sum()
    g.replay()
    torch.cuda.synchronize()
    ref = z.clone()
    big = torch.randn(1 << 23, device=dev)
    h = big.cpu()                                        # 32 MiB to pageable host memory
    g.replay()
    torch.cuda.synchronize()
    print(f"plain torch graph: before {ref.item():.6f} after a 32 MiB .cpu(): {z.item():.6f} same={bool(ref == z)}", flush=True)
    return h


def feature_graph(N=8192, B=4):
    from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
    from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, flatten_parameters
    from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
    args = make_args(cuda=0, subsample_size=N, ratio1=1024 / N, r1=1.0, ratio2=0.25, r2=2.0)
    torch.manual_seed(0)
    model = PointNet2(args).train()
    flatten_parameters(model)
    opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3)
    h = make_batch(B, N)
    inp = {"cloud": h["cloud"].to(dev), "xyz": h["xyz"].to(dev), "fps_start": torch.zeros(2, B, dtype=torch.int32, device=dev),
           "gt": h["coverages"].to(dev), "pdf": h["pdf_all"].to(dev)}
    geo = model.alloc_geometry(B, N, dev)
    model._geometry(inp["xyz"], inp["fps_start"], out=geo)
    keep = {}

    def feature_step():
        opt.zero_grad(set_to_none=True)
        cov, proba = model({"cloud": inp["cloud"], "xyz": inp["xyz"], "fps_start": inp["fps_start"], "geometry": geo})
        pred = project_to_plotwise_coverages(cov, inp["cloud"], args)
        loss, _ = losses.total_loss(pred, proba, inp["gt"], inp["pdf"], args.m, args.e)
        saved = cov.grad_fn.saved
        loss.backward()
        keep.update(cov=cov, proba=proba, pred=pred, loss=loss, flat_grad=model._last_flat_grad)
        for k, v in saved.__dict__.items():
            if isinstance(v, torch.Tensor):
                keep["s." + k] = v
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        feature_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        feature_step()
    g.replay()
    torch.cuda.synchronize()
    snap = {k: v.clone() for k, v in keep.items()}
    g.replay()
    torch.cuda.synchronize()
    same = [k for k, v in keep.items() if not torch.equal(torch.nan_to_num(v.float()), torch.nan_to_num(snap[k].float()))]
    print(f"feature graph replayed twice: tensors that differ: {same}", flush=True)
    for what, fn in (("a 512 B .cpu()", lambda: inp["gt"].cpu()), ("a clone on the device", lambda: inp["cloud"].clone()),
                     ("a 1.3 MiB .cpu()", lambda: inp["cloud"].cpu()), ("pin_memory of a host tensor", lambda: torch.zeros(1 << 20).pin_memory())):
        r = fn()
        g.replay()
        torch.cuda.synchronize()
        bad = [k for k, v in keep.items() if not torch.equal(torch.nan_to_num(v.float()), torch.nan_to_num(snap[k].float()))]
        nan = [k for k, v in keep.items() if v.is_floating_point() and not bool(torch.isfinite(v).all()) and bool(torch.isfinite(snap[k]).all())]
        print(f"after {what}: differ: {bad[:12]}{' ...' if len(bad) > 12 else ''}; newly non-finite: {nan[:12]}; loss {keep['loss'].item():.6f}",
              flush=True)


def memset_node(nbytes=153600):
    """hipMemsetAsync captured as a memset NODE between kernel nodes: is it ordered on replay?  buf := 0 ; buf += 1 ;
    out = copy(buf): every element of `out` must be 1 after every replay."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    n = nbytes // 8
    buf = torch.full((n,), 7, dtype=torch.int64, device=dev)
    big = torch.randn(1 << 19, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        buf.add_(1)
        out = buf.clone()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w = big * 2.0                                   # a kernel node in front
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        buf.add_(1)
        out = buf.clone()
    assert rc == 0
    res = []
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        res.append((int((out == 1).sum()), int((out == 0).sum()), int(out.max())))
    h = big.cpu()
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        res.append((int((out == 1).sum()), int((out == 0).sum()), int(out.max())))
    print(f"memset node {nbytes} B, {n} elements: (ones, zeros, max) per replay, a 2 MiB .cpu() after the third: {res}", flush=True)


if __name__ == "__main__":
    plain_torch()
    for nb in (512, 38400, 153600, 4 << 20):
        memset_node(nb)
    feature_graph()
