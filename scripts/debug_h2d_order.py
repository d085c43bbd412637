#!/usr/bin/env python3
"""Is a kernel that follows an async pinned-host -> device copy on the SAME stream ordered after it?  (Diagnostic for the
NaN of `bench.py --host-inputs`: it appears at step 0, with or without overlap, only at the metric's buffer sizes.)"""
import sys

import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
side = torch.cuda.Stream()
side2 = torch.cuda.Stream()


def check(nbytes, stream, dtype=torch.float32, pre_wait=False, reps=4):
    n = nbytes // torch.empty(0, dtype=dtype).element_size()
    bad_after_kernel = bad_after_sync = 0
    for r in range(reps):
        src = (torch.arange(n, dtype=torch.float64) % 1000 + r + 1).to(dtype).pin_memory()
        dst = torch.zeros(n, dtype=dtype, device=dev)
        torch.cuda.synchronize()
        if pre_wait:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            stream.wait_event(ev)
        with torch.cuda.stream(stream):
            dst.copy_(src, non_blocking=True)
            out = dst.clone()                      # a kernel on the same stream, right behind the copy
        torch.cuda.synchronize()
        bad_after_kernel += int((out.cpu() != src).sum())
        bad_after_sync += int((dst.cpu() != src).sum())
    return bad_after_kernel, bad_after_sync


print("is_pinned:", torch.zeros(10).pin_memory().is_pinned(), flush=True)
for nbytes in (512, 256 << 10, 4 << 20, 6 << 20, 12 << 20, 20 << 20, 64 << 20):
    for name, st in (("default", torch.cuda.current_stream()), ("side", side)):
        for pre in (False, True):
            k, s = check(nbytes, st, pre_wait=pre)
            print(f"{nbytes:>10d} B  stream={name:8s} pre_wait={pre!s:5s}  wrong after same-stream kernel: {k:>9d}   wrong after device sync: {s}",
                  flush=True)
k, s = check(12 << 20, side, dtype=torch.float64)
print(f"float64 12 MiB side: {k} {s}")

# several side streams copying at once, then kernels (the pipeline's prime())
srcs = [(torch.rand(5 << 20) + i).pin_memory() for i in range(6)]
dsts = [torch.zeros(5 << 20, device=dev) for _ in range(6)]
outs = [None] * 6
torch.cuda.synchronize()
for i in range(6):
    st = (side, side2, torch.cuda.current_stream())[i % 3]
    with torch.cuda.stream(st):
        dsts[i].copy_(srcs[i], non_blocking=True)
        outs[i] = dsts[i] * 1.0
torch.cuda.synchronize()
print("concurrent:", [int((outs[i].cpu() != srcs[i]).sum()) for i in range(6)], [int((dsts[i].cpu() != srcs[i]).sum()) for i in range(6)])
