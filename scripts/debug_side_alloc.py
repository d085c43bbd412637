"""Experiment: a device block allocated on a SIDE stream, filled there by an H2D copy, consumed on the main stream
(`record_stream`), freed at the next iteration -- does the caching allocator reuse it, or hipMalloc anew every step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
host = torch.empty(524288 * 3, dtype=torch.float64).pin_memory()
work = torch.empty(64 << 20, dtype=torch.float32, device=dev)
keep = None
for mode in ("record_stream", "storage_pool"):
    for it in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n0 = torch.cuda.memory_stats()["segment.all.allocated"]
        work.mul_(1.0001)                      # ~0.1 ms of main-stream work in front (the "forward")
        with torch.cuda.stream(side):
            out = torch.empty(host.shape, dtype=host.dtype, device=dev)
            out.copy_(host, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        main.wait_event(ev)
        out.record_stream(main)
        s = out.sum()                          # consumer on main
        keep = (out, s)                        # freed when the next iteration replaces it (as a saved-for-backward tensor would be)
        x = float(s)                           # the loop's .item()
        dt = (time.perf_counter() - t0) * 1e3
        print(mode, it, f"{dt:7.3f} ms  new segments {torch.cuda.memory_stats()['segment.all.allocated'] - n0}", flush=True)
    break
print("use_count api:", hasattr(out.untyped_storage(), "_use_count"), out.untyped_storage()._use_count() if hasattr(out.untyped_storage(), "_use_count") else None)
