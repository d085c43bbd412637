"""How fast can the HOST fill pinned memory, and how fast does it leave for the device?  (the drop-in's upload ring)
Variants: torch's pin_memory() buffers (hipHostMalloc, default flags) against ordinary pageable tensors registered in place
with hipHostRegister; a pageable `.to(device)` for reference."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

n = 21 * 1024 * 1024 // 4
src = torch.rand(n)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - a) / reps * 1e3


mb = n * 4 / 1e6
print(f"threads {torch.get_num_threads()}, buffer {mb:.1f} MB")
pin = torch.empty(n).pin_memory()
print(f"cpu copy into torch pin_memory buffer : {t(lambda: pin.copy_(src)):8.3f} ms")
reg = torch.empty(n)
rc = torch.cuda.cudart().cudaHostRegister(reg.data_ptr(), n * 4, 0)
print("hostRegister rc", rc, "is_pinned", reg.is_pinned())
print(f"cpu copy into registered buffer       : {t(lambda: reg.copy_(src)):8.3f} ms")
plain = torch.empty(n)
print(f"cpu copy into pageable buffer         : {t(lambda: plain.copy_(src)):8.3f} ms")
out = torch.empty(n, device=dev)
print(f"H2D from torch pin_memory (async)     : {t(lambda: out.copy_(pin, non_blocking=True)):8.3f} ms")
print(f"H2D from registered (async)           : {t(lambda: out.copy_(reg, non_blocking=True)):8.3f} ms")
print(f"H2D from pageable (.copy_)            : {t(lambda: out.copy_(src)):8.3f} ms")
a = time.perf_counter()
for _ in range(3):
    tmp = torch.empty(n)
    torch.cuda.cudart().cudaHostRegister(tmp.data_ptr(), n * 4, 0)
    torch.cuda.cudart().cudaHostUnregister(tmp.data_ptr())
print(f"register + unregister of a fresh buffer: {(time.perf_counter() - a) / 3 * 1e3:8.3f} ms")
for th in (1, 4, 8, 16):
    torch.set_num_threads(th)
    print(f"  threads {th:2d}: cpu copy into registered {t(lambda: reg.copy_(src)):8.3f} ms")
