"""Does hipExtStreamCreateWithCUMask work here, and how do mask bits map to CUs?  Times a bandwidth-bound and a compute-bound
torch kernel on streams restricted to the first n mask bits."""
import ctypes
import time
import torch

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits_set, total_words=8):
    words = (ctypes.c_uint32 * total_words)()
    for i in bits_set:
        words[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(total_words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

x = torch.randn(64 << 20, device="cuda")
y = torch.empty_like(x)
a = torch.randn(4096, 4096, device="cuda")
def bench(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3):
            torch.mul(x, 2.0, out=y)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            torch.mul(x, 2.0, out=y)
        stream.synchronize()
        t1 = (time.perf_counter() - t0) / 10
        b = a
        for _ in range(2):
            b = a @ a
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b = a @ a
        stream.synchronize()
        t2 = (time.perf_counter() - t0) / 5
    print(f"{label:28s} stream-copy {2 * x.numel() * 4 / t1 / 1e12:.2f} TB/s   fp32 gemm {2 * 4096 ** 3 / t2 / 1e12:.1f} TFLOP/s", flush=True)

bench(torch.cuda.Stream(), "no mask")
for n in (256, 128, 64, 32):
    bench(masked_stream(range(n)), f"first {n} bits")
bench(masked_stream(range(0, 256, 4)), "every 4th bit of 256")
bench(masked_stream(range(0, 256, 8)), "every 8th bit of 256")
