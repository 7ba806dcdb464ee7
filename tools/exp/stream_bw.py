"""Measured stream ceilings of the part (torch's own fill / reduce / copy kernels on buffers far past the 256 MB
Infinity Cache): what a write-only, a read-only and a copy stream reach.  Context for the roofline fractions of the
write-heavy kernels (node side, first-layer node stream, pool zero)."""
import torch
dev = "cuda"
n = 1 << 30                      # 4 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

t = timed(lambda: a.zero_());            print("write-only  fill 4 GiB   : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: a.sum());              print("read-only   sum  4 GiB   : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: b.copy_(a));           print("copy        4+4 GiB      : %.2f TB/s (read + write bytes)" % (8 * n / t / 1e12))
h = torch.empty(n, dtype=torch.bfloat16, device=dev)
t = timed(lambda: h.copy_(a));           print("convert     4 GiB -> 2 GiB bf16 : %.2f TB/s (read + write bytes)" % (6 * n / t / 1e12))
for mb in (64, 128, 205, 512):
    m = mb * (1 << 20) // 4
    c = a[:m]
    t = timed(lambda: c.zero_(), reps=20); print("write-only  fill %4d MB  : %.2f TB/s" % (mb, 4 * m / t / 1e12))
