"""Time of one pinned host -> device copy of the cfg-2 batch size on the GPU box (HIP events)."""
import torch
for nbytes in (64 << 10, 256 << 10, 1300000, 4 << 20, 64 << 20):
    pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dev.copy_(pin, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print("%9d bytes: %7.1f us  %6.1f GB/s" % (nbytes, t * 1e6, nbytes / t / 1e9))
