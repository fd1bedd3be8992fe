#!/usr/bin/env python3
"""What the HBM of this box sustains for plain streaming (developer probe): a device-to-device
copy (read + write), a read-only reduction and a write-only fill, with torch kernels."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 32  # bytes
a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
b = torch.empty_like(a)


def t(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


ms = t(lambda: b.copy_(a))
print(f"copy   {ms:7.3f} ms  {2 * n / ms / 1e6:7.0f} GB/s (read + write)")
v = a.view(torch.int64)
ms = t(lambda: v.sum())
print(f"read   {ms:7.3f} ms  {n / ms / 1e6:7.0f} GB/s")
ms = t(lambda: b.fill_(7))
print(f"write  {ms:7.3f} ms  {n / ms / 1e6:7.0f} GB/s")
