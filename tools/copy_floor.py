#!/usr/bin/env python3
"""Memory-only floors for the RGBA pass on this box: a device-to-device copy of one 6000x4000 RGBA32F frame (384 MB read +
384 MB written, the traffic of rpf_dev_apply) with torch's copy kernel and with hipMemcpyDtoDAsync, and a fill (write only)."""
import torch
n = 6000 * 4000 * 4
a = torch.rand(n, device="cuda")
b = torch.empty_like(a)
def t(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a))
print(f"copy 384 MB -> 384 MB: {ms:.4f} ms = {2 * n * 4 / ms / 1e12:.2f} TB/s (read + write)")
ms = t(lambda: b.fill_(1.0))
print(f"fill 384 MB: {ms:.4f} ms = {n * 4 / ms / 1e12:.2f} TB/s (write)")
ms = t(lambda: torch.add(a, 1.0, out=b))
print(f"add  384 MB -> 384 MB: {ms:.4f} ms = {2 * n * 4 / ms / 1e12:.2f} TB/s (read + write)")
s = t(lambda: a.sum())
print(f"sum  384 MB: {s:.4f} ms = {n * 4 / s / 1e12:.2f} TB/s (read)")
