#!/usr/bin/env python3
"""Device-side rate of the decoded-image ingest (rpf_dev_rgb_u8 / _u16_to_rgba_f32) on one 6000x4000 frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rawphotoforge_amd as R

gpu = R.GpuProcessor(0)
H, W = 4000, 6000
s = torch.cuda.current_stream().cuda_stream
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for bits, dt in ((8, torch.uint8), (16, torch.int16)):
    img = torch.randint(0, 127, (H, W, 3), dtype=dt, device="cuda")
    for lut in (None, torch.rand(1 << bits, dtype=torch.float32, device="cuda")):
        f = lambda: R.dev_rgb_int_to_rgba_f32(gpu, img.data_ptr(), bits, H * W, out.data_ptr(), lut.data_ptr() if lut is not None else 0, s)
        for _ in range(20):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        bpp = 3 * bits // 8 + 16
        print(f"{bits}-bit RGB -> RGBA32F, {'table' if lut is not None else 'v / MAX'}: {ms:.4f} ms per 24 MP = "
              f"{H * W / ms / 1e6:.0f} k MP/s, {bpp * H * W / ms / 1e9:.2f} TB/s of {bpp} B/px")
