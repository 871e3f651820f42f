#!/usr/bin/env python3
"""Fixed cost of a fused launch: 6000-wide bands of 250 ... 4000 rows, back-to-back launches (us per launch) and the
kernel alone (HIP-event pair around single launches) — what a row band of a sharded frame costs beyond its share."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
if len(sys.argv) > 2:
    gpu.set_kernel_variant(int(sys.argv[2]))
praw = R.RawParams(**syn.RAW)
stream = torch.cuda.current_stream().cuda_stream
W, H = 6000, 4000
for rows in (16, 64, 128, 250, 500, 1000, 2000, 4000):
    y0 = (H - rows) // 2 & ~1
    band = tiles.make_band_buffer(syn.bayer_uniform(rows, W, seed=1), device="cuda")
    out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
    ed = R.PhotoEditor.for_tile(gpu, W, H, y0, rows)
    syn.configure(ed, syn.PARAM_SETS[sys.argv[1] if len(sys.argv) > 1 else "D"])
    fn = lambda: ed.dev_develop(band.data_ptr(), praw, out.data_ptr(), R.OUT_RGBA_F32, stream)
    for _ in range(300): fn()
    torch.cuda.synchronize()
    n = 400
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) / n * 1e3
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in pairs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ks = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
    print(f"{W} x {rows:4d}: back-to-back {b2b:7.1f} us/launch = {b2b / rows * 1e3:6.2f} ns/row   kernel alone median {ks[50]:7.1f} min {ks[0]:7.1f} us", flush=True)
