#!/usr/bin/env python3
"""Host-side cost of enqueueing one bench step (python + ctypes + HIP launch), which bounds the N > 1 rate when it
exceeds the kernel time: single-launch schedule vs the three-launch overlap schedule (no neighbours on one GPU, so
without the RCCL call itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
W, H = 6000, 4000
band = tiles.make_band_buffer(syn.bayer_uniform(H, W, seed=1), device="cuda")
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for force, sched in ((False, "serial"), (True, "serial"), (True, "side"), (True, "pipelined"), (False, "serial")):
    dev = tiles.BandDeveloper(gpu, W, H, 0, H, R.RawParams(**syn.RAW), lambda ed: syn.configure(ed, syn.PARAM_SETS["D"]),
                              force_split=force, schedule=sched)
    for _ in range(20):
        dev.step(band, out)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        dev.step(band, out)
    t1 = time.perf_counter()
    dev.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{'three launches, schedule ' + sched if force else 'single launch'}: host enqueue {1e6 * (t1 - t0) / n:.1f} us/step, "
          f"GPU-complete {1e6 * (t2 - t0) / n:.1f} us/step")
