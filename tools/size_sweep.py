"""Kernel time per megapixel of the RGBA pass (k_apply) and of the fused kernel at several frame sizes: a frame whose
input + output fit the 256-MiB Infinity Cache against frames that stream from HBM (is a pass bound by HBM latency?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
praw = R.RawParams(**syn.RAW)
stream = torch.cuda.current_stream().cuda_stream
for (h, w) in ((1000, 1500), (2000, 3000), (4000, 6000), (8000, 12000)):
    cfa = syn.bayer_uniform(h, w)
    band = tiles.make_band_buffer(cfa, device="cuda")
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    lin = torch.empty_like(out)
    ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
    syn.configure(ed, syn.PARAM_SETS[sys.argv[1] if len(sys.argv) > 1 else "D"])
    R.dev_demosaic(gpu, band.data_ptr(), praw, w, h, 0, h, lin.data_ptr(), stream)
    res = {}
    for name, fn in (("apply", lambda: ed.dev_apply(lin.data_ptr(), out.data_ptr(), R.OUT_RGBA_F32, stream)),
                     ("develop", lambda: ed.dev_develop(band.data_ptr(), praw, out.data_ptr(), R.OUT_RGBA_F32, stream))):
        n = max(20, int(2000 * 1.5e6 / (h * w)))
        for _ in range(n // 2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n
    mp = h * w / 1e6
    print(f"{w}x{h}: apply {res['apply']*1e3:8.1f} us = {res['apply']*1e3/mp:6.2f} us/MP   develop {res['develop']*1e3:8.1f} us = {res['develop']*1e3/mp:6.2f} us/MP", flush=True)
