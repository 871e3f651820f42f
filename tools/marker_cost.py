#!/usr/bin/env python3
"""What the packets between two launches of ONE stream cost on the GPU: back-to-back launches of a 6000 x 500 band, the
same with an event recorded after every launch, with a wait for an event that another stream completed long ago, and
with both (the per-step packets of the exchange-ahead schedule)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
raw = R.RawParams(**syn.RAW)
conf = lambda ed: syn.configure(ed, syn.PARAM_SETS["D"])
W, H, y0, rows = 6000, 4000, 1500, 500
band = tiles.make_band_buffer(syn.bayer_uniform(rows, W, seed=1), device="cuda")
out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
dev = tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf)
side = torch.cuda.Stream()
old = torch.cuda.Event()
with torch.cuda.stream(side):
    old.record()
torch.cuda.synchronize()
evs = [torch.cuda.Event() for _ in range(8)]


def run(name, rec, wait, n=400):
    def steps(m):
        for k in range(m):
            if wait:
                torch.cuda.current_stream().wait_event(old)
            dev.step(band, out)
            if rec:
                evs[k & 7].record()
    steps(100)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        steps(n)
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / n)
    print(f"  {name:60s} {best:6.1f} us/step", flush=True)


run("back-to-back launches", False, False)
run("+ an event recorded after every launch", True, False)
run("+ a wait for a long-completed event before every launch", False, True)
run("+ both", True, True)
