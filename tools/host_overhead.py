#!/usr/bin/env python3
"""Host-side cost of enqueueing one bench step, and what the step then takes on the GPU, for the schedules of the
N > 1 path on ONE GPU: the library's own sharded step (rpf_shard_develop: one C call per step) without neighbours,
with the interior / edges split forced, and with a real RCCL exchange in which the rank is its own neighbour; next to
the torch.distributed schedule of round 2 (tiles.BandDeveloper, three launches from Python).  The band geometries are
the per-rank ones of BASELINE configs[2] at 8 and 1 GPUs and of configs[3] at 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
raw = R.RawParams(**syn.RAW)
params = sys.argv[1] if len(sys.argv) > 1 else "D"
conf = lambda ed: syn.configure(ed, syn.PARAM_SETS[params])


def run(name, dev, band, out, n=400, **kw):
    for _ in range(100):
        dev.step(band, out, **kw)
    dev.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.step(band, out, **kw)
    t1 = time.perf_counter()
    dev.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  {name:98s} host enqueue {1e6 * (t1 - t0) / n:6.1f} us/step, GPU-complete {1e6 * (t2 - t0) / n:6.1f} us/step", flush=True)


for W, H, y0, rows in ((6000, 4000, 1500, 500), (12000, 8000, 3000, 1000), (6000, 4000, 0, 4000)):
    print(f"band {W} x {rows} of a {W} x {H} frame, edit set {params}:")
    band = tiles.make_band_buffer(syn.bayer_uniform(rows, W, seed=1), device="cuda")
    out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
    if 0 < y0 and y0 + rows < H:   # two step pipelines that share nothing: shard (communicator + thread), stream, buffers each
        lanes = []
        for i in range(2):
            shl = R.Shard(gpu, None, 0, 1)
            shl.set_neighbours(0, 0)
            lanes.append((tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=shl), band.clone(), torch.empty_like(out),
                          torch.cuda.Stream(priority=-i), shl))   # different priorities: see include/rpf.h

        class PingPong:
            k = 0

            def step(self, band, out):
                d, b, o, st, _ = lanes[self.k & 1]
                self.k += 1
                d.step_async(b, o, stream=st.cuda_stream)

            def join(self):
                for ln in lanes:
                    ln[4].flush()
        run("C ABI, two independent pipelines (rpf_shard_develop_async), RCCL exchange with itself in each", PingPong(), band, out)
        for ln in lanes:
            ln[4].close()
    run("C ABI, one launch (world 1)", tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf), band, out)
    if 0 < y0 and y0 + rows < H:
        sh = R.Shard(gpu, None, 0, 1)
        sh.set_neighbours(0, 0)
        run("C ABI, RCCL exchange with itself, then one launch",
            tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=sh), band, out)
    run("torch schedule of round 2 (3 launches from Python, side stream)",
        tiles.BandDeveloper(gpu, W, H, y0, rows, raw, conf, force_split=True, schedule="side"), band, out, exchange=False)
