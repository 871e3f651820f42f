#!/usr/bin/env python3
"""The schedules of a sharded step on ONE GPU (the rank is its own neighbour: a real RCCL group on a one-rank
communicator): interior + edges without an exchange, the exchange hidden behind the interior (RPF_SHARD_OVERLAP), exchange
then one launch, and the exchange one frame ahead on the side stream (rpf_shard_exchange_ahead, two band buffers) — each
with the shard's events as shipped and with hipEventDisableSystemFence (RPF_SHARD_EVENT_FLAGS=0x20000000)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles

gpu = R.GpuProcessor(0)
raw = R.RawParams(**syn.RAW)
params = sys.argv[1] if len(sys.argv) > 1 else "D"
conf = lambda ed: syn.configure(ed, syn.PARAM_SETS[params])


def run(dev, band, out, n=400, pipelined=None, **kw):
    def steps(m):
        if pipelined is None:
            for _ in range(m):
                dev.step(band, out, **kw)
            return
        bands = (band, pipelined)  # two band buffers: the exchange of the next one runs beside this one's kernel
        dev.exchange_ahead(bands[0])
        for k in range(m - 1):
            dev.exchange_ahead(bands[(k + 1) & 1], after_develop=True)
            dev.step(bands[k & 1], out, **kw)
        dev.step(bands[(m - 1) & 1], out, **kw)

    steps(100)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        steps(n)
        t1 = time.perf_counter()
        dev.join()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = (1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n)
        best = r if best is None or r[1] < best[1] else best
    return best


for W, H, y0, rows in ((6000, 4000, 1500, 500), (12000, 8000, 3000, 1000)):
    print(f"band {W} x {rows} of a {W} x {H} frame, edit set {params}: host enqueue / GPU-complete us per step", flush=True)
    band = tiles.make_band_buffer(syn.bayer_uniform(rows, W, seed=1), device="cuda")
    out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
    band2 = band.clone()
    h, g = run(tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf), band, out)
    print(f"  one launch, no exchange                          {h:6.1f} {g:6.1f}", flush=True)
    for evf in ("0", "0x20000000"):
        for rep in range(2):
            os.environ["RPF_SHARD_EVENT_FLAGS"] = evf
            sh = R.Shard(gpu, None, 0, 1)
            sh.set_neighbours(0, 0)
            a = run(tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=sh, force_split=True), band, out, exchange=False)
            b = run(tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=sh, overlap=True), band, out)
            c = run(tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=sh), band, out)
            d = run(tiles.ShardedDeveloper(gpu, W, H, y0, rows, raw, conf, shard=sh), band, out, pipelined=band2)
            print(f"  event flags {evf:>10s}: split, no exchange {a[0]:5.1f} {a[1]:5.1f} | hidden exchange "
                  f"{b[0]:5.1f} {b[1]:5.1f} | exchange then launch {c[0]:5.1f} {c[1]:5.1f} | exchange one frame ahead "
                  f"{d[0]:5.1f} {d[1]:5.1f}", flush=True)
            sh.close()
