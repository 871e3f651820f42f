#!/usr/bin/env python3
"""Rehearsal of the N > 1 path on ONE GPU: start with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rehearse_ranks.py
(N <= 5 on the shared boxes: the launcher counts towards their limit of 6 processes per GPU.)  Every rank uses cuda:0 and the process group is gloo (RCCL refuses two ranks on one device; gloo moves CUDA tensors
through the host).  What is exercised is everything but the transport: the band split, tiles.BandDeveloper's
three-launch schedule with a REAL asynchronous halo exchange between ranks, repeated steps on the same buffers, a new
frame in the same buffers, and the result of every band against the oracle's whole-frame answer (bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import oracle as O
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles
from tests import util

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
gpu = R.GpuProcessor(0)
W, H = 1000, 96 * world + 10          # ragged: bands differ in size
rp, praw = util.raw_params(), util.to_raw(util.raw_params())
spec = syn.PARAM_SETS["S"]
p = util.oracle_params(spec)
y0, rows = tiles.split_rows(H, world)[rank]
ok = True
for schedule in ("side", "serial", "pipelined"):
    dev = tiles.BandDeveloper(gpu, W, H, y0, rows, praw, lambda e: syn.configure(e, spec), rank, world, schedule=schedule)
    assert dev.split
    band = None
    out = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
    for frame in range(2):            # a second frame through the SAME band buffer (cached P2P ops)
        cfa = syn.bayer_natural(H, W, seed=500 + frame)
        fresh = tiles.make_band_buffer(cfa[y0:y0 + rows], device="cuda")
        if band is None:
            band = fresh
        else:
            band.copy_(fresh)         # halo rows zeroed again: the exchange must refill them
        dist.barrier()
        for _ in range(3):            # repeated steps: stream ordering of band / out reuse
            dev.step(band, out)
        dev.join()
        torch.cuda.synchronize()
        want = O.develop(cfa, rp, [p], y0=y0, rows=rows)
        same = np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        ok = ok and same
        print(f"rank {rank}/{world} schedule {schedule} frame {frame}: rows [{y0}, {y0 + rows}) "
              f"{'bit-identical to the oracle' if same else 'MISMATCH'}", flush=True)
        dist.barrier()
# the same with a packed band (12-bit LSB-first): the halo rows exchanged are packed rows
bits, order = 12, R.PACK_LSB_FIRST
rp12 = O.RawParams(black=(200,) * 4, white=4095)
cfa = syn.bayer_natural(H, W, seed=600) & 0x0FFF
pad = np.zeros((rows + 4, W), np.uint16)
pad[2:-2] = cfa[y0:y0 + rows]
stride = O.packed_row_stride(W, bits)
pband = torch.from_numpy(O.pack_bayer(pad, bits, stride, order=order)).cuda()
devp = tiles.BandDeveloper(gpu, W, H, y0, rows, util.to_raw(rp12), lambda e: syn.configure(e, spec), rank, world,
                           packed=(bits, stride, order))
out = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
dist.barrier()
for _ in range(2):
    devp.step(pband, out)
torch.cuda.synchronize()
same = np.array_equal(out.cpu().numpy().view(np.uint32), O.develop(cfa, rp12, [p], y0=y0, rows=rows).view(np.uint32))
ok = ok and same
print(f"rank {rank}/{world} packed 12-bit LSB-first band: {'bit-identical to the oracle' if same else 'MISMATCH'}", flush=True)
dist.barrier()
t = torch.tensor([1.0 if ok else 0.0], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("REHEARSAL", "PASSED" if t.item() == 1.0 else "FAILED", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
