#!/usr/bin/env python3
"""bench.py — megapixels/s of the full RAW -> sRGB chain on a 6000x4000 Bayer frame (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the fused kernel (demosaic + black level/WB + camera matrix + the whole of
wgsl main: tone, brightness curve, OKLCH curves, gamma) over one synthetic frame band that is already
resident in HBM.  N = 1: BASELINE configs[2], the 6000x4000 frame.  N > 1: weak scaling — a
6000 x (4000 N) frame cut into N row bands, one per rank; every step each rank develops the interior of its
band while a side stream exchanges its 2-row Bayer halos with its neighbours (RCCL send/recv) and develops the
8 edge rows on either side (rawphotoforge_amd/tiles.py BandDeveloper).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, from HIP-event
kernel times on the launch stream) and `cpu_baseline` (the oracle on the host cores, rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 6000, 4000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PX = {"develop": 18, "batch": 18, "demosaic": 18, "apply": 32,   # u16 in + RGBA32F out
                "unpack": 3.5,                                  # 12-bit packed in (1.5 B) + u16 out
                "develop_packed": 17.75}                        # 14-bit packed in (1.75 B) + RGBA32F out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preheat", type=int, default=200,
                    help="untimed steps before the W warmup steps (same count on every rank): an idle MI355X needs "
                         "~50 launches (15 ms) to reach its steady clocks, and steady state is what a production "
                         "service sees (profiles/r01_experiments.md)")
    ap.add_argument("--workload", default="develop",
                    choices=["develop", "demosaic", "apply", "unpack", "develop_packed", "batch"],
                    help="develop = BASELINE configs[2] (the headline); batch = configs[4]: a pool of --frames different "
                         "6000x4000 frames per GPU, one whole frame per step through the same fused kernel, no halo exchange "
                         "(replicas; the pool is far larger than the 256-MiB Infinity Cache, so every input comes from HBM)")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE configs[3]: ONE 12000x8000 frame row-tiled over the N ranks (strong scaling; bands of "
                         "8000/N rows on even boundaries) instead of the default weak scaling (a 6000x4000 band per rank)")
    ap.add_argument("--frames", type=int, default=8, help="--workload batch: frames in each GPU's pool (8 x 8 GPUs = 64)")
    ap.add_argument("--params", default="D", choices=["D", "S", "G", "V", "T"])
    ap.add_argument("--data", default="U", choices=["U", "N"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--out", default="rgba", choices=["rgba", "rgb", "u8"], help="output layout of the fused kernel")
    ap.add_argument("--masks", type=int, default=1,
                    help="N = 1 GPU: total number of masks (main + N-1 user masks of blobs covering ~half of the frame, "
                         "each with edit set S); the headline figure is --masks 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1.  nccl (= RCCL) is the measured configuration; gloo lets N ranks "
                         "share ONE GPU (RCCL refuses that) to rehearse the N > 1 code path — its timings mean nothing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange halos, then one launch per band")
    ap.add_argument("--pack", default="msb", choices=["msb", "mipi"],
                    help="develop_packed: 14-bit MSB-first bit stream (default) or MIPI CSI-2 RAW14 groups")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="--workload batch: skip the streaming read (rpf_dev_prefetch) in front of every frame")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP-event pairs (kernel duration for `roofline`) are recorded around every n-th launch of the timed "
                         "region: an event pair costs ~3.5 us of launch gap, which would otherwise be charged to `value`")
    ap.add_argument("--cpu-rows", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import rawphotoforge_amd as R
    from rawphotoforge_amd import tiles
    from rawphotoforge_amd import synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    gpu = R.GpuProcessor(local)
    gpu.set_kernel_variant(args.variant)
    batch = args.workload == "batch"
    global W, H
    if args.strong:
        if args.workload not in ("develop", "develop_packed", "demosaic"):
            raise SystemExit("--strong applies to the row-tiled workloads")
        W, H = 12000, 8000
        frame_h = H
        y0, rows = tiles.split_rows(H, world)[rank]
    else:
        frame_h = H if batch else H * world      # batch: every rank develops whole 6000x4000 frames of its own
        y0, rows = (0, H) if batch else (rank * H, H)
    praw = R.RawParams(**syn.RAW)
    praw_c = praw.to_c()
    spec = syn.PARAM_SETS[args.params]
    gen = syn.bayer_uniform if args.data == "U" else syn.bayer_natural
    cfa = gen(rows, W, seed=20260630 + rank)          # this rank's band (rank 0 at N=1: THE frame)
    band = tiles.make_band_buffer(cfa, device="cuda")   # (rows + 4, W) with halo rows
    def configure(e):
        syn.configure(e, spec)
        for m in range(1, args.masks):   # user masks: smooth blobs thresholded at 0.5 (mask_range 0 -> v >= 0 keeps all,
            yy, xx = np.mgrid[0:rows, 0:W]  # so the blobs are shifted below zero outside)
            blob = np.sin(xx * (0.002 * m) + m) * np.cos(yy * (0.003 * m) - m)
            e.add_mask(f"user{m}", np.where(blob > 0, 1.0, -1.0).astype(np.float32))
            syn.configure(e, syn.PARAM_SETS["S"], mask_name=f"user{m}", main=False)

    ed = R.PhotoEditor.for_tile(gpu, W, frame_h, y0, rows)
    configure(ed)
    dev = tiles.BandDeveloper(gpu, W, frame_h, y0, rows, praw, configure, rank, 1 if batch else world,
                              overlap=not args.no_overlap)
    pool = None
    if batch:   # rank r holds frames r * frames .. (r + 1) * frames - 1 of the set (seeds 20260630 + i)
        nframes = max(1, args.frames)
        cfa = gen(rows, W, seed=20260630 + rank * nframes)          # frame 0 of this rank (the CPU baseline's frame)
        pool = [tiles.make_band_buffer(cfa, device="cuda")]
        pool += [tiles.make_band_buffer(gen(rows, W, seed=20260630 + rank * nframes + i), device="cuda")
                 for i in range(1, nframes)]
    out_fmt, out_bpp = {"rgba": (R.OUT_RGBA_F32, 16), "rgb": (R.OUT_RGB_F32, 12), "u8": (R.OUT_RGB_U8, 3)}[args.out]
    if args.workload != "develop" and args.out != "rgba":
        raise SystemExit("--out applies to the fused develop workload")
    out = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")   # large enough for every layout
    lin = None
    if args.workload == "apply":
        lin = torch.empty_like(out)
        R.dev_demosaic(gpu, band.data_ptr(), praw, W, frame_h, y0, rows, lin.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)

    packed = None
    if args.workload == "unpack":   # the integer stage in front: 12-bit MSB-first packed -> the band buffer's own rows
        packed = torch.from_numpy(syn.pack_msb(cfa & 0x0FFF, 12)).cuda()
    packed_band, packed_stride = None, 0
    if args.workload == "develop_packed":   # the same frame as "develop" (14-bit values), bit-packed incl. halo rows
        host_packed = (syn.pack_mipi if args.pack == "mipi" else syn.pack_msb)(band.cpu().numpy().view(np.uint16), 14)
        packed_stride = host_packed.shape[1]
        packed_band = torch.from_numpy(host_packed).cuda()
        dev_packed = tiles.BandDeveloper(gpu, W, frame_h, y0, rows, praw, configure, rank, world,
                                         overlap=not args.no_overlap,
                                         packed=(14, packed_stride, R.PACK_MIPI if args.pack == "mipi" else R.PACK_MSB_FIRST))
    stream = torch.cuda.current_stream().cuda_stream
    stride_ev = max(1, args.event_stride)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % stride_ev == 0 else None
          for i in range(args.steps)]

    nstep = [0]

    def step(i=None):
        if batch:   # one whole frame of the pool per step: no neighbours, nothing to exchange
            frame = pool[nstep[0] % len(pool)]
            if not args.no_prefetch:   # a fresh frame: one streaming read lands it in the Infinity Cache (inside the step time)
                R.dev_prefetch(gpu, frame.data_ptr(), frame.numel() * 2, stream)
            dev.step(frame, out, out_fmt, out_bpp, False, ev[i] if i is not None else None)
            nstep[0] += 1
            return
        if args.workload == "develop":   # halo exchange (N > 1) hidden behind the interior rows
            dev.step(band, out, out_fmt, out_bpp, True, ev[i] if i is not None else None)
            return
        if args.workload == "develop_packed":   # the same on the packed band (its halo rows are packed rows too)
            dev_packed.step(packed_band, out, R.OUT_RGBA_F32, 16, True, ev[i] if i is not None else None)
            return
        tiles.exchange_halos(band, rank, world)          # no-op at N = 1
        if i is not None and ev[i] is not None:
            ev[i][0].record()
        if args.workload == "unpack":
            R.dev_unpack_bayer(gpu, packed.data_ptr(), packed.shape[1], W, rows, 12, band[2:].data_ptr(), stream)
        elif args.workload == "demosaic":
            R.dev_demosaic(gpu, band.data_ptr(), praw, W, frame_h, y0, rows, out.data_ptr(), stream)
        else:
            ed.dev_apply(lin.data_ptr(), out.data_ptr(), R.OUT_RGBA_F32, stream)
        if i is not None and ev[i] is not None:
            ev[i][1].record()

    preheat_steps = max(0, args.preheat)
    for _ in range(preheat_steps):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kern_ms = sorted(e[0].elapsed_time(e[1]) for e in ev if e is not None)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    mp_per_step = (frame_h if args.strong else world * rows) * W / 1e6   # the whole job: all ranks' pixels per step
    value = mp_per_step * args.steps / dt
    bpp = BYTES_PER_PX[args.workload] if args.out == "rgba" else 2 + out_bpp
    timed_rows = (dev.parts[0][1] if args.workload in ("develop", "batch") else
                  dev_packed.parts[0][1] if args.workload == "develop_packed" else rows)   # the launch the events bracket
    achieved = bpp * timed_rows * W / (kern_avg_ms * 1e-3) / 1e9   # GB/s, algorithmic bytes of ONE launch

    result = {
        "metric": "megapixels/sec full RAW\u2192sRGB chain, 6000\u00d74000 Bayer; % HBM roofline",   # BASELINE.json, verbatim
        "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": ((f"ONE {W}x{H} frame in {world} row band(s) of ~{H // world} rows, " if args.strong else "")
                         + f"{W}x{rows} RGGB u16 -> RGBA32F per GPU, fused {'develop' if batch else args.workload} "
                         f"(demosaic+WB+camera matrix+tone+curves+OKLCH+gamma), edit set {args.params}, "
                         f"data ({args.data}), "
                         + (f"batch of {world * len(pool)} different frames, {len(pool)} per GPU, one frame per GPU per step "
                            f"(replicas, no exchange)" if batch else f"frame {W}x{frame_h} in {world} row band(s)")
                         + (f", input 14-bit {'MIPI CSI-2 RAW14 groups' if args.pack == 'mipi' else 'MSB-first bit stream'}"
                            if args.workload == "develop_packed" else "")
                         + (f", {args.masks} masks" if args.masks > 1 else "")
                         + (", REHEARSAL over gloo: ranks may share a GPU, timings are not a measurement"
                            if world > 1 and args.backend != "nccl" else "")
                         + (", 2-row halo exchange per step" + ("" if args.no_overlap else
                            " hidden behind the interior rows (edge rows launched after it)") if world > 1 and not batch else "")),
            "variant": args.variant, "out": args.out, "preheat_steps": preheat_steps, "kernel_ms_median": round(kern_ms[len(kern_ms) // 2], 4),
            "kernel_ms_min": round(kern_ms[0], 4),
            "kernel_timing": f"HIP-event pairs around {len(kern_ms)} of the {args.steps} timed launches (every {stride_ev}th)",
        },
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel_ms_avg": round(kern_avg_ms, 4), "bytes_per_px": bpp},
    }

    # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 wraps
    # the command): they come from the FETCH_SIZE / WRITE_SIZE passes of the same command, summarised by
    # tools/summarize_profiles.py into profiles/hbm_traffic.json (see profiles/*_summary.md).
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            tr = json.load(f).get(args.workload)
        if tr and world == 1 and args.params == "D" and args.data == "U" and not args.strong:
            result["roofline"]["traffic"] = tr["traffic_bytes"]
            result["roofline"]["traffic_unit"] = "HBM bytes per launch (algorithmic: %d)" % (bpp * rows * W)
            result["roofline"]["traffic_source"] = tr["source"]
            if "valu" in tr:   # the secondary ceiling SURVEY.md §8(d) asks to report when frac < 0.7
                result["roofline"]["valu"] = tr["valu"]
    except (OSError, ValueError):
        pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as O   # the checker, timed as the reported CPU baseline (never on the GPU path)
        from tests import util
        raw, p = util.raw_params(), util.oracle_params(spec)
        # every host core this process may use (the box shares its CPUs by cgroup quota, which OpenMP does
        # not see); as many whole frames as fit in ~args.cpu_seconds of wall time (>= 1)
        cores = min(O.max_threads(), len(os.sched_getaffinity(0)))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        O.set_threads(cores)
        O.develop(cfa, raw, [p], y0=0, rows=64)            # warm the library / thread pool
        host_out = np.zeros((rows, W, 4), np.float32)       # allocated and touched once, like the GPU's output
        times = []
        t_end = time.perf_counter() + args.cpu_seconds
        while not times or (time.perf_counter() < t_end and len(times) < 20):
            t1 = time.perf_counter()
            O.develop(cfa, raw, [p], out=host_out)
            times.append(time.perf_counter() - t1)
        ct = sorted(times)[len(times) // 2]
        # the same frame on ONE thread, on a bounded band, for the per-core figure
        O.set_threads(1)
        t1 = time.perf_counter()
        O.develop(cfa, raw, [p], y0=(rows - args.cpu_rows) // 2 & ~1, rows=args.cpu_rows)
        c1 = time.perf_counter() - t1
        # BASELINE configs[0]: python-legacy's CPU path (demosaic + WB + apply_gamma((2.222, 4.5)), editor.py:47-76) on a
        # 512x512 frame, as restated in the oracle (pinned by tests/golden/legacy_helpers.npz), 1 thread / all cores
        small = syn.bayer_uniform(512, 512, seed=20260630)
        legacy = {}
        for nthreads in (1, cores):
            O.set_threads(nthreads)
            O.legacy_develop(small, raw)
            ts = []
            for _ in range(9):
                t1 = time.perf_counter()
                O.legacy_develop(small, raw)
                ts.append(time.perf_counter() - t1)
            legacy[nthreads] = 512 * 512 / 1e6 / sorted(ts)[len(ts) // 2]
        O.set_threads(cores)
        result["cpu_baseline"] = {
            "value": round(W * rows / 1e6 / ct, 3), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": (f"oracle/rpf_oracle.c (C restatement of the same chain, OpenMP, {cores} threads) on the whole "
                       f"{W}x{rows} frame, median of {len(times)} passes ({sum(times):.1f} s wall, "
                       f"{sum(times) * cores:.0f} core-s); single thread on {args.cpu_rows} rows: "
                       f"{args.cpu_rows * W / 1e6 / c1:.3f} MP/s; python-legacy flavour (demosaic + WB + apply_gamma) "
                       f"on 512x512: {legacy[1]:.1f} MP/s on 1 thread, {legacy[cores]:.1f} MP/s on {cores}"),
        }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
