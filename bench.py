#!/usr/bin/env python3
"""bench.py — megapixels/s of the full RAW -> sRGB chain on a 6000x4000 Bayer frame (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the fused kernel (demosaic + black level/WB + camera matrix + the whole of wgsl main: tone,
brightness curve, OKLCH curves, gamma) over one synthetic frame that is already resident in HBM.

N = 1: BASELINE configs[2], the 6000x4000 frame.
N > 1 (default): STRONG scaling of that same frame, as BASELINE.json's north_star states it — THE 6000x4000 frame cut
into N row bands on even boundaries (tiles.split_rows), one band per rank; every step each rank exchanges its 2-row
Bayer halos with its neighbours (RCCL ncclSend/ncclRecv, issued by librpf_hip.so itself: include/rpf.h
rpf_shard_develop) and develops its band.  --strong: the same for BASELINE configs[3], ONE 12000x8000 frame.
--weak: a 6000 x (4000 N) frame, one 6000x4000 band per rank (round 2's default).  --workload batch: configs[4],
whole frames per GPU, no exchange.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, from HIP-event kernel times on
the launch stream), `cpu_baseline` (the oracle on the host cores, rank 0, N = 1) and, at N = 1 for the headline
workload, `secondary`: the same kernel under edit set S and on a pool of frames that does not fit the Infinity Cache.
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


# The card's shader clock and socket power while the timed region runs (hwmon freq1_input / power1_input of the card's sysfs
# node): the fused kernels are bound by VALU issue, so their time scales with the clock the power limit leaves them — which
# differs by ~5 % between boxes of one pool.  Sampled by a CHILD process that only reads two files (no GIL shared with the launch
# loop, no GPU touched), time-stamped with CLOCK_MONOTONIC like the region's ends.
_CLOCK_SAMPLER = r"""
import sys, time
paths = sys.argv[1:]
try:
    while True:
        t = time.monotonic()
        vals = []
        for p in paths:
            try:
                with open(p) as f:
                    vals.append(f.read().strip() or "-")
            except OSError:
                vals.append("-")
        sys.stdout.write("%.6f %s\n" % (t, " ".join(vals)))
        sys.stdout.flush()
        time.sleep(0.001)
except (BrokenPipeError, KeyboardInterrupt):
    pass
"""


def start_clock_sampler(pci_bus_id):
    """(process, None) or (None, reason): the sampler on the hwmon node of the card at `pci_bus_id`."""
    import glob
    import subprocess
    try:
        for dev in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(dev)).lower() != pci_bus_id.lower():
                continue
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                f, p = os.path.join(hw, "freq1_input"), os.path.join(hw, "power1_input")
                if os.access(f, os.R_OK):
                    # (under rocprofv3 this process carries the profiler's preload: the child reads two files and needs none of it)
                    env = {k: v for k, v in os.environ.items()
                           if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
                    return subprocess.Popen([sys.executable, "-c", _CLOCK_SAMPLER, f, p if os.access(p, os.R_OK) else "/nonexistent"],
                                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env), None
        return None, "no readable hwmon freq1_input for " + pci_bus_id
    except OSError as ex:
        return None, str(ex)


def stop_clock_sampler(proc):
    """Ends the sampler; its samples as (CLOCK_MONOTONIC seconds, MHz, W or None)."""
    proc.terminate()
    try:
        text = proc.communicate(timeout=5)[0]
    except Exception:
        proc.kill()
        return []
    rows = []
    for line in text.splitlines():
        parts = line.split()
        if len(parts) == 3 and parts[1].isdigit():
            rows.append((float(parts[0]), int(parts[1]) / 1e6, int(parts[2]) / 1e6 if parts[2].isdigit() else None))
    return rows


def clocks_between(rows, t_begin, t_end):
    """Median / min / max shader clock (MHz) and median power (W) of the samples inside [t_begin, t_end] (CLOCK_MONOTONIC); when
    the region is shorter than a few samples, of the 50 ms that end with it (the launches before it run the same kernel)."""
    import statistics
    inside = [r for r in rows if t_begin <= r[0] <= t_end]
    window = "the timed region"
    if len(inside) < 5:
        inside = [r for r in rows if t_end - 0.05 <= r[0] <= t_end]
        window = "the 50 ms that end with the timed region (the launches before it, of the same kernel, included)"
    if not inside:
        return None
    mhz = [r[1] for r in inside]
    watts = [r[2] for r in inside if r[2] is not None]
    return {"sclk_mhz_median": round(statistics.median(mhz), 1), "sclk_mhz_min": round(min(mhz), 1), "sclk_mhz_max": round(max(mhz), 1),
            "power_w_median": round(statistics.median(watts), 1) if watts else None, "samples": len(inside), "window": window,
            "source": "hwmon freq1_input / power1_input of the card, read every ~1 ms by a child process"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-clocks", action="store_true", help="do not sample the card's shader clock and power (roofline.clocks)")
    ap.add_argument("--preheat", type=int, default=1000,
                    help="untimed steps before the W warmup steps (same count on every rank): an idle MI355X needs "
                         "~50 launches (15 ms) to reach its clocks and a few hundred more to settle (a 20-step timed region "
                         "reads 0.5-2 %% higher after 1000 preheat steps, 0.16 s, than after 200), and steady state is what "
                         "a production service sees (profiles/r01_experiments.md)")
    ap.add_argument("--workload", default="develop",
                    choices=["develop", "demosaic", "apply", "unpack", "develop_packed", "batch"],
                    help="develop = BASELINE configs[2] (the headline); batch = configs[4]: a pool of --frames different "
                         "6000x4000 frames per GPU, one whole frame per step through the same fused kernel, no halo exchange "
                         "(replicas; the pool is far larger than the 256-MiB Infinity Cache, so every input comes from HBM)")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE configs[3]: ONE 12000x8000 frame row-tiled over the N ranks instead of the 6000x4000 one")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: weak scaling, a 6000x4000 band per rank of a 6000 x (4000 N) frame (the default is strong "
                         "scaling of ONE 6000x4000 frame, which is what BASELINE.json's north_star states)")
    ap.add_argument("--frames", type=int, default=8, help="--workload batch: frames in each GPU's pool (8 x 8 GPUs = 64)")
    ap.add_argument("--params", default="D", choices=["D", "S", "G", "V", "T", "B", "C"])
    ap.add_argument("--data", default="U", choices=["U", "N"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--precision", default="exact", choices=["exact", "fast"],
                    help="exact (default, the headline): every builtin correctly rounded, bit-identical to the oracle; fast = "
                         "RPF_PRECISION_FAST (include/rpf.h rpf_set_precision): binary32-only builtins within 1 ulp, contracted "
                         "matrix products — reported under secondary.fast_precision by the default command")
    ap.add_argument("--out", default="rgba", choices=["rgba", "rgb", "u8"], help="output layout of the fused kernel")
    ap.add_argument("--masks", type=int, default=1,
                    help="N = 1 GPU: total number of masks (main + N-1 user masks of blobs covering ~half of the frame, "
                         "each with edit set S); the headline figure is --masks 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1.  nccl (= RCCL) is the measured configuration; gloo lets N ranks "
                         "share ONE GPU (RCCL refuses that) to rehearse the N > 1 code path — its timings mean nothing")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "torch"],
                    help="N > 1: who issues the halo exchange.  rccl = librpf_hip.so's own communicator (rpf_shard_develop: one "
                         "C call per step); torch = torch.distributed batch_isend_irecv from Python (tiles.BandDeveloper, round "
                         "2's path); auto = rccl, falling back to torch if the library's communicator cannot be created on "
                         "every rank (always torch under --backend gloo)")
    ap.add_argument("--halo", default="step", choices=["step", "once"],
                    help="N > 1: step (default) = every step is a NEW frame: its halo rows are exchanged in every step; once = an "
                         "editing session on ONE resident frame, the reference's own use (the image is loaded once, then "
                         "apply_adjustments runs per slider move, photo-editor-web/src/lib.rs:198): the halo rows are exchanged "
                         "before the first step and every step is the band's launch alone (RPF_SHARD_NO_EXCHANGE).  Not the "
                         "headline: stated in config.workload")
    ap.add_argument("--schedule", default="auto", choices=["auto", "pingpong", "serial"],
                    help="N > 1, per step and rank: serial = exchange, then one launch, on one stream (rpf_shard_develop); pingpong = TWO "
                         "step pipelines that share nothing — two RCCL communicators (two rpf_shards), two streams, two band buffers, "
                         "each shard's own thread enqueuing its steps (rpf_shard_develop_async), frames alternating between them: one "
                         "frame's exchange and launch latency run beside the other frame's kernel without a cross-stream dependency.  "
                         "auto (default) = serial for the headline line — the one schedule whose every ordering is a stream order — and, "
                         "when the library's two communicators exist, the same K steps again as pingpong, reported under "
                         "secondary.pingpong (a stall there ends at the library's deadline with the serial line intact)")
    ap.add_argument("--pack", default="msb", choices=["msb", "mipi"],
                    help="develop_packed: 14-bit MSB-first bit stream (default) or MIPI CSI-2 RAW14 groups")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="--workload batch: skip the streaming read (rpf_dev_prefetch) in front of every frame")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP-event pairs (kernel duration for `roofline`) are recorded around every n-th launch of the timed "
                         "region: an event pair costs ~3.5 us of launch gap, which would otherwise be charged to `value`")
    ap.add_argument("--rehearse-ranks", type=int, default=0, metavar="M",
                    help="--gpus 1 only: run the band of rank M/2 of an M-rank job on this one GPU, through the N > 1 code "
                         "path of this file, with a one-rank RCCL communicator whose rank is its own neighbour above and "
                         "below (rpf_shard_set_neighbours(0, 0)).  `value` is this ONE band's rate; the line says REHEARSAL")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` measurements (edit set S, frame pool)")
    ap.add_argument("--secondary-steps", type=int, default=150)
    ap.add_argument("--cpu-rows", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        W, H = 12000, 8000
    if (args.strong or args.weak) and args.workload not in ("develop", "develop_packed", "demosaic"):
        raise SystemExit("--strong / --weak apply to the row-tiled workloads")
    rehearse = args.rehearse_ranks
    if rehearse and (world != 1 or rehearse < 2 or batch or args.workload not in ("develop", "develop_packed")):
        raise SystemExit("--rehearse-ranks M (M >= 2) is a one-process rehearsal of the fused develop workloads")
    vworld, vrank = (rehearse, rehearse // 2) if rehearse else (world, rank)   # the job whose band this process runs
    sharded = vworld > 1 and not batch
    if batch:              # every rank develops whole 6000x4000 frames of its own
        mode, frame_h, (y0, rows) = "replicas", H, (0, H)
    elif args.weak:
        mode, frame_h, (y0, rows) = "weak", H * vworld, (vrank * H, H)
    else:                  # ONE frame; at N = 1 the band is the frame
        mode, frame_h, (y0, rows) = "strong", H, tiles.split_rows(H, vworld)[vrank]
    praw = R.RawParams(**syn.RAW)
    spec = syn.PARAM_SETS[args.params]
    gen = syn.bayer_uniform if args.data == "U" else syn.bayer_natural
    if mode == "strong" and vworld > 1:   # this rank's rows of THE frame (seed 20260630)
        cfa = np.ascontiguousarray(gen(frame_h, W, seed=20260630)[y0:y0 + rows])
    else:
        cfa = gen(rows, W, seed=20260630 + rank)          # rank 0 at N = 1: THE frame
    band = tiles.make_band_buffer(cfa, device="cuda")   # (rows + 4, W) with halo rows

    def configure_with(pspec, precision=None):
        def configure(e):
            syn.configure(e, pspec)
            e.set_precision(R.PRECISION_FAST if (precision or args.precision) == "fast" else R.PRECISION_EXACT)
            for m in range(1, args.masks):   # user masks: smooth blobs thresholded at 0.5 (mask_range 0 -> v >= 0 keeps all,
                yy, xx = np.mgrid[0:rows, 0:W]  # so the blobs are shifted below zero outside)
                blob = np.sin(xx * (0.002 * m) + m) * np.cos(yy * (0.003 * m) - m)
                e.add_mask(f"user{m}", np.where(blob > 0, 1.0, -1.0).astype(np.float32))
                syn.configure(e, syn.PARAM_SETS["S"], mask_name=f"user{m}", main=False)
        return configure
    configure = configure_with(spec)

    # ---- who exchanges the halos at N > 1 -------------------------------------------------------------------------
    exchange_note, shard, shard_b = None, None, None
    # the pingpong schedule's second communicator — only where it will be used: an idle communicator is not free
    want_two = (args.schedule in ("auto", "pingpong") and args.halo == "step"
                and args.workload in ("develop", "develop_packed"))
    if rehearse:
        shard = R.Shard(gpu, None, 0, 1)
        shard.set_neighbours(0, 0)
        if want_two:
            shard_b = R.Shard(gpu, None, 0, 1)
            shard_b.set_neighbours(0, 0)
        exchange_note = ("REHEARSAL on one GPU: librpf_hip.so's ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on a one-rank "
                         "communicator, the rank its own neighbour above and below (no xGMI hop in this figure)")
    elif sharded:
        # (RPF_BENCH_TRY_RCCL: try the library's communicator under gloo too — on a one-GPU box it must fail on the duplicate
        # device and the fallback must engage: the rehearsal of this very branch)
        want_rccl = args.exchange in ("auto", "rccl") and (args.backend == "nccl" or bool(os.environ.get("RPF_BENCH_TRY_RCCL")))
        err = ""
        if want_rccl:
            ids = [None, "", None]   # two communicators: the pingpong schedule runs two independent step pipelines
            if rank == 0:
                try:
                    ids[0] = R.Shard.unique_id()
                    ids[2] = R.Shard.unique_id() if want_two else None
                except R.PhotoEditorError as ex:   # e.g. librccl cannot be loaded: the other ranks must hear of it
                    ids[0], ids[1] = None, str(ex)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is None:
                err = ids[1]
            else:
                try:
                    shard = R.Shard(gpu, ids[0], rank, world)
                    if want_two:
                        shard_b = R.Shard(gpu, ids[2], rank, world)
                except R.PhotoEditorError as ex:   # every rank must agree on the path: gathered below
                    err = str(ex)
            oks = [None] * world
            dist.all_gather_object(oks, err)
            if any(oks):
                if args.exchange == "rccl":
                    raise SystemExit("rpf_shard_create failed: " + "; ".join(o for o in oks if o))
                shard, shard_b, err = None, None, "; ".join(sorted({o for o in oks if o}))
        if shard is not None:
            exchange_note = ("librpf_hip.so: ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on its own RCCL communicator, "
                             "one rpf_shard_develop call per step")
        else:
            exchange_note = "torch.distributed batch_isend_irecv from Python (tiles.BandDeveloper)" + (
                f" — fallback, the library's communicator failed: {err}" if err else "")

    schedule = args.schedule
    halo_once = sharded and args.halo == "once"
    if halo_once:
        if args.workload not in ("develop", "develop_packed") or schedule not in ("auto", "serial"):
            raise SystemExit("--halo once: a fused develop workload, no --schedule")
        schedule = "serial"
    lib_sched_ok = sharded and shard is not None and args.workload in ("develop", "develop_packed")
    also_pingpong = schedule == "auto" and lib_sched_ok and shard_b is not None and not halo_once
    if schedule == "auto":
        schedule = "serial"
    if schedule == "pingpong" and not lib_sched_ok:
        raise SystemExit("--schedule pingpong needs the library's communicator (N > 1, --exchange rccl) and a fused develop workload")

    def make_developer(cfgfn, packed=None, lane_shard=None):
        if sharded and shard is not None:
            return tiles.ShardedDeveloper(gpu, W, frame_h, y0, rows, praw, cfgfn, shard=lane_shard or shard, packed=packed)
        return tiles.BandDeveloper(gpu, W, frame_h, y0, rows, praw, cfgfn, rank, world if sharded else 1,
                                   overlap=False, packed=packed)

    ed = R.PhotoEditor.for_tile(gpu, W, frame_h, y0, rows)
    configure(ed)
    dev = make_developer(configure)
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
        dev_packed = make_developer(configure, packed=(14, packed_stride, R.PACK_MIPI if args.pack == "mipi" else R.PACK_MSB_FIRST))
    stream = torch.cuda.current_stream().cuda_stream
    stride_ev = max(1, args.event_stride)

    def event_pairs(n):
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % stride_ev == 0 else None
                for i in range(n)]
    ev = event_pairs(args.steps)
    nstep = [0]

    def step(i=None):
        e = ev[i] if i is not None else None
        if batch:   # one whole frame of the pool per step: no neighbours, nothing to exchange
            frame = pool[nstep[0] % len(pool)]
            if e:   # the pair brackets the prefetch pass too: it is part of what a fresh frame costs
                e[0].record()
            if not args.no_prefetch:   # a fresh frame: one streaming read lands it in the Infinity Cache (inside the step time)
                R.dev_prefetch(gpu, frame.data_ptr(), frame.numel() * 2, stream)
            dev.step(frame, out, out_fmt, out_bpp, False)
            if e:
                e[1].record()
            nstep[0] += 1
            return
        if args.workload in ("develop", "develop_packed") and schedule == "pingpong":
            # frames alternate between two pipelines that share nothing (shard = communicator + thread, stream, band and
            # output buffer each); the call hands the step to the shard's thread: every step is one exchange + one launch
            lane = nstep[0] & 1
            pp_devs[lane].step_async(pp_bands[lane], pp_outs[lane], out_fmt if args.workload == "develop" else R.OUT_RGBA_F32,
                                     True, pp_streams[lane].cuda_stream, timed=e is not None)
            nstep[0] += 1
            return
        if args.workload == "develop":   # N > 1: halo exchange + band, one call
            dev.step(band, out, out_fmt, out_bpp, not halo_once, e)
            return
        if args.workload == "develop_packed":   # the same on the packed band (its halo rows are packed rows too)
            dev_packed.step(packed_band, out, R.OUT_RGBA_F32, 16, not halo_once, e)
            return
        if sharded:
            if shard is not None:
                shard.exchange(band.data_ptr(), W * 2, rows, stream)
            else:
                tiles.exchange_halos(band, rank, world)
        if e:
            e[0].record()
        if args.workload == "unpack":
            R.dev_unpack_bayer(gpu, packed.data_ptr(), packed.shape[1], W, rows, 12, band[2:].data_ptr(), stream)
        elif args.workload == "demosaic":
            R.dev_demosaic(gpu, band.data_ptr(), praw, W, frame_h, y0, rows, out.data_ptr(), stream)
        else:
            ed.dev_apply(lin.data_ptr(), out.data_ptr(), R.OUT_RGBA_F32, stream)
        if e:
            e[1].record()

    if halo_once:   # the session's one exchange: with the frame's upload, not with every edit
        if args.workload == "develop":
            dev.step(band, out, out_fmt, out_bpp, True, None)
        else:
            dev_packed.step(packed_band, out, R.OUT_RGBA_F32, 16, True, None)
        torch.cuda.synchronize()

    pp_devs = pp_bands = pp_outs = pp_streams = None
    if schedule == "pingpong" or also_pingpong:
        if args.workload == "develop":
            pp_devs = (dev, make_developer(configure, lane_shard=shard_b))
            pp_bands = (band, band.clone())
        else:
            pp_devs = (dev_packed, make_developer(configure, packed=dev_packed.packed, lane_shard=shard_b))
            pp_bands = (packed_band, packed_band.clone())
        pp_outs = (out, torch.empty_like(out))
        # DIFFERENT priorities: with equal ones the two pipelines can lock in phase (both exchanges together, then both
        # kernels sharing the CUs: 37 instead of 25 us per step on a 6000x500 band); a priority makes that state unstable
        pp_streams = (torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0))

    def drain():
        # pingpong: the shards' threads have enqueued everything handed to them, then their streams have drained; every wait
        # is the library's bounded one (rpf_shard_drain: RPF_WAIT_TIMEOUT_MS, 30 s) — a neighbour that never arrives ends
        # in a GpuComputeError naming rank and phase, not in a run killed at the driver's limit with nothing written
        if schedule == "pingpong":
            shard.drain(pp_streams[0].cuda_stream)
            shard_b.drain(pp_streams[1].cuda_stream)
        elif sharded and shard is not None:
            shard.drain(stream)
        torch.cuda.synchronize()

    sampler = None
    if rank == 0 and not args.no_clocks:
        try:
            _p = torch.cuda.get_device_properties(local)
            sampler, _ = start_clock_sampler("%04x:%02x:%02x.0" % (getattr(_p, "pci_domain_id", 0), getattr(_p, "pci_bus_id", 0),
                                                                   getattr(_p, "pci_device_id", 0)))
        except Exception:   # a figure beside the measurement, never a reason to lose it
            sampler = None
        if sampler is not None:   # whatever ends this run early, the child does not outlive it
            import atexit
            atexit.register(lambda p=sampler: p.poll() is None and p.kill())
    preheat_steps = max(0, args.preheat)
    for _ in range(preheat_steps):
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    if schedule == "pingpong":   # forget the launches timed so far
        shard.timing()
        shard_b.timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t0_mono = time.monotonic()
    for i in range(args.steps):
        step(i)
    drain()                                # this rank's K steps are done (device-wide synchronise inside)
    dt = time.perf_counter() - t0          # the job's time is the MAX of these over the ranks (below): the closing barrier
    t_head = (t0_mono, time.monotonic())
    # The sampler ends with the timed region.  (Its readings during the `secondary` legs were tried and are not reported: the
    # sensors average over a window, and after the host-side pauses between those short legs they showed 2333 MHz at 337 W for
    # a leg that ran slower than at 1940 — only a reading after ~0.2 s of continuous load, as here, is the kernel's clock.)
    samples = stop_clock_sampler(sampler) if sampler is not None else []
    if world > 1:                          # brackets the region but its own latency (a collective: ~0.1 ms, a fifth of 20
        dist.barrier()                     # steps on a 500-row band) is not part of any rank's K steps
    torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if schedule == "pingpong":   # the launches' own start / completion signals, summed inside the library (RPF_SHARD_TIMED)
        tn = [shard.timing(), shard_b.timing()]
        kern_avg_ms = (tn[0][1] + tn[1][1]) / max(1, tn[0][0] + tn[1][0])
        kern_ms = None   # the library sums the timed launches: a mean, no median / minimum
    else:
        kern_ms = sorted(e[0].elapsed_time(e[1]) for e in ev if e is not None)
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
    # ---- N > 1: the same K steps again through the two-pipeline schedule (secondary.pingpong) ----------------------------
    pp_result, stalled = None, False
    if also_pingpong:
        serial_schedule, schedule = schedule, "pingpong"
        err = ""
        try:
            nstep[0] = 0
            for _ in range(min(preheat_steps, 200) + args.warmup):
                step()
            drain()
            shard.timing()
            shard_b.timing()
        except R.PhotoEditorError as ex:
            err = str(ex)
        def agree(e):   # every rank learns whether all of them got through
            if world == 1:
                return [e]
            got = [None] * world
            dist.all_gather_object(got, e)
            return got
        errs = agree(err)
        if not any(errs):
            try:
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    step(i)
                drain()
                dt2 = time.perf_counter() - t1
                tn = [shard.timing(), shard_b.timing()]
            except R.PhotoEditorError as ex:
                err = str(ex)
            errs = agree(err)
        if any(errs):
            stalled = True
            pp_result = {"error": "; ".join(sorted({e for e in errs if e}))[:1500],
                         "note": "the two-pipeline schedule did not complete on this node; the headline line above is the serial schedule"}
        else:
            if world > 1:
                t = torch.tensor([dt2], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            job_mp = (rows if rehearse else (frame_h if mode == "strong" else world * rows)) * W / 1e6
            pp_result = {"value": round(job_mp * args.steps / dt2, 1), "unit": "MP/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                         "kernel_ms_mean": round((tn[0][1] + tn[1][1]) / max(1, tn[0][0] + tn[1][0]), 4),
                         "note": "schedule pingpong: TWO independent step pipelines per rank (two RCCL communicators with a thread each, "
                                 "two streams, two band and output buffers, frames alternating); every step is still one exchange + "
                                 "one launch.  Two frames are in flight, so this is throughput of a stream of frames, not the "
                                 "latency of one; `value` of this line is the serial schedule"}
        schedule = serial_schedule

    mp_per_step = (frame_h if mode == "strong" else world * rows) * W / 1e6   # the whole job: all ranks' pixels per step
    if rehearse:
        mp_per_step = rows * W / 1e6   # one band of the job is all that ran
    value = mp_per_step * args.steps / dt
    bpp = BYTES_PER_PX[args.workload] if args.out == "rgba" else 2 + out_bpp
    the_dev = dev_packed if args.workload == "develop_packed" else dev
    whole_step_events = sharded and shard is not None and args.workload in ("develop", "develop_packed") and schedule != "pingpong"
    timed_rows = rows if (whole_step_events or args.workload not in ("develop", "batch", "develop_packed")) else the_dev.parts[0][1]
    achieved = bpp * timed_rows * W / (kern_avg_ms * 1e-3) / 1e9   # GB/s, algorithmic bytes of what the events bracket

    # what the ranks are: world size as the process group reports it, backend, and each rank's device (distinct PCI bus
    # ids = distinct GPUs; the shard's view of its neighbours when the library exchanges)
    props = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "device": local, "name": props.name,
          "pci_bus_id": (shard.info()["pci_bus_id"] if shard is not None else
                         "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                               getattr(props, "pci_device_id", 0))),
          "band": [int(y0), int(rows)]}
    if shard is not None:
        me["neighbours"] = [shard.info()["up"], shard.info()["down"]]
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    if mode == "strong" and vworld > 1:
        shape_note = f"ONE {W}x{H} frame in {vworld} row bands of ~{H // vworld} rows (strong scaling), "
    elif mode == "weak" and vworld > 1:
        shape_note = f"frame {W}x{frame_h} in {vworld} row bands of {rows} rows (weak scaling), "
    else:
        shape_note = ""
    if rehearse:
        shape_note = (f"REHEARSAL of rank {vrank} of {vworld} on ONE GPU (value = this one band's rate, not a job's): " + shape_note)
    pristine = args.params == "D" and args.masks == 1
    result = {
        "metric": "megapixels/sec full RAW→sRGB chain, 6000×4000 Bayer; % HBM roofline",   # BASELINE.json, verbatim
        "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak" if (mode == "weak" or batch) else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (shape_note + f"{W}x{rows} RGGB u16 -> RGBA32F per GPU, fused {'develop' if batch else args.workload} "
                         f"(demosaic+WB+camera matrix+tone+curves+OKLCH+gamma), edit set {args.params}"
                         + (" (the reference's defaults = a photo that was just opened: the pristine-'main' specialisation "
                            "of the kernel, compile-time flags, identity brightness curve elided)" if pristine else "")
                         + f", data ({args.data}), "
                         + (f"batch of {world * len(pool)} different frames, {len(pool)} per GPU, one frame per GPU per step "
                            f"(replicas, no exchange)" if batch else
                            "the same frame every step: its 48-MB input stays in the 256-MiB Infinity Cache "
                            "(see secondary.fresh_input for input from HBM)")
                         + (f", input 14-bit {'MIPI CSI-2 RAW14 groups' if args.pack == 'mipi' else 'MSB-first bit stream'}"
                            if args.workload == "develop_packed" else "")
                         + (f", {args.masks} masks" if args.masks > 1 else "")
                         + (", REHEARSAL over gloo: ranks may share a GPU, timings are not a measurement"
                            if world > 1 and args.backend != "nccl" else "")
                         + (", EDITING SESSION on one resident frame: 2-row halo exchange once, before the first step; every "
                            "step is the band's launch alone (not the headline mode: --halo step exchanges in every step)"
                            if halo_once else
                            ", 2-row halo exchange per step, " + {
                             "serial": "then one launch over the band",
                             "pingpong": "inside each of TWO independent step pipelines (two RCCL communicators with a thread each, "
                                         "two streams, two band and output buffers; frames alternate): one frame's exchange runs "
                                         "beside the other frame's kernel; every step is one exchange + one launch",
                         }[schedule] if sharded else "")),
            "mode": mode, "schedule": (("none (halo once)" if halo_once else schedule) if sharded else None), "input_resident": "hbm" if batch else "infinity_cache",
            "variant": args.variant, "precision": args.precision, "out": args.out, "preheat_steps": preheat_steps,
            "kernel_ms_median": (round(kern_ms[len(kern_ms) // 2], 4) if kern_ms else None),
            "kernel_ms_min": (round(kern_ms[0], 4) if kern_ms else None),
            "kernel_timing": (f"HIP-event pairs around {len(kern_ms) if kern_ms else (tn[0][0] + tn[1][0])} of the {args.steps} timed "
                              + ("launches, by their own start / completion signals (hipExtLaunchKernelGGL events inside "
                                 "rpf_shard_develop_async; the exchange is a kernel of RCCL's in front of each)" if schedule == "pingpong" else
                                 "steps (exchange + launches of one rpf_shard_develop call)" if whole_step_events else
                                 "steps (prefetch pass + launch)" if batch else "launches") + f" (every {stride_ev}th)"),
            "world_size": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else None),
            "exchange": exchange_note, "ranks": ranks,
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
        if tr and world == 1 and not rehearse and args.params == "D" and args.data == "U" and not args.strong:
            result["roofline"]["traffic"] = tr["traffic_bytes"]
            result["roofline"]["traffic_measured_in_this_run"] = False
            result["roofline"]["traffic_unit"] = "HBM bytes per launch (algorithmic: %d)" % (bpp * rows * W)
            result["roofline"]["traffic_source"] = tr["source"]
            if "valu" in tr:   # the secondary ceiling SURVEY.md §8(d) asks to report when frac < 0.7
                result["roofline"]["valu"] = dict(tr["valu"])
    except (OSError, ValueError):
        pass

    # ---- secondary figures in the same line (N = 1, the headline workload): the kernel a user's edits dispatch (edit
    # set S: run-time flags, brightness curve in LDS, vignette) and input that really comes from HBM (a pool of frames
    # larger than the Infinity Cache, with and without the rpf_dev_prefetch pass, which is inside the timed pair)
    if (world == 1 and not rehearse and args.workload == "develop" and not args.no_secondary and args.masks == 1
            and args.out == "rgba"):
        nsec = max(8, args.secondary_steps)

        def timed(fn, n):
            for _ in range(300):   # ~50 ms of the leg's own kernel first: the clock it will be measured at (cf. --preheat)
                fn()
            torch.cuda.synchronize()
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in pairs:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in pairs)
            return sum(ts) / len(ts), ts[len(ts) // 2]

        def entry(avg, med, note):
            gbs = 18 * rows * W / (avg * 1e-3) / 1e9
            return {"kernel_ms_avg": round(avg, 4), "kernel_ms_median": round(med, 4), "MP/s": round(rows * W / 1e6 / (avg * 1e-3), 1),
                    "achieved_GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "note": note}
        sec = {}
        dev_s = make_developer(configure_with(syn.PARAM_SETS["S"]))
        sec["edit_set_S"] = entry(*timed(lambda: dev_s.step(band, out, out_fmt, out_bpp, False), nsec),
                                  "same frame (Infinity-Cache-resident), edit set S: run-time flags, brightness curve staged "
                                  "in LDS, vignette (its position-only part read from the editor's falloff map: +4 B/px of traffic), all tone "
                                  "sliders — the kernel a user's edits dispatch")
        del dev_s
        if args.precision == "exact":
            # RPF_PRECISION_FAST on the same frame and edit set: kernel time, and how far its pixels are from the exact
            # kernel's (= the oracle's, bit for bit: tests/test_gpu_parity.py).  Up to the brightness curve the two kernels
            # compute the same bits, so with this edit set the hue index is the only quantity that can flip.
            dev_f = make_developer(configure_with(spec, "fast"))
            fe = entry(*timed(lambda: dev_f.step(band, out, out_fmt, out_bpp, False), nsec),
                       "RPF_PRECISION_FAST specialisation of the same fused kernel (rpf_fast.h: binary32-only pow / sqrt / cos / sin "
                       "within 1 ulp, hue index from a binary32 polynomial, contracted 3x3 products), same frame, edit set "
                       + args.params + "; NOT the headline: `value` is the exact kernel")
            out_exact = torch.empty_like(out)
            dev.step(band, out_exact, out_fmt, out_bpp, False)
            dev_f.step(band, out, out_fmt, out_bpp, False)
            torch.cuda.synchronize()
            dd = (out[..., :3] - out_exact[..., :3]).abs().amax(dim=2)
            ulp = (out[..., :3].contiguous().view(torch.int32) - out_exact[..., :3].contiguous().view(torch.int32)).abs()
            hist = torch.bincount(ulp.flatten().clamp(max=8), minlength=9).tolist()
            fe["against_exact_kernel"] = {
                "pixels": int(dd.numel()), "bit_identical_pixels": round(float((dd == 0).sum()) / dd.numel(), 5),
                "ulp_histogram_per_channel_0_to_8plus": [round(h / ulp.numel(), 5) for h in hist],
                "median_ulp": float(ulp.flatten().float().median()),
                "pixels_within_1e-5": round(float((dd <= 1e-5).sum()) / dd.numel(), 6),
                "index_flip_rate": round(float((dd > 1e-5).sum()) / dd.numel(), 6),
                "index_flip_rate_note": "share of pixels further than 1e-5 from the exact kernel's: a hue index that differs by one "
                                        "moves the hue by 1/65535 of a turn (the brightness-curve indices cannot differ)",
                "max_abs_difference": round(float(dd.max()), 7)}
            sec["fast_precision"] = fe
            del dev_f, out_exact, dd, ulp
            dev_fs = make_developer(configure_with(syn.PARAM_SETS["S"], "fast"))   # ... and under the edits a user's session dispatches
            sec["fast_precision_edit_set_S"] = entry(*timed(lambda: dev_fs.step(band, out, out_fmt, out_bpp, False), nsec),
                                                     "RPF_PRECISION_FAST, edit set S (run-time flags, brightness curve in LDS, vignette map, all tone sliders)")
            del dev_fs
        npool = 8
        frames = [band] + [tiles.make_band_buffer(gen(rows, W, seed=20260700 + i), device="cuda") for i in range(1, npool)]
        k = [0]

        def fresh(prefetch):
            f = frames[k[0] % npool]
            k[0] += 1
            if prefetch:
                R.dev_prefetch(gpu, f.data_ptr(), f.numel() * 2, stream)
            dev.step(f, out, out_fmt, out_bpp, False)
        sec["fresh_input"] = {
            "pool": f"{npool} different {W}x{rows} frames ({npool * 48} MB of input + the 384-MB output between two uses of a "
                    f"frame: nothing stays in the 256-MiB Infinity Cache), edit set {args.params}",
            "with_prefetch": entry(*timed(lambda: fresh(True), nsec), "rpf_dev_prefetch (streaming read) + fused kernel, both inside the event pair"),
            "without_prefetch": entry(*timed(lambda: fresh(False), nsec), "fused kernel alone, its row loads served from HBM"),
        }
        del frames
        result["secondary"] = sec

    # the card's clock and power during the headline's timed region
    c = clocks_between(samples, *t_head) if samples else None
    if c:
        result["roofline"]["clocks"] = c
        v = result["roofline"].get("valu")
        if v and v.get("wave_instructions_per_launch"):   # at THIS run's kernel time and clock: cycles per wave-instruction per SIMD
            ns = kern_avg_ms * 1e6 * 1024 / v["wave_instructions_per_launch"]
            v["cycles_per_instruction_per_simd_in_this_run"] = round(ns * c["sclk_mhz_median"] / 1e3, 3)

    if rank == 0 and world == 1 and not rehearse and not args.no_cpu_baseline:
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
            # BASELINE configs[0] / north_star "python-legacy's CPU path timed on the node's own host cores": the oracle's
            # restatement of python-legacy's demosaic + WB + apply_gamma (pinned by tests/golden/legacy_helpers.npz)
            "legacy_512x512_mp_s_1t": round(legacy[1], 2), "legacy_512x512_mp_s_all_cores": round(legacy[cores], 2),
            "legacy_cores": cores, "single_thread_mp_s": round(args.cpu_rows * W / 1e6 / c1, 3),
        }
    if pp_result is not None:
        result.setdefault("secondary", {})["pingpong"] = pp_result
    if rank == 0:
        print(json.dumps(result), flush=True)
    if stalled:   # kernels of the stalled pipelines still sit on the device: leave without waiting for them
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        if shard is not None:
            torch.cuda.synchronize()
            shard.close()
            if shard_b is not None:
                shard_b.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
