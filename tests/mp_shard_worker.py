"""One rank of tests/test_shard_multiproc.py (TEST INFRASTRUCTURE): launched by torch.distributed.run, W processes on ONE
GPU, librpf_hip.so's communicators through the test-only RCCL stand-in (RPF_RCCL_LIB).  Every rank develops ITS band of the
6000x4000 frame through the library's sharded step — in line on one stream, and as two independent step pipelines — and
checks the band against the oracle's rows of the whole frame.  Prints one line "rank r ok ..." or raises."""
import argparse
import os
import sys


def report(text):
    """One line to the pipe the ranks share, as ONE write (a pipe write of up to PIPE_BUF bytes is not split): text and newline
    written separately let another rank's line land between them."""
    sys.stdout.flush()
    os.write(1, (text.replace("\n", " ")[:3900] + "\n").encode())


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="parity", choices=["parity", "wedge"])
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import oracle as O
    import rawphotoforge_amd as R
    from rawphotoforge_amd import synthetic as syn
    from rawphotoforge_amd import tiles
    from tests import util

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    gpu = R.GpuProcessor(0)
    W, H = args.width, args.height
    y0, rows = tiles.split_rows(H, world)[rank]
    frames = [syn.bayer_uniform(H, W, seed=20260630 + i) for i in range(2)]   # two different frames, one per pipeline
    rp, praw = util.raw_params(), R.RawParams(**syn.RAW)
    spec = syn.PARAM_SETS["S"]

    ids = [None, None]
    if rank == 0:
        ids = [R.Shard.unique_id(), R.Shard.unique_id()]
    dist.broadcast_object_list(ids, src=0)
    shards = [R.Shard(gpu, ids[0], rank, world), R.Shard(gpu, ids[1], rank, world)]   # same order on every rank
    info = shards[0].info()
    assert (info["rank"], info["world"]) == (rank, world)
    assert info["up"] == (rank - 1 if rank > 0 else -1) and info["down"] == (rank + 1 if rank < world - 1 else -1)

    def band_of(frame):   # this rank's rows, halo rows ZEROED: only the exchange can fill them
        b = torch.zeros((rows + 4, W), dtype=torch.int16, device="cuda")
        b[2:rows + 2] = torch.from_numpy(np.ascontiguousarray(frame[y0:y0 + rows]).view(np.int16)).cuda()
        return b

    devs = [tiles.ShardedDeveloper(gpu, W, H, y0, rows, praw, lambda e: syn.configure(e, spec), shard=s) for s in shards]
    bands = [band_of(f) for f in frames]
    outs = [torch.empty((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    p = util.oracle_params(spec)
    want = [O.develop(f, rp, [p], y0=y0, rows=rows) for f in frames]

    def check(tag):
        for i in range(2):
            got = outs[i].cpu().numpy()
            d = util.ulp_diff(got, want[i])
            assert d.max() == 0, f"rank {rank} {tag} frame {i}: {int((d > 0).sum())} values differ (max {int(d.max())} ulp)"

    if args.mode == "wedge":
        # the stand-in drops rank 0's sends (RPF_STANDIN_DROP_SENDS_FROM=0): rank 1's exchange never completes.  The
        # library's own deadline must fire — an error naming the rank and the phase — instead of a hang.
        import time
        for s in shards:
            s.set_timeout(1500)
        t0 = time.time()
        devs[0].step_async(bands[0], outs[0], R.OUT_RGBA_F32, True, streams[0].cuda_stream)
        err = None
        try:
            shards[0].drain(streams[0].cuda_stream)
        except R.PhotoEditorError as ex:
            err = str(ex)
        dt = time.time() - t0
        if rank == 1:
            assert err is not None and dt < 10.0, (err, dt)
            assert "rank 1 of" in err and "halo exchange" in err and "not enqueued within 1500 ms" in err, err
        report(f"rank {rank} wedge: {'deadline fired after %.1f s: %s' % (dt, err) if err else 'no error (%.1f s)' % dt}")
        os._exit(0)   # the stuck thread sits in the stand-in until ITS timeout: leave without joining it

    # 1. exchange, then one launch, on one stream (rpf_shard_develop) — both frames through shard 0
    for i in range(2):
        outs[i].zero_()
        devs[0].step(bands[i], outs[i], R.OUT_RGBA_F32, 16, True)
    torch.cuda.synchronize()
    check("serial")
    # 2. two independent step pipelines (rpf_shard_develop_async on two communicators), frames alternating, K steps each
    for b in bands:   # halo rows back to zero: the async steps must fill them again
        b[:2].zero_()
        b[rows + 2:].zero_()
    for o in outs:
        o.zero_()
    for k in range(args.steps):
        for lane in range(2):
            devs[lane].step_async(bands[lane], outs[lane], R.OUT_RGBA_F32, True, streams[lane].cuda_stream, timed=(k == args.steps - 1))
    for lane in range(2):
        shards[lane].drain(streams[lane].cuda_stream)
    n0, ms0 = shards[0].timing()
    assert n0 == 1 and ms0 > 0
    check("pingpong")
    # 3. a setter and add_mask while steps are still queued (no flush in between), then destroy with steps queued
    ed = devs[0].ed
    for k in range(4):
        devs[0].step_async(bands[0], outs[0], R.OUT_RGBA_F32, True, streams[0].cuda_stream)
    ed.set_brightness_tone_curve(control_points_x=[0, 30000, 65535], control_points_y=[0, 40000, 65535])
    m = np.zeros((rows, W), np.float32)
    m[:, W // 2:] = 1.0
    ed.add_mask("right", m)               # grows the curve tables: the old ones are freed — after the queued steps
    devs[0].step_async(bands[0], outs[0], R.OUT_RGBA_F32, True, streams[0].cuda_stream)
    shards[0].drain(streams[0].cuda_stream)
    p2 = util.oracle_params(spec)
    p2.brightness_tone_curve = O.curve_from_control_points([0, 30000, 65535], [0, 40000, 65535])
    masks = np.stack([np.ones((rows, W), np.float32), O.binarize_mask(m)])
    want2 = O.develop(frames[0], rp, [p2, util.params_D()], masks=masks, y0=y0, rows=rows)
    d = util.ulp_diff(outs[0].cpu().numpy(), want2)
    assert d.max() == 0, f"rank {rank} after setters: {int((d > 0).sum())} values differ"
    for k in range(3):
        devs[1].step_async(bands[1], outs[1], R.OUT_RGBA_F32, True, streams[1].cuda_stream)
    devs[1].ed.close()                     # rpf_editor_destroy with steps queued: waits for them, no fault
    shards[1].drain(streams[1].cuda_stream)
    dist.barrier()
    torch.cuda.synchronize()
    for s in shards:
        s.close()
    report(f"rank {rank} ok: band [{y0}, {y0 + rows}) of {W}x{H}, serial + pingpong + setters-while-queued bit-identical to the oracle")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
