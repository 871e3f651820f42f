"""The sharded step behind the C ABI (include/rpf.h rpf_shard_*; rawphotoforge_amd/csrc/rpf_shard.cpp): the code RCCL
drives on an 8-GPU node, exercised on ONE GPU — with pre-filled halos at the per-rank geometries of BASELINE
configs[2] over 8 GPUs (6000 x 500 bands) and configs[3] (12000 x 1000 bands), and with a real RCCL exchange in which a
rank is its own neighbour above and below (its first rows land in its top halo, its last rows in its bottom halo).
All comparisons are bit-for-bit against the oracle's answer for the band."""
import numpy as np
import pytest

import oracle as O
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn
from rawphotoforge_amd import tiles
from tests import util

gpu_only = pytest.mark.gpu


def _exact(got, want):
    d = util.ulp_diff(np.asarray(got, np.float32).reshape(want.shape), want)
    assert d.max() == 0, f"{int((d > 0).sum())} of {d.size} values differ, max {int(d.max())} ulp"


def _band_with_halos(frame, y0, rows):
    """(rows + 4, W) int16 view of frame rows y0 - 2 .. y0 + rows + 1 (zero outside the frame: never read)."""
    h, w = frame.shape
    pad = np.zeros((rows + 4, w), np.uint16)
    lo, hi = max(y0 - 2, 0), min(y0 + rows + 2, h)
    pad[lo - (y0 - 2):hi - (y0 - 2)] = frame[lo:hi]
    return pad


@gpu_only
@pytest.mark.parametrize("geom", [(6000, 4000, 8, 3, "D"), (6000, 4000, 8, 7, "S"), (12000, 8000, 8, 0, "D"),
                                  (12000, 8000, 8, 4, "S"), (300, 231, 3, 1, "G"), (300, 231, 3, 2, "S")])
def test_band_with_prefilled_halos_equals_oracle(gpu, geom):
    """One rank's band through rpf_shard_develop with its halos already in place: the per-rank geometry of the 8-GPU
    configurations, against the oracle's rows of the whole frame (the oracle computes only the band: a whole 96-MP
    frame would take minutes)."""
    torch = pytest.importorskip("torch")
    w, h, world, rank, pname = geom
    y0, rows = tiles.split_rows(h, world)[rank]
    frame = util.bayer_uniform(h, w, seed=77)
    rp = util.raw_params()
    spec = syn.PARAM_SETS[pname]
    want = O.develop(frame, rp, [util.oracle_params(spec)], y0=y0, rows=rows)
    band = torch.from_numpy(_band_with_halos(frame, y0, rows).view(np.int16)).cuda()
    out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    sd = tiles.ShardedDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), lambda e: syn.configure(e, spec))
    for _ in range(3):  # repeated steps: stream order covers the reuse of band and out
        sd.step(band, out, R.OUT_RGBA_F32, exchange=False)
    torch.cuda.synchronize()
    _exact(out.cpu().numpy(), want)


@gpu_only
def test_bands_with_user_masks_and_odd_rows(gpu):
    """Band-sized user masks on bands with an odd number of rows and an odd first row."""
    torch = pytest.importorskip("torch")
    h, w = 207, 140
    cfa = util.bayer_uniform(h, w, seed=5)
    rp = util.raw_params()
    rng = np.random.default_rng(21)
    main, a, b = util.params_D(), util.params_S(), util.params_G()
    ma = np.where(rng.random((h, w)) > 0.5, 1.0, -1.0).astype(np.float32)
    mb = np.full((h, w), -1.0, np.float32)
    mb[4:150, 30:120] = 1.0
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(ma), O.binarize_mask(mb)])
    want = O.develop(cfa, rp, [main, a, b], masks=masks)
    for y0, rows in ((0, 91), (91, 116), (33, 47)):
        def configure(e, y0=y0, rows=rows):
            e.add_mask("a", ma[y0:y0 + rows])
            e.add_mask("b", mb[y0:y0 + rows].reshape(-1))
            util.configure_editor(e, main)
            util.configure_editor(e, a, "a", main=False)
            util.configure_editor(e, b, "b", main=False)
        band = torch.from_numpy(_band_with_halos(cfa, y0, rows).view(np.int16)).cuda()
        out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
        sd = tiles.ShardedDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), configure)
        assert sd.ed.num_masks == 3
        sd.step(band, out, exchange=False)
        torch.cuda.synchronize()
        _exact(out.cpu().numpy(), want[y0:y0 + rows])


@gpu_only
@pytest.mark.parametrize("pattern", [O.CFA_RGGB, O.CFA_BGGR, O.CFA_GRBG, O.CFA_GBRG])
def test_bands_on_every_cfa_phase_and_odd_first_rows(gpu, pattern):
    """The horizontal-pair kernel starts its strips on an R column and its steps on an R row: for CFA layouts that begin with
    G or B, and for bands whose first row is odd, a strip's first column / a band's first step is virtual.  Every pixel
    of the band must still be written exactly once, with the oracle's bits (single mask: k_hp; ragged width)."""
    torch = pytest.importorskip("torch")
    h, w = 211, 333
    cfa = util.bayer_uniform(h, w, seed=40 + pattern)
    rp = O.RawParams(cfa_pattern=pattern, black=(500, 520, 530, 510), wb=(2.1, 1.0, 1.02, 1.6))
    spec = syn.PARAM_SETS["S"]
    want = O.develop(cfa, rp, [util.oracle_params(spec)])
    for y0, rows in ((0, 91), (91, 120), (33, 47), (64, 32), (1, 209)):
        band = torch.from_numpy(_band_with_halos(cfa, y0, rows).view(np.int16)).cuda()
        out = torch.full((rows, w, 4), -7.0, dtype=torch.float32, device="cuda")
        sd = tiles.ShardedDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), lambda e: syn.configure(e, spec))
        sd.step(band, out, exchange=False)
        torch.cuda.synchronize()
        _exact(out.cpu().numpy(), want[y0:y0 + rows])


@gpu_only
def test_rccl_exchange_with_itself_as_neighbour(gpu):
    """ncclCommInitRank / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd as rpf_shard_develop issues them, on a
    one-rank communicator whose rank is its own neighbour above and below.  The band is the middle of a frame whose
    rows above / below it are the band's own first / last two rows: the exchanged halos must reproduce the oracle's
    answer for that frame, with garbage in the halo rows beforehand."""
    torch = pytest.importorskip("torch")
    w, rows = 520, 96
    band_rows = util.bayer_uniform(rows, w, seed=9)
    frame = np.concatenate([band_rows[:2], band_rows, band_rows[-2:]])   # y0 = 2: the halos ARE those copies
    h = frame.shape[0]
    rp = util.raw_params()
    spec = syn.PARAM_SETS["S"]
    want = O.develop(frame, rp, [util.oracle_params(spec)], y0=2, rows=rows)
    pad = np.full((rows + 4, w), 12345, np.uint16)
    pad[2:-2] = band_rows
    band = torch.from_numpy(pad.view(np.int16)).cuda()
    out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    shard = R.Shard(gpu, None, 0, 1)
    shard.set_neighbours(0, 0)
    info = shard.info()
    assert (info["rank"], info["world"], info["up"], info["down"]) == (0, 1, 0, 0) and ":" in info["pci_bus_id"]
    sd = tiles.ShardedDeveloper(gpu, w, h, 2, rows, util.to_raw(rp), lambda e: syn.configure(e, spec), shard=shard)
    for _ in range(3):
        sd.step(band, out)
    torch.cuda.synchronize()
    got_band = band.cpu().numpy().view(np.uint16)
    assert np.array_equal(got_band[:2], band_rows[:2]) and np.array_equal(got_band[-2:], band_rows[-2:])
    _exact(out.cpu().numpy(), want)


@gpu_only
def test_sharded_step_with_more_masks_than_the_fused_kernel_takes(gpu):
    """A band whose editor holds 20 masks (the reference has no limit, lib.rs:481-499): the step becomes the halo exchange, the
    demosaic-only kernel and the RGBA pass on the band (rpf_shard.cpp launch_band) — synchronous and handed to the shard's thread,
    timed — and a packed band says what to do instead."""
    torch = pytest.importorskip("torch")
    w, rows, n = 260, 64, 20
    band_rows = util.bayer_natural(rows, w, seed=19)
    frame = np.concatenate([band_rows[:2], band_rows, band_rows[-2:]])   # the rank is its own neighbour: y0 = 2
    h = frame.shape[0]
    rp = util.raw_params()
    rng = np.random.default_rng(20)
    params = [util.random_params(rng, trivial_prob=0.4) for _ in range(n)]
    for p in params:
        p.lens_distortion = 0
    raw_masks = [(rng.random((rows, w)) > 0.7).astype(np.float32) for _ in range(n - 1)]
    masks = np.ones((n, rows, w), np.float32)   # band-sized, like the editor's
    for m in range(1, n):
        masks[m] = O.binarize_mask(raw_masks[m - 1])
    want = O.develop(frame, rp, params, masks=masks, y0=2, rows=rows)

    def configure(e):
        for m in range(1, n):
            e.add_mask(f"m{m}", raw_masks[m - 1])
        for m, p in enumerate(params):
            util.configure_editor(e, p, None if m == 0 else f"m{m}", main=(m == 0))
    pad = np.full((rows + 4, w), 999, np.uint16)
    pad[2:-2] = band_rows
    band = torch.from_numpy(pad.view(np.int16)).cuda()
    out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    shard = R.Shard(gpu, None, 0, 1)
    shard.set_neighbours(0, 0)
    sd = tiles.ShardedDeveloper(gpu, w, h, 2, rows, util.to_raw(rp), configure, shard=shard)
    assert sd.ed.num_masks == n
    sd.step(band, out)
    torch.cuda.synchronize()
    _exact(out.cpu().numpy(), want)
    out.zero_()
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        sd.step_async(band, out, stream=s, timed=True)
    shard.drain(s)
    torch.cuda.synchronize()
    _exact(out.cpu().numpy(), want)
    launches, total_ms = shard.timing()
    assert launches == 3 and total_ms > 0
    packed = torch.zeros((rows + 4, (w * 12 + 7) // 8), dtype=torch.uint8, device="cuda")
    sdp = tiles.ShardedDeveloper(gpu, w, h, 2, rows, util.to_raw(rp), configure, shard=shard, packed=(12, packed.shape[1], R.PACK_MSB_FIRST))
    with pytest.raises(R.InvalidArgument, match="unpack"):
        sdp.step(packed, out, exchange=False)


@gpu_only
def test_two_independent_step_pipelines(gpu):
    """rpf_shard_develop_async: steps handed to the shards' own threads.  Two shards (two one-rank communicators, each rank
    its own neighbour), two streams, two band buffers with DIFFERENT frames and garbage halos, frames alternating for ten
    steps; then flush + synchronise: each output must be the oracle's for the frame of its pipeline, and the timed launches
    must have been counted."""
    torch = pytest.importorskip("torch")
    w, rows = 520, 96
    rp = util.raw_params()
    spec = syn.PARAM_SETS["S"]
    lanes = []
    for seed in (31, 32):
        br = util.bayer_uniform(rows, w, seed=seed)
        frame = np.concatenate([br[:2], br, br[-2:]])
        want = O.develop(frame, rp, [util.oracle_params(spec)], y0=2, rows=rows)
        pad = np.full((rows + 4, w), 4000 + seed, np.uint16)
        pad[2:-2] = br
        shard = R.Shard(gpu, None, 0, 1)
        shard.set_neighbours(0, 0)
        lanes.append(dict(want=want, br=br, band=torch.from_numpy(pad.view(np.int16)).cuda(),
                          out=torch.full((rows, w, 4), -7.0, dtype=torch.float32, device="cuda"), shard=shard,
                          stream=torch.cuda.Stream(),
                          dev=tiles.ShardedDeveloper(gpu, w, rows + 4, 2, rows, util.to_raw(rp), lambda e: syn.configure(e, spec),
                                                     shard=shard)))
    torch.cuda.synchronize()
    for k in range(10):
        ln = lanes[k & 1]
        ln["dev"].step_async(ln["band"], ln["out"], stream=ln["stream"].cuda_stream, timed=(k % 2 == 0) or k == 9)
    for ln in lanes:
        ln["shard"].flush()
    torch.cuda.synchronize()
    for ln in lanes:
        got = ln["band"].cpu().numpy().view(np.uint16)
        assert np.array_equal(got[:2], ln["br"][:2]) and np.array_equal(got[-2:], ln["br"][-2:])
        _exact(ln["out"].cpu().numpy(), ln["want"])
    n0, ms0 = lanes[0]["shard"].timing()
    n1, ms1 = lanes[1]["shard"].timing()
    assert (n0, n1) == (5, 1) and ms0 > 0 and ms1 > 0
    assert lanes[0]["shard"].timing() == (0, 0.0)          # counted once
    # unknown flags are refused at the call; a failure inside a handed-over step surfaces at flush
    ln = lanes[0]
    with pytest.raises(R.InvalidArgument):
        ln["shard"].develop_async(ln["dev"].ed, ln["band"].data_ptr(), ln["dev"].raw, ln["out"].data_ptr(), flags=2)
    for ln in lanes:
        ln["shard"].close()


@gpu_only
def test_async_pipelines_under_load(gpu):
    """1500 steps through two pipelines with flushes, timing reads and an edit-state change in between (after a flush, as
    include/rpf.h asks): no step lost, the outputs follow the edit state, the timed launches add up."""
    torch = pytest.importorskip("torch")
    w, rows = 264, 48
    rp = util.raw_params()
    specs = [syn.PARAM_SETS["D"], syn.PARAM_SETS["S"]]
    lanes = []
    for i in range(2):
        br = util.bayer_uniform(rows, w, seed=70 + i)
        frame = np.concatenate([br[:2], br, br[-2:]])
        pad = np.zeros((rows + 4, w), np.uint16)
        pad[2:-2] = br
        shard = R.Shard(gpu, None, 0, 1)
        shard.set_neighbours(0, 0)
        lanes.append(dict(frame=frame, band=torch.from_numpy(pad.view(np.int16)).cuda(),
                          out=torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda"), shard=shard,
                          stream=torch.cuda.Stream(priority=-i),
                          dev=tiles.ShardedDeveloper(gpu, w, rows + 4, 2, rows, util.to_raw(rp), lambda e: syn.configure(e, specs[0]),
                                                     shard=shard)))
    timed = [0, 0]
    for phase, spec in enumerate(specs):
        if phase:   # everything handed over has been enqueued and has run: now the edit state may change
            for ln in lanes:
                syn.configure(ln["dev"].ed, spec)
        for k in range(750):
            ln = lanes[k & 1]
            t = k % 7 == 0
            timed[k & 1] += t
            ln["dev"].step_async(ln["band"], ln["out"], stream=ln["stream"].cuda_stream, timed=t)
            if k % 200 == 199:
                lanes[0]["shard"].flush()
        for ln in lanes:
            ln["shard"].flush()
        torch.cuda.synchronize()
        for ln in lanes:
            _exact(ln["out"].cpu().numpy(), O.develop(ln["frame"], rp, [util.oracle_params(spec)], y0=2, rows=rows))
    got = [ln["shard"].timing() for ln in lanes]
    assert [g[0] for g in got] == timed and all(g[1] > 0 for g in got)
    for ln in lanes:
        ln["shard"].close()


@gpu_only
def test_packed_band_through_the_sharded_step(gpu):
    """12-bit LSB-first packed rows (halo rows packed too) through rpf_shard_develop."""
    torch = pytest.importorskip("torch")
    h, w, y0, rows = 160, 256, 40, 80
    frame = util.bayer_uniform(h, w, seed=3) & 0x0FFF
    rp = O.RawParams(black=(256,) * 4, white=4095.0)
    spec = syn.PARAM_SETS["S"]
    want = O.develop(frame, rp, [util.oracle_params(spec)], y0=y0, rows=rows)
    packed = O.pack_bayer(_band_with_halos(frame, y0, rows), 12, order=O.PACK_LSB_FIRST)
    pb = torch.from_numpy(np.ascontiguousarray(packed)).cuda()
    out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    sd = tiles.ShardedDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), lambda e: syn.configure(e, spec),
                                packed=(12, packed.shape[1], R.PACK_LSB_FIRST))
    sd.step(pb, out, exchange=False)
    torch.cuda.synchronize()
    _exact(out.cpu().numpy(), want)


@gpu_only
def test_shard_argument_errors(gpu):
    torch = pytest.importorskip("torch")
    with pytest.raises(R.InvalidArgument):
        R.Shard(gpu, None, 0, 2)           # more than one rank needs the id
    with pytest.raises(R.InvalidArgument):
        R.Shard(gpu, None, 3, 2)
    sh = R.Shard(gpu, None, 0, 1)
    with pytest.raises(R.InvalidArgument):
        sh.set_neighbours(1, -1)           # outside the communicator
    ed = R.PhotoEditor.for_tile(gpu, 64, 64, 0, 64)
    ed.set_lens_distortion_correction(10)
    band = torch.zeros((68, 64), dtype=torch.int16, device="cuda")
    out = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(R.InvalidArgument):  # a whole-frame resample does not shard
        sh.develop(ed, band.data_ptr(), util.to_raw(util.raw_params()), out.data_ptr())


def test_shard_entry_points_refuse_null_handles(hip_lib):
    """Device-free: the sharding entry points exist, reject null handles with RPF_ERR_INVALID_ARGUMENT and leave a
    message behind (a processor cannot exist without a GPU, so nothing behind them can compute here)."""
    import ctypes as C
    from rawphotoforge_amd import _native as N
    lib = N.lib()
    h = C.c_void_p()
    assert lib.rpf_shard_create(None, None, 0, 1, C.byref(h)) == N.ERR_INVALID_ARGUMENT and not h.value
    assert N.last_error()
    assert lib.rpf_shard_set_neighbours(None, 0, 0) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_exchange(None, None, 0, 0, None) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_drain(None, None) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_set_timeout(None, 0) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_develop_async(None, None, None, 0, 0, 0, None, None, 0, None, 0) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_flush(None) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_timing(None, None, None) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_develop(None, None, None, 0, 0, 0, None, None, 0, None, 0) == N.ERR_INVALID_ARGUMENT
    assert lib.rpf_shard_info(None, None, None, None, None, None, 0) == N.ERR_INVALID_ARGUMENT
    lib.rpf_shard_destroy(None)   # a no-op, like free(NULL)


@gpu_only
def test_bench_two_ranks_rehearsal_and_rccl_fallback(tmp_path):
    """`bench.py --gpus 2` end to end as the driver launches it (torch.distributed.run, one process per rank), on ONE GPU:
    the process group over gloo, and — RPF_BENCH_TRY_RCCL — both ranks really calling rpf_shard_create with a broadcast
    unique id.  RCCL must refuse two ranks on one device, every rank must learn of it, and the run must complete on the
    torch.distributed exchange with the reason in the JSON line: strong scaling of THE 6000x4000 frame in two bands."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RPF_BENCH_TRY_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--preheat", "4",
           "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    r = util.bench_line(out.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["mode"] == "strong" and r["config"]["world_size"] == 2
    assert r["config"]["backend"] == "gloo" and "REHEARSAL" in r["config"]["workload"] and "ONE 6000x4000 frame" in r["config"]["workload"]
    assert [x["band"] for x in r["config"]["ranks"]] == [[0, 2000], [2000, 2000]]
    assert "fallback" in r["config"]["exchange"] and "ncclCommInitRank" in r["config"]["exchange"]
    assert r["value"] > 0 and abs(r["ms_per_step"] - 24.0 / r["value"] * 1e3) < 1e-2 * r["ms_per_step"]
    assert "secondary" not in r and "cpu_baseline" not in r
    assert r["config"]["schedule"] == "serial"


@gpu_only
@pytest.mark.parametrize("schedule", ["auto", "pingpong", "serial", "once"])
def test_bench_sharded_code_path_rehearsed_on_one_gpu(schedule):
    """The N > 1 branch of bench.py with the LIBRARY's communicator (what the driver's --gpus N run takes on a node),
    rehearsed in one process: rank 4 of 8's band of THE frame, the rank its own neighbour.  The default there is two
    independent step pipelines (pingpong: two shards with a thread each, two streams, rpf_shard_develop_async); the
    exchange-ahead schedule, exchange-then-launch and the editing-session mode are selectable."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "3", "--preheat", "5",
           "--rehearse-ranks", "8"] + {"serial": ["--schedule", "serial"], "once": ["--halo", "once"],
                                       "pingpong": ["--schedule", "pingpong"], "auto": []}[schedule]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    c = r["config"]
    # the default: the headline line is exchange-then-launch on one stream, the two-pipeline schedule runs after it and is
    # reported under secondary.pingpong
    assert c["schedule"] == {"once": "none (halo once)", "auto": "serial"}.get(schedule, schedule)
    if schedule == "auto":
        pp = r["secondary"]["pingpong"]
        assert "error" not in pp and pp["value"] > 0 and pp["ms_per_step"] > 0
    assert ("EDITING SESSION" in c["workload"]) == (schedule == "once")
    assert c["mode"] == "strong" and "REHEARSAL of rank 4 of 8" in c["workload"]
    assert c["ranks"][0]["band"] == [2000, 500] and c["ranks"][0]["neighbours"] == [0, 0]
    assert "one-rank" in c["exchange"] and r["n_gpus"] == 1
    assert r["value"] > 0 and abs(r["ms_per_step"] - 3.0 / r["value"] * 1e3) < 1e-2 * r["ms_per_step"]
    assert (schedule == "auto" or "secondary" not in r) and "cpu_baseline" not in r
