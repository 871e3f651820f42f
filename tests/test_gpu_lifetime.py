"""Device memory comes back: editors of every kind — RGB, Bayer, tile, with more masks than one coverage word (their planes, the
parameter block in device memory, the band's linear intermediate), a vignette map, a fused hue table, shards with queued steps —
created, used and destroyed many times leave the device's free memory where it was."""
import gc

import numpy as np
import pytest

import rawphotoforge_amd as R
from rawphotoforge_amd import tiles
from tests import util

pytestmark = pytest.mark.gpu


def _cycle(gpu, torch, i):
    rng = np.random.default_rng(i)
    h, w = 96, 160
    img = rng.random((h, w, 3), dtype=np.float32)
    cfa = util.bayer_natural(h, w, seed=i)
    rp = util.raw_params()
    mask = (rng.random((h, w)) > 0.5).astype(np.float32)
    n_masks = 20 if i % 2 else 3
    eds = [R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w), R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp))]
    for ed in eds:
        for m in range(1, n_masks):
            ed.add_mask(f"m{m}", mask)
        util.configure_editor(ed, util.params_G())          # hue / saturation / lightness curves
        ed.set_vignette(-30)                                  # the falloff map
        ed.apply_adjustments()
        ed.rgba()
        ed.remove_mask("m1")
        ed.apply_adjustments()
        ed.to_rgb_array3_u8()
    one = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp))   # one mask with curves: the fused hue table
    util.configure_editor(one, util.params_G())
    one.apply_adjustments()
    one.rgba()
    pv = eds[1].create_preview(64)
    pv.apply_adjustments()
    pv.rgba()
    # a tile editor with many masks through the band entry point and a sharded step handed to the shard's thread
    band = np.zeros((h + 4, w), np.uint16)
    band[2:-2] = cfa
    d_band = torch.from_numpy(band.view(np.int16)).cuda()
    d_out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def configure(e):
        for m in range(1, n_masks):
            e.add_mask(f"m{m}", mask)
        util.configure_editor(e, util.params_S())
    shard = R.Shard(gpu, None, 0, 1)
    shard.set_neighbours(0, 0)
    sd = tiles.ShardedDeveloper(gpu, w, h + 4, 2, h, util.to_raw(rp), configure, shard=shard)
    sd.ed.dev_develop(d_band.data_ptr(), util.to_raw(rp), d_out.data_ptr(), R.OUT_RGBA_F32, s)
    for _ in range(3):
        sd.step_async(d_band, d_out, stream=s)
    shard.drain(s)
    torch.cuda.synchronize()
    for e in eds + [one, pv, sd.ed]:
        e.close()
    shard.close()
    del d_band, d_out


def test_device_memory_returns_after_many_editor_lifetimes(gpu):
    torch = pytest.importorskip("torch")
    for i in range(3):          # allocator pools, lazily built tables, the stand-alone streams' first use
        _cycle(gpu, torch, i)
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        _cycle(gpu, torch, 100 + i)
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    lost = free0 - free1
    # one leaked image-sized buffer per cycle (245 KiB at this size: a band's linear intermediate, a mask plane set, an image) is
    # ~10 MiB after 40 cycles; a leaked 1-KiB block could hide in the runtime's own pools — this test is about the big ones
    assert lost < 2 << 20, f"{lost / 2**20:.1f} MiB of device memory did not come back after 40 cycles"   # measured: 0
