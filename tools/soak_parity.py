#!/usr/bin/env python3
"""Soak: random edit states (tests/util.random_params: every combination of the identity- / constant-curve fast paths), random
frame sizes, CFA patterns, mask counts and row bands, for a given number of seconds, the GPU against the oracle bit for bit.
The same comparison as tests/test_gpu_parity.py::test_random_parameter_space_on_gpu, longer and wider — test
infrastructure (it drives the oracle), run by hand on the GPU box:  python tools/soak_parity.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rawphotoforge_amd as R
import oracle as O
from tests import util

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
gpu = R.GpuProcessor(0)
t_end = time.time() + seconds
n = bad = fast_n = fast_bad = 0
kinds = {}


def smooth_oklch(params):
    """Where the comparison with the oracle means something for the fast precision: every hue / saturation curve of every mask
    is identity, constant or PCHIP-smooth, and every per-hue LIGHTNESS curve is constant.  A random-array curve makes the
    REFERENCE discontinuous at every hue index; and the hue of an achromatic pixel (a, b = rounding noise: greys, clipped
    highlights) is decided by that noise — harmless for its chroma (~0 times any gain), but a lightness curve that varies with
    the hue then hands such a pixel whatever gain the noise picks, in the reference as in any other evaluation of the chain."""
    for p in params:
        for c in (p.hue_tone_curve, p.saturation_tone_curve):
            if np.abs(np.diff(np.asarray(c, np.int64))).max() > 64:
                return False
        lc = np.asarray(p.lightness_tone_curve)
        if (lc != lc[0]).any():
            return False
    return True


def check_fast(got_fast, want, what, params):
    """RPF_PRECISION_FAST on the same case: finite, alpha 1, and — with smooth OKLCH curves — all but a few per cent of the
    pixels (those where a hue index steps) within 1e-5 of the oracle."""
    global fast_n, fast_bad
    g = np.asarray(got_fast, np.float32).reshape(want.shape)
    fast_n += 1
    far = (np.abs(g[..., :3].astype(np.float64) - want[..., :3]).max(-1) > 1e-5).mean()
    if not smooth_oklch(params):
        far = 0.0
    if not np.isfinite(g).all() or not np.array_equal(g[..., 3], want[..., 3]) or far > 0.04 + 20.0 / g[..., 0].size:
        fast_bad += 1
        print(f"FAST out of bounds, case {n} {what}: {far:.4f} of the pixels beyond 1e-5, finite {np.isfinite(g).all()}", flush=True)
while time.time() < t_end:
    h, w = int(rng.integers(3, 140)), int(rng.integers(3, 420))
    n_masks = int(rng.choice([1, 1, 1, 2, 3])) if rng.random() > 0.04 else int(rng.integers(15, 40))   # now and then: past one coverage word
    params = [util.random_params(rng) for _ in range(n_masks)]
    params[0].vignette = 0 if rng.random() < 0.4 else int(rng.integers(-100, 101))
    masks = np.ones((n_masks, h, w), np.float32)
    raw_masks = []
    for m in range(1, n_masks):
        mm = rng.random((h, w)).astype(np.float32) - np.float32(0.3)
        raw_masks.append(mm)
        masks[m] = O.binarize_mask(mm)
    bayer = rng.random() < 0.7
    if bayer and n_masks == 1 and h >= 8 and rng.random() < 0.4:
        # a row band of the frame through the device-pointer layer (for_tile editor, halo rows from the frame itself)
        import torch
        from rawphotoforge_amd import tiles
        cfa = util.bayer_natural(h, w, seed=int(rng.integers(1 << 30)))
        rp = O.RawParams(cfa_pattern=int(rng.integers(0, 4)))
        y0 = int(rng.integers(0, h - 3))
        rows = int(rng.integers(3, h - y0 + 1))
        pad = np.zeros((rows + 4, w), np.uint16)
        lo, hi = max(y0 - 2, 0), min(y0 + rows + 2, h)
        pad[lo - (y0 - 2):hi - (y0 - 2)] = cfa[lo:hi]
        band = torch.from_numpy(pad.view(np.int16)).cuda()
        out = torch.full((rows, w, 4), -7.0, dtype=torch.float32, device="cuda")
        ed = R.PhotoEditor.for_tile(gpu, w, h, y0, rows)
        util.configure_editor(ed, params[0])
        ed.dev_develop(band.data_ptr(), util.to_raw(rp), out.data_ptr(), R.OUT_RGBA_F32, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = O.develop(cfa, rp, params, y0=y0, rows=rows)
        d = util.ulp_diff(out.cpu().numpy().reshape(want.shape), want)
        n += 1
        key = ("band", 1, params[0].vignette != 0)
        kinds[key] = kinds.get(key, 0) + 1
        if d.max() != 0:
            bad += 1
            print(f"MISMATCH case {n}: band {y0}+{rows} of {h}x{w} {key}: {int((d > 0).sum())} values, max {int(d.max())} ulp", flush=True)
        ed.set_precision(R.PRECISION_FAST)
        ed.dev_develop(band.data_ptr(), util.to_raw(rp), out.data_ptr(), R.OUT_RGBA_F32, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        check_fast(out.cpu().numpy(), want, key, params)
        continue
    if bayer:
        cfa = util.bayer_natural(h, w, seed=int(rng.integers(1 << 30)))
        rp = O.RawParams(cfa_pattern=int(rng.integers(0, 4)))
        ed = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp))
        want = O.develop(cfa, rp, params, masks=masks)
    else:
        img = (rng.random((h, w, 3), dtype=np.float32) * np.float32(1.2) - np.float32(0.05))
        ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
        want = O.apply(np.concatenate([img, np.ones((h, w, 1), np.float32)], -1), params, masks=masks)
    for m in range(1, n_masks):
        ed.add_mask(f"m{m}", raw_masks[m - 1])
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    ed.apply_adjustments()
    got = np.asarray(ed.rgba(), np.float32).reshape(want.shape)
    d = util.ulp_diff(got, want)
    n += 1
    key = ("bayer" if bayer else "rgb", n_masks, params[0].vignette != 0)
    kinds[key] = kinds.get(key, 0) + 1
    if d.max() != 0:
        bad += 1
        print(f"MISMATCH case {n}: {h}x{w} {key}: {int((d > 0).sum())} values, max {int(d.max())} ulp", flush=True)
    ed.set_precision(R.PRECISION_FAST)
    ed.apply_adjustments()
    check_fast(ed.rgba(), want, key, params)
    if n % 50 == 0:
        print(f"{n} cases, {bad} mismatches; fast precision: {fast_n} cases, {fast_bad} out of bounds", flush=True)
print(f"soak done: {n} cases in {seconds:.0f} s, {bad} mismatches; fast precision: {fast_n} cases, {fast_bad} out of bounds; "
      f"by (source, masks, vignette): {sorted(kinds.items())}")
sys.exit(1 if (bad or fast_bad) else 0)
