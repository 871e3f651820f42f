"""Preview pyramid (web/main.ts:652-695, 958-1026): oracle restatement on the CPU, device resize on the GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from rawphotoforge_amd import _native as N


def _js_resize(src, target):
    """numpy float64 transcription of resizeFloat32RGBLongEdge."""
    sh, sw, _ = src.shape
    if sw >= sh:
        dw, dh = target, int(np.floor(sh * (target / sw) + 0.5))
    else:
        dh, dw = target, int(np.floor(sw * (target / sh) + 0.5))
    s = src.astype(np.float64)
    ys = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    y0 = np.maximum(np.floor(ys), 0).astype(int); y1 = np.minimum(y0 + 1, sh - 1); ty = (ys - y0)[:, None, None]
    xs = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    x0 = np.maximum(np.floor(xs), 0).astype(int); x1 = np.minimum(x0 + 1, sw - 1); tx = (xs - x0)[None, :, None]
    cx0 = s[y0][:, x0] * (1.0 - tx) + s[y0][:, x1] * tx
    cx1 = s[y1][:, x0] * (1.0 - tx) + s[y1][:, x1] * tx
    return (cx0 * (1.0 - ty) + cx1 * ty).astype(np.float32)


CASES = [(40, 60, 16), (61, 37, 20), (10, 10, 25), (400, 267, 128), (33, 5, 7), (2, 90, 45)]


def test_oracle_resize_equals_float64_transcription():
    rng = np.random.default_rng(0)
    for h, w, t in CASES:
        img = rng.random((h, w, 3), dtype=np.float32)
        assert np.array_equal(O.resize_long_edge(img, t), _js_resize(img, t)), (h, w, t)


def test_preview_size_abi_equals_oracle():
    for sw, sh, t in [(6000, 4000, 1280), (6000, 4000, 400), (4000, 6000, 1280), (1001, 999, 333), (3, 7, 2), (5, 5, 5)]:
        dw, dh = C.c_uint32(), C.c_uint32()
        assert N.lib().rpf_preview_size(sw, sh, t, C.byref(dw), C.byref(dh)) == N.OK
        assert (dw.value, dh.value) == O.preview_size(sw, sh, t)
    assert O.preview_size(6000, 4000, 1280) == (1280, 853)


@pytest.mark.gpu
def test_gpu_preview_editors_bit_exact(gpu):
    import rawphotoforge_amd as R
    from tests import util
    rng = np.random.default_rng(5)
    for h, w, t in CASES:
        img = rng.random((h, w, 3), dtype=np.float32)
        full = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
        small = full.create_preview(t)
        want = O.resize_long_edge(img, t)
        assert (small.height, small.width) == want.shape[:2]
        assert np.array_equal(small.to_flat_vec_rgb().reshape(want.shape), want)
        p = util.params_S()
        util.configure_editor(small, p)
        small.apply_adjustments()
        rgba = np.concatenate([want, np.ones(want.shape[:2] + (1,), np.float32)], -1)
        assert np.array_equal(small.rgba(), O.apply(rgba, [p]))
    # the pyramid of a RAW frame is built from its demosaiced linear image
    cfa = util.bayer_natural(120, 200)
    rp = util.raw_params()
    mid = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp)).create_preview(64)
    want = O.resize_long_edge(O.demosaic(cfa, rp)[..., :3], 64)
    assert np.array_equal(mid.to_flat_vec_rgb().reshape(want.shape), want)
    with pytest.raises(R.GpuComputeError):
        R.PhotoEditor.for_tile(gpu, 64, 64, 0, 64).create_preview(16)
