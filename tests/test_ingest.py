"""Decoded 8- / 16-bit RGB ingest (rust/photo-editor/src/image.rs:412-447 behind the decoder): oracle known answers on
the CPU, the device conversion bit-exact on the GPU, and an editor created from integer samples against one created from
the converted floats."""
import numpy as np
import pytest

import oracle as O


def test_oracle_quotients_are_single_binary32_divisions():
    v8 = np.arange(256, dtype=np.uint8)
    img = np.stack([v8, v8[::-1], np.roll(v8, 7)], -1).reshape(16, 16, 3)
    got = O.rgb_int_to_rgba(img)
    assert np.array_equal(got[..., :3], img.astype(np.float32) / np.float32(255.0)) and np.all(got[..., 3] == 1.0)
    assert got[0, 0, 0] == 0.0 and got.reshape(-1, 4)[255, 0] == 1.0
    v16 = np.arange(65536, dtype=np.uint16)
    img = np.stack([v16, v16[::-1], np.roll(v16, 4099)], -1).reshape(256, 256, 3)
    got = O.rgb_int_to_rgba(img)
    assert np.array_equal(got[..., :3], img.astype(np.float32) / np.float32(65535.0))
    # the JavaScript loader's v / 65535 (binary64 quotient stored as binary32, image.ts:175-189) lands on the same floats
    be = img.astype(">u2").tobytes()
    ppm = b"P6\n256 256\n65535\n" + be
    assert np.array_equal(O.ppm16_load(ppm), got)


def test_oracle_table_replaces_the_quotient():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    lut = rng.standard_normal(256).astype(np.float32)
    got = O.rgb_int_to_rgba(img, lut)
    assert np.array_equal(got[..., :3], lut[img]) and np.all(got[..., 3] == 1.0)
    img = rng.integers(0, 65536, (5, 7, 3), dtype=np.uint16)
    lut = rng.random(65536, dtype=np.float32)
    assert np.array_equal(O.rgb_int_to_rgba(img, lut)[..., :3], lut[img])
    with pytest.raises(ValueError):
        O.rgb_int_to_rgba(img, lut[:256])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_gpu_rgb_int_bit_exact(gpu, dtype):
    torch = pytest.importorskip("torch")
    import rawphotoforge_amd as R
    bits = 8 * np.dtype(dtype).itemsize
    rng = np.random.default_rng(40 + bits)
    s = torch.cuda.current_stream().cuda_stream
    for h, w in ((1, 1), (3, 5), (64, 64), (17, 1000), (401, 1023)):
        img = rng.integers(0, 1 << bits, (h, w, 3), dtype=dtype)
        if h * w >= 1 << bits:     # every sample value at least once
            img.reshape(-1)[:1 << bits] = np.arange(1 << bits, dtype=dtype)
        d = torch.from_numpy(img.view(np.int16) if bits == 16 else img).cuda()
        for lut in (None, rng.standard_normal(1 << bits).astype(np.float32)):
            out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
            dl = torch.from_numpy(lut).cuda() if lut is not None else None
            R.dev_rgb_int_to_rgba_f32(gpu, d.data_ptr(), bits, h * w, out.data_ptr(), dl.data_ptr() if dl is not None else 0, s)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), O.rgb_int_to_rgba(img, lut).view(np.uint32)), (h, w)
    # an odd device address for 16-bit samples is refused, not dereferenced
    with pytest.raises(R.InvalidArgument):
        R.dev_rgb_int_to_rgba_f32(gpu, 1, 16, 4, 16)
    with pytest.raises(R.InvalidArgument):
        R.dev_rgb_int_to_rgba_f32(gpu, 16, 12, 4, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_editor_from_integer_samples(gpu, dtype):
    """read_image's tail on the device == converting on the host and calling new_from_rgb_f32 (image.rs:443-445)."""
    import rawphotoforge_amd as R
    from tests import util
    bits = 8 * np.dtype(dtype).itemsize
    rng = np.random.default_rng(50 + bits)
    h, w = 37, 53
    img = rng.integers(0, 1 << bits, (h, w, 3), dtype=dtype)
    # a stand-in for the sRGB -> linear step (the caller's table): monotone, inside [0, 1]
    lut = (np.arange(1 << bits, dtype=np.float64) / ((1 << bits) - 1)) ** 2.2
    for table in (None, lut.astype(np.float32)):
        rgba = O.rgb_int_to_rgba(img, table)
        ed = R.PhotoEditor.new_from_rgb_int(gpu, img, table)
        assert (ed.width, ed.height) == (w, h)
        p = util.params_S()
        util.configure_editor(ed, p)
        ed.apply_adjustments()
        want = O.apply(rgba, [p])
        assert np.array_equal(np.asarray(ed.rgba(), np.float32).reshape(want.shape).view(np.uint32), want.view(np.uint32))
    with pytest.raises(R.FailedToReadImage):
        R.PhotoEditor.new_from_rgb_int(gpu, np.zeros((0, 4, 3), dtype))
    with pytest.raises(R.FailedToReadImage):
        R.PhotoEditor.new_from_rgb_int(gpu, np.zeros((4, 4, 3), np.float32))
    with pytest.raises(R.InvalidArgument):
        R.PhotoEditor.new_from_rgb_int(gpu, img, np.zeros(7, np.float32))
