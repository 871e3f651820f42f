"""Packed-RAW unpack (integer stage): oracle known answers on the CPU, kernel bit-exact on the GPU."""
import numpy as np
import pytest

import oracle as O


def test_oracle_known_answers():
    # 12-bit: AB CD EF -> ABC, DEF;  10-bit: FF C0 00 00 01 -> 3FF, 000, 000, 001;  14-bit; 16-bit big-endian
    assert O.unpack_bayer(np.array([[0xAB, 0xCD, 0xEF, 0]], np.uint8), 2, 12).tolist() == [[0xABC, 0xDEF]]
    assert O.unpack_bayer(np.array([[0xFF, 0xC0, 0x00, 0x00, 0x01, 0, 0, 0]], np.uint8), 4, 10).tolist() == [[1023, 0, 0, 1]]
    assert O.unpack_bayer(np.array([[0xFF, 0xFC, 0x00, 0x04, 0, 0, 0, 0]], np.uint8), 2, 14).tolist() == [[16383, 0]]
    assert O.unpack_bayer(np.array([[0x12, 0x34, 0xAB, 0xCD]], np.uint8), 2, 16).tolist() == [[0x1234, 0xABCD]]


@pytest.mark.parametrize("bits", [10, 12, 14, 16])
def test_oracle_round_trip(bits):
    rng = np.random.default_rng(bits)
    for w in (1, 15, 16, 17, 100, 257):
        a = rng.integers(0, 1 << bits, (7, w), dtype=np.uint16)
        p = O.pack_bayer(a, bits)
        assert p.shape[1] % 4 == 0 and np.array_equal(O.unpack_bayer(p, w, bits), a)
        # numpy cross-check through a bit matrix
        bitsm = np.unpackbits(p, axis=1)[:, :w * bits].reshape(7, w, bits)
        assert np.array_equal((bitsm * (1 << np.arange(bits - 1, -1, -1))).sum(-1), a)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [10, 12, 14, 16])
def test_gpu_unpack_bit_exact(gpu, bits):
    torch = pytest.importorskip("torch")
    import rawphotoforge_amd as R
    rng = np.random.default_rng(100 + bits)
    for h, w, extra in ((9, 64, 0), (5, 1000, 0), (4, 37, 0), (3, 16, 4), (6, 6000, 0), (2, 15, 0), (3, 24, 3)):
        a = rng.integers(0, 1 << bits, (h, w), dtype=np.uint16)
        stride = O.packed_row_stride(w, bits) + extra          # `extra` = 3 makes rows unaligned (slow path)
        p = O.pack_bayer(a, bits, stride)
        d = torch.from_numpy(p).cuda()
        out = torch.zeros((h, w), dtype=torch.int16, device="cuda")
        R.dev_unpack_bayer(gpu, d.data_ptr(), stride, w, h, bits, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint16), a), (h, w, extra)


@pytest.mark.gpu
def test_gpu_unpack_into_band_buffer_then_develop(gpu):
    """The integer stage feeding the float stages: 12-bit packed -> rows 2.. of a band buffer -> fused kernel."""
    torch = pytest.importorskip("torch")
    import rawphotoforge_amd as R
    from tests import util
    h, w = 64, 128
    cfa = util.bayer_natural(h, w) & 0x0FFF
    rp = O.RawParams(black=(256,) * 4, white=4095)
    p = O.pack_bayer(cfa, 12)
    band = torch.zeros((h + 4, w), dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    R.dev_unpack_bayer(gpu, torch.from_numpy(p).cuda().data_ptr(), p.shape[1], w, h, 12, band[2:].data_ptr(), s)
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
    ed.dev_develop(band.data_ptr(), util.to_raw(rp), out.data_ptr(), R.OUT_RGBA_F32, s)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.develop(cfa, rp, [O.EditParameters()]))


@pytest.mark.gpu
def test_gpu_unpack_rejects_bad_arguments(gpu):
    import rawphotoforge_amd as R
    with pytest.raises(R.PhotoEditorError):
        R.dev_unpack_bayer(gpu, 1, 64, 16, 1, 11, 1)       # unsupported bit depth
    with pytest.raises(R.InvalidArgument):
        R.dev_unpack_bayer(gpu, 1, 8, 16, 1, 12, 1)        # stride shorter than a row
