"""16-bit PPM ingest (web-ts/core/image.ts:146-195 loadPpm): header grammar on the CPU through the C ABI and the
oracle, sample conversion bit-exact on the GPU."""
import numpy as np
import pytest

import oracle as O
from tests.util import configure_editor, params_D


def make_ppm(img16: np.ndarray, header: bytes | None = None) -> bytes:
    h, w, _ = img16.shape
    head = header if header is not None else f"P6\n{w} {h}\n65535\n".encode()
    return head + img16.astype(">u2").tobytes()


def test_f64_quotient_rounds_like_f32_division():
    # loadPpm divides in binary64 and stores into a Float32Array; the kernel divides in binary32
    k = np.arange(65536)
    assert np.array_equal((k.astype(np.float64) / 65535.0).astype(np.float32), k.astype(np.float32) / np.float32(65535.0))


def test_oracle_known_answers():
    img = np.array([[[0, 65535, 1], [0x1234, 0x00FF, 0xFF00]]], np.uint16)
    f = make_ppm(img)
    assert f[:16] == b"P6\n2 1\n65535\n\x00\x00\xff"
    assert O.ppm16_header(f) == (0, 2, 1, 13)
    out = O.ppm16_load(f)
    assert out.shape == (1, 2, 4)
    expect = np.array([[[0.0, 1.0, 1 / 65535, 1.0], [0x1234 / 65535, 0xFF / 65535, 0xFF00 / 65535, 1.0]]])
    assert np.array_equal(out, expect.astype(np.float32))


HEADER_CASES = [
    (b"P6\n3 2\n65535\n", 0),
    (b"  P6 \r\n3 2\r\n65535\r\n", 0),          # readLine trims
    (b"P5\n3 2\n65535\n", 1),                   # "PPM format must be P6"
    (b"P6\n3  2\n65535\n", 2),                  # split(" ") leaves an empty field
    (b"P6\n3\n2\n65535\n", 2),
    (b"P6\n0 2\n65535\n", 2),
    (b"P6\n3 2\n255\n", 3),                     # "Only 16bit PPM (max=65535) supported"
    (b"P6\n# c\n3 2\n65535\n", 2),              # no comment support in the reference
    (b"P6\n3 2\n65535", 4),                     # header not terminated
    (b"P6\n3 2", 4),
]


@pytest.mark.parametrize("head,status", HEADER_CASES)
def test_header_grammar_matches_oracle(head, status):
    import rawphotoforge_amd as R
    payload = bytes(range(36)) if status == 0 else b""
    assert O.ppm16_header(head + payload)[0] == status
    if status == 0:
        assert R.ppm16_parse_header(head + payload) == (3, 2, len(head))
        assert O.ppm16_header(head + payload) == (0, 3, 2, len(head))
        with pytest.raises(R.FailedToReadImage):       # one byte short
            R.ppm16_parse_header(head + payload[:-1])
        assert O.ppm16_header(head + payload[:-1])[0] == 4
    else:
        with pytest.raises(R.FailedToReadImage):
            R.ppm16_parse_header(head + payload)


def test_error_text_follows_reference():
    import rawphotoforge_amd as R
    with pytest.raises(R.FailedToReadImage, match="PPM format must be P6"):
        R.ppm16_parse_header(b"P3\n1 1\n65535\n" + bytes(6))
    with pytest.raises(R.FailedToReadImage, match=r"Only 16bit PPM \(max=65535\) supported"):
        R.ppm16_parse_header(b"P6\n1 1\n255\n" + bytes(6))


@pytest.mark.gpu
def test_gpu_ppm_editor_bit_exact(gpu):
    import rawphotoforge_amd as R
    rng = np.random.default_rng(77)
    for h, w in ((1, 1), (3, 5), (61, 127), (480, 640)):
        img = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
        img.reshape(-1)[:3] = (0, 65535, 1)
        f = make_ppm(img)                                  # header lengths vary -> payload offsets of both parities
        ed = R.PhotoEditor.new_from_ppm16(gpu, f)
        assert (ed.width, ed.height) == (w, h)
        want = O.ppm16_load(f)
        assert np.array_equal(ed.to_flat_vec_rgba().reshape(h, w, 4), want)
        # the editor then behaves like one built from the same pixels as f32
        p = params_D()
        configure_editor(ed, p)
        ed.apply_adjustments()
        ref = O.apply(want, [p])
        got = ed.to_flat_vec_rgba().reshape(h, w, 4)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_gpu_ppm_device_layer_and_errors(gpu):
    torch = pytest.importorskip("torch")
    import rawphotoforge_amd as R
    rng = np.random.default_rng(78)
    img = rng.integers(0, 65536, (33, 65, 3), dtype=np.uint16)
    payload = np.frombuffer(img.astype(">u2").tobytes(), np.uint8)
    d = torch.from_numpy(payload.copy()).cuda()
    out = torch.empty((33, 65, 4), dtype=torch.float32, device="cuda")
    R.dev_rgb16be_to_rgba_f32(gpu, d.data_ptr(), 33 * 65, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.ppm16_load(make_ppm(img)))
    with pytest.raises(R.InvalidArgument):
        R.dev_rgb16be_to_rgba_f32(gpu, d.data_ptr() + 1, 10, out.data_ptr(), 0)
    with pytest.raises(R.FailedToReadImage):
        R.PhotoEditor.new_from_ppm16(gpu, b"P6\n4 4\n65535\n" + bytes(10))
