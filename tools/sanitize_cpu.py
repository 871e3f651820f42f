"""Driver of tools/sanitize_cpu.sh: runs the oracle, the CPU twin of the device arithmetic and the host state
machine (built with -fsanitize=address,undefined) over the shapes / parameter combinations the tests use."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.environ["RPF_ASAN_DIR"]
import oracle as O  # noqa: E402

O.build = lambda force=False: os.path.join(D, "librpf_oracle_asan.so")
from tests import util  # noqa: E402

util.build_twin = lambda force=False: os.path.join(D, "librpf_twin_asan.so")
rng = np.random.default_rng(0)
for _ in range(6):
    ps = [util.random_params(rng) for _ in range(int(rng.integers(1, 4)))]
    rgb = rng.random((5000, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)
    img = np.concatenate([rgb, np.ones((5000, 1), np.float32)], 1).reshape(1, 5000, 4)
    assert np.array_equal(util.twin_apply(rgb, ps).view(np.uint32), O.apply(img, ps).reshape(5000, 4).view(np.uint32))
# round 4: coverage planes past one word (edit_chain's WideMasks; plane counts on both sides of a boundary, a partial last plane)
# and the binary32-only chain in both of its hue forms
for n_masks in (2, 16, 17, 33):
    ps = [util.random_params(rng, trivial_prob=0.4) for _ in range(n_masks)]
    for p_ in ps:
        p_.vignette = 0
        p_.lens_distortion = 0
    npx = 777
    rgb = rng.random((npx, 3), dtype=np.float32)
    masks = np.ones((n_masks, 1, npx), np.float32)
    for m in range(1, n_masks):
        masks[m, 0] = (rng.random(npx) > 0.5).astype(np.float32)
    img = np.concatenate([rgb, np.ones((npx, 1), np.float32)], 1).reshape(1, npx, 4)
    assert np.array_equal(util.twin_apply_masked(rgb, ps, masks).view(np.uint32), O.apply(img, ps, masks=masks).reshape(npx, 4).view(np.uint32))
for fast in (True, "builtins"):
    util.twin_apply(rng.random((3000, 3), dtype=np.float32), [util.params_S(), util.params_G()], fast=fast)
    util.twin_apply(rng.random((3000, 3), dtype=np.float32), [util.params_G()], fast=fast)
for shp in ((3, 3), (5, 8), (33, 47)):
    for pat in range(4):
        O.develop(util.bayer_uniform(*shp), O.RawParams(cfa_pattern=pat), [util.params_S()])
try:
    O.pchip_interpolate([0, 5, 5], [0, 1, 2], [1])
except O.PchipError:
    pass
for bits in (10, 12, 14, 16):
    a = rng.integers(0, 1 << bits, (3, 37), dtype=np.uint16)
    assert np.array_equal(O.unpack_bayer(O.pack_bayer(a, bits), 37, bits), a)
    for order in (O.PACK_LSB_FIRST, O.PACK_MIPI):   # ragged rows: the last CSI-2 group / the last byte are partial
        if order == O.PACK_MIPI and bits == 16:
            continue
        for w in (1, 3, 37):
            assert np.array_equal(O.unpack_bayer(O.pack_bayer(a[:, :w], bits, order=order), w, bits, order=order), a[:, :w])
for dt in (np.uint8, np.uint16):    # decoded-image ingest, with and without a table
    img = rng.integers(0, np.iinfo(dt).max + 1, (5, 7, 3), dtype=dt)
    O.rgb_int_to_rgba(img)
    O.rgb_int_to_rgba(img, rng.random(np.iinfo(dt).max + 1, dtype=np.float32))
O.resize_long_edge(rng.random((33, 5, 3), dtype=np.float32), 7)
O.legacy_tone_lut(0.5, 20, 30, -20, 5, -5)
O.legacy_apply_gamma(rng.random(1000, dtype=np.float32), (2.222, 4.5))
# later additions: hue index (table gather), legacy flavour, PPM header grammar on hostile input
tw, orc = util.twin(), O.lib()
n = 200_000
y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 2, n)).astype(np.float32)
x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 2, n)).astype(np.float32)
y[:8] = [0, -0.0, 0, -0.0, 1e-45, -1e-45, 3e38, -3e38]
x[:8] = [0, 0, -0.0, -0.0, 1, -1, 3e38, 1e-45]
a, b = np.empty(n, np.uint32), np.empty(n, np.uint32)
tw.twin_hue_index_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(n), a.ctypes.data_as(C.c_void_p))
orc.orc_hue_index_many(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_size_t(n), b.ctypes.data_as(C.c_void_p))
assert int((a != b).sum()) <= 1          # the documented (-3.5 * 2^-149, 0) corner is in the list above
for v in (0.0, 0.04045, 0.5, 1.0, 1.7):
    tw.twin_srgb_to_linear(v), tw.twin_pow_4(v)
img = rng.random((9, 11, 4), dtype=np.float32)
O.apply(img, [O.EditParameters(vignette=-40, lens_distortion=30)], legacy="encoded")
for blob in (b"", b"P6", b"P6\n", b"P6\n3 2\n65535\n", b"P6\n99999999999 2\n65535\n", b"P6\n3 2\n65535\n" + bytes(36),
             b"\n\n\n\n", b"P6\n" + b"9" * 200 + b"\n1\n"):
    O.ppm16_header(blob)
    arr = np.frombuffer(blob + b"\0", np.uint8)   # +1: a non-empty buffer to point at; len excludes it
    w_, h_, o_ = C.c_uint32(), C.c_uint32(), C.c_size_t()
    rc = tw.twin_ppm_header(arr.ctypes.data_as(C.c_void_p), C.c_size_t(len(blob)), C.byref(w_), C.byref(h_), C.byref(o_))
    assert (rc == 0) == (O.ppm16_header(blob)[0] == 0)
assert util.twin().twin_state_exercise() == 0
print("sanitizer run complete: no ASan / UBSan report")
