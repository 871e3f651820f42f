"""The RAW front half (this build's own definition — the reference takes it from rawler / rawpy) restated a second
time from that text in numpy binary32 array arithmetic — np.pad(mode="reflect") for the reflect-101 border, whole-plane sums,
np.where by site — against the oracle's per-pixel C loop: bit-exact on every CFA layout, odd sizes and the smallest frame."""
import numpy as np
import pytest

import oracle as O
from tests import util

F = np.float32
PHASE = {O.CFA_RGGB: (0, 0), O.CFA_BGGR: (1, 1), O.CFA_GRBG: (1, 0), O.CFA_GBRG: (0, 1)}   # (xo, yo): cell = ((x + xo) & 1, (y + yo) & 1)


def demosaic_numpy(cfa, rp):
    h, w = cfa.shape
    xo, yo = PHASE[rp.cfa_pattern]
    yy, xx = np.mgrid[0:h, 0:w]
    cx, cy = (xx + xo) & 1, (yy + yo) & 1
    cell = cx + 2 * cy                                   # R, G (red row), G (blue row), B
    black = np.asarray(rp.black, F)[cell]
    rng_ = F(rp.white) - black                           # white - black[c]
    scale = np.asarray(rp.wb, F)[cell] / rng_            # wb[c] / (white - black[c])
    s = np.minimum(np.maximum(cfa.astype(F) - black, F(0)), rng_) * scale
    p = np.pad(s, 2, mode="reflect")                     # reflect-101: index -1 -> 1, -2 -> 2 (keeps the CFA phase)

    def at(dx, dy):
        return p[2 + dy:2 + dy + h, 2 + dx:2 + dx + w]
    c = at(0, 0)
    s1 = at(0, -1) + at(0, 1)
    e1 = at(-1, 0) + at(1, 0)
    d = (at(-1, -1) + at(1, -1)) + (at(-1, 1) + at(1, 1))
    v2 = at(0, -2) + at(0, 2)
    h2 = at(-2, 0) + at(2, 0)
    cross = ((F(4) * c + F(2) * (s1 + e1)) - (v2 + h2)) / F(8)
    diag = ((F(6) * c + F(2) * d) - F(1.5) * (v2 + h2)) / F(8)
    horiz = ((((F(5) * c + F(4) * e1) - d) - h2) + F(0.5) * v2) / F(8)
    vert = ((((F(5) * c + F(4) * s1) - d) - v2) + F(0.5) * h2) / F(8)
    rb_site, r_site, red_row = cx == cy, (cx == 0) & (cy == 0), cy == 0
    R = np.where(rb_site, np.where(r_site, c, diag), np.where(red_row, horiz, vert))
    G = np.where(rb_site, cross, c)
    B = np.where(rb_site, np.where(r_site, diag, c), np.where(red_row, vert, horiz))
    m = np.asarray(rp.cam2rgb, F)
    out = [m[3 * k] * R + m[3 * k + 1] * G + m[3 * k + 2] * B for k in range(3)]
    return np.stack(out + [np.ones((h, w), F)], -1)


@pytest.mark.parametrize("pattern", [O.CFA_RGGB, O.CFA_BGGR, O.CFA_GRBG, O.CFA_GBRG])
def test_numpy_front_half_equals_oracle(pattern):
    rng = np.random.default_rng(31 + pattern)
    for h, w in ((3, 3), (4, 5), (37, 64), (128, 131)):
        cfa = rng.integers(0, 16384, (h, w), dtype=np.uint16)
        cfa[0, 0], cfa[-1, -1] = 0, 16383                      # below the black level / at the white level: both clamps
        rp = O.RawParams(cfa_pattern=pattern, black=(500.0, 512.0, 520.0, 508.0), white=16000.0,
                         wb=(2.1, 1.0, 1.03, 1.6), cam2rgb=(1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5))
        want = O.demosaic(cfa, rp)
        got = demosaic_numpy(cfa, rp)
        bad = np.argwhere(want.view(np.uint32) != got.view(np.uint32))
        assert bad.size == 0, (pattern, h, w, len(bad), bad[:4])


def test_numpy_front_half_on_the_bench_inputs():
    cfa = util.bayer_natural(96, 160, seed=5)
    rp = util.raw_params()
    assert np.array_equal(O.demosaic(cfa, rp).view(np.uint32), demosaic_numpy(cfa, rp).view(np.uint32))
