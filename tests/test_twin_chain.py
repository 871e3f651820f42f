"""The kernels' per-pixel arithmetic (rpf_chain.h) and the product's host-side parameter packing
(rpf_host.h: make_mask_params) compiled for the CPU, against the oracle: bit-exact."""
import numpy as np

import oracle as O
from tests import util


def _pixels(n, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.random((n, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)
    rgb[:2000] = rng.random((2000, 1), dtype=np.float32)  # greys: a, b ~ 0, hue ill-conditioned
    rgb[2000:2100] = 0
    rgb[2100:2200] = 1
    rgb[2200:2300] = np.float32([1, 0, 0])
    rgb[2300:2400] = np.float32([0, 0, 1])
    return rgb


def _check(masks_params, n=400_000, seed=5):
    rgb = _pixels(n, seed)
    img = np.concatenate([rgb, np.ones((n, 1), np.float32)], 1).reshape(1, n, 4)
    want = O.apply(img, masks_params).reshape(n, 4)
    got = util.twin_apply(rgb, masks_params)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_defaults():
    _check([util.params_D()])


def test_stress_set():
    p = util.params_S()
    p.vignette = 0  # the vignette is applied before the chain, by the kernel
    _check([p])


def test_general_curves():
    _check([util.params_G()])


def test_two_active_masks():
    s = util.params_S()
    s.vignette = 0
    _check([s, util.params_G()], n=200_000)


def test_tone_corner_values():
    for kw in (dict(exposure=10.0), dict(exposure=-10.0), dict(contrast=100), dict(contrast=-100),
               dict(shadow=100, highlight=-100), dict(black=100, white=-100), dict(black=-100, white=100),
               dict(wb_temperature=100, wb_tint=100), dict(wb_temperature=-100, wb_tint=-100)):
        _check([O.EditParameters(**kw)], n=50_000, seed=9)


def test_random_parameter_space():
    rng = np.random.default_rng(2026)
    for _ in range(30):
        n_masks = int(rng.integers(1, 4))
        _check([util.random_params(rng) for _ in range(n_masks)], n=20_000, seed=int(rng.integers(1 << 30)))
