"""The operator state machine behind the C ABI (PhotoEditor setters, curves, masks, errors) on a
headless editor — no GPU needed.  Expected behaviour is the reference's: lib.rs:227-505."""
import numpy as np
import pytest

import oracle as O
import rawphotoforge_amd as R
from tests import util


@pytest.fixture
def ed():
    return R.PhotoEditor.headless(6, 4)


def test_defaults_match_reference(ed):
    g = ed.gpu_params()
    assert (g.r_gain, g.g_gain, g.b_gain) == (1.0, 1.0, 1.0)
    assert (g.exposure, g.contrast, g.shadow, g.highlight, g.black, g.white, g.vignette, g.lens_distortion) == (0,) * 8
    i = np.arange(65536, dtype=np.int32)
    assert np.array_equal(ed.curve(R.CURVE_BRIGHTNESS), i) and np.array_equal(ed.curve(R.CURVE_HUE), i)
    assert (ed.curve(R.CURVE_SATURATION) == 32767).all() and (ed.curve(R.CURVE_LIGHTNESS) == 32767).all()
    assert ed.num_masks == 1 and (ed.width, ed.height) == (6, 4)


def test_setters_clamp_like_the_reference(ed):
    ed.set_whitebalance(250, -250)
    ed.set_tone(20.0, 500, -500, 101, -101, 7)
    ed.set_vignette(1000)
    ed.set_lens_distortion_correction(-1000)
    want = O.convert_params(O.EditParameters(exposure=10.0, contrast=100, shadow=-100, highlight=100, black=-100,
                                             white=7, wb_temperature=100, wb_tint=-100, vignette=100,
                                             lens_distortion=-100))
    got = ed.gpu_params()
    for f, _ in got._fields_:
        assert getattr(got, f) == getattr(want, f), f
    ed.set_tone(-20.0, 0, 0, 0, 0, 0)
    assert ed.gpu_params().exposure == -10.0


def test_param_conversion_bit_exact_with_oracle(ed):
    rng = np.random.default_rng(0)
    for _ in range(200):
        v = rng.integers(-100, 101, 9)
        e = float(np.float32(rng.uniform(-10, 10)))
        ed.set_whitebalance(int(v[0]), int(v[1]))
        ed.set_tone(e, *[int(x) for x in v[2:7]])
        ed.set_vignette(int(v[7]))
        ed.set_lens_distortion_correction(int(v[8]))
        want = O.convert_params(O.EditParameters(e, *[int(x) for x in v[2:7]], int(v[0]), int(v[1]), int(v[7]),
                                                 int(v[8])))
        got = ed.gpu_params()
        for f, _ in got._fields_:
            assert getattr(got, f) == getattr(want, f), f


def test_curve_from_control_points_equals_oracle_pchip(ed):
    cases = [([0, 65535], [0, 65535]), ([0, 65535], [32767, 32767]),
             ([0, 16384, 49152, 65535], [0, 20000, 52000, 65535]),
             ([0, 100, 30000, 30001, 65535], [65535, 0, 70000, -5, 100]),   # overshoot -> clamp(0, 65535)
             ([10000, 50000], [5, 60000])]                                  # clamp outside the range
    setters = [ed.set_brightness_tone_curve, ed.set_oklch_hue_curve, ed.set_oklch_saturation_curve,
               ed.set_oklch_lightness_curve]
    for k, (xs, ys) in enumerate(cases):
        want = O.curve_from_control_points(xs, ys)
        setters[k % 4](control_points_x=xs, control_points_y=ys)
        assert np.array_equal(ed.curve(k % 4), want)
        assert want.min() >= 0 and want.max() <= 65535


def test_full_curve_is_stored_unclamped(ed):
    # lib.rs:321: a supplied curve is taken as is; the clamp happens at fetch (wgsl:188)
    c = np.arange(65536, dtype=np.int32) * 2 - 100
    ed.set_brightness_tone_curve(curve=c)
    assert np.array_equal(ed.curve(R.CURVE_BRIGHTNESS), c)


def test_pchip_abi_equals_oracle():
    rng = np.random.default_rng(1)
    for _ in range(50):
        n = int(rng.integers(2, 12))
        xs = np.sort(rng.choice(65536, n, replace=False)).astype(np.int32)
        ys = rng.integers(-1000, 70000, n).astype(np.int32)
        ev = rng.integers(-10, 65600, 500).astype(np.int32)
        assert np.array_equal(R.pchip_interpolate(xs, ys, ev), O.pchip_interpolate(xs, ys, ev))


def test_curve_errors_in_reference_order(ed):
    with pytest.raises(R.MissingCurveOrControlPoints):
        ed.set_oklch_hue_curve()
    with pytest.raises(R.MissingCurveOrControlPoints):
        ed.set_oklch_hue_curve(control_points_y=[1, 2])
    with pytest.raises(R.MissingControlPoints):
        ed.set_oklch_hue_curve(control_points_x=[0, 1])
    with pytest.raises(R.MismatchedLengths):
        ed.set_oklch_hue_curve(control_points_x=[0, 1], control_points_y=[0])
    with pytest.raises(R.EmptyControlPoints):
        ed.set_oklch_hue_curve(control_points_x=[], control_points_y=[])
    with pytest.raises(R.NotEnoughPoints):
        ed.set_oklch_hue_curve(control_points_x=[5], control_points_y=[5])
    with pytest.raises(R.NotStrictlyIncreasing):
        ed.set_oklch_hue_curve(control_points_x=[0, 5, 5], control_points_y=[0, 1, 2])
    with pytest.raises(R.InvalidCurveLength):
        ed.set_oklch_hue_curve(curve=[1, 2, 3])
    # a failed setter leaves the state untouched
    assert np.array_equal(ed.curve(R.CURVE_HUE), np.arange(65536))
    # validation precedes the mask lookup (lib.rs:314-341)
    with pytest.raises(R.InvalidCurveLength):
        ed.set_oklch_hue_curve(curve=[1], mask_name="nope")
    with pytest.raises(R.MaskNotFound):
        ed.set_oklch_hue_curve(curve=np.arange(65536), mask_name="nope")
    with pytest.raises(R.NotStrictlyIncreasing) as e:
        R.pchip_interpolate([0, 3, 2], [0, 1, 2], [1])
    assert e.value.index == 1


def test_masks_bookkeeping(ed):
    m = np.zeros((4, 6), np.float32)
    ed.add_mask("sky", m)
    ed.add_mask("face", m)
    assert ed.num_masks == 3
    ed.set_tone(1.0, 0, 0, 0, 0, 0, "sky")
    assert ed.gpu_params("sky").exposure == 1.0 and ed.gpu_params().exposure == 0.0
    with pytest.raises(R.MaskNotFound):
        ed.set_whitebalance(1, 1, "nope")
    ed.remove_mask("main")          # lib.rs:501-505: "main" is undeletable
    assert ed.num_masks == 3
    ed.remove_mask("sky")
    assert ed.num_masks == 2
    with pytest.raises(R.MaskNotFound):
        ed.gpu_params("sky")
    ed.set_vignette(5)
    ed.reset()                      # lib.rs:227-235
    assert ed.num_masks == 1 and ed.gpu_params().vignette == 0.0
    for i in range(3 * R.FUSED_MASKS):   # lib.rs:481-499: no limit on the number of masks
        ed.add_mask(f"m{i}", m)
    assert ed.num_masks == 3 * R.FUSED_MASKS + 1
    with pytest.raises(R.InvalidArgument):
        ed.add_mask("bad shape", np.zeros((3, 3), np.float32))


def test_headless_editor_cannot_compute(ed):
    with pytest.raises(R.GpuComputeError):
        ed.apply_adjustments()
    with pytest.raises(R.GpuComputeError):
        ed.to_flat_vec_rgba()


def test_precision_is_a_per_editor_hint():
    """rpf_set_precision / rpf_get_precision on a headless editor (no device needed for the bookkeeping): exact by default,
    unknown values refused, reset() leaves it alone (it is not edit state)."""
    import rawphotoforge_amd as R
    ed = R.PhotoEditor.headless(8, 8)
    assert ed.precision == R.PRECISION_EXACT
    ed.set_precision(R.PRECISION_FAST)
    assert ed.precision == R.PRECISION_FAST
    with pytest.raises(R.InvalidArgument):
        ed.set_precision(7)
    ed.reset()
    assert ed.precision == R.PRECISION_FAST
