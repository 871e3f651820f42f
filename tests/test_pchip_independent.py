"""A second, independently written restatement of the reference's PCHIP (rust/photo-editor/src/interpolation.rs:11-123) — numpy
binary32 array arithmetic, vectorised over the evaluation points, np.searchsorted for the interval — against the oracle's scalar C
loop and the product's C++ (rpf_host.h).  The oracle and the product were written by the same hand from the same file; this one
shares no code, no compiler and no control flow with them: agreement on every one of the 65 536 entries, for thousands of random
control-point sets, is what a transcription slip in either (an operand order, a slope rule, an off-by-one in the search) would break."""
import numpy as np
import pytest

import oracle as O
import rawphotoforge_amd as R

F = np.float32


def pchip_numpy(xs, ys, x_eval):
    x = np.asarray(xs).astype(F)                     # interpolation.rs:35-37: everything in binary32
    y = np.asarray(ys).astype(F)
    xe = np.asarray(x_eval).astype(F)
    n = x.size
    h = x[1:] - x[:-1]                               # :46-54
    assert (h > 0).all()
    dl = (y[1:] - y[:-1]) / h
    m = np.zeros(n, F)                               # :57-78
    m[0], m[-1] = dl[0], dl[-1]
    if n > 2:
        a, b = dl[:-1], dl[1:]
        w1 = F(2.0) * h[1:] + h[:-1]
        w2 = h[1:] + F(2.0) * h[:-1]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inner = (w1 + w2) / (w1 / a + w2 / b)
        m[1:-1] = np.where(a * b <= F(0.0), F(0.0), inner)
    i = np.clip(np.searchsorted(x, xe, side="right") - 1, 0, n - 2)   # :92-101 (an exact hit on x[j] gives j)
    hv = h[i]
    t = (xe - x[i]) / hv                             # :103-106
    t2 = t * t
    t3 = t2 * t
    h00 = F(2.0) * t3 - F(3.0) * t2 + F(1.0)         # :109-112
    h10 = t3 - F(2.0) * t2 + t
    h01 = F(-2.0) * t3 + F(3.0) * t2
    h11 = t3 - t2
    out = h00 * y[i] + h10 * hv * m[i] + h01 * y[i + 1] + h11 * hv * m[i + 1]   # :119, left to right
    out = np.where(xe <= x[0], y[0], np.where(xe >= x[-1], y[-1], out))          # :82-89
    # `as i32` (:122): toward zero, saturating, NaN -> 0
    o64 = np.nan_to_num(out.astype(np.float64), nan=0.0, posinf=2.0**31, neginf=-2.0**31)
    return np.clip(np.trunc(o64), -2.0**31, 2.0**31 - 1).astype(np.int64).astype(np.int32)


def _random_points(rng):
    n = int(rng.integers(2, 17))
    kind = rng.random()
    if kind < 0.5:      # a UI curve: x across the 16-bit range, y anywhere near it (setters clamp afterwards)
        xs = np.sort(rng.choice(65536, n, replace=False))
        ys = rng.integers(-3000, 70000, n)
    elif kind < 0.8:    # clustered x (tiny intervals), monotone y
        x0 = int(rng.integers(0, 60000))
        xs = x0 + np.sort(rng.choice(400, n, replace=False))
        ys = np.sort(rng.integers(0, 65536, n))
    else:               # wide values: the binary32 conversions of x and y round
        xs = np.sort(rng.choice(2**26, n, replace=False)) - 2**25
        ys = rng.integers(-2**27, 2**27, n)
    return xs.astype(np.int64), ys.astype(np.int64)


def test_numpy_restatement_equals_oracle_and_product_on_random_control_points():
    rng = np.random.default_rng(20261005)
    ev = np.arange(65536)
    for case in range(1500):
        xs, ys = _random_points(rng)
        xf = xs.astype(F)
        if (xf[1:] - xf[:-1] <= 0).any():   # the binary32 conversion merged two x: the reference raises (checked below)
            with pytest.raises(O.PchipError):
                O.pchip_interpolate(xs, ys, ev)
            continue
        e = ev if case % 3 else rng.integers(-2**20, 2**27, 4096)
        want = pchip_numpy(xs, ys, e)
        got_o = O.pchip_interpolate(xs, ys, e)
        assert np.array_equal(got_o, want), (case, xs, ys, np.flatnonzero(got_o != want)[:5])
        if np.abs(xs).max() < 2**31 and np.abs(ys).max() < 2**31:
            got_p = R.pchip_interpolate(xs, ys, e)
            assert np.array_equal(got_p, want), (case, xs, ys)


def test_numpy_restatement_reproduces_the_two_point_table_counts():
    """the UI's default 2-point identity gives 15 795 entries equal to i - 1 after truncation, the 2-point constant
    32767 gives a mix of 32766 / 32767 — counts the survey derived on its own, reproduced by this restatement too."""
    ev = np.arange(65536)
    ident = pchip_numpy([0, 65535], [0, 65535], ev)
    assert int((ident == ev - 1).sum()) == 15795 and int((ident == ev).sum()) == 65536 - 15795
    const = pchip_numpy([0, 65535], [32767, 32767], ev)
    assert set(np.unique(const)) == {32766, 32767} and int((const == 32766).sum()) == 2532
