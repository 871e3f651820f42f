#!/usr/bin/env python3
"""Derive the polynomial constants hard-coded in rawphotoforge_amd/csrc/rpf_math.h.

Build-time tool (needs mpmath, present in the build container only).  Prints C
hex-float literals; the header carries the printed values so nothing here runs
at build or run time.

  atan:  atan(q) = q + q^3 * P(q^2) for |q| <= tan(pi/8) (+ margin), P of degree 9
"""
import mpmath as mp

mp.mp.prec = 200


def hexd(x):
    return float(x).hex()


def atan_poly(deg=9, qmax=mp.mpf("0.4143")):
    smax = qmax * qmax

    def g(s):
        s = mp.mpf(s)
        if s < mp.mpf(2) ** -60:
            return -mp.mpf(1) / 3 + s / 5
        q = mp.sqrt(s)
        return (mp.atan(q) - q) / (q * s)

    coeffs, err = mp.chebyfit(g, [0, smax], deg + 1, error=True)
    coeffs = [float(c) for c in coeffs[::-1]]  # ascending powers, rounded to binary64
    # true error of the ROUNDED polynomial, relative to atan(q), sampled densely
    worst = mp.mpf(0)
    for i in range(1, 4001):
        q = qmax * i / 4000
        s = q * q
        p = mp.mpf(0)
        for c in reversed(coeffs):
            p = p * s + mp.mpf(c)
        approx = q + q * s * p
        worst = max(worst, abs(approx / mp.atan(q) - 1))
    return coeffs, err, worst


def hue_candidate_poly(nterms=5):
    """atan(q) * 65535 / (2 pi) ~ q * P(q^2) on [0, 1] for hue_index's candidate (numpy only): iteratively re-weighted
    least squares towards the minimax fit; returns the binary32 coefficients and the max error in index units."""
    import numpy as np
    n = 2000
    z = (np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) / 2
    q = np.sqrt(z)
    a = np.vander(z, nterms, increasing=True) * q[:, None]
    w = np.ones(n)
    for _ in range(60):
        c, *_ = np.linalg.lstsq(a * w[:, None], np.arctan(q) * w, rcond=None)
        e = a @ c - np.arctan(q)
        w = w * (1 + 3 * np.abs(e) / np.abs(e).max())
        w /= w.mean()
    k = 65535 / (2 * np.pi)
    c32 = [float(np.float32(v * k)) for v in c]
    qq = np.linspace(0, 1, 200001)
    err = qq * np.polyval(c32[::-1], qq * qq) - np.arctan(qq) * k
    return c32, float(np.abs(err).max())


def fast_constants():
    """rpf_fast.h: the eight-term hue-position polynomial (index units, divisor f32(2 pi)), the sine / cosine minimax
    polynomials on [-pi/4, pi/4] and the three-term split of pi/2 (numpy + mpmath)."""
    import numpy as np

    def lawson(basis, target, weight, iters=300):
        w = np.ones(len(target))
        for _ in range(iters):
            cc, *_ = np.linalg.lstsq(basis * (w * weight)[:, None], target * w * weight, rcond=None)
            e = (basis @ cc - target) * weight
            w = w * (1 + 3 * np.abs(e) / np.abs(e).max())
            w /= w.mean()
        return cc, np.abs(e).max()
    n = 4000
    z = (np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) / 2
    q = np.sqrt(z)
    c, _ = lawson(np.vander(z, 8, increasing=True) * q[:, None], np.arctan(q), np.ones(n))
    k = 65535 / float(np.float32(2.0 * 3.14159265359))
    print("// hue_position_fast / hue_fraction_fast:", ", ".join(f"{float(np.float32(v * k))!r}f" for v in c))
    r = (np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) / 2 * 0.7954
    r = r[r > 1e-4]
    zz = r * r
    S, es = lawson(np.vander(zz, 3, increasing=True), np.array([float((mp.sin(x) - x) / x ** 3) for x in map(mp.mpf, r)]), r ** 3 / np.sin(r))
    Cc, ec = lawson(np.vander(zz, 3, increasing=True), np.array([float((mp.cos(x) - 1 + x * x / 2) / x ** 4) for x in map(mp.mpf, r)]), zz * zz / np.cos(r))
    print("// sin:", [float(np.float32(v)) for v in S], "rel err", es, " cos:", [float(np.float32(v)) for v in Cc], "rel err", ec)
    H = mp.mpf(1.5703125)
    rest = mp.pi / 2 - H
    e = mp.floor(mp.log(rest, 2))
    M = mp.floor(rest / mp.mpf(2) ** (e - 10)) * mp.mpf(2) ** (e - 10)
    print("// pi/2 =", float(H), "+", repr(float(M)), "+", repr(float(np.float32(float(rest - M)))), "; 2/pi =", float(np.float32(2 / np.pi)))


def binomials(c_float32, kmax=5):
    """binomial(c, k), k = 1..kmax, for c a binary32 value, as binary64 hex literals (PowCoef in rpf_math.h)."""
    from fractions import Fraction
    c = Fraction(float(c_float32))
    b, out = Fraction(1), []
    for k in range(1, kmax + 1):
        b = b * (c - (k - 1)) / k
        out.append(float(b).hex())
    return out


if __name__ == "__main__":
    import numpy as _np
    print("// hue_index candidate polynomial (index units), max error:", hue_candidate_poly())
    fast_constants()
    for _name, _c in (("cbrt", _np.float32(1 / 3)), ("gamma", _np.float32(1 / 2.4)), ("decode", _np.float32(2.4))):
        print(f"// PowCoef {_name}:", binomials(_c))
    for deg in (7, 8, 9):
        c, err, worst = atan_poly(deg)
        print(f"// atan P degree {deg}: chebyfit err {mp.nstr(err, 5)}, rel err of rounded poly {mp.nstr(worst, 5)}"
              f" = 2^{mp.nstr(mp.log(worst, 2), 5)}")
        if deg in (8, 9):
            print("static const double ATAN_P[] = {")
            for v in c:
                print(f"    {hexd(v)},")
            print("};")
    print("// pi/4, pi/2, pi as binary64")
    for name, v in (("PI_4", mp.pi / 4), ("PI_2", mp.pi / 2), ("PI", mp.pi)):
        print(f"static const double K_{name} = {hexd(v)};")
    pif = float(mp.mpf(3.14159265359))  # wgsl:78 literal as binary64 ...
    import struct
    pif32 = struct.unpack("f", struct.pack("f", pif))[0]  # ... converted to binary32
    print(f"// hue step 2*f32(3.14159265359)/65535 as binary64 (f32 pi = {pif32!r})")
    print(f"static const double K_HUE_STEP = {hexd(mp.mpf(2) * mp.mpf(pif32) / 65535)};")
