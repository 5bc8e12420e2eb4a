#!/usr/bin/env python3
"""Derives the polynomial coefficients of the binary32 functions of include/nexus_fmath.h (sinf, cosf, expf, logf, atan2f, asinf).

No table is transcribed from anywhere: every polynomial is the weighted minimax fit (Lawson's iteratively re-weighted least
squares, in binary64 on Chebyshev nodes) of the function's own series remainder, printed as C float literals with all nine
significant digits.  tools/fmath_exhaustive.c then measures the functions built from them over EVERY binary32 argument.

    python tools/fmath_coeffs.py        # prints the coefficient blocks of nexus_fmath.h
"""
import numpy as np


def cheb_nodes(lo, hi, n):
    k = np.arange(n)
    return 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / n)


def lawson(x, g, w, degree, rounds=200):
    """minimises max |w (P - g)| over polynomials P of `degree` in x; returns the coefficients, lowest power first"""
    v = np.vander(x, degree + 1, increasing=True)
    lam = np.ones_like(x) / len(x)
    best, best_err = None, np.inf
    for _ in range(rounds):
        sw = np.sqrt(lam) * w
        c, *_ = np.linalg.lstsq(v * sw[:, None], g * sw, rcond=None)
        err = np.abs(w * (v @ c - g))
        if err.max() < best_err:
            best, best_err = c, err.max()
        lam = lam * err
        lam /= lam.sum()
    return best, best_err


def f32(c):
    return [float(np.float32(v)) for v in c]


def show(name, c, err, note):
    print("/* %s: weighted minimax error %.2e (%s) */" % (name, err, note))
    print("   ", ", ".join("%.9gf" % v for v in f32(c)))


def main():
    n = 4000
    # sin r = r + r z (S0 + S1 z + S2 z^2),  z = r^2 <= (pi/4)^2;  error relative to sin r ~ z (P - g)
    z = cheb_nodes(1e-9, (np.pi / 4 * 1.0001) ** 2, n)
    r = np.sqrt(z)
    c, e = lawson(z, (np.sin(r) / r - 1.0) / z, z * r / np.sin(r), 2)
    show("sin", c, e, "relative")
    # cos r = 1 - z/2 + z^2 (C0 + C1 z + C2 z^2);  cos r >= 0.707: error relative to cos r
    c, e = lawson(z, (np.cos(r) - 1.0 + 0.5 * z) / (z * z), z * z / np.cos(r), 2)
    show("cos", c, e, "relative")
    # e^r = 1 + r + r^2 (E0 + ... + E4 r^4),  |r| <= ln 2 / 2
    r = cheb_nodes(-0.5 * np.log(2) * 1.0001, 0.5 * np.log(2) * 1.0001, n)
    r = r[np.abs(r) > 1e-6]
    c, e = lawson(r, (np.expm1(r) - r) / (r * r), r * r / np.exp(r), 4)
    show("exp", c, e, "relative")
    # ln m = 2 s + 2 s z (L0 + L1 z + L2 z^2),  s = (m - 1) / (m + 1), z = s^2 <= ((sqrt 2 - 1) / (sqrt 2 + 1))^2
    smax = (np.sqrt(2) - 1) / (np.sqrt(2) + 1)
    z = cheb_nodes(1e-9, (smax * 1.0001) ** 2, n)
    s = np.sqrt(z)
    c, e = lawson(z, (np.arctanh(s) / s - 1.0) / z, z * s / np.arctanh(s), 2)
    show("log", c, e, "relative to ln m")
    # atan a = a + a z (A0 + ... + A8 z^8),  z = a^2 <= 1
    z = cheb_nodes(1e-9, 1.0, n)
    a = np.sqrt(z)
    for deg in (8,):
        c, e = lawson(z, (np.arctan(a) / a - 1.0) / z, z * a / np.arctan(a), deg)
        show("atan, degree %d" % deg, c, e, "relative")
    # asin x = x + x z (R0 + ... + R4 z^4),  z = x^2 <= 1/4
    z = cheb_nodes(1e-9, 0.25, n)
    x = np.sqrt(z)
    for deg in (4,):
        c, e = lawson(z, (np.arcsin(x) / x - 1.0) / z, z * x / np.arcsin(x), deg)
        show("asin, degree %d" % deg, c, e, "relative")
    # the constants of the reductions, as the binary32 / binary64 values nearest to them
    import mpmath as mp
    mp.mp.dps = 40
    hi = np.float32(float(mp.pi / 2))
    print("pi/2 hi %.9gf lo %.9gf" % (hi, np.float32(float(mp.pi / 2 - mp.mpf(float(hi))))))
    hi = np.float32(float(mp.pi))
    print("pi   hi %.9gf lo %.9gf" % (hi, np.float32(float(mp.pi - mp.mpf(float(hi))))))
    ln2hi = np.frombuffer(np.uint32(np.frombuffer(np.float32(float(mp.log(2))).tobytes(), np.uint32)[0] & 0xfffff000).tobytes(), np.float32)[0]
    print("ln2  hi %.9gf lo %.9gf  (hi: the low 12 bits cleared, k * hi exact for |k| < 2^12)" % (ln2hi, np.float32(float(mp.log(2) - mp.mpf(float(ln2hi))))))
    print("log2e %.9gf  2/pi %.17g" % (np.float32(float(1 / mp.log(2))), float(2 / mp.pi)))


if __name__ == "__main__":
    main()
