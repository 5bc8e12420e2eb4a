"""include/nexus_fmath.h — the transcendental functions the shading path uses, one text compiled by the device code and by the
oracle (replacing libm's sin / cos / expf / logf / sinf / cosf / atan2f / asinf / pow at
/root/reference/Nexus/src/Cuda/Random.cuh:119-121, Cuda/BSDF/Microfacet.cuh:18,75, Cuda/PathTracer/PathTracer.cu:65-83, Utils/Utils.h:51-54).

CPU: the oracle's build of the text against a 50-digit reference (mpmath) — float functions (binary32 arithmetic since round 5;
exhaustively measured by tools/fmath_exhaustive.c) within 2 ulp, double ones within 4 ulp — and the IEEE / C Annex F special cases.  GPU: the device's build gives the same bits as the oracle's on a million arguments
per function, specials included: what makes frames comparable with np.array_equal."""
import numpy as np
import pytest

from nexus_amd import pod
from tests import oracle_lib as O

OPS = pod.NXF_OPS


def _args(name, n, seed):
    """arguments that cover what the path feeds the function, and well beyond"""
    rng = np.random.default_rng(seed)
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)  # noqa: E731
    if name in ("sin", "cos"):
        return np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, n // 2), rng.uniform(-1e5, 1e5, n // 4), rng.normal(size=n // 4) * 1e-3]), None
    if name in ("sinf", "cosf"):
        return f32(np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, n // 2), rng.uniform(-1e4, 1e4, n // 4), rng.normal(size=n // 4) * 1e-3])), None
    if name == "exp":
        return rng.uniform(-750, 710, n), None
    if name == "expf":
        return f32(np.concatenate([rng.uniform(-110, 89, n // 2), -np.exp(rng.uniform(-20, 5, n // 2))])), None
    if name == "log":
        return np.concatenate([np.exp(rng.uniform(-745, 709, n // 2)), 1 + rng.uniform(-1e-2, 1e-2, n // 2)]), None
    if name == "logf":
        return f32(np.concatenate([np.exp(rng.uniform(-103, 88, n // 3)), rng.uniform(0, 1, n // 3), 1 - np.exp(rng.uniform(-16, 0, n // 3))])), None
    if name == "pow":
        return np.concatenate([rng.uniform(0, 1, n // 2), np.exp(rng.uniform(-8, 8, n // 2))]), np.concatenate([np.full(n // 2, 0.45454545454), rng.uniform(-3, 3, n // 2)])
    if name == "atan2":
        return rng.normal(size=n) * np.exp(rng.uniform(-30, 30, n)), rng.normal(size=n) * np.exp(rng.uniform(-30, 30, n))
    if name == "atan2f":
        return f32(rng.normal(size=n) * np.exp(rng.uniform(-10, 10, n))), f32(rng.normal(size=n) * np.exp(rng.uniform(-10, 10, n)))
    if name == "asin":
        return np.concatenate([rng.uniform(-1, 1, n // 2), np.sign(rng.normal(size=n // 2)) * (1 - np.exp(rng.uniform(-35, 0, n // 2)))]), None
    if name == "asinf":
        return f32(np.concatenate([rng.uniform(-1, 1, n // 2), np.sign(rng.normal(size=n // 2)) * (1 - np.exp(rng.uniform(-16, 0, n // 2)))])), None
    raise KeyError(name)


_SPECIAL = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 0.5, -0.5, 1e-310, 5e-324, 1e-45, 1e-38, 1e308, 3e38, 89.0, -104.0, -150.0, 709.8, -745.2, 2.0, -2.0,
                     1.0000001, np.pi, -np.pi, np.pi / 2, 1e9, 1e16, 1e300])


def _special_pairs():
    a, b = np.meshgrid(_SPECIAL, _SPECIAL)
    return a.ravel().copy(), b.ravel().copy()


@pytest.mark.parametrize("name", sorted(OPS))
def test_the_shared_functions_against_a_50_digit_reference(name):
    # (not importorskip: this is the one check of the shared functions that is independent of their own text, and a box without
    #  mpmath must say so instead of passing with it silently gone)
    try:
        import mpmath as mp
    except ImportError:
        pytest.fail("mpmath is not importable here: the 50-digit comparison of include/nexus_fmath.h cannot run")
    mp.mp.dps = 50
    ref = {"sin": mp.sin, "cos": mp.cos, "exp": mp.exp, "log": mp.log, "pow": mp.power, "atan2": mp.atan2, "asin": mp.asin}[name.rstrip("f")]
    is_float = name.endswith("f")
    a, b = _args(name, 6000, seed=11)
    got = O.fmath_batch(OPS[name], a, b)
    worst = 0.0
    for i in range(len(a)):
        r = ref(mp.mpf(float(a[i]))) if b is None else ref(mp.mpf(float(a[i])), mp.mpf(float(b[i])))
        rf = float(r)
        if not np.isfinite(rf) or rf == 0.0:
            continue
        if is_float and (abs(rf) > 3.4e38 or abs(rf) < 1.2e-38):
            continue  # (overflow / gradual underflow: checked by the specials test)
        if not is_float and abs(rf) < 2.3e-308:
            continue
        ulp = float(np.spacing(np.float32(abs(rf)))) if is_float else float(np.spacing(abs(rf)))
        worst = max(worst, float(abs(mp.mpf(float(got[i])) - r) / ulp))
    # sin / cos of arguments up to 1e5: the two-term reduction leaves ~1e-16 * |x| of absolute error; pow = exp(y ln x) carries
    # the rounding of y ln x, i.e. about |y ln x| ulp (the one caller, LinearToGamma, rounds the result to float)
    bound = 2.0 if is_float else {"sin": 64.0, "cos": 64.0, "pow": 32.0}.get(name, 4.0)
    print("%s: worst error %.4f ulp over %d arguments" % (name, worst, len(a)))
    assert worst <= bound


def test_the_special_cases_follow_ieee_and_annex_f():
    with np.errstate(all="ignore"):
        x = _SPECIAL
        def same(got, want):  # NaN where NaN is due, and the same sign everywhere else (zeros included)
            return np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(want[~np.isnan(want)]))
        small = np.abs(x) < 1e6
        assert np.allclose(O.fmath_batch(OPS["sin"], x[small]), np.sin(x[small]), rtol=1e-14, atol=0, equal_nan=True)
        assert np.allclose(O.fmath_batch(OPS["cos"], x[small]), np.cos(x[small]), rtol=1e-14, atol=0, equal_nan=True)
        big = O.fmath_batch(OPS["sin"], x[~small])
        assert np.all(np.isnan(big) | (np.abs(big) <= 1.0))  # huge arguments: not accurate, but bounded and the same on both sides
        assert same(O.fmath_batch(OPS["sin"], np.array([0.0, -0.0])), np.array([0.0, -0.0]))
        assert np.allclose(O.fmath_batch(OPS["exp"], x), np.exp(x), rtol=1e-14, atol=0, equal_nan=True)
        assert np.allclose(O.fmath_batch(OPS["log"], x), np.log(x), rtol=1e-14, atol=0, equal_nan=True)
        xf = x.astype(np.float32)
        assert np.allclose(O.fmath_batch(OPS["expf"], xf.astype(np.float64)), np.exp(xf).astype(np.float64), rtol=2e-7, atol=0, equal_nan=True)
        assert np.allclose(O.fmath_batch(OPS["logf"], xf.astype(np.float64)), np.log(xf).astype(np.float64), rtol=2e-7, atol=0, equal_nan=True)
        assert np.allclose(O.fmath_batch(OPS["asin"], x), np.arcsin(x), rtol=1e-14, atol=0, equal_nan=True)
        assert np.allclose(O.fmath_batch(OPS["asinf"], xf.astype(np.float64)), np.arcsin(xf).astype(np.float64), rtol=2e-7, atol=0, equal_nan=True)
        a, b = _special_pairs()
        got, want = O.fmath_batch(OPS["atan2"], a, b), np.arctan2(a, b)
        assert np.allclose(got, want, rtol=1e-14, atol=0, equal_nan=True) and same(got, want)
        af, bf = a.astype(np.float32), b.astype(np.float32)
        got, want = O.fmath_batch(OPS["atan2f"], af.astype(np.float64), bf.astype(np.float64)), np.arctan2(af, bf).astype(np.float64)
        assert np.allclose(got, want, rtol=2e-7, atol=0, equal_nan=True) and same(got, want)
        # x^y for the bases the path has (>= 0); a negative base is NaN by this text's own rule
        nonneg = ~(a < 0) & ~(np.signbit(a) & (a == 0))
        got, want = O.fmath_batch(OPS["pow"], a[nonneg], b[nonneg]), np.power(a[nonneg], b[nonneg])
        assert np.allclose(got, want, rtol=2e-13, atol=0, equal_nan=True)
        assert np.all(np.isnan(O.fmath_batch(OPS["pow"], np.array([-2.0, -0.5]), np.array([2.0, 0.5]))))


def test_the_float_functions_coefficients_are_what_the_fitting_script_derives():
    """include/nexus_fmath.h says its binary32 polynomials are fitted (tools/fmath_coeffs.py: weighted minimax fits of each function's
    series remainder), not transcribed from anywhere: every coefficient the script prints must stand in the header, digit for digit."""
    import io
    import os
    import re
    import runpy
    from contextlib import redirect_stdout

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = io.StringIO()
    with redirect_stdout(buf):
        runpy.run_path(os.path.join(root, "tools", "fmath_coeffs.py"), run_name="__main__")
    printed = re.findall(r"-?\d\.\d+(?:e[+-]\d+)?f", buf.getvalue())
    assert len(printed) >= 30
    header = open(os.path.join(root, "include", "nexus_fmath.h")).read()
    missing = [c for c in printed if c not in header]
    assert not missing, "coefficients the script derives that the header does not hold: %r" % missing


def test_tonemap_uses_the_shared_pow():
    """LinearToGamma (Utils/Utils.h:51-54) through nxf_pow: the 8-bit values of the oracle's tonemap equal a 50-digit evaluation
    wherever that is not within 1e-6 of a rounding boundary"""
    rng = np.random.default_rng(2)
    rgb = rng.uniform(0, 4, (2000, 3)).astype(np.float32)
    for c in rgb[:200]:
        packed = O.lib().orc_tonemap_rgba8(O._ptr(np.ascontiguousarray(c, np.float32)))
        for k in range(3):
            x = np.float32(c[k]) * np.float32(0.6)
            x = np.clip((x * (np.float32(2.51) * x + np.float32(0.03))) / (x * (np.float32(2.43) * x + np.float32(0.59)) + np.float32(0.14)), 0, 1).astype(np.float32)
            y = float(x) ** 0.45454545454 * 255.0
            if abs(y - round(y)) > 1e-4:
                assert ((packed >> (8 * k)) & 0xff) == int(np.float32(np.float32(float(x) ** 0.45454545454) * np.float32(255.0)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OPS))
def test_device_and_oracle_compute_the_same_bits(name):
    """The point of the shared text: hipcc for gfx950 and gcc for x86-64 produce identical results — a million arguments per
    function plus every pair of special values, compared as bit patterns."""
    from nexus_amd import capi

    a, b = _args(name, 1 << 20, seed=5)
    sa, sb = _special_pairs()
    if name.endswith("f"):
        with np.errstate(all="ignore"):
            sa, sb = sa.astype(np.float32).astype(np.float64), sb.astype(np.float32).astype(np.float64)
    a = np.concatenate([a, sa])
    b = None if b is None else np.concatenate([b, sb])
    with capi.Context(64, 64) as ctx:
        got = ctx.fmath_batch(OPS[name], a, b)
    want = O.fmath_batch(OPS[name], a, b)
    diff = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))
    assert len(diff) == 0, "%s: %d of %d differ, first at %r: device %r, oracle %r" % (name, len(diff), len(a), a[diff[:3]], got[diff[:3]], want[diff[:3]])
