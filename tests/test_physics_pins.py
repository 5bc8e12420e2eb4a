"""Algorithm-independent pins of the shading path: what the light transport must converge to, whatever the code.

The reference holds no test or fixture for this path, so device and oracle are pinned against each other bit
for bit — and since round 4 they even share the text of the transcendental functions.  A restatement error in the light pdf, the
power heuristic, the `lastPdf` handling or the Russian roulette (PathTracer.cu:167-175, 262-289, 363-382) would be common to
both sides and invisible to every bit-equality test.  These tests do not compare the two sides with each other; they compare each
of them with physics:

  (i)   the reference's own validation (README.md:16-27): the Cornell box rendered with next-event estimation + MIS and without
        (`useMIS`, PathTracer.cu:352-385, 431-432) converges to the same image;
  (ii)  white furnace: inside a closed Lambertian box of albedo rho whose walls all emit L, a path of at most n vertices collects
        L (1 + rho + ... + rho^(n-1)) — with Russian roulette on, with and without MIS (every wall is a light the NEE samples);
  (iii) direct light of a rectangular emitter on a parallel diffuse floor: radiance = rho L F(x), F the closed-form point-to-
        rectangle form factor — estimated by BSDF sampling alone and by NEE + MIS.

Each estimate comes with its own measured standard error (frames are independent: the random numbers are keyed by frame), the
comparisons are z-scores per 16 x 16 pixel block and colour channel; the bars are |z| < 4.5 for every block and a mean z^2 below
1.6 (its expectation is 1), and a floor on the estimates' precision so that a pass means something.  On the device (-m gpu) the
sample counts are what a 2 Gsamples/s path affords; the oracle twins (CPU) run the same checks at the size a few seconds allow."""
import numpy as np
import pytest

from nexus_amd import capi, pod, scenegen, workloads
from tests import oracle_lib as O
from tests import scene_helpers as SH

BLOCK = 16


def _block_ids(W, H):
    jj, ii = np.mgrid[0:H, 0:W]
    return ((jj // BLOCK) * (W // BLOCK) + ii // BLOCK).reshape(-1), (W // BLOCK) * (H // BLOCK)


class _Estimate:
    """running mean and variance of the per-frame block means"""

    def __init__(self, W, H):
        self.ids, self.nb = _block_ids(W, H)
        self.per_block = np.bincount(self.ids, minlength=self.nb).astype(np.float64)
        self.n = 0
        self.s = np.zeros((self.nb, 3))
        self.s2 = np.zeros((self.nb, 3))

    def add(self, radiance):  # (pixels, 3) of one frame
        m = np.stack([np.bincount(self.ids, weights=radiance[:, c].astype(np.float64), minlength=self.nb) for c in range(3)], 1) / self.per_block[:, None]
        self.n += 1
        self.s += m
        self.s2 += m * m

    @property
    def mean(self):
        return self.s / self.n

    @property
    def se(self):  # standard error of the mean
        var = np.maximum(self.s2 / self.n - self.mean ** 2, 0.0) * self.n / max(1, self.n - 1)
        return np.sqrt(var / self.n)


def _oracle_estimate(scene, W, H, frames, threads=8):
    w = O.Wavefront(scene.oracle(), W * H, None, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_REFERENCE)
    e = _Estimate(W, H)
    for f in range(1, frames + 1):
        w.render(f, threads=threads)
        e.add(w.radiance())
    w.close()
    return e


def _gpu_estimate(ctx, scene, W, H, frames, per_pass=64):
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    ctx.set_frames_per_pass(per_pass)
    ctx.reset_frame_number()
    e = _Estimate(W, H)
    assert frames % per_pass == 0
    for _ in range(frames // per_pass):
        ctx.render_frame()
        r = ctx.read_radiance().reshape(per_pass, W * H, 3)
        for k in range(per_pass):
            e.add(r[k])
    return e


def _z(a_mean, a_se, b_mean, b_se, systematic=0.0):
    """z-scores of a - b; `systematic`: a relative allowance on b (model error of a closed form evaluated per pixel centre)"""
    return (a_mean - b_mean) / np.sqrt(a_se ** 2 + b_se ** 2 + (systematic * np.abs(b_mean)) ** 2 + 1e-30)


def _assert_agree(z, what):
    z = z[np.isfinite(z)]
    print("%s: %d comparisons, max |z| %.2f, mean z^2 %.2f" % (what, z.size, np.abs(z).max(), (z * z).mean()))
    assert np.abs(z).max() < 4.5, what
    assert (z * z).mean() < 1.6, what


# ---- scenes ------------------------------------------------------------------------------------------------------------

RHO = np.array([0.3, 0.5, 0.7])
LE = 1.5


def _furnace_scene(W, H, path_length, use_mis):
    """the inside of the cube [-1, 1]^3: one mesh of 12 triangles, Lambertian albedo RHO, every wall emitting LE"""
    c = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    faces = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    box = np.concatenate([scenegen.quad(c[a], c[b], c[d], c[e]) for a, b, d, e in faces])
    mats = np.array([pod.make_material(pod.MAT_DIFFUSE, albedo=tuple(RHO), emissive=(1.0, 1.0, 1.0), intensity=LE)], dtype=pod.MAT_DT)
    cam = capi.camera_init((0.1, -0.2, 0.3), (0.3, 0.2, -1.0) / np.linalg.norm((0.3, 0.2, -1.0)), 70.0, W, H, 5.0, 0.0)
    sc = SH.BuiltScene([box], [(0, 0, workloads.IDENTITY)], materials=mats, camera=cam,
                       settings=workloads.make_settings(use_mis=use_mis, path_length=path_length, background=(1, 1, 1), background_intensity=0.0))
    sc.lights = SH.mesh_lights(sc.instances, sc.materials)
    assert len(sc.lights) == 1
    return sc


def _furnace_expectation(path_length):
    return LE * sum(RHO ** k for k in range(path_length))


def _furnace_rule_loss(scene, W, H, path_length, samples=2_000_000, seed=7):
    """What the reference's pdf validity rule takes out of the furnace under MIS, by an independent Monte Carlo of the GEOMETRY.

    With MIS a BSDF-sampled hit on an emitter is weighted power_heuristic(lastPdf, lightPdf) and the light sample of the previous
    vertex carries the complement — unless lightPdf = d^2 / (lights x triangles x area x cos) is not `valid`, i.e. <= 1e-4
    (Sampler.cuh:58-61): then the hit's weight is 0 (PathTracer.cu:379-380) and the light sample is rejected as well (:274), so that
    configuration contributes NOTHING.  Behind a far, small light that never happens; in a box whose walls all emit it does, for the
    short segments near edges and corners: here d^2 / (24 cos) <= 1e-4.  Returns g[k], k = 1 .. path_length - 1: the probability that
    the k-th indirect segment is such a configuration, for cosine-distributed directions from the points the camera sees through a
    16 x 16 pixel block (one row of the result per block) — so that E[MIS] = L sum_k rho^k (1 - g[k]), g[0] = 0 (the first vertex is
    never weighted)."""
    rng = np.random.RandomState(seed)
    cam = scene.camera
    pos = cam["position"].astype(np.float64)
    x, y = rng.rand(samples, 1), rng.rand(samples, 1)
    d = cam["lowerLeftCorner"].astype(np.float64) + cam["viewportX"].astype(np.float64) * x + cam["viewportY"].astype(np.float64) * y - pos
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = np.repeat(pos[None, :], samples, 0)
    nbx = W // BLOCK
    block = (np.minimum((y[:, 0] * H).astype(int), H - 1) // BLOCK) * nbx + np.minimum((x[:, 0] * W).astype(int), W - 1) // BLOCK
    nb = nbx * (H // BLOCK)
    per_block = np.bincount(block, minlength=nb).astype(np.float64)

    def exit_point(p, d):
        t = np.full(len(p), np.inf)
        axis_hit = np.zeros(len(p), int)
        for a in range(3):
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = np.where(d[:, a] > 0, (1 - p[:, a]) / d[:, a], np.where(d[:, a] < 0, (-1 - p[:, a]) / d[:, a], np.inf))
            closer = tt < t
            t = np.where(closer, tt, t)
            axis_hit = np.where(closer, a, axis_hit)
        return t, axis_hit

    g = [np.zeros(nb)]
    t, axis = exit_point(p, d)
    p = np.clip(p + d * t[:, None], -1, 1)
    for _ in range(1, path_length):
        n = np.zeros_like(p)  # inward normal of the wall the path stands on
        n[np.arange(len(p)), axis] = -np.sign(p[np.arange(len(p)), axis])
        r1, r2 = rng.rand(len(p)), rng.rand(len(p))
        phi, B = 2 * np.pi * r1, np.sqrt(r2)
        u = np.roll(n, 1, axis=1)
        v = np.cross(n, u)
        w = u * (np.cos(phi) * B)[:, None] + v * (np.sin(phi) * B)[:, None] + n * np.sqrt(1 - r2)[:, None]
        t, axis2 = exit_point(p, w)
        cos_l = np.abs(w[np.arange(len(p)), axis2])
        with np.errstate(divide="ignore", invalid="ignore"):
            light_pdf = t * t / (12.0 * 2.0 * cos_l)  # 1 light, 12 triangles of area 2
        g.append(np.bincount(block, weights=(~(light_pdf > 1e-4)).astype(np.float64), minlength=nb) / per_block)
        p = np.clip(p + w * t[:, None], -1, 1)
        axis = axis2
    return np.stack(g, 1)  # [block][k]


FLOOR_RHO = 0.6
LIGHT_LE = 8.0
LIGHT = (-0.5, 0.7, -0.4, 0.3, 1.2)  # x0, x1, z0, z1, height


def _quad_light_scene(W, H, use_mis):
    """a diffuse floor (y = 0) under a rectangular emitter parallel to it; the camera sees the floor only; paths of two vertices:
    the floor point and what its light sample / BSDF sample finds"""
    x0, x1, z0, z1, h = LIGHT
    floor = scenegen.quad((-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8))
    light = scenegen.quad((x0, h, z0), (x1, h, z0), (x1, h, z1), (x0, h, z1))
    mats = np.array([pod.make_material(pod.MAT_DIFFUSE, albedo=(FLOOR_RHO,) * 3),
                     pod.make_material(pod.MAT_DIFFUSE, albedo=(0.0, 0.0, 0.0), emissive=(1.0, 1.0, 1.0), intensity=LIGHT_LE)], dtype=pod.MAT_DT)
    eye = np.array((3.0, 0.8, 0.3))  # below the emitter's height, looking down at the floor beneath it: no ray can reach the emitter first
    fwd = np.array((0.0, 0.0, 0.0)) - eye
    cam = capi.camera_init(tuple(eye), fwd / np.linalg.norm(fwd), 20.0, W, H, 5.0, 0.0)
    sc = SH.BuiltScene([floor, light], [(0, 0, workloads.IDENTITY), (1, 1, workloads.IDENTITY)], materials=mats, camera=cam,
                       settings=workloads.make_settings(use_mis=use_mis, path_length=2, background=(1, 1, 1), background_intensity=0.0))
    sc.lights = SH.mesh_lights(sc.instances, sc.materials)
    assert len(sc.lights) == 1
    return sc


def _form_factor(px, pz):
    """differential area at (px, 0, pz), normal +y, to the rectangle LIGHT parallel to it: sum over the four corners of
    G(a, b) = [a / sqrt(a^2 + h^2) atan(b / sqrt(a^2 + h^2)) + b / sqrt(b^2 + h^2) atan(a / sqrt(b^2 + h^2))] / (2 pi)"""
    x0, x1, z0, z1, h = LIGHT

    def G(a, b):
        ra, rb = np.sqrt(a * a + h * h), np.sqrt(b * b + h * h)
        return (a / ra * np.arctan(b / ra) + b / rb * np.arctan(a / rb)) / (2.0 * np.pi)

    return G(x1 - px, z1 - pz) - G(x0 - px, z1 - pz) - G(x1 - px, z0 - pz) + G(x0 - px, z0 - pz)


def _quad_light_expectation(scene, W, H, sub=4):
    """rho L F at the floor point each pixel sees, averaged over sub x sub positions in the pixel, then over blocks"""
    cam = scene.camera
    pos = cam["position"].astype(np.float64)
    acc = np.zeros(W * H)
    jj, ii = np.mgrid[0:H, 0:W]
    for a in range(sub):
        for b in range(sub):
            x = ((ii + (a + 0.5) / sub) / W).reshape(-1, 1)
            y = ((jj + (b + 0.5) / sub) / H).reshape(-1, 1)
            d = cam["lowerLeftCorner"].astype(np.float64) + cam["viewportX"].astype(np.float64) * x + cam["viewportY"].astype(np.float64) * y - pos
            t = -pos[1] / d[:, 1]
            assert np.all(t > 0), "every pixel must see the floor"
            p = pos + d * t[:, None]
            assert np.all(np.abs(p[:, 0]) < 8) and np.all(np.abs(p[:, 2]) < 8)
            acc += FLOOR_RHO * LIGHT_LE * _form_factor(p[:, 0], p[:, 2])
    acc /= sub * sub
    ids, nb = _block_ids(W, H)
    return (np.bincount(ids, weights=acc, minlength=nb) / np.bincount(ids, minlength=nb))[:, None] * np.ones((1, 3))


# ---- the checks, on an estimator (oracle or device) ---------------------------------------------------------------------

def _check_furnace(estimate, frames, rel_se_bar):
    for use_mis in (False, True):
        for n in (1, 3, 6):
            e = estimate(lambda W, H: _furnace_scene(W, H, n, use_mis), frames)
            want = _furnace_expectation(n)[None, :] * np.ones((e.nb, 1))
            if use_mis and n > 1:  # (minus what the reference's pdf validity rule drops near the box's edges: _furnace_rule_loss)
                g = _furnace_rule_loss(_furnace_scene(e.W, e.H, n, use_mis), e.W, e.H, n)
                assert 0.001 < g[:, 1].mean() < 0.03
                want = LE * sum(RHO[None, :] ** k * (1.0 - g[:, k:k + 1]) for k in range(n))
            assert (e.se / want).max() < rel_se_bar or n == 1, "the estimate is too noisy for its pass to mean anything"
            if n == 1:  # the first vertex only: every pixel is exactly L, no randomness involved
                assert np.allclose(e.mean, LE, rtol=1e-6) and e.se.max() < 1e-6
                continue
            _assert_agree(_z(e.mean, e.se, want, 0.0), "white furnace, %d vertices, useMIS %s" % (n, use_mis))


def _check_quad_light(estimate, frames, rel_se_bar):
    est = {}
    for use_mis in (False, True):
        e = estimate(lambda W, H: _quad_light_scene(W, H, use_mis), frames)
        est[use_mis] = e
        want = _quad_light_expectation(_quad_light_scene(e.W, e.H, use_mis), e.W, e.H)
        assert np.median(e.se / want) < rel_se_bar, "the estimate is too noisy for its pass to mean anything"
        _assert_agree(_z(e.mean, e.se, want, 0.0, systematic=2e-3), "rectangular light over a diffuse floor against the form factor, useMIS %s" % use_mis)
    _assert_agree(_z(est[True].mean, est[True].se, est[False].mean, est[False].se), "rectangular light: NEE + MIS against BSDF sampling alone")
    # (what the NEE is for: at equal sample counts its estimate is the tighter one)
    assert np.median(est[True].se) < np.median(est[False].se)


def _check_cornell(estimate, frames, rel_se_bar):
    est = {}
    for use_mis in (False, True):
        est[use_mis] = estimate(lambda W, H: SH.cornell_scene(W, H, path_length=4, use_mis=use_mis), frames)
    a, b = est[True], est[False]
    lit = b.mean > 0.02  # (blocks that see the open front of the box are black in both)
    assert lit.mean() > 0.5
    assert np.median((b.se / np.maximum(b.mean, 1e-9))[lit]) < rel_se_bar, "the estimate is too noisy for its pass to mean anything"
    _assert_agree(_z(a.mean, a.se, b.mean, b.se)[lit], "Cornell box (configs[0]): useMIS on against off")
    assert np.median(a.se[lit]) < np.median(b.se[lit])


# ---- oracle twins (CPU) --------------------------------------------------------------------------------------------------

def _oracle_estimator(W, H):
    def estimate(make_scene, frames):
        e = _oracle_estimate(make_scene(W, H), W, H, frames)
        e.W, e.H = W, H
        return e
    return estimate


def test_oracle_white_furnace():
    _check_furnace(_oracle_estimator(32, 32), 96, 0.05)


def test_oracle_rectangular_light_matches_the_form_factor():
    _check_quad_light(_oracle_estimator(32, 32), 384, 0.08)


def test_oracle_cornell_mis_and_naive_converge_to_the_same_image():
    _check_cornell(_oracle_estimator(32, 32), 512, 0.15)


# ---- device (through the C-ABI) -------------------------------------------------------------------------------------------

def _gpu_estimator(gpu_ctx_factory, W, H):
    def estimate(make_scene, frames):
        ctx = gpu_ctx_factory(W, H)
        e = _gpu_estimate(ctx, make_scene(W, H), W, H, frames)
        e.W, e.H = W, H
        return e
    return estimate


@pytest.mark.gpu
def test_device_white_furnace(gpu_ctx_factory):
    _check_furnace(_gpu_estimator(gpu_ctx_factory, 64, 64), 1024, 0.01)


@pytest.mark.gpu
def test_device_rectangular_light_matches_the_form_factor(gpu_ctx_factory):
    _check_quad_light(_gpu_estimator(gpu_ctx_factory, 64, 64), 4096, 0.02)


@pytest.mark.gpu
def test_device_cornell_mis_and_naive_converge_to_the_same_image(gpu_ctx_factory):
    _check_cornell(_gpu_estimator(gpu_ctx_factory, 96, 96), 8192, 0.03)
