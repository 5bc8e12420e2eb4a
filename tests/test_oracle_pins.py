"""Pins for the CPU oracle.  The reference ships no tests or golden vectors and cannot be built here, so the
oracle is pinned by ground truth that does not depend on its own algorithm: brute-force intersection, independent
re-computation of the published hash / RNG definitions, analytic identities and structural invariants."""
import ctypes as C

import numpy as np
import pytest

from nexus_amd import capi, pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH


# ---- RNG: independent Python restatement of Bob Jenkins' one-at-a-time mix and Marsaglia's xorshift32 13/17/5 ----------

def _jenkins(x):
    M = 0xFFFFFFFF
    x = (x + (x << 10)) & M
    x ^= x >> 6
    x = (x + (x << 3)) & M
    x ^= x >> 11
    x = (x + (x << 15)) & M
    return x


def _xorshift(s):
    M = 0xFFFFFFFF
    s ^= (s << 13) & M
    s ^= s >> 17
    s ^= (s << 5) & M
    return s & M


def test_jenkins_and_seeding_known_answers():
    L = O.lib()
    for x in (0, 1, 2, 12345, 0xDEADBEEF, 0xFFFFFFFF):
        assert L.orc_jenkins(x) == _jenkins(x)
    # InitRNG(pixel, res, frame) = jenkins(max(1, (px + py*resX) ^ jenkins(frame)))  — Random.cuh:71-77
    for px, py, res, frame in [(0, 0, 512, 1), (17, 300, 1920, 7), (511, 511, 512, 4)]:
        s = ((px + py * res) & 0xFFFFFFFF) ^ _jenkins(frame)
        assert L.orc_rng_init_pixel(px, py, res, frame) == _jenkins(s if s else 1)
    # the slot variant is the pixel variant at (1, index)  — Random.cuh:79-82
    assert L.orc_rng_init_index(99, 640, 3) == L.orc_rng_init_pixel(1, 99, 640, 3)
    # a zero pre-hash state is replaced by 1
    f = 5
    zero_px = _jenkins(f)  # px ^ jenkins(f) == 0
    assert L.orc_rng_init_pixel(zero_px, 0, 1 << 20, f) == _jenkins(1)


def test_rand_stream_matches_xorshift_definition():
    L = O.lib()
    state = C.c_uint32(0x12345678)
    s = 0x12345678
    for _ in range(200):
        v = L.orc_rand(C.byref(state))
        s = _xorshift(s)
        assert state.value == s
        expect = np.array([0x3F800000 | (s >> 9)], dtype=np.uint32).view(np.float32)[0] - np.float32(1.0)
        assert np.float32(v) == expect and 0.0 <= v < 1.0


# ---- traversal against brute force ------------------------------------------------------------------------------------

@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene, lambda: SH.cornell_scene(64, 64)])
def test_bvh8_traversal_equals_brute_force(make_scene):
    scene = make_scene()
    orc = scene.oracle()
    rays = np.concatenate([scenegen.random_rays(3000, seed=5, radius=5.0, target_extent=2.0), scenegen.interior_rays(3000, seed=6, extent=1.5)])
    st = O.TraceStats()
    got = orc.trace_closest(rays, st)
    want = orc.brute_closest(rays)
    assert (want["hitDistance"] < 1e29).mean() > 0.05
    assert np.array_equal(got["hitDistance"].view(np.uint32), want["hitDistance"].view(np.uint32))
    same = (got["triIdx"] == want["triIdx"]) & (got["instanceIdx"] == want["instanceIdx"])
    assert same.mean() > 0.999  # exact ties between two triangles may resolve to either id
    assert st.maxStack < 32, "the reference's stack (TRAVERSAL_STACK_SIZE 32) must suffice"
    tmax = np.where(want["hitDistance"] < 1e29, want["hitDistance"] * np.float32(1.0005), 3.0).astype(np.float32)
    assert np.array_equal(orc.trace_any(rays, tmax), orc.brute_any(rays, tmax))
    tmax2 = (want["hitDistance"] * np.float32(0.9995)).astype(np.float32)
    assert np.array_equal(orc.trace_any(rays, tmax2), orc.brute_any(rays, tmax2))


def test_multithreaded_trace_is_identical():
    scene = SH.instanced_scene(seed=4, n_inst=5)
    orc = scene.oracle()
    rays = scenegen.random_rays(20000, seed=9, radius=5.0, target_extent=2.0)
    assert SH.hit_records_equal(orc.trace_closest(rays), orc.trace_closest(rays, threads=4))


def test_bvh2_traversal_equals_brute_force():
    tris = scenegen.displaced_torus(32, 16, seed=3)
    nodes, idx = O.bvh2_build(tris)
    b = O._Bvh2()
    assert O.lib().orc_bvh2_build(O._ptr(tris), len(tris), C.byref(b)) == 0
    rays = scenegen.random_rays(4000, seed=2, radius=4.0, target_extent=1.2)
    hits = np.zeros(len(rays), dtype=pod.HIT_DT)
    O.lib().orc_bvh2_trace_closest(C.byref(b), O._ptr(tris), O._ptr(rays), len(rays), O._ptr(hits))
    O.lib().orc_bvh2_free(C.byref(b))
    scene = SH.BuiltScene([tris], [(0, 0, SH.IDENTITY)])
    want = scene.oracle().brute_closest(rays)
    assert np.array_equal(hits["hitDistance"].view(np.uint32), want["hitDistance"].view(np.uint32))
    assert (hits["triIdx"] == want["triIdx"]).mean() > 0.999


def test_child_trace_mask_layout():
    """One hand-built node: inner child in slot 2, a 2-triangle leaf in slot 5; ray along +x through both boxes."""
    node = np.zeros(1, dtype=pod.NODE_DT)
    node["p"] = (0, 0, 0)
    node["e"] = (127, 127, 127)  # scale 2^0
    node["imask"] = 1 << 2
    node["childBaseIdx"] = 7
    node["triangleBaseIdx"] = 40
    node["meta"][0][2] = 0x20 | (24 + 2)
    node["meta"][0][5] = 0b01100000 | 3  # two triangles at offset 3
    for s, (lo, hi) in {2: (10, 20), 5: (30, 40)}.items():
        for ax in ("x", "y", "z"):
            node["qlo" + ax][0][s] = lo if ax == "x" else 0
            node["qhi" + ax][0][s] = hi if ax == "x" else 50
    out = np.zeros(4, dtype=np.uint32)
    org = np.array([-5.0, 25.0, 25.0], np.float32)
    d = np.array([1.0, 0.25, 0.125], np.float32)  # all positive: octant 0, invOctant 7
    O.lib().orc_child_trace(O._ptr(node), O._ptr(org), O._ptr(d), np.float32(1e30), O._ptr(out))
    assert out[0] == 7 and out[2] == 40
    assert out[1] & 0xFF == 1 << 2                       # imask rides in the low byte
    assert (out[1] >> 24) == 1 << (2 ^ 7)                # inner hit at bit 24 + (slot ^ invOctant)
    assert out[3] == 0b11 << 3                           # unary count 2 at the leaf's offset
    far = np.zeros(4, dtype=np.uint32)
    O.lib().orc_child_trace(O._ptr(node), O._ptr(org), O._ptr(d), np.float32(12.0), O._ptr(far))
    assert far[1] >> 24 == 0 and far[3] == 0             # both boxes start beyond tmax = 12 (x in [10,20] is 15..25 away)


# ---- BSDF identities --------------------------------------------------------------------------------------------------

def _sample(mat, wi, seed):
    rng = C.c_uint32(seed)
    wo, thr, pdf = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float()
    m = np.array([mat], dtype=pod.MAT_DT)
    ok = O.lib().orc_bsdf_sample(O._ptr(m), O._ptr(np.asarray(wi, np.float32)), C.byref(rng), O._ptr(wo), O._ptr(thr), C.byref(pdf))
    return ok, wo, thr, pdf.value


def _eval(mat, wi, wo):
    thr, pdf = np.zeros(3, np.float32), C.c_float()
    m = np.array([mat], dtype=pod.MAT_DT)
    ok = O.lib().orc_bsdf_eval(O._ptr(m), O._ptr(np.asarray(wi, np.float32)), O._ptr(np.asarray(wo, np.float32)), O._ptr(thr), C.byref(pdf))
    return ok, thr, pdf.value


def test_lambert_sample_and_eval_agree():
    mat = pod.make_material(pod.MAT_DIFFUSE, albedo=(0.2, 0.5, 0.8))
    wi = np.array([0.3, -0.2, 0.93], np.float32)
    zs = []
    for seed in range(1, 400):
        ok, wo, thr, pdf = _sample(mat, wi, seed * 7919)
        if not ok:
            continue
        assert abs(np.linalg.norm(wo) - 1) < 1e-5 and wo[2] > 0
        assert np.allclose(thr, (0.2, 0.5, 0.8)) and abs(pdf - wo[2] / np.pi) < 1e-6
        ok2, thr2, pdf2 = _eval(mat, wi, wo)
        assert ok2 and abs(pdf2 - pdf) < 1e-6 and np.allclose(thr2 / pdf2, thr, rtol=1e-5)  # f cos / pdf == sample weight
        zs.append(wo[2])
    assert abs(np.mean(zs) - 2.0 / 3.0) < 0.05  # E[cos] of a cosine-weighted hemisphere


@pytest.mark.parametrize("mtype", [pod.MAT_PLASTIC, pod.MAT_DIELECTRIC, pod.MAT_CONDUCTOR])
def test_microfacet_samples_are_well_formed(mtype):
    mat = pod.make_material(mtype, albedo=(0.9, 0.9, 0.9), roughness=0.4, ior=1.5)
    wi = np.array([0.4, 0.1, 0.91], np.float32)
    wi /= np.linalg.norm(wi)
    n_ok = 0
    for seed in range(1, 600):
        ok, wo, thr, pdf = _sample(mat, wi, seed * 104729)
        if not ok:
            continue
        n_ok += 1
        assert np.all(np.isfinite(wo)) and np.all(np.isfinite(thr)) and np.isfinite(pdf) and pdf > 0
        assert np.all(thr >= 0)
        if mtype != pod.MAT_DIELECTRIC:
            assert wo[2] * wi[2] >= 0  # reflection only
            if mtype == pod.MAT_CONDUCTOR:
                ok2, thr2, pdf2 = _eval(mat, wi, wo / np.linalg.norm(wo))
                if ok2:  # the extension's Eval is the lobe the sampler draws from: same pdf
                    assert abs(pdf2 - pdf) <= 2e-3 * max(1.0, pdf)
    assert n_ok > 300


def test_tonemap_known_answers():
    L = O.lib()
    for rgb in [(0, 0, 0), (0.18, 0.18, 0.18), (1, 0.5, 0.25), (10, 10, 10), (0.01, 2.0, 0.3)]:
        v = np.asarray(rgb, np.float32)
        got = L.orc_tonemap_rgba8(O._ptr(v))
        x = np.asarray(rgb, np.float64) * 0.6
        y = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1) ** 0.45454545454
        want = np.floor(np.clip(y, 0, 1) * 255.0 + 1e-9).astype(int)
        for c in range(3):
            assert abs(((got >> (8 * c)) & 0xFF) - want[c]) <= 1
        assert got >> 24 == 255


def test_texture_fetch_wrap_bilinear_srgb():
    img = np.zeros((2, 2, 4), np.uint8)
    img[0, 0] = (255, 0, 0, 255)
    img[0, 1] = (0, 255, 0, 128)
    img[1, 0] = (0, 0, 255, 255)
    img[1, 1] = (255, 255, 255, 0)
    desc = O._TexDesc(2, 2, O._ptr(img))
    out = np.zeros(4, np.float32)
    O.lib().orc_tex2d(C.byref(desc), 0.25, 0.25, O._ptr(out))  # centre of texel (0,0)
    assert np.allclose(out, (1, 0, 0, 1), atol=1e-6)
    O.lib().orc_tex2d(C.byref(desc), 0.5, 0.25, O._ptr(out))   # halfway between texels (0,0) and (1,0): linear in decoded space
    assert np.allclose(out, (0.5, 0.5, 0, (255 + 128) / 510.0), atol=2e-3)
    O.lib().orc_tex2d(C.byref(desc), 1.25, -0.75, O._ptr(out))  # wrap
    assert np.allclose(out, (1, 0, 0, 1), atol=1e-6)
    mid = np.zeros((1, 1, 4), np.uint8)
    mid[0, 0] = (188, 188, 188, 255)
    d1 = O._TexDesc(1, 1, O._ptr(mid))
    O.lib().orc_tex2d(C.byref(d1), 0.3, 0.8, O._ptr(out))
    assert abs(out[0] - 0.5029) < 2e-3  # sRGB 188 -> linear ~0.503


# ---- frame-level invariants -------------------------------------------------------------------------------------------

def test_cornell_frame_invariants():
    W = H = 64
    scene = SH.cornell_scene(W, H, path_length=4)
    w = O.Wavefront(scene.oracle(), W * H)
    w.render(1)
    q = w.queue_sizes()
    assert q["traceSize"][0] == W * H
    for b in range(1, 5):
        assert q["diffuseSize"][b] <= q["traceSize"][b - 1]            # logic only forwards hits that survive roulette
        assert q["traceSize"][b] <= q["diffuseSize"][b]                # shade emits at most one continuation per request
        assert q["traceShadowSize"][b] <= q["diffuseSize"][b]
        assert q["plasticSize"][b] == q["dielectricSize"][b] == q["conductorSize"][b] == 0
    assert q["traceSize"][4] == 0                                      # bounce == pathLength stops (PathTracer.cu:392)
    rad = w.radiance()
    assert np.all(np.isfinite(rad)) and rad.min() >= 0 and 0.05 < rad.mean() < 5
    # the running mean of identical frames is the frame
    w.accumulate(1)
    assert np.array_equal(w.accumulation(), rad)
    # different frame numbers give different samples, same statistics
    w.render(2)
    assert not np.array_equal(w.radiance(), rad)
