"""GPU parity of the shading building blocks, one function at a time through the C-ABI test hooks: the four BSDFs'
Sample / Eval and the software texture fetch against the oracle's, on seeded grids.

No tolerance: these functions use sqrt / division (exact on both sides) and sin cos exp log — since round 4 the shared text of
include/nexus_fmath.h on both sides — so the accept / reject decision, the xorshift state after a Sample and every value of an
accepted sample are compared as bit patterns (a NaN matches a NaN: the two instruction sets give it different sign bits)."""
import ctypes as C

import numpy as np
import pytest

from nexus_amd import pod
from tests import oracle_lib as O
from tests import scene_helpers as SH

pytestmark = pytest.mark.gpu

MATERIALS = {
    "diffuse": pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.5, 0.3)),
    "plastic": pod.make_material(pod.MAT_PLASTIC, albedo=(0.8, 0.3, 0.2), roughness=0.4, ior=1.5),
    "plastic_smooth": pod.make_material(pod.MAT_PLASTIC, albedo=(0.2, 0.3, 0.9), roughness=0.05, ior=1.33),
    "dielectric": pod.make_material(pod.MAT_DIELECTRIC, albedo=(0.95, 0.97, 1.0), roughness=0.2, ior=1.45),
    "dielectric_rough": pod.make_material(pod.MAT_DIELECTRIC, albedo=(1.0, 1.0, 1.0), roughness=0.7, ior=1.8),
    "conductor": pod.make_material(pod.MAT_CONDUCTOR, roughness=0.3, conductor_ior=(0.2, 0.9, 1.1), conductor_k=(3.9, 2.4, 2.2)),
}


def _unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _queries(n, seed, both_sides):
    rng = np.random.RandomState(seed)
    q = np.zeros(n, dtype=pod.BSDF_QUERY_DT)
    wi = _unit(rng.normal(size=(n, 3)))
    wo = _unit(rng.normal(size=(n, 3)))
    if not both_sides:
        wi[:, 2] = np.abs(wi[:, 2])
    # a share of grazing and near-normal directions
    wi[: n // 8, 2] *= 0.02
    wi[n // 8: n // 4, :2] *= 0.01
    q["wi"] = _unit(wi).astype(np.float32)
    q["wo"] = wo.astype(np.float32)
    q["rng"] = rng.randint(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    return q


def _oracle_sample(mat, q):
    L = O.lib()
    m = np.array([mat], dtype=pod.MAT_DT)
    out = np.zeros(len(q), dtype=pod.BSDF_RESULT_DT)
    wo, thr, pdf = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float()
    for k in range(len(q)):
        s = C.c_uint32(int(q["rng"][k]))
        wi = np.ascontiguousarray(q["wi"][k])
        ok = L.orc_bsdf_sample(O._ptr(m), O._ptr(wi), C.byref(s), O._ptr(wo), O._ptr(thr), C.byref(pdf))
        out["ok"][k], out["wo"][k], out["throughput"][k], out["pdf"][k], out["rngOut"][k] = ok, wo, thr, pdf.value, s.value
    return out


def _oracle_eval(mat, q):
    L = O.lib()
    m = np.array([mat], dtype=pod.MAT_DT)
    out = np.zeros(len(q), dtype=pod.BSDF_RESULT_DT)
    thr, pdf = np.zeros(3, np.float32), C.c_float()
    for k in range(len(q)):
        wi, wo = np.ascontiguousarray(q["wi"][k]), np.ascontiguousarray(q["wo"][k])
        ok = L.orc_bsdf_eval(O._ptr(m), O._ptr(wi), O._ptr(wo), O._ptr(thr), C.byref(pdf))
        out["ok"][k], out["throughput"][k], out["pdf"][k] = ok, thr, pdf.value
    return out


def _compare(got, want, fields, name):
    assert np.array_equal(got["ok"], want["ok"]), (name, "accept/reject decisions differ", int((got["ok"] != want["ok"]).sum()))
    both = want["ok"] == 1
    assert both.sum() > 0.2 * len(got), (name, "too few accepted samples to compare")
    for f in fields:
        a, b = np.ascontiguousarray(got[f][both]), np.ascontiguousarray(want[f][both])
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (name, f, "%d of %d values differ" % (int((~same).sum()), same.size), a[~same][:4], b[~same][:4])


@pytest.mark.parametrize("name", sorted(MATERIALS))
def test_bsdf_sample_matches_oracle(gpu_ctx_factory, name):
    mat = MATERIALS[name]
    ctx = gpu_ctx_factory(16, 16)
    q = _queries(6000, seed=sorted(MATERIALS).index(name), both_sides=name.startswith("dielectric"))
    got = ctx.bsdf_sample_batch(mat, q)
    want = _oracle_sample(mat, q)
    _compare(got, want, ("wo", "throughput", "pdf"), name)
    # the random stream itself is integer arithmetic: identical wherever both sides took the same branches
    assert np.array_equal(got["rngOut"], want["rngOut"])


@pytest.mark.parametrize("name", sorted(MATERIALS))
def test_bsdf_eval_matches_oracle(gpu_ctx_factory, name):
    mat = MATERIALS[name]
    ctx = gpu_ctx_factory(16, 16)
    q = _queries(6000, seed=100 + sorted(MATERIALS).index(name), both_sides=name.startswith("dielectric"))
    # outgoing directions the BSDF can actually produce (a random pair has a vanishing pdf for the glossy lobes): the
    # oracle's own samples for the first two thirds, random directions for the rest
    smp = _oracle_sample(mat, q)
    use = (smp["ok"] == 1) & (np.arange(len(q)) < 4000)
    q["wo"][use] = smp["wo"][use]
    if not name.startswith("dielectric"):
        q["wo"][~use, 2] = np.abs(q["wo"][~use, 2])
    got = ctx.bsdf_eval_batch(mat, q)
    want = _oracle_eval(mat, q)
    _compare(got, want, ("throughput", "pdf"), name)


def test_texture_fetch_matches_oracle(gpu_ctx_factory):
    ctx = gpu_ctx_factory(16, 16)
    maps = {"diffuse": SH.checker_texture(64, 32, 1, alpha=True), "emissive": SH.checker_texture(31, 17, 2), "hdr": SH.checker_texture(128, 64, 3)}
    ids = {k: ctx.upload_texture(k, img) for k, img in maps.items()}
    rng = np.random.RandomState(9)
    uv = rng.uniform(-2.5, 3.5, size=(20000, 2)).astype(np.float32)   # wrap addressing on both sides of [0, 1)
    uv[:200] = np.round(uv[:200] * 64) / 64                            # texel centres / edges
    for kind, img in maps.items():
        got = ctx.tex2d_batch(kind, ids[kind] if kind != "hdr" else 0, uv)
        desc = O._TexDesc(img.shape[1], img.shape[0], O._ptr(np.ascontiguousarray(img)))
        want = np.zeros_like(got)
        out = np.zeros(4, np.float32)
        for k in range(len(uv)):
            O.lib().orc_tex2d(C.byref(desc), float(uv[k, 0]), float(uv[k, 1]), O._ptr(out))
            want[k] = out
        # same operation order on both sides, sRGB decode through a 256-entry table: exact
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), kind
