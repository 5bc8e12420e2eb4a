"""GPU parity, trace kernels: the HIP closest-hit / any-hit traversal (through the C-ABI) against the CPU oracle,
bit for bit, and against brute-force ground truth."""
import numpy as np
import pytest

from nexus_amd import pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH

pytestmark = pytest.mark.gpu


def _rays_for(scene, n, seed):
    a = scenegen.random_rays(n // 2, seed=seed, radius=5.0, target_extent=2.0)
    b = scenegen.interior_rays(n - n // 2, seed=seed + 1, extent=2.0)
    return np.concatenate([a, b])


@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene])
def test_closest_hit_bit_exact_vs_oracle(gpu_ctx_factory, make_scene):
    scene = make_scene()
    ctx = gpu_ctx_factory(256, 256)
    scene.upload(ctx)
    rays = _rays_for(scene, 50000, seed=11)
    got = ctx.trace_batch(rays)
    want = scene.oracle().trace_closest(rays)
    assert (want["hitDistance"] < 1e29).mean() > 0.05, "test scene must produce hits"
    assert SH.hit_records_equal(got, want), "GPU hit records differ from the oracle's"


def test_closest_hit_vs_brute_force(gpu_ctx_factory):
    scene = SH.instanced_scene(seed=5, n_inst=6)
    ctx = gpu_ctx_factory(256, 256)
    scene.upload(ctx)
    rays = _rays_for(scene, 4096, seed=21)
    got = ctx.trace_batch(rays)
    want = scene.oracle().brute_closest(rays)
    # brute force visits triangles in another order: distances must agree exactly, ids wherever the distance is unique
    assert np.array_equal(got["hitDistance"].view(np.uint32), want["hitDistance"].view(np.uint32))
    same = (got["triIdx"] == want["triIdx"]) & (got["instanceIdx"] == want["instanceIdx"])
    assert same.mean() > 0.999


def test_batch_larger_than_queue_capacity(gpu_ctx_factory):
    scene = SH.soup_scene(n=2000, seed=4)
    ctx = gpu_ctx_factory(64, 64)  # capacity 4096 rays: forces chunking
    scene.upload(ctx)
    rays = _rays_for(scene, 10000, seed=31)
    assert SH.hit_records_equal(ctx.trace_batch(rays), scene.oracle().trace_closest(rays))


def test_any_hit_vs_oracle_and_brute_force(gpu_ctx_factory):
    scene = SH.instanced_scene(seed=9, n_inst=10)
    ctx = gpu_ctx_factory(256, 256)
    scene.upload(ctx)
    rays = _rays_for(scene, 20000, seed=41)
    orc = scene.oracle()
    closest = orc.trace_closest(rays)
    rng = np.random.RandomState(0)
    # tmax just beyond / just short of the closest hit, and arbitrary
    tmax = np.where(closest["hitDistance"] < 1e29, closest["hitDistance"] * rng.choice([0.999, 1.001], len(rays)), 10.0).astype(np.float32)
    got = ctx.trace_shadow_batch(rays, tmax)
    assert np.array_equal(got, orc.trace_any(rays, tmax))
    sub = slice(0, 2000)
    assert np.array_equal(got[sub], orc.brute_any(rays[sub], tmax[sub]))


def test_trace_stats_match_oracle_counts(gpu_ctx_factory):
    scene = SH.instanced_scene(seed=2, n_inst=8)
    ctx = gpu_ctx_factory(128, 128)
    scene.upload(ctx)
    rays = _rays_for(scene, 8192, seed=51)
    ctx.enable_trace_stats(True)
    ctx.read_trace_stats(reset=True)
    ctx.trace_batch(rays)
    closest, _ = ctx.read_trace_stats(reset=True)
    ctx.enable_trace_stats(False)
    st = O.TraceStats()
    scene.oracle().trace_closest(rays, st)
    want = st.as_dict()
    assert closest["rays"] == len(rays)
    for k in ("nodes", "tris", "instances"):
        assert closest[k] == want[k], k


def test_empty_and_single_ray(gpu_ctx_factory):
    scene = SH.soup_scene(n=500, seed=8)
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    assert len(ctx.trace_batch(np.zeros(0, dtype=pod.RAY_DT))) == 0
    rays = _rays_for(scene, 2, seed=3)[:1]
    assert SH.hit_records_equal(ctx.trace_batch(rays), scene.oracle().trace_closest(rays))


def mixed_identity_scene():
    """Identity-placed and rotated / scaled instances side by side: the per-record identity flag is set for some records
    only, so the scene-wide flag is off and the kernel decides per lane."""
    from nexus_amd import capi

    meshes = [scenegen.random_soup(1200, seed=12, extent=0.7, size=0.09), scenegen.displaced_torus(40, 20, seed=12, major=0.6, minor=0.2)]
    return SH.BuiltScene(meshes, [(0, 0, SH.IDENTITY), (1, 0, SH.IDENTITY), (1, 0, capi.mat4_from_trs((0.9, 0.2, -0.4), (25, 50, 75), (0.8, 1.2, 1.0))),
                                  (0, 0, capi.mat4_from_trs((-0.7, -0.3, 0.5), (0, 90, 0))), (0, 0, capi.mat4_from_trs((0.5, 0.0, 0.0)))])


@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene, mixed_identity_scene])
def test_axis_aligned_rays_with_signed_zero_components(gpu_ctx_factory, make_scene):
    """Directions with exact +0 / -0 components (1/dir = +inf / -inf) through identity and rotated instances: the
    instance-entry shortcut for rays a transform leaves unchanged must not alter a single bit."""
    scene = make_scene()
    ctx = gpu_ctx_factory(128, 128)
    scene.upload(ctx)
    rng = np.random.RandomState(61)
    dirs = []
    for axis in range(3):
        for sgn in (1.0, -1.0):
            for z0 in (0.0, -0.0):
                for z1 in (0.0, -0.0):
                    d = [z0, z1]
                    d.insert(axis, sgn)
                    dirs.append(d)
    dirs = np.array(dirs, dtype=np.float32)
    n = 4096
    rays = np.zeros(n, dtype=pod.RAY_DT)
    rays["direction"] = dirs[rng.randint(0, len(dirs), n)]
    rays["origin"] = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    rays["origin"][: n // 4] = np.round(rays["origin"][: n // 4] * 4) / 4  # some origins on exact grid values
    # nearly axis-aligned: tiny and denormal components instead of exact zeros (1/dir huge or inf after the division)
    tiny = rays[: n // 2].copy()
    d = tiny["direction"].copy()
    d[d == 0] = rng.choice(np.array([1e-30, -1e-30, 1e-42, -1e-42, 3e-7, -3e-7], dtype=np.float32), size=int((d == 0).sum()))
    tiny["direction"] = d
    rays = np.concatenate([rays, tiny])
    got = ctx.trace_batch(rays)
    want = scene.oracle().trace_closest(rays)
    # exact zeros make the reference's slab test produce NaNs that cull the node (mirrored behaviour), so only the
    # nearly-aligned half is expected to hit anything
    assert (want["hitDistance"][n:] < 1e29).mean() > 0.02
    assert SH.hit_records_equal(got, want)
    # the any-hit variant shares the instance entry: same rays, tmax just beyond / just short of the closest hit
    tmax = np.where(want["hitDistance"] < 1e29, want["hitDistance"] * rng.choice([0.999, 1.001], len(rays)), 10.0).astype(np.float32)
    assert np.array_equal(ctx.trace_shadow_batch(rays, tmax), scene.oracle().trace_any(rays, tmax))


def test_identity_instances_in_a_mixed_scene_trace_like_the_oracle(gpu_ctx_factory):
    """Ordinary rays through a scene whose instances are partly identity-placed (flagged records skip the transform) and
    partly rotated: hit records and visit counts equal the oracle's."""
    scene = mixed_identity_scene()
    ctx = gpu_ctx_factory(128, 128)
    scene.upload(ctx)
    rays = _rays_for(scene, 30000, seed=91)
    ctx.enable_trace_stats(True)
    ctx.read_trace_stats(reset=True)
    got = ctx.trace_batch(rays)
    closest, _ = ctx.read_trace_stats(reset=True)
    ctx.enable_trace_stats(False)
    st = O.TraceStats()
    want = scene.oracle().trace_closest(rays, st)
    assert (want["hitDistance"] < 1e29).mean() > 0.05
    assert SH.hit_records_equal(got, want)
    for k in ("nodes", "tris", "instances"):
        assert closest[k] == st.as_dict()[k], k


def test_degenerate_geometry_planar_meshes_and_zero_area_triangles(gpu_ctx_factory):
    """Nodes with zero extent on an axis (every triangle in one plane: the frame exponent comes from log2(0)), zero-area
    triangles (1/det = inf, NaN barycentrics) and sliver triangles: same hits as the oracle, bit for bit, and the planar
    meshes are still hit."""
    rng = np.random.RandomState(71)
    # an axis-aligned planar grid (z = 0.25 exactly) and a tilted one
    g = np.linspace(-1, 1, 17, dtype=np.float32)
    quads = []
    for i in range(16):
        for j in range(16):
            p = lambda a, b: (g[a], g[b], np.float32(0.25))
            quads.append([p(i, j), p(i + 1, j), p(i + 1, j + 1)])
            quads.append([p(i, j), p(i + 1, j + 1), p(i, j + 1)])
    planar = pod.make_triangles(np.array(quads, dtype=np.float32))
    soup = scenegen.random_soup(400, seed=9, extent=0.8, size=0.1)
    bad = soup[:60].copy()
    bad["pos1"][:20] = bad["pos0"][:20]                                     # two equal vertices
    bad["pos2"][20:40] = bad["pos0"][20:40] * 0.5 + bad["pos1"][20:40] * 0.5  # three collinear vertices
    bad["pos2"][40:60] = bad["pos1"][40:60] + np.float32(1e-7)               # slivers
    mixed = np.concatenate([soup, bad])
    scene = SH.BuiltScene([planar, mixed], [(0, 0, SH.IDENTITY), (1, 0, SH.IDENTITY),
                                            (0, 0, __import__("nexus_amd").capi.mat4_from_trs((0.2, -0.3, 0.1), (35, 20, 10), (1.3, 0.7, 1.0)))])
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rays = _rays_for(scene, 40000, seed=81)
    got = ctx.trace_batch(rays)
    want = scene.oracle().trace_closest(rays)
    assert SH.hit_records_equal(got, want)
    planar_hits = (want["hitDistance"] < 1e29) & ((want["instanceIdx"] == 0) | (want["instanceIdx"] == 2))
    assert planar_hits.mean() > 0.05
    tmax = np.where(want["hitDistance"] < 1e29, want["hitDistance"] * 1.001, 10.0).astype(np.float32)
    assert np.array_equal(ctx.trace_shadow_batch(rays, tmax), scene.oracle().trace_any(rays, tmax))
