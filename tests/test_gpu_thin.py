"""The thin kernel (nx_trace.hip: the last long rays of a dry trace wave, searched by a whole wave in any order) must return the
reference's hit record — the one ITS visiting order finds — for every ray.  In the pass graph only a few rays per launch take that
route; here the test hook nxhip_debug_set_thin makes every wave hand over the first 64 rays it takes (64 lanes, 0 iterations) and lets the ray-batch hooks use the hand-over, so tens of thousands of arbitrary rays go through the cooperative
search and are compared with the oracle's records bit for bit: random and instanced scenes, rotated / scaled instances, rays along
the axes with signed zeros, and — the case the search's window exists for — rays through the shared edges and vertices of a
tessellated plane and through coincident triangles, where several triangles lie at exactly or nearly the same distance and the
result is whichever the reference's order meets first."""
import numpy as np
import pytest

from nexus_amd import capi, pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH
from tests.test_gpu_trace import _rays_for, mixed_identity_scene

pytestmark = pytest.mark.gpu


def _thin_ctx(gpu_ctx_factory, scene, size=256):
    ctx = gpu_ctx_factory(size, size)
    scene.upload(ctx)
    ctx.debug_set_thin(lanes=64, iters=0, in_hooks=True)
    return ctx


def _check_closest(ctx, scene, rays, min_handed=0.2):
    """min_handed: the share of the rays (of the last chunk of the batch) the thin kernel must have been handed — rays that end in
    their first iteration (a miss at the root) never reach the hand-over"""
    got = ctx.trace_batch(rays)
    handed = ctx.debug_thin_counts()[0]
    want = scene.oracle().trace_closest(rays)
    print("handed to the thin kernel: %d of %d rays" % (handed, len(rays)))
    assert handed >= min_handed * min(len(rays), 65536), "the thin kernel must have seen a good part of the rays (%d of %d)" % (handed, len(rays))
    assert SH.hit_records_equal(got, want), "hit records through the thin kernel differ from the oracle's"
    return got


@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene, mixed_identity_scene])
def test_closest_hit_through_the_thin_kernel_is_bit_exact(gpu_ctx_factory, make_scene):
    scene = make_scene()
    ctx = _thin_ctx(gpu_ctx_factory, scene)
    got = _check_closest(ctx, scene, _rays_for(scene, 40000, seed=17))
    assert (got["hitDistance"] < 1e29).mean() > 0.05
    # ... and with the product's rule (16 lanes, 16 iterations) the same records again
    ctx.debug_set_thin(lanes=16, iters=16, in_hooks=True)
    assert SH.hit_records_equal(ctx.trace_batch(_rays_for(scene, 40000, seed=17)), got)


@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene, mixed_identity_scene])
def test_rays_handed_over_in_the_middle_of_their_traversal_are_continued_bit_exactly(gpu_ctx_factory, make_scene):
    """Round 6: a handed-over ray brings its traversal state (stack, current groups, the hit found so far, the instance it is inside)
    and the search CONTINUES it.  The hook hands every ray over after exactly k loop iterations, k = 1 ... 40: rays at the root, rays
    in the TLAS, rays inside rotated instances with a hit in hand, rays one step from their end — closest hit and any hit — all must
    end with the oracle's records."""
    scene = make_scene()
    ctx = gpu_ctx_factory(256, 256)
    scene.upload(ctx)
    rays = _rays_for(scene, 30000, seed=31)
    orc = scene.oracle()
    want = orc.trace_closest(rays)
    rng = np.random.RandomState(3)
    tmax = np.where(want["hitDistance"] < 1e29, want["hitDistance"] * rng.choice([0.999, 1.001], len(rays)), 10.0).astype(np.float32)
    want_any = orc.trace_any(rays, tmax)
    for k in (1, 2, 3, 5, 8, 13, 25, 40):
        ctx.debug_set_thin(lanes=64, iters=k, in_hooks=True, any_time=True)
        got = ctx.trace_batch(rays)
        handed = ctx.debug_thin_counts()[0]
        assert SH.hit_records_equal(got, want), "closest hit, handed over after %d iterations" % k
        got_any = ctx.trace_shadow_batch(rays, tmax)
        handed_any = ctx.debug_thin_counts()[1]
        assert np.array_equal(got_any, want_any), "any hit, handed over after %d iterations" % k
        print("after %2d iterations: %5d closest-hit and %5d any-hit rays continued by the thin kernel" % (k, handed, handed_any))
        assert handed > (2000 if k <= 8 else 10) or make_scene is not SH.soup_scene
    # ... and a pool kept small: seeds and their children are put back and taken again
    ctx.debug_set_thin_pool(96)
    ctx.debug_set_thin(lanes=64, iters=6, in_hooks=True, any_time=True)
    assert SH.hit_records_equal(ctx.trace_batch(rays), want)
    assert np.array_equal(ctx.trace_shadow_batch(rays, tmax), want_any)
    ctx.debug_set_thin_pool(0)


@pytest.mark.parametrize("make_scene", [SH.soup_scene, SH.instanced_scene])
def test_a_pool_that_runs_over_puts_items_back_and_finds_the_same_records(gpu_ctx_factory, make_scene):
    """A round of the search whose children do not fit the wave's pool puts items back and goes on with fewer items per round; only a
    single item that cannot expand gives the ray to the in-order replay.  The product's pool (1 024 items) is not filled by these
    scenes, so the hook lowers the limit to 64 / 96 / 160 items: every wide node then takes that path, and the records stay the
    oracle's bit for bit — closest hit and any hit."""
    scene = make_scene()
    ctx = _thin_ctx(gpu_ctx_factory, scene)
    rays = _rays_for(scene, 30000, seed=23)
    for slots in (64, 96, 160):
        ctx.debug_set_thin_pool(slots)
        _check_closest(ctx, scene, rays)
    orc = scene.oracle()
    closest = orc.trace_closest(rays)
    tmax = np.where(closest["hitDistance"] < 1e29, closest["hitDistance"] * np.float32(1.001), 10.0).astype(np.float32)
    ctx.debug_set_thin_pool(64)
    assert np.array_equal(ctx.trace_shadow_batch(rays, tmax), orc.trace_any(rays, tmax))
    ctx.debug_set_thin_pool(0)


def test_any_hit_through_the_thin_kernel(gpu_ctx_factory):
    scene = SH.instanced_scene(seed=9, n_inst=10)
    ctx = _thin_ctx(gpu_ctx_factory, scene)
    rays = _rays_for(scene, 30000, seed=43)
    orc = scene.oracle()
    closest = orc.trace_closest(rays)
    rng = np.random.RandomState(1)
    tmax = np.where(closest["hitDistance"] < 1e29, closest["hitDistance"] * rng.choice([0.999, 1.001], len(rays)), 10.0).astype(np.float32)
    got = ctx.trace_shadow_batch(rays, tmax)
    assert ctx.debug_thin_counts()[1] > 5000
    assert np.array_equal(got, orc.trace_any(rays, tmax))


def _grid_plane(n, y=0.0, seed=None, jitter=0.0):
    """2 n^2 triangles tiling [-1, 1]^2 at height y: every interior vertex is shared by six triangles, every interior edge by two"""
    g = np.linspace(-1.0, 1.0, n + 1).astype(np.float32)
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (g[i], y, g[j]), (g[i + 1], y, g[j]), (g[i + 1], y, g[j + 1]), (g[i], y, g[j + 1])
            tris.append((a, b, c))
            tris.append((a, c, d))
    return pod.make_triangles(np.asarray(tris, np.float32))


def test_rays_through_shared_edges_vertices_and_coincident_triangles(gpu_ctx_factory):
    """Several triangles at exactly (coincident copies, shared edges hit head-on) or nearly (a second sheet 1e-7 above) the closest
    distance: the search must notice and fall back to the reference's own order."""
    n = 24
    sheet = _grid_plane(n, 0.0)
    twin = _grid_plane(n, 0.0)                      # an exact copy: every hit has a partner at the same distance
    near = _grid_plane(n, np.float32(1.0e-7))       # a sheet within rounding of the first
    far = _grid_plane(n, -0.5)
    scene = SH.BuiltScene([sheet, twin, near, far], [(0, 0, SH.IDENTITY), (1, 0, SH.IDENTITY), (2, 0, SH.IDENTITY), (3, 0, SH.IDENTITY),
                                                      (0, 0, capi.mat4_from_trs((0.0, 0.25, 0.0), (0, 30, 0), (0.5, 1.0, 0.5)))])
    ctx = _thin_ctx(gpu_ctx_factory, scene)
    g = np.linspace(-1.0, 1.0, n + 1)
    rng = np.random.RandomState(5)
    pts = []
    for _ in range(6000):  # grid vertices, points on grid edges, on diagonals, and arbitrary ones
        kind = rng.randint(4)
        i, j = rng.randint(1, n), rng.randint(1, n)
        if kind == 0:
            pts.append((g[i], g[j]))
        elif kind == 1:
            pts.append((g[i], rng.uniform(-1, 1)))
        elif kind == 2:
            t = rng.uniform(0, 1)
            pts.append((g[i] + t * (g[i + 1] - g[i]) if i < n else g[i], g[j] + t * (g[j + 1] - g[j]) if j < n else g[j]))
        else:
            pts.append((rng.uniform(-1, 1), rng.uniform(-1, 1)))
    pts = np.asarray(pts, np.float64)
    rays = np.zeros(2 * len(pts), dtype=pod.RAY_DT)
    # straight down onto the points, and slanted rays through them
    rays["origin"][: len(pts)] = np.stack([pts[:, 0], np.full(len(pts), 1.0), pts[:, 1]], 1).astype(np.float32)
    rays["direction"][: len(pts)] = (0.0, -1.0, 0.0)
    eye = np.array([0.3, 1.7, -0.4])
    target = np.stack([pts[:, 0], np.zeros(len(pts)), pts[:, 1]], 1)
    d = target - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["origin"][len(pts):] = eye.astype(np.float32)
    rays["direction"][len(pts):] = d.astype(np.float32)
    got = _check_closest(ctx, scene, rays, min_handed=0.3)
    # (the rays that come straight down onto a flat sheet miss it in the reference's arithmetic — 0 * inf in the slab test of a box
    #  without height — and so they do here; the slanted ones hit)
    assert (got["hitDistance"][len(pts):] < 1e29).mean() > 0.9


def test_axis_aligned_rays_through_the_thin_kernel(gpu_ctx_factory):
    scene = mixed_identity_scene()
    ctx = _thin_ctx(gpu_ctx_factory, scene, 128)
    rng = np.random.RandomState(61)
    dirs = []
    for axis in range(3):
        for sgn in (1.0, -1.0):
            for z0 in (0.0, -0.0):
                for z1 in (0.0, -0.0):
                    d = [z0, z1]
                    d.insert(axis, sgn)
                    dirs.append(d)
    dirs = np.asarray(dirs, np.float32)
    rays = np.zeros(6000, dtype=pod.RAY_DT)
    rays["direction"] = dirs[rng.randint(0, len(dirs), len(rays))]
    rays["origin"] = rng.uniform(-1.2, 1.2, (len(rays), 3)).astype(np.float32) - 3.0 * rays["direction"]
    _check_closest(ctx, scene, rays, min_handed=0.005)  # (most of these rays miss the scene's root box and end at once)


def test_whole_frames_with_every_ray_handed_over_mid_traversal(gpu_ctx_factory):
    """The pass graph's own launches (not the ray-batch hooks): with the any-time hook every ray of every level — primary rays that
    started from an ENTRY STATE, continuation rays, shadow rays — is handed over after k iterations and continued by the thin kernel
    of its level, as long as the level's lists have room.  Frames, accumulation and queue sizes must equal the run without any
    hand-over, in the SCAN pipeline (fast compaction) and in the classic one (ordered compaction, reference random numbers)."""
    from nexus_amd import multigpu, workloads

    W, H = 256, 144
    scene = workloads.config2(W, H, 160, 80, 5, cls=SH.BuiltScene)
    for modes, entry in (((pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED), True),
                         ((pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED), False),
                         ((pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE), False)):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(*modes)
        ctx.set_pixel_order(pod.ORDER_TILES)
        ctx.set_entry_points(entry)
        ctx.set_tail_bounce(0)
        ctx.set_frames_per_pass(2)

        def run():
            ctx.reset_frame_number()
            for _ in range(2):
                ctx.render_frame()
                ctx.accumulate()
            return ctx.read_radiance().view(np.uint32).copy(), ctx.read_accumulation().view(np.uint32).copy(), ctx.read_queue_sizes()

        ctx.debug_set_thin(lanes=64, iters=1 << 20, in_hooks=False)  # (no wave ever qualifies: no hand-over at all)
        want = run()
        for k in (1, 4, 11):
            ctx.debug_set_thin(lanes=64, iters=k, in_hooks=False, any_time=True)
            got = run()
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (modes, entry, k)
            assert SH.queue_sizes_identical(got[2], want[2]), (modes, entry, k)
            handed = [ctx.debug_thin_counts_of_pass(b) for b in range(0, 4)]
            print("%s entry %s k %d: handed over per level (closest, any) %s" % (modes, entry, k, handed))
            assert handed[0][0] > 10000 and handed[1][0] > 1000 and handed[1][1] > 100, "the levels' rays did go through the thin kernel (up to the lists' 32 768 entries each)"
        ctx.sync()
