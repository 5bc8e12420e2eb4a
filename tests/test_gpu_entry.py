"""Entry points of the primary rays (nxhip_set_entry_points, nx_entry.hip): the traversal of a run of 64 primary paths starts from
the state its first node steps provably share instead of the TLAS root.  The bar: every frame equals the frame without them bit
for bit — and the oracle's — while the primary level visits fewer nodes."""
import numpy as np
import pytest

from nexus_amd import capi, multigpu, pod, scenegen, workloads
from tests import oracle_lib as O
from tests import scene_helpers as SH

pytestmark = pytest.mark.gpu


def _torus_scene(W, H, nu=192, nv=96, cls=SH.BuiltScene):
    return workloads.config2(W, H, nu, nv, 5, cls=cls)


def _frames(ctx, n, per_pass=1):
    ctx.set_frames_per_pass(per_pass)
    ctx.reset_frame_number()
    out = []
    for _ in range(n):
        ctx.render_frame()
        ctx.accumulate()
        out.append(ctx.read_radiance())
    return out, ctx.read_accumulation(), ctx.read_queue_sizes()


def _primary_nodes_per_ray(ctx):
    ctx.enable_trace_stats(True)
    ctx.read_trace_stats(reset=True)
    ctx.set_tail_bounce(0)
    ctx.reset_frame_number()
    ctx.render_frame()
    closest, _ = ctx.read_trace_stats(reset=True)
    ctx.enable_trace_stats(False)
    return closest["nodes"] / max(1, closest["rays"])


@pytest.mark.parametrize("order", ["tiles", "rows", "rank 1 of 3"])
def test_identity_scene_frames_are_unchanged_and_fewer_nodes_are_visited(gpu_ctx_factory, order):
    W, H = 256, 144
    scene = _torus_scene(W, H)
    pm = np.arange(W * H, dtype=np.uint32)
    if order == "tiles":
        pm = multigpu.tiled_order(pm, W)
    elif order == "rank 1 of 3":
        pm = multigpu.tiled_order(multigpu.tile_pixel_map(W, H, 1, 3, 8), W)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ctx.set_pixel_map(pm)
    ctx.set_tail_bounce(0)  # (the queue sizes of every bounce are compared below)
    base, base_acc, base_q = _frames(ctx, 3, per_pass=2)
    nodes_off = _primary_nodes_per_ray(ctx)
    ctx.set_entry_points(True)
    got, got_acc, got_q = _frames(ctx, 3, per_pass=2)
    nodes_on = _primary_nodes_per_ray(ctx)
    states = ctx.read_entry_states()
    assert len(states) == (len(pm) + 63) // 64 and (states[:, 19] >= 1).mean() > 0.4, "a good part of the runs must start below the root"
    for f in range(3):
        assert np.array_equal(got[f].view(np.uint32), base[f].view(np.uint32)), "frame pass %d" % f
    assert np.array_equal(got_acc.view(np.uint32), base_acc.view(np.uint32))
    assert SH.queue_sizes_identical(got_q, base_q)
    print("nodes per ray over one frame's launches: %.2f without, %.2f with entry points (%s)" % (nodes_off, nodes_on, order))
    assert nodes_on < nodes_off - 0.1
    # ... and the oracle's frame (which starts every ray at the root)
    w = O.Wavefront(scene.oracle(), len(pm), pm, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED)
    w.render(1, threads=8)
    ctx.set_frames_per_pass(1)
    ctx.reset_frame_number()
    ctx.render_frame()
    assert SH.frames_identical(ctx.read_radiance(), w.radiance(), "against the oracle")
    # (switching them off again restores the root start)
    on_one = _primary_nodes_per_ray(ctx)
    ctx.set_entry_points(False)
    assert _primary_nodes_per_ray(ctx) > on_one + 0.1 and len(ctx.read_entry_states()) == 0


def test_transformed_instances_lens_and_reference_modes_are_unchanged(gpu_ctx_factory):
    """Rotated instances (the walk stops in front of them), a camera with a lens (no entry states at all), slot-keyed random numbers
    with ordered compaction (the classic pipeline): same frames with and without."""
    for make, modes in ((lambda: SH.cornell_scene(160, 160, path_length=4), (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)),
                        (lambda: SH.material_zoo_scene(96, 64, path_length=4), (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)),
                        (lambda: SH.cornell_scene(160, 160, path_length=4), (pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE))):
        scene = make()
        W, H = int(scene.camera["resolution"][0]), int(scene.camera["resolution"][1])
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(*modes)
        ctx.set_tail_bounce(0)
        base, base_acc, base_q = _frames(ctx, 2)
        ctx.set_entry_points(True)
        got, got_acc, got_q = _frames(ctx, 2)
        for f in range(2):
            assert np.array_equal(got[f].view(np.uint32), base[f].view(np.uint32))
        assert np.array_equal(got_acc.view(np.uint32), base_acc.view(np.uint32)) and SH.queue_sizes_identical(got_q, base_q)


def test_a_camera_inside_the_geometry_and_axis_parallel_views(gpu_ctx_factory):
    """Bundles that straddle an octant boundary (a view straight down an axis), a camera inside the root box, a camera whose rays
    all miss: the walk must stop or conclude exactly what every ray concludes."""
    W, H = 128, 128
    torus = scenegen.displaced_torus(96, 48, seed=3, major=1.0, minor=0.45, amp=0.05, center=(0.0, 0.5, 0.0))
    floor = scenegen.quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6))
    light = scenegen.quad((-1.2, 4.0, -1.2), (1.2, 4.0, -1.2), (1.2, 4.0, 1.2), (-1.2, 4.0, 1.2))
    mats = np.array([pod.make_material(pod.MAT_DIFFUSE, albedo=(0.6, 0.5, 0.4)), pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.7, 0.7)),
                     pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 1.0, 1.0), intensity=20.0)], dtype=pod.MAT_DT)
    views = [((0.0, 3.0, 0.0), (0.0, -1.0, 0.0)),        # straight down: every bundle near the centre straddles two octants
             ((0.3, 0.5, 0.2), (1.0, 0.05, 0.1)),        # inside the torus's root box
             ((0.0, 3.0, 9.0), (0.0, 0.3, 1.0)),         # looking away: every ray misses
             ((5.0, 0.4, 0.0), (-1.0, 0.0, 0.0))]        # along -x exactly
    for eye, fwd in views:
        f = np.asarray(fwd, np.float64)
        cam = capi.camera_init(eye, f / np.linalg.norm(f), 60.0, W, H, 5.0, 0.0)
        scene = SH.BuiltScene([torus, floor, light], [(i, i, workloads.IDENTITY) for i in range(3)], materials=mats, camera=cam,
                              settings=workloads.make_settings(use_mis=True, path_length=3, background=(0.2, 0.3, 0.4), background_intensity=1.0))
        scene.lights = SH.mesh_lights(scene.instances, scene.materials)
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
        ctx.set_pixel_map(multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W))
        base, base_acc, _ = _frames(ctx, 2)
        ctx.set_entry_points(True)
        got, got_acc, _ = _frames(ctx, 2)
        for k in range(2):
            assert np.array_equal(got[k].view(np.uint32), base[k].view(np.uint32)), (eye, fwd)
        assert np.array_equal(got_acc.view(np.uint32), base_acc.view(np.uint32))


def test_a_camera_far_from_the_origin_widens_the_bundle_or_starts_at_the_root(gpu_ctx_factory):
    """generate_kernel's directions are (llc + vpX x + vpY y) - position in binary32; far from the origin with a short focus
    distance their rounding is a good part of a pixel.  The bundle's margin follows the operands: frames stay bit-equal, and a camera
    whose rounding exceeds a quarter of a pixel gets root states only."""
    W, H = 256, 144
    c = np.array((1000.0, 0.5, -800.0))
    torus = scenegen.displaced_torus(128, 64, seed=3, major=1.0, minor=0.45, amp=0.05, center=tuple(c))
    floor = scenegen.quad(tuple(c + (-6, -0.5, -6)), tuple(c + (-6, -0.5, 6)), tuple(c + (6, -0.5, 6)), tuple(c + (6, -0.5, -6)))
    light = scenegen.quad(tuple(c + (-1.2, 3.5, -1.2)), tuple(c + (1.2, 3.5, -1.2)), tuple(c + (1.2, 3.5, 1.2)), tuple(c + (-1.2, 3.5, 1.2)))
    mats = np.array([pod.make_material(pod.MAT_DIFFUSE, albedo=(0.6, 0.5, 0.4)), pod.make_material(pod.MAT_DIFFUSE, albedo=(0.7, 0.7, 0.7)),
                     pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 1.0, 1.0), intensity=20.0)], dtype=pod.MAT_DT)
    eye = c + (3.0, 1.5, 3.0)
    fwd = (c - eye) / np.linalg.norm(c - eye)
    for focus, expect_walk in ((5.0, True), (0.05, False)):
        cam = capi.camera_init(tuple(eye), fwd, 60.0, W, H, focus, 0.0)
        scene = SH.BuiltScene([torus, floor, light], [(i, i, workloads.IDENTITY) for i in range(3)], materials=mats, camera=cam,
                              settings=workloads.make_settings(use_mis=True, path_length=3, background=(0.2, 0.3, 0.4), background_intensity=1.0))
        scene.lights = SH.mesh_lights(scene.instances, scene.materials)
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
        ctx.set_pixel_map(multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W))
        base, base_acc, _ = _frames(ctx, 3)
        ctx.set_entry_points(True)
        got, got_acc, _ = _frames(ctx, 3)
        states = ctx.read_entry_states()
        walked = float((states[:, 19] >= 1).mean())
        print("focus %g: %.2f of the runs start below the root" % (focus, walked))
        assert (walked > 0.2) if expect_walk else (walked == 0.0)
        for k in range(3):
            assert np.array_equal(got[k].view(np.uint32), base[k].view(np.uint32)), focus
        assert np.array_equal(got_acc.view(np.uint32), base_acc.view(np.uint32))


def test_on_off_on_keeps_the_states_and_the_frames(gpu_ctx_factory):
    """on -> render -> off -> render -> on -> render.  The pass graph of the first "on" must not be replayed with the
    table that "off" freed: the entry kernel takes table and count from the slot's DeviceState, so the third render walks the states
    into the new table (steps > 0) and every frame equals the frame without entry points."""
    W, H = 256, 144
    scene = _torus_scene(W, H)
    pm = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)
    for in_flight in (1, 3):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        ctx.set_pixel_map(pm)
        ctx.set_passes_in_flight(in_flight)

        def one(n=3):
            ctx.reset_frame_number()
            for _ in range(n):
                ctx.render_frame()
                ctx.accumulate()
            return ctx.read_radiance().view(np.uint32).copy()

        base = one()
        ctx.set_entry_points(True)
        first = one()
        s1 = ctx.read_entry_states()
        ctx.set_entry_points(False)
        assert len(ctx.read_entry_states()) == 0
        off = one()
        ctx.set_entry_points(True)
        again = one()
        s2 = ctx.read_entry_states()
        assert (s1[:, 19] >= 1).mean() > 0.4 and np.array_equal(s1, s2), "the states are walked again after off / on"
        for got in (first, off, again):
            assert np.array_equal(got, base), in_flight


def test_passes_in_flight_have_their_own_entry_tables(gpu_ctx_factory):
    """Several passes in flight: every slot's pass graph writes the slot's OWN entry table, which only that pass's primary launch
    reads (round 5 shared one table between the slots, rewritten by every pass with identical bytes — a race that was only safe
    while every writer stored the same bytes; an earlier two-store version rendered another image with 6 passes in flight)."""
    W, H = 320, 200
    scene = _torus_scene(W, H, nu=256, nv=128)
    pm = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)

    def render(entry, in_flight, per_pass):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        ctx.set_pixel_map(pm)
        ctx.set_entry_points(entry)
        ctx.set_frames_per_pass(per_pass)
        ctx.set_passes_in_flight(in_flight)
        ctx.reset_frame_number()
        for _ in range(24 // per_pass):
            ctx.render_frame()
            ctx.accumulate()
        return ctx.read_accumulation()

    want = render(False, 1, 2)
    for in_flight, per_pass in ((6, 2), (4, 1), (2, 8)):
        got = render(True, in_flight, per_pass)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (in_flight, per_pass)
