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


def test_passes_in_flight_share_the_entry_table(gpu_ctx_factory):
    """Several passes in flight: every pass's graph rewrites the one entry table while the others' closest-hit launches read it.  The
    kernel stores each state once, from registers (same bytes every time); a version that first stored the root's state and then the
    walked one let a concurrent reader see a mixture (round 5: bench.py with 6 passes in flight rendered another image)."""
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
