"""GPU parity, whole wavefront: frames rendered by the HIP path (through the C-ABI) against the CPU oracle.

The bar is equality of bits: radiance, accumulation, RGBA8 and every queue size of every bounce.  Integer / byte work and
everything built from + - * / sqrt fma was exact from the start; since round 4 the transcendental functions (sin cos exp
log pow atan2 asin) are one shared text of IEEE operations (include/nexus_fmath.h) on both sides, so nothing is left that
could differ by an ulp and flip a Russian-roulette or lobe decision.  (Until round 3 these tests asserted ">= 99.5 % of the
pixels within 1e-3".)"""
import numpy as np
import pytest

from nexus_amd import pod
from tests import oracle_lib as O
from tests import scene_helpers as SH

pytestmark = pytest.mark.gpu



def _render_gpu(ctx, frames, accumulate=True):
    out = []
    ctx.reset_frame_number()
    for _ in range(frames):
        ctx.render_frame()
        if accumulate:
            ctx.accumulate()
        out.append(ctx.read_radiance())
    return out


def _render_oracle(scene, n, frames, rng_mode, conductor_mode, pixel_map=None):
    w = O.Wavefront(scene.oracle(), n, pixel_map, rng_mode, conductor_mode)
    out = []
    for f in range(1, frames + 1):
        w.render(f, threads=8)
        w.accumulate(f)
        out.append(w.radiance())
    return w, out


def _check_queue_sizes(got, want, path_length):
    assert SH.queue_sizes_identical(got, want, path_length + 2)


@pytest.mark.parametrize("rng_mode,compact_mode", [(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED), (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST),
                                                   (pod.RNG_PIXEL_KEYED, pod.COMPACT_ORDERED)])
def test_cornell_frames_match_oracle(gpu_ctx_factory, rng_mode, compact_mode):
    W = H = 160
    scene = SH.cornell_scene(W, H, path_length=4)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(rng_mode, compact_mode, pod.CONDUCTOR_REFERENCE)
    ctx.set_tail_bounce(0)  # the queues of every bounce are inspected below
    got = _render_gpu(ctx, 3)
    orc, want = _render_oracle(scene, W * H, 3, rng_mode, pod.CONDUCTOR_REFERENCE)
    for f in range(3):
        assert SH.frames_identical(got[f], want[f], "cornell frame %d" % (f + 1))
    _check_queue_sizes(ctx.read_queue_sizes(), orc.queue_sizes(), 4)
    # accumulated image + tonemap (LinearToGamma's pow is the shared text too)
    assert SH.frames_identical(ctx.read_accumulation(), orc.accumulation(), "cornell accumulation")
    assert np.array_equal(ctx.read_rgba8(), orc.rgba8())


def test_config1_cornell_512_single_frame(gpu_ctx_factory):
    """BASELINE.json configs[0] at its full size: 512 x 512, 4 bounces, diffuse only, fixed seed (frame 1)."""
    W = H = 512
    scene = SH.cornell_scene(W, H, path_length=4)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE)
    ctx.set_tail_bounce(0)  # the queues of every bounce are inspected below
    got = _render_gpu(ctx, 1)[0]
    orc, want = _render_oracle(scene, W * H, 1, pod.RNG_REFERENCE_SLOT, pod.CONDUCTOR_REFERENCE)
    assert SH.frames_identical(got, want[0], "configs[0] frame 1")
    _check_queue_sizes(ctx.read_queue_sizes(), orc.queue_sizes(), 4)


@pytest.mark.parametrize("conductor_mode", [pod.CONDUCTOR_REFERENCE, pod.CONDUCTOR_EXTENDED])
@pytest.mark.parametrize("rng_mode,compact_mode", [(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED), (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST)])
def test_material_zoo_matches_oracle(gpu_ctx_factory, rng_mode, compact_mode, conductor_mode):
    W, H = 96, 64
    scene = SH.material_zoo_scene(W, H, path_length=5)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(rng_mode, compact_mode, conductor_mode)
    ctx.set_tail_bounce(0)  # the queues of every bounce are inspected below (the tail kernel is covered by test_gpu_tail.py)
    got = _render_gpu(ctx, 4)
    orc, want = _render_oracle(scene, W * H, 4, rng_mode, conductor_mode)
    q = orc.queue_sizes()
    assert q["plasticSize"][1] > 0 and q["dielectricSize"][1] > 0 and q["conductorSize"][1] > 0 and q["diffuseSize"][1] > 0
    for f in range(4):
        assert SH.frames_identical(got[f], want[f], "material zoo frame %d" % (f + 1))
    _check_queue_sizes(ctx.read_queue_sizes(), q, 5)


def test_fast_pixel_keyed_is_bitwise_reproducible(gpu_ctx_factory):
    """With the RNG keyed by pixel, racing slot allocation cannot change the image: two runs are identical bit for bit."""
    W, H = 128, 96
    scene = SH.material_zoo_scene(W, H, path_length=5)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    a = _render_gpu(ctx, 3)
    b = _render_gpu(ctx, 3)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    # and the ordered single-workgroup mode gives the very same image
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_ORDERED, pod.CONDUCTOR_EXTENDED)
    c = _render_gpu(ctx, 3)
    for x, y in zip(a, c):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_tile_split_equals_full_frame(gpu_ctx_factory):
    """Multi-GPU partition: interleaved row tiles rendered separately reassemble to the single-GPU image, bit for bit
    (pixel-keyed RNG).  Also checked against the oracle rendering the same tile."""
    W, H, G, TILE = 96, 64, 3, 8
    scene = SH.material_zoo_scene(W, H, path_length=4)
    full = gpu_ctx_factory(W, H)
    scene.upload(full)
    full.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ref = _render_gpu(full, 2)[-1]
    out = np.zeros_like(ref)
    for rank in range(G):
        rows = [r for r in range(H) if (r // TILE) % G == rank]
        pm = np.concatenate([np.arange(r * W, (r + 1) * W, dtype=np.uint32) for r in rows])
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        ctx.set_pixel_map(pm)
        tile = _render_gpu(ctx, 2)[-1]
        out[pm] = tile
        if rank == 1:
            _, want = _render_oracle(scene, len(pm), 2, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED, pixel_map=pm)
            assert SH.frames_identical(tile, want[-1], "tile of rank 1")
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_accumulate_external_matches_local(gpu_ctx_factory):
    W, H = 64, 48
    scene = SH.cornell_scene(W, H, path_length=3)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    ctx.reset_frame_number()
    for f in range(1, 4):
        ctx.render_frame()
        ctx.accumulate()
    want_acc, want8 = ctx.read_accumulation(), ctx.read_rgba8()
    ctx.reset_frame_number()
    for f in range(1, 4):
        ctx.render_frame()
        ctx.accumulate_external(ctx.radiance_device_ptr(), W * H, f, None)
    assert np.array_equal(ctx.read_accumulation().view(np.uint32), want_acc.view(np.uint32))
    assert np.array_equal(ctx.read_rgba8(), want8)


def test_pixel_query_and_resize(gpu_ctx_factory):
    W, H = 64, 64
    scene = SH.cornell_scene(W, H, path_length=2)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_pixel_query(W // 2, 2)  # bottom centre: the floor (instance 0)
    ctx.render_frame()
    orc = scene.oracle()
    hits = orc.trace_closest(np.zeros(0, dtype=pod.RAY_DT))
    inst = ctx.get_selected_instance()
    assert 0 <= inst < len(scene.instances)
    ctx.resize(32, 32)
    scene.camera = __import__("nexus_amd").capi.camera_init((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, 32, 32, 5.0, 0.0)
    ctx.set_camera(scene.camera)
    assert ctx.frame_number() == 0
    ctx.render_frame()
    ctx.accumulate()
    _, want = _render_oracle(scene, 32 * 32, 1, pod.RNG_REFERENCE_SLOT, pod.CONDUCTOR_REFERENCE)
    assert ctx.read_radiance().shape == (32 * 32, 3)


def test_frames_per_pass_equals_single_frames(gpu_ctx_factory):
    """Batching S frames into one pass changes how much work each launch carries, not the result: every frame slice and
    the accumulated image are bit-identical to S single-frame passes (pixel-keyed RNG)."""
    W, H, S = 80, 48, 3
    scene = SH.material_zoo_scene(W, H, path_length=4)
    a = gpu_ctx_factory(W, H)
    scene.upload(a)
    a.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    singles = _render_gpu(a, 2 * S)
    acc_single = a.read_accumulation()
    b = gpu_ctx_factory(W, H)
    scene.upload(b)
    b.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    b.set_frames_per_pass(S)
    for p in range(2):
        b.render_frame()
        b.accumulate()
        rad = b.read_radiance().reshape(S, W * H, 3)
        for s in range(S):
            assert np.array_equal(rad[s].view(np.uint32), singles[p * S + s].view(np.uint32)), (p, s)
    assert b.frame_number() == 2 * S
    assert np.array_equal(b.read_accumulation().view(np.uint32), acc_single.view(np.uint32))
    assert np.array_equal(b.read_rgba8(), a.read_rgba8())


def test_short_last_pass_and_compose_tiles(gpu_ctx_factory):
    """A frame budget that is not a multiple of frames-per-pass ends with a shorter pass (no reset of the frame counter
    or the accumulation), and tiles accumulated per rank + nxhip_compose_tiles give the single-context image bit for bit."""
    import ctypes as C

    W, H, frames, S = 64, 48, 7, 3
    from nexus_amd import multigpu

    scene = SH.material_zoo_scene(W, H, path_length=4)
    a = gpu_ctx_factory(W, H)
    scene.upload(a)
    a.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    for _ in range(frames):
        a.render_frame()
        a.accumulate()
    acc_ref = a.read_accumulation()
    rgba_ref = a.read_rgba8()

    b = gpu_ctx_factory(W, H)
    scene.upload(b)
    b.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    b.set_frames_per_pass(S)
    b.render(frames)  # passes of 3, 3, 1
    assert b.frame_number() == frames and b.frames_per_pass == S
    assert np.array_equal(b.read_accumulation().view(np.uint32), acc_ref.view(np.uint32))
    assert np.array_equal(b.read_rgba8(), rgba_ref)

    # two "ranks" on one GPU: interleaved tiles, per-rank accumulation, root-side composition into caller-owned buffers
    hip = C.CDLL("libamdhip64.so.7")  # the HIP runtime libnexus_amd.so is already bound to (matched by soname)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    world = 2
    n_full = W * H
    d_acc, d_rgba = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_acc), n_full * 16) == 0 and hip.hipMalloc(C.byref(d_rgba), n_full * 4) == 0
    try:
        for r in range(world):
            pm = multigpu.tile_pixel_map(W, H, r, world, tile_rows=4)
            c = gpu_ctx_factory(W, H)
            scene.upload(c)
            c.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
            c.set_pixel_map(pm)
            c.set_frames_per_pass(S)
            c.render(frames)
            c.sync()
            d_map = C.c_void_p()
            assert hip.hipMalloc(C.byref(d_map), len(pm) * 4) == 0
            assert hip.hipMemcpy(d_map, pm.ctypes.data_as(C.c_void_p), len(pm) * 4, 1) == 0
            c.compose_tiles(c.accumulation_device_ptr(), len(pm), d_map.value, d_acc.value, d_rgba.value)
            c.sync()
            hip.hipFree(d_map)
        acc = np.zeros((n_full, 4), np.float32)
        rgba = np.zeros(n_full, np.uint32)
        assert hip.hipMemcpy(acc.ctypes.data_as(C.c_void_p), d_acc, n_full * 16, 2) == 0
        assert hip.hipMemcpy(rgba.ctypes.data_as(C.c_void_p), d_rgba, n_full * 4, 2) == 0
    finally:
        hip.hipFree(d_acc)
        hip.hipFree(d_rgba)
    assert np.array_equal(acc[:, :3].view(np.uint32), acc_ref.reshape(-1, 3).view(np.uint32))
    assert np.array_equal(rgba, rgba_ref.reshape(-1))


def test_kernel_timing_modes_do_not_change_results(gpu_ctx_factory):
    W, H = 64, 48
    scene = SH.material_zoo_scene(W, H, path_length=3)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    want = _render_gpu(ctx, 2)[-1]
    for in_graph in (False, True):
        ctx.enable_kernel_timing(True, in_graph=in_graph)
        ctx.read_kernel_times(reset=True)
        got = _render_gpu(ctx, 2)[-1]
        kt = ctx.read_kernel_times(reset=True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert kt["trace"]["launches"] == 2 * 4 and kt["shadow"]["launches"] == 2 * 3 and kt["accumulate"]["launches"] == 2
        assert all(v["ms"] > 0 for k, v in kt.items() if v["launches"])
    ctx.enable_kernel_timing(False)
    assert np.array_equal(_render_gpu(ctx, 2)[-1].view(np.uint32), want.view(np.uint32))


def test_cross_table_indices_are_validated_before_launch(gpu_ctx_factory):
    """Indices the kernels follow without a bounds test (instance -> material, material -> texture, light -> instance) are
    checked on the host; a scene without lights renders (NEE is skipped) instead of indexing an empty light table."""
    from nexus_amd.capi import NexusError

    W, H = 48, 32
    scene = SH.material_zoo_scene(W, H, path_length=3)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.render_frame()
    bad = scene.materials.copy()
    bad["diffuseMapId"][1] = 7
    ctx.set_materials(bad)
    with pytest.raises(NexusError):
        ctx.render_frame()
    ctx.set_materials(scene.materials[:2])  # instances refer to materials 0..5
    with pytest.raises(NexusError):
        ctx.render_frame()
    ctx.set_materials(scene.materials)
    lights = scene.lights.copy()
    lights["meshId"][0] = len(scene.instances)
    ctx.set_lights(lights)
    with pytest.raises(NexusError):
        ctx.render_frame()
    # no lights at all: defined behaviour, identical in the oracle
    ctx.set_lights(np.zeros(0, pod.LIGHT_DT))
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    got = _render_gpu(ctx, 1)[-1]
    scene.lights = np.zeros(0, pod.LIGHT_DT)
    _, want = _render_oracle(scene, W * H, 1, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED)
    assert np.isfinite(got).all() and SH.frames_identical(got, want[-1], "no lights")
    assert ctx.read_queue_sizes()["traceShadowSize"][1] == 0


def test_accumulation_checkpoint_resumes_bit_for_bit(gpu_ctx_factory, tmp_path):
    from nexus_amd import imageio

    W, H = 48, 32
    scene = SH.material_zoo_scene(W, H, path_length=3)
    a = gpu_ctx_factory(W, H)
    scene.upload(a)
    a.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    for _ in range(6):
        a.render_frame()
        a.accumulate()
    want_acc, want_px = a.read_accumulation(), a.read_rgba8()

    b = gpu_ctx_factory(W, H)
    scene.upload(b)
    b.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    for _ in range(4):
        b.render_frame()
        b.accumulate()
    ckpt = tmp_path / "acc.pfm"
    imageio.write_pfm(str(ckpt), b.read_accumulation(), W, H)
    frame = b.frame_number()
    b.close()

    c = gpu_ctx_factory(W, H)
    scene.upload(c)
    c.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    acc, w, h = imageio.read_pfm(str(ckpt))
    assert (w, h) == (W, H)
    c.write_accumulation(acc, frame)
    assert c.frame_number() == 4
    for _ in range(2):
        c.render_frame()
        c.accumulate()
    assert np.array_equal(c.read_accumulation().view(np.uint32), want_acc.view(np.uint32))
    assert np.array_equal(c.read_rgba8(), want_px)


@pytest.mark.parametrize("W,H,path_length", [(1, 1, 3), (33, 17, 1), (7, 64, 40), (130, 3, 6)])
def test_ragged_viewports_and_extreme_path_lengths(gpu_ctx_factory, W, H, path_length):
    """Viewports that are not multiples of the wave / workgroup / tile sizes, a single pixel, pathLength 1 and a long one."""
    scene = SH.material_zoo_scene(W, H, path_length=path_length)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    frames = 3
    got = _render_gpu(ctx, frames)
    w, want = _render_oracle(scene, W * H, frames, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED)
    for f in range(frames):
        assert got[f].shape == (W * H, 3) and np.isfinite(got[f]).all()
        assert SH.frames_identical(got[f], want[f], "%dx%d pathLength %d frame %d" % (W, H, path_length, f + 1))
    q, qo = ctx.read_queue_sizes(), w.queue_sizes()
    assert q["traceSize"][0] == W * H
    assert all(int(q["traceSize"][b]) == 0 for b in range(path_length + 1, path_length + 3))
    # the same frames with several frames per pass and with an 8x8-tiled pixel order
    from nexus_amd import multigpu

    pm = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)
    ctx.set_pixel_map(pm)
    ctx.set_frames_per_pass(frames)
    ctx.reset_frame_number()
    ctx.render_frame()
    rad = ctx.read_radiance().reshape(frames, W * H, 3)
    for f in range(frames):
        full = np.zeros_like(rad[f])
        full[pm] = rad[f]
        assert np.array_equal(full.view(np.uint32), got[f].view(np.uint32)), f


def test_passes_in_flight_render_the_same_image(gpu_ctx_factory):
    """nxhip_set_passes_in_flight: consecutive passes run concurrently in their own slots (queues, stream, graph instance) and
    are folded into the one accumulation in order — accumulation, last radiance, queue sizes and pixel query are bit-identical
    to one pass at a time, also with pass sizes that change on the way."""
    W, H = 96, 64
    scene = SH.material_zoo_scene(W, H, path_length=4)
    schedule = [2, 2, 1, 3, 2, 2, 2, 1]
    results = []
    for R in (1, 3, 4):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        ctx.set_frames_per_pass(3)
        ctx.set_passes_in_flight(R)
        ctx.set_tail_bounce(0)  # queue sizes are compared below; the tail kernel with passes in flight: test_gpu_tail.py
        ctx.set_pixel_query(40, 30)
        for i, n in enumerate(schedule):
            ctx.set_frames_per_pass(n)
            ctx.render_frame()
            ctx.accumulate()
        results.append((ctx.read_accumulation(), ctx.read_rgba8(), ctx.read_radiance(), ctx.read_queue_sizes()["traceSize"][:5].copy(), ctx.get_selected_instance(),
                        ctx.frame_number()))
    for got in results[1:]:
        assert np.array_equal(got[0].view(np.uint32), results[0][0].view(np.uint32))
        assert np.array_equal(got[1], results[0][1])
        assert np.array_equal(got[2].view(np.uint32), results[0][2].view(np.uint32))
        assert np.array_equal(got[3], results[0][3]) and got[4] == results[0][4] and got[5] == results[0][5] == sum(schedule)


def test_pass_through_at_the_first_hit_keeps_the_camera_origin_for_mis(gpu_ctx_factory):
    """A half-transparent sheet in front of a light that fills the view: half of the paths pass through the sheet at bounce 1
    (path state untouched, PathTracer.cu:372-386) and hit the emitter at bounce 2, where the MIS weight measures the
    distance from the path's previous vertex — still the camera origin.  Pixelwise against the oracle."""
    from nexus_amd import capi, scenegen
    W, H = 96, 64
    sheet = scenegen.quad((-3, -2, 0.0), (3, -2, 0.0), (3, 2, 0.0), (-3, 2, 0.0))
    light = scenegen.quad((-4, -3, -2.0), (4, -3, -2.0), (4, 3, -2.0), (-4, 3, -2.0))
    mats = np.array([pod.make_material(pod.MAT_DIFFUSE, albedo=(0.6, 0.6, 0.6), opacity=0.5),
                     pod.make_material(pod.MAT_DIFFUSE, albedo=(0.8, 0.8, 0.8), emissive=(1.0, 0.8, 0.6), intensity=3.0)], dtype=pod.MAT_DT)
    cam = capi.camera_init((0.0, 0.0, 3.0), (0.0, 0.0, -1.0), 40.0, W, H, 5.0, 0.0)
    scene = SH.BuiltScene([sheet, light], [(0, 0, SH.IDENTITY), (1, 1, SH.IDENTITY)], materials=mats, camera=cam,
                          settings=O.make_settings(use_mis=True, path_length=4, background=(0, 0, 0), background_intensity=0.0))
    scene.lights = SH.mesh_lights(scene.instances, scene.materials)
    for rng_mode, compact_mode in ((pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST), (pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED)):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(rng_mode, compact_mode, pod.CONDUCTOR_REFERENCE)
        got = _render_gpu(ctx, 3)
        _, want = _render_oracle(scene, W * H, 3, rng_mode, pod.CONDUCTOR_REFERENCE)
        for f in range(3):
            assert want[f].max() > 0.5
            assert SH.frames_identical(got[f], want[f], "pass-through rng mode %d frame %d" % (rng_mode, f + 1))


def test_ordered_compaction_with_passes_in_flight_and_across_the_epoch_wrap(gpu_ctx_factory):
    """The grid-wide ordered compaction keeps its tile status words between launches and tells them apart by a serial (pass epoch
    of the slot, bounce, kernel kind).  Two things no single-pass test reaches: several passes in flight, each in a slot with its
    own status words and its own epoch count; and the wrap of the epoch after 2^20 - 1 passes (a viewer at a thousand one-frame
    passes per second is there in 17 minutes), where the words are cleared in stream order and the count starts over.  Frames
    must stay those of the serial oracle, bit for bit, on both sides of the wrap."""
    W, H = 128, 96  # 12 288 paths: 48 tiles of 256, 12 of 1 024
    scene = SH.material_zoo_scene(W, H, path_length=5)
    frames = 9
    orc, want = _render_oracle(scene, W * H, frames, pod.RNG_REFERENCE_SLOT, pod.CONDUCTOR_EXTENDED)
    for R in (1, 3):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_EXTENDED)
        ctx.set_tail_bounce(0)
        ctx.set_passes_in_flight(R)
        ctx.reset_frame_number()
        ctx.render_frame()
        ctx.accumulate()
        ctx.sync()
        ctx.debug_set_scan_epoch((1 << 20) - 4)  # three more passes per slot, then the wrap
        got = [ctx.read_radiance()]
        for _ in range(frames - 1):
            ctx.render_frame()
            ctx.accumulate()
            got.append(ctx.read_radiance())
        ctx.sync()
        for f in range(frames):
            if R == 1 or f == frames - 1:  # (with passes in flight read_radiance returns the last ACCUMULATED pass: compare the end state)
                assert SH.frames_identical(got[f], want[f], "ordered, %d in flight, frame %d" % (R, f + 1))
        assert SH.frames_identical(ctx.read_accumulation(), orc.accumulation(), "accumulation, %d in flight" % R)
        assert np.array_equal(ctx.read_rgba8(), orc.rgba8())


@pytest.mark.parametrize("rng_mode,compact_mode", [(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED), (pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST)])
@pytest.mark.parametrize("W,H", [(1, 1), (33, 31), (32, 32), (41, 25), (23, 89), (64, 32), (683, 3), (241, 17)],
                         ids=["1", "1023", "1024", "1025", "2047", "2048", "2049", "4097"])
def test_path_counts_at_the_tile_boundaries_of_the_slot_allocators(gpu_ctx_factory, rng_mode, compact_mode, W, H):
    """The logic kernel hands out slots per tile of 2 048 items (two per thread, sub-tile 0's slots in front of sub-tile 1's), the
    material kernels per 256 (racing) or 1 024 (ordered): viewports whose pixel count — the size of the first queue — sits on, one
    below and one above those edges, in the serial-order mode and in the racing one, every frame and queue size against the oracle."""
    scene = SH.material_zoo_scene(W, H, path_length=4)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(rng_mode, compact_mode, pod.CONDUCTOR_EXTENDED)
    ctx.set_tail_bounce(0)
    got = _render_gpu(ctx, 2)
    orc, want = _render_oracle(scene, W * H, 2, rng_mode, pod.CONDUCTOR_EXTENDED)
    for f in range(2):
        assert SH.frames_identical(got[f], want[f], "%d paths, frame %d" % (W * H, f + 1))
    _check_queue_sizes(ctx.read_queue_sizes(), orc.queue_sizes(), 4)
