"""The tail kernel (nxhip_set_tail_bounce): the late bounces of every path in one launch instead of a graph level per
kernel and bounce.  It calls the same logic / shade / traversal functions in the same per-pixel order, so accumulation and
last-frame radiance must equal the level-by-level pipeline bit for bit — on all four material kinds, textures, instanced
and transformed BLASes, environment NEE / MIS, several frames per pass and passes in flight."""
import numpy as np
import pytest

from nexus_amd import pod
from tests import scene_helpers as SH

pytestmark = pytest.mark.gpu


def _render(gpu_ctx_factory, scene, W, H, tail, frames_per_pass=2, passes=3, in_flight=1, env_sampling=False):
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    if env_sampling:
        ctx.set_env_sampling(True)
    ctx.set_frames_per_pass(frames_per_pass)
    ctx.set_passes_in_flight(in_flight)
    ctx.set_tail_bounce(tail)
    for _ in range(passes):
        ctx.render_frame()
        ctx.accumulate()
    return ctx.read_accumulation(), ctx.read_radiance(), ctx.read_rgba8()


@pytest.mark.parametrize("tail", [2, 3, 5])
def test_tail_kernel_equals_the_level_by_level_pass(gpu_ctx_factory, tail):
    W, H = 96, 64
    scene = SH.material_zoo_scene(W, H, path_length=6)
    want = _render(gpu_ctx_factory, scene, W, H, 0)
    got = _render(gpu_ctx_factory, scene, W, H, tail)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    assert np.array_equal(got[2], want[2])
    assert np.any(want[0] != 0.0)


def test_tail_kernel_with_passes_in_flight_and_last_bounce(gpu_ctx_factory):
    W, H = 80, 48
    scene = SH.material_zoo_scene(W, H, path_length=5)
    want = _render(gpu_ctx_factory, scene, W, H, 0, frames_per_pass=1, passes=6)
    for tail, R in ((5, 1), (2, 3), (4, 2)):  # tail = pathLength: only the last logic / shade step
        got = _render(gpu_ctx_factory, scene, W, H, tail, frames_per_pass=1, passes=6, in_flight=R)
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (tail, R)


def test_tail_kernel_is_ignored_where_it_does_not_apply(gpu_ctx_factory):
    """Slot-keyed random numbers / ordered compaction keep the level-by-level graph: the setting must not change a bit."""
    W, H = 64, 48
    scene = SH.material_zoo_scene(W, H, path_length=4)
    out = []
    for tail in (0, 2):
        ctx = gpu_ctx_factory(W, H)
        scene.upload(ctx)
        ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE)
        ctx.set_tail_bounce(tail)
        for _ in range(2):
            ctx.render_frame()
            ctx.accumulate()
        out.append(ctx.read_accumulation())
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))
    with pytest.raises(Exception):
        ctx.set_tail_bounce(1)


def test_tail_kernel_with_environment_sampling(gpu_ctx_factory):
    W, H = 80, 48
    scene = SH.material_zoo_scene(W, H, path_length=5)
    want = _render(gpu_ctx_factory, scene, W, H, 0, env_sampling=True)
    got = _render(gpu_ctx_factory, scene, W, H, 2, env_sampling=True)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))


def test_tail_kernel_at_the_benchmark_size(gpu_ctx_factory):
    """BASELINE.json configs[1] (1 M triangles, 1920x1080, 8 bounces), two frames, passes in flight: the accumulation with the
    tail kernel from bounce 3 and from bounce 5 equals the level-by-level one bit for bit."""
    from tests import config_scenes as CS
    W, H = 1920, 1080
    scene = CS.config2(W, H)
    want = _render(gpu_ctx_factory, scene, W, H, 0, frames_per_pass=1, passes=2)
    for tail, R in ((3, 1), (5, 2)):
        got = _render(gpu_ctx_factory, scene, W, H, tail, frames_per_pass=1, passes=2, in_flight=R)
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (tail, R)
        assert np.array_equal(got[2], want[2])
