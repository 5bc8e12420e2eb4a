"""The library's own multi-GPU step (include/nexus_hip.h, nxhip_mgpu_*): tile split on the CPU, RCCL gather on the GPU box."""
import numpy as np
import pytest

from nexus_amd import capi, multigpu, pod
from tests import scene_helpers as SH


@pytest.mark.parametrize("width,height,world,tile_rows", [(64, 40, 1, 5), (64, 40, 2, 5), (1920, 1080, 8, 5), (100, 48, 4, 3), (13, 16, 2, 8)])
def test_tile_pixel_map_equals_the_python_partition(width, height, world, tile_rows):
    seen = np.zeros(width * height, dtype=bool)
    for rank in range(world):
        rows = multigpu.tile_pixel_map(width, height, rank, world, tile_rows)
        assert np.array_equal(capi.tile_pixel_map(width, height, world, rank, tile_rows, tiled=False), rows)
        tiled = capi.tile_pixel_map(width, height, world, rank, tile_rows, tiled=True)
        assert np.array_equal(tiled, multigpu.tiled_order(rows, width))
        assert not seen[tiled].any()
        seen[tiled] = True
    assert seen.all()


def test_tile_pixel_map_rejects_unequal_splits():
    with pytest.raises(capi.NexusError):
        capi.tile_pixel_map(64, 41, 2, 0, 5)
    with pytest.raises(capi.NexusError):
        capi.tile_pixel_map(64, 40, 2, 2, 5)


@pytest.mark.gpu
def test_single_rank_rccl_gather_equals_the_local_image(gpu_ctx_factory):
    """world size 1 over the real RCCL path: ncclCommInitRank, ncclGather on the context's stream, compose.  The full image
    on the root must equal, bit for bit, the image of a context without the tile split."""
    W, H = 96, 80
    scene = SH.material_zoo_scene(W, H, path_length=4)
    ref = gpu_ctx_factory(W, H)
    scene.upload(ref)
    ref.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ctx.mgpu_init(1, 0, capi.mgpu_unique_id(), 5)
    assert ctx.local_count == W * H
    for _ in range(3):
        for c in (ref, ctx):
            c.render_frame()
            c.accumulate()
        ctx.mgpu_gather()
    want_acc, want_px = ref.read_accumulation(), ref.read_rgba8()
    got_acc, got_px = ctx.mgpu_read_accumulation(), ctx.mgpu_read_rgba8()
    assert np.array_equal(got_acc.view(np.uint32), want_acc.view(np.uint32))
    assert np.array_equal(got_px, want_px)
    ctx.mgpu_shutdown()


@pytest.mark.gpu
def test_a_resize_after_the_tile_split_is_refused_by_the_gather_not_launched(gpu_ctx_factory):
    """nxhip_resize / nxhip_set_pixel_map change the pixel set the split's gather buffers and maps were built for.
    The gather then returns an error status instead of launching a collective and compose kernels with stale sizes."""
    from tests import scene_helpers as SH

    W, H = 64, 40
    scene = SH.cornell_scene(W, H, path_length=2)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.mgpu_init(1, 0, capi.mgpu_unique_id(), 5)
    ctx.render_frame()
    ctx.accumulate()
    ctx.mgpu_gather()
    ctx.resize(W, 2 * H)
    with pytest.raises(capi.NexusError, match="changed after the tile split"):
        ctx.mgpu_gather()
    ctx.mgpu_shutdown()
    ctx.mgpu_init(1, 0, capi.mgpu_unique_id(), 5)  # set up again for the new size: works
    cam = capi.camera_init((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, W, 2 * H, 5.0, 0.0)
    ctx.set_camera(cam)
    ctx.render_frame()
    ctx.accumulate()
    ctx.mgpu_gather()
    assert len(ctx.mgpu_read_rgba8()) == W * 2 * H
    ctx.mgpu_shutdown()
