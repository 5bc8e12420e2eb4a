"""TLAS refit (nxh_tlas_refit): with unchanged instances it reproduces the builder's bytes; after the instances move it
still bounds them — closest hits through the refitted tree equal brute force and a rebuilt tree's."""
import numpy as np
import pytest

from nexus_amd import capi, pod, scenegen
from tests import scene_helpers as SH


def _moved(scene, seed):
    rng = np.random.RandomState(seed)
    insts = []
    for i, old in enumerate(scene.instances):
        xf = capi.mat4_from_trs(rng.uniform(-2.5, 2.5, 3), rng.uniform(0, 360, 3), rng.uniform(0.4, 1.6, 3))
        insts.append(capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0]))
    return np.array(insts, dtype=pod.INST_DT)


def _rays(n, seed):
    a = scenegen.random_rays(n // 2, seed=seed, radius=6.0, target_extent=2.5)
    b = scenegen.interior_rays(n - n // 2, seed=seed + 1, extent=2.5)
    return np.concatenate([a, b])


@pytest.mark.parametrize("n_inst", [1, 7, 20, 300])
def test_refit_with_unchanged_instances_is_the_builders_output(n_inst):
    scene = SH.instanced_scene(seed=4, n_inst=n_inst)
    again = capi.tlas_refit(scene.tlas_nodes, scene.tlas_idx, scene.instances)
    assert again.tobytes() == np.ascontiguousarray(scene.tlas_nodes).tobytes()


def test_refit_after_moving_instances_matches_brute_force_and_rebuild():
    scene = SH.instanced_scene(seed=5, n_inst=40)
    moved = _moved(scene, seed=11)
    refit_nodes = capi.tlas_refit(scene.tlas_nodes, scene.tlas_idx, moved)
    assert refit_nodes.tobytes() != np.ascontiguousarray(scene.tlas_nodes).tobytes()
    # only frames and quantised boxes change
    for f in ("imask", "childBaseIdx", "triangleBaseIdx", "meta"):
        assert np.array_equal(refit_nodes[f], scene.tlas_nodes[f]), f
    rays = _rays(6000, 21)
    refit = SH.BuiltScene.__new__(SH.BuiltScene)
    refit.__dict__.update(scene.__dict__)
    refit.instances, refit.tlas_nodes = moved, refit_nodes
    got = refit.oracle().trace_closest(rays)
    rebuilt_nodes, rebuilt_idx = capi.tlas_build(moved)
    rebuilt = SH.BuiltScene.__new__(SH.BuiltScene)
    rebuilt.__dict__.update(scene.__dict__)
    rebuilt.instances, rebuilt.tlas_nodes, rebuilt.tlas_idx = moved, rebuilt_nodes, rebuilt_idx
    want = rebuilt.oracle().trace_closest(rays)
    assert (want["hitDistance"] < 1e29).mean() > 0.05
    assert np.array_equal(got["hitDistance"].view(np.uint32), want["hitDistance"].view(np.uint32))
    same = (got["triIdx"] == want["triIdx"]) & (got["instanceIdx"] == want["instanceIdx"])
    assert same.mean() > 0.999  # equidistant hits may resolve differently: the visiting order differs
    sub = slice(0, 300)
    bf = refit.oracle().brute_closest(rays[sub])
    assert np.array_equal(got["hitDistance"][sub].view(np.uint32), bf["hitDistance"].view(np.uint32))


def test_refit_rejects_malformed_input():
    scene = SH.instanced_scene(seed=6, n_inst=12)
    bad = scene.tlas_idx.copy()
    bad[0] = 99
    with pytest.raises(capi.NexusError):
        capi.tlas_refit(scene.tlas_nodes, bad, scene.instances)
    nodes = scene.tlas_nodes.copy()
    nodes["childBaseIdx"][0] = 0  # a node may not be its own descendant
    if nodes["imask"][0]:
        with pytest.raises(capi.NexusError):
            capi.tlas_refit(nodes, scene.tlas_idx, scene.instances)


@pytest.mark.gpu
def test_gpu_traces_a_refitted_tlas_like_the_oracle(gpu_ctx_factory):
    scene = SH.instanced_scene(seed=7, n_inst=60)
    moved = _moved(scene, seed=13)
    scene.tlas_nodes = capi.tlas_refit(scene.tlas_nodes, scene.tlas_idx, moved)
    scene.instances = moved
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rays = _rays(30000, 31)
    assert SH.hit_records_equal(ctx.trace_batch(rays), scene.oracle().trace_closest(rays))


@pytest.mark.gpu
@pytest.mark.parametrize("n_inst,n_moved", [(60, 60), (1000, 1000), (1000, 37), (1, 1)])
def test_device_side_refit_equals_the_host_refit_and_traces_like_the_oracle(gpu_ctx_factory, n_inst, n_moved):
    """nxhip_set_instance_transforms: the instance records and the refitted TLAS left in HBM are, byte for byte, what
    nexus::BVHInstance::SetTransform + nexus::collapse::Refit produce on the host; rays traced through them equal the oracle
    tracing the host-refitted scene."""
    scene = SH.instanced_scene(seed=8, n_inst=n_inst)
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rng = np.random.RandomState(17)
    ids = rng.permutation(n_inst)[:n_moved].astype(np.uint32)
    xfs = np.array([capi.mat4_from_trs(rng.uniform(-2.5, 2.5, 3), rng.uniform(0, 360, 3), rng.uniform(0.4, 1.6, 3)) for _ in ids], dtype=np.float32)
    before = ctx.trace_batch(_rays(2000, 3))
    ctx.set_instance_transforms(ids, xfs)
    want_inst = scene.instances.copy()
    for i, xf in zip(ids, xfs):
        old = scene.instances[i]
        want_inst[i] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0])
    want_nodes = capi.tlas_refit(scene.tlas_nodes, scene.tlas_idx, want_inst)
    got_nodes, got_inst = ctx.read_tlas(len(scene.tlas_nodes), n_inst)
    assert got_inst.tobytes() == want_inst.tobytes()
    assert got_nodes.tobytes() == want_nodes.tobytes()
    scene.instances, scene.tlas_nodes = want_inst, want_nodes
    rays = _rays(20000, 41)
    got = ctx.trace_batch(rays)
    assert SH.hit_records_equal(got, scene.oracle().trace_closest(rays))
    assert not SH.hit_records_equal(ctx.trace_batch(_rays(2000, 3)), before) or n_inst == 1
    with pytest.raises(capi.NexusError):
        ctx.set_instance_transforms(np.array([n_inst], np.uint32), xfs[:1])


@pytest.mark.gpu
def test_device_side_transforms_keep_the_identity_shortcut_honest(gpu_ctx_factory):
    """A scene placed with identity transforms only (the trace kernels then skip the transform rows altogether), one instance
    rotated on the device and later put back: after each step the GPU traces what the oracle traces for that placement."""
    meshes = [scenegen.random_soup(900, seed=21, extent=0.6, size=0.1), scenegen.displaced_torus(32, 16, seed=21, major=0.5, minor=0.2)]
    scene = SH.BuiltScene(meshes, [(0, 0, SH.IDENTITY), (1, 0, SH.IDENTITY), (0, 0, SH.IDENTITY)])
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rays = _rays(12000, 51)
    # signed zeros in the directions as well: the shortcut must leave those rays to the general path
    rays["direction"][:200, 0] = np.float32(-0.0)
    rays["direction"][200:400, 2] = np.float32(0.0)
    assert SH.hit_records_equal(ctx.trace_batch(rays), scene.oracle().trace_closest(rays))

    def place(xf):
        ctx.set_instance_transforms(np.array([1], np.uint32), np.array([xf], dtype=np.float32))
        old = scene.instances[1]
        inst = scene.instances.copy()
        inst[1] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0])
        moved = SH.BuiltScene.__new__(SH.BuiltScene)
        moved.__dict__.update(scene.__dict__)
        moved.instances, moved.tlas_nodes = inst, capi.tlas_refit(scene.tlas_nodes, scene.tlas_idx, inst)
        return moved

    rotated = place(capi.mat4_from_trs((0.3, -0.2, 0.1), (30, 60, 10), (1.2, 0.9, 1.0)))
    got = ctx.trace_batch(rays)
    assert SH.hit_records_equal(got, rotated.oracle().trace_closest(rays))
    back = place(SH.IDENTITY)
    got = ctx.trace_batch(rays)
    assert SH.hit_records_equal(got, back.oracle().trace_closest(rays))
    assert SH.hit_records_equal(got, scene.oracle().trace_closest(rays))
