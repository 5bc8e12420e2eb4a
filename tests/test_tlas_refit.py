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


def _geometry_bounds(scene, instances):
    """world-space box of every instance's triangles (exact: the vertices through the instance transform)"""
    lo, hi = [], []
    for inst in instances:
        m = scene.meshes[int(inst["bvhIdx"])]
        T = np.asarray(inst["transform"], np.float64).reshape(4, 4)
        v = np.concatenate([m["pos0"], m["pos1"], m["pos2"]]).astype(np.float64)
        w = v @ T[:3, :3].T + T[:3, 3]
        lo.append(w.min(0))
        hi.append(w.max(0))
    return np.array(lo), np.array(hi)


def _check_tlas_structure(nodes, idx, instances, geometry=None):
    """every instance exactly once; children behind their parent; every leaf slot's quantised box contains its instances'
    world boxes — the record's (BVHInstance::SetTransform's, from the BLAS root frame), or with `geometry` = (lo, hi) per instance
    the triangles' own: the device build tightens the record's box to what the BLAS root's children hold —; inner children
    consecutive"""
    from tests.test_builder_parity import _decode_children

    assert sorted(idx.tolist()) == list(range(len(instances)))
    seen_nodes, seen = set(), set()
    stack = [0]
    while stack:
        ni = stack.pop()
        assert ni not in seen_nodes and ni < len(nodes)
        seen_nodes.add(ni)
        inner = []
        for s, kind, lo, hi, first, count in _decode_children(nodes[ni]):
            eps = 1e-5 * np.maximum(1.0, np.abs(hi))
            if kind == "inner":
                assert first > ni
                inner.append(first)
                stack.append(first)
            else:
                assert 1 <= count <= 3
                for k in range(first, first + count):
                    assert k not in seen
                    seen.add(k)
                    inst = instances[idx[k]]
                    want_lo, want_hi = (inst["boundsMin"], inst["boundsMax"]) if geometry is None else (geometry[0][idx[k]], geometry[1][idx[k]])
                    assert np.all(lo <= want_lo + eps) and np.all(hi >= want_hi - eps), (ni, s, k)
        assert inner == list(range(inner[0], inner[0] + len(inner))) if inner else True
    assert len(seen_nodes) == len(nodes) and len(seen) == len(instances)


@pytest.mark.gpu
@pytest.mark.parametrize("n_inst", [1, 2, 9, 60, 1000])
def test_device_built_tlas_is_valid_and_traces_like_the_oracle(gpu_ctx_factory, n_inst):
    """nxhip_rebuild_tlas: the TLAS built on the device over the instances' world boxes is a valid conservative tree; the GPU
    traces it exactly as the oracle traces the same tree (hit records and visit counts), and finds the hits of the
    host-built tree and of brute force."""
    scene = SH.instanced_scene(seed=9, n_inst=n_inst)
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rays = _rays(20000, 61)
    host = scene.oracle().trace_closest(rays)
    assert SH.hit_records_equal(ctx.trace_batch(rays), host)
    nodes, idx = ctx.rebuild_tlas(scene.instances)
    _check_tlas_structure(nodes, idx, scene.instances, _geometry_bounds(scene, scene.instances))
    assert len(nodes) <= max(1, n_inst)
    rebuilt = SH.BuiltScene.__new__(SH.BuiltScene)
    rebuilt.__dict__.update(scene.__dict__)
    rebuilt.tlas_nodes, rebuilt.tlas_idx = nodes, idx
    got = ctx.trace_batch(rays)
    assert SH.hit_records_equal(got, rebuilt.oracle().trace_closest(rays))
    assert np.array_equal(got["hitDistance"].view(np.uint32), host["hitDistance"].view(np.uint32))
    same = (got["triIdx"] == host["triIdx"]) & (got["instanceIdx"] == host["instanceIdx"])
    assert same.mean() > 0.999
    sub = slice(0, 200)
    bf = scene.oracle().brute_closest(rays[sub])
    assert np.array_equal(got["hitDistance"][sub].view(np.uint32), bf["hitDistance"].view(np.uint32))
    # the installed tree takes device-side refits like a host-built one
    if n_inst >= 9:
        rng = np.random.RandomState(3)
        ids = rng.permutation(n_inst)[:5].astype(np.uint32)
        xfs = np.array([capi.mat4_from_trs(rng.uniform(-2.5, 2.5, 3), rng.uniform(0, 360, 3), rng.uniform(0.4, 1.6, 3)) for _ in ids], dtype=np.float32)
        ctx.set_instance_transforms(ids, xfs)
        moved = scene.instances.copy()
        for i, xf in zip(ids, xfs):
            old = scene.instances[i]
            moved[i] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0])
        rebuilt.instances, rebuilt.tlas_nodes = moved, capi.tlas_refit(nodes, idx, moved)
        assert SH.hit_records_equal(ctx.trace_batch(rays), rebuilt.oracle().trace_closest(rays))
        # the refitted tree on the device still bounds the geometry — with the tighter boxes it was built from, not the records'
        refitted, _ = ctx.read_tlas(len(nodes), n_inst)
        _check_tlas_structure(refitted, idx, moved, _geometry_bounds(scene, moved))
        if n_inst >= 60:
            assert refitted.tobytes() != rebuilt.tlas_nodes.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_device_tlas_boxes_hold_under_extreme_transforms(gpu_ctx_factory, seed):
    """The device build takes an instance's box from two levels of its BLAS instead of the root frame (nx_instbox.h): a box that
    is too tight loses hits.  Placements far from ordinary — scales from 1/300 to 300, mirrored, rotated about all three axes —
    rebuilt on the device, then moved once more there: the closest hits equal brute force over all triangles, and the trees
    bound the transformed triangles.  (A mesh flattened by a zero scale has no inverse transform; the reference then traverses
    it with the identity, and neither its boxes nor anybody's bound those hits: such an instance keeps the record's box —
    `singular` below only asks that the build and the traversal end.)"""
    rng = np.random.RandomState(100 + seed)
    meshes = [scenegen.random_soup(600, seed=seed, extent=0.5, size=0.1), scenegen.displaced_torus(32, 16, seed=seed, major=0.5, minor=0.2),
              scenegen.height_field(12, seed=seed, amp=0.0)]

    def placement(i):
        scale = np.exp(rng.uniform(np.log(1 / 300.0), np.log(300.0), 3)) if i % 3 == 0 else rng.uniform(0.3, 2.0, 3)
        if i % 4 == 1:
            scale[rng.randint(3)] *= -1.0          # mirrored
        return capi.mat4_from_trs(rng.uniform(-3, 3, 3), rng.uniform(0, 360, 3), scale)

    n_inst = 24
    scene = SH.BuiltScene(meshes, [(i % 3, 0, placement(i)) for i in range(n_inst)])
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    rays = np.concatenate([scenegen.random_rays(3000, seed=seed, radius=8.0, target_extent=3.5), scenegen.interior_rays(3000, seed=seed + 1, extent=3.5)])
    nodes, idx = ctx.rebuild_tlas(scene.instances)
    _check_tlas_structure(nodes, idx, scene.instances, _geometry_bounds(scene, scene.instances))
    got = ctx.trace_batch(rays)
    bf = scene.oracle().brute_closest(rays)
    assert (bf["hitDistance"] < 1e29).mean() > 0.05
    assert np.array_equal(got["hitDistance"].view(np.uint32), bf["hitDistance"].view(np.uint32))
    # moved on the device: the refit keeps the tight boxes up to date
    ids = np.arange(n_inst, dtype=np.uint32)
    xfs = np.array([placement(i + 1) for i in range(n_inst)], dtype=np.float32)
    ctx.set_instance_transforms(ids, xfs)
    moved = scene.instances.copy()
    for i, xf in zip(ids, xfs):
        old = scene.instances[i]
        moved[i] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0])
    refitted, _ = ctx.read_tlas(len(nodes), n_inst)
    _check_tlas_structure(refitted, idx, moved, _geometry_bounds(scene, moved))
    after = SH.BuiltScene.__new__(SH.BuiltScene)
    after.__dict__.update(scene.__dict__)
    after.instances = moved
    after.tlas_nodes, after.tlas_idx = capi.tlas_build(moved)
    bf = after.oracle().brute_closest(rays)
    got = ctx.trace_batch(rays)
    assert np.array_equal(got["hitDistance"].view(np.uint32), bf["hitDistance"].view(np.uint32))
    # singular placements: the build, the refit and the traversal end
    flat = np.array([capi.mat4_from_trs(rng.uniform(-3, 3, 3), rng.uniform(0, 360, 3), (1.0, 0.0, 1.5)) for _ in range(4)], dtype=np.float32)
    ctx.set_instance_transforms(np.array([1, 5, 9, 13], np.uint32), flat)
    assert len(ctx.trace_batch(rays)) == len(rays)
    for k, i in enumerate((1, 5, 9, 13)):
        old = moved[i]
        moved[i] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), flat[k], scene.blas[int(old["bvhIdx"])][0][0])
    nodes2, idx2 = ctx.rebuild_tlas(moved)
    assert len(ctx.trace_batch(rays)) == len(rays)
    # ... and such an instance keeps the RECORD's box, in the build and in the refit alike.  Whether a matrix counts as singular is
    # the inverse's own decision (Mat4::Inverted's 4 x 4 cofactor determinant == 0 -> identity), not a second determinant with
    # its own rounding: a projection onto a plane and a matrix with a zero row among ordinary placements, geometry bounds for
    # the ordinary ones, record bounds for the singular ones.
    proj = capi.mat4_from_trs((0.5, -0.2, 0.3), (20.0, 40.0, 60.0), (1.0, 1.0, 1.0)).reshape(4, 4).astype(np.float64)
    nrm = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    P = np.eye(4)
    P[:3, :3] -= np.outer(nrm, nrm)                       # rank 2: every point onto the plane through the origin
    special = {1: (P @ proj).astype(np.float32).reshape(16), 5: flat[1], 9: np.diag([1.0, 0.0, 2.0, 1.0]).astype(np.float32).reshape(16)}
    for i, xf in special.items():
        old = moved[i]
        moved[i] = capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), xf, scene.blas[int(old["bvhIdx"])][0][0])
        assert np.array_equal(np.asarray(moved[i]["invTransform"]).reshape(4, 4), np.eye(4, dtype=np.float32)), "Mat4::Inverted's fallback"
    glo, ghi = _geometry_bounds(scene, moved)
    for i in special:
        glo[i], ghi[i] = moved[i]["boundsMin"], moved[i]["boundsMax"]
    nodes3, idx3 = ctx.rebuild_tlas(moved)
    _check_tlas_structure(nodes3, idx3, moved, (glo, ghi))
    ctx.set_instance_transforms(np.array(sorted(special), np.uint32), np.array([special[i] for i in sorted(special)], np.float32))
    refit3, _ = ctx.read_tlas(len(nodes3), n_inst)
    _check_tlas_structure(refit3, idx3, moved, (glo, ghi))
    assert len(ctx.trace_batch(rays)) == len(rays)


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [-1, 0, 16])
def test_device_tlas_build_survives_bounds_that_are_not_numbers(gpu_ctx_factory, builder):
    """instances with NaN / infinite world bounds through nxhip_rebuild_tlas (all three device builders): it returns, every
    instance is in the tree once, a traversal of the tree ends"""
    scene = SH.instanced_scene(seed=9, n_inst=60)
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    ctx.set_device_builder(builder)
    bad = scene.instances.copy()
    bad["boundsMin"][3, 0] = np.nan
    bad["boundsMax"][7] = np.nan
    bad["boundsMax"][11] = np.inf
    bad["boundsMin"][13, 1] = -np.inf
    nodes, idx = ctx.rebuild_tlas(bad)
    assert sorted(idx.tolist()) == list(range(len(bad))) and len(nodes) >= 1
    got = ctx.trace_batch(_rays(4000, 5))
    assert len(got) == 4000 and (got["hitDistance"] < 1e29).sum() > 100


@pytest.mark.gpu
def test_device_tlas_build_of_sixteen_thousand_instances_is_fast(gpu_ctx_factory):
    """The case SURVEY.md section 8 row f3 names: the reference's O(n^2) clustering at 16 k instances (0.8 s for this repo's
    faster host version of it) against the device build."""
    import time

    rng = np.random.RandomState(12)
    mesh = scenegen.displaced_torus(16, 8, seed=3, major=0.4, minor=0.15)
    n = 16000
    placements = [(0, 0, capi.mat4_from_trs(rng.uniform(-40, 40, 3), rng.uniform(0, 360, 3), rng.uniform(0.5, 1.5, 3))) for _ in range(n)]
    nodes0, idx0 = capi.bvh8_build(mesh)
    insts = np.array([capi.instance_init(0, 0, xf, nodes0[0]) for (_m, _mat, xf) in placements], dtype=pod.INST_DT)
    ctx = gpu_ctx_factory(64, 64)
    ctx.upload_blas(nodes0, mesh, idx0)
    ctx.rebuild_tlas(insts[:100])  # first use: code objects, sort temporaries
    t0 = time.time()
    nodes, idx = ctx.rebuild_tlas(insts)
    dt = time.time() - t0
    print("device TLAS build of %d instances: %.4f s including upload and install, %d nodes" % (n, dt, len(nodes)))
    assert dt < 0.5

    class _OneMesh:
        meshes = [mesh]

    _check_tlas_structure(nodes, idx, insts, _geometry_bounds(_OneMesh, insts))
    ctx.set_materials(np.array([pod.make_material()], dtype=pod.MAT_DT))
    rays = scenegen.random_rays(4000, seed=5, radius=60.0, target_extent=40.0)
    got = ctx.trace_batch(rays)
    assert (got["hitDistance"] < 1e29).mean() > 0.02
    sc = SH.BuiltScene.__new__(SH.BuiltScene)
    sc.__dict__.update(dict(blas=[(nodes0, mesh, idx0)], instances=insts, tlas_nodes=nodes, tlas_idx=idx, materials=np.array([pod.make_material()], dtype=pod.MAT_DT),
                            lights=np.zeros(0, pod.LIGHT_DT), camera=None, settings=SH.workloads.make_settings(), diffuse_maps=[], emissive_maps=[], hdr_map=None, env_sampling=False))
    assert SH.hit_records_equal(got, sc.oracle().trace_closest(rays))
