"""BLAS built on the device (nxhip_build_blas: LBVH + wide collapse, nx_lbvh.hip): structural validity and conservative
bounds of the 80-byte nodes, and the same hits as the SAH-built BLAS / brute force through the unchanged traversal kernels."""
import numpy as np
import pytest

from nexus_amd import capi, pod, scenegen
from tests import scene_helpers as SH
from tests.test_builder_parity import _decode_children

pytestmark = pytest.mark.gpu


def _check_structure(nodes, idx, tris):
    assert sorted(idx.tolist()) == list(range(len(tris))), "every triangle exactly once"
    tmin = np.minimum(np.minimum(tris["pos0"], tris["pos1"]), tris["pos2"]).astype(np.float64)
    tmax = np.maximum(np.maximum(tris["pos0"], tris["pos1"]), tris["pos2"]).astype(np.float64)
    seen_nodes, seen_prims = set(), set()
    stack = [0]
    while stack:
        ni = stack.pop()
        assert ni not in seen_nodes and ni < len(nodes)
        seen_nodes.add(ni)
        total, inner = 0, []
        for s, kind, lo, hi, first, count in _decode_children(nodes[ni]):
            eps = 1e-6 * np.maximum(1.0, np.abs(hi))
            if kind == "inner":
                inner.append(first)
                stack.append(first)
            else:
                assert 1 <= count <= 3
                total += count
                for k in range(first, first + count):
                    assert k not in seen_prims
                    seen_prims.add(k)
                    t = idx[k]
                    assert np.all(lo <= tmin[t] + eps) and np.all(hi >= tmax[t] - eps), (ni, s, k)
        assert total <= 24
        assert inner == list(range(inner[0], inner[0] + len(inner))) if inner else True, "inner children are consecutive in slot order"
    assert len(seen_nodes) == len(nodes) and len(seen_prims) == len(tris)


MESHES = {
    "soup": lambda: scenegen.random_soup(5000, seed=3),
    "soup40k": lambda: scenegen.random_soup(40000, seed=8, extent=1.0, size=0.03),
    "torus": lambda: scenegen.displaced_torus(96, 48, seed=2),
    "tiny": lambda: scenegen.random_soup(5, seed=1),
    "nine": lambda: scenegen.random_soup(9, seed=4),
    "one": lambda: scenegen.random_soup(1, seed=5),
    "same_centroids": lambda: np.repeat(scenegen.random_soup(1, seed=6), 300),
    "planar": lambda: scenegen.height_field(40, seed=7, amp=0.0),
}


@pytest.mark.parametrize("radius", [0, 16, 3, -1])
@pytest.mark.parametrize("name", list(MESHES))
def test_device_built_blas_is_a_valid_conservative_cwbvh(gpu_ctx_factory, name, radius):
    """the three device builders: the radix tree (radius 0), locally-ordered clustering with a wide and a narrow search window,
    and the top-down binned SAH build (NXHIP_BUILDER_SAH = -1)"""
    tris = np.ascontiguousarray(MESHES[name](), dtype=pod.TRI_DT)
    ctx = gpu_ctx_factory(32, 32)
    ctx.set_device_builder(radius)
    bid = ctx.build_blas(tris)
    nodes, idx = ctx.read_blas(bid, len(tris))
    _check_structure(nodes, idx, tris)


@pytest.mark.parametrize("radius", [0, 16, -1])
def test_device_builders_survive_triangles_that_are_not_numbers(gpu_ctx_factory, radius):
    """NaN and infinite vertices (a damaged file): every builder still ends, with every triangle exactly once in a tree the
    traversal walks to its end.  (Round 3 found the collapse's cost table calling any subtree a leaf once its cost was +inf:
    slots of thousands of primitives, writes past the slot's three entries, a device that never came back.)  With NaNs alone
    the boxes stay finite — fmin / fmax drop them — and the rays keep finding the intact triangles; an infinite vertex makes
    the root frame infinite, as it would for any builder, and nothing can be hit."""
    tris = np.ascontiguousarray(scenegen.random_soup(4000, seed=12, extent=1.0, size=0.05), dtype=pod.TRI_DT)
    rng = np.random.RandomState(3)
    victims = rng.choice(len(tris), 40, replace=False)
    rays = scenegen.interior_rays(20000, seed=5, extent=1.0)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    hits = {}
    for kind in ("nan", "inf", "mixed"):
        bad = tris.copy()
        if kind in ("nan", "mixed"):
            bad["pos0"][victims[:15], 0] = np.nan
        if kind in ("inf", "mixed"):
            bad["pos1"][victims[15:30]] = np.inf
            bad["pos2"][victims[30:], 2] = -np.inf
        ctx = gpu_ctx_factory(32, 32)
        ctx.set_device_builder(radius)
        bid = ctx.build_blas(bad)
        nodes, idx = ctx.read_blas(bid, len(bad))
        assert sorted(idx.tolist()) == list(range(len(bad))), kind
        assert 0 < len(nodes) < len(bad)
        inst = np.array([capi.instance_init(bid, 0, ident, nodes[0])], dtype=pod.INST_DT)
        inst["boundsMin"], inst["boundsMax"] = -4.0, 4.0  # (the root frame of such a tree may not be a number either)
        tn, ti = capi.tlas_build(inst)
        ctx.set_tlas(tn, ti, inst)
        got = ctx.trace_batch(rays)  # ends: a tree
        hits[kind] = int((got["hitDistance"] < 1e29).sum())
    assert hits["nan"] > 1000


@pytest.mark.parametrize("builders", [(16, 0), (-1, -1)], ids=["clustering+radix", "top-down-sah"])
def test_device_built_blas_traces_like_the_sah_build_and_brute_force(gpu_ctx_factory, builders):
    """Same scene twice: BLASes from the host SAH builder vs. built on the device; the oracle traverses the SAH version.
    Hit distances are identical (the closest hit does not depend on the tree); ids may differ only on equidistant ties."""
    meshes = [scenegen.displaced_torus(128, 64, seed=3, major=0.6, minor=0.25, amp=0.05), scenegen.random_soup(3000, seed=9, extent=0.6, size=0.05)]
    rng = np.random.RandomState(5)
    placements = [(i % 2, 0, capi.mat4_from_trs(rng.uniform(-2, 2, 3), rng.uniform(0, 360, 3), rng.uniform(0.6, 1.4, 3))) for i in range(12)]
    scene = SH.BuiltScene(meshes, placements)
    rays = np.concatenate([scenegen.random_rays(20000, seed=7, radius=6.0, target_extent=2.5), scenegen.interior_rays(20000, seed=8, extent=2.5)])
    want = scene.oracle().trace_closest(rays)
    ref = gpu_ctx_factory(32, 32)
    scene.upload(ref)
    assert SH.hit_records_equal(ref.trace_batch(rays), want)
    ctx = gpu_ctx_factory(32, 32)
    first, second = builders
    ctx.set_device_builder(first)
    ids = [ctx.build_blas(scene.meshes[0])]
    ctx.set_device_builder(second)
    ids.append(ctx.build_blas(scene.meshes[1]))
    assert ids == [0, 1]
    # the instances' world bounds come from the BLAS root frame (BVHInstance.cpp:8-21): recompute them for the device-built roots
    roots = [ctx.read_blas(i, len(m))[0][0] for i, m in zip(ids, scene.meshes)]
    insts = np.array([capi.instance_init(int(old["bvhIdx"]), int(old["materialId"]), old["transform"], roots[int(old["bvhIdx"])]) for old in scene.instances], dtype=pod.INST_DT)
    tlas_nodes, tlas_idx = capi.tlas_build(insts)
    ctx.set_tlas(tlas_nodes, tlas_idx, insts)
    got = ctx.trace_batch(rays)
    assert (want["hitDistance"] < 1e29).mean() > 0.1
    assert np.array_equal(got["hitDistance"].view(np.uint32), want["hitDistance"].view(np.uint32))
    same = (got["triIdx"] == want["triIdx"]) & (got["instanceIdx"] == want["instanceIdx"])
    assert same.mean() > 0.999
    occl = ctx.trace_shadow_batch(rays[:5000], np.full(5000, 3.0, np.float32))
    assert np.array_equal(occl, ref.trace_shadow_batch(rays[:5000], np.full(5000, 3.0, np.float32)))


def test_two_device_builds_of_a_mesh_are_the_same_tree(gpu_ctx_factory):
    """The top-down build hands out node numbers and level slots with atomics, but what it decides from — counts, min / max boxes,
    scan positions — does not depend on their order: two builds of a mesh give trees that every ray walks the same way (same hit
    records including equidistant ties, same numbers of nodes and triangles visited), which is what lets every rank of a tile
    split build its own copy."""
    tris = np.ascontiguousarray(scenegen.displaced_torus(160, 80, seed=6, major=0.6, minor=0.25, amp=0.05), dtype=pod.TRI_DT)
    rays = np.concatenate([scenegen.random_rays(20000, seed=3, radius=4.0, target_extent=1.2), scenegen.interior_rays(20000, seed=4, extent=1.0)])
    ident = np.eye(4, dtype=np.float32).reshape(16)
    outs = []
    for _ in range(2):
        ctx = gpu_ctx_factory(32, 32)
        bid = ctx.build_blas(tris)
        nodes, idx = ctx.read_blas(bid, len(tris))
        inst = np.array([capi.instance_init(bid, 0, ident, nodes[0])], dtype=pod.INST_DT)
        tn, ti = capi.tlas_build(inst)
        ctx.set_tlas(tn, ti, inst)
        ctx.enable_trace_stats(True)
        ctx.read_trace_stats(reset=True)
        hits = ctx.trace_batch(rays)
        st, _ = ctx.read_trace_stats(reset=True)
        outs.append((len(nodes), sorted(idx.tolist()) == list(range(len(tris))), hits, st["nodes"], st["tris"]))
    a, b = outs
    assert a[0] == b[0] and a[1] and b[1]
    assert SH.hit_records_equal(a[2], b[2])
    assert (a[3], a[4]) == (b[3], b[4])


@pytest.mark.parametrize("builder", [0, -1], ids=["radix", "top-down-sah"])
def test_device_build_of_a_million_triangles_is_fast_and_valid(gpu_ctx_factory, builder):
    import time

    tris = scenegen.displaced_torus(1024, 512, seed=1, major=1.0, minor=0.45, amp=0.06)
    ctx = gpu_ctx_factory(32, 32)
    ctx.set_device_builder(builder)
    ctx.build_blas(scenegen.random_soup(100, seed=1))  # first use: code objects, sort temporaries
    t0 = time.time()
    bid = ctx.build_blas(tris)
    dt = time.time() - t0
    nodes, idx = ctx.read_blas(bid, len(tris))
    print("device build (%s) of %d triangles: %.3f s including the upload, %d nodes" % ("radix tree" if builder == 0 else "top-down SAH", len(tris), dt, len(nodes)))
    assert dt < 2.0
    assert sorted(idx.tolist()) == list(range(len(tris)))
    assert len(nodes) < len(tris) // 3
    rays = scenegen.random_rays(20000, seed=2, radius=4.0, target_extent=1.5)
    got = ctx.trace_batch if False else None
    host_nodes, host_idx = capi.bvh8_build(tris, threads=0)
    a = gpu_ctx_factory(32, 32)
    a.upload_blas(host_nodes, tris, host_idx)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    for c, root in ((a, host_nodes[0]), (ctx, nodes[0])):
        inst = np.array([capi.instance_init(0 if c is a else bid, 0, ident, root)], dtype=pod.INST_DT)
        tn, ti = capi.tlas_build(inst)
        c.set_tlas(tn, ti, inst)
    ha, hb = a.trace_batch(rays), ctx.trace_batch(rays)
    assert (ha["hitDistance"] < 1e29).mean() > 0.2
    assert np.array_equal(ha["hitDistance"].view(np.uint32), hb["hitDistance"].view(np.uint32))


def _grid_scene(ctx, blas_ids, roots, spacing=3.0):
    """every BLAS once, on a lattice: (instances, TLAS installed)"""
    inst = []
    side = int(np.ceil(len(blas_ids) ** (1.0 / 3.0)))
    for k, (bid, root) in enumerate(zip(blas_ids, roots)):
        m = np.eye(4, dtype=np.float32)
        m[0, 3], m[1, 3], m[2, 3] = spacing * (k % side), spacing * ((k // side) % side), spacing * (k // (side * side))
        inst.append(capi.instance_init(bid, 0, m.reshape(16), root))
    inst = np.array(inst, dtype=pod.INST_DT)
    tn, ti = capi.tlas_build(inst)
    ctx.set_tlas(tn, ti, inst)
    return inst, side


def test_batched_blas_build_gives_every_mesh_the_tree_of_its_single_build(gpu_ctx_factory):
    """nxhip_build_blas_batch: many meshes as one forest (per-mesh Morton order, one root segment each, the single build's level
    loops over all of them).  Every mesh must come out as the tree its own nxhip_build_blas gives — judged like two single builds
    of one mesh are (node numbering follows the order of atomics): same node count, a valid conservative CWBVH over its own
    triangles with mesh-local indices, the same root node, and every ray walks it the same way (hit records including equidistant
    ties, nodes and triangles visited).  The mix: single nodes (1, 5, 8 triangles), the smallest real trees (9, 10), duplicates
    of one mesh, coincident triangles (equal Morton codes, position splits), a planar grid, soups and tori up to 40 000."""
    meshes = [MESHES[k]() for k in ("one", "tiny", "nine", "soup", "torus", "same_centroids", "planar", "soup40k")]
    meshes += [scenegen.random_soup(8, seed=11), scenegen.random_soup(10, seed=12), scenegen.displaced_torus(64, 20, seed=13, major=0.7, minor=0.2, amp=0.04)]
    meshes += [meshes[4].copy(), meshes[2].copy()]  # the same mesh twice in one batch
    meshes += [scenegen.random_soup(int(n), seed=100 + i, extent=0.8, size=0.08) for i, n in enumerate(np.random.RandomState(5).randint(11, 3000, size=24))]
    meshes = [np.ascontiguousarray(m, dtype=pod.TRI_DT) for m in meshes]
    counts = [len(m) for m in meshes]

    single = gpu_ctx_factory(32, 32)
    ids_s = [single.build_blas(m) for m in meshes]
    trees_s = [single.read_blas(b, n) for b, n in zip(ids_s, counts)]
    batch = gpu_ctx_factory(32, 32)
    ids_b = batch.build_blas_batch(meshes)
    assert ids_b == list(range(len(meshes)))
    trees_b = batch.read_blas_batch(ids_b[0], counts)
    for k, ((ns, _), (nb, ib)) in enumerate(zip(trees_s, trees_b)):
        assert len(ns) == len(nb), "mesh %d (%d triangles): %d nodes alone, %d in the batch" % (k, counts[k], len(ns), len(nb))
        assert ns[0].tobytes() == nb[0].tobytes() or counts[k] > 8, "single-node meshes are the same bytes"
        assert np.array_equal(ns[0]["p"], nb[0]["p"]) and np.array_equal(ns[0]["e"], nb[0]["e"]), "same root frame"
        if counts[k] <= 5000:
            _check_structure(nb, ib, meshes[k])
        else:
            assert sorted(ib.tolist()) == list(range(counts[k]))
    # the same scene over both sets of trees: every ray walks them the same way
    stats = []
    for ctx, ids, trees in ((single, ids_s, trees_s), (batch, ids_b, trees_b)):
        _, side = _grid_scene(ctx, ids, [t[0][0] for t in trees])
        ext = 3.0 * side
        rays = np.concatenate([scenegen.random_rays(60000, seed=3, radius=2.0 * ext, target_extent=ext / 2), scenegen.interior_rays(60000, seed=4, extent=ext / 2)])
        rays["origin"] += np.float32(ext / 2 - 1.5)
        ctx.enable_trace_stats(True)
        ctx.read_trace_stats(reset=True)
        hits = ctx.trace_batch(rays)
        st, _ = ctx.read_trace_stats(reset=True)
        stats.append((hits, st["nodes"], st["tris"], st["instances"]))
    assert (stats[0][0]["hitDistance"] < 1e29).mean() > 0.1
    assert SH.hit_records_equal(stats[0][0], stats[1][0])
    assert stats[0][1:] == stats[1][1:]


def test_a_thousand_meshes_of_a_thousand_triangles_in_one_build(gpu_ctx_factory):
    """1 000 meshes x 1 000 triangles (a glTF scene's worth of small meshes) in one nxhip_build_blas_batch call,
    timed from the host arrays to installed BLASes, against the same meshes built one by one."""
    import time

    rng = np.random.RandomState(9)
    base = scenegen.displaced_torus(25, 20, seed=3, major=0.6, minor=0.25, amp=0.05)
    assert len(base) == 1000
    meshes = []
    for k in range(1000):
        m = base.copy()
        s = np.float32(rng.uniform(0.5, 2.0))
        for f in ("pos0", "pos1", "pos2"):
            m[f] = m[f] * s + rng.uniform(-0.2, 0.2, size=(1, 3)).astype(np.float32)
        meshes.append(np.ascontiguousarray(m, dtype=pod.TRI_DT))
    ctx = gpu_ctx_factory(32, 32)
    ctx.build_blas_batch(meshes[:8])  # first use: code objects, sort temporaries, the pinned staging buffer
    ctx.sync()
    t0 = time.perf_counter()
    ids = ctx.build_blas_batch(meshes)
    ctx.sync()
    dt_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    for m in meshes[:100]:
        ctx.build_blas(m)
    ctx.sync()
    dt_single = (time.perf_counter() - t0) * 10.0
    print("1000 meshes x 1000 triangles: one batched build %.1f ms; one by one %.0f ms (100 of them timed)" % (dt_batch * 1e3, dt_single * 1e3))
    assert ids == list(range(8, 1008))
    assert dt_batch < 0.05, "%.1f ms" % (dt_batch * 1e3)
    trees = ctx.read_blas_batch(ids[0], [1000] * 1000)
    for k in (0, 499, 999):
        _check_structure(trees[k][0], trees[k][1], meshes[k])
    # one of them against its own single build
    single = gpu_ctx_factory(32, 32)
    bid = single.build_blas(meshes[500])
    ns, _ = single.read_blas(bid, 1000)
    assert len(ns) == len(trees[500][0])
