"""The kept C++ host API (nexus::Scene / AssetManager / PathTracer, include/nexus/) driven with the reference's call
sequence.  CPU part: scene bookkeeping (instances, lights, TLAS).  GPU part: a frame through the facade equals a frame
through the bare C-ABI with the same inputs."""
import os

import numpy as np
import pytest

from nexus_amd import capi, loaders, pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH


def _cornell_facade(width, height, path_length, before_meshes=None):
    ls = loaders.load_glb(os.path.join(SH.GOLDEN, "cornell_box.glb"))
    sc = capi.Scene(width, height)
    if before_meshes is not None:
        before_meshes(sc)
    mats = ls.materials.copy()
    mats["type"] = pod.MAT_DIFFUSE
    for m in mats:
        sc.add_material(m)
    mesh_ids = [sc.add_mesh(m) for m in ls.meshes]
    for inst in ls.instances:
        sc.create_instance(mesh_ids[inst["mesh"]], inst["material"], inst["position"], inst["rotation"], inst["scale"])
    sc.set_camera((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, 5.0, 0.0)
    sc.set_render_settings(O.make_settings(use_mis=True, path_length=path_length))
    sc.update()
    return sc


def test_scene_bookkeeping_matches_reference_rules():
    sc = _cornell_facade(32, 32, 4)
    assert sc.instance_count() == 8
    assert sc.light_count() == 1  # only the emissive "light" primitive (Scene.cpp:142-176)


@pytest.mark.gpu
def test_facade_frame_equals_direct_capi_frame(gpu_ctx_factory):
    W = H = 96
    sc = _cornell_facade(W, H, 4)
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE)
    pt.update_device_scene(sc)
    for _ in range(2):
        pt.render(sc)
    assert pt.frame_number() == 2
    got_rad, got_px = pt.read_radiance(), pt.read_pixels()

    scene = SH.cornell_scene(W, H, path_length=4)
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE)
    ctx.reset_frame_number()
    for _ in range(2):
        ctx.render_frame()
        ctx.accumulate()
    assert np.array_equal(got_rad.view(np.uint32), ctx.read_radiance().view(np.uint32))
    assert np.array_equal(got_px, ctx.read_rgba8())
    pt.close()


@pytest.mark.gpu
def test_the_benched_configuration_through_the_facade_equals_the_ctypes_path(gpu_ctx_factory, tmp_path):
    """bench.py's headline settings — configs[1]'s scene read from a Wavefront .obj by OBJLoader, BLASes and TLAS
    built on the device, pixel-keyed random numbers, extended conductor, 8 x 8 pixel tiles (nxhip_set_pixel_order: the order comes from
    the LIBRARY now, not from Python), entry points, several frames per Render() call — rendered through nexus::Scene / PathTracer
    give the RGBA8 image the ctypes path gives for workloads.config2 built in memory.  (examples/nexus_bench.cpp is this call
    sequence in C++; bench.py --through-facade times it.)"""
    from nexus_amd import multigpu, workloads

    W, H, nu, nv, frames = 256, 144, 96, 48, 4
    want_scene = workloads.config2(W, H, nu, nv, 5, cls=SH.BuiltScene)
    loaders.write_obj(str(tmp_path / "mesh.obj"), want_scene.meshes[0])

    sc = capi.Scene(W, H)
    pt = capi.PathTracer(W, H)
    pt.set_device_blas_build(sc, True)
    sc.set_device_tlas(True)
    sc.load_file(str(tmp_path) + "/", "mesh.obj")  # mesh 0, its default material, instance 0
    mats = want_scene.materials
    ids = [sc.add_material(m) for m in mats]
    floor_id = sc.add_mesh(want_scene.meshes[1], ids[1])
    light_id = sc.add_mesh(want_scene.meshes[2], ids[2])
    sc.create_instance(floor_id, ids[1])
    sc.create_instance(light_id, ids[2])
    sc.assign_material(0, ids[0])
    eye = np.array((0.0, 3.3, 4.9))
    fwd = np.array((0.0, 0.35, 0.0)) - eye  # (workloads._look: normalised in binary64, rounded by the binding)
    sc.set_camera(tuple(eye), tuple(fwd / np.linalg.norm(fwd)), 52.0, 5.0, 0.0)
    sc.set_render_settings(want_scene.settings)
    sc.update()
    assert sc.instance_count() == 3 and sc.light_count() == 1
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    pt.set_pixel_order(pod.ORDER_TILES)
    pt.set_entry_points(True)
    pt.set_frames_per_pass(frames)
    pt.update_device_scene(sc)
    for _ in range(3):
        pt.render(sc)
    assert pt.frame_number() == 3 * frames
    got = pt.read_pixels()
    pt.set_device_blas_build(sc, False)
    pt.close()

    ctx = gpu_ctx_factory(W, H)
    want_scene.upload(ctx, device_bvh=True, device_tlas=True)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ctx.set_pixel_map(multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W))  # (the Python twin of the library's order)
    ctx.set_entry_points(True)
    ctx.set_frames_per_pass(frames)
    ctx.reset_frame_number()
    for _ in range(3):
        ctx.render_frame()
        ctx.accumulate()
    want = ctx.read_rgba8()
    assert np.array_equal(got, want), "%d of %d pixels differ" % (int((got != want).sum()), got.size)
    # ... and nxhip_set_pixel_order gives the very map the Python helper computes
    ctx.set_pixel_order(pod.ORDER_TILES)
    ctx.set_frames_per_pass(frames)
    ctx.reset_frame_number()
    for _ in range(3):
        ctx.render_frame()
        ctx.accumulate()
    assert np.array_equal(ctx.read_rgba8(), want)
    assert np.array_equal(capi.tile_pixel_map(W, H, 1, 0, 1, tiled=True), multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W))


@pytest.mark.gpu
def test_facade_small_pass_measures_do_not_change_the_image():
    """PathTracer::SetFramesPerPass / SetPassesInFlight (and the tail kernel that small keyed passes use by default): six
    frames as six Render() calls, as three calls of two frames, and with three calls in flight — the same pixels."""
    W, H = 96, 64
    images = []
    for frames, in_flight in ((1, 1), (2, 1), (1, 3), (2, 3)):
        sc = _cornell_facade(W, H, 6)
        pt = capi.PathTracer(W, H)
        pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        pt.update_device_scene(sc)
        pt.set_frames_per_pass(frames)
        pt.set_passes_in_flight(in_flight)
        for _ in range(6 // frames):
            pt.render(sc)
        assert pt.frame_number() == 6
        images.append(pt.read_pixels())
        pt.close()
    for img in images[1:]:
        assert np.array_equal(img, images[0])
    assert len(np.unique(images[0])) > 16


def _cornell_from_file(width, height, path_length, before_load=None):
    """The reference's own call: Scene::CreateMeshInstanceFromFile (C++ glb reader), materials as the file gives them."""
    sc = capi.Scene(width, height)
    if before_load:
        before_load(sc)
    sc.load_file(SH.GOLDEN + os.sep, "cornell_box.glb")
    sc.set_camera((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, 5.0, 0.0)
    sc.set_render_settings(O.make_settings(use_mis=True, path_length=path_length))
    sc.update()
    return sc


def test_scene_from_file_bookkeeping():
    sc = _cornell_from_file(32, 32, 4)
    assert sc.instance_count() == 8 and sc.light_count() == 1
    with pytest.raises(capi.NexusError):
        capi.Scene(32, 32).load_file(SH.GOLDEN + os.sep, "no_such_file.glb")


@pytest.mark.gpu
def test_scene_loaded_from_file_renders_like_the_python_built_scene(gpu_ctx_factory):
    W = H = 64
    sc = _cornell_from_file(W, H, 4)
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    pt.update_device_scene(sc)
    for _ in range(2):
        pt.render(sc)
    got = pt.read_radiance()
    scene = SH.cornell_scene(W, H, path_length=4, force_diffuse=False)  # the file's materials are PLASTIC
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    ctx.reset_frame_number()
    for _ in range(2):
        ctx.render_frame()
        ctx.accumulate()
    assert np.array_equal(got.view(np.uint32), ctx.read_radiance().view(np.uint32))
    pt.close()


@pytest.mark.gpu
def test_moving_instances_with_tlas_refit_renders_like_a_rebuild():
    """Dynamic transforms through the facade: the refit extension and the reference's rebuild give the same image
    (pixel-keyed RNG; hits are those of a correct TLAS either way)."""
    W = H = 64
    imgs = []
    for refit in (False, True):
        sc = _cornell_facade(W, H, 3)
        sc.set_tlas_refit(refit)
        pt = capi.PathTracer(W, H)
        pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
        pt.update_device_scene(sc)
        pt.render(sc)
        # move the two boxes (instances 5 and 6 of the Cornell file) and render the edited scene from frame 1 again
        sc.set_instance_transform(5, (0.35, 0.05, 0.25), (90.0, 10.0, 0.0), (1.0, 1.0, 1.0))
        sc.set_instance_transform(6, (-0.3, 0.1, -0.2), (90.0, -25.0, 0.0), (0.9, 0.9, 1.1))
        sc.update()
        pt.update_device_scene(sc)
        pt.reset_frame_number()
        pt.render(sc)
        imgs.append(pt.read_radiance())
        pt.close()
    assert np.isfinite(imgs[0]).all() and imgs[0].max() > 0
    assert SH.image_agreement(imgs[1], imgs[0], 1e-6) >= 0.999


@pytest.mark.gpu
def test_scene_with_the_tlas_left_to_the_device_renders_like_the_host_built_one():
    """Scene::SetDeviceTlasBuild: no TLAS builder runs on the host — the device builds the tree from the instances when the
    scene is first sent and when an instance is added, and refits it when instances move.  Same image as the reference's
    host rebuild at every stage (pixel-keyed RNG; hits are those of a correct TLAS either way)."""
    W = H = 64
    imgs = {False: [], True: []}
    for device_tlas in (False, True):
        sc = _cornell_facade(W, H, 3)
        sc.set_device_tlas(device_tlas)
        sc.update()
        pt = capi.PathTracer(W, H)
        pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)

        def frame():
            pt.update_device_scene(sc)
            pt.reset_frame_number()
            pt.render(sc)
            imgs[device_tlas].append(pt.read_radiance())

        frame()
        sc.set_instance_transform(5, (0.35, 0.05, 0.25), (90.0, 10.0, 0.0), (1.0, 1.0, 1.0))  # moved: device-side refit
        sc.update()
        frame()
        sc.create_instance(6, 1, position=(0.0, 1.2, 0.0), rotation_deg=(90.0, 45.0, 0.0), scale=(0.5, 0.5, 0.5))  # added: rebuild
        sc.update()
        frame()
        assert sc.instance_count() == 9
        pt.close()
    for a, b in zip(imgs[False], imgs[True]):
        assert np.isfinite(a).all() and a.max() > 0
        assert SH.image_agreement(b, a, 1e-6) >= 0.999
    assert SH.image_agreement(imgs[True][2], imgs[True][0], 1e-6) < 0.999  # the edits are visible


@pytest.mark.gpu
def test_meshes_built_on_the_device_through_the_facade_render_like_host_built_ones():
    """PathTracer::SetDeviceBlasBuild: AssetManager::CreateBVH hands the triangles to nxhip_build_blas and keeps the tree that
    comes back, instances take their world bounds from that tree's root frame, nothing is uploaded a second time.  Same image
    as with the host builder's trees (closest hits do not depend on the tree; the Cornell box has no equidistant ties off its
    shared edges), also after a mesh is added later and after the scene starts over."""
    W = H = 64
    extra = scenegen.displaced_torus(48, 24, seed=5, major=0.3, minor=0.12, amp=0.02, center=(0.0, 1.0, 0.0))
    imgs = {}
    for device in (False, True):
        pt = capi.PathTracer(W, H)
        pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
        frames = []
        for round_ in range(2):  # the second round: a new scene on the same PathTracer — the device's BLAS list starts over
            sc = _cornell_facade(W, H, 3, before_meshes=(lambda s: pt.set_device_blas_build(s, True)) if device else None)
            pt.update_device_scene(sc)
            pt.reset_frame_number()
            pt.render(sc)
            frames.append(pt.read_radiance())
            mesh = sc.add_mesh(np.ascontiguousarray(extra, dtype=pod.TRI_DT))
            sc.create_instance(mesh, 1, position=(0.0, 0.0, 0.0), rotation_deg=(0.0, 30.0, 0.0), scale=(1.0, 1.0, 1.0))
            sc.update()
            pt.update_device_scene(sc)
            pt.reset_frame_number()
            pt.render(sc)
            frames.append(pt.read_radiance())
            if device:
                nodes, idx = pt.read_blas(8, len(extra))  # the ninth BLAS of the scene: the torus
                assert sorted(idx.tolist()) == list(range(len(extra))) and len(nodes) > 10
                pt.set_device_blas_build(sc, False)
        imgs[device] = frames
        pt.close()
    for a, b in zip(imgs[False], imgs[True]):
        assert np.isfinite(a).all() and a.max() > 0
        assert SH.image_agreement(b, a, 1e-6) >= 0.999
    assert SH.image_agreement(imgs[True][1], imgs[True][0], 1e-6) < 0.999  # the added mesh is visible
    assert np.array_equal(imgs[True][0], imgs[True][2]) and np.array_equal(imgs[True][1], imgs[True][3])  # the second scene renders as the first


@pytest.mark.gpu
def test_a_file_loaded_with_the_device_builder_on_builds_its_meshes_in_one_batch():
    """OBJLoader::LoadOBJ creates the BVHs of a file's meshes together (AssetManager::CreateBVHs); with
    PathTracer::SetDeviceBlasBuild on that is ONE nxhip_build_blas_batch call for the eight meshes of cornell_box.glb (the
    reference: one CreateBVH per aiMesh, Assets/OBJLoader.cpp:213-239).  Same image as with the host-built trees, and the trees
    the asset manager keeps are the device's."""
    W = H = 64
    imgs = {}
    for device in (False, True):
        pt = capi.PathTracer(W, H)
        pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
        sc = _cornell_from_file(W, H, 4, before_load=(lambda s: pt.set_device_blas_build(s, True)) if device else None)
        pt.update_device_scene(sc)
        for _ in range(2):
            pt.render(sc)
        imgs[device] = pt.read_radiance()
        if device:
            # Cornell box: 8 meshes of 2 - 12 triangles: single-node trees (at most 8 triangles) and one real one
            for bid in range(8):
                nodes, idx = pt.read_blas(bid, 64)
                assert len(nodes) >= 1
            pt.set_device_blas_build(sc, False)
        pt.close()
    assert np.isfinite(imgs[False]).all() and imgs[False].max() > 0
    assert SH.image_agreement(imgs[True], imgs[False], 1e-6) >= 0.999


@pytest.mark.gpu
def test_cpp_example_program_renders_the_same_image(tmp_path):
    """examples/nexus_render.cpp (the reference's render loop through the kept C++ API, no Python in the process) against
    the same scene driven through the ctypes facade."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "nexus_render")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "nexus_render.cpp"), "-o", exe,
           "-L" + os.path.join(root, "nexus_amd", "lib"), "-lnexus_amd", "-Wl,-rpath," + os.path.join(root, "nexus_amd", "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    W = H = 64
    out = str(tmp_path / "cornell.ppm")
    r = subprocess.run([exe, SH.GOLDEN + os.sep, "cornell_box.glb", out, str(W), str(H), "3", "4"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, NEXUS_DETERMINISTIC="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "8 instances, 1 lights" in r.stdout
    data = open(out, "rb").read()
    header = b"P6\n%d %d\n255\n" % (W, H)
    assert data.startswith(header) and len(data) == len(header) + W * H * 3
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)[::-1]  # back to bottom row first

    sc = _cornell_from_file(W, H, 4)
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    pt.update_device_scene(sc)
    for _ in range(3):
        pt.render(sc)
    want = pt.read_pixels().reshape(H, W).view(np.uint8).reshape(H, W, 4)[..., :3]
    pt.close()
    assert np.array_equal(got, want)
    # the same program with the BVHs and the TLAS built on the device: the same picture up to equidistant ties
    out2 = str(tmp_path / "cornell_device.ppm")
    r = subprocess.run([exe, SH.GOLDEN + os.sep, "cornell_box.glb", out2, str(W), str(H), "3", "4"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, NEXUS_DETERMINISTIC="1", NEXUS_DEVICE_BUILDERS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BVHs built on the device" in r.stdout
    dev = np.frombuffer(open(out2, "rb").read()[len(header):], np.uint8).reshape(H, W, 3)[::-1]
    assert (np.abs(dev.astype(int) - got.astype(int)).max(axis=2) <= 1).mean() > 0.995
    # bad input: a message and a non-zero status, no crash
    r = subprocess.run([exe, SH.GOLDEN + os.sep, "missing.glb", out], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_sphere_glb_renders_like_the_oracle(gpu_ctx_factory):
    """The reference's second demo asset (two glass spheres, DIELECTRIC through transmission > 0, ior 1.45 / 2.5): device
    frames against the CPU oracle, and the C++ file path against the Python-built scene bit for bit."""
    W = H = 96
    path = os.path.join(SH.GOLDEN, "cornell_box_sphere.glb")
    scene = SH.glb_scene(path, W, H, path_length=6)
    assert sum(len(m) for m in scene.meshes) == 2188
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, pod.CONDUCTOR_REFERENCE)
    w = O.Wavefront(scene.oracle(), W * H)
    for f in (1, 2):
        ctx.render_frame()
        ctx.accumulate()
        w.render(f)
        w.accumulate(f)
        assert SH.frames_identical(ctx.read_radiance(), w.radiance(), "cornell_box_sphere.glb frame %d" % f)  # (long glass paths included)
    # through Scene::CreateMeshInstanceFromFile (C++ reader + builders)
    sc = capi.Scene(W, H)
    sc.load_file(SH.GOLDEN + os.sep, "cornell_box_sphere.glb")
    sc.set_camera((0.0, 1.0, 3.9), (0.0, 0.0, -1.0), 40.0, 5.0, 0.0)
    sc.set_render_settings(O.make_settings(use_mis=True, path_length=6))
    sc.update()
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    pt.update_device_scene(sc)
    pt.render(sc)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    ctx.reset_frame_number()
    ctx.render_frame()
    assert np.array_equal(pt.read_radiance().view(np.uint32), ctx.read_radiance().view(np.uint32))
    pt.close()


@pytest.mark.gpu
def test_textured_glb_through_the_facade_renders_like_the_python_built_scene(gpu_ctx_factory, tmp_path):
    """glTF images -> IMGLoader -> AssetManager::AddTexture -> nxhip_upload_texture (OBJLoader.cpp:115-160, Texture.cpp:10-39):
    same radiance, bit for bit, as uploading the Python-decoded textures directly; and against the oracle's software sampler."""
    from tests.test_loaders import _textured_glb

    W, H = 64, 48
    path = str(tmp_path / "textured.glb")
    _textured_glb(path, np.random.RandomState(21))
    cam = dict(eye=(0.0, 1.5, 2.5), forward=(0.0, -0.55, -0.83), hfov=50.0)
    scene = SH.glb_scene(path, W, H, path_length=3, **cam)
    assert len(scene.diffuse_maps) == 1 and len(scene.emissive_maps) == 1 and len(scene.lights) == 1
    ctx = gpu_ctx_factory(W, H)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    ctx.render_frame()
    want = ctx.read_radiance()
    assert float(want.max()) > 0.0  # the textured emissive quad is seen
    w = O.Wavefront(scene.oracle(), W * H, None, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_REFERENCE)
    w.render(1)
    assert SH.frames_identical(want, w.radiance(), "textured glb")
    sc = capi.Scene(W, H)
    sc.load_file(str(tmp_path) + os.sep, "textured.glb")
    sc.set_camera(cam["eye"], cam["forward"], cam["hfov"], 5.0, 0.0)
    sc.set_render_settings(O.make_settings(use_mis=True, path_length=3))
    sc.update()
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_REFERENCE)
    pt.update_device_scene(sc)
    pt.render(sc)
    assert np.array_equal(pt.read_radiance().view(np.uint32), want.view(np.uint32))
    pt.close()
