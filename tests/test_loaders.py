"""Scene ingestion: the .glb reader on the reference's own Cornell box asset (facts from SURVEY.md Appendix C) and the
.obj reader on a generated file."""
import os

import numpy as np
import pytest

from nexus_amd import loaders, pod
from tests import scene_helpers as SH


def test_cornell_glb_facts():
    ls = loaders.load_glb(os.path.join(SH.GOLDEN, "cornell_box.glb"))
    assert [len(m) for m in ls.meshes] == [2, 2, 2, 2, 2, 10, 10, 2]  # floor ceiling back right left shortBox tallBox light
    assert sum(len(m) for m in ls.meshes) == 32 and len(ls.instances) == 8
    assert ls.material_names == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "shortBox", "tallBox", "light"]
    assert np.allclose(ls.materials["u"][3][:3], (0.14, 0.45, 0.091), atol=1e-6)
    assert np.allclose(ls.materials["u"][4][:3], (0.63, 0.065, 0.05), atol=1e-6)
    light = ls.materials[7]
    assert np.allclose(light["emissive"], 1.0) and light["intensity"] == 35.0
    assert all(abs(m["u"][3] - 0.9) < 1e-4 and m["u"][4] == 1.0 and m["type"] == pod.MAT_PLASTIC for m in ls.materials)
    for inst in ls.instances:  # node rotation quaternion (0.7071, 0, 0, 0.7071) = +90 degrees about X
        assert np.allclose(inst["rotation"], (90, 0, 0), atol=1e-3) and np.allclose(inst["scale"], 1, atol=1e-6) and np.allclose(inst["position"], 0)


def test_cornell_scene_bounds_and_lights():
    sc = SH.cornell_scene(32, 32)
    assert len(sc.lights) == 1 and sc.lights[0]["meshId"] == 7
    # after the rotation the box spans x [-1.02, 1], y [0, 1.99], z [-1.04, 0.99] (open side towards +z)
    pts = []
    for inst, (nodes, tris, idx) in zip(sc.instances, sc.blas):
        m = inst["transform"].reshape(4, 4)
        for k in ("pos0", "pos1", "pos2"):
            p = tris[k] @ m[:3, :3].T + m[:3, 3]
            pts.append(p)
    pts = np.concatenate(pts)
    assert np.allclose(pts.min(0), (-1.02, 0.0, -1.04), atol=2e-3) and np.allclose(pts.max(0), (1.0, 1.99, 0.99), atol=2e-3)


def test_obj_reader(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4//1 -3//1 -2//1\n")
    ls = loaders.load_obj(str(p))
    assert len(ls.meshes) == 1 and len(ls.meshes[0]) == 3  # quad fan = 2 triangles + 1
    t = ls.meshes[0]
    assert np.allclose(t["pos0"][0], (0, 0, 0)) and np.allclose(t["pos2"][1], (0, 1, 0))
    assert np.allclose(t["normal0"], (0, 0, 1))


# ---- the C++ reader (nexus::OBJLoader through the C-ABI) against the Python one ------------------------------------

def _same_scene(cpp, py):
    meshes, mats, insts = cpp
    assert len(meshes) == len(py.meshes)
    for a, b in zip(meshes, py.meshes):
        assert a.tobytes() == np.ascontiguousarray(b).tobytes()
    assert mats.tobytes() == np.ascontiguousarray(py.materials).tobytes()
    assert len(insts) == len(py.instances)
    for a, b in zip(insts, py.instances):
        assert a["mesh"] == b["mesh"] and a["material"] == b["material"]
        for k in ("position", "rotation", "scale"):
            assert np.allclose(a[k], b[k], rtol=0, atol=1e-5), (k, a[k], b[k])


def test_cpp_glb_reader_equals_python_reader():
    from nexus_amd import capi

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_box.glb")
    cpp = capi.load_scene_file(path)
    py = loaders.load_glb(path)
    _same_scene(cpp, py)
    assert len(cpp[0]) == 8 and sum(len(m) for m in cpp[0]) == 32


def test_cpp_obj_reader_equals_python_reader(tmp_path):
    from nexus_amd import capi

    rng = np.random.RandomState(3)
    p = tmp_path / "soup.obj"
    lines = []
    for _ in range(40):
        lines.append("v %.6f %.6f %.6f" % tuple(rng.uniform(-1, 1, 3)))
    for _ in range(10):
        lines.append("vn %.6f %.6f %.6f" % tuple(rng.uniform(-1, 1, 3)))
        lines.append("vt %.6f %.6f" % tuple(rng.uniform(0, 1, 2)))
    for _ in range(25):  # triangles, quads and a pentagon; positive and negative indices; all three index forms
        n = rng.choice([3, 3, 4, 5])
        vs = rng.randint(1, 41, n)
        lines.append("f " + " ".join("%d/%d/%d" % (v, rng.randint(1, 11), rng.randint(1, 11)) for v in vs))
    lines.append("f -1/-1/-1 -2/-2/-2 -3/-3/-3")
    p.write_text("\n".join(lines) + "\n")
    _same_scene(capi.load_scene_file(str(p)), loaders.load_obj(str(p)))
    # positions only: face normals are generated
    q = tmp_path / "plain.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4 2\n")
    _same_scene(capi.load_scene_file(str(q)), loaders.load_obj(str(q)))


def test_cpp_reader_rejects_bad_input(tmp_path):
    from nexus_amd import capi

    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\x00\x00\x00" + b"\xff" * 24)
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(bad))
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(tmp_path / "missing.obj"))
    idx = tmp_path / "idx.obj"
    idx.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(idx))
    with pytest.raises(capi.NexusError):
        capi.load_scene_file(str(tmp_path / "scene.fbx"))
