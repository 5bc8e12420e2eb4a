"""Committed vectors (tests/golden/oracle_pins.npz, written by tests/golden/make_golden.py).  CPU: the oracle and the host
builders still reproduce them.  GPU: the HIP path reproduces them without a live oracle run.  Everything bit for bit —
radiance, accumulation, RGBA8 and queue sizes included, now that the transcendental functions are the shared text of
include/nexus_fmath.h (until round 3 they came from libm / ocml and were compared within 1e-3 on 99.5 % of the pixels)."""
import importlib.util
import os

import numpy as np
import pytest

from nexus_amd import pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    return np.load(os.path.join(GOLDEN, "oracle_pins.npz"))


def _generator():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_and_builders_reproduce_the_golden_vectors():
    g = _golden()
    new = _generator().generate()
    assert sorted(new.keys()) == sorted(g.files)
    for k in g.files:
        if "radiance" in k or "accumulation" in k:
            assert SH.frames_identical(new[k], g[k], k), k
        else:
            assert np.array_equal(new[k], g[k]), k


def _scenes():
    return {
        "soup2k": SH.soup_scene(n=2000, seed=1),
        "torus50k": SH.BuiltScene([scenegen.displaced_torus(250, 100, seed=3)], [(0, 0, SH.IDENTITY)]),
        "instanced": SH.instanced_scene(seed=3, n_inst=20),
        "cornell": SH.cornell_scene(64, 64, path_length=4),
    }


@pytest.mark.gpu
def test_gpu_hit_records_equal_golden(gpu_ctx_factory):
    g = _golden()
    gen = _generator()
    for name, scene in _scenes().items():
        ctx = gpu_ctx_factory(64, 64)
        scene.upload(ctx)
        rays = np.frombuffer(g[name + "_rays"].tobytes(), dtype=pod.RAY_DT)
        got = gen.hits_to_u32(ctx.trace_batch(rays))
        assert np.array_equal(got, g[name + "_hits"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("tag,rng_mode,compact_mode", [("slot", pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED), ("keyed", pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST)])
def test_gpu_cornell_frames_match_golden(gpu_ctx_factory, tag, rng_mode, compact_mode):
    g = _golden()
    scene = _scenes()["cornell"]
    ctx = gpu_ctx_factory(64, 64)
    scene.upload(ctx)
    ctx.set_modes(rng_mode, compact_mode, pod.CONDUCTOR_REFERENCE)
    ctx.set_tail_bounce(0)  # the golden queue sizes are those of the level-by-level pass
    ctx.reset_frame_number()
    keys = ("traceSize", "traceShadowSize", "diffuseSize", "plasticSize", "dielectricSize", "conductorSize")
    for f in range(4):
        ctx.render_frame()
        ctx.accumulate()
        assert SH.frames_identical(ctx.read_radiance(), g["cornell_radiance_" + tag][f], "cornell %s frame %d" % (tag, f + 1))
        q = ctx.read_queue_sizes()
        got = np.stack([np.asarray(q[k][:6]) for k in keys])
        assert np.array_equal(got, g["cornell_queues_" + tag][f]), f
        assert got[0, 0] == 64 * 64
    assert SH.frames_identical(ctx.read_accumulation(), g["cornell_accumulation_" + tag], "cornell %s accumulation" % tag)
    assert np.array_equal(ctx.read_rgba8(), g["cornell_rgba8_" + tag])


@pytest.mark.gpu
def test_gpu_material_zoo_matches_golden(gpu_ctx_factory):
    g = _golden()
    scene = SH.material_zoo_scene(48, 32, path_length=4)
    ctx = gpu_ctx_factory(48, 32)
    scene.upload(ctx)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    ctx.reset_frame_number()
    for f in range(2):
        ctx.render_frame()
        ctx.accumulate()
        assert SH.frames_identical(ctx.read_radiance(), g["zoo_radiance_keyed"][f], "material zoo frame %d" % (f + 1))


@pytest.mark.parametrize("rng_mode", [pod.RNG_REFERENCE_SLOT, pod.RNG_PIXEL_KEYED])
def test_threaded_oracle_equals_the_serial_oracle(rng_mode):
    """orc_wavefront_render with worker threads (what bench.py's cpu_baseline times: trace, logic and shade on ranges, slots
    handed out afterwards in item order) reproduces the serial run bit for bit — radiance and every queue size — in both
    RNG modes, on a scene with all material kinds, textures, NEE and an environment."""
    scene = SH.material_zoo_scene(width=160, height=96, path_length=5)
    n = 160 * 96
    serial = O.Wavefront(scene.oracle(), n, None, rng_mode, pod.CONDUCTOR_EXTENDED)
    threaded = O.Wavefront(scene.oracle(), n, None, rng_mode, pod.CONDUCTOR_EXTENDED)
    for f in (1, 2, 3):
        serial.render(f, threads=1)
        threaded.render(f, threads=5)
        assert np.array_equal(serial.radiance().view(np.uint32), threaded.radiance().view(np.uint32))
        qs, qt = serial.queue_sizes(), threaded.queue_sizes()
        for k in qs:
            assert np.array_equal(qs[k], qt[k]), k
    assert serial.queue_sizes()["traceSize"][1] > 4096  # large enough for the threaded path to have been taken
