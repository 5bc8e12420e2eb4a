"""Committed vectors (tests/golden/oracle_pins.npz, written by tests/golden/make_golden.py).  CPU: the oracle and the host
builders still reproduce them.  GPU: the HIP path reproduces them without a live oracle run.  Bit patterns exactly;
radiance / queue sizes (libm-dependent on the CPU side) within the tolerances below."""
import importlib.util
import os

import numpy as np
import pytest

from nexus_amd import pod, scenegen
from tests import oracle_lib as O
from tests import scene_helpers as SH

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIXEL_TOL = 1e-3     # |got - want| <= 1e-3 * max(1, |want|) per channel
MIN_AGREE = 0.995    # fraction of pixels that must satisfy it (a path that branches differently is a different sample)


def _golden():
    return np.load(os.path.join(GOLDEN, "oracle_pins.npz"))


def _generator():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _queues_close(got, want):
    return np.all(np.abs(got.astype(np.int64) - want.astype(np.int64)) <= np.maximum(4, 0.01 * np.abs(want)))


def test_oracle_and_builders_reproduce_the_golden_vectors():
    g = _golden()
    new = _generator().generate()
    assert sorted(new.keys()) == sorted(g.files)
    for k in g.files:
        if "radiance" in k or "accumulation" in k:
            a, b = new[k].reshape(-1, 3), g[k].reshape(-1, 3)
            assert SH.image_agreement(a, b, PIXEL_TOL) >= MIN_AGREE, k
        elif "queues" in k:
            assert _queues_close(new[k], g[k]), k
        elif "rgba8" in k:
            assert (new[k] == g[k]).mean() >= 0.98, k
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
        assert SH.image_agreement(ctx.read_radiance(), g["cornell_radiance_" + tag][f], PIXEL_TOL) >= MIN_AGREE, f
        q = ctx.read_queue_sizes()
        got = np.stack([np.asarray(q[k][:6]) for k in keys])
        assert _queues_close(got, g["cornell_queues_" + tag][f]), f
        assert got[0, 0] == 64 * 64
    assert SH.image_agreement(ctx.read_accumulation(), g["cornell_accumulation_" + tag], PIXEL_TOL) >= MIN_AGREE
    assert (ctx.read_rgba8() == g["cornell_rgba8_" + tag]).mean() >= 0.98


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
        assert SH.image_agreement(ctx.read_radiance(), g["zoo_radiance_keyed"][f], PIXEL_TOL) >= 0.99, f


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
