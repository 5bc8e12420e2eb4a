"""N > 1 path on CPU: world_size-2 gloo run of the tile split + gather + root accumulate, against a single-rank render."""
import os
import subprocess
import sys

import socket

import numpy as np
import pytest

from nexus_amd import multigpu, pod
from tests import oracle_lib as O
from tests import scene_helpers as SH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_maps_partition_the_image():
    for W, H, G in [(1920, 1080, 8), (1920, 1080, 4), (1920, 1080, 2), (64, 40, 2), (16, 6, 3)]:
        t = multigpu.tile_rows_for(H, G)
        maps = [multigpu.tile_pixel_map(W, H, r, G, t) for r in range(G)]
        assert len({len(m) for m in maps}) == 1, "every rank must own the same number of pixels"
        allp = np.sort(np.concatenate(maps))
        assert np.array_equal(allp, np.arange(W * H, dtype=np.uint32))
        tiles = [m.astype(np.float32)[:, None] for m in maps]
        assert np.array_equal(multigpu.reassemble(W, H, G, t, tiles)[:, 0], np.arange(W * H, dtype=np.float32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_two_rank_gloo_gather_equals_single_rank(tmp_path):
    out = str(tmp_path / "acc.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", _free_port(),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    W, H, FRAMES = 48, 40, 2
    scene = SH.cornell_scene(W, H, path_length=3)
    w = O.Wavefront(scene.oracle(), W * H, None, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_REFERENCE)
    acc = None
    for f in range(1, FRAMES + 1):
        w.render(f)
        acc = multigpu.running_mean(acc, w.radiance(), f)
    # pixel-keyed RNG: the two-rank image is the single-rank image, bit for bit, under both exchange schemes
    assert got.shape[0] == 2
    assert np.array_equal(got[0].view(np.uint32), acc.view(np.uint32)), "radiance gather + root accumulate"
    assert np.array_equal(got[1].view(np.uint32), acc.view(np.uint32)), "per-rank accumulation + tile gather"


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_render_the_single_rank_image(tmp_path):
    """bench.py's own N > 1 code path (zero-copy torch view of the accumulation, per-pass gather, compose on the root,
    pass-size changes across ranks) as fresh child processes: 2 ranks sharing the one GPU of the box over gloo.  The PNG
    must equal the 1-rank PNG byte for byte (pixel-keyed RNG) — under strong scaling (K steps = K frames in total) that of the
    same command, under weak scaling (the default: one frame per GPU and step, K steps = 2 K frames) that of the 1-rank run
    with twice the steps and warm-up."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NX_BENCH_BACKEND="gloo", NX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    size = ["--reps", "1", "--frames-per-pass", "2", "--width", "256", "--height", "160", "--no-cpu-baseline"]
    k5, k10 = ["--steps", "5", "--warmup", "2"], ["--steps", "10", "--warmup", "4"]

    def run(cmd_prefix, extra, png):
        r = subprocess.run(cmd_prefix + [os.path.join(ROOT, "bench.py")] + extra + size + ["--png", png], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])

    def two_ranks():
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", _free_port()]

    one5, one10, strong, weak = (str(tmp_path / n) for n in ("one5.png", "one10.png", "strong.png", "weak.png"))
    run([sys.executable], k5 + ["--no-roofline"], one5)
    run([sys.executable], k10 + ["--no-roofline"], one10)
    line = run(two_ranks(), ["--gpus", "2", "--scaling", "strong"] + k5, strong)
    assert open(one5, "rb").read() == open(strong, "rb").read()
    assert line["scaling"] == "strong" and line["config"]["frames_timed"] == 5 and "strong_scaling" not in line["config"]
    # a distributed run reports every rank's own time, gathered to rank 0 (a bad scaling curve must be readable from one run)
    pr = line["config"]["per_rank"]
    assert len(pr["median_ms_by_rank"]) == 2 and pr["slowest_rank"] in (0, 1) and pr["backend"] == "gloo" and line["n_gpus"] == 2
    # ... and rank 0's roofline block (its launches on its half of the pixels; rank 1 waits at the closing barrier meanwhile)
    rf = line["roofline"]
    assert rf["scope"].startswith("rank 0 of 2") and rf["rays_per_launch"] > 0 and "cpu_baseline" not in line

    line = run(two_ranks(), ["--gpus", "2", "--no-roofline"] + k5, weak)
    assert open(one10, "rb").read() == open(weak, "rb").read()
    cfg = line["config"]
    assert line["scaling"] == "weak" and line["steps"] == 5 and cfg["frames_per_step"] == 2 and cfg["frames_timed"] == 10 and cfg["samples_timed"] == 256 * 160 * 10
    assert abs(line["value"] - cfg["samples_timed"] / (line["ms_per_step"] * 5 * 1e-3) / 1e6) < 1e-2 * line["value"]
    # the strong-scaling region of the same command is timed as well, as a second figure
    assert cfg["strong_scaling"]["frames_timed"] == 5 and cfg["strong_scaling"]["value"] > 0
