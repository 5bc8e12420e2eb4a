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
    must equal the 1-rank PNG byte for byte (pixel-keyed RNG)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NX_BENCH_BACKEND="gloo", NX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "5", "--warmup", "2", "--reps", "1", "--frames-per-pass", "2", "--width", "256", "--height", "160",
              "--no-cpu-baseline", "--no-roofline"]
    one, two = str(tmp_path / "one.png"), str(tmp_path / "two.png")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--png", one], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", _free_port(),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + [c for c in common if c != "--no-roofline"] + ["--png", two]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert open(one, "rb").read() == open(two, "rb").read()
    # a distributed run reports every rank's own time, gathered to rank 0 (a bad scaling curve must be readable from one run)
    import json

    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    pr = line["config"]["per_rank"]
    assert len(pr["median_ms_by_rank"]) == 2 and pr["slowest_rank"] in (0, 1) and pr["backend"] == "gloo" and line["n_gpus"] == 2
    # ... and rank 0's roofline block (its launches on its half of the pixels; rank 1 waits at the closing barrier meanwhile)
    rf = line["roofline"]
    assert rf["scope"].startswith("rank 0 of 2") and rf["rays_per_launch"] > 0 and "cpu_baseline" not in line
