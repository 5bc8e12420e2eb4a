"""N > 1 path on CPU: world_size-2 gloo run of the tile split + gather + root accumulate, against a single-rank render."""
import os
import subprocess
import sys

import numpy as np

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


def test_two_rank_gloo_gather_equals_single_rank(tmp_path):
    out = str(tmp_path / "acc.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611",
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
