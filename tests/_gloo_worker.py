"""Worker of tests/test_multi_rank_gloo.py: one rank of the N > 1 path on CPU (gloo).  Each rank renders its tiles with
the CPU oracle (standing in for the HIP path, which needs a GPU).  Two exchange schemes, both with ONE gather per pass:
  "tiles"    (what bench.py runs over RCCL) every rank accumulates its own tiles; the accumulated tiles are gathered and
             scattered into the full image on rank 0;
  "radiance" the per-frame radiance tiles are gathered and rank 0 accumulates (nxhip_accumulate_external)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nexus_amd import multigpu, pod  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tests import scene_helpers as SH  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, FRAMES = 48, 40, 2
    tile = multigpu.tile_rows_for(H, world)
    scene = SH.cornell_scene(W, H, path_length=3)
    pm = multigpu.tile_pixel_map(W, H, rank, world, tile)
    w = O.Wavefront(scene.oracle(), len(pm), pm, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_REFERENCE)
    acc = None          # scheme "radiance": root-side accumulation of gathered radiance
    composed = None     # scheme "tiles": gathered per-rank accumulations
    for f in range(1, FRAMES + 1):
        w.render(f)
        w.accumulate(f)
        rad = torch.from_numpy(w.radiance().copy())
        gathered = [torch.zeros_like(rad) for _ in range(world)] if rank == 0 else None
        dist.gather(rad, gathered, dst=0)
        if rank == 0:
            full = multigpu.reassemble(W, H, world, tile, [g.numpy() for g in gathered])
            acc = multigpu.running_mean(acc, full, f)
        mine = torch.from_numpy(w.accumulation().copy())
        tiles = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, tiles, dst=0)
        if rank == 0:
            composed = multigpu.reassemble(W, H, world, tile, [t.numpy() for t in tiles])
    if rank == 0:
        np.save(out_path, np.stack([acc, composed]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
