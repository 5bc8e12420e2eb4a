"""Frame partition for the multi-GPU path: interleaved row tiles, scene replicated, one gather per pass.

The reference is single-GPU (no NCCL/MPI anywhere under /root/reference/Nexus/src); this is the added data-parallel
layer.  Pixels are independent, so rank r renders the rows whose tile index (row // tile_rows) is congruent to r modulo
the world size, with its own queues sized for its share.  With the pixel-keyed RNG the reassembled image is bit-identical
to the single-GPU image.
"""
import numpy as np


def tile_rows_for(height, world, preferred=(5, 8, 4, 6, 3, 2, 1)):
    """A tile height for which every rank gets the same number of rows (RCCL gather wants equal counts)."""
    for t in preferred:
        if height % (t * world) == 0:
            return t
    raise ValueError("no tile height divides %d rows evenly over %d ranks" % (height, world))


def tile_pixel_map(width, height, rank, world, tile_rows):
    """Global pixel index of every local pixel of `rank`, in local order (ascending rows, ascending x)."""
    rows = np.arange(height)
    mine = rows[(rows // tile_rows) % world == rank]
    return (mine[:, None].astype(np.int64) * width + np.arange(width)[None, :]).reshape(-1).astype(np.uint32)


def reassemble(width, height, world, tile_rows, tiles):
    """tiles[r]: (n_local, C) array rendered by rank r -> (width*height, C) image in pixel order."""
    out = np.zeros((width * height,) + tuple(tiles[0].shape[1:]), dtype=tiles[0].dtype)
    seen = np.zeros(width * height, dtype=bool)
    for r in range(world):
        pm = tile_pixel_map(width, height, r, world, tile_rows)
        assert len(pm) == len(tiles[r]) and not seen[pm].any()
        seen[pm] = True
        out[pm] = tiles[r]
    assert seen.all()
    return out


def running_mean(acc, radiance, frame):
    """AccumulateKernel's update (/root/reference/Nexus/src/Cuda/PathTracer/PathTracer.cu:489-492) in float32."""
    if frame == 1:
        return radiance.astype(np.float32).copy()
    return (acc + (radiance - acc) / np.float32(frame)).astype(np.float32)


def tiled_order(pixel_map, width, tile_w=8, tile_h=8):
    """Reorder a rank's pixels tile by tile (tile_w x tile_h blocks of the image, row-major inside a tile) so that the 64
    rays a wave fetches together are a compact block of the image instead of a 64 x 1 strip.  Any order is legal: the
    pixel map is what tells the device which global pixel a path belongs to, and with the pixel-keyed RNG the image does
    not depend on it."""
    pm = np.asarray(pixel_map, dtype=np.int64)
    y, x = pm // width, pm % width
    # rows of one rank are not contiguous under the interleaved split: tile over the rank-local row index
    rows = np.unique(y)
    local_row = np.searchsorted(rows, y)
    key = ((local_row // tile_h) * ((width + tile_w - 1) // tile_w) + (x // tile_w)) * (tile_w * tile_h) + (local_row % tile_h) * tile_w + (x % tile_w)
    return pm[np.argsort(key, kind="stable")].astype(np.uint32)
