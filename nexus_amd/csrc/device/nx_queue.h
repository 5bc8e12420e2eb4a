// nx_queue.h — the regions of a queue (nx_device.h, Counters) as the kernels address them.
#pragma once
#include "nx_device.h"
#include "nx_math.h"

namespace nxd {

// One queue at one bounce as a CONSUMER sees it: items are numbered 0 .. total - 1 through the regions in turn, item c lives
// in slot region * cap + offset.  Eight sizes, read once per kernel.
struct QueueView {
    int total;
    int cap;
    int end[kQueueShards];  // end[k] = items in regions 0 .. k
    NXD int slot(int c) const
    {
        int region = 0, before = 0;
#pragma unroll
        for (int k = 0; k < kQueueShards - 1; k++) {
            const bool past = c >= end[k];
            region += past ? 1 : 0;
            before = past ? end[k] : before;
        }
        return region * cap + (c - before);
    }
};

template <class SizePtr>  // (global-address-space or generic pointer to region 0's size word; region k's is kRegionStride * k on)
NXD QueueView queue_view(SizePtr size0, uint32_t cap)
{
    QueueView v;
    v.cap = (int)cap;
    int run = 0;
#pragma unroll
    for (int k = 0; k < kQueueShards; k++) {
        run += size0[k * kRegionStride];
        v.end[k] = run;
    }
    v.total = run;
    return v;
}

// Which region a PRODUCER's tile appends to: the tiles of its input queue are cut into kQueueShards contiguous runs, run r
// feeds region r.  Contiguous, not round robin: the regions are what the trace kernels' XCD groups walk, and an XCD whose rays
// come from one stretch of the previous queue — one part of the image, by and large — keeps the subtrees it touches in its own
// L2 (round-robin tiles: +6 % HBM traffic per ray and -3 % on the 10 M-triangle scene, whose records do not fit the caches).
// A region's fill stays bounded: ceil(tiles / 8) tiles of T items each, i.e. at most M / 8 + T per producer kernel.
struct ProducerRegions {
    int tilesPerRegion, tileItems, shards;
    NXD int of_tile(int firstItemOfTile) const { return shards > 1 ? min((firstItemOfTile / tileItems) / tilesPerRegion, shards - 1) : 0; }
};
NXD ProducerRegions producer_regions(const DeviceState* S, int inputItems, int tileItems)
{
    const int shards = (int)S->queueShards;
    const int tiles = (inputItems + tileItems - 1) / tileItems;
    return ProducerRegions{max(1, (tiles + shards - 1) / shards), tileItems, shards};
}

// The dense numbering generate_kernel and the ray-batch hooks use for `count` items: cut into contiguous, 64-aligned pieces, one
// per region in use.  piece = items per region.
NXD uint32_t dense_piece(uint32_t count, uint32_t shards) { return ((count + shards * kWave - 1u) / (shards * kWave)) * kWave; }

}  // namespace nxd
