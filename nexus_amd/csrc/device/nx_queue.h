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

// A PRODUCER workgroup's region: its index modulo the regions in use (1 with ordered compaction: everything in region 0).
NXD int producer_region(const DeviceState* S) { return S->queueShards > 1u ? (int)(blockIdx.x & (kQueueShards - 1)) : 0; }

// The dense numbering generate_kernel and the ray-batch hooks use for `count` items: cut into contiguous, 64-aligned pieces, one
// per region in use.  piece = items per region.
NXD uint32_t dense_piece(uint32_t count, uint32_t shards) { return ((count + shards * kWave - 1u) / (shards * kWave)) * kWave; }

}  // namespace nxd
