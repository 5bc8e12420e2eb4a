// Micro-benchmark: what dependent per-lane record gathers does one MI355X sustain?  It is the memory access pattern of the
// traversal kernel with everything else taken away: every active lane follows its own chain of records (next index stored in
// the record), K 16-byte loads per record exactly as nx_trace.hip fetches a node (K = 5) or a triangle (K = 3), `valu`
// filler FMAs per step standing in for the slab / triangle test.
//   gather <footprint MiB> <stride B> <K> <active lanes> <waves/SIMD> <valu per step> <hot fraction %> <hot KiB>
// prints lane-loads per ns per CU (the quantity TCP_TOTAL_CACHE_ACCESSES / CU / ns measures on the real kernel), records per
// ns for the chip, and the time one step of one wave takes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int K>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ recs, uint32_t strideChunks, uint32_t nrec, int steps, int activeLanes, int valu, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    uint32_t idx = (uint32_t)((blockIdx.x * 256u + threadIdx.x) * 2654435761u) % nrec;
    uint32_t acc = 0;
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    if (lane < activeLanes) {
        for (int s = 0; s < steps; s++) {
            const uint4* p = recs + (size_t)idx * strideChunks;
            uint4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = p[k];
            uint32_t nxt = v[0].x;
#pragma unroll
            for (int k = 1; k < K; k++) { nxt += v[k].y; acc ^= v[k].z; }  // .y of the later chunks is 0: the step waits for all K loads
            idx = nxt;
            for (int i = 0; i < valu; i += 4) {
                f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + idx + (uint32_t)(f0 + f1 + f2 + f3);
}

// The same chains fetched cooperatively: the wave's active lanes publish their record addresses (shuffle), consecutive
// lanes then load consecutive 16-byte chunks of one record straight into LDS (global_load_lds, 12.8 records per instruction,
// neighbouring lanes in the same cache line), and every lane reads its record back from LDS.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_cvoid;
template <int K>
__global__ void __launch_bounds__(256) gather_coop(const uint4* __restrict__ recs, uint32_t strideChunks, uint32_t nrec, int steps, int activeLanes, int valu, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[4 * 64 * K * 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* const myStage = stage + wave * 64 * K * 4;
    uint32_t idx = (uint32_t)((blockIdx.x * 256u + threadIdx.x) * 2654435761u) % nrec;
    uint32_t acc = 0;
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    const int total = activeLanes * K;  // the active lanes are lanes [0, activeLanes): rank == lane
    for (int s = 0; s < steps; s++) {
        const unsigned long long addr = (unsigned long long)(recs + (size_t)idx * strideChunks);
        for (int base = 0; base < total; base += 64) {
            const int c = base + lane;
            const int r = c / K, part = c - r * K;
            const unsigned long long a = __shfl(addr, r & 63) + (unsigned long long)(part * 16);
            if (c < total) __builtin_amdgcn_global_load_lds((g_cvoid*)a, (lds_void*)(myStage + base * 4), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < activeLanes) {
            const uint4* p = (const uint4*)(myStage + lane * K * 4);
            uint4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = p[k];
            uint32_t nxt = v[0].x;
#pragma unroll
            for (int k = 1; k < K; k++) { nxt += v[k].y; acc ^= v[k].z; }
            idx = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < valu; i += 4) {
            f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + idx + (uint32_t)(f0 + f1 + f2 + f3);
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }

int main(int argc, char** argv)
{
    const double mib = argc > 1 ? atof(argv[1]) : 110;
    const int stride = argc > 2 ? atoi(argv[2]) : 80;
    const int K = argc > 3 ? atoi(argv[3]) : 5;
    const int lanes = argc > 4 ? atoi(argv[4]) : 64;
    const int wps = argc > 5 ? atoi(argv[5]) : 6;
    const int valu = argc > 6 ? atoi(argv[6]) : 0;
    const int hotPct = argc > 7 ? atoi(argv[7]) : 0;
    const int hotKiB = argc > 8 ? atoi(argv[8]) : 16;
    const bool coop = argc > 9 && atoi(argv[9]) != 0;
    if (K * 16 > stride || stride % 16) { fprintf(stderr, "K * 16 must fit the stride\n"); return 2; }
    const uint32_t nrec = (uint32_t)(mib * 1048576.0 / stride);
    const uint32_t hotN = (uint32_t)(hotKiB * 1024 / stride);
    std::vector<uint32_t> host((size_t)nrec * (stride / 4), 0u);
    for (uint32_t i = 0; i < nrec; i++) {
        const bool hot = (int)(rnd() % 100u) < hotPct;
        host[(size_t)i * (stride / 4)] = hot ? rnd() % hotN : rnd() % nrec;
        for (int k = 1; k < stride / 16; k++) host[(size_t)i * (stride / 4) + 4 * k + 2] = rnd();
    }
    uint4* d; uint32_t* out;
    if (hipMalloc(&d, host.size() * 4) != hipSuccess) return 1;
    hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * wps;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 400;
    auto launch = [&](int st) {
        if (coop) {
            if (K == 3) gather_coop<3><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out);
            else gather_coop<5><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out);
            return;
        }
        switch (K) {
        case 1: gather<1><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out); break;
        case 3: gather<3><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out); break;
        case 4: gather<4><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out); break;
        case 5: gather<5><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out); break;
        default: gather<8><<<blocks, 256>>>(d, stride / 16, nrec, st, lanes, valu, out); break;
        }
    };
    launch(20); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0); launch(steps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6;
    const double records = (double)blocks * 4 * lanes * steps;
    printf("%s footprint %7.1f MiB stride %3d K %d lanes %2d waves/SIMD %d valu %3d hot %2d%%/%dKiB : %7.3f ms  %6.3f lane-loads/ns/CU  %7.2f records/ns  step %6.0f ns\n",
           coop ? "coop  " : "direct", mib, stride, K, lanes, wps, valu, hotPct, hotKiB, best, records * K / ns / 256.0, records / ns, ns / steps);
    return 0;
}
