// atomics.hip — how many returning device-scope atomicAdds per microsecond does MI355X sustain, as a function of how many
// distinct words they go to and how far apart those words are?  (The slot allocation of the logic / material kernels and the
// ray fetch of the trace kernels hand out queue slots with such atomics: DESIGN.md section 6.)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/atomics.hip -o /tmp/atomics && /tmp/atomics
// Every workgroup (256 threads) issues `iters` atomics from ONE thread, each followed by a workgroup barrier and a dependent
// use of the result — the pattern of SlotAllocator::alloc.  `words` = distinct counters (workgroup w uses word w % words),
// `stride` = bytes between them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) atomics_kernel(int* base, int words, size_t strideInts, int iters, int* sink)
{
    __shared__ int got;
    int* word = base + (size_t)(blockIdx.x % words) * strideInts;
    int acc = 0;
    for (int i = 0; i < iters; i++) {
        if (threadIdx.x == 0) got = atomicAdd(word, 1);
        __syncthreads();
        acc += got;
        __syncthreads();
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main()
{
    const int cus = 256, wgPerCu = 5, iters = 200;
    const int grid = cus * wgPerCu;
    const size_t span = (size_t)64 << 20;  // 64 words up to 1 MiB apart
    int* buf = nullptr;
    int* sink = nullptr;
    if (hipMalloc(&buf, span + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::printf("%d workgroups x %d atomics each (one per barrier interval)\n", grid, iters);
    std::printf("%8s %10s %12s %14s\n", "words", "stride B", "ms", "atomics/us");
    const int wordCounts[] = {1, 2, 8, 64};
    const size_t strides[] = {4, 128, 4096, 65536, 1 << 20};
    for (int words : wordCounts)
        for (size_t stride : strides) {
            if (words == 1 && stride != 4) continue;
            hipMemset(buf, 0, span + 4096);
            atomics_kernel<<<grid, 256>>>(buf, words, stride / 4, 10, sink);  // warm-up
            hipDeviceSynchronize();
            hipEventRecord(a);
            atomics_kernel<<<grid, 256>>>(buf, words, stride / 4, iters, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, a, b);
            std::printf("%8d %10zu %12.3f %14.1f\n", words, stride, ms, (double)grid * iters / (ms * 1e3));
        }
    hipFree(buf);
    hipFree(sink);
    return 0;
}
