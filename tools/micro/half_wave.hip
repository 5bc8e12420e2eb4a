// Micro-benchmark: does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty (gfx950 SIMD-32)?
// Every wave runs a long dependent-free FMA stream under an exec mask chosen by `mode`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long mask)
{
    const int lane = threadIdx.x & 63;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
                a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; unsigned long long mask; } modes[] = {{"all 64 lanes", ~0ull}, {"lower half (0-31)", 0xffffffffull}, {"upper half (32-63)", 0xffffffff00000000ull},
        {"even lanes", 0x5555555555555555ull}, {"lanes 0-15", 0xffffull}, {"one lane", 1ull}, {"lanes 0-31 + lane 32", 0x1ffffffffull}};
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;  // 256-thread blocks = 4 waves = 1 per SIMD
        for (auto& m : modes) {
            k<<<blocks, 256>>>(d, 10, m.mask); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<blocks, 256>>>(d, 2000, m.mask); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double inst = 2000.0 * 16 * 8;  // VALU instructions per wave
            printf("waves/SIMD %d  %-22s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", wavesPerSimd, m.name, ms, ms * 1e-3 * 2.4e9 / (inst * wavesPerSimd));
        }
    }
    return 0;
}
