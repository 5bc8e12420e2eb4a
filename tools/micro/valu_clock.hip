// Micro-benchmark: the price of a wave64 VALU instruction in SHADER CLOCK CYCLES, measured by the shader's own cycle counter
// (s_memtime via clock64()) beside the constant 100 MHz counter (s_memrealtime via wall_clock64()), so that the figure does not
// rest on an assumed frequency (tools/micro/valu_cost.hip converts event times at a nominal 2.4 GHz).  Every wave runs the same
// independent instruction stream; per kernel: cycles per wave-instruction per SIMD = mean over waves of (cycles a wave was
// resident) / (instructions of one wave x waves per SIMD), and the clock the shader ran at = cycles / wall time.
// Every CU holds EXACTLY N workgroups of 256 threads (one wave per SIMD each): a launch of N x CUs workgroups that each ask for
// 160 KiB / N of LDS cannot be placed any other way.  (Without that the dispatcher spreads the workgroups unevenly — some CUs get
// N + 2, some N - 2 — and the kernel's time is the fullest CU's: the 2.7 "cycles" of profiles/r04_valu_cost_microbench.txt.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define K8(OP, ARGS) asm volatile(OP " %0, %0, " ARGS "\n " OP " %1, %1, " ARGS "\n " OP " %2, %2, " ARGS "\n " OP " %3, %3, " ARGS "\n " \
                                  OP " %4, %4, " ARGS "\n " OP " %5, %5, " ARGS "\n " OP " %6, %6, " ARGS "\n " OP " %7, %7, " ARGS \
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d))
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, long long* stamps, int iters)
{
    extern __shared__ float lds[];
    if (iters < 0) lds[threadIdx.x] = 1.0f;  // (never: the allocation is what matters)
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c = 1.0001f, d = 0.5f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(K8("v_fma_f32", "%8, %9");) }
        else if (KIND == 1) { REP16(K8("v_max3_f32", "%8, %9");) }
        else if (KIND == 2) {  // alternating fma / max3, 4 + 4
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 3) { REP16(K8("v_add_f32", "%8");) }
        else if (KIND == 4) {  // 6 fma + 2 max3
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 5) {  // 2 fma + 6 max3
            REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 6) {  // v_cvt_f32_ubyte0..3 (what the node decode converts with)
            REP16(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                               "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 7) {  // 4 fma + 4 cvt, alternating
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cvt_f32_ubyte1 %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_cvt_f32_ubyte3 %3, %8\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_cvt_f32_ubyte1 %5, %9\n v_fma_f32 %6, %6, %8, %9\n v_cvt_f32_ubyte3 %7, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 8) { REP16(K8("v_and_b32", "%8");) }
        else if (KIND == 9) {  // v_cndmask_b32 with vcc
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : );)
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * wave] = t1 - t0;
        stamps[2 * wave + 1] = w1 - w0;
    }
}

template <int KIND>
static void run(const char* name, float* d, long long* stamps, int cus)
{
    for (int wavesPerSimd : {1, 2, 3, 4, 5, 8}) {
        const int blocks = cus * wavesPerSimd, waves = blocks * 4, iters = 20000;
        const size_t ldsBytes = (size_t)(160 * 1024 / wavesPerSimd) & ~(size_t)1023;  // exactly wavesPerSimd workgroups fit a CU
        hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        k<KIND><<<blocks, 256, ldsBytes>>>(d, stamps, 10); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<KIND><<<blocks, 256, ldsBytes>>>(d, stamps, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * waves);
        hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0, cycMax = 0, cycMin = 1e30;
        for (int w = 0; w < waves; w++) {
            cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1];
            cycMax = cyc > 0 && (double)h[2 * w] > cycMax ? (double)h[2 * w] : cycMax;
            cycMin = (double)h[2 * w] < cycMin ? (double)h[2 * w] : cycMin;
        }
        cyc /= waves; wall /= waves;
        const double inst = (double)iters * 16 * 8;
        printf("%-28s waves/SIMD %d  kernel %8.3f ms  wave resident %9.0f cycles (min %9.0f max %9.0f) = %7.3f ms -> shader clock %5.3f GHz; "
               "cycles per wave-instruction per SIMD: %5.2f by the cycle counter, %5.2f by event time x the measured clock\n",
               name, wavesPerSimd, ms, cyc, cycMin, cycMax, wall / 1e5, cyc / (wall * 10.0), cyc / (inst * wavesPerSimd), ms * 1e-3 * (cyc / (wall * 1e-8)) / (inst * wavesPerSimd));
    }
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    long long* stamps; hipMalloc(&stamps, (size_t)cus * 8 * 4 * 2 * sizeof(long long));
    printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
    run<0>("v_fma_f32", d, stamps, cus);
    run<3>("v_add_f32", d, stamps, cus);
    run<1>("v_max3_f32", d, stamps, cus);
    run<2>("v_fma_f32 / v_max3_f32 4+4", d, stamps, cus);
    run<4>("v_fma_f32 / v_max3_f32 6+2", d, stamps, cus);
    run<5>("v_fma_f32 / v_max3_f32 2+6", d, stamps, cus);
    run<6>("v_cvt_f32_ubyteN", d, stamps, cus);
    run<7>("v_fma_f32 / v_cvt 4+4", d, stamps, cus);
    run<8>("v_and_b32", d, stamps, cus);
    run<9>("v_cndmask_b32 vcc", d, stamps, cus);
    return 0;
}
