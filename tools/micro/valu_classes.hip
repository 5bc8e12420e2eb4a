// Micro-benchmark: which wave64 VALU opcodes issue at the full rate on a gfx950 SIMD and which go through the slower unit —
// measured with EIGHT waves per SIMD (32 waves per CU = every wave slot: the only occupancy at which the placement of waves on
// SIMDs is forced to be even; at 3-5 workgroups per CU the dispatcher's uneven placement inflates the figure, which is what
// profiles/r04_valu_cost_microbench.txt's "2.7 cycles" were) and converted with the shader clock measured in the kernel
// (s_memtime cycles / s_memrealtime time), not a nominal frequency.  Independent instruction streams, inline assembly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define K8(OP, ARGS) asm volatile(OP " %0, " ARGS "\n " OP " %1, " ARGS "\n " OP " %2, " ARGS "\n " OP " %3, " ARGS "\n " \
                                  OP " %4, " ARGS "\n " OP " %5, " ARGS "\n " OP " %6, " ARGS "\n " OP " %7, " ARGS \
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d), "v"(e), "s"(sm))
template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, long long* stamps, int iters, unsigned long long sm)
{
    uint32_t a0 = threadIdx.x * 2654435761u, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t c = 0x3f800123u, d = 0x3f000456u, e = 5u;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(K8("v_fma_f32", "%8, %9, %9");) }
        else if (KIND == 1) { REP16(K8("v_bfi_b32", "%8, %9, %10");) }
        else if (KIND == 2) { REP16(K8("v_bitop3_b32", "%8, %9, %10 bitop3:0xca");) }
        else if (KIND == 3) { REP16(K8("v_ashrrev_i32", "31, %8");) }
        else if (KIND == 4) { REP16(K8("v_lshrrev_b32", "%10, %8");) }
        else if (KIND == 5) { REP16(K8("v_lshlrev_b32", "%10, %8");) }
        else if (KIND == 6) { REP16(K8("v_xor_b32", "%8, %9");) }
        else if (KIND == 7) { REP16(K8("v_bfe_u32", "%8, 5, 3");) }
        else if (KIND == 8) { REP16(K8("v_cndmask_b32", "%8, %9, %11");) }
        else if (KIND == 9) { REP16(K8("v_and_or_b32", "%8, %9, %10");) }
        else if (KIND == 10) { REP16(K8("v_or3_b32", "%8, %9, %10");) }
        else if (KIND == 11) { REP16(K8("v_max_i32", "%8, %9");) }
        else if (KIND == 12) { REP16(K8("v_min_f32", "%8, %9");) }
        else if (KIND == 13) { REP16(K8("v_max3_i32", "%8, %9, %10");) }
        else if (KIND == 14) { REP16(K8("v_mul_lo_u32", "%8, %9");) }
        else if (KIND == 15) { REP16(K8("v_mul_u32_u24", "%8, %9");) }
        else if (KIND == 16) { REP16(K8("v_add_u32", "%8, %9");) }
        else if (KIND == 17) { REP16(K8("v_sub_u32", "%8, %9");) }
        else if (KIND == 18) { REP16(K8("v_lshl_add_u32", "%8, 1, %9");) }
        else if (KIND == 19) { REP16(K8("v_add3_u32", "%8, %9, %10");) }
        else if (KIND == 20) { REP16(K8("v_perm_b32", "%8, %9, %10");) }
        else if (KIND == 21) { REP16(K8("v_alignbit_b32", "%8, %9, %10");) }
        else if (KIND == 22) { REP16(K8("v_cvt_f32_ubyte1", "%8");) }
        else if (KIND == 23) { REP16(K8("v_mul_f32", "%8, %9");) }
        else if (KIND == 24) { REP16(K8("v_sub_f32", "%8, %9");) }
        else if (KIND == 25) { REP16(K8("v_or_b32", "%8, %9");) }
        else if (KIND == 26) { REP16(K8("v_and_b32", "%8, %9");) }
        else if (KIND == 27) { REP16(K8("v_mov_b32", "%8");) }
        else if (KIND == 28) { REP16(K8("v_cmp_le_f32 vcc,", "%8");) }  // (writes vcc, reads a_k and c)
        else if (KIND == 29) { REP16(K8("v_med3_f32", "%8, %9, %10");) }
        else if (KIND == 30) { REP16(K8("v_xad_u32", "%8, %9, %10");) }
        else if (KIND == 31) { REP16(K8("v_mad_u32_u24", "%8, %9, %10");) }
        else if (KIND == 32) { REP16(K8("v_lshl_or_b32", "%8, %10, %9");) }
        else if (KIND == 33) { REP16(K8("v_cvt_f32_u32", "%8");) }
        else if (KIND == 34) { REP16(K8("v_rcp_f32", "%8");) }
        else if (KIND == 35) { REP16(K8("v_max_f32", "%8, %9");) }
        else if (KIND == 36) { REP16(K8("v_ashrrev_i32", "%10, %8");) }
        else if (KIND == 37) { REP16(K8("v_bfm_b32", "%8, %9");) }
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
static void run(const char* name, uint32_t* d, long long* stamps, int cus)
{
    const int wavesPerSimd = 8, blocks = cus * wavesPerSimd, waves = blocks * 4, iters = 8000;
    k<KIND><<<blocks, 256>>>(d, stamps, 10, 0x5555555555555555ull); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, stamps, iters, 0x5555555555555555ull); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves);
    (void)hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int w = 0; w < waves; w++) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
    const double clockGHz = cyc / (wall * 10.0), inst = (double)iters * 16 * 8;
    printf("%-22s 8 waves per SIMD: kernel %7.3f ms at %5.3f GHz -> %5.2f cycles per wave-instruction per SIMD\n", name, ms, clockGHz, ms * 1e-3 * clockGHz * 1e9 / (inst * wavesPerSimd));
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    long long* stamps; (void)hipMalloc(&stamps, (size_t)cus * 8 * 4 * 2 * sizeof(long long));
    printf("%s, %d CUs\n", p.gcnArchName, cus);
    run<0>("v_fma_f32", d, stamps, cus); run<23>("v_mul_f32", d, stamps, cus); run<24>("v_sub_f32", d, stamps, cus);
    run<26>("v_and_b32", d, stamps, cus); run<25>("v_or_b32", d, stamps, cus); run<6>("v_xor_b32", d, stamps, cus); run<27>("v_mov_b32", d, stamps, cus);
    run<16>("v_add_u32", d, stamps, cus); run<17>("v_sub_u32", d, stamps, cus);
    run<4>("v_lshrrev_b32", d, stamps, cus); run<3>("v_ashrrev_i32 31", d, stamps, cus); run<36>("v_ashrrev_i32 v", d, stamps, cus); run<5>("v_lshlrev_b32", d, stamps, cus);
    run<1>("v_bfi_b32", d, stamps, cus); run<2>("v_bitop3_b32", d, stamps, cus); run<37>("v_bfm_b32", d, stamps, cus);
    run<7>("v_bfe_u32", d, stamps, cus); run<8>("v_cndmask_b32 sgpr", d, stamps, cus); run<9>("v_and_or_b32", d, stamps, cus); run<10>("v_or3_b32", d, stamps, cus);
    run<18>("v_lshl_add_u32", d, stamps, cus); run<32>("v_lshl_or_b32", d, stamps, cus); run<19>("v_add3_u32", d, stamps, cus); run<30>("v_xad_u32", d, stamps, cus);
    run<11>("v_max_i32", d, stamps, cus); run<12>("v_min_f32", d, stamps, cus); run<35>("v_max_f32", d, stamps, cus); run<13>("v_max3_i32", d, stamps, cus); run<29>("v_med3_f32", d, stamps, cus);
    run<14>("v_mul_lo_u32", d, stamps, cus); run<15>("v_mul_u32_u24", d, stamps, cus); run<31>("v_mad_u32_u24", d, stamps, cus);
    run<20>("v_perm_b32", d, stamps, cus); run<21>("v_alignbit_b32", d, stamps, cus);
    run<22>("v_cvt_f32_ubyte1", d, stamps, cus); run<33>("v_cvt_f32_u32", d, stamps, cus); run<28>("v_cmp_le_f32 vcc", d, stamps, cus); run<34>("v_rcp_f32", d, stamps, cus);
    return 0;
}
