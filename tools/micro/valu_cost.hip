// Micro-benchmark: what does ONE wave64 VALU instruction of a given kind cost on a gfx950 SIMD when several waves share it —
// the regime of the trace kernel (5 waves per SIMD, bound by VALU issue: DESIGN.md section 6)?  Candidates for re-writing the
// node decode (213 instructions: 48 v_cvt_f32_ubyteN + 48 v_fma_f32 + ...): v_fma_mix_f32 (an f16 operand converted inside the
// FMA: a byte zero-extended to 16 bits is the f16 denormal b * 2^-24, so cvt + fma could become one instruction), v_perm_b32,
// v_mul_lo_u32 (the compiler's `x * 0xff`), the integer 3-operand forms.  Independent instruction streams, inline assembly so that
// the instruction measured is the instruction written.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b0 = threadIdx.x * 2654435761u, b1 = b0 ^ 0x9e3779b9u;
    float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (KIND == 1) {  // v_fma_mix_f32, src0 = f16 (low half of b0 / high half of b1)
            REP16(asm volatile("v_fma_mix_f32 %0, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %11, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %11, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %11, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d), "v"(b0 & 0x00ff00ffu), "v"(b1 & 0x00ff00ffu));)
        } else if (KIND == 2) {  // v_cvt_f32_ubyteN
            REP16(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                               "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (KIND == 3) {  // v_perm_b32
            REP16(asm volatile("v_perm_b32 %0, %8, %9, %10\n v_perm_b32 %1, %8, %9, %10\n v_perm_b32 %2, %8, %9, %10\n v_perm_b32 %3, %8, %9, %10\n"
                               "v_perm_b32 %4, %8, %9, %10\n v_perm_b32 %5, %8, %9, %10\n v_perm_b32 %6, %8, %9, %10\n v_perm_b32 %7, %8, %9, %10"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(0x0c010c00u));)
        } else if (KIND == 4) {  // v_mul_lo_u32
            REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                               "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0 | 1u));)
        } else if (KIND == 5) {  // v_max3_i32
            REP16(asm volatile("v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %1, %1, %8, %9\n v_max3_i32 %2, %2, %8, %9\n v_max3_i32 %3, %3, %8, %9\n"
                               "v_max3_i32 %4, %4, %8, %9\n v_max3_i32 %5, %5, %8, %9\n v_max3_i32 %6, %6, %8, %9\n v_max3_i32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (KIND == 6) {  // v_lshl_or_b32
            REP16(asm volatile("v_lshl_or_b32 %0, %8, %9, %0\n v_lshl_or_b32 %1, %8, %9, %1\n v_lshl_or_b32 %2, %8, %9, %2\n v_lshl_or_b32 %3, %8, %9, %3\n"
                               "v_lshl_or_b32 %4, %8, %9, %4\n v_lshl_or_b32 %5, %8, %9, %5\n v_lshl_or_b32 %6, %8, %9, %6\n v_lshl_or_b32 %7, %8, %9, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1 & 31u));)
        } else if (KIND == 7) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3)
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
        } else if (KIND == 8) {  // v_cmp_le_f32 into vcc
            REP16(asm volatile("v_cmp_le_f32 vcc, %0, %8\n v_cmp_le_f32 vcc, %1, %8\n v_cmp_le_f32 vcc, %2, %8\n v_cmp_le_f32 vcc, %3, %8\n"
                               "v_cmp_le_f32 vcc, %4, %8\n v_cmp_le_f32 vcc, %5, %8\n v_cmp_le_f32 vcc, %6, %8\n v_cmp_le_f32 vcc, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
        } else if (KIND == 9) {  // v_pk_fma_f32 (two FMAs per instruction)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                               : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&c), "v"(*(double*)&d));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}


#define K8(OP, A) asm volatile(OP " %0, " A "\n" OP " %1, " A "\n" OP " %2, " A "\n" OP " %3, " A "\n" OP " %4, " A "\n" OP " %5, " A "\n" OP " %6, " A "\n" OP " %7, " A \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(sm))
template <int KIND>
__global__ void __launch_bounds__(256) k2(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b0 = threadIdx.x * 2654435761u | 0x3f000000u, b1 = (b0 ^ 0x9e3779b9u) & 15u;
    const unsigned long long sm = 0x5555aaaa3333ccccull + (unsigned long long)iters;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(K8("v_mul_f32", "%8, %8");) }
        else if (KIND == 1) { REP16(K8("v_add_f32", "%8, %8");) }
        else if (KIND == 2) { REP16(K8("v_max_f32", "%8, %8");) }
        else if (KIND == 3) { REP16(K8("v_max3_f32", "%8, %8, %8");) }
        else if (KIND == 4) { REP16(K8("v_and_b32", "%8, %9");) }
        else if (KIND == 5) { REP16(K8("v_lshlrev_b32", "%9, %8");) }
        else if (KIND == 6) { REP16(K8("v_bfe_u32", "%8, %9, 8");) }
        else if (KIND == 7) { REP16(K8("v_add_u32", "%8, %9");) }
        else if (KIND == 8) { REP16(K8("v_cndmask_b32", "%8, %9, %10");) }
        else if (KIND == 9) { REP16(K8("v_mov_b32", "%8");) }
        else if (KIND == 10) { REP16(K8("v_rcp_f32", "%8");) }
        else if (KIND == 11) { REP16(K8("v_cvt_f32_u32", "%8");) }
        else if (KIND == 12) { REP16(K8("v_sub_f32", "%8, %8");) }
        else if (KIND == 13) { REP16(K8("v_or3_b32", "%8, %9, %8");) }
        else if (KIND == 14) { REP16(K8("v_min3_i32", "%8, %9, %8");) }
        else if (KIND == 15) { REP16(K8("v_fmac_f32", "%8, %8");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
static void run2(const char* name, float* d, hipEvent_t e0, hipEvent_t e1)
{
    for (int wavesPerSimd : {1, 5}) {
        const int blocks = 256 * wavesPerSimd;
        k2<KIND><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k2<KIND><<<blocks, 256>>>(d, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = 2000.0 * 16 * 8;
        printf("%-44s waves/SIMD %d  %8.3f ms  -> %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / (inst * wavesPerSimd));
    }
}

template <int KIND>
static void run(const char* name, float* d, hipEvent_t e0, hipEvent_t e1)
{
    for (int wavesPerSimd : {1, 5}) {
        const int blocks = 256 * wavesPerSimd;
        k<KIND><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = 2000.0 * 16 * 8;
        printf("%-44s waves/SIMD %d  %8.3f ms  -> %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / (inst * wavesPerSimd));
    }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<0>("v_fma_f32", d, e0, e1);
    run<1>("v_fma_mix_f32 (f16 src0, both halves)", d, e0, e1);
    run<2>("v_cvt_f32_ubyte0..3", d, e0, e1);
    run<3>("v_perm_b32", d, e0, e1);
    run<4>("v_mul_lo_u32", d, e0, e1);
    run<5>("v_max3_i32", d, e0, e1);
    run<6>("v_lshl_or_b32", d, e0, e1);
    run<7>("v_cndmask_b32 (vcc)", d, e0, e1);
    run<8>("v_cmp_le_f32 -> vcc", d, e0, e1);
    run<9>("v_pk_fma_f32 (2 FMAs each)", d, e0, e1);
    run2<0>("v_mul_f32", d, e0, e1); run2<1>("v_add_f32", d, e0, e1); run2<12>("v_sub_f32", d, e0, e1); run2<15>("v_fmac_f32 (VOP2)", d, e0, e1);
    run2<2>("v_max_f32", d, e0, e1); run2<3>("v_max3_f32", d, e0, e1); run2<14>("v_min3_i32", d, e0, e1);
    run2<4>("v_and_b32", d, e0, e1); run2<5>("v_lshlrev_b32", d, e0, e1); run2<6>("v_bfe_u32", d, e0, e1); run2<7>("v_add_u32", d, e0, e1); run2<13>("v_or3_b32", d, e0, e1);
    run2<8>("v_cndmask_b32 (SGPR-pair mask)", d, e0, e1); run2<9>("v_mov_b32", d, e0, e1); run2<10>("v_rcp_f32", d, e0, e1); run2<11>("v_cvt_f32_u32", d, e0, e1);
    return 0;
}
