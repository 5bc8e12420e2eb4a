// Micro-benchmark: does it matter on a gfx950 SIMD which VGPR BANKS (register number mod 4) the source operands of a VALU
// instruction come from?  tools/micro/valu_cost.hip priced instruction kinds between 2.7 and 4.5 cycles per wave64 instruction at 5
// waves per SIMD without an obvious pattern by opcode (v_fma 2.85, v_fmac 4.07, v_max3_i32 4.36, v_and 2.7, v_lshlrev 4.15): the
// operand registers differed between the kinds.  Here the SAME instruction is timed with explicit registers: three sources in three
// banks, two in one bank, all three in one bank; likewise two-source instructions.  Independent chains (8 destinations), several
// waves per SIMD, as in valu_cost.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#define REP8(X) X X X X X X X X
#define INIT "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 1.0\n v_mov_b32 v67, 1.0\n v_mov_b32 v68, 0.5\n v_mov_b32 v69, 0.5\n v_mov_b32 v70, 0.5\n v_mov_b32 v71, 0.5\n" \
             "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n"

// dst v40+i, src0 = dst; B = second source register, C = third
#define FMA(i, B, C) "v_fma_f32 v" #i ", v" #i ", v" #B ", v" #C "\n"
#define ADD(i, B) "v_add_f32 v" #i ", v" #i ", v" #B "\n"
#define MAX3(i, B, C) "v_max3_i32 v" #i ", v" #i ", v" #B ", v" #C "\n"
#define CVT(i, B) "v_cvt_f32_ubyte1 v" #i ", v" #B "\n"
#define FMAC(i, B, C) "v_fmac_f32 v" #i ", v" #B ", v" #C "\n"              /* VOP2: dst += B * C */
#define FMA3(i, B, C) "v_fma_f32 v" #i ", v" #B ", v" #C ", v" #i "\n"      /* VOP3, the same arithmetic */
#define OP2(op, i, B) op " v" #i ", v" #i ", v" #B "\n"
#define ALL8(M, ...) M(40, __VA_ARGS__) M(41, __VA_ARGS__) M(42, __VA_ARGS__) M(43, __VA_ARGS__) M(44, __VA_ARGS__) M(45, __VA_ARGS__) M(46, __VA_ARGS__) M(47, __VA_ARGS__)
#define OPK(i, op) OP2(op, i, 65)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float r = 0.0f;
    asm volatile(INIT ::: CLOBBERS);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0)  // fma, three banks: dst bank b, sources b, b+1, b+2
            asm volatile(REP8(FMA(40, 65, 70) FMA(41, 66, 71) FMA(42, 67, 68) FMA(43, 64, 69) FMA(44, 65, 70) FMA(45, 66, 71) FMA(46, 67, 68) FMA(47, 64, 69))::: CLOBBERS);
        else if (KIND == 1)  // fma, two sources share a bank (src1 in the bank of src0)
            asm volatile(REP8(FMA(40, 64, 69) FMA(41, 65, 70) FMA(42, 66, 71) FMA(43, 67, 68) FMA(44, 64, 69) FMA(45, 65, 70) FMA(46, 66, 71) FMA(47, 67, 68))::: CLOBBERS);
        else if (KIND == 2)  // fma, all three sources in one bank
            asm volatile(REP8(FMA(40, 64, 68) FMA(41, 65, 69) FMA(42, 66, 70) FMA(43, 67, 71) FMA(44, 64, 68) FMA(45, 65, 69) FMA(46, 66, 70) FMA(47, 67, 71))::: CLOBBERS);
        else if (KIND == 3)  // fma, src1 and src2 the SAME register (one read?)
            asm volatile(REP8(FMA(40, 65, 65) FMA(41, 66, 66) FMA(42, 67, 67) FMA(43, 64, 64) FMA(44, 65, 65) FMA(45, 66, 66) FMA(46, 67, 67) FMA(47, 64, 64))::: CLOBBERS);
        else if (KIND == 4)  // add, two banks
            asm volatile(REP8(ADD(40, 65) ADD(41, 66) ADD(42, 67) ADD(43, 64) ADD(44, 65) ADD(45, 66) ADD(46, 67) ADD(47, 64))::: CLOBBERS);
        else if (KIND == 5)  // add, one bank
            asm volatile(REP8(ADD(40, 64) ADD(41, 65) ADD(42, 66) ADD(43, 67) ADD(44, 64) ADD(45, 65) ADD(46, 66) ADD(47, 67))::: CLOBBERS);
        else if (KIND == 6)  // max3_i32, three banks
            asm volatile(REP8(MAX3(40, 65, 70) MAX3(41, 66, 71) MAX3(42, 67, 68) MAX3(43, 64, 69) MAX3(44, 65, 70) MAX3(45, 66, 71) MAX3(46, 67, 68) MAX3(47, 64, 69))::: CLOBBERS);
        else if (KIND == 7)  // max3_i32, one bank
            asm volatile(REP8(MAX3(40, 64, 68) MAX3(41, 65, 69) MAX3(42, 66, 70) MAX3(43, 67, 71) MAX3(44, 64, 68) MAX3(45, 65, 69) MAX3(46, 66, 70) MAX3(47, 67, 71))::: CLOBBERS);
        else if (KIND == 8)  // cvt_f32_ubyte1 (one source), source in another bank than the destination
            asm volatile(REP8(CVT(40, 65) CVT(41, 66) CVT(42, 67) CVT(43, 64) CVT(44, 65) CVT(45, 66) CVT(46, 67) CVT(47, 64))::: CLOBBERS);
        else if (KIND == 9)  // cvt_f32_ubyte1, source in the destination's bank
            asm volatile(REP8(CVT(40, 64) CVT(41, 65) CVT(42, 66) CVT(43, 67) CVT(44, 64) CVT(45, 65) CVT(46, 66) CVT(47, 67))::: CLOBBERS);
        else if (KIND == 10)  // fma with DEPENDENT chains of length 1: only 2 destinations alternate (latency exposure at this occupancy)
            asm volatile(REP8(FMA(40, 65, 70) FMA(41, 66, 71) FMA(40, 65, 70) FMA(41, 66, 71) FMA(40, 65, 70) FMA(41, 66, 71) FMA(40, 65, 70) FMA(41, 66, 71))::: CLOBBERS);
        else if (KIND == 11)  // v_fmac_f32 (VOP2), what the compiler prefers when the addend dies
            asm volatile(REP8(FMAC(40, 65, 70) FMAC(41, 66, 71) FMAC(42, 67, 68) FMAC(43, 64, 69) FMAC(44, 65, 70) FMAC(45, 66, 71) FMAC(46, 67, 68) FMAC(47, 64, 69))::: CLOBBERS);
        else if (KIND == 12)  // the same arithmetic as v_fma_f32 (VOP3) with the destination as the addend
            asm volatile(REP8(FMA3(40, 65, 70) FMA3(41, 66, 71) FMA3(42, 67, 68) FMA3(43, 64, 69) FMA3(44, 65, 70) FMA3(45, 66, 71) FMA3(46, 67, 68) FMA3(47, 64, 69))::: CLOBBERS);
        else if (KIND == 13) asm volatile(REP8(ALL8(OPK, "v_or_b32"))::: CLOBBERS);
        else if (KIND == 14) asm volatile(REP8(ALL8(OPK, "v_xor_b32"))::: CLOBBERS);
        else if (KIND == 15) asm volatile(REP8(ALL8(OPK, "v_lshrrev_b32"))::: CLOBBERS);
        else if (KIND == 16) asm volatile(REP8(ALL8(OPK, "v_mul_f32"))::: CLOBBERS);
        else if (KIND == 17) asm volatile(REP8(ALL8(OPK, "v_max_i32"))::: CLOBBERS);
        else if (KIND == 18) asm volatile(REP8(ALL8(OPK, "v_min_f32"))::: CLOBBERS);
        else if (KIND == 19) asm volatile(REP8(ALL8(OPK, "v_sub_u32"))::: CLOBBERS);
        else if (KIND == 20) asm volatile(REP8(ALL8(OPK, "v_mul_legacy_f32"))::: CLOBBERS);
        else if (KIND == 21) asm volatile(REP8(ALL8(OPK, "v_and_b32"))::: CLOBBERS);
        else if (KIND == 22) asm volatile(REP8(ALL8(OPK, "v_ashrrev_i32"))::: CLOBBERS);
        else if (KIND == 23) asm volatile(REP8(ALL8(OPK, "v_lshlrev_b32"))::: CLOBBERS);
        else if (KIND == 24)  // fast and slow class alternating in ONE wave: 4 fma + 4 cvt per group (sum of the costs, or the larger?)
            asm volatile(REP8(FMA(40, 65, 70) CVT(44, 65) FMA(41, 66, 71) CVT(45, 66) FMA(42, 67, 68) CVT(46, 67) FMA(43, 64, 69) CVT(47, 64))::: CLOBBERS);
        else if (KIND == 25)  // 6 fma + 2 cvt per group
            asm volatile(REP8(FMA(40, 65, 70) FMA(41, 66, 71) FMA(42, 67, 68) CVT(46, 67) FMA(43, 64, 69) FMA(44, 65, 70) FMA(45, 66, 71) CVT(47, 64))::: CLOBBERS);
        else if (KIND == 26)  // 2 fma + 6 cvt per group
            asm volatile(REP8(FMA(40, 65, 70) CVT(42, 67) CVT(43, 64) CVT(44, 65) FMA(41, 66, 71) CVT(45, 66) CVT(46, 67) CVT(47, 64))::: CLOBBERS);
        else if (KIND == 27)  // 4 fma then 4 max3 (two slow kinds are one class?) -> cvt + max3 alternating
            asm volatile(REP8(CVT(40, 65) MAX3(44, 65, 70) CVT(41, 66) MAX3(45, 66, 71) CVT(42, 67) MAX3(46, 67, 68) CVT(43, 64) MAX3(47, 64, 69))::: CLOBBERS);
    }
    asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v44\n v_add_f32 %0, %0, v45\n v_add_f32 %0, %0, v46\n v_add_f32 %0, %0, v47" : "=v"(r)::CLOBBERS);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, int wavesPerSimd, float* out)
{
    const int iters = 2000, insts = iters * 64;
    const int blocks = 256 * wavesPerSimd;  // 256-thread workgroups = 4 waves = one per SIMD of a CU; `wavesPerSimd` of them per CU
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double clock = 2.4e9;
    std::printf("%-62s waves/SIMD %d  %8.3f ms  -> %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms, ms * 1e-3 * clock / ((double)insts * wavesPerSimd));
}

int main()
{
    float* out;
    hipMalloc(&out, 1024);
    for (int w : {1, 2, 5, 8}) {
        run<0>("v_fma_f32   sources in three banks", w, out);
        run<1>("v_fma_f32   src0 and src1 in one bank", w, out);
        run<2>("v_fma_f32   all three sources in one bank", w, out);
        run<3>("v_fma_f32   src1 = src2 (same register)", w, out);
        run<4>("v_add_f32   two banks", w, out);
        run<5>("v_add_f32   one bank", w, out);
        run<6>("v_max3_i32  three banks", w, out);
        run<7>("v_max3_i32  one bank", w, out);
        run<8>("v_cvt_f32_ubyte1  source in another bank than the destination", w, out);
        run<9>("v_cvt_f32_ubyte1  source in the destination's bank", w, out);
        run<10>("v_fma_f32   three banks, only two independent chains", w, out);
        run<11>("v_fmac_f32 (VOP2)  d += a * b", w, out);
        run<12>("v_fma_f32  (VOP3)  d = a * b + d", w, out);
        run<13>("v_or_b32", w, out);
        run<14>("v_xor_b32", w, out);
        run<15>("v_lshrrev_b32", w, out);
        run<16>("v_mul_f32", w, out);
        run<17>("v_max_i32", w, out);
        run<18>("v_min_f32", w, out);
        run<19>("v_sub_u32", w, out);
        run<20>("v_mul_legacy_f32", w, out);
        run<21>("v_and_b32", w, out);
        run<22>("v_ashrrev_i32", w, out);
        run<23>("v_lshlrev_b32", w, out);
        run<24>("4 v_fma + 4 v_cvt_f32_ubyte alternating (per instruction)", w, out);
        run<25>("6 v_fma + 2 v_cvt_f32_ubyte", w, out);
        run<26>("2 v_fma + 6 v_cvt_f32_ubyte", w, out);
        run<27>("4 v_cvt_f32_ubyte + 4 v_max3_i32 alternating", w, out);
    }
    return 0;
}
