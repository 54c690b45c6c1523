// micro-benchmark (round 3): issue cost of the instruction types in the LDS-staged trilinear loop (vr_tslab.hip), each
// as ONE asm block of 128 independent instructions per loop iteration, fixed registers, 6 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define OPS8(op) op " v10, v2, v3\n" op " v11, v2, v3\n" op " v12, v2, v3\n" op " v13, v2, v3\n" op " v14, v2, v3\n" op " v15, v2, v3\n" op " v16, v2, v3\n" op " v17, v2, v3\n"
#define OPS8_1(op) op " v10, v2\n" op " v11, v3\n" op " v12, v2\n" op " v13, v3\n" op " v14, v2\n" op " v15, v3\n" op " v16, v2\n" op " v17, v3\n"
#define OPS8_3(op) op " v10, v2, v3, v4\n" op " v11, v2, v3, v4\n" op " v12, v2, v3, v4\n" op " v13, v2, v3, v4\n" op " v14, v2, v3, v4\n" op " v15, v2, v3, v4\n" op " v16, v2, v3, v4\n" op " v17, v2, v3, v4\n"
#define CLOB "v2", "v3", "v4", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "vcc", "scc", "s10", "s11"

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(R16(OPS8("v_add_f32")) ::: CLOB);
        if (OP == 1) asm volatile(R16(OPS8("v_max_f32")) ::: CLOB);
        if (OP == 2) asm volatile(R16(OPS8_1("v_fract_f32")) ::: CLOB);
        if (OP == 3) asm volatile(R16(OPS8_1("v_cvt_i32_f32")) ::: CLOB);
        if (OP == 4) asm volatile(R16(OPS8_1("v_cvt_f32_u32")) ::: CLOB);
        if (OP == 5) asm volatile(R16(OPS8_3("v_add_lshl_u32")) ::: CLOB);
        if (OP == 6) asm volatile(R16(OPS8_3("v_lshl_add_u32")) ::: CLOB);
        if (OP == 7) asm volatile(R16(OPS8("v_add_u32")) ::: CLOB);
        if (OP == 8) asm volatile(R16("v_cndmask_b32 v10, v2, v3, vcc\n v_cndmask_b32 v11, v2, v3, vcc\n v_cndmask_b32 v12, v2, v3, vcc\n v_cndmask_b32 v13, v2, v3, vcc\n"
                                      "v_cndmask_b32 v14, v2, v3, vcc\n v_cndmask_b32 v15, v2, v3, vcc\n v_cndmask_b32 v16, v2, v3, vcc\n v_cndmask_b32 v17, v2, v3, vcc\n") ::: CLOB);
        if (OP == 9) asm volatile(R16("v_cndmask_b32_e64 v10, v2, v3, s[10:11]\n v_cndmask_b32_e64 v11, v2, v3, s[10:11]\n v_cndmask_b32_e64 v12, v2, v3, s[10:11]\n v_cndmask_b32_e64 v13, v2, v3, s[10:11]\n"
                                      "v_cndmask_b32_e64 v14, v2, v3, s[10:11]\n v_cndmask_b32_e64 v15, v2, v3, s[10:11]\n v_cndmask_b32_e64 v16, v2, v3, s[10:11]\n v_cndmask_b32_e64 v17, v2, v3, s[10:11]\n") ::: CLOB);
        if (OP == 10) asm volatile(R16("v_cmp_lt_f32 vcc, v2, v3\n v_cmp_lt_f32 vcc, v3, v2\n v_cmp_lt_f32 vcc, v2, v3\n v_cmp_lt_f32 vcc, v3, v2\n v_cmp_lt_f32 vcc, v2, v3\n v_cmp_lt_f32 vcc, v3, v2\n v_cmp_lt_f32 vcc, v2, v3\n v_cmp_lt_f32 vcc, v3, v2\n") ::: CLOB);
        if (OP == 11) asm volatile(R16("v_cmp_eq_u32_e64 s[10:11], v2, v3\n v_cmp_eq_u32_e64 s[10:11], v3, v2\n v_cmp_eq_u32_e64 s[10:11], v2, v3\n v_cmp_eq_u32_e64 s[10:11], v3, v2\n"
                                       "v_cmp_eq_u32_e64 s[10:11], v2, v3\n v_cmp_eq_u32_e64 s[10:11], v3, v2\n v_cmp_eq_u32_e64 s[10:11], v2, v3\n v_cmp_eq_u32_e64 s[10:11], v3, v2\n") ::: CLOB);
        if (OP == 12) asm volatile(R16(OPS8_3("v_fma_f32")) ::: CLOB);
        if (OP == 13) asm volatile(R16(OPS8("v_ashrrev_i32")) ::: CLOB);
        if (OP == 14) asm volatile(R16("v_cvt_f32_u32_sdwa v10, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v11, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                       "v_cvt_f32_u32_sdwa v12, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v13, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                       "v_cvt_f32_u32_sdwa v14, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v15, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                       "v_cvt_f32_u32_sdwa v16, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v17, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n") ::: CLOB);
        if (OP == 15) asm volatile(R16(OPS8("v_ldexp_f32")) ::: CLOB);
        if (OP == 16) asm volatile(R16(OPS8_1("v_cvt_f32_ubyte0")) ::: CLOB);
        if (OP == 17) asm volatile(R16(OPS8_3("v_mad_u32_u24")) ::: CLOB);
        if (OP == 18) asm volatile(R16(OPS8_3("v_and_or_b32")) ::: CLOB);
        if (OP == 19) asm volatile(R16(OPS8_1("v_mov_b32")) ::: CLOB);
        if (OP == 20) asm volatile(R16(OPS8_3("v_perm_b32")) ::: CLOB);
        if (OP == 22) asm volatile(R16(OPS8("v_max_i32")) ::: CLOB);
        if (OP == 23) asm volatile(R16(OPS8("v_min_u32")) ::: CLOB);
        if (OP == 24) asm volatile(R16(OPS8("v_and_b32")) ::: CLOB);
        if (OP == 25) asm volatile(R16(OPS8("v_lshlrev_b32")) ::: CLOB);
        if (OP == 26) asm volatile(R16(OPS8_3("v_bfe_u32")) ::: CLOB);
        if (OP == 27) asm volatile(R16(OPS8("v_mul_u32_u24")) ::: CLOB);
        if (OP == 28) asm volatile(R16(OPS8_3("v_med3_f32")) ::: CLOB);
        if (OP == 29) asm volatile(R16(OPS8("v_mul_f32")) ::: CLOB);
        if (OP == 30) asm volatile(R16(OPS8_3("v_add3_u32")) ::: CLOB);
        if (OP == 31) asm volatile(R16(OPS8("v_sub_f32")) ::: CLOB);
        if (OP == 32) asm volatile(R16("v_add_f32 v10, 1.0, v2\n v_add_f32 v11, 0.5, v3\n v_add_f32 v12, -0.5, v2\n v_add_f32 v13, 2.0, v3\n v_add_f32 v14, 1.0, v2\n v_add_f32 v15, 0.5, v3\n v_add_f32 v16, -0.5, v2\n v_add_f32 v17, 2.0, v3\n") ::: CLOB);
        if (OP == 33) asm volatile(R16("v_add_f32 v10, s10, v2\n v_add_f32 v11, s11, v3\n v_add_f32 v12, s10, v2\n v_add_f32 v13, s11, v3\n v_add_f32 v14, s10, v2\n v_add_f32 v15, s11, v3\n v_add_f32 v16, s10, v2\n v_add_f32 v17, s11, v3\n") ::: CLOB);
        if (OP == 34) asm volatile(R16("v_fma_f32 v10, v2, s10, v3\n v_fma_f32 v11, v2, s11, v3\n v_fma_f32 v12, v2, s10, v3\n v_fma_f32 v13, v2, s11, v3\n v_fma_f32 v14, v2, s10, v3\n v_fma_f32 v15, v2, s11, v3\n v_fma_f32 v16, v2, s10, v3\n v_fma_f32 v17, v2, s11, v3\n") ::: CLOB);
        if (OP == 35) asm volatile(R16("v_add_f32_e64 v10, v2, v3 clamp\n v_add_f32_e64 v11, v2, v3 clamp\n v_add_f32_e64 v12, v2, v3 clamp\n v_add_f32_e64 v13, v2, v3 clamp\n v_add_f32_e64 v14, v2, v3 clamp\n v_add_f32_e64 v15, v2, v3 clamp\n v_add_f32_e64 v16, v2, v3 clamp\n v_add_f32_e64 v17, v2, v3 clamp\n") ::: CLOB);
        if (OP == 36) asm volatile(R16("v_add_f32_e64 v10, v2, v3\n v_add_f32_e64 v11, v2, v3\n v_add_f32_e64 v12, v2, v3\n v_add_f32_e64 v13, v2, v3\n v_add_f32_e64 v14, v2, v3\n v_add_f32_e64 v15, v2, v3\n v_add_f32_e64 v16, v2, v3\n v_add_f32_e64 v17, v2, v3\n") ::: CLOB);
        if (OP == 37) asm volatile(R16("v_add_f32_e64 v10, v2, -0.5 clamp\n v_add_f32_e64 v11, v3, -0.5 clamp\n v_add_f32_e64 v12, v2, -0.5 clamp\n v_add_f32_e64 v13, v3, -0.5 clamp\n v_add_f32_e64 v14, v2, -0.5 clamp\n v_add_f32_e64 v15, v3, -0.5 clamp\n v_add_f32_e64 v16, v2, -0.5 clamp\n v_add_f32_e64 v17, v3, -0.5 clamp\n") ::: CLOB);
        if (OP == 38) asm volatile(R16("v_mul_f32_e64 v10, v2, v3 clamp\n v_mul_f32_e64 v11, v2, v3 clamp\n v_mul_f32_e64 v12, v2, v3 clamp\n v_mul_f32_e64 v13, v2, v3 clamp\n v_mul_f32_e64 v14, v2, v3 clamp\n v_mul_f32_e64 v15, v2, v3 clamp\n v_mul_f32_e64 v16, v2, v3 clamp\n v_mul_f32_e64 v17, v2, v3 clamp\n") ::: CLOB);
        if (OP == 39) asm volatile(R16("v_fma_f32 v10, v2, v3, v4 clamp\n v_fma_f32 v11, v2, v3, v4 clamp\n v_fma_f32 v12, v2, v3, v4 clamp\n v_fma_f32 v13, v2, v3, v4 clamp\n v_fma_f32 v14, v2, v3, v4 clamp\n v_fma_f32 v15, v2, v3, v4 clamp\n v_fma_f32 v16, v2, v3, v4 clamp\n v_fma_f32 v17, v2, v3, v4 clamp\n") ::: CLOB);
        if (OP == 40) asm volatile(R16("v_add_f32_e64 v10, v2, v3 mul:2\n v_add_f32_e64 v11, v2, v3 mul:2\n v_add_f32_e64 v12, v2, v3 mul:2\n v_add_f32_e64 v13, v2, v3 mul:2\n v_add_f32_e64 v14, v2, v3 mul:2\n v_add_f32_e64 v15, v2, v3 mul:2\n v_add_f32_e64 v16, v2, v3 mul:2\n v_add_f32_e64 v17, v2, v3 mul:2\n") ::: CLOB);
        if (OP == 41) asm volatile(R16(OPS8_1("v_trunc_f32")) ::: CLOB);
        if (OP == 42) asm volatile(R16("v_add_u32 v10, s10, v2\n v_add_u32 v11, s11, v3\n v_add_u32 v12, s10, v2\n v_add_u32 v13, s11, v3\n v_add_u32 v14, s10, v2\n v_add_u32 v15, s11, v3\n v_add_u32 v16, s10, v2\n v_add_u32 v17, s11, v3\n") ::: CLOB);
        if (OP == 43) asm volatile(R16("v_fma_f32 v10, 2.0, v2, v3\n v_fma_f32 v11, 4.0, v2, v3\n v_fma_f32 v12, 2.0, v2, v3\n v_fma_f32 v13, 4.0, v2, v3\n v_fma_f32 v14, 2.0, v2, v3\n v_fma_f32 v15, 4.0, v2, v3\n v_fma_f32 v16, 2.0, v2, v3\n v_fma_f32 v17, 4.0, v2, v3\n") ::: CLOB);
        if (OP == 21) asm volatile(R16("s_and_b64 s[10:11], s[10:11], vcc\n s_or_b64 s[10:11], s[10:11], vcc\n s_and_b64 s[10:11], s[10:11], vcc\n s_or_b64 s[10:11], s[10:11], vcc\n"
                                       "s_and_b64 s[10:11], s[10:11], vcc\n s_or_b64 s[10:11], s[10:11], vcc\n s_and_b64 s[10:11], s[10:11], vcc\n s_or_b64 s[10:11], s[10:11], vcc\n") ::: CLOB);
    }
    float r;
    asm volatile("v_mov_b32 %0, v10" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name)
{
    const int wps = 6, blocks = 256 * wps, iters = 1000;
    float *out; (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s %8.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz, 6 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 128.0 * wps));
    (void)hipFree(out); fflush(stdout);
}

int main()
{
    run<0>("v_add_f32"); run<12>("v_fma_f32"); run<1>("v_max_f32"); run<2>("v_fract_f32"); run<3>("v_cvt_i32_f32"); run<4>("v_cvt_f32_u32"); run<14>("v_cvt_f32_u32_sdwa WORD_0");
    run<16>("v_cvt_f32_ubyte0"); run<15>("v_ldexp_f32"); run<5>("v_add_lshl_u32"); run<6>("v_lshl_add_u32"); run<7>("v_add_u32"); run<13>("v_ashrrev_i32"); run<17>("v_mad_u32_u24");
    run<18>("v_and_or_b32"); run<20>("v_perm_b32"); run<19>("v_mov_b32"); run<8>("v_cndmask_b32 (vcc)"); run<9>("v_cndmask_b32_e64 (sgpr pair)"); run<10>("v_cmp_lt_f32 -> vcc"); run<11>("v_cmp_eq_u32_e64 -> sgpr");
    run<21>("s_and_b64 / s_or_b64");
    run<22>("v_max_i32"); run<23>("v_min_u32"); run<24>("v_and_b32"); run<25>("v_lshlrev_b32"); run<26>("v_bfe_u32"); run<27>("v_mul_u32_u24"); run<28>("v_med3_f32"); run<29>("v_mul_f32"); run<30>("v_add3_u32"); run<31>("v_sub_f32");
    run<32>("v_add_f32 (inline constant src0)"); run<33>("v_add_f32 (sgpr src0)"); run<34>("v_fma_f32 (sgpr src1)");
    // round 4: the VOP3 output modifiers (a clamp to [0, 1] on coordinates scaled by a power of two would replace v_max_f32(x, 0))
    run<36>("v_add_f32_e64 (VOP3 encoding)"); run<35>("v_add_f32_e64 clamp"); run<37>("v_add_f32_e64 -0.5 clamp"); run<38>("v_mul_f32_e64 clamp"); run<39>("v_fma_f32 clamp"); run<40>("v_add_f32_e64 mul:2");
    run<41>("v_trunc_f32"); run<42>("v_add_u32 (sgpr src0)"); run<43>("v_fma_f32 (inline constant src0)");
    return 0;
}
