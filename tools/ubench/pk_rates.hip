// micro-benchmark (round 3): are packed fp32 VALU ops really slower than scalar ones on gfx950?
// valu_rates.hip measured v_pk_* at ~2.5x a scalar op, but its asm blocks passed 64-bit operands through
// "+v"(*(double*)&a) constraints, which may make the compiler shuffle register pairs between the blocks.
// Here every variant is ONE asm block inside the loop with fixed, explicitly numbered registers.
//   hipcc --offload-arch=gfx950 -O3 pk_rates.hip -o pk_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) {        // 16 x 8 independent scalar v_fma_f32
            asm volatile(R16("v_fma_f32 v10, v10, v2, v3\n v_fma_f32 v11, v11, v2, v3\n v_fma_f32 v12, v12, v2, v3\n v_fma_f32 v13, v13, v2, v3\n"
                             "v_fma_f32 v14, v14, v2, v3\n v_fma_f32 v15, v15, v2, v3\n v_fma_f32 v16, v16, v2, v3\n v_fma_f32 v17, v17, v2, v3\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
        } else if (OP == 1) { // 16 x 8 v_pk_fma_f32 on 8 independent register pairs
            asm volatile(R16("v_pk_fma_f32 v[10:11], v[10:11], v[2:3], v[4:5]\n v_pk_fma_f32 v[12:13], v[12:13], v[2:3], v[4:5]\n"
                             "v_pk_fma_f32 v[14:15], v[14:15], v[2:3], v[4:5]\n v_pk_fma_f32 v[16:17], v[16:17], v[2:3], v[4:5]\n"
                             "v_pk_fma_f32 v[18:19], v[18:19], v[2:3], v[4:5]\n v_pk_fma_f32 v[20:21], v[20:21], v[2:3], v[4:5]\n"
                             "v_pk_fma_f32 v[22:23], v[22:23], v[2:3], v[4:5]\n v_pk_fma_f32 v[24:25], v[24:25], v[2:3], v[4:5]\n")
                         ::: "v2", "v3", "v4", "v5", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25");
        } else if (OP == 2) { // v_pk_add_f32
            asm volatile(R16("v_pk_add_f32 v[10:11], v[10:11], v[2:3]\n v_pk_add_f32 v[12:13], v[12:13], v[2:3]\n"
                             "v_pk_add_f32 v[14:15], v[14:15], v[2:3]\n v_pk_add_f32 v[16:17], v[16:17], v[2:3]\n"
                             "v_pk_add_f32 v[18:19], v[18:19], v[2:3]\n v_pk_add_f32 v[20:21], v[20:21], v[2:3]\n"
                             "v_pk_add_f32 v[22:23], v[22:23], v[2:3]\n v_pk_add_f32 v[24:25], v[24:25], v[2:3]\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25");
        } else if (OP == 3) { // v_pk_mul_f32
            asm volatile(R16("v_pk_mul_f32 v[10:11], v[10:11], v[2:3]\n v_pk_mul_f32 v[12:13], v[12:13], v[2:3]\n"
                             "v_pk_mul_f32 v[14:15], v[14:15], v[2:3]\n v_pk_mul_f32 v[16:17], v[16:17], v[2:3]\n"
                             "v_pk_mul_f32 v[18:19], v[18:19], v[2:3]\n v_pk_mul_f32 v[20:21], v[20:21], v[2:3]\n"
                             "v_pk_mul_f32 v[22:23], v[22:23], v[2:3]\n v_pk_mul_f32 v[24:25], v[24:25], v[2:3]\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25");
        } else if (OP == 4) { // scalar v_add_f32 / v_mul_f32 / v_sub_f32 mix (the lerp's three ops)
            asm volatile(R16("v_sub_f32 v10, v11, v2\n v_mul_f32 v12, v10, v3\n v_add_f32 v13, v12, v2\n v_sub_f32 v14, v15, v2\n"
                             "v_mul_f32 v16, v14, v3\n v_add_f32 v17, v16, v2\n v_sub_f32 v18, v19, v2\n v_mul_f32 v20, v18, v3\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
        } else if (OP == 5) { // v_cvt_f32_u32 with SDWA word select
            asm volatile(R16("v_cvt_f32_u32_sdwa v10, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v11, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa v12, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v13, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa v14, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v15, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa v16, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa v17, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
        } else if (OP == 6) { // plain v_cvt_f32_u32
            asm volatile(R16("v_cvt_f32_u32 v10, v2\n v_cvt_f32_u32 v11, v3\n v_cvt_f32_u32 v12, v2\n v_cvt_f32_u32 v13, v3\n"
                             "v_cvt_f32_u32 v14, v2\n v_cvt_f32_u32 v15, v3\n v_cvt_f32_u32 v16, v2\n v_cvt_f32_u32 v17, v3\n")
                         ::: "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
        } else if (OP == 7) { // v_pk_fma_f32 with op_sel (broadcast of the low half of src1): the lerp with one weight for both halves
            asm volatile(R16("v_pk_fma_f32 v[10:11], v[10:11], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[12:13], v[12:13], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 v[14:15], v[14:15], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[16:17], v[16:17], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 v[18:19], v[18:19], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[20:21], v[20:21], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 v[22:23], v[22:23], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[24:25], v[24:25], v[2:3], v[4:5] op_sel_hi:[1,0,1]\n")
                         ::: "v2", "v3", "v4", "v5", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25");
        }
    }
    float r;
    asm volatile("v_mov_b32 %0, v10" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + acc + seed;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 1000;     // 256-thread blocks = 4 waves = one per SIMD
    float *out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * 128.0 * waves_per_simd;
    printf("%-44s %d waves/SIMD %8.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 6}) {
        if (w == 1) { run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("v_pk_add_f32", 1); run<3>("v_pk_mul_f32", 1); run<4>("v_sub/v_mul/v_add_f32", 1); run<5>("v_cvt_f32_u32_sdwa WORD_x", 1); run<6>("v_cvt_f32_u32", 1); run<7>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 1); }
        if (w == 2) { run<0>("v_fma_f32", 2); run<1>("v_pk_fma_f32", 2); run<2>("v_pk_add_f32", 2); run<3>("v_pk_mul_f32", 2); run<4>("v_sub/v_mul/v_add_f32", 2); run<5>("v_cvt_f32_u32_sdwa WORD_x", 2); run<6>("v_cvt_f32_u32", 2); run<7>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 2); }
        if (w == 6) { run<0>("v_fma_f32", 6); run<1>("v_pk_fma_f32", 6); run<2>("v_pk_add_f32", 6); run<3>("v_pk_mul_f32", 6); run<4>("v_sub/v_mul/v_add_f32", 6); run<5>("v_cvt_f32_u32_sdwa WORD_x", 6); run<6>("v_cvt_f32_u32", 6); run<7>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 6); }
    }
    return 0;
}
