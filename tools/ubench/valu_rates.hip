// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU/LDS
// instruction types the ray-march inner loop uses.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    __shared__ float2 lds[4096];
    if (OP == 9) { for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = make_float2(e, e); __syncthreads(); }
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (OP == 1) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&i0));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_i32_f32 %0, %8\n v_cvt_i32_f32 %1, %8\n v_cvt_i32_f32 %2, %8\n v_cvt_i32_f32 %3, %8\n v_cvt_i32_f32 %4, %8\n v_cvt_i32_f32 %5, %8\n v_cvt_i32_f32 %6, %8\n v_cvt_i32_f32 %7, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a0));) }
        if (OP == 3) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(iters));) }
        if (OP == 4) { REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));) }
        if (OP == 5) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));) }
        if (OP == 6) { REP8(asm volatile("v_med3_i32 %0, %0, %8, %1\n v_med3_i32 %1, %1, %8, %2\n v_med3_i32 %2, %2, %8, %3\n v_med3_i32 %3, %3, %8, %4\n v_med3_i32 %4, %4, %8, %5\n v_med3_i32 %5, %5, %8, %6\n v_med3_i32 %6, %6, %8, %7\n v_med3_i32 %7, %7, %8, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(iters));) }
        if (OP == 7) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&i0));) }
        if (OP == 8) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));) }
        if (OP == 9) {
            // 8 independent random-ish ds_read_b64 per step
            float2 r0 = lds[(i0 * 37) & 4095], r1 = lds[(i1 * 53) & 4095], r2 = lds[(i2 * 71) & 4095], r3 = lds[(i3 * 91) & 4095];
            float2 r4 = lds[(i4 * 13) & 4095], r5 = lds[(i5 * 29) & 4095], r6 = lds[(i6 * 41) & 4095], r7 = lds[(i7 * 67) & 4095];
            i0 += (int)r0.x; i1 += (int)r1.x; i2 += (int)r2.x; i3 += (int)r3.x; i4 += (int)r4.y; i5 += (int)r5.y; i6 += (int)r6.y; i7 += (int)r7.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP>
void run(const char *name, int per_iter)
{
    const int blocks = 256 * 8, iters = 2000;   // 8 blocks x 4 waves = 32 waves/CU, 8 per SIMD
    float *out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * per_iter * 8.0;   // 8 waves per SIMD
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}

int main()
{
    run<0>("v_add_f32", 64); run<5>("v_mul_f32", 64); run<1>("v_pk_add_f32", 64); run<7>("v_pk_mul/fma_f32", 64);
    run<2>("v_cvt_i32_f32", 64); run<3>("v_mad_u32_u24", 64); run<4>("v_lshlrev_b32", 64); run<6>("v_med3_i32", 64);
    run<8>("v_lshl_add_u32/v_add3_u32", 64); run<9>("ds_read_b64 x8 + 8 cvt/add", 8);
    return 0;
}
