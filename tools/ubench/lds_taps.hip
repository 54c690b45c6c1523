// micro-benchmark (round 3): cost of the 8 trilinear taps when the bricks sit in LDS.
// A wavefront's 64 lanes are an 8x8 sheet of sample points `pitch` voxels apart (1.27 .. 1.78 on cfg3) at one
// depth; every lane reads its 2x2x2 cell from a torus of 4x4x4-voxel u16 bricks (x fastest inside a brick).
//   SLOT  bytes per brick slot (128 = packed, 136/144/160 = padded against bank aliasing between bricks)
//   MODE  0: eight ds_read_u16      1: four ds_read_b32 at the x0 tap (2-byte aligned when x0 is odd: replayed?)
//         2: eight ds_read_u16 with _d16 / _d16_hi pairs (4 result registers)
//   hipcc --offload-arch=gfx950 -O3 lds_taps.hip -o lds_taps
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RA = 6, RB = 6, RZ = 3;          // torus of brick slots (108 slots)

template <int SLOT, int MODE>
__global__ __launch_bounds__(512) void k(unsigned *out, int iters, float pitch, float zstep)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[RA * RB * RZ * SLOT + 64];
    for (int e = threadIdx.x; e < (RA * RB * RZ * SLOT) / 4; e += 512) reinterpret_cast<unsigned *>(ring)[e] = e * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float fx0 = 1.3f + pitch * (float)(lane & 7) + 0.37f * (float)(wave & 3), fy0 = 0.9f + pitch * (float)(lane >> 3) + 0.21f * (float)(wave >> 2);
    const float fz = 0.4f + 0.01f * (float)lane;
    unsigned acc = 0;
    // the eight tap offsets are computed once (no address arithmetic in the timed loop); every iteration shifts them
    // by a wave-uniform number of whole brick layers, so the bank pattern between the lanes stays that of the sheet
    unsigned base[8];
    {
        const int i0 = (int)fx0, j0 = (int)fy0, k0 = (int)fz;
        auto off = [&](int i, int j, int kk) -> unsigned {
            const unsigned slot = (unsigned)((i >> 2) % RA) + (unsigned)RA * ((unsigned)((j >> 2) % RB) + (unsigned)RB * (unsigned)((kk >> 2) % (RZ - 1)));
            return slot * (unsigned)SLOT + 2u * (unsigned)((i & 3) + 4 * (j & 3) + 16 * (kk & 3));
        };
        base[0] = off(i0, j0, k0); base[1] = off(i0 + 1, j0, k0); base[2] = off(i0, j0 + 1, k0); base[3] = off(i0 + 1, j0 + 1, k0);
        base[4] = off(i0, j0, k0 + 1); base[5] = off(i0 + 1, j0, k0 + 1); base[6] = off(i0, j0 + 1, k0 + 1); base[7] = off(i0 + 1, j0 + 1, k0 + 1);
    }
    for (int it = 0; it < iters; it++) {
        const unsigned shift = (unsigned)(it & 1) * (unsigned)(RA * RB * SLOT) + (unsigned)((it >> 1) & 3) * 32u;   // next layer of slots / next z row inside the brick
        unsigned o[8];
#pragma unroll
        for (int t = 0; t < 8; t++) o[t] = base[t] + shift;
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; t++) acc += *reinterpret_cast<const unsigned short *>(ring + o[t]);
        } else if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(o[t]) : "memory");
                acc += v;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                unsigned v = 0;
                asm volatile("ds_read_u16_d16 %0, %1\n ds_read_u16_d16_hi %0, %2\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(o[t]), "v"(o[t + 1]) : "memory");
                acc += v;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int SLOT, int MODE>
void run(const char *name, float pitch, int wgs_per_cu)
{
    const int blocks = 256 * wgs_per_cu, iters = 4000;
    unsigned *out; (void)hipMalloc(&out, (size_t)blocks * 512 * sizeof(unsigned));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<SLOT, MODE><<<blocks, 512>>>(out, 10, pitch, 0.98f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<SLOT, MODE><<<blocks, 512>>>(out, iters, pitch, 0.98f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    // per CU: wgs_per_cu x 8 wavefronts x iters samples; cycles per wave-sample (8 taps) at 2.4 GHz
    printf("%-34s slot %3d pitch %.2f %d wg/CU: %7.3f ms -> %6.1f CU-cycles per wave-sample (8 taps)\n", name, SLOT, pitch, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8.0 * wgs_per_cu));
    (void)hipFree(out);
}

int main()
{
    for (float pitch : {1.27f, 1.5f, 1.78f}) {
        run<128, 0>("8 x ds_read_u16", pitch, 2); run<136, 0>("8 x ds_read_u16", pitch, 2); run<144, 0>("8 x ds_read_u16", pitch, 2); run<160, 0>("8 x ds_read_u16", pitch, 2);
        run<128, 1>("4 x ds_read_b32 (x pair)", pitch, 2); run<144, 1>("4 x ds_read_b32 (x pair)", pitch, 2);
        run<128, 2>("4 x (ds_read_u16_d16 + _d16_hi)", pitch, 2); run<144, 2>("4 x (ds_read_u16_d16 + _d16_hi)", pitch, 2);
    }
    return 0;
}
