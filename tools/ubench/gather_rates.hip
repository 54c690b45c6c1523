// micro-benchmark: vector-memory gather throughput of one CU's TA/TCP path for the access
// shapes of the ray-march (64 lanes = 8 rows x 8 neighbouring texels).
// hipcc --offload-arch=gfx950 -O3 gather_rates.hip -o gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// PATTERN: 0 = 64 consecutive elements (fully coalesced)
//          1 = 8 rows x 8 consecutive elements, row pitch 2048 B (separate cache lines)
//          2 = 8 rows x 8 elements with stride 1.25 (x = lane*5/4), pitch 2048 B   (like 1.27 voxels/pixel)
//          3 = same as 2 but 4x4x4 bricks of 128 B: rows of one brick-row are 8 B apart
//          4 = all lanes the same element
template <typename T, int PATTERN, int BYTES_WINDOW>
__global__ __launch_bounds__(256) void k(const T *__restrict__ p, uint32_t *out, int iters)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    unsigned lx = lane & 7, ly = lane >> 3, base;
    if (PATTERN == 0) base = lane;
    else if (PATTERN == 1) base = ly * (2048 / sizeof(T)) + lx;
    else if (PATTERN == 2) base = ly * (2048 / sizeof(T)) + (lx * 5) / 4;
    else if (PATTERN == 3) { unsigned x = (lx * 5) / 4, y = (ly * 5) / 4; base = (y >> 2) * (4096 / sizeof(T)) + (x >> 2) * 64 + (y & 3) * 4 + (x & 3); }
    else base = 0;
    const unsigned mask = BYTES_WINDOW / sizeof(T) - 1;
    unsigned a = base + wave * 977u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc += p[(a + u * 4099u) & mask];      // 8 independent gathers in flight
        }
        a += 16411u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename T, int PATTERN, int WIN>
void run(const char *name, const void *buf)
{
    const int blocks = 256 * 8, iters = 500;
    uint32_t *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, PATTERN, WIN><<<blocks, 256>>>((const T *)buf, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, PATTERN, WIN><<<blocks, 256>>>((const T *)buf, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waveloads_per_cu = (double)iters * 8 * 32;     // 32 waves per CU
    printf("%-58s %8.3f ms -> %6.1f cycles per wave-load per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / waveloads_per_cu);
    hipFree(out);
}

int main()
{
    void *buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20);
#define RUNS(T, TN) \
    run<T, 0, 16384>(TN " coalesced, 16 KiB window (L1)", buf); \
    run<T, 4, 16384>(TN " broadcast, 16 KiB window (L1)", buf); \
    run<T, 1, 16384>(TN " 8x8 dense rows, 16 KiB window (L1)", buf); \
    run<T, 2, 16384>(TN " 8x8 stride 1.25, 16 KiB window (L1)", buf); \
    run<T, 3, 16384>(TN " 8x8 stride 1.25 bricked, 16 KiB window (L1)", buf); \
    run<T, 2, 1 << 20>(TN " 8x8 stride 1.25, 1 MiB window (L2)", buf); \
    run<T, 3, 1 << 20>(TN " 8x8 stride 1.25 bricked, 1 MiB window (L2)", buf); \
    run<T, 2, 64 << 20>(TN " 8x8 stride 1.25, 64 MiB window (MALL)", buf); \
    run<T, 3, 64 << 20>(TN " 8x8 stride 1.25 bricked, 64 MiB window (MALL)", buf);
    RUNS(uint8_t, "u8 ") RUNS(uint16_t, "u16") RUNS(uint32_t, "u32")
    return 0;
}
