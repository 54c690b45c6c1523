// micro-benchmark: achievable HBM read bandwidth on this box for (a) a streaming read of a
// 2 GiB buffer with 16-byte loads and (b) the same bytes fetched as 128-byte lines in a
// scattered (hashed) order, one line per 8-lane group -- the shape of a gather's miss stream.
// hipcc --offload-arch=gfx950 -O3 stream_read.hip -o stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void stream_k(const uint4 *__restrict__ p, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// every group of 8 lanes reads one 128-byte line; lines visited in a hashed order
__global__ __launch_bounds__(256) void scatter_k(const uint4 *__restrict__ p, size_t nlines, uint32_t *out, uint32_t mul)
{
    uint32_t acc = 0;
    const size_t groups = (size_t)gridDim.x * blockDim.x / 8;
    const size_t g0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
    const unsigned sub = threadIdx.x & 7u;
    for (size_t g = g0; g < nlines; g += groups) {
        const size_t line = (g * (size_t)mul) & (nlines - 1);   // nlines is a power of two, mul odd: a permutation
        const uint4 v = p[line * 8 + sub];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t bytes = 2ull << 30;
    void *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMemset(buf, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a);
                if (mode == 0) stream_k<<<blocks, 256>>>((const uint4 *)buf, bytes / 16, out);
                else scatter_k<<<blocks, 256>>>((const uint4 *)buf, bytes / 128, out, mode == 1 ? 0x9E3779B1u : 40503u);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%-28s blocks %5d: %.3f ms -> %.2f TB/s\n", mode == 0 ? "streaming 16 B/lane" : (mode == 1 ? "scattered 128 B lines (far)" : "scattered 128 B lines (near)"),
                   blocks, best, bytes / (best * 1e-3) / 1e12);
        }
    }
    // device-to-device copy (read + write)
    void *dst; hipMalloc(&dst, bytes);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a); hipMemcpyAsync(dst, buf, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("hipMemcpy D2D 2 GiB: %.3f ms -> %.2f TB/s read + %.2f TB/s write\n", best, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    return 0;
}
