// micro-benchmark: throughput of MISSING gathers as a function of wavefronts per CU.
// Every wavefront runs batches of 8 independent 2-byte gathers; each gather touches LINES distinct random 128-byte lines of a
// 2 GiB buffer (lanes grouped 64/LINES per line), then waits for the batch (like the ray-march loop).
//   hipcc --offload-arch=gfx950 -O3 -o miss_rate miss_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LINES>
__global__ __launch_bounds__(64) void k(const uint8_t *buf, uint32_t lines_mask, int iters, uint32_t *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (int)0x7fffffff, 0x00020000);
    const uint32_t lane = threadIdx.x, grp = lane / (64 / LINES), sub = lane % (64 / LINES);
    uint32_t h = (blockIdx.x * 0x9E3779B9u) ^ (grp * 0x85ebca6bu) ^ 0x1234567u, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            h = h * 1664525u + 1013904223u;                         // same sequence in all lanes of a group
            const uint32_t line = (h >> 7) & lines_mask;
            v[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(line * 128u + sub * 2u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const size_t bytes = 2147483648ull - 4096;
    uint8_t *buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    uint32_t *out; hipMalloc(&out, 256 * 64 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t lines_mask = (1u << 23) - 1;                       // 8 M lines = 1 GiB window
    printf("%-6s %-10s %12s %14s %16s %14s\n", "lines", "waves/CU", "ms", "cyc/gather/CU", "line fills/us/CU", "GB/s chip");
    auto run = [&](int lines, int waves_per_cu) {
        const int blocks = 256 * waves_per_cu, iters = 600 / (waves_per_cu > 8 ? 2 : 1);
        auto launch = [&](int n) {
            if (lines == 4) k<4><<<blocks, 64>>>(buf, lines_mask, n, out);
            if (lines == 8) k<8><<<blocks, 64>>>(buf, lines_mask, n, out);
            if (lines == 16) k<16><<<blocks, 64>>>(buf, lines_mask, n, out);
            if (lines == 32) k<32><<<blocks, 64>>>(buf, lines_mask, n, out);
        };
        launch(20); hipDeviceSynchronize();
        hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double gathers_per_cu = (double)iters * 8.0 * waves_per_cu, fills = gathers_per_cu * lines;
        printf("%-6d %-10d %12.3f %14.1f %16.1f %14.0f\n", lines, waves_per_cu, ms, ms * 1e-3 * 2.4e9 / gathers_per_cu,
               fills / (ms * 1e3), fills * 256.0 * 128.0 / (ms * 1e-3) / 1e9);
    };
    for (int lines : {4, 16, 32})
        for (int w : {1, 2, 4, 8, 16, 24, 32}) run(lines, w);
    return 0;
}
