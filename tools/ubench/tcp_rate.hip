// micro-benchmark: what a 64-lane gather costs in the vector L1 (TCP) when every line HITS, as a function of
// how the lanes' addresses are spread over cache lines.  hipcc --offload-arch=gfx950 -O3 -o tcp_rate tcp_rate.hip
// Each wavefront issues `iters` x 8 independent buffer loads from a small L1-resident table (table_lines x 128 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES>
__device__ __forceinline__ uint32_t ld(__amdgpu_buffer_rsrc_t rs, uint32_t off)
{
    if (BYTES == 1) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off, 0, 0);
    if (BYTES == 2) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
    if (BYTES == 4) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    if (BYTES == 8) { typedef uint32_t u2 __attribute__((ext_vector_type(2))); const u2 q = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); return q.x ^ q.y; }
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return q.x ^ q.y ^ q.z ^ q.w;
}

// offsets[p][lane]: byte offset of lane's first load; consecutive loads add `stride` (mod table bytes)
template <int BYTES>
__global__ __launch_bounds__(256) void k(const uint8_t *table, uint32_t table_bytes, const uint32_t *offsets, uint32_t stride,
                                         int iters, uint32_t *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)table_bytes, 0x00020000);
    uint32_t off = offsets[threadIdx.x & 63];
    uint32_t acc = 0;
    const uint32_t mask = table_bytes - 1;
    for (int it = 0; it < iters; it++) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { v[u] = ld<BYTES>(rs, off); off = (off + stride) & mask; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

struct Pattern { const char *name; std::vector<uint32_t> off; };

int main()
{
    const uint32_t table_bytes = 8192;                       // 64 lines: L1-resident
    uint8_t *table; hipMalloc(&table, table_bytes); hipMemset(table, 1, table_bytes);
    uint32_t *offs, *out; hipMalloc(&offs, 64 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<Pattern> pats;
    auto add = [&](const char *n, auto f) { Pattern p{n, std::vector<uint32_t>(64)}; for (int l = 0; l < 64; l++) p.off[l] = f(l) & (table_bytes - 1); pats.push_back(p); };
    add("all lanes one address", [](int) { return 0u; });
    add("consecutive 2-byte (1 line)", [](int l) { return 2u * l; });
    add("consecutive 4-byte (2 lines)", [](int l) { return 4u * l; });
    add("2 lines, lanes 0-31 / 32-63", [](int l) { return (l / 32) * 128u + 2u * (l % 32); });
    add("4 lines, blocks of 16 lanes", [](int l) { return (l / 16) * 128u + 2u * (l % 16); });
    add("8 lines, blocks of 8 lanes", [](int l) { return (l / 8) * 128u + 2u * (l % 8); });
    add("16 lines, blocks of 4 lanes", [](int l) { return (l / 4) * 128u + 2u * (l % 4); });
    add("32 lines, blocks of 2 lanes", [](int l) { return (l / 2) * 128u + 2u * (l % 2); });
    add("64 lines, one per lane", [](int l) { return l * 128u; });
    add("4 lines, interleaved (lane%4)", [](int l) { return (l % 4) * 128u + 2u * (l / 4); });
    add("8 lines, interleaved (lane%8)", [](int l) { return (l % 8) * 128u + 2u * (l / 8); });
    add("16 lines, interleaved (lane%16)", [](int l) { return (l % 16) * 128u + 2u * (l / 16); });
    add("ray-march like: 8 rows x ~3.5 bricks", [](int l) { const int row = l / 8, x = (l % 8) * 13 / 10; return (row * 13 / 10 / 4) * 1024u + (row * 13 / 10 % 4) * 8u + (x / 4) * 128u + (x % 4) * 2u; });
    add("same, 12-bit packed (odd bytes)", [](int l) { const int row = l / 8, x = (l % 8) * 13 / 10; return (row * 13 / 10 / 4) * 768u + (row * 13 / 10 % 4) * 6u + (x / 4) * 96u + (x % 4) * 3u / 2u; });
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 400;                 // 32 wavefronts per CU
    printf("%-40s %10s %10s %10s %10s %10s   (cycles per wave-gather per CU at 2.4 GHz; 32 waves/CU)\n", "pattern", "1-byte", "2-byte", "4-byte", "8-byte", "16-byte");
    for (auto &p : pats) {
        hipMemcpy(offs, p.off.data(), 256, hipMemcpyHostToDevice);
        double cyc[5];
        for (int w = 0; w < 5; w++) {
            auto launch = [&](int n) {
                if (w == 0) k<1><<<blocks, 256>>>(table, table_bytes, offs, 0, n, out);
                if (w == 1) k<2><<<blocks, 256>>>(table, table_bytes, offs, 0, n, out);
                if (w == 2) k<4><<<blocks, 256>>>(table, table_bytes, offs, 0, n, out);
                if (w == 3) k<8><<<blocks, 256>>>(table, table_bytes, offs, 0, n, out);
                if (w == 4) k<16><<<blocks, 256>>>(table, table_bytes, offs, 0, n, out);
            };
            launch(50); launch(iters); hipDeviceSynchronize();
            hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double gathers_per_cu = (double)iters * 8.0 * 32.0;
            cyc[w] = ms * 1e-3 * 2.4e9 / gathers_per_cu;
        }
        printf("%-40s %10.2f %10.2f %10.2f %10.2f %10.2f\n", p.name, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4]);
    }
    return 0;
}
