// micro-benchmark + hardware check for the LDS-staged gather (slab kernel):
//   1. LDS-DMA (global_load_lds_dwordx4, per-lane source address, wave-uniform LDS destination)
//      followed by 2-byte LDS reads at ODD byte addresses (the 12-bit packed copy needs them):
//      every value is checked against the bytes in global memory;
//   2. rate of the gather shapes: ds_read_u16 at even / odd addresses, 8x8 lanes reading a
//      1.27-voxel-per-pixel footprint of 4x4x4 bricks (96 B and 128 B slots);
//   3. chip-wide LDS-DMA streaming rate with the slab kernel's issue pattern (one 1-KiB piece
//      per wave per phase, a few pieces in flight).
// hipcc --offload-arch=gfx950 -O3 lds_slab.hip -o lds_slab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(LDS_AS const char *)p; }

// one 1-KiB piece: lane l's 16 bytes from gsrc land at lds_dst + 16*l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);      // the "s" operand must be provably wave-uniform
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

constexpr int SLAB = 32768;

// ---- 1. correctness
__global__ __launch_bounds__(512) void dma_check(const uint8_t *__restrict__ g, uint32_t nchunks, uint32_t *bad, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[SLAB];
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr(slab));
    auto src_of = [&](uint32_t chunk) { return (uint64_t)(mix(chunk * 2654435761u + seed + blockIdx.x * 977u) % nchunks) * 16u; };
    for (int n = 0; n < 4; n++) {
        const uint32_t piece = w * 4u + n, chunk = piece * 64u + lane;
        if ((mix(chunk + seed) & 7u) != 0u)                    // some lanes masked off: their 16 bytes keep the old content
            glds16(g + src_of(chunk), base + piece * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t nbad = 0;
    for (int t = 0; t < 64; t++) {
        const uint32_t o = mix(threadIdx.x * 64u + t + seed) % (SLAB - 1);      // any byte offset, odd ones included
        uint16_t v;
        __builtin_memcpy(&v, &slab[o], 2);
        uint32_t want = 0;
        bool known = true;
        for (int k = 0; k < 2; k++) {
            const uint32_t chunk = (o + k) / 16u;
            if ((mix(chunk + seed) & 7u) == 0u) known = false;  // masked lane: content undefined
            else want |= (uint32_t)g[src_of(chunk) + (o + k) % 16u] << (8 * k);
        }
        if (known && want != v) nbad++;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// ---- 2. gather rates.  MODE 0: u16 reads at even addresses, 128-B slots; 1: u16 reads at
// floor(1.5*e) (odd half the time), 96-B slots; 2: the same through two u8 reads
template <int MODE>
__global__ __launch_bounds__(512) void gather_rate(uint32_t *out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t slab[SLAB];
    __shared__ uint16_t tabx[64], taby[64], tabz[64];
    for (int i = threadIdx.x; i < SLAB; i += 512) slab[i] = (uint8_t)mix(i);
    constexpr int RX = 12, RY = 7, SLOT = MODE == 0 ? 128 : 96;
    if (threadIdx.x < 64) {
        const unsigned i = threadIdx.x;
        if (MODE == 0) {
            tabx[i] = ((i >> 2) % RX) * SLOT + (i & 3) * 2; taby[i] = ((i >> 2) % RY) * RX * SLOT + (i & 3) * 8; tabz[i] = ((i >> 2) % 3) * RX * RY * SLOT + (i & 3) * 32;
        } else {
            tabx[i] = ((i >> 2) % RX) * SLOT + ((i & 3) * 3) / 2; taby[i] = ((i >> 2) % RY) * RX * SLOT + (i & 3) * 6; tabz[i] = ((i >> 2) % 3) * RX * RY * SLOT + (i & 3) * 24;
        }
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    // 8x8 pixels per wave, 4x2 waves per workgroup, 1.27 voxels per pixel
    float fx = (float)((w & 3u) * 8u + (lane & 7u)) * 1.27f + 0.3f, fy = (float)((w >> 2) * 8u + (lane >> 3)) * 1.27f + 0.7f, fz = 0.1f;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int vi = (int)fx, vj = (int)fy, vk = (int)fz;
            const uint32_t off = (uint32_t)tabx[vi & 63] + taby[vj & 63] + tabz[vk & 63];
            if (MODE == 2) {
                v[u] = (uint32_t)slab[off] | ((uint32_t)slab[off + 1] << 8);
            } else {
                uint16_t t;
                __builtin_memcpy(&t, &slab[off], 2);
                v[u] = t;
            }
            fz += 0.97f;
            if (fz >= 12.0f) fz -= 12.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
        fx += 0.013f; fy += 0.007f;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

// ---- 3. LDS-DMA streaming: every wave issues one 1-KiB piece per phase into a ring of
// RING pieces per wave, waits until at most DEPTH pieces are in flight, barrier per phase
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_stream(const uint8_t *__restrict__ g, uint64_t bytes, uint32_t *out, int phases)
{
    constexpr int RING = 4;
    __shared__ __attribute__((aligned(16))) uint8_t slab[8 * RING * 1024];
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr(slab)) + w * (RING * 1024u);
    // each workgroup streams its own contiguous region (bricks of one tile column are contiguous in z only
    // piecewise; this is the friendly case)
    const uint64_t per_wg = (bytes / gridDim.x) & ~(uint64_t)8191;
    const uint8_t *src = g + (uint64_t)blockIdx.x * per_wg + w * 1024u + lane * 16u;
    uint32_t acc = 0;
    const int max_ph = (int)(per_wg / 8192u);
    if (phases > max_ph) phases = max_ph;
    for (int p = 0; p < phases; p++) {
        glds16(src + (uint64_t)p * 8192u, base + (uint32_t)(p % RING) * 1024u);
        if (DEPTH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __syncthreads();
        if (p >= DEPTH) acc += slab[w * (RING * 1024u) + (uint32_t)((p - DEPTH) % RING) * 1024u + lane * 16u];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <typename F>
static float timed(F &&launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const uint64_t bytes = 2ull << 30;
    uint8_t *g; hipMalloc(&g, bytes);
    {   // pseudo-random bytes
        std::vector<uint8_t> h(64 << 20);
        uint32_t s = 12345; for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (uint64_t o = 0; o < bytes; o += h.size()) hipMemcpy(g + o, h.data(), h.size(), hipMemcpyHostToDevice);
    }
    uint32_t *bad, *out; hipMalloc(&bad, 4); hipMalloc(&out, 4096 * 512 * 4);
    hipMemset(bad, 0, 4);
    for (uint32_t seed = 1; seed <= 8; seed++) dma_check<<<1024, 512>>>(g, (uint32_t)((64u << 20) / 16u), bad, seed);
    uint32_t hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA + unaligned 2-byte LDS reads: %u mismatches over %d checked reads  (%s)\n", hb, 8 * 1024 * 512 * 64, hb ? "FAIL" : "ok");

    const int iters = 2000, blocks = 512;          // 2 workgroups of 8 waves per CU
    const double wave_gathers_per_cu = (double)iters * 8 * 16;
    float ms = timed([&] { gather_rate<0><<<blocks, 512>>>(out, iters); });
    printf("gather u16 even addresses, 128-B slots : %7.3f ms -> %5.1f cycles per wave-gather per CU (3 table reads + 1 slab read each)\n", ms, ms * 1e-3 * 2.4e9 / wave_gathers_per_cu);
    ms = timed([&] { gather_rate<1><<<blocks, 512>>>(out, iters); });
    printf("gather u16 floor(1.5 e),   96-B slots  : %7.3f ms -> %5.1f cycles per wave-gather per CU\n", ms, ms * 1e-3 * 2.4e9 / wave_gathers_per_cu);
    ms = timed([&] { gather_rate<2><<<blocks, 512>>>(out, iters); });
    printf("gather 2 x u8 floor(1.5 e), 96-B slots : %7.3f ms -> %5.1f cycles per wave-gather per CU\n", ms, ms * 1e-3 * 2.4e9 / wave_gathers_per_cu);

    const int phases = 256;
    auto rate = [&](float ms_, int nblocks) { return (double)nblocks * phases * 8192.0 / (ms_ * 1e-3) / 1e12; };
    for (int nb : {256, 512, 1024}) {
        ms = timed([&] { dma_stream<0><<<nb, 512>>>(g, bytes, out, phases); }); printf("LDS-DMA stream %4d WGs depth 0: %7.3f ms  %5.2f TB/s\n", nb, ms, rate(ms, nb));
        ms = timed([&] { dma_stream<1><<<nb, 512>>>(g, bytes, out, phases); }); printf("LDS-DMA stream %4d WGs depth 1: %7.3f ms  %5.2f TB/s\n", nb, ms, rate(ms, nb));
        ms = timed([&] { dma_stream<2><<<nb, 512>>>(g, bytes, out, phases); }); printf("LDS-DMA stream %4d WGs depth 2: %7.3f ms  %5.2f TB/s\n", nb, ms, rate(ms, nb));
        ms = timed([&] { dma_stream<3><<<nb, 512>>>(g, bytes, out, phases); }); printf("LDS-DMA stream %4d WGs depth 3: %7.3f ms  %5.2f TB/s\n", nb, ms, rate(ms, nb));
    }
    return hb ? 1 : 0;
}
