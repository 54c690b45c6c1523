// vr_helpers.hip -- helper kernels (volume generators, re-brick, statistics, skip grid, tile work estimates, apron / packed copies, assembly) and their host launchers; VR_TU_MAIN units only
// Part of vr_kernels.hip (textually included there, inside namespace vr, once per translation unit VR_TU): not a
// translation unit of its own.  Split out of vr_kernels.hip in round 6; the arithmetic contract is stated in that file's header.
#ifndef VR_TU_MAIN
#error "include through vr_kernels.hip"
#endif

// ------------------------------------------------------------------ helper kernels
__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t isqrt_u64(uint64_t v)
{
    uint64_t r = (uint64_t)sqrt((double)v);
    while (r * r > v) r--;
    while ((r + 1) * (r + 1) <= v) r++;
    return (uint32_t)r;
}

// linear voxel index -> storage index (identity for VR_LAYOUT_LINEAR)
__device__ __forceinline__ uint64_t storage_index(int layout, uint32_t i, uint32_t j, uint32_t k, uint32_t nx,
                                                  uint32_t ny, uint32_t bnx, uint32_t bny)
{
    if (layout == 0) return (uint64_t)i + (uint64_t)nx * ((uint64_t)j + (uint64_t)ny * (uint64_t)k);
    const uint64_t brick = (uint64_t)(i >> BRICK_LX) + (uint64_t)bnx * ((uint64_t)(j >> BRICK_LY) + (uint64_t)bny * (uint64_t)(k >> BRICK_LZ));
    return brick * 64u + ((i & (BRICK_X - 1u)) | ((j & (BRICK_Y - 1u)) << BRICK_LX) | ((k & (BRICK_Z - 1u)) << (BRICK_LX + BRICK_LY)));
}

template <typename VoxelT>
__global__ __launch_bounds__(256) void gen_volume_kernel(VoxelT *__restrict__ out, int kind, uint32_t nx,
                                                         uint32_t ny, uint32_t nz, uint32_t param, int layout,
                                                         uint32_t bnx, uint32_t bny)
{
    const uint64_t total = (uint64_t)nx * ny * nz;
    for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
         lin += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(lin % nx), j = (uint32_t)((lin / nx) % ny), k = (uint32_t)(lin / ((uint64_t)nx * ny));
        const int64_t a = 2 * (int64_t)i + 1 - (int64_t)nx, b = 2 * (int64_t)j + 1 - (int64_t)ny,
                      c = 2 * (int64_t)k + 1 - (int64_t)nz;
        const int64_t r2 = a * a + b * b + c * c;
        int64_t v;
        if (kind == 0) {   // VR_SYNTH_SPHERE_U8: param = radius
            const int64_t twoR = 2 * (int64_t)param;
            v = r2 < twoR * twoR ? 255 - (255 * (int64_t)isqrt_u64((uint64_t)r2)) / twoR : 0;
        } else {           // VR_SYNTH_NOISE_BALL: param = seed
            int64_t N = nx > ny ? nx : ny; N = N > (int64_t)nz ? N : (int64_t)nz;
            const int64_t vmax = sizeof(VoxelT) == 2 ? 4095 : 255;
            const int shift = sizeof(VoxelT) == 2 ? 2 : 6;
            const int64_t n2 = N * N;
            const int64_t base = r2 < n2 ? (vmax * (n2 - r2)) / n2 : 0;
            const uint32_t h = fmix32(((uint32_t)lin ^ (uint32_t)(lin >> 32) * 0x9E3779B1u) ^ param) & 0xFFu;
            v = base + (int64_t)(h >> shift);
            if (v > vmax) v = vmax;
            if (kind == 2) v += 1000;   // VR_SYNTH_NOISE_BALL_CT (16-bit): the same ball stored as CT data is, offset by +1000
        }
        out[storage_index(layout, i, j, k, nx, ny, bnx, bny)] = (VoxelT)v;
    }
}

// re-layout between linear and bricked storage (dir 0: linear -> bricked, 1: back)
template <typename VoxelT>
__global__ __launch_bounds__(256) void relayout_kernel(const VoxelT *__restrict__ in, VoxelT *__restrict__ out,
                                                       uint32_t nx, uint32_t ny, uint32_t nz, uint32_t bnx,
                                                       uint32_t bny, int to_linear)
{
    const uint64_t total = (uint64_t)nx * ny * nz;
    for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
         lin += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(lin % nx), j = (uint32_t)((lin / nx) % ny), k = (uint32_t)(lin / ((uint64_t)nx * ny));
        const uint64_t b = storage_index(1, i, j, k, nx, ny, bnx, bny);
        if (to_linear) out[lin] = in[b];
        else out[b] = in[lin];
    }
}

// dataset min/max scan (src/RendererCore.cpp:362-379, including the skipped index
// 8390640) and the raw 256-bin counts of :386-399
template <typename VoxelT>
__global__ __launch_bounds__(256) void stats_kernel(const VoxelT *__restrict__ vol, uint32_t nx, uint32_t ny,
                                                    uint32_t nz, int layout, uint32_t bnx, uint32_t bny,
                                                    int pass, float scale255, unsigned *minmax, unsigned *hist)
{
    __shared__ unsigned lh[256];
    if (threadIdx.x < 256) lh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t total = (uint64_t)nx * ny * nz;
    unsigned mn = 0xffffffffu, mx = 0, xmn = 0xffffffffu, xmx = 0;
    for (uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lin < total;
         lin += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(lin % nx), j = (uint32_t)((lin / nx) % ny), k = (uint32_t)(lin / ((uint64_t)nx * ny));
        const unsigned v = vol[storage_index(layout, i, j, k, nx, ny, bnx, bny)];
        if (pass == 0) {
            xmn = v < xmn ? v : xmn; xmx = v > xmx ? v : xmx;        // exact range (kernel selection)
            if (lin == 8390640ull) continue;                          // reference scan skips this index
            mn = v < mn ? v : mn; mx = v > mx ? v : mx;
        } else {
            unsigned bin = v;
            if (sizeof(VoxelT) == 2) {
                // val = std::round(val * 255.0f / max_dataset_val) stored to uint16_t
                const float f = roundf(((float)v * 255.0f) / scale255);
                bin = (unsigned)f & 0xffffu;
            }
            if (bin != 0 && bin < 256) atomicAdd(&lh[bin], 1u);
        }
    }
    if (pass == 0) {
        // wave-level then one atomic per wave
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (unsigned)__shfl_xor((int)mn, o)); mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            xmn = min(xmn, (unsigned)__shfl_xor((int)xmn, o)); xmx = max(xmx, (unsigned)__shfl_xor((int)xmx, o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&minmax[0], mn); atomicMax(&minmax[1], mx);
            atomicMin(&minmax[2], xmn); atomicMax(&minmax[3], xmx);
        }
    } else {
        __syncthreads();
        if (threadIdx.x < 256 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
    }
}

// per-cell (8x8x8 voxels) maximum, then its 3x3x3 dilation: the grid the fast kernel probes
// for exact empty-space skipping
template <typename VoxelT>
__global__ __launch_bounds__(256) void cellmax_kernel(const VoxelT *__restrict__ vol, uint16_t *__restrict__ out, uint32_t nx,
                                                      uint32_t ny, uint32_t nz, int layout, uint32_t bnx, uint32_t bny,
                                                      uint32_t cnx, uint32_t cny, uint32_t cnz)
{
    // one wavefront per cell: 64 lanes x 8 voxels
    const uint64_t cell = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (cell >= (uint64_t)cnx * cny * cnz) return;
    const uint32_t ci = (uint32_t)(cell % cnx), cj = (uint32_t)((cell / cnx) % cny), ck = (uint32_t)(cell / ((uint64_t)cnx * cny));
    const unsigned lane = threadIdx.x & 63u;
    unsigned m = 0;
    for (unsigned t = lane; t < 512u; t += 64u) {
        const uint32_t i = ci * 8u + (t & 7u), j = cj * 8u + ((t >> 3) & 7u), k = ck * 8u + (t >> 6);
        if (i < nx && j < ny && k < nz) {
            const unsigned v = vol[storage_index(layout, i, j, k, nx, ny, bnx, bny)];
            m = v > m ? v : m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) out[cell] = (uint16_t)m;
}

__global__ __launch_bounds__(256) void dilate_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int cnx,
                                                     int cny, int cnz)
{
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (int64_t)cnx * cny * cnz) return;
    const int ci = (int)(cell % cnx), cj = (int)((cell / cnx) % cny), ck = (int)(cell / ((int64_t)cnx * cny));
    unsigned m = 0;
    for (int dk = -1; dk <= 1; dk++)
        for (int dj = -1; dj <= 1; dj++)
            for (int di = -1; di <= 1; di++) {
                const int i = ci + di, j = cj + dj, k = ck + dk;
                if (i < 0 || j < 0 || k < 0 || i >= cnx || j >= cny || k >= cnz) continue;
                const unsigned v = in[(int64_t)i + (int64_t)cnx * ((int64_t)j + (int64_t)cny * k)];
                m = v > m ? v : m;
            }
    out[cell] = (uint16_t)m;
}


// ------------------------------------------------------------------ host launchers of the helper kernels
hipError_t launch_build_skip_grid(const void *vol, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                                  uint32_t bnx, uint32_t bny, uint16_t *tmp, uint16_t *out, hipStream_t st)
{
    const uint32_t cnx = (nx + 7) / 8, cny = (ny + 7) / 8, cnz = (nz + 7) / 8;
    const uint64_t cells = (uint64_t)cnx * cny * cnz;
    const unsigned blocks_a = (unsigned)((cells + 3) / 4), blocks_b = (unsigned)((cells + 255) / 256);
    if (bytes_per_voxel == 1)
        hipLaunchKernelGGL(cellmax_kernel<uint8_t>, dim3(blocks_a), dim3(256), 0, st, (const uint8_t *)vol, tmp, nx, ny, nz,
                           layout, bnx, bny, cnx, cny, cnz);
    else
        hipLaunchKernelGGL(cellmax_kernel<uint16_t>, dim3(blocks_a), dim3(256), 0, st, (const uint16_t *)vol, tmp, nx, ny, nz,
                           layout, bnx, bny, cnx, cny, cnz);
    hipLaunchKernelGGL(dilate_kernel, dim3(blocks_b), dim3(256), 0, st, tmp, out, (int)cnx, (int)cny, (int)cnz);
    return hipGetLastError();
}

// the smallest value of the dilated grid: a threshold below it can skip nothing (host: refreshSkipGrid -- such a launch runs the
// instances without skipping, whose loops are a few percent leaner).  Computed ONCE when the grid is built (round-5 advisor: the
// per-threshold count it replaces drained the launch stream on every frame of a window / transfer-function slider drag)
__global__ __launch_bounds__(256) void min_cell_kernel(const uint16_t *__restrict__ grid, uint64_t cells, unsigned *__restrict__ out)
{
    unsigned m = 0xffffffffu;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (uint64_t)gridDim.x * blockDim.x) m = min(m, (unsigned)grid[c]);
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63u) == 0u) atomicMin(out, m);
}

hipError_t launch_min_cell(const uint16_t *grid, uint64_t cells, unsigned *out, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0xff, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)std::min<uint64_t>((cells + 255) / 256, 2048u);
    hipLaunchKernelGGL(min_cell_kernel, dim3(blocks), dim3(256), 0, st, grid, cells, out);
    return hipGetLastError();
}

// Expected cost of every tile under empty-space skipping (host: refreshTileSchedule).  The tile order is longest-first by the
// tiles' expected ray lengths (tile_schedule.cpp); with skipping a ray's cost is the length of its VISIBLE stretch plus a small
// share of the empty one, and a frame is only two or three rounds of tiles, so the order must know it: tiles through the
// ball's centre first, not the ones that cross the cube's empty corners (cfg3 windowed, TRILINEAR: the frame ended 0.3 ms after
// the work was done, waiting for full-length tiles that had started last).  One thread per (tile, probe pixel): the same nine
// probe pixels as the host's estimate, 128 equidistant probes of the dilated cell-max grid along the ray's stretch in the box.
// An ESTIMATE: positions go through the affine map, not the shader's divisions; the image never depends on it.
__global__ __launch_bounds__(256) void tile_visible_work_kernel(const FrameParams P, const uint16_t *__restrict__ grid, int rows, unsigned tiles_x,
                                                                unsigned tiles_y, unsigned tile_w, unsigned tile_h, float empty_cost,
                                                                unsigned *__restrict__ work)
{
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned tile = id / 9u, probe = id % 9u;
    if (tile >= tiles_x * tiles_y) return;
    const unsigned tx = tile % tiles_x, ty = tile / tiles_x;
    const int lx = min((int)(tx * tile_w + (probe % 3u) * (tile_w - 1u) / 2u), P.img_w - 1);
    int ly = (int)(ty * tile_h + (probe / 3u) * (tile_h - 1u) / 2u);
    ly = min(ly, rows - 1);
    int py;
    if (P.stripe_count > 1) {
        const int st = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (st * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    py = min(py, P.img_h - 1);
    const Ray ray = compute_ray(P, (float)lx + 0.5f, (float)py + 0.5f);
    float t_min = 0.0f, t_max = 0.0f;
    if (!intersect_ray_aabb(P, ray, t_min, t_max)) return;
    const float len = fminf((t_max - t_min) / P.step, (float)P.max_steps);
    constexpr int K = 128;
    int visible = 0;
    for (int k = 0; k < K; k++) {
        const float t = t_min + (t_max - t_min) * (((float)k + 0.5f) / (float)K);
        const float ax = ray.ox + ray.dx * t, ay = ray.oy + ray.dy * t, az = ray.oz + ray.dz * t;
        const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
        const float uzr = (az + P.half[2]) * P.rext[2], uz = 1.0f - uzr;
        float tcx = ux, tcy = uy, tcz = uz;
        if (P.view_top == 1) { tcy = uzr; tcz = uy; }
        else if (P.view_bottom == 1) { tcy = uz; tcz = 1.0f - uy; }
        const int ci = min(max((int)(tcx * P.fdim[0]) >> 3, 0), P.cnx - 1), cj = min(max((int)(tcy * P.fdim[1]) >> 3, 0), P.cny - 1);
        const int ck = min(max((int)(tcz * P.fdim[2]) >> 3, 0), P.cnz - 1);
        visible += (int)grid[((size_t)ck * (size_t)P.cny + (size_t)cj) * (size_t)P.cnx + (size_t)ci] > P.skip_thresh ? 1 : 0;
    }
    const float f = (float)visible / (float)K;
    const float w = fmaxf(len * (f + (1.0f - f) * empty_cost), 1.0e-3f);
    atomicMax(&work[2u * tile], __float_as_uint(w));                     // (positive floats order like their bit patterns)
    atomicMax(&work[2u * tile + 1u], __float_as_uint(fmaxf(len, 1.0e-3f)));
}

// The same for the LDS-staged TRILINEAR kernel, which skips a tile's brick LAYER only when no ray of the tile touches anything
// visible in it (vr_tslab.hip: SKIP): one thread per (tile, cell layer along the tile's major axis); the layer counts as sampled
// when any of the nine probe rays stands in a visible cell where it crosses the layer's middle plane, as crossed when they are all in
// empty ones, and not at all where no probe ray is inside the box.  (With per-ray visibility the order was right for views along an axis and wrong for
// oblique ones, whose slanted layers mostly touch the ball somewhere: 1.46 -> 2.18 ms at the off-axis pose.)
__global__ __launch_bounds__(256) void tile_layer_work_kernel(const FrameParams P, const uint16_t *__restrict__ grid, int rows, unsigned tiles_x,
                                                              unsigned tiles_y, unsigned tile_w, unsigned tile_h, float empty_cost, unsigned kmax,
                                                              float *__restrict__ work)
{
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned tile = id / kmax, j = id % kmax;
    if (tile >= tiles_x * tiles_y) return;
    const unsigned tx = tile % tiles_x, ty = tile / tiles_x;
    auto to_voxel = [&](float ax, float ay, float az, float (&v)[3]) {
        const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
        const float uzr = (az + P.half[2]) * P.rext[2], uz = 1.0f - uzr;
        float tcx = ux, tcy = uy, tcz = uz;
        if (P.view_top == 1) { tcy = uzr; tcz = uy; }
        else if (P.view_bottom == 1) { tcy = uz; tcz = 1.0f - uy; }
        v[0] = tcx * P.fdim[0]; v[1] = tcy * P.fdim[1]; v[2] = tcz * P.fdim[2];
    };
    int m = 2;
    bool in_box = false, visible = false;
    const int cn[3] = {P.cnx, P.cny, P.cnz};
    // (probe 4, the tile's central pixel, first: it names the major axis)
    for (unsigned q = 0; q < 9u; q++) {
        const unsigned probe = q == 0u ? 4u : (q <= 4u ? q - 1u : q);
        const int lx = min((int)(tx * tile_w + (probe % 3u) * (tile_w - 1u) / 2u), P.img_w - 1);
        const int ly = min((int)(ty * tile_h + (probe / 3u) * (tile_h - 1u) / 2u), rows - 1);
        int py;
        if (P.stripe_count > 1) {
            const int st = ly / P.stripe_rows, r = ly % P.stripe_rows;
            py = (st * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
        } else {
            py = P.row_begin + ly;
        }
        py = min(py, P.img_h - 1);
        const Ray ray = compute_ray(P, (float)lx + 0.5f, (float)py + 0.5f);
        float E[3], F[3];
        to_voxel(ray.ox, ray.oy, ray.oz, E);
        to_voxel(ray.ox + ray.dx, ray.oy + ray.dy, ray.oz + ray.dz, F);
        const float G[3] = {F[0] - E[0], F[1] - E[1], F[2] - E[2]};
        if (q == 0u) {
            const float g0 = fabsf(G[0]), g1 = fabsf(G[1]), g2 = fabsf(G[2]);
            m = (g0 >= g1 && g0 >= g2) ? 0 : (g1 >= g2 ? 1 : 2);
            if ((int)j >= cn[m]) return;
        }
        float t_min = 0.0f, t_max = 0.0f;
        if (!intersect_ray_aabb(P, ray, t_min, t_max)) continue;
        if (!(fabsf(G[m]) > 1.0e-6f)) continue;
        const float t = ((float)(8u * j + 4u) - E[m]) / G[m];
        if (!(t >= t_min && t <= t_max)) continue;
        in_box = true;
        const int ci = min(max((int)(E[0] + t * G[0]) >> 3, 0), P.cnx - 1), cj = min(max((int)(E[1] + t * G[1]) >> 3, 0), P.cny - 1);
        const int ck = min(max((int)(E[2] + t * G[2]) >> 3, 0), P.cnz - 1);
        visible = visible || (int)grid[((size_t)ck * (size_t)P.cny + (size_t)cj) * (size_t)P.cnx + (size_t)ci] > P.skip_thresh;
    }
    if (in_box) { atomicAdd(&work[2u * tile], visible ? 1.0f : empty_cost); atomicAdd(&work[2u * tile + 1u], 1.0f); }
}

hipError_t launch_tile_visible_work(const FrameParams &P, const uint16_t *grid, int rows, unsigned tile_w, unsigned tile_h, float empty_cost,
                                    int per_layer, float *work, hipStream_t st)
{
    const unsigned tiles_x = (unsigned)((P.img_w + (int)tile_w - 1) / (int)tile_w), tiles_y = (unsigned)((rows + (int)tile_h - 1) / (int)tile_h);
    const unsigned n = tiles_x * tiles_y;
    hipError_t e = hipMemsetAsync(work, 0, (size_t)n * 2u * sizeof(float), st);
    if (e != hipSuccess) return e;
    if (per_layer) {
        const unsigned kmax = (unsigned)std::max(P.cnx, std::max(P.cny, P.cnz));
        const uint64_t threads = (uint64_t)n * kmax;
        hipLaunchKernelGGL(tile_layer_work_kernel, dim3((unsigned)((threads + 255u) / 256u)), dim3(256), 0, st, P, grid, rows, tiles_x, tiles_y, tile_w, tile_h,
                           empty_cost, kmax, work);
    } else {
        hipLaunchKernelGGL(tile_visible_work_kernel, dim3((n * 9u + 255u) / 256u), dim3(256), 0, st, P, grid, rows, tiles_x, tiles_y, tile_w, tile_h,
                           empty_cost, reinterpret_cast<unsigned *>(work));
    }
    return hipGetLastError();
}

// 12-bit packed copy of a u16 volume whose voxels are all <= 4095: voxel with storage index s
// occupies bits [12s, 12s + 12) of a little-endian bit stream (8 voxels -> 3 dwords).  Any
// the resident volume (linear or cube-bricked) -> TRILINEAR's apron copy (vr_device.h: build_axis_tables_apron)
template <typename VoxelT>
__global__ __launch_bounds__(256) void relayout_apron_kernel(const VoxelT *__restrict__ vol, VoxelT *__restrict__ out, uint32_t nx, uint32_t ny,
                                                             uint32_t nz, int layout, uint32_t bnx, uint32_t bny, int order, uint64_t total)
{
    const uint32_t abx = (nx + 3u) >> 2, aby = (ny + 3u) >> 2;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t brick = s / APRON_BRICK_VOXELS;
        const uint32_t r = (uint32_t)(s % APRON_BRICK_VOXELS), pl = r % 5u, ql = (r / 5u) & 3u, ml = r / 20u;   // fastest (apron), middle, slowest
        const uint32_t xl = order == 2 ? ml : pl, yl = order == 0 ? ql : (order == 1 ? ml : pl), zl = order == 0 ? ml : ql;
        const uint32_t bx = (uint32_t)(brick % abx), by = (uint32_t)((brick / abx) % aby), bz = (uint32_t)(brick / ((uint64_t)abx * aby));
        const uint32_t i = min(4u * bx + xl, nx - 1u), j = min(4u * by + yl, ny - 1u), k = min(4u * bz + zl, nz - 1u);   // edge voxels repeat
        out[s] = vol[storage_index(layout, i, j, k, nx, ny, bnx, bny)];
    }
}

hipError_t launch_relayout_apron(const void *vol, void *out, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                                 uint32_t bnx, uint32_t bny, int order, hipStream_t st)
{
    const uint64_t total = apron_voxels((int)nx, (int)ny, (int)nz);
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 256u * 64u);
    if (bytes_per_voxel == 1)
        hipLaunchKernelGGL(relayout_apron_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t *)vol, (uint8_t *)out, nx, ny, nz, layout, bnx, bny, order, total);
    else
        hipLaunchKernelGGL(relayout_apron_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)vol, (uint16_t *)out, nx, ny, nz, layout, bnx, bny, order, total);
    return hipGetLastError();
}

// storage order works; the fast kernel reads the bricked one (PK12).
__global__ __launch_bounds__(256) void pack12_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ dst, uint64_t ngroups, uint32_t base)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        const uint4 q = src[g];
        // voxel - base: the caller guarantees base <= voxel <= base + 4095 for every voxel of the volume
        const uint32_t v0 = ((q.x & 0xffffu) - base) & 0xfffu, v1 = ((q.x >> 16) - base) & 0xfffu, v2 = ((q.y & 0xffffu) - base) & 0xfffu,
                       v3 = ((q.y >> 16) - base) & 0xfffu;
        const uint32_t v4 = ((q.z & 0xffffu) - base) & 0xfffu, v5 = ((q.z >> 16) - base) & 0xfffu, v6 = ((q.w & 0xffffu) - base) & 0xfffu,
                       v7 = ((q.w >> 16) - base) & 0xfffu;
        dst[3 * g + 0] = v0 | (v1 << 12) | (v2 << 24);
        dst[3 * g + 1] = (v2 >> 8) | (v3 << 4) | (v4 << 16) | (v5 << 28);
        dst[3 * g + 2] = (v5 >> 4) | (v6 << 8) | (v7 << 20);
    }
}


hipError_t launch_pack12(const void *src_u16, void *dst, uint64_t voxels, uint32_t base, hipStream_t st)
{
    hipLaunchKernelGGL(pack12_kernel, dim3(256 * 16), dim3(256), 0, st, (const uint4 *)src_u16, (uint32_t *)dst, voxels / 8u, base);
    return hipGetLastError();
}

// RGBA32F target -> RGBA8 (display / screenshot precision): glReadPixels' float -> unorm8 rule,
// round(clamp(c, 0, 1) * 255), NaN -> 0; row order unchanged
__global__ __launch_bounds__(256) void to_rgba8_kernel(const float4 *__restrict__ fb, uint32_t *__restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = fb[i];
    auto q = [](float v) -> uint32_t {
        v = v != v ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
        return (uint32_t)floorf(v * 255.0f + 0.5f);
    };
    out[i] = q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
}

hipError_t launch_to_rgba8(const void *fb_rgba32f, void *out_rgba8, uint64_t pixels, hipStream_t st)
{
    hipLaunchKernelGGL(to_rgba8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, st, (const float4 *)fb_rgba32f,
                       (uint32_t *)out_rgba8, pixels);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void assemble_kernel(const float *__restrict__ gathered, float4 *__restrict__ frame, int W, int H,
                                                       int n, int local_rows, int stripe_rows, int channels)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), gy = (int)blockIdx.y;
    if (x >= W || gy >= H) return;
    int k, lr;
    if (stripe_rows == 0) { k = gy / local_rows; lr = gy % local_rows; }
    else { const int s = gy / stripe_rows; k = s % n; lr = (s / n) * stripe_rows + gy % stripe_rows; }
    const size_t src = ((size_t)k * (size_t)local_rows + (size_t)lr) * (size_t)W + (size_t)x;
    float4 px;
    if (channels == 2) { const float2 ga = reinterpret_cast<const float2 *>(gathered)[src]; px = make_float4(ga.x, ga.x, ga.x, ga.y); }
    else px = reinterpret_cast<const float4 *>(gathered)[src];
    frame[(size_t)gy * (size_t)W + (size_t)x] = px;
}

hipError_t launch_assemble(const void *gathered, float4 *frame, int W, int H, int n, int local_rows, int stripe_rows,
                           int channels, hipStream_t st)
{
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, st, (const float *)gathered, frame, W, H,
                       n, local_rows, stripe_rows, channels);
    return hipGetLastError();
}

// streaming read of a device buffer with 16-byte loads: the box's achievable HBM read rate,
// measured next to the ray-march (vr_measure_stream_read)
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *__restrict__ p, uint64_t n16, unsigned *sink)
{
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1u);   // keeps the loads alive; practically never taken
}

hipError_t launch_stream_read(const void *p, uint64_t bytes, unsigned *sink, hipStream_t st)
{
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, st, (const uint4 *)p, bytes / 16u, sink);
    return hipGetLastError();
}

hipError_t launch_certify_div(float b, float r, unsigned *d_bad, hipStream_t st)
{
    hipLaunchKernelGGL(certify_div_kernel, dim3((1u << 23) / 256u), dim3(256), 0, st, b, r, d_bad);
    return hipGetLastError();
}

hipError_t launch_gen_volume(void *out, int bytes_per_voxel, int kind, uint32_t nx, uint32_t ny, uint32_t nz,
                             uint32_t param, int layout, uint32_t bnx, uint32_t bny, hipStream_t st)
{
    const unsigned blocks = 256 * 16;
    if (bytes_per_voxel == 1)
        hipLaunchKernelGGL(gen_volume_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (uint8_t *)out, kind, nx, ny,
                           nz, param, layout, bnx, bny);
    else
        hipLaunchKernelGGL(gen_volume_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (uint16_t *)out, kind, nx,
                           ny, nz, param, layout, bnx, bny);
    return hipGetLastError();
}

hipError_t launch_relayout(const void *in, void *out, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz,
                           uint32_t bnx, uint32_t bny, int to_linear, hipStream_t st)
{
    const unsigned blocks = 256 * 16;
    if (bytes_per_voxel == 1)
        hipLaunchKernelGGL(relayout_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t *)in,
                           (uint8_t *)out, nx, ny, nz, bnx, bny, to_linear);
    else
        hipLaunchKernelGGL(relayout_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)in,
                           (uint16_t *)out, nx, ny, nz, bnx, bny, to_linear);
    return hipGetLastError();
}

hipError_t launch_stats(const void *vol, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                        uint32_t bnx, uint32_t bny, int pass, float scale255, unsigned *d_minmax, unsigned *d_hist,
                        hipStream_t st)
{
    const unsigned blocks = 256 * 8;
    if (bytes_per_voxel == 1)
        hipLaunchKernelGGL(stats_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t *)vol, nx, ny, nz,
                           layout, bnx, bny, pass, scale255, d_minmax, d_hist);
    else
        hipLaunchKernelGGL(stats_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, (const uint16_t *)vol, nx, ny,
                           nz, layout, bnx, bny, pass, scale255, d_minmax, d_hist);
    return hipGetLastError();
}
