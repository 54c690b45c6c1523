// vr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the ray-march hot path.
//
// Replaces the GLSL compute shader /root/reference/VolumeRenderer.cs:55-238 that
// RendererCore::render() dispatches (/root/reference/src/RendererCore.cpp:138-163).
// gfx950 has no texture units (tex3D is compile-time unavailable), so the volume is
// sampled in software from plain HBM allocations.
//
// Arithmetic contract (tests/test_ref_gl_goldens.py, tests/test_parity_gpu.py): the arithmetic of the reference shader AS
// EXECUTED by a GL (Mesa 23.2.1 llvmpipe; measured per operation by oracle/ref_gl/probe_arith.py, frames pinned by
// tests/golden/ref_gl/): every fp32 operator is one correctly rounded IEEE-754 operation in the shader's order, nothing
// is contracted, normalize(v) = v * (1 / sqrt(dot)) with the dot summed from the last component to the first, and the
// top view's 1 - (1 - z) is z.  Compiled with -ffp-contract=off and hipcc's default correctly rounded fp32 divide/sqrt.
// The only fused operations are the explicit fmaf() of div_cert(), which is used only for divisors that
// certify_div_kernel() has proven (exhaustively over all 2^23 significands) to give the correctly rounded quotient.
//
// Thread mapping: one pixel per lane, an 8x8 pixel tile per 64-wide wavefront (8 rows x 8 columns: neighbouring rays
// touch neighbouring voxels).  Fast kernel: 8 wavefronts = 32x16 pixels per 512-thread workgroup, marching in lockstep;
// relay kernel: 4 wavefronts per 8x8 tile, two tiles per workgroup; generic kernel: 4 wavefronts = 16x16 pixels.
// Workgroups take their tile from a host-built longest-first table dealt round-robin to the 8 XCDs (tile_schedule.cpp);
// without a table, tile_of_block() gives every XCD whole tile rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vr_device.h"
#include "vr_lds_dma.h"

// Translation units: the ray-march kernels of one (voxel type, layout) pair are ~150 template
// instances each, so the Makefile compiles this file five times in parallel:
//   VR_TU = 0..3  ray-march kernels of (u8|u16) x (linear|bricked) + their dispatch
//   VR_TU = -1    helper kernels, launchers and the top-level launch_raymarch
//   VR_TU = -2    (default) everything in one translation unit
#ifndef VR_TU
#define VR_TU -2
#endif
#define VR_TU_MAIN (VR_TU < 0)

namespace vr {

// Exhaustive proof for one divisor: for every significand a in [1,2) (sign and
// exponent do not change the rounding pattern in the normal range) the Markstein
// quotient equals IEEE division.  bad != 0 afterwards means "use DIV_EXACT".
#if VR_TU_MAIN
__global__ void certify_div_kernel(float b, float r, unsigned *bad)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 2^23 threads
    float a = __uint_as_float(0x3f800000u | i);
    float q = div_cert(a, b, r);
    float ref = a / b;
    if (__float_as_uint(q) != __float_as_uint(ref)) atomicOr(bad, 1u);
    // also the scaled-down operand range the texcoord numerator lives in
    float a2 = a * 0.00390625f;
    if (__float_as_uint(div_cert(a2, b, r)) != __float_as_uint(a2 / b)) atomicOr(bad, 1u);
}
#endif  // VR_TU_MAIN

// ------------------------------------------------------------------ generic kernel
// One kernel that follows the shader line by line and takes every mode as a run-time
// (wave-uniform) switch.  It is the correctness backbone: every configuration the
// specialised kernels do not cover runs here.
// BIG = false: 32-bit voxel offsets through a bounds-checked buffer resource (VoxelAddr /
// VoxelFetch); the eight taps of TRILINEAR share their per-axis address terms.
template <typename VoxelT, int LAYOUT, bool COUNT, bool BIG>
__global__ __launch_bounds__(256) void raymarch_generic_kernel(const FrameParams P, const int filter,
                                                               const int is_mip, const int divmode,
                                                               const uint32_t vol_bytes,
                                                               const VoxelT *__restrict__ vol,
                                                               const float4 *__restrict__ tf,
                                                               float4 *__restrict__ fb,
                                                               uint32_t *__restrict__ spp,
                                                               const unsigned tiles_x, const unsigned tiles_y)
{
    unsigned tx, ty;
    tile_of_block(blockIdx.x, tiles_x, tiles_y, tx, ty);
    if (tx == 0xffffffffu) return;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(tx * 16u + (wave & 1u) * 8u + (lane & 7u));
    const int ly = (int)(ty * 16u + (wave >> 1) * 8u + (lane >> 3));
    // local row -> global row (contiguous shard or cyclic stripes)
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    if (px >= P.col_lim || py >= P.row_lim || py >= P.row_end) return;

    const Ray ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
    float t_min = 0.0f, t_max = 0.0f;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
    uint32_t fetches = 0;
    if (intersect_ray_aabb(P, ray, t_min, t_max)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, BIG ? 0 : (int)pair_load_extent(vol_bytes), 0x00020000);
        // per-axis terms of VoxelAddr<LAYOUT, false>::at(i, j, k) = X(i) + Y(j) + Z(k)  (mod 2^32)
        auto term_x = [&](int i) -> uint32_t {
            return LAYOUT == 0 ? (uint32_t)i : mad_u24((uint32_t)i >> BRICK_LX, 64u - (uint32_t)BRICK_X, (uint32_t)i);
        };
        auto term_y = [&](int j) -> uint32_t {
            if (LAYOUT == 0) return mad_u24((uint32_t)j, (uint32_t)P.nx, 0u);
            return mad_u24(BRICK_LY ? (uint32_t)j >> BRICK_LY : (uint32_t)j, P.bstride_y, BRICK_LY ? (uint32_t)j << BRICK_LX : 0u);
        };
        auto term_z = [&](int k) -> uint32_t {
            if (LAYOUT == 0) return mad_u24(mad_u24((uint32_t)k, (uint32_t)P.ny, 0u), (uint32_t)P.nx, 0u);
            return mad_u24(BRICK_LZ ? (uint32_t)k >> BRICK_LZ : (uint32_t)k, P.bstride_z, BRICK_LZ ? (uint32_t)k << (BRICK_LX + BRICK_LY) : 0u);
        };
        auto tap = [&](uint32_t off) -> float { return (float)VoxelFetch<VoxelT, false>::load(vol, rs, off); };
        const float EPSILON = 0.000001f;
        const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
        const float p0x = sx + ray.dx * EPSILON, p0y = sy + ray.dy * EPSILON, p0z = sz + ray.dz * EPSILON;
        const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
        float qx = p0x, qy = p0y, qz = p0z;
        for (int i = 0; i < P.max_steps; i++) {
            if (P.accum == 1) {
                const float fi = (float)i;
                qx = p0x + fi * dsx; qy = p0y + fi * dsy; qz = p0z + fi * dsz;
            }
            // cartesianToTextureCoord (VolumeRenderer.cs:175-192)
            float ux = qx + P.half[0], uy = qy + P.half[1], uz = qz + P.half[2];
            if (divmode == DIV_CERT) {
                ux = div_cert(ux, P.ext[0], P.rext[0]);
                uy = div_cert(uy, P.ext[1], P.rext[1]);
                uz = div_cert(uz, P.ext[2], P.rext[2]);
            } else {
                ux = ux / P.ext[0]; uy = uy / P.ext[1]; uz = uz / P.ext[2];
            }
            const float uzr = uz;   // z before the flip of :185
            uz = 1.0f - uz;
            float tcx, tcy, tcz;
            if (P.view_top == 1) { tcx = ux; tcy = uzr; tcz = uy; }   // 1 - (1 - z) as the GL compiles it: z
            else if (P.view_bottom == 1) { tcx = ux; tcy = uz; tcz = 1.0f - uy; }
            else { tcx = ux; tcy = uy; tcz = uz; }
            if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || d3 >= 0.95f)
                break;
            float s;
            if (filter == 0) {
                const int vi = clampi(floor_to_int_sat(tcx * P.fdim[0]), 0, P.nx - 1);
                const int vj = clampi(floor_to_int_sat(tcy * P.fdim[1]), 0, P.ny - 1);
                const int vk = clampi(floor_to_int_sat(tcz * P.fdim[2]), 0, P.nz - 1);
                s = BIG ? fetch_voxel<VoxelT, LAYOUT>(P, vol, vi, vj, vk) : tap(term_x(vi) + term_y(vj) + term_z(vk));
            } else {
                const float u = tcx * P.fdim[0] - 0.5f, v = tcy * P.fdim[1] - 0.5f, w = tcz * P.fdim[2] - 0.5f;
                const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
                const float ax = u - fu, ay = v - fv, az = w - fw;
                const int iu = (int)fu, iv = (int)fv, iw = (int)fw;
                const int i0 = clampi(iu, 0, P.nx - 1), i1 = clampi(iu + 1, 0, P.nx - 1);
                const int j0 = clampi(iv, 0, P.ny - 1), j1 = clampi(iv + 1, 0, P.ny - 1);
                const int k0 = clampi(iw, 0, P.nz - 1), k1 = clampi(iw + 1, 0, P.nz - 1);
                float c000, c100, c010, c110, c001, c101, c011, c111;
                if (BIG) {
                    c000 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j0, k0); c100 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j0, k0);
                    c010 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j1, k0); c110 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j1, k0);
                    c001 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j0, k1); c101 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j0, k1);
                    c011 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j1, k1); c111 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j1, k1);
                } else {
                    const uint32_t x0 = term_x(i0), x1 = term_x(i1), y0 = term_y(j0), y1 = term_y(j1);
                    const uint32_t z0 = term_z(k0), z1 = term_z(k1);
                    // the two x-neighbours of a tap pair are adjacent in memory unless i0 is the last
                    // voxel of its brick row (or of the volume): every lane fetches its x0 tap together
                    // with the next storage element in one load (bounds-checked: the last element of the
                    // buffer reads 0 there), and only the lanes whose x1 lies elsewhere fetch it again
                    const bool pair = i1 == i0 + 1 && (LAYOUT == 0 || ((uint32_t)i0 & (BRICK_X - 1u)) != BRICK_X - 1u);
                    auto tap2 = [&](uint32_t off, float &lo, float &hi) {
                        if (sizeof(VoxelT) == 1) {
                            const uint32_t v = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
                            lo = (float)(v & 0xffu); hi = (float)(v >> 8);
                        } else {
                            const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off << 1), 0, 0);
                            lo = (float)(v & 0xffffu); hi = (float)(v >> 16);
                        }
                    };
                    tap2(x0 + y0 + z0, c000, c100); tap2(x0 + y1 + z0, c010, c110);
                    tap2(x0 + y0 + z1, c001, c101); tap2(x0 + y1 + z1, c011, c111);
                    if (!pair) {
                        c100 = tap(x1 + y0 + z0); c110 = tap(x1 + y1 + z0); c101 = tap(x1 + y0 + z1); c111 = tap(x1 + y1 + z1);
                    }
                }
                const float c00 = tri_lerp(c000, c100, ax), c10 = tri_lerp(c010, c110, ax);
                const float c01 = tri_lerp(c001, c101, ax), c11 = tri_lerp(c011, c111, ax);
                const float c0 = tri_lerp(c00, c10, ay), c1 = tri_lerp(c01, c11, ay);
                s = tri_lerp(c0, c1, az);
            }
            fetches++;
            // window (VolumeRenderer.cs:122-124; Q4: max==min defined as 0)
            s = gl_min(gl_max(s, P.fmin), P.fmax);
            if (P.fden == 0.0f) s = 0.0f;
            else if (s <= P.fmax && s >= P.fmin) s = (s - P.fmin) / P.fden;
            float s0 = s, s1 = s, s2 = s, s3 = s;
            if (P.tf_len > 1) {
                const float fi = s * (float)(P.tf_len - 1) + 0.5f;
                int idx = floor_to_int_sat(fi);
                idx = clampi(idx, 0, P.tf_len - 1);
                const float4 t = tf[idx];
                s0 = t.x; s1 = t.y; s2 = t.z; s3 = t.w;
            }
            if (is_mip == 1) {
                s0 *= P.alpha_scale; s1 *= P.alpha_scale; s2 *= P.alpha_scale; s3 *= P.alpha_scale;
                if (d3 < s3) { d0 = s0; d1 = s1; d2 = s2; d3 = s3; }
            } else {
                s3 *= P.alpha_scale;
                s0 *= s3; s1 *= s3; s2 *= s3;
                const float om = 1.0f - d3;
                d0 += s0 * om; d1 += s1 * om; d2 += s2 * om; d3 += s3 * om;
                if (d3 > 0.99f) break;
            }
            if (P.accum == 0) { qx += dsx; qy += dsy; qz += dsz; }
        }
    }
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    store_pixel(P, fb, pix, d0, d1, d2, d3);
    if (COUNT) spp[pix] = fetches;
}

// Workgroup = 512 threads = 8 wavefronts = a 32x16-pixel tile (4x2 wave tiles of 8x8).
// Normally the host's longest-first tile table (tile_schedule.cpp) maps blocks to tiles.  The
// arithmetic order below is the table-less fallback (RendererCore::tile_order = 0): blocks go
// to XCDs in chunks of FAST_CHUNK horizontally adjacent tiles, and with an odd number of chunks
// per tile row the owner (chunk index mod 8) rotates from row to row, so every XCD gets an
// equal share of every image region.
constexpr unsigned FAST_THREADS = 512, FAST_TILE_W = kFastTileW, FAST_TILE_H = kFastTileH, FAST_CHUNK = kFastChunk;

struct FastGrid { unsigned tiles_x, tiles_y, chunks_per_row, blocks; };

[[maybe_unused]] static inline FastGrid fast_grid(int img_w, int rows)
{
    FastGrid g;
    g.tiles_x = (unsigned)((img_w + (int)FAST_TILE_W - 1) / (int)FAST_TILE_W);
    g.tiles_y = (unsigned)((rows + (int)FAST_TILE_H - 1) / (int)FAST_TILE_H);
    g.chunks_per_row = (g.tiles_x + FAST_CHUNK - 1) / FAST_CHUNK;
    g.chunks_per_row |= 1u;                                   // odd: owners rotate per row
    const unsigned chunks = g.chunks_per_row * g.tiles_y;
    g.blocks = ((chunks + 7u) / 8u) * 8u * FAST_CHUNK;
    return g;
}

__device__ __forceinline__ bool fast_tile_of_block(unsigned b, unsigned tiles_x, unsigned tiles_y,
                                                   unsigned chunks_per_row, unsigned &tx, unsigned &ty)
{
    const unsigned xcd = b & 7u, slot = b >> 3;
    const unsigned c = (slot / FAST_CHUNK) * 8u + xcd;       // chunk index, owner = c % 8
    ty = c / chunks_per_row;
    tx = (c % chunks_per_row) * FAST_CHUNK + (slot % FAST_CHUNK);
    return ty < tiles_y && tx < tiles_x;
}

// POW2: every volume dimension is a power of two and the box extents are 1 (DIV_UNIT), so
// texcoord*N is an exact scaling and the prefix can be marched in voxel units:
// fl(q*S + ds*S) == S*fl(q + ds) for a power-of-two S, bit for bit (also in the subnormal
// range, where fp32 addition is exact), which drops the three per-sample multiplies.
// NOCLAMP: the dataset's exact min/max lie inside the window, so clamp() is the identity.
// MODE: 0 = grey-ramp emission/absorption composite (rayMarchVolume, :104-139)
//       1 = MIP with the grey ramp (MIP(), :141-173): all four channels carry max(v*alpha)
//       2 = composite through the 1-D transfer function table (needs LUT; entries are RGBA)
// SKIPT: empty-space skipping compiled in (its probe state costs ~18 VGPRs = one workgroup
// of occupancy per CU, so the headline variant is also built without it).
// BATCH: samples per gather batch (8: the skip grid's dilation covers exactly that).
template <typename VoxelT, int LAYOUT, int DIVTC, int VIEW, bool BIG, bool LUT, bool POW2, bool NOCLAMP, int MODE, bool SKIPT, int BATCH, bool ATAB, bool PK12, bool PIPE = false>
__global__ __launch_bounds__(512, 1) void raymarch_fast_kernel(const FrameParams P,
                                                            const VoxelT *__restrict__ vol,
                                                            const float4 *__restrict__ tf,
                                                            const uint32_t vol_bytes,
                                                            float4 *__restrict__ fb,
                                                            uint32_t *__restrict__ spp,
                                                            const unsigned tiles_x, const unsigned tiles_y,
                                                            const unsigned chunks_per_row,
                                                            const uint32_t *__restrict__ tile_table,
                                                            const uint16_t *__restrict__ skip_grid,
                                                            const uint32_t skip_grid_bytes,
                                                            const void *__restrict__ packed12,
                                                            const uint32_t packed12_bytes)
{
    static_assert(BATCH == 8 || !SKIPT, "empty-space skipping assumes 8-sample batches");
    static_assert(!PIPE || (!SKIPT && !BIG), "the pipelined loop issues gathers for lanes without a next batch: bounds-checked buffer loads, no skip branch");
    static_assert(!PK12 || (ATAB && sizeof(VoxelT) == 2 && LAYOUT == 1 && BATCH <= 8), "12-bit copy: u16 bricks through the address tables");
    __shared__ float lut[LUT ? (ATAB && BIG ? FAST_BIG_LUT_FLOATS : FAST_LUT_MAX * 2) : 4];   // 32 KiB: 4096 x (c,a) or 256 x (r,g,b,a) + index bytes
    __shared__ uint32_t axis_tab[ATAB ? (BIG ? FAST_AXIS_TAB_BIG_MAX : FAST_AXIS_TAB_MAX) : 1];
    static_assert(!(ATAB && BIG) || LAYOUT == 1, "64-bit address tables exist for the bricked layout");
    unsigned tx, ty;
    if (tile_table) {                                       // host-built longest-first order
        const uint32_t t = tile_table[blockIdx.x];
        if (t == 0xffffffffu) return;                        // padding block
        tx = t & 0xffffu; ty = t >> 16;
    } else if (!fast_tile_of_block(blockIdx.x, tiles_x, tiles_y, chunks_per_row, tx, ty)) {
        return;                                              // padding block
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned mx = lane & 7u, my = lane >> 3;
    const int lx = (int)(tx * FAST_TILE_W + (wave & 3u) * 8u + mx);
    const int ly = (int)(ty * FAST_TILE_H + (wave >> 2) * 8u + my);
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    const bool in_image = !(px >= P.col_lim || py >= P.row_lim || py >= P.row_end);

    Ray ray = {};
    float t_min = 0.0f, t_max = 0.0f;
    bool hit = false;
    if (in_image) {
        ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
        hit = intersect_ray_aabb(P, ray, t_min, t_max);
    }
    if (LUT || ATAB) {
        // tabulate only if some ray of the workgroup enters the volume
        if (__syncthreads_or(hit ? 1 : 0)) {
            if constexpr (ATAB) build_axis_tables<VoxelT, LAYOUT, PK12, BIG>(P, axis_tab, (int)FAST_THREADS);
            const int n = LUT ? P.max_val - P.min_val + 1 : 0;
            for (int e = (int)threadIdx.x; e < n; e += (int)FAST_THREADS) {
                const float s = (float)(P.min_val + e);          // == clamp(float(texel), fmin, fmax)
                const float v = div_cert(s - P.fmin, P.fden, P.rden);
                if (MODE >= 2) {
                    // classification through the transfer function: index = round(v*(len-1)) here,
                    // src.a *= alpha_scale, src.rgb *= src.a in the 256-entry table below
                    int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                    idx = clampi(idx, 0, P.tf_len - 1);
                    reinterpret_cast<uint8_t *>(lut)[FAST_TF_ENTRIES * 16 + e] = (uint8_t)idx;
                } else if (MODE == 0 && P.tf_grey != 0) {
                    // a GREY transfer function (r == g == b in every entry: the reference's own colour ramp, whose colour
                    // widget is commented out) composited: the table's (c, a) pair is the whole classification, so the
                    // grey-ramp instance runs it -- one LDS read per sample and 3 workgroups per CU instead of MODE 2's two
                    // reads and 2 workgroups (cfg4: 3.26 -> ms of the grey mode).  Entries with MODE 2's own operations;
                    // r == g == b of the frame holds bit for bit because the three channels see identical operands.
                    int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                    idx = clampi(idx, 0, P.tf_len - 1);
                    const float4 t = tf[idx];
                    const float a = t.w * P.alpha_scale;
                    lut[2 * e + 0] = t.x * a; lut[2 * e + 1] = a;
                } else {
                    const float a = v * P.alpha_scale;
                    lut[2 * e + 0] = v * a; lut[2 * e + 1] = a;
                }
            }
            if (LUT && MODE >= 2) {
                for (int e = (int)threadIdx.x; e < P.tf_len; e += (int)FAST_THREADS) {
                    const float4 t = tf[e];
                    const float a = t.w * P.alpha_scale;
                    if (MODE == 3) {            // MIP(): s *= alpha_scale on all four channels (VolumeRenderer.cs:164)
                        lut[4 * e + 0] = t.x * P.alpha_scale; lut[4 * e + 1] = t.y * P.alpha_scale; lut[4 * e + 2] = t.z * P.alpha_scale;
                    } else {                    // composite: src.a *= alpha_scale, src.rgb *= src.a (:130-131)
                        lut[4 * e + 0] = t.x * a; lut[4 * e + 1] = t.y * a; lut[4 * e + 2] = t.z * a;
                    }
                    lut[4 * e + 3] = a;
                }
            }
            __syncthreads();
        }
    }

    float drgb = 0.0f, dg = 0.0f, db = 0.0f, da = 0.0f;   // MODE 0/1: r == g == b bit for bit, only drgb is carried
    uint32_t fetches = 0;
    {                                   // every thread runs the (barrier-carrying) batch loop
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, BIG ? 0 : (int)vol_bytes, 0x00020000);
        // PK12: the prefix gathers from the 12-bit packed copy of the bricked volume (voxel with
        // storage index s = bits [12s, 12s+12) of the stream: 25 % fewer cache lines per frame)
        const __amdgpu_buffer_rsrc_t rs12 = __builtin_amdgcn_make_buffer_rsrc((void *)packed12, 0, (int)packed12_bytes, 0x00020000);
        const float EPSILON = 0.000001f;
        const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
        float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
        const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;

        // ---- checked head (vr_device.h: head_steps): the samples before the safe prefix can start.  Their positions are
        //      stepped here (the shader's additions); they are sampled through the checked loop further down
        const float hqx = qx, hqy = qy, hqz = qz;
        const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
        for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
        // ---- safe prefix length (samples head .. head + k_safe - 1)
        const int k_safe = hit ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;

        // texcoord -> voxel index of a position; valid (unclamped) inside the prefix
        auto voxel_of = [&](float ax, float ay, float az, int &vi, int &vj, int &vk) {
            const float ux = div_mode<DIVTC>(ax + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(ay + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(az + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }          // view_top
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }     // view_bottom
            vi = (int)(tcx * P.fdim[0]);
            vj = (int)(tcy * P.fdim[1]);
            vk = (int)(tcz * P.fdim[2]);
        };
        constexpr int LUT_SHIFT = 3;
        // byte offset of entry 0 relative to texel*entry_bytes (MODE 2: of index byte 0 relative to texel)
        // PK12: the packed copy stores voxel - pk12_base (the dataset minimum), so a volume whose RANGE fits 12 bits
        // packs, whatever its offset (CT data stored 1000 .. 5095).  classify() then works on the packed values: the
        // window limits and the table bias are shifted by the base instead of the texel -- no instruction is added to
        // the sample loop; the rare checked steps subtract the base from the u16 voxel they read.
        const int pkb = PK12 ? P.pk12_base : 0;
        const int wmin = P.min_val - pkb, wmax = P.max_val - pkb;
        const int lut_bias = MODE >= 2 ? FAST_TF_ENTRIES * 16 - wmin : -8 * wmin;
        uint32_t lut_entry0 = lds_offset_of(lut) + (uint32_t)lut_bias;
        asm volatile("" : "+v"(lut_entry0));
        // window + classification of one texel -> premultiplied colour c (cg, cb only in MODE 2)
        // and opacity a of VolumeRenderer.cs:130-131
        auto classify = [&](uint32_t texel, float &c, float &cg, float &cb, float &a) {
            if (LUT) {
                int t = (int)texel;
                if (!NOCLAMP) t = med3_i32(t, wmin, wmax);   // clamp(texel, min_val, max_val), min <= max
                if (MODE >= 2) {
                    const uint32_t idx = reinterpret_cast<const uint8_t *>(lut)[(uint32_t)(t + lut_bias)];
                    const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
                    c = q.x; cg = q.y; cb = q.z; a = q.w;
                } else {
                    // one v_lshl_add_u32 with the table's LDS address + bias in a VECTOR register (the compiler's form was shift,
                    // mask, add with an SGPR operand: two more slow-rate instructions per sample)
                    uint32_t entry;
                    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(entry) : "v"(t), "v"(lut_entry0));
                    static_assert(LUT_SHIFT == 3, "the shift is spelled out in the instruction above");
                    VR_LDS_AS const float *ca = reinterpret_cast<VR_LDS_AS const float *>((size_t)entry);      // (one ds_read_b64)
                    c = ca[0]; a = ca[1];
                }
            } else {
                float s = (float)(texel + (uint32_t)pkb);
                s = fminf(fmaxf(s, P.fmin), P.fmax);        // operands are never NaN here
                s = div_cert(s - P.fmin, P.fden, P.rden);
                a = s * P.alpha_scale;
                c = s * a;
            }
        };
        // POW2: per box axis the scale is the dimension of the voxel axis it maps to
        const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
        float Qx = qx * Sx, Qy = qy * Sy, Qz = qz * Sz;
        const float dSx = dsx * Sx, dSy = dsy * Sy, dSz = dsz * Sz;
        const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;
        // (in VECTOR registers for the sample loop: a VALU instruction with an SGPR operand issues at the slow rate on gfx950, 4.2
        // cycles instead of 2.3 -- fp32 and integer alike, tools/ubench/valu_rates2.hip)
        float Szv = Sz, Syv = Sy;
        asm volatile("" : "+v"(Szv), "+v"(Syv));
        // ---- exact empty-space skipping (vr_set_skip_empty): a batch is skipped when the
        // dilated cell-max grid says every voxel within one 8^3 cell of the batch's middle
        // sample classifies to (0,0,0,0), i.e. compositing it cannot change a single bit of
        // dest.  The grid is probed one batch ahead at an APPROXIMATE position (closed form
        // V0 + k*dV in voxel units; the 3x3x3 dilation covers the +-4.5 voxels of the batch
        // and the approximation error), so the probe never delays the gathers.
        const bool skip_on = SKIPT && P.skip_empty != 0 && skip_grid != nullptr;
        const __amdgpu_buffer_rsrc_t rs_grid = __builtin_amdgcn_make_buffer_rsrc((void *)skip_grid, 0, (int)skip_grid_bytes, 0x00020000);
        float V0x = 0.0f, V0y = 0.0f, V0z = 0.0f, dVx = 0.0f, dVy = 0.0f, dVz = 0.0f;
        if (skip_on) {
            // affine map position -> voxel coordinate per voxel axis (flips included)
            auto to_voxel = [&](float ax, float ay, float az, float &fx, float &fy, float &fz) {
                const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
                const float uzr = (az + P.half[2]) * P.rext[2], uz = 1.0f - uzr;
                float tcx = ux, tcy = uy, tcz = uz;
                if (VIEW == 1) { tcy = uzr; tcz = uy; }
                else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
                fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
            };
            float ex, ey, ez;
            to_voxel(qx, qy, qz, V0x, V0y, V0z);
            to_voxel(qx + dsx, qy + dsy, qz + dsz, ex, ey, ez);
            dVx = ex - V0x; dVy = ey - V0y; dVz = ez - V0z;
        }
        // probe for the batch whose first sample has index k0
        auto probe = [&](int k0) -> uint32_t {
            const float km = (float)k0 + 0.5f * (float)(BATCH - 1);
            int ci = (int)(V0x + km * dVx) >> 3, cj = (int)(V0y + km * dVy) >> 3, ck = (int)(V0z + km * dVz) >> 3;
            ci = min(max(ci, 0), P.cnx - 1); cj = min(max(cj, 0), P.cny - 1); ck = min(max(ck, 0), P.cnz - 1);
            const uint32_t cell = mad_u24(mad_u24((uint32_t)ck, (uint32_t)P.cny, (uint32_t)cj), (uint32_t)P.cnx, (uint32_t)ci);
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs_grid, (int)(cell << 1), 0, 0);
        };
        uint32_t cell_next = 0xffffffffu;       // probe result for the next batch to be issued
        int k_issue = 0;                        // index of the first sample of the next batch to be issued
        const uint32_t *tab_x = axis_tab, *tab_y = axis_tab + (ATAB ? P.nx : 0), *tab_z = axis_tab + (ATAB ? P.nx + P.ny : 0);
        // gathers of one batch: BATCH consecutive samples from the current position;
        // returns true when the batch is skipped (positions still advance, bit-exactly)
        // PIPE (true software pipelining): the gathers of the next batch are issued by every live lane, also by the
        // lanes that have no next batch (commit == false: their positions are restored, their loads hit the buffer
        // bounds check or fetch voxels nobody composites).  A branch around the loads makes the number of loads in
        // flight at the following s_waitcnt path-dependent, and the compiler then waits for the NEW batch before it
        // lets the previous one be composited.  With PIPE a wavefront keeps 16 gathers per lane in flight: that
        // shortens the serial chain of an under-filled launch (one GPU's shard of a multi-GPU frame) and only adds L1
        // pressure to a full one, so the host picks it per launch (vr_frame.h: LaunchConfig::pipelined).
        constexpr bool SPEC = PIPE;
        // The batch arrays hold the loaded voxels in their own width: the zero-extension is then an operation of
        // consume(), not of issue() -- as a 32-bit value it would be materialised (and the loads waited for) at
        // the end of the divergent region the gathers are issued in.
        using RawT = typename std::conditional<BIG, uint32_t, VoxelT>::type;
        auto issue = [&](RawT (&v)[BATCH], uint32_t &nib, bool commit) -> bool {   // nib (PK12): bit offset (0 / 4) of sample u in nibble u
            const float Qx0 = Qx, Qy0 = Qy, Qz0 = Qz, qx0 = qx, qy0 = qy, qz0 = qz;
            bool skip = false;
            if (skip_on) {
                skip = (int)cell_next <= P.skip_thresh;
                cell_next = probe(k_issue + BATCH);
                k_issue += BATCH;
            }
            if (skip) {
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    if (POW2) { Qx += dSx; Qy += dSy; Qz += dSz; }
                    else { qx += dsx; qy += dsy; qz += dsz; }
                }
                return true;
            }
            typename VoxelAddr<LAYOUT, BIG>::type off[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                int vi, vj, vk;
                if (POW2) {
                    // voxel units: Q = q*S, U = Q + half*S = texcoord*S before the flips
                    const float ux = Qx + Hx, uy = Qy + Hy, uzr = Qz + Hz, uz = Szv - uzr;
                    float fx = ux, fy = uy, fz = uz;
                    if (VIEW == 1) { fy = uzr; fz = uy; }
                    else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
                    vi = (int)fx; vj = (int)fy; vk = (int)fz;
                    Qx += dSx; Qy += dSy; Qz += dSz;
                } else {
                    voxel_of(qx, qy, qz, vi, vj, vk);
                    qx += dsx; qy += dsy; qz += dsz;
                }
                if (ATAB && BIG) off[u] = (typename VoxelAddr<LAYOUT, BIG>::type)((uint64_t)(tab_x[vi] + tab_y[vj]) + ((uint64_t)tab_z[vk] << 4));   // elements
                else if (ATAB) off[u] = (typename VoxelAddr<LAYOUT, BIG>::type)(tab_x[vi] + tab_y[vj] + tab_z[vk]);   // bytes
                else off[u] = VoxelAddr<LAYOUT, BIG>::at(P, vi, vj, vk);
                if (PK12) nib = u == 0 ? ((uint32_t)vi & 1u) << 2 : ((((uint32_t)vi & 1u) << (4 * u + 2)) | nib);   // odd x: upper 12 of the 16 bits
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (ATAB && !BIG) {
                    v[u] = sizeof(VoxelT) == 1 ? (RawT)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off[u], 0, 0)
                                               : (RawT)__builtin_amdgcn_raw_buffer_load_b16(PK12 ? rs12 : rs, (int)off[u], 0, 0);
                    continue;
                }
                v[u] = (RawT)VoxelFetch<VoxelT, BIG>::load(vol, rs, off[u]);
            }
            if (SPEC && !commit) {
                if (POW2) { Qx = Qx0; Qy = Qy0; Qz = Qz0; }
                else { qx = qx0; qy = qy0; qz = qz0; }
            }
            return false;
        };
        int i = 0;
        // front-to-back compositing of one batch; returns true when the ray terminated.
        // alpha_scale in [0,1] (fast-path precondition) makes dest.a non-decreasing and <= 1,
        // so "dest.a < 0.95 before the LAST sample" proves the shader's per-sample test
        // `dest.a >= 0.95 -> break` (VolumeRenderer.cs:118) passed for the whole batch; only
        // the batch in which a ray terminates is replayed with the literal per-sample tests.
        // one sample onto the destination (front-to-back composite, or MIP's running maximum)
        auto accumulate = [&](float c, float cg, float cb, float a) {
            if (MODE == 1) {
                if (da < a) da = a;                          // dest = src when dest.a < src.a (:165-168)
            } else if (MODE == 3) {
                if (da < a) { drgb = c; dg = cg; db = cb; da = a; }
            } else {
                const float om = 1.0f - da;
                drgb += c * om;
                if (MODE == 2) { dg += cg * om; db += cb * om; }
                da += a * om;
            }
        };
        // PK12: the 12 bits of sample u start at bit 0 or 4 of the 16 loaded (nibble u of nib)
        auto texel_of = [&](RawT raw, uint32_t nib, int u) -> uint32_t {
            if (!PK12) return (uint32_t)raw;
            const uint32_t sh = __builtin_amdgcn_ubfe(nib, 4 * u, 4);
            __builtin_assume(sh <= 4u);
            return ((uint32_t)raw >> sh) & 0xfffu;
        };
        auto consume_live = [&](const RawT (&v)[BATCH], bool skipped, uint32_t nib) -> bool {
            if (skipped) { i += BATCH; return false; }   // every sample of the batch adds exactly zero
            float c[BATCH], cg[BATCH], cbl[BATCH], a[BATCH];
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            float da_last = 0.0f;
            // table look-ups in two halves: half the live registers, the second half's LDS
            // latency hides behind the first half's dependent compositing chain
            constexpr int HALF = BATCH / 2;
#pragma unroll
            for (int u = 0; u < HALF; u++) classify(texel_of(v[u], nib, u), c[u], cg[u], cbl[u], a[u]);
#pragma unroll
            for (int u = HALF; u < BATCH; u++) classify(texel_of(v[u], nib, u), c[u], cg[u], cbl[u], a[u]);
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (u == BATCH - 1) da_last = da;
                accumulate(c[u], cg[u], cbl[u], a[u]);
            }
            if (da_last < 0.95f) { i += BATCH; return false; }
            drgb = drgb0; dg = dg0; db = db0; da = da0;
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (da >= 0.95f) return true;
                accumulate(c[u], cg[u], cbl[u], a[u]);
                i++;
            }
            return da >= 0.95f;
        };
        // PIPE: every lane of the wavefront runs the batch (no branch between the gathers and their use, see
        // issue()); the lanes that are not `live` get their state back.
        auto consume = [&](const RawT (&v)[BATCH], bool skipped, uint32_t nib, bool live) -> bool {
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            const int i0 = i;
            bool term = consume_live(v, skipped, nib);
            if (SPEC && !live) { drgb = drgb0; dg = dg0; db = db0; da = da0; i = i0; term = false; }
            return term;
        };

        bool done = false;
        // one iteration of the shader's loop at position (x, y, z), literally (dest.a > 0.99 of :134 is implied by the
        // dest.a >= 0.95 test of the next iteration and changes nothing); true = the loop ends here
        auto checked_step = [&](float &x, float &y, float &z, float stx, float sty, float stz) -> bool {
            const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) return true;
            const int vi = min((int)(tcx * P.fdim[0]), P.nx - 1);
            const int vj = min((int)(tcy * P.fdim[1]), P.ny - 1);
            const int vk = min((int)(tcz * P.fdim[2]), P.nz - 1);
            float c, cg = 0.0f, cb = 0.0f, a;
            classify(VoxelFetch<VoxelT, BIG>::load(vol, rs, VoxelAddr<LAYOUT, BIG>::at(P, vi, vj, vk)) - (uint32_t)pkb, c, cg, cb, a);
            accumulate(c, cg, cb, a);
            x += stx; y += sty; z += stz;
            return false;
        };
        // ---- checked head: samples 0 .. head - 1
        if (head > 0) {
            float x = hqx, y = hqy, z = hqz;
            for (int h = 0; h < head && !done; h++) {
                if (i >= P.max_steps || checked_step(x, y, z, dsx, dsy, dsz)) done = true;     // the ray ends inside its head
                else i++;
            }
        }
        // ---- safe prefix: software-pipelined, the next batch's gathers are in flight
        //      while the current batch is composited
        const int nb = done ? 0 : k_safe / BATCH;
        // The 8 wavefronts of a workgroup advance in lockstep (one barrier per two batches):
        // their rays cross the same voxel rows / bricks at the same time, so a cache line
        // fetched for one wavefront is still in the CU's L1 when its neighbours need it.
        {
            RawT va[BATCH], vb[BATCH];
            uint32_t nib_a = 0, nib_b = 0;
            bool skip_a = false, skip_b = false;
            int b = 0;
            bool fin = nb == 0;
            if (!fin) {
                if (skip_on) cell_next = probe(0);
                skip_a = issue(va, nib_a, true);
            }
            // lockstep: one plain barrier per 16 samples; every 4th doubles as the vote "all rays finished"
            // (__syncthreads_and is three barriers and a cross-lane reduction: 0.464 -> 0.458 ms on cfg3)
            // (the loop is rotated by hand -- barrier / vote at the bottom -- so that the loop-carried values leave an iteration in
            // the registers its body wrote them to: with the test at the top the register allocator copied six of them to other
            // registers at the loop header and back after the barrier, every iteration: 17 v_mov fewer per 16 samples, 0.4647 ->
            // 0.462 ms on cfg3, bit-identical)
            if (!__syncthreads_and(fin ? 1 : 0)) {
                unsigned it = 0;
                bool stop = false;
                do {
                    if (SPEC) {
                        // straight-line body: the compiler's s_waitcnt sees exactly eight gathers issued behind the
                        // eight it is about to use, on every path (ISA: vmcnt(15) ... vmcnt(8))
                        if (__builtin_amdgcn_ballot_w64(!fin) != 0ull) {                  // (0: nothing left in this wavefront)
                            bool live = !fin;
                            skip_b = issue(vb, nib_b, live && b + 1 < nb);
                            if (consume(va, skip_a, nib_a, live)) { done = true; fin = true; }
                            else if (live && ++b >= nb) fin = true;
                            live = !fin;
                            skip_a = issue(va, nib_a, live && b + 1 < nb);
                            if (consume(vb, skip_b, nib_b, live)) { done = true; fin = true; }
                            else if (live && ++b >= nb) fin = true;
                        }
                    } else {
                        if (!fin) {
                            if (b + 1 < nb) skip_b = issue(vb, nib_b, true);
                            if (consume(va, skip_a, nib_a, true)) { done = true; fin = true; }
                            else if (++b >= nb) fin = true;
                        }
                        if (!fin) {
                            if (b + 1 < nb) skip_a = issue(va, nib_a, true);
                            if (consume(vb, skip_b, nib_b, true)) { done = true; fin = true; }
                            else if (++b >= nb) fin = true;
                        }
                    }
                    it++;
                    if ((it & 3u) == 0u) stop = __syncthreads_and(fin ? 1 : 0) != 0;
                    else __syncthreads();
                } while (!stop);
            }
        }
        // back to box units for the tail (exact: S is a power of two); the step is re-derived
        // from its scaled copy so that only one of the two is live across the batch loop
        float tsx = dsx, tsy = dsy, tsz = dsz;
        if (POW2) {
            qx = Qx / Sx; qy = Qy / Sy; qz = Qz / Sz;
            tsx = dSx / Sx; tsy = dSy / Sy; tsz = dSz / Sz;
        }
        // ---- checked tail: the shader's loop
        if (hit && !done) {
            for (; i < P.max_steps; i++)
                if (checked_step(qx, qy, qz, tsx, tsy, tsz)) break;
        }
        fetches = (uint32_t)i;
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MODE >= 2) store_pixel(P, fb, pix, drgb, dg, db, da);
    else if (MODE == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    if (spp) spp[pix] = fetches;
}

// ------------------------------------------------------------------ trilinear kernel
// TRILINEAR (GL's linear rule, fixed lerp order x, y, z -- oracle/vr_oracle.c) with the fast
// kernel's skeleton: 8x8 pixels per wavefront, 512-thread workgroups in lockstep, longest-first
// tile table, safe prefix without the six bound tests, checked tail, per-axis address tables in
// LDS (X[i0], X[i1], Y[j0], ... : a neighbour across a brick boundary costs nothing extra).
// Two samples (16 taps) are gathered per batch and software-pipelined.  Grey ramp, composite
// (MIPM = 0) or MIP (MIPM = 1), iterative accumulation, alpha_scale in [0,1], 32-bit offsets.
// Every sample goes through the generic kernel's operations in the generic kernel's order.
constexpr int TRI_BATCH = 2;

// APRON: `vol` is the apron copy (vr_device.h: build_axis_tables_apron) -- x neighbours are always one load.
template <typename VoxelT, int LAYOUT, int DIVTC, int VIEW, bool POW2, int MIPM, bool APRON = false>
__global__ __launch_bounds__(512) void raymarch_tri_kernel(const FrameParams P, const VoxelT *__restrict__ vol,
                                                           const uint32_t vol_bytes, float4 *__restrict__ fb,
                                                           uint32_t *__restrict__ spp,
                                                           const uint32_t *__restrict__ tile_table)
{
    __shared__ uint32_t axis_tab[FAST_AXIS_TAB_MAX];
    const uint32_t t = tile_table[blockIdx.x];
    if (t == 0xffffffffu) return;                            // padding block
    const unsigned tx = t & 0xffffu, ty = t >> 16;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(tx * FAST_TILE_W + (wave & 3u) * 8u + (lane & 7u));
    const int ly = (int)(ty * FAST_TILE_H + (wave >> 2) * 8u + (lane >> 3));
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int st = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (st * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    const bool in_image = !(px >= P.col_lim || py >= P.row_lim || py >= P.row_end);
    Ray ray = {};
    float t_min = 0.0f, t_max = 0.0f;
    bool hit = false;
    if (in_image) {
        ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
        hit = intersect_ray_aabb(P, ray, t_min, t_max);
    }
    if (__syncthreads_or(hit ? 1 : 0)) {
        if (APRON) build_axis_tables_apron<VoxelT>(P, axis_tab, 512);
        else build_axis_tables<VoxelT, LAYOUT, false>(P, axis_tab, 512);
        __syncthreads();
    }
    const uint32_t *tab_x = axis_tab, *tab_y = axis_tab + P.nx, *tab_z = axis_tab + P.nx + P.ny;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, (int)pair_load_extent(vol_bytes), 0x00020000);

    const float EPSILON = 0.000001f;
    const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
    float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
    const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
    // checked head (vr_device.h: head_steps): positions stepped here, sampled through the checked loop below
    const float hqx = qx, hqy = qy, hqz = qz;
    const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
    for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
    const int k_safe = hit ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;
    const int nxm1 = P.nx - 1, nym1 = P.ny - 1, nzm1 = P.nz - 1;

    // texcoord * dim of a position (before the -0.5 of the linear filter)
    const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
    float Qx = qx * Sx, Qy = qy * Sy, Qz = qz * Sz;          // POW2: voxel-unit marching (see the fast kernel)
    const float dSx = dsx * Sx, dSy = dsy * Sy, dSz = dsz * Sz;
    const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;
    // (in VECTOR registers for the sample loop: a VALU instruction with an SGPR operand issues at the slow rate on gfx950, 4.2
    // cycles instead of 2.3 -- fp32 and integer alike, tools/ubench/valu_rates2.hip)
    float Szv = Sz, Syv = Sy;
    asm volatile("" : "+v"(Szv), "+v"(Syv));
    auto scaled_texcoord = [&](float ax, float ay, float az, float &fx, float &fy, float &fz) {
        const float ux = div_mode<DIVTC>(ax + P.half[0], P.ext[0], P.rext[0]);
        const float uy = div_mode<DIVTC>(ay + P.half[1], P.ext[1], P.rext[1]);
        const float uzr = div_mode<DIVTC>(az + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
        const float uz = 1.0f - uzr;
        float tcx = ux, tcy = uy, tcz = uz;
        if (VIEW == 1) { tcy = uzr; tcz = uy; }
        else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
        fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
    };
    // tap offsets (bytes) and the 3 lerp weights of the sample at scaled texcoord f: off[0..3] = the
    // x0 taps of the four (y, z) corners, off[4..7] = their x1 neighbours; `pair` = x1 is the next
    // storage element of x0 (same brick row / volume row), so one load fetches both
    auto taps_of = [&](float fx, float fy, float fz, uint32_t (&off)[8], float &ax, float &ay, float &az, bool &pair) {
        const float u = fx - 0.5f, v = fy - 0.5f, w = fz - 0.5f;
        const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
        ax = u - fu; ay = v - fv; az = w - fw;
        const int iu = (int)fu, iv = (int)fv, iw = (int)fw;
        const int i0 = med3_i32(iu, 0, nxm1), i1 = med3_i32(iu + 1, 0, nxm1);
        const int j0 = med3_i32(iv, 0, nym1), j1 = med3_i32(iv + 1, 0, nym1);
        const int k0 = med3_i32(iw, 0, nzm1), k1 = med3_i32(iw + 1, 0, nzm1);
        const uint32_t x0 = tab_x[i0], x1 = tab_x[i1], y0 = tab_y[j0], y1 = tab_y[j1], z0 = tab_z[k0], z1 = tab_z[k1];
        pair = APRON ? i1 != i0 : x1 == x0 + (uint32_t)sizeof(VoxelT);      // APRON: only the clamped taps at the volume's faces are not a pair
        off[0] = x0 + y0 + z0; off[1] = x0 + y1 + z0; off[2] = x0 + y0 + z1; off[3] = x0 + y1 + z1;
        off[4] = x1 + y0 + z0; off[5] = x1 + y1 + z0; off[6] = x1 + y0 + z1; off[7] = x1 + y1 + z1;
    };
    // tv[0..3]: x0 tap in the low half, next storage element in the high half; tv[4..7]: the x1
    // taps of the lanes whose x1 lies elsewhere (brick / volume edge)
    // reuse: 0 = fetch both z planes of the cell, 1 = this sample's near plane is the previous sample's far
    // plane (the ray moved one voxel on in z inside the same (x, y) cell), 2 = same cell as the previous sample.
    // Consecutive samples of a ray that runs along z share half or all of their taps: the words are taken
    // from the previous sample when they are composited (consume), and their loads are not issued at all.
    auto load_taps = [&](const uint32_t (&off)[8], bool pair, int reuse, uint32_t *tv) {
        auto word = [&](uint32_t o) { return sizeof(VoxelT) == 1 ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)o, 0, 0)
                                                                 : (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)o, 0, 0); };
        if (reuse == 0) { tv[0] = word(off[0]); tv[1] = word(off[1]); }
        if (reuse != 2) { tv[2] = word(off[2]); tv[3] = word(off[3]); }
        if (!pair) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                tv[4 + k] = sizeof(VoxelT) == 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off[4 + k], 0, 0)
                                                : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off[4 + k], 0, 0);
        }
    };
    // interpolation, window and classification of one sample (the generic kernel's operations)
    auto shade = [&](const uint32_t *tv, bool pair, float ax, float ay, float az, float &c, float &a) {
        constexpr uint32_t M = sizeof(VoxelT) == 1 ? 0xffu : 0xffffu;
        constexpr int SH = sizeof(VoxelT) == 1 ? 8 : 16;
        const float c000 = (float)(tv[0] & M), c010 = (float)(tv[1] & M), c001 = (float)(tv[2] & M), c011 = (float)(tv[3] & M);
        const float c100 = (float)(pair ? tv[0] >> SH : tv[4]), c110 = (float)(pair ? tv[1] >> SH : tv[5]);
        const float c101 = (float)(pair ? tv[2] >> SH : tv[6]), c111 = (float)(pair ? tv[3] >> SH : tv[7]);
        const float c00 = tri_lerp(c000, c100, ax), c10 = tri_lerp(c010, c110, ax);
        const float c01 = tri_lerp(c001, c101, ax), c11 = tri_lerp(c011, c111, ax);
        const float c0 = tri_lerp(c00, c10, ay), c1 = tri_lerp(c01, c11, ay);
        float s = tri_lerp(c0, c1, az);
        s = fminf(fmaxf(s, P.fmin), P.fmax);                 // never NaN here
        s = div_cert(s - P.fmin, P.fden, P.rden);
        a = s * P.alpha_scale;
        c = s * a;
    };
    float drgb = 0.0f, da = 0.0f;
    auto accumulate = [&](float c, float a) {
        if (MIPM == 1) {
            if (da < a) da = a;
        } else {
            const float om = 1.0f - da;
            drgb += c * om;
            da += a * om;
        }
    };
    int i = 0;
    uint32_t poff0 = 0xffffffffu, poff1 = 0xffffffffu, poff2 = 0xffffffffu, poff3 = 0xffffffffu;   // pair-word offsets of the last sample issued
    bool ppair = false;
    uint32_t E0 = 0, E1 = 0, E2 = 0, E3 = 0;                 // pair words of the last sample composited
    // gathers of one batch (TRI_BATCH consecutive samples); positions advance with the shader's additions
    auto issue = [&](uint32_t (&tv)[TRI_BATCH * 8], float (&wt)[TRI_BATCH * 3], bool (&pr)[TRI_BATCH], int (&ru)[TRI_BATCH]) {
        uint32_t off[TRI_BATCH][8];
#pragma unroll
        for (int u = 0; u < TRI_BATCH; u++) {
            float fx, fy, fz;
            if (POW2) {
                const float ux = Qx + Hx, uy = Qy + Hy, uzr = Qz + Hz, uz = Szv - uzr;
                fx = ux; fy = uy; fz = uz;
                if (VIEW == 1) { fy = uzr; fz = uy; }
                else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
                Qx += dSx; Qy += dSy; Qz += dSz;
            } else {
                scaled_texcoord(qx, qy, qz, fx, fy, fz);
                qx += dsx; qy += dsy; qz += dsz;
            }
            taps_of(fx, fy, fz, off[u], wt[3 * u + 0], wt[3 * u + 1], wt[3 * u + 2], pr[u]);
            const bool can = pr[u] && ppair;
            const bool near_is_far = off[u][0] == poff2 && off[u][1] == poff3, near_is_near = off[u][0] == poff0 && off[u][1] == poff1;
            const bool far_is_far = off[u][2] == poff2 && off[u][3] == poff3;
            ru[u] = can ? ((near_is_near && far_is_far) ? 2 : (near_is_far ? 1 : 0)) : 0;
            poff0 = off[u][0]; poff1 = off[u][1]; poff2 = off[u][2]; poff3 = off[u][3]; ppair = pr[u];
        }
#pragma unroll
        for (int u = 0; u < TRI_BATCH; u++) load_taps(off[u], pr[u], ru[u], &tv[8 * u]);
    };
    // returns true when the ray terminated (see the fast kernel: batch early-termination)
    auto consume = [&](const uint32_t (&tv)[TRI_BATCH * 8], const float (&wt)[TRI_BATCH * 3], const bool (&pr)[TRI_BATCH], const int (&ru)[TRI_BATCH]) -> bool {
        float c[TRI_BATCH], a[TRI_BATCH];
#pragma unroll
        for (int u = 0; u < TRI_BATCH; u++) {
            uint32_t w[8];
            w[0] = ru[u] == 0 ? tv[8 * u + 0] : (ru[u] == 1 ? E2 : E0); w[1] = ru[u] == 0 ? tv[8 * u + 1] : (ru[u] == 1 ? E3 : E1);
            w[2] = ru[u] == 2 ? E2 : tv[8 * u + 2]; w[3] = ru[u] == 2 ? E3 : tv[8 * u + 3];
            w[4] = tv[8 * u + 4]; w[5] = tv[8 * u + 5]; w[6] = tv[8 * u + 6]; w[7] = tv[8 * u + 7];
            E0 = w[0]; E1 = w[1]; E2 = w[2]; E3 = w[3];
            shade(w, pr[u], wt[3 * u + 0], wt[3 * u + 1], wt[3 * u + 2], c[u], a[u]);
        }
        const float drgb0 = drgb, da0 = da;
        float da_last = 0.0f;
#pragma unroll
        for (int u = 0; u < TRI_BATCH; u++) {
            if (u == TRI_BATCH - 1) da_last = da;
            accumulate(c[u], a[u]);
        }
        if (da_last < 0.95f) { i += TRI_BATCH; return false; }
        drgb = drgb0; da = da0;
#pragma unroll
        for (int u = 0; u < TRI_BATCH; u++) {
            if (da >= 0.95f) return true;
            accumulate(c[u], a[u]);
            i++;
        }
        return da >= 0.95f;
    };

    bool done = false;
    // one iteration of the shader's loop at (x, y, z), literally; true = the loop ends here
    auto checked_step = [&](float &x, float &y, float &z, float stx, float sty, float stz) -> bool {
        const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
        const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
        const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
        const float uz = 1.0f - uzr;
        float tcx = ux, tcy = uy, tcz = uz;
        if (VIEW == 1) { tcy = uzr; tcz = uy; }
        else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
        if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) return true;
        uint32_t off[8], tv[8];
        float ax, ay, az, c, a;
        bool pair;
        taps_of(tcx * P.fdim[0], tcy * P.fdim[1], tcz * P.fdim[2], off, ax, ay, az, pair);
        load_taps(off, pair, 0, tv);
        shade(tv, pair, ax, ay, az, c, a);
        accumulate(c, a);
        x += stx; y += sty; z += stz;
        return false;
    };
    if (head > 0) {                                          // checked head: samples 0 .. head - 1
        float x = hqx, y = hqy, z = hqz;
        for (int h = 0; h < head && !done; h++) {
            if (i >= P.max_steps || checked_step(x, y, z, dsx, dsy, dsz)) done = true;
            else i++;
        }
    }
    const int nb = done ? 0 : k_safe / TRI_BATCH;
    {
        uint32_t va[TRI_BATCH * 8], vb[TRI_BATCH * 8];
        float wa[TRI_BATCH * 3], wb[TRI_BATCH * 3];
        bool pa[TRI_BATCH], pb[TRI_BATCH];
        int ra[TRI_BATCH], rb[TRI_BATCH];
        int b = 0;
        bool fin = nb == 0;
        if (!fin) issue(va, wa, pa, ra);
        for (unsigned it = 0;; it++) {
            // lockstep: one plain barrier per 16 samples, every 4th doubles as the vote "all rays finished" (as in the fast kernel)
            if ((it & 3u) == 0u) { if (__syncthreads_and(fin ? 1 : 0)) break; }
            else __syncthreads();
#pragma unroll 1
            for (int rep = 0; rep < 4 && !fin; rep++) {      // 16 samples between two barriers
                if (b + 1 < nb) issue(vb, wb, pb, rb);
                if (consume(va, wa, pa, ra)) { done = true; fin = true; break; }
                if (++b >= nb) { fin = true; break; }
                if (b + 1 < nb) issue(va, wa, pa, ra);
                if (consume(vb, wb, pb, rb)) { done = true; fin = true; break; }
                if (++b >= nb) { fin = true; break; }
            }
        }
    }
    float tsx = dsx, tsy = dsy, tsz = dsz;
    if (POW2) {
        qx = Qx / Sx; qy = Qy / Sy; qz = Qz / Sz;           // exact: S is a power of two
        tsx = dSx / Sx; tsy = dSy / Sy; tsz = dSz / Sz;
    }
    // ---- checked tail: the shader's loop
    if (hit && !done) {
        for (; i < P.max_steps; i++)
            if (checked_step(qx, qy, qz, tsx, tsy, tsz)) break;
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MIPM == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    if (spp) spp[pix] = (uint32_t)i;
}

// ------------------------------------------------------------------ relay kernel
// A ray is a serial chain of ~1000 dependent samples; issued by ONE wavefront it advances at
// ~300 cycles per sample, so a launch that cannot fill the chip's 8192 wave slots (one GPU's
// shard of a multi-GPU frame) costs >= 0.16 ms however idle the chip is, and even the full
// frame is only ~1.3 rounds of such chains.  Here FOUR wavefronts march one 8x8-pixel tile
// as a relay: wavefront w owns batches w, w+4, w+8, ...; for its batch it generates the
// addresses, gathers and classifies on its own, and only the front-to-back compositing
// recurrence is handed from wavefront to wavefront through LDS (state = dest colour, dest
// alpha, sample count per ray; one sequence word published with release/acquire at
// workgroup scope).  Every sample still goes through exactly the shader's operations in
// the shader's order; only WHICH wavefront executes them changes.
// Round 6: the ray POSITIONS no longer travel.  Rounds 1-5 handed the position at the start of
// batch n + 1 from the wavefront that had generated batch n's addresses to the next one through
// LDS -- a second serial chain of ~130 hand-overs beside the compositing one.  The positions do
// not depend on the data, so every wavefront now iterates the shader's additions itself, in the
// shader's order (32 per axis between its own batches n and n + 4: the four wavefronts perform
// each addition four times instead of once, in issue slots a sparse launch leaves idle).  One
// rank's shard of the cfg3 frame at N = 8: 0.106 -> 0.084 ms, N = 4: 0.182 -> 0.153 (now ahead of
// the fast kernel's 0.163), bit-identical frames (profiles/r06_relay_own_positions.txt).
// Same preconditions as the fast kernel's headline shape (NEAREST, grey-ramp composite,
// iterative accumulation, default view, 32-bit offsets, alpha_scale in [0,1]).
// measured on cfg3: 2 or 8 wavefronts per tile, batches of 16, 4 tiles per workgroup are all slower
// (round 6, with the wavefronts' own positions: six or eight wavefronts per tile, one tile per workgroup: N = 8 shard 0.150 / 0.148 ms
// against 0.087 -- profiles/r06_relay_own_positions.txt)
#ifndef VR_RELAY_WAVES
#define VR_RELAY_WAVES 4
#define VR_RELAY_BATCH 8
#define VR_RELAY_TILES 2
#endif
constexpr int RELAY_WAVES = VR_RELAY_WAVES, RELAY_BATCH = VR_RELAY_BATCH;
// two tiles share one workgroup (and one 32 KiB classification table): 4 workgroups = 8 tiles
// = 32 wavefronts per CU, the wave-slot limit, instead of 4 tiles per CU
constexpr int RELAY_TILES = VR_RELAY_TILES, RELAY_THREADS = 64 * RELAY_WAVES * RELAY_TILES;

struct RelayState {
    float rgb[2][64];      // red (== green == blue in the grey modes)
    float g[2][64], b[2][64];   // green / blue: only the transfer-function modes carry them
    float a[2][64];
    int i[2][64];
    unsigned seq;          // number of batches composited so far; bit 31: no ray of the tile needs another batch (one word: one LDS load per poll)
    unsigned final_n;      // the state slot holding the result is final_n & 1
};
constexpr unsigned RELAY_STOP = 0x80000000u;

// VIEW / MODE as in the fast kernel (round 2: every mode and view of a sparse shard gets the relay)
template <typename VoxelT, int LAYOUT, int DIVTC, bool LUT, bool POW2, bool NOCLAMP, bool ATAB, bool PK12, int VIEW, int MODE>
__global__ __launch_bounds__(RELAY_THREADS) void raymarch_relay_kernel(const FrameParams P,
                                                             const VoxelT *__restrict__ vol,
                                                             const float4 *__restrict__ tf,
                                                             const uint32_t vol_bytes,
                                                             float4 *__restrict__ fb,
                                                             uint32_t *__restrict__ spp,
                                                             const uint32_t *__restrict__ tile_table,
                                                             const void *__restrict__ packed12,
                                                             const uint32_t packed12_bytes)
{
    static_assert(!PK12 || (ATAB && sizeof(VoxelT) == 2 && LAYOUT == 1), "12-bit copy: u16 bricks through the address tables");
    __shared__ float lut[LUT ? FAST_LUT_MAX * 2 : 4];
    __shared__ uint32_t axis_tab[ATAB ? FAST_AXIS_TAB_MAX : 1];
    __shared__ RelayState rs_all[RELAY_TILES];
    // block b -> (32x16 tile of the longest-first table, pair of 8x8 sub-tiles); the 4 pairs of
    // a tile are consecutive blocks of ONE XCD (b & 7 is the XCD)
    const unsigned b = blockIdx.x;
    constexpr unsigned GROUPS = 8u / RELAY_TILES;                        // workgroups per 32x16-pixel tile
    const uint32_t tile = tile_table[(b / (8u * GROUPS)) * 8u + (b & 7u)];
    if (tile == 0xffffffffu) return;
    const unsigned tl = threadIdx.x / (64u * RELAY_WAVES);              // tile of this wavefront within the workgroup
    RelayState &rs = rs_all[tl];
    const unsigned sub = ((b >> 3) % GROUPS) * RELAY_TILES + tl;
    const unsigned lane = threadIdx.x & 63u, w = (threadIdx.x >> 6) % RELAY_WAVES;
    const int lx = (int)((tile & 0xffffu) * FAST_TILE_W + (sub & 3u) * 8u + (lane & 7u));
    const int ly = (int)((tile >> 16) * FAST_TILE_H + (sub >> 2) * 8u + (lane >> 3));
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    const bool in_image = !(px >= P.col_lim || py >= P.row_lim || py >= P.row_end);

    Ray ray = {};
    float t_min = 0.0f, t_max = 0.0f;
    bool hit = false;
    if (in_image) {
        ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
        hit = intersect_ray_aabb(P, ray, t_min, t_max);
    }
    const int any_hit = __syncthreads_or(hit ? 1 : 0);
    static_assert(MODE < 2 || LUT, "the transfer-function modes classify through the table");
    if (LUT && any_hit) {
        const int n = P.max_val - P.min_val + 1;
        for (int e = (int)threadIdx.x; e < n; e += RELAY_THREADS) {
            const float s = (float)(P.min_val + e);
            const float v = div_cert(s - P.fmin, P.fden, P.rden);
            if (MODE >= 2) {                                     // two-level table, as in the fast kernel
                int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                idx = clampi(idx, 0, P.tf_len - 1);
                reinterpret_cast<uint8_t *>(lut)[FAST_TF_ENTRIES * 16 + e] = (uint8_t)idx;
            } else if (MODE == 0 && P.tf_grey != 0) {            // grey transfer function on the grey-ramp instance (see the fast kernel)
                int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                idx = clampi(idx, 0, P.tf_len - 1);
                const float4 t = tf[idx];
                const float a = t.w * P.alpha_scale;
                lut[2 * e + 0] = t.x * a; lut[2 * e + 1] = a;
            } else {
                const float a = v * P.alpha_scale;
                lut[2 * e + 0] = v * a; lut[2 * e + 1] = a;
            }
        }
        if (MODE >= 2) {
            for (int e = (int)threadIdx.x; e < P.tf_len; e += RELAY_THREADS) {
                const float4 t = tf[e];
                const float a = t.w * P.alpha_scale;
                if (MODE == 3) { lut[4 * e + 0] = t.x * P.alpha_scale; lut[4 * e + 1] = t.y * P.alpha_scale; lut[4 * e + 2] = t.z * P.alpha_scale; }
                else { lut[4 * e + 0] = t.x * a; lut[4 * e + 1] = t.y * a; lut[4 * e + 2] = t.z * a; }
                lut[4 * e + 3] = a;
            }
        }
    }
    if (ATAB && any_hit) build_axis_tables<VoxelT, LAYOUT, PK12>(P, axis_tab, RELAY_THREADS);
    if (w == 0) { rs.rgb[0][lane] = 0.0f; rs.g[0][lane] = 0.0f; rs.b[0][lane] = 0.0f; rs.a[0][lane] = 0.0f; rs.i[0][lane] = 0; }
    if (w == 0 && lane == 0) { rs.seq = 0u; rs.final_n = 0u; }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, (int)vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs12 = __builtin_amdgcn_make_buffer_rsrc((void *)packed12, 0, (int)packed12_bytes, 0x00020000);
    const uint32_t *tab_x = axis_tab, *tab_y = axis_tab + (ATAB ? P.nx : 0), *tab_z = axis_tab + (ATAB ? P.nx + P.ny : 0);
    const float EPSILON = 0.000001f;
    const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
    float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
    const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
    // checked head (vr_device.h: head_steps): every wavefront steps the positions identically; wavefront 0 samples them
    // through the checked loop before its first batch (it is the one that reads state slot 0)
    const float hqx = qx, hqy = qy, hqz = qz;
    const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
    for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
    // A ray can END inside its head: a head sample fails the shader's bounds test (o + d*t_min rounds to a point
    // outside the box when the eye is far away) or the step budget runs out.  That part of the head does not depend
    // on the voxel data, so every wavefront of the relay evaluates it for itself: such a ray has no prefix batches
    // and no tail (wavefront 0's literal head loop below stops at the same sample).
    bool head_ended = false;
    {
        float x = hqx, y = hqy, z = hqz;
        for (int h = 0; h < head && !head_ended; h++) {
            const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            if (h >= P.max_steps || tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f) head_ended = true;
            x += dsx; y += dsy; z += dsz;
        }
    }
    const int k_safe = (hit && !head_ended) ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;
    const int nb = k_safe / RELAY_BATCH;                 // batches of THIS ray
    int nbmax = nb;                                      // batches of the tile (same in all 4 wavefronts)
    for (int o = 32; o > 0; o >>= 1) nbmax = max(nbmax, __shfl_xor(nbmax, o));

    // marching units: voxels for POW2 (see the fast kernel), box units otherwise.  Only the
    // step is kept in registers; the position itself lives in LDS between wavefronts.
    // (per box axis the scale is the dimension of the voxel axis it maps to: y and z swap in the rotated views)
    const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
    const float mx = POW2 ? dsx * Sx : dsx, my = POW2 ? dsy * Sy : dsy, mz = POW2 ? dsz * Sz : dsz;
    const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;
    // (in VECTOR registers for the sample loop: a VALU instruction with an SGPR operand issues at the slow rate on gfx950, 4.2
    // cycles instead of 2.3 -- fp32 and integer alike, tools/ubench/valu_rates2.hip)
    float Szv = Sz, Syv = Sy;
    asm volatile("" : "+v"(Szv), "+v"(Syv));
    const int pkb = PK12 ? P.pk12_base : 0;      // see the fast kernel: classify() works on packed values (voxel - base)
    const int wmin = P.min_val - pkb, wmax = P.max_val - pkb;
    const int lut_bias = MODE >= 2 ? FAST_TF_ENTRIES * 16 - wmin : -8 * wmin;

    // waits until `n` batches have been composited; gives up when the tile has been stopped (early ray termination
    // of every ray, or the last batch): returns false, nothing may be touched any more
    auto wait_seq = [&](int n) -> bool {
        for (;;) {
            const unsigned v = __hip_atomic_load(&rs.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v & RELAY_STOP) return false;
            if ((v & ~RELAY_STOP) >= (unsigned)n) return true;
            // (no s_sleep between the polls: a relay launch leaves the SIMDs' issue slots idle anyway, and the 64-cycle naps sat on
            // the compositing chain -- N = 8 shard 0.087 -> 0.082 ms, N = 4 0.155 -> 0.152, N = 2 0.286 -> 0.278)
        }
    };
    bool stopped = false;
    // this wavefront's own copy of the ray position, at the start of batch own_n: the shader's additions (one fma with the
    // multiplicand 1.0 IS the addition; 0.0 keeps the position of a ray whose prefix has ended), every one of them, in order
    float ox = POW2 ? qx * Sx : qx, oy = POW2 ? qy * Sy : qy, oz = POW2 ? qz * Sz : qz;
    int own_n = 0;
    auto own_advance_to = [&](int n) {
        for (; own_n < n; own_n++) {
            const float vf = own_n < nb ? 1.0f : 0.0f;
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) { ox = __builtin_fmaf(mx, vf, ox); oy = __builtin_fmaf(my, vf, oy); oz = __builtin_fmaf(mz, vf, oz); }
        }
    };
    // gathers of batch n (if this ray still needs them); returns whether v[] is valid
    auto issue = [&](int n, uint32_t (&v)[RELAY_BATCH], uint32_t &nib, float da_seen) -> bool {
        if (stopped || (__hip_atomic_load(&rs.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & RELAY_STOP)) { stopped = true; return false; }
        own_advance_to(n);
        float x = ox, y = oy, z = oz;
        const bool need = n < nb && da_seen < 0.95f;
        uint32_t off[RELAY_BATCH];
        if (n < nb) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                int vi, vj, vk;
                if (POW2) {
                    const float ux = x + Hx, uy = y + Hy, uzr = z + Hz, uz = Szv - uzr;
                    float fx = ux, fy = uy, fz = uz;
                    if (VIEW == 1) { fy = uzr; fz = uy; }
                    else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
                    vi = (int)fx; vj = (int)fy; vk = (int)fz;
                } else {
                    const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
                    const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
                    const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
                    const float uz = 1.0f - uzr;
                    float tcx = ux, tcy = uy, tcz = uz;
                    if (VIEW == 1) { tcy = uzr; tcz = uy; }
                    else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
                    vi = (int)(tcx * P.fdim[0]); vj = (int)(tcy * P.fdim[1]); vk = (int)(tcz * P.fdim[2]);
                }
                if (ATAB) off[u] = tab_x[vi] + tab_y[vj] + tab_z[vk];                    // bytes
                else off[u] = VoxelAddr<LAYOUT, false>::at(P, vi, vj, vk);
                if (PK12) nib = u == 0 ? ((uint32_t)vi & 1u) << 2 : ((((uint32_t)vi & 1u) << (4 * u + 2)) | nib);
                x += mx; y += my; z += mz;
            }
        }
        ox = x; oy = y; oz = z; own_n = n + 1;           // the position at the start of batch n + 1
        if (need) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                if (ATAB)
                    v[u] = sizeof(VoxelT) == 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)off[u], 0, 0)
                                               : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(PK12 ? rs12 : rsrc, (int)off[u], 0, 0);
                else
                    v[u] = VoxelFetch<VoxelT, false>::load(vol, rsrc, off[u]);
            }
        }
        return need;
    };
    auto classify = [&](uint32_t texel, float &c, float &cg, float &cb, float &a) {
        if (LUT) {
            int t = (int)texel;
            if (!NOCLAMP) t = med3_i32(t, wmin, wmax);
            if (MODE >= 2) {
                const uint32_t idx = reinterpret_cast<const uint8_t *>(lut)[(uint32_t)(t + lut_bias)];
                const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
                c = q.x; cg = q.y; cb = q.z; a = q.w;
                return;
            }
            const float2 ca = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(lut) + (uint32_t)((t << 3) + lut_bias));
            c = ca.x; a = ca.y;
        } else {
            float s = (float)(texel + (uint32_t)pkb);
            s = fminf(fmaxf(s, P.fmin), P.fmax);
            s = div_cert(s - P.fmin, P.fden, P.rden);
            a = s * P.alpha_scale;
            c = s * a;
        }
    };
    float da_seen = 0.0f;
    // one sample onto the destination: front-to-back composite, or MIP's running maximum (see the fast kernel)
    auto accumulate = [&](float &drgb, float &dg, float &db, float &da, float c, float cg, float cb, float a) {
        if (MODE == 1) {
            if (da < a) da = a;
        } else if (MODE == 3) {
            if (da < a) { drgb = c; dg = cg; db = cb; da = a; }
        } else {
            const float om = 1.0f - da;
            drgb += c * om;
            if (MODE == 2) { dg += cg * om; db += cb * om; }
            da += a * om;
        }
    };
    // take over the recurrence for batch n, composite, hand it on
    auto relay = [&](int n, const uint32_t (&v)[RELAY_BATCH], uint32_t nib, bool valid) {
        float c[RELAY_BATCH], cg[RELAY_BATCH], cb[RELAY_BATCH], a[RELAY_BATCH];
        if (valid) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                cg[u] = cb[u] = 0.0f;
                classify(PK12 ? __builtin_amdgcn_ubfe(v[u], __builtin_amdgcn_ubfe(nib, 4 * u, 4), 12) : v[u], c[u], cg[u], cb[u], a[u]);
            }
        }
        if (stopped || !wait_seq(n)) { stopped = true; return; }
        const int slot = n & 1;
        float drgb = rs.rgb[slot][lane], da = rs.a[slot][lane], dg = 0.0f, db = 0.0f;
        if (MODE >= 2) { dg = rs.g[slot][lane]; db = rs.b[slot][lane]; }
        int i = rs.i[slot][lane];
        if (valid) {
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            float da_last = 0.0f;
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                if (u == RELAY_BATCH - 1) da_last = da;
                accumulate(drgb, dg, db, da, c[u], cg[u], cb[u], a[u]);
            }
            if (da_last < 0.95f) {
                i += RELAY_BATCH;
            } else {                                     // the batch in which the ray terminates: literal per-sample tests
                drgb = drgb0; dg = dg0; db = db0; da = da0;
#pragma unroll
                for (int u = 0; u < RELAY_BATCH; u++) {
                    if (da >= 0.95f) break;
                    accumulate(drgb, dg, db, da, c[u], cg[u], cb[u], a[u]);
                    i++;
                }
            }
        }
        da_seen = da;
        rs.rgb[slot ^ 1][lane] = drgb; rs.a[slot ^ 1][lane] = da; rs.i[slot ^ 1][lane] = i;
        if (MODE >= 2) { rs.g[slot ^ 1][lane] = dg; rs.b[slot ^ 1][lane] = db; }
        // does any ray of the tile need another batch?  (terminated rays and rays whose prefix
        // ends here do not)
        const bool more = hit && da < 0.95f && n + 1 < nb;
        const bool last = !__any(more ? 1 : 0);
        if (last) {
            if (lane == 0) rs.final_n = (unsigned)(n + 1);
            stopped = true;
        }
        if (lane == 0) __hip_atomic_store(&rs.seq, (unsigned)(n + 1) | (last ? RELAY_STOP : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    {
        if (w == 0 && head > 0) {                            // samples 0 .. head - 1, literally, into state slot 0
            float drgb = 0.0f, dg = 0.0f, db = 0.0f, da = 0.0f, x = hqx, y = hqy, z = hqz;
            int i = 0;
            for (int h = 0; h < head && i < P.max_steps; h++) {
                const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
                const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
                const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
                const float uz = 1.0f - uzr;
                float tcx = ux, tcy = uy, tcz = uz;
                if (VIEW == 1) { tcy = uzr; tcz = uy; }
                else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
                if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) break;
                const int vi = min((int)(tcx * P.fdim[0]), P.nx - 1);
                const int vj = min((int)(tcy * P.fdim[1]), P.ny - 1);
                const int vk = min((int)(tcz * P.fdim[2]), P.nz - 1);
                float c, cg = 0.0f, cb = 0.0f, a;
                classify(VoxelFetch<VoxelT, false>::load(vol, rsrc, VoxelAddr<LAYOUT, false>::at(P, vi, vj, vk)) - (uint32_t)pkb, c, cg, cb, a);
                accumulate(drgb, dg, db, da, c, cg, cb, a);
                x += dsx; y += dsy; z += dsz;
                i++;
            }
            rs.rgb[0][lane] = drgb; rs.a[0][lane] = da; rs.i[0][lane] = i;
            if (MODE >= 2) { rs.g[0][lane] = dg; rs.b[0][lane] = db; }
            da_seen = da;
        }
        uint32_t va[RELAY_BATCH], vb[RELAY_BATCH];
        uint32_t nib_a = 0, nib_b = 0;
        bool ok_a = false, ok_b = false;
        int n = (int)w;
        if (n < nbmax) ok_a = issue(n, va, nib_a, da_seen);
        while (n < nbmax && !stopped) {
            if (n + RELAY_WAVES < nbmax) ok_b = issue(n + RELAY_WAVES, vb, nib_b, da_seen);
            relay(n, va, nib_a, ok_a);
            n += RELAY_WAVES;
            if (n >= nbmax || stopped) break;
            if (n + RELAY_WAVES < nbmax) ok_a = issue(n + RELAY_WAVES, va, nib_a, da_seen);
            relay(n, vb, nib_b, ok_b);
            n += RELAY_WAVES;
        }
    }
    // every wavefront of the workgroup leaves its relay loop exactly once; afterwards the
    // state / position words are quiescent and wavefront 0 of each tile finishes the rays
    __syncthreads();
    if (w != 0) return;
    const unsigned fin = rs.final_n;                     // == nbmax unless the tile stopped early
    float drgb = rs.rgb[fin & 1][lane], da = rs.a[fin & 1][lane], dg = 0.0f, db = 0.0f;
    if (MODE >= 2) { dg = rs.g[fin & 1][lane]; db = rs.b[fin & 1][lane]; }
    int i = rs.i[fin & 1][lane];
    // a ray that still needs its tail finished its prefix at batch nb <= fin, and its position has not changed since
    own_advance_to((int)fin);
    qx = ox; qy = oy; qz = oz;
    if (POW2) { qx = qx / Sx; qy = qy / Sy; qz = qz / Sz; }   // exact: S is a power of two
    const float tsx = POW2 ? mx / Sx : mx, tsy = POW2 ? my / Sy : my, tsz = POW2 ? mz / Sz : mz;
    if (hit && !head_ended) {
        const int nxm1 = P.nx - 1, nym1 = P.ny - 1, nzm1 = P.nz - 1;
        for (; i < P.max_steps; i++) {
            const float ux = div_mode<DIVTC>(qx + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(qy + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(qz + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) break;
            const int vi = min((int)(tcx * P.fdim[0]), nxm1);
            const int vj = min((int)(tcy * P.fdim[1]), nym1);
            const int vk = min((int)(tcz * P.fdim[2]), nzm1);
            float c, cg = 0.0f, cb = 0.0f, a;
            classify(VoxelFetch<VoxelT, false>::load(vol, rsrc, VoxelAddr<LAYOUT, false>::at(P, vi, vj, vk)) - (uint32_t)pkb, c, cg, cb, a);
            accumulate(drgb, dg, db, da, c, cg, cb, a);
            qx += tsx; qy += tsy; qz += tsz;
        }
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MODE >= 2) store_pixel(P, fb, pix, drgb, dg, db, da);
    else if (MODE == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    if (spp) spp[pix] = hit ? (uint32_t)i : 0u;
}

#if VR_TU_MAIN
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

// ------------------------------------------------------------------ launchers
#endif  // VR_TU_MAIN

[[maybe_unused]] static inline unsigned padded_blocks(unsigned tiles_x, unsigned tiles_y)
{
    // every XCD gets ceil(tiles_y/8) tile rows' worth of slots; extras are padding
    const unsigned rows0 = (tiles_y + 7u) / 8u;
    return rows0 * tiles_x * 8u;
}

template <typename VoxelT, int LAYOUT, bool COUNT>
static hipError_t launch_generic(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                                 float4 *fb, uint32_t *spp, unsigned tiles_x, unsigned tiles_y, hipStream_t st)
{
    const int div = L.divmode_tc == DIV_EXACT ? DIV_EXACT : DIV_CERT;
    if (L.big_offsets)
        hipLaunchKernelGGL((raymarch_generic_kernel<VoxelT, LAYOUT, COUNT, true>), dim3(padded_blocks(tiles_x, tiles_y)),
                           dim3(256), 0, st, P, L.filter, L.mip, div, 0u, (const VoxelT *)vol, tf, fb, spp, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((raymarch_generic_kernel<VoxelT, LAYOUT, COUNT, false>), dim3(padded_blocks(tiles_x, tiles_y)),
                           dim3(256), 0, st, P, L.filter, L.mip, div, (uint32_t)L.vol_bytes32, (const VoxelT *)vol, tf, fb,
                           spp, tiles_x, tiles_y);
    return hipGetLastError();
}

template <typename VoxelT, int LAYOUT, int DIVTC, int VIEW, bool BIG, bool LUT, bool POW2, bool NOCLAMP, int MODE, bool SKIPT, int BATCH, bool ATAB, bool PK12>
static hipError_t launch_fast(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                              uint32_t *spp, int rows, hipStream_t st)
{
    const FastGrid g = fast_grid(P.img_w, rows);
    const unsigned blocks = L.tile_table ? L.tile_table_blocks : g.blocks;
    if constexpr (!SKIPT && !BIG && ATAB && BATCH == 8) {
        if (L.short_batches) {      // rays that end early (high opacity): 4-sample batches waste less behind the terminating sample
            hipLaunchKernelGGL((raymarch_fast_kernel<VoxelT, LAYOUT, DIVTC, VIEW, BIG, LUT, POW2, NOCLAMP, MODE, SKIPT, 4, ATAB, PK12, false>), dim3(blocks),
                               dim3(FAST_THREADS), 0, st, P, (const VoxelT *)vol, tf, (uint32_t)L.vol_bytes32, fb, spp,
                               g.tiles_x, g.tiles_y, g.chunks_per_row, L.tile_table, L.skip_grid, L.skip_grid_bytes,
                               PK12 ? L.packed12 : nullptr, PK12 ? L.packed12_bytes : 0u);
            return hipGetLastError();
        }
        if (L.pipelined) {          // under-filled launch: 16 gathers per lane in flight shorten the serial chain
            hipLaunchKernelGGL((raymarch_fast_kernel<VoxelT, LAYOUT, DIVTC, VIEW, BIG, LUT, POW2, NOCLAMP, MODE, SKIPT, BATCH, ATAB, PK12, true>), dim3(blocks),
                               dim3(FAST_THREADS), 0, st, P, (const VoxelT *)vol, tf, (uint32_t)L.vol_bytes32, fb, spp,
                               g.tiles_x, g.tiles_y, g.chunks_per_row, L.tile_table, L.skip_grid, L.skip_grid_bytes,
                               PK12 ? L.packed12 : nullptr, PK12 ? L.packed12_bytes : 0u);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((raymarch_fast_kernel<VoxelT, LAYOUT, DIVTC, VIEW, BIG, LUT, POW2, NOCLAMP, MODE, SKIPT, BATCH, ATAB, PK12, false>), dim3(blocks),
                       dim3(FAST_THREADS), 0, st, P, (const VoxelT *)vol, tf, (uint32_t)L.vol_bytes32, fb, spp,
                       g.tiles_x, g.tiles_y, g.chunks_per_row, L.tile_table, L.skip_grid, L.skip_grid_bytes,
                       PK12 ? L.packed12 : nullptr, PK12 ? L.packed12_bytes : 0u);
    return hipGetLastError();
}

// The relay kernel serves sparse launches (host: fewer than 256 active tiles -- one rank's shard of a multi-GPU
// frame -- or vr_set_kernel_variant 3) with 32-bit offsets and no skipping.  The headline shape (grey
// composite, default view) has every table / clamp / packing variant; the other modes and views (MIP, transfer
// function, view top / bottom: round 2) are built for the configuration such shards actually have -- a
// classification table and per-axis address tables in LDS (+ the 12-bit packed copy) -- and anything else stays
// with the fast kernel.
static bool relay_selected(const FrameParams &P, const LaunchConfig &L)
{
    if (!(L.sparse_shard && L.tile_table && !L.big_offsets && !(P.skip_empty != 0 && L.skip_grid != nullptr))) return false;
    const bool headline = !L.mip && (P.tf_len <= 1 || P.tf_grey != 0) && P.view_top != 1 && P.view_bottom != 1;
    return headline || (L.use_lut != 0 && P.nx + P.ny + P.nz <= FAST_AXIS_TAB_MAX);
}

template <typename VoxelT, int LAYOUT, int VIEW, int MODE>
static hipError_t dispatch_relay(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                 uint32_t *spp, hipStream_t st)
{
    const bool lut = L.use_lut != 0, noclamp = lut && L.lut_noclamp != 0;
    const bool pow2 = L.pow2_dims != 0 && L.divmode_tc == DIV_UNIT;
    const dim3 grid(L.tile_table_blocks * (8u / RELAY_TILES)), block(RELAY_THREADS);
    const bool atab = P.nx + P.ny + P.nz <= FAST_AXIS_TAB_MAX;
    constexpr bool CAN_PK12 = sizeof(VoxelT) == 2 && LAYOUT == 1;
    const bool pk12 = atab && CAN_PK12 && L.packed12 != nullptr;
#define VR_RELAY2(TC, LT, P2, NC, AT, PK)                                                                         \
    do {                                                                                                          \
        hipLaunchKernelGGL((raymarch_relay_kernel<VoxelT, LAYOUT, TC, LT, P2, NC, AT, PK, VIEW, MODE>), grid, block, 0, st, P, \
                           (const VoxelT *)vol, tf, (uint32_t)L.vol_bytes32, fb, spp, L.tile_table,                 \
                           (PK) ? L.packed12 : nullptr, (PK) ? L.packed12_bytes : 0u);                             \
        return hipGetLastError();                                                                                 \
    } while (0)
    if constexpr (VIEW == 0 && MODE == 0) {
#define VR_RELAY(TC, LT, P2, NC)                                                                                  \
    do {                                                                                                          \
        if (pk12) VR_RELAY2(TC, LT, P2, NC, true, CAN_PK12);                                                      \
        if (atab) VR_RELAY2(TC, LT, P2, NC, true, false);                                                         \
        VR_RELAY2(TC, LT, P2, NC, false, false);                                                                  \
    } while (0)
        if (L.divmode_tc == DIV_CERT) {
            if (lut) { if (noclamp) VR_RELAY(DIV_CERT, true, false, true); else VR_RELAY(DIV_CERT, true, false, false); }
            VR_RELAY(DIV_CERT, false, false, false);
        }
        if (pow2) {
            if (lut) { if (noclamp) VR_RELAY(DIV_UNIT, true, true, true); else VR_RELAY(DIV_UNIT, true, true, false); }
            VR_RELAY(DIV_UNIT, false, true, false);
        }
        if (lut) { if (noclamp) VR_RELAY(DIV_UNIT, true, false, true); else VR_RELAY(DIV_UNIT, true, false, false); }
        VR_RELAY(DIV_UNIT, false, false, false);
#undef VR_RELAY
    } else {
        // relay_selected() has checked: table + address tables
#define VR_RELAY(TC, P2)                                                                                          \
    do {                                                                                                          \
        if (pk12) VR_RELAY2(TC, true, P2, false, true, CAN_PK12);                                                 \
        VR_RELAY2(TC, true, P2, false, true, false);                                                              \
    } while (0)
        if (L.divmode_tc == DIV_CERT) VR_RELAY(DIV_CERT, false);
        if (pow2) VR_RELAY(DIV_UNIT, true);
        VR_RELAY(DIV_UNIT, false);
#undef VR_RELAY
    }
#undef VR_RELAY2
}

template <typename VoxelT, int LAYOUT, int VIEW, bool BIG, int MODE>
static hipError_t dispatch_fast3(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                                 float4 *fb, uint32_t *spp, int rows, hipStream_t st)
{
    // the no-clamp specialisation is kept for the headline mode only (compile time)
    const bool lut = L.use_lut != 0, noclamp = MODE == 0 && lut && L.lut_noclamp != 0;
    const bool pow2 = L.pow2_dims != 0 && L.divmode_tc == DIV_UNIT;
    // sparse shards: four wavefronts relay one 8x8 tile
    if constexpr (!BIG) {
        if (relay_selected(P, L)) return dispatch_relay<VoxelT, LAYOUT, VIEW, MODE>(P, L, vol, tf, fb, spp, st);
    }
    // the skipping-free build exists for the headline shape only (MODE 0, default view)
    constexpr bool HEADLINE = MODE == 0 && VIEW == 0;
    const bool noskip = HEADLINE && !(P.skip_empty != 0 && L.skip_grid != nullptr);
    // LDS address tables whenever offsets are 32-bit and nx + ny + nz entries fit; the 12-bit
    // packed copy (host: refreshPacked12) rides on them
    // (volumes beyond 32-bit offsets: 64-bit tables for the bricked layout, with a classification table cut to
    // 8 KiB -- the window has to fit it)
    constexpr bool CAN_ATAB = !BIG || LAYOUT == 1, CAN_PK12 = !BIG && sizeof(VoxelT) == 2 && LAYOUT == 1;
    const int64_t win_width = (int64_t)P.max_val - (int64_t)P.min_val + 1;
    const bool atab = CAN_ATAB && (BIG ? P.nx + P.ny + P.nz <= FAST_AXIS_TAB_BIG_MAX && (!lut || win_width <= (MODE >= 2 ? 4096 : FAST_BIG_LUT_FLOATS / 2))
                                       : P.nx + P.ny + P.nz <= FAST_AXIS_TAB_MAX);
    const bool pk12 = atab && CAN_PK12 && L.packed12 != nullptr;
#define VR_LAUNCH2(TC, LT, P2, NC, SK)                                                                                    \
    (pk12 ? launch_fast<VoxelT, LAYOUT, TC, VIEW, BIG, LT, P2, (NC) && MODE == 0, MODE, SK, 8, CAN_ATAB, CAN_PK12>(P, L, vol, tf, fb, spp, rows, st) \
     : atab ? launch_fast<VoxelT, LAYOUT, TC, VIEW, BIG, LT, P2, (NC) && MODE == 0, MODE, SK, 8, CAN_ATAB, false>(P, L, vol, tf, fb, spp, rows, st)  \
            : launch_fast<VoxelT, LAYOUT, TC, VIEW, BIG, LT, P2, (NC) && MODE == 0, MODE, SK, 8, false, false>(P, L, vol, tf, fb, spp, rows, st))
#define VR_LAUNCH(TC, LT, P2, NC) (noskip ? VR_LAUNCH2(TC, LT, P2, NC, !HEADLINE) : VR_LAUNCH2(TC, LT, P2, NC, true))
    if (L.divmode_tc == DIV_CERT) {
        if (lut) return noclamp ? VR_LAUNCH(DIV_CERT, true, false, true) : VR_LAUNCH(DIV_CERT, true, false, false);
        if (MODE < 2) return VR_LAUNCH(DIV_CERT, false, false, false);
        return hipErrorInvalidValue;
    }
    if (pow2) {
        if (lut) return noclamp ? VR_LAUNCH(DIV_UNIT, true, true, true) : VR_LAUNCH(DIV_UNIT, true, true, false);
        if (MODE < 2) return VR_LAUNCH(DIV_UNIT, false, true, false);
        return hipErrorInvalidValue;
    }
    if (lut) return noclamp ? VR_LAUNCH(DIV_UNIT, true, false, true) : VR_LAUNCH(DIV_UNIT, true, false, false);
    if (MODE < 2) return VR_LAUNCH(DIV_UNIT, false, false, false);
    return hipErrorInvalidValue;
#undef VR_LAUNCH
#undef VR_LAUNCH2
}

template <typename VoxelT, int LAYOUT, int VIEW, bool BIG>
static hipError_t dispatch_fast2(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                                 float4 *fb, uint32_t *spp, int rows, hipStream_t st)
{
    if (L.mip && P.tf_len > 1) return dispatch_fast3<VoxelT, LAYOUT, VIEW, BIG, 3>(P, L, vol, tf, fb, spp, rows, st);
    if (L.mip) return dispatch_fast3<VoxelT, LAYOUT, VIEW, BIG, 1>(P, L, vol, tf, fb, spp, rows, st);
    // (a grey transfer function whose window fits the (c, a) table runs on the grey-ramp instances: FrameParams::tf_grey)
    if (P.tf_len > 1 && P.tf_grey == 0) return dispatch_fast3<VoxelT, LAYOUT, VIEW, BIG, 2>(P, L, vol, tf, fb, spp, rows, st);
    return dispatch_fast3<VoxelT, LAYOUT, VIEW, BIG, 0>(P, L, vol, tf, fb, spp, rows, st);
}

template <typename VoxelT, int LAYOUT>
static hipError_t dispatch_fast(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                                float4 *fb, uint32_t *spp, int rows, hipStream_t st)
{
    const int view = P.view_top == 1 ? 1 : (P.view_bottom == 1 ? 2 : 0);
    if (L.big_offsets) {
        if (view == 0) return dispatch_fast2<VoxelT, LAYOUT, 0, true>(P, L, vol, tf, fb, spp, rows, st);
        if (view == 1) return dispatch_fast2<VoxelT, LAYOUT, 1, true>(P, L, vol, tf, fb, spp, rows, st);
        return dispatch_fast2<VoxelT, LAYOUT, 2, true>(P, L, vol, tf, fb, spp, rows, st);
    }
    if (view == 0) return dispatch_fast2<VoxelT, LAYOUT, 0, false>(P, L, vol, tf, fb, spp, rows, st);
    if (view == 1) return dispatch_fast2<VoxelT, LAYOUT, 1, false>(P, L, vol, tf, fb, spp, rows, st);
    return dispatch_fast2<VoxelT, LAYOUT, 2, false>(P, L, vol, tf, fb, spp, rows, st);
}

// TRILINEAR through the batched kernel (host: tri_path_eligible)
template <typename VoxelT, int LAYOUT>
static hipError_t dispatch_tri(const FrameParams &P, const LaunchConfig &L, const void *vol, float4 *fb, uint32_t *spp,
                               hipStream_t st)
{
    const bool pow2 = L.pow2_dims != 0 && L.divmode_tc == DIV_UNIT;
    const int view = P.view_top == 1 ? 1 : (P.view_bottom == 1 ? 2 : 0);
    const dim3 grid(L.tile_table_blocks), block(512);
#define VR_TRI(TC, VW, P2, MP)                                                                                   \
    do {                                                                                                         \
        if (LAYOUT == 1 && L.apron != nullptr) {                                                                 \
            hipLaunchKernelGGL((raymarch_tri_kernel<VoxelT, LAYOUT, TC, VW, P2, MP, LAYOUT == 1>), grid, block, 0, st, P, \
                               (const VoxelT *)L.apron, (uint32_t)L.apron_bytes, fb, spp, L.tile_table);                    \
            return hipGetLastError();                                                                            \
        }                                                                                                        \
        hipLaunchKernelGGL((raymarch_tri_kernel<VoxelT, LAYOUT, TC, VW, P2, MP, false>), grid, block, 0, st, P,   \
                           (const VoxelT *)vol, (uint32_t)L.vol_bytes32, fb, spp, L.tile_table);                  \
        return hipGetLastError();                                                                                \
    } while (0)
#define VR_TRI_V(TC, P2, MP)                                                                                     \
    do {                                                                                                         \
        if (view == 0) VR_TRI(TC, 0, P2, MP);                                                                    \
        if (view == 1) VR_TRI(TC, 1, P2, MP);                                                                    \
        VR_TRI(TC, 2, P2, MP);                                                                                   \
    } while (0)
    if (L.divmode_tc == DIV_CERT) { if (L.mip) VR_TRI_V(DIV_CERT, false, 1); VR_TRI_V(DIV_CERT, false, 0); }
    if (pow2) { if (L.mip) VR_TRI_V(DIV_UNIT, true, 1); VR_TRI_V(DIV_UNIT, true, 0); }
    if (L.mip) VR_TRI_V(DIV_UNIT, false, 1);
    VR_TRI_V(DIV_UNIT, false, 0);
#undef VR_TRI_V
#undef VR_TRI
}

// ray-march launch of one (voxel type, layout) pair: the specialised kernels when `fast`, else
// the generic one
template <typename T, int LAY>
static hipError_t raymarch_tu(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                              uint32_t *spp, int rows, int fast, hipStream_t st)
{
    if (fast == 1) return dispatch_fast<T, LAY>(P, L, vol, tf, fb, spp, rows, st);
    if (fast == 2) return dispatch_tri<T, LAY>(P, L, vol, fb, spp, st);
    const unsigned tiles_x = (unsigned)((P.img_w + 15) / 16), tiles_y = (unsigned)((rows + 15) / 16);
    return spp ? launch_generic<T, LAY, true>(P, L, vol, tf, fb, spp, tiles_x, tiles_y, st)
               : launch_generic<T, LAY, false>(P, L, vol, tf, fb, spp, tiles_x, tiles_y, st);
}

// one empty kernel per translation unit: launching it makes the runtime load that unit's code
// object (vr_load_shader does this up front; a cold first ray-march launch costs ~16 ms otherwise)
#if VR_TU >= 0
#define VR_WARM_CAT2(a, b) a##b
#define VR_WARM_CAT(a, b) VR_WARM_CAT2(a, b)
__global__ void VR_WARM_CAT(warm_kernel_tu, VR_TU)() {}
hipError_t VR_WARM_CAT(launch_warm_tu, VR_TU)(hipStream_t st)
{
    hipLaunchKernelGGL(VR_WARM_CAT(warm_kernel_tu, VR_TU), dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
#endif

#define VR_TU_ARGS const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb, uint32_t *spp, int rows, int fast, hipStream_t st
#if VR_TU == 0
hipError_t launch_raymarch_tu0(VR_TU_ARGS) { return raymarch_tu<uint8_t, 0>(P, L, vol, tf, fb, spp, rows, fast, st); }
#elif VR_TU == 1
hipError_t launch_raymarch_tu1(VR_TU_ARGS) { return raymarch_tu<uint8_t, 1>(P, L, vol, tf, fb, spp, rows, fast, st); }
#elif VR_TU == 2
hipError_t launch_raymarch_tu2(VR_TU_ARGS) { return raymarch_tu<uint16_t, 0>(P, L, vol, tf, fb, spp, rows, fast, st); }
#elif VR_TU == 3
hipError_t launch_raymarch_tu3(VR_TU_ARGS) { return raymarch_tu<uint16_t, 1>(P, L, vol, tf, fb, spp, rows, fast, st); }
#endif

#if VR_TU_MAIN
// The specialised kernel covers NEAREST + iterative accumulation with a non-degenerate
// window whose divisions were certified and alpha_scale in [0,1]: grey-ramp composite,
// grey-ramp MIP, and composite through a transfer-function table that fits LDS.
// Everything else (TRILINEAR, closed-form accumulation, MIP + TF, ...) is generic.
bool fast_path_eligible(const FrameParams &P, const LaunchConfig &L)
{
    const bool tf = P.tf_len > 1;
    if (tf && !L.use_lut) return false;
    return !L.generic && L.filter == 0 && P.accum == 0 && P.fden > 0.0f &&
           P.max_val > P.min_val && L.divmode_win == DIV_CERT && L.divmode_tc != DIV_EXACT &&
           P.alpha_scale >= 0.0f && P.alpha_scale <= 1.0f;
}

// TRILINEAR has a batched kernel for the grey modes: same preconditions as the fast path apart
// from the filter, plus 32-bit offsets, address tables that fit LDS and a tile table
bool tri_path_candidate(const FrameParams &P, const LaunchConfig &L)
{
    return !L.generic && L.filter == 1 && P.accum == 0 && P.tf_len <= 1 && P.fden > 0.0f && P.max_val > P.min_val &&
           L.divmode_win == DIV_CERT && L.divmode_tc != DIV_EXACT && P.alpha_scale >= 0.0f && P.alpha_scale <= 1.0f &&
           !L.big_offsets && P.nx + P.ny + P.nz <= FAST_AXIS_TAB_MAX;
}

static bool tri_path_eligible(const FrameParams &P, const LaunchConfig &L) { return tri_path_candidate(P, L) && L.tile_table != nullptr; }

// TRILINEAR on the LDS-staged kernel (vr_tslab.hip): every mode (the 256-entry transfer-function table sits in LDS), any
// volume size (64-bit DMA addresses), the bricked layout; the apron copy must be resident (host: refreshApron)
bool tri_slab_candidate(const FrameParams &P, const LaunchConfig &L)
{
    if (L.generic || L.filter != 1 || P.accum != 0 || !(P.fden > 0.0f) || !(P.max_val > P.min_val) || L.divmode_win != DIV_CERT ||
        L.divmode_tc == DIV_EXACT || !(P.alpha_scale >= 0.0f && P.alpha_scale <= 1.0f) || L.layout != 1) return false;
    if (P.tf_len > 256) return false;
    const uint64_t bricks = (uint64_t)P.bnx * (uint64_t)P.bny * (uint64_t)P.bnz;
    if (bricks >= (1ull << 32) || (uint64_t)P.bnx * (uint64_t)P.bny >= (1ull << 24)) return false;
    return P.nx <= 32768 && P.ny <= 32768 && P.nz <= 32768;           // 16-bit brick indices in the load plan; plan and tables cover a tile's own range only
}

static bool tri_slab_selected(const FrameParams &P, const LaunchConfig &L)
{
    return L.tri_slab && L.apron != nullptr && L.tile_table != nullptr && tri_slab_candidate(P, L);
}

hipError_t launch_raymarch_slab_tri_u8(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                       uint32_t *spp, hipStream_t st);
hipError_t launch_raymarch_slab_tri_u16(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                        uint32_t *spp, hipStream_t st);

int launch_local_rows(const FrameParams &P)
{
    int rows;
    if (P.stripe_count > 1) {
        const int nstripes_total = (P.img_h + P.stripe_rows - 1) / P.stripe_rows;
        const int mine = (nstripes_total - P.stripe_index + P.stripe_count - 1) / P.stripe_count;
        rows = mine * P.stripe_rows;
    } else {
        rows = P.row_end - P.row_begin;
    }
    return rows;
}

#if VR_TU == -1
hipError_t launch_raymarch_tu0(VR_TU_ARGS);
hipError_t launch_raymarch_tu1(VR_TU_ARGS);
hipError_t launch_raymarch_tu2(VR_TU_ARGS);
hipError_t launch_raymarch_tu3(VR_TU_ARGS);
#endif

hipError_t launch_raymarch(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                           float4 *fb, uint32_t *spp, hipStream_t st, const char **kernel_name)
{
    const int rows = launch_local_rows(P);   // local image rows covered by this launch
    if (rows <= 0 || P.img_w <= 0) return hipSuccess;
    if (tri_slab_selected(P, L)) {
        if (kernel_name) *kernel_name = "raymarch_slab_tri_kernel";
        return L.bytes_per_voxel == 1 ? launch_raymarch_slab_tri_u8(P, L, vol, tf, fb, spp, st) : launch_raymarch_slab_tri_u16(P, L, vol, tf, fb, spp, st);
    }
    const int fast = fast_path_eligible(P, L) ? 1 : (tri_path_eligible(P, L) ? 2 : 0);
    if (kernel_name)
        *kernel_name = fast == 0 ? "raymarch_generic_kernel"
                                 : (fast == 2 ? "raymarch_tri_kernel"
                                              : (relay_selected(P, L) ? "raymarch_relay_kernel" : "raymarch_fast_kernel"));
    const int tu = (L.bytes_per_voxel == 1 ? 0 : 2) + (L.layout == 0 ? 0 : 1);
#if VR_TU == -1
    switch (tu) {
    case 0: return launch_raymarch_tu0(P, L, vol, tf, fb, spp, rows, fast, st);
    case 1: return launch_raymarch_tu1(P, L, vol, tf, fb, spp, rows, fast, st);
    case 2: return launch_raymarch_tu2(P, L, vol, tf, fb, spp, rows, fast, st);
    default: return launch_raymarch_tu3(P, L, vol, tf, fb, spp, rows, fast, st);
    }
#else
    switch (tu) {
    case 0: return raymarch_tu<uint8_t, 0>(P, L, vol, tf, fb, spp, rows, fast, st);
    case 1: return raymarch_tu<uint8_t, 1>(P, L, vol, tf, fb, spp, rows, fast, st);
    case 2: return raymarch_tu<uint16_t, 0>(P, L, vol, tf, fb, spp, rows, fast, st);
    default: return raymarch_tu<uint16_t, 1>(P, L, vol, tf, fb, spp, rows, fast, st);
    }
#endif
}

__global__ void warm_kernel_main() {}
#if VR_TU == -1
hipError_t launch_warm_tu0(hipStream_t st);
hipError_t launch_warm_tu1(hipStream_t st);
hipError_t launch_warm_tu2(hipStream_t st);
hipError_t launch_warm_tu3(hipStream_t st);
#endif
hipError_t launch_warm_modules(hipStream_t st)
{
    hipLaunchKernelGGL(warm_kernel_main, dim3(1), dim3(64), 0, st);
    hipError_t e = hipGetLastError();
#if VR_TU == -1
    if (e == hipSuccess) e = launch_warm_tu0(st);
    if (e == hipSuccess) e = launch_warm_tu1(st);
    if (e == hipSuccess) e = launch_warm_tu2(st);
    if (e == hipSuccess) e = launch_warm_tu3(st);
#endif
    return e;
}

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

// how many cells of the dilated grid can be skipped at all for a threshold (host: refreshSkipGrid -- none: the launch runs the
// instances without skipping, whose loops are a few percent leaner)
__global__ __launch_bounds__(256) void count_cells_le_kernel(const uint16_t *__restrict__ grid, uint64_t cells, int thresh, unsigned long long *__restrict__ count)
{
    unsigned n = 0;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (uint64_t)gridDim.x * blockDim.x) n += (int)grid[c] <= thresh ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) n += (unsigned)__shfl_xor((int)n, o);
    if ((threadIdx.x & 63u) == 0u && n != 0u) atomicAdd(count, (unsigned long long)n);
}

hipError_t launch_count_cells_le(const uint16_t *grid, uint64_t cells, int thresh, unsigned long long *count, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)std::min<uint64_t>((cells + 255) / 256, 2048u);
    hipLaunchKernelGGL(count_cells_le_kernel, dim3(blocks), dim3(256), 0, st, grid, cells, thresh, count);
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

#endif  // VR_TU_MAIN

}  // namespace vr
