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

// ------------------------------------------------------------------ the kernels, one file each
#include "vr_generic.hip"
#include "vr_fast.hip"
#include "vr_tri.hip"
#include "vr_relay.hip"

// ------------------------------------------------------------------ launchers

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
        if (kernel_name) *kernel_name = "raymarch_tslab_kernel";          // the symbol a profiler shows (vr_tslab.hip)
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

#include "vr_helpers.hip"
#endif  // VR_TU_MAIN

}  // namespace vr
