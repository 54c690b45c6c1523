// vr_frame.h -- per-frame constant block shared by the host (renderer_core.cpp) and
// the gfx950 kernels (vr_kernels.hip).  It is the HIP-side equivalent of the
// reference's 7 uniforms + 84-byte Camera UBO (VolumeRenderer.cs:28-44) plus the
// values main() derives from them once per invocation (VolumeRenderer.cs:62-83).
#pragma once
#include <stdint.h>

namespace vr {

enum : int { DIV_UNIT = 0, DIV_CERT = 1, DIV_EXACT = 2 };

// VR_LAYOUT_BRICKED: the volume is stored as bricks of 64 voxels (128 B of u16, one cache
// line; 64 B of u8), BRICK_X x BRICK_Y x BRICK_Z voxels each, x fastest inside a brick,
// bricks in x-y-z order.  The shape is a build constant shared by host and kernels.
#ifndef VR_BRICK_LX
#define VR_BRICK_LX 2
#define VR_BRICK_LY 2
#define VR_BRICK_LZ 2
#endif
constexpr int BRICK_LX = VR_BRICK_LX, BRICK_LY = VR_BRICK_LY, BRICK_LZ = VR_BRICK_LZ;
constexpr int BRICK_X = 1 << BRICK_LX, BRICK_Y = 1 << BRICK_LY, BRICK_Z = 1 << BRICK_LZ;
static_assert(BRICK_LX + BRICK_LY + BRICK_LZ == 6 && BRICK_LX >= 1, "bricks hold 64 voxels");

// TRILINEAR's apron copy (vr_device.h: build_axis_tables_apron): 4x4x4 bricks stored as 5x4x4
constexpr uint32_t APRON_BRICK_VOXELS = 80;
constexpr uint64_t apron_voxels(int nx, int ny, int nz)
{
    return (uint64_t)((nx + 3) >> 2) * (uint64_t)((ny + 3) >> 2) * (uint64_t)((nz + 3) >> 2) * APRON_BRICK_VOXELS;
}

// widest window (in voxel values) the specialised kernels classify through LDS with a transfer
// function: one index byte per value behind the 256-entry RGBA table in a 32 KiB array
constexpr int FAST_TF_WINDOW_MAX = 4096 * 8 - 256 * 16;

struct FrameParams {
    float cam[21];                 // view_mat columns, eye, view_plane_dist
    int32_t img_w, img_h;          // imageSize(render_texture)
    int32_t row_begin, row_end;    // global rows rendered by this launch
    int32_t col_lim, row_lim;      // Q1 limits (== img_w/img_h when the quirk is off)
    int32_t stripe_rows, stripe_index, stripe_count;  // cyclic row stripes (count 1 = off)
    int32_t fb_compact;            // 1: target holds only this shard's rows (local row index)
    int32_t fb_format;             // 0: RGBA32F; 1: (grey, alpha) float2 per pixel (grey modes only: r == g == b)
    int32_t nx, ny, nz;            // textureSize(vol_tex3D)
    float fdim[3];                 // float(textureSize)
    float half[3];                 // half_len
    float pmin[3], pmax[3];        // bb after centring
    float ext[3];                  // bb.p_max + half_len   (cartesianToTextureCoord)
    float rext[3];                 // RN(1/ext) for the certified division
    float step;                    // step_size
    float alpha_scale;
    float fmin, fmax, fden;        // vec4(min_val), vec4(max_val), float(max_val-min_val)
    float rden;                    // RN(1/fden)
    int32_t min_val, max_val;
    int32_t view_top, view_bottom;
    int32_t max_steps;             // 10000
    int32_t accum;                 // VR_ACCUM_*
    int32_t tf_len;                // 0 = grey ramp
    int32_t tf_grey;               // 1: the transfer function is grey (r == g == b in every entry) and this launch folds it into the grey-ramp kernels' (c, a) table (NEAREST composite, window <= the table)
    int32_t skip_empty;            // exact empty-space skipping on the dilated cell-max grid
    int32_t skip_thresh;           // a cell is empty when its (dilated) max voxel <= skip_thresh
    int32_t cnx, cny, cnz;         // cells (8x8x8 voxels) per axis
    // bricked layout (VR_LAYOUT_BRICKED): bricks of 4x4x4 voxels, x-fastest inside
    int32_t bnx, bny, bnz;         // bricks per axis
    uint32_t bstride_y, bstride_z; // brick-row / brick-slab strides minus the in-brick part (see VoxelAddr)
    int32_t pk12_base;             // the 12-bit packed copy holds voxel - pk12_base (the dataset minimum); 0 without a packed copy
};

struct LaunchConfig {
    int bytes_per_voxel;           // 1 | 2
    int filter;                    // VR_FILTER_*
    int mip;                       // 0 | 1
    int divmode_tc;                // DIV_* for the texcoord division
    int divmode_win;               // DIV_CERT | DIV_EXACT for the window division
    int layout;                    // VR_LAYOUT_*
    int generic;                   // force the generic (always-checked) kernel
    int big_offsets;               // voxel offsets need 64-bit arithmetic
    int use_lut;                   // LDS classification table (window width <= 4096 entries)
    int lut_noclamp;               // exact dataset range lies inside the window
    int pow2_dims;                 // nx, ny, nz are powers of two
    uint32_t vol_bytes32;          // volume allocation size for the buffer descriptor (!big)
    const uint16_t *skip_grid;     // device: dilated per-cell max (nullptr = no skipping)
    uint32_t skip_grid_bytes;
    int sparse_shard;              // use the 4-wavefront relay kernel when the shape allows
    const uint32_t *tile_table;    // device: work-ordered block -> tile table (nullptr = arithmetic order)
    uint32_t tile_table_blocks;
    const uint32_t *tile_table_tall;   // device: the same for 16x32-pixel tiles (8 wavefronts, 2 x 4: the staged trilinear kernel's tall shape; nullptr = none)
    uint32_t tile_table_tall_blocks;
    const uint32_t *tile_table_small;  // device: the same for 16x16-pixel tiles (4 wavefronts, 2 x 2: the staged trilinear kernel's small shape; nullptr = none)
    uint32_t tile_table_small_blocks;
    const void *packed12;          // device: 12-bit packed copy of the bricked u16 volume (nullptr = none)
    uint32_t packed12_bytes;
    const void *apron;             // device: TRILINEAR's apron copy of the volume (nullptr = none; vr_device.h)
    const void *apron_y, *apron_x; // device: the apron copy with the bricks' planes along y / x slowest (orders 1 and 2 of relayout_apron_kernel; nullptr = none): half layers of the staged kernel
    uint64_t apron_bytes;          // (beyond 4 GiB only the LDS-staged trilinear kernel uses it: no buffer descriptor)
    int tri_slab;                  // TRILINEAR on the LDS-staged kernel where eligible (vr_tslab.hip): 1 = 32x16-pixel tiles, order-0 copy, whole layers (kernel variant 6); 2 = that with staging switched off (variant 7); 3 = 16-bit volumes: per-axis copies and the layer thickness per tile (variant 8); 4 = 16x32-pixel tiles with rows (variant 9; 16-bit volumes as 3, 8-bit ones with whole layers); 5 = as 1 on 53 KiB, three workgroups per CU (variant 10); 6 = 16x16-pixel tiles on four wavefronts and 40 KiB, four workgroups per CU (variant 11; 16-bit volumes with the per-axis copies when they are resident)
    int short_batches;             // fast kernel with 4-sample batches (rays expected to end early: alpha_scale >= 0.5)
    int pipelined;                 // fast kernel with the software-pipelined batch loop (under-filled launches; vr_set_kernel_variant 5 forces it)
};

}  // namespace vr
