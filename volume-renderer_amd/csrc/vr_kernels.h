// vr_kernels.h -- host-callable launchers of vr_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vr_frame.h"

namespace vr {

// true when (P, L) runs on the specialised NEAREST/composite/iterative kernel
bool fast_path_eligible(const FrameParams &P, const LaunchConfig &L);
// TRILINEAR in a grey mode with everything the batched trilinear kernel needs except the tile table
bool tri_path_candidate(const FrameParams &P, const LaunchConfig &L);

// TRILINEAR on the LDS-staged kernel (vr_tslab.hip): the trilinear path's preconditions with any mode (grey, MIP,
// transfer function), the bricked layout with its apron copy resident, torus tables that fit LDS; volumes beyond
// 32-bit offsets included
bool tri_slab_candidate(const FrameParams &P, const LaunchConfig &L);

// local image rows one launch covers (stripe padding included)
int launch_local_rows(const FrameParams &P);

// The ray-march launch (replaces glDispatchCompute, src/RendererCore.cpp:150).
// spp != nullptr selects the instrumented variant that also stores fetch counts.
hipError_t launch_raymarch(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf,
                           float4 *fb, uint32_t *spp, hipStream_t st, const char **kernel_name);

// dilated per-cell (8^3 voxels) maximum for exact empty-space skipping; tmp/out hold
// ceil(nx/8)*ceil(ny/8)*ceil(nz/8) uint16 each
hipError_t launch_build_skip_grid(const void *vol, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                                  uint32_t bnx, uint32_t bny, uint16_t *tmp, uint16_t *out, hipStream_t st);

// smallest value of the dilated grid (*out: device, 4 bytes): thresholds below it skip nothing
hipError_t launch_min_cell(const uint16_t *grid, uint64_t cells, unsigned *out, hipStream_t st);

// expected cost per tile under empty-space skipping (a scheduling estimate: visible stretch + empty_cost x empty stretch of the
// longest of nine probe rays); work[2 * tiles_x * tiles_y] floats: per tile the cost with skipping and the cost without it (0, 0 = no
// probe ray enters the box) -- the caller scales its own geometric estimate by their ratio
// per_layer: the LDS-staged TRILINEAR kernel's granularity (a tile's brick layer is skipped when NO probe ray touches anything visible in it)
hipError_t launch_tile_visible_work(const FrameParams &P, const uint16_t *grid, int rows, unsigned tile_w, unsigned tile_h, float empty_cost,
                                    int per_layer, float *work, hipStream_t st);

// *d_bad must be zeroed by the caller; non-zero afterwards = divisor not certified
hipError_t launch_certify_div(float b, float r, unsigned *d_bad, hipStream_t st);

hipError_t launch_gen_volume(void *out, int bytes_per_voxel, int kind, uint32_t nx, uint32_t ny, uint32_t nz,
                             uint32_t param, int layout, uint32_t bnx, uint32_t bny, hipStream_t st);

hipError_t launch_relayout(const void *in, void *out, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz,
                           uint32_t bnx, uint32_t bny, int to_linear, hipStream_t st);

// pass 0: min/max into d_minmax[2] (init {0xffffffff, 0}); pass 1: 256-bin counts
hipError_t launch_stats(const void *vol, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                        uint32_t bnx, uint32_t bny, int pass, float scale255, unsigned *d_minmax, unsigned *d_hist,
                        hipStream_t st);

// 12-bit packed copy (voxels a multiple of 8, every voxel <= 4095): 1.5 bytes per voxel
hipError_t launch_pack12(const void *src_u16, void *dst, uint64_t voxels, uint32_t base, hipStream_t st);

// RGBA32F -> RGBA8 with glReadPixels' conversion rule
hipError_t launch_to_rgba8(const void *fb_rgba32f, void *out_rgba8, uint64_t pixels, hipStream_t st);

// one empty launch per translation unit: loads every code object of the library (vr_load_shader)
hipError_t launch_warm_modules(hipStream_t st);
hipError_t launch_warm_tslab(hipStream_t st);     // the seven units of the LDS-staged TRILINEAR kernel (vr_tslab.hip)

// TRILINEAR's apron copy: 5x4x4-stored 4x4x4 bricks (vr_device.h); `out` holds apron_voxels(nx, ny, nz) voxels.
// order 0: x fastest (5 wide: the apron), y, z slowest; 1: x (apron), z, y slowest; 2: y (apron), z, x slowest -- the
// bricks themselves stay in x-y-z order (vr_tslab.hip: half a brick along the slowest axis is contiguous)
hipError_t launch_relayout_apron(const void *vol, void *out, int bytes_per_voxel, uint32_t nx, uint32_t ny, uint32_t nz, int layout,
                                 uint32_t bnx, uint32_t bny, int order, hipStream_t st);

// multi-GPU assembly on the root device (vr_group.cpp): gathered = n rank-major compact shards of local_rows x W
// pixels with `channels` floats each (2 = (grey, alpha), 4 = RGBA); stripe_rows = 0: contiguous row blocks,
// else cyclic stripes of that many rows; frame = W x H RGBA32F
hipError_t launch_assemble(const void *gathered, float4 *frame, int W, int H, int n, int local_rows, int stripe_rows,
                           int channels, hipStream_t st);

// streaming 16-byte reads of `bytes` bytes (bandwidth probe)
hipError_t launch_stream_read(const void *p, uint64_t bytes, unsigned *sink, hipStream_t st);

}  // namespace vr
