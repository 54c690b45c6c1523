// tile_schedule.h -- work-ordered, XCD-balanced block -> tile table for the fast kernel.
//
// Rays that cross the whole volume take ~4x longer than rays clipping a corner, and one
// frame is only ~1.3 "rounds" of resident wavefronts, so dispatch order decides the tail:
// blocks are issued longest-first (LPT) and dealt round-robin to the 8 XCDs (block b runs
// on XCD b % 8).  A scheduling heuristic only: pixels are independent, the image does not
// depend on it.
#pragma once
#include <stdint.h>

#include <vector>

#include "vr_frame.h"

namespace vr {

constexpr unsigned kFastTileW = 32, kFastTileH = 16, kFastChunk = 4;   // kFastChunk: table-less fallback order
constexpr unsigned kFastChunkW = 1, kFastChunkH = 1;                    // chunk of tiles dealt to one XCD
constexpr uint32_t kTilePadding = 0xffffffffu;

// table[b] = tile_x | tile_y << 16 for block b, kTilePadding for padding blocks.
// `rows` = local image rows of this launch (stripe padding included).
// returns the number of tiles with a non-zero work estimate (*max_ray_samples: the longest ray's expected sample count)
// tile_h / tile_w: rows / columns per tile (kFastTileH x kFastTileW; the staged trilinear kernel also runs 32x32-pixel tiles on
// 16-wavefront workgroups and 16x32-pixel ones)
// work_scale (tiles_x * tiles_y floats in (0, 1], row-major; nullptr = none): the ORDER follows the geometric ray lengths times
// these -- empty-space skipping, where a tile's cost is its rays' visible stretch (vr_kernels.hip: tile_visible_work_kernel)
unsigned buildTileSchedule(const FrameParams &P, int rows, std::vector<uint32_t> &table, double *max_ray_samples = nullptr, unsigned tile_h = kFastTileH,
                           unsigned tile_w = kFastTileW, const float *work_override = nullptr);

// how close the view is to a volume axis: |largest component| of the central ray's direction in voxel units, 1 = along
// an axis, 0.58 = along the space diagonal.  Launch heuristics only (RendererCore::prepareLaunch).
double viewAxisAlignment(const FrameParams &P);
// the central ray's two smaller |components| (voxel units) over its largest: 0, 0 = along an axis; 1, 1 = a body diagonal
void viewAxisRatios(const FrameParams &P, double &mid_over_max, double &min_over_max);

// cheap fingerprint of everything the schedule depends on; with_camera = false: of everything
// that decides WHICH tiles exist (image, shard, volume box) -- the camera only decides their order
uint64_t tileScheduleKey(const FrameParams &P, int rows, bool with_camera = true);

}  // namespace vr
