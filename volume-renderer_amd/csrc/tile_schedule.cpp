// tile_schedule.cpp -- see tile_schedule.h
#include "tile_schedule.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace vr {

namespace {

// expected number of samples of the ray through pixel centre (px, py): slab test of
// VolumeRenderer.cs:218-238 in double precision (an estimate is all that is needed)
double raySamples(const FrameParams &P, double px, double py)
{
    const float *c = P.cam;
    const double w = P.img_w, h = P.img_h;
    const double aspect = w / h;
    double d[3] = {aspect * (2.0 * px / w - 1.0), 2.0 * py / h - 1.0, -(double)c[20]};
    double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (double &v : d) v /= len;
    double m[3];
    for (int r = 0; r < 3; r++) m[r] = c[r] * d[0] + c[4 + r] * d[1] + c[8 + r] * d[2];
    len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    double tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; a++) {
        const double dir = m[a] / len, o = c[16 + a];
        const double t0 = (P.pmin[a] - o) / dir, t1 = (P.pmax[a] - o) / dir;
        tmin = std::max(tmin, std::min(t0, t1));
        tmax = std::min(tmax, std::max(t0, t1));
    }
    if (!(tmax > std::max(tmin, 0.0))) return 0.0;
    return std::min((tmax - tmin) / (double)P.step, (double)P.max_steps);
}

int globalRow(const FrameParams &P, int ly)
{
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        return (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    }
    return P.row_begin + ly;
}

}  // namespace

static void centralRayVoxelUnits(const FrameParams &P, double m[3])
{
    const float *c = P.cam;
    const bool swz = P.view_top == 1 || P.view_bottom == 1;          // box y / z carry voxel z / y in the rotated views
    const double vdim[3] = {(double)P.nx, swz ? (double)P.nz : (double)P.ny, swz ? (double)P.ny : (double)P.nz};
    for (int r = 0; r < 3; r++) {
        const double ext = (double)P.ext[r] > 0.0 ? (double)P.ext[r] : 1.0;
        m[r] = -(double)c[8 + r] * (double)c[20] * vdim[r] / ext;
    }
}

void viewAxisRatios(const FrameParams &P, double &mid_over_max, double &min_over_max)
{
    double m[3];
    centralRayVoxelUnits(P, m);
    double v[3] = {std::fabs(m[0]), std::fabs(m[1]), std::fabs(m[2])};
    std::sort(v, v + 3);
    mid_over_max = v[2] > 0.0 ? v[1] / v[2] : 0.0;
    min_over_max = v[2] > 0.0 ? v[0] / v[2] : 0.0;
}

double viewAxisAlignment(const FrameParams &P)
{
    const float *c = P.cam;
    // central ray: direction (0, 0, -view_plane_dist) through the view matrix, then box units -> voxel units per axis
    // (a permutation of the axes -- the rotated views -- does not change the measure)
    double m[3], n2 = 0.0, big = 0.0;
    const bool swz = P.view_top == 1 || P.view_bottom == 1;          // box y / z carry voxel z / y in the rotated views
    const double vdim[3] = {(double)P.nx, swz ? (double)P.nz : (double)P.ny, swz ? (double)P.ny : (double)P.nz};
    for (int r = 0; r < 3; r++) {
        m[r] = -(double)c[8 + r] * (double)c[20];
        const double ext = (double)P.ext[r] > 0.0 ? (double)P.ext[r] : 1.0;
        m[r] *= vdim[r] / ext;
        n2 += m[r] * m[r];
        big = std::max(big, std::fabs(m[r]));
    }
    return n2 > 0.0 ? big / std::sqrt(n2) : 1.0;
}

uint64_t tileScheduleKey(const FrameParams &P, int rows, bool with_camera)
{
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; }
    };
    if (with_camera) mix(P.cam, sizeof(P.cam));
    mix(&P.img_w, sizeof(int32_t) * 9);      // img_w .. stripe_count (contiguous int32 fields)
    mix(P.pmin, sizeof(P.pmin)); mix(P.pmax, sizeof(P.pmax));
    mix(&P.step, sizeof(P.step)); mix(&P.max_steps, sizeof(P.max_steps));
    mix(&rows, sizeof(rows));
    return hsh;
}

unsigned buildTileSchedule(const FrameParams &P, int rows, std::vector<uint32_t> &table, double *max_ray_samples, unsigned tile_h, unsigned tile_w,
                           const float *work_override)
{
    unsigned active_tiles = 0;
    double longest = 0.0;
    const unsigned tiles_x = (unsigned)((P.img_w + (int)tile_w - 1) / (int)tile_w);
    const unsigned tiles_y = (unsigned)((rows + (int)tile_h - 1) / (int)tile_h);
    // a chunk = CW x CH neighbouring tiles that go to one XCD.  Measured on cfg3 (1x1 ... 16x4):
    // single tiles win -- balance across the XCDs matters more than sharing brick rows in one
    // L2 (round 1: 0.59 ms vs 0.62 ms for 4x1, 0.69 ms for 16x4; re-measured in round 6 on the packed, pipelined kernel with
    // L2 counters, chunks and one compact equal-work region per XCD: 0.461 ms vs 0.465-0.498, HBM reads only 9 % lower at
    // best -- profiles/r06_xcd_ownership.txt).
    unsigned CW = kFastChunkW, CH = kFastChunkH;
    if (P.stripe_count > 1) CH = 1;          // cyclic stripes: vertically adjacent local tiles are not neighbours
    const unsigned cpr = (tiles_x + CW - 1) / CW, cpc = (tiles_y + CH - 1) / CH, per_chunk = CW * CH;
    struct Chunk { double work; unsigned cx, cy; };
    std::vector<Chunk> chunks;
    chunks.reserve((size_t)cpr * cpc);
    std::vector<double> tile_work((size_t)tiles_x * tiles_y, 0.0);
    for (unsigned ty = 0; ty < tiles_y; ty++) {
        // probe rows: top, middle, bottom of the tile (global rows of this shard)
        const int ly0 = (int)(ty * tile_h);
        const int pr[3] = {globalRow(P, ly0), globalRow(P, ly0 + (int)tile_h / 2), globalRow(P, ly0 + (int)tile_h - 1)};
        for (unsigned tx = 0; tx < tiles_x; tx++) {
            const double x0 = tx * (double)tile_w;
            double wmax = 0.0;
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) {
                    const double px = std::min(x0 + k * ((double)tile_w - 1.0) / 2.0, (double)P.img_w - 1.0) + 0.5;
                    const double py = std::min((double)pr[r], (double)P.img_h - 1.0) + 0.5;
                    wmax = std::max(wmax, raySamples(P, px, py));
                }
            // (a tile the geometric estimate calls active keeps a non-zero cost: the table's padding logic is about WHICH tiles exist)
            tile_work[(size_t)ty * tiles_x + tx] = (work_override && wmax > 0.0) ? std::max(wmax * (double)work_override[(size_t)ty * tiles_x + tx], 1.0e-3) : wmax;
            if (wmax > 0.0) active_tiles++;
            longest = std::max(longest, wmax);
        }
    }
    for (unsigned cy = 0; cy < cpc; cy++)
        for (unsigned cx = 0; cx < cpr; cx++) {
            double work = 0.0;
            for (unsigned v = 0; v < CH; v++)
                for (unsigned u = 0; u < CW; u++) {
                    const unsigned tx = cx * CW + u, ty = cy * CH + v;
                    if (tx < tiles_x && ty < tiles_y) work += tile_work[(size_t)ty * tiles_x + tx];
                }
            chunks.push_back({work, cx, cy});
        }
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk &a, const Chunk &b) { return a.work > b.work; });
    // chunk of rank r -> XCD r % 8, its (r / 8)-th chunk; block b runs on XCD b % 8
    const size_t nchunks = chunks.size();
    const size_t per_xcd = (nchunks + 7) / 8;
    table.assign(per_xcd * 8 * per_chunk, kTilePadding);
    for (size_t r = 0; r < nchunks; r++) {
        const size_t xcd = r % 8, pos = r / 8;
        for (unsigned t = 0; t < per_chunk; t++) {
            const unsigned tx = chunks[r].cx * CW + t % CW, ty = chunks[r].cy * CH + t / CW;
            if (tx >= tiles_x || ty >= tiles_y) continue;
            const size_t b = (pos * per_chunk + t) * 8 + xcd;
            table[b] = tx | (ty << 16);
        }
    }
    if (max_ray_samples) *max_ray_samples = longest;
    return active_tiles;
}

}  // namespace vr
