// tile_schedule.cpp -- see tile_schedule.h
#include "tile_schedule.h"

#include <algorithm>
#include <cmath>
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

uint64_t tileScheduleKey(const FrameParams &P, int rows)
{
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; }
    };
    mix(P.cam, sizeof(P.cam));
    mix(&P.img_w, sizeof(int32_t) * 9);      // img_w .. stripe_count (contiguous int32 fields)
    mix(P.pmin, sizeof(P.pmin)); mix(P.pmax, sizeof(P.pmax));
    mix(&P.step, sizeof(P.step)); mix(&P.max_steps, sizeof(P.max_steps));
    mix(&rows, sizeof(rows));
    return hsh;
}

unsigned buildTileSchedule(const FrameParams &P, int rows, std::vector<uint32_t> &table)
{
    unsigned active_tiles = 0;
    const unsigned tiles_x = (unsigned)((P.img_w + (int)kFastTileW - 1) / (int)kFastTileW);
    const unsigned tiles_y = (unsigned)((rows + (int)kFastTileH - 1) / (int)kFastTileH);
    const unsigned cpr = (tiles_x + kFastChunk - 1) / kFastChunk;
    struct Chunk { double work; unsigned cx, ty; };
    std::vector<Chunk> chunks;
    chunks.reserve((size_t)cpr * tiles_y);
    for (unsigned ty = 0; ty < tiles_y; ty++) {
        // probe rows: top, middle, bottom of the tile (global rows of this shard)
        const int ly0 = (int)(ty * kFastTileH);
        const int pr[3] = {globalRow(P, ly0), globalRow(P, ly0 + (int)kFastTileH / 2), globalRow(P, ly0 + (int)kFastTileH - 1)};
        for (unsigned cx = 0; cx < cpr; cx++) {
            double work = 0.0;
            for (unsigned t = 0; t < kFastChunk; t++) {
                const unsigned tx = cx * kFastChunk + t;
                if (tx >= tiles_x) break;
                const double x0 = tx * (double)kFastTileW;
                double wmax = 0.0;
                for (int r = 0; r < 3; r++)
                    for (int k = 0; k < 3; k++) {
                        const double px = std::min(x0 + k * ((double)kFastTileW - 1.0) / 2.0, (double)P.img_w - 1.0) + 0.5;
                        const double py = std::min((double)pr[r], (double)P.img_h - 1.0) + 0.5;
                        wmax = std::max(wmax, raySamples(P, px, py));
                    }
                work += wmax;
                if (wmax > 0.0) active_tiles++;
            }
            chunks.push_back({work, cx, ty});
        }
    }
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk &a, const Chunk &b) { return a.work > b.work; });
    // chunk of rank r -> XCD r % 8, its (r / 8)-th chunk; block b runs on XCD b % 8
    const size_t nchunks = chunks.size();
    const size_t per_xcd = (nchunks + 7) / 8;
    table.assign(per_xcd * 8 * kFastChunk, kTilePadding);
    for (size_t r = 0; r < nchunks; r++) {
        const size_t xcd = r % 8, pos = r / 8;
        for (unsigned t = 0; t < kFastChunk; t++) {
            const unsigned tx = chunks[r].cx * kFastChunk + t;
            if (tx >= tiles_x) continue;
            const size_t b = (pos * kFastChunk + t) * 8 + xcd;
            table[b] = tx | (chunks[r].ty << 16);
        }
    }
    return active_tiles;
}

}  // namespace vr
