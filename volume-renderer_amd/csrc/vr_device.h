// vr_device.h -- device-side helpers shared by the ray-march translation units
// (vr_kernels.hip, vr_tslab.hip): GLSL built-ins restated, ray set-up, the slab test, voxel
// addressing, the safe-prefix bound.  See vr_kernels.hip for the arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vr_frame.h"
#include "tile_schedule.h"
#include "vr_kernels.h"

namespace vr {

// ------------------------------------------------------------------ GLSL built-ins
__device__ __forceinline__ float gl_min(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gl_max(float x, float y) { return (x < y) ? y : x; }

// Markstein quotient: with r = RN(1/b), q0 = RN(a*r), e = a - b*q0 (exact, FMA),
// q = RN(q0 + e*r).  Correctly rounded for the divisors certify_div_kernel accepted.
__device__ __forceinline__ float div_cert(float a, float b, float r)
{
    float q0 = a * r;
    float e = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(e, r, q0);
}

template <int DIVMODE>
__device__ __forceinline__ float div_mode(float a, float b, float r)
{
    if (DIVMODE == DIV_UNIT) return a;          // b == 1.0f exactly
    if (DIVMODE == DIV_CERT) return div_cert(a, b, r);
    return a / b;
}

// ------------------------------------------------------------------ tile mapping
// Blocks are numbered so that the hardware's round-robin block->XCD placement
// (block b runs on XCD b % 8; a speed heuristic only) gives every XCD whole tile
// rows: tile row = f(b % 8, ...).  Consecutive tile rows go to different XCDs so
// the empty top/bottom of the image does not idle whole XCDs.
__device__ __forceinline__ void tile_of_block(unsigned b, unsigned tiles_x, unsigned tiles_y,
                                              unsigned &tx, unsigned &ty)
{
    const unsigned xcd = b & 7u, slot = b >> 3;
    // XCD x owns tile rows x, x+8, x+16, ...; rows_x = number of those rows
    const unsigned rows_x = (tiles_y + 7u - xcd) >> 3;
    const unsigned cap = rows_x * tiles_x;
    if (slot < cap) {
        ty = xcd + 8u * (slot / tiles_x);
        tx = slot % tiles_x;
    } else {
        tx = 0xffffffffu; ty = 0xffffffffu;   // padding block
    }
}

// ------------------------------------------------------------------ ray setup
// final store of a pixel: RGBA32F, or (grey, alpha) float2 when the caller asked for the
// half-size target of the grey modes (vr_set_framebuffer_format; r == g == b there)
__device__ __forceinline__ void store_pixel(const FrameParams &P, float4 *__restrict__ fb, size_t pix, float r, float g, float b, float a)
{
    if (P.fb_format == 1) reinterpret_cast<float2 *>(fb)[pix] = make_float2(r, a);
    else fb[pix] = make_float4(r, g, b, a);
}

struct Ray { float ox, oy, oz, dx, dy, dz; };

// VolumeRenderer.cs:194-216.  normalize() is what the GLSL compiler makes of it and the reference's GL executes
// (Mesa 23.2.1, measured by oracle/ref_gl/probe_arith.py and pinned by tests/golden/ref_gl_*.npz):
//   normalize(v) = v * (1 / sqrt(dot(v, v))),  dot summed from the LAST component to the first;
// mat4 * vec4 is the column chain ((c0*x + c1*y) + c2*z) + c3*w.  Each operation one rounding.
__device__ __forceinline__ Ray compute_ray(const FrameParams &P, float pixel_x, float pixel_y)
{
    const float *c = P.cam;
    const float fw = (float)P.img_w, fh = (float)P.img_h;
    const float aspect = (fw * 1.0f) / fh;
    const float x = aspect * (((2.0f * pixel_x) / fw) - 1.0f);
    const float y = ((2.0f * pixel_y) / fh) - 1.0f;
    const float z = -c[20];
    const float w = 0.0f;
    float rs = 1.0f / sqrtf(((w * w + z * z) + y * y) + x * x);
    const float dx = x * rs, dy = y * rs, dz = z * rs, dw = w * rs;
    const float mx = ((c[0] * dx + c[4] * dy) + c[8] * dz) + c[12] * dw;
    const float my = ((c[1] * dx + c[5] * dy) + c[9] * dz) + c[13] * dw;
    const float mz = ((c[2] * dx + c[6] * dy) + c[10] * dz) + c[14] * dw;
    const float mw = ((c[3] * dx + c[7] * dy) + c[11] * dz) + c[15] * dw;
    rs = 1.0f / sqrtf(((mw * mw + mz * mz) + my * my) + mx * mx);
    Ray r;
    r.dx = mx * rs; r.dy = my * rs; r.dz = mz * rs;
    r.ox = c[16]; r.oy = c[17]; r.oz = c[18];
    return r;
}

// VolumeRenderer.cs:218-238
__device__ __forceinline__ bool intersect_ray_aabb(const FrameParams &P, const Ray &r, float &t_min,
                                                   float &t_max)
{
    float tmax = __builtin_inff(), tmin = -__builtin_inff();
    const float ix = 1.0f / r.dx, iy = 1.0f / r.dy, iz = 1.0f / r.dz;
    const float mnx = (P.pmin[0] - r.ox) * ix, mny = (P.pmin[1] - r.oy) * iy, mnz = (P.pmin[2] - r.oz) * iz;
    const float mxx = (P.pmax[0] - r.ox) * ix, mxy = (P.pmax[1] - r.oy) * iy, mxz = (P.pmax[2] - r.oz) * iz;
    tmin = gl_max(tmin, gl_min(mnx, mxx));
    tmax = gl_min(tmax, gl_max(mnx, mxx));
    tmin = gl_max(tmin, gl_min(mny, mxy));
    tmax = gl_min(tmax, gl_max(mny, mxy));
    if (tmax < tmin) return false;
    tmin = gl_max(tmin, gl_min(mnz, mxz));
    tmax = gl_min(tmax, gl_max(mnz, mxz));
    t_min = tmin; t_max = tmax;
    return tmax > gl_max(tmin, 0.0f);
}

// ------------------------------------------------------------------ volume access
template <typename VoxelT, int LAYOUT>
__device__ __forceinline__ float fetch_voxel(const FrameParams &P, const VoxelT *__restrict__ vol, int i,
                                             int j, int k)
{
    if (LAYOUT == 0) {
        const uint64_t idx = (uint64_t)(uint32_t)i +
                             (uint64_t)(uint32_t)P.nx * ((uint64_t)(uint32_t)j + (uint64_t)(uint32_t)P.ny * (uint64_t)(uint32_t)k);
        return (float)vol[idx];
    } else {
        const uint32_t bi = (uint32_t)i >> BRICK_LX, bj = (uint32_t)j >> BRICK_LY, bk = (uint32_t)k >> BRICK_LZ;
        const uint64_t brick = (uint64_t)bi + (uint64_t)(uint32_t)P.bnx * ((uint64_t)bj + (uint64_t)(uint32_t)P.bny * (uint64_t)bk);
        const uint32_t in = ((uint32_t)i & (BRICK_X - 1u)) | (((uint32_t)j & (BRICK_Y - 1u)) << BRICK_LX) |
                            (((uint32_t)k & (BRICK_Z - 1u)) << (BRICK_LX + BRICK_LY));
        return (float)vol[brick * 64u + in];
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The TRILINEAR lerp, a + t * (b - a) as ONE fused multiply-add: fma(t, b - a, a).  This is the arithmetic of a real GL's
// linear filter -- Mesa llvmpipe's GL_LINEAR on an R32F 3-D texture (lp_build_lerp -> fmuladd), measured bit for bit through
// the reference shader with `usampler3D` -> `sampler3D` (tests/golden/ref_gl/tri_*, oracle/ref_gl/mint_ref_gl_goldens.py) --
// and what the oracle's TRILINEAR is defined as since round 5 (rounds 1-4: three separately rounded operations).  The
// compositing and every NEAREST operation stay unfused: the executed reference does not contract them.
__device__ __forceinline__ float tri_lerp(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }

// float -> int with the saturation the oracle's (int64) floor + clamp produces
__device__ __forceinline__ int floor_to_int_sat(float f)
{
    f = floorf(f);
    if (!(f > -2147483648.0f)) return (f != f) ? 0 : -2147483647 - 1;
    if (f >= 2147483648.0f) return 2147483647;
    return (int)f;
}

// Per-axis address tables (ATAB): tab[0..nx) = X, tab[nx..nx+ny) = Y, tab[nx+ny..) = Z with
// X[i] + Y[j] + Z[k] = byte offset of voxel (i,j,k) = sizeof(VoxelT) * VoxelAddr<LAYOUT,false>::at.
// PK12 (12-bit packed copy): the Y and Z element terms are multiples of BRICK_X (even), so
// floor(1.5*(x + y + z)) = floor(1.5*x) + 1.5*y + 1.5*z.  Called by all threads of a workgroup.
// BIG (volumes beyond 32-bit offsets, bricked layout only): X and Y hold ELEMENT offsets (their sum stays far
// below 2^32) and Z holds its element offset / 16 (the z term, 64*bnx*bny*(k>>2) + 16*(k&3), is a multiple of 16
// and reaches 2^33 at 2048^3): offset = X[i] + Y[j] + (uint64(Z[k]) << 4).
template <typename VoxelT, int LAYOUT, bool PK12, bool BIG = false>
__device__ __forceinline__ void build_axis_tables(const FrameParams &P, uint32_t *tab, int nthreads)
{
    static_assert(!BIG || (LAYOUT == 1 && !PK12 && BRICK_LX + BRICK_LY >= 4), "64-bit tables: bricked layout, z term a multiple of 16");
    const int na = P.nx + P.ny + P.nz;
    for (int e = (int)threadIdx.x; e < na; e += nthreads) {
        uint32_t t;
        if (e < P.nx) {
            const uint32_t i = (uint32_t)e;
            t = LAYOUT == 0 ? i : i + (64u - (uint32_t)BRICK_X) * (i >> BRICK_LX);
        } else if (e < P.nx + P.ny) {
            const uint32_t j = (uint32_t)(e - P.nx);
            t = LAYOUT == 0 ? j * (uint32_t)P.nx
                            : (BRICK_LY ? (j << BRICK_LX) + P.bstride_y * (j >> BRICK_LY) : P.bstride_y * j);
        } else {
            const uint32_t k = (uint32_t)(e - P.nx - P.ny);
            if (BIG) {
                // bstride_z is kept modulo 2^32 by the host for the 32-bit kernels: rebuild the term in 64 bits
                const uint64_t bsz = 64ull * (uint64_t)(uint32_t)P.bnx * (uint64_t)(uint32_t)P.bny - (BRICK_LZ ? 64ull : 0ull);
                const uint64_t tz = (BRICK_LZ ? ((uint64_t)k << (BRICK_LX + BRICK_LY)) + bsz * (uint64_t)(k >> BRICK_LZ) : bsz * (uint64_t)k);
                tab[e] = (uint32_t)(tz >> 4);
                continue;
            }
            t = LAYOUT == 0 ? k * (uint32_t)P.ny * (uint32_t)P.nx
                            : (BRICK_LZ ? (k << (BRICK_LX + BRICK_LY)) + P.bstride_z * (k >> BRICK_LZ) : P.bstride_z * k);
        }
        if (BIG) { tab[e] = t; continue; }
        tab[e] = PK12 ? (uint32_t)((3ull * (uint64_t)t) >> 1) : t * (uint32_t)sizeof(VoxelT);
    }
}

// Apron layout (TRILINEAR's own copy of the volume; host: RendererCore::refreshApron): 4x4x4 bricks stored as 5x4x4 --
// every brick also holds the x-neighbours of its last column (x = 4b .. 4b+4, clamped to the volume: CLAMP_TO_EDGE for
// free) -- so the two x taps of a trilinear sample are ALWAYS adjacent storage elements and one load fetches both.
// Without it a quarter of the lanes need four extra single-tap gathers per sample, and a gather costs the L1 at least
// 16 cycles however few lanes are active.  80 voxels per brick: +25 % bytes for that copy.
// byte offset of voxel (i, j, k) in the apron copy = X[i] + Y[j] + Z[k]
template <typename VoxelT>
__device__ __forceinline__ void build_axis_tables_apron(const FrameParams &P, uint32_t *tab, int nthreads)
{
    const uint32_t bnx = (uint32_t)(P.nx + 3) >> 2, bny = (uint32_t)(P.ny + 3) >> 2;
    const int na = P.nx + P.ny + P.nz;
    for (int e = (int)threadIdx.x; e < na; e += nthreads) {
        uint32_t t;
        if (e < P.nx) { const uint32_t i = (uint32_t)e; t = (i >> 2) * APRON_BRICK_VOXELS + (i & 3u); }
        else if (e < P.nx + P.ny) { const uint32_t j = (uint32_t)(e - P.nx); t = (j >> 2) * bnx * APRON_BRICK_VOXELS + (j & 3u) * 5u; }
        else { const uint32_t k = (uint32_t)(e - P.nx - P.ny); t = (k >> 2) * bnx * bny * APRON_BRICK_VOXELS + (k & 3u) * 20u; }
        tab[e] = t * (uint32_t)sizeof(VoxelT);
    }
}

// ------------------------------------------------------------------ fast kernel
// NEAREST + composite + iterative accumulation + grey ramp: the reference's own
// configuration, and the one BASELINE.json's metric is quoted on.
//
// Per ray the samples are split into
//   * a "safe" prefix of k_safe samples whose texcoords are provably inside (0,1)
//     (so the shader's six bound tests and the CLAMP_TO_EDGE clamps cannot fire),
//   * a checked tail that repeats the shader's tests literally.
// Proof sketch for the prefix (DESIGN.md "safe prefix"): the iterated position
// pos_k differs from the exact line pos_0 + k*dstep by at most k*2^-24*B, B a bound
// on |pos|; both are linear in k, so a sample range is safe iff its two end points
// lie inside the box shrunk by that error plus a fixed 16-ulp margin.
//
// The prefix is marched in batches of BATCH samples: positions do not depend on
// the voxel data, so the batch's BATCH gathers are issued back to back (memory-
// level parallelism: the loop is latency-bound otherwise) and composited in order
// afterwards, with the shader's early-termination tests between samples.  Gathers
// past an early termination are speculative reads inside the volume; they change
// neither the result nor the reported fetch count.
//
// Classification table (LUT = true): the window map + alpha scaling of a sample is a
// pure function of the integer voxel value, so each workgroup first tabulates
//   e -> (c, a) = (v*a, v*alpha_scale),  v = (float(min_val+e) - min_val)/(max_val-min_val)
// for e in [0, max_val-min_val] in LDS with the shader's own operations (so entries are
// bit-identical to the per-sample computation) and a sample then costs one ds_read_b64.
constexpr int FAST_LUT_MAX = 4096;      // entries (x 8 B = 32 KiB of the CU's 160 KiB LDS)
// MODE 2 (transfer function) is two-level: 256 premultiplied RGBA entries (4 KiB) + one index
// byte per windowed voxel value in the remaining 28 KiB, so any window up to 28672 values fits
constexpr int FAST_TF_ENTRIES = 256;
static_assert(FAST_TF_WINDOW_MAX == FAST_LUT_MAX * 8 - FAST_TF_ENTRIES * 16, "vr_frame.h: FAST_TF_WINDOW_MAX");
// Address tables (ATAB): the byte offset of voxel (i,j,k) is X[i] + Y[j] + Z[k] in both layouts,
// so the ~10 integer VALU ops of VoxelAddr become three LDS look-ups and one add; integer ops
// issue at ~1.6x the cost of fp32 ops on gfx950 and are 40 % of the inner loop's issue time.
constexpr int FAST_AXIS_TAB_MAX = 3072; // entries: nx + ny + nz (x 4 B = 12 KiB)
// the same for volumes beyond 32-bit offsets (2048^3 = 6144 entries, 24 KiB); their classification table is
// cut to 8 KiB -- 1024 (c,a) entries, or the 256 RGBA entries + 4096 index bytes -- so that three workgroups
// still fit a CU
constexpr int FAST_AXIS_TAB_BIG_MAX = 6144, FAST_BIG_LUT_FLOATS = 2048;

__device__ __forceinline__ int med3_i32(int a, int b, int c)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}

// element offset of voxel (i,j,k).  BIG = false: 32-bit arithmetic with 24-bit
// multiplies (host guarantees every factor < 2^24 and storage voxels < 2^32).
template <int LAYOUT, bool BIG>
struct VoxelAddr {
    using type = typename std::conditional<BIG, uint64_t, uint32_t>::type;
    __device__ static __forceinline__ type at(const FrameParams &P, int i, int j, int k)
    {
        if (LAYOUT == 0) {
            if (BIG)
                return (type)((uint64_t)(uint32_t)i + (uint64_t)(uint32_t)P.nx * ((uint64_t)(uint32_t)j + (uint64_t)(uint32_t)P.ny * (uint64_t)(uint32_t)k));
            const uint32_t row = mad_u24((uint32_t)k, (uint32_t)P.ny, (uint32_t)j);
            return (type)mad_u24(row, (uint32_t)P.nx, (uint32_t)i);
        } else {
            if (BIG) {
                const uint64_t brick = (uint64_t)((uint32_t)i >> BRICK_LX) + (uint64_t)(uint32_t)P.bnx * ((uint64_t)((uint32_t)j >> BRICK_LY) + (uint64_t)(uint32_t)P.bny * (uint64_t)((uint32_t)k >> BRICK_LZ));
                return (type)(brick * 64u + (((uint32_t)i & (BRICK_X - 1u)) | (((uint32_t)j & (BRICK_Y - 1u)) << BRICK_LX) |
                                             (((uint32_t)k & (BRICK_Z - 1u)) << (BRICK_LX + BRICK_LY))));
            }
            // (i & (BX-1)) + 64*(i>>LX) = i + (64-BX)*(i>>LX); likewise j (scaled by BX) and k
            // (scaled by BX*BY):
            //   offset = [i + BX*j + BX*BY*k] + (64-BX)*(i>>LX) + bstride_y*(j>>LY) + bstride_z*(k>>LZ)
            // = 2 shift-adds + 3 shifts + 3 chained 24-bit mads; an axis the brick does not
            // split (L == 0) is one mad with the full stride
            uint32_t t = (uint32_t)i;
            if (BRICK_LY) t = ((uint32_t)j << BRICK_LX) + t;
            if (BRICK_LZ) t = ((uint32_t)k << (BRICK_LX + BRICK_LY)) + t;
            t = mad_u24((uint32_t)i >> BRICK_LX, 64u - (uint32_t)BRICK_X, t);
            t = mad_u24(BRICK_LY ? (uint32_t)j >> BRICK_LY : (uint32_t)j, P.bstride_y, t);
            return (type)mad_u24(BRICK_LZ ? (uint32_t)k >> BRICK_LZ : (uint32_t)k, P.bstride_z, t);
        }
    }
};

// voxel fetch: buffer load with a 32-bit offset (hardware bounds check) when the volume
// is below 4 GiB, plain global load otherwise
template <typename VoxelT, bool BIG>
struct VoxelFetch {
    __device__ static __forceinline__ uint32_t load(const VoxelT *__restrict__ vol, __amdgpu_buffer_rsrc_t rs, typename std::conditional<BIG, uint64_t, uint32_t>::type off)
    {
        if (BIG) return (uint32_t)vol[off];
        if (sizeof(VoxelT) == 1) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off, 0, 0);
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(off << 1), 0, 0);
    }
};

// Number of leading samples of a ray whose texcoords are provably inside (0,1)^3 (see the
// fast kernel's header): q = first sample position, ds = per-sample step, box units.
// budget = samples the shader's loop has left for this ray (max_steps minus the checked head's samples)
__device__ __forceinline__ int safe_prefix_length(const FrameParams &P, float qx, float qy, float qz, float dsx,
                                                  float dsy, float dsz, int budget)
{
    // B bounds |pos| for every sample that is still inside the box
    const float hm = fmaxf(fmaxf(P.half[0], P.half[1]), P.half[2]);
    const float B = hm + fmaxf(fmaxf(fabsf(dsx), fabsf(dsy)), fabsf(dsz)) + 1e-3f;
    const float e = B * 1.1920929e-7f;          // 2 * 2^-24 * B per step
    const float base = hm * 9.5367432e-7f;      // 16 * 2^-24 * hm fixed margin
    float kmax = (float)P.max_steps;
    bool ok = true;
    const float q[3] = { qx, qy, qz }, ds[3] = { dsx, dsy, dsz };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lim = P.half[a] - base;     // shrunk half extent
        // position after k steps must satisfy  q + k*ds + k*e <= lim  and  q + k*ds - k*e >= -lim
        ok = ok && (q[a] <= lim) && (q[a] >= -lim) && (lim > 0.0f);
        const float up = ds[a] + e;             // > 0: upper face approached
        const float dn = ds[a] - e;             // < 0: lower face approached
        if (up > 0.0f) kmax = fminf(kmax, (lim - q[a]) / up);
        if (dn < 0.0f) kmax = fminf(kmax, (-lim - q[a]) / dn);
    }
    int k_safe = 0;
    if (ok && kmax > 4.0f) k_safe = (int)(kmax * 0.999f) - 2;
    if (!(k_safe > 0)) k_safe = 0;
    if (k_safe > budget) k_safe = budget;
    return k_safe > 0 ? k_safe : 0;
}

// The first sample of a ray sits EPSILON * |dir| = 1e-6 * |dir_a| inside the face it entered through -- closer to it
// than the prefix's fixed margin (16 ulps of the half extent) whenever |dir_a| < ~0.48, i.e. for about half of the rays
// of an oblique view: safe_prefix_length() then refuses the whole ray and it marches sample by sample through the
// checked loop.  head_steps() = how many leading samples have to take the checked loop so that the position is inside
// the shrunk box (normally 1: one step is ~1e-3 * |dir_a|); the safe prefix is then computed from the sample after them.
// 0 = the ray starts inside already, or it never gets there within HEAD_STEPS_MAX samples (then nothing changes).
constexpr int HEAD_STEPS_MAX = 16;
__device__ __forceinline__ int head_steps(const FrameParams &P, float qx, float qy, float qz, float dsx, float dsy, float dsz)
{
    const float hm = fmaxf(fmaxf(P.half[0], P.half[1]), P.half[2]);
    const float B = hm + fmaxf(fmaxf(fabsf(dsx), fabsf(dsy)), fabsf(dsz)) + 1e-3f;
    const float e = B * 1.1920929e-7f, base = hm * 9.5367432e-7f;       // as in safe_prefix_length
    const float q[3] = { qx, qy, qz }, ds[3] = { dsx, dsy, dsz };
    float need = 0.0f;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lim = P.half[a] - base;
        if (q[a] > lim) {                       // above the shrunk box: must move down by more than the error grows
            const float v = ds[a] + e;
            ok = ok && v < 0.0f;
            need = fmaxf(need, (lim - q[a]) / v);
        } else if (q[a] < -lim) {
            const float v = ds[a] - e;
            ok = ok && v > 0.0f;
            need = fmaxf(need, (-lim - q[a]) / v);
        }
    }
    if (!ok || !(need >= 0.0f) || need > (float)(HEAD_STEPS_MAX - 2)) return 0;
    return need > 0.0f ? (int)need + 2 : 0;     // + 1 rounds up, + 1 more sample of slack
}

// Size for the buffer descriptor of kernels that fetch an x-neighbour pair with one load: a
// pair load on the LAST voxel of the buffer reaches 2 (u16) / 1 (u8) bytes past it, and a raw
// buffer load that is partly out of range returns 0 for ALL of it.  The allocation carries
// slack behind the volume (renderer_core.cpp: allocVolume), so the descriptor may cover 4
// bytes more.
__device__ __forceinline__ uint32_t pair_load_extent(uint32_t vol_bytes)
{
    return vol_bytes > 0xfffffffbu ? vol_bytes : vol_bytes + 4u;
}

}  // namespace vr
