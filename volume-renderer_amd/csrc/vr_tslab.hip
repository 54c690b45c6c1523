// vr_tslab.hip -- TRILINEAR ray-march with the volume's bricks staged in LDS, for gfx950 (MI355X / CDNA4).
//
// Same arithmetic as every other trilinear path of this library (GL's linear rule on the shader's sample positions,
// VolumeRenderer.cs:121 with the filter state of src/RendererCore.cpp:414-415; fixed lerp order x, y, z, each lerp one fma, as in
// oracle/vr_oracle.c and as Mesa llvmpipe's GL_LINEAR computes it; iterative position accumulation; checked head + safe prefix + checked tail, vr_device.h).
// What changes is where the prefix's eight taps per sample come from, and in which ORDER a workgroup's rays advance.
//
// Why: the batched trilinear kernel (vr_kernels.hip) issues 4-8 scattered global gathers per sample; on gfx950 the
// vector L1 pays 4 cycles per distinct cache line per gather instruction, and that alone costs more than the kernel's
// arithmetic (round 2: 843 M tag look-ups, 1.93 ms on the 1024^3 workload).  Here a workgroup (8 wavefronts = a 32x16-pixel
// tile) keeps the bricks its rays are crossing in LDS and reads the taps with ds_read_u16 / ds_read_u8:
//   * staged source = the APRON copy of the volume (vr_device.h: every 4x4x4 brick stored as 5x4x4, the x neighbours of
//     its last column included, clamped at the volume's faces): a slot is 160 B (u16) / 80 B (u8); the x1 tap of every
//     corner pair is the x0 tap's next element (a ds_read immediate offset), so a sample needs FOUR tap addresses --
//     X[i0] + Y[j0|j0+1] + Z[k0|k0+1] -- out of five look-ups in per-axis torus tables; 160-byte slots also spread the
//     bricks of one sheet over the LDS banks (128-byte slots: 2-3x the conflicts, tools/ubench/lds_taps.hip);
//   * bricks arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR staging), whole brick
//     LAYERS along the tile's major axis m (the voxel axis the rays advance along fastest); a layer lives in slot
//     (i mod RA, j mod RB, L mod RZ) of a torus, so a voxel's LDS offset stays a sum of three table entries;
//   * LAYER-SYNCHRONOUS marching: phase p of the workgroup is brick layer L_p = L_0 + sgn * p.  In phase p every ray
//     takes exactly the samples whose cell (k0 = floor(f_m - 0.5)) lies in layer L_p -- 4-5 samples at one voxel per
//     step, as many as it takes for oblique rays -- then waits at ONE barrier.  All rays of the tile therefore read
//     layers L_p and L_p + 1 only, whatever their directions: the ring needs 2 live layers + 1-2 being loaded (RZ = 3
//     or 4) instead of the 5-6 a sample-synchronous scheme needs, and an oblique view costs one layer-thickness of
//     drift in the footprint, not a whole oblique sheet (round 2's sample-synchronous NEAREST kernel, vr_slab.hip,
//     staged 4 % of an off-axis frame; the first trilinear build on it fell back to global taps in 27 % of its epochs);
//   * the footprint of layer L is CLOSED FORM: in voxel coordinates a ray is the line E + t * G (E = eye, one G per
//     pixel), and at the plane f_m = c its other two coordinates are fractional-linear in the pixel position, so over
//     the tile's pixel rectangle their extremes sit at the four corner pixels.  Thread L evaluates the four corner
//     rays at the two planes bounding layer L's samples and publishes the brick rectangle in LDS: no anchors, no
//     epochs, no per-phase reductions.  Margins: half a voxel for the taps, 1/16 voxel + the rounding drift of the
//     iterated positions (k * 2^-23 * N voxels) for the difference between the lines and the marched positions;
//   * no clamps in the prefix: u = max(f - 0.5, 0) makes the low edge exact (cell 0 with weight 0 == the shader's two
//     clamped taps of voxel 0), the high edge is the apron column (x) or a duplicated last table entry (y, z).
//   * round 4, 16-bit volumes (PERM instances): HALF LAYERS.  An oblique tile's layers are 240-400 bricks as rectangles
//     (tools/model/tslab_footprint2.py); three of them at 160 B a slot do not fit 80 KiB, and the whole LDS of a CU for
//     one 8-wavefront workgroup costs 1.5x in issue rate (1.25 -> 1.86 ms at the default pose).  So a tile whose whole
//     layers do not fit marches in layers TWO voxels thick: a slot is half a brick (80 B), the ring holds three
//     300-slot half layers in 80 KiB, twice the phases.  Half a brick must be contiguous for the DMA, i.e. the brick's
//     planes along the MAJOR axis must be its slowest dimension: one apron copy per major axis (order 0: x fastest with
//     the apron, y, z slowest = the copy above; order 1: x, z, y slowest; order 2: y fastest with the apron, z, x
//     slowest), built on first use.  The pair of taps one LDS address yields then lies along the tile's first minor axis
//     (x, or y when the rays advance along x); the lerp order stays x, y, z.  The thickness is chosen per tile: whole
//     layers where they fit (fewer barriers), half layers otherwise.
// A tile whose corner rays disagree on the major axis or its sign, or whose layers do not fit the ring three deep,
// marches its prefix on global taps (same arithmetic, slower); head and tail always do.
// Correctness does not depend on the rectangles being tight, only on their being supersets; the parity tests compare
// whole frames bit for bit against the oracle and the other trilinear kernels.
//
// No MFMA: eight taps, seven lerps and a 5-flop recurrence per sample.
#include <cstdlib>

#include "vr_device.h"
#include "vr_lds_dma.h"

namespace vr {

// Workgroup shapes (template parameters NW = wavefronts, LDSKB = KiB of LDS the workgroup declares, TW = wavefronts per tile row):
//   NW = 8, LDSKB = 80 is what ships: one 32x16- or 16x32-pixel tile per workgroup, two workgroups per CU.  The ring holds three
//   144-slot layers of 160 B for the 1024^3 u16 workload at 1080p; u8 slots are half as large, but a 2048^3 volume's plan + tables
//   take 20 KiB.  NW = 8, LDSKB = 53 ships too (round 4, tri_slab 5): THREE workgroups per CU, six wavefronts per SIMD at 80 VGPRs (43
//   dwords spilled, none in the sample loop).  In round 3 it measured the same time as two (512^3: 0.711 vs 0.713 ms, the loop was
//   bound by VALU issue); with round 4's cheaper loop the extra wavefronts cover the LDS round trips: 512^3 u16 0.520 -> 0.452 ms,
//   1024^3 u8 1.061 -> 0.932, the cfg2 shape 0.445 -> 0.417 -- where a tile's layers fit 53 KiB; where they do not (1024^3 u16:
//   1.14 -> 1.84) the work model keeps the 80-KiB shapes.  ONE workgroup per CU (8 wavefronts on the whole 160 KiB) is 1.5x slower (1.25 -> 1.86 ms): two wavefronts
//   per SIMD do not cover each other's LDS round trips.  16 wavefronts on a 32x32-pixel tile and the whole LDS: + 10 % at the
//   default pose, and a frame is three rounds of long tiles.  Both whole-LDS shapes were measured with whole and half layers
//   (round 4) and are not instantiated any more; the parameters stay.
constexpr float TS_MARGIN = 0.0625f;                      // voxels, on top of the drift bound
constexpr int TS_FB_BATCH = 4;                            // samples whose taps a tile that is not staged requests together (six or eight: no faster)
constexpr float TS_MIN_AXIS = 0.3f;                       // |G_m| >= this * |G|_inf at all four corners, else not staged

template <typename VoxelT, int MODE, int NW, int LDSKB, bool PERM, int TW = 4>
struct TslabCfg {
    static constexpr int THREADS = 64 * NW;                                       // TW x NW/TW wavefronts of 8x8 pixels
    static constexpr int TILE_W = 8 * TW, TILE_H = 8 * (NW / TW);
    static constexpr int WGS_PER_CU = LDSKB <= 26 ? 6 : (LDSKB <= 32 ? 5 : (LDSKB <= 40 ? 4 : (LDSKB <= 53 ? 3 : (LDSKB <= 80 ? 2 : 1))));   // what 160 KiB hold
    static constexpr int WAVES_PER_SIMD = WGS_PER_CU * NW / 4;
    static constexpr int BRICK_BYTES = (int)APRON_BRICK_VOXELS * (int)sizeof(VoxelT);   // a brick of the apron copies: 80 B (u8) / 160 B (u16)
    static constexpr int LUT_BYTES = MODE >= 2 ? 4096 : 16;                       // 256 premultiplied RGBA entries
    static constexpr int MISC_BYTES = 512;
    // one region carved per tile into [plan | torus tables | ring]: the tables' size follows the tile's own index ranges
    // (2 bytes per voxel index of the two minor axes, 4 per index of the major axis)
    static constexpr int REGION = (LDSKB * 1024 - 512 - LUT_BYTES - MISC_BYTES) / 16 * 16;
    // 1-KiB DMA pieces per wavefront per layer: what three layers of the ring can hold (80 KiB: u16 153 slots of 160 B,
    // u8 409 of 80 B); a third of the region in 16-byte chunks whatever the slot size
    static constexpr int MAX_PIECES = (NW == 8 && LDSKB == 80 && !PERM) ? (sizeof(VoxelT) == 1 ? 4 : 3) : (REGION / 3 / 16 + THREADS - 1) / THREADS;
    // half layers (two voxels thick, half-brick slots): per-major-axis copies, and a half brick must be whole 16-byte chunks
    static constexpr bool HALF_OK = PERM && (BRICK_BYTES / 2) % 16 == 0;
};

#if defined(VR_EXPERIMENTS) && defined(VR_X_STATS)   // per-tile statistics build (tools/tslab_stats.py)
#define VR_TSLAB_STAT(...) __VA_ARGS__
#else
#define VR_TSLAB_STAT(...)
#endif
// Checked build (make TSLAB_TAG=_chk TSLAB_DEFS=-DVR_TSLAB_CHECK -> lib/libvr_core_chk.so, tests/test_tslab_plan_guard.py): the
// staged path's correctness rests on every tap's brick lying inside the rectangle its layer was planned and loaded with -- a
// brick outside it would alias another slot of the torus silently.  This build tests that for every staged sample (the four
// corner pairs: brick inside the planned rectangle of a layer that is resident in this phase) and reports the number of
// violations per pixel in place of the fetch count: 0x40000000 | staged << 29 | min(violations, 0xffff).
#if defined(VR_TSLAB_CHECK)
#define VR_TSLAB_CHK(...) __VA_ARGS__
#else
#define VR_TSLAB_CHK(...)
#endif

// PERM: per-major-axis apron copies (src = order 0, src_y = order 1, src_x = order 2), the tap pair along the first minor axis,
// layer thickness 4 or 2 per tile; otherwise the order-0 copy for every tile, pairs along x, whole layers
// TW: wavefronts per tile row -- 4: 32x16-pixel tiles (32x32 with 16 wavefronts); 2: 16x32-pixel tiles, whose brick rectangles are
// a fifth smaller where the view's shear runs along the image's x direction (views near a body diagonal of the volume)
// SKIP (round 5): exact empty-space skipping (vr_set_skip_empty; FrameParams::skip_empty / skip_thresh, the dilated cell-max grid of
// the NEAREST kernels) at the granularity this kernel works in -- a tile's brick LAYER: a layer whose rectangle lies in cells that
// classify to exactly (0,0,0,0) is not requested from HBM, and the phase that would sample it only steps the rays through it
// (the shader's position additions, nothing else).  Frames and per-pixel sample counts are bit-identical with and without it.
template <typename VoxelT, int DIVTC, int VIEW, bool POW2, int MODE, int NW, int LDSKB, bool PERM, int TW = 4, bool SKIP = false>
__global__ __launch_bounds__(64 * NW, (TslabCfg<VoxelT, MODE, NW, LDSKB, PERM, TW>::WAVES_PER_SIMD)) void raymarch_tslab_kernel(const FrameParams P,
                                                                       const VoxelT *__restrict__ vol,
                                                                       const uint8_t *__restrict__ src,
                                                                       const uint8_t *__restrict__ src_y,
                                                                       const uint8_t *__restrict__ src_x,
                                                                       const float4 *__restrict__ tf,
                                                                       float4 *__restrict__ fb,
                                                                       uint32_t *__restrict__ spp,
                                                                       const uint32_t *__restrict__ tile_table,
                                                                       const int no_stage,
                                                                       const uint16_t *__restrict__ skip_grid)
{
    using C = TslabCfg<VoxelT, MODE, NW, LDSKB, PERM, TW>;
    // rows (the ring's rows with their own brick ranges, below): compiled into the 16x32-pixel shapes only -- the views that need rows are
    // the ones those tiles suit, and the other shapes keep their leaner phase loop (with the rows' loader and table producer compiled in,
    // tiles that never use them run 2-4 % slower: measured on every shape)
    constexpr bool ROWS = TW == 2 && NW == 8;
    constexpr int TS_NW = NW, TS_THREADS = C::THREADS;
    __shared__ __attribute__((aligned(16))) uint8_t ring[C::REGION];
    __shared__ __attribute__((aligned(16))) float lut[C::LUT_BYTES / 4];
    __shared__ float corner[5][4];              // rows 0..3: G of the corner rays, row 4: E (voxel coordinates)
    __shared__ int red[12];                     // 0: min first progress, 1: max last progress, 2: max prefix length, 3 / 4: max rectangle extents,
                                                // 5: clamp flag, 6..9: min / max brick index of the live rectangles along a and b
    // checked build: one bit per planned layer that was REQUESTED (issue_layer) -- with empty-space skipping a layer is fetched only
    // when some sampling phase reads it, and check_taps() holds every staged tap to "its layer was asked for"
    VR_TSLAB_CHK(constexpr int CHK_REQ_WORDS = 64; __shared__ uint32_t chk_req[CHK_REQ_WORDS];)
    static_assert(sizeof(corner) + sizeof(red) VR_TSLAB_CHK(+ 64 * 4) <= C::MISC_BYTES, "LDS budget");

    const uint32_t t = tile_table[blockIdx.x];
    if (t == 0xffffffffu) return;                                       // padding block
    VR_TSLAB_STAT(const uint64_t st_entry = clock64();)
    const unsigned tx = t & 0xffffu, ty = t >> 16;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(tx * (unsigned)C::TILE_W + (wave % (unsigned)TW) * 8u + (lane & 7u));
    const int ly = (int)(ty * (unsigned)C::TILE_H + (wave / (unsigned)TW) * 8u + (lane >> 3));
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int st = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (st * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    const bool in_image = !(px >= P.col_lim || py >= P.row_lim || py >= P.row_end);

    // every thread has a ray (the tile's corner pixels bound the footprint even when they lie outside the image)
    const Ray ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
    float t_min = 0.0f, t_max = 0.0f;
    const bool hit = in_image && intersect_ray_aabb(P, ray, t_min, t_max);
    if (!__syncthreads_or(hit ? 1 : 0)) {                               // no ray of the tile enters the volume
        if (in_image) {
            const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
            store_pixel(P, fb, pix, 0.0f, 0.0f, 0.0f, 0.0f);
            if (spp) spp[pix] = 0u;
        }
        return;
    }
    if (MODE >= 2) {                                                     // 256 premultiplied RGBA entries, as in the fast kernel
        for (int e = (int)threadIdx.x; e < P.tf_len; e += TS_THREADS) {
            const float4 q = tf[e];
            const float a = q.w * P.alpha_scale;
            if (MODE == 3) { lut[4 * e + 0] = q.x * P.alpha_scale; lut[4 * e + 1] = q.y * P.alpha_scale; lut[4 * e + 2] = q.z * P.alpha_scale; }
            else { lut[4 * e + 0] = q.x * a; lut[4 * e + 1] = q.y * a; lut[4 * e + 2] = q.z * a; }
            lut[4 * e + 3] = a;
        }
        __syncthreads();                                                 // the checked head below classifies through the table
    }

    float drgb = 0.0f, dg = 0.0f, db = 0.0f, da = 0.0f;
    const float EPSILON = 0.000001f;
    const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
    float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
    const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
    // checked head (vr_device.h: head_steps): positions stepped here, sampled through the checked loop below
    const float hqx = qx, hqy = qy, hqz = qz;
    const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
    for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
    const int k_safe = hit ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;

    // marching units: voxels for POW2 (exact, see raymarch_fast_kernel), box units otherwise
    const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
    float Qx = qx * Sx, Qy = qy * Sy, Qz = qz * Sz;
    const float dSx = dsx * Sx, dSy = dsy * Sy, dSz = dsz * Sz;
    // Wave-uniform constants of the sample loop are kept in VECTOR registers on purpose: on gfx950 a fp32 add / mul / fma
    // with an SGPR operand issues at the slow rate (4.2 cycles per wavefront instruction instead of 2.3,
    // tools/ubench/valu_rates2.hip), and the loop is bound by VALU issue.
    auto in_vgpr = [](float v) { asm volatile("" : "+v"(v)); return v; };
    const float Hx = in_vgpr(P.half[0] * Sx), Hy = in_vgpr(P.half[1] * Sy), Hz = in_vgpr(P.half[2] * Sz), Szv = in_vgpr(Sz), Syv = in_vgpr(Sy);
    const float k_magic_f = in_vgpr(8388608.0f);
    uint32_t k_magic_u = 0x4B000000u;
    asm volatile("" : "+v"(k_magic_u));
    const float k_fmin = in_vgpr(P.fmin), k_fmax = in_vgpr(P.fmax), k_fden = in_vgpr(P.fden), k_rden = in_vgpr(P.rden), k_alpha = in_vgpr(P.alpha_scale);

    // voxel coordinates (floats, per voxel axis i/j/k) of a box position: an affine map, exact for POW2 and within a
    // few ulp otherwise -- only the load plan uses it, behind its margins
    auto voxel_float = [&](float ax, float ay, float az, float &fx, float &fy, float &fz) {
        const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
        const float uzr = (az + P.half[2]) * P.rext[2], uz = 1.0f - uzr;
        float tcx = ux, tcy = uy, tcz = uz;
        if (VIEW == 1) { tcy = uzr; tcz = uy; }
        else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
        fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
    };
    // scaled texcoord (texcoord * dimension per voxel axis) of the CURRENT position, the shader's operations
    auto scaled_here = [&](float &fx, float &fy, float &fz) {
        if (POW2) {
            const float ux = Qx + Hx, uy = Qy + Hy, uzr = Qz + Hz, uz = Szv - uzr;
            fx = ux; fy = uy; fz = uz;
            if (VIEW == 1) { fy = uzr; fz = uy; }
            else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
        } else {
            const float ux = div_mode<DIVTC>(qx + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(qy + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(qz + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
        }
    };
    auto advance = [&]() {
        if (POW2) { Qx += dSx; Qy += dSy; Qz += dSz; }
        else { qx += dsx; qy += dsy; qz += dsz; }
    };

    // ---- TRILINEAR from the resident volume (head, tail, tiles that are not staged): the shader's clamped taps, literally
    const int nxm1 = P.nx - 1, nym1 = P.ny - 1, nzm1 = P.nz - 1;
    auto taps_global = [&](float fx, float fy, float fz, uint32_t *tv, float &ax, float &ay, float &az) {
        const float u = fx - 0.5f, v = fy - 0.5f, w = fz - 0.5f;
        const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
        ax = u - fu; ay = v - fv; az = w - fw;
        const int iu = (int)fu, iv = (int)fv, iw = (int)fw;
        const int i0 = med3_i32(iu, 0, nxm1), i1 = med3_i32(iu + 1, 0, nxm1);
        const int j0 = med3_i32(iv, 0, nym1), j1 = med3_i32(iv + 1, 0, nym1);
        const int k0 = med3_i32(iw, 0, nzm1), k1 = med3_i32(iw + 1, 0, nzm1);
        tv[0] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i0, j0, k0)]; tv[1] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i1, j0, k0)];
        tv[2] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i0, j1, k0)]; tv[3] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i1, j1, k0)];
        tv[4] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i0, j0, k1)]; tv[5] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i1, j0, k1)];
        tv[6] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i0, j1, k1)]; tv[7] = (uint32_t)vol[VoxelAddr<1, true>::at(P, i1, j1, k1)];
    };
    // interpolation (x, then y, then z), window and classification of one sample from its eight taps
    // tv[0..7] = (x0, x1) of the (y0,z0), (y1,z0), (y0,z1), (y1,z1) corners -- the generic kernel's operations
    const float tf_scale = (float)(P.tf_len - 1);
    // window + classification of one interpolated sample (VolumeRenderer.cs:122-131 / :164)
    auto classify = [&](float sv, float &c, float &cg, float &cb, float &a) {
        sv = __builtin_amdgcn_fmed3f(sv, k_fmin, k_fmax);                // == min(max(sv, fmin), fmax): fmin <= fmax, never NaN here; one instruction
        // (the clamp moved behind the division as the last fma's clamp modifier -- below the window the quotient is negative, above
        // it >= 1 -- is bit-exact too (20 000-trial campaign) and saves the v_med3, but measured no faster: 1.1465 vs 1.1428 ms)
        sv = div_cert(sv - k_fmin, k_fden, k_rden);
        if (MODE >= 2) {
            int idx = (int)(sv * tf_scale + 0.5f);                       // sv in [0, 1]: truncation == floor
            idx = med3_i32(idx, 0, P.tf_len - 1);
            const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
            c = q.x; cg = q.y; cb = q.z; a = q.w;
        } else {
            a = sv * k_alpha;
            c = sv * a;
        }
    };
    auto shade = [&](const uint32_t *tv, float ax, float ay, float az, float &c, float &cg, float &cb, float &a) {
        const float c000 = (float)tv[0], c100 = (float)tv[1], c010 = (float)tv[2], c110 = (float)tv[3];
        const float c001 = (float)tv[4], c101 = (float)tv[5], c011 = (float)tv[6], c111 = (float)tv[7];
        const float c00 = tri_lerp(c000, c100, ax), c10 = tri_lerp(c010, c110, ax);
        const float c01 = tri_lerp(c001, c101, ax), c11 = tri_lerp(c011, c111, ax);
        const float c0 = tri_lerp(c00, c10, ay), c1 = tri_lerp(c01, c11, ay);
        classify(tri_lerp(c0, c1, az), c, cg, cb, a);
    };
    auto accumulate = [&](float c, float cg, float cb, float a) {
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
    int i = 0;
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
        uint32_t tv[8];
        float ax, ay, az, c, cg = 0.0f, cb = 0.0f, a;
        taps_global(tcx * P.fdim[0], tcy * P.fdim[1], tcz * P.fdim[2], tv, ax, ay, az);
        shade(tv, ax, ay, az, c, cg, cb, a);
        accumulate(c, cg, cb, a);
        x += stx; y += sty; z += stz;
        return false;
    };
    // (round 6, measured and not kept: head and tail in batches of two iterations with their 16 fetches in flight together, as in the
    // fast and relay kernels' checked_batch() -- cfg1 shape 0.1176 -> 0.1179 ms, cfg2 shape 0.3799 -> 0.3805, cfg3 1.116 -> 1.120: a tile's
    // wavefronts wait at the phase barriers, not on these round trips)
    bool head_ended = false;
    if (head > 0) {                                                      // samples 0 .. head - 1
        float x = hqx, y = hqy, z = hqz;
        for (int h = 0; h < head && !head_ended; h++) {
            if (i >= P.max_steps || checked_step(x, y, z, dsx, dsy, dsz)) head_ended = true;
            else i++;
        }
    }
    bool done = head_ended;
    int rem = head_ended ? 0 : k_safe;                                   // prefix samples still to take

    // ================================================================== the tile's load plan
    // corner rays in voxel coordinates: line E + t * G
    const bool is_corner = (wave == 0 && lane == 0) || (wave == TW - 1 && lane == 7) || (wave == NW - TW && lane == 56) || (wave == NW - 1 && lane == 63);
    if (is_corner) {
        const int cidx = (wave >= NW - TW ? 2 : 0) + ((wave % (unsigned)TW) == (unsigned)(TW - 1) ? 1 : 0);
        float ex, ey, ez, gx, gy, gz;
        voxel_float(ray.ox, ray.oy, ray.oz, ex, ey, ez);
        voxel_float(ray.ox + ray.dx, ray.oy + ray.dy, ray.oz + ray.dz, gx, gy, gz);
        corner[cidx][0] = gx - ex; corner[cidx][1] = gy - ey; corner[cidx][2] = gz - ez;
        if (cidx == 0) { corner[4][0] = ex; corner[4][1] = ey; corner[4][2] = ez; }
    }
    __syncthreads();
    float G[4][3], E[3];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int a = 0; a < 3; a++) G[c][a] = uniform_f(corner[c][a]);
#pragma unroll
    for (int a = 0; a < 3; a++) E[a] = uniform_f(corner[4][a]);
    // major axis: where corner 0 advances fastest; all four corners must agree on its sign and not graze it
    int ax_m = 2;
    {
        const float g0 = fabsf(G[0][0]), g1 = fabsf(G[0][1]), g2 = fabsf(G[0][2]);
        ax_m = (g0 >= g1 && g0 >= g2) ? 0 : (g1 >= g2 ? 1 : 2);
    }
    const int ax_a = ax_m == 0 ? 1 : 0, ax_b = ax_m == 2 ? 1 : 2;
    bool stage = no_stage != 1;                                          // (vr_set_kernel_variant 7: every tile on global taps, the cross-check of that path)
    int sgn = 1;
    {
        bool pos = true, neg = true;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float gm = sel3(ax_m, G[c][0], G[c][1], G[c][2]);
            const float gmax = fmaxf(fmaxf(fabsf(G[c][0]), fabsf(G[c][1])), fabsf(G[c][2]));
            pos = pos && gm >= TS_MIN_AXIS * gmax && gm > 0.0f;
            neg = neg && -gm >= TS_MIN_AXIS * gmax && gm < 0.0f;
        }
        stage = stage && (pos || neg);
        sgn = neg ? -1 : 1;
    }
    // the apron copy this tile's bricks stream from: stored with the major axis slowest (PERM), or the order-0 copy
    const uint8_t *const src_m = PERM ? (ax_m == 2 ? src : (ax_m == 1 ? src_y : src_x)) : src;
    const int nbr0 = P.bnx, nbr1 = P.bny, nbr2 = P.bnz;                // bricks per voxel axis
    const int nbr_m = sel3(ax_m, nbr0, nbr1, nbr2), nbr_a = sel3(ax_a, nbr0, nbr1, nbr2), nbr_b = sel3(ax_b, nbr0, nbr1, nbr2);
    const int ndim_m = sel3(ax_m, P.nx, P.ny, P.nz);
    const int ndim_a = sel3(ax_a, P.nx, P.ny, P.nz), ndim_b = sel3(ax_b, P.nx, P.ny, P.nz);
    // margin between the corner lines and the marched positions: rounding drift of k iterated additions (<= k * 2^-24 * |pos|
    // per axis, |pos| in voxels <= N) + the evaluation error of the lines themselves (eye far away: |E| * 2^-22)
    const float nmax = fmaxf(fmaxf(P.fdim[0], P.fdim[1]), P.fdim[2]);
    const float emax = fmaxf(fmaxf(fabsf(E[0]), fabsf(E[1])), fabsf(E[2]));

    // Everything below depends on the layer thickness T = 1 << LSH voxels (4: whole bricks; 2: half bricks, PERM instances
    // with 16-byte-divisible half bricks only): the plan is made for whole layers first and, when they do not fit the ring
    // three deep, once more for half layers.
    int LSH = 2;
    int slot_b = C::BRICK_BYTES;                                         // bytes per ring slot: a brick or half a brick
    int c_first = 0, c_last = 0, kmax = 0, n_phases = 0, L0 = 0, Llo = 0, Lhi = -1, plan_bytes = 0;
    bool any_prefix = false;
    float delta = 0.0f;
    uint8_t *plan_b = nullptr;                                           // plan entry of layer L at plan_b + (L - Llo) * plan_stride
    int plan_stride = 8;
    // ROWS (round 4, last): the ring's rows (along b) may hold their own brick ranges instead of the rectangle's -- what a view
    // near a body diagonal needs, where a tile's footprint is a slanted band and its bounding rectangle half empty.  The band
    // is one strip  c1 <= a + strip_k * b <= c2  per layer (c1, c2: support of the footprint's eight corner points) intersected
    // with the rectangle; strip_k is the tile's (direction of the footprint's longest side, chosen on its farthest layer).
    bool use_strip = false, try_rows = false;
    float strip_k = 0.0f;
    int rowtab_off = 0;
    // SKIP: one bit per planned layer in the plan entry's spare bits (29 of its first word: a brick index is below 2^13,
    // vr_kernels.hip: tri_slab_candidate): the layer holds something visible (below)
    const bool skip_on = SKIP && P.skip_empty != 0 && skip_grid != nullptr;
    constexpr uint32_t LOB_MASK = SKIP ? 0x1fffu : 0xffffu;              // the entry's second 16-bit field
    // brick range [alo, alo + w) of brick row r of a layer whose band is c1 <= a + strip_k * b <= c2 inside bricks lo_a .. hi_a:
    // the taps of row r come from positions with b in [4 r - mrg, 4 r + 4 + mrg]
    auto strip_row = [&](float c1, float c2, int lo_a, int hi_a, int r, int &alo, int &w) {
        const float mrg = 0.5f + delta;
        const float y0 = (float)(4 * r) - mrg, y1 = (float)(4 * r + 4) + mrg;
        const float k0 = strip_k * y0, k1 = strip_k * y1;
        const float a_lo = c1 - fmaxf(k0, k1), a_hi = c2 - fminf(k0, k1);
        const float big = 1.0e9f;
        alo = clampi((int)floorf(fmaxf(fminf(a_lo - mrg, big), -big)) >> 2, lo_a, hi_a);
        const int ahi = clampi((int)floorf(fmaxf(fminf(a_hi + mrg, big), -big)) >> 2, lo_a, hi_a);
        w = max(ahi - alo + 1, 1);
    };
    int RA = 1, RB = 1, ia_lo = 0, ib_lo = 0, im_lo = 0, na_e = 0, nb_e = 0, nm_e = 0, taba_bytes = 0, tabb_bytes = 0, head_bytes = 0, slots_avail = 0;
    bool fits = false;
    VR_TSLAB_STAT(unsigned st_reason = 0;)
    for (;;) {
        const int T = 1 << LSH;
        const int nlay_m = nbr_m << (2 - LSH);                           // layers along the major axis
        if (threadIdx.x == 0) { red[0] = 0x7fffffff; red[1] = -0x7fffffff; red[2] = 0; red[3] = 0; red[4] = 0; red[5] = 0; red[6] = 0x7fffffff; red[7] = -1; red[8] = 0x7fffffff; red[9] = -1; red[10] = 0; }
        __syncthreads();
        // layer (along m) of the cell a position's taps start in: floor(max(f_m - 0.5, 0)) >> LSH
        auto layer_of = [&](float fm) -> int { return min((int)fmaxf(fm - 0.5f, 0.0f), ndim_m - 1) >> LSH; };
        {
            // first / last layer of this ray's prefix in progress coordinates (sgn * layer), prefix length: workgroup extremes
            // (wavefront reductions first: one lane per wavefront touches the LDS words)
            int r_first = 0x7fffffff, r_last = -0x7fffffff, r_len = 0;
            if (rem > 0) {
                float fx, fy, fz;
                scaled_here(fx, fy, fz);
                const int l_first = layer_of(sel3(ax_m, fx, fy, fz));
                float lx2, ly2, lz2;
                const float kk = (float)(rem - 1);
                voxel_float((POW2 ? Qx / Sx : qx) + kk * dsx, (POW2 ? Qy / Sy : qy) + kk * dsy, (POW2 ? Qz / Sz : qz) + kk * dsz, lx2, ly2, lz2);
                const int l_last = layer_of(sel3(ax_m, lx2, ly2, lz2));
                r_first = sgn * l_first;
                r_last = sgn * l_last + 1;                               // + 1: the closed form may sit one layer short
                r_len = rem;
            }
            r_first = wave_min_i(r_first); r_last = wave_max_i(r_last); r_len = wave_max_i(r_len);
            if (lane == 0) { atomicMin(&red[0], r_first); atomicMax(&red[1], r_last); atomicMax(&red[2], r_len); }
        }
        __syncthreads();
        c_first = uniform_i(red[0]); kmax = uniform_i(red[2]);
        c_last = uniform_i(red[1]);
        any_prefix = kmax > 0;
        // progress coordinates stay inside the volume
        c_last = sgn > 0 ? min(c_last, nlay_m - 1) : min(c_last, 0);
        n_phases = any_prefix ? c_last - c_first + 1 : 0;
        L0 = sgn * c_first;                                              // layer of phase 0
        delta = TS_MARGIN + (float)kmax * 1.2e-7f * nmax + emax * 2.4e-7f;
        // LDS carve-up: [plan | torus tables | ring].  Plan and tables cover only what this tile can touch -- the layers
        // between one before its first and three past its last (the prefetch distance), and the voxel indices of those
        // layers' rectangles -- so a 2048^3 volume costs the ring ~1 KiB instead of 20 and oblique footprints get the rest.
        // The tables hold BYTE offsets, 16-bit for the two minor axes (a slot row / a layer is < 64 KiB), 32-bit for the
        // major axis (ring base + layer slot + plane); they are addressed through virtual bases (real base - first index).
        Llo = max(sgn > 0 ? c_first - 1 : -(c_last + 3), 0); Lhi = min(sgn > 0 ? c_last + 3 : -(c_first - 1), nlay_m - 1);
        const int n_plan = max(Lhi - Llo + 1, 0);
        const float mrg = 0.5f + delta;                                  // taps: voxels floor(f - 0.5) and + 1 per axis
        // the corner rays' crossings of the two planes bounding layer L's samples: (a, b) of corner c at plane 0 / 1
        auto layer_points = [&](int L, float (&pa)[8], float (&pb)[8]) {
            const float c_lo = (float)(T * L) - 0.5f - delta, c_hi = (float)(T * L + T) + 0.5f + delta;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float gm = sel3(ax_m, G[c][0], G[c][1], G[c][2]), ga = sel3(ax_a, G[c][0], G[c][1], G[c][2]), gb = sel3(ax_b, G[c][0], G[c][1], G[c][2]);
                const float em = sel3(ax_m, E[0], E[1], E[2]), ea = sel3(ax_a, E[0], E[1], E[2]), eb = sel3(ax_b, E[0], E[1], E[2]);
                const float t1 = (c_lo - em) / gm, t2 = (c_hi - em) / gm;
                pa[c] = ea + t1 * ga; pa[c + 4] = ea + t2 * ga; pb[c] = eb + t1 * gb; pb[c + 4] = eb + t2 * gb;
            }
        };
        // ---- rows or rectangles?  On the tile's farthest layer (footprints grow with the distance from the eye): the widest row
        // of the band along each of the footprint's three directions (tile edge x, tile edge y, the rays' sweep across the layer)
        // against the rectangle's width; rows when they save 15 % of the slots.  Only tried when rectangles have failed (try_rows):
        // a tile that fits as rectangles keeps the cheaper loader
        use_strip = false; strip_k = 0.0f;
        if (ROWS && try_rows && stage && any_prefix && n_plan > 0) {
            const float em = sel3(ax_m, E[0], E[1], E[2]);
            const int Lf = fabsf((float)(T * Lhi) - em) >= fabsf((float)(T * Llo) - em) ? Lhi : Llo;
            float pa[8], pb[8];
            layer_points(Lf, pa, pb);
            float amin = pa[0], amax = pa[0], bmin = pb[0], bmax = pb[0];
#pragma unroll
            for (int q = 1; q < 8; q++) { amin = fminf(amin, pa[q]); amax = fmaxf(amax, pa[q]); bmin = fminf(bmin, pb[q]); bmax = fmaxf(bmax, pb[q]); }
            const float w_rect = (amax - amin + 2.0f * mrg) * 0.25f + 1.25f, n_rows = (bmax - bmin + 2.0f * mrg) * 0.25f + 1.25f;
            const float va[3] = {pa[1] - pa[0], pa[2] - pa[0], pa[4] - pa[0]}, vb[3] = {pb[1] - pb[0], pb[2] - pb[0], pb[4] - pb[0]};
            float w_best = w_rect;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (!(fabsf(vb[d]) * 8.0f > fabsf(va[d])) || !(fabsf(vb[d]) > 1.0e-3f)) continue;       // band nearly parallel to the rows: no use
                const float k = -va[d] / vb[d];
                float c1 = pa[0] + k * pb[0], c2 = c1;
#pragma unroll
                for (int q = 1; q < 8; q++) { const float cc = pa[q] + k * pb[q]; c1 = fminf(c1, cc); c2 = fmaxf(c2, cc); }
                const float w = ((c2 - c1) + fabsf(k) * (4.0f + 2.0f * mrg) + 2.0f * mrg) * 0.25f + 1.25f;
                if (w < w_best) { w_best = w; strip_k = k; }
            }
            use_strip = w_best <= 0.85f * w_rect && n_rows <= 30.0f;
            if (!use_strip) strip_k = 0.0f;
        }
        plan_stride = (ROWS && use_strip) ? 16 : 8;
        plan_bytes = (n_plan * plan_stride + 15) & ~15;
        plan_b = ring - (ptrdiff_t)Llo * plan_stride;                    // plan entry of layer L (Llo <= L <= Lhi) at plan_b + L * plan_stride
        if (stage && any_prefix && plan_bytes <= C::REGION / 2) {
            int m_bb = 0;                                                // widest bounding rectangle (its extent must fit the entry's 8 bits)
            int m_dda = 0, m_ddb = 0, m_loa = 0x7fffffff, m_hia = -1, m_lob = 0x7fffffff, m_hib = -1;
            for (int L = Llo + (int)threadIdx.x; L <= Lhi; L += TS_THREADS) {
                float pa[8], pb[8];
                layer_points(L, pa, pb);
                float amin = pa[0], amax = pa[0], bmin = pb[0], bmax = pb[0];
#pragma unroll
                for (int q = 1; q < 8; q++) { amin = fminf(amin, pa[q]); amax = fmaxf(amax, pa[q]); bmin = fminf(bmin, pb[q]); bmax = fmaxf(bmax, pb[q]); }
                // taps: voxels floor(f - 0.5) and + 1 per axis
                const float big = 1.0e9f;
                int lo_a = clampi((int)floorf(fmaxf(fminf(amin - 0.5f - delta, big), -big)) >> 2, 0, nbr_a - 1);
                int hi_a = clampi((int)floorf(fmaxf(fminf(amax + 0.5f + delta, big), -big)) >> 2, 0, nbr_a - 1);
                int lo_b = clampi((int)floorf(fmaxf(fminf(bmin - 0.5f - delta, big), -big)) >> 2, 0, nbr_b - 1);
                int hi_b = clampi((int)floorf(fmaxf(fminf(bmax + 0.5f + delta, big), -big)) >> 2, 0, nbr_b - 1);
                // (the guard's negative control, checked build only: every rectangle one brick short on each side, still well-formed)
                VR_TSLAB_CHK(if (no_stage == 3) { lo_a = min(lo_a + 1, hi_a); hi_a = max(hi_a - 1, lo_a); lo_b = min(lo_b + 1, hi_b); hi_b = max(hi_b - 1, lo_b); })
                const int dda = hi_a - lo_a, ddb = hi_b - lo_b;
                const uint2 e01 = make_uint2((uint32_t)lo_a | ((uint32_t)lo_b << 16), (uint32_t)min(dda, 255) | ((uint32_t)min(ddb, 255) << 8));
                int width = dda;                                         // widest row of the layer - 1
                if (ROWS && use_strip) {
                    float c1 = pa[0] + strip_k * pb[0], c2 = c1;
#pragma unroll
                    for (int q = 1; q < 8; q++) { const float cc = pa[q] + strip_k * pb[q]; c1 = fminf(c1, cc); c2 = fmaxf(c2, cc); }
                    c1 -= 0.03125f; c2 += 0.03125f;                      // (rounding of the products: far below 1/32 voxel)
                    width = 0;
                    for (int r = lo_b; r <= hi_b; r++) { int alo, w; strip_row(c1, c2, lo_a, hi_a, r, alo, w); width = max(width, w - 1); }
                    *reinterpret_cast<uint4 *>(plan_b + (ptrdiff_t)L * 16) = make_uint4(e01.x, e01.y, __float_as_uint(c1), __float_as_uint(c2));
                } else {
                    *reinterpret_cast<uint2 *>(plan_b + (ptrdiff_t)L * 8) = e01;
                }
                m_bb = max(m_bb, dda);
                m_dda = max(m_dda, width); m_ddb = max(m_ddb, ddb);      // every planned layer is one some phase reads or prefetches
                m_loa = min(m_loa, lo_a); m_hia = max(m_hia, hi_a); m_lob = min(m_lob, lo_b); m_hib = max(m_hib, hi_b);
            }
            m_dda = wave_max_i(m_dda); m_ddb = wave_max_i(m_ddb); m_bb = wave_max_i(m_bb);
            m_loa = wave_min_i(m_loa); m_hia = wave_max_i(m_hia); m_lob = wave_min_i(m_lob); m_hib = wave_max_i(m_hib);
            if (lane == 0) {
                atomicMax(&red[3], m_dda); atomicMax(&red[4], m_ddb); atomicMax(&red[10], m_bb);
                atomicMin(&red[6], m_loa); atomicMax(&red[7], m_hia); atomicMin(&red[8], m_lob); atomicMax(&red[9], m_hib);
            }
        }
        __syncthreads();
        RA = uniform_i(red[3]) + 1; RB = uniform_i(red[4]) + 1;
        // voxel index ranges of the tables: [first, last] per role, + 1 entry for the "index + 1" look-ups (at the high face
        // it repeats the last voxel: the + 1 tap of the last cell is the clamped one)
        ia_lo = 4 * clampi(uniform_i(red[6]), 0, nbr_a);
        const int ia_hi = min(4 * clampi(uniform_i(red[7]), -1, nbr_a) + 4, ndim_a);   // (nothing planned: empty)
        ib_lo = 4 * clampi(uniform_i(red[8]), 0, nbr_b);
        const int ib_hi = min(4 * clampi(uniform_i(red[9]), -1, nbr_b) + 4, ndim_b);
        im_lo = T * Llo;
        const int im_hi = min(T * Lhi + T, ndim_m);
        na_e = max(ia_hi - ia_lo + 1, 0); nb_e = max(ib_hi - ib_lo + 1, 0); nm_e = max(im_hi - im_lo + 1, 0);
        taba_bytes = (na_e * 2 + 3) & ~3; tabb_bytes = (nb_e * 2 + 3) & ~3;
        const int tabm_bytes = nm_e * 4;
        // (the ring starts on a 256-byte boundary: the 1-KiB DMA pieces then land on whole LDS rows, 1 % on the 1024^3 workload)
        // (rows: four row tables of 32 entries of 8 bytes behind the torus tables, one per layer in flight)
        rowtab_off = (plan_bytes + taba_bytes + tabb_bytes + tabm_bytes + 7) & ~7;
        head_bytes = (rowtab_off + ((ROWS && use_strip) ? 4 * 32 * 8 : 0) + 255) & ~255;
        slot_b = C::BRICK_BYTES >> (2 - LSH);
        slots_avail = (C::REGION - head_bytes) / slot_b;
        const int layer_slots_max = C::MAX_PIECES * TS_THREADS / (slot_b / 16);
        fits = head_bytes <= C::REGION / 2 && !(RA * RB > layer_slots_max || RA * RB * 3 > slots_avail || RA > 255 || RB > 255 || RA * RB * slot_b > 65535) &&
               uniform_i(red[10]) <= 255 && (!use_strip || RB <= 32);
        VR_TSLAB_STAT(st_reason = !stage ? 1u : (RA * RB > layer_slots_max ? 2u : (RA * RB * 3 > slots_avail ? 3u : (RA * RB * slot_b > 65535 ? 4u : 0u)));)
        // the attempts: whole layers as rectangles; half layers as rectangles (instances with per-axis copies); the thinnest as rows
        if (fits || !stage || !any_prefix || try_rows) break;
        if (C::HALF_OK && LSH == 2) LSH = 1;                             // once more with half layers
        else if (ROWS) try_rows = true;                                  // once more with rows
        else break;
        __syncthreads();                                                 // (every thread has read red[] before it is reset)
    }
    if (!fits) stage = false;
    const int T = 1 << LSH;
    uint16_t *tab_a = reinterpret_cast<uint16_t *>(ring + plan_bytes), *tab_b = reinterpret_cast<uint16_t *>(ring + plan_bytes + taba_bytes);
    uint32_t *tab_m = reinterpret_cast<uint32_t *>(ring + plan_bytes + taba_bytes + tabb_bytes);
    uint8_t *const slots = ring + (stage ? head_bytes : 0);             // the torus of brick slots
    const int CHR = slot_b / 16;                                         // 16-byte chunks per slot
    const int RZ = stage ? min(slots_avail / (RA * RB), 4) : 1;
    const int LA = RZ - 2;                                               // phases of prefetch distance: 1 or 2
    const uint32_t layer_bytes = (uint32_t)(RA * RB * slot_b);
    const int pieces = (RA * RB * CHR + TS_THREADS - 1) / TS_THREADS;
    const uint32_t str0 = 1u, str1 = (uint32_t)P.bnx, str2 = (uint32_t)P.bnx * (uint32_t)P.bny;   // brick index strides
    const uint32_t sA = sel3(ax_a, str0, str1, str2), sB = sel3(ax_b, str0, str1, str2), sM = sel3(ax_m, str0, str1, str2);
    // per-lane loader constants of piece q: torus coordinates (ta, tb) of the slot this lane's 16-byte chunk belongs to,
    // the chunk's index inside the slot, and whether the slot exists
    // ... and ld_rel: the byte offset of that chunk's source relative to the layer's torus origin, (ta * sA + tb * sB) * brick bytes
    // + 16 * chunk: per layer only the wrap-around terms are added (32-bit per lane; the 64-bit part of the address is scalar)
    int ld_ta[C::MAX_PIECES], ld_tb[C::MAX_PIECES];
    uint32_t ld_rel[C::MAX_PIECES];
#pragma unroll
    for (int q = 0; q < C::MAX_PIECES; q++) { ld_ta[q] = ld_tb[q] = 0x3fffffff; ld_rel[q] = 0; }      // (a slot that does not exist: beyond every rectangle)
    // the wrap-around terms: a torus column / row below the rectangle's origin holds the brick one ring further
    const uint32_t wrap_a = (uint32_t)RA * sA * (uint32_t)C::BRICK_BYTES, wrap_b = (uint32_t)RB * sB * (uint32_t)C::BRICK_BYTES;
    // (relative offsets are 32-bit: (RA + 1) * sA + (RB + 1) * sB bricks must stay below 2^31 bytes -- any volume a GPU holds)
    if ((uint64_t)(RA + 257) * (uint64_t)sA * (uint64_t)C::BRICK_BYTES + (uint64_t)(RB + 1) * (uint64_t)sB * (uint64_t)C::BRICK_BYTES >= (1ull << 31)) stage = false;
    if (stage && any_prefix) {
        // torus position of every layer's rectangle origin (second word of the plan entries)
        for (int L = Llo + (int)threadIdx.x; L <= Lhi; L += TS_THREADS) {
            uint2 *pe = reinterpret_cast<uint2 *>(plan_b + (ptrdiff_t)L * plan_stride);
            uint2 e = *pe;
            e.y |= ((e.x & 0xffffu) % (uint32_t)RA) << 16 | ((e.x >> 16) % (uint32_t)RB) << 24;
            *pe = e;
        }
        // torus tables: N + 1 entries per axis (the last one repeats voxel N - 1: the + 1 tap of the last cell is the
        // clamped one): byte offset of the slot of brick (i >> 2) mod R + of the position inside the 5x4x4 apron brick
        // (strides 1 / 5 / 20 voxels: by voxel axis x / y / z in the order-0 copy, by role a / b / m in the per-axis copies);
        // the major axis' entries include the ring's LDS base, so a tap address is the plain sum of three entries
        const int na = na_e + nb_e + nm_e;
        for (int e = (int)threadIdx.x; e < na; e += TS_THREADS) {
            int role, ii;                                                // role 0 / 1 / 2 = axis a / b / m
            if (e < na_e) { role = 0; ii = min(ia_lo + e, ndim_a - 1); }
            else if (e < na_e + nb_e) { role = 1; ii = min(ib_lo + e - na_e, ndim_b - 1); }
            else { role = 2; ii = min(im_lo + e - na_e - nb_e, ndim_m - 1); }
            const int axis = role == 0 ? ax_a : (role == 1 ? ax_b : ax_m);
            const int R = role == 0 ? RA : (role == 1 ? RB : RZ);
            const uint32_t stride = role == 0 ? (uint32_t)slot_b : (role == 1 ? (uint32_t)(RA * slot_b) : layer_bytes);
            const int order = PERM ? role : axis;
            const uint32_t in = (uint32_t)(ii & (role == 2 ? T - 1 : 3)) * (order == 0 ? 1u : (order == 1 ? 5u : 20u)) * (uint32_t)sizeof(VoxelT);
            const uint32_t off = (uint32_t)((ii >> (role == 2 ? LSH : 2)) % R) * stride + in;
            if (role == 0) tab_a[e] = (uint16_t)off;
            else if (role == 1) tab_b[e - na_e] = (uint16_t)off;
            else tab_m[e - na_e - nb_e] = off + lds_offset_of(slots);
        }
#pragma unroll
        for (int q = 0; q < C::MAX_PIECES; q++) {
            const int c = (q * TS_NW + (int)wave) * 64 + (int)lane;
            const int slot = CHR == 5 ? c / 5 : (CHR == 10 ? c / 10 : c / CHR);
            const int part = c - slot * CHR, tb = slot / RA, ta = slot - tb * RA;
            if (slot < RA * RB && q < pieces) {
                ld_ta[q] = ta; ld_tb[q] = tb;
                ld_rel[q] = ((uint32_t)ta * sA + (uint32_t)tb * sB) * (uint32_t)C::BRICK_BYTES + (uint32_t)part * 16u;
            }
        }
        if (skip_on) {
            // ---- SKIP: which planned layers hold anything visible.  A sample of phase L has its base voxel (floor(f - 0.5) per
            // axis) inside layer L's rectangle and inside the layer, i.e. in one of the 8^3-voxel cells (brick >> 1 along a and b,
            // (T L) >> 3 along m) the rectangle touches; the grid holds every cell's maximum over the cell AND its 26 neighbours, so
            // it bounds all eight taps (the + 1 neighbours may lie in the next cell).  An interpolated value never exceeds its largest
            // tap (each lerp is one fma of values between its end points, rounding is monotone), so where every touched cell is
            // <= skip_thresh (host: the largest voxel value that classifies to exactly zero, with everything below it) every sample
            // of the phase adds exactly nothing.  A lane per cell, eight cell layers per wavefront with their loads in flight together.
            __syncthreads();                                             // (the torus positions above were derived from the entries' first words)
            const uint32_t cst1 = (uint32_t)P.cnx, cst2 = (uint32_t)P.cnx * (uint32_t)P.cny;
            const uint32_t csA = sel3(ax_a, 1u, cst1, cst2), csB = sel3(ax_b, 1u, cst1, cst2), csM = sel3(ax_m, 1u, cst1, cst2);
            // (layers of one cell layer -- two whole, four half layers -- are tested together, on the union of their rectangles)
            const int GL = 3 - LSH, G = 1 << GL;                         // layers per cell layer
            const int g_lo = Llo >> GL, g_hi = Lhi >> GL;
            constexpr int SU = 8;                                        // cell layers per wavefront and round: eight loads in flight per lane
            for (int gb = g_lo + SU * (int)wave; gb <= g_hi; gb += SU * TS_NW) {
                uint32_t vmax[SU];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int g = gb + u;
                    int lo_a = 0x7fffffff, hi_a = -1, lo_b = 0x7fffffff, hi_b = -1;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int L = (g << GL) + k;
                        if (k < G && g <= g_hi && L >= Llo && L <= Lhi) {
                            const uint2 e = *reinterpret_cast<const uint2 *>(plan_b + (ptrdiff_t)L * plan_stride);
                            const uint32_t ex = __builtin_amdgcn_readfirstlane(e.x), ey = __builtin_amdgcn_readfirstlane(e.y);
                            const int la_ = (int)(ex & 0xffffu), lb_ = (int)((ex >> 16) & LOB_MASK);
                            lo_a = min(lo_a, la_); hi_a = max(hi_a, la_ + (int)(ey & 255u));
                            lo_b = min(lo_b, lb_); hi_b = max(hi_b, lb_ + (int)((ey >> 8) & 255u));
                        }
                    }
                    const int ca0 = lo_a >> 1, cb0 = lo_b >> 1, nca = (hi_a >> 1) - ca0 + 1;
                    const int ncell = hi_a < 0 ? 0 : nca * ((hi_b >> 1) - cb0 + 1);
                    const uint32_t cbase = (uint32_t)g * csM + (uint32_t)ca0 * csA + (uint32_t)cb0 * csB;
                    const float rn = 1.0f / (float)nca;
                    auto cell_of = [&](int c) -> uint32_t {              // c-th cell of the rectangle, row-major (c < 2^16: the quotient is exact)
                        const int cb = (int)(((float)c + 0.5f) * rn), ca = c - cb * nca;
                        return cbase + (uint32_t)ca * csA + (uint32_t)cb * csB;
                    };
                    vmax[u] = (int)lane < ncell ? (uint32_t)skip_grid[cell_of((int)lane)] : 0u;
                    if (ncell > 64)                                      // (wide rectangles: the rows of the tall tiles)
                        for (int c = (int)lane + 64; c < ncell; c += 64) vmax[u] = max(vmax[u], (uint32_t)skip_grid[cell_of(c)]);
                }
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const bool visible = __any((int)vmax[u] > P.skip_thresh ? 1 : 0) != 0;
                    const int L = ((gb + u) << GL) + (int)lane;          // a lane per layer of the cell layer
                    if (visible && (int)lane < G && L >= Llo && L <= Lhi) *reinterpret_cast<uint32_t *>(plan_b + (ptrdiff_t)L * plan_stride) |= 1u << 29;
                }
            }
        }
    }
    VR_TSLAB_CHK(if (threadIdx.x < (unsigned)CHK_REQ_WORDS) chk_req[threadIdx.x] = 0u;)
    __syncthreads();
    const uint32_t ring_base = lds_offset_of(slots);
    // SKIP: does layer X hold anything visible (per lane; everything does without skipping) / is it read by a phase that samples --
    // as its own layer, or as the layer of the + 1 taps of the one below
    auto layer_visible = [&](int X) -> bool {
        if (!skip_on) return true;
        if (X < Llo || X > Lhi) return false;
        return ((*reinterpret_cast<const uint32_t *>(plan_b + (ptrdiff_t)X * plan_stride) >> 29) & 1u) != 0u;
    };
    auto layer_needed_lane = [&](int X) -> bool { return layer_visible(X) || layer_visible(X - 1); };
    auto layer_needed = [&](int X) -> bool { return !skip_on || __any(layer_needed_lane(X) ? 1 : 0) != 0; };   // (uniform X: the prologue's requests)

    // request the bricks of layer L (its rectangle of the plan) into slot L mod RZ.  Per lane and piece: two wrap-around tests
    // (is this lane's torus column / row below the rectangle's torus origin?), two extent tests, two masked adds -- the layer's
    // base address (source copy + the brick of torus origin (0, 0) + the half of the brick) is scalar arithmetic
    uint2 *const rowtab = reinterpret_cast<uint2 *>(ring + rowtab_off);
    // rows: the table of layer L -- per brick row r of its rectangle the row's own brick range, as (byte offset of the row's
    // torus origin relative to the layer's, torus column of its first brick | number of bricks << 8).  Written by the lanes of
    // ONE wavefront a phase before the layer is requested (the phase's barrier publishes it)
    auto make_rows = [&](int L) {
        if (L < Llo || L > Lhi) return;
        const uint4 e = *reinterpret_cast<const uint4 *>(plan_b + (ptrdiff_t)L * 16);
        const int lo_a = (int)(e.x & 0xffffu), lo_b = (int)((e.x >> 16) & LOB_MASK), dda = (int)(e.y & 255u), ddb = (int)((e.y >> 8) & 255u), la = (int)((e.y >> 16) & 255u);
        if ((int)lane <= ddb && lane < 32u) {
            int alo, w;
            strip_row(__uint_as_float(e.z), __uint_as_float(e.w), lo_a, lo_a + dda, lo_b + (int)lane, alo, w);
            const int la_row = alo % RA;
            const uint32_t rowbase = (uint32_t)((alo - la_row) - (lo_a - la)) * sA * (uint32_t)C::BRICK_BYTES;     // a multiple of RA bricks, >= 0
            rowtab[(L & 3) * 32 + (int)lane] = make_uint2(rowbase, (uint32_t)la_row | ((uint32_t)w << 8));
        }
    };
    // (round 4, measured and not kept: the lanes' torus-relative positions and wrapped source offsets cached in registers and
    // recomputed under a wave-uniform branch only when the rectangle's torus origin moves -- two compares per piece and layer
    // instead of 13 VALU: SLOWER, 1.142 -> 1.153 ms at the default pose, 0.448 -> 0.465 on the cfg2 shape, orbit poses + 1-3 %.
    // A phase's instructions outside the sample loop are not what its time follows)
    auto issue_layer = [&](int L) {
        if (L < Llo || L > Lhi) return;
        VR_TSLAB_CHK(if (threadIdx.x == 0 && L - Llo < 32 * CHK_REQ_WORDS) atomicOr(&chk_req[(L - Llo) >> 5], 1u << ((L - Llo) & 31));)
        const uint2 e = *reinterpret_cast<const uint2 *>(plan_b + (ptrdiff_t)L * plan_stride);
        const uint32_t ex = __builtin_amdgcn_readfirstlane(e.x), ey = __builtin_amdgcn_readfirstlane(e.y);
        const int lo_a = (int)(ex & 0xffffu), lo_b = (int)((ex >> 16) & LOB_MASK);
        const int dda = (int)(ey & 255u), ddb = (int)((ey >> 8) & 255u), la = (int)((ey >> 16) & 255u), lb = (int)(ey >> 24);
        const int lz = L % RZ;
        // brick layer and, for half layers, which half of the brick (its first or last two planes along m: 40 contiguous voxels)
        const uint64_t origin = (uint64_t)(uint32_t)(lo_a - la) * sA + (uint64_t)(uint32_t)(lo_b - lb) * sB + (uint64_t)(uint32_t)(L >> (2 - LSH)) * sM;
        const uint8_t *const base = src_m + origin * (uint64_t)C::BRICK_BYTES + (LSH == 1 ? (uint64_t)(L & 1) * (uint64_t)(C::BRICK_BYTES / 2) : 0ull);
        const uint32_t dst = ring_base + (uint32_t)lz * layer_bytes + (uint32_t)wave * 1024u;
        if (ROWS && use_strip) {
            // rows: this lane's row tells where its bricks start on the torus and how many there are
            // (all row entries first, then the requests: the LDS round trips overlap)
            uint2 re[C::MAX_PIECES];
            int obs[C::MAX_PIECES], mbs[C::MAX_PIECES];
#pragma unroll
            for (int q = 0; q < C::MAX_PIECES; q++) {
                const int db_ = ld_tb[q] - lb;
                mbs[q] = db_ >> 31; obs[q] = db_ + (mbs[q] & RB);
                re[q] = rowtab[(L & 3) * 32 + (obs[q] & 31)];
            }
#pragma unroll
            for (int q = 0; q < C::MAX_PIECES; q++) {
                if (q >= pieces) break;
                const int da_ = ld_ta[q] - (int)(re[q].y & 255u), ma = da_ >> 31, oa = da_ + (ma & RA);
                if (obs[q] <= ddb && obs[q] >= 0 && oa < (int)(re[q].y >> 8))
                    glds16_rel(base, ld_rel[q] + ((uint32_t)ma & wrap_a) + ((uint32_t)mbs[q] & wrap_b) + re[q].x, dst + (uint32_t)(q * TS_NW) * 1024u);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < C::MAX_PIECES; q++) {
            if (q >= pieces) break;
            const int da_ = ld_ta[q] - la, db_ = ld_tb[q] - lb;
            const int ma = da_ >> 31, mb = db_ >> 31;                   // all ones: wraps around
            const int oa = da_ + (ma & RA), ob = db_ + (mb & RB);
            if (oa <= dda && ob <= ddb)
                glds16_rel(base, ld_rel[q] + ((uint32_t)ma & wrap_a) + ((uint32_t)mb & wrap_b), dst + (uint32_t)(q * TS_NW) * 1024u);
        }
    };

    VR_TSLAB_STAT(unsigned st_samples = 0, st_iters = 0, st_skipped = 0, st_runs = 0; const uint64_t st_clk0 = clock64(), st_wall0 = wall_clock64();)
    VR_TSLAB_CHK(unsigned chk_violations = 0;)
    // The staged march, compiled once per major axis M (the tables of the minor axes are 16-bit, M's 32-bit; the layer of
    // a sample is its M index >> LSH).  PA = the voxel axis along which one LDS address yields a pair of taps (the apron
    // axis of the copy the bricks came from): x, or the first minor axis in the per-axis copies (y when M is x).
    auto staged_march = [&](auto m_tag) {
        constexpr int M = decltype(m_tag)::value, A = M == 0 ? 1 : 0;          // minor axes: A and the other one
        constexpr int PA = PERM ? A : 0, O1 = PA == 0 ? 1 : 0, O2 = 2;         // the other two voxel axes, in x-y-z order
        // ---- prologue: the layers phases 0 .. LA-1 read
        if (ROWS && use_strip) {                                         // rows: the tables of the prologue's layers and of the one phase 0 requests, a wavefront each
            const int first = sgn > 0 ? L0 : L0 + 1 - LA - 1;                // (sgn > 0: L0 .. L0 + LA + 1; else L0 - LA .. L0 + 1)
            if ((int)wave < LA + 2) make_rows(first + (int)wave);
            __syncthreads();
        }
        if (sgn > 0) { for (int l = 0; l <= LA; l++) if (layer_needed(L0 + l)) issue_layer(L0 + l); }
        else { for (int l = 1; l >= 1 - LA; l--) if (layer_needed(L0 + l)) issue_layer(L0 + l); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // prepared sample (the one at the current position): tap BYTE addresses of the four corner pairs -- (O1, O2) = (0,0),
        // (1,0), (0,1), (1,1), i.e. (y0,z0), (y1,z0), (y0,z1), (y1,z1) when the pairs lie along x --, the three weights, its
        // layer along m.  prepare() runs once per sample -- right after the advance --
        // so a ray that waits for its layer's phase keeps the values across the barrier instead of recomputing them.
        uint32_t a00 = 0, a10 = 0, a01 = 0, a11 = 0;
        float wx = 0.0f, wy = 0.0f, wz = 0.0f;
        int lay = -0x7fffffff;                                            // of the prepared sample, + LAY_BIAS
        const int LAY_BIAS = (int)(0x4B000000u >> (2 + LSH));
        // virtual table bases: real base - first index (unsigned wrap-around is fine, the sum is what is used)
        const uint32_t tab_a_b = lds_offset_of(tab_a) - 2u * (uint32_t)ia_lo, tab_b_b = lds_offset_of(tab_b) - 2u * (uint32_t)ib_lo, tab_m_b = lds_offset_of(tab_m) - 4u * (uint32_t)im_lo;
        // byte offset of voxel index `t` (a non-negative whole number held as a float; and of t + 1) along voxel axis AX, from the
        // table of the role AX plays.  No float -> int conversion and no shift (v_cvt_i32_f32 and v_lshl_add_u32 issue at the slow
        // rate, 4.35 cycles each): 2^23 + 2 t (4 t for the 32-bit table) is exact, its BITS are 0x4B000000 + 2 t, and the table's
        // virtual base minus 0x4B000000 sits in a vector register -- one fma and one integer add at the fast rate
        const float k_magic = in_vgpr(8388608.0f);
        uint32_t mb_a = tab_a_b - 0x4B000000u, mb_b = tab_b_b - 0x4B000000u, mb_m = tab_m_b - 0x4B000000u;
        asm volatile("" : "+v"(mb_a), "+v"(mb_b), "+v"(mb_m));
        auto look = [&](auto ax_tag, float t, uint32_t &o0, uint32_t &o1, uint32_t &bits) {
            constexpr int AX = decltype(ax_tag)::value;
            if (AX == M) {
                bits = __float_as_uint(__builtin_fmaf(t, 4.0f, k_magic));
                VR_LDS_AS const uint32_t *e = reinterpret_cast<VR_LDS_AS const uint32_t *>((size_t)(bits + mb_m));
                o0 = e[0]; if (AX != PA) o1 = e[1];
            } else {
                bits = __float_as_uint(__builtin_fmaf(t, 2.0f, k_magic));
                VR_LDS_AS const uint16_t *e = reinterpret_cast<VR_LDS_AS const uint16_t *>((size_t)(bits + (AX == A ? mb_a : mb_b)));
                o0 = e[0]; if (AX != PA) o1 = e[1];
            }
        };
        // (the weights are derived from ux, uy, uz by the caller AFTER the previous sample has used its own: no copies)
        float ux = 0.0f, uy = 0.0f, uz = 0.0f;
        VR_TSLAB_CHK(int chk_idx[3] = {0, 0, 0};
        // violations of the prepared sample's taps in phase L: voxel indices (idx_a | idx_a + 1 inside one apron slot), (idx_b, idx_b + 1), (idx_m, idx_m + 1)
        auto check_taps = [&](int L) -> unsigned {
            unsigned bad = 0;
            const int ia = chk_idx[ax_a], ib = chk_idx[ax_b], im = chk_idx[ax_m];
            for (int db = 0; db < 2; db++)
                for (int dm = 0; dm < 2; dm++) {
                    const int jb = min(ib + db, ndim_b - 1), jm = min(im + dm, ndim_m - 1);
                    const int lyr = jm >> LSH;
                    // resident in phase L: L and the layer above it (the + 1 taps), one more in the marching direction when the ring is four deep
                    const int lo_res = sgn > 0 ? L : L - (LA >= 2 ? 1 : 0), hi_res = sgn > 0 ? L + 1 + (LA >= 2 ? 1 : 0) : L + 1;
                    if (lyr < lo_res || lyr > hi_res || lyr < Llo || lyr > Lhi) { bad++; continue; }
                    if (lyr - Llo < 32 * CHK_REQ_WORDS && ((chk_req[(lyr - Llo) >> 5] >> ((lyr - Llo) & 31)) & 1u) == 0u) { bad++; continue; }   // never requested
                    const uint2 e = *reinterpret_cast<const uint2 *>(plan_b + (ptrdiff_t)lyr * plan_stride);
                    const int lo_a = (int)(e.x & 0xffffu), lo_b = (int)((e.x >> 16) & LOB_MASK), dda = (int)(e.y & 255u), ddb = (int)((e.y >> 8) & 255u);
                    const int ba = ia >> 2, bb = jb >> 2;
                    if (ba < lo_a || ba > lo_a + dda || bb < lo_b || bb > lo_b + ddb) { bad++; continue; }
                    if (ROWS && use_strip) {                             // rows: inside the row's own range (what make_rows had loaded)
                        const uint4 e4 = *reinterpret_cast<const uint4 *>(plan_b + (ptrdiff_t)lyr * 16);
                        int alo, w;
                        strip_row(__uint_as_float(e4.z), __uint_as_float(e4.w), lo_a, lo_a + dda, bb, alo, w);
                        if (ba < alo || ba >= alo + w) bad++;
                    }
                    if (ia < 0 || ia > ndim_a - 1 || ib < 0 || im < 0) bad++;
                }
            return bad;
        };)
        // The shader's u = max(f - 0.5, 0), i = floor(u), w = u - i WITHOUT the max (v_max_f32 issues at the slow rate, 4.5 cycles;
        // three per sample): a sample the march takes lies in the box, f >= 0, so u' = f - 0.5 > -1 and the conversion's
        // truncation already gives i = 0 where the max would; the weight is (u' - trunc(u')) with the subtraction's CLAMP
        // modifier (output clamped to [0, 1], free: tools/ubench/valu_rates2.hip) -- u' itself below zero, i.e. 0, and
        // u - floor(u) exactly (both are exact subtractions) everywhere else.  (A lane's position one step beyond its last
        // sample may lie outside the box: its look-ups and taps read whatever LDS holds there -- out-of-range LDS reads
        // return zero -- and are never composited.)
        float tx = 0.0f, ty = 0.0f, tz = 0.0f;
        auto prepare = [&]() {
            float fx, fy, fz;
            scaled_here(fx, fy, fz);
            ux = fx - 0.5f; uy = fy - 0.5f; uz = fz - 0.5f;
            tx = __builtin_truncf(ux); ty = __builtin_truncf(uy); tz = __builtin_truncf(uz);   // == floor(max(u, 0)) as a float: u > -1 (-0.0 counts as 0 below)
            VR_TSLAB_CHK(chk_idx[0] = (int)tx; chk_idx[1] = (int)ty; chk_idx[2] = (int)tz;)
            uint32_t p0, punused = 0, q0, q1, z0, z1, bp, bq, bz;
            look(std::integral_constant<int, PA>{}, PA == 0 ? tx : ty, p0, punused, bp);   // the second taps of the pairs are the first ones' next elements (apron)
            look(std::integral_constant<int, O1>{}, O1 == 0 ? tx : ty, q0, q1, bq);
            look(std::integral_constant<int, O2>{}, tz, z0, z1, bz);
            // the layer, biased: (0x4B000000 + 4 i) >> (2 + LSH) == LAY_BIAS + (i >> LSH)
            lay = (int)((M == PA ? bp : (M == O1 ? bq : bz)) >> (2 + LSH));
            const uint32_t pz0 = p0 + z0, pz1 = p0 + z1;
            a00 = pz0 + q0; a10 = pz0 + q1; a01 = pz1 + q0; a11 = pz1 + q1;
        };
        auto sat_sub = [](float a, float b) { float r; asm("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; };
        auto weights = [&]() { wx = sat_sub(ux, tx); wy = sat_sub(uy, ty); wz = sat_sub(uz, tz); };
        prepare(); weights();
        // samples of the prefix taken so far / to take, as floats: the per-sample bookkeeping is then one fp32 add
        const bool ahead_ok = LA >= 2;
        float takenf = 0.0f;
        const float limitf = (float)rem;
        // SKIP: the phases come in RUNS -- stretches of layers with something visible, marched by the phase loop as it is without
        // skipping (requests included: at most LA + 1 layers per run are fetched for nothing), and stretches of empty layers, which
        // the rays cross in ONE loop of position additions behind which only the layers the next sampling phases read are requested.
        // (Two kinds of phase inside one loop cost every sampling phase 15 %: with a second sample loop next to it the register
        // allocator puts the loop-carried values of the first back into other registers between the phases -- measured, 1.09 ->
        // 1.26 ms with nothing to skip; as runs those copies happen once per run.)
        auto request_of = [&](int q) -> int { const int Lq = L0 + sgn * q; return sgn > 0 ? Lq + LA + 1 : Lq - LA; };   // the layer phase q requests
        bool tile_done = false;
        for (int p = 0; p < n_phases && !tile_done;) {
            int p_end = n_phases;                                        // end of the run that starts at phase p
            bool sampling = true;
            if (SKIP && skip_on) {
                // the next 64 phases' layers at once: a lane each
                const int q = p + (int)lane;
                const uint64_t vm = __ballot((q < n_phases && layer_visible(L0 + sgn * q)) ? 1 : 0);
                sampling = (vm & 1ull) != 0ull;
                const uint64_t change = sampling ? ~vm : vm;             // (phases beyond the last count as empty)
                p_end = min(p + (change != 0ull ? (int)__builtin_ctzll(change) : 64), n_phases);
            }
            if (SKIP && !sampling) {
                const int run = p_end - p;
                VR_TSLAB_STAT(st_skipped += (unsigned)run; st_runs++;)
                // the requests of the run's phases that a sampling phase behind the run reads (at most LA + 1 layers, consecutive)
                const int ql = p + (int)lane;
                uint64_t need = __ballot(((int)lane < run && layer_needed_lane(request_of(ql))) ? 1 : 0);
                if (ROWS && use_strip) {                                 // rows: their tables first, a wavefront each
                    uint64_t nm = need;
                    for (int w = 0; nm != 0ull; w++, nm &= nm - 1ull)
                        if ((int)wave == w) make_rows(request_of(p + (int)__builtin_ctzll(nm)));
                    __syncthreads();
                }
                for (; need != 0ull; need &= need - 1ull) issue_layer(request_of(p + (int)__builtin_ctzll(need)));
                if (ROWS && use_strip && (int)wave == TS_NW - 1) make_rows(request_of(p_end));   // ... and of the layer the phase behind the run requests
                // the rays: position additions until they stand in the first layer behind the run (or have no prefix sample left);
                // a position's layer needs its major coordinate alone, and the sample a ray stops at is prepared once, behind the loop
                const int lim = L0 + sgn * p_end + LAY_BIAS, sm = sgn >> 31;
                const bool marching = da < 0.95f;                        // (dest.a does not change in here)
                bool here = marching && takenf < limitf && (((lay - lim) ^ sm) - sm) < 0;
                if (__any(here ? 1 : 0)) {
                    do {
                        const float vf = here ? 1.0f : 0.0f;
                        if (POW2) { Qx = __builtin_fmaf(dSx, vf, Qx); Qy = __builtin_fmaf(dSy, vf, Qy); Qz = __builtin_fmaf(dSz, vf, Qz); }
                        else { qx = __builtin_fmaf(dsx, vf, qx); qy = __builtin_fmaf(dsy, vf, qy); qz = __builtin_fmaf(dsz, vf, qz); }
                        takenf += vf;
                        float fx, fy, fz;
                        scaled_here(fx, fy, fz);
                        const float tm = __builtin_truncf((M == 0 ? fx : (M == 1 ? fy : fz)) - 0.5f);
                        lay = (int)(__float_as_uint(__builtin_fmaf(tm, 4.0f, k_magic)) >> (2 + LSH));
                        VR_TSLAB_STAT(st_iters++;)
                        here = marching && takenf < limitf && (((lay - lim) ^ sm) - sm) < 0;
                    } while (__any(here ? 1 : 0));
                    prepare(); weights();
                }
                slab_wait_pieces(0);
                if (__syncthreads_and(!(takenf < limitf && da < 0.95f) ? 1 : 0)) tile_done = true;
                p = p_end;
                continue;
            }
            for (; p < p_end; p++) {
            const int L = L0 + sgn * p, Lnext = L + sgn;
            // the layer that phase p + LA reads first
            issue_layer(sgn > 0 ? L + LA + 1 : L - LA);
            // (SKIP: no running ahead out of a run -- the layer behind it is not sampled and may not have been requested)
            const bool ahead_p = SKIP ? (ahead_ok && p + 1 < p_end) : ahead_ok;
            if (ROWS && use_strip && (int)wave == (p & (TS_NW - 1))) make_rows(sgn > 0 ? L + LA + 2 : L - LA - 1);   // rows: the table of the layer the NEXT phase requests
            // ---- this phase's samples: the ones whose cell lies in layer L.  The body is straight-line code for the whole
            // wavefront: a lane without a sample in this layer reads taps at its (valid, unchanged) prepared addresses and
            // drops the result -- exec-mask branches around the body cost more scalar instructions than the arithmetic
            // they would save (39 SALU per iteration in the branchy build).  Predication is arithmetic where that is exact:
            // vf = 1.0 / 0.0, position += step * vf (one fma: == the shader's addition, or the position itself), the
            // compositing weight (1 - dest.a) * vf.  A ray that has reached dest.a >= 0.95 takes no more samples (the
            // shader's test before every sample, VolumeRenderer.cs:118; dest.a never decreases).
            // (a second, select-free copy of the body for the iterations in which every lane has a sample -- four out of five --
            // measured SLOWER: 1.65 vs 1.51 ms; the look-ups issued at the end of one copy are consumed by either)
            // (round 4, ISA inspection: with the sample loop written `for (;;) { if (!any(here)) break; ... }` and two instances of
            // it per phase (with / without the low-face clamp) the register allocator kept the 22 loop-carried values in other
            // registers outside the loops: three sets of 22 v_mov per phase, ~150 issue cycles against ~860 for the phase's 4.3
            // iterations.  One loop per clamped / unclamped stretch of phases (1.25 -> 1.30 ms) and phases + samples as one flat
            // loop (1.34 ms) moved the copies INTO the sample loop.  What removes them: ONE instance of the loop, rotated by hand
            // -- the continuation test at the bottom, `if (any) do { ... } while (any)` -- so that the values leave the loop in
            // the registers the body wrote them to: no copies at all, 306 -> 178 VALU in the phase loop's text, 1.26 -> 1.17 ms.
            // The single instance always clamps (three v_max, 13 cycles per iteration: less than the copies cost).
            // Not available on this chip: d16 LDS loads into registers whose high halves hold 0x4B00 (the register would BE
            // the float 2^23 + v, no conversion): with SRAM ECC the hardware zeroes the unused half (tools/ubench/d16_preserve.hip))
            auto phase_samples = [&]() {
                bool alive = takenf < limitf && da < 0.95f;
                bool here = alive && lay == L + LAY_BIAS;
                if (__any(here ? 1 : 0)) do {
                    // RZ = 4: the layer BEHIND the two this phase reads is resident as well (requested two phases ago, landed before
                    // the last barrier), so a ray may run one layer ahead of the phase: lanes whose own layer has 4 samples fill the
                    // idle slot of the iteration that the 5-sample lanes need, and own a sample less in the next phase -- the
                    // wavefront then takes ~4.1 iterations per layer instead of max(4, 5) = 5.  The phase still ends when no lane is
                    // left IN layer L.
                    const bool valid = here || (alive && lay == Lnext + LAY_BIAS && ahead_p);
                    VR_TSLAB_STAT(st_iters++; st_samples += valid ? 1 : 0;)
                    VR_TSLAB_CHK(if (valid) chk_violations += check_taps(L);)
                    const float vf = valid ? 1.0f : 0.0f;
                    VR_LDS_AS const VoxelT *p00 = reinterpret_cast<VR_LDS_AS const VoxelT *>((size_t)a00), *p10 = reinterpret_cast<VR_LDS_AS const VoxelT *>((size_t)a10);
                    VR_LDS_AS const VoxelT *p01 = reinterpret_cast<VR_LDS_AS const VoxelT *>((size_t)a01), *p11 = reinterpret_cast<VR_LDS_AS const VoxelT *>((size_t)a11);
                    // pairs along x: p00 = (y0, z0) ...; pairs along y (PA == 1): p00 = (x0, z0), p10 = (x1, z0), p01 = (x0, z1), p11 = (x1, z1)
                    const uint32_t v000 = p00[0], v100 = PA == 0 ? p00[1] : p10[0], v010 = PA == 0 ? p10[0] : p00[1], v110 = p10[1];
                    const uint32_t v001 = p01[0], v101 = PA == 0 ? p01[1] : p11[0], v011 = PA == 0 ? p11[0] : p01[1], v111 = p11[1];
                    // the next sample's position and table look-ups travel with the taps (a lane that did not advance prepares
                    // the same sample again: same values)
                    if (POW2) { Qx = __builtin_fmaf(dSx, vf, Qx); Qy = __builtin_fmaf(dSy, vf, Qy); Qz = __builtin_fmaf(dSz, vf, Qz); }
                    else { qx = __builtin_fmaf(dsx, vf, qx); qy = __builtin_fmaf(dsy, vf, qy); qz = __builtin_fmaf(dsz, vf, qz); }
                    takenf += vf;
                    prepare();
                    const float ax = wx, ay = wy, az = wz;
                    float c00, c10, c01, c11;
                    // the seven lerps are one fma each (vr_device.h: tri_lerp): the arithmetic of a real GL's linear filter, and
                    // 2.4 % (16-bit) to 5.5 % (8-bit, small volumes) fewer issue cycles than three operations each
                    // (profiles/r05_fma_lerp_experiment.txt)
                    auto lerp = [](float a, float b, float t) { return tri_lerp(a, b, t); };
                    if (sizeof(VoxelT) == 1) {
                        // integer -> float without v_cvt (4.4 cycles each, eight per sample): 2^23 | v IS the float 2^23 + v for
                        // v < 2^23 (v_or_b32: 2.3 cycles); the x differences need no un-biasing -- (2^23 + a) - (2^23 + b) == a - b
                        // exactly -- so only the four x0 taps pay a subtraction.  8-bit taps only (1.154 -> 1.131 ms on a 1024^3
                        // volume): 16-bit taps are converted straight out of the register halves the loads packed them into, and
                        // lose (1.243 -> 1.269 ms)
                        const float b000 = __uint_as_float(v000 | k_magic_u), b100 = __uint_as_float(v100 | k_magic_u), b010 = __uint_as_float(v010 | k_magic_u), b110 = __uint_as_float(v110 | k_magic_u);
                        const float b001 = __uint_as_float(v001 | k_magic_u), b101 = __uint_as_float(v101 | k_magic_u), b011 = __uint_as_float(v011 | k_magic_u), b111 = __uint_as_float(v111 | k_magic_u);
                        const float c000 = b000 - k_magic_f, c010 = b010 - k_magic_f, c001 = b001 - k_magic_f, c011 = b011 - k_magic_f;
                        // (b - a of two biased taps is the exact difference of the voxels, as before)
                        c00 = __builtin_fmaf(ax, b100 - b000, c000); c10 = __builtin_fmaf(ax, b110 - b010, c010);
                        c01 = __builtin_fmaf(ax, b101 - b001, c001); c11 = __builtin_fmaf(ax, b111 - b011, c011);
                    } else {
                        const float c000 = (float)v000, c100 = (float)v100, c010 = (float)v010, c110 = (float)v110;
                        const float c001 = (float)v001, c101 = (float)v101, c011 = (float)v011, c111 = (float)v111;
                        c00 = lerp(c000, c100, ax); c10 = lerp(c010, c110, ax);
                        c01 = lerp(c001, c101, ax); c11 = lerp(c011, c111, ax);
                    }
                    const float c0 = lerp(c00, c10, ay), c1 = lerp(c01, c11, ay);
                    float c, cg = 0.0f, cb = 0.0f, a;
                    classify(lerp(c0, c1, az), c, cg, cb, a);
                    if (MODE == 1) {
                        da = (valid && da < a) ? a : da;
                    } else if (MODE == 3) {
                        const bool take = valid && da < a;
                        drgb = take ? c : drgb; dg = take ? cg : dg; db = take ? cb : db; da = take ? a : da;
                    } else {
                        const float om = __builtin_fmaf(-da, vf, vf);       // (1 - dest.a) or 0: x + y * 0 == x, the sample of a lane that has none adds nothing
                        drgb += c * om;
                        if (MODE == 2) { dg += cg * om; db += cb * om; }
                        da += a * om;
                    }
                    weights();                                               // of the sample just prepared
                    alive = takenf < limitf && da < 0.95f;
                    here = alive && lay == L + LAY_BIAS;
                } while (__any(here ? 1 : 0));
            };
            phase_samples();
            // ---- what the next phase reads must have landed before its barrier: with one phase of prefetch distance
            // that is the layer just requested, with two it was requested a phase ago
            slab_wait_pieces(0);                                         // (with two phases of distance the extra layer serves the rays that run ahead)
            // every 8th phase the barrier doubles as the vote "no ray of the tile has prefix samples left"
            if ((p & 7) == 7) { if (__syncthreads_and(!(takenf < limitf && da < 0.95f) ? 1 : 0)) { tile_done = true; break; } }
            else __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // no DMA may outlive the workgroup's LDS
        i += (int)takenf;
        rem -= (int)takenf;
    };
    if (stage && any_prefix) {
        if (ax_m == 0) staged_march(std::integral_constant<int, 0>{});
        else if (ax_m == 1) staged_march(std::integral_constant<int, 1>{});
        else staged_march(std::integral_constant<int, 2>{});
    } else if (any_prefix) {
        // ---- not staged: the prefix on global taps (positions inside the safe prefix: no bounds tests).  A tile that is
        // not staged is usually one of a few in a frame of staged ones, and its eight wavefronts share ONE compute unit's
        // vector L1 (4 cycles per cache line per gather instruction): what it costs the frame is its own duration, i.e.
        // gather instructions per sample x ~1000 samples.  So it gathers from the APRON copy like the batched kernel
        // (vr_kernels.hip: raymarch_tri_kernel) -- the x1 tap is the x0 tap's next element: four pair loads per sample
        // instead of eight single taps -- through per-axis offset tables built in the LDS the ring does not need (64-bit
        // along z: volumes beyond 4 GiB), with the staged path's clamp-free index rule, and the taps of TS_FB_BATCH
        // consecutive samples are requested together (16 independent loads per lane) before the first is composited.
        // Samples behind the one that ends the ray were fetched for nothing (valid addresses: the safe prefix).
        const int gxy_bytes = ((P.nx + P.ny + 1) * 4 + 7) & ~7;
        const bool fb_tables = gxy_bytes + (P.nz + 1) * 8 <= C::REGION;   // (an axis too long for that: the loop below)
        uint32_t *gx = reinterpret_cast<uint32_t *>(ring), *gy = gx + P.nx;
        uint64_t *gz = reinterpret_cast<uint64_t *>(ring + gxy_bytes);
        if (fb_tables) {
            const uint32_t sz = (uint32_t)sizeof(VoxelT), bnx = (uint32_t)P.bnx;
            const uint64_t bxy = (uint64_t)P.bnx * (uint64_t)P.bny;
            for (int e = (int)threadIdx.x; e < P.nx + P.ny + P.nz + 2; e += TS_THREADS) {
                if (e < P.nx) { const uint32_t ii = (uint32_t)e; gx[e] = ((ii >> 2) * APRON_BRICK_VOXELS + (ii & 3u)) * sz; }
                else if (e < P.nx + P.ny + 1) { const uint32_t jj = (uint32_t)min(e - P.nx, P.ny - 1); gy[e - P.nx] = ((jj >> 2) * bnx * APRON_BRICK_VOXELS + (jj & 3u) * 5u) * sz; }
                else { const uint64_t kk = (uint64_t)min(e - P.nx - P.ny - 1, P.nz - 1); gz[e - P.nx - P.ny - 1] = ((kk >> 2) * bxy * APRON_BRICK_VOXELS + (kk & 3u) * 20u) * sz; }
            }
            __syncthreads();
            auto pair_word = [&](uint64_t off) -> uint32_t {           // (x0, x1): one load, aligned to the voxel size only
                if (sizeof(VoxelT) == 2) { uint32_t w; __builtin_memcpy(&w, src + off, 4); return w; }
                uint16_t w; __builtin_memcpy(&w, src + off, 2); return (uint32_t)w;
            };
            // (no lockstep barriers here: tried -- one per two batches -- and slower, 1.63 -> 1.87 ms at a pose where half the tiles
            // are staged: the unstaged tiles' wavefronts fill the issue slots the staged ones leave at their barriers)
            // Consecutive samples of a ray share taps: when the (x, y) cell did not change, the near z plane of a sample is the far
            // plane of the one before (or the other way round, or the whole cell is the same) -- those pair words are taken from
            // the previous sample when it is composited and their loads are not issued (as in the batched kernel: a fifth of the
            // gathers even at oblique poses, 40 % when the view runs along z).
            uint32_t pxy0 = 0xffffffffu, pxy1 = 0xffffffffu;             // cell of the last sample issued
            int pk0 = -0x40000000;
            uint32_t E0 = 0, E1 = 0, E2 = 0, E3 = 0;                     // pair words of the last sample composited
            // SKIP: a tile that is not staged skips per ray and batch, like the NEAREST kernels -- the dilated cell-max grid at the
            // batch's MIDDLE position bounds every tap of its four samples: their base voxels lie within 1.5 steps of it and the
            // + 1 taps one voxel further, i.e. within the 3x3x3 cells around its cell as long as a step advances less than four voxels
            // per axis (tested here: fb_skip).  The grid is probed ahead, at positions from the affine map (closed form in the sample
            // index: a few ulp off, the margin above is voxels).  A skipped batch takes the shader's four position additions and nothing else.
            // (SIXTEEN batches' probes at a time, issued together and kept as one bit each: probing one -- or four -- batches ahead
            // inside the loop ran 3x SLOWER than sampling, 1.38 -> 4.2 ms with every tile on this path: vector-memory operations
            // complete in order and the compiler cannot count them across the loop's two paths, so every iteration waited for the
            // probe it had just issued, and a skipped batch has nothing to hide that behind)
            // (round-5 advisor: the probes are RE-ANCHORED at the marched position every FB_AHEAD batches -- V0 is evaluated from the
            // current position, k counts from there -- so the closed form's error is bounded by 64 steps' worth, k * N * 2^-23 voxels =
            // a quarter voxel on a 32768-voxel axis, instead of growing over the whole prefix; and a step may advance at most THREE voxels
            // per axis: 1.5 steps + the tap's 1.5 voxels stay 2 voxels inside the 8 the dilated grid covers beyond the probe's cell)
            float V0x = 0.0f, V0y = 0.0f, V0z = 0.0f, dVx = 0.0f, dVy = 0.0f, dVz = 0.0f;   // voxel coordinates of the position k steps from the anchor: V0 + k dV
            bool fb_skip = false;
            const float rSx = 1.0f / Sx, rSy = 1.0f / Sy, rSz = 1.0f / Sz;   // (POW2: exact)
            auto fb_anchor = [&]() { voxel_float(POW2 ? Qx * rSx : qx, POW2 ? Qy * rSy : qy, POW2 ? Qz * rSz : qz, V0x, V0y, V0z); };
            if (skip_on) {
                const float bx = POW2 ? Qx * rSx : qx, by = POW2 ? Qy * rSy : qy, bz = POW2 ? Qz * rSz : qz;
                float ex, ey, ez;
                fb_anchor();
                voxel_float(bx + dsx, by + dsy, bz + dsz, ex, ey, ez);
                dVx = ex - V0x; dVy = ey - V0y; dVz = ez - V0z;
                fb_skip = fmaxf(fmaxf(fabsf(dVx), fabsf(dVy)), fabsf(dVz)) <= 3.0f;
            }
            auto fb_probe = [&](float k) -> uint32_t {                   // grid value at the position k steps from the anchor
                const float fx = __builtin_fmaf(k, dVx, V0x), fy = __builtin_fmaf(k, dVy, V0y), fz = __builtin_fmaf(k, dVz, V0z);
                const int ci = min((int)fmaxf(fx - 0.5f, 0.0f) >> 3, P.cnx - 1), cj = min((int)fmaxf(fy - 0.5f, 0.0f) >> 3, P.cny - 1);
                const int ck = min((int)fmaxf(fz - 0.5f, 0.0f) >> 3, P.cnz - 1);
                return (uint32_t)skip_grid[((size_t)ck * (size_t)P.cny + (size_t)cj) * (size_t)P.cnx + (size_t)ci];
            };
            constexpr float FB_MID = 0.5f * (float)(TS_FB_BATCH - 1);
            constexpr int FB_AHEAD = 16;                                 // batches probed together
            uint32_t emask = 0u;                                         // bit j: batch j from here is empty
            int eleft = 0;                                               // batches the mask still covers
            float kf = 0.0f;                                             // samples taken since the probes' anchor
            while (rem >= TS_FB_BATCH && !done) {
                if (fb_skip) {
                    if (eleft == 0) {
                        fb_anchor(); kf = 0.0f;                          // the anchor moves to where the ray stands now
                        emask = 0u;
#pragma unroll
                        for (int j = 0; j < FB_AHEAD; j++)
                            emask |= ((int)fb_probe(kf + (float)(j * TS_FB_BATCH) + FB_MID) <= P.skip_thresh ? 1u : 0u) << j;
                        eleft = FB_AHEAD;
                    }
                    // (wavefront-uniform: with the decision per lane the lanes of a wavefront fell out of step -- the taps of one
                    // gather spread over more cache lines, the loop body ran for a few lanes at a time: 15 % fewer gather instructions
                    // for 48 % fewer samples, 24 % MORE vector instructions, 1.38 -> 5.1 ms with every tile on this path)
                    const bool empty = __all((emask & 1u) != 0u ? 1 : 0) != 0;
                    emask >>= 1; eleft--;
                    kf += (float)TS_FB_BATCH;
                    if (empty) {
                        if (da >= 0.95f) { done = true; break; }
#pragma unroll
                        for (int u = 0; u < TS_FB_BATCH; u++) advance();
                        i += TS_FB_BATCH; rem -= TS_FB_BATCH;
                        pxy0 = pxy1 = 0xffffffffu; pk0 = -0x40000000;   // (no taps to hand on)
                        continue;
                    }
                }
                uint32_t pw[TS_FB_BATCH][4];
                float wt[TS_FB_BATCH][3];
                int ru[TS_FB_BATCH];                                     // 0 load both planes, 1 near = previous far, 2 same cell, 3 far = previous near
#pragma unroll
                for (int u = 0; u < TS_FB_BATCH; u++) {
                    float fx, fy, fz;
                    scaled_here(fx, fy, fz);
                    const float ux = fmaxf(fx - 0.5f, 0.0f), uy = fmaxf(fy - 0.5f, 0.0f), uz = fmaxf(fz - 0.5f, 0.0f);
                    const int i0 = (int)ux, j0 = (int)uy, k0 = (int)uz;   // == floor: u >= 0; i0 <= n - 1 inside the safe prefix
                    wt[u][0] = __builtin_amdgcn_fractf(ux); wt[u][1] = __builtin_amdgcn_fractf(uy); wt[u][2] = __builtin_amdgcn_fractf(uz);
                    const uint32_t xy0 = gx[i0] + gy[j0], xy1 = gx[i0] + gy[j0 + 1];
                    const uint64_t z0 = gz[k0], z1 = gz[k0 + 1];
                    const int dk = k0 - pk0;
                    const int r = (xy0 == pxy0 && xy1 == pxy1) ? (dk == 0 ? 2 : (dk == 1 ? 1 : (dk == -1 ? 3 : 0))) : 0;
                    ru[u] = r;
                    pw[u][0] = pw[u][1] = pw[u][2] = pw[u][3] = 0u;
                    if (r == 0 || r == 3) { pw[u][0] = pair_word(z0 + xy0); pw[u][1] = pair_word(z0 + xy1); }
                    if (r == 0 || r == 1) { pw[u][2] = pair_word(z1 + xy0); pw[u][3] = pair_word(z1 + xy1); }
                    pxy0 = xy0; pxy1 = xy1; pk0 = k0;
                    advance();
                }
#pragma unroll
                for (int u = 0; u < TS_FB_BATCH; u++) {
                    if (da >= 0.95f) { done = true; break; }
                    constexpr uint32_t VM = sizeof(VoxelT) == 1 ? 0xffu : 0xffffu;
                    constexpr int VS = sizeof(VoxelT) == 1 ? 8 : 16;
                    const int r = ru[u];
                    const uint32_t w0 = (r == 0 || r == 3) ? pw[u][0] : (r == 1 ? E2 : E0), w1 = (r == 0 || r == 3) ? pw[u][1] : (r == 1 ? E3 : E1);
                    const uint32_t w2 = (r == 0 || r == 1) ? pw[u][2] : (r == 2 ? E2 : E0), w3 = (r == 0 || r == 1) ? pw[u][3] : (r == 2 ? E3 : E1);
                    E0 = w0; E1 = w1; E2 = w2; E3 = w3;
                    const uint32_t tv[8] = {w0 & VM, w0 >> VS, w1 & VM, w1 >> VS, w2 & VM, w2 >> VS, w3 & VM, w3 >> VS};
                    float c, cg = 0.0f, cb = 0.0f, a;
                    shade(tv, wt[u][0], wt[u][1], wt[u][2], c, cg, cb, a);
                    accumulate(c, cg, cb, a);
                    i++; rem--;
                }
            }
        }
        while (rem > 0 && !done) {
            if (da >= 0.95f) { done = true; break; }
            float fx, fy, fz, ax, ay, az, c, cg = 0.0f, cb = 0.0f, a;
            uint32_t tv[8];
            scaled_here(fx, fy, fz);
            taps_global(fx, fy, fz, tv, ax, ay, az);
            shade(tv, ax, ay, az, c, cg, cb, a);
            accumulate(c, cg, cb, a);
            advance();
            i++; rem--;
        }
    }

    // ---- checked tail: the shader's loop, literally (also finishes any prefix sample the phases left behind)
    float tsx = dsx, tsy = dsy, tsz = dsz;
    if (POW2) {
        qx = Qx / Sx; qy = Qy / Sy; qz = Qz / Sz;                        // exact: S is a power of two
        tsx = dSx / Sx; tsy = dSy / Sy; tsz = dSz / Sz;
    }
    if (hit && !done) {
        for (; i < P.max_steps; i++)
            if (checked_step(qx, qy, qz, tsx, tsy, tsz)) break;
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MODE >= 2) store_pixel(P, fb, pix, drgb, dg, db, da);
    else if (MODE == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    VR_TSLAB_STAT(      // per-tile statistics instead of the fetch counts of the tile's first pixels (tools/tslab_stats.py)
    if (spp && threadIdx.x == 0) { spp[pix] = 0x80000000u | (stage ? 1u : 0u) | (st_reason << 1) | ((unsigned)RZ << 4) | ((unsigned)(LSH == 1) << 7) | ((unsigned)min(RA * RB, 255) << 8) | ((unsigned)min(n_phases, 4095) << 16); return; }
    // shader-clock ticks and 100 MHz wall ticks of the staged loop (threads 1, 2), iterations and samples of wavefront 0 (threads 3, 4)
    if (spp && threadIdx.x == 1) { spp[pix] = (uint32_t)(clock64() - st_clk0); return; }
    if (spp && threadIdx.x == 2) { spp[pix] = (uint32_t)(wall_clock64() - st_wall0); return; }
    if (spp && threadIdx.x == 3) { spp[pix] = st_iters; return; }
    if (spp && threadIdx.x == 4) { spp[pix] = st_samples; return; }
    if (spp && threadIdx.x == 5) { spp[pix] = (uint32_t)(st_clk0 - st_entry); return; }     // set-up: ray, head, plan, tables
    if (spp && threadIdx.x == 6) { spp[pix] = st_skipped | (st_runs << 16); return; })      // SKIP: phases crossed in empty runs, runs
    VR_TSLAB_CHK(if (spp) { spp[pix] = 0x40000000u | ((stage && any_prefix) ? 0x20000000u : 0u) | min(chk_violations, 0xffffu); return; })
    if (spp) spp[pix] = (uint32_t)i;
}

// ------------------------------------------------------------------ dispatch
// One translation unit per (voxel type, workgroup shape): VR_TSLAB_TU = 0 / 4 / 6 / 7: u8; 1 .. 3, 5, 8, 9: u16 (below)
template <typename VoxelT, int NW, int LDSKB, bool PERM, int TW, int DIVTC, int VIEW, bool POW2, int MODE, bool SKIP>
static hipError_t launch_tslab2(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                uint32_t *spp, hipStream_t st)
{
    static_assert((NW == 8 && (TW == 4 || TW == 2)) || (NW == 4 && TW == 2), "tile tables exist for 32x16-, 16x32- and 16x16-pixel tiles");
    const uint32_t *table = NW == 4 ? L.tile_table_small : (TW == 2 ? L.tile_table_tall : L.tile_table);
    const uint32_t blocks = NW == 4 ? L.tile_table_small_blocks : (TW == 2 ? L.tile_table_tall_blocks : L.tile_table_blocks);
    int no_stage = L.tri_slab == 2 ? 1 : 0;
    VR_TSLAB_CHK(if (std::getenv("VR_TSLAB_SABOTAGE") != nullptr) no_stage = 3;)   // checked build only: the plan guard's negative control
    hipLaunchKernelGGL((raymarch_tslab_kernel<VoxelT, DIVTC, VIEW, POW2, MODE, NW, LDSKB, PERM, TW, SKIP>), dim3(blocks), dim3(64 * NW), 0, st, P,
                       (const VoxelT *)vol, (const uint8_t *)L.apron, (const uint8_t *)L.apron_y, (const uint8_t *)L.apron_x, tf, fb, spp, table,
                       no_stage, SKIP ? L.skip_grid : nullptr);
    return hipGetLastError();
}

// the SKIP instances run only the launches that asked for empty-space skipping (host: refreshSkipGrid): every other launch
// keeps the phase loop without the layer flags
template <typename VoxelT, int NW, int LDSKB, bool PERM, int TW, int DIVTC, int VIEW, bool POW2, int MODE>
static hipError_t launch_tslab(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                               uint32_t *spp, hipStream_t st)
{
    if (P.skip_empty != 0 && L.skip_grid != nullptr)
        return launch_tslab2<VoxelT, NW, LDSKB, PERM, TW, DIVTC, VIEW, POW2, MODE, true>(P, L, vol, tf, fb, spp, st);
    return launch_tslab2<VoxelT, NW, LDSKB, PERM, TW, DIVTC, VIEW, POW2, MODE, false>(P, L, vol, tf, fb, spp, st);
}

template <typename VoxelT, int NW, int LDSKB, bool PERM, int TW, int VIEW, int MODE>
static hipError_t dispatch_tslab3(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                  uint32_t *spp, hipStream_t st)
{
    const bool pow2 = L.pow2_dims != 0 && L.divmode_tc == DIV_UNIT;
    if (L.divmode_tc == DIV_CERT) return launch_tslab<VoxelT, NW, LDSKB, PERM, TW, DIV_CERT, VIEW, false, MODE>(P, L, vol, tf, fb, spp, st);
    if (pow2) return launch_tslab<VoxelT, NW, LDSKB, PERM, TW, DIV_UNIT, VIEW, true, MODE>(P, L, vol, tf, fb, spp, st);
    return launch_tslab<VoxelT, NW, LDSKB, PERM, TW, DIV_UNIT, VIEW, false, MODE>(P, L, vol, tf, fb, spp, st);
}

template <typename VoxelT, int NW, int LDSKB, bool PERM, int TW = 4>
static hipError_t dispatch_tslab(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                 uint32_t *spp, hipStream_t st)
{
    const int view = P.view_top == 1 ? 1 : (P.view_bottom == 1 ? 2 : 0);
    const int mode = (L.mip ? 1 : 0) + (P.tf_len > 1 ? 2 : 0);
#define VR_TSLAB_M(VW)                                                                                         \
    switch (mode) {                                                                                            \
    case 0: return dispatch_tslab3<VoxelT, NW, LDSKB, PERM, TW, VW, 0>(P, L, vol, tf, fb, spp, st);                \
    case 1: return dispatch_tslab3<VoxelT, NW, LDSKB, PERM, TW, VW, 1>(P, L, vol, tf, fb, spp, st);                \
    case 2: return dispatch_tslab3<VoxelT, NW, LDSKB, PERM, TW, VW, 2>(P, L, vol, tf, fb, spp, st);                \
    default: return dispatch_tslab3<VoxelT, NW, LDSKB, PERM, TW, VW, 3>(P, L, vol, tf, fb, spp, st);               \
    }
    if (view == 0) { VR_TSLAB_M(0) }
    if (view == 1) { VR_TSLAB_M(1) }
    VR_TSLAB_M(2)
#undef VR_TSLAB_M
}

#define VR_TSLAB_ARGS const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb, uint32_t *spp, hipStream_t st
// By workgroup shape (LaunchConfig::tri_slab; vr_set_kernel_variant 6 .. 10):
//   tri_slab 1 (2 = staging off): 32x16-pixel tiles, 80 KiB, order-0 copy, whole layers (8- and 16-bit volumes)
//   tri_slab 3: 16-bit volumes, the same tiles with the per-axis copies, layer thickness per tile (half layers where whole ones do not fit)
//   tri_slab 4: 16x32-pixel tiles (two wavefronts wide, four tall), rows for the tiles that fit no other way; 16-bit volumes with
//               the per-axis copies and the thickness per tile, 8-bit ones with whole layers
//   tri_slab 5: the shape of 1 on 53 KiB, three workgroups per CU (8- and 16-bit volumes)
//   tri_slab 6: 16x16-pixel tiles, four wavefronts, 40 KiB: four workgroups per CU (round 6; 16-bit volumes with the per-axis copies and the
//               thickness per tile when those copies are resident, whole layers from the order-0 copy otherwise)
// (round 4 also measured, and did not keep: whole layers only on a CU's whole LDS -- 32x16 tiles 1.96-2.13 ms over the orbit poses,
// 32x32 tiles with 16 wavefronts 1.64-3.6 --; half layers on 32x32 tiles / 160 KiB 1.63-1.89, off-axis 2.73; half layers on 32x16 tiles
// with the whole LDS 1.90-2.06, off-axis 2.39: behind the two shapes above at every pose once the 16x32 tiles had rows)
hipError_t launch_tslab_u16_half(VR_TSLAB_ARGS);
hipError_t launch_tslab_u16_halftall(VR_TSLAB_ARGS);
hipError_t launch_tslab_u8_tall(VR_TSLAB_ARGS);
hipError_t launch_tslab_u16_three(VR_TSLAB_ARGS);
hipError_t launch_tslab_u8_three(VR_TSLAB_ARGS);
hipError_t launch_tslab_u8_small(VR_TSLAB_ARGS);
hipError_t launch_tslab_u16_small(VR_TSLAB_ARGS);
hipError_t launch_tslab_u16_smallperm(VR_TSLAB_ARGS);

#ifndef VR_TSLAB_TU
#define VR_TSLAB_TU -1
#endif
#if VR_TSLAB_TU == 0 || VR_TSLAB_TU == -1
hipError_t launch_raymarch_slab_tri_u8(VR_TSLAB_ARGS)
{
    if (L.tri_slab == 4 && L.tile_table_tall != nullptr) return launch_tslab_u8_tall(P, L, vol, tf, fb, spp, st);
    if (L.tri_slab == 5) return launch_tslab_u8_three(P, L, vol, tf, fb, spp, st);
    if (L.tri_slab == 6 && L.tile_table_small != nullptr) return launch_tslab_u8_small(P, L, vol, tf, fb, spp, st);
    return dispatch_tslab<uint8_t, 8, 80, false>(P, L, vol, tf, fb, spp, st);
}
#endif
#if VR_TSLAB_TU == 1 || VR_TSLAB_TU == -1
hipError_t launch_raymarch_slab_tri_u16(VR_TSLAB_ARGS)
{
    const bool perm_ok = L.apron_y != nullptr && L.apron_x != nullptr;
    if (L.tri_slab == 3 && perm_ok) return launch_tslab_u16_half(P, L, vol, tf, fb, spp, st);
    if (L.tri_slab == 4 && perm_ok && L.tile_table_tall != nullptr) return launch_tslab_u16_halftall(P, L, vol, tf, fb, spp, st);
    if (L.tri_slab == 5) return launch_tslab_u16_three(P, L, vol, tf, fb, spp, st);
    if (L.tri_slab == 6 && L.tile_table_small != nullptr)
        return perm_ok ? launch_tslab_u16_smallperm(P, L, vol, tf, fb, spp, st) : launch_tslab_u16_small(P, L, vol, tf, fb, spp, st);
    return dispatch_tslab<uint16_t, 8, 80, false>(P, L, vol, tf, fb, spp, st);
}
#endif
#if VR_TSLAB_TU == 2 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u16_half(VR_TSLAB_ARGS) { return dispatch_tslab<uint16_t, 8, 80, true>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 3 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u16_halftall(VR_TSLAB_ARGS) { return dispatch_tslab<uint16_t, 8, 80, true, 2>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 4 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u8_tall(VR_TSLAB_ARGS) { return dispatch_tslab<uint8_t, 8, 80, false, 2>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 5 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u16_three(VR_TSLAB_ARGS) { return dispatch_tslab<uint16_t, 8, 53, false>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 6 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u8_three(VR_TSLAB_ARGS) { return dispatch_tslab<uint8_t, 8, 53, false>(P, L, vol, tf, fb, spp, st); }
#endif
// round 6: 16x16-pixel tiles on FOUR wavefronts and 40 KiB, four workgroups per CU (tri_slab 6): what pays where a tile's brick layers are
// small against the ring -- volumes up to ~512^3, launches that leave workgroup slots empty -- because twice as many tiles end the launch
// with a shorter longest tile and barriers join four wavefronts instead of eight (cfg1 shape 0.138 -> 0.118 ms, cfg2 shape 0.402 -> 0.381);
// a 1024^3 volume's layers do not fit 40 KiB three deep (default pose 1.12 -> 1.22 ms, off-axis 1.55 -> 2.74: 40 % of the tiles on global
// taps), so the work model offers the shape only below that size.  (Measured and dropped: the same tiles on 26 KiB, six workgroups per
// CU at 80 VGPRs -- cfg1 0.1175 vs 0.1186 ms, cfg2 0.3808 vs 0.3810: nothing for another translation unit.)
#if VR_TSLAB_TU == 7 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u8_small(VR_TSLAB_ARGS) { return dispatch_tslab<uint8_t, 4, 40, false, 2>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 8 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u16_small(VR_TSLAB_ARGS) { return dispatch_tslab<uint16_t, 4, 40, false, 2>(P, L, vol, tf, fb, spp, st); }
#endif
#if VR_TSLAB_TU == 9 || VR_TSLAB_TU == -1
hipError_t launch_tslab_u16_smallperm(VR_TSLAB_ARGS) { return dispatch_tslab<uint16_t, 4, 40, true, 2>(P, L, vol, tf, fb, spp, st); }
#endif

// one empty kernel per translation unit (like vr_kernels.hip's warm_kernel_*): launching it makes the runtime inflate and
// load that unit's code object.  launch_warm_tslab() touches all ten when TRILINEAR is selected (vr_set_filter /
// vr_load_shader), so neither the first TRILINEAR frame nor the first frame of a shape the measured choice tries pays it.
#if VR_TSLAB_TU >= 0
#define VR_TSLAB_CAT2(a, b) a##b
#define VR_TSLAB_CAT(a, b) VR_TSLAB_CAT2(a, b)
__global__ void VR_TSLAB_CAT(warm_kernel_tslab, VR_TSLAB_TU)() {}
hipError_t VR_TSLAB_CAT(launch_warm_tslab_tu, VR_TSLAB_TU)(hipStream_t st)
{
    hipLaunchKernelGGL(VR_TSLAB_CAT(warm_kernel_tslab, VR_TSLAB_TU), dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
#endif
#if VR_TSLAB_TU == 0
hipError_t launch_warm_tslab_tu1(hipStream_t st);
hipError_t launch_warm_tslab_tu2(hipStream_t st);
hipError_t launch_warm_tslab_tu3(hipStream_t st);
hipError_t launch_warm_tslab_tu4(hipStream_t st);
hipError_t launch_warm_tslab_tu5(hipStream_t st);
hipError_t launch_warm_tslab_tu6(hipStream_t st);
hipError_t launch_warm_tslab_tu7(hipStream_t st);
hipError_t launch_warm_tslab_tu8(hipStream_t st);
hipError_t launch_warm_tslab_tu9(hipStream_t st);
hipError_t launch_warm_tslab(hipStream_t st)
{
    hipError_t e = launch_warm_tslab_tu0(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu1(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu2(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu3(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu4(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu5(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu6(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu7(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu8(st);
    if (e == hipSuccess) e = launch_warm_tslab_tu9(st);
    return e;
}
#elif VR_TSLAB_TU == -1
hipError_t launch_warm_tslab(hipStream_t) { return hipSuccess; }      // single-unit build: nothing to pre-load separately
#endif

}  // namespace vr

