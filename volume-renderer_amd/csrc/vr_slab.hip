// vr_slab.hip -- the LDS-staged ray-march kernel for gfx950 (MI355X / CDNA4).
//
// Same arithmetic as raymarch_fast_kernel (vr_kernels.hip): NEAREST sampling, iterative position
// accumulation, the shader's operations in the shader's order (VolumeRenderer.cs:104-173), safe
// prefix + checked tail.  What changes is WHERE the prefix gathers read from.
//
// Why: round 1's counters showed the scattered 1/2-byte gathers saturating the vector L1's
// address/tag path (41.8 tag look-ups per wave-gather, TA stalled 97 % of its busy time), not HBM.
// Here the workgroup (8 wavefronts = a 32x16-pixel tile, all rays in lockstep by sample index)
// keeps the bricks its rays are about to cross in LDS:
//   * the bricked volume (4x4x4-voxel bricks: 64 B for u8, 96 B in the 12-bit packed copy) is
//     fetched brick by brick with LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per
//     wave-instruction, per-lane source address, no VGPR staging) -- whole 16-B chunks, every
//     byte of a fetched line is used, ~20x fewer tag look-ups per sample;
//   * LDS holds a TORUS of RA x RB x RZ brick slots: brick (i,j,k) lives in slot
//     (i mod R_i, j mod R_j, k mod R_k), so the LDS offset of a voxel is still the sum of three
//     per-axis table entries (16-bit, built per workgroup) and the gather costs exactly what a
//     gather through the fast kernel's address tables costs, with ds_read_u8 in place of
//     buffer_load_ubyte;
//   * the bricks are streamed LAYER by layer along the tile's entry-face normal m (the axis on
//     which the tile's sample positions at equal sample index spread least): layer L is requested
//     a few phases (4 samples each) before the first ray reaches it and its slots are re-used
//     RZ layers later; inside a layer only the rectangle the rays can touch while the layer is
//     alive is fetched;
//   * which layers and which rectangle: from linear bounds.  At every epoch (16 phases) the exact
//     positions of the live rays are min/max-reduced over the workgroup ("anchor"); every ray
//     advances by a constant per-step vector, so  anchor_min + j*step_min - margin  <=  position
//     after j more steps  <=  anchor_max + j*step_max + margin  for every live ray.  One
//     wavefront turns the bounds into a 16-phase load plan (one lane per phase) and publishes it
//     in LDS; the other wavefronts only read their phase's entry;
//   * when the bounds do not fit the torus (a tile straddling a box edge, a grazing face) the
//     torus is re-dimensioned for the current footprint (cold start), and when nothing fits the
//     epoch runs on plain global gathers -- same arithmetic, only slower.
// Correctness does not depend on the plan being tight, only on it being a superset; every
// parity test compares whole frames bit for bit, so a brick missing from LDS cannot hide.
//
// No MFMA: the path is a gather + a 25-flop recurrence per sample.
#include "vr_device.h"
#include "vr_lds_dma.h"

namespace vr {

constexpr int SLAB_NW = 8, SLAB_THREADS = 64 * SLAB_NW;   // one 32x16-pixel tile per workgroup, like the fast kernel
#if defined(VR_EXPERIMENTS) && defined(VR_X_PHASE)
constexpr int SLAB_PHASE = VR_X_PHASE;
#else
constexpr int SLAB_PHASE = 8;          // samples per phase = per workgroup barrier (two brick layers at one voxel per step; 4: 0.85 ms, 8: 0.68 ms on cfg3)
#endif
#if defined(VR_EXPERIMENTS) && defined(VR_X_EPOCH)
constexpr int SLAB_EPOCH = VR_X_EPOCH;
#else
constexpr int SLAB_EPOCH = 16;         // phases per load plan / between two anchors
#endif
#if defined(VR_EXPERIMENTS) && defined(VR_X_LA)
constexpr int SLAB_LA = VR_X_LA;
#else
constexpr int SLAB_LA = 2;             // phases of prefetch distance asked for (the ring depth may allow less)
#endif
#if defined(VR_EXPERIMENTS) && defined(VR_X_LDSKB)
constexpr int SLAB_LDS_BYTES = VR_X_LDSKB * 1024 - 512;
#else
constexpr int SLAB_LDS_BYTES = 80 * 1024 - 512;           // two workgroups per CU (160 KiB)
#endif
constexpr float SLAB_MARGIN = 0.0625f; // voxels: covers the rounding of the iterated positions over an epoch (< 2^-7 voxel for N <= 4096)
constexpr int SLAB_MAX_PIECES = 2;     // 1-KiB pieces per wavefront per layer

// LDS budget of one (voxel type, packing, mode) family
template <typename VoxelT, bool PK12, int MODE>
struct SlabCfg {
    static constexpr int SLOT = PK12 ? 96 : 64 * (int)sizeof(VoxelT);   // bytes per brick slot
    static constexpr int CH = SLOT / 16;                                 // 16-byte chunks per slot
    static constexpr int LUT_ENTRIES = sizeof(VoxelT) == 1 ? 256 : 4096;
    // grey modes: u8 -> 256 (c, a) pairs; 12-bit -> NO table: the window map is computed per sample with the
    // shader's own operations (certified quotient: 6 VALU instead of one more LDS round trip) and the
    // 32 KiB it would take go to the ring, which is what decides how far ahead the loads can run;
    // transfer function: 256 RGBA entries + one index byte per window value
    static constexpr bool HAS_LUT = MODE >= 2 || sizeof(VoxelT) == 1;
    static constexpr int LUT_BYTES = MODE >= 2 ? 4096 + LUT_ENTRIES : (sizeof(VoxelT) == 1 ? LUT_ENTRIES * 8 : 16);
    static constexpr int TAB_ENTRIES = sizeof(VoxelT) == 1 ? 6144 : 3072;   // nx + ny + nz
    static constexpr int MISC_BYTES = 1024;
    // torus tables: 16-bit entries while the ring stays below 64 KiB (two workgroups per CU), 32-bit beyond
    static constexpr bool WIDE_TAB = SLAB_LDS_BYTES - LUT_BYTES - TAB_ENTRIES * 2 - MISC_BYTES > 65535 + 16384;
    static constexpr int TAB_BYTES = TAB_ENTRIES * (WIDE_TAB ? 4 : 2);
    static constexpr int RING_RAW = SLAB_LDS_BYTES - LUT_BYTES - TAB_BYTES - MISC_BYTES;
    static constexpr int RING_BYTES = (!WIDE_TAB && RING_RAW > 65535 ? 65535 : RING_RAW) / SLOT * SLOT;
    static constexpr int CAP = RING_BYTES / SLOT;                           // brick slots
    static constexpr int LAYER_SLOTS_MAX = SLAB_MAX_PIECES * SLAB_NW * 64 / CH;
};

template <typename VoxelT, bool PK12, int DIVTC, int VIEW, bool POW2, bool NOCLAMP, int MODE>
__global__ __launch_bounds__(SLAB_THREADS, 4) void raymarch_slab_kernel(const FrameParams P,
                                                                        const VoxelT *__restrict__ vol,
                                                                        const uint8_t *__restrict__ src,
                                                                        const float4 *__restrict__ tf,
                                                                        float4 *__restrict__ fb,
                                                                        uint32_t *__restrict__ spp,
                                                                        const uint32_t *__restrict__ tile_table)
{
    using C = SlabCfg<VoxelT, PK12, MODE>;
    static_assert(!PK12 || sizeof(VoxelT) == 2, "the 12-bit packed copy belongs to u16 volumes");
    static_assert(PK12 || sizeof(VoxelT) == 1, "raw staging is built for u8 volumes (64-byte bricks)");
    constexpr int BATCH = SLAB_PHASE;
    __shared__ __attribute__((aligned(16))) uint8_t ring[C::RING_BYTES];
    __shared__ __attribute__((aligned(16))) float lut[C::LUT_BYTES / 4];
    using TabT = typename std::conditional<C::WIDE_TAB, uint32_t, uint16_t>::type;
    __shared__ TabT tab[C::TAB_ENTRIES];
    __shared__ __attribute__((aligned(16))) uint4 plan[SLAB_EPOCH];
    __shared__ float red[SLAB_NW][8];
    __shared__ int geo[8];
    static_assert(SLAB_EPOCH <= 64 && sizeof(plan) + sizeof(red) + sizeof(geo) <= C::MISC_BYTES, "LDS budget");

    const uint32_t t = tile_table[blockIdx.x];
    if (t == 0xffffffffu) return;                                       // padding block
    const unsigned tx = t & 0xffffu, ty = t >> 16;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(tx * kFastTileW + (wave & 3u) * 8u + (lane & 7u));
    const int ly = (int)(ty * kFastTileH + (wave >> 2) * 8u + (lane >> 3));
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
    if (!__syncthreads_or(hit ? 1 : 0)) {                               // no ray of the tile enters the volume
        if (in_image) {
            const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
            store_pixel(P, fb, pix, 0.0f, 0.0f, 0.0f, 0.0f);
            if (spp) spp[pix] = 0u;
        }
        return;
    }
    // ---- classification table, with the shader's own operations (see raymarch_fast_kernel)
    if (C::HAS_LUT) {
        const int n = P.max_val - P.min_val + 1;
        for (int e = (int)threadIdx.x; e < n; e += SLAB_THREADS) {
            const float s = (float)(P.min_val + e);
            const float v = div_cert(s - P.fmin, P.fden, P.rden);
            if (MODE >= 2) {
                int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                idx = clampi(idx, 0, P.tf_len - 1);
                reinterpret_cast<uint8_t *>(lut)[4096 + e] = (uint8_t)idx;
            } else {
                const float a = v * P.alpha_scale;
                lut[2 * e + 0] = v * a; lut[2 * e + 1] = a;
            }
        }
        if (MODE >= 2) {
            for (int e = (int)threadIdx.x; e < P.tf_len; e += SLAB_THREADS) {
                const float4 q = tf[e];
                const float a = q.w * P.alpha_scale;
                if (MODE == 3) { lut[4 * e + 0] = q.x * P.alpha_scale; lut[4 * e + 1] = q.y * P.alpha_scale; lut[4 * e + 2] = q.z * P.alpha_scale; }
                else { lut[4 * e + 0] = q.x * a; lut[4 * e + 1] = q.y * a; lut[4 * e + 2] = q.z * a; }
                lut[4 * e + 3] = a;
            }
        }
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
    const int nb = k_safe / BATCH;                                      // phases of this ray's prefix

    // marching units: voxels for POW2 (exact, see raymarch_fast_kernel), box units otherwise
    const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
    float Qx = qx * Sx, Qy = qy * Sy, Qz = qz * Sz;
    const float dSx = dsx * Sx, dSy = dsy * Sy, dSz = dsz * Sz;
    const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;

    // voxel coordinates (floats, per voxel axis i/j/k) of a position: exact for POW2, within a few
    // ulp otherwise -- only the load plan's bounds use them, behind SLAB_MARGIN
    auto voxel_float = [&](float ax, float ay, float az, float &fx, float &fy, float &fz) {
        if (POW2) {
            const float ux = ax + Hx, uy = ay + Hy, uz = Sz - (az + Hz);
            fx = ux; fy = uy; fz = uz;
            if (VIEW == 1) { fy = Sz - uz; fz = uy; }
            else if (VIEW == 2) { fy = uz; fz = Sy - uy; }
        } else {
            const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
            const float uz = 1.0f - (az + P.half[2]) * P.rext[2];
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = 1.0f - uz; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
        }
    };
    const float INF = __builtin_inff();
    // ---- per-step advance of the rays, min/max over the workgroup (constant for the whole tile)
    float dmn0, dmn1, dmn2, dmx0, dmx1, dmx2;
    int nbmax;
    {
        float f0x, f0y, f0z, f1x, f1y, f1z;
        if (POW2) { voxel_float(Qx, Qy, Qz, f0x, f0y, f0z); voxel_float(Qx + dSx, Qy + dSy, Qz + dSz, f1x, f1y, f1z); }
        else { voxel_float(qx, qy, qz, f0x, f0y, f0z); voxel_float(qx + dsx, qy + dsy, qz + dsz, f1x, f1y, f1z); }
        const bool live = nb > 0;
        const float ex = f1x - f0x, ey = f1y - f0y, ez = f1z - f0z;
        const float a0 = wave_min_f(live ? ex : INF), a1 = wave_min_f(live ? ey : INF), a2 = wave_min_f(live ? ez : INF);
        const float b0 = wave_max_f(live ? ex : -INF), b1 = wave_max_f(live ? ey : -INF), b2 = wave_max_f(live ? ez : -INF);
        const float nm = wave_max_f((float)nb);
        if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; red[wave][3] = b0; red[wave][4] = b1; red[wave][5] = b2; red[wave][6] = nm; }
        __syncthreads();
        float r0 = INF, r1 = INF, r2 = INF, r3 = -INF, r4 = -INF, r5 = -INF, r6 = 0.0f;
        for (int w = 0; w < SLAB_NW; w++) {
            r0 = fminf(r0, red[w][0]); r1 = fminf(r1, red[w][1]); r2 = fminf(r2, red[w][2]);
            r3 = fmaxf(r3, red[w][3]); r4 = fmaxf(r4, red[w][4]); r5 = fmaxf(r5, red[w][5]); r6 = fmaxf(r6, red[w][6]);
        }
        dmn0 = uniform_f(r0); dmn1 = uniform_f(r1); dmn2 = uniform_f(r2);
        dmx0 = uniform_f(r3); dmx1 = uniform_f(r4); dmx2 = uniform_f(r5);
        nbmax = uniform_i((int)r6);
        __syncthreads();                                                // red[] is reused by the anchors
    }
    const int nbr0 = P.bnx, nbr1 = P.bny, nbr2 = P.bnz;                // bricks per voxel axis
    const uint32_t str0 = 1u, str1 = (uint32_t)P.bnx, str2 = (uint32_t)P.bnx * (uint32_t)P.bny;   // brick index strides

    // ---- torus state (workgroup-uniform)
    int ax_m = 2, ax_a = 0, ax_b = 1, sgn = 1, RA = 1, RB = 1, RZ = 1;
    uint32_t sA = 1, sB = 1, sM = 1;
    uint32_t layer_bytes = 0;
    int pieces = 1;                                                     // 1-KiB pieces per wavefront per layer
    // per-lane loader constants of piece q: torus coordinates (ta, tb) of the slot this lane's 16-byte chunk
    // belongs to, the chunk's index inside the slot, and whether the slot exists
    int ld_ta[SLAB_MAX_PIECES], ld_tb[SLAB_MAX_PIECES], ld_part[SLAB_MAX_PIECES];
    bool ld_ok[SLAB_MAX_PIECES];
#pragma unroll
    for (int q = 0; q < SLAB_MAX_PIECES; q++) { ld_ta[q] = ld_tb[q] = ld_part[q] = 0; ld_ok[q] = false; }
    const uint32_t ring_base = lds_offset_of(ring);

    // (re)build the torus tables and the loader constants for the dimensions in geo[]
    auto adopt_geometry = [&]() {
        ax_m = uniform_i(geo[0]); sgn = uniform_i(geo[1]); RA = uniform_i(geo[2]); RB = uniform_i(geo[3]); RZ = uniform_i(geo[4]);
        ax_a = ax_m == 0 ? 1 : 0; ax_b = ax_m == 2 ? 1 : 2;
        sA = sel3(ax_a, str0, str1, str2); sB = sel3(ax_b, str0, str1, str2); sM = sel3(ax_m, str0, str1, str2);
        layer_bytes = (uint32_t)(RA * RB * C::SLOT);
        pieces = (RA * RB * C::CH + SLAB_NW * 64 - 1) / (SLAB_NW * 64);
        const int na = P.nx + P.ny + P.nz;
        for (int e = (int)threadIdx.x; e < na; e += SLAB_THREADS) {
            int axis, i;
            if (e < P.nx) { axis = 0; i = e; } else if (e < P.nx + P.ny) { axis = 1; i = e - P.nx; } else { axis = 2; i = e - P.nx - P.ny; }
            const int R = axis == ax_a ? RA : (axis == ax_b ? RB : RZ);
            const uint32_t stride = axis == ax_a ? (uint32_t)C::SLOT : (axis == ax_b ? (uint32_t)(RA * C::SLOT) : layer_bytes);
            const uint32_t in = (uint32_t)(i & 3) << (2 * axis);          // element offset inside the brick: x + 4y + 16z
            const uint32_t inb = PK12 ? (3u * in) >> 1 : in;               // bytes (12-bit stream: floor(1.5 e), exact for the even y/z terms)
            tab[e] = (TabT)((uint32_t)((i >> 2) % R) * stride + inb);
        }
#pragma unroll
        for (int q = 0; q < SLAB_MAX_PIECES; q++) {
            const int c = (q * SLAB_NW + (int)wave) * 64 + (int)lane;
            const int slot = c / C::CH;
            ld_part[q] = c - slot * C::CH;
            ld_tb[q] = slot / RA;
            ld_ta[q] = slot - ld_tb[q] * RA;
            ld_ok[q] = slot < RA * RB && q < pieces;
        }
    };

    // ---- the load plan of one epoch, computed by wavefront 0 (lane = phase of the epoch).
    // Progress coordinate of a layer along m: c = sgn * brick index, so "ahead" is always larger.
    float Amin0 = 0, Amin1 = 0, Amin2 = 0, Amax0 = 0, Amax1 = 0, Amax2 = 0;   // anchor (uniform)
    int front = 0;                       // wavefront 0: first layer (progress coordinate) not requested yet
    bool have_ring = false;              // the previous epoch left a consistent ring behind
    auto plan_epoch = [&](int ma, int sg, int ra, int rb, int rz, bool cold, int fcarry, int steps_left, uint4 &entry, int &f_after) -> bool {
        const int a_ = ma == 0 ? 1 : 0, b_ = ma == 2 ? 1 : 2;
        const float MG = SLAB_MARGIN;
        auto lo = [&](int x, float j) { return sel3(x, Amin0, Amin1, Amin2) + j * sel3(x, dmn0, dmn1, dmn2) - MG; };
        auto hi = [&](int x, float j) { return sel3(x, Amax0, Amax1, Amax2) + j * sel3(x, dmx0, dmx1, dmx2) + MG; };
        auto blo = [&](int x, float j1, float j2) { const int b = (int)floorf(fminf(lo(x, j1), lo(x, j2))) >> 2; return clampi(b, 0, sel3(x, nbr0, nbr1, nbr2) - 1); };
        auto bhi = [&](int x, float j1, float j2) { const int b = (int)floorf(fmaxf(hi(x, j1), hi(x, j2))) >> 2; return clampi(b, 0, sel3(x, nbr0, nbr1, nbr2) - 1); };
        auto trail = [&](float j1, float j2) { return sg > 0 ? blo(ma, j1, j2) : -bhi(ma, j1, j2); };
        auto lead = [&](float j1, float j2) { return sg > 0 ? bhi(ma, j1, j2) : -blo(ma, j1, j2); };
        const int i = (int)(lane & (SLAB_EPOCH - 1));
        constexpr float PH = (float)BATCH, LAST = (float)(BATCH - 1);       // a phase = samples j0 .. j0 + LAST
        const float j0 = PH * (float)i;
        const int trail_i = trail(j0, j0 + LAST), lead_i = lead(j0, j0 + LAST);
        const int lead_next = lead(j0 + PH, j0 + PH + LAST);
        const int f_i = min(lead(j0 + PH * SLAB_LA, j0 + PH * SLAB_LA + LAST), trail_i + rz - 1) + 1;
        const int f_p = min(lead(j0 - PH + PH * SLAB_LA, j0 - 1.0f + PH * SLAB_LA), trail(j0 - PH, j0 - 1.0f) + rz - 1) + 1;
        const int fc = cold ? trail(0.0f, LAST) : fcarry;
        const int Fb = i == 0 ? fc : max(fc, f_p);
        const int Fa = max(max(fc, f_i), Fb);
        const int n = Fa - Fb;
        const int keep = min(max(Fa - 1 - lead_next, 0), n);
        bool over = false;
        // phases beyond the end of the longest ray need nothing
        const bool used = BATCH * i < steps_left;
        if ((i > 0 || !cold) && lead_i >= Fb) over = true;              // a layer of this phase was never requested
        if (lead_i >= Fa && i == 0 && cold) over = true;
        if (lead_next >= Fa) over = true;                               // the ring cannot hold what the next phase needs
        if (lead_i - trail_i + 1 > rz) over = true;
        if (n > (cold && i == 0 ? 15 : 6)) over = true;
        int lo_a = 0, lo_b = 0, dda = 0, ddb = 0;
        if (n > 0) {
            const int lastb = sg * (Fa - 1);                             // brick index of the last layer requested now
            float jx;
            if (sg > 0) jx = ((float)(4 * (lastb + 1)) - (sel3(ma, Amin0, Amin1, Amin2) - MG)) / sel3(ma, dmn0, dmn1, dmn2);
            else jx = ((sel3(ma, Amax0, Amax1, Amax2) + MG) - (float)(4 * lastb)) / -sel3(ma, dmx0, dmx1, dmx2);
            if (!(jx < j0 + 600.0f)) over = true;
            const float j2 = fmaxf(jx, j0 + LAST) + 1.0f;
            lo_a = blo(a_, j0, j2); lo_b = blo(b_, j0, j2);
            dda = bhi(a_, j0, j2) - lo_a; ddb = bhi(b_, j0, j2) - lo_b;
            if (dda >= ra || ddb >= rb) over = true;
        }
        const int first_b = sg * Fb;                                      // brick index of the first layer requested now
        const uint32_t la = (uint32_t)(lo_a % ra), lb = (uint32_t)(lo_b % rb), lz = (uint32_t)(max(first_b, 0) % rz);
        entry.x = (uint32_t)first_b;
        entry.y = (uint32_t)lo_a | ((uint32_t)lo_b << 16);
        entry.z = (uint32_t)n | ((uint32_t)keep << 4) | ((uint32_t)dda << 8) | ((uint32_t)ddb << 13) | (la << 18) | (lb << 23);
        entry.w = lz | (cold && i == 0 ? 0x100u : 0u);
        f_after = Fa;
        return used && over;
    };

    // ---- one phase of the gathers: BATCH consecutive samples from the current position
    const TabT *tab_x = tab, *tab_y = tab + P.nx, *tab_z = tab + P.nx + P.ny;
    auto advance_index = [&](int &vi, int &vj, int &vk) {
        if (POW2) {
            const float ux = Qx + Hx, uy = Qy + Hy, uz = Sz - (Qz + Hz);
            float fx = ux, fy = uy, fz = uz;
            if (VIEW == 1) { fy = Sz - uz; fz = uy; }
            else if (VIEW == 2) { fy = uz; fz = Sy - uy; }
            vi = (int)fx; vj = (int)fy; vk = (int)fz;
            Qx += dSx; Qy += dSy; Qz += dSz;
        } else {
            const float ux = div_mode<DIVTC>(qx + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(qy + P.half[1], P.ext[1], P.rext[1]);
            float uz = div_mode<DIVTC>(qz + P.half[2], P.ext[2], P.rext[2]);
            uz = 1.0f - uz;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = 1.0f - uz; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            vi = (int)(tcx * P.fdim[0]); vj = (int)(tcy * P.fdim[1]); vk = (int)(tcz * P.fdim[2]);
            qx += dsx; qy += dsy; qz += dsz;
        }
    };
    // staged gathers come in two halves.  prepare(): positions of the phase's BATCH samples -> LDS offsets
    // through the torus tables (tables and positions do not depend on what the DMA is doing, so this runs
    // BEFORE the phase's barrier, overlapped with the other wavefronts).  fetch(): the ring reads proper, after
    // the barrier (two byte reads for the 12-bit stream: a 2-byte LDS read at an odd address is replayed for
    // ~64 cycles on gfx950, tools/ubench/lds_slab.hip); v then holds the 16 stream bits around the voxel
    auto prepare = [&](uint32_t (&off)[BATCH], uint32_t &nib) {
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            int vi, vj, vk;
            advance_index(vi, vj, vk);
            off[u] = (uint32_t)tab_x[vi] + (uint32_t)tab_y[vj] + (uint32_t)tab_z[vk];
            if (PK12) nib = u == 0 ? ((uint32_t)vi & 1u) << 2 : ((((uint32_t)vi & 1u) << (4 * u + 2)) | nib);
        }
    };
    auto fetch = [&](const uint32_t (&off)[BATCH], uint32_t (&v)[BATCH]) {
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
#if defined(VR_EXPERIMENTS) && defined(VR_X_NOGATHER)
            v[u] = off[u] & 0xfffu; continue;
#endif
            if (PK12) v[u] = (uint32_t)ring[off[u]] | ((uint32_t)ring[off[u] + 1] << 8);
            else v[u] = (uint32_t)ring[off[u]];
        }
    };
    // not staged (an epoch whose footprint does not fit LDS): plain global loads of the voxels
    auto gather_global = [&](uint32_t (&v)[BATCH]) {
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            int vi, vj, vk;
            advance_index(vi, vj, vk);
            v[u] = (uint32_t)vol[VoxelAddr<1, true>::at(P, vi, vj, vk)];
        }
    };
    const int lut_bias = -P.min_val;
    auto classify = [&](uint32_t texel, float &c, float &cg, float &cb, float &a) {
        int tt = (int)texel;
        if (!NOCLAMP) tt = med3_i32(tt, P.min_val, P.max_val);
        const uint32_t e = (uint32_t)(tt + lut_bias);
        if (MODE >= 2) {
            const uint32_t idx = reinterpret_cast<const uint8_t *>(lut)[4096u + e];
            const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
            c = q.x; cg = q.y; cb = q.z; a = q.w;
        } else if (C::HAS_LUT) {
            const float2 ca = reinterpret_cast<const float2 *>(lut)[e];
            c = ca.x; a = ca.y;
        } else {
            // clamp(float(texel), min, max) == float(clamp(texel, min, max)): the conversion is exact and monotone
            const float v = div_cert((float)tt - P.fmin, P.fden, P.rden);   // VolumeRenderer.cs:122-124
            a = v * P.alpha_scale;                                       // :130 / :164
            c = v * a;                                                   // :131
        }
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
        float uz = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);
        uz = 1.0f - uz;
        float tcx = ux, tcy = uy, tcz = uz;
        if (VIEW == 1) { tcy = 1.0f - uz; tcz = uy; }
        else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
        if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) return true;
        const int vi = min((int)(tcx * P.fdim[0]), P.nx - 1);
        const int vj = min((int)(tcy * P.fdim[1]), P.ny - 1);
        const int vk = min((int)(tcz * P.fdim[2]), P.nz - 1);
        float c, cg = 0.0f, cb = 0.0f, a;
        classify((uint32_t)vol[VoxelAddr<1, true>::at(P, vi, vj, vk)], c, cg, cb, a);
        accumulate(c, cg, cb, a);
        x += stx; y += sty; z += stz;
        return false;
    };
    bool head_ended = false;
    if (head > 0) {                                                      // samples 0 .. head - 1
        float x = hqx, y = hqy, z = hqz;
        for (int h = 0; h < head && !head_ended; h++) {
            if (i >= P.max_steps || checked_step(x, y, z, dsx, dsy, dsz)) head_ended = true;
            else i++;
        }
    }
    // compositing of one phase; returns true when the ray terminated (batch early-termination, see the fast kernel)
    auto consume = [&](const uint32_t (&v)[BATCH], uint32_t nib, bool raw) -> bool {
        float c[BATCH], cg[BATCH], cb[BATCH], a[BATCH];
        const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
        float da_last = 0.0f;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const uint32_t texel = (PK12 && !raw) ? __builtin_amdgcn_ubfe(v[u], __builtin_amdgcn_ubfe(nib, 4 * u, 4), 12) : v[u];
            classify(texel, c[u], cg[u], cb[u], a[u]);
        }
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (u == BATCH - 1) da_last = da;
            accumulate(c[u], cg[u], cb[u], a[u]);
        }
        if (da_last < 0.95f) { i += BATCH; return false; }
        drgb = drgb0; dg = dg0; db = db0; da = da0;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (da >= 0.95f) return true;
            accumulate(c[u], cg[u], cb[u], a[u]);
            i++;
        }
        return da >= 0.95f;
    };

    // ---- the phase loop: ONE barrier per phase (it orders "every wavefront has finished reading the layers
    // of phase p-1" before their slots are overwritten, and "the pieces each wavefront waited for have landed"
    // before anybody reads them); every 8th phase the barrier doubles as the vote "all rays finished" (a ray
    // ends early by early ray termination only; the longest prefix, nbmax phases, bounds the loop anyway)
    bool done = head_ended;
    bool fin = nb == 0 || head_ended;
    int b = 0;                                   // phases this ray has marched (== the phase counter while it is live)
    bool epoch_staged = false;
    uint32_t off[BATCH] = {};                    // prepared LDS offsets of the next phase's samples
    uint32_t nib = 0;
    bool prepared = false;
    uint4 entry_next = make_uint4(0u, 0u, 0u, 0u);   // the next phase's plan entry, read before the barrier
#if defined(VR_EXPERIMENTS) && defined(VR_X_STATS)
    unsigned st_epochs = 0, st_cold = 0, st_fallback = 0;
#endif
    for (int p = 0; p < nbmax; p++) {
        if ((p & 7) == 0) { if (__syncthreads_and(fin ? 1 : 0)) break; }
        else __syncthreads();
        const int pe = p & (SLAB_EPOCH - 1);
        if (pe == 0) {
            // ---- anchor: exact positions of the live rays, min/max over the workgroup
            float fx, fy, fz;
            if (POW2) voxel_float(Qx, Qy, Qz, fx, fy, fz); else voxel_float(qx, qy, qz, fx, fy, fz);
            const float a0 = wave_min_f(fin ? INF : fx), a1 = wave_min_f(fin ? INF : fy), a2 = wave_min_f(fin ? INF : fz);
            const float b0 = wave_max_f(fin ? -INF : fx), b1 = wave_max_f(fin ? -INF : fy), b2 = wave_max_f(fin ? -INF : fz);
            if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; red[wave][3] = b0; red[wave][4] = b1; red[wave][5] = b2; }
            __syncthreads();
            if (wave == 0) {
                float r0 = INF, r1 = INF, r2 = INF, r3 = -INF, r4 = -INF, r5 = -INF;
                for (int w = 0; w < SLAB_NW; w++) {
                    r0 = fminf(r0, red[w][0]); r1 = fminf(r1, red[w][1]); r2 = fminf(r2, red[w][2]);
                    r3 = fmaxf(r3, red[w][3]); r4 = fmaxf(r4, red[w][4]); r5 = fmaxf(r5, red[w][5]);
                }
                Amin0 = uniform_f(r0); Amin1 = uniform_f(r1); Amin2 = uniform_f(r2);
                Amax0 = uniform_f(r3); Amax1 = uniform_f(r4); Amax2 = uniform_f(r5);
                const int steps_left = BATCH * (nbmax - p);
                uint4 entry;
                int f_after = 0;
                int status = 0;                                          // 0: staged, ring carried over; 1: staged after a re-dimension (cold); 2: not staged
                bool over = true;
                if (have_ring) over = __any(plan_epoch(ax_m, sgn, RA, RB, RZ, false, front, steps_left, entry, f_after) ? 1 : 0) != 0;
                if (over) {
                    // re-dimension the torus for the footprint of the next few epochs
                    const float spr0 = Amax0 - Amin0, spr1 = Amax1 - Amin1, spr2 = Amax2 - Amin2;
                    const int ma = (spr0 <= spr1 && spr0 <= spr2) ? 0 : (spr1 <= spr2 ? 1 : 2);   // entry-face normal: least spread at equal sample index
                    const float dlo = sel3(ma, dmn0, dmn1, dmn2), dhi = sel3(ma, dmx0, dmx1, dmx2);
                    const int sg = (dlo > 0.0f && dhi > 0.0f) ? 1 : ((dlo < 0.0f && dhi < 0.0f) ? -1 : 0);
                    const float vslow = fminf(fabsf(dlo), fabsf(dhi)), vfast = fmaxf(fabsf(dlo), fabsf(dhi));
#if defined(VR_EXPERIMENTS) && defined(VR_X_HORIZON)
                    constexpr int HORIZON_EPOCHS = VR_X_HORIZON;
#else
                    constexpr int HORIZON_EPOCHS = 1;
#endif
                    const float horizon = fminf((float)(BATCH * SLAB_EPOCH * HORIZON_EPOCHS), (float)steps_left) + (float)(BATCH * SLAB_LA + 8);
                    auto span = [&](int x) { return (sel3(x, Amax0, Amax1, Amax2) - sel3(x, Amin0, Amin1, Amin2)) + horizon * (sel3(x, dmx0, dmx1, dmx2) - sel3(x, dmn0, dmn1, dmn2)) + 2.0f * SLAB_MARGIN; };
                    status = 2;
                    if (sg != 0 && vslow >= 0.125f) {
                        const int a_ = ma == 0 ? 1 : 0, b_ = ma == 2 ? 1 : 2;
                        const float life = (float)(BATCH * (SLAB_LA + 1)) + (4.0f + span(ma)) / vslow + 4.0f;   // steps from request to last use of a layer
                        auto width = [&](int x) {
                            const float drift = fmaxf(fabsf(sel3(x, dmn0, dmn1, dmn2)), fabsf(sel3(x, dmx0, dmx1, dmx2))) * life;
                            return (int)((span(x) + drift) * 0.25f) + 2;
                        };
                        const int ra = min(width(a_), sel3(a_, nbr0, nbr1, nbr2)), rb = min(width(b_), sel3(b_, nbr0, nbr1, nbr2));
                        const int rzmin = (int)((span(ma) + vfast * (float)(2 * BATCH)) * 0.25f) + 2;
                        if (ra <= 31 && rb <= 31 && ra * rb <= C::LAYER_SLOTS_MAX) {
                            const int rz = min(min(C::CAP / (ra * rb), 15), sel3(ma, nbr0, nbr1, nbr2) + 1);   // <= 15: a cold start names all of them in one entry
                            if (rz >= rzmin - 1 && rz >= 2) {        // (rzmin is the worst brick alignment; the per-phase checks of plan_epoch decide)
                                if (!(__any(plan_epoch(ma, sg, ra, rb, rz, true, 0, steps_left, entry, f_after) ? 1 : 0) != 0)) {
                                    status = 1;
                                    if (lane == 0) { geo[0] = ma; geo[1] = sg; geo[2] = ra; geo[3] = rb; geo[4] = rz; }
                                }
                            }
                        }
                    }
                }
                if (status != 2) {
                    if (lane < SLAB_EPOCH) plan[lane] = entry;
                    front = uniform_i(__shfl(f_after, SLAB_EPOCH - 1));
                    have_ring = true;
                } else {
                    have_ring = false;
                }
                if (lane == 0) geo[5] = status;
            }
            __syncthreads();
            const int status = uniform_i(geo[5]);
            epoch_staged = status != 2;
#if defined(VR_EXPERIMENTS) && defined(VR_X_STATS)
            st_epochs++; st_cold += status == 1; st_fallback += status == 2;
#endif
            if (status == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing of the old torus may still be landing
                adopt_geometry();
                __syncthreads();
            }
            if (epoch_staged) entry_next = plan[0];
        }
        // ---- loader: request the layers this phase's plan entry names
        int tail_pieces = 0;
        if (epoch_staged) {
            const uint4 e = entry_next;
            const int first_b = uniform_i((int)e.x);
            const uint32_t ey = __builtin_amdgcn_readfirstlane(e.y), ez = __builtin_amdgcn_readfirstlane(e.z), ew = __builtin_amdgcn_readfirstlane(e.w);
            const int n = (int)(ez & 15u);
            if (n > 0) {
                const int lo_a = (int)(ey & 0xffffu), lo_b = (int)(ey >> 16);
                const int keep = (int)((ez >> 4) & 15u), dda = (int)((ez >> 8) & 31u), ddb = (int)((ez >> 13) & 31u);
                const int la = (int)((ez >> 18) & 31u), lb = (int)((ez >> 23) & 31u);
                int lz = (int)(ew & 0xffu);
                const bool cold = (ew & 0x100u) != 0u;
                // this lane's brick inside the rectangle, per piece (the same for every layer of the entry)
                uint32_t rel[SLAB_MAX_PIECES];
                bool ok[SLAB_MAX_PIECES];
#pragma unroll
                for (int q = 0; q < SLAB_MAX_PIECES; q++) {
                    int oa = ld_ta[q] - la, ob = ld_tb[q] - lb;
                    if (oa < 0) oa += RA;
                    if (ob < 0) ob += RB;
                    ok[q] = ld_ok[q] && oa <= dda && ob <= ddb;
                    rel[q] = __umul24((uint32_t)(lo_a + oa), sA) + __umul24((uint32_t)(lo_b + ob), sB);
                }
                for (int l = 0; l < n; l++) {
                    const uint32_t layer_idx = (uint32_t)(first_b + sgn * l) * sM;   // uniform
#pragma unroll
                    for (int q = 0; q < SLAB_MAX_PIECES; q++) {
                        if (q >= pieces) break;
                        if (__any(ok[q] ? 1 : 0)) {
                            if (ok[q]) {
                                const uint8_t *g = src + (uint64_t)(rel[q] + layer_idx) * (uint64_t)C::SLOT + (uint64_t)(ld_part[q] * 16);
#if !(defined(VR_EXPERIMENTS) && defined(VR_X_NODMA))
                                glds16(g, ring_base + (uint32_t)lz * layer_bytes + (uint32_t)(q * SLAB_NW + (int)wave) * 1024u);
#else
                                asm volatile("" :: "v"(g));
#endif
                            }
                            if (l >= n - keep) tail_pieces++;
                        }
                    }
                    lz += sgn;
                    if (lz >= RZ) lz = 0;
                    if (lz < 0) lz = RZ - 1;
                }
                if (cold) {                                              // first phase on a fresh torus: everything is needed now
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    tail_pieces = 0;
                }
            }
        }
        // ---- this phase's four samples
        // (the two branches do not share their compositing code on purpose: the compiler waits for the
        // global loads of the unstaged branch where their results are used, and a wait placed at a common
        // join would also drain the staged branch's LDS-DMA pieces, which it cannot tell apart)
        if (!fin) {
            bool term;
            if (epoch_staged) {
                uint32_t v[BATCH];
                if (!prepared) prepare(off, nib);
                fetch(off, v);
                term = consume(v, nib, false);
            } else {
                uint32_t v[BATCH];
                gather_global(v);
                term = consume(v, 0u, true);
                asm volatile("" ::: "memory");
            }
            if (term) { done = true; fin = true; }
            else if (++b >= nb) fin = true;
        }
        // ---- the next phase's table look-ups and plan entry, ahead of its barrier (not across an epoch
        // boundary: the anchor wants the positions of the phase's first sample, and the plan may change)
        prepared = false;
        if (epoch_staged && pe != SLAB_EPOCH - 1) {
            entry_next = plan[pe + 1];
            if (!fin) { prepare(off, nib); prepared = true; }
        }
        // ---- what the next phase reads must have landed before its barrier
#if !(defined(VR_EXPERIMENTS) && defined(VR_X_NOWAIT))
        if (epoch_staged) slab_wait_pieces(tail_pieces);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup's LDS

    // ---- checked tail: the shader's loop, literally
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
#if defined(VR_EXPERIMENTS) && defined(VR_X_STATS)      // per-tile load-plan statistics instead of the fetch count of the tile's first pixel
    if (spp && threadIdx.x == 0) { spp[pix] = 0x80000000u | st_epochs | (st_cold << 8) | (st_fallback << 16) | ((unsigned)(RA * RB) << 24 & 0x7f000000u); return; }
#endif
    if (spp) spp[pix] = (uint32_t)i;
}

// ------------------------------------------------------------------ dispatch
template <typename VoxelT, bool PK12, int DIVTC, int VIEW, bool POW2, bool NOCLAMP, int MODE>
static hipError_t launch_slab(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                              uint32_t *spp, hipStream_t st)
{
    hipLaunchKernelGGL((raymarch_slab_kernel<VoxelT, PK12, DIVTC, VIEW, POW2, NOCLAMP, MODE>), dim3(L.tile_table_blocks),
                       dim3(SLAB_THREADS), 0, st, P, (const VoxelT *)vol, (const uint8_t *)(PK12 ? L.packed12 : vol), tf, fb, spp,
                       L.tile_table);
    return hipGetLastError();
}

template <typename VoxelT, bool PK12, int VIEW, int MODE>
static hipError_t dispatch_slab3(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                 uint32_t *spp, hipStream_t st)
{
    const bool noclamp = MODE == 0 && L.lut_noclamp != 0;               // like the fast kernel: headline mode only
    const bool pow2 = L.pow2_dims != 0 && L.divmode_tc == DIV_UNIT;
#define VR_SLAB(TC, P2) (noclamp ? launch_slab<VoxelT, PK12, TC, VIEW, P2, MODE == 0, MODE>(P, L, vol, tf, fb, spp, st) \
                                 : launch_slab<VoxelT, PK12, TC, VIEW, P2, false, MODE>(P, L, vol, tf, fb, spp, st))
    if (L.divmode_tc == DIV_CERT) return VR_SLAB(DIV_CERT, false);
    if (pow2) return VR_SLAB(DIV_UNIT, true);
    return VR_SLAB(DIV_UNIT, false);
#undef VR_SLAB
}

template <typename VoxelT, bool PK12>
static hipError_t dispatch_slab(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                uint32_t *spp, hipStream_t st)
{
    const int view = P.view_top == 1 ? 1 : (P.view_bottom == 1 ? 2 : 0);
    const int mode = (L.mip ? 1 : 0) + (P.tf_len > 1 ? 2 : 0);
#define VR_SLAB_M(VW)                                                                                          \
    switch (mode) {                                                                                            \
    case 0: return dispatch_slab3<VoxelT, PK12, VW, 0>(P, L, vol, tf, fb, spp, st);                            \
    case 1: return dispatch_slab3<VoxelT, PK12, VW, 1>(P, L, vol, tf, fb, spp, st);                            \
    case 2: return dispatch_slab3<VoxelT, PK12, VW, 2>(P, L, vol, tf, fb, spp, st);                            \
    default: return dispatch_slab3<VoxelT, PK12, VW, 3>(P, L, vol, tf, fb, spp, st);                           \
    }
    if (view == 0) { VR_SLAB_M(0) }
    if (view == 1) { VR_SLAB_M(1) }
    VR_SLAB_M(2)
#undef VR_SLAB_M
}

#ifndef VR_SLAB_TU
#define VR_SLAB_TU -1
#endif
#if VR_SLAB_TU == 0 || VR_SLAB_TU == -1
hipError_t launch_raymarch_slab_u8(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                   uint32_t *spp, hipStream_t st)
{
    return dispatch_slab<uint8_t, false>(P, L, vol, tf, fb, spp, st);
}
#endif
#if VR_SLAB_TU == 1 || VR_SLAB_TU == -1
hipError_t launch_raymarch_slab_pk12(const FrameParams &P, const LaunchConfig &L, const void *vol, const float4 *tf, float4 *fb,
                                     uint32_t *spp, hipStream_t st)
{
    return dispatch_slab<uint16_t, true>(P, L, vol, tf, fb, spp, st);
}
#endif

}  // namespace vr
