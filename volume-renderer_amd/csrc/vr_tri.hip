// vr_tri.hip -- the batched TRILINEAR kernel on global-memory taps (raymarch_tri_kernel)
// Part of vr_kernels.hip (textually included there, inside namespace vr, once per translation unit VR_TU): not a
// translation unit of its own.  Split out of vr_kernels.hip in round 6; the arithmetic contract is stated in that file's header.
#ifndef VR_TU_MAIN
#error "include through vr_kernels.hip"
#endif

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
