// vr_relay.hip -- the relay kernel: four wavefronts march one 8x8-pixel tile (sparse launches, multi-GPU shards)
// Part of vr_kernels.hip (textually included there, inside namespace vr, once per translation unit VR_TU): not a
// translation unit of its own.  Split out of vr_kernels.hip in round 6; the arithmetic contract is stated in that file's header.
#ifndef VR_TU_MAIN
#error "include through vr_kernels.hip"
#endif

// ------------------------------------------------------------------ relay kernel
// A ray is a serial chain of ~1000 dependent samples; issued by ONE wavefront it advances at
// ~300 cycles per sample, so a launch that cannot fill the chip's 8192 wave slots (one GPU's
// shard of a multi-GPU frame) costs >= 0.16 ms however idle the chip is, and even the full
// frame is only ~1.3 rounds of such chains.  Here FOUR wavefronts march one 8x8-pixel tile
// as a relay: wavefront w owns batches w, w+4, w+8, ...; for its batch it generates the
// addresses, gathers and classifies on its own, and only the front-to-back compositing
// recurrence is handed from wavefront to wavefront through LDS (state = dest colour, dest
// alpha, sample count per ray; one sequence word published with release/acquire at
// workgroup scope).  Every sample still goes through exactly the shader's operations in
// the shader's order; only WHICH wavefront executes them changes.
// Round 6: the ray POSITIONS no longer travel.  Rounds 1-5 handed the position at the start of
// batch n + 1 from the wavefront that had generated batch n's addresses to the next one through
// LDS -- a second serial chain of ~130 hand-overs beside the compositing one.  The positions do
// not depend on the data, so every wavefront now iterates the shader's additions itself, in the
// shader's order (32 per axis between its own batches n and n + 4: the four wavefronts perform
// each addition four times instead of once, in issue slots a sparse launch leaves idle).  One
// rank's shard of the cfg3 frame at N = 8: 0.106 -> 0.084 ms, N = 4: 0.182 -> 0.153 (now ahead of
// the fast kernel's 0.163), bit-identical frames (profiles/r06_relay_own_positions.txt).
// Same preconditions as the fast kernel's headline shape (NEAREST, grey-ramp composite,
// iterative accumulation, default view, 32-bit offsets, alpha_scale in [0,1]).
// measured on cfg3: 2 or 8 wavefronts per tile, batches of 16, 4 tiles per workgroup are all slower
// (round 6, with the wavefronts' own positions: six or eight wavefronts per tile, one tile per workgroup: N = 8 shard 0.150 / 0.148 ms
// against 0.087 -- profiles/r06_relay_own_positions.txt)
#ifndef VR_RELAY_WAVES
#define VR_RELAY_WAVES 4
#define VR_RELAY_BATCH 8
#define VR_RELAY_TILES 2
#endif
constexpr int RELAY_WAVES = VR_RELAY_WAVES, RELAY_BATCH = VR_RELAY_BATCH;
// two tiles share one workgroup (and one 32 KiB classification table): 4 workgroups = 8 tiles
// = 32 wavefronts per CU, the wave-slot limit, instead of 4 tiles per CU
constexpr int RELAY_TILES = VR_RELAY_TILES, RELAY_THREADS = 64 * RELAY_WAVES * RELAY_TILES;

struct RelayState {
    float rgb[2][64];      // red (== green == blue in the grey modes)
    float g[2][64], b[2][64];   // green / blue: only the transfer-function modes carry them
    float a[2][64];
    int i[2][64];
    unsigned seq;          // number of batches composited so far; bit 31: no ray of the tile needs another batch (one word: one LDS load per poll)
    unsigned final_n;      // the state slot holding the result is final_n & 1
};
constexpr unsigned RELAY_STOP = 0x80000000u;

// VIEW / MODE as in the fast kernel (round 2: every mode and view of a sparse shard gets the relay)
template <typename VoxelT, int LAYOUT, int DIVTC, bool LUT, bool POW2, bool NOCLAMP, bool ATAB, bool PK12, int VIEW, int MODE>
__global__ __launch_bounds__(RELAY_THREADS) void raymarch_relay_kernel(const FrameParams P,
                                                             const VoxelT *__restrict__ vol,
                                                             const float4 *__restrict__ tf,
                                                             const uint32_t vol_bytes,
                                                             float4 *__restrict__ fb,
                                                             uint32_t *__restrict__ spp,
                                                             const uint32_t *__restrict__ tile_table,
                                                             const void *__restrict__ packed12,
                                                             const uint32_t packed12_bytes)
{
    static_assert(!PK12 || (ATAB && sizeof(VoxelT) == 2 && LAYOUT == 1), "12-bit copy: u16 bricks through the address tables");
    __shared__ float lut[LUT ? FAST_LUT_MAX * 2 : 4];
    __shared__ uint32_t axis_tab[ATAB ? FAST_AXIS_TAB_MAX : 1];
    __shared__ RelayState rs_all[RELAY_TILES];
    // block b -> (32x16 tile of the longest-first table, pair of 8x8 sub-tiles); the 4 pairs of
    // a tile are consecutive blocks of ONE XCD (b & 7 is the XCD)
    const unsigned b = blockIdx.x;
    constexpr unsigned GROUPS = 8u / RELAY_TILES;                        // workgroups per 32x16-pixel tile
    const uint32_t tile = tile_table[(b / (8u * GROUPS)) * 8u + (b & 7u)];
    if (tile == 0xffffffffu) return;
    const unsigned tl = threadIdx.x / (64u * RELAY_WAVES);              // tile of this wavefront within the workgroup
    RelayState &rs = rs_all[tl];
    const unsigned sub = ((b >> 3) % GROUPS) * RELAY_TILES + tl;
    const unsigned lane = threadIdx.x & 63u, w = (threadIdx.x >> 6) % RELAY_WAVES;
    const int lx = (int)((tile & 0xffffu) * FAST_TILE_W + (sub & 3u) * 8u + (lane & 7u));
    const int ly = (int)((tile >> 16) * FAST_TILE_H + (sub >> 2) * 8u + (lane >> 3));
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
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
    const int any_hit = __syncthreads_or(hit ? 1 : 0);
    static_assert(MODE < 2 || LUT, "the transfer-function modes classify through the table");
    if (LUT && any_hit) {
        const int n = P.max_val - P.min_val + 1;
        for (int e = (int)threadIdx.x; e < n; e += RELAY_THREADS) {
            const float s = (float)(P.min_val + e);
            const float v = div_cert(s - P.fmin, P.fden, P.rden);
            if (MODE >= 2) {                                     // two-level table, as in the fast kernel
                int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                idx = clampi(idx, 0, P.tf_len - 1);
                reinterpret_cast<uint8_t *>(lut)[FAST_TF_ENTRIES * 16 + e] = (uint8_t)idx;
            } else if (MODE == 0 && P.tf_grey != 0) {            // grey transfer function on the grey-ramp instance (see the fast kernel)
                int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                idx = clampi(idx, 0, P.tf_len - 1);
                const float4 t = tf[idx];
                const float a = t.w * P.alpha_scale;
                lut[2 * e + 0] = t.x * a; lut[2 * e + 1] = a;
            } else {
                const float a = v * P.alpha_scale;
                lut[2 * e + 0] = v * a; lut[2 * e + 1] = a;
            }
        }
        if (MODE >= 2) {
            for (int e = (int)threadIdx.x; e < P.tf_len; e += RELAY_THREADS) {
                const float4 t = tf[e];
                const float a = t.w * P.alpha_scale;
                if (MODE == 3) { lut[4 * e + 0] = t.x * P.alpha_scale; lut[4 * e + 1] = t.y * P.alpha_scale; lut[4 * e + 2] = t.z * P.alpha_scale; }
                else { lut[4 * e + 0] = t.x * a; lut[4 * e + 1] = t.y * a; lut[4 * e + 2] = t.z * a; }
                lut[4 * e + 3] = a;
            }
        }
    }
    if (ATAB && any_hit) build_axis_tables<VoxelT, LAYOUT, PK12>(P, axis_tab, RELAY_THREADS);
    if (w == 0) { rs.rgb[0][lane] = 0.0f; rs.g[0][lane] = 0.0f; rs.b[0][lane] = 0.0f; rs.a[0][lane] = 0.0f; rs.i[0][lane] = 0; }
    if (w == 0 && lane == 0) { rs.seq = 0u; rs.final_n = 0u; }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, (int)vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs12 = __builtin_amdgcn_make_buffer_rsrc((void *)packed12, 0, (int)packed12_bytes, 0x00020000);
    const uint32_t *tab_x = axis_tab, *tab_y = axis_tab + (ATAB ? P.nx : 0), *tab_z = axis_tab + (ATAB ? P.nx + P.ny : 0);
    const float EPSILON = 0.000001f;
    const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
    float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
    const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
    // checked head (vr_device.h: head_steps): every wavefront steps the positions identically; wavefront 0 samples them
    // through the checked loop before its first batch (it is the one that reads state slot 0)
    const float hqx = qx, hqy = qy, hqz = qz;
    const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
    for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
    // A ray can END inside its head: a head sample fails the shader's bounds test (o + d*t_min rounds to a point
    // outside the box when the eye is far away) or the step budget runs out.  That part of the head does not depend
    // on the voxel data, so every wavefront of the relay evaluates it for itself: such a ray has no prefix batches
    // and no tail (wavefront 0's literal head loop below stops at the same sample).
    bool head_ended = false;
    {
        float x = hqx, y = hqy, z = hqz;
        for (int h = 0; h < head && !head_ended; h++) {
            const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            if (h >= P.max_steps || tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f) head_ended = true;
            x += dsx; y += dsy; z += dsz;
        }
    }
    const int k_safe = (hit && !head_ended) ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;
    const int nb = k_safe / RELAY_BATCH;                 // batches of THIS ray
    int nbmax = nb;                                      // batches of the tile (same in all 4 wavefronts)
    for (int o = 32; o > 0; o >>= 1) nbmax = max(nbmax, __shfl_xor(nbmax, o));

    // marching units: voxels for POW2 (see the fast kernel), box units otherwise.  Only the
    // step is kept in registers; the position itself lives in LDS between wavefronts.
    // (per box axis the scale is the dimension of the voxel axis it maps to: y and z swap in the rotated views)
    const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
    const float mx = POW2 ? dsx * Sx : dsx, my = POW2 ? dsy * Sy : dsy, mz = POW2 ? dsz * Sz : dsz;
    const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;
    // (in VECTOR registers for the sample loop: a VALU instruction with an SGPR operand issues at the slow rate on gfx950, 4.2
    // cycles instead of 2.3 -- fp32 and integer alike, tools/ubench/valu_rates2.hip)
    float Szv = Sz, Syv = Sy;
    asm volatile("" : "+v"(Szv), "+v"(Syv));
    const int pkb = PK12 ? P.pk12_base : 0;      // see the fast kernel: classify() works on packed values (voxel - base)
    const int wmin = P.min_val - pkb, wmax = P.max_val - pkb;
    const int lut_bias = MODE >= 2 ? FAST_TF_ENTRIES * 16 - wmin : -8 * wmin;

    // waits until `n` batches have been composited; gives up when the tile has been stopped (early ray termination
    // of every ray, or the last batch): returns false, nothing may be touched any more
    auto wait_seq = [&](int n) -> bool {
        for (;;) {
            const unsigned v = __hip_atomic_load(&rs.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v & RELAY_STOP) return false;
            if ((v & ~RELAY_STOP) >= (unsigned)n) return true;
            // (no s_sleep between the polls: a relay launch leaves the SIMDs' issue slots idle anyway, and the 64-cycle naps sat on
            // the compositing chain -- N = 8 shard 0.087 -> 0.082 ms, N = 4 0.155 -> 0.152, N = 2 0.286 -> 0.278)
        }
    };
    bool stopped = false;
    // this wavefront's own copy of the ray position, at the start of batch own_n: the shader's additions (one fma with the
    // multiplicand 1.0 IS the addition; 0.0 keeps the position of a ray whose prefix has ended), every one of them, in order
    float ox = POW2 ? qx * Sx : qx, oy = POW2 ? qy * Sy : qy, oz = POW2 ? qz * Sz : qz;
    int own_n = 0;
    auto own_advance_to = [&](int n) {
        for (; own_n < n; own_n++) {
            const float vf = own_n < nb ? 1.0f : 0.0f;
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) { ox = __builtin_fmaf(mx, vf, ox); oy = __builtin_fmaf(my, vf, oy); oz = __builtin_fmaf(mz, vf, oz); }
        }
    };
    // gathers of batch n (if this ray still needs them); returns whether v[] is valid
    auto issue = [&](int n, uint32_t (&v)[RELAY_BATCH], uint32_t &nib, float da_seen) -> bool {
        if (stopped || (__hip_atomic_load(&rs.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & RELAY_STOP)) { stopped = true; return false; }
        own_advance_to(n);
        float x = ox, y = oy, z = oz;
        const bool need = n < nb && da_seen < 0.95f;
        uint32_t off[RELAY_BATCH];
        if (n < nb) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                int vi, vj, vk;
                if (POW2) {
                    const float ux = x + Hx, uy = y + Hy, uzr = z + Hz, uz = Szv - uzr;
                    float fx = ux, fy = uy, fz = uz;
                    if (VIEW == 1) { fy = uzr; fz = uy; }
                    else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
                    vi = (int)fx; vj = (int)fy; vk = (int)fz;
                } else {
                    const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
                    const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
                    const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
                    const float uz = 1.0f - uzr;
                    float tcx = ux, tcy = uy, tcz = uz;
                    if (VIEW == 1) { tcy = uzr; tcz = uy; }
                    else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
                    vi = (int)(tcx * P.fdim[0]); vj = (int)(tcy * P.fdim[1]); vk = (int)(tcz * P.fdim[2]);
                }
                if (ATAB) off[u] = tab_x[vi] + tab_y[vj] + tab_z[vk];                    // bytes
                else off[u] = VoxelAddr<LAYOUT, false>::at(P, vi, vj, vk);
                if (PK12) nib = u == 0 ? ((uint32_t)vi & 1u) << 2 : ((((uint32_t)vi & 1u) << (4 * u + 2)) | nib);
                x += mx; y += my; z += mz;
            }
        }
        ox = x; oy = y; oz = z; own_n = n + 1;           // the position at the start of batch n + 1
        if (need) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                if (ATAB)
                    v[u] = sizeof(VoxelT) == 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)off[u], 0, 0)
                                               : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(PK12 ? rs12 : rsrc, (int)off[u], 0, 0);
                else
                    v[u] = VoxelFetch<VoxelT, false>::load(vol, rsrc, off[u]);
            }
        }
        return need;
    };
    auto classify = [&](uint32_t texel, float &c, float &cg, float &cb, float &a) {
        if (LUT) {
            int t = (int)texel;
            if (!NOCLAMP) t = med3_i32(t, wmin, wmax);
            if (MODE >= 2) {
                const uint32_t idx = reinterpret_cast<const uint8_t *>(lut)[(uint32_t)(t + lut_bias)];
                const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
                c = q.x; cg = q.y; cb = q.z; a = q.w;
                return;
            }
            const float2 ca = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(lut) + (uint32_t)((t << 3) + lut_bias));
            c = ca.x; a = ca.y;
        } else {
            float s = (float)(texel + (uint32_t)pkb);
            s = fminf(fmaxf(s, P.fmin), P.fmax);
            s = div_cert(s - P.fmin, P.fden, P.rden);
            a = s * P.alpha_scale;
            c = s * a;
        }
    };
    float da_seen = 0.0f;
    // one sample onto the destination: front-to-back composite, or MIP's running maximum (see the fast kernel)
    auto accumulate = [&](float &drgb, float &dg, float &db, float &da, float c, float cg, float cb, float a) {
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
    // take over the recurrence for batch n, composite, hand it on
    auto relay = [&](int n, const uint32_t (&v)[RELAY_BATCH], uint32_t nib, bool valid) {
        float c[RELAY_BATCH], cg[RELAY_BATCH], cb[RELAY_BATCH], a[RELAY_BATCH];
        if (valid) {
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                cg[u] = cb[u] = 0.0f;
                classify(PK12 ? __builtin_amdgcn_ubfe(v[u], __builtin_amdgcn_ubfe(nib, 4 * u, 4), 12) : v[u], c[u], cg[u], cb[u], a[u]);
            }
        }
        if (stopped || !wait_seq(n)) { stopped = true; return; }
        const int slot = n & 1;
        float drgb = rs.rgb[slot][lane], da = rs.a[slot][lane], dg = 0.0f, db = 0.0f;
        if (MODE >= 2) { dg = rs.g[slot][lane]; db = rs.b[slot][lane]; }
        int i = rs.i[slot][lane];
        if (valid) {
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            float da_last = 0.0f;
#pragma unroll
            for (int u = 0; u < RELAY_BATCH; u++) {
                if (u == RELAY_BATCH - 1) da_last = da;
                accumulate(drgb, dg, db, da, c[u], cg[u], cb[u], a[u]);
            }
            if (da_last < 0.95f) {
                i += RELAY_BATCH;
            } else {                                     // the batch in which the ray terminates: literal per-sample tests
                drgb = drgb0; dg = dg0; db = db0; da = da0;
#pragma unroll
                for (int u = 0; u < RELAY_BATCH; u++) {
                    if (da >= 0.95f) break;
                    accumulate(drgb, dg, db, da, c[u], cg[u], cb[u], a[u]);
                    i++;
                }
            }
        }
        da_seen = da;
        rs.rgb[slot ^ 1][lane] = drgb; rs.a[slot ^ 1][lane] = da; rs.i[slot ^ 1][lane] = i;
        if (MODE >= 2) { rs.g[slot ^ 1][lane] = dg; rs.b[slot ^ 1][lane] = db; }
        // does any ray of the tile need another batch?  (terminated rays and rays whose prefix
        // ends here do not)
        const bool more = hit && da < 0.95f && n + 1 < nb;
        const bool last = !__any(more ? 1 : 0);
        if (last) {
            if (lane == 0) rs.final_n = (unsigned)(n + 1);
            stopped = true;
        }
        if (lane == 0) __hip_atomic_store(&rs.seq, (unsigned)(n + 1) | (last ? RELAY_STOP : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // Up to RELAY_TAILK iterations of the shader's loop (at most `limit`), as in the fast kernel's checked_batch(): positions, bound tests
    // and addresses of all of them first, their fetches in flight together, classification and compositing in the shader's order up
    // to the first failed test.  Wavefront 0 runs the head and the tail of a tile's rays while the other three wait: every memory
    // round trip saved there is saved for the whole tile.  true = the loop ended.
    constexpr int RELAY_TAILK = 4;
    const int nxm1 = P.nx - 1, nym1 = P.ny - 1, nzm1 = P.nz - 1;
    auto checked_batch = [&](float &x, float &y, float &z, float stx, float sty, float stz, int limit,
                             float &drgb, float &dg, float &db, float &da, int &i) -> bool {
        bool out[RELAY_TAILK];
        uint32_t tex[RELAY_TAILK];
        float px = x, py = y, pz = z;
#pragma unroll
        for (int u = 0; u < RELAY_TAILK; u++) {
            const float ux = div_mode<DIVTC>(px + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(py + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(pz + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            out[u] = tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || u >= limit;
            tex[u] = 0u;
            if (!out[u]) {
                const int vi = min((int)(tcx * P.fdim[0]), nxm1), vj = min((int)(tcy * P.fdim[1]), nym1), vk = min((int)(tcz * P.fdim[2]), nzm1);
                tex[u] = VoxelFetch<VoxelT, false>::load(vol, rsrc, VoxelAddr<LAYOUT, false>::at(P, vi, vj, vk));
            }
            px += stx; py += sty; pz += stz;
        }
#pragma unroll
        for (int u = 0; u < RELAY_TAILK; u++) {
            if (u >= limit) return false;
            if (i >= P.max_steps || out[u] || da >= 0.95f) return true;
            float c, cg = 0.0f, cb = 0.0f, a;
            classify(tex[u] - (uint32_t)pkb, c, cg, cb, a);
            accumulate(drgb, dg, db, da, c, cg, cb, a);
            x += stx; y += sty; z += stz;
            i++;
        }
        return false;
    };
    {
        if (w == 0 && head > 0) {                            // samples 0 .. head - 1, literally, into state slot 0
            float drgb = 0.0f, dg = 0.0f, db = 0.0f, da = 0.0f, x = hqx, y = hqy, z = hqz;
            int i = 0;
            for (int h = 0; h < head; h += RELAY_TAILK)
                if (checked_batch(x, y, z, dsx, dsy, dsz, head - h, drgb, dg, db, da, i)) break;
            rs.rgb[0][lane] = drgb; rs.a[0][lane] = da; rs.i[0][lane] = i;
            if (MODE >= 2) { rs.g[0][lane] = dg; rs.b[0][lane] = db; }
            da_seen = da;
        }
        uint32_t va[RELAY_BATCH], vb[RELAY_BATCH];
        uint32_t nib_a = 0, nib_b = 0;
        bool ok_a = false, ok_b = false;
        int n = (int)w;
        if (n < nbmax) ok_a = issue(n, va, nib_a, da_seen);
        while (n < nbmax && !stopped) {
            if (n + RELAY_WAVES < nbmax) ok_b = issue(n + RELAY_WAVES, vb, nib_b, da_seen);
            relay(n, va, nib_a, ok_a);
            n += RELAY_WAVES;
            if (n >= nbmax || stopped) break;
            if (n + RELAY_WAVES < nbmax) ok_a = issue(n + RELAY_WAVES, va, nib_a, da_seen);
            relay(n, vb, nib_b, ok_b);
            n += RELAY_WAVES;
        }
    }
    // every wavefront of the workgroup leaves its relay loop exactly once; afterwards the
    // state / position words are quiescent and wavefront 0 of each tile finishes the rays
    __syncthreads();
    if (w != 0) return;
    const unsigned fin = rs.final_n;                     // == nbmax unless the tile stopped early
    float drgb = rs.rgb[fin & 1][lane], da = rs.a[fin & 1][lane], dg = 0.0f, db = 0.0f;
    if (MODE >= 2) { dg = rs.g[fin & 1][lane]; db = rs.b[fin & 1][lane]; }
    int i = rs.i[fin & 1][lane];
    // a ray that still needs its tail finished its prefix at batch nb <= fin, and its position has not changed since
    own_advance_to((int)fin);
    qx = ox; qy = oy; qz = oz;
    if (POW2) { qx = qx / Sx; qy = qy / Sy; qz = qz / Sz; }   // exact: S is a power of two
    const float tsx = POW2 ? mx / Sx : mx, tsy = POW2 ? my / Sy : my, tsz = POW2 ? mz / Sz : mz;
    if (hit && !head_ended) {
        while (!checked_batch(qx, qy, qz, tsx, tsy, tsz, RELAY_TAILK, drgb, dg, db, da, i)) {}
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MODE >= 2) store_pixel(P, fb, pix, drgb, dg, db, da);
    else if (MODE == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    if (spp) spp[pix] = hit ? (uint32_t)i : 0u;
}
