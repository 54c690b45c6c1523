// vr_fast.hip -- the specialised NEAREST kernel (raymarch_fast_kernel): 32x16-pixel tiles, LDS tables, safe prefix, batched gathers
// Part of vr_kernels.hip (textually included there, inside namespace vr, once per translation unit VR_TU): not a
// translation unit of its own.  Split out of vr_kernels.hip in round 6; the arithmetic contract is stated in that file's header.
#ifndef VR_TU_MAIN
#error "include through vr_kernels.hip"
#endif

// Workgroup = 512 threads = 8 wavefronts = a 32x16-pixel tile (4x2 wave tiles of 8x8).
// Normally the host's longest-first tile table (tile_schedule.cpp) maps blocks to tiles.  The
// arithmetic order below is the table-less fallback (RendererCore::tile_order = 0): blocks go
// to XCDs in chunks of FAST_CHUNK horizontally adjacent tiles, and with an odd number of chunks
// per tile row the owner (chunk index mod 8) rotates from row to row, so every XCD gets an
// equal share of every image region.
constexpr unsigned FAST_THREADS = 512, FAST_TILE_W = kFastTileW, FAST_TILE_H = kFastTileH, FAST_CHUNK = kFastChunk;

struct FastGrid { unsigned tiles_x, tiles_y, chunks_per_row, blocks; };

[[maybe_unused]] static inline FastGrid fast_grid(int img_w, int rows)
{
    FastGrid g;
    g.tiles_x = (unsigned)((img_w + (int)FAST_TILE_W - 1) / (int)FAST_TILE_W);
    g.tiles_y = (unsigned)((rows + (int)FAST_TILE_H - 1) / (int)FAST_TILE_H);
    g.chunks_per_row = (g.tiles_x + FAST_CHUNK - 1) / FAST_CHUNK;
    g.chunks_per_row |= 1u;                                   // odd: owners rotate per row
    const unsigned chunks = g.chunks_per_row * g.tiles_y;
    g.blocks = ((chunks + 7u) / 8u) * 8u * FAST_CHUNK;
    return g;
}

__device__ __forceinline__ bool fast_tile_of_block(unsigned b, unsigned tiles_x, unsigned tiles_y,
                                                   unsigned chunks_per_row, unsigned &tx, unsigned &ty)
{
    const unsigned xcd = b & 7u, slot = b >> 3;
    const unsigned c = (slot / FAST_CHUNK) * 8u + xcd;       // chunk index, owner = c % 8
    ty = c / chunks_per_row;
    tx = (c % chunks_per_row) * FAST_CHUNK + (slot % FAST_CHUNK);
    return ty < tiles_y && tx < tiles_x;
}

// POW2: every volume dimension is a power of two and the box extents are 1 (DIV_UNIT), so
// texcoord*N is an exact scaling and the prefix can be marched in voxel units:
// fl(q*S + ds*S) == S*fl(q + ds) for a power-of-two S, bit for bit (also in the subnormal
// range, where fp32 addition is exact), which drops the three per-sample multiplies.
// NOCLAMP: the dataset's exact min/max lie inside the window, so clamp() is the identity.
// MODE: 0 = grey-ramp emission/absorption composite (rayMarchVolume, :104-139)
//       1 = MIP with the grey ramp (MIP(), :141-173): all four channels carry max(v*alpha)
//       2 = composite through the 1-D transfer function table (needs LUT; entries are RGBA)
// SKIPT: empty-space skipping compiled in (its probe state costs ~18 VGPRs = one workgroup
// of occupancy per CU, so the headline variant is also built without it).
// BATCH: samples per gather batch (8: the skip grid's dilation covers exactly that).
template <typename VoxelT, int LAYOUT, int DIVTC, int VIEW, bool BIG, bool LUT, bool POW2, bool NOCLAMP, int MODE, bool SKIPT, int BATCH, bool ATAB, bool PK12, bool PIPE = false>
__global__ __launch_bounds__(512, 1) void raymarch_fast_kernel(const FrameParams P,
                                                            const VoxelT *__restrict__ vol,
                                                            const float4 *__restrict__ tf,
                                                            const uint32_t vol_bytes,
                                                            float4 *__restrict__ fb,
                                                            uint32_t *__restrict__ spp,
                                                            const unsigned tiles_x, const unsigned tiles_y,
                                                            const unsigned chunks_per_row,
                                                            const uint32_t *__restrict__ tile_table,
                                                            const uint16_t *__restrict__ skip_grid,
                                                            const uint32_t skip_grid_bytes,
                                                            const void *__restrict__ packed12,
                                                            const uint32_t packed12_bytes)
{
    static_assert(BATCH == 8 || !SKIPT, "empty-space skipping assumes 8-sample batches");
    static_assert(!PIPE || (!SKIPT && !BIG), "the pipelined loop issues gathers for lanes without a next batch: bounds-checked buffer loads, no skip branch");
    static_assert(!PK12 || (ATAB && sizeof(VoxelT) == 2 && LAYOUT == 1 && BATCH <= 8), "12-bit copy: u16 bricks through the address tables");
    __shared__ float lut[LUT ? (ATAB && BIG ? FAST_BIG_LUT_FLOATS : FAST_LUT_MAX * 2) : 4];   // 32 KiB: 4096 x (c,a) or 256 x (r,g,b,a) + index bytes
    __shared__ uint32_t axis_tab[ATAB ? (BIG ? FAST_AXIS_TAB_BIG_MAX : FAST_AXIS_TAB_MAX) : 1];
    static_assert(!(ATAB && BIG) || LAYOUT == 1, "64-bit address tables exist for the bricked layout");
    unsigned tx, ty;
    if (tile_table) {                                       // host-built longest-first order
        const uint32_t t = tile_table[blockIdx.x];
        if (t == 0xffffffffu) return;                        // padding block
        tx = t & 0xffffu; ty = t >> 16;
    } else if (!fast_tile_of_block(blockIdx.x, tiles_x, tiles_y, chunks_per_row, tx, ty)) {
        return;                                              // padding block
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned mx = lane & 7u, my = lane >> 3;
    const int lx = (int)(tx * FAST_TILE_W + (wave & 3u) * 8u + mx);
    const int ly = (int)(ty * FAST_TILE_H + (wave >> 2) * 8u + my);
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
    if (LUT || ATAB) {
        // tabulate only if some ray of the workgroup enters the volume
        if (__syncthreads_or(hit ? 1 : 0)) {
            if constexpr (ATAB) build_axis_tables<VoxelT, LAYOUT, PK12, BIG>(P, axis_tab, (int)FAST_THREADS);
            const int n = LUT ? P.max_val - P.min_val + 1 : 0;
            for (int e = (int)threadIdx.x; e < n; e += (int)FAST_THREADS) {
                const float s = (float)(P.min_val + e);          // == clamp(float(texel), fmin, fmax)
                const float v = div_cert(s - P.fmin, P.fden, P.rden);
                if (MODE >= 2) {
                    // classification through the transfer function: index = round(v*(len-1)) here,
                    // src.a *= alpha_scale, src.rgb *= src.a in the 256-entry table below
                    int idx = floor_to_int_sat(v * (float)(P.tf_len - 1) + 0.5f);
                    idx = clampi(idx, 0, P.tf_len - 1);
                    reinterpret_cast<uint8_t *>(lut)[FAST_TF_ENTRIES * 16 + e] = (uint8_t)idx;
                } else if (MODE == 0 && P.tf_grey != 0) {
                    // a GREY transfer function (r == g == b in every entry: the reference's own colour ramp, whose colour
                    // widget is commented out) composited: the table's (c, a) pair is the whole classification, so the
                    // grey-ramp instance runs it -- one LDS read per sample and 3 workgroups per CU instead of MODE 2's two
                    // reads and 2 workgroups (cfg4: 3.26 -> ms of the grey mode).  Entries with MODE 2's own operations;
                    // r == g == b of the frame holds bit for bit because the three channels see identical operands.
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
            if (LUT && MODE >= 2) {
                for (int e = (int)threadIdx.x; e < P.tf_len; e += (int)FAST_THREADS) {
                    const float4 t = tf[e];
                    const float a = t.w * P.alpha_scale;
                    if (MODE == 3) {            // MIP(): s *= alpha_scale on all four channels (VolumeRenderer.cs:164)
                        lut[4 * e + 0] = t.x * P.alpha_scale; lut[4 * e + 1] = t.y * P.alpha_scale; lut[4 * e + 2] = t.z * P.alpha_scale;
                    } else {                    // composite: src.a *= alpha_scale, src.rgb *= src.a (:130-131)
                        lut[4 * e + 0] = t.x * a; lut[4 * e + 1] = t.y * a; lut[4 * e + 2] = t.z * a;
                    }
                    lut[4 * e + 3] = a;
                }
            }
            __syncthreads();
        }
    }

    float drgb = 0.0f, dg = 0.0f, db = 0.0f, da = 0.0f;   // MODE 0/1: r == g == b bit for bit, only drgb is carried
    uint32_t fetches = 0;
    {                                   // every thread runs the (barrier-carrying) batch loop
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, BIG ? 0 : (int)vol_bytes, 0x00020000);
        // PK12: the prefix gathers from the 12-bit packed copy of the bricked volume (voxel with
        // storage index s = bits [12s, 12s+12) of the stream: 25 % fewer cache lines per frame)
        const __amdgpu_buffer_rsrc_t rs12 = __builtin_amdgcn_make_buffer_rsrc((void *)packed12, 0, (int)packed12_bytes, 0x00020000);
        const float EPSILON = 0.000001f;
        const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
        float qx = sx + ray.dx * EPSILON, qy = sy + ray.dy * EPSILON, qz = sz + ray.dz * EPSILON;
        const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;

        // ---- checked head (vr_device.h: head_steps): the samples before the safe prefix can start.  Their positions are
        //      stepped here (the shader's additions); they are sampled through the checked loop further down
        const float hqx = qx, hqy = qy, hqz = qz;
        const int head = hit ? head_steps(P, qx, qy, qz, dsx, dsy, dsz) : 0;
        for (int h = 0; h < head; h++) { qx += dsx; qy += dsy; qz += dsz; }
        // ---- safe prefix length (samples head .. head + k_safe - 1)
        const int k_safe = hit ? safe_prefix_length(P, qx, qy, qz, dsx, dsy, dsz, P.max_steps - head) : 0;

        // texcoord -> voxel index of a position; valid (unclamped) inside the prefix
        auto voxel_of = [&](float ax, float ay, float az, int &vi, int &vj, int &vk) {
            const float ux = div_mode<DIVTC>(ax + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(ay + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(az + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }          // view_top
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }     // view_bottom
            vi = (int)(tcx * P.fdim[0]);
            vj = (int)(tcy * P.fdim[1]);
            vk = (int)(tcz * P.fdim[2]);
        };
        constexpr int LUT_SHIFT = 3;
        // byte offset of entry 0 relative to texel*entry_bytes (MODE 2: of index byte 0 relative to texel)
        // PK12: the packed copy stores voxel - pk12_base (the dataset minimum), so a volume whose RANGE fits 12 bits
        // packs, whatever its offset (CT data stored 1000 .. 5095).  classify() then works on the packed values: the
        // window limits and the table bias are shifted by the base instead of the texel -- no instruction is added to
        // the sample loop; the rare checked steps subtract the base from the u16 voxel they read.
        const int pkb = PK12 ? P.pk12_base : 0;
        const int wmin = P.min_val - pkb, wmax = P.max_val - pkb;
        const int lut_bias = MODE >= 2 ? FAST_TF_ENTRIES * 16 - wmin : -8 * wmin;
        uint32_t lut_entry0 = lds_offset_of(lut) + (uint32_t)lut_bias;
        asm volatile("" : "+v"(lut_entry0));
        // window + classification of one texel -> premultiplied colour c (cg, cb only in MODE 2)
        // and opacity a of VolumeRenderer.cs:130-131
        auto classify = [&](uint32_t texel, float &c, float &cg, float &cb, float &a) {
            if (LUT) {
                int t = (int)texel;
                if (!NOCLAMP) t = med3_i32(t, wmin, wmax);   // clamp(texel, min_val, max_val), min <= max
                if (MODE >= 2) {
                    const uint32_t idx = reinterpret_cast<const uint8_t *>(lut)[(uint32_t)(t + lut_bias)];
                    const float4 q = reinterpret_cast<const float4 *>(lut)[idx];
                    c = q.x; cg = q.y; cb = q.z; a = q.w;
                } else {
                    // one v_lshl_add_u32 with the table's LDS address + bias in a VECTOR register (the compiler's form was shift,
                    // mask, add with an SGPR operand: two more slow-rate instructions per sample)
                    uint32_t entry;
                    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(entry) : "v"(t), "v"(lut_entry0));
                    static_assert(LUT_SHIFT == 3, "the shift is spelled out in the instruction above");
                    VR_LDS_AS const float *ca = reinterpret_cast<VR_LDS_AS const float *>((size_t)entry);      // (one ds_read_b64)
                    c = ca[0]; a = ca[1];
                }
            } else {
                float s = (float)(texel + (uint32_t)pkb);
                s = fminf(fmaxf(s, P.fmin), P.fmax);        // operands are never NaN here
                s = div_cert(s - P.fmin, P.fden, P.rden);
                a = s * P.alpha_scale;
                c = s * a;
            }
        };
        // POW2: per box axis the scale is the dimension of the voxel axis it maps to
        const float Sx = P.fdim[0], Sy = VIEW == 0 ? P.fdim[1] : P.fdim[2], Sz = VIEW == 0 ? P.fdim[2] : P.fdim[1];
        float Qx = qx * Sx, Qy = qy * Sy, Qz = qz * Sz;
        const float dSx = dsx * Sx, dSy = dsy * Sy, dSz = dsz * Sz;
        const float Hx = P.half[0] * Sx, Hy = P.half[1] * Sy, Hz = P.half[2] * Sz;
        // (in VECTOR registers for the sample loop: a VALU instruction with an SGPR operand issues at the slow rate on gfx950, 4.2
        // cycles instead of 2.3 -- fp32 and integer alike, tools/ubench/valu_rates2.hip)
        float Szv = Sz, Syv = Sy;
        asm volatile("" : "+v"(Szv), "+v"(Syv));
        // ---- exact empty-space skipping (vr_set_skip_empty): a batch is skipped when the
        // dilated cell-max grid says every voxel within one 8^3 cell of the batch's middle
        // sample classifies to (0,0,0,0), i.e. compositing it cannot change a single bit of
        // dest.  The grid is probed one batch ahead at an APPROXIMATE position (closed form
        // V0 + k*dV in voxel units; the 3x3x3 dilation covers the +-4.5 voxels of the batch
        // and the approximation error), so the probe never delays the gathers.
        const bool skip_on = SKIPT && P.skip_empty != 0 && skip_grid != nullptr;
        const __amdgpu_buffer_rsrc_t rs_grid = __builtin_amdgcn_make_buffer_rsrc((void *)skip_grid, 0, (int)skip_grid_bytes, 0x00020000);
        float V0x = 0.0f, V0y = 0.0f, V0z = 0.0f, dVx = 0.0f, dVy = 0.0f, dVz = 0.0f;
        if (skip_on) {
            // affine map position -> voxel coordinate per voxel axis (flips included)
            auto to_voxel = [&](float ax, float ay, float az, float &fx, float &fy, float &fz) {
                const float ux = (ax + P.half[0]) * P.rext[0], uy = (ay + P.half[1]) * P.rext[1];
                const float uzr = (az + P.half[2]) * P.rext[2], uz = 1.0f - uzr;
                float tcx = ux, tcy = uy, tcz = uz;
                if (VIEW == 1) { tcy = uzr; tcz = uy; }
                else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
                fx = tcx * P.fdim[0]; fy = tcy * P.fdim[1]; fz = tcz * P.fdim[2];
            };
            float ex, ey, ez;
            to_voxel(qx, qy, qz, V0x, V0y, V0z);
            to_voxel(qx + dsx, qy + dsy, qz + dsz, ex, ey, ez);
            dVx = ex - V0x; dVy = ey - V0y; dVz = ez - V0z;
        }
        // probe for the batch whose first sample has index k0
        auto probe = [&](int k0) -> uint32_t {
            const float km = (float)k0 + 0.5f * (float)(BATCH - 1);
            int ci = (int)(V0x + km * dVx) >> 3, cj = (int)(V0y + km * dVy) >> 3, ck = (int)(V0z + km * dVz) >> 3;
            ci = min(max(ci, 0), P.cnx - 1); cj = min(max(cj, 0), P.cny - 1); ck = min(max(ck, 0), P.cnz - 1);
            const uint32_t cell = mad_u24(mad_u24((uint32_t)ck, (uint32_t)P.cny, (uint32_t)cj), (uint32_t)P.cnx, (uint32_t)ci);
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs_grid, (int)(cell << 1), 0, 0);
        };
        uint32_t cell_next = 0xffffffffu;       // probe result for the next batch to be issued
        int k_issue = 0;                        // index of the first sample of the next batch to be issued
        const uint32_t *tab_x = axis_tab, *tab_y = axis_tab + (ATAB ? P.nx : 0), *tab_z = axis_tab + (ATAB ? P.nx + P.ny : 0);
        // gathers of one batch: BATCH consecutive samples from the current position;
        // returns true when the batch is skipped (positions still advance, bit-exactly)
        // PIPE (true software pipelining): the gathers of the next batch are issued by every live lane, also by the
        // lanes that have no next batch (commit == false: their positions are restored, their loads hit the buffer
        // bounds check or fetch voxels nobody composites).  A branch around the loads makes the number of loads in
        // flight at the following s_waitcnt path-dependent, and the compiler then waits for the NEW batch before it
        // lets the previous one be composited.  With PIPE a wavefront keeps 16 gathers per lane in flight: that
        // shortens the serial chain of an under-filled launch (one GPU's shard of a multi-GPU frame) and only adds L1
        // pressure to a full one, so the host picks it per launch (vr_frame.h: LaunchConfig::pipelined).
        constexpr bool SPEC = PIPE;
        // The batch arrays hold the loaded voxels in their own width: the zero-extension is then an operation of
        // consume(), not of issue() -- as a 32-bit value it would be materialised (and the loads waited for) at
        // the end of the divergent region the gathers are issued in.
        using RawT = typename std::conditional<BIG, uint32_t, VoxelT>::type;
        auto issue = [&](RawT (&v)[BATCH], uint32_t &nib, bool commit) -> bool {   // nib (PK12): bit offset (0 / 4) of sample u in nibble u
            const float Qx0 = Qx, Qy0 = Qy, Qz0 = Qz, qx0 = qx, qy0 = qy, qz0 = qz;
            bool skip = false;
            if (skip_on) {
                skip = (int)cell_next <= P.skip_thresh;
                cell_next = probe(k_issue + BATCH);
                k_issue += BATCH;
            }
            if (skip) {
#pragma unroll
                for (int u = 0; u < BATCH; u++) {
                    if (POW2) { Qx += dSx; Qy += dSy; Qz += dSz; }
                    else { qx += dsx; qy += dsy; qz += dsz; }
                }
                return true;
            }
            typename VoxelAddr<LAYOUT, BIG>::type off[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                int vi, vj, vk;
                if (POW2) {
                    // voxel units: Q = q*S, U = Q + half*S = texcoord*S before the flips
                    const float ux = Qx + Hx, uy = Qy + Hy, uzr = Qz + Hz, uz = Szv - uzr;
                    float fx = ux, fy = uy, fz = uz;
                    if (VIEW == 1) { fy = uzr; fz = uy; }
                    else if (VIEW == 2) { fy = uz; fz = Syv - uy; }
                    vi = (int)fx; vj = (int)fy; vk = (int)fz;
                    Qx += dSx; Qy += dSy; Qz += dSz;
                } else {
                    voxel_of(qx, qy, qz, vi, vj, vk);
                    qx += dsx; qy += dsy; qz += dsz;
                }
                if (ATAB && BIG) off[u] = (typename VoxelAddr<LAYOUT, BIG>::type)((uint64_t)(tab_x[vi] + tab_y[vj]) + ((uint64_t)tab_z[vk] << 4));   // elements
                else if (ATAB) off[u] = (typename VoxelAddr<LAYOUT, BIG>::type)(tab_x[vi] + tab_y[vj] + tab_z[vk]);   // bytes
                else off[u] = VoxelAddr<LAYOUT, BIG>::at(P, vi, vj, vk);
                if (PK12) nib = u == 0 ? ((uint32_t)vi & 1u) << 2 : ((((uint32_t)vi & 1u) << (4 * u + 2)) | nib);   // odd x: upper 12 of the 16 bits
            }
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (ATAB && !BIG) {
                    v[u] = sizeof(VoxelT) == 1 ? (RawT)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off[u], 0, 0)
                                               : (RawT)__builtin_amdgcn_raw_buffer_load_b16(PK12 ? rs12 : rs, (int)off[u], 0, 0);
                    continue;
                }
                v[u] = (RawT)VoxelFetch<VoxelT, BIG>::load(vol, rs, off[u]);
            }
            if (SPEC && !commit) {
                if (POW2) { Qx = Qx0; Qy = Qy0; Qz = Qz0; }
                else { qx = qx0; qy = qy0; qz = qz0; }
            }
            return false;
        };
        int i = 0;
        // front-to-back compositing of one batch; returns true when the ray terminated.
        // alpha_scale in [0,1] (fast-path precondition) makes dest.a non-decreasing and <= 1,
        // so "dest.a < 0.95 before the LAST sample" proves the shader's per-sample test
        // `dest.a >= 0.95 -> break` (VolumeRenderer.cs:118) passed for the whole batch; only
        // the batch in which a ray terminates is replayed with the literal per-sample tests.
        // one sample onto the destination (front-to-back composite, or MIP's running maximum)
        auto accumulate = [&](float c, float cg, float cb, float a) {
            if (MODE == 1) {
                if (da < a) da = a;                          // dest = src when dest.a < src.a (:165-168)
            } else if (MODE == 3) {
                if (da < a) { drgb = c; dg = cg; db = cb; da = a; }
            } else {
                const float om = 1.0f - da;
                drgb += c * om;
                if (MODE == 2) { dg += cg * om; db += cb * om; }
                da += a * om;
            }
        };
        // PK12: the 12 bits of sample u start at bit 0 or 4 of the 16 loaded (nibble u of nib)
        auto texel_of = [&](RawT raw, uint32_t nib, int u) -> uint32_t {
            if (!PK12) return (uint32_t)raw;
            const uint32_t sh = __builtin_amdgcn_ubfe(nib, 4 * u, 4);
            __builtin_assume(sh <= 4u);
            return ((uint32_t)raw >> sh) & 0xfffu;
        };
        auto consume_live = [&](const RawT (&v)[BATCH], bool skipped, uint32_t nib) -> bool {
            if (skipped) { i += BATCH; return false; }   // every sample of the batch adds exactly zero
            float c[BATCH], cg[BATCH], cbl[BATCH], a[BATCH];
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            float da_last = 0.0f;
            // table look-ups in two halves: half the live registers, the second half's LDS
            // latency hides behind the first half's dependent compositing chain
            constexpr int HALF = BATCH / 2;
#pragma unroll
            for (int u = 0; u < HALF; u++) classify(texel_of(v[u], nib, u), c[u], cg[u], cbl[u], a[u]);
#pragma unroll
            for (int u = HALF; u < BATCH; u++) classify(texel_of(v[u], nib, u), c[u], cg[u], cbl[u], a[u]);
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (u == BATCH - 1) da_last = da;
                accumulate(c[u], cg[u], cbl[u], a[u]);
            }
            if (da_last < 0.95f) { i += BATCH; return false; }
            drgb = drgb0; dg = dg0; db = db0; da = da0;
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (da >= 0.95f) return true;
                accumulate(c[u], cg[u], cbl[u], a[u]);
                i++;
            }
            return da >= 0.95f;
        };
        // PIPE: every lane of the wavefront runs the batch (no branch between the gathers and their use, see
        // issue()); the lanes that are not `live` get their state back.
        auto consume = [&](const RawT (&v)[BATCH], bool skipped, uint32_t nib, bool live) -> bool {
            const float drgb0 = drgb, dg0 = dg, db0 = db, da0 = da;
            const int i0 = i;
            bool term = consume_live(v, skipped, nib);
            if (SPEC && !live) { drgb = drgb0; dg = dg0; db = db0; da = da0; i = i0; term = false; }
            return term;
        };

        bool done = false;
        // one iteration of the shader's loop at position (x, y, z), literally (dest.a > 0.99 of :134 is implied by the
        // dest.a >= 0.95 test of the next iteration and changes nothing); true = the loop ends here
        auto checked_step = [&](float &x, float &y, float &z, float stx, float sty, float stz) -> bool {
            const float ux = div_mode<DIVTC>(x + P.half[0], P.ext[0], P.rext[0]);
            const float uy = div_mode<DIVTC>(y + P.half[1], P.ext[1], P.rext[1]);
            const float uzr = div_mode<DIVTC>(z + P.half[2], P.ext[2], P.rext[2]);   // z before the flip of :185
            const float uz = 1.0f - uzr;
            float tcx = ux, tcy = uy, tcz = uz;
            if (VIEW == 1) { tcy = uzr; tcz = uy; }
            else if (VIEW == 2) { tcy = uz; tcz = 1.0f - uy; }
            if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || da >= 0.95f) return true;
            const int vi = min((int)(tcx * P.fdim[0]), P.nx - 1);
            const int vj = min((int)(tcy * P.fdim[1]), P.ny - 1);
            const int vk = min((int)(tcz * P.fdim[2]), P.nz - 1);
            float c, cg = 0.0f, cb = 0.0f, a;
            classify(VoxelFetch<VoxelT, BIG>::load(vol, rs, VoxelAddr<LAYOUT, BIG>::at(P, vi, vj, vk)) - (uint32_t)pkb, c, cg, cb, a);
            accumulate(c, cg, cb, a);
            x += stx; y += sty; z += stz;
            return false;
        };
        // Up to TAILK iterations of the shader's loop from (x, y, z), at most `limit` of them -- the same operations as TAILK calls of
        // checked_step(), in another ORDER: first the positions (the shader's own iterated additions), bound tests and voxel
        // addresses of all TAILK iterations, their fetches issued together, then classification and compositing in the shader's
        // order, stopping at the first iteration whose test fails.  A fetch behind that point was speculative (a valid, clamped
        // address inside the volume, or not issued at all when its own bound test failed) and is dropped.  Why: the literal loop
        // pays one dependent memory round trip per iteration -- the ~5 samples between the end of the safe prefix and the box's
        // face plus the prefix's remainder (< BATCH) are 8-12 serial round trips per ray, 4 % of a cfg3 tile's time and 15-20 % of
        // a 256^3 tile's (round 3: profiles/r03_fast_kernel_phases.txt); batched they are two or three.  true = the loop ended.
        // Measured (round 6, kernel ms, literal loop -> batches of 4): cfg1 shape 0.0517 -> 0.0475 (-8 %), cfg2 shape 0.1765 -> 0.1747,
        // cfg3 0.463 -> 0.462; batches of 8 lose (cfg1 0.0544).  Not in ONE family of instances: the plain (not pipelined) 8-sample loop
        // on the 12-bit packed copy -- a candidate of the measured choice on cfg3's shards -- ran 4-14 % SLOWER with it (N = 1 / 4
        // shards 0.474 -> 0.493 / 0.187 -> 0.213, at fewer registers: a code-layout effect) and keeps the literal loop.
#ifndef VR_TAIL_BATCH
#define VR_TAIL_BATCH 4
#endif
        constexpr int TAILK = (PIPE || BATCH == 4 || !PK12) ? VR_TAIL_BATCH : 1;
        auto checked_batch = [&](float &x, float &y, float &z, float stx, float sty, float stz, int limit) -> bool {
            bool out[TAILK];
            uint32_t tex[TAILK];
            float px = x, py = y, pz = z;
#pragma unroll
            for (int u = 0; u < TAILK; u++) {
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
                    const int vi = min((int)(tcx * P.fdim[0]), P.nx - 1);
                    const int vj = min((int)(tcy * P.fdim[1]), P.ny - 1);
                    const int vk = min((int)(tcz * P.fdim[2]), P.nz - 1);
                    tex[u] = VoxelFetch<VoxelT, BIG>::load(vol, rs, VoxelAddr<LAYOUT, BIG>::at(P, vi, vj, vk));
                }
                px += stx; py += sty; pz += stz;
            }
#pragma unroll
            for (int u = 0; u < TAILK; u++) {
                if (u >= limit) return false;                            // (the caller's budget, not the shader's: the loop goes on)
                if (i >= P.max_steps || out[u] || da >= 0.95f) return true;
                float c, cg = 0.0f, cb = 0.0f, a;
                classify(tex[u] - (uint32_t)pkb, c, cg, cb, a);
                accumulate(c, cg, cb, a);
                x += stx; y += sty; z += stz;
                i++;
            }
            return false;
        };
        // ---- checked head: samples 0 .. head - 1
        if (head > 0) {
            float x = hqx, y = hqy, z = hqz;
            if (TAILK > 1) {
                for (int h = 0; h < head && !done; h += TAILK) done = checked_batch(x, y, z, dsx, dsy, dsz, head - h);     // (true: the ray ends inside its head)
            } else {
                for (int h = 0; h < head && !done; h++) {
                    if (i >= P.max_steps || checked_step(x, y, z, dsx, dsy, dsz)) done = true;     // the ray ends inside its head
                    else i++;
                }
            }
        }
        // ---- safe prefix: software-pipelined, the next batch's gathers are in flight
        //      while the current batch is composited
        const int nb = done ? 0 : k_safe / BATCH;
        // The 8 wavefronts of a workgroup advance in lockstep (one barrier per two batches):
        // their rays cross the same voxel rows / bricks at the same time, so a cache line
        // fetched for one wavefront is still in the CU's L1 when its neighbours need it.
        {
            RawT va[BATCH], vb[BATCH];
            uint32_t nib_a = 0, nib_b = 0;
            bool skip_a = false, skip_b = false;
            int b = 0;
            bool fin = nb == 0;
            if (!fin) {
                if (skip_on) cell_next = probe(0);
                skip_a = issue(va, nib_a, true);
            }
            // lockstep: one plain barrier per 16 samples; every 4th doubles as the vote "all rays finished"
            // (__syncthreads_and is three barriers and a cross-lane reduction: 0.464 -> 0.458 ms on cfg3)
            // (the loop is rotated by hand -- barrier / vote at the bottom -- so that the loop-carried values leave an iteration in
            // the registers its body wrote them to: with the test at the top the register allocator copied six of them to other
            // registers at the loop header and back after the barrier, every iteration: 17 v_mov fewer per 16 samples, 0.4647 ->
            // 0.462 ms on cfg3, bit-identical)
            if (!__syncthreads_and(fin ? 1 : 0)) {
                unsigned it = 0;
                bool stop = false;
                do {
                    if (SPEC) {
                        // straight-line body: the compiler's s_waitcnt sees exactly eight gathers issued behind the
                        // eight it is about to use, on every path (ISA: vmcnt(15) ... vmcnt(8))
                        if (__builtin_amdgcn_ballot_w64(!fin) != 0ull) {                  // (0: nothing left in this wavefront)
                            bool live = !fin;
                            skip_b = issue(vb, nib_b, live && b + 1 < nb);
                            if (consume(va, skip_a, nib_a, live)) { done = true; fin = true; }
                            else if (live && ++b >= nb) fin = true;
                            live = !fin;
                            skip_a = issue(va, nib_a, live && b + 1 < nb);
                            if (consume(vb, skip_b, nib_b, live)) { done = true; fin = true; }
                            else if (live && ++b >= nb) fin = true;
                        }
                    } else {
                        if (!fin) {
                            if (b + 1 < nb) skip_b = issue(vb, nib_b, true);
                            if (consume(va, skip_a, nib_a, true)) { done = true; fin = true; }
                            else if (++b >= nb) fin = true;
                        }
                        if (!fin) {
                            if (b + 1 < nb) skip_a = issue(va, nib_a, true);
                            if (consume(vb, skip_b, nib_b, true)) { done = true; fin = true; }
                            else if (++b >= nb) fin = true;
                        }
                    }
                    it++;
                    if ((it & 3u) == 0u) stop = __syncthreads_and(fin ? 1 : 0) != 0;
                    else __syncthreads();
                } while (!stop);
            }
        }
        // back to box units for the tail (exact: S is a power of two); the step is re-derived
        // from its scaled copy so that only one of the two is live across the batch loop
        float tsx = dsx, tsy = dsy, tsz = dsz;
        if (POW2) {
            qx = Qx / Sx; qy = Qy / Sy; qz = Qz / Sz;
            tsx = dSx / Sx; tsy = dSy / Sy; tsz = dSz / Sz;
        }
        // ---- checked tail: the shader's loop
        if (hit && !done) {
            if (TAILK > 1) {
                while (!checked_batch(qx, qy, qz, tsx, tsy, tsz, TAILK)) {}
            } else {
                for (; i < P.max_steps; i++)
                    if (checked_step(qx, qy, qz, tsx, tsy, tsz)) break;
            }
        }
        fetches = (uint32_t)i;
    }
    if (!in_image) return;
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    if (MODE >= 2) store_pixel(P, fb, pix, drgb, dg, db, da);
    else if (MODE == 1) store_pixel(P, fb, pix, da, da, da, da);
    else store_pixel(P, fb, pix, drgb, drgb, drgb, da);
    if (spp) spp[pix] = fetches;
}
