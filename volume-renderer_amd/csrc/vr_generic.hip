// vr_generic.hip -- the generic ray-march kernel: the shader line by line, every mode a run-time switch
// Part of vr_kernels.hip (textually included there, inside namespace vr, once per translation unit VR_TU): not a
// translation unit of its own.  Split out of vr_kernels.hip in round 6; the arithmetic contract is stated in that file's header.
#ifndef VR_TU_MAIN
#error "include through vr_kernels.hip"
#endif

// ------------------------------------------------------------------ generic kernel
// One kernel that follows the shader line by line and takes every mode as a run-time
// (wave-uniform) switch.  It is the correctness backbone: every configuration the
// specialised kernels do not cover runs here.
// BIG = false: 32-bit voxel offsets through a bounds-checked buffer resource (VoxelAddr /
// VoxelFetch); the eight taps of TRILINEAR share their per-axis address terms.
template <typename VoxelT, int LAYOUT, bool COUNT, bool BIG>
__global__ __launch_bounds__(256) void raymarch_generic_kernel(const FrameParams P, const int filter,
                                                               const int is_mip, const int divmode,
                                                               const uint32_t vol_bytes,
                                                               const VoxelT *__restrict__ vol,
                                                               const float4 *__restrict__ tf,
                                                               float4 *__restrict__ fb,
                                                               uint32_t *__restrict__ spp,
                                                               const unsigned tiles_x, const unsigned tiles_y)
{
    unsigned tx, ty;
    tile_of_block(blockIdx.x, tiles_x, tiles_y, tx, ty);
    if (tx == 0xffffffffu) return;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(tx * 16u + (wave & 1u) * 8u + (lane & 7u));
    const int ly = (int)(ty * 16u + (wave >> 1) * 8u + (lane >> 3));
    // local row -> global row (contiguous shard or cyclic stripes)
    int px = lx, py;
    if (P.stripe_count > 1) {
        const int s = ly / P.stripe_rows, r = ly % P.stripe_rows;
        py = (s * P.stripe_count + P.stripe_index) * P.stripe_rows + r;
    } else {
        py = P.row_begin + ly;
    }
    if (px >= P.col_lim || py >= P.row_lim || py >= P.row_end) return;

    const Ray ray = compute_ray(P, (float)px + 0.5f, (float)py + 0.5f);
    float t_min = 0.0f, t_max = 0.0f;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
    uint32_t fetches = 0;
    if (intersect_ray_aabb(P, ray, t_min, t_max)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, BIG ? 0 : (int)pair_load_extent(vol_bytes), 0x00020000);
        // per-axis terms of VoxelAddr<LAYOUT, false>::at(i, j, k) = X(i) + Y(j) + Z(k)  (mod 2^32)
        auto term_x = [&](int i) -> uint32_t {
            return LAYOUT == 0 ? (uint32_t)i : mad_u24((uint32_t)i >> BRICK_LX, 64u - (uint32_t)BRICK_X, (uint32_t)i);
        };
        auto term_y = [&](int j) -> uint32_t {
            if (LAYOUT == 0) return mad_u24((uint32_t)j, (uint32_t)P.nx, 0u);
            return mad_u24(BRICK_LY ? (uint32_t)j >> BRICK_LY : (uint32_t)j, P.bstride_y, BRICK_LY ? (uint32_t)j << BRICK_LX : 0u);
        };
        auto term_z = [&](int k) -> uint32_t {
            if (LAYOUT == 0) return mad_u24(mad_u24((uint32_t)k, (uint32_t)P.ny, 0u), (uint32_t)P.nx, 0u);
            return mad_u24(BRICK_LZ ? (uint32_t)k >> BRICK_LZ : (uint32_t)k, P.bstride_z, BRICK_LZ ? (uint32_t)k << (BRICK_LX + BRICK_LY) : 0u);
        };
        auto tap = [&](uint32_t off) -> float { return (float)VoxelFetch<VoxelT, false>::load(vol, rs, off); };
        const float EPSILON = 0.000001f;
        const float sx = ray.ox + ray.dx * t_min, sy = ray.oy + ray.dy * t_min, sz = ray.oz + ray.dz * t_min;
        const float p0x = sx + ray.dx * EPSILON, p0y = sy + ray.dy * EPSILON, p0z = sz + ray.dz * EPSILON;
        const float dsx = ray.dx * P.step, dsy = ray.dy * P.step, dsz = ray.dz * P.step;
        float qx = p0x, qy = p0y, qz = p0z;
        for (int i = 0; i < P.max_steps; i++) {
            if (P.accum == 1) {
                const float fi = (float)i;
                qx = p0x + fi * dsx; qy = p0y + fi * dsy; qz = p0z + fi * dsz;
            }
            // cartesianToTextureCoord (VolumeRenderer.cs:175-192)
            float ux = qx + P.half[0], uy = qy + P.half[1], uz = qz + P.half[2];
            if (divmode == DIV_CERT) {
                ux = div_cert(ux, P.ext[0], P.rext[0]);
                uy = div_cert(uy, P.ext[1], P.rext[1]);
                uz = div_cert(uz, P.ext[2], P.rext[2]);
            } else {
                ux = ux / P.ext[0]; uy = uy / P.ext[1]; uz = uz / P.ext[2];
            }
            const float uzr = uz;   // z before the flip of :185
            uz = 1.0f - uz;
            float tcx, tcy, tcz;
            if (P.view_top == 1) { tcx = ux; tcy = uzr; tcz = uy; }   // 1 - (1 - z) as the GL compiles it: z
            else if (P.view_bottom == 1) { tcx = ux; tcy = uz; tcz = 1.0f - uy; }
            else { tcx = ux; tcy = uy; tcz = uz; }
            if (tcx > 1.0f || tcy > 1.0f || tcz > 1.0f || tcx < 0.0f || tcy < 0.0f || tcz < 0.0f || d3 >= 0.95f)
                break;
            float s;
            if (filter == 0) {
                const int vi = clampi(floor_to_int_sat(tcx * P.fdim[0]), 0, P.nx - 1);
                const int vj = clampi(floor_to_int_sat(tcy * P.fdim[1]), 0, P.ny - 1);
                const int vk = clampi(floor_to_int_sat(tcz * P.fdim[2]), 0, P.nz - 1);
                s = BIG ? fetch_voxel<VoxelT, LAYOUT>(P, vol, vi, vj, vk) : tap(term_x(vi) + term_y(vj) + term_z(vk));
            } else {
                const float u = tcx * P.fdim[0] - 0.5f, v = tcy * P.fdim[1] - 0.5f, w = tcz * P.fdim[2] - 0.5f;
                const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
                const float ax = u - fu, ay = v - fv, az = w - fw;
                const int iu = (int)fu, iv = (int)fv, iw = (int)fw;
                const int i0 = clampi(iu, 0, P.nx - 1), i1 = clampi(iu + 1, 0, P.nx - 1);
                const int j0 = clampi(iv, 0, P.ny - 1), j1 = clampi(iv + 1, 0, P.ny - 1);
                const int k0 = clampi(iw, 0, P.nz - 1), k1 = clampi(iw + 1, 0, P.nz - 1);
                float c000, c100, c010, c110, c001, c101, c011, c111;
                if (BIG) {
                    c000 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j0, k0); c100 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j0, k0);
                    c010 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j1, k0); c110 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j1, k0);
                    c001 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j0, k1); c101 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j0, k1);
                    c011 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i0, j1, k1); c111 = fetch_voxel<VoxelT, LAYOUT>(P, vol, i1, j1, k1);
                } else {
                    const uint32_t x0 = term_x(i0), x1 = term_x(i1), y0 = term_y(j0), y1 = term_y(j1);
                    const uint32_t z0 = term_z(k0), z1 = term_z(k1);
                    // the two x-neighbours of a tap pair are adjacent in memory unless i0 is the last
                    // voxel of its brick row (or of the volume): every lane fetches its x0 tap together
                    // with the next storage element in one load (bounds-checked: the last element of the
                    // buffer reads 0 there), and only the lanes whose x1 lies elsewhere fetch it again
                    const bool pair = i1 == i0 + 1 && (LAYOUT == 0 || ((uint32_t)i0 & (BRICK_X - 1u)) != BRICK_X - 1u);
                    auto tap2 = [&](uint32_t off, float &lo, float &hi) {
                        if (sizeof(VoxelT) == 1) {
                            const uint32_t v = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
                            lo = (float)(v & 0xffu); hi = (float)(v >> 8);
                        } else {
                            const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off << 1), 0, 0);
                            lo = (float)(v & 0xffffu); hi = (float)(v >> 16);
                        }
                    };
                    tap2(x0 + y0 + z0, c000, c100); tap2(x0 + y1 + z0, c010, c110);
                    tap2(x0 + y0 + z1, c001, c101); tap2(x0 + y1 + z1, c011, c111);
                    if (!pair) {
                        c100 = tap(x1 + y0 + z0); c110 = tap(x1 + y1 + z0); c101 = tap(x1 + y0 + z1); c111 = tap(x1 + y1 + z1);
                    }
                }
                const float c00 = tri_lerp(c000, c100, ax), c10 = tri_lerp(c010, c110, ax);
                const float c01 = tri_lerp(c001, c101, ax), c11 = tri_lerp(c011, c111, ax);
                const float c0 = tri_lerp(c00, c10, ay), c1 = tri_lerp(c01, c11, ay);
                s = tri_lerp(c0, c1, az);
            }
            fetches++;
            // window (VolumeRenderer.cs:122-124; Q4: max==min defined as 0)
            s = gl_min(gl_max(s, P.fmin), P.fmax);
            if (P.fden == 0.0f) s = 0.0f;
            else if (s <= P.fmax && s >= P.fmin) s = (s - P.fmin) / P.fden;
            float s0 = s, s1 = s, s2 = s, s3 = s;
            if (P.tf_len > 1) {
                const float fi = s * (float)(P.tf_len - 1) + 0.5f;
                int idx = floor_to_int_sat(fi);
                idx = clampi(idx, 0, P.tf_len - 1);
                const float4 t = tf[idx];
                s0 = t.x; s1 = t.y; s2 = t.z; s3 = t.w;
            }
            if (is_mip == 1) {
                s0 *= P.alpha_scale; s1 *= P.alpha_scale; s2 *= P.alpha_scale; s3 *= P.alpha_scale;
                if (d3 < s3) { d0 = s0; d1 = s1; d2 = s2; d3 = s3; }
            } else {
                s3 *= P.alpha_scale;
                s0 *= s3; s1 *= s3; s2 *= s3;
                const float om = 1.0f - d3;
                d0 += s0 * om; d1 += s1 * om; d2 += s2 * om; d3 += s3 * om;
                if (d3 > 0.99f) break;
            }
            if (P.accum == 0) { qx += dsx; qy += dsy; qz += dsz; }
        }
    }
    const size_t pix = (size_t)(P.fb_compact ? ly : py) * (size_t)P.img_w + (size_t)px;
    store_pixel(P, fb, pix, d0, d1, d2, d3);
    if (COUNT) spp[pix] = fetches;
}
