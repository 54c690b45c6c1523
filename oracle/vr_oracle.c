/*
 * vr_oracle.c -- TEST INFRASTRUCTURE ONLY (see vr_oracle.h).  Parity pinned to the reference
 * shader executed under Mesa llvmpipe (tests/golden/ref_gl_*.npz), bit for bit.
 *
 * Scalar fp32 restatement of /root/reference/VolumeRenderer.cs, one function per
 * shader function, keeping the shader's operation order.  Build with
 * -ffp-contract=off and without -ffast-math: every + - * / sqrt below is one
 * correctly rounded IEEE-754 binary32 operation, which is the arithmetic contract
 * the HIP kernel is held to (tests compare bit patterns, the 1e-4 tolerance of
 * BASELINE.json is the fallback bar).
 *
 * GLSL operators are one correctly rounded operation each (measured on the pinned GL:
 * a/b is the correctly rounded quotient, a*b+c is never contracted).  Built-ins:
 *   min(x,y) = y<x ? y : x        max(x,y) = x<y ? y : x
 *   clamp(x,a,b) = min(max(x,a),b)
 *   M*v = ((M[0]*v.x + M[1]*v.y) + M[2]*v.z) + M[3]*v.w
 *   dot / length / normalize: two models, see VRO_ARITH_* in vr_oracle.h --
 *     MESA (default)  dot summed last component to first; normalize(v) = v * (1/sqrt(dot))
 *     SPEC            dot summed first to last;            normalize(v) = v / sqrt(dot)
 * texture() on the integer volume is NEAREST with CLAMP_TO_EDGE:
 *   i = clamp(int(floor(u*N)), 0, N-1)          (RendererCore.cpp:408-419)
 * TRILINEAR (north-star mode, no reference semantics) is GL's linear rule:
 *   u' = u*N - 0.5, i0 = floor(u'), f = u' - i0, both taps clamped to edge,
 *   lerp(a,b,t) = fma(t, b-a, a) (one rounding: the executed GL's linear filter), x first, then y, then z.
 */
#include "vr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z, w; } v4;

static inline float gl_min(float x, float y) { return (y < x) ? y : x; }
static inline float gl_max(float x, float y) { return (x < y) ? y : x; }
static inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }

/* normalize(): VRO_ARITH_SPEC = the specification text, v / sqrt(left-to-right dot);
   VRO_ARITH_MESA (default) = what Mesa's GLSL compiler emits and llvmpipe executes (measured by
   oracle/ref_gl/probe_arith.py): v * (1 / sqrt(dot)), dot summed from the LAST component
   to the first (NIR's fdot lowering).  Each op is still one correctly rounded binary32 op. */
static inline v4 v4_normalize(v4 v, int arith)
{
    if (arith == VRO_ARITH_MESA) {
        float rs = 1.0f / sqrtf(((v.w * v.w + v.z * v.z) + v.y * v.y) + v.x * v.x);
        v4 m = { v.x * rs, v.y * rs, v.z * rs, v.w * rs };
        return m;
    }
    float len = sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w);
    v4 r = { v.x / len, v.y / len, v.z / len, v.w / len };
    return r;
}

/* length(vec3(a,b,c)) in the two arithmetic models (see v4_normalize) */
static inline float v3_length(float a, float b, float c, int arith)
{
    return arith == VRO_ARITH_MESA ? sqrtf((c * c + b * b) + a * a) : sqrtf((a * a + b * b) + c * c);
}

/* per-frame constants: main() lines 62-83 of VolumeRenderer.cs */
typedef struct {
    float pmin[3], pmax[3], half[3], ext[3]; /* ext = bb.p_max + half_len (:180) */
    float step;                              /* :109 / :146 */
    float fdim[3];                           /* float(textureSize) */
    float fmin, fmax, fden;                  /* vec4(min_val), vec4(max_val), float(max-min) */
} frame_consts;

static void frame_setup(const vro_params *p, frame_consts *fc)
{
    /* VolumeRenderer.cs:65-66 */
    int max_dim = p->nx > p->ny ? p->nx : p->ny;
    max_dim = max_dim > p->nz ? max_dim : p->nz;
    int swz = (p->view_bottom == 1 || p->view_top == 1);
    /* :68-71  vec4(vol_size.xzy|xyz, 1) */
    float d0 = (float)p->nx, d1 = swz ? (float)p->nz : (float)p->ny, d2 = swz ? (float)p->ny : (float)p->nz;
    float s0 = p->voxel_size[0], s1 = swz ? p->voxel_size[2] : p->voxel_size[1],
          s2 = swz ? p->voxel_size[1] : p->voxel_size[2];
    float fmax_dim = (float)max_dim;
    /* :73 bb.p_max /= max_dim ; :75-78 bb.p_max *= voxel_size */
    float pm[3] = { (d0 / fmax_dim) * s0, (d1 / fmax_dim) * s1, (d2 / fmax_dim) * s2 };
    for (int i = 0; i < 3; i++) {
        fc->half[i] = pm[i] / 2.0f;            /* :81 */
        fc->pmin[i] = 0.0f - fc->half[i];      /* :82, p_min starts (0,0,0,1) :24 */
        fc->pmax[i] = pm[i] - fc->half[i];     /* :83 */
        fc->ext[i] = fc->pmax[i] + fc->half[i];/* :180 */
    }
    fc->fdim[0] = (float)p->nx; fc->fdim[1] = (float)p->ny; fc->fdim[2] = (float)p->nz;
    /* :109 length(p_max-p_min) / length(vec3(vol_size.xzy))   (composite)
       :146 length(p_max-p_min) / length(vec3(vol_size.xyz))   (MIP)        */
    float e0 = fc->pmax[0] - fc->pmin[0], e1 = fc->pmax[1] - fc->pmin[1], e2 = fc->pmax[2] - fc->pmin[2];
    float num = v3_length(e0, e1, e2, p->arith);
    float fx = (float)p->nx, fy = (float)p->ny, fz = (float)p->nz;
    float den = p->is_mip == 1 ? v3_length(fx, fy, fz, p->arith) : v3_length(fx, fz, fy, p->arith);
    fc->step = num / den;
    fc->fmin = (float)p->min_val;
    fc->fmax = (float)p->max_val;
    fc->fden = (float)(p->max_val - p->min_val);
}

/* VolumeRenderer.cs:194-216 */
static void compute_ray(const vro_params *p, float pixel_x, float pixel_y, v4 *origin, v4 *dir)
{
    const float *c = p->cam;
    float fw = (float)p->img_w, fh = (float)p->img_h;
    float aspect = (fw * 1.0f) / fh;
    float x = aspect * (((2.0f * pixel_x) / fw) - 1.0f);
    float y = ((2.0f * pixel_y) / fh) - 1.0f;
    float z = -c[20];
    v4 d = { x, y, z, 0.0f };
    d = v4_normalize(d, p->arith);
    v4 m;
    m.x = ((c[0] * d.x + c[4] * d.y) + c[8] * d.z) + c[12] * d.w;
    m.y = ((c[1] * d.x + c[5] * d.y) + c[9] * d.z) + c[13] * d.w;
    m.z = ((c[2] * d.x + c[6] * d.y) + c[10] * d.z) + c[14] * d.w;
    m.w = ((c[3] * d.x + c[7] * d.y) + c[11] * d.z) + c[15] * d.w;
    *dir = v4_normalize(m, p->arith);
    origin->x = c[16]; origin->y = c[17]; origin->z = c[18]; origin->w = c[19];
}

/* VolumeRenderer.cs:218-238 */
static int intersect_ray_aabb(const frame_consts *fc, const v4 *o, const v4 *d, float *t_min, float *t_max)
{
    float tmax = INFINITY, tmin = -INFINITY;
    float ix = 1.0f / d->x, iy = 1.0f / d->y, iz = 1.0f / d->z;
    float mnx = (fc->pmin[0] - o->x) * ix, mny = (fc->pmin[1] - o->y) * iy, mnz = (fc->pmin[2] - o->z) * iz;
    float mxx = (fc->pmax[0] - o->x) * ix, mxy = (fc->pmax[1] - o->y) * iy, mxz = (fc->pmax[2] - o->z) * iz;
    tmin = gl_max(tmin, gl_min(mnx, mxx));
    tmax = gl_min(tmax, gl_max(mnx, mxx));
    tmin = gl_max(tmin, gl_min(mny, mxy));
    tmax = gl_min(tmax, gl_max(mny, mxy));
    *t_min = tmin; *t_max = tmax;
    if (tmax < tmin)
        return 0;
    tmin = gl_max(tmin, gl_min(mnz, mxz));
    tmax = gl_min(tmax, gl_max(mnz, mxz));
    *t_min = tmin; *t_max = tmax;
    return tmax > gl_max(tmin, 0.0f);
}

/* VolumeRenderer.cs:175-192 */
static inline void cartesian_to_texcoord(const vro_params *p, const frame_consts *fc,
                                         float px, float py, float pz, float tc[3])
{
    px = px + fc->half[0]; py = py + fc->half[1]; pz = pz + fc->half[2];
    px = px / fc->ext[0];  py = py / fc->ext[1];  pz = pz / fc->ext[2];
    /* :185-187.  view_top reads 1 - (1 - z): Mesa's algebraic pass folds that to z
       (nir_opt_algebraic, a + -(a + b) -> -b; seen as 4 pixels of a 256x256 top view). */
    float pz_in = pz;
    pz = 1.0f - pz;
    if (p->view_top == 1) { tc[0] = px; tc[1] = (p->arith == VRO_ARITH_MESA) ? pz_in : 1.0f - pz; tc[2] = py; }
    else if (p->view_bottom == 1) { tc[0] = px; tc[1] = pz; tc[2] = 1.0f - py; }
    else { tc[0] = px; tc[1] = py; tc[2] = pz; }
}

static inline float fetch_voxel(const vro_params *p, int64_t i, int64_t j, int64_t k)
{
    int64_t idx = i + (int64_t)p->nx * (j + (int64_t)p->ny * k);
    if (p->bytes_per_voxel == 1) return (float)((const uint8_t *)p->volume)[idx];
    return (float)((const uint16_t *)p->volume)[idx];
}

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* texture(vol_tex3D, tc).r  (VolumeRenderer.cs:121,159) */
static inline float sample_volume(const vro_params *p, const frame_consts *fc, const float tc[3])
{
    if (p->filter == VRO_FILTER_NEAREST) {
        int64_t i = clampi((int64_t)floorf(tc[0] * fc->fdim[0]), 0, p->nx - 1);
        int64_t j = clampi((int64_t)floorf(tc[1] * fc->fdim[1]), 0, p->ny - 1);
        int64_t k = clampi((int64_t)floorf(tc[2] * fc->fdim[2]), 0, p->nz - 1);
        return fetch_voxel(p, i, j, k);
    }
    float u = tc[0] * fc->fdim[0] - 0.5f, v = tc[1] * fc->fdim[1] - 0.5f, w = tc[2] * fc->fdim[2] - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    float ax = u - fu, ay = v - fv, az = w - fw;
    int64_t i0 = clampi((int64_t)fu, 0, p->nx - 1), i1 = clampi((int64_t)fu + 1, 0, p->nx - 1);
    int64_t j0 = clampi((int64_t)fv, 0, p->ny - 1), j1 = clampi((int64_t)fv + 1, 0, p->ny - 1);
    int64_t k0 = clampi((int64_t)fw, 0, p->nz - 1), k1 = clampi((int64_t)fw + 1, 0, p->nz - 1);
    float c000 = fetch_voxel(p, i0, j0, k0), c100 = fetch_voxel(p, i1, j0, k0);
    float c010 = fetch_voxel(p, i0, j1, k0), c110 = fetch_voxel(p, i1, j1, k0);
    float c001 = fetch_voxel(p, i0, j0, k1), c101 = fetch_voxel(p, i1, j0, k1);
    float c011 = fetch_voxel(p, i0, j1, k1), c111 = fetch_voxel(p, i1, j1, k1);
    if (p->lerp_unfused) {  /* rounds 1-4: three separately rounded operations per lerp (kept as a measurement) */
        float c00 = c000 + ax * (c100 - c000), c10 = c010 + ax * (c110 - c010);
        float c01 = c001 + ax * (c101 - c001), c11 = c011 + ax * (c111 - c011);
        float c0 = c00 + ay * (c10 - c00), c1 = c01 + ay * (c11 - c01);
        return c0 + az * (c1 - c0);
    }
    /* lerp(a, b, t) = fma(t, b - a, a): what a real GL's linear filter computes (Mesa llvmpipe, GL_LINEAR on an R32F
       texture through the reference shader with usampler3D -> sampler3D: bit-identical, tests/golden/ref_gl/tri_*) */
    float c00 = fmaf(ax, c100 - c000, c000), c10 = fmaf(ax, c110 - c010, c010);
    float c01 = fmaf(ax, c101 - c001, c001), c11 = fmaf(ax, c111 - c011, c011);
    float c0 = fmaf(ay, c10 - c00, c00), c1 = fmaf(ay, c11 - c01, c01);
    return fmaf(az, c1 - c0, c0);
}

/* window mapping, VolumeRenderer.cs:122-124 (Q4: max==min defined as 0) */
static inline float window_map(const frame_consts *fc, float s)
{
    s = gl_clamp(s, fc->fmin, fc->fmax);
    if (fc->fden == 0.0f) return 0.0f;
    if (s <= fc->fmax && s >= fc->fmin)
        s = (s - fc->fmin) / fc->fden;
    return s;
}

/* src colour before alpha_scale: grey ramp (reference) or 1-D TF (N3) */
static inline void classify(const vro_params *p, float v, float src[4])
{
    if (p->tf_rgba && p->tf_len > 1) {
        float fi = v * (float)(p->tf_len - 1) + 0.5f;
        int idx = (int)floorf(fi);
        idx = idx < 0 ? 0 : (idx > p->tf_len - 1 ? p->tf_len - 1 : idx);
        src[0] = p->tf_rgba[4 * idx + 0]; src[1] = p->tf_rgba[4 * idx + 1];
        src[2] = p->tf_rgba[4 * idx + 2]; src[3] = p->tf_rgba[4 * idx + 3];
    } else {
        src[0] = src[1] = src[2] = src[3] = v;
    }
}

/* rayMarchVolume (VolumeRenderer.cs:104-139) and MIP (:141-173) */
static uint32_t march(const vro_params *p, const frame_consts *fc, const v4 *o, const v4 *d,
                      float t_min, float out[4])
{
    /* :107 start_point = origin + dir*t_min ; :114 pos = start_point + dir*EPSILON */
    const float EPSILON = 0.000001f;
    float sx = o->x + d->x * t_min, sy = o->y + d->y * t_min, sz = o->z + d->z * t_min;
    float p0x = sx + d->x * EPSILON, p0y = sy + d->y * EPSILON, p0z = sz + d->z * EPSILON;
    float dsx = d->x * fc->step, dsy = d->y * fc->step, dsz = d->z * fc->step;
    float px = p0x, py = p0y, pz = p0z;
    float dest[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t fetches = 0;
    for (int i = 0; i < p->max_steps; i++) {
        if (p->accum == VRO_ACCUM_CLOSED_FORM) {
            float fi = (float)i;
            px = p0x + fi * dsx; py = p0y + fi * dsy; pz = p0z + fi * dsz;
        }
        float tc[3];
        cartesian_to_texcoord(p, fc, px, py, pz, tc);
        if (tc[0] > 1.0f || tc[1] > 1.0f || tc[2] > 1.0f ||
            tc[0] < 0.0f || tc[1] < 0.0f || tc[2] < 0.0f || dest[3] >= 0.95f)
            break;
        float s = sample_volume(p, fc, tc);
        fetches++;
        float v = window_map(fc, s);
        float src[4];
        classify(p, v, src);
        if (p->is_mip == 1) {
            /* :164-168 */
            src[0] *= p->alpha_scale; src[1] *= p->alpha_scale;
            src[2] *= p->alpha_scale; src[3] *= p->alpha_scale;
            if (dest[3] < src[3]) { dest[0] = src[0]; dest[1] = src[1]; dest[2] = src[2]; dest[3] = src[3]; }
        } else {
            /* :130-135 */
            src[3] *= p->alpha_scale;
            src[0] *= src[3]; src[1] *= src[3]; src[2] *= src[3];
            float om = 1.0f - dest[3];
            dest[0] += src[0] * om; dest[1] += src[1] * om;
            dest[2] += src[2] * om; dest[3] += src[3] * om;
            if (dest[3] > 0.99f)
                break;
        }
        if (p->accum == VRO_ACCUM_ITERATIVE) { px += dsx; py += dsy; pz += dsz; } /* :136 */
    }
    out[0] = dest[0]; out[1] = dest[1]; out[2] = dest[2]; out[3] = dest[3];
    return fetches;
}

static uint64_t render_row(const vro_params *p, const frame_consts *fc, int py, float *rgba, uint32_t *spp)
{
    uint64_t total = 0;
    int wlim = p->trunc_grid ? (p->img_w / 16) * 16 : p->img_w;
    for (int px = 0; px < wlim; px++) {
        v4 o, d;
        compute_ray(p, (float)px + 0.5f, (float)py + 0.5f, &o, &d);   /* :86 */
        float t_min, t_max, col[4] = { 0, 0, 0, 0 };
        uint32_t n = 0;
        if (intersect_ray_aabb(fc, &o, &d, &t_min, &t_max))
            n = march(p, fc, &o, &d, t_min, col);
        float *dst = rgba + ((size_t)py * p->img_w + px) * 4;
        dst[0] = col[0]; dst[1] = col[1]; dst[2] = col[2]; dst[3] = col[3];
        if (spp) spp[(size_t)py * p->img_w + px] = n;
        total += n;
    }
    return total;
}

int vro_render(const vro_params *p, float *rgba, uint32_t *spp, uint64_t *total_samples)
{
    if (!p || !rgba || !p->volume || p->img_w <= 0 || p->img_h <= 0 || p->nx <= 0 || p->ny <= 0 ||
        p->nz <= 0 || (p->bytes_per_voxel != 1 && p->bytes_per_voxel != 2))
        return 1;
    frame_consts fc;
    frame_setup(p, &fc);
    int r0 = p->row_begin < 0 ? 0 : p->row_begin;
    int r1 = p->row_end > p->img_h ? p->img_h : p->row_end;
    if (p->trunc_grid) {
        int hlim = (p->img_h / 16) * 16;
        if (r1 > hlim) r1 = hlim;
    }
    uint64_t total = 0;
#ifdef _OPENMP
    if (p->threads > 1) {
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total) num_threads(p->threads)
        for (int py = r0; py < r1; py++)
            total += render_row(p, &fc, py, rgba, spp);
    } else
#endif
    {
        for (int py = r0; py < r1; py++)
            total += render_row(p, &fc, py, rgba, spp);
    }
    if (total_samples) *total_samples = total;
    return 0;
}

/* ------------------------------------------------------------------ camera */
/* src/Camera.cpp.  glm built-ins restated: normalize(v) = v * (1/sqrt(dot(v,v))),
   cross(a,b) = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y). */

static const float VRO_PI = 3.14159265358979323846264338327950288f;

static void cam_normalize3(float v[4])
{
    float inv = 1.0f / sqrtf(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]);
    v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
}

static void cam_set_view(vro_camera *c, const float eye[4], const float side[4], const float up[4],
                         const float look[4])
{
    /* Camera.cpp:46-57 */
    memcpy(c->eye, eye, sizeof(float) * 4);
    memcpy(c->side, side, sizeof(float) * 4); cam_normalize3(c->side);
    memcpy(c->up, up, sizeof(float) * 4); cam_normalize3(c->up);
    memcpy(c->look_at, look, sizeof(float) * 4); cam_normalize3(c->look_at);
    /* view2world_mat = mat4(side, up, -look_at, eye) -- note: the UN-normalised
       arguments (Camera.cpp:56 uses the parameters, which shadow the members) */
    for (int i = 0; i < 4; i++) {
        c->view2world[0 + i] = side[i];
        c->view2world[4 + i] = up[i];
        c->view2world[8 + i] = -look[i];
        c->view2world[12 + i] = eye[i];
    }
}

void vro_camera_reset(vro_camera *c)
{
    /* Camera.cpp:30-44 */
    const float eye[4] = { 0, 0, 3, 1 }, side[4] = { 1, 0, 0, 0 }, up[4] = { 0, 1, 0, 0 }, look[4] = { 0, 0, -1, 0 };
    cam_set_view(c, eye, side, up, look);
    c->zenith = VRO_PI / 2.0f;
    c->azimuth = 0;
    c->radius = 3;
}

void vro_camera_init(vro_camera *c, float y_fov, float rot_speed, float mov_speed)
{
    /* Camera.cpp:16-23 */
    memset(c, 0, sizeof(*c));
    c->y_fov = y_fov; c->rotation_speed = rot_speed; c->mov_speed = mov_speed;
    c->view_plane_dist = 1 / tanf(y_fov * VRO_PI / 360);
    c->is_changed = 1;
    vro_camera_reset(c);
}

void vro_camera_orient(vro_camera *c, float zoom, float zenith, float azimuth)
{
    /* Camera.cpp:83-151 */
    if (zenith == 0 && azimuth == 0) {
        for (int i = 0; i < 4; i++)
            c->eye[i] = zoom > 0 ? c->eye[i] + c->look_at[i] : c->eye[i] - c->look_at[i];
        c->radius = sqrtf((c->eye[0] * c->eye[0] + c->eye[1] * c->eye[1]) + c->eye[2] * c->eye[2]);
        for (int i = 0; i < 4; i++) c->view2world[12 + i] = c->eye[i];
        c->is_changed = 1;
        return;
    }
    float pi2 = VRO_PI * 2;
    float nz = c->zenith + zenith * c->rotation_speed;
    nz = nz < 0.0f ? 0.0f : (nz > VRO_PI ? VRO_PI : nz);    /* glm::clamp = min(max(x,lo),hi) */
    float na = c->azimuth + azimuth * c->rotation_speed;
    if (na < 0) na = pi2 - na;                               /* Q13: sic */
    else if (na > pi2) na = na - pi2;
    if (nz == c->zenith && na == c->azimuth) return;
    c->zenith = nz; c->azimuth = na;
    c->eye[0] = c->radius * sinf(c->zenith) * sinf(c->azimuth);
    c->eye[1] = c->radius * cosf(c->zenith);
    c->eye[2] = c->radius * sinf(c->zenith) * cosf(c->azimuth);
    c->eye[3] = 1;
    for (int i = 0; i < 4; i++) c->look_at[i] = -c->eye[i];
    c->look_at[3] = 0;
    cam_normalize3(c->look_at);
    float *l = c->look_at;
    if (c->zenith == 0 || c->zenith == VRO_PI) {
        /* side = rotate(I, azimuth, (0,1,0)) * (1,0,0,0) = (cos a, 0, -sin a, 0) */
        float ca = cosf(c->azimuth), sa = sinf(c->azimuth);
        /* glm::rotate: Rotate[0] = (c + (1-c)*ax*ax, (1-c)*ax*ay + s*az, (1-c)*ax*az - s*ay) with axis (0,1,0) */
        c->side[0] = ca + (1.0f - ca) * 0.0f * 0.0f;
        c->side[1] = (1.0f - ca) * 0.0f * 1.0f + sa * 0.0f;
        c->side[2] = (1.0f - ca) * 0.0f * 0.0f - sa * 1.0f;
        c->side[3] = 0;
    } else {
        /* cross(look, (0,1,0)) */
        c->side[0] = l[1] * 0.0f - 1.0f * l[2];
        c->side[1] = l[2] * 0.0f - 0.0f * l[0];
        c->side[2] = l[0] * 1.0f - 0.0f * l[1];
        c->side[3] = 0;
    }
    float *s = c->side;
    c->up[0] = s[1] * l[2] - l[1] * s[2];
    c->up[1] = s[2] * l[0] - l[2] * s[0];
    c->up[2] = s[0] * l[1] - l[0] * s[1];
    c->up[3] = 0;
    cam_normalize3(c->side);
    cam_normalize3(c->up);
    for (int i = 0; i < 4; i++) {
        c->view2world[0 + i] = c->side[i];
        c->view2world[4 + i] = c->up[i];
        c->view2world[8 + i] = -c->look_at[i];
        c->view2world[12 + i] = c->eye[i];
    }
    c->is_changed = 1;
}

void vro_camera_block(const vro_camera *c, float out[21])
{
    /* Camera.cpp:59-80: 4 columns, then eye.xyz,1, then view_plane_dist */
    memcpy(out, c->view2world, sizeof(float) * 16);
    out[16] = c->eye[0]; out[17] = c->eye[1]; out[18] = c->eye[2]; out[19] = 1;
    out[20] = c->view_plane_dist;
}

/* ------------------------------------------------------- synthetic volumes */
static uint32_t isqrt_u64(uint64_t v)
{
    uint64_t r = (uint64_t)sqrt((double)v);
    while (r * r > v) r--;
    while ((r + 1) * (r + 1) <= v) r++;
    return (uint32_t)r;
}

void vro_gen_sphere_u8(uint8_t *out, int32_t n, int32_t radius)
{
    int64_t lim = (int64_t)(2 * radius) * (2 * radius);
    for (int32_t k = 0; k < n; k++)
        for (int32_t j = 0; j < n; j++)
            for (int32_t i = 0; i < n; i++) {
                int64_t a = 2 * i + 1 - n, b = 2 * j + 1 - n, c = 2 * k + 1 - n;
                int64_t r2 = a * a + b * b + c * c;
                uint8_t v = 0;
                if (r2 < lim) v = (uint8_t)(255 - (255 * (int64_t)isqrt_u64((uint64_t)r2)) / (2 * radius));
                out[(size_t)i + (size_t)n * ((size_t)j + (size_t)n * (size_t)k)] = v;
            }
}

static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

void vro_gen_noise_ball(void *out, int32_t nx, int32_t ny, int32_t nz, int32_t bpv, uint32_t seed)
{
    int64_t N = nx > ny ? nx : ny; N = N > nz ? N : nz;
    int64_t vmax = bpv == 2 ? 4095 : 255;
    int shift = bpv == 2 ? 2 : 6;      /* noise amplitude 0..63 (u16) / 0..3 (u8) */
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                int64_t a = 2 * i + 1 - nx, b = 2 * j + 1 - ny, c = 2 * k + 1 - nz;
                int64_t r2 = a * a + b * b + c * c, n2 = N * N;
                int64_t base = r2 < n2 ? (vmax * (n2 - r2)) / n2 : 0;
                uint64_t lin = (uint64_t)i + (uint64_t)nx * ((uint64_t)j + (uint64_t)ny * (uint64_t)k);
                uint32_t h = fmix32(((uint32_t)lin ^ (uint32_t)(lin >> 32) * 0x9E3779B1u) ^ seed) & 0xFFu;
                int64_t v = base + (int64_t)(h >> shift);
                if (v > vmax) v = vmax;
                if (bpv == 2) ((uint16_t *)out)[lin] = (uint16_t)v;
                else ((uint8_t *)out)[lin] = (uint8_t)v;
            }
}

/* ----------------------------------------------- spline transfer function */
/* src/CubicSpline.cpp:50-115 (coefficients) and :20-40 (evaluation at an
   integer iso value), per RGBA channel.  knots5 = n x (iso, r, g, b, a). */
int vro_spline_tf(const float *knots5, int32_t n, float iso_max, float *lut, int32_t len)
{
    (void)iso_max;
    if (n < 2 || len < 2) return 1;
    int segs = n - 1;
    float *co = (float *)malloc(sizeof(float) * (size_t)n);
    float *delta = (float *)malloc(sizeof(float) * (size_t)n);
    float *deriv = (float *)malloc(sizeof(float) * (size_t)n);
    float *A = (float *)malloc(sizeof(float) * (size_t)segs * 4);
    if (!co || !delta || !deriv || !A) { free(co); free(delta); free(deriv); free(A); return 2; }
    for (int ch = 0; ch < 4; ch++) {
#define Y(i) knots5[(i) * 5 + 1 + ch]
        co[0] = 0.5f;
        for (int i = 1; i < segs; i++) co[i] = 1.0f / ((4.0f * 1.0f) - co[i - 1]);
        co[segs] = 1.0f / ((2.0f * 1.0f) - co[segs - 1]);
        delta[0] = 3.0f * (Y(1) - Y(0)) * co[0];
        for (int i = 1; i < segs; i++) delta[i] = (3.0f * (Y(i + 1) - Y(i - 1)) - delta[i - 1]) * co[i];
        delta[segs] = (3.0f * (Y(segs) - Y(segs - 1)) - delta[segs - 1]) * co[segs];
        deriv[segs] = delta[segs];
        for (int i = segs - 1; i >= 0; i--) deriv[i] = delta[i] - co[i] * deriv[i + 1];
        for (int i = 0; i < segs; i++) {
            A[i * 4 + 0] = Y(i);
            A[i * 4 + 1] = deriv[i];
            A[i * 4 + 2] = 3.0f * (Y(i + 1) - Y(i)) - 2.0f * deriv[i] - deriv[i + 1];
            A[i * 4 + 3] = 2.0f * (Y(i) - Y(i + 1)) + deriv[i] + deriv[i + 1];
        }
        for (int e = 0; e < len; e++) {
            int iso = e;   /* LUT entry e <-> integer iso value e (len = 256 in the widget) */
            float val = 0.0f; int done = 0, seg = 0; float t = 0.0f;
            for (int i = 0; i < n; i++) {
                int ki = (int)knots5[i * 5];
                if (ki == iso) { val = Y(i); done = 1; break; }
                else if (ki > iso) {
                    /* the reference indexes segment -1 when iso precedes the first
                       knot (UB); defined here as extrapolating segment 0 */
                    seg = i > 0 ? i - 1 : 0;
                    int k0 = (int)knots5[seg * 5], k1 = (int)knots5[(seg + 1) * 5];
                    t = (float)(iso - k0) / (float)(k1 - k0);
                    break;
                }
            }
            if (!done) {
                const float *a = &A[seg * 4];
                val = a[0] + t * (a[1] + t * (a[2] + t * a[3]));
            }
            val = val < 0.0f ? 0.0f : (val > 1.0f ? 1.0f : val);
            lut[e * 4 + ch] = val;
        }
#undef Y
    }
    free(co); free(delta); free(deriv); free(A);
    return 0;
}
