/*
 * vr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference ray-march path
 * (/root/reference/VolumeRenderer.cs:55-238 driven by
 * /root/reference/src/RendererCore.cpp:138-163 with the camera block of
 * /root/reference/src/Camera.cpp:59-80).  It is the parity checker for the HIP
 * path and the timed CPU baseline of bench.py; nothing in the product
 * (volume-renderer_amd/, include/) may include, link or call it.
 *
 * PARITY PINNED (round 5): the reference's unmodified shader is executed in the
 * authoring container by Mesa 23.2.1 llvmpipe through oracle/ref_gl/ref_gl_runner.c
 * (a DRI-swrast loader; real GLSL compiler, real texture unit emulation), with the
 * reference's own GL call sequence; the frames are committed as
 * tests/golden/ref_gl_*.npz and this oracle reproduces every one of them bit for bit
 * (tests/test_ref_gl_goldens.py).  One stated deviation when minting: MIN/MAG filter
 * GL_NEAREST instead of the reference's GL_LINEAR, because with GL_LINEAR the integer
 * texture is incomplete and the reference renders all zeros on a conformant GL
 * (measured) -- Mesa's own app-compat switch force_integer_tex_nearest
 * gives the same frames with the reference's calls untouched.
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VRO_FILTER_NEAREST = 0, VRO_FILTER_TRILINEAR = 1 };
enum { VRO_ACCUM_ITERATIVE = 0, VRO_ACCUM_CLOSED_FORM = 1 };
/* Rounding model of the GLSL built-ins normalize()/length()/dot() and of one algebraic
   fold (everything else is one correctly rounded binary32 op per GLSL operator in both):
     MESA  (default, the contract) what Mesa 23.2.1's GLSL compiler emits for the
           reference shader and llvmpipe executes, measured op by op with
           oracle/ref_gl/probe_arith.py: normalize(v) = v * (1/sqrt(dot)), dot summed
           last component to first, and view_top's 1-(1-z) folded to z.  In this mode
           the oracle reproduces the reference shader run under that GL BIT FOR BIT on
           every golden frame (tests/test_ref_gl_goldens.py); the HIP kernels are held
           to it bit for bit.
     SPEC  the GLSL specification text read literally: normalize(v) = v / sqrt(dot),
           dot summed first to last, no folding.  Kept to measure how far a different
           conformant GL may land from the pinned one (lattice-degenerate rays flip a
           voxel: ~0.05 % of cfg2's pixels move by more than 1e-4). */
enum { VRO_ARITH_MESA = 0, VRO_ARITH_SPEC = 1 };

typedef struct vro_params {
    /* image (imageSize(render_texture), VolumeRenderer.cs:57) */
    int32_t img_w, img_h;
    /* rows [row_begin,row_end) are rendered, with the GLOBAL row index used for
       the ray (multi-GPU row shards); others are left untouched.  */
    int32_t row_begin, row_end;
    /* Q1: faithful integer-division dispatch grid (RendererCore.cpp:121-122):
       pixels with x >= (W/16)*16 or y >= (H/16)*16 are never written. */
    int32_t trunc_grid;
    /* volume texture: x fastest, then y, then z (RendererCore.cpp:419) */
    int32_t nx, ny, nz;
    int32_t bytes_per_voxel;           /* 1 = R8UI, 2 = R16UI (host order) */
    const void *volume;
    /* Camera UBO block, 21 floats (VolumeRenderer.cs:28-36) */
    float cam[21];
    /* uniforms loc 0..6 (VolumeRenderer.cs:38-44) -- min/max are the values the
       shader sees, i.e. AFTER the +1000 of RendererCore.cpp:66-67,77-78 */
    float alpha_scale;
    float voxel_size[3];
    int32_t min_val, max_val;
    int32_t is_mip, view_top, view_bottom;
    /* build-defined switches */
    int32_t filter;                    /* VRO_FILTER_* */
    int32_t accum;                     /* VRO_ACCUM_*  */
    int32_t max_steps;                 /* 10000 in the reference (Q6) */
    /* optional 1-D transfer function: tf_len RGBA entries; NULL = reference
       grey ramp.  Index = windowed value scaled to [0,tf_len-1] (N3). */
    const float *tf_rgba;
    int32_t tf_len;
    int32_t threads;                   /* <=1: scalar single thread; >1: OpenMP rows */
    int32_t arith;                     /* VRO_ARITH_* */
    int32_t lerp_unfused;              /* TRILINEAR: 0 = each lerp is one fma (the definition: the executed GL's linear filter);
                                          1 = a + t * (b - a) in three roundings (rounds 1-4; measurement only) */
} vro_params;

/* Renders into rgba (img_h*img_w*4 floats, row 0 = bottom, GL convention).
   samples_per_pixel (optional, img_h*img_w u32) receives the number of texture
   fetches of each rendered pixel.  Returns total fetches via *total_samples.
   Return 0 on success. */
int vro_render(const vro_params *p, float *rgba, uint32_t *samples_per_pixel,
               uint64_t *total_samples);

/* ---- camera (src/Camera.cpp) ---- */
typedef struct vro_camera {
    float eye[4], side[4], up[4], look_at[4];
    float view2world[16];              /* column-major mat4 */
    float view_plane_dist, y_fov, rotation_speed, mov_speed;
    float zenith, azimuth, radius;
    int32_t is_changed;
} vro_camera;

void vro_camera_init(vro_camera *c, float y_fov, float rot_speed, float mov_speed);
void vro_camera_reset(vro_camera *c);
void vro_camera_orient(vro_camera *c, float zoom, float zenith, float azimuth);
void vro_camera_block(const vro_camera *c, float out21[21]);

/* ---- synthetic volumes; all-integer, seedable ---- */
/* cfg0/cfg1: u8 sphere.  r2=(2i+1-N)^2+...; v = r2>=(2R)^2 ? 0 : 255-(255*isqrt(r2))/(2R) */
void vro_gen_sphere_u8(uint8_t *out, int32_t n, int32_t radius);
/* cfg2/cfg3/cfg4: base + hashed noise; vmax = 4095 (u16) or 255 (u8) */
void vro_gen_noise_ball(void *out, int32_t nx, int32_t ny, int32_t nz,
                        int32_t bytes_per_voxel, uint32_t seed);

/* ---- natural cubic spline transfer function (src/CubicSpline.cpp) ---- */
/* knots: n entries of (iso_value, r,g,b,a) = 5 floats each, iso ascending.
   Fills lut_rgba[len*4], evaluating at iso = i*(iso_max/(len-1)); rgba clamped
   to [0,1] (AlphaControlSplineWidget.cpp:247). */
int vro_spline_tf(const float *knots5, int32_t n, float iso_max, float *lut_rgba,
                  int32_t len);

#ifdef __cplusplus
}
#endif
#endif
