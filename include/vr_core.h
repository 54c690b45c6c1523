/*
 * vr_core.h -- C ABI of libvr_core.so, the MI355X (gfx950) replacement for the GPU
 * ray-march path of gallickgunner/Volume-Renderer.
 *
 * The reference has no FFI: the boundary is the C++ surface of `RendererCore`
 * (include/RendererCore.h:9-44) as used by its only caller `RendererGUI`
 * (a friend class, src/RendererGUI.cpp).  Each entry point below names the
 * reference member / call site it replaces.  All calls are single-threaded per
 * handle, like the reference (everything runs on the GL thread).
 *
 * Conventions: functions returning `int` return 0 (VR_OK) on success and a
 * VR_E_* code otherwise; the text is available from vr_last_error().
 * Recoverable errors that the reference reports through its `title`/`msg`
 * strings (src/RendererGUI.cpp:90-95) are ALSO queued for vr_take_message().
 * Nothing here falls back to a CPU renderer: without a HIP device vr_render()
 * fails with VR_E_NO_DEVICE.
 */
#ifndef VR_CORE_H
#define VR_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vr_renderer *vr_handle;

enum {
    VR_OK = 0,
    VR_E_INVALID = 1,     /* bad argument / call order                        */
    VR_E_NO_DEVICE = 2,   /* no HIP device (or host-only handle) for a GPU op */
    VR_E_HIP = 3,         /* HIP runtime error                                */
    VR_E_IO = 4,          /* file open / parse error                          */
    VR_E_NOMEM = 5
};

/* NEAREST: what the reference's texture() yields -- its GL_LINEAR on an integer texture is refused by a conformant GL (all-zero
   frame) and read as NEAREST by the drivers it was written on; frames are bit-identical to the reference shader executed under
   Mesa llvmpipe (tests/golden/ref_gl/).  TRILINEAR (north-star's filter; no reference semantics): GL's linear rule with
   CLAMP_TO_EDGE, x then y then z, each lerp one fma -- bit-identical to llvmpipe's GL_LINEAR on an R32F texture of the same
   voxels through the reference shader with `usampler3D` spelled `sampler3D` (tests/golden/ref_gl/tri_*). */
enum { VR_FILTER_NEAREST = 0, VR_FILTER_TRILINEAR = 1 };
enum { VR_ACCUM_ITERATIVE = 0, VR_ACCUM_CLOSED_FORM = 1 };
enum { VR_LAYOUT_LINEAR = 0, VR_LAYOUT_BRICKED = 1 };
enum { VR_SYNTH_SPHERE_U8 = 0, VR_SYNTH_NOISE_BALL = 1,
       VR_SYNTH_NOISE_BALL_CT = 2 /* 16-bit only: the noise ball + 1000, i.e. stored the way the reference expects CT data (src/RendererCore.cpp:66-67) */ };

/* quirk switches (SURVEY.md section 8(a) quirk list); default = VR_QUIRK_DEFAULT */
enum {
    VR_QUIRK_TRUNC_GRID = 1u << 0,   /* Q1: dispatch W/16 x H/16 groups, drop the rest
                                        (src/RendererCore.cpp:121-122). Default OFF. */
    VR_QUIRK_U16_OFFSET = 1u << 1,   /* Q10: +1000 on min/max for 16-bit data
                                        (src/RendererCore.cpp:66-67,77-78). Default ON. */
    VR_QUIRK_DEFAULT = VR_QUIRK_U16_OFFSET
};

/* ---- lifetime: RendererCore::RendererCore / ~RendererCore / setup()
        (src/RendererCore.cpp:13-44; RendererGUI.cpp:38-40) ------------------- */
/* device >= 0: HIP device ordinal.  device = -1: host-only handle (camera, file
   parsing, transfer-function building work; every GPU operation fails loudly). */
int vr_create(vr_handle *out, int device);
void vr_destroy(vr_handle h);
/* window_size / framebuffer_size + setup(): allocates the RGBA32F off-screen
   target fb_w x fb_h in HBM (setupFBO, src/RendererCore.cpp:184-219). */
int vr_setup(vr_handle h, int win_w, int win_h, int fb_w, int fb_h);
const char *vr_last_error(vr_handle h);
/* msg/title polling (src/RendererGUI.cpp:90-95): returns 1 and fills the buffers
   if a message was pending (and clears it), 0 otherwise. */
int vr_take_message(vr_handle h, char *title, size_t title_cap, char *msg, size_t msg_cap);

/* ---- camera: Camera::setOrientation / resetCamera / setUBO
        (src/Camera.cpp:30-151; bound at RendererGUI.cpp:42-46, :363) ---------- */
int vr_camera_orient(vr_handle h, float zoom, float zenith, float azimuth);
int vr_camera_reset(vr_handle h);
/* Camera::setViewMatrix(eye, side, up, look_at) (include/Camera.h:18, src/Camera.cpp:46-57):
   four xyzw vectors; the members are normalised, the matrix is built from the arguments as
   given (columns side, up, -look_at, eye), exactly like the reference */
int vr_camera_set_view_matrix(vr_handle h, const float eye4[4], const float side4[4], const float up4[4],
                              const float look_at4[4]);
int vr_camera_set_block(vr_handle h, const float block21[21]);
int vr_camera_get_block(vr_handle h, float block21[21]);

/* ---- shader: loadShader(fn, reload) (src/RendererCore.cpp:112-136;
        RendererGUI.cpp:51,144,216).  The HIP kernel is built in; this records the
        name (so `loaded_shader` is non-empty) and computes workgroups_x/y. ------ */
int vr_load_shader(vr_handle h, const char *path, int reload);
int vr_workgroups(vr_handle h, int *wg_x, int *wg_y);
const char *vr_loaded_shader(vr_handle h);
const char *vr_loaded_dataset(vr_handle h);

/* ---- volume: readVolumeData / checkRawInfFile (src/RendererCore.cpp:46-54,
        242-447; RendererGUI.cpp:196-197,209) --------------------------------- */
int vr_check_raw_inf_file(vr_handle h, const char *path);          /* 1 = sidecar exists */
/* tex3D_dim / voxel_size as typed into the raw-inf panel (RendererGUI.cpp:423-429):
   used when a .raw has no .inf sidecar (the sidecar is then written). */
int vr_set_dims(vr_handle h, int nx, int ny, int nz);
int vr_set_spacing(vr_handle h, float sx, float sy, float sz);
/* datasize_bytes = 1|2 (RendererGUI.cpp:128,134).  64-bit sizes throughout (F5). */
int vr_read_volume_file(vr_handle h, const char *path, int datasize_bytes);
/* memory-based loaders (new; the reference only loads files): host pointer is
   copied, x fastest then y then z, host byte order. */
int vr_set_volume(vr_handle h, const void *host_voxels, int nx, int ny, int nz,
                  int datasize_bytes, float sx, float sy, float sz);
/* generate a synthetic volume directly in HBM (never crosses PCIe).
   kind = VR_SYNTH_SPHERE_U8 (param = radius in voxels), VR_SYNTH_NOISE_BALL or VR_SYNTH_NOISE_BALL_CT
   (param = seed). */
int vr_generate_synthetic(vr_handle h, int kind, int nx, int ny, int nz, int datasize_bytes,
                          uint32_t param);
/* copy the resident volume back to the host in x-fastest linear order */
int vr_read_volume(vr_handle h, void *host_voxels, size_t bytes);
int vr_get_dims(vr_handle h, int dims3[3], float spacing3[3], int *datasize_bytes);
/* min_dataset_val / max_dataset_val (src/RendererCore.cpp:360-384) */
int vr_get_dataset_range(vr_handle h, int *min_val, int *max_val);
/* 256-bin display histogram (src/RendererCore.cpp:386-405) */
int vr_histogram(vr_handle h, float hist256[256]);
/* measurement aid (SURVEY 8d "confirm the HBM peak on the box"): best-of-reps streaming read of
   the resident volume with 16-byte loads; *gbps = 1e9 bytes per second.  No reference
   counterpart (the reference reads GL_TIME_ELAPSED only, src/RendererCore.cpp:149-153). */
int vr_measure_stream_read(vr_handle h, int reps, double *gbps);
/* Multi-GPU, one process per GPU: on the rank that received the shards, turn the rank-major gather buffer
   -- n shards of local_rows x fb_w pixels, `channels` floats each (4: RGBA32F, 2: (grey, alpha), see
   vr_set_framebuffer_format) -- into the fb_w x fb_h RGBA32F frame the GUI blits (src/RendererGUI.cpp:158-162):
   one kernel on `hip_stream` (0 = the handle's stream) that undoes the row interleave (stripe_rows > 0: cyclic stripes
   as in vr_set_row_stripes; 0: contiguous blocks as in vr_set_row_range) and expands (grey, alpha).  Both pointers are
   device memory of this handle's device.  vr_group_render() does the same internally. */
int vr_assemble_shards(vr_handle h, const void *gathered_device, void *frame_device, int n, int local_rows, int stripe_rows,
                       int channels, void *hip_stream);

/* ---- PVM / DDS codec (host only, no handle): readPVMvolume of the reference's
        include/ddsbase.h:29-35 (src/ddsbase.cpp:768-858).  Returns a malloc'ed payload of
        width*height*depth*components bytes (free with vr_free) or NULL; 16-bit payloads
        are big-endian as stored (V^3 convention). ---------------------- */
unsigned char *vr_read_pvm_volume(const char *filename, unsigned int *width, unsigned int *height,
                                  unsigned int *depth, unsigned int *components, float *scalex,
                                  float *scaley, float *scalez);
/* the reference's checksum() (src/ddsbase.cpp:869-893) */
unsigned int vr_checksum(const unsigned char *data, unsigned int bytes);
void vr_free(void *p);

/* ---- uniforms: setAlpha / setMIP / setInitialCameraRotation / setMinVal /
        setMaxVal (src/RendererCore.cpp:56-98; RendererGUI.cpp:336-358,382-385) -- */
int vr_set_alpha(vr_handle h, float alpha_scale);
int vr_set_mip(vr_handle h, int use_mip);
int vr_set_view(vr_handle h, int rotate_to_top, int rotate_to_bottom); /* also resets camera */
int vr_set_window(vr_handle h, int min_val, int max_val);
int vr_get_window(vr_handle h, int *min_val, int *max_val);

/* ---- build-defined switches (no reference equivalent) ---------------------- */
int vr_set_filter(vr_handle h, int filter);          /* VR_FILTER_*  (F4)            */
int vr_set_accum(vr_handle h, int accum);            /* VR_ACCUM_*   (Q8)            */
int vr_set_quirks(vr_handle h, uint32_t quirks);     /* VR_QUIRK_*                   */
int vr_set_layout(vr_handle h, int layout);          /* VR_LAYOUT_* (default BRICKED); re-lays the volume out */
int vr_set_skip_empty(vr_handle h, int enable);      /* exact empty-space skipping (north-star's "adaptive step"): frames and sample
                                                        counts are bit-identical with it; NEAREST kernels per ray and batch, the
                                                        LDS-staged TRILINEAR kernel per tile and brick layer (round 5), tiles on
                                                        global taps per ray and batch.  No reference equivalent                  */
/* kernel selection: 0 = automatic (specialised kernels when the configuration allows; launches far from filling the
   chip -- fewer than 256 active 32x16 tiles, 1024 when the view is oblique to the volume axes -- use the 4-wavefront
   relay kernel; the fast kernel runs its software-pipelined batch loop unless alpha_scale >= 0.5),
   1 = always the generic line-by-line kernel (cross-check / debugging), 2 = the fast kernel with its
   plain loop: never the relay kernel, never the pipelined loop, 3 = automatic but always the relay kernel
   when the shape allows, 4 = retired (the NEAREST LDS-staged kernel of round 2 lost to the default kernels in every
   cell of the round-3 sweep, profiles/r03_work_model_sweep.txt, and was deleted; the value is refused),
   5 = the fast kernel with the pipelined loop, never the relay,
   6 = TRILINEAR on the LDS-staged kernel (the apron copy's bricks streamed into LDS, eight ds_read taps per
   sample; every mode incl. the transfer function; volumes beyond 4 GiB) wherever it is eligible,
   7 = that kernel with staging switched off: every tile takes the path of tiles whose brick layers do not fit LDS
   (pair loads from the apron copy, four samples' taps in flight) -- the cross-check of that path,
   8 / 9 = the LDS-staged kernel with one apron copy per major axis (so that every tile marches through layers of its
   own major axis with unit-stride DMA pieces) and the layer thickness chosen per tile: whole brick layers where they fit the tile's
   LDS ring, half layers (two voxels, 80-byte slots) where they do not -- 8 on 32x16-pixel tiles, 9 on 16x32-pixel tiles whose ring
   can also hold, per brick row, that row's own range (tiles whose bounding rectangles fit no other way: views near a body
   diagonal).  16-bit volumes; on 8-bit ones 8 runs as 6 and 9 as 6 on 16x32-pixel tiles.
   10 = the shape of 6 on 53 KiB of LDS: three workgroups per CU, six wavefronts per SIMD -- faster (6-13 %) wherever the tiles' brick
   layers fit (8-bit volumes, 16-bit ones up to ~512^3 or at 4K), slower where they do not; a candidate of the measured choice.
   11 = the LDS-staged kernel on 16x16-pixel tiles (four wavefronts, 40 KiB, four workgroups per CU; 16-bit volumes at oblique views
   with the per-axis copies and the thickness per tile): the first guess for volumes up to 640 voxels per axis, a candidate for
   launches that leave workgroup slots empty (round 6: cfg1 shape 0.138 -> 0.118 ms, cfg2 shape 0.402 -> 0.381).
   Frames are bit-identical under every variant. */
int vr_set_kernel_variant(vr_handle h, int variant);
/* 1 (default): under kernel variant 0 the launch is a MEASURED choice -- every candidate kernel of a configuration
   renders identical bits, so the first frames after a change of configuration (image, shard, volume, window, opacity,
   pose bucket; opacity and window width in coarse buckets, so a slider drag does not re-explore per frame) try the candidates
   in a shuffled order, twice each (HIP events on the launch stream, read without blocking), and the fastest is kept -- the
   heuristic's choice unless a rival is 2 % faster; a settled configuration is measured once more 96 frames later (clock
   ramp), entries are evicted least-recently-used first.  A configuration seen for fewer than three consecutive launches,
   and vr_set_autotune(0), get the heuristic's choice only.  Replaces round 2's tile-count thresholds; no reference equivalent (the reference has one shader). */
int vr_set_autotune(vr_handle h, int enable);
/* what the last launch ran as (for tests and tools; no reference equivalent): bit 0 relay kernel, bit 1 pipelined batch loop,
   bit 2 four-sample batches, bits 3..6 the LDS-staged trilinear kernel's shape (0 = not that kernel; the numbers of
   vr_set_kernel_variant 6 .. 11 minus 5), bit 8 (256): the measured choice is still EXPLORING this configuration -- the
   frame was a trial of one candidate (up to ~45 % slower than the settled choice), not the settled kernel */
int vr_get_launch_choice(vr_handle h);
/* The SETTLED entries of the measured choice as a flat blob, and back (round 6).  The reference dispatches one shader per frame and
   never tries anything (src/RendererCore.cpp:138-163); a handle that imports what another handle or an earlier process measured does
   the same for every configuration the blob knows: frame 1 runs on the settled kernel, no trial frames, no re-validation.
   vr_export_choices: *bytes = the size the blob needs (call with buf NULL to ask); it is written when capacity suffices, else VR_E_INVALID.
   vr_import_choices: *accepted = entries taken over; a blob measured on another device model or by another build of the library is
   well-formed but not trusted: 0 entries, VR_OK.  Malformed blobs: VR_E_INVALID. */
int vr_export_choices(vr_handle h, void *buf, size_t capacity, size_t *bytes);
int vr_import_choices(vr_handle h, const void *buf, size_t bytes, int *accepted);
/* 1 (default): when every voxel of a bricked 16-bit volume is <= 4095 (12-bit data) the
   specialised kernel gathers from a lossless 12-bit packed copy kept beside the volume (25 %
   fewer cache lines per frame; frames are bit-identical); 0: never.  Build-defined. */
int vr_set_pack12(vr_handle h, int enable);
/* bytes of the packed copy the last vr_render* gathered from (0: the launch used the volume as loaded) */
int vr_get_pack12_bytes(vr_handle h, size_t *bytes);
/* 1 (default): with VR_FILTER_TRILINEAR on the bricked layout the batched trilinear kernel gathers from an "apron"
   copy kept beside the volume -- every 4x4x4 brick stored as 5x4x4, i.e. with the x neighbours of its last column --
   so that the two x taps of a sample are one load for every lane (+25 % of the volume's bytes; 2.27 -> 1.8 ms on the
   1024^3 workload; frames are bit-identical); 0: never.  Build-defined, like the packed copy. */
int vr_set_trilinear_copy(vr_handle h, int enable);
/* bytes of the apron copy the last vr_render* gathered from (0: none) */
int vr_get_trilinear_copy_bytes(vr_handle h, size_t *bytes);
/* Device memory this handle holds right now (any pointer may be NULL): the volume as loaded (what the reference keeps
   in its 3-D texture, src/RendererCore.cpp:419); the optional speed copies built beside it (12-bit packed copy, apron
   copies: up to 0.75 + 3 x 1.25 volumes for a 16-bit volume viewed obliquely with TRILINEAR); everything else (targets,
   tables, skip grid, staging).  No reference equivalent (the GL driver owns the reference's memory). */
int vr_get_resident_bytes(vr_handle h, uint64_t *volume, uint64_t *copies, uint64_t *other);
/* Upper bound, in bytes, on the optional copies together.  VR_COPY_BUDGET_AUTO (default): a copy is built only while the
   device keeps max(1 GiB, a tenth of its memory) free after it, so that mandatory allocations that come later (skip grid,
   targets, a group's frame slots) still succeed.  0 = no copies at all.  Lowering the budget frees copies that no longer
   fit (per-axis aprons first, then the apron, then the packed copy); frames are bit-identical with or without any copy. */
#define VR_COPY_BUDGET_AUTO UINT64_MAX
int vr_set_copy_budget(vr_handle h, uint64_t bytes);
int vr_get_copy_budget(vr_handle h, uint64_t *bytes);
/* 1-D transfer function (N3): n knots of (iso in 0..255, r,g,b,a); n = 0 restores the
   reference grey ramp.  Built with the natural cubic spline of src/CubicSpline.cpp. */
int vr_set_transfer_function(vr_handle h, const int32_t *iso, const float *rgba4, int n);
int vr_get_transfer_lut(vr_handle h, float lut_rgba[256 * 4]);
/* image-row shard rendered by this handle: global rows [row_begin,row_end); default all. */
int vr_set_row_range(vr_handle h, int row_begin, int row_end);
/* cyclic row stripes: this handle renders stripes s with s % count == index, each
   `stripe_rows` rows tall (count = 1 disables). */
int vr_set_row_stripes(vr_handle h, int stripe_rows, int index, int count);
/* compact = 1: the target holds only this handle's rows, local row r of the shard at
   offset r*fb_w (contiguous shard: r = row - row_begin; stripes: stripe-major).  Only
   meaningful with an external target; vr_local_rows() gives its height in rows. */
int vr_set_framebuffer_compact(vr_handle h, int compact);
/* format of an EXTERNAL target: VR_FB_RGBA32F (default, 4 floats per pixel) or
   VR_FB_GREYALPHA32F (2 floats per pixel: the grey value and alpha; in the reference's grey
   modes r == g == b bit for bit, so this is the same frame in half the bytes -- what a
   multi-GPU gather moves).  Rendering with a transfer function into it fails (VR_E_INVALID). */
#define VR_FB_RGBA32F 0
#define VR_FB_GREYALPHA32F 1
int vr_set_framebuffer_format(vr_handle h, int format);
int vr_local_rows(vr_handle h);
/* launch on a caller-owned HIP stream (hipStream_t as void*); NULL = own stream */
int vr_set_stream(vr_handle h, void *hip_stream);
/* render into a caller-owned device buffer of fb_w*fb_h*4 floats; NULL = own target */
int vr_set_framebuffer_external(vr_handle h, void *device_rgba);

/* ---- render(): src/RendererCore.cpp:138-163 (RendererGUI.cpp:101) ----------- */
/* Launches the ray-march kernel into the off-screen target, times it with HIP
   events and adds the ms to kerneltime_sum (blocking, like the reference's
   GL_TIME_ELAPSED read-back). */
int vr_render(vr_handle h);
/* same launch without events or host synchronisation (for benches / graphs) */
int vr_render_async(vr_handle h);
int vr_synchronize(vr_handle h);
/* kerneltime_sum read-and-zero (RendererGUI.cpp:58,60) */
float vr_kernel_ms_take(vr_handle h);
/* instrumented (untimed) launch: number of volume fetches per pixel / total */
int vr_count_samples(vr_handle h, uint64_t *total, uint32_t *per_pixel, size_t n_pixels);
void *vr_framebuffer_device(vr_handle h);
int vr_read_pixels(vr_handle h, float *rgba, size_t n_floats);     /* D2H of the target */
/* the same frame as RGBA8 (fb_w*fb_h*4 bytes, row 0 = bottom like the target), converted on the
   device with glReadPixels' rule round(clamp(c,0,1)*255) (the precision of the reference's own
   read-back, src/RendererCore.cpp:170): a quarter of the bytes over PCIe for display */
int vr_read_pixels_rgba8(vr_handle h, unsigned char *rgba8, size_t n_bytes);
/* presentation for a display that is not this GPU (replaces the reference's on-GPU blit of the target to the back
   buffer, src/RendererCore.cpp:158-162): call once per frame after vr_render.  The frame is converted to RGBA8 on the
   launch stream and copied into one of THREE pinned host buffers on a separate copy stream, so the copy of frame i runs
   under the kernel of frame i + 1; *frame receives the PREVIOUS call's frame (fb_w x fb_h RGBA8, row 0 = bottom; on the
   first call: this frame, after a wait).  The pointer stays valid until the next-but-one call (call n hands out
   slot (n - 1) % 3, call n + 1 fills slot (n + 1) % 3, call n + 2 overwrites it).  One frame of display
   latency for an interactive loop that never waits on PCIe: 0.46 ms kernel + 8 MB copy overlapped, instead of 1.3 ms
   for kernel + blocking RGBA32F read-back. */
int vr_present_rgba8(vr_handle h, const unsigned char **frame);
/* saveImage(fn, ext) (src/RendererCore.cpp:165-182): ext ".png" | ".jpg" (quality 100, as
   :177) | ".bmp" of the reference's dialog (RendererGUI.cpp:221), plus ".ppm" */
int vr_save_image(vr_handle h, const char *path, const char *ext);
/* the writers behind vr_save_image for an RGB8 image already in host memory (rows top first,
   stride in bytes); host only, no handle.  Returns VR_OK or VR_E_IO / VR_E_INVALID. */
int vr_write_image_rgb8(const char *path, const char *ext, int width, int height,
                        const unsigned char *rgb, int stride_bytes);

/* ---- one node, several GPUs, one process (shards the single dispatch of
        src/RendererCore.cpp:149-151 so that a caller shaped like RendererGUI::run,
        src/RendererGUI.cpp:100-101, uses every GPU by calling vr_group_render where it called
        render()).  One renderer per device: the volume is replicated (load / generate it on every
        member), the image rows are sharded, the shards are gathered on devices[0] -- over RCCL
        (grouped ncclSend/ncclRecv across xGMI, librccl loaded on first use) when the devices are
        distinct, with peer copies otherwise -- and assembled there into one RGBA32F frame.
        Members are ordinary handles: configure them with the calls above (the same values on every
        member); their row range / stripes / target are owned by the group. ------------------------ */
typedef struct vr_group *vr_group_handle;
int vr_group_create(vr_group_handle *out, const int *devices, int n);
void vr_group_destroy(vr_group_handle g);
int vr_group_size(vr_group_handle g);
vr_handle vr_group_member(vr_group_handle g, int rank);
/* vr_setup on every member + the shard plan: partition 0 = cyclic stripes of `stripe_rows` rows
   (balanced: only ~75 % of the rows hit the box at the default camera), 1 = contiguous row blocks */
int vr_group_setup(vr_group_handle g, int win_w, int win_h, int fb_w, int fb_h, int partition, int stripe_rows);
/* 1 (default): RCCL when every member has its own device; 0: always peer copies; 2: like 1, and a
   one-member group also creates its communicator and every shard -- the root's own included --
   travels through a grouped ncclSend/ncclRecv (on a one-GPU box: a send to self, which executes
   the same RCCL calls, datatype, counts and stream wiring as the multi-GPU gather).
   Call before vr_group_setup. */
int vr_group_set_transport(vr_group_handle g, int use_rccl);
const char *vr_group_transport(vr_group_handle g);     /* what vr_group_setup chose */
/* render(): every member's shard kernel (concurrently, one stream per device), the gather and the
   assembly; blocks until the frame is complete on devices[0].  Adds the slowest member's kernel
   time to the group's kerneltime_sum.  == vr_group_render_async + vr_group_wait. */
int vr_group_render(vr_group_handle g);
/* the same frame without the block (replaces the reference's blocking timer read-back,
   src/RendererCore.cpp:152, for callers that can run a frame ahead): vr_group_render_async
   enqueues the next frame into one of THREE frame slots -- shard kernels on the members' render
   streams, gather and assembly on separate transfer streams -- and returns; at most two frames
   may be in flight (VR_E_INVALID otherwise), so the slot of the last completed frame is never the
   one a frame issued next is assembled into: what vr_group_framebuffer_device returned stays
   that frame until the next vr_group_wait.  vr_group_wait blocks until the OLDEST frame in
   flight is assembled and makes it the frame vr_group_framebuffer_device / vr_group_read_pixels
   return.  Keeping one frame in flight overlaps the gather of frame i with the kernels of
   frame i + 1. */
int vr_group_render_async(vr_group_handle g);
int vr_group_wait(vr_group_handle g);
float vr_group_kernel_ms_take(vr_group_handle g);
void *vr_group_framebuffer_device(vr_group_handle g);   /* fb_w x fb_h RGBA32F on devices[0] */
int vr_group_read_pixels(vr_group_handle g, float *rgba, size_t n_floats);
int vr_group_present_rgba8(vr_group_handle g, const unsigned char **frame);   /* vr_present_rgba8 of the assembled frame */
const char *vr_group_last_error(vr_group_handle g);

/* name of the kernel variant the last vr_render* launched (for profiles/tests) */
const char *vr_last_kernel_name(vr_handle h);

#ifdef __cplusplus
}
#endif
#endif /* VR_CORE_H */
