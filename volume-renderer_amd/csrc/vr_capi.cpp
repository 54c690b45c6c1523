// vr_capi.cpp -- extern "C" surface of libvr_core.so (include/vr_core.h) over
// vr::RendererCore.  Plays the role of the reference's `friend class RendererGUI`
// (include/RendererCore.h:18): it writes the public fields and calls the set*()
// methods exactly where the GUI does (file:line per function in vr_core.h).
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/vr_core.h"
#include "renderer_core.h"
#include "vr_handle.h"
#include "volume_io.h"


namespace {

thread_local std::string g_create_error;

int fail(vr_handle h, int code, const std::string &what)
{
    if (h) h->core.last_error = what;
    return code;
}

template <typename F>
int guarded(vr_handle h, F &&f)
{
    if (!h) return VR_E_INVALID;
    try {
        f(h->core);
        return VR_OK;
    } catch (const vr::NoDeviceError &e) {
        return fail(h, VR_E_NO_DEVICE, e.what());
    } catch (const vr::HipError &e) {
        return fail(h, VR_E_HIP, e.what());
    } catch (const vr::IoError &e) {
        return fail(h, VR_E_IO, e.what());
    } catch (const std::invalid_argument &e) {
        return fail(h, VR_E_INVALID, e.what());
    } catch (const std::bad_alloc &) {
        return fail(h, VR_E_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(h, VR_E_INVALID, e.what());
    }
}

void copy_str(char *dst, size_t cap, const std::string &s)
{
    if (!dst || cap == 0) return;
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(dst, s.data(), n);
    dst[n] = '\0';
}

}  // namespace

extern "C" {

int vr_create(vr_handle *out, int device)
{
    if (!out) return VR_E_INVALID;
    *out = nullptr;
    try {
        *out = new vr_renderer(device);
        return VR_OK;
    } catch (const vr::NoDeviceError &e) {
        g_create_error = e.what();
        return VR_E_NO_DEVICE;
    } catch (const vr::HipError &e) {
        g_create_error = e.what();
        return VR_E_HIP;
    } catch (const std::exception &e) {
        g_create_error = e.what();
        return VR_E_INVALID;
    }
}

void vr_destroy(vr_handle h) { delete h; }

const char *vr_last_error(vr_handle h) { return h ? h->core.last_error.c_str() : g_create_error.c_str(); }

int vr_setup(vr_handle h, int win_w, int win_h, int fb_w, int fb_h)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (win_w <= 0 || win_h <= 0 || fb_w <= 0 || fb_h <= 0) throw std::invalid_argument("vr_setup: sizes must be positive");
        c.window_size[0] = win_w; c.window_size[1] = win_h;
        c.framebuffer_size[0] = fb_w; c.framebuffer_size[1] = fb_h;
        c.setup();
    });
}

int vr_take_message(vr_handle h, char *title, size_t title_cap, char *msg, size_t msg_cap)
{
    if (!h) return 0;
    vr::RendererCore &c = h->core;
    if (c.msg.empty() && c.title.empty()) return 0;
    copy_str(title, title_cap, c.title);
    copy_str(msg, msg_cap, c.msg);
    c.title.clear();
    c.msg.clear();
    return 1;
}

int vr_camera_orient(vr_handle h, float zoom, float zenith, float azimuth)
{
    return guarded(h, [&](vr::RendererCore &c) { c.main_cam.setOrientation(zoom, zenith, azimuth); });
}

int vr_camera_reset(vr_handle h)
{
    return guarded(h, [&](vr::RendererCore &c) { c.main_cam.resetCamera(); c.main_cam.is_changed = true; });
}

int vr_camera_set_view_matrix(vr_handle h, const float eye4[4], const float side4[4], const float up4[4],
                              const float look_at4[4])
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!eye4 || !side4 || !up4 || !look_at4) throw std::invalid_argument("vr_camera_set_view_matrix: null vector");
        auto v = [](const float *p) { return vr::Vec4{p[0], p[1], p[2], p[3]}; };
        c.main_cam.setViewMatrix(v(eye4), v(side4), v(up4), v(look_at4));
        c.main_cam.is_changed = true;
    });
}

int vr_camera_set_block(vr_handle h, const float block21[21])
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!block21) throw std::invalid_argument("vr_camera_set_block: null block");
        c.main_cam.setBlock(block21);
    });
}

int vr_camera_get_block(vr_handle h, float block21[21])
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!block21) throw std::invalid_argument("vr_camera_get_block: null block");
        std::vector<float> v;
        const bool changed = c.main_cam.is_changed;
        c.main_cam.setUBO(v);
        c.main_cam.is_changed = changed;
        std::memcpy(block21, v.data(), sizeof(float) * 21);
    });
}

int vr_load_shader(vr_handle h, const char *path, int reload)
{
    if (!h) return VR_E_INVALID;
    int rc = VR_OK;
    const int g = guarded(h, [&](vr::RendererCore &c) {
        if (!c.loadShader(path ? path : "", reload != 0)) rc = VR_E_IO;
    });
    return g != VR_OK ? g : rc;
}

int vr_workgroups(vr_handle h, int *wg_x, int *wg_y)
{
    if (!h) return VR_E_INVALID;
    if (wg_x) *wg_x = h->core.workgroups_x;
    if (wg_y) *wg_y = h->core.workgroups_y;
    return VR_OK;
}

const char *vr_loaded_shader(vr_handle h) { return h ? h->core.loaded_shader.c_str() : ""; }
const char *vr_loaded_dataset(vr_handle h) { return h ? h->core.loaded_dataset.c_str() : ""; }

int vr_check_raw_inf_file(vr_handle h, const char *path)
{
    if (!h || !path) return 0;
    return h->core.checkRawInfFile(path) ? 1 : 0;
}

int vr_set_dims(vr_handle h, int nx, int ny, int nz)
{
    return guarded(h, [&](vr::RendererCore &c) { c.tex3D_dim[0] = nx; c.tex3D_dim[1] = ny; c.tex3D_dim[2] = nz; });
}

int vr_set_spacing(vr_handle h, float sx, float sy, float sz)
{
    return guarded(h, [&](vr::RendererCore &c) { c.voxel_size[0] = sx; c.voxel_size[1] = sy; c.voxel_size[2] = sz; });
}

int vr_read_volume_file(vr_handle h, const char *path, int datasize_bytes)
{
    if (!h) return VR_E_INVALID;
    int rc = VR_OK;
    const int g = guarded(h, [&](vr::RendererCore &c) {
        if (!path) throw std::invalid_argument("vr_read_volume_file: null path");
        if (datasize_bytes != 1 && datasize_bytes != 2) throw std::invalid_argument("datasize_bytes must be 1 or 2");
        c.datasize_bytes = datasize_bytes;          // RendererGUI.cpp:128,134
        c.title.clear(); c.msg.clear();
        c.readVolumeData(path);                     // RendererGUI.cpp:197
        if (c.title != "File Loaded!") { rc = VR_E_IO; if (c.last_error.empty()) c.last_error = c.msg; }
    });
    return g != VR_OK ? g : rc;
}

int vr_set_volume(vr_handle h, const void *host_voxels, int nx, int ny, int nz, int datasize_bytes, float sx,
                  float sy, float sz)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setVolume(host_voxels, nx, ny, nz, datasize_bytes, sx, sy, sz); });
}

int vr_generate_synthetic(vr_handle h, int kind, int nx, int ny, int nz, int datasize_bytes, uint32_t param)
{
    return guarded(h, [&](vr::RendererCore &c) { c.generateSynthetic(kind, nx, ny, nz, datasize_bytes, param); });
}

int vr_read_volume(vr_handle h, void *host_voxels, size_t bytes)
{
    return guarded(h, [&](vr::RendererCore &c) { c.readVolume(host_voxels, bytes); });
}

int vr_get_dims(vr_handle h, int dims3[3], float spacing3[3], int *datasize_bytes)
{
    if (!h) return VR_E_INVALID;
    for (int i = 0; i < 3; i++) {
        if (dims3) dims3[i] = h->core.tex3D_dim[i];
        if (spacing3) spacing3[i] = h->core.voxel_size[i];
    }
    if (datasize_bytes) *datasize_bytes = h->core.datasize_bytes;
    return VR_OK;
}

int vr_get_dataset_range(vr_handle h, int *min_val, int *max_val)
{
    if (!h) return VR_E_INVALID;
    if (min_val) *min_val = h->core.min_dataset_val;
    if (max_val) *max_val = h->core.max_dataset_val;
    return VR_OK;
}

int vr_histogram(vr_handle h, float hist256[256])
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!hist256) throw std::invalid_argument("vr_histogram: null buffer");
        c.computeHistogram(hist256);
    });
}

int vr_measure_stream_read(vr_handle h, int reps, double *gbps)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!gbps) throw std::invalid_argument("vr_measure_stream_read: null result");
        *gbps = c.measureStreamRead(reps);
    });
}

int vr_assemble_shards(vr_handle h, const void *gathered_device, void *frame_device, int n, int local_rows, int stripe_rows,
                       int channels, void *hip_stream)
{
    return guarded(h, [&](vr::RendererCore &c) { c.assembleShards(gathered_device, frame_device, n, local_rows, stripe_rows, channels, hip_stream); });
}

int vr_set_alpha(vr_handle h, float alpha_scale)
{
    return guarded(h, [&](vr::RendererCore &c) { c.alpha_scale = alpha_scale; c.setAlpha(); });   // RendererGUI.cpp:336-337
}

int vr_set_mip(vr_handle h, int use_mip)
{
    return guarded(h, [&](vr::RendererCore &c) { c.use_mip = use_mip != 0; c.setMIP(); });        // RendererGUI.cpp:342-343
}

int vr_set_view(vr_handle h, int rotate_to_top, int rotate_to_bottom)
{
    return guarded(h, [&](vr::RendererCore &c) {                                                  // RendererGUI.cpp:347-358
        c.rotate_to_top = rotate_to_top != 0;
        c.rotate_to_bottom = rotate_to_bottom != 0;
        c.setInitialCameraRotation();
    });
}

int vr_set_window(vr_handle h, int min_val, int max_val)
{
    return guarded(h, [&](vr::RendererCore &c) {                                                  // RendererGUI.cpp:382-385
        c.min_val = min_val; c.max_val = max_val;
        c.setMinVal(); c.setMaxVal();
    });
}

int vr_get_window(vr_handle h, int *min_val, int *max_val)
{
    if (!h) return VR_E_INVALID;
    if (min_val) *min_val = h->core.min_val;
    if (max_val) *max_val = h->core.max_val;
    return VR_OK;
}

int vr_set_filter(vr_handle h, int filter)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (filter != VR_FILTER_NEAREST && filter != VR_FILTER_TRILINEAR) throw std::invalid_argument("unknown filter");
        c.filter = filter;
        if (filter == VR_FILTER_TRILINEAR) c.warmTrilinear();       // code objects of the staged kernel: loaded here, not by a frame
    });
}

int vr_set_accum(vr_handle h, int accum)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (accum != VR_ACCUM_ITERATIVE && accum != VR_ACCUM_CLOSED_FORM) throw std::invalid_argument("unknown accum mode");
        c.accum = accum;
    });
}

int vr_set_quirks(vr_handle h, uint32_t quirks)
{
    return guarded(h, [&](vr::RendererCore &c) {
        c.setQuirks(quirks);
    });
}

int vr_set_layout(vr_handle h, int layout)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setLayout(layout); });
}

int vr_set_skip_empty(vr_handle h, int enable)
{
    return guarded(h, [&](vr::RendererCore &c) { c.skip_empty = enable != 0; });
}

int vr_set_autotune(vr_handle h, int enable)
{
    if (!h) return VR_E_INVALID;
    h->core.autotune = enable ? 1 : 0;
    return VR_OK;
}

int vr_set_kernel_variant(vr_handle h, int variant)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (variant < 0 || variant > 11 || variant == 4) throw std::invalid_argument("unknown kernel variant");   // 4: retired in round 3
        c.force_generic = variant;
    });
}

int vr_set_pack12(vr_handle h, int enable)
{
    return guarded(h, [&](vr::RendererCore &c) { c.pack12 = enable != 0; });
}

int vr_set_trilinear_copy(vr_handle h, int enable)
{
    return guarded(h, [&](vr::RendererCore &c) { c.tri_apron = enable != 0; });
}

int vr_get_trilinear_copy_bytes(vr_handle h, size_t *bytes)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!bytes) throw std::invalid_argument("vr_get_trilinear_copy_bytes: null result");
        *bytes = c.lastApronBytes();
    });
}

int vr_get_resident_bytes(vr_handle h, uint64_t *volume, uint64_t *copies, uint64_t *other)
{
    return guarded(h, [&](vr::RendererCore &c) {
        uint64_t v = 0, k = 0, o = 0;
        c.residentBytes(v, k, o);
        if (volume) *volume = v;
        if (copies) *copies = k;
        if (other) *other = o;
    });
}

int vr_set_copy_budget(vr_handle h, uint64_t bytes)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setCopyBudget(bytes); });
}

int vr_get_copy_budget(vr_handle h, uint64_t *bytes)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!bytes) throw std::invalid_argument("vr_get_copy_budget: null result");
        *bytes = c.copyBudget();
    });
}

int vr_get_pack12_bytes(vr_handle h, size_t *bytes)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!bytes) throw std::invalid_argument("vr_get_pack12_bytes: null result");
        *bytes = c.lastPacked12Bytes();
    });
}

int vr_set_transfer_function(vr_handle h, const int32_t *iso, const float *rgba4, int n)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setTransferFunction(iso, rgba4, n); });
}

int vr_get_transfer_lut(vr_handle h, float lut_rgba[256 * 4])
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (!lut_rgba) throw std::invalid_argument("vr_get_transfer_lut: null buffer");
        c.getTransferLut(lut_rgba);
    });
}

int vr_set_row_range(vr_handle h, int row_begin, int row_end)
{
    return guarded(h, [&](vr::RendererCore &c) {
        if (row_begin < 0 || (row_end >= 0 && row_end < row_begin)) throw std::invalid_argument("bad row range");
        c.setRowRange(row_begin, row_end);
    });
}

int vr_set_row_stripes(vr_handle h, int stripe_rows, int index, int count)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setRowStripes(stripe_rows, index, count); });
}

int vr_set_framebuffer_format(vr_handle h, int format)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setFramebufferFormat(format); });
}

int vr_set_framebuffer_compact(vr_handle h, int compact)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setFramebufferCompact(compact != 0); });
}

int vr_local_rows(vr_handle h) { return h ? h->core.localRows() : 0; }

int vr_set_stream(vr_handle h, void *hip_stream)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setStream(reinterpret_cast<hipStream_t>(hip_stream)); });
}

int vr_set_framebuffer_external(vr_handle h, void *device_rgba)
{
    return guarded(h, [&](vr::RendererCore &c) { c.setExternalFramebuffer(device_rgba); });
}

int vr_render(vr_handle h)
{
    return guarded(h, [&](vr::RendererCore &c) { c.render(); });
}

int vr_render_async(vr_handle h)
{
    return guarded(h, [&](vr::RendererCore &c) { c.renderAsync(); });
}

int vr_synchronize(vr_handle h)
{
    return guarded(h, [&](vr::RendererCore &c) { c.synchronize(); });
}

float vr_kernel_ms_take(vr_handle h)
{
    if (!h) return 0.0f;
    const float v = h->core.kerneltime_sum;
    h->core.kerneltime_sum = 0.0f;
    return v;
}

int vr_count_samples(vr_handle h, uint64_t *total, uint32_t *per_pixel, size_t n_pixels)
{
    return guarded(h, [&](vr::RendererCore &c) { c.countSamples(total, per_pixel, n_pixels); });
}

void *vr_framebuffer_device(vr_handle h) { return h ? h->core.framebufferDevice() : nullptr; }

int vr_read_pixels(vr_handle h, float *rgba, size_t n_floats)
{
    return guarded(h, [&](vr::RendererCore &c) { c.readPixels(rgba, n_floats); });
}

int vr_read_pixels_rgba8(vr_handle h, unsigned char *rgba8, size_t n_bytes)
{
    return guarded(h, [&](vr::RendererCore &c) { c.readPixelsRGBA8(rgba8, n_bytes); });
}

int vr_present_rgba8(vr_handle h, const unsigned char **frame)
{
    if (!frame) return VR_E_INVALID;
    *frame = nullptr;
    return guarded(h, [&](vr::RendererCore &c) { *frame = c.presentRGBA8(nullptr); });
}

int vr_save_image(vr_handle h, const char *path, const char *ext)
{
    if (!h) return VR_E_INVALID;
    int rc = VR_OK;
    const int g = guarded(h, [&](vr::RendererCore &c) {
        if (!path || !ext) throw std::invalid_argument("vr_save_image: null argument");
        if (!c.saveImage(path, ext)) { rc = VR_E_IO; c.last_error = "saveImage failed (unsupported extension or write error)"; }
    });
    return g != VR_OK ? g : rc;
}

int vr_write_image_rgb8(const char *path, const char *ext, int width, int height, const unsigned char *rgb,
                        int stride_bytes)
{
    if (!path || !ext || !rgb || width <= 0 || height <= 0 || stride_bytes < width * 3) return VR_E_INVALID;
    try {
        const std::string e = ext;
        bool ok = false;
        if (e == ".png") ok = vr::writePNG(path, width, height, rgb, stride_bytes);
        else if (e == ".jpg") ok = vr::writeJPEG(path, width, height, rgb, stride_bytes, 100);
        else if (e == ".bmp") ok = vr::writeBMP(path, width, height, rgb, stride_bytes);
        else if (e == ".ppm") ok = vr::writePPM(path, width, height, rgb, stride_bytes);
        else return VR_E_INVALID;
        return ok ? VR_OK : VR_E_IO;
    } catch (...) {
        return VR_E_IO;
    }
}

unsigned char *vr_read_pvm_volume(const char *filename, unsigned int *width, unsigned int *height,
                                  unsigned int *depth, unsigned int *components, float *scalex, float *scaley,
                                  float *scalez)
{
    if (!filename) return nullptr;
    vr::PvmVolume v;
    std::string err;
    try {
        if (!vr::readPVMvolume(filename, v, err)) { g_create_error = err; return nullptr; }
    } catch (const std::exception &e) {
        g_create_error = e.what();
        return nullptr;
    }
    if (!components && v.components != 1) { g_create_error = "multi-component PVM needs a components pointer"; return nullptr; }
    unsigned char *out = static_cast<unsigned char *>(std::malloc(v.data.size() ? v.data.size() : 1));
    if (!out) return nullptr;
    std::memcpy(out, v.data.data(), v.data.size());
    if (width) *width = v.width;
    if (height) *height = v.height;
    if (depth) *depth = v.depth;
    if (components) *components = v.components;
    if (scalex) *scalex = v.scalex;
    if (scaley) *scaley = v.scaley;
    if (scalez) *scalez = v.scalez;
    return out;
}

unsigned int vr_checksum(const unsigned char *data, unsigned int bytes)
{
    // sum of cipher*value with cipher = 271*cipher + value (mod 2^32), cipher0 = 1
    unsigned int sum = 0, cipher = 1;
    for (unsigned int i = 0; i < bytes; i++) {
        const unsigned int v = data[i];
        cipher = 271u * cipher + v;
        sum += cipher * v;
    }
    return sum;
}

void vr_free(void *p) { std::free(p); }

int vr_export_choices(vr_handle h, void *buf, size_t capacity, size_t *bytes)
{
    return guarded(h, [&](vr::RendererCore &c) {
        const size_t need = c.exportChoices(buf, capacity);
        if (bytes) *bytes = need;
        if (buf && capacity < need) throw std::invalid_argument("vr_export_choices: buffer too small");
    });
}

int vr_import_choices(vr_handle h, const void *buf, size_t bytes, int *accepted)
{
    return guarded(h, [&](vr::RendererCore &c) {
        const int n = c.importChoices(buf, bytes);
        if (accepted) *accepted = n;
    });
}

int vr_get_launch_choice(vr_handle h) { return h ? h->core.lastLaunchChoice() : 0; }
const char *vr_last_kernel_name(vr_handle h) { return h ? h->core.lastKernelName() : ""; }

}  // extern "C"
