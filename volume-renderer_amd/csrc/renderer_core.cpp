// renderer_core.cpp -- see renderer_core.h.  Host side of the ray-march path:
// owns the HBM objects (volume, RGBA32F target, transfer-function table), turns the
// reference's uniform/UBO state into the kernel's FrameParams and launches it.
// Compiled with -ffp-contract=off: buildFrame() repeats the fp32 operations of the
// shader's main() (VolumeRenderer.cs:62-83,109) in the shader's order.
#include "renderer_core.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "tile_schedule.h"
#include "volume_io.h"
#include "vr_kernels.h"

namespace vr {

namespace {
constexpr uint32_t kQuirkTruncGrid = 1u << 0, kQuirkU16Offset = 1u << 1;
constexpr int kLocalSize = 16;   // layout(local_size_x = 16, local_size_y = 16), VolumeRenderer.cs:3
}  // namespace

RendererCore::RendererCore(int device) : main_cam(30), histogram(256, 0.0f), device_(device)
{
    if (device_ >= 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) throw NoDeviceError("no HIP device available");
        if (device_ >= n) throw NoDeviceError("HIP device ordinal out of range");
        check(hipSetDevice(device_), "hipSetDevice");
        check(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking), "hipStreamCreate");
        check(hipEventCreate(&ev0_), "hipEventCreate");
        check(hipEventCreate(&ev1_), "hipEventCreate");
        for (TuneSlot &s : tune_slot_) { check(hipEventCreate(&s.ev0), "hipEventCreate"); check(hipEventCreate(&s.ev1), "hipEventCreate"); }
        check(hipMalloc(&d_scratch_, sizeof(unsigned) * 264), "hipMalloc(scratch)");
    }
}

RendererCore::~RendererCore()
{
    if (device_ >= 0) {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(stream());
        freeVolume();
        if (d_fb_) (void)hipFree(d_fb_);
        if (d_tf_) (void)hipFree(d_tf_);
        if (d_tile_table_) (void)hipFree(d_tile_table_);
        if (d_tile_table_tall_) (void)hipFree(d_tile_table_tall_);
        if (d_tile_table_small_) (void)hipFree(d_tile_table_small_);
        if (d_tile_work_) (void)hipFree(d_tile_work_);
        if (d_spp_) (void)hipFree(d_spp_);
        if (d_scratch_) (void)hipFree(d_scratch_);
        if (d_rgba8_) (void)hipFree(d_rgba8_);
        releasePresent();
        if (ev0_) (void)hipEventDestroy(ev0_);
        if (ev1_) (void)hipEventDestroy(ev1_);
        for (TuneSlot &s : tune_slot_) { if (s.ev0) (void)hipEventDestroy(s.ev0); if (s.ev1) (void)hipEventDestroy(s.ev1); }
        if (own_stream_) (void)hipStreamDestroy(own_stream_);
    }
}

void RendererCore::requireDevice(const char *what) const
{
    if (device_ < 0) throw NoDeviceError(std::string(what) + ": host-only handle, no HIP device");
    hipError_t e = hipSetDevice(device_);
    if (e != hipSuccess) throw HipError(e, std::string("hipSetDevice: ") + hipGetErrorString(e));
}

void RendererCore::check(hipError_t e, const char *what) const
{
    if (e != hipSuccess) throw HipError(e, std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------- setup / FBO / UBO
void RendererCore::setup()
{
    setupFBO();
}

void RendererCore::setupFBO()
{
    if (framebuffer_size[0] <= 0 || framebuffer_size[1] <= 0)
        throw std::runtime_error("Framebuffer not complete. Error code: zero-sized framebuffer");
    if (device_ < 0) return;   // host-only handle: nothing to allocate
    requireDevice("setupFBO");
    if (d_fb_) { check(hipFree(d_fb_), "hipFree(fb)"); d_fb_ = nullptr; }
    const size_t n = (size_t)framebuffer_size[0] * (size_t)framebuffer_size[1];
    check(hipMalloc(reinterpret_cast<void **>(&d_fb_), n * sizeof(float4)), "hipMalloc(framebuffer)");
    check(hipMemsetAsync(d_fb_, 0, n * sizeof(float4), stream()), "hipMemset(framebuffer)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
}

void RendererCore::setupUBO(bool)
{
    cam_block_.clear();
    main_cam.setUBO(cam_block_);
}

// ---------------------------------------------------------------- uniforms
void RendererCore::setAlpha()
{
    if (cs_program_) u_.alpha_scale = alpha_scale;
}

void RendererCore::setMinVal()
{
    if (cs_program_) u_.min_val = (datasize_bytes == 2 && (quirks & kQuirkU16Offset)) ? min_val + 1000 : min_val;
}

void RendererCore::setMaxVal()
{
    if (cs_program_) u_.max_val = (datasize_bytes == 2 && (quirks & kQuirkU16Offset)) ? max_val + 1000 : max_val;
}

void RendererCore::setMIP()
{
    if (cs_program_) u_.is_MIP = use_mip ? 1 : 0;
}

void RendererCore::setInitialCameraRotation()
{
    if (cs_program_) {
        main_cam.resetCamera();
        main_cam.is_changed = true;
        u_.view_top = rotate_to_top ? 1 : 0;
        u_.view_bottom = rotate_to_bottom ? 1 : 0;
    }
}

void RendererCore::setUniforms()
{
    if (cs_program_) {
        u_.voxel_size[0] = voxel_size[0];
        u_.voxel_size[1] = voxel_size[1];
        u_.voxel_size[2] = voxel_size[2];
    }
    setAlpha();
    setMinVal();
    setMaxVal();
    setMIP();
    setInitialCameraRotation();
}

bool RendererCore::loadShader(std::string fn, bool reload)
{
    // The kernel is compiled into this library; `fn` is only recorded so the GUI's
    // "shader loaded" state (RendererGUI.cpp:150) behaves the same.
    if (reload) fn = loaded_shader;
    if (fn.empty()) {
        setMessage("Error!", "Failed to open Shader file.");
        loaded_shader.clear();
        return false;
    }
    const size_t idx = fn.find_last_of('/');
    loaded_shader = idx == std::string::npos ? fn : fn.substr(idx + 1);
    cs_program_ = true;
    updateWorkgroups();
    // the reference compiles + links the shader here (:112-118); the HIP code objects are loaded now
    // instead of by the first render() (a cold first launch costs ~16 ms)
    if (device_ >= 0) {
        requireDevice("loadShader");
        check(launch_warm_modules(stream()), "module pre-load");
        check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
        tslab_warm_ = false;
        if (filter == 1) warmTrilinear();
    }
    u_.alpha_scale = alpha_scale;
    if (!loaded_dataset.empty()) {
        setUniforms();
        main_cam.resetCamera();
        main_cam.is_changed = true;
    }
    setMessage("Shader Loaded!", "Shader Loaded Successfully!");
    return true;
}

void RendererCore::updateWorkgroups()
{
    if (quirks & kQuirkTruncGrid) {
        workgroups_x = window_size[0] / kLocalSize;   // src/RendererCore.cpp:121-122
        workgroups_y = window_size[1] / kLocalSize;
    } else {
        workgroups_x = (window_size[0] + kLocalSize - 1) / kLocalSize;
        workgroups_y = (window_size[1] + kLocalSize - 1) / kLocalSize;
    }
}

void RendererCore::setQuirks(uint32_t q)
{
    quirks = q;
    setMinVal(); setMaxVal();                 // Q10 changes what the kernel sees as the window
    if (cs_program_) updateWorkgroups();      // Q1 changes the dispatch grid; nothing else is touched
}

// ---------------------------------------------------------------- volume
// bricks per axis of the VR_LAYOUT_BRICKED storage (vr_frame.h: BRICK_X x BRICK_Y x BRICK_Z voxels)
static inline uint32_t bricksX(int n) { return (uint32_t)((n + BRICK_X - 1) / BRICK_X); }
static inline uint32_t bricksY(int n) { return (uint32_t)((n + BRICK_Y - 1) / BRICK_Y); }
static inline uint32_t bricksZ(int n) { return (uint32_t)((n + BRICK_Z - 1) / BRICK_Z); }

size_t RendererCore::storageVoxels(int nx, int ny, int nz, int lay) const
{
    if (lay == 0) return (size_t)nx * (size_t)ny * (size_t)nz;
    return (size_t)bricksX(nx) * (size_t)bricksY(ny) * (size_t)bricksZ(nz) * 64u;
}

void RendererCore::freeVolume()
{
    if (d_vol_) { (void)hipFree(d_vol_); d_vol_ = nullptr; vol_alloc_bytes_ = 0; }
    res_dims_[0] = res_dims_[1] = res_dims_[2] = 0; res_bytes_ = 0;
    if (d_skip_grid_) { (void)hipFree(d_skip_grid_); d_skip_grid_ = nullptr; skip_grid_cells_ = 0; }
    if (tile_table_skip_sig_ != 0) tile_table_skip_sig_ = -1;            // (an order built on the old volume's visibility)
    if (d_vol12_) { (void)hipFree(d_vol12_); d_vol12_ = nullptr; vol12_bytes_ = 0; }
    vol12_failed_ = false;
    if (d_apron_) { (void)hipFree(d_apron_); d_apron_ = nullptr; apron_bytes_ = 0; }
    for (void *&q : d_apron_perm_) if (q) { (void)hipFree(q); q = nullptr; }
    apron_failed_ = false;
    apron_perm_failed_ = false;
}

void RendererCore::allocVolume(int nx, int ny, int nz, int bytes, int lay)
{
    requireDevice("volume upload");
    freeVolume();
    // + one x/y slab of slack so a one-past-the-edge index can never fault
    const size_t need = (storageVoxels(nx, ny, nz, lay) + (size_t)nx * (size_t)ny + 256) * (size_t)bytes;
    check(hipMalloc(&d_vol_, need), "hipMalloc(volume)");
    vol_alloc_bytes_ = need;
    vol_layout_ = lay;
    if (lay != 0) check(hipMemsetAsync(d_vol_, 0, need, stream()), "hipMemset(volume)");
}

void RendererCore::setVolume(const void *host, int nx, int ny, int nz, int bytes, float sx, float sy, float sz)
{
    if (!host || nx <= 0 || ny <= 0 || nz <= 0 || (bytes != 1 && bytes != 2))
        throw std::invalid_argument("setVolume: bad dimensions or datasize_bytes");
    requireDevice("setVolume");
    // nx*ny*nz*bytes must not wrap (three ints of up to 2^31 each): bound it at 2^48 bytes by division
    if ((uint64_t)nx > (1ull << 48) / (uint64_t)ny / (uint64_t)nz / (uint64_t)bytes)
        throw std::invalid_argument("setVolume: volume too large");
    const size_t lin_bytes = (size_t)nx * (size_t)ny * (size_t)nz * (size_t)bytes;
    const uint32_t bnx = bricksX(nx), bny = bricksY(ny);
    if (layout == 0) {
        allocVolume(nx, ny, nz, bytes, 0);
        check(hipMemcpyAsync(d_vol_, host, lin_bytes, hipMemcpyHostToDevice, stream()), "hipMemcpy(volume)");
    } else {
        void *staging = nullptr;
        check(hipMalloc(&staging, lin_bytes), "hipMalloc(staging)");
        hipError_t e = hipMemcpyAsync(staging, host, lin_bytes, hipMemcpyHostToDevice, stream());
        if (e == hipSuccess) {
            try { allocVolume(nx, ny, nz, bytes, 1); } catch (...) { (void)hipFree(staging); throw; }
            e = launch_relayout(staging, d_vol_, bytes, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz, bnx, bny, 0, stream());
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream());
        (void)hipFree(staging);
        check(e, "volume relayout");
    }
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    res_dims_[0] = nx; res_dims_[1] = ny; res_dims_[2] = nz; res_bytes_ = bytes;   // what is resident now
    tex3D_dim[0] = nx; tex3D_dim[1] = ny; tex3D_dim[2] = nz;
    voxel_size[0] = sx; voxel_size[1] = sy; voxel_size[2] = sz;
    datasize_bytes = bytes;
    afterVolumeLoaded("<memory>");
}

void RendererCore::generateSynthetic(int kind, int nx, int ny, int nz, int bytes, uint32_t param)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || (bytes != 1 && bytes != 2) || (kind != 0 && kind != 1 && kind != 2) ||
        (kind == 0 && bytes != 1) || (kind == 2 && bytes != 2))
        throw std::invalid_argument("generateSynthetic: bad arguments");
    if ((uint64_t)nx > (1ull << 48) / (uint64_t)ny / (uint64_t)nz / (uint64_t)bytes)
        throw std::invalid_argument("generateSynthetic: volume too large");
    requireDevice("generateSynthetic");
    allocVolume(nx, ny, nz, bytes, layout);
    check(launch_gen_volume(d_vol_, bytes, kind, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz, param, layout,
                            bricksX(nx), bricksY(ny), stream()),
          "gen_volume_kernel");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    res_dims_[0] = nx; res_dims_[1] = ny; res_dims_[2] = nz; res_bytes_ = bytes;   // what is resident now
    tex3D_dim[0] = nx; tex3D_dim[1] = ny; tex3D_dim[2] = nz;
    voxel_size[0] = voxel_size[1] = voxel_size[2] = 1.0f;
    datasize_bytes = bytes;
    afterVolumeLoaded(kind == 0 ? "<synthetic sphere>" : (kind == 1 ? "<synthetic noise ball>" : "<synthetic noise ball + 1000>"));
}

void RendererCore::readVolume(void *host, size_t bytes)
{
    requireDevice("readVolume");
    if (!d_vol_) throw std::runtime_error("readVolume: no dataset loaded");
    const size_t lin_bytes = (size_t)res_dims_[0] * res_dims_[1] * res_dims_[2] * (size_t)res_bytes_;
    if (!host || bytes < lin_bytes) throw std::invalid_argument("readVolume: buffer too small");
    if (vol_layout_ == 0) {
        check(hipMemcpyAsync(host, d_vol_, lin_bytes, hipMemcpyDeviceToHost, stream()), "hipMemcpy(D2H volume)");
        check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
        return;
    }
    void *staging = nullptr;
    check(hipMalloc(&staging, lin_bytes), "hipMalloc(staging)");
    hipError_t e = launch_relayout(d_vol_, staging, res_bytes_, (uint32_t)res_dims_[0], (uint32_t)res_dims_[1],
                                   (uint32_t)res_dims_[2], bricksX(res_dims_[0]),
                                   bricksY(res_dims_[1]), 1, stream());
    if (e == hipSuccess) e = hipMemcpyAsync(host, staging, lin_bytes, hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    (void)hipFree(staging);
    check(e, "readVolume");
}

void RendererCore::setLayout(int lay)
{
    if (lay != 0 && lay != 1) throw std::invalid_argument("setLayout: unknown layout");
    layout = lay;
    if (!d_vol_ || vol_layout_ == lay) return;
    requireDevice("setLayout");
    const int nx = res_dims_[0], ny = res_dims_[1], nz = res_dims_[2], bytes = res_bytes_;
    const int old_layout = vol_layout_;
    const size_t old_alloc = vol_alloc_bytes_;
    void *old = d_vol_;
    d_vol_ = nullptr;
    auto keep = [&]() { res_dims_[0] = nx; res_dims_[1] = ny; res_dims_[2] = nz; res_bytes_ = bytes; };   // allocVolume drops the resident state
    try { allocVolume(nx, ny, nz, bytes, lay); } catch (...) { d_vol_ = old; vol_layout_ = old_layout; vol_alloc_bytes_ = old_alloc; keep(); throw; }
    keep();
    hipError_t e = launch_relayout(old, d_vol_, bytes, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz,
                                   bricksX(nx), bricksY(ny), lay == 0 ? 1 : 0, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    (void)hipFree(old);
    check(e, "relayout");
}

// min/max scan of src/RendererCore.cpp:360-384 as a device reduction
void RendererCore::scanDatasetRange()
{
    // scratch: [0..1] reference-style min/max (index 8390640 skipped), [2..3] exact min/max
    const unsigned init[4] = {0xffffffffu, 0u, 0xffffffffu, 0u};
    check(hipMemcpyAsync(d_scratch_, init, sizeof(init), hipMemcpyHostToDevice, stream()), "hipMemcpy(scratch)");
    check(launch_stats(d_vol_, res_bytes_, (uint32_t)res_dims_[0], (uint32_t)res_dims_[1], (uint32_t)res_dims_[2],
                       vol_layout_, bricksX(res_dims_[0]), bricksY(res_dims_[1]), 0, 1.0f,
                       d_scratch_, d_scratch_ + 4, stream()),
          "stats_kernel");
    unsigned mm[4];
    check(hipMemcpyAsync(mm, d_scratch_, sizeof(mm), hipMemcpyDeviceToHost, stream()), "hipMemcpy(scratch)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    exact_min_ = (int)mm[2];
    exact_max_ = (int)mm[3];
    if (res_bytes_ == 2) {
        // reference initial values: max_value = -1, min_value = 9000000
        const int mx = mm[1] == 0u && mm[0] == 0xffffffffu ? -1 : (int)mm[1];
        const int mn = mm[0] == 0xffffffffu ? 9000000 : (int)mm[0];
        max_val = max_dataset_val = mx;
        min_val = min_dataset_val = mn;
    } else {
        min_val = min_dataset_val = 0;
        max_val = max_dataset_val = 255;
    }
}

void RendererCore::computeHistogram(float out[256])
{
    requireDevice("histogram");
    if (!d_vol_) throw std::runtime_error("histogram: no dataset loaded");
    check(hipMemsetAsync(d_scratch_, 0, sizeof(unsigned) * 264, stream()), "hipMemset(scratch)");
    check(launch_stats(d_vol_, res_bytes_, (uint32_t)res_dims_[0], (uint32_t)res_dims_[1], (uint32_t)res_dims_[2],
                       vol_layout_, bricksX(res_dims_[0]), bricksY(res_dims_[1]), 1,
                       (float)max_dataset_val, d_scratch_, d_scratch_ + 4, stream()),
          "stats_kernel");
    unsigned counts[256];
    check(hipMemcpyAsync(counts, d_scratch_ + 4, sizeof(counts), hipMemcpyDeviceToHost, stream()), "hipMemcpy(hist)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    // src/RendererCore.cpp:361,400-405: the normaliser starts from the dataset max
    // (16-bit) or -1 (8-bit) and is raised to the largest bin count
    double max_value = res_bytes_ == 2 ? (double)max_dataset_val : -1.0;
    for (int i = 1; i < 256; i++) max_value = std::max(max_value, (double)counts[i]);
    for (int i = 0; i < 256; i++) {
        histogram[i] = i == 0 ? 0.0f : (float)counts[i] * 100.0f / (float)max_value;
        out[i] = histogram[i];
    }
}

// the receiving rank's half of the multi-process gather (vr_assemble_shards): de-interleave + (grey, alpha) expansion
void RendererCore::assembleShards(const void *gathered, void *frame, int n, int local_rows, int stripe_rows, int channels, void *hip_stream)
{
    requireDevice("assembleShards");
    if (!gathered || !frame) throw std::invalid_argument("assembleShards: null buffer");
    if (framebuffer_size[0] <= 0 || framebuffer_size[1] <= 0) throw std::runtime_error("assembleShards: setup() has not run");
    if (n < 1 || local_rows < 1 || stripe_rows < 0 || (channels != 2 && channels != 4)) throw std::invalid_argument("assembleShards: bad shard geometry");
    // every frame row must exist in the gather buffer
    const long long rows_held = stripe_rows == 0 ? (long long)n * local_rows
                                                 : (long long)(local_rows / stripe_rows) * n * stripe_rows;
    if (rows_held < framebuffer_size[1] || (stripe_rows > 0 && local_rows % stripe_rows != 0)) throw std::invalid_argument("assembleShards: the shards do not cover the frame");
    check(launch_assemble(gathered, static_cast<float4 *>(frame), framebuffer_size[0], framebuffer_size[1], n, local_rows, stripe_rows, channels,
                          hip_stream ? static_cast<hipStream_t>(hip_stream) : stream()), "assemble kernel");
}

double RendererCore::measureStreamRead(int reps)
{
    requireDevice("measureStreamRead");
    if (!d_vol_) throw std::runtime_error("measureStreamRead: no dataset loaded");
    const uint64_t bytes = (uint64_t)storageVoxels(res_dims_[0], res_dims_[1], res_dims_[2], vol_layout_) * (uint64_t)res_bytes_ & ~15ull;
    hipEvent_t a = nullptr, b = nullptr;
    check(hipEventCreate(&a), "hipEventCreate");
    check(hipEventCreate(&b), "hipEventCreate");
    double best_ms = 1e30;
    hipError_t e = hipMemsetAsync(d_scratch_, 0, sizeof(unsigned), stream());
    for (int r = 0; r < std::max(reps, 1) + 1 && e == hipSuccess; r++) {      // first pass warms up
        e = hipEventRecord(a, stream());
        if (e == hipSuccess) e = launch_stream_read(d_vol_, bytes, d_scratch_, stream());
        if (e == hipSuccess) e = hipEventRecord(b, stream());
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (r > 0 && (double)ms < best_ms) best_ms = (double)ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    check(e, "stream_read_kernel");
    return (double)bytes / (best_ms * 1e-3) / 1e9;
}

void RendererCore::afterVolumeLoaded(const std::string &name)
{
    scanDatasetRange();
    setMessage("File Loaded!", "File Loaded Successfully!");
    if (!loaded_shader.empty()) {
        setUniforms();
        main_cam.resetCamera();
        main_cam.is_changed = true;
    }
    const size_t idx = name.find_last_of('/');
    loaded_dataset = idx == std::string::npos ? name : name.substr(idx + 1);
}

bool RendererCore::checkRawInfFile(std::string fn)
{
    std::ifstream inf_file(fn + ".inf");
    return bool(inf_file);
}

void RendererCore::readVolumeData(std::string fn)
{
    if (fn.size() < 3) { setMessage("Error!", "Failed to Open RAW file..."); return; }
    if (datasize_bytes != 1 && datasize_bytes != 2) {
        setMessage("Error!", "Select the data size (UINT8/UINT16) before loading a dataset.");
        return;
    }
    const std::string ext = fn.substr(fn.length() - 3, 3);
    std::vector<uint8_t> voxels;
    int dims[3];
    float spacing[3];
    if (ext == "raw") {
        if (checkRawInfFile(fn)) {
            RawInf inf;
            std::string t, m;
            if (!parseRawInf(fn + ".inf", inf, t, m)) { setMessage(t, m); return; }
            for (int i = 0; i < 3; i++) { dims[i] = inf.dims[i]; spacing[i] = inf.spacing[i]; }
        } else {
            // no sidecar: use the values typed into the GUI and write one (:304-317)
            for (int i = 0; i < 3; i++) { dims[i] = tex3D_dim[i]; spacing[i] = voxel_size[i]; }
            writeRawInf(fn + ".inf", dims, spacing);
        }
        const uint64_t len = (uint64_t)std::max(dims[0], 0) * (uint64_t)std::max(dims[1], 0) * (uint64_t)std::max(dims[2], 0);
        {
            std::ifstream probe(fn, std::ios::binary);
            if (!probe) { setMessage("Error!", "Failed to Open RAW file..."); return; }
        }
        if (len == 0) {
            setMessage("Invalid Data Size!",
                       "Texture Dimensions shouldn't contain any zeroes. Please provide a valid .raw.inf file.");
            return;
        }
        if (!readRawFile(fn, len * (uint64_t)datasize_bytes, voxels)) {
            setMessage("Error!", "Failed to Open RAW file...");
            return;
        }
    } else {
        PvmVolume pvm;
        std::string err;
        if (!readPVMvolume(fn, pvm, err)) { setMessage("Error!", "Error reading PVM file"); last_error = err; return; }
        dims[0] = (int)pvm.width; dims[1] = (int)pvm.height; dims[2] = (int)pvm.depth;
        spacing[0] = pvm.scalex; spacing[1] = pvm.scaley; spacing[2] = pvm.scalez;
        if ((int)pvm.components != datasize_bytes) {
            // the reference reinterprets the payload with the GUI-selected size and
            // reads out of bounds when they disagree; refuse instead
            setMessage("Error!", "Error reading PVM file");
            last_error = "PVM component count does not match datasize_bytes";
            return;
        }
        voxels.swap(pvm.data);   // Q9: 16-bit payload bytes are taken in host order, no swap (:347,368,393)
    }
    std::cout << "Dataset dimensions: " << dims[0] << ", " << dims[1] << ", " << dims[2] << std::endl;
    std::cout << "Dataset Aspect ratio: " << spacing[0] << ", " << spacing[1] << ", " << spacing[2] << std::endl;
    const int bytes = datasize_bytes;
    setVolume(voxels.data(), dims[0], dims[1], dims[2], bytes, spacing[0], spacing[1], spacing[2]);
    const size_t idx = fn.find_last_of('/');
    loaded_dataset = idx == std::string::npos ? fn : fn.substr(idx + 1);
}

// ---------------------------------------------------------------- transfer function
void RendererCore::setTransferFunction(const int32_t *iso, const float *rgba4, int n)
{
    if (n == 0) {
        tf_lut_.clear();
        return;
    }
    std::vector<float> lut;
    if (!iso || !rgba4 || !buildSplineLUT(iso, rgba4, n, lut))
        throw std::invalid_argument("setTransferFunction: need >= 2 knots with ascending iso values");
    tf_lut_.swap(lut);
    tf_grey_ = true;                         // r == g == b bit for bit in every entry (the reference's black -> white ramp always is)
    for (size_t e = 0; e < tf_lut_.size() / 4 && tf_grey_; e++)
        tf_grey_ = std::memcmp(&tf_lut_[4 * e], &tf_lut_[4 * e + 1], sizeof(float)) == 0 && std::memcmp(&tf_lut_[4 * e], &tf_lut_[4 * e + 2], sizeof(float)) == 0;
    if (device_ >= 0) {
        requireDevice("setTransferFunction");
        if (!d_tf_) check(hipMalloc(reinterpret_cast<void **>(&d_tf_), 256 * sizeof(float4)), "hipMalloc(tf)");
        check(hipMemcpyAsync(d_tf_, tf_lut_.data(), 256 * sizeof(float4), hipMemcpyHostToDevice, stream()), "hipMemcpy(tf)");
        check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    }
}

void RendererCore::getTransferLut(float *lut) const
{
    if (tf_lut_.empty()) {
        for (int i = 0; i < 256; i++) lut[4 * i] = lut[4 * i + 1] = lut[4 * i + 2] = lut[4 * i + 3] = (float)i / 255.0f;
    } else {
        std::memcpy(lut, tf_lut_.data(), sizeof(float) * 1024);
    }
}

// ---------------------------------------------------------------- render
void RendererCore::setRowStripes(int rows, int index, int count)
{
    if (count < 1 || index < 0 || index >= count || (count > 1 && rows < 1))
        throw std::invalid_argument("setRowStripes: bad stripe spec");
    stripe_rows_ = rows; stripe_index_ = index; stripe_count_ = count;
}

int RendererCore::localRows() const
{
    const int h = framebuffer_size[1];
    if (stripe_count_ > 1) {
        const int total = (h + stripe_rows_ - 1) / stripe_rows_;
        const int mine = (total - stripe_index_ + stripe_count_ - 1) / stripe_count_;
        return mine * stripe_rows_;
    }
    const int b = std::max(0, row_begin_), e = row_end_ < 0 ? h : std::min(row_end_, h);
    return std::max(0, e - b);
}

bool RendererCore::certifyDivisor(float b)
{
    uint32_t bits;
    std::memcpy(&bits, &b, 4);
    auto it = cert_cache_.find(bits);
    if (it != cert_cache_.end()) return it->second;
    bool ok = false;
    if (std::isfinite(b) && b > 1e-30f && b < 1e30f) {
        const float r = 1.0f / b;
        check(hipMemsetAsync(d_scratch_, 0, sizeof(unsigned), stream()), "hipMemset(scratch)");
        check(launch_certify_div(b, r, d_scratch_, stream()), "certify_div_kernel");
        unsigned bad = 1;
        check(hipMemcpyAsync(&bad, d_scratch_, sizeof(unsigned), hipMemcpyDeviceToHost, stream()), "hipMemcpy(scratch)");
        check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
        ok = (bad == 0);
    }
    cert_cache_[bits] = ok;
    return ok;
}

void RendererCore::buildFrame(FrameParams &P, LaunchConfig &L)
{
    std::memset(&P, 0, sizeof(P));
    std::memset(&L, 0, sizeof(L));
    if (cam_block_.size() != 21 || main_cam.is_changed) setupUBO(true);
    std::memcpy(P.cam, cam_block_.data(), sizeof(float) * 21);
    P.img_w = framebuffer_size[0];
    P.img_h = framebuffer_size[1];
    P.row_begin = std::max(0, row_begin_);
    P.row_end = row_end_ < 0 ? P.img_h : std::min(row_end_, P.img_h);
    if (quirks & kQuirkTruncGrid) {
        P.col_lim = (P.img_w / kLocalSize) * kLocalSize;
        P.row_lim = (P.img_h / kLocalSize) * kLocalSize;
    } else {
        P.col_lim = P.img_w;
        P.row_lim = P.img_h;
    }
    P.stripe_rows = stripe_count_ > 1 ? stripe_rows_ : 1;
    P.stripe_index = stripe_index_;
    P.stripe_count = stripe_count_;
    P.fb_compact = fb_compact_ ? 1 : 0;
    P.fb_format = (fb_format_ == 1 && ext_fb_) ? 1 : 0;
    const int nx = res_dims_[0], ny = res_dims_[1], nz = res_dims_[2];
    P.nx = nx; P.ny = ny; P.nz = nz;
    P.fdim[0] = (float)nx; P.fdim[1] = (float)ny; P.fdim[2] = (float)nz;
    P.bnx = (int)bricksX(nx); P.bny = (int)bricksY(ny); P.bnz = (int)bricksZ(nz);
    // ---- main(): VolumeRenderer.cs:65-83
    int max_dim = std::max(nx, ny);
    max_dim = std::max(max_dim, nz);
    const bool swz = (u_.view_bottom == 1 || u_.view_top == 1);
    const float d0 = (float)nx, d1 = swz ? (float)nz : (float)ny, d2 = swz ? (float)ny : (float)nz;
    const float s0 = u_.voxel_size[0], s1 = swz ? u_.voxel_size[2] : u_.voxel_size[1],
                s2 = swz ? u_.voxel_size[1] : u_.voxel_size[2];
    const float fmax_dim = (float)max_dim;
    const float pm[3] = {(d0 / fmax_dim) * s0, (d1 / fmax_dim) * s1, (d2 / fmax_dim) * s2};
    for (int i = 0; i < 3; i++) {
        P.half[i] = pm[i] / 2.0f;
        P.pmin[i] = 0.0f - P.half[i];
        P.pmax[i] = pm[i] - P.half[i];
        P.ext[i] = P.pmax[i] + P.half[i];
        P.rext[i] = 1.0f / P.ext[i];
    }
    // ---- step_size: VolumeRenderer.cs:109 (composite, .xzy) / :146 (MIP, .xyz)
    const float e0 = P.pmax[0] - P.pmin[0], e1 = P.pmax[1] - P.pmin[1], e2 = P.pmax[2] - P.pmin[2];
    // length(vec3) = sqrt of the dot summed from the last component to the first, as the reference's GL executes it
    // (oracle/ref_gl/probe_arith.py); .xzy in rayMarchVolume, .xyz in MIP
    const float num = std::sqrt((e2 * e2 + e1 * e1) + e0 * e0);
    const float fx = (float)nx, fy = (float)ny, fz = (float)nz;
    const float den = u_.is_MIP == 1 ? std::sqrt((fz * fz + fy * fy) + fx * fx) : std::sqrt((fy * fy + fz * fz) + fx * fx);
    P.step = num / den;
    P.alpha_scale = u_.alpha_scale;
    P.min_val = u_.min_val; P.max_val = u_.max_val;
    P.fmin = (float)u_.min_val; P.fmax = (float)u_.max_val;
    P.fden = (float)(u_.max_val - u_.min_val);
    P.rden = P.fden != 0.0f ? 1.0f / P.fden : 0.0f;
    P.view_top = u_.view_top; P.view_bottom = u_.view_bottom;
    P.max_steps = 10000;   // VolumeRenderer.cs:115
    P.accum = accum;
    P.tf_len = tf_lut_.empty() ? 0 : 256;
    P.skip_empty = skip_empty;

    L.bytes_per_voxel = res_bytes_;
    L.filter = filter;
    L.mip = u_.is_MIP;
    L.layout = vol_layout_;
    L.generic = force_generic == 1 ? 1 : 0;
    // kernel variants 6 .. 11: TRILINEAR on the LDS-staged kernel wherever it is eligible, in one of its shapes (0: see refreshTileSchedule)
    L.tri_slab = (force_generic >= 6 && force_generic <= 11) ? force_generic - 5 : 0;
    L.pipelined = 0;
    L.short_batches = 0;
    // 32-bit voxel offsets with 24-bit multiplies (VoxelAddr) whenever the volume allows
    {
        // VoxelAddr: the strides carry minus the part of the in-brick term the split axis repeats
        const uint64_t bsy = 64ull * (uint64_t)P.bnx - (BRICK_LY ? (uint64_t)(BRICK_X * BRICK_Y) : 0ull);
        const uint64_t bsz = 64ull * (uint64_t)P.bnx * (uint64_t)P.bny - (BRICK_LZ ? 64ull : 0ull);
        P.bstride_y = (uint32_t)bsy;
        P.bstride_z = (uint32_t)bsz;
        const uint64_t storage = storageVoxels(nx, ny, nz, vol_layout_);
        const uint64_t lim24 = 1ull << 24;
        bool small = storage < (1ull << 32) && (uint64_t)nx < lim24 && (uint64_t)ny < lim24 && (uint64_t)nz < lim24;
        if (vol_layout_ == 0) small = small && (uint64_t)ny * (uint64_t)nz < lim24;
        else small = small && bsy < lim24 && bsz < lim24;
        const uint64_t bytes = storage * (uint64_t)res_bytes_;
        if (bytes >= (1ull << 32)) small = false;
        L.big_offsets = small ? 0 : 1;
        L.vol_bytes32 = small ? (uint32_t)bytes : 0u;
        const int64_t width = (int64_t)u_.max_val - (int64_t)u_.min_val + 1;
        // LDS classification table: 4096 (c,a) entries, or with a transfer function one index byte per
        // window value behind the 256-entry RGBA table (vr_kernels.hip: FAST_TF_WINDOW_MAX)
        L.use_lut = (width >= 2 && width <= (tf_lut_.empty() ? 4096 : FAST_TF_WINDOW_MAX) && (tf_lut_.empty() || tf_lut_.size() / 4 <= 256)) ? 1 : 0;
        L.lut_noclamp = (exact_min_ >= u_.min_val && exact_max_ <= u_.max_val) ? 1 : 0;
        // a grey transfer function under NEAREST composite whose window fits the grey-ramp kernels' (c, a) table is folded
        // into that table: the launch runs on the MODE 0 instances (vr_kernels.hip: raymarch_fast_kernel, LUT build)
        P.tf_grey = (!tf_lut_.empty() && tf_grey_ && filter == 0 && u_.is_MIP != 1 && L.use_lut != 0 && width <= 4096) ? 1 : 0;   // (NEAREST only: the fold lives in the fast / relay kernels' tables)
        auto is_pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
        L.pow2_dims = (is_pow2(nx) && is_pow2(ny) && is_pow2(nz)) ? 1 : 0;
    }
    // division strategy: unit extents need no division at all; other divisors use the
    // 3-op Markstein quotient only after an exhaustive on-device certification
    const bool unit = P.ext[0] == 1.0f && P.ext[1] == 1.0f && P.ext[2] == 1.0f;
    if (unit) L.divmode_tc = DIV_UNIT;
    else L.divmode_tc = (certifyDivisor(P.ext[0]) && certifyDivisor(P.ext[1]) && certifyDivisor(P.ext[2])) ? DIV_CERT : DIV_EXACT;
    L.divmode_win = (P.fden > 0.0f && certifyDivisor(P.fden)) ? DIV_CERT : DIV_EXACT;
}

void RendererCore::setFramebufferFormat(int fmt)
{
    if (fmt != 0 && fmt != 1) throw std::invalid_argument("setFramebufferFormat: unknown format");
    fb_format_ = fmt;
}

// everything a launch needs that may touch the host or synchronise (certification, skip grid,
// tile order, packed copy): done BEFORE the timed region of render()
float4 *RendererCore::prepareLaunch(FrameParams &P, LaunchConfig &L)
{
    requireDevice("render");
    if (!cs_program_) throw std::runtime_error("render: no shader loaded (call loadShader first)");
    if (!d_vol_) throw std::runtime_error("render: no dataset loaded");
    float4 *fb = ext_fb_ ? reinterpret_cast<float4 *>(ext_fb_) : d_fb_;
    if (!fb) throw std::runtime_error("render: setup() has not allocated the framebuffer");
    if (fb_format_ == 1 && ext_fb_ && !tf_lut_.empty())
        throw std::invalid_argument("render: the (grey, alpha) target format needs a grey mode (no transfer function)");
    if (filter == 1) warmTrilinear();          // (a caller that set the public field directly; a no-op once done)
    buildFrame(P, L);
    refreshSkipGrid(P, L);
    refreshTileSchedule(P, L);
    refreshPacked12(P, L);
    refreshApron(P, L);
    // (the half-layer shapes need the per-axis copies: without them -- allocation failed -- the round-3 choice)
    if ((L.tri_slab == 3 || L.tri_slab == 4) && (L.apron_y == nullptr || L.apron_x == nullptr) && force_generic == 0) L.tri_slab = tri_path_candidate(P, L) ? 0 : 1;
    // the specialised kernels gather from the packed copy when their address tables fit (vr_kernels.hip: dispatch_fast3)
    last_packed12_bytes_ = (L.packed12 && P.nx + P.ny + P.nz <= 3072) ? (size_t)L.packed12_bytes : 0;
    tuneChoose(P, L);
    return fb;
}

// the LDS-staged TRILINEAR kernel's seven code objects (vr_tslab.hip), loaded when TRILINEAR is selected -- vr_set_filter,
// or vr_load_shader with the filter already set -- not by the first frame that needs one of them (round-4 advisor: the
// measured choice tries several shapes mid-interaction, each of which would pay a lazy decompress + load)
void RendererCore::warmTrilinear()
{
    if (device_ < 0 || tslab_warm_ || !cs_program_) return;
    requireDevice("warmTrilinear");
    check(launch_warm_tslab(stream()), "module pre-load (trilinear)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    tslab_warm_ = true;
}

void RendererCore::launch(uint32_t *spp)
{
    FrameParams P;
    LaunchConfig L;
    float4 *fb = prepareLaunch(P, L);
    // a measurement launch of the work model: events around the kernel, read later without blocking (tuneCollect)
    const bool measure = tune_measure_ && spp == nullptr && tune_count_ < kTuneSlots;
    TuneSlot &slot = tune_slot_[(tune_head_ + tune_count_) % kTuneSlots];
    if (measure) check(hipEventRecord(slot.ev0, stream()), "hipEventRecord");
    check(launch_raymarch(P, L, d_vol_, d_tf_, fb, spp, stream(), &last_kernel_), "raymarch launch");
    if (measure) {
        check(hipEventRecord(slot.ev1, stream()), "hipEventRecord");
        slot.key = tune_key_; slot.cand = tune_cand_; slot.gen = tune_gen_;
        tune_count_++;
        tuneIssued(tune_key_, tune_gen_, tune_cand_);                   // counted only now that its events are on the stream
    }
    tune_measure_ = false;
}

// TRILINEAR on 16-bit volumes: whole brick layers are the first guess -- and the per-axis copies are not built -- only while the
// central ray keeps this share of its length along one volume axis: at 0.973 (a fifth of a voxel of shear per voxel) the
// per-tile layer thickness already wins, 1.33 against 1.46 ms on cfg3; at 1.0 whole layers do, 1.17 against 1.20
static constexpr double kTriWholeLayerAlignment = 0.985;
// TRILINEAR: the small tile shape (16x16 pixels, four wavefronts, 40 KiB) is the FIRST GUESS where a tile's brick layers are small
// against its ring -- volumes up to 640 voxels along every axis (cfg1 shape 0.138 -> 0.118 ms, cfg2 shape 0.402 -> 0.381) -- and a
// CANDIDATE of the measured choice everywhere: on 1024^3 volumes it loses at the default pose (u16 1.12 -> 1.22 ms) and at strongly
// oblique ones (zenith 60 / azimuth 45: 1.55 -> 2.75), but wins over most of an orbit (u8: 0.92-1.02 against 1.01-1.11 ms at five of
// seven poses; u16 with the per-axis copies: 1.22 / 1.31 / 1.22 against 1.36 / 1.40 / 1.31; profiles/r06_trilinear_orbit.txt) -- which
// pose does which is exactly what the measurement is for.
static bool tri_small_tiles_offered(const FrameParams &P, const LaunchConfig &L, unsigned tile_active)
{
    (void)P; (void)tile_active;
    return L.tile_table_small != nullptr;
}

// ---- the measured work model.  Which kernel is fastest for a launch depends on how many tiles have work, how long
// their rays are, how early they end and how oblique the view is (tools/config_sweep.py: the relay kernel wins a
// 1024^3 shard with 171 active tiles by 25 %, loses a 256^3 frame with 560 by 45 %).  Round 2 chose by tile-count
// thresholds fitted on one configuration; here the launch is MEASURED: all candidates render identical bits, so the
// first frames of a configuration try them in turn (the heuristic's choice first), a few times each, and the fastest
// is kept.  Keyed by everything that shapes the launch; the pose enters through the view's axis alignment and the
// number of active tiles (buckets), so an orbiting camera re-uses what it has learnt.
void RendererCore::tuneIssued(uint64_t key, uint64_t gen, int cand_value)
{
    auto it = tune_.find(key);
    if (it == tune_.end() || it->second.gen != gen) return;
    for (int c = 0; c < it->second.ncand; c++) if (it->second.cand[c] == cand_value) { it->second.issued[c]++; return; }
}

// `cand_value` is the candidate's bit set, `gen` the entry's generation when the measurement was launched: an entry that was
// evicted and re-created for the same key has another shuffled order (an index would credit the wrong candidate), and a
// result launched before a re-validation must not count towards it (round-4 advisor)
void RendererCore::tuneRecord(uint64_t key, uint64_t gen, int cand_value, float ms)
{
    auto it = tune_.find(key);
    if (it == tune_.end() || it->second.gen != gen) return;
    int cand = -1;
    for (int c = 0; c < it->second.ncand; c++) if (it->second.cand[c] == cand_value) cand = c;
    if (cand < 0) return;
    TuneEntry &e = it->second;
    e.best_ms[cand] = e.tries[cand] == 0 ? ms : std::min(e.best_ms[cand], ms);
    e.tries[cand]++;
    bool all = true;
    for (int c = 0; c < e.ncand; c++) all = all && e.tries[c] >= kTuneTries;
    if (all) {
        // the heuristic's choice wins ties: a rival has to be 2 % faster, more than two event timings of one kernel differ by
        int heur = 0;
        for (int c = 0; c < e.ncand; c++) if (e.cand[c] == e.heur) heur = c;
        int best = heur;
        for (int c = 0; c < e.ncand; c++)
            if (c != heur && e.best_ms[c] < e.best_ms[best] * (best == heur ? 0.98f : 1.0f)) best = c;
        e.settled = best;
        e.frames_settled = 0;
    }
}

void RendererCore::tuneCollect()
{
    while (tune_count_ > 0) {                                            // the stream completes them in order
        TuneSlot &s = tune_slot_[tune_head_];
        const hipError_t q = hipEventQuery(s.ev1);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return; }
        tune_head_ = (tune_head_ + 1) % kTuneSlots;
        tune_count_--;
        if (q != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess) tuneRecord(s.key, s.gen, s.cand, ms);
        else (void)hipGetLastError();
    }
}

// ---- settled choices as a blob.  The reference dispatches ONE shader per frame and never tries anything
// (src/RendererCore.cpp:138-163); with an imported table neither does this library for the configurations the table knows.
#ifndef VR_BUILD_ID
#define VR_BUILD_ID 0ull
#endif
namespace {
struct ChoiceHeader { char magic[8]; uint32_t version, count; uint64_t build_id; char device[64]; };
struct ChoiceRecord { uint64_t key; int32_t ncand, settled_value, heur, cand[8]; };
static_assert(sizeof(ChoiceHeader) == 88 && sizeof(ChoiceRecord) == 56, "blob layout");
void choiceDeviceName(int device, char (&out)[64])
{
    // the device MODEL as a string of attributes, not hipDeviceProp_t::name: under rocprofv3 hipGetDeviceProperties came back
    // without a name (round 6: the profiled bench run rejected the blob the run before it had written), attribute queries do not differ
    std::memset(out, 0, sizeof(out));
    if (device < 0) return;
    int cus = 0, clk = 0, l2 = 0, bus = 0, major = 0, minor = 0;
    size_t mem = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, device);
    (void)hipDeviceGetAttribute(&l2, hipDeviceAttributeL2CacheSize, device);
    (void)hipDeviceGetAttribute(&bus, hipDeviceAttributeMemoryBusWidth, device);
    (void)hipDeviceGetAttribute(&major, hipDeviceAttributeComputeCapabilityMajor, device);
    (void)hipDeviceGetAttribute(&minor, hipDeviceAttributeComputeCapabilityMinor, device);
    (void)hipDeviceTotalMem(&mem, device);
    (void)hipGetLastError();
    std::snprintf(out, sizeof(out), "cc%d.%d cu%d clk%d l2_%d bus%d mem%zuG", major, minor, cus, clk / 1000, l2 >> 10, bus, mem >> 30);
}
}  // namespace

size_t RendererCore::exportChoices(void *buf, size_t capacity)
{
    static_assert(kTuneCand == 8, "ChoiceRecord::cand");
    tuneCollect();                                                       // measurements whose events have completed since the last launch count
    uint32_t n = 0;
    for (const auto &kv : tune_) if (kv.second.settled >= 0) n++;
    const size_t need = sizeof(ChoiceHeader) + (size_t)n * sizeof(ChoiceRecord);
    if (!buf || capacity < need) return need;
    ChoiceHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "VRCHOICE", 8);
    h.version = 1; h.count = n; h.build_id = (uint64_t)VR_BUILD_ID;
    choiceDeviceName(device_, h.device);
    std::memcpy(buf, &h, sizeof(h));
    char *q = static_cast<char *>(buf) + sizeof(h);
    for (const auto &kv : tune_) {
        const TuneEntry &e = kv.second;
        if (e.settled < 0) continue;
        ChoiceRecord r;
        std::memset(&r, 0, sizeof(r));
        r.key = kv.first; r.ncand = e.ncand; r.settled_value = e.cand[e.settled]; r.heur = e.heur;
        for (int c = 0; c < e.ncand; c++) r.cand[c] = e.cand[c];
        std::memcpy(q, &r, sizeof(r));
        q += sizeof(r);
    }
    return need;
}

int RendererCore::importChoices(const void *buf, size_t bytes)
{
    if (!buf || bytes < sizeof(ChoiceHeader)) throw std::invalid_argument("importChoices: blob too short");
    ChoiceHeader h;
    std::memcpy(&h, buf, sizeof(h));
    if (std::memcmp(h.magic, "VRCHOICE", 8) != 0 || h.version != 1) throw std::invalid_argument("importChoices: not a choices blob of this library");
    if (bytes < sizeof(h) + (size_t)h.count * sizeof(ChoiceRecord)) throw std::invalid_argument("importChoices: truncated blob");
    char mine[64];
    choiceDeviceName(device_, mine);
    if (h.build_id != (uint64_t)VR_BUILD_ID || std::memcmp(h.device, mine, sizeof(mine)) != 0) return 0;   // measured elsewhere: not trusted
    const char *q = static_cast<const char *>(buf) + sizeof(h);
    int accepted = 0;
    for (uint32_t i = 0; i < h.count && tune_.size() < (size_t)kTuneEntries; i++, q += sizeof(ChoiceRecord)) {
        ChoiceRecord r;
        std::memcpy(&r, q, sizeof(r));
        if (r.ncand < 2 || r.ncand > kTuneCand) continue;
        int settled = -1;
        for (int c = 0; c < r.ncand; c++) if (r.cand[c] == r.settled_value) settled = c;
        if (settled < 0) continue;
        TuneEntry e;
        e.ncand = r.ncand; e.heur = r.heur; e.settled = settled;
        for (int c = 0; c < r.ncand; c++) { e.cand[c] = r.cand[c]; e.tries[c] = e.issued[c] = kTuneTries; }
        e.revalidated = true;                                            // it was measured at sustained clocks where it came from
        e.last_use = tune_clock_;
        e.gen = ++tune_gen_counter_;
        tune_[r.key] = e;
        accepted++;
    }
    return accepted;
}

void RendererCore::tuneChoose(const FrameParams &P, LaunchConfig &L)
{
    tune_measure_ = false;
    tuneCollect();
    const int prior = L.filter == 1 ? (L.tri_slab << 3) : ((L.sparse_shard ? 1 : 0) | (L.pipelined ? 2 : 0) | (L.short_batches ? 4 : 0));
    last_choice_ = prior;
    if (!autotune || force_generic != 0 || !L.tile_table) return;
    // candidates, the heuristic's choice (what buildFrame / refreshTileSchedule left in L) first
    int cand[kTuneCand], n = 0;
    auto add = [&](int c) { for (int k = 0; k < n; k++) if (cand[k] == c) return; if (n < kTuneCand) cand[n++] = c; };
    if (fast_path_eligible(P, L)) {
        const bool relay_ok = !L.big_offsets && !(P.skip_empty != 0 && L.skip_grid != nullptr);
        add(prior & (relay_ok ? 7 : 6));
        add(P.alpha_scale >= 0.5f ? 4 : 2);                              // fast kernel: four-sample batches / pipelined loop
        add(0);                                                          // fast kernel, plain loop
        if (relay_ok) add(1);
    } else if (L.filter == 1 && L.apron != nullptr && tri_slab_candidate(P, L)) {
        // TRILINEAR: the LDS-staged kernel in its shapes, the batched kernel where it can run
        add(prior);
        add(1 << 3);
        add(5 << 3);                                                     // three workgroups per CU: wins where the tiles' layers fit 53 KiB
        // 16x16-pixel tiles on four wavefronts: launches whose 32x16-pixel tiles leave workgroup slots empty (two per CU)
        if (tri_small_tiles_offered(P, L, tile_active_)) add(6 << 3);
        if (tri_path_candidate(P, L)) add(0);
        if (L.apron_y != nullptr && L.apron_x != nullptr) {
            add(3 << 3);
            if (L.tile_table_tall != nullptr) add(4 << 3);
        } else if (L.bytes_per_voxel == 1 && L.tile_table_tall != nullptr && viewAxisAlignment(P) < 0.92) {
            add(4 << 3);                                                 // 8-bit volumes, oblique views: whole layers on 16x32-pixel tiles
        }
    }
    if (n < 2) return;
    uint64_t key = tileScheduleKey(P, launch_local_rows(P), false);
    auto mix = [&](uint64_t v) { key ^= v + 0x9E3779B97F4A7C15ull + (key << 6) + (key >> 2); };
    mix((uint64_t)(viewAxisAlignment(P) * 25.0));
    if (L.filter == 1) {
        // the staged trilinear kernel's time follows how many tiles' brick layers fit its ring, i.e. HOW the view is oblique
        // (orbit poses of one alignment bucket: 1.2 ... 2.5 ms on a 1024^3 u8 volume): both direction ratios, in eighths
        double r1, r2;
        viewAxisRatios(P, r1, r2);
        mix((uint64_t)(r1 * 8.0) << 8 | (uint64_t)(r2 * 8.0));
    }
    mix((uint64_t)(std::log2((double)std::max(tile_active_, 1u)) * 3.0));
    // opacity and window enter in coarse buckets (how early rays end, how wide the classification table is): a slider dragged
    // through hundreds of values stays in one or two entries instead of opening a new one -- and a new exploration -- per frame
    const double a = (double)P.alpha_scale;
    mix((uint64_t)(a >= 0.5 ? 1000 : (a > 0.0 ? (int)std::floor(std::log2(a) * 0.5) + 500 : 0)));
    mix((uint64_t)std::floor(std::log2((double)std::max(P.max_val - P.min_val, 1))) << 8 | (uint64_t)(L.use_lut != 0) << 1 | (uint64_t)(L.lut_noclamp != 0));
    mix((uint64_t)P.nx << 40 | (uint64_t)P.ny << 20 | (uint64_t)P.nz);
    mix((uint64_t)L.filter | (uint64_t)L.mip << 1 | (uint64_t)(P.tf_len > 1) << 2 | (uint64_t)(P.skip_empty != 0) << 3 | (uint64_t)L.layout << 4 |
        (uint64_t)L.bytes_per_voxel << 5 | (uint64_t)(L.packed12 != nullptr) << 8 | (uint64_t)(L.apron != nullptr) << 9 | (uint64_t)P.fb_format << 10 |
        (uint64_t)P.view_top << 11 | (uint64_t)P.view_bottom << 12 | (uint64_t)(L.apron_y != nullptr) << 13 | (uint64_t)prior << 16);
    tune_clock_++;
    tune_same_key_run_ = key == tune_last_key_ ? tune_same_key_run_ + 1 : 0;
    tune_last_key_ = key;
    auto it = tune_.find(key);
    if (it == tune_.end()) {
        // a configuration that is only passing through (an orbiting camera crossing a pose bucket, a slider crossing an opacity
        // bucket) gets the heuristic's choice; exploring starts when the same key has come three times in a row
        if (tune_same_key_run_ < 2) return;
        if (tune_.size() >= (size_t)kTuneEntries) {                      // evict the least recently used eighth
            std::vector<std::pair<uint64_t, uint64_t>> age;
            age.reserve(tune_.size());
            for (auto &kv : tune_) age.emplace_back(kv.second.last_use, kv.first);
            std::nth_element(age.begin(), age.begin() + kTuneEntries / 8, age.end());
            for (int k = 0; k < kTuneEntries / 8; k++) tune_.erase(age[(size_t)k].second);
        }
        it = tune_.emplace(key, TuneEntry{}).first;
        TuneEntry &e = it->second;
        e.ncand = n;
        e.gen = ++tune_gen_counter_;
        for (int k = 0; k < n; k++) e.cand[k] = cand[k];
        e.heur = cand[0];
        // shuffled measuring order (Fisher-Yates on a counter-seeded LCG)
        uint64_t r = key ^ (tune_clock_ * 0x9E3779B97F4A7C15ull);
        for (int k = n - 1; k > 0; k--) {
            r = r * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(e.cand[k], e.cand[(int)((r >> 33) % (uint64_t)(k + 1))]);
        }
    }
    TuneEntry &e = it->second;
    e.last_use = tune_clock_;
    int use;
    if (e.settled >= 0) {
        use = e.settled;
        // one re-measurement at sustained clocks: whatever settled during the clock ramp is checked again, once
        if (!e.revalidated && ++e.frames_settled >= kTuneRevalidateFrames) {
            e.revalidated = true;
            for (int c = 0; c < e.ncand; c++) { e.tries[c] = 0; e.issued[c] = 0; e.best_ms[c] = 0.0f; }
            e.settled = -1;
            e.next = 0;
            e.gen = ++tune_gen_counter_;                                 // results still in flight belong to the old generation
        }
    } else {
        // the next candidate that still needs a measurement launched; when every measurement is in flight (a burst of
        // asynchronous frames) or every event pair is taken: the best so far, the heuristic's choice before any result
        int pick = -1;
        if (tune_count_ < kTuneSlots)
            for (int k = 0; k < e.ncand && pick < 0; k++) {
                const int c = (e.next + k) % e.ncand;
                if (e.issued[c] < kTuneTries) pick = c;
            }
        if (pick >= 0) {
            use = pick;
            e.next = (pick + 1) % e.ncand;
            // issued[] is counted where the event pair is really recorded (launch / render): a launch that cannot measure
            // (the instrumented sample count, every slot taken) leaves the candidate to be picked again
            tune_measure_ = true; tune_key_ = key; tune_cand_ = e.cand[use]; tune_gen_ = e.gen;
        } else {
            int best = -1;
            for (int c = 0; c < e.ncand; c++)
                if (e.tries[c] > 0 && (best < 0 || e.best_ms[c] < e.best_ms[best])) best = c;
            if (best < 0)
                for (int c = 0; c < e.ncand; c++) if (e.cand[c] == e.heur) best = c;
            use = best < 0 ? 0 : best;
            // a measurement whose events were lost (a failed launch) would leave the entry waiting for ever: with nothing in
            // flight, whatever has not reported is launched again
            if (tune_count_ == 0)
                for (int c = 0; c < e.ncand; c++) if (e.tries[c] < kTuneTries) e.issued[c] = e.tries[c];
        }
    }
    const int c = e.cand[use];
    L.sparse_shard = (c & 1) ? 1 : 0; L.pipelined = (c & 2) ? 1 : 0; L.short_batches = (c & 4) ? 1 : 0; L.tri_slab = (c >> 3) & 15;
    last_choice_ = c | (e.settled < 0 ? 256 : 0);                        // bit 8: a trial frame of a configuration still being explored
}

// 12-bit packed copy (vr_set_pack12, default on): when the voxels of a bricked u16 volume span at most 4096
// values (max - min <= 4095: 12-bit CT data whatever its offset -- the reference's own +1000 convention,
// src/RendererCore.cpp:66-67, stores 1000 .. 5095 -- and BASELINE's synthetic volume) the fast kernel's prefix
// gathers from a lossless copy of (voxel - min) with 1.5 bytes per voxel: the launch is bound by the number of
// cache lines it moves, and this is 25 % fewer.  The minimum is folded into the window limits and
// the table bias the kernel already carries (FrameParams::pk12_base), so the sample loop is the same code.  The
// u16 volume stays resident for every other kernel and for the checked head / tail of each ray.
void RendererCore::refreshPacked12(FrameParams &P, LaunchConfig &L)
{
    L.packed12 = nullptr;
    L.packed12_bytes = 0;
    P.pk12_base = 0;
    if (!pack12 || res_bytes_ != 2 || vol_layout_ != 1 || L.big_offsets || exact_max_ - exact_min_ > 4095) return;
    if (!fast_path_eligible(P, L)) return;
    const size_t voxels = storageVoxels(res_dims_[0], res_dims_[1], res_dims_[2], 1);
    const size_t bytes = voxels / 2 * 3;
    if (bytes + 16 >= (1ull << 32)) return;
    if (!d_vol12_) {
        if (vol12_failed_) return;
        if (!copyFits(bytes + 16)) { vol12_failed_ = true; return; }      // (not asked again every frame; vr_set_copy_budget / the next volume re-arm it)
        if (hipMalloc(&d_vol12_, bytes + 16) != hipSuccess) {     // an optimisation only: render from the volume as loaded
            (void)hipGetLastError();
            d_vol12_ = nullptr;
            vol12_failed_ = true;
            return;
        }
        hipError_t e = launch_pack12(d_vol_, d_vol12_, voxels, (uint32_t)exact_min_, stream());
        vol12_base_ = exact_min_;
        if (e == hipSuccess) e = hipStreamSynchronize(stream());
        if (e != hipSuccess) { (void)hipFree(d_vol12_); d_vol12_ = nullptr; }
        check(e, "pack12_kernel");
        vol12_bytes_ = bytes;
    }
    L.packed12 = d_vol12_;
    L.packed12_bytes = (uint32_t)vol12_bytes_;
    P.pk12_base = vol12_base_;
}

// TRILINEAR's apron copy (vr_set_trilinear_copy, default on; vr_device.h: build_axis_tables_apron): the bricked
// volume once more with every 4x4x4 brick stored as 5x4x4, so that the two x taps of a sample are one load for
// every lane.  Built on first use from the resident volume, dropped with it; +25 % of the volume's bytes.
// Purely a speed device (2.27 -> ~1.8 ms on cfg3): the copy holds the same voxels, frames are bit-identical.
void RendererCore::refreshApron(const FrameParams &P, LaunchConfig &L)
{
    L.apron = nullptr;
    L.apron_y = L.apron_x = nullptr;
    L.apron_bytes = 0;
    last_apron_bytes_ = 0;
    if (!tri_apron || apron_failed_ || vol_layout_ != 1 || !(tri_path_candidate(P, L) || tri_slab_candidate(P, L))) return;
    const uint64_t bytes = apron_voxels(res_dims_[0], res_dims_[1], res_dims_[2]) * (uint64_t)res_bytes_;
    if (bytes + 16 >= (1ull << 32) && !tri_slab_candidate(P, L)) return;   // the batched kernel gathers through a 32-bit buffer descriptor
    if (!d_apron_) {
        if (!copyFits(bytes + 16)) { apron_failed_ = true; return; }   // over the budget / too little device memory left: not an error, and not asked again every frame
        if (hipMalloc(&d_apron_, bytes + 16) != hipSuccess) {         // an optimisation only
            (void)hipGetLastError();
            d_apron_ = nullptr;
            apron_failed_ = true;
            return;
        }
        hipError_t e = hipMemsetAsync(static_cast<char *>(d_apron_) + bytes, 0, 16, stream());
        if (e == hipSuccess)
            e = launch_relayout_apron(d_vol_, d_apron_, res_bytes_, (uint32_t)res_dims_[0], (uint32_t)res_dims_[1], (uint32_t)res_dims_[2],
                                      vol_layout_, (uint32_t)bricksX(res_dims_[0]), (uint32_t)bricksY(res_dims_[1]), 0, stream());
        if (e == hipSuccess) e = hipStreamSynchronize(stream());
        if (e != hipSuccess) { (void)hipFree(d_apron_); d_apron_ = nullptr; }
        check(e, "relayout_apron_kernel");
        apron_bytes_ = (size_t)bytes;
    }
    L.apron = d_apron_;
    L.apron_bytes = (uint64_t)apron_bytes_;
    last_apron_bytes_ = apron_bytes_;
    // Half layers of the staged kernel (vr_tslab.hip, 16-bit volumes): two more copies with the bricks' planes along y / x
    // slowest, so that half a brick along any major axis is 80 contiguous bytes.  Built the first time a view is oblique
    // to the volume axes (or kernel variants 8 / 9 ask for them); +2 x 1.25 volumes of HBM.
    // Only when a launch can use them (round-4 advisor): the staged kernel must be able to run (it walks the tile table),
    // and either a half-layer shape is forced (variants 8 / 9) or the automatic choice may pick one -- the heuristic's first
    // guess for an oblique view, or a candidate of the measured choice.  Both or none: a lone copy is freed again.
    const bool oblique = viewAxisAlignment(P) < kTriWholeLayerAlignment;
    const bool half_layers_possible = (force_generic >= 8 && force_generic <= 9) || (force_generic == 11 && oblique) ||
                                      (force_generic == 0 && oblique && (L.tri_slab >= 3 || autotune));
    const bool want_perm = res_bytes_ == 2 && tri_slab_candidate(P, L) && L.tile_table != nullptr && half_layers_possible;
    if (want_perm && !apron_perm_failed_ && !(d_apron_perm_[0] && d_apron_perm_[1])) {
        if (!copyFits(2 * (bytes + 16))) apron_perm_failed_ = true;     // (re-armed by vr_set_copy_budget and by the next volume)
        for (int o = 0; o < 2 && !apron_perm_failed_; o++) {
            if (d_apron_perm_[o]) continue;
            if (hipMalloc(&d_apron_perm_[o], bytes + 16) != hipSuccess) {
                (void)hipGetLastError();
                d_apron_perm_[o] = nullptr;
                apron_perm_failed_ = true;
                break;
            }
            hipError_t e = hipMemsetAsync(static_cast<char *>(d_apron_perm_[o]) + bytes, 0, 16, stream());
            if (e == hipSuccess)
                e = launch_relayout_apron(d_vol_, d_apron_perm_[o], res_bytes_, (uint32_t)res_dims_[0], (uint32_t)res_dims_[1], (uint32_t)res_dims_[2],
                                          vol_layout_, (uint32_t)bricksX(res_dims_[0]), (uint32_t)bricksY(res_dims_[1]), o + 1, stream());
            if (e == hipSuccess) e = hipStreamSynchronize(stream());
            if (e != hipSuccess) { (void)hipFree(d_apron_perm_[o]); d_apron_perm_[o] = nullptr; }
            check(e, "relayout_apron_kernel");
        }
        if (apron_perm_failed_)                                         // one without the other is 1.25 volumes of dead memory, in the low-memory case
            for (void *&q : d_apron_perm_) if (q) { (void)hipFree(q); q = nullptr; }
    }
    if (d_apron_perm_[0] && d_apron_perm_[1]) {
        L.apron_y = d_apron_perm_[0];
        L.apron_x = d_apron_perm_[1];
        last_apron_bytes_ += 2 * apron_bytes_;
    }
}

uint64_t RendererCore::copiesBytes() const
{
    uint64_t b = d_vol12_ ? (uint64_t)vol12_bytes_ + 16 : 0;
    if (d_apron_) b += (uint64_t)apron_bytes_ + 16;
    for (void *q : d_apron_perm_) if (q) b += (uint64_t)apron_bytes_ + 16;
    return b;
}

bool RendererCore::copyFits(uint64_t bytes) const
{
    if (copy_budget_ != kCopyBudgetAuto) return copiesBytes() + bytes <= copy_budget_;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return true; }
    const uint64_t reserve = std::max<uint64_t>(kCopyReserveBytes, (uint64_t)total_b / 10);
    return (uint64_t)free_b >= bytes + reserve;       // mandatory allocations (skip grid, targets, group slots) keep room
}

void RendererCore::setCopyBudget(uint64_t bytes)
{
    copy_budget_ = bytes;
    vol12_failed_ = apron_failed_ = apron_perm_failed_ = false;          // a new budget re-arms what an old one refused
    if (device_ < 0 || bytes == kCopyBudgetAuto) return;
    requireDevice("setCopyBudget");
    // over the new budget: drop copies, the least valuable first (per-axis aprons, apron, packed copy)
    if (copiesBytes() > bytes) check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    if (copiesBytes() > bytes) for (void *&q : d_apron_perm_) if (q) { (void)hipFree(q); q = nullptr; }
    if (copiesBytes() > bytes && d_apron_) { (void)hipFree(d_apron_); d_apron_ = nullptr; apron_bytes_ = 0; }
    if (copiesBytes() > bytes && d_vol12_) { (void)hipFree(d_vol12_); d_vol12_ = nullptr; vol12_bytes_ = 0; }
}

void RendererCore::residentBytes(uint64_t &volume, uint64_t &copies, uint64_t &other) const
{
    volume = (uint64_t)vol_alloc_bytes_;
    copies = copiesBytes();
    const uint64_t px = (uint64_t)framebuffer_size[0] * (uint64_t)framebuffer_size[1];
    other = (d_fb_ ? px * 16 : 0) + (d_tf_ ? 256 * 16 : 0) + (uint64_t)spp_capacity_ * 4 + (d_scratch_ ? 264 * 4 : 0) +
            (d_skip_grid_ ? (uint64_t)skip_grid_cells_ * 2 : 0) + (uint64_t)rgba8_capacity_ + (uint64_t)present_capacity_ * kPresentSlots +
            ((uint64_t)tile_table_capacity_ + (uint64_t)tile_table_tall_capacity_ + (uint64_t)tile_table_small_capacity_ + (uint64_t)tile_work_capacity_) * 4;
}

// Exact empty-space skipping (vr_set_skip_empty): the fast kernel may skip a batch of
// samples only if every voxel it could touch classifies to (0,0,0,0), so compositing it
// cannot change dest.  "Classifies to zero" is a prefix of the window: voxel <= min_val maps
// to v = 0 (grey ramp / MIP); with a transfer function the prefix is as long as its leading
// zero-opacity entries.  The grid holds, per 8^3-voxel cell, the maximum over the cell and
// its 26 neighbours.
void RendererCore::refreshSkipGrid(FrameParams &P, LaunchConfig &L)
{
    L.skip_grid = nullptr;
    L.skip_grid_bytes = 0;
    P.skip_empty = 0;
    // TRILINEAR on the LDS-staged kernel skips whole brick layers of a tile (vr_tslab.hip: SKIP): same grid, same threshold --
    // an interpolated value never exceeds its largest tap, and the classification of everything <= thresh is exactly zero
    const bool staged_tri = filter == 1 && tri_slab_candidate(P, L);
    if (!skip_empty || !(fast_path_eligible(P, L) || staged_tri)) return;
    const int nx = res_dims_[0], ny = res_dims_[1], nz = res_dims_[2];
    if (!staged_tri) {
        // a batch of 8 samples must stay within +-1 cell of its middle sample: bound the voxel
        // advance per step on every voxel axis (|dir| <= 1)
        float max_delta = 0.0f;
        const bool swz = (u_.view_bottom == 1 || u_.view_top == 1);
        const float vdim[3] = {(float)nx, swz ? (float)nz : (float)ny, swz ? (float)ny : (float)nz};   // voxel axis behind each box axis
        for (int a = 0; a < 3; a++) max_delta = std::max(max_delta, P.step * vdim[a] / P.ext[a]);
        if (!(4.0f * max_delta + 0.6f <= 8.0f)) return;
    }
    // threshold: largest voxel value whose classification is exactly zero
    int thresh;
    if (P.alpha_scale == 0.0f) {
        thresh = 65535;
    } else if (tf_lut_.empty()) {
        thresh = u_.min_val;                             // v = (clamp(t) - min)/den = 0  <=>  t <= min_val
    } else {
        int e = 0;
        const int width = u_.max_val - u_.min_val;
        // TRILINEAR classifies values BETWEEN the integers too: they reach every table entry up to the one of the largest tap
        // (the index is monotone in the value), so there the whole leading run of entries has to be invisible
        int zero_run = 0;
        while (zero_run < 256 && tf_lut_[4 * zero_run + 3] * P.alpha_scale == 0.0f) zero_run++;
        for (; e <= width; e++) {
            const float s = (float)(u_.min_val + e);
            const float v = (s - P.fmin) / P.fden;       // == the kernel's certified quotient
            int idx = (int)std::floor(v * 255.0f + 0.5f);
            idx = std::min(std::max(idx, 0), 255);
            if (staged_tri ? idx >= zero_run : tf_lut_[4 * idx + 3] * P.alpha_scale != 0.0f) break;
        }
        thresh = u_.min_val + e - 1;                     // e == 0: even the lowest entry is visible
        if (e == 0) return;
    }
    const uint32_t cnx = (uint32_t)(nx + 7) / 8, cny = (uint32_t)(ny + 7) / 8, cnz = (uint32_t)(nz + 7) / 8;
    const size_t cells = (size_t)cnx * cny * cnz;
    if (cells * 2 >= (1ull << 32) || cnx >= (1u << 24) || (uint64_t)cny * cnz >= (1u << 24)) return;
    if (!d_skip_grid_) {
        uint16_t *tmp = nullptr;
        check(hipMalloc(reinterpret_cast<void **>(&tmp), std::max<size_t>(cells * sizeof(uint16_t), 8)), "hipMalloc(skip grid tmp)");   // (also holds the one-word minimum below)
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_skip_grid_), cells * sizeof(uint16_t));
        if (e == hipSuccess)
            e = launch_build_skip_grid(d_vol_, res_bytes_, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz, vol_layout_,
                                       bricksX(nx), bricksY(ny), tmp, d_skip_grid_, stream());
        // ... and its smallest cell, into the scratch cell grid (one word), read back with the build's own synchronisation
        unsigned gmin = 0;
        if (e == hipSuccess) e = launch_min_cell(d_skip_grid_, (uint64_t)cells, reinterpret_cast<unsigned *>(tmp), stream());
        if (e == hipSuccess) e = hipMemcpyAsync(&gmin, tmp, sizeof(gmin), hipMemcpyDeviceToHost, stream());
        if (e == hipSuccess) e = hipStreamSynchronize(stream());
        (void)hipFree(tmp);
        if (e != hipSuccess && d_skip_grid_) { (void)hipFree(d_skip_grid_); d_skip_grid_ = nullptr; }
        check(e, "build skip grid");
        skip_grid_cells_ = cells;
        skip_grid_min_ = gmin;
    }
    // nothing to skip at this threshold (a window that starts at the data's floor): the launch takes the instances without
    // skipping -- theirs are the leaner loops (staged TRILINEAR 3.8 %, its tiles on global taps 20 %, the NEAREST kernels a
    // workgroup of occupancy).  A host comparison against the grid's smallest cell: no device work, no synchronisation per frame.
    if (thresh < (int)skip_grid_min_) return;
    P.skip_empty = 1;
    P.skip_thresh = thresh;
    P.cnx = (int32_t)cnx; P.cny = (int32_t)cny; P.cnz = (int32_t)cnz;
    L.skip_grid = d_skip_grid_;
    L.skip_grid_bytes = (uint32_t)(cells * sizeof(uint16_t));
}

// Rebuild the longest-first block order when the camera / image / shard changed.  The
// table is tiny (one word per 32x16-pixel tile) and is uploaded on the launch stream.
void RendererCore::refreshTileSchedule(const FrameParams &P, LaunchConfig &L)
{
    L.tile_table = nullptr;
    L.tile_table_blocks = 0;
    L.tile_table_tall = nullptr;
    L.tile_table_tall_blocks = 0;
    L.tile_table_small = nullptr;
    L.tile_table_small_blocks = 0;
    if (!tile_order || !(fast_path_eligible(P, L) || tri_path_candidate(P, L) || tri_slab_candidate(P, L))) return;
    const int rows = launch_local_rows(P);
    if (rows <= 0) return;
    // The table must list exactly the tiles of this image / shard (shape key); the camera only
    // decides their ORDER, a speed heuristic.  While the camera moves (GUI orbit: every frame) the
    // cached order is kept until the camera block has drifted noticeably, so that the host-side
    // rebuild + upload (~0.7 ms at 1080p) is not paid per frame.
    const uint64_t shape_key = tileScheduleKey(P, rows, false);
    float drift = 0.0f;
    for (int i = 0; i < 21; i++) drift = std::max(drift, std::fabs(P.cam[i] - tile_table_cam_[i]));
    const bool need32 = filter == 1 && tri_slab_candidate(P, L) && (P.stripe_count <= 1 || P.stripe_rows % 32 == 0 || force_generic == 9);   // 16x32-pixel tiles for the staged trilinear kernel's tall shape
    const bool need16 = filter == 1 && tri_slab_candidate(P, L);       // 16x16-pixel tiles for its small shape (four wavefronts; any stripe height that is a multiple of 16)
    // empty-space skipping: the order follows the tiles' VISIBLE work (vr_kernels.hip: tile_visible_work_kernel); it changes with
    // the threshold (window / transfer function), not only with the camera
    // (NEAREST skips per ray and batch: every pose.  The staged TRILINEAR kernel skips per tile and brick layer, and its tiles
    // that do not fit the ring skip nothing: where the view is oblique the estimate misjudges exactly the longest tiles -- the
    // off-axis pose ran 1.46 / 1.55 / 2.2 ms depending on where they landed, and still 1.15 / 1.65 from one process to the next
    // once those tiles skipped per batch -- so there the order stays the geometric one: 1.19-1.20 every time)
    const bool skip_order = P.skip_empty != 0 && L.skip_grid != nullptr && (filter == 0 || viewAxisAlignment(P) >= 0.92);
    // (in 32nds of the data's range: a window slider dragged under skipping would otherwise rebuild the table -- host sort, device
    // estimate, two synchronisations -- on every frame; the order is a heuristic and changes gradually with the threshold)
    const int64_t skip_sig = skip_order ? 1 + ((int64_t)std::max(P.skip_thresh, 0) * 32) / std::max<int64_t>((int64_t)exact_max_ + 1, 1) : 0;
    if (shape_key != tile_table_key_ || !d_tile_table_ || !(drift <= 0.05f) || (need32 && tile_table_tall_blocks_ == 0) || (need16 && tile_table_small_blocks_ == 0) ||
        skip_sig != tile_table_skip_sig_) {
        std::vector<uint32_t> table;
        std::vector<float> work, work_tall, work_small;
        if (skip_order) {
            // (one small kernel and a synchronous copy per table, only when the table is rebuilt; an empty sample costs about
            // a seventh of a sampled one in both kernel families)
            auto estimate = [&](unsigned tw, unsigned th, std::vector<float> &out) {
                const size_t n = (size_t)((P.img_w + (int)tw - 1) / (int)tw) * (size_t)((rows + (int)th - 1) / (int)th);
                if (2 * n > tile_work_capacity_) {
                    if (d_tile_work_) { check(hipFree(d_tile_work_), "hipFree(tile work)"); d_tile_work_ = nullptr; }
                    check(hipMalloc(reinterpret_cast<void **>(&d_tile_work_), 2 * n * sizeof(float)), "hipMalloc(tile work)");
                    tile_work_capacity_ = 2 * n;
                }
                check(launch_tile_visible_work(P, L.skip_grid, rows, tw, th, 0.15f, filter == 1 ? 1 : 0, d_tile_work_, stream()), "tile_visible_work_kernel");
                std::vector<float> both(2 * n);
                check(hipMemcpyAsync(both.data(), d_tile_work_, 2 * n * sizeof(float), hipMemcpyDeviceToHost, stream()), "hipMemcpy(tile work)");
                check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
                out.resize(n);
                for (size_t t = 0; t < n; t++) out[t] = both[2 * t + 1] > 0.0f ? std::min(both[2 * t] / both[2 * t + 1], 1.0f) : 1.0f;   // cost with skipping / without
            };
            estimate(kFastTileW, kFastTileH, work);
            if (need32) estimate(16u, 32u, work_tall);
            if (need16) estimate(16u, 16u, work_small);
        }
        tile_active_ = buildTileSchedule(P, rows, table, &tile_longest_, kFastTileH, kFastTileW, skip_order ? work.data() : nullptr);
        // synchronous copies: the tables are pageable temporaries
        check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
        auto upload = [&](uint32_t *&d, size_t &capacity, size_t &blocks, const std::vector<uint32_t> &t) {
            if (t.size() > capacity) {
                if (d) { check(hipFree(d), "hipFree(tile table)"); d = nullptr; }
                check(hipMalloc(reinterpret_cast<void **>(&d), t.size() * sizeof(uint32_t)), "hipMalloc(tile table)");
                capacity = t.size();
            }
            if (!t.empty()) check(hipMemcpy(d, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(tile table)");
            blocks = t.size();
        };
        upload(d_tile_table_, tile_table_capacity_, tile_table_blocks_, table);
        tile_table_tall_blocks_ = 0;
        if (need32) {
            std::vector<uint32_t> tall;
            (void)buildTileSchedule(P, rows, tall, nullptr, 32u, 16u, skip_order ? work_tall.data() : nullptr);
            upload(d_tile_table_tall_, tile_table_tall_capacity_, tile_table_tall_blocks_, tall);
        }
        tile_table_small_blocks_ = 0;
        if (need16) {
            std::vector<uint32_t> small;
            tile_active_small_ = buildTileSchedule(P, rows, small, nullptr, 16u, 16u, skip_order ? work_small.data() : nullptr);
            upload(d_tile_table_small_, tile_table_small_capacity_, tile_table_small_blocks_, small);
        }
        tile_table_key_ = shape_key;
        tile_table_skip_sig_ = skip_sig;
        std::memcpy(tile_table_cam_, P.cam, sizeof(tile_table_cam_));
    }
    L.tile_table = d_tile_table_;
    L.tile_table_blocks = (uint32_t)tile_table_blocks_;
    if (need32 && tile_table_tall_blocks_ > 0) { L.tile_table_tall = d_tile_table_tall_; L.tile_table_tall_blocks = (uint32_t)tile_table_tall_blocks_; }
    if (need16 && tile_table_small_blocks_ > 0) { L.tile_table_small = d_tile_table_small_; L.tile_table_small_blocks = (uint32_t)tile_table_small_blocks_; }
    // Kernel choice per launch (all bit-identical; measured on cfg3 after the checked-head fix, tools/pose_sweep.py and
    // tools/shard_ms.py; `aligned` = central ray within ~23 degrees of a volume axis):
    //   * relay kernel (4 wavefronts per 8x8 tile) for launches far from filling the chip -- fewer than 256 active
    //     32x16 tiles when the view is aligned (N = 8 shard: 0.107 vs 0.142 ms), fewer than 1024 when it is oblique
    //     (its rays are longer chains with more L1 misses per step: N = 2 shards 0.31-0.36 vs 0.36-0.47 ms);
    //   * otherwise the fast kernel, with its software-pipelined batch loop unless the opacity scale says rays end
    //     early (alpha >= 0.5: the speculative batch is wasted, 0.198 vs 0.189 ms): equal at the default pose
    //     (0.454 ms), 3-14 % faster over oblique poses, 12 % at an N = 4 shard (0.163 vs 0.185 ms).
    const bool aligned = viewAxisAlignment(P) >= 0.92;
    // TRILINEAR, first guess (the work model then measures): the LDS-staged kernel (vr_tslab.hip) -- whole brick layers, two
    // workgroups per CU -- when the view is aligned with a volume axis (cfg3 default pose: every tile staged, 1.25 vs 1.93 ms on
    // the batched kernel) and for 8-bit volumes at every pose (80-byte slots: orbit poses 1.1-1.4 ms against 2.1).  A 16-bit
    // volume's oblique layers do not fit three deep (240-400 bricks of 160 B): those views take the per-axis copies and half
    // layers (round 4: orbit poses 1.43-1.62 ms against 2.9-3.4 whole-layer / 2.6-3.0 batched), on 16x32-pixel tiles with the
    // ring's rows holding their own brick ranges when the view runs near a body diagonal of the volume (1.68 ms against 3.0).
    if (force_generic == 0 && filter == 1 && tri_slab_candidate(P, L)) {
        if (std::max(P.nx, std::max(P.ny, P.nz)) <= 640 && L.tile_table_small != nullptr) {
            L.tri_slab = 6;                                              // small volumes: 16x16-pixel tiles (round 6)
        } else if (viewAxisAlignment(P) >= kTriWholeLayerAlignment || L.bytes_per_voxel == 1) {
            L.tri_slab = 1;
        } else {
            double r1, r2;
            viewAxisRatios(P, r1, r2);
            // 16x32-pixel tiles only where 32 consecutive LOCAL rows are 32 consecutive image rows: with cyclic stripes of 16
            // rows a tall tile spans two stripes stripe_count*16 rows apart, its corner-ray hull (and its load plan) is that
            // much taller, and nearly every tile would march unstaged (round-4 advisor)
            const bool tall_ok = P.stripe_count <= 1 || P.stripe_rows % 32 == 0;
            L.tri_slab = (r1 > 0.7 && r2 > 0.55 && tall_ok) ? 4 : 3;
        }
    }
    // first guess of the work model (kernel variant 0 then measures, tuneChoose): the relay kernel pays when the launch is
    // a single under-filled round of long serial rays -- few active tiles, scaled by how long the rays are (1045 samples
    // on the configuration the 256 / 1024 were measured on; a 256^3 volume's 262-sample rays want 4x fewer tiles)
    const double len_scale = std::min(1.0, std::max(0.15, tile_longest_ / 1000.0));
    L.sparse_shard = ((double)tile_active_ < (aligned ? 256.0 : 1024.0) * len_scale) ? 1 : 0;
    if (force_generic == 2) L.sparse_shard = 0;   // kernel variant 2: never the relay kernel
    if (force_generic == 3) L.sparse_shard = 1;          // kernel variant 3: always (when the shape allows)
    L.pipelined = (force_generic == 0 && P.alpha_scale < 0.5f) ? 1 : 0;
    L.short_batches = (force_generic == 0 && P.alpha_scale >= 0.5f) ? 1 : 0;     // 0.174 vs 0.190 ms at alpha = 1 (cfg3)
    if (force_generic == 5) { L.pipelined = 1; L.sparse_shard = 0; }   // kernel variant 5: fast kernel, pipelined loop, never the relay
}

void RendererCore::render()
{
    requireDevice("render");
    // the reference brackets glDispatchCompute with a GL_TIME_ELAPSED query and blocks
    // on its result (src/RendererCore.cpp:149-153); same shape with HIP events
    // (GL_TIME_ELAPSED covers the dispatch only: all host-side preparation happens first)
    FrameParams P;
    LaunchConfig L;
    float4 *fb = prepareLaunch(P, L);
    const bool measure = tune_measure_;
    const uint64_t mkey = tune_key_, mgen = tune_gen_;
    const int mcand = tune_cand_;
    tune_measure_ = false;
    check(hipEventRecord(ev0_, stream()), "hipEventRecord");
    check(launch_raymarch(P, L, d_vol_, d_tf_, fb, nullptr, stream(), &last_kernel_), "raymarch launch");
    check(hipEventRecord(ev1_, stream()), "hipEventRecord");
    check(hipEventSynchronize(ev1_), "hipEventSynchronize");
    float ms = 0.0f;
    check(hipEventElapsedTime(&ms, ev0_, ev1_), "hipEventElapsedTime");
    kerneltime_sum += ms;
    if (measure) { tuneIssued(mkey, mgen, mcand); tuneRecord(mkey, mgen, mcand, ms); }   // the blocking path measures anyway
}

void RendererCore::renderAsync()
{
    launch(nullptr);
}

void RendererCore::synchronize()
{
    requireDevice("synchronize");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
}

void RendererCore::countSamples(uint64_t *total, uint32_t *per_pixel, size_t n_pixels)
{
    requireDevice("countSamples");
    const size_t n = (size_t)framebuffer_size[0] * (size_t)framebuffer_size[1];
    if (per_pixel && n_pixels < n) throw std::invalid_argument("countSamples: per_pixel buffer too small");
    if (spp_capacity_ < n) {
        if (d_spp_) { check(hipFree(d_spp_), "hipFree(spp)"); d_spp_ = nullptr; }
        check(hipMalloc(reinterpret_cast<void **>(&d_spp_), n * sizeof(uint32_t)), "hipMalloc(spp)");
        spp_capacity_ = n;
    }
    check(hipMemsetAsync(d_spp_, 0, n * sizeof(uint32_t), stream()), "hipMemset(spp)");
    launch(d_spp_);
    std::vector<uint32_t> host(n);
    check(hipMemcpyAsync(host.data(), d_spp_, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream()), "hipMemcpy(spp)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
    uint64_t sum = 0;
    for (uint32_t v : host) sum += v;
    if (total) *total = sum;
    if (per_pixel) std::memcpy(per_pixel, host.data(), n * sizeof(uint32_t));
}

void RendererCore::readPixels(float *rgba, size_t n_floats)
{
    requireDevice("readPixels");
    const size_t n = (size_t)framebuffer_size[0] * (size_t)framebuffer_size[1] * 4;
    if (!rgba || n_floats < n) throw std::invalid_argument("readPixels: buffer too small");
    if (ext_fb_ && (fb_compact_ || fb_format_ == 1))
        throw std::invalid_argument("readPixels: the external target is compact and/or (grey, alpha): it does not hold fb_w x fb_h RGBA32F pixels");
    const void *src = framebufferDevice();
    if (!src) throw std::runtime_error("readPixels: no framebuffer");
    check(hipMemcpyAsync(rgba, src, n * sizeof(float), hipMemcpyDeviceToHost, stream()), "hipMemcpy(D2H framebuffer)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
}

void RendererCore::readPixelsRGBA8(uint8_t *rgba8, size_t n_bytes)
{
    requireDevice("readPixelsRGBA8");
    const size_t n = (size_t)framebuffer_size[0] * (size_t)framebuffer_size[1];
    if (!rgba8 || n_bytes < n * 4) throw std::invalid_argument("readPixelsRGBA8: buffer too small");
    if (ext_fb_ && (fb_compact_ || fb_format_ == 1))
        throw std::invalid_argument("readPixelsRGBA8: the external target is compact and/or (grey, alpha): it does not hold fb_w x fb_h RGBA32F pixels");
    const void *src = framebufferDevice();
    if (!src) throw std::runtime_error("readPixelsRGBA8: no framebuffer");
    if (rgba8_capacity_ < n * 4) {
        if (d_rgba8_) { (void)hipFree(d_rgba8_); d_rgba8_ = nullptr; rgba8_capacity_ = 0; }
        check(hipMalloc(&d_rgba8_, n * 4), "hipMalloc(rgba8)");
        rgba8_capacity_ = n * 4;
    }
    check(launch_to_rgba8(src, d_rgba8_, n, stream()), "to_rgba8_kernel");
    check(hipMemcpyAsync(rgba8, d_rgba8_, n * 4, hipMemcpyDeviceToHost, stream()), "hipMemcpy(D2H rgba8)");
    check(hipStreamSynchronize(stream()), "hipStreamSynchronize");
}

void RendererCore::releasePresent()
{
    if (present_stream_) (void)hipStreamSynchronize(present_stream_);
    for (int s = 0; s < kPresentSlots; s++) {
        if (d_present_[s]) { (void)hipFree(d_present_[s]); d_present_[s] = nullptr; }
        if (h_present_[s]) { (void)hipHostFree(h_present_[s]); h_present_[s] = nullptr; }
        if (present_converted_[s]) { (void)hipEventDestroy(present_converted_[s]); present_converted_[s] = nullptr; }
        if (present_copied_[s]) { (void)hipEventDestroy(present_copied_[s]); present_copied_[s] = nullptr; }
    }
    if (present_stream_) { (void)hipStreamDestroy(present_stream_); present_stream_ = nullptr; }
    present_capacity_ = 0;
    present_count_ = 0;
}

const uint8_t *RendererCore::presentRGBA8(const void *src_frame)
{
    requireDevice("presentRGBA8");
    const size_t n = (size_t)framebuffer_size[0] * (size_t)framebuffer_size[1];
    if (n == 0) throw std::runtime_error("presentRGBA8: setup() has not sized the framebuffer");
    if (!src_frame) {
        if (ext_fb_ && (fb_compact_ || fb_format_ == 1))
            throw std::invalid_argument("presentRGBA8: the external target is compact and/or (grey, alpha): it does not hold fb_w x fb_h RGBA32F pixels");
        src_frame = framebufferDevice();
        if (!src_frame) throw std::runtime_error("presentRGBA8: no framebuffer");
    }
    if (present_capacity_ != n * 4) {
        releasePresent();
        check(hipStreamCreateWithFlags(&present_stream_, hipStreamNonBlocking), "hipStreamCreate(present)");
        for (int s = 0; s < kPresentSlots; s++) {
            check(hipMalloc(&d_present_[s], n * 4), "hipMalloc(present)");
            check(hipHostMalloc(reinterpret_cast<void **>(&h_present_[s]), n * 4, hipHostMallocDefault), "hipHostMalloc(present)");
            check(hipEventCreateWithFlags(&present_converted_[s], hipEventDisableTiming), "hipEventCreate");
            check(hipEventCreateWithFlags(&present_copied_[s], hipEventDisableTiming), "hipEventCreate");
        }
        present_capacity_ = n * 4;
    }
    // three slots: call n converts and copies into slot n % 3 and hands out slot (n - 1) % 3, which call n + 1 leaves alone and
    // call n + 2 overwrites -- the pointer stays valid until the next-but-one call, as include/vr_core.h promises (round-3
    // advisor: with two slots it was the next call)
    const int s = present_count_ % kPresentSlots;
    // the slot's previous copy (three frames ago) must have left its staging buffer before it is overwritten
    if (present_count_ >= kPresentSlots) check(hipStreamWaitEvent(stream(), present_copied_[s], 0), "hipStreamWaitEvent");
    check(launch_to_rgba8(src_frame, d_present_[s], n, stream()), "to_rgba8_kernel");
    check(hipEventRecord(present_converted_[s], stream()), "hipEventRecord");
    check(hipStreamWaitEvent(present_stream_, present_converted_[s], 0), "hipStreamWaitEvent");
    check(hipMemcpyAsync(h_present_[s], d_present_[s], n * 4, hipMemcpyDeviceToHost, present_stream_), "hipMemcpy(D2H present)");
    check(hipEventRecord(present_copied_[s], present_stream_), "hipEventRecord");
    const int ready = present_count_ == 0 ? s : (s + kPresentSlots - 1) % kPresentSlots;   // the previous frame; the first call waits for its own
    present_count_++;
    check(hipEventSynchronize(present_copied_[ready]), "hipEventSynchronize");
    return h_present_[ready];
}

// src/RendererCore.cpp:165-182: RGB8 read-back of the target, written top row first
// (stbi_flip_vertically_on_write).  glReadPixels' float->unorm8 conversion is
// round(clamp(c,0,1)*255).
bool RendererCore::saveImage(std::string fn, std::string ext)
{
    const int w = framebuffer_size[0], h = framebuffer_size[1];
    std::vector<float> rgba((size_t)w * h * 4);
    readPixels(rgba.data(), rgba.size());
    const int stride = w * 3;
    std::vector<uint8_t> rgb((size_t)stride * h);
    for (int y = 0; y < h; y++) {
        const float *src = &rgba[(size_t)(h - 1 - y) * w * 4];
        uint8_t *dst = &rgb[(size_t)y * stride];
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                float v = src[4 * x + c];
                v = v != v ? 0.0f : std::fmin(std::fmax(v, 0.0f), 1.0f);
                dst[3 * x + c] = (uint8_t)std::floor(v * 255.0f + 0.5f);
            }
    }
    if (ext == ".png") return writePNG(fn, w, h, rgb.data(), stride);
    if (ext == ".bmp") return writeBMP(fn, w, h, rgb.data(), stride);
    if (ext == ".ppm") return writePPM(fn, w, h, rgb.data(), stride);
    if (ext == ".jpg") return writeJPEG(fn, w, h, rgb.data(), stride, 100);   // :176-177
    return false;
}

}  // namespace vr
