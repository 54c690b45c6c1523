// renderer_core.h -- host side of the MI355X ray-march path.
//
// Mirrors the reference's RendererCore (include/RendererCore.h:9-44): same member
// and method names, same call semantics, with the OpenGL objects replaced by HIP
// allocations and the compute-shader dispatch replaced by vr::launch_raymarch.
// Members the reference keeps private-but-friend (RendererGUI pokes them directly,
// include/RendererCore.h:18) are public here; vr_capi.cpp is the "friend".
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "camera.h"
#include "vr_frame.h"

namespace vr {

struct HipError : std::runtime_error {
    hipError_t code;
    HipError(hipError_t c, const std::string &what) : std::runtime_error(what), code(c) {}
};
struct NoDeviceError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct IoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class RendererCore {
public:
    explicit RendererCore(int device);      // device < 0: host-only (no HIP calls ever)
    ~RendererCore();
    RendererCore(const RendererCore &) = delete;
    RendererCore &operator=(const RendererCore &) = delete;

    void setup();                            // src/RendererCore.cpp:34-44
    void render();                           // src/RendererCore.cpp:138-163 (timed, blocking)
    void renderAsync();                      // same launch, no events / no host sync
    void synchronize();

    // ---- reference-named members (private+friend in the reference)
    void setAlpha();                         // :56-60
    void setMinVal();                        // :62-71
    void setMaxVal();                        // :73-82
    void setMIP();                           // :84-88
    void setUniforms();                      // :100-110
    void setInitialCameraRotation();         // :90-98
    void setupFBO();                         // :184-219
    void setupUBO(bool is_update = false);   // :221-240
    void readVolumeData(std::string fn);     // :242-447
    bool checkRawInfFile(std::string fn);    // :46-54
    bool saveImage(std::string fn, std::string ext);  // :165-182
    void readPixelsRGBA8(uint8_t *rgba8, size_t n_bytes);   // the own target converted on the device: 4x fewer bytes over PCIe
    // Presentation (replaces the on-GPU blit of src/RendererCore.cpp:158-162 for a display that is not this GPU): the frame
    // in `src_frame` (device RGBA32F, fb_w x fb_h; nullptr = this handle's target) is converted to RGBA8 on the launch
    // stream and copied to one of TWO pinned host buffers on a copy stream; returns the frame enqueued by the PREVIOUS call
    // (complete by now: its copy ran under the kernel that followed it), or this one after a wait on the first call.
    const uint8_t *presentRGBA8(const void *src_frame);
    bool loadShader(std::string fn, bool reload);     // :112-136

    Camera main_cam;
    std::vector<float> histogram;
    std::string loaded_dataset, loaded_shader, msg, title;
    float alpha_scale = 1, kerneltime_sum = 0;
    int workgroups_x = 0, workgroups_y = 0, datasize_bytes = -1, min_val = 0, max_val = 0, max_dataset_val = 0,
        min_dataset_val = 0;
    bool use_mip = false, rotate_to_bottom = false, rotate_to_top = false;
    float voxel_size[3] = {1.0f, 1.0f, 1.0f};
    int tex3D_dim[3] = {0, 0, 0};
    int window_size[2] = {0, 0}, framebuffer_size[2] = {0, 0};

    // ---- additions (no reference equivalent)
    void setVolume(const void *host, int nx, int ny, int nz, int bytes, float sx, float sy, float sz);
    void generateSynthetic(int kind, int nx, int ny, int nz, int bytes, uint32_t param);
    void readVolume(void *host, size_t bytes);
    void setLayout(int layout);
    void setQuirks(uint32_t q);               // workgroups_x/y + window uniforms only (camera, messages untouched)
    void setTransferFunction(const int32_t *iso, const float *rgba4, int n);
    void getTransferLut(float *lut1024) const;
    void countSamples(uint64_t *total, uint32_t *per_pixel, size_t n_pixels);
    void readPixels(float *rgba, size_t n_floats);
    void *framebufferDevice() const { return ext_fb_ ? ext_fb_ : d_fb_; }
    void setStream(hipStream_t s) { user_stream_ = s; }
    hipStream_t streamHandle() const { return stream(); }     // the stream the next launch goes to
    bool greyMode() const { return tf_lut_.empty(); }          // r == g == b in every pixel (no transfer function)
    void prepareForLaunch() { FrameParams P; LaunchConfig L; (void)prepareLaunch(P, L); }   // certification, tile order, packed copy: host work a timed launch should not carry
    void setExternalFramebuffer(void *p) { ext_fb_ = p; }
    void setRowRange(int b, int e) { row_begin_ = b; row_end_ = e; }
    void setRowStripes(int rows, int index, int count);
    void setFramebufferCompact(bool on) { fb_compact_ = on; }
    void setFramebufferFormat(int fmt);      // 0 RGBA32F, 1 (grey, alpha) float2 -- external targets, grey modes
    int localRows() const;   // rows this handle renders (stripe padding included)
    void computeHistogram(float out[256]);
    // best-of-`reps` streaming read of the resident volume; returns GB/s (1e9 bytes per second)
    double measureStreamRead(int reps);
    void assembleShards(const void *gathered, void *frame, int n, int local_rows, int stripe_rows, int channels, void *hip_stream);
    const char *lastKernelName() const { return last_kernel_; }
    // settled entries of the measured work model as a flat blob (vr_export_choices / vr_import_choices): a handle that imports what
    // another handle -- or an earlier process -- measured starts every known configuration on its settled kernel, with no trial frames.
    // A blob is only accepted on the device model and library build it was measured with (returns 0 entries otherwise).
    size_t exportChoices(void *buf, size_t capacity);           // bytes the blob needs; written when capacity suffices
    int importChoices(const void *buf, size_t bytes);           // entries accepted
    int lastLaunchChoice() const { return last_choice_; }     // 1 relay, 2 pipelined loop, 4 short batches, tri_slab << 3, 256: the work model is still exploring this configuration
    size_t lastPacked12Bytes() const { return last_packed12_bytes_; }
    size_t lastApronBytes() const { return last_apron_bytes_; }
    // device memory this handle holds right now: the volume as loaded; the optional speed copies (12-bit packed copy, apron
    // copies); everything else (targets, tables, skip grid, staging)
    void residentBytes(uint64_t &volume, uint64_t &copies, uint64_t &other) const;
    // upper bound on the optional copies' total bytes (vr_set_copy_budget).  kCopyBudgetAuto: allowed while the device keeps
    // kCopyReserveBytes (and a tenth of its memory) free after the allocation.  Lowering it frees copies that no longer fit.
    static constexpr uint64_t kCopyBudgetAuto = ~0ull, kCopyReserveBytes = 1ull << 30;
    void setCopyBudget(uint64_t bytes);
    uint64_t copyBudget() const { return copy_budget_; }
    bool hasDevice() const { return device_ >= 0; }
    void warmTrilinear();                    // pre-load the staged TRILINEAR kernel's code objects (once)

    int filter = 0, accum = 0, skip_empty = 0;
    int layout = 1;          // VR_LAYOUT_BRICKED: the faster HBM layout is the default (vr_set_layout)
    uint32_t quirks = 2u;   // VR_QUIRK_DEFAULT
    int force_generic = 0;
    int pack12 = 1;                          // 1: keep a 12-bit packed copy for the fast kernel when the data allow (vr_set_pack12)
    int tri_apron = 1;                       // 1: keep an apron copy (5x4x4-stored bricks) for the TRILINEAR kernel (vr_set_trilinear_copy)
    int tile_order = 1;      // 1: longest-first tile schedule, 0: arithmetic order
    int autotune = 1;        // 1: kernel variant 0 measures its candidate kernels on the first frames of a configuration and keeps the fastest (vr_set_autotune)
    std::string last_error;

private:
    // the values the kernel sees: the reference's GL uniform state (set by set*())
    struct Uniforms {
        float alpha_scale = 1;
        float voxel_size[3] = {1, 1, 1};
        int min_val = 0, max_val = 0, is_MIP = 0, view_top = 0, view_bottom = 0;
    } u_;
    std::vector<float> cam_block_;           // the "UBO" contents (21 floats)
    bool cs_program_ = false;                // cs_programID != 0

    int device_ = -1;
    hipStream_t own_stream_ = nullptr, user_stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    void *d_vol_ = nullptr;
    size_t vol_alloc_bytes_ = 0;
    // geometry and voxel type of the volume that is RESIDENT (set only after a successful upload).
    // tex3D_dim / datasize_bytes above are GUI inputs of readVolumeData, like the reference's fields;
    // the reference's shader reads textureSize() of the uploaded texture (VolumeRenderer.cs:62), not them.
    int res_dims_[3] = {0, 0, 0};
    int res_bytes_ = 0;
    int vol_layout_ = 0;
    float4 *d_fb_ = nullptr;
    void *ext_fb_ = nullptr;
    float4 *d_tf_ = nullptr;
    uint32_t *d_spp_ = nullptr;
    size_t spp_capacity_ = 0;
    unsigned *d_scratch_ = nullptr;          // 4 + 256 words (+ slack)
    std::vector<float> tf_lut_;              // 256 x RGBA, empty = grey ramp
    bool tf_grey_ = false;                   // every entry of tf_lut_ has r == g == b bit for bit
    int exact_min_ = 0, exact_max_ = 65535;   // exact voxel range of the resident volume
    int row_begin_ = 0, row_end_ = -1;
    int stripe_rows_ = 0, stripe_index_ = 0, stripe_count_ = 1;
    bool fb_compact_ = false;
    int fb_format_ = 0;
    std::map<uint32_t, bool> cert_cache_;    // divisor bits -> certified
    uint16_t *d_skip_grid_ = nullptr;        // dilated cell-max grid (built lazily, dropped with the volume)
    void *d_rgba8_ = nullptr;                // RGBA8 staging of the target (readPixelsRGBA8)
    size_t rgba8_capacity_ = 0;
    static constexpr int kPresentSlots = 3;  // the frame handed out by call n (slot (n - 1) % 3) is written again by call n + 2: valid until the next-but-one call
    void *d_present_[kPresentSlots] = {};    // presentRGBA8: device staging per slot
    uint8_t *h_present_[kPresentSlots] = {}; // ... and pinned host frames
    hipEvent_t present_converted_[kPresentSlots] = {}, present_copied_[kPresentSlots] = {};
    hipStream_t present_stream_ = nullptr;
    size_t present_capacity_ = 0;
    int present_count_ = 0;                  // frames enqueued so far
    void releasePresent();
    void *d_vol12_ = nullptr;                // 12-bit packed copy of the bricked u16 volume (lazily, dropped with the volume)
    size_t vol12_bytes_ = 0;
    bool vol12_failed_ = false;              // allocation of the packed copy failed for this volume: do not retry
    size_t last_packed12_bytes_ = 0;         // packed copy used by the last launch (0 = none)
    int vol12_base_ = 0;                     // the packed copy holds voxel - vol12_base_ (the dataset minimum when it was built)
    void refreshPacked12(FrameParams &P, LaunchConfig &L);
    void *d_apron_ = nullptr;                // TRILINEAR's apron copy of the volume (vr_frame.h: apron_voxels)
    void *d_apron_perm_[2] = {nullptr, nullptr};   // the same with the bricks' planes along y / x slowest (half layers of the staged kernel; 16-bit volumes, oblique views)
    bool apron_perm_failed_ = false;
    size_t apron_bytes_ = 0, last_apron_bytes_ = 0;
    bool apron_failed_ = false;
    void refreshApron(const FrameParams &P, LaunchConfig &L);
    uint64_t copy_budget_ = kCopyBudgetAuto;
    uint64_t copiesBytes() const;            // optional copies resident now
    bool copyFits(uint64_t bytes) const;     // may another optional copy of `bytes` be allocated (budget / free device memory)?
    size_t skip_grid_cells_ = 0;
    unsigned skip_grid_min_ = 0;             // smallest cell value of the grid (read back once when it is built): a threshold below it skips nothing and the launch keeps the instances without skipping
    void refreshSkipGrid(FrameParams &P, LaunchConfig &L);
    uint32_t *d_tile_table_ = nullptr;       // work-ordered block -> tile table (tile_schedule.h)
    size_t tile_table_capacity_ = 0, tile_table_blocks_ = 0;
    uint32_t *d_tile_table_tall_ = nullptr;  // the same for 16x32-pixel tiles (the staged trilinear kernel's tall shape); built with the table above
    size_t tile_table_tall_capacity_ = 0, tile_table_tall_blocks_ = 0;
    uint32_t *d_tile_table_small_ = nullptr; // the same for 16x16-pixel tiles (the staged trilinear kernel's four-wavefront shape)
    size_t tile_table_small_capacity_ = 0, tile_table_small_blocks_ = 0;
    unsigned tile_active_small_ = 0;
    uint64_t tile_table_key_ = 0;
    int64_t tile_table_skip_sig_ = 0;        // what the order was built for: 0 = geometric ray lengths, threshold + 1 = visible work under empty-space skipping (-1: rebuild)
    float *d_tile_work_ = nullptr;           // per-tile cost estimates of that (scratch of refreshTileSchedule)
    size_t tile_work_capacity_ = 0;
    float tile_table_cam_[21] = {};          // camera block the cached order was built for
    unsigned tile_active_ = 0;               // tiles with work in the cached schedule
    double tile_longest_ = 0.0;              // expected samples of its longest ray
    void refreshTileSchedule(const FrameParams &P, LaunchConfig &L);
    const char *last_kernel_ = "";
    bool tslab_warm_ = false;
    // ---- measured work model (kernel variant 0): every candidate kernel of a configuration renders the same bits, so
    // the first frames of a configuration double as measurements -- each candidate is timed a few times (HIP events on
    // the launch stream, polled without blocking), the fastest is kept until the configuration changes.
    // A candidate = bit set: 1 relay kernel, 2 pipelined batch loop, 4 four-sample batches; bits 3 .. 6: LaunchConfig::tri_slab
    // (0 = batched trilinear kernel, 1 / 3 / 4 = the LDS-staged kernel in one of its shapes).
    // An entry settles on the fastest candidate after kTuneTries measurements of each (in an order shuffled per entry, so no
    // candidate is always the one measured on cold clocks), is measured ONCE more kTuneRevalidateFrames frames later (the first
    // ~25 frames after idle run ~15 % slow while the clocks ramp) and is evicted least-recently-used first.
    static constexpr int kTuneCand = 8, kTuneTries = 2, kTuneRevalidateFrames = 96, kTuneEntries = 512;
    struct TuneEntry {
        int ncand = 0, cand[kTuneCand] = {}, tries[kTuneCand] = {}, issued[kTuneCand] = {}, settled = -1, next = 0;   // issued: measurements launched (a burst of asynchronous frames launches kTuneTries per candidate, not one per frame until the results arrive)
        int heur = 0;                        // the heuristic's choice (a candidate value): it wins ties
        float best_ms[kTuneCand] = {};
        int frames_settled = 0;              // launches since it settled
        bool revalidated = false;
        uint64_t last_use = 0;               // launch counter at its last use (eviction order)
        uint64_t gen = 0;                    // generation: results of measurements launched for an earlier life of this key (evicted and re-created, or before a re-validation) are dropped
    };
    std::map<uint64_t, TuneEntry> tune_;
    static constexpr int kTuneSlots = 12;    // asynchronous measurements that may be in flight (a burst of renderAsync calls)
    struct TuneSlot { hipEvent_t ev0 = nullptr, ev1 = nullptr; uint64_t key = 0, gen = 0; int cand = -1; };   // cand: the candidate VALUE (bit set), not its index in the entry's shuffled order
    TuneSlot tune_slot_[kTuneSlots];
    int tune_head_ = 0, tune_count_ = 0;     // ring of in-flight slots: oldest at tune_head_
    bool tune_measure_ = false;              // the launch being prepared is a measurement of candidate tune_cand_ of entry tune_key_
    uint64_t tune_key_ = 0;
    int tune_cand_ = -1;                     // candidate VALUE being measured
    uint64_t tune_gen_ = 0, tune_gen_counter_ = 0;
    uint64_t tune_clock_ = 0, tune_last_key_ = 0;   // launches seen; the previous launch's key and for how many launches in a row
    int tune_same_key_run_ = 0;
    int last_choice_ = 0;                    // candidate bits of the last launch (vr_get_launch_choice)
    void tuneChoose(const FrameParams &P, LaunchConfig &L);
    void tuneRecord(uint64_t key, uint64_t gen, int cand, float ms);
    void tuneIssued(uint64_t key, uint64_t gen, int cand);   // a measurement of `cand` was really launched (events recorded)
    void tuneCollect();

    hipStream_t stream() const { return user_stream_ ? user_stream_ : own_stream_; }
    void requireDevice(const char *what) const;
    void check(hipError_t e, const char *what) const;
    void freeVolume();
    void allocVolume(int nx, int ny, int nz, int bytes, int layout);
    size_t storageVoxels(int nx, int ny, int nz, int layout) const;
    void afterVolumeLoaded(const std::string &name);
    void scanDatasetRange();
    bool certifyDivisor(float b);
    void buildFrame(FrameParams &P, LaunchConfig &L);
    float4 *prepareLaunch(FrameParams &P, LaunchConfig &L);
    void launch(uint32_t *spp);
    void updateWorkgroups();
    void setMessage(const std::string &t, const std::string &m) { title = t; msg = m; }
};

}  // namespace vr
