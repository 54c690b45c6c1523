// vr_group.cpp -- native multi-GPU rendering behind the C ABI (include/vr_core.h: vr_group_*).
//
// One process, one RendererCore per device ("single process + ncclCommInitAll over
// 8 devices; no launcher needed"): the volume is replicated, the image is split by rows (cyclic
// stripes or contiguous blocks), every device renders its shard into a compact local target on
// its own stream, the shards travel to the root device and one small kernel there undoes the
// interleave.  It shards the single dispatch of the reference,
// glDispatchCompute(workgroups_x, workgroups_y, 1) (src/RendererCore.cpp:149-151), so that a
// caller shaped like RendererGUI::run (src/RendererGUI.cpp:100-101) can use every GPU of a node
// by calling vr_group_render() where it called render().
//
// Transport: RCCL over xGMI when the group's devices are distinct -- a gather to the root as
// grouped ncclSend/ncclRecv (the root's point-to-point links to its 7 peers work concurrently;
// an all-gather would move 8x the bytes nobody asked for).  librccl is loaded on first use
// (dlopen), so single-GPU users never load it.  When two members share a device (validation on
// a one-GPU box) or RCCL is unavailable, the shards move with hipMemcpyPeerAsync on the root's
// stream behind per-member events -- same buffers, same assembly.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/vr_core.h"
#include "renderer_core.h"
#include "vr_handle.h"
#include "vr_kernels.h"


namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err)
    {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            err = "librccl.so lacks a required entry point";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
};

RcclApi g_rccl;

}  // namespace

struct vr_group {
    std::vector<std::unique_ptr<vr_renderer>> members;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;            // empty: copy transport
    std::vector<void *> local;                 // compact shard target per member (on its device)
    std::vector<hipEvent_t> rendered, t0, t1;  // per member, on its device
    void *gathered = nullptr;                  // root: n * local_rows * fb_w * C floats, rank-major
    float4 *frame = nullptr;                   // root: assembled fb_w x fb_h RGBA32F
    int fb_w = 0, fb_h = 0, local_rows = 0, stripe_rows = 16, partition = 0, channels = 4;
    int want_rccl = 1;                         // 0 never, 1 when the devices are distinct, 2 also for a one-member group (probe)
    float kerneltime_sum = 0.0f;
    std::string last_error, transport = "none";

    ~vr_group() { release(); for (ncclComm_t c : comms) if (c) g_rccl.CommDestroy(c); }
    void release()
    {
        for (size_t r = 0; r < members.size(); r++) {
            (void)hipSetDevice(devices[r]);
            if (r < local.size() && local[r]) (void)hipFree(local[r]);
            if (r < rendered.size() && rendered[r]) (void)hipEventDestroy(rendered[r]);
            if (r < t0.size() && t0[r]) (void)hipEventDestroy(t0[r]);
            if (r < t1.size() && t1[r]) (void)hipEventDestroy(t1[r]);
        }
        local.clear(); rendered.clear(); t0.clear(); t1.clear();
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (gathered) { (void)hipFree(gathered); gathered = nullptr; }
        if (frame) { (void)hipFree(frame); frame = nullptr; }
    }
};

namespace {

int gfail(vr_group_handle g, int code, const std::string &what)
{
    if (g) g->last_error = what;
    return code;
}

#define VRG_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return gfail(g, VR_E_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

int vr_group_create(vr_group_handle *out, const int *devices, int n)
{
    if (!out || !devices || n < 1 || n > 64) return VR_E_INVALID;
    *out = nullptr;
    std::unique_ptr<vr_group> g(new vr_group);
    try {
        for (int r = 0; r < n; r++) {
            if (devices[r] < 0) return VR_E_INVALID;
            g->members.emplace_back(new vr_renderer(devices[r]));
            g->devices.push_back(devices[r]);
        }
    } catch (const vr::NoDeviceError &) {
        return VR_E_NO_DEVICE;
    } catch (const std::exception &) {
        return VR_E_HIP;
    }
    *out = g.release();
    return VR_OK;
}

void vr_group_destroy(vr_group_handle g) { delete g; }

int vr_group_size(vr_group_handle g) { return g ? (int)g->members.size() : 0; }

vr_handle vr_group_member(vr_group_handle g, int rank)
{
    if (!g || rank < 0 || rank >= (int)g->members.size()) return nullptr;
    return g->members[(size_t)rank].get();
}

const char *vr_group_last_error(vr_group_handle g) { return g ? g->last_error.c_str() : "null group"; }
const char *vr_group_transport(vr_group_handle g) { return g ? g->transport.c_str() : ""; }

int vr_group_set_transport(vr_group_handle g, int use_rccl)
{
    if (!g) return VR_E_INVALID;
    if (use_rccl < 0 || use_rccl > 2) return gfail(g, VR_E_INVALID, "vr_group_set_transport: 0, 1 or 2");
    g->want_rccl = use_rccl;
    return VR_OK;
}

int vr_group_setup(vr_group_handle g, int win_w, int win_h, int fb_w, int fb_h, int partition, int stripe_rows)
{
    if (!g) return VR_E_INVALID;
    if (win_w <= 0 || win_h <= 0 || fb_w <= 0 || fb_h <= 0 || (partition != 0 && partition != 1) || stripe_rows < 1)
        return gfail(g, VR_E_INVALID, "vr_group_setup: bad sizes / partition");
    const int n = (int)g->members.size();
    g->release();
    g->fb_w = fb_w; g->fb_h = fb_h; g->partition = partition; g->stripe_rows = stripe_rows;
    if (n == 1) g->local_rows = fb_h;
    else if (partition == 1) g->local_rows = (fb_h + n - 1) / n;
    else g->local_rows = ((fb_h + stripe_rows - 1) / stripe_rows + n - 1) / n * stripe_rows;
    g->local.assign((size_t)n, nullptr);
    g->rendered.assign((size_t)n, nullptr); g->t0.assign((size_t)n, nullptr); g->t1.assign((size_t)n, nullptr);
    const size_t shard_bytes = (size_t)g->local_rows * (size_t)fb_w * sizeof(float4);   // sized for RGBA; (grey, alpha) uses half
    for (int r = 0; r < n; r++) {
        vr_handle h = g->members[(size_t)r].get();
        const int rc = vr_setup(h, win_w, win_h, fb_w, fb_h);
        if (rc != VR_OK) return gfail(g, rc, std::string("member setup: ") + vr_last_error(h));
        VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
        VRG_HIP(hipMalloc(&g->local[(size_t)r], shard_bytes));
        VRG_HIP(hipMemset(g->local[(size_t)r], 0, shard_bytes));
        VRG_HIP(hipEventCreateWithFlags(&g->rendered[(size_t)r], hipEventDisableTiming));
        VRG_HIP(hipEventCreate(&g->t0[(size_t)r]));
        VRG_HIP(hipEventCreate(&g->t1[(size_t)r]));
        vr::RendererCore &c = h->core;
        c.setExternalFramebuffer(g->local[(size_t)r]);
        c.setFramebufferCompact(true);
        if (n == 1) { c.setRowRange(0, -1); c.setRowStripes(1, 0, 1); }
        else if (partition == 1) {
            c.setRowStripes(1, 0, 1);
            const int b = std::min(r * g->local_rows, fb_h);
            c.setRowRange(b, std::min(b + g->local_rows, fb_h));
        } else {
            c.setRowRange(0, -1);
            c.setRowStripes(stripe_rows, r, n);
        }
    }
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(hipMalloc(&g->gathered, shard_bytes * (size_t)n));
    VRG_HIP(hipMalloc(reinterpret_cast<void **>(&g->frame), (size_t)fb_w * (size_t)fb_h * sizeof(float4)));
    VRG_HIP(hipMemset(g->frame, 0, (size_t)fb_w * (size_t)fb_h * sizeof(float4)));
    // transport: RCCL when every member has its own device
    for (ncclComm_t c : g->comms) if (c) g_rccl.CommDestroy(c);
    g->comms.clear();
    g->transport = n == 1 ? "none (one member)" : "hipMemcpyPeerAsync";
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == g->devices.size();
    if ((n > 1 && distinct && g->want_rccl) || (n == 1 && g->want_rccl == 2)) {
        std::string err;
        if (g_rccl.load(err)) {
            g->comms.assign((size_t)n, nullptr);
            const ncclResult_t rc = g_rccl.CommInitAll(g->comms.data(), n, g->devices.data());
            if (rc == ncclSuccess) g->transport = n == 1 ? "rccl (one member: communicator only)" : "rccl send/recv gather";
            else { g->comms.clear(); g->last_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc) + " (using peer copies)"; }
        } else {
            g->last_error = err + " (using peer copies)";
        }
    }
    if (g->comms.empty() && n > 1 && distinct) {
        for (int r = 1; r < n; r++) {                                   // peer copies want peer access where the topology allows it
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->devices[0], g->devices[(size_t)r]) == hipSuccess && can) {
                (void)hipSetDevice(g->devices[0]);
                (void)hipDeviceEnablePeerAccess(g->devices[(size_t)r], 0);
                (void)hipGetLastError();
            }
        }
    }
    return VR_OK;
}

int vr_group_render(vr_group_handle g)
{
    if (!g) return VR_E_INVALID;
    const int n = (int)g->members.size();
    if (!g->frame) return gfail(g, VR_E_INVALID, "vr_group_render: call vr_group_setup first");
    // (grey, alpha) shards when every member renders a grey mode: half the bytes on the wire
    bool grey = true;
    for (auto &m : g->members) grey = grey && m->core.greyMode();
    g->channels = grey ? 2 : 4;
    const size_t shard_floats = (size_t)g->local_rows * (size_t)g->fb_w * (size_t)g->channels;
    try {
        for (int r = 0; r < n; r++) {
            vr::RendererCore &c = g->members[(size_t)r]->core;
            c.setFramebufferFormat(grey ? 1 : 0);
            c.prepareForLaunch();                                       // host-side preparation outside the timed region
        }
        for (int r = 0; r < n; r++) {
            vr::RendererCore &c = g->members[(size_t)r]->core;
            VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
            VRG_HIP(hipEventRecord(g->t0[(size_t)r], c.streamHandle()));
            c.renderAsync();
            VRG_HIP(hipEventRecord(g->t1[(size_t)r], c.streamHandle()));
            VRG_HIP(hipEventRecord(g->rendered[(size_t)r], c.streamHandle()));
        }
    } catch (const vr::NoDeviceError &e) {
        return gfail(g, VR_E_NO_DEVICE, e.what());
    } catch (const vr::HipError &e) {
        return gfail(g, VR_E_HIP, e.what());
    } catch (const std::exception &e) {
        return gfail(g, VR_E_INVALID, e.what());
    }
    hipStream_t root = g->members[0]->core.streamHandle();
    char *dst = static_cast<char *>(g->gathered);
    const size_t shard_bytes = shard_floats * sizeof(float);
    if (!g->comms.empty()) {
        ncclResult_t rc = g_rccl.GroupStart();
        for (int r = 1; r < n && rc == ncclSuccess; r++) {
            rc = g_rccl.Recv(dst + (size_t)r * shard_bytes, shard_floats, ncclFloat, r, g->comms[0], root);
            if (rc == ncclSuccess)
                rc = g_rccl.Send(g->local[(size_t)r], shard_floats, ncclFloat, 0, g->comms[(size_t)r], g->members[(size_t)r]->core.streamHandle());
        }
        const ncclResult_t rc2 = g_rccl.GroupEnd();
        if (rc != ncclSuccess || rc2 != ncclSuccess)
            return gfail(g, VR_E_HIP, std::string("rccl gather: ") + g_rccl.GetErrorString(rc != ncclSuccess ? rc : rc2));
        VRG_HIP(hipSetDevice(g->devices[0]));
        VRG_HIP(hipMemcpyAsync(dst, g->local[0], shard_bytes, hipMemcpyDeviceToDevice, root));
    } else {
        VRG_HIP(hipSetDevice(g->devices[0]));
        for (int r = 0; r < n; r++) {
            if (r > 0) VRG_HIP(hipStreamWaitEvent(root, g->rendered[(size_t)r], 0));
            if (g->devices[(size_t)r] == g->devices[0])
                VRG_HIP(hipMemcpyAsync(dst + (size_t)r * shard_bytes, g->local[(size_t)r], shard_bytes, hipMemcpyDeviceToDevice, root));
            else
                VRG_HIP(hipMemcpyPeerAsync(dst + (size_t)r * shard_bytes, g->devices[0], g->local[(size_t)r], g->devices[(size_t)r], shard_bytes, root));
        }
    }
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(vr::launch_assemble(g->gathered, g->frame, g->fb_w, g->fb_h, n, g->local_rows, g->partition == 1 || n == 1 ? 0 : g->stripe_rows,
                                g->channels, root));
    VRG_HIP(hipStreamSynchronize(root));
    float worst = 0.0f;
    for (int r = 0; r < n; r++) {
        VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
        VRG_HIP(hipEventSynchronize(g->t1[(size_t)r]));
        float ms = 0.0f;
        VRG_HIP(hipEventElapsedTime(&ms, g->t0[(size_t)r], g->t1[(size_t)r]));
        worst = std::max(worst, ms);
    }
    g->kerneltime_sum += worst;
    VRG_HIP(hipSetDevice(g->devices[0]));
    return VR_OK;
}

float vr_group_kernel_ms_take(vr_group_handle g)
{
    if (!g) return 0.0f;
    const float v = g->kerneltime_sum;
    g->kerneltime_sum = 0.0f;
    return v;
}

void *vr_group_framebuffer_device(vr_group_handle g) { return g ? g->frame : nullptr; }

int vr_group_read_pixels(vr_group_handle g, float *rgba, size_t n_floats)
{
    if (!g) return VR_E_INVALID;
    const size_t need = (size_t)g->fb_w * (size_t)g->fb_h * 4u;
    if (!g->frame || !rgba || n_floats < need) return gfail(g, VR_E_INVALID, "vr_group_read_pixels: no frame / buffer too small");
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(hipMemcpy(rgba, g->frame, need * sizeof(float), hipMemcpyDeviceToHost));
    return VR_OK;
}

}  // extern "C"
