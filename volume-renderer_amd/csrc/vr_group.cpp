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
// (dlopen) and its handful of types is declared here, so neither single-GPU users nor a ROCm
// install without the RCCL development headers need it.  When two members share a device
// (validation on a one-GPU box) or RCCL is unavailable, the shards move with hipMemcpyPeerAsync
// behind per-member events -- same buffers, same assembly.
//
// Pipeline: THREE frame slots, at most two frames in flight.  vr_group_render_async() enqueues frame i -- shard kernels
// on the members' render streams into slot i % 3, the gather and the assembly on separate transfer streams -- and
// returns; vr_group_wait() blocks until the oldest frame in flight is assembled.  With one frame always in flight the
// gather + assembly of frame i overlap the shard kernels of frame i + 1 (what bench.py's torch.distributed path does with
// its two slots); the third slot keeps the last COMPLETED frame (the one vr_group_framebuffer_device hands out) from
// being the target of the frame issued next.  vr_group_render() = render_async + wait: blocking, like the reference's
// render().  A frame whose issue fails anywhere is drained (every stream it touched) and counts as never issued.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

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

// the part of RCCL's C interface this file uses (rccl.h: ncclComm_t is an opaque pointer, ncclResult_t /
// ncclDataType_t are plain enums; ncclSuccess = 0, ncclFloat32 = 7 in every NCCL 2.x / RCCL release)
typedef struct ncclComm *rccl_comm_t;
typedef int rccl_result_t;
typedef int rccl_dtype_t;
constexpr rccl_result_t kRcclSuccess = 0;
constexpr rccl_dtype_t kRcclFloat32 = 7;

struct RcclApi {
    void *lib = nullptr;
    rccl_result_t (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
    rccl_result_t (*CommDestroy)(rccl_comm_t) = nullptr;
    rccl_result_t (*GroupStart)() = nullptr;
    rccl_result_t (*GroupEnd)() = nullptr;
    rccl_result_t (*Send)(const void *, size_t, rccl_dtype_t, int, rccl_comm_t, hipStream_t) = nullptr;
    rccl_result_t (*Recv)(void *, size_t, rccl_dtype_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(rccl_result_t) = nullptr;
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

// Three frame slots, at most two frames in flight: the slot of the last COMPLETED frame -- the one vr_group_framebuffer_device /
// read_pixels / present hand out -- is then never the target of a frame that may be issued next (round-3 advisor: with two
// slots the second vr_group_render_async after a vr_group_wait assembled into the frame a consumer was still reading).
constexpr int kSlots = 3, kInFlight = 2;

}  // namespace

struct vr_group {
    std::vector<std::unique_ptr<vr_renderer>> members;
    std::vector<int> devices;
    std::vector<rccl_comm_t> comms;            // empty: copy transport
    // per member, on its device: compact shard targets (one per frame slot), a transfer stream (the send side of
    // the gather: the render stream is free for the next frame's kernel), events
    std::vector<void *> local[kSlots];
    std::vector<hipStream_t> xfer;
    std::vector<hipEvent_t> rendered[kSlots], sent[kSlots], t0[kSlots], t1[kSlots];
    // root device: rank-major gathered shards and the assembled frame per slot, the gather/assembly stream
    void *gathered[kSlots] = {};
    float4 *frame[kSlots] = {};
    hipStream_t gather_stream = nullptr;
    hipEvent_t assembled[kSlots] = {};
    bool slot_used[kSlots] = {};   // the slot's events have been recorded at least once
    int issued = 0, completed = 0;             // frames enqueued / waited for; in flight = issued - completed <= kInFlight
    int current = -1;                          // slot of the last completed frame (vr_group_framebuffer_device)
    int channels_of[kSlots] = {4, 4, 4};
    int fb_w = 0, fb_h = 0, local_rows = 0, stripe_rows = 16, partition = 0;
    int want_rccl = 1;                         // 0 never, 1 when the devices are distinct, 2 also for a one-member group (probe: self send/recv)
    float kerneltime_sum = 0.0f;
    std::string last_error, transport = "none";

    ~vr_group() { release(); for (rccl_comm_t c : comms) if (c) g_rccl.CommDestroy(c); }
    // drop everything vr_group_setup allocated and detach the members from it: a member handle rendered directly
    // afterwards (or after a failed setup) writes to its own target again, never to freed memory
    void release()
    {
        for (size_t r = 0; r < members.size(); r++) {
            (void)hipSetDevice(devices[r]);
            vr::RendererCore &c = members[r]->core;
            if (c.hasDevice()) (void)hipStreamSynchronize(c.streamHandle());
            if (r < xfer.size() && xfer[r]) { (void)hipStreamSynchronize(xfer[r]); (void)hipStreamDestroy(xfer[r]); }
            for (int s = 0; s < kSlots; s++) {
                if (r < local[s].size() && local[s][r]) (void)hipFree(local[s][r]);
                for (auto *v : {&rendered[s], &sent[s], &t0[s], &t1[s]})
                    if (r < v->size() && (*v)[r]) (void)hipEventDestroy((*v)[r]);
            }
            c.setExternalFramebuffer(nullptr);
            c.setFramebufferCompact(false);
            c.setFramebufferFormat(0);
            c.setRowRange(0, -1);
            c.setRowStripes(1, 0, 1);
        }
        xfer.clear();
        for (int s = 0; s < kSlots; s++) { local[s].clear(); rendered[s].clear(); sent[s].clear(); t0[s].clear(); t1[s].clear(); }
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (gather_stream) { (void)hipStreamSynchronize(gather_stream); (void)hipStreamDestroy(gather_stream); gather_stream = nullptr; }
        for (int s = 0; s < kSlots; s++) {
            if (gathered[s]) { (void)hipFree(gathered[s]); gathered[s] = nullptr; }
            if (frame[s]) { (void)hipFree(frame[s]); frame[s] = nullptr; }
            if (assembled[s]) { (void)hipEventDestroy(assembled[s]); assembled[s] = nullptr; }
            slot_used[s] = false;
        }
        (void)hipGetLastError();
        issued = completed = 0;
        current = -1;
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

int group_setup(vr_group_handle g, int win_w, int win_h, int fb_w, int fb_h, int partition, int stripe_rows)
{
    const int n = (int)g->members.size();
    g->fb_w = fb_w; g->fb_h = fb_h; g->partition = partition; g->stripe_rows = stripe_rows;
    if (n == 1) g->local_rows = fb_h;
    else if (partition == 1) g->local_rows = (fb_h + n - 1) / n;
    else g->local_rows = ((fb_h + stripe_rows - 1) / stripe_rows + n - 1) / n * stripe_rows;
    g->xfer.assign((size_t)n, nullptr);
    for (int s = 0; s < kSlots; s++) {
        g->local[s].assign((size_t)n, nullptr);
        g->rendered[s].assign((size_t)n, nullptr); g->sent[s].assign((size_t)n, nullptr);
        g->t0[s].assign((size_t)n, nullptr); g->t1[s].assign((size_t)n, nullptr);
    }
    const size_t shard_bytes = (size_t)g->local_rows * (size_t)fb_w * sizeof(float4);   // sized for RGBA; (grey, alpha) uses half
    for (int r = 0; r < n; r++) {
        vr_handle h = g->members[(size_t)r].get();
        const int rc = vr_setup(h, win_w, win_h, fb_w, fb_h);
        if (rc != VR_OK) return gfail(g, rc, std::string("member setup: ") + vr_last_error(h));
        VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
        VRG_HIP(hipStreamCreateWithFlags(&g->xfer[(size_t)r], hipStreamNonBlocking));
        for (int s = 0; s < kSlots; s++) {
            VRG_HIP(hipMalloc(&g->local[s][(size_t)r], shard_bytes));
            VRG_HIP(hipMemset(g->local[s][(size_t)r], 0, shard_bytes));
            VRG_HIP(hipEventCreateWithFlags(&g->rendered[s][(size_t)r], hipEventDisableTiming));
            VRG_HIP(hipEventCreateWithFlags(&g->sent[s][(size_t)r], hipEventDisableTiming));
            VRG_HIP(hipEventCreate(&g->t0[s][(size_t)r]));
            VRG_HIP(hipEventCreate(&g->t1[s][(size_t)r]));
        }
        vr::RendererCore &c = h->core;
        c.setExternalFramebuffer(g->local[0][(size_t)r]);
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
    VRG_HIP(hipStreamCreateWithFlags(&g->gather_stream, hipStreamNonBlocking));
    for (int s = 0; s < kSlots; s++) {
        VRG_HIP(hipMalloc(&g->gathered[s], shard_bytes * (size_t)n));
        VRG_HIP(hipMalloc(reinterpret_cast<void **>(&g->frame[s]), (size_t)fb_w * (size_t)fb_h * sizeof(float4)));
        VRG_HIP(hipMemset(g->frame[s], 0, (size_t)fb_w * (size_t)fb_h * sizeof(float4)));
        VRG_HIP(hipEventCreateWithFlags(&g->assembled[s], hipEventDisableTiming));
    }
    VRG_HIP(hipDeviceSynchronize());
    // transport: RCCL when every member has its own device
    for (rccl_comm_t c : g->comms) if (c) g_rccl.CommDestroy(c);
    g->comms.clear();
    g->transport = n == 1 ? "none (one member)" : "hipMemcpyPeerAsync";
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == g->devices.size();
    if ((n > 1 && distinct && g->want_rccl) || (n == 1 && g->want_rccl == 2)) {
        std::string err;
        if (g_rccl.load(err)) {
            g->comms.assign((size_t)n, nullptr);
            const rccl_result_t rc = g_rccl.CommInitAll(g->comms.data(), n, g->devices.data());
            if (rc == kRcclSuccess) g->transport = n == 1 ? "rccl (one member: self send/recv)" : "rccl send/recv gather";
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
    g->release();
    int rc;
    try {
        rc = group_setup(g, win_w, win_h, fb_w, fb_h, partition, stripe_rows);
    } catch (const std::exception &e) {
        rc = gfail(g, VR_E_INVALID, e.what());
    }
    if (rc != VR_OK) {                                                  // nothing half-built stays behind: vr_group_render refuses to run
        const std::string why = g->last_error;
        g->release();
        g->last_error = why;
    }
    return rc;
}

// A frame that failed half-way through its issue (some members launched, others not; a transfer or the assembly refused):
// nothing of it may still be running when the caller issues the next one into the same slot, and the frame counts as never
// issued.  ONE cleanup for every failure exit of the issue (round-3 / round-4 advisor): the members' render streams, their
// transfer streams and the root's gather stream are all drained.
static void drain_group(vr_group_handle g)
{
    for (size_t r = 0; r < g->members.size(); r++) {
        (void)hipSetDevice(g->devices[r]);
        vr::RendererCore &c = g->members[r]->core;
        if (c.hasDevice()) (void)hipStreamSynchronize(c.streamHandle());
        if (r < g->xfer.size() && g->xfer[r]) (void)hipStreamSynchronize(g->xfer[r]);
    }
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
    if (g->gather_stream) (void)hipStreamSynchronize(g->gather_stream);
    (void)hipGetLastError();
}

static int group_issue_frame(vr_group_handle g, bool &touched);

int vr_group_render_async(vr_group_handle g)
{
    if (!g) return VR_E_INVALID;
    bool touched = false;                                              // false while only arguments / state were checked
    const int rc = group_issue_frame(g, touched);
    if (rc != VR_OK && touched) {
        const std::string why = g->last_error;
        drain_group(g);
        g->last_error = why;
    }
    return rc;
}

static int group_issue_frame(vr_group_handle g, bool &touched)
{
    const int n = (int)g->members.size();
    if (!g->frame[0]) return gfail(g, VR_E_INVALID, "vr_group_render: call vr_group_setup first");
    if (g->issued - g->completed >= kInFlight) return gfail(g, VR_E_INVALID, "vr_group_render_async: two frames in flight already (call vr_group_wait)");
    const int s = g->issued % kSlots;
    // (grey, alpha) shards when every member renders a grey mode: half the bytes on the wire
    bool grey = true;
    for (auto &m : g->members) grey = grey && m->core.greyMode();
    const int channels = grey ? 2 : 4;
    g->channels_of[s] = channels;
    const size_t shard_floats = (size_t)g->local_rows * (size_t)g->fb_w * (size_t)channels;
    const size_t shard_bytes = shard_floats * sizeof(float);
    const bool rccl = !g->comms.empty();
    touched = true;
    try {
        for (int r = 0; r < n; r++) {
            vr::RendererCore &c = g->members[(size_t)r]->core;
            c.setExternalFramebuffer(g->local[s][(size_t)r]);
            c.setFramebufferFormat(grey ? 1 : 0);
            c.prepareForLaunch();                                       // host-side preparation outside the timed region
        }
        for (int r = 0; r < n; r++) {
            vr::RendererCore &c = g->members[(size_t)r]->core;
            VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
            // the slot's shard target is free again once frame i - 2 has left it
            if (g->slot_used[s]) VRG_HIP(hipStreamWaitEvent(c.streamHandle(), g->sent[s][(size_t)r], 0));
            VRG_HIP(hipEventRecord(g->t0[s][(size_t)r], c.streamHandle()));
            c.renderAsync();
            VRG_HIP(hipEventRecord(g->t1[s][(size_t)r], c.streamHandle()));
            VRG_HIP(hipEventRecord(g->rendered[s][(size_t)r], c.streamHandle()));
        }
    } catch (const vr::NoDeviceError &e) {
        return gfail(g, VR_E_NO_DEVICE, e.what());
    } catch (const vr::HipError &e) {
        return gfail(g, VR_E_HIP, e.what());
    } catch (const std::exception &e) {
        return gfail(g, VR_E_INVALID, e.what());
    }
    hipStream_t gs = g->gather_stream;
    char *dst = static_cast<char *>(g->gathered[s]);
    if (rccl) {
        // every member's send goes on its transfer stream behind its kernel; the root's receives on the gather stream.
        // Every communicator sees ONE stream: in probe mode the root's send to itself is issued on the gather stream
        // like its receives (RCCL 2.26 does not order a self send/recv that straddles two streams of one communicator
        // with the later work of the receiving stream -- measured here: the assembly ran before the bytes arrived).
        const bool self = n == 1 || g->want_rccl == 2;                  // the root's own shard through ncclSend/ncclRecv too
        for (int r = 1; r < n; r++) {
            VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
            VRG_HIP(hipStreamWaitEvent(g->xfer[(size_t)r], g->rendered[s][(size_t)r], 0));
        }
        VRG_HIP(hipSetDevice(g->devices[0]));
        VRG_HIP(hipStreamWaitEvent(gs, g->rendered[s][0], 0));
        rccl_result_t rc = g_rccl.GroupStart();
        for (int r = self ? 0 : 1; r < n && rc == kRcclSuccess; r++) {
            rc = g_rccl.Recv(dst + (size_t)r * shard_bytes, shard_floats, kRcclFloat32, r, g->comms[0], gs);
            if (rc == kRcclSuccess)
                rc = g_rccl.Send(g->local[s][(size_t)r], shard_floats, kRcclFloat32, 0, g->comms[(size_t)r], r == 0 ? gs : g->xfer[(size_t)r]);
        }
        const rccl_result_t rc2 = g_rccl.GroupEnd();
        if (rc != kRcclSuccess || rc2 != kRcclSuccess)
            return gfail(g, VR_E_HIP, std::string("rccl gather: ") + g_rccl.GetErrorString(rc != kRcclSuccess ? rc : rc2));
        for (int r = 1; r < n; r++) {
            VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
            VRG_HIP(hipEventRecord(g->sent[s][(size_t)r], g->xfer[(size_t)r]));
        }
        VRG_HIP(hipSetDevice(g->devices[0]));
        if (!self)                                                      // the root's own shard: a plain copy on its device
            VRG_HIP(hipMemcpyAsync(dst, g->local[s][0], shard_bytes, hipMemcpyDeviceToDevice, gs));
        VRG_HIP(hipEventRecord(g->sent[s][0], gs));
    } else {
        VRG_HIP(hipSetDevice(g->devices[0]));
        for (int r = 0; r < n; r++) {
            VRG_HIP(hipStreamWaitEvent(gs, g->rendered[s][(size_t)r], 0));
            if (g->devices[(size_t)r] == g->devices[0])
                VRG_HIP(hipMemcpyAsync(dst + (size_t)r * shard_bytes, g->local[s][(size_t)r], shard_bytes, hipMemcpyDeviceToDevice, gs));
            else
                VRG_HIP(hipMemcpyPeerAsync(dst + (size_t)r * shard_bytes, g->devices[0], g->local[s][(size_t)r], g->devices[(size_t)r], shard_bytes, gs));
        }
    }
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(vr::launch_assemble(g->gathered[s], g->frame[s], g->fb_w, g->fb_h, n, g->local_rows, g->partition == 1 || n == 1 ? 0 : g->stripe_rows,
                                channels, gs));
    VRG_HIP(hipEventRecord(g->assembled[s], gs));
    if (!rccl) {
        // the copies were issued by the root: a member's shard target is free when the frame is assembled
        for (int r = 0; r < n; r++) {
            VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
            VRG_HIP(hipStreamWaitEvent(g->xfer[(size_t)r], g->assembled[s], 0));
            VRG_HIP(hipEventRecord(g->sent[s][(size_t)r], g->xfer[(size_t)r]));
        }
        VRG_HIP(hipSetDevice(g->devices[0]));
    }
    g->slot_used[s] = true;
    g->issued++;
    return VR_OK;
}

int vr_group_wait(vr_group_handle g)
{
    if (!g) return VR_E_INVALID;
    if (g->issued == g->completed) return gfail(g, VR_E_INVALID, "vr_group_wait: no frame in flight");
    const int n = (int)g->members.size();
    const int s = g->completed % kSlots;
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(hipEventSynchronize(g->assembled[s]));
    float worst = 0.0f;
    for (int r = 0; r < n; r++) {
        VRG_HIP(hipSetDevice(g->devices[(size_t)r]));
        VRG_HIP(hipEventSynchronize(g->t1[s][(size_t)r]));
        float ms = 0.0f;
        VRG_HIP(hipEventElapsedTime(&ms, g->t0[s][(size_t)r], g->t1[s][(size_t)r]));
        worst = std::max(worst, ms);
    }
    g->kerneltime_sum += worst;
    g->current = s;
    g->completed++;
    VRG_HIP(hipSetDevice(g->devices[0]));
    return VR_OK;
}

int vr_group_render(vr_group_handle g)
{
    if (!g) return VR_E_INVALID;
    while (g->issued > g->completed) {                                  // frames left in flight by the async interface
        const int rc = vr_group_wait(g);
        if (rc != VR_OK) return rc;
    }
    const int rc = vr_group_render_async(g);
    return rc != VR_OK ? rc : vr_group_wait(g);
}

float vr_group_kernel_ms_take(vr_group_handle g)
{
    if (!g) return 0.0f;
    const float v = g->kerneltime_sum;
    g->kerneltime_sum = 0.0f;
    return v;
}

void *vr_group_framebuffer_device(vr_group_handle g) { return (g && g->current >= 0) ? g->frame[g->current] : (g ? g->frame[0] : nullptr); }

int vr_group_read_pixels(vr_group_handle g, float *rgba, size_t n_floats)
{
    if (!g) return VR_E_INVALID;
    const size_t need = (size_t)g->fb_w * (size_t)g->fb_h * 4u;
    const float4 *src = g->current >= 0 ? g->frame[g->current] : g->frame[0];
    if (!src || !rgba || n_floats < need) return gfail(g, VR_E_INVALID, "vr_group_read_pixels: no frame / buffer too small");
    VRG_HIP(hipSetDevice(g->devices[0]));
    VRG_HIP(hipMemcpy(rgba, src, need * sizeof(float), hipMemcpyDeviceToHost));
    return VR_OK;
}

int vr_group_present_rgba8(vr_group_handle g, const unsigned char **frame)
{
    if (!g || !frame) return VR_E_INVALID;
    *frame = nullptr;
    const float4 *src = g->current >= 0 ? g->frame[g->current] : nullptr;
    if (!src) return gfail(g, VR_E_INVALID, "vr_group_present_rgba8: no completed frame");
    try {
        VRG_HIP(hipSetDevice(g->devices[0]));
        *frame = g->members[0]->core.presentRGBA8(src);
    } catch (const std::exception &e) {
        return gfail(g, VR_E_HIP, e.what());
    }
    return VR_OK;
}

}  // extern "C"
