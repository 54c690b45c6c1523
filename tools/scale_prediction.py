#!/usr/bin/env python3
"""What a run of `bench.py --gpus N` on an 8-GPU node should show, predicted from ONE MI355X (round-5 verdict item 2): no multi-GPU node
has been available to this build, so the first hardware run needs something to be compared with.

Measured here, per N in {1, 2, 4, 8} on the headline workload (1024^3 u16 @ 1920x1080, NEAREST, alpha 0.004, cyclic 16-row stripes):
  shard_kernel_ms      every rank's shard kernel (measured choice, sustained clocks): max and mean over the N ranks
  assemble_ms          vr_assemble_shards of N (grey, alpha) shards into the RGBA32F frame on the root (one kernel)
  rccl_self_ms         a grouped ncclSend + ncclRecv of a full (grey, alpha) frame to SELF through vr_group (transport 2) minus the same
                       frame with a device copy: RCCL's software path on this box, no xGMI involved -- a floor for the gather's latency
Modelled (stated, not measured): the root receives N - 1 shards of W * H * 8 / N bytes concurrently over its N - 1 point-to-point
xGMI links at `link_gbps` per direction (default 54 = 70 % of the 76.8 GB/s a 153.6 GB/s-bidirectional link gives one way).
Predicted: ms_per_step(N) = max(shard_kernel_ms.max, gather_ms + assemble_ms) with the two-slot overlap bench.py runs
(`pipelined`), and their sum without overlap (`serial`, an upper bound); host launch overhead is not modelled.
Writes profiles/r06_scale_prediction.json.   usage: tools/scale_prediction.py [link_gbps]"""
import importlib, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402  (device buffers for the assembly measurement; also makes torch's HIP runtime the one loaded)
vra = importlib.import_module("volume-renderer_amd")
sharding = importlib.import_module("volume-renderer_amd.sharding")
R = vra.renderer
LINK = float(sys.argv[1]) if len(sys.argv) > 1 else 54.0
W, H = 1920, 1080


def configure(r):
    r.loadShader("VolumeRenderer.cs"); r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED)
    r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
    r.setWindow(0, 4095); r.setAlpha(0.004)


def kernel_ms(r, n=30):
    for _ in range(320):                                  # exploration, settling, the one re-validation; sustained clocks
        r.render()
    r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


out = {"workload": "synthetic noise-ball 1024^3 uint16, 1920x1080, NEAREST, alpha 0.004, cyclic 16-row stripes, (grey, alpha) shards",
       "measured_on": "one MI355X (every rank's shard rendered in turn on the same device)", "link_gbps_per_direction_assumed": LINK, "per_n": {}}
r = vra.RendererCore(0)
r.setup((W, H)); configure(r)
dev = torch.device("cuda", 0)
for N in (1, 2, 4, 8):
    per_rank, kernels = [], set()
    for rank in range(N):
        r.setRowStripes(16 if N > 1 else 1, rank, N)
        per_rank.append(round(kernel_ms(r), 4)); kernels.add(r.last_kernel_name)
    local_rows = sharding.plan_rows(H, N, 0, "stripes", 16).local_rows if N > 1 else H      # every rank's (padded) shard height
    entry = {"shard_kernel_ms": {"max": max(per_rank), "mean": round(sum(per_rank) / N, 4), "per_rank": per_rank}, "kernels": sorted(kernels)}
    if N > 1:
        gathered = torch.zeros((N * local_rows, W, 2), dtype=torch.float32, device=dev)
        frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream()
        for _ in range(20):
            r.assembleShards(gathered.data_ptr(), frame.data_ptr(), N, local_rows, 16, 2, s.cuda_stream)
        a.record(s)
        for _ in range(100):
            r.assembleShards(gathered.data_ptr(), frame.data_ptr(), N, local_rows, 16, 2, s.cuda_stream)
        b.record(s); torch.cuda.synchronize()
        entry["assemble_ms"] = round(a.elapsed_time(b) / 100, 4)
        shard_bytes = local_rows * W * 8
        entry["shard_bytes"] = shard_bytes
        entry["gather_wire_ms_modelled"] = round(shard_bytes / (LINK * 1e9) * 1e3, 4)      # N - 1 links in parallel: one shard's time
    out["per_n"][str(N)] = entry
r.setRowStripes(1, 0, 1)
r.close()

# RCCL's software path on this box: a one-member group whose shard travels through ncclSend / ncclRecv to self
rccl = {}
try:
    for mode, name in ((0, "device_copy"), (2, "rccl_self")):
        g = vra.renderer.RendererGroup([0])
        g.setTransport(mode); g.setup((W, H))            # (the transport is fixed at setup)
        g.each(configure)
        for _ in range(200):
            g.render()
        t0 = time.perf_counter()
        for _ in range(100):
            g.render()
        rccl[name] = {"frame_wall_ms": round((time.perf_counter() - t0) * 10, 4), "transport": g.transport}
        g.close()
    rccl["rccl_self_minus_copy_ms"] = round(rccl["rccl_self"]["frame_wall_ms"] - rccl["device_copy"]["frame_wall_ms"], 4)
except Exception as exc:                                   # librccl missing on the box: the model then has no software floor
    rccl["error"] = repr(exc)[:200]
out["rccl_software_path"] = rccl
floor = max(rccl.get("rccl_self_minus_copy_ms", 0.0), 0.0)
base = out["per_n"]["1"]["shard_kernel_ms"]["max"]
for N in ("2", "4", "8"):
    e = out["per_n"][N]
    gather = e["gather_wire_ms_modelled"] + floor
    e["gather_ms_predicted"] = round(gather, 4)
    e["predicted_ms_per_step"] = {"pipelined": round(max(e["shard_kernel_ms"]["max"], gather + e["assemble_ms"]), 4),
                                  "serial": round(e["shard_kernel_ms"]["max"] + gather + e["assemble_ms"], 4)}
    e["predicted_speedup_over_1gpu"] = {k: round(base / v, 2) for k, v in e["predicted_ms_per_step"].items()}
out["note"] = ("kernel-only ceiling = 1-GPU kernel / shard kernel; the pipelined figure assumes bench.py's two frame slots hide the gather behind the "
               "next frame's kernel; host launch overhead (one Python step: events, ctypes launch, dist.gather) is not in the model and will show on the N = 8 line")
(ROOT / "profiles").mkdir(exist_ok=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
Path(ROOT / "gpurun_out" / "r06_scale_prediction.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
