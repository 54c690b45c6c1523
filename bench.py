#!/usr/bin/env python3
"""bench.py -- the ray-march hot path on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W
  N>1 works both ways: launched by torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK /
  WORLD_SIZE / MASTER_* from the environment), or as the plain command above -- bench.py then
  starts the N ranks itself (re-executing itself under torch.distributed.run on 127.0.0.1) and
  still prints exactly ONE JSON line.

A "step" is one frame: one launch of the ray-march kernel over this rank's image-row
shard, plus (N>1) the RCCL gather of the shards to rank 0 (--collective all_gather: to every rank) and the
assembly kernel there (de-interleave + (grey, alpha) -> RGBA).  Workload (config.workload): synthetic 1024^3 uint16 volume generated in HBM,
1920x1080 RGBA32F target, reference default camera, NEAREST sampling (the reference's
effective filter), window [0,4095], alpha_scale 0.004 ("deep": no early ray
termination, every ray traverses the whole box).  The frame is fixed while N grows
(strong scaling): rank k renders cyclic 16-row stripes k, k+N, ...

Prints ONE JSON line on rank 0.  `value` = samples actually fetched by the whole job per
second (Msamples/s), inputs resident in HBM; `mpixels_per_s` is the companion number of
BASELINE.json's metric.  `roofline` uses the algorithmic bytes of SURVEY 8(d):
S*b + W*H*16 per launch over the kernel's average launch duration measured with HIP
events on the launch stream.  `cpu_baseline` is the scalar oracle (oracle/vr_oracle.c,
"port") timed on this box's host cores on a bounded row sample of the same frame.

Set-up (untimed, like generating the volume) ends with --clock-ramp-frames frames (default
150, ~75 ms): an idle MI355X reaches its sustained clocks only after ~50 ms of load, and the
metric is the sustained rate of a renderer that is running; and with frames rendered one at a time until the
measured launch choice of this configuration has settled (`config.settle_frames`; a handful when
profiles/launch_choices.bin -- vr_import_choices -- already holds it).  Then W warm-up steps, a barrier,
EXACTLY K timed steps, a barrier.  `config.clock_ramp_frames` records it; 0 switches both off.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6300 achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--clock-ramp-frames", type=int, default=150,
                    help="untimed frames rendered during set-up, before the W warm-up steps: an idle MI355X needs "
                         "~50 ms of load to reach its sustained clocks (the first ~25 frames run ~15 %% slower)")
    ap.add_argument("--volume", type=int, default=1024, help="synthetic volume edge (voxels)")
    ap.add_argument("--dims", type=int, nargs=3, default=None, help="non-cubic synthetic volume NX NY NZ (overrides --volume)")
    ap.add_argument("--bytes", type=int, default=2, choices=(1, 2))
    ap.add_argument("--synth", choices=("noise_ball", "sphere", "noise_ball_ct"), default="noise_ball",
                    help="synthetic generator: the seeded noise ball (configs 2-4), config 0/1's integer sphere (uint8, radius 7/16 of the edge), or the "
                         "16-bit noise ball stored the way CT data is (every voxel + 1000; default window 1000 5095)")
    ap.add_argument("--skip-empty", action="store_true", help="exact empty-space skipping (config 4)")
    ap.add_argument("--window", type=int, nargs=2, default=None, help="min max (default: full range of the generator)")
    ap.add_argument("--tf", action="store_true", help="default alpha-spline transfer function (config 4)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--alpha", type=float, default=0.004)
    ap.add_argument("--filter", choices=("nearest", "trilinear"), default="nearest")
    ap.add_argument("--layout", choices=("linear", "bricked"), default=os.environ.get("VR_BENCH_LAYOUT", "bricked"))
    ap.add_argument("--partition", choices=("stripes", "contiguous"), default="stripes")
    ap.add_argument("--stripe-rows", type=int, default=16)
    ap.add_argument("--gather-format", choices=("auto", "rgba", "ga"), default="auto",
                    help="N > 1: what the all_gather moves -- RGBA32F, or (grey, alpha) float2 in the grey modes "
                         "(r == g == b there; expanded to RGBA after the gather); auto = ga when the mode is grey")
    ap.add_argument("--collective", choices=("gather", "all_gather"), default="gather",
                    help="N > 1: gather the shards to rank 0 (default: the frame is needed in one place, like the reference's "
                         "single framebuffer; RCCL send/recv over rank 0's point-to-point links) or all_gather them to every rank")
    ap.add_argument("--pose", choices=("default", "offaxis"), default="default")
    ap.add_argument("--mip", action="store_true", help="maximum intensity projection (VolumeRenderer.cs:141-173; RendererCore::setMIP)")
    ap.add_argument("--view", choices=("front", "top", "bottom"), default="front",
                    help="the reference's initial camera rotation (VolumeRenderer.cs:186-189; RendererCore::setInitialCameraRotation)")
    ap.add_argument("--kernel-variant", type=int, default=0, help="vr_set_kernel_variant (0 auto, 1 generic, 2 fast kernel with the plain loop, 3 always relay, 5 fast kernel with the pipelined loop)")
    ap.add_argument("--no-pack12", action="store_true", help="never gather from the 12-bit packed copy (vr_set_pack12(0))")
    ap.add_argument("--shard", type=int, nargs=2, default=None, metavar=("WORLD", "RANK"),
                    help="single process: time only the kernel of rank RANK's shard of a WORLD-GPU frame (no collective)")
    ap.add_argument("--choices", default=str(ROOT / "profiles" / "launch_choices.bin"),
                    help="blob of settled launch choices (vr_import_choices) loaded into every renderer of this run when it was measured by this build on "
                         "this device model; 'none' = start cold")
    ap.add_argument("--save-choices", action="store_true", help="write what this run's renderers settled on back to --choices (merged with what was imported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=0, help="0 = auto (about 15 s of CPU work)")
    ap.add_argument("--extras", action="store_true",
                    help="also time the slow secondary measurements (PCIe read-back, orbiting camera); the secondary "
                         "kernel regimes (unpacked, shallow, off-axis, trilinear) are always in `extras` at N=1")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary kernel regimes")
    ap.add_argument("--native-group", action="store_true",
                    help="N > 1 without torch.distributed: ONE process drives --gpus members through the C ABI's vr_group_* "
                         "(a renderer per device, RCCL send/recv gather to device 0, assembly kernel there); when the box has "
                         "fewer devices than members, members share devices (validation only)")
    ap.add_argument("--dataset", default=None,
                    help="render a volume file (.raw + .raw.inf, or .pvm) instead of the synthetic volume; --bytes gives its "
                         "voxel size.  VR_DATA_BONSAI / VR_DATA_HEAD name the reference's two datasets (README.md:6-7) for "
                         "--dataset bonsai / --dataset head")
    return ap.parse_args()


def self_launch(args, run=None) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU, RCCL over
    xGMI) by re-executing this file under torch.distributed.run, forward rank 0's JSON line.  If that run fails
    (first contact with a node's RCCL / rendezvous), retry ONCE through --native-group (one process, the C ABI's
    vr_group_*: no torch.distributed, its own RCCL communicators) and say in the line which launcher produced it.
    `run` = subprocess.run (a test hook)."""
    import socket
    import subprocess

    run = run or subprocess.run
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    me = str(Path(__file__).resolve())

    def json_line(proc):
        line = None
        for ln in (proc.stdout or "").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)
        return line

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), me] + sys.argv[1:]
    proc = run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = json_line(proc)
    if proc.returncode == 0 and line is not None:
        print(line, flush=True)
        return 0
    print(f"[bench] the {args.gpus}-rank torch.distributed.run failed (exit code {proc.returncode}); retrying once through --native-group",
          file=sys.stderr)
    proc2 = run([sys.executable, me] + sys.argv[1:] + ["--native-group"], env=env, stdout=subprocess.PIPE, text=True)
    line = json_line(proc2)
    if proc2.returncode != 0 or line is None:
        print(f"[bench] the native-group retry failed too (exit code {proc2.returncode})", file=sys.stderr)
        return proc2.returncode or proc.returncode or 1
    rec = json.loads(line)
    rec.setdefault("config", {})["launcher_fallback"] = f"torch.distributed.run exited with code {proc.returncode}; this line is the native vr_group retry"
    print(json.dumps(rec), flush=True)
    return 0


def run_native_group(args) -> int:
    """the headline frame through vr_group_*: one process, one renderer per device, the frame complete on device 0.
    A step = one frame through vr_group_render_async() + vr_group_wait(): shard kernels on every device concurrently +
    gather + assembly, one frame kept in flight; wall clock around K completed frames."""
    import numpy as np
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    vra = importlib.import_module("volume-renderer_amd")
    R = vra.renderer
    ndev = torch.cuda.device_count()
    if ndev < 1:
        print("bench.py needs a HIP device (no CPU fallback exists)", file=sys.stderr)
        return 1
    n, W, H, N, b = args.gpus, args.width, args.height, args.volume, args.bytes
    devices = [r % ndev for r in range(n)]
    dims = tuple(args.dims) if args.dims else (N, N, N)
    vmax = 4095 if b == 2 else 255
    win = tuple(args.window) if args.window else (0, vmax)
    g = vra.RendererGroup(devices)
    g.setup((W, H), partition=args.partition, stripe_rows=args.stripe_rows)

    def configure(m):
        m.loadShader("VolumeRenderer.cs"); m.setQuirks(0)
        m.setLayout(R.LAYOUT_BRICKED if args.layout == "bricked" else R.LAYOUT_LINEAR)
        m.generateSynthetic(R.SYNTH_NOISE_BALL, dims, b, 0x9E3779B9)
        m.setWindow(*win); m.setAlpha(args.alpha); m.setKernelVariant(args.kernel_variant); m.setPack12(not args.no_pack12)
        m.setFilter(R.FILTER_TRILINEAR if args.filter == "trilinear" else R.FILTER_NEAREST)
        if args.pose == "offaxis":
            m.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    g.each(configure)
    # (set-up: clock ramp + the members' measured launch choices settle -- exploration, 96 frames, one re-validation -- before the timed frames)
    for _ in range(max(args.clock_ramp_frames, 0) // 4 + (220 if args.kernel_variant == 0 and args.clock_ramp_frames > 0 else 0) + args.warmup):
        g.render()
    g.kernelMsTake()
    # two frame slots (vr_group_render_async / vr_group_wait): the gather + assembly of frame i overlap the shard
    # kernels of frame i + 1, like the torch.distributed path; exactly K frames are completed inside the timed region
    t0 = time.perf_counter()
    g.renderAsync()
    for _ in range(args.steps - 1):
        g.renderAsync()
        g.wait()
    g.wait()
    ms_per_step = (time.perf_counter() - t0) * 1e3 / args.steps
    kernel_ms = g.kernelMsTake() / args.steps
    frame = g.readPixels()
    with vra.RendererCore(devices[0]) as single:          # untimed self-check against one device's own full frame
        single.setup((W, H)); configure(single)
        single.render()
        want = single.readPixels()
        samples = single.countSamples()
    ok = bool(np.array_equal(frame.view(np.uint32), want.view(np.uint32)))
    print(json.dumps({
        "metric": "Msamples/sec (+ Mpixels/sec), 1024^3 uint16 @ 1920x1080", "value": round(samples / (ms_per_step * 1e-3) / 1e6, 1),
        "unit": "Msamples/s", "mpixels_per_s": round(W * H / (ms_per_step * 1e-3) / 1e6, 1), "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "kernel_ms": round(kernel_ms, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic noise-ball {'x'.join(map(str, dims))} uint{8 * b}, {W}x{H} RGBA32F, {args.filter.upper()} filter, "
                               f"window [{win[0]},{win[1]}], alpha_scale {args.alpha}, {args.layout} layout",
                   "samples_per_frame": samples, "launcher": "native vr_group (one process, C ABI; vr_group_render_async / vr_group_wait, two frame slots)", "devices": devices,
                   "transport": g.transport, "partition": args.partition, "kernel": g.members[0].last_kernel_name},
        "multi_gpu_frame_bit_exact": ok, "n_ranks_seen": len(g.members)}), flush=True)
    g.close()
    return 0 if ok else 1


def main():
    args = parse_args()
    if args.native_group:
        raise SystemExit(run_native_group(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))      # before torch / HIP are touched in this process
    import numpy as np
    # the host driver only supports dmabuf IPC: RCCL / cross-process device memory need this
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    init_fallback = None
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    if os.environ.get("VR_BENCH_BACKEND", "nccl") != "nccl" or os.environ.get("VR_BENCH_FAIL_NCCL_INIT"):
        local_rank = local_rank % torch.cuda.device_count()      # validation: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm.  VR_BENCH_BACKEND=gloo is a validation hook only (several
        # ranks sharing one GPU, gather staged through host memory).
        from datetime import timedelta

        backend = os.environ.get("VR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            # The nccl -> gloo fallback must be ONE decision for the whole job (round-5 advisor: decided per rank, a partial
            # failure left ranks in different default groups until the 180 s timeout).  The ranks of this single-node job agree
            # through marker files keyed by the rendezvous port: every rank reports whether its init came back, waits for all
            # reports, and if any failed, EVERY rank drops to gloo.
            import glob
            import tempfile
            tag = os.path.join(tempfile.gettempdir(), f"vr_bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}_{os.getppid()}")      # (the launcher's pid: markers of an earlier, crashed job never match)
            ok, why = True, ""
            try:
                fail_hook = os.environ.get("VR_BENCH_FAIL_NCCL_INIT")
                if fail_hook and (fail_hook == "1" or str(rank) in fail_hook.split(",")):
                    raise RuntimeError("VR_BENCH_FAIL_NCCL_INIT (test hook)")
                dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=180))
            except Exception as exc:
                ok, why = False, repr(exc)[:200]
            Path(f"{tag}_rank{rank}.{'ok' if ok else 'fail'}").write_text(why)
            t_wait, files = time.perf_counter(), []
            while True:                                                  # ONE snapshot decides, the same on every rank
                files = sorted(glob.glob(f"{tag}_rank*.*"))
                if len(files) >= world or time.perf_counter() - t_wait > 200.0:
                    break
                time.sleep(0.05)
            failed = [f for f in files if f.endswith(".fail")]
            if failed or len(files) < world:                             # RCCL unusable somewhere: the shards travel over gloo, and the line says so
                init_fallback = (why or ("rank " + failed[0].rsplit("_rank", 1)[1].split(".")[0] + " failed its nccl init" if failed
                                         else "a rank never reported its nccl init"))[:200]
                print(f"[bench] rank {rank}: nccl backend unusable on this job ({init_fallback}); every rank continues on gloo", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo", timeout=timedelta(seconds=180))
            dist.barrier()                                               # every rank has read the markers
            if rank == 0:
                for f in files:
                    try:
                        os.unlink(f)
                    except OSError:
                        pass
        else:
            dist.init_process_group(backend)
        # a host-side group next to RCCL: bookkeeping reductions (sample totals, timings, flags) and -- if the RCCL
        # preflight below fails -- the fallback transport of the shards themselves (through host memory)
        host_pg = dist.new_group(backend="gloo", timeout=timedelta(seconds=180))

    def host_reduce(values, op="sum", dtype=None):
        """all-reduce of a few host numbers over the gloo group (never touches RCCL)"""
        if world == 1:
            return list(values)
        t = torch.tensor(list(values), dtype=dtype or torch.float64)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op], group=host_pg)
        return t.tolist()

    vra = importlib.import_module("volume-renderer_amd")
    from importlib import import_module

    sharding = import_module("volume-renderer_amd.sharding")
    R = vra.renderer

    W, H, N, b = args.width, args.height, args.volume, args.bytes
    vmax = 4095 if b == 2 else 255
    choices_load(args)
    r = vra.RendererCore(local_rank)
    r.setup((W, H))
    r.loadShader("VolumeRenderer.cs")
    headline_choices_imported = choices_import(r)
    r.setQuirks(0)   # explicit window below is what the kernel sees (no +1000, no truncated grid)
    r.setLayout(R.LAYOUT_BRICKED if args.layout == "bricked" else R.LAYOUT_LINEAR)
    dims = tuple(args.dims) if args.dims else (N, N, N)
    dataset = args.dataset
    if dataset in ("bonsai", "head"):            # the reference's datasets, when supplied on the box
        dataset = os.environ.get("VR_DATA_BONSAI" if dataset == "bonsai" else "VR_DATA_HEAD")
        if not dataset:
            raise SystemExit("--dataset bonsai/head needs VR_DATA_BONSAI / VR_DATA_HEAD to name the file")
    if dataset:
        r.readVolumeData(dataset, b)
        dims = tuple(r.dims[0])
        win = tuple(args.window) if args.window else r.window
    else:
        if args.synth == "sphere":
            r.generateSynthetic(R.SYNTH_SPHERE_U8, dims, 1, dims[0] * 7 // 16)
        elif args.synth == "noise_ball_ct":
            r.generateSynthetic(R.SYNTH_NOISE_BALL_CT, dims, 2, 0x9E3779B9)
        else:
            r.generateSynthetic(R.SYNTH_NOISE_BALL, dims, b, 0x9E3779B9)
        win = tuple(args.window) if args.window else ((1000, 5095) if args.synth == "noise_ball_ct" else (0, vmax))
        if args.synth == "noise_ball_ct" and not args.window:
            args.window = list(win)            # cpu_baseline / extras read the window from args
    r.setWindow(*win)
    if args.tf:   # the widget's default alpha knots (AlphaControlSplineWidget.cpp:56-59), black->white ramp
        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
    r.setSkipEmpty(args.skip_empty)
    r.setKernelVariant(args.kernel_variant)
    r.setPack12(not args.no_pack12)
    r.setAlpha(args.alpha)
    r.setFilter(R.FILTER_TRILINEAR if args.filter == "trilinear" else R.FILTER_NEAREST)
    r.setMIP(args.mip)
    if args.view != "front":
        r.setInitialCameraRotation(args.view == "top", args.view == "bottom")      # (also resets the camera, like the reference)
    if args.pose == "offaxis":
        r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)   # zenith 60 deg, azimuth 45 deg

    plan = sharding.plan_rows(H, world, rank, args.partition, args.stripe_rows)
    if args.shard and world == 1:
        plan = sharding.plan_rows(H, args.shard[0], args.shard[1], args.partition, args.stripe_rows)
    sharding.apply_plan(r, plan)
    if args.shard and world == 1:
        plan = sharding.RowPlan(H, 1, 0, "contiguous", args.stripe_rows, plan.local_rows)   # gather-free bookkeeping
    # a dedicated (non-null) torch stream: the kernel, the HIP events and RCCL all use it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    r.setStream(stream.cuda_stream)
    # Two frame slots: with N > 1 the RCCL all_gather + de-interleave of frame i runs on a
    # second stream while the kernel of frame i+1 renders into the other slot.
    nslots = 2 if world > 1 else 1
    comm_stream = torch.cuda.Stream(dev) if world > 1 else None
    # N > 1 in a grey mode: shards are rendered and gathered as (grey, alpha) -- half the bytes over xGMI
    grey_alpha = world > 1 and not args.tf and args.gather_format in ("auto", "ga")
    if args.gather_format == "ga" and args.tf:
        raise SystemExit("--gather-format ga needs a grey mode (no --tf)")
    C4 = 2 if grey_alpha else 4
    r.setFramebufferFormat(R.FB_GREYALPHA32F if grey_alpha else R.FB_RGBA32F)
    locals_ = [torch.zeros((plan.local_rows, W, C4), dtype=torch.float32, device=dev) for _ in range(nslots)]
    gathered = [torch.empty((world * plan.local_rows, W, C4), dtype=torch.float32, device=dev) for _ in range(nslots)] if world > 1 else [None]
    frames = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(nslots)] if world > 1 else [None]
    torch.cuda.synchronize(dev)        # the targets' zero fills ran on torch's default stream: done before any kernel on `stream` writes them
    ev_rendered = [torch.cuda.Event() for _ in range(nslots)]
    ev_gathered = [torch.cuda.Event() for _ in range(nslots)]
    local = locals_[0]
    r.setFramebufferExternal(local.data_ptr())
    r.setFramebufferCompact(True)
    index = torch.as_tensor(sharding.gather_index(plan), device=dev) if world > 1 and plan.mode == "stripes" else None

    # ---- untimed: exact sample count of this rank's shard (instrumented kernel)
    r.setFramebufferExternal(0)
    r.setFramebufferCompact(False)
    my_samples = r.countSamples()
    r.setFramebufferExternal(local.data_ptr())
    r.setFramebufferCompact(True)
    total_samples = int(host_reduce([my_samples], "sum", torch.int64)[0])

    step_no = [0]
    trial_frames = [0]                                   # launches that were a TRIAL of the measured choice (vr_get_launch_choice bit 8)
    root = 0 if (world > 1 and args.collective == "gather") else None
    # ---- RCCL preflight (untimed): the exact collective of a step, once, on a zero frame.  Every rank reports over the
    # host group whether it came back; if any did not, ALL ranks switch the shards to the host-staged gloo transport and
    # the line says so -- a measured line on a degraded transport beats no line on the node's first RCCL contact.
    transport = {"name": "RCCL (nccl backend) over xGMI" if world > 1 and dist.get_backend() == "nccl" else ("gloo (validation)" if world > 1 else None),
                 "group": None, "host_staged": None}
    if world > 1 and init_fallback:
        transport["name"] = f"gloo through host memory (FALLBACK: the nccl backend failed to initialise: {init_fallback})"
    if world > 1 and (dist.get_backend() == "nccl" or os.environ.get("VR_BENCH_FAIL_PREFLIGHT")) and not os.environ.get("VR_BENCH_SKIP_PREFLIGHT"):
        ok, why = 1, ""
        try:
            if os.environ.get("VR_BENCH_FAIL_PREFLIGHT"):
                raise RuntimeError("VR_BENCH_FAIL_PREFLIGHT (test hook)")
            sharding.gather_frame(locals_[0], plan, out=gathered[0], index=index, root=root)
            torch.cuda.synchronize(dev)
        except Exception as exc:
            ok, why = 0, repr(exc)[:200]
        if int(host_reduce([ok], "min", torch.int64)[0]) == 0:
            transport = {"name": "gloo through host memory (FALLBACK: the RCCL preflight failed" + (f" on rank {rank}: {why}" if why else " on another rank") + ")",
                         "group": host_pg, "host_staged": True}

    def step(ev_pair=None):
        slot = step_no[0] % nslots
        step_no[0] += 1
        if world > 1:
            stream.wait_event(ev_gathered[slot])            # slot free again (frame i-2 gathered)
            r.setFramebufferExternal(locals_[slot].data_ptr())
        if ev_pair:
            ev_pair[0].record(stream)
        r.renderAsync()
        if ev_pair:
            ev_pair[1].record(stream)
            trial_frames[0] += (r.last_launch_choice >> 8) & 1
        if world == 1:
            return locals_[0][:H]
        ev_rendered[slot].record(stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev_rendered[slot])
            frame = sharding.gather_frame(locals_[slot], plan, out=gathered[slot], index=index, root=root,
                                          assembler=r, frame_out=frames[slot],     # one kernel: de-interleave + (grey, alpha) -> RGBA
                                          group=transport["group"], host_staged=transport["host_staged"])
            ev_gathered[slot].record(comm_stream)
        return frame

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(group=transport["group"])
            torch.cuda.synchronize(dev)

    frame = None
    for _ in range(max(args.clock_ramp_frames, 0)):     # set-up: bring the GPU to its sustained clocks
        frame = step()
    barrier()
    # set-up, disclosed as config.settle_frames: the measured launch choice of a configuration this process has not seen (no
    # imported entry) tries its candidates over the first frames and re-validates once 96 frames later; those frames are rendered
    # HERE, one at a time, until 100 consecutive launches were the settled kernel (at most 400) -- the timed region below then
    # contains no trial launches (config.trial_frames_in_timed_region counts them all the same)
    settle_frames, quiet = 0, 0
    if args.clock_ramp_frames > 0 and args.kernel_variant == 0:
        # (the short way only for the configuration the committed blob was measured on -- the default command line; anything else, a
        # shard of the frame included, is a configuration of its own and gets the whole exploration + re-validation before timing)
        need_quiet = 3 if (headline_choices_imported > 0 and headline_key(args) == "1024^3x2B_1920x1080_nearest_bricked_a0.004" and not args.shard) else 100
        # (N > 1: every step is a collective, so every rank renders the same FIXED number of frames -- and the committed blob was
        # measured on full frames, a rank's shard is a configuration of its own: exploration, 96 frames, the one re-validation)
        while settle_frames < 220 if world > 1 else (settle_frames < 400 and quiet < need_quiet):
            frame = step()
            torch.cuda.synchronize(dev)
            quiet = 0 if (r.last_launch_choice >> 8) & 1 else quiet + 1
            settle_frames += 1
    barrier()
    for _ in range(args.warmup):
        frame = step()
    barrier()
    # kernel-only duration: HIP events on the launch stream around each launch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame = step(ev[i])
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = sum(a.elapsed_time(c) for a, c in ev) / args.steps
    elapsed, kernel_ms_max = host_reduce([elapsed, kernel_ms], "max")
    per_rank_kernel_ms = None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(kernel_ms, 4), group=host_pg)
        per_rank_kernel_ms = per_rank
    ms_per_step = elapsed * 1e3 / args.steps

    # untimed self-check of the N > 1 path: the gathered frame must equal this rank's own
    # full-frame render bit for bit
    gather_ok = None
    if world > 1:
        torch.cuda.synchronize(dev)
        r.setRowRange(0, -1); r.setRowStripes(1, 0, 1)
        r.setFramebufferExternal(0); r.setFramebufferCompact(False)
        r.render()
        full = r.readPixels()
        gather_ok = True                                    # ranks that hold no frame (gather to root) have nothing to compare
        if frame is not None:
            gather_ok = bool(np.array_equal(full.view(np.uint32), frame.cpu().numpy().view(np.uint32)))
        sharding.apply_plan(r, plan)
        r.setFramebufferExternal(locals_[0].data_ptr()); r.setFramebufferCompact(True)
        gather_ok = bool(int(host_reduce([1 if gather_ok else 0], "min", torch.int64)[0]))
        if transport["group"] is None:
            t = torch.tensor([1], dtype=torch.int64, device=dev)     # ranks counted by the RCCL communicator itself
            dist.all_reduce(t)
            n_ranks_seen = int(t.item())
        else:
            n_ranks_seen = int(host_reduce([1], "sum", torch.int64)[0])

    result = None
    if rank == 0:
        value = total_samples / (ms_per_step * 1e-3) / 1e6
        mpix = W * H / (ms_per_step * 1e-3) / 1e6
        alg_bytes = my_samples * b + plan_pixels(plan, W) * 16      # this rank's launch
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "Msamples/sec (+ Mpixels/sec), 1024^3 uint16 @ 1920x1080",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "mpixels_per_s": round(mpix, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "kernel_ms": round(kernel_ms_max, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",                                   # the arithmetic type of the path (voxels: config.voxel_dtype)
            "data": "synthetic",
            "config": {
                "workload": (f"file {Path(dataset).name} " if dataset else "synthetic noise-ball ") + f"{'x'.join(map(str, dims))} uint{8 * b} "
                            + ("" if dataset else "(generated in HBM, seed 0x9E3779B9), ")
                            + f"{W}x{H} RGBA32F, reference default camera" + (" (off-axis pose)" if args.pose != "default" else "")
                            + f", {args.filter.upper()} filter, window [{win[0]},{win[1]}], alpha_scale {args.alpha}, "
                            + ("transfer function, " if args.tf else "") + ("empty-space skipping, " if args.skip_empty else "")
                            + f"iterative accumulation, {args.layout} layout",
                "samples_per_frame": total_samples,
                "partition": "single GPU" if world == 1 else f"{args.partition} rows x{world}"
                             + (f" ({args.stripe_rows}-row stripes)" if args.partition == "stripes" else "")
                             + f" + RCCL {args.collective}" + (" of (grey, alpha) shards" if grey_alpha else "") + (" to rank 0" if root is not None else ""),
                "kernel": r.last_kernel_name,
                "voxel_dtype": "u%d" % (8 * b),
                "clock_ramp_frames": max(args.clock_ramp_frames, 0),
                "settle_frames": settle_frames,
                # the measured launch choice: entries taken over from profiles/launch_choices.bin (vr_import_choices; 0 = this run
                # explored its candidates itself during the untimed ramp) and how many of the TIMED launches were still trials
                "launch_choices_imported": headline_choices_imported,
                "trial_frames_in_timed_region": trial_frames[0],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": load_traffic(args, world, r.last_kernel_name),
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if world == 1:
            # the bound that binds: VALU issue next to HBM (a committed PMC pass of this command x the hot loops' measured
            # cost per instruction); `roofline` above stays the metric's own HBM figure
            result["roofline_valu"] = valu_roofline(headline_key(args), r.last_kernel_name, kernel_ms)
            rv = result["roofline_valu"] or {}
            # what binds, from counters where there are some: the vector L1's look-up rate when its PMC figure is there and
            # highest (round 5: removing 7 % of the loop's VALU issue cycles moved nothing, so "valu" -- a cost MODEL -- is only
            # named when no L1 figure contradicts it)
            # every candidate with the KIND of evidence behind it (round-5 verdict: an argmax over unlike fractions proves nothing):
            bounds = [{"name": "hbm", "frac": result["roofline"]["frac"],
                       "kind": "measured kernel time against the metric's algorithmic bytes at the datasheet peak (8 TB/s)"}]
            if rv.get("frac") is not None:
                bounds.append({"name": "valu issue", "frac": rv["frac"],
                               "kind": "MODELLED: counted wave-instructions (PMC pass of this command) x per-instruction issue costs from a micro-benchmark"})
            if (rv.get("l1_rate") or {}).get("frac") is not None:
                bounds.append({"name": "vector L1 look-up rate", "frac": rv["l1_rate"]["frac"],
                               "kind": "MEASURED counter (TCP_TOTAL_CACHE_ACCESSES per CU per shader cycle) against a capacity of ~1 per cycle measured by tools/ubench/tcp_rate.hip"})
            result["bounds"] = bounds
            # what binds: a measured counter at or near its capacity (>= 0.9) is named; a modelled fraction is only named when no
            # measured one reaches that, and says so
            measured = [b_ for b_ in bounds[1:] if b_["kind"].startswith("MEASURED") and b_["frac"] >= 0.9]
            if measured:
                result["binding_bound"] = max(measured, key=lambda b_: b_["frac"])["name"] + " (measured counter)"
            elif len(bounds) > 1:
                top = max(bounds, key=lambda b_: b_["frac"])
                result["binding_bound"] = top["name"] + (" (modelled)" if top["kind"].startswith("MODELLED") else "")
            else:
                result["binding_bound"] = "hbm (no counter pass committed for these sources)"
        if world == 1:
            # the box's own achievable HBM read rate (streaming read of the resident volume),
            # measured after the timed region ("confirm the peak on the box")
            try:
                stream_gbps = r.measureStreamRead(5)
                vol_bytes = int(dims[0]) * int(dims[1]) * int(dims[2]) * b
                packed = r.pack12Bytes()
                if packed:      # the launch gathered from the lossless 12-bit copy (1.5 B per voxel)
                    result["config"]["resident_copy"] = f"12-bit packed copy of the uint16 volume ({packed} B) for the prefix gathers"
                    vol_bytes = packed
                result["roofline"]["measured_stream_read"] = round(stream_gbps, 1)
                # reading the volume once + writing the frame at that rate: the floor of any layout
                result["roofline"]["compulsory_floor_ms"] = round((vol_bytes + W * H * 16) / stream_gbps / 1e6, 4)
            except Exception as exc:                      # measurement aid only
                result["roofline"]["measured_stream_read"] = None
                print(f"[bench] stream-read probe failed: {exc}", file=sys.stderr)
        if world > 1:
            result["multi_gpu_frame_bit_exact"] = gather_ok
            result["n_ranks_seen"] = n_ranks_seen
            result["transport"] = transport["name"]
            result["per_rank_kernel_ms"] = per_rank_kernel_ms
            result["config"]["launcher"] = "torch.distributed.run (one process per GPU)"
            result["overlap"] = f"{args.collective} of frame i on a second stream overlaps the kernel of frame i+1"
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, r, frame, value)
        if world == 1 and not args.no_extras and not args.shard:
            result["extras"] = extras(args, r, local, stream, b, W, H)
    choices_collect(r)
    if rank == 0 and world == 1 and not args.no_extras and not args.shard and not args.dataset and args.volume == 1024 and not args.dims:
        # the other BASELINE configs (1, 2, 4), each on its own renderer; the headline's volume is released first
        r.close()
        try:
            result["extras"].update(config_extras(local_rank))
        except Exception as exc:                              # evidence only: never costs the headline line
            result["extras"]["baseline_configs_error"] = repr(exc)
    if world > 1:
        dist.barrier(group=host_pg)
        dist.destroy_process_group()
    r.close()
    if rank == 0 and args.save_choices and args.choices and args.choices != "none":
        merged = choices_merge([CHOICES["blob"]] + CHOICES["exported"])
        if merged:
            Path(args.choices).write_bytes(merged)
            print(f"[bench] {len(merged)} bytes of settled launch choices written to {args.choices}", file=sys.stderr)
    if result is not None:
        print(json.dumps(result), flush=True)


CHOICES = {"blob": None, "imported": 0, "exported": []}


def choices_load(args):
    """the committed blob of settled launch choices (profiles/launch_choices.bin), if any"""
    if args.choices and args.choices != "none" and Path(args.choices).is_file():
        CHOICES["blob"] = Path(args.choices).read_bytes()


def choices_import(r):
    """give renderer r what an earlier run measured (only accepted from the same library build on the same device model)"""
    if CHOICES["blob"]:
        try:
            n = r.importChoices(CHOICES["blob"])
            CHOICES["imported"] += n
            return n
        except Exception as exc:                              # a stale / foreign file never costs the run
            print(f"[bench] launch choices not imported: {exc}", file=sys.stderr)
    return 0


def choices_collect(r):
    try:
        CHOICES["exported"].append(r.exportChoices())
    except Exception as exc:
        print(f"[bench] launch choices not exported: {exc}", file=sys.stderr)


def choices_merge(blobs):
    """union by key of blobs with the same header (88-byte header, 56-byte records: csrc/renderer_core.cpp ChoiceHeader / ChoiceRecord); later blobs win"""
    import struct

    head, recs = None, {}
    for b in blobs:
        if not b or len(b) < 88 or b[:8] != b"VRCHOICE":
            continue
        h = bytearray(b[:88])
        ident = bytes(h[:12]) + bytes(h[16:])                 # everything but the count
        if head is None:
            head = (ident, h)
        elif head[0] != ident:
            continue
        n = struct.unpack_from("<I", b, 12)[0]
        for i in range(n):
            rec = b[88 + 56 * i: 88 + 56 * (i + 1)]
            if len(rec) == 56:
                recs[rec[:8]] = rec
    if head is None:
        return b""
    h = head[1]
    struct.pack_into("<I", h, 12, len(recs))
    return bytes(h) + b"".join(recs.values())


def plan_pixels(plan, W):
    import numpy as np

    return int((plan.global_rows() >= 0).sum()) * W


def kernel_source_hash() -> str:
    """SHA-256 over the kernel sources: a PMC traffic figure is only quoted for the kernels it was
    measured on (tools/pmc_traffic.py stores the same hash next to each figure)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / "volume-renderer_amd" / "csrc").glob("*.hip")) + sorted((ROOT / "volume-renderer_amd" / "csrc").glob("vr_*.h")) \
            + sorted((ROOT / "volume-renderer_amd" / "csrc").glob("tile_schedule.*")):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def headline_key(args):
    """key of this command in profiles/traffic.json / valu.json, or None for commands that have no committed PMC pass"""
    if args.dims or args.window or args.tf or args.skip_empty or args.pose != "default" or args.dataset or args.no_pack12 or args.synth != "noise_ball" or args.mip or args.view != "front":
        return None
    return f"{args.volume}^3x{args.bytes}B_{args.width}x{args.height}_{args.filter}_{args.layout}_a{args.alpha}"


def load_traffic(args, world, kernel):
    """HBM bytes per launch from the committed PMC profile of this same command
    (profiles/traffic.json, written by tools/pmc_traffic.py), or null -- also null when the
    kernels have changed since the counters were collected (source hash mismatch) or the
    counters belong to another kernel than the one this run launched."""
    if world != 1:
        return None
    return traffic_entry(headline_key(args), kernel)


def cpu_baseline(args, r, frame, gpu_msamples):
    """Scalar oracle ("port") on host cores: a bounded row sample of the same frame,
    single thread; also checks those rows against the GPU frame."""
    import numpy as np
    import oracle   # test infrastructure: timed as the CPU baseline, never on the product path

    W, H, b = args.width, args.height, args.bytes
    vmax = 4095 if b == 2 else 255
    win = tuple(args.window) if args.window else (0, vmax)
    vol = r.readVolume()
    cam = r.getCameraBlock()
    tf = r.getTransferLut() if args.tf else None
    gpu = frame.cpu().numpy()

    def run(rows, threads):
        p = oracle.OracleParams(W, H, cam=cam, alpha_scale=args.alpha, min_val=win[0], max_val=win[1], is_mip=int(args.mip),
                                view_top=int(args.view == "top"), view_bottom=int(args.view == "bottom"),
                                filter=1 if args.filter == "trilinear" else 0, threads=threads, tf_rgba=tf)
        out = np.zeros((H, W, 4), dtype=np.float32)
        samples, secs = 0, 0.0
        for y in rows:
            p.row_begin, p.row_end = int(y), int(y) + 1
            t0 = time.perf_counter()
            _, s = oracle.render(vol, p, out=out)
            secs += time.perf_counter() - t0
            samples += s
        return out, samples, secs

    # calibrate on 4 rows through the middle, then pick a stride for ~15 s
    probe_rows = [H // 2 - 60, H // 2 - 20, H // 2 + 20, H // 2 + 60]
    _, ps, pt = run(probe_rows, 1)
    rate = ps / max(pt, 1e-9)
    stride = args.cpu_row_stride
    if stride <= 0:
        est_full = 4.8e8 * (max(vol.shape) / 1024.0) * (W * H) / (1920 * 1080) / max(rate, 1.0)
        stride = max(1, int(np.ceil(est_full / 15.0)))
    rows = list(range(stride // 2, H, stride))
    out, samples, secs = run(rows, 1)
    diff = float(np.max(np.abs(out[rows] - gpu[rows])))
    bitexact = bool(np.array_equal(out[rows].view(np.uint32), gpu[rows].view(np.uint32)))

    # the OpenMP row-parallel variant of the same oracle on every host core: a
    # contiguous block of rows about the image centre sized for ~5 s
    cores = os.cpu_count() or 1
    all_cores = None
    if cores > 1:
        per_row = max(ps / len(probe_rows), 1.0)
        nrows = int(min(H, max(cores, rate * cores * 5.0 / per_row)))
        y0 = max(0, H // 2 - nrows // 2)
        p = oracle.OracleParams(W, H, cam=cam, alpha_scale=args.alpha, min_val=win[0], max_val=win[1], is_mip=int(args.mip),
                                view_top=int(args.view == "top"), view_bottom=int(args.view == "bottom"),
                                filter=1 if args.filter == "trilinear" else 0, threads=cores, tf_rgba=tf)
        p.row_begin, p.row_end = y0, min(H, y0 + nrows)
        out2 = np.zeros((H, W, 4), dtype=np.float32)
        t0 = time.perf_counter()
        _, s2 = oracle.render(vol, p, out=out2)
        t2 = time.perf_counter() - t0
        blk = slice(p.row_begin, p.row_end)
        all_cores = {
            "value": round(s2 / t2 / 1e6, 2), "unit": "Msamples/s", "cores": cores,
            "sample": f"rows {p.row_begin}..{p.row_end - 1} ({s2} samples, {t2:.1f} s), OpenMP over rows",
            "parity_bit_exact_on_sample": bool(np.array_equal(out2[blk].view(np.uint32), gpu[blk].view(np.uint32))),
        }
    return {
        "value": round(samples / secs / 1e6, 2),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"every {stride}th image row ({len(rows)} of {H} rows, {samples} samples, {secs:.1f} s) of the same frame",
        "host_cores_available": os.cpu_count(),
        "parity_max_abs_diff_on_sample": diff,
        "parity_bit_exact_on_sample": bitexact,
        "gpu_over_cpu": round(gpu_msamples / (samples / secs / 1e6), 1),
        "all_cores": all_cores,
    }


def traffic_entry(key, kernel):
    """PMC traffic (HBM bytes per launch) of one configuration from profiles/traffic.json, only when it was collected on
    these kernel sources AND for the kernel this run launched (round 3 quoted another kernel's counters)"""
    p = ROOT / "profiles" / "traffic.json"
    try:
        entry = json.loads(p.read_text()).get(key) if key else None
        if isinstance(entry, dict) and entry.get("kernel_source_hash") == kernel_source_hash() and entry.get("kernel") == kernel:
            return entry.get("bytes")
    except Exception:
        pass
    return None


def valu_roofline(key, kernel, kernel_ms):
    """VALU-issue fraction of one configuration: measured wave-instructions (profiles/valu.json: SQ_INSTS_VALU and the
    launch's shader cycles, tools/pmc_valu.py) x the cost per instruction of the kernel family's hot loops
    (profiles/valu_cpi.json, tools/valu_cpi.py) / (1024 SIMDs x cycles).  The cycles are the PMC pass's own
    (GRBM_GUI_ACTIVE / 8), scaled to this run's kernel time at the same clock.  None when there is no committed pass
    for these kernel sources and this kernel."""
    try:
        v = json.loads((ROOT / "profiles" / "valu.json").read_text()).get(key) if key else None
        cpi_all = json.loads((ROOT / "profiles" / "valu_cpi.json").read_text())
        if not isinstance(v, dict) or v.get("kernel") != kernel or v.get("kernel_source_hash") != kernel_source_hash():
            return None
        fam = kernel
        if fam == "raymarch_tslab_kernel" and ", true>" in v.get("instance", ""):
            fam = "raymarch_tslab_kernel_half"
        cpi = cpi_all["families"][fam]["cpi"]
        issue_cycles = v["valu_wave_insts"] * cpi / 1024.0
        res = {"bound": "valu", "wave_insts": v["valu_wave_insts"], "cycles_per_inst": cpi, "issue_cycles_per_simd": round(issue_cycles, 1),
               "shader_cycles": v["shader_cycles"], "frac": round(issue_cycles / v["shader_cycles"], 4), "kernel": kernel,
               "unit": "VALU issue cycles / shader cycles (PMC pass: profiles/valu.json; costs: profiles/valu_cpi.json)"}
        c = v.get("counters") or {}
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            # the vector L1's address-processing rate (docs/lab-notebook.md "The second limit"; profiles/r05_q4_addressing_experiment.txt):
            # one access per cycle per CU, so accesses / (256 CUs x shader cycles) is the share of the launch the L1s were busy
            res["l1_rate"] = {"tcp_accesses": c["TCP_TOTAL_CACHE_ACCESSES_sum"], "frac": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * v["shader_cycles"]), 4),
                              "accesses_per_gather": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"], 1) if c.get("SQ_INSTS_VMEM_RD") else None,
                              "unit": "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x shader cycles) = tag look-ups per CU per shader cycle; the L1 processes about one "
                                      "per cycle (4 per distinct 128-B line of a gather, 16 at least: tools/ubench/tcp_rate.hip), so values near or above 1 mean saturated "
                                      "(the two counters come from separate PMC passes: a few % of slack)"}
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVE_CYCLES"):
            # counters only, no cost model (round-4 verdict item 7): VALU-active time of the chip's 1024 SIMDs if the counter's
            # documented unit (quad-cycles) holds, and the share of the waves' lifetime spent waiting on any instruction
            res["measured"] = {"valu_active_frac_if_quad_cycles": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * v["shader_cycles"]), 4),
                               "wait_inst_any_over_wave_cycles": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                               "mean_waves_per_simd_if_quad_cycles": round(c["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * v["shader_cycles"]), 2),
                               "counters": c}
        return res
    except Exception:
        return None


def sparse_row_parity(r, rows, **oracle_kw):
    """the rows `rows` of the frame r renders now against the CPU oracle (test infrastructure: the checker).  Returns
    (bit_exact, max_abs_diff)."""
    import numpy as np
    import oracle

    r.render()
    got = r.readPixels()
    H, W = got.shape[0], got.shape[1]
    vol = oracle_kw.pop("vol")
    p = oracle.OracleParams(W, H, cam=r.getCameraBlock(), threads=min(os.cpu_count() or 1, 16), **oracle_kw)
    want = np.zeros((H, W, 4), dtype=np.float32)
    for y in rows:
        p.row_begin, p.row_end = int(y), int(y) + 1
        oracle.render(vol, p, out=want)
    return bool(np.array_equal(got[rows].view(np.uint32), want[rows].view(np.uint32))), float(np.max(np.abs(got[rows] - want[rows])))


def config_extras(device):
    """BASELINE.json configs 1, 2 and 4 on one MI355X, timed like the headline (kernel only, HIP events, sustained
    clocks): kernel_ms, S, roofline fraction (algorithmic bytes S*b + W*H*16), kernel name, PMC traffic when known.
      cfg1_shape             256^3 u8 sphere (Bonsai's shape; the file itself when VR_DATA_BONSAI names it), 1280x720,
                             alpha_scale 1 (the reference's default), fixed step
      cfg2_shape_ert_window  512x512x452 u16 noise ball (Head CT's shape; the .pvm when VR_DATA_HEAD names it), 1080p,
                             window [1000, 5095] = the reference's +1000 quirk on [0, 4095], alpha 0.05: early ray termination
      cfg4_grey              2048^3 u8 (8 GiB, 64-bit offsets), 3840x2160, grey ramp, window [8, 255], alpha 0.004
      cfg4_tf_skip           the same through the default alpha-spline transfer function + exact empty-space skipping
    cfg1 / cfg2 / cfg4_grey are also timed with TRILINEAR filtering (<name>_trilinear), cfg4 with TRILINEAR + skipping in both modes."""
    import numpy as np

    vra = importlib.import_module("volume-renderer_amd")
    R = vra.renderer
    out = {}

    def timed(r, name, b, W, H, steps, key=None, **okw):
        s = r.countSamples()
        ok, diff = sparse_row_parity(r, [H // 4 + 3, H // 2, (3 * H) // 4 - 5], **okw)      # three rows through the volume, against the oracle
        # sustained clocks AND a settled launch choice: the measured work model tries its candidates over the first frames of a
        # configuration (a 6-ms frame of config 4 had only a dozen of them in 0.08 s, and the timed steps then included trial launches)
        t0, frames = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.08 or frames < 96:
            for _ in range(4):
                r.renderAsync()
            r.synchronize()
            frames += 4
        r.render(); r.kernelMsTake()
        for _ in range(steps):
            r.render()
        ms = r.kernelMsTake() / steps
        gbps = (s * b + W * H * 16) / (ms * 1e-3) / 1e9
        out[name] = {"kernel_ms": round(ms, 4), "samples": s, "msamples_per_s": round(s / ms / 1e3, 1),
                     "mpixels_per_s": round(W * H / ms / 1e3, 1), "roofline_frac": round(gbps / HBM_PEAK_GBPS, 4),
                     "kernel": r.last_kernel_name, "traffic": traffic_entry(key or name, r.last_kernel_name),
                     "roofline_valu": valu_roofline(key or name, r.last_kernel_name, ms),
                     "parity_bit_exact_on_sample": ok, "parity_max_abs_diff_on_sample": diff}
        fv = (out[name]["roofline_valu"] or {}).get("frac")
        out[name]["bound"] = "valu" if fv is not None and fv > out[name]["roofline_frac"] else "hbm"

    class renderer:
        """a renderer per configuration that starts on the imported launch choices and hands back what it settled on"""

        def __init__(self, W, H):
            self.r = vra.RendererCore(device)
            self.r.setup((W, H)); self.r.loadShader("VolumeRenderer.cs"); self.r.setQuirks(0); self.r.setLayout(R.LAYOUT_BRICKED)
            choices_import(self.r)

        def __enter__(self):
            return self.r

        def __exit__(self, *exc):
            choices_collect(self.r)
            self.r.close()

    with renderer(1280, 720) as r:
        f = os.environ.get("VR_DATA_BONSAI")
        if f:
            r.readVolumeData(f, 1)
        else:
            r.generateSynthetic(R.SYNTH_SPHERE_U8, (256, 256, 256), 1, 112)
        r.setWindow(0, 255); r.setAlpha(1.0)
        vol = r.readVolume()
        timed(r, "cfg1_shape", 1, 1280, 720, 40, vol=vol, alpha_scale=1.0, min_val=0, max_val=255)
        out["cfg1_shape"]["data"] = Path(f).name if f else "synthetic sphere 256^3 u8"
        r.setFilter(R.FILTER_TRILINEAR)                      # the north-star's filter on the same configuration
        timed(r, "cfg1_shape_trilinear", 1, 1280, 720, 40, vol=vol, alpha_scale=1.0, min_val=0, max_val=255, filter=1)
        # the same two with exact empty-space skipping switched on (the sphere sits in zeros: the space in front of it is skipped)
        r.setSkipEmpty(True)
        timed(r, "cfg1_shape_trilinear_skip", 1, 1280, 720, 40, vol=vol, alpha_scale=1.0, min_val=0, max_val=255, filter=1)
        r.setFilter(R.FILTER_NEAREST)
        timed(r, "cfg1_shape_skip", 1, 1280, 720, 40, vol=vol, alpha_scale=1.0, min_val=0, max_val=255)
        r.setSkipEmpty(False)
    with renderer(1920, 1080) as r:
        f = os.environ.get("VR_DATA_HEAD")
        if f:
            r.readVolumeData(f, 2)
            lo, hi = r.window
            r.setWindow(lo, hi)
        else:
            r.generateSynthetic(R.SYNTH_NOISE_BALL, (512, 512, 452), 2, 0x9E3779B9)
            r.setWindow(1000, 5095)
        r.setAlpha(0.05)
        vol = r.readVolume()
        lo, hi = r.window
        timed(r, "cfg2_shape_ert_window", 2, 1920, 1080, 40, vol=vol, alpha_scale=0.05, min_val=lo, max_val=hi)
        out["cfg2_shape_ert_window"]["data"] = Path(f).name if f else "synthetic noise ball 512x512x452 u16"
        r.setFilter(R.FILTER_TRILINEAR)
        timed(r, "cfg2_shape_ert_window_trilinear", 2, 1920, 1080, 40, vol=vol, alpha_scale=0.05, min_val=lo, max_val=hi, filter=1)
        # ... and with exact empty-space skipping: everything at or below the window's lower end (the air around a CT scan) is skipped
        r.setSkipEmpty(True)
        timed(r, "cfg2_shape_ert_window_trilinear_skip", 2, 1920, 1080, 40, vol=vol, alpha_scale=0.05, min_val=lo, max_val=hi, filter=1)
        r.setFilter(R.FILTER_NEAREST)
        timed(r, "cfg2_shape_ert_window_skip", 2, 1920, 1080, 40, vol=vol, alpha_scale=0.05, min_val=lo, max_val=hi)
        r.setSkipEmpty(False)
    with renderer(3840, 2160) as r:
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (2048, 2048, 2048), 1, 0x9E3779B9)
        r.setWindow(8, 255); r.setAlpha(0.004)
        vol = r.readVolume()                                 # 8 GiB over PCIe once, for the oracle's rows
        timed(r, "cfg4_grey", 1, 3840, 2160, 10, vol=vol, alpha_scale=0.004, min_val=8, max_val=255)
        r.setFilter(R.FILTER_TRILINEAR)                      # 10 GiB apron copy next to the 8 GiB volume; the LDS-staged kernel (64-bit DMA addresses)
        timed(r, "cfg4_grey_trilinear", 1, 3840, 2160, 5, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, filter=1)
        out["cfg4_grey_trilinear"]["apron_copy_bytes"] = r.trilinearCopyBytes()
        r.setMIP(True)                                       # the reference's second mode at config 4's size, both filters
        timed(r, "cfg4_mip_trilinear", 1, 3840, 2160, 5, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, filter=1, is_mip=1)
        r.setFilter(R.FILTER_NEAREST)
        timed(r, "cfg4_mip", 1, 3840, 2160, 10, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, is_mip=1)
        r.setMIP(False)
        r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [0.55, 0.55, 0.55, 0.759], [0.58, 0.58, 0.58, 0.45], [1, 1, 1, 1]])
        r.setSkipEmpty(True)
        timed(r, "cfg4_tf_skip", 1, 3840, 2160, 10, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, tf_rgba=r.getTransferLut())
        # north-star's config 4 in full: TRILINEAR + "adaptive step" (exact empty-space skipping, per tile and brick layer on the
        # LDS-staged kernel) + the transfer function in LDS; and the grey ramp with skipping next to cfg4_grey_trilinear
        r.setFilter(R.FILTER_TRILINEAR)
        timed(r, "cfg4_tf_skip_trilinear", 1, 3840, 2160, 5, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, tf_rgba=r.getTransferLut(), filter=1)
        r.setTransferFunction()
        timed(r, "cfg4_grey_skip_trilinear", 1, 3840, 2160, 5, vol=vol, alpha_scale=0.004, min_val=8, max_val=255, filter=1)
    for name, e in out.items():
        if "skip" in name:
            # S counts LOGICAL samples (the frame and the per-pixel counts are those of the launch without skipping); the skipped ones are
            # never fetched, so this entry's roofline fraction is samples-per-second in bytes, not bytes moved -- `traffic` is what moved
            e["empty_space_skipping"] = True
    return out


def extras(args, r, local, stream, b, W, H):
    """secondary regimes of the same workload, timed the same way (kernel only, HIP events on the
    launch stream, single GPU), each with its own roofline fraction (algorithmic bytes S*b + W*H*16
    of THAT regime): the headline without the 12-bit packed copy, the shallow (early-ray-termination)
    regime, the off-axis pose and TRILINEAR filtering"""
    import torch

    vra = importlib.import_module("volume-renderer_amd")
    R = vra.renderer
    out = {}

    vol_host = [None]

    def timed(name, steps=20, **okw):
        r.setFramebufferExternal(0); r.setFramebufferCompact(False)
        s = r.countSamples()
        if vol_host[0] is None:
            vol_host[0] = r.readVolume()
        vmax = 4095 if b == 2 else 255
        win = tuple(args.window) if args.window else (0, vmax)
        kw = dict(vol=vol_host[0], alpha_scale=args.alpha, min_val=win[0], max_val=win[1], filter=1 if args.filter == "trilinear" else 0,
                  tf_rgba=r.getTransferLut() if args.tf else None, is_mip=int(args.mip), view_top=int(args.view == "top"),
                  view_bottom=int(args.view == "bottom"))
        kw.update(okw)
        ok, diff = sparse_row_parity(r, [H // 4 + 3, H // 2, (3 * H) // 4 - 5], **kw)
        r.setFramebufferExternal(local.data_ptr()); r.setFramebufferCompact(True)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:      # back to sustained clocks (the CPU baseline left the GPU idle)
            for _ in range(10):
                r.renderAsync()
            torch.cuda.synchronize()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            r.renderAsync()
        c.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(c) / steps
        gbps = (s * b + W * H * 16) / (ms * 1e-3) / 1e9
        out[name] = {"kernel_ms": round(ms, 4), "samples": s, "msamples_per_s": round(s / ms / 1e3, 1),
                     "mpixels_per_s": round(W * H / ms / 1e3, 1), "roofline_frac": round(gbps / HBM_PEAK_GBPS, 4),
                     "kernel": r.last_kernel_name, "traffic": traffic_entry(name, r.last_kernel_name),
                     "roofline_valu": valu_roofline(name, r.last_kernel_name, ms),
                     "parity_bit_exact_on_sample": ok, "parity_max_abs_diff_on_sample": diff}
        fv = (out[name]["roofline_valu"] or {}).get("frac")
        out[name]["bound"] = "valu" if fv is not None and fv > out[name]["roofline_frac"] else "hbm"

    if args.extras:
        # PCIe-inclusive: kernel + D2H of the finished RGBA32F frame (vr_read_pixels), for DESIGN.md
        r.setFramebufferExternal(0); r.setFramebufferCompact(False)
        r.render(); r.readPixels()
        t0 = time.perf_counter()
        for _ in range(5):
            r.renderAsync()
            r.readPixels()
        out["frame_plus_readback_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 4)
        r.readPixelsRGBA8()
        t0 = time.perf_counter()
        for _ in range(5):
            r.renderAsync()
            r.readPixelsRGBA8()
        out["frame_plus_rgba8_readback_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 4)
        # the shim's presentation path: vr_render + vr_present_rgba8 (two pinned host frames, the copy of frame i under kernel i + 1)
        r.render(); r.presentRGBA8(copy=False)
        t0 = time.perf_counter()
        for _ in range(20):
            r.render()
            r.presentRGBA8(copy=False)
        out["frame_plus_present_rgba8_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        r.setFramebufferExternal(local.data_ptr()); r.setFramebufferCompact(True)
    if b == 2 and not args.no_pack12 and r.pack12Bytes():
        r.setPack12(False)
        timed("headline_without_pack12")       # general 16-bit data (voxels above 4095 somewhere)
        r.setPack12(True)
    r.setAlpha(1.0)
    timed("shallow_alpha1_ert", alpha_scale=1.0)
    r.setAlpha(args.alpha)
    if args.filter == "nearest":
        r.setFilter(R.FILTER_TRILINEAR)
        timed("trilinear_deep", steps=10, filter=1)
        out["trilinear_deep"]["apron_copy_bytes"] = r.trilinearCopyBytes()      # TRILINEAR's own copy of the volume (vr_set_trilinear_copy)
        r.setFilter(R.FILTER_NEAREST)
    if args.pose == "default":
        r.cameraOrient(0.0, -(3.14159265 / 6) / 0.7, (3.14159265 / 4) / 0.7)
        timed("offaxis_deep", steps=10)
        if args.filter == "nearest":
            r.setFilter(R.FILTER_TRILINEAR)
            timed("trilinear_offaxis_deep", steps=10, filter=1)      # oblique view: half layers of the staged kernel (round 3: the batched kernel)
            r.setFilter(R.FILTER_NEAREST)
        r.resetCamera()
    if args.pose == "default" and not args.mip and args.view == "front":
        # the reference's other modes at size (round-5 verdict): MIP (VolumeRenderer.cs:141-173) and the two rotated views
        # (:186-189; src/RendererCore.cpp:84-98 -- they permute the volume axes the rays advance along in memory), NEAREST and TRILINEAR
        for name, setup, okw in (("mip_deep", lambda: r.setMIP(True), dict(is_mip=1)),
                                 ("view_top_deep", lambda: r.setInitialCameraRotation(True, False), dict(view_top=1)),
                                 ("view_bottom_deep", lambda: r.setInitialCameraRotation(False, True), dict(view_bottom=1))):
            setup()
            timed(name if args.filter == "nearest" else "trilinear_" + name, **okw)
            if args.filter == "nearest":
                r.setFilter(R.FILTER_TRILINEAR)
                timed("trilinear_" + name, steps=10, filter=1, **okw)
                r.setFilter(R.FILTER_NEAREST)
            r.setMIP(False); r.setInitialCameraRotation(False, False)
        r.resetCamera()
    if args.extras:
        # interactive use: the camera moves every frame (GUI orbit), host work included (wall clock)
        r.render()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r.cameraOrient(0.0, 0.0, 0.002)
            r.renderAsync()
        torch.cuda.synchronize()
        out["orbiting_camera_wall_ms_per_frame"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        r.resetCamera()
    if b == 2 and not args.no_pack12 and not args.dataset and args.synth == "noise_ball" and not args.window and args.filter == "nearest":
        # the same volume stored the way CT data is (every voxel + 1000, the reference's own convention,
        # src/RendererCore.cpp:66-67) under the window 1000 .. 5095: the 12-bit copy packs (voxel - min), so this
        # must run at the headline's rate, not at headline_without_pack12's (last: it replaces the resident volume)
        dims = tuple(args.dims) if args.dims else (args.volume,) * 3
        r.setFramebufferExternal(0); r.setFramebufferCompact(False)
        r.generateSynthetic(R.SYNTH_NOISE_BALL_CT, dims, 2, 0x9E3779B9)
        r.setWindow(1000, 5095)
        vol_host[0] = None
        timed("headline_offset1000", min_val=1000, max_val=5095)
        out["headline_offset1000"]["packed_copy_bytes"] = r.pack12Bytes()
    return out


if __name__ == "__main__":
    main()
