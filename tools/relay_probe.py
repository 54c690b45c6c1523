import importlib, os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
vra = importlib.import_module("volume-renderer_amd"); R = vra.renderer
sharding = importlib.import_module("volume-renderer_amd.sharding")
r = vra.RendererCore(0); r.setup((1920, 1080)); r.loadShader("x.cs"); r.setQuirks(0); r.setLayout(1)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
for world in (1, 2, 4, 8):
    for variant in (2, 3):
        r.setKernelVariant(variant)
        plan = sharding.plan_rows(1080, world, 1, "stripes", 16)
        sharding.apply_plan(r, plan)
        for _ in range(3): r.render()
        r.kernelMsTake()
        for _ in range(10): r.render()
        print(f"N={world} variant {variant} {r.last_kernel_name}: {r.kernelMsTake()/10:.3f} ms")
