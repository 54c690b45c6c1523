#!/usr/bin/env python3
"""kernel time of one rank's shard of the cfg3 frame at N = 8 (16-row stripes, rank 5) in every mode of the
specialised kernels, relay kernel (automatic for sparse launches) against the fast kernel (variant 2)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vra = importlib.import_module("volume-renderer_amd"); R = vra.renderer
r = vra.RendererCore(0); r.setup((1920, 1080)); r.loadShader("x.cs"); r.setQuirks(0); r.setLayout(1)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095)
r.setRowStripes(16, 5, 8)


def ms(variant):
    r.setKernelVariant(variant)
    for _ in range(200):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(30):
        r.render()
    return r.kernelMsTake() / 30, r.last_kernel_name


for name, setup in (("grey composite", lambda: None), ("MIP", lambda: (r.setMIP(True), r.setAlpha(0.3))),
                    ("transfer function", lambda: r.setTransferFunction([0, 141, 149, 255], [[0, 0, 0, 0], [.55, .55, .55, .759], [.58, .58, .58, .45], [1, 1, 1, 1]])),
                    ("view top", lambda: r.setInitialCameraRotation(True, False))):
    r.setMIP(False); r.setTransferFunction(); r.setInitialCameraRotation(False, False); r.setAlpha(0.004)
    setup()
    a, ka = ms(0)
    b, kb = ms(2)
    print(f"{name:20s} N=8 shard: {ka} {a:.3f} ms   {kb} {b:.3f} ms")
