import importlib, time, sys
sys.path.insert(0, "/root/repo")
import numpy as np
vra = importlib.import_module("volume-renderer_amd"); R = vra.renderer
r = vra.RendererCore(0); r.setup((1920, 1080)); r.loadShader("x.cs"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,)*3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
for _ in range(3): r.render()
r.kernelMsTake()
t0 = time.perf_counter()
for i in range(50): r.render()
t1 = time.perf_counter(); k = r.kernelMsTake()
print(f"static camera: {(t1-t0)/50*1e3:.3f} ms/frame wall, kernel {k/50:.3f} ms")
t0 = time.perf_counter()
for i in range(50):
    r.cameraOrient(0.0, 0.0, 0.002)      # orbit a little every frame (GUI mouse drag)
    r.render()
t1 = time.perf_counter(); k = r.kernelMsTake()
print(f"orbiting camera: {(t1-t0)/50*1e3:.3f} ms/frame wall, kernel {k/50:.3f} ms")
t0 = time.perf_counter()
for i in range(50):
    r.setWindow(0, 4095 - i)             # drag the window slider: a new divisor every frame
    r.render()
t1 = time.perf_counter(); k = r.kernelMsTake()
print(f"window slider: {(t1-t0)/50*1e3:.3f} ms/frame wall, kernel {k/50:.3f} ms")
t0 = time.perf_counter()
for i in range(50):
    r.setAlpha(0.004 + 1e-5 * i)
    r.render()
t1 = time.perf_counter(); k = r.kernelMsTake()
print(f"alpha slider: {(t1-t0)/50*1e3:.3f} ms/frame wall, kernel {k/50:.3f} ms")
