import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
r = vra.RendererCore(0)
r.setup((1920, 1080)); r.loadShader("x"); r.setQuirks(0)
r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
r.setWindow(64, 4095); r.setAlpha(0.004)
def ms(n=20):
    for _ in range(100):
        r.renderAsync()
    r.synchronize(); r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return round(r.kernelMsTake() / n, 4)
for pose in ("default", "offaxis"):
    r.resetCamera()
    if pose == "offaxis":
        r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    out = {}
    r.setSkipEmpty(False); out["plain"] = ms()
    r.setSkipEmpty(True); out["skip"] = ms()
    print(pose, r.last_kernel_name, out, flush=True)
