#!/usr/bin/env python3
"""debug aid: does a fresh handle that imports profiles/launch_choices.bin start the headline configuration on the settled kernel?
prints the launch choice of its first frames (bit 8 = trial) and the keys of the blob"""
import importlib, struct, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
vra = importlib.import_module("volume-renderer_amd")
R = vra.renderer
def cold_blob():
    r = vra.RendererCore(0)
    r.setup((1920, 1080)); r.loadShader("VolumeRenderer.cs"); r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED)
    r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
    r.setWindow(0, 4095); r.setAlpha(0.004)
    for _ in range(300):
        r.render()
    b = r.exportChoices()
    r.close()
    return b


blob = Path(sys.argv[1]).read_bytes() if len(sys.argv) > 1 else cold_blob()
n = struct.unpack_from("<I", blob, 12)[0]
print("blob:", len(blob), "bytes,", n, "entries, build id %016x" % struct.unpack_from("<Q", blob, 16)[0], "device", blob[24:88].split(b"\0")[0])
keys = {}
for i in range(n):
    key, ncand, settled, heur = struct.unpack_from("<Qiii", blob, 88 + 56 * i)
    keys[key] = settled


def headline(import_first):
    r = vra.RendererCore(0)
    r.setup((1920, 1080)); r.loadShader("VolumeRenderer.cs")
    acc = r.importChoices(blob) if import_first else None
    r.setQuirks(0); r.setLayout(R.LAYOUT_BRICKED)
    r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024, 1024, 1024), 2, 0x9E3779B9)
    r.setWindow(0, 4095); r.setSkipEmpty(False); r.setKernelVariant(0); r.setPack12(True); r.setAlpha(0.004); r.setFilter(R.FILTER_NEAREST)
    if not import_first:
        acc = r.importChoices(blob)
    seq = []
    for _ in range(40):
        r.renderAsync(); seq.append(r.last_launch_choice)
    r.synchronize()
    mine = r.exportChoices()
    m = struct.unpack_from("<I", mine, 12)[0]
    mykeys = [struct.unpack_from("<Qiii", mine, 88 + 56 * i) for i in range(m)]
    print("import", "before" if import_first else "after", "the volume: accepted", acc, "choices", seq, "kernel", r.last_kernel_name)
    print("   this handle's table:", [(hex(k), s, k in keys) for k, _, s, _ in mykeys])
    r.close()


headline(True)
headline(False)
