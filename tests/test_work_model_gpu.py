"""The measured work model (RendererCore::tuneChoose, vr_set_autotune): what it settles on must be what a sweep of the forced
kernel variants finds at sustained clocks -- also when it starts on a cold GPU (round-3 verdict: "can settle on a cold-clock
measurement and never looks again") -- and a slider drag must not keep it exploring (round-3 advisor)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ms_of(r, n=30):
    r.render(); r.kernelMsTake()
    for _ in range(n):
        r.render()
    return r.kernelMsTake() / n


def sustained(r, seconds=0.15):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            r.renderAsync()
        r.synchronize()


@pytest.mark.parametrize("case", ["nearest_default", "trilinear_oblique_u16", "nearest_shard8"])
def test_a_cold_renderer_ends_on_the_sweeps_choice(vra, case):
    R = vra.renderer
    time.sleep(1.5)                                              # let the clocks drop: the first frames below are measured cold
    with vra.RendererCore(0) as r:
        r.setup((1920, 1080)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (1024,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
        forced = (2, 5, 3)
        if case == "trilinear_oblique_u16":
            r.setFilter(R.FILTER_TRILINEAR); r.cameraOrient(0.0, 0.66, -1.65)
            forced = (2, 6, 8, 9, 10)
        elif case == "nearest_shard8":
            r.setRowStripes(16, 3, 8)
        for _ in range(300):                                     # a cold start: exploration, settling, the one re-validation
            r.render()
        choice, kernel = r.last_launch_choice, r.last_kernel_name
        sustained(r)
        auto = ms_of(r)
        assert r.last_launch_choice == choice                   # it stays settled
        sweep = {}
        for v in forced:
            r.setKernelVariant(v)
            sustained(r, 0.05)
            sweep[v] = ms_of(r)
        r.setKernelVariant(0)
        best = min(sweep.values())
        print(f"{case}: work model -> choice {choice} ({kernel}) {auto:.4f} ms; forced variants {{{', '.join(f'{v}: {t:.4f}' for v, t in sweep.items())}}}")
        assert auto <= best * 1.04, (case, choice, auto, sweep)


def test_a_slider_drag_does_not_keep_the_model_exploring(vra):
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup((1920, 1080)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (512,) * 3, 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.02)
        seen = []
        for i in range(240):                                     # opacity and window sliders dragged every frame
            r.setAlpha(0.02 + 1e-5 * i)
            r.setWindow(0, 4095 - i)
            r.render()
            seen.append(r.last_launch_choice)
        final = seen[-1]
        off = sum(1 for c in seen[40:] if c != final)
        assert off <= 16, (final, off, seen[:60])                # one exploration per bucket crossed at most, not one per frame
        assert len(set(seen[-60:])) == 1


def test_imported_choices_start_a_cold_handle_on_the_settled_kernel(vra):
    """vr_export_choices / vr_import_choices (round 6): what one handle measured, a new handle -- here standing in for a new
    process -- starts on: frame 1 is the settled kernel, no frame is a trial (bit 8 of vr_get_launch_choice), and the frames
    are the same bits.  A blob from another build / device is well-formed but takes over nothing; a mangled one is refused.
    The reference never tries anything: one dispatch per frame (src/RendererCore.cpp:138-163)."""
    R = vra.renderer

    def fresh():
        r = vra.RendererCore(0)
        r.setup((1920, 1080)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (512, 512, 512), 2, 0x9E3779B9); r.setWindow(0, 4095); r.setAlpha(0.004)
        r.setRowStripes(16, 3, 8)                               # a sparse shard: relay against fast kernel, a real choice
        return r

    with fresh() as a:
        for _ in range(300):
            a.render()
        settled = a.last_launch_choice
        assert settled & 256 == 0
        want = a.readPixels().copy()
        blob = a.exportChoices()
        assert blob[:8] == b"VRCHOICE" and len(blob) >= 88 + 56
    with fresh() as b:
        assert b.importChoices(blob) >= 1
        trials, choices = 0, set()
        for _ in range(40):
            b.render()
            trials += (b.last_launch_choice >> 8) & 1
            choices.add(b.last_launch_choice)
        assert trials == 0 and choices == {settled}, (trials, choices, settled)
        assert np.array_equal(b.readPixels().view(np.uint32), want.view(np.uint32))
        # another build's blob: same layout, other build id -> nothing taken over
        foreign = bytearray(blob); foreign[16] ^= 0x5a
        assert b.importChoices(bytes(foreign)) == 0
        with pytest.raises(vra.VRError):
            b.importChoices(b"VRCHOICX" + blob[8:])
        with pytest.raises(vra.VRError):
            b.importChoices(blob[:100])
    with fresh() as c:                                           # the control: without the blob the first frames do explore
        trials = 0
        for _ in range(12):
            c.render()
            trials += (c.last_launch_choice >> 8) & 1
        assert trials > 0
