"""Device memory of the optional speed copies is observable and bounded (round-4 verdict weak item 8, advisor item 1):
vr_get_resident_bytes, vr_set_copy_budget; frames are bit-identical whatever the budget allows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def frame_of(r):
    r.render()
    return r.readPixels().copy()


def test_copies_are_reported_bounded_and_invisible(vra, oracle):
    R = vra.renderer
    N = 128
    with vra.RendererCore(0) as r:
        r.setup((320, 240)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 2, 5); r.setWindow(0, 4095); r.setAlpha(0.05)
        vol_bytes = N ** 3 * 2
        v, k, o = r.residentBytes()
        assert v >= vol_bytes and k == 0 and o >= 320 * 240 * 16           # nothing optional before the first frame
        # NEAREST: the 12-bit packed copy (0.75 volumes)
        nearest = frame_of(r)
        v, k, o = r.residentBytes()
        assert r.pack12Bytes() > 0 and vol_bytes * 3 // 4 <= k <= vol_bytes * 3 // 4 + 4096
        # TRILINEAR at an oblique pose: + the apron copy and the two per-axis ones (3 x 1.25 volumes)
        r.setFilter(R.FILTER_TRILINEAR); r.cameraOrient(0.0, 0.66, -1.65)
        tri = frame_of(r)
        k_all = r.residentBytes()[1]
        assert k_all >= vol_bytes * 3 // 4 + int(1.25 * vol_bytes)
        assert r.trilinearCopyBytes() > 0
        # budget 0: every copy is freed, nothing is built, frames do not change by one bit
        r.setCopyBudget(0)
        assert r.residentBytes()[1] == 0
        assert np.array_equal(frame_of(r).view(np.uint32), tri.view(np.uint32))
        assert r.residentBytes()[1] == 0 and r.trilinearCopyBytes() == 0
        r.setFilter(R.FILTER_NEAREST); r.resetCamera()
        assert np.array_equal(frame_of(r).view(np.uint32), nearest.view(np.uint32))
        assert r.pack12Bytes() == 0 and r.residentBytes()[1] == 0
        # room for the packed copy only
        r.setCopyBudget(vol_bytes)
        assert np.array_equal(frame_of(r).view(np.uint32), nearest.view(np.uint32))
        assert r.pack12Bytes() > 0
        r.setFilter(R.FILTER_TRILINEAR); r.cameraOrient(0.0, 0.66, -1.65)
        assert np.array_equal(frame_of(r).view(np.uint32), tri.view(np.uint32))
        assert r.residentBytes()[1] <= vol_bytes                              # the apron did not fit next to it
        # back to automatic: the copies return
        r.setCopyBudget(r.COPY_BUDGET_AUTO)
        assert np.array_equal(frame_of(r).view(np.uint32), tri.view(np.uint32))
        assert r.residentBytes()[1] >= k_all - 4096
        # a new volume drops everything that belonged to the old one
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (64, 64, 64), 2, 5)
        assert r.residentBytes()[1] == 0


def test_per_axis_copies_are_not_built_when_no_launch_can_use_them(vra):
    """advisor (round 4): the two per-axis apron copies (2 x 1.25 volumes) only when a half-layer shape can be chosen:
    not with the measured choice off and a whole-layer first guess, not for forced whole-layer variants"""
    R = vra.renderer
    N = 96
    vol_bytes = N ** 3 * 2
    with vra.RendererCore(0) as r:
        r.setup((256, 192)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (N, N, N), 2, 9); r.setWindow(0, 4095); r.setAlpha(0.05)
        r.setFilter(R.FILTER_TRILINEAR); r.setPack12(False)
        r.setKernelVariant(6)                                                  # whole layers, forced
        r.cameraOrient(0.0, 0.66, -1.65)
        a = frame_of(r)
        assert r.residentBytes()[1] < 2 * vol_bytes                            # one apron copy (1.25 volumes), not three
        r.setKernelVariant(8)                                                  # half layers, forced: now they are needed
        b = frame_of(r)
        assert r.residentBytes()[1] > 3 * vol_bytes
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_trial_frames_of_the_measured_choice_are_flagged(vra):
    """vr_get_launch_choice bit 8: a caller can tell a trial frame from the settled kernel"""
    R = vra.renderer
    with vra.RendererCore(0) as r:
        r.setup((640, 480)); assert r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
        r.generateSynthetic(R.SYNTH_NOISE_BALL, (256, 256, 256), 2, 1); r.setWindow(0, 4095); r.setAlpha(0.01)
        flags = []
        for _ in range(60):
            r.render()
            flags.append(r.last_launch_choice)
        assert any(f & 256 for f in flags[:20]), flags[:20]                   # it explored ...
        assert not any(f & 256 for f in flags[-10:]), flags[-10:]             # ... and settled
        assert len({f for f in flags[-10:]}) == 1
        r.setAutotune(False)
        r.render()
        assert not (r.last_launch_choice & 256)
