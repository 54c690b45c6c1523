"""vr_group_*: the native (single-process) multi-GPU path of the C ABI.  A one-GPU box can only
put several members on the same device -- that exercises the shard plan, the compact (grey, alpha)
/ RGBA shard targets, the gather buffers and the assembly kernel with the peer-copy transport; the
RCCL transport is initialised and used for real with a one-member group (ncclCommInitAll over one
device is legal) and on a multi-GPU node whenever the devices are distinct."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def configure(r, vra, vol, tf):
    assert r.loadShader("VolumeRenderer.cs")
    r.setQuirks(0)
    r.setVolume(vol)
    r.setWindow(5, 240)
    r.setAlpha(0.05)
    r.cameraOrient(0, 0.06 * 5, 0.06 * 11)
    if tf:
        r.setTransferFunction([0, 90, 160, 255], [[0, 0, 0, 0], [0.9, 0.2, 0.1, 0.3], [0.2, 0.8, 0.3, 0.1], [1, 1, 1, 0.9]])


@pytest.mark.parametrize("tf", [False, True], ids=["grey_float2_shards", "tf_rgba_shards"])
@pytest.mark.parametrize("partition,stripe_rows", [("stripes", 16), ("stripes", 5), ("contiguous", 16)])
@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_group_frame_equals_single_device_frame(vra, n, partition, stripe_rows, tf):
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 256, size=(48, 40, 56), dtype=np.uint8)
    size = (203, 157)                                   # neither a multiple of the stripe height nor of n
    with vra.RendererCore(0) as single:
        single.setup(size)
        configure(single, vra, vol, tf)
        single.render()
        want = single.readPixels()
    with vra.RendererGroup([0] * n) as g:
        g.setup(size, partition=partition, stripe_rows=stripe_rows)
        g.each(lambda m: configure(m, vra, vol, tf))
        g.render()
        got = g.readPixels()
        assert g.kernelMsTake() > 0.0
        assert ("peer" in g.transport.lower()) == (n > 1)
        g.render()                                      # a second frame re-uses every buffer
        again = g.readPixels()
    assert np.array_equal(bits(got), bits(want)), (n, partition, stripe_rows, tf)
    assert np.array_equal(bits(again), bits(want))


def test_rccl_loads_and_initialises_on_this_box(vra):
    """transport mode 2: a one-member group still creates its RCCL communicator (dlopen of librccl.so,
    ncclCommInitAll over one device) -- what a multi-GPU node does, minus the peers"""
    with vra.RendererGroup([0]) as g:
        g.setTransport(2)
        g.setup((64, 48))
        assert g.transport.startswith("rccl"), g.transport
    with vra.RendererGroup([0]) as g:
        g.setup((64, 48))
        assert g.transport.startswith("none")           # default: nothing to gather, RCCL stays unloaded
    with vra.RendererGroup([0, 0]) as g:                # duplicates can never form a communicator: peer copies
        g.setup((64, 48))
        assert "peer" in g.transport.lower()


@pytest.mark.parametrize("tf", [False, True], ids=["grey_float2_shard", "tf_rgba_shard"])
def test_rccl_self_send_recv_moves_the_shard(vra, tf):
    """transport mode 2 on a one-member group: the shard travels through a grouped ncclSend + ncclRecv to
    self before the assembly kernel -- g_rccl.Send / Recv, ncclFloat32, the element count and the stream wiring
    of the multi-GPU gather execute for real, and the frame must still be the single-device frame bit for bit"""
    rng = np.random.default_rng(11)
    vol = rng.integers(0, 256, size=(48, 40, 56), dtype=np.uint8)
    size = (203, 157)
    poses = [(0.06 * 5, 0.06 * 11), (0.06 * 2, -0.06 * 7), (-0.06 * 9, 0.06 * 3), (0.06, 0.06)]
    want = []
    with vra.RendererCore(0) as single:
        single.setup(size)
        configure(single, vra, vol, tf)
        for ze, az in poses:
            single.resetCamera(); single.cameraOrient(0, ze, az)
            single.render()
            want.append(single.readPixels())
    assert not np.array_equal(bits(want[0]), bits(want[2]))
    with vra.RendererGroup([0]) as g:
        g.setTransport(2)
        g.setup(size)
        assert "self send/recv" in g.transport, g.transport
        g.each(lambda m: configure(m, vra, vol, tf))
        for k, (ze, az) in enumerate(poses):            # both frame slots, each re-used; a different frame every time: stale bytes show
            g.each(lambda m: (m.resetCamera(), m.cameraOrient(0, ze, az)))
            g.render()
            assert np.array_equal(bits(g.readPixels()), bits(want[k])), k


@pytest.mark.parametrize("n", [1, 2, 4])
def test_group_async_pipeline_two_frames_in_flight(vra, n):
    """vr_group_render_async / vr_group_wait: two frame slots; the frame returned by wait() is the OLDEST in
    flight; a third render_async without a wait is refused; the camera may change between frames"""
    rng = np.random.default_rng(5)
    vol = rng.integers(0, 256, size=(48, 40, 56), dtype=np.uint8)
    size = (203, 157)
    poses = [(0.06 * 5, 0.06 * 11), (0.06 * 2, -0.06 * 7), (-0.06 * 9, 0.06 * 3), (0.0, 0.0), (0.06, 0.06)]
    want = []
    with vra.RendererCore(0) as single:
        single.setup(size)
        configure(single, vra, vol, False)
        for ze, az in poses:
            single.resetCamera(); single.cameraOrient(0, ze, az)
            single.render()
            want.append(single.readPixels())
    with vra.RendererGroup([0] * n) as g:
        g.setup(size, partition="stripes", stripe_rows=8)
        g.each(lambda m: configure(m, vra, vol, False))
        with pytest.raises(vra.VRError):
            g.wait()                                    # nothing in flight
        def enqueue(k):
            ze, az = poses[k]
            g.each(lambda m: (m.resetCamera(), m.cameraOrient(0, ze, az)))
            g.renderAsync()
        enqueue(0); enqueue(1)
        with pytest.raises(vra.VRError):
            g.renderAsync()                             # two frames in flight already
        for k in range(len(poses)):
            g.wait()
            assert np.array_equal(bits(g.readPixels()), bits(want[k])), (n, k)
            if k + 2 < len(poses):
                enqueue(k + 2)
        assert g.kernelMsTake() > 0.0
        g.render()                                      # the blocking call still works afterwards
        assert np.array_equal(bits(g.readPixels()), bits(want[-1]))


def test_member_is_detached_after_a_new_setup(vra):
    """vr_group_setup frees the previous shard targets: a member rendered directly (outside the group) must then
    write to its own framebuffer, never to the freed one (round-2 advisor finding)"""
    rng = np.random.default_rng(9)
    vol = rng.integers(0, 256, size=(32, 32, 32), dtype=np.uint8)
    with vra.RendererCore(0) as single:
        single.setup((96, 64))
        configure(single, vra, vol, False)
        single.render()
        want = single.readPixels()
    with vra.RendererGroup([0, 0]) as g:
        g.setup((96, 64))
        g.each(lambda m: configure(m, vra, vol, False))
        g.render()
        with pytest.raises(vra.VRError):
            g.setup((96, 64), stripe_rows=0)            # refused before anything is touched
        with pytest.raises(vra.VRError):
            g.setup((-1, 64))
        g.setup((96, 64), partition="contiguous")       # re-plan: old targets are gone, members re-attached
        g.render()
        assert np.array_equal(bits(g.readPixels()), bits(want))
