"""Multi-rank image-row sharding on CPU: world_size 2 over gloo.

Each rank renders ITS rows (with the CPU oracle standing in for the device kernel: this
test covers the partition / all_gather / de-interleave logic of
volume-renderer_amd/sharding.py that bench.py runs over RCCL) into a compact local target;
the gathered frame must equal the single-rank frame bit for bit.
"""
import importlib
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, stripe_rows, W, H, out_dir, grey_alpha=False, root=None):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    sharding = importlib.import_module("volume-renderer_amd.sharding")

    vol = oracle.gen_noise_ball((24, 20, 28), 1, 77)
    plan = sharding.plan_rows(H, world, rank, mode, stripe_rows)
    rows = plan.global_rows()
    local = np.zeros((plan.local_rows, W, 4), dtype=np.float32)
    p = oracle.OracleParams(W, H, alpha_scale=0.05)
    full = np.zeros((H, W, 4), dtype=np.float32)
    for lr, g in enumerate(rows):
        if g < 0:
            continue
        p.row_begin, p.row_end = int(g), int(g) + 1
        oracle.render(vol, p, out=full)
        local[lr] = full[g]
    if grey_alpha:      # what vr_set_framebuffer_format(VR_FB_GREYALPHA32F) makes the kernel store: (grey, alpha)
        local = np.ascontiguousarray(local[..., [0, 3]])
    frame = sharding.gather_frame(torch.from_numpy(local), plan, root=root)
    assert (frame is None) == (root is not None and rank != root)
    if frame is not None:
        np.save(os.path.join(out_dir, f"frame_{mode}_{rank}.npy"), frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,stripe_rows,grey_alpha", [("contiguous", 16, False), ("stripes", 8, False), ("stripes", 16, False),
                                                         ("stripes", 16, True), ("contiguous", 16, True)])
def test_two_rank_gather_reassembles_the_frame(mode, stripe_rows, grey_alpha, tmp_path, oracle):
    import torch.multiprocessing as mp

    W, H, world = 40, 53, 2          # ragged: 53 rows
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, stripe_rows, W, H, str(tmp_path), grey_alpha), nprocs=world, join=True)
    vol = oracle.gen_noise_ball((24, 20, 28), 1, 77)
    want, _ = oracle.render(vol, oracle.OracleParams(W, H, alpha_scale=0.05))
    for rank in range(world):
        got = np.load(tmp_path / f"frame_{mode}_{rank}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (mode, rank)


def test_row_plans_cover_every_row_exactly_once():
    sharding = importlib.import_module("volume-renderer_amd.sharding")
    for H in (1, 15, 16, 53, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            for mode, sr in (("contiguous", 16), ("stripes", 8), ("stripes", 16)):
                seen = np.zeros(H, dtype=int)
                plans = [sharding.plan_rows(H, world, k, mode, sr) for k in range(world)]
                assert len({p.local_rows for p in plans}) == 1          # equal all_gather counts
                for p in plans:
                    g = p.global_rows()
                    seen[g[g >= 0]] += 1
                assert (seen == 1).all(), (H, world, mode)
                idx = sharding.gather_index(plans[0])
                assert sorted(idx.tolist()) == sorted(set(idx.tolist())) and len(idx) == H


def test_default_camera_stripes_balance_better_than_blocks():
    """contiguous 135-row blocks leave ranks 0 and 7 idle at the default camera"""
    sharding = importlib.import_module("volume-renderer_amd.sharding")
    H = 1080
    work = np.zeros(H)
    work[137:943] = 1.0
    for mode, expect_max in (("contiguous", 135), ("stripes", 112)):
        per_rank = []
        for k in range(8):
            g = sharding.plan_rows(H, 8, k, mode, 16).global_rows()
            per_rank.append(work[g[g >= 0]].sum())
        assert max(per_rank) <= expect_max
        if mode == "contiguous":
            assert min(per_rank) == 0
        else:
            assert min(per_rank) >= 96


@pytest.mark.parametrize("mode,stripe_rows,grey_alpha,world", [("stripes", 16, True, 2), ("contiguous", 16, False, 3), ("stripes", 8, False, 3)])
def test_gather_to_root_reassembles_the_frame_on_the_root_only(mode, stripe_rows, grey_alpha, world, tmp_path, oracle):
    """the default collective of bench.py --gpus N: dist.gather to rank 0 (RCCL send/recv), other ranks get None"""
    import torch.multiprocessing as mp

    W, H = 40, 53
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, stripe_rows, W, H, str(tmp_path), grey_alpha, 0), nprocs=world, join=True)
    vol = oracle.gen_noise_ball((24, 20, 28), 1, 77)
    want, _ = oracle.render(vol, oracle.OracleParams(W, H, alpha_scale=0.05))
    got = np.load(tmp_path / f"frame_{mode}_0.npy")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert not any((tmp_path / f"frame_{mode}_{k}.npy").exists() for k in range(1, world))
