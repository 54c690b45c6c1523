"""Image-row sharding of one frame across the GPUs of a node.

Pixels are independent, so the frame is split by rows with the volume replicated:
  * "contiguous": rank k renders rows [k*ceil(H/G), ...)  (north-star default shape)
  * "stripes":    cyclic stripes of `stripe_rows` rows: stripe s belongs to rank s % G
                  (balanced: at the default camera only ~75 % of the rows hit the box,
                  SURVEY F7)
Every rank renders into a COMPACT local target of `local_rows` rows (equal on all ranks,
padded), one collective over RCCL moves the shards -- a gather to the root rank (default in
bench.py: only the root needs the frame) or an all_gather -- and an index_select undoes the
interleave.  Pure index logic + torch.distributed: runs on gloo/CPU in the tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class RowPlan:
    img_h: int
    world: int
    rank: int
    mode: str            # "contiguous" | "stripes"
    stripe_rows: int
    local_rows: int      # rows of the (padded) compact local target, equal on all ranks

    @property
    def row_range(self):
        """(begin, end) global rows for mode == contiguous"""
        b = min(self.rank * self.local_rows, self.img_h)
        return b, min(b + self.local_rows, self.img_h)

    def global_rows(self, rank: int | None = None) -> np.ndarray:
        """global row of every local row of `rank` (-1 = padding)"""
        k = self.rank if rank is None else rank
        loc = np.arange(self.local_rows)
        if self.mode == "contiguous":
            g = k * self.local_rows + loc
        else:
            s, r = loc // self.stripe_rows, loc % self.stripe_rows
            g = (s * self.world + k) * self.stripe_rows + r
        return np.where(g < self.img_h, g, -1)


def plan_rows(img_h: int, world: int, rank: int, mode: str = "stripes", stripe_rows: int = 16) -> RowPlan:
    if mode not in ("contiguous", "stripes"):
        raise ValueError(mode)
    if world == 1:
        return RowPlan(img_h, 1, 0, "contiguous", stripe_rows, img_h)
    if mode == "contiguous":
        local = -(-img_h // world)
    else:
        nstripes = -(-img_h // stripe_rows)
        local = -(-nstripes // world) * stripe_rows
    return RowPlan(img_h, world, rank, mode, stripe_rows, local)


def apply_plan(renderer, plan: RowPlan) -> None:
    """configure a RendererCore handle for its shard (C ABI: vr_set_row_range / _stripes)"""
    if plan.world == 1:
        renderer.setRowRange(0, -1)
        renderer.setRowStripes(1, 0, 1)
    elif plan.mode == "contiguous":
        renderer.setRowStripes(1, 0, 1)
        renderer.setRowRange(*plan.row_range)
    else:
        renderer.setRowRange(0, -1)
        renderer.setRowStripes(plan.stripe_rows, plan.rank, plan.world)


def gather_index(plan: RowPlan):
    """index[g] = position of global row g in the rank-major gathered buffer
    (world * local_rows rows)"""
    idx = np.full(plan.img_h, -1, dtype=np.int64)
    for k in range(plan.world):
        g = plan.global_rows(k)
        ok = g >= 0
        idx[g[ok]] = k * plan.local_rows + np.nonzero(ok)[0]
    assert (idx >= 0).all()
    return idx


_GA_SELECT = {}
_GATHER_VIEWS = {}


def expand_grey_alpha(ga):
    """[..., 2] (grey, alpha) -> [..., 4] (grey, grey, grey, alpha): the RGBA32F frame of the
    reference's grey modes, in which r == g == b bit for bit"""
    import torch

    sel = _GA_SELECT.get(ga.device)
    if sel is None:             # once per device: building it per frame is a blocking host-to-device copy
        sel = _GA_SELECT[ga.device] = torch.tensor([0, 0, 0, 1], device=ga.device)
    return ga.index_select(-1, sel)


def gather_frame(local, plan: RowPlan, out=None, index=None, root=None, assembler=None, frame_out=None, group=None, host_staged=None):
    """gather the compact shards and return the assembled [H, W, 4] frame.

    root = None: all_gather -- every rank ends up with the frame.  root = k: a GATHER to rank k (RCCL
    implements it as grouped send/recv: the root's point-to-point xGMI links to its peers work
    concurrently, and 1/world of an all_gather's bytes move); rank k returns the frame, the others None.

    local: torch tensor [local_rows, W, C] on this rank's device, C = 4 (RGBA32F) or C = 2
    ((grey, alpha) targets, vr_set_framebuffer_format: half the bytes on the wire; expanded to
    RGBA after the gather).  One collective (RCCL all_gather over xGMI on GPUs) + one
    index_select to undo the interleave.

    assembler = a RendererCore on the same device + frame_out = a preallocated [H, W, 4] float32 tensor: the
    de-interleave and the (grey, alpha) expansion are done by ONE kernel of the C ABI (vr_assemble_shards) on the
    current torch stream instead of two index_select launches.

    group = the process group of the collective (None = the default group).  host_staged = True moves the shards
    through host memory (device -> host, the collective on CPU tensors, host -> device): what a gloo group needs,
    and bench.py's fallback transport when the RCCL preflight fails; None = decide from the group's backend.
    """
    import torch
    import torch.distributed as dist

    grey_alpha = local.shape[-1] == 2
    if plan.world == 1:
        frame = local[: plan.img_h]
        return expand_grey_alpha(frame) if grey_alpha else frame
    caller_owns_out = out is not None
    if out is None:
        out = torch.empty((plan.world * plan.local_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if root is not None:
        is_root = plan.rank == root
        staged = local.is_cuda and (dist.get_backend(group) == "gloo" if host_staged is None else bool(host_staged))   # gloo: through host memory
        if is_root:
            dst = torch.empty(out.shape, dtype=out.dtype) if staged else out
            # the per-rank views of a CALLER-OWNED gather buffer are built once, not per frame (a buffer allocated
            # above lives for this call only: caching its views would pin it)
            cacheable = caller_owns_out and not staged
            key = (dst.data_ptr(), tuple(dst.shape), plan.world, str(dst.dtype), str(dst.device))
            pieces = _GATHER_VIEWS.get(key) if cacheable else None
            if pieces is None:
                pieces = list(dst.view((plan.world, plan.local_rows) + tuple(local.shape[1:])).unbind(0))
                if cacheable:
                    if len(_GATHER_VIEWS) > 16:
                        _GATHER_VIEWS.clear()
                    _GATHER_VIEWS[key] = pieces
            dist.gather(local.cpu() if staged else local, gather_list=pieces, dst=root, group=group)
            if staged:
                out.copy_(dst)
        else:
            dist.gather(local.cpu() if staged else local, dst=root, group=group)
            return None
    elif local.is_cuda and (dist.get_backend(group) == "gloo" if host_staged is None else bool(host_staged)):
        # gloo (validation: several ranks sharing one GPU; bench.py's fallback transport): through host memory
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local, group=group)
    if assembler is not None and frame_out is not None and out.is_cuda:
        assembler.assembleShards(out.data_ptr(), frame_out.data_ptr(), plan.world, plan.local_rows,
                                 0 if plan.mode == "contiguous" else plan.stripe_rows, local.shape[-1],
                                 torch.cuda.current_stream(out.device).cuda_stream)
        return frame_out
    if plan.mode == "contiguous":
        frame = out[: plan.img_h]
    else:
        if index is None:
            index = torch.as_tensor(gather_index(plan), device=local.device)
        frame = out.index_select(0, index)
    return expand_grey_alpha(frame) if grey_alpha else frame
