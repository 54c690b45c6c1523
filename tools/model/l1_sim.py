#!/usr/bin/env python3
"""Model of one CU's vector L1 under the fast kernel's access stream (no GPU needed).

A CU runs G workgroups (32x16-pixel tiles, 8 wavefronts of 8x8 pixels, lockstep every
`sync` batches); every wavefront issues `batch` gathers back to back per batch; the L1 is an
LRU cache of `lines` 128-B lines (pending lines count as present).  Reports L1 misses per
wavefront-gather for a camera pose -- the quantity the launch time is proportional to
(DESIGN.md section 6).  usage: l1_sim.py [default|offaxis] [--wgs 3] [--batch 8] [--lines 256] ...
"""
import argparse
import math
from collections import OrderedDict

import numpy as np


def camera(pose):
    # reference default: eye (0,0,3), basis I; off-axis: orbit zenith 60 deg, azimuth 45 deg about the origin
    if pose == "default":
        eye = np.array([0, 0, 3.0]); side = np.array([1.0, 0, 0]); up = np.array([0, 1.0, 0]); look = np.array([0, 0, -1.0])
    else:
        zen, azi = math.radians(60), math.radians(45)
        eye = 3.0 * np.array([math.sin(zen) * math.sin(azi), math.cos(zen), math.sin(zen) * math.cos(azi)])
        look = -eye / np.linalg.norm(eye)
        side = np.cross(look, np.array([0, 1.0, 0])); side /= np.linalg.norm(side)
        up = np.cross(side, look)
    return eye, side, up, look


def tile_rays(pose, tx, ty, W=1920, H=1080):
    eye, side, up, look = camera(pose)
    d = 1.0 / math.tan(math.radians(15))
    px = tx * 32 + np.arange(32) + 0.5
    py = ty * 16 + np.arange(16) + 0.5
    X, Y = np.meshgrid(px, py)
    x = (W / H) * (2 * X / W - 1); y = 2 * Y / H - 1
    dirs = x[..., None] * side + y[..., None] * up + d * look
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    return eye, dirs            # [16, 32, 3]


def slab(eye, dirs):
    inv = 1.0 / dirs
    t0 = (-0.5 - eye) * inv; t1 = (0.5 - eye) * inv
    tmin = np.minimum(t0, t1).max(-1); tmax = np.maximum(t0, t1).min(-1)
    return tmin, tmax


def line_of(pos, N, brick, bytes_per_voxel):
    """pos in box units [-0.5,0.5]^3 -> cache-line id of the voxel (bricked, x-fastest bricks)"""
    v = np.clip(((pos + 0.5) * N).astype(np.int64), 0, N - 1)
    v[..., 2] = N - 1 - v[..., 2]
    bx, by, bz = brick
    b = (v[..., 0] // bx) + (N // bx) * ((v[..., 1] // by) + (N // by) * (v[..., 2] // bz))
    if bytes_per_voxel == 2:
        return b                                   # 64 voxels x 2 B = one line
    # 12-bit packing: brick = 96 B, contiguous
    inb = (v[..., 0] % bx) + bx * ((v[..., 1] % by) + by * (v[..., 2] % bz))
    return (b * 96 + (inb * 3) // 2) // 128


def simulate(pose, wgs, batch, lines, sync, N, brick, bpv, steps, seed, tiles=None):
    rng = np.random.default_rng(seed)
    # pick tiles whose centre ray crosses the box for a long distance
    cands = []
    for ty in range(68):
        for tx in range(60):
            eye, dirs = tile_rays(pose, tx, ty)
            tmin, tmax = slab(eye, dirs[8:9, 16:17])
            if (tmax - tmin)[0, 0] > 0.9:
                cands.append((tx, ty))
    chosen = tiles or [cands[i] for i in rng.choice(len(cands), size=wgs, replace=False)]
    state = []
    for (tx, ty) in chosen:
        eye, dirs = tile_rays(pose, tx, ty)
        tmin, _ = slab(eye, dirs)
        start = eye + dirs * tmin[..., None]
        state.append((start, dirs / N))
    cache = OrderedDict()
    misses = loads = 0
    nb = steps // batch
    for b in range(nb):
        # all workgroups, all waves, interleaved per load instruction (round-robin issue)
        for u in range(batch):
            k = b * batch + u
            for (start, dstep) in state:
                pos = start + dstep * k
                ln = line_of(pos, N, brick, bpv)           # [16, 32]
                for wy in range(2):
                    for wx in range(4):
                        uniq = np.unique(ln[wy * 8:(wy + 1) * 8, wx * 8:(wx + 1) * 8])
                        loads += 1
                        for l in uniq.tolist():
                            if l in cache:
                                cache.move_to_end(l)
                            else:
                                misses += 1
                                cache[l] = True
                                if len(cache) > lines:
                                    cache.popitem(last=False)
    return misses / loads


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("pose", nargs="?", default="default")
    ap.add_argument("--wgs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lines", type=int, default=256)
    ap.add_argument("--sync", type=int, default=2)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--brick", type=int, nargs=3, default=(4, 4, 4))
    ap.add_argument("--bpv", type=float, default=2)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    m = simulate(a.pose, a.wgs, a.batch, a.lines, a.sync, a.n, tuple(a.brick), 2 if a.bpv == 2 else 1.5, a.steps, a.seed)
    print(f"{a.pose}: wgs {a.wgs} batch {a.batch} L1 lines {a.lines} brick {tuple(a.brick)} bytes/voxel {a.bpv}: {m:.2f} L1 misses per wavefront-gather")
