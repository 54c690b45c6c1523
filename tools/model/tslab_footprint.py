#!/usr/bin/env python3
"""CPU model of the LDS-staged TRILINEAR kernel's per-tile load plan (vr_tslab.hip): for every 32x16-pixel tile of a
pose, the torus size RA x RB its brick layers need
  * as bounding RECTANGLES of the four corner rays' crossings (round 3's plan), and
  * as SHEARED rows: rows along the minor axis b in which the tile's long edge advances fastest, each row holding the
    bricks a in [a_base(L) + sh(b), ... + RA) with sh(b) = floor(s * b): the footprint of a rotated tile is a
    parallelogram, and planes parallel to a layer cut the tile's ray pyramid in homothetic figures, so one slope s
    per tile serves every layer.
Prints how many tiles fit the ring (3 layers deep) under either plan.
  tools/model/tslab_footprint.py [default|offaxis|randK] [N] [bytes] [W H]"""
import importlib, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
vra = importlib.import_module("volume-renderer_amd")

pose = sys.argv[1] if len(sys.argv) > 1 else "offaxis"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bpv = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ((1920, 1080) if N <= 1024 else (3840, 2160))
r = vra.RendererCore(-1)
if pose == "offaxis":
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
elif pose.startswith("rand"):
    rng = np.random.default_rng(int(pose[4:] or 0))
    r.cameraOrient(0.0, float(rng.uniform(-1, 1)), float(rng.uniform(-3, 3)))
c = r.getCameraBlock().astype(np.float64)

SLOT = 80 * bpv
REGION = (80 * 1024 - 512 - 16 - 512) // 16 * 16
tail = (N // 4 * 8) + 2 * (N + 1) * 2 + (N + 1) * 4
slots_avail = (REGION - tail) // SLOT
LAYER_MAX = 3 * 8 * 64 // (SLOT // 16)
print(f"{pose} {N}^3 x {bpv} B, {W}x{H}: {slots_avail} slots, <= {min(LAYER_MAX, slots_avail // 3)} per layer")


def ray_dirs(px, py):
    aspect = W / H
    x = aspect * (2 * px / W - 1); y = 2 * py / H - 1; z = -c[20] * np.ones_like(x)
    ln = np.sqrt(x * x + y * y + z * z)
    dx, dy, dz = x / ln, y / ln, z / ln
    m = np.stack([c[0] * dx + c[4] * dy + c[8] * dz, c[1] * dx + c[5] * dy + c[9] * dz, c[2] * dx + c[6] * dy + c[10] * dz], -1)
    return m / np.linalg.norm(m, axis=-1, keepdims=True)


def to_voxel(p):   # box position -> voxel coordinates (view 0: tc = (u, v, 1 - w) * N), cubic volume, unit box
    u = p + 0.5
    return np.stack([u[..., 0], u[..., 1], 1 - u[..., 2]], -1) * N


eye = c[16:19]
E = to_voxel(eye)
tiles_x, tiles_y = (W + 31) // 32, (H + 15) // 16
res = []
for ty in range(tiles_y):
    for tx in range(tiles_x):
        cx = np.array([tx * 32 + 0.5, tx * 32 + 31.5, tx * 32 + 0.5, tx * 32 + 31.5])
        cy = np.array([ty * 16 + 0.5, ty * 16 + 0.5, ty * 16 + 15.5, ty * 16 + 15.5])
        d = ray_dirs(cx, cy)
        G = to_voxel(eye + d) - E                                    # 4 x 3
        # does the tile hit the box at all (corner rays only: a model)
        with np.errstate(all="ignore"):
            t0 = (0 - E) / G; t1 = (N - E) / G
            tmin = np.max(np.minimum(t0, t1), axis=1); tmax = np.min(np.maximum(t0, t1), axis=1)
        if not np.any(tmax > np.maximum(tmin, 0)):
            continue
        g0 = np.abs(G[0]); m = int(np.argmax(g0))
        gm = G[:, m]; gmax = np.abs(G).max(axis=1)
        if not (np.all(gm >= 0.3 * gmax) or np.all(-gm >= 0.3 * gmax)):
            res.append((0, 0, 0, 0, 0, 0, 0)); continue
        a_ax, b_ax = (1 if m == 0 else 0), (1 if m == 2 else 2)
        tin, tout = np.max(np.where(tmax > tmin, tmin, -np.inf)), np.max(np.where(tmax > tmin, tmax, -np.inf))
        fm_in = np.clip(E[m] + G[:, m] * np.maximum(tmin, 0), 0, N); fm_out = np.clip(E[m] + G[:, m] * tmax, 0, N)
        l_lo, l_hi = int(min(fm_in.min(), fm_out.min())) >> 2, min(int(max(fm_in.max(), fm_out.max())) >> 2, N // 4 - 1)
        delta = 0.0625 + 1100 * 1.2e-7 * N + np.abs(E).max() * 2.4e-7
        ra_r = rb_r = ra_s = rb_s = 0
        # shear: the tile's long edge (corner 0 -> 1); rows along whichever minor axis it advances in faster
        Ls = np.arange(l_lo, l_hi + 1)
        pts_a, pts_b = [], []
        for cl in (4.0 * Ls - 0.5 - delta, 4.0 * Ls + 4.5 + delta):
            t = (cl[:, None] - E[m]) / G[None, :, m]                 # layers x 4
            pts_a.append(E[a_ax] + t * G[None, :, a_ax]); pts_b.append(E[b_ax] + t * G[None, :, b_ax])
        A = np.concatenate(pts_a, axis=1); B = np.concatenate(pts_b, axis=1)        # layers x 8
        lo_a = np.clip(np.floor(A.min(1) - 0.5 - delta).astype(int) >> 2, 0, N // 4 - 1); hi_a = np.clip(np.floor(A.max(1) + 0.5 + delta).astype(int) >> 2, 0, N // 4 - 1)
        lo_b = np.clip(np.floor(B.min(1) - 0.5 - delta).astype(int) >> 2, 0, N // 4 - 1); hi_b = np.clip(np.floor(B.max(1) + 0.5 + delta).astype(int) >> 2, 0, N // 4 - 1)
        ra_r, rb_r = int((hi_a - lo_a).max()) + 1, int((hi_b - lo_b).max()) + 1
        # sheared: edge vector at the middle layer
        mid = len(Ls) // 2
        ua, ub = A[mid, 1] - A[mid, 0], B[mid, 1] - B[mid, 0]
        if abs(ub) >= abs(ua):
            P_, Q_, s = B, A, ua / ub          # rows along b (p = b), sheared coordinate a
        else:
            P_, Q_, s = A, B, ub / ua
        # brick rows p; sheared coordinate in BRICKS: q_brick - floor(s * p_brick) must lie in [base(L), base(L) + RA)
        lo_p = np.clip(np.floor(P_.min(1) - 0.5 - delta).astype(int) >> 2, 0, N // 4 - 1); hi_p = np.clip(np.floor(P_.max(1) + 0.5 + delta).astype(int) >> 2, 0, N // 4 - 1)
        C = Q_ - s * P_                                              # voxels, layers x 8
        # a tap at (p, q) with q - s p in [cmin, cmax] (+- margins); its brick: (p >> 2, q >> 2); q_brick - floor(s * p_brick):
        # q/4 - s*p/4 in [cmin/4, cmax/4]; floors add at most: q_brick in (q/4 - 1, q/4]; s*p_brick within |s|*(3/4) of s*p/4; floor: 1
        lo_c = np.floor((C.min(1) - (0.5 + delta) * (1 + abs(s))) / 4 - abs(s) * 0.75 - 1).astype(int)
        hi_c = np.floor((C.max(1) + (0.5 + delta) * (1 + abs(s))) / 4 + abs(s) * 0.75 + 1).astype(int)
        ra_s, rb_s = int((hi_c - lo_c).max()) + 1, int((hi_p - lo_p).max()) + 1
        # per-row extents: hull of the frustum slab's projection = its 12 edges (4 per plane + 4 along the corner rays)
        pairs = [(0, 1), (1, 3), (3, 2), (2, 0), (4, 5), (5, 7), (7, 6), (6, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
        ra_t = rb_t = 0
        for li in range(0, len(Ls), max(1, len(Ls) // 24)):          # a sample of the layers (the far ones are the largest)
            a8, b8 = A[li], B[li]
            mrg = 0.5 + delta
            rows = range(int(np.floor(b8.min() - mrg)) >> 2, (int(np.floor(b8.max() + mrg)) >> 2) + 1)
            wmax = 0
            for rb in rows:
                y0, y1 = 4 * rb - mrg, 4 * rb + 4 + mrg            # taps of this brick row come from positions within the margin of it
                lo, hi = np.inf, -np.inf
                for (u, v) in pairs:
                    bu, bv, au, av = b8[u], b8[v], a8[u], a8[v]
                    t0, t1 = 0.0, 1.0
                    if bu == bv:
                        if not (y0 <= bu <= y1): continue
                    else:
                        ta, tb = (y0 - bu) / (bv - bu), (y1 - bu) / (bv - bu)
                        t0, t1 = max(0.0, min(ta, tb)), min(1.0, max(ta, tb))
                        if t0 > t1: continue
                    for t in (t0, t1):
                        a = au + t * (av - au); lo = min(lo, a); hi = max(hi, a)
                if hi < lo: continue
                w = (int(np.floor(hi + mrg)) >> 2) - (int(np.floor(lo - mrg)) >> 2) + 1
                wmax = max(wmax, w)
            ra_t, rb_t = max(ra_t, wmax), max(rb_t, len(rows))
        res.append((1, ra_r, rb_r, ra_s, rb_s, ra_t, rb_t))
res = np.array(res)
ok = res[:, 0] == 1
rect = res[ok, 1] * res[ok, 2]; shear = res[ok, 3] * res[ok, 4]
lim = min(LAYER_MAX, slots_avail // 3)
print(f"tiles hitting the box {len(res)}, with an agreed major axis {ok.sum()}")
print(f"rectangles: slots percentiles 10/50/90/100 {np.percentile(rect, [10, 50, 90, 100])}; fit {np.mean(rect <= lim):.3f}")
print(f"sheared   : slots percentiles 10/50/90/100 {np.percentile(shear, [10, 50, 90, 100])}; fit {np.mean(shear <= lim):.3f}; best of both fit {np.mean(np.minimum(rect, shear) <= lim):.3f}")
tight = res[ok, 5] * res[ok, 6]
print(f"row-tight : slots percentiles 10/50/90/100 {np.percentile(tight, [10, 50, 90, 100])}; fit {np.mean(tight <= lim):.3f}; RA p50/max {np.percentile(res[ok, 5], 50)}/{res[ok, 5].max()}, RB p50/max {np.percentile(res[ok, 6], 50)}/{res[ok, 6].max()}")
