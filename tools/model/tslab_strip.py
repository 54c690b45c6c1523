#!/usr/bin/env python3
"""model: the ring as ROWS (along b) whose extents come from one slanted strip c1 <= n . (a, b) <= c2 intersected with the layer's
bounding rectangle -- n perpendicular to the best of the footprint's three directions (tile edge x, tile edge y, the sweep of the rays
across the layer).  Slots = rows x widest row.  Compared with the bounding rectangle and the exact hull rows."""
import importlib.util, sys
from pathlib import Path
import numpy as np
spec = importlib.util.spec_from_file_location("fp2", Path(__file__).resolve().parent / "tslab_footprint2.py")
fp2 = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fp2)
except SystemExit:
    pass
N, W, H = fp2.N, fp2.W, fp2.H


def strip_model(c, tile_w, tile_h, T):
    def ray_dirs(px, py):
        aspect = W / H
        x = aspect * (2 * px / W - 1); y = 2 * py / H - 1; z = -c[20] * np.ones_like(x)
        ln = np.sqrt(x * x + y * y + z * z)
        dx, dy, dz = x / ln, y / ln, z / ln
        m = np.stack([c[0] * dx + c[4] * dy + c[8] * dz, c[1] * dx + c[5] * dy + c[9] * dz, c[2] * dx + c[6] * dy + c[10] * dz], -1)
        return m / np.linalg.norm(m, axis=-1, keepdims=True)

    def to_voxel(p):
        u = p + 0.5
        return np.stack([u[..., 0], u[..., 1], 1 - u[..., 2]], -1) * N

    eye = c[16:19]; E = to_voxel(eye)
    out = []
    for ty in range((H + tile_h - 1) // tile_h):
        for tx in range((W + tile_w - 1) // tile_w):
            cx = np.array([tx * tile_w + 0.5, tx * tile_w + tile_w - 0.5] * 2)
            cy = np.array([ty * tile_h + 0.5] * 2 + [ty * tile_h + tile_h - 0.5] * 2)
            G = to_voxel(eye + ray_dirs(cx, cy)) - E
            with np.errstate(all="ignore"):
                t0 = (0 - E) / G; t1 = (N - E) / G
                tmin = np.max(np.minimum(t0, t1), axis=1); tmax = np.min(np.maximum(t0, t1), axis=1)
            hitc = tmax > np.maximum(tmin, 0)
            if not np.any(hitc):
                continue
            work = float(np.max(np.where(hitc, tmax - np.maximum(tmin, 0), 0)))
            m = int(np.argmax(np.abs(G[0])))
            gm = G[:, m]; gmax = np.abs(G).max(axis=1)
            if not (np.all(gm >= 0.3 * gmax) or np.all(-gm >= 0.3 * gmax)):
                out.append((work, 1e9, 1e9)); continue
            a_ax, b_ax = (1 if m == 0 else 0), (1 if m == 2 else 2)
            fm_in = np.clip(E[m] + G[:, m] * np.maximum(tmin, 0), 0, N)[hitc]; fm_out = np.clip(E[m] + G[:, m] * tmax, 0, N)[hitc]
            l_lo, l_hi = int(min(fm_in.min(), fm_out.min())) // T, min(int(max(fm_in.max(), fm_out.max())) // T, N // T - 1)
            delta = 0.0625 + 1100 * 1.2e-7 * N + np.abs(E).max() * 2.4e-7
            mrg = 0.5 + delta
            Ls = np.arange(l_lo, l_hi + 1)
            # candidate directions at the middle layer
            Lm = Ls[len(Ls) // 2]
            tm = (np.array([T * Lm - 0.5 - delta, T * Lm + T + 0.5 + delta])[:, None] - E[m]) / G[None, :, m]
            am = E[a_ax] + tm * G[None, :, a_ax]; bm = E[b_ax] + tm * G[None, :, b_ax]
            dirs = [(am[0, 1] - am[0, 0], bm[0, 1] - bm[0, 0]), (am[0, 2] - am[0, 0], bm[0, 2] - bm[0, 0]), (am[1, 0] - am[0, 0], bm[1, 0] - bm[0, 0])]
            bb = 0
            best = None
            for dd in dirs + [None]:
                da, db = dd if dd is not None else (None, None)
                tot = 0
                for L in Ls[:: max(1, len(Ls) // 12)]:
                    cl = np.array([T * L - 0.5 - delta, T * L + T + 0.5 + delta])
                    t = (cl[:, None] - E[m]) / G[None, :, m]
                    a8 = (E[a_ax] + t * G[None, :, a_ax]).ravel(); b8 = (E[b_ax] + t * G[None, :, b_ax]).ravel()
                    lo_a, hi_a = int(np.floor(a8.min() - mrg)) >> 2, int(np.floor(a8.max() + mrg)) >> 2
                    lo_b, hi_b = int(np.floor(b8.min() - mrg)) >> 2, int(np.floor(b8.max() + mrg)) >> 2
                    nrows = hi_b - lo_b + 1
                    if da is None:                       # bounding rectangle
                        tot = max(tot, (hi_a - lo_a + 1) * nrows); continue
                    na, nb = db, -da                     # normal of the direction
                    if abs(na) < 1e-9 * (abs(na) + abs(nb)) or abs(nb / na) > 8:   # rows nearly parallel to the strip: no use
                        tot = 1e9; break
                    c8 = na * a8 + nb * b8
                    c1, c2 = c8.min(), c8.max()
                    wmax = 0
                    for r in range(lo_b, hi_b + 1):
                        y0, y1 = 4 * r - mrg, 4 * r + 4 + mrg
                        cand = [(c1 - nb * y0) / na, (c1 - nb * y1) / na, (c2 - nb * y0) / na, (c2 - nb * y1) / na]
                        a_lo = max(min(cand), a8.min()); a_hi = min(max(cand), a8.max())
                        if a_hi < a_lo:
                            continue
                        wmax = max(wmax, (int(np.floor(a_hi + mrg)) >> 2) - (int(np.floor(a_lo - mrg)) >> 2) + 1)
                    tot = max(tot, wmax * nrows)
                if da is None:
                    bb = tot
                elif best is None or tot < best:
                    best = tot
            out.append((work, bb, min(best, bb)))
    return np.array(out)


P = fp2.poses()
lim = ((80 * 1024 - 512 - 16 - 512 - 12288) // 80) // 3
for tw, th in ((32, 16), (16, 32)):
    for pn in (sys.argv[1:] or ["offaxis", "ze-0.47az-1.33", "ze-1.19az+1.93"]):
        r = strip_model(P[pn], tw, th, 2)
        wk = r[:, 0] / r[:, 0].sum()
        print(f"{tw}x{th} {pn:16s} bbox p50 {np.percentile(r[:,1],50):5.0f} p90 {np.percentile(r[:,1],90):5.0f} fit({lim}) {float(np.sum(wk*(r[:,1]<=lim))):.2f} | strip rows p50 {np.percentile(r[:,2],50):5.0f} p90 {np.percentile(r[:,2],90):5.0f} fit {float(np.sum(wk*(r[:,2]<=lim))):.2f} fit RZ4 {float(np.sum(wk*(r[:,2]*4<=lim*3))):.2f}", flush=True)
