#!/usr/bin/env python3
"""CPU model of what the LDS-staged TRILINEAR kernel's brick layers need per tile (vr_tslab.hip), over a set of poses,
for the design space of round 4:
  tile shape (W x H pixels per workgroup), layer thickness T (4 = whole bricks, 2 = half bricks), and three ways of
  laying a layer out in the ring:
    bbox  : bounding rectangle of the footprint, RA x RB slots (round 3's plan);
    rows  : torus of RB rows x (widest row) slots -- every row holds its own contiguous range [alo(b), alo(b) + w(b)),
            stored at (a mod RAmax): the tap address stays X[i] + Y[j] + Z[k] with static tables; best of the two
            orientations (rows along a or along b);
    dense : sum of the row widths (needs per-layer row tables in the tap path).
The footprint of a layer is the hull of the four corner rays' crossings of the two planes bounding the layer's samples
(+- the kernel's margins); row extents come from all 28 segments between those 8 points.
Prints, per pose, the percentiles of the slots needed and the fraction of tiles that fit `lim` slots per layer.
  tools/model/tslab_footprint2.py [N] [W H]"""
import importlib, itertools, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
vra = importlib.import_module("volume-renderer_amd")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ((1920, 1080) if N <= 1024 else (3840, 2160))


def poses():
    out = {}
    r = vra.RendererCore(-1)
    out["default"] = r.getCameraBlock().astype(np.float64)
    r.cameraOrient(0.0, -(np.pi / 6) / 0.7, (np.pi / 4) / 0.7)
    out["offaxis"] = r.getCameraBlock().astype(np.float64)
    rng = np.random.default_rng(7)                  # tools/tri_ms.py orbit6
    for k in range(6):
        r.resetCamera()
        ze, az = float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-3, 3))
        r.cameraOrient(0.0, ze, az)
        out[f"ze{ze:+.2f}az{az:+.2f}"] = r.getCameraBlock().astype(np.float64)
    return out


PAIRS = list(itertools.combinations(range(8), 2))


def row_extents(p8, q8, T_unused, mrg):
    """rows = brick rows of coordinate p; returns (first row, [lo_q brick, hi_q brick] per row) of the hull of the 8 points"""
    r0, r1 = int(np.floor(p8.min() - mrg)) >> 2, int(np.floor(p8.max() + mrg)) >> 2
    lo = np.full(r1 - r0 + 1, np.inf); hi = np.full(r1 - r0 + 1, -np.inf)
    y0 = 4.0 * np.arange(r0, r1 + 1) - mrg; y1 = y0 + 4.0 + 2 * mrg
    for (u, v) in PAIRS:
        pu, pv, qu, qv = p8[u], p8[v], q8[u], q8[v]
        if pu == pv:
            inside = (y0 <= pu) & (pu <= y1)
            lo = np.where(inside, np.minimum(lo, min(qu, qv)), lo); hi = np.where(inside, np.maximum(hi, max(qu, qv)), hi)
            continue
        ta, tb = (y0 - pu) / (pv - pu), (y1 - pu) / (pv - pu)
        t0, t1 = np.maximum(0.0, np.minimum(ta, tb)), np.minimum(1.0, np.maximum(ta, tb))
        ok = t0 <= t1
        for t in (t0, t1):
            q = qu + t * (qv - qu)
            lo = np.where(ok, np.minimum(lo, q), lo); hi = np.where(ok, np.maximum(hi, q), hi)
    ok = hi >= lo
    lob = (np.floor(np.where(ok, lo, 0) - mrg).astype(int) >> 2); hib = (np.floor(np.where(ok, hi, 0) + mrg).astype(int) >> 2)
    w = np.where(ok, hib - lob + 1, 0)
    return w


def model(c, tile_w, tile_h, T):
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

    eye = c[16:19]
    E = to_voxel(eye)
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
                out.append((work, 1e9, 1e9, 1e9)); continue
            a_ax, b_ax = (1 if m == 0 else 0), (1 if m == 2 else 2)
            fm_in = np.clip(E[m] + G[:, m] * np.maximum(tmin, 0), 0, N); fm_out = np.clip(E[m] + G[:, m] * tmax, 0, N)
            fm_in, fm_out = fm_in[hitc], fm_out[hitc]
            l_lo, l_hi = int(min(fm_in.min(), fm_out.min())) // T, min(int(max(fm_in.max(), fm_out.max())) // T, N // T - 1)
            delta = 0.0625 + 1100 * 1.2e-7 * N + np.abs(E).max() * 2.4e-7
            mrg = 0.5 + delta
            bb = rw = dn = 0
            Ls = np.arange(l_lo, l_hi + 1)
            for L in Ls[:: max(1, len(Ls) // 12)]:
                cl = np.array([T * L - 0.5 - delta, T * L + T + 0.5 + delta])
                t = (cl[:, None] - E[m]) / G[None, :, m]
                a8 = (E[a_ax] + t * G[None, :, a_ax]).ravel(); b8 = (E[b_ax] + t * G[None, :, b_ax]).ravel()
                na = (int(np.floor(a8.max() + mrg)) >> 2) - (int(np.floor(a8.min() - mrg)) >> 2) + 1
                nb = (int(np.floor(b8.max() + mrg)) >> 2) - (int(np.floor(b8.min() - mrg)) >> 2) + 1
                bb = max(bb, na * nb)
                w_b = row_extents(b8, a8, T, mrg)          # rows of constant b, extents along a
                w_a = row_extents(a8, b8, T, mrg)
                rw = max(rw, min(int(w_b.max()) * len(w_b), int(w_a.max()) * len(w_a)))
                dn = max(dn, min(int(w_b.sum()), int(w_a.sum())))
            out.append((work, bb, rw, dn))
    return np.array(out)


if __name__ != "__main__":
    raise SystemExit      # (imported for model() / poses(): tools/model/tslab_tileshape.py)
configs = [  # name, tile_w, tile_h, T, slot bytes, LDS bytes per workgroup
    ("32x16 T4 160B  80K (r3)", 32, 16, 4, 160, 80),
    ("32x16 T4 160B 160K (v8)", 32, 16, 4, 160, 160),
    ("32x32 T4 160B 160K (v9)", 32, 32, 4, 160, 160),
    ("64x16 T4 160B 160K     ", 64, 16, 4, 160, 160),
    ("32x16 T2  80B  80K     ", 32, 16, 2, 80, 80),
    ("32x32 T2  80B 160K     ", 32, 32, 2, 80, 160),
    ("32x16 T4 120B  80K     ", 32, 16, 4, 120, 80),
    ("32x32 T4 120B 160K     ", 32, 32, 4, 120, 160),
]
P = poses()
for name, tw, th, T, slot, kb in configs:
    avail = (kb * 1024 - 512 - 16 - 512 - 4096) // slot     # ~4 KiB of plan + tables
    lim = avail // 3
    print(f"== {name}: {lim} slots per layer (RZ = 3)")
    for pn, c in P.items():
        r = model(c, tw, th, T)
        wk = r[:, 0] / r[:, 0].sum()
        line = f"   {pn:18s} tiles {len(r):5d}"
        for k, lab in ((1, "bbox"), (2, "rows"), (3, "dense")):
            v = r[:, k]
            fit = float(np.sum(wk * (v <= lim)))
            line += f" | {lab} p50 {np.percentile(v, 50):5.0f} p90 {np.percentile(v, 90):5.0f} fit(work) {fit:.2f}"
        print(line, flush=True)
