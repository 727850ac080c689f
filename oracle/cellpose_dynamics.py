"""CPU restatement of Cellpose's flow -> mask post-processing.  TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

The reference calls ``cellpose.models.CellposeModel.eval`` (R/model.py:206-215, :270-290); cellpose (>= 4.0.8,
/root/reference/uv.lock) is a third-party dependency that is neither vendored under /root/reference nor importable
offline, and the reference's tests mock it (RT/test_model.py:124-376), so no golden vector exists for this step.
This module restates the PUBLISHED algorithm -- Stringer, Wang, Michaelos, Pachitariu, "Cellpose: a generalist
algorithm for cellular segmentation", Nature Methods 18 (2021), and the package's ``dynamics.py``
(``compute_masks`` -> ``follow_flows`` / ``steps_interp`` -> ``get_masks``) as recalled from its public source --
with numpy / scipy, and is what the HIP kernels (csrc/amt_dynamics.hip) are compared with.  Stated choices where
the package's behaviour is unspecified: seeds of equal count are ordered by decreasing raster index (a stable
argsort followed by the package's ``[::-1]``); the flow-error filter (``remove_bad_flow_masks``) and the per-mask
hole filling are NOT part of this restatement.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

RPAD = 20


def follow_flows(dP: np.ndarray, cell: np.ndarray, niter: int = 200) -> np.ndarray:
    """End positions (2, H, W) float32 of every pixel after ``niter`` Euler steps through ``dP * cell / 5``.

    Only cell pixels with ``|dY * cell / 5| > 1e-3`` move (``follow_flows``: ``inds = abs(dP[0]) > 1e-3``).  The flow is
    sampled as ``torch.nn.functional.grid_sample(..., align_corners=False)`` (bilinear, zero padding) samples it after
    the positions have been normalised by ``size - 1`` (``steps_interp``): sample coordinate
    ``s = u * size / (size - 1) - 0.5``; positions are clamped to [0, size - 1].  float32 arithmetic."""
    H, W = cell.shape
    f = (dP.astype(np.float32) * cell.astype(np.float32)) / np.float32(5.0)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    moving = cell & (np.abs(f[0]) > np.float32(1e-3))
    y, x = yy[moving].copy(), xx[moving].copy()
    sy, sx = np.float32(H) / np.float32(H - 1), np.float32(W) / np.float32(W - 1)
    half = np.float32(0.5)
    for _ in range(niter):
        fy, fx = y * sy - half, x * sx - half
        gy, gx = np.floor(fy), np.floor(fx)
        iy, ix = gy.astype(np.int64), gx.astype(np.int64)
        wy, wx = fy - gy, fx - gx
        vy = np.zeros_like(y)
        vx = np.zeros_like(x)
        for a in (0, 1):
            for b in (0, 1):
                cy, cx = iy + a, ix + b
                ok = (cy >= 0) & (cy < H) & (cx >= 0) & (cx < W)
                w = (wy if a else np.float32(1) - wy) * (wx if b else np.float32(1) - wx)
                sel = np.flatnonzero(ok)
                vy[sel] += w[sel] * f[0, cy[sel], cx[sel]]
                vx[sel] += w[sel] * f[1, cy[sel], cx[sel]]
        y = np.clip(y + vy, np.float32(0), np.float32(H - 1))
        x = np.clip(x + vx, np.float32(0), np.float32(W - 1))
    p = np.stack([yy, xx])
    p[0][moving] = y
    p[1][moving] = x
    return p


def get_masks(p: np.ndarray, max_size_fraction: float = 0.4, min_size: int = 15):
    """``get_masks``: histogram of the end positions -> seeds -> grown seed regions -> labels, then the size filters
    and the renumbering (``fastremap.renumber`` numbers labels in order of first appearance)."""
    H, W = p.shape[1:]
    hy = p[0].astype(np.int32).ravel() + RPAD
    hx = p[1].astype(np.int32).ravel() + RPAD
    Hp, Wp = H + 2 * RPAD, W + 2 * RPAD
    h = np.zeros((Hp, Wp), np.int64)
    np.add.at(h, (hy, hx), 1)
    hmax = ndi.maximum_filter1d(ndi.maximum_filter1d(h, 5, axis=0), 5, axis=1)
    seeds = np.flatnonzero(((h - hmax) > -1e-6) & (h > 10))
    counts = h.ravel()[seeds]
    order = np.lexsort((seeds, counts))[::-1]  # count descending, ties: raster index descending
    seeds = seeds[order]
    M = np.zeros((Hp, Wp), np.int32)
    for k, s in enumerate(seeds):
        cy, cx = divmod(int(s), Wp)
        cur = {(cy, cx)}
        for _ in range(5):
            nxt = set()
            for (y, x) in cur:
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        q = (y + dy, x + dx)
                        if 0 <= q[0] < Hp and 0 <= q[1] < Wp and h[q] > 2:
                            nxt.add(q)
            cur = nxt
        for q in cur:
            M[q] = k + 1  # later seeds overwrite earlier ones
    lab = M[hy, hx].reshape(H, W)
    big = int(H * W * max_size_fraction)
    ids, cnt = np.unique(lab, return_counts=True)
    first = {int(i): int(np.flatnonzero(lab.ravel() == i)[0]) for i in ids if i > 0}
    keep = [int(i) for i, c in zip(ids, cnt) if i > 0 and min_size <= c <= big]
    keep.sort(key=lambda i: first[i])
    out = np.zeros_like(lab)
    for new, i in enumerate(keep, start=1):
        out[lab == i] = new
    return out


def compute_masks(dP: np.ndarray, cellprob: np.ndarray, cellprob_threshold: float = 0.0, niter: int = 200,
                  min_size: int = 15, max_size_fraction: float = 0.4) -> np.ndarray:
    """(2, H, W) flows + (H, W) cell probability -> int32 labels (``compute_masks`` without the flow-error filter)."""
    cell = cellprob > np.float32(cellprob_threshold)
    if not cell.any():
        return np.zeros(cellprob.shape, np.int32)
    p = follow_flows(dP, cell, niter)
    return get_masks(p, max_size_fraction, min_size).astype(np.int32)
