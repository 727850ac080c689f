"""CPU restatement of Cellpose's flow -> mask post-processing.  TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

The reference calls ``cellpose.models.CellposeModel.eval`` (R/model.py:206-215, :270-290); cellpose (>= 4.0.8,
/root/reference/uv.lock) is a third-party dependency that is neither vendored under /root/reference nor importable
offline, and the reference's tests mock it (RT/test_model.py:124-376), so no golden vector exists for this step.
This module restates the PUBLISHED algorithm -- Stringer, Wang, Michaelos, Pachitariu, "Cellpose: a generalist
algorithm for cellular segmentation", Nature Methods 18 (2021), and the package's ``dynamics.py``
(``compute_masks`` -> ``follow_flows`` / ``steps_interp`` -> ``get_masks``) as recalled from its public source --
with numpy / scipy, and is what the HIP kernels (csrc/amt_dynamics.hip) are compared with.  Stated choices where
the package's behaviour is unspecified: seeds of equal count are ordered by decreasing raster index (a stable
argsort followed by the package's ``[::-1]``).

Round 3 adds, from the same sources and just as unpinned: the flow-error quality filter (``remove_bad_flow_masks`` ->
``metrics.flow_error`` -> ``dynamics.masks_to_flows``: heat diffusion from every mask's centre, float64), the final
``utils.fill_holes_and_remove_small_masks``, the diameter rescaling around the network (``transforms.resize_image``,
bilinear) and the tiled forward pass (``transforms.make_tiles`` / ``average_tiles``).
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
                  min_size: int = 15, max_size_fraction: float = 0.4, flow_threshold: float | None = None,
                  fill_holes: bool = False) -> np.ndarray:
    """(2, H, W) flows + (H, W) cell probability -> int32 labels.

    ``flow_threshold`` None / 0: ``compute_masks`` without the flow-error filter (rounds 1-2).  Otherwise the package's
    order: ``get_masks`` (size ceiling, renumbering) -> ``remove_bad_flow_masks`` -> ``fill_holes_and_remove_small_masks``
    (``fill_holes`` False leaves the holes: only its size floor and renumbering)."""
    cell = cellprob > np.float32(cellprob_threshold)
    if not cell.any():
        return np.zeros(cellprob.shape, np.int32)
    p = follow_flows(dP, cell, niter)
    if not flow_threshold and not fill_holes:
        return get_masks(p, max_size_fraction, min_size).astype(np.int32)
    masks = get_masks(p, max_size_fraction, 0).astype(np.int32)
    if flow_threshold and masks.max() > 0:
        masks = remove_bad_flow_masks(masks, dP, flow_threshold)
    return fill_holes_and_remove_small_masks(masks, min_size, fill_holes)


# ---- flow-error filter ------------------------------------------------------------------------------------------------
def mask_centers(masks: np.ndarray):
    """``dynamics.get_centers``: per label (ascending) the mask pixel closest to the mean position of its pixels (first
    in raster order among equals), and ext = height + width + 2 of its bounding box.  Absent labels: centre (-1, -1)."""
    K = int(masks.max())
    centers = np.full((K, 2), -1, np.int64)
    ext = np.zeros(K, np.int64)
    for k, sl in enumerate(ndi.find_objects(masks)):
        if sl is None:
            continue
        yi, xi = np.nonzero(masks[sl] == k + 1)
        ymed, xmed = yi.mean(), xi.mean()
        imin = int(((xi - xmed) ** 2 + (yi - ymed) ** 2).argmin())
        centers[k] = (yi[imin] + sl[0].start, xi[imin] + sl[1].start)
        ext[k] = (sl[0].stop - sl[0].start) + (sl[1].stop - sl[1].start) + 2
    return centers, ext


def masks_to_flows(masks: np.ndarray, n_iter: int | None = None):
    """``dynamics.masks_to_flows_gpu``: T = 0; ``n_iter`` = 2 * max(ext) times: T[centres] += 1, then every mask pixel
    becomes the mean over its 3 x 3 neighbourhood of the values that lie in the SAME mask (others count as 0), all
    pixels at once, float64.  Flow = central differences of T (neighbouring values taken as they are, other masks
    included), normalised to unit length with 1e-60 added to the norm.  Returns ((2, H, W) float64, T)."""
    H, W = masks.shape
    centers, ext = mask_centers(masks)
    if n_iter is None:
        n_iter = int(2 * ext.max()) if ext.size else 0
    mp = np.pad(masks.astype(np.int64), 1)
    T = np.zeros(mp.shape, np.float64)
    fg = mp > 0
    yy, xx = np.nonzero(fg)
    offs = [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (-1, 1), (1, -1), (1, 1)]
    same = [mp[yy + dy, xx + dx] == mp[yy, xx] for dy, dx in offs]
    cy, cx = centers[centers[:, 0] >= 0].T + 1 if len(centers) else (np.zeros(0, int), np.zeros(0, int))
    for _ in range(n_iter):
        T[cy, cx] += 1
        acc = np.zeros(len(yy), np.float64)
        for (dy, dx), ok in zip(offs, same):  # the package's order of the nine terms (torch's mean sums them in order)
            acc = acc + T[yy + dy, xx + dx] * ok
        T[yy, xx] = acc / 9
    dy = T[yy + 1, xx] - T[yy - 1, xx]
    dx = T[yy, xx + 1] - T[yy, xx - 1]
    norm = 1e-60 + np.sqrt(dy ** 2 + dx ** 2)
    mu = np.zeros((2, H, W), np.float64)
    mu[0, yy - 1, xx - 1] = dy / norm
    mu[1, yy - 1, xx - 1] = dx / norm
    return mu, T[1:-1, 1:-1]


def flow_error(masks: np.ndarray, dP_net: np.ndarray):
    """``metrics.flow_error``: per label, the mean over its pixels of the squared difference between the flows
    re-derived from the masks and the network's flows / 5, summed over the two components."""
    K = int(masks.max())
    mu, _ = masks_to_flows(masks)
    err = np.zeros(K, np.float64)
    idx = np.arange(1, K + 1)
    for i in range(2):
        d = (mu[i] - dP_net[i].astype(np.float32) / np.float32(5.0)) ** 2
        err += np.nan_to_num(ndi.mean(d, masks, idx))
    return err


def remove_bad_flow_masks(masks: np.ndarray, dP_net: np.ndarray, threshold: float = 0.4) -> np.ndarray:
    err = flow_error(masks, dP_net)
    bad = 1 + np.flatnonzero(err > threshold)
    out = masks.copy()
    out[np.isin(out, bad)] = 0
    return out


def fill_holes_and_remove_small_masks(masks: np.ndarray, min_size: int = 15, fill_holes: bool = True) -> np.ndarray:
    """``utils.fill_holes_and_remove_small_masks``: labels in ascending order; a label with fewer than ``min_size``
    pixels is dropped, every other one is hole-filled inside its bounding box (``scipy.ndimage.binary_fill_holes``,
    written OVER whatever lies in the hole) and renumbered 1, 2, ... in that order.  IN SEQUENCE: a later label sees
    the image as the earlier ones left it."""
    out = masks.copy()
    j = 0
    for i, sl in enumerate(ndi.find_objects(masks)):
        if sl is None:
            continue
        msk = out[sl] == i + 1
        npix = int(msk.sum())
        if min_size > 0 and npix < min_size:
            out[sl][msk] = 0
        elif npix > 0:
            if fill_holes:
                msk = ndi.binary_fill_holes(msk)
            out[sl][msk] = j + 1
            j += 1
    return out


# ---- around the network: diameter rescaling and tiles --------------------------------------------------------------------
def resize_bilinear(a: np.ndarray, Ly: int, Lx: int) -> np.ndarray:
    """(..., H, W) -> (..., Ly, Lx) as ``cv2.resize(..., interpolation=INTER_LINEAR)`` / ``torch.nn.functional.interpolate(
    mode='bilinear', align_corners=False)`` without antialiasing: sample position (i + 0.5) * H / Ly - 0.5, clamped to the
    image, float32 arithmetic."""
    a = np.asarray(a, np.float32)
    H, W = a.shape[-2:]

    def taps(n_in, n_out):
        s = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        s = np.maximum(s, np.float32(0))
        i0 = np.minimum(np.floor(s).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w1 = (s - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, w1

    y0, y1, wy = taps(H, Ly)
    x0, x1, wx = taps(W, Lx)
    rows = a[..., y0, :] * (np.float32(1) - wy)[:, None] + a[..., y1, :] * wy[:, None]
    return (rows[..., x0] * (np.float32(1) - wx) + rows[..., x1] * wx).astype(np.float32)


def tile_starts(L: int, bsize: int, tile_overlap: float = 0.1):
    """``transforms.make_tiles``: tile origins along one axis (tiles of min(bsize, L) pixels)."""
    tile_overlap = min(0.5, max(0.05, tile_overlap))
    b = min(bsize, L)
    n = 1 if L <= bsize else int(np.ceil((1.0 + 2 * tile_overlap) * L / bsize))
    return np.linspace(0, L - b, n).astype(int), b


def taper_mask(ly: int, lx: int, sig: float = 7.5) -> np.ndarray:
    """``transforms._taper_mask``: sigmoid roll-off towards the tile edges."""
    bsize = max(224, max(ly, lx))
    xm = np.arange(bsize)
    xm = np.abs(xm - xm.mean())
    m = 1 / (1 + np.exp((xm - (bsize / 2 - 20)) / sig))
    m = m * m[:, np.newaxis]
    return m[bsize // 2 - ly // 2: bsize // 2 + ly // 2 + ly % 2, bsize // 2 - lx // 2: bsize // 2 + lx // 2 + lx % 2]


def tiled_apply(fn, img: np.ndarray, bsize: int = 256, tile_overlap: float = 0.1, nout: int = 3) -> np.ndarray:
    """``run_net`` tiling: ``fn`` maps (N, C, b, b) tiles to (N, nout, b, b); outputs are blended with the taper mask
    (``average_tiles``)."""
    C, Ly, Lx = img.shape
    ys, by = tile_starts(Ly, bsize, tile_overlap)
    xs, bx = tile_starts(Lx, bsize, tile_overlap)
    tiles = np.stack([img[:, y:y + by, x:x + bx] for y in ys for x in xs])
    out = fn(tiles)
    mask = taper_mask(by, bx).astype(np.float32)
    yf = np.zeros((nout, Ly, Lx), np.float32)
    navg = np.zeros((Ly, Lx), np.float32)
    k = 0
    for y in ys:
        for x in xs:
            yf[:, y:y + by, x:x + bx] += out[k] * mask
            navg[y:y + by, x:x + bx] += mask
            k += 1
    return yf / navg
