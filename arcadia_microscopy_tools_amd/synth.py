"""Deterministic synthetic fields of view (SURVEY.md section 8d generator spec).

Host-side data generation only (numpy + scipy.ndimage.gaussian_filter to blur the canvases); it is
not part of the accelerated path.  One FOV = (C, Y, X) uint16 with channels
(BRIGHTFIELD, DAPI, FITC, TRITC); a plate = FOV indices 0..383 (wells A01..P24).
"""
from __future__ import annotations

import numpy as np

CHANNEL_NAMES = ("BRIGHTFIELD", "DAPI", "FITC", "TRITC")


def _disks(shape, cy, cx, rad):
    canvas = np.zeros(shape, dtype=np.float64)
    H, W = shape
    for y, x, r in zip(cy, cx, rad):
        r_i = int(np.ceil(r))
        y0, y1 = max(0, y - r_i), min(H, y + r_i + 1)
        x0, x1 = max(0, x - r_i), min(W, x + r_i + 1)
        yy, xx = np.ogrid[y0:y1, x0:x1]
        canvas[y0:y1, x0:x1][(yy - y) ** 2 + (xx - x) ** 2 < r * r] = 1.0
    return canvas


def synth_fov(fov_index: int, size: int = 2048, n_nuclei: int | None = None) -> np.ndarray:
    """One synthetic (4, size, size) uint16 field of view.

    At size 2048 this is the generator of SURVEY.md section 8d (1500 nuclei, radii 6..13); for smaller
    sizes the nucleus count scales with the area so the density stays the same.
    """
    from scipy import ndimage as ndi

    rng = np.random.default_rng(1000 + int(fov_index))
    if n_nuclei is None:
        n_nuclei = max(1, int(round(1500 * (size / 2048.0) ** 2)))
    margin = 20 if size > 60 else 2
    cy = rng.integers(margin, size - margin, n_nuclei)
    cx = rng.integers(margin, size - margin, n_nuclei)
    rad = rng.integers(6, 14, n_nuclei)
    shape = (size, size)

    def clip16(a):
        return np.clip(a, 0, 65535).astype(np.uint16)

    canvas = _disks(shape, cy, cx, rad.astype(np.float64))
    amp = rng.uniform(2000, 12000)
    dapi = ndi.gaussian_filter(canvas, 1.5) * amp + rng.normal(400, 30, shape)
    bright = rng.normal(12900, 1500, shape)
    c2 = _disks(shape, cy, cx, rad * 1.8)
    fitc = ndi.gaussian_filter(c2, 3.0) * rng.uniform(300, 3000) + rng.normal(480, 40, shape)
    c3 = _disks(shape, cy, cx, rad * 2.2)
    tritc = ndi.gaussian_filter(c3, 3.0) * rng.uniform(100, 800) + rng.normal(170, 15, shape)
    return np.stack([clip16(bright), clip16(dapi), clip16(fitc), clip16(tritc)], axis=0)


def well_id(fov_index: int) -> str:
    """384-well plate id (rows A..P, columns 1..24) for a FOV index 0..383 (R/microplate.py:24-45 style)."""
    row, col = divmod(int(fov_index) % 384, 24)
    return f"{chr(ord('A') + row)}{col + 1:02d}"


def synthetic_flows(shape, n_cells: int, seed: int = 0, noise: float = 0.0):
    """A flow field of the kind the network is trained to produce: for every synthetic cell (a disk) the unit vectors
    that point to its centre times a smooth profile, plus the logit-like cell probability.  -> (dP, cellprob, truth)."""
    rng = np.random.default_rng(seed)
    H, W = shape
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    dP = np.zeros((2, H, W), np.float32)
    prob = np.full((H, W), -6.0, np.float32)
    truth = np.zeros((H, W), np.int32)
    for k in range(n_cells):
        r = float(rng.integers(7, 14))
        cy, cx = float(rng.uniform(r + 2, H - r - 2)), float(rng.uniform(r + 2, W - r - 2))
        d = np.hypot(yy - cy, xx - cx)
        inside = (d < r) & (truth == 0)
        truth[inside] = k + 1
        norm = np.maximum(d, 1e-3)
        dP[0][inside] = (-(yy - cy) / norm)[inside] * 5.0 * np.minimum(d / 2.0, 1.0)[inside]
        dP[1][inside] = (-(xx - cx) / norm)[inside] * 5.0 * np.minimum(d / 2.0, 1.0)[inside]
        prob[inside] = 6.0
    if noise:
        dP += rng.normal(0, noise, dP.shape).astype(np.float32)
    return dP, prob, truth
