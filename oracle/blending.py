"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's channel overlay (R/blending.py:116-226).

``create_overlay(background, layers)``: the background (2-D, clipped to [0, 1]) is broadcast to RGB, then every
layer is composited in order: its intensities go through a two-stop ``LinearSegmentedColormap`` (transparent grey
(0.5, 0.5, 0.5, 0) -> colour when ``zero_transparent``, opaque black -> colour otherwise; R/blending.py:200-219),
``alpha = opacity * rgba[..., 3]``, and the canvas becomes ``clip(alpha * rgb + (1 - alpha) * canvas, 0, 1)`` (ALPHA,
:178-184) or ``clip(canvas + alpha * rgb, 0, 1)`` (ADDITIVE, :187-193).

The colormap is matplotlib's (3.10: ``colors.py`` ``_create_lookup_table`` and ``Colormap.__call__``), restated here:
a 256-entry table ``lut[i] = y0 + i/255 * (y1 - y0)`` evaluated exactly as matplotlib does (``np.linspace`` then the
``distance * (y1 - y0) + y0`` form), and the lookup ``index = trunc(x * 256)`` with 256 folded into 255.  Pinned by
tests/golden/overlay_64.npz (real matplotlib 3.10.8, tools/make_golden_overlay.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
from __future__ import annotations

import numpy as np

N_LUT = 256


def hex_to_rgb(color: str):
    """matplotlib.colors.to_rgba for '#RRGGBB' / '#RGB' strings: components / 255."""
    c = color.lstrip("#")
    if len(c) == 3:
        c = "".join(ch * 2 for ch in c)
    return tuple(int(c[i:i + 2], 16) / 255 for i in (0, 2, 4))


def build_lut(color: str, zero_transparent: bool) -> np.ndarray:
    """(256, 4) float64 table of ``LinearSegmentedColormap.from_list(name, [stop0, color])``."""
    r, g, b = hex_to_rgb(color)
    stop0 = (0.5, 0.5, 0.5, 0.0) if zero_transparent else (0.0, 0.0, 0.0, 1.0)
    stop1 = (r, g, b, 1.0)
    xind = np.linspace(0, 1, N_LUT)
    lut = np.empty((N_LUT, 4), dtype=np.float64)
    for k in range(4):
        y0, y1 = stop0[k], stop1[k]
        # matplotlib: x = [0, 1]; distance = (xind[1:-1] - x[ind-1]) / (x[ind] - x[ind-1]); lut = distance*(y0[ind]-y1[ind-1]) + y1[ind-1]
        distance = (xind[1:-1] - 0.0) / (1.0 - 0.0)
        mid = distance * (y1 - y0) + y0
        lut[:, k] = np.clip(np.concatenate([[y0], mid, [y1]]), 0.0, 1.0)
    return lut


def apply_lut(lut: np.ndarray, x: np.ndarray) -> np.ndarray:
    """``Colormap.__call__`` for float input in [0, 1]."""
    xa = np.array(x, dtype=np.float64, copy=True)
    xa *= N_LUT
    xa[xa == N_LUT] = N_LUT - 1
    idx = np.clip(xa, 0, N_LUT - 1).astype(int)
    return lut[idx]


def create_overlay(background: np.ndarray, layers) -> np.ndarray:
    """layers: iterable of (color_hex, intensities, opacity, zero_transparent, mode) with mode 'alpha' | 'additive'."""
    bg = np.clip(np.asarray(background, dtype=np.float64), 0.0, 1.0)
    canvas = np.repeat(bg[:, :, np.newaxis], 3, axis=2)
    for color, inten, opacity, zero_transparent, mode in layers:
        x = np.clip(np.asarray(inten, dtype=np.float64), 0.0, 1.0)
        rgba = apply_lut(build_lut(color, zero_transparent), x)
        rgb = rgba[..., :3]
        alpha = opacity * rgba[..., 3:4]
        if mode == "additive":
            canvas = np.clip(canvas + alpha * rgb, 0.0, 1.0)
        else:
            canvas = np.clip(alpha * rgb + (1 - alpha) * canvas, 0.0, 1.0)
    return canvas
