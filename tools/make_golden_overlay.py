"""Golden vectors for the channel overlay with the REAL matplotlib (>= 3.10, the reference's pin) of this container:

    python tools/make_golden_overlay.py        # system interpreter: matplotlib 3.10.8, numpy 2.2

Restates the few lines of R/blending.py:116-226 around ``LinearSegmentedColormap`` (the reference package itself is not
importable here: SURVEY.md section 8c) and stores inputs + expected RGB canvases in tests/golden/overlay_64.npz.
"""
import os

import matplotlib
import numpy as np
from matplotlib.colors import LinearSegmentedColormap

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "overlay_64.npz")


def cmap_for(color, zero_transparent):
    stops = [(0.5, 0.5, 0.5, 0.0), color] if zero_transparent else [(0.0, 0.0, 0.0, 1.0), color]
    return LinearSegmentedColormap.from_list(f"_chan_{color}", stops)


def overlay(background, layers):
    background = np.clip(background, 0.0, 1.0)
    canvas = np.repeat(background[:, :, np.newaxis], 3, axis=2)
    for color, inten, opacity, zero_transparent, mode in layers:
        inten = np.clip(inten, 0.0, 1.0)
        rgba = cmap_for(color, zero_transparent)(inten)
        rgb = rgba[..., :3]
        alpha = opacity * rgba[..., 3:4]
        if mode == "additive":
            canvas = np.clip(canvas + alpha * rgb, 0.0, 1.0)
        else:
            canvas = np.clip(alpha * rgb + (1 - alpha) * canvas, 0.0, 1.0)
    return canvas


rng = np.random.default_rng(4)
H, W = 64, 80
bg = rng.random((H, W))
dapi, fitc, tritc = rng.random((H, W)) ** 2, rng.random((H, W)) ** 3, rng.random((H, W))
dapi[0, :8] = [0.0, 1.0, 0.5, 255 / 256, 1 / 256, 0.999999, 1e-12, 0.25]  # table edges
out = {
    "versions": np.array([matplotlib.__version__, np.__version__]),
    "background": bg, "dapi": dapi, "fitc": fitc, "tritc": tritc,
    "lut_dapi_t": cmap_for("#0033FF", True)(np.arange(256) / 256.0 + 1e-9),
    "lut_tritc_o": cmap_for("#FFBF00", False)(np.arange(256) / 256.0 + 1e-9),
}
cases = {
    "alpha3": [("#0033FF", dapi, 1.0, True, "alpha"), ("#07FF00", fitc, 0.8, True, "alpha"),
               ("#FFBF00", tritc, 0.5, True, "alpha")],
    "additive3": [("#0033FF", dapi, 1.0, True, "additive"), ("#07FF00", fitc, 1.0, True, "additive"),
                  ("#FFBF00", tritc, 0.7, True, "additive")],
    "opaque_mixed": [("#A30000", dapi, 1.0, False, "alpha"), ("#07FF00", fitc, 0.6, True, "additive")],
    "short_hex": [("#F0A", tritc, 0.9, True, "alpha")],
}
for name, layers in cases.items():
    out[name] = overlay(bg, layers)
out["range"] = overlay(bg * 1.5 - 0.2, [("#0033FF", dapi * 1.3 - 0.1, 1.0, True, "alpha")])  # clipped inputs
np.savez_compressed(OUT, **out)
print("overlay golden:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim == 3})
