"""create_overlay under random colours (6- and 3-digit hex), opacities, transparency flags, blend modes, layer counts
(0..8), shapes and value ranges (incl. exact 0 / 1, NaN-free out-of-range values that must be clipped) against the CPU
restatement pinned to matplotlib (oracle/blending.py).  Usage: fuzz_overlay.py [cases] [seed]."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import BlendMode, Channel, Layer, create_overlay
from oracle import blending as ob

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(ncases):
    H, W = int(rng.integers(1, 90)), int(rng.integers(1, 130))

    def plane():
        kind = int(rng.integers(0, 4))
        x = rng.random((H, W))
        if kind == 1:
            x = np.round(x * 4) / 4          # exact table positions incl. 0 and 1
        elif kind == 2:
            x = x * 1.6 - 0.3                # out of range: clipped
        elif kind == 3:
            x = (x > 0.5).astype(np.float64)
        return x

    bg = plane()
    layers, ref_layers = [], []
    for _ in range(int(rng.integers(0, 9))):
        digits = 3 if rng.random() < 0.2 else 6
        color = "#" + "".join(rng.choice(list("0123456789ABCDEFabcdef"), digits))
        x = plane()
        opacity = float(rng.choice([0.0, 1.0, rng.random()]))
        zt = bool(rng.integers(0, 2))
        mode = (BlendMode.ALPHA, BlendMode.ADDITIVE)[int(rng.integers(0, 2))]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            layers.append(Layer(Channel("X", color), x, opacity, zt, mode))
        ref_layers.append((color, x, opacity, zt, "additive" if mode is BlendMode.ADDITIVE else "alpha"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = create_overlay(bg, layers)
    want = ob.create_overlay(bg, ref_layers)
    ok = got.shape == want.shape and got.dtype == np.float64 and np.array_equal(got, want)
    if not ok:
        bad += 1
        print("MISMATCH", case, (H, W), len(layers), float(np.abs(got - want).max()) if got.shape == want.shape else got.shape,
              flush=True)
print("cases", ncases, "BAD", bad)
sys.exit(1 if bad else 0)
