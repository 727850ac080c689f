"""Cost decomposition of the packed uint16 erosion kernel by footprint shape (32 planes of 2048^2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(0)
d = ctx.asarray(rng.integers(0, 65536, (32, 2048, 2048)).astype(np.uint16))
out = ctx.empty((32, 2048, 2048), np.uint16)
fps = {"1x1": np.ones((1, 1), np.uint8), "1x5": np.ones((1, 5), np.uint8), "5x1": np.ones((5, 1), np.uint8),
       "5x5": np.ones((5, 5), np.uint8), "disk2": hipops.disk(2), "disk7": hipops.disk(7), "15x1": np.ones((15, 1), np.uint8),
       "1x15": np.ones((1, 15), np.uint8)}
for name, fp in fps.items():
    hipops.erosion(d, fp, out=out); ctx.synchronize()
    t = ctx.timer(); t.start()
    for _ in range(5):
        hipops.erosion(d, fp, out=out)
    t.stop(); ctx.synchronize()
    print(f"{name:6s} {t.elapsed_ms()/5*1e3:8.1f} us")
