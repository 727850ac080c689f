"""uint16 histogram of ONE full-range 2048^2 plane (the case of R/operations.py:186-192 on a raw image): us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(0)
for B in (1, 2, 4):
    u = ctx.asarray(rng.integers(0, 65536, (B, 2048, 2048)).astype(np.uint16))
    for _ in range(5):
        h = hipops.histogram_u16(u)
    ctx.synchronize()
    t = ctx.timer(); t.start()
    for _ in range(50):
        h = hipops.histogram_u16(u)
    t.stop()
    ref = np.bincount(u.numpy()[0].ravel(), minlength=65536)
    assert np.array_equal(h.numpy()[0], ref)
    print(f"hist_u16, {B} plane(s): {t.elapsed_ms() / 50 * 1e3 / B:.1f} us per plane")
