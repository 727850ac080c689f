"""Where is hist_f64_kernel's floor?  Per-plane time of the float64 histogram (+ byte plane of bins) for batches that
stream from HBM (48 planes = 1.6 GB) and for batches that fit the 256 MB memory-side cache (1-4 planes), and of the
uint16 histogram (2 B/px) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context

ctx = get_context()
rng = np.random.default_rng(0)
S = 2048
for B in (1, 2, 4, 8, 16, 48):
    one = rng.random((S, S))
    f = ctx.empty((B, S, S), np.float64)
    for i in range(B):
        ctx.asarray(one, out=f[i])
    mm = hipops.minmax(f)
    thr = ctx.empty((f.shape[0],), np.float64)
    code = ctx.empty((f.shape[0],), np.float64)
    bins = ctx.empty(f.shape, np.uint8)
    for name, fn in (("hist_f64", lambda: hipops.histogram_f64(f, 256, mm=mm)),
                     ("otsu_bins", lambda: hipops.threshold_otsu_bins(f, mm, thr, code, bins))):
        for _ in range(3):
            fn()
        ctx.synchronize()
        t = ctx.timer()
        t.start()
        for _ in range(20):
            fn()
        t.stop()
        ms = t.elapsed_ms() / 20
        print(f"B={f.shape[0]:3d} {name:10s} {ms * 1e3:8.1f} us  = {ms * 1e3 / f.shape[0]:6.2f} us/plane", flush=True)
u = ctx.asarray(rng.integers(0, 65536, (48, S, S)).astype(np.uint16))
for _ in range(3):
    hipops.histogram_u16(u)
ctx.synchronize()
t = ctx.timer(); t.start()
for _ in range(20):
    hipops.histogram_u16(u)
t.stop()
ms = t.elapsed_ms() / 20
print(f"B= 48 hist_u16   {ms * 1e3:8.1f} us  = {ms * 1e3 / 48:6.2f} us/plane")
