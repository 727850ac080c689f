"""Host-side cost of the int64 -> int32 narrowing that SegmentationMask's label upload performs (numpy casting copy on
the copy threads) and of the extrema pass of its constructor."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, ".")
from arcadia_microscopy_tools_amd.masks import _extrema

rng = np.random.default_rng(0)
a = rng.integers(0, 1500, (2048, 2048)).astype(np.int64)
dst = np.empty(a.size, np.int32)
flat = a.reshape(-1)
for threads in (1, 4, 8):
    with ThreadPoolExecutor(threads) as ex:
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            step = flat.size // 8
            futs = [ex.submit(np.copyto, dst[o:o + step], flat[o:o + step], "unsafe") for o in range(0, flat.size, step)]
            [f.result() for f in futs]
            best = min(best, time.perf_counter() - t0)
        print(f"narrow int64->int32, 8 chunks on {threads} threads: {best * 1e3:.2f} ms")
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); _extrema(a); best = min(best, time.perf_counter() - t0)
print(f"_extrema (int64 2048^2): {best * 1e3:.2f} ms")
