"""Time the wide-radius Gaussian (sigma = 16, the default of subtract_background_dog) and the DoG on 32 planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(0)
d = ctx.asarray(rng.integers(0, 65536, (32, 2048, 2048)).astype(np.uint16))
out = ctx.empty((32, 2048, 2048), np.float64)
for name, fn in (("gauss16", lambda: hipops.gaussian(d, 16.0, out=out)),
                 ("dog0.6/16", lambda: hipops.difference_of_gaussians(d, 0.6, 16.0, out=out))):
    fn(); ctx.synchronize()
    t = ctx.timer(); t.start()
    for _ in range(5):
        fn()
    t.stop(); ctx.synchronize()
    print(f"{name:10s} {t.elapsed_ms()/5*1e3:8.1f} us")
