"""Host-side enqueue cost of one config-3 chain call (no synchronisation inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter

ctx = get_context()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fovs = np.stack([synth.synth_fov(i, size=512) for i in range(B)])
d = ctx.asarray(fovs)
seg = FovSegmenter(B, 4, 512, 512, ctx=ctx)
seg.run_c3(d); ctx.synchronize()
for n in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(n):
        seg.run_c3(d)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e3*(t1-t0)/n:.3f} ms/call, total {1e3*(t2-t0)/n:.3f} ms/call")
