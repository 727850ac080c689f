"""Time of the exact (sequential emulation) watershed path on full-size planes, against the parallel paths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops, synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = get_context()
fov = synth.synth_fov(3, size=size)
seg = FovSegmenter(1, 4, size, size, ctx=ctx)
seg.run_c3(ctx.asarray(fov[None]))
ctx.synchronize()
mask, d2, markers = seg.mask_a, seg.d2, seg.markers
for name, kw in (("seeds_first", dict(seeds_first=True)), ("plain raster", dict(seeds_first=False, ties="raster")),
                 ("plain exact", dict(seeds_first=False, ties="exact"))):
    flags = ctx.empty((1,), np.int32)
    hipops.watershed_edt(d2, markers, mask, ties_out=flags, **kw); ctx.synchronize()
    t0 = time.perf_counter()
    out = hipops.watershed_edt(d2, markers, mask, ties_out=flags, **kw); ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:14s} {dt*1e3:9.2f} ms  tied={int(flags.numpy()[0])} labelled px={int((out.numpy()>0).sum())}")
