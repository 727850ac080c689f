"""Run ONE device operator repeatedly on a synthetic batch (for rocprofv3 --pmc / --kernel-trace passes).

usage: python tools/probe_op.py gaussian|toc|otsu|edt|peaks [--batch 32] [--reps 10]
"""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from arcadia_microscopy_tools_amd import hipops, synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter

ap = argparse.ArgumentParser()
ap.add_argument("op")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--size", type=int, default=2048)
a = ap.parse_args()
ctx = get_context()
uniq = np.stack([synth.synth_fov(i, size=a.size) for i in range(4)])
fovs = ctx.asarray(np.concatenate([uniq] * (a.batch // 4)))
seg = FovSegmenter(a.batch, 4, a.size, a.size, ctx=ctx)
seg.run_c3(fovs)
from arcadia_microscopy_tools_amd import _hip
import numpy as _np
dapi = ctx.empty((a.batch, a.size, a.size), _np.uint16)
gcopy = ctx.empty((a.batch, a.size, a.size), _np.float64)
ctx.synchronize()
t = ctx.timer()
t.start()
for _ in range(a.reps):
    if a.op == "gaussian":
        hipops.gaussian(fovs, 2.0, channel=1, out=seg.gauss)
    elif a.op == "otsu":
        hipops.threshold_otsu(seg.gauss, out=seg.thr)
    elif a.op == "toc":
        hipops.threshold_open_close(seg.gauss, seg.thr, seg.footprint, out=seg.mask_a)
    elif a.op == "edt":
        hipops.edt(seg.mask_a, want_edt=False, d2_out=seg.d2)
    elif a.op == "peaks":
        hipops.peak_mask(seg.d2, seg.mask_a, 5, out=seg.peaks)
    elif a.op == "props":
        hipops.regionprops_full(seg.labels, fovs, seg.max_cells, out=seg.table, iout=seg.itable)
    elif a.op == "convert":
        hipops.to_float64(dapi, out=seg.gauss)
    elif a.op == "memset":
        import ctypes
        from arcadia_microscopy_tools_amd import _hip
        _hip.check(ctx._lib.amt_memset(ctx.handle, seg.gauss.ptr, 0, seg.gauss.nbytes), "memset")
    elif a.op == "d2d":
        _hip.check(ctx._lib.amt_memcpy_d2d(ctx.handle, seg.gauss.ptr, gcopy.ptr, seg.gauss.nbytes), "d2d")
    else:
        raise SystemExit("unknown op")
t.stop()
ctx.synchronize()
print(f"{a.op}: {t.elapsed_ms() / a.reps:.3f} ms per call (batch {a.batch})")
