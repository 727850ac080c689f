"""Cycle split of the sequential heap emulation (ws_global_kernel) on one tied 2048^2 plane.  Needs the instrumented
library: build amt_watershed.hip with -DWS_STATS into tools/variants/libamt_hip_dbg.so (see tools/ws_stats.py) and run
    AMT_HIP_LIB=$PWD/tools/variants/libamt_hip_dbg.so python tools/wsg_stats.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import _hip, synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter

ctx = get_context()
lib = ctypes.CDLL(os.environ["AMT_HIP_LIB"])
fov = np.stack([synth.synth_fov(3)])
seg = FovSegmenter(1, 4, 2048, 2048, ctx=ctx, relief="plain", ties="exact", props=False)
d = ctx.asarray(fov)
seg.run_c3(d); ctx.synchronize()
lib.amt_ws_debug_reset()
t0 = time.perf_counter(); seg.run_c3(d); ctx.synchronize(); el = time.perf_counter() - t0
v = (ctypes.c_ulonglong * 8)()
lib.amt_ws_debug_read(v)
npop, cpop, npush, cpush, nr, crg, crl, ctot = [int(x) for x in v]
clk = ctot / el / 1e6 if el else 0
print(f"chain {el*1e3:.0f} ms; loop {ctot/1e6:.0f} M ticks (~{clk:.0f} MHz counter); pops {npop}, pushes {npush}, rounds {nr} ({nr/max(npop,1):.2f} per pop)")
print(f"per pop: whole pop phase {cpop/max(npop,1):.0f} ticks, of which rounds touching HBM {crg/max(npop,1):.0f}, LDS-only rounds {crl/max(npop,1):.0f}; push phase {cpush/max(npop,1):.0f} per pop ({cpush/max(npush,1):.0f} per push)")
