"""Step statistics of the batch flood (ws_flood_batch_kernel): build the instrumented library first,

    cd arcadia_microscopy_tools_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
        -fno-fast-math -DWS_STATS -c amt_watershed.hip -o /tmp/ws_dbg.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libamt_hip_dbg.so $(ls build/*.o | grep -v watershed) /tmp/ws_dbg.o

then run this on the GPU box.  Round 2 (4 synthetic FOVs): 805 flooded components, 61 steps per component, 13.5 pops
per step, 1.7 destination buckets per step."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libamt_hip_dbg.so")
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter
ctx = get_context()
lib = _hip.load_library()
fovs = np.stack([synth.synth_fov(i) for i in range(4)])
seg = FovSegmenter(4, 4, 2048, 2048, ctx=ctx)
d = ctx.asarray(fovs)
seg.run_c3(d); ctx.synchronize()
lib.amt_ws_debug_reset()
seg.run_c3(d); ctx.synchronize()
v = (ctypes.c_ulonglong * 8)()
lib.amt_ws_debug_read(v)
steps, pops, loops, comps, npx, nb = [int(x) for x in v[:6]]
print(f"components {comps}, tile px {npx} (avg {npx/max(comps,1):.0f}), buckets avg {nb/max(comps,1):.0f}")
print(f"steps {steps}, pops {pops}, avg batch {pops/max(steps,1):.2f}, push-loop iterations {loops} ({loops/max(steps,1):.2f} per step), steps per component {steps/max(comps,1):.1f}")
