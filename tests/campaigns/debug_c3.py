"""Stage-by-stage comparison of the device config-3 chain with the oracle for given synthetic FOVs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.segment import FovSegmenter
from oracle import chains

size = int(sys.argv[1]); idxs = [int(v) for v in sys.argv[2:]]
ctx = get_context()
for i in idxs:
    fov = synth.synth_fov(i, size=size)
    seg = FovSegmenter(1, 4, size, size, ctx=ctx, max_cells=512, fused=False)
    lab = seg.run_c3(ctx.asarray(fov[None])).numpy()[0].astype(np.int64)
    ref, inter = chains.c3_labels(fov[1])
    res = {
        "mask": np.array_equal(seg.mask_a.numpy()[0].astype(bool), inter["mask"]),
        "d2": np.array_equal(np.sqrt(seg.d2.numpy()[0].astype(np.float64)), inter["edt"]),
        "markers": np.array_equal(seg.markers.numpy()[0], inter["markers"]),
        "ws": np.array_equal(seg.ws.numpy()[0], inter["watershed"]),
        "labels": np.array_equal(lab, ref),
    }
    print(i, res, "nmarkers", seg.nmarkers.numpy(), int(inter["markers"].max()), "ncells", seg.ncells.numpy(), int(ref.max()))
    if not res["labels"]:
        for k, (a, b) in {"markers": (seg.markers.numpy()[0], inter["markers"]), "ws": (seg.ws.numpy()[0], inter["watershed"]), "labels": (lab, ref)}.items():
            d = np.argwhere(a != b)
            print("  ", k, "ndiff", len(d), d[:5].tolist(), [(int(a[y, x]), int(b[y, x])) for y, x in d[:5]])
