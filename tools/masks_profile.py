"""Where the one-call API route (SegmentationModel.batch_masks) spends its time: FOV/s for worker threads x images per
chunk x FOVs per call, and the host-side split of a call (send = fill + submit, collect = wait + rows, enqueue)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
from bench import synth_fovs  # noqa: E402
from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, DAPI, FITC, TRITC  # noqa: E402
from arcadia_microscopy_tools_amd.maskbatch import MaskBatcher  # noqa: E402
from arcadia_microscopy_tools_amd.model import SegmentationModel  # noqa: E402

S = 2048
uniq = synth_fovs(list(range(8)), S)
chans = (BRIGHTFIELD, DAPI, FITC, TRITC)
model = SegmentationModel(backend="classical")
MaskBatcher.times = {}

for B, workers, chunk in [(48, 1, 4), (48, 2, 4), (48, 2, 6), (48, 4, 4), (48, 4, 2), (48, 3, 4), (192, 2, 4), (192, 2, 8),
                          (192, 4, 4), (192, 4, 8), (192, 1, 8)]:
    fovs = [uniq[i % 8] for i in range(B)]
    share = -(-B // workers)
    shares = [fovs[i:i + share] for i in range(0, B, share)]
    acc = {}

    def mwork(part):
        t0 = time.perf_counter()
        masks = model.batch_masks(part, chans, nuclear=DAPI, batch_size=chunk)
        t1 = time.perf_counter()
        props = [m.cell_properties for m in masks]
        t2 = time.perf_counter()
        acc["batch_masks"] = acc.get("batch_masks", 0.0) + t1 - t0
        acc["cell_properties"] = acc.get("cell_properties", 0.0) + t2 - t1
        return props

    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(mwork, shares))
        list(ex.map(mwork, shares))
        MaskBatcher.times.clear()
        acc.clear()
        steps = 3 if B < 100 else 2
        t0 = time.perf_counter()
        for _ in range(steps):
            list(ex.map(mwork, shares))
        el = time.perf_counter() - t0
    n = B * steps
    split = ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in {**acc, **MaskBatcher.times}.items())
    print(f"B {B:3d} workers {workers} chunk {chunk}: {n / el:7.1f} FOV/s; thread-ms per FOV: {split}", flush=True)
