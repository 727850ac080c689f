"""Single-image latency of the reference-level API (host numpy in, host numpy out) on one 2048^2 field of view."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.channels import DAPI, FITC, TRITC, BRIGHTFIELD
from arcadia_microscopy_tools_amd.masks import SegmentationMask
from arcadia_microscopy_tools_amd.model import SegmentationModel
from arcadia_microscopy_tools_amd.operations import rescale_by_percentile, subtract_background_dog, apply_threshold

fov = synth.synth_fov(3)
dapi = fov[1]
model = SegmentationModel(backend="classical")
chans = {BRIGHTFIELD: fov[0], DAPI: fov[1], FITC: fov[2], TRITC: fov[3]}


def timeit(name, fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {dt*1e3:8.2f} ms")
    return out


labels = timeit("segment (u16 2048^2 -> int64)", lambda: model.segment(dapi))
timeit("SegmentationMask + cell_properties", lambda: SegmentationMask(labels, chans, outline_extractor="skimage").cell_properties)
timeit("rescale_by_percentile (u16)", lambda: rescale_by_percentile(dapi))
timeit("subtract_background_dog (u16)", lambda: subtract_background_dog(dapi))
timeit("apply_threshold otsu (u16)", lambda: apply_threshold(dapi))
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        SegmentationMask(labels, chans, outline_extractor="skimage").cell_properties
        model.segment(dapi)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
