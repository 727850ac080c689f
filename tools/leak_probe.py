"""Do repeated reference-level calls hold on to memory?  Host RSS, the context's cached device bytes and the page-locked
result pool after every 100 rounds of segment() + SegmentationMask(...).cell_properties + the two preprocessing
operators on one 2048^2 field of view (results dropped each round).  Device memory in use is read from rocm-smi."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil
from arcadia_microscopy_tools_amd import device as dv, synth
from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, DAPI, FITC, TRITC
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.masks import SegmentationMask
from arcadia_microscopy_tools_amd.model import SegmentationModel
from arcadia_microscopy_tools_amd.operations import rescale_by_percentile, subtract_background_dog


def vram_used_mb():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True, timeout=20).stdout
        row = [l for l in out.splitlines() if l and l[0].isdigit() or l.startswith("card")][-1].split(",")
        return int(row[2]) / 2**20
    except Exception:
        return float("nan")


fov = synth.synth_fov(3)
chans = {BRIGHTFIELD: fov[0], DAPI: fov[1], FITC: fov[2], TRITC: fov[3]}
model = SegmentationModel(backend="classical")
proc = psutil.Process()
ctx = get_context()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.time()
for i in range(rounds + 1):
    labels = model.segment(fov[1])
    props = SegmentationMask(labels, chans, outline_extractor="skimage").cell_properties
    a = subtract_background_dog(fov[1])
    b = rescale_by_percentile(fov[1])
    del labels, props, a, b
    if i % 100 == 0:
        pool = dv._result_pool
        print(f"round {i:4d}: rss {proc.memory_info().rss / 2**20:8.1f} MB, device cache {ctx._pool_bytes / 2**20:7.1f} MB, "
              f"pinned out {pool.out / 2**20:6.1f} kept {pool.kept / 2**20:6.1f} MB, vram {vram_used_mb():8.1f} MB, "
              f"{time.time() - t0:5.1f} s", flush=True)
