"""Census of the flood work of synthetic FOVs, computed with the CPU oracle (no GPU): per mask component (4-connected)
the marker count, bounding-box area with the sentinel ring, largest d2, pixel count -> the flood class it falls into."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage as ndi
from arcadia_microscopy_tools_amd import synth
from oracle import chains, skops

CLASSES = (("S", 2048, 512), ("M", 4096, 512), ("M2", 8192, 1024), ("L", 24576, 2048), ("X", 32512, 2048))
tot = {}
for fi in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    fov = synth.synth_fov(fi)
    mask, _, _ = chains.c2_mask(fov[1])
    edt = skops.distance_transform_edt(mask)
    d2 = np.rint(edt * edt).astype(np.int64)
    markers, _ = skops.peak_markers(edt, mask, 5)
    comp, nc = ndi.label(mask)
    sl = ndi.find_objects(comp)
    idx = np.arange(1, nc + 1)
    nmk = np.zeros(nc + 1, np.int64)
    mkc = comp[markers > 0]
    mkl = markers[markers > 0]
    # distinct marker labels per component
    pairs = np.unique(np.stack([mkc, mkl]), axis=1)
    np.add.at(nmk, pairs[0], 1)
    cmax = ndi.maximum(d2, comp, idx)
    size = ndi.sum(mask, comp, idx)
    stats = {"uniform": 0, "none": 0}
    for c in range(nc):
        k = nmk[c + 1]
        if k == 0:
            stats["none"] += 1
            continue
        if k == 1:
            stats["uniform"] += 1
            continue
        s = sl[c]
        area = (s[0].stop - s[0].start + 2) * (s[1].stop - s[1].start + 2)
        for name, px, nb in CLASSES:
            if area <= px and cmax[c] < nb:
                break
        else:
            name = "G"
        e = stats.setdefault(name, [0, 0, 0, 0, 0])
        e[0] += 1
        e[1] += area
        e[2] += int(size[c])
        e[3] += int(k)
        e[4] = max(e[4], area)
    print(f"FOV {fi}: {nc} components; ", {k: (v if isinstance(v, int) else dict(n=v[0], bbox_px_avg=v[1] // max(v[0], 1), px_avg=v[2] // max(v[0], 1), markers_avg=round(v[3] / max(v[0], 1), 1), bbox_max=v[4])) for k, v in stats.items()})
