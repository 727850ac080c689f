"""Randomised differential test of labelling (4 / 8-connected, masks and integer images), clear_border, relabel_sequential
and the region-property tables (morphology + intensities) against the oracle, on random images whose components range
from single pixels to blobs spanning many 64 x 64 tiles, thin diagonal structures, rings with holes, labels that touch
the frame.   usage: python tests/campaigns/fuzz_labels_props.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage as ndi
from arcadia_microscopy_tools_amd import _hip, hipops
from arcadia_microscopy_tools_amd.device import get_context
from oracle import regionprops as orp
from oracle import skops

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context()
bad = 0
MORPH = ("area", "centroid", "bbox", "area_convex", "solidity", "perimeter", "eccentricity", "axis_major_length",
         "axis_minor_length", "orientation")
for case in range(ncases):
    H, W = int(rng.integers(40, 400)), int(rng.integers(40, 500))
    yy, xx = np.mgrid[0:H, 0:W]
    kind = int(rng.integers(0, 4))
    if kind == 0:  # percolation-like noise
        m = rng.random((H, W)) < rng.uniform(0.3, 0.65)
    elif kind == 1:  # smooth blobs
        m = ndi.gaussian_filter(rng.random((H, W)), rng.uniform(1.5, 6.0)) > 0.5
    elif kind == 2:  # rings, diagonals, big shapes
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(2, 12))):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(3, 90)
            d = np.hypot(yy - cy, xx - cx)
            m |= (d <= r) & (d >= r - rng.integers(1, 6)) if rng.random() < 0.5 else d <= r
        for _ in range(int(rng.integers(0, 4))):
            o = int(rng.integers(-W, W))
            m |= np.abs(yy - xx - o) <= rng.integers(0, 2)
    else:  # dense mask with thin background cracks
        m = ndi.gaussian_filter(rng.random((H, W)), 2.0) > 0.47
    ok = True
    dm = ctx.asarray(m)
    for conn in (1, 2):
        lab, cnt = hipops.label(dm, connectivity=conn)
        ref = skops.label(m, conn)
        ok &= np.array_equal(lab.numpy(), ref) and int(cnt.numpy()[0]) == int(ref.max())
    lab8 = skops.label(m, 2).astype(np.int32)
    # integer image: components of EQUAL values
    vals = (lab8 % 3 + 1) * (lab8 > 0)
    li, ci = hipops.label(ctx.asarray(vals.astype(np.int32)), connectivity=2)
    ok &= np.array_equal(li.numpy(), skops.label(vals, 2))
    # clear_border + relabel
    cleared = skops.clear_border(lab8)
    got_c = hipops.clear_border(ctx.asarray(lab8)).numpy()
    ok &= np.array_equal(got_c, cleared)
    K = int(lab8.max())
    if cleared.max() > 0:
        rl, rc = hipops.relabel_sequential(ctx.asarray(cleared.astype(np.int32)), max(K, 1))
        refr = skops.relabel_sequential(cleared)
        ok &= np.array_equal(rl.numpy(), refr) and int(rc.numpy()[0]) == int(refr.max())
    # region properties of up to 400 labels (the oracle loops over regions in Python)
    if 0 < K <= 400:
        chans = rng.integers(0, 65536, (3, H, W)).astype(np.uint16)
        t, it = hipops.regionprops_full(ctx.asarray(lab8[None]), ctx.asarray(chans[None]), K)
        t, it = t.numpy()[0], it.numpy()[0]
        ref = orp.regionprops_table(lab8, None, ("label",) + MORPH)
        cols = {c: t[:, i] for i, c in enumerate(_hip.RP_COLS)}
        for c in ("area", "area_convex", "bbox-0", "bbox-1", "bbox-2", "bbox-3"):
            ok &= np.array_equal(cols[c], ref[c])
        for c in ("centroid-0", "centroid-1", "perimeter", "solidity", "axis_major_length", "axis_minor_length"):
            ok &= np.allclose(cols[c], ref[c], rtol=1e-9, atol=1e-9)
        ok &= np.allclose(cols["eccentricity"], ref["eccentricity"], atol=1e-6)
        for c in range(3):
            ri = orp.regionprops_table(lab8, chans[c], ("label",) + orp.INTENSITY_PROPS)
            ok &= np.allclose(it[:, c, 0], ri["intensity_mean"], rtol=1e-12)
            ok &= np.array_equal(it[:, c, 1], ri["intensity_max"]) and np.array_equal(it[:, c, 2], ri["intensity_min"])
            ok &= np.allclose(it[:, c, 3], ri["intensity_std"], rtol=1e-9, atol=1e-9)
    print(case, (H, W), "kind", kind, "labels", K, "ok", bool(ok), flush=True)
    bad += not ok
print("BAD", bad)
