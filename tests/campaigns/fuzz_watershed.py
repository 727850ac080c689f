"""Randomised differential test of the EDT watershed (every flood class: blobs from a few hundred to > 100,000 pixels of
bounding box, thin and thick) against the oracle's heap flood of the seeded relief; the planes of a case go through
ONE batch call and through the fused clear_border + relabel tail.   usage: python tests/campaigns/fuzz_watershed.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
from oracle import skops
from oracle.watershed import watershed

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context()
bad = 0
for case in range(ncases):
    H, W = int(rng.integers(200, 520)), int(rng.integers(260, 900))
    if case % 2:
        W = (W + 15) // 16 * 16  # widths the run-table path of the fused tail takes (with a marker list, below)
    yy, xx = np.mgrid[0:H, 0:W]
    planes = []
    for b in range(2):
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(3, 40))):
            cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
            scale = float(rng.choice([6, 10, 16, 30, 60, 110]))
            ry, rx = rng.uniform(0.4, 1.0) * scale, rng.uniform(0.4, 1.0) * scale
            if rng.random() < 0.2:
                m[max(0, cy - int(ry)): cy + int(ry), max(0, cx - int(rx)): cx + int(rx)] = True
            else:
                m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        if rng.random() < 0.5:  # keep the frame clear in half of the planes
            m[0] = m[-1] = False
            m[:, 0] = m[:, -1] = False
        planes.append(m)
    masks = np.stack(planes)
    dm = ctx.asarray(masks)
    d2, _ = hipops.edt(dm)
    refs, mks = [], []
    md = int(rng.integers(2, 8))
    for b in range(2):
        edt = skops.distance_transform_edt(masks[b])
        markers, _ = skops.peak_markers(edt, masks[b], md)
        mks.append(markers.astype(np.int32))
        refs.append(watershed(skops.seeded_flood_image(edt, markers), markers, mask=masks[b]) if markers.max() else
                    np.zeros((H, W), np.int64))
    dmk = ctx.asarray(np.stack(mks))
    got = hipops.watershed_edt(d2, dmk, dm, seeds_first=True).numpy()
    ok = np.array_equal(got, np.stack(refs))
    nl = ctx.asarray(np.array([m.max() for m in mks], np.int32))
    K = int(max(1, max(m.max() for m in mks)))
    lab, cnt = hipops.watershed_edt_cleared(d2, dmk, dm, nl, K, ctx.empty(masks.shape, np.int32))
    ok2 = True
    for b in range(2):
        cleared = skops.clear_border(refs[b])
        ref = skops.relabel_sequential(cleared) if cleared.max() > 0 else cleared
        ok2 &= np.array_equal(lab.numpy()[b], ref) and int(cnt.numpy()[b]) == int(ref.max())
    # the same through the marker-list entry point (the chain's; run tables instead of the parent plane when W % 16 == 0)
    cap = int(max(np.count_nonzero(m) for m in mks)) + 8
    klist, kcount = np.zeros((2, cap), np.int32), np.zeros(2, np.int32)
    for b in range(2):
        idx = np.flatnonzero(mks[b])
        klist[b, :idx.size], kcount[b] = idx, idx.size
    lab3, cnt3 = hipops.watershed_edt_cleared(d2, dmk, dm, nl, K, ctx.empty(masks.shape, np.int32),
                                              marker_list=(ctx.asarray(klist), ctx.asarray(kcount)))
    ok2 &= np.array_equal(lab3.numpy(), lab.numpy()) and np.array_equal(cnt3.numpy(), cnt.numpy())
    print(case, (H, W), "markers", [int(m.max()) for m in mks], "md", md, "watershed", ok, "fused tail", ok2, flush=True)
    bad += (not ok) + (not ok2)
print("BAD", bad)
