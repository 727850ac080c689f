"""Tiny and ragged shapes (1 x 1 up to ~70 x 90, every width residue, planes shorter than a filter's reach) through
every operator of the path, against the CPU oracle / scipy.  The fast kernels have minimum sizes and alignment rules;
this sweep lives in the fallbacks and at the switch-over points.  Usage: fuzz_tiny.py [cases] [seed]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage as ndi
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
from arcadia_microscopy_tools_amd.operations import apply_threshold, rescale_by_percentile, subtract_background_dog
from oracle import skops, regionprops as orp

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context()
bad = 0
counts = {}


def check(name, ok, info):
    global bad
    counts[name] = counts.get(name, 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH", name, info, flush=True)


def blobs(H, W):
    m = np.zeros((H, W), bool)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(int(rng.integers(1, 6))):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, max(2, min(H, W) // 2 + 1))
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    if rng.random() < 0.3:
        m |= rng.random((H, W)) < 0.15
    return m


for case in range(ncases):
    if rng.random() < 0.35:
        H, W = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    else:
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    shp = (H, W)
    img = rng.integers(0, 65536, shp).astype(np.uint16)
    if rng.random() < 0.3:  # smooth content: ties and plateaus
        img = (ndi.uniform_filter(img.astype(np.float64), 3) // 257 * 257).astype(np.uint16)
    d = ctx.asarray(img)
    # Gaussians
    for sigma in (0.6, 2.0):
        mode = ("nearest", "reflect", "mirror", "constant")[int(rng.integers(0, 4))]
        got = hipops.gaussian(d, sigma, mode=mode).numpy()
        exp = skops.gaussian(img, sigma, mode=mode)
        check("gaussian", np.array_equal(got, exp), (shp, sigma, mode))
    got = subtract_background_dog(img, low_sigma=0.6, high_sigma=float(rng.choice([1.5, 4.0, 16.0])))
    check("dog finite", got.shape == shp and np.isfinite(got).all() and (got >= 0).all(), shp)
    dog = skops.difference_of_gaussians(img, 0.6, 3.0)
    exp = np.clip(dog - np.percentile(dog, 0), 0, None)
    check("dog", np.array_equal(subtract_background_dog(img, low_sigma=0.6, high_sigma=3.0), exp), shp)
    # percentile rescale
    lo, hi = sorted(rng.uniform(0, 100, 2))
    if hi - lo > 1e-3:
        got = rescale_by_percentile(img, percentile_range=(lo, hi))
        if img.min() == img.max():
            exp = np.zeros(shp)
        else:
            p1, p2 = np.percentile(img, (lo, hi))
            exp = skops.rescale_intensity(img, (p1, p2), (0.0, 1.0)) if p1 != p2 else None
        if exp is not None:
            check("rescale", np.array_equal(got, exp), (shp, lo, hi))
    # thresholds
    for method in ("otsu", "li", "yen", "isodata", "triangle", "mean"):
        got = apply_threshold(img, method=method)
        if img.min() == img.max():
            exp = np.zeros(shp, bool)
        else:
            exp = img > getattr(skops, "threshold_" + method)(img)
        check("threshold " + method, np.array_equal(got, exp), shp)
    w = int(rng.choice([3, 5, 9, 15]))
    for method in ("niblack", "sauvola"):
        got = apply_threshold(img, method=method, window_size=w)
        exp = np.zeros(shp, bool) if img.min() == img.max() else \
            img > getattr(skops, "threshold_" + method)(img, window_size=w)
        check("threshold " + method, np.array_equal(got, exp), (shp, w))
    b = int(rng.choice([3, 5, 11]))
    got = apply_threshold(img, method="local", block_size=b)
    exp = np.zeros(shp, bool) if img.min() == img.max() else img > skops.threshold_local(img, b)
    check("threshold local", np.array_equal(got, exp), (shp, b))
    # grey morphology / median
    fps = [skops.disk(1), skops.disk(2), skops.disk(3), np.ones((3, 3), np.uint8), np.ones((1, 5), np.uint8),
           np.ones((5, 1), np.uint8), np.ones((2, 2), np.uint8), np.ones((4, 3), np.uint8), skops.disk(7)]
    fp = fps[int(rng.integers(0, len(fps)))]
    for name in ("erosion", "dilation", "opening", "closing", "white_tophat"):
        got = getattr(hipops, name)(d, fp).numpy()
        exp = getattr(skops, name)(img, fp)
        check(name, np.array_equal(got, exp), (shp, fp.shape))
    mfp = fps[int(rng.integers(0, 6))]
    mode = ("nearest", "reflect", "constant")[int(rng.integers(0, 3))]
    check("median", np.array_equal(hipops.median(d, mfp, mode=mode).numpy(), skops.median(img, mfp, mode=mode)),
          (shp, mfp.shape, mode))
    # binary morphology, labels, EDT, props
    m = blobs(H, W)
    dm = ctx.asarray(m)
    for name in ("binary_erosion", "binary_dilation", "binary_opening", "binary_closing"):
        bfp = (skops.disk(1), skops.disk(2), np.ones((3, 3), np.uint8))[int(rng.integers(0, 3))]
        check(name, np.array_equal(getattr(hipops, name)(dm, bfp).numpy(), getattr(skops, name)(m, bfp)), (shp, bfp.shape))
    for conn in (1, 2):
        lab, cnt = hipops.label(dm, connectivity=conn)
        exp = skops.label(m, connectivity=conn)
        check("label", np.array_equal(lab.numpy(), exp) and int(cnt.numpy()[0]) == exp.max(), (shp, conn))
    ilab = (rng.integers(0, 4, shp) * m).astype(np.int32)
    lab, cnt = hipops.label(ctx.asarray(ilab), connectivity=2)
    exp = skops.label(ilab, connectivity=2)
    check("label int", np.array_equal(lab.numpy(), exp), shp)
    if m.any():
        d2, e = hipops.edt(dm)
        exp = skops.distance_transform_edt(m)
        check("edt", np.array_equal(e.numpy(), exp) and np.array_equal(d2.numpy(), np.rint(exp * exp).astype(np.int64)), shp)
    lab8 = skops.label(m, connectivity=2).astype(np.int32)
    k = int(lab8.max())
    if k:
        cb = hipops.clear_border(ctx.asarray(lab8)).numpy()
        check("clear_border", np.array_equal(cb, skops.clear_border(lab8)), shp)
        chans = rng.integers(0, 65536, (2, H, W)).astype(np.uint16)
        mt, it = hipops.regionprops_full(ctx.asarray(lab8[None]), ctx.asarray(chans[None]), k)
        from arcadia_microscopy_tools_amd.segment import assemble_cell_properties
        got = assemble_cell_properties(mt.numpy()[0][:k], it.numpy()[0][:k], ("A", "B"))
        exp = orp.cell_properties(lab8.astype(np.int64), {"A": chans[0], "B": chans[1]})
        okp = all(np.allclose(got[c], exp[c], rtol=1e-5, atol=1e-8, equal_nan=True) for c in exp if c != "orientation")
        okp = okp and np.array_equal(got["area"], exp["area"]) and np.array_equal(got["label"], exp["label"])
        if not okp:
            for c in exp:
                if not np.allclose(got[c], exp[c], rtol=1e-5, atol=1e-8, equal_nan=True):
                    print("   column", c, got[c][:6], exp[c][:6])
        check("regionprops", okp, (shp, k))
    if case % 50 == 49:
        print(f"{case + 1}/{ncases} cases, bad {bad}", flush=True)
# config 3 end to end on small, non-square windows of synthetic FOVs (nuclei cut by the frame, a handful of cells)
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.segment import segment_fovs
from oracle import chains
for case in range(max(10, ncases // 10)):
    H2, W2 = int(rng.integers(12, 160)), int(rng.integers(12, 160))
    big = synth.synth_fov(3000 + case, size=192)
    y0, x0 = int(rng.integers(0, 192 - H2 + 1)), int(rng.integers(0, 192 - W2 + 1))
    fovs = np.ascontiguousarray(np.stack([big[:, y0:y0 + H2, x0:x0 + W2], big[:, :H2, :W2]]))
    res = segment_fovs(fovs, max_cells=512)
    lab, tabs = res.labels_numpy(), res.feature_tables()
    for j in range(2):
        rl, rpp = chains.c3_chain(fovs[j])
        check("c3 labels", np.array_equal(lab[j], rl), ((H2, W2), j))
        check("c3 props", all(np.allclose(tabs[j][c], rpp[c], rtol=1e-5, atol=1e-8) for c in rpp if c != "orientation"),
              ((H2, W2), j, int(rl.max())))
print({k: v for k, v in sorted(counts.items())})
print("BAD", bad)
sys.exit(1 if bad else 0)
