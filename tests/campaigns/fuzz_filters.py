"""Randomised differential test of the round-2 filter kernels against scipy (run on the GPU box):
strip kernels (grey min / max with random centred-run footprints, the fused top-hat subtraction, medians), the
LDS-DMA wide Gaussian, over random shapes (every strip / segment seam position), boundary modes and constants.
usage: python tests/campaigns/fuzz_filters.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage as ndi
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context()
bad = 0


def run_footprint(ry, rx, symmetric):
    fp = np.zeros((2 * ry + 1, 2 * rx + 1), np.uint8)
    for y in range(2 * ry + 1):
        yy = min(y, 2 * ry - y) if symmetric else y
        h = int(rng.integers(-1, rx + 1)) if not symmetric else int((yy + 1) * rx // (ry + 1))
        if h >= 0:
            fp[y, rx - h: rx + h + 1] = 1
    fp[ry, rx] = 1
    fp[0, rx] = fp[-1, rx] = 1  # keep the declared height
    fp[ry, 0] = fp[ry, -1] = 1  # and width
    fp[ry, :] = 1
    return fp


for case in range(ncases):
    H = int(rng.integers(16, 220))
    W = 8 * int(rng.integers(2, 190))
    kind = rng.integers(0, 4)
    img = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    if rng.random() < 0.3:
        img = (img // 8192 * 8192).astype(np.uint16)  # ties
    d = ctx.asarray(img)
    mode = ("reflect", "nearest", "constant")[int(rng.integers(0, 3))]
    cval = int(rng.integers(0, 65536))
    try:
        if kind == 0:  # min / max, random centred-run footprint
            ry, rx = int(rng.integers(0, 8)), int(rng.integers(0, 8))
            fp = run_footprint(ry, rx, bool(rng.integers(0, 2)))
            for op, ref in ((0, ndi.minimum_filter), (1, ndi.maximum_filter)):
                got = hipops._rank(d, fp if op == 0 else np.ascontiguousarray(fp[::-1, ::-1]), op, mode, cval, None).numpy()
                exp = ref(img, footprint=fp, mode=mode, cval=cval)
                if not np.array_equal(got, exp):
                    bad += 1
                    print("MISMATCH minmax", case, H, W, fp.shape, mode, op, int((got != exp).sum()), flush=True)
        elif kind == 1:  # white top-hat (fused subtraction), disks and squares
            r = int(rng.integers(1, 8))
            fp = hipops.disk(r) if rng.random() < 0.5 else np.ones((2 * r + 1, 2 * int(rng.integers(0, 8)) + 1), np.uint8)
            got = hipops.white_tophat(d, fp).numpy()
            exp = ndi.white_tophat(img, footprint=fp)
            if not np.array_equal(got, exp):
                bad += 1
                print("MISMATCH tophat", case, H, W, fp.shape, int((got != exp).sum()), flush=True)
        elif kind == 2:  # median
            fps = [np.ones((3, 3), np.uint8), hipops.disk(1), hipops.disk(2), np.ones((5, 5), np.uint8)]
            o = np.ones((5, 5), np.uint8); o[0, 0] = o[0, 4] = o[4, 0] = o[4, 4] = 0
            fps.append(o)
            fp = fps[int(rng.integers(0, len(fps)))]
            got = hipops.median(d, fp, mode=mode, cval=cval).numpy()
            exp = ndi.median_filter(img, footprint=fp, mode=mode, cval=cval)
            if not np.array_equal(got, exp):
                bad += 1
                print("MISMATCH median", case, H, W, fp.shape, mode, int((got != exp).sum()), flush=True)
        else:  # wide Gaussian (two-pass path: radius > 12), widths that do / do not take the LDS-DMA kernels
            W2 = 64 * int(rng.integers(4, 14)) if rng.random() < 0.7 else W
            H2 = int(rng.integers(20, 150))
            im2 = rng.integers(0, 65536, (H2, W2)).astype(np.uint16)
            sigma = float(rng.uniform(3.2, 20.0))
            gmode = ("nearest", "reflect", "mirror", "constant", "wrap")[int(rng.integers(0, 5))]
            got = hipops.gaussian(ctx.asarray(im2), sigma, mode=gmode, cval=0.125).numpy()
            exp = ndi.gaussian_filter(im2.astype(np.float64) * (1.0 / 65535), sigma, mode=gmode, cval=0.125)
            if not np.array_equal(got, exp):
                bad += 1
                print("MISMATCH gaussian", case, H2, W2, sigma, gmode, float(np.abs(got - exp).max()), flush=True)
    except Exception as e:  # an unexpected refusal is a finding too
        bad += 1
        print("ERROR", case, kind, H, W, repr(e)[:200], flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} bad", flush=True)
print("BAD", bad)
