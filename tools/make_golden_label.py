"""Golden vectors for skimage.measure.label on INTEGER images (equal-value components).

Run with the build container's conda interpreter (scikit-image 0.18.3):
    /opt/conda/bin/python3.9 tools/make_golden_label.py
Writes tests/golden/label_int_cases.npz: random multi-valued images (touching regions with different
values, diagonal contacts, single rows / columns) and the watershed label image of c2c3_256.npz, each
with skimage.measure.label(connectivity 1 and 2) and skimage.segmentation.clear_border.
"""
import os

import numpy as np
import skimage
from skimage import measure, segmentation

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")


def main():
    rng = np.random.default_rng(20261004)
    out = {"versions": f"skimage {skimage.__version__} numpy {np.__version__}"}
    cases = []
    for i in range(40):
        H, W = (int(v) for v in rng.integers(1, 48, 2))
        nval = int(rng.integers(1, 5))
        img = rng.integers(0, nval + 1, (H, W)).astype(np.int64) * int(rng.integers(1, 1000))
        cases.append(img)
    c3 = np.load(os.path.join(GOLD, "c2c3_256.npz"))
    cases.append(c3["watershed"].astype(np.int64))
    cases.append(c3["labels8"].astype(np.int64))
    for i, img in enumerate(cases):
        out[f"img_{i}"] = img
        out[f"lab2_{i}"] = measure.label(img).astype(np.int64)
        out[f"lab1_{i}"] = measure.label(img, connectivity=1).astype(np.int64)
        if img.ndim == 2 and min(img.shape) >= 1:
            out[f"cleared_{i}"] = segmentation.clear_border(img)
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(GOLD, "label_int_cases.npz"), **out)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
