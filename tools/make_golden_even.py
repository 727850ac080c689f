"""Golden vectors for EVEN-sized footprints from the real scikit-image 0.18.3 / scipy 1.7.1 (run with the build
container's conda interpreter: /opt/conda/bin/python3.9 tools/make_golden_even.py).  scikit-image pads an even grey
footprint with a zero row / column in front (erosion, dilation) or behind (second half of opening / closing); the binary
operators, the median and white_tophat hand the even footprint to scipy as it is."""
import numpy as np
import skimage
from skimage import filters, morphology

rng = np.random.default_rng(123)
img = rng.integers(0, 65536, (40, 56)).astype(np.uint16)
img[10:20, 10:30] //= 8
mask = rng.random((40, 56)) < 0.55
out = {"img": img, "mask": mask, "skimage_version": np.array(skimage.__version__)}
fps = {"s2": np.ones((2, 2), np.uint8), "s4": np.ones((4, 4), np.uint8), "r2x3": np.ones((2, 3), np.uint8),
       "r3x4": np.ones((3, 4), np.uint8), "c4x1": np.ones((4, 1), np.uint8),
       "L4": np.array([[1, 0, 0, 0], [1, 0, 0, 0], [1, 1, 1, 0], [0, 0, 1, 1]], np.uint8)}
for name, fp in fps.items():
    out[f"fp_{name}"] = fp
    out[f"erosion_{name}"] = morphology.erosion(img, fp)
    out[f"dilation_{name}"] = morphology.dilation(img, fp)
    out[f"opening_{name}"] = morphology.opening(img, fp)
    out[f"closing_{name}"] = morphology.closing(img, fp)
    out[f"tophat_{name}"] = morphology.white_tophat(img, fp)
    out[f"median_{name}"] = filters.median(img, fp)
    out[f"berosion_{name}"] = morphology.binary_erosion(mask, fp)
    out[f"bdilation_{name}"] = morphology.binary_dilation(mask, fp)
    out[f"bopening_{name}"] = morphology.binary_opening(mask, fp)
    out[f"bclosing_{name}"] = morphology.binary_closing(mask, fp)
np.savez_compressed("tests/golden/even_footprints.npz", **out)
print("wrote tests/golden/even_footprints.npz with", len(out), "arrays; scikit-image", skimage.__version__)
