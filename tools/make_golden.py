"""Generate tests/golden/*.npz with the REAL scikit-image / scipy of the build container.

Run with the conda interpreter that has scikit-image 0.18.3 + scipy 1.7.1 + numpy 1.26.4:

    /opt/conda/bin/python3.9 tools/make_golden.py

(the reference pins scikit-image 0.25.2, which is not installable offline -- SURVEY.md section 8c; the
0.18.3 property names are mapped to the reference's 0.25.2 names here).  Every file stores the INPUT
arrays next to the expected outputs, so the tests never depend on RNG reproducibility across numpy
versions.  The ND2 pixel block comes from the reference's own test fixture
(src/arcadia_microscopy_tools/tests/data/example-multichannel.nd2, layout: SURVEY.md A.10).
"""
import importlib.util
import os
import struct
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import scipy  # noqa: E402
import skimage  # noqa: E402
from scipy import ndimage as ndi  # noqa: E402
from skimage import exposure, filters, measure, morphology, segmentation  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF_ND2 = "/root/reference/src/arcadia_microscopy_tools/tests/data/example-multichannel.nd2"

spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "arcadia_microscopy_tools_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

VERSIONS = np.array([skimage.__version__, scipy.__version__, np.__version__])


def read_nd2_pixels(path):
    """Minimal ND2 chunk-map walk: returns (C, Y, X) uint16 of ImageDataSeq|0!."""
    with open(path, "rb") as f:
        data = f.read()
    (map_off,) = struct.unpack("<Q", data[-8:])
    magic, name_len, data_len = struct.unpack("<IIQ", data[map_off:map_off + 16])
    assert magic == 0x0ABECEDA
    payload = data[map_off + 16 + name_len: map_off + 16 + name_len + data_len]
    key = b"ImageDataSeq|0!"
    p = payload.find(key)
    off, size = struct.unpack("<QQ", payload[p + len(key): p + len(key) + 16])
    magic, name_len, data_len = struct.unpack("<IIQ", data[off:off + 16])
    assert magic == 0x0ABECEDA
    start = off + 16 + name_len + 8  # skip the 8-byte timestamp
    px = np.frombuffer(data, dtype="<u2", count=256 * 256 * 4, offset=start).reshape(256, 256, 4)
    return np.ascontiguousarray(px.transpose(2, 0, 1))


RENAME = {
    "convex_area": "area_convex",
    "major_axis_length": "axis_major_length",
    "minor_axis_length": "axis_minor_length",
    "mean_intensity": "intensity_mean",
    "max_intensity": "intensity_max",
    "min_intensity": "intensity_min",
}


def sk_props(labels, intensity=None):
    props = ["label", "area", "centroid", "bbox", "convex_area", "perimeter", "eccentricity", "solidity",
             "major_axis_length", "minor_axis_length", "orientation"]
    t = measure.regionprops_table(labels, properties=props)
    out = {("rp_" + RENAME.get(k, k)): np.asarray(v) for k, v in t.items()}
    if intensity is not None:
        ti = measure.regionprops_table(labels, intensity_image=intensity,
                                       properties=["mean_intensity", "max_intensity", "min_intensity"])
        for k, v in ti.items():
            out["rp_" + RENAME[k]] = np.asarray(v)
        # intensity_std does not exist in 0.18.3: population std of the masked pixels (0.25.2 definition)
        std = [np.std(intensity[labels == lab]) for lab in t["label"]]
        out["rp_intensity_std"] = np.asarray(std)
    return out


def gold_nd2():
    px = read_nd2_pixels(REF_ND2)
    dapi = px[1]
    out = dict(pixels=px, versions=VERSIONS)
    out["otsu"] = filters.threshold_otsu(dapi)
    out["isodata"] = filters.threshold_isodata(dapi)
    out["yen"] = filters.threshold_yen(dapi)
    out["triangle"] = filters.threshold_triangle(dapi)
    out["mean"] = filters.threshold_mean(dapi)
    out["li"] = filters.threshold_li(dapi)
    mask = dapi > out["otsu"]
    out["labels8"] = measure.label(mask).astype(np.int64)
    out["labels4"] = measure.label(mask, connectivity=1).astype(np.int64)
    out["cleared"] = segmentation.clear_border(out["labels8"])
    out.update(sk_props(out["labels8"], dapi))
    # reference ops on the fixture (R/operations.py restated by skimage calls)
    dog = filters.difference_of_gaussians(px[2], 0.6, 16.0)
    out["dog_fitc"] = dog
    out["bgsub_fitc_p90"] = np.clip(dog - np.percentile(dog, 90), 0, None)
    p1, p2 = np.percentile(px[2], (1, 99))
    out["rescale_fitc_1_99"] = exposure.rescale_intensity(px[2], in_range=(p1, p2), out_range=(0, 1))
    np.savez_compressed(os.path.join(OUT, "nd2_multichannel.npz"), **out)
    print("nd2: otsu", out["otsu"], "labels", out["labels8"].max(), "fg", mask.sum())


def gold_c2c3():
    fov = synth.synth_fov(0, size=256)
    dapi = fov[1]
    out = dict(fov=fov, versions=VERSIONS)
    g = filters.gaussian(dapi, sigma=2)
    out["gauss2"] = g
    t = filters.threshold_otsu(g)
    out["otsu_gauss2"] = t
    m0 = g > t
    se = morphology.disk(2)
    m1 = morphology.binary_opening(m0, se)
    m2 = morphology.binary_closing(m1, se)
    out["mask_thr"], out["mask_open"], out["mask"] = m0, m1, m2
    out["labels8"] = measure.label(m2).astype(np.int64)
    edt = ndi.distance_transform_edt(m2)
    out["edt"] = edt
    mm = 5
    peaks = (edt == ndi.maximum_filter(edt, size=2 * mm + 1, mode="constant")) & m2 & (edt > 0)
    peaks[:mm, :] = False
    peaks[-mm:, :] = False
    peaks[:, :mm] = False
    peaks[:, -mm:] = False
    markers = ndi.label(peaks)[0].astype(np.int32)
    out["markers"] = markers
    relief = -edt
    idx = np.flatnonzero(markers.ravel())
    M = idx.size
    base = np.floor(relief.min()) - 1.0
    r = relief.copy().ravel()
    r[idx] = base - (M - np.arange(M, dtype=np.float64))
    relief = r.reshape(edt.shape)
    out["relief"] = relief
    ws = segmentation.watershed(relief, markers, mask=m2)
    out["watershed"] = ws.astype(np.int32)
    out["watershed_plain"] = segmentation.watershed(-edt, markers, mask=m2).astype(np.int32)
    cleared = segmentation.clear_border(ws)
    out["cleared"] = cleared.astype(np.int32)
    labels = segmentation.relabel_sequential(cleared)[0].astype(np.int64)
    out["labels"] = labels
    out.update(sk_props(labels, None))
    for ci, name in enumerate(synth.CHANNEL_NAMES):
        p = sk_props(labels, fov[ci])
        for k in ("intensity_mean", "intensity_max", "intensity_min", "intensity_std"):
            out[f"rp_{k}_{name.lower()}"] = p["rp_" + k]
    np.savez_compressed(os.path.join(OUT, "c2c3_256.npz"), **out)
    print("c2c3: otsu", t, "labels8", out["labels8"].max(), "markers", markers.max(), "cells", labels.max())


def gold_watershed():
    rng = np.random.default_rng(7)
    cases = {}
    n = 0
    for i in range(60):
        H, W = rng.integers(8, 41, 2)
        kind = i % 3
        if kind == 0:
            img = rng.integers(0, rng.integers(1, 6), (H, W)).astype(np.float64)
            mask = rng.random((H, W)) < 0.85 if i % 2 else None
        elif kind == 1:
            blob = ndi.binary_dilation(rng.random((H, W)) < 0.08, iterations=rng.integers(2, 5))
            img = -ndi.distance_transform_edt(blob)
            mask = blob
        else:
            img = ndi.gaussian_filter(rng.random((H, W)), 1.5)
            mask = None
        markers = np.zeros((H, W), np.int32)
        k = rng.integers(1, 8)
        ys = rng.integers(0, H, k)
        xs = rng.integers(0, W, k)
        markers[ys, xs] = np.arange(1, k + 1)
        if i % 5 == 0 and H > 4:  # a multi-pixel marker
            markers[H // 2, W // 2:W // 2 + 3] = k + 1
        for conn in (1, 2):
            ws = segmentation.watershed(img, markers, connectivity=conn, mask=mask)
            cases[f"img_{n}"] = img
            cases[f"markers_{n}"] = markers
            cases[f"mask_{n}"] = np.ones((H, W), bool) if mask is None else mask
            cases[f"conn_{n}"] = conn
            cases[f"out_{n}"] = ws.astype(np.int32)
            n += 1
    cases["n"] = n
    cases["versions"] = VERSIONS
    np.savez_compressed(os.path.join(OUT, "watershed_cases.npz"), **cases)
    print("watershed cases:", n)


def gold_ops():
    fov = synth.synth_fov(3, size=192)
    u = fov[1]
    out = dict(u16=u, versions=VERSIONS)
    for s in (0.6, 1.0, 2.0, 5.0):
        out[f"gauss_{s}"] = filters.gaussian(u, sigma=s)
    out["dog_0.6_16"] = filters.difference_of_gaussians(u, 0.6, 16.0)
    out["dog_1_4"] = filters.difference_of_gaussians(u, 1.0, 4.0)
    g = out["gauss_2.0"]
    for name in ("otsu", "yen", "isodata", "triangle", "mean", "li"):
        f = getattr(filters, "threshold_" + name)
        out[f"thr_{name}_u16"] = f(u)
        out[f"thr_{name}_f64"] = f(g)
    try:
        out["thr_minimum_u16"] = filters.threshold_minimum(u)
    except RuntimeError:
        out["thr_minimum_u16"] = np.nan
    out["thr_local_35"] = filters.threshold_local(u, 35)
    out["thr_local_35_mean"] = filters.threshold_local(u, 35, method="mean")
    out["thr_niblack_15"] = filters.threshold_niblack(u, window_size=15, k=0.2)
    out["thr_sauvola_15"] = filters.threshold_sauvola(u, window_size=15, k=0.2)
    hist, centers = exposure.histogram(g, nbins=256)
    out["hist_f64"], out["hist_f64_centers"] = hist, centers
    for q in ((0, 100), (1, 99), (0.1, 99.9), (2, 98)):
        p1, p2 = np.percentile(u, q)
        out[f"pct_u16_{q[0]}_{q[1]}"] = np.array([p1, p2])
        out[f"rescale_u16_{q[0]}_{q[1]}"] = exposure.rescale_intensity(u, in_range=(p1, p2), out_range=(0, 1))
        p1, p2 = np.percentile(g, q)
        out[f"pct_f64_{q[0]}_{q[1]}"] = np.array([p1, p2])
        out[f"rescale_f64_{q[0]}_{q[1]}"] = exposure.rescale_intensity(g, in_range=(p1, p2), out_range=(0, 1))
    m = u > filters.threshold_otsu(u)
    out["mask"] = m
    for r in (1, 2, 3):
        se = morphology.disk(r)
        out[f"berode_d{r}"] = morphology.binary_erosion(m, se)
        out[f"bdilate_d{r}"] = morphology.binary_dilation(m, se)
        out[f"bopen_d{r}"] = morphology.binary_opening(m, se)
        out[f"bclose_d{r}"] = morphology.binary_closing(m, se)
        out[f"erode_d{r}"] = morphology.erosion(u, se)
        out[f"dilate_d{r}"] = morphology.dilation(u, se)
        out[f"open_d{r}"] = morphology.opening(u, se)
        out[f"close_d{r}"] = morphology.closing(u, se)
        out[f"median_d{r}"] = filters.median(u, se)
    out["berode_cross"] = morphology.binary_erosion(m)
    out["bdilate_cross"] = morphology.binary_dilation(m)
    out["tophat_d7"] = morphology.white_tophat(u, morphology.disk(7))
    out["tophat_d3"] = morphology.white_tophat(u, morphology.disk(3))
    out["median_3x3"] = filters.median(u)
    out["edt"] = ndi.distance_transform_edt(m)
    out["label8"] = measure.label(m).astype(np.int64)
    out["label4"] = measure.label(m, connectivity=1).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "ops_192.npz"), **out)
    print("ops: done")


def gold_disks():
    """The shapes the reference's own tests build (RT/test_masks.py:14-30) + known answers."""
    from skimage.draw import disk

    img = np.zeros((80, 80), dtype=np.int64)
    for lab, (c, r) in enumerate([((15, 15), 5), ((40, 40), 8), ((62, 60), 11)], start=1):
        rr, cc = disk(c, r, shape=img.shape)
        img[rr, cc] = lab
    out = dict(labels=img, versions=VERSIONS)
    out.update(sk_props(img, None))
    np.savez_compressed(os.path.join(OUT, "disks_80.npz"), **out)
    print("disks: area", out["rp_area"], "perimeter", out["rp_perimeter"], "convex", out["rp_area_convex"])


def outline_label_image():
    """96 x 96 label image with the shapes the outline extractor has to survive: blobs, a ring (hole), a hole
    longer than its outer boundary, diagonal-only contact, cells clipped by every image edge, a single pixel,
    a label made of two separate pieces, a thin line, and a label number gap."""
    lab = np.zeros((96, 96), dtype=np.int64)
    yy, xx = np.mgrid[0:96, 0:96]
    lab[(yy - 20) ** 2 + (xx - 20) ** 2 <= 81] = 1                          # disk
    ring = ((yy - 20) ** 2 + (xx - 55) ** 2 <= 144) & ((yy - 20) ** 2 + (xx - 55) ** 2 > 25)
    lab[ring] = 2                                                           # ring: outer + hole contour
    lab[40:60, 5:40] = 3                                                    # rectangle with a comb-shaped hole
    for k in range(8, 37, 4):
        lab[43:57, k:k + 2] = 0
    lab[44:46, 8:38] = 0
    lab[70, 10] = 4                                                         # single pixel
    lab[72:75, 20:23] = 5
    lab[75:78, 23:26] = 5                                                   # two squares touching diagonally
    lab[0:6, 30:42] = 6                                                     # clipped by the top edge
    lab[85:96, 0:7] = 7                                                     # bottom-left corner
    lab[60:75, 90:96] = 9                                                   # right edge (label 8 is unused)
    lab[80:83, 40:43] = 10
    lab[86:90, 50:55] = 10                                                  # one label, two pieces
    lab[64, 45:70] = 11                                                     # horizontal line, 1 px thick
    lab[70:92, 62] = 12                                                     # vertical line
    rng = np.random.default_rng(3)
    noise = rng.random((20, 24)) < 0.55
    lab[30:50, 66:90][noise] = 13                                           # ragged random blob with many holes
    return lab


def gold_outlines():
    """find_contours case table (all 16 squares) and ``_extract_outlines_skimage`` (R/masks.py:82-115)."""
    from skimage.measure import _find_contours_cy as cy

    out = {"versions": VERSIONS}
    table = np.full((16, 2, 2, 2), np.nan)
    for case in range(16):
        img = np.array([[case & 1, (case >> 1) & 1], [(case >> 2) & 1, (case >> 3) & 1]], dtype=np.double)
        for k, (f, t) in enumerate(cy._get_contour_segments(img, 0.5, False, None)):
            table[case, k, 0] = f
            table[case, k, 1] = t
    out["case_table"] = table
    images = {"shapes": outline_label_image()}
    fov = synth.synth_fov(7, size=256)
    g = filters.gaussian(fov[1], sigma=2.0)
    m = g > filters.threshold_otsu(g)
    m = morphology.binary_closing(morphology.binary_opening(m, morphology.disk(2)), morphology.disk(2))
    images["nuclei"] = measure.label(m).astype(np.int64)
    for name, lab in images.items():
        h, w = lab.shape
        pts, offs = [], [0]
        for region in measure.regionprops(lab):
            minr, minc, maxr, maxc = region.bbox
            r0, c0, r1, c1 = max(minr - 1, 0), max(minc - 1, 0), min(maxr + 1, h), min(maxc + 1, w)
            crop = (lab[r0:r1, c0:c1] == region.label).astype(np.uint8)
            contours = measure.find_contours(crop, level=0.5)
            if contours:
                main = max(contours, key=len)
                main = main + np.array([r0, c0])
                pts.append(main)
                offs.append(offs[-1] + len(main))
            else:
                offs.append(offs[-1])
        out[f"{name}_labels"] = lab
        out[f"{name}_points"] = np.concatenate(pts, axis=0) if pts else np.zeros((0, 2))
        out[f"{name}_offsets"] = np.array(offs, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "outlines_96.npz"), **out)
    print("outlines:", {k: v.shape for k, v in out.items() if k.endswith("points")})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["nd2", "c2c3", "watershed", "ops", "disks", "outlines"]
    for w in which:
        globals()["gold_" + w]()
