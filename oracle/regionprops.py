"""Oracle for ``skimage.measure.regionprops_table`` as the reference uses it.  TEST INFRASTRUCTURE ONLY.

Follows SK/measure/_regionprops.py (0.18.3; per-region python loop over ``ndi.find_objects``,
bbox-cropped masks, local moments) with the property NAMES of the reference's pinned 0.25.2
(R/masks.py:15-35: area_convex, axis_major_length, intensity_mean, ... -- SURVEY.md A.9), and
R/masks.py:247-328 for the derived columns (circularity, volume, key renames).
"""
from __future__ import annotations

import itertools
from math import atan2, pi, sqrt

import numpy as np
from scipy import ndimage as ndi

STREL_4 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)


def perimeter(image: np.ndarray) -> float:
    """SK/measure/_regionprops_utils.py:186-249 with neighbourhood=4."""
    image = image.astype(np.uint8)
    eroded = ndi.binary_erosion(image, STREL_4, border_value=0)
    border = image - eroded
    w = np.zeros(50, dtype=np.double)
    w[[5, 7, 15, 17, 25, 27]] = 1
    w[[21, 33]] = sqrt(2)
    w[[13, 23]] = (1 + sqrt(2)) / 2
    pim = ndi.convolve(border, np.array([[10, 2, 10], [2, 1, 2], [10, 2, 10]]), mode="constant", cval=0)
    hist = np.bincount(pim.ravel(), minlength=50)
    return float(hist @ w)


def moments_central(image: np.ndarray, center, order: int = 3) -> np.ndarray:
    """SK/measure/_moments.py:195-250."""
    calc = image.astype(float)
    for dim, dim_length in enumerate(image.shape):
        delta = np.arange(dim_length, dtype=float) - center[dim]
        powers = delta[:, np.newaxis] ** np.arange(order + 1)
        calc = np.rollaxis(calc, dim, image.ndim)
        calc = np.dot(calc, powers)
        calc = np.rollaxis(calc, -1, dim)
    return calc


def inertia_tensor(mu: np.ndarray) -> np.ndarray:
    """SK/measure/_moments.py:379-428 in 2-D."""
    mu0 = mu[0, 0]
    result = np.zeros((2, 2))
    corners = np.array([mu[2, 0], mu[0, 2]])
    result[0, 0], result[1, 1] = (np.sum(corners) - corners) / mu0
    result[0, 1] = result[1, 0] = -mu[1, 1] / mu0
    return result


def inertia_tensor_eigvals(T: np.ndarray):
    """SK/measure/_moments.py:431-470."""
    eigvals = np.linalg.eigvalsh(T)
    eigvals = np.clip(eigvals, 0, None, out=eigvals)
    return sorted(eigvals, reverse=True)


def convex_area_exact(mask: np.ndarray) -> int:
    """``area_convex``: number of pixel centres inside or on the convex hull of the pixel diamonds.

    SK/morphology/convex_hull.py: hull candidates are the first/last set pixel of every row and
    column, each expanded to its four edge mid-points (r+-0.5, c), (r, c+-0.5); Qhull hull; the hull
    image is every pixel centre inside or on the polygon (0.25.2: ``grid_points_in_poly(...,
    binarize=False) >= 1`` with include_borders=True).  All coordinates are half-integers, so the
    computation is restated in exact integer arithmetic on doubled coordinates.
    """
    ys, xs = np.nonzero(mask)
    if ys.size == 0:
        return 0
    pts = set()
    for y in np.unique(ys):
        row = xs[ys == y]
        for x in (row.min(), row.max()):
            Y, X = 2 * int(y), 2 * int(x)
            pts.update([(Y - 1, X), (Y + 1, X), (Y, X - 1), (Y, X + 1)])
    pts = sorted(pts)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    hull = lower[:-1] + upper[:-1]
    n = len(hull)
    H, W = mask.shape
    yy, xx = np.mgrid[0:H, 0:W]
    Y2 = 2 * yy.astype(np.int64)
    X2 = 2 * xx.astype(np.int64)
    inside = np.ones(mask.shape, dtype=bool)
    for i in range(n):
        a, b = hull[i], hull[(i + 1) % n]
        c = (b[0] - a[0]) * (X2 - a[1]) - (b[1] - a[1]) * (Y2 - a[0])
        inside &= c >= 0
    return int(inside.sum())


# reference property names (skimage 0.25.2) handled by the oracle
MORPH_PROPS = (
    "label",
    "centroid",
    "area",
    "area_convex",
    "perimeter",
    "eccentricity",
    "solidity",
    "axis_major_length",
    "axis_minor_length",
    "orientation",
    "bbox",
)
INTENSITY_PROPS = ("intensity_mean", "intensity_max", "intensity_min", "intensity_std")


def regionprops_table(label_image: np.ndarray, intensity_image=None, properties=("label", "bbox")):
    """``ski.measure.regionprops_table`` (SK/measure/_regionprops.py:586-702,1146-1158) for the hot path.

    Regions are enumerated with ``ndi.find_objects`` (ascending label); absent labels are skipped.
    Output columns use 0.25.2 naming (``centroid-0``, ``bbox-0`` ...) and float64 for area /
    area_convex / intensity extrema.
    """
    props = list(properties)
    cols: dict[str, list] = {}

    def put(name, v):
        cols.setdefault(name, []).append(v)

    objects = ndi.find_objects(label_image)
    for i, sl in enumerate(objects):
        if sl is None:
            continue
        lab = i + 1
        img = label_image[sl] == lab
        need_moments = any(
            p in props for p in ("eccentricity", "axis_major_length", "axis_minor_length", "orientation")
        )
        if need_moments:
            u8 = img.astype(np.uint8)
            M = moments_central(u8, (0, 0), order=1)  # raw moments M00, M10, M01
            local_centroid = (M[1, 0] / M[0, 0], M[0, 1] / M[0, 0])
            mu = moments_central(u8, local_centroid, order=3)
            T = inertia_tensor(mu)
            l1, l2 = inertia_tensor_eigvals(T)
        area = float(np.sum(img))
        convex = None  # regionprops caches area_convex per region (SK/measure/_regionprops.py `_cached`)
        for p in props:
            if p == "label":
                put("label", lab)
            elif p == "area":
                put("area", area)
            elif p == "centroid":
                idx = np.nonzero(img)
                put("centroid-0", (idx[0] + sl[0].start).mean())
                put("centroid-1", (idx[1] + sl[1].start).mean())
            elif p == "bbox":
                put("bbox-0", sl[0].start)
                put("bbox-1", sl[1].start)
                put("bbox-2", sl[0].stop)
                put("bbox-3", sl[1].stop)
            elif p == "area_convex":
                convex = float(convex_area_exact(img)) if convex is None else convex
                put("area_convex", convex)
            elif p == "solidity":
                convex = float(convex_area_exact(img)) if convex is None else convex
                put("solidity", area / convex)
            elif p == "perimeter":
                put("perimeter", perimeter(img))
            elif p == "eccentricity":
                put("eccentricity", 0.0 if l1 == 0 else sqrt(1 - l2 / l1))
            elif p == "axis_major_length":
                put("axis_major_length", 4 * sqrt(l1))
            elif p == "axis_minor_length":
                put("axis_minor_length", 4 * sqrt(l2))
            elif p == "orientation":
                a, b, b, c = T.flat
                if a - c == 0:
                    put("orientation", -pi / 4.0 if b < 0 else pi / 4.0)
                else:
                    put("orientation", 0.5 * atan2(-2 * b, c - a))
            elif p in INTENSITY_PROPS:
                if intensity_image is None:
                    raise AttributeError("No intensity image specified.")
                vals = intensity_image[sl][img]
                if p == "intensity_mean":
                    put(p, np.mean(vals))
                elif p == "intensity_max":
                    put(p, float(np.max(vals)))
                elif p == "intensity_min":
                    put(p, float(np.min(vals)))
                else:
                    put(p, np.std(vals))
            else:
                raise AttributeError(f"oracle regionprops: unsupported property {p}")
    out = {}
    for k, v in cols.items():
        out[k] = np.asarray(v, dtype=np.int64 if k == "label" or k.startswith("bbox") else np.float64)
    if not out:
        for p in props:
            if p == "centroid":
                out["centroid-0"] = np.zeros(0)
                out["centroid-1"] = np.zeros(0)
            else:
                out[p] = np.zeros(0)
    return out


def cell_properties(label_image, intensity_image_dict=None, property_names=None, intensity_property_names=None):
    """R/masks.py:247-328 restated on top of the oracle ``regionprops_table``.

    ``intensity_image_dict`` maps a channel NAME (str) to its 2-D image.
    """
    from_default = [
        "label", "centroid", "volume", "area", "area_convex", "perimeter", "eccentricity",
        "circularity", "solidity", "axis_major_length", "axis_minor_length", "orientation",
    ]
    property_names = list(from_default if property_names is None else property_names)
    if intensity_property_names is None:
        intensity_property_names = list(INTENSITY_PROPS) if intensity_image_dict else []
    needs_circ = "circularity" in property_names
    needs_vol = "volume" in property_names
    sk_props = [p for p in property_names if p not in ("circularity", "volume")]
    added = set()
    for dep in ["area", "perimeter"] if needs_circ else []:
        if dep not in sk_props:
            sk_props.append(dep)
            added.add(dep)
    for dep in ["axis_major_length", "axis_minor_length"] if needs_vol else []:
        if dep not in sk_props:
            sk_props.append(dep)
            added.add(dep)
    properties = regionprops_table(label_image, properties=sk_props)
    if needs_circ:
        area = properties["area"]
        per = properties["perimeter"]
        with np.errstate(divide="ignore", invalid="ignore"):
            properties["circularity"] = np.where(per > 0, (4.0 * np.pi * area) / (per**2), 0.0)
    if needs_vol:
        a = properties["axis_major_length"] / 2.0
        b = properties["axis_minor_length"] / 2.0
        properties["volume"] = np.where((a > 0) & (b > 0), (4.0 / 3.0) * np.pi * a * b * b, 0.0)
    for p in added:
        properties.pop(p, None)
    if "centroid-0" in properties:
        properties["centroid_y"] = properties.pop("centroid-0")
    if "centroid-1" in properties:
        properties["centroid_x"] = properties.pop("centroid-1")
    if intensity_image_dict and intensity_property_names:
        for name, inten in intensity_image_dict.items():
            ch = regionprops_table(label_image, intensity_image=inten, properties=intensity_property_names)
            for k, v in ch.items():
                properties[f"{k}_{name.lower()}"] = v
    return properties


_ = itertools  # (kept for parity with the upstream module's imports)
