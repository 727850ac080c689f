"""numpy/scipy restatement of the scikit-image calls on the hot path.  TEST INFRASTRUCTURE ONLY.

Abbreviations in citations: R/ = /root/reference/src/arcadia_microscopy_tools/,
SK/ = scikit-image 0.18.3 source (site-packages/skimage), SP/ = scipy/ndimage.
Each function bottoms out in the same scipy.ndimage / numpy C kernels scikit-image wraps,
so it is the reference's CPU path minus the thin scikit-image python wrappers
(scikit-image itself is not installable on the GPU box).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

# --------------------------------------------------------------------------------------
# dtype conversion -- SK/util/dtype.py:319 (img_as_float on uint16 multiplies by 1/65535)
# --------------------------------------------------------------------------------------


def img_as_float(image: np.ndarray) -> np.ndarray:
    """SK/util/dtype.py:319 -- unsigned int -> float64 is ``x * (1.0 / imax)`` (a multiply)."""
    if image.dtype == np.uint16:
        return image.astype(np.float64) * (1.0 / 65535)
    if image.dtype == np.uint8:
        return image.astype(np.float64) * (1.0 / 255)
    if image.dtype == bool:
        return image.astype(np.float64)
    if image.dtype in (np.float32, np.float64):
        return image.astype(np.float64)
    if image.dtype.kind == "u":  # SK/util/dtype.py:_convert, unsigned -> float: multiply by 1 / imax
        return np.multiply(image, 1.0 / int(np.iinfo(image.dtype).max), dtype=np.float64)
    if image.dtype.kind == "i":  # signed -> float: (x + 0.5) * 2 / (imax - imin)
        info = np.iinfo(image.dtype)
        out = np.add(image, 0.5, dtype=np.float64)
        out *= 2 / (int(info.max) - int(info.min))
        return out
    raise TypeError(f"oracle img_as_float: unsupported dtype {image.dtype}")


# --------------------------------------------------------------------------------------
# Gaussian / DoG -- SK/filters/_gaussian.py:12-126,160-290 ; SP/_filters.py:226-236,314-430
# --------------------------------------------------------------------------------------


def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """SP/_filters.py:226-236,314-323 -- the 1-D kernel scipy builds (order 0)."""
    radius = int(truncate * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x**2)
    phi_x = phi_x / phi_x.sum()
    return phi_x


def gaussian(image: np.ndarray, sigma: float, mode: str = "nearest", truncate: float = 4.0):
    """``skimage.filters.gaussian`` on a 2-D image (SK/filters/_gaussian.py:119-126)."""
    return ndi.gaussian_filter(img_as_float(image), sigma, mode=mode, truncate=truncate)


def difference_of_gaussians(image: np.ndarray, low_sigma: float, high_sigma: float):
    """SK/filters/_gaussian.py:258-290 (called by R/operations.py:91)."""
    f = img_as_float(image)
    im1 = ndi.gaussian_filter(f, low_sigma, mode="nearest", truncate=4.0)
    im2 = ndi.gaussian_filter(f, high_sigma, mode="nearest", truncate=4.0)
    return im1 - im2


# --------------------------------------------------------------------------------------
# histogram + global thresholds -- SK/exposure/exposure.py:38-144, SK/filters/thresholding.py
# --------------------------------------------------------------------------------------


def histogram(image: np.ndarray, nbins: int = 256):
    """SK/exposure/exposure.py:38-74,122-144 with source_range='image'."""
    image = image.ravel()
    if np.issubdtype(image.dtype, np.integer):
        image_min = int(image.min())
        image_max = int(image.max())
        # one bin per integer from image_min to image_max.  scikit-image counts non-negative images from 0 and drops
        # the bins below image_min, and shifts images with negative values by image_min first (_offset_array):
        # either way the counts are those of image - image_min
        hist = np.bincount((image.astype(np.int64) - image_min).ravel(), minlength=image_max - image_min + 1)
        bin_centers = np.arange(image_min, image_max + 1)
        return hist, bin_centers
    hist, bin_edges = np.histogram(image, bins=nbins, range=None)
    bin_centers = (bin_edges[:-1] + bin_edges[1:]) / 2.0
    return hist, bin_centers


def _counts_centers(image, nbins=256):
    counts, centers = histogram(image, nbins)
    return counts.astype(float), centers


def threshold_otsu(image: np.ndarray, nbins: int = 256):
    """SK/filters/thresholding.py:321-350."""
    first_pixel = image.ravel()[0]
    if np.all(image == first_pixel):
        return first_pixel
    counts, bin_centers = _counts_centers(image, nbins)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * bin_centers) / weight1
    mean2 = (np.cumsum((counts * bin_centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = np.argmax(variance12)
    return bin_centers[idx]


def threshold_yen(image: np.ndarray, nbins: int = 256):
    """SK/filters/thresholding.py:394-411."""
    counts, bin_centers = _counts_centers(image, nbins)
    if bin_centers.size == 1:
        return bin_centers[0]
    pmf = counts.astype(np.float32) / counts.sum()
    P1 = np.cumsum(pmf)
    P1_sq = np.cumsum(pmf**2)
    P2_sq = np.cumsum(pmf[::-1] ** 2)[::-1]
    crit = np.log(((P1_sq[:-1] * P2_sq[1:]) ** -1) * (P1[:-1] * (1.0 - P1[:-1])) ** 2)
    return bin_centers[crit.argmax()]


def threshold_isodata(image: np.ndarray, nbins: int = 256):
    """SK/filters/thresholding.py:474-528 (return_all=False)."""
    counts, bin_centers = _counts_centers(image, nbins)
    if len(bin_centers) == 1:
        return bin_centers[0]
    counts = counts.astype(np.float32)
    csuml = np.cumsum(counts)
    csumh = csuml[-1] - csuml
    intensity_sum = counts * bin_centers
    csum_intensity = np.cumsum(intensity_sum)
    lower = csum_intensity[:-1] / csuml[:-1]
    higher = (csum_intensity[-1] - csum_intensity[:-1]) / csumh[:-1]
    all_mean = (lower + higher) / 2.0
    bin_width = bin_centers[1] - bin_centers[0]
    distances = all_mean - bin_centers[:-1]
    thresholds = bin_centers[:-1][(distances >= 0) & (distances < bin_width)]
    return thresholds[0]


def threshold_mean(image: np.ndarray):
    """SK/filters/thresholding.py:830."""
    return np.mean(image)


def threshold_triangle(image: np.ndarray, nbins: int = 256):
    """SK/filters/thresholding.py:866-907."""
    hist, bin_centers = histogram(image.ravel(), nbins)
    nbins = len(hist)
    arg_peak_height = np.argmax(hist)
    peak_height = hist[arg_peak_height]
    arg_low_level, arg_high_level = np.where(hist > 0)[0][[0, -1]]
    flip = arg_peak_height - arg_low_level < arg_high_level - arg_peak_height
    if flip:
        hist = hist[::-1]
        arg_low_level = nbins - arg_high_level - 1
        arg_peak_height = nbins - arg_peak_height - 1
    del arg_high_level
    width = arg_peak_height - arg_low_level
    x1 = np.arange(width)
    y1 = hist[x1 + arg_low_level]
    norm = np.sqrt(peak_height**2 + width**2)
    peak_height = peak_height / norm
    width = width / norm
    length = peak_height * x1 - width * y1
    arg_level = np.argmax(length) + arg_low_level
    if flip:
        arg_level = nbins - arg_level - 1
    return bin_centers[arg_level]


def threshold_li(image: np.ndarray, tolerance=None, initial_guess=None):
    """SK/filters/thresholding.py:642-707 (default initial guess, no callback)."""
    image = image[~np.isnan(image)]
    if image.size == 0:
        return np.nan
    if np.all(image == image.flat[0]):
        return image.flat[0]
    if np.any(np.isinf(image)):
        return np.inf
    image_min = np.min(image)
    image = image - image_min
    tolerance = tolerance or np.min(np.diff(np.unique(image))) / 2
    if initial_guess is None:
        t_next = np.mean(image)
    else:
        t_next = initial_guess - image_min
    t_curr = -2 * tolerance
    while abs(t_next - t_curr) > tolerance:
        t_curr = t_next
        foreground = image > t_curr
        mean_fore = np.mean(image[foreground])
        mean_back = np.mean(image[~foreground])
        t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore))
    return t_next + image_min


def threshold_minimum(image: np.ndarray, nbins: int = 256, max_iter: int = 10000):
    """SK/filters/thresholding.py:763-799."""

    def find_local_maxima_idx(hist):
        maximum_idxs = list()
        direction = 1
        for i in range(hist.shape[0] - 1):
            if direction > 0:
                if hist[i + 1] < hist[i]:
                    direction = -1
                    maximum_idxs.append(i)
            else:
                if hist[i + 1] > hist[i]:
                    direction = 1
        return maximum_idxs

    counts, bin_centers = _counts_centers(image, nbins)
    smooth_hist = counts.astype(np.float64, copy=False)
    for counter in range(max_iter):
        smooth_hist = ndi.uniform_filter1d(smooth_hist, 3)
        maximum_idxs = find_local_maxima_idx(smooth_hist)
        if len(maximum_idxs) < 3:
            break
    if len(maximum_idxs) != 2:
        raise RuntimeError("Unable to find two maxima in histogram")
    elif counter == max_iter - 1:
        raise RuntimeError("Maximum iteration reached for histogram smoothing")
    threshold_idx = np.argmin(smooth_hist[maximum_idxs[0]:maximum_idxs[1] + 1])
    return bin_centers[maximum_idxs[0] + threshold_idx]


def threshold_local(image, block_size, method="gaussian", offset=0, mode="reflect", param=None, cval=0):
    """SK/filters/thresholding.py:206-236."""
    if block_size % 2 == 0:
        raise ValueError(
            "The kwarg ``block_size`` must be odd! Given ``block_size`` {0} is even.".format(block_size)
        )
    thresh_image = np.zeros(image.shape, "double")
    if method == "gaussian":
        sigma = (block_size - 1) / 6.0 if param is None else param
        ndi.gaussian_filter(image, sigma, output=thresh_image, mode=mode, cval=cval)
    elif method == "mean":
        mask = 1.0 / block_size * np.ones((block_size,))
        ndi.convolve1d(image, mask, axis=0, output=thresh_image, mode=mode, cval=cval)
        ndi.convolve1d(thresh_image, mask, axis=1, output=thresh_image, mode=mode, cval=cval)
    elif method == "median":
        ndi.median_filter(image, block_size, output=thresh_image, mode=mode, cval=cval)
    else:
        raise ValueError("oracle threshold_local: unsupported method " + method)
    return thresh_image - offset


def _mean_std(image, w):
    """SK/filters/thresholding.py:910-964 (integral-image window mean / std)."""
    import itertools

    if not isinstance(w, (list, tuple, np.ndarray)):
        w = (w,) * image.ndim
    pad_width = tuple((k // 2 + 1, k // 2) for k in w)
    padded = np.pad(image.astype("float"), pad_width, mode="reflect")
    padded_sq = padded * padded
    integral = padded.copy()
    integral_sq = padded_sq.copy()
    for i in range(image.ndim):
        integral = np.cumsum(integral, axis=i)
        integral_sq = np.cumsum(integral_sq, axis=i)
    kern = np.zeros(tuple(k + 1 for k in w))
    for indices in itertools.product(*([[0, -1]] * image.ndim)):
        kern[indices] = (-1) ** (image.ndim % 2 != np.sum(indices) % 2)
    total_window_size = np.prod(w)
    sum_full = ndi.correlate(integral, kern, mode="constant")
    m = _crop(sum_full, pad_width) / total_window_size
    sum_sq_full = ndi.correlate(integral_sq, kern, mode="constant")
    g2 = _crop(sum_sq_full, pad_width) / total_window_size
    s = np.sqrt(np.clip(g2 - m * m, 0, None))
    return m, s


def _crop(ar, crop_width):
    slices = tuple(slice(a, ar.shape[i] - b) for i, (a, b) in enumerate(crop_width))
    return ar[slices]


def threshold_niblack(image, window_size=15, k=0.2):
    """SK/filters/thresholding.py:1026-1027."""
    m, s = _mean_std(image, window_size)
    return m - k * s


def threshold_sauvola(image, window_size=15, k=0.2, r=None):
    """SK/filters/thresholding.py:1083-1087."""
    if r is None:
        # dtype_limits(image, clip_negative=False): integer range, (False, True) for bool, (-1, 1) for floats
        if image.dtype == bool:
            imin, imax = 0, 1
        elif image.dtype.kind in "ui":
            imin, imax = int(np.iinfo(image.dtype).min), int(np.iinfo(image.dtype).max)
        elif image.dtype.kind == "f":
            imin, imax = -1, 1
        else:
            raise TypeError("oracle sauvola: dtype")
        r = 0.5 * (imax - imin)
    m, s = _mean_std(image, window_size)
    return m * (1 + k * ((s / r) - 1))


# --------------------------------------------------------------------------------------
# exposure -- SK/exposure/exposure.py:313-428 ; numpy percentile
# --------------------------------------------------------------------------------------


def rescale_intensity(image, in_range, out_range):
    """SK/exposure/exposure.py:405-428 with tuple in_range and tuple out_range (float output)."""
    imin, imax = map(float, in_range)
    omin, omax = map(float, out_range)
    image = np.clip(image, imin, imax)
    if imin != imax:
        image = (image - imin) / (imax - imin)
        return np.asarray(image * (omax - omin) + omin, dtype=np.float64)
    return np.clip(image, omin, omax).astype(np.float64)


# --------------------------------------------------------------------------------------
# morphology -- SK/morphology/binary.py, grey.py, selem.py ; SK/filters/_median.py
# --------------------------------------------------------------------------------------


def disk(radius: int, dtype=np.uint8) -> np.ndarray:
    """SK/morphology/selem.py ``disk``: X**2 + Y**2 <= radius**2."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X**2 + Y**2) <= radius**2, dtype=dtype)


def cross3() -> np.ndarray:
    """default structuring element: ndi.generate_binary_structure(2, 1) (SK/morphology/selem.py)."""
    return ndi.generate_binary_structure(2, 1).astype(np.uint8)


def binary_erosion(image, selem=None):
    """SK/morphology/binary.py:42 (border_value=True)."""
    selem = cross3() if selem is None else selem
    return ndi.binary_erosion(image, structure=selem, border_value=True)


def binary_dilation(image, selem=None):
    """SK/morphology/binary.py:77 (border value False)."""
    selem = cross3() if selem is None else selem
    return ndi.binary_dilation(image, structure=selem)


def binary_opening(image, selem=None):
    """SK/morphology/binary.py:82-113 -- dilation(erosion(image))."""
    return binary_dilation(binary_erosion(image, selem), selem)


def binary_closing(image, selem=None):
    """SK/morphology/binary.py:116-147 -- erosion(dilation(image))."""
    return binary_erosion(binary_dilation(image, selem), selem)


def _mirror_selem(selem):
    """SK/morphology/grey.py:52-81 ``_invert_selem``: flip along every axis."""
    return selem[::-1, ::-1]


def _shift_selem(selem, shift: bool):
    """SK/morphology/grey.py:14-50: an even side gets a zero row / column, in front of the footprint (shift False, the
    default of erosion / dilation) or behind it (shift True: the second half of opening / closing)."""
    selem = np.asarray(selem)
    if selem.ndim != 2:
        return selem
    m, n = selem.shape
    if m % 2 == 0:
        extra = np.zeros((1, n), selem.dtype)
        selem = np.vstack((selem, extra)) if shift else np.vstack((extra, selem))
        m += 1
    if n % 2 == 0:
        extra = np.zeros((m, 1), selem.dtype)
        selem = np.hstack((selem, extra)) if shift else np.hstack((extra, selem))
    return selem


def erosion(image, selem=None, shift: bool = False):
    """SK/morphology/grey.py:131-187 (scipy default mode='reflect')."""
    selem = cross3() if selem is None else selem
    return ndi.grey_erosion(image, footprint=_shift_selem(selem, shift))


def dilation(image, selem=None, shift: bool = False):
    """SK/morphology/grey.py:190-254 (footprint shifted, then mirrored before scipy)."""
    selem = cross3() if selem is None else selem
    return ndi.grey_dilation(image, footprint=_mirror_selem(_shift_selem(selem, shift)))


def _eccentric(image, selem, first, second):
    """SK/morphology/grey.py:84-127 ``pad_for_eccentric_selems``: even footprint sides -> edge-pad the image by
    (side - 1) along that axis, run the pair, crop back."""
    selem = cross3() if selem is None else np.asarray(selem)
    pads = [(n - 1, n - 1) if n % 2 == 0 else (0, 0) for n in selem.shape]
    if not any(p[0] for p in pads):
        return second(first(image, selem), selem, shift=True)
    padded = np.pad(image, pads, mode="edge")
    res = second(first(padded, selem), selem, shift=True)
    return res[pads[0][0]: res.shape[0] - pads[0][0], pads[1][0]: res.shape[1] - pads[1][0]]


def opening(image, selem=None):
    """SK/morphology/grey.py:255-303: the dilation runs with shift_x = shift_y = True."""
    return _eccentric(image, selem, erosion, dilation)


def closing(image, selem=None):
    """SK/morphology/grey.py:305-353: the erosion runs with shift_x = shift_y = True."""
    return _eccentric(image, selem, dilation, erosion)


def white_tophat(image, selem=None):
    """SK/morphology/grey.py:356-427 -> ndi.white_tophat (image - opening, scipy's own opening)."""
    selem = cross3() if selem is None else selem
    if image.dtype == bool:
        raise NotImplementedError
    return ndi.white_tophat(image, footprint=selem)


def median(image, selem=None, mode="nearest"):
    """SK/filters/_median.py -> ndi.median_filter(footprint, mode='nearest')."""
    selem = np.ones((3, 3), dtype=np.uint8) if selem is None else selem
    return ndi.median_filter(image, footprint=selem, mode=mode)


# --------------------------------------------------------------------------------------
# labelling -- SK/measure/_label.py, SK/segmentation/_clear_border.py, _join.py
# --------------------------------------------------------------------------------------

_FULL8 = np.ones((3, 3), dtype=bool)


def label(image, connectivity: int = 2):
    """``skimage.measure.label`` (default connectivity = ndim = 8-connected in 2-D).

    bool input -> ndimage.label(structure=ones((3,3))) (SK/measure/_label.py); integer input ->
    components of EQUAL-valued non-zero pixels, numbered in raster order of first pixel
    (verified against the Cython path).
    """
    structure = _FULL8 if connectivity == 2 else ndi.generate_binary_structure(2, 1)
    if image.dtype == bool:
        return ndi.label(image, structure=structure)[0].astype(np.int64)
    # integer path: one raster pass of equal-value unions (oracle/clabel.c), the cost of scikit-image's
    # own Cython labelling -- clear_border on a ~1,350-label watershed image goes through here
    from .clabel import label_int

    return label_int(image, connectivity)


def _label_int_per_value(image, connectivity: int = 2):
    """The integer path restated with scipy only (one ndimage.label per value, then renumbering by
    first raster pixel): O(values x pixels), kept as the cross-check of oracle/clabel.c in the tests."""
    structure = _FULL8 if connectivity == 2 else ndi.generate_binary_structure(2, 1)
    out = np.zeros(image.shape, dtype=np.int64)
    nxt = 0
    firsts = []
    for v in np.unique(image):
        if v == 0:
            continue
        lab, n = ndi.label(image == v, structure=structure)
        sel = lab > 0
        out[sel] = lab[sel] + nxt
        nxt += n
    if nxt == 0:
        return out
    flat = out.ravel()
    first_idx = np.full(nxt + 1, flat.size, dtype=np.int64)
    np.minimum.at(first_idx, flat, np.arange(flat.size))
    order = np.argsort(first_idx[1:], kind="stable")
    remap = np.zeros(nxt + 1, dtype=np.int64)
    remap[order + 1] = np.arange(1, nxt + 1)
    return remap[out]


def clear_border(labels):
    """SK/segmentation/_clear_border.py (buffer_size=0, bgval=0, no mask); R/masks.py:56."""
    image = labels
    borders = np.zeros_like(image, dtype=bool)
    borders[0, :] = borders[-1, :] = True
    borders[:, 0] = borders[:, -1] = True
    lab = label(image)
    number = np.max(lab) + 1
    borders_indices = np.unique(lab[borders])
    indices = np.arange(number + 1)
    label_mask = np.isin(indices, borders_indices)
    mask = label_mask[lab.ravel()].reshape(lab.shape)
    image = image.copy()
    image[mask] = 0
    return image


def relabel_sequential(label_field):
    """SK/segmentation/_join.py:46 (offset=1); R/masks.py:65.  Returns the relabelled array only."""
    in_vals = np.unique(label_field)
    if in_vals[0] == 0:
        out_vals = np.concatenate([[0], np.arange(1, len(in_vals))])
    else:
        out_vals = np.arange(1, len(in_vals) + 1)
    fw = np.zeros(int(in_vals[-1]) + 1, dtype=np.int64)
    fw[in_vals] = out_vals
    return fw[label_field]


# --------------------------------------------------------------------------------------
# distance transform, markers -- scipy.ndimage ; SURVEY.md A.4, A.8
# --------------------------------------------------------------------------------------


def distance_transform_edt(mask):
    return ndi.distance_transform_edt(mask)


def peak_markers(edt, mask, min_distance: int = 5):
    """Marker recipe of SURVEY.md A.8 (the reference defines none).

    peaks = (edt == maximum_filter(edt, (2m+1)^2, mode='constant')) & mask & (edt > 0), a border
    of width m cleared; markers = ndimage.label(peaks) with scipy's default 4-connected structure.
    """
    m = int(min_distance)
    size = 2 * m + 1
    peaks = (edt == ndi.maximum_filter(edt, size=size, mode="constant")) & mask & (edt > 0)
    if m > 0:
        peaks[:m, :] = False
        peaks[-m:, :] = False
        peaks[:, :m] = False
        peaks[:, -m:] = False
    markers, n = ndi.label(peaks)
    return markers.astype(np.int32), n


def seeded_flood_image(edt, markers):
    """Watershed relief for the config-3 recipe: -edt, with marker pixels lowered to DISTINCT values.

    scikit-image's flood pops equal-(value, age) heap entries in an order that depends on the
    internals of its binary heap.  All marker pixels enter with age 0, so equal-valued
    markers (ubiquitous on an EDT) make the label image depend on that artefact.  The recipe therefore
    gives the k-th marker pixel in raster order the value ``floor(min(-edt)) - 1 - (M - k)``: every
    marker pops before any other pixel, in raster order, and the result is a function of
    (value, insertion age) only.  This is an ordinary ``watershed(image, markers, mask=mask)`` call.
    """
    relief = -edt
    idx = np.flatnonzero(markers.ravel())
    M = idx.size
    base = np.floor(relief.min()) - 1.0
    r = relief.copy().ravel()
    r[idx] = base - (M - np.arange(M, dtype=np.float64))
    return r.reshape(edt.shape)
