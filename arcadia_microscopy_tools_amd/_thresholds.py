"""Global threshold selection from a DEVICE-computed histogram.

The pixel-sized work (histogram, min/max, masked sums) runs on the GPU; what is left is arithmetic on
<= 65,536 (integer images) or 256 (float images) counts, evaluated here with the same numpy
expressions scikit-image uses, so the thresholds are bit-identical by construction
(SURVEY.md A.11).  Citations: SK/ = scikit-image 0.18.3 ``filters/thresholding.py``.

Attribution: ``yen``, ``isodata``, ``triangle`` and ``minimum`` below follow scikit-image's expressions closely,
variable names included -- bit-identical thresholds require the same floating-point expressions in the same order.
scikit-image is Copyright (C) 2019, the scikit-image team, and distributed under the BSD 3-Clause License
(https://github.com/scikit-image/scikit-image/blob/main/LICENSE.txt): redistribution in source form must retain
that copyright notice, the list of conditions and the disclaimer ("THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS
AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES ... ARE DISCLAIMED"); neither the name of skimage nor
the names of its contributors may be used to endorse or promote products derived from this software without specific
prior written permission.  See THIRD_PARTY_NOTICES.md at the repository root for the full text.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


def counts_centers_u16(hist65536: np.ndarray):
    """SK/exposure/exposure.py:63-74: one bin per integer from image_min to image_max."""
    nz = np.flatnonzero(hist65536)
    lo, hi = int(nz[0]), int(nz[-1])
    return hist65536[lo:hi + 1].astype(np.int64), np.arange(lo, hi + 1)


def counts_centers_f64(hist: np.ndarray, vmin: float, vmax: float):
    """SK/exposure/exposure.py:139-140: np.histogram edges (np.linspace) -> centres."""
    edges = np.linspace(vmin, vmax, len(hist) + 1)
    return hist.astype(np.int64), (edges[:-1] + edges[1:]) / 2.0


def otsu(counts, bin_centers):
    """SK thresholding.py:321-350 on a histogram: the bin centre that maximises the between-class variance.  (uint16
    images take ``amt_i_otsu_from_hist`` on the device; this is for integer images that travel shifted.)"""
    counts = counts.astype(float)
    if len(bin_centers) == 1:
        return bin_centers[0]
    w_lo = np.cumsum(counts)
    w_hi = np.cumsum(counts[::-1])[::-1]
    weighted = counts * bin_centers
    m_lo = np.cumsum(weighted) / w_lo
    m_hi = (np.cumsum(weighted[::-1]) / w_hi[::-1])[::-1]
    between = w_lo[:-1] * w_hi[1:] * (m_lo[:-1] - m_hi[1:]) ** 2
    return bin_centers[np.argmax(between)]


def yen(counts, bin_centers):
    """SK thresholding.py:394-411."""
    counts = counts.astype(float)
    if bin_centers.size == 1:
        return bin_centers[0]
    pmf = counts.astype(np.float32) / counts.sum()
    P1 = np.cumsum(pmf)
    P1_sq = np.cumsum(pmf**2)
    P2_sq = np.cumsum(pmf[::-1] ** 2)[::-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        crit = np.log(((P1_sq[:-1] * P2_sq[1:]) ** -1) * (P1[:-1] * (1.0 - P1[:-1])) ** 2)
    return bin_centers[crit.argmax()]


def isodata(counts, bin_centers):
    """SK thresholding.py:474-528 (return_all=False)."""
    counts = counts.astype(float)
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


def triangle(hist, bin_centers):
    """SK thresholding.py:866-907."""
    nbins = len(hist)
    arg_peak_height = np.argmax(hist)
    peak_height = hist[arg_peak_height]
    arg_low_level, arg_high_level = np.where(hist > 0)[0][[0, -1]]
    flip = arg_peak_height - arg_low_level < arg_high_level - arg_peak_height
    if flip:
        hist = hist[::-1]
        arg_low_level = nbins - arg_high_level - 1
        arg_peak_height = nbins - arg_peak_height - 1
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


def minimum(counts, bin_centers, max_iter: int = 10000):
    """SK thresholding.py:763-799; raises RuntimeError like scikit-image when no two maxima exist."""

    def local_maxima(h):
        # scikit-image walks the histogram with a direction flag (+1 rising, -1 falling; flat steps keep it) and
        # records i whenever a rise turns into a fall.  After every non-flat step the flag equals the step's sign,
        # so the recorded positions are the falling steps whose previous non-flat step rose (or that come first):
        # one vector pass instead of a Python loop over 65,536 bins, up to 10,000 times.
        step = np.diff(h)
        nz = np.flatnonzero(step)
        if nz.size == 0:
            return []
        falling = step[nz] < 0
        prev_rising = np.ones(nz.size, bool)
        prev_rising[1:] = ~falling[:-1]
        return nz[falling & prev_rising].tolist()

    smooth = counts.astype(np.float64, copy=False)
    maxima: list[int] = []
    counter = 0
    for counter in range(max_iter):
        smooth = ndi.uniform_filter1d(smooth, 3)
        maxima = local_maxima(smooth)
        if len(maxima) < 3:
            break
    if len(maxima) != 2:
        raise RuntimeError("Unable to find two maxima in histogram")
    elif counter == max_iter - 1:
        raise RuntimeError("Maximum iteration reached for histogram smoothing")
    k = np.argmin(smooth[maxima[0]:maxima[1] + 1])
    return bin_centers[maxima[0] + k]


def mean_from_hist(counts, values):
    """np.mean of an integer image from its exact histogram (sums of integers are exact in float64)."""
    return float(np.sum(counts * values)) / float(np.sum(counts))


def li_from_hist(counts, values, tolerance=None, initial_guess=None, wrap_dtype=None):
    """SK thresholding.py:642-707 for integer images, evaluated on the exact histogram: every mean the
    iteration needs is an exact integer sum divided by a count, as in numpy's own float64 reduction.

    ``wrap_dtype``: the caller's integer dtype.  scikit-image subtracts the minimum IN that dtype (``image -=
    image_min``), so a signed image whose range exceeds the dtype's maximum (int8 from -128 to 127, ...) wraps around
    there and the iteration runs on the wrapped values -- reproduced here value for value (the result is then as
    meaningless as scikit-image's own, possibly nan, but it is the reference's)."""
    values = values.astype(np.int64)
    counts = np.asarray(counts)
    image_min = int(values[0])
    v = values - image_min
    if wrap_dtype is not None and np.dtype(wrap_dtype).kind == "i" and int(v[-1]) > int(np.iinfo(wrap_dtype).max):
        half = int(np.iinfo(wrap_dtype).max) + 1
        v = (v + half) % (2 * half) - half  # two's-complement wrap of x - min in the image's own dtype
        order = np.argsort(v, kind="stable")
        v, counts = v[order], counts[order]
    present = v[counts > 0]
    if present.size == 1:
        return values[0]
    tolerance = tolerance or np.min(np.diff(present)) / 2
    total_sum = np.sum(counts * v)
    total_cnt = np.sum(counts)
    t_next = total_sum / total_cnt if initial_guess is None else initial_guess - image_min
    t_curr = -2 * tolerance
    csum = np.cumsum(counts * v)
    ccnt = np.cumsum(counts)
    with np.errstate(invalid="ignore", divide="ignore"):  # wrapped (negative) values: nan, as in scikit-image
        while abs(t_next - t_curr) > tolerance:
            t_curr = t_next
            k = int(np.searchsorted(v, t_curr, side="right")) - 1  # last value <= t_curr
            s_le = csum[k] if k >= 0 else 0
            c_le = ccnt[k] if k >= 0 else 0
            mean_back = np.float64(s_le) / np.float64(c_le)
            mean_fore = np.float64(total_sum - s_le) / np.float64(total_cnt - c_le)
            t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore))
    return t_next + image_min
