"""Image operations of the reference (R/operations.py:10-216) on the GPU.

``rescale_by_percentile``, ``subtract_background_dog``, ``crop_to_center`` and ``apply_threshold`` keep the
reference's signatures, validation, error messages and edge-case results.  Each accepts a numpy array
(uploaded, processed, downloaded: a drop-in call) or a ``DeviceArray`` (stays in HBM, so a ``Pipeline``
of these functions makes one upload and one download).  All pixel-sized arithmetic runs in HIP kernels
behind the C ABI; there is no CPU fallback.

n-D inputs follow the reference, which hands whatever array it is given to numpy / scikit-image: percentiles and
global thresholds are taken over the WHOLE stack, a difference of Gaussians filters EVERY axis (leading axes first, as
scipy does), ``crop_to_center`` crops the last two axes.  To treat a (T, Y, X) / (Z, Y, X) stack plane by plane use
``Pipeline(parallel=True)``, which maps the operations over axis 0 as the reference does (R/pipeline.py:139-149).
The local thresholds follow suit: Niblack / Sauvola windows span every axis of a stack (``window_size`` an odd integer
or one per axis), ``threshold_local`` filters every axis with its Gaussian; its ``mean`` / ``median`` methods are 2-D only.

dtypes: uint8 / uint16 / float64 are computed as the reference computes them (bit-exact).  Other integer
types are converted on upload: to uint16 when the values fit (exact), otherwise to float64 (exact below 2^53).  The
histogram thresholds bin integers one bin per value as scikit-image does: an integer image beyond the uint16 range
travels as ``x - min`` when its RANGE fits 65,536 values, as float64 with one device bin per value up to a range of
2^26 values (``amt_hist_range_f64``), and is refused beyond that (scikit-image itself would need gigabytes per array).
Where scikit-image's result depends on the dtype itself -- ``img_as_float`` inside the Gaussians of the DoG, Sauvola's
default ``r`` -- the CALLER's dtype decides (``_img_as_float_plan``, ``_sauvola_r``), not the type the data travels as.
float32 / float16 are computed in float64: numpy / scikit-image keep float32 arithmetic for them, so results agree
to float32 rounding (~1e-7 relative), inside the 1e-5 bar for float outputs but not bit for bit.
"""
from __future__ import annotations

from typing import Literal

import numpy as np

from . import _thresholds, hipops
from .device import DeviceArray, get_context
from .pipeline import device_operator
from .typing import BoolArray, Float64Array, ScalarArray

_MAX_INTEGER_BINS = 1 << 26  # one uint32 bin per integer value: 256 MB on the device, 512 MB per float64 host array

_SUPPORTED_METHODS = ("otsu", "li", "yen", "isodata", "mean", "minimum", "triangle", "local", "niblack", "sauvola")


def _to_device(intensities, what: str, integer_histogram: bool = False, shifted: list | None = None):
    """-> (DeviceArray, was_numpy).  See the module docstring for the dtype rules.  ``integer_histogram``: the caller
    bins integer images by value (scikit-image's histogram), so integers beyond uint16 cannot be taken -- unless the
    caller passes ``shifted`` (a one-element list) and the image's RANGE fits 65,536 values: it then travels as
    ``x - min`` in uint16 and ``shifted[0]`` receives ``min`` (scikit-image bins such images from image_min to
    image_max, one bin per integer: SK/exposure/exposure.py:38-74)."""
    if isinstance(intensities, DeviceArray):
        d = intensities
        if d.dtype not in (np.uint16, np.float64):
            raise TypeError(f"{what}: device arrays must be uint16 or float64, got {d.dtype}")
        return d, False
    a = np.asarray(intensities)
    if a.dtype == np.bool_:
        a = a.astype(np.uint16)
    elif a.dtype == np.uint8:
        a = a.astype(np.uint16)
    elif np.issubdtype(a.dtype, np.integer) and a.dtype != np.uint16:
        lo, hi = (int(a.min()), int(a.max())) if a.size else (0, 0)
        if lo >= 0 and hi <= 65535:
            a = a.astype(np.uint16)  # exact: same values, same integer histogram
        elif integer_histogram and shifted is not None and hi - lo <= 65535:
            shifted[0] = lo
            # subtract in a type that holds the whole range: int8 [-128, 127] - (-128) overflows int8 itself
            wide = a if a.dtype == np.uint64 else a.astype(np.int64)
            a = (wide - wide.dtype.type(lo)).astype(np.uint16)
        elif integer_histogram and shifted is not None and hi - lo < _MAX_INTEGER_BINS:
            # one bin per value over a range beyond uint16: the image travels as float64 (exact below 2^53) and is
            # binned by amt_hist_range_f64; shifted[1] tells the caller the range
            shifted[0] = lo
            shifted.append(hi - lo + 1)
            a = a.astype(np.float64)
        elif integer_histogram:
            raise NotImplementedError(
                f"{what}: integer image with values in [{lo}, {hi}]: scikit-image bins integers one bin per value "
                f"({hi - lo + 1:,} bins here); the device path stops at {_MAX_INTEGER_BINS:,}"
            )
        else:
            a = a.astype(np.float64)
    elif a.dtype in (np.float32, np.float16):
        a = a.astype(np.float64)
    elif a.dtype not in (np.uint16, np.float64):
        raise TypeError(f"{what}: dtype {a.dtype} is not supported on the MI355X path")
    return get_context().asarray(np.ascontiguousarray(a)), True


def _img_as_float_plan(intensities):
    """How ``skimage.util.img_as_float`` (SK/util/dtype.py:280-330, called by ``filters.gaussian``) treats the dtype of a
    host array -> (array to upload, scale for the device Gaussian or None for its default).  uint16 and uint8 travel as
    uint16 and are multiplied by 1/65535 resp. 1/255 on the device, bool by 1; wider unsigned integers are multiplied by
    1/imax and signed integers mapped by (x + 0.5) * 2 / (imax - imin) on the host, in float64, exactly as scikit-image
    does, and travel as float64."""
    if isinstance(intensities, DeviceArray):
        return intensities, None
    a = np.asarray(intensities)
    if a.dtype == np.bool_:
        return a, 1.0
    if a.dtype == np.uint8:
        return a, 1.0 / 255
    if a.dtype.kind == "u" and a.dtype != np.uint16:
        return np.multiply(a, 1.0 / int(np.iinfo(a.dtype).max), dtype=np.float64), None
    if a.dtype.kind == "i":
        info = np.iinfo(a.dtype)
        f = np.add(a, 0.5, dtype=np.float64)
        f *= 2 / (int(info.max) - int(info.min))
        return f, None
    return a, None


def _sauvola_r(dtype) -> float:
    """threshold_sauvola's default ``r``: half the dtype's range, ``dtype_limits(image, clip_negative=False)``
    (SK/filters/thresholding.py:1079-1081; floats count as (-1, 1), bool as (False, True))."""
    dt = np.dtype(dtype)
    if dt == np.bool_:
        return 0.5
    if dt.kind in "ui":
        info = np.iinfo(dt)
        return 0.5 * (int(info.max) - int(info.min))
    return 1.0


def _flat(d: DeviceArray) -> DeviceArray:
    """The whole array as ONE plane (1, size): statistics over a stack are statistics over all its samples."""
    return d if d.ndim == 2 else d.reshape(1, d.size)


def _result(d: DeviceArray, was_numpy: bool):
    return d.numpy() if was_numpy else d


def _min_max(d: DeviceArray):
    mm = hipops.percentile(_flat(d), (0.0, 100.0)).numpy()[0]
    return mm[0], mm[1]


@device_operator
def rescale_by_percentile(
    intensities: ScalarArray,
    percentile_range: tuple[float, float] = (0, 100),
    out_range: tuple[float, float] = (0, 1),
) -> ScalarArray:
    """Percentile-based contrast stretching (R/operations.py:10-54): ``np.percentile`` then
    ``skimage.exposure.rescale_intensity(in_range=(p1, p2), out_range=out_range)``; float64 result."""
    if not (0 <= percentile_range[0] < percentile_range[1] <= 100):
        raise ValueError(
            f"Invalid percentile range: {percentile_range}. "
            f"Values must be in ascending order between 0 and 100."
        )
    if not isinstance(intensities, DeviceArray) and np.asarray(intensities).size == 0:
        return np.zeros_like(intensities, dtype=float)
    d, was_numpy = _to_device(intensities, "rescale_by_percentile")
    lo, hi = _min_max(d)
    if lo == hi:  # constant image (R/operations.py:43-44)
        const = np.full(d.shape, out_range[0], dtype=float)
        return const if was_numpy else d.ctx.asarray(const)
    f = _flat(d)
    p = hipops.percentile(f, percentile_range)
    return _result(hipops.rescale(f, p, out_range).reshape(d.shape), was_numpy)


@device_operator
def subtract_background_dog(
    intensities: ScalarArray,
    low_sigma: float = 0.6,
    high_sigma: float = 16.0,
    percentile: float = 0,
) -> Float64Array:
    """Difference-of-Gaussians background subtraction (R/operations.py:57-97):
    ``clip(dog - np.percentile(dog, percentile), 0, None)`` with ``dog = difference_of_gaussians(x, low, high)``."""
    if not (0 <= percentile <= 100):
        raise ValueError(f"Percentile must be between 0 and 100, got {percentile}")
    if low_sigma >= high_sigma:
        raise ValueError(f"low_sigma ({low_sigma}) must be smaller than high_sigma ({high_sigma})")
    intensities, scale = _img_as_float_plan(intensities)
    d, was_numpy = _to_device(intensities, "subtract_background_dog")
    dog = (hipops.difference_of_gaussians(d, low_sigma, high_sigma, scale=scale) if d.ndim == 2 else
           hipops.difference_of_gaussians_nd(d, low_sigma, high_sigma, scale=scale))
    f = _flat(dog)
    level = hipops.percentile(f, percentile)
    hipops.sub_clip0(f, level, out=f)
    return _result(dog, was_numpy)


@device_operator
def crop_to_center(intensities: ScalarArray, output_shape: tuple[int, int]) -> ScalarArray:
    """Centre crop on the last two axes, clamped to the image size (R/operations.py:100-132).
    numpy input -> a view, as in the reference; DeviceArray input -> a cropped device copy."""
    height, width = intensities.shape[-2:]
    crop_height, crop_width = output_shape
    crop_width = min(width, crop_width)
    crop_height = min(height, crop_height)
    left = (width - crop_width) // 2
    top = (height - crop_height) // 2
    if isinstance(intensities, DeviceArray):
        return hipops.crop(intensities, top, left, crop_height, crop_width)
    return intensities[..., top: top + crop_height, left: left + crop_width]


def _global_threshold(d: DeviceArray, method: str, kwargs: dict) -> float:
    """Threshold VALUE for the histogram-based methods: histogram on the device, selection on <= 65,536 counts."""
    nbins = int(kwargs.pop("nbins", 256))
    if d.dtype == np.uint16:
        hist = hipops.histogram_u16(d).numpy()[0]
        counts, centers = _thresholds.counts_centers_u16(hist)
    else:
        h, mm = hipops.histogram_f64(d, nbins)
        mmv = mm.numpy()[0]
        counts, centers = _thresholds.counts_centers_f64(h.numpy()[0], mmv[0], mmv[1])
    if method == "yen":
        return _thresholds.yen(counts, centers)
    if method == "isodata":
        return _thresholds.isodata(counts, centers)
    if method == "triangle":
        return _thresholds.triangle(counts, centers)
    if method == "minimum":
        return _thresholds.minimum(counts, centers, **kwargs)
    raise AssertionError(method)


def _mean_threshold(d: DeviceArray) -> float:
    if d.dtype == np.uint16:
        hist = hipops.histogram_u16(d).numpy()[0]
        counts, centers = _thresholds.counts_centers_u16(hist)
        return _thresholds.mean_from_hist(counts, centers)
    inf = d.ctx.asarray(np.array([np.inf]))
    s = hipops.masked_sums(d, inf).numpy()[0]
    return s[0] / s[1]


def _li_threshold(d: DeviceArray, tolerance=None, initial_guess=None) -> float:
    """``skimage.filters.threshold_li`` (SK thresholding.py:642-707).  Integer images: exact, from the
    histogram.  Float images: every iteration's two class means come from one device reduction."""
    if d.dtype == np.uint16:
        hist = hipops.histogram_u16(d).numpy()[0]
        counts, centers = _thresholds.counts_centers_u16(hist)
        return _thresholds.li_from_hist(counts, centers, tolerance, initial_guess)
    ctx = d.ctx
    image_min, image_max = _min_max(d)
    if tolerance is None:
        # smallest gap between distinct values / 2: for float images take it from the data's own resolution
        # (scikit-image sorts the unique values; any tolerance below the true gap yields the same fixed point)
        tolerance = np.spacing(max(abs(image_min), abs(image_max))) / 2
    tot = hipops.masked_sums(d, ctx.asarray(np.array([np.inf]))).numpy()[0]
    t_next = (tot[0] / tot[1]) if initial_guess is None else float(initial_guess)
    t_curr = t_next - 2 * tolerance - 1.0
    # work in the original value space: means shift by image_min exactly as scikit-image's (image - min)
    for _ in range(10000):
        if abs(t_next - t_curr) <= tolerance:
            break
        t_curr = t_next
        s = hipops.masked_sums(d, ctx.asarray(np.array([t_curr]))).numpy()[0]
        mean_back = s[0] / s[1] - image_min
        mean_fore = s[2] / s[3] - image_min
        t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore)) + image_min
    return t_next


def _local_threshold(d: DeviceArray, block_size, method="gaussian", offset=0, mode="reflect", param=None, cval=0):
    """``skimage.filters.threshold_local`` (SK thresholding.py:206-236) -> float64 threshold image on the device."""
    if block_size % 2 == 0:
        raise ValueError(
            "The kwarg ``block_size`` must be odd! Given ``block_size`` {0} is even.".format(block_size)
        )
    if method == "gaussian":
        sigma = (block_size - 1) / 6.0 if param is None else param
        # scikit-image filters EVERY axis of an n-D image (ndi.gaussian_filter on the whole stack)
        t = hipops.gaussian_nd(d, sigma, mode=mode, cval=cval, scale=1.0)
    elif d.ndim != 2:
        raise NotImplementedError(
            f"threshold_local(method='{method}') filters every axis of an n-D image in scikit-image; the device path "
            "does that for method='gaussian' only -- map a stack over its first axis with Pipeline(parallel=True)"
        )
    elif method == "mean":
        t = hipops.uniform_filter(d, block_size, mode=mode, cval=cval)
    elif method == "median":
        t = hipops.median(d, np.ones((block_size, block_size), np.uint8), mode=mode, cval=cval)
        if t.dtype != np.float64:
            t = hipops.to_float64(t)
    else:
        raise ValueError(f"threshold_local method '{method}' is not supported on the device path")
    if offset:
        t = hipops.add_scalar(t, -float(offset))
    return t


@device_operator
def apply_threshold(
    intensities: ScalarArray,
    method: Literal[
        "otsu", "li", "yen", "isodata", "mean", "minimum", "triangle", "local", "niblack", "sauvola"
    ] = "otsu",
    **kwargs,
) -> BoolArray:
    """``intensities > threshold_<method>(intensities, **kwargs)`` (R/operations.py:135-216); boolean result."""
    if not isinstance(intensities, DeviceArray) and np.asarray(intensities).size == 0:
        return np.zeros_like(intensities, dtype=bool)
    method_lower = method.lower()
    if method_lower not in _SUPPORTED_METHODS:
        raise ValueError(
            f"Unsupported thresholding method: '{method}'. "
            f"Supported methods: {', '.join(_SUPPORTED_METHODS)}"
        )
    src_dtype = intensities.dtype if isinstance(intensities, DeviceArray) else np.asarray(intensities).dtype
    shifted = [0]
    hist_method = method_lower in ("otsu", "yen", "isodata", "triangle", "minimum", "mean", "li")
    d, was_numpy = _to_device(intensities, "apply_threshold", integer_histogram=hist_method,
                              shifted=shifted if hist_method else None)
    offset = int(shifted[0])
    wide_bins = int(shifted[1]) if len(shifted) > 1 else 0  # integer image beyond a 65,536-value range
    shape = d.shape
    if d.ndim != 2 and method_lower not in ("local", "niblack", "sauvola"):
        d = _flat(d)  # global methods: ONE threshold from the histogram of the whole stack
    ctx = d.ctx
    lo, hi = _min_max(d)
    if lo == hi:  # constant image (R/operations.py:201-202)
        z = np.zeros(shape, dtype=bool)
        return z if was_numpy else ctx.asarray(z)
    kw = dict(kwargs)
    if wide_bins:
        # scikit-image's histogram of such an image has one bin per integer from min to max; the counts come from the
        # device, the selection runs on them with the same numpy expressions, and x > t is exact on the float64 image
        counts = hipops.histogram_range(_flat(d), offset, wide_bins).numpy()[0].astype(np.int64)
        centers = np.arange(offset, offset + wide_bins)
        if method_lower == "otsu":
            kw.pop("nbins", None)
            t = _thresholds.otsu(counts, centers)
        elif method_lower == "mean":
            t = _thresholds.mean_from_hist(counts, centers)
        elif method_lower == "li":
            t = _thresholds.li_from_hist(counts, centers, wrap_dtype=src_dtype, **kw)
        else:
            kw.pop("nbins", None)
            t = getattr(_thresholds, method_lower)(counts, centers, **kw)
        if np.isnan(t):
            z = np.zeros(shape, dtype=bool)
            return z if was_numpy else ctx.asarray(z)
        mask = hipops.greater_than(d, ctx.asarray(np.array([float(t)])))
    elif offset:
        # integer image beyond uint16 whose range fits: histogram of x - min on the device, the reference's threshold
        # from (counts, arange(min, max + 1)) on the host, and x > t  <=>  x - min > floor(t) - min for integers
        hist = hipops.histogram_u16(d).numpy()[0]
        counts, centers = _thresholds.counts_centers_u16(hist)
        centers = centers + offset
        if method_lower == "otsu":
            kw.pop("nbins", None)
            t = _thresholds.otsu(counts, centers)
        elif method_lower == "mean":
            t = _thresholds.mean_from_hist(counts, centers)
        elif method_lower == "li":
            t = _thresholds.li_from_hist(counts, centers, wrap_dtype=src_dtype, **kw)
        else:
            kw.pop("nbins", None)
            t = getattr(_thresholds, method_lower)(counts, centers, **kw)
        if np.isnan(t):  # x > nan is False everywhere (li on a signed image that wrapped in its own dtype)
            z = np.zeros(shape, dtype=bool)
            return z if was_numpy else ctx.asarray(z)
        mask = hipops.greater_than(d, ctx.asarray(np.array([float(int(np.floor(t)) - offset)])))
    elif method_lower == "otsu":
        thr = hipops.threshold_otsu(d, nbins=int(kw.pop("nbins", 256)))
        mask = hipops.greater_than(d, thr)
    elif method_lower in ("yen", "isodata", "triangle", "minimum"):
        t = _global_threshold(d, method_lower, kw)
        mask = hipops.greater_than(d, ctx.asarray(np.array([float(t)])))
    elif method_lower == "mean":
        mask = hipops.greater_than(d, ctx.asarray(np.array([float(_mean_threshold(d))])))
    elif method_lower == "li":
        mask = hipops.greater_than(d, ctx.asarray(np.array([float(_li_threshold(d, **kw))])))
    elif method_lower == "local":
        if "block_size" not in kw:
            raise TypeError("threshold_local() missing 1 required positional argument: 'block_size'")
        mask = hipops.greater_than_image(d, _local_threshold(d, **kw))
    else:  # niblack / sauvola (SK/filters/thresholding.py:967-1087)
        if method_lower == "sauvola" and kw.get("r") is None:
            kw["r"] = _sauvola_r(src_dtype)  # half the range of the CALLER's dtype, not of the uint16 it travels as
        # the window spans EVERY axis of a stack, as in scikit-image (window_size: one odd integer or one per axis)
        mask = hipops.greater_than_image(d, hipops.window_threshold(d, method=method_lower, nd=d.ndim > 2, **kw))
    if mask.shape != shape:
        is_bool = mask.is_bool
        mask = mask.reshape(shape)
        mask.is_bool = is_bool
    return _result(mask, was_numpy)
