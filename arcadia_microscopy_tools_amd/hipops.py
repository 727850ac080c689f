"""Batch device operators: thin, shape-checked Python wrappers over the C ABI (include/amt_hip.h).

Every function takes ``DeviceArray`` inputs whose LAST TWO axes are (Y, X); all leading axes are
independent planes processed by one launch.  ``out=`` lets a caller reuse buffers (no allocation, no
implicit synchronisation inside a pipeline).  Nothing here touches the host except where stated
(small parameter tables; results are only copied back by ``DeviceArray.numpy()``).

The functions mirror the scikit-image / scipy calls the reference makes (file:line in each docstring).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _hip
from .device import Context, DeviceArray

_c_double_p = ctypes.POINTER(ctypes.c_double)


def _lib():
    return _hip.load_library()


def _planes(a: DeviceArray):
    if a.ndim < 2:
        raise ValueError(f"expected at least a 2-D array, got shape {a.shape}")
    H, W = a.shape[-2:]
    return a.nplanes, int(H), int(W)


def _out(ctx: Context, out, shape, dtype) -> DeviceArray:
    if out is None:
        return ctx.empty(shape, dtype)
    if tuple(out.shape) != tuple(shape) or out.dtype != np.dtype(dtype):
        raise ValueError(f"out has shape {out.shape} dtype {out.dtype}, need {tuple(shape)} {np.dtype(dtype)}")
    if out.ctx is not ctx:
        # the operator runs on the INPUT's stream: an output owned by another context would be written without any
        # ordering against that context's work
        raise ValueError("out belongs to another context than the input; bind the input with DeviceArray.on(ctx)")
    return out


def _in_code(a: DeviceArray) -> int:
    if a.dtype == np.uint16:
        return _hip.U16
    if a.dtype == np.float64:
        return _hip.F64
    raise TypeError(f"device filters accept uint16 or float64 images, got {a.dtype}")


def _host_f64(arr):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------------
# filters
# --------------------------------------------------------------------------------------------------
def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """The 1-D kernel scipy builds (SP/_filters.py:226-236,314-323); computed with the host's numpy so the
    device shares np.exp's rounding with the CPU path."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x**2)
    return phi / phi.sum()


def gaussian(a: DeviceArray, sigma: float, mode: str = "nearest", cval: float = 0.0, truncate: float = 4.0,
             scale: float | None = None, out: DeviceArray | None = None, channel: int | None = None,
             minmax_out: DeviceArray | None = None) -> DeviceArray:
    """``skimage.filters.gaussian`` per plane (SK/filters/_gaussian.py:119-126): uint16 input is first
    converted like ``img_as_float`` (x * (1/65535), SK/util/dtype.py:319) unless ``scale`` is given."""
    ctx = a.ctx
    n, H, W = _planes(a)
    ptr, stride, oshape = a.ptr, 0, a.shape
    if channel is not None:  # a is (..., C, Y, X): filter that channel of every stack in one launch
        if a.ndim < 3:
            raise ValueError("channel= needs a (..., C, Y, X) array")
        C = a.shape[-3]
        n, stride, oshape = n // C, C * H * W, a.shape[:-3] + (H, W)
        ptr = a.ptr + int(channel) * H * W * a.dtype.itemsize
    o = _out(ctx, out, oshape, np.float64)
    if sigma <= 1e-15:  # scipy skips axes with sigma <= 1e-15 (SP/_filters.py:423)
        w = np.ones(1)
    else:
        w = gaussian_weights(sigma, truncate)
    r = (len(w) - 1) // 2
    wa, wp = _host_f64(w)
    if scale is None:
        scale = 1.0 / 65535 if a.dtype == np.uint16 else 1.0
    if minmax_out is not None and (minmax_out.dtype != np.float64 or minmax_out.size != 2 * n):
        raise ValueError("minmax_out must be a float64 DeviceArray with 2 values per output plane")
    _hip.check(_lib().amt_gaussian(ctx.handle, ptr, _in_code(a), float(scale), o.ptr, n, H, W, wp, r,
                                   _hip.MODES[mode], float(cval), stride,
                                   minmax_out.ptr if minmax_out is not None else None), "amt_gaussian")
    return o


def gaussian_nd(a: DeviceArray, sigma: float, mode: str = "nearest", cval: float = 0.0, truncate: float = 4.0,
                out: DeviceArray | None = None, scale: float | None = None) -> DeviceArray:
    """``skimage.filters.gaussian`` of ONE n-D image the way scikit-image / scipy filter it: EVERY axis, leading axes
    first (SP/_filters.py:412-430), the intermediate kept in float64 -- unlike ``gaussian``, whose leading axes are
    independent planes.  2-D arrays go straight to ``gaussian``."""
    if a.ndim <= 2:
        return gaussian(a, sigma, mode, cval, truncate, scale=scale, out=out)
    ctx = a.ctx
    if sigma <= 1e-15:
        return gaussian(a, sigma, mode, cval, truncate, scale=scale, out=out)
    w = gaussian_weights(sigma, truncate)
    r = (len(w) - 1) // 2
    wa, wp = _host_f64(w)
    if scale is None:
        scale = 1.0 / 65535 if a.dtype == np.uint16 else 1.0
    cur, cur_scale = a, scale
    shape = a.shape
    for k in range(a.ndim - 2):  # the leading axes, one 1-D pass each
        outer = int(np.prod(shape[:k], dtype=np.int64)) if k else 1
        L, inner = int(shape[k]), int(np.prod(shape[k + 1:], dtype=np.int64))
        nxt = ctx.empty(shape, np.float64)
        _hip.check(_lib().amt_convolve_axis0(ctx.handle, cur.ptr, _in_code(cur), float(cur_scale), nxt.ptr, outer, L, inner,
                                             wp, r, _hip.MODES[mode], float(cval)), "amt_convolve_axis0")
        cur, cur_scale = nxt, 1.0
    return gaussian(cur, sigma, mode, cval, truncate, scale=cur_scale, out=out)


def gaussian_otsu_codes_supported(a: DeviceArray, sigma: float, mode: str = "nearest", truncate: float = 4.0,
                                  channel: int | None = None) -> bool:
    """Whether ``gaussian_otsu_codes`` can take this batch (uint16, fused-radius Gaussian, aligned rows)."""
    if a.dtype != np.uint16 or sigma <= 1e-15:
        return False
    _, H, W = _planes(a)
    r = int(truncate * float(sigma) + 0.5)
    stride = a.shape[-3] * H * W if channel is not None else 0
    ptr = a.ptr + (int(channel) * H * W * 2 if channel is not None else 0)
    return bool(_lib().amt_gaussian_otsu_codes_supported(H, W, r, _hip.MODES[mode], stride)) and ptr % 16 == 0


def gaussian_otsu_codes(a: DeviceArray, sigma: float, codes: DeviceArray, thr: DeviceArray, thr_code: DeviceArray,
                        minmax: DeviceArray, hist: DeviceArray, mode: str = "nearest", truncate: float = 4.0,
                        channel: int | None = None) -> DeviceArray:
    """``threshold_otsu(gaussian(a, sigma))`` per plane WITHOUT the float64 image: ``thr`` receives the thresholds,
    ``codes`` (uint16, one per pixel) and ``thr_code`` continue the chain -- ``gaussian(a) > thr`` is exactly
    ``codes > thr_code`` (csrc/amt_filters.hip, gauss_lds_kernel EPI 2), so ``threshold_open_close(codes, thr_code)``
    yields the mask of the separate operators.  ``minmax`` (n, 2) and ``hist`` (n, 256) receive np.histogram's range
    and counts.  All outputs are caller-owned device arrays (nothing is allocated per call)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    ptr, stride = a.ptr, 0
    if channel is not None:
        C = a.shape[-3]
        n, stride = n // C, C * H * W
        ptr = a.ptr + int(channel) * H * W * a.dtype.itemsize
    if a.dtype != np.uint16:
        raise TypeError("gaussian_otsu_codes takes uint16 images")
    for name, arr, dt, size in (("codes", codes, np.uint16, n * H * W), ("thr", thr, np.float64, n),
                                ("thr_code", thr_code, np.float64, n), ("minmax", minmax, np.float64, 2 * n),
                                ("hist", hist, np.uint32, 256 * n)):
        if arr.dtype != dt or arr.size != size:
            raise ValueError(f"{name} must be a {np.dtype(dt).name} DeviceArray of {size} elements")
        if arr.ctx is not ctx:
            raise ValueError(f"{name} belongs to another context than the input")
    w = gaussian_weights(sigma, truncate)
    r = (len(w) - 1) // 2
    wa, wp = _host_f64(w)
    _hip.check(_lib().amt_gaussian_otsu_codes(ctx.handle, ptr, 1.0 / 65535, n, H, W, wp, r, _hip.MODES[mode], stride,
                                              minmax.ptr, hist.ptr, thr.ptr, thr_code.ptr, codes.ptr),
               "amt_gaussian_otsu_codes")
    return codes


def difference_of_gaussians(a: DeviceArray, low_sigma: float, high_sigma: float, mode: str = "nearest",
                            cval: float = 0.0, truncate: float = 4.0, out: DeviceArray | None = None,
                            scale: float | None = None) -> DeviceArray:
    """``skimage.filters.difference_of_gaussians`` (SK/filters/_gaussian.py:258-290; R/operations.py:91).  ``scale``:
    the factor of ``img_as_float`` (default: 1/65535 for uint16 planes, 1 for float64)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, a.shape, np.float64)
    wl, wh = gaussian_weights(low_sigma, truncate), gaussian_weights(high_sigma, truncate)
    wla, wlp = _host_f64(wl)
    wha, whp = _host_f64(wh)
    if scale is None:
        scale = 1.0 / 65535 if a.dtype == np.uint16 else 1.0
    _hip.check(_lib().amt_dog(ctx.handle, a.ptr, _in_code(a), scale, o.ptr, n, H, W, wlp, (len(wl) - 1) // 2, whp,
                              (len(wh) - 1) // 2, _hip.MODES[mode], float(cval)), "amt_dog")
    return o


def difference_of_gaussians_nd(a: DeviceArray, low_sigma: float, high_sigma: float, mode: str = "nearest",
                               cval: float = 0.0, truncate: float = 4.0, out: DeviceArray | None = None,
                               scale: float | None = None) -> DeviceArray:
    """``skimage.filters.difference_of_gaussians`` of ONE n-D image: both Gaussians filter EVERY axis
    (SK/filters/_gaussian.py:284-290), unlike ``difference_of_gaussians``, whose leading axes are independent planes."""
    ctx = a.ctx
    lo = gaussian_nd(a, low_sigma, mode, cval, truncate, scale=scale)
    hi = gaussian_nd(a, high_sigma, mode, cval, truncate, scale=scale)
    o = _out(ctx, out, a.shape, np.float64)
    _hip.check(_lib().amt_subtract(ctx.handle, lo.ptr, hi.ptr, o.ptr, _hip.F64, a.size), "amt_subtract")
    return o


def sub_clip0(a: DeviceArray, level: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    """``np.clip(a - level, 0, None)`` with one level per plane (R/operations.py:97)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, a.shape, np.float64)
    _hip.check(_lib().amt_sub_clip0_f64(ctx.handle, a.ptr, level.ptr, o.ptr, n, H * W), "amt_sub_clip0_f64")
    return o


def rescale(a: DeviceArray, in_range: DeviceArray, out_range=(0.0, 1.0), out: DeviceArray | None = None):
    """``skimage.exposure.rescale_intensity(a, in_range=(p1, p2), out_range=(lo, hi))`` per plane with the
    (p1, p2) pairs on the device (SK/exposure/exposure.py:405-428; R/operations.py:50)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, a.shape, np.float64)
    _hip.check(_lib().amt_rescale(ctx.handle, a.ptr, _in_code(a), in_range.ptr, float(out_range[0]),
                                  float(out_range[1]), o.ptr, n, H * W), "amt_rescale")
    return o


def uniform_filter(a: DeviceArray, size: int, mode: str = "reflect", cval: float = 0.0, out=None) -> DeviceArray:
    """``ndi.convolve1d(x, ones(size)/size)`` along axis 0 then axis 1 (threshold_local's 'mean',
    SK/filters/thresholding.py:224-229); the image is NOT rescaled (no img_as_float)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if size % 2 == 0:
        raise ValueError("uniform_filter: size must be odd")
    o = _out(ctx, out, a.shape, np.float64)
    w = 1.0 / size * np.ones((size,))
    wa, wp = _host_f64(w)
    _hip.check(_lib().amt_gaussian(ctx.handle, a.ptr, _in_code(a), 1.0, o.ptr, n, H, W, wp, (size - 1) // 2,
                                   _hip.MODES[mode], float(cval), 0, None), "amt_gaussian")
    return o


def to_float64(a: DeviceArray, scale: float = 1.0, out=None) -> DeviceArray:
    """uint16 -> float64 (``x * scale``; ``img_as_float`` uses 1/65535, SK/util/dtype.py:319)."""
    if a.dtype != np.uint16:
        raise TypeError("to_float64 expects uint16")
    o = _out(a.ctx, out, a.shape, np.float64)
    _hip.check(_lib().amt_convert_u16_f64(a.ctx.handle, a.ptr, float(scale), o.ptr, a.size), "amt_convert_u16_f64")
    return o


def add_scalar(a: DeviceArray, s: float, out=None) -> DeviceArray:
    o = _out(a.ctx, out, a.shape, np.float64)
    _hip.check(_lib().amt_add_scalar_f64(a.ctx.handle, a.ptr, float(s), o.ptr, a.size), "amt_add_scalar_f64")
    return o


def deinterleave(yxc: DeviceArray, C: int, out: DeviceArray | None = None) -> DeviceArray:
    """(..., Y, X, C) interleaved ND2 frames -> (..., C, Y, X) (R/nikon.py:25-43)."""
    ctx = yxc.ctx
    if yxc.ndim < 3 or yxc.shape[-1] != C:
        raise ValueError("deinterleave expects (..., Y, X, C)")
    H, W = yxc.shape[-3:-1]
    lead = yxc.shape[:-3]
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    o = _out(ctx, out, tuple(lead) + (C, H, W), np.uint16)
    _hip.check(_lib().amt_deinterleave_u16(ctx.handle, yxc.ptr, o.ptr, n, H, W, C), "amt_deinterleave_u16")
    return o


# --------------------------------------------------------------------------------------------------
# statistics and thresholds
# --------------------------------------------------------------------------------------------------
def histogram_u16(a: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    """65536-bin uint32 histogram per plane (np.bincount of SK/exposure/exposure.py:70)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, (n, 65536), np.uint32)
    _hip.check(_lib().amt_hist_u16(ctx.handle, a.ptr, o.ptr, n, H * W), "amt_hist_u16")
    return o


def histogram_range(a: DeviceArray, lo: int, nbins: int) -> DeviceArray:
    """One bin per integer ``lo .. lo + nbins - 1`` of an integer-valued float64 array (per plane): scikit-image's
    histogram of integer images whose range exceeds uint16 (SK/exposure/exposure.py:63-74)."""
    ctx = a.ctx
    if a.dtype != np.float64:
        raise TypeError("histogram_range expects the float64 image an integer image beyond uint16 travels as")
    n, H, W = _planes(a)
    o = ctx.empty((n, int(nbins)), np.uint32)
    _hip.check(_lib().amt_hist_range_f64(ctx.handle, a.ptr, float(lo), int(nbins), o.ptr, n, H * W), "amt_hist_range_f64")
    return o


def minmax(a: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, (n, 2), np.float64)
    _hip.check(_lib().amt_minmax_f64(ctx.handle, a.ptr, o.ptr, n, H * W), "amt_minmax_f64")
    return o


def histogram_f64(a: DeviceArray, nbins: int = 256, mm: DeviceArray | None = None, out: DeviceArray | None = None):
    """``np.histogram(a, bins=nbins)`` counts per plane + the (min, max) pairs (SK/exposure/exposure.py:139)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if mm is None:
        mm = minmax(a)
    o = _out(ctx, out, (n, nbins), np.uint32)
    _hip.check(_lib().amt_hist_f64(ctx.handle, a.ptr, mm.ptr, o.ptr, nbins, n, H * W), "amt_hist_f64")
    return o, mm


def percentile(a: DeviceArray, q, out: DeviceArray | None = None) -> DeviceArray:
    """``np.percentile(a, q)`` (method 'linear') per plane -> (nplanes, len(q)) float64 on the device."""
    ctx = a.ctx
    n, H, W = _planes(a)
    qa = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if np.any(qa < 0) or np.any(qa > 100):
        raise ValueError("Percentiles must be in the range [0, 100]")
    o = _out(ctx, out, (n, len(qa)), np.float64)
    qh, qp = _host_f64(qa)
    if a.dtype == np.uint16:
        _hip.check(_lib().amt_percentile_u16(ctx.handle, a.ptr, qp, len(qa), o.ptr, n, H * W), "amt_percentile_u16")
    elif a.dtype == np.float64:
        _hip.check(_lib().amt_percentile_f64(ctx.handle, a.ptr, qp, len(qa), o.ptr, n, H * W), "amt_percentile_f64")
    else:
        raise TypeError(f"percentile: unsupported dtype {a.dtype}")
    return o


def masked_sums(a: DeviceArray, thr: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    """Per plane {sum(a <= t), count(a <= t), sum(a > t), count(a > t)} for float64 images (bit-reproducible)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if a.dtype != np.float64:
        raise TypeError("masked_sums expects float64")
    o = _out(ctx, out, (n, 4), np.float64)
    _hip.check(_lib().amt_masked_sums_f64(ctx.handle, a.ptr, thr.ptr, o.ptr, n, H * W), "amt_masked_sums_f64")
    return o


def crop(a: DeviceArray, top: int, left: int, h: int, w: int, out: DeviceArray | None = None) -> DeviceArray:
    """``a[..., top:top+h, left:left+w]`` as a new device array (R/operations.py:132)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, a.shape[:-2] + (h, w), a.dtype)
    _hip.check(_lib().amt_copy_rect(ctx.handle, a.ptr, o.ptr, a.dtype.itemsize, n, H, W, int(top), int(left), int(h),
                                    int(w)), "amt_copy_rect")
    o.is_bool = a.is_bool
    return o


def pad_edge(a: DeviceArray, py: int, px: int) -> DeviceArray:
    """``np.pad(a, ((py, py), (px, px)), mode='edge')`` per plane."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = ctx.empty(a.shape[:-2] + (H + 2 * py, W + 2 * px), a.dtype)
    _hip.check(_lib().amt_pad_edge(ctx.handle, a.ptr, o.ptr, a.dtype.itemsize, n, H, W, int(py), int(px)), "amt_pad_edge")
    o.is_bool = a.is_bool
    return o


def threshold_otsu(a: DeviceArray, nbins: int = 256, out: DeviceArray | None = None,
                   minmax: DeviceArray | None = None) -> DeviceArray:
    """``skimage.filters.threshold_otsu`` per plane, value stays on the device (SK/filters/thresholding.py:321-350).
    ``minmax`` = per-plane [min, max] already known for a float64 image (``gaussian(..., minmax_out=)``)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, (n,), np.float64)
    if minmax is not None and (a.dtype != np.float64 or minmax.dtype != np.float64 or minmax.size != 2 * n):
        raise ValueError("minmax needs a float64 image and 2 float64 values per plane")
    _hip.check(_lib().amt_threshold_value(ctx.handle, a.ptr, _in_code(a), _hip.THR_OTSU, nbins, o.ptr, None, n, H * W,
                                          minmax.ptr if minmax is not None else None), "amt_threshold_value")
    return o


def greater_than(a: DeviceArray, thr: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    """``a > thr[plane]`` -> uint8 0/1 mask (R/operations.py:216)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    o = _out(ctx, out, a.shape, np.uint8)
    _hip.check(_lib().amt_threshold_gt(ctx.handle, a.ptr, _in_code(a), thr.ptr, o.ptr, n, H * W), "amt_threshold_gt")
    o.is_bool = True
    return o


def window_threshold(a: DeviceArray, window_size=15, method: str = "niblack", k: float = 0.2, r=None,
                     out: DeviceArray | None = None, nd: bool = False) -> DeviceArray:
    """``skimage.filters.threshold_niblack`` / ``threshold_sauvola`` threshold image (float64)
    (SK/filters/thresholding.py:967-1087).  Default: per plane, ``window_size`` an odd integer or (rows, columns).
    ``nd=True``: ``a`` is ONE n-D image and the window spans every axis, as scikit-image treats a stack -- an odd
    integer (the same on every axis) or one value per axis.  ``r`` defaults to half the dtype range as in scikit-image."""
    ctx = a.ctx
    n, H, W = _planes(a)
    nlead = a.ndim - 2 if nd else 0
    if isinstance(window_size, (tuple, list, np.ndarray)):
        if len(window_size) != 2 + nlead:
            raise ValueError("window_size must be an integer or one value per image axis"
                             + ("" if nd else " (rows, columns)"))
        ws = [int(v) for v in window_size]
    else:
        ws = [int(window_size)] * (2 + nlead)
    for v in ws:
        if v % 2 == 0:
            raise ValueError(f"Window size {v} is even.")
    wy, wx = ws[-2], ws[-1]
    if r is None:
        r = 0.5 * 65535 if a.dtype == np.uint16 else 1.0  # 0.5 * (imax - imin); float range is (-1, 1)
    o = _out(ctx, out, a.shape, np.float64)
    meth = 0 if method == "niblack" else 1
    if nlead and any(v != 1 for v in ws[:nlead]):
        import ctypes

        shp = (ctypes.c_int * nlead)(*[int(v) for v in a.shape[:nlead]])
        win = (ctypes.c_int * nlead)(*ws[:nlead])
        _hip.check(_lib().amt_window_threshold_nd(ctx.handle, a.ptr, _in_code(a), o.ptr, nlead, shp, win, H, W, wy, wx,
                                                  meth, float(k), float(r)), "amt_window_threshold")
    else:  # windows of one sample along the leading axes: every plane on its own
        _hip.check(_lib().amt_window_threshold_yx(ctx.handle, a.ptr, _in_code(a), o.ptr, n, H, W, wy, wx, meth, float(k),
                                                  float(r)), "amt_window_threshold")
    return o


def greater_than_image(a: DeviceArray, thr_image: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    ctx = a.ctx
    o = _out(ctx, out, a.shape, np.uint8)
    _hip.check(_lib().amt_threshold_gt_image(ctx.handle, a.ptr, _in_code(a), thr_image.ptr, o.ptr, a.size),
               "amt_threshold_gt_image")
    o.is_bool = True
    return o


# --------------------------------------------------------------------------------------------------
# morphology
# --------------------------------------------------------------------------------------------------
def disk(radius: int) -> np.ndarray:
    """``skimage.morphology.disk`` (X**2 + Y**2 <= r**2)."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X**2 + Y**2) <= radius**2, dtype=np.uint8)


def cross3() -> np.ndarray:
    """skimage's default footprint: ndi.generate_binary_structure(2, 1)."""
    return np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)


def _fp(footprint, even: str = "scipy"):
    """The footprint as a C-contiguous uint8 array with ODD sides (what the C ABI takes).  An even side is padded with
    a zero row / column, which is exact: scipy centres a filter of even size s at s // 2, i.e. where the odd filter
    of size s + 1 with the zero line APPENDED is centred (``even="scipy"``: ndi.grey_erosion / grey_dilation /
    median_filter / binary_erosion / binary_dilation all agree with their padded form, dilation's origin shift
    included); scikit-image's grey erosion / dilation PREPEND it (``even="skimage"``, SK/morphology/grey.py:14-50
    with shift_x = shift_y = False) and the second halves of its opening / closing append it (``"skimage-shift"``)."""
    fp = np.asarray(cross3() if footprint is None else np.asarray(footprint) != 0, dtype=np.uint8)
    if fp.ndim != 2:
        raise ValueError("footprint must be 2-D")
    m, n = fp.shape
    if m % 2 == 0:
        z = np.zeros((1, n), np.uint8)
        fp = np.vstack((z, fp)) if even == "skimage" else np.vstack((fp, z))
        m += 1
    if n % 2 == 0:
        z = np.zeros((m, 1), np.uint8)
        fp = np.hstack((z, fp)) if even == "skimage" else np.hstack((fp, z))
    return np.ascontiguousarray(fp)


def _binary(which: str, a: DeviceArray, footprint, out, border_value=None):
    ctx = a.ctx
    if a.dtype != np.uint8:
        raise TypeError("binary morphology expects a uint8 / bool mask on the device")
    n, H, W = _planes(a)
    fp = _fp(footprint)
    o = _out(ctx, out, a.shape, np.uint8)
    fpp = fp.ctypes.data_as(ctypes.c_void_p)
    lib = _lib()
    if which == "erode":
        rc = lib.amt_binary_erode(ctx.handle, a.ptr, o.ptr, n, H, W, fpp, fp.shape[0], fp.shape[1],
                                  1 if border_value is None else int(border_value))
    elif which == "dilate":
        rc = lib.amt_binary_dilate(ctx.handle, a.ptr, o.ptr, n, H, W, fpp, fp.shape[0], fp.shape[1],
                                   0 if border_value is None else int(border_value))
    elif which == "open":
        rc = lib.amt_binary_open(ctx.handle, a.ptr, o.ptr, n, H, W, fpp, fp.shape[0], fp.shape[1])
    else:
        rc = lib.amt_binary_close(ctx.handle, a.ptr, o.ptr, n, H, W, fpp, fp.shape[0], fp.shape[1])
    _hip.check(rc, "amt_binary_" + which)
    o.is_bool = True
    return o


def binary_erosion(a, footprint=None, out=None):
    """``skimage.morphology.binary_erosion`` (SK/morphology/binary.py:42; outside counts as True)."""
    return _binary("erode", a, footprint, out)


def binary_dilation(a, footprint=None, out=None):
    """``skimage.morphology.binary_dilation`` (SK/morphology/binary.py:77)."""
    return _binary("dilate", a, footprint, out)


def binary_opening(a, footprint=None, out=None):
    """``skimage.morphology.binary_opening`` = dilation(erosion(a)), fused (SK/morphology/binary.py:82-113)."""
    return _binary("open", a, footprint, out)


def binary_closing(a, footprint=None, out=None):
    """``skimage.morphology.binary_closing`` = erosion(dilation(a)), fused (SK/morphology/binary.py:116-147)."""
    return _binary("close", a, footprint, out)


def threshold_otsu_bins(a: DeviceArray, minmax: DeviceArray, thr: DeviceArray, thr_code: DeviceArray,
                        bins: DeviceArray) -> DeviceArray:
    """``threshold_otsu`` of float64 planes whose [min, max] is known, leaving every sample's 256-bin index in the uint8
    plane ``bins`` and the threshold's bin (times two) in ``thr_code``: ``threshold_open_close(a, thr, bins=bins,
    thr_code=thr_code)`` then compares by the byte plane (amt_otsu_f64_bins)."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if a.dtype != np.float64 or bins.dtype != np.uint8 or bins.size != a.size:
        raise ValueError("threshold_otsu_bins: float64 image and a uint8 bin plane of the same size")
    for arr, size in ((minmax, 2 * n), (thr, n), (thr_code, n)):
        if arr.dtype != np.float64 or arr.size != size:
            raise ValueError("threshold_otsu_bins: minmax (n, 2), thr (n,), thr_code (n,) must be float64")
    _hip.check(_lib().amt_otsu_f64_bins(ctx.handle, a.ptr, minmax.ptr, thr.ptr, thr_code.ptr, bins.ptr, n, H * W),
               "amt_otsu_f64_bins")
    return thr


def threshold_open_close(a: DeviceArray, thr: DeviceArray, footprint=None, out=None, bins=None, thr_code=None) -> DeviceArray:
    """``binary_closing(binary_opening(a > thr[plane], fp), fp)`` as one packed chain (no intermediate masks).
    ``bins`` / ``thr_code`` from ``threshold_otsu_bins``: the comparison reads 1 byte per pixel instead of 8."""
    ctx = a.ctx
    n, H, W = _planes(a)
    fp = _fp(footprint)
    o = _out(ctx, out, a.shape, np.uint8)
    if bins is not None:
        _hip.check(_lib().amt_threshold_open_close_bins(ctx.handle, a.ptr, bins.ptr, thr.ptr, thr_code.ptr, o.ptr, n, H, W,
                                                        fp.ctypes.data_as(ctypes.c_void_p), fp.shape[0], fp.shape[1]),
                   "amt_threshold_open_close_bins")
        o.is_bool = True
        return o
    _hip.check(_lib().amt_threshold_open_close(ctx.handle, a.ptr, _in_code(a), thr.ptr, o.ptr, n, H, W,
                                               fp.ctypes.data_as(ctypes.c_void_p), fp.shape[0], fp.shape[1]),
               "amt_threshold_open_close")
    o.is_bool = True
    return o


def _rank(a: DeviceArray, footprint, op: int, mode: str, cval: float, out):
    ctx = a.ctx
    n, H, W = _planes(a)
    fp = _fp(footprint)  # callers that follow scikit-image's even-size rule pass an odd footprint already
    o = _out(ctx, out, a.shape, a.dtype)
    _hip.check(_lib().amt_rank_filter(ctx.handle, a.ptr, o.ptr, _in_code(a), n, H, W,
                                      fp.ctypes.data_as(ctypes.c_void_p), fp.shape[0], fp.shape[1], op,
                                      _hip.MODES[mode], float(cval)), "amt_rank_filter")
    return o


def erosion(a, footprint=None, out=None, shift: bool = False):
    """``skimage.morphology.erosion`` -> ndi.grey_erosion(footprint), mode 'reflect' (SK/morphology/grey.py:185);
    ``shift`` = scikit-image's shift_x = shift_y (only matters for even-sized footprints)."""
    return _rank(a, _fp(footprint, "skimage-shift" if shift else "skimage"), 0, "reflect", 0.0, out)


def dilation(a, footprint=None, out=None, shift: bool = False):
    """``skimage.morphology.dilation``: skimage mirrors the footprint and scipy mirrors it back
    (SK/morphology/grey.py:242-251), i.e. max over in[p + s] for s in the ORIGINAL (odd-padded) footprint."""
    fp = _fp(footprint, "skimage-shift" if shift else "skimage")
    return _rank(a, np.ascontiguousarray(fp[::-1, ::-1]), 1, "reflect", 0.0, out)


def _eccentric(a, footprint, first, second, out):
    """scikit-image's opening / closing: the second half runs with shift_x = shift_y = True, and for a footprint with an
    even side the image is first edge-padded by (side - 1) pixels along that axis and the result cropped back
    (SK/morphology/grey.py:84-127, :255-353)."""
    shape = (3, 3) if footprint is None else np.shape(footprint)
    py = shape[0] - 1 if shape[0] % 2 == 0 else 0
    px = shape[1] - 1 if shape[1] % 2 == 0 else 0
    if not (py or px):
        return second(first(a, footprint), footprint, out=out, shift=True)
    H, W = a.shape[-2:]
    res = second(first(pad_edge(a, py, px), footprint), footprint, shift=True)
    return crop(res, py, px, H, W, out=out)


def opening(a, footprint=None, out=None):
    """``skimage.morphology.opening`` = dilation(erosion(a), shift_x=True, shift_y=True) (SK/morphology/grey.py:255-303)."""
    return _eccentric(a, footprint, erosion, dilation, out)


def closing(a, footprint=None, out=None):
    """``skimage.morphology.closing`` = erosion(dilation(a), shift_x=True, shift_y=True) (SK/morphology/grey.py:305-353)."""
    return _eccentric(a, footprint, dilation, erosion, out)


def white_tophat(a, footprint=None, out=None):
    """``skimage.morphology.white_tophat`` -> ndi.white_tophat = a - grey_opening(a) with scipy's own
    opening (grey_erosion then grey_dilation, both with the footprint as given; SK/morphology/grey.py:425)."""
    fp = _fp(footprint)
    er = _rank(a, fp, 0, "reflect", 0.0, None)
    n, H, W = _planes(a)
    o = _out(a.ctx, out, a.shape, a.dtype)
    # the dilation stores image - dilation(erosion) directly (amt_rank_filter_sub)
    _hip.check(_lib().amt_rank_filter_sub(a.ctx.handle, er.ptr, a.ptr, o.ptr, _in_code(a), n, H, W,
                                          fp.ctypes.data_as(ctypes.c_void_p), fp.shape[0], fp.shape[1], 1,
                                          _hip.MODES["reflect"], 0.0), "amt_rank_filter_sub")
    return o


def median(a, footprint=None, mode: str = "nearest", cval: float = 0.0, out=None):
    """``skimage.filters.median`` -> ndi.median_filter(footprint, mode='nearest') (SK/filters/_median.py)."""
    fp = np.ones((3, 3), dtype=np.uint8) if footprint is None else footprint
    return _rank(a, fp, 2, mode, cval, out)


# --------------------------------------------------------------------------------------------------
# labelling
# --------------------------------------------------------------------------------------------------
def label(a: DeviceArray, connectivity: int = 2, out: DeviceArray | None = None, count: DeviceArray | None = None):
    """``skimage.measure.label`` per plane (default 8-connected; raster numbering) -> (int32 labels, counts).
    R/masks.py:63; SURVEY.md A.5."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if a.dtype == np.uint8:
        code = _hip.U8
    elif a.dtype == np.int32:
        code = _hip.I32
    else:
        raise TypeError(f"label: unsupported dtype {a.dtype}")
    o = _out(ctx, out, a.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    if code == _hip.U8 and a.is_bool:
        # a bool array (0 / 1 bytes by construction): the truth-value entry point, whose run-table path launches nothing
        # that stands by for other byte values
        _hip.check(_lib().amt_label_mask(ctx.handle, a.ptr, o.ptr, c.ptr, n, H, W, int(connectivity)), "amt_label_mask")
        return o, c
    _hip.check(_lib().amt_label(ctx.handle, a.ptr, code, o.ptr, c.ptr, n, H, W, int(connectivity)), "amt_label")
    return o, c


def label_sparse_capacity(H: int, W: int) -> int:
    return max(65536, (H * W) // 16)


def label_sparse(a: DeviceArray, connectivity: int = 2, capacity: int | None = None, out=None, count=None, keep=None):
    """``label`` for sparse uint8 masks (at most ``capacity`` foreground pixels per plane, default max(65536,
    plane size / 16)); a plane that overflows reports count -1.

    ``keep`` = (int32 (n, capacity) list, int32 (n,) counts), both owned by the caller, turns the full-plane clear of
    ``out`` into a clear of the pixels the previous call wrote (``amt_label_sparse_reuse``): ``out`` must then be the
    same array every time, zeroed once together with the counts."""
    ctx = a.ctx
    n, H, W = _planes(a)
    if a.dtype != np.uint8:
        raise TypeError("label_sparse expects a uint8 / bool mask")
    if capacity is None:
        capacity = label_sparse_capacity(H, W)
    o = _out(ctx, out, a.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    if keep is not None:
        klist, kcount = keep
        if out is None:
            raise ValueError("keep= needs the caller's persistent out= plane")
        if klist.dtype != np.int32 or klist.size != n * int(capacity) or kcount.dtype != np.int32 or kcount.size != n:
            raise ValueError("keep must be (int32 (n, capacity), int32 (n,))")
        _hip.check(_lib().amt_label_sparse_reuse(ctx.handle, a.ptr, o.ptr, c.ptr, n, H, W, int(connectivity),
                                                 int(capacity), klist.ptr, kcount.ptr), "amt_label_sparse_reuse")
        return o, c
    _hip.check(_lib().amt_label_sparse(ctx.handle, a.ptr, o.ptr, c.ptr, n, H, W, int(connectivity), int(capacity)),
               "amt_label_sparse")
    return o, c


def clear_border(labels: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
    """``skimage.segmentation.clear_border`` (buffer_size 0) per plane (R/masks.py:56)."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    if labels.dtype != np.int32:
        raise TypeError("clear_border expects int32 labels on the device")
    o = _out(ctx, out, labels.shape, np.int32)
    _hip.check(_lib().amt_clear_border(ctx.handle, labels.ptr, o.ptr, n, H, W), "amt_clear_border")
    return o


def relabel_sequential(labels: DeviceArray, max_label: int, out=None, count=None):
    """``skimage.segmentation.relabel_sequential`` per plane -> (labels, counts) (R/masks.py:65)."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    o = _out(ctx, out, labels.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    _hip.check(_lib().amt_relabel_sequential(ctx.handle, labels.ptr, o.ptr, c.ptr, n, H * W, int(max_label)),
               "amt_relabel_sequential")
    return o, c


def clear_border_relabel(labels: DeviceArray, max_label: int, out=None, count=None, nlabels: DeviceArray | None = None):
    """``relabel_sequential(clear_border(labels))`` in one pass (R/masks.py:56,65) for label images whose
    labels are each one connected component (outputs of ``label`` / ``watershed``).  ``nlabels`` (int32 per
    plane, on the device): the caller vouches that each plane holds exactly the labels 1..nlabels[plane], which
    spares the pass that looks for the labels present."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    o = _out(ctx, out, labels.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    if nlabels is not None and (nlabels.dtype != np.int32 or nlabels.size != n):
        raise ValueError("nlabels must hold one int32 per plane")
    _hip.check(_lib().amt_clear_border_relabel(ctx.handle, labels.ptr, o.ptr, c.ptr, n, H, W, int(max_label),
                                               nlabels.ptr if nlabels is not None else None),
               "amt_clear_border_relabel")
    return o, c


def keep_labels(labels: DeviceArray, keep: DeviceArray, max_label: int, out=None) -> DeviceArray:
    """``np.where(np.isin(labels, kept), labels, 0)`` with keep = (nplanes, max_label+1) uint8 (R/masks.py:399-403)."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    o = _out(ctx, out, labels.shape, np.int32)
    _hip.check(_lib().amt_keep_labels(ctx.handle, labels.ptr, keep.ptr, o.ptr, n, H * W, int(max_label)),
               "amt_keep_labels")
    return o


def to_int64(labels: DeviceArray, out=None) -> DeviceArray:
    o = _out(labels.ctx, out, labels.shape, np.int64)
    _hip.check(_lib().amt_cast_i32_i64(labels.ctx.handle, labels.ptr, o.ptr, labels.size), "amt_cast_i32_i64")
    return o


def max_per_plane(labels: DeviceArray, out=None) -> DeviceArray:
    ctx = labels.ctx
    n, H, W = _planes(labels)
    o = _out(ctx, out, (n,), np.int32)
    _hip.check(_lib().amt_max_i32(ctx.handle, labels.ptr, o.ptr, n, H * W), "amt_max_i32")
    return o


# --------------------------------------------------------------------------------------------------
# distance transform, markers, watershed
# --------------------------------------------------------------------------------------------------
def edt(mask: DeviceArray, want_d2: bool = True, want_edt: bool = True, d2_out=None, edt_out=None):
    """``scipy.ndimage.distance_transform_edt`` per plane -> (d2 int32 | None, edt float64 | None)."""
    ctx = mask.ctx
    n, H, W = _planes(mask)
    d2 = _out(ctx, d2_out, mask.shape, np.int32) if want_d2 else None
    e = _out(ctx, edt_out, mask.shape, np.float64) if want_edt else None
    _hip.check(_lib().amt_edt(ctx.handle, mask.ptr, d2.ptr if d2 else None, e.ptr if e else None, n, H, W), "amt_edt")
    return d2, e


def peak_mask(d2: DeviceArray, mask: DeviceArray, min_distance: int = 5, out=None, keep=None, status=None) -> DeviceArray:
    """Peaks of the EDT per the config-3 marker recipe.  ``keep`` (the lists ``label_sparse(keep=)``
    maintains) + ``status`` (that call's counts) turn the clear of the persistent ``out`` plane into a clear of the
    previous run's peaks (``amt_peak_mask_reuse``)."""
    ctx = d2.ctx
    n, H, W = _planes(d2)
    o = _out(ctx, out, d2.shape, np.uint8)
    if keep is not None:
        if out is None or status is None:
            raise ValueError("keep= needs the caller's persistent out= plane and the previous counts (status=)")
        klist, kcount = keep
        _hip.check(_lib().amt_peak_mask_reuse(ctx.handle, d2.ptr, mask.ptr, o.ptr, n, H, W, int(min_distance), klist.ptr,
                                              kcount.ptr, klist.size // n, status.ptr), "amt_peak_mask_reuse")
        o.is_bool = True
        return o
    _hip.check(_lib().amt_peak_mask(ctx.handle, d2.ptr, mask.ptr, o.ptr, n, H, W, int(min_distance)), "amt_peak_mask")
    o.is_bool = True
    return o


def _ws_ties(ctx, n: int, ties_policy: str, ties_out):
    """The per-plane tie flags: the caller's array, a fresh one when the policy needs them, else none (the hot path
    allocates nothing)."""
    if ties_policy not in _hip.WS_TIES:
        raise ValueError(f"ties must be one of {sorted(_hip.WS_TIES)}, got {ties_policy!r}")
    if ties_out is not None:
        return _out(ctx, ties_out, (n,), np.int32)
    return ctx.empty((n,), np.int32) if ties_policy in ("report", "refuse") else None


def _ws_finish(ctx, ties_policy: str, ties, what: str):
    if ties_policy == "refuse":
        t = ties.numpy()
        if t.any():
            raise ValueError(
                f"{what}: equal-valued markers inside one mask component in plane(s) {np.flatnonzero(t).tolist()}: "
                "scikit-image orders them by the moves of its binary heap; use ties='exact' (sequential emulation, "
                "bit-identical, slow) or ties='raster'")


def watershed_edt(d2: DeviceArray, markers: DeviceArray, mask: DeviceArray, seeds_first: bool = True, out=None,
                  connectivity: int = 1, ties: str = "exact", ties_out: DeviceArray | None = None):
    """``skimage.segmentation.watershed(-sqrt(d2), markers, connectivity, mask=mask)`` on the exact integer d2.
    ``seeds_first``: the config-3 recipe (the seeded relief of oracle/skops.py:seeded_flood_image; no ties possible).
    ``ties``: what to do with equal-valued markers inside one component (include/amt_hip.h): 'exact' (default;
    bit-identical to scikit-image through a sequential emulation of its heap for the affected planes), 'raster'
    (fast, documented deviation), 'report' (raster + per-plane flags in ``ties_out``), 'refuse' (raise)."""
    ctx = d2.ctx
    n, H, W = _planes(d2)
    o = _out(ctx, out, d2.shape, np.int32)
    t = _ws_ties(ctx, n, ties, ties_out)
    _hip.check(_lib().amt_watershed_edt_ex(ctx.handle, d2.ptr, markers.ptr, mask.ptr, o.ptr, n, H, W,
                                           1 if seeds_first else 0, int(connectivity), _hip.WS_TIES[ties],
                                           None if t is None else t.ptr),
               "amt_watershed_edt")
    _ws_finish(ctx, ties, t, "watershed_edt")
    return o


def watershed_edt_cleared(d2: DeviceArray, markers: DeviceArray, mask: DeviceArray, nlabels: DeviceArray,
                          max_label: int, scratch: DeviceArray, out=None, count=None, marker_list=None):
    """``relabel_sequential(clear_border(watershed(seeded relief, markers, mask=mask)))`` in one call (the tail of
    config 3; R/masks.py:56,65): identical to ``watershed_edt(seeds_first=True)`` + ``clear_border_relabel`` but the
    watershed image is never written out.  ``markers`` are numbered 1..nlabels[plane]; ``scratch`` is an int32 plane
    batch the flood may use.  Returns (int32 labels, counts)."""
    ctx = d2.ctx
    n, H, W = _planes(d2)
    o = _out(ctx, out, d2.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    if scratch.dtype != np.int32 or scratch.size != d2.size or scratch.ctx is not ctx:
        raise ValueError("scratch must be an int32 array of the batch's size on the same context")
    if marker_list is not None:
        # (list, counts) as label_sparse(keep=) leaves them: every non-zero pixel of `markers`; the component statistics
        # then skip the marker plane
        klist, kcount = marker_list
        if klist.dtype != np.int32 or kcount.dtype != np.int32 or kcount.size != n or klist.size % n:
            raise ValueError("marker_list must be (int32 (n, capacity), int32 (n,))")
        _hip.check(_lib().amt_watershed_edt_cleared_sparse(ctx.handle, d2.ptr, markers.ptr, mask.ptr, scratch.ptr, o.ptr,
                                                           c.ptr, n, H, W, int(max_label), nlabels.ptr, klist.ptr,
                                                           kcount.ptr, klist.size // n),
                   "amt_watershed_edt_cleared_sparse")
        return o, c
    _hip.check(_lib().amt_watershed_edt_cleared(ctx.handle, d2.ptr, markers.ptr, mask.ptr, scratch.ptr, o.ptr, c.ptr, n, H,
                                                W, int(max_label), nlabels.ptr), "amt_watershed_edt_cleared")
    return o, c


def watershed(relief: DeviceArray, markers: DeviceArray, mask: DeviceArray, out=None, connectivity: int = 1,
              ties: str = "exact", ties_out: DeviceArray | None = None):
    """``skimage.segmentation.watershed(relief, markers, connectivity, mask=mask)`` for float64 relief
    (``ties`` as in ``watershed_edt``; connectivity 2 always runs the sequential emulation)."""
    ctx = relief.ctx
    n, H, W = _planes(relief)
    o = _out(ctx, out, relief.shape, np.int32)
    t = _ws_ties(ctx, n, ties, ties_out)
    _hip.check(_lib().amt_watershed_f64_ex(ctx.handle, relief.ptr, markers.ptr, mask.ptr, o.ptr, n, H, W,
                                           int(connectivity), _hip.WS_TIES[ties], None if t is None else t.ptr),
               "amt_watershed_f64")
    _ws_finish(ctx, ties, t, "watershed")
    return o


# --------------------------------------------------------------------------------------------------
# region properties
# --------------------------------------------------------------------------------------------------
def regionprops(labels: DeviceArray, max_label: int, out=None) -> DeviceArray:
    """Morphology table (nplanes, max_label, RP_NCOLS) float64; column order ``_hip.RP_COLS``."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    o = _out(ctx, out, (n, max_label, _hip.RP_NCOLS), np.float64)
    _hip.check(_lib().amt_regionprops(ctx.handle, labels.ptr, o.ptr, n, H, W, int(max_label)), "amt_regionprops")
    return o


def regionprops_full(labels: DeviceArray, intensity: DeviceArray, max_label: int, out=None, iout=None):
    """Morphology table + intensity table in one call (shared bounding-box pass and per-label scan)."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    if intensity.dtype != np.uint16:
        raise TypeError("intensity images must be uint16 on the device path")
    if intensity.ndim < 3 or intensity.shape[-2:] != labels.shape[-2:]:
        raise ValueError("intensity must be (..., C, Y, X) matching the label planes")
    C = int(intensity.shape[-3])
    if intensity.size != n * C * H * W:
        raise ValueError("intensity / labels plane count mismatch")
    o = _out(ctx, out, (n, max_label, _hip.RP_NCOLS), np.float64)
    io = _out(ctx, iout, (n, max_label, C, 4), np.float64)
    _hip.check(_lib().amt_regionprops_full_u16(ctx.handle, labels.ptr, intensity.ptr, C, o.ptr, io.ptr, n, H, W,
                                               int(max_label)), "amt_regionprops_full_u16")
    return o, io


def pack_plate_rows(table: DeviceArray, itable, ncells: DeviceArray, fov_index0: int = 0, fov_index=None,
                    out=None, nrows_out=None):
    """Dense per-FOV tables (B, K, RP_NCOLS) [+ (B, K, C, 4)] + cell counts (B,) -> one row per cell,
    ``[fov index, label, morphology columns, C x {mean, max, min, std}]``: the block a rank contributes to the
    plate's all-gather.  Returns (rows (B*K, 16 + 4C) float64 of which the first ``nrows`` are
    written, nrows (1,) int64; -1 flags a field of view whose table overflowed)."""
    ctx = table.ctx
    if table.ndim != 3 or table.shape[2] != _hip.RP_NCOLS or table.dtype != np.float64:
        raise ValueError(f"table must be (B, K, {_hip.RP_NCOLS}) float64, got {table.shape} {table.dtype}")
    B, K = int(table.shape[0]), int(table.shape[1])
    C = 0
    if itable is not None:
        if itable.ndim != 4 or itable.shape[:2] != (B, K) or itable.shape[3] != 4 or itable.dtype != np.float64:
            raise ValueError(f"itable must be (B, K, C, 4) float64, got {itable.shape} {itable.dtype}")
        C = int(itable.shape[2])
    if ncells.dtype != np.int32 or ncells.size != B:
        raise ValueError("ncells must be (B,) int32")
    if fov_index is not None and (fov_index.dtype != np.int32 or fov_index.size != B):
        raise ValueError("fov_index must be (B,) int32")
    ncols = 2 + _hip.RP_NCOLS + 4 * C
    if out is None:
        out = ctx.empty((max(B * K, 1), ncols), np.float64)
    elif out.dtype != np.float64 or out.ndim != 2 or out.shape[1] != ncols or out.shape[0] < B * K:
        raise ValueError(f"out must be (>= {B * K}, {ncols}) float64, got {out.shape} {out.dtype}")
    nr = _out(ctx, nrows_out, (1,), np.int64)
    _hip.check(_lib().amt_pack_plate_rows(ctx.handle, table.ptr, None if itable is None else itable.ptr, ncells.ptr,
                                          B, K, C, None if fov_index is None else fov_index.ptr, int(fov_index0),
                                          out.ptr, int(out.shape[0]), nr.ptr), "amt_pack_plate_rows")
    return out, nr


def regionprops_intensity(labels: DeviceArray, intensity: DeviceArray, max_label: int, out=None) -> DeviceArray:
    """Intensity table (nplanes, max_label, C, 4) = {mean, max, min, std}; ``intensity`` is (..., C, Y, X)
    uint16 or float64 with one (C, Y, X) stack per label plane."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    if intensity.dtype not in (np.uint16, np.float64):
        raise TypeError("intensity images must be uint16 or float64 on the device path")
    if intensity.ndim < 3 or intensity.shape[-2:] != labels.shape[-2:]:
        raise ValueError("intensity must be (..., C, Y, X) matching the label planes")
    C = int(intensity.shape[-3])
    if intensity.size != n * C * H * W:
        raise ValueError("intensity / labels plane count mismatch")
    o = _out(ctx, out, (n, max_label, C, 4), np.float64)
    if intensity.dtype == np.uint16:  # exact integer accumulation
        _hip.check(_lib().amt_regionprops_intensity_u16(ctx.handle, labels.ptr, intensity.ptr, C, o.ptr, n, H, W,
                                                        int(max_label)), "amt_regionprops_intensity_u16")
    else:  # float64 images: two-sweep mean / std as numpy computes them
        _hip.check(_lib().amt_regionprops_intensity_f64(ctx.handle, labels.ptr, intensity.ptr, C, o.ptr, n, H, W,
                                                        int(max_label)), "amt_regionprops_intensity_f64")
    return o


def cellpose_masks(dP: DeviceArray, cellprob: DeviceArray, cellprob_threshold: float = 0.0, niter: int = 200,
                   min_size: int = 15, max_size_fraction: float = 0.4, max_seeds: int = 16384, out=None, count=None,
                   flow_threshold: float = 0.0, fill_holes: bool = False):
    """Cellpose's flow -> mask post-processing on the device (include/amt_hip.h ``amt_cellpose_masks_ex``; parity
    unpinned): ``dP`` (..., 2, Y, X) float32 flows (dY, dX), ``cellprob`` (..., Y, X) float32 -> (int32 labels, counts).
    ``flow_threshold`` > 0 adds the flow-error filter (``remove_bad_flow_masks``), ``fill_holes`` the hole filling of
    ``fill_holes_and_remove_small_masks``; with neither, the rounds-1/2 entry point.  A plane that produces more than
    ``max_seeds`` seeds reports count -1."""
    ctx = dP.ctx
    if dP.dtype != np.float32 or cellprob.dtype != np.float32:
        raise TypeError("cellpose_masks expects float32 flows and cell probabilities")
    n, H, W = _planes(cellprob)
    if dP.ndim < 3 or dP.shape[-3] != 2 or dP.shape[-2:] != cellprob.shape[-2:] or dP.size != 2 * cellprob.size:
        raise ValueError(f"dP must be (..., 2, Y, X) matching cellprob (..., Y, X); got {dP.shape} and {cellprob.shape}")
    if flow_threshold is None:
        flow_threshold = 0.0
    if flow_threshold < 0:
        raise ValueError(f"flow_threshold must be non-negative, got {flow_threshold}")
    o = _out(ctx, out, cellprob.shape, np.int32)
    c = _out(ctx, count, (n,), np.int32)
    _hip.check(_lib().amt_cellpose_masks_ex(ctx.handle, dP.ptr, cellprob.ptr, o.ptr, c.ptr, n, H, W,
                                            float(cellprob_threshold), int(niter), int(min_size),
                                            float(max_size_fraction), int(max_seeds), float(flow_threshold),
                                            1 if fill_holes else 0), "amt_cellpose_masks_ex")
    return o, c


def cellpose_flow_error(labels: DeviceArray, dP: DeviceArray, max_label: int, nlabels: DeviceArray | None = None):
    """``cellpose.metrics.flow_error`` on the device: (..., Y, X) int32 labels carrying 1..K and (..., 2, Y, X) float32
    network flows -> (nplanes, max_label) float64 errors (absent labels 0).  Parity unpinned."""
    ctx = labels.ctx
    if labels.dtype != np.int32 or dP.dtype != np.float32:
        raise TypeError("cellpose_flow_error expects int32 labels and float32 flows")
    n, H, W = _planes(labels)
    if dP.ndim < 3 or dP.shape[-3] != 2 or dP.shape[-2:] != labels.shape[-2:] or dP.size != 2 * labels.size:
        raise ValueError(f"dP must be (..., 2, Y, X) matching labels (..., Y, X); got {dP.shape} and {labels.shape}")
    if nlabels is None:
        nlabels = ctx.asarray(np.full((n,), int(max_label), np.int32))
    err = ctx.empty((n, int(max_label)), np.float64)
    _hip.check(_lib().amt_cellpose_flow_error(ctx.handle, labels.ptr, dP.ptr, nlabels.ptr, err.ptr, n, H, W,
                                              int(max_label)), "amt_cellpose_flow_error")
    return err


def fill_holes_remove_small(labels: DeviceArray, max_label: int, min_size: int = 15, fill_holes: bool = True,
                            nlabels: DeviceArray | None = None):
    """``cellpose.utils.fill_holes_and_remove_small_masks`` on (..., Y, X) int32 labels carrying 1..max_label ->
    (new int32 labels, counts).  Parity unpinned (oracle/cellpose_dynamics.py)."""
    ctx = labels.ctx
    if labels.dtype != np.int32:
        raise TypeError("fill_holes_remove_small expects int32 labels")
    n, H, W = _planes(labels)
    out = labels.copy()
    if nlabels is None:
        nlabels = ctx.asarray(np.full((n,), int(max_label), np.int32))
    c = ctx.empty((n,), np.int32)
    _hip.check(_lib().amt_fill_holes_remove_small(ctx.handle, out.ptr, nlabels.ptr, c.ptr, n, H, W, int(max_label),
                                                  int(min_size), 1 if fill_holes else 0), "amt_fill_holes_remove_small")
    return out, c


def label_bboxes(labels: DeviceArray, max_label: int) -> np.ndarray:
    """(nplanes, max_label, 4) int32 {min row, min col, max row, max col}, inclusive; absent labels have max < min."""
    ctx = labels.ctx
    n, H, W = _planes(labels)
    if labels.dtype != np.int32:
        raise TypeError("label_bboxes expects int32 labels")
    o = ctx.empty((n, int(max_label), 4), np.int32)
    _hip.check(_lib().amt_label_bboxes(ctx.handle, labels.ptr, o.ptr, n, H, W, int(max_label)), "amt_label_bboxes")
    return o.numpy()


def cell_outlines(labels: DeviceArray, max_label: int) -> list[np.ndarray]:
    """Outline of every label present in ONE int32 label plane, as ``_extract_outlines_skimage`` returns them
    (R/masks.py:82-115): marching squares at level 0.5 on the bounding box padded by one pixel, longest contour,
    (row, col) float64 vertices in image coordinates, ascending label order, empty (0, 2) arrays where no
    contour exists.  The walks run on the device; the host only sizes the scratch and output buffers."""
    ctx = labels.ctx
    if labels.ndim != 2 or labels.dtype != np.int32:
        raise TypeError("cell_outlines expects one (Y, X) int32 label plane")
    H, W = labels.shape
    bb = label_bboxes(labels, max(int(max_label), 1))[0]
    present = np.nonzero(bb[:, 2] >= bb[:, 0])[0]
    nlab = len(present)
    if nlab == 0:
        return []
    boxes = np.empty((nlab, 5), dtype=np.int32)
    boxes[:, 0] = present + 1
    boxes[:, 1] = np.maximum(bb[present, 0] - 1, 0)
    boxes[:, 2] = np.maximum(bb[present, 1] - 1, 0)
    boxes[:, 3] = np.minimum(bb[present, 2] + 2, H)
    boxes[:, 4] = np.minimum(bb[present, 3] + 2, W)
    squares = np.maximum(boxes[:, 3] - boxes[:, 1] - 1, 0).astype(np.int64) * np.maximum(
        boxes[:, 4] - boxes[:, 2] - 1, 0)
    voff = np.concatenate([[0], np.cumsum(squares)]).astype(np.int64)
    d_boxes, d_voff = ctx.asarray(boxes), ctx.asarray(voff)
    nvis = int(voff[-1])
    d_vis = ctx.empty((max(nvis, 1),), np.uint8)
    d_info = ctx.empty((nlab, 4), np.int32)
    _hip.check(_lib().amt_contours_find(ctx.handle, labels.ptr, H, W, nlab, d_boxes.ptr, d_voff.ptr, d_vis.ptr, nvis,
                                        d_info.ptr), "amt_contours_find")
    info = d_info.numpy()
    poff = np.concatenate([[0], np.cumsum(info[:, 0].astype(np.int64))]).astype(np.int64)
    total = int(poff[-1])
    if total == 0:
        return [np.array([]).reshape(0, 2) for _ in range(nlab)]
    d_poff = ctx.asarray(poff)
    d_pts = ctx.empty((total, 2), np.float64)
    _hip.check(_lib().amt_contours_emit(ctx.handle, labels.ptr, H, W, nlab, d_boxes.ptr, d_info.ptr, d_poff.ptr,
                                        d_pts.ptr), "amt_contours_emit")
    pts = d_pts.numpy()
    return [pts[poff[i]:poff[i + 1]].copy() if poff[i + 1] > poff[i] else np.array([]).reshape(0, 2)
            for i in range(nlab)]


def cell_outlines_borders(labels: DeviceArray, max_label: int, skip_first_value: bool = True) -> list[np.ndarray]:
    """Outline of every label of ONE int32 label plane as the reference's default "cellpose" extractor returns
    them (R/masks.py:68-79): ``cellpose.utils.outlines_list`` = OpenCV ``findContours(masks == n, RETR_EXTERNAL,
    CHAIN_APPROX_NONE)`` per label, contour with the most points, (y, x) int64 pixel coordinates after the
    reference's column swap, ``np.zeros((0, 2))`` when the border has fewer than five points.  ``outlines_list``
    iterates ``np.unique(masks)[1:]``, i.e. it drops the smallest value present whatever it is; with
    ``skip_first_value`` the same happens here (label 1 is dropped when the plane has no background pixel).
    Border following runs on the device.  Parity unpinned (no OpenCV offline), see oracle/contours.py."""
    ctx = labels.ctx
    if labels.ndim != 2 or labels.dtype != np.int32:
        raise TypeError("cell_outlines_borders expects one (Y, X) int32 label plane")
    H, W = labels.shape
    bb = label_bboxes(labels, max(int(max_label), 1))[0]
    present = np.nonzero(bb[:, 2] >= bb[:, 0])[0]
    if skip_first_value and len(present):
        area = (bb[present, 2] - bb[present, 0] + 1).astype(np.int64) * (bb[present, 3] - bb[present, 1] + 1)
        # a background pixel exists unless the labels tile the plane; only then is it worth counting pixels
        if area.sum() >= H * W:
            cols = list(_hip.RP_COLS)
            tab = regionprops(labels, max(int(max_label), 1)).numpy()[0]
            if int(round(tab[:, cols.index("area")].sum())) >= H * W:
                present = present[1:]
    nlab = len(present)
    if nlab == 0:
        return []
    boxes = np.empty((nlab, 5), dtype=np.int32)
    boxes[:, 0] = present + 1
    boxes[:, 1] = bb[present, 0]
    boxes[:, 2] = bb[present, 1]
    boxes[:, 3] = bb[present, 2] + 1
    boxes[:, 4] = bb[present, 3] + 1
    d_boxes = ctx.asarray(boxes)
    d_marks = ctx.empty((H, W), np.int8)
    d_info = ctx.empty((nlab, 3), np.int32)
    _hip.check(_lib().amt_borders_find(ctx.handle, labels.ptr, H, W, nlab, d_boxes.ptr, d_marks.ptr, d_info.ptr),
               "amt_borders_find")
    info = d_info.numpy()
    npts = np.where(info[:, 0] > 4, info[:, 0], 0).astype(np.int64)
    poff = np.concatenate([[0], np.cumsum(npts)]).astype(np.int64)
    total = int(poff[-1])
    if total == 0:
        return [np.zeros((0, 2)) for _ in range(nlab)]
    d_poff = ctx.asarray(poff)
    d_pts = ctx.empty((total, 2), np.int32)
    _hip.check(_lib().amt_borders_emit(ctx.handle, labels.ptr, H, W, nlab, d_boxes.ptr, d_info.ptr, d_poff.ptr,
                                       d_pts.ptr), "amt_borders_emit")
    pts = d_pts.numpy().astype(np.int64)
    return [pts[poff[i]:poff[i + 1]].copy() if poff[i + 1] > poff[i] else np.zeros((0, 2)) for i in range(nlab)]
