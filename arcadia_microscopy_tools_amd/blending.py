"""Fluorescence overlays on the device (reference: R/blending.py:14-226).

Same names, arguments, validation, warnings and results as the reference: ``BlendMode``, ``Layer``,
``overlay_channels``, ``create_overlay``.  The colour maps are matplotlib's two-stop ``LinearSegmentedColormap``
tables (256 entries), built here with numpy exactly as matplotlib builds them -- matplotlib itself is not imported --
and all layers are composited by ONE HIP kernel (``amt_overlay``).  Inputs may be numpy arrays (copied to the device,
result returned as numpy) or ``DeviceArray`` planes (result stays on the device).
"""
from __future__ import annotations

import ctypes
import warnings
from dataclasses import dataclass
from enum import Enum
from functools import lru_cache

import numpy as np

from . import _hip
from .channels import Channel
from .device import DeviceArray, get_context

_N_LUT = 256


class BlendMode(Enum):
    """How a foreground layer is composited onto the canvas (R/blending.py:14-30): ALPHA = Porter-Duff "over"
    (order matters), ADDITIVE = contributions accumulate and are clipped (order does not matter)."""

    ALPHA = "alpha"
    ADDITIVE = "additive"


def _min_max(a):
    if isinstance(a, DeviceArray):
        from . import hipops

        mm = hipops.minmax(a).numpy().reshape(-1, 2)
        return float(mm[:, 0].min()), float(mm[:, 1].max())
    return float(a.min()), float(a.max())


@dataclass
class Layer:
    """A single layer of an overlay (R/blending.py:33-71): channel (colour), 2-D intensities in [0, 1], opacity,
    ``zero_transparent`` (transparent grey -> colour, else black -> colour) and blend mode."""

    channel: Channel
    intensities: "np.ndarray | DeviceArray"
    opacity: float = 1.0
    zero_transparent: bool = True
    blend_mode: BlendMode = BlendMode.ALPHA

    def __post_init__(self) -> None:
        if self.intensities.ndim != 2:
            raise ValueError(f"Expected 2D intensities array, got shape {self.intensities.shape}")
        if not 0 <= self.opacity <= 1:
            raise ValueError(f"Opacity must be in [0, 1], got {self.opacity}")
        lo, hi = _min_max(self.intensities)
        if lo < 0.0 or hi > 1.0:
            warnings.warn(
                f"Layer '{self.channel.name}' has intensity values outside [0, 1] "
                f"(min={lo:.4g}, max={hi:.4g}). Values will be clipped, which "
                f"may indicate missing normalization.",
                stacklevel=2,
            )
            if not isinstance(self.intensities, DeviceArray):
                self.intensities = np.clip(self.intensities, 0.0, 1.0)
            # device planes are clipped inside the kernel (same values, no extra pass)


def _hex_to_rgb(color: str):
    c = color.lstrip("#")
    if len(c) == 3:
        c = "".join(ch * 2 for ch in c)
    return tuple(int(c[i:i + 2], 16) / 255 for i in (0, 2, 4))


@lru_cache(maxsize=64)
def _build_lut(color: str, zero_transparent: bool) -> np.ndarray:
    """The (256, 4) table of ``LinearSegmentedColormap.from_list(name, [stop0, color])`` (R/blending.py:200-219):
    matplotlib evaluates ``np.linspace(0, 1, 256)`` and ``distance * (y1 - y0) + y0`` per component."""
    r, g, b = _hex_to_rgb(color)
    stop0 = (0.5, 0.5, 0.5, 0.0) if zero_transparent else (0.0, 0.0, 0.0, 1.0)
    stop1 = (r, g, b, 1.0)
    xind = np.linspace(0, 1, _N_LUT)
    lut = np.empty((_N_LUT, 4), dtype=np.float64)
    for k in range(4):
        y0, y1 = stop0[k], stop1[k]
        distance = (xind[1:-1] - 0.0) / (1.0 - 0.0)
        lut[:, k] = np.clip(np.concatenate([[y0], distance * (y1 - y0) + y0, [y1]]), 0.0, 1.0)
    lut.setflags(write=False)
    return lut


class _LutColormap:
    """What the reference gets from ``LinearSegmentedColormap.from_list`` (R/blending.py:204-221), reduced to what it
    uses: a callable that maps floats in [0, 1] to RGBA rows of the 256-entry table (``Colormap.__call__``: the
    index is ``int(x * 256)``, 1.0 lands in the last entry)."""

    def __init__(self, name: str, table: np.ndarray):
        self.name, self.N, self._table = name, table.shape[0], table

    def __call__(self, x):
        scaled = np.array(x, dtype=np.float64, copy=True) * self.N
        scaled[scaled == self.N] = self.N - 1
        return self._table[np.clip(scaled, 0, self.N - 1).astype(int)]


@lru_cache(maxsize=64)
def _build_colormap(color: str, zero_transparent: bool) -> _LutColormap:
    """The reference's cached two-stop colour map (R/blending.py:204-221) as a host-side callable; the device kernel
    reads the same table (``_build_lut``)."""
    return _LutColormap(f"_chan_{color}", _build_lut(color, zero_transparent))


def _blend_alpha(background, foreground, alpha):
    """Porter-Duff 'over' on host arrays (R/blending.py:186-192); ``amt_overlay`` evaluates the same expression."""
    return np.clip(alpha * foreground + (1 - alpha) * background, 0.0, 1.0)


def _blend_additive(background, foreground, alpha):
    """Additive compositing on host arrays (R/blending.py:195-201)."""
    return np.clip(background + alpha * foreground, 0.0, 1.0)


def _composite(background, foreground, alpha, mode):
    """Dispatch on the blend mode (R/blending.py:174-183)."""
    return (_blend_additive if mode is BlendMode.ADDITIVE else _blend_alpha)(background, foreground, alpha)


def _gray_to_rgb(image):
    """(H, W) -> (H, W, 3) on the host (R/blending.py:224-226)."""
    return np.repeat(np.asarray(image)[:, :, np.newaxis], 3, axis=2)


def overlay_channels(background, channel_intensities: dict, *, opacity: float = 1.0, zero_transparent: bool = True,
                     blend_mode: BlendMode = BlendMode.ALPHA):
    """Overlay with uniform settings for all channels (R/blending.py:74-113)."""
    layers = [Layer(channel, intensities, opacity, zero_transparent, blend_mode)
              for channel, intensities in channel_intensities.items()]
    return create_overlay(background, layers)


def create_overlay(background, layers: list[Layer]):
    """Composite ``layers`` onto ``background`` (R/blending.py:116-171): RGB image (H, W, 3) float64."""
    if background.ndim != 2:
        raise ValueError(f"Expected 2D background array, got shape {background.shape}")
    lo, hi = _min_max(background)
    if lo < 0.0 or hi > 1.0:
        warnings.warn(
            f"Background has values outside [0, 1] (min={lo:.4g}, max={hi:.4g}). "
            f"Values will be clipped, which may indicate missing normalization.",
            stacklevel=2,
        )
    for layer in layers:
        if tuple(layer.intensities.shape) != tuple(background.shape):
            raise ValueError(
                f"Layer '{layer.channel.name}' has shape "
                f"{tuple(layer.intensities.shape)}, but background has shape "
                f"{tuple(background.shape)}."
            )
    on_host = not isinstance(background, DeviceArray)
    ctx = get_context() if on_host else background.ctx

    def dev(a):
        if isinstance(a, DeviceArray):
            if a.dtype != np.float64:
                raise TypeError("device planes of an overlay must be float64")
            return a
        return ctx.asarray(np.ascontiguousarray(a, dtype=np.float64))

    H, W = (int(s) for s in background.shape)
    d_bg = dev(background)
    canvas = None
    # the kernel composites up to 8 layers per call; longer lists continue from the previous canvas' ... which is RGB,
    # so they are refused rather than approximated
    if len(layers) > 8:
        raise NotImplementedError("create_overlay composites at most 8 layers on the device path")
    planes = [dev(l.intensities) for l in layers]
    n = len(layers)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[p.ptr for p in planes])
    luts = np.ascontiguousarray(np.stack([_build_lut(l.channel.color, bool(l.zero_transparent)) for l in layers])
                                if n else np.zeros((1, _N_LUT, 4)))
    opac = np.ascontiguousarray([float(l.opacity) for l in layers] or [0.0], dtype=np.float64)
    mode = np.ascontiguousarray([1 if l.blend_mode is BlendMode.ADDITIVE else 0 for l in layers] or [0], dtype=np.int32)
    canvas = ctx.empty((H, W, 3), np.float64)
    _hip.check(_hip.load_library().amt_overlay(ctx.handle, d_bg.ptr, ptrs, n, luts.ctypes.data, opac.ctypes.data,
                                               mode.ctypes.data, canvas.ptr, H, W), "amt_overlay")
    return canvas.numpy() if on_host else canvas
