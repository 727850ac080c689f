"""``SegmentationMask``: label image -> per-cell feature table, on the GPU (reference: R/masks.py:15-467).

Same constructor, validation, error messages, immutability, property names, derived columns and key
order as the reference.  ``label_image`` (clear_border + label / relabel_sequential) and
``cell_properties`` (two ``regionprops_table`` passes in the reference, one per channel for intensity)
run as HIP kernels; only the (num_cells x columns) table crosses PCIe.
"""
from __future__ import annotations

import warnings
from collections.abc import Mapping
import numpy as np

from .typing import Float64Array, Int64Array, ScalarArray

class cached_property:  # noqa: N801 -- used like functools.cached_property
    """``functools.cached_property`` without its lock.  Up to Python 3.11 the standard descriptor holds ONE lock per
    property for ALL instances, so worker threads that each compute ``cell_properties`` of their OWN mask (one context
    per thread, R/pipeline.py:145-146) run one after the other (measured: 8 threads, 2.5 -> 31 ms per call).  Two
    threads racing on the SAME instance may both compute the value; the results are identical and one wins."""

    def __init__(self, func):
        self.func = func
        self.name = func.__name__
        self.__doc__ = func.__doc__

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        d = obj.__dict__
        if self.name not in d:
            d[self.name] = self.func(obj)
        return d[self.name]


# columns of `cell_properties` when the caller names none (the reference's lists, R/masks.py:15-35)
DEFAULT_CELL_PROPERTY_NAMES = (
    "label centroid volume area area_convex perimeter eccentricity circularity solidity "
    "axis_major_length axis_minor_length orientation"
).split()
DEFAULT_INTENSITY_PROPERTY_NAMES = ["intensity_" + stat for stat in ("mean", "max", "min", "std")]


def _process_mask_device(mask_image, remove_edge_cells: bool, mx=None):
    """R/masks.py:38-65 on the device -> (int32 DeviceArray of sequential labels, count)."""
    from . import hipops
    from .device import get_context

    ctx = get_context()
    if mask_image.dtype == bool:
        lab, cnt = hipops.label(ctx.asarray(mask_image), connectivity=2)
        k = int(cnt.numpy()[0])
        if remove_edge_cells:
            # components of a label image are connected by construction: flag-and-renumber pass
            lab, cnt = hipops.clear_border_relabel(lab, max(k, 1))
            k = int(cnt.numpy()[0])
            if k == 0:
                raise ValueError(
                    "No cells remain after removing edge cells. Try setting remove_edge_cells=False."
                )
        return lab, k
    mx = int(mask_image.max() if mx is None else mx)
    if mx >= 2**31 - 1:
        raise ValueError("label values above 2**31 - 2 are not supported on the device path")
    lab = ctx.asarray(mask_image, dtype=np.int32)  # narrowed while it moves into the staging buffer
    if remove_edge_cells:
        # arbitrary label images: a label may have several pieces, only those touching the frame go
        lab = hipops.clear_border(lab)
        if int(hipops.max_per_plane(lab).numpy()[0]) == 0:
            raise ValueError(
                "No cells remain after removing edge cells. Try setting remove_edge_cells=False."
            )
    lab, cnt = hipops.relabel_sequential(lab, mx)
    return lab, int(cnt.numpy()[0])


def _process_mask(mask_image, remove_edge_cells: bool):
    """The reference's module-level helper (R/masks.py:38-65): host array in, int64 label image out."""
    return _process_mask_device(np.asarray(mask_image), remove_edge_cells)[0].numpy_int64()


def _sequential_labels_on_device(label_image):
    """Arbitrary positive labels -> (int32 device plane numbered 1..k in ascending order of the labels, k)."""
    from . import hipops
    from .device import get_context

    a = np.asarray(label_image)
    if a.ndim != 2:
        raise ValueError("label_image must be a 2D array")
    mx = int(a.max()) if a.size else 0
    if mx <= 0:
        return None, 0
    if mx >= 2**31 - 1:
        raise ValueError("label values above 2**31 - 2 are not supported on the device path")
    lab, cnt = hipops.relabel_sequential(get_context().asarray(a, dtype=np.int32), mx)
    return lab, int(cnt.numpy()[0])


def _extract_outlines_skimage(label_image):
    """The reference's module-level helper (R/masks.py:82-115): one (n, 2) float64 (y, x) outline per label present
    in ``label_image``, in ascending label order -- marching squares on the device (``hipops.cell_outlines``)."""
    from . import hipops

    lab, k = _sequential_labels_on_device(label_image)
    return hipops.cell_outlines(lab, k) if k else []


def _extract_outlines_cellpose(label_image):
    """The reference's module-level helper (R/masks.py:68-79): ``cellpose.utils.outlines_list`` per label, (y, x)
    order -- border following on the device (``hipops.cell_outlines_borders``; parity unpinned)."""
    from . import hipops

    lab, k = _sequential_labels_on_device(label_image)
    return hipops.cell_outlines_borders(lab, k) if k else []


def _extrema(a: np.ndarray):
    """(min, max) of a 2-D array.  Planes of a megapixel and more are reduced band by band on the host-copy threads:
    integer images with one foreign call per band (``device.host_extrema``), other element types with numpy in pieces
    of 256 KB whose minimum AND maximum are taken while the piece is in cache.  The image crosses the memory bus once
    (a 2048^2 int64 label image is 33 MB, and with a worker thread per context the host's memory system and the
    interpreter lock are what bound the reference-level calls)."""
    from .device import _pool4, host_extrema

    if a.size < (1 << 20) or a.shape[0] < 8:
        return host_extrema(a) or (a.min(), a.max())
    rows = max(1, (256 << 10) // max(1, a.shape[1] * a.itemsize))

    def band(b):
        got = host_extrema(b)
        if got is not None:
            return got
        mn, mx = b[:1].min(), b[:1].max()
        for r in range(0, b.shape[0], rows):
            piece = b[r:r + rows]
            mn, mx = min(mn, piece.min()), max(mx, piece.max())
        return mn, mx

    step = -(-a.shape[0] // 4)
    futs = [_pool4().submit(band, a[r:r + step]) for r in range(0, a.shape[0], step)]
    parts = [f.result() for f in futs]
    return min(p[0] for p in parts), max(p[1] for p in parts)


def _as_one_block(planes):
    """The (C, Y, X) array whose channels ``planes`` are, when they lie back to back in memory (the usual case:
    ``{channel: fov[i]}`` of one acquisition); None otherwise."""
    p0 = planes[0]
    if len(planes) < 2 or any(p.dtype != p0.dtype or p.shape != p0.shape or not p.flags["C_CONTIGUOUS"] for p in planes):
        return None
    if any(p.ctypes.data != p0.ctypes.data + c * p0.nbytes for c, p in enumerate(planes)):
        return None
    base = p0.base
    while base is not None and not (isinstance(base, np.ndarray) and base.flags["C_CONTIGUOUS"]
                                    and base.ctypes.data <= p0.ctypes.data
                                    and base.ctypes.data + base.nbytes >= planes[-1].ctypes.data + p0.nbytes):
        base = getattr(base, "base", None)
    if base is None or any(p.base is None for p in planes):
        return None
    # a view of the owner that covers exactly the planes (no copy): the owner keeps the memory alive
    flat = base.reshape(-1).view(np.uint8)
    o = p0.ctypes.data - base.ctypes.data
    return flat[o:o + len(planes) * p0.nbytes].view(p0.dtype).reshape((len(planes),) + p0.shape)


def _check_label_plane(mask_image):
    """Constructor checks of the mask, in the reference's order and wording (R/masks.py:169-176) -> the image's
    largest value (kept for the device path's range check)."""
    if not isinstance(mask_image, np.ndarray):
        raise TypeError("mask_image must be a numpy array")
    if mask_image.ndim != 2:
        raise ValueError("mask_image must be a 2D array")
    # one pass for both extrema and no temporaries (the image is 33 MB at 2048^2 int64)
    mn, mx = _extrema(mask_image) if mask_image.size else (0, 0)
    if mask_image.dtype.kind not in "bu" and mn < 0:
        raise ValueError("mask_image must have non-negative values")
    if not mx:
        raise ValueError("mask_image contains no cells (all values are 0)")
    return mx


def _checked_intensities(images, shape):
    """None, or a NEW dict channel -> 2-D array of the mask's shape; the arrays themselves stay shared with the
    caller (R/masks.py:178-194)."""
    if images is None:
        return None
    if not isinstance(images, Mapping):
        raise TypeError("intensity_image_dict must be a Mapping of channels to 2D arrays")
    for channel, plane in images.items():
        label = f"Intensity image for '{channel.name}'"
        if not isinstance(plane, np.ndarray):
            raise TypeError(f"{label} must be a numpy array")
        if plane.ndim != 2:
            raise ValueError(f"{label} must be 2D")
        if plane.shape != shape:
            raise ValueError(f"{label} must have same shape as mask_image")
    return dict(images)


class SegmentationMask:
    """A label (or boolean) mask with the intensity images that belong to it, and the per-cell measurements derived
    from them (constructor contract: R/masks.py:118-208).

    mask_image                bool (foreground) or integer (labels) 2-D array with at least one cell
    intensity_image_dict      optional mapping Channel -> 2-D intensity image of the same shape
    remove_edge_cells         drop cells that touch the image frame before numbering (default True)
    outline_extractor         "cellpose" (pixel borders, the default) or "skimage" (sub-pixel contours)
    property_names            columns of `cell_properties` (default DEFAULT_CELL_PROPERTY_NAMES)
    intensity_property_names  per-channel intensity columns (default: all four when intensity images are given)

    The six constructor arguments cannot be reassigned afterwards; everything derived is computed on the GPU on
    first access and cached.
    """

    _CTOR_FIELDS = ("mask_image", "intensity_image_dict", "remove_edge_cells", "outline_extractor", "property_names",
                    "intensity_property_names")

    def __init__(self, mask_image, intensity_image_dict=None, remove_edge_cells=True, outline_extractor="cellpose",
                 property_names=None, intensity_property_names=None):
        mx = _check_label_plane(mask_image)
        channels = _checked_intensities(intensity_image_dict, mask_image.shape)
        if property_names is None:
            property_names = list(DEFAULT_CELL_PROPERTY_NAMES)
        if intensity_property_names is None:
            intensity_property_names = list(DEFAULT_INTENSITY_PROPERTY_NAMES) if channels else []
        given = (mask_image, channels, remove_edge_cells, outline_extractor, property_names, intensity_property_names)
        for name, value in zip(self._CTOR_FIELDS, given):
            object.__setattr__(self, name, value)
        object.__setattr__(self, "_mask_max", mx)
        object.__setattr__(self, "_sealed", True)

    @classmethod
    def _from_device(cls, labels, num_cells: int, morph: np.ndarray, inten, intensity_image_dict=None,
                     outline_extractor="cellpose", property_names=None, intensity_property_names=None):
        """A mask whose label plane was COMPUTED on the device (``SegmentationModel.batch_masks``): ``labels`` is the
        int32 plane after edge-cell removal and sequential numbering, ``morph`` / ``inten`` the feature rows of its
        ``num_cells`` cells.  Equal in every derived attribute to ``SegmentationMask(labels.numpy_int64(), ...)``;
        ``mask_image`` is that same label image, downloaded when it is first read."""
        if num_cells <= 0:
            raise ValueError("mask_image contains no cells (all values are 0)")
        self = object.__new__(cls)
        channels = _checked_intensities(intensity_image_dict, tuple(labels.shape[-2:]))
        if property_names is None:
            property_names = list(DEFAULT_CELL_PROPERTY_NAMES)
        if intensity_property_names is None:
            intensity_property_names = list(DEFAULT_INTENSITY_PROPERTY_NAMES) if channels else []
        given = (channels, True, outline_extractor, property_names, intensity_property_names)
        for name, value in zip(cls._CTOR_FIELDS[1:], given):
            object.__setattr__(self, name, value)
        d = self.__dict__
        d["_mask_max"] = int(num_cells)
        d["_labels_device"] = (labels, int(num_cells))
        d["_rows"] = (morph, inten)
        d["_sealed"] = True
        return self

    def __getattr__(self, name):
        # only reached for attributes that are not set: the lazily downloaded label image of a device-born mask
        if name == "mask_image" and "_rows" in self.__dict__:
            image = self.__dict__["_labels_device"][0].numpy_int64()
            self.__dict__["mask_image"] = image
            return image
        raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in self._CTOR_FIELDS and self.__dict__.get("_sealed", False):
            raise AttributeError(
                f"Cannot modify '{name}' after SegmentationMask is initialized. Create a new instance instead."
            )
        object.__setattr__(self, name, value)

    def __repr__(self):
        chans = [c.name for c in self.intensity_image_dict] if self.intensity_image_dict else []
        if "mask_image" not in self.__dict__:  # device-born: do not download the image for a repr
            shape = tuple(self.__dict__["_labels_device"][0].shape[-2:])
            return (f"SegmentationMask(shape={shape}, dtype=int64, channels={chans}, "
                    f"remove_edge_cells={self.remove_edge_cells}, outline_extractor={self.outline_extractor!r})")
        return (f"SegmentationMask(shape={self.mask_image.shape}, dtype={self.mask_image.dtype}, channels={chans}, "
                f"remove_edge_cells={self.remove_edge_cells}, outline_extractor={self.outline_extractor!r})")

    # ---------------------------------------------------------------------------------------------
    @cached_property
    def _labels_device(self):
        return _process_mask_device(self.mask_image, self.remove_edge_cells, self._mask_max)

    @cached_property
    def label_image(self) -> Int64Array:
        """Consecutive int64 labels 1..K, background 0, edge cells removed if requested (R/masks.py:210-218)."""
        lab, _ = self._labels_device
        return lab.numpy_int64()

    @cached_property
    def num_cells(self) -> int:
        return int(self._labels_device[1])

    @cached_property
    def cell_outlines(self) -> list[Float64Array]:
        """Per-cell outlines, one (n, 2) array of (y, x) vertices per cell (R/masks.py:229-245).

        Both extractors run on the device.  ``outline_extractor="skimage"``: marching squares at level 0.5 on each
        cell's padded bounding box, longest contour, in scikit-image's vertex order, float64 (R/masks.py:82-115;
        pinned by tests/golden/outlines_96.npz).  ``outline_extractor="cellpose"`` (the reference's default):
        ``cellpose.utils.outlines_list`` = OpenCV ``findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE)`` per cell, the
        border with the most points, int64 pixel coordinates, empty when it has fewer than five points
        (R/masks.py:68-79); restated from the published border-following algorithm, parity unpinned (neither
        package exists offline, the reference's tests hold no vector for it)."""
        from . import hipops

        lab, k = self._labels_device
        plane = lab[0] if lab.ndim == 3 else lab
        if self.outline_extractor == "cellpose":
            return hipops.cell_outlines_borders(plane, int(k))
        return hipops.cell_outlines(plane, int(k))

    @cached_property
    def cell_properties(self) -> dict[str, ScalarArray]:
        """Morphology + per-channel intensity features, one entry per cell ordered by label
        (R/masks.py:247-328; columns of scikit-image's ``regionprops_table``)."""
        from . import hipops
        from .device import get_context
        from .segment import assemble_cell_properties

        assert self.property_names is not None
        assert self.intensity_property_names is not None
        if "_rows" in self.__dict__:  # measured by the chain that made the labels (SegmentationModel.batch_masks)
            morph, inten = self.__dict__["_rows"]
            names = [c.name for c in self.intensity_image_dict] if self.intensity_image_dict else []
            use = inten if (names and self.intensity_property_names) else None
            return assemble_cell_properties(morph, use, names, list(self.property_names),
                                            list(self.intensity_property_names))
        lab, k = self._labels_device
        k = int(k)
        inten = None
        names: list[str] = []
        if self.intensity_image_dict and self.intensity_property_names:
            planes = []
            for channel, img in self.intensity_image_dict.items():
                planes.append(np.asarray(img))
                names.append(channel.name)
            ctx = get_context()
            # uint8 / uint16 images are accumulated exactly (integer sums); any other dtype the reference accepts
            # (R/masks.py:178-190: "any 2-D ndarray") is measured in float64, as regionprops does.  The planes go to
            # the device one by one into their slot of a (C, Y, X) buffer: no host-side stack
            exact = all(p.dtype in (np.uint8, np.uint16) for p in planes)
            dt = np.uint16 if exact else np.float64
            stack = ctx.empty((len(planes),) + tuple(lab.shape[-2:]), dt)
            whole = _as_one_block(planes)
            if whole is not None:  # the planes are the channels of ONE (C, Y, X) array: one staged transfer
                ctx.asarray(whole, out=stack)
            else:
                for c, p in enumerate(planes):
                    ctx.asarray(p, out=stack[c])  # converted to ``dt`` on its way through the staging buffer
            if exact and lab.size == stack.size // len(planes):
                # morphology + intensities share the bounding-box pass and the per-label scan
                m, it = hipops.regionprops_full(lab.reshape((1,) + tuple(lab.shape[-2:])),
                                                stack.reshape((1,) + stack.shape), max(k, 1))
                morph, inten = m.numpy()[0][:k], it.numpy()[0][:k]
            else:
                morph = hipops.regionprops(lab, max(k, 1)).numpy()[0][:k]
                inten = hipops.regionprops_intensity(lab, stack, max(k, 1)).numpy()[0][:k]
        else:
            morph = hipops.regionprops(lab, max(k, 1)).numpy()[0][:k]
        return assemble_cell_properties(morph, inten, names, list(self.property_names),
                                        list(self.intensity_property_names))

    @cached_property
    def centroids_yx(self) -> Float64Array:
        """(num_cells, 2) array of [y, x] centroids (R/masks.py:330-353)."""
        if self.property_names is None:
            raise ValueError("property_names cannot be None.")
        if "centroid" not in self.property_names:
            message = ("Centroid property not available. Include 'centroid' in property_names to get centroid "
                       "coordinates. Returning empty array.")
            warnings.warn(message, UserWarning, stacklevel=2)
            return np.empty((0, 2), dtype=float)
        table = self.cell_properties
        return np.stack([table["centroid_y"], table["centroid_x"]], axis=1).astype(float)

    def filter(self, property_name: str, min_value: float | None = None,
               max_value: float | None = None) -> "SegmentationMask":
        """New mask keeping the cells with ``min_value <= property <= max_value`` (R/masks.py:355-418)."""
        from . import hipops
        from .device import get_context

        assert self.property_names is not None
        assert self.intensity_property_names is not None
        if min_value is None and max_value is None:
            raise ValueError("At least one of min_value or max_value must be provided.")
        if property_name not in self.cell_properties:
            raise ValueError(
                f"Property '{property_name}' not found. "
                f"Available properties: {list(self.cell_properties.keys())}"
            )
        values = self.cell_properties[property_name]
        keep = np.ones(self.num_cells, dtype=bool)
        if min_value is not None:
            keep &= values >= min_value
        if max_value is not None:
            keep &= values <= max_value
        flags = np.zeros((1, self.num_cells + 1), dtype=np.uint8)
        flags[0, 1:] = keep
        lab, _ = self._labels_device
        kept = hipops.keep_labels(lab, get_context().asarray(flags), self.num_cells)
        new_label_image = kept.numpy_int64()
        if not keep.any():
            raise ValueError(
                f"No cells remain after filtering '{property_name}' "
                f"with min={min_value}, max={max_value}."
            )
        # the survivors keep their pixels; numbering restarts at 1 and edge cells are NOT re-examined
        settings = {name: getattr(self, name) for name in self._CTOR_FIELDS[1:]}
        settings.update(mask_image=new_label_image, remove_edge_cells=False,
                        property_names=list(self.property_names),
                        intensity_property_names=list(self.intensity_property_names))
        return SegmentationMask(**settings)

    def convert_properties_to_microns(self, pixel_size_um: float) -> dict[str, ScalarArray]:
        """Scale lengths / areas / volumes to microns with ``_um`` / ``_um2`` / ``_um3`` key suffixes
        (R/masks.py:420-467); dimensionless and intensity columns pass through."""
        linear = {"perimeter", "axis_major_length", "axis_minor_length"}
        area = {"area", "area_convex"}
        volume = {"volume"}
        tensor = {"inertia_tensor", "inertia_tensor_eigvals"}
        out = {}
        for name, vals in self.cell_properties.items():
            if name in linear:
                out[f"{name}_um"] = vals * pixel_size_um
            elif name in area or name in tensor:
                out[f"{name}_um2"] = vals * (pixel_size_um**2)
            elif name in volume:
                out[f"{name}_um3"] = vals * (pixel_size_um**3)
            else:
                out[name] = vals
        return out
