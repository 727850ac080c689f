"""The reference's metadata records (R/metadata_structures.py:14-141) with the same names and fields.

The hot path needs the channel list and the ``sizes`` mapping; the per-channel records below carry what the ND2
reader can fill in from the file itself (pixel size, z step, objective, exposure, zoom, binning: ``nd2lite.py``), so
that ``image.metadata.instrument.channel_metadata_list[i].resolution.xy_step_um`` feeds
``SegmentationMask.convert_properties_to_microns`` as it does with the reference.  Every field beyond ``channel`` is
optional here (the reference's full parser -- timestamps, measured axis values, light sources -- is out of scope,
SURVEY.md 2a #10).
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields, make_dataclass
from datetime import datetime
from enum import Flag, auto
from typing import Any

from .channels import Channel


class DimensionFlags(Flag):
    """Which dimensions an acquisition has (R/metadata_structures.py:34-67)."""

    SPATIAL_2D = 0
    MULTICHANNEL = auto()  # member order (= bit values) as in R/metadata_structures.py:37-43
    Z_STACK = auto()
    TIMELAPSE = auto()
    SPECTRAL = auto()
    RGB = auto()
    MONTAGE = auto()

    def _has(self, flag: "DimensionFlags") -> bool:
        return bool(self & flag)

    # the reference's convenience tests (R/metadata_structures.py:45-67)
    is_multichannel = property(lambda self: self._has(DimensionFlags.MULTICHANNEL))
    is_zstack = property(lambda self: self._has(DimensionFlags.Z_STACK))
    is_timelapse = property(lambda self: self._has(DimensionFlags.TIMELAPSE))
    is_spectral = property(lambda self: self._has(DimensionFlags.SPECTRAL))
    is_rgb = property(lambda self: self._has(DimensionFlags.RGB))
    is_montage = property(lambda self: self._has(DimensionFlags.MONTAGE))


def dimension_field(dimension: DimensionFlags, default: Any = None) -> Any:
    """A dataclass field that must be set when the acquisition has ``dimension`` (R/metadata_structures.py:14-16)."""
    return field(default=default, metadata={"requires_dimension": dimension})


class DimensionValidatorMixin:
    """``validate(dimensions)``: every field tagged for a dimension the acquisition has must not be None
    (R/metadata_structures.py:19-31)."""

    def validate(self, dimensions: DimensionFlags) -> None:
        for info in fields(self):  # type: ignore[arg-type]
            needed = info.metadata.get("requires_dimension")
            if needed and (dimensions & needed) and getattr(self, info.name) is None:
                raise ValueError(f"{info.name} is required for {needed.name}")


def _record(name: str, doc: str, spec, mixin: bool = True):
    """A dataclass from a compact table: (field, type[, dimension that requires it]) rows; fields without a default
    come first and are mandatory, the others default to None (tagged with their dimension where given)."""
    rows = []
    for row in spec:
        fname, ftype = row[0], row[1]
        if len(row) == 2:
            rows.append((fname, ftype))
        elif row[2] is None:
            rows.append((fname, ftype, field(default=None)))
        else:
            rows.append((fname, ftype, dimension_field(row[2])))
    cls = make_dataclass(name, rows, bases=(DimensionValidatorMixin,) if mixin else ())
    cls.__doc__, cls.__module__ = doc, __name__
    return cls


_Z, _T, _W, _M = DimensionFlags.Z_STACK, DimensionFlags.TIMELAPSE, DimensionFlags.SPECTRAL, DimensionFlags.MONTAGE

NominalDimensions = _record(
    "NominalDimensions", "Nominal sampling of every axis (R/metadata_structures.py:70-95).",
    (("x_size_px", int), ("y_size_px", int), ("xy_step_um", float),
     ("z_size_px", "int | None", _Z), ("z_step_um", "float | None", _Z),
     ("t_size_px", "int | None", _T), ("t_step_ms", "float | None", _T),
     ("w_size_px", "int | None", _W), ("w_step_nm", "float | None", _W)))

MeasuredDimensions = _record(
    "MeasuredDimensions", "Axis values as recorded during the acquisition (R/metadata_structures.py:98-116); the minimal "
    "ND2 reader leaves them unset.",
    (("x_values_um", Any, _M), ("y_values_um", Any, _M), ("z_values_um", Any, _Z), ("t_values_ms", Any, _T),
     ("w_values_nm", Any, _W)))

AcquisitionSettings = _record(
    "AcquisitionSettings", "Camera / scanner settings of one channel (R/metadata_structures.py:119-140).",
    tuple((n, t, None) for n, t in (
        ("exposure_time_s", "float | None"), ("zoom", "float | None"), ("binning", "str | None"),
        ("pixel_dwell_time_us", "float | None"), ("line_scan_speed_hz", "float | None"),
        ("line_averaging", "int | None"), ("line_accumulation", "int | None"), ("frame_averaging", "int | None"),
        ("frame_accumulation", "int | None"))))

MicroscopeConfig = _record(
    "MicroscopeConfig", "Objective and light source (R/metadata_structures.py:143-158).",
    (("magnification", int), ("numerical_aperture", float), ("objective", "str | None", None),
     ("light_source", "str | None", None), ("power_mw", "float | None", None)), mixin=False)


@dataclass
class ChannelMetadata:
    """Per-channel metadata (R/metadata_structures.py:161-178; same field order).  Only ``channel`` is mandatory here;
    ``resolution`` / ``acquisition`` / ``optics`` are filled by the ND2 reader, ``timestamp`` / ``measured`` stay None."""

    channel: Channel
    timestamp: datetime | None = None
    dimensions: DimensionFlags = DimensionFlags(0)
    resolution: NominalDimensions | None = None
    measured: MeasuredDimensions | None = None
    acquisition: AcquisitionSettings | None = None
    optics: MicroscopeConfig | None = None

    def __post_init__(self) -> None:
        if self.resolution is not None:
            self.resolution.validate(self.dimensions)
