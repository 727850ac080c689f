"""The reference's metadata records (R/metadata_structures.py:14-141) with the same names and fields.

The hot path needs the channel list and the ``sizes`` mapping; the per-channel records below carry what the ND2
reader can fill in from the file itself (pixel size, z step, objective, exposure, zoom, binning: ``nd2lite.py``), so
that ``image.metadata.instrument.channel_metadata_list[i].resolution.xy_step_um`` feeds
``SegmentationMask.convert_properties_to_microns`` as it does with the reference.  Every field beyond ``channel`` is
optional here (the reference's full parser -- timestamps, measured axis values, light sources -- is out of scope,
SURVEY.md 2a #10).
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
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


@dataclass
class NominalDimensions(DimensionValidatorMixin):
    """Nominal sampling of every axis (R/metadata_structures.py:70-95)."""

    x_size_px: int
    y_size_px: int
    xy_step_um: float
    z_size_px: int | None = dimension_field(DimensionFlags.Z_STACK)
    z_step_um: float | None = dimension_field(DimensionFlags.Z_STACK)
    t_size_px: int | None = dimension_field(DimensionFlags.TIMELAPSE)
    t_step_ms: float | None = dimension_field(DimensionFlags.TIMELAPSE)
    w_size_px: int | None = dimension_field(DimensionFlags.SPECTRAL)
    w_step_nm: float | None = dimension_field(DimensionFlags.SPECTRAL)


@dataclass
class MeasuredDimensions(DimensionValidatorMixin):
    """Axis values as recorded during the acquisition (R/metadata_structures.py:98-116); the minimal ND2 reader leaves
    them unset."""

    x_values_um: Any = dimension_field(DimensionFlags.MONTAGE)
    y_values_um: Any = dimension_field(DimensionFlags.MONTAGE)
    z_values_um: Any = dimension_field(DimensionFlags.Z_STACK)
    t_values_ms: Any = dimension_field(DimensionFlags.TIMELAPSE)
    w_values_nm: Any = dimension_field(DimensionFlags.SPECTRAL)


@dataclass
class AcquisitionSettings(DimensionValidatorMixin):
    """Camera / scanner settings of one channel (R/metadata_structures.py:119-140)."""

    exposure_time_s: float | None = None
    zoom: float | None = None
    binning: str | None = None
    pixel_dwell_time_us: float | None = None
    line_scan_speed_hz: float | None = None
    line_averaging: int | None = None
    line_accumulation: int | None = None
    frame_averaging: int | None = None
    frame_accumulation: int | None = None


@dataclass
class MicroscopeConfig:
    """Objective and light source (R/metadata_structures.py:143-158)."""

    magnification: int
    numerical_aperture: float
    objective: str | None = None
    light_source: str | None = None
    power_mw: float | None = None


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
