"""``MicroscopyImage`` and its metadata containers (reference: R/microscopy.py:17-308).

Same validation, properties, error messages and channel-slicing semantics as the reference.  Two
device-side additions: ``to_device()`` keeps the whole (..., C, Y, X) stack in HBM, where
``get_channel_intensities`` is a pointer offset (the reference returns a numpy view, R/microscopy.py:279-282),
and ``from_nd2_path`` reads the pixel block of uncompressed ND2 files with the built-in minimal reader
(``nd2lite``; the reference delegates to the ``nd2`` package, R/nikon.py:25-43).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from functools import cached_property
from pathlib import Path
from typing import Any

import numpy as np
from numpy import uint16

from .channels import Channel
from .exceptions import MetadataWarning
from .metadata_structures import ChannelMetadata, DimensionFlags
from .pipeline import Pipeline
from .typing import ScalarArray, UInt16Array


@dataclass
class InstrumentMetadata:
    """``sizes`` (e.g. {'C': 4, 'Y': 2048, 'X': 2048}) plus one ``ChannelMetadata`` per channel."""

    sizes: dict[str, int]
    channel_metadata_list: list[ChannelMetadata]

    def __post_init__(self) -> None:
        for key in ("X", "Y"):
            if key not in self.sizes:
                raise ValueError(
                    f"sizes must contain '{key}' dimension, got keys: {list(self.sizes.keys())}"
                )
        expected = self.sizes.get("C", 1)
        actual = len(self.channel_metadata_list)
        if actual != expected:
            raise ValueError(
                f"Number of channel metadata entries ({actual}) does not match "
                f"the channel dimension size ({expected}) in sizes"
            )

    @property
    def channel_axis(self) -> int | None:
        axes = list(self.sizes)
        return axes.index("C") if "C" in axes else None

    @cached_property
    def dimensions(self) -> DimensionFlags:
        flags = DimensionFlags(0)
        for cm in self.channel_metadata_list:
            flags |= cm.dimensions
        if len(self.channel_metadata_list) > 1:
            flags |= DimensionFlags.MULTICHANNEL
        return flags


@dataclass
class Metadata:
    """Instrument metadata plus optional sample metadata."""

    instrument: InstrumentMetadata
    sample: dict[str, Any] | None = None

    def __repr__(self) -> str:
        shown = [f"sizes={self.instrument.sizes}",
                 f"channels={[entry.channel.name for entry in self.instrument.channel_metadata_list]}"]
        if self.sample:
            shown.append(f"sample={self.sample}")
        return "<Metadata " + ", ".join(shown) + ">"


@dataclass
class MicroscopyImage:
    """Image intensities (canonically uint16) with metadata for all channels."""

    intensities: UInt16Array
    metadata: Metadata

    def __post_init__(self) -> None:
        sizes = self.metadata.instrument.sizes
        shape, wanted = self.intensities.shape, tuple(sizes.values())
        if shape != wanted:  # the axes of `sizes` ARE the axes of the array, in order (R/microscopy.py:117-123)
            raise ValueError(f"Intensities shape {shape} does not match metadata sizes {sizes} (expected shape {wanted})")
        dtype = self.intensities.dtype
        if dtype != uint16:  # uint16 is the canonical acquisition dtype (R/microscopy.py:125-131)
            note = f"Expected uint16 intensities, got {dtype}. Some operations may behave unexpectedly."
            warnings.warn(note, MetadataWarning, stacklevel=2)

    def __repr__(self) -> str:
        flat = self.intensities.reshape(-1)
        if flat.size <= 10:
            body = f"intensities={list(flat)}"
        else:
            head = ", ".join(map(str, flat[:3].tolist()))
            tail = ", ".join(map(str, flat[-3:].tolist()))
            body = f"intensities=[{head}, ..., {tail}]"
        names = [channel.name for channel in self.channels]
        return f"<MicroscopyImage sizes={self.sizes}, channels={names}, {body}, dtype={self.intensities.dtype}>"

    # -- constructors -----------------------------------------------------------------------------
    @classmethod
    def from_array(cls, intensities: np.ndarray, channels: list[Channel], axes: str | None = None,
                   sample_metadata: dict[str, Any] | None = None) -> "MicroscopyImage":
        """Build an image from a bare array: ``axes`` names the dimensions (default 'CYX' / 'YX')."""
        if axes is None:
            axes = "YX" if intensities.ndim == 2 else "CYX" if intensities.ndim == 3 else None
        if axes is None or len(axes) != intensities.ndim:
            raise ValueError("axes must name every dimension of intensities (e.g. 'TCYX')")
        sizes = {a: int(s) for a, s in zip(axes, intensities.shape)}
        meta = InstrumentMetadata(sizes, [ChannelMetadata(ch) for ch in channels])
        return cls(intensities, Metadata(meta, sample_metadata))

    @classmethod
    def from_nd2_path(cls, nd2_path: Path, channels: list[Channel] | None = None,
                      sample_metadata: dict[str, Any] | None = None) -> "MicroscopyImage":
        """Read an uncompressed ND2 file (reference: R/microscopy.py:154-176 -> R/nikon.py:25-43)."""
        from .nd2lite import load_nd2

        intensities, instrument = load_nd2(Path(nd2_path), channels)
        return cls(intensities, Metadata(instrument, sample_metadata))

    @classmethod
    def from_lif_path(cls, lif_path: Path, image_name: str, channels: list[Channel] | None = None,
                      sample_metadata: dict[str, Any] | None = None) -> "MicroscopyImage":
        """The reference reads Leica LIF files through the ``liffile`` package (R/microscopy.py:178-203,
        R/leica.py); file loaders other than uncompressed ND2 are outside this package: load the
        pixels with the reference's loader and hand them to ``from_array``."""
        raise NotImplementedError(
            "Leica LIF loading is not part of the MI355X hot path: read the image with the reference's "
            "arcadia_microscopy_tools.leica.load_lif_image and pass the array to MicroscopyImage.from_array"
        )

    # -- properties -------------------------------------------------------------------------------
    @property
    def shape(self) -> tuple[int, ...]:
        return self.intensities.shape

    @property
    def sizes(self) -> dict[str, int]:
        return self.metadata.instrument.sizes

    @property
    def dimensions(self) -> DimensionFlags:
        return self.metadata.instrument.dimensions

    @property
    def channels(self) -> list[Channel]:
        return [cm.channel for cm in self.metadata.instrument.channel_metadata_list]

    @property
    def channel_axis(self) -> int | None:
        return self.metadata.instrument.channel_axis

    @property
    def num_channels(self) -> int:
        return len(self.metadata.instrument.channel_metadata_list)

    @staticmethod
    def _resolve_channel_name(channel: str | Channel) -> str:
        return getattr(channel, "name", channel)

    # -- channel access (R/microscopy.py:241-282) ---------------------------------------------------
    def _channel_index(self, channel: str | Channel) -> int:
        name = self._resolve_channel_name(channel)
        names = [ch.name for ch in self.channels]
        if name not in names:
            raise ValueError(f"Channel '{name}' not found in image. Available channels: {names}")
        return names.index(name)

    def get_channel_intensities(self, channel: str | Channel) -> UInt16Array:
        """All data of one channel (T / Z axes preserved); a numpy VIEW, no copy."""
        index = self._channel_index(channel)
        if self.num_channels == 1:
            return self.intensities
        if self.channel_axis is None:
            raise ValueError("Channel axis not found in metadata")
        slices: list[slice | int] = [slice(None)] * self.intensities.ndim
        slices[self.channel_axis] = index
        return self.intensities[tuple(slices)]

    def apply_pipeline(self, pipeline: Pipeline, channel: str | Channel) -> ScalarArray:
        """``pipeline(self.get_channel_intensities(channel))`` (R/microscopy.py:284-308)."""
        return pipeline(self.get_channel_intensities(channel))

    # -- device residency ---------------------------------------------------------------------------
    def to_device(self, ctx=None):
        """Upload the stack once; returns a ``DeviceImage`` whose channel access is a pointer offset."""
        from .device import get_context

        ctx = ctx or get_context()
        return DeviceImage(ctx.asarray(np.ascontiguousarray(self.intensities)), self)


class DeviceImage:
    """A ``MicroscopyImage`` whose intensities live in HBM."""

    def __init__(self, data, image: MicroscopyImage):
        self.data = data
        self.image = image

    def get_channel_intensities(self, channel: str | Channel):
        """Device view of one channel.  Needs the channel axis to be the leading axis (or the one right
        before (Y, X) with all earlier axes of size 1) so that the channel is contiguous in memory."""
        img = self.image
        index = img._channel_index(channel)
        if img.num_channels == 1:
            return self.data
        axis = img.channel_axis
        if axis is None:
            raise ValueError("Channel axis not found in metadata")
        lead = int(np.prod(self.data.shape[:axis], dtype=np.int64))
        if lead != 1:
            raise NotImplementedError(
                "device channel views need the channel axis first (e.g. (C, Y, X) or (C, Z, Y, X)); "
                "use the host image for (T, C, Y, X) stacks"
            )
        flat = self.data.reshape(self.data.shape[axis:])
        return flat[index]
