"""Minimal stand-ins for the reference's metadata records (R/metadata_structures.py:14-141).

The hot path only needs the channel list and the ``sizes`` mapping; the descriptive acquisition
metadata parsed from ND2 / LIF files is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Flag, auto

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


@dataclass
class ChannelMetadata:
    """Per-channel metadata: only the channel identity and its dimension flags are kept here."""

    channel: Channel
    dimensions: DimensionFlags = DimensionFlags(0)
