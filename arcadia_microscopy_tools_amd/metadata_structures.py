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

    MULTICHANNEL = auto()
    TIMELAPSE = auto()
    Z_STACK = auto()
    SPECTRAL = auto()
    RGB = auto()
    MONTAGE = auto()


@dataclass
class ChannelMetadata:
    """Per-channel metadata: only the channel identity and its dimension flags are kept here."""

    channel: Channel
    dimensions: DimensionFlags = DimensionFlags(0)
