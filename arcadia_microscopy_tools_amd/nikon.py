"""``nikon.load_nd2`` under the reference's module name (R/nikon.py:25-49): the minimal reader of ``nd2lite.py``
(uncompressed ND2 pixels, channel names, loop axes, pixel size / objective / exposure per channel) -- no ``nd2`` package."""
from __future__ import annotations

from pathlib import Path

from .channels import Channel
from .nd2lite import load_nd2 as _load_nd2
from .nd2lite import resolve_optical_config as _resolve_optical_config  # noqa: F401  (same private name as R/nikon.py:52)


def load_nd2(nd2_path: Path, channels: list[Channel] | None = None):
    """-> (uint16 intensities, InstrumentMetadata), as the reference's loader returns them."""
    return _load_nd2(Path(nd2_path), channels)
