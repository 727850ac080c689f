"""Channel definitions (reference: R/channels.py:35-117).

Only what the hot path needs: the frozen ``Channel`` record (its ``name`` keys the intensity columns of
``SegmentationMask.cell_properties``) and the predefined channels.  ``Channel.from_wavelength`` /
``wavelength_to_hex`` depend on the ``colour`` package (visualisation) and are out of scope.
"""
from __future__ import annotations

import re
from dataclasses import dataclass

_HEX_RE = re.compile(r"^#(?:[0-9a-fA-F]{3}){1,2}$")


@dataclass(frozen=True)
class Channel:
    """A microscopy imaging channel (name, display colour, optional excitation / emission in nm)."""

    name: str
    color: str
    excitation_nm: float | None = None
    emission_nm: float | None = None

    def __post_init__(self) -> None:
        if not _HEX_RE.match(self.color):
            raise ValueError(f"color must be a hex code like '#FF0000', got '{self.color}'")
        if self.excitation_nm is not None and self.excitation_nm <= 0:
            raise ValueError("excitation_nm must be positive")
        if self.emission_nm is not None and self.emission_nm <= 0:
            raise ValueError("emission_nm must be positive")


BRIGHTFIELD = Channel("BRIGHTFIELD", "#FFFFFF")
DIC = Channel("DIC", "#FFFFFF")
PHASE = Channel("PHASE", "#DDDDDD")
DAPI = Channel("DAPI", "#0033FF", excitation_nm=405, emission_nm=450)
FITC = Channel("FITC", "#07FF00", excitation_nm=488, emission_nm=512)
TRITC = Channel("TRITC", "#FFBF00", excitation_nm=561, emission_nm=595)
CY5 = Channel("CY5", "#A30000", excitation_nm=640, emission_nm=665)
SRS = Channel("SRS", "#E63535")
E_CARS = Channel("E-CARS", "#AB1299")
F_CARS = Channel("F-CARS", "#AB1299")
E_SHG = Channel("E-SHG", "#F29B4F")
F_SHG = Channel("F-SHG", "#F29B4F")

CHANNELS: dict[str, Channel] = {
    ch.name: ch for ch in [BRIGHTFIELD, DIC, PHASE, DAPI, FITC, TRITC, CY5, SRS, E_CARS, F_CARS, E_SHG, F_SHG]
}
