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


# name, display colour, excitation / emission maximum in nm (values of R/channels.py:93-117)
_PREDEFINED = (
    ("BRIGHTFIELD", "#FFFFFF", None, None),
    ("DIC", "#FFFFFF", None, None),
    ("PHASE", "#DDDDDD", None, None),
    ("DAPI", "#0033FF", 405, 450),
    ("FITC", "#07FF00", 488, 512),
    ("TRITC", "#FFBF00", 561, 595),
    ("CY5", "#A30000", 640, 665),
    ("SRS", "#E63535", None, None),
    ("E-CARS", "#AB1299", None, None),
    ("F-CARS", "#AB1299", None, None),
    ("E-SHG", "#F29B4F", None, None),
    ("F-SHG", "#F29B4F", None, None),
)
CHANNELS: dict[str, Channel] = {row[0]: Channel(*row) for row in _PREDEFINED}
# module-level names as the reference has them (a dash in a channel name becomes an underscore)
BRIGHTFIELD, DIC, PHASE = CHANNELS["BRIGHTFIELD"], CHANNELS["DIC"], CHANNELS["PHASE"]
DAPI, FITC, TRITC, CY5 = CHANNELS["DAPI"], CHANNELS["FITC"], CHANNELS["TRITC"], CHANNELS["CY5"]
SRS, E_CARS, F_CARS = CHANNELS["SRS"], CHANNELS["E-CARS"], CHANNELS["F-CARS"]
E_SHG, F_SHG = CHANNELS["E-SHG"], CHANNELS["F-SHG"]
