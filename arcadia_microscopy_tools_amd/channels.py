"""Channel definitions (reference: R/channels.py:35-117).

The frozen ``Channel`` record (its ``name`` keys the intensity columns of ``SegmentationMask.cell_properties``), the
predefined channels, and ``wavelength_to_hex`` / ``Channel.from_wavelength`` (R/channels.py:13-91).  The reference
takes the colour of a wavelength from the ``colour`` package (tabulated CIE 1931 colour-matching functions); that
package is not a dependency here, so the functions below use the published analytic fit of the same functions
(Wyman, Sloan & Shirley 2013, multi-lobe Gaussians) followed by the standard XYZ -> sRGB matrix and transfer curve.
PARITY UNPINNED for the hex digits: same interface, range check and error text, colours equal to within the fit
(a few counts per component); nothing on the hot path depends on them.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass

_HEX_RE = re.compile(r"^#(?:[0-9a-fA-F]{3}){1,2}$")

# CIE 1931 2-degree observer as sums of piecewise Gaussians: (weight, centre nm, sigma below, sigma above)
_CMF_LOBES = {
    "x": ((1.056, 599.8, 37.9, 31.0), (0.362, 442.0, 16.0, 26.7), (-0.065, 501.1, 20.4, 26.2)),
    "y": ((0.821, 568.8, 46.9, 40.5), (0.286, 530.9, 16.3, 31.1)),
    "z": ((1.217, 437.0, 11.8, 36.0), (0.681, 459.0, 26.0, 13.8)),
}
_XYZ_TO_LINEAR_SRGB = ((3.2406, -1.5372, -0.4986), (-0.9689, 1.8758, 0.0415), (0.0557, -0.2040, 1.0570))


def _tristimulus(wavelength_nm: float, lobes) -> float:
    total = 0.0
    for weight, centre, below, above in lobes:
        t = (wavelength_nm - centre) / (below if wavelength_nm < centre else above)
        total += weight * math.exp(-0.5 * t * t)
    return total


def wavelength_to_hex(wavelength_nm: float) -> str:
    """Display colour "#RRGGBB" of a visible wavelength, 360-780 nm (R/channels.py:13-33)."""
    if not 360 <= wavelength_nm <= 780:
        raise ValueError(
            f"Wavelength must be in the visible range (360-780 nm), got {wavelength_nm} nm"
        )
    xyz = [_tristimulus(wavelength_nm, _CMF_LOBES[k]) for k in "xyz"]
    digits = []
    for row in _XYZ_TO_LINEAR_SRGB:
        linear = sum(m * v for m, v in zip(row, xyz))
        linear = min(max(linear, 0.0), 1.0)
        encoded = 12.92 * linear if linear <= 0.0031308 else 1.055 * linear ** (1.0 / 2.4) - 0.055
        digits.append(int(min(max(encoded, 0.0), 1.0) * 255))
    return "#{:02X}{:02X}{:02X}".format(*digits)


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

    @classmethod
    def from_wavelength(cls, wavelength_nm: float, *, name: str | None = None, is_excitation: bool = True) -> "Channel":
        """A channel coloured after a visible wavelength (R/channels.py:59-88): named "<wavelength>nm" unless ``name``
        is given; the wavelength, rounded to 0.1 nm, is stored as excitation (default) or emission."""
        color = wavelength_to_hex(wavelength_nm)
        stored = round(wavelength_nm, 1)
        return cls(name or f"{wavelength_nm:.0f}nm", color,
                   excitation_nm=stored if is_excitation else None,
                   emission_nm=None if is_excitation else stored)


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
