"""SegmentationMask: labels -> feature table on the device (reference: R/masks.py)."""
from __future__ import annotations

# property lists of the reference (R/masks.py:15-35)
DEFAULT_CELL_PROPERTY_NAMES = [
    "label",
    "centroid",
    "volume",
    "area",
    "area_convex",
    "perimeter",
    "eccentricity",
    "circularity",
    "solidity",
    "axis_major_length",
    "axis_minor_length",
    "orientation",
]

DEFAULT_INTENSITY_PROPERTY_NAMES = [
    "intensity_mean",
    "intensity_max",
    "intensity_min",
    "intensity_std",
]
