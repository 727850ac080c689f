"""The BASELINE.json configurations as CPU chains over the oracle ops.  TEST INFRASTRUCTURE ONLY.

C2 (configs[1]): Gaussian(sigma=2) -> Otsu -> '>' -> binary open(disk 2) -> close(disk 2) -> label (8-conn).
C3 (configs[2]): C2 mask on DAPI -> EDT -> peak markers (min_distance 5) -> watershed(seeded relief,
    markers, mask) -> clear_border -> relabel_sequential -> cell_properties (morphology + 4-channel
    intensity).  Stage list: SURVEY.md section 8d; marker recipe: SURVEY.md A.8 and
    ``skops.seeded_flood_image``.
These are also what ``bench.py`` times as ``cpu_baseline`` (kind "port").
"""
from __future__ import annotations

import numpy as np

from . import regionprops as rp
from . import skops
from .watershed import watershed

CHANNEL_NAMES = ("BRIGHTFIELD", "DAPI", "FITC", "TRITC")


def c2_mask(plane_u16: np.ndarray, sigma: float = 2.0, radius: int = 2):
    g = skops.gaussian(plane_u16, sigma)
    t = skops.threshold_otsu(g)
    m = g > t
    se = skops.disk(radius)
    m = skops.binary_opening(m, se)
    m = skops.binary_closing(m, se)
    return m, g, t


def c2_chain(plane_u16: np.ndarray, sigma: float = 2.0, radius: int = 2):
    """Returns the int64 label image (8-connected, raster numbering)."""
    m, _, _ = c2_mask(plane_u16, sigma, radius)
    return skops.label(m)


def c3_labels(dapi_u16: np.ndarray, sigma: float = 2.0, radius: int = 2, min_distance: int = 5, relief: str = "seeded"):
    """DAPI plane -> sequential int64 nuclei labels (edge cells removed). Also returns intermediates.
    ``relief``: "seeded" (the config-3 recipe, ``skops.seeded_flood_image``) or "plain" (SURVEY.md A.8 as written:
    ``watershed(-edt, markers, mask)``, equal-valued markers ordered by scikit-image's heap)."""
    mask, _, _ = c2_mask(dapi_u16, sigma, radius)
    edt = skops.distance_transform_edt(mask)
    markers, _ = skops.peak_markers(edt, mask, min_distance)
    relief = skops.seeded_flood_image(edt, markers) if relief == "seeded" else -edt
    ws = watershed(relief, markers, mask=mask, connectivity=1)
    cleared = skops.clear_border(ws)
    labels = skops.relabel_sequential(cleared).astype(np.int64) if cleared.max() > 0 else cleared.astype(np.int64)
    return labels, dict(mask=mask, edt=edt, markers=markers, watershed=ws)


def c3_chain(fov_u16: np.ndarray, channel_names=CHANNEL_NAMES, dapi_index: int = 1, **kw):
    """(C,Y,X) uint16 FOV -> (labels int64, feature dict as R/masks.py:247-328 returns it)."""
    labels, _ = c3_labels(fov_u16[dapi_index], **kw)
    inten = {name: fov_u16[i] for i, name in enumerate(channel_names)}
    props = rp.cell_properties(labels, inten)
    return labels, props
