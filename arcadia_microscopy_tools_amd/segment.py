"""Batch segmentation + region-props drivers for the BASELINE.json configurations, fully on the device.

Stages (CPU restatement: oracle/chains.py):
  C2  Gaussian(sigma) -> Otsu -> '>' -> binary opening(disk r) -> binary closing(disk r) -> label (8-conn)
  C3  C2 mask on the DAPI channel -> EDT -> peak markers (min_distance) -> watershed (seeded relief)
      -> clear_border -> relabel_sequential -> morphology table + per-channel intensity table

``FovSegmenter`` preallocates every buffer for a batch of B fields of view, enqueues the whole chain
on one HIP stream without any host synchronisation, and only copies labels / tables back on request.
This is the classical backend behind ``SegmentationModel.segment`` (reference signature:
R/model.py:171-215) and what bench.py times.
"""
from __future__ import annotations

import numpy as np

from . import _hip, hipops
from .device import Context, DeviceArray, get_context

DEFAULT_CHANNELS = ("BRIGHTFIELD", "DAPI", "FITC", "TRITC")


class StageTimes:
    """HIP-event timers around each stage of one run (all on the segmenter's stream)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.timers: dict[str, object] = {}
        self.order: list[str] = []

    def begin(self, name):
        t = self.timers.get(name)
        if t is None:
            t = self.ctx.timer()
            self.timers[name] = t
            self.order.append(name)
        t.start()
        return t

    def ms(self) -> dict[str, float]:
        return {k: self.timers[k].elapsed_ms() for k in self.order}


class FovSegmenter:
    """Preallocated config-2 / config-3 chain for B fields of view of shape (C, H, W) uint16."""

    def __init__(self, batch: int, C: int, H: int, W: int, *, sigma: float = 2.0, radius: int = 2,
                 min_distance: int = 5, max_cells: int = 4096, dapi_index: int = 1, ctx: Context | None = None,
                 props: bool = True, profile: bool = False, fused: bool = True, low_traffic: bool = False, bin_plane: bool = True,
                 relief: str = "seeded", ties: str = "exact", marker_list: bool = True):
        self.ctx = ctx or get_context()
        self.B, self.C, self.H, self.W = int(batch), int(C), int(H), int(W)
        self.sigma, self.radius, self.min_distance = float(sigma), int(radius), int(min_distance)
        self.max_cells, self.dapi_index, self.props = int(max_cells), int(dapi_index), bool(props)
        # fused = watershed + clear_border + relabel in one C-ABI call (self.ws is then only flood scratch: use
        # fused=False to inspect the watershed image itself)
        self.fused = bool(fused)
        # relief: "seeded" = the config-3 recipe (marker pixels spread first: oracle/skops.py:seeded_flood_image; cannot
        # tie); "plain" = SURVEY.md A.8 as written, watershed(-edt, markers, mask), whose equal-valued markers are
        # resolved by ``ties`` ('exact' = scikit-image's single heap, emulated for the planes that tie)
        if relief not in ("seeded", "plain"):
            raise ValueError(f"relief must be 'seeded' or 'plain', got {relief!r}")
        self.relief, self.ties = relief, ties
        if relief == "plain":
            self.fused = False
        self.tied = None  # per-plane tie flags of the last plain-relief run
        # marker_list: hand the watershed the list of marker pixels label_sparse keeps (its statistics pass then skips
        # the marker plane); False = the round-2 call (A/B, identical labels)
        self.marker_list = bool(marker_list)
        # low_traffic = Gaussian -> Otsu -> '>' through amt_gaussian_otsu_codes: the float64 smoothed planes are never
        # made (8 instead of 26 bytes of HBM traffic per pixel, 25 MB less memory per field of view; same masks, same
        # thresholds).  Off by default: the Gaussian is fp64-issue bound, so computing it twice costs more time than
        # the two reads of the float64 plane it saves (rocprofv3, 32 FOVs: 1.13 ms against 0.93 ms)
        self.low_traffic = bool(low_traffic)
        # bin_plane: the Otsu histogram pass leaves every sample's bin as a byte plane and '>' reads that plane
        # (amt_otsu_f64_bins / amt_threshold_open_close_bins); False = compare the float64 plane (same masks)
        self.bin_plane = bool(bin_plane)
        self.footprint = hipops.disk(self.radius)
        c, B = self.ctx, self.B
        shp = (B, self.H, self.W)
        # Gaussian -> Otsu -> '>' without the float64 image when the fused path takes this shape (2-byte codes instead)
        self.codes_path = None  # decided on the first batch (needs the batch's alignment)
        self._gauss = None
        self.codes = self.thr_code = self.ghist = None
        self._bins = self._thr_code = None
        self.thr = c.empty((B,), np.float64)
        self.gmm = c.empty((B, 2), np.float64)  # [min, max] of the smoothed image, folded in by the Gaussian
        self.mask_a = c.empty(shp, np.uint8)
        self.mask_b = c.empty(shp, np.uint8)
        self.labels8 = None  # config 2 only, allocated on demand
        self.count8 = c.empty((B,), np.int32)
        self.d2 = c.empty(shp, np.int32)
        self.peaks = c.zeros(shp, np.uint8)  # zeroed once; see self._marker_keep
        # marker planes are zeroed ONCE: every run clears only the pixels the previous run wrote (label_sparse keep=)
        self.markers = c.zeros(shp, np.int32)
        self._marker_keep = (c.empty((B, hipops.label_sparse_capacity(self.H, self.W)), np.int32), c.zeros((B,), np.int32))
        self.nmarkers = c.zeros((B,), np.int32)
        self.ws = c.empty(shp, np.int32)
        self.labels = c.empty(shp, np.int32)
        self.ncells = c.empty((B,), np.int32)
        self.table = c.empty((B, self.max_cells, _hip.RP_NCOLS), np.float64)
        self.itable = c.empty((B, self.max_cells, self.C, 4), np.float64)
        self.times = StageTimes(self.ctx) if profile else None
        self._ran = None

    # ---------------------------------------------------------------------------------------------
    def _stage(self, name):
        if self.times is not None:
            if getattr(self, "_open", None) is not None:
                self._open.stop()
            self._open = self.times.begin(name)

    def _end(self):
        if self.times is not None and getattr(self, "_open", None) is not None:
            self._open.stop()
            self._open = None

    def _check_fovs(self, fovs: DeviceArray) -> DeviceArray:
        """Validate the batch and bind it to THIS segmenter's context: operators run on the stream of their
        input's context, and a batch part may have been uploaded by another context (bench.py's stream split, the
        feeder's buffers) -- every kernel of the chain must be on self.ctx's stream."""
        if fovs.dtype != np.uint16 or fovs.shape != (self.B, self.C, self.H, self.W):
            raise ValueError(f"expected uint16 FOV batch of shape {(self.B, self.C, self.H, self.W)}, got "
                             f"{fovs.dtype} {fovs.shape}")
        return fovs.on(self.ctx)

    def mask_chain(self, fovs: DeviceArray) -> DeviceArray:
        """Gaussian -> Otsu -> '>' -> opening -> closing on the DAPI channel of every FOV."""
        if self.codes_path is None:
            self.codes_path = self.low_traffic and hipops.gaussian_otsu_codes_supported(
                fovs, self.sigma, channel=self.dapi_index)
            if self.codes_path:
                shp = (self.B, self.H, self.W)
                self.codes = self.ctx.empty(shp, np.uint16)
                self.thr_code = self.ctx.empty((self.B,), np.float64)
                self.ghist = self.ctx.empty((self.B, 256), np.uint32)
        if self.codes_path:
            # two passes over the uint16 input (min / max, then histogram + 2-byte codes): `gaussian > thr` is exactly
            # `codes > thr_code`, and the float64 plane (8 B/px written, read twice) is never made
            self._stage("gaussian_otsu")
            hipops.gaussian_otsu_codes(fovs, self.sigma, self.codes, self.thr, self.thr_code, self.gmm, self.ghist,
                                       channel=self.dapi_index)
            self._stage("threshold_open_close")
            hipops.threshold_open_close(self.codes, self.thr_code, self.footprint, out=self.mask_a)
            return self.mask_a
        self._stage("gaussian")
        hipops.gaussian(fovs, self.sigma, channel=self.dapi_index, out=self.gauss, minmax_out=self.gmm)
        if not self.bin_plane:
            self._stage("otsu")
            hipops.threshold_otsu(self.gauss, out=self.thr, minmax=self.gmm)
            self._stage("threshold_open_close")
            hipops.threshold_open_close(self.gauss, self.thr, self.footprint, out=self.mask_a)
            return self.mask_a
        if self._bins is None:
            # the histogram pass leaves every sample's bin as a byte plane, and the threshold comparison reads that
            # plane instead of the float64 one (8 -> 1 byte per pixel, identical masks)
            self._bins = self.ctx.empty((self.B, self.H, self.W), np.uint8)
            self._thr_code = self.ctx.empty((self.B,), np.float64)
        self._stage("otsu")
        hipops.threshold_otsu_bins(self.gauss, self.gmm, self.thr, self._thr_code, self._bins)
        # '>' + opening + closing as one bit-packed chain (identical to the three separate operators)
        self._stage("threshold_open_close")
        hipops.threshold_open_close(self.gauss, self.thr, self.footprint, out=self.mask_a, bins=self._bins,
                                    thr_code=self._thr_code)
        return self.mask_a

    @property
    def gauss(self) -> DeviceArray:
        """The float64 smoothed DAPI planes of the separate-operator path (allocated on first use)."""
        if self._gauss is None:
            self._gauss = self.ctx.empty((self.B, self.H, self.W), np.float64)
        return self._gauss

    def run_c2(self, fovs: DeviceArray) -> DeviceArray:
        """BASELINE configs[1]: the mask chain + 8-connected labelling.  Returns int32 labels (B, H, W)."""
        fovs = self._check_fovs(fovs)
        if self.labels8 is None:
            self.labels8 = self.ctx.empty((self.B, self.H, self.W), np.int32)
        mask = self.mask_chain(fovs)
        self._stage("label8")
        hipops.label(mask, 2, out=self.labels8, count=self.count8)
        self._end()
        self._ran = "c2"
        return self.labels8

    def run_c3(self, fovs: DeviceArray) -> DeviceArray:
        """BASELINE configs[2]: nuclei watershed + morphology / intensity tables.  Returns int32 labels."""
        fovs = self._check_fovs(fovs)
        mask = self.mask_chain(fovs)
        self._stage("edt")
        hipops.edt(mask, want_edt=False, d2_out=self.d2)
        self._stage("peaks")
        hipops.peak_mask(self.d2, mask, self.min_distance, out=self.peaks, keep=self._marker_keep, status=self.nmarkers)
        self._stage("markers")
        hipops.label_sparse(self.peaks, 1, out=self.markers, count=self.nmarkers, keep=self._marker_keep)  # sparse
        if self.fused:
            # watershed + clear_border + relabel_sequential in one call: the watershed image is never written out
            # (self.ws only receives the pixels of flooded components)
            self._stage("watershed_clear_relabel")
            hipops.watershed_edt_cleared(self.d2, self.markers, mask, self.nmarkers, self.max_cells, scratch=self.ws,
                                         out=self.labels, count=self.ncells,
                                         marker_list=self._marker_keep if self.marker_list else None)
        else:
            self._stage("watershed")
            if self.relief == "plain":
                if self.tied is None:
                    self.tied = self.ctx.zeros((self.B,), np.int32)
                hipops.watershed_edt(self.d2, self.markers, mask, seeds_first=False, out=self.ws, ties=self.ties,
                                     ties_out=self.tied)
            else:
                hipops.watershed_edt(self.d2, self.markers, mask, seeds_first=True, out=self.ws)
            # every watershed label is one 4-connected region grown from one marker component, so
            # clear_border + relabel_sequential (R/masks.py:56,65) collapse into one flag-and-renumber pass; the
            # markers lie inside the mask, so exactly the labels 1..nmarkers occur in the result
            self._stage("clear_border")
            hipops.clear_border_relabel(self.ws, self.max_cells, out=self.labels, count=self.ncells,
                                        nlabels=self.nmarkers)
        if self.props:
            self._stage("regionprops")
            hipops.regionprops_full(self.labels, fovs, self.max_cells, out=self.table, iout=self.itable)
        self._end()
        self._ran = "c3"
        return self.labels

    # ---------------------------------------------------------------------------------------------
    def result(self, channel_names=DEFAULT_CHANNELS) -> "SegmentationResult":
        if self._ran != "c3":
            raise RuntimeError("run_c3() has not been called")
        nm = self.nmarkers.numpy()
        if (nm < 0).any():
            raise _hip.HipError("the peak mask of a field of view exceeded the sparse-labelling capacity")
        if (nm > self.max_cells).any():
            raise _hip.HipError(
                f"a field of view produced {int(nm.max())} markers but max_cells={self.max_cells}; "
                "raise max_cells")
        return SegmentationResult(self, channel_names)


class SegmentationResult:
    """Views onto a finished config-3 run; host copies are made lazily."""

    def __init__(self, seg: FovSegmenter, channel_names):
        self.seg = seg
        self.channel_names = tuple(channel_names)
        self.ncells = seg.ncells.numpy()

    def labels_device(self) -> DeviceArray:
        return self.seg.labels

    def labels_numpy(self) -> np.ndarray:
        """int64 label images (B, H, W), background 0 (dtype contract: R/model.py:215, R/masks.py:63-65)."""
        return self.seg.labels.numpy_int64()

    def feature_tables(self) -> list[dict[str, np.ndarray]]:
        """One dict per FOV with the keys of ``SegmentationMask.cell_properties`` (R/masks.py:247-328,
        default property lists R/masks.py:15-35)."""
        t = self.seg.table.numpy()
        it = self.seg.itable.numpy() if self.seg.props else None
        out = []
        for b in range(self.seg.B):
            k = int(self.ncells[b])
            out.append(assemble_cell_properties(t[b, :k], None if it is None else it[b, :k], self.channel_names))
        return out


def assemble_cell_properties(morph: np.ndarray, inten: np.ndarray | None, channel_names,
                             property_names=None, intensity_property_names=None) -> dict[str, np.ndarray]:
    """Device tables -> the dict R/masks.py:247-328 builds (same keys, same order, derived columns on the
    host exactly as the reference derives them: circularity :292-297, volume :302-305, renames :311-314)."""
    from .masks import DEFAULT_CELL_PROPERTY_NAMES, DEFAULT_INTENSITY_PROPERTY_NAMES

    property_names = list(DEFAULT_CELL_PROPERTY_NAMES if property_names is None else property_names)
    if intensity_property_names is None:
        intensity_property_names = list(DEFAULT_INTENSITY_PROPERTY_NAMES) if inten is not None else []
    col = {c: morph[:, i] for i, c in enumerate(_hip.RP_COLS)}
    k = morph.shape[0]
    needs_circ = "circularity" in property_names
    needs_vol = "volume" in property_names
    sk_props = [p for p in property_names if p not in ("circularity", "volume")]
    added = set()
    for dep in ["area", "perimeter"] if needs_circ else []:
        if dep not in sk_props:
            sk_props.append(dep)
            added.add(dep)
    for dep in ["axis_major_length", "axis_minor_length"] if needs_vol else []:
        if dep not in sk_props:
            sk_props.append(dep)
            added.add(dep)
    props: dict[str, np.ndarray] = {}
    for p in sk_props:
        if p == "label":
            props["label"] = np.arange(1, k + 1, dtype=np.int64)
        elif p == "centroid":
            props["centroid-0"] = col["centroid-0"].copy()
            props["centroid-1"] = col["centroid-1"].copy()
        elif p == "bbox":
            for i in range(4):
                props[f"bbox-{i}"] = col[f"bbox-{i}"].astype(np.int64)
        elif p in col:
            props[p] = col[p].copy()
        else:
            raise AttributeError(f"property '{p}' is not available on the device path")
    if needs_circ:
        area, per = props["area"], props["perimeter"]
        with np.errstate(divide="ignore", invalid="ignore"):
            props["circularity"] = np.where(per > 0, (4.0 * np.pi * area) / (per**2), 0.0)
    if needs_vol:
        a = props["axis_major_length"] / 2.0
        b = props["axis_minor_length"] / 2.0
        props["volume"] = np.where((a > 0) & (b > 0), (4.0 / 3.0) * np.pi * a * b * b, 0.0)
    for p in added:
        props.pop(p, None)
    if "centroid-0" in props:
        props["centroid_y"] = props.pop("centroid-0")
    if "centroid-1" in props:
        props["centroid_x"] = props.pop("centroid-1")
    if inten is not None and intensity_property_names:
        order = {"intensity_mean": 0, "intensity_max": 1, "intensity_min": 2, "intensity_std": 3}
        for ci, name in enumerate(channel_names):
            for p in intensity_property_names:
                if p not in order:
                    raise AttributeError(f"intensity property '{p}' is not available on the device path")
                props[f"{p}_{str(name).lower()}"] = inten[:, ci, order[p]].copy()
    return props


def segment_fovs(fovs, *, ctx: Context | None = None, channel_names=DEFAULT_CHANNELS, **kw) -> SegmentationResult:
    """Convenience: (B, C, H, W) uint16 (numpy or DeviceArray) -> config-3 ``SegmentationResult``."""
    ctx = ctx or get_context()
    d = fovs if isinstance(fovs, DeviceArray) else ctx.asarray(np.ascontiguousarray(fovs, dtype=np.uint16))
    B, C, H, W = d.shape
    seg = FovSegmenter(B, C, H, W, ctx=ctx, **kw)
    seg.run_c3(d)
    return seg.result(channel_names)
