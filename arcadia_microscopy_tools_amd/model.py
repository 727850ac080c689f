"""``SegmentationModel``: the ``segment()`` / ``batch_segment()`` surface of the reference (R/model.py:28-290).

The reference wraps Cellpose-SAM, whose weights are fetched from the network by name (R/model.py:160-169)
and therefore cannot be loaded offline.  This class keeps the reference's constructor fields, parameter
resolution / validation (same messages), return dtype (int64 labels, background 0) and per-image failure
semantics of ``batch_segment`` (warning + ``None``), and adds a ``backend`` switch.  The DEFAULT is the reference's own
behaviour (``backend="cellpose"``): ``SegmentationModel()`` never silently runs a different algorithm -- without the
cellpose package it fails loudly (``RuntimeError: Failed to load Cellpose model``), and the module keeps the name
``CellposeModel`` at module level, where the reference has it (its tests patch it there).

  backend="classical": the nuclei chain of BASELINE.json config 3 on the GPU -- Gaussian -> Otsu ->
      opening/closing -> EDT -> peak markers -> watershed -> sequential labels (``segment.FovSegmenter``).
      ``cell_diameter_px`` sets the marker spacing (min_distance = round(diameter / 6), 5 px for the default
      30 px).  The Cellpose-specific parameters have no meaning for this algorithm: passing a NON-DEFAULT
      ``flow_threshold`` / ``cellprob_threshold`` / ``num_iterations`` or any ``**cellpose_kwargs`` raises
      (nothing is accepted and ignored); ``batch_size`` is the number of images per launch of ``batch_segment``.
  backend="cellpose" (default): delegates to ``cellpose.models.CellposeModel`` exactly like the reference when that
      package and its weights are available (PyTorch-ROCm device selection is unchanged: ``torch.cuda`` is
      the ROCm device).  Without the package, a flow network given as ``network=`` (any ``torch.nn.Module`` mapping
      (N, C, H, W) images to (N, 3, H, W) = dY, dX, cellprob -- e.g. a locally stored checkpoint) runs in bf16
      through PyTorch-ROCm and its output goes through the HIP flow -> mask post-processing
      (``cellpose_hip.segment_image``: diameter rescaling, tiled forward in batches of ``batch_size``, flow following,
      flow-error filter at ``flow_threshold``, size filters and hole filling; restated from the published algorithm,
      parity unpinned); with neither, ``RuntimeError`` as before.
  backend="cellpose-hip": always the ``network=`` + HIP post-processing route (``network="standin"`` builds the
      random-weight architectural stand-in of ``cellpose_hip.make_standin`` -- for throughput measurements only,
      its masks mean nothing).
"""
from __future__ import annotations

import logging
import warnings
from collections.abc import Sequence
from dataclasses import dataclass, field
from typing import Any, TypedDict

import numpy as np

from .exceptions import SegmentationWarning
from .typing import Float64Array, Int64Array

logger = logging.getLogger(__name__)

try:  # the reference imports this name at module level (R/model.py:9); kept so that it can be patched / replaced
    from cellpose.models import CellposeModel
except Exception as _cellpose_error:  # not installed (or its own imports fail): constructing it says so
    _CELLPOSE_IMPORT_ERROR = _cellpose_error

    class CellposeModel:  # type: ignore[no-redef]
        """Placeholder for ``cellpose.models.CellposeModel`` on machines without the cellpose package."""

        def __init__(self, *args, **kwargs):
            raise ImportError(f"the cellpose package is not available ({_CELLPOSE_IMPORT_ERROR}); use "
                              "backend='classical', or backend='cellpose-hip' with network=")


class CellposeParams(TypedDict):
    """Resolved parameters in ``CellposeModel.eval`` naming (R/model.py:18-25)."""

    diameter: float
    flow_threshold: float
    cellprob_threshold: float
    niter: int | None
    batch_size: int


@dataclass
class SegmentationModel:
    """High-throughput cell segmentation (defaults: R/model.py:67-72)."""

    default_cell_diameter_px: float = 30
    default_flow_threshold: float = 0.4
    default_cellprob_threshold: float = 0
    default_num_iterations: int | None = None
    default_batch_size: int = 8
    device: Any = field(default=None)
    backend: str = "cellpose"
    sigma: float = 2.0
    opening_radius: int = 2
    network: Any = field(default=None, repr=False)
    compute_dtype: str = "bf16"
    _model: Any = field(default=None, init=False, repr=False)
    _net: Any = field(default=None, init=False, repr=False)

    def __post_init__(self) -> None:
        if self.backend not in ("classical", "cellpose", "cellpose-hip"):
            raise ValueError(f"backend must be 'classical', 'cellpose' or 'cellpose-hip', got '{self.backend}'")
        if self.backend == "cellpose-hip" and self.network is None:
            raise ValueError("backend 'cellpose-hip' needs network= (a torch.nn.Module, or 'standin')")
        if self.device is None and self.backend in ("cellpose", "cellpose-hip"):
            self.device = self.find_best_available_device()

    # -- parameters (R/model.py:80-132) ---------------------------------------------------------------
    def _resolve_and_validate_parameters(self, cell_diameter_px, flow_threshold, cellprob_threshold,
                                         num_iterations, batch_size) -> CellposeParams:
        p: CellposeParams = {
            "diameter": cell_diameter_px if cell_diameter_px is not None else self.default_cell_diameter_px,
            "flow_threshold": flow_threshold if flow_threshold is not None else self.default_flow_threshold,
            "cellprob_threshold": cellprob_threshold
            if cellprob_threshold is not None
            else self.default_cellprob_threshold,
            "niter": num_iterations if num_iterations is not None else self.default_num_iterations,
            "batch_size": batch_size if batch_size is not None else self.default_batch_size,
        }
        if p["diameter"] <= 0:
            raise ValueError(f"Cell diameter [px] must be positive, got {p['diameter']}")
        if p["flow_threshold"] < 0:
            raise ValueError(f"Flow threshold must be non-negative, got {p['flow_threshold']}")
        if not (-10 <= p["cellprob_threshold"] <= 10):
            raise ValueError(
                f"Cell probability threshold must be between -10 and 10, got {p['cellprob_threshold']}"
            )
        return p

    @staticmethod
    def find_best_available_device():
        """CUDA/ROCm GPU > MPS > CPU, as a ``torch.device`` (R/model.py:134-158); on ROCm ``torch.cuda`` IS the
        MI355X."""
        import torch

        if torch.cuda.is_available():
            device = torch.device("cuda")
            name = torch.cuda.get_device_name(0)
            mem = torch.cuda.get_device_properties(0).total_memory / (1024**3)
            logger.info(f"Using CUDA GPU: {name} with {mem:.1f} GB memory")
        elif torch.backends.mps.is_available():
            device = torch.device("mps")
            logger.info("Using Apple Metal Performance Shaders (MPS) for acceleration.")
        else:
            device = torch.device("cpu")
            logger.info(f"No GPU acceleration available. Using CPU with {torch.get_num_threads()} threads.")
        return device

    @property
    def cellpose_model(self):
        """Lazy Cellpose model (R/model.py:160-169); only for backend='cellpose'."""
        if self._model is None:
            logger.info(f"Loading Cellpose-SAM model on {self.device}")
            try:
                self._model = CellposeModel(device=self.device)  # the module-level name, as in the reference
            except Exception as e:
                raise RuntimeError(f"Failed to load Cellpose model: {e}") from e
        return self._model

    # -- flow network + HIP post-processing ---------------------------------------------------------------
    def _use_network(self) -> bool:
        """backend 'cellpose-hip', or backend 'cellpose' without the cellpose package but with a network."""
        if self.backend == "cellpose-hip":
            return True
        if self.backend == "cellpose" and self.network is not None:
            return "_CELLPOSE_IMPORT_ERROR" in globals()  # the package is absent: the network + HIP route
        return False

    def _flow_network(self):
        if self._net is None:
            from . import cellpose_hip

            standin = isinstance(self.network, str) and self.network == "standin"
            net = cellpose_hip.make_standin() if standin else self.network
            net, dt = cellpose_hip.prepare_network(net, self.device, self.compute_dtype)
            if standin and self.compute_dtype == "bf16":
                # the stand-in's own graph is known: its forward runs with the elementwise glue fused into HIP passes
                from .device import get_context

                net = cellpose_hip.FusedStandIn(net, get_context())
            self._net = (net, dt)
        return self._net

    def _segment_network(self, intensities: np.ndarray, params: CellposeParams, cellpose_kwargs=None) -> Int64Array:
        """Every parameter of R/model.py:171-215 reaches the route and acts there (``cellpose_hip.segment_image``):
        ``diameter`` resizes around the network, ``batch_size`` is the tile batch, ``flow_threshold`` the flow-error
        filter.  Extra ``**cellpose_kwargs`` are the post-processing / tiling options of ``CellposeModel.eval`` this
        route implements; any other name is refused (never ignored)."""
        from . import cellpose_hip

        kw = dict(cellpose_kwargs or {})
        unknown = sorted(set(kw) - set(cellpose_hip._EVAL_KWARGS))
        if unknown:
            raise TypeError(f"the network + HIP route does not implement CellposeModel.eval option(s) {unknown}; "
                            f"supported: {list(cellpose_hip._EVAL_KWARGS)}")
        net, dt = self._flow_network()
        return cellpose_hip.segment_image(net, intensities, self.device, dt,
                                          cellprob_threshold=params["cellprob_threshold"], niter=params["niter"],
                                          batch_size=params["batch_size"], flow_threshold=params["flow_threshold"],
                                          diameter=params["diameter"], **kw)

    # -- classical backend ------------------------------------------------------------------------------
    def _segment_classical(self, intensities: np.ndarray, params: CellposeParams) -> Int64Array:
        from . import hipops
        from .device import get_context

        a = np.asarray(intensities)
        if a.ndim == 3:
            a = a[0]  # ([channel], height, width): the nuclear channel comes first
        if a.ndim != 2:
            raise ValueError(f"expected an image of shape ([channel], height, width), got {np.shape(intensities)}")
        if a.dtype == np.uint8:
            a = a.astype(np.uint16)
        if a.dtype not in (np.uint16, np.float64):
            a = a.astype(np.float64)
        ctx = get_context()
        d = ctx.asarray(np.ascontiguousarray(a))
        g = hipops.gaussian(d, self.sigma)
        thr = hipops.threshold_otsu(g)
        m0 = hipops.greater_than(g, thr)
        fp = hipops.disk(self.opening_radius)
        m1 = hipops.binary_closing(hipops.binary_opening(m0, fp), fp)
        d2, _ = hipops.edt(m1, want_edt=False)
        min_distance = max(1, int(round(params["diameter"] / 6.0)))
        peaks = hipops.peak_mask(d2, m1, min_distance)
        markers, nmark = hipops.label(peaks, connectivity=1)
        ws = hipops.watershed_edt(d2, markers, m1, seeds_first=True)
        k = int(nmark.numpy()[0])
        labels, _ = hipops.relabel_sequential(ws, max(k, 1))
        return labels.numpy_int64()

    # -- public API (R/model.py:171-290) ----------------------------------------------------------------
    def segment(self, intensities: Float64Array, cell_diameter_px: float | None = None,
                flow_threshold: float | None = None, cellprob_threshold: float | None = None,
                num_iterations: int | None = None, batch_size: int | None = None,
                **cellpose_kwargs: Any) -> Int64Array:
        """Segment one image -> int64 label image (background 0).

        Raises ``ValueError`` for out-of-range parameters and ``RuntimeError`` when the backend fails
        (R/model.py:206-215)."""
        params = self._resolve_and_validate_parameters(
            cell_diameter_px, flow_threshold, cellprob_threshold, num_iterations, batch_size
        )
        try:
            mask = self._segment_one(intensities, params, cellpose_kwargs)
        except Exception as e:
            raise RuntimeError(f"Cellpose segmentation failed: {e}") from e
        return mask if mask.dtype == np.int64 else mask.astype(np.int64)

    def _segment_one(self, intensities, params: CellposeParams, cellpose_kwargs) -> np.ndarray:
        if self._use_network():
            return self._segment_network(intensities, params, cellpose_kwargs)
        if self.backend == "cellpose":
            mask, *_ = self.cellpose_model.eval(x=intensities, **params, **cellpose_kwargs)
            return mask
        self._refuse_unused_classical(params, cellpose_kwargs)
        return self._segment_classical(intensities, params)

    def _refuse_unused_classical(self, params: CellposeParams, cellpose_kwargs) -> None:
        """backend='classical' has no flow field: a caller who tunes a Cellpose-only parameter is told so."""
        if cellpose_kwargs:
            raise TypeError(f"backend='classical' takes no CellposeModel.eval options, got {sorted(cellpose_kwargs)}")
        for name, default in (("flow_threshold", self.default_flow_threshold),
                              ("cellprob_threshold", self.default_cellprob_threshold),
                              ("niter", self.default_num_iterations)):
            if params[name] != default:
                raise ValueError(f"backend='classical' does not use {name} (got {params[name]}, the model's default is "
                                 f"{default}); only cell_diameter_px and batch_size apply to the watershed chain")

    def _batch_classical(self, images, params: CellposeParams):
        """Images of one shape through ``FovSegmenter``, ``batch_size`` images per batch of launches (the per-image loop
        costs a host round trip per image); None if the batch does not qualify (mixed shapes / dtypes, 3-D inputs).
        The segmenter of a (batch, shape) is kept per thread: its buffers are reused by the next call."""
        import threading

        from . import hipops
        from .device import get_context
        from .segment import FovSegmenter

        arrs = []
        for im in images:
            a = np.asarray(im)
            if a.ndim == 3:
                a = a[0]
            if a.ndim != 2 or a.dtype not in (np.uint8, np.uint16):
                return None
            arrs.append(a)
        if not arrs or any(a.shape != arrs[0].shape for a in arrs):
            return None
        H, W = arrs[0].shape
        ctx = get_context()
        min_distance = max(1, int(round(params["diameter"] / 6.0)))
        chunk = max(1, min(int(params["batch_size"]), len(arrs)))
        cache = self.__dict__.setdefault("_seg_cache", threading.local())
        out: list = []
        for i0 in range(0, len(arrs), chunk):
            part = arrs[i0:i0 + chunk]
            key = (len(part), H, W, min_distance, self.sigma, self.opening_radius, id(ctx))
            seg = getattr(cache, "seg", None)
            if seg is None or getattr(cache, "key", None) != key:
                seg = FovSegmenter(len(part), 1, H, W, sigma=self.sigma, radius=self.opening_radius,
                                   min_distance=min_distance, max_cells=max(4096, (H * W) // 64), dapi_index=0, ctx=ctx,
                                   props=False, fused=False)
                cache.seg, cache.key = seg, key
                cache.inp = ctx.empty((len(part), 1, H, W), np.uint16)
            for j, a in enumerate(part):  # each plane converted on its way through the page-locked staging buffer
                ctx.asarray(a, dtype=np.uint16, out=cache.inp[j][0])
            seg.run_c3(cache.inp)
            nm = seg.nmarkers.numpy()
            if (nm < 0).any() or (nm > seg.max_cells).any():
                return None
            # segment() keeps edge cells (clear_border belongs to SegmentationMask): relabel the watershed image itself
            labels, _ = hipops.relabel_sequential(seg.ws, seg.max_cells)
            for j in range(len(part)):
                out.append(labels[j].numpy_int64())
        return out

    def batch_masks(self, intensities_batch: Sequence[np.ndarray], channels, nuclear=0,
                    cell_diameter_px: float | None = None, batch_size: int | None = None, *,
                    remove_edge_cells: bool = True, outline_extractor: str = "cellpose", property_names=None,
                    intensity_property_names=None) -> list:
        """``batch_segment`` and the ``SegmentationMask`` of every image in one call (an addition to the reference's
        interface; the two-call form stays available and gives the same results).

        intensities_batch   images of shape (C, height, width), one plane per entry of ``channels``
        channels            the ``Channel`` of every plane (keys of each mask's ``intensity_image_dict``)
        nuclear             index (or ``Channel``) of the plane that is segmented

        Returns, per image, ``SegmentationMask(self.segment(image[nuclear]), dict(zip(channels, image)),
        remove_edge_cells=..., ...)`` -- the same label image, cell count, outlines and ``cell_properties`` -- or
        ``None`` with a ``SegmentationWarning`` where that construction fails (no cells).  With
        ``backend="classical"`` and uint8 / uint16 images of one shape every image crosses the bus ONCE: labels and
        feature rows stay where the chain made them, only the rows come to the host (``maskbatch.MaskBatcher``), and
        ``mask_image`` (there: the label image after edge-cell removal) / ``label_image`` are downloaded on first
        access."""
        from .masks import SegmentationMask

        channels = list(channels)
        nuc = channels.index(nuclear) if nuclear in channels and not isinstance(nuclear, (int, np.integer)) else int(nuclear)
        if not 0 <= nuc < len(channels):
            raise ValueError(f"nuclear must name one of the {len(channels)} channels, got {nuclear!r}")
        params = self._resolve_and_validate_parameters(cell_diameter_px, None, None, None, batch_size)
        mask_kw = dict(outline_extractor=outline_extractor, property_names=property_names,
                       intensity_property_names=intensity_property_names)
        images = [np.asarray(im) for im in intensities_batch]
        for im in images:
            if im.ndim != 3 or im.shape[0] != len(channels):
                raise ValueError(f"expected images of shape ({len(channels)}, height, width), got {im.shape}")
        rows = None
        if (self.backend == "classical" and remove_edge_cells and images
                and all(im.dtype in (np.uint8, np.uint16) and im.shape == images[0].shape for im in images)):
            rows = self._masks_classical(images, nuc, params)
        out: list = []
        if rows is not None:
            for im, (lab, k, morph, inten) in zip(images, rows):
                if k <= 0:
                    warnings.warn("no cells in an image of the batch (or none away from the frame): None at its index",
                                  SegmentationWarning, stacklevel=2)
                    out.append(None)
                    continue
                out.append(SegmentationMask._from_device(lab, k, morph, inten, dict(zip(channels, im)), **mask_kw))
            return out
        labels = self.batch_segment([im[nuc] for im in images], cell_diameter_px=cell_diameter_px,
                                    batch_size=batch_size, show_progress=False)
        for i, (im, lab) in enumerate(zip(images, labels)):
            try:
                mask = None if lab is None else SegmentationMask(lab, dict(zip(channels, im)),
                                                                 remove_edge_cells=remove_edge_cells, **mask_kw)
                if mask is not None:
                    mask.num_cells  # edge-cell removal may leave nothing: found here, not on a later access
            except ValueError as e:
                warnings.warn(f"no SegmentationMask for image {i}: {e}", SegmentationWarning, stacklevel=2)
                mask = None
            out.append(mask)
        return out

    def _masks_classical(self, images, nuclear: int, params: CellposeParams):
        """The one-pass route of ``batch_masks``; None when a table of the chain overflowed (dense noise)."""
        import threading

        from .device import get_context
        from .maskbatch import MaskBatcher

        C, H, W = images[0].shape
        ctx = get_context()
        min_distance = max(1, int(round(params["diameter"] / 6.0)))
        chunk = max(1, min(int(params["batch_size"]), len(images)))
        cache = self.__dict__.setdefault("_mask_cache", threading.local())
        key = (chunk, C, H, W, nuclear, min_distance, self.sigma, self.opening_radius, id(ctx))
        mb = getattr(cache, "mb", None)
        if mb is None or cache.key != key:
            if mb is not None:
                mb.close()
            mb = MaskBatcher(chunk, C, H, W, nuclear=nuclear, sigma=self.sigma, radius=self.opening_radius,
                             min_distance=min_distance, ctx=ctx)
            cache.mb, cache.key = mb, key
        return mb.run(images)

    def batch_segment(self, intensities_batch: Sequence[Float64Array], cell_diameter_px: float | None = None,
                      flow_threshold: float | None = None, cellprob_threshold: float | None = None,
                      num_iterations: int | None = None, batch_size: int | None = None,
                      show_progress: bool = True, **cellpose_kwargs: Any) -> list[Int64Array | None]:
        """Segment several images with one parameter set; a failing image yields a ``SegmentationWarning`` and
        ``None`` at its index, the rest continue (R/model.py:217-290)."""
        params = self._resolve_and_validate_parameters(
            cell_diameter_px, flow_threshold, cellprob_threshold, num_iterations, batch_size
        )
        masks: list[Int64Array | None] = []
        if self.backend == "classical" and len(intensities_batch) > 1:
            self._refuse_unused_classical(params, cellpose_kwargs)
            try:
                done = self._batch_classical(intensities_batch, params)
            except Exception as e:  # fall back to the per-image loop, which reports failures image by image
                logger.debug("batched classical path failed (%s: %s); segmenting image by image", type(e).__name__, e)
                done = None
            if done is not None:
                return done
        iterator = enumerate(intensities_batch)
        if show_progress:
            try:
                from .utils import get_tqdm  # notebook-aware, as R/model.py:270-273

                iterator = get_tqdm()(iterator, total=len(intensities_batch), desc="Segmenting")
            except Exception:
                pass
        for i, intensities in iterator:
            try:
                mask = self._segment_one(intensities, params, cellpose_kwargs)
                masks.append(mask.astype(np.int64, copy=False))
            except Exception as e:
                warnings.warn(
                    f"Cellpose segmentation failed on image {i}: {e}",
                    SegmentationWarning,
                    stacklevel=2,
                )
                masks.append(None)
        return masks
