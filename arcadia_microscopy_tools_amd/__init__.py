"""MI355X-native drop-in for the per-image preprocessing + segmentation + region-props hot path of
arcadia-microscopy-tools (names re-exported as in the reference, src/arcadia_microscopy_tools/__init__.py:1-20).

Importing the package never touches the GPU; the HIP library (libamt_hip.so) is loaded on first use, and
a missing library or missing MI355X raises ``HipUnavailableError`` -- there is no CPU fallback.
"""
from ._hip import HipError, HipUnavailableError
from .blending import BlendMode, Layer, create_overlay, overlay_channels
from .channels import Channel
from .exceptions import MetadataWarning, SegmentationWarning
from .microscopy import MicroscopyImage
from .pipeline import ImageOperation, Pipeline

__version__ = "0.4.1+amd.1"

# the reference's public names plus the two error types of the HIP boundary
__all__ = sorted(
    {"BlendMode", "Channel", "ImageOperation", "Layer", "MetadataWarning", "MicroscopyImage", "Pipeline",
     "SegmentationWarning", "create_overlay", "overlay_channels"} | {"HipError", "HipUnavailableError"}
)
