"""MI355X-native drop-in for the per-image preprocessing + segmentation + region-props hot path of
arcadia-microscopy-tools (reference: src/arcadia_microscopy_tools/__init__.py:1-20 for the names).

Importing the package never touches the GPU; the HIP library (libamt_hip.so) is loaded on first use
and a missing library or missing MI355X raises ``HipUnavailableError`` (there is no CPU fallback).
"""
from ._hip import HipError, HipUnavailableError

__version__ = "0.4.1+amd.0"

__all__ = ["HipError", "HipUnavailableError"]
