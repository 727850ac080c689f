"""Run existing code unchanged: ``import arcadia_microscopy_tools_amd.compat as c; c.install()`` makes
``import arcadia_microscopy_tools`` (and its hot-path submodules) resolve to this package, so scripts written against the
reference -- ``from arcadia_microscopy_tools.operations import rescale_by_percentile`` and so on -- run on the MI355X
without an edit.  Only the modules this package implements are aliased (the hot path of SURVEY.md section 8 and its
neighbours); ``leica`` (LIF files) stays the reference's own concern and raises ImportError here.

``install()`` refuses to shadow a real installation of the reference unless ``force=True``; ``uninstall()`` removes
the aliases again.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys

REFERENCE_NAME = "arcadia_microscopy_tools"
# reference submodule -> module of this package with the same public names
SUBMODULES = ("blending", "channels", "exceptions", "masks", "metadata_structures", "microplate", "microscopy", "model",
              "nikon", "operations", "pipeline", "typing", "utils")
_installed: list[str] = []


def install(force: bool = False) -> None:
    """Alias the reference's import names to this package for the running interpreter."""
    if REFERENCE_NAME in sys.modules and sys.modules[REFERENCE_NAME].__name__ != __package__ and not force:
        raise ImportError(f"{REFERENCE_NAME} is already imported from {sys.modules[REFERENCE_NAME].__file__}; "
                          "pass force=True to replace it for this process")
    if not force and REFERENCE_NAME not in sys.modules:
        try:
            found = importlib.util.find_spec(REFERENCE_NAME)
        except (ImportError, ValueError):
            found = None
        if found is not None:
            raise ImportError(f"{REFERENCE_NAME} is installed ({found.origin}); pass force=True to shadow it")
    package = importlib.import_module(__package__)
    sys.modules[REFERENCE_NAME] = package
    _installed.append(REFERENCE_NAME)
    for name in SUBMODULES:
        module = importlib.import_module(f"{__package__}.{name}")
        sys.modules[f"{REFERENCE_NAME}.{name}"] = module
        _installed.append(f"{REFERENCE_NAME}.{name}")


def uninstall() -> None:
    """Remove the aliases ``install()`` created."""
    while _installed:
        sys.modules.pop(_installed.pop(), None)
