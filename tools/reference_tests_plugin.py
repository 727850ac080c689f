"""pytest plugin: run the REFERENCE's own test files against this package.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tools python -m pytest -p reference_tests_plugin -p no:cacheprovider \\
        --import-mode=importlib --rootdir=/tmp -q <reference checkout>/src/arcadia_microscopy_tools/tests/test_model.py

Loading the plugin aliases ``arcadia_microscopy_tools`` (and its hot-path submodules) to ``arcadia_microscopy_tools_amd``
(``compat.install``), so the reference's tests import this package under the reference's names.  Nothing is written
into the reference checkout (no byte code, no pytest cache).  Round 2, no GPU in the build container: test_model 23/23,
test_channels 17/17, test_microplate 8/8, test_microscopy 3/3, test_pipeline 36/39 and test_blending 22/36 -- every
failure is ``HipUnavailableError`` (the test computes on the device); test_masks additionally imports scikit-image.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import arcadia_microscopy_tools_amd.compat as compat  # noqa: E402

compat.install(force=True)
