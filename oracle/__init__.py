"""CPU oracle for the arcadia-microscopy-tools hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, with numpy + scipy.ndimage (the same upstream C kernels that
scikit-image wraps) plus one plain-C file for the watershed flood, the arithmetic the
reference reaches through scikit-image on its per-image preprocessing + segmentation +
region-props path.  Every function cites the reference call site (``R/`` =
``/root/reference/src/arcadia_microscopy_tools/``) and the scikit-image 0.18.3 source
line (``SK/`` = ``site-packages/skimage``) it follows.

Rules (see DESIGN.md "Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import this package; the product (``arcadia_microscopy_tools_amd``) never does;
  * parity pin: ``tests/golden/*.npz`` were produced by running the real scikit-image 0.18.3
    / scipy 1.7.1 (conda env of the build container) through ``tools/make_golden.py``;
    ``tests/test_oracle_golden.py`` checks this package against those vectors and against
    the known answers on the reference's own ND2 fixture.
"""
