"""ctypes binding of libamt_hip.so (include/amt_hip.h).  No torch types cross this boundary.

The library must be built first (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C arcadia_microscopy_tools_amd/csrc``).  There is NO CPU fallback: if the shared library is
missing, or no MI355X is visible when a context is requested, the product raises ``HipUnavailableError``.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMT_HIP_LIB: another build of the same library (instrumented / experimental variants made by tools/), never a fallback
LIB_PATH = os.environ.get("AMT_HIP_LIB") or os.path.join(_HERE, "libamt_hip.so")

# element type codes (amt_hip.h)
U8, U16, I32, F64, I64, F32 = 0, 1, 2, 3, 4, 5
MODE_NEAREST, MODE_REFLECT, MODE_MIRROR, MODE_CONSTANT, MODE_WRAP = 0, 1, 2, 3, 4
MODES = {"nearest": 0, "reflect": 1, "mirror": 2, "constant": 3, "wrap": 4}
THR_OTSU = 0
WS_TIES = {"exact": 0, "raster": 1, "report": 2, "refuse": 2}
RP_COLS = (
    "area", "centroid-0", "centroid-1", "bbox-0", "bbox-1", "bbox-2", "bbox-3", "perimeter",
    "axis_major_length", "axis_minor_length", "eccentricity", "orientation", "area_convex", "solidity",
)
RP_NCOLS = len(RP_COLS)


class HipUnavailableError(RuntimeError):
    """libamt_hip.so could not be loaded, or no gfx950 device is usable."""


class HipError(RuntimeError):
    """A libamt_hip call failed (message from amt_last_error)."""


_P = c_void_p
_SIGS = {
    # name: (restype, argtypes)
    "amt_device_count": (c_int, []),
    "amt_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "amt_ctx_create_on_stream": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "amt_ctx_destroy": (c_int, [_P]),
    "amt_ctx_set_fork": (c_int, [_P, c_int]),
    "amt_ctx_stream": (c_int, [_P, POINTER(c_void_p)]),
    "amt_last_error": (c_char_p, []),
    "amt_version": (c_char_p, []),
    "amt_device_name": (c_int, [_P, c_char_p, c_int]),
    "amt_malloc": (c_int, [_P, c_size_t, POINTER(c_void_p)]),
    "amt_free": (c_int, [_P, _P]),
    "amt_memcpy_h2d": (c_int, [_P, _P, _P, c_size_t]),
    "amt_memcpy_d2h": (c_int, [_P, _P, _P, c_size_t]),
    "amt_memcpy_d2d": (c_int, [_P, _P, _P, c_size_t]),
    "amt_memset": (c_int, [_P, _P, c_int, c_size_t]),
    "amt_sync": (c_int, [_P]),
    "amt_stream_wait": (c_int, [_P, _P]),
    "amt_event_create": (c_int, [_P, POINTER(c_void_p)]),
    "amt_event_record": (c_int, [_P, _P]),
    "amt_event_wait": (c_int, [_P, _P]),
    "amt_event_sync": (c_int, [_P, _P]),
    "amt_event_destroy": (c_int, [_P, _P]),
    "amt_host_alloc": (c_int, [c_size_t, POINTER(c_void_p)]),
    "amt_host_copy": (c_int, [c_void_p, c_void_p, c_size_t]),
    "amt_host_minmax_int": (c_int, [c_void_p, c_int, c_int, c_size_t, c_void_p]),
    "amt_host_narrow_i64_i32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "amt_host_free": (c_int, [_P]),
    "amt_timer_create": (c_int, [_P, POINTER(c_void_p)]),
    "amt_timer_start": (c_int, [_P, _P]),
    "amt_timer_stop": (c_int, [_P, _P]),
    "amt_timer_elapsed_ms": (c_int, [_P, _P, POINTER(c_float)]),
    "amt_timer_destroy": (c_int, [_P, _P]),
    "amt_deinterleave_u16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_gaussian": (c_int, [_P, _P, c_int, c_double, _P, c_int, c_int, c_int, _P, c_int, c_int, c_double,
                             c_size_t, _P]),
    "amt_dog": (c_int, [_P, _P, c_int, c_double, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_double]),
    "amt_sub_clip0_f64": (c_int, [_P, _P, _P, _P, c_int, c_size_t]),
    "amt_rescale": (c_int, [_P, _P, c_int, _P, c_double, c_double, _P, c_int, c_size_t]),
    "amt_convert_u16_f64": (c_int, [_P, _P, c_double, _P, c_size_t]),
    "amt_add_scalar_f64": (c_int, [_P, _P, c_double, _P, c_size_t]),
    "amt_hist_u16": (c_int, [_P, _P, _P, c_int, c_size_t]),
    "amt_hist_range_f64": (c_int, [_P, _P, ctypes.c_double, ctypes.c_int64, _P, c_int, c_size_t]),
    "amt_minmax_f64": (c_int, [_P, _P, _P, c_int, c_size_t]),
    "amt_hist_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_size_t]),
    "amt_percentile_u16": (c_int, [_P, _P, _P, c_int, _P, c_int, c_size_t]),
    "amt_percentile_f64": (c_int, [_P, _P, _P, c_int, _P, c_int, c_size_t]),
    "amt_masked_sums_f64": (c_int, [_P, _P, _P, _P, c_int, c_size_t]),
    "amt_pad_edge": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "amt_copy_rect": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "amt_threshold_value": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_int, c_size_t, _P]),
    "amt_threshold_gt": (c_int, [_P, _P, c_int, _P, _P, c_int, c_size_t]),
    "amt_window_threshold": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_double]),
    "amt_window_threshold_yx": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_double]),
    "amt_window_threshold_nd": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double,
                                        c_double]),
    "amt_threshold_gt_image": (c_int, [_P, _P, c_int, _P, _P, c_size_t]),
    "amt_binary_erode": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int]),
    "amt_binary_dilate": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int]),
    "amt_binary_open": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int]),
    "amt_binary_close": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int]),
    "amt_threshold_open_close": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, c_int]),
    "amt_otsu_f64_bins": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_size_t]),
    "amt_threshold_open_close_bins": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int]),
    "amt_rank_filter": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_double]),
    "amt_gaussian_otsu_codes_supported": (c_int, [c_int, c_int, c_int, c_int, c_size_t]),
    "amt_gaussian_otsu_codes": (c_int, [_P, _P, c_double, c_int, c_int, c_int, _P, c_int, c_int, c_size_t, _P, _P, _P, _P,
                                        _P]),
    "amt_rank_filter_sub": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int,
                                    c_double]),
    "amt_subtract": (c_int, [_P, _P, _P, _P, c_int, c_size_t]),
    "amt_label": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_label_mask": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_nn_affine_act_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "amt_label_sparse": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int]),
    "amt_label_sparse_reuse": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "amt_clear_border": (c_int, [_P, _P, _P, c_int, c_int, c_int]),
    "amt_relabel_sequential": (c_int, [_P, _P, _P, _P, c_int, c_size_t, c_int]),
    "amt_clear_border_relabel": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "amt_keep_labels": (c_int, [_P, _P, _P, _P, c_int, c_size_t, c_int]),
    "amt_cast_i32_i64": (c_int, [_P, _P, _P, c_size_t]),
    "amt_edt": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int]),
    "amt_peak_mask": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_peak_mask_reuse": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "amt_watershed_edt": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_watershed_f64": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int]),
    "amt_watershed_edt_cleared": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "amt_watershed_edt_cleared_sparse": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P,
                                                 c_int]),
    "amt_watershed_edt_ex": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "amt_watershed_f64_ex": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "amt_regionprops": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_regionprops_intensity_u16": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int]),
    "amt_regionprops_full_u16": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_regionprops_intensity_f64": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int]),
    "amt_convolve_axis0": (c_int, [_P, _P, c_int, c_double, _P, c_int, c_int, c_int, _P, c_int, c_int, c_double]),
    "amt_max_i32": (c_int, [_P, _P, _P, c_int, c_size_t]),
    "amt_pack_plate_rows": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "amt_label_bboxes": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_contours_find": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "amt_contours_emit": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "amt_borders_find": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "amt_borders_emit": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "amt_cellpose_masks": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int, c_float, c_int]),
    "amt_cellpose_masks_ex": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_int, c_float, c_int,
                                      c_float, c_int]),
    "amt_cellpose_flow_error": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int]),
    "amt_fill_holes_remove_small": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "amt_overlay": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int]),
}

_lib = None
_lib_lock = threading.Lock()
_runtime = None  # which HIP runtime the process ended up with: "torch" | "system"


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (same SONAME as
    /opt/rocm's).  If libamt_hip.so pulls in the system runtime first and torch is imported later, the process
    holds TWO runtimes and torch's cannot open the GPU ("No HIP GPUs are available") -- and torch.distributed /
    RCCL (plate.py) must see the memory and streams this library uses.  So when a torch wheel with a bundled
    runtime is installed, that runtime is loaded first and libamt_hip.so binds to it; torch itself is NOT imported.
    ``AMT_HIP_RUNTIME=system`` keeps the system runtime (single-GPU processes that never import torch)."""
    global _runtime
    import importlib.util
    import sys

    choice = os.environ.get("AMT_HIP_RUNTIME", "auto")
    if choice == "system" and "torch" not in sys.modules:
        _runtime = "system"
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
                _runtime = "torch"
                return
            except OSError:
                pass
    _runtime = "system"


def hip_runtime() -> str:
    """"torch" when the process shares PyTorch's bundled HIP runtime, "system" for /opt/rocm's."""
    load_library()
    return _runtime


# entry points that may wait (for the device, the allocator or host memory): the interpreter lock is released around them
_BLOCKING = frozenset((
    "amt_sync", "amt_event_sync", "amt_memcpy_h2d", "amt_memcpy_d2h", "amt_memcpy_d2d", "amt_host_alloc", "amt_host_free",
    "amt_host_copy", "amt_host_minmax_int", "amt_host_narrow_i64_i32", "amt_ctx_create", "amt_ctx_create_on_stream",
    "amt_ctx_destroy", "amt_malloc", "amt_free", "amt_timer_elapsed_ms",
))


def load_library():
    """Load libamt_hip.so and declare every prototype.  Raises HipUnavailableError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailableError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
            )
        _share_hip_runtime_with_torch()
        try:
            lib = ctypes.CDLL(LIB_PATH)
            quick = ctypes.PyDLL(LIB_PATH) if os.environ.get("AMT_GIL", "keep") == "keep" else lib
        except OSError as e:  # missing ROCm runtime etc.
            raise HipUnavailableError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            for handle in {id(lib): lib, id(quick): quick}.values():
                fn = getattr(handle, name)  # AttributeError = header / library mismatch: fail loudly
                fn.restype = res
                fn.argtypes = args
            if name not in _BLOCKING:
                # enqueue-only entry points (a few microseconds each) are called WITHOUT releasing the interpreter
                # lock: a chain is ~50 such calls, and worker threads that drop and re-take the lock around each of
                # them spend their time handing it to one another (tools/api_profile.py: four workers were no faster
                # than one).  Calls that wait for the device or move host memory keep releasing it.
                setattr(lib, name, getattr(quick, name))
        _lib = lib
    return _lib


def exported_names():
    return sorted(_SIGS)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load_library().amt_last_error().decode("utf-8", "replace")
        if rc == -4:
            raise HipUnavailableError(msg)
        if rc == -1:
            raise ValueError(msg)
        if rc == -3:
            raise MemoryError(msg)
        raise HipError(f"{what}: {msg} (code {rc})")
