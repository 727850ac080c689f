"""The operator boundary: ``ImageOperation`` and ``Pipeline`` (reference interface: R/pipeline.py:11-173).

Construction rules, call semantics, error messages and dtype handling are the reference's (its tests match on the
messages); the implementation is this package's own.  One addition: when every operation of a pipeline is a
*device operator* (a function of this package marked with ``device_operator``), a 2-D image is uploaded once, the
operators are chained on the GPU without intermediate host copies, and the result is downloaded once.  Arbitrary
callables keep working as in the reference (they receive and return numpy arrays).
"""
from __future__ import annotations

import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_DEVICE_FLAG = "_amt_device_operator"
_DEVICE_DTYPES = (np.dtype(np.uint16), np.dtype(np.float64))


def device_operator(func):
    """Mark ``func(intensities, *args, **kwargs)`` as accepting and returning ``DeviceArray`` as well as numpy."""
    setattr(func, _DEVICE_FLAG, True)
    return func


def is_device_operator(func) -> bool:
    return getattr(func, _DEVICE_FLAG, False) is True


class ImageOperation:
    """``func`` with its trailing arguments bound: calling the operation with an image evaluates
    ``func(image, *args, **kwargs)`` (R/pipeline.py:11-60).  Frozen after construction, hashable, and equal to
    another operation with the same function and arguments."""

    __slots__ = ("func", "args", "kwargs")
    _FROZEN = "ImageOperation instances are immutable"

    def __init__(self, func, *args, **kwargs):
        for name, value in (("func", func), ("args", args), ("kwargs", kwargs)):
            object.__setattr__(self, name, value)  # the only writes these slots ever see

    def __setattr__(self, name, value):
        raise AttributeError(self._FROZEN)

    def __delattr__(self, name):
        raise AttributeError(self._FROZEN)

    def _identity(self):
        return self.func, self.args, self.kwargs

    def __call__(self, intensities):
        return self.func(intensities, *self.args, **self.kwargs)

    def __eq__(self, other):
        if isinstance(other, ImageOperation):
            return self._identity() == other._identity()
        return NotImplemented

    def __hash__(self):
        return hash((self.func, self.args, tuple(sorted(self.kwargs.items()))))

    def __repr__(self):
        shown = [repr(a) for a in self.args]
        shown.extend(f"{key}={value!r}" for key, value in self.kwargs.items())
        return "{}({})".format(self.func.__name__, ", ".join(shown))

    @property
    def on_device(self) -> bool:
        return is_device_operator(self.func)


class Pipeline:
    """Operations applied one after the other (R/pipeline.py:63-173).

    ``operations``   the callables, in order (a tuple is accepted and stored as a list);
    ``copy``         work on a copy of the input (meaningless with ``parallel``: that mode always builds a new array);
    ``preserve_dtype``  cast the result back to the dtype of the input;
    ``parallel``     treat axis 0 as a stack of independent images and map the operations over it with a thread pool
                     (the input must have at least three axes); every worker thread drives the GPU through its own
                     context / HIP stream, so slices overlap on the device;
    ``max_workers``  size of that pool (``None``: the executor's default).
    """

    _OPTIONS = ("copy", "preserve_dtype", "parallel", "max_workers")

    def __init__(self, operations, copy=False, preserve_dtype=False, parallel=False, max_workers=None):
        self.operations = list(operations) if isinstance(operations, tuple) else operations
        self.copy = copy
        self.preserve_dtype = preserve_dtype
        self.parallel = parallel
        self.max_workers = max_workers
        self._check()

    def _check(self):
        if not self.operations:
            raise ValueError("Pipeline must have at least one operation")
        for op in self.operations:
            if not callable(op):
                raise TypeError("All operations must be callable (wrap functions with ImageOperation)")
        if self.max_workers is not None and self.max_workers < 1:
            raise ValueError(f"max_workers must be at least 1, got {self.max_workers}")
        if self.copy and self.parallel:
            warnings.warn("copy=True has no effect when parallel=True. Parallel mode always produces a new output array.",
                          UserWarning, stacklevel=3)

    # ---- one image (or one slice of the stack) ------------------------------------------------------------------
    def _device_chain_applies(self, image) -> bool:
        return (
            isinstance(image, np.ndarray) and image.ndim == 2 and image.size > 0 and image.dtype in _DEVICE_DTYPES
            and all(isinstance(op, ImageOperation) and op.on_device for op in self.operations)
        )

    def _apply_operations(self, intensities):
        image = intensities.copy() if self.copy else intensities
        if self._device_chain_applies(image):
            from .device import DeviceArray, get_context

            resident = get_context().asarray(image)  # one upload ...
            for op in self.operations:
                resident = op(resident)
            return resident.numpy() if isinstance(resident, DeviceArray) else resident  # ... one download
        for op in self.operations:
            image = op(image)
        return image

    # ---- the whole input ---------------------------------------------------------------------------------------
    def _map_over_axis0(self, stack):
        if stack.ndim < 3:
            raise ValueError(
                f"Parallel mode requires at least 3D input (got {stack.ndim}D). "
                "The first axis is used to distribute work across threads."
            )
        with ThreadPoolExecutor(max_workers=self.max_workers) as pool:
            slices = list(pool.map(self._apply_operations, stack))
        return np.array(slices, dtype=stack.dtype) if self.preserve_dtype else np.array(slices)

    def __call__(self, intensities):
        if self.parallel:
            return self._map_over_axis0(intensities)
        out = self._apply_operations(intensities)
        if self.preserve_dtype and out.dtype != intensities.dtype:
            out = out.astype(intensities.dtype)
        return out

    def __len__(self):
        return len(self.operations)

    def __eq__(self, other):
        if not isinstance(other, Pipeline):
            return NotImplemented
        return self.operations == other.operations and all(
            getattr(self, name) == getattr(other, name) for name in self._OPTIONS)

    __hash__ = None  # mutable, like the reference's dataclass

    def __repr__(self):
        body = ", ".join(map(repr, self.operations))
        flags = [f"{name}=True" for name in self._OPTIONS[:3] if getattr(self, name)]
        if self.max_workers is not None:
            flags.append(f"max_workers={self.max_workers}")
        return "Pipeline([" + body + "]" + "".join(", " + f for f in flags) + ")"
