"""The operator boundary: ``ImageOperation`` and ``Pipeline`` (reference: R/pipeline.py:11-173).

Same construction rules, call semantics, error messages and dtype handling as the reference; this is
the seam the HIP operators sit behind.  One addition: when every operation of a pipeline is a *device
operator* (a function of this package marked with ``device_operator``), the pipeline uploads the image
once, chains the operators on the GPU without intermediate host copies, and downloads the result once.
Arbitrary callables keep working exactly as in the reference (they receive and return numpy arrays).
"""
from __future__ import annotations

import warnings
from collections.abc import Callable
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from .typing import ScalarArray


def device_operator(func):
    """Mark ``func(intensities, *args, **kwargs)`` as accepting and returning ``DeviceArray`` as well as numpy."""
    func._amt_device_operator = True
    return func


def is_device_operator(func) -> bool:
    return bool(getattr(func, "_amt_device_operator", False))


class ImageOperation:
    """An immutable, hashable ``func(intensities, *args, **kwargs)`` closure (R/pipeline.py:11-60)."""

    __slots__ = ("func", "args", "kwargs")

    def __init__(self, func: Callable[..., ScalarArray], *args: object, **kwargs: object) -> None:
        object.__setattr__(self, "func", func)
        object.__setattr__(self, "args", args)
        object.__setattr__(self, "kwargs", kwargs)

    def __setattr__(self, name: str, value: object) -> None:
        raise AttributeError("ImageOperation instances are immutable")

    def __delattr__(self, name: str) -> None:
        raise AttributeError("ImageOperation instances are immutable")

    def __call__(self, intensities: ScalarArray) -> ScalarArray:
        return self.func(intensities, *self.args, **self.kwargs)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, ImageOperation):
            return NotImplemented
        return (self.func, self.args, self.kwargs) == (other.func, other.args, other.kwargs)

    def __hash__(self) -> int:
        return hash((self.func, self.args, tuple(sorted(self.kwargs.items()))))

    def __repr__(self) -> str:
        parts = [repr(a) for a in self.args] + [f"{k}={v!r}" for k, v in self.kwargs.items()]
        return f"{self.func.__name__}({', '.join(parts)})"

    @property
    def on_device(self) -> bool:
        return is_device_operator(self.func)


@dataclass
class Pipeline:
    """A sequence of image operations applied in order (R/pipeline.py:63-173).

    Attributes mirror the reference: ``operations``, ``copy`` (copy the input first; ignored when
    ``parallel``), ``preserve_dtype`` (cast the result back to the input dtype), ``parallel`` (map the
    operations over the slices of axis 0 with a thread pool; needs >= 3-D input), ``max_workers``.
    Each worker thread drives the GPU through its own context / HIP stream, so slices overlap on the device.
    """

    operations: list[ImageOperation]
    copy: bool = False
    preserve_dtype: bool = False
    parallel: bool = False
    max_workers: int | None = None

    def __post_init__(self) -> None:
        if isinstance(self.operations, tuple):
            self.operations = list(self.operations)
        if not self.operations:
            raise ValueError("Pipeline must have at least one operation")
        if not all(callable(op) for op in self.operations):
            raise TypeError("All operations must be callable (wrap functions with ImageOperation)")
        if self.max_workers is not None and self.max_workers < 1:
            raise ValueError(f"max_workers must be at least 1, got {self.max_workers}")
        if self.parallel and self.copy:
            warnings.warn(
                "copy=True has no effect when parallel=True. "
                "Parallel mode always produces a new output array.",
                UserWarning,
                stacklevel=2,
            )

    # ---------------------------------------------------------------------------------------------
    def _all_on_device(self) -> bool:
        return all(isinstance(op, ImageOperation) and op.on_device for op in self.operations)

    def _apply_operations(self, intensities: ScalarArray) -> ScalarArray:
        """Apply all operations to one array (a whole image, or one slice in parallel mode)."""
        from .device import DeviceArray, get_context

        out = intensities.copy() if self.copy else intensities
        if (
            isinstance(out, np.ndarray)
            and out.ndim == 2
            and out.size > 0
            and out.dtype in (np.uint16, np.float64)
            and self._all_on_device()
        ):
            # device-resident chain: one upload, one download
            dev = get_context().asarray(out)
            for operation in self.operations:
                dev = operation(dev)
            return dev.numpy() if isinstance(dev, DeviceArray) else dev
        for operation in self.operations:
            out = operation(out)
        return out

    def __call__(self, intensities: ScalarArray) -> ScalarArray:
        if self.parallel:
            if intensities.ndim < 3:
                raise ValueError(
                    f"Parallel mode requires at least 3D input (got {intensities.ndim}D). "
                    "The first axis is used to distribute work across threads."
                )
            with ThreadPoolExecutor(max_workers=self.max_workers) as executor:
                processed = list(executor.map(self._apply_operations, intensities))
            if self.preserve_dtype:
                return np.array(processed, dtype=intensities.dtype)
            return np.array(processed)

        result = self._apply_operations(intensities)
        if self.preserve_dtype and result.dtype != intensities.dtype:
            return result.astype(intensities.dtype)
        return result

    def __len__(self) -> int:
        return len(self.operations)

    def __repr__(self) -> str:
        ops = ", ".join(repr(op) for op in self.operations)
        params = []
        if self.copy:
            params.append("copy=True")
        if self.preserve_dtype:
            params.append("preserve_dtype=True")
        if self.parallel:
            params.append("parallel=True")
        if self.max_workers is not None:
            params.append(f"max_workers={self.max_workers}")
        tail = f", {', '.join(params)}" if params else ""
        return f"Pipeline([{ops}]{tail})"
