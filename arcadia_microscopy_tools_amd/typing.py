"""Array type aliases of the reference (R/typing.py:1-13): the dtype contract of the hot path."""
import numpy as np
from numpy.typing import NDArray

BoolArray = NDArray[np.bool_]
UByteArray = NDArray[np.uint8]
UInt16Array = NDArray[np.uint16]
Int64Array = NDArray[np.int64]
Float32Array = NDArray[np.float32]
Float64Array = NDArray[np.float64]

ScalarArray = BoolArray | UByteArray | UInt16Array | Int64Array | Float32Array | Float64Array
