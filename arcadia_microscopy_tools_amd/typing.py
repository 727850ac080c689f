"""Array type aliases under the names the reference exports (R/typing.py): the dtype contract of the hot path --
bool masks, uint8 / uint16 intensities, int64 labels, float32 / float64 results."""
from typing import Union

import numpy as np
import numpy.typing as npt


def _array_of(scalar_type):
    return npt.NDArray[scalar_type]


BoolArray, UByteArray, UInt16Array = _array_of(np.bool_), _array_of(np.uint8), _array_of(np.uint16)
Int64Array = _array_of(np.int64)
Float32Array, Float64Array = _array_of(np.float32), _array_of(np.float64)

# whatever an ImageOperation may receive or return
ScalarArray = Union[BoolArray, UByteArray, UInt16Array, Int64Array, Float32Array, Float64Array]
