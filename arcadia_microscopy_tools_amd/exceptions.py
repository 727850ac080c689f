"""Warning categories under the reference's names (R/exceptions.py), so that callers' warning filters keep working."""


class MetadataWarning(UserWarning):
    """Raised as a warning when acquisition metadata is missing or contradictory and a default stands in for it."""


class SegmentationWarning(UserWarning):
    """Raised as a warning when segmenting one image of a batch failed and its slot in the result is empty."""
