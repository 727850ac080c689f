"""Plate metadata that keys the per-plate feature table: ``Well`` and ``MicroplateLayout`` with the reference's
interface (R/microplate.py:10-251 -- same names, normalisation, exceptions and messages), so that
``plate.plate_dataframe(rows, channels, layout=MicroplateLayout.from_csv(...))`` joins sample names and well
properties onto the table the GPUs gathered.  Host-side bookkeeping only; nothing here touches the device.
"""
from __future__ import annotations

from collections.abc import Iterator, Sequence
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

_MAX_COLUMN = 48  # R/microplate.py:38-40


def normalize_well_id(well_id: str) -> str:
    """"a1" -> "A01": capital row letter A-Z + two-digit column 1-48 (R/microplate.py:24-45; the ValueErrors and
    their wording are the reference's)."""
    if not well_id or len(well_id) < 2:
        raise ValueError("Well ID must be at least 2 characters (e.g., 'A1' or 'A01')")
    letter = well_id[0].upper()
    if not "A" <= letter <= "Z":
        raise ValueError(f"Row must be A-Z, got '{letter}'")
    try:
        number = int(well_id[1:])
    except ValueError as err:
        raise ValueError(f"Could not parse column number from '{well_id}'") from err
    if not 1 <= number <= _MAX_COLUMN:
        raise ValueError(f"Column must be 1-{_MAX_COLUMN}, got {number}")
    return f"{letter}{number:02d}"


@dataclass(frozen=True)
class Well:
    """One well: normalised ``id`` ("A01"), ``sample`` name, free-form ``properties`` (R/microplate.py:10-91)."""

    id: str
    sample: str = ""
    properties: dict[str, Any] = field(default_factory=dict)

    def __post_init__(self) -> None:
        object.__setattr__(self, "id", normalize_well_id(self.id))

    @property
    def row(self) -> str:
        return self.id[0]

    @property
    def column(self) -> int:
        return int(self.id[1:])

    def __str__(self) -> str:
        return self.id

    def __repr__(self) -> str:
        tail = f", properties={self.properties!r}" if self.properties else ""
        return f"Well(id='{self.id}', sample='{self.sample}'{tail})"

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "Well":
        """A CSV record -> Well: 'well_id' (required, a string), 'sample' (optional), every other key a property
        (R/microplate.py:66-91)."""
        if "well_id" not in data:
            raise ValueError("Dictionary must contain 'well_id' key")
        wid = data["well_id"]
        if not isinstance(wid, str):
            raise ValueError(f"well_id must be a string, got {type(wid).__name__}")
        extra = {key: value for key, value in data.items() if key != "well_id" and key != "sample"}
        return cls(wid, data.get("sample", ""), extra)


@dataclass(frozen=True)
class MicroplateLayout:
    """The wells of one plate, looked up by id in either spelling ("A1" / "A01") (R/microplate.py:93-251)."""

    wells: Sequence[Well]
    _layout: dict[str, Well] = field(init=False, repr=False)

    def __post_init__(self) -> None:
        index: dict[str, Well] = {}
        for well in self.wells:
            if well.id in index:
                raise ValueError(f"Duplicate well ID: '{well.id}'")
            index[well.id] = well
        object.__setattr__(self, "_layout", index)

    @property
    def layout(self) -> dict[str, Well]:
        return self._layout

    @property
    def rows(self) -> list[str]:
        return sorted({w.row for w in self._layout.values()})

    @property
    def columns(self) -> list[int]:
        return sorted({w.column for w in self._layout.values()})

    @property
    def well_ids(self) -> list[str]:
        return sorted(self._layout)

    def __getitem__(self, well_id: str) -> Well:
        try:
            key = normalize_well_id(well_id)
        except ValueError as err:
            raise KeyError(f"Invalid well ID '{well_id}': {err}") from None
        if key not in self._layout:
            raise KeyError(f"Well ID '{well_id}' not found in plate layout.")
        return self._layout[key]

    def __len__(self) -> int:
        return len(self._layout)

    def __contains__(self, well_id: str) -> bool:
        try:
            return normalize_well_id(well_id) in self._layout
        except ValueError:
            return False

    def __iter__(self) -> Iterator[Well]:
        return iter(self._layout.values())

    @classmethod
    def from_csv(cls, csv_path: Path, **kwargs) -> "MicroplateLayout":
        """CSV with a 'well_id' column, an optional 'sample' column and any property columns; ``kwargs`` go to
        ``pandas.read_csv`` (R/microplate.py:185-212)."""
        import pandas as pd

        frame = pd.read_csv(csv_path, **kwargs)
        if frame.empty:
            raise ValueError(f"CSV file '{csv_path}' is empty")
        if "well_id" not in frame.columns:
            raise ValueError(
                f"CSV file '{csv_path}' missing required 'well_id' column. "
                f"Found columns: {list(frame.columns)}"
            )
        return cls([Well.from_dict(record) for record in frame.to_dict("records")])

    def to_dataframe(self):
        """One row per well: well_id, row, column, sample, then the properties (R/microplate.py:214-236)."""
        import pandas as pd

        if not self._layout:
            return pd.DataFrame()
        return pd.DataFrame([{"well_id": w.id, "row": w.row, "column": w.column, "sample": w.sample, **w.properties}
                             for w in self._layout.values()])

    def display(self) -> str:
        """Grid of sample names, plate rows down and columns across, '-' for absent wells
        (R/microplate.py:238-251)."""
        frame = self.to_dataframe()
        if frame.empty:
            return "Empty plate layout"
        return frame.pivot(index="row", columns="column", values="sample").fillna("-").to_string()
