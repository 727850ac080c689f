"""``configure_logging`` and ``get_tqdm`` of the reference (R/utils.py:6-40); IPython is optional here."""
from __future__ import annotations

import logging


def configure_logging(verbose: bool) -> None:
    """Root logger at DEBUG (``verbose``) or INFO with the reference's line format."""
    logging.basicConfig(
        level=logging.DEBUG if verbose else logging.INFO,
        format="%(asctime)s - %(name)s - %(levelname)s :: %(message)s",
        datefmt="%Y-%m-%d %H:%M:%S",
    )


def _in_ipython() -> bool:
    try:
        from IPython import get_ipython
    except ImportError:
        return False
    return get_ipython() is not None


def get_tqdm():
    """``tqdm.notebook.tqdm`` inside IPython / Jupyter, ``tqdm.tqdm`` otherwise."""
    if _in_ipython():
        from tqdm.notebook import tqdm
    else:
        from tqdm import tqdm
    return tqdm
