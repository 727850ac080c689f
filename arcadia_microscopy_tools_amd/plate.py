"""Plate-level sharding: independent fields of view over the GPUs of one node + one all-gather.

Fields of view are independent units (the reference's own unit of parallel work is "slice of axis 0",
R/pipeline.py:145-146, and "image in batch", R/model.py:276), so a plate is sharded by FOV index with
NO collective on the data path.  The only exchange is the final per-plate feature table: one
all-gather of row counts and one of row-padded float64 blocks over RCCL (torch.distributed backend
"nccl" on ROCm; "gloo" in the CPU tests).  Label images stay on the GPU that produced them.
"""
from __future__ import annotations

import numpy as np

from . import _hip

# packed feature-table columns: ids, morphology (R/masks.py:15-28 after the reference's renames),
# then {mean, max, min, std} per channel
ID_COLS = ("fov_index", "label")
MORPH_COLS = (
    "centroid_y", "centroid_x", "volume", "area", "area_convex", "perimeter", "eccentricity", "circularity",
    "solidity", "axis_major_length", "axis_minor_length", "orientation",
)
INTENSITY_STATS = ("intensity_mean", "intensity_max", "intensity_min", "intensity_std")


def table_columns(channel_names) -> list[str]:
    cols = list(ID_COLS) + list(MORPH_COLS)
    for ch in channel_names:
        for s in INTENSITY_STATS:
            cols.append(f"{s}_{str(ch).lower()}")
    return cols


def shard_indices(n_items: int, rank: int, world: int) -> list[int]:
    """Contiguous block partition of FOV indices 0..n_items-1 (blocks differ by at most one item)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def pack_rows(fov_indices, tables, channel_names) -> np.ndarray:
    """Per-FOV ``cell_properties`` dicts -> one (rows, ncols) float64 block in ``table_columns`` order."""
    cols = table_columns(channel_names)
    blocks = []
    for idx, t in zip(fov_indices, tables):
        k = len(t["label"]) if "label" in t else len(next(iter(t.values())))
        blk = np.empty((k, len(cols)), dtype=np.float64)
        blk[:, 0] = idx
        for j, c in enumerate(cols[1:], start=1):
            blk[:, j] = t[c]
        blocks.append(blk)
    if not blocks:
        return np.zeros((0, len(cols)), dtype=np.float64)
    return np.concatenate(blocks, axis=0)


def all_gather_rows(local_rows, group=None):
    """All-gather row blocks of different lengths: returns the (sum rows, ncols) table on every rank.

    ``local_rows`` is a 2-D float64 torch tensor (CUDA under nccl/RCCL, CPU under gloo).  Two collectives:
    row counts, then blocks padded to the maximum count (an exact all-gather-v)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if local_rows.dim() != 2:
        raise ValueError("local_rows must be 2-D")
    ncols = local_rows.shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=local_rows.device)
    mine = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    max_rows = int(counts.max().item())
    padded = torch.zeros((max_rows, ncols), dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    gathered = torch.empty((world * max_rows, ncols), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    gathered = gathered.view(world, max_rows, ncols)
    parts = [gathered[r, : int(counts[r].item())] for r in range(world)]
    return torch.cat(parts, dim=0), counts


class DevicePackedTables:
    """Device-resident packed blocks for the bench / plate loop: the segmenter's raw tables
    (B, max_cells, 14) + (B, max_cells, C, 4) + counts are gathered as they are (fixed shape, so a single
    all_gather_into_tensor per array and no host round trip inside the timed region)."""

    def __init__(self, seg, torch_device):
        import torch

        self.seg = seg
        self.torch = torch
        B, K, C = seg.B, seg.max_cells, seg.C
        self.table = torch.empty((B, K, _hip.RP_NCOLS), dtype=torch.float64, device=torch_device)
        self.itable = torch.empty((B, K, C, 4), dtype=torch.float64, device=torch_device)
        self.ncells = torch.empty((B,), dtype=torch.int32, device=torch_device)

    def adopt(self):
        """Point the segmenter's output tables at the torch allocations (so RCCL can send them)."""
        from .device import DeviceArray

        s = self.seg
        s.table = DeviceArray(s.ctx, self.table.data_ptr(), tuple(self.table.shape), np.float64)
        s.itable = DeviceArray(s.ctx, self.itable.data_ptr(), tuple(self.itable.shape), np.float64)
        s.ncells = DeviceArray(s.ctx, self.ncells.data_ptr(), tuple(self.ncells.shape), np.int32)

    def all_gather(self, group=None):
        import torch.distributed as dist

        torch = self.torch
        world = dist.get_world_size(group)
        g_table = torch.empty((world,) + tuple(self.table.shape), dtype=self.table.dtype, device=self.table.device)
        g_itable = torch.empty((world,) + tuple(self.itable.shape), dtype=self.itable.dtype, device=self.table.device)
        g_n = torch.empty((world,) + tuple(self.ncells.shape), dtype=self.ncells.dtype, device=self.table.device)
        dist.all_gather_into_tensor(g_n, self.ncells, group=group)
        dist.all_gather_into_tensor(g_table, self.table, group=group)
        dist.all_gather_into_tensor(g_itable, self.itable, group=group)
        return g_table, g_itable, g_n
