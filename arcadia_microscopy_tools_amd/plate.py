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


class PlateLayout:
    """Byte layout of the per-plate feature blocks and of what the all-gather leaves behind (device independent,
    so the world_size-2 gloo test on CPU exercises exactly what the RCCL path uses).

    One block = one batch of B fields of view of one rank: [B*K*14 float64 | B*K*C*4 float64 | B int32, padded to
    8 bytes].  A rank's buffer holds [step][block]; the gathered buffer holds [step][rank][block]."""

    def __init__(self, steps: int, world: int, B: int, K: int, C: int):
        self.steps, self.world, self.B, self.K, self.C = int(steps), int(world), int(B), int(K), int(C)
        self.n_table = self.B * self.K * _hip.RP_NCOLS * 8
        self.n_itable = self.B * self.K * self.C * 4 * 8
        self.n_cells = (self.B * 4 + 7) // 8 * 8
        self.step_bytes = self.n_table + self.n_itable + self.n_cells
        self.local_bytes = self.steps * self.step_bytes
        self.gathered_bytes = self.steps * self.world * self.step_bytes

    def block_offsets(self, slot: int, b0: int = 0):
        """Byte offsets, inside a rank's buffer, of the table / intensity table / cell counts of FOV b0 of a block."""
        base = slot * self.step_bytes
        return (base + b0 * self.K * _hip.RP_NCOLS * 8,
                base + self.n_table + b0 * self.K * self.C * 4 * 8,
                base + self.n_table + self.n_itable + b0 * 4)

    def gather_step(self, local, gathered, slot: int, group=None):
        """The plate's ONE collective: all-gather block `slot` of every rank (uint8 torch tensors)."""
        import torch.distributed as dist

        lo = slot * self.step_bytes
        go = slot * self.world * self.step_bytes
        dist.all_gather_into_tensor(gathered[go: go + self.world * self.step_bytes],
                                    local[lo: lo + self.step_bytes], group=group)

    def unpack(self, gathered):
        """(table, itable, ncells) with leading axes (rank, step * B + fov) from a gathered uint8 tensor."""
        import torch

        S, Wd, B, K, C = self.steps, self.world, self.B, self.K, self.C
        g = gathered.view(S, Wd, self.step_bytes)
        t = g[:, :, : self.n_table].contiguous().view(torch.float64).view(S, Wd, B, K, _hip.RP_NCOLS)
        it = g[:, :, self.n_table: self.n_table + self.n_itable].contiguous().view(torch.float64).view(S, Wd, B, K, C, 4)
        nc = g[:, :, self.n_table + self.n_itable: self.n_table + self.n_itable + B * 4].contiguous().view(
            torch.int32).view(S, Wd, B)
        t = t.permute(1, 0, 2, 3, 4).reshape(Wd, S * B, K, _hip.RP_NCOLS)
        it = it.permute(1, 0, 2, 3, 4, 5).reshape(Wd, S * B, K, C, 4)
        nc = nc.permute(1, 0, 2).reshape(Wd, S * B)
        return t, it, nc


class PlateTables:
    """The per-rank blocks of the per-plate feature tables, and the ONE all-gather each plate gets.

    A rank processes its shard in `steps` batches; a batch of all ranks together is one plate (world x B fields of
    view: 384 at 2 GPUs with the bench's B = 192).  The batch's segmenters write their morphology table, intensity
    table and cell counts straight into the batch's slice of one torch-allocated byte buffer (no copies); when the
    batch's kernels have been enqueued, a single ``all_gather_into_tensor`` over RCCL for that plate is enqueued
    on a side stream, ordered after the compute streams with events, so it travels over xGMI while the next batch
    computes (BASELINE.json north_star: "RCCL all-gather over xGMI only for the final per-plate label/feature
    table").  Layout of one batch block (bytes): [B*K*14 float64 | B*K*C*4 float64 | B int32 (padded to 8 bytes)];
    the gathered buffer holds [step][rank][block]."""

    def __init__(self, segs, steps, torch_device, group=None):
        import torch
        import torch.distributed as dist

        from .device import Context

        self.torch, self.dist, self.group = torch, dist, group
        self.segs = list(segs)
        self.steps = int(steps)
        B = sum(s.B for s in self.segs)
        K, C = self.segs[0].max_cells, self.segs[0].C
        self.B, self.K, self.C = B, K, C
        self.world = dist.get_world_size(group)
        self.layout = PlateLayout(self.steps, self.world, B, K, C)
        self.step_bytes = self.layout.step_bytes
        self.local = torch.zeros(self.layout.local_bytes, dtype=torch.uint8, device=torch_device)
        self.gathered = torch.empty(self.layout.gathered_bytes, dtype=torch.uint8, device=torch_device)
        self.done = [False] * self.steps
        # the collective's stream is one of the library's own HIP streams handed to torch as an external stream
        self.gctx = Context(self.segs[0].ctx.device)
        self.stream = torch.cuda.ExternalStream(self.gctx.stream_ptr, device=torch_device)

    def point(self, step: int):
        """Make the segmenters write the tables of batch `step` (0-based, modulo `steps`) into that batch's block."""
        from .device import DeviceArray

        slot = step % self.steps
        ptr = self.local.data_ptr()
        K, C = self.K, self.C
        if self.done[slot]:  # an earlier exchange of this block may still be reading it
            for s in self.segs:
                s.ctx.wait_for(self.gctx)
        b0 = 0
        for s in self.segs:
            o_t, o_i, o_c = self.layout.block_offsets(slot, b0)
            s.table = DeviceArray(s.ctx, ptr + o_t, (s.B, K, _hip.RP_NCOLS), np.float64)
            s.itable = DeviceArray(s.ctx, ptr + o_i, (s.B, K, C, 4), np.float64)
            s.ncells = DeviceArray(s.ctx, ptr + o_c, (s.B,), np.int32)
            b0 += s.B
        self.done[slot] = False

    def gather_step(self, step: int):
        """Enqueue the all-gather of batch `step` after everything the compute streams have been given so far."""
        slot = step % self.steps
        for s in self.segs:
            self.gctx.wait_for(s.ctx)
        with self.torch.cuda.stream(self.stream):
            self.layout.gather_step(self.local, self.gathered, slot, self.group)
        self.done[slot] = True

    def all_gather(self):
        """Enqueue the all-gather of every batch that has not been exchanged yet; returns the gathered buffer."""
        for slot in range(self.steps):
            if not self.done[slot]:
                self.gather_step(slot)
        return self.gathered

    def result(self):
        """(table, itable, ncells) of everything gathered: leading axes (rank, step * B + fov)."""
        self.stream.synchronize()
        return self.layout.unpack(self.gathered)


def well_id(fov_index: int, n_columns: int = 24, fovs_per_well: int = 1) -> str:
    """Well of a field of view for row-major acquisition: FOV 0 -> "A01" ... FOV 383 -> "P24" on a 384-well
    plate; same normalised form as the reference's ``Well.id`` (capital row letter + two-digit
    column, R/microplate.py:24-45, columns 1..48, rows A..Z)."""
    if fov_index < 0 or fovs_per_well <= 0 or not 1 <= n_columns <= 48:
        raise ValueError(f"bad plate geometry: fov {fov_index}, {n_columns} columns, {fovs_per_well} FOVs per well")
    row, col = divmod(int(fov_index) // fovs_per_well, n_columns)
    if row >= 26:
        raise ValueError(f"FOV {fov_index} is beyond row Z of a {n_columns}-column plate")
    return f"{chr(ord('A') + row)}{col + 1:02d}"


def plate_rows(table, itable, ncells, channel_names, fov_indices=None) -> np.ndarray:
    """Dense per-plate blocks (``PlateTables.result()`` with the leading (rank, FOV) axes flattened) -> one packed
    (total cells, ncols) float64 table in ``table_columns`` order, rows ordered by FOV then label; derived columns
    (circularity, volume) exactly as ``cell_properties`` derives them (segment.assemble_cell_properties).

    table (F, K, RP_NCOLS), itable (F, K, C, 4), ncells (F,): numpy arrays; fov_indices defaults to 0..F-1."""
    from .segment import assemble_cell_properties

    table, itable, ncells = np.asarray(table), np.asarray(itable), np.asarray(ncells)
    F, K, ncol = table.shape
    channel_names = list(channel_names)
    if ncol != _hip.RP_NCOLS or itable.shape != (F, K, len(channel_names), 4) or ncells.shape != (F,):
        raise ValueError("plate_rows: inconsistent block shapes")
    if (ncells < 0).any() or (ncells > K).any():
        raise ValueError("plate_rows: a field of view overflowed its max_cells slot")
    fov_indices = list(range(F)) if fov_indices is None else [int(i) for i in fov_indices]
    tables = [assemble_cell_properties(table[f, : ncells[f]], itable[f, : ncells[f]], channel_names)
              for f in range(F)]
    return pack_rows(fov_indices, tables, channel_names)


def plate_dataframe(rows: np.ndarray, channel_names, n_columns: int = 24, fovs_per_well: int = 1, layout=None):
    """Packed plate table -> pandas DataFrame keyed by well: columns ``well_id``, ``fov_index``, ``label``, the
    morphology and per-channel intensity columns of ``table_columns``; with ``layout`` (any mapping
    well id -> object with ``sample`` / ``properties``, e.g. the reference's ``MicroplateLayout``,
    R/microplate.py:94-183) the well's sample and properties are joined as extra columns."""
    import pandas as pd

    cols = table_columns(channel_names)
    rows = np.asarray(rows, dtype=np.float64)
    if rows.ndim != 2 or rows.shape[1] != len(cols):
        raise ValueError(f"expected a (rows, {len(cols)}) table, got {rows.shape}")
    df = pd.DataFrame(rows, columns=cols)
    df["fov_index"] = df["fov_index"].astype(np.int64)
    df["label"] = df["label"].astype(np.int64)
    df.insert(0, "well_id", [well_id(i, n_columns, fovs_per_well) for i in df["fov_index"]])
    if layout is not None:
        samples, props = [], {}
        for n, w in enumerate(df["well_id"]):
            well = layout[w] if w in layout else None
            samples.append(getattr(well, "sample", "") if well is not None else "")
            for k, v in (getattr(well, "properties", {}) or {}).items() if well is not None else ():
                props.setdefault(k, [None] * len(df))[n] = v
        df["sample"] = samples
        for k, v in props.items():
            df[k] = v
    return df
