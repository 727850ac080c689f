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


def packed_ncols(C: int) -> int:
    """Columns of a packed per-cell row: fov index, label, the RP_NCOLS morphology columns, C x {mean, max, min, std}."""
    return 2 + _hip.RP_NCOLS + 4 * int(C)


class RowExchange:
    """The plate's exchange, counts first, then rows -- an exact all-gather-v of the per-rank
    blocks of packed feature rows, pipelined so that the host never waits for the step it has just enqueued.

    ``submit(rows, nrows)`` enqueues the all-gather of the row COUNT of one block (world x int64) and its copy to
    page-locked host memory; one block later (``lag``), when that small exchange has long finished, the host reads
    the counts, and every rank enqueues ONE ``all_gather_into_tensor`` of ``max(counts)`` rows -- the bytes that
    travel are the rows that exist (~1,360 of the 2,048 a field of view may hold), not the dense tables.
    Device independent: the same code runs over RCCL on HIP streams (``stream`` = the side stream the collectives
    are enqueued on) and over gloo on CPU tensors (tests/test_distributed_gloo.py, world_size 2)."""

    def __init__(self, ncols: int, device, group=None, stream=None, lag: int = 1, keep: int | None = None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ncols, self.device, self.stream, self.lag = int(ncols), torch.device(device), stream, int(lag)
        self.on_gpu = self.device.type == "cuda"
        # rehearsal on a single GPU: two ranks share the card and exchange through gloo, staged on the host
        self.host_staged = self.on_gpu and dist.get_backend(group) == "gloo"
        self.pending: list[dict] = []
        # (gathered (world, max_rows, ncols), counts list) per block; `keep` bounds how many stay referenced (a
        # consumer that writes each plate out as it arrives needs only the newest few; None keeps all)
        self.finished: list[tuple] = []
        self.keep = keep
        self.n_finished = 0
        # page-locked landing buffers for the counts, allocated once (hipHostMalloc / hipHostFree per step would
        # stall the queues)
        self._host_ring = ([torch.empty(self.world, dtype=torch.int64).pin_memory() for _ in range(self.lag + 2)]
                           if self.on_gpu else None)
        self._host_next = 0

    def _stream_ctx(self):
        import contextlib

        if self.on_gpu and self.stream is not None:
            return self.torch.cuda.stream(self.stream)
        return contextlib.nullcontext()

    def _all_gather(self, out, inp):
        if self.host_staged:
            if self.stream is not None:
                self.stream.synchronize()
            h_out = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(h_out, inp.cpu(), group=self.group)
            out.copy_(h_out)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def submit(self, rows, nrows):
        """rows: (cap, ncols) float64 tensor of which the first ``nrows[0]`` rows are this rank's block;
        nrows: (1,) int64 tensor on the same device (written by the pack kernel: the host does not know it yet)."""
        torch = self.torch
        if rows.dim() != 2 or rows.shape[1] != self.ncols or rows.dtype != torch.float64 or not rows.is_contiguous():
            raise ValueError(f"rows must be a contiguous (cap, {self.ncols}) float64 tensor")
        with self._stream_ctx():
            counts = torch.empty(self.world, dtype=torch.int64, device=self.device)
            self._all_gather(counts, nrows.reshape(1))
            event = None
            if self.on_gpu:
                host = self._host_ring[self._host_next % len(self._host_ring)]
                self._host_next += 1
                host.copy_(counts, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self.stream if self.stream is not None else torch.cuda.current_stream(self.device))
            else:
                host = counts
        self.pending.append(dict(rows=rows, host=host, event=event, issued=False))
        self.pump(self.lag)

    def _issue(self, e):
        torch = self.torch
        if e["event"] is not None:
            e["event"].synchronize()
        counts = [int(v) for v in e["host"].tolist()]
        if min(counts) < 0:
            raise RuntimeError(f"a rank reported an overflowed feature table (row counts {counts}); raise max_cells")
        max_rows = max(counts)
        rows = e["rows"]
        with self._stream_ctx():
            if max_rows > rows.shape[0]:  # another rank holds more rows than this rank's buffer: pad a copy
                padded = torch.zeros((max_rows, self.ncols), dtype=rows.dtype, device=rows.device)
                padded[: rows.shape[0]] = rows
                rows = padded
            gathered = torch.empty((self.world, max_rows, self.ncols), dtype=torch.float64, device=self.device)
            if max_rows > 0:
                self._all_gather(gathered.view(-1), rows[:max_rows].reshape(-1))
        self.finished.append((gathered, counts))
        self.n_finished += 1
        if self.keep is not None and len(self.finished) > self.keep:
            del self.finished[: len(self.finished) - self.keep]
        e["issued"] = True
        e["rows"] = None

    def pump(self, keep: int):
        """Issue the row all-gather of every submitted block except the newest ``keep``."""
        todo = [e for e in self.pending if not e["issued"]]
        for e in todo[: max(0, len(todo) - keep)]:
            self._issue(e)
        self.pending = [e for e in self.pending if not e["issued"]]

    def flush(self):
        self.pump(0)

    def results(self):
        """[(rows (sum counts, ncols) tensor ordered by rank, counts list)] per submitted block, in order."""
        self.flush()
        if self.on_gpu and self.stream is not None:
            self.stream.synchronize()
        out = []
        for gathered, counts in self.finished:
            parts = [gathered[r, : counts[r]] for r in range(self.world)]
            out.append((self.torch.cat(parts, dim=0) if parts else gathered.reshape(0, self.ncols), counts))
        return out


class PlateTables:
    """The per-rank blocks of the per-plate feature tables, and the ONE row exchange each plate gets.

    A rank processes its shard in steps; a step of all ranks together is one plate (BASELINE configs[3]: 384
    fields of view, 48 per GPU at 8 GPUs).  The step's segmenters write their dense tables (B x K rows) and cell
    counts straight into a staging block (a ring of ``slots`` blocks); when the step's kernels have been enqueued,
    a side stream -- ordered after the compute streams with events -- compacts the block to one row per cell
    (``amt_pack_plate_rows``) and hands it to ``RowExchange``: row counts first, then ONE RCCL all-gather of the rows
    that exist, travelling over xGMI while the next step computes (BASELINE.json north_star: "RCCL all-gather over
    xGMI only for the final per-plate label/feature table").  Ranks may hold different numbers of fields of view."""

    def __init__(self, segs, torch_device, group=None, slots: int = 4, cap_fovs: int | None = None,
                 keep: int | None = None):
        import torch
        import torch.distributed as dist

        from .device import Context, DeviceArray

        self.torch, self.dist, self.group = torch, dist, group
        self.segs = list(segs)
        B = sum(s.B for s in self.segs)
        K, C = self.segs[0].max_cells, self.segs[0].C
        if any(s.max_cells != K or s.C != C for s in self.segs):
            raise ValueError("all segmenters of a rank must share max_cells and the channel count")
        self.B, self.K, self.C = B, K, C
        self.ncols = packed_ncols(C)
        self.slots = max(2, int(slots))
        cap_fovs = B if cap_fovs is None else max(int(cap_fovs), B)
        # staging ring: [slot][B*K*RP_NCOLS f64 | B*K*C*4 f64 | B int32]
        self.n_table = B * K * _hip.RP_NCOLS * 8
        self.n_itable = B * K * C * 4 * 8
        self.n_cells = (B * 4 + 7) // 8 * 8
        self.slot_bytes = self.n_table + self.n_itable + self.n_cells
        self.staging = torch.zeros(self.slots * self.slot_bytes, dtype=torch.uint8, device=torch_device)
        # packed rows: one buffer per slot (a block's rows must stay put until its lagged all-gather has run)
        self.rows = [torch.empty((max(cap_fovs * K, 1), self.ncols), dtype=torch.float64, device=torch_device)
                     for _ in range(self.slots)]
        self.nrows = [torch.zeros(1, dtype=torch.int64, device=torch_device) for _ in range(self.slots)]
        # the side stream is one of the library's own HIP streams, handed to torch as an external stream
        self.gctx = Context(self.segs[0].ctx.device)
        self.stream = torch.cuda.ExternalStream(self.gctx.stream_ptr, device=torch_device)
        self.exchange = RowExchange(self.ncols, torch_device, group, stream=self.stream, lag=1, keep=keep)
        self._DeviceArray = DeviceArray
        self.used = [False] * self.slots
        self.packed_ev = [self.gctx.event() for _ in range(self.slots)]  # "the pack kernel has read this block"
        torch.cuda.synchronize(torch_device)  # torch filled the buffers on ITS stream; the library's streams do not wait for it

    def _offsets(self, slot: int, b0: int):
        base = slot * self.slot_bytes
        return (base + b0 * self.K * _hip.RP_NCOLS * 8,
                base + self.n_table + b0 * self.K * self.C * 4 * 8,
                base + self.n_table + self.n_itable + b0 * 4)

    def point(self, step: int):
        """Make the segmenters write the tables of step `step` into that step's staging block."""
        DeviceArray = self._DeviceArray
        slot = step % self.slots
        ptr = self.staging.data_ptr()
        K, C = self.K, self.C
        if self.used[slot]:  # the pack kernel of the block's previous use (`slots` steps ago) may still be reading it
            for s in self.segs:
                self.packed_ev[slot].wait(s.ctx)
        b0 = 0
        for s in self.segs:
            o_t, o_i, o_c = self._offsets(slot, b0)
            s.table = DeviceArray(s.ctx, ptr + o_t, (s.B, K, _hip.RP_NCOLS), np.float64)
            s.itable = DeviceArray(s.ctx, ptr + o_i, (s.B, K, C, 4), np.float64)
            s.ncells = DeviceArray(s.ctx, ptr + o_c, (s.B,), np.int32)
            b0 += s.B
        self.used[slot] = True

    def gather_step(self, step: int, fov_index0: int = 0):
        """After everything the compute streams have been given so far: pack the step's block and submit it."""
        from . import hipops

        DeviceArray = self._DeviceArray
        slot = step % self.slots
        for s in self.segs:
            self.gctx.wait_for(s.ctx)
        B, K, C = self.B, self.K, self.C
        o_t, o_i, o_c = self._offsets(slot, 0)
        ptr = self.staging.data_ptr()
        g = self.gctx
        hipops.pack_plate_rows(
            DeviceArray(g, ptr + o_t, (B, K, _hip.RP_NCOLS), np.float64),
            DeviceArray(g, ptr + o_i, (B, K, C, 4), np.float64),
            DeviceArray(g, ptr + o_c, (B,), np.int32), fov_index0=fov_index0,
            out=DeviceArray(g, self.rows[slot].data_ptr(), tuple(self.rows[slot].shape), np.float64),
            nrows_out=DeviceArray(g, self.nrows[slot].data_ptr(), (1,), np.int64))
        self.packed_ev[slot].record(g)
        self.exchange.submit(self.rows[slot], self.nrows[slot])

    def all_gather(self):
        """Issue the row exchanges that are still held back by the one-step lag."""
        self.exchange.flush()

    def result(self):
        """[(rows (cells of all ranks, ncols) float64 numpy, counts per rank)] per step, in step order (the newest
        ``keep`` steps when the exchange was built with a bound)."""
        return [(r.cpu().numpy(), c) for r, c in self.exchange.results()]


class HostTables:
    """One GPU, no collective: the per-plate feature table delivered to the HOST, inside the step that produced it.

    Every segmenter (= context = HIP stream) compacts ITS fields of view to one 256-byte row per cell with the pack
    kernel of ``PlateTables`` and copies the rows that exist into page-locked host memory -- all on its own stream:
    the row COUNT travels first (8 bytes), and ``lag`` (two) steps later, when that copy has long finished and the
    following steps are already enqueued, the host reads it and enqueues the copy of exactly that many rows.  The host
    never waits for the step it has just enqueued, and no stream ever waits for another one (measured on the 48-FOV
    plate: ordering one side stream behind the four compute streams cost 0.8 ms per 4.3 ms step before a single byte
    moved; a copy of ~4 MB in each compute stream costs < 0.1 ms of that stream).  No torch involved.  The plate's table
    is the concatenation of the segmenters' blocks, which own consecutive fields of view."""

    def __init__(self, segs, slots: int = 4, lag: int = 2):
        from .device import DeviceArray, pinned_empty

        self.segs = list(segs)
        K, C = self.segs[0].max_cells, self.segs[0].C
        if any(s.max_cells != K or s.C != C for s in self.segs):
            raise ValueError("all segmenters must share max_cells and the channel count")
        self.B, self.K, self.C = sum(s.B for s in self.segs), K, C
        self.ncols = packed_ncols(C)
        self.lag = max(1, int(lag))
        self.slots = max(self.lag + 2, int(slots))
        self._DeviceArray = DeviceArray
        self.parts = []
        b0 = 0
        for s in self.segs:
            c = s.ctx
            n_table = s.B * K * _hip.RP_NCOLS * 8
            n_itable = s.B * K * C * 4 * 8
            n_cells = (s.B * 4 + 7) // 8 * 8
            slot_bytes = n_table + n_itable + n_cells
            self.parts.append(dict(
                seg=s, b0=b0, n_table=n_table, n_itable=n_itable, slot_bytes=slot_bytes,
                staging=c.zeros((self.slots * slot_bytes,), np.uint8),
                rows=[c.empty((max(s.B * K, 1), self.ncols), np.float64) for _ in range(self.slots)],
                nrows=[c.zeros((1,), np.int64) for _ in range(self.slots)],
                pin=[(pinned_empty((max(s.B * K, 1), self.ncols), np.float64), pinned_empty((1,), np.int64))
                     for _ in range(self.slots)],
                count_ev=[c.event() for _ in range(self.slots)], rows_ev=[c.event() for _ in range(self.slots)]))
            b0 += s.B
            c.synchronize()
        self.pending: list[int] = []            # steps whose rows have not been requested yet
        self.counts: dict[int, list[int]] = {}  # step -> rows delivered per segmenter
        self.bytes_delivered = 0

    def _views(self, p, slot):
        s, K, C = p["seg"], self.K, self.C
        base = p["staging"].ptr + slot * p["slot_bytes"]
        D = self._DeviceArray
        return (D(s.ctx, base, (s.B, K, _hip.RP_NCOLS), np.float64),
                D(s.ctx, base + p["n_table"], (s.B, K, C, 4), np.float64),
                D(s.ctx, base + p["n_table"] + p["n_itable"], (s.B,), np.int32))

    def point(self, step: int):
        """Make the segmenters write the tables of step ``step`` into that step's staging block (the pack kernel of
        the block's previous use ran on the same stream: stream order protects it)."""
        slot = step % self.slots
        for p in self.parts:
            s = p["seg"]
            s.table, s.itable, s.ncells = self._views(p, slot)

    def deliver_step(self, step: int, fov_index0: int = 0):
        """After the step's kernels have been enqueued: every segmenter packs its block on its own stream and sends the
        row count to the host; the rows of the step ``lag`` steps back are requested."""
        from . import hipops

        slot = step % self.slots
        for p in self.parts:
            s = p["seg"]
            t, it, nc = self._views(p, slot)
            hipops.pack_plate_rows(t, it, nc, fov_index0=fov_index0 + p["b0"], out=p["rows"][slot], nrows_out=p["nrows"][slot])
            s.ctx.copy_to_host_async(p["pin"][slot][1].array, p["nrows"][slot])
            p["count_ev"][slot].record(s.ctx)
        self.pending.append(step)
        self._request(flush=False)

    def _request(self, flush: bool):
        """Enqueue the row copies of the pending steps: all of them with ``flush``, else all but the newest ``lag``."""
        todo = self.pending if flush else self.pending[: max(0, len(self.pending) - self.lag)]
        self.pending = self.pending[len(todo):]
        for step in todo:
            slot = step % self.slots
            counts = []
            for p in self.parts:
                p["count_ev"][slot].synchronize()
                n = int(p["pin"][slot][1].array[0])
                if n < 0:
                    raise RuntimeError("a field of view overflowed its feature table (row count -1); raise max_cells")
                p["seg"].ctx.copy_to_host_async(p["pin"][slot][0].array, p["rows"][slot], n * self.ncols * 8)
                p["rows_ev"][slot].record(p["seg"].ctx)
                counts.append(n)
                self.bytes_delivered += n * self.ncols * 8
            self.counts[step] = counts

    def flush(self):
        self._request(flush=True)

    def synchronize(self):
        for p in self.parts:
            p["seg"].ctx.synchronize()

    def rows_of(self, step: int) -> np.ndarray:
        """The delivered rows of ``step``, the segmenters' blocks one after the other (a copy out of the page-locked
        blocks, which are reused ``slots`` steps later)."""
        if step in self.pending:
            self.flush()
        slot = step % self.slots
        out = []
        for p, n in zip(self.parts, self.counts[step]):
            p["rows_ev"][slot].synchronize()
            out.append(p["pin"][slot][0].array[:n])
        return np.concatenate(out) if out else np.zeros((0, self.ncols))

    def close(self):
        self.synchronize()
        for p in self.parts:
            for rows, cnt in p["pin"]:
                rows.close()
                cnt.close()


def rows_to_table(rows: np.ndarray, channel_names) -> np.ndarray:
    """Packed rows ``[fov, label, RP columns, C x 4 intensity]`` (any order of fields of view) -> the plate table in
    ``table_columns`` order, rows sorted by field of view then label, derived columns (circularity, volume) exactly
    as ``cell_properties`` derives them (R/masks.py:292-305 via segment.assemble_cell_properties)."""
    from .segment import assemble_cell_properties

    rows = np.asarray(rows, dtype=np.float64)
    channel_names = list(channel_names)
    C = len(channel_names)
    if rows.ndim != 2 or rows.shape[1] != packed_ncols(C):
        raise ValueError(f"expected (rows, {packed_ncols(C)}) packed rows, got {rows.shape}")
    order = np.lexsort((rows[:, 1], rows[:, 0]))
    rows = rows[order]
    fovs, starts = np.unique(rows[:, 0], return_index=True)
    bounds = list(starts) + [len(rows)]
    tables, idx = [], []
    for i, f in enumerate(fovs):
        blk = rows[bounds[i]: bounds[i + 1]]
        morph = blk[:, 2: 2 + _hip.RP_NCOLS]
        inten = blk[:, 2 + _hip.RP_NCOLS:].reshape(len(blk), C, 4)
        tables.append(assemble_cell_properties(morph, inten, channel_names))
        idx.append(int(f))
    return pack_rows(idx, tables, channel_names)


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
