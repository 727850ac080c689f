"""The N > 1 path on CPU: FOV sharding + the feature-table all-gather with the gloo backend, world_size 2
(the GPU run uses the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from arcadia_microscopy_tools_amd import plate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chans = ("DAPI", "FITC")
        cols = plate.table_columns(chans)
        mine = plate.shard_indices(5, rank, world)  # 5 FOVs over 2 ranks: 3 + 2
        tables = []
        for idx in mine:
            k = idx + 1  # FOV idx has idx+1 cells
            t = {c: np.full(k, 100.0 * idx + j, dtype=np.float64) for j, c in enumerate(cols[1:])}
            t["label"] = np.arange(1, k + 1, dtype=np.float64)
            tables.append(t)
        local = torch.from_numpy(plate.pack_rows(mine, tables, chans))
        full, counts = plate.all_gather_rows(local)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"counts{rank}.npy"), counts.numpy())
    finally:
        dist.destroy_process_group()


def test_all_gather_rows_gloo_world2(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b), "every rank must hold the same gathered table"
    counts = np.load(tmp_path / "counts0.npy")
    assert counts.tolist() == [1 + 2 + 3, 4 + 5]
    assert a.shape == (15, 2 + 12 + 8)
    # rows are ordered by rank then FOV then label; FOV index column and label column intact
    assert a[:, 0].tolist() == [0] + [1] * 2 + [2] * 3 + [3] * 4 + [4] * 5
    assert a[:6, 1].tolist() == [1, 1, 2, 1, 2, 3]
    assert a[-1, 2] == 100.0 * 4 + 1


def _exchange_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from arcadia_microscopy_tools_amd import plate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        C = 2
        ncols = plate.packed_ncols(C)
        ex = plate.RowExchange(ncols, "cpu", lag=1)
        steps = 4
        # rank 0 owns 3 fields of view, rank 1 owns 2 (a 5-FOV plate): row capacities differ between the ranks,
        # and in step 3 rank 0 holds MORE rows than rank 1's buffer can take (the padded-copy path)
        cap = (3 if rank == 0 else 2) * 4
        issued = []
        for step in range(steps):
            n = {0: [5, 0, 2, 12], 1: [3, 7, 8, 1]}[rank][step]
            rows = torch.full((cap, ncols), -1.0, dtype=torch.float64)
            for r in range(n):
                rows[r, 0] = 10 * rank + step          # fov index
                rows[r, 1] = r + 1                     # label
                rows[r, 2:] = 1000.0 * rank + 100.0 * step + r + torch.arange(ncols - 2, dtype=torch.float64) / 100.0
            ex.submit(rows, torch.tensor([n], dtype=torch.int64))
            issued.append(len(ex.finished))
        ex.flush()
        res = ex.results()
        np.savez(os.path.join(out_dir, f"ex{rank}.npz"), issued=np.array(issued),
                 **{f"rows{i}": r.numpy() for i, (r, _) in enumerate(res)},
                 **{f"counts{i}": np.array(c) for i, (_, c) in enumerate(res)})
    finally:
        dist.destroy_process_group()


def test_row_exchange_gloo_world2(tmp_path):
    """plate.RowExchange (what plate.PlateTables runs over RCCL on a side stream), world_size 2 with gloo: counts
    first, rows one step later, ranks with different capacities and row counts, an empty block, and a block larger
    than the other rank's buffer.  Every rank ends with every rank's rows of every step, ordered by rank."""
    import torch.multiprocessing as mp

    from arcadia_microscopy_tools_amd import plate

    port = _free_port()
    mp.spawn(_exchange_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ex0.npz"), np.load(tmp_path / "ex1.npz")
    ncols = plate.packed_ncols(2)
    expect = {0: [5, 0, 2, 12], 1: [3, 7, 8, 1]}
    assert a["issued"].tolist() == [0, 1, 2, 3]  # the row all-gather of a step is issued one step later
    for step in range(4):
        ra, rb = a[f"rows{step}"], b[f"rows{step}"]
        assert np.array_equal(ra, rb), f"step {step}: both ranks must hold the same table"
        counts = a[f"counts{step}"].tolist()
        assert counts == [expect[0][step], expect[1][step]]
        assert ra.shape == (sum(counts), ncols)
        off = 0
        for rank in range(2):
            for r in range(counts[rank]):
                row = ra[off + r]
                assert row[0] == 10 * rank + step and row[1] == r + 1
                assert np.array_equal(row[2:], 1000.0 * rank + 100.0 * step + r + np.arange(ncols - 2) / 100.0)
            off += counts[rank]


def test_row_exchange_overflow_is_loud(tmp_path):
    """A rank whose pack kernel flagged an overflowed table (count -1) makes every rank raise."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_overflow_worker, args=(2, port), nprocs=2, join=True)


def _overflow_worker(rank, world, port):
    import torch
    import torch.distributed as dist

    from arcadia_microscopy_tools_amd import plate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ncols = plate.packed_ncols(1)
        ex = plate.RowExchange(ncols, "cpu", lag=0)
        rows = torch.zeros((4, ncols), dtype=torch.float64)
        with pytest.raises(RuntimeError, match="overflowed feature table"):
            ex.submit(rows, torch.tensor([-1 if rank == 1 else 2], dtype=torch.int64))
    finally:
        dist.destroy_process_group()
