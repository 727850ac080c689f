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
