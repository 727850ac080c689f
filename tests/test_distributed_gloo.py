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


def _plate_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from arcadia_microscopy_tools_amd import _hip, plate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, B, K, C = 3, 2, 4, 2
        lay = plate.PlateLayout(S, world, B, K, C)
        local = torch.zeros(lay.local_bytes, dtype=torch.uint8)
        gathered = torch.empty(lay.gathered_bytes, dtype=torch.uint8)
        buf = local.numpy()
        for step in range(S):
            for b in range(B):  # what a segmenter would write through the pointers of PlateTables.point()
                o_t, o_i, o_c = lay.block_offsets(step, b)
                code = 1000.0 * rank + 100.0 * step + 10.0 * b
                buf[o_t: o_t + K * _hip.RP_NCOLS * 8].view(np.float64)[:] = code + np.arange(K * _hip.RP_NCOLS) / 100.0
                buf[o_i: o_i + K * C * 4 * 8].view(np.float64)[:] = -code - np.arange(K * C * 4) / 100.0
                buf[o_c: o_c + 4].view(np.int32)[:] = 1 + rank + step + b
            lay.gather_step(local, gathered, step)  # one collective per plate
        t, it, nc = lay.unpack(gathered)
        np.savez(os.path.join(out_dir, f"plate{rank}.npz"), t=t.numpy(), it=it.numpy(), nc=nc.numpy())
    finally:
        dist.destroy_process_group()


def test_plate_layout_all_gather_gloo_world2(tmp_path):
    """The byte layout and the per-plate all-gather of plate.PlateLayout (what plate.PlateTables runs over RCCL),
    world_size 2 with gloo: every rank ends with every rank's blocks, ordered (rank, step * B + fov)."""
    import torch.multiprocessing as mp

    from arcadia_microscopy_tools_amd import _hip

    port = _free_port()
    mp.spawn(_plate_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "plate0.npz"), np.load(tmp_path / "plate1.npz")
    S, B, K, C = 3, 2, 4, 2
    for key in ("t", "it", "nc"):
        assert np.array_equal(a[key], b[key]), key
    assert a["t"].shape == (2, S * B, K, _hip.RP_NCOLS) and a["it"].shape == (2, S * B, K, C, 4)
    assert a["nc"].shape == (2, S * B)
    for rank in range(2):
        for step in range(S):
            for fov in range(B):
                code = 1000.0 * rank + 100.0 * step + 10.0 * fov
                f = step * B + fov
                assert np.array_equal(a["t"][rank, f].ravel(), code + np.arange(K * _hip.RP_NCOLS) / 100.0)
                assert np.array_equal(a["it"][rank, f].ravel(), -code - np.arange(K * C * 4) / 100.0)
                assert a["nc"][rank, f] == 1 + rank + step + fov
