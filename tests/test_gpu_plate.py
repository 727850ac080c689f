"""GPU tests of the steps either side of the hot path: the plate's feature-table exchange (BASELINE configs[3],
SURVEY.md 8(e)) on a one-rank RCCL group, the ND2 -> device de-interleave (SURVEY.md 8(f) rank 1, R/nikon.py:25-43)
and the host -> HBM feeder, each checked against the CPU oracle / the golden fixture pixels."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHANNELS = ("BRIGHTFIELD", "DAPI", "FITC", "TRITC")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_rows(fovs, fov_indices):
    """Per-FOV oracle tables -> the plate table in plate.table_columns order."""
    from arcadia_microscopy_tools_amd import plate
    from oracle import chains

    tables = [chains.c3_chain(f)[1] for f in fovs]
    return plate.pack_rows(fov_indices, tables, CHANNELS), tables


def _assert_plate_table(got, ref, cols):
    assert got.shape == ref.shape
    for j, c in enumerate(cols):
        if c == "orientation":
            sym = np.isclose(np.abs(ref[:, j]), np.pi / 4)
            d = (got[:, j] - ref[:, j] + np.pi / 2) % np.pi - np.pi / 2
            np.testing.assert_allclose(d[~sym], 0, atol=1e-8, err_msg=c)
        elif c == "eccentricity":
            np.testing.assert_allclose(got[:, j], ref[:, j], atol=1e-6, err_msg=c)
        elif c in ("fov_index", "label", "area", "area_convex") or c.startswith(("intensity_max", "intensity_min")):
            assert np.array_equal(got[:, j], ref[:, j]), c
        else:
            np.testing.assert_allclose(got[:, j], ref[:, j], rtol=1e-5, err_msg=c)


def test_pack_plate_rows_kernel():
    """amt_pack_plate_rows against a numpy restatement: ragged counts, an empty FOV, explicit FOV indices, and the
    overflow flag."""
    from arcadia_microscopy_tools_amd import _hip, hipops
    from arcadia_microscopy_tools_amd.device import get_context

    ctx = get_context()
    rng = np.random.default_rng(5)
    B, K, C = 7, 150, 3
    table = rng.normal(size=(B, K, _hip.RP_NCOLS))
    itable = rng.normal(size=(B, K, C, 4))
    ncells = np.array([150, 0, 1, 77, 64, 65, 128], np.int32)
    fidx = np.array([40, 41, 42, 50, 51, 52, 60], np.int32)
    for use_idx in (False, True):
        rows, nrows = hipops.pack_plate_rows(ctx.asarray(table), ctx.asarray(itable), ctx.asarray(ncells), fov_index0=9,
                                             fov_index=ctx.asarray(fidx) if use_idx else None)
        n = int(nrows.numpy()[0])
        assert n == int(ncells.sum())
        got = rows.numpy()[:n]
        ref = []
        for b in range(B):
            for r in range(ncells[b]):
                f = fidx[b] if use_idx else 9 + b
                ref.append(np.concatenate([[f, r + 1], table[b, r], itable[b, r].ravel()]))
        assert np.array_equal(got, np.array(ref))
    bad = ncells.copy()
    bad[3] = K + 1
    _, nrows = hipops.pack_plate_rows(ctx.asarray(table), ctx.asarray(itable), ctx.asarray(bad))
    assert int(nrows.numpy()[0]) == -1
    # morphology only (C = 0)
    rows, nrows = hipops.pack_plate_rows(ctx.asarray(table), None, ctx.asarray(ncells))
    got = rows.numpy()[: int(nrows.numpy()[0])]
    assert got.shape[1] == 2 + _hip.RP_NCOLS and np.array_equal(got[0, 2:], table[0, 0])


def test_plate_tables_single_rank_rccl():
    """plate.PlateTables on a one-rank RCCL (backend "nccl") group: two streams' segmenters write two plates'
    tables into the staging ring, each plate is packed and exchanged (counts, then rows, one step later), and what
    arrives equals the segmenters' own tables and the per-FOV oracle tables."""
    import torch
    import torch.distributed as dist

    from arcadia_microscopy_tools_amd import plate, synth
    from arcadia_microscopy_tools_amd.device import Context
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        S, K = 256, 96
        plates = [np.stack([synth.synth_fov(20 * p + i, size=S) for i in range(3)]) for p in range(2)]
        ctxs = [Context(0), Context(0)]
        segs = [FovSegmenter(2, 4, S, S, ctx=ctxs[0], max_cells=K), FovSegmenter(1, 4, S, S, ctx=ctxs[1], max_cells=K)]
        pt = plate.PlateTables(segs, torch.device("cuda", 0), slots=2)
        local = []
        for step, fovs in enumerate(plates):
            pt.point(step)
            d0, d1 = ctxs[0].asarray(fovs[:2]), ctxs[1].asarray(fovs[2:])
            segs[0].run_c3(d0)
            segs[1].run_c3(d1)
            pt.gather_step(step, fov_index0=100 * step)
            assert len(pt.exchange.finished) == step  # the rows of a plate travel one step later
            for c in ctxs:
                c.synchronize()
            local.append([(s.table.numpy().copy(), s.itable.numpy().copy(), s.ncells.numpy().copy()) for s in segs])
        pt.all_gather()
        res = pt.result()
        assert len(res) == 2
        cols = plate.table_columns(CHANNELS)
        for step, (rows, counts) in enumerate(res):
            assert len(counts) == 1 and rows.shape == (counts[0], plate.packed_ncols(4))
            # gathered == local: the dense tables the segmenters wrote, row by row
            nc = np.concatenate([l[2] for l in local[step]])
            t = np.concatenate([l[0] for l in local[step]])
            it = np.concatenate([l[1] for l in local[step]])
            assert counts[0] == int(nc.sum())
            off = 0
            for b in range(3):
                blk = rows[off: off + nc[b]]
                assert np.all(blk[:, 0] == 100 * step + b) and np.array_equal(blk[:, 1], np.arange(1, nc[b] + 1))
                assert np.array_equal(blk[:, 2:16], t[b, : nc[b]])
                assert np.array_equal(blk[:, 16:], it[b, : nc[b]].reshape(nc[b], -1))
                off += nc[b]
            # plate table == the per-FOV oracle tables
            got = plate.rows_to_table(rows, CHANNELS)
            ref, _ = _oracle_rows(plates[step], [100 * step + b for b in range(3)])
            _assert_plate_table(got, ref, cols)
            df = plate.plate_dataframe(got, CHANNELS, n_columns=24)
            assert len(df) == counts[0] and df["well_id"].iloc[0] == plate.well_id(100 * step)
    finally:
        dist.destroy_process_group()


def test_deinterleave_and_load_nd2_on_device(golden, tmp_path):
    """(Y, X, C) ND2 frames -> (C, Y, X) on the device: amt_deinterleave_u16 against the numpy transpose on the
    reference fixture's pixels (tests/golden/nd2_multichannel.npz), multi-frame and odd sizes, then the whole
    load_nd2(use_device=True) path incl. padded rows, and config 1 (Otsu 2742 / 1297 px / 20 labels) on the result."""
    from conftest import write_synthetic_nd2

    from arcadia_microscopy_tools_amd import hipops, nd2lite
    from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, DAPI, FITC, TRITC
    from arcadia_microscopy_tools_amd.device import get_context
    from arcadia_microscopy_tools_amd.microscopy import MicroscopyImage
    from arcadia_microscopy_tools_amd.operations import apply_threshold

    ctx = get_context()
    g = golden("nd2_multichannel")
    px = g["pixels"]  # (4, 256, 256)
    yxc = np.ascontiguousarray(px.transpose(1, 2, 0))
    assert np.array_equal(hipops.deinterleave(ctx.asarray(yxc[None]), 4).numpy()[0], px)
    rng = np.random.default_rng(3)
    for shape in ((3, 37, 45, 4), (2, 64, 64, 2), (1, 5, 7, 3), (2, 33, 130, 1)):
        fr = rng.integers(0, 65536, shape).astype(np.uint16)
        assert np.array_equal(hipops.deinterleave(ctx.asarray(fr), shape[3]).numpy(), fr.transpose(0, 3, 1, 2))
    chans = [BRIGHTFIELD, DAPI, FITC, TRITC]
    f = write_synthetic_nd2(tmp_path / "fixture.nd2", yxc[None])
    arr, meta = nd2lite.load_nd2(f, channels=chans, use_device=True)
    assert arr.dtype == np.uint16 and np.array_equal(arr, px) and meta.sizes == {"C": 4, "Y": 256, "X": 256}
    odd = px[:, :37, :45]
    fp = write_synthetic_nd2(tmp_path / "padded.nd2", np.stack([odd, odd[:, ::-1]]).transpose(0, 2, 3, 1), row_pad_bytes=6)
    arr2, _ = nd2lite.load_nd2(fp, channels=chans, use_device=True)
    assert np.array_equal(arr2[0], odd) and np.array_equal(arr2[1], odd[:, ::-1])
    # config 1 on what the loader produced: DAPI Otsu + label (known answers SURVEY.md 8c)
    im = MicroscopyImage.from_nd2_path(f, channels=chans)
    mask = apply_threshold(im.get_channel_intensities(DAPI), "otsu")
    assert int(mask.sum()) == 1297
    lab, count = hipops.label(ctx.asarray(mask.astype(np.uint8)), 2)
    assert int(count.numpy()[0]) == 20 and np.array_equal(lab.numpy(), g["labels8"])


def test_fov_feeder_against_oracle():
    """Batches streamed from page-locked host memory through the double-buffered feeder, segmented while the next
    batch is in flight, against the CPU oracle's labels for every field of view."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.device import Context
    from arcadia_microscopy_tools_amd.feeder import FovFeeder
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    ctx = Context(0)
    batches = [np.stack([synth.synth_fov(10 * b + i, size=192) for i in range(2)]) for b in range(3)]
    seg = FovSegmenter(2, 4, 192, 192, ctx=ctx, max_cells=128)
    feeder = FovFeeder(batches[0].shape)
    feeder.host(0)[...] = batches[0]
    feeder.submit(0)
    got = []
    for i in range(3):
        slot = i % 2
        d = feeder.acquire(slot, [ctx])
        if i + 1 < 3:
            feeder.host(1 - slot)[...] = batches[i + 1]
            feeder.submit(1 - slot)
        lab = seg.run_c3(d)
        feeder.release(slot, [ctx])
        got.append(lab.numpy().copy())  # numpy() synchronises the compute stream
    feeder.close()
    for bi, (labs, fovs) in enumerate(zip(got, batches)):
        for j, (lab, fov) in enumerate(zip(labs, fovs)):
            ref = chains.c3_chain(fov)[0]
            assert np.array_equal(lab.astype(np.int64), ref), (
                f"batch {bi} FOV {j}: {int((lab != ref).sum())} px differ, max {int(lab.max())} vs {int(ref.max())}")
