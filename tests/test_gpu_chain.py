"""GPU parity of the BASELINE configurations (C2, C3) through the batch drivers, against the oracle
chains and the golden vectors; plus size-independent properties at 2048 x 2048."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from arcadia_microscopy_tools_amd.device import get_context

    return get_context()


def _check_props(props, ref, rtol=1e-5):
    assert list(props.keys()) == list(ref.keys())
    for k in ref:
        if k == "orientation":
            # an orientation is an axis: +pi/2 and -pi/2 coincide (atan2 of a signed zero); exactly
            # symmetric regions (|theta| = pi/4) are version-sensitive in scikit-image
            sym = np.isclose(np.abs(ref[k]), np.pi / 4)
            d = (props[k] - ref[k] + np.pi / 2) % np.pi - np.pi / 2
            np.testing.assert_allclose(d[~sym], 0, atol=1e-8, err_msg=k)
        elif k == "eccentricity":
            np.testing.assert_allclose(props[k], ref[k], atol=1e-6, err_msg=k)
        elif k in ("label", "area", "area_convex") or k.startswith(("intensity_max", "intensity_min")):
            assert np.array_equal(props[k], ref[k]), k
        else:
            np.testing.assert_allclose(props[k], ref[k], rtol=rtol, err_msg=k)


def test_c2_c3_golden_256(ctx, golden):
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    g = golden("c2c3_256")
    fov = g["fov"]
    d = ctx.asarray(fov[None])
    seg = FovSegmenter(1, 4, 256, 256, ctx=ctx, max_cells=64)
    lab8 = seg.run_c2(d).numpy()[0]
    assert np.array_equal(lab8, g["labels8"])
    assert np.array_equal(seg.mask_a.numpy()[0], g["mask"])
    labels = seg.run_c3(d).numpy()[0]
    assert np.array_equal(labels, g["labels"])
    # the same chain with the watershed image materialised (separate watershed / clear_border + relabel calls)
    seg2 = FovSegmenter(1, 4, 256, 256, ctx=ctx, max_cells=64, fused=False)
    labels2 = seg2.run_c3(d).numpy()[0]
    assert np.array_equal(seg2.ws.numpy()[0], g["watershed"])
    assert np.array_equal(labels2, g["labels"]) and np.array_equal(seg2.ncells.numpy(), seg.ncells.numpy())
    res = seg.result()
    from oracle import chains

    _, ref = chains.c3_chain(fov)
    _check_props(res.feature_tables()[0], ref)


def test_gaussian_otsu_codes_equal_separate_operators(ctx):
    """The mask chain without the float64 plane (amt_gaussian_otsu_codes): thresholds, histograms, min / max and the
    thresholded masks must equal those of gaussian -> threshold_otsu -> '>' bit for bit; planes with different
    statistics in one batch, a constant plane, sigma with radius 4 / 8 / 12, sizes whose last tile is partial."""
    from arcadia_microscopy_tools_amd import hipops, synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    rng = np.random.default_rng(17)
    H, W = 300, 520
    fov = synth.synth_fov(3, size=520)[:, :H, :]
    planes = np.stack([fov[1], rng.integers(0, 65536, (H, W)).astype(np.uint16),
                       np.full((H, W), 1234, np.uint16), (fov[1] // 64 * 64).astype(np.uint16),
                       rng.integers(0, 3, (H, W)).astype(np.uint16) * 30000])
    d = ctx.asarray(planes)
    n = planes.shape[0]
    for sigma in (1.0, 2.0, 3.0):
        assert hipops.gaussian_otsu_codes_supported(d, sigma)
        codes, thr, tc = ctx.empty(planes.shape, np.uint16), ctx.empty((n,), np.float64), ctx.empty((n,), np.float64)
        mm, hist = ctx.empty((n, 2), np.float64), ctx.empty((n, 256), np.uint32)
        hipops.gaussian_otsu_codes(d, sigma, codes, thr, tc, mm, hist)
        g = hipops.gaussian(d, sigma)
        thr_ref = hipops.threshold_otsu(g).numpy()
        assert np.array_equal(thr.numpy(), thr_ref), sigma
        gn = g.numpy()
        assert np.array_equal(mm.numpy(), np.stack([gn.min(axis=(1, 2)), gn.max(axis=(1, 2))], axis=1))
        hf, _ = hipops.histogram_f64(g)
        hn = hist.numpy()
        for b in range(n):
            if gn[b].min() < gn[b].max():
                assert np.array_equal(hn[b], hf.numpy()[b]), (sigma, b)
        mask = hipops.greater_than(codes, tc).numpy()
        assert np.array_equal(mask, gn > thr_ref[:, None, None]), sigma
        # every other threshold of the code space is exact too: code > 2k  <=>  value > centre of bin k
        lo, hi = mm.numpy()[0]
        edges = np.linspace(lo, hi, 257)
        centres = (edges[:-1] + edges[1:]) / 2.0
        c0 = codes.numpy()[0]
        for k in (0, 1, 17, 128, 254, 255):
            assert np.array_equal(c0 > 2 * k, gn[0] > centres[k]), (sigma, k)
    # through the batch driver: the code path and the separate operators give the same masks and labels
    fovs = np.stack([synth.synth_fov(i, size=512) for i in (4, 6)])
    df = ctx.asarray(fovs)
    a = FovSegmenter(2, 4, 512, 512, ctx=ctx, max_cells=512, low_traffic=True)
    b = FovSegmenter(2, 4, 512, 512, ctx=ctx, max_cells=512)
    la, lb = a.run_c3(df).numpy(), b.run_c3(df).numpy()
    assert a.codes_path and not b.codes_path
    assert np.array_equal(a.thr.numpy(), b.thr.numpy())
    assert np.array_equal(a.mask_a.numpy(), b.mask_a.numpy()) and np.array_equal(la, lb)


def test_otsu_bins_mask_equals_plain_comparison(ctx):
    """threshold_otsu_bins + threshold_open_close(bins=...) -- the comparison by the byte plane of histogram bins -- against
    threshold_otsu + threshold_open_close on the float64 plane: same thresholds, same masks, on widths that take the
    16-pixel kernel (multiples of 64) and widths that fall back, smooth / noisy / constant / two-valued planes."""
    from arcadia_microscopy_tools_amd import hipops, synth

    rng = np.random.default_rng(23)
    for H, W in ((130, 512), (70, 520), (96, 64)):
        fov = synth.synth_fov(5, size=max(H, W))[:, :H, :W]
        planes = np.stack([fov[1], rng.integers(0, 65536, (H, W)).astype(np.uint16), np.full((H, W), 999, np.uint16),
                           rng.integers(0, 2, (H, W)).astype(np.uint16) * 40000])
        d = ctx.asarray(planes)
        n = planes.shape[0]
        mm = ctx.empty((n, 2), np.float64)
        g = hipops.gaussian(d, 2.0, minmax_out=mm)
        thr, tc = ctx.empty((n,), np.float64), ctx.empty((n,), np.float64)
        bins = ctx.empty(planes.shape, np.uint8)
        hipops.threshold_otsu_bins(g, mm, thr, tc, bins)
        ref_thr = hipops.threshold_otsu(g, minmax=mm)
        assert np.array_equal(thr.numpy(), ref_thr.numpy()), (H, W)
        gn = g.numpy()
        lo, hi = mm.numpy()[0]
        edges = np.linspace(lo, hi, 257)
        ref_bins = np.clip(np.searchsorted(edges, gn[0], side="right") - 1, 0, 255)
        assert np.array_equal(bins.numpy()[0], ref_bins.astype(np.uint8)), (H, W)
        fp = hipops.disk(2)
        got = hipops.threshold_open_close(g, thr, fp, bins=bins, thr_code=tc).numpy()
        ref = hipops.threshold_open_close(g, ref_thr, fp).numpy()
        assert np.array_equal(got, ref), (H, W)
        # every threshold that is a bin centre works through the bin plane: shift the threshold's bin by hand
        centres = (edges[:-1] + edges[1:]) / 2.0
        for k in (0, 3, 100, 254, 255):
            t1 = ctx.asarray(np.array([centres[k]] * n))
            c1 = ctx.asarray(np.array([2.0 * k] * n))
            a = hipops.threshold_open_close(g[0:1], t1[0:1], fp, bins=bins[0:1], thr_code=c1[0:1]).numpy()
            b = hipops.threshold_open_close(g[0:1], t1[0:1], fp).numpy()
            assert np.array_equal(a, b), (H, W, k)


def test_segmenter_reuse_sparse_clears(ctx):
    """A FovSegmenter keeps its peak / marker planes between runs and clears only what the previous run wrote
    (label_sparse keep=, peak_mask keep=): different batches through ONE segmenter must equal fresh segmenters, and
    the low-level pair must survive a run whose peak list overflowed its capacity."""
    from arcadia_microscopy_tools_amd import hipops, synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    sets = [np.stack([synth.synth_fov(i, size=384) for i in idx]) for idx in ((11, 12), (13, 14), (11, 14))]
    seg = FovSegmenter(2, 4, 384, 384, ctx=ctx, max_cells=256)
    for k, fovs in enumerate(sets):
        d = ctx.asarray(fovs)
        got = seg.run_c3(d).numpy()
        fresh = FovSegmenter(2, 4, 384, 384, ctx=ctx, max_cells=256)
        ref = fresh.run_c3(d).numpy()
        assert np.array_equal(got, ref), k
        assert np.array_equal(seg.peaks.numpy(), fresh.peaks.numpy()) and np.array_equal(seg.markers.numpy(), fresh.markers.numpy())
        assert np.array_equal(seg.table.numpy()[:, : int(seg.ncells.numpy().max())],
                              fresh.table.numpy()[:, : int(fresh.ncells.numpy().max())])
    # low level, with a tiny list capacity so that the middle run overflows
    rng = np.random.default_rng(2)
    H, W, cap = 96, 128, 64
    peaks, markers = ctx.zeros((1, H, W), np.uint8), ctx.zeros((1, H, W), np.int32)
    keep = (ctx.empty((1, cap), np.int32), ctx.zeros((1,), np.int32))
    status = ctx.zeros((1,), np.int32)
    seen = set()
    for it, density in enumerate((0.002, 0.2, 0.003, 0.0, 0.004)):
        m = np.zeros((1, H, W), np.uint8)
        m[0, 4:-4, 4:-4] = 1
        # a relief whose local maxima are isolated random pixels: every one of them is a peak at min_distance 1
        d2 = np.zeros((1, H, W), np.int32)
        pts = rng.random((1, H, W)) < density
        pts[:, ::2, :] = False
        pts[:, :, ::2] = False
        pts &= m.astype(bool)
        d2[pts] = 50
        dd2, dm = ctx.asarray(d2), ctx.asarray(m)
        hipops.peak_mask(dd2, dm, 1, out=peaks, keep=keep, status=status)
        ref_peaks = hipops.peak_mask(dd2, dm, 1).numpy()
        assert np.array_equal(peaks.numpy(), ref_peaks), it
        hipops.label_sparse(peaks, 1, capacity=cap, out=markers, count=status, keep=keep)
        ref_markers, ref_count = hipops.label_sparse(ctx.asarray(ref_peaks), 1, capacity=cap)
        assert np.array_equal(status.numpy(), ref_count.numpy()), it
        seen.add(int(ref_count.numpy()[0]) >= 0)
        if ref_count.numpy()[0] >= 0:
            assert np.array_equal(markers.numpy(), ref_markers.numpy()), it
    assert seen == {True, False}  # the dense run overflowed the 64-entry lists, the others did not


def test_c3_batch_vs_oracle(ctx):
    """Several different FOVs in one batch, odd image size, every FOV checked against the CPU oracle."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import segment_fovs
    from oracle import chains

    fovs = np.stack([synth.synth_fov(i, size=384) for i in (1, 2, 5)])
    res = segment_fovs(fovs, ctx=ctx, max_cells=256)
    labels = res.labels_numpy()
    assert labels.dtype == np.int64
    tables = res.feature_tables()
    for b in range(3):
        ref_labels, ref_props = chains.c3_chain(fovs[b])
        assert np.array_equal(labels[b], ref_labels), f"fov {b}"
        _check_props(tables[b], ref_props)
    odd = np.stack([synth.synth_fov(9, size=384)[:, :301, :333]])
    res = segment_fovs(np.ascontiguousarray(odd), ctx=ctx, max_cells=256)
    ref_labels, ref_props = chains.c3_chain(odd[0])
    assert np.array_equal(res.labels_numpy()[0], ref_labels)
    _check_props(res.feature_tables()[0], ref_props)


def test_c3_full_size_properties(ctx):
    """2048 x 2048 (BASELINE size): properties that do not need the CPU oracle at full size, plus a full
    oracle comparison of ONE FOV (the oracle takes a few seconds per 2048^2 FOV)."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    fovs = np.stack([synth.synth_fov(i) for i in (0, 1)])
    d = ctx.asarray(fovs)
    seg = FovSegmenter(2, 4, 2048, 2048, ctx=ctx)
    labels = seg.run_c3(d).numpy()
    res = seg.result()
    for b in range(2):
        lab = labels[b]
        k = int(res.ncells[b])
        assert k > 500
        # sequential labels 1..K, every label present, none on the border frame
        assert np.array_equal(np.unique(lab), np.arange(0, k + 1))
        frame = np.concatenate([lab[0], lab[-1], lab[:, 0], lab[:, -1]])
        assert frame.max() == 0
        # labels live inside the mask; areas sum to the labelled pixel count
        t = res.feature_tables()[b]
        assert int(t["area"].sum()) == int((lab > 0).sum())
        assert np.all((lab > 0) <= (seg.mask_a.numpy()[b] > 0))
        # idempotence: relabelling the result changes nothing
        from arcadia_microscopy_tools_amd import hipops

        again, cnt = hipops.relabel_sequential(ctx.asarray(lab), k)
        assert np.array_equal(again.numpy(), lab) and cnt.numpy()[0] == k
    # rerunning the same batch is bit-reproducible (no order-dependent arithmetic)
    labels2 = seg.run_c3(d).numpy()
    assert np.array_equal(labels, labels2)
    t2 = seg.result().feature_tables()[0]
    t1 = res.feature_tables()[0]
    for kk in t1:
        assert np.array_equal(t1[kk], t2[kk]), kk
    ref_labels, ref_props = chains.c3_chain(fovs[0])
    assert np.array_equal(labels[0].astype(np.int64), ref_labels)
    _check_props(res.feature_tables()[0], ref_props)


def test_c2_full_size_vs_oracle(ctx):
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    fov = synth.synth_fov(3)
    seg = FovSegmenter(1, 4, 2048, 2048, ctx=ctx)
    lab = seg.run_c2(ctx.asarray(fov[None])).numpy()[0]
    assert np.array_equal(lab, chains.c2_chain(fov[1]))


def test_degenerate_fovs(ctx):
    """Empty / constant / saturated / tiny fields of view must not crash or hang, and must match the oracle."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains, skops

    S = 192
    base = synth.synth_fov(11, size=S)
    fovs = np.stack([base, base, base, base, base])
    fovs[1] = 0                                   # all dark: constant image, empty mask
    fovs[2, 1] = 65535                            # saturated DAPI: constant
    fovs[3, 1, :, :] = 300
    fovs[3, 1, 60:130, 50:150] = 9000             # one big rectangle (single component, plateau peaks)
    fovs[4, 1, :8, :] = 60000                     # bright band touching the border + nuclei
    seg = FovSegmenter(5, 4, S, S, ctx=ctx, max_cells=256)
    labels = seg.run_c3(ctx.asarray(fovs)).numpy()
    res = seg.result()
    tables = res.feature_tables()
    for b in range(5):
        mask, _, _ = chains.c2_mask(fovs[b, 1])
        if mask.any():
            edt = skops.distance_transform_edt(mask)
            markers, _ = skops.peak_markers(edt, mask, 5)
            from oracle.watershed import watershed

            ws = watershed(skops.seeded_flood_image(edt, markers), markers, mask=mask)
            cleared = skops.clear_border(ws)
            ref = skops.relabel_sequential(cleared) if cleared.max() > 0 else cleared
        else:
            ref = np.zeros((S, S), np.int64)
        assert np.array_equal(labels[b], ref), f"fov {b}"
        assert res.ncells[b] == ref.max()
        assert len(tables[b]["area"]) == ref.max()
    assert res.ncells[1] == 0 and res.ncells[2] == 0
    # config 2 on the same inputs
    lab8 = seg.run_c2(ctx.asarray(fovs)).numpy()
    for b in range(5):
        assert np.array_equal(lab8[b], chains.c2_chain(fovs[b, 1])), f"c2 fov {b}"


def test_tiny_and_odd_shapes(ctx):
    from arcadia_microscopy_tools_amd import hipops
    from oracle import skops

    rng = np.random.default_rng(21)
    for shape in ((1, 1), (1, 7), (9, 1), (3, 3), (17, 65), (65, 129)):
        m = rng.random(shape) < 0.6
        d = ctx.asarray(m)
        assert np.array_equal(hipops.label(d)[0].numpy(), skops.label(m)), shape
        if min(shape) >= 3:
            assert np.array_equal(hipops.binary_opening(d, skops.disk(1)).numpy(),
                                  skops.binary_opening(m, skops.disk(1))), shape
        u = rng.integers(0, 60000, shape).astype(np.uint16)
        du = ctx.asarray(u)
        assert np.array_equal(hipops.gaussian(du, 1.0).numpy(), skops.gaussian(u, 1.0)), shape
        assert hipops.threshold_otsu(du).numpy()[0] == skops.threshold_otsu(u) or u.size == 1, shape
        if m.any() and not m.all():
            assert np.array_equal(hipops.edt(d)[1].numpy(), skops.distance_transform_edt(m)), shape


def test_c3_small_and_non_tile_sizes(ctx):
    """Whole chain on sizes far from the kernels' tile sizes (smaller than one tile, just above one tile, prime)."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import segment_fovs
    from oracle import chains

    big = synth.synth_fov(21, size=256)
    for h, w in ((40, 50), (65, 129), (127, 193), (97, 64)):
        fov = np.ascontiguousarray(big[:, 30:30 + h, 17:17 + w])
        res = segment_fovs(fov[None], ctx=ctx, max_cells=128)
        ref_labels, ref_props = chains.c3_chain(fov)
        assert np.array_equal(res.labels_numpy()[0], ref_labels), (h, w)
        if ref_labels.max() > 0:
            _check_props(res.feature_tables()[0], ref_props)


def test_fused_watershed_tail_equals_separate_calls(ctx):
    """amt_watershed_edt_cleared (watershed + clear_border + relabel in one call, no watershed image) against the three
    separate calls and the oracle, on windows cut out of larger fields of view so that nuclei DO touch the frame (the
    synthetic generator keeps them away from it), plus a plane without any marker, at odd sizes."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    for size in (200, 333):
        big = np.stack([synth.synth_fov(40 + i, size=size + 60) for i in range(3)])
        fovs = np.ascontiguousarray(big[:, :, 25:25 + size, 31:31 + size])
        fovs[2, 1] = 300  # constant DAPI plane: no mask, no markers, no cells
        d = ctx.asarray(fovs)
        a = FovSegmenter(3, 4, size, size, ctx=ctx, max_cells=256, fused=True)
        b = FovSegmenter(3, 4, size, size, ctx=ctx, max_cells=256, fused=False)
        la, lb = a.run_c3(d).numpy(), b.run_c3(d).numpy()
        assert np.array_equal(la, lb)
        assert np.array_equal(a.ncells.numpy(), b.ncells.numpy()) and a.ncells.numpy()[2] == 0
        for k in range(2):
            assert int(b.ws.numpy()[k].max()) > int(lb[k].max()) > 0  # clear_border did drop frame-touching nuclei
            assert np.array_equal(la[k].astype(np.int64), chains.c3_labels(fovs[k, 1])[0])
        np.testing.assert_array_equal(a.table.numpy()[:, :8], b.table.numpy()[:, :8])


def test_c3_plain_relief_exact_ties():
    """SURVEY.md A.8 as written -- watershed(-edt, markers, mask) -- through the batch driver: planes whose markers tie
    (on an EDT relief: nearly all) are re-flooded by the single-heap emulation and equal scikit-image's result
    (oracle: the C restatement of its heap, pinned by tests/golden/watershed_cases.npz); bench.py reports this recipe
    as sublines.a8_exact."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.device import get_context
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    ctx = get_context()
    fovs = np.stack([synth.synth_fov(40 + i, size=384) for i in range(3)])
    seg = FovSegmenter(3, 4, 384, 384, ctx=ctx, relief="plain", ties="exact")
    seg.run_c3(ctx.asarray(fovs))
    got = seg.labels.numpy()
    assert seg.tied.numpy().any()  # the recipe does tie on these planes
    for b in range(3):
        want, _ = chains.c3_labels(fovs[b, 1], relief="plain")
        assert np.array_equal(got[b], want), (b, int((got[b] != want).sum()))
    seeded = FovSegmenter(3, 4, 384, 384, ctx=ctx)
    seeded.run_c3(ctx.asarray(fovs))
    assert seeded.labels.numpy().max() == got.max()  # same markers, same number of nuclei; the borders may differ


def test_host_tables_deliver_what_the_segmenters_computed():
    """plate.HostTables (the N = 1 delivery bench.py times): two contexts, six steps with changing batches; the rows
    that arrive on the host, two steps lagged, are the packed device tables of THAT step, in FOV order, and equal the
    per-FOV oracle tables."""
    from arcadia_microscopy_tools_amd import _hip, synth
    from arcadia_microscopy_tools_amd.device import Context, get_context
    from arcadia_microscopy_tools_amd.plate import HostTables, rows_to_table, plate_rows
    from arcadia_microscopy_tools_amd.segment import DEFAULT_CHANNELS, FovSegmenter
    from oracle import chains

    dev = get_context().device
    ctxs = [Context(dev), Context(dev)]
    sizes = (2, 1)
    segs = [FovSegmenter(n, 4, 256, 256, ctx=c, max_cells=512) for n, c in zip(sizes, ctxs)]
    ht = HostTables(segs, slots=4, lag=2)
    batches = [np.stack([synth.synth_fov(10 * step + i, size=256) for i in range(3)]) for step in range(6)]
    kept = {}
    for step, fovs in enumerate(batches):
        ht.point(step)
        off = 0
        for sg in segs:
            sg.run_c3(sg.ctx.asarray(fovs[off:off + sg.B]))
            off += sg.B
        ht.deliver_step(step, fov_index0=100 * step)
        if step >= 2:  # the rows of step - 2 have been requested by now
            kept[step - 2] = ht.rows_of(step - 2).copy()
    ht.flush()
    for step in (4, 5):
        kept[step] = ht.rows_of(step).copy()
    for step, fovs in enumerate(batches):
        rows = kept[step]
        assert rows.shape[1] == 2 + _hip.RP_NCOLS + 16
        assert np.array_equal(np.unique(rows[:, 0]), 100 * step + np.arange(3))
        table = rows_to_table(rows, DEFAULT_CHANNELS)
        ref = []
        for i in range(3):
            lab, props = chains.c3_chain(fovs[i])
            ref.append(props)
        from arcadia_microscopy_tools_amd.plate import pack_rows
        want = pack_rows([100 * step + i for i in range(3)], ref, DEFAULT_CHANNELS)
        assert table.shape == want.shape, step
        from arcadia_microscopy_tools_amd.plate import table_columns
        cols = table_columns(DEFAULT_CHANNELS)
        for j, name in enumerate(cols):
            if name == "orientation":  # an axis: compared modulo pi, exactly symmetric regions left out
                sym = np.isclose(np.abs(want[:, j]), np.pi / 4)
                dd = (table[:, j] - want[:, j] + np.pi / 2) % np.pi - np.pi / 2
                np.testing.assert_allclose(dd[~sym], 0, atol=1e-8, err_msg=f"{step} {name}")
            else:
                np.testing.assert_allclose(table[:, j], want[:, j], rtol=1e-5, atol=1e-6, err_msg=f"{step} {name}")
    ht.close()


def test_c3_on_widths_that_take_the_run_table_path():
    """Config 3 on planes whose width is a multiple of 16 but not of 64 and whose height is not a multiple of 64 (tiles
    cut by the image's edge on the run-table path of the watershed stage, the one-pass EDT's tail columns): labels
    identical to the oracle."""
    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.device import get_context
    from arcadia_microscopy_tools_amd.segment import FovSegmenter
    from oracle import chains

    ctx = get_context()
    for H, W in ((129, 208), (300, 464), (200, 80)):
        fov = np.stack([synth.synth_fov(40 + i, size=max(H, W))[:, :H, :W].copy() for i in range(2)])
        seg = FovSegmenter(2, 4, H, W, ctx=ctx, max_cells=4096)
        lab = seg.run_c3(ctx.asarray(fov)).numpy()
        for b in range(2):
            ref, _ = chains.c3_chain(fov[b])
            assert np.array_equal(lab[b], ref), (H, W, b)
