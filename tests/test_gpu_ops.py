"""GPU parity: every HIP operator against the CPU oracle and the committed golden vectors.
All calls go through the C ABI (ctypes -> libamt_hip.so)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from arcadia_microscopy_tools_amd.device import get_context

    return get_context()


@pytest.fixture(scope="module")
def ops():
    from arcadia_microscopy_tools_amd import hipops

    return hipops


def test_gaussian_bit_exact(ctx, ops, golden):
    from oracle import skops

    g = golden("ops_192")
    u = g["u16"]
    d = ctx.asarray(u)
    for s in (0.6, 1.0, 2.0, 3.0, 5.0, 16.0):
        out = ops.gaussian(d, s).numpy()
        ref = skops.gaussian(u, s)
        assert np.array_equal(out, ref), f"sigma={s}: max abs diff {np.abs(out - ref).max()}"
    for mode in ("reflect", "mirror", "constant", "wrap", "nearest"):
        out = ops.gaussian(d, 2.0, mode=mode, cval=0.25).numpy()
        from scipy import ndimage as ndi

        ref = ndi.gaussian_filter(skops.img_as_float(u), 2.0, mode=mode, cval=0.25)
        assert np.array_equal(out, ref), mode
    # float64 input, batch of planes, odd sizes
    rng = np.random.default_rng(0)
    f = rng.random((3, 37, 101))
    out = ops.gaussian(ctx.asarray(f), 1.5).numpy()
    for i in range(3):
        assert np.array_equal(out[i], skops.gaussian(f[i], 1.5))
    # golden (skimage 0.18.3 + numpy 1.26 weights): last-bit agreement
    np.testing.assert_allclose(ops.gaussian(d, 2.0).numpy(), g["gauss_2.0"], rtol=0, atol=3e-16)


def test_gaussian_wide_radius_lds_dma_paths(ctx, ops):
    """The two-pass wide-radius Gaussian with LDS-DMA staging (W % 64 == 0 and W >= 256 + 2r): even and odd radii
    (tile origin shifted by one column), every boundary mode (wrap keeps the register-staged horizontal pass), heights
    that are not multiples of the tile, uint16 and float64 input, several planes; bit for bit against scipy."""
    from oracle import skops
    from scipy import ndimage as ndi

    rng = np.random.default_rng(77)
    u = rng.integers(0, 65536, (2, 200, 512)).astype(np.uint16)
    d = ctx.asarray(u)
    for s in (16.0, 15.8, 5.0, 20.3):
        out = ops.gaussian(d, s).numpy()
        for b in range(2):
            assert np.array_equal(out[b], skops.gaussian(u[b], s)), (s, b)
    f = rng.random((1, 77, 640))
    df = ctx.asarray(f)
    for mode in ("reflect", "mirror", "constant", "wrap", "nearest"):
        out = ops.gaussian(d, 16.0, mode=mode, cval=0.25).numpy()
        assert np.array_equal(out[1], ndi.gaussian_filter(skops.img_as_float(u[1]), 16.0, mode=mode, cval=0.25)), mode
        out = ops.gaussian(df, 7.0, mode=mode, cval=-1.5).numpy()
        assert np.array_equal(out[0], ndi.gaussian_filter(f[0], 7.0, mode=mode, cval=-1.5)), mode
    out = ops.difference_of_gaussians(d, 0.6, 16.0).numpy()
    assert np.array_equal(out[0], skops.difference_of_gaussians(u[0], 0.6, 16.0))
    # r = 64 has instances of its own (compile-time radius, fully unrolled): float64 input, rows that end inside a tile
    f2 = rng.random((2, 150, 448))
    for mode in ("reflect", "constant", "nearest"):
        out = ops.gaussian(ctx.asarray(f2), 16.0, mode=mode, cval=0.5).numpy()
        for b in range(2):
            assert np.array_equal(out[b], ndi.gaussian_filter(f2[b], 16.0, mode=mode, cval=0.5)), (mode, b)


def test_dog_bit_exact(ctx, ops, golden):
    from oracle import skops

    u = golden("ops_192")["u16"]
    out = ops.difference_of_gaussians(ctx.asarray(u), 0.6, 16.0).numpy()
    assert np.array_equal(out, skops.difference_of_gaussians(u, 0.6, 16.0))


def test_histograms_otsu_threshold(ctx, ops, golden):
    from oracle import skops

    g = golden("ops_192")
    u = g["u16"]
    d = ctx.asarray(u)
    h = ops.histogram_u16(d).numpy()[0]
    assert np.array_equal(h, np.bincount(u.ravel(), minlength=65536))
    t = ops.threshold_otsu(d).numpy()[0]
    assert t == skops.threshold_otsu(u) == g["thr_otsu_u16"]
    gz = skops.gaussian(u, 2.0)
    dg = ctx.asarray(gz)
    hf, mm = ops.histogram_f64(dg)
    rh, rc = skops.histogram(gz)
    assert np.array_equal(hf.numpy()[0], rh)
    assert np.array_equal(mm.numpy()[0], [gz.min(), gz.max()])
    tf = ops.threshold_otsu(dg).numpy()[0]
    assert tf == skops.threshold_otsu(gz)
    m = ops.greater_than(dg, ctx.asarray(np.array([tf]))).numpy()
    assert m.dtype == bool and np.array_equal(m, gz > tf)
    # bright image: values above the LDS window of the histogram kernel
    big = (u.astype(np.uint32) * 5 % 65536).astype(np.uint16)
    assert np.array_equal(ops.histogram_u16(ctx.asarray(big)).numpy()[0], np.bincount(big.ravel(), minlength=65536))
    assert ops.threshold_otsu(ctx.asarray(big)).numpy()[0] == skops.threshold_otsu(big)
    # constant image: skimage returns the value itself
    const = np.full((16, 16), 77, np.uint16)
    assert ops.threshold_otsu(ctx.asarray(const)).numpy()[0] == 77


def test_nd2_fixture_otsu_label(ctx, ops, golden):
    """BASELINE configs[0] on the GPU: fixture DAPI -> Otsu 2742 -> 1297 px -> 20 labels."""
    g = golden("nd2_multichannel")
    px = ctx.asarray(g["pixels"])  # (4, 256, 256): channel = pointer offset
    dapi = px[1]
    t = ops.threshold_otsu(dapi)
    assert t.numpy()[0] == 2742
    m = ops.greater_than(dapi, t)
    assert int(m.numpy().sum()) == 1297
    lab, cnt = ops.label(m)
    assert cnt.numpy()[0] == 20
    assert np.array_equal(lab.numpy(), g["labels8"])
    lab4, cnt4 = ops.label(m, connectivity=1)
    assert cnt4.numpy()[0] == 25 and np.array_equal(lab4.numpy(), g["labels4"])
    assert np.array_equal(ops.clear_border(lab).numpy(), g["cleared"])


def test_percentile_rescale(ctx, ops, golden):
    g = golden("ops_192")
    u = g["u16"]
    gz = g["gauss_2.0"]
    for img in (u, gz):
        d = ctx.asarray(img)
        for q in ((0, 100), (1, 99), (0.1, 99.9), (2, 98), (37.123, 50)):
            p = ops.percentile(d, q)
            ref = np.percentile(img, q)
            assert np.array_equal(p.numpy()[0], ref), (img.dtype, q)
            r = ops.rescale(d, p, (0, 1)).numpy()
            from oracle import skops

            assert np.array_equal(r, skops.rescale_intensity(img, (ref[0], ref[1]), (0, 1)))
        single = ops.percentile(d, 90).numpy()
        assert single.shape == (1, 1) and single[0, 0] == np.percentile(img, 90)
    # float images with massive ties (candidate lists overflow -> full-scan fallback), negative values, several
    # planes with different distributions in one call
    rng = np.random.default_rng(8)
    planes = np.stack([
        np.full((128, 160), 0.25),
        rng.integers(0, 3, (128, 160)).astype(np.float64),
        rng.normal(0.0, 1e-3, (128, 160)),
        np.exp(rng.normal(0.0, 4.0, (128, 160))) * np.sign(rng.normal(size=(128, 160))),
        np.clip(rng.normal(-1.2, 1.0, (128, 160)), 0, None),  # a clipped image: ~88 % exact zeros (R/operations.py:97)
    ])
    for q in ((0, 100), (1, 99), (50,), (12.5, 87.5, 99.99)):
        got = ops.percentile(ctx.asarray(planes), q).numpy()
        for b in range(planes.shape[0]):
            assert np.array_equal(got[b], np.atleast_1d(np.percentile(planes[b], q))), (b, q)


def test_percentile_f64_sampled_path(ctx, ops):
    """Planes of >= 65,536 samples take the one-pass path (sampled brackets + exact counts + list select): against
    np.percentile, bit for bit, on smooth / heavy-tailed / constant / few-valued / mostly-zero (clipped) / sorted and
    spatially structured planes, with up to six percentiles per call (two bracket groups) incl. 0 and 100."""
    rng = np.random.default_rng(21)
    H, W = 300, 352
    yy, xx = np.mgrid[0:H, 0:W]
    planes = np.stack([
        rng.normal(0.3, 0.05, (H, W)),
        np.exp(rng.normal(0.0, 3.0, (H, W))) * np.sign(rng.normal(size=(H, W))),
        np.full((H, W), -7.5),
        rng.integers(0, 4, (H, W)).astype(np.float64),
        np.clip(rng.normal(-1.2, 1.0, (H, W)), 0, None),
        np.arange(H * W, dtype=np.float64).reshape(H, W) / 7.0,
        np.sin(xx / 9.0) * np.cos(yy / 13.0) + (xx // 64) * 0.5,      # stripes with the period of nothing in the hash
        np.where((yy // 50 + xx // 50) % 2 == 0, 1.0, rng.random((H, W))),  # half the plane one value
    ])
    d = ctx.asarray(planes)
    for q in ((1, 99), (90,), (0, 100), (50,), (0.01, 37.123, 50, 62.5, 99.9, 100), (0, 0.5, 99.5)):
        got = ops.percentile(d, q).numpy()
        for b in range(planes.shape[0]):
            assert np.array_equal(got[b], np.atleast_1d(np.percentile(planes[b], q))), (b, q)
    # one big plane at the benchmark size, heavy ties included
    big = np.clip(rng.normal(-0.5, 1.0, (2048, 2048)), 0, None)
    assert np.array_equal(ops.percentile(ctx.asarray(big), (1, 99)).numpy()[0], np.percentile(big, (1, 99)))
    # an odd number of samples (scalar loads instead of 16-byte pairs), planes that start at odd element offsets
    odd = rng.normal(0.0, 1.0, (3, 257, 257))
    got = ops.percentile(ctx.asarray(odd), (1, 50, 99)).numpy()
    for b in range(3):
        assert np.array_equal(got[b], np.percentile(odd[b], (1, 50, 99))), b
    # values sorted in space: whole 1,024-sample runs of a block fall inside one bracket, its LDS stage fills up and the
    # rest of its keys go straight to the list (a sorted megapixel; a ramp repeated with the period of the block stride,
    # so that EVERY run of some blocks lies inside the bracket; the same descending)
    ramp = np.sort(rng.normal(0.0, 1.0, 1 << 18))
    srt = np.stack([
        np.sort(rng.random(1 << 20)).reshape(1024, 1024),
        np.tile(ramp, 4).reshape(1024, 1024),
        np.tile(ramp[::-1], 4).reshape(1024, 1024) * 3.0 - 1.0,
    ])
    ds = ctx.asarray(srt)
    for q in ((90,), (1, 99), (0.5, 25, 50, 99.9), (10, 20, 30, 40, 60)):
        got = ops.percentile(ds, q).numpy()
        for b in range(srt.shape[0]):
            assert np.array_equal(got[b], np.atleast_1d(np.percentile(srt[b], q))), (b, q)


def test_binary_morphology(ctx, ops, golden):
    from oracle import skops

    g = golden("ops_192")
    m = g["mask"]
    d = ctx.asarray(m)
    for r in (1, 2, 3):
        se = skops.disk(r)
        assert np.array_equal(ops.binary_erosion(d, se).numpy(), g[f"berode_d{r}"])
        assert np.array_equal(ops.binary_dilation(d, se).numpy(), g[f"bdilate_d{r}"])
        assert np.array_equal(ops.binary_opening(d, se).numpy(), g[f"bopen_d{r}"])
        assert np.array_equal(ops.binary_closing(d, se).numpy(), g[f"bclose_d{r}"])
    assert np.array_equal(ops.binary_erosion(d).numpy(), g["berode_cross"])
    assert np.array_equal(ops.binary_dilation(d).numpy(), g["bdilate_cross"])
    # masks touching the border, non-multiple-of-tile sizes, batch
    rng = np.random.default_rng(3)
    mb = rng.random((2, 45, 70)) < 0.6
    db = ctx.asarray(mb)
    se = skops.disk(2)
    for i in range(2):
        assert np.array_equal(ops.binary_opening(db, se).numpy()[i], skops.binary_opening(mb[i], se))
        assert np.array_equal(ops.binary_closing(db, se).numpy()[i], skops.binary_closing(mb[i], se))
        assert np.array_equal(ops.binary_erosion(db, se).numpy()[i], skops.binary_erosion(mb[i], se))


def test_grey_morphology_median_tophat(ctx, ops, golden):
    from oracle import skops

    g = golden("ops_192")
    u = g["u16"]
    d = ctx.asarray(u)
    for r in (1, 2, 3):
        se = skops.disk(r)
        assert np.array_equal(ops.erosion(d, se).numpy(), g[f"erode_d{r}"])
        assert np.array_equal(ops.dilation(d, se).numpy(), g[f"dilate_d{r}"])
        assert np.array_equal(ops.opening(d, se).numpy(), g[f"open_d{r}"])
        assert np.array_equal(ops.closing(d, se).numpy(), g[f"close_d{r}"])
        assert np.array_equal(ops.median(d, se).numpy(), g[f"median_d{r}"])
    assert np.array_equal(ops.median(d).numpy(), g["median_3x3"])
    assert np.array_equal(ops.white_tophat(d, skops.disk(3)).numpy(), g["tophat_d3"])
    assert np.array_equal(ops.white_tophat(d, skops.disk(7)).numpy(), g["tophat_d7"])
    gz = g["gauss_2.0"]
    assert np.array_equal(ops.median(ctx.asarray(gz), skops.disk(2)).numpy(), skops.median(gz, skops.disk(2)))
    assert np.array_equal(ops.erosion(ctx.asarray(gz), skops.disk(2)).numpy(), skops.erosion(gz, skops.disk(2)))


def test_even_and_asymmetric_footprints(ctx, ops, golden):
    """Even-sized footprints against the real scikit-image (tests/golden/even_footprints.npz): the host pads them to odd
    ones the way scikit-image / scipy place them; plus asymmetric odd footprints through every grey operator."""
    from oracle import skops

    g = golden("even_footprints")
    img, mask = g["img"], g["mask"]
    d, dm = ctx.asarray(img), ctx.asarray(mask)
    for name in ("s2", "s4", "r2x3", "r3x4", "c4x1", "L4"):
        fp = g[f"fp_{name}"]
        assert np.array_equal(ops.erosion(d, fp).numpy(), g[f"erosion_{name}"]), name
        assert np.array_equal(ops.dilation(d, fp).numpy(), g[f"dilation_{name}"]), name
        assert np.array_equal(ops.opening(d, fp).numpy(), g[f"opening_{name}"]), name
        assert np.array_equal(ops.closing(d, fp).numpy(), g[f"closing_{name}"]), name
        assert np.array_equal(ops.white_tophat(d, fp).numpy(), g[f"tophat_{name}"]), name
        assert np.array_equal(ops.median(d, fp).numpy(), g[f"median_{name}"]), name
        assert np.array_equal(ops.binary_erosion(dm, fp).numpy(), g[f"berosion_{name}"]), name
        assert np.array_equal(ops.binary_dilation(dm, fp).numpy(), g[f"bdilation_{name}"]), name
        assert np.array_equal(ops.binary_opening(dm, fp).numpy(), g[f"bopening_{name}"]), name
        assert np.array_equal(ops.binary_closing(dm, fp).numpy(), g[f"bclosing_{name}"]), name
    rng = np.random.default_rng(31)
    big = rng.integers(0, 65536, (70, 520)).astype(np.uint16)  # wide enough for the strip kernels
    db = ctx.asarray(big)
    for fp in (np.array([[1, 1, 0], [0, 1, 0], [0, 0, 0]], np.uint8), np.array([[0, 0, 1, 1, 1]], np.uint8),
               np.array([[1, 0, 0], [1, 1, 0], [1, 1, 1], [0, 0, 1], [0, 0, 1]], np.uint8), np.ones((4, 6), np.uint8)):
        for fn in ("erosion", "dilation", "opening", "closing", "white_tophat", "median"):
            assert np.array_equal(getattr(ops, fn)(db, fp).numpy(), getattr(skops, fn)(big, fp)), (fp.shape, fn)


def test_label_sparse_reuse_clears_only_previous_pixels(ctx, ops):
    """label_sparse(keep=...) -- the marker planes of a batch driver: the output plane is zeroed once and every call
    undoes only the previous call's writes.  Three different masks in a row (with an overflowing one in between) must
    give what a fresh label_sparse gives, pixels of earlier masks included (they must be zero again)."""
    rng = np.random.default_rng(9)
    shape = (3, 150, 200)
    cap = 4096
    out = ctx.zeros(shape, np.int32)
    keep = (ctx.empty((3, cap), np.int32), ctx.zeros((3,), np.int32))
    for it, p in enumerate((0.01, 0.05, 0.3, 0.02, 0.0, 0.04)):  # 0.3 overflows the capacity of 4,096 pixels
        m = rng.random(shape) < p
        d = ctx.asarray(m)
        got, cnt = ops.label_sparse(d, 1, capacity=cap, out=out, keep=keep)
        ref, rcnt = ops.label_sparse(d, 1, capacity=cap)
        assert np.array_equal(cnt.numpy(), rcnt.numpy()), it
        if (rcnt.numpy() >= 0).all():
            assert np.array_equal(got.numpy(), ref.numpy()), it
        else:
            assert (cnt.numpy() == -1).all()


def test_label_random(ctx, ops):
    from oracle import skops

    rng = np.random.default_rng(5)
    for shape, p in (((64, 64), 0.5), ((33, 130), 0.59), ((200, 77), 0.4), ((1, 50), 0.5), ((50, 1), 0.5)):
        m = rng.random(shape) < p
        for conn in (1, 2):
            lab, cnt = ops.label(ctx.asarray(m), connectivity=conn)
            ref = skops.label(m, conn)
            assert cnt.numpy()[0] == ref.max()
            assert np.array_equal(lab.numpy(), ref), (shape, conn)
    # integer input: touching regions with different values stay separate
    vals = rng.integers(0, 4, (60, 90)).astype(np.int32)
    lab, cnt = ops.label(ctx.asarray(vals))
    assert np.array_equal(lab.numpy(), skops.label(vals))
    # uint8 images that are NOT 0 / 1 masks are labelled by equal value too: widths that take the bit-parallel mask
    # kernel must notice and fall back (ccl_tile_bits_kernel -> ccl_tile_fallback_kernel); a batch mixes both kinds
    for shape in ((64, 96), (130, 256)):
        v8 = rng.integers(0, 4, shape).astype(np.uint8)
        for conn in (1, 2):
            lab, cnt = ops.label(ctx.asarray(v8), connectivity=conn)
            assert np.array_equal(lab.numpy(), skops.label(v8.astype(np.int32), conn)), (shape, conn)
        both = np.stack([v8, (v8 > 1).astype(np.uint8)])
        lab, cnt = ops.label(ctx.asarray(both), connectivity=2)
        for b in range(2):
            assert np.array_equal(lab.numpy()[b], skops.label(both[b].astype(np.int32), 2)), (shape, b)
    # masks whose width is a multiple of 16 (the bit-parallel kernel), incl. tiles cut by the image's edge
    for shape, p in (((64, 64), 0.5), ((70, 144), 0.6), ((200, 80), 0.45), ((3, 16), 0.7), ((129, 2048), 0.55)):
        m = rng.random(shape) < p
        for conn in (1, 2):
            lab, cnt = ops.label(ctx.asarray(m), connectivity=conn)
            ref = skops.label(m, conn)
            assert cnt.numpy()[0] == ref.max() and np.array_equal(lab.numpy(), ref), (shape, conn)
    # empty
    lab, cnt = ops.label(ctx.asarray(np.zeros((20, 20), bool)))
    assert cnt.numpy()[0] == 0 and lab.numpy().max() == 0


def test_clear_border_relabel(ctx, ops):
    from oracle import skops

    rng = np.random.default_rng(6)
    vals = (rng.integers(0, 6, (80, 120)) * (rng.random((80, 120)) < 0.7)).astype(np.int32)
    cb = ops.clear_border(ctx.asarray(vals)).numpy()
    assert np.array_equal(cb, skops.clear_border(vals))
    sparse = (vals * 7).astype(np.int32)
    rl, cnt = ops.relabel_sequential(ctx.asarray(sparse), int(sparse.max()))
    assert np.array_equal(rl.numpy(), skops.relabel_sequential(sparse))
    assert cnt.numpy()[0] == len(np.unique(sparse)) - 1


def test_edt_peaks(ctx, ops, golden):
    from oracle import skops

    g = golden("c2c3_256")
    m = g["mask"]
    d2, e = ops.edt(ctx.asarray(m))
    assert np.array_equal(e.numpy(), g["edt"])
    assert np.array_equal(d2.numpy().astype(np.float64), np.round(g["edt"] ** 2))
    pk = ops.peak_mask(d2, ctx.asarray(m), 5)
    markers, cnt = ops.label(pk, connectivity=1)
    assert np.array_equal(markers.numpy(), g["markers"])
    # random masks incl. rows/columns without background, thick blobs
    rng = np.random.default_rng(8)
    for shape, p in (((50, 64), 0.9), ((97, 33), 0.97), ((40, 40), 0.5), ((60, 72), 0.995), ((31, 136), 0.9),
                     ((20, 2056), 0.999)):
        mm = rng.random(shape) < p
        mm[0, 0] = False
        d2, e = ops.edt(ctx.asarray(mm))
        assert np.array_equal(e.numpy(), skops.distance_transform_edt(mm)), shape


def test_edt_without_background_and_windows_beyond_the_image(ctx, ops):
    """Two corners found by tests/campaigns/fuzz_tiny.py: a plane without a single zero pixel (scipy's feature transform
    then measures from index (-1, 0)), and Niblack / Sauvola windows larger than the image (numpy's 'reflect' padding
    bounces as often as it takes)."""
    from arcadia_microscopy_tools_amd.operations import apply_threshold
    from oracle import skops

    for shape in ((1, 1), (1, 2), (2, 2), (7, 5), (40, 70), (130, 64)):
        m = np.ones(shape, bool)
        d2, e = ops.edt(ctx.asarray(m))
        ref = skops.distance_transform_edt(m)
        assert np.array_equal(e.numpy(), ref), shape
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        assert np.array_equal(d2.numpy(), (yy + 1) ** 2 + xx ** 2), shape
    # a stack in which only ONE plane has no background
    m = np.ones((2, 9, 11), bool)
    m[1, 4, 5] = False
    e = ops.edt(ctx.asarray(m))[1].numpy()
    assert np.array_equal(e[0], skops.distance_transform_edt(m[0])) and np.array_equal(e[1], skops.distance_transform_edt(m[1]))
    rng = np.random.default_rng(3)
    for shape, w in (((3, 4), 9), ((1, 6), 5), ((5, 1), 3), ((2, 2), 15), ((10, 30), 25), ((30, 7), 15)):
        img = rng.integers(0, 65536, shape).astype(np.uint16)
        for method in ("niblack", "sauvola"):
            got = apply_threshold(img, method=method, window_size=w)
            assert np.array_equal(got, img > getattr(skops, "threshold_" + method)(img, window_size=w)), (shape, w, method)
        f = img.astype(np.float64) / 65535.0
        # float64 images: window sums in a different order than scikit-image's integral images -> rounding only
        np.testing.assert_allclose(ops.window_threshold(ctx.asarray(f), w, "niblack", 0.2).numpy(),
                                   skops.threshold_niblack(f, w, 0.2), rtol=1e-7, atol=1e-12)


def test_peak_mask_random(ctx, ops):
    """peak_mask against scipy's maximum_filter formulation (oracle/skops.py:peak_markers) on integer
    reliefs with plateaus and ties, every strip / row-block boundary, m = 0, 1, 5, 16, masks unrelated to the
    relief, two planes per call."""
    import scipy.ndimage as ndi

    rng = np.random.default_rng(77)
    cases = [((70, 130), 5, 40), ((129, 63), 1, 6), ((64, 62), 0, 3), ((200, 190), 16, 1000), ((33, 300), 5, 3),
             ((131, 125), 2, 2)]
    for shape, m, hi in cases:
        d2 = rng.integers(0, hi + 1, size=(2,) + shape).astype(np.int32)
        d2[1] = (ndi.uniform_filter(d2[1].astype(np.float64), 5) * 3).astype(np.int32)  # smooth ridges and plateaus
        d2[0][rng.random(shape) < 0.3] = 0
        mask = rng.random((2,) + shape) < 0.8
        pk = ops.peak_mask(ctx.asarray(d2), ctx.asarray(mask), m).numpy().astype(bool)
        for b in range(2):
            size = 2 * m + 1
            ref = (d2[b] == ndi.maximum_filter(d2[b], size=size, mode="constant")) & mask[b] & (d2[b] > 0)
            if m > 0:
                ref[:m, :] = False
                ref[-m:, :] = False
                ref[:, :m] = False
                ref[:, -m:] = False
            assert np.array_equal(pk[b], ref), (shape, m, b)


def test_watershed(ctx, ops, golden):
    from oracle import skops
    from oracle.watershed import watershed

    g = golden("c2c3_256")
    m, markers = g["mask"], g["markers"]
    dm, dmk = ctx.asarray(m), ctx.asarray(markers)
    d2, e = ops.edt(dm)
    ws = ops.watershed_edt(d2, dmk, dm, seeds_first=True).numpy()
    assert np.array_equal(ws, g["watershed"])
    # general float64 relief (heap flood) on the same seeded relief
    ws2 = ops.watershed(ctx.asarray(g["relief"]), dmk, dm).numpy()
    assert np.array_equal(ws2, g["watershed"])
    # random smooth reliefs with distinct marker values, no mask restriction
    rng = np.random.default_rng(11)
    from scipy import ndimage as ndi

    for i in range(6):
        H, W = rng.integers(20, 70, 2)
        img = ndi.gaussian_filter(rng.random((H, W)), 2.0)
        mk = np.zeros((H, W), np.int32)
        k = int(rng.integers(2, 9))
        ys, xs = rng.integers(0, H, k), rng.integers(0, W, k)
        mk[ys, xs] = np.arange(1, k + 1)
        mask = rng.random((H, W)) < 0.9 if i % 2 else np.ones((H, W), bool)
        out = ops.watershed(ctx.asarray(img), ctx.asarray(mk), ctx.asarray(mask)).numpy()
        assert np.array_equal(out, watershed(img, mk, mask=mask)), i


def test_watershed_every_flood_class(ctx, ops):
    """Components of every size class of the EDT flood -- bounding boxes of ~1.5 k, 3.5 k, 7 k, 20 k, 30 k pixels (the five
    LDS classes) and 45 k / 90 k pixels or a distance above the bucket limit (the HBM queue path) -- each with several
    markers, in one plane and as a batch of two, through ``watershed_edt(seeds_first=True)`` and the fused
    ``watershed_edt_cleared``; bit-identical to the oracle's heap flood of the seeded relief."""
    from oracle import skops
    from oracle.watershed import watershed

    H, W = 640, 1200
    yy, xx = np.mgrid[0:H, 0:W]

    def blobs(seed):
        rng = np.random.default_rng(seed)
        m = np.zeros((H, W), bool)
        # (centre y, centre x, half height, half width): chains of overlapping ellipses inside the box
        for cy, cx, hy, hx in ((40, 40, 16, 22), (40, 140, 25, 33), (60, 300, 35, 48), (150, 120, 65, 75),
                               (150, 420, 80, 92), (330, 150, 100, 110), (420, 600, 140, 160)):
            for _ in range(6):
                oy, ox = rng.integers(-hy // 2, hy // 2 + 1), rng.integers(-hx // 2, hx // 2 + 1)
                ry, rx = rng.integers(hy // 3, hy // 2 + 1), rng.integers(hx // 3, hx // 2 + 1)
                m |= ((yy - cy - oy) / ry) ** 2 + ((xx - cx - ox) / rx) ** 2 <= 1.0
        m[200:330, 700:880] = True  # a 130 x 180 rectangle: EDT up to 65 -> d2 > 2,048 buckets
        for cy, cx in ((50, 960), (50, 990 + seed)):  # two lobes of radius 30: box ~62 x 95 (the 8,192-pixel class)
            m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= 30 ** 2
        for cy in (250, 330):  # 2 x 2 disks of radius 44, 80 apart: box ~170 x 170, d2 < 2,048 (the 32,512-pixel class)
            for cx in (980, 1060):
                m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= 44 ** 2
        m[0] = m[-1] = False
        m[:, 0] = m[:, -1] = False
        return m

    masks = np.stack([blobs(5), blobs(6)])
    dm = ctx.asarray(masks)
    d2, _ = ops.edt(dm)
    refs = []
    mks = []
    for b in range(2):
        edt = skops.distance_transform_edt(masks[b])
        markers, _ = skops.peak_markers(edt, masks[b], 5)
        assert markers.max() >= 12
        mks.append(markers.astype(np.int32))
        refs.append(watershed(skops.seeded_flood_image(edt, markers), markers, mask=masks[b]))
    dmk = ctx.asarray(np.stack(mks))
    for b in range(2):  # one plane at a time
        got = ops.watershed_edt(d2[b:b + 1], dmk[b:b + 1], dm[b:b + 1], seeds_first=True).numpy()[0]
        assert np.array_equal(got, refs[b]), b
    got = ops.watershed_edt(d2, dmk, dm, seeds_first=True).numpy()
    assert np.array_equal(got, np.stack(refs))
    # the fused tail on the same planes
    nl = ctx.asarray(np.array([m.max() for m in mks], np.int32))
    scratch = ctx.empty(masks.shape, np.int32)
    lab, cnt = ops.watershed_edt_cleared(d2, dmk, dm, nl, int(max(m.max() for m in mks)), scratch)
    for b in range(2):
        cleared = skops.clear_border(refs[b])
        ref = skops.relabel_sequential(cleared) if cleared.max() > 0 else cleared
        assert np.array_equal(lab.numpy()[b], ref) and cnt.numpy()[b] == ref.max(), b
    # ... and through the marker-list entry point (the chain's): at this width (a multiple of 16) the stage works from the
    # mask's run tables instead of a parent plane -- statistics, every flood class incl. the HBM queue path, frame marks,
    # final mapping
    cap = int(max(np.count_nonzero(m) for m in mks)) + 8
    klist, kcount = np.zeros((2, cap), np.int32), np.zeros(2, np.int32)
    for b in range(2):
        idx = np.flatnonzero(mks[b])
        klist[b, :idx.size], kcount[b] = idx, idx.size
    lab2, cnt2 = ops.watershed_edt_cleared(d2, dmk, dm, nl, int(max(m.max() for m in mks)), ctx.empty(masks.shape, np.int32),
                                           marker_list=(ctx.asarray(klist), ctx.asarray(kcount)))
    assert np.array_equal(lab2.numpy(), lab.numpy()) and np.array_equal(cnt2.numpy(), cnt.numpy())


def test_watershed_skimage_golden_cases(ctx, ops, golden):
    """All 120 real scikit-image 0.18.3 cases of tests/golden/watershed_cases.npz (tie-heavy integer reliefs and
    -EDT inputs, with / without mask, connectivity 1 and 2) through the C ABI: bit-identical under the default
    tie policy ('exact'), and the cases in which 'raster' differs are exactly cases that were reported as tied."""
    g = golden("watershed_cases")
    n = int(g["n"])
    assert n == 120
    tied = differ = 0
    for i in range(n):
        img, mk, mask, conn = g[f"img_{i}"], g[f"markers_{i}"], g[f"mask_{i}"], int(g[f"conn_{i}"])
        dimg, dmk, dmask = ctx.asarray(np.ascontiguousarray(img, np.float64)), ctx.asarray(mk), ctx.asarray(mask)
        out = ops.watershed(dimg, dmk, dmask, connectivity=conn).numpy()
        assert np.array_equal(out, g[f"out_{i}"]), f"case {i} (connectivity {conn})"
        if conn == 1:
            flags = ctx.empty((1,), np.int32)
            fast = ops.watershed(dimg, dmk, dmask, ties="report", ties_out=flags).numpy()
            t = int(flags.numpy()[0])
            tied += t
            if not np.array_equal(fast, g[f"out_{i}"]):
                differ += 1
                assert t == 1, f"case {i}: the raster-order flood differs from scikit-image but no tie was reported"
                with pytest.raises(ValueError, match="equal-valued markers"):
                    ops.watershed(dimg, dmk, dmask, ties="refuse")
        else:
            with pytest.raises(ValueError, match="connectivity 2"):
                ops.watershed(dimg, dmk, dmask, connectivity=2, ties="raster")
    assert tied > 0 and differ > 0  # the corpus does exercise the heap-order artefact


def test_watershed_plain_edt_relief(ctx, ops, golden):
    """watershed(-edt, markers, mask) as SURVEY.md A.8 writes it (equal-valued peak markers are ubiquitous on an EDT):
    the bucket flood with the exact tie policy equals the real scikit-image output, in a batch with an untied plane."""
    from oracle import skops

    g = golden("c2c3_256")
    m, markers = g["mask"], g["markers"]
    # plane 1: the same mask with markers of ONE component only, distinct d2 -> no tie, the parallel result stands
    edt = skops.distance_transform_edt(m)
    m2 = np.zeros_like(markers)
    ys, xs = np.nonzero(markers)
    m2[ys[0], xs[0]] = 1
    dm = ctx.asarray(np.stack([m, m]))
    dmk = ctx.asarray(np.stack([markers, m2]))
    d2, _ = ops.edt(dm)
    flags = ctx.empty((2,), np.int32)
    ws = ops.watershed_edt(d2, dmk, dm, seeds_first=False, ties_out=flags).numpy()
    assert np.array_equal(ws[0], g["watershed_plain"])
    from oracle.watershed import watershed

    assert np.array_equal(ws[1], watershed(-edt, m2, mask=m))
    assert flags.numpy().tolist() == [1, 0]
    fast = ops.watershed_edt(d2, dmk, dm, seeds_first=False, ties="raster").numpy()
    assert np.array_equal(fast[1], ws[1]) and not np.array_equal(fast[0], g["watershed_plain"])
    # float64 relief entry point on the same input
    ws64 = ops.watershed(ctx.asarray(np.stack([-edt, -edt])), dmk, dm).numpy()
    assert np.array_equal(ws64[0], g["watershed_plain"]) and np.array_equal(ws64[1], ws[1])
    # connectivity 2 through the bucket entry point (sequential emulation)
    ws8 = ops.watershed_edt(d2[:1], dmk[:1], dm[:1], seeds_first=False, connectivity=2).numpy()[0]
    assert np.array_equal(ws8, watershed(-edt, markers, mask=m, connectivity=2))


def test_regionprops(ctx, ops, golden):
    from arcadia_microscopy_tools_amd import _hip

    for name, key in (("disks_80", "labels"), ("c2c3_256", "labels"), ("nd2_multichannel", "labels8")):
        g = golden(name)
        lab = g[key].astype(np.int32)
        K = int(lab.max())
        t = ops.regionprops(ctx.asarray(lab), K).numpy()[0]
        cols = {c: t[:, i] for i, c in enumerate(_hip.RP_COLS)}
        assert np.array_equal(cols["area"], g["rp_area"])
        assert np.array_equal(cols["area_convex"], g["rp_area_convex"]), name
        for i in range(4):
            assert np.array_equal(cols[f"bbox-{i}"], g[f"rp_bbox-{i}"])
        np.testing.assert_allclose(cols["centroid-0"], g["rp_centroid-0"], rtol=1e-12)
        np.testing.assert_allclose(cols["centroid-1"], g["rp_centroid-1"], rtol=1e-12)
        np.testing.assert_allclose(cols["perimeter"], g["rp_perimeter"], rtol=1e-12)
        np.testing.assert_allclose(cols["axis_major_length"], g["rp_axis_major_length"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(cols["axis_minor_length"], g["rp_axis_minor_length"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(cols["eccentricity"], g["rp_eccentricity"], atol=1e-6)
        np.testing.assert_allclose(cols["solidity"], g["rp_solidity"], rtol=1e-12)
        sym = np.isclose(np.abs(g["rp_orientation"]), np.pi / 4)
        np.testing.assert_allclose(cols["orientation"][~sym], g["rp_orientation"][~sym], atol=1e-8)
        np.testing.assert_allclose(np.abs(cols["orientation"][sym]), np.pi / 4)


def test_grey_morphology_u16_packed_kernel(ctx, ops):
    """uint16 erosion / dilation / opening / closing / top-hat through the packed small-footprint kernel (symmetric run
    footprints up to 15 x 15) against scipy (oracle/skops.py), on widths that are and are not multiples of 8, image
    sizes smaller than a tile and not multiples of it, every boundary the halo has to reflect, and a footprint that
    does NOT qualify (off-centre run) to make sure the generic kernel still takes it."""
    from oracle import skops
    from scipy import ndimage as ndi

    rng = np.random.default_rng(41)
    fps = {"cross": skops.cross3(), "sq3": np.ones((3, 3), np.uint8), "sq5": np.ones((5, 5), np.uint8),
           "sq7x3": np.ones((7, 3), np.uint8), "col5": np.ones((5, 1), np.uint8), "row7": np.ones((1, 7), np.uint8)}
    for r in (1, 2, 3, 5, 7):
        fps[f"disk{r}"] = skops.disk(r)
    diamond = (np.abs(np.arange(-4, 5))[:, None] + np.abs(np.arange(-4, 5))[None, :] <= 4).astype(np.uint8)
    fps["diamond4"] = diamond
    gap = skops.disk(3).copy()
    gap[1] = 0  # a footprint with an empty row
    fps["disk3_gap"] = gap
    for shape in ((70, 200), (33, 131), (160, 264), (9, 20)):
        img = rng.integers(0, 65536, (2,) + shape).astype(np.uint16)
        img[1] = (ndi.gaussian_filter(img[1].astype(float), 3) ).astype(np.uint16)
        d = ctx.asarray(img)
        for name, fp in fps.items():
            er, di = ops.erosion(d, fp).numpy(), ops.dilation(d, fp).numpy()
            for b in range(2):
                assert np.array_equal(er[b], skops.erosion(img[b], fp)), (shape, name, b, "erosion")
                assert np.array_equal(di[b], skops.dilation(img[b], fp)), (shape, name, b, "dilation")
        for fp in (skops.disk(2), skops.disk(7)):
            assert np.array_equal(ops.opening(d, fp).numpy()[0], skops.opening(img[0], fp))
            assert np.array_equal(ops.closing(d, fp).numpy()[1], skops.closing(img[1], fp))
            assert np.array_equal(ops.white_tophat(d, fp).numpy()[1], skops.white_tophat(img[1], fp))
    # the register kernel's seams: strips of 512 columns (a width of exactly one / two strips, one group more, a last
    # strip of a single group, a last group in lane 62), row segments of 64 / 128 rows (heights around the segment length), and the three
    # boundary modes it takes (grey morphology itself only uses 'reflect')
    from arcadia_microscopy_tools_amd import hipops as _h
    for shape in ((16, 512), (67, 1024), (130, 520), (200, 1000), (40, 1528), (129, 2048)):
        img = rng.integers(0, 65536, shape).astype(np.uint16)
        d = ctx.asarray(img)
        fps["col15"], fps["r3x11"] = np.ones((15, 1), np.uint8), np.ones((3, 11), np.uint8)
        for name in ("sq3", "col5", "row7", "sq7x3", "col15", "r3x11", "disk2", "disk5", "disk7", "diamond4", "disk3_gap"):
            fp = fps[name]
            assert np.array_equal(ops.erosion(d, fp).numpy(), skops.erosion(img, fp)), (shape, name, "erosion")
            assert np.array_equal(ops.dilation(d, fp).numpy(), skops.dilation(img, fp)), (shape, name, "dilation")
        for name in ("disk2", "disk7", "sq3"):  # the subtraction fused into the dilation
            assert np.array_equal(ops.white_tophat(d, fps[name]).numpy(), skops.white_tophat(img, fps[name])), (shape, name)
        for mode, cval in (("nearest", 0), ("constant", 0), ("constant", 40000)):
            for fp in (fps["disk2"], fps["disk7"], fps["sq7x3"]):
                got = _h._rank(d, fp, 0, mode, cval, None).numpy()
                assert np.array_equal(got, ndi.minimum_filter(img, footprint=fp, mode=mode, cval=cval)), (shape, mode)
                got = _h._rank(d, fp, 1, mode, cval, None).numpy()
                assert np.array_equal(got, ndi.maximum_filter(img, footprint=fp, mode=mode, cval=cval)), (shape, mode)
    # median through the register kernel (3 x 3, cross, disk(2), 5 x 5 with and without corners), images with many ties
    oct5 = np.ones((5, 5), np.uint8)
    oct5[0, 0] = oct5[0, 4] = oct5[4, 0] = oct5[4, 4] = 0
    med_fps = {"sq3": fps["sq3"], "cross": fps["cross"], "disk2": fps["disk2"], "sq5": fps["sq5"], "oct5": oct5}
    for shape in ((16, 512), (67, 1024), (130, 520), (33, 1016), (129, 2048)):
        img = rng.integers(0, 65536, shape).astype(np.uint16)
        few = rng.integers(0, 4, shape).astype(np.uint16) * 21845  # ties, and both ends of the uint16 range
        for im in (img, few):
            d = ctx.asarray(im)
            for name, fp in med_fps.items():
                assert np.array_equal(ops.median(d, fp).numpy(), skops.median(im, fp)), (shape, name, "median")
            for mode, cval in (("reflect", 0), ("constant", 0), ("constant", 40000)):
                got = ops.median(d, fps["disk2"], mode=mode, cval=cval).numpy()
                assert np.array_equal(got, ndi.median_filter(im, footprint=fps["disk2"], mode=mode, cval=cval)), (shape, mode)
    off = np.zeros((3, 5), np.uint8)
    off[0, 0:3] = 1
    off[1, 1:4] = 1
    off[2, 2:5] = 1  # runs that are not centred: generic path
    img = rng.integers(0, 65536, (40, 52)).astype(np.uint16)
    assert np.array_equal(ops.erosion(ctx.asarray(img), off).numpy(), ndi.grey_erosion(img, footprint=off))
    assert np.array_equal(ops.dilation(ctx.asarray(img), off).numpy(),
                          ndi.grey_dilation(img, footprint=off[::-1, ::-1]))


def test_convex_area_all_heights(ctx, ops):
    """area_convex for labels of every height class: both hull kernels (LDS chains for labels of <= 48 rows, HBM
    chains above), labels touching row 0, one-pixel and one-row labels, concave and fragmented shapes -- against the
    oracle's exact integer restatement of scikit-image's convex_hull_image (oracle/regionprops.py)."""
    from arcadia_microscopy_tools_amd import _hip
    from oracle.regionprops import convex_area_exact

    rng = np.random.default_rng(17)
    H, W = 300, 420
    lab = np.zeros((H, W), np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    k = 0
    shapes = [(20, 30, 9, 9), (0, 80, 14, 30), (60, 60, 24, 10), (70, 150, 25, 40), (150, 90, 49, 20),
              (160, 260, 70, 60), (290, 300, 9, 50)]
    for cy, cx, ry, rx in shapes:  # ellipses: 19 ... 141 rows tall; the second one is clipped by row 0
        k += 1
        lab[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = k
    k += 1
    lab[5, 400] = k                      # one pixel
    k += 1
    lab[40, 330:415] = k                 # one row
    k += 1
    lab[100:230, 400] = k                # one column, 130 rows
    k += 1
    lab[240:290, 10:60][np.triu(np.ones((50, 50), bool))] = k   # triangle, 50 rows (just above the LDS limit)
    k += 1
    lab[200:248, 340:390][np.tril(np.ones((48, 50), bool))] = k  # 48 rows (exactly the LDS limit)
    k += 1
    m = rng.random((40, 40)) < 0.3       # fragmented label: convex hull of scattered pixels
    lab[250:290, 100:140][m] = k
    t = ops.regionprops(ctx.asarray(np.stack([lab, lab[::-1].copy()])), k).numpy()
    col = _hip.RP_COLS.index("area_convex")
    for b, L in enumerate((lab, lab[::-1])):
        for l in range(1, k + 1):
            assert t[b, l - 1, col] == convex_area_exact(L == l), (b, l)


def test_regionprops_intensity(ctx, ops, golden):
    g = golden("c2c3_256")
    lab = g["labels"].astype(np.int32)
    K = int(lab.max())
    t = ops.regionprops_intensity(ctx.asarray(lab), ctx.asarray(g["fov"]), K).numpy()[0]
    for ci, name in enumerate(("brightfield", "dapi", "fitc", "tritc")):
        for j, k in enumerate(("intensity_mean", "intensity_max", "intensity_min", "intensity_std")):
            np.testing.assert_allclose(t[:, ci, j], g[f"rp_{k}_{name}"], rtol=1e-10, err_msg=f"{k}_{name}")


def test_label_sparse(ctx, ops, golden):
    from oracle import skops

    rng = np.random.default_rng(31)
    for shape, p in (((64, 64), 0.05), ((130, 257), 0.02), ((300, 301), 0.1), ((40, 40), 0.0)):
        m = rng.random(shape) < p
        m2 = np.stack([m, np.roll(m, 3, axis=1)])
        for conn in (1, 2):
            lab, cnt = ops.label_sparse(ctx.asarray(m2), connectivity=conn)
            for b in range(2):
                ref = skops.label(m2[b], conn)
                assert cnt.numpy()[b] == ref.max()
                assert np.array_equal(lab.numpy()[b], ref), (shape, conn, b)
    g = golden("c2c3_256")
    pk = ops.peak_mask(ops.edt(ctx.asarray(g["mask"]))[0], ctx.asarray(g["mask"]), 5)
    assert np.array_equal(ops.label_sparse(pk, 1)[0].numpy(), g["markers"])
    # overflow is reported, not silently wrong
    dense = np.ones((64, 64), bool)
    _, cnt = ops.label_sparse(ctx.asarray(dense), capacity=100)
    assert cnt.numpy()[0] == -1


def test_threshold_open_close_fused(ctx, ops):
    """The one-kernel '>' + opening + closing chain against the three separate operators of the oracle:
    odd sizes, widths beyond one 32-word column chunk, a footprint too large for the fused kernel, and the
    Gaussian's folded min / max against numpy."""
    from oracle import skops

    rng = np.random.default_rng(5)
    for shape, rad in (((70, 130), 1), ((200, 333), 2), ((96, 2200), 2), ((150, 190), 3), ((90, 100), 6)):
        img = rng.random(shape)
        img = skops.gaussian((img * 60000).astype(np.uint16), 1.5)
        thr = float(np.quantile(img, 0.55))
        fp = skops.disk(rad)
        ref = skops.binary_closing(skops.binary_opening(img > thr, fp), fp)
        d = ctx.asarray(img[None])
        got = ops.threshold_open_close(d, ctx.asarray(np.array([thr])), fp).numpy()[0]
        assert np.array_equal(got.astype(bool), ref), (shape, rad)
    u = rng.integers(0, 65535, (3, 97, 140)).astype(np.uint16)
    mm = ctx.empty((3, 2), np.float64)
    g = ops.gaussian(ctx.asarray(u), 2.0, minmax_out=mm).numpy()
    assert np.array_equal(mm.numpy(), np.stack([g.min(axis=(1, 2)), g.max(axis=(1, 2))], axis=1))
    thr_a = ops.threshold_otsu(ctx.asarray(g)).numpy()
    thr_b = ops.threshold_otsu(ctx.asarray(g), minmax=mm).numpy()
    assert np.array_equal(thr_a, thr_b)


def test_gaussian_lds_path(ctx, ops):
    """The direct-to-LDS Gaussian kernel (uint16 input, W a multiple of 8 and >= 256) against the oracle: several
    radii (tile halo 8 and 16), the three supported boundary modes, heights below / above one row chunk, widths with a
    partial last tile, a strided channel of a (C, Y, X) batch, and the fused min / max."""
    from oracle import skops

    rng = np.random.default_rng(17)
    for (h, w), sigma in (((40, 256), 2.0), ((300, 264), 2.0), ((77, 512), 0.6), ((130, 496), 3.0), ((64, 1024), 1.0)):
        u = rng.integers(0, 65535, (h, w)).astype(np.uint16)
        d = ctx.asarray(u)
        assert np.array_equal(ops.gaussian(d, sigma).numpy(), skops.gaussian(u, sigma)), (h, w, sigma)
    u = rng.integers(0, 65535, (96, 320)).astype(np.uint16)
    from scipy import ndimage as ndi

    for mode in ("nearest", "reflect", "mirror", "constant", "wrap"):  # the last two take the register-load kernel
        ref = ndi.gaussian_filter(u.astype(np.float64) * (1.0 / 65535), 2.0, mode=mode, cval=0.25)
        assert np.array_equal(ops.gaussian(ctx.asarray(u), 2.0, mode=mode, cval=0.25).numpy(), ref), mode
    batch = rng.integers(0, 65535, (3, 4, 72, 264)).astype(np.uint16)
    mm = ctx.empty((3, 2), np.float64)
    g = ops.gaussian(ctx.asarray(batch), 2.0, channel=2, minmax_out=mm).numpy()
    for b in range(3):
        assert np.array_equal(g[b], skops.gaussian(batch[b, 2], 2.0)), b
    assert np.array_equal(mm.numpy(), np.stack([g.min(axis=(1, 2)), g.max(axis=(1, 2))], axis=1))
