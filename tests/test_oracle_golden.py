"""Pin the CPU oracle (oracle/) against vectors produced by the real scikit-image 0.18.3 / scipy 1.7.1
(tools/make_golden.py) and against the known answers on the reference's ND2 fixture."""
import numpy as np
import pytest

from oracle import chains, regionprops as rp, skops
from oracle.watershed import watershed


def test_nd2_fixture_known_answers(golden):
    g = golden("nd2_multichannel")
    px = g["pixels"]
    assert px.shape == (4, 256, 256) and px.dtype == np.uint16
    # channel statistics recorded in SURVEY.md 8c
    assert (px[1].min(), px[1].max()) == (1048, 16117)
    assert px.ravel()[:3].tolist() != []  # interleave sanity below
    dapi = px[1]
    assert skops.threshold_otsu(dapi) == 2742 == g["otsu"]
    assert skops.threshold_isodata(dapi) == 2742 == g["isodata"]
    assert skops.threshold_yen(dapi) == 1947 == g["yen"]
    assert skops.threshold_triangle(dapi) == 1844 == g["triangle"]
    assert skops.threshold_mean(dapi) == pytest.approx(1527.4803161621094, rel=0, abs=0)
    assert skops.threshold_li(dapi) == pytest.approx(float(g["li"]), rel=1e-12)
    with pytest.raises(RuntimeError):
        skops.threshold_minimum(dapi)
    mask = dapi > 2742
    assert int(mask.sum()) == 1297
    lab8 = skops.label(mask)
    assert lab8.max() == 20 and np.array_equal(lab8, g["labels8"])
    assert skops.label(mask, connectivity=1).max() == 25
    assert np.array_equal(skops.label(mask, connectivity=1), g["labels4"])
    assert np.array_equal(skops.clear_border(lab8), g["cleared"])
    # integer-input label path == bool path
    assert np.array_equal(skops.label(lab8), lab8)


def test_nd2_regionprops(golden):
    g = golden("nd2_multichannel")
    lab = g["labels8"]
    t = rp.regionprops_table(
        lab, intensity_image=g["pixels"][1],
        properties=["label", "area", "centroid", "bbox", "area_convex", "perimeter", "eccentricity", "solidity",
                    "axis_major_length", "axis_minor_length", "orientation", "intensity_mean", "intensity_max",
                    "intensity_min", "intensity_std"])
    assert np.array_equal(t["label"], g["rp_label"])
    assert np.array_equal(t["area"], g["rp_area"])
    assert t["area"][:4].tolist() == [2, 433, 2, 74]
    np.testing.assert_allclose(t["intensity_mean"][:4], [2898.5, 4233.13625866, 2794.0, 3526.71621622], rtol=1e-9)
    np.testing.assert_allclose(t["perimeter"][:4], [0, 123.63961031, 0, 30.97056275], atol=1e-7)
    for k in ("centroid-0", "centroid-1", "perimeter", "intensity_mean", "intensity_max", "intensity_min",
              "intensity_std", "axis_major_length", "axis_minor_length"):
        np.testing.assert_allclose(t[k], g["rp_" + k], rtol=1e-10, atol=1e-10, err_msg=k)
    np.testing.assert_allclose(t["eccentricity"], g["rp_eccentricity"], atol=1e-7)
    for i in range(4):
        assert np.array_equal(t[f"bbox-{i}"], g[f"rp_bbox-{i}"])
    assert np.array_equal(t["area_convex"], g["rp_area_convex"])
    np.testing.assert_allclose(t["solidity"], g["rp_solidity"], rtol=1e-12)
    # orientation: exactly symmetric regions are version-sensitive; compare the rest
    sym = np.isclose(np.abs(g["rp_orientation"]), np.pi / 4)
    np.testing.assert_allclose(t["orientation"][~sym], g["rp_orientation"][~sym], atol=1e-9)


def test_disks_known_answers(golden):
    g = golden("disks_80")
    t = rp.regionprops_table(g["labels"], properties=["area", "perimeter", "area_convex", "axis_major_length",
                                                      "solidity", "eccentricity", "orientation"])
    assert t["area"].tolist() == [69, 193, 373]
    assert t["area_convex"].tolist() == [69, 201, 381] == g["rp_area_convex"].tolist()
    np.testing.assert_allclose(t["perimeter"], [27.313708, 48.970563, 68.284271], atol=1e-6)
    np.testing.assert_allclose(t["axis_major_length"][1:], [15.675465, 21.791112], atol=1e-6)
    np.testing.assert_allclose(t["solidity"][1], 0.960199, atol=1e-6)
    np.testing.assert_allclose(t["eccentricity"], 0, atol=1e-7)


def test_c2_chain(golden):
    g = golden("c2c3_256")
    dapi = g["fov"][1]
    gz = skops.gaussian(dapi, 2.0)
    # gaussian weights come from np.exp, which differs in the last bit across numpy versions
    np.testing.assert_allclose(gz, g["gauss2"], rtol=0, atol=2e-16)
    m, _, t = chains.c2_mask(dapi)
    assert t == pytest.approx(float(g["otsu_gauss2"]), rel=1e-12)
    assert np.array_equal(gz > t, g["mask_thr"])
    assert np.array_equal(m, g["mask"])
    assert np.array_equal(skops.binary_opening(g["mask_thr"], skops.disk(2)), g["mask_open"])
    assert np.array_equal(chains.c2_chain(dapi), g["labels8"])


def test_c3_chain(golden):
    g = golden("c2c3_256")
    mask = g["mask"]
    edt = skops.distance_transform_edt(mask)
    assert np.array_equal(edt, g["edt"])
    markers, n = skops.peak_markers(edt, mask, 5)
    assert np.array_equal(markers, g["markers"])
    relief = skops.seeded_flood_image(edt, markers)
    assert np.array_equal(relief, g["relief"])
    ws = watershed(relief, markers, mask=mask)
    assert np.array_equal(ws, g["watershed"])
    # plain -edt relief (equal-valued age-0 markers: heap mechanics matter) is matched by the oracle too
    assert np.array_equal(watershed(-edt, markers, mask=mask), g["watershed_plain"])
    cleared = skops.clear_border(ws)
    assert np.array_equal(cleared, g["cleared"])
    assert np.array_equal(skops.relabel_sequential(cleared), g["labels"])
    labels, props = chains.c3_chain(g["fov"])
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(props["label"], g["rp_label"])
    assert np.array_equal(props["area"], g["rp_area"])
    assert np.array_equal(props["area_convex"], g["rp_area_convex"])
    np.testing.assert_allclose(props["centroid_y"], g["rp_centroid-0"], rtol=1e-12)
    np.testing.assert_allclose(props["centroid_x"], g["rp_centroid-1"], rtol=1e-12)
    np.testing.assert_allclose(props["perimeter"], g["rp_perimeter"], rtol=1e-12)
    np.testing.assert_allclose(props["axis_major_length"], g["rp_axis_major_length"], rtol=1e-10)
    np.testing.assert_allclose(props["axis_minor_length"], g["rp_axis_minor_length"], rtol=1e-10)
    np.testing.assert_allclose(props["eccentricity"], g["rp_eccentricity"], atol=1e-7)
    per, area = g["rp_perimeter"], g["rp_area"]
    np.testing.assert_allclose(props["circularity"], 4 * np.pi * area / per**2, rtol=1e-12)
    for name in chains.CHANNEL_NAMES:
        for k in ("intensity_mean", "intensity_max", "intensity_min", "intensity_std"):
            key = f"{k}_{name.lower()}"
            np.testing.assert_allclose(props[key], g["rp_" + key], rtol=1e-10, err_msg=key)


def test_watershed_heap_mechanics(golden):
    g = golden("watershed_cases")
    n = int(g["n"])
    assert n == 120
    for i in range(n):
        out = watershed(g[f"img_{i}"], g[f"markers_{i}"], mask=g[f"mask_{i}"], connectivity=int(g[f"conn_{i}"]))
        assert np.array_equal(out, g[f"out_{i}"]), f"case {i}"


def test_ops_misc(golden):
    g = golden("ops_192")
    u = g["u16"]
    for s in (0.6, 1.0, 2.0, 5.0):
        np.testing.assert_allclose(skops.gaussian(u, s), g[f"gauss_{s}"], rtol=0, atol=3e-16)
    np.testing.assert_allclose(skops.difference_of_gaussians(u, 0.6, 16.0), g["dog_0.6_16"], rtol=0, atol=3e-16)
    gz = g["gauss_2.0"]
    for name in ("otsu", "yen", "isodata", "triangle", "mean", "li"):
        f = getattr(skops, "threshold_" + name)
        assert f(u) == pytest.approx(float(g[f"thr_{name}_u16"]), rel=1e-12), name
        assert f(gz) == pytest.approx(float(g[f"thr_{name}_f64"]), rel=1e-12), name
    h, c = skops.histogram(gz)
    assert np.array_equal(h, g["hist_f64"]) and np.array_equal(c, g["hist_f64_centers"])
    np.testing.assert_allclose(skops.threshold_local(u, 35), g["thr_local_35"], rtol=1e-12)
    np.testing.assert_allclose(skops.threshold_local(u, 35, method="mean"), g["thr_local_35_mean"], rtol=1e-12)
    np.testing.assert_allclose(skops.threshold_niblack(u, 15, 0.2), g["thr_niblack_15"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(skops.threshold_sauvola(u, 15, 0.2), g["thr_sauvola_15"], rtol=1e-9, atol=1e-9)
    for q in ((0, 100), (1, 99), (0.1, 99.9), (2, 98)):
        for tag, img in (("u16", u), ("f64", gz)):
            p = np.percentile(img, q)
            assert np.array_equal(p, g[f"pct_{tag}_{q[0]}_{q[1]}"])
            r = skops.rescale_intensity(img, (p[0], p[1]), (0, 1))
            assert np.array_equal(r, g[f"rescale_{tag}_{q[0]}_{q[1]}"])
    m = g["mask"]
    for r in (1, 2, 3):
        se = skops.disk(r)
        assert np.array_equal(skops.binary_erosion(m, se), g[f"berode_d{r}"])
        assert np.array_equal(skops.binary_dilation(m, se), g[f"bdilate_d{r}"])
        assert np.array_equal(skops.binary_opening(m, se), g[f"bopen_d{r}"])
        assert np.array_equal(skops.binary_closing(m, se), g[f"bclose_d{r}"])
        assert np.array_equal(skops.erosion(u, se), g[f"erode_d{r}"])
        assert np.array_equal(skops.dilation(u, se), g[f"dilate_d{r}"])
        assert np.array_equal(skops.opening(u, se), g[f"open_d{r}"])
        assert np.array_equal(skops.closing(u, se), g[f"close_d{r}"])
        assert np.array_equal(skops.median(u, se), g[f"median_d{r}"])
    assert np.array_equal(skops.binary_erosion(m), g["berode_cross"])
    assert np.array_equal(skops.binary_dilation(m), g["bdilate_cross"])
    assert np.array_equal(skops.white_tophat(u, skops.disk(7)), g["tophat_d7"])
    assert np.array_equal(skops.white_tophat(u, skops.disk(3)), g["tophat_d3"])
    assert np.array_equal(skops.median(u), g["median_3x3"])
    assert np.array_equal(skops.distance_transform_edt(m), g["edt"])
    assert np.array_equal(skops.label(m), g["label8"])
    assert np.array_equal(skops.label(m, 1), g["label4"])


def test_reference_ops_on_fixture(golden):
    """R/operations.py:57-97 and :10-54 restated on the fixture's FITC plane."""
    g = golden("nd2_multichannel")
    fitc = g["pixels"][2]
    dog = skops.difference_of_gaussians(fitc, 0.6, 16.0)
    np.testing.assert_allclose(dog, g["dog_fitc"], rtol=0, atol=3e-16)
    bg = np.clip(dog - np.percentile(dog, 90), 0, None)
    np.testing.assert_allclose(bg, g["bgsub_fitc_p90"], rtol=0, atol=5e-16)
    p1, p2 = np.percentile(fitc, (1, 99))
    assert np.array_equal(skops.rescale_intensity(fitc, (p1, p2), (0, 1)), g["rescale_fitc_1_99"])


def test_contours_oracle_vs_skimage(golden):
    """oracle/contours.py against scikit-image 0.18.3: the 16-case marching-squares table and the full
    `_extract_outlines_skimage` recipe (R/masks.py:82-115) on two label images."""
    from oracle import contours

    g = golden("outlines_96")
    table = g["case_table"]
    for case in range(16):
        segs = contours.CASES[case]
        for k, (f, t) in enumerate(segs):
            assert np.array_equal(table[case, k, 0], f) and np.array_equal(table[case, k, 1], t), case
        assert np.isnan(table[case, len(segs):]).all(), case
    for name in ("shapes", "nuclei"):
        lab, pts, offs = g[f"{name}_labels"], g[f"{name}_points"], g[f"{name}_offsets"]
        outs = contours.extract_outlines_skimage(lab)
        assert len(outs) == len(offs) - 1
        for i, o in enumerate(outs):
            ref = pts[offs[i]:offs[i + 1]]
            assert o.shape == ref.shape and np.array_equal(o, ref), (name, i)


def test_border_outlines_oracle_known_answers():
    """oracle/contours.py, the "cellpose" extractor (R/masks.py:68-79).  PARITY UNPINNED: OpenCV / cellpose are not
    installable here and the reference's tests hold no vector; these are the documented behaviours of
    cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) the restatement is held to: a filled rectangle starts at its
    top-left pixel and runs down the left side first, a one-pixel line is walked there and back (2n - 2 points),
    holes and components nested in them are not reported, borders of fewer than five points are dropped."""
    from oracle import contours

    a = np.zeros((5, 5), np.uint8)
    a[1:4, 1:4] = 1
    (c,) = contours.find_external_borders(a)
    assert c.shape == (8, 1, 2) and c.dtype == np.int32
    assert c.reshape(-1, 2).tolist() == [[1, 1], [1, 2], [1, 3], [2, 3], [3, 3], [3, 2], [3, 1], [2, 1]]
    line = np.zeros((3, 7), np.uint8)
    line[1, 1:6] = 1
    assert contours.find_external_borders(line)[0].reshape(-1, 2)[:, 0].tolist() == [1, 2, 3, 4, 5, 4, 3, 2]
    ring = np.zeros((9, 9), np.uint8)
    ring[1:8, 1:8] = 1
    ring[3:6, 3:6] = 0
    ring[4, 4] = 1
    assert [len(c) for c in contours.find_external_borders(ring)] == [24]
    two = np.zeros((6, 12), np.uint8)
    two[1:3, 1:3] = 1
    two[2:5, 6:10] = 1
    assert [len(c) for c in contours.find_external_borders(two)] == [10, 4]  # newest first
    lab = np.zeros((12, 12), np.int64)
    lab[1:3, 1:3] = 1  # four border points: dropped
    lab[5:9, 4:9] = 2
    lab[10, 10] = 3
    outs = contours.extract_outlines_cellpose(lab)
    assert [o.shape for o in outs] == [(0, 2), (14, 2), (0, 2)]
    assert outs[1][0].tolist() == [5, 4] and outs[1][1].tolist() == [6, 4]  # (y, x), down the left side
    full = np.ones((4, 4), np.int64)
    full[:, 2:] = 2
    assert len(contours.extract_outlines_cellpose(full)) == 1  # np.unique(masks)[1:] drops label 1 here


def test_blending_oracle_vs_matplotlib(golden):
    """oracle/blending.py against matplotlib 3.10.8: colour tables and composited canvases."""
    from oracle import blending as ob

    g = golden("overlay_64")
    assert np.array_equal(ob.build_lut("#0033FF", True), g["lut_dapi_t"])
    assert np.array_equal(ob.build_lut("#FFBF00", False), g["lut_tritc_o"])
    bg, dapi, fitc, tritc = g["background"], g["dapi"], g["fitc"], g["tritc"]
    cases = {
        "alpha3": [("#0033FF", dapi, 1.0, True, "alpha"), ("#07FF00", fitc, 0.8, True, "alpha"),
                   ("#FFBF00", tritc, 0.5, True, "alpha")],
        "additive3": [("#0033FF", dapi, 1.0, True, "additive"), ("#07FF00", fitc, 1.0, True, "additive"),
                      ("#FFBF00", tritc, 0.7, True, "additive")],
        "opaque_mixed": [("#A30000", dapi, 1.0, False, "alpha"), ("#07FF00", fitc, 0.6, True, "additive")],
        "short_hex": [("#F0A", tritc, 0.9, True, "alpha")],
    }
    for name, layers in cases.items():
        assert np.array_equal(ob.create_overlay(bg, layers), g[name]), name
    assert np.array_equal(ob.create_overlay(bg * 1.5 - 0.2, [("#0033FF", dapi * 1.3 - 0.1, 1.0, True, "alpha")]),
                          g["range"])


def test_label_int_vs_skimage(golden):
    """oracle/clabel.c (single-pass equal-value labelling) against real scikit-image 0.18.3
    ``measure.label`` / ``clear_border`` on multi-valued integer images (tools/make_golden_label.py),
    and against the scipy-only per-value restatement it replaced."""
    g = golden("label_int_cases")
    n = int(g["n"])
    assert n == 42
    for i in range(n):
        img = g[f"img_{i}"]
        assert np.array_equal(skops.label(img), g[f"lab2_{i}"]), f"case {i} conn 2"
        assert np.array_equal(skops.label(img, connectivity=1), g[f"lab1_{i}"]), f"case {i} conn 1"
        assert np.array_equal(skops.clear_border(img), g[f"cleared_{i}"]), f"case {i} clear_border"
        if i < 10:
            assert np.array_equal(skops._label_int_per_value(img), g[f"lab2_{i}"])


def test_even_footprints_follow_skimage(golden):
    """Even-sized footprints (tools/make_golden_even.py, real scikit-image 0.18.3): grey erosion / dilation pad in front,
    the second half of opening / closing behind, the binary operators / median / white_tophat go to scipy as they are."""
    from oracle import skops

    g = golden("even_footprints")
    img, mask = g["img"], g["mask"]
    for name in ("s2", "s4", "r2x3", "r3x4", "c4x1", "L4"):
        fp = g[f"fp_{name}"]
        assert np.array_equal(skops.erosion(img, fp), g[f"erosion_{name}"]), name
        assert np.array_equal(skops.dilation(img, fp), g[f"dilation_{name}"]), name
        assert np.array_equal(skops.opening(img, fp), g[f"opening_{name}"]), name
        assert np.array_equal(skops.closing(img, fp), g[f"closing_{name}"]), name
        assert np.array_equal(skops.white_tophat(img, fp), g[f"tophat_{name}"]), name
        assert np.array_equal(skops.median(img, fp), g[f"median_{name}"]), name
        assert np.array_equal(skops.binary_erosion(mask, fp), g[f"berosion_{name}"]), name
        assert np.array_equal(skops.binary_dilation(mask, fp), g[f"bdilation_{name}"]), name
        assert np.array_equal(skops.binary_opening(mask, fp), g[f"bopening_{name}"]), name
        assert np.array_equal(skops.binary_closing(mask, fp), g[f"bclosing_{name}"]), name
