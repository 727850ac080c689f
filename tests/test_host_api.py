"""Host-side logic of the drop-in boundary (no GPU): construction rules, validation and error messages of
ImageOperation / Pipeline / MicroscopyImage / SegmentationMask / SegmentationModel mirror the reference's own
tests (RT/test_pipeline.py, RT/test_microscopy.py, RT/test_masks.py, RT/test_model.py); the C ABI library
loads and exports every symbol include/amt_hip.h declares; the product fails loudly without a GPU."""
import os
import re
import warnings

import numpy as np
import pytest

import arcadia_microscopy_tools_amd as amt
from arcadia_microscopy_tools_amd import _hip
from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, CHANNELS, DAPI, FITC, TRITC, Channel
from arcadia_microscopy_tools_amd.masks import DEFAULT_CELL_PROPERTY_NAMES, SegmentationMask
from arcadia_microscopy_tools_amd.microscopy import MicroscopyImage
from arcadia_microscopy_tools_amd.model import SegmentationModel
from arcadia_microscopy_tools_amd.operations import (apply_threshold, crop_to_center, rescale_by_percentile,
                                                     subtract_background_dog)
from arcadia_microscopy_tools_amd.pipeline import ImageOperation, Pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def double_intensity(x):
    return x * 2


def add_ten(x):
    return x + 10


# ---- C ABI ------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "amt_hip.h")).read()
    declared = set(re.findall(r"\b(amt_[a-z0-9_]+)\s*\(", header))
    declared -= {"amt_ctx"}
    lib = _hip.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libamt_hip.so does not export {name}"
    assert declared == set(_hip.exported_names()), declared ^ set(_hip.exported_names())
    assert b"gfx950" in lib.amt_version()


def test_no_gpu_fails_loudly():
    lib = _hip.load_library()
    if lib.amt_device_count() > 0:
        pytest.skip("a GPU is present")
    from arcadia_microscopy_tools_amd.device import Context

    with pytest.raises(amt.HipUnavailableError, match="MI355X"):
        Context(0)
    with pytest.raises(amt.HipUnavailableError):
        rescale_by_percentile(np.arange(16, dtype=np.uint16).reshape(4, 4))


# ---- ImageOperation / Pipeline (RT/test_pipeline.py) ----------------------------------------------------
def test_image_operation_contract():
    op = ImageOperation(np.add, 5)
    assert op.func == np.add and op.args == (5,) and op.kwargs == {}
    np.testing.assert_array_equal(op(np.array([1, 2, 3])), [6, 7, 8])
    assert ImageOperation(np.clip, a_min=0, a_max=100).kwargs == {"a_min": 0, "a_max": 100}
    assert "double_intensity" in repr(ImageOperation(double_intensity))
    with pytest.raises(AttributeError):
        op.func = add_ten
    with pytest.raises(AttributeError):
        del op.func
    assert ImageOperation(np.add, 5) == ImageOperation(np.add, 5) != ImageOperation(np.add, 10)
    assert hash(ImageOperation(np.add, 5, k=1)) == hash(ImageOperation(np.add, 5, k=1))
    assert ImageOperation(rescale_by_percentile).on_device and not ImageOperation(double_intensity).on_device


def test_pipeline_contract():
    with pytest.raises(ValueError, match="at least one operation"):
        Pipeline(operations=[])
    with pytest.raises(TypeError, match="callable"):
        Pipeline(operations=[1])
    with pytest.raises(ValueError, match="max_workers must be at least 1"):
        Pipeline(operations=[ImageOperation(add_ten)], max_workers=0)
    with pytest.warns(UserWarning, match="copy=True has no effect"):
        Pipeline(operations=[ImageOperation(add_ten)], copy=True, parallel=True)
    p = Pipeline(operations=(ImageOperation(double_intensity), ImageOperation(add_ten)))
    assert len(p) == 2 and p.copy is False and p.preserve_dtype is False and p.parallel is False
    img = np.array([1, 2, 3], dtype=np.uint16)
    out = p(img)
    np.testing.assert_array_equal(out, [12, 14, 16])
    assert out.dtype == np.uint16
    pf = Pipeline([ImageOperation(lambda x: x.astype(float) / x.max())], preserve_dtype=True)
    assert pf(img).dtype == np.uint16
    assert "Pipeline([" in repr(p) and "preserve_dtype=True" in repr(pf)
    # copy=True protects the input from in-place operations
    def inplace(x):
        x += 1
        return x
    src = np.zeros(3, np.uint16)
    Pipeline([ImageOperation(inplace)], copy=True)(src)
    assert src.sum() == 0
    # parallel mode
    stack = np.arange(24, dtype=np.uint16).reshape(3, 2, 4)
    pp = Pipeline([ImageOperation(double_intensity)], parallel=True, max_workers=2)
    np.testing.assert_array_equal(pp(stack), stack * 2)
    with pytest.raises(ValueError, match="Parallel mode requires at least 3D input"):
        pp(np.zeros((4, 4), np.uint16))
    ppd = Pipeline([ImageOperation(lambda x: x / 2.0)], parallel=True, preserve_dtype=True)
    assert ppd(stack).dtype == np.uint16


# ---- operations: validation that must not need a GPU (R/operations.py:34-38,85-88,122-132,199-211) -------
def test_operations_validation():
    img = np.arange(64, dtype=np.uint16).reshape(8, 8)
    with pytest.raises(ValueError, match="Invalid percentile range"):
        rescale_by_percentile(img, percentile_range=(50, 10))
    with pytest.raises(ValueError, match="Percentile must be between 0 and 100"):
        subtract_background_dog(img, percentile=101)
    with pytest.raises(ValueError, match="must be smaller than high_sigma"):
        subtract_background_dog(img, low_sigma=5, high_sigma=2)
    empty = np.zeros((0, 4), np.uint16)
    assert rescale_by_percentile(empty).dtype == float and rescale_by_percentile(empty).shape == (0, 4)
    assert apply_threshold(empty).dtype == bool
    # crop_to_center is pure slicing on numpy input (a view, as in the reference)
    big = np.arange(100).reshape(10, 10)
    c = crop_to_center(big, (4, 6))
    assert c.shape == (4, 6) and c.base is not None and c[0, 0] == big[3, 2]
    assert crop_to_center(big, (50, 50)).shape == (10, 10)
    assert crop_to_center(np.zeros((3, 9, 9)), (5, 3)).shape == (3, 5, 3)


# ---- MicroscopyImage (RT/test_microscopy.py + R/microscopy.py:115-131,241-282) ---------------------------
def _image(arr, chans, axes=None):
    return MicroscopyImage.from_array(arr, chans, axes)


def test_microscopy_image_contract():
    arr = np.arange(4 * 6 * 5, dtype=np.uint16).reshape(4, 6, 5)
    im = _image(arr, [BRIGHTFIELD, DAPI, FITC, TRITC])
    assert im.shape == (4, 6, 5) and im.sizes == {"C": 4, "Y": 6, "X": 5}
    assert im.num_channels == 4 and im.channel_axis == 0 and [c.name for c in im.channels][1] == "DAPI"
    d = im.get_channel_intensities(DAPI)
    assert d.shape == (6, 5) and np.shares_memory(d, arr) and np.array_equal(d, arr[1])
    assert np.array_equal(im.get_channel_intensities("FITC"), arr[2])
    with pytest.raises(ValueError, match="Channel 'CY5' not found in image"):
        im.get_channel_intensities("CY5")
    assert "MicroscopyImage" in repr(im) and "DAPI" in repr(im)
    # (T, C, Y, X): channel axis 1, temporal axis preserved
    t = np.zeros((3, 2, 4, 4), np.uint16)
    t[:, 1] = 7
    im4 = _image(t, [DAPI, FITC], "TCYX")
    assert im4.channel_axis == 1 and im4.get_channel_intensities(FITC).shape == (3, 4, 4)
    assert im4.get_channel_intensities(FITC).min() == 7
    # single channel returns everything
    s = _image(np.zeros((5, 4, 4), np.uint16), [DAPI], "TYX")
    assert s.get_channel_intensities(DAPI).shape == (5, 4, 4) and s.channel_axis is None
    with pytest.raises(ValueError, match="does not match metadata sizes"):
        MicroscopyImage(np.zeros((2, 2), np.uint16), im.metadata)
    with pytest.warns(amt.MetadataWarning, match="Expected uint16"):
        _image(np.zeros((4, 4), np.float32), [DAPI])
    with pytest.raises(ValueError, match="Number of channel metadata entries"):
        _image(np.zeros((2, 4, 4), np.uint16), [DAPI])
    out = im.apply_pipeline(Pipeline([ImageOperation(double_intensity)]), DAPI)
    assert np.array_equal(out, arr[1] * 2)


def test_channels():
    assert CHANNELS["DAPI"] is DAPI and DAPI.excitation_nm == 405
    with pytest.raises(ValueError, match="hex code"):
        Channel("X", "red")
    with pytest.raises(ValueError, match="excitation_nm must be positive"):
        Channel("X", "#FFF", excitation_nm=-1)
    assert hash(Channel("A", "#FFFFFF")) == hash(Channel("A", "#FFFFFF"))


def test_nd2lite_reads_golden_fixture(golden, tmp_path):
    """The ND2 chunk walk + lite-variant attributes, on a synthetic file written here (the reference's
    fixture itself is not available on the GPU box; its pixels are pinned in tests/golden)."""
    from conftest import write_synthetic_nd2

    from arcadia_microscopy_tools_amd import nd2lite

    px = golden("nd2_multichannel")["pixels"]  # (4, 256, 256)
    f = write_synthetic_nd2(tmp_path / "synthetic.nd2", px.transpose(1, 2, 0)[None])
    arr, meta = nd2lite.load_nd2(f, channels=[BRIGHTFIELD, DAPI, FITC, TRITC], use_device=False)
    assert arr.shape == (4, 256, 256) and arr.dtype == np.uint16 and np.array_equal(arr, px)
    assert meta.sizes == {"C": 4, "Y": 256, "X": 256}
    # padded rows (uiWidthBytes > X * C * 2, e.g. odd widths): the row stride is honoured, not assumed
    odd = px[:, :37, :45]
    fp = write_synthetic_nd2(tmp_path / "padded.nd2", np.stack([odd, odd[:, ::-1]]).transpose(0, 2, 3, 1), row_pad_bytes=6)
    arr2, meta2 = nd2lite.load_nd2(fp, channels=[BRIGHTFIELD, DAPI, FITC, TRITC], use_device=False)
    assert arr2.shape == (2, 4, 37, 45) and np.array_equal(arr2[0], odd) and np.array_equal(arr2[1], odd[:, ::-1])
    assert meta2.sizes == {"T": 2, "C": 4, "Y": 37, "X": 45}
    assert nd2lite.resolve_optical_config("Mono") is BRIGHTFIELD
    assert nd2lite.resolve_optical_config("GFP 488 nm") is FITC
    assert nd2lite.resolve_optical_config("FITC BP") is FITC and nd2lite.resolve_optical_config("zzz") is None


# ---- SegmentationMask validation (RT/test_masks.py:297-349) -----------------------------------------------
def test_segmentation_mask_validation():
    m = np.zeros((8, 8), np.int64)
    m[2:5, 2:5] = 1
    with pytest.raises(TypeError, match="mask_image must be a numpy array"):
        SegmentationMask(mask_image=[[0, 1]])
    with pytest.raises(ValueError, match="must be a 2D array"):
        SegmentationMask(mask_image=np.zeros((2, 2, 2), np.int64))
    with pytest.raises(ValueError, match="non-negative"):
        SegmentationMask(mask_image=-m)
    with pytest.raises(ValueError, match="contains no cells"):
        SegmentationMask(mask_image=np.zeros((4, 4), np.int64))
    with pytest.raises(TypeError, match="must be a Mapping"):
        SegmentationMask(mask_image=m, intensity_image_dict=[m])
    with pytest.raises(TypeError, match="Intensity image for 'DAPI' must be a numpy array"):
        SegmentationMask(mask_image=m, intensity_image_dict={DAPI: [[1]]})
    with pytest.raises(ValueError, match="must be 2D"):
        SegmentationMask(mask_image=m, intensity_image_dict={DAPI: np.zeros((2, 8, 8))})
    with pytest.raises(ValueError, match="same shape as mask_image"):
        SegmentationMask(mask_image=m, intensity_image_dict={DAPI: np.zeros((4, 4))})
    d = {DAPI: np.zeros((8, 8), np.uint16)}
    sm = SegmentationMask(mask_image=m, intensity_image_dict=d)
    assert sm.property_names == DEFAULT_CELL_PROPERTY_NAMES and sm.property_names is not DEFAULT_CELL_PROPERTY_NAMES
    assert sm.intensity_property_names == ["intensity_mean", "intensity_max", "intensity_min", "intensity_std"]
    assert sm.intensity_image_dict is not d and sm.intensity_image_dict[DAPI] is d[DAPI]
    assert SegmentationMask(mask_image=m).intensity_property_names == []
    with pytest.raises(AttributeError, match="Cannot modify 'mask_image'"):
        sm.mask_image = m
    with pytest.raises(AttributeError, match="Cannot modify 'remove_edge_cells'"):
        sm.remove_edge_cells = False


# ---- SegmentationModel parameter handling (RT/test_model.py:40-80,453-490) ----------------------------------
def test_segmentation_model_parameters():
    default = SegmentationModel()  # the reference's behaviour: Cellpose, device picked, nothing loaded yet
    assert default.backend == "cellpose" and default.device is not None and default._model is None
    model = SegmentationModel(backend="classical")
    assert (model.default_cell_diameter_px, model.default_flow_threshold, model.default_cellprob_threshold,
            model.default_num_iterations, model.default_batch_size) == (30, 0.4, 0, None, 8)
    p = model._resolve_and_validate_parameters(None, None, None, None, None)
    assert p == {"diameter": 30, "flow_threshold": 0.4, "cellprob_threshold": 0, "niter": None, "batch_size": 8}
    p = model._resolve_and_validate_parameters(50, 0.6, 1.0, 200, 16)
    assert p == {"diameter": 50, "flow_threshold": 0.6, "cellprob_threshold": 1.0, "niter": 200, "batch_size": 16}
    img = np.zeros((8, 8))
    with pytest.raises(ValueError, match="Cell diameter .* must be positive"):
        model.segment(img, cell_diameter_px=-5)
    with pytest.raises(ValueError, match="Flow threshold must be non-negative"):
        model.segment(img, flow_threshold=-0.1)
    with pytest.raises(ValueError, match="between -10 and 10"):
        model.segment(img, cellprob_threshold=11)
    with pytest.raises(ValueError, match="between -10 and 10"):
        model.batch_segment([img], cellprob_threshold=-11, show_progress=False)
    with pytest.raises(ValueError, match="backend must be"):
        SegmentationModel(backend="nope")
    # a failing backend surfaces as RuntimeError from segment() and as warning + None from batch_segment()
    lib = _hip.load_library()
    if lib.amt_device_count() == 0:
        with pytest.raises(RuntimeError, match="Cellpose segmentation failed"):
            model.segment(img)
        with pytest.warns(amt.SegmentationWarning, match="failed on image 0"):
            out = model.batch_segment([img, img], show_progress=False)
        assert out == [None, None]
    cp = SegmentationModel(backend="cellpose", device="cpu")
    with pytest.raises(RuntimeError, match="Failed to load Cellpose model|Cellpose segmentation failed"):
        cp.segment(img)


def test_plate_sharding_and_packing():
    from arcadia_microscopy_tools_amd import plate, synth

    parts = [plate.shard_indices(384, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(384)) and all(len(p) == 48 for p in parts)
    parts = [plate.shard_indices(10, r, 4) for r in range(4)]
    assert [len(p) for p in parts] == [3, 3, 2, 2] and sorted(sum(parts, [])) == list(range(10))
    with pytest.raises(ValueError):
        plate.shard_indices(4, 4, 4)
    assert synth.well_id(0) == "A01" and synth.well_id(24) == "B01" and synth.well_id(383) == "P24"
    cols = plate.table_columns(("DAPI", "FITC"))
    assert cols[:2] == ["fov_index", "label"] and cols[-1] == "intensity_std_fitc" and len(cols) == 2 + 12 + 8
    t = {c: np.arange(3, dtype=float) for c in cols[1:]}
    rows = plate.pack_rows([5], [t], ("DAPI", "FITC"))
    assert rows.shape == (3, len(cols)) and (rows[:, 0] == 5).all() and rows[2, 1] == 2


def test_plate_well_keyed_export():
    """Dense plate blocks -> packed rows -> well-keyed DataFrame (well ids in the
    reference's normalised form, R/microplate.py:24-45)."""
    from arcadia_microscopy_tools_amd import _hip, plate
    from arcadia_microscopy_tools_amd.segment import assemble_cell_properties

    assert [plate.well_id(i) for i in (0, 23, 24, 383)] == ["A01", "A24", "B01", "P24"]
    assert plate.well_id(9, n_columns=12, fovs_per_well=4) == "A03"
    with pytest.raises(ValueError):
        plate.well_id(26 * 24)
    rng = np.random.default_rng(0)
    F, K, chans = 4, 6, ["DAPI", "FITC"]
    table = rng.random((F, K, _hip.RP_NCOLS)) + 1.0
    itable = rng.random((F, K, 2, 4))
    ncells = np.array([3, 0, 6, 1])
    rows = plate.plate_rows(table, itable, ncells, chans, fov_indices=[0, 1, 24, 383])
    cols = plate.table_columns(chans)
    assert rows.shape == (10, len(cols))
    one = assemble_cell_properties(table[2, :6], itable[2, :6], chans)
    blk = rows[rows[:, 0] == 24]
    assert np.array_equal(blk[:, 1], np.arange(1, 7))
    for name in ("area", "circularity", "volume", "intensity_std_fitc"):
        assert np.array_equal(blk[:, cols.index(name)], one[name])
    with pytest.raises(ValueError, match="overflowed"):
        plate.plate_rows(table, itable, np.array([3, 0, 7, 1]), chans)

    class _Well:
        def __init__(self, sample, properties):
            self.sample, self.properties = sample, properties

    df = plate.plate_dataframe(rows, chans, layout={"A01": _Well("ctrl", {"dose": 1.0})})
    assert list(df.columns[:3]) == ["well_id", "fov_index", "label"] and len(df) == 10
    assert df["well_id"].tolist() == ["A01"] * 3 + ["B01"] * 6 + ["P24"]
    assert df["sample"].tolist()[:4] == ["ctrl", "ctrl", "ctrl", ""] and df["dose"].iloc[0] == 1.0


# ---- host side of the transfers: result blocks and chunking ----------------------------------------------------
def test_result_block_lives_as_long_as_its_last_view():
    import ctypes
    import gc

    from arcadia_microscopy_tools_amd import device as dv

    returned = []

    class Pool:
        def _put(self, ptr, nbytes):
            returned.append((ptr, nbytes))

    buf = ctypes.create_string_buffer(1 << 16)
    block = dv._PinnedBlock(ctypes.addressof(buf), 1 << 16, Pool())
    a = np.asarray(block)[: 64 * 64 * 8].view(np.int64).reshape(64, 64)
    assert a.flags.writeable and not a.flags.owndata
    row = a[3]
    del block, a
    gc.collect()
    assert returned == []  # a view is still alive
    row[:] = 7
    assert bytes(buf[3 * 64 * 8: 3 * 64 * 8 + 8]) == (7).to_bytes(8, "little")
    del row
    gc.collect()
    assert returned == [(ctypes.addressof(buf), 1 << 16)]


def test_transfer_chunks_cover_the_array_once():
    from arcadia_microscopy_tools_amd import device as dv

    for n in (1, 4095, 4096, 1 << 20, (1 << 22) + 17, 2048 * 2048):
        parts = dv._chunks(n)
        assert len(parts) <= dv._PIPE_CHUNKS and parts[0][0] == 0
        assert all(o % 4096 == 0 for o, _ in parts)
        assert sum(m for _, m in parts) == n and all(parts[i][0] + parts[i][1] == parts[i + 1][0]
                                                     for i in range(len(parts) - 1))


def test_label_plane_extrema_threaded_equals_numpy():
    from arcadia_microscopy_tools_amd.masks import _extrema

    rng = np.random.default_rng(5)
    for shape in ((7, 9), (1030, 1100), (2048, 2048)):
        a = rng.integers(-3, 1000, shape)
        assert _extrema(a) == (a.min(), a.max())
        assert _extrema(a[:, ::2]) == (a[:, ::2].min(), a[:, ::2].max())


# ---- Well / MicroplateLayout (RT/test_microplate.py) and the plate table keyed by them ----------------------------
def test_well_and_layout_follow_the_reference(tmp_path):
    from arcadia_microscopy_tools_amd.microplate import MicroplateLayout, Well

    w = Well(id="A01", sample="sample1")
    assert (w.id, w.sample, w.row, w.column, str(w)) == ("A01", "sample1", "A", 1, "A01")
    assert Well(id="a1").id == "A01" and Well("p24").id == "P24" and Well("H048").column == 48
    assert repr(Well("b2", "x")) == "Well(id='B02', sample='x')"
    assert repr(Well("b2", "x", {"c": 1})) == "Well(id='B02', sample='x', properties={'c': 1})"
    for bad, msg in (("A", "Well ID must be at least 2 characters"), ("", "Well ID must be at least 2 characters"),
                     ("1A", "Row must be A-Z, got '1'"), ("Ax", "Could not parse column number from 'Ax'"),
                     ("A49", "Column must be 1-48, got 49"), ("A0", "Column must be 1-48, got 0")):
        with pytest.raises(ValueError, match=re.escape(msg)):
            Well(id=bad)
    with pytest.raises(Exception):
        w.sample = "other"  # frozen
    well = Well.from_dict({"well_id": "B02", "sample": "test_sample", "concentration": 10})
    assert (well.id, well.sample, well.properties) == ("B02", "test_sample", {"concentration": 10})
    with pytest.raises(ValueError, match="Dictionary must contain 'well_id' key"):
        Well.from_dict({"sample": "s"})
    with pytest.raises(ValueError, match="well_id must be a string, got int"):
        Well.from_dict({"well_id": 7})

    layout = MicroplateLayout([Well(id="A01", sample="s1"), Well(id="B02", sample="s2", properties={"dose": 0.5})])
    assert len(layout) == 2 and "A01" in layout and "b2" in layout and "C03" not in layout and "??" not in layout
    assert layout["A1"].sample == "s1" and layout["B02"].properties == {"dose": 0.5}
    assert (layout.rows, layout.columns, layout.well_ids) == (["A", "B"], [1, 2], ["A01", "B02"])
    assert [x.id for x in layout] == ["A01", "B02"] and layout.layout["A01"] is layout["A01"]
    with pytest.raises(KeyError, match="not found in plate layout"):
        layout["C03"]
    with pytest.raises(KeyError, match="Invalid well ID 'A99'"):
        layout["A99"]
    with pytest.raises(ValueError, match="Duplicate well ID: 'A01'"):
        MicroplateLayout([Well(id="A01", sample="s1"), Well(id="a1", sample="s2")])
    df = layout.to_dataframe()
    assert len(df) == 2 and list(df.columns) == ["well_id", "row", "column", "sample", "dose"]
    assert df["well_id"].tolist() == ["A01", "B02"] and df["column"].tolist() == [1, 2]
    grid = layout.display().splitlines()
    assert grid[0].split() == ["column", "1", "2"] and grid[2].split() == ["A", "s1", "-"] and \
        grid[3].split() == ["B", "-", "s2"]
    assert MicroplateLayout([]).display() == "Empty plate layout" and MicroplateLayout([]).to_dataframe().empty

    csv = tmp_path / "plate.csv"
    csv.write_text("well_id,sample,dose\na1,dmso,0.0\nA02,drug,1.5\n")
    from_csv = MicroplateLayout.from_csv(csv)
    assert from_csv.well_ids == ["A01", "A02"] and from_csv["A2"].sample == "drug" and \
        from_csv["A02"].properties == {"dose": 1.5}
    (tmp_path / "nokey.csv").write_text("well,sample\nA1,x\n")
    with pytest.raises(ValueError, match="missing required 'well_id' column"):
        MicroplateLayout.from_csv(tmp_path / "nokey.csv")
    (tmp_path / "empty.csv").write_text("well_id,sample\n")
    with pytest.raises(ValueError, match="is empty"):
        MicroplateLayout.from_csv(tmp_path / "empty.csv")


def test_plate_table_keyed_by_microplate_layout():
    from arcadia_microscopy_tools_amd import plate
    from arcadia_microscopy_tools_amd.microplate import MicroplateLayout, Well

    names = ["DAPI", "FITC"]
    cols = plate.table_columns(names)
    rows = np.zeros((5, len(cols)))
    rows[:, cols.index("fov_index")] = [0, 0, 1, 25, 383]
    rows[:, cols.index("label")] = [1, 2, 1, 1, 1]
    rows[:, cols.index("area")] = [10, 20, 30, 40, 50]
    layout = MicroplateLayout([Well("A1", "dmso", {"dose": 0.0}), Well("A2", "drug", {"dose": 1.5}), Well("P24", "edge")])
    df = plate.plate_dataframe(rows, names, layout=layout)
    assert df["well_id"].tolist() == ["A01", "A01", "A02", "B02", "P24"]
    assert df["sample"].tolist() == ["dmso", "dmso", "drug", "", "edge"]
    assert df["dose"].tolist()[:3] == [0.0, 0.0, 1.5] and df["dose"].isna().tolist()[3:] == [True, True]
    assert df["area"].tolist() == [10, 20, 30, 40, 50] and df["label"].dtype == np.int64
    merged = df.merge(layout.to_dataframe()[["well_id", "row", "column"]], on="well_id", how="left")
    assert merged["row"].tolist() == ["A", "A", "A", np.nan, "P"] or merged["row"].isna().tolist() == [False, False, False, True, False]


def test_minimum_threshold_vector_search_equals_the_loop():
    """``_thresholds.minimum`` finds the histogram's local maxima in one vector pass; scikit-image (and the oracle)
    walk it with a direction flag.  Same indices on random histograms with plateaus, same threshold on an image."""
    from scipy import ndimage as ndi

    from arcadia_microscopy_tools_amd import _thresholds
    from oracle import skops

    rng = np.random.default_rng(0)
    img = (ndi.gaussian_filter(rng.random((96, 96)), 3) * 2000).astype(np.uint16)
    img[:48] += 3000
    counts, centers = skops._counts_centers(img, 256)
    assert _thresholds.minimum(counts, centers) == skops.threshold_minimum(img)
    flat = np.full((16, 16), 7, np.uint16)
    flat[0, 0] = 9
    c2, b2 = skops._counts_centers(flat, 256)
    with pytest.raises(RuntimeError, match="Unable to find two maxima"):
        _thresholds.minimum(c2, b2)
    with pytest.raises(RuntimeError, match="Unable to find two maxima"):
        skops.threshold_minimum(flat)


def test_img_as_float_plan_follows_skimage_per_dtype():
    from arcadia_microscopy_tools_amd import operations as op
    from oracle import skops

    vals = np.array([[0, 1, 2], [3, 100, 127]])
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64, np.bool_, np.float64):
        a = vals.astype(dt)
        b, scale = op._img_as_float_plan(a)
        if scale is not None:
            got = b.astype(np.float64) * scale
        elif b.dtype == np.uint16:
            got = b.astype(np.float64) * (1.0 / 65535)
        else:
            got = b
        assert got.dtype == np.float64 and np.array_equal(got, skops.img_as_float(a)), np.dtype(dt)
    assert [op._sauvola_r(d) for d in (np.uint8, np.uint16, np.int16, np.bool_, np.float32, np.float64)] == \
        [127.5, 32767.5, 32767.5, 0.5, 1.0, 1.0]


def test_nd2_loop_axes_come_from_the_experiment_tree(tmp_path):
    """(T, Z, C, Y, X) naming of multi-frame ND2 files: 'T' / 'Z' / 'P' from SLxExperiment.eType, outermost loop
    first, as ``nd2.ND2File.sizes`` reports them (R/nikon.py:197-210 derives the dimension flags from those)."""
    from conftest import write_synthetic_nd2

    from arcadia_microscopy_tools_amd import nd2lite
    from arcadia_microscopy_tools_amd.metadata_structures import DimensionFlags

    rng = np.random.default_rng(4)
    frames = rng.integers(0, 65536, (6, 10, 12, 2)).astype(np.uint16)
    chans = [DAPI, FITC]
    want = frames.transpose(0, 3, 1, 2)
    for loops, sizes, flags in (
            ([(1, 2), (4, 3)], {"T": 2, "Z": 3, "C": 2, "Y": 10, "X": 12}, DimensionFlags.TIMELAPSE | DimensionFlags.Z_STACK),
            ([(4, 6)], {"Z": 6, "C": 2, "Y": 10, "X": 12}, DimensionFlags.Z_STACK),
            ([(4, 6), (6, 2)], {"Z": 6, "C": 2, "Y": 10, "X": 12}, DimensionFlags.Z_STACK),   # spectral level: no axis
            ([(2, 3), (8, 2)], {"P": 3, "T": 2, "C": 2, "Y": 10, "X": 12}, DimensionFlags.MONTAGE | DimensionFlags.TIMELAPSE),
            ([(1, 1), (4, 6)], {"Z": 6, "C": 2, "Y": 10, "X": 12}, DimensionFlags.Z_STACK),   # one-step loop dropped
            ([(1, 4)], {"T": 6, "C": 2, "Y": 10, "X": 12}, DimensionFlags.TIMELAPSE),         # counts do not match
            (None, {"T": 6, "C": 2, "Y": 10, "X": 12}, DimensionFlags.TIMELAPSE)):            # no experiment chunk
        f = write_synthetic_nd2(tmp_path / "loops.nd2", frames, loops=loops)
        arr, meta = nd2lite.load_nd2(f, channels=chans, use_device=False)
        assert meta.sizes == sizes and arr.shape == tuple(sizes.values()), (loops, meta.sizes)
        assert all(cm.dimensions == flags for cm in meta.channel_metadata_list), loops
        assert np.array_equal(arr.reshape(want.shape), want), loops


def test_nd2_timelapse_without_a_period_still_loads(tmp_path):
    """A calibrated time-lapse whose loop states no period ('no delay' acquisitions, non-equidistant loops, frame
    counts taken as T): the optional resolution record is left out, the read does not fail; with a period it is
    filled.  Two files read one after the other do not inherit each other's steps."""
    from conftest import write_synthetic_nd2

    from arcadia_microscopy_tools_amd import nd2lite

    frames = np.arange(4 * 6 * 8 * 2, dtype=np.uint16).reshape(4, 6, 8, 2)
    with_period = write_synthetic_nd2(tmp_path / "tp.nd2", frames, loops=[(1, 4)], calibration=0.65, period_ms=250.0)
    arr, meta = nd2lite.load_nd2(with_period, channels=[DAPI, FITC], use_device=False)
    res = meta.channel_metadata_list[0].resolution
    assert arr.shape == (4, 2, 6, 8) and res.xy_step_um == 0.65 and res.t_size_px == 4 and res.t_step_ms == 250.0
    for loops in ([(1, 4)], [(8, 4)], None):
        f = write_synthetic_nd2(tmp_path / "t0.nd2", frames, loops=loops, calibration=0.65)
        arr, meta = nd2lite.load_nd2(f, channels=[DAPI, FITC], use_device=False)
        assert meta.sizes == {"T": 4, "C": 2, "Y": 6, "X": 8}
        assert all(cm.resolution is None for cm in meta.channel_metadata_list), loops
    single = write_synthetic_nd2(tmp_path / "s.nd2", frames[:1], calibration=0.65)
    _, meta = nd2lite.load_nd2(single, channels=[DAPI, FITC], use_device=False)
    assert meta.channel_metadata_list[0].resolution.xy_step_um == 0.65


def test_nd2lite_on_the_reference_fixtures_when_present():
    """Plumbing check against the files the reference's own tests read (RT/data, known-metadata.yml sizes); skipped on
    machines without /root/reference (the pixels of config 1 are pinned in tests/golden/nd2_multichannel.npz)."""
    import pathlib

    from arcadia_microscopy_tools_amd import nd2lite
    from arcadia_microscopy_tools_amd.metadata_structures import DimensionFlags

    data = pathlib.Path("/root/reference/src/arcadia_microscopy_tools/tests/data")
    if not (data / "example-zstack.nd2").exists():
        pytest.skip("reference fixtures are not on this machine")
    want = {"example-multichannel.nd2": ({"C": 4, "Y": 256, "X": 256}, DimensionFlags(0)),
            "example-pbmc.nd2": ({"C": 4, "Y": 256, "X": 256}, DimensionFlags(0)),
            "example-cerevisiae.nd2": ({"C": 2, "Y": 256, "X": 256}, DimensionFlags(0)),
            "example-timelapse.nd2": ({"T": 53, "Y": 64, "X": 64}, DimensionFlags.TIMELAPSE),
            "example-zstack.nd2": ({"Z": 11, "Y": 128, "X": 128}, DimensionFlags.Z_STACK)}
    # first channel of each file: (xy step um, z planes, z step um, magnification, NA, zoom, binning, exposure s) --
    # the values the reference's own RT/test_microscopy.py checks (RT/data/known-metadata.yml)
    known = {"example-multichannel.nd2": (0.323390342594048, 1, 1.0, 20.0, 0.75, 1.0, "1x1", 0.02),
             "example-timelapse.nd2": (0.325, 1, 1.0, 40.0, 0.95, 1.0, "2x2", 0.5),
             "example-zstack.nd2": (0.323390342594048, 11, 6.0, 20.0, 0.75, 1.0, "1x1", 0.5)}
    for name, (sizes, flags) in want.items():
        arr, meta = nd2lite.load_nd2(data / name, use_device=False)
        assert meta.sizes == sizes and arr.shape == tuple(sizes.values()) and arr.dtype == np.uint16, name
        cm = meta.channel_metadata_list[0]
        assert cm.dimensions == flags, name
        if name in known:
            xy, nz, dz, mag, na, zoom, binning, exposure = known[name]
            r, a, o = cm.resolution, cm.acquisition, cm.optics
            assert (r.x_size_px, r.y_size_px, r.z_size_px) == (sizes["X"], sizes["Y"], nz), name
            assert np.isclose(r.xy_step_um, xy) and np.isclose(r.z_step_um, dz), name
            assert (o.magnification, o.numerical_aperture) == (mag, na), name
            assert (a.zoom, a.binning) == (zoom, binning) and np.isclose(a.exposure_time_s, exposure), name
    multi = nd2lite.load_nd2(data / "example-multichannel.nd2", use_device=False)[1].channel_metadata_list
    assert [round(c.acquisition.exposure_time_s, 3) for c in multi] == [0.02, 1.0, 1.0, 1.0]  # per-channel exposure


def test_compat_install_aliases_the_reference_import_names():
    """Scripts written against the reference run unchanged after compat.install(): the reference's import names
    resolve to this package's modules (same objects), and uninstall() removes them again."""
    import importlib
    import sys

    from arcadia_microscopy_tools_amd import compat

    assert "arcadia_microscopy_tools" not in sys.modules
    compat.install()
    try:
        import arcadia_microscopy_tools as ref  # noqa: F401
        from arcadia_microscopy_tools import ImageOperation as RefOp, Pipeline as RefPipeline
        from arcadia_microscopy_tools.channels import DAPI as RefDapi
        from arcadia_microscopy_tools.masks import SegmentationMask as RefMask
        from arcadia_microscopy_tools.microplate import MicroplateLayout as RefLayout
        from arcadia_microscopy_tools.operations import rescale_by_percentile as ref_rescale

        from arcadia_microscopy_tools_amd.microplate import MicroplateLayout

        assert ref is amt and RefOp is ImageOperation and RefPipeline is Pipeline and RefDapi is DAPI
        assert RefMask is SegmentationMask and RefLayout is MicroplateLayout and ref_rescale is rescale_by_percentile
        assert importlib.import_module("arcadia_microscopy_tools.model").SegmentationModel is SegmentationModel
        assert callable(importlib.import_module("arcadia_microscopy_tools.nikon").load_nd2)
        assert callable(importlib.import_module("arcadia_microscopy_tools.utils").get_tqdm)
        with pytest.raises(ImportError):
            importlib.import_module("arcadia_microscopy_tools.leica")
    finally:
        compat.uninstall()
    assert "arcadia_microscopy_tools" not in sys.modules and "arcadia_microscopy_tools.masks" not in sys.modules


def test_blending_host_helpers_of_the_reference():
    """RT/test_blending.py exercises ``_blend_alpha`` / ``_blend_additive`` / ``_build_colormap`` / ``_gray_to_rgb``
    directly; same names and contracts here (host numpy, no GPU)."""
    from arcadia_microscopy_tools_amd.blending import _blend_additive, _blend_alpha, _build_colormap, _gray_to_rgb
    from oracle import blending as ob

    bg, fg = np.full((4, 5, 3), 0.25), np.full((4, 5, 3), 0.75)
    assert np.array_equal(_blend_alpha(bg, fg, np.zeros((4, 5, 1))), bg)
    assert np.array_equal(_blend_alpha(bg, fg, np.ones((4, 5, 1))), fg)
    assert np.allclose(_blend_alpha(bg, fg, np.full((4, 5, 1), 0.5)), 0.5)
    assert np.array_equal(_blend_additive(bg, fg, np.zeros((4, 5, 1))), bg)
    assert _blend_additive(fg, fg, np.ones((4, 5, 1))).max() == 1.0  # clipped
    assert np.array_equal(_blend_additive(bg, fg, np.ones((4, 5, 1))), _blend_additive(fg, bg, np.ones((4, 5, 1))))
    t, o = _build_colormap("#FF0000", True), _build_colormap("#FF0000", False)
    assert t is _build_colormap("#FF0000", True) and t is not o  # cached per (colour, transparency)
    assert t(np.array([0.0]))[0, 3] == 0.0 and o(np.array([0.0]))[0, 3] == 1.0
    assert np.array_equal(t(np.array([1.0]))[0], [1.0, 0.0, 0.0, 1.0])
    x = np.random.default_rng(0).random((6, 7))
    assert np.array_equal(t(x), ob.apply_lut(ob.build_lut("#FF0000", True), x))
    g = _gray_to_rgb(x)
    assert g.shape == (6, 7, 3) and all(np.array_equal(g[..., c], x) for c in range(3))


def test_metadata_records_validate_like_the_reference():
    """NominalDimensions / MeasuredDimensions tag fields with the dimension that needs them; ChannelMetadata validates
    its resolution against its flags (R/metadata_structures.py:14-31, :176-178)."""
    from arcadia_microscopy_tools_amd.metadata_structures import (AcquisitionSettings, ChannelMetadata, DimensionFlags,
                                                                  MeasuredDimensions, MicroscopeConfig, NominalDimensions)

    flat = NominalDimensions(x_size_px=8, y_size_px=9, xy_step_um=0.3)
    flat.validate(DimensionFlags(0))
    with pytest.raises(ValueError, match="z_size_px is required for Z_STACK"):
        flat.validate(DimensionFlags.Z_STACK)
    with pytest.raises(ValueError, match="t_size_px is required for TIMELAPSE"):
        ChannelMetadata(DAPI, dimensions=DimensionFlags.TIMELAPSE, resolution=flat)
    stack = NominalDimensions(8, 9, 0.3, z_size_px=5, z_step_um=2.0)
    cm = ChannelMetadata(DAPI, dimensions=DimensionFlags.Z_STACK, resolution=stack,
                         acquisition=AcquisitionSettings(exposure_time_s=0.1, binning="2x2"),
                         optics=MicroscopeConfig(magnification=20, numerical_aperture=0.75))
    assert cm.dimensions.is_zstack and not cm.dimensions.is_timelapse and cm.timestamp is None and cm.measured is None
    with pytest.raises(ValueError, match="z_values_um is required for Z_STACK"):
        MeasuredDimensions().validate(DimensionFlags.Z_STACK)
    assert ChannelMetadata(FITC).resolution is None  # arrays wrapped with from_array carry no acquisition metadata
