"""Minimal reader for uncompressed modern ND2 files: pixel blocks + channel names, no third-party package.

The reference loads ND2 through the ``nd2`` package (R/nikon.py:25-43: ``nd2.ND2File.asarray()`` plus a
large metadata parser).  The hot path only needs the pixels and the channel list, so this module walks the
chunk map itself (layout: SURVEY.md A.10), decodes the "lite variant" attribute blocks far enough to get
width / height / component count / frame count and the optical-configuration names, and de-interleaves the
(Y, X, C) frames into (C, Y, X) on the GPU (``hipops.deinterleave``) or, for plumbing without a GPU, with a
numpy transpose.  The loop axes (time / z / position) are named from the experiment tree.  Compressed frames are out
of scope (row "next 1").
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

from .channels import BRIGHTFIELD, CHANNELS, FITC, Channel
from .metadata_structures import (AcquisitionSettings, ChannelMetadata, DimensionFlags, MicroscopeConfig,
                                  NominalDimensions)

_MAGIC = 0x0ABECEDA
_MAP_SIG = b"ND2 CHUNK MAP SIGNATURE 0000001!"

_ALIASES = {"MONO": BRIGHTFIELD, "GFP": FITC}


def resolve_optical_config(name: str) -> Channel | None:
    """Optical-configuration name -> predefined channel (same matching order as R/nikon.py:52-76)."""
    key = name.upper()
    if key in CHANNELS:
        return CHANNELS[key]
    for alias, ch in _ALIASES.items():
        if alias in key:
            return ch
    matches = [n for n in CHANNELS if n in key]
    if matches:
        return CHANNELS[max(matches, key=len)]
    return None


def read_chunk_map(data: bytes) -> dict[bytes, tuple[int, int]]:
    """{chunk name: (file offset of the chunk header, payload size)} from the map at the end of the file."""
    (map_off,) = struct.unpack("<Q", data[-8:])
    magic, name_len, data_len = struct.unpack("<IIQ", data[map_off:map_off + 16])
    if magic != _MAGIC:
        raise ValueError("not a modern ND2 file (bad chunk-map magic)")
    payload = data[map_off + 16 + name_len: map_off + 16 + name_len + data_len]
    out: dict[bytes, tuple[int, int]] = {}
    pos = 0
    while pos < len(payload):
        end = payload.find(b"!", pos)
        if end < 0:
            break
        name = payload[pos:end + 1]
        if name == _MAP_SIG:
            break
        off, size = struct.unpack("<QQ", payload[end + 1:end + 17])
        out[name] = (off, size)
        pos = end + 17
    return out


def chunk_payload(data: bytes, entry: tuple[int, int]) -> bytes:
    off, _ = entry
    magic, name_len, data_len = struct.unpack("<IIQ", data[off:off + 16])
    if magic != _MAGIC:
        raise ValueError("corrupt ND2 chunk header")
    start = off + 16 + name_len
    return data[start:start + data_len]


def parse_lite_variant(buf: bytes, count: int | None = None, pos: int = 0):
    """Decode a CLxLiteVariant block into nested dicts (types: 1 bool, 2 i32, 3 u32, 4 i64, 5 u64, 6 f64,
    7 pointer, 8 utf-16 string, 9 byte array, 11 level)."""
    out: dict[str, object] = {}
    n = 0
    while pos < len(buf) and (count is None or n < count):
        if pos + 2 > len(buf):
            break
        typ, nlen = buf[pos], buf[pos + 1]
        pos += 2
        name = buf[pos:pos + 2 * nlen].decode("utf-16-le", "replace").rstrip("\x00")
        pos += 2 * nlen
        if typ == 1:
            val = bool(buf[pos])
            pos += 1
        elif typ in (2, 3):
            val = struct.unpack("<i" if typ == 2 else "<I", buf[pos:pos + 4])[0]
            pos += 4
        elif typ in (4, 5, 7):
            val = struct.unpack("<q" if typ == 4 else "<Q", buf[pos:pos + 8])[0]
            pos += 8
        elif typ == 6:
            val = struct.unpack("<d", buf[pos:pos + 8])[0]
            pos += 8
        elif typ == 8:
            end = pos
            while end + 1 < len(buf) and buf[end:end + 2] != b"\x00\x00":
                end += 2
            val = buf[pos:end].decode("utf-16-le", "replace")
            pos = end + 2
        elif typ == 9:
            (blen,) = struct.unpack("<Q", buf[pos:pos + 8])
            val = buf[pos + 8:pos + 8 + blen]
            pos += 8 + blen
        elif typ == 11:
            item_count, length = struct.unpack("<IQ", buf[pos:pos + 12])
            # `length` counts from the start of this entry's header to the end of its items; a table of
            # item_count 8-byte offsets follows the items
            header = 2 + 2 * nlen
            start = pos + 12
            val, _ = parse_lite_variant(buf, item_count, start)
            pos = pos - header + length + 8 * item_count
        else:
            raise ValueError(f"unknown lite-variant type {typ}")
        if name in out:
            if not isinstance(out[name], list):
                out[name] = [out[name]]
            out[name].append(val)
        else:
            out[name] = val
        n += 1
    return out, pos


def read_attributes(data: bytes, cmap) -> dict[str, int]:
    attrs, _ = parse_lite_variant(chunk_payload(data, cmap[b"ImageAttributesLV!"]))
    a = attrs.get("SLxImageAttributes", attrs)
    return {
        "width": int(a["uiWidth"]),
        "height": int(a["uiHeight"]),
        "components": int(a["uiComp"]),
        "bits": int(a["uiBpcInMemory"]),
        "frames": int(a["uiSequenceCount"]),
        "width_bytes": int(a.get("uiWidthBytes", 0)),
        "compression": int(a.get("eCompression", 2)),
    }


def read_channel_names(data: bytes, cmap, n_components: int) -> list[str]:
    """Optical-configuration names of the planes, in component order (best effort)."""
    key = b"ImageMetadataSeqLV|0!"
    if key not in cmap:
        return []
    try:
        meta, _ = parse_lite_variant(chunk_payload(data, cmap[key]))
        planes = meta["SLxPictureMetadata"]["sPicturePlanes"]["sPlaneNew"]
        names = []
        for k in sorted(planes, key=lambda s: int("".join(c for c in s if c.isdigit()) or 0)):
            p = planes[k]
            names.append(str(p.get("sOpticalConfigName") or p.get("sDescription") or k))
        return names[:n_components]
    except Exception:
        return []


def read_channel_settings(data: bytes, cmap, n_components: int) -> list[dict]:
    """Per component: pixel size, objective, exposure, zoom and binning as the acquisition software stored them
    (``SLxPictureMetadata`` of ``ImageMetadataSeqLV|0!``: ``dCalibration`` and one ``sSampleSetting`` per plane) --
    the values R/nikon.py:220-300 takes from the ``nd2`` package's channel records.  Missing pieces stay None."""
    key = b"ImageMetadataSeqLV|0!"
    out = [dict(xy_step_um=None, magnification=None, numerical_aperture=None, exposure_time_s=None, zoom=None,
                binning=None) for _ in range(n_components)]
    if key not in cmap:
        return out
    try:
        pic = parse_lite_variant(chunk_payload(data, cmap[key]))[0]["SLxPictureMetadata"]
    except Exception:
        return out
    cal = pic.get("dCalibration")
    planes = pic.get("sPicturePlanes", {}) if isinstance(pic.get("sPicturePlanes"), dict) else {}
    settings = planes.get("sSampleSetting", {}) if isinstance(planes.get("sSampleSetting"), dict) else {}
    order = sorted(settings, key=lambda s: int("".join(c for c in s if c.isdigit()) or 0))
    for i, rec in enumerate(out):
        if isinstance(cal, float) and cal > 0:
            rec["xy_step_um"] = cal
        st = settings.get(order[i]) if i < len(order) else None
        if not isinstance(st, dict):
            continue
        obj = st.get("pObjectiveSetting") if isinstance(st.get("pObjectiveSetting"), dict) else {}
        if isinstance(obj.get("dObjectiveMag"), float) and obj["dObjectiveMag"] > 0:
            rec["magnification"] = obj["dObjectiveMag"]
        if isinstance(obj.get("dObjectiveNA"), float) and obj["dObjectiveNA"] > 0:
            rec["numerical_aperture"] = obj["dObjectiveNA"]
        if isinstance(st.get("dExposureTime"), float) and st["dExposureTime"] >= 0:
            rec["exposure_time_s"] = st["dExposureTime"] / 1000.0  # stored in milliseconds
        dev = st.get("pDeviceSetting") if isinstance(st.get("pDeviceSetting"), dict) else {}
        if isinstance(dev.get("m_dZoomPosition"), float):
            rec["zoom"] = dev["m_dZoomPosition"]
        cam = st.get("pCameraSetting") if isinstance(st.get("pCameraSetting"), dict) else {}
        fmt = cam.get("FormatQuality") or cam.get("FormatFast") or {}
        desc = fmt.get("fmtDesc", {}) if isinstance(fmt, dict) else {}
        if isinstance(desc.get("dBinningX"), float) and isinstance(desc.get("dBinningY"), float):
            rec["binning"] = f"{desc['dBinningX']:g}x{desc['dBinningY']:g}"
    return out


_LOOP_AXIS = {1: "T", 8: "T", 2: "P", 4: "Z"}  # SLxExperiment.eType: time, non-equidistant time, XY position, z stack


def read_experiment_loops(data: bytes, cmap, steps: dict | None = None) -> list[tuple[str, int]]:
    """[(axis letter, count)] of the acquisition loops, outermost first, from the ``SLxExperiment`` tree of the
    ``ImageMetadataLV!`` chunk (what ``nd2.ND2File.sizes`` is built from, R/nikon.py:197-210 reads 'T' / 'Z' / 'P'
    from it).  Loops of one step and loop kinds without a frame axis (spectral, eType 6) are skipped; [] when the
    chunk is absent or cannot be read.  ``steps`` (optional dict) receives {"Z": z step in um, "T": period in ms} for
    the loops that state a positive step."""
    key = b"ImageMetadataLV!"
    if key not in cmap:
        return []
    try:
        meta, _ = parse_lite_variant(chunk_payload(data, cmap[key]))
    except Exception:
        return []
    loops: list[tuple[str, int]] = []
    level = meta.get("SLxExperiment", meta)
    while isinstance(level, dict):
        pars = level.get("uLoopPars")
        if isinstance(pars, dict) and len(pars) == 1 and isinstance(next(iter(pars.values())), dict):
            pars = next(iter(pars.values()))  # some writers wrap the parameters in an unnamed level
        count = int(pars.get("uiCount", 1)) if isinstance(pars, dict) else 1
        axis = _LOOP_AXIS.get(int(level.get("eType", 0) or 0))
        if axis is not None and count > 1:
            loops.append((axis, count))
            if isinstance(pars, dict):
                step = pars.get("dZStep") if axis == "Z" else pars.get("dAvgPeriodDiff") or pars.get("dPeriod")
                if steps is not None and isinstance(step, float) and step > 0:
                    steps[axis] = step
        nxt = level.get("ppNextLevelEx") if int(level.get("uiNextLevelCount", 0) or 0) > 0 else None
        if isinstance(nxt, dict) and "eType" not in nxt:
            nxt = next((v for v in nxt.values() if isinstance(v, dict)), None)  # first (only) child level
        level = nxt
    return loops


def read_frames_interleaved(path: Path):
    """-> (frames (N, Y, X, C) uint16 little-endian, attributes, channel names)."""
    data = Path(path).read_bytes()
    cmap = read_chunk_map(data)
    at = read_attributes(data, cmap)
    if at["bits"] != 16:
        raise NotImplementedError(f"only 16-bit ND2 frames are supported (got {at['bits']} bits)")
    W, H, C, N = at["width"], at["height"], at["components"], at["frames"]
    # rows may be padded: uiWidthBytes is the row stride in bytes (the `nd2` reader honours it; R/nikon.py:41)
    row_bytes = W * C * 2
    stride = at["width_bytes"] or row_bytes
    if stride < row_bytes or stride % 2:
        raise ValueError(f"ND2 attributes are inconsistent: uiWidthBytes {stride} < {W} px x {C} components x 2 bytes")
    frame_bytes = stride * (H - 1) + row_bytes if H > 0 else 0
    frames = np.empty((N, H, W, C), dtype="<u2")
    for i in range(N):
        key = b"ImageDataSeq|%d!" % i
        if key not in cmap:
            raise ValueError(f"ND2 file has no chunk {key!r}")
        payload = chunk_payload(data, cmap[key])
        if len(payload) < 8 + frame_bytes:
            raise NotImplementedError("compressed ND2 frames are not supported")
        rows = np.ndarray((H, W, C), dtype="<u2", buffer=payload, offset=8, strides=(stride, C * 2, 2))
        frames[i] = rows
    at["loop_steps"] = {}
    at["loops"] = read_experiment_loops(data, cmap, at["loop_steps"])
    at["channel_settings"] = read_channel_settings(data, cmap, C)
    return frames, at, read_channel_names(data, cmap, C)


def load_nd2(nd2_path: Path, channels: list[Channel] | None = None, use_device: bool | None = None):
    """-> (intensities uint16, InstrumentMetadata) like R/nikon.py:25-43.

    Shapes follow ``nd2.ND2File.asarray()`` for the supported cases: (Y, X), (C, Y, X), (N, Y, X) or
    (N, C, Y, X) with size-1 axes dropped.  The loop axes of multi-frame files are named after the experiment's
    loops ('T' time, 'Z' z stack, 'P' positions, outermost first: ``read_experiment_loops``) when their counts
    multiply to the number of frames; otherwise the frames form one 'T' axis.
    """
    from .microscopy import InstrumentMetadata

    frames, at, names = read_frames_interleaved(Path(nd2_path))
    N, H, W, C = frames.shape
    if use_device is None:
        from . import _hip

        try:
            use_device = _hip.load_library().amt_device_count() > 0
        except Exception:
            use_device = False
    if C > 1 and use_device:
        from . import hipops
        from .device import get_context

        cyx = hipops.deinterleave(get_context().asarray(frames), C).numpy()
    else:
        cyx = np.ascontiguousarray(frames.transpose(0, 3, 1, 2))
    sizes: dict[str, int] = {}
    if N > 1:
        loops = at.get("loops") or []
        if loops and int(np.prod([c for _, c in loops])) == N and len({a for a, _ in loops}) == len(loops):
            sizes.update(loops)
        else:
            sizes["T"] = N
    if C > 1:
        sizes["C"] = C
    sizes["Y"], sizes["X"] = H, W
    out = cyx.reshape(tuple(sizes.values())).astype(np.uint16, copy=False)
    if channels is None:
        channels = []
        for i in range(C):
            ch = resolve_optical_config(names[i]) if i < len(names) else None
            channels.append(ch or Channel(names[i] if i < len(names) else f"CH{i}", "#FFFFFF"))
    flags = DimensionFlags(0)  # as R/nikon.py:197-210 derives them from the sizes
    if sizes.get("T", 1) > 1:
        flags |= DimensionFlags.TIMELAPSE
    if sizes.get("Z", 1) > 1:
        flags |= DimensionFlags.Z_STACK
    if sizes.get("P", 1) > 1:
        flags |= DimensionFlags.MONTAGE
    steps = at.get("loop_steps", {})
    records = []
    for i, ch in enumerate(channels):
        st = at["channel_settings"][i] if i < len(at.get("channel_settings", [])) else {}
        resolution = acquisition = optics = None
        timelapse = sizes.get("T", 1) > 1
        # a time loop without a stated period (non-equidistant loops, "no delay" acquisitions, frame counts that had
        # to be taken as T): the optional calibration record is left out rather than failing the whole read
        if st.get("xy_step_um") is not None and (not timelapse or steps.get("T") is not None):
            # z_size_px / z_step_um default to one plane of 1 um, as the nd2 package reports 2-D acquisitions
            resolution = NominalDimensions(
                x_size_px=W, y_size_px=H, xy_step_um=st["xy_step_um"], z_size_px=sizes.get("Z", 1),
                z_step_um=steps.get("Z", 1.0), t_size_px=sizes.get("T") if timelapse else None,
                t_step_ms=steps.get("T") if timelapse else None)
        if any(st.get(k) is not None for k in ("exposure_time_s", "zoom", "binning")):
            acquisition = AcquisitionSettings(exposure_time_s=st.get("exposure_time_s"), zoom=st.get("zoom"),
                                              binning=st.get("binning"))
        if st.get("magnification") is not None and st.get("numerical_aperture") is not None:
            optics = MicroscopeConfig(magnification=st["magnification"], numerical_aperture=st["numerical_aperture"])
        records.append(ChannelMetadata(ch, dimensions=flags, resolution=resolution, acquisition=acquisition, optics=optics))
    meta = InstrumentMetadata(sizes, records)
    return out, meta
