"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the skimage outline extractor of the reference.

Reference: ``_extract_outlines_skimage`` (R/masks.py:82-115): for every region of ``regionprops(label_image)``
the bounding box is padded by one pixel (clamped to the image), ``find_contours(crop == label, level=0.5)`` is
run on the crop, the LONGEST contour is kept (``max(contours, key=len)``: the first one among equals) and shifted
back to image coordinates; regions without a contour give an empty (0, 2) array.

``skimage.measure.find_contours`` (SK/measure/_find_contours.py, _find_contours_cy.pyx) = marching squares:
segments are produced square by square in raster order with the 16-case table below (pinned against
scikit-image 0.18.3 on all 16 two-by-two images, tools/make_golden.py:gold_outlines), then assembled by
``_assemble_contours``.  What the assembly does, and what the device kernels reproduce without dictionaries:
  * pieces are always joined head -> tail in orientation order, so a contour is its chain of segments in
    orientation order;
  * a contour's position in the output list is the creation index of its oldest piece = its first segment in
    raster order;
  * a CLOSED contour starts (and ends) at the to-point of its LAST segment in raster order (the segment that
    closes it), an OPEN one (clipped by the crop edge) starts at its first point.
Pinned by tests/golden/outlines_96.npz (real scikit-image 0.18.3; the reference pins 0.25.2 whose algorithm is
the same).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
from __future__ import annotations

from collections import deque

import numpy as np

# edges of a unit square with upper-left corner (0, 0): point = (row, col)
TOP, LEFT, RIGHT, BOTTOM = (0.0, 0.5), (0.5, 0.0), (0.5, 1.0), (1.0, 0.5)
# square_case = 1*ul + 2*ur + 4*ll + 8*lr  ->  list of (from, to) segments, in skimage's emission order
# (fully_connected='low': cases 6 and 9 separate the two high pixels)
CASES = {
    0: [], 15: [],
    1: [(TOP, LEFT)], 2: [(RIGHT, TOP)], 3: [(RIGHT, LEFT)], 4: [(LEFT, BOTTOM)], 5: [(TOP, BOTTOM)],
    6: [(RIGHT, TOP), (LEFT, BOTTOM)], 7: [(RIGHT, BOTTOM)], 8: [(BOTTOM, RIGHT)],
    9: [(TOP, LEFT), (BOTTOM, RIGHT)], 10: [(BOTTOM, TOP)], 11: [(BOTTOM, LEFT)], 12: [(LEFT, RIGHT)],
    13: [(TOP, RIGHT)], 14: [(LEFT, TOP)],
}


def contour_segments(binary: np.ndarray):
    """Segments ((r, c), (r, c)) of the 0.5 level set of a 0/1 image, in skimage's order."""
    b = np.asarray(binary).astype(bool)
    h, w = b.shape
    out = []
    for r0 in range(h - 1):
        for c0 in range(w - 1):
            case = int(b[r0, c0]) | int(b[r0, c0 + 1]) << 1 | int(b[r0 + 1, c0]) << 2 | int(b[r0 + 1, c0 + 1]) << 3
            for f, t in CASES[case]:
                out.append(((r0 + f[0], c0 + f[1]), (r0 + t[0], c0 + t[1])))
    return out


def assemble_contours(segments):
    """SK/measure/_find_contours.py:_assemble_contours, restated."""
    current_index = 0
    contours, starts, ends = {}, {}, {}
    for from_point, to_point in segments:
        if from_point == to_point:
            continue
        tail, tail_num = starts.pop(to_point, (None, None))
        head, head_num = ends.pop(from_point, (None, None))
        if tail is not None and head is not None:
            if tail is head:
                head.append(to_point)
            elif tail_num > head_num:
                head.extend(tail)
                ends.pop(tail[-1])
                contours.pop(tail_num, None)
                starts[head[0]] = (head, head_num)
                ends[head[-1]] = (head, head_num)
            else:
                tail.extendleft(reversed(head))
                starts.pop(head[0])
                contours.pop(head_num, None)
                starts[tail[0]] = (tail, tail_num)
                ends[tail[-1]] = (tail, tail_num)
        elif tail is None and head is None:
            new_contour = deque((from_point, to_point))
            contours[current_index] = new_contour
            starts[from_point] = (new_contour, current_index)
            ends[to_point] = (new_contour, current_index)
            current_index += 1
        elif head is None:
            tail.appendleft(from_point)
            starts[from_point] = (tail, tail_num)
        else:
            head.append(to_point)
            ends[to_point] = (head, head_num)
    return [np.array(c, dtype=np.float64) for _, c in sorted(contours.items())]


def find_contours(binary: np.ndarray):
    """``skimage.measure.find_contours(binary.astype(uint8), level=0.5)`` for 0/1 images."""
    if binary.ndim != 2 or binary.shape[0] < 2 or binary.shape[1] < 2:
        raise ValueError("Input array must be at least 2x2.")
    return assemble_contours(contour_segments(binary))


def label_bboxes(label_image: np.ndarray):
    """{label: (minr, minc, maxr, maxc)} half-open, ascending labels (``regionprops`` order)."""
    lab = np.asarray(label_image)
    out = {}
    for l in np.unique(lab):
        if l <= 0:
            continue
        rr, cc = np.nonzero(lab == l)
        out[int(l)] = (int(rr.min()), int(cc.min()), int(rr.max()) + 1, int(cc.max()) + 1)
    return out


def extract_outlines_skimage(label_image: np.ndarray):
    """R/masks.py:82-115."""
    lab = np.asarray(label_image)
    h, w = lab.shape
    outlines = []
    for l, (minr, minc, maxr, maxc) in label_bboxes(lab).items():
        r0, c0 = max(minr - 1, 0), max(minc - 1, 0)
        r1, c1 = min(maxr + 1, h), min(maxc + 1, w)
        crop = lab[r0:r1, c0:c1] == l
        contours = find_contours(crop)
        if contours:
            main = max(contours, key=len)
            outlines.append(main + np.array([r0, c0], dtype=np.float64))
        else:
            outlines.append(np.array([]).reshape(0, 2))
    return outlines


# ---------------------------------------------------------------------------------------------------------------
# The "cellpose" extractor (R/masks.py:68-79) -- PARITY UNPINNED.
# ``cellpose.utils.outlines_list(masks, multiprocessing=False)`` (cellpose 4.0.8, R's uv.lock) is, per label n of
# ``np.unique(masks)[1:]``: ``cv2.findContours((masks == n).astype(uint8), RETR_EXTERNAL, CHAIN_APPROX_NONE)``,
# the contour with the most points (``np.argmax``: first maximum of the returned list), ``.astype(int).squeeze()``,
# kept if ``len(pix) > 4`` else ``np.zeros((0, 2))``; the reference then swaps (x, y) -> (y, x).
# Neither cellpose nor OpenCV (opencv-python-headless 4.11.0.86) can be imported here and the reference's tests hold
# no vector for this extractor, so the functions below restate the published Suzuki-Abe border following the way
# OpenCV's scanner runs it (raster scan, 8-neighbourhood, start pixel = first pixel of the border in raster order,
# outer borders run counter-clockwise on the screen: a filled rectangle gives top-left, down the left side, along
# the bottom, up the right side, back along the top) and nothing pins them to real OpenCV output.
# ---------------------------------------------------------------------------------------------------------------
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _follow_border(img: np.ndarray, y0: int, x0: int):
    """Outer border from (y0, x0) on a zero-padded int8 image (1 = unvisited, 2 / -126 = visited); marks it."""
    s = 4
    while True:
        s = (s - 1) & 7
        if img[y0 + _DY[s], x0 + _DX[s]] != 0 or s == 4:
            break
    if img[y0 + _DY[s], x0 + _DX[s]] == 0:
        img[y0, x0] = -126
        return [(x0, y0)]
    y1, x1 = y0 + _DY[s], x0 + _DX[s]
    y3, x3 = y0, x0
    pts = []
    while True:
        s_end = s
        while True:
            s += 1
            y4, x4 = y3 + _DY[s & 7], x3 + _DX[s & 7]
            if img[y4, x4] != 0:
                break
        s &= 7
        if s != 0 and s - 1 < s_end:
            img[y3, x3] = -126
        elif img[y3, x3] == 1:
            img[y3, x3] = 2
        pts.append((x3, y3))
        if (y4, x4) == (y0, x0) and (y3, x3) == (y1, x1):
            return pts
        y3, x3 = y4, x4
        s = (s + 4) & 7


def find_external_borders(binary: np.ndarray):
    """``cv2.findContours(binary, RETR_EXTERNAL, CHAIN_APPROX_NONE)[-2]``: list of (N, 1, 2) int32 (x, y) arrays,
    newest border first (OpenCV links every new contour at the head of its sibling list)."""
    b = np.asarray(binary) != 0
    h, w = b.shape
    img = np.zeros((h + 2, w + 2), dtype=np.int8)
    img[1:-1, 1:-1] = b
    found = []
    for y in range(1, h + 1):
        prev, lnbd = 0, 0
        for x in range(1, w + 1):
            p = int(img[y, x])
            if prev == 0 and p == 1 and lnbd <= 0:
                pts = _follow_border(img, y, x)
                found.append(np.array(pts, dtype=np.int32).reshape(-1, 1, 2) - 1)
                p = int(img[y, x])
            if p not in (0, 1):
                lnbd = p
            prev = p
    return found[::-1]


def extract_outlines_cellpose(label_image: np.ndarray):
    """R/masks.py:68-79 with ``outlines_list`` inlined (see the note above: parity unpinned)."""
    lab = np.asarray(label_image)
    outlines = []
    for n in np.unique(lab)[1:]:
        mn = lab == n
        if mn.sum() > 0:
            contours = find_external_borders(mn)
            cmax = int(np.argmax([c.shape[0] for c in contours]))
            pix = contours[cmax].astype(int).squeeze()
            outlines.append(pix if len(pix) > 4 else np.zeros((0, 2)))
    return [o[:, [1, 0]] if len(o) > 0 else o for o in outlines]
