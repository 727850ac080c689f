// Per-cell outlines: marching squares at level 0.5 on the padded bounding-box crop of every label, longest
// contour kept -- the skimage extractor behind SegmentationMask.cell_outlines (R/masks.py:82-115,
// SK/measure/_find_contours.py, _find_contours_cy.pyx; CPU restatement: oracle/contours.py).
//
// scikit-image builds segments square by square and stitches them with dictionaries.  What that assembly
// produces can be stated without dictionaries (oracle/contours.py header), and that is what runs here:
//   * a contour is the chain of its segments in orientation order; following it is local: the to-edge of a
//     segment is the from-edge of exactly one segment of the square across that edge;
//   * contours are listed by their FIRST segment in raster order -- so a raster scan over the squares that
//     skips segments already walked discovers them in list order, and `max(contours, key=len)` is "the first
//     contour with the strictly largest segment count";
//   * a closed contour starts at the to-point of its LAST segment in raster order, an open one (clipped by the
//     crop edge) at its first point.
// One thread per label (the walks are sequential; a wave works on 64 labels at once).  Two kernels with a host
// step in between (the outline lengths size the output): contours_find marks walked segments in a scratch
// byte per square and reports {points, closed, start square, start segment}; contours_emit re-walks the chosen
// contour and writes (row, col) float64 pairs in image coordinates.
#include "amt_internal.h"

namespace {

// edges of a square: T = 0, L = 1, R = 2, B = 3 (opposite = 3 - e)
// per case: nseg | from0 << 2 | to0 << 4 | from1 << 6 | to1 << 8   (case = ul + 2 ur + 4 ll + 8 lr)
__device__ __constant__ unsigned short CASE_ENC[16] = {
    0,
    1 | 0 << 2 | 1 << 4,                      // 1: T -> L
    1 | 2 << 2 | 0 << 4,                      // 2: R -> T
    1 | 2 << 2 | 1 << 4,                      // 3: R -> L
    1 | 1 << 2 | 3 << 4,                      // 4: L -> B
    1 | 0 << 2 | 3 << 4,                      // 5: T -> B
    2 | 2 << 2 | 0 << 4 | 1 << 6 | 3 << 8,    // 6: R -> T, L -> B
    1 | 2 << 2 | 3 << 4,                      // 7: R -> B
    1 | 3 << 2 | 2 << 4,                      // 8: B -> R
    2 | 0 << 2 | 1 << 4 | 3 << 6 | 2 << 8,    // 9: T -> L, B -> R
    1 | 3 << 2 | 0 << 4,                      // 10: B -> T
    1 | 3 << 2 | 1 << 4,                      // 11: B -> L
    1 | 1 << 2 | 2 << 4,                      // 12: L -> R
    1 | 0 << 2 | 2 << 4,                      // 13: T -> R
    1 | 1 << 2 | 0 << 4,                      // 14: L -> T
    0,
};

struct crop {
    const int* L;  // label plane
    int W;         // image width
    int label;
    int r_lo, c_lo;  // crop origin in the image
    int sh, sw;      // squares: (h - 1) x (w - 1)
};

__device__ __forceinline__ int nseg(unsigned enc) { return enc & 3; }
__device__ __forceinline__ int seg_from(unsigned enc, int k) { return (enc >> (2 + 4 * k)) & 3; }
__device__ __forceinline__ int seg_to(unsigned enc, int k) { return (enc >> (4 + 4 * k)) & 3; }

__device__ __forceinline__ unsigned sq_enc(const crop& c, int r0, int c0) {
    const int* p = c.L + (size_t)(c.r_lo + r0) * c.W + (c.c_lo + c0);
    const int cs = (p[0] == c.label ? 1 : 0) | (p[1] == c.label ? 2 : 0) | (p[c.W] == c.label ? 4 : 0) |
                   (p[c.W + 1] == c.label ? 8 : 0);
    return CASE_ENC[cs];
}

// the square across edge e of (r0, c0); false when it is outside the crop
__device__ __forceinline__ bool across(const crop& c, int e, int& r0, int& c0) {
    r0 += (e == 0) ? -1 : (e == 3) ? 1 : 0;
    c0 += (e == 1) ? -1 : (e == 2) ? 1 : 0;
    return r0 >= 0 && r0 < c.sh && c0 >= 0 && c0 < c.sw;
}

// (row, col) of the midpoint of edge e of square (r0, c0), image coordinates
__device__ __forceinline__ void edge_point(const crop& c, int r0, int c0, int e, double& y, double& x) {
    y = (double)(c.r_lo + r0) + ((e == 1 || e == 2) ? 0.5 : (e == 3) ? 1.0 : 0.0);
    x = (double)(c.c_lo + c0) + ((e == 0 || e == 3) ? 0.5 : (e == 2) ? 1.0 : 0.0);
}

__device__ __forceinline__ bool load_crop(crop& c, const int* labels, int H, int W, const int* box) {
    c.L = labels;
    c.W = W;
    c.label = box[0];
    c.r_lo = box[1];
    c.c_lo = box[2];
    const int h = box[3] - box[1], w = box[4] - box[2];
    c.sh = h - 1;
    c.sw = w - 1;
    (void)H;
    return h >= 2 && w >= 2;
}

}  // namespace

// boxes: nlab x 5 ints {label, r_lo, c_lo, r_hi, c_hi} (padded, clamped crop, half-open)
// voff : nlab + 1 offsets into `visited` (one byte per square of the crop, zero on entry)
// info : nlab x 4 ints {points, closed, start square (r0 * sw + c0), start segment}
__global__ void __launch_bounds__(64) contours_find_kernel(const int* __restrict__ labels, int H, int W, int nlab,
                                                           const int* __restrict__ boxes,
                                                           const long long* __restrict__ voff,
                                                           unsigned char* __restrict__ visited, int* __restrict__ info) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlab) return;
    crop c;
    int* out = info + (size_t)li * 4;
    if (!load_crop(c, labels, H, W, boxes + (size_t)li * 5)) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    unsigned char* vis = visited + voff[li];
    // a crop has at most two segments per square: every walk below is cut off there, so the kernel terminates
    // even if the label image changes under it
    const long long max_steps = 2ll * c.sh * c.sw + 2;
    int best_n = 0, best_closed = 0, best_sq = 0, best_k = 0;
    for (int r0 = 0; r0 < c.sh; ++r0) {
        for (int c0 = 0; c0 < c.sw; ++c0) {
            const unsigned enc0 = sq_enc(c, r0, c0);
            for (int k = 0; k < nseg(enc0); ++k) {
                const int sq0 = r0 * c.sw + c0;
                if ((vis[sq0] >> k) & 1) continue;
                // a contour nobody has walked yet, met at its first segment in raster order: walk it forward
                int n = 0, closed = 0;
                int cr = r0, cc = c0, ck = k;
                unsigned enc = enc0;
                int last_key = -1, last_sq = sq0, last_k = k;
                while (n < max_steps) {
                    const int sq = cr * c.sw + cc;
                    vis[sq] |= (unsigned char)(1 << ck);
                    ++n;
                    const int key = sq * 2 + ck;
                    if (key > last_key) {
                        last_key = key;
                        last_sq = sq;
                        last_k = ck;
                    }
                    const int e = seg_to(enc, ck);
                    int nr = cr, nc = cc;
                    if (!across(c, e, nr, nc)) break;  // clipped by the crop edge: open contour
                    const unsigned nenc = sq_enc(c, nr, nc);
                    const int nk = seg_from(nenc, 0) == 3 - e ? 0 : 1;
                    if (nr == r0 && nc == c0 && nk == k) {
                        closed = 1;
                        break;
                    }
                    cr = nr;
                    cc = nc;
                    ck = nk;
                    enc = nenc;
                }
                int st_sq = last_sq, st_k = last_k;
                if (!closed) {  // walk backwards from the first segment to the beginning of the chain
                    cr = r0;
                    cc = c0;
                    ck = k;
                    enc = enc0;
                    while (n < max_steps) {
                        const int e = seg_from(enc, ck);
                        int nr = cr, nc = cc;
                        if (!across(c, e, nr, nc)) break;
                        const unsigned nenc = sq_enc(c, nr, nc);
                        const int nk = seg_to(nenc, 0) == 3 - e ? 0 : 1;
                        vis[nr * c.sw + nc] |= (unsigned char)(1 << nk);
                        ++n;
                        cr = nr;
                        cc = nc;
                        ck = nk;
                        enc = nenc;
                    }
                    st_sq = cr * c.sw + cc;
                    st_k = ck;
                }
                if (n > best_n) {  // strictly longer: the first of equally long contours wins, as max() does
                    best_n = n;
                    best_closed = closed;
                    best_sq = st_sq;
                    best_k = st_k;
                }
            }
        }
    }
    out[0] = best_n ? best_n + 1 : 0;
    out[1] = best_closed;
    out[2] = best_sq;
    out[3] = best_k;
}

// poff: nlab + 1 offsets (in points) into points_out; points_out: (row, col) float64 pairs
__global__ void __launch_bounds__(64) contours_emit_kernel(const int* __restrict__ labels, int H, int W, int nlab,
                                                           const int* __restrict__ boxes, const int* __restrict__ info,
                                                           const long long* __restrict__ poff,
                                                           double* __restrict__ points_out) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlab) return;
    const int* inf = info + (size_t)li * 4;
    const int npts = inf[0];
    if (npts == 0) return;
    crop c;
    if (!load_crop(c, labels, H, W, boxes + (size_t)li * 5)) return;
    double* P = points_out + 2 * poff[li];
    int cr = inf[2] / c.sw, cc = inf[2] - cr * c.sw, ck = inf[3];
    unsigned enc = sq_enc(c, cr, cc);
    int written = 0;
    double y, x;
    if (!inf[1]) {  // open: the chain's first point, then every to-point
        edge_point(c, cr, cc, seg_from(enc, ck), y, x);
        P[0] = y;
        P[1] = x;
        written = 1;
    }
    // closed: to(last), to(e1), ..., to(last) again; open: to(e0), to(e1), ...
    while (written < npts) {
        const int e = seg_to(enc, ck);
        edge_point(c, cr, cc, e, y, x);
        P[2 * written] = y;
        P[2 * written + 1] = x;
        ++written;
        if (written == npts) break;
        if (!across(c, e, cr, cc)) break;  // cannot happen for a consistent info record
        enc = sq_enc(c, cr, cc);
        ck = seg_from(enc, 0) == 3 - e ? 0 : 1;
    }
}

extern "C" int amt_contours_find(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                                 const int64_t* voff_dev, uint8_t* visited_dev, size_t visited_bytes,
                                 int32_t* info_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && H > 0 && W > 0 && nlab >= 0, "contours_find: bad arguments");
    if (nlab == 0) return AMT_OK;
    AMT_REQUIRE(boxes_dev && voff_dev && info_dev && (visited_dev || visited_bytes == 0), "contours_find: null pointer");
    if (visited_bytes) AMT_HIP_CHECK(hipMemsetAsync(visited_dev, 0, visited_bytes, ctx->stream));
    hipLaunchKernelGGL(contours_find_kernel, dim3((nlab + 63) / 64), dim3(64), 0, ctx->stream, labels, H, W, nlab,
                       boxes_dev, (const long long*)voff_dev, visited_dev, info_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_contours_emit(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                                 const int32_t* info_dev, const int64_t* poff_dev, double* points_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && H > 0 && W > 0 && nlab >= 0, "contours_emit: bad arguments");
    if (nlab == 0) return AMT_OK;
    AMT_REQUIRE(boxes_dev && info_dev && poff_dev && points_dev, "contours_emit: null pointer");
    hipLaunchKernelGGL(contours_emit_kernel, dim3((nlab + 63) / 64), dim3(64), 0, ctx->stream, labels, H, W, nlab,
                       boxes_dev, info_dev, (const long long*)poff_dev, points_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Pixel-border outlines: the "cellpose" extractor of the reference (R/masks.py:68-79).
// cellpose.utils.outlines_list (4.0.8) runs, per label n, cv2.findContours(masks == n, RETR_EXTERNAL,
// CHAIN_APPROX_NONE) (opencv-python-headless 4.11.0.86 in R's uv.lock) and keeps the contour with the most
// points.  findContours is Suzuki-Abe border following; what is restated here is its published raster scan:
//   * a pixel with a zero left neighbour that is still unmarked starts an OUTER border, unless the last border
//     pixel passed on this row carries a positive mark (we are inside an already traced border: nested
//     components are not external);
//   * the border is followed from the start pixel i0: the first non-zero neighbour clockwise from "left" is i1;
//     from then on the next pixel is the first non-zero neighbour counter-clockwise after the direction we came
//     from; every visited pixel is one output point (CHAIN_APPROX_NONE); the walk ends when the current pixel is
//     i1 and the next one is i0 again;
//   * a visited pixel is marked -126 when its right neighbour was examined and found zero, else 2.
// Directions: 0 = +x, then counter-clockwise on the screen (1 = +x -y, 2 = -y, ... 7 = +x +y).
// One thread per label over its tight bounding box; marks live in one H x W byte plane (a pixel belongs to one
// label).  Neither cv2 nor cellpose exists offline: parity unpinned (oracle/contours.py says the same).
// ---------------------------------------------------------------------------------------------------------
struct bbox_t {
    const int* lab;
    int W, label, r0, c0, r1, c1;  // half-open, tight
};

__device__ __forceinline__ bool bd_nz(const bbox_t& b, int y, int x) {
    return y >= b.r0 && y < b.r1 && x >= b.c0 && x < b.c1 && b.lab[(size_t)y * b.W + x] == b.label;
}

__device__ __forceinline__ void bd_step(int s, int& dy, int& dx) {
    // s & 7: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards)
    const int k = s & 7;
    dx = (k == 0 || k == 1 || k == 7) ? 1 : ((k >= 3 && k <= 5) ? -1 : 0);
    dy = (k >= 1 && k <= 3) ? -1 : ((k >= 5) ? 1 : 0);
}

// Follows the outer border that starts at (y0, x0).  marks may be null (emit pass).  pts (y, x) int32 pairs may
// be null (find pass).  Returns the number of points (0 if the step bound is hit: cannot happen for a
// consistent image, it only guarantees termination).
__device__ int bd_trace(const bbox_t& b, int y0, int x0, signed char* marks, int* pts, int max_steps) {
    int s = 4, dy, dx;
    const int s_stop = 4;
    bool found = false;
    do {
        s = (s - 1) & 7;
        bd_step(s, dy, dx);
        if (bd_nz(b, y0 + dy, x0 + dx)) {
            found = true;
            break;
        }
    } while (s != s_stop);
    if (!found) {  // isolated pixel
        if (marks) marks[(size_t)y0 * b.W + x0] = (signed char)-126;
        if (pts) {
            pts[0] = y0;
            pts[1] = x0;
        }
        return 1;
    }
    const int y1 = y0 + dy, x1 = x0 + dx;
    int y3 = y0, x3 = x0, n = 0;
    for (int step = 0; step < max_steps; ++step) {
        const int s_end = s;
        int y4 = y3, x4 = x3;
        while (s < 15) {
            ++s;
            bd_step(s, dy, dx);
            y4 = y3 + dy;
            x4 = x3 + dx;
            if (bd_nz(b, y4, x4)) break;
        }
        s &= 7;
        if (marks) {
            signed char* m = marks + (size_t)y3 * b.W + x3;
            if ((unsigned)(s - 1) < (unsigned)s_end) *m = (signed char)-126;
            else if (*m == 0) *m = 2;
        }
        if (pts) {
            pts[2 * n] = y3;
            pts[2 * n + 1] = x3;
        }
        ++n;
        if (y4 == y0 && x4 == x0 && y3 == y1 && x3 == x1) return n;
        y3 = y4;
        x3 = x4;
        s = (s + 4) & 7;
    }
    return 0;
}

// boxes: nlab x 5 ints {label, r0, c0, r1, c1} tight half-open; info: nlab x 3 ints {points, start y, start x}
__global__ void __launch_bounds__(64) borders_find_kernel(const int* __restrict__ labels, int H, int W, int nlab,
                                                          const int* __restrict__ boxes, signed char* marks,
                                                          int* __restrict__ info) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlab) return;
    const int* bx = boxes + (size_t)li * 5;
    bbox_t b{labels, W, bx[0], bx[1], bx[2], bx[3], bx[4]};
    int best_n = 0, best_y = 0, best_x = 0;
    const bool ok = b.r0 >= 0 && b.c0 >= 0 && b.r1 <= H && b.c1 <= W && b.r0 < b.r1 && b.c0 < b.c1;
    if (ok) {
        const int max_steps = 8 * (b.r1 - b.r0) * (b.c1 - b.c0) + 16;
        for (int y = b.r0; y < b.r1; ++y) {
            int prev = 0, lnbd = 0;  // value of the previous pixel / of the last border pixel passed on this row
            for (int x = b.c0; x < b.c1; ++x) {
                int p = 0;
                if (labels[(size_t)y * W + x] == b.label) {
                    const int m = marks[(size_t)y * W + x];
                    p = m ? m : 1;
                }
                if (prev == 0 && p == 1 && lnbd <= 0) {
                    const int n = bd_trace(b, y, x, marks, nullptr, max_steps);
                    // the contour list comes back newest first and argmax keeps the first maximum:
                    // among equals the LAST border found in raster order wins
                    if (n >= best_n && n > 0) {
                        best_n = n;
                        best_y = y;
                        best_x = x;
                    }
                    p = marks[(size_t)y * W + x];
                }
                if (p != 0 && p != 1) lnbd = p;
                prev = p;
            }
        }
    }
    info[(size_t)li * 3 + 0] = best_n;
    info[(size_t)li * 3 + 1] = best_y;
    info[(size_t)li * 3 + 2] = best_x;
}

// poff: nlab + 1 offsets (in points) into points_out = (y, x) int32 pairs; labels with fewer than five border
// points have an empty slot (cellpose keeps only len(pix) > 4; the host passes poff accordingly)
__global__ void __launch_bounds__(64) borders_emit_kernel(const int* __restrict__ labels, int H, int W, int nlab,
                                                          const int* __restrict__ boxes, const int* __restrict__ info,
                                                          const long long* __restrict__ poff,
                                                          int* __restrict__ points_out) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nlab) return;
    const long long lo = poff[li], hi = poff[li + 1];
    const int n = info[(size_t)li * 3];
    if (hi - lo != n || n == 0) return;
    const int* bx = boxes + (size_t)li * 5;
    bbox_t b{labels, W, bx[0], bx[1], bx[2], bx[3], bx[4]};
    if (!(b.r0 >= 0 && b.c0 >= 0 && b.r1 <= H && b.c1 <= W)) return;
    bd_trace(b, info[(size_t)li * 3 + 1], info[(size_t)li * 3 + 2], nullptr, points_out + 2 * lo, n);
}

extern "C" int amt_borders_find(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                                int8_t* marks_dev, int32_t* info_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && H > 0 && W > 0 && nlab >= 0, "borders_find: bad arguments");
    if (nlab == 0) return AMT_OK;
    AMT_REQUIRE(boxes_dev && marks_dev && info_dev, "borders_find: null pointer");
    AMT_HIP_CHECK(hipMemsetAsync(marks_dev, 0, (size_t)H * W, ctx->stream));
    hipLaunchKernelGGL(borders_find_kernel, dim3((nlab + 63) / 64), dim3(64), 0, ctx->stream, labels, H, W, nlab,
                       boxes_dev, (signed char*)marks_dev, info_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_borders_emit(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                                const int32_t* info_dev, const int64_t* poff_dev, int32_t* points_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && H > 0 && W > 0 && nlab >= 0, "borders_emit: bad arguments");
    if (nlab == 0) return AMT_OK;
    AMT_REQUIRE(boxes_dev && info_dev && poff_dev && points_dev, "borders_emit: null pointer");
    hipLaunchKernelGGL(borders_emit_kernel, dim3((nlab + 63) / 64), dim3(64), 0, ctx->stream, labels, H, W, nlab,
                       boxes_dev, info_dev, (const long long*)poff_dev, points_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
