// Marker-based watershed (priority flood) restricted to a mask.
//
// Contract: skimage.segmentation.watershed(image, markers, connectivity=1, mask=mask) -- SURVEY.md A.1:
// pop the pending pixel with the smallest (value, insertion age); every unlabelled masked 4-neighbour,
// visited in the order (north, west, east, south), takes the popped pixel's label AT PUSH TIME and is
// queued with the next age.  All marker pixels start with age 0.
//
// What is order-defining and what is not:
//   * (value, age) is unique for every queued pixel except among the age-0 marker pixels.  scikit-image
//     breaks those ties by the internal moves of its ONE binary heap over the whole image, so the order of two
//     tied markers depends on every other element that passes through the heap -- other mask components
//     included (measured with the oracle: flooding each component with its own heap of the same mechanics
//     changes 20 of 6,458 labelled pixels of tests/golden/c2c3_256.npz).  Tied planes therefore have no parallel
//     decomposition; the parallel flood below breaks such ties in raster order and REPORTS them (ties[plane]),
//     and under AMT_WS_TIES_EXACT a tied plane is re-flooded by ws_global_kernel, a sequential emulation of
//     scikit-image's heap (bit-identical, one lane).  Ties between markers of DIFFERENT components, or inside a
//     component whose markers all carry one label, cannot change the result and are not reported.
//     The config-3 recipe gives marker pixels distinct, lowest values (oracle/skops.py:seeded_flood_image; here
//     `seeds_first`), for which ties cannot occur.
//   * the flood never crosses between 4-connected components of the mask, and the relative order of two
//     pixels of one component does not depend on the other components.  Components are therefore
//     independent work items:
//       - a component whose marker pixels all carry ONE label ends up entirely with that label: it is
//         filled by a pixel-parallel kernel, no flood at all;
//       - otherwise the component is flooded sequentially by one lane.  If its bounding box fits an LDS
//         tile (three size classes), a wave stages label / d2 / FIFO links in LDS, lane 0 floods at LDS
//         latency, and the wave writes the labels back; larger components use queues in HBM.
//
// amt_watershed_edt: relief = -sqrt(d2) with d2 an exact non-negative integer, so the priority queue
// is a bucket queue indexed by d2 (largest d2 = lowest relief first) with a FIFO per bucket: insertion
// age order inside a bucket is push order.  A pixel is pushed at most once, so each FIFO is a linked
// list threaded through one link per pixel.
// amt_watershed_f64: arbitrary float64 relief; per-component binary heap keyed (value, age, raster).
#include "amt_internal.h"
#include <hip/hip_ext.h>

// component classes
// LB = an L component too large for a persistent flood workgroup that owns only half a CU's LDS (ws_flood_persist_kernel)
enum { CLS_NONE = 0, CLS_UNIFORM = 1, CLS_XS = 2, CLS_S = 3, CLS_M = 4, CLS_M2 = 5, CLS_L = 6, CLS_X = 7, CLS_LB = 8, CLS_G = 9 };
constexpr int WS_NLISTS = 7;  // worklists XS, S, M, M2, L, X, LB
constexpr int WS_CTR = 24;    // ints per plane: work counters [0..7), HBM flood [7], has_g [8], list sizes [9..16), ncomp [16]
// LDS tile classes: max pixels of the bounding box, max d2 (bucket count - 1).  XS / S / M / M2 / L are flooded by
// ws_flood_batch_kernel (6 bytes of LDS per pixel + 8 per bucket), X -- the few boxes between L and the 15-bit
// index limit -- by the one-pop-at-a-time ws_flood_lds_kernel (4 bytes per pixel).
// Round 3: a class XS (1,280 px / 256 buckets, 9.2 KB: 17 workgroups per CU) for the 46 % of the synthetic plate's flooded
// components that fit it was measured and is switched OFF (AMT_WS_XS_PX 0): occupancy does bound the small classes
// (padding S: 10 workgroups per CU 468 us, 5 per CU 711 us), but every class launch has a floor of ~250 us -- its
// longest component chain -- so splitting S (468 us) gave XS 255 + S 310 us, the stage stayed at 2.51 ms per 48 FOVs and
// the 48-FOV plate's went from 1.07 to 1.16 ms.
#ifndef AMT_WS_XS_PX
#define AMT_WS_XS_PX 0
#define AMT_WS_XS_NB 256
#endif
#ifndef AMT_WS_S_PX
#define AMT_WS_S_PX 2048
#define AMT_WS_S_NB 512
#endif
#ifndef AMT_WS_M_PX
// 4,096 px (27 KB with the bucket words: five workgroups per CU).  Measured against 8,192 px / 1,024 buckets (two per
// CU): watershed stage 1.86 -> 1.73 ms per 32 FOVs, 0.95 -> 0.79 ms per 12 FOVs; boxes above 4,096 px join class M2 / L
#define AMT_WS_M_PX 4096
#define AMT_WS_M_NB 512
#endif
constexpr int XS_PX = AMT_WS_XS_PX, XS_NB = AMT_WS_XS_NB;
constexpr int S_PX = AMT_WS_S_PX, S_NB = AMT_WS_S_NB;
constexpr int M_PX = AMT_WS_M_PX, M_NB = AMT_WS_M_NB;
#ifndef AMT_WS_M2_PX
#define AMT_WS_M2_PX 8192
#endif
constexpr int M2_PX = AMT_WS_M2_PX, M2_NB = 1024;  // 54 KB: two workgroups per CU (0 = class unused)
constexpr int L_PX = 24576, L_NB = 2048;
constexpr int X_PX = 32512, X_NB = 2048;  // pixel indices must fit the 15-bit link field

struct comp_row {
    int cmax;            // max d2 (bucket mode) or pixel count (heap mode)
    int mcnt;            // marker pixels
    int x0, y0, x1, y1;  // bounding box (inclusive)
    int labmin, labmax;  // marker label range
    int cls;
    int root;            // flat index of the component's first pixel (its union-find root)
    int npix;            // pixels of the component (run-table statistics only; 0 = not counted)
};

// out = label of the component for single-label components, markers * mask elsewhere.
// F[root] = that label (0 for components that need a flood), written by ws_classify_kernel.
// only_planes (nullable): planes whose flag is 0 are skipped (the fused path seeds only planes that hold a component for
// the HBM flood, which works in `out` itself).
__global__ void __launch_bounds__(256) ws_seed_kernel(const int* __restrict__ markers, const int* __restrict__ L,
                                                      const int* __restrict__ F, int* __restrict__ out, size_t n,
                                                      const int* __restrict__ only_planes, amt_runtabs rt, int W) {
    if (only_planes && !only_planes[blockIdx.y]) return;
    const size_t base = (size_t)blockIdx.y * n;
    if (rt.tbits) {
        // no parent plane: a look-up per pixel (rare: only planes that hold a component too large for an LDS tile)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const long long ri = amt_rt_px_run(rt, blockIdx.y, (int)(i / W), (int)(i % W));
            int v = 0;
            if (ri >= 0) {
                const int f = F[base + amt_rt_run_root(rt, ri, W)];
                v = f > 0 ? f : (f == 0 ? markers[base + i] : 0);
            }
            out[base + i] = v;
        }
        return;
    }
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        if (i0 + 3 < n && ((base + i0) & 3) == 0) {
            const int4 r = *reinterpret_cast<const int4*>(L + base + i0);
            int4 v = make_int4(0, 0, 0, 0);
            if (r.x >= 0 || r.y >= 0 || r.z >= 0 || r.w >= 0) {
                const int4 m = *reinterpret_cast<const int4*>(markers + base + i0);
                // pixel -> tile root, whose F entry was copied from the component root (amt_i_propagate_roots)
                const int fx = r.x >= 0 ? F[base + r.x] : 0, fy = r.y >= 0 ? F[base + r.y] : 0;
                const int fz = r.z >= 0 ? F[base + r.z] : 0, fw = r.w >= 0 ? F[base + r.w] : 0;
                v.x = r.x >= 0 ? (fx > 0 ? fx : fx == 0 ? m.x : 0) : 0;
                v.y = r.y >= 0 ? (fy > 0 ? fy : fy == 0 ? m.y : 0) : 0;
                v.z = r.z >= 0 ? (fz > 0 ? fz : fz == 0 ? m.z : 0) : 0;
                v.w = r.w >= 0 ? (fw > 0 ? fw : fw == 0 ? m.w : 0) : 0;
            }
            *reinterpret_cast<int4*>(out + base + i0) = v;
        } else {
            for (size_t i = i0; i < n && i < i0 + 4; ++i) {
                const int r = L[base + i];
                int v = 0;
                if (r >= 0) {
                    const int f = F[base + r];
                    v = f > 0 ? f : (f == 0 ? markers[base + i] : 0);
                }
                out[base + i] = v;
            }
        }
    }
}

// compress the listed tile roots (every one then points straight at its component's root) and give every component
// root a dense 1-based id in T (arbitrary order; nothing in the output depends on the numbering)
__global__ void __launch_bounds__(256) ws_roots_kernel(int* __restrict__ Lall, int* __restrict__ Tall,
                                                       const int* __restrict__ rootlist, const int* __restrict__ nroots,
                                                       int* __restrict__ ncomp, size_t cap, size_t n) {
    const int plane = blockIdx.z, shard = plane * gridDim.y + blockIdx.y;  // one list per tile row
    int* L = Lall + (size_t)plane * n;
    int* T = Tall + (size_t)plane * n;
    const int cnt = nroots[shard] < (int)cap ? nroots[shard] : (int)cap;
    const int* lst = rootlist + (size_t)shard * cap;
    const int lane = threadIdx.x & 63;
    for (int k0 = blockIdx.x * 256; k0 < cnt; k0 += gridDim.x * 256) {  // block-uniform bounds: the ballot below is whole-wave
        const int k = k0 + threadIdx.x;
        bool is_root = false;
        int t = 0;
        if (k < cnt) {
            t = lst[k];
            int r = L[t];
            int p = L[r];
            while (p != r) {  // roots only ever point at smaller indices: the chain ends at the component root
                r = p;
                p = L[r];
            }
            if (r != t) L[t] = r;
            else is_root = true;
        }
        // one counter per plane: a returning atomic per ROOT queued ~1,000 deep on one address; now one per wave
        const unsigned long long m = __ballot(is_root);
        if (m) {
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&ncomp[plane], __popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (is_root) T[t] = base + __popcll(m & ((1ull << lane) - 1ull)) + 1;
        }
    }
}

__global__ void __launch_bounds__(256) ws_rows_init_kernel(comp_row* __restrict__ rows, const int* __restrict__ ncomp,
                                                           size_t plane_stride) {
    comp_row* r = rows + (size_t)blockIdx.y * plane_stride;
    const int nc = ncomp[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nc; i += gridDim.x * 256) {
        comp_row c;
        c.cmax = 0;
        c.mcnt = 0;
        c.x0 = c.y0 = 0x7fffffff;
        c.x1 = c.y1 = -1;
        c.labmin = 0x7fffffff;
        c.labmax = 0;
        c.cls = CLS_NONE;
        c.root = -1;
        c.npix = 0;
        r[i] = c;
    }
}

// per-component max d2 (bucket count) or size (heap capacity), marker count, marker label range and
// bounding box.  A wave covers 64 consecutive pixels of 8 rows (loads issued up front).  Few pixels touch
// the component row at all:
//   * run heads (first pixel of a run of equal roots) update the bounding box (and the size in heap mode);
//   * marker pixels (sparse) update the marker count and label range;
//   * max d2: only pixels that are >= their 4 neighbours inside the wave's 64 x 8 strip can be the component's
//     maximum, so only those issue an atomicMax (the true maximum always passes the test).
template <bool MARKER_PLANE>
__global__ void __launch_bounds__(256) ws_stats_kernel(const int* __restrict__ d2, const int* __restrict__ L,
                                                       const int* __restrict__ T, const int* __restrict__ markers,
                                                       comp_row* __restrict__ rows, size_t row_stride, int H, int W,
                                                       int use_d2) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int yb = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 8;
    if (yb >= H) return;
    const size_t base = (size_t)blockIdx.z * H * W;
    int rs[8], vs[8], ms[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int y = yb + k;
        rs[k] = (x < W && y < H) ? L[base + (size_t)y * W + x] : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = base + (size_t)(yb + k) * W + x;
        ms[k] = (MARKER_PLANE && rs[k] >= 0) ? markers[i] : 0;  // with a marker list the plane is not read at all
        vs[k] = 0;
    }
    // component row of every foreground pixel, gathered NOW with the marker / relief loads (a few hot lines: the
    // tile roots) -- fetched where it is needed, it was a dependent round trip per row
    int ts[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ts[k] = rs[k] >= 0 ? T[base + rs[k]] : 0;
    if (use_d2) {
        // raw values, nothing computed next to the loads (that would wait for each of them and serialise the
        // rows); negative reliefs need no clamp: only v > 0 can be a candidate, and v >= neighbour holds for a
        // negative neighbour exactly as it would for 0
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t i = base + (size_t)(yb + k) * W + x;
            vs[k] = rs[k] >= 0 ? d2[i] : 0;
        }
    }
    comp_row* prow = rows + (size_t)blockIdx.z * row_stride;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = rs[k];
        const unsigned long long fg = __ballot(r >= 0);
        if (!fg) continue;  // uniform
        const int left = amt_lane_left(r);
        const bool head = r >= 0 && ((lane == 0) || (left != r));
        const int v = vs[k];
        bool cand = false;
        if (use_d2) {
            const int vl = amt_lane_left(v), vr = amt_lane_right(v);
            cand = r >= 0 && v > 0 && (lane == 0 || v >= vl) && (lane == 63 || v >= vr) &&
                   (k == 0 || v >= vs[k > 0 ? k - 1 : 0]) && (k == 7 || v >= vs[k < 7 ? k + 1 : 7]);
        }
        const int lab = ms[k];
        // r is the TILE root of the pixel (equal inside a run, since a run never leaves its tile); its T entry was
        // copied from the component root (amt_i_propagate_roots)
        if (head || cand || lab != 0) {
            comp_row* c = prow + (ts[k] - 1);
            if (cand) atomicMax(&c->cmax, v);
            if (lab != 0) {
                atomicAdd(&c->mcnt, 1);
                atomicMin(&c->labmin, lab);
                atomicMax(&c->labmax, lab);
            }
        }
        // run geometry (needs the ballot of ALL lanes, so it sits outside the divergent branch)
        const unsigned long long hb = __ballot(head || r < 0);  // run boundaries: heads and background
        // a run with a pixel of its own component right above (below) it cannot be the component's first (last) row:
        // inside the strip's eight rows that spares four of five y atomics (the strip's first / last row always asks)
        const unsigned long long ab = __ballot(r >= 0 && k > 0 && rs[k > 0 ? k - 1 : 0] >= 0 && ts[k > 0 ? k - 1 : 0] == ts[k]);
        const unsigned long long bl = __ballot(r >= 0 && k < 7 && rs[k < 7 ? k + 1 : 7] >= 0 && ts[k < 7 ? k + 1 : 7] == ts[k]);
        // The same idea for the x extent (round 3; it cut the pass's ~5 M atomics per 48 planes): a run that starts at
        // column a cannot set the component's x0 if the pixel right above its first pixel belongs to the component (that
        // row's run starts at <= a) or the pixel below-left does (that row's run starts at < a) -- "above: <=, below: <"
        // cannot point in a circle, so every chain of such rows ends at one that asks.  Mirrored for x1 at the run's
        // LAST pixel.  Rows 0 / 7 of the strip know only one neighbour row and ask unless that one covers them.
        const int cme = r >= 0 ? ts[k] : 0;
        const int cab = (k > 0 && rs[k > 0 ? k - 1 : 0] >= 0) ? ts[k > 0 ? k - 1 : 0] : 0;   // component right above
        const int cbe = (k < 7 && rs[k < 7 ? k + 1 : 7] >= 0) ? ts[k < 7 ? k + 1 : 7] : 0;   // ... right below
        const int cbl = amt_lane_left(cbe), cbr = amt_lane_right(cbe);                       // below-left / below-right
        const int right = amt_lane_right(r);
        const bool tail = r >= 0 && (lane == 63 || right != r);
        if (tail && !(cab == cme || (lane < 63 && cbr == cme))) atomicMax(&(prow + (cme - 1))->x1, x);
        if (head) {
            const unsigned long long later = hb & ~((2ull << lane) - 1ull);
            const int end_lane = later ? (__ffsll((long long)later) - 2) : 63;
            const int len = end_lane - lane + 1;
            const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
            comp_row* c = prow + (ts[k] - 1);
            // the run that starts at the component root itself: the one pixel that is its own parent (r = L[pixel])
            if ((size_t)r == (size_t)(yb + k) * W + x) c->root = r;
            if (!use_d2) atomicAdd(&c->cmax, len);
            if (!(cab == cme || (lane > 0 && cbl == cme))) atomicMin(&c->x0, x);
            if (!(ab & run)) atomicMin(&c->y0, yb + k);
            if (!(bl & run)) atomicMax(&c->y1, yb + k);
        }
    }
}

// ---- the same statistics from the mask's run tables (round 3; amt_internal.h) ------------------------------------------
// ws_stats_kernel reads the parent plane (4 bytes per pixel) only to learn which component a pixel belongs to, and asks
// for a component's row once per pixel.  A tile's run table says the same with one entry per RUN: a wave per tile, first
// a lane per run (component id of the run's tile root; the ordinal of the root's own run), then
//   * a lane per ROW: bounding box from the run's ends, with the filters of ws_stats_kernel expressed on row words (a run
//     with a mask pixel right above / below it cannot be its component's first / last row, ...), and the component root;
//   * a lane per four pixels: d2 is read only where the mask is set, maxima are folded per TILE ROOT in LDS, one global
//     atomic per tile root.
// Tiles with more than SR_CAP runs (noise) take a per-run path without LDS tables.
constexpr int SR_CAP = 512;

__global__ void __launch_bounds__(256) ws_stats_runs_kernel(const int* __restrict__ d2all, const unsigned long long* __restrict__ tbits,
                                                            const unsigned short* __restrict__ rtab,
                                                            const int* __restrict__ nruns, const int* __restrict__ Lall,
                                                            const int* __restrict__ Tall, comp_row* __restrict__ rows,
                                                            size_t row_stride, int H, int W, int segs, int trows, int ntiles,
                                                            int* __restrict__ rcomp) {
    __shared__ int comp_s[4][SR_CAP];
    __shared__ int cm_s[4][SR_CAP];
    __shared__ int np_s[4][SR_CAP];
    __shared__ unsigned short kr_s[4][SR_CAP];
    __shared__ unsigned long long bits_s[4][64];
    __shared__ int off_s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wv;
    if (t >= ntiles) return;  // whole wave
    const int nr = nruns[t];
    if (nr == 0) return;
    const int bx = t % segs, ty = (t / segs) % trows, plane = t / (segs * trows);
    const size_t n = (size_t)H * W;
    const int* T = Tall + (size_t)plane * n;
    const int* L = Lall + (size_t)plane * n;
    const int* d2 = d2all + (size_t)plane * n;
    comp_row* prow = rows + (size_t)plane * row_stride;
    const int x0 = bx * 64, ty0 = ty * 64;
    const unsigned long long w = tbits[(size_t)t * 64 + lane];
    const unsigned long long heads = w & ~(w << 1);
    const int cnt = __popcll(heads);
    const int off = ccl_wave_incl_scan(cnt, lane) - cnt;
    bits_s[wv][lane] = w;
    off_s[wv][lane] = off;
    const bool fast = nr <= SR_CAP;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (fast) {
        for (int k = lane; k < nr; k += 64) {
            const int e = rtab[(size_t)t * RT_CAP + k];
            const int rrow = e >> 6, rcol = e & 63;
            const int cid = T[(ty0 + rrow) * W + x0 + rcol];
            comp_s[wv][k] = cid;
            rcomp[(size_t)t * RT_CAP + k] = cid;  // run -> component for the look-ups of the floods, marker lists, frame
            const unsigned long long rw = bits_s[wv][rrow];
            kr_s[wv][k] = (unsigned short)(off_s[wv][rrow] + __popcll((rw & ~(rw << 1)) & ((2ull << rcol) - 1ull)) - 1);
            cm_s[wv][k] = 0;
            np_s[wv][k] = 0;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    // ---- a lane per row: run geometry ----
    {
        const int y = ty0 + lane;
        const unsigned long long above = lane > 0 ? bits_s[wv][lane - 1] : 0ull;   // row 0 / 63 of the tile: unknown, always ask
        const unsigned long long below = lane < 63 ? bits_s[wv][lane + 1] : 0ull;
        int j = 0;
        for (unsigned long long h = heads; h; h &= h - 1, ++j) {
            const int b = __ffsll((long long)h) - 1;
            const unsigned long long tt = w >> b;
            const int len = ~tt ? __ffsll((long long)~tt) - 1 : 64;
            const unsigned long long rm = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << b;
            const int k = off + j;
            const int e = rtab[(size_t)t * RT_CAP + k];
            const int c = fast ? comp_s[wv][k] : T[(ty0 + (e >> 6)) * W + x0 + (e & 63)];
            if (!fast) rcomp[(size_t)t * RT_CAP + k] = c;
            comp_row* cr = prow + (c - 1);
            const int last = b + len - 1;
            if (!(above & rm)) atomicMin(&cr->y0, y);
            if (!(below & rm)) atomicMax(&cr->y1, y);
            // x0: a mask pixel right above the run's first pixel, or below it AND below-left of it, belongs to a run of
            // this component that starts at <= / < this one; mirrored for x1 ("above: <=, below: <" cannot form a circle)
            if (!(((above >> b) & 1ull) || (b > 0 && ((below >> (b - 1)) & 3ull) == 3ull))) atomicMin(&cr->x0, x0 + b);
            if (!(((above >> last) & 1ull) || (last < 63 && ((below >> last) & 3ull) == 3ull))) atomicMax(&cr->x1, x0 + last);
            if (e == ((lane << 6) | b)) {  // the tile root's own run; the component's root is its own parent
                const int p = y * W + x0 + b;
                if (L[p] == p) cr->root = p;
            }
            if (fast) atomicAdd(&np_s[wv][kr_s[wv][k]], len);  // pixels per tile root (the floods size their queues by it)
            if (!fast) {
                atomicAdd(&cr->npix, len);
                int m = 0;
                for (int x = b; x <= last; ++x) {
                    const int v = d2[(size_t)y * W + x0 + x];
                    m = v > m ? v : m;
                }
                if (m > 0) atomicMax(&cr->cmax, m);
            }
        }
    }
    if (!fast) return;
    // ---- a lane per four pixels: max d2 per tile root ----
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = x0 + c4;
    int4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = rsub + 4 * j;
        const unsigned nib = (unsigned)(bits_s[wv][row] >> c4) & 15u;
        v[j] = make_int4(0, 0, 0, 0);
        if (nib) v[j] = *reinterpret_cast<const int4*>(d2 + (size_t)(ty0 + row) * W + xg);  // mask bits exist inside the image only
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = rsub + 4 * j;
        const unsigned long long ww = bits_s[wv][row];
        const unsigned nib = (unsigned)(ww >> c4) & 15u;
        if (!nib) continue;
        const unsigned long long hw = ww & ~(ww << 1);
        const unsigned hnib = (unsigned)(hw >> c4) & 15u;
        const int k0 = off_s[wv][row] + __popcll(hw & ((1ull << c4) - 1ull)) - 1;  // the run open at the nibble's left edge
        const int vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        const int first = __ffs((int)nib) - 1;
        if ((hnib >> (first + 1)) == 0u) {
            // one run in these four pixels (the usual case): one LDS atomic
            int m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) m = ((nib >> i) & 1u) && vv[i] > m ? vv[i] : m;
            if (m > 0) atomicMax(&cm_s[wv][kr_s[wv][k0 + __popc(hnib & ((2u << first) - 1u))]], m);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (((nib >> i) & 1u) && vv[i] > 0) atomicMax(&cm_s[wv][kr_s[wv][k0 + __popc(hnib & ((2u << i) - 1u))]], vv[i]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nr; k += 64)
        if (kr_s[wv][k] == k) {
            comp_row* cr = prow + (comp_s[wv][k] - 1);
            if (cm_s[wv][k] > 0) atomicMax(&cr->cmax, cm_s[wv][k]);
            atomicAdd(&cr->npix, np_s[wv][k]);
        }
}

// marker statistics from the LIST of marker pixels (amt_label_sparse_reuse keeps it): a few thousand pixels per plane
// instead of a 4-byte read of every pixel of the marker plane in ws_stats_kernel
__global__ void __launch_bounds__(256) ws_marker_stats_kernel(const int* __restrict__ mk_list, const int* __restrict__ mk_count,
                                                              int mk_cap, const int* __restrict__ markers,
                                                              const int* __restrict__ L, const int* __restrict__ T,
                                                              comp_row* __restrict__ rows, size_t row_stride, size_t n,
                                                              amt_runtabs rt, int W) {
    const int plane = blockIdx.y;
    const int cnt = mk_count[plane] < mk_cap ? mk_count[plane] : mk_cap;
    const int* lst = mk_list + (size_t)plane * mk_cap;
    const size_t base = (size_t)plane * n;
    comp_row* prow = rows + (size_t)plane * row_stride;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int p = lst[k];
        int cid;
        if (rt.tbits) {
            const long long ri = amt_rt_px_run(rt, plane, p / W, p % W);
            if (ri < 0) continue;
            cid = rt.rcomp[ri];
        } else {
            const int r = L[base + p];
            if (r < 0) continue;  // markers * mask: a marker pixel outside the mask does not exist
            cid = T[base + r];
        }
        const int lab = markers[base + p];
        if (lab == 0) continue;
        comp_row* c = prow + (cid - 1);
        atomicAdd(&c->mcnt, 1);
        atomicMin(&c->labmin, lab);
        atomicMax(&c->labmax, lab);
    }
}

// classify components; publish marker-list sizes (moff) and queue sizes (boff) for the HBM path
__global__ void __launch_bounds__(256) ws_classify_kernel(comp_row* __restrict__ rows, const int* __restrict__ ncomp,
                                                          int* __restrict__ moff, int* __restrict__ boff,
                                                          int* __restrict__ has_g, int* __restrict__ wl,
                                                          int* __restrict__ wl_count, int* __restrict__ Fall, size_t n,
                                                          int nplanes, size_t row_stride, int use_d2, int pf_slots) {
    comp_row* r = rows + (size_t)blockIdx.y * row_stride;
    int* mo = moff + (size_t)blockIdx.y * row_stride;
    int* bo = boff + (size_t)blockIdx.y * row_stride;
    const int nc = ncomp[blockIdx.y];
    const int lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * 256; i0 < nc; i0 += gridDim.x * 256) {  // block-uniform bounds: whole-wave ballots below
        const int i = i0 + threadIdx.x;
        const bool live = i < nc;
        comp_row c = r[live ? i : 0];
        int cls;
        if (!live) {
            cls = CLS_NONE;
        } else if (c.mcnt == 0) {
            cls = CLS_NONE;
        } else if (c.labmin == c.labmax) {
            cls = CLS_UNIFORM;
        } else if (!use_d2) {
            cls = CLS_G;
        } else {
            const long long area = (long long)(c.x1 - c.x0 + 3) * (c.y1 - c.y0 + 3);  // with the sentinel ring
            if (c.labmax >= 0xFFFF) cls = CLS_G;  // labels are kept as 16-bit values in LDS
            else if (area <= XS_PX && c.cmax < XS_NB) cls = CLS_XS;
            else if (area <= S_PX && c.cmax < S_NB) cls = CLS_S;
            else if (area <= M_PX && c.cmax < M_NB) cls = CLS_M;
            else if (area <= M2_PX && c.cmax < M2_NB) cls = CLS_M2;
            else if (area <= L_PX && c.cmax < L_NB) {
                // bytes the persistent flood would reserve for it (6 per tile cell and per bucket, 2,560-byte slots)
                const long long qn = c.npix > 0 ? c.npix : area;  // queue entries: one per pixel of the component
                const long long need = (((area + 1) & ~1ll) * 4 + ((qn + 1) & ~1ll) * 2 + ((c.cmax + 2) & ~1) * 6 + 2559) / 2560;
                cls = (pf_slots > 0 && need > pf_slots) ? CLS_LB : CLS_L;
            }
            else if (area <= X_PX && c.cmax < X_NB) cls = CLS_X;
            else cls = CLS_G;
        }
        if (live) {
            r[i].cls = cls;
            // per-root fill value: the label of a single-label component, 0 = flooded (labels come from the flood),
            // -1 = no marker at all (stays background)
            Fall[(size_t)blockIdx.y * n + c.root] = cls == CLS_UNIFORM ? c.labmin : (cls == CLS_NONE ? -1 : 0);
            if (cls == CLS_G) has_g[blockIdx.y] = 1;
            mo[i] = cls == CLS_G ? c.mcnt : 0;
            bo[i] = cls == CLS_G ? (use_d2 ? c.cmax + 1 : c.cmax) : 0;
        }
        // per-class worklists of this plane (order is irrelevant: components are independent).  One slot counter per
        // class and plane: a returning atomic per COMPONENT queued hundreds deep on one address (the kernel took 60 us
        // for 50,000 components); a wave now reserves its slots with one atomic per class it holds
        for (int k = 0; k <= CLS_LB - CLS_XS; ++k) {
            const unsigned long long m = __ballot(live && cls == CLS_XS + k);
            if (!m) continue;
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&wl_count[k * nplanes + blockIdx.y], __popcll(m));
            base = __shfl(base, leader);
            if (live && cls == CLS_XS + k)
                wl[((size_t)k * nplanes + blockIdx.y) * row_stride + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
}

// marker lists of the HBM-path components (unordered fill; each lane sorts its own list)
__global__ void __launch_bounds__(256) ws_fill_markers_kernel(const int* __restrict__ L, const int* __restrict__ T,
                                                              const int* __restrict__ out,
                                                              const comp_row* __restrict__ rows,
                                                              const int* __restrict__ moff, int* __restrict__ cursor,
                                                              int* __restrict__ mlist, const int* __restrict__ has_g,
                                                              size_t row_stride, size_t n, amt_runtabs rt, int W) {
    if (!has_g[blockIdx.y]) return;  // the common case: every component fitted an LDS tile
    const size_t base = (size_t)blockIdx.y * n;
    const size_t cb = (size_t)blockIdx.y * row_stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int cid;
        if (rt.tbits) {
            if (out[base + i] == 0) continue;
            const long long ri = amt_rt_px_run(rt, blockIdx.y, (int)(i / W), (int)(i % W));
            if (ri < 0) continue;
            cid = rt.rcomp[ri] - 1;
        } else {
            const int r = L[base + i];
            if (r < 0 || out[base + i] == 0) continue;
            cid = T[base + r] - 1;
        }
        if (rows[cb + cid].cls != CLS_G) continue;
        const int pos = atomicAdd(&cursor[cb + cid], 1);
        mlist[base + moff[cb + cid] + pos] = (int)i;
    }
}

// exclusive scan of d[0 .. len) in place by one workgroup of 1024 threads; returns the total to every thread
__device__ __forceinline__ int ws_block_scan_excl(int* d, int len, int* s, int* carry) {
    if (threadIdx.x == 0) *carry = 0;
    __syncthreads();
    for (int start = 0; start < len; start += 1024) {
        const int i = start + threadIdx.x;
        const int v = i < len ? d[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const int c = *carry;
        if (i < len) d[i] = c + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) *carry = c + s[1023];
        __syncthreads();
    }
    return *carry;
}

// The HBM flood's preparation on the fused path, ONE launch: a workgroup per plane that leaves at once unless the plane
// holds a component too large for an LDS tile (almost never on nuclei) -- what used to be six launches that each found
// nothing to do (seed of the plane, two scans, two fills, the marker lists).  A plane that does hold one is prepared by
// its one workgroup: slow (two passes of 1,024 threads over the plane), but the flood of such a component takes longer.
__global__ void __launch_bounds__(1024) ws_g_prep_kernel(const int* __restrict__ markers, const int* __restrict__ Lall,
                                                         const int* __restrict__ Tall, const int* __restrict__ Fall,
                                                         int* __restrict__ outall, const comp_row* __restrict__ rows,
                                                         int* __restrict__ moff, int* __restrict__ boff,
                                                         int* __restrict__ cursor, int* __restrict__ mlist,
                                                         int* __restrict__ head, const int* __restrict__ ncomp,
                                                         int* __restrict__ mtot, int* __restrict__ btot,
                                                         const int* __restrict__ has_g, size_t row_stride, size_t n,
                                                         size_t bstride, amt_runtabs rt, int W) {
    const int plane = blockIdx.x;
    if (!has_g[plane]) {
        if (threadIdx.x == 0) mtot[plane] = btot[plane] = 0;
        return;
    }
    __shared__ int s[1024];
    __shared__ int carry;
    const size_t base = (size_t)plane * n, cb = (size_t)plane * row_stride;
    const int* L = Lall + base;
    const int* T = Tall + base;
    const int* F = Fall + base;
    int* out = outall + base;
    const int nc = ncomp[plane];
    // out = label of the component for single-label components, markers * mask elsewhere (ws_seed_kernel)
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        int r;
        if (rt.tbits) {
            const long long ri = amt_rt_px_run(rt, plane, (int)(i / W), (int)(i % W));
            r = ri < 0 ? -1 : amt_rt_run_root(rt, ri, W);
        } else {
            r = L[i];
        }
        int v = 0;
        if (r >= 0) {
            const int f = F[r];
            v = f > 0 ? f : (f == 0 ? markers[base + i] : 0);
        }
        out[i] = v;
    }
    const int mt = ws_block_scan_excl(moff + cb, nc, s, &carry);
    const int bt = ws_block_scan_excl(boff + cb, nc, s, &carry);
    if (threadIdx.x == 0) {
        mtot[plane] = mt;
        btot[plane] = bt;
    }
    for (int i = threadIdx.x; i < nc; i += 1024) cursor[cb + i] = 0;
    for (int i = threadIdx.x; i < bt; i += 1024) head[(size_t)plane * bstride + i] = -1;
    __syncthreads();  // `out`, the scanned offsets and the cursors are this workgroup's own writes
    // marker lists of the HBM-path components (ws_fill_markers_kernel)
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        if (out[i] == 0) continue;
        int cid;
        if (rt.tbits) {
            const long long ri = amt_rt_px_run(rt, plane, (int)(i / W), (int)(i % W));
            if (ri < 0) continue;
            cid = rt.rcomp[ri] - 1;
        } else {
            const int r = L[i];
            if (r < 0) continue;
            cid = T[r] - 1;
        }
        if (rows[cb + cid].cls != CLS_G) continue;
        const int pos = atomicAdd(&cursor[cb + cid], 1);
        mlist[base + moff[cb + cid] + pos] = (int)i;
    }
}

__global__ void __launch_bounds__(256) ws_fill_value_kernel(int* __restrict__ buf, const int* __restrict__ total,
                                                            size_t plane_stride, int value) {
    int* b = buf + (size_t)blockIdx.y * plane_stride;
    const int tot = total[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) b[i] = value;
}

__global__ void __launch_bounds__(256) ws_init_kernel(int* __restrict__ a, int na, int* __restrict__ b, int nb, int* __restrict__ c,
                                                      int nc) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) a[i] = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) b[i] = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nc; i += gridDim.x * 256) c[i] = 0;
}

__global__ void ws_zero_counters_kernel(int* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = 0;
}

__device__ __forceinline__ void lane_sort(int* a, int n) {
    for (int i = 1; i < n; ++i) {
        int v = a[i], j = i - 1;
        while (j >= 0 && a[j] > v) {
            a[j + 1] = a[j];
            --j;
        }
        a[j + 1] = v;
    }
}

// ---- LDS-tile flood (bucket queue, relief = -sqrt(d2)) ----------------------------------------------
// One wave per component.  The bounding box plus a one-pixel sentinel ring is staged in LDS:
//   cell[i] (u32) = label (low 16 bits; 0 = unclaimed, 0xFFFF = not in this component / ring)
//                   | field << 16 (15 bits) | marker flag (bit 31)
//                   field = d2 while the pixel is unclaimed; once the pixel has been pushed its d2 is dead
//                   and the field becomes the FIFO link to the pixel pushed after it into the same bucket
//   ht[b]   (u32) = head (low 16) | tail (high 16) of bucket b; head 0xFFFF = empty
// A single wave issues about one instruction every four clocks, so the flood is bound by the number of
// instructions per pop, not by LDS latency.  Per pop: lanes 0..3 fetch the four neighbour cells and lane 4
// the popped cell (label + link) in ONE ds_read; the claim test and the label write are one vector
// compare / one masked ds_write; only the claimed neighbours (one per pop on average) go through the
// scalar FIFO append.  The current bucket's head/tail live in scalar registers.  The sentinel ring
// removes all bounds arithmetic.
template <int TILE_PX, int NB, int CLS>
__global__ void __launch_bounds__(64) ws_flood_lds_kernel(const int* __restrict__ d2all, const int* __restrict__ Lall,
                                                          const int* __restrict__ Tall, int* __restrict__ outall,
                                                          const comp_row* __restrict__ rows,
                                                          const int* __restrict__ wl, const int* __restrict__ wl_count,
                                                          int* __restrict__ counters, size_t row_stride, int H, int W,
                                                          int seeds_first, int* __restrict__ ties,
                                                          const int* __restrict__ mkall, amt_runtabs rt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned* cell = reinterpret_cast<unsigned*>(smem_raw);
    unsigned* ht = cell + TILE_PX;
    unsigned short* half = reinterpret_cast<unsigned short*>(cell);  // half[2 * i + 1] = field of cell i
    const int plane = blockIdx.y;
    const size_t n = (size_t)H * W;
    const int* d2 = d2all + (size_t)plane * n;
    const int* L = Lall + (size_t)plane * n;
    const int* T = Tall + (size_t)plane * n;
    int* out = outall + (size_t)plane * n;
    const int* mk = mkall + (size_t)plane * n;  // marker labels (the membership test keeps them inside the mask)
    const comp_row* rr = rows + (size_t)plane * row_stride;
    const int* mylist = wl + (size_t)plane * row_stride;
    const int nwork = wl_count[plane];
    const int lane = threadIdx.x;
    while (true) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&counters[plane], 1);
        k = __builtin_amdgcn_readfirstlane(k);  // uniform: the component row and all flood state stay scalar
        if (k >= nwork) break;
        const int c = mylist[k];
        const comp_row cr = rr[c];
        const int tw = cr.x1 - cr.x0 + 3, th = cr.y1 - cr.y0 + 3;  // padded tile
        const int npx = tw * th;
        const int nb = cr.cmax + 1;
        const unsigned inv_tw = 0xFFFFFFFFu / (unsigned)tw + 1u;  // i / tw == (i * inv_tw) >> 32 for i < 65536
        // ---- stage the bounding box + ring: membership is "root of the pixel == root of the component", so
        // the three loads of a pixel are independent; 4 pixels per lane = 12 loads in flight ----
        for (int i0 = 0; i0 < npx; i0 += 256) {
            int rr4[4], o4[4], d4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 64 + lane;
                const int ty = (int)(((unsigned long long)(unsigned)i * inv_tw) >> 32), tx = i - ty * tw;
                rr4[u] = -1;
                o4[u] = 0;
                d4[u] = 0;
                if (i < npx && ty >= 1 && ty < th - 1 && tx >= 1 && tx < tw - 1) {
                    const size_t g = (size_t)(cr.y0 + ty - 1) * W + (cr.x0 + tx - 1);
                    if (rt.tbits) {
                        const long long ri = amt_rt_px_run(rt, plane, cr.y0 + ty - 1, cr.x0 + tx - 1);
                        rr4[u] = (int)ri;
                    } else {
                        rr4[u] = L[g];
                    }
                    o4[u] = mk[g];
                    d4[u] = d2[g];
                }
            }
            const int* idtab = rt.tbits ? rt.rcomp : T;
#pragma unroll
            for (int u = 0; u < 4; ++u) rr4[u] = rr4[u] >= 0 ? idtab[rr4[u]] : 0;  // tile root (or run) -> component id (1-based)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 64 + lane;
                if (i < npx) {
                    unsigned cv = 0xFFFFu;
                    if (rr4[u] == c + 1) {
                        const int d = d4[u] < 0 ? 0 : d4[u];
                        cv = ((unsigned)o4[u] & 0xFFFFu) | ((unsigned)d << 16) | (o4[u] != 0 ? 0x80000000u : 0u);
                    }
                    cell[i] = cv;
                }
            }
        }
        for (int i = lane; i < nb; i += 64) ht[i] = 0xFFFFFFFFu;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // ---- the flood: one sequential thread of control; the WHOLE wave executes it in lock step so that
        // all state is provably uniform (SGPRs, scalar branches); lanes >= 5 just mirror lane 4 ----
        constexpr unsigned NONE = 0xFFFFu;
        int cur = -1;                   // current bucket, -1 = none yet
        unsigned head = NONE, tail = 0; // FIFO of the current bucket (register copy; ht[cur] is stale)
        int raise = -1;                 // highest bucket above `cur` that received a push since the last switch
        const int offk = lane == 0 ? -tw : lane == 1 ? -1 : lane == 2 ? 1 : lane == 3 ? tw : 0;
        auto uni = [&](unsigned v) -> unsigned { return __builtin_amdgcn_readfirstlane(v); };
        auto push = [&](int q, int b) {
            if (b == cur) {
                if (head == NONE) head = (unsigned)q; else half[2 * tail + 1] = (unsigned short)q;
                tail = (unsigned)q;
            } else {
                const unsigned h = uni(ht[b]);
                if ((h & 0xFFFFu) == NONE) {
                    ht[b] = (unsigned)q | ((unsigned)q << 16);
                } else {
                    half[2 * (h >> 16) + 1] = (unsigned short)q;
                    ht[b] = (h & 0xFFFFu) | ((unsigned)q << 16);
                }
                raise = b > raise ? b : raise;
            }
        };
        // claim the unlabelled neighbours of p in the order N, W, E, S (lanes 0..3; lane 4 fetches cell[p]).
        // With `pop`, p is the head of the current bucket and is unlinked BEFORE its neighbours are pushed.
        auto spread = [&](int p, bool pop) {
            const int q = p + offk;
            const unsigned c = cell[q];
            const unsigned cp = (unsigned)__builtin_amdgcn_readlane((int)c, 4);
            if (pop) head = (head == tail) ? NONE : (cp >> 16);  // the field of a linked cell is its successor
            const bool claim = (c & 0xFFFFu) == 0;
            unsigned m = (unsigned)__ballot(claim);
            if (m) {
                if (claim) cell[q] = c | (cp & 0xFFFFu);
                do {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    const unsigned ck = (unsigned)__builtin_amdgcn_readlane((int)c, k);
                    push(__builtin_amdgcn_readlane(q, k), (int)((ck >> 16) & 0x7FFF));
                } while (m);
            }
        };
        // ---- markers in raster order (wave ballots) ----
        for (int i0 = 0; i0 < npx; i0 += 64) {
            const int i = i0 + lane;
            const bool ismk = i < npx && (cell[i] & 0x80000000u);
            unsigned long long m = __ballot(ismk);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int p = i0 + b;
                if (seeds_first) {
                    spread(p, false);
                } else {
                    const int bk = (int)((uni(cell[p]) >> 16) & 0x7FFF);
                    // a second marker in one bucket = two age-0 entries of equal value in this component
                    const bool occupied = bk == cur ? head != NONE : (uni(ht[bk]) & 0xFFFFu) != NONE;
                    if (occupied && lane == 0) ties[plane] = 1;
                    push(p, bk);
                }
            }
        }
        {
            while (true) {
                if (raise > cur) {  // a push landed above the current bucket: switch to it
                    if (cur >= 0) ht[cur] = head == NONE ? 0xFFFFFFFFu : (head | (tail << 16));
                    cur = raise;
                    const unsigned h = uni(ht[cur]);
                    head = h & 0xFFFFu;
                    tail = h >> 16;
                }
                raise = -1;
                if (head == NONE) {  // current bucket exhausted: walk down, 64 buckets per LDS round trip
                    if (cur >= 0) ht[cur] = 0xFFFFFFFFu;
                    while (cur >= 0) {
                        const int bi = cur - 1 - lane;
                        const unsigned h = bi >= 0 ? ht[bi] : 0xFFFFFFFFu;
                        const unsigned long long m = __ballot((h & 0xFFFFu) != NONE);
                        if (m) {
                            const int k = __ffsll((long long)m) - 1;
                            cur = cur - 1 - k;
                            const unsigned hk = (unsigned)__builtin_amdgcn_readlane((int)h, k);
                            head = hk & 0xFFFFu;
                            tail = hk >> 16;
                            break;
                        }
                        cur -= 64;
                    }
                    if (cur < 0) break;
                }
                spread((int)head, true);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- write back ----
        for (int i = lane; i < npx; i += 64) {
            const unsigned lv = cell[i] & 0xFFFFu;
            if (lv != 0xFFFFu) {
                const int ty = (int)(((unsigned long long)(unsigned)i * inv_tw) >> 32), tx = i - ty * tw;
                out[(size_t)(cr.y0 + ty - 1) * W + (cr.x0 + tx - 1)] = (int)lv;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- LDS-tile flood, a whole FIFO at a time ------------------------------------------------------------------
// The sequential flood pops ONE pixel per step; on an EDT relief a bucket (= one value of d2) of a cluster of nuclei
// holds tens of pixels, and consecutive pops of one bucket interact only through (a) the cells they claim and (b) the
// order of their pushes.  So a step takes up to 64 entries off the head of the current bucket, one per lane, and
// reproduces the sequential outcome exactly:
//   * every lane reads its four neighbour cells; a neighbour that is unlabelled at the start of the step is a target;
//   * if a target lies in a HIGHER bucket than the current one, the sequential flood would continue there right after
//     that pop: the step is cut after the first such lane (it certainly claims that cell -- an earlier claimant would
//     itself be the first such lane), the later entries stay queued;
//   * a cell targeted by several (lane, neighbour) pairs goes to the smallest (lane, N-W-E-S) pair, as when popped one
//     by one: every pair does an LDS atomic max of a priority ticket into the label field, then re-reads it; the
//     winner replaces the ticket with its label (targets were unlabelled, so tickets cannot be mistaken for labels);
//   * winners are appended to the bucket of their cell in (lane, neighbour) order = sequential push order: for each
//     distinct destination bucket the ranks come from ballots.
// A pixel is pushed at most once and always into the bucket of its own d2, so a bucket's queue is an array segment
// whose size is known up front: the staging pass histograms d2 over the component, a wave scan turns the counts into
// segment offsets, and a bucket is two cursors -- no links to chase.
//   cell[i]  (u32) = label (low 16; 0 = unclaimed, 0xFFFF = not in this component / ring) | d2 << 16 | marker << 31
//   queue[i] (u16) = tile indices, bucket segments back to back
//   off[b], hd[b], tl[b] (u16): segment start, next entry to pop, next free slot
#ifdef WS_STATS
__device__ unsigned long long ws_dbg[8];
extern "C" int amt_ws_debug_read(unsigned long long* host8) {
    return hipMemcpyFromSymbol(host8, HIP_SYMBOL(ws_dbg), 64) == hipSuccess ? 0 : -2;
}
extern "C" int amt_ws_debug_reset() {
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(ws_dbg), z, 64) == hipSuccess ? 0 : -2;
}
#endif
// One component, flooded by one wave in the LDS arrays it is given (cell: npx words, cnt: nb words, offs: nb halves, queue:
// npx halves).  Shared by the per-class kernels (fixed layout per class) and the persistent kernel (layout per component).
__device__ __forceinline__ void ws_flood_component(unsigned* cell, unsigned* cnt, unsigned short* offs, unsigned short* queue,
                                                   const int* __restrict__ d2, const int* __restrict__ L,
                                                   const int* __restrict__ T, int* __restrict__ out,
                                                   const int* __restrict__ mk, const comp_row& cr, int c, int W,
                                                   int seeds_first, int* __restrict__ ties, int plane, int lane,
                                                   const amt_runtabs& rt) {
    auto uni = [&](unsigned v) -> unsigned { return __builtin_amdgcn_readfirstlane(v); };
    const int tw = cr.x1 - cr.x0 + 3, th = cr.y1 - cr.y0 + 3;  // padded tile
    const int npx = tw * th;
    const int nb = cr.cmax + 1;
#ifdef WS_STATS
    if (lane == 0) { atomicAdd(&ws_dbg[3], 1ull); atomicAdd(&ws_dbg[4], (unsigned long long)npx); atomicAdd(&ws_dbg[5], (unsigned long long)nb); }
#endif
    const unsigned inv_tw = 0xFFFFFFFFu / (unsigned)tw + 1u;  // i / tw == (i * inv_tw) >> 32 for i < 65536
    for (int i = lane; i < nb; i += 64) cnt[i] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- stage the bounding box + ring, histogram d2 over the component ----
    // Eight tile cells per lane and round: 24 unconditional loads from clamped coordinates in flight, then the eight
    // dependent tile-root -> component-id gathers; validity is applied where the values are used (a conditional load
    // would put a wait behind each of them).  A 10,000-pixel box is staged in 20 round-trip pairs instead of 40.
    constexpr int SU = 6;
    for (int i0 = 0; i0 < npx; i0 += 64 * SU) {
        int rr4[SU], o4[SU], d4[SU];
        bool in4[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 64 + lane;
            const int ty = (int)(((unsigned long long)(unsigned)i * inv_tw) >> 32), tx = i - ty * tw;
            in4[u] = i < npx && ty >= 1 && ty < th - 1 && tx >= 1 && tx < tw - 1;
            int gy = cr.y0 + ty - 1, gx = cr.x0 + tx - 1;
            gy = gy < cr.y0 ? cr.y0 : (gy > cr.y1 ? cr.y1 : gy);
            gx = gx < cr.x0 ? cr.x0 : (gx > cr.x1 ? cr.x1 : gx);
            const size_t g = (size_t)gy * W + gx;
            if (rt.tbits) {
                // no parent plane: the pixel's row word and row offset now, its run's component id in the second round
                const size_t tl = ((size_t)plane * rt.trows + (gy >> 6)) * rt.segs + (gx >> 6);
                const unsigned long long wd = rt.tbits[tl * 64 + (gy & 63)];
                const int ro = rt.roff[tl * 64 + (gy & 63)];
                const int col = gx & 63;
                const int kk = ro + __popcll((wd & ~(wd << 1)) & ((2ull << col) - 1ull)) - 1;
                rr4[u] = ((wd >> col) & 1ull) ? (int)(tl * RT_CAP) + kk : -1;
            } else {
                rr4[u] = L[g];
            }
            o4[u] = mk[g];
            d4[u] = d2[g];
        }
        int id4[SU];
        const int* idtab = rt.tbits ? rt.rcomp : T;
#pragma unroll
        for (int u = 0; u < SU; ++u) id4[u] = idtab[rr4[u] < 0 ? 0 : rr4[u]];  // tile root (or run) -> component id (1-based)
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 64 + lane;
            if (i < npx) {
                unsigned cv = 0xFFFFu;
                if (in4[u] && rr4[u] >= 0 && id4[u] == c + 1) {
                    const int d = d4[u] < 0 ? 0 : d4[u];
                    cv = ((unsigned)o4[u] & 0xFFFFu) | ((unsigned)d << 16) | (o4[u] != 0 ? 0x80000000u : 0u);
                    atomicAdd(&cnt[d], 1u);
                }
                cell[i] = cv;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- counts -> segment offsets (each lane owns a contiguous run of buckets), cursors = segment start ----
    {
        const int per = (nb + 63) / 64;
        const int b0 = lane * per;
        int sum = 0;
        for (int j = 0; j < per; ++j) sum += (b0 + j < nb) ? (int)cnt[b0 + j] : 0;
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        int run = incl - sum;
        for (int j = 0; j < per; ++j) {
            if (b0 + j < nb) {
                const int cj = (int)cnt[b0 + j];
                offs[b0 + j] = (unsigned short)run;
                cnt[b0 + j] = (unsigned)run | ((unsigned)run << 16);  // hd = tl = segment start
                run += cj;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int cur = -1;    // current bucket (-1: none yet)
    int raise = -1;  // highest bucket that received a push since the last switch
    // a lane's own cell of the ring's top row (cells 0 .. tw - 1, all 0xFFFF and never a target): where its LDS operations
    // go when it has nothing to claim, so that they need no divergent block and do not pile up on one address
    const int dmy = lane - (int)(((unsigned long long)(unsigned)lane * inv_tw) >> 32) * tw;
    const int offN = -tw, offW = -1, offE = 1, offS = tw;
    // One step over the pixels p (one per active lane, lane order = pop order).  `limit`: the bucket the pixels
    // were popped from (a target above it cuts the step) or a value above every bucket (marker spreading).
    // Returns the number of lanes that took part.
    auto step = [&](int p, bool active, int limit, bool marker_push) -> int {
        unsigned cN = 0xFFFFu, cW = 0xFFFFu, cE = 0xFFFFu, cS = 0xFFFFu, cp = 0;
        const int pa = active ? p : 0;
        if (!marker_push) {
            cN = cell[pa + offN];
            cW = cell[pa + offW];
            cE = cell[pa + offE];
            cS = cell[pa + offS];
        }
        cp = cell[pa];
        bool tN = active && !marker_push && (cN & 0xFFFFu) == 0, tW = active && !marker_push && (cW & 0xFFFFu) == 0;
        bool tE = active && !marker_push && (cE & 0xFFFFu) == 0, tS = active && !marker_push && (cS & 0xFFFFu) == 0;
        const int bN = (int)((cN >> 16) & 0x7FFF), bW = (int)((cW >> 16) & 0x7FFF);
        const int bE = (int)((cE >> 16) & 0x7FFF), bS = (int)((cS >> 16) & 0x7FFF);
        const bool higher = (tN && bN > limit) || (tW && bW > limit) || (tE && bE > limit) || (tS && bS > limit);
        const unsigned long long hm = __ballot(higher);
        const unsigned long long am = __ballot(active);
        int took = __popcll(am);
        if (hm) {
            const int first = __ffsll((long long)hm) - 1;
            took = __popcll(am & ((2ull << first) - 1ull));
            if (lane > first) tN = tW = tE = tS = false;
        }
        bool wN = false, wW = false, wE = false, wS = false;
        int qN = pa + offN, qW = pa + offW, qE = pa + offE, qS = pa + offS;
        int dN = bN, dW = bW, dE = bE, dS = bS;
        if (marker_push) {  // the pixel itself is queued (markers entering by value): one "claim" per lane, slot N
            wN = active;
            qN = pa;
            dN = (int)((cp >> 16) & 0x7FFF);
        } else if (__ballot(tN || tW || tE || tS)) {
            // tickets: the smallest (lane, neighbour) pair holds the largest ticket
            const unsigned tk = 0x100u - (unsigned)(lane * 4);
            // (the atomics keep their blocks: unconditional ones on per-lane ring cells measured no faster)
            if (tN) atomicMax(&cell[qN], (cN & 0xFFFF0000u) | (tk - 0));
            if (tW) atomicMax(&cell[qW], (cW & 0xFFFF0000u) | (tk - 1));
            if (tE) atomicMax(&cell[qE], (cE & 0xFFFF0000u) | (tk - 2));
            if (tS) atomicMax(&cell[qS], (cS & 0xFFFF0000u) | (tk - 3));
            // re-reads and label writes WITHOUT a divergent block each (a lone wave pays every exec save / branch /
            // restore in full): lanes without a claim read and rewrite cell `dmy`, theirs in the top row of the sentinel ring that no
            // flood ever claims (its value, 0xFFFF, is what they write back)
            const unsigned rN = cell[tN ? qN : dmy], rW = cell[tW ? qW : dmy], rE = cell[tE ? qE : dmy], rS = cell[tS ? qS : dmy];
            wN = tN && (rN & 0xFFFFu) == tk - 0;
            wW = tW && (rW & 0xFFFFu) == tk - 1;
            wE = tE && (rE & 0xFFFFu) == tk - 2;
            wS = tS && (rS & 0xFFFFu) == tk - 3;
            const unsigned lab = cp & 0xFFFFu;
            cell[wN ? qN : dmy] = wN ? ((cN & 0xFFFF0000u) | lab) : 0xFFFFu;
            cell[wW ? qW : dmy] = wW ? ((cW & 0xFFFF0000u) | lab) : 0xFFFFu;
            cell[wE ? qE : dmy] = wE ? ((cE & 0xFFFF0000u) | lab) : 0xFFFFu;
            cell[wS ? qS : dmy] = wS ? ((cS & 0xFFFF0000u) | lab) : 0xFFFFu;
        }
        // append the winners, one destination bucket at a time, in (lane, N-W-E-S) order.  Every bucket is handled
        // once per step, so its cursor word is fetched for all claims up front (one LDS round trip, not one per
        // bucket).  (Measured alternative: reserving slots with returning LDS atomics and ranking the tickets of a
        // run afterwards was 3x slower -- the claims of a step go to FEW buckets, iso-distance contours, with many
        // claims each.)
        unsigned long long pend = __ballot(wN || wW || wE || wS);
        unsigned hN = 0, hW = 0, hE = 0, hS = 0;
        if (pend) {  // (uniform) unconditional reads, bucket 0 for lanes without a claim
            hN = cnt[wN ? dN : 0];
            hW = cnt[wW ? dW : 0];
            hE = cnt[wE ? dE : 0];
            hS = cnt[wS ? dS : 0];
        }
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef WS_STATS
        if (lane == 0 && !marker_push) { atomicAdd(&ws_dbg[0], 1ull); atomicAdd(&ws_dbg[1], (unsigned long long)took); }
#endif
        while (pend) {
#ifdef WS_STATS
            if (lane == 0) atomicAdd(&ws_dbg[2], 1ull);
#endif
            const int l0 = __ffsll((long long)pend) - 1;
            const int sel = wN ? dN : wW ? dW : wE ? dE : dS;  // this lane's first pending destination
            const unsigned selh = wN ? hN : wW ? hW : wE ? hE : hS;
            const int bsel = __builtin_amdgcn_readlane(sel, l0);
            const unsigned ht = (unsigned)__builtin_amdgcn_readlane((int)selh, l0);
            const bool mN = wN && dN == bsel, mW = wW && dW == bsel, mE = wE && dE == bsel, mS = wS && dS == bsel;
            const unsigned long long sN = __ballot(mN), sW = __ballot(mW), sE = __ballot(mE), sS = __ballot(mS);
            const int before = __popcll(sN & lt_mask) + __popcll(sW & lt_mask) + __popcll(sE & lt_mask) +
                               __popcll(sS & lt_mask);
            const int total = __popcll(sN) + __popcll(sW) + __popcll(sE) + __popcll(sS);
            const int base = (int)(ht >> 16);
            if (marker_push && ((ht >> 16) != (ht & 0xFFFFu) || total > 1) && lane == 0) ties[plane] = 1;
            int r = base + before;
            if (mN) queue[r++] = (unsigned short)qN;
            if (mW) queue[r++] = (unsigned short)qW;
            if (mE) queue[r++] = (unsigned short)qE;
            if (mS) queue[r++] = (unsigned short)qS;
            if (lane == l0) cnt[bsel] = (ht & 0xFFFFu) | ((unsigned)(base + total) << 16);
            raise = bsel > raise ? bsel : raise;
            wN = wN && !mN;
            wW = wW && !mW;
            wE = wE && !mE;
            wS = wS && !mS;
            pend = __ballot(wN || wW || wE || wS);
        }
        return took;
    };
    // ---- markers in raster order, 64 tile cells at a time ----
    for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        const bool ismk = i < npx && (cell[i] & 0x80000000u);
        if (__ballot(ismk)) step(i, ismk, 0x7FFFFFFF, !seeds_first);
    }
    // ---- the flood ----
    while (true) {
        if (raise > cur) cur = raise;
        raise = -1;
        unsigned ht = cur >= 0 ? uni(cnt[cur]) : 0u;
        if (cur < 0 || (ht & 0xFFFFu) == (ht >> 16)) {  // current bucket exhausted: walk down, 64 buckets a time
            int found = -1;
            int top = cur < 0 ? nb : cur;  // buckets below `top` are candidates
            while (top > 0) {
                const int bi = top - 1 - lane;
                const unsigned h = bi >= 0 ? cnt[bi] : 0u;
                const unsigned long long m = __ballot(bi >= 0 && (h & 0xFFFFu) != (h >> 16));
                if (m) {
                    found = top - 1 - (__ffsll((long long)m) - 1);
                    break;
                }
                top -= 64;
            }
            if (found < 0) break;
            cur = found;
            ht = uni(cnt[cur]);
        }
        const int hd = (int)(ht & 0xFFFFu), tl = (int)(ht >> 16);
        const int navail = tl - hd < 64 ? tl - hd : 64;
        const bool active = lane < navail;
        const int p = active ? (int)queue[hd + lane] : 0;
        const int took = step(p, active, cur, false);  // pushes into the current bucket only move its tl
        if (lane == 0) {
            const unsigned now = cnt[cur];  // tl may have grown
            cnt[cur] = (unsigned)(hd + took) | (now & 0xFFFF0000u);
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- write back ----
    for (int i = lane; i < npx; i += 64) {
        const unsigned lv = cell[i] & 0xFFFFu;
        if (lv != 0xFFFFu) {
            const int ty = (int)(((unsigned long long)(unsigned)i * inv_tw) >> 32), tx = i - ty * tw;
            out[(size_t)(cr.y0 + ty - 1) * W + (cr.x0 + tx - 1)] = (int)lv;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int TILE_PX, int NB>
__global__ void __launch_bounds__(64) ws_flood_batch_kernel(const int* __restrict__ d2all, const int* __restrict__ Lall,
                                                            const int* __restrict__ Tall, int* __restrict__ outall,
                                                            const comp_row* __restrict__ rows,
                                                            const int* __restrict__ wl, const int* __restrict__ wl_count,
                                                            int* __restrict__ counters, size_t row_stride, int H, int W,
                                                            int seeds_first, int* __restrict__ ties,
                                                            const int* __restrict__ mkall, amt_runtabs rt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned* cell = reinterpret_cast<unsigned*>(smem_raw);
    unsigned* cnt = cell + TILE_PX;                                       // NB words: histogram, then {hd | tl << 16}
    unsigned short* offs = reinterpret_cast<unsigned short*>(cnt + NB);  // NB segment starts
    unsigned short* queue = offs + NB;                                    // TILE_PX entries
    const int plane = blockIdx.y;
    const size_t n = (size_t)H * W;
    const int* d2 = d2all + (size_t)plane * n;
    const int* L = Lall + (size_t)plane * n;
    const int* T = Tall + (size_t)plane * n;
    int* out = outall + (size_t)plane * n;
    const int* mk = mkall + (size_t)plane * n;  // marker labels (the membership test keeps them inside the mask)
    const comp_row* rr = rows + (size_t)plane * row_stride;
    const int* mylist = wl + (size_t)plane * row_stride;
    const int nwork = wl_count[plane];
    const int lane = threadIdx.x;
    while (true) {
        int kk = 0;
        if (lane == 0) kk = atomicAdd(&counters[plane], 1);
        kk = __builtin_amdgcn_readfirstlane(kk);
        if (kk >= nwork) break;
        const int c = mylist[kk];
        const comp_row cr = rr[c];
        ws_flood_component(cell, cnt, offs, queue, d2, L, T, out, mk, cr, c, W, seeds_first, ties, plane, lane, rt);
    }
}

// ---- one launch for all LDS classes ---------------------------------------------------------------------------------
// Round 3.  The per-class launches are bound by LDS capacity x time: every workgroup reserves its class's MAXIMUM (15 KB
// for a component that needs 8), a class cannot finish before its longest component, and the classes compete for the same
// LDS whichever way they are launched (in order 1.41 ms, overlapped 1.06 ms per 48 planes -- the sum barely moves).
// Here ONE workgroup of 16 waves owns a CU's whole LDS and hands it out in 2,560-byte slots: a wave takes the next
// component of a global list (largest classes first), reserves exactly the slots ITS box and buckets need (compare-and-
// swap on a 64-bit mask of slots), floods with the same code as the class kernels, and returns the slots.
//   * two lists: "big" (classes L, M2) and "small" (M, S, XS).  Wave 0 of a workgroup serves the big list first and the
//     other waves wait for its first reservation, so a component that needs most of a CU starts at time zero instead of
//     behind fifteen small ones; when a list runs dry its waves move to the other one.
//   * a wave that finds the LDS full sleeps and retries; no wave waits while holding slots, so the workgroup always drains.
constexpr int PF_SLOT = 2560;
constexpr int PF_SLOTS_FULL = 63;            // data slots of a workgroup that owns a whole CU (the 64th: control words)
constexpr int PF_SLOTS_HALF = 31;            // ... that owns half a CU: two such workgroups, or other kernels, share it
constexpr int PF_WAVES = 16;                 // upper bound (launch bounds); the launch decides
constexpr int PF_NCLS = 5;                   // XS, S, M, M2, L (worklist slots 0..4)

// idx[c * (nplanes + 1) + plane] = components of class c in the planes before `plane` (the last entry: all of them);
// ctl[0] / ctl[1] = cursors of the big / small list (zeroed), ctl[2] / ctl[3] = their lengths
__global__ void __launch_bounds__(64) ws_flood_index_kernel(const int* __restrict__ wl_count, int* __restrict__ idx,
                                                            int* __restrict__ ctl, int nplanes) {
    __shared__ int tot[PF_NCLS];
    if (threadIdx.x < PF_NCLS) {
        const int c = threadIdx.x;
        int run = 0;
        for (int p = 0; p < nplanes; ++p) {
            idx[c * (nplanes + 1) + p] = run;
            run += wl_count[c * nplanes + p];
        }
        idx[c * (nplanes + 1) + nplanes] = run;
        tot[c] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = 0;
        ctl[2] = tot[4] + tot[3];           // big: L then M2
        ctl[3] = tot[2] + tot[1] + tot[0];  // small: M, S, XS
    }
}

__global__ void __launch_bounds__(PF_WAVES * 64) ws_flood_persist_kernel(
    const int* __restrict__ d2all, const int* __restrict__ Lall, const int* __restrict__ Tall, int* __restrict__ outall,
    const comp_row* __restrict__ rows, const int* __restrict__ wl, const int* __restrict__ idx, int* __restrict__ ctl,
    size_t row_stride, int H, int W, int seeds_first, int* __restrict__ ties, const int* __restrict__ mkall, int nplanes,
    int pf_slots, amt_runtabs rt) {
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    unsigned long long* slotmask = reinterpret_cast<unsigned long long*>(pf_smem + (size_t)pf_slots * PF_SLOT);
    int* first_done = reinterpret_cast<int*>(slotmask + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        *slotmask = ~0ull << pf_slots;  // bit s set = slot s taken; the slots from pf_slots on do not exist (control block)
        *first_done = 0;
    }
    __syncthreads();
    const size_t n = (size_t)H * W;
    const int nbig = ctl[2], nsmall = ctl[3];
    // class order inside a list and the list each wave starts with
    int list = wave == 0 ? 0 : 1;
    bool first = true;
    int dry = 0;  // lists found empty
    while (dry < 2) {
        int g = 0;
        if (lane == 0) g = atomicAdd(&ctl[list], 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= (list == 0 ? nbig : nsmall)) {
            if (first && wave == 0 && lane == 0) __hip_atomic_store(first_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = false;
            ++dry;
            list ^= 1;
            continue;
        }
        // item g of the list -> class, plane, component
        int cls, within = g;
        if (list == 0) {
            const int nL = idx[4 * (nplanes + 1) + nplanes];
            cls = within < nL ? 4 : 3;
            if (cls == 3) within -= nL;
        } else {
            const int nM = idx[2 * (nplanes + 1) + nplanes], nS = idx[1 * (nplanes + 1) + nplanes];
            cls = within < nM ? 2 : (within < nM + nS ? 1 : 0);
            within -= cls == 2 ? 0 : (cls == 1 ? nM : nM + nS);
        }
        const int* ci = idx + cls * (nplanes + 1);
        int lo = 0, hi = nplanes - 1;  // the last plane whose offset is <= within
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ci[mid] <= within) lo = mid; else hi = mid - 1;
        }
        const int plane = lo;
        const int c = wl[((size_t)cls * nplanes + plane) * row_stride + (within - ci[plane])];
        const comp_row cr = rows[(size_t)plane * row_stride + c];
        const int npx = (cr.x1 - cr.x0 + 3) * (cr.y1 - cr.y0 + 3);
        const int nb = cr.cmax + 1;
        const int npx2 = (npx + 1) & ~1, nb2 = (nb + 1) & ~1;  // keep every array 4-byte aligned
        // a pixel is queued once: with the component's pixel count (run-table statistics) the queue holds that many
        // entries instead of one per cell of the padded box -- LDS bytes x time is what bounds this launch
        const int nq2 = ((cr.npix > 0 ? cr.npix : npx) + 1) & ~1;
        const int need = (npx2 * 4 + nq2 * 2 + nb2 * 6 + PF_SLOT - 1) / PF_SLOT;  // slots
        // the other waves of the workgroup hold back until wave 0 has placed its first (big) component
        if (wave != 0 && first) {
            while (__hip_atomic_load(first_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
            first = false;
        }
        int pos = -1;
        if (lane == 0) {
            const unsigned long long want = need >= 64 ? ~0ull : ((1ull << need) - 1ull);
            while (true) {
                const unsigned long long m = __hip_atomic_load(slotmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned long long x = ~m;  // free slots; bit i of x survives iff slots i .. i + need - 1 are free
                for (int rem = need - 1, sh = 1; rem > 0; sh <<= 1) {
                    const int t = sh < rem ? sh : rem;
                    x &= x >> t;
                    rem -= t;
                }
                if (x) {
                    const int at = __ffsll((long long)x) - 1;
                    unsigned long long expect = m;
                    if (__hip_atomic_compare_exchange_strong(slotmask, &expect, m | (want << at), __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        pos = at;
                        break;
                    }
                } else {
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (first && wave == 0) {
            if (lane == 0) __hip_atomic_store(first_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = false;
        }
        unsigned* cell = reinterpret_cast<unsigned*>(pf_smem + (size_t)pos * PF_SLOT);
        unsigned* cnt = cell + npx2;
        unsigned short* offs = reinterpret_cast<unsigned short*>(cnt + nb2);
        unsigned short* queue = offs + nb2;
        ws_flood_component(cell, cnt, offs, queue, d2all + (size_t)plane * n, Lall + (size_t)plane * n, Tall + (size_t)plane * n,
                           outall + (size_t)plane * n, mkall + (size_t)plane * n, cr, c, W, seeds_first, ties, plane, lane, rt);
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            const unsigned long long want = need >= 64 ? ~0ull : ((1ull << need) - 1ull);
            __hip_atomic_fetch_and(slotmask, ~(want << pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // a workgroup whose wave 0 never placed anything must not leave its siblings waiting (they only wait while `first`)
    if (wave == 0 && lane == 0) __hip_atomic_store(first_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- HBM bucket-queue flood (components too large for an LDS tile) ---------------------------------
__global__ void __launch_bounds__(64) ws_flood_edt_kernel(const int* __restrict__ d2all, const uint8_t* __restrict__ maskall,
                                                          int* __restrict__ outall, int* __restrict__ nextall,
                                                          int* __restrict__ headall, int* __restrict__ tailall,
                                                          int* __restrict__ mlistall, const comp_row* __restrict__ rows,
                                                          const int* __restrict__ moffall, const int* __restrict__ boffall,
                                                          const int* __restrict__ ncomp, int* __restrict__ counters,
                                                          size_t row_stride, int H, int W, size_t n, size_t bstride,
                                                          int seeds_first, int* __restrict__ ties) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * n;
    const size_t cb = (size_t)plane * row_stride;
    const int* d2 = d2all + base;
    const uint8_t* mask = maskall + base;
    int* out = outall + base;
    int* next = nextall + base;
    const int nc = ncomp[plane];
    const int nbo[4] = {-W, -1, +1, +W};
    while (true) {
        const int c = atomicAdd(&counters[plane], 1);
        if (c >= nc) break;
        if (rows[cb + c].cls != CLS_G) continue;
        const int nm = rows[cb + c].mcnt;
        int* ml = mlistall + base + moffall[cb + c];
        int* hd = headall + (size_t)plane * bstride + boffall[cb + c];
        int* tl = tailall + (size_t)plane * bstride + boffall[cb + c];
        int cur = -1;
        lane_sort(ml, nm);

        auto push = [&](int p, int b) {
            next[p] = -1;
            if (hd[b] < 0) {
                hd[b] = p;
            } else {
                next[tl[b]] = p;
            }
            tl[b] = p;
            if (b > cur) cur = b;
        };
        auto spread = [&](int p) {
            const int lab = out[p];
            const int py = p / W, px = p - py * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 0 && py == 0) continue;
                if (k == 1 && px == 0) continue;
                if (k == 2 && px == W - 1) continue;
                if (k == 3 && py == H - 1) continue;
                const int q = p + nbo[k];
                if (mask[q] && out[q] == 0) {
                    out[q] = lab;
                    int b = d2[q];
                    push(q, b < 0 ? 0 : b);
                }
            }
        };

        if (seeds_first) {
            for (int i = 0; i < nm; ++i) spread(ml[i]);
        } else {
            for (int i = 0; i < nm; ++i) {
                int b = d2[ml[i]];
                b = b < 0 ? 0 : b;
                if (hd[b] >= 0) ties[plane] = 1;  // second marker of equal value in this component
                push(ml[i], b);
            }
        }
        while (true) {
            while (cur >= 0 && hd[cur] < 0) --cur;
            if (cur < 0) break;
            const int p = hd[cur];
            hd[cur] = next[p];
            spread(p);
        }
    }
}

// ---- binary-heap flood (arbitrary float64 relief) --------------------------------------------------
struct hp_elem {
    double value;
    int age;
    int index;
};

__device__ __forceinline__ bool hp_less(const hp_elem& a, const hp_elem& b) {
    if (a.value != b.value) return a.value < b.value;
    if (a.age != b.age) return a.age < b.age;
    return a.index < b.index;  // only age-0 markers can tie on (value, age): raster order
}

__global__ void __launch_bounds__(64) ws_flood_heap_kernel(const double* __restrict__ relall,
                                                           const uint8_t* __restrict__ maskall, int* __restrict__ outall,
                                                           hp_elem* __restrict__ heapall, int* __restrict__ mlistall,
                                                           const comp_row* __restrict__ rows,
                                                           const int* __restrict__ moffall, const int* __restrict__ boffall,
                                                           const int* __restrict__ ncomp, int* __restrict__ counters,
                                                           size_t row_stride, int H, int W, size_t n, size_t hstride,
                                                           int* __restrict__ ties) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * n;
    const size_t cb = (size_t)plane * row_stride;
    const double* rel = relall + base;
    const uint8_t* mask = maskall + base;
    int* out = outall + base;
    const int nc = ncomp[plane];
    const int nbo[4] = {-W, -1, +1, +W};
    while (true) {
        const int c = atomicAdd(&counters[plane], 1);
        if (c >= nc) break;
        if (rows[cb + c].cls != CLS_G) continue;
        const int nm = rows[cb + c].mcnt;
        int* ml = mlistall + base + moffall[cb + c];
        hp_elem* hp = heapall + (size_t)plane * hstride + boffall[cb + c];
        int items = 0;
        int age = 0;
        lane_sort(ml, nm);
        auto hpush = [&](const hp_elem& e) {
            int child = items++;
            while (child > 0) {
                int parent = (child - 1) >> 1;
                hp_elem pe = hp[parent];
                if (hp_less(e, pe)) {
                    hp[child] = pe;
                    child = parent;
                } else
                    break;
            }
            hp[child] = e;
        };
        auto hpop = [&]() {
            hp_elem top = hp[0];
            hp_elem last = hp[--items];
            int parent = 0;
            while (true) {
                int child = 2 * parent + 1;
                if (child >= items) break;
                hp_elem ce = hp[child];
                if (child + 1 < items) {
                    hp_elem c2 = hp[child + 1];
                    if (hp_less(c2, ce)) {
                        ce = c2;
                        ++child;
                    }
                }
                if (hp_less(ce, last)) {
                    hp[parent] = ce;
                    parent = child;
                } else
                    break;
            }
            if (items > 0) hp[parent] = last;
            return top;
        };
        for (int i = 0; i < nm; ++i) {
            hp_elem e;
            e.value = rel[ml[i]];
            e.age = 0;
            e.index = ml[i];
            hpush(e);
        }
        bool have0 = false;
        double last0 = 0.0;
        while (items > 0) {
            hp_elem e = hpop();
            const int p = e.index;
            if (e.age == 0) {  // age-0 entries pop in non-decreasing value order: equal values are adjacent among them
                if (have0 && e.value == last0) ties[plane] = 1;
                have0 = true;
                last0 = e.value;
            }
            const int lab = out[p];
            const int py = p / W, px = p - py * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 0 && py == 0) continue;
                if (k == 1 && px == 0) continue;
                if (k == 2 && px == W - 1) continue;
                if (k == 3 && py == H - 1) continue;
                const int q = p + nbo[k];
                if (mask[q] && out[q] == 0) {
                    out[q] = lab;
                    hp_elem ne;
                    ne.value = rel[q];
                    ne.age = ++age;
                    ne.index = q;
                    hpush(ne);
                }
            }
        }
    }
}

// ---- sequential emulation of scikit-image's flood (one heap per plane) --------------------------------------
// markers are pushed in raster order with age 0; pop the smallest (value, age) -- STRICT comparisons,
// so equal keys are ordered by the heap's own moves: push = append, then swap with the parent (child + 1) / 2 - 1
// while strictly smaller; pop = move the last element to the root, then repeatedly take the right child if it is
// strictly smaller than the left, and swap with the parent if strictly smaller.  Labels are assigned at push time;
// neighbours in the order N, W, E, S (connectivity 1) or N, E, W, S, NW, NE, SW, SE (connectivity 2: the order
// scikit-image 0.18.3's offset sort produces; version sensitive).  Only planes whose `run` flag is
// set are touched.  One lane does the heap; the wave scans for marker pixels 64 at a time.
__global__ void __launch_bounds__(256) ws_global_init_kernel(const int* __restrict__ markers,
                                                             const uint8_t* __restrict__ mask, int* __restrict__ out,
                                                             const int* __restrict__ run, size_t n) {
    if (run && !run[blockIdx.y]) return;
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[base + i] = mask[base + i] ? markers[base + i] : 0;  // markers.astype(int32) * mask
}

// Round 3: the whole wave works on the ONE heap.  The heap's mechanics are scikit-image's, element for element; what
// changed is how many memory round trips an operation takes:
//   * an element is 16 bytes {key, age, index} with key = an unsigned code that orders like the relief (0x7fffffff - d2,
//     or the order-preserving image of the float64), so "smaller" is two integer compares;
//   * levels 0..12 of the heap (8,191 elements, 128 KB) live in LDS, deeper levels in HBM;
//   * push: the ancestors of the new slot are known up front -- lane j loads ancestor j, one ballot tells how far the
//     element rises, and the lanes shift the overtaken ancestors down in one round of stores (one round trip instead of
//     one per level);
//   * pop: the path down is data dependent, so 62 lanes fetch the whole 5-level subtree under the current position
//     (1-2 round trips for the LDS levels + one for the HBM levels instead of 2 loads per level), the wave walks it with
//     cross-lane reads, and the lanes on the path move their elements up in one round of stores;
//   * the popped pixel's neighbours (mask, label, relief) are requested BEFORE the sift-down and arrive under it.
// One wave per plane; round 2's one-lane version took 2.1 s per tied 2048^2 plane (4.9 us per pixel).
struct gh_elem {
    unsigned long long key;
    unsigned age;
    unsigned idx;
};
constexpr int GH_LDS_N = 8191;  // levels 0..12

template <bool USE_D2>
__global__ void __launch_bounds__(64) ws_global_kernel(const void* __restrict__ reliefall,
                                                       const uint8_t* __restrict__ maskall, int* __restrict__ outall,
                                                       hp_elem* __restrict__ heapall, const int* __restrict__ run,
                                                       int H, int W, size_t hstride, int conn) {
    extern __shared__ __attribute__((aligned(16))) char gh_smem[];
    gh_elem* sh = reinterpret_cast<gh_elem*>(gh_smem);
    const int plane = blockIdx.x;
    if (run && !run[plane]) return;
    const size_t n = (size_t)H * W;
    const uint8_t* mask = maskall + (size_t)plane * n;
    int* out = outall + (size_t)plane * n;
    gh_elem* gh = reinterpret_cast<gh_elem*>(heapall + (size_t)plane * hstride);  // same 16 bytes per slot
    const double* rel = USE_D2 ? nullptr : (const double*)reliefall + (size_t)plane * n;
    const int* d2 = USE_D2 ? (const int*)reliefall + (size_t)plane * n : nullptr;
    const int lane = threadIdx.x;
    constexpr unsigned long long INF = ~0ull;
    auto key_of_d2 = [](int d) -> unsigned long long { return (unsigned long long)(0x7fffffffu - (unsigned)(d < 0 ? 0 : d)); };
    auto key_of_f64 = [](double v) -> unsigned long long { return amt_f64_key(v + 0.0); };  // -0.0 orders like +0.0
    auto ld = [&](int i) -> gh_elem { return i < GH_LDS_N ? sh[i] : gh[i]; };
    auto st = [&](int i, const gh_elem& e) {
        if (i < GH_LDS_N) sh[i] = e; else gh[i] = e;
    };
    // integer relief: the 31-bit code and the age are ONE 64-bit key (code << 32 | age), "smaller" is one compare
    auto less = [](unsigned long long ka, unsigned aa, unsigned long long kb, unsigned ab) -> bool {
        return USE_D2 ? ka < kb : (ka < kb || (ka == kb && aa < ab));
    };
    auto rl64 = [](unsigned long long v, int l) -> unsigned long long {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
        return ((unsigned long long)hi << 32) | lo;
    };
    int items = 0;  // uniform
    gh_elem tail;            // the element in slot items - 1 when tail_valid (uniform)
    tail.key = 0;
    tail.age = 0;
    tail.idx = 0;
    bool tail_valid = false;
    // ---- push: executed by the whole wave, e is uniform ----
    auto push = [&](unsigned long long ekey, unsigned eage, unsigned eidx) {
        const int c = items++;
        const int depth = 31 - __clz(c + 1);  // number of ancestors of slot c
        gh_elem a;
        a.key = 0;
        a.age = 0;
        a.idx = 0;
        const bool anc = lane >= 1 && lane <= depth;
        // (the ancestors below the LDS levels were written by this CU a moment ago and come from its L1: keeping them
        // in registers from one push to the next, or requesting them ahead, was measured and bought nothing)
        if (anc) a = ld(((c + 1) >> lane) - 1);
        const bool rises = anc && less(ekey, eage, a.key, a.age);
        const unsigned long long m = __ballot(rises) >> 1;  // bit j-1: the element overtakes ancestor j
        const int up = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;  // consecutive overtaken ancestors from the bottom
        if (anc && lane <= up) st(((c + 1) >> (lane - 1)) - 1, a);  // ancestor j moves down into slot of j - 1
        if (lane == 0) {
            gh_elem en;
            en.key = ekey;
            en.age = eage;
            en.idx = eidx;
            st(((c + 1) >> up) - 1, en);
        }
        // what now sits in the LAST slot (c): the new element, or its parent if it rose -- the next pop moves exactly
        // this element to the root and need not fetch it from HBM
        tail_valid = true;
        if (up == 0) {
            tail.key = ekey;
            tail.age = eage;
            tail.idx = eidx;
        } else {
            tail.key = rl64(a.key, 1);
            tail.age = (unsigned)__builtin_amdgcn_readlane((int)a.age, 1);
            tail.idx = (unsigned)__builtin_amdgcn_readlane((int)a.idx, 1);
        }
    };
    // markers in raster order (age 0)
    for (size_t i0 = 0; i0 < n; i0 += 64) {
        const size_t i = i0 + lane;
        const int v = i < n ? out[i] : 0;
        unsigned long long mk = __ballot(v != 0);
        if (!mk) continue;
        unsigned long long kv = 0;
        if (v != 0) kv = USE_D2 ? key_of_d2(d2[i]) << 32 : key_of_f64(rel[i]);
        // heap entries carry the pixel as (y << 16 | x): the flood needs both coordinates of every popped pixel, and a
        // division per pop is ~40 dependent instructions of a wave that runs alone
        const size_t iy = i0 / (size_t)W;
        int ly = (int)iy, lx = (int)(i0 - iy * (size_t)W) + lane;
        while (lx >= W) {
            lx -= W;
            ++ly;
        }
        const unsigned pk = ((unsigned)ly << 16) | (unsigned)lx;
        while (mk) {
            const int b = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            // a wave's loads see its own earlier stores: no explicit wait
            push(rl64(kv, b), 0u, (unsigned)__builtin_amdgcn_readlane((int)pk, b));
        }
    }
    const int nnb = conn == 1 ? 4 : 8;
    // neighbour order: N, W, E, S (connectivity 1); N, E, W, S, NW, NE, SW, SE (connectivity 2, scikit-image 0.18.3)
    int dy = 0, dx = 0;
    if (conn == 1) {
        dy = lane == 0 ? -1 : lane == 3 ? 1 : 0;
        dx = lane == 1 ? -1 : lane == 2 ? 1 : 0;
    } else {
        const int dys[8] = {-1, 0, 0, 1, -1, -1, 1, 1}, dxs[8] = {0, 1, -1, 0, -1, 1, -1, 1};
        dy = lane < 8 ? dys[lane & 7] : 0;
        dx = lane < 8 ? dxs[lane & 7] : 0;
    }
    // relative position of this lane's node in a 5-level subtree: r = lane + 1 (1..62), level lv, offset o
    const int r = lane + 1;
    const int lv = 31 - __clz(r + 1);
    const int o = r + 1 - (1 << lv);
    // The preference bits (RM below: bit 2 a + 1 = "relative node a prefers its right child", a = 0: the subtree's root)
    // that put THIS lane's node on the min-child path: every ancestor must prefer the child on the way down to it.
    unsigned long long need_r = 0, need_l = 0;
    {
        int c = r;
        while (c > 0 && lane < 62) {
            const int par = (c - 1) >> 1;
            const unsigned long long bit = 1ull << (2 * par + 1);
            if (c == 2 * par + 2) need_r |= bit; else need_l |= bit;
            c = par;
        }
    }
    // slot of this lane's node's parent: ((pos + 1) << par_lv) - 1 + par_off (par_lv = 0, par_off = 0: pos itself)
    const int prel = (r - 1) >> 1;
    const int par_lv = prel > 0 ? 31 - __clz(prel + 1) : 0;
    const int par_off = prel > 0 ? prel + 1 - (1 << par_lv) : 0;
    unsigned age = 0;
#ifdef WS_STATS
    unsigned long long g_pop = 0, g_push = 0, g_rl = 0, g_rg = 0, g_npush = 0, g_npop = 0, g_nr = 0;
    const unsigned long long g_t0 = __builtin_readcyclecounter();
#define GCLK() __builtin_readcyclecounter()
#endif
    while (items > 0) {
#ifdef WS_STATS
        const unsigned long long g_a = GCLK();
#endif
        const gh_elem top = sh[0];
        const int py = (int)(top.idx >> 16), px = (int)(top.idx & 0xffffu);
        const int p = py * W + px;
        // neighbour data, requested now, consumed after the sift-down
        const int qy = py + dy, qx = px + dx;
        const bool inside = lane < nnb && qy >= 0 && qy < H && qx >= 0 && qx < W;
        const int q = inside ? qy * W + qx : p;
        const unsigned qpk = ((unsigned)qy << 16) | (unsigned)qx;
        const uint8_t qm = mask[q];
        const int qo = out[q];
        unsigned long long qk = 0;
        if (USE_D2) qk = key_of_d2(d2[q]); else qk = key_of_f64(rel[q]);
        const int lab = out[p];
        // ---- pop ----
        --items;
        if (items == 0) tail_valid = false;
        if (items > 0) {
            const gh_elem last = tail_valid ? tail : ld(items);
            tail_valid = false;
            int pos = 0;
            while (true) {
#ifdef WS_STATS
                const unsigned long long g_r0 = GCLK();
                const bool g_glob = (((pos + 1) << 5) - 1) >= GH_LDS_N;
#endif
                // the 62 nodes of the five levels under pos
                const int ab = ((pos + 1) << lv) - 1 + o;  // slots stay far below 2^31 (a plane has < 2^26 pixels)
                const bool valid = lane < 62 && ab < items;
                gh_elem e;
                e.key = INF;
                e.age = 0;
                e.idx = 0;
                if (valid) e = ld(ab);
                // Which child every node of the subtree prefers, and which nodes are smaller than `last`, for all 62 nodes
                // AT ONCE (two ballots); the path is then bit arithmetic on the masks.  (Walking with cross-lane reads cost
                // ~30 dependent vector instructions per level, 17 levels per pop: most of the 3.7 us per pixel.)
                // Lane L holds relative node L + 1, so siblings sit in lanes (2 i, 2 i + 1): the RIGHT child (odd lane)
                // compares itself with its left neighbour -- one lane shift, no LDS crossbar.  RM bit 2 a + 1 = "node a
                // prefers its right child".  A missing node carries the key INF: never preferred, never smaller than `last`.
                const unsigned long long kleft = ((unsigned long long)(unsigned)amt_lane_left((int)(unsigned)(e.key >> 32)) << 32) |
                                                 (unsigned)amt_lane_left((int)(unsigned)e.key);
                const unsigned aleft = (unsigned)amt_lane_left((int)e.age);
                const unsigned long long RM = __ballot((lane & 1) && lane < 62 && less(e.key, e.age, kleft, aleft));
                const unsigned long long LT = __ballot(lane < 62 && less(e.key, e.age, last.key, last.age));
                // the min-child path through the five levels, all at once: a node is on it iff all its ancestors prefer
                // the child that leads to it; the path's elements move up as far as they are smaller than `last` (one
                // node per level, and lanes are numbered level by level: "above the first that is not" = lower lanes)
                const unsigned long long onpath = __ballot(lane < 62 && (RM & need_r) == need_r && (RM & need_l) == 0ull);
                const unsigned long long blocked = onpath & ~LT;
                const bool placed = blocked != 0ull;  // a child is missing or not smaller than `last`: it settles here
                const unsigned long long path = placed ? onpath & ((blocked & (0ull - blocked)) - 1ull) : onpath;
                const int cur = path ? 64 - __clzll((long long)path) : 0;  // relative index of the node `last` would sit in
                // lanes on the path hand their element to the parent slot
                if ((path >> lane) & 1ull) st(((pos + 1) << par_lv) - 1 + par_off, e);
                int cab = pos;
                if (cur > 0) {
                    const int clv = 31 - __clz(cur + 1);
                    cab = ((pos + 1) << clv) - 1 + (cur + 1 - (1 << clv));
                }
#ifdef WS_STATS
                {
                    const unsigned long long d = GCLK() - g_r0;
                    if (g_glob) g_rg += d; else g_rl += d;
                    ++g_nr;
                }
#endif
                if (placed) {
                    if (lane == 0) st(cab, last);
                    break;
                }
                pos = cab;
            }
        }
#ifdef WS_STATS
        g_pop += GCLK() - g_a;
        ++g_npop;
        const unsigned long long g_b = GCLK();
#endif
        // ---- spread: unlabelled masked neighbours take the label at push time, in neighbour order ----
        unsigned long long claim = __ballot(inside && qm != 0 && qo == 0);
        while (claim) {
            const int k = __ffsll((long long)claim) - 1;
            claim &= claim - 1;
            const int qq = __builtin_amdgcn_readlane(q, k);
            if (lane == 0) out[qq] = lab;
            ++age;
            push(USE_D2 ? (rl64(qk, k) << 32) | age : rl64(qk, k), age, (unsigned)__builtin_amdgcn_readlane((int)qpk, k));
#ifdef WS_STATS
            ++g_npush;
#endif
        }
#ifdef WS_STATS
        g_push += GCLK() - g_b;
#endif
    }
#ifdef WS_STATS
    if (lane == 0 && plane == 0) {
        ws_dbg[0] = g_npop, ws_dbg[1] = g_pop, ws_dbg[2] = g_npush, ws_dbg[3] = g_push;
        ws_dbg[4] = g_nr, ws_dbg[5] = g_rg, ws_dbg[6] = g_rl, ws_dbg[7] = GCLK() - g_t0;
    }
#undef GCLK
#endif
}

__global__ void ws_set_flags_kernel(int* f, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = v;
}

// ---- fused tail: clear_border + relabel_sequential on the flood's result (R/masks.py:56,65) ---------------------
// A watershed label is ONE 4-connected region, so it is removed iff one of its pixels lies on the frame.  The label of
// a pixel: F of its tile root (> 0: the single label of its component; 0: flooded -> the flood wrote it into ws;
// < 0: component without markers).  P[l] = 2 marks frame-touching labels.
__device__ __forceinline__ int ws_pixel_label(int r, const int* __restrict__ F, const int* __restrict__ ws, size_t base,
                                              size_t i) {
    if (r < 0) return 0;
    const int f = F[base + r];
    return f > 0 ? f : (f == 0 ? ws[base + i] : 0);
}

__global__ void __launch_bounds__(256) ws_frame_mark_kernel(const int* __restrict__ L, const int* __restrict__ F,
                                                            const int* __restrict__ ws, int* __restrict__ present, int H,
                                                            int W, int max_label, amt_runtabs rt) {
    const size_t n = (size_t)H * W;
    const size_t base = (size_t)blockIdx.y * n;
    int* P = present + (size_t)blockIdx.y * (max_label + 1);
    const int perim = 2 * W + 2 * H;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < perim; k += gridDim.x * 256) {
        int y, x;
        if (k < W) {
            y = 0;
            x = k;
        } else if (k < 2 * W) {
            y = H - 1;
            x = k - W;
        } else if (k < 2 * W + H) {
            y = k - 2 * W;
            x = 0;
        } else {
            y = k - 2 * W - H;
            x = W - 1;
        }
        const size_t i = (size_t)y * W + x;
        int r;
        if (rt.tbits) {
            const long long ri = amt_rt_px_run(rt, blockIdx.y, y, x);
            r = ri < 0 ? -1 : amt_rt_run_root(rt, ri, W);
        } else {
            r = L[base + i];
        }
        const int v = ws_pixel_label(r, F, ws, base, i);
        if (v > 0 && v <= max_label) P[v] = 2;
    }
}

// labels_out = map[label of the pixel]: the ONE full-plane pass of the fused path (16-byte loads of the parent plane,
// 16-byte stores); the flood's plane is read only at pixels of flooded components
__global__ void __launch_bounds__(256) ws_final_kernel(const int* __restrict__ L, const int* __restrict__ F,
                                                       const int* __restrict__ ws, const int* __restrict__ map,
                                                       int* __restrict__ out, size_t n, int max_label) {
    const size_t base = (size_t)blockIdx.y * n;
    const int* M = map + (size_t)blockIdx.y * (max_label + 1);
    const unsigned ml = (unsigned)max_label;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        if (i0 + 3 < n && ((base + i0) & 3) == 0) {
            const int4 r = *reinterpret_cast<const int4*>(L + base + i0);
            int4 o = make_int4(0, 0, 0, 0);
            if (r.x >= 0 || r.y >= 0 || r.z >= 0 || r.w >= 0) {
                const int fx = r.x >= 0 ? F[base + r.x] : -1, fy = r.y >= 0 ? F[base + r.y] : -1;
                const int fz = r.z >= 0 ? F[base + r.z] : -1, fw = r.w >= 0 ? F[base + r.w] : -1;
                int4 w = make_int4(0, 0, 0, 0);
                if (fx == 0 || fy == 0 || fz == 0 || fw == 0) w = *reinterpret_cast<const int4*>(ws + base + i0);
                const int vx = fx > 0 ? fx : (fx == 0 ? w.x : 0), vy = fy > 0 ? fy : (fy == 0 ? w.y : 0);
                const int vz = fz > 0 ? fz : (fz == 0 ? w.z : 0), vw = fw > 0 ? fw : (fw == 0 ? w.w : 0);
                o.x = (unsigned)(vx - 1) < ml ? M[vx] : 0;
                o.y = (unsigned)(vy - 1) < ml ? M[vy] : 0;
                o.z = (unsigned)(vz - 1) < ml ? M[vz] : 0;
                o.w = (unsigned)(vw - 1) < ml ? M[vw] : 0;
            }
            *reinterpret_cast<int4*>(out + base + i0) = o;
        } else {
            for (size_t i = i0; i < n && i < i0 + 4; ++i) {
                const int v = ws_pixel_label(L[base + i], F, ws, base, i);
                out[base + i] = (unsigned)(v - 1) < ml ? M[v] : 0;
            }
        }
    }
}

// presence_fill + ws_frame_mark + drop_flagged + presence_scan (amt_label.hip) for one plane per workgroup:
// P[l] = new label of l (0 = absent or touching the frame), count[plane] = number of survivors
__global__ void __launch_bounds__(1024) ws_label_map_kernel(const int* __restrict__ L, const int* __restrict__ F,
                                                            const int* __restrict__ ws, int* __restrict__ present,
                                                            const int* __restrict__ nlabels, int* __restrict__ count_dev,
                                                            int H, int W, int max_label, amt_runtabs rt) {
    __shared__ int s[1024];
    __shared__ int carry;
    const int plane = blockIdx.x;
    const size_t n = (size_t)H * W;
    const size_t base = (size_t)plane * n;
    int* P = present + (size_t)plane * (max_label + 1);
    const int k = nlabels[plane] < max_label ? nlabels[plane] : max_label;
    for (int l = threadIdx.x; l <= max_label; l += 1024) P[l] = (l >= 1 && l <= k) ? 1 : 0;
    __syncthreads();
    const int perim = 2 * W + 2 * H;
    for (int q = threadIdx.x; q < perim; q += 1024) {
        int y, x;
        if (q < W) {
            y = 0;
            x = q;
        } else if (q < 2 * W) {
            y = H - 1;
            x = q - W;
        } else if (q < 2 * W + H) {
            y = q - 2 * W;
            x = 0;
        } else {
            y = q - 2 * W - H;
            x = W - 1;
        }
        const size_t i = (size_t)y * W + x;
        int r;
        if (rt.tbits) {
            const long long ri = amt_rt_px_run(rt, plane, y, x);
            r = ri < 0 ? -1 : amt_rt_run_root(rt, ri, W);
        } else {
            r = L[base + i];
        }
        const int v = ws_pixel_label(r, F, ws, base, i);
        if (v > 0 && v <= max_label) P[v] = 0;  // touches the frame: dropped
    }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int start = 1; start <= max_label; start += 1024) {
        const int i = start + threadIdx.x;
        const int v = i <= max_label ? P[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = s[threadIdx.x];
        const int c = carry;
        if (i <= max_label) P[i] = v ? c + incl : 0;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) count_dev[plane] = carry;
}

// ws_final_kernel from the run tables: a wave per tile turns (row words, run table, the final label of every tile root,
// the label map) into final labels -- the parent plane is not read; the flood's plane only where a flooded component has
// pixels.  ws_final_map_kernel first replaces F at every listed tile root by what its pixels are to receive (the mapped
// label of a single-label component, -1 = flooded: look in the flood's plane, 0 = background), so that a run costs ONE
// gather here.  The write-out keeps to the rule of this file: all loads of eight rows are in flight (the flood plane's,
// predicated, and then the map's) before the first store.
__global__ void __launch_bounds__(256) ws_final_map_kernel(int* __restrict__ Fall, const int* __restrict__ map,
                                                           const int* __restrict__ rootlist, const int* __restrict__ nroots,
                                                           size_t cap, size_t n, int max_label) {
    const int plane = blockIdx.z, shard = plane * gridDim.y + blockIdx.y;
    int* F = Fall + (size_t)plane * n;
    const int* M = map + (size_t)plane * (max_label + 1);
    const int cnt = nroots[shard] < (int)cap ? nroots[shard] : (int)cap;
    const int* lst = rootlist + (size_t)shard * cap;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int t = lst[k], f = F[t];
        F[t] = f > 0 ? ((unsigned)(f - 1) < (unsigned)max_label ? M[f] : 0) : (f == 0 ? -1 : 0);
    }
}

__global__ void __launch_bounds__(256) ws_final_runs_kernel(const unsigned long long* __restrict__ tbits,
                                                            const unsigned short* __restrict__ rtab,
                                                            const int* __restrict__ nruns, const int* __restrict__ Fall,
                                                            const int* __restrict__ wsall, const int* __restrict__ map,
                                                            int* __restrict__ outall, int H, int W, int segs, int trows,
                                                            int ntiles, int max_label) {
    __shared__ int lab_s[4][SR_CAP];
    __shared__ unsigned long long bits_s[4][64];
    __shared__ int off_s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wv;
    if (t >= ntiles) return;  // whole wave
    const int bx = t % segs, ty = (t / segs) % trows, plane = t / (segs * trows);
    const size_t n = (size_t)H * W;
    const int* F = Fall + (size_t)plane * n;
    const int* ws = wsall + (size_t)plane * n;
    const int* M = map + (size_t)plane * (max_label + 1);
    int* out = outall + (size_t)plane * n;
    const unsigned ml = (unsigned)max_label;
    const int x0 = bx * 64, ty0 = ty * 64;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = x0 + c4;
    const int nr = nruns[t];
    if (nr == 0) {  // uniform: nothing but background
        if (xg < W) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int y = ty0 + rsub + 4 * j;
                if (y < H) *reinterpret_cast<int4*>(out + (size_t)y * W + xg) = make_int4(0, 0, 0, 0);
            }
        }
        return;
    }
    const unsigned long long w = tbits[(size_t)t * 64 + lane];
    const int cnt = __popcll(w & ~(w << 1));
    bits_s[wv][lane] = w;
    off_s[wv][lane] = ccl_wave_incl_scan(cnt, lane) - cnt;
    const bool fast = nr <= SR_CAP;  // otherwise (noise): a gather per pixel
    if (fast)
        for (int k = lane; k < nr; k += 64) lab_s[wv][k] = F[ccl_rt_root(rtab, (size_t)t, k, ty, bx, W)];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (xg >= W) return;
#ifndef AMT_WS_FQ
#define AMT_WS_FQ 4
#endif
    constexpr int FQ = AMT_WS_FQ;  // rows per lane whose loads are in flight together
    const int* lb = lab_s[wv];
    auto lab = [&](int k) -> int { return fast ? lb[k] : F[ccl_rt_root(rtab, (size_t)t, k, ty, bx, W)]; };
#pragma unroll
    for (int half = 0; half < 16 / FQ; ++half) {
        int4 o[FQ], f[FQ];
        bool need[FQ];
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int row = rsub + 4 * (half * FQ + q);
            const unsigned long long ww = bits_s[wv][row];
            const unsigned long long hw = ww & ~(ww << 1);
            const unsigned nib = (unsigned)(ww >> c4) & 15u, hnib = (unsigned)(hw >> c4) & 15u;
            o[q] = make_int4(0, 0, 0, 0);
            if (nib) {
                // one divergent block, selects inside (a block per pixel costs scalar issue slots: exec save, branch,
                // restore); a background pixel reads the run left of it, or run 0 of the tile when there is none
                const int k0 = off_s[wv][row] + __popcll(hw & ((1ull << c4) - 1ull)) - 1;
                const int ka = k0 + (int)(hnib & 1u), kb = k0 + __popc(hnib & 3u), kc = k0 + __popc(hnib & 7u), kd = k0 + __popc(hnib);
                const int la = lab(ka < 0 ? 0 : ka), lb2 = lab(kb < 0 ? 0 : kb), lc = lab(kc < 0 ? 0 : kc), ld = lab(kd < 0 ? 0 : kd);
                o[q].x = (nib & 1u) ? la : 0;
                o[q].y = (nib & 2u) ? lb2 : 0;
                o[q].z = (nib & 4u) ? lc : 0;
                o[q].w = (nib & 8u) ? ld : 0;
            }
            need[q] = (o[q].x | o[q].y | o[q].z | o[q].w) < 0;  // pixels of a flooded component: labels in the flood's plane
        }
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int y = ty0 + rsub + 4 * (half * FQ + q);
            f[q] = make_int4(0, 0, 0, 0);
            if (need[q]) f[q] = *reinterpret_cast<const int4*>(ws + (size_t)y * W + xg);
        }
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            if (need[q]) {
                if (o[q].x < 0) o[q].x = (unsigned)(f[q].x - 1) < ml ? M[f[q].x] : 0;
                if (o[q].y < 0) o[q].y = (unsigned)(f[q].y - 1) < ml ? M[f[q].y] : 0;
                if (o[q].z < 0) o[q].z = (unsigned)(f[q].z - 1) < ml ? M[f[q].z] : 0;
                if (o[q].w < 0) o[q].w = (unsigned)(f[q].w - 1) < ml ? M[f[q].w] : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int y = ty0 + rsub + 4 * (half * FQ + q);
            if (y < H) *reinterpret_cast<int4*>(out + (size_t)y * W + xg) = o[q];
        }
    }
}

// AMT_WS_ANYORDER=0 keeps the flood classes of a context without auxiliary streams strictly in order (A/B switch)
static bool ws_anyorder() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_ANYORDER");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// AMT_WS_PERSIST: 0 = the per-class flood launches of round 2, 1 = one persistent launch, a workgroup per CU with all
// its LDS, 2 = two workgroups per CU with half the LDS each (A/B switch; identical results)
static int ws_persist() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_PERSIST");
        v = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    }
    return v;
}
// AMT_WS_PF_SLOTS: 2,560-byte LDS slots of a mode-1 workgroup (8..63; fewer leave LDS on the CU for other kernels)
static int ws_pf_slots() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_PF_SLOTS");
        v = e ? atoi(e) : PF_SLOTS_FULL;
        if (v < 8 || v > PF_SLOTS_FULL) v = PF_SLOTS_FULL;
    }
    return v;
}
static int ws_pf_waves() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_PF_WAVES");
        v = e ? atoi(e) : 10;
        if (v < 1 || v > PF_WAVES) v = 10;
    }
    return v;
}

// AMT_WS_RUNS=0: the fused chain reads the parent plane in its statistics and final passes (A/B switch; same results)
static bool ws_runs_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_RUNS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// AMT_WS_LDS_PAD=bytes: extra dynamic LDS per flood workgroup (occupancy experiments only)
static size_t ws_lds_pad() {
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_WS_LDS_PAD");
        v = e ? atol(e) : 0;
        if (v < 0) v = 0;
    }
    return (size_t)v;
}

static int watershed_common(amt_ctx* ctx, const void* relief, bool use_d2, const int32_t* markers,
                            const uint8_t* mask, int32_t* out, int nplanes, int H, int W, int seeds_first,
                            int connectivity, int tie_policy, int32_t* ties_dev, int32_t* fused_labels = nullptr,
                            int32_t* fused_count = nullptr, const int32_t* nlabels_dev = nullptr, int max_label = 0,
                            const int32_t* mk_list = nullptr, const int32_t* mk_count = nullptr, int mk_cap = 0) {
    // fused_labels != nullptr: `out` is only the flood's scratch plane; the result is clear_border + relabel_sequential
    // of the watershed, written to fused_labels / fused_count (connectivity 1 only)
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(relief && markers && mask && out && nplanes >= 0 && H > 0 && W > 0, "watershed: bad arguments");
    AMT_REQUIRE((size_t)H * W < 0x3fffffffull, "watershed: plane too large");
    AMT_REQUIRE((const void*)markers != (const void*)out, "watershed: markers and out must not alias");
    AMT_REQUIRE(connectivity == 1 || connectivity == 2, "watershed: connectivity must be 1 or 2, got %d", connectivity);
    AMT_REQUIRE(tie_policy == AMT_WS_TIES_EXACT || tie_policy == AMT_WS_TIES_RASTER || tie_policy == AMT_WS_TIES_REPORT,
                "watershed: unknown tie policy %d", tie_policy);
    AMT_REQUIRE(connectivity == 1 || !seeds_first, "watershed: seeds_first is defined for connectivity 1 only");
    AMT_REQUIRE(connectivity == 1 || tie_policy == AMT_WS_TIES_EXACT,
                "watershed: connectivity 2 exists only as the sequential single-heap emulation (AMT_WS_TIES_EXACT)");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const size_t np = (size_t)nplanes * n;
    // seeds_first cannot tie (marker pixels are spread first, in raster order, before anything is queued by value)
    const bool exact = tie_policy == AMT_WS_TIES_EXACT && !seeds_first;
    // the single-heap emulation packs a pixel as (y << 16 | x) and keeps heap slots in 32-bit arithmetic
    AMT_REQUIRE(!(exact || connectivity == 2) || (H <= 65535 && W <= 65535 && n <= ((size_t)1 << 26)),
                "watershed: exact tie handling / connectivity 2 support planes of at most 2^26 pixels, got %d x %d", H, W);
    if (connectivity == 2) {
        // 8-connected floods: no component decomposition here -- every plane goes through the sequential emulation
        AMT_TRY(amt_arena_begin(ctx, amt_align(np * sizeof(hp_elem)) + amt_align((size_t)nplanes * 4)));
        hp_elem* gheap = arena_take_t<hp_elem>(ctx, np);
        if (ties_dev) {
            hipLaunchKernelGGL(ws_set_flags_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, ties_dev, nplanes, 0);
            AMT_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(ws_global_init_kernel, dim3(amt_grid_for(n, 256, 1024), nplanes), dim3(256), 0, ctx->stream,
                           markers, mask, out, (const int*)nullptr, n);
        AMT_LAUNCH_CHECK();
        if (use_d2)
            hipLaunchKernelGGL((ws_global_kernel<true>), dim3(nplanes), dim3(64), GH_LDS_N * sizeof(gh_elem), ctx->stream, relief, mask, out, gheap,
                               (const int*)nullptr, H, W, n, 2);
        else
            hipLaunchKernelGGL((ws_global_kernel<false>), dim3(nplanes), dim3(64), GH_LDS_N * sizeof(gh_elem), ctx->stream, relief, mask, out, gheap,
                               (const int*)nullptr, H, W, n, 2);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    // per-component rows: a 4-connected component needs a background pixel between itself and the next one in
    // its row, so a plane has at most ceil(n / 2) components; only the first ncomp[plane] rows are touched
    const size_t row_stride = n / 2 + 1;
    const size_t nr = (size_t)nplanes * row_stride;
    // HBM queues: bucket mode needs <= n + ncomp ints for head and tail each (only the used prefix is
    // initialised); the heap needs <= n elements per plane.
    const size_t bstride = use_d2 ? n + row_stride : n;
    const int trows = amt_i_tile_rows(H);
    const size_t lcap = amt_i_rootlist_cap(W);
    const size_t nlist = (size_t)nplanes * trows;
    size_t need = 5 * amt_align(np * 4) + amt_align(nlist * lcap * 4) + amt_align(nlist * 4) + 10 * amt_align(nr * 4) + amt_align(nr * sizeof(comp_row)) +
                  9 * amt_align(nplanes * 4 * WS_CTR) + amt_align((size_t)nplanes * 4) + amt_align((size_t)PF_NCLS * (nplanes + 1) * 4) +
                  amt_align(64);
    need += use_d2 ? 2 * amt_align((size_t)nplanes * bstride * 4) : amt_align((size_t)nplanes * bstride * sizeof(hp_elem));
    // the sequential emulation's heap (every pixel is pushed at most once): the float64 path reuses its per-component
    // heap space, the bucket path needs it extra -- and only when ties are to be resolved exactly
    if (exact && use_d2) need += amt_align(np * sizeof(hp_elem));
    const size_t msz = (size_t)nplanes * ((size_t)max_label + 1);
    if (fused_labels) need += amt_align(msz * 4);
    const size_t ccl_ints = amt_i_ccl_scratch_ints(nplanes, H, W);
    need += amt_align(ccl_ints * 4);
    // the fused chain's path (d2 relief, marker list, clear_border + relabel fused): the mask's run tables serve the
    // statistics and the final mapping instead of the parent plane (AMT_WS_RUNS=0: the parent plane everywhere)
    const int segs = (W + 63) / 64;
    const int ntiles = nplanes * trows * segs;
    const bool runs = use_d2 && fused_labels && mk_list && ws_runs_enabled() && amt_i_ccl_runs_ok(mask, H, W, nplanes) &&
                      (reinterpret_cast<uintptr_t>(fused_labels) & 15) == 0 && (reinterpret_cast<uintptr_t>(relief) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    if (runs)
        need += amt_align((size_t)ntiles * 64 * 8) + amt_align((size_t)ntiles * RT_CAP * 2) + amt_align((size_t)ntiles * 4) +
                amt_align((size_t)ntiles * 64 * 2) + amt_align((size_t)ntiles * RT_CAP * 4);
    AMT_TRY(amt_arena_begin(ctx, need));
    int* L = arena_take_t<int>(ctx, np);
    int* T = arena_take_t<int>(ctx, np);
    int* moff = arena_take_t<int>(ctx, nr);
    int* boff = arena_take_t<int>(ctx, nr);
    int* cursor = arena_take_t<int>(ctx, nr);
    int* mlist = arena_take_t<int>(ctx, np);
    int* next = arena_take_t<int>(ctx, np);
    int* F = arena_take_t<int>(ctx, np);  // per-root fill value (only root positions are used)
    int* rootlist = arena_take_t<int>(ctx, nlist * lcap);  // tile-local roots, one list per tile row
    int* nroots = arena_take_t<int>(ctx, nlist);
    comp_row* rows = arena_take_t<comp_row>(ctx, nr);
    int* btot = arena_take_t<int>(ctx, nplanes);
    int* mtot = arena_take_t<int>(ctx, nplanes);
    // per plane: work counters of the six LDS flood classes XS / S / M / M2 / L / X [0..6) and of the HBM flood [6],
    // has_g [7], worklist sizes of the six classes [8..14), number of components [14]
    int* counters = arena_take_t<int>(ctx, (size_t)nplanes * WS_CTR);
    int* wl = arena_take_t<int>(ctx, (size_t)WS_NLISTS * nr);  // worklists of the LDS classes
    int* pf_idx = arena_take_t<int>(ctx, (size_t)PF_NCLS * (nplanes + 1));  // the persistent flood's item index
    int* pf_ctl = arena_take_t<int>(ctx, 16);
    int* ccl_scratch = arena_take_t<int>(ctx, ccl_ints);  // the tile labelling's flag + the tiles' column words
    int* ties = ties_dev ? ties_dev : arena_take_t<int>(ctx, nplanes);
    int* P = fused_labels ? arena_take_t<int>(ctx, msz) : nullptr;
    unsigned long long* tbits = runs ? arena_take_t<unsigned long long>(ctx, (size_t)ntiles * 64) : nullptr;
    unsigned short* rtab = runs ? arena_take_t<unsigned short>(ctx, (size_t)ntiles * RT_CAP) : nullptr;
    int* nruns = runs ? arena_take_t<int>(ctx, (size_t)ntiles) : nullptr;
    unsigned short* roff = runs ? arena_take_t<unsigned short>(ctx, (size_t)ntiles * 64) : nullptr;
    int* rcomp = runs ? arena_take_t<int>(ctx, (size_t)ntiles * RT_CAP) : nullptr;  // only the runs that exist are touched
    // with run tables the parent plane holds tile roots only; every pixel look-up goes through the tables
    amt_runtabs rt = {tbits, roff, rtab, rcomp, segs, trows};
    int *head = nullptr, *tail = nullptr;
    hp_elem* heap = nullptr;
    hp_elem* gheap = nullptr;
    if (use_d2) {
        head = arena_take_t<int>(ctx, (size_t)nplanes * bstride);
        tail = arena_take_t<int>(ctx, (size_t)nplanes * bstride);
        if (exact) gheap = arena_take_t<hp_elem>(ctx, np);
    } else {
        heap = arena_take_t<hp_elem>(ctx, (size_t)nplanes * bstride);
        gheap = heap;  // bstride == n: the per-component heaps are dead when the emulation starts
    }

    int* ncomp = counters + 16 * (size_t)nplanes;
    // counters, tie flags and the root lists' counts in ONE launch (three trivial launches cost ~5 us each in a stage
    // of 1.2 ms)
    hipLaunchKernelGGL(ws_init_kernel, dim3(16), dim3(256), 0, ctx->stream, counters, nplanes * WS_CTR, ties, nplanes, nroots,
                       (int)nlist);
    AMT_LAUNCH_CHECK();
    // components of the mask with dense ids in T (the order of the ids is irrelevant: components are
    // independent work items and nothing in the output depends on their numbering)
    if (runs) AMT_TRY(amt_i_ccl_tileroots_runs_u8(ctx, mask, L, rootlist, nroots, nplanes, H, W, tbits, rtab, nruns, roff));
    else AMT_TRY(amt_i_ccl_tileroots_u8(ctx, mask, L, rootlist, nroots, nplanes, H, W, ccl_scratch));
    hipLaunchKernelGGL(ws_roots_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, L, T, rootlist, nroots, ncomp,
                       lcap, n);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_i_propagate_roots(ctx, T, L, rootlist, nroots, nplanes, H, W));
    hipLaunchKernelGGL(ws_rows_init_kernel, dim3(64, nplanes), dim3(256), 0, ctx->stream, rows, ncomp, row_stride);
    AMT_LAUNCH_CHECK();
    int* has_g = counters + 8 * nplanes;
    int* wl_count = counters + 9 * nplanes;  // [WS_NLISTS][nplanes]
    const int pf_mode = use_d2 ? ws_persist() : 0;  // 0 = class launches, 1 = a whole CU per workgroup, 2 = half a CU
    const int pf_slots = pf_mode == 1 ? ws_pf_slots() : (pf_mode == 2 ? PF_SLOTS_HALF : 0);
    if (runs) {
        hipLaunchKernelGGL(ws_stats_runs_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, ctx->stream, (const int*)relief, tbits, rtab,
                           nruns, L, T, rows, row_stride, H, W, segs, trows, ntiles, rcomp);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ws_marker_stats_kernel, dim3(16, nplanes), dim3(256), 0, ctx->stream, mk_list, mk_count, mk_cap,
                           markers, L, T, rows, row_stride, n, rt, W);
    } else if (mk_list) {
        // the caller knows where the marker pixels are: the dense pass skips the marker plane
        hipLaunchKernelGGL((ws_stats_kernel<false>), dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream,
                           use_d2 ? (const int*)relief : (const int*)nullptr, L, T, markers, rows, row_stride, H, W,
                           use_d2 ? 1 : 0);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ws_marker_stats_kernel, dim3(16, nplanes), dim3(256), 0, ctx->stream, mk_list, mk_count, mk_cap,
                           markers, L, T, rows, row_stride, n, rt, W);
    } else {
        hipLaunchKernelGGL((ws_stats_kernel<true>), dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream,
                           use_d2 ? (const int*)relief : (const int*)nullptr, L, T, markers, rows, row_stride, H, W,
                           use_d2 ? 1 : 0);
    }
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ws_classify_kernel, dim3(64, nplanes), dim3(256), 0, ctx->stream, rows, ncomp, moff, boff, has_g,
                       wl, wl_count, F, n, nplanes, row_stride, use_d2 ? 1 : 0, pf_slots);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_i_propagate_roots(ctx, F, L, rootlist, nroots, nplanes, H, W));
    const bool gprep = fused_labels && use_d2;  // the fused chain: the HBM flood's whole preparation in one launch
    if (gprep) {
        hipLaunchKernelGGL(ws_g_prep_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, markers, L, T, F, out, rows, moff, boff,
                           cursor, mlist, head, ncomp, mtot, btot, has_g, row_stride, n, bstride, rt, W);
        AMT_LAUNCH_CHECK();
    } else {
        // the marker lists of the HBM flood's components: nearly always no plane has one, and 4,096 workgroups per plane
        // that only read a flag and leave cost 43 us per 48 planes -- 128 (grid-stride) do when a plane does have one
        dim3 g1(amt_grid_for(n, 256, 128), nplanes);
        // the seed pass gives `out` its final value everywhere except in flooded components.  The fused path needs no
        // such plane (its final pass derives every pixel from F and the flood's sparse writes): it seeds only planes that
        // hold a component for the HBM flood, which works in `out` itself -- same consideration for its grid
        hipLaunchKernelGGL(ws_seed_kernel, dim3(amt_grid_for(n, 1024, fused_labels ? 128 : 4096), nplanes), dim3(256), 0,
                           ctx->stream, markers, L, F, out, n, fused_labels ? (const int*)has_g : (const int*)nullptr, rt, W);
        AMT_LAUNCH_CHECK();
        // ---- HBM-path bookkeeping (usually empty: only components too large for an LDS tile) ----
        AMT_TRY(amt_scan_excl_dev(ctx, moff, ncomp, row_stride, mtot, nplanes));
        AMT_TRY(amt_scan_excl_dev(ctx, boff, ncomp, row_stride, btot, nplanes));
        hipLaunchKernelGGL(ws_fill_value_kernel, dim3(64, nplanes), dim3(256), 0, ctx->stream, cursor, ncomp, row_stride, 0);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ws_fill_markers_kernel, g1, dim3(256), 0, ctx->stream, L, T, out, rows, moff, cursor, mlist,
                           has_g, row_stride, n, rt, W);
        AMT_LAUNCH_CHECK();
    }
    if (use_d2) {
        const size_t ldsXS = (size_t)XS_PX * 6 + (size_t)XS_NB * 6;
        const size_t ldsS = (size_t)S_PX * 6 + (size_t)S_NB * 6;
        const size_t ldsM = (size_t)M_PX * 6 + (size_t)M_NB * 6;
        const size_t ldsM2 = (size_t)(M2_PX > 0 ? M2_PX : 64) * 6 + (size_t)M2_NB * 6;
        const size_t ldsL = (size_t)L_PX * 6 + (size_t)L_NB * 6;
        const size_t ldsX = (size_t)X_PX * 4 + (size_t)X_NB * 4;
        // the LDS classes and the HBM path are independent, latency-bound and use few waves each:
        // run them side by side (fork / join on the context's auxiliary streams)
        if (!gprep) {
            hipLaunchKernelGGL(ws_fill_value_kernel, dim3(256, nplanes), dim3(256), 0, ctx->stream, head, btot, bstride, -1);
            AMT_LAUNCH_CHECK();
        }
        AMT_TRY(amt_fork(ctx));
        // workgroups per plane and class (a workgroup only takes components of its own plane).  Measured: 2 x / 4 x as
        // many change nothing at 1, 12 or 32 planes per launch -- the chains per workgroup are not what bounds a class
        const int gXS = 96, gS = XS_PX > 0 ? 80 : 128, gM = XS_PX > 0 ? 48 : 64, gM2 = XS_PX > 0 ? 16 : 32, gL = 16;
        // Without auxiliary streams the classes still overlap: the first flood is an ordinary (barrier) launch, the
        // others carry hipExtAnyOrderLaunch -- their packets have no barrier bit, so the command processor dispatches
        // them while the earlier floods are still running; the next ordinary launch waits for all of them.
        const bool any = ctx->fork == 0 && ws_anyorder();
        int nflood = 0;
        auto flood = [&](const void* fn, int gx, size_t lds, hipStream_t st, int cls_slot) -> int {
            const int* a_d2 = (const int*)relief;
            const int* a_L = L;
            const int* a_T = T;
            int* a_out = out;
            const comp_row* a_rows = rows;
            const int* a_wl = wl + (size_t)cls_slot * nplanes * row_stride;
            const int* a_wlc = wl_count + cls_slot * nplanes;
            int* a_cnt = counters + cls_slot * nplanes;  // the class's work counter
            size_t a_rs = row_stride;
            int a_H = H, a_W = W, a_sf = seeds_first;
            int* a_ties = ties;
            const int* a_mk = markers;
            amt_runtabs a_rt = rt;
            void* args[] = {&a_d2, &a_L, &a_T, &a_out, &a_rows, &a_wl, &a_wlc, &a_cnt, &a_rs, &a_H, &a_W, &a_sf, &a_ties, &a_mk, &a_rt};
            AMT_HIP_CHECK(hipExtLaunchKernel(fn, dim3(gx, nplanes), dim3(64), args, lds + (cls_slot <= 2 ? ws_lds_pad() : 0), st, nullptr, nullptr,
                                             (any && nflood > 0) ? (int)hipExtAnyOrderLaunch : 0));
            ++nflood;
            return AMT_OK;
        };
        const bool persist = pf_mode != 0;
        if (persist) {
            // one persistent launch for the five batch classes, then LB (L boxes beyond a workgroup that owns less than a
            // whole CU), X and the HBM flood beside it.  Mode 1 (default): a workgroup of 10 waves per CU with all its
            // LDS; mode 2: two workgroups per CU with half each.  Measured on one box (watershed stage per 48 FOVs in one
            // context / 48-FOV plate / 192-FOV default, FOV/s): class launches 2.53 ms / 10.4 k / 11.5-11.7 k; whole CU,
            // 8 waves 2.11 / 10.4 k / 11.3 k; 10 waves 2.02 / 10.4 k / 11.6 k; 16 waves 2.10; 46 slots 2.16 / 10.2 k /
            // 10.9 k; half a CU x 2 2.26 / 10.4 k / 11.1 k.  The stage gains 0.5 ms; with four contexts side by side the
            // floods were already hidden behind the other contexts' streaming kernels, so the totals do not move.
            hipLaunchKernelGGL(ws_flood_index_kernel, dim3(1), dim3(64), 0, ctx->stream, wl_count, pf_idx, pf_ctl, nplanes);
            AMT_LAUNCH_CHECK();
            const int* a_d2 = (const int*)relief;
            const int* a_L = L;
            const int* a_T = T;
            int* a_out = out;
            const comp_row* a_rows = rows;
            const int* a_wl = wl;
            const int* a_idx = pf_idx;
            int* a_ctl = pf_ctl;
            size_t a_rs = row_stride;
            int a_H = H, a_W = W, a_sf = seeds_first, a_np = nplanes, a_slots = pf_slots;
            int* a_ties = ties;
            const int* a_mk = markers;
            amt_runtabs a_rt = rt;
            void* args[] = {&a_d2, &a_L, &a_T, &a_out, &a_rows, &a_wl, &a_idx, &a_ctl, &a_rs, &a_H, &a_W, &a_sf, &a_ties, &a_mk,
                            &a_np, &a_slots, &a_rt};
            const int wgs = pf_mode == 1 ? ctx->num_cus : 2 * ctx->num_cus;
            AMT_HIP_CHECK(hipExtLaunchKernel((const void*)ws_flood_persist_kernel, dim3(wgs), dim3(ws_pf_waves() * 64), args,
                                             (size_t)(pf_slots + 1) * PF_SLOT, ctx->stream, nullptr, nullptr, 0));
            ++nflood;
            if (pf_slots < PF_SLOTS_FULL) AMT_TRY(flood((const void*)ws_flood_batch_kernel<L_PX, L_NB>, gL, ldsL, ctx->stream, 6));
        } else {
            AMT_TRY(flood((const void*)ws_flood_batch_kernel<L_PX, L_NB>, gL, ldsL, ctx->stream, 4));
        }
        AMT_TRY(flood((const void*)ws_flood_lds_kernel<X_PX, X_NB, CLS_X>, 8, ldsX, ctx->stream, 5));
        if (!persist) {
        AMT_TRY(flood((const void*)ws_flood_batch_kernel<M2_PX, M2_NB>, gM2, ldsM2, ctx->aux[2], 3));
        AMT_TRY(flood((const void*)ws_flood_batch_kernel<M_PX, M_NB>, gM, ldsM, ctx->aux[0], 2));
        AMT_TRY(flood((const void*)ws_flood_batch_kernel<S_PX, S_NB>, gS, ldsS, ctx->aux[1], 1));
        if (XS_PX > 0) AMT_TRY(flood((const void*)ws_flood_batch_kernel<(XS_PX > 0 ? XS_PX : 64), XS_NB>, gXS, ldsXS, ctx->aux[2], 0));
        }
        {
            const int* a_d2 = (const int*)relief;
            void* args[] = {&a_d2, (void*)&mask, &out, &next, &head, &tail, &mlist, (void*)&rows, &moff, &boff, &ncomp, nullptr,
                            (void*)&row_stride, &H, &W, (void*)&n, (void*)&bstride, &seeds_first, &ties};
            int* a_cnt = counters + 7 * nplanes;
            args[11] = &a_cnt;
            AMT_HIP_CHECK(hipExtLaunchKernel((const void*)ws_flood_edt_kernel, dim3(4, nplanes), dim3(64), args, 0, ctx->aux[1],
                                             nullptr, nullptr, any ? (int)hipExtAnyOrderLaunch : 0));
        }
        AMT_TRY(amt_join(ctx));
    } else {
        hipLaunchKernelGGL(ws_flood_heap_kernel, dim3(64, nplanes), dim3(64), 0, ctx->stream, (const double*)relief,
                           mask, out, heap, mlist, rows, moff, boff, ncomp, counters + 7 * nplanes, row_stride, H, W, n,
                           bstride, ties);
    }
    AMT_LAUNCH_CHECK();
    if (exact) {
        // planes in which two markers of one component tied: scikit-image's answer comes from its single heap
        hipLaunchKernelGGL(ws_global_init_kernel, dim3(amt_grid_for(n, 256, 1024), nplanes), dim3(256), 0, ctx->stream,
                           markers, mask, out, (const int*)ties, n);
        AMT_LAUNCH_CHECK();
        if (use_d2)
            hipLaunchKernelGGL((ws_global_kernel<true>), dim3(nplanes), dim3(64), GH_LDS_N * sizeof(gh_elem), ctx->stream, relief, mask, out, gheap,
                               (const int*)ties, H, W, n, 1);
        else
            hipLaunchKernelGGL((ws_global_kernel<false>), dim3(nplanes), dim3(64), GH_LDS_N * sizeof(gh_elem), ctx->stream, relief, mask, out, gheap,
                               (const int*)ties, H, W, n, 1);
        AMT_LAUNCH_CHECK();
    }
    if (fused_labels) {
        // the label map of clear_border + relabel_sequential: present labels, frame-touching ones dropped, survivors
        // numbered -- one workgroup per plane does the four steps (they were four launches over a few thousand entries)
        hipLaunchKernelGGL(ws_label_map_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, L, F, out, P, nlabels_dev, fused_count,
                           H, W, max_label, rt);
        AMT_LAUNCH_CHECK();
        if (runs) {
            hipLaunchKernelGGL(ws_final_map_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, F, P, rootlist, nroots, lcap,
                               n, max_label);
            AMT_LAUNCH_CHECK();
        }
        if (runs)
            hipLaunchKernelGGL(ws_final_runs_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, ctx->stream, tbits, rtab, nruns, F, out, P,
                               fused_labels, H, W, segs, trows, ntiles, max_label);
        else
            hipLaunchKernelGGL(ws_final_kernel, dim3(amt_grid_for(n, 1024, 4096), nplanes), dim3(256), 0, ctx->stream, L, F, out,
                               P, fused_labels, n, max_label);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

extern "C" int amt_watershed_edt_cleared(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                                         int32_t* ws_scratch, int32_t* labels_out, int32_t* count_dev, int nplanes,
                                         int H, int W, int max_label, const int32_t* nlabels_dev) {
    AMT_REQUIRE(ws_scratch && labels_out && count_dev && nlabels_dev && max_label >= 0,
                "watershed_edt_cleared: bad arguments");
    AMT_REQUIRE(ws_scratch != labels_out, "watershed_edt_cleared: scratch and output must not alias");
    return watershed_common(ctx, d2, true, markers, mask, ws_scratch, nplanes, H, W, 1, 1, AMT_WS_TIES_EXACT, nullptr,
                            labels_out, count_dev, nlabels_dev, max_label);
}

extern "C" int amt_watershed_edt_cleared_sparse(amt_ctx* ctx, const int32_t* d2, const int32_t* markers,
                                                const uint8_t* mask, int32_t* ws_scratch, int32_t* labels_out,
                                                int32_t* count_dev, int nplanes, int H, int W, int max_label,
                                                const int32_t* nlabels_dev, const int32_t* marker_list,
                                                const int32_t* marker_count, int list_capacity) {
    AMT_REQUIRE(ws_scratch && labels_out && count_dev && nlabels_dev && max_label >= 0 && marker_list && marker_count &&
                    list_capacity > 0,
                "watershed_edt_cleared_sparse: bad arguments");
    AMT_REQUIRE(ws_scratch != labels_out, "watershed_edt_cleared_sparse: scratch and output must not alias");
    return watershed_common(ctx, d2, true, markers, mask, ws_scratch, nplanes, H, W, 1, 1, AMT_WS_TIES_EXACT, nullptr,
                            labels_out, count_dev, nlabels_dev, max_label, marker_list, marker_count, list_capacity);
}

extern "C" int amt_watershed_edt_ex(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                                    int32_t* out, int nplanes, int H, int W, int seeds_first, int connectivity,
                                    int tie_policy, int32_t* ties_dev) {
    return watershed_common(ctx, d2, true, markers, mask, out, nplanes, H, W, seeds_first, connectivity, tie_policy,
                            ties_dev);
}

extern "C" int amt_watershed_f64_ex(amt_ctx* ctx, const double* relief, const int32_t* markers, const uint8_t* mask,
                                    int32_t* out, int nplanes, int H, int W, int connectivity, int tie_policy,
                                    int32_t* ties_dev) {
    return watershed_common(ctx, relief, false, markers, mask, out, nplanes, H, W, 0, connectivity, tie_policy, ties_dev);
}

extern "C" int amt_watershed_edt(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                                 int32_t* out, int nplanes, int H, int W, int seeds_first) {
    return watershed_common(ctx, d2, true, markers, mask, out, nplanes, H, W, seeds_first, 1, AMT_WS_TIES_EXACT, nullptr);
}

extern "C" int amt_watershed_f64(amt_ctx* ctx, const double* relief, const int32_t* markers, const uint8_t* mask,
                                 int32_t* out, int nplanes, int H, int W) {
    return watershed_common(ctx, relief, false, markers, mask, out, nplanes, H, W, 0, 1, AMT_WS_TIES_EXACT, nullptr);
}
