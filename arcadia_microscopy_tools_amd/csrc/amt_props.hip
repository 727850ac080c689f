// Region properties as segmented integer reductions keyed by label.
//
// Reference call site: R/masks.py:286-326 (ski.measure.regionprops_table, once for morphology and once
// per channel for intensity).  Property definitions: SK/measure/_regionprops.py:277-468,
// SK/measure/_regionprops_utils.py:186-249 (perimeter), SK/measure/_moments.py:379-445 (inertia tensor),
// SK/morphology/convex_hull.py (area_convex) -- SURVEY.md section 8a row A12, A.7, A.12.
//
// All accumulation is integer (counts, coordinate sums, intensity sums) and therefore independent of
// the order in which pixels arrive; floating point only enters in the final per-label kernel.
// A wave covers 64 consecutive pixels of one row; equal-label runs inside it are reduced in closed
// form (coordinates) or with wave prefix sums (intensities), so each run costs one set of atomics.
#include "amt_internal.h"

typedef unsigned long long u64;

// accumulator slots per (plane, label)
enum { A_N = 0, A_SY, A_SX, A_SYY, A_SXX, A_SXY, A_C1, A_C2, A_C3, A_NACC };
// bbox ints per (plane, label): miny, minx, maxy, maxx

__global__ void __launch_bounds__(256) rp_init_kernel(u64* __restrict__ acc, int* __restrict__ bbox, size_t nlab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nlab; i += (size_t)gridDim.x * 256) {
        for (int k = 0; k < A_NACC; ++k) acc[i * A_NACC + k] = 0;
        bbox[i * 4 + 0] = 0x7fffffff;
        bbox[i * 4 + 1] = 0x7fffffff;
        bbox[i * 4 + 2] = -1;
        bbox[i * 4 + 3] = -1;
    }
}

__device__ __forceinline__ u64 sum_sq_upto(u64 m) { return m * (m + 1) * (2 * m + 1) / 6; }  // 0^2 + ... + m^2

// exact a*b - c*d for 64-bit operands, as a double (the difference itself must fit in 127 bits)
__device__ __forceinline__ double diff_of_products(u64 a, u64 b, u64 c, u64 d) {
    u64 lo1 = a * b, hi1 = __umul64hi(a, b);
    u64 lo2 = c * d, hi2 = __umul64hi(c, d);
    // (hi1:lo1) - (hi2:lo2), signed
    bool neg = (hi1 < hi2) || (hi1 == hi2 && lo1 < lo2);
    u64 hi, lo;
    if (!neg) {
        lo = lo1 - lo2;
        hi = hi1 - hi2 - (lo1 < lo2 ? 1 : 0);
    } else {
        lo = lo2 - lo1;
        hi = hi2 - hi1 - (lo2 < lo1 ? 1 : 0);
    }
    double v = (double)hi * 18446744073709551616.0 + (double)lo;
    return neg ? -v : v;
}

// bounding boxes: a wave covers 64 consecutive pixels of 8 rows (all 8 loads issued up front); every
// run of equal labels costs one set of min/max atomics.
__global__ void __launch_bounds__(256) rp_bbox_kernel(const int* __restrict__ labels, int* __restrict__ bbox, int H,
                                                      int W, int max_label) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int yb = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 8;
    if (yb >= H) return;
    const size_t n = (size_t)H * W;
    const int plane = blockIdx.z;
    int labs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int y = yb + k;
        int v = 0;
        if (x < W && y < H) v = labels[(size_t)plane * n + (size_t)y * W + x];
        labs[k] = (v < 0 || v > max_label) ? 0 : v;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lab = labs[k];
        const int left = __shfl_up(lab, 1);
        const bool head = (lane == 0) || (left != lab);
        const u64 heads = __ballot(head);
        if (lab != 0 && head) {
            const u64 later = heads & ~((2ull << lane) - 1ull);
            const int end_lane = later ? (__ffsll((long long)later) - 2) : 63;
            int* B = bbox + ((size_t)plane * max_label + (lab - 1)) * 4;
            atomicMin(&B[0], yb + k);
            atomicMin(&B[1], x);
            atomicMax(&B[2], yb + k);
            atomicMax(&B[3], x + (end_lane - lane));
        }
    }
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned t = __shfl_xor(v, off);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned t = __shfl_xor(v, off);
        v = t > v ? t : v;
    }
    return v;
}

// One wave per label scans the label's bounding box (and nothing else): exact integer moment sums,
// per-row extents for the convex hull, and {sum, sum of squares, min, max} of up to 4 intensity channels
// per call, all accumulated privately per lane and reduced once -- no atomics, run-to-run identical.
constexpr int RP_MAXC = 4;
__global__ void __launch_bounds__(64) rp_label_kernel(const int* __restrict__ labels, const int* __restrict__ bbox,
                                                      const int* __restrict__ hoff, const int* __restrict__ htot,
                                                      int2* __restrict__ rows, size_t cap, u64* __restrict__ acc,
                                                      const uint16_t* __restrict__ inten, int C, int c0, int nc,
                                                      double* __restrict__ itable, int H, int W, int max_label,
                                                      int want_morph) {
    const int plane = blockIdx.y;
    const int l = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t li = (size_t)plane * max_label + l;
    const int y0 = bbox[li * 4 + 0], x0 = bbox[li * 4 + 1], y1 = bbox[li * 4 + 2], x1 = bbox[li * 4 + 3];
    if (y1 < y0) {  // label absent from this plane
        if (itable && lane < 4 * nc) itable[(li * C + c0) * 4 + lane] = 0.0;
        return;
    }
    const size_t n = (size_t)H * W;
    const int* L = labels + (size_t)plane * n;
    const uint16_t* I = inten ? inten + ((size_t)plane * C + c0) * n : nullptr;
    const int want = l + 1;
    const bool rows_ok = want_morph && (size_t)htot[plane] <= cap;
    int2* myrows = rows + (size_t)plane * cap + (size_t)hoff[li];
    u64 cnt = 0, sy = 0, sx = 0, syy = 0, sxx = 0, sxy = 0;
    u64 s[RP_MAXC], q[RP_MAXC];
    unsigned mn[RP_MAXC], mx[RP_MAXC];
#pragma unroll
    for (int c = 0; c < RP_MAXC; ++c) {
        s[c] = 0;
        q[c] = 0;
        mn[c] = 0xffffffffu;
        mx[c] = 0;
    }
#ifndef AMT_RP_RPS
#define AMT_RP_RPS 3
#endif
    constexpr int RPS = AMT_RP_RPS;  // rows per step
    // eight rows per step: their label loads are issued together, then the intensity loads of the hits
    for (int yb = y0; yb <= y1; yb += RPS) {
        int rmin[RPS], rmax[RPS];
#pragma unroll
        for (int j = 0; j < RPS; ++j) {
            rmin[j] = 0x7fffffff;
            rmax[j] = -1;
        }
        for (int xb = x0; xb <= x1; xb += 64) {
            const int x = xb + lane;
            // a lane's pixels of this block share ONE column: sx, sxx and sxy follow from the lane's pixel count and row
            // sum at the end of the block (x * count, x^2 * count, x * sum of y) -- per row a lane only counts and adds y
            unsigned bcnt = 0, bsy = 0;  // <= 8 rows: no overflow
            // ONE round trip per step: the labels and the intensities of the box's eight rows are requested together,
            // unconditionally, from clamped coordinates (the intensities used to wait for the labels they depend on: two
            // dependent round trips per step of a wave that has nothing else to do); what lies outside the box or the
            // label is discarded where it is used
            const int xc = x <= x1 ? x : x1;
            int lv[RPS];
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int yc = yb + j <= y1 ? yb + j : y1;
                lv[j] = L[(size_t)yc * W + xc];
            }
            unsigned iv[RPS][RP_MAXC];
            if (I) {
#pragma unroll
                for (int j = 0; j < RPS; ++j) {
                    const int yc = yb + j <= y1 ? yb + j : y1;
#pragma unroll
                    for (int c = 0; c < RP_MAXC; ++c) iv[j][c] = I[(size_t)(c < nc ? c : 0) * n + (size_t)yc * W + xc];
                }
            }
#pragma unroll
            for (int j = 0; j < RPS; ++j)
                if (x > x1 || yb + j > y1) lv[j] = 0;
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int y = yb + j;
                const bool m = lv[j] == want;
                if (m) {
                    bcnt += 1u;
                    bsy += (unsigned)y;
                    syy += (u64)(unsigned)y * (unsigned)y;
                    if (I) {
#pragma unroll
                        for (int c = 0; c < RP_MAXC; ++c)
                            if (c < nc) {
                                const unsigned v = iv[j][c];
                                s[c] += v;
                                q[c] += (u64)v * v;
                                mn[c] = v < mn[c] ? v : mn[c];
                                mx[c] = v > mx[c] ? v : mx[c];
                            }
                    }
                }
                const u64 bal = __ballot(m);
                if (bal) {
                    const int first = xb + __ffsll((long long)bal) - 1;
                    const int last = xb + 63 - __clzll((long long)bal);
                    rmin[j] = first < rmin[j] ? first : rmin[j];
                    rmax[j] = last > rmax[j] ? last : rmax[j];
                }
            }
            cnt += bcnt;
            sy += bsy;
            sx += (u64)(unsigned)x * bcnt;
            sxx += (u64)(unsigned)x * (unsigned)x * bcnt;
            sxy += (u64)(unsigned)x * bsy;
        }
        if (rows_ok && lane == 0) {
#pragma unroll
            for (int j = 0; j < RPS; ++j)
                if (yb + j <= y1) myrows[yb + j - y0] = make_int2(rmin[j], rmax[j]);
        }
    }
    cnt = wave_sum_u64(cnt);
    if (want_morph) {
        sy = wave_sum_u64(sy);
        sx = wave_sum_u64(sx);
        syy = wave_sum_u64(syy);
        sxx = wave_sum_u64(sxx);
        sxy = wave_sum_u64(sxy);
        if (lane == 0) {
            u64* A = acc + li * A_NACC;
            A[A_N] = cnt;
            A[A_SY] = sy;
            A[A_SX] = sx;
            A[A_SYY] = syy;
            A[A_SXX] = sxx;
            A[A_SXY] = sxy;
        }
    }
    if (itable) {
#pragma unroll
        for (int c = 0; c < RP_MAXC; ++c)
            if (c < nc) {
                const u64 S = wave_sum_u64(s[c]), Q = wave_sum_u64(q[c]);
                const unsigned lo = wave_min_u32(mn[c]), hi = wave_max_u32(mx[c]);
                if (lane == 0) {
                    double* t = itable + (li * C + c0 + c) * 4;
                    if (cnt == 0) {
                        t[0] = t[1] = t[2] = t[3] = 0.0;
                    } else {
                        const double dn = (double)cnt;
                        t[0] = (double)S / dn;
                        t[1] = (double)hi;
                        t[2] = (double)lo;
                        const double nv = diff_of_products(cnt, Q, S, S);  // n*Sxx - Sx^2 = n^2 * var
                        const double var = nv / (dn * dn);
                        t[3] = sqrt(var < 0.0 ? 0.0 : var);
                    }
                }
            }
    }
}

// perimeter: border pixels (4-neighbourhood, outside = background) and their 3x3 weighted codes
// (SK/measure/_regionprops_utils.py:186-249); with `bbox` the kernel also folds the bounding boxes (per-run
// atomics, only for runs that can be an extreme of their label), which spares the morphology path a separate
// pass over the label image.
// No LDS and no barrier: a wave owns a strip of PR_IN columns (+ 2 halo columns on
// either side = 64 lanes) and slides down PR_ROWS rows (+ 2 halo rows above / below) with a three-row window in
// registers.  Horizontal neighbours come from DPP wave shifts, the "is a border pixel" flag rides in bit 31 of the
// label (w = v | flag), so "neighbour is a border pixel of MY label" is one compare, w(q) == w(p).  Rows whose
// whole window is background skip everything.  Loads are issued 17 rows ahead of their use.
constexpr int PR_ROWS = 64, PR_IN = 60, PR_BATCH = 17, PR_NBATCH = (PR_ROWS + 4) / PR_BATCH;
static_assert(PR_BATCH * PR_NBATCH == PR_ROWS + 4, "row batches must tile the strip");

__global__ void __launch_bounds__(256) rp_perimeter_rows_kernel(const int* __restrict__ labels, u64* __restrict__ acc,
                                                                int H, int W, int max_label, int* __restrict__ bbox) {
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip * PR_IN >= W) return;  // whole wave; the kernel has no barrier
    const int plane = blockIdx.z;
    const size_t n = (size_t)H * W;
    const int* L = labels + (size_t)plane * n;
    const int x = strip * PR_IN - 2 + lane;
    const int y0 = blockIdx.y * PR_ROWS;
    const bool xin = x >= 0 && x < W;
    const bool inner = xin && lane >= 2 && lane < 2 + PR_IN;
    const unsigned ml = (unsigned)max_label;
    u64* accp = acc + (size_t)plane * max_label * A_NACC;
    int* bbp = bbox ? bbox + (size_t)plane * max_label * 4 : nullptr;

    // UNCONDITIONAL loads from clamped coordinates; positions outside the image are zeroed where the value is
    // used (a select next to the load would be turned back into a branch around it, and the wait that comes with
    // it serialises the loads)
    const int xc = x < 0 ? 0 : (x < W ? x : W - 1);
    auto load_row = [&](int y) -> int {
        const int yc = y < 0 ? 0 : (y < H ? y : H - 1);
        return L[(size_t)yc * W + xc];
    };
    int cur[PR_BATCH], nxt[PR_BATCH];
#pragma unroll
    for (int j = 0; j < PR_BATCH; ++j) cur[j] = load_row(y0 - 2 + j);
    // window: v1 / v2 = label rows r-1 / r-2 (+ their horizontal neighbours), wA / wB = flagged rows r-2 / r-3
    int v1 = 0, v1l = 0, v1r = 0, v2 = 0, v2l = 0, v2r = 0;
    int wA = 0, wAl = 0, wAr = 0, wB = 0, wBl = 0, wBr = 0;
    for (int b = 0; b < PR_NBATCH; ++b) {
        if (b + 1 < PR_NBATCH) {
#pragma unroll
            for (int j = 0; j < PR_BATCH; ++j) nxt[j] = load_row(y0 - 2 + (b + 1) * PR_BATCH + j);
        }
#pragma unroll
        for (int j = 0; j < PR_BATCH; ++j) {
            const int r = y0 - 2 + b * PR_BATCH + j;  // row of v0
            int v0 = cur[j];
            v0 = (xin && r >= 0 && r < H && (unsigned)(v0 - 1) < ml) ? v0 : 0;
#if defined(PR_EXP) && PR_EXP == 1
            v1 |= v0;  // what-if: the loads alone
            continue;
#endif
            if (__ballot((v0 | v1 | v2 | wB) != 0) == 0ull) continue;  // uniform: the window stays all zero
            const int v0l = amt_lane_left(v0), v0r = amt_lane_right(v0);
            // flagged row r-1
            const bool b1 = v1 != 0 && (v2 != v1 || v0 != v1 || v1l != v1 || v1r != v1);
            const int w1 = b1 ? (v1 | (int)0x80000000) : v1;
            const int w1l = amt_lane_left(w1), w1r = amt_lane_right(w1);
            // bounding box: runs of row r-1 (above v2, below v0)
            const int yb = r - 1;
            if (bbp && yb >= y0 && yb < y0 + PR_ROWS) {  // uniform
                // the four requests as flat per-lane conditions behind ONE uniform test: interior rows of a blob ask for
                // nothing (their runs have the label above, below and further out on either side), and every divergent
                // block costs scalar issue slots whether a lane enters it or not
                const bool live = inner && v1 != 0, head = live && v1l != v1;
                const bool q0 = head && v2 != v1, q2 = head && v0 != v1, q1 = head && v2l != v1 && v0l != v1;
                const bool q3 = live && v1r != v1 && v2r != v1 && v0r != v1;  // tail of a run
                if (__ballot(q0 || q1 || q2 || q3)) {
                    int* B = bbp + (size_t)(v1 - 1) * 4;
                    if (q0) atomicMin(&B[0], yb);
                    if (q2) atomicMax(&B[2], yb);
                    if (q1) atomicMin(&B[1], x);
                    if (q3) atomicMax(&B[3], x);
                }
            }
            // perimeter code of row r-2 (centre wA, above wB, below w1)
            const int yo = r - 2;
            if (inner && wA < 0 && yo >= y0 && yo < y0 + PR_ROWS) {
                const int key = wA;
                const int code = 1 + 2 * ((wB == key) + (w1 == key) + (wAl == key) + (wAr == key)) +
                                 10 * ((wBl == key) + (wBr == key) + (w1l == key) + (w1r == key));
                // class of the code without a branch per case (the pass is bound by SCALAR issue: every divergent block
                // costs an exec save, a branch and a restore): codes are <= 49, the three sets are bit masks
                constexpr unsigned long long M1 = (1ull << 5) | (1ull << 7) | (1ull << 15) | (1ull << 17) | (1ull << 25) | (1ull << 27);
                constexpr unsigned long long M2 = (1ull << 21) | (1ull << 33), M3 = (1ull << 13) | (1ull << 23);
                const int cls = ((M1 >> code) & 1ull) ? A_C1 : ((M2 >> code) & 1ull) ? A_C2 : ((M3 >> code) & 1ull) ? A_C3 : -1;
                if (cls >= 0) atomicAdd(&accp[(size_t)((key & 0x7fffffff) - 1) * A_NACC + cls], 1ull);
            }
            wB = wA, wBl = wAl, wBr = wAr;
            wA = w1, wAl = w1l, wAr = w1r;
            v2 = v1, v2l = v1l, v2r = v1r;
            v1 = v0, v1l = v0l, v1r = v0r;
        }
#pragma unroll
        for (int j = 0; j < PR_BATCH; ++j) cur[j] = nxt[j];
    }
#if defined(PR_EXP) && PR_EXP == 1
    if (v1 == 0x12345678) acc[0] = 1;
#endif
}

// ---- convex area (exact integer hull of the pixel diamonds) -------------------------------------
// rows[plane][off(label) + (y - miny)] = {min x, max x} of the label in that row
__global__ void __launch_bounds__(256) rp_heights_kernel(const int* __restrict__ bbox, int* __restrict__ hoff,
                                                         int max_label) {
    const int plane = blockIdx.y;
    for (int l = blockIdx.x * 256 + threadIdx.x; l < max_label; l += gridDim.x * 256) {
        const int* B = bbox + ((size_t)plane * max_label + l) * 4;
        hoff[(size_t)plane * max_label + l] = B[2] >= B[0] ? B[2] - B[0] + 1 : 0;
    }
}

// labels of at most HULL_HMAX rows and HULL_WMAX columns go to rp_hull_lds_kernel, the rest to rp_hull_kernel
constexpr int HULL_HMAX = 48, HULL_WMAX = 250;

__device__ __forceinline__ long long floor_div(long long a, long long b) {  // b > 0
    long long q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ long long ceil_div(long long a, long long b) {  // b > 0
    long long q = a / b;
    return (a % b != 0 && a > 0) ? q + 1 : q;
}

// One lane per label.  Work in doubled coordinates (Y = 2y, X = 2x): row y with extent [a, b]
// contributes the diamond points (2y-1, 2a), (2y+1, 2a), (2y, 2a-1) on the left and
// (2y-1, 2b), (2y+1, 2b), (2y, 2b+1) on the right (the inner diamond points can never be extreme).
// Left chain = lower-left convex boundary of {(Y, minX(Y))}, right chain of {(Y, maxX(Y))}, built by a
// monotone scan in Y; then pixel centres (2y, 2x) with XL(2y) <= 2x <= XR(2y) are counted exactly.
__global__ void __launch_bounds__(64) rp_hull_kernel(const int* __restrict__ bbox, const int* __restrict__ hoff,
                                                     const int* __restrict__ htot, const int2* __restrict__ rows,
                                                     int2* __restrict__ chainL, int2* __restrict__ chainR, size_t cap,
                                                     double* __restrict__ table, int max_label, int skip_h) {
    const int plane = blockIdx.y;
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= max_label) return;
    const size_t li = (size_t)plane * max_label + l;
    double* trow = table + li * AMT_RP_NCOLS;
    const int miny = bbox[li * 4 + 0], maxy = bbox[li * 4 + 2];
    if (maxy < miny) {
        if (skip_h == 0) trow[AMT_RP_AREA_CONVEX] = 0.0;
        return;
    }
    const int h = maxy - miny + 1;
    if (h <= skip_h && bbox[li * 4 + 3] - bbox[li * 4 + 1] + 1 <= HULL_WMAX) return;  // done by rp_hull_lds_kernel
    const size_t off = (size_t)hoff[li];
    if ((size_t)htot[plane] > cap || off + (size_t)h > cap) {  // capacity exceeded (fragmented labels)
        trow[AMT_RP_AREA_CONVEX] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const int2* r = rows + (size_t)plane * cap + off;
    // chains hold up to 2h+1 vertices each; scratch capacity per plane is 3*cap: offset 3*off, len 3*h
    int2* CL = chainL + (size_t)plane * 3 * cap + 3 * off;
    int2* CR = chainR + (size_t)plane * 3 * cap + 3 * off;
    int nl = 0, nr = 0;
    // iterate Y = 2*miny-1 .. 2*maxy+1
    for (int Y = 2 * miny - 1; Y <= 2 * maxy + 1; ++Y) {
        int mn = 0x7fffffff, mx = -0x7fffffff;
        if (Y & 1) {  // odd: shared by rows (Y-1)/2 and (Y+1)/2
            int ya = (Y - 1) / 2, yb = (Y + 1) / 2;
            if (Y < 0) {  // only for miny == 0: Y = -1 -> rows -1 (none) and 0
                ya = -1;
                yb = 0;
            }
            if (ya >= miny && ya <= maxy) {
                int2 e = r[ya - miny];
                if (e.y >= e.x) {
                    mn = min(mn, 2 * e.x);
                    mx = max(mx, 2 * e.y);
                }
            }
            if (yb >= miny && yb <= maxy) {
                int2 e = r[yb - miny];
                if (e.y >= e.x) {
                    mn = min(mn, 2 * e.x);
                    mx = max(mx, 2 * e.y);
                }
            }
        } else {
            int2 e = r[Y / 2 - miny];
            if (e.y >= e.x) {
                mn = 2 * e.x - 1;
                mx = 2 * e.y + 1;
            }
        }
        if (mx < mn) continue;  // no pixel of this label contributes at this Y
        // left chain: keep it convex towards -X: pop while the last vertex is not strictly left of the
        // segment (prev -> new)
        while (nl >= 2) {
            int2 p0 = CL[nl - 2], p1 = CL[nl - 1];
            long long cr = (long long)(p1.x - p0.x) * (mn - p0.y) - (long long)(p1.y - p0.y) * (Y - p0.x);
            // points are (Y, X) stored as (x=Y, y=X); cr = dY1*dX2 - dX1*dY2 ; left chain needs cr > 0 to keep p1
            if (cr <= 0) --nl; else break;
        }
        CL[nl++] = make_int2(Y, mn);
        while (nr >= 2) {
            int2 p0 = CR[nr - 2], p1 = CR[nr - 1];
            long long cr = (long long)(p1.x - p0.x) * (mx - p0.y) - (long long)(p1.y - p0.y) * (Y - p0.x);
            if (cr >= 0) --nr; else break;
        }
        CR[nr++] = make_int2(Y, mx);
    }
    // count pixel centres
    long long count = 0;
    int il = 0, ir = 0;
    for (int y = miny; y <= maxy; ++y) {
        const int Y = 2 * y;
        while (il + 1 < nl && CL[il + 1].x <= Y) ++il;
        while (ir + 1 < nr && CR[ir + 1].x <= Y) ++ir;
        // left bound XL(Y) on segment il -> il+1 (or vertex if exactly at / past the end)
        long long xmin, xmax;
        {
            int2 p0 = CL[il];
            if (p0.x == Y || il + 1 >= nl) {
                xmin = ceil_div(p0.y, 2);
            } else {
                int2 p1 = CL[il + 1];
                long long dY = p1.x - p0.x;  // > 0
                long long num = (long long)p0.y * dY + (long long)(p1.y - p0.y) * (Y - p0.x);  // XL * dY
                xmin = ceil_div(num, 2 * dY);
            }
        }
        {
            int2 p0 = CR[ir];
            if (p0.x == Y || ir + 1 >= nr) {
                xmax = floor_div(p0.y, 2);
            } else {
                int2 p1 = CR[ir + 1];
                long long dY = p1.x - p0.x;
                long long num = (long long)p0.y * dY + (long long)(p1.y - p0.y) * (Y - p0.x);
                xmax = floor_div(num, 2 * dY);
            }
        }
        if (xmax >= xmin) count += xmax - xmin + 1;
    }
    trow[AMT_RP_AREA_CONVEX] = (double)count;
}

// The same hull for labels that fit a small box (at most HULL_HMAX rows and HULL_WMAX columns: every nucleus-sized
// label): one lane per label, row extents and both chains in LDS, interleaved over the wave ([word][lane]), all
// arithmetic in 32 bits.  Coordinates are taken relative to the box (x - x0, Y - Y0), so a row is one 16-bit word
// (min | max << 8), a chain vertex one 16-bit word ((Y - Y0) << 9 | (X - X0)) and 8 waves fit a CU; every product
// stays below 2^20.  The HBM-scratch kernel above takes the labels that do not fit.
constexpr int HULL_CH = 2 * HULL_HMAX + 1;
__device__ __forceinline__ int floor_div32(int a, int b) {  // b > 0
    const int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int ceil_div32(int a, int b) {  // b > 0
    const int q = a / b;
    return (a % b != 0 && a > 0) ? q + 1 : q;
}
__global__ void __launch_bounds__(64) rp_hull_lds_kernel(const int* __restrict__ bbox, const int* __restrict__ hoff,
                                                         const int* __restrict__ htot, const int2* __restrict__ rows,
                                                         size_t cap, double* __restrict__ table, int max_label) {
    __shared__ unsigned short s_rows[HULL_HMAX * 64];
    __shared__ unsigned short s_cl[HULL_CH * 64];
    __shared__ unsigned short s_cr[HULL_CH * 64];
    const int plane = blockIdx.y;
    const int lane = threadIdx.x;
    const int l = blockIdx.x * 64 + lane;
    if (l >= max_label) return;
    const size_t li = (size_t)plane * max_label + l;
    double* trow = table + li * AMT_RP_NCOLS;
    const int miny = bbox[li * 4 + 0], x0 = bbox[li * 4 + 1], maxy = bbox[li * 4 + 2], x1 = bbox[li * 4 + 3];
    if (maxy < miny) {
        trow[AMT_RP_AREA_CONVEX] = 0.0;
        return;
    }
    const int h = maxy - miny + 1;
    if (h > HULL_HMAX || x1 - x0 + 1 > HULL_WMAX) return;  // rp_hull_kernel takes it
    const size_t off = (size_t)hoff[li];
    if ((size_t)htot[plane] > cap || off + (size_t)h > cap) {  // capacity exceeded (fragmented labels)
        trow[AMT_RP_AREA_CONVEX] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const int2* r = rows + (size_t)plane * cap + off;
    // eight row extents per round trip: loads from clamped indices, issued before the first of them is used (a
    // load -> LDS store loop waits for every load in turn)
    for (int k0 = 0; k0 < h; k0 += 8) {
        int2 e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = r[min(k0 + u, h - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < h)  // an empty row (max < min) keeps that property: 255 | 0 << 8
                s_rows[(k0 + u) * 64 + lane] =
                    e[u].y >= e[u].x ? (unsigned short)((e[u].x - x0) | ((e[u].y - x0) << 8)) : (unsigned short)255;
    }
    // doubled, box-relative coordinates: Yr = Y - (2 miny - 1) in [0, 2h], Xr = X - (2 x0 - 1) in [0, 2w]
    int nl = 0, nr = 0;
    for (int Yr = 0; Yr <= 2 * h; ++Yr) {
        int mn = 0x7fffffff, mx = -1;
        if (!(Yr & 1)) {  // odd Y: shared by the rows above and below
            const int ka = (Yr >> 1) - 1, kb = Yr >> 1;
            if (ka >= 0) {
                const unsigned e = s_rows[ka * 64 + lane];
                if ((e >> 8) >= (e & 255)) {
                    mn = min(mn, 2 * (int)(e & 255) + 1);
                    mx = max(mx, 2 * (int)(e >> 8) + 1);
                }
            }
            if (kb < h) {
                const unsigned e = s_rows[kb * 64 + lane];
                if ((e >> 8) >= (e & 255)) {
                    mn = min(mn, 2 * (int)(e & 255) + 1);
                    mx = max(mx, 2 * (int)(e >> 8) + 1);
                }
            }
        } else {
            const unsigned e = s_rows[(Yr >> 1) * 64 + lane];
            if ((e >> 8) >= (e & 255)) {
                mn = 2 * (int)(e & 255);
                mx = 2 * (int)(e >> 8) + 2;
            }
        }
        if (mx < mn) continue;  // no pixel of this label contributes at this Y
        while (nl >= 2) {
            const int v0 = s_cl[(nl - 2) * 64 + lane], v1 = s_cl[(nl - 1) * 64 + lane];
            const int cr = ((v1 >> 9) - (v0 >> 9)) * (mn - (v0 & 511)) - ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));
            if (cr <= 0) --nl; else break;
        }
        s_cl[(nl++) * 64 + lane] = (unsigned short)((Yr << 9) | mn);
        while (nr >= 2) {
            const int v0 = s_cr[(nr - 2) * 64 + lane], v1 = s_cr[(nr - 1) * 64 + lane];
            const int cr = ((v1 >> 9) - (v0 >> 9)) * (mx - (v0 & 511)) - ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));
            if (cr >= 0) --nr; else break;
        }
        s_cr[(nr++) * 64 + lane] = (unsigned short)((Yr << 9) | mx);
    }
    // count pixel centres: row k sits at Yr = 2k + 1; pixel x at Xr = 2 (x - x0) + 1
    int count = 0;
    int il = 0, ir = 0;
    for (int k = 0; k < h; ++k) {
        const int Yr = 2 * k + 1;
        while (il + 1 < nl && (int)(s_cl[(il + 1) * 64 + lane] >> 9) <= Yr) ++il;
        while (ir + 1 < nr && (int)(s_cr[(ir + 1) * 64 + lane] >> 9) <= Yr) ++ir;
        int xmin, xmax;  // pixel index bounds: XL <= 2 x + 1 <= XR
        {
            const int v0 = s_cl[il * 64 + lane];
            if ((v0 >> 9) == Yr || il + 1 >= nl) {
                xmin = ceil_div32((v0 & 511) - 1, 2);
            } else {
                const int v1 = s_cl[(il + 1) * 64 + lane];
                const int dY = (v1 >> 9) - (v0 >> 9);  // > 0
                const int num = ((v0 & 511) - 1) * dY + ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));  // (XL - 1) * dY
                xmin = ceil_div32(num, 2 * dY);
            }
        }
        {
            const int v0 = s_cr[ir * 64 + lane];
            if ((v0 >> 9) == Yr || ir + 1 >= nr) {
                xmax = floor_div32((v0 & 511) - 1, 2);
            } else {
                const int v1 = s_cr[(ir + 1) * 64 + lane];
                const int dY = (v1 >> 9) - (v0 >> 9);
                const int num = ((v0 & 511) - 1) * dY + ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));
                xmax = floor_div32(num, 2 * dY);
            }
        }
        if (xmax >= xmin) count += xmax - xmin + 1;
    }
    trow[AMT_RP_AREA_CONVEX] = (double)count;
}

// The same, TWO lanes per label (round 3): the kernel above is a chain of dependent instructions per lane -- ~100 steps in
// Y with two monotone stacks each -- on ~1,000 waves per 48 planes, one per SIMD, nothing to hide a latency behind (222 us).
// The left and the right chain are independent: lane 2 i builds the left one, lane 2 i + 1 the right one (a pop is
// "cr <= 0" on the left, "cr >= 0" on the right), each interpolates its own bound per row, the pair exchanges bounds by a
// lane swap; half the LDS per wave, twice the waves.
__global__ void __launch_bounds__(64) rp_hull_lds2_kernel(const int* __restrict__ bbox, const int* __restrict__ hoff,
                                                          const int* __restrict__ htot, const int2* __restrict__ rows,
                                                          size_t cap, double* __restrict__ table, int max_label) {
    __shared__ unsigned short s_rows[HULL_HMAX * 32];
    __shared__ unsigned short s_ch[HULL_CH * 64];
    const int plane = blockIdx.y;
    const int lane = threadIdx.x, side = lane & 1, pr = lane >> 1;
    const int l = blockIdx.x * 32 + pr;
    if (l >= max_label) return;
    const size_t li = (size_t)plane * max_label + l;
    double* trow = table + li * AMT_RP_NCOLS;
    const int miny = bbox[li * 4 + 0], x0 = bbox[li * 4 + 1], maxy = bbox[li * 4 + 2], x1 = bbox[li * 4 + 3];
    if (maxy < miny) {
        if (side == 0) trow[AMT_RP_AREA_CONVEX] = 0.0;
        return;
    }
    const int h = maxy - miny + 1;
    if (h > HULL_HMAX || x1 - x0 + 1 > HULL_WMAX) return;  // rp_hull_kernel takes it
    const size_t off = (size_t)hoff[li];
    if ((size_t)htot[plane] > cap || off + (size_t)h > cap) {  // capacity exceeded (fragmented labels)
        if (side == 0) trow[AMT_RP_AREA_CONVEX] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const int2* r = rows + (size_t)plane * cap + off;
    // the pair shares the loading: lane `side` takes the rows k = side (mod 2), eight per round trip
    for (int k0 = side; k0 < h; k0 += 16) {
        int2 e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = r[min(k0 + 2 * u, h - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 2 * u < h)  // an empty row (max < min) keeps that property: 255 | 0 << 8
                s_rows[(k0 + 2 * u) * 32 + pr] =
                    e[u].y >= e[u].x ? (unsigned short)((e[u].x - x0) | ((e[u].y - x0) << 8)) : (unsigned short)255;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // doubled, box-relative coordinates: Yr = Y - (2 miny - 1) in [0, 2h], Xr = X - (2 x0 - 1) in [0, 2w]
    int nc = 0;
    for (int Yr = 0; Yr <= 2 * h; ++Yr) {
        int mn = 0x7fffffff, mx = -1;
        if (!(Yr & 1)) {  // odd Y: shared by the rows above and below
            const int ka = (Yr >> 1) - 1, kb = Yr >> 1;
            if (ka >= 0) {
                const unsigned e = s_rows[ka * 32 + pr];
                if ((e >> 8) >= (e & 255)) {
                    mn = min(mn, 2 * (int)(e & 255) + 1);
                    mx = max(mx, 2 * (int)(e >> 8) + 1);
                }
            }
            if (kb < h) {
                const unsigned e = s_rows[kb * 32 + pr];
                if ((e >> 8) >= (e & 255)) {
                    mn = min(mn, 2 * (int)(e & 255) + 1);
                    mx = max(mx, 2 * (int)(e >> 8) + 1);
                }
            }
        } else {
            const unsigned e = s_rows[(Yr >> 1) * 32 + pr];
            if ((e >> 8) >= (e & 255)) {
                mn = 2 * (int)(e & 255);
                mx = 2 * (int)(e >> 8) + 2;
            }
        }
        if (mx < mn) continue;  // no pixel of this label contributes at this Y
        const int val = side ? mx : mn;
        while (nc >= 2) {
            const int v0 = s_ch[(nc - 2) * 64 + lane], v1 = s_ch[(nc - 1) * 64 + lane];
            const int cr = ((v1 >> 9) - (v0 >> 9)) * (val - (v0 & 511)) - ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));
            if (side ? cr >= 0 : cr <= 0) --nc; else break;
        }
        s_ch[(nc++) * 64 + lane] = (unsigned short)((Yr << 9) | val);
    }
    // count pixel centres: row k sits at Yr = 2k + 1; pixel x at Xr = 2 (x - x0) + 1
    int count = 0;
    int ic = 0;
    for (int k = 0; k < h; ++k) {
        const int Yr = 2 * k + 1;
        while (ic + 1 < nc && (int)(s_ch[(ic + 1) * 64 + lane] >> 9) <= Yr) ++ic;
        int b;  // this lane's bound of the pixel index: XL <= 2 x + 1 (left, rounded up) or 2 x + 1 <= XR (right, down)
        const int v0 = s_ch[ic * 64 + lane];
        if ((v0 >> 9) == Yr || ic + 1 >= nc) {
            b = side ? floor_div32((v0 & 511) - 1, 2) : ceil_div32((v0 & 511) - 1, 2);
        } else {
            const int v1 = s_ch[(ic + 1) * 64 + lane];
            const int dY = (v1 >> 9) - (v0 >> 9);  // > 0
            const int num = ((v0 & 511) - 1) * dY + ((v1 & 511) - (v0 & 511)) * (Yr - (v0 >> 9));  // (X - 1) * dY
            b = side ? floor_div32(num, 2 * dY) : ceil_div32(num, 2 * dY);
        }
        const int other = __shfl_xor(b, 1);
        const int xmin = side ? other : b, xmax = side ? b : other;
        if (xmax >= xmin) count += xmax - xmin + 1;
    }
    if (side == 0) trow[AMT_RP_AREA_CONVEX] = (double)count;
}

// ---- final per-label columns ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rp_final_kernel(const u64* __restrict__ acc, const int* __restrict__ bbox,
                                                       double* __restrict__ table, size_t nlab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nlab; i += (size_t)gridDim.x * 256) {
        const u64* A = acc + i * A_NACC;
        const int* B = bbox + i * 4;
        double* t = table + i * AMT_RP_NCOLS;
        const u64 n = A[A_N];
        if (n == 0) {
            for (int k = 0; k < AMT_RP_NCOLS; ++k)
                if (k != AMT_RP_AREA_CONVEX) t[k] = 0.0;
            continue;
        }
        const double dn = (double)n;
        t[AMT_RP_AREA] = dn;
        t[AMT_RP_CENTROID_Y] = (double)A[A_SY] / dn;
        t[AMT_RP_CENTROID_X] = (double)A[A_SX] / dn;
        t[AMT_RP_BBOX_Y0] = (double)B[0];
        t[AMT_RP_BBOX_X0] = (double)B[1];
        t[AMT_RP_BBOX_Y1] = (double)(B[2] + 1);
        t[AMT_RP_BBOX_X1] = (double)(B[3] + 1);
        const double SQ2 = 1.4142135623730951;
        t[AMT_RP_PERIMETER] = (double)A[A_C1] + (double)A[A_C2] * SQ2 + (double)A[A_C3] * ((1.0 + SQ2) / 2.0);
        // n * mu20 etc. as exact integers (translation invariant)
        const double Nyy = diff_of_products(n, A[A_SYY], A[A_SY], A[A_SY]);  // n*Syy - Sy^2 = n * mu20
        const double Nxx = diff_of_products(n, A[A_SXX], A[A_SX], A[A_SX]);  // n * mu02
        const double Nxy = diff_of_products(n, A[A_SXY], A[A_SX], A[A_SY]);  // n * mu11
        const double n2 = dn * dn;
        const double a = Nxx / n2;   // T[0,0] = mu02 / mu00
        const double c = Nyy / n2;   // T[1,1] = mu20 / mu00
        const double b = -Nxy / n2;  // T[0,1] = -mu11 / mu00
        const double tr = (a + c) * 0.5;
        const double hd = (a - c) * 0.5;
        const double rad = sqrt(hd * hd + b * b);
        double l1 = tr + rad, l2 = tr - rad;
        l1 = l1 < 0.0 ? 0.0 : l1;
        l2 = l2 < 0.0 ? 0.0 : l2;
        t[AMT_RP_AXIS_MAJOR] = 4.0 * sqrt(l1);
        t[AMT_RP_AXIS_MINOR] = 4.0 * sqrt(l2);
        t[AMT_RP_ECCENTRICITY] = l1 == 0.0 ? 0.0 : sqrt(1.0 - l2 / l1);
        double orient;
        if (Nxx == Nyy) {  // a - c == 0, decided on the exact integer numerators
            orient = (Nxy > 0.0) ? -0.78539816339744828 : 0.78539816339744828;  // b < 0 <=> mu11 > 0
        } else {
            orient = 0.5 * atan2(-2.0 * b, c - a);
        }
        t[AMT_RP_ORIENTATION] = orient;
    }
}

__global__ void __launch_bounds__(256) rp_solidity_kernel(double* __restrict__ table, size_t nlab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nlab; i += (size_t)gridDim.x * 256) {
        double* t = table + i * AMT_RP_NCOLS;
        t[AMT_RP_SOLIDITY] = t[AMT_RP_AREA] > 0.0 ? t[AMT_RP_AREA] / t[AMT_RP_AREA_CONVEX] : 0.0;
    }
}

// want_morph: fill the AMT_RP_* table; itable != NULL: fill {mean, max, min, std} per label and channel
static int regionprops_common(amt_ctx* ctx, const int32_t* labels, const uint16_t* intensity, int C, double* table_dev,
                              double* itable_dev, int nplanes, int H, int W, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    const bool want_morph = table_dev != nullptr;
    if (nplanes == 0 || max_label == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const size_t nlab = (size_t)nplanes * max_label;
    const size_t cap = want_morph ? n : 1;  // row-extent entries per plane (sum of bbox heights)
    size_t need = amt_align(nlab * A_NACC * 8) + amt_align(nlab * 16) + amt_align(nlab * 4) + amt_align(nplanes * 4) +
                  amt_align((size_t)nplanes * cap * 8) + 2 * amt_align((size_t)nplanes * 3 * cap * 8);
    AMT_TRY(amt_arena_begin(ctx, need));
    u64* acc = arena_take_t<u64>(ctx, nlab * A_NACC);
    int* bbox = arena_take_t<int>(ctx, nlab * 4);
    int* hoff = arena_take_t<int>(ctx, nlab);
    int* htot = arena_take_t<int>(ctx, nplanes);
    int2* rows = arena_take_t<int2>(ctx, (size_t)nplanes * cap);
    int2* chainL = arena_take_t<int2>(ctx, (size_t)nplanes * 3 * cap);
    int2* chainR = arena_take_t<int2>(ctx, (size_t)nplanes * 3 * cap);
    hipLaunchKernelGGL(rp_init_kernel, dim3(amt_grid_for(nlab, 256, 1024)), dim3(256), 0, ctx->stream, acc, bbox, nlab);
    AMT_LAUNCH_CHECK();
    if (want_morph) {  // the perimeter pass stages every label tile anyway: it folds the bounding boxes too
        const int nstrips = (W + PR_IN - 1) / PR_IN;
        dim3 gper((nstrips + 3) / 4, (H + PR_ROWS - 1) / PR_ROWS, nplanes);
        hipLaunchKernelGGL(rp_perimeter_rows_kernel, gper, dim3(256), 0, ctx->stream, labels, acc, H, W, max_label, bbox);
    } else {
        hipLaunchKernelGGL(rp_bbox_kernel, dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream,
                           labels, bbox, H, W, max_label);
    }
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(rp_heights_kernel, dim3(amt_grid_for(max_label, 256, 256), nplanes), dim3(256), 0, ctx->stream,
                       bbox, hoff, max_label);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_scan_excl(ctx, hoff, max_label, (size_t)max_label, htot, nplanes));
    // per-label scan: moments + row extents on the first call, intensity channels in groups of RP_MAXC
    const int groups = intensity ? (C + RP_MAXC - 1) / RP_MAXC : 1;
    for (int g = 0; g < groups; ++g) {
        const int c0 = g * RP_MAXC;
        const int nc = intensity ? ((C - c0) < RP_MAXC ? (C - c0) : RP_MAXC) : 0;
        hipLaunchKernelGGL(rp_label_kernel, dim3(max_label, nplanes), dim3(64), 0, ctx->stream, labels, bbox, hoff, htot,
                           rows, cap, acc, intensity, C, c0, nc, intensity ? itable_dev : (double*)nullptr, H, W,
                           max_label, (want_morph && g == 0) ? 1 : 0);
        AMT_LAUNCH_CHECK();
    }
    if (want_morph) {
        hipLaunchKernelGGL(rp_final_kernel, dim3(amt_grid_for(nlab, 256, 1024)), dim3(256), 0, ctx->stream, acc, bbox,
                           table_dev, nlab);
        AMT_LAUNCH_CHECK();
        const int skip_h = HULL_HMAX;
        {
            static const bool two = !(getenv("AMT_RP_HULL2") && getenv("AMT_RP_HULL2")[0] == '0');  // A/B switch; same results
            if (two)
                hipLaunchKernelGGL(rp_hull_lds2_kernel, dim3((max_label + 31) / 32, nplanes), dim3(64), 0, ctx->stream, bbox,
                                   hoff, htot, rows, cap, table_dev, max_label);
            else
                hipLaunchKernelGGL(rp_hull_lds_kernel, dim3((max_label + 63) / 64, nplanes), dim3(64), 0, ctx->stream, bbox,
                                   hoff, htot, rows, cap, table_dev, max_label);
            AMT_LAUNCH_CHECK();
        }
        // labels taller than HULL_HMAX rows or wider than HULL_WMAX columns: chains in HBM scratch
        hipLaunchKernelGGL(rp_hull_kernel, dim3((max_label + 63) / 64, nplanes), dim3(64), 0, ctx->stream, bbox, hoff,
                           htot, rows, chainL, chainR, cap, table_dev, max_label, skip_h);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(rp_solidity_kernel, dim3(amt_grid_for(nlab, 256, 1024)), dim3(256), 0, ctx->stream,
                           table_dev, nlab);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

extern "C" int amt_regionprops(amt_ctx* ctx, const int32_t* labels, double* table_dev, int nplanes, int H, int W,
                               int max_label) {
    AMT_REQUIRE(labels && table_dev && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0, "regionprops: bad arguments");
    return regionprops_common(ctx, labels, nullptr, 0, table_dev, nullptr, nplanes, H, W, max_label);
}

extern "C" int amt_regionprops_intensity_u16(amt_ctx* ctx, const int32_t* labels, const uint16_t* intensity, int C,
                                             double* table_dev, int nplanes, int H, int W, int max_label) {
    AMT_REQUIRE(labels && intensity && table_dev && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0 && C >= 1,
                "regionprops_intensity: bad arguments");
    return regionprops_common(ctx, labels, intensity, C, nullptr, table_dev, nplanes, H, W, max_label);
}

extern "C" int amt_regionprops_full_u16(amt_ctx* ctx, const int32_t* labels, const uint16_t* intensity, int C,
                                        double* table_dev, double* itable_dev, int nplanes, int H, int W,
                                        int max_label) {
    AMT_REQUIRE(labels && intensity && table_dev && itable_dev && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0 &&
                    C >= 1,
                "regionprops_full: bad arguments");
    return regionprops_common(ctx, labels, intensity, C, table_dev, itable_dev, nplanes, H, W, max_label);
}

// ---- bounding boxes only (what the outline extractor needs, R/masks.py:99) ---------------------------
__global__ void __launch_bounds__(256) bbox_init_kernel(int* __restrict__ bbox, size_t nlab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nlab; i += (size_t)gridDim.x * 256) {
        bbox[i * 4 + 0] = 0x7fffffff;
        bbox[i * 4 + 1] = 0x7fffffff;
        bbox[i * 4 + 2] = -1;
        bbox[i * 4 + 3] = -1;
    }
}

// ---- intensity statistics of float64 images (R/masks.py:319-323 accepts any 2-D ndarray as intensity image) -------
// One wave per label over its bounding box, two sweeps as np.mean / np.std make them: the mean first, then the mean
// of the squared deviations (population std).  float64 sums depend on the order of addition: results
// agree with numpy's pairwise sums to ~1e-15 relative, not bit for bit (the uint16 entry point is exact).
__global__ void __launch_bounds__(64) rp_intensity_f64_kernel(const int* __restrict__ labels, const int* __restrict__ bbox,
                                                              const double* __restrict__ inten, int C,
                                                              double* __restrict__ itable, int H, int W, int max_label) {
    const int plane = blockIdx.y, l = blockIdx.x, lane = threadIdx.x;
    const size_t li = (size_t)plane * max_label + l;
    const int y0 = bbox[li * 4 + 0], x0 = bbox[li * 4 + 1], y1 = bbox[li * 4 + 2], x1 = bbox[li * 4 + 3];
    double* out = itable + li * (size_t)C * 4;
    if (y1 < y0) {
        for (int i = lane; i < C * 4; i += 64) out[i] = 0.0;
        return;
    }
    const size_t n = (size_t)H * W;
    const int* lab = labels + (size_t)plane * n;
    auto wsum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    for (int c = 0; c < C; ++c) {
        const double* img = inten + ((size_t)plane * C + c) * n;
        double s = 0.0, cnt = 0.0, mn = __longlong_as_double(0x7ff0000000000000ll), mx = -mn;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0 + lane; x <= x1; x += 64)
                if (lab[(size_t)y * W + x] == l + 1) {
                    const double v = img[(size_t)y * W + x];
                    s += v;
                    cnt += 1.0;
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
        s = wsum(s);
        cnt = wsum(cnt);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        const double mean = s / cnt;
        double q = 0.0;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0 + lane; x <= x1; x += 64)
                if (lab[(size_t)y * W + x] == l + 1) {
                    const double d = img[(size_t)y * W + x] - mean;
                    q += d * d;
                }
        q = wsum(q);
        if (lane == 0) {
            out[c * 4 + 0] = mean;
            out[c * 4 + 1] = mx;
            out[c * 4 + 2] = mn;
            out[c * 4 + 3] = sqrt(q / cnt);
        }
    }
}

extern "C" int amt_regionprops_intensity_f64(amt_ctx* ctx, const int32_t* labels, const double* intensity, int C,
                                             double* table_dev, int nplanes, int H, int W, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && intensity && table_dev && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0 && C >= 1,
                "regionprops_intensity_f64: bad arguments");
    if (nplanes == 0 || max_label == 0) return AMT_OK;
    const size_t nlab = (size_t)nplanes * max_label;
    AMT_TRY(amt_arena_begin(ctx, amt_align(nlab * 16)));
    int* bbox = arena_take_t<int>(ctx, nlab * 4);
    hipLaunchKernelGGL(bbox_init_kernel, dim3(amt_grid_for(nlab, 256, 1024)), dim3(256), 0, ctx->stream, bbox, nlab);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(rp_bbox_kernel, dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream, labels, bbox,
                       H, W, max_label);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(rp_intensity_f64_kernel, dim3(max_label, nplanes), dim3(64), 0, ctx->stream, labels, bbox, intensity, C,
                       table_dev, H, W, max_label);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_label_bboxes(amt_ctx* ctx, const int32_t* labels, int32_t* bbox_dev, int nplanes, int H, int W,
                                int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && bbox_dev && nplanes >= 0 && H > 0 && W > 0 && max_label >= 1, "label_bboxes: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t nlab = (size_t)nplanes * max_label;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(amt_grid_for(nlab, 256, 1024)), dim3(256), 0, ctx->stream, bbox_dev, nlab);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(rp_bbox_kernel, dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream, labels,
                       bbox_dev, H, W, max_label);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
