// Separable Gaussian / difference-of-Gaussians and the elementwise ops of R/operations.py.
//
// Numerics contract: float64 everywhere, NO fma contraction (the library is
// built with -ffp-contract=off), per-sample evaluation order of scipy's symmetric correlate1d:
//     acc = x[i]*w[c];  for j = r, r-1, ..., 1:  acc += (x[i-j] + x[i+j]) * w[c-j]
// axis 0 first, its float64 result is the input of axis 1.  The weights arrive from the host
// (numpy), so np.exp rounding is shared with the CPU path.
#include "amt_internal.h"

template <typename TIn>
__device__ __forceinline__ double load_as_f64(const TIn* p, size_t i, double scale);
template <>
__device__ __forceinline__ double load_as_f64<uint16_t>(const uint16_t* p, size_t i, double scale) {
    return (double)p[i] * scale;
}
template <>
__device__ __forceinline__ double load_as_f64<double>(const double* p, size_t i, double scale) {
    return p[i];
}

// ------------------------------------------------------------------------------------------------
// Register-blocked two-pass path for large radii (13 <= r <= 128, e.g. sigma = 16 of subtract_background_dog).
// A thread produces EIGHT consecutive outputs along the filter axis.  scipy's order adds the pairs from the
// outside in: at distance j the eight outputs need x[o+k-j] and x[o+k+j], k = 0..7 -- two windows of eight
// samples that each slide by ONE sample when j decreases.  So a step costs 2 new LDS reads (instead of 16) and
// 24 float64 operations; the windows rotate by renaming (the loop is unrolled over 8 steps), the weights are
// wave-uniform scalar loads.  LDS traffic drops 8x and the passes become fp64-issue bound.
//   vertical   : lanes = 64 columns, the tile keeps the RAW input type (uint16 tiles are 4x smaller);
//   horizontal : lanes = 64 rows (row pitch odd in doubles: conflict-free), results go back through LDS so
//                that the stores are coalesced along x.
// ------------------------------------------------------------------------------------------------
template <typename TIn>
__device__ __forceinline__ double cvt_f64(TIn v, double scale);
template <>
__device__ __forceinline__ double cvt_f64<uint16_t>(uint16_t v, double scale) {
    return (double)v * scale;
}
template <>
__device__ __forceinline__ double cvt_f64<double>(double v, double scale) {
    return v;
}

// RUNTIME radius.  Eight outputs at positions o .. o+7 of a line; `at(i)` returns sample i (already converted); `w` = the
// 2r+1 weights IN LDS (scalar loads share their counter with LDS reads and return out of order, so a scalar
// weight load inside the steps forces a full wait on the in-flight LDS reads; LDS weight reads stay in order).
// The two samples that enter the windows are requested TWO steps ahead (software pipeline), so the LDS latency
// hides behind 48 float64 operations even with a single wave per SIMD.
template <typename F>
__device__ __forceinline__ void blocked8_rot(F at, int o, const double* w, int r, double acc[8]) {
    double L[8], Rw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc[k] = at(o + k) * w[r];
        L[k] = at(o + k - r);
        Rw[k] = at(o + k + r);
    }
    double pl = at(o + 8 - r), pr = at(o + r - 1);  // enter after the first step
    // one step at distance j = r - st; s = st & 7 selects the register renaming of the rotating windows
#define AMT_BLOCKED8_STEP(s, st)                                                                   \
    {                                                                                              \
        const int j = r - (st);                                                                    \
        const double nl = at(o + 9 - j), nr = at(o + j - 2); /* enter after the NEXT step */        \
        const double wj = w[st];                                                                   \
        double t_[8];                                                                              \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) t_[k] = L[(k + (s)) & 7] + Rw[(k - (s)) & 7]; \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) t_[k] = t_[k] * wj;                          \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) acc[k] += t_[k];                             \
        L[(s) & 7] = pl;          /* the left window drops its first sample and gains x[o+8-j] */   \
        Rw[(7 - (s)) & 7] = pr;   /* the right window drops its last sample and gains x[o+j-1] */   \
        pl = nl;                                                                                   \
        pr = nr;                                                                                   \
    }
    int st = 0;
    for (; st + 8 <= r; st += 8) {  // full blocks of eight steps: no guards, free scheduling
        AMT_BLOCKED8_STEP(0, st + 0)
        AMT_BLOCKED8_STEP(1, st + 1)
        AMT_BLOCKED8_STEP(2, st + 2)
        AMT_BLOCKED8_STEP(3, st + 3)
        AMT_BLOCKED8_STEP(4, st + 4)
        AMT_BLOCKED8_STEP(5, st + 5)
        AMT_BLOCKED8_STEP(6, st + 6)
        AMT_BLOCKED8_STEP(7, st + 7)
    }
    if (st + 0 < r) AMT_BLOCKED8_STEP(0, st + 0)
    if (st + 1 < r) AMT_BLOCKED8_STEP(1, st + 1)
    if (st + 2 < r) AMT_BLOCKED8_STEP(2, st + 2)
    if (st + 3 < r) AMT_BLOCKED8_STEP(3, st + 3)
    if (st + 4 < r) AMT_BLOCKED8_STEP(4, st + 4)
    if (st + 5 < r) AMT_BLOCKED8_STEP(5, st + 5)
    if (st + 6 < r) AMT_BLOCKED8_STEP(6, st + 6)
#undef AMT_BLOCKED8_STEP
}

// COMPILE-TIME radius RC (the reference's own preprocessing: sigma 16 -> r = 64), everything unrolled.
// Steps come in blocks of eight.  Within a block the left window at step s is the last 8 - s samples of the block's
// OLD eight samples followed by the first s of its NEW eight (the right window mirrors that), so output k reads
// old[k + s] or new[k + s - 8] -- static register names, nothing is inserted or rotated.  The next block's NEW samples
// are requested one block ahead, each into the register its predecessor vacated (old[s] is dead after step s): a
// request is ~800 clocks old when it is used and no value is ever copied.  Round 2's form kept ONE rotating window and
// replaced all sixteen live slots in every block; the compiler paid for that with 18 register moves per 192 float64
// operations at the loop's back edge (ISA of conv_h8g_kernel) plus waits on samples requested two steps ahead.
// With the block loop unrolled completely no value crosses a back edge; with a runtime radius the same source needs
// 160 registers and is slower than the rotating form above (measured), so that one keeps serving the other radii.
// Measured (32 planes of 2048^2, r = 64): vertical 1,130 -> 1,076 us, horizontal 1,300 -> 1,241 us.
template <int RC, typename F>
__device__ __forceinline__ void blocked8(F at, int o, const double* w, int r_, double acc[8]) {
    if constexpr (RC == 0) {
        blocked8_rot(at, o, w, r_, acc);
        return;
    }
    constexpr int r = RC ? RC : 16;
    static_assert(RC == 0 || RC >= 8, "block-ahead requests stay inside the chunk's samples for r >= 8 only");
    double LA[8], RA[8], LB[8], RB[8];
    // every sample index is written as b + a non-negative constant, b = o - r made opaque: LDS instructions take
    // unsigned immediate offsets only, and the compiler otherwise spends a vector add on every request left of o
    int b = o - r;
    asm volatile("" : "+v"(b));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc[k] = at(b + r + k) * w[r];
        LA[k] = at(b + k);
        RA[k] = at(b + 2 * r + k);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        LB[k] = at(b + k + 8);
        RB[k] = at(b + 2 * r + k - 8);
    }
    // neither the optimiser nor the scheduler may move an LDS read or a step's arithmetic across a step boundary: the
    // (empty) statement orders memory operations and "rewrites" the eight sums, so each step's operations sit between
    // two of them.  Left alone, instruction selection puts every request of an unrolled block first and the arithmetic
    // after (458 - 512 registers, spills)
#define AMT_B8_FENCE                                                                                       \
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),    \
                      "+v"(acc[6]), "+v"(acc[7])::"memory");                                                \
    __builtin_amdgcn_sched_barrier(0);
    // one step at distance j = r - st, s = st & 7; Lo / Ro: the block's old samples, Ln / Rn: its new ones
#define AMT_B8_STEP(Lo, Ln, Ro, Rn, s, st)                                                            \
    {                                                                                                 \
        const double wj = wnext;                                                                      \
        wnext = w[(st) + 1]; /* asked for a step ahead (w[r] exists: the centre weight) */            \
        double t_[8];                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                 \
            t_[k] = (k + (s) < 8 ? Lo[(k + (s)) & 7] : Ln[(k + (s)) & 7]) +                            \
                    (k - (s) >= 0 ? Ro[(k - (s)) & 7] : Rn[(k - (s)) & 7]);                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) t_[k] = t_[k] * wj;                             \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) acc[k] += t_[k];                                \
        AMT_B8_FENCE /* requests stay where the pipeline puts them (register pressure) */              \
    }
    // after step s of the block that starts at st0: old[s] (left) and old[7 - s] (right) are dead and receive the
    // samples the block after the next one calls new
#define AMT_B8_FETCH(Lo, Ro, s, st0)                                                                  \
    {                                                                                                 \
        Lo[s] = at(b + (st0) + 16 + (s));                                                               \
        Ro[7 - (s)] = at(b + 2 * r - (st0) - 16 + 7 - (s));                                               \
        AMT_B8_FENCE                                                                                  \
    }
#define AMT_B8_BLOCK(Lo, Ln, Ro, Rn, st0)                                                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 0, (st0) + 0) AMT_B8_FETCH(Lo, Ro, 0, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 1, (st0) + 1) AMT_B8_FETCH(Lo, Ro, 1, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 2, (st0) + 2) AMT_B8_FETCH(Lo, Ro, 2, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 3, (st0) + 3) AMT_B8_FETCH(Lo, Ro, 3, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 4, (st0) + 4) AMT_B8_FETCH(Lo, Ro, 4, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 5, (st0) + 5) AMT_B8_FETCH(Lo, Ro, 5, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 6, (st0) + 6) AMT_B8_FETCH(Lo, Ro, 6, st0)                             \
    AMT_B8_STEP(Lo, Ln, Ro, Rn, 7, (st0) + 7) AMT_B8_FETCH(Lo, Ro, 7, st0)
#define AMT_B8_TAIL(Lo, Ln, Ro, Rn, st0)                                                              \
    if ((st0) + 0 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 0, (st0) + 0)                                       \
    if ((st0) + 1 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 1, (st0) + 1)                                       \
    if ((st0) + 2 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 2, (st0) + 2)                                       \
    if ((st0) + 3 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 3, (st0) + 3)                                       \
    if ((st0) + 4 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 4, (st0) + 4)                                       \
    if ((st0) + 5 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 5, (st0) + 5)                                       \
    if ((st0) + 6 < r) AMT_B8_STEP(Lo, Ln, Ro, Rn, 6, (st0) + 6)
    // a block that starts at st0 fetches x[o - r + st0 + 16 .. + 23] and x[o + r - st0 - 16 .. - 9]: with st0 + 8 <= r
    // (and r >= 8, this path's radii start at 13) every fetched index lies in [o - 8, o + 15], inside the samples the
    // eight outputs read anyway
    int st = 0;
    double wnext = w[0];
#pragma unroll
    for (; st + 16 <= r; st += 16) {
        AMT_B8_BLOCK(LA, LB, RA, RB, st)
        AMT_B8_BLOCK(LB, LA, RB, RA, st + 8)
    }
    if (st + 8 <= r) {
        AMT_B8_BLOCK(LA, LB, RA, RB, st)
        st += 8;
        AMT_B8_TAIL(LB, LA, RB, RA, st)
    } else {
        AMT_B8_TAIL(LA, LB, RA, RB, st)
    }
#undef AMT_B8_STEP
#undef AMT_B8_FENCE
#undef AMT_B8_FETCH
#undef AMT_B8_BLOCK
#undef AMT_B8_TAIL
}

template <typename TIn>
__global__ void __launch_bounds__(256) conv_v8_kernel(const TIn* __restrict__ in, double scale,
                                                      double* __restrict__ out, int H, int W,
                                                      const double* __restrict__ wts, int r, int mode, double cval,
                                                      int TH, size_t in_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int rows = TH + 2 * r;
    TIn* tile = reinterpret_cast<TIn*>(smem_raw);                            // rows x 64
    double* wl = reinterpret_cast<double*>(smem_raw + (((size_t)rows * 64 * sizeof(TIn) + 7) & ~(size_t)7));  // 2r + 1
    unsigned char* inside = reinterpret_cast<unsigned char*>(wl + 2 * r + 1);  // rows: 0 = cval row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const TIn* src = in + (size_t)blockIdx.z * in_stride;
    const int xc = x < W ? x : W - 1;
    int any_out = 0;
    for (int i = threadIdx.x; i < 2 * r + 1; i += 256) wl[i] = wts[i];
    for (int k0 = wave * 8; k0 < rows; k0 += 32) {
        TIn v[8];
        int yy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            yy[u] = k0 + u < rows ? amt_map_index(y0 - r + k0 + u, H, mode) : -1;
            v[u] = yy[u] >= 0 ? src[(size_t)yy[u] * W + xc] : (TIn)0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < rows) {
                tile[(k0 + u) * 64 + lane] = v[u];
                if (lane == 0) inside[k0 + u] = yy[u] >= 0 ? 1 : 0;
                any_out |= yy[u] < 0 ? 1 : 0;
            }
    }
    const int has_out = __syncthreads_or(any_out);  // only 'constant' mode tiles at the image border
    auto at = [&](int i) -> double {  // i = tile row
        return inside[i] ? cvt_f64<TIn>(tile[i * 64 + lane], scale) : cval;
    };
    auto at_in = [&](int i) -> double { return cvt_f64<TIn>(tile[i * 64 + lane], scale); };
    for (int c = wave; c * 8 < TH; c += 4) {
        const int q0 = c * 8;  // first output row of the chunk (relative to y0)
        if (y0 + q0 >= H) break;
        double acc[8];
        if (has_out)
            blocked8_rot(at, q0 + r, wl, r, acc);
        else
            blocked8_rot(at_in, q0 + r, wl, r, acc);
        if (x < W) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (y0 + q0 + k < H) out[plane + (size_t)(y0 + q0 + k) * W + x] = acc[k];
        }
    }
}

// `minuend` (nullable; may alias out): the kernel stores minuend - result instead of the result -- the second Gaussian
// of a difference of Gaussians subtracts itself from the first in its own epilogue (no third pass over the planes)
__global__ void __launch_bounds__(256) conv_h8_kernel(const double* __restrict__ in, double* out, int H, int W,
                                                      const double* __restrict__ wts, int r, int mode, double cval,
                                                      int TW, const double* minuend) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int cols = TW + 2 * r;
    const int pitch = cols | 1;  // odd number of doubles per row
    double* tile = reinterpret_cast<double*>(smem_raw);  // 64 x pitch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * TW;
    const int y0 = blockIdx.y * 64;
    const size_t plane = (size_t)blockIdx.z * H * W;
    // boundary-mapped source column of every tile column, computed once (one call site of the general mapping)
    double* wl = tile + (size_t)64 * pitch;  // 2r + 1
    int* xmap = reinterpret_cast<int*>(wl + 2 * r + 1);
    for (int i = threadIdx.x; i < 2 * r + 1; i += 256) wl[i] = wts[i];
    for (int k = threadIdx.x; k < cols; k += 256) xmap[k] = amt_map_index(x0 - r + k, W, mode);
    __syncthreads();
    // staging: wave w owns rows w, w + 4, ...; 64 consecutive columns of 8 rows in flight per step
    for (int k0 = 0; k0 < cols; k0 += 64) {
        const int k = k0 + lane;
        const int xx = k < cols ? xmap[k] : -1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = wave + 4 * (half * 8 + u);
                v[u] = (xx >= 0 && y0 + row < H) ? in[plane + (size_t)(y0 + row) * W + xx] : cval;
            }
            if (k < cols) {
#pragma unroll
                for (int u = 0; u < 8; ++u) tile[(wave + 4 * (half * 8 + u)) * pitch + k] = v[u];
            }
        }
    }
    __syncthreads();
    const double* myrow = tile + lane * pitch;
    auto at = [&](int i) -> double { return myrow[i]; };
    // every wave walks over its chunks of eight outputs; a lane stores its eight consecutive doubles (64 bytes,
    // whole sectors) straight from registers
    const int y = y0 + lane;
    for (int c = wave; c * 8 < TW; c += 4) {
        double res[8];
        blocked8_rot(at, c * 8 + r, wl, r, res);
        const int x = x0 + c * 8;
        if (y < H && x < W) {
            double* dst = out + plane + (size_t)y * W + x;
            const double* mn = minuend ? minuend + plane + (size_t)y * W + x : nullptr;
            if (x + 7 < W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                if (mn) {
                    double2 m2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) m2[k] = *reinterpret_cast<const double2*>(mn + 2 * k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        res[2 * k] = m2[k].x - res[2 * k];
                        res[2 * k + 1] = m2[k].y - res[2 * k + 1];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2*>(dst + k) = make_double2(res[k], res[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (x + k < W) dst[k] = mn ? mn[k] - res[k] : res[k];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fused single-kernel path for small radii (r <= 12): one HBM read of the input, one write of the
// result, no intermediate in HBM.
//   * a 256-thread block owns 256 - 2R output columns (+ R halo columns each side) and a chunk of rows;
//     thread t owns input column x0 - R + t and slides DOWN it four rows at a time: a register window of
//     2R+4 converted samples gives the axis-0 results of four rows per step (four new loads in flight,
//     every load coalesced across the wave);
//   * the axis-0 rows are collected four at a time in a double-buffered LDS group; after ONE barrier per
//     group, wave q finishes row q of the group: each lane reads 2R+4 neighbours (128-bit LDS reads) and
//     produces four adjacent outputs -- 5 LDS values per output instead of 2R+1, 32 contiguous bytes per
//     lane on the store side;
//   * both passes use scipy's order: centre tap first, then pairs outermost -> innermost;
//   * the weights are wave-uniform and stay in scalar registers; the boundary-mapped source rows are
//     computed once per block (one call site of the general index mapping keeps the code small enough
//     for the instruction cache);
//   * optionally the block folds min / max of its outputs into per-plane ordered keys (what Otsu's
//     histogram range needs), which saves a full re-read of the result.
// ------------------------------------------------------------------------------------------------
constexpr int FR_MAX = 12;

// one-instruction min / max (fmin / fmax add a canonicalisation per operand; the values here are never NaN unless
// the input is, in which case the histogram range is meaningless anyway)
__device__ __forceinline__ double vmin_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename TIn, int R>
__global__ void __launch_bounds__(256) gauss_fused_kernel(const TIn* __restrict__ in, double scale,
                                                          double* __restrict__ out, int H, int W,
                                                          const double* __restrict__ wts, int mode, double cval,
                                                          size_t in_stride, int TH,
                                                          unsigned long long* __restrict__ keys) {
    constexpr int K = 2 * R + 1;
    constexpr int OUTW = 256 - 2 * R;
    constexpr int NSEG = (OUTW + 3) / 4;
    __shared__ __attribute__((aligned(16))) double rowbuf[2][4][256 + 4];
    __shared__ int ymap[256 + 2 * FR_MAX + 4];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * OUTW;
    const int y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const TIn* src = in + (size_t)blockIdx.z * in_stride;
    double w[R + 1];  // w[j] = weight at distance j from the centre (uniform -> scalar loads)
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wts[R - j];
    const int xg = x0 - R + t;
    const int xm = amt_map_index(xg, W, mode);  // -1: outside in 'constant' mode
    for (int k = t; k < TH + 2 * R + 4; k += 256) ymap[k] = amt_map_index(y0 - R + k, H, mode);
    __syncthreads();
    // k = window row: image row y0 - R + k.  The load is UNCONDITIONAL from clamped coordinates and the conversion /
    // cval substitution happens after the whole group has been issued (anything computed next to a conditional
    // load puts a wait behind it and serialises the rows)
    const int xmc = xm < 0 ? 0 : xm;
    auto load_raw = [&](int k) -> TIn {
        const int yy = ymap[k];
        return src[(size_t)(yy < 0 ? 0 : yy) * W + xmc];
    };
    auto to_sample = [&](TIn raw, int k) -> double {
        return (ymap[k] < 0 || xm < 0) ? cval : load_as_f64<TIn>(&raw, 0, scale);
    };
    // axis-1 role of this thread: row q of a group, outputs 4 * seg .. 4 * seg + 3 of the block
    const int q = t >> 6, seg = t & 63;
    const int xo = x0 + 4 * seg;
    double vlo = __builtin_huge_val(), vhi = -__builtin_huge_val();  // folded to ordered keys at the end
    auto finish_group = [&](int buf, int rg, int nrows) {  // rg = first row (relative to y0) of the group
        if (q < nrows && seg < NSEG) {
            const double* c0 = &rowbuf[buf][q][4 * seg];
            double c[2 * R + 4];
#pragma unroll
            for (int i = 0; i < 2 * R + 4; ++i) c[i] = c0[i];
            double a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double a2 = c[i + R] * w[0];
#pragma unroll
                for (int j = R; j >= 1; --j) a2 += (c[i + R - j] + c[i + R + j]) * w[j];
                a[i] = a2;
            }
            double* dst = out + plane + (size_t)(y0 + rg + q) * W + xo;
            if (4 * seg + 3 < OUTW && xo + 3 < W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                reinterpret_cast<double2*>(dst)[0] = make_double2(a[0], a[1]);
                reinterpret_cast<double2*>(dst)[1] = make_double2(a[2], a[3]);
                if (keys) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        vlo = vmin_f64(vlo, a[i]);
                        vhi = vmax_f64(vhi, a[i]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (4 * seg + i < OUTW && xo + i < W) {
                        dst[i] = a[i];
                        if (keys) {
                            vlo = vmin_f64(vlo, a[i]);
                            vhi = vmax_f64(vhi, a[i]);
                        }
                    }
                }
            }
        }
    };
    // window of K + 3 samples: rows rg - R .. rg + R + 3 of the current group of four output rows
    double win[K + 3];
    {
        TIn raw[K - 1];
#pragma unroll
        for (int k = 0; k < K - 1; ++k) raw[k] = load_raw(k);
#pragma unroll
        for (int k = 0; k < K - 1; ++k) win[k] = to_sample(raw[k], k);
    }
    const int rows = (y0 + TH <= H) ? TH : (H - y0);
    for (int rg = 0; rg < rows; rg += 4) {
        TIn raw4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) raw4[i] = load_raw(rg + 2 * R + i);  // four loads in flight
#pragma unroll
        for (int i = 0; i < 4; ++i) win[K - 1 + i] = to_sample(raw4[i], rg + 2 * R + i);
        const int buf = (rg >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double acc = win[R + i] * w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) acc += (win[R + i - j] + win[R + i + j]) * w[j];
            // scipy extends the axis-0 RESULT with cval along axis 1 in 'constant' mode
            if (xm < 0) acc = cval;
            rowbuf[buf][i][t] = acc;
        }
        __syncthreads();
        finish_group(buf, rg, rows - rg < 4 ? rows - rg : 4);
#pragma unroll
        for (int k = 0; k < K - 1; ++k) win[k] = win[k + 4];
    }
    if (keys) {
        unsigned long long klo = amt_f64_key(vlo), khi = amt_f64_key(vhi);
        if (vlo > vhi) {  // this thread stored nothing
            klo = ~0ull;
            khi = 0ull;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(klo, off), h2 = __shfl_xor(khi, off);
            klo = l2 < klo ? l2 : klo;
            khi = h2 > khi ? h2 : khi;
        }
        if ((t & 63) == 0) {  // four pairs of atomics per block
            atomicMin(&keys[2 * blockIdx.z], klo);
            atomicMax(&keys[2 * blockIdx.z + 1], khi);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same fused Gaussian for uint16 input with the INPUT side decoupled from the waves' registers: the raw
// uint16 rows of the block's 256 columns go straight from HBM into an LDS ring with `global_load_lds_dwordx4`
// (16 bytes per lane, no VGPRs, issued three row-groups ahead by two of the four waves), and every thread picks
// its column's new samples out of LDS when a group starts.  HBM latency is then covered by three groups of
// arithmetic instead of by occupancy.  Conditions (else the register-load kernel above runs): W a multiple of 8
// and 16-byte aligned planes (so that tiles start on 16-byte chunks and no chunk straddles the image edge),
// boundary modes nearest / reflect / mirror (the extension of a column is another column of the same tile).
// Tile: 256 columns starting RP = roundup(R, 8) left of the first output; OUTW = 256 - 2 RP outputs.
// ------------------------------------------------------------------------------------------------
// EPI selects what happens to the finished samples (the arithmetic in front of it is the same code, so every variant
// produces the same float64 values bit for bit):
//   0  store them (and fold min / max into `keys` when given) -- amt_gaussian;
//   1  fold min / max into `keys` only -- first pass of amt_gaussian_otsu_codes: the float64 plane never exists;
//   2  second pass of amt_gaussian_otsu_codes: np.histogram bin b of every sample (edges = np.linspace(min, max, 257),
//      repaired against both edges exactly like hist_f64_kernel) counted into `hist`, and the uint16 code
//      2 b + (sample > centre of bin b) stored: for the Otsu threshold t = centre of bin k,
//      sample > t  <=>  code > 2 k  (a sample of a bin above k is >= its lower edge > centre_k, one of a bin below is
//      < centre_k, and inside bin k the low bit IS the comparison), so the mask chain continues from 2 bytes per pixel.
constexpr int GL_NBINS = 256;
template <int R, int EPI>
__global__ void __launch_bounds__(256) gauss_lds_kernel(const uint16_t* __restrict__ in, double scale,
                                                        double* __restrict__ out, int H, int W,
                                                        const double* __restrict__ wts, int mode, size_t in_stride,
                                                        int TH, unsigned long long* __restrict__ keys,
                                                        const double* __restrict__ minmax, uint32_t* __restrict__ hist,
                                                        uint16_t* __restrict__ codes) {
    constexpr int K = 2 * R + 1;
    constexpr int RP = (R + 7) & ~7;
    constexpr int OUTW = 256 - 2 * RP;
    constexpr int NSEG = OUTW / 4;
    constexpr int RING = 16;  // raw rows: four groups of four
    // ONE LDS object, carved by hand: with several __shared__ arrays the compiler cannot tell the LDS-DMA writes into
    // the ring from reads of the other arrays and drains the ring (vmcnt(0)) in front of every LDS read
    constexpr int ROWBUF_B = 2 * 4 * (256 + 4) * 8, RAW_B = RING * 256 * 2, YMAP_B = (256 + 2 * FR_MAX + 8) * 4;
    constexpr int EDGES_B = EPI == 2 ? ((GL_NBINS + 1) * 8 + 8) : 0, LH_B = EPI == 2 ? 4 * GL_NBINS * 4 : 0;
    __shared__ __attribute__((aligned(16))) char lds[ROWBUF_B + RAW_B + YMAP_B + EDGES_B + LH_B];
    double(*rowbuf)[4][256 + 4] = reinterpret_cast<double(*)[4][256 + 4]>(lds);
    unsigned short(*raw)[256] = reinterpret_cast<unsigned short(*)[256]>(lds + ROWBUF_B);
    int* ymap = reinterpret_cast<int*>(lds + ROWBUF_B + RAW_B);
    double* edges = reinterpret_cast<double*>(lds + ROWBUF_B + RAW_B + YMAP_B);
    uint32_t* lh = reinterpret_cast<uint32_t*>(lds + ROWBUF_B + RAW_B + YMAP_B + EDGES_B);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int x0 = blockIdx.x * OUTW;
    const int xt0 = x0 - RP;  // image column of tile column 0 (a multiple of 8)
    const int y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const uint16_t* src = in + (size_t)blockIdx.z * in_stride;
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wts[R - j];
    for (int k = t; k < TH + 2 * R + 8; k += 256) ymap[k] = amt_map_index(y0 - R + k, H, mode);
    double h_lo = 0.0, h_hi = 0.0, h_norm = 0.0;
    if (EPI == 2) {
        h_lo = minmax[2 * blockIdx.z];
        h_hi = minmax[2 * blockIdx.z + 1];
        const double step = (h_hi - h_lo) / (double)GL_NBINS;
        h_norm = (double)GL_NBINS / (h_hi - h_lo);
        for (int i = t; i <= GL_NBINS; i += 256) edges[i] = i == GL_NBINS ? h_hi : (double)i * step + h_lo;
        for (int i = t; i < 4 * GL_NBINS; i += 256) lh[i] = 0;
    }
    __syncthreads();
    // this thread's column: tile column t; outside the image it reads the tile column of its boundary-mapped image
    const int xm = amt_map_index(xt0 + t, W, mode);
    int ti = xm - xt0;  // in [0, 256) for every column that an output needs; the others are clamped (never used)
    ti = ti < 0 ? 0 : (ti > 255 ? 255 : ti);
    const int rows = (y0 + TH <= H) ? TH : (H - y0);
    const int ngroups = (rows + 3) >> 2;
    // loader role (waves 0 and 1): lane l moves the 16-byte chunk (l & 31) of row (2 * wave + (l >> 5)) of a group
    const int lrow = 2 * wave + (lane >> 5);
    int lcol = xt0 + 8 * (lane & 31);
    lcol = lcol < 0 ? 0 : (lcol > W - 8 ? W - 8 : lcol);  // chunks entirely outside the image are never read
    auto issue_group = [&](int g) {  // window rows 2R + 4g .. + 3 -> ring slots (4g .. 4g + 3) & 15
        if (wave < 2 && g < ngroups) {
            const int yy = ymap[2 * R + 4 * g + lrow];
            const uint16_t* gp = src + (size_t)yy * W + lcol;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)&raw[(4 * g + 2 * wave) & (RING - 1)][0],
                                             16, 0, 0);
        }
    };
    issue_group(0);
    issue_group(1);
    issue_group(2);
    // initial window: rows -R .. R-1 of the first output row, ordinary loads (once per block)
    double win[K + 3];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) win[k] = (double)src[(size_t)ymap[k] * W + xm] * scale;
    const int q = wave, seg = lane;
    const int xo = x0 + 4 * seg;
    double vlo = __builtin_huge_val(), vhi = -__builtin_huge_val();
    __builtin_amdgcn_s_waitcnt(0x0F72);  // vmcnt(2): group 0 has landed (groups 1, 2 may be in flight)
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        const int rg = g << 2;
        issue_group(g + 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) win[K - 1 + i] = (double)raw[(rg + i) & (RING - 1)][ti] * scale;
        const int buf = g & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double acc = win[R + i] * w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) acc += (win[R + i - j] + win[R + i + j]) * w[j];
            rowbuf[buf][i][t] = acc;
        }
        // Group g + 1 must have landed before the barrier that publishes it.  The vector-memory counter retires in
        // order, so a loader wave may leave outstanding exactly what it issued AFTER that load: the loads of groups
        // g+2 and g+3 and the two 16-byte stores of each of the groups g-2 and g-1 (waves 0 and 1 always own a row of a
        // full group).  Near the end of the block, where fewer loads are issued, it simply drains.
        if (wave < 2) {
            if (g + 3 >= ngroups)
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            else if (g >= 2)
                __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
            else if (g == 1)
                __builtin_amdgcn_s_waitcnt(0x0F74);  // vmcnt(4)
            else
                __builtin_amdgcn_s_waitcnt(0x0F72);  // vmcnt(2)
        }
        __syncthreads();
        const int nrows = rows - rg < 4 ? rows - rg : 4;
        if (q < nrows && seg < NSEG) {
            const double* c0 = &rowbuf[buf][q][RP - R + 4 * seg];
            double c[2 * R + 4];
#pragma unroll
            for (int i = 0; i < 2 * R + 4; ++i) c[i] = c0[i];
            double a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double a2 = c[i + R] * w[0];
#pragma unroll
                for (int j = R; j >= 1; --j) a2 += (c[i + R - j] + c[i + R + j]) * w[j];
                a[i] = a2;
            }
            // W and the tile origin are multiples of 8: the four outputs are inside the image together or not at
            // all, and the destination is 16-byte aligned -> exactly two store instructions per row
            if (EPI == 0) {
                double* dst = out + plane + (size_t)(y0 + rg + q) * W + xo;
                if (xo < W) {
                    reinterpret_cast<double2*>(dst)[0] = make_double2(a[0], a[1]);
                    reinterpret_cast<double2*>(dst)[1] = make_double2(a[2], a[3]);
                }
            }
            if (EPI <= 1 && keys && xo < W) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vlo = vmin_f64(vlo, a[i]);
                    vhi = vmax_f64(vhi, a[i]);
                }
            }
            if (EPI == 2 && xo < W) {
                unsigned cd[4];
                if (!(h_lo < h_hi)) {  // constant plane: every code 0, the threshold code is 0, the mask empty
                    cd[0] = cd[1] = cd[2] = cd[3] = 0;
                } else {
                    uint32_t* mine = lh + wave * GL_NBINS;
                    int b4[4];
                    double e0[4], e1[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int b = (int)((a[i] - h_lo) * h_norm);
                        b = b < 0 ? 0 : (b > GL_NBINS - 1 ? GL_NBINS - 1 : b);
                        b4[i] = b;
                        e0[i] = edges[b];
                        e1[i] = edges[b + 1];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double v = a[i];
                        int b = b4[i];
                        if (v < e0[i] || (b < GL_NBINS - 1 && v >= e1[i])) {  // within rounding distance of an edge
                            while (b > 0 && v < edges[b]) --b;
                            while (b < GL_NBINS - 1 && v >= edges[b + 1]) ++b;
                            e0[i] = edges[b];
                            e1[i] = edges[b + 1];
                        }
                        cd[i] = ((unsigned)b << 1) | (v > (e0[i] + e1[i]) / 2.0 ? 1u : 0u);
                        b4[i] = b;
                    }
                    // count: a smoothed background wave sits in one or two bins -- the first lane's bin is counted once
                    // for every lane that shares it, twice in a row, the rest lane by lane (as hist_f64_kernel)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int b = b4[i];
                        const unsigned long long act = __ballot(1);
                        const int l0 = __ffsll((long long)act) - 1;
                        const int b0 = __shfl(b, l0);
                        const unsigned long long s0 = __ballot(b == b0);
                        if (lane == l0) atomicAdd(&mine[b0], (unsigned)__popcll(s0));
                        unsigned long long rest = act & ~s0;
                        if (rest) {
                            const int l1 = __ffsll((long long)rest) - 1;
                            const int b1 = __shfl(b, l1);
                            const unsigned long long s1 = __ballot(b == b1);
                            if (lane == l1) atomicAdd(&mine[b1], (unsigned)__popcll(s1));
                            rest &= ~s1;
                            if ((rest >> lane) & 1ull) atomicAdd(&mine[b], 1u);
                        }
                    }
                }
                uint16_t* cdst = codes + plane + (size_t)(y0 + rg + q) * W + xo;
                *reinterpret_cast<uint2*>(cdst) = make_uint2(cd[0] | (cd[1] << 16), cd[2] | (cd[3] << 16));
            }
        }
#pragma unroll
        for (int k = 0; k < K - 1; ++k) win[k] = win[k + 4];
    }
    if (EPI <= 1 && keys) {
        unsigned long long klo = amt_f64_key(vlo), khi = amt_f64_key(vhi);
        if (vlo > vhi) {
            klo = ~0ull;
            khi = 0ull;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(klo, off), h2 = __shfl_xor(khi, off);
            klo = l2 < klo ? l2 : klo;
            khi = h2 > khi ? h2 : khi;
        }
        if (lane == 0) {
            atomicMin(&keys[2 * blockIdx.z], klo);
            atomicMax(&keys[2 * blockIdx.z + 1], khi);
        }
    }
    if (EPI == 2) {
        __syncthreads();
        uint32_t* gh = hist + (size_t)blockIdx.z * GL_NBINS;
        for (int i = t; i < GL_NBINS; i += 256) {
            const uint32_t c = lh[i] + lh[GL_NBINS + i] + lh[2 * GL_NBINS + i] + lh[3 * GL_NBINS + i];
            if (c) atomicAdd(&gh[i], c);
        }
    }
}

template <typename TIn, int R>
static int launch_fused(amt_ctx* ctx, const TIn* in, double scale, double* out, int nplanes, int H, int W,
                        const double* wdev, int mode, double cval, size_t in_stride, unsigned long long* keys) {
    if constexpr (sizeof(TIn) == 2) {
        // direct-to-LDS input path (see gauss_lds_kernel): needs chunk-aligned rows and an in-tile boundary extension
        const bool aligned = (W % 8 == 0) && W >= 256 && (in_stride % 8 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                             ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        const bool modeok = mode == AMT_MODE_NEAREST || mode == AMT_MODE_REFLECT || mode == AMT_MODE_MIRROR;
        const char* off = getenv("AMT_GAUSS_LDS");
        if (aligned && modeok && H > 2 * R && !(off && off[0] == '0')) {
            constexpr int RP = (R + 7) & ~7;
            constexpr int OUTW2 = 256 - 2 * RP;
            const int gx2 = (W + OUTW2 - 1) / OUTW2;
            int TH2 = 256;
            while (TH2 > 32 && (long long)gx2 * ((H + TH2 - 1) / TH2) * nplanes < 4LL * ctx->num_cus) TH2 >>= 1;
            dim3 grid2(gx2, (H + TH2 - 1) / TH2, nplanes);
            hipLaunchKernelGGL((gauss_lds_kernel<R, 0>), grid2, dim3(256), 0, ctx->stream, (const uint16_t*)in, scale, out,
                               H, W, wdev, mode, in_stride, TH2, keys, (const double*)nullptr, (uint32_t*)nullptr,
                               (uint16_t*)nullptr);
            AMT_LAUNCH_CHECK();
            return AMT_OK;
        }
    }
    constexpr int OUTW = 256 - 2 * R;
    const int gx = (W + OUTW - 1) / OUTW;
    // rows per block: long chunks amortise the 2R warm-up rows, but keep >= ~4 blocks per CU in flight
    int TH = 256;
    while (TH > 32 && (long long)gx * ((H + TH - 1) / TH) * nplanes < 4LL * ctx->num_cus) TH >>= 1;
    dim3 grid(gx, (H + TH - 1) / TH, nplanes);
    hipLaunchKernelGGL((gauss_fused_kernel<TIn, R>), grid, dim3(256), 0, ctx->stream, in, scale, out, H, W, wdev,
                       mode, cval, in_stride, TH, keys);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ------------------------------------------------------------------------------------------------
// The same two passes with the tile staged by LDS-DMA (`global_load_lds_dwordx4`): a wave issues ALL its tile loads
// back to back (no registers, nothing waits until the one vmcnt(0) in front of the barrier), and the horizontal tiles
// are a quarter as tall so that four workgroups share a CU: one's staging overlaps the others' arithmetic.
// Measured (32 planes of 2048^2, r = 64, rocprofv3): horizontal 1,990 -> 1,600 us, vertical 1,193 -> 1,109 us.  What is
// left is arithmetic: tools/probes/fp64_probe.hip shows a SIMD retiring one fp64 instruction per 4.5 clocks at best
// (4 waves x 8 independent chains; 6.7 with one wave) => 0.74 ms per pass for the 193 operations per sample; with the
// staging removed the horizontal pass still takes ~1.05 ms (conflict-free LDS reads: -4 %, more ILP in the step: 0).
//   horizontal: 16 rows x 256 tile columns (pitch 258 doubles = a multiple of 16 bytes; lane L reads row L & 15, the
//               four lane groups of a wave work on different chunks), tile origin at an even column so that every lane moves an
//               aligned pair; columns outside the image are loaded from clamped addresses and then replaced in LDS by
//               their boundary image (reflect / mirror / nearest: another tile column; constant: cval).
//   vertical  : (TH + 2r) rows x 64 columns of the RAW type; one instruction moves 1 KiB = 8 (uint16) or 2 (float64)
//               boundary-mapped rows.
// ------------------------------------------------------------------------------------------------
constexpr int H8G_CP = 256, H8G_PITCH = 258;

template <int H8G_ROWS, int RC = 0>
__global__ void __launch_bounds__(256) conv_h8g_kernel(const double* __restrict__ in, double* out, int H, int W,
                                                       const double* __restrict__ wts, int r, int mode, double cval,
                                                       int TW, const double* minuend, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);        // H8G_ROWS x H8G_PITCH
    double* wl = tile + (size_t)H8G_ROWS * H8G_PITCH;          // 2r + 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // consecutive workgroups go to different XCDs (8, each with its own L2); tiles that are neighbours along x share
    // their 2r halo columns, so the x-tiles of a band are given to ONE XCD, back to back
    int bx = blockIdx.x, by = blockIdx.y;
    if (remap) {
        const int nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = nb >> 3;  // the host only asks for the remap when nb % 8 == 0
        const int v = (lin & 7) * per + (lin >> 3);
        bx = v % gridDim.x;
        by = v / gridDim.x;
    }
    const int x0 = bx * TW;
    const int y0 = by * H8G_ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int xs = (x0 - r) & ~1;  // even image column of tile column 0 (two's complement: rounds down)
    const int delta = x0 - r - xs;
    for (int i = threadIdx.x; i < 2 * r + 1; i += 256) wl[i] = wts[i];
    // staging: wave w moves rows (ROWS / 4) w .. + ROWS / 4 - 1, two 1 KiB chunks (128 doubles) per row
#pragma unroll
    for (int rr = 0; rr < H8G_ROWS / 4; ++rr) {
        const int row = wave * (H8G_ROWS / 4) + rr;
        const int y = min(y0 + row, H - 1);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            int col = xs + ch * 128 + 2 * lane;
            col = col < 0 ? 0 : (col > W - 2 ? W - 2 : col);
            const double* gp = in + plane + (size_t)y * W + col;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(tile + row * H8G_PITCH + ch * 128),
                                             16, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    if (xs < 0 || xs + H8G_CP > W) {  // uniform: a tile that reaches over the image edge
        for (int idx = threadIdx.x; idx < H8G_ROWS * H8G_CP; idx += 256) {
            const int row = idx >> 8, c = idx & 255;
            const int gx = xs + c;
            if (gx < 0 || gx >= W) {
                const int m = amt_map_index(gx, W, mode);
                int mi = m - xs;  // inside the tile for every column an output needs; the others are clamped
                mi = mi < 0 ? 0 : (mi > H8G_CP - 1 ? H8G_CP - 1 : mi);
                tile[row * H8G_PITCH + c] = m < 0 ? cval : tile[row * H8G_PITCH + mi];
            }
        }
        __syncthreads();
    }
    constexpr int NG = 64 / H8G_ROWS;  // lane groups of a wave, each on a chunk of its own
    const int rowl = lane & (H8G_ROWS - 1), half = lane / H8G_ROWS;
    const double* myrow = tile + rowl * H8G_PITCH + delta;
    auto at = [&](int i) -> double { return myrow[i]; };
    const int nchunks = TW >> 3;
    const int y = y0 + rowl;
    for (int cb = wave * NG; cb < nchunks; cb += 4 * NG) {
        const int cw = cb + half;                       // this lane group's chunk
        const int c = cw < nchunks ? cw : nchunks - 1;  // an idle half recomputes the last chunk and stores nothing
        double res[8];
        blocked8<RC>(at, c * 8 + r, wl, r, res);
        const int x = x0 + c * 8;
        if (cw < nchunks && y < H && x < W) {
            double* dst = out + plane + (size_t)y * W + x;
            const double* mn = minuend ? minuend + plane + (size_t)y * W + x : nullptr;
            if (x + 7 < W) {
                if (mn) {
                    double2 m2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) m2[k] = *reinterpret_cast<const double2*>(mn + 2 * k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        res[2 * k] = m2[k].x - res[2 * k];
                        res[2 * k + 1] = m2[k].y - res[2 * k + 1];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2*>(dst + k) = make_double2(res[k], res[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (x + k < W) dst[k] = mn ? mn[k] - res[k] : res[k];
            }
        }
    }
}

template <typename TIn, int RC = 0>
__global__ void __launch_bounds__(256) conv_v8g_kernel(const TIn* __restrict__ in, double scale,
                                                       double* __restrict__ out, int H, int W,
                                                       const double* __restrict__ wts, int r, int mode, double cval,
                                                       int TH, size_t in_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int RPI = 1024 / (64 * (int)sizeof(TIn));  // rows per instruction: 8 (uint16) or 2 (float64)
    constexpr int LPR = 64 / RPI;                        // lanes per row
    constexpr int EPL = 16 / (int)sizeof(TIn);           // elements per lane
    const int rows = TH + 2 * r;
    const int rows_pad = (rows + RPI - 1) / RPI * RPI;
    TIn* tile = reinterpret_cast<TIn*>(smem_raw);  // rows_pad x 64
    double* wl = reinterpret_cast<double*>(smem_raw + (size_t)rows_pad * 64 * sizeof(TIn));  // 2r + 1
    unsigned char* inside = reinterpret_cast<unsigned char*>(wl + 2 * r + 1);                // rows: 0 = cval row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xb = blockIdx.x * 64;  // W is a multiple of 64 (the host checks)
    const int x = xb + lane;
    const int y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const TIn* src = in + (size_t)blockIdx.z * in_stride;
    for (int i = threadIdx.x; i < 2 * r + 1; i += 256) wl[i] = wts[i];
    int any_out = 0;
    for (int k = threadIdx.x; k < rows; k += 256) {
        const int yy = amt_map_index(y0 - r + k, H, mode);
        inside[k] = yy >= 0 ? 1 : 0;
        any_out |= yy < 0 ? 1 : 0;
    }
    // staging: one instruction = RPI consecutive tile rows; the row of a lane is mapped by that lane
    const int lrow = lane / LPR, lcol = (lane % LPR) * EPL;
    for (int k0 = wave * RPI; k0 < rows_pad; k0 += 4 * RPI) {
        const int k = k0 + lrow;
        int yy = amt_map_index(y0 - r + (k < rows ? k : rows - 1), H, mode);
        yy = yy < 0 ? 0 : yy;  // 'constant' rows are replaced where they are read
        const TIn* gp = src + (size_t)yy * W + xb + lcol;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(tile + (size_t)k0 * 64), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    const int has_out = __syncthreads_or(any_out);
    auto at = [&](int i) -> double { return inside[i] ? cvt_f64<TIn>(tile[i * 64 + lane], scale) : cval; };
    auto at_in = [&](int i) -> double { return cvt_f64<TIn>(tile[i * 64 + lane], scale); };
    for (int c = wave; c * 8 < TH; c += 4) {
        const int q0 = c * 8;
        if (y0 + q0 >= H) break;
        double acc[8];
        if (has_out)
            blocked8<RC>(at, q0 + r, wl, r, acc);
        else
            blocked8<RC>(at_in, q0 + r, wl, r, acc);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (y0 + q0 + k < H) out[plane + (size_t)(y0 + q0 + k) * W + x] = acc[k];
    }
}

template <typename TIn>
static int gaussian_typed(amt_ctx* ctx, const TIn* in, double scale, double* out, double* tmp, int nplanes, int H,
                          int W, const double* wdev, int r, int mode, double cval, size_t in_stride,
                          unsigned long long* keys, const double* minuend = nullptr) {
    // minuend: only the two-pass path (r > FR_MAX) can subtract in its epilogue; callers check
    switch (r) {
#define AMT_FUSED_CASE(RR) \
    case RR:               \
        return launch_fused<TIn, RR>(ctx, in, scale, out, nplanes, H, W, wdev, mode, cval, in_stride, keys);
        AMT_FUSED_CASE(1)
        AMT_FUSED_CASE(2)
        AMT_FUSED_CASE(3)
        AMT_FUSED_CASE(4)
        AMT_FUSED_CASE(5)
        AMT_FUSED_CASE(6)
        AMT_FUSED_CASE(7)
        AMT_FUSED_CASE(8)
        AMT_FUSED_CASE(9)
        AMT_FUSED_CASE(10)
        AMT_FUSED_CASE(11)
        AMT_FUSED_CASE(12)
#undef AMT_FUSED_CASE
        default:
            break;
    }
    // register-blocked two-pass path
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(tmp) & 15) == 0 &&
                         (in_stride * sizeof(TIn)) % 16 == 0 && getenv("AMT_GAUSS_NO_GLDS") == nullptr;
    bool v_done = false, h_done = false;
    const bool no_rc = getenv("AMT_GAUSS_NO_RC") != nullptr;  // A/B: the runtime-radius instances for r = 64 too
    if (aligned && W % 64 == 0 && H > 2 * r) {  // vertical pass, LDS-DMA staging
        int TH = sizeof(TIn) == 2 ? 128 : 64;
        const int rows_pad = (TH + 2 * r + 7) & ~7;
        const size_t smem0 = (size_t)rows_pad * 64 * sizeof(TIn) + (size_t)(2 * r + 1) * 8 + (size_t)(TH + 2 * r);
        dim3 g0(W / 64, (H + TH - 1) / TH, nplanes);
        if (r == 64 && !no_rc)
            hipLaunchKernelGGL((conv_v8g_kernel<TIn, 64>), g0, dim3(256), smem0, ctx->stream, in, scale, tmp, H, W, wdev, r,
                               mode, cval, TH, in_stride);
        else
            hipLaunchKernelGGL((conv_v8g_kernel<TIn>), g0, dim3(256), smem0, ctx->stream, in, scale, tmp, H, W, wdev, r,
                               mode, cval, TH, in_stride);
        AMT_LAUNCH_CHECK();
        v_done = true;
    }
    if (!v_done) {
        // vertical: rows per block chosen so that the raw-type tile stays within ~96 KB
        int TH = 128;
        while (TH > 32 && ((size_t)(TH + 2 * r) * 64 * sizeof(TIn) + (size_t)(TH + 2 * r)) > 96 * 1024) TH >>= 1;
        const size_t smem0 = (size_t)(TH + 2 * r) * 64 * sizeof(TIn) + 8 + (size_t)(2 * r + 1) * 8 + (size_t)(TH + 2 * r);
        dim3 g0((W + 63) / 64, (H + TH - 1) / TH, nplanes);
        hipLaunchKernelGGL((conv_v8_kernel<TIn>), g0, dim3(256), smem0, ctx->stream, in, scale, tmp, H, W, wdev, r, mode,
                           cval, TH, in_stride);
        AMT_LAUNCH_CHECK();
    }
    // horizontal pass, LDS-DMA staging: the tile (TW outputs + 2r halo + the alignment column) must fit 256 columns,
    // and an out-of-image column must have its boundary image inside the tile
    const int tw_g = ((H8G_CP - 2 * r - 1) / 8) * 8;
    const bool h_ok = aligned && W % 2 == 0 && tw_g >= 64 && W >= H8G_CP + 2 * r && mode != AMT_MODE_WRAP;
    if (h_ok) {
        const int TW = tw_g > 128 ? 128 : tw_g;
        // 16-row tiles: four workgroups (16 waves) per CU; measured 4 % faster than 32-row tiles (two per CU)
        constexpr int ROWS = 16;
        const size_t smem1 = (size_t)ROWS * H8G_PITCH * sizeof(double) + (size_t)(2 * r + 1) * 8;
        dim3 g1((W + TW - 1) / TW, (H + ROWS - 1) / ROWS, nplanes);
        const int remap = (g1.x * g1.y) % 8 == 0;
        if (r == 64 && !no_rc)
            hipLaunchKernelGGL((conv_h8g_kernel<ROWS, 64>), g1, dim3(256), smem1, ctx->stream, tmp, out, H, W, wdev, r, mode,
                               cval, TW, minuend, remap);
        else
            hipLaunchKernelGGL((conv_h8g_kernel<ROWS>), g1, dim3(256), smem1, ctx->stream, tmp, out, H, W, wdev, r, mode,
                               cval, TW, minuend, remap);
        AMT_LAUNCH_CHECK();
        h_done = true;
    }
    if (!h_done) {
        // horizontal: 64 rows x TW columns per block; wide tiles amortise the 2r halo columns
        int TW = 128;
        while (TW > 32 && (size_t)64 * ((TW + 2 * r) | 1) * sizeof(double) > 136 * 1024) TW >>= 1;
        const size_t smem1 = (size_t)64 * ((TW + 2 * r) | 1) * sizeof(double) + (size_t)(2 * r + 1) * 8 +
                             (size_t)(TW + 2 * r) * sizeof(int);
        dim3 g1((W + TW - 1) / TW, (H + 63) / 64, nplanes);
        hipLaunchKernelGGL(conv_h8_kernel, g1, dim3(256), smem1, ctx->stream, tmp, out, H, W, wdev, r, mode, cval, TW,
                           minuend);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

static int check_gauss_args(const void* in, int in_dtype, double* out, int nplanes, int H, int W, const double* w,
                            int r) {
    AMT_REQUIRE(in && out && w, "gaussian: null pointer");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "gaussian: in_dtype must be AMT_U16 or AMT_F64");
    AMT_REQUIRE(nplanes >= 0 && H > 0 && W > 0, "gaussian: bad shape %d x %d x %d", nplanes, H, W);
    AMT_REQUIRE(r >= 0 && r <= 128, "gaussian: radius %d unsupported (0..128, i.e. sigma <= 32)", r);
    return AMT_OK;
}

static int gaussian_dispatch(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, double* tmp,
                             int nplanes, int H, int W, const double* wdev, int r, int mode, double cval,
                             size_t in_stride = 0, unsigned long long* keys = nullptr,
                             const double* minuend = nullptr) {
    if (in_stride == 0) in_stride = (size_t)H * W;
    if (in_dtype == AMT_U16)
        return gaussian_typed<uint16_t>(ctx, (const uint16_t*)in, scale, out, tmp, nplanes, H, W, wdev, r, mode, cval,
                                        in_stride, keys, minuend);
    return gaussian_typed<double>(ctx, (const double*)in, 1.0, out, tmp, nplanes, H, W, wdev, r, mode, cval, in_stride,
                                  keys, minuend);
}

__global__ void convert_u16_f64_kernel(const uint16_t* __restrict__ in, double scale, double* __restrict__ out,
                                       size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i] * scale;
}

__global__ void copy_f64_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

extern "C" int amt_convert_u16_f64(amt_ctx* ctx, const uint16_t* in, double scale, double* out, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out, "convert_u16_f64: null pointer");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(convert_u16_f64_kernel, dim3(amt_grid_for(n, 256)), dim3(256), 0, ctx->stream, in, scale, out,
                       n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_gaussian(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int H,
                            int W, const double* weights, int radius, int mode, double cval, size_t in_plane_stride,
                            double* minmax_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_TRY(check_gauss_args(in, in_dtype, out, nplanes, H, W, weights, radius));
    AMT_REQUIRE(in_plane_stride == 0 || in_plane_stride >= (size_t)H * W, "gaussian: in_plane_stride smaller than a plane");
    AMT_REQUIRE(radius > 0 || in_plane_stride == 0 || in_plane_stride == (size_t)H * W,
                "gaussian: strided input needs radius >= 1");
    if (nplanes == 0) return AMT_OK;
    size_t n = (size_t)nplanes * H * W;
    const size_t kbytes = minmax_dev ? amt_align((size_t)2 * nplanes * 8) : 0;
    if (radius == 0) {  // sigma too small: scipy's kernel is the single weight 1.0
        if (in_dtype == AMT_U16) {
            AMT_TRY(amt_convert_u16_f64(ctx, (const uint16_t*)in, scale, out, n));
        } else {
            hipLaunchKernelGGL(copy_f64_kernel, dim3(amt_grid_for(n, 256)), dim3(256), 0, ctx->stream,
                               (const double*)in, out, n);
            AMT_LAUNCH_CHECK();
        }
        if (minmax_dev) {
            AMT_TRY(amt_arena_begin(ctx, kbytes));
            unsigned long long* keys = (unsigned long long*)amt_arena_take(ctx, kbytes);
            AMT_TRY(amt_i_minmax_f64(ctx, out, keys, minmax_dev, nplanes, (size_t)H * W));
        }
        return AMT_OK;
    }
    size_t wbytes = amt_align((2 * radius + 1) * sizeof(double));
    size_t tmpbytes = radius > FR_MAX ? amt_align(n * sizeof(double)) : 0;
    AMT_TRY(amt_arena_begin(ctx, wbytes + tmpbytes + kbytes));
    double* wdev = (double*)amt_arena_take(ctx, wbytes);
    double* tmp = tmpbytes ? (double*)amt_arena_take(ctx, tmpbytes) : nullptr;
    unsigned long long* keys = kbytes ? (unsigned long long*)amt_arena_take(ctx, kbytes) : nullptr;
    AMT_TRY(amt_param_upload(ctx, wdev, weights, (2 * radius + 1) * sizeof(double)));
    if (!minmax_dev)
        return gaussian_dispatch(ctx, in, in_dtype, scale, out, tmp, nplanes, H, W, wdev, radius, mode, cval,
                                 in_plane_stride);
    if (radius > FR_MAX) {  // two-pass path: min / max in a pass of their own
        AMT_TRY(gaussian_dispatch(ctx, in, in_dtype, scale, out, tmp, nplanes, H, W, wdev, radius, mode, cval,
                                  in_plane_stride));
        return amt_i_minmax_f64(ctx, out, keys, minmax_dev, nplanes, (size_t)H * W);
    }
    AMT_TRY(amt_i_minmax_init(ctx, keys, nplanes));
    AMT_TRY(gaussian_dispatch(ctx, in, in_dtype, scale, out, tmp, nplanes, H, W, wdev, radius, mode, cval,
                              in_plane_stride, keys));
    return amt_i_minmax_finish(ctx, keys, minmax_dev, nplanes);
}

// ---- Gaussian -> Otsu without the float64 plane (the mask chain of BASELINE configs[1] / [2]) ------------------------
// R/ callers: ski.filters.gaussian -> ski.filters.threshold_otsu -> `>` (SURVEY.md A.7/A.8 recipes through
// R/pipeline.py:25-45).  Two passes of the fused uint16 Gaussian over the input: the first folds min / max, the second
// recomputes the samples, counts np.histogram's 256 bins and stores 2-byte codes; Otsu runs on the histogram and the
// threshold comparison continues on the codes (amt_threshold_open_close / amt_threshold_gt on the uint16 code plane
// with thr_code).  26 bytes of HBM traffic per pixel (write + two reads of a float64 plane, read uint16) become 8.
template <int R>
static int launch_codes(amt_ctx* ctx, const uint16_t* in, double scale, int nplanes, int H, int W, const double* wdev,
                        int mode, size_t in_stride, unsigned long long* keys, double* mm, uint32_t* hist,
                        uint16_t* codes) {
    constexpr int RP = (R + 7) & ~7;
    constexpr int OUTW2 = 256 - 2 * RP;
    const int gx2 = (W + OUTW2 - 1) / OUTW2;
    int TH2 = 256;
    while (TH2 > 32 && (long long)gx2 * ((H + TH2 - 1) / TH2) * nplanes < 4LL * ctx->num_cus) TH2 >>= 1;
    dim3 grid2(gx2, (H + TH2 - 1) / TH2, nplanes);
    AMT_TRY(amt_i_minmax_init(ctx, keys, nplanes));
    hipLaunchKernelGGL((gauss_lds_kernel<R, 1>), grid2, dim3(256), 0, ctx->stream, in, scale, (double*)nullptr, H, W,
                       wdev, mode, in_stride, TH2, keys, (const double*)nullptr, (uint32_t*)nullptr, (uint16_t*)nullptr);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_i_minmax_finish(ctx, keys, mm, nplanes));
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * GL_NBINS * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL((gauss_lds_kernel<R, 2>), grid2, dim3(256), 0, ctx->stream, in, scale, (double*)nullptr, H, W,
                       wdev, mode, in_stride, TH2, (unsigned long long*)nullptr, (const double*)mm, hist, codes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_gaussian_otsu_codes_supported(int H, int W, int radius, int mode, size_t in_plane_stride) {
    const bool modeok = mode == AMT_MODE_NEAREST || mode == AMT_MODE_REFLECT || mode == AMT_MODE_MIRROR;
    return radius >= 1 && radius <= FR_MAX && modeok && W % 8 == 0 && W >= 256 && H > 2 * radius &&
           (in_plane_stride == 0 || in_plane_stride % 8 == 0);
}

extern "C" int amt_gaussian_otsu_codes(amt_ctx* ctx, const uint16_t* in, double scale, int nplanes, int H, int W,
                                       const double* weights, int radius, int mode, size_t in_plane_stride,
                                       double* minmax_dev, uint32_t* hist_dev, double* thr_dev, double* thr_code_dev,
                                       uint16_t* codes) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && weights && minmax_dev && hist_dev && thr_dev && thr_code_dev && codes && nplanes >= 0,
                "gaussian_otsu_codes: bad arguments");
    AMT_REQUIRE(amt_gaussian_otsu_codes_supported(H, W, radius, mode, in_plane_stride) &&
                    (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0,
                "gaussian_otsu_codes: unsupported shape / radius / mode / alignment (ask "
                "amt_gaussian_otsu_codes_supported; the separate operators handle every case)");
    if (nplanes == 0) return AMT_OK;
    if (in_plane_stride == 0) in_plane_stride = (size_t)H * W;
    const size_t wbytes = amt_align((2 * radius + 1) * sizeof(double)), kbytes = amt_align((size_t)2 * nplanes * 8);
    AMT_TRY(amt_arena_begin(ctx, wbytes + kbytes));
    double* wdev = (double*)amt_arena_take(ctx, wbytes);
    unsigned long long* keys = (unsigned long long*)amt_arena_take(ctx, kbytes);
    AMT_TRY(amt_param_upload(ctx, wdev, weights, (2 * radius + 1) * sizeof(double)));
    switch (radius) {
#define AMT_CODES_CASE(RR)                                                                                            \
    case RR:                                                                                                          \
        AMT_TRY(launch_codes<RR>(ctx, in, scale, nplanes, H, W, wdev, mode, in_plane_stride, keys, minmax_dev, hist_dev, \
                                 codes));                                                                             \
        break;
        AMT_CODES_CASE(1)
        AMT_CODES_CASE(2)
        AMT_CODES_CASE(3)
        AMT_CODES_CASE(4)
        AMT_CODES_CASE(5)
        AMT_CODES_CASE(6)
        AMT_CODES_CASE(7)
        AMT_CODES_CASE(8)
        AMT_CODES_CASE(9)
        AMT_CODES_CASE(10)
        AMT_CODES_CASE(11)
        AMT_CODES_CASE(12)
#undef AMT_CODES_CASE
        default:
            amt_set_error("gaussian_otsu_codes: radius %d", radius);
            return AMT_EINVAL;
    }
    return amt_i_otsu_from_hist(ctx, hist_dev, minmax_dev, GL_NBINS, thr_dev, thr_code_dev, nplanes);
}

__global__ void sub_inplace_kernel(double* __restrict__ a, const double* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = a[i] - b[i];
}

// One axis of an n-D separable filter: the array is viewed as nplanes x L x inner, the filter runs along L (scipy
// filters the axes of an n-D image one after the other, leading axes first; the last two axes go through amt_gaussian).
extern "C" int amt_convolve_axis0(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int L,
                                  int inner, const double* weights, int radius, int mode, double cval) {
    AMT_TRY(amt_set_device(ctx));
    AMT_TRY(check_gauss_args(in, in_dtype, out, nplanes, L, inner, weights, radius));
    AMT_REQUIRE(radius >= 1, "convolve_axis0: radius must be >= 1");
    AMT_REQUIRE((const void*)in != (const void*)out, "convolve_axis0: in-place operation is not supported");
    if (nplanes == 0) return AMT_OK;
    const int r = radius;
    size_t wbytes = amt_align((2 * r + 1) * sizeof(double));
    AMT_TRY(amt_arena_begin(ctx, wbytes));
    double* wdev = (double*)amt_arena_take(ctx, wbytes);
    AMT_TRY(amt_param_upload(ctx, wdev, weights, (2 * r + 1) * sizeof(double)));
    const size_t esz = in_dtype == AMT_U16 ? 2 : 8;
    int TH = 128;
    while (TH > 32 && ((size_t)(TH + 2 * r) * 64 * esz + (size_t)(TH + 2 * r)) > 96 * 1024) TH >>= 1;
    const size_t smem0 = (size_t)(TH + 2 * r) * 64 * esz + 8 + (size_t)(2 * r + 1) * 8 + (size_t)(TH + 2 * r);
    dim3 g0((inner + 63) / 64, (L + TH - 1) / TH, nplanes);
    if (in_dtype == AMT_U16)
        hipLaunchKernelGGL((conv_v8_kernel<uint16_t>), g0, dim3(256), smem0, ctx->stream, (const uint16_t*)in, scale, out, L,
                           inner, wdev, r, mode, cval, TH, (size_t)L * inner);
    else
        hipLaunchKernelGGL((conv_v8_kernel<double>), g0, dim3(256), smem0, ctx->stream, (const double*)in, 1.0, out, L, inner,
                           wdev, r, mode, cval, TH, (size_t)L * inner);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_dog(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int H,
                       int W, const double* w_lo, int r_lo, const double* w_hi, int r_hi, int mode, double cval) {
    AMT_TRY(amt_set_device(ctx));
    AMT_TRY(check_gauss_args(in, in_dtype, out, nplanes, H, W, w_lo, r_lo));
    AMT_TRY(check_gauss_args(in, in_dtype, out, nplanes, H, W, w_hi, r_hi));
    AMT_REQUIRE(r_lo >= 1 && r_hi >= 1, "dog: radii must be >= 1");
    if (nplanes == 0) return AMT_OK;
    size_t n = (size_t)nplanes * H * W;
    size_t wl = amt_align((2 * r_lo + 1) * sizeof(double)), wh = amt_align((2 * r_hi + 1) * sizeof(double));
    size_t nb = amt_align(n * sizeof(double));
    bool need_tmp = (r_lo > FR_MAX) || (r_hi > FR_MAX);
    // a wide second Gaussian (two-pass path) subtracts itself from the first in its horizontal pass: no ghi plane, no
    // subtraction pass (the reference's default sigmas 0.6 / 16 take this route)
    const bool fused_sub = r_hi > FR_MAX;
    AMT_TRY(amt_arena_begin(ctx, wl + wh + (fused_sub ? 0 : nb) + (need_tmp ? nb : 0)));
    double* wlo_d = (double*)amt_arena_take(ctx, wl);
    double* whi_d = (double*)amt_arena_take(ctx, wh);
    double* ghi = fused_sub ? nullptr : (double*)amt_arena_take(ctx, nb);
    double* tmp = need_tmp ? (double*)amt_arena_take(ctx, nb) : nullptr;
    AMT_TRY(amt_param_upload(ctx, wlo_d, w_lo, (2 * r_lo + 1) * sizeof(double)));
    AMT_TRY(amt_param_upload(ctx, whi_d, w_hi, (2 * r_hi + 1) * sizeof(double)));
    AMT_TRY(gaussian_dispatch(ctx, in, in_dtype, scale, out, tmp, nplanes, H, W, wlo_d, r_lo, mode, cval));
    if (fused_sub) {
        AMT_TRY(gaussian_dispatch(ctx, in, in_dtype, scale, out, tmp, nplanes, H, W, whi_d, r_hi, mode, cval, 0, nullptr,
                                  out));
        return AMT_OK;
    }
    AMT_TRY(gaussian_dispatch(ctx, in, in_dtype, scale, ghi, tmp, nplanes, H, W, whi_d, r_hi, mode, cval));
    hipLaunchKernelGGL(sub_inplace_kernel, dim3(amt_grid_for(n, 256)), dim3(256), 0, ctx->stream, out, ghi, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- elementwise -------------------------------------------------------------------------------
__global__ void sub_clip0_kernel(const double* __restrict__ in, const double* __restrict__ level,
                                 double* __restrict__ out, size_t n) {
    const double lv = level[blockIdx.y];
    const size_t base = (size_t)blockIdx.y * n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double v = in[base + i] - lv;
        // np.clip(v, 0, None) == maximum(v, 0): NaN propagates, -0.0 -> compares equal to 0
        out[base + i] = (v < 0.0) ? 0.0 : v;
    }
}

extern "C" int amt_sub_clip0_f64(amt_ctx* ctx, const double* in, const double* level_dev, double* out, int nplanes,
                                 size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && level_dev && out, "sub_clip0: null pointer");
    if (nplanes == 0 || n == 0) return AMT_OK;
    dim3 grid(amt_grid_for(n, 256, 2048), nplanes);
    hipLaunchKernelGGL(sub_clip0_kernel, grid, dim3(256), 0, ctx->stream, in, level_dev, out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// SK/exposure/exposure.py:405-428: image = clip(image, imin, imax) IN THE INPUT DTYPE'S VALUE SPACE
// (np.clip of a uint16 array with float bounds yields float64), then (image - imin) / (imax - imin),
// then * (omax - omin) + omin.  imin == imax -> clip(image, omin, omax).
template <typename TIn>
__global__ void rescale_kernel(const TIn* __restrict__ in, const double* __restrict__ range, double omin, double omax,
                               double* __restrict__ out, size_t n) {
    const double imin = range[2 * blockIdx.y], imax = range[2 * blockIdx.y + 1];
    const size_t base = (size_t)blockIdx.y * n;
    const double den = imax - imin;
    const double osc = omax - omin;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double v = (double)in[base + i];
        if (imin != imax) {
            v = v < imin ? imin : (v > imax ? imax : v);
            v = (v - imin) / den;
            out[base + i] = v * osc + omin;
        } else {
            v = v < imin ? imin : (v > imax ? imax : v);
            out[base + i] = v < omin ? omin : (v > omax ? omax : v);
        }
    }
}

extern "C" int amt_rescale(amt_ctx* ctx, const void* in, int in_dtype, const double* range_dev, double omin,
                           double omax, double* out, int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && range_dev && out, "rescale: null pointer");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "rescale: in_dtype must be AMT_U16 or AMT_F64");
    if (nplanes == 0 || n == 0) return AMT_OK;
    dim3 grid(amt_grid_for(n, 256, 2048), nplanes);
    if (in_dtype == AMT_U16)
        hipLaunchKernelGGL((rescale_kernel<uint16_t>), grid, dim3(256), 0, ctx->stream, (const uint16_t*)in, range_dev,
                           omin, omax, out, n);
    else
        hipLaunchKernelGGL((rescale_kernel<double>), grid, dim3(256), 0, ctx->stream, (const double*)in, range_dev,
                           omin, omax, out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

template <typename T>
__global__ void subtract_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (T)(a[i] - b[i]);
}

extern "C" int amt_subtract(amt_ctx* ctx, const void* a, const void* b, void* out, int dtype, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(a && b && out, "subtract: null pointer");
    if (n == 0) return AMT_OK;
    dim3 grid(amt_grid_for(n, 256));
    switch (dtype) {
        case AMT_U16:
            hipLaunchKernelGGL((subtract_kernel<uint16_t>), grid, dim3(256), 0, ctx->stream, (const uint16_t*)a,
                               (const uint16_t*)b, (uint16_t*)out, n);
            break;
        case AMT_F64:
            hipLaunchKernelGGL((subtract_kernel<double>), grid, dim3(256), 0, ctx->stream, (const double*)a,
                               (const double*)b, (double*)out, n);
            break;
        case AMT_U8:
            hipLaunchKernelGGL((subtract_kernel<uint8_t>), grid, dim3(256), 0, ctx->stream, (const uint8_t*)a,
                               (const uint8_t*)b, (uint8_t*)out, n);
            break;
        default:
            amt_set_error("subtract: unsupported dtype %d", dtype);
            return AMT_EINVAL;
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

__global__ void deinterleave_u16_kernel(const uint16_t* __restrict__ yxc, uint16_t* __restrict__ cyx, size_t npix,
                                        int C) {
    const size_t fbase = (size_t)blockIdx.y * npix * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < npix * C; i += stride) {
        size_t p = i / C;
        int c = (int)(i - p * C);
        cyx[fbase + (size_t)c * npix + p] = yxc[fbase + i];
    }
}

extern "C" int amt_deinterleave_u16(amt_ctx* ctx, const uint16_t* yxc, uint16_t* cyx, int nplanes, int H, int W,
                                    int C) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(yxc && cyx && H > 0 && W > 0 && C > 0, "deinterleave: bad arguments");
    if (nplanes == 0) return AMT_OK;
    size_t npix = (size_t)H * W;
    dim3 grid(amt_grid_for(npix * C, 256, 4096), nplanes);
    hipLaunchKernelGGL(deinterleave_u16_kernel, grid, dim3(256), 0, ctx->stream, yxc, cyx, npix, C);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

__global__ void add_scalar_kernel(const double* __restrict__ in, double s, double* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i] + s;
}

// out = in + s  (threshold_local's "thresh_image - offset", SK/filters/thresholding.py:236)
extern "C" int amt_add_scalar_f64(amt_ctx* ctx, const double* in, double s, double* out, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out, "add_scalar_f64: null pointer");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(add_scalar_kernel, dim3(amt_grid_for(n, 256)), dim3(256), 0, ctx->stream, in, s, out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- rectangular copy: crop_to_center on a device-resident image (R/operations.py:100-132) ----------
__global__ void copy_rect_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int esz, int H, int W,
                                 int top, int left, int h, int w) {
    const size_t row_bytes = (size_t)w * esz;
    const uint8_t* s = src + ((size_t)blockIdx.z * H * W + (size_t)(top + blockIdx.y) * W + left) * esz;
    uint8_t* d = dst + ((size_t)blockIdx.z * h + blockIdx.y) * row_bytes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

// ---- np.pad(image, ((py, py), (px, px)), mode='edge'): scikit-image pads the image this way before an opening /
// closing with an even-sized footprint (SK/morphology/grey.py:84-127, pad_for_eccentric_selems) ------------------
__global__ void pad_edge_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int esz, int H, int W, int py,
                                int px) {
    const int Ho = H + 2 * py, Wo = W + 2 * px;
    const int yo = blockIdx.y;
    const int ys = min(max(yo - py, 0), H - 1);
    const uint8_t* s = src + ((size_t)blockIdx.z * H + ys) * W * esz;
    uint8_t* d = dst + ((size_t)blockIdx.z * Ho + yo) * Wo * esz;
    for (int xo = blockIdx.x * blockDim.x + threadIdx.x; xo < Wo; xo += gridDim.x * blockDim.x) {
        const int xs = min(max(xo - px, 0), W - 1);
        for (int b = 0; b < esz; ++b) d[(size_t)xo * esz + b] = s[(size_t)xs * esz + b];
    }
}

extern "C" int amt_pad_edge(amt_ctx* ctx, const void* src, void* dst, int elem_size, int nplanes, int H, int W, int py,
                            int px) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(src && dst && elem_size > 0 && nplanes >= 0 && H > 0 && W > 0 && py >= 0 && px >= 0,
                "pad_edge: bad arguments");
    if (nplanes == 0) return AMT_OK;
    dim3 grid((unsigned)((W + 2 * px + 255) / 256), H + 2 * py, nplanes);
    if (grid.x > 64) grid.x = 64;
    hipLaunchKernelGGL(pad_edge_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)src, (uint8_t*)dst, elem_size, H,
                       W, py, px);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_copy_rect(amt_ctx* ctx, const void* src, void* dst, int elem_size, int nplanes, int H, int W,
                             int top, int left, int h, int w) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(src && dst && elem_size > 0 && nplanes >= 0 && H > 0 && W > 0, "copy_rect: bad arguments");
    AMT_REQUIRE(top >= 0 && left >= 0 && h >= 0 && w >= 0 && top + h <= H && left + w <= W,
                "copy_rect: rectangle outside the image");
    if (nplanes == 0 || h == 0 || w == 0) return AMT_OK;
    dim3 grid((unsigned)(((size_t)w * elem_size + 255) / 256), h, nplanes);
    if (grid.x > 64) grid.x = 64;
    hipLaunchKernelGGL(copy_rect_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)src, (uint8_t*)dst, elem_size,
                       H, W, top, left, h, w);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
