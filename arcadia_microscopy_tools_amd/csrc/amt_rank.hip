// Grey-scale erosion / dilation / median over an arbitrary footprint (uint16 or float64 images).
//
// Semantics: scipy.ndimage.grey_erosion / grey_dilation / median_filter(footprint=...), which is what
// skimage.morphology.erosion / dilation / opening / closing / white_tophat and skimage.filters.median
// call (SK/morphology/grey.py:185,251; SK/filters/_median.py) -- ImageOperation callables in the
// reference (R/pipeline.py:25-45).  Flat footprint, anchor at the centre:
//   erosion  out[p] = min_{s in S} in[p + s]
//   dilation out[p] = max_{s in S} in[p - s]   (the caller passes skimage's already-mirrored footprint,
//                                               scipy mirrors it again, so offsets are applied as +s of
//                                               the ORIGINAL footprint's reflection -> see host layer)
//   median   out[p] = element of rank |S| // 2 of {in[p + s]}
// Boundary: scipy's `mode` (reflect for grey morphology, nearest for skimage's median).
// One 256-thread workgroup produces a 16 x 64 tile from an LDS-staged tile with halo.
#include "amt_common.h"

constexpr int RT_H = 16, RT_W = 64;
constexpr int RK_MAX_OFFS = 1024;

template <typename T>
struct key_traits;
template <>
struct key_traits<uint16_t> {
    typedef unsigned key_t;
    static constexpr int BITS = 16;
    __device__ static __forceinline__ key_t to_key(uint16_t v) { return v; }
    __device__ static __forceinline__ uint16_t from_key(key_t k) { return (uint16_t)k; }
};
template <>
struct key_traits<double> {
    typedef unsigned long long key_t;
    static constexpr int BITS = 64;
    __device__ static __forceinline__ key_t to_key(double v) { return amt_f64_key(v); }
    __device__ static __forceinline__ double from_key(key_t k) { return amt_key_f64(k); }
};

template <typename T>
__global__ void __launch_bounds__(256) rank_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                   const int2* __restrict__ offs_g, int noffs, int ry, int rx, int op,
                                                   int mode, T cval) {
    typedef typename key_traits<T>::key_t key_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int pitch = RT_W + 2 * rx;
    const int rows = RT_H + 2 * ry;
    key_t* tile = reinterpret_cast<key_t*>(smem_raw);
    int2* offs = reinterpret_cast<int2*>(smem_raw + amt_align((size_t)rows * pitch * sizeof(key_t), 16));
    const int x0 = blockIdx.x * RT_W, y0 = blockIdx.y * RT_H;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = threadIdx.x; i < noffs; i += 256) offs[i] = offs_g[i];
    for (int i = threadIdx.x; i < rows * pitch; i += 256) {
        int ky = i / pitch, kx = i - ky * pitch;
        int yy = amt_map_index(y0 - ry + ky, H, mode);
        int xx = amt_map_index(x0 - rx + kx, W, mode);
        T v = cval;
        if (yy >= 0 && xx >= 0) v = in[plane + (size_t)yy * W + xx];
        tile[i] = key_traits<T>::to_key(v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RT_H * RT_W; i += 256) {
        int ky = i / RT_W, kx = i - ky * RT_W;
        int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        const key_t* c = tile + (ky + ry) * pitch + (kx + rx);
        key_t r;
        if (op == 0) {
            r = ~(key_t)0;
            for (int k = 0; k < noffs; ++k) {
                key_t v = c[offs[k].y * pitch + offs[k].x];
                r = v < r ? v : r;
            }
        } else if (op == 1) {
            r = 0;
            for (int k = 0; k < noffs; ++k) {
                key_t v = c[-offs[k].y * pitch - offs[k].x];
                r = v > r ? v : r;
            }
        } else {
            // rank select: smallest key K such that #{v <= K} >= rank + 1, found bit by bit
            const int rank = noffs / 2;
            key_t prefix = 0;
            for (int bit = key_traits<T>::BITS - 1; bit >= 0; --bit) {
                // candidates share `prefix` above `bit`; count those with this bit clear
                const key_t himask = (bit == key_traits<T>::BITS - 1 && key_traits<T>::BITS == 64)
                                         ? (key_t)0
                                         : (~(key_t)0) << (bit + 1);
                const key_t trial = prefix | (((key_t)1 << bit) - 1);  // largest key with bit clear under prefix
                int cnt = 0;
                for (int k = 0; k < noffs; ++k) {
                    key_t v = c[offs[k].y * pitch + offs[k].x];
                    cnt += (v <= trial) ? 1 : 0;
                }
                (void)himask;
                if (cnt < rank + 1) prefix |= ((key_t)1 << bit);
            }
            r = prefix;
        }
        out[plane + (size_t)y * W + x] = key_traits<T>::from_key(r);
    }
}

// ---- median over small footprints (|S| <= 49): selection in registers ---------------------------------
// The |S| samples are fetched once from the LDS tile (offsets are wave-uniform scalars) and the median is found
// by "forgetful selection": keep a working set of |S|/2 + 2 values, repeatedly discard its minimum and maximum
// (neither can be the median) and admit the next sample, until three values remain.  About 1.5 compare-exchanges
// per step and element, all min / max instructions on registers -- versus 16 (64) passes over the footprint in
// the generic bitwise rank select.
template <typename K>
__device__ __forceinline__ void cmpxchg(K& a, K& b) {
    const K lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

template <typename K, int N>
__device__ __forceinline__ K median_select(const K (&v)[N]) {
    constexpr int W0 = N / 2 + 2;
    K w[W0];
#pragma unroll
    for (int i = 0; i < W0; ++i) w[i] = v[i < N ? i : N - 1];
#pragma unroll
    for (int m = (W0 < N ? W0 : N); m >= 3; --m) {
#pragma unroll
        for (int i = 0; i < m / 2; ++i) cmpxchg(w[i], w[m - 1 - i]);   // lower half <= upper half pairwise
#pragma unroll
        for (int i = 1; i < (m + 1) / 2; ++i) cmpxchg(w[0], w[i]);     // minimum -> w[0]
#pragma unroll
        for (int i = m / 2; i < m - 1; ++i) cmpxchg(w[i], w[m - 1]);   // maximum -> w[m - 1]
        if (m > 3) w[0] = v[W0 + (W0 - m) < N ? W0 + (W0 - m) : N - 1];  // drop both, admit the next sample
    }
    return w[1];
}

template <typename T, int N>
__global__ void __launch_bounds__(256) median_small_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                           const int2* __restrict__ offs_g, int ry, int rx, int mode,
                                                           T cval) {
    typedef typename key_traits<T>::key_t key_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int pitch = RT_W + 2 * rx;
    const int rows = RT_H + 2 * ry;
    key_t* tile = reinterpret_cast<key_t*>(smem_raw);
    int* xmap = reinterpret_cast<int*>(smem_raw + amt_align((size_t)rows * pitch * sizeof(key_t), 16));
    int* ymap = xmap + pitch;
    const int x0 = blockIdx.x * RT_W, y0 = blockIdx.y * RT_H;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < pitch; i += 256) xmap[i] = amt_map_index(x0 - rx + i, W, mode);
    for (int i = threadIdx.x; i < rows; i += 256) ymap[i] = amt_map_index(y0 - ry + i, H, mode);
    __syncthreads();
    // staging: wave w owns tile rows w, w + 4, ...; 64 columns of up to 8 rows in flight
    for (int k0 = 0; k0 < pitch; k0 += 64) {
        const int kx = k0 + lane;
        const int xx = kx < pitch ? xmap[kx] : -1;
        for (int r0 = wave; r0 < rows; r0 += 32) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ky = r0 + 4 * u;
                const int yy = ky < rows ? ymap[ky] : -1;
                v[u] = (xx >= 0 && yy >= 0) ? in[plane + (size_t)yy * W + xx] : cval;
            }
            if (kx < pitch) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + 4 * u < rows) tile[(r0 + 4 * u) * pitch + kx] = key_traits<T>::to_key(v[u]);
            }
        }
    }
    __syncthreads();
    int off[N];
#pragma unroll
    for (int k = 0; k < N; ++k) off[k] = offs_g[k].y * pitch + offs_g[k].x;  // uniform: scalar loads
    for (int i = threadIdx.x; i < RT_H * RT_W; i += 256) {
        const int ky = i / RT_W, kx = i - ky * RT_W;
        const int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        const key_t* c = tile + (ky + ry) * pitch + (kx + rx);
        key_t v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = c[off[k]];
        out[plane + (size_t)y * W + x] = key_traits<T>::from_key(median_select<key_t, N>(v));
    }
}

template <typename T>
static bool launch_median_small(amt_ctx* ctx, const T* in, T* out, int nplanes, int H, int W, const int2* offs, int noffs,
                                int ry, int rx, int mode, T cval) {
    dim3 grid((W + RT_W - 1) / RT_W, (H + RT_H - 1) / RT_H, nplanes);
    const size_t ksz = sizeof(typename key_traits<T>::key_t);
    const size_t smem = amt_align((size_t)(RT_H + 2 * ry) * (RT_W + 2 * rx) * ksz, 16) +
                        (size_t)(RT_W + 2 * rx + RT_H + 2 * ry) * sizeof(int);
    switch (noffs) {
#define AMT_MEDIAN_CASE(NN)                                                                                          \
    case NN:                                                                                                         \
        hipLaunchKernelGGL((median_small_kernel<T, NN>), grid, dim3(256), smem, ctx->stream, in, out, H, W, offs, ry, \
                           rx, mode, cval);                                                                          \
        return true;
        AMT_MEDIAN_CASE(5)
        AMT_MEDIAN_CASE(9)
        AMT_MEDIAN_CASE(13)
        AMT_MEDIAN_CASE(21)
        AMT_MEDIAN_CASE(25)
        AMT_MEDIAN_CASE(29)
        AMT_MEDIAN_CASE(37)
        AMT_MEDIAN_CASE(45)
        AMT_MEDIAN_CASE(49)
#undef AMT_MEDIAN_CASE
        default:
            return false;
    }
}

// ---- erosion / dilation over footprints whose rows are contiguous runs (disks, squares, diamonds, ...) ------
// min / max over a run of length L = op of two overlapping power-of-two windows (sparse table): level k of the
// LDS tile holds M_k[r][x] = op over [x, x + 2^k).  A footprint row then costs two LDS reads instead of L; the
// levels are built once per tile (one pass per level).
struct run3 {
    int dy, lo, hi;  // offsets dx in [lo, hi] at row offset dy (already mirrored for dilation)
};
constexpr int RUN_MAX_ROWS = 63;

template <typename T, bool ISMAX>
__global__ void __launch_bounds__(256) minmax_runs_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                          const run3* __restrict__ runs_g, int nruns, int ry, int rx,
                                                          int kmax, int mode, T cval) {
    typedef typename key_traits<T>::key_t key_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int pitch = RT_W + 2 * rx;
    const int rows = RT_H + 2 * ry;
    const int tsz = rows * pitch;
    key_t* lev = reinterpret_cast<key_t*>(smem_raw);  // (kmax + 1) levels of tsz keys
    int* xmap = reinterpret_cast<int*>(smem_raw + amt_align((size_t)(kmax + 1) * tsz * sizeof(key_t), 16));
    int* ymap = xmap + pitch;
    __shared__ run3 runs[RUN_MAX_ROWS];
    const int x0 = blockIdx.x * RT_W, y0 = blockIdx.y * RT_H;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nruns; i += 256) runs[i] = runs_g[i];
    for (int i = threadIdx.x; i < pitch; i += 256) xmap[i] = amt_map_index(x0 - rx + i, W, mode);
    for (int i = threadIdx.x; i < rows; i += 256) ymap[i] = amt_map_index(y0 - ry + i, H, mode);
    __syncthreads();
    for (int k0 = 0; k0 < pitch; k0 += 64) {
        const int kx = k0 + lane;
        const int xx = kx < pitch ? xmap[kx] : -1;
        for (int r0 = wave; r0 < rows; r0 += 32) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ky = r0 + 4 * u;
                const int yy = ky < rows ? ymap[ky] : -1;
                v[u] = (xx >= 0 && yy >= 0) ? in[plane + (size_t)yy * W + xx] : cval;
            }
            if (kx < pitch) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + 4 * u < rows) lev[(r0 + 4 * u) * pitch + kx] = key_traits<T>::to_key(v[u]);
            }
        }
    }
    __syncthreads();
    for (int k = 1; k <= kmax; ++k) {  // M_k[x] = op(M_{k-1}[x], M_{k-1}[x + 2^(k-1)]) (clamped at the row end)
        const key_t* src = lev + (size_t)(k - 1) * tsz;
        key_t* dst = lev + (size_t)k * tsz;
        const int h = 1 << (k - 1);
        for (int i = threadIdx.x; i < tsz; i += 256) {
            const int r = i / pitch, x = i - r * pitch;
            const key_t a = src[i];
            const key_t b = x + h < pitch ? src[i + h] : a;
            dst[i] = ISMAX ? (a > b ? a : b) : (a < b ? a : b);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < RT_H * RT_W; i += 256) {
        const int ky = i / RT_W, kx = i - ky * RT_W;
        const int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        key_t r = ISMAX ? (key_t)0 : ~(key_t)0;
        for (int q = 0; q < nruns; ++q) {
            const run3 rn = runs[q];
            const int len = rn.hi - rn.lo + 1;
            const int k = 31 - __clz(len);
            const key_t* row = lev + (size_t)k * tsz + (ky + ry + rn.dy) * pitch + (kx + rx);
            const key_t a = row[rn.lo], b = row[rn.hi - (1 << k) + 1];
            const key_t m = ISMAX ? (a > b ? a : b) : (a < b ? a : b);
            r = ISMAX ? (m > r ? m : r) : (m < r ? m : r);
        }
        out[plane + (size_t)y * W + x] = key_traits<T>::from_key(r);
    }
}

extern "C" int amt_rank_filter(amt_ctx* ctx, const void* in, void* out, int dtype, int nplanes, int H, int W,
                               const uint8_t* footprint, int fh, int fw, int op, int mode, double cval) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && footprint && nplanes >= 0 && H > 0 && W > 0, "rank_filter: bad arguments");
    AMT_REQUIRE(in != out, "rank_filter: in-place operation is not supported");
    AMT_REQUIRE(dtype == AMT_U16 || dtype == AMT_F64, "rank_filter: dtype must be AMT_U16 or AMT_F64");
    AMT_REQUIRE(op >= 0 && op <= 2, "rank_filter: op must be 0 (erosion), 1 (dilation) or 2 (median)");
    AMT_REQUIRE((fh & 1) && (fw & 1) && fh >= 1 && fw >= 1 && fh <= 63 && fw <= 63,
                "rank_filter: footprint must be odd-sized and at most 63 x 63 (got %d x %d)", fh, fw);
    static thread_local int2 host[RK_MAX_OFFS];
    int noffs = 0;
    for (int y = 0; y < fh; ++y)
        for (int x = 0; x < fw; ++x)
            if (footprint[y * fw + x]) {
                AMT_REQUIRE(noffs < RK_MAX_OFFS, "rank_filter: footprint has more than %d cells", RK_MAX_OFFS);
                host[noffs].x = x - fw / 2;
                host[noffs].y = y - fh / 2;
                ++noffs;
            }
    AMT_REQUIRE(noffs > 0, "rank_filter: empty footprint");
    if (nplanes == 0) return AMT_OK;
    const int ry = fh / 2, rx = fw / 2;
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(int2) * noffs)));
    int2* offs = (int2*)amt_arena_take(ctx, sizeof(int2) * noffs);
    AMT_TRY(amt_param_upload(ctx, offs, host, sizeof(int2) * noffs));
    if (op <= 1 && noffs >= 21) {  // erosion / dilation: rows of the footprint that are contiguous runs
        static thread_local run3 hruns[RUN_MAX_ROWS];
        int nruns = 0, maxlen = 1;
        bool ok = true;
        for (int y = 0; y < fh && ok; ++y) {
            int lo = -1, hi = -1, cnt = 0;
            for (int x = 0; x < fw; ++x)
                if (footprint[y * fw + x]) {
                    if (lo < 0) lo = x;
                    hi = x;
                    ++cnt;
                }
            if (cnt == 0) continue;
            if (hi - lo + 1 != cnt) ok = false;  // holes in this row: generic kernel
            const int dy = y - fh / 2, a = lo - fw / 2, b = hi - fw / 2;
            // erosion reads in[p + s]; dilation reads in[p - s]
            hruns[nruns].dy = op == 0 ? dy : -dy;
            hruns[nruns].lo = op == 0 ? a : -b;
            hruns[nruns].hi = op == 0 ? b : -a;
            ++nruns;
            maxlen = cnt > maxlen ? cnt : maxlen;
        }
        int kmax = 0;
        while ((2 << kmax) <= maxlen) ++kmax;
        const size_t ksz2 = dtype == AMT_U16 ? sizeof(unsigned) : sizeof(unsigned long long);
        const size_t tsz = (size_t)(RT_H + 2 * ry) * (RT_W + 2 * rx);
        const size_t smem2 = amt_align((size_t)(kmax + 1) * tsz * ksz2, 16) +
                             (size_t)(RT_W + 2 * rx + RT_H + 2 * ry) * sizeof(int);
        if (ok && smem2 <= 96 * 1024) {
            AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(run3) * nruns)));
            run3* druns = (run3*)amt_arena_take(ctx, sizeof(run3) * nruns);
            AMT_TRY(amt_param_upload(ctx, druns, hruns, sizeof(run3) * nruns));
            dim3 grid2((W + RT_W - 1) / RT_W, (H + RT_H - 1) / RT_H, nplanes);
            if (dtype == AMT_U16) {
                if (op == 0)
                    hipLaunchKernelGGL((minmax_runs_kernel<uint16_t, false>), grid2, dim3(256), smem2, ctx->stream,
                                       (const uint16_t*)in, (uint16_t*)out, H, W, druns, nruns, ry, rx, kmax, mode,
                                       (uint16_t)cval);
                else
                    hipLaunchKernelGGL((minmax_runs_kernel<uint16_t, true>), grid2, dim3(256), smem2, ctx->stream,
                                       (const uint16_t*)in, (uint16_t*)out, H, W, druns, nruns, ry, rx, kmax, mode,
                                       (uint16_t)cval);
            } else {
                if (op == 0)
                    hipLaunchKernelGGL((minmax_runs_kernel<double, false>), grid2, dim3(256), smem2, ctx->stream,
                                       (const double*)in, (double*)out, H, W, druns, nruns, ry, rx, kmax, mode, cval);
                else
                    hipLaunchKernelGGL((minmax_runs_kernel<double, true>), grid2, dim3(256), smem2, ctx->stream,
                                       (const double*)in, (double*)out, H, W, druns, nruns, ry, rx, kmax, mode, cval);
            }
            AMT_LAUNCH_CHECK();
            return AMT_OK;
        }
    }
    if (op == 2) {  // median over a small footprint: register selection
        const bool done = dtype == AMT_U16
                              ? launch_median_small<uint16_t>(ctx, (const uint16_t*)in, (uint16_t*)out, nplanes, H, W, offs,
                                                              noffs, ry, rx, mode, (uint16_t)cval)
                              : launch_median_small<double>(ctx, (const double*)in, (double*)out, nplanes, H, W, offs,
                                                            noffs, ry, rx, mode, cval);
        if (done) {
            AMT_LAUNCH_CHECK();
            return AMT_OK;
        }
    }
    dim3 grid((W + RT_W - 1) / RT_W, (H + RT_H - 1) / RT_H, nplanes);
    const size_t ksz = dtype == AMT_U16 ? sizeof(unsigned) : sizeof(unsigned long long);
    size_t smem = amt_align((size_t)(RT_H + 2 * ry) * (RT_W + 2 * rx) * ksz, 16) + sizeof(int2) * noffs;
    if (dtype == AMT_U16)
        hipLaunchKernelGGL((rank_kernel<uint16_t>), grid, dim3(256), smem, ctx->stream, (const uint16_t*)in,
                           (uint16_t*)out, H, W, offs, noffs, ry, rx, op, mode, (uint16_t)cval);
    else
        hipLaunchKernelGGL((rank_kernel<double>), grid, dim3(256), smem, ctx->stream, (const double*)in, (double*)out,
                           H, W, offs, noffs, ry, rx, op, mode, cval);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
