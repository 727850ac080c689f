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
#include "amt_internal.h"

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

// ---- uint16 erosion / dilation over small symmetric run footprints (disk(1..7), squares, diamonds, crosses) --------
// The footprint's rows are runs centred on the origin, so  out[y][x] = op over dy of H_{h(dy)}[y + dy][x]  with
// H_h[y][x] = op over in[y][x - h .. x + h]: a horizontal pass builds H_h for the distinct half-widths h (cumulatively:
// H_k = op(H_{k-1}, in[x - k], in[x + k])), a vertical pass combines 2 ry + 1 of those rows.  Both passes work on 8
// pixels per lane as four packed pairs (v_pk_min_u16 / v_pk_max_u16; a shift by an odd number of columns is one
// v_alignbit), 16-byte LDS reads and writes, 16-byte HBM loads and stores.  Versus the sparse-table kernel above (one
// pixel per lane, 32-bit keys): grey erosion disk(2) 597 -> ~150 us and top-hat disk(7) 4.3 -> ~1 ms per 32 planes.
typedef unsigned short mm_u16x2 __attribute__((ext_vector_type(2)));
// tile = MM_TH rows x MM_TW columns (template parameters): occupancy decides -- measured for disk(2), per 32 planes:
// 32 x 128 (31 KiB of LDS, 5 blocks per CU) 299 us, 32 x 256 (59 KiB, 2 blocks) 535 us
struct mm_params {
    int ry, ns, hmax;
    int hws[8];   // distinct half-widths, hws[0] is the smallest; array s holds H_{hws[s]}
    int sel[15];  // sel[dy + ry] = array index for footprint row dy, -1 = no cell in that row
    // the same, resolved for the kernel (every index it uses is a compile-time constant after unrolling, so the
    // fields are scalar registers -- a search through hws[] per step was a chain of dependent scalar loads):
    int store_at[8];  // store_at[k] = LDS array (0 = first of arrs) that receives H_k, -1 = H_k is not needed
    int row_arr[15];  // row_arr[dy + ry] = -1 no cell, 0 = the input tile (half-width 0), a + 1 = arrs[a]
    int row_h[15];    // row_h[dy + ry] = half-width of footprint row dy, -1 = no cell (register kernel)
};

template <bool ISMAX>
__device__ __forceinline__ unsigned mm_pk(unsigned a, unsigned b) {
    mm_u16x2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const mm_u16x2 r = ISMAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    unsigned o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}

template <bool ISMAX, int MM_TH, int MM_TW>
__global__ void __launch_bounds__(256) mm_u16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int H,
                                                     int W, mm_params P, int mode, uint16_t cval) {
    constexpr int MM_PITCH = MM_TW + 16, MM_G = MM_PITCH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int rows = MM_TH + 2 * P.ry;
    const size_t asz = (size_t)rows * MM_PITCH;  // u16 elements per array
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem_raw);  // the input tile = H_0
    uint16_t* arrs = tile + asz;                              // P.ns arrays H_{hws[s]} (hws[s] > 0)
    const int x0 = blockIdx.x * MM_TW, y0 = blockIdx.y * MM_TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const bool aligned_w = (W & 7) == 0;
    // ---- load: 8 columns (16 bytes) per task; the tile starts 8 columns left of x0 (16-byte aligned when W % 8 == 0).
    // Four tasks per thread and round: all wide loads are issued before the first LDS write (a load -> store loop
    // waits for every load in turn); tasks that need the element-wise path (halo beyond the image, unaligned widths)
    // are done in a second sweep
    const int ntask = rows * MM_G;
    for (int t0 = threadIdx.x; t0 < ntask; t0 += 256 * 4) {
        uint4 v[4];
        int slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tsk = t0 + u * 256;
            const int row = tsk / MM_G, g = tsk - row * MM_G;
            const int yy = tsk < ntask ? amt_map_index(y0 - P.ry + row, H, mode) : -1;
            const int xg = x0 - 8 + 8 * g;
            const bool wide = yy >= 0 && aligned_w && xg >= 0 && xg + 7 < W;
            slot[u] = tsk < ntask ? (wide ? tsk : -2 - tsk) : -1;
            // unconditional load from a clamped address; the value is used only if `wide`
            const size_t off = wide ? plane + (size_t)yy * W + xg : plane;
            v[u] = *reinterpret_cast<const uint4*>(in + (off & ~(size_t)7));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (slot[u] >= 0) {
                const int row = slot[u] / MM_G, g = slot[u] - row * MM_G;
                *reinterpret_cast<uint4*>(tile + (size_t)row * MM_PITCH + 8 * g) = v[u];
            } else if (slot[u] <= -2) {
                const int tsk = -2 - slot[u];
                const int row = tsk / MM_G, g = tsk - row * MM_G;
                const int yy = amt_map_index(y0 - P.ry + row, H, mode);
                const int xg = x0 - 8 + 8 * g;
                unsigned short e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int xx = amt_map_index(xg + k, W, mode);
                    e[k] = (yy >= 0 && xx >= 0) ? in[plane + (size_t)yy * W + xx] : cval;
                }
                *reinterpret_cast<uint4*>(tile + (size_t)row * MM_PITCH + 8 * g) =
                    make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
                               e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
            }
        }
    }
    __syncthreads();
    // ---- horizontal: H_k for k = 1 .. hmax, stored for the k that the footprint uses
    if (P.hmax > 0) {
        for (int tsk = threadIdx.x; tsk < rows * (MM_G - 2); tsk += 256) {
            const int row = tsk / (MM_G - 2), g = 1 + (tsk - row * (MM_G - 2));
            const uint16_t* src = tile + (size_t)row * MM_PITCH + 8 * g;
            unsigned p[12];
            const uint4 a = *reinterpret_cast<const uint4*>(src - 8), b = *reinterpret_cast<const uint4*>(src),
                        c = *reinterpret_cast<const uint4*>(src + 8);
            p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w;
            p[4] = b.x; p[5] = b.y; p[6] = b.z; p[7] = b.w;
            p[8] = c.x; p[9] = c.y; p[10] = c.z; p[11] = c.w;
            unsigned acc[4] = {p[4], p[5], p[6], p[7]};
#pragma unroll
            for (int k = 1; k <= 7; ++k) {
                if (k > P.hmax) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned lft, rgt;  // columns (2i - k, 2i - k + 1) and (2i + k, 2i + k + 1) of the centre group
                    if (k & 1) {
                        const int al = 4 + i - (k + 1) / 2, ar = 4 + i + (k - 1) / 2;
                        lft = __builtin_amdgcn_alignbit(p[al + 1], p[al], 16);
                        rgt = __builtin_amdgcn_alignbit(p[ar + 1], p[ar], 16);
                    } else {
                        lft = p[4 + i - k / 2];
                        rgt = p[4 + i + k / 2];
                    }
                    acc[i] = mm_pk<ISMAX>(acc[i], mm_pk<ISMAX>(lft, rgt));
                }
                if (P.store_at[k] >= 0)  // uniform
                    *reinterpret_cast<uint4*>(arrs + (size_t)P.store_at[k] * asz + (size_t)row * MM_PITCH + 8 * g) =
                        make_uint4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        __syncthreads();
    }
    // ---- vertical: combine the footprint's rows; 8 outputs per task
    for (int tsk = threadIdx.x; tsk < MM_TH * (MM_TW / 8); tsk += 256) {
        const int r = tsk / (MM_TW / 8), g = 1 + (tsk - r * (MM_TW / 8));
        const int y = y0 + r, x = x0 + 8 * (g - 1);
        if (y >= H || x >= W) continue;
        unsigned acc[4];
        const unsigned ident = ISMAX ? 0u : 0xFFFFFFFFu;
        acc[0] = acc[1] = acc[2] = acc[3] = ident;
#pragma unroll
        for (int di = 0; di < 15; ++di) {
            if (di > 2 * P.ry) break;
            const int ra = P.row_arr[di];  // uniform, a scalar register
            if (ra < 0) continue;
            const uint16_t* base = ra == 0 ? tile : arrs + (size_t)(ra - 1) * asz;
            const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)(r + di) * MM_PITCH + 8 * g);
            acc[0] = mm_pk<ISMAX>(acc[0], v.x);
            acc[1] = mm_pk<ISMAX>(acc[1], v.y);
            acc[2] = mm_pk<ISMAX>(acc[2], v.z);
            acc[3] = mm_pk<ISMAX>(acc[3], v.w);
        }
        uint16_t* dst = out + plane + (size_t)y * W + x;
        if (aligned_w && x + 7 < W) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (x + k < W) dst[k] = (uint16_t)((acc[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
        }
    }
}

// ---- the same filter with NO LDS and no barrier: a wave slides down a strip of 64 x 8 columns -------------------
// Each lane owns 8 consecutive pixels of the row (one 16-byte load; see mm_geo for the strip seams), the neighbouring groups
// come from the adjacent lanes by DPP wave shifts, the run minima H_k are built in registers, and every input row r
// is folded straight into the 2 RY + 1 output rows it belongs to (out[y] gets H_{h(r - y)}[r]): the accumulators of
// the pending output rows and the rows loaded ahead live in registers whose indices are compile-time constants
// because the row loop is unrolled by 2 RY + 1.  Rows are requested 2 RY + 1 steps ahead, so a wave keeps ~(2 RY + 1)
// KiB in flight -- the tile kernel above (load -> barrier -> LDS -> barrier -> store) kept ~15 KiB per CU and ran
// at 2 TB/s whatever the footprint (tools/mm_probe.py: 244 us per 32 planes for a 1 x 1 footprint).
__device__ __forceinline__ uint4 mm_lane_left(uint4 v) {
    return make_uint4((unsigned)amt_lane_left((int)v.x), (unsigned)amt_lane_left((int)v.y),
                      (unsigned)amt_lane_left((int)v.z), (unsigned)amt_lane_left((int)v.w));
}
__device__ __forceinline__ uint4 mm_lane_right(uint4 v) {
    return make_uint4((unsigned)amt_lane_right((int)v.x), (unsigned)amt_lane_right((int)v.y),
                      (unsigned)amt_lane_right((int)v.z), (unsigned)amt_lane_right((int)v.w));
}
// the same shifts, but the lane without a source (lane 0 / lane 63) keeps `old` (DPP without bound_ctrl)
__device__ __forceinline__ uint4 mm_lane_left_old(uint4 old, uint4 v) {
    return make_uint4((unsigned)__builtin_amdgcn_update_dpp((int)old.x, (int)v.x, 0x138, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.y, (int)v.y, 0x138, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.z, (int)v.z, 0x138, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.w, (int)v.w, 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint4 mm_lane_right_old(uint4 old, uint4 v) {
    return make_uint4((unsigned)__builtin_amdgcn_update_dpp((int)old.x, (int)v.x, 0x130, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.y, (int)v.y, 0x130, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.z, (int)v.z, 0x130, 0xf, 0xf, false),
                      (unsigned)__builtin_amdgcn_update_dpp((int)old.w, (int)v.w, 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned mm_swap16(unsigned a) { return __builtin_amdgcn_alignbit(a, a, 16); }

// A wave owns a strip of 64 x 8 = 512 columns, every lane one group of 8 pixels.  The groups left of lane 0 and right of
// lane 63 belong to the neighbouring strips: those two lanes fetch them with one more (exec-masked) load per row, and
// the DPP shifts hand them over as the `old` value of the lane that has no source.  At the image edges the missing
// group is the boundary extension of the edge group itself (reflect: the group reversed; nearest: its edge pixel;
// constant: cval); in a partial last strip the lane just beyond the edge carries that extension for its left neighbour.
struct mm_geo {
    int xg;               // first column of this lane's group
    bool inside;          // the group lies in the image
    bool first_g, last_g; // the image's first / last group
    bool beyond;          // the group just right of the image (partial last strip)
    bool need_h;          // lane 0 / 63 with a real neighbouring strip
    bool edge_wave;       // uniform: first or last strip
    unsigned lane_off, halo_off;  // byte offsets inside a row
};
__device__ __forceinline__ mm_geo mm_geometry(int strip, int nstrips, int lane, int W) {
    mm_geo g;
    g.xg = strip * 512 + 8 * lane;
    g.inside = g.xg + 7 < W;
    g.first_g = g.xg == 0;
    g.last_g = g.xg + 8 == W;
    g.beyond = g.xg == W;
    const bool hl = lane == 0 && strip > 0, hr = lane == 63 && g.xg + 8 < W;
    g.need_h = hl || hr;
    g.edge_wave = strip == 0 || strip == nstrips - 1;
    g.lane_off = (unsigned)(g.inside ? g.xg : 0) * 2u;
    g.halo_off = (unsigned)(hl ? g.xg - 8 : (hr ? g.xg + 8 : 0)) * 2u;
    return g;
}
// boundary extension on the left (of the first group s) / right (of the last group s)
__device__ __forceinline__ uint4 mm_extend(uint4 s, bool left, int mode, unsigned cv2) {
    if (mode == AMT_MODE_REFLECT) return make_uint4(mm_swap16(s.w), mm_swap16(s.z), mm_swap16(s.y), mm_swap16(s.x));
    if (mode == AMT_MODE_NEAREST) {
        const unsigned b = left ? ((s.x & 0xFFFFu) | (s.x << 16)) : ((s.w >> 16) | (s.w & 0xFFFF0000u));
        return make_uint4(b, b, b, b);
    }
    return make_uint4(cv2, cv2, cv2, cv2);
}
// lf / rt = the groups left / right of every lane's own group; `cur` is rewritten in the lane beyond the right edge
__device__ __forceinline__ void mm_neighbours(const mm_geo& g, int lane, uint4& cur, uint4 hal, int mode, unsigned cv2,
                                              uint4& lf, uint4& rt) {
    if (g.edge_wave) {
        const uint4 nl = mm_lane_left(cur);
        if (g.beyond) cur = mm_extend(nl, false, mode, cv2);
        if (g.first_g) hal = mm_extend(cur, true, mode, cv2);
        if (g.last_g && lane == 63) hal = mm_extend(cur, false, mode, cv2);
    }
    lf = mm_lane_left_old(hal, cur);
    rt = mm_lane_right_old(hal, cur);
}

// SHAPE: 0 = half-widths read from the kernel arguments (any centred-run footprint), 1 = disk(RY), 2 = rectangle
// (2 RY + 1 rows of one half-width P.hmax, folded once after the horizontal loop):
// for the disks every half-width is a compile-time constant, which keeps the 15-fold unrolled body of disk(7)
// inside the instruction cache (the run-time form compares every footprint row against every k: 72 KiB of code and
// 2.4 ms per 32 planes for disk(7); the constexpr form ~25 KiB).
__host__ __device__ constexpr int mm_isqrt(int v) {
    int r = 0;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}
__host__ __device__ constexpr int mm_shape_h(int shape, int ry, int dy) {
    return shape == 1 ? mm_isqrt(ry * ry - dy * dy) : ry;  // disk: x^2 + dy^2 <= ry^2 ; square: ry
}
__host__ __device__ constexpr int mm_shape_hmax(int shape, int ry) { return ry; }

// two uint16 differences (wrapping, as numpy's uint16 subtraction)
__device__ __forceinline__ unsigned mm_pk_sub(unsigned a, unsigned b) {
    mm_u16x2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const mm_u16x2 r = x - y;
    unsigned o;
    __builtin_memcpy(&o, &r, 4);
    return o;
}

// SUB: the stored value is minuend - result (white top-hat = image - dilation(erosion(image)): no separate subtraction
// pass); the minuend's rows travel through a register ring of their own, requested as far ahead as the input rows.
template <bool ISMAX, int RY, int SHAPE, bool SUB>
__global__ void __launch_bounds__(256) mmr_u16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int H,
                                                      int W, mm_params P, int mode, uint16_t cval, int seg_rows,
                                                      int nstrips, int nsegs, const uint16_t* __restrict__ minuend) {
    // requires W % 8 == 0, H > 2 RY + 8 and mode in {reflect, nearest, constant} (the host checks): the groups beyond
    // the left / right image edge are then functions of the neighbouring lane's group and need no loads of their own
    constexpr int PER = 2 * RY + 1;               // pending output rows
    constexpr int U = PER * ((8 + PER - 1) / PER);  // unroll factor: a multiple of PER that is >= 8
    constexpr int D = 8;                          // rows requested ahead (8 KiB per wave in flight), D <= U
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave id inside the plane
    if (wid >= nstrips * nsegs) return;                   // whole waves only: the DPP shifts need all 64 lanes
    const int strip = wid % nstrips, seg = wid / nstrips;
    const size_t plane = (size_t)blockIdx.y * H * W;
    const mm_geo g = mm_geometry(strip, nstrips, lane, W);
    const int xg = g.xg;
    const int y_begin = __builtin_amdgcn_readfirstlane(seg * seg_rows);
    const int y_end = min(H, y_begin + seg_rows);
    const bool inside = g.inside;
    const unsigned ident = ISMAX ? 0u : 0xFFFFFFFFu;
    const unsigned cv2 = (unsigned)cval | ((unsigned)cval << 16);
    const unsigned lane_off = g.lane_off;  // byte offset inside a row
    const char* const in_plane = reinterpret_cast<const char*>(in + plane);
    const int refl = mode == AMT_MODE_REFLECT;
    // unconditional load from a boundary-mapped, clamped row (scalar arithmetic, no branches); rows outside the image in
    // 'constant' mode and groups outside the image are replaced where the value is USED
    auto load_row = [&](int r) -> uint4 {
        const int lo = refl ? -r - 1 : 0, hi = refl ? 2 * H - 1 - r : H - 1;
        int yy = r < 0 ? lo : (r >= H ? hi : r);
        yy = min(max(yy, 0), H - 1);  // rows requested past the segment's last needed row are never used
        const char* rowp = in_plane + (size_t)yy * W * 2;
        return *reinterpret_cast<const uint4*>(rowp + lane_off);
    };
    // the neighbouring strips' edge groups (lanes 0 / 63 only), requested DH rows ahead
    constexpr int DH = 4;
    auto load_halo = [&](int r) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g.need_h) {
            const int lo = refl ? -r - 1 : 0, hi = refl ? 2 * H - 1 - r : H - 1;
            int yy = r < 0 ? lo : (r >= H ? hi : r);
            yy = min(max(yy, 0), H - 1);
            v = *reinterpret_cast<const uint4*>(in_plane + (size_t)yy * W * 2 + g.halo_off);
        }
        return v;
    };
    uint4 inreg[U], halreg[U], acc[PER];
#pragma unroll
    for (int u = 0; u < D; ++u) inreg[u] = load_row(y_begin - RY + u);
#pragma unroll
    for (int u = 0; u < DH; ++u) halreg[u] = load_halo(y_begin - RY + u);
#pragma unroll
    for (int u = 0; u < PER; ++u) acc[u] = make_uint4(ident, ident, ident, ident);
    const bool can_store = inside;
    uint16_t* const out_lane = out + plane + (inside ? xg : 0);
    // minuend row of output row yo = r - RY, requested D steps before that row is stored
    const char* const sub_plane = reinterpret_cast<const char*>(SUB ? minuend + plane : in);
    auto load_sub = [&](int yo) -> uint4 {
        const int yy = min(max(yo, 0), H - 1);
        return *reinterpret_cast<const uint4*>(sub_plane + (size_t)yy * W * 2 + lane_off);
    };
    uint4 subreg[SUB ? U : 1];
    if (SUB) {
#pragma unroll
        for (int u = 0; u < D; ++u) subreg[u % (SUB ? U : 1)] = load_sub(y_begin - 2 * RY + u);
    }
    const int hmax = SHAPE == 1 ? mm_shape_hmax(SHAPE, RY) : P.hmax;
    for (int base = y_begin - RY; base < y_end + RY; base += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u;
            uint4 cur = inreg[u];
            inreg[(u + D) % U] = load_row(r + D);
            uint4 subv = make_uint4(0, 0, 0, 0);
            if (SUB) {
                subv = subreg[u % (SUB ? U : 1)];
                subreg[(u + D) % (SUB ? U : 1)] = load_sub(r - RY + D);
            }
            uint4 hal = halreg[u];
            halreg[(u + DH) % U] = load_halo(r + DH);
            if (mode == AMT_MODE_CONSTANT && (r < 0 || r >= H)) {
                cur = make_uint4(cv2, cv2, cv2, cv2);
                hal = cur;
            }
            unsigned p[12];
            if (hmax > 0) {
                uint4 lf, rt;
                mm_neighbours(g, lane, cur, hal, mode, cv2, lf, rt);
                p[0] = lf.x, p[1] = lf.y, p[2] = lf.z, p[3] = lf.w;
                p[8] = rt.x, p[9] = rt.y, p[10] = rt.z, p[11] = rt.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = p[8 + i] = ident;
            }
            p[4] = cur.x, p[5] = cur.y, p[6] = cur.z, p[7] = cur.w;
            unsigned hk[4] = {p[4], p[5], p[6], p[7]};
            // fold H_k of row r into the output rows y = r - dy whose footprint row dy has half-width k
            auto fold = [&](int k) {
#pragma unroll
                for (int di = 0; di < PER; ++di) {
                    const int hsel = SHAPE == 1 ? mm_shape_h(1, RY, di - RY) : (SHAPE == 2 ? -2 : P.row_h[di]);
                    if (hsel != k) continue;
                    const int j = ((u - (di - RY)) % PER + PER) % PER;
                    acc[j].x = mm_pk<ISMAX>(acc[j].x, hk[0]);
                    acc[j].y = mm_pk<ISMAX>(acc[j].y, hk[1]);
                    acc[j].z = mm_pk<ISMAX>(acc[j].z, hk[2]);
                    acc[j].w = mm_pk<ISMAX>(acc[j].w, hk[3]);
                    // the accumulators feed only the (conditional) store: without this the compiler sinks every fold
                    // into the store's branch and keeps the H_k of the last 2 RY + 1 rows alive instead (256 VGPRs)
                    asm volatile("" : "+v"(acc[j].x), "+v"(acc[j].y), "+v"(acc[j].z), "+v"(acc[j].w));
                }
            };
            if (SHAPE != 2) fold(0);
#pragma unroll
            for (int k = 1; k <= 7; ++k) {
                if (k > hmax) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned lft, rgt;
                    if (k & 1) {
                        const int al = 4 + i - (k + 1) / 2, ar = 4 + i + (k - 1) / 2;
                        lft = __builtin_amdgcn_alignbit(p[al + 1], p[al], 16);
                        rgt = __builtin_amdgcn_alignbit(p[ar + 1], p[ar], 16);
                    } else {
                        lft = p[4 + i - k / 2];
                        rgt = p[4 + i + k / 2];
                    }
                    hk[i] = mm_pk<ISMAX>(hk[i], mm_pk<ISMAX>(lft, rgt));
                }
                if (SHAPE != 2) fold(k);
            }
            if (SHAPE == 2) fold(-2);  // every row of a rectangle takes H_hmax
            // output row r - RY has now seen all its input rows
            const int yo = r - RY;
            const int jo = ((u - RY) % PER + PER) % PER;
            uint4 res = acc[jo];
            if (SUB) {
                res.x = mm_pk_sub(subv.x, res.x), res.y = mm_pk_sub(subv.y, res.y);
                res.z = mm_pk_sub(subv.z, res.z), res.w = mm_pk_sub(subv.w, res.w);
            }
            if (yo >= y_begin && yo < y_end && can_store) *reinterpret_cast<uint4*>(out_lane + (size_t)yo * W) = res;
            acc[jo] = make_uint4(ident, ident, ident, ident);
        }
    }
}

// ---- median over a small footprint with the same register-strip skeleton -----------------------------------------
// Footprints whose rows are centred runs, symmetric top / bottom, <= 5 x 5 (3 x 3, cross, disk(2), 5 x 5).  The last
// 2 RY + 1 rows stay in a register ring (the lane's four pixel-pair words + one word of either neighbour lane), the
// samples of an output pair are register moves (dx = +-2) or one v_alignbit (dx = +-1), and the median of the N samples
// of each pixel pair is a packed forgetful selection (window of N / 2 + 2, min and max leave, the next sample enters):
// 39 compare-exchanges = 78 v_pk_min / v_pk_max_u16 per two pixels for disk(2).
__device__ __forceinline__ void med_ce(unsigned& a, unsigned& b) {
    const unsigned lo = mm_pk<false>(a, b), hi = mm_pk<true>(a, b);
    a = lo;
    b = hi;
}
// minimum of v[LO .. LO + S - 1] to v[LO], maximum to v[LO + S - 1]
template <int LO, int S, int NV>
__device__ __forceinline__ void med_minmax(unsigned (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < S / 2; ++i) med_ce(v[LO + i], v[LO + S - 1 - i]);
    constexpr int HL = (S + 1) / 2;  // the middle element of an odd window belongs to both halves
#pragma unroll
    for (int i = 1; i < HL; ++i) med_ce(v[LO], v[LO + i]);
#pragma unroll
    for (int i = S - HL; i < S - 1; ++i) med_ce(v[LO + i], v[LO + S - 1]);
}
template <int LO, int S, int NEXT, int NV>
__device__ __forceinline__ unsigned med_step(unsigned (&v)[NV]) {
    if constexpr (S <= 1) {
        return v[LO];
    } else {
        med_minmax<LO, S, NV>(v);
        if constexpr (NEXT < NV) {
            v[LO + S - 1] = v[NEXT];
            return med_step<LO + 1, S - 1, NEXT + 1, NV>(v);
        } else {
            return med_step<LO + 1, S - 2, NEXT, NV>(v);
        }
    }
}

template <int RY, int H0, int H1, int H2>
__global__ void __launch_bounds__(256) medr_u16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int H,
                                                       int W, int mode, uint16_t cval, int seg_rows, int nstrips,
                                                       int nsegs) {
    constexpr int PER = 2 * RY + 1;
    constexpr int U = PER * ((8 + PER - 1) / PER);
    constexpr int D = 8;
    constexpr int N = (2 * H0 + 1) + (RY >= 1 ? 2 * (2 * H1 + 1) : 0) + (RY >= 2 ? 2 * (2 * H2 + 1) : 0);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nstrips * nsegs) return;
    const int strip = wid % nstrips, seg = wid / nstrips;
    const size_t plane = (size_t)blockIdx.y * H * W;
    const mm_geo g = mm_geometry(strip, nstrips, lane, W);
    const int xg = g.xg;
    const int y_begin = __builtin_amdgcn_readfirstlane(seg * seg_rows);
    const int y_end = min(H, y_begin + seg_rows);
    const bool inside = g.inside;
    const unsigned cv2 = (unsigned)cval | ((unsigned)cval << 16);
    const unsigned lane_off = g.lane_off;
    const char* const in_plane = reinterpret_cast<const char*>(in + plane);
    const int refl = mode == AMT_MODE_REFLECT;
    auto load_row = [&](int r) -> uint4 {
        const int lo = refl ? -r - 1 : 0, hi = refl ? 2 * H - 1 - r : H - 1;
        int yy = r < 0 ? lo : (r >= H ? hi : r);
        yy = min(max(yy, 0), H - 1);
        return *reinterpret_cast<const uint4*>(in_plane + (size_t)yy * W * 2 + lane_off);
    };
    constexpr int DH = 4;
    auto load_halo = [&](int r) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g.need_h) {
            const int lo = refl ? -r - 1 : 0, hi = refl ? 2 * H - 1 - r : H - 1;
            int yy = r < 0 ? lo : (r >= H ? hi : r);
            yy = min(max(yy, 0), H - 1);
            v = *reinterpret_cast<const uint4*>(in_plane + (size_t)yy * W * 2 + g.halo_off);
        }
        return v;
    };
    uint4 inreg[U], halreg[U];
    unsigned ring[PER][6];
#pragma unroll
    for (int u = 0; u < D; ++u) inreg[u] = load_row(y_begin - RY + u);
#pragma unroll
    for (int u = 0; u < DH; ++u) halreg[u] = load_halo(y_begin - RY + u);
    const bool can_store = inside;
    uint16_t* const out_lane = out + plane + (inside ? xg : 0);
    for (int base = y_begin - RY; base < y_end + RY; base += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u;
            uint4 cur = inreg[u];
            inreg[(u + D) % U] = load_row(r + D);
            uint4 hal = halreg[u];
            halreg[(u + DH) % U] = load_halo(r + DH);
            if (mode == AMT_MODE_CONSTANT && (r < 0 || r >= H)) {
                cur = make_uint4(cv2, cv2, cv2, cv2);
                hal = cur;
            }
            uint4 lf, rt;
            mm_neighbours(g, lane, cur, hal, mode, cv2, lf, rt);
            const int slot = u % PER;
            ring[slot][0] = lf.w;
            ring[slot][1] = cur.x, ring[slot][2] = cur.y, ring[slot][3] = cur.z, ring[slot][4] = cur.w;
            ring[slot][5] = rt.x;
            const int yo = r - RY;
            if (yo >= y_begin && yo < y_end) {  // uniform: the warm-up rows of a segment skip the selection
                unsigned res[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned v[N];
                    int n = 0;
#pragma unroll
                    for (int dy = -RY; dy <= RY; ++dy) {
                        const int s = ((u - RY + dy) % PER + PER) % PER;
                        const int h = dy == 0 ? H0 : ((dy == 1 || dy == -1) ? H1 : H2);
#pragma unroll
                        for (int dx = -h; dx <= h; ++dx) {
                            unsigned val;
                            if (dx == 0) val = ring[s][1 + i];
                            else if (dx == -2) val = ring[s][i];
                            else if (dx == 2) val = ring[s][2 + i];
                            else if (dx == -1) val = __builtin_amdgcn_alignbit(ring[s][1 + i], ring[s][i], 16);
                            else val = __builtin_amdgcn_alignbit(ring[s][2 + i], ring[s][1 + i], 16);
                            v[n++] = val;
                        }
                    }
                    res[i] = med_step<0, N / 2 + 2, N / 2 + 2, N>(v);
                }
                if (can_store)
                    *reinterpret_cast<uint4*>(out_lane + (size_t)yo * W) = make_uint4(res[0], res[1], res[2], res[3]);
            }
        }
    }
}

// 1 = launched.  Qualifying footprints: see the kernel's comment.
static int medr_try(amt_ctx* ctx, const uint16_t* in, uint16_t* out, int nplanes, int H, int W, const uint8_t* footprint,
                    int fh, int fw, int mode, uint16_t cval) {
    if ((W & 7) || W < 16 || H < 16 || fh > 5 || fw > 5) return 0;
    if (!(mode == AMT_MODE_REFLECT || mode == AMT_MODE_NEAREST || mode == AMT_MODE_CONSTANT)) return 0;
    if (getenv("AMT_MM_TILE") != nullptr) return 0;
    const int ry = fh / 2, rx = fw / 2;
    int hw[3] = {-1, -1, -1};
    for (int y = 0; y < fh; ++y) {
        int x0 = -1, x1 = -1, cnt = 0;
        for (int x = 0; x < fw; ++x)
            if (footprint[y * fw + x]) {
                if (x0 < 0) x0 = x;
                x1 = x;
                ++cnt;
            }
        if (cnt == 0 || cnt != x1 - x0 + 1 || x0 + x1 != 2 * rx) return 0;  // one run, centred
        const int h = (x1 - x0) / 2, a = y < ry ? ry - y : y - ry;
        if (hw[a] >= 0 && hw[a] != h) return 0;  // top / bottom symmetric
        hw[a] = h;
    }
    const int key = ry * 1000 + hw[0] * 100 + (ry >= 1 ? hw[1] : 0) * 10 + (ry >= 2 ? hw[2] : 0);
    const int seg_rows = 64;
    const int nstrips = (W + 511) / 512, nsegs = (H + seg_rows - 1) / seg_rows;
    dim3 grid((nstrips * nsegs + 3) / 4, nplanes);
#define AMT_MEDR(RYV, A, B, C)                                                                                       \
    hipLaunchKernelGGL((medr_u16_kernel<RYV, A, B, C>), grid, dim3(256), 0, ctx->stream, in, out, H, W, mode, cval, \
                       seg_rows, nstrips, nsegs)
    switch (key) {
        case 1110: AMT_MEDR(1, 1, 1, 0); break;  // 3 x 3
        case 1100: AMT_MEDR(1, 1, 0, 0); break;  // cross = disk(1)
        case 2210: AMT_MEDR(2, 2, 1, 0); break;  // disk(2)
        case 2222: AMT_MEDR(2, 2, 2, 2); break;  // 5 x 5
        case 2221: AMT_MEDR(2, 2, 2, 1); break;  // 5 x 5 without its corners
        default: return 0;
    }
#undef AMT_MEDR
    return 1;
}

// host side: does the footprint qualify?  (rows are runs centred on the origin, ry <= 7, half-widths <= 7)
static bool mm_plan(const uint8_t* footprint, int fh, int fw, mm_params* P) {
    const int ry = fh / 2, rx = fw / 2;
    if (ry > 7 || rx > 7) return false;
    P->ry = ry;
    P->ns = 0;
    P->hmax = 0;
    for (int y = 0; y < fh; ++y) {
        int lo = -1, hi = -1, cnt = 0;
        for (int x = 0; x < fw; ++x)
            if (footprint[y * fw + x]) {
                if (lo < 0) lo = x;
                hi = x;
                ++cnt;
            }
        if (cnt == 0) {
            P->sel[y] = -1;
            continue;
        }
        if (hi - lo + 1 != cnt || (lo - rx) != -(hi - rx)) return false;  // holes, or not centred
        const int h = hi - rx;
        int s = 0;
        while (s < P->ns && P->hws[s] != h) ++s;
        if (s == P->ns) {
            if (P->ns == 8) return false;
            P->hws[P->ns++] = h;
        }
        P->sel[y] = h;  // half-width for now; replaced by the array index below
        P->hmax = h > P->hmax ? h : P->hmax;
    }
    for (int i = 1; i < P->ns; ++i)  // ascending half-widths
        for (int j = i; j > 0 && P->hws[j - 1] > P->hws[j]; --j) {
            const int t = P->hws[j];
            P->hws[j] = P->hws[j - 1];
            P->hws[j - 1] = t;
        }
    for (int y = 0; y < fh; ++y)
        if (P->sel[y] >= 0) {
            int s = 0;
            while (P->hws[s] != P->sel[y]) ++s;
            P->sel[y] = s;
        }
    const int off0 = P->hws[0] == 0 ? 1 : 0;
    for (int k = 0; k < 8; ++k) P->store_at[k] = -1;
    for (int i = 0; i < P->ns; ++i)
        if (P->hws[i] > 0) P->store_at[P->hws[i]] = i - off0;
    for (int y = 0; y < 15; ++y) P->row_arr[y] = P->row_h[y] = -1;
    for (int y = 0; y < fh; ++y)
        if (P->sel[y] >= 0) {
            P->row_arr[y] = P->hws[P->sel[y]] == 0 ? 0 : P->sel[y] - off0 + 1;
            P->row_h[y] = P->hws[P->sel[y]];
        }
    return P->ns > 0;
}

// 1 = the register kernel was launched and, if `minuend` was given, stored minuend - result; 0 = not applicable
static int rank_filter_impl(amt_ctx* ctx, const void* in, void* out, int dtype, int nplanes, int H, int W,
                            const uint8_t* footprint, int fh, int fw, int op, int mode, double cval,
                            const void* minuend, int* fused);

extern "C" int amt_rank_filter(amt_ctx* ctx, const void* in, void* out, int dtype, int nplanes, int H, int W,
                               const uint8_t* footprint, int fh, int fw, int op, int mode, double cval) {
    int fused = 0;
    return rank_filter_impl(ctx, in, out, dtype, nplanes, H, W, footprint, fh, fw, op, mode, cval, nullptr, &fused);
}

extern "C" int amt_rank_filter_sub(amt_ctx* ctx, const void* in, const void* minuend, void* out, int dtype, int nplanes,
                                   int H, int W, const uint8_t* footprint, int fh, int fw, int op, int mode,
                                   double cval) {
    AMT_REQUIRE(minuend && minuend != out, "rank_filter_sub: minuend must be given and differ from out");
    int fused = 0;
    AMT_TRY(rank_filter_impl(ctx, in, out, dtype, nplanes, H, W, footprint, fh, fw, op, mode, cval, minuend, &fused));
    if (fused) return AMT_OK;
    return amt_subtract(ctx, minuend, out, out, dtype, (size_t)nplanes * H * W);  // elementwise: in place is fine
}

static int rank_filter_impl(amt_ctx* ctx, const void* in, void* out, int dtype, int nplanes, int H, int W,
                            const uint8_t* footprint, int fh, int fw, int op, int mode, double cval,
                            const void* minuend, int* fused) {
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
    if (op <= 1 && dtype == AMT_U16 && (size_t)H * W >= 16) {  // uint16, small symmetric run footprint: packed kernel
        mm_params P;
        if (mm_plan(footprint, fh, fw, &P)) {
            // a footprint whose rows are centred runs is its own mirror image left-right; dilation also flips it
            // top-bottom (out[p] = max over s of in[p - s])
            if (op == 1)
                for (int a = 0, b = fh - 1; a < b; ++a, --b) {
                    const int t = P.row_arr[a];
                    P.row_arr[a] = P.row_arr[b];
                    P.row_arr[b] = t;
                    const int t2 = P.row_h[a];
                    P.row_h[a] = P.row_h[b];
                    P.row_h[b] = t2;
                }
            // register kernel (no LDS, rows loaded 2 ry + 1 steps ahead): aligned widths and boundary modes whose
            // out-of-image columns mirror the edge group (wrap / mirror keep the tile kernel)
            if ((W & 7) == 0 && W >= 16 && H >= 16 && (mode == AMT_MODE_REFLECT || mode == AMT_MODE_NEAREST || mode == AMT_MODE_CONSTANT) &&
                getenv("AMT_MM_TILE") == nullptr) {
                // disks get compile-time half-widths, rectangles one run-time half-width; anything else (crosses,
                // diamonds, gapped disks) the run-time table, up to 7 rows
                int shape = 0;
                {
                    bool disk = fh == fw, rect = true;
                    for (int y = 0; y < fh; ++y) {
                        const int h = P.row_h[y];
                        if (h != mm_shape_h(1, P.ry, y - P.ry)) disk = false;
                        if (h != P.hmax) rect = false;
                    }
                    shape = rect ? 2 : (disk ? 1 : 0);
                }
                const bool sub = minuend != nullptr && op == 1;
                const int seg_rows = P.ry <= 3 ? 64 : 128;
                const int nstrips = (W + 511) / 512, nsegs = (H + seg_rows - 1) / seg_rows;
                dim3 gridr((nstrips * nsegs + 3) / 4, nplanes);
                bool launched = true;
#define AMT_MMR_GO(RYV, SH)                                                                                            \
    do {                                                                                                               \
        if (op == 0)                                                                                                   \
            hipLaunchKernelGGL((mmr_u16_kernel<false, RYV, SH, false>), gridr, dim3(256), 0, ctx->stream,              \
                               (const uint16_t*)in, (uint16_t*)out, H, W, P, mode, (uint16_t)cval, seg_rows, nstrips,  \
                               nsegs, (const uint16_t*)nullptr);                                                       \
        else if (!sub)                                                                                                 \
            hipLaunchKernelGGL((mmr_u16_kernel<true, RYV, SH, false>), gridr, dim3(256), 0, ctx->stream,               \
                               (const uint16_t*)in, (uint16_t*)out, H, W, P, mode, (uint16_t)cval, seg_rows, nstrips,  \
                               nsegs, (const uint16_t*)nullptr);                                                       \
        else                                                                                                           \
            hipLaunchKernelGGL((mmr_u16_kernel<true, RYV, SH, true>), gridr, dim3(256), 0, ctx->stream,                \
                               (const uint16_t*)in, (uint16_t*)out, H, W, P, mode, (uint16_t)cval, seg_rows, nstrips,  \
                               nsegs, (const uint16_t*)minuend);                                                       \
    } while (0)
#define AMT_MMR_CASE(RYV)                                       \
    case RYV:                                                   \
        if (shape == 1) AMT_MMR_GO(RYV, 1);                     \
        else if (shape == 2) AMT_MMR_GO(RYV, 2);                \
        else if (RYV <= 3) AMT_MMR_GO((RYV <= 3 ? RYV : 0), 0); \
        else launched = false;                                  \
        break;
                switch (P.ry) {
                    AMT_MMR_CASE(0)
                    AMT_MMR_CASE(1)
                    AMT_MMR_CASE(2)
                    AMT_MMR_CASE(3)
                    AMT_MMR_CASE(4)
                    AMT_MMR_CASE(5)
                    AMT_MMR_CASE(6)
                    AMT_MMR_CASE(7)
                    default:
                        launched = false;
                }
#undef AMT_MMR_CASE
#undef AMT_MMR_GO
                if (launched) {
                    AMT_LAUNCH_CHECK();
                    *fused = sub ? 1 : 0;
                    return AMT_OK;
                }
            }
            const int narr = P.ns - (P.hws[0] == 0 ? 1 : 0);
            // 16-row tiles for small footprints (more blocks per CU), 32-row tiles where the halo would dominate
            const int th = P.ry <= 3 ? 16 : 32, tw = 128;
            const size_t smem3 = (size_t)(1 + narr) * (th + 2 * P.ry) * sizeof(uint16_t) * (tw + 16);
            if (smem3 <= 150 * 1024) {
                dim3 grid3((W + tw - 1) / tw, (H + th - 1) / th, nplanes);
#define AMT_MM_LAUNCH(MAXF, THV)                                                                                      \
    hipLaunchKernelGGL((mm_u16_kernel<MAXF, THV, 128>), grid3, dim3(256), smem3, ctx->stream, (const uint16_t*)in,    \
                       (uint16_t*)out, H, W, P, mode, (uint16_t)cval)
                if (op == 0 && th == 16) AMT_MM_LAUNCH(false, 16);
                else if (op == 0) AMT_MM_LAUNCH(false, 32);
                else if (th == 16) AMT_MM_LAUNCH(true, 16);
                else AMT_MM_LAUNCH(true, 32);
#undef AMT_MM_LAUNCH
                AMT_LAUNCH_CHECK();
                return AMT_OK;
            }
        }
    }
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
    if (op == 2 && dtype == AMT_U16 &&
        medr_try(ctx, (const uint16_t*)in, (uint16_t*)out, nplanes, H, W, footprint, fh, fw, mode, (uint16_t)cval)) {
        AMT_LAUNCH_CHECK();
        return AMT_OK;
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
