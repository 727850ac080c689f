// Cross-translation-unit internals of libamt_hip.so (not part of the C ABI).
#pragma once
#include "amt_common.h"

// ---- exclusive prefix sum over int32 arrays: one 1024-thread workgroup per plane ---------------
// data[plane][0..len) is replaced by its exclusive scan; total[plane] (optional) receives the sum.
template <int ITEMS>
__global__ void __launch_bounds__(1024) amt_scan_excl_kernel(int* __restrict__ data, int len, size_t plane_stride,
                                                             int* __restrict__ total,
                                                             const int* __restrict__ len_dev = nullptr) {
    __shared__ int s[1024];
    __shared__ int carry;
    int* d = data + (size_t)blockIdx.x * plane_stride;
    if (len_dev) len = len_dev[blockIdx.x];  // per-plane length that only the device knows
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int start = 0; start < len; start += 1024 * ITEMS) {
        const int b = start + threadIdx.x * ITEMS;
        int v[ITEMS];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            v[k] = (b + k < len) ? d[b + k] : 0;
            sum += v[k];
        }
        s[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const int c = carry;
        int run = c + s[threadIdx.x] - sum;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            if (b + k < len) d[b + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) total[blockIdx.x] = carry;
}

static inline int amt_scan_excl(amt_ctx* ctx, int* data, int len, size_t plane_stride, int* total, int nplanes) {
    if (len <= 4096)
        hipLaunchKernelGGL((amt_scan_excl_kernel<4>), dim3(nplanes), dim3(1024), 0, ctx->stream, data, len,
                           plane_stride, total, (const int*)nullptr);
    else
        hipLaunchKernelGGL((amt_scan_excl_kernel<16>), dim3(nplanes), dim3(1024), 0, ctx->stream, data, len,
                           plane_stride, total, (const int*)nullptr);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// same, with the per-plane length read from device memory (len_dev[plane] <= plane_stride)
static inline int amt_scan_excl_dev(amt_ctx* ctx, int* data, const int* len_dev, size_t plane_stride, int* total,
                                    int nplanes) {
    hipLaunchKernelGGL((amt_scan_excl_kernel<8>), dim3(nplanes), dim3(1024), 0, ctx->stream, data, 0, plane_stride,
                       total, len_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- float64 min / max per plane on ordered 64-bit keys (amt_stats.hip) ---------------------------
// keys = 2 * nplanes scratch words; init -> (producers fold with atomicMin / atomicMax on amt_f64_key) -> finish
int amt_i_minmax_init(amt_ctx* ctx, unsigned long long* keys, int nplanes);
int amt_i_minmax_finish(amt_ctx* ctx, const unsigned long long* keys, double* out, int nplanes);
int amt_i_minmax_f64(amt_ctx* ctx, const double* in, unsigned long long* keys, double* out, int nplanes, size_t n);
// Otsu threshold of float64 histograms (np.histogram edges from minmax); thr_code (nullable) = 2 * bin index
int amt_i_otsu_from_hist(amt_ctx* ctx, const uint32_t* hist, const double* minmax, int nbins, double* thr,
                         double* thr_code, int nplanes);

// ---- connected components (amt_label.hip) -------------------------------------------------------
// L[plane][p] = flat index of the component's first pixel (its union-find root), -1 for background.
// Components are sets of equal-valued non-zero pixels; conn8 selects 8- vs 4-connectivity.
// blk (nullable) = nplanes * amt_i_rank_blocks(n) ints receiving the per-block root counts for amt_i_rank_roots.
int amt_i_ccl_roots(amt_ctx* ctx, const void* in, int in_dtype, int* L, int* blk, int nplanes, int H, int W,
                    int conn8);
// 4-connected components of a uint8 mask without the per-pixel compression pass: pixels point at their tile-local
// root; tile-local roots are appended to one list per TILE ROW, rootlist[(plane * amt_i_tile_rows(H) + tile_row) *
// amt_i_rootlist_cap(W) ...], counted in nroots[plane * tile_rows + tile_row] (zero on entry).  Callers compress the
// listed roots themselves and resolve a pixel as L[L[p]].
int amt_i_tile_rows(int H);
size_t amt_i_rootlist_cap(int W);
// multi (nullable): one int of scratch; with it a 0 / 1 mask takes the bit-parallel tile kernel (other byte values
// are detected and redone by the pixel kernel)
size_t amt_i_ccl_scratch_ints(int nplanes, int H, int W);  // ints of scratch behind `multi`
int amt_i_ccl_tileroots_u8(amt_ctx* ctx, const uint8_t* in, int* L, int* rootlist, int* nroots, int nplanes, int H,
                           int W, int* multi = nullptr);
// Run tables of a 0 / 1 (or truth-value) mask, per 64 x 64 tile t = (plane * tile_rows + tile_row) * segments + segment:
//   tbits[t * 64 + row]   the row's 64 pixels as a word (bit i = column i)
//   rtab[t * RT_CAP + k]  for the tile's k-th run in raster order: (row << 6 | column) of the first pixel of its TILE ROOT
//   nruns[t]              number of runs
// A pixel's run: runs of the rows above (prefix sum of the rows' head counts) + heads at or before it in its own row.
//   roff[t * 64 + row]    runs of the tile's rows above `row` (optional: random look-ups need it, tile-wide passes scan)
constexpr int RT_CAP = 2048;  // 64 rows x at most 32 runs
bool amt_i_ccl_runs_ok(const void* in, int H, int W, int nplanes);
// The watershed's labelling of its mask from run tables alone: 4-connected components of the NON-ZERO bytes; L is written
// at the tile roots only (L[root] = root, then the seams' unions), the tile roots are listed as amt_i_ccl_tileroots_u8
// lists them
int amt_i_ccl_tileroots_runs_u8(amt_ctx* ctx, const uint8_t* in, int* L, int* rootlist, int* nroots, int nplanes, int H,
                                int W, unsigned long long* tbits, unsigned short* rtab, int* nruns, unsigned short* roff);
// what a kernel needs for random look-ups "pixel -> run -> component" (rcomp[t * RT_CAP + k] = the 1-based component id of
// run k, written by the watershed's statistics pass); tbits == nullptr: no run tables, the caller reads its parent plane
struct amt_runtabs {
    const unsigned long long* tbits;
    const unsigned short* roff;
    const unsigned short* rtab;
    const int* rcomp;
    int segs, trows;
};
// A[t] = A[L[t]] for every listed tile root t (lists compressed): a pixel then reaches its component's entry of A
// with one hop through its tile root
int amt_i_propagate_roots(amt_ctx* ctx, int* A, const int* L, const int* rootlist, const int* nroots, int nplanes, int H,
                          int W);
// T[plane][root] = 1-based rank of the root in raster order; count_dev[plane] = number of roots.
// blk = scratch of nplanes * amt_i_rank_blocks(n) ints.
int amt_i_rank_blocks(size_t n);
int amt_i_rank_roots(amt_ctx* ctx, const int* L, int* T, int* blk, int* count_dev, int nplanes, size_t n);

// label map halves (amt_label.hip): see amt_i_presence_fill
int amt_i_presence_fill(amt_ctx* ctx, int* P, const int* nlabels_dev, int max_label, int nplanes);
int amt_i_drop_and_scan(amt_ctx* ctx, int* P, int max_label, int* count_dev, int nplanes);

// Value of the neighbouring lane by DPP wave shift (a VALU move, no LDS crossbar as ds_bpermute needs): lane 0 of
// amt_lane_left / lane 63 of amt_lane_right receive 0, every caller masks those lanes itself.  All 64 lanes must be
// active (call from wave-uniform control flow only).
#ifdef __HIPCC__
__device__ __forceinline__ int ccl_wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
// flat index (inside its plane) of the tile root of run k of tile (ty, bx)
__device__ __forceinline__ int ccl_rt_root(const unsigned short* __restrict__ rtab, size_t tile, int k, int ty, int bx,
                                           int W) {
    const int e = rtab[tile * RT_CAP + k];
    return (ty * 64 + (e >> 6)) * W + bx * 64 + (e & 63);
}
// run of pixel (y, x) of `plane`: its index into rtab / rcomp (tile * RT_CAP + ordinal), or -1 for a background pixel
__device__ __forceinline__ long long amt_rt_px_run(const amt_runtabs& rt, int plane, int y, int x) {
    const size_t t = ((size_t)plane * rt.trows + (y >> 6)) * rt.segs + (x >> 6);
    const int row = y & 63, col = x & 63;
    const unsigned long long w = rt.tbits[t * 64 + row];
    const int ro = rt.roff[t * 64 + row];
    if (!((w >> col) & 1ull)) return -1;
    return (long long)(t * RT_CAP) + ro + __popcll((w & ~(w << 1)) & ((2ull << col) - 1ull)) - 1;
}
// flat index (inside its plane) of the tile root of the run with table index ri (as amt_rt_px_run returns it)
__device__ __forceinline__ int amt_rt_run_root(const amt_runtabs& rt, long long ri, int W) {
    const size_t t = (size_t)ri / RT_CAP;
    const int e = rt.rtab[ri];
    const int bx = (int)(t % rt.segs), ty = (int)((t / rt.segs) % rt.trows);
    return (ty * 64 + (e >> 6)) * W + bx * 64 + (e & 63);
}
__device__ __forceinline__ int amt_lane_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int amt_lane_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ long long amt_lane_left(long long v) {
    const unsigned lo = (unsigned)amt_lane_left((int)(unsigned)v), hi = (unsigned)amt_lane_left((int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long amt_lane_right(long long v) {
    const unsigned lo = (unsigned)amt_lane_right((int)(unsigned)v), hi = (unsigned)amt_lane_right((int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#endif
