// Exact Euclidean distance transform and the peak mask of the config-3 marker recipe.
//
// scipy.ndimage.distance_transform_edt(mask) == sqrt(float64(d2)) bitwise, with d2 the exact integer
// squared distance to the nearest zero pixel.  d2 is computed separably:
//   pass 1 (rows)   : g(y,x) = distance to the nearest zero pixel in row y (G_INF if none), found by an
//                     outward search in an LDS copy of the row;
//   pass 2 (columns): d2(y,x) = min_k (k^2 + g(y+-k,x)^2), scanning k outward while k^2 < best; a wave
//                     reads 64 consecutive x of row y+-k, so every access is coalesced.
// Both searches are exact and cost O(distance) per pixel, which is what nuclei-sized objects need; they
// degrade (never fail) on very large solid regions.
#include "amt_internal.h"

constexpr int G_INF = 0x3fffffff;

__global__ void __launch_bounds__(256) edt_rows_kernel(const uint8_t* __restrict__ mask, int* __restrict__ g, int H,
                                                       int W) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint8_t* row = reinterpret_cast<uint8_t*>(smem_raw);
    const size_t base = ((size_t)blockIdx.y * H + blockIdx.x) * W;
    for (int x = threadIdx.x; x < W; x += 256) row[x] = mask[base + x];
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        int d = 0;
        if (row[x]) {
            d = G_INF;
            for (int k = 1; x - k >= 0 || x + k < W; ++k) {
                if ((x - k >= 0 && row[x - k] == 0) || (x + k < W && row[x + k] == 0)) {
                    d = k;
                    break;
                }
            }
        }
        g[base + x] = d;
    }
}

__global__ void __launch_bounds__(256) edt_cols_kernel(const int* __restrict__ g, int* __restrict__ d2_out,
                                                       double* __restrict__ edt_out, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int* gp = g + plane;
    const size_t i = (size_t)y * W + x;
    long long best;
    const int g0 = gp[i];
    if (g0 == 0) {
        best = 0;
    } else {
        best = g0 == G_INF ? (long long)0x7fffffffffffll : (long long)g0 * g0;
        for (long long k = 1; k * k < best; ++k) {
            const int yu = y - (int)k, yd = y + (int)k;
            if (yu < 0 && yd >= H) break;
            if (yu >= 0) {
                int gv = gp[(size_t)yu * W + x];
                if (gv != G_INF) {
                    long long c = k * k + (long long)gv * gv;
                    best = c < best ? c : best;
                }
            }
            if (yd < H) {
                int gv = gp[(size_t)yd * W + x];
                if (gv != G_INF) {
                    long long c = k * k + (long long)gv * gv;
                    best = c < best ? c : best;
                }
            }
        }
        if (best > 0x7fffffffll) best = 0x7fffffffll;  // no zero pixel anywhere: saturate
    }
    if (d2_out) d2_out[plane + i] = (int)best;
    if (edt_out) edt_out[plane + i] = sqrt((double)best);
}

extern "C" int amt_edt(amt_ctx* ctx, const uint8_t* mask, int32_t* d2_out, double* edt_out, int nplanes, int H,
                       int W) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(mask && (d2_out || edt_out) && nplanes >= 0 && H > 0 && W > 0, "edt: bad arguments");
    AMT_REQUIRE(H <= 32768 && W <= 32768, "edt: image larger than 32768 pixels per side");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 4)));
    int* g = arena_take_t<int>(ctx, (size_t)nplanes * n);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(H, nplanes), dim3(256), (size_t)amt_align(W, 16), ctx->stream, mask, g, H,
                       W);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(edt_cols_kernel, dim3((W + 63) / 64, (H + 3) / 4, nplanes), dim3(256), 0, ctx->stream, g, d2_out,
                       edt_out, H, W);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- peak mask ---------------------------------------------------------------------------------
// peaks = (d2 == maximum_filter(d2, size=2m+1, mode='constant' (0))) & mask & (d2 > 0), border m cleared
// (comparing the integer d2 is equivalent to comparing sqrt(d2)).
// One 256-thread block per 32 x 64 tile: d2 tile + halo m staged in LDS, separable maximum (rows, then
// columns) in LDS, compare.  Tiles without any masked pixel with d2 > 0 exit after the staging pass.
constexpr int PK_H = 32, PK_W = 64, PK_MAXM = 16;
__global__ void __launch_bounds__(256) peaks_tile_kernel(const int* __restrict__ d2, const uint8_t* __restrict__ mask,
                                                         uint8_t* __restrict__ peaks, int H, int W, int m) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int TW = PK_W + 2 * m, TH = PK_H + 2 * m;
    int* tile = reinterpret_cast<int*>(smem_raw);  // TH x TW
    int* rmax = tile + TH * TW;                    // TH x PK_W
    __shared__ int any_fg;
    const int x0 = blockIdx.x * PK_W, y0 = blockIdx.y * PK_H;
    const size_t base = (size_t)blockIdx.z * H * W;
    if (threadIdx.x == 0) any_fg = 0;
    __syncthreads();
    int local_any = 0;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ky = i / TW, kx = i - ky * TW;
        const int y = y0 - m + ky, x = x0 - m + kx;
        int v = 0;  // constant 0 outside the image
        if (y >= 0 && y < H && x >= 0 && x < W) v = d2[base + (size_t)y * W + x];
        tile[i] = v;
        if (v > 0 && ky >= m && ky < TH - m && kx >= m && kx < TW - m) local_any = 1;
    }
    if (local_any) any_fg = 1;
    __syncthreads();
    if (!any_fg) {  // nothing can be a peak here
        for (int i = threadIdx.x; i < PK_H * PK_W; i += 256) {
            const int ky = i / PK_W, kx = i - ky * PK_W;
            const int y = y0 + ky, x = x0 + kx;
            if (y < H && x < W) peaks[base + (size_t)y * W + x] = 0;
        }
        return;
    }
    for (int i = threadIdx.x; i < TH * PK_W; i += 256) {
        const int ky = i / PK_W, kx = i - ky * PK_W;
        const int* c = tile + ky * TW + kx;  // window kx .. kx + 2m of the padded row
        int best = 0;
        for (int k = 0; k <= 2 * m; ++k) best = c[k] > best ? c[k] : best;
        rmax[i] = best;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PK_H * PK_W; i += 256) {
        const int ky = i / PK_W, kx = i - ky * PK_W;
        const int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        const int v = tile[(ky + m) * TW + (kx + m)];
        uint8_t r = 0;
        if (v > 0 && y >= m && y < H - m && x >= m && x < W - m && mask[base + (size_t)y * W + x]) {
            int best = 0;
            for (int k = 0; k <= 2 * m; ++k) {
                const int t = rmax[(ky + k) * PK_W + kx];
                best = t > best ? t : best;
            }
            r = (v == best) ? 1 : 0;
        }
        peaks[base + (size_t)y * W + x] = r;
    }
}

extern "C" int amt_peak_mask(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H,
                             int W, int min_distance) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(d2 && mask && peaks && nplanes >= 0 && H > 0 && W > 0, "peak_mask: bad arguments");
    AMT_REQUIRE(min_distance >= 0 && min_distance <= PK_MAXM, "peak_mask: min_distance %d out of range 0..%d",
                min_distance, PK_MAXM);
    if (nplanes == 0) return AMT_OK;
    const int m = min_distance;
    const size_t smem = ((size_t)(PK_H + 2 * m) * (PK_W + 2 * m) + (size_t)(PK_H + 2 * m) * PK_W) * sizeof(int);
    dim3 grid((W + PK_W - 1) / PK_W, (H + PK_H - 1) / PK_H, nplanes);
    hipLaunchKernelGGL(peaks_tile_kernel, grid, dim3(256), smem, ctx->stream, d2, mask, peaks, H, W, m);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
