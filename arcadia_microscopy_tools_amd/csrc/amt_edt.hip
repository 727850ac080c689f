// Exact Euclidean distance transform and the peak mask of the config-3 marker recipe.
//
// scipy.ndimage.distance_transform_edt(mask) == sqrt(float64(d2)) bitwise, with d2 the exact integer
// squared distance to the nearest zero pixel.  d2 is computed separably:
//   pass 1 (columns): g(y,x) = distance to the nearest zero pixel in column x (INF if none);
//   pass 2 (rows)   : d2(y,x) = min_k (k^2 + g(y,x+-k)^2), scanning k outward while k^2 < best.
// The outward scan is exact and costs O(distance) per pixel, which is what nuclei-sized objects need;
// it degrades (never fails) on very large solid regions.
#include "amt_internal.h"

constexpr int G_INF = 0x3fffffff;

// one thread per column, two sweeps; coalesced across the wave (consecutive x).
__global__ void __launch_bounds__(256) edt_cols_kernel(const uint8_t* __restrict__ mask, int* __restrict__ g, int H,
                                                       int W) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    int d = G_INF;  // distance to the last zero seen above
    for (int y = 0; y < H; ++y) {
        size_t i = base + (size_t)y * W + x;
        if (mask[i] == 0)
            d = 0;
        else if (d != G_INF)
            d += 1;
        g[i] = d;
    }
    d = G_INF;
    for (int y = H - 1; y >= 0; --y) {
        size_t i = base + (size_t)y * W + x;
        int up = g[i];
        if (up == 0)
            d = 0;
        else if (d != G_INF)
            d += 1;
        g[i] = d < up ? d : up;
    }
}

__global__ void __launch_bounds__(256) edt_rows_kernel(const int* __restrict__ g, int* __restrict__ d2_out,
                                                       double* __restrict__ edt_out, int H, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const size_t row = (size_t)blockIdx.z * H * W + (size_t)y * W;
    const int* gr = g + row;
    long long best;
    int g0 = gr[x];
    if (g0 == 0) {
        best = 0;
    } else {
        best = g0 == G_INF ? (long long)0x7fffffffffffll : (long long)g0 * g0;
        for (long long k = 1; k * k < best; ++k) {
            int xl = x - (int)k, xr = x + (int)k;
            if (xl < 0 && xr >= W) break;
            if (xl >= 0) {
                int gv = gr[xl];
                if (gv != G_INF) {
                    long long c = k * k + (long long)gv * gv;
                    best = c < best ? c : best;
                }
            }
            if (xr < W) {
                int gv = gr[xr];
                if (gv != G_INF) {
                    long long c = k * k + (long long)gv * gv;
                    best = c < best ? c : best;
                }
            }
        }
        if (best > 0x7fffffffll) best = 0x7fffffffll;  // no zero pixel anywhere near: saturate
    }
    if (d2_out) d2_out[row + x] = (int)best;
    if (edt_out) edt_out[row + x] = sqrt((double)best);
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
    hipLaunchKernelGGL(edt_cols_kernel, dim3((W + 255) / 256, nplanes), dim3(256), 0, ctx->stream, mask, g, H, W);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(edt_rows_kernel, dim3((W + 255) / 256, H, nplanes), dim3(256), 0, ctx->stream, g, d2_out,
                       edt_out, H, W);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- peak mask ---------------------------------------------------------------------------------
// peaks = (d2 == maximum_filter(d2, size=2m+1, mode='constant' (0))) & mask & (d2 > 0), border m cleared
// (comparing the integer d2 is equivalent to comparing sqrt(d2)).
__global__ void __launch_bounds__(256) rowmax_kernel(const int* __restrict__ d2, int* __restrict__ out, int H, int W,
                                                     int m) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const size_t row = (size_t)blockIdx.z * H * W + (size_t)y * W;
    int best = 0;  // constant 0 outside
    int x0 = x - m < 0 ? 0 : x - m, x1 = x + m >= W ? W - 1 : x + m;
    for (int k = x0; k <= x1; ++k) {
        int v = d2[row + k];
        best = v > best ? v : best;
    }
    out[row + x] = best;
}

__global__ void __launch_bounds__(256) colmax_peaks_kernel(const int* __restrict__ d2, const int* __restrict__ rmax,
                                                           const uint8_t* __restrict__ mask,
                                                           uint8_t* __restrict__ peaks, int H, int W, int m) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const size_t base = (size_t)blockIdx.z * H * W;
    const size_t i = base + (size_t)y * W + x;
    uint8_t r = 0;
    const int v = d2[i];
    if (v > 0 && mask[i] && y >= m && y < H - m && x >= m && x < W - m) {
        int best = 0;
        int y0 = y - m < 0 ? 0 : y - m, y1 = y + m >= H ? H - 1 : y + m;
        for (int k = y0; k <= y1; ++k) {
            int t = rmax[base + (size_t)k * W + x];
            best = t > best ? t : best;
        }
        r = (v == best) ? 1 : 0;
    }
    peaks[i] = r;
}

extern "C" int amt_peak_mask(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H,
                             int W, int min_distance) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(d2 && mask && peaks && nplanes >= 0 && H > 0 && W > 0, "peak_mask: bad arguments");
    AMT_REQUIRE(min_distance >= 0 && min_distance <= 64, "peak_mask: min_distance %d out of range 0..64", min_distance);
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 4)));
    int* rmax = arena_take_t<int>(ctx, (size_t)nplanes * n);
    dim3 grid((W + 255) / 256, H, nplanes);
    hipLaunchKernelGGL(rowmax_kernel, grid, dim3(256), 0, ctx->stream, d2, rmax, H, W, min_distance);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(colmax_peaks_kernel, grid, dim3(256), 0, ctx->stream, d2, rmax, mask, peaks, H, W, min_distance);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
