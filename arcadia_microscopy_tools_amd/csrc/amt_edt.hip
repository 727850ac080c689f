// Exact Euclidean distance transform and the peak mask of the config-3 marker recipe.
//
// scipy.ndimage.distance_transform_edt(mask) == sqrt(float64(d2)) bitwise, with d2 the exact integer
// squared distance to the nearest zero pixel.  d2 is computed separably:
//   pass 1 (rows)   : g(y,x) = distance to the nearest zero pixel in row y (G_INF if none).  The row is
//                     turned into 64-pixel bit words of "zero" flags (one ballot each); the nearest zero on
//                     either side of a pixel is a clz / ffs on its own word, walking to further words only
//                     across solid 64-pixel stretches.  g is stored as uint16 (sides are <= 32768).
//   pass 2 (columns): d2(y,x) = min_k (k^2 + g(y+-k,x)^2), scanning k outward while k^2 < best.  A block
//                     stages 64 columns x (32 + 2*24) rows of g in LDS; only searches deeper than the halo
//                     continue in HBM (coalesced: a wave reads 64 consecutive x of row y+-k).
// Both searches are exact and cost O(distance) per pixel, which is what nuclei-sized objects need; they
// degrade (never fail) on very large solid regions.
#include "amt_internal.h"

constexpr unsigned G_INF = 0xFFFFu;  // no zero pixel in this row
constexpr int EC_ROWS = 32, EC_HALO = 24, EC_TROWS = EC_ROWS + 2 * EC_HALO;

__global__ void __launch_bounds__(256) edt_rows_kernel(const uint8_t* __restrict__ mask, unsigned short* __restrict__ g,
                                                       int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* zw = reinterpret_cast<unsigned long long*>(smem_raw);  // zero flags, 64 pixels per word
    const size_t base = ((size_t)blockIdx.y * H + blockIdx.x) * W;
    const int nw = (W + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (int x0 = (threadIdx.x >> 6) * 64; x0 < W; x0 += 256) {
        const int x = x0 + lane;
        const unsigned long long z = __ballot(x < W && mask[base + x] == 0);  // beyond W: "not zero"
        if (lane == 0) zw[x0 >> 6] = z;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const int wi = x >> 6, bit = x & 63;
        const unsigned long long own = zw[wi];
        unsigned d = 0;
        if (!((own >> bit) & 1ull)) {
            // nearest zero to the left
            unsigned dl = G_INF, dr = G_INF;
            unsigned long long m = own & ((1ull << bit) - 1ull);
            int w = wi;
            while (m == 0 && w > 0) m = zw[--w];
            if (m) dl = (unsigned)(x - (w * 64 + 63 - __clzll((long long)m)));
            // nearest zero to the right
            m = bit == 63 ? 0ull : (own >> (bit + 1)) << (bit + 1);
            w = wi;
            while (m == 0 && w + 1 < nw) m = zw[++w];
            if (m) dr = (unsigned)(w * 64 + __ffsll((long long)m) - 1 - x);
            d = dl < dr ? dl : dr;
        }
        g[base + x] = (unsigned short)d;
    }
}

__global__ void __launch_bounds__(256) edt_cols_kernel(const unsigned short* __restrict__ g, int* __restrict__ d2_out,
                                                       double* __restrict__ edt_out, int H, int W) {
    __shared__ unsigned short tile[EC_TROWS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * EC_ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const unsigned short* gp = g + plane;
    const int xc = x < W ? x : W - 1;
    // all loads of a wave are issued before the first LDS store
    unsigned short gv[EC_TROWS / 4];
#pragma unroll
    for (int j = 0; j < EC_TROWS / 4; ++j) {
        const int y = y0 - EC_HALO + wave + 4 * j;
        gv[j] = (y >= 0 && y < H && x < W) ? gp[(size_t)y * W + xc] : (unsigned short)G_INF;
    }
#pragma unroll
    for (int j = 0; j < EC_TROWS / 4; ++j) tile[wave + 4 * j][lane] = gv[j];
    __syncthreads();
    if (x >= W) return;
#pragma unroll 1
    for (int j = 0; j < EC_ROWS / 4; ++j) {
        const int ly = EC_HALO + wave * (EC_ROWS / 4) + j;
        const int y = y0 - EC_HALO + ly;
        if (y >= H) break;
        const unsigned g0 = tile[ly][lane];
        unsigned best = 0;
        if (g0 != 0) {
            best = g0 == G_INF ? 0xFFFFFFFFu : g0 * g0;
            unsigned k = 1;
            for (; k <= (unsigned)EC_HALO && k * k < best; ++k) {
                const unsigned gu = tile[ly - (int)k][lane], gd = tile[ly + (int)k][lane];
                const unsigned gm = gu < gd ? gu : gd;
                if (gm != G_INF) {
                    const unsigned c = k * k + gm * gm;
                    best = c < best ? c : best;
                }
            }
            for (; k < 65536u && (unsigned long long)k * k < best; ++k) {  // beyond the LDS halo
                const int yu = y - (int)k, yd = y + (int)k;
                if (yu < 0 && yd >= H) break;
                unsigned gm = G_INF;
                if (yu >= 0) gm = gp[(size_t)yu * W + x];
                if (yd < H) {
                    const unsigned gd = gp[(size_t)yd * W + x];
                    gm = gd < gm ? gd : gm;
                }
                if (gm != G_INF) {
                    const unsigned c = k * k + gm * gm;  // < 2^31: both terms < 2^30
                    best = c < best ? c : best;
                }
            }
            if (best > 0x7fffffffu) best = 0x7fffffffu;  // no zero pixel anywhere: saturate
        }
        const size_t i = plane + (size_t)y * W + x;
        if (d2_out) d2_out[i] = (int)best;
        if (edt_out) edt_out[i] = sqrt((double)best);
    }
}

extern "C" int amt_edt(amt_ctx* ctx, const uint8_t* mask, int32_t* d2_out, double* edt_out, int nplanes, int H,
                       int W) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(mask && (d2_out || edt_out) && nplanes >= 0 && H > 0 && W > 0, "edt: bad arguments");
    AMT_REQUIRE(H <= 32768 && W <= 32768, "edt: image larger than 32768 pixels per side");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 2)));
    unsigned short* g = arena_take_t<unsigned short>(ctx, (size_t)nplanes * n);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(H, nplanes), dim3(256), (size_t)((W + 63) / 64) * 8, ctx->stream, mask, g,
                       H, W);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(edt_cols_kernel, dim3((W + 63) / 64, (H + EC_ROWS - 1) / EC_ROWS, nplanes), dim3(256), 0,
                       ctx->stream, g, d2_out, edt_out, H, W);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- peak mask ---------------------------------------------------------------------------------
// peaks = (d2 == maximum_filter(d2, size=2m+1, mode='constant' (0))) & mask & (d2 > 0), border m cleared
// (comparing the integer d2 is equivalent to comparing sqrt(d2)).
// "d2 equals the window maximum" == "nothing in the window exceeds d2", so no maximum image is built:
// one 256-thread block stages a 32 x 64 tile of d2 plus a halo of m in LDS; a pixel is first compared with
// its 8 neighbours (which rejects all but a few percent of the foreground), and only the survivors scan the
// full (2m+1)^2 window with early exit.  Tiles without any d2 > 0 exit after the staging pass.
constexpr int PK_H = 64, PK_W = 64, PK_MAXM = 16;
__global__ void __launch_bounds__(256) peaks_tile_kernel(const int* __restrict__ d2, const uint8_t* __restrict__ mask,
                                                         uint8_t* __restrict__ peaks, int H, int W, int m) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int TW = PK_W + 2 * m, TH = PK_H + 2 * m;
    int* tile = reinterpret_cast<int*>(smem_raw);  // TH x TW
    const int x0 = blockIdx.x * PK_W, y0 = blockIdx.y * PK_H;
    const size_t base = (size_t)blockIdx.z * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int local_any = 0;
    // staging: wave w owns tile rows w, w + 4, ...: the 64 centre columns (aligned, coalesced) and the 2m halo
    // columns; ALL loads of a wave are issued before the first LDS store (TH <= 64, i.e. <= 16 rows per wave)
    constexpr int RPW = (PK_H + 2 * PK_MAXM + 3) / 4;
    int vc[RPW], vh[RPW];
    const int kxh = lane < m ? lane : PK_W + lane;  // halo column of lanes < 2m
    const int xh = x0 - m + kxh;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int ky = wave + 4 * j;
        const int y = y0 - m + ky;
        const bool yok = ky < TH && y >= 0 && y < H;
        const int x = x0 + lane;
        vc[j] = (yok && x < W) ? d2[base + (size_t)y * W + x] : 0;  // constant 0 outside the image
        vh[j] = (yok && lane < 2 * m && xh >= 0 && xh < W) ? d2[base + (size_t)y * W + xh] : 0;
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int ky = wave + 4 * j;
        if (ky < TH) {
            tile[ky * TW + m + lane] = vc[j];
            if (lane < 2 * m) tile[ky * TW + kxh] = vh[j];
            if (vc[j] > 0 && ky >= m && ky < TH - m) local_any = 1;
        }
    }
    const int any_fg = __syncthreads_or(local_any);
    // each thread owns 4 consecutive pixels of a row: 16 threads per row, 16 rows per pass
    const int kx4 = (threadIdx.x & 15) * 4;
    for (int ky = threadIdx.x >> 4; ky < PK_H; ky += 16) {
        const int y = y0 + ky;
        if (y >= H) break;
        unsigned packed = 0;
        if (any_fg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kx = kx4 + i, x = x0 + kx;
                const int* c = tile + (ky + m) * TW + (kx + m);
                const int v = c[0];
                if (v > 0 && y >= m && y < H - m && x >= m && x < W - m) {
                    bool gt = false;
                    if (m >= 1)
                        gt = c[-1] > v || c[1] > v || c[-TW] > v || c[TW] > v || c[-TW - 1] > v || c[-TW + 1] > v ||
                             c[TW - 1] > v || c[TW + 1] > v;
                    if (!gt && mask[base + (size_t)y * W + x]) {
                        for (int dy = -m; dy <= m && !gt; ++dy) {
                            const int* row = c + dy * TW;
                            for (int dx = -m; dx <= m; ++dx) gt |= row[dx] > v;
                        }
                        if (!gt) packed |= 1u << (8 * i);
                    }
                }
            }
        }
        const int x = x0 + kx4;
        uint8_t* o = peaks + base + (size_t)y * W + x;
        if (x + 3 < W && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
            *reinterpret_cast<unsigned*>(o) = packed;
        } else {
            for (int i = 0; i < 4 && x + i < W; ++i) o[i] = (packed >> (8 * i)) & 1u;
        }
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
    const size_t smem = (size_t)(PK_H + 2 * m) * (PK_W + 2 * m) * sizeof(int);
    dim3 grid((W + PK_W - 1) / PK_W, (H + PK_H - 1) / PK_H, nplanes);
    hipLaunchKernelGGL(peaks_tile_kernel, grid, dim3(256), smem, ctx->stream, d2, mask, peaks, H, W, m);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
