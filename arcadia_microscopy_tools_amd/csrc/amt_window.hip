// Niblack / Sauvola local thresholds: window mean and standard deviation
// (SK/filters/thresholding.py:910-964 `_mean_std`, :1026-1027 niblack, :1083-1087 sauvola;
// R/operations.py:194-195).
//
// scikit-image pads the image by numpy's 'reflect' (whole-sample symmetric, scipy's 'mirror'), builds two
// float64 integral images and takes 2^d corner differences; the window of output pixel (r, c) is the
// w x w square centred on it.  Here the two window sums (x and x^2) are formed directly: a row pass
// builds w-wide horizontal sums from an LDS prefix sum of the mirrored row, a column pass slides a
// w-tall running sum down each column.  For uint16 images both sums are exact integers (scikit-image's
// integral images round once the running sum of squares exceeds 2^53); for float64 images each window is
// summed afresh in a fixed order.  The threshold image is written as float64.
#include "amt_common.h"

typedef unsigned long long u64;

// ---- uint16: exact integer sums ------------------------------------------------------------------
// rows: hs[y][x] = sum_{dx=-h..h} in[y][mirror(x+dx)], hq likewise for squares
__global__ void __launch_bounds__(256) win_rows_u16_kernel(const uint16_t* __restrict__ in, unsigned* __restrict__ hs,
                                                           u64* __restrict__ hq, int H, int W, int h) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int PW = W + 2 * h;  // padded row
    u64* ps = reinterpret_cast<u64*>(smem_raw);  // prefix of x   (PW + 1)
    u64* pq = ps + (PW + 1);                     // prefix of x^2 (PW + 1)
    const size_t base = ((size_t)blockIdx.y * H + blockIdx.x) * W;
    // chunked prefix: each thread owns a contiguous chunk, then a block scan of the chunk totals
    const int per = (PW + 255) / 256;
    const int b = threadIdx.x * per;
    u64 s = 0, q = 0;
    for (int k = 0; k < per; ++k) {
        int i = b + k;
        if (i < PW) {
            int x = amt_map_index(i - h, W, AMT_MODE_MIRROR);
            u64 v = in[base + x];
            s += v;
            q += v * v;
        }
    }
    __shared__ u64 ts[256], tq[256];
    ts[threadIdx.x] = s;
    tq[threadIdx.x] = q;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        u64 a = threadIdx.x >= off ? ts[threadIdx.x - off] : 0, c = threadIdx.x >= off ? tq[threadIdx.x - off] : 0;
        __syncthreads();
        ts[threadIdx.x] += a;
        tq[threadIdx.x] += c;
        __syncthreads();
    }
    u64 rs = ts[threadIdx.x] - s, rq = tq[threadIdx.x] - q;  // exclusive prefix of this chunk
    if (threadIdx.x == 0) {
        ps[0] = 0;
        pq[0] = 0;
    }
    for (int k = 0; k < per; ++k) {
        int i = b + k;
        if (i < PW) {
            int x = amt_map_index(i - h, W, AMT_MODE_MIRROR);
            u64 v = in[base + x];
            rs += v;
            rq += v * v;
            ps[i + 1] = rs;
            pq[i + 1] = rq;
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        // window = padded indices x .. x + 2h  ->  prefix[x + 2h + 1] - prefix[x]
        hs[base + x] = (unsigned)(ps[x + 2 * h + 1] - ps[x]);
        hq[base + x] = pq[x + 2 * h + 1] - pq[x];
    }
}

__device__ __forceinline__ double local_thr(double m, double s, int method, double k, double r) {
    // niblack: m - k*s ; sauvola: m * (1 + k * ((s / r) - 1))
    if (method == 0) return m - k * s;
    return m * (1.0 + k * ((s / r) - 1.0));
}

// cols: one thread per column slides a (2h+1)-tall running sum down the rows of hs / hq
__global__ void __launch_bounds__(256) win_cols_u16_kernel(const unsigned* __restrict__ hs, const u64* __restrict__ hq,
                                                           double* __restrict__ thr, int H, int W, int h, int method,
                                                           double k, double r, int TH, int hx) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y0 = blockIdx.y * TH;
    const size_t base = (size_t)blockIdx.z * H * W;
    const double wsz = (double)(2 * h + 1) * (double)(2 * hx + 1);  // rows of the window x its columns
    u64 s = 0, q = 0;
    for (int dy = -h; dy <= h; ++dy) {
        int yy = amt_map_index(y0 + dy, H, AMT_MODE_MIRROR);
        s += hs[base + (size_t)yy * W + x];
        q += hq[base + (size_t)yy * W + x];
    }
    const int y1 = (y0 + TH < H) ? y0 + TH : H;
    for (int y = y0; y < y1; ++y) {
        const double m = (double)s / wsz;
        const double g2 = (double)q / wsz;
        double var = g2 - m * m;
        var = var < 0.0 ? 0.0 : var;
        thr[base + (size_t)y * W + x] = local_thr(m, sqrt(var), method, k, r);
        int ya = amt_map_index(y + h + 1, H, AMT_MODE_MIRROR), yr = amt_map_index(y - h, H, AMT_MODE_MIRROR);
        s += hs[base + (size_t)ya * W + x];
        s -= hs[base + (size_t)yr * W + x];
        q += hq[base + (size_t)ya * W + x];
        q -= hq[base + (size_t)yr * W + x];
    }
}

// ---- float64: every window summed afresh (rows then columns), fixed order -------------------------
__global__ void __launch_bounds__(256) win_rows_f64_kernel(const double* __restrict__ in, double* __restrict__ hs,
                                                           double* __restrict__ hq, int H, int W, int h) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const size_t base = ((size_t)blockIdx.z * H + blockIdx.y) * W;
    double s = 0.0, q = 0.0;
    for (int dx = -h; dx <= h; ++dx) {
        double v = in[base + amt_map_index(x + dx, W, AMT_MODE_MIRROR)];
        s += v;
        q += v * v;
    }
    hs[base + x] = s;
    hq[base + x] = q;
}

__global__ void __launch_bounds__(256) win_cols_f64_kernel(const double* __restrict__ hs, const double* __restrict__ hq,
                                                           double* __restrict__ thr, int H, int W, int h, int method,
                                                           double k, double r, int hx) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const size_t base = (size_t)blockIdx.z * H * W;
    const double wsz = (double)(2 * h + 1) * (double)(2 * hx + 1);  // rows of the window x its columns
    double s = 0.0, q = 0.0;
    for (int dy = -h; dy <= h; ++dy) {
        int yy = amt_map_index(y + dy, H, AMT_MODE_MIRROR);
        s += hs[base + (size_t)yy * W + x];
        q += hq[base + (size_t)yy * W + x];
    }
    const double m = s / wsz;
    double var = q / wsz - m * m;
    var = var < 0.0 ? 0.0 : var;
    thr[base + (size_t)y * W + x] = local_thr(m, sqrt(var), method, k, r);
}

extern "C" int amt_window_threshold(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nplanes, int H,
                                    int W, int window_size, int method, double k, double r) {
    return amt_window_threshold_yx(ctx, in, in_dtype, thr_image, nplanes, H, W, window_size, window_size, method, k, r);
}

extern "C" int amt_window_threshold_yx(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nplanes, int H,
                                       int W, int window_y, int window_x, int method, double k, double r) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_image && nplanes >= 0 && H > 0 && W > 0, "window_threshold: bad arguments");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "window_threshold: dtype must be AMT_U16 or AMT_F64");
    AMT_REQUIRE(window_y >= 1 && (window_y & 1), "Window size %d is even.", window_y);
    AMT_REQUIRE(window_x >= 1 && (window_x & 1), "Window size %d is even.", window_x);
    AMT_REQUIRE(window_y <= 255 && window_x <= 255, "window_threshold: window %d x %d larger than 255", window_y, window_x);
    AMT_REQUIRE(method == 0 || method == 1, "window_threshold: method must be 0 (niblack) or 1 (sauvola)");
    const int h = window_y / 2, hx = window_x / 2;  // h: rows (the column kernels), hx: columns (the row kernels)
    // windows reaching past the far edge reflect again: amt_map_index is periodic, as numpy's 'reflect' padding is
    if (nplanes == 0) return AMT_OK;
    const size_t np = (size_t)nplanes * H * W;
    if (in_dtype == AMT_U16) {
        AMT_TRY(amt_arena_begin(ctx, amt_align(np * 4) + amt_align(np * 8)));
        unsigned* hs = arena_take_t<unsigned>(ctx, np);
        u64* hq = arena_take_t<u64>(ctx, np);
        size_t smem = (size_t)2 * (W + 2 * hx + 1) * sizeof(u64);
        AMT_REQUIRE(smem <= 150 * 1024, "window_threshold: row too long for the LDS prefix (W = %d)", W);
        hipLaunchKernelGGL(win_rows_u16_kernel, dim3(H, nplanes), dim3(256), smem, ctx->stream, (const uint16_t*)in, hs,
                           hq, H, W, hx);
        AMT_LAUNCH_CHECK();
        const int TH = 64;
        hipLaunchKernelGGL(win_cols_u16_kernel, dim3((W + 255) / 256, (H + TH - 1) / TH, nplanes), dim3(256), 0,
                           ctx->stream, hs, hq, thr_image, H, W, h, method, k, r, TH, hx);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    AMT_TRY(amt_arena_begin(ctx, 2 * amt_align(np * 8)));
    double* hs = arena_take_t<double>(ctx, np);
    double* hq = arena_take_t<double>(ctx, np);
    dim3 grid((W + 255) / 256, H, nplanes);
    hipLaunchKernelGGL(win_rows_f64_kernel, grid, dim3(256), 0, ctx->stream, (const double*)in, hs, hq, H, W, hx);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(win_cols_f64_kernel, grid, dim3(256), 0, ctx->stream, hs, hq, thr_image, H, W, h, method, k, r,
                       hx);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
