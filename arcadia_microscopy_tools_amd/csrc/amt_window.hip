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
                                                           double k, double r, int TH, int hx,
                                                           u64* __restrict__ Sout, u64* __restrict__ Qout) {
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
        if (Sout) {  // the 2-D window sums themselves: an n-D window goes on along the leading axes
            Sout[base + (size_t)y * W + x] = s;
            Qout[base + (size_t)y * W + x] = q;
        } else {
            const double m = (double)s / wsz;
            const double g2 = (double)q / wsz;
            double var = g2 - m * m;
            var = var < 0.0 ? 0.0 : var;
            thr[base + (size_t)y * W + x] = local_thr(m, sqrt(var), method, k, r);
        }
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
                                                           double k, double r, int hx, double* __restrict__ Sout,
                                                           double* __restrict__ Qout) {
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
    if (Sout) {
        Sout[base + (size_t)y * W + x] = s;
        Qout[base + (size_t)y * W + x] = q;
        return;
    }
    const double m = s / wsz;
    double var = q / wsz - m * m;
    var = var < 0.0 ? 0.0 : var;
    thr[base + (size_t)y * W + x] = local_thr(m, sqrt(var), method, k, r);
}

// ---- n-D windows: the 2-D sums of every plane are summed on along each leading axis ----------------
// out[o][t][i] = sum_{dt=-h..h} in[o][mirror(t+dt)][i]  (ascending dt: a fixed order for float64)
template <typename T>
__global__ void __launch_bounds__(256) win_axis_kernel(const T* __restrict__ in, T* __restrict__ out, size_t total, int L,
                                                       size_t inner, int h) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t i = idx % inner, ot = idx / inner;
    const int t = (int)(ot % (size_t)L);
    const size_t o = ot / (size_t)L;
    T acc = 0;
    for (int dt = -h; dt <= h; ++dt) acc += in[(o * L + (size_t)amt_map_index(t + dt, L, AMT_MODE_MIRROR)) * inner + i];
    out[idx] = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) win_finish_kernel(const T* __restrict__ S, const T* __restrict__ Q,
                                                         double* __restrict__ thr, size_t total, double wsz, int method,
                                                         double k, double r) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const double m = (double)S[idx] / wsz;
    double var = (double)Q[idx] / wsz - m * m;
    var = var < 0.0 ? 0.0 : var;
    thr[idx] = local_thr(m, sqrt(var), method, k, r);
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
                           ctx->stream, hs, hq, thr_image, H, W, h, method, k, r, TH, hx, (u64*)nullptr, (u64*)nullptr);
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
                       hx, (double*)nullptr, (double*)nullptr);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

template <typename T>
static int window_nd_tail(amt_ctx* ctx, T* S, T* Q, T* S2, T* Q2, double* thr_image, int nlead, const int* lead_shape,
                          const int* lead_window, size_t plane, double wsz, int method, double k, double r) {
    size_t total = plane;
    for (int a = 0; a < nlead; ++a) total *= (size_t)lead_shape[a];
    const unsigned blocks = (unsigned)((total + 255) / 256);
    for (int a = nlead - 1; a >= 0; --a) {  // innermost leading axis first
        if (lead_window[a] == 1) continue;
        size_t inner = plane;
        for (int b = a + 1; b < nlead; ++b) inner *= (size_t)lead_shape[b];
        hipLaunchKernelGGL((win_axis_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, S, S2, total, lead_shape[a],
                           inner, lead_window[a] / 2);
        hipLaunchKernelGGL((win_axis_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, Q, Q2, total, lead_shape[a],
                           inner, lead_window[a] / 2);
        AMT_LAUNCH_CHECK();
        T* t = S; S = S2; S2 = t;
        t = Q; Q = Q2; Q2 = t;
    }
    hipLaunchKernelGGL((win_finish_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, S, Q, thr_image, total, wsz,
                       method, k, r);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// Niblack / Sauvola of ONE n-D image, the window spanning EVERY axis as in scikit-image (_mean_std builds n-D integral
// images): lead_shape / lead_window describe the axes in front of (Y, X).
extern "C" int amt_window_threshold_nd(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nlead,
                                       const int* lead_shape, const int* lead_window, int H, int W, int window_y,
                                       int window_x, int method, double k, double r) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_image && H > 0 && W > 0 && nlead >= 0 && nlead <= 6 && (nlead == 0 || (lead_shape && lead_window)),
                "window_threshold_nd: bad arguments");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "window_threshold: dtype must be AMT_U16 or AMT_F64");
    AMT_REQUIRE(method == 0 || method == 1, "window_threshold: method must be 0 (niblack) or 1 (sauvola)");
    AMT_REQUIRE(window_y >= 1 && (window_y & 1), "Window size %d is even.", window_y);
    AMT_REQUIRE(window_x >= 1 && (window_x & 1), "Window size %d is even.", window_x);
    AMT_REQUIRE(window_y <= 255 && window_x <= 255, "window_threshold: window %d x %d larger than 255", window_y, window_x);
    size_t nplanes = 1;
    double wsz = (double)window_y * (double)window_x;
    for (int a = 0; a < nlead; ++a) {
        AMT_REQUIRE(lead_shape[a] > 0, "window_threshold_nd: empty axis");
        AMT_REQUIRE(lead_window[a] >= 1 && (lead_window[a] & 1), "Window size %d is even.", lead_window[a]);
        AMT_REQUIRE(lead_window[a] <= 255, "window_threshold: window %d larger than 255", lead_window[a]);
        nplanes *= (size_t)lead_shape[a];
        wsz *= (double)lead_window[a];
    }
    AMT_REQUIRE(nplanes <= 0x7fffffffu, "window_threshold_nd: too many planes");
    const int h = window_y / 2, hx = window_x / 2;
    const size_t plane = (size_t)H * W, np = nplanes * plane;
    if (in_dtype == AMT_U16) {
        AMT_TRY(amt_arena_begin(ctx, amt_align(np * 4) + 5 * amt_align(np * 8)));
        unsigned* hs = arena_take_t<unsigned>(ctx, np);
        u64* hq = arena_take_t<u64>(ctx, np);
        u64 *S = arena_take_t<u64>(ctx, np), *Q = arena_take_t<u64>(ctx, np);
        u64 *S2 = arena_take_t<u64>(ctx, np), *Q2 = arena_take_t<u64>(ctx, np);
        size_t smem = (size_t)2 * (W + 2 * hx + 1) * sizeof(u64);
        AMT_REQUIRE(smem <= 150 * 1024, "window_threshold: row too long for the LDS prefix (W = %d)", W);
        hipLaunchKernelGGL(win_rows_u16_kernel, dim3(H, (unsigned)nplanes), dim3(256), smem, ctx->stream,
                           (const uint16_t*)in, hs, hq, H, W, hx);
        AMT_LAUNCH_CHECK();
        const int TH = 64;
        hipLaunchKernelGGL(win_cols_u16_kernel, dim3((W + 255) / 256, (H + TH - 1) / TH, (unsigned)nplanes), dim3(256), 0,
                           ctx->stream, hs, hq, (double*)nullptr, H, W, h, method, k, r, TH, hx, S, Q);
        AMT_LAUNCH_CHECK();
        return window_nd_tail<u64>(ctx, S, Q, S2, Q2, thr_image, nlead, lead_shape, lead_window, plane, wsz, method, k, r);
    }
    AMT_TRY(amt_arena_begin(ctx, 6 * amt_align(np * 8)));
    double *hs = arena_take_t<double>(ctx, np), *hq = arena_take_t<double>(ctx, np);
    double *S = arena_take_t<double>(ctx, np), *Q = arena_take_t<double>(ctx, np);
    double *S2 = arena_take_t<double>(ctx, np), *Q2 = arena_take_t<double>(ctx, np);
    dim3 grid((W + 255) / 256, H, (unsigned)nplanes);
    hipLaunchKernelGGL(win_rows_f64_kernel, grid, dim3(256), 0, ctx->stream, (const double*)in, hs, hq, H, W, hx);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(win_cols_f64_kernel, grid, dim3(256), 0, ctx->stream, hs, hq, (double*)nullptr, H, W, h, method, k,
                       r, hx, S, Q);
    AMT_LAUNCH_CHECK();
    return window_nd_tail<double>(ctx, S, Q, S2, Q2, thr_image, nlead, lead_shape, lead_window, plane, wsz, method, k, r);
}
