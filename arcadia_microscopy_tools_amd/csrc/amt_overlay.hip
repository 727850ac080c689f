// Channel overlay: background + colour-mapped fluorescence layers -> RGB (R/blending.py:116-226).
//
// Per pixel, in the reference's float64 evaluation order (no FMA contraction):
//   canvas = (b, b, b), b = clip(background, 0, 1)
//   for every layer: x = clip(intensity, 0, 1); rgba = lut[trunc(x * 256), 256 folded into 255]
//                    alpha = opacity * rgba.a
//                    ALPHA   : canvas = clip(alpha * rgb + (1 - alpha) * canvas, 0, 1)
//                    ADDITIVE: canvas = clip(canvas + alpha * rgb, 0, 1)
// The 256-entry tables are matplotlib's LinearSegmentedColormap tables, built by the host layer with numpy exactly
// as matplotlib builds them (oracle/blending.py pins both against matplotlib 3.10.8).  All layers are composited in
// ONE pass: every intensity plane is read once, the canvas never round-trips through HBM between layers.
#include "amt_internal.h"

constexpr int OV_MAX_LAYERS = 8;

struct ov_params {
    const double* layer[OV_MAX_LAYERS];
    double opacity[OV_MAX_LAYERS];
    int mode[OV_MAX_LAYERS];
    int nlayers;
};

__device__ __forceinline__ double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

__global__ void __launch_bounds__(256) overlay_kernel(const double* __restrict__ bg, const ov_params* __restrict__ pp,
                                                      const double* __restrict__ luts, double* __restrict__ out,
                                                      size_t n) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lut = reinterpret_cast<double*>(smem_raw);  // nlayers x 256 x 4
    __shared__ ov_params p;
    if (threadIdx.x == 0) p = *pp;
    __syncthreads();
    for (int i = threadIdx.x; i < p.nlayers * 1024; i += 256) lut[i] = luts[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double b = clip01(bg[i]);
        double r = b, g = b, bl = b;
        for (int l = 0; l < p.nlayers; ++l) {
            const double x = clip01(p.layer[l][i]);
            double xa = x * 256.0;
            if (xa == 256.0) xa = 255.0;
            int idx = (int)xa;  // truncation, as ndarray.astype(int) does
            idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
            const double* c = lut + ((size_t)l * 256 + idx) * 4;
            const double alpha = p.opacity[l] * c[3];
            if (p.mode[l] == 1) {
                r = clip01(r + alpha * c[0]);
                g = clip01(g + alpha * c[1]);
                bl = clip01(bl + alpha * c[2]);
            } else {
                const double ia = 1.0 - alpha;
                r = clip01(alpha * c[0] + ia * r);
                g = clip01(alpha * c[1] + ia * g);
                bl = clip01(alpha * c[2] + ia * bl);
            }
        }
        out[3 * i + 0] = r;
        out[3 * i + 1] = g;
        out[3 * i + 2] = bl;
    }
}

extern "C" int amt_overlay(amt_ctx* ctx, const double* background, const double* const* layers_host, int nlayers,
                           const double* luts_host, const double* opacity_host, const int32_t* mode_host,
                           double* out_rgb, int H, int W) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(background && out_rgb && H > 0 && W > 0, "overlay: bad arguments");
    AMT_REQUIRE(nlayers >= 0 && nlayers <= OV_MAX_LAYERS, "overlay: at most %d layers per call (got %d)", OV_MAX_LAYERS,
                nlayers);
    AMT_REQUIRE(nlayers == 0 || (layers_host && luts_host && opacity_host && mode_host), "overlay: null layer tables");
    ov_params hp;
    memset(&hp, 0, sizeof(hp));
    hp.nlayers = nlayers;
    for (int l = 0; l < nlayers; ++l) {
        AMT_REQUIRE(layers_host[l] != nullptr, "overlay: layer %d is null", l);
        AMT_REQUIRE(opacity_host[l] >= 0.0 && opacity_host[l] <= 1.0, "Opacity must be in [0, 1], got %g", opacity_host[l]);
        AMT_REQUIRE(mode_host[l] == 0 || mode_host[l] == 1, "overlay: blend mode must be 0 (alpha) or 1 (additive)");
        hp.layer[l] = layers_host[l];
        hp.opacity[l] = opacity_host[l];
        hp.mode[l] = mode_host[l];
    }
    const size_t lbytes = (size_t)(nlayers > 0 ? nlayers : 1) * 1024 * sizeof(double);
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(ov_params)) + amt_align(lbytes)));
    ov_params* dp = (ov_params*)amt_arena_take(ctx, sizeof(ov_params));
    double* dl = (double*)amt_arena_take(ctx, lbytes);
    AMT_TRY(amt_param_upload(ctx, dp, &hp, sizeof(hp)));
    if (nlayers) AMT_TRY(amt_param_upload(ctx, dl, luts_host, (size_t)nlayers * 1024 * sizeof(double)));
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(overlay_kernel, dim3(amt_grid_for(n, 256, 2048)), dim3(256), lbytes, ctx->stream, background, dp, dl,
                       out_rgb, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
