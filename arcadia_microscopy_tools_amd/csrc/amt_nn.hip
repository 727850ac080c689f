// Elementwise glue of the config-5 flow network (BASELINE configs[4]; R/model.py:211 hands the image to Cellpose's
// network -- the convolutions run through PyTorch-ROCm / MIOpen).
//
// A rocprofv3 trace of the stand-in's forward pass (8 tiles of 2 x 1024^2, bf16, channels-last) shows where its time
// goes: convolutions 10.8 ms, and 27.6 ms of memory-bound glue that the eager framework runs as one kernel per
// operation -- batch norm 8.8 ms (35 launches), ReLU 3.8, the additions of skip / style / residual 3.4, and three
// nearest-neighbour upsamplings at 3.5 ms apiece (35 x what their bytes need).  Every convolution of the network is
// "batch norm -> [ReLU] -> conv" on a sum of up to three terms, so ONE pass can feed it:
//     out = act(((x | upsample2x(x)) + y + pre_bias[c] + style[n, c]) * scale[c] + shift[c])
// with scale / shift = the folded inference-time batch norm and pre_bias = the biases of the convolutions that produced
// x and y (the framework adds a convolution's bias in a pass of its own: 3.8 ms per forward; they run without it).  bf16 in and out, arithmetic in float32, one rounding.
#include "amt_internal.h"

__device__ __forceinline__ float nn_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short nn_f32_to_bf16(float f) {  // round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// one thread = 8 consecutive channels of one output pixel (16-byte loads / stores); C % 8 == 0.  A workgroup works inside
// ONE image row (blockIdx.y = n * H + h), so the only per-thread index arithmetic is the split of its position in the row
// into (pixel, channel group) -- a shift when C / 8 is a power of two (it is for every layer of the network); with a flat
// index the three divisions by runtime values cost more than the 16 bytes a thread moves.
__global__ void __launch_bounds__(256) nn_affine_act_kernel(const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                            const float* __restrict__ style,
                                                            const float* __restrict__ pre_bias,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, uint4* __restrict__ out,
                                                            uint4* __restrict__ sum_out, int H, int W, int C8, int c8_shift,
                                                            int relu, int upsample, int row0) {
    const int row = row0 + (int)blockIdx.y;  // n * H + h
    const int n_img = row / H, h = row - n_img * H;  // uniform
    const int per_row = W * C8;
    const size_t row_base = (size_t)row * per_row;
    const size_t src_row = upsample ? ((size_t)n_img * (H >> 1) + (h >> 1)) * (size_t)((W >> 1) * C8) : row_base;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per_row; i += gridDim.x * 256) {
        const int w = c8_shift >= 0 ? i >> c8_shift : i / C8;
        const int cg = i - w * C8;
        const size_t t = row_base + i;
        const size_t src = upsample ? src_row + (size_t)(w >> 1) * C8 + cg : t;
        const uint4 xv = x[src];
        uint4 yv = make_uint4(0, 0, 0, 0);
        if (y) yv = y[t];
        const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = nn_bf16_to_f32((unsigned short)(xs[k] & 0xFFFFu)) + nn_bf16_to_f32((unsigned short)(ys[k] & 0xFFFFu));
            v[2 * k + 1] = nn_bf16_to_f32((unsigned short)(xs[k] >> 16)) + nn_bf16_to_f32((unsigned short)(ys[k] >> 16));
        }
        const int c0 = cg * 8;
        if (pre_bias) {  // the biases of the convolutions that produced x and y (run without them)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += pre_bias[c0 + k];
        }
        if (sum_out) {  // x + y itself is a value of the network (a block's residual input): kept in bf16, style not in it
            uint4 s;
            s.x = nn_f32_to_bf16(v[0]) | ((unsigned)nn_f32_to_bf16(v[1]) << 16);
            s.y = nn_f32_to_bf16(v[2]) | ((unsigned)nn_f32_to_bf16(v[3]) << 16);
            s.z = nn_f32_to_bf16(v[4]) | ((unsigned)nn_f32_to_bf16(v[5]) << 16);
            s.w = nn_f32_to_bf16(v[6]) | ((unsigned)nn_f32_to_bf16(v[7]) << 16);
            sum_out[t] = s;
        }
        if (style) {
            const float* st = style + (size_t)n_img * (C8 * 8) + c0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += st[k];
        }
        unsigned short o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float r = v[k] * scale[c0 + k] + shift[c0 + k];
            if (relu) r = r > 0.f ? r : 0.f;
            o[k] = nn_f32_to_bf16(r);
        }
        uint4 ov;
        ov.x = o[0] | ((unsigned)o[1] << 16);
        ov.y = o[2] | ((unsigned)o[3] << 16);
        ov.z = o[4] | ((unsigned)o[5] << 16);
        ov.w = o[6] | ((unsigned)o[7] << 16);
        out[t] = ov;
    }
}

extern "C" int amt_nn_affine_act_bf16(amt_ctx* ctx, const void* x, const void* y, const float* style,
                                      const float* pre_bias, const float* scale, const float* shift, void* out, void* sum_out, int N, int H, int W, int C, int relu,
                                      int upsample) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(x && scale && shift && out && N >= 0 && H > 0 && W > 0 && C > 0, "nn_affine_act: bad arguments");
    AMT_REQUIRE(C % 8 == 0, "nn_affine_act: the channel count must be a multiple of 8, got %d", C);
    AMT_REQUIRE(!upsample || (H % 2 == 0 && W % 2 == 0), "nn_affine_act: upsampling needs even output sides");
    AMT_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(y) |
                  reinterpret_cast<uintptr_t>(sum_out)) & 15) == 0,
                "nn_affine_act: tensors must be 16-byte aligned");
    if (N == 0) return AMT_OK;
    AMT_REQUIRE((size_t)N * H < 0x7fffffffull && (size_t)W * (C / 8) < 0x7fffffffull, "nn_affine_act: tensor too large");
    const int C8 = C / 8;
    int c8_shift = -1;
    if ((C8 & (C8 - 1)) == 0)
        for (c8_shift = 0; (1 << c8_shift) < C8; ++c8_shift) {}
    const int rows = N * H;
    for (int row0 = 0; row0 < rows; row0 += 65535) {  // grid.y is limited to 65,535 rows per launch
        const int nr = rows - row0 < 65535 ? rows - row0 : 65535;
        hipLaunchKernelGGL(nn_affine_act_kernel, dim3(amt_grid_for((size_t)W * C8, 256, 64), (unsigned)nr), dim3(256), 0,
                           ctx->stream, (const uint4*)x, (const uint4*)y, style, pre_bias, scale, shift, (uint4*)out,
                           (uint4*)sum_out, H, W, C8, c8_shift, relu, upsample, row0);
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
