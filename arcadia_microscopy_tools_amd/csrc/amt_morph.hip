// Binary morphology on uint8 0/1 masks: erode, dilate, and fused open / close.
//
// Semantics = skimage.morphology.binary_* (SK/morphology/binary.py:42,77,82-147), which call
// scipy.ndimage.binary_erosion(structure, border_value=True) and binary_dilation(structure):
//   erosion : out[p] = AND_{s in S} in[p + s]   (outside the image counts as `border_value`)
//   dilation: out[p] = OR_{s in S}  in[p - s]   (outside counts as `border_value`, 0 in skimage)
// S = the non-zero footprint cells as offsets from the centre.
//
// Layout: one 256-thread workgroup produces a 32 x 64 output tile from an LDS-staged input tile with
// halo; the fused open/close kernels keep the intermediate (tile + one halo) in LDS as well, so a
// pair of primitives costs one HBM read and one HBM write of the mask.
#include "amt_common.h"

constexpr int MT_H = 32;
constexpr int MT_W = 64;
constexpr int MAX_OFFS = 1024;

struct offs_t {
    int n;
    int ry, rx;  // half extents
    // dy, dx pairs follow in device memory
};

// stage `in` (image coords y0-hy .. , x0-hx ..) into LDS; outside image -> `outside`
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ in, size_t plane, int H, int W, int y0, int x0,
                                           int hy, int hx, int th, int tw, uint8_t* __restrict__ lds, int pitch,
                                           uint8_t outside) {
    const int rows = th + 2 * hy, cols = tw + 2 * hx;
    for (int i = threadIdx.x; i < rows * cols; i += 256) {
        int ky = i / cols, kx = i - ky * cols;
        int y = y0 - hy + ky, x = x0 - hx + kx;
        uint8_t v = outside;
        if (y >= 0 && y < H && x >= 0 && x < W) v = in[plane + (size_t)y * W + x] ? 1 : 0;
        lds[ky * pitch + kx] = v;
    }
}

// one primitive over an LDS region: dst (rows x cols, origin = image (oy, ox)) from src whose origin is
// (oy - ry, ox - rx).  ERODE: AND over +offsets; DILATE: OR over -offsets.
// positions of dst outside the image are set to `dst_outside` (border value seen by the NEXT primitive).
template <bool ERODE>
__device__ __forceinline__ void lds_primitive(const uint8_t* __restrict__ src, int spitch, uint8_t* __restrict__ dst,
                                              int dpitch, int rows, int cols, int oy, int ox, int H, int W, int ry,
                                              int rx, const int2* __restrict__ offs, int noffs, uint8_t dst_outside) {
    for (int i = threadIdx.x; i < rows * cols; i += 256) {
        int ky = i / cols, kx = i - ky * cols;
        int y = oy + ky, x = ox + kx;
        uint8_t r;
        if (y < 0 || y >= H || x < 0 || x >= W) {
            r = dst_outside;
        } else {
            const uint8_t* c = src + (ky + ry) * spitch + (kx + rx);
            if (ERODE) {
                r = 1;
                for (int k = 0; k < noffs; ++k) r &= c[offs[k].y * spitch + offs[k].x];
            } else {
                r = 0;
                for (int k = 0; k < noffs; ++k) r |= c[-offs[k].y * spitch - offs[k].x];
            }
        }
        dst[ky * dpitch + kx] = r;
    }
}

template <bool ERODE>
__global__ void __launch_bounds__(256) morph1_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H,
                                                     int W, const int2* __restrict__ offs_g, int noffs, int ry, int rx,
                                                     int border_value) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int pitch = MT_W + 2 * rx;
    uint8_t* tile = reinterpret_cast<uint8_t*>(smem_raw);
    int2* offs = reinterpret_cast<int2*>(smem_raw + amt_align((size_t)(MT_H + 2 * ry) * pitch, 16));
    const int x0 = blockIdx.x * MT_W, y0 = blockIdx.y * MT_H;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = threadIdx.x; i < noffs; i += 256) offs[i] = offs_g[i];
    stage_tile(in, plane, H, W, y0, x0, ry, rx, MT_H, MT_W, tile, pitch, (uint8_t)(border_value ? 1 : 0));
    __syncthreads();
    for (int i = threadIdx.x; i < MT_H * MT_W; i += 256) {
        int ky = i / MT_W, kx = i - ky * MT_W;
        int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        const uint8_t* c = tile + (ky + ry) * pitch + (kx + rx);
        uint8_t r;
        if (ERODE) {
            r = 1;
            for (int k = 0; k < noffs; ++k) r &= c[offs[k].y * pitch + offs[k].x];
        } else {
            r = 0;
            for (int k = 0; k < noffs; ++k) r |= c[-offs[k].y * pitch - offs[k].x];
        }
        out[plane + (size_t)y * W + x] = r;
    }
}

// fused pair: OPEN = dilate(erode(x)) ; CLOSE = erode(dilate(x)) with skimage's border rules
// (erosion sees outside = 1, dilation sees outside = 0).
template <bool OPEN>
__global__ void __launch_bounds__(256) morph2_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H,
                                                     int W, const int2* __restrict__ offs_g, int noffs, int ry,
                                                     int rx) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int p0 = MT_W + 4 * rx;  // input tile pitch (halo 2r)
    const int p1 = MT_W + 2 * rx;  // intermediate pitch (halo r)
    uint8_t* t0 = reinterpret_cast<uint8_t*>(smem_raw);
    size_t o1 = amt_align((size_t)(MT_H + 4 * ry) * p0, 16);
    uint8_t* t1 = t0 + o1;
    size_t o2 = o1 + amt_align((size_t)(MT_H + 2 * ry) * p1, 16);
    int2* offs = reinterpret_cast<int2*>(smem_raw + o2);
    const int x0 = blockIdx.x * MT_W, y0 = blockIdx.y * MT_H;
    const size_t plane = (size_t)blockIdx.z * H * W;
    for (int i = threadIdx.x; i < noffs; i += 256) offs[i] = offs_g[i];
    // first primitive's view of the outside: erosion 1, dilation 0
    stage_tile(in, plane, H, W, y0, x0, 2 * ry, 2 * rx, MT_H, MT_W, t0, p0, (uint8_t)(OPEN ? 1 : 0));
    __syncthreads();
    // intermediate on tile + halo r; outside positions as the SECOND primitive sees them
    if (OPEN)
        lds_primitive<true>(t0, p0, t1, p1, MT_H + 2 * ry, MT_W + 2 * rx, y0 - ry, x0 - rx, H, W, ry, rx, offs, noffs,
                            (uint8_t)0);
    else
        lds_primitive<false>(t0, p0, t1, p1, MT_H + 2 * ry, MT_W + 2 * rx, y0 - ry, x0 - rx, H, W, ry, rx, offs, noffs,
                             (uint8_t)1);
    __syncthreads();
    for (int i = threadIdx.x; i < MT_H * MT_W; i += 256) {
        int ky = i / MT_W, kx = i - ky * MT_W;
        int y = y0 + ky, x = x0 + kx;
        if (y >= H || x >= W) continue;
        const uint8_t* c = t1 + (ky + ry) * p1 + (kx + rx);
        uint8_t r;
        if (OPEN) {
            r = 0;
            for (int k = 0; k < noffs; ++k) r |= c[-offs[k].y * p1 - offs[k].x];
        } else {
            r = 1;
            for (int k = 0; k < noffs; ++k) r &= c[offs[k].y * p1 + offs[k].x];
        }
        out[plane + (size_t)y * W + x] = r;
    }
}

static int build_offsets(const uint8_t* fp, int fh, int fw, int2* host, int* n) {
    AMT_REQUIRE(fp && fh >= 1 && fw >= 1 && (fh & 1) && (fw & 1),
                "footprint must have odd height and width (got %d x %d)", fh, fw);
    AMT_REQUIRE(fh <= 63 && fw <= 63, "footprint larger than 63 x 63 is not supported");
    int k = 0;
    for (int y = 0; y < fh; ++y)
        for (int x = 0; x < fw; ++x)
            if (fp[y * fw + x]) {
                AMT_REQUIRE(k < MAX_OFFS, "footprint has more than %d cells", MAX_OFFS);
                host[k].x = x - fw / 2;
                host[k].y = y - fh / 2;
                ++k;
            }
    AMT_REQUIRE(k > 0, "footprint is empty");
    *n = k;
    return AMT_OK;
}

static int morph_common(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                        const uint8_t* footprint, int fh, int fw, int which, int border_value) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "binary morphology: bad arguments");
    AMT_REQUIRE(in != out, "binary morphology: in-place operation is not supported");
    int2 host[MAX_OFFS];
    int noffs = 0;
    AMT_TRY(build_offsets(footprint, fh, fw, host, &noffs));
    if (nplanes == 0) return AMT_OK;
    const int ry = fh / 2, rx = fw / 2;
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(int2) * noffs)));
    int2* offs = arena_take_t<int2>(ctx, noffs);
    AMT_TRY(amt_param_upload(ctx, offs, host, sizeof(int2) * noffs));
    dim3 grid((W + MT_W - 1) / MT_W, (H + MT_H - 1) / MT_H, nplanes);
    if (which <= 1) {
        size_t smem = amt_align((size_t)(MT_H + 2 * ry) * (MT_W + 2 * rx), 16) + sizeof(int2) * noffs;
        if (which == 0)
            hipLaunchKernelGGL((morph1_kernel<true>), grid, dim3(256), smem, ctx->stream, in, out, H, W, offs, noffs, ry,
                               rx, border_value);
        else
            hipLaunchKernelGGL((morph1_kernel<false>), grid, dim3(256), smem, ctx->stream, in, out, H, W, offs, noffs,
                               ry, rx, border_value);
    } else {
        size_t smem = amt_align((size_t)(MT_H + 4 * ry) * (MT_W + 4 * rx), 16) +
                      amt_align((size_t)(MT_H + 2 * ry) * (MT_W + 2 * rx), 16) + sizeof(int2) * noffs;
        AMT_REQUIRE(smem <= 160 * 1024, "footprint %d x %d too large for the fused open/close kernel", fh, fw);
        if (which == 2)
            hipLaunchKernelGGL((morph2_kernel<true>), grid, dim3(256), smem, ctx->stream, in, out, H, W, offs, noffs, ry,
                               rx);
        else
            hipLaunchKernelGGL((morph2_kernel<false>), grid, dim3(256), smem, ctx->stream, in, out, H, W, offs, noffs,
                               ry, rx);
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_binary_erode(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                                const uint8_t* footprint, int fh, int fw, int border_value) {
    return morph_common(ctx, in, out, nplanes, H, W, footprint, fh, fw, 0, border_value);
}
extern "C" int amt_binary_dilate(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                                 const uint8_t* footprint, int fh, int fw, int border_value) {
    return morph_common(ctx, in, out, nplanes, H, W, footprint, fh, fw, 1, border_value);
}
extern "C" int amt_binary_open(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                               const uint8_t* footprint, int fh, int fw) {
    return morph_common(ctx, in, out, nplanes, H, W, footprint, fh, fw, 2, 0);
}
extern "C" int amt_binary_close(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                                const uint8_t* footprint, int fh, int fw) {
    return morph_common(ctx, in, out, nplanes, H, W, footprint, fh, fw, 3, 0);
}
