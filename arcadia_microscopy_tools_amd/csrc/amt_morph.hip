// Binary morphology on uint8 0/1 masks: erode, dilate, and fused open / close.
//
// Semantics = skimage.morphology.binary_* (SK/morphology/binary.py:42,77,82-147), which call
// scipy.ndimage.binary_erosion(structure, border_value=True) and binary_dilation(structure):
//   erosion : out[p] = AND_{s in S} in[p + s]   (outside the image counts as `border_value`)
//   dilation: out[p] = OR_{s in S}  in[p - s]   (outside counts as `border_value`, 0 in skimage)
// S = the non-zero footprint cells as offsets from the centre.
//
// Layout: the mask is packed to one BIT per pixel (a wave ballot turns 64 consecutive pixels into one
// 64-bit word), the primitives run on the packed rows -- a footprint offset (dy, dx) is one funnel shift
// of (previous | current | next) word of row y + dy followed by an AND / OR, i.e. |S| word operations
// per 64 pixels -- and the result is unpacked to bytes.  The packed plane (H * ceil(W / 64) words; 512 KB
// at 2048^2) lives in L2, so an opening or closing costs one HBM read and one HBM write of the byte mask.
#include "amt_common.h"

typedef unsigned long long u64;
constexpr int MAX_OFFS = 1024;

constexpr int PACK_WORDS_PER_WAVE = 16;

// A wave packs PACK_WORDS_PER_WAVE consecutive words of the (row-major) packed plane: lane i tests pixel
// word * 64 + i of the word's row.  The word coordinates are wave-uniform and advance incrementally (no division per
// word); every load is UNCONDITIONAL from clamped coordinates and nothing is computed from a value next to its load,
// so all PACK_WORDS_PER_WAVE loads of a wave are in flight together (a test inside the bounds check would put a
// wait behind each load).  TEST(value) decides the bit.
template <typename T, typename TEST>
__device__ __forceinline__ void pack_words(const T* __restrict__ src, u64* __restrict__ dst, int H, int W, int WW,
                                           TEST test) {
    const int lane = threadIdx.x & 63;
    const unsigned nwords = (unsigned)H * (unsigned)WW;  // < 2^31: H, W <= 32768
    const unsigned w0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * PACK_WORDS_PER_WAVE);
    if (w0 >= nwords) return;
    const int y0 = (int)(w0 / (unsigned)WW), wx0 = (int)(w0 - (unsigned)y0 * (unsigned)WW);
    T val[PACK_WORDS_PER_WAVE];
    {
        int y = y0, wx = wx0;
#pragma unroll
        for (int k = 0; k < PACK_WORDS_PER_WAVE; ++k) {
            const int yc = y < H ? y : H - 1;
            const int x = wx * 64 + lane;
            val[k] = src[(size_t)yc * W + (x < W ? x : W - 1)];
            if (++wx == WW) {
                wx = 0;
                ++y;
            }
        }
    }
    int y = y0, wx = wx0;
#pragma unroll
    for (int k = 0; k < PACK_WORDS_PER_WAVE; ++k) {
        const bool inside = y < H && wx * 64 + lane < W;  // beyond W: bit 0
        const u64 m = __ballot(inside && test(val[k]));
        if (lane == 0 && y < H) dst[w0 + k] = m;
        if (++wx == WW) {
            wx = 0;
            ++y;
        }
    }
}

__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ in, u64* __restrict__ packed, int H,
                                                   int W, int WW) {
    const size_t plane = blockIdx.y;
    pack_words(in + plane * (size_t)H * W, packed + plane * (size_t)H * WW, H, W, WW,
               [](uint8_t v) { return v != 0; });
}

// same packing, fused with the comparison `in > thr[plane]` (R/operations.py:216): the byte mask of the
// threshold never exists
template <typename T>
__global__ void __launch_bounds__(256) pack_gt_kernel(const T* __restrict__ in, const double* __restrict__ thr,
                                                      u64* __restrict__ packed, int H, int W, int WW) {
    const size_t plane = blockIdx.y;
    const double t = thr[plane];
    pack_words(in + plane * (size_t)H * W, packed + plane * (size_t)H * WW, H, W, WW,
               [t](T v) { return (double)v > t; });
}

// `vals > thr[plane]` from the byte plane of 256-bin indices (amt_otsu_f64_bins): thr is the centre of bin k, so a sample
// of a higher bin is above it (>= that bin's lower edge > centre_k), one of a lower bin below it, and only a sample of
// bin k itself has to be looked at -- 1 byte per pixel instead of 8, plus the float64 values along the objects' rims.
// A thread owns 16 consecutive pixels (one 16-byte load), four threads make a 64-pixel word.  W % 64 == 0.
__global__ void __launch_bounds__(256) pack_bins_kernel(const uint8_t* __restrict__ bins, const double* __restrict__ vals,
                                                        const double* __restrict__ thr,
                                                        const double* __restrict__ thr_code, u64* __restrict__ packed,
                                                        size_t n) {
    const size_t plane = blockIdx.y;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // 16-pixel group of the plane
    if (q * 16 >= n) return;                                  // whole quads leave together (n % 64 == 0)
    const double t = thr[plane];
    const unsigned k = (unsigned)(thr_code[plane] * 0.5);
    const uint4 b16 = *reinterpret_cast<const uint4*>(bins + plane * n + q * 16);
    const unsigned wv[4] = {b16.x, b16.y, b16.z, b16.w};
    const double* v = vals + plane * n + q * 16;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned b = (wv[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        bool bit = b > k;
        if (b == k) bit = v[j] > t;
        m |= (bit ? 1u : 0u) << j;
    }
    const u64 m1 = (u64)(unsigned)__shfl_down((int)m, 1), m2 = (u64)(unsigned)__shfl_down((int)m, 2),
              m3 = (u64)(unsigned)__shfl_down((int)m, 3);
    if ((threadIdx.x & 3) == 0) packed[plane * (n / 64) + q / 4] = (u64)m | (m1 << 16) | (m2 << 32) | (m3 << 48);
}

__device__ __forceinline__ unsigned spread4(unsigned nib) {
    return (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
}

__global__ void __launch_bounds__(256) unpack_kernel(const u64* __restrict__ packed, uint8_t* __restrict__ out, int H,
                                                     int W, int WW) {
    // one thread per 16 pixels of a row (one 16-byte store when W % 16 == 0)
    const size_t plane = blockIdx.y;
    const int per_row = (W + 15) / 16;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (size_t)H * per_row) return;
    const int y = (int)(q / per_row);
    const int x = (int)(q - (size_t)y * per_row) * 16;
    const u64 w = packed[(plane * H + y) * WW + (x >> 6)];
    const unsigned bits = (unsigned)(w >> (x & 63)) & 0xFFFFu;
    uint8_t* o = out + (plane * H + y) * W + x;
    if ((W & 15) == 0) {
        uint4 r;
        r.x = spread4(bits & 0xFu);
        r.y = spread4((bits >> 4) & 0xFu);
        r.z = spread4((bits >> 8) & 0xFu);
        r.w = spread4((bits >> 12) & 0xFu);
        *reinterpret_cast<uint4*>(o) = r;
    } else {
        for (int k = 0; k < 16 && x + k < W; ++k) o[k] = (bits >> k) & 1u;
    }
}

// bits of word `wx` of row y as the primitive sees them: outside the image = border (all ones / zeros),
// and the bits of the last word beyond W also read as border
__device__ __forceinline__ u64 row_word(const u64* __restrict__ rows, int y, int wx, int H, int WW, int W,
                                        u64 border) {
    if (y < 0 || y >= H || wx < 0 || wx >= WW) return border;
    u64 v = rows[(size_t)y * WW + wx];
    if (wx == WW - 1 && (W & 63)) {
        const u64 valid = (1ull << (W & 63)) - 1ull;
        v = (v & valid) | (border & ~valid);
    }
    return v;
}

// offsets are sorted by dy (row-major footprint scan), so the three words of a source row are loaded once
template <bool ERODE>
__global__ void __launch_bounds__(256) packed_prim_kernel(const u64* __restrict__ in, u64* __restrict__ out, int H,
                                                          int W, int WW, const int2* __restrict__ offs_g, int noffs,
                                                          int border_value) {
    __shared__ int2 offs[MAX_OFFS];
    for (int i = threadIdx.x; i < noffs; i += 256) offs[i] = offs_g[i];
    __syncthreads();
    const int wx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (wx >= WW || y >= H) return;
    const u64* rows = in + (size_t)blockIdx.z * H * WW;
    const u64 border = border_value ? ~0ull : 0ull;
    u64 acc = ERODE ? ~0ull : 0ull;
    int cur_dy = 0x7fffffff;
    u64 p = 0, c = 0, nx = 0;
    for (int k = 0; k < noffs; ++k) {
        // erosion reads in[p + s]; dilation reads in[p - s]
        const int dy = ERODE ? offs[k].y : -offs[k].y;
        const int dx = ERODE ? offs[k].x : -offs[k].x;
        if (dy != cur_dy) {
            cur_dy = dy;
            p = row_word(rows, y + dy, wx - 1, H, WW, W, border);
            c = row_word(rows, y + dy, wx, H, WW, W, border);
            nx = row_word(rows, y + dy, wx + 1, H, WW, W, border);
        }
        u64 v;
        if (dx == 0)
            v = c;
        else if (dx > 0)
            v = (c >> dx) | (nx << (64 - dx));  // pixel x + dx
        else
            v = (c << (-dx)) | (p >> (64 + dx));  // pixel x - |dx|
        acc = ERODE ? (acc & v) : (acc | v);
    }
    out[((size_t)blockIdx.z * H + y) * WW + wx] = acc;
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

// which: 0 erode, 1 dilate, 2 open (erode then dilate), 3 close (dilate then erode)
static int morph_common(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                        const uint8_t* footprint, int fh, int fw, int which, int border_value) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "binary morphology: bad arguments");
    static thread_local int2 host[MAX_OFFS];
    int noffs = 0;
    AMT_TRY(build_offsets(footprint, fh, fw, host, &noffs));
    if (nplanes == 0) return AMT_OK;
    const int WW = (W + 63) / 64;
    const size_t words = (size_t)nplanes * H * WW;
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(int2) * noffs) + 2 * amt_align(words * 8)));
    int2* offs = arena_take_t<int2>(ctx, noffs);
    u64* pa = arena_take_t<u64>(ctx, words);
    u64* pb = arena_take_t<u64>(ctx, words);
    AMT_TRY(amt_param_upload(ctx, offs, host, sizeof(int2) * noffs));
    const size_t nwords = (size_t)H * WW;
    const unsigned gpack = (unsigned)((nwords + 4 * PACK_WORDS_PER_WAVE - 1) / (4 * PACK_WORDS_PER_WAVE));
    hipLaunchKernelGGL(pack_kernel, dim3(gpack, nplanes), dim3(256), 0, ctx->stream, in, pa, H, W, WW);
    AMT_LAUNCH_CHECK();
    dim3 gp((WW + 63) / 64, (H + 3) / 4, nplanes);
    u64* result = pb;
    switch (which) {
        case 0:
            hipLaunchKernelGGL((packed_prim_kernel<true>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs,
                               border_value);
            break;
        case 1:
            hipLaunchKernelGGL((packed_prim_kernel<false>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs,
                               border_value);
            break;
        case 2:  // skimage: erosion sees outside = 1, dilation sees outside = 0
            hipLaunchKernelGGL((packed_prim_kernel<true>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs, 1);
            AMT_LAUNCH_CHECK();
            hipLaunchKernelGGL((packed_prim_kernel<false>), gp, dim3(256), 0, ctx->stream, pb, pa, H, W, WW, offs, noffs,
                               0);
            result = pa;
            break;
        default:
            hipLaunchKernelGGL((packed_prim_kernel<false>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs,
                               0);
            AMT_LAUNCH_CHECK();
            hipLaunchKernelGGL((packed_prim_kernel<true>), gp, dim3(256), 0, ctx->stream, pb, pa, H, W, WW, offs, noffs, 1);
            result = pa;
            break;
    }
    AMT_LAUNCH_CHECK();
    const size_t nq = (size_t)H * ((W + 15) / 16);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((nq + 255) / 256), nplanes), dim3(256), 0, ctx->stream, result, out,
                       H, W, WW);
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

// ---- fused threshold -> opening -> closing ---------------------------------------------------------
// `binary_closing(binary_opening(in > thr))` (the Gaussian -> Otsu -> '>' -> open -> close mask chain of
// BASELINE configs[1]/[2]) in two kernels:
//   pack_gt_kernel      streams the image once and leaves one BIT per pixel (high occupancy, HBM-bound);
//   toc_fused_kernel    a block owns a band of TOC_BR rows x up to TOC_TWW words (64 px each): it copies the band
//                       plus a halo of 4 * ry rows (and one halo word left / right) of the packed plane (L2-resident,
//                       1/64 of the image) into LDS, runs erosion, dilation, dilation, erosion LDS -> LDS, and
//                       unpacks the band to bytes -- the only other HBM traffic of the chain.
// Positions outside the IMAGE read as each primitive's own border value (scipy's border_value: 1 for erosion,
// 0 for dilation); positions outside the TILE also read as border, which only corrupts halo words that the band
// never uses.
constexpr int TOC_BR = 32, TOC_TWW = 32, TOC_MAX_OFFS = 256;

// The LDS tile is physically padded by PADR rows above / below and one word left / right that are never
// written, so the primitives need no tile-bound checks; the border substitution for positions outside the
// image (and for the bits of the last word beyond W) is done when a word is WRITTEN, with the border value
// of the primitive that will read it next.
struct toc_tile {
    int TR, TW;          // tile rows / words per tile row (logical)
    int PW;              // physical words per row = TW + 2
    int y_top, wx_left;  // image row of tile row 0, image word of tile column 0
    int H, W, WW;
};
constexpr int TOC_PADR = 4;  // >= ry of the fused path (4 * ry <= 16)

__device__ __forceinline__ u64 toc_fix(const toc_tile& t, int r, int wc, u64 v, u64 next_border) {
    const int y = t.y_top + r, wx = t.wx_left + wc;
    if (y < 0 || y >= t.H || wx < 0 || wx >= t.WW) return next_border;
    if (wx == t.WW - 1 && (t.W & 63)) {
        const u64 valid = (1ull << (t.W & 63)) - 1ull;
        v = (v & valid) | (next_border & ~valid);
    }
    return v;
}

// buf points at logical (row 0, word 0); rows are PW words apart.  Only tile rows [r_lo, r_hi) are produced: each
// primitive of the chain needs ry fewer rows on either side than the one before it.  `inv` = ceil(2^20 / TW) turns
// the row / column split of the flat index into a multiply (exact for idx < 2^11, TW <= 34).
// DISK2: the footprint is disk(2) (the 13 cells |dy| + |dx| <= 2), unrolled with constant shifts; otherwise the
// generic offset list is walked.
template <bool ERODE, bool DISK2>
__device__ __forceinline__ void toc_prim(const u64* src, u64* dst, const toc_tile& t, const int2* offs, int noffs,
                                         u64 next_border, int r_lo, int r_hi, unsigned inv) {
    const int count = (r_hi - r_lo) * t.TW;
    for (int idx = threadIdx.x; idx < count; idx += 256) {
        const int rr = (int)(((unsigned)idx * inv) >> 20);
        const int r = r_lo + rr, wc = idx - rr * t.TW;
        u64 acc;
        if (DISK2) {
            const u64* s0 = src + r * t.PW + wc;
            const u64 a2 = s0[-2 * t.PW], b2 = s0[2 * t.PW];
            const u64 ap = s0[-t.PW - 1], ac = s0[-t.PW], an = s0[-t.PW + 1];
            const u64 cp = s0[-1], cc = s0[0], cn = s0[1];
            const u64 bp = s0[t.PW - 1], bc = s0[t.PW], bn = s0[t.PW + 1];
#define TOC_L(c, p, d) (((c) << (d)) | ((p) >> (64 - (d))))  /* pixel x - d */
#define TOC_R(c, n, d) (((c) >> (d)) | ((n) << (64 - (d))))  /* pixel x + d */
            if (ERODE) {
                acc = a2 & b2 & ac & TOC_L(ac, ap, 1) & TOC_R(ac, an, 1) & bc & TOC_L(bc, bp, 1) & TOC_R(bc, bn, 1) &
                      cc & TOC_L(cc, cp, 1) & TOC_R(cc, cn, 1) & TOC_L(cc, cp, 2) & TOC_R(cc, cn, 2);
            } else {
                acc = a2 | b2 | ac | TOC_L(ac, ap, 1) | TOC_R(ac, an, 1) | bc | TOC_L(bc, bp, 1) | TOC_R(bc, bn, 1) |
                      cc | TOC_L(cc, cp, 1) | TOC_R(cc, cn, 1) | TOC_L(cc, cp, 2) | TOC_R(cc, cn, 2);
            }
#undef TOC_L
#undef TOC_R
        } else {
            acc = ERODE ? ~0ull : 0ull;
            int cur_dy = 0x7fffffff;
            u64 p = 0, c = 0, nx = 0;
            for (int k = 0; k < noffs; ++k) {
                const int dy = ERODE ? offs[k].y : -offs[k].y;
                const int dx = ERODE ? offs[k].x : -offs[k].x;
                if (dy != cur_dy) {
                    cur_dy = dy;
                    const u64* row = src + (r + dy) * t.PW + wc;
                    p = row[-1];
                    c = row[0];
                    nx = row[1];
                }
                u64 v;
                if (dx == 0)
                    v = c;
                else if (dx > 0)
                    v = (c >> dx) | (nx << (64 - dx));
                else
                    v = (c << (-dx)) | (p >> (64 + dx));
                acc = ERODE ? (acc & v) : (acc | v);
            }
        }
        dst[r * t.PW + wc] = toc_fix(t, r, wc, acc, next_border);
    }
}

template <bool DISK2>
__global__ void __launch_bounds__(256) toc_fused_kernel(const u64* __restrict__ packed, uint8_t* __restrict__ out,
                                                        int H, int W, int WW, const int2* __restrict__ offs_g,
                                                        int noffs, int HY) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int2 offs[TOC_MAX_OFFS];
    const int tww = (WW - (int)blockIdx.x * TOC_TWW) < TOC_TWW ? (WW - (int)blockIdx.x * TOC_TWW) : TOC_TWW;
    toc_tile t;
    t.TR = TOC_BR + 2 * HY;
    t.TW = tww + 2;
    t.PW = t.TW + 2;
    t.y_top = (int)blockIdx.y * TOC_BR - HY;
    t.wx_left = (int)blockIdx.x * TOC_TWW - 1;
    t.H = H;
    t.W = W;
    t.WW = WW;
    const int phys = (t.TR + 2 * TOC_PADR) * t.PW;
    u64* physA = reinterpret_cast<u64*>(smem_raw);
    u64* physB = physA + phys;
    u64* bufA = physA + TOC_PADR * t.PW + 1;
    u64* bufB = physB + TOC_PADR * t.PW + 1;
    for (int i = threadIdx.x; i < noffs; i += 256) offs[i] = offs_g[i];
    for (int i = threadIdx.x; i < 2 * phys; i += 256) physA[i] = 0ull;  // padding is never written again
    __syncthreads();
    const size_t plane = blockIdx.z;
    const u64* src = packed + plane * (size_t)H * WW;
    // ---- 1. copy the packed band + halo (outside the image: erosion's border, all ones) ----
    const unsigned inv_tw = ((1u << 20) + (unsigned)t.TW - 1u) / (unsigned)t.TW;  // idx / TW as a multiply (idx < 2^11)
    for (int idx = threadIdx.x; idx < t.TR * t.TW; idx += 256) {
        const int r = (int)(((unsigned)idx * inv_tw) >> 20), wc = idx - r * t.TW;
        const int y = t.y_top + r, wx = t.wx_left + wc;
        const u64 v = (y >= 0 && y < H && wx >= 0 && wx < WW) ? src[(size_t)y * WW + wx] : 0ull;
        bufA[r * t.PW + wc] = toc_fix(t, r, wc, v, ~0ull);
    }
    __syncthreads();
    // ---- 2. opening = erosion, dilation; closing = dilation, erosion (skimage: erosion sees outside = 1,
    //         dilation sees outside = 0) ----
    const int ry = HY / 4;
    const unsigned inv = inv_tw;
    toc_prim<true, DISK2>(bufA, bufB, t, offs, noffs, 0ull, ry, t.TR - ry, inv);
    __syncthreads();
    toc_prim<false, DISK2>(bufB, bufA, t, offs, noffs, 0ull, 2 * ry, t.TR - 2 * ry, inv);
    __syncthreads();
    toc_prim<false, DISK2>(bufA, bufB, t, offs, noffs, ~0ull, 3 * ry, t.TR - 3 * ry, inv);
    __syncthreads();
    toc_prim<true, DISK2>(bufB, bufA, t, offs, noffs, 0ull, 4 * ry, t.TR - 4 * ry, inv);
    __syncthreads();
    // ---- 3. unpack the band: 16 pixels per thread and step ----
    const int per_row = tww * 4;
    const unsigned inv_pr = ((1u << 20) + (unsigned)per_row - 1u) / (unsigned)per_row;  // q < 2^12, per_row <= 128
    for (int q = threadIdx.x; q < TOC_BR * per_row; q += 256) {
        const int r = (int)(((unsigned)q * inv_pr) >> 20), rem = q - r * per_row;
        const int wc = rem >> 2, part = rem & 3;
        const int y = (int)blockIdx.y * TOC_BR + r;
        const int x = ((int)blockIdx.x * TOC_TWW + wc) * 64 + part * 16;
        if (y >= H || x >= W) continue;
        const u64 w = bufA[(HY + r) * t.PW + 1 + wc];
        const unsigned bits = (unsigned)(w >> (part * 16)) & 0xFFFFu;
        uint8_t* o = out + (plane * H + y) * W + x;
        if (x + 15 < W && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
            uint4 rr;
            rr.x = spread4(bits & 0xFu);
            rr.y = spread4((bits >> 4) & 0xFu);
            rr.z = spread4((bits >> 8) & 0xFu);
            rr.w = spread4((bits >> 12) & 0xFu);
            *reinterpret_cast<uint4*>(o) = rr;
        } else {
            for (int k = 0; k < 16 && x + k < W; ++k) o[k] = (bits >> k) & 1u;
        }
    }
}

// `binary_closing(binary_opening(in > thr))` in one packed chain: compare -> 4 word-level primitives ->
// unpack (the Gaussian -> Otsu -> '>' -> open -> close mask chain of BASELINE configs[1]/[2]).
static int threshold_open_close_impl(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev, uint8_t* out,
                                     int nplanes, int H, int W, const uint8_t* footprint, int fh, int fw,
                                     const uint8_t* bins, const double* thr_code_dev);

extern "C" int amt_threshold_open_close(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev,
                                        uint8_t* out, int nplanes, int H, int W, const uint8_t* footprint, int fh,
                                        int fw) {
    return threshold_open_close_impl(ctx, in, in_dtype, thr_dev, out, nplanes, H, W, footprint, fh, fw, nullptr, nullptr);
}

extern "C" int amt_threshold_open_close_bins(amt_ctx* ctx, const double* in, const uint8_t* bins, const double* thr_dev,
                                             const double* thr_code_dev, uint8_t* out, int nplanes, int H, int W,
                                             const uint8_t* footprint, int fh, int fw) {
    AMT_REQUIRE(bins && thr_code_dev, "threshold_open_close_bins: the bin plane and the threshold's bin are required");
    return threshold_open_close_impl(ctx, in, AMT_F64, thr_dev, out, nplanes, H, W, footprint, fh, fw, bins, thr_code_dev);
}

static int threshold_open_close_impl(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev, uint8_t* out,
                                     int nplanes, int H, int W, const uint8_t* footprint, int fh, int fw,
                                     const uint8_t* bins, const double* thr_code_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_dev && out && nplanes >= 0 && H > 0 && W > 0, "threshold_open_close: bad arguments");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "threshold_open_close: dtype must be AMT_U16 or AMT_F64");
    static thread_local int2 host[MAX_OFFS];
    int noffs = 0;
    AMT_TRY(build_offsets(footprint, fh, fw, host, &noffs));
    if (nplanes == 0) return AMT_OK;
    const int WW = (W + 63) / 64;
    const size_t words = (size_t)nplanes * H * WW;
    const int ry = fh / 2, rx = fw / 2;
    if (noffs <= TOC_MAX_OFFS && 4 * ry <= 16 && 4 * rx <= 64) {
        // single fused kernel: halo of 4 * ry rows and one 64-pixel word per side covers the four primitives
        AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(int2) * noffs) + amt_align(words * 8)));
        int2* offs = arena_take_t<int2>(ctx, noffs);
        u64* pa = arena_take_t<u64>(ctx, words);
        AMT_TRY(amt_param_upload(ctx, offs, host, sizeof(int2) * noffs));
        const size_t nwords = (size_t)H * WW;
        const unsigned gpack = (unsigned)((nwords + 4 * PACK_WORDS_PER_WAVE - 1) / (4 * PACK_WORDS_PER_WAVE));
        const size_t npx = (size_t)H * W;
        if (bins && (W & 63) == 0 && ((reinterpret_cast<uintptr_t>(bins) | npx) & 15) == 0)
            hipLaunchKernelGGL(pack_bins_kernel, dim3((unsigned)((npx / 16 + 255) / 256), nplanes), dim3(256), 0, ctx->stream,
                               bins, (const double*)in, thr_dev, thr_code_dev, pa, npx);
        else if (in_dtype == AMT_F64)
            hipLaunchKernelGGL((pack_gt_kernel<double>), dim3(gpack, nplanes), dim3(256), 0, ctx->stream,
                               (const double*)in, thr_dev, pa, H, W, WW);
        else
            hipLaunchKernelGGL((pack_gt_kernel<uint16_t>), dim3(gpack, nplanes), dim3(256), 0, ctx->stream,
                               (const uint16_t*)in, thr_dev, pa, H, W, WW);
        AMT_LAUNCH_CHECK();
        const int HY = 4 * ry;
        const int tww = WW < TOC_TWW ? WW : TOC_TWW;
        const size_t smem = (size_t)2 * (TOC_BR + 2 * HY + 2 * TOC_PADR) * (tww + 4) * sizeof(u64);
        dim3 grid((WW + TOC_TWW - 1) / TOC_TWW, (H + TOC_BR - 1) / TOC_BR, nplanes);
        bool disk2 = fh == 5 && fw == 5 && noffs == 13;  // disk(2): the 13 cells with |dy| + |dx| <= 2
        for (int k = 0; k < noffs && disk2; ++k)
            disk2 = (host[k].x < 0 ? -host[k].x : host[k].x) + (host[k].y < 0 ? -host[k].y : host[k].y) <= 2;
        if (disk2)
            hipLaunchKernelGGL(toc_fused_kernel<true>, grid, dim3(256), smem, ctx->stream, pa, out, H, W, WW, offs,
                               noffs, HY);
        else
            hipLaunchKernelGGL(toc_fused_kernel<false>, grid, dim3(256), smem, ctx->stream, pa, out, H, W, WW, offs,
                               noffs, HY);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(int2) * noffs) + 2 * amt_align(words * 8)));
    int2* offs = arena_take_t<int2>(ctx, noffs);
    u64* pa = arena_take_t<u64>(ctx, words);
    u64* pb = arena_take_t<u64>(ctx, words);
    AMT_TRY(amt_param_upload(ctx, offs, host, sizeof(int2) * noffs));
    const size_t nwords = (size_t)H * WW;
    const unsigned gpack = (unsigned)((nwords + 4 * PACK_WORDS_PER_WAVE - 1) / (4 * PACK_WORDS_PER_WAVE));
    if (in_dtype == AMT_F64)
        hipLaunchKernelGGL((pack_gt_kernel<double>), dim3(gpack, nplanes), dim3(256), 0, ctx->stream, (const double*)in,
                           thr_dev, pa, H, W, WW);
    else
        hipLaunchKernelGGL((pack_gt_kernel<uint16_t>), dim3(gpack, nplanes), dim3(256), 0, ctx->stream,
                           (const uint16_t*)in, thr_dev, pa, H, W, WW);
    AMT_LAUNCH_CHECK();
    dim3 gp((WW + 63) / 64, (H + 3) / 4, nplanes);
    // opening: erosion (outside = 1) then dilation (outside = 0); closing: dilation then erosion
    hipLaunchKernelGGL((packed_prim_kernel<true>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs, 1);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL((packed_prim_kernel<false>), gp, dim3(256), 0, ctx->stream, pb, pa, H, W, WW, offs, noffs, 0);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL((packed_prim_kernel<false>), gp, dim3(256), 0, ctx->stream, pa, pb, H, W, WW, offs, noffs, 0);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL((packed_prim_kernel<true>), gp, dim3(256), 0, ctx->stream, pb, pa, H, W, WW, offs, noffs, 1);
    AMT_LAUNCH_CHECK();
    const size_t nq = (size_t)H * ((W + 15) / 16);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((nq + 255) / 256), nplanes), dim3(256), 0, ctx->stream, pa, out, H,
                       W, WW);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
