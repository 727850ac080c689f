// Connected-component labelling, clear_border, relabel_sequential and label utilities.
//
// Reference call sites: R/masks.py:56 (ski.segmentation.clear_border), :63 (ski.measure.label),
// :65 (ski.segmentation.relabel_sequential), :399-403 (np.isin / np.where in filter()).
// Contract: components of EQUAL-valued non-zero pixels, 8-connected by default,
// numbered 1..K in raster order of each component's first pixel.
//
// Algorithm: lock-free union-find in HBM whose root is the component's minimum flat index (= its
// first pixel in raster order), so the raster numbering is a prefix sum over the root flags.
#include "amt_internal.h"

__device__ __forceinline__ int uf_find(const int* __restrict__ L, int a) {
    int p = L[a];
    while (p != a) {
        a = p;
        p = L[a];
    }
    return a;
}

__device__ __forceinline__ int uf_find_volatile(int* L, int a) {
    int p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != a) {
        a = p;
        p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return a;
}

// union by minimum index (Komura-style): the larger root is redirected to the smaller one.
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find_volatile(L, a);
        b = uf_find_volatile(L, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        // a > b: try to hang a below b
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;  // someone else moved a meanwhile; retry from there
    }
}

// Initial labels: every foreground pixel points at the first pixel of its horizontal run of equal values
// inside its 64-pixel wave segment (found with one ballot), so whole runs are already one tree of
// depth 1 and the merge step only has to stitch runs.
template <typename T>
__global__ void __launch_bounds__(256) ccl_init_kernel(const T* __restrict__ in, int* __restrict__ Lall, int H, int W) {
    const size_t n = (size_t)H * W;
    const T* img = in + (size_t)blockIdx.z * n;
    int* L = Lall + (size_t)blockIdx.z * n;
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int p = y * W + (x < W ? x : W - 1);
    const long long v = x < W ? (long long)img[p] : -1;  // -1 never equals a pixel value of the lanes inside
    const long long left = __shfl_up(v, 1);
    const bool head = (lane == 0) || (left != v);
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull);
    const int start_lane = 63 - __clzll((long long)upto);
    if (x < W) L[p] = v != 0 ? p - (lane - start_lane) : -1;
}

// Merge step.  Runs are stitched (a) across 64-pixel segment boundaries (lane 0 with its west pixel) and
// (b) vertically: p joins its north pixel q only when p or q starts a run -- otherwise (p-1, q-1) is the
// same pair of runs and is handled further left.  For 8-connectivity the diagonals matter only when north
// differs: NW unless west matches (then west reaches NW as its own north), NE unless east matches (then
// east reaches NE as its own north).
template <typename T>
__global__ void __launch_bounds__(256) ccl_merge_kernel(const T* __restrict__ in, int* __restrict__ Lall, int H, int W,
                                                        int conn8) {
    const size_t n = (size_t)H * W;
    const T* img = in + (size_t)blockIdx.z * n;
    int* L = Lall + (size_t)blockIdx.z * n;
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int p = y * W + x;
    const T v = img[p];
    if (v == 0) return;
    const bool w_same = x > 0 && img[p - 1] == v;
    if (w_same && lane == 0) uf_union(L, p, p - 1);
    if (y > 0) {
        const int q = p - W;
        if (img[q] == v) {
            if (!w_same || img[q - 1] != v) uf_union(L, p, q);
        } else if (conn8) {
            if (x > 0 && !w_same && img[q - 1] == v) uf_union(L, p, q - 1);
            if (x + 1 < W && img[q + 1] == v && img[p + 1] != v) uf_union(L, p, q + 1);
        }
    }
}

// ---- raster renumbering: exclusive prefix sum over root flags ----------------------------------
constexpr int RN_CHUNK = 2048;  // pixels per block in the compress / rank passes (256 threads x 8)

// Path compression (every pixel points at its root afterwards) fused with the per-block root count that
// the renumbering needs: a pixel is a root iff L[p] == p, which compression never changes.
__global__ void __launch_bounds__(256) ccl_compress_count_kernel(int* __restrict__ L, int* __restrict__ blockcnt,
                                                                 size_t n, int nblk) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        if (i < n) {
            const int l = L[base + i];
            if (l >= 0) {
                int r = l;
                int p = L[base + r];
                while (p != r) {
                    r = p;
                    p = L[base + r];
                }
                if (r != l) L[base + i] = r;
                c += (r == (int)i) ? 1 : 0;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && blockcnt) blockcnt[(size_t)blockIdx.y * nblk + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// rank of every root (1-based) written at the root's own position of T
__global__ void __launch_bounds__(256) root_rank_kernel(const int* __restrict__ L, const int* __restrict__ blockoff,
                                                        int* __restrict__ T, size_t n, int nblk) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    __shared__ int wave_tot[4];
    __shared__ int running;
    if (threadIdx.x == 0) running = blockoff[(size_t)blockIdx.y * nblk + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 8; ++k) {
        size_t i = start + (size_t)k * 256 + threadIdx.x;
        bool is_root = i < n && L[base + i] == (int)i;
        unsigned long long m = __ballot(is_root);
        int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        int run = running;
        if (is_root) T[base + i] = run + woff + before + 1;
        __syncthreads();
        if (threadIdx.x == 0) running = run + wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) apply_rank_kernel(const int* __restrict__ L, const int* __restrict__ T,
                                                         int* __restrict__ out, size_t n) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int l = L[base + i];
        out[base + i] = l >= 0 ? T[base + l] : 0;
    }
}

template <typename T>
static int ccl_roots(amt_ctx* ctx, const T* in, int* L, int* blk, int nplanes, int H, int W, int conn8) {
    const size_t n = (size_t)H * W;
    const int nblk = amt_i_rank_blocks(n);
    dim3 g2((W + 63) / 64, (H + 3) / 4, nplanes);
    hipLaunchKernelGGL((ccl_init_kernel<T>), g2, dim3(256), 0, ctx->stream, in, L, H, W);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL((ccl_merge_kernel<T>), g2, dim3(256), 0, ctx->stream, in, L, H, W, conn8);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ccl_compress_count_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, L, blk, n, nblk);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_i_ccl_roots(amt_ctx* ctx, const void* in, int in_dtype, int* L, int* blk, int nplanes, int H, int W,
                    int conn8) {
    if (in_dtype == AMT_U8) return ccl_roots<uint8_t>(ctx, (const uint8_t*)in, L, blk, nplanes, H, W, conn8);
    return ccl_roots<int32_t>(ctx, (const int32_t*)in, L, blk, nplanes, H, W, conn8);
}

int amt_i_rank_blocks(size_t n) { return (int)((n + RN_CHUNK - 1) / RN_CHUNK); }

int amt_i_rank_roots(amt_ctx* ctx, const int* L, int* T, int* blk, int* count_dev, int nplanes, size_t n) {
    const int nblk = amt_i_rank_blocks(n);  // blk holds the per-block root counts written by amt_i_ccl_roots
    AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, count_dev, nplanes));
    hipLaunchKernelGGL(root_rank_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, L, blk, T, n, nblk);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_label(amt_ctx* ctx, const void* in, int in_dtype, int32_t* out, int32_t* count_dev, int nplanes,
                         int H, int W, int connectivity) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "label: bad arguments");
    AMT_REQUIRE(in_dtype == AMT_U8 || in_dtype == AMT_I32, "label: in_dtype must be AMT_U8 or AMT_I32");
    AMT_REQUIRE(connectivity == 1 || connectivity == 2, "label: connectivity must be 1 or 2");
    AMT_REQUIRE((size_t)H * W < 0x7fffffffull, "label: plane too large");
    AMT_REQUIRE((const void*)in != (const void*)out, "label: in-place operation is not supported");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int nblk = amt_i_rank_blocks(n);
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 4) + amt_align((size_t)nplanes * nblk * 4)));
    int* T = arena_take_t<int>(ctx, (size_t)nplanes * n);
    int* blk = arena_take_t<int>(ctx, (size_t)nplanes * nblk);
    int* L = out;
    AMT_TRY(amt_i_ccl_roots(ctx, in, in_dtype, L, blk, nplanes, H, W, connectivity == 2));
    AMT_TRY(amt_i_rank_roots(ctx, L, T, blk, count_dev, nplanes, n));
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    hipLaunchKernelGGL(apply_rank_kernel, g1, dim3(256), 0, ctx->stream, L, T, out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- clear_border ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) frame_flag_kernel(const int* __restrict__ L, int* __restrict__ T, int H, int W) {
    const size_t n = (size_t)H * W;
    const size_t base = (size_t)blockIdx.y * n;
    const int perim = 2 * W + 2 * H;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < perim; k += gridDim.x * 256) {
        int y, x;
        if (k < W) {
            y = 0;
            x = k;
        } else if (k < 2 * W) {
            y = H - 1;
            x = k - W;
        } else if (k < 2 * W + H) {
            y = k - 2 * W;
            x = 0;
        } else {
            y = k - 2 * W - H;
            x = W - 1;
        }
        int l = L[base + (size_t)y * W + x];
        if (l >= 0) T[base + l] = 1;
    }
}

__global__ void __launch_bounds__(256) clear_flagged_kernel(const int* __restrict__ in, const int* __restrict__ L,
                                                            const int* __restrict__ T, int* __restrict__ out,
                                                            size_t n) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int l = L[base + i];
        int v = in[base + i];
        out[base + i] = (l >= 0 && T[base + l]) ? 0 : v;
    }
}

extern "C" int amt_clear_border(amt_ctx* ctx, const int32_t* in, int32_t* out, int nplanes, int H, int W) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "clear_border: bad arguments");
    AMT_REQUIRE((size_t)H * W < 0x7fffffffull, "clear_border: plane too large");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    AMT_TRY(amt_arena_begin(ctx, 2 * amt_align((size_t)nplanes * n * 4)));
    int* L = arena_take_t<int>(ctx, (size_t)nplanes * n);
    int* T = arena_take_t<int>(ctx, (size_t)nplanes * n);
    AMT_HIP_CHECK(hipMemsetAsync(T, 0, (size_t)nplanes * n * 4, ctx->stream));
    AMT_TRY(ccl_roots<int32_t>(ctx, in, L, nullptr, nplanes, H, W, 1));
    dim3 gf(amt_grid_for((size_t)2 * W + 2 * H, 256, 64), nplanes);
    hipLaunchKernelGGL(frame_flag_kernel, gf, dim3(256), 0, ctx->stream, L, T, H, W);
    AMT_LAUNCH_CHECK();
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    hipLaunchKernelGGL(clear_flagged_kernel, g1, dim3(256), 0, ctx->stream, in, L, T, out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- relabel_sequential ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) presence_kernel(const int* __restrict__ in, int* __restrict__ present, size_t n,
                                                       int max_label) {
    const size_t base = (size_t)blockIdx.y * n;
    int* P = present + (size_t)blockIdx.y * (max_label + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int v = in[base + i];
        if (v > 0 && v <= max_label && P[v] == 0) P[v] = 1;
    }
}

// in place: present[l] (0/1) -> new label of l (0 if absent); label 0 -> 0
__global__ void __launch_bounds__(1024) presence_scan_kernel(int* __restrict__ present, int max_label,
                                                             int* __restrict__ count_dev) {
    __shared__ int s[1024];
    __shared__ int carry;
    int* P = present + (size_t)blockIdx.x * (max_label + 1);
    if (threadIdx.x == 0) {
        carry = 0;
        P[0] = 0;
    }
    __syncthreads();
    for (int start = 1; start <= max_label; start += 1024) {
        int i = start + threadIdx.x;
        int v = i <= max_label ? P[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = s[threadIdx.x];
        int c = carry;
        if (i <= max_label) P[i] = v ? c + incl : 0;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && count_dev) count_dev[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(256) map_labels_kernel(const int* __restrict__ in, const int* __restrict__ map,
                                                         int* __restrict__ out, size_t n, int max_label) {
    const size_t base = (size_t)blockIdx.y * n;
    const int* M = map + (size_t)blockIdx.y * (max_label + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int v = in[base + i];
        out[base + i] = (v > 0 && v <= max_label) ? M[v] : 0;
    }
}

extern "C" int amt_relabel_sequential(amt_ctx* ctx, const int32_t* in, int32_t* out, int32_t* count_dev, int nplanes,
                                      size_t n, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && max_label >= 0, "relabel_sequential: bad arguments");
    if (nplanes == 0) return AMT_OK;
    size_t msz = (size_t)nplanes * ((size_t)max_label + 1);
    AMT_TRY(amt_arena_begin(ctx, amt_align(msz * 4)));
    int* P = arena_take_t<int>(ctx, msz);
    AMT_HIP_CHECK(hipMemsetAsync(P, 0, msz * 4, ctx->stream));
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    if (n) {
        hipLaunchKernelGGL(presence_kernel, g1, dim3(256), 0, ctx->stream, in, P, n, max_label);
        AMT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(presence_scan_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, P, max_label, count_dev);
    AMT_LAUNCH_CHECK();
    if (n) {
        hipLaunchKernelGGL(map_labels_kernel, g1, dim3(256), 0, ctx->stream, in, P, out, n, max_label);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

// ---- clear_border + relabel_sequential fused, for label images whose labels are each ONE component --
// (outputs of amt_label / amt_watershed_*): a label touches the border iff one of its pixels lies on the
// frame, so no re-labelling is needed.  present[l] = 1 if l occurs, 2 if it also touches the frame.
__global__ void __launch_bounds__(256) frame_mark_kernel(const int* __restrict__ in, int* __restrict__ present, int H,
                                                         int W, int max_label) {
    const size_t n = (size_t)H * W;
    const int* img = in + (size_t)blockIdx.y * n;
    int* P = present + (size_t)blockIdx.y * (max_label + 1);
    const int perim = 2 * W + 2 * H;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < perim; k += gridDim.x * 256) {
        int y, x;
        if (k < W) {
            y = 0;
            x = k;
        } else if (k < 2 * W) {
            y = H - 1;
            x = k - W;
        } else if (k < 2 * W + H) {
            y = k - 2 * W;
            x = 0;
        } else {
            y = k - 2 * W - H;
            x = W - 1;
        }
        int v = img[(size_t)y * W + x];
        if (v > 0 && v <= max_label) P[v] = 2;
    }
}

__global__ void __launch_bounds__(256) drop_flagged_kernel(int* __restrict__ present, int max_label) {
    int* P = present + (size_t)blockIdx.y * (max_label + 1);
    for (int l = blockIdx.x * 256 + threadIdx.x; l <= max_label; l += gridDim.x * 256) P[l] = (P[l] == 1) ? 1 : 0;
}

extern "C" int amt_clear_border_relabel(amt_ctx* ctx, const int32_t* in, int32_t* out, int32_t* count_dev, int nplanes,
                                        int H, int W, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0, "clear_border_relabel: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    size_t msz = (size_t)nplanes * ((size_t)max_label + 1);
    AMT_TRY(amt_arena_begin(ctx, amt_align(msz * 4)));
    int* P = arena_take_t<int>(ctx, msz);
    AMT_HIP_CHECK(hipMemsetAsync(P, 0, msz * 4, ctx->stream));
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    hipLaunchKernelGGL(presence_kernel, g1, dim3(256), 0, ctx->stream, in, P, n, max_label);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(frame_mark_kernel, dim3(amt_grid_for((size_t)2 * W + 2 * H, 256, 64), nplanes), dim3(256), 0,
                       ctx->stream, in, P, H, W, max_label);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(drop_flagged_kernel, dim3(amt_grid_for((size_t)max_label + 1, 256, 64), nplanes), dim3(256), 0,
                       ctx->stream, P, max_label);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(presence_scan_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, P, max_label, count_dev);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(map_labels_kernel, g1, dim3(256), 0, ctx->stream, in, P, out, n, max_label);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

__global__ void __launch_bounds__(256) keep_labels_kernel(const int* __restrict__ in, const uint8_t* __restrict__ keep,
                                                          int* __restrict__ out, size_t n, int max_label) {
    const size_t base = (size_t)blockIdx.y * n;
    const uint8_t* K = keep + (size_t)blockIdx.y * (max_label + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int v = in[base + i];
        out[base + i] = (v > 0 && v <= max_label && K[v]) ? v : 0;
    }
}

extern "C" int amt_keep_labels(amt_ctx* ctx, const int32_t* in, const uint8_t* keep_dev, int32_t* out, int nplanes,
                               size_t n, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && keep_dev && out && nplanes >= 0 && max_label >= 0, "keep_labels: bad arguments");
    if (nplanes == 0 || n == 0) return AMT_OK;
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    hipLaunchKernelGGL(keep_labels_kernel, g1, dim3(256), 0, ctx->stream, in, keep_dev, out, n, max_label);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

__global__ void cast_i32_i64_kernel(const int* __restrict__ in, long long* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (long long)in[i];
}

extern "C" int amt_cast_i32_i64(amt_ctx* ctx, const int32_t* in, int64_t* out, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out, "cast_i32_i64: null pointer");
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(cast_i32_i64_kernel, dim3(amt_grid_for(n, 256)), dim3(256), 0, ctx->stream, in,
                       (long long*)out, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

__global__ void __launch_bounds__(256) max_i32_kernel(const int* __restrict__ in, int* __restrict__ mx, size_t n) {
    const size_t base = (size_t)blockIdx.y * n;
    int m = -2147483647 - 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int v = in[base + i];
        m = v > m ? v : m;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&mx[blockIdx.y], m);
}

__global__ void fill_i32_kernel(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" int amt_max_i32(amt_ctx* ctx, const int32_t* in, int32_t* max_dev, int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && max_dev && nplanes >= 0, "max_i32: bad arguments");
    if (nplanes == 0) return AMT_OK;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, max_dev,
                       -2147483647 - 1, nplanes);
    AMT_LAUNCH_CHECK();
    if (n) {
        dim3 g(amt_grid_for(n, 256 * 8, 512), nplanes);
        hipLaunchKernelGGL(max_i32_kernel, g, dim3(256), 0, ctx->stream, in, max_dev, n);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

// ---- sparse labelling ---------------------------------------------------------------------------------
// For masks with few foreground pixels (the EDT peak markers: a few thousand pixels in 4 Mpx) the dense
// passes above mostly stream background.  Here the foreground pixels are compacted IN RASTER ORDER into a
// list (ballot ranks + block prefix), `out` doubles as the pixel -> list-index map, union-find runs on the
// list only (root = smallest list index = first raster pixel of the component), roots are ranked by a scan
// over the list and the labels are scattered back.  Same numbering as amt_label by construction.
__global__ void __launch_bounds__(256) sp_count_kernel(const uint8_t* __restrict__ in, int* __restrict__ blockcnt,
                                                       size_t n, int nblk) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        if (i < n) c += in[base + i] != 0 ? 1 : 0;
    }
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[(size_t)blockIdx.y * nblk + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256) sp_compact_kernel(const uint8_t* __restrict__ in, const int* __restrict__ blockoff,
                                                         int* __restrict__ list, int* __restrict__ parent,
                                                         int* __restrict__ out, size_t n, int nblk, int cap) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    int* lst = list + (size_t)blockIdx.y * cap;
    int* par = parent + (size_t)blockIdx.y * cap;
    __shared__ int wave_tot[4];
    __shared__ int running;
    if (threadIdx.x == 0) running = blockoff[(size_t)blockIdx.y * nblk + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        const bool fg = i < n && in[base + i] != 0;
        const unsigned long long m = __ballot(fg);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        const int run = running;
        if (fg) {
            const int idx = run + woff + before;
            if (idx < cap) {
                lst[idx] = (int)i;
                par[idx] = idx;
                out[base + i] = idx + 1;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) running = run + wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) sp_merge_kernel(const int* __restrict__ list, int* __restrict__ parent,
                                                       const int* __restrict__ out, const int* __restrict__ total,
                                                       int H, int W, int cap, int conn8) {
    const size_t n = (size_t)H * W;
    const int K = total[blockIdx.y] < cap ? total[blockIdx.y] : cap;
    const int* lst = list + (size_t)blockIdx.y * cap;
    int* par = parent + (size_t)blockIdx.y * cap;
    const int* o = out + (size_t)blockIdx.y * n;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
        const int p = lst[k];
        const int y = p / W, x = p - y * W;
        if (x > 0 && o[p - 1] > 0) uf_union(par, k, o[p - 1] - 1);
        if (y > 0) {
            if (o[p - W] > 0) uf_union(par, k, o[p - W] - 1);
            if (conn8) {
                if (x > 0 && o[p - W - 1] > 0) uf_union(par, k, o[p - W - 1] - 1);
                if (x + 1 < W && o[p - W + 1] > 0) uf_union(par, k, o[p - W + 1] - 1);
            }
        }
    }
}

// flags[k] = 1 for roots (after full compression of parent[k])
__global__ void __launch_bounds__(256) sp_compress_kernel(int* __restrict__ parent, int* __restrict__ flags,
                                                          const int* __restrict__ total, int cap) {
    const int K = total[blockIdx.y] < cap ? total[blockIdx.y] : cap;
    int* par = parent + (size_t)blockIdx.y * cap;
    int* fl = flags + (size_t)blockIdx.y * cap;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
        int r = par[k];
        int q = par[r];
        while (q != r) {
            r = q;
            q = par[r];
        }
        par[k] = r;
        fl[k] = (r == k) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) sp_write_kernel(const int* __restrict__ list, const int* __restrict__ parent,
                                                       const int* __restrict__ rank, const int* __restrict__ total,
                                                       const int* __restrict__ nroots, int* __restrict__ out,
                                                       int* __restrict__ count_dev, size_t n, int cap) {
    const int tot = total[blockIdx.y];
    const int K = tot < cap ? tot : cap;
    const int* lst = list + (size_t)blockIdx.y * cap;
    const int* par = parent + (size_t)blockIdx.y * cap;
    const int* rk = rank + (size_t)blockIdx.y * cap;
    int* o = out + (size_t)blockIdx.y * n;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) o[lst[k]] = rk[par[k]] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0 && count_dev) count_dev[blockIdx.y] = tot > cap ? -1 : nroots[blockIdx.y];
}

__global__ void sp_clamp_kernel(int* total_clamped, const int* total, int cap, int nplanes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nplanes) total_clamped[i] = total[i] < cap ? total[i] : cap;
}

extern "C" int amt_label_sparse(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H,
                                int W, int connectivity, int capacity) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "label_sparse: bad arguments");
    AMT_REQUIRE(connectivity == 1 || connectivity == 2, "label_sparse: connectivity must be 1 or 2");
    AMT_REQUIRE(capacity >= 1, "label_sparse: capacity must be positive");
    AMT_REQUIRE((size_t)H * W < 0x7fffffffull, "label_sparse: plane too large");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int nblk = amt_i_rank_blocks(n);
    const size_t capn = (size_t)nplanes * capacity;
    AMT_TRY(amt_arena_begin(ctx, 3 * amt_align(capn * 4) + amt_align((size_t)nplanes * nblk * 4) +
                                     3 * amt_align(nplanes * 4)));
    int* list = arena_take_t<int>(ctx, capn);
    int* parent = arena_take_t<int>(ctx, capn);
    int* flags = arena_take_t<int>(ctx, capn);
    int* blk = arena_take_t<int>(ctx, (size_t)nplanes * nblk);
    int* total = arena_take_t<int>(ctx, nplanes);
    int* total_c = arena_take_t<int>(ctx, nplanes);
    int* nroots = arena_take_t<int>(ctx, nplanes);
    AMT_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)nplanes * n * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(sp_count_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, in, blk, n, nblk);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, total, nplanes));
    hipLaunchKernelGGL(sp_compact_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, in, blk, list, parent, out, n,
                       nblk, capacity);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sp_clamp_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, total_c, total, capacity,
                       nplanes);
    AMT_LAUNCH_CHECK();
    const unsigned gk = amt_grid_for((size_t)capacity, 256, 64);
    hipLaunchKernelGGL(sp_merge_kernel, dim3(gk, nplanes), dim3(256), 0, ctx->stream, list, parent, out, total, H, W,
                       capacity, connectivity == 2 ? 1 : 0);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sp_compress_kernel, dim3(gk, nplanes), dim3(256), 0, ctx->stream, parent, flags, total, capacity);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_scan_excl_dev(ctx, flags, total_c, (size_t)capacity, nroots, nplanes));
    hipLaunchKernelGGL(sp_write_kernel, dim3(gk, nplanes), dim3(256), 0, ctx->stream, list, parent, flags, total, nroots,
                       out, count_dev, n, capacity);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
