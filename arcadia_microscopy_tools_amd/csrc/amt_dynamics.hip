// Cellpose post-processing: network output (flows dY, dX and cell probability) -> label image.
//
// The reference reaches this through cellpose.models.CellposeModel.eval (R/model.py:206-215, :270-290); cellpose is a
// third-party dependency that is absent offline (cellpose >= 4.0.8 in /root/reference/uv.lock), so this file restates
// the PUBLISHED algorithm (Stringer et al., Nature Methods 2021, "Cellpose"; cellpose/dynamics.py: compute_masks ->
// follow_flows -> get_masks) and parity with the package is UNPINNED: no vector of the reference covers it.  The
// CPU restatement the kernels are checked against is oracle/cellpose_dynamics.py.
//
//   1. cell pixels = cellprob > threshold; the flow field is dP * cell / 5.
//   2. every cell pixel whose |dY| exceeds 1e-3 follows the flow for `niter` Euler steps, the flow sampled bilinearly
//      the way torch.nn.functional.grid_sample(align_corners=False, zero padding) samples it after cellpose's
//      normalisation by (size - 1): sample coordinate s = u * size / (size - 1) - 0.5; positions stay in
//      [0, size - 1]; float32 throughout.  All other pixels keep their own position.
//   3. histogram of the (truncated) end positions on a grid padded by 20; seeds = bins that equal their 5 x 5 maximum
//      (scipy maximum_filter1d along both axes, mode 'reflect') and hold more than 10 pixels; seeds ordered by
//      decreasing count (ties: decreasing raster index); each seed grows for 5 rounds into its 3 x 3 neighbours whose
//      bins hold more than 2 pixels; overlapping regions go to the LATER seed.
//   4. a pixel takes the label of the region its end position lies in; labels covering more than max_size_fraction of
//      the image are dropped, labels smaller than min_size are dropped, the rest are renumbered 1..K in raster order of
//      first appearance.
// Not implemented here (stated in DESIGN.md): the flow-error quality filter (remove_bad_flow_masks, which needs the
// flows re-derived from the masks) and the per-mask hole filling of fill_holes_and_remove_small_masks.
#include "amt_internal.h"

constexpr int CP_RPAD = 20;

// ---- 1 + 2: follow the flow ------------------------------------------------------------------------------------
// pos[plane][y][x] = (hy << 16) | hx with hy = int(y_end) + CP_RPAD, hx = int(x_end) + CP_RPAD
__global__ void __launch_bounds__(256) cp_follow_kernel(const float* __restrict__ dP, const float* __restrict__ prob,
                                                        float thr, int niter, unsigned* __restrict__ pos, int H, int W) {
    const size_t n = (size_t)H * W;
    const float* dY = dP + (size_t)blockIdx.y * 2 * n;
    const float* dX = dY + n;
    const float* pr = prob + (size_t)blockIdx.y * n;
    unsigned* po = pos + (size_t)blockIdx.y * n;
    const float sy = (float)H / (float)(H - 1), sx = (float)W / (float)(W - 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y0 = (int)(i / W), x0 = (int)(i - (size_t)y0 * W);
        float y = (float)y0, x = (float)x0;
        const bool cell = pr[i] > thr;
        const bool moves = cell && fabsf(dY[i] / 5.0f) > 1e-3f;
        if (moves) {
            for (int t = 0; t < niter; ++t) {
                // grid_sample(align_corners=False) of the normalised position: s = u * size / (size - 1) - 0.5
                const float fy = y * sy - 0.5f, fx = x * sx - 0.5f;
                const float gy = floorf(fy), gx = floorf(fx);
                const int iy = (int)gy, ix = (int)gx;
                const float wy = fy - gy, wx = fx - gx;
                float vy = 0.0f, vx = 0.0f;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int yy = iy + a, xx = ix + b;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {  // zero padding outside
                            const size_t j = (size_t)yy * W + xx;
                            const bool cj = pr[j] > thr;  // dP * cell / 5
                            const float w = (a ? wy : 1.0f - wy) * (b ? wx : 1.0f - wx);
                            vy += w * ((cj ? dY[j] : 0.0f) / 5.0f);
                            vx += w * ((cj ? dX[j] : 0.0f) / 5.0f);
                        }
                    }
                y = fminf(fmaxf(y + vy, 0.0f), (float)(H - 1));
                x = fminf(fmaxf(x + vx, 0.0f), (float)(W - 1));
            }
        }
        po[i] = ((unsigned)((int)y + CP_RPAD) << 16) | (unsigned)((int)x + CP_RPAD);
    }
}

// ---- 3: histogram, seeds -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cp_hist_kernel(const unsigned* __restrict__ pos, int* __restrict__ hist, size_t n,
                                                      int Wp, size_t hn) {
    const unsigned* po = pos + (size_t)blockIdx.y * n;
    int* h = hist + (size_t)blockIdx.y * hn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned p = po[i];
        atomicAdd(&h[(size_t)(p >> 16) * Wp + (p & 0xFFFFu)], 1);
    }
}

__device__ __forceinline__ int cp_reflect(int i, int n) {  // scipy 'reflect': d c b a | a b c d | d c b a
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// seeds = (h == max over the 5 x 5 window with reflected borders) && h > 10; appended to seeds[plane][] as
// (count << 32) | flat index of the padded grid
__global__ void __launch_bounds__(256) cp_seeds_kernel(const int* __restrict__ hist, unsigned long long* __restrict__ seeds,
                                                       int* __restrict__ nseeds, int Hp, int Wp, int cap) {
    const size_t hn = (size_t)Hp * Wp;
    const int* h = hist + (size_t)blockIdx.y * hn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hn; i += (size_t)gridDim.x * 256) {
        const int v = h[i];
        if (v <= 10) continue;
        const int y = (int)(i / Wp), x = (int)(i - (size_t)y * Wp);
        int mx = 0;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx) {
                const int w = h[(size_t)cp_reflect(y + dy, Hp) * Wp + cp_reflect(x + dx, Wp)];
                mx = w > mx ? w : mx;
            }
        if (v == mx) {
            const int k = atomicAdd(&nseeds[blockIdx.y], 1);
            if (k < cap) seeds[(size_t)blockIdx.y * cap + k] = ((unsigned long long)(unsigned)v << 32) | (unsigned)i;
        }
    }
}

// order[plane][k] = the seed that is k-th by (count descending, flat index descending): rank by counting (few seeds)
__global__ void __launch_bounds__(256) cp_rank_kernel(const unsigned long long* __restrict__ seeds,
                                                      const int* __restrict__ nseeds, unsigned* __restrict__ order,
                                                      int cap) {
    const int ns = nseeds[blockIdx.y] < cap ? nseeds[blockIdx.y] : cap;
    const unsigned long long* s = seeds + (size_t)blockIdx.y * cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ns; i += gridDim.x * 256) {
        const unsigned long long me = s[i];
        int rank = 0;
        for (int j = 0; j < ns; ++j) rank += s[j] > me;  // the packed key orders by count, then by index
        order[(size_t)blockIdx.y * cap + rank] = (unsigned)(me & 0xFFFFFFFFull);
    }
}

// one wave per seed: five rounds of "3 x 3 neighbours with more than 2 pixels in their bin" on an 11 x 11 patch; the
// region's bins get label k + 1, a later seed overwrites an earlier one (atomicMax: labels grow with k)
__global__ void __launch_bounds__(64) cp_grow_kernel(const int* __restrict__ hist, const unsigned* __restrict__ order,
                                                     const int* __restrict__ nseeds, int* __restrict__ M, int Hp, int Wp,
                                                     int cap) {
    const int plane = blockIdx.y;
    const int ns = nseeds[plane] < cap ? nseeds[plane] : cap;
    const size_t hn = (size_t)Hp * Wp;
    const int* h = hist + (size_t)plane * hn;
    __shared__ unsigned char cur[121], good[121];
    for (int k = blockIdx.x; k < ns; k += gridDim.x) {
        const unsigned idx = order[(size_t)plane * cap + k];
        const int cy = (int)(idx / Wp), cx = (int)(idx - (size_t)cy * Wp);
        for (int t = threadIdx.x; t < 121; t += 64) {
            const int y = cy + t / 11 - 5, x = cx + t % 11 - 5;
            good[t] = (y >= 0 && y < Hp && x >= 0 && x < Wp && h[(size_t)y * Wp + x] > 2) ? 1 : 0;
            cur[t] = t == 60 ? 1 : 0;
        }
        __syncthreads();
        for (int it = 0; it < 5; ++it) {
            unsigned char nv[2] = {0, 0};
            for (int u = 0, t = threadIdx.x; t < 121; t += 64, ++u) {
                const int py = t / 11, px = t % 11;
                unsigned char any = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int qy = py + dy, qx = px + dx;
                        if (qy >= 0 && qy < 11 && qx >= 0 && qx < 11) any |= cur[qy * 11 + qx];
                    }
                nv[u] = any & good[t];
            }
            __syncthreads();
            for (int u = 0, t = threadIdx.x; t < 121; t += 64, ++u) cur[t] = nv[u];
            __syncthreads();
        }
        for (int t = threadIdx.x; t < 121; t += 64)
            if (cur[t]) atomicMax(&M[(size_t)plane * hn + (size_t)(cy + t / 11 - 5) * Wp + (cx + t % 11 - 5)], k + 1);
        __syncthreads();
    }
}

// ---- 4: labels, size filters, renumbering ----------------------------------------------------------------------------
// lab[i] = M[pos[i]]; count[label] += 1; first[label] = min(i)
__global__ void __launch_bounds__(256) cp_assign_kernel(const unsigned* __restrict__ pos, const int* __restrict__ M,
                                                        int* __restrict__ lab, int* __restrict__ count,
                                                        int* __restrict__ first, size_t n, int Wp, size_t hn, int cap) {
    const unsigned* po = pos + (size_t)blockIdx.y * n;
    const int* m = M + (size_t)blockIdx.y * hn;
    int* l = lab + (size_t)blockIdx.y * n;
    int* c = count + (size_t)blockIdx.y * (cap + 1);
    int* f = first + (size_t)blockIdx.y * (cap + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned p = po[i];
        const int v = m[(size_t)(p >> 16) * Wp + (p & 0xFFFFu)];
        l[i] = v;
        if (v > 0) {
            atomicAdd(&c[v], 1);
            atomicMin(&f[v], (int)i);
        }
    }
}

// keep[label] = min_size <= count <= big; map[label] = 1 + number of kept labels that appear earlier in raster order
__global__ void __launch_bounds__(256) cp_map_kernel(const int* __restrict__ count, const int* __restrict__ first,
                                                     const int* __restrict__ nseeds, int* __restrict__ map,
                                                     int* __restrict__ nout, int cap, int min_size, long long big) {
    const int plane = blockIdx.y;
    const int ns = nseeds[plane] < cap ? nseeds[plane] : cap;
    const int* c = count + (size_t)plane * (cap + 1);
    const int* f = first + (size_t)plane * (cap + 1);
    int* m = map + (size_t)plane * (cap + 1);
    auto kept = [&](int l) { return c[l] >= min_size && c[l] > 0 && (long long)c[l] <= big; };
    for (int l = 1 + blockIdx.x * 256 + threadIdx.x; l <= ns; l += gridDim.x * 256) {
        int v = 0;
        if (kept(l)) {
            v = 1;
            for (int j = 1; j <= ns; ++j) v += (j != l && kept(j) && f[j] < f[l]) ? 1 : 0;
        }
        m[l] = v;
        if (v && nseeds[plane] <= cap) atomicMax(&nout[plane], v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        m[0] = 0;
        if (nseeds[plane] > cap) nout[plane] = -1;  // more seeds than max_seeds: the caller must not trust the labels
    }
}

__global__ void __launch_bounds__(256) cp_apply_kernel(int* __restrict__ lab, const int* __restrict__ map, size_t n,
                                                       int cap) {
    int* l = lab + (size_t)blockIdx.y * n;
    const int* m = map + (size_t)blockIdx.y * (cap + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) l[i] = m[l[i]];
}

__global__ void cp_fill_kernel(int* p, size_t n, int v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

extern "C" int amt_cellpose_masks(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out,
                                  int32_t* count_dev, int nplanes, int H, int W, float cellprob_threshold, int niter,
                                  int min_size, float max_size_fraction, int max_seeds) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(dP && cellprob && labels_out && count_dev && nplanes >= 0, "cellpose_masks: bad arguments");
    AMT_REQUIRE(H >= 2 && W >= 2 && H + 2 * CP_RPAD < 65536 && W + 2 * CP_RPAD < 65536,
                "cellpose_masks: image of %d x %d is outside 2..65495 per side", H, W);
    AMT_REQUIRE(niter >= 0 && max_seeds >= 1 && min_size >= 0, "cellpose_masks: bad niter / max_seeds / min_size");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int Hp = H + 2 * CP_RPAD, Wp = W + 2 * CP_RPAD;
    const size_t hn = (size_t)Hp * Wp;
    const int cap = max_seeds;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 4) + 2 * amt_align((size_t)nplanes * hn * 4) +
                                     amt_align((size_t)nplanes * cap * 8) + amt_align((size_t)nplanes * cap * 4) +
                                     3 * amt_align((size_t)nplanes * (cap + 1) * 4) + amt_align((size_t)nplanes * 4)));
    unsigned* pos = arena_take_t<unsigned>(ctx, (size_t)nplanes * n);
    int* hist = arena_take_t<int>(ctx, (size_t)nplanes * hn);
    int* M = arena_take_t<int>(ctx, (size_t)nplanes * hn);
    unsigned long long* seeds = arena_take_t<unsigned long long>(ctx, (size_t)nplanes * cap);
    unsigned* order = arena_take_t<unsigned>(ctx, (size_t)nplanes * cap);
    int* count = arena_take_t<int>(ctx, (size_t)nplanes * (cap + 1));
    int* first = arena_take_t<int>(ctx, (size_t)nplanes * (cap + 1));
    int* map = arena_take_t<int>(ctx, (size_t)nplanes * (cap + 1));
    int* nseeds = arena_take_t<int>(ctx, nplanes);
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * hn * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(M, 0, (size_t)nplanes * hn * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)nplanes * (cap + 1) * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(nseeds, 0, (size_t)nplanes * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(count_dev, 0, (size_t)nplanes * 4, ctx->stream));
    hipLaunchKernelGGL(cp_fill_kernel, dim3(amt_grid_for((size_t)nplanes * (cap + 1), 256, 1024)), dim3(256), 0, ctx->stream,
                       first, (size_t)nplanes * (cap + 1), 0x7fffffff);
    AMT_LAUNCH_CHECK();
    dim3 gpx(amt_grid_for(n, 256, 4096), nplanes), ghist(amt_grid_for(hn, 256, 4096), nplanes);
    hipLaunchKernelGGL(cp_follow_kernel, gpx, dim3(256), 0, ctx->stream, dP, cellprob, cellprob_threshold, niter, pos, H, W);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_hist_kernel, gpx, dim3(256), 0, ctx->stream, pos, hist, n, Wp, hn);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_seeds_kernel, ghist, dim3(256), 0, ctx->stream, hist, seeds, nseeds, Hp, Wp, cap);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_rank_kernel, dim3(amt_grid_for((size_t)cap, 256, 64), nplanes), dim3(256), 0, ctx->stream, seeds,
                       nseeds, order, cap);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_grow_kernel, dim3(512, nplanes), dim3(64), 0, ctx->stream, hist, order, nseeds, M, Hp, Wp, cap);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_assign_kernel, gpx, dim3(256), 0, ctx->stream, pos, M, labels_out, count, first, n, Wp, hn, cap);
    AMT_LAUNCH_CHECK();
    const long long big = (long long)((double)n * (double)max_size_fraction);
    hipLaunchKernelGGL(cp_map_kernel, dim3(amt_grid_for((size_t)cap, 256, 64), nplanes), dim3(256), 0, ctx->stream, count,
                       first, nseeds, map, count_dev, cap, min_size, big);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_apply_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, map, n, cap);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
