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
//   5. (amt_cellpose_masks_ex, flow_threshold > 0) the flow-error quality filter, remove_bad_flow_masks ->
//      metrics.flow_error -> dynamics.masks_to_flows: every mask diffuses heat from its centre (the mask pixel nearest
//      to the mean position) for 2 * max(height + width + 2) Jacobi steps of "mean over the 3 x 3 neighbours of the
//      same mask" in float64, the flow is the normalised central difference of that field, and a mask whose mean
//      squared difference to the network's flow / 5 exceeds the threshold is dropped.
//   6. (amt_cellpose_masks_ex) utils.fill_holes_and_remove_small_masks: labels ascending, masks below min_size dropped,
//      the others hole-filled inside their bounding box (scipy binary_fill_holes: background not 4-connected to the
//      box's border) and renumbered 1..K.
#include "amt_internal.h"
#include <hip/hip_ext.h>

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

// The same in two kernels (round 3): cp_follow_kernel keeps a whole wave in its 200-step loop for the one pixel in four
// that moves, and every bilinear tap costs three gathers, a select and two divisions.  cp_follow_prep_kernel computes the
// field the taps really read -- (cell ? dP : 0) / 5 as one float2 per pixel, the same float32 operations once instead of
// 800 times -- writes the resting pixels' positions and LISTS the moving pixels; cp_follow_list_kernel then follows the
// listed pixels, full waves, one 8-byte gather per tap.  Same arithmetic in the same order: bit-identical positions.
__global__ void __launch_bounds__(256) cp_follow_prep_kernel(const float* __restrict__ dP, const float* __restrict__ prob,
                                                             float thr, float2* __restrict__ fld, unsigned* __restrict__ pos,
                                                             int* __restrict__ list, int* __restrict__ nlist, int H, int W) {
    const size_t n = (size_t)H * W;
    const float* dY = dP + (size_t)blockIdx.y * 2 * n;
    const float* dX = dY + n;
    const float* pr = prob + (size_t)blockIdx.y * n;
    float2* f = fld + (size_t)blockIdx.y * n;
    unsigned* po = pos + (size_t)blockIdx.y * n;
    int* lst = list + (size_t)blockIdx.y * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int s_wtot[4], s_base;
    // a workgroup takes 1,024 pixels per step (four per thread) and reserves their list slots with ONE atomic: a returning
    // atomic per wave queued thousands deep on the plane's one counter (the pass took 1.2 ms for 200 MB of traffic)
    for (size_t i0 = (size_t)blockIdx.x * 1024; i0 < n; i0 += (size_t)gridDim.x * 1024) {  // block-uniform bounds
        bool mv[4];
        int before = 0, wtot = 0;  // movers of this wave in earlier slots / in all four
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + (size_t)u * 256 + threadIdx.x;
            mv[u] = false;
            if (i < n) {
                const bool cell = pr[i] > thr;
                const float vy = (cell ? dY[i] : 0.0f) / 5.0f, vx = (cell ? dX[i] : 0.0f) / 5.0f;
                f[i] = make_float2(vy, vx);
                mv[u] = cell && fabsf(dY[i] / 5.0f) > 1e-3f;
                if (!mv[u]) {
                    const int y0 = (int)(i / W), x0 = (int)(i - (size_t)y0 * W);
                    po[i] = ((unsigned)(y0 + CP_RPAD) << 16) | (unsigned)(x0 + CP_RPAD);
                }
            }
        }
        int slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long m = __ballot(mv[u]);
            slot[u] = wtot + __popcll(m & ((1ull << lane) - 1ull));
            wtot += __popcll(m);
        }
        (void)before;
        if (lane == 0) s_wtot[wave] = wtot;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
            s_base = tot ? atomicAdd(&nlist[blockIdx.y], tot) : 0;
        }
        __syncthreads();
        int base = s_base;
        for (int w2 = 0; w2 < wave; ++w2) base += s_wtot[w2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mv[u]) lst[base + slot[u]] = (int)(i0 + (size_t)u * 256 + threadIdx.x);
        __syncthreads();  // s_wtot / s_base are rewritten by the next step
    }
}

__global__ void __launch_bounds__(256) cp_follow_list_kernel(const float2* __restrict__ fld, int niter,
                                                             unsigned* __restrict__ pos, const int* __restrict__ list,
                                                             const int* __restrict__ nlist, int H, int W) {
    const size_t n = (size_t)H * W;
    const float2* f = fld + (size_t)blockIdx.y * n;
    unsigned* po = pos + (size_t)blockIdx.y * n;
    const int* lst = list + (size_t)blockIdx.y * n;
    const int cnt = nlist[blockIdx.y];
    const float sy = (float)H / (float)(H - 1), sx = (float)W / (float)(W - 1);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int i = lst[k];
        const int y0 = i / W, x0 = i - y0 * W;
        float y = (float)y0, x = (float)x0;
        for (int t = 0; t < niter; ++t) {
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
                        const float2 v = f[(size_t)yy * W + xx];
                        const float w = (a ? wy : 1.0f - wy) * (b ? wx : 1.0f - wx);
                        vy += w * v.x;
                        vx += w * v.y;
                    }
                }
            y = fminf(fmaxf(y + vy, 0.0f), (float)(H - 1));
            x = fminf(fmaxf(x + vx, 0.0f), (float)(W - 1));
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
    const int lane = threadIdx.x & 63;
    for (size_t i0 = (size_t)blockIdx.x * 256; i0 < n; i0 += (size_t)gridDim.x * 256) {  // block-uniform: whole waves below
        const size_t i = i0 + threadIdx.x;
        int v = 0;
        if (i < n) {
            const unsigned p = po[i];
            v = m[(size_t)(p >> 16) * Wp + (p & 0xFFFFu)];
            l[i] = v;
        }
        // neighbouring pixels mostly share their label: ONE pair of atomics per run of equal labels inside the wave (the
        // run's first lane adds the run's length; its index is the run's smallest) instead of one pair per pixel
        const int vl = amt_lane_left(v);
        const bool head = v > 0 && (lane == 0 || vl != v);
        const unsigned long long bounds = __ballot(head || v <= 0);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : bounds & ~((2ull << lane) - 1ull);
            const int end = later ? __ffsll((long long)later) - 1 : 64;
            atomicAdd(&c[v], end - lane);
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


// ---- 5: flow-error filter ------------------------------------------------------------------------------------------
// per label: bounding box, pixel count, sums of the row / column coordinates (exact integers).  One atomic set per RUN
// of equal labels in a wave's 64 pixels, not per pixel.
struct cp_box {
    int y0, y1, x0, x1;
};

__global__ void __launch_bounds__(256) cp_boxinit_kernel(cp_box* __restrict__ box, int* __restrict__ cnt,
                                                         unsigned long long* __restrict__ sums, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        cp_box b;
        b.y0 = b.x0 = 0x7fffffff;
        b.y1 = b.x1 = -1;
        box[i] = b;
        cnt[i] = 0;
        sums[2 * i] = sums[2 * i + 1] = 0ull;
    }
}

__global__ void __launch_bounds__(256) cp_boxes_kernel(const int* __restrict__ lab, cp_box* __restrict__ box,
                                                       int* __restrict__ cnt, unsigned long long* __restrict__ sums, int H,
                                                       int W, int cap) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = blockIdx.x * 64 + lane;
    if (y >= H) return;
    const size_t n = (size_t)H * W;
    int v = x < W ? lab[(size_t)blockIdx.z * n + (size_t)y * W + x] : 0;
    if (v > cap) v = 0;  // outside the caller's label range: ignored
    const int left = amt_lane_left(v);
    const bool head = v > 0 && (lane == 0 || left != v);
    const unsigned long long bounds = __ballot(lane == 0 || left != v);  // starts of runs of equal values
    if (head) {
        const unsigned long long later = bounds & ~((2ull << lane) - 1ull);
        const int end_lane = later ? (__ffsll((long long)later) - 2) : 63;
        const int len = end_lane - lane + 1;
        const size_t k = (size_t)blockIdx.z * (cap + 1) + v;
        atomicMin(&box[k].y0, y);
        atomicMax(&box[k].y1, y);
        atomicMin(&box[k].x0, x);
        atomicMax(&box[k].x1, x + len - 1);
        atomicAdd(&cnt[k], len);
        atomicAdd(&sums[2 * k], (unsigned long long)y * len);
        atomicAdd(&sums[2 * k + 1], (unsigned long long)len * x + (unsigned long long)len * (len - 1) / 2);
    }
}

// one wave per label: the mask pixel nearest to the mean position (first in raster order of the box among equals),
// ext = height + width + 2; nit[plane] = 2 * max ext
__global__ void __launch_bounds__(64) cp_centers_kernel(const int* __restrict__ lab, const cp_box* __restrict__ box,
                                                        const int* __restrict__ cnt,
                                                        const unsigned long long* __restrict__ sums,
                                                        const int* __restrict__ nlab, int* __restrict__ center,
                                                        int* __restrict__ nit, int H, int W, int cap) {
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    const size_t n = (size_t)H * W;
    const int* L = lab + (size_t)plane * n;
    const int lane = threadIdx.x;
    for (int l = 1 + blockIdx.x; l <= K; l += gridDim.x) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        const cp_box b = box[k];
        const int c = cnt[k];
        if (c <= 0) {
            if (lane == 0) center[2 * k] = center[2 * k + 1] = -1;
            continue;
        }
        const int h = b.y1 - b.y0 + 1, w = b.x1 - b.x0 + 1;
        // means of the box-relative coordinates, as numpy takes them (integer sums are exact in float64)
        const double ym = (double)(long long)(sums[2 * k] - (unsigned long long)c * b.y0) / (double)c;
        const double xm = (double)(long long)(sums[2 * k + 1] - (unsigned long long)c * b.x0) / (double)c;
        double best = 1e300;
        int besti = 0x7fffffff;
        for (int i = lane; i < h * w; i += 64) {
            const int ry = i / w, rx = i - ry * w;
            if (L[(size_t)(b.y0 + ry) * W + b.x0 + rx] == l) {
                const double ex = (double)rx - xm, ey = (double)ry - ym;
                const double d = ex * ex + ey * ey;
                if (d < best) {  // ascending i per lane: the first minimum stays
                    best = d;
                    besti = i;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(besti, o);
            if (ob < best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
        if (lane == 0) {
            center[2 * k] = b.y0 + besti / w;
            center[2 * k + 1] = b.x0 + besti % w;
            atomicMax(&nit[plane], 2 * (h + w + 2));
        }
    }
}

// Heat diffusion of ONE label per workgroup.  A label reads and writes only its own pixels (neighbours of another
// label count as 0), so all labels share the two float64 planes of the image.  LDS variant: the box plus a one-pixel
// ring lives in LDS (two float64 buffers + the 9-bit "same label" masks); labels whose tile exceeds TILE_MAX are left
// to the next class, the last class (TILE_MAX == 0) iterates in the global planes themselves.
// Tout[plane] receives the final field at the label's pixels.
template <int TILE_MAX, int TILE_MIN>
__global__ void __launch_bounds__(256) cp_diffuse_kernel(const int* __restrict__ lab, const cp_box* __restrict__ box,
                                                         const int* __restrict__ cnt, const int* __restrict__ center,
                                                         const int* __restrict__ nlab, const int* __restrict__ nit,
                                                         double* __restrict__ TA, double* __restrict__ TB, int H, int W,
                                                         int cap) {
    extern __shared__ __attribute__((aligned(16))) char cp_smem[];
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    const size_t n = (size_t)H * W;
    const int* L = lab + (size_t)plane * n;
    const int iters = nit[plane];
    double* ga = TA + (size_t)plane * n;
    double* gb = TB + (size_t)plane * n;
    for (int l = 1 + blockIdx.x; l <= K; l += gridDim.x) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        if (cnt[k] <= 0) continue;
        const cp_box b = box[k];
        const int th = b.y1 - b.y0 + 3, tw = b.x1 - b.x0 + 3;  // with the ring
        const int npx = th * tw;
        if (npx <= TILE_MIN || (TILE_MAX > 0 && npx > TILE_MAX)) continue;  // another class takes it
        const int cy = center[2 * k], cx = center[2 * k + 1];
        if (TILE_MAX > 0) {
            double* A = reinterpret_cast<double*>(cp_smem);
            double* B = A + TILE_MAX;
            unsigned short* same = reinterpret_cast<unsigned short*>(B + TILE_MAX);
            for (int i = threadIdx.x; i < npx; i += 256) {
                const int ty = i / tw, tx = i - ty * tw;
                const int y = b.y0 + ty - 1, x = b.x0 + tx - 1;
                unsigned m = 0;
                if (ty >= 1 && ty < th - 1 && tx >= 1 && tx < tw - 1 && L[(size_t)y * W + x] == l) {
                    // bit j: neighbour j carries the same label; order (0,0) (-1,0) (1,0) (0,-1) (0,1) (-1,-1) (-1,1) (1,-1) (1,1)
                    const int dy[9] = {0, -1, 1, 0, 0, -1, -1, 1, 1}, dx[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1};
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int yy = y + dy[j], xx = x + dx[j];
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W && L[(size_t)yy * W + xx] == l) m |= 1u << j;
                    }
                }
                same[i] = (unsigned short)m;
                A[i] = 0.0;
                B[i] = 0.0;
            }
            __syncthreads();
            const int ci = (cy - b.y0 + 1) * tw + (cx - b.x0 + 1);
            double* cur = A;
            double* nxt = B;
            for (int it = 0; it < iters; ++it) {
                if (threadIdx.x == 0) cur[ci] += 1.0;
                __syncthreads();
                for (int i = threadIdx.x; i < npx; i += 256) {
                    const unsigned m = same[i];
                    if (m) {
                        double acc = 0.0;
                        acc = acc + cur[i];
                        acc = acc + ((m >> 1) & 1 ? cur[i - tw] : 0.0);
                        acc = acc + ((m >> 2) & 1 ? cur[i + tw] : 0.0);
                        acc = acc + ((m >> 3) & 1 ? cur[i - 1] : 0.0);
                        acc = acc + ((m >> 4) & 1 ? cur[i + 1] : 0.0);
                        acc = acc + ((m >> 5) & 1 ? cur[i - tw - 1] : 0.0);
                        acc = acc + ((m >> 6) & 1 ? cur[i - tw + 1] : 0.0);
                        acc = acc + ((m >> 7) & 1 ? cur[i + tw - 1] : 0.0);
                        acc = acc + ((m >> 8) & 1 ? cur[i + tw + 1] : 0.0);
                        nxt[i] = acc / 9.0;
                    }
                }
                __syncthreads();
                double* t = cur;
                cur = nxt;
                nxt = t;
            }
            for (int i = threadIdx.x; i < npx; i += 256)
                if (same[i]) {
                    const int ty = i / tw, tx = i - ty * tw;
                    ga[(size_t)(b.y0 + ty - 1) * W + (b.x0 + tx - 1)] = cur[i];
                }
            __syncthreads();
        } else {
            // global planes: ga / gb are zero at this label's pixels on entry
            double* cur = ga;
            double* nxt = gb;
            const int bh = th - 2, bw = tw - 2;
            for (int it = 0; it < iters; ++it) {
                if (threadIdx.x == 0) cur[(size_t)cy * W + cx] += 1.0;
                __syncthreads();
                for (int i = threadIdx.x; i < bh * bw; i += 256) {
                    const int y = b.y0 + i / bw, x = b.x0 + i % bw;
                    const size_t p = (size_t)y * W + x;
                    if (L[p] != l) continue;
                    const int dy[9] = {0, -1, 1, 0, 0, -1, -1, 1, 1}, dx[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1};
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int yy = y + dy[j], xx = x + dx[j];
                        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W && L[(size_t)yy * W + xx] == l;
                        acc = acc + (ok ? cur[(size_t)yy * W + xx] : 0.0);
                    }
                    nxt[p] = acc / 9.0;
                }
                __threadfence_block();
                __syncthreads();
                double* t = cur;
                cur = nxt;
                nxt = t;
            }
            if (cur != ga) {  // odd number of steps: the result sits in gb
                for (int i = threadIdx.x; i < bh * bw; i += 256) {
                    const size_t p = (size_t)(b.y0 + i / bw) * W + (b.x0 + i % bw);
                    if (L[p] == l) ga[p] = gb[p];
                }
            }
            __syncthreads();
        }
    }
}

// one wave per label: err = mean((mu_y - dY / 5)^2) + mean((mu_x - dX / 5)^2) over the label's pixels, with
// mu = (T[y+1] - T[y-1], T[x+1] - T[x-1]) / (1e-60 + norm); neighbouring values are taken as they are (other labels'
// fields included, 0 outside the image and on background)
__global__ void __launch_bounds__(64) cp_flow_error_kernel(const int* __restrict__ lab, const cp_box* __restrict__ box,
                                                           const int* __restrict__ cnt, const int* __restrict__ nlab,
                                                           const double* __restrict__ T, const float* __restrict__ dP,
                                                           double* __restrict__ err, int H, int W, int cap) {
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    const size_t n = (size_t)H * W;
    const int* L = lab + (size_t)plane * n;
    const double* t = T + (size_t)plane * n;
    const float* dY = dP + (size_t)plane * 2 * n;
    const float* dX = dY + n;
    const int lane = threadIdx.x;
    for (int l = 1 + blockIdx.x; l <= K; l += gridDim.x) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        const int c = cnt[k];
        if (c <= 0) {
            if (lane == 0) err[k] = 0.0;
            continue;
        }
        const cp_box b = box[k];
        const int h = b.y1 - b.y0 + 1, w = b.x1 - b.x0 + 1;
        double sy = 0.0, sx = 0.0;
        for (int i = lane; i < h * w; i += 64) {
            const int y = b.y0 + i / w, x = b.x0 + i % w;
            const size_t p = (size_t)y * W + x;
            if (L[p] != l) continue;
            // background pixels of the field plane are never written and stay 0
            const double up = y > 0 ? t[p - W] : 0.0, dn = y < H - 1 ? t[p + W] : 0.0;
            const double lf = x > 0 ? t[p - 1] : 0.0, rt = x < W - 1 ? t[p + 1] : 0.0;
            const double gy = dn - up, gx = rt - lf;
            const double nrm = 1e-60 + sqrt(gy * gy + gx * gx);
            const double ey = gy / nrm - (double)(dY[p] / 5.0f), ex = gx / nrm - (double)(dX[p] / 5.0f);
            sy += ey * ey;
            sx += ex * ex;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sy += __shfl_xor(sy, o);
            sx += __shfl_xor(sx, o);
        }
        if (lane == 0) err[k] = sy / (double)c + sx / (double)c;
    }
}

// keep[l] = 1 for labels that survive the flow filter and the size floor (ascending scan -> new numbers)
__global__ void __launch_bounds__(256) cp_keep_kernel(const int* __restrict__ cnt, const double* __restrict__ err,
                                                      const int* __restrict__ nlab, int* __restrict__ keep, int cap,
                                                      int min_size, double flow_threshold) {
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    for (int l = blockIdx.x * 256 + threadIdx.x; l <= cap; l += gridDim.x * 256) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        int v = 0;
        if (l >= 1 && l <= K && cnt[k] > 0 && cnt[k] >= min_size) v = (flow_threshold > 0.0 && err[k] > flow_threshold) ? 0 : 1;
        keep[k] = v;
    }
}

// map[l] = keep ? (kept labels below l) + 1 : 0 from the exclusive scan in `scan`; nout = number kept (or -1 kept as is)
__global__ void __launch_bounds__(256) cp_newmap_kernel(const int* __restrict__ keep, const int* __restrict__ scan,
                                                        const int* __restrict__ total, int* __restrict__ map,
                                                        int* __restrict__ nout, int cap) {
    const int plane = blockIdx.y;
    for (int l = blockIdx.x * 256 + threadIdx.x; l <= cap; l += gridDim.x * 256) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        map[k] = keep[k] ? scan[k] + 1 : 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nout[plane] >= 0) nout[plane] = total[plane];
}

// ---- 6: hole filling ----------------------------------------------------------------------------------------------
// One workgroup per kept label: `outside` = background of the label's box that is 4-connected to the box's border
// (scipy.ndimage.binary_fill_holes on the cropped mask); every other non-member pixel of the box is a hole pixel and
// receives the label's NEW number in `fill` (atomicMax: where two labels' holes overlap the later label wins, as in
// the package's sequential loop).  A hole pixel that carries another surviving label means that the sequential
// result depends on the order of the labels: nested[plane] is set and cp_holes_seq_kernel redoes the plane.
// The box's byte map lives in LDS when it fits, otherwise in a byte plane of the image (boxes may overlap there,
// so the global variant is only used one label at a time per plane: by the sequential kernel).
__device__ __forceinline__ int cp_fill_box(unsigned char* st, int h, int w, int* changed_flag) {
    // st: 1 = member, 0 = unknown background; marks outside background with 2.  Returns nothing useful; loops until
    // no pixel changes.  All 256 threads of the workgroup call it.
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const int y = i / w, x = i - y * w;
        if (st[i] == 0 && (y == 0 || y == h - 1 || x == 0 || x == w - 1)) st[i] = 2;
    }
    __syncthreads();
    while (true) {
        if (threadIdx.x == 0) *changed_flag = 0;
        __syncthreads();
        int ch = 0;
        // each thread sweeps whole rows forwards and backwards, then whole columns: far propagation per pass
        for (int y = threadIdx.x; y < h; y += 256) {
            unsigned char* r = st + (size_t)y * w;
            for (int x = 1; x < w; ++x)
                if (r[x] == 0 && r[x - 1] == 2) { r[x] = 2; ch = 1; }
            for (int x = w - 2; x >= 0; --x)
                if (r[x] == 0 && r[x + 1] == 2) { r[x] = 2; ch = 1; }
        }
        __syncthreads();
        for (int x = threadIdx.x; x < w; x += 256) {
            for (int y = 1; y < h; ++y)
                if (st[(size_t)y * w + x] == 0 && st[(size_t)(y - 1) * w + x] == 2) { st[(size_t)y * w + x] = 2; ch = 1; }
            for (int y = h - 2; y >= 0; --y)
                if (st[(size_t)y * w + x] == 0 && st[(size_t)(y + 1) * w + x] == 2) { st[(size_t)y * w + x] = 2; ch = 1; }
        }
        if (ch) *changed_flag = 1;
        __syncthreads();
        const int any = *changed_flag;
        __syncthreads();
        if (!any) break;
    }
    return 0;
}

constexpr int CP_HOLE_TILE = 48 * 1024;  // bytes of LDS for the box map of the parallel kernel

__global__ void __launch_bounds__(256) cp_holes_kernel(const int* __restrict__ lab, const cp_box* __restrict__ box,
                                                       const int* __restrict__ keep, const int* __restrict__ map,
                                                       const int* __restrict__ nlab, int* __restrict__ fill,
                                                       int* __restrict__ nested, int H, int W, int cap) {
    __shared__ unsigned char st[CP_HOLE_TILE];
    __shared__ int changed;
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    const size_t n = (size_t)H * W;
    const int* L = lab + (size_t)plane * n;
    const int* kp = keep + (size_t)plane * (cap + 1);
    for (int l = 1 + blockIdx.x; l <= K; l += gridDim.x) {
        if (!kp[l]) continue;
        const cp_box b = box[(size_t)plane * (cap + 1) + l];
        const int h = b.y1 - b.y0 + 1, w = b.x1 - b.x0 + 1;
        if (h < 3 || w < 3) continue;  // no interior: nothing can be enclosed
        if ((long long)h * w > CP_HOLE_TILE) {  // too large for the LDS map: the sequential kernel handles the plane
            if (threadIdx.x == 0) nested[plane] = 1;
            continue;
        }
        for (int i = threadIdx.x; i < h * w; i += 256) st[i] = L[(size_t)(b.y0 + i / w) * W + b.x0 + i % w] == l ? 1 : 0;
        __syncthreads();
        cp_fill_box(st, h, w, &changed);
        const int nl = map[(size_t)plane * (cap + 1) + l];
        for (int i = threadIdx.x; i < h * w; i += 256) {
            if (st[i] == 0) {  // enclosed
                const size_t p = (size_t)(b.y0 + i / w) * W + b.x0 + i % w;
                const int other = L[p];
                if (other > 0 && other <= K && kp[other]) nested[plane] = 1;
                atomicMax(&fill[(size_t)plane * n + p], nl);
            }
        }
        __syncthreads();
    }
}

// labels_out = fill ? fill : map[lab]
__global__ void __launch_bounds__(256) cp_apply_fill_kernel(int* __restrict__ lab, const int* __restrict__ map,
                                                            const int* __restrict__ fill, size_t n, int cap) {
    int* l = lab + (size_t)blockIdx.y * n;
    const int* m = map + (size_t)blockIdx.y * (cap + 1);
    const int* f = fill ? fill + (size_t)blockIdx.y * n : nullptr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int fv = f ? f[i] : 0;
        const int v = l[i];
        l[i] = fv ? fv : ((unsigned)v <= (unsigned)cap ? m[v] : 0);
    }
}

// The package's loop, label by label, for planes whose holes contain other masks (or whose boxes exceed the LDS map):
// ONE workgroup per flagged plane.  `work` = a copy of the labels after the flow filter (get_masks numbering, dropped
// labels already 0); the result is written into it in place; `bytes` = an H x W byte plane per image for box maps.
__global__ void __launch_bounds__(256) cp_holes_seq_kernel(int* __restrict__ work, const cp_box* __restrict__ box,
                                                           const int* __restrict__ nlab, const int* __restrict__ nested,
                                                           unsigned char* __restrict__ bytes, int* __restrict__ nout, int H,
                                                           int W, int cap, int min_size) {
    __shared__ int changed, npix_s;
    const int plane = blockIdx.x;
    if (!nested[plane] || nout[plane] < 0) return;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    const size_t n = (size_t)H * W;
    int* L = work + (size_t)plane * n;
    unsigned char* st = bytes + (size_t)plane * n;
    int j = 0;
    for (int l = 1; l <= K; ++l) {
        const cp_box b = box[(size_t)plane * (cap + 1) + l];
        if (b.y1 < b.y0) continue;
        const int h = b.y1 - b.y0 + 1, w = b.x1 - b.x0 + 1;
        if (threadIdx.x == 0) npix_s = 0;
        __syncthreads();
        int mine = 0;
        for (int i = threadIdx.x; i < h * w; i += 256) {
            const int v = L[(size_t)(b.y0 + i / w) * W + b.x0 + i % w] == l ? 1 : 0;
            st[i] = (unsigned char)v;
            mine += v;
        }
        if (mine) atomicAdd(&npix_s, mine);
        __threadfence_block();
        __syncthreads();
        const int npix = npix_s;
        __syncthreads();
        if (npix == 0) continue;
        if (min_size > 0 && npix < min_size) {
            for (int i = threadIdx.x; i < h * w; i += 256)
                if (st[i] == 1) L[(size_t)(b.y0 + i / w) * W + b.x0 + i % w] = 0;
        } else {
            cp_fill_box(st, h, w, &changed);
            for (int i = threadIdx.x; i < h * w; i += 256)
                if (st[i] != 2) L[(size_t)(b.y0 + i / w) * W + b.x0 + i % w] = j + 1;
            ++j;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) nout[plane] = j;
}

// copy the sequential kernel's planes over the parallel result; drop = labels with map 0 become 0 in `work` beforehand
__global__ void __launch_bounds__(256) cp_seq_prepare_kernel(const int* __restrict__ lab, const int* __restrict__ keep_flow,
                                                             const int* __restrict__ nested, int* __restrict__ work,
                                                             size_t n, int cap) {
    if (!nested[blockIdx.y]) return;
    const int* l = lab + (size_t)blockIdx.y * n;
    int* o = work + (size_t)blockIdx.y * n;
    const int* kf = keep_flow + (size_t)blockIdx.y * (cap + 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = l[i];
        o[i] = (v > 0 && kf[v]) ? v : 0;
    }
}

__global__ void __launch_bounds__(256) cp_seq_commit_kernel(int* __restrict__ out, const int* __restrict__ work,
                                                            const int* __restrict__ nested, size_t n) {
    if (!nested[blockIdx.y]) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[(size_t)blockIdx.y * n + i] = work[(size_t)blockIdx.y * n + i];
}

__global__ void cp_seq_count_kernel(int* __restrict__ count_dev, const int* __restrict__ total,
                                    const int* __restrict__ nested, int nplanes) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nplanes && nested[p] && count_dev[p] >= 0) count_dev[p] = total[p];
}

// flow-filter-only keep flags (size floor not applied): what the sequential kernel starts from
__global__ void __launch_bounds__(256) cp_keepflow_kernel(const int* __restrict__ cnt, const double* __restrict__ err,
                                                          const int* __restrict__ nlab, int* __restrict__ keep, int cap,
                                                          double flow_threshold) {
    const int plane = blockIdx.y;
    const int K = nlab[plane] < 0 ? 0 : (nlab[plane] < cap ? nlab[plane] : cap);
    for (int l = blockIdx.x * 256 + threadIdx.x; l <= cap; l += gridDim.x * 256) {
        const size_t k = (size_t)plane * (cap + 1) + l;
        keep[k] = (l >= 1 && l <= K && cnt[k] > 0 && !(flow_threshold > 0.0 && err[k] > flow_threshold)) ? 1 : 0;
    }
}

// boxes -> centres -> diffusion (three size classes side by side) -> per-label flow error.  labels carry 1..nlab[plane].
static int cp_flow_errors(amt_ctx* ctx, const int* labels, const float* dP, const int* nlab, cp_box* box, int* cnt,
                          unsigned long long* sums, int* center, int* nit, double* TA, double* TB, double* err, int nplanes,
                          int H, int W, int cap, bool want_error) {
    const size_t n = (size_t)H * W;
    const size_t nl = (size_t)nplanes * (cap + 1);
    hipLaunchKernelGGL(cp_boxinit_kernel, dim3(amt_grid_for(nl, 256, 1024)), dim3(256), 0, ctx->stream, box, cnt, sums, nl);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_boxes_kernel, dim3((W + 63) / 64, (H + 3) / 4, nplanes), dim3(256), 0, ctx->stream, labels, box, cnt,
                       sums, H, W, cap);
    AMT_LAUNCH_CHECK();
    if (!want_error) return AMT_OK;
    AMT_HIP_CHECK(hipMemsetAsync(nit, 0, (size_t)nplanes * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(TA, 0, (size_t)nplanes * n * 8, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(TB, 0, (size_t)nplanes * n * 8, ctx->stream));
    const int glab = cap < 2048 ? cap : 2048;
    hipLaunchKernelGGL(cp_centers_kernel, dim3(glab, nplanes), dim3(64), 0, ctx->stream, labels, box, cnt, sums, nlab, center,
                       nit, H, W, cap);
    AMT_LAUNCH_CHECK();
    // the three classes are independent (a label belongs to exactly one): the second and third launch carry no
    // barrier bit and overlap the first
    {
        void* args[] = {(void*)&labels, (void*)&box, (void*)&cnt, (void*)&center, (void*)&nlab, (void*)&nit, (void*)&TA,
                        (void*)&TB, (void*)&H, (void*)&W, (void*)&cap};
        AMT_HIP_CHECK(hipExtLaunchKernel((const void*)cp_diffuse_kernel<2304, 0>, dim3(glab, nplanes), dim3(256), args,
                                         (size_t)2304 * 18, ctx->stream, nullptr, nullptr, 0));
        AMT_HIP_CHECK(hipExtLaunchKernel((const void*)cp_diffuse_kernel<6400, 2304>, dim3(glab < 512 ? glab : 512, nplanes),
                                         dim3(256), args, (size_t)6400 * 18, ctx->stream, nullptr, nullptr,
                                         (int)hipExtAnyOrderLaunch));
        AMT_HIP_CHECK(hipExtLaunchKernel((const void*)cp_diffuse_kernel<0, 6400>, dim3(glab < 256 ? glab : 256, nplanes),
                                         dim3(256), args, 0, ctx->stream, nullptr, nullptr, (int)hipExtAnyOrderLaunch));
    }
    hipLaunchKernelGGL(cp_flow_error_kernel, dim3(glab, nplanes), dim3(64), 0, ctx->stream, labels, box, cnt, nlab, TA, dP, err,
                       H, W, cap);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

static size_t cp_finish_bytes(int nplanes, size_t n, int cap, int fill_holes);
static int cp_finish(amt_ctx* ctx, int32_t* labels_out, const int* nlab1, int32_t* count_dev, const cp_box* box,
                     const int* cnt, const double* err, float flow_threshold, int min_size, int fill_holes, int nplanes,
                     int H, int W, int cap);

static int cellpose_masks_impl(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out,
                               int32_t* count_dev, int nplanes, int H, int W, float cellprob_threshold, int niter,
                               int min_size, float max_size_fraction, int max_seeds, float flow_threshold, int fill_holes) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(dP && cellprob && labels_out && count_dev && nplanes >= 0, "cellpose_masks: bad arguments");
    AMT_REQUIRE(H >= 2 && W >= 2 && H + 2 * CP_RPAD < 65536 && W + 2 * CP_RPAD < 65536,
                "cellpose_masks: image of %d x %d is outside 2..65495 per side", H, W);
    AMT_REQUIRE(niter >= 0 && max_seeds >= 1 && min_size >= 0, "cellpose_masks: bad niter / max_seeds / min_size");
    AMT_REQUIRE(flow_threshold >= 0.0f, "cellpose_masks: flow_threshold must be non-negative");
    if (nplanes == 0) return AMT_OK;
    const bool post = flow_threshold > 0.0f || fill_holes;
    const size_t n = (size_t)H * W;
    const int Hp = H + 2 * CP_RPAD, Wp = W + 2 * CP_RPAD;
    const size_t hn = (size_t)Hp * Wp;
    const int cap = max_seeds;
    const size_t nl = (size_t)nplanes * (cap + 1);
    size_t need = amt_align((size_t)nplanes * n * 4) + 2 * amt_align((size_t)nplanes * hn * 4) +
                  amt_align((size_t)nplanes * cap * 8) + amt_align((size_t)nplanes * cap * 4) + 3 * amt_align(nl * 4) +
                  amt_align((size_t)nplanes * 4);
    if (post)
        need += amt_align(nl * sizeof(cp_box)) + amt_align(nl * 4) + amt_align(nl * 16) + 2 * amt_align(nl * 8) +
                2 * amt_align((size_t)nplanes * 4) + 2 * amt_align((size_t)nplanes * n * 8) +
                cp_finish_bytes(nplanes, n, cap, fill_holes);
    need += amt_align((size_t)nplanes * n * 8) + amt_align((size_t)nplanes * n * 4) + amt_align((size_t)nplanes * 4);
    AMT_TRY(amt_arena_begin(ctx, need));
    unsigned* pos = arena_take_t<unsigned>(ctx, (size_t)nplanes * n);
    float2* fld = arena_take_t<float2>(ctx, (size_t)nplanes * n);  // (cell ? dP : 0) / 5, y and x
    int* mvlist = arena_take_t<int>(ctx, (size_t)nplanes * n);     // the pixels that move
    int* nmv = arena_take_t<int>(ctx, nplanes);
    AMT_HIP_CHECK(hipMemsetAsync(nmv, 0, (size_t)nplanes * 4, ctx->stream));
    int* hist = arena_take_t<int>(ctx, (size_t)nplanes * hn);
    int* M = arena_take_t<int>(ctx, (size_t)nplanes * hn);
    unsigned long long* seeds = arena_take_t<unsigned long long>(ctx, (size_t)nplanes * cap);
    unsigned* order = arena_take_t<unsigned>(ctx, (size_t)nplanes * cap);
    int* count = arena_take_t<int>(ctx, nl);
    int* first = arena_take_t<int>(ctx, nl);
    int* map = arena_take_t<int>(ctx, nl);
    int* nseeds = arena_take_t<int>(ctx, nplanes);
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * hn * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(M, 0, (size_t)nplanes * hn * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(count, 0, nl * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(nseeds, 0, (size_t)nplanes * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(count_dev, 0, (size_t)nplanes * 4, ctx->stream));
    hipLaunchKernelGGL(cp_fill_kernel, dim3(amt_grid_for(nl, 256, 1024)), dim3(256), 0, ctx->stream, first, nl, 0x7fffffff);
    AMT_LAUNCH_CHECK();
    dim3 gpx(amt_grid_for(n, 256, 4096), nplanes), ghist(amt_grid_for(hn, 256, 4096), nplanes);
    if (n < 0x7fffffffull) {
        // the field the taps read, the resting pixels' positions and the list of the moving ones; then the listed pixels
        hipLaunchKernelGGL(cp_follow_prep_kernel, gpx, dim3(256), 0, ctx->stream, dP, cellprob, cellprob_threshold, fld, pos,
                           mvlist, nmv, H, W);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(cp_follow_list_kernel, gpx, dim3(256), 0, ctx->stream, (const float2*)fld, niter, pos,
                           (const int*)mvlist, (const int*)nmv, H, W);
    } else {
        hipLaunchKernelGGL(cp_follow_kernel, gpx, dim3(256), 0, ctx->stream, dP, cellprob, cellprob_threshold, niter, pos, H, W);
    }
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
    // with the later stages, get_masks applies only its size ceiling; the floor belongs to step 6
    hipLaunchKernelGGL(cp_map_kernel, dim3(amt_grid_for((size_t)cap, 256, 64), nplanes), dim3(256), 0, ctx->stream, count,
                       first, nseeds, map, count_dev, cap, post ? 0 : min_size, big);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_apply_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, map, n, cap);
    AMT_LAUNCH_CHECK();
    if (!post) return AMT_OK;
    cp_box* box = arena_take_t<cp_box>(ctx, nl);
    int* cnt = arena_take_t<int>(ctx, nl);
    unsigned long long* sums = arena_take_t<unsigned long long>(ctx, 2 * nl);
    double* err = arena_take_t<double>(ctx, nl);
    int* center = arena_take_t<int>(ctx, 2 * nl);
    int* nit = arena_take_t<int>(ctx, nplanes);
    int* nlab1 = arena_take_t<int>(ctx, nplanes);  // number of get_masks labels (count_dev becomes the final count)
    double* TA = arena_take_t<double>(ctx, (size_t)nplanes * n);
    double* TB = arena_take_t<double>(ctx, (size_t)nplanes * n);
    AMT_HIP_CHECK(hipMemcpyAsync(nlab1, count_dev, (size_t)nplanes * 4, hipMemcpyDeviceToDevice, ctx->stream));
    const bool want_err = flow_threshold > 0.0f;
    AMT_TRY(cp_flow_errors(ctx, labels_out, dP, nlab1, box, cnt, sums, center, nit, TA, TB, err, nplanes, H, W, cap, want_err));
    return cp_finish(ctx, labels_out, nlab1, count_dev, box, cnt, err, flow_threshold, min_size, fill_holes, nplanes, H, W,
                     cap);
}

// step 6 (and the flow filter's verdicts): keep flags -> new numbers -> hole filling -> labels_io in place.
// nlab1[plane] = number of labels in labels_io (1..nlab1), count_dev receives the final number (stays -1 where it was -1).
// Scratch comes from the caller's arena reservation (cp_finish_bytes).
static size_t cp_finish_bytes(int nplanes, size_t n, int cap, int fill_holes) {
    const size_t nl = (size_t)nplanes * (cap + 1);
    return 4 * amt_align(nl * 4) + 2 * amt_align((size_t)nplanes * 4) +
           (fill_holes ? 2 * amt_align((size_t)nplanes * n * 4) + amt_align((size_t)nplanes * n) : 0);
}

static int cp_finish(amt_ctx* ctx, int32_t* labels_out, const int* nlab1, int32_t* count_dev, const cp_box* box,
                     const int* cnt, const double* err, float flow_threshold, int min_size, int fill_holes, int nplanes,
                     int H, int W, int cap) {
    const size_t n = (size_t)H * W;
    const size_t nl = (size_t)nplanes * (cap + 1);
    dim3 gpx(amt_grid_for(n, 256, 4096), nplanes);
    int* keep = arena_take_t<int>(ctx, nl);
    int* keepflow = arena_take_t<int>(ctx, nl);
    int* scan = arena_take_t<int>(ctx, nl);
    int* map2 = arena_take_t<int>(ctx, nl);
    int* total = arena_take_t<int>(ctx, nplanes);
    int* nested = arena_take_t<int>(ctx, nplanes);
    dim3 glab(amt_grid_for((size_t)cap + 1, 256, 64), nplanes);
    hipLaunchKernelGGL(cp_keep_kernel, glab, dim3(256), 0, ctx->stream, cnt, err, nlab1, keep, cap, min_size,
                       (double)flow_threshold);
    AMT_LAUNCH_CHECK();
    AMT_HIP_CHECK(hipMemcpyAsync(scan, keep, nl * 4, hipMemcpyDeviceToDevice, ctx->stream));
    AMT_TRY(amt_scan_excl(ctx, scan, cap + 1, (size_t)cap + 1, total, nplanes));
    hipLaunchKernelGGL(cp_newmap_kernel, glab, dim3(256), 0, ctx->stream, keep, scan, total, map2, count_dev, cap);
    AMT_LAUNCH_CHECK();
    if (!fill_holes) {
        hipLaunchKernelGGL(cp_apply_fill_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, map2, (const int*)nullptr, n, cap);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    int* fill = arena_take_t<int>(ctx, (size_t)nplanes * n);
    int* work = arena_take_t<int>(ctx, (size_t)nplanes * n);
    unsigned char* bytes = arena_take_t<unsigned char>(ctx, (size_t)nplanes * n);
    AMT_HIP_CHECK(hipMemsetAsync(fill, 0, (size_t)nplanes * n * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(nested, 0, (size_t)nplanes * 4, ctx->stream));
    const int gl = cap < 2048 ? cap : 2048;
    hipLaunchKernelGGL(cp_holes_kernel, dim3(gl, nplanes), dim3(256), 0, ctx->stream, labels_out, box, keep, map2, nlab1,
                       fill, nested, H, W, cap);
    AMT_LAUNCH_CHECK();
    // planes whose holes hold other masks: the package's label-by-label loop, one workgroup per plane
    hipLaunchKernelGGL(cp_keepflow_kernel, glab, dim3(256), 0, ctx->stream, cnt, err, nlab1, keepflow, cap,
                       (double)flow_threshold);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_seq_prepare_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, keepflow, nested, work, n, cap);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_holes_seq_kernel, dim3(nplanes), dim3(256), 0, ctx->stream, work, box, nlab1, nested, bytes,
                       total, H, W, cap, min_size);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_apply_fill_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, map2, fill, n, cap);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_seq_commit_kernel, gpx, dim3(256), 0, ctx->stream, labels_out, work, nested, n);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cp_seq_count_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, count_dev, total, nested,
                       nplanes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_cellpose_masks(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out,
                                  int32_t* count_dev, int nplanes, int H, int W, float cellprob_threshold, int niter,
                                  int min_size, float max_size_fraction, int max_seeds) {
    return cellpose_masks_impl(ctx, dP, cellprob, labels_out, count_dev, nplanes, H, W, cellprob_threshold, niter, min_size,
                               max_size_fraction, max_seeds, 0.0f, 0);
}

extern "C" int amt_cellpose_masks_ex(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out,
                                     int32_t* count_dev, int nplanes, int H, int W, float cellprob_threshold, int niter,
                                     int min_size, float max_size_fraction, int max_seeds, float flow_threshold,
                                     int fill_holes) {
    return cellpose_masks_impl(ctx, dP, cellprob, labels_out, count_dev, nplanes, H, W, cellprob_threshold, niter, min_size,
                               max_size_fraction, max_seeds, flow_threshold, fill_holes);
}

extern "C" int amt_cellpose_flow_error(amt_ctx* ctx, const int32_t* labels, const float* dP, const int32_t* nlabels_dev,
                                       double* err_out, int nplanes, int H, int W, int max_label) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels && dP && nlabels_dev && err_out && nplanes >= 0 && H >= 1 && W >= 1 && max_label >= 1,
                "cellpose_flow_error: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int cap = max_label;
    const size_t nl = (size_t)nplanes * (cap + 1);
    AMT_TRY(amt_arena_begin(ctx, amt_align(nl * sizeof(cp_box)) + amt_align(nl * 4) + amt_align(nl * 16) + amt_align(nl * 8) +
                                     amt_align(nl * 8) + amt_align((size_t)nplanes * 4) + 2 * amt_align((size_t)nplanes * n * 8)));
    cp_box* box = arena_take_t<cp_box>(ctx, nl);
    int* cnt = arena_take_t<int>(ctx, nl);
    unsigned long long* sums = arena_take_t<unsigned long long>(ctx, 2 * nl);
    int* center = arena_take_t<int>(ctx, 2 * nl);
    double* err = arena_take_t<double>(ctx, nl);
    int* nit = arena_take_t<int>(ctx, nplanes);
    double* TA = arena_take_t<double>(ctx, (size_t)nplanes * n);
    double* TB = arena_take_t<double>(ctx, (size_t)nplanes * n);
    AMT_TRY(cp_flow_errors(ctx, labels, dP, nlabels_dev, box, cnt, sums, center, nit, TA, TB, err, nplanes, H, W, cap, true));
    // err is (cap + 1) per plane with slot 0 unused: hand out slots 1..cap
    for (int p = 0; p < nplanes; ++p)
        AMT_HIP_CHECK(hipMemcpyAsync(err_out + (size_t)p * cap, err + (size_t)p * (cap + 1) + 1, (size_t)cap * 8,
                                     hipMemcpyDeviceToDevice, ctx->stream));
    return AMT_OK;
}

extern "C" int amt_fill_holes_remove_small(amt_ctx* ctx, int32_t* labels_io, const int32_t* nlabels_dev, int32_t* count_dev,
                                           int nplanes, int H, int W, int max_label, int min_size, int fill_holes) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(labels_io && nlabels_dev && count_dev && nplanes >= 0 && H >= 1 && W >= 1 && max_label >= 1 && min_size >= 0,
                "fill_holes_remove_small: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int cap = max_label;
    const size_t nl = (size_t)nplanes * (cap + 1);
    AMT_TRY(amt_arena_begin(ctx, amt_align(nl * sizeof(cp_box)) + amt_align(nl * 4) + amt_align(nl * 16) +
                                     cp_finish_bytes(nplanes, n, cap, fill_holes)));
    cp_box* box = arena_take_t<cp_box>(ctx, nl);
    int* cnt = arena_take_t<int>(ctx, nl);
    unsigned long long* sums = arena_take_t<unsigned long long>(ctx, 2 * nl);
    AMT_TRY(cp_flow_errors(ctx, labels_io, nullptr, nlabels_dev, box, cnt, sums, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nplanes, H, W, cap, false));
    AMT_HIP_CHECK(hipMemcpyAsync(count_dev, nlabels_dev, (size_t)nplanes * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return cp_finish(ctx, labels_io, nlabels_dev, count_dev, box, cnt, nullptr, 0.0f, min_size, fill_holes, nplanes, H, W, cap);
}
