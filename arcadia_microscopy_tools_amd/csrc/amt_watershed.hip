// Marker-based watershed (priority flood) restricted to a mask.
//
// Contract: skimage.segmentation.watershed(image, markers, connectivity=1, mask=mask) -- SURVEY.md A.1:
// pop the pending pixel with the smallest (value, insertion age); every unlabelled masked 4-neighbour,
// visited in the order (north, west, east, south), takes the popped pixel's label AT PUSH TIME and is
// queued with the next age.  All marker pixels start with age 0.
//
// What is order-defining and what is not:
//   * (value, age) is unique for every queued pixel except among the age-0 marker pixels.  scikit-image
//     breaks those ties by the internal moves of its binary heap; this implementation breaks them by
//     raster order (a stable queue).  The config-3 recipe gives marker pixels distinct, lowest values
//     (oracle/skops.py:seeded_flood_image; here `seeds_first`), for which both definitions coincide.
//   * the flood never crosses between 4-connected components of the mask, and the relative order of two
//     pixels of one component does not depend on the other components.  Each component is therefore
//     flooded independently and sequentially by ONE LANE; thousands of components run concurrently,
//     lanes pull components from a per-plane work counter until it is exhausted.
//
// amt_watershed_edt: relief = -sqrt(d2) with d2 an exact non-negative integer, so the priority queue
// is a bucket queue indexed by d2 (largest d2 = lowest relief first) with a FIFO per bucket: insertion
// age order inside a bucket is push order.  A pixel is pushed at most once, so each FIFO is a linked
// list threaded through one int per pixel.
// amt_watershed_f64: arbitrary float64 relief; per-component binary heap keyed (value, age, raster).
#include "amt_internal.h"

// ---- per-component bookkeeping -------------------------------------------------------------------
// out = markers * mask ; per-root accumulators cleared
__global__ void __launch_bounds__(256) ws_init_kernel(const int* __restrict__ markers, const uint8_t* __restrict__ mask,
                                                      const int* __restrict__ L, int* __restrict__ out,
                                                      int* __restrict__ rootMax, int* __restrict__ rootCnt, size_t n) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        out[base + i] = mask[base + i] ? markers[base + i] : 0;
        if (L[base + i] == (int)i) {
            rootMax[base + i] = 0;
            rootCnt[base + i] = 0;
        }
    }
}

template <typename TV>
__device__ __forceinline__ int bucket_of(TV v);
template <>
__device__ __forceinline__ int bucket_of<int>(int v) {
    return v < 0 ? 0 : v;
}

__global__ void __launch_bounds__(256) ws_stats_kernel(const int* __restrict__ d2, const int* __restrict__ L,
                                                       const int* __restrict__ out, int* __restrict__ rootMax,
                                                       int* __restrict__ rootCnt, size_t n, int use_d2) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int r = L[base + i];
        if (r < 0) continue;
        if (use_d2) {
            int v = d2[base + i];
            if (v > 0) atomicMax(&rootMax[base + r], v);
        } else {
            atomicAdd(&rootMax[base + r], 1);  // component size (heap capacity)
        }
        if (out[base + i] != 0) atomicAdd(&rootCnt[base + r], 1);
    }
}

// roots publish their compact rows: cmax[cid], mcnt[cid], bsz[cid] (bucket count), and reset rootCnt
__global__ void __launch_bounds__(256) ws_compact_kernel(const int* __restrict__ L, const int* __restrict__ T,
                                                         int* __restrict__ rootMax, int* __restrict__ rootCnt,
                                                         int* __restrict__ cmax, int* __restrict__ mcnt,
                                                         int* __restrict__ bsz, int* __restrict__ moff, size_t n,
                                                         int use_d2) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (L[base + i] == (int)i) {
            int cid = T[base + i] - 1;
            int mx = rootMax[base + i];
            int mc = rootCnt[base + i];
            cmax[base + cid] = mx;
            mcnt[base + cid] = mc;
            moff[base + cid] = mc;
            // bucket queue: buckets 0..mx ; heap: capacity = component size (only needed when it has markers)
            bsz[base + cid] = mc > 0 ? (use_d2 ? mx + 1 : mx) : 0;
            rootCnt[base + i] = 0;  // reused as the fill cursor of the marker list
        }
    }
}

__global__ void __launch_bounds__(256) ws_fill_markers_kernel(const int* __restrict__ L, const int* __restrict__ T,
                                                              const int* __restrict__ out,
                                                              const int* __restrict__ moff, int* __restrict__ rootCnt,
                                                              int* __restrict__ mlist, size_t n) {
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int r = L[base + i];
        if (r < 0 || out[base + i] == 0) continue;
        int cid = T[base + r] - 1;
        int pos = atomicAdd(&rootCnt[base + r], 1);
        mlist[base + moff[base + cid] + pos] = (int)i;
    }
}

__global__ void __launch_bounds__(256) ws_fill_neg1_kernel(int* __restrict__ buf, const int* __restrict__ total,
                                                           size_t plane_stride) {
    int* b = buf + (size_t)blockIdx.y * plane_stride;
    const int tot = total[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) b[i] = -1;
}

__global__ void ws_zero_counters_kernel(int* c, int nplanes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nplanes) c[i] = 0;
}

__device__ __forceinline__ void lane_sort(int* a, int n) {
    for (int i = 1; i < n; ++i) {
        int v = a[i], j = i - 1;
        while (j >= 0 && a[j] > v) {
            a[j + 1] = a[j];
            --j;
        }
        a[j + 1] = v;
    }
}

// ---- bucket-queue flood (relief = -sqrt(d2)) ------------------------------------------------------
__global__ void __launch_bounds__(64) ws_flood_edt_kernel(const int* __restrict__ d2all, const uint8_t* __restrict__ maskall,
                                                          int* __restrict__ outall, int* __restrict__ nextall,
                                                          int* __restrict__ headall, int* __restrict__ tailall,
                                                          int* __restrict__ mlistall, const int* __restrict__ cmaxall,
                                                          const int* __restrict__ mcntall, const int* __restrict__ moffall,
                                                          const int* __restrict__ boffall, const int* __restrict__ ncomp,
                                                          int* __restrict__ counters, int H, int W, size_t n,
                                                          size_t bstride, int seeds_first) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * n;
    const int* d2 = d2all + base;
    const uint8_t* mask = maskall + base;
    int* out = outall + base;
    int* next = nextall + base;
    const int nc = ncomp[plane];
    const int nbo[4] = {-W, -1, +1, +W};
    while (true) {
        const int c = atomicAdd(&counters[plane], 1);
        if (c >= nc) break;
        const int nm = mcntall[base + c];
        if (nm == 0) continue;
        int* ml = mlistall + base + moffall[base + c];
        int* hd = headall + (size_t)plane * bstride + boffall[base + c];
        int* tl = tailall + (size_t)plane * bstride + boffall[base + c];
        int cur = -1;
        lane_sort(ml, nm);

        auto push = [&](int p, int b) {
            next[p] = -1;
            if (hd[b] < 0) {
                hd[b] = p;
            } else {
                next[tl[b]] = p;
            }
            tl[b] = p;
            if (b > cur) cur = b;
        };
        auto spread = [&](int p) {
            const int lab = out[p];
            const int py = p / W, px = p - py * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 0 && py == 0) continue;
                if (k == 1 && px == 0) continue;
                if (k == 2 && px == W - 1) continue;
                if (k == 3 && py == H - 1) continue;
                const int q = p + nbo[k];
                if (mask[q] && out[q] == 0) {
                    out[q] = lab;
                    push(q, bucket_of<int>(d2[q]));
                }
            }
        };

        if (seeds_first) {
            for (int i = 0; i < nm; ++i) spread(ml[i]);
        } else {
            for (int i = 0; i < nm; ++i) push(ml[i], bucket_of<int>(d2[ml[i]]));
        }
        while (true) {
            while (cur >= 0 && hd[cur] < 0) --cur;
            if (cur < 0) break;
            const int p = hd[cur];
            hd[cur] = next[p];
            spread(p);
        }
    }
}

// ---- binary-heap flood (arbitrary float64 relief) --------------------------------------------------
struct hp_elem {
    double value;
    int age;
    int index;
};

__device__ __forceinline__ bool hp_less(const hp_elem& a, const hp_elem& b) {
    if (a.value != b.value) return a.value < b.value;
    if (a.age != b.age) return a.age < b.age;
    return a.index < b.index;  // only age-0 markers can tie on (value, age): raster order
}

__global__ void __launch_bounds__(64) ws_flood_heap_kernel(const double* __restrict__ relall,
                                                           const uint8_t* __restrict__ maskall, int* __restrict__ outall,
                                                           hp_elem* __restrict__ heapall, int* __restrict__ mlistall,
                                                           const int* __restrict__ mcntall, const int* __restrict__ moffall,
                                                           const int* __restrict__ boffall, const int* __restrict__ ncomp,
                                                           int* __restrict__ counters, int H, int W, size_t n,
                                                           size_t hstride) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * n;
    const double* rel = relall + base;
    const uint8_t* mask = maskall + base;
    int* out = outall + base;
    const int nc = ncomp[plane];
    const int nbo[4] = {-W, -1, +1, +W};
    while (true) {
        const int c = atomicAdd(&counters[plane], 1);
        if (c >= nc) break;
        const int nm = mcntall[base + c];
        if (nm == 0) continue;
        int* ml = mlistall + base + moffall[base + c];
        hp_elem* hp = heapall + (size_t)plane * hstride + boffall[base + c];
        int items = 0;
        int age = 0;
        lane_sort(ml, nm);
        auto hpush = [&](const hp_elem& e) {
            int child = items++;
            while (child > 0) {
                int parent = (child - 1) >> 1;
                hp_elem pe = hp[parent];
                if (hp_less(e, pe)) {
                    hp[child] = pe;
                    child = parent;
                } else
                    break;
            }
            hp[child] = e;
        };
        auto hpop = [&]() {
            hp_elem top = hp[0];
            hp_elem last = hp[--items];
            int parent = 0;
            while (true) {
                int child = 2 * parent + 1;
                if (child >= items) break;
                hp_elem ce = hp[child];
                if (child + 1 < items) {
                    hp_elem c2 = hp[child + 1];
                    if (hp_less(c2, ce)) {
                        ce = c2;
                        ++child;
                    }
                }
                if (hp_less(ce, last)) {
                    hp[parent] = ce;
                    parent = child;
                } else
                    break;
            }
            if (items > 0) hp[parent] = last;
            return top;
        };
        for (int i = 0; i < nm; ++i) {
            hp_elem e;
            e.value = rel[ml[i]];
            e.age = 0;
            e.index = ml[i];
            hpush(e);
        }
        while (items > 0) {
            hp_elem e = hpop();
            const int p = e.index;
            const int lab = out[p];
            const int py = p / W, px = p - py * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 0 && py == 0) continue;
                if (k == 1 && px == 0) continue;
                if (k == 2 && px == W - 1) continue;
                if (k == 3 && py == H - 1) continue;
                const int q = p + nbo[k];
                if (mask[q] && out[q] == 0) {
                    out[q] = lab;
                    hp_elem ne;
                    ne.value = rel[q];
                    ne.age = ++age;
                    ne.index = q;
                    hpush(ne);
                }
            }
        }
    }
}

static int watershed_common(amt_ctx* ctx, const void* relief, bool use_d2, const int32_t* markers,
                            const uint8_t* mask, int32_t* out, int nplanes, int H, int W, int seeds_first) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(relief && markers && mask && out && nplanes >= 0 && H > 0 && W > 0, "watershed: bad arguments");
    AMT_REQUIRE((size_t)H * W < 0x3fffffffull, "watershed: plane too large");
    AMT_REQUIRE((const void*)markers != (const void*)out, "watershed: markers and out must not alias");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const size_t np = (size_t)nplanes * n;
    const int nblk = amt_i_rank_blocks(n);
    // queue storage: bucket queue needs <= n + 2*ncomp <= 3n ints for head and tail each (only the used
    // prefix is initialised); the heap needs <= n elements per plane.
    const size_t bstride = use_d2 ? 3 * n : n;
    size_t need = 9 * amt_align(np * 4) + amt_align((size_t)nplanes * nblk * 4) + 4 * amt_align(nplanes * 4);
    need += use_d2 ? 2 * amt_align((size_t)nplanes * bstride * 4) : amt_align((size_t)nplanes * bstride * sizeof(hp_elem));
    AMT_TRY(amt_arena_begin(ctx, need));
    int* L = arena_take_t<int>(ctx, np);
    int* T = arena_take_t<int>(ctx, np);
    int* rootMax = arena_take_t<int>(ctx, np);
    int* rootCnt = arena_take_t<int>(ctx, np);
    int* cmax = arena_take_t<int>(ctx, np);
    int* mcnt = arena_take_t<int>(ctx, np);
    int* moff = arena_take_t<int>(ctx, np);
    int* boff = arena_take_t<int>(ctx, np);
    int* mlist = arena_take_t<int>(ctx, np);
    int* blk = arena_take_t<int>(ctx, (size_t)nplanes * nblk);
    int* ncomp = arena_take_t<int>(ctx, nplanes);
    int* btot = arena_take_t<int>(ctx, nplanes);
    int* mtot = arena_take_t<int>(ctx, nplanes);
    int* counters = arena_take_t<int>(ctx, nplanes);
    int *head = nullptr, *tail = nullptr, *next = nullptr;
    hp_elem* heap = nullptr;
    if (use_d2) {
        head = arena_take_t<int>(ctx, (size_t)nplanes * bstride);
        tail = arena_take_t<int>(ctx, (size_t)nplanes * bstride);
        next = T;  // T is dead once the marker lists are filled (see below): reuse as the FIFO links
    } else {
        heap = arena_take_t<hp_elem>(ctx, (size_t)nplanes * bstride);
    }

    AMT_TRY(amt_i_ccl_roots(ctx, mask, AMT_U8, L, nplanes, H, W, /*conn8=*/0));
    AMT_TRY(amt_i_rank_roots(ctx, L, T, blk, ncomp, nplanes, n));
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    hipLaunchKernelGGL(ws_init_kernel, g1, dim3(256), 0, ctx->stream, markers, mask, L, out, rootMax, rootCnt, n);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ws_stats_kernel, g1, dim3(256), 0, ctx->stream, use_d2 ? (const int*)relief : (const int*)nullptr,
                       L, out, rootMax, rootCnt, n, use_d2 ? 1 : 0);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ws_compact_kernel, g1, dim3(256), 0, ctx->stream, L, T, rootMax, rootCnt, cmax, mcnt, boff, moff,
                       n, use_d2 ? 1 : 0);
    AMT_LAUNCH_CHECK();
    // exclusive scans over the compact per-component rows (length = ncomp[plane] <= n; scan the device count)
    hipLaunchKernelGGL(ws_zero_counters_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, counters, nplanes);
    AMT_LAUNCH_CHECK();
    AMT_TRY(amt_scan_excl_dev(ctx, moff, ncomp, n, mtot, nplanes));
    AMT_TRY(amt_scan_excl_dev(ctx, boff, ncomp, n, btot, nplanes));
    hipLaunchKernelGGL(ws_fill_markers_kernel, g1, dim3(256), 0, ctx->stream, L, T, out, moff, rootCnt, mlist, n);
    AMT_LAUNCH_CHECK();
    dim3 gf(64, nplanes);
    if (use_d2) {
        hipLaunchKernelGGL(ws_fill_neg1_kernel, dim3(256, nplanes), dim3(256), 0, ctx->stream, head, btot, bstride);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ws_flood_edt_kernel, gf, dim3(64), 0, ctx->stream, (const int*)relief, mask, out, next, head,
                           tail, mlist, cmax, mcnt, moff, boff, ncomp, counters, H, W, n, bstride, seeds_first);
    } else {
        hipLaunchKernelGGL(ws_flood_heap_kernel, gf, dim3(64), 0, ctx->stream, (const double*)relief, mask, out, heap,
                           mlist, mcnt, moff, boff, ncomp, counters, H, W, n, bstride);
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_watershed_edt(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                                 int32_t* out, int nplanes, int H, int W, int seeds_first) {
    return watershed_common(ctx, d2, true, markers, mask, out, nplanes, H, W, seeds_first);
}

extern "C" int amt_watershed_f64(amt_ctx* ctx, const double* relief, const int32_t* markers, const uint8_t* mask,
                                 int32_t* out, int nplanes, int H, int W) {
    return watershed_common(ctx, relief, false, markers, mask, out, nplanes, H, W, 0);
}
