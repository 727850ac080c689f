// Histograms, min/max, exact percentiles, Otsu threshold and comparison kernels.
//
// Reference call sites: R/operations.py:47,94 (np.percentile), :186-216 (apply_threshold ->
// ski.filters.threshold_*), SK/exposure/exposure.py:38-144 (histogram), SK/filters/thresholding.py:321-350.
// All counting is integer and therefore order-independent; the only floating-point sequences whose
// order matters (the 256-bin cumulative sums of Otsu on float images) are evaluated sequentially, in
// numpy's order, by one lane.
#include "amt_common.h"

// ------------------------------------------------------------------------------------------------
// uint16 histogram: 65536 uint32 bins per plane.  A 1024-thread workgroup privatises a WINDOW of 32768 bins in
// 128 KiB of LDS and takes a chunk of at least 262,144 pixels, so that the flush (one global atomic per non-empty
// bin) is a fraction of the counting.  Window [0, 32768) comes first; only if the chunk held a value with the top bit
// set (microscopy cameras rarely fill it) does the workgroup read its chunk a second time for [32768, 65536) --
// round 2 sent those values to global atomics one by one, which made a full-range plane (82 us) ten times slower
// than a 12-bit one.
// ------------------------------------------------------------------------------------------------
constexpr int HIST_LDS_BINS = 32768;

__global__ void __launch_bounds__(1024) hist_u16_kernel(const uint16_t* __restrict__ in, uint32_t* __restrict__ hist,
                                                        size_t n, int blocks_per_plane) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem_raw);
    const int plane = blockIdx.x / blocks_per_plane;
    const int part = blockIdx.x - plane * blocks_per_plane;
    const uint16_t* src = in + (size_t)plane * n;
    uint32_t* gh = hist + (size_t)plane * 65536;
    // 8 pixels (16 bytes) per thread per step when aligned
    const size_t nvec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) ? n / 8 : 0;
    const uint4* v4 = reinterpret_cast<const uint4*>(src);
    // this workgroup's contiguous share of the vector part (the scalar tail is strided: a few pixels)
    const size_t per = (nvec + blocks_per_plane - 1) / blocks_per_plane;
    const size_t v0 = (size_t)part * per, v1 = v0 + per < nvec ? v0 + per : nvec;
    for (unsigned win = 0; win < 2; ++win) {
        for (int i = threadIdx.x; i < HIST_LDS_BINS; i += 1024) lh[i] = 0;
        __syncthreads();
        unsigned other = 0;  // values of the other window seen
        for (size_t i = v0 + threadIdx.x; i < v1; i += 1024) {
            const uint4 q = v4[i];
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t a = wds[k] & 0xffffu, b = wds[k] >> 16;
                if ((a >> 15) == win) atomicAdd(&lh[a & 0x7fffu], 1u); else other = 1;
                if ((b >> 15) == win) atomicAdd(&lh[b & 0x7fffu], 1u); else other = 1;
            }
        }
        for (size_t i = nvec * 8 + (size_t)part * 1024 + threadIdx.x; i < n; i += (size_t)blocks_per_plane * 1024) {
            const uint32_t a = src[i];
            if ((a >> 15) == win) atomicAdd(&lh[a & 0x7fffu], 1u); else other = 1;
        }
        const int more = __syncthreads_or((int)other);  // also orders the counting before the flush
        for (int i = threadIdx.x; i < HIST_LDS_BINS; i += 1024) {
            const uint32_t c = lh[i];
            if (c) atomicAdd(&gh[win * HIST_LDS_BINS + i], c);
        }
        if (win == 0 && !more) break;  // uniform: nothing for the upper window in this chunk
        __syncthreads();
    }
}

// Few planes: no atomics leave the workgroup at all.  Every workgroup counts a part of HIST_PART_PX pixels into ALL
// 65536 bins as 16-bit counters (two per LDS word; a counter cannot overflow within a part of fewer than 65536
// pixels), stores its 128 KiB as plain coalesced stores, and a second kernel adds the parts up.  A single full-range
// 2048^2 plane: 82 us (round 2) -> 54 us (two windows) -> see DESIGN.md for this path.
constexpr int HIST_PART_PX = 65528;   // multiple of 8 (16-byte loads), below 65536 (half-size parts: 28 -> 39 us per plane)
constexpr int HIST_MAX_PARTS = 1024;  // 128 MiB of partial histograms at most

__global__ void __launch_bounds__(1024) hist_u16_part_kernel(const uint16_t* __restrict__ in, uint32_t* __restrict__ partial,
                                                             size_t n, int parts) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem_raw);  // 32768 words = 65536 16-bit counters
    const int plane = blockIdx.x / parts;
    const int part = blockIdx.x - plane * parts;
    const uint16_t* src = in + (size_t)plane * n;
    uint4* l4 = reinterpret_cast<uint4*>(lh);
    for (int i = threadIdx.x; i < 8192; i += 1024) l4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const size_t p0 = (size_t)part * HIST_PART_PX;
    const size_t p1 = p0 + HIST_PART_PX < n ? p0 + HIST_PART_PX : n;
    const bool aligned = (reinterpret_cast<uintptr_t>(src + p0) & 15) == 0;
    const size_t nvec = aligned ? (p1 - p0) / 8 : 0;
    const uint4* v4 = reinterpret_cast<const uint4*>(src + p0);
    for (size_t i = threadIdx.x; i < nvec; i += 1024) {
        const uint4 q = v4[i];
        const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = wds[k] & 0xffffu, b = wds[k] >> 16;
            atomicAdd(&lh[a >> 1], 1u << ((a & 1u) * 16));
            atomicAdd(&lh[b >> 1], 1u << ((b & 1u) * 16));
        }
    }
    for (size_t i = p0 + nvec * 8 + threadIdx.x; i < p1; i += 1024) {
        const uint32_t a = src[i];
        atomicAdd(&lh[a >> 1], 1u << ((a & 1u) * 16));
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(partial + (size_t)blockIdx.x * 32768);
    for (int i = threadIdx.x; i < 8192; i += 1024) out[i] = l4[i];
}

// hist[plane][2 w], [2 w + 1] = sum over the plane's parts of the two 16-bit halves of word w.  Four lanes share a word
// and take every fourth part (the sums meet through two lane exchanges): a thread that walks all ~64 parts alone waits
// for them a few at a time (10 us per single plane; 5 with four lanes per word).
__global__ void __launch_bounds__(256) hist_u16_reduce_kernel(const uint32_t* __restrict__ partial, uint32_t* __restrict__ hist,
                                                              int parts) {
    const int plane = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int w = t >> 2, q = t & 3;  // word 0 .. 32767, quarter of the parts
    const uint32_t* p = partial + (size_t)plane * parts * 32768 + w;
    uint32_t lo = 0, hi = 0;
    for (int k = q; k < parts; k += 4) {
        const uint32_t v = p[(size_t)k * 32768];
        lo += v & 0xffffu;
        hi += v >> 16;
    }
    lo += __shfl_xor(lo, 1);
    hi += __shfl_xor(hi, 1);
    lo += __shfl_xor(lo, 2);
    hi += __shfl_xor(hi, 2);
    if (q == 0) reinterpret_cast<uint2*>(hist + (size_t)plane * 65536)[w] = make_uint2(lo, hi);
}

// bytes of scratch hist_u16_launch wants for this call (0: it takes the two-window kernel)
static size_t hist_u16_scratch_bytes(int nplanes, size_t n) {
    if (n == 0 || nplanes <= 0) return 0;
    const size_t parts = (n + HIST_PART_PX - 1) / HIST_PART_PX;
    if (parts * (size_t)nplanes > (size_t)HIST_MAX_PARTS) return 0;
    return amt_align(parts * (size_t)nplanes * 32768 * sizeof(uint32_t));
}

static int hist_u16_launch(amt_ctx* ctx, const uint16_t* in, uint32_t* hist, int nplanes, size_t n,
                           uint32_t* scratch = nullptr) {
    if (n == 0) {
        AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * 65536 * sizeof(uint32_t), ctx->stream));
        return AMT_OK;
    }
    if (scratch && hist_u16_scratch_bytes(nplanes, n) > 0) {
        const int parts = (int)((n + HIST_PART_PX - 1) / HIST_PART_PX);
        hipLaunchKernelGGL(hist_u16_part_kernel, dim3(nplanes * parts), dim3(1024), 32768 * sizeof(uint32_t), ctx->stream, in,
                           scratch, n, parts);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(hist_u16_reduce_kernel, dim3(512, nplanes), dim3(256), 0, ctx->stream, scratch, hist, parts);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * 65536 * sizeof(uint32_t), ctx->stream));
    // >= 262,144 pixels per workgroup (4 flushed bins per counted pixel at worst), and no more workgroups than keep
    // every CU busy twice over
    int bpp = (int)((n + 262143) / 262144);
    int cap = (2 * ctx->num_cus + nplanes - 1) / nplanes;
    if (cap < 1) cap = 1;
    if (bpp > cap) bpp = cap;
    if (bpp < 1) bpp = 1;
    hipLaunchKernelGGL(hist_u16_kernel, dim3(nplanes * bpp), dim3(1024), HIST_LDS_BINS * sizeof(uint32_t), ctx->stream,
                       in, hist, n, bpp);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_hist_u16(amt_ctx* ctx, const uint16_t* in, uint32_t* hist, int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && hist && nplanes >= 0, "hist_u16: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t sb = hist_u16_scratch_bytes(nplanes, n);
    uint32_t* scratch = nullptr;
    if (sb) {
        AMT_TRY(amt_arena_begin(ctx, sb));
        scratch = reinterpret_cast<uint32_t*>(amt_arena_take(ctx, sb));
    }
    return hist_u16_launch(ctx, in, hist, nplanes, n, scratch);
}

// ------------------------------------------------------------------------------------------------
// float64 min / max per plane (wave shuffle reduction -> one 64-bit atomic per wave on ordered keys)
// ------------------------------------------------------------------------------------------------
__global__ void minmax_init_kernel(unsigned long long* keys, int nplanes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nplanes) {
        keys[2 * i] = ~0ull;
        keys[2 * i + 1] = 0ull;
    }
}

__global__ void __launch_bounds__(256) minmax_f64_kernel(const double* __restrict__ in,
                                                         unsigned long long* __restrict__ keys, size_t n) {
    const int plane = blockIdx.y;
    const double* src = in + (size_t)plane * n;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long k = amt_f64_key(src[i]);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    // one pair of atomics per BLOCK (a few hundred per plane): same-address atomics serialise in L2
    __shared__ unsigned long long s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            lo = s_lo[k] < lo ? s_lo[k] : lo;
            hi = s_hi[k] > hi ? s_hi[k] : hi;
        }
        atomicMin(&keys[2 * plane], lo);
        atomicMax(&keys[2 * plane + 1], hi);
    }
}

__global__ void minmax_finish_kernel(const unsigned long long* keys, double* out, int nplanes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * nplanes) out[i] = amt_key_f64(keys[i]);
}

int amt_i_minmax_init(amt_ctx* ctx, unsigned long long* keys, int nplanes) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, keys, nplanes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_i_minmax_finish(amt_ctx* ctx, const unsigned long long* keys, double* out, int nplanes) {
    hipLaunchKernelGGL(minmax_finish_kernel, dim3((2 * nplanes + 63) / 64), dim3(64), 0, ctx->stream, keys, out,
                       nplanes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

static int minmax_f64_launch(amt_ctx* ctx, const double* in, unsigned long long* keys, double* out, int nplanes,
                             size_t n) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, keys, nplanes);
    AMT_LAUNCH_CHECK();
    if (n) {
        dim3 grid(amt_grid_for(n, 256 * 16, 256), nplanes);
        hipLaunchKernelGGL(minmax_f64_kernel, grid, dim3(256), 0, ctx->stream, in, keys, n);
        AMT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(minmax_finish_kernel, dim3((2 * nplanes + 63) / 64), dim3(64), 0, ctx->stream, keys, out,
                       nplanes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_i_minmax_f64(amt_ctx* ctx, const double* in, unsigned long long* keys, double* out, int nplanes, size_t n) {
    return minmax_f64_launch(ctx, in, keys, out, nplanes, n);
}

extern "C" int amt_minmax_f64(amt_ctx* ctx, const double* in, double* minmax_dev, int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && minmax_dev && nplanes >= 0, "minmax_f64: bad arguments");
    if (nplanes == 0) return AMT_OK;
    AMT_TRY(amt_arena_begin(ctx, amt_align(2 * nplanes * sizeof(unsigned long long))));
    unsigned long long* keys = arena_take_t<unsigned long long>(ctx, 2 * nplanes);
    return minmax_f64_launch(ctx, in, keys, minmax_dev, nplanes, n);
}

// ------------------------------------------------------------------------------------------------
// np.histogram(x, bins=nbins, range=(min,max)): edges = np.linspace(min, max, nbins+1) i.e.
// edges[i] = i*step + min with step = (max-min)/nbins and edges[nbins] = max; a sample belongs to the
// bin with edges[i] <= x < edges[i+1] (last bin closed).  numpy estimates the bin by scaling and then
// repairs it against the edges; repairing from any estimate gives the same bin.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double linspace_edge(double lo, double hi, double step, int i, int nbins) {
    return i == nbins ? hi : (double)i * step + lo;
}

// bins (nullable, nbins <= 256): the bin of every sample as a byte plane -- for a threshold that is the centre of bin k
// (Otsu), `sample > threshold` is decided by the byte alone except inside bin k (amt_threshold_open_close_bins)
__global__ void __launch_bounds__(256) hist_f64_kernel(const double* __restrict__ in,
                                                       const double* __restrict__ minmax,
                                                       uint32_t* __restrict__ hist, int nbins, size_t n,
                                                       uint8_t* __restrict__ bins) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* edges = reinterpret_cast<double*>(smem_raw);              // nbins + 1
    uint32_t* lh = reinterpret_cast<uint32_t*>(edges + nbins + 1);    // 4 waves x nbins
    const int plane = blockIdx.y;
    const double lo = minmax[2 * plane], hi = minmax[2 * plane + 1];
    const double step = (hi - lo) / (double)nbins;
    for (int i = threadIdx.x; i <= nbins; i += 256) edges[i] = linspace_edge(lo, hi, step, i, nbins);
    for (int i = threadIdx.x; i < 4 * nbins; i += 256) lh[i] = 0;
    __syncthreads();
    uint32_t* mine = lh + (threadIdx.x >> 6) * nbins;
    const double norm = (double)nbins / (hi - lo);
    const double* src = in + (size_t)plane * n;
    const int lane = threadIdx.x & 63;
    // four CONSECUTIVE samples per thread and step (two 16-byte loads, ONE packed 4-byte store of their bins); a
    // smoothed image is mostly background, so whole waves often fall into ONE bin: those add their population
    // count once instead of 64 same-address LDS atomics
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 31) == 0;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        double v4[4];
        if (vec) {
            const double2 a = *reinterpret_cast<const double2*>(src + i0), c = *reinterpret_cast<const double2*>(src + i0 + 2);
            v4[0] = a.x, v4[1] = a.y, v4[2] = c.x, v4[3] = c.y;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = (i0 + u < n) ? src[i0 + u] : __builtin_nan("");
        }
        // first guess of every bin and BOTH of its edges (eight LDS reads in flight, no dependent chain); the guess
        // is exact unless the value sits within rounding distance of an edge -- only then the walk below runs
        int b4[4];
        double elo4[4], ehi4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double v = v4[u];
            const bool ok = v >= lo && v <= hi;  // NaN / out of range: not counted
            int b = ok ? (int)((v - lo) * norm) : 0;
            b = b < 0 ? 0 : (b > nbins - 1 ? nbins - 1 : b);
            b4[u] = b;
            elo4[u] = (double)b * step + lo;  // == edges[b] (b < nbins), without the LDS round trip
            ehi4[u] = b + 1 == nbins ? hi : (double)(b + 1) * step + lo;
        }
        unsigned packed = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double v = v4[u];
            const bool ok = v >= lo && v <= hi;
            int b = b4[u];
            if (ok && (v < elo4[u] || (b < nbins - 1 && v >= ehi4[u]))) {
                while (b > 0 && v < edges[b]) --b;
                while (b < nbins - 1 && v >= edges[b + 1]) ++b;
            }
            // (a constant plane has no bins: lo == hi makes the scaling infinite -- its bytes are 0 like its threshold's)
            packed |= (unsigned)(ok && lo < hi ? b : 0) << (8 * u);
            const unsigned long long act = __ballot(ok);
            if (!act) continue;
            // two rounds of "first lane's bin, counted once for everyone who shares it" (64 same-address LDS atomics
            // cost 64 slots; a smoothed background wave straddles at most one bin edge), then lane by lane
            const int l0 = __ffsll((long long)act) - 1;
            const int b0 = __shfl(b, l0);
            const unsigned long long s0 = __ballot(ok && b == b0);
            if (lane == l0) atomicAdd(&mine[b0], (unsigned)__popcll(s0));
            unsigned long long rest = act & ~s0;
            if (rest) {
                const int l1 = __ffsll((long long)rest) - 1;
                const int b1 = __shfl(b, l1);
                const unsigned long long s1 = __ballot(ok && b == b1);
                if (lane == l1) atomicAdd(&mine[b1], (unsigned)__popcll(s1));
                rest &= ~s1;
                if ((rest >> lane) & 1ull) atomicAdd(&mine[b], 1u);
            }
        }
        if (bins) {
            uint8_t* dst = bins + (size_t)plane * n + i0;
            if (vec) {
                *reinterpret_cast<unsigned*>(dst) = packed;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < n) dst[u] = (uint8_t)(packed >> (8 * u));
            }
        }
    }
    __syncthreads();
    uint32_t* gh = hist + (size_t)plane * nbins;
    for (int i = threadIdx.x; i < nbins; i += 256) {
        uint32_t c = lh[i] + lh[nbins + i] + lh[2 * nbins + i] + lh[3 * nbins + i];
        if (c) atomicAdd(&gh[i], c);
    }
}

static int hist_f64_launch(amt_ctx* ctx, const double* in, const double* minmax, uint32_t* hist, int nbins,
                           int nplanes, size_t n, uint8_t* bins = nullptr) {
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * nbins * sizeof(uint32_t), ctx->stream));
    if (n == 0) return AMT_OK;
    size_t smem = (size_t)(nbins + 1) * sizeof(double) + (size_t)4 * nbins * sizeof(uint32_t);
    static const int hg = getenv("AMT_HIST_GRID") ? atoi(getenv("AMT_HIST_GRID")) : 512;  // A/B: blocks per plane
    dim3 grid(amt_grid_for(n, 256 * 16, hg), nplanes);
    hipLaunchKernelGGL(hist_f64_kernel, grid, dim3(256), smem, ctx->stream, in, minmax, hist, nbins, n, bins);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_hist_f64(amt_ctx* ctx, const double* in, const double* minmax_dev, uint32_t* hist, int nbins,
                            int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && minmax_dev && hist && nplanes >= 0, "hist_f64: bad arguments");
    AMT_REQUIRE(nbins >= 1 && nbins <= 4096, "hist_f64: nbins %d out of range 1..4096", nbins);
    if (nplanes == 0) return AMT_OK;
    return hist_f64_launch(ctx, in, minmax_dev, hist, nbins, nplanes, n);
}

// ------------------------------------------------------------------------------------------------
// One bin per integer value over an arbitrary range: scikit-image's histogram of integer images
// (SK/exposure/exposure.py:63-74: np.bincount(image - image_min, minlength = image_max - image_min + 1)) for images whose
// range exceeds the 65,536 values amt_hist_u16 covers.  The image travels as float64 (integers are exact below 2^53);
// counts go straight to global atomics -- with millions of bins per image they rarely collide.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hist_range_kernel(const double* __restrict__ in, double lo, long long nbins,
                                                         uint32_t* __restrict__ hist, size_t n) {
    const double* src = in + (size_t)blockIdx.y * n;
    uint32_t* h = hist + (size_t)blockIdx.y * (size_t)nbins;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double d = src[i] - lo;
        if (d >= 0.0 && d < (double)nbins) atomicAdd(&h[(long long)d], 1u);
    }
}

extern "C" int amt_hist_range_f64(amt_ctx* ctx, const double* in, double lo, int64_t nbins, uint32_t* hist, int nplanes,
                                  size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && hist && nplanes >= 0 && nbins >= 1, "hist_range_f64: bad arguments");
    AMT_REQUIRE(nbins <= (1ll << 28), "hist_range_f64: %lld bins are more than 2^28", (long long)nbins);
    if (nplanes == 0) return AMT_OK;
    AMT_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)nplanes * (size_t)nbins * sizeof(uint32_t), ctx->stream));
    if (n == 0) return AMT_OK;
    hipLaunchKernelGGL(hist_range_kernel, dim3(amt_grid_for(n, 256, 4096), nplanes), dim3(256), 0, ctx->stream, in, lo,
                       (long long)nbins, hist, n);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ------------------------------------------------------------------------------------------------
// Otsu (SK/filters/thresholding.py:336-348) on a histogram, one 1024-thread workgroup per plane.
//   EXACT_INT: uint16 images -- bins image_min..image_max, bin centre = value; all cumulative sums
//              are exact integers, so the parallel scan equals numpy's sequential float64 cumsum.
//   else     : float images -- nbins (256) bins, centres = edge midpoints; the cumulative sums are
//              evaluated sequentially by lane 0 in numpy's order.
// threshold = centre[argmax(w1[:-1] * w2[1:] * (m1[:-1] - m2[1:])**2)], first maximum wins.
// A constant image returns its value (skimage's early exit).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_argmax_first(double v, int idx, double* s_val, int* s_idx, double& best,
                                                   int& best_idx) {
    // NaN never wins (np.argmax would return the first NaN, which cannot occur here: all weights > 0)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double v2 = __shfl_xor(v, off);
        int i2 = __shfl_xor(idx, off);
        if (v2 > v || (v2 == v && i2 < idx)) {
            v = v2;
            idx = i2;
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_val[wave] = v;
        s_idx[wave] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = s_val[0];
        int bi = s_idx[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (s_val[k] > bv || (s_val[k] == bv && s_idx[k] < bi)) {
                bv = s_val[k];
                bi = s_idx[k];
            }
        s_val[0] = bv;
        s_idx[0] = bi;
    }
    __syncthreads();
    best = s_val[0];
    best_idx = s_idx[0];
    __syncthreads();
}

__global__ void __launch_bounds__(1024) otsu_u16_kernel(const uint32_t* __restrict__ hist, double* __restrict__ thr) {
    // 65536 bins, 64 per thread
    __shared__ unsigned long long s_cnt[1024];
    __shared__ unsigned long long s_sum[1024];
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_min, s_max;
    const int plane = blockIdx.x;
    const uint32_t* h = hist + (size_t)plane * 65536;
    const int t = threadIdx.x;
    const int b0 = t * 64;
    if (t == 0) {
        s_min = 65536;
        s_max = -1;
    }
    __syncthreads();
    unsigned long long cnt = 0, sum = 0;
    int lmin = 65536, lmax = -1;
    for (int k = 0; k < 64; ++k) {
        uint32_t c = h[b0 + k];
        if (c) {
            if (lmin == 65536) lmin = b0 + k;
            lmax = b0 + k;
        }
        cnt += c;
        sum += (unsigned long long)c * (unsigned)(b0 + k);
    }
    if (lmin < 65536) atomicMin(&s_min, lmin);
    if (lmax >= 0) atomicMax(&s_max, lmax);
    s_cnt[t] = cnt;
    s_sum[t] = sum;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 1024 per-thread totals
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long c2 = 0, s2 = 0;
        if (t >= off) {
            c2 = s_cnt[t - off];
            s2 = s_sum[t - off];
        }
        __syncthreads();
        s_cnt[t] += c2;
        s_sum[t] += s2;
        __syncthreads();
    }
    const unsigned long long total_cnt = s_cnt[1023], total_sum = s_sum[1023];
    const int vmin = s_min, vmax = s_max;
    if (vmax < 0 || vmin == vmax) {  // empty or constant image
        if (t == 0) thr[plane] = vmax < 0 ? 0.0 : (double)vmin;
        return;
    }
    unsigned long long run_cnt = s_cnt[t] - cnt, run_sum = s_sum[t] - sum;  // exclusive prefix
    double best = -1.0;
    int best_idx = 0x7fffffff;
    for (int k = 0; k < 64; ++k) {
        int v = b0 + k;
        uint32_t c = h[v];
        run_cnt += c;
        run_sum += (unsigned long long)c * (unsigned)v;
        if (v >= vmin && v < vmax) {
            double w1 = (double)run_cnt;
            double w2 = (double)(total_cnt - run_cnt);
            double m1 = (double)run_sum / w1;
            double m2 = (double)(total_sum - run_sum) / w2;
            double d = m1 - m2;
            double var = (w1 * w2) * (d * d);
            if (var > best) {
                best = var;
                best_idx = v;
            }
        }
    }
    double bv;
    int bi;
    block_argmax_first(best, best_idx, s_val, s_idx, bv, bi);
    if (t == 0) thr[plane] = (double)bi;
}

// thr_code (nullable): 2 * (index of the threshold's bin), the threshold in the code space of amt_gaussian_otsu_codes
__global__ void __launch_bounds__(256) otsu_f64_kernel(const uint32_t* __restrict__ hist,
                                                       const double* __restrict__ minmax, int nbins,
                                                       double* __restrict__ thr, double* __restrict__ thr_code) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* ctr = reinterpret_cast<double*>(smem_raw);  // nbins
    double* w1 = ctr + nbins;
    double* w2 = w1 + nbins;
    double* m1 = w2 + nbins;
    double* m2 = m1 + nbins;
    __shared__ double s_val[4];
    __shared__ int s_idx[4];
    const int plane = blockIdx.x;
    const uint32_t* h = hist + (size_t)plane * nbins;
    const double lo = minmax[2 * plane], hi = minmax[2 * plane + 1];
    if (!(lo < hi)) {  // constant image (or NaN): skimage returns the first pixel
        if (threadIdx.x == 0) {
            thr[plane] = lo;
            if (thr_code) thr_code[plane] = 0.0;
        }
        return;
    }
    const double step = (hi - lo) / (double)nbins;
    for (int i = threadIdx.x; i < nbins; i += 256) {
        double e0 = linspace_edge(lo, hi, step, i, nbins), e1 = linspace_edge(lo, hi, step, i + 1, nbins);
        ctr[i] = (e0 + e1) / 2.0;
    }
    // counts and counts * centres into both directions' arrays (the running sums overwrite them in place)
    for (int i = threadIdx.x; i < nbins; i += 256) {
        const double c = (double)h[i];
        w1[i] = w2[i] = c;
        m1[i] = m2[i] = c * ctr[i];
    }
    __syncthreads();
    // the two cumulative sums are sequential by definition (np.cumsum's order fixes the rounding): one lane each, in
    // different waves, with the divisions left to everybody afterwards -- the chains are 2 x nbins additions long
    // instead of nbins x (load, convert, multiply, two additions, division) twice over (50 -> ~10 us for 256 bins)
    if (threadIdx.x == 0) {
        double cw = w1[0], cs = m1[0];
        for (int i = 1; i < nbins; ++i) {  // np.cumsum(counts), np.cumsum(counts * centers)
            cw = cw + w1[i];
            cs = cs + m1[i];
            w1[i] = cw;
            m1[i] = cs;
        }
    } else if (threadIdx.x == 64) {
        double cw = w2[nbins - 1], cs = m2[nbins - 1];
        for (int i = nbins - 2; i >= 0; --i) {  // reversed cumsums
            cw = cw + w2[i];
            cs = cs + m2[i];
            w2[i] = cw;
            m2[i] = cs;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256) {
        m1[i] = m1[i] / w1[i];
        m2[i] = m2[i] / w2[i];
    }
    __syncthreads();
    double best = -1.0;
    int best_idx = 0x7fffffff;
    for (int i = threadIdx.x; i < nbins - 1; i += 256) {
        double d = m1[i] - m2[i + 1];
        double var = (w1[i] * w2[i + 1]) * (d * d);
        if (var > best) {  // NaN (0/0 for leading empty bins cannot occur: bin 0 holds the minimum)
            best = var;
            best_idx = i;
        }
    }
    double bv;
    int bi;
    block_argmax_first(best, best_idx, s_val, s_idx, bv, bi);
    if (threadIdx.x == 0) {
        thr[plane] = ctr[bi == 0x7fffffff ? 0 : bi];
        if (thr_code) thr_code[plane] = (double)(2 * (bi == 0x7fffffff ? 0 : bi));
    }
}

int amt_i_otsu_from_hist(amt_ctx* ctx, const uint32_t* hist, const double* minmax, int nbins, double* thr,
                         double* thr_code, int nplanes) {
    hipLaunchKernelGGL(otsu_f64_kernel, dim3(nplanes), dim3(256), (size_t)5 * nbins * sizeof(double), ctx->stream, hist,
                       minmax, nbins, thr, thr_code);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_threshold_value(amt_ctx* ctx, const void* in, int in_dtype, int method, int nbins, double* thr_dev,
                                   int32_t* status_dev, int nplanes, size_t n, const double* minmax_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_dev && nplanes >= 0, "threshold_value: bad arguments");
    AMT_REQUIRE(!minmax_dev || in_dtype == AMT_F64, "threshold_value: minmax_dev applies to float64 input only");
    AMT_REQUIRE(method == AMT_THR_OTSU,
                "threshold_value: only AMT_THR_OTSU runs fully on the device; other methods are evaluated by the "
                "host layer on amt_hist_* output");
    AMT_REQUIRE(in_dtype == AMT_U16 || in_dtype == AMT_F64, "threshold_value: dtype must be AMT_U16 or AMT_F64");
    if (nplanes == 0) return AMT_OK;
    if (status_dev) AMT_HIP_CHECK(hipMemsetAsync(status_dev, 0, (size_t)nplanes * sizeof(int32_t), ctx->stream));
    if (in_dtype == AMT_U16) {
        const size_t sb = hist_u16_scratch_bytes(nplanes, n);
        AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * 65536 * sizeof(uint32_t)) + sb));
        uint32_t* hist = arena_take_t<uint32_t>(ctx, (size_t)nplanes * 65536);
        uint32_t* scratch = sb ? reinterpret_cast<uint32_t*>(amt_arena_take(ctx, sb)) : nullptr;
        AMT_TRY(hist_u16_launch(ctx, (const uint16_t*)in, hist, nplanes, n, scratch));
        hipLaunchKernelGGL(otsu_u16_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, hist, thr_dev);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    AMT_REQUIRE(nbins >= 2 && nbins <= 4096, "threshold_value: nbins %d out of range", nbins);
    AMT_TRY(amt_arena_begin(ctx, amt_align(2 * nplanes * 8) * 2 + amt_align((size_t)nplanes * nbins * 4)));
    unsigned long long* keys = arena_take_t<unsigned long long>(ctx, 2 * nplanes);
    double* mm = arena_take_t<double>(ctx, 2 * nplanes);
    uint32_t* hist = arena_take_t<uint32_t>(ctx, (size_t)nplanes * nbins);
    if (minmax_dev)
        mm = const_cast<double*>(minmax_dev);
    else
        AMT_TRY(minmax_f64_launch(ctx, (const double*)in, keys, mm, nplanes, n));
    AMT_TRY(hist_f64_launch(ctx, (const double*)in, mm, hist, nbins, nplanes, n));
    return amt_i_otsu_from_hist(ctx, hist, mm, nbins, thr_dev, nullptr, nplanes);
}

// Otsu on float64 planes whose [min, max] is known (the Gaussian folds it in), leaving the 256-bin index of every sample
// behind as a byte plane: thr_dev = the threshold (bit-identical to amt_threshold_value), thr_code_dev = 2 * its bin.
extern "C" int amt_otsu_f64_bins(amt_ctx* ctx, const double* in, const double* minmax_dev, double* thr_dev,
                                 double* thr_code_dev, uint8_t* bins, int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && minmax_dev && thr_dev && thr_code_dev && bins && nplanes >= 0, "otsu_f64_bins: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const int nbins = 256;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * nbins * 4)));
    uint32_t* hist = arena_take_t<uint32_t>(ctx, (size_t)nplanes * nbins);
    AMT_TRY(hist_f64_launch(ctx, in, minmax_dev, hist, nbins, nplanes, n, bins));
    return amt_i_otsu_from_hist(ctx, hist, minmax_dev, nbins, thr_dev, thr_code_dev, nplanes);
}

// ------------------------------------------------------------------------------------------------
// comparisons
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void threshold_gt_kernel(const T* __restrict__ in, const double* __restrict__ thr, uint8_t* __restrict__ out,
                                    size_t n) {
    const double t = thr[blockIdx.y];
    const size_t base = (size_t)blockIdx.y * n;
    // 4 pixels per thread -> one 32-bit store
    size_t nq = n / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        size_t i = base + q * 4;
        uint32_t m = ((double)in[i] > t ? 1u : 0u) | ((double)in[i + 1] > t ? 0x100u : 0u) |
                     ((double)in[i + 2] > t ? 0x10000u : 0u) | ((double)in[i + 3] > t ? 0x1000000u : 0u);
        if (((base + q * 4) & 3) == 0)
            *reinterpret_cast<uint32_t*>(out + i) = m;
        else {
            out[i] = m & 1;
            out[i + 1] = (m >> 8) & 1;
            out[i + 2] = (m >> 16) & 1;
            out[i + 3] = (m >> 24) & 1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = base + nq * 4 + threadIdx.x;
        out[i] = (double)in[i] > t ? 1 : 0;
    }
}

extern "C" int amt_threshold_gt(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev, uint8_t* out,
                                int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_dev && out && nplanes >= 0, "threshold_gt: bad arguments");
    if (nplanes == 0 || n == 0) return AMT_OK;
    dim3 grid(amt_grid_for(n / 4 + 1, 256, 2048), nplanes);
    if (in_dtype == AMT_U16)
        hipLaunchKernelGGL((threshold_gt_kernel<uint16_t>), grid, dim3(256), 0, ctx->stream, (const uint16_t*)in,
                           thr_dev, out, n);
    else if (in_dtype == AMT_F64)
        hipLaunchKernelGGL((threshold_gt_kernel<double>), grid, dim3(256), 0, ctx->stream, (const double*)in, thr_dev,
                           out, n);
    else {
        amt_set_error("threshold_gt: dtype must be AMT_U16 or AMT_F64");
        return AMT_EINVAL;
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

template <typename T>
__global__ void threshold_gt_image_kernel(const T* __restrict__ in, const double* __restrict__ thr,
                                          uint8_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i] > thr[i] ? 1 : 0;
}

extern "C" int amt_threshold_gt_image(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_image,
                                      uint8_t* out, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_image && out, "threshold_gt_image: bad arguments");
    if (n == 0) return AMT_OK;
    dim3 grid(amt_grid_for(n, 256));
    if (in_dtype == AMT_U16)
        hipLaunchKernelGGL((threshold_gt_image_kernel<uint16_t>), grid, dim3(256), 0, ctx->stream, (const uint16_t*)in,
                           thr_image, out, n);
    else if (in_dtype == AMT_F64)
        hipLaunchKernelGGL((threshold_gt_image_kernel<double>), grid, dim3(256), 0, ctx->stream, (const double*)in,
                           thr_image, out, n);
    else {
        amt_set_error("threshold_gt_image: dtype must be AMT_U16 or AMT_F64");
        return AMT_EINVAL;
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ------------------------------------------------------------------------------------------------
// np.percentile, method 'linear' (numpy _function_base_impl.py: virtual index (n-1)*q/100, _lerp):
//   lerp = a + (b-a)*t, and for t >= 0.5: b - (b-a)*(1-t).
// uint16: exact order statistics from the 65536-bin histogram.
// float64: exact MSB-first radix select on order-preserving 64-bit keys, 8 bits per pass, all
//          requested ranks resolved together.
// ------------------------------------------------------------------------------------------------
struct rank_req {
    long long lo;  // floor(virtual index); hi = min(lo + 1, n - 1), or lo == n-1 when above bounds
    long long hi;
    double t;
};

static void make_rank_reqs(const double* q_host, int nq, size_t n, rank_req* out) {
    for (int i = 0; i < nq; ++i) {
        double quant = q_host[i] / 100.0;
        double v = (double)(n - 1) * quant;
        long long lo, hi;
        if (v >= (double)(n - 1)) {
            lo = hi = (long long)n - 1;
        } else if (v < 0) {
            lo = hi = 0;
        } else {
            lo = (long long)__builtin_floor(v);
            hi = lo + 1;
        }
        // gamma = virtual - previous index as numpy computes it (previous = -1 -> n when above bounds,
        // where a == b makes the value irrelevant)
        out[i].lo = lo;
        out[i].hi = hi;
        out[i].t = v - __builtin_floor(v);
        if (v >= (double)(n - 1) || v < 0) out[i].t = 0.0;
    }
}

__device__ __forceinline__ double np_lerp(double a, double b, double t) {
    double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    return r;
}

__global__ void __launch_bounds__(1024) percentile_u16_kernel(const uint32_t* __restrict__ hist,
                                                              const rank_req* __restrict__ reqs, int nq,
                                                              double* __restrict__ out) {
    __shared__ unsigned long long s_cnt[1024];
    const int plane = blockIdx.x;
    const uint32_t* h = hist + (size_t)plane * 65536;
    const int t = threadIdx.x, b0 = t * 64;
    unsigned long long cnt = 0;
    for (int k = 0; k < 64; ++k) cnt += h[b0 + k];
    s_cnt[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long c2 = (t >= off) ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += c2;
        __syncthreads();
    }
    const unsigned long long before = s_cnt[t] - cnt;  // samples with value < b0
    __shared__ int s_lo[64], s_hi[64];
    for (int qi = 0; qi < nq; ++qi) {
        unsigned long long rlo = (unsigned long long)reqs[qi].lo, rhi = (unsigned long long)reqs[qi].hi;
        unsigned long long run = before;
        for (int k = 0; k < 64; ++k) {
            unsigned long long nxt = run + h[b0 + k];
            if (rlo >= run && rlo < nxt) s_lo[qi] = b0 + k;
            if (rhi >= run && rhi < nxt) s_hi[qi] = b0 + k;
            run = nxt;
        }
    }
    __syncthreads();
    if (t < nq) out[(size_t)plane * nq + t] = np_lerp((double)s_lo[t], (double)s_hi[t], reqs[t].t);
}

extern "C" int amt_percentile_u16(amt_ctx* ctx, const uint16_t* in, const double* q_host, int nq, double* out_dev,
                                  int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && q_host && out_dev && nplanes >= 0, "percentile_u16: bad arguments");
    AMT_REQUIRE(nq >= 1 && nq <= 64, "percentile_u16: nq %d out of range 1..64", nq);
    AMT_REQUIRE(n >= 1, "percentile_u16: empty image");
    for (int i = 0; i < nq; ++i)
        AMT_REQUIRE(q_host[i] >= 0.0 && q_host[i] <= 100.0, "Percentiles must be in the range [0, 100]");
    if (nplanes == 0) return AMT_OK;
    rank_req reqs[64];
    make_rank_reqs(q_host, nq, n, reqs);
    const size_t sb = hist_u16_scratch_bytes(nplanes, n);
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * 65536 * 4) + amt_align(sizeof(reqs)) + sb));
    uint32_t* hist = arena_take_t<uint32_t>(ctx, (size_t)nplanes * 65536);
    rank_req* rd = arena_take_t<rank_req>(ctx, 64);
    uint32_t* scratch = sb ? reinterpret_cast<uint32_t*>(amt_arena_take(ctx, sb)) : nullptr;
    AMT_TRY(amt_param_upload(ctx, rd, reqs, sizeof(rank_req) * nq));
    AMT_TRY(hist_u16_launch(ctx, in, hist, nplanes, n, scratch));
    hipLaunchKernelGGL(percentile_u16_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, hist, rd, nq, out_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// radix select state per (plane, rank): prefix (resolved high bits), remaining rank within prefix
struct sel_state {
    unsigned long long prefix;
    unsigned long long rank;
};

__global__ void sel_init_kernel(sel_state* st, const rank_req* reqs, int nq, int nplanes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nplanes * nq * 2) {
        int r = i % (2 * nq);
        int qi = r >> 1;
        st[i].prefix = 0;
        st[i].rank = (unsigned long long)((r & 1) ? reqs[qi].hi : reqs[qi].lo);
    }
}

// counts[plane][slot][256]: histogram of the next 8-bit digit among keys matching each slot's prefix
// With `cand` (pass SEL_LIST_PASS) every key that matches a slot's prefix is also appended to that slot's candidate
// list (order is irrelevant for a selection); the remaining passes then read the lists instead of the image.
constexpr int SEL_LIST_PASS = 3;  // after it 32 bits are resolved; the lists hold the keys matching the first 24
__global__ void __launch_bounds__(256) sel_count_kernel(const double* __restrict__ in,
                                                        const sel_state* __restrict__ st,
                                                        uint32_t* __restrict__ counts, int nslots, int pass, size_t n,
                                                        unsigned long long* __restrict__ cand,
                                                        unsigned* __restrict__ ncand, size_t cap,
                                                        const int* __restrict__ only_planes) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem_raw);  // nslots x 256
    unsigned long long* pre = reinterpret_cast<unsigned long long*>(lh + nslots * 256);
    const int plane = blockIdx.y;
    if (only_planes && !only_planes[plane]) return;  // passes 4..7 exist only for planes whose lists overflowed
    for (int i = threadIdx.x; i < nslots * 256; i += 256) lh[i] = 0;
    if (threadIdx.x < nslots) pre[threadIdx.x] = st[plane * nslots + threadIdx.x].prefix;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    const double* src = in + (size_t)plane * n;
    // identical prefixes (e.g. lo and hi of one percentile) share the first matching slot: bit s of `uniq` is set
    // when slot s is the first of its group
    unsigned uniq = 0;
    for (int s = 0; s < nslots; ++s) {
        bool dup = false;
        for (int s2 = 0; s2 < s; ++s2) dup |= (pre[s2] == pre[s]);
        if (!dup) uniq |= 1u << s;
    }
    const int lane = threadIdx.x & 63;
    // four independent loads per thread and step; in the first passes whole waves fall into one (slot, digit):
    // those add their population count once instead of 64 same-address LDS atomics
    for (size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        unsigned long long k4[4];
        bool ok4[4];
        double raw4[4];
        // unconditional loads (clamped to the last element); keys are formed once all four are in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            ok4[u] = i < n;
            raw4[u] = src[ok4[u] ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) k4[u] = amt_f64_key(raw4[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long k = k4[u];
            const unsigned digit = (unsigned)(k >> shift) & 255u;
            const unsigned long long hk = k & himask;
            int slot = -1;
            for (int s = 0; s < nslots; ++s)
                if (ok4[u] && ((uniq >> s) & 1u) && hk == pre[s]) slot = s;
            const int bin = slot >= 0 ? slot * 256 + (int)digit : -1;
            const unsigned long long act = __ballot(bin >= 0);
            if (!act) continue;
            // two rounds of "first lane's bin, counted once for all lanes that share it", then lane by lane
            const int first = __ffsll((long long)act) - 1;
            const int bin0 = __shfl(bin, first);
            const unsigned long long s0 = __ballot(bin >= 0 && bin == bin0);
            if (lane == first) atomicAdd(&lh[bin0], (unsigned)__popcll(s0));
            unsigned long long rest = act & ~s0;
            if (rest) {
                const int l1 = __ffsll((long long)rest) - 1;
                const int bin1 = __shfl(bin, l1);
                const unsigned long long s1 = __ballot(bin >= 0 && bin == bin1);
                if (lane == l1) atomicAdd(&lh[bin1], (unsigned)__popcll(s1));
                rest &= ~s1;
                if ((rest >> lane) & 1ull) atomicAdd(&lh[bin], 1u);
            }
            if (cand) {
                // one atomic per wave and slot (massive ties -- a clipped image is mostly zeros -- would otherwise
                // hammer ONE counter with millions of atomics); a list that has already overflowed is abandoned: the
                // counter only has to end above `cap`
                for (int sl = 0; sl < nslots; ++sl) {
                    const unsigned long long m = __ballot(slot == sl);
                    if (!m) continue;
                    unsigned* ctr = &ncand[plane * nslots + sl];
                    unsigned base = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader) base = (*ctr > cap) ? (unsigned)cap + 1u : atomicAdd(ctr, (unsigned)__popcll(m));
                    base = __shfl(base, leader);
                    if (slot == sl) {
                        const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                        if (pos < cap) cand[((size_t)plane * nslots + sl) * cap + pos] = k;
                    }
                }
            }
        }
    }
    __syncthreads();
    uint32_t* g = counts + (size_t)plane * nslots * 256;
    for (int i = threadIdx.x; i < nslots * 256; i += 256)
        if (lh[i]) atomicAdd(&g[i], lh[i]);
}

// passes after SEL_LIST_PASS, ALL in one launch: one block per (plane, slot) walks that slot's candidate list
// (or, if the list overflowed -- massive ties, e.g. a constant image -- the whole plane) once per remaining digit,
// picks the digit and narrows its own state.  Slots never depend on each other here: a slot whose group shared a
// list in pass SEL_LIST_PASS just reads the list that hangs on the group's first slot.
__global__ void __launch_bounds__(256) sel_list_passes_kernel(const double* __restrict__ in, sel_state* __restrict__ st,
                                                              int nslots, size_t n,
                                                              const unsigned long long* __restrict__ cand,
                                                              const unsigned* __restrict__ ncand, size_t cap,
                                                              const int* __restrict__ overflow) {
    __shared__ uint32_t lh[256];
    __shared__ unsigned long long s_prefix, s_rank;
    const int s = blockIdx.x, plane = blockIdx.y;
    if (overflow[plane]) return;  // resolved by the full passes 4..7 instead
    sel_state* sp = st + (size_t)plane * nslots;
    if (threadIdx.x == 0) {
        s_prefix = sp[s].prefix;
        s_rank = sp[s].rank;
    }
    __syncthreads();
    // the list of a group of equal 24-bit prefixes hangs on its first slot (the top 24 bits never change again)
    int src = s;
    for (int s2 = 0; s2 < s; ++s2)
        if (sp[s2].prefix >> 40 == s_prefix >> 40) {
            src = s2;
            break;
        }
    const unsigned cnt = ncand[plane * nslots + src];
    const unsigned long long* lst = cand + ((size_t)plane * nslots + src) * cap;
    const double* srcp = in + (size_t)plane * n;
    for (int pass = SEL_LIST_PASS + 1; pass < 8; ++pass) {
        lh[threadIdx.x] = 0;
        __syncthreads();
        const int shift = 56 - 8 * pass;
        const unsigned long long himask = ~0ull << (shift + 8);
        const unsigned long long mine = s_prefix;
        if (cnt <= cap) {
            for (unsigned i = threadIdx.x; i < cnt; i += 256) {
                const unsigned long long k = lst[i];
                if ((k & himask) == mine) atomicAdd(&lh[(unsigned)(k >> shift) & 255u], 1u);
            }
        } else {
            for (size_t i = threadIdx.x; i < n; i += 256) {
                const unsigned long long k = amt_f64_key(srcp[i]);
                if ((k & himask) == mine) atomicAdd(&lh[(unsigned)(k >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = 0;
            int d = 0;
            for (; d < 256; ++d) {
                const unsigned long long nxt = run + lh[d];
                if (s_rank < nxt) break;
                run = nxt;
            }
            if (d == 256) d = 255;
            s_prefix = mine | ((unsigned long long)d << shift);
            s_rank -= run;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sp[s].prefix = s_prefix;
        sp[s].rank = s_rank;
    }
}

// one thread per (plane, slot): pick the digit from the counts of this pass.  State is double-buffered (read
// st_in, write st_out) because slots of one plane read each other's prefixes to find their shared counts.
__global__ void sel_pick_kernel(const sel_state* __restrict__ st_in, sel_state* __restrict__ st_out,
                                const uint32_t* __restrict__ counts, int nslots, int pass, int nplanes,
                                const int* __restrict__ only_planes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nplanes * nslots) return;
    int plane = i / nslots, s = i - plane * nslots;
    if (only_planes && !only_planes[plane]) {  // the plane's state passes through unchanged
        st_out[i] = st_in[i];
        return;
    }
    // find the first slot with the same prefix (that is where the counts were accumulated)
    int src = s;
    for (int s2 = 0; s2 < s; ++s2)
        if (st_in[plane * nslots + s2].prefix == st_in[i].prefix) {
            src = s2;
            break;
        }
    const uint32_t* c = counts + ((size_t)plane * nslots + src) * 256;
    unsigned long long rank = st_in[i].rank, run = 0;
    int d = 0;
    for (; d < 256; ++d) {
        unsigned long long nxt = run + c[d];
        if (rank < nxt) break;
        run = nxt;
    }
    if (d == 256) d = 255;
    const int shift = 56 - 8 * pass;
    st_out[i].prefix = st_in[i].prefix | ((unsigned long long)d << shift);
    st_out[i].rank = rank - run;
}

// overflow[plane] = 1 if one of the plane's candidate lists did not hold all the keys that share its 24-bit prefix
// (massive ties or near-ties): that plane is resolved by four more FULL passes on the whole chip instead of the lists
__global__ void sel_overflow_kernel(const unsigned* __restrict__ ncand, int* __restrict__ overflow, int nslots,
                                    int nplanes, unsigned cap) {
    const int plane = blockIdx.x * blockDim.x + threadIdx.x;
    if (plane >= nplanes) return;
    int o = 0;
    for (int s = 0; s < nslots; ++s) o |= ncand[plane * nslots + s] > cap;
    overflow[plane] = o;
}

__global__ void sel_finish_kernel(const sel_state* st, const rank_req* reqs, int nq, int nplanes, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nplanes * nq) return;
    int plane = i / nq, qi = i - plane * nq;
    double a = amt_key_f64(st[(size_t)plane * 2 * nq + 2 * qi].prefix);
    double b = amt_key_f64(st[(size_t)plane * 2 * nq + 2 * qi + 1].prefix);
    out[i] = np_lerp(a, b, reqs[qi].t);
}

// ---- float64 percentiles in ONE full pass: sampled splitters, exact selection ------------------------------------
// For planes of >= PQ_MIN_N samples.  (1) pq_sample: PQ_M keys from hashed positions of the plane, sorted in LDS; for
// every requested percentile a BRACKET [a, b] of two sample keys around its expected position, wide enough (+-4 sigma
// of the sample rank + 8) that the wanted order statistics lie inside it except once in ~10^4 brackets.  (2) pq_classify:
// the one full read -- per bracket the exact counts of keys < a, == a, == b and the list of keys strictly between.
// (3) pq_resolve: a rank inside the "== a" / "== b" classes is that key (massive ties -- a clipped image is mostly
// zeros -- cost nothing); a rank inside the list is found by a radix select over the list (a few per cent of the plane,
// by the same idea one level down: a sorted sample of the LIST gives a sub-bracket, one pass over the list collects the
// keys inside it into LDS, a sort of those gives the answer; a list of <= 16,384 keys is sorted outright).  A rank
// outside a bracket, or a list / LDS buffer that overflowed, makes that one block fall back to an exact radix select
// (over the list, or over the whole plane: slow, ~10 ms, about once in 10^4 brackets).  Every path returns exact order
// statistics; np.percentile's lerp follows.
constexpr int PQ_M = 8192;
constexpr size_t PQ_MIN_N = 65536;

// ascending bitonic sort of N (power of two, >= 2048) keys in LDS by a 1024-thread block.  Thread t owns the pairs
// t + 1024 u; the 64 pairs of a wave span ONE run of 128 consecutive keys whenever the stride is <= 64, and the same
// run for every such stride -- so those stages only need the wave's own LDS order (a wave's LDS operations complete in
// program order) and the block barrier is kept for the stages that cross waves: 33 barriers instead of 91 for 8,192
// keys (pq_sample 115 -> 70 us, pq_resolve 180 -> 125 us per 32 planes).
template <int N>
__device__ __forceinline__ void pq_block_sort(unsigned long long* S) {
    const int t = threadIdx.x;
    int prev = 0;  // stride of the stage before (the caller's barrier stands before the first stage)
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride > 64 || prev > 64) __syncthreads();  // uniform
#pragma unroll
            for (int u = 0; u < N / 2048; ++u) {
                const int i = t + u * 1024;  // pair index
                const int lo = (i / stride) * 2 * stride + (i % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long x = S[lo], y = S[hi];
                if ((x > y) == up) {
                    S[lo] = y;
                    S[hi] = x;
                }
            }
            prev = stride;
        }
    }
    __syncthreads();
}

// bracket of sample positions around the expected positions plo..phi of the wanted ranks in a sorted sample of m keys
__device__ __forceinline__ void pq_bracket_positions(double plo, double phi, double m, double sigmas, long long* ia,
                                                     long long* ib) {
    const double dlo = sigmas * sqrt(fmax(plo * (m - plo) / m, 0.0)) + 8.0;
    const double dhi = sigmas * sqrt(fmax(phi * (m - phi) / m, 0.0)) + 8.0;
    *ia = (long long)floor(plo - dlo);
    *ib = (long long)ceil(phi + dhi);
}
struct pq_bracket {
    unsigned long long a, b;  // sample keys; has_a / has_b = 0: open end
    int has_a, has_b;
    unsigned below, eq_a, eq_b, n_in;  // exact counts from the classify pass (n_in may exceed the list capacity)
};

__global__ void __launch_bounds__(1024) pq_sample_kernel(const double* __restrict__ in, const rank_req* __restrict__ reqs,
                                                         int nq, pq_bracket* __restrict__ br, size_t n) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* S = reinterpret_cast<unsigned long long*>(smem_raw);  // PQ_M keys
    const int plane = blockIdx.x, t = threadIdx.x;
    const double* src = in + (size_t)plane * n;
#pragma unroll
    for (int u = 0; u < PQ_M / 1024; ++u) {
        const unsigned k = (unsigned)(t + u * 1024);
        // multiplicative hashing spreads the sample over the plane without any period related to the image width
        const size_t idx = (size_t)(((unsigned long long)k * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull * (unsigned)plane) % n);
        S[k] = amt_f64_key(src[idx]);
    }
    __syncthreads();
    pq_block_sort<PQ_M>(S);
    if (t < nq) {
        const double m = (double)PQ_M;
        long long ia, ib;
        pq_bracket_positions((double)reqs[t].lo / (double)n * m, (double)reqs[t].hi / (double)n * m, m, 4.0, &ia, &ib);
        pq_bracket b;
        b.has_a = ia >= 0;
        b.has_b = ib < PQ_M;
        b.a = b.has_a ? S[ia] : 0ull;
        b.b = b.has_b ? S[ib] : ~0ull;
        b.below = b.eq_a = b.eq_b = b.n_in = 0;
        br[(size_t)plane * nq + t] = b;
    }
}

// Keys inside a bracket are first collected in an LDS stage of the block (wave-aggregated LDS atomics) and flushed to
// the bracket's list with ONE global atomic per flush: a per-wave global atomic on the list counter would serialise
// ~50,000 atomics per plane on one address.  (Measured alternative: a stage per wave with the fill level in a register
// and no block barriers ran 1.7x slower, 803 vs 476 us per 32 planes.)
constexpr int PQ_STAGE = 2048;  // keys of one bracket a block stages in LDS (it meets ~230 on average; the rest overflow to the list)
constexpr int PQ_GROUP = 4;     // brackets per classify launch (LDS: PQ_GROUP x PQ_STAGE keys)
// NQ: brackets of this launch (1 .. PQ_GROUP), unrolled exactly, so that one or two brackets spill nothing.  VEC2: the plane
// is read in 16-byte pairs (even n, aligned base).  A thread has PQ_VPT samples of one iteration in flight and the next
// iteration's PQ_VPT requested: with four samples per thread a wave waited a full memory round trip per 2 KB (5 us under
// load, 40 us per block, 2.1 TB/s -- rocprofv3 counters: 27 vector instructions per 64 samples, waves idle 70 %).
constexpr int PQ_VPT = 8;
template <int NQ, bool VEC2>
__global__ void __launch_bounds__(256) pq_classify_kernel(const double* __restrict__ in, pq_bracket* __restrict__ br,
                                                          int nq_all, int j0, unsigned long long* __restrict__ lists,
                                                          size_t cap, size_t n, uint4* __restrict__ partial) {
    // brackets j0 .. j0 + nq - 1 of the plane's nq_all
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(smem_raw);  // nq x PQ_STAGE keys
    constexpr int nq = NQ;
    br += j0;
    lists += (size_t)j0 * cap;
    __shared__ pq_bracket sb[8];
    __shared__ unsigned sc[8][3];
    __shared__ unsigned s_fill[8], s_base[8];
    const int plane = blockIdx.y;
    if (threadIdx.x < nq) sb[threadIdx.x] = br[(size_t)plane * nq_all + threadIdx.x];
    if (threadIdx.x < 24) sc[threadIdx.x / 3][threadIdx.x % 3] = 0;
    if (threadIdx.x < 8) s_fill[threadIdx.x] = 0;
    __syncthreads();
    const double* src = in + (size_t)plane * n;
    const int lane = threadIdx.x & 63;
    // The comparisons are the only per-lane work: a compare writes a 64-lane mask, and counting, combining and
    // ranking masks is scalar arithmetic that the vector unit never sees.  (The first version kept per-lane counters
    // and one LDS atomic per value: ~250 vector instructions per 256 samples and bracket, and the pass was bound by
    // instruction issue at 2.3 TB/s.)  Brackets live in scalar registers; an open end is a sentinel + a uniform flag.
    unsigned long long A[NQ], Bk[NQ];
    bool HA[NQ], HB[NQ], SAME[NQ];
    unsigned w_below[NQ], w_eqa[NQ], w_eqb[NQ];  // wave-uniform counts
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        A[j] = sb[j].a;
        Bk[j] = sb[j].b;
        HA[j] = sb[j].has_a != 0;
        HB[j] = sb[j].has_b != 0;
        SAME[j] = HA[j] && A[j] == Bk[j];
        w_below[j] = w_eqa[j] = w_eqb[j] = 0;
    }
    const size_t per_iter = (size_t)gridDim.x * 256 * PQ_VPT;
    const size_t iters = (n + per_iter - 1) / per_iter;
    // No barrier inside the walk: the waves of a block run freely and their loads stay in flight (a flush check per
    // 1,024 samples cost two block barriers each and held the pass at 2.3 TB/s).  The stage is emptied once, at the end.
    auto flush = [&]() {
        __syncthreads();
        for (int j = 0; j < nq; ++j) {
            const unsigned fill = min(s_fill[j], (unsigned)PQ_STAGE);  // uniform; reservations beyond the stage went to the list
            if (fill == 0) continue;
            if (threadIdx.x == 0) s_base[j] = atomicAdd(&br[(size_t)plane * nq_all + j].n_in, fill);
            __syncthreads();
            const size_t base = s_base[j];
            for (unsigned i = threadIdx.x; i < fill; i += 256)
                if (base + i < cap) lists[((size_t)plane * nq_all + j) * cap + base + i] = stage[(size_t)j * PQ_STAGE + i];
            __syncthreads();
        }
    };
    // the samples of iteration it + 1 are requested before those of iteration it are processed; sample u of thread t
    // is element u * 256 + t of the block's run (scalar) or pair (u / 2) * 256 + t, half u & 1 (VEC2)
    double nxt[PQ_VPT];
    auto index_of = [&](size_t it, int u) -> size_t {
        const size_t run = it * per_iter + (size_t)blockIdx.x * (256 * PQ_VPT);
        return VEC2 ? run + (size_t)(u >> 1) * 512 + 2 * threadIdx.x + (u & 1) : run + (size_t)u * 256 + threadIdx.x;
    };
    auto request = [&](size_t it) {
        if constexpr (VEC2) {
#pragma unroll
            for (int p2 = 0; p2 < PQ_VPT / 2; ++p2) {  // unconditional loads from clamped (even) indices, all in flight
                const size_t i = index_of(it, 2 * p2);
                const double2 v = *reinterpret_cast<const double2*>(src + (i + 1 < n ? i : n - 2));
                nxt[2 * p2] = v.x;
                nxt[2 * p2 + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < PQ_VPT; ++u) {
                const size_t i = index_of(it, u);
                nxt[u] = src[i < n ? i : n - 1];
            }
        }
    };
    request(0);
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    for (size_t it = 0; it < iters; ++it) {
        unsigned long long k4[PQ_VPT], okm[PQ_VPT];
#pragma unroll
        for (int u = 0; u < PQ_VPT; ++u) {
            okm[u] = __ballot(index_of(it, u) < n);
            k4[u] = amt_f64_key(nxt[u]);
        }
        if (it + 1 < iters) request(it + 1);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            unsigned long long ins[PQ_VPT];
            unsigned total = 0;
#pragma unroll
            for (int u = 0; u < PQ_VPT; ++u) {
                const unsigned long long k = k4[u];
                const unsigned long long lt_a = __ballot(k < A[j]) & okm[u], eq_a = __ballot(k == A[j]) & okm[u];
                const unsigned long long lt_b = __ballot(k < Bk[j]) & okm[u], eq_b = __ballot(k == Bk[j]) & okm[u];
                const unsigned long long gt_a = okm[u] & ~lt_a & ~eq_a;
                ins[u] = (HA[j] ? gt_a : okm[u]) & (HB[j] ? lt_b : okm[u]);
                w_below[j] += HA[j] ? (unsigned)__popcll(lt_a) : 0u;
                w_eqa[j] += HA[j] ? (unsigned)__popcll(eq_a) : 0u;
                w_eqb[j] += (HB[j] && !SAME[j]) ? (unsigned)__popcll(eq_b) : 0u;
                total += (unsigned)__popcll(ins[u]);
            }
            if (total) {  // uniform: ONE stage reservation per wave, bracket and PQ_VPT values
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_fill[j], total);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (base + total <= (unsigned)PQ_STAGE) {
#pragma unroll
                    for (int u = 0; u < PQ_VPT; ++u) {
                        const unsigned pos = base + (unsigned)__popcll(ins[u] & lanes_below);
                        if ((ins[u] >> lane) & 1ull) stage[(size_t)j * PQ_STAGE + pos] = k4[u];
                        base += (unsigned)__popcll(ins[u]);
                    }
                } else {
                    // the stage is full (a block meets ~230 keys of a bracket on average; this is a plane whose values
                    // are sorted in space): what still fits goes to the stage, the rest straight to the list
                    unsigned long long ov[PQ_VPT];
                    unsigned nover = 0, b2 = base;
#pragma unroll
                    for (int u = 0; u < PQ_VPT; ++u) {
                        const unsigned pos = b2 + (unsigned)__popcll(ins[u] & lanes_below);
                        const bool mine = (ins[u] >> lane) & 1ull;
                        if (mine && pos < (unsigned)PQ_STAGE) stage[(size_t)j * PQ_STAGE + pos] = k4[u];
                        ov[u] = __ballot(mine && pos >= (unsigned)PQ_STAGE);
                        nover += (unsigned)__popcll(ov[u]);
                        b2 += (unsigned)__popcll(ins[u]);
                    }
                    unsigned gb = 0;
                    if (lane == 0) gb = atomicAdd(&br[(size_t)plane * nq_all + j].n_in, nover);
                    gb = (unsigned)__builtin_amdgcn_readfirstlane((int)gb);
#pragma unroll
                    for (int u = 0; u < PQ_VPT; ++u) {
                        const size_t pos = (size_t)gb + (unsigned)__popcll(ov[u] & lanes_below);
                        if (((ov[u] >> lane) & 1ull) && pos < cap)
                            lists[((size_t)plane * nq_all + j) * cap + pos] = k4[u];
                        gb += (unsigned)__popcll(ov[u]);
                    }
                }
            }
        }
    }
    flush();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (w_below[j]) atomicAdd(&sc[j][0], w_below[j]);
            if (w_eqa[j]) atomicAdd(&sc[j][1], w_eqa[j]);
            if (w_eqb[j]) atomicAdd(&sc[j][2], w_eqb[j]);
        }
    }
    __syncthreads();
    // The block's counts go to a slot of its own and pq_resolve adds the slots up.  Atomics of every block of a plane
    // on the bracket's one cache line cost ~0.4 us EACH, one after the other (measured: 128 / 512 / 2,048 blocks per
    // plane -> 0.55 / 0.8 / 3.3 ms per 32 planes): that line, not the 1 GB read, was what bounded this pass.  The list
    // cursor (one returning atomic per block and bracket) is all that still meets there.
    if (threadIdx.x < nq)
        partial[((size_t)plane * nq_all + j0 + threadIdx.x) * gridDim.x + blockIdx.x] =
            make_uint4(sc[threadIdx.x][0], sc[threadIdx.x][1], sc[threadIdx.x][2], 0u);
}

// k-th smallest (0-based) of `cnt` keys produced by `key_at(i)`, all of which share the bits above byte `first_pass`;
// one 1024-thread block, LDS histogram per byte
template <typename KeyAt>
__device__ unsigned long long pq_block_select(KeyAt key_at, size_t cnt, unsigned long long rank, unsigned long long prefix,
                                              int first_pass, uint32_t* lh, unsigned long long* s_prefix,
                                              unsigned long long* s_rank) {
    if (threadIdx.x == 0) {
        *s_prefix = prefix;
        *s_rank = rank;
    }
    __syncthreads();
    for (int pass = first_pass; pass < 8; ++pass) {
        if (threadIdx.x < 256) lh[threadIdx.x] = 0;
        __syncthreads();
        const int shift = 56 - 8 * pass;
        const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        const unsigned long long mine = *s_prefix;
        for (size_t i = threadIdx.x; i < cnt; i += 1024) {
            const unsigned long long k = key_at(i);
            if ((k & himask) == mine) atomicAdd(&lh[(unsigned)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = 0;
            int d = 0;
            for (; d < 256; ++d) {
                const unsigned long long nxt = run + lh[d];
                if (*s_rank < nxt) break;
                run = nxt;
            }
            if (d == 256) d = 255;
            *s_prefix = mine | ((unsigned long long)d << shift);
            *s_rank -= run;
        }
        __syncthreads();
    }
    return *s_prefix;
}

constexpr int PQ_BUF = 16384;  // keys the resolve block can hold (and sort) in LDS
constexpr int PQ_M2 = 4096;    // sample of a list

// the (r0)-th and, if two, the (r0 + 1)-th smallest key of lst[0..c): see the header comment.  B = PQ_BUF keys of LDS.
__device__ void pq_list_select(const unsigned long long* __restrict__ lst, size_t c, unsigned long long r0, int two,
                               unsigned long long a1, unsigned long long b1, int has_ab, unsigned long long* B,
                               uint32_t* lh, unsigned long long* sh64, unsigned* sh32, unsigned long long* out2) {
    const int t = threadIdx.x;
    auto radix = [&](unsigned long long r) -> unsigned long long {  // bytes above the first differing one are shared
        int first = 0;
        unsigned long long prefix = 0;
        if (has_ab) {
            const unsigned long long x = a1 ^ b1;
            first = x ? (__clzll((long long)x) >> 3) : 7;
            prefix = first ? (a1 & (~0ull << (64 - 8 * first))) : 0ull;
        }
        return pq_block_select([&](size_t i) { return lst[i]; }, c, r, prefix, first, lh, &sh64[0], &sh64[1]);
    };
    if (c <= (size_t)PQ_BUF) {  // sort the list outright
        const int cap2 = c <= 2048 ? 2048 : c <= 4096 ? 4096 : c <= 8192 ? 8192 : PQ_BUF;
        for (int i = t; i < cap2; i += 1024) B[i] = (size_t)i < c ? lst[i] : ~0ull;
        __syncthreads();
        if (cap2 == 2048) pq_block_sort<2048>(B);
        else if (cap2 == 4096) pq_block_sort<4096>(B);
        else if (cap2 == 8192) pq_block_sort<8192>(B);
        else pq_block_sort<PQ_BUF>(B);
        out2[0] = B[r0];
        out2[1] = two ? B[r0 + 1] : B[r0];
        __syncthreads();
        return;
    }
#pragma unroll
    for (int u = 0; u < PQ_M2 / 1024; ++u) {
        const unsigned k = (unsigned)(t + u * 1024);
        B[k] = lst[(size_t)(((unsigned long long)k * 0x9E3779B97F4A7C15ull + 12345ull) % c)];
    }
    __syncthreads();
    pq_block_sort<PQ_M2>(B);
    long long ia, ib;
    const double m2 = (double)PQ_M2;
    pq_bracket_positions((double)r0 / (double)c * m2, (double)(r0 + (two ? 1 : 0)) / (double)c * m2, m2, 3.5, &ia, &ib);
    const bool ha = ia >= 0, hb = ib < PQ_M2;
    const unsigned long long a2 = ha ? B[ia] : 0ull, b2 = hb ? B[ib] : ~0ull;
    __syncthreads();  // everybody has read the sample: B is free
    if (t < 4) sh32[t] = 0;  // below, eq_a, eq_b, fill
    __syncthreads();
    const int lane = t & 63;
    unsigned cb = 0, ca = 0, ce = 0;
    // one block walks the whole list (a few per cent of the plane): eight entries per thread in flight, or the walk is
    // a chain of ~100 dependent memory round trips
    constexpr int LU = 8;
    const size_t iters = (c + 1024 * LU - 1) / (1024 * LU);
    for (size_t it = 0; it < iters; ++it) {
        unsigned long long k8[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const size_t i = (it * LU + u) * 1024 + t;
            k8[u] = lst[i < c ? i : c - 1];
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const size_t i = (it * LU + u) * 1024 + t;
            const bool ok = i < c;
            const unsigned long long k = k8[u];
            cb += ok && ha && k < a2;
            ca += ok && ha && k == a2;
            ce += ok && hb && k == b2 && !(ha && a2 == b2);
            const bool inside = ok && (!ha || k > a2) && (!hb || k < b2);
            const unsigned long long m = __ballot(inside);
            if (m) {
                unsigned base = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) base = atomicAdd(&sh32[3], (unsigned)__popcll(m));
                base = __shfl(base, leader);
                const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (inside && pos < (unsigned)PQ_BUF) B[pos] = k;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cb += __shfl_xor(cb, o);
        ca += __shfl_xor(ca, o);
        ce += __shfl_xor(ce, o);
    }
    if (lane == 0) {
        atomicAdd(&sh32[0], cb);
        atomicAdd(&sh32[1], ca);
        atomicAdd(&sh32[2], ce);
    }
    __syncthreads();
    const unsigned long long e0 = sh32[0], e1 = e0 + sh32[1], fill = sh32[3], e2 = e1 + fill, e3 = e2 + sh32[2];
    __syncthreads();
    const bool sortable = fill <= (unsigned long long)PQ_BUF;
    if (sortable) {  // sort the smallest power of two that holds the keys (uniform choice)
        const int cap2 = fill <= 2048 ? 2048 : fill <= 4096 ? 4096 : fill <= 8192 ? 8192 : PQ_BUF;
        for (int i = t; i < cap2; i += 1024)
            if ((unsigned long long)i >= fill) B[i] = ~0ull;
        __syncthreads();
        if (cap2 == 2048) pq_block_sort<2048>(B);
        else if (cap2 == 4096) pq_block_sort<4096>(B);
        else if (cap2 == 8192) pq_block_sort<8192>(B);
        else pq_block_sort<PQ_BUF>(B);
    }
    for (int w = 0; w < 2; ++w) {
        const unsigned long long r = r0 + (unsigned long long)(w && two ? 1 : 0);
        unsigned long long key;
        if (ha && r >= e0 && r < e1) key = a2;
        else if (sortable && r >= e1 && r < e2) key = B[r - e1];
        else if (hb && r >= e2 && r < e3) key = b2;
        else key = radix(r);  // outside the sub-bracket, or too many keys inside it
        out2[w] = key;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) pq_resolve_kernel(const double* __restrict__ in, const rank_req* __restrict__ reqs,
                                                          const pq_bracket* __restrict__ br, int nq,
                                                          const unsigned long long* __restrict__ lists, size_t cap,
                                                          sel_state* __restrict__ res, size_t n,
                                                          const uint4* __restrict__ partial, int nparts) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* B = reinterpret_cast<unsigned long long*>(smem_raw);  // PQ_BUF keys
    __shared__ uint32_t lh[256];
    __shared__ unsigned long long sh64[2];
    __shared__ unsigned sh32[4];
    const int j = blockIdx.x, plane = blockIdx.y;
    pq_bracket b = br[(size_t)plane * nq + j];
    {  // the per-block counts of the classify pass, added up (see there)
        __shared__ unsigned s_sum[3];
        if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
        __syncthreads();
        unsigned c0 = 0, c1 = 0, c2 = 0;
        for (int i = threadIdx.x; i < nparts; i += 1024) {
            const uint4 p = partial[((size_t)plane * nq + j) * nparts + i];
            c0 += p.x, c1 += p.y, c2 += p.z;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c0 += __shfl_xor(c0, o);
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
        }
        if ((threadIdx.x & 63) == 0 && (c0 | c1 | c2)) {
            atomicAdd(&s_sum[0], c0);
            atomicAdd(&s_sum[1], c1);
            atomicAdd(&s_sum[2], c2);
        }
        __syncthreads();
        b.below = s_sum[0];
        b.eq_a = s_sum[1];
        b.eq_b = s_sum[2];
    }
    const unsigned long long* lst = lists + ((size_t)plane * nq + j) * cap;
    const double* src = in + (size_t)plane * n;
    const unsigned long long rlo = (unsigned long long)reqs[j].lo, rhi = (unsigned long long)reqs[j].hi;
    const unsigned long long e0 = b.below, e1 = e0 + b.eq_a, e2 = e1 + b.n_in, e3 = e2 + b.eq_b;
    unsigned long long keys[2];
    bool have[2] = {false, false};
    // both ranks inside the list: one list selection serves both
    if (b.n_in <= cap && rlo >= e1 && rhi < e2) {
        unsigned long long out2[2];
        pq_list_select(lst, (size_t)b.n_in, rlo - e1, rhi != rlo, b.a, b.b, b.has_a && b.has_b, B, lh, sh64, sh32, out2);
        keys[0] = out2[0];
        keys[1] = out2[1];
        have[0] = have[1] = true;
    }
    for (int w = 0; w < 2; ++w) {
        if (have[w]) continue;
        const unsigned long long r = w ? rhi : rlo;
        unsigned long long key;
        if (w == 1 && rhi == rlo) {
            key = keys[0];
        } else if (b.has_a && r >= e0 && r < e1) {
            key = b.a;
        } else if (b.n_in <= cap && r >= e1 && r < e2) {
            unsigned long long out2[2];
            pq_list_select(lst, (size_t)b.n_in, r - e1, 0, b.a, b.b, b.has_a && b.has_b, B, lh, sh64, sh32, out2);
            key = out2[0];
        } else if (b.has_b && r >= e2 && r < e3) {
            key = b.b;
        } else {
            // outside the bracket (or the list overflowed): exact select over the whole plane
            key = pq_block_select([&](size_t i) { return amt_f64_key(src[i]); }, n, r, 0ull, 0, lh, &sh64[0], &sh64[1]);
        }
        keys[w] = key;
    }
    if (threadIdx.x == 0) {
        for (int w = 0; w < 2; ++w) {
            res[(size_t)plane * 2 * nq + 2 * j + w].prefix = keys[w];
            res[(size_t)plane * 2 * nq + 2 * j + w].rank = 0;
        }
    }
}

static int percentile_f64_radix(amt_ctx* ctx, const double* in, const rank_req* reqs, int nq, double* out_dev, int nplanes,
                                size_t n);

extern "C" int amt_percentile_f64(amt_ctx* ctx, const double* in, const double* q_host, int nq, double* out_dev,
                                  int nplanes, size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && q_host && out_dev && nplanes >= 0, "percentile_f64: bad arguments");
    AMT_REQUIRE(nq >= 1 && nq <= 8, "percentile_f64: nq %d out of range 1..8", nq);
    AMT_REQUIRE(n >= 1, "percentile_f64: empty image");
    for (int i = 0; i < nq; ++i)
        AMT_REQUIRE(q_host[i] >= 0.0 && q_host[i] <= 100.0, "Percentiles must be in the range [0, 100]");
    if (nplanes == 0) return AMT_OK;
    rank_req reqs[8];
    make_rank_reqs(q_host, nq, n, reqs);
    if (n < PQ_MIN_N) return percentile_f64_radix(ctx, in, reqs, nq, out_dev, nplanes, n);
    const size_t cap = n / 8 + 4096;
    const size_t nbr = (size_t)nplanes * nq;
    // blocks per plane of the classify pass: enough of them to fill the chip over all planes, few enough that their list
    // cursors do not queue on the bracket's cache line
    static const int pq_grid_env = getenv("AMT_PQ_GRID") ? atoi(getenv("AMT_PQ_GRID")) : 0;  // A/B
    int want = pq_grid_env > 0 ? pq_grid_env : 4096 / nplanes;
    want = want < 64 ? 64 : (want > 256 && pq_grid_env <= 0 ? 256 : want);
    const unsigned gparts = amt_grid_for(n, 256 * PQ_VPT, (unsigned)want);
    AMT_TRY(amt_arena_begin(ctx, amt_align(sizeof(rank_req) * 8) + amt_align(sizeof(pq_bracket) * nbr) +
                                     amt_align(nbr * cap * 8) + amt_align(sizeof(sel_state) * 2 * nbr) +
                                     amt_align(sizeof(uint4) * nbr * gparts)));
    rank_req* rd = arena_take_t<rank_req>(ctx, 8);
    pq_bracket* br = arena_take_t<pq_bracket>(ctx, nbr);
    unsigned long long* lists = arena_take_t<unsigned long long>(ctx, nbr * cap);
    sel_state* res = arena_take_t<sel_state>(ctx, 2 * nbr);
    uint4* partial = arena_take_t<uint4>(ctx, nbr * gparts);
    AMT_TRY(amt_param_upload(ctx, rd, reqs, sizeof(rank_req) * nq));
    hipLaunchKernelGGL(pq_sample_kernel, dim3(nplanes), dim3(1024), (size_t)PQ_M * 8, ctx->stream, in, rd, nq, br, n);
    AMT_LAUNCH_CHECK();
    for (int j0 = 0; j0 < nq; j0 += PQ_GROUP) {  // one full read per group of four percentiles
        const int nj = nq - j0 < PQ_GROUP ? nq - j0 : PQ_GROUP;
        const dim3 gc(gparts, nplanes);
        const size_t sm = (size_t)nj * PQ_STAGE * 8;
        const bool vec2 = n % 2 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
#define AMT_PQ_CLASSIFY(NQ_)                                                                                          \
    if (vec2)                                                                                                         \
        hipLaunchKernelGGL((pq_classify_kernel<NQ_, true>), gc, dim3(256), sm, ctx->stream, in, br, nq, j0, lists, cap, n, partial); \
    else                                                                                                              \
        hipLaunchKernelGGL((pq_classify_kernel<NQ_, false>), gc, dim3(256), sm, ctx->stream, in, br, nq, j0, lists, cap, n, partial);
        switch (nj) {
            case 1: AMT_PQ_CLASSIFY(1) break;
            case 2: AMT_PQ_CLASSIFY(2) break;
            case 3: AMT_PQ_CLASSIFY(3) break;
            default: AMT_PQ_CLASSIFY(4) break;
        }
#undef AMT_PQ_CLASSIFY
        AMT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(pq_resolve_kernel, dim3(nq, nplanes), dim3(1024), (size_t)PQ_BUF * 8, ctx->stream, in, rd, br, nq, lists,
                       cap, res, n, partial, (int)gparts);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_finish_kernel, dim3((nplanes * nq + 63) / 64), dim3(64), 0, ctx->stream, res, rd, nq, nplanes,
                       out_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// the 8-bit MSB radix select over the whole plane (four full passes + candidate lists): planes below PQ_MIN_N samples
static int percentile_f64_radix(amt_ctx* ctx, const double* in, const rank_req* reqs, int nq, double* out_dev, int nplanes,
                                size_t n) {
    const int nslots = 2 * nq;
    const int total = nplanes * nslots;
    // candidate lists: keys matching the first 24 resolved bits (a 1/4096 slice of the exponent / mantissa space);
    // n / 16 entries per slot cover everything but massive ties, which fall back to full scans
    const size_t cap = n / 16 + 1024;
    size_t need = amt_align(sizeof(rank_req) * 8) + amt_align(sizeof(sel_state) * 2 * total) +
                  amt_align((size_t)8 * total * 256 * 4) + amt_align((size_t)total * cap * 8) +
                  amt_align((size_t)total * 4) + amt_align((size_t)nplanes * 4);
    AMT_TRY(amt_arena_begin(ctx, need));
    rank_req* rd = arena_take_t<rank_req>(ctx, 8);
    sel_state* st = arena_take_t<sel_state>(ctx, 2 * (size_t)total);
    uint32_t* counts = arena_take_t<uint32_t>(ctx, (size_t)8 * total * 256);
    unsigned long long* cand = arena_take_t<unsigned long long>(ctx, (size_t)total * cap);
    unsigned* ncand = arena_take_t<unsigned>(ctx, (size_t)total);
    int* overflow = arena_take_t<int>(ctx, nplanes);
    AMT_TRY(amt_param_upload(ctx, rd, reqs, sizeof(rank_req) * nq));
    hipLaunchKernelGGL(sel_init_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, st, rd, nq, nplanes);
    AMT_LAUNCH_CHECK();
    size_t smem = (size_t)nslots * 256 * 4 + (size_t)nslots * 8;
    // counts of the four full passes live side by side (one memset); the state ping-pongs between two buffers
    AMT_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)8 * total * 256 * 4, ctx->stream));
    AMT_HIP_CHECK(hipMemsetAsync(ncand, 0, (size_t)total * 4, ctx->stream));
    sel_state* cur = st;
    sel_state* nxt = st + total;
    for (int pass = 0; pass <= SEL_LIST_PASS; ++pass) {
        uint32_t* cnt = counts + (size_t)pass * total * 256;
        dim3 grid(amt_grid_for(n, 256 * 16, 512), nplanes);
        hipLaunchKernelGGL(sel_count_kernel, grid, dim3(256), smem, ctx->stream, in, cur, cnt, nslots, pass, n,
                           pass == SEL_LIST_PASS ? cand : (unsigned long long*)nullptr, ncand, cap, (const int*)nullptr);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(sel_pick_kernel, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, cur, nxt, cnt, nslots, pass,
                           nplanes, (const int*)nullptr);
        AMT_LAUNCH_CHECK();
        sel_state* t = cur;
        cur = nxt;
        nxt = t;
    }
    hipLaunchKernelGGL(sel_overflow_kernel, dim3((nplanes + 63) / 64), dim3(64), 0, ctx->stream, ncand, overflow, nslots,
                       nplanes, (unsigned)cap);
    AMT_LAUNCH_CHECK();
    // planes whose lists overflowed (massive ties): four more full passes, every block of the chip on them -- an even
    // number, so the state ends in the buffer the list kernel works on; other planes' blocks exit at once
    for (int pass = SEL_LIST_PASS + 1; pass < 8; ++pass) {
        uint32_t* cnt = counts + (size_t)pass * total * 256;
        dim3 grid(amt_grid_for(n, 256 * 16, 512), nplanes);
        hipLaunchKernelGGL(sel_count_kernel, grid, dim3(256), smem, ctx->stream, in, cur, cnt, nslots, pass, n,
                           (unsigned long long*)nullptr, ncand, cap, (const int*)overflow);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(sel_pick_kernel, dim3((total + 63) / 64), dim3(64), 0, ctx->stream, cur, nxt, cnt, nslots, pass,
                           nplanes, (const int*)overflow);
        AMT_LAUNCH_CHECK();
        sel_state* t = cur;
        cur = nxt;
        nxt = t;
    }
    hipLaunchKernelGGL(sel_list_passes_kernel, dim3(nslots, nplanes), dim3(256), 0, ctx->stream, in, cur, nslots, n, cand,
                       ncand, cap, (const int*)overflow);
    AMT_LAUNCH_CHECK();
    st = cur;
    hipLaunchKernelGGL(sel_finish_kernel, dim3((nplanes * nq + 63) / 64), dim3(64), 0, ctx->stream, st, rd, nq, nplanes,
                       out_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ------------------------------------------------------------------------------------------------
// Deterministic masked sums for threshold_mean / threshold_li on float images
// (SK/filters/thresholding.py:830, :642-707): per plane {sum(x <= t), count(x <= t), sum(x > t), count(x > t)}.
// Fixed reduction tree (per-thread strided sum -> wave shuffle -> block -> one ordered pass over the
// block partials), so repeated runs give identical bits; numpy's pairwise order differs in the last ulps.
// ------------------------------------------------------------------------------------------------
constexpr int MS_BLOCKS = 256;

__global__ void __launch_bounds__(256) masked_sums_partial_kernel(const double* __restrict__ in,
                                                                  const double* __restrict__ thr,
                                                                  double* __restrict__ partial, size_t n) {
    const int plane = blockIdx.y;
    const double t = thr[plane];
    const double* src = in + (size_t)plane * n;
    double s_le = 0.0, s_gt = 0.0, c_le = 0.0, c_gt = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)MS_BLOCKS * 256) {
        double v = src[i];
        if (v > t) {
            s_gt += v;
            c_gt += 1.0;
        } else {
            s_le += v;
            c_le += 1.0;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s_le += __shfl_xor(s_le, off);
        s_gt += __shfl_xor(s_gt, off);
        c_le += __shfl_xor(c_le, off);
        c_gt += __shfl_xor(c_gt, off);
    }
    __shared__ double sm[4][4];
    if ((threadIdx.x & 63) == 0) {
        int w = threadIdx.x >> 6;
        sm[w][0] = s_le;
        sm[w][1] = c_le;
        sm[w][2] = s_gt;
        sm[w][3] = c_gt;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
        partial[((size_t)plane * MS_BLOCKS + blockIdx.x) * 4 + threadIdx.x] = v;
    }
}

__global__ void masked_sums_final_kernel(const double* __restrict__ partial, double* __restrict__ out) {
    const int plane = blockIdx.x;
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int b = 0; b < MS_BLOCKS; ++b) v += partial[((size_t)plane * MS_BLOCKS + b) * 4 + threadIdx.x];
        out[plane * 4 + threadIdx.x] = v;
    }
}

extern "C" int amt_masked_sums_f64(amt_ctx* ctx, const double* in, const double* thr_dev, double* out_dev, int nplanes,
                                   size_t n) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && thr_dev && out_dev && nplanes >= 0, "masked_sums_f64: bad arguments");
    if (nplanes == 0) return AMT_OK;
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * MS_BLOCKS * 4 * sizeof(double))));
    double* partial = arena_take_t<double>(ctx, (size_t)nplanes * MS_BLOCKS * 4);
    hipLaunchKernelGGL(masked_sums_partial_kernel, dim3(MS_BLOCKS, nplanes), dim3(256), 0, ctx->stream, in, thr_dev,
                       partial, n);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(masked_sums_final_kernel, dim3(nplanes), dim3(64), 0, ctx->stream, partial, out_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
