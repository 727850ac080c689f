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

#include <type_traits>

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

// ---- run-based, tile-local union-find -------------------------------------------------------------
// Runs: maximal horizontal stretches of equal non-zero values inside one 64-pixel wave segment (found with
// one ballot).  Every pixel initially points at the first pixel of its run, so whole runs are trees of depth 1
// and only RUN PAIRS have to be stitched.
//   ccl_tile_kernel    one block per 64 x 64 tile: each wave loads 16 rows (all in flight), the tile's
//                      union-find lives in LDS (unions at LDS latency, no global atomics), and every pixel
//                      is written ONCE, already pointing at the tile-local root (as a global flat index);
//   ccl_border_kernel  stitches tiles in HBM: the row pair across each tile boundary and the column pair
//                      across each 64-pixel segment boundary.
// Vertical rule: p joins its north pixel q only when p or q starts a run -- otherwise (p-1, q-1) is the same
// pair of runs and is handled further left.  For 8-connectivity the diagonals matter only when north differs:
// NW unless west matches (then west reaches NW as its own north), NE unless east matches.
constexpr int STRIP_R = 16;            // rows per wave
constexpr int TILE_R = 4 * STRIP_R;    // rows per block
// pixel values travel through shuffles / compares in the narrowest type that still has room for a sentinel
// ("outside the image", equal to no pixel value): 32 bits for uint8 masks, 64 bits for int32 label images
template <typename T>
struct ccl_wide;
template <>
struct ccl_wide<uint8_t> {
    typedef int type;
    static constexpr int NOVAL = -1;
};
template <>
struct ccl_wide<int32_t> {
    typedef long long type;
    static constexpr long long NOVAL = -(1ll << 40);
};

__device__ __forceinline__ int lds_find(int* S, int a) {
    int p = __hip_atomic_load(&S[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != a) {
        a = p;
        p = __hip_atomic_load(&S[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return a;
}

__device__ __forceinline__ void lds_union(int* S, int a, int b) {
    while (true) {
        a = lds_find(S, a);
        b = lds_find(S, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&S[a], b);
        if (old == a) return;
        a = old;
    }
}

// stitch row y (values v, index p) to row y-1 (values up) inside one 64-lane segment; executed by all lanes.
// LDS = true: L is the tile's LDS array and `pitch` = 64; otherwise L is the plane in HBM and pitch = W.
template <bool CONN8, bool LDS, typename V>
__device__ __forceinline__ void ccl_stitch_rows(int* L, int p, int pitch, int lane, V v, V up, V noval) {
    const V w = amt_lane_left(v), upw = amt_lane_left(up);
    const bool head = lane == 0 || w != v;
    const bool up_head = lane == 0 || upw != up;
    V e = 0, upe = 0;
    if (CONN8) {
        e = amt_lane_right(v);
        upe = amt_lane_right(up);
    }
    if (v == 0 || v == noval) return;
    int q = -1;
    if (up == v) {
        if (head || up_head) q = p - pitch;
    } else if (CONN8) {
        if (lane > 0 && w != v && upw == v) q = p - pitch - 1;
        if (lane < 63 && upe == v && e != v) {
            if (q >= 0) {  // both diagonals: two unions
                if (LDS) lds_union(L, p, q); else uf_union(L, p, q);
            }
            q = p - pitch + 1;
        }
    }
    if (q >= 0) {
        if (LDS) lds_union(L, p, q); else uf_union(L, p, q);
    }
}

// rootlist / nroots (nullable): every tile-local root (global flat index) is appended to the list of its TILE ROW,
// rootlist[(plane * tile_rows + tile_row) * cap ...] with cap = TILE_R * W entries (order arbitrary; one counter per
// tile row keeps the reserving atomics of a plane off a single address) -- callers that only need roots compressed
// walk these lists instead of the whole plane.
template <typename T, bool CONN8>
__device__ __forceinline__ void ccl_tile_do(const T* __restrict__ in, int* __restrict__ Lall, int H, int W,
                                            int* __restrict__ rootlist, int* __restrict__ nroots, size_t cap, int bx, int by,
                                            int bz, int ntr) {
    __shared__ __attribute__((aligned(16))) int S[TILE_R * 64];
    typedef typename ccl_wide<T>::type V;
    constexpr V NOVAL = ccl_wide<T>::NOVAL;
    __shared__ V vlast[4][64];
    const size_t n = (size_t)H * W;
    const T* img = in + (size_t)bz * n;
    int* L = Lall + (size_t)bz * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = bx * 64, ty0 = by * TILE_R;
    const int x = x0 + lane;
    const int xc = x < W ? x : W - 1;
    const int r0 = wave * STRIP_R;  // first tile row of this wave
    V v[STRIP_R];
    unsigned fgrows = 0;
    {
        // unconditional loads from clamped coordinates, widened / invalidated afterwards: a conversion next to a
        // conditional load would put a wait behind every one of them
        T raw[STRIP_R];
#pragma unroll
        for (int k = 0; k < STRIP_R; ++k) {
            const int y = ty0 + r0 + k;
            raw[k] = img[(size_t)(y < H ? y : H - 1) * W + xc];
        }
#pragma unroll
        for (int k = 0; k < STRIP_R; ++k) v[k] = (x < W && ty0 + r0 + k < H) ? (V)raw[k] : NOVAL;
    }
#pragma unroll
    for (int k = 0; k < STRIP_R; ++k) {
        // uniform: a row segment without foreground takes no part in anything below -- its slots are marked
        // "background" right away (what the write-out expects) and every later phase skips it
        if (__ballot(v[k] != 0 && v[k] != NOVAL) == 0ull) {
            S[(r0 + k) * 64 + lane] = -1;
            continue;
        }
        fgrows |= 1u << k;
        const V left = amt_lane_left(v[k]);
        const bool head = (lane == 0) || (left != v[k]);
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = heads & ((2ull << lane) - 1ull);
        const int start_lane = 63 - __clzll((long long)upto);
        S[(r0 + k) * 64 + lane] = (r0 + k) * 64 + start_lane;
    }
    vlast[wave][lane] = v[STRIP_R - 1];
    __syncthreads();
    // a row without foreground starts no union: most rows of a sparse mask are skipped here
#pragma unroll
    for (int k = 1; k < STRIP_R; ++k)
        if ((fgrows >> k) & 1u) ccl_stitch_rows<CONN8, true>(S, (r0 + k) * 64 + lane, 64, lane, v[k], v[k - 1], NOVAL);
    if (wave > 0 && (fgrows & 1u)) ccl_stitch_rows<CONN8, true>(S, r0 * 64 + lane, 64, lane, v[0], vlast[wave - 1][lane], NOVAL);
    __syncthreads();
    // Phase A: every pixel finds its tile root and writes it back (path compression: phase B then needs one LDS
    // read); each wave counts its roots with ballots.  The block reserves its slice of the list with ONE atomic.
    __shared__ int s_wroots[4], s_base;
    int wroots = 0;
#pragma unroll
    for (int k = 0; k < STRIP_R; ++k) {
        if (!((fgrows >> k) & 1u)) continue;  // uniform: nothing but background, already marked
        const int y = ty0 + r0 + k;
        const int own = (r0 + k) * 64 + lane;
        bool is_root = false;
        if (x < W && y < H && v[k] != 0) {
            const int r = lds_find(S, own);
            if (r != own) S[own] = r;
            is_root = r == own;
        } else {
            S[own] = -1;  // background / outside: no find ever walks through these slots
        }
        if (rootlist) wroots += __popcll(__ballot(is_root));
    }
    if (rootlist) {
        if (lane == 0) s_wroots[wave] = wroots;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_wroots[0] + s_wroots[1] + s_wroots[2] + s_wroots[3];
            s_base = tot ? atomicAdd(&nroots[bz * ntr + by], tot) : 0;
        }
    }
    __syncthreads();
    // Phase B: write the labels (global index of the tile root) and append the roots.  A lane now owns FOUR
    // consecutive pixels of a row of its wave's strip (one 16-byte LDS read, one 16-byte store); the roots are
    // appended in (wave, row group, pixel-of-four, lane) order, which is as arbitrary as any other.
    int run = 0;
    if (rootlist) {
        run = s_base;
        for (int w2 = 0; w2 < wave; ++w2) run += s_wroots[w2];
    }
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = x0 + c4;
#pragma unroll
    for (int j = 0; j < STRIP_R / 4; ++j) {
        const int row = r0 + rsub + 4 * j;
        const int y = ty0 + row;
        const int own0 = row * 64 + c4;
        const int4 r4 = *reinterpret_cast<const int4*>(&S[own0]);
        const int rr[4] = {r4.x, r4.y, r4.z, r4.w};
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rr[i] >= 0 ? (ty0 + (rr[i] >> 6)) * W + x0 + (rr[i] & 63) : -1;
        if (y < H) {
            const size_t idx = (size_t)y * W + xg;
            if (xg + 3 < W && (((size_t)bz * n + idx) & 3) == 0) {
                *reinterpret_cast<int4*>(L + idx) = make_int4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xg + i < W) L[idx + i] = o[i];
            }
        }
        if (rootlist) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool is_root = rr[i] == own0 + i;  // only foreground pixels inside the image point at themselves
                const unsigned long long m = __ballot(is_root);
                if (is_root) {
                    const size_t pos = (size_t)run + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) rootlist[((size_t)bz * ntr + by) * cap + pos] = o[i];
                }
                run += __popcll(m);
            }
        }
    }
}

template <typename T, bool CONN8>
__global__ void __launch_bounds__(256) ccl_tile_kernel(const T* __restrict__ in, int* __restrict__ Lall, int H, int W,
                                                       int* __restrict__ rootlist, int* __restrict__ nroots,
                                                       size_t cap) {
    ccl_tile_do<T, CONN8>(in, Lall, H, W, rootlist, nroots, cap, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// the fallback behind ccl_tile_bits_kernel: a few hundred workgroups that leave at once unless that kernel gave up
// (*only_if != 0), and then walk over all tiles
template <typename T, bool CONN8>
__global__ void __launch_bounds__(256) ccl_tile_fallback_kernel(const T* __restrict__ in, int* __restrict__ Lall, int H, int W,
                                                                int* __restrict__ rootlist, int* __restrict__ nroots,
                                                                size_t cap, const int* __restrict__ only_if, int ntx,
                                                                int ntr, int nplanes) {
    if (!*only_if) return;
    const int total = ntx * ntr * nplanes;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int bx = t % ntx, by = (t / ntx) % ntr, bz = t / (ntx * ntr);
        ccl_tile_do<T, CONN8>(in, Lall, H, W, rootlist, nroots, cap, bx, by, bz, ntr);
        __syncthreads();
    }
}

// ---- the same tile labelling for 0 / 1 byte masks, bit-parallel (round 3) --------------------------------------------
// ccl_tile_kernel spends ~78 lane-operations per pixel (four waves per tile, a lane per pixel and row): the pass was
// bound by vector-instruction issue, not by LDS or HBM.  For a MASK the rows of a 64 x 64 tile are 64-bit words, and
// everything that is not the final write-out can be done on words by ONE wave with a lane per ROW:
//   * the lane loads its row's 64 bytes (four 16-byte loads) and squeezes them into a word (a multiply per four bytes);
//   * run heads = w & ~(w << 1); every RUN is an element of the tile's LDS union-find (8 KB: a row has at most 32 runs;
//     ids in raster order of the runs' first pixels, so the root is again the component's first pixel);
//   * the lane stitches its runs to the row above (the neighbour lane's word): the runs of `above` that meet a run --
//     widened by one column for 8-connectivity -- are the 1-segments of (run & above), one union each;
//   * heads are compressed, and the write-out maps lanes to four consecutive pixels of a row as before; a pixel finds
//     its run's head by a count-leading-zeros on its row's word.
// Values other than 0 / 1 (a uint8 image labelled "by equal value") set *multi, the caller resets the root lists and
// ccl_tile_kernel redoes the planes.  Requires W % 16 == 0 and 16-byte aligned rows (the host checks).
__device__ __forceinline__ int ccl_run_start(unsigned long long w, int x) {
    const unsigned long long z = ~w & ((1ull << x) - 1ull);  // zeros below x
    return z ? 64 - __clzll((long long)z) : 0;
}

// union-find over RUNS: entry of run id (row << 5 | ordinal of the run in its row; a row has at most 32 runs) =
// parent id << 6 | the parent run's first column -- ids grow in raster order of the runs' first pixels, so "smaller entry
// wins" is "smaller id wins", and a find returns the root's position along with its id
__device__ __forceinline__ int ccl_rfind(int* S, int id) {
    int e = __hip_atomic_load(&S[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((e >> 6) != id) {
        id = e >> 6;
        e = __hip_atomic_load(&S[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return e;
}
__device__ __forceinline__ void ccl_runion(int* S, int a, int b) {
    while (true) {
        int ea = ccl_rfind(S, a), eb = ccl_rfind(S, b);
        if (ea == eb) return;
        if (ea < eb) {
            const int t = ea;
            ea = eb;
            eb = t;
        }
        const int old = atomicMin(&S[ea >> 6], eb);  // hang the later root below the earlier one
        if (old == ea) return;
        a = old >> 6;  // somebody moved it meanwhile: retry from there
        b = eb >> 6;
    }
}

// the lane's row of a 64 x 64 tile as a word (bit i = byte i is 1); `other` collects bits of bytes that are not 0 / 1
// NONZERO: bit i = byte i is not 0 (a watershed's mask is a truth value; `other` stays 0)
template <bool NONZERO = false>
__device__ __forceinline__ unsigned long long ccl_bits_load_row(const uint8_t* __restrict__ img, int H, int W, int x0, int y,
                                                                unsigned& other) {
    unsigned long long w = 0;
    other = 0;
    const uint8_t* rowp = img + (size_t)(y < H ? y : H - 1) * W;
    uint4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xs = x0 + 16 * j;
        q[j] = *reinterpret_cast<const uint4*>(rowp + (xs < W ? xs : 0));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
        unsigned sixteen = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned one = v[k] & 0x01010101u;
            if (NONZERO) one = ((((v[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v[k]) >> 7) & 0x01010101u;
            else other |= v[k] & 0xFEFEFEFEu;
            sixteen |= ((one * 0x01020408u) >> 24) << (4 * k);  // byte i -> bit i
        }
        if (x0 + 16 * j < W) w |= (unsigned long long)(sixteen & 0xFFFFu) << (16 * j);
    }
    if (y >= H) {
        w = 0;
        other = 0;
    }
    return w;
}

// the tile's union-find over runs (a lane per row): own parents, stitched to the row above, compressed -- afterwards
// S[run] = the entry of the run's tile root.  Returns the number of root runs of the lane's row.
template <bool CONN8>
__device__ __forceinline__ int ccl_bits_unionfind(int* S, unsigned long long w, unsigned long long heads, int lane) {
    {
        int j = 0;
        for (unsigned long long h = heads; h; h &= h - 1, ++j) {
            const int b = __ffsll((long long)h) - 1;
            S[lane * 32 + j] = ((lane * 32 + j) << 6) | b;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    {
        unsigned long long above = __shfl_up(w, 1);
        if (lane == 0) above = 0;
        const unsigned long long aheads = above & ~(above << 1);
        int j = 0;
        for (unsigned long long h = heads; h && above; h &= h - 1, ++j) {
            const int b = __ffsll((long long)h) - 1;
            const unsigned long long t = w >> b;
            const int len = ~t ? __ffsll((long long)~t) - 1 : 64;  // trailing ones: the run's length
            unsigned long long rm = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << b;
            if (CONN8) rm |= (rm << 1) | (rm >> 1);
            const unsigned long long ov = rm & above;
            for (unsigned long long oh = ov & ~(ov << 1); oh; oh &= oh - 1) {
                const int c = __ffsll((long long)oh) - 1;  // a pixel of the run above: its ordinal = heads at or below c
                const int aj = __popcll(aheads & ((2ull << c) - 1ull)) - 1;
                ccl_runion(S, lane * 32 + j, (lane - 1) * 32 + aj);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int nroot = 0;
    {
        int j = 0;
        for (unsigned long long h = heads; h; h &= h - 1, ++j) {
            const int own = lane * 32 + j;
            const int e = ccl_rfind(S, own);
            if ((e >> 6) != own) S[own] = e;
            nroot += (e >> 6) == own;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    return nroot;
}

template <bool CONN8>
__global__ void __launch_bounds__(64) ccl_tile_bits_kernel(const uint8_t* __restrict__ in, int* __restrict__ Lall, int H,
                                                           int W, int* __restrict__ rootlist, int* __restrict__ nroots,
                                                           size_t cap, int* __restrict__ multi,
                                                           unsigned long long* __restrict__ colbits) {
    __shared__ int S[64 * 32];  // 8 KB: eighteen tiles per CU
    __shared__ unsigned long long bits[64];
    const size_t n = (size_t)H * W;
    const uint8_t* img = in + (size_t)blockIdx.z * n;
    int* L = Lall + (size_t)blockIdx.z * n;
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, ty0 = blockIdx.y * 64;
    unsigned other;
    const unsigned long long w = ccl_bits_load_row(img, H, W, x0, ty0 + lane, other);
    if (__ballot(other != 0) && lane == 0) atomicOr(multi, 1);
    bits[lane] = w;
    {
        // the tile's first and last column, a bit per row: all the seam pass needs to stitch this tile to its left and
        // right neighbours (reading those two columns from the image costs a 64-byte line per PIXEL: 8 MB per plane)
        const unsigned long long lc = __ballot((w & 1ull) != 0), rc = __ballot((w >> 63) != 0);
        if (lane == 0) {
            unsigned long long* cb = colbits + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2;
            cb[0] = lc;
            cb[1] = rc;
        }
    }
    const unsigned long long heads = w & ~(w << 1);
    const int nroot = ccl_bits_unionfind<CONN8>(S, w, heads, lane);
    // ---- reserve the tile's slice of its tile row's list ----
    int run = 0;
    if (rootlist) {
        int tot = nroot;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
        int base = 0;
        if (lane == 0 && tot) base = atomicAdd(&nroots[blockIdx.z * gridDim.y + blockIdx.y], tot);
        run = __builtin_amdgcn_readfirstlane(base);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- write-out: a lane owns four consecutive pixels of a row ----
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = x0 + c4;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int row = rsub + 4 * j;
        const int y = ty0 + row;
        const unsigned long long ww = bits[row];
        const unsigned long long hw = ww & ~(ww << 1);
        int o[4];
        bool isroot[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = c4 + i;
            o[i] = -1;
            isroot[i] = false;
            if ((ww >> x) & 1ull) {
                const int id = row * 32 + __popcll(hw & ((2ull << x) - 1ull)) - 1;  // the pixel's run
                const int e = S[id];                                                // runs point straight at the root run now
                o[i] = (ty0 + (e >> 11)) * W + x0 + (e & 63);
                isroot[i] = (e >> 6) == id && (e & 63) == x;                        // the first pixel of a root run
            }
        }
        if (y < H && xg < W) {  // W % 16 == 0: the four pixels are inside the image together
            const size_t idx = (size_t)y * W + xg;
            *reinterpret_cast<int4*>(L + idx) = make_int4(o[0], o[1], o[2], o[3]);
        }
        if (rootlist) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned long long m = __ballot(isroot[i]);
                if (isroot[i]) {
                    const size_t pos = (size_t)run + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) rootlist[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * cap + pos] = o[i];
                }
                run += __popcll(m);
            }
        }
    }
}

// ---- run tables instead of a parent plane (round 3) -------------------------------------------------------------------
// amt_label on a 0 / 1 mask used to move 13 bytes per pixel: the tile pass wrote an int32 parent per pixel (4), the
// numbering pass read it back (4) and wrote the labels (4).  A tile's labelling IS its run table, so the tile pass now
// leaves only that: the 64 row words (512 B), one 16-bit entry per run = (row << 6 | column) of the run's tile root
// (runs in raster order; ~100 per tile on nuclei masks, at most 2,048), the run count, and L[p] = p at the tile roots
// themselves -- the union-find of the seams and the raster numbering only ever touch tile roots.  The seams are stitched
// from the row words (a wave per tile boundary), and the last pass expands (row words, run table, rank of every tile
// root) into labels: 1 + ~0.2 bytes read and 4 written per pixel.
// (RT_CAP, ccl_wave_incl_scan, ccl_rt_root: amt_internal.h -- the watershed reads the tables too)

// WS: the watershed's variant -- the mask is a truth value (no fallback, multi unused) and the rows' run offsets are
// stored for random look-ups
template <bool CONN8, bool WS = false>
__global__ void __launch_bounds__(64) ccl_tile_runs_kernel(const uint8_t* __restrict__ in, int* __restrict__ Lall, int H,
                                                           int W, int* __restrict__ rootlist, int* __restrict__ nroots,
                                                           size_t cap, int* __restrict__ multi,
                                                           unsigned long long* __restrict__ tbits,
                                                           unsigned short* __restrict__ rtab, int* __restrict__ nruns,
                                                           unsigned short* __restrict__ roff) {
    __shared__ int S[64 * 32];
    const size_t n = (size_t)H * W;
    const uint8_t* img = in + (size_t)blockIdx.z * n;
    int* L = Lall + (size_t)blockIdx.z * n;
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, ty0 = blockIdx.y * 64;
    const size_t tile = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    unsigned other;
    const unsigned long long w = ccl_bits_load_row<WS>(img, H, W, x0, ty0 + lane, other);
    if (!WS && __ballot(other != 0) && lane == 0) atomicOr(multi, 1);
    tbits[tile * 64 + lane] = w;
    const unsigned long long heads = w & ~(w << 1);
    const int nroot = ccl_bits_unionfind<CONN8>(S, w, heads, lane);
    // compact run ordinals (raster order) and the tile's slice of its tile row's root list
    const int cnt = __popcll(heads);
    const int incl = ccl_wave_incl_scan(cnt, lane);
    if (WS && roff) roff[tile * 64 + lane] = (unsigned short)(incl - cnt);
    const int rincl = ccl_wave_incl_scan(nroot, lane);
    const int tot = __shfl(rincl, 63);
    int base = 0;
    if (lane == 0 && tot) base = atomicAdd(&nroots[blockIdx.z * gridDim.y + blockIdx.y], tot);
    size_t pos = (size_t)__builtin_amdgcn_readfirstlane(base) + (rincl - nroot);
    unsigned short* rt = rtab + tile * RT_CAP + (incl - cnt);
    int j = 0;
    for (unsigned long long h = heads; h; h &= h - 1, ++j) {
        const int own = lane * 32 + j;
        const int e = S[own];  // the root run's entry: id << 6 | first column, id = row << 5 | ordinal in the row
        rt[j] = (unsigned short)(((e >> 11) << 6) | (e & 63));
        if ((e >> 6) == own) {
            const int pix = (ty0 + lane) * W + x0 + (e & 63);
            L[pix] = pix;
            if (pos < cap) rootlist[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * cap + pos] = pix;
            ++pos;
        }
    }
    if (lane == 63) nruns[tile] = incl;
}

// The seams from the tiles' row words: a wave per tile boundary.  jobs [0, segs * (trows - 1)): the row pair across a
// horizontal boundary, a lane per column; the rest: the column pair across a vertical boundary, a lane per row (with
// the two diagonal pairs at the corner where four tiles meet).  A union is skipped where a neighbouring lane's union
// implies it (same run pair, or pixels that are vertical / horizontal neighbours inside their own tiles).
template <bool CONN8>
__global__ void __launch_bounds__(256) ccl_seams_runs_kernel(const unsigned long long* __restrict__ tbits,
                                                             const unsigned short* __restrict__ rtab,
                                                             const int* __restrict__ nruns, int* __restrict__ Lall, int H,
                                                             int W, int segs, int trows, const int* __restrict__ multi) {
    if (multi && *multi) return;
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nrowjobs = segs * (trows - 1);
    if (job >= nrowjobs + (segs - 1) * trows) return;  // whole wave
    int* L = Lall + (size_t)blockIdx.z * H * W;
    const size_t t0 = (size_t)blockIdx.z * trows * segs;
    if (job < nrowjobs) {
        const int bx = job % segs, ty = job / segs + 1;
        const size_t A = t0 + (size_t)(ty - 1) * segs + bx, B = A + segs;
        const unsigned long long wa = tbits[A * 64 + 63], wb = tbits[B * 64];
        const unsigned long long reach = CONN8 ? (wa | (wa << 1) | (wa >> 1)) : wa;
        if (!(reach & wb)) return;
        const unsigned long long ha = wa & ~(wa << 1), hb = wb & ~(wb << 1);
        const int offA = nruns[A] - __popcll(ha);
        const unsigned long long upto = (2ull << lane) - 1ull;
        if (!((wa >> lane) & 1ull)) return;
        const int ka = offA + __popcll(ha & upto) - 1;
        int kb0 = -1, kb1 = -1;  // ordinals of the runs of B's first row to join
        if (!CONN8) {
            const unsigned long long ov = wa & wb;
            if (((ov & ~(ov << 1)) >> lane) & 1ull) kb0 = __popcll(hb & upto) - 1;
        } else {
            const bool al = lane > 0 && ((wa >> (lane - 1)) & 1ull);
            const bool bl = lane > 0 && ((wb >> (lane - 1)) & 1ull), bc = (wb >> lane) & 1ull;
            const bool br = lane < 63 && ((wb >> (lane + 1)) & 1ull);
            if (bc && !al) kb0 = __popcll(hb & upto) - 1;
            else if (bl && !bc && !al) kb0 = __popcll(hb & (upto >> 1)) - 1;
            if (br && !bc) kb1 = __popcll(hb & ((4ull << lane) - 1ull)) - 1;
        }
        if (kb0 < 0 && kb1 < 0) return;
        const int pa = ccl_rt_root(rtab, A, ka, ty - 1, bx, W);
        if (kb0 >= 0) uf_union(L, pa, ccl_rt_root(rtab, B, kb0, ty, bx, W));
        if (kb1 >= 0) uf_union(L, pa, ccl_rt_root(rtab, B, kb1, ty, bx, W));
    } else {
        const int cj = job - nrowjobs;
        const int bx = cj % (segs - 1) + 1, ty = cj / (segs - 1);
        const size_t A = t0 + (size_t)ty * segs + bx - 1, B = A + 1;
        const unsigned long long wA = tbits[A * 64 + lane], wB = tbits[B * 64 + lane];
        const bool a = (wA >> 63) & 1ull, b = wB & 1ull;
        const unsigned long long anya = __ballot(a), anyb = __ballot(b);
        // whole wave: every pair has a pixel of THIS tile row (4-connected: one on either side)
        if (CONN8 ? (!anya && !anyb) : (!anya || !anyb)) return;
        const int cA = __popcll(wA & ~(wA << 1)), cB = __popcll(wB & ~(wB << 1));
        const int kA = ccl_wave_incl_scan(cA, lane) - 1;       // the row's last run in A (it ends at column 63 if a)
        const int kB = ccl_wave_incl_scan(cB, lane) - cB;      // the row's first run in B (it starts at column 0 if b)
        // the row above: the lane below this one, or row 63 of the tiles above
        bool a_up = __shfl_up((int)a, 1), b_up = __shfl_up((int)b, 1);
        int kA_up = __shfl_up(kA, 1), kB_up = __shfl_up(kB, 1);
        size_t A_up = A, B_up = B;
        int ty_up = ty;
        if (lane == 0) {
            a_up = b_up = false;
            if (CONN8 && ty > 0) {
                A_up = A - segs;
                B_up = B - segs;
                ty_up = ty - 1;
                const unsigned long long ua = tbits[A_up * 64 + 63], ub = tbits[B_up * 64 + 63];
                a_up = (ua >> 63) & 1ull;
                b_up = ub & 1ull;
                kA_up = nruns[A_up] - 1;                               // the tile's very last run
                kB_up = nruns[B_up] - __popcll(ub & ~(ub << 1));       // the first run of its last row
            }
        }
        if (a && b && !(a_up && b_up)) uf_union(L, ccl_rt_root(rtab, A, kA, ty, bx - 1, W), ccl_rt_root(rtab, B, kB, ty, bx, W));
        if (CONN8) {
            if (b && a_up && !a && !b_up)  // north-west of b
                uf_union(L, ccl_rt_root(rtab, B, kB, ty, bx, W), ccl_rt_root(rtab, A_up, kA_up, ty_up, bx - 1, W));
            if (a && b_up && !b && !a_up)  // north-east of a
                uf_union(L, ccl_rt_root(rtab, A, kA, ty, bx - 1, W), ccl_rt_root(rtab, B_up, kB_up, ty_up, bx, W));
        }
    }
}

// labels from (row words, run table, T at the tile roots): a wave per tile, four tiles per workgroup.  T[tile root] is
// what the caller wants written for the root's component (the raster rank in amt_label).  Tiles with more than XR_CAP runs
// (noise) gather per pixel instead of through LDS.
constexpr int XR_CAP = 512;
__global__ void __launch_bounds__(256) ccl_expand_runs_kernel(const unsigned long long* __restrict__ tbits,
                                                              const unsigned short* __restrict__ rtab,
                                                              const int* __restrict__ nruns, const int* __restrict__ Tall,
                                                              int* __restrict__ outall, int H, int W, int segs, int trows,
                                                              int ntiles, const int* __restrict__ multi) {
    if (multi && *multi) return;
    __shared__ int lab_s[4][XR_CAP];
    __shared__ unsigned long long bits_s[4][64];
    __shared__ int off_s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wv;
    if (t >= ntiles) return;  // whole wave
    const int bx = t % segs, ty = (t / segs) % trows, plane = t / (segs * trows);
    const size_t n = (size_t)H * W;
    const int* T = Tall + (size_t)plane * n;
    int* out = outall + (size_t)plane * n;
    const int x0 = bx * 64, ty0 = ty * 64;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = x0 + c4;
    const int nr = nruns[t];
    if (nr == 0) {  // uniform: nothing but background
        if (xg < W) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int y = ty0 + rsub + 4 * j;
                if (y < H) *reinterpret_cast<int4*>(out + (size_t)y * W + xg) = make_int4(0, 0, 0, 0);
            }
        }
        return;
    }
    const unsigned long long w = tbits[(size_t)t * 64 + lane];
    const int cnt = __popcll(w & ~(w << 1));
    bits_s[wv][lane] = w;
    off_s[wv][lane] = ccl_wave_incl_scan(cnt, lane) - cnt;
    const bool fast = nr <= XR_CAP;
    if (fast)
        for (int k = lane; k < nr; k += 64) lab_s[wv][k] = T[ccl_rt_root(rtab, (size_t)t, k, ty, bx, W)];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (xg >= W) return;
    const int* lb = lab_s[wv];
    auto lab = [&](int k) -> int { return fast ? lb[k] : T[ccl_rt_root(rtab, (size_t)t, k, ty, bx, W)]; };
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int row = rsub + 4 * j;
        const int y = ty0 + row;
        const unsigned long long ww = bits_s[wv][row];
        const unsigned long long hw = ww & ~(ww << 1);
        const unsigned nib = (unsigned)(ww >> c4) & 15u, hnib = (unsigned)(hw >> c4) & 15u;
        int4 o = make_int4(0, 0, 0, 0);
        if (nib) {
            // the run of pixel c4 + i = heads at or before it, minus one
            const int k0 = off_s[wv][row] + __popcll(hw & ((1ull << c4) - 1ull)) - 1;
            // selects, not a block per pixel (scalar issue); a background pixel reads the run left of it or run 0
            const int ka = k0 + (int)(hnib & 1u), kb = k0 + __popc(hnib & 3u), kc = k0 + __popc(hnib & 7u), kd = k0 + __popc(hnib);
            const int la = lab(ka < 0 ? 0 : ka), lb2 = lab(kb < 0 ? 0 : kb), lc = lab(kc < 0 ? 0 : kc), ld = lab(kd < 0 ? 0 : kd);
            o.x = (nib & 1u) ? la : 0;
            o.y = (nib & 2u) ? lb2 : 0;
            o.z = (nib & 4u) ? lc : 0;
            o.w = (nib & 8u) ? ld : 0;
        }
        if (y < H) *reinterpret_cast<int4*>(out + (size_t)y * W + xg) = o;
    }
}

// the fallback's preparation: if the bit kernel met a byte other than 0 / 1, forget the roots it listed
__global__ void ccl_reset_lists_kernel(int* __restrict__ nroots, size_t nlist, const int* __restrict__ multi) {
    if (!*multi) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlist; i += (size_t)gridDim.x * blockDim.x) nroots[i] = 0;
}

// blockIdx.y selects the job: [0, nrow_jobs) = strip-boundary rows, the rest = segment-boundary columns
template <typename T, bool CONN8>
__global__ void __launch_bounds__(256) ccl_border_kernel(const T* __restrict__ in, int* __restrict__ Lall, int H, int W,
                                                         int nrow_blocks, const int* __restrict__ cols_only_if = nullptr,
                                                         const int* __restrict__ all_only_if = nullptr) {
    if (all_only_if && !*all_only_if) return;  // the run-table path stitched the seams (amt_label on a 0 / 1 mask)
    const size_t n = (size_t)H * W;
    const T* img = in + (size_t)blockIdx.z * n;
    int* L = Lall + (size_t)blockIdx.z * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.y < nrow_blocks) {
        // row pairs (y-1, y) with y a multiple of TILE_R; one wave per 64-pixel segment
        const int y = ((int)blockIdx.y * 4 + wave + 1) * TILE_R;
        if (y >= H) return;
        const int x = blockIdx.x * 64 + lane;
        const int xc = x < W ? x : W - 1;
        typedef typename ccl_wide<T>::type V;
        constexpr V NOVAL = ccl_wide<T>::NOVAL;
        const V v = x < W ? (V)img[(size_t)y * W + xc] : NOVAL;
        const V up = x < W ? (V)img[(size_t)(y - 1) * W + xc] : NOVAL;
        ccl_stitch_rows<CONN8, false>(L, y * W + xc, W, lane, v, up, NOVAL);
    } else {
        // column pairs (x-1, x) with x a multiple of 64; one thread per (row, boundary)
        if (cols_only_if && !*cols_only_if) return;  // ccl_border_cols_bits_kernel does them from the tiles' column bits
        const int nbound = (W - 1) / 64;  // boundaries at x = 64, 128, ...
        const int t = (((int)blockIdx.y - nrow_blocks) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (nbound == 0 || t >= H * nbound) return;
        const int y = t / nbound, x = (t % nbound + 1) * 64;
        const int pb = y * W + x, pa = pb - 1;
        const T vb = img[pb], va = img[pa];
        if (vb != 0 && va == vb) uf_union(L, pb, pa);
        if (CONN8 && y > 0) {
            if (vb != 0 && img[pa - W] == vb) uf_union(L, pb, pa - W);  // NW of b
            if (va != 0 && img[pb - W] == va) uf_union(L, pa, pb - W);  // NE of a
        }
    }
}

// The column seams of a 0 / 1 mask from the column bits ccl_tile_bits_kernel left: a wave per (tile boundary, tile row),
// a lane per row.  Stands down when the batch held other byte values (the byte version above then runs).
template <bool CONN8>
__global__ void __launch_bounds__(256) ccl_border_cols_bits_kernel(const unsigned long long* __restrict__ colbits,
                                                                   int* __restrict__ Lall, int H, int W, int segs, int trows,
                                                                   const int* __restrict__ multi) {
    if (*multi) return;
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= (segs - 1) * trows) return;  // whole wave
    const int bx = job % (segs - 1) + 1, ty = job / (segs - 1);
    const unsigned long long* cb = colbits + ((size_t)blockIdx.z * trows + ty) * segs * 2;
    const unsigned long long ra = cb[(bx - 1) * 2 + 1], lb = cb[bx * 2];  // a = last column of the left tile, b = first of the right
    int* L = Lall + (size_t)blockIdx.z * H * W;
    const int y = ty * 64 + lane, x = bx * 64;
    const int pb = y * W + x, pa = pb - 1;
    const bool va = (ra >> lane) & 1ull, vb = (lb >> lane) & 1ull;  // rows beyond H carry zero bits
    if (va && vb) uf_union(L, pb, pa);
    if (CONN8) {
        // the row above: the lane below this one, or the last row of the tile row above
        unsigned long long ra_up = ra << 1, lb_up = lb << 1;
        if (ty > 0) {
            const unsigned long long* cu = cb - (size_t)segs * 2;
            ra_up |= cu[(bx - 1) * 2 + 1] >> 63;
            lb_up |= cu[bx * 2] >> 63;
        }
        if (vb && ((ra_up >> lane) & 1ull)) uf_union(L, pb, pa - W);  // NW of b
        if (va && ((lb_up >> lane) & 1ull)) uf_union(L, pa, pb - W);  // NE of a
    }
}

// ---- raster renumbering: exclusive prefix sum over root flags ----------------------------------
constexpr int RN_CHUNK = 2048;  // pixels per block in the compress / rank passes (256 threads x 8)

// Path compression (every pixel points at its root afterwards) fused with the per-block root count that
// the raster renumbering needs: a pixel is a root iff L[p] == p, which compression never changes.
// The first two hops of all eight pixels of a thread are issued as independent loads; only deeper chains loop.
__global__ void __launch_bounds__(256) ccl_compress_count_kernel(int* __restrict__ L, int* __restrict__ blockcnt,
                                                                 size_t n, int nblk) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    int l[8], r[8], r2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        l[k] = i < n ? L[base + i] : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = l[k] >= 0 ? L[base + l[k]] : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) r2[k] = (r[k] >= 0 && r[k] != l[k]) ? L[base + r[k]] : r[k];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        bool is_root = false;
        if (l[k] >= 0) {
            int root = r[k], p = r2[k];
            while (p != root) {
                root = p;
                p = L[base + root];
            }
            if (root != l[k]) L[base + i] = root;
            is_root = root == (int)i;
            c += is_root ? 1 : 0;
        }
    }
    if (!blockcnt) return;
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[(size_t)blockIdx.y * nblk + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// rank of every root (1-based) written at the root's own position of T.  A block owns RN_CHUNK pixels in the
// order (k, thread): all eight loads are issued first, each wave ballots its root flags per k, and ONE barrier
// later every root knows how many roots precede it in the block.
__global__ void __launch_bounds__(256) root_rank_kernel(const int* __restrict__ L, const int* __restrict__ blockoff,
                                                        int* __restrict__ T, size_t n, int nblk) {
    const size_t base = (size_t)blockIdx.y * n;
    const size_t start = (size_t)blockIdx.x * RN_CHUNK;
    __shared__ int wtot[8][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        l[k] = i < n ? L[base + i] : -1;
    }
    unsigned long long m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        m[k] = __ballot(l[k] == (int)i && i < n);
        if (lane == 0) wtot[k][wave] = __popcll(m[k]);
    }
    __syncthreads();
    int run = blockoff[(size_t)blockIdx.y * nblk + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = start + (size_t)k * 256 + threadIdx.x;
        int before = run;
        for (int w = 0; w < wave; ++w) before += wtot[k][w];
        if ((m[k] >> lane) & 1ull) T[base + i] = before + __popcll(m[k] & ((1ull << lane) - 1ull)) + 1;
        run += wtot[k][0] + wtot[k][1] + wtot[k][2] + wtot[k][3];
    }
}

template <typename T>
static int ccl_roots(amt_ctx* ctx, const T* in, int* L, int* blk, int nplanes, int H, int W, int conn8) {
    const size_t n = (size_t)H * W;
    const int nblk = amt_i_rank_blocks(n);
    const int segs = (W + 63) / 64;
    dim3 gs(segs, (H + TILE_R - 1) / TILE_R, nplanes);
    const int nrow_jobs = (H - 1) / TILE_R;               // tile boundaries
    const int nrow_blocks = (nrow_jobs + 3) / 4;
    const int ncol_jobs = H * ((W - 1) / 64);             // (row, segment boundary) pairs
    const int ncol_blocks = (ncol_jobs + 256 * segs - 1) / (256 * segs);
    dim3 gb(segs, nrow_blocks + ncol_blocks, nplanes);
    if (conn8) {
        hipLaunchKernelGGL((ccl_tile_kernel<T, true>), gs, dim3(256), 0, ctx->stream, in, L, H, W, (int*)nullptr,
                           (int*)nullptr, (size_t)0);
        AMT_LAUNCH_CHECK();
        if (gb.y > 0) hipLaunchKernelGGL((ccl_border_kernel<T, true>), gb, dim3(256), 0, ctx->stream, in, L, H, W, nrow_blocks);
    } else {
        hipLaunchKernelGGL((ccl_tile_kernel<T, false>), gs, dim3(256), 0, ctx->stream, in, L, H, W, (int*)nullptr,
                           (int*)nullptr, (size_t)0);
        AMT_LAUNCH_CHECK();
        if (gb.y > 0) hipLaunchKernelGGL((ccl_border_kernel<T, false>), gb, dim3(256), 0, ctx->stream, in, L, H, W, nrow_blocks);
    }
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ccl_compress_count_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, L, blk, n, nblk);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_i_ccl_roots(amt_ctx* ctx, const void* in, int in_dtype, int* L, int* blk, int nplanes, int H, int W,
                    int conn8) {
    if (in_dtype == AMT_U8)
        return ccl_roots<uint8_t>(ctx, (const uint8_t*)in, L, blk, nplanes, H, W, conn8);
    return ccl_roots<int32_t>(ctx, (const int32_t*)in, L, blk, nplanes, H, W, conn8);
}

// 4-connected components of a uint8 mask WITHOUT the per-pixel compression pass: afterwards every foreground pixel
// points at its tile-local root and every tile-local root is listed in rootlist (nroots[plane] entries, zero on
// entry); the caller compresses the listed roots (find + path compression) and resolves pixels with two hops,
// L[L[p]].
int amt_i_tile_rows(int H) { return (H + TILE_R - 1) / TILE_R; }
size_t amt_i_rootlist_cap(int W) { return (size_t)TILE_R * W; }

// AMT_CCL_RUNS=0: amt_label keeps the parent plane for masks too (A/B switch; identical results)
static bool ccl_runs_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_CCL_RUNS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// AMT_CCL_BITS=0: the pixel-per-lane tile kernel for masks too (A/B switch; identical results)
static bool ccl_bits_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_CCL_BITS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// ints of scratch behind `multi`: the flag (16 ints) + two 64-bit column words per tile
size_t amt_i_ccl_scratch_ints(int nplanes, int H, int W) {
    return 16 + (size_t)nplanes * ((W + 63) / 64) * ((H + TILE_R - 1) / TILE_R) * 4;
}

// multi (nullable): amt_i_ccl_scratch_ints of scratch; with it, uint8 inputs take the bit-parallel tile kernel first
template <typename T, bool CONN8>
static int ccl_tileroots(amt_ctx* ctx, const T* in, int* L, int* rootlist, int* nroots, int nplanes, int H, int W,
                         int* multi = nullptr) {
    const size_t cap = amt_i_rootlist_cap(W);
    const int segs = (W + 63) / 64;
    dim3 gs(segs, (H + TILE_R - 1) / TILE_R, nplanes);
    const int nrow_jobs = (H - 1) / TILE_R;
    const int nrow_blocks = (nrow_jobs + 3) / 4;
    const int ncol_jobs = H * ((W - 1) / 64);
    const int ncol_blocks = (ncol_jobs + 256 * segs - 1) / (256 * segs);
    dim3 gb(segs, nrow_blocks + ncol_blocks, nplanes);
    bool done = false;
    if (std::is_same<T, uint8_t>::value && multi && ccl_bits_enabled() && W % 16 == 0 &&
        (reinterpret_cast<uintptr_t>(in) & 15) == 0 && ((size_t)H * W) % 16 == 0) {
        // masks: the bit-parallel tile kernel; a plane batch that turns out to hold other byte values is redone below
        AMT_HIP_CHECK(hipMemsetAsync(multi, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL((ccl_tile_bits_kernel<CONN8>), gs, dim3(64), 0, ctx->stream, (const uint8_t*)in, L, H, W, rootlist,
                           nroots, cap, multi, reinterpret_cast<unsigned long long*>(multi + 16));
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ccl_reset_lists_kernel, dim3(8), dim3(256), 0, ctx->stream, nroots,
                           (size_t)nplanes * gs.y, (const int*)multi);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL((ccl_tile_fallback_kernel<T, CONN8>), dim3(512), dim3(256), 0, ctx->stream, in, L, H, W, rootlist,
                           nroots, cap, (const int*)multi, (int)gs.x, (int)gs.y, nplanes);
        AMT_LAUNCH_CHECK();
        done = true;
    }
    if (!done) {
        hipLaunchKernelGGL((ccl_tile_kernel<T, CONN8>), gs, dim3(256), 0, ctx->stream, in, L, H, W, rootlist, nroots, cap);
        AMT_LAUNCH_CHECK();
    }
    if (gb.y > 0) {
        // with the bit kernel's column words the byte version only stitches the rows (its column jobs leave at once
        // unless the batch turned out to hold other byte values)
        hipLaunchKernelGGL((ccl_border_kernel<T, CONN8>), gb, dim3(256), 0, ctx->stream, in, L, H, W, nrow_blocks,
                           done ? (const int*)multi : (const int*)nullptr);
        AMT_LAUNCH_CHECK();
        if (done && segs > 1) {
            const int jobs = (segs - 1) * (int)gs.y;
            hipLaunchKernelGGL((ccl_border_cols_bits_kernel<CONN8>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const unsigned long long*>(multi + 16), L, H, W, segs, (int)gs.y,
                               (const int*)multi);
            AMT_LAUNCH_CHECK();
        }
    }
    return AMT_OK;
}

// the watershed's labelling of its mask from run tables (amt_internal.h)
bool amt_i_ccl_runs_ok(const void* in, int H, int W, int nplanes) {
    return ccl_runs_enabled() && W % 16 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && ((size_t)H * W) % 16 == 0 &&
           (size_t)nplanes * amt_i_tile_rows(H) * ((W + 63) / 64) * RT_CAP < 0x7fffffffull;  // run indices are ints
}

int amt_i_ccl_tileroots_runs_u8(amt_ctx* ctx, const uint8_t* in, int* L, int* rootlist, int* nroots, int nplanes, int H,
                                int W, unsigned long long* tbits, unsigned short* rtab, int* nruns, unsigned short* roff) {
    const int segs = (W + 63) / 64, trows = amt_i_tile_rows(H);
    hipLaunchKernelGGL((ccl_tile_runs_kernel<false, true>), dim3(segs, trows, nplanes), dim3(64), 0, ctx->stream, in, L, H, W,
                       rootlist, nroots, amt_i_rootlist_cap(W), (int*)nullptr, tbits, rtab, nruns, roff);
    AMT_LAUNCH_CHECK();
    const int jobs = segs * (trows - 1) + (segs - 1) * trows;
    if (jobs > 0) {
        hipLaunchKernelGGL((ccl_seams_runs_kernel<false>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream, tbits,
                           rtab, nruns, L, H, W, segs, trows, (const int*)nullptr);
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

int amt_i_ccl_tileroots_u8(amt_ctx* ctx, const uint8_t* in, int* L, int* rootlist, int* nroots, int nplanes, int H,
                           int W, int* multi) {
    return ccl_tileroots<uint8_t, false>(ctx, in, L, rootlist, nroots, nplanes, H, W, multi);
}

// A[t] = A[component root of t] for every listed tile root t (the lists must be compressed already): afterwards a
// pixel reaches its component's entry of A with ONE hop through its tile root
__global__ void __launch_bounds__(256) roots_propagate_kernel(int* __restrict__ Aall, const int* __restrict__ Lall,
                                                              const int* __restrict__ rootlist,
                                                              const int* __restrict__ nroots, size_t cap, size_t n) {
    const int plane = blockIdx.z, shard = plane * gridDim.y + blockIdx.y;
    int* A = Aall + (size_t)plane * n;
    const int* L = Lall + (size_t)plane * n;
    const int cnt = nroots[shard] < (int)cap ? nroots[shard] : (int)cap;
    const int* lst = rootlist + (size_t)shard * cap;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int t = lst[k], r = L[t];
        if (r != t) A[t] = A[r];
    }
}

int amt_i_propagate_roots(amt_ctx* ctx, int* A, const int* L, const int* rootlist, const int* nroots, int nplanes, int H,
                          int W) {
    hipLaunchKernelGGL(roots_propagate_kernel, dim3(4, amt_i_tile_rows(H), nplanes), dim3(256), 0, ctx->stream, A, L,
                       rootlist, nroots, amt_i_rootlist_cap(W), (size_t)H * W);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// compress the listed tile roots, count the component roots of every RN_CHUNK-pixel chunk (what the raster
// renumbering scans) and set the root's bit in the plane's root bitmap; blockcnt and bitmap must be zero on entry
__global__ void __launch_bounds__(256) roots_compress_count_kernel(int* __restrict__ Lall, const int* __restrict__ rootlist,
                                                                   const int* __restrict__ nroots,
                                                                   int* __restrict__ blockcnt,
                                                                   unsigned long long* __restrict__ bitmap, size_t cap,
                                                                   size_t n, int nblk) {
    const int plane = blockIdx.z, shard = plane * gridDim.y + blockIdx.y;
    int* L = Lall + (size_t)plane * n;
    const int cnt = nroots[shard] < (int)cap ? nroots[shard] : (int)cap;
    const int* lst = rootlist + (size_t)shard * cap;
    unsigned long long* bm = bitmap + (size_t)plane * ((n + 63) / 64);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int t = lst[k];
        int r = L[t];
        int p = L[r];
        while (p != r) {
            r = p;
            p = L[r];
        }
        if (r != t) {
            L[t] = r;
        } else {
            atomicAdd(&blockcnt[(size_t)plane * nblk + t / RN_CHUNK], 1);
            atomicOr(&bm[t >> 6], 1ull << (t & 63));
        }
    }
}

// T[t] = raster rank (1-based) of the component of every listed tile root t: roots before the root's chunk
// (scanned counts) + set bits of the root bitmap before the root inside its chunk.  Only the lists are read, not
// the label plane.
__global__ void __launch_bounds__(256) roots_rank_kernel(int* __restrict__ Tall, const int* __restrict__ Lall,
                                                         const int* __restrict__ rootlist, const int* __restrict__ nroots,
                                                         const int* __restrict__ blockoff,
                                                         const unsigned long long* __restrict__ bitmap, size_t cap,
                                                         size_t n, int nblk) {
    const int plane = blockIdx.z, shard = plane * gridDim.y + blockIdx.y;
    int* T = Tall + (size_t)plane * n;
    const int* L = Lall + (size_t)plane * n;
    const int cnt = nroots[shard] < (int)cap ? nroots[shard] : (int)cap;
    const int* lst = rootlist + (size_t)shard * cap;
    const unsigned long long* bm = bitmap + (size_t)plane * ((n + 63) / 64);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int t = lst[k], r = L[t];
        const int chunk = r / RN_CHUNK, wr = r >> 6;
        int rank = blockoff[(size_t)plane * nblk + chunk] + 1;
        for (int w = chunk * (RN_CHUNK / 64); w < wr; ++w) rank += __popcll(bm[w]);
        rank += __popcll(bm[wr] & ((1ull << (r & 63)) - 1ull));
        T[t] = rank;
    }
}

// out = rank of the component of every pixel: pixel -> tile root, whose T entry was copied from the component root
__global__ void __launch_bounds__(256) apply_rank_kernel(const int* __restrict__ L, const int* __restrict__ T,
                                                         int* __restrict__ out, size_t n,
                                                         const int* __restrict__ only_if = nullptr) {
    if (only_if && !*only_if) return;
    const size_t base = (size_t)blockIdx.y * n;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        if (i0 + 3 < n && ((base + i0) & 3) == 0) {
            const int4 l = *reinterpret_cast<const int4*>(L + base + i0);
            int4 o;
            o.x = l.x >= 0 ? T[base + l.x] : 0;
            o.y = l.y >= 0 ? T[base + l.y] : 0;
            o.z = l.z >= 0 ? T[base + l.z] : 0;
            o.w = l.w >= 0 ? T[base + l.w] : 0;
            *reinterpret_cast<int4*>(out + base + i0) = o;
        } else {
            for (size_t i = i0; i < n && i < i0 + 4; ++i) {
                const int l = L[base + i];
                out[base + i] = l >= 0 ? T[base + l] : 0;
            }
        }
    }
}

int amt_i_rank_blocks(size_t n) { return (int)((n + RN_CHUNK - 1) / RN_CHUNK); }

int amt_i_rank_roots(amt_ctx* ctx, const int* L, int* T, int* blk, int* count_dev, int nplanes, size_t n) {
    const int nblk = amt_i_rank_blocks(n);  // blk holds the per-block root counts written by amt_i_ccl_roots
    AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, count_dev, nplanes));
    hipLaunchKernelGGL(root_rank_kernel, dim3(nblk, nplanes), dim3(256), 0, ctx->stream, L, blk, T, n, nblk);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

static int label_impl(amt_ctx* ctx, const void* in, int in_dtype, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                      int connectivity, bool truth);

extern "C" int amt_label(amt_ctx* ctx, const void* in, int in_dtype, int32_t* out, int32_t* count_dev, int nplanes,
                         int H, int W, int connectivity) {
    return label_impl(ctx, in, in_dtype, out, count_dev, nplanes, H, W, connectivity, false);
}

extern "C" int amt_label_mask(amt_ctx* ctx, const uint8_t* mask, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                              int connectivity) {
    return label_impl(ctx, mask, AMT_U8, out, count_dev, nplanes, H, W, connectivity, true);
}

// truth: the uint8 input is a truth value (foreground = byte != 0, as for a bool array): the run-table path then needs no
// "other byte values" flag and none of the byte kernels that stand by for it
static int label_impl(amt_ctx* ctx, const void* in, int in_dtype, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                      int connectivity, bool truth) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "label: bad arguments");
    AMT_REQUIRE(in_dtype == AMT_U8 || in_dtype == AMT_I32, "label: in_dtype must be AMT_U8 or AMT_I32");
    AMT_REQUIRE(connectivity == 1 || connectivity == 2, "label: connectivity must be 1 or 2");
    AMT_REQUIRE((size_t)H * W < 0x7fffffffull, "label: plane too large");
    AMT_REQUIRE((const void*)in != (const void*)out, "label: in-place operation is not supported");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int nblk = amt_i_rank_blocks(n);
    // union-find parents L, ranks T, the lists of tile-local roots (one list per tile row; components are sets of
    // EQUAL-valued pixels, so every pixel can be a root of its own), per-chunk root counts
    const int trows = amt_i_tile_rows(H);
    const size_t cap = amt_i_rootlist_cap(W);
    const size_t nlist = (size_t)nplanes * trows;
    const size_t nwords = (n + 63) / 64;
    const size_t ntiles_all = (size_t)nplanes * trows * ((W + 63) / 64);  // run tables of the 0 / 1 mask path
    AMT_TRY(amt_arena_begin(ctx, 2 * amt_align((size_t)nplanes * n * 4) + amt_align(nlist * cap * 4) +
                                     amt_align((size_t)nplanes * nblk * 4) + amt_align(nlist * 4) +
                                     amt_align((size_t)nplanes * nwords * 8) +
                                     amt_align(amt_i_ccl_scratch_ints(nplanes, H, W) * 4) +
                                     amt_align(ntiles_all * 64 * 8) + amt_align(ntiles_all * RT_CAP * 2) +
                                     amt_align(ntiles_all * 4)));
    int* L = arena_take_t<int>(ctx, (size_t)nplanes * n);
    // "a byte other than 0 / 1 was seen" + the tiles' column words (ccl_tile_bits_kernel)
    int* multi = arena_take_t<int>(ctx, amt_i_ccl_scratch_ints(nplanes, H, W));
    int* T = arena_take_t<int>(ctx, (size_t)nplanes * n);
    int* rootlist = arena_take_t<int>(ctx, nlist * cap);
    int* blk = arena_take_t<int>(ctx, (size_t)nplanes * nblk);
    int* nroots = arena_take_t<int>(ctx, nlist);
    unsigned long long* bitmap = arena_take_t<unsigned long long>(ctx, (size_t)nplanes * nwords);
    // chunk counts, list counts and the root bitmap were taken from the arena one after the other: ONE fill clears them
    AMT_HIP_CHECK(hipMemsetAsync(blk, 0, (size_t)((char*)(bitmap + (size_t)nplanes * nwords) - (char*)blk), ctx->stream));
    // tile-local union-find + seams; only the listed tile roots are compressed, pixels resolve in two hops
    const int segs = (W + 63) / 64;
    const int ntiles = nplanes * trows * segs;
    const bool runs = in_dtype == AMT_U8 && ccl_runs_enabled() && W % 16 == 0 &&
                      (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                      n % 16 == 0 && (size_t)ntiles * 64 < 0x7fffffffull;
    if (runs) {
        // 0 / 1 masks: run tables instead of a parent plane.  A batch that turns out to hold other byte values raises
        // *multi; every run-table kernel then stands down and the byte kernels -- which otherwise leave at once -- redo it
        unsigned long long* tbits = arena_take_t<unsigned long long>(ctx, (size_t)ntiles * 64);
        unsigned short* rtab = arena_take_t<unsigned short>(ctx, (size_t)ntiles * RT_CAP);
        int* nruns = arena_take_t<int>(ctx, (size_t)ntiles);
        const uint8_t* in8 = (const uint8_t*)in;
        const bool c8 = connectivity == 2;
        dim3 gs(segs, trows, nplanes);
        if (truth) {
            // a truth-value mask: tile pass, seams, the numbering of the listed tile roots, expansion -- nothing stands by
            if (c8)
                hipLaunchKernelGGL((ccl_tile_runs_kernel<true, true>), gs, dim3(64), 0, ctx->stream, in8, L, H, W, rootlist, nroots,
                                   cap, (int*)nullptr, tbits, rtab, nruns, (unsigned short*)nullptr);
            else
                hipLaunchKernelGGL((ccl_tile_runs_kernel<false, true>), gs, dim3(64), 0, ctx->stream, in8, L, H, W, rootlist, nroots,
                                   cap, (int*)nullptr, tbits, rtab, nruns, (unsigned short*)nullptr);
            AMT_LAUNCH_CHECK();
            const int jobs = segs * (trows - 1) + (segs - 1) * trows;
            if (jobs > 0) {
                if (c8)
                    hipLaunchKernelGGL((ccl_seams_runs_kernel<true>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream,
                                       tbits, rtab, nruns, L, H, W, segs, trows, (const int*)nullptr);
                else
                    hipLaunchKernelGGL((ccl_seams_runs_kernel<false>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream,
                                       tbits, rtab, nruns, L, H, W, segs, trows, (const int*)nullptr);
                AMT_LAUNCH_CHECK();
            }
            hipLaunchKernelGGL(roots_compress_count_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, L, rootlist,
                               nroots, blk, bitmap, cap, n, nblk);
            AMT_LAUNCH_CHECK();
            AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, count_dev, nplanes));
            hipLaunchKernelGGL(roots_rank_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, T, L, rootlist, nroots,
                               blk, bitmap, cap, n, nblk);
            AMT_LAUNCH_CHECK();
            hipLaunchKernelGGL(ccl_expand_runs_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, ctx->stream, tbits, rtab, nruns, T, out,
                               H, W, segs, trows, ntiles, (const int*)nullptr);
            AMT_LAUNCH_CHECK();
            return AMT_OK;
        }
        AMT_HIP_CHECK(hipMemsetAsync(multi, 0, sizeof(int), ctx->stream));
        if (c8)
            hipLaunchKernelGGL((ccl_tile_runs_kernel<true>), gs, dim3(64), 0, ctx->stream, in8, L, H, W, rootlist, nroots, cap,
                               multi, tbits, rtab, nruns, (unsigned short*)nullptr);
        else
            hipLaunchKernelGGL((ccl_tile_runs_kernel<false>), gs, dim3(64), 0, ctx->stream, in8, L, H, W, rootlist, nroots, cap,
                               multi, tbits, rtab, nruns, (unsigned short*)nullptr);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ccl_reset_lists_kernel, dim3(8), dim3(256), 0, ctx->stream, nroots, nlist, (const int*)multi);
        AMT_LAUNCH_CHECK();
        const int nrow_jobs = (H - 1) / TILE_R;
        const int nrow_blocks = (nrow_jobs + 3) / 4;
        const int ncol_jobs = H * ((W - 1) / 64);
        const int ncol_blocks = (ncol_jobs + 256 * segs - 1) / (256 * segs);
        dim3 gb(segs, nrow_blocks + ncol_blocks, nplanes);
        if (c8) {
            hipLaunchKernelGGL((ccl_tile_fallback_kernel<uint8_t, true>), dim3(512), dim3(256), 0, ctx->stream, in8, L, H, W,
                               rootlist, nroots, cap, (const int*)multi, segs, trows, nplanes);
            if (gb.y > 0)
                hipLaunchKernelGGL((ccl_border_kernel<uint8_t, true>), gb, dim3(256), 0, ctx->stream, in8, L, H, W, nrow_blocks,
                                   (const int*)nullptr, (const int*)multi);
        } else {
            hipLaunchKernelGGL((ccl_tile_fallback_kernel<uint8_t, false>), dim3(512), dim3(256), 0, ctx->stream, in8, L, H, W,
                               rootlist, nroots, cap, (const int*)multi, segs, trows, nplanes);
            if (gb.y > 0)
                hipLaunchKernelGGL((ccl_border_kernel<uint8_t, false>), gb, dim3(256), 0, ctx->stream, in8, L, H, W, nrow_blocks,
                                   (const int*)nullptr, (const int*)multi);
        }
        AMT_LAUNCH_CHECK();
        const int jobs = segs * (trows - 1) + (segs - 1) * trows;
        if (jobs > 0) {
            if (c8)
                hipLaunchKernelGGL((ccl_seams_runs_kernel<true>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream,
                                   tbits, rtab, nruns, L, H, W, segs, trows, (const int*)multi);
            else
                hipLaunchKernelGGL((ccl_seams_runs_kernel<false>), dim3((jobs + 3) / 4, 1, nplanes), dim3(256), 0, ctx->stream,
                                   tbits, rtab, nruns, L, H, W, segs, trows, (const int*)multi);
            AMT_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(roots_compress_count_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, L, rootlist,
                           nroots, blk, bitmap, cap, n, nblk);
        AMT_LAUNCH_CHECK();
        AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, count_dev, nplanes));
        hipLaunchKernelGGL(roots_rank_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, T, L, rootlist, nroots,
                           blk, bitmap, cap, n, nblk);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(ccl_expand_runs_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, ctx->stream, tbits, rtab, nruns, T, out,
                           H, W, segs, trows, ntiles, (const int*)multi);
        AMT_LAUNCH_CHECK();
        hipLaunchKernelGGL(apply_rank_kernel, dim3(amt_grid_for(n, 1024, 256), nplanes), dim3(256), 0, ctx->stream, L, T, out, n,
                           (const int*)multi);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    if (in_dtype == AMT_U8) {
        if (connectivity == 2)
            AMT_TRY((ccl_tileroots<uint8_t, true>(ctx, (const uint8_t*)in, L, rootlist, nroots, nplanes, H, W, multi)));
        else
            AMT_TRY((ccl_tileroots<uint8_t, false>(ctx, (const uint8_t*)in, L, rootlist, nroots, nplanes, H, W, multi)));
    } else {
        if (connectivity == 2)
            AMT_TRY((ccl_tileroots<int32_t, true>(ctx, (const int32_t*)in, L, rootlist, nroots, nplanes, H, W)));
        else
            AMT_TRY((ccl_tileroots<int32_t, false>(ctx, (const int32_t*)in, L, rootlist, nroots, nplanes, H, W)));
    }
    hipLaunchKernelGGL(roots_compress_count_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, L, rootlist,
                       nroots, blk, bitmap, cap, n, nblk);
    AMT_LAUNCH_CHECK();
    // raster numbering from the lists alone: scan the per-chunk root counts, then rank every listed tile root
    AMT_TRY(amt_scan_excl(ctx, blk, nblk, (size_t)nblk, count_dev, nplanes));
    hipLaunchKernelGGL(roots_rank_kernel, dim3(4, trows, nplanes), dim3(256), 0, ctx->stream, T, L, rootlist, nroots,
                       blk, bitmap, cap, n, nblk);
    AMT_LAUNCH_CHECK();
    dim3 g1(amt_grid_for(n, 1024, 4096), nplanes);
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
    const unsigned ml = (unsigned)max_label;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (size_t)gridDim.x * 1024) {
        if (i0 + 3 < n && ((base + i0) & 3) == 0) {
            const int4 v = *reinterpret_cast<const int4*>(in + base + i0);
            int4 o;  // v - 1 < max_label as unsigned  <=>  1 <= v <= max_label
            o.x = (unsigned)(v.x - 1) < ml ? M[v.x] : 0;
            o.y = (unsigned)(v.y - 1) < ml ? M[v.y] : 0;
            o.z = (unsigned)(v.z - 1) < ml ? M[v.z] : 0;
            o.w = (unsigned)(v.w - 1) < ml ? M[v.w] : 0;
            *reinterpret_cast<int4*>(out + base + i0) = o;
        } else {
            for (size_t i = i0; i < n && i < i0 + 4; ++i) {
                const int v = in[base + i];
                out[base + i] = (unsigned)(v - 1) < ml ? M[v] : 0;
            }
        }
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

// present[l] = 1 for l = 1 .. nlabels[plane] (the caller vouches that exactly these labels occur)
__global__ void __launch_bounds__(256) presence_fill_kernel(int* __restrict__ present, const int* __restrict__ nlabels,
                                                            int max_label) {
    int* P = present + (size_t)blockIdx.y * (max_label + 1);
    const int k = nlabels[blockIdx.y] < max_label ? nlabels[blockIdx.y] : max_label;
    for (int l = blockIdx.x * 256 + threadIdx.x; l <= max_label; l += gridDim.x * 256) P[l] = (l >= 1 && l <= k) ? 1 : 0;
}

// the two halves of the label map for callers that find the frame-touching labels themselves (amt_watershed.hip's fused
// watershed + clear_border + relabel): P = nplanes x (max_label + 1) ints; fill -> caller sets P[l] = 2 for every label
// to drop -> drop_and_scan leaves P[l] = new label (0 = dropped) and the number of survivors in count_dev
int amt_i_presence_fill(amt_ctx* ctx, int* P, const int* nlabels_dev, int max_label, int nplanes) {
    hipLaunchKernelGGL(presence_fill_kernel, dim3(amt_grid_for((size_t)max_label + 1, 256, 64), nplanes), dim3(256), 0,
                       ctx->stream, P, nlabels_dev, max_label);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
int amt_i_drop_and_scan(amt_ctx* ctx, int* P, int max_label, int* count_dev, int nplanes) {
    hipLaunchKernelGGL(drop_flagged_kernel, dim3(amt_grid_for((size_t)max_label + 1, 256, 64), nplanes), dim3(256), 0,
                       ctx->stream, P, max_label);
    AMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(presence_scan_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, P, max_label, count_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_clear_border_relabel(amt_ctx* ctx, const int32_t* in, int32_t* out, int32_t* count_dev, int nplanes,
                                        int H, int W, int max_label, const int32_t* nlabels_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0 && max_label >= 0, "clear_border_relabel: bad arguments");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    size_t msz = (size_t)nplanes * ((size_t)max_label + 1);
    AMT_TRY(amt_arena_begin(ctx, amt_align(msz * 4)));
    int* P = arena_take_t<int>(ctx, msz);
    dim3 g1(amt_grid_for(n, 256, 4096), nplanes);
    if (nlabels_dev) {  // labels 1 .. nlabels[plane] are known to be present: no pass over the image
        hipLaunchKernelGGL(presence_fill_kernel, dim3(amt_grid_for((size_t)max_label + 1, 256, 64), nplanes), dim3(256),
                           0, ctx->stream, P, nlabels_dev, max_label);
    } else {
        AMT_HIP_CHECK(hipMemsetAsync(P, 0, msz * 4, ctx->stream));
        hipLaunchKernelGGL(presence_kernel, g1, dim3(256), 0, ctx->stream, in, P, n, max_label);
    }
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
constexpr int SP_CHUNK = 4096;  // pixels per block: 256 threads x 16 bytes

// bit j of the result = byte j of the thread's 16 pixels is non-zero (one 128-bit load when aligned)
__device__ __forceinline__ unsigned sp_load16(const uint8_t* __restrict__ p, size_t i, size_t n) {
    unsigned bits = 0;
    if (i + 15 < n && ((reinterpret_cast<uintptr_t>(p + i) & 15) == 0)) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // high bit of every non-zero byte, gathered into 4 consecutive bits
            unsigned t = (((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k]) & 0x80808080u;
            t = (t >> 7) | (t >> 14) | (t >> 21) | (t >> 28);
            bits |= (t & 0xFu) << (4 * k);
        }
    } else {
        for (int k = 0; k < 16; ++k)
            if (i + k < n && p[i + k] != 0) bits |= 1u << k;
    }
    return bits;
}

__global__ void __launch_bounds__(256) sp_count_kernel(const uint8_t* __restrict__ in, int* __restrict__ blockcnt,
                                                       size_t n, int nblk) {
    const uint8_t* src = in + (size_t)blockIdx.y * n;
    const size_t i = (size_t)blockIdx.x * SP_CHUNK + (size_t)threadIdx.x * 16;
    int c = i < n ? __popc(sp_load16(src, i, n)) : 0;
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
    const uint8_t* src = in + base;
    int* lst = list + (size_t)blockIdx.y * cap;
    int* par = parent + (size_t)blockIdx.y * cap;
    const size_t i = (size_t)blockIdx.x * SP_CHUNK + (size_t)threadIdx.x * 16;
    unsigned bits = i < n ? sp_load16(src, i, n) : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exclusive prefix sum of the per-thread counts across the block (raster order = thread order)
    const int cnt = __popc(bits);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __shared__ int wave_tot[4];
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int idx = blockoff[(size_t)blockIdx.y * nblk + blockIdx.x] + incl - cnt;
    for (int w = 0; w < wave; ++w) idx += wave_tot[w];
    while (bits) {
        const int k = __ffs((int)bits) - 1;
        bits &= bits - 1;
        if (idx < cap) {
            lst[idx] = (int)(i + k);
            par[idx] = idx;
            out[base + i + k] = idx + 1;
        }
        ++idx;
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

// out[keep_list[k]] = 0 for the pixels the previous call wrote (sparse clear instead of a full-plane memset)
__global__ void __launch_bounds__(256) sp_unwrite_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                                         int* __restrict__ out, size_t n, int cap) {
    const int K = count[blockIdx.y] < cap ? count[blockIdx.y] : cap;
    const int* lst = list + (size_t)blockIdx.y * cap;
    int* o = out + (size_t)blockIdx.y * n;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) o[lst[k]] = 0;
}

static int label_sparse_impl(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                             int connectivity, int capacity, int32_t* keep_list, int32_t* keep_count);

extern "C" int amt_label_sparse(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H,
                                int W, int connectivity, int capacity) {
    return label_sparse_impl(ctx, in, out, count_dev, nplanes, H, W, connectivity, capacity, nullptr, nullptr);
}

extern "C" int amt_label_sparse_reuse(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes,
                                      int H, int W, int connectivity, int capacity, int32_t* keep_list,
                                      int32_t* keep_count) {
    AMT_REQUIRE(keep_list && keep_count, "label_sparse_reuse: keep_list / keep_count are required");
    return label_sparse_impl(ctx, in, out, count_dev, nplanes, H, W, connectivity, capacity, keep_list, keep_count);
}

static int label_sparse_impl(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                             int connectivity, int capacity, int32_t* keep_list, int32_t* keep_count) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(in && out && nplanes >= 0 && H > 0 && W > 0, "label_sparse: bad arguments");
    AMT_REQUIRE(connectivity == 1 || connectivity == 2, "label_sparse: connectivity must be 1 or 2");
    AMT_REQUIRE(capacity >= 1, "label_sparse: capacity must be positive");
    AMT_REQUIRE((size_t)H * W < 0x7fffffffull, "label_sparse: plane too large");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const int nblk = (int)((n + SP_CHUNK - 1) / SP_CHUNK);
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
    const unsigned gk0 = amt_grid_for((size_t)capacity, 256, 64);
    if (keep_list) {
        // `out` is zero except where the previous call on these buffers wrote: undo exactly those writes (the list of
        // this call replaces the kept one below) -- 0.5 GB of memset per 32 planes of 2048^2 becomes a few thousand stores
        hipLaunchKernelGGL(sp_unwrite_kernel, dim3(gk0, nplanes), dim3(256), 0, ctx->stream, keep_list, keep_count, out, n,
                           capacity);
        AMT_LAUNCH_CHECK();
        list = keep_list;
        total_c = keep_count;
    } else {
        AMT_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)nplanes * n * sizeof(int32_t), ctx->stream));
    }
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
