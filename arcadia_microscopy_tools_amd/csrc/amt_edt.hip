// Exact Euclidean distance transform and the peak mask of the config-3 marker recipe.
//
// scipy.ndimage.distance_transform_edt(mask) == sqrt(float64(d2)) bitwise, with d2 the exact integer
// squared distance to the nearest zero pixel.  d2 is computed separably:
//   pass 1 (rows)   : g(y,x) = distance to the nearest zero pixel in row y (G_INF if none).  The row is
//                     turned into 64-pixel bit words of "zero" flags (one ballot each); the nearest zero on
//                     either side of a pixel is a clz / ffs on its own word, walking to further words only
//                     across solid 64-pixel stretches.  g is stored as uint16 (sides are <= 32768).
//   pass 2 (columns): d2(y,x) = min_k (k^2 + g(y+-k,x)^2), scanning k outward while k^2 < best.  A block
//                     stages 64 columns x (64 + 2*16) or (32 + 2*24) rows of g in LDS; only searches deeper than the halo
//                     continue in HBM (coalesced: a wave reads 64 consecutive x of row y+-k).
// Both searches are exact and cost O(distance) per pixel, which is what nuclei-sized objects need; they
// degrade (never fail) on very large solid regions.
#include "amt_internal.h"

constexpr unsigned G_INF = 0xFFFFu;  // no zero pixel in this row
// column tiles: 64 rows + 2 x 16 halo rows for batches (least staging per output row), 32 + 2 x 24 when a call
// has too few tiles to fill the chip (single planes)

__global__ void __launch_bounds__(256) edt_rows_kernel(const uint8_t* __restrict__ mask, unsigned short* __restrict__ g,
                                                       int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* zw = reinterpret_cast<unsigned long long*>(smem_raw);  // zero flags, 64 pixels per word
    const size_t base = ((size_t)blockIdx.y * H + blockIdx.x) * W;
    const int nw = (W + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (int x0 = (threadIdx.x >> 6) * 64; x0 < W; x0 += 256) {
        const int x = x0 + lane;
        const unsigned long long z = __ballot(x < W && mask[base + x] == 0);  // beyond W: "not zero"
        if (lane == 0) zw[x0 >> 6] = z;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const int wi = x >> 6, bit = x & 63;
        const unsigned long long own = zw[wi];
        unsigned d = 0;
        if (!((own >> bit) & 1ull)) {
            // nearest zero to the left
            unsigned dl = G_INF, dr = G_INF;
            unsigned long long m = own & ((1ull << bit) - 1ull);
            int w = wi;
            while (m == 0 && w > 0) m = zw[--w];
            if (m) dl = (unsigned)(x - (w * 64 + 63 - __clzll((long long)m)));
            // nearest zero to the right
            m = bit == 63 ? 0ull : (own >> (bit + 1)) << (bit + 1);
            w = wi;
            while (m == 0 && w + 1 < nw) m = zw[++w];
            if (m) dr = (unsigned)(w * 64 + __ffsll((long long)m) - 1 - x);
            d = dl < dr ? dl : dr;
        }
        g[base + x] = (unsigned short)d;
    }
}

// Rows whose width is a multiple of 8 (and 16-byte aligned planes): a thread owns 8 consecutive pixels -- one
// 8-byte load, the "is zero" flags of its bytes by carry arithmetic, one byte of the row's flag words in LDS; the
// nearest zero left of the group (clz) and right of it (ffs) seed a forward and a backward sweep over the 8
// pixels, and the 8 distances leave in one 16-byte store.
__global__ void __launch_bounds__(256) edt_rows8_kernel(const uint8_t* __restrict__ mask, unsigned short* __restrict__ g,
                                                        int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* zw = reinterpret_cast<unsigned long long*>(smem_raw);  // zero flags, 64 pixels per word
    uint8_t* zb = reinterpret_cast<uint8_t*>(smem_raw);
    const size_t base = ((size_t)blockIdx.y * H + blockIdx.x) * W;
    const int nw = (W + 63) / 64, nb = W >> 3;
    for (int b = nb + threadIdx.x; b < nw * 8; b += 256) zb[b] = 0;  // beyond W: "not zero"
    for (int b = threadIdx.x; b < nb; b += 256) {
        const unsigned long long v = *reinterpret_cast<const unsigned long long*>(mask + base + (size_t)b * 8);
        unsigned long long t = (v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
        t = ~(t | v | 0x7f7f7f7f7f7f7f7full);                          // 0x80 in every zero byte
        zb[b] = (uint8_t)(((t >> 7) * 0x0102040810204080ull) >> 56);   // bit i = byte i is zero
    }
    __syncthreads();
    constexpr unsigned BIG = 0x20000u;  // > any distance inside a row (sides are <= 32768)
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int x0 = b * 8, wi = x0 >> 6, bit0 = x0 & 63;
        const unsigned long long own = zw[wi];
        const unsigned z8 = (unsigned)(own >> bit0) & 0xffu;
        uint4 out = make_uint4(0u, 0u, 0u, 0u);
        if (z8 != 0xffu) {
            unsigned dl = BIG, dr = BIG;  // distance of pixel x0 to the nearest zero left of the group / of x0+7 right
            unsigned long long m = own & ((1ull << bit0) - 1ull);
            int w = wi;
            while (m == 0 && w > 0) m = zw[--w];
            if (m) dl = (unsigned)(x0 - (w * 64 + 63 - __clzll((long long)m)));
            const int sh = bit0 + 8;
            m = sh == 64 ? 0ull : (own >> sh) << sh;
            w = wi;
            while (m == 0 && w + 1 < nw) m = zw[++w];
            if (m) dr = (unsigned)(w * 64 + __ffsll((long long)m) - 1 - (x0 + 7));
            unsigned L[8], R[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned prev = i == 0 ? dl : L[i - 1] + 1u;
                L[i] = ((z8 >> i) & 1u) ? 0u : (prev < BIG ? prev : BIG);
            }
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                const unsigned nxt = i == 7 ? dr : R[i + 1] + 1u;
                R[i] = ((z8 >> i) & 1u) ? 0u : (nxt < BIG ? nxt : BIG);
            }
            unsigned d[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned mn = L[i] < R[i] ? L[i] : R[i];
                d[i] = mn >= G_INF ? G_INF : mn;
            }
            out.x = d[0] | (d[1] << 16);
            out.y = d[2] | (d[3] << 16);
            out.z = d[4] | (d[5] << 16);
            out.w = d[6] | (d[7] << 16);
        }
        *reinterpret_cast<uint4*>(g + base + x0) = out;
    }
}

template <int EC_ROWS, int EC_HALO>
__global__ void __launch_bounds__(256) edt_cols_kernel(const unsigned short* __restrict__ g, int* __restrict__ d2_out,
                                                       double* __restrict__ edt_out, int H, int W) {
    constexpr int EC_TROWS = EC_ROWS + 2 * EC_HALO;
    __shared__ unsigned short tile[EC_TROWS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * EC_ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const unsigned short* gp = g + plane;
    const int xc = x < W ? x : W - 1;
    // all loads of a wave are issued before the first LDS store
    unsigned short gv[EC_TROWS / 4];
#pragma unroll
    for (int j = 0; j < EC_TROWS / 4; ++j) {
        const int y = y0 - EC_HALO + wave + 4 * j;
        gv[j] = (y >= 0 && y < H && x < W) ? gp[(size_t)y * W + xc] : (unsigned short)G_INF;
    }
#pragma unroll
    for (int j = 0; j < EC_TROWS / 4; ++j) tile[wave + 4 * j][lane] = gv[j];
    __syncthreads();
    if (x >= W) return;
#pragma unroll 1
    for (int j = 0; j < EC_ROWS / 4; ++j) {
        const int ly = EC_HALO + wave * (EC_ROWS / 4) + j;
        const int y = y0 - EC_HALO + ly;
        if (y >= H) break;
        const unsigned g0 = tile[ly][lane];
        unsigned best = 0;
        if (g0 != 0) {
            best = g0 == G_INF ? 0xFFFFFFFFu : g0 * g0;
            unsigned k = 1;
            for (; k <= (unsigned)EC_HALO && k * k < best; ++k) {
                const unsigned gu = tile[ly - (int)k][lane], gd = tile[ly + (int)k][lane];
                const unsigned gm = gu < gd ? gu : gd;
                if (gm != G_INF) {
                    const unsigned c = k * k + gm * gm;
                    best = c < best ? c : best;
                }
            }
            for (; k < 65536u && (unsigned long long)k * k < best; ++k) {  // beyond the LDS halo
                const int yu = y - (int)k, yd = y + (int)k;
                if (yu < 0 && yd >= H) break;
                unsigned gm = G_INF;
                if (yu >= 0) gm = gp[(size_t)yu * W + x];
                if (yd < H) {
                    const unsigned gd = gp[(size_t)yd * W + x];
                    gm = gd < gm ? gd : gm;
                }
                if (gm != G_INF) {
                    const unsigned c = k * k + gm * gm;  // < 2^31: both terms < 2^30
                    best = c < best ? c : best;
                }
            }
            // no zero pixel in the whole plane (a column sees none only then): scipy's feature transform then
            // measures from index (-1, 0) -- distance_transform_edt(np.ones((2, 2))) == [[1, sqrt 2], [2, sqrt 5]]
            if (best > 0x7fffffffu) best = (unsigned)(y + 1) * (unsigned)(y + 1) + (unsigned)x * (unsigned)x;
        }
        const size_t i = plane + (size_t)y * W + x;
        if (d2_out) d2_out[i] = (int)best;
        if (edt_out) edt_out[i] = sqrt((double)best);
    }
}

// ---- both passes in one kernel, from a packed plane of zero flags (round 3) -------------------------------------------
// The row pass wrote g as uint16 (2 bytes per pixel) and the column pass read it back with its halo (3): more traffic
// than the mask (1) and the result (4) together.  g is a function of the row's ZERO FLAGS alone, and those are one bit per
// pixel: edt_zero_words_kernel packs them (64 pixels per word, 1/8 byte per pixel), and a column tile computes the g of
// its 64 x (ROWS + 2 HALO) window itself -- three words per row (its own segment and the ones left / right of it) sit
// in LDS, a thread owns 8 pixels of a row exactly as edt_rows8_kernel does, and only a row without a zero pixel within 64
// columns walks further words in global memory.  Searches deeper than the halo evaluate g(y +- k, x) from the words too.
__global__ void __launch_bounds__(256) edt_zero_words_kernel(const uint8_t* __restrict__ mask, unsigned long long* __restrict__ zw,
                                                             int H, int W, int WW, size_t nwords, int fast) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nwords) return;
    const int wi = (int)(i % WW);
    const size_t row = i / WW;  // plane * H + y
    const uint8_t* p = mask + row * W + (size_t)wi * 64;
    const int valid = W - wi * 64 < 64 ? W - wi * 64 : 64;
    unsigned long long z = 0;
    if (fast) {  // W % 16 == 0 and a 16-byte aligned plane: 16-byte loads
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const uint4*>(p + (16 * j < valid ? 16 * j : 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            unsigned sixteen = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned nz = ((((v[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v[k]) >> 7) & 0x01010101u;  // byte != 0
                sixteen |= (((nz ^ 0x01010101u) * 0x01020408u) >> 24) << (4 * k);                          // byte i == 0 -> bit i
            }
            if (16 * j < valid) z |= (unsigned long long)(sixteen & 0xFFFFu) << (16 * j);
        }
    } else {
        for (int k = 0; k < valid; ++k) z |= (unsigned long long)(p[k] == 0) << k;
    }
    zw[i] = z;  // pixels beyond W: "not zero"
}

// distance of (y, x) to the nearest zero pixel of row y (G_INF if the row has none), from the row's zero-flag words
__device__ __forceinline__ unsigned edt_g_at(const unsigned long long* __restrict__ zr, int WW, int x) {
    const int wi = x >> 6, bit = x & 63;
    const unsigned long long own = zr[wi];
    if ((own >> bit) & 1ull) return 0u;
    unsigned dl = G_INF, dr = G_INF;
    unsigned long long m = own & ((1ull << bit) - 1ull);
    int w = wi;
    while (m == 0 && w > 0) m = zr[--w];
    if (m) dl = (unsigned)(x - (w * 64 + 63 - __clzll((long long)m)));
    m = bit == 63 ? 0ull : (own >> (bit + 1)) << (bit + 1);
    w = wi;
    while (m == 0 && w + 1 < WW) m = zr[++w];
    if (m) dr = (unsigned)(w * 64 + __ffsll((long long)m) - 1 - x);
    const unsigned d = dl < dr ? dl : dr;
    return d >= G_INF ? G_INF : d;
}

template <int EC_ROWS, int EC_HALO>
__global__ void __launch_bounds__(256) edt_bits_kernel(const unsigned long long* __restrict__ zw, int* __restrict__ d2_out,
                                                       double* __restrict__ edt_out, int H, int W, int WW) {
    constexpr int EC_TROWS = EC_ROWS + 2 * EC_HALO;
    __shared__ __attribute__((aligned(16))) unsigned short tile[EC_TROWS][64];
    __shared__ unsigned long long zs[EC_TROWS][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = blockIdx.x;  // the tile's word column
    const int y0 = blockIdx.y * EC_ROWS;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const unsigned long long* zp = zw + (size_t)blockIdx.z * H * WW;
    // ---- the zero flags of the window: own word, left and right neighbour (outside the image: no zero) ----
    for (int i = threadIdx.x; i < EC_TROWS * 3; i += 256) {
        const int r = i / 3, c = i - r * 3;
        const int y = y0 - EC_HALO + r, w = wi - 1 + c;
        zs[r][c] = (y >= 0 && y < H && w >= 0 && w < WW) ? zp[(size_t)y * WW + w] : 0ull;
    }
    __syncthreads();
    // ---- g of the window: a thread owns 8 consecutive pixels of a row (edt_rows8_kernel's arithmetic) ----
    constexpr unsigned BIG = 0x20000u;  // > any distance inside a row (sides are <= 32768)
#if defined(EDT_EXP) && EDT_EXP == 2
    for (int task = threadIdx.x; task < EC_TROWS * 8; task += 256) *reinterpret_cast<uint4*>(&tile[task >> 3][(task & 7) * 8]) = make_uint4(0, 0, 0, 0);
    if (false)
#endif
    for (int task = threadIdx.x; task < EC_TROWS * 8; task += 256) {
        const int r = task >> 3, b = task & 7;
        const int y = y0 - EC_HALO + r;
        uint4 out = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // rows outside the image: G_INF
        if (y >= 0 && y < H) {
            const int bit0 = b * 8, x0 = wi * 64 + bit0;
            const unsigned long long own = zs[r][1];
            const unsigned z8 = (unsigned)(own >> bit0) & 0xffu;
            out = make_uint4(0u, 0u, 0u, 0u);
            if (z8 != 0xffu) {
                unsigned dl = BIG, dr = BIG;  // distance of pixel x0 to the nearest zero left of the group / of x0+7 right
                unsigned long long m = own & ((1ull << bit0) - 1ull);
                int w = wi;
                if (m == 0 && w > 0) {
                    m = zs[r][0];
                    --w;
                    const unsigned long long* zr = zp + (size_t)y * WW;
                    while (m == 0 && w > 0) m = zr[--w];
                }
                if (m) dl = (unsigned)(x0 - (w * 64 + 63 - __clzll((long long)m)));
                const int sh = bit0 + 8;
                m = sh == 64 ? 0ull : (own >> sh) << sh;
                w = wi;
                if (m == 0 && w + 1 < WW) {
                    m = zs[r][2];
                    ++w;
                    const unsigned long long* zr = zp + (size_t)y * WW;
                    while (m == 0 && w + 1 < WW) m = zr[++w];
                }
                if (m) dr = (unsigned)(w * 64 + __ffsll((long long)m) - 1 - (x0 + 7));
                unsigned L[8], R[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned prev = i == 0 ? dl : L[i - 1] + 1u;
                    L[i] = ((z8 >> i) & 1u) ? 0u : (prev < BIG ? prev : BIG);
                }
#pragma unroll
                for (int i = 7; i >= 0; --i) {
                    const unsigned nxt = i == 7 ? dr : R[i + 1] + 1u;
                    R[i] = ((z8 >> i) & 1u) ? 0u : (nxt < BIG ? nxt : BIG);
                }
                unsigned d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned mn = L[i] < R[i] ? L[i] : R[i];
                    d[i] = mn >= G_INF ? G_INF : mn;
                }
                out.x = d[0] | (d[1] << 16);
                out.y = d[2] | (d[3] << 16);
                out.z = d[4] | (d[5] << 16);
                out.w = d[6] | (d[7] << 16);
            }
        }
        *reinterpret_cast<uint4*>(&tile[r][b * 8]) = out;
    }
    __syncthreads();
    // ---- the column search (edt_cols_kernel's, with g beyond the halo evaluated from the words).  A lane owns FOUR
    // consecutive pixels of a row: one 8-byte LDS read brings the four g of a row above / below, the four searches are
    // independent chains that share every LDS wait, and the results leave in one 16-byte store (a wave writes 4 rows x
    // 256 bytes; with a pixel per lane the stores alone took 223 us per 48 planes, 3.6 TB/s) ----
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int xg = wi * 64 + c4;
    if (xg >= W) return;
    const bool vec = xg + 3 < W && (W & 3) == 0 && (!d2_out || (reinterpret_cast<uintptr_t>(d2_out) & 15) == 0) &&
                     (!edt_out || (reinterpret_cast<uintptr_t>(edt_out) & 15) == 0);
#pragma unroll 1
    for (int j = 0; j < EC_ROWS / 16; ++j) {
        const int ly = EC_HALO + wave * (EC_ROWS / 4) + rsub + 4 * j;
        const int y = y0 - EC_HALO + ly;
        if (y >= H) continue;
        unsigned best[4];
        {
            const uint2 gq = *reinterpret_cast<const uint2*>(&tile[ly][c4]);
            const unsigned g0[4] = {gq.x & 0xFFFFu, gq.x >> 16, gq.y & 0xFFFFu, gq.y >> 16};
#pragma unroll
            for (int i = 0; i < 4; ++i) best[i] = g0[i] == G_INF ? 0xFFFFFFFFu : g0[i] * g0[i];
        }
#if defined(EDT_EXP) && EDT_EXP == 1
        if (false)
#endif
        for (unsigned k = 1; k <= (unsigned)EC_HALO; ++k) {
            const unsigned kk = k * k;
            if (!(kk < best[0] || kk < best[1] || kk < best[2] || kk < best[3])) break;
            const uint2 uq = *reinterpret_cast<const uint2*>(&tile[ly - (int)k][c4]);
            const uint2 dq = *reinterpret_cast<const uint2*>(&tile[ly + (int)k][c4]);
            const unsigned gu[4] = {uq.x & 0xFFFFu, uq.x >> 16, uq.y & 0xFFFFu, uq.y >> 16};
            const unsigned gd[4] = {dq.x & 0xFFFFu, dq.x >> 16, dq.y & 0xFFFFu, dq.y >> 16};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned gm = gu[i] < gd[i] ? gu[i] : gd[i];
                // gm == G_INF: 65535^2 + k^2 < 2^32 stays above every finite candidate and below the "no zero" mark
                const unsigned c = gm == G_INF ? 0xFFFFFFFFu : kk + gm * gm;
                best[i] = c < best[i] ? c : best[i];  // a candidate at distance k cannot beat best <= k^2: no need to mask
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned bq = best[i];
            const int x = xg + i;
            if (bq != 0 && x < W) {
                for (unsigned kk = EC_HALO + 1; kk < 65536u && (unsigned long long)kk * kk < bq; ++kk) {  // beyond the LDS halo
                    const int yu = y - (int)kk, yd = y + (int)kk;
                    if (yu < 0 && yd >= H) break;
                    unsigned gm = G_INF;
                    if (yu >= 0) gm = edt_g_at(zp + (size_t)yu * WW, WW, x);
                    if (yd < H) {
                        const unsigned gd = edt_g_at(zp + (size_t)yd * WW, WW, x);
                        gm = gd < gm ? gd : gm;
                    }
                    if (gm != G_INF) {
                        const unsigned c = kk * kk + gm * gm;  // < 2^31: both terms < 2^30
                        bq = c < bq ? c : bq;
                    }
                }
                // no zero pixel in the whole plane: scipy's feature transform then measures from index (-1, 0)
                if (bq > 0x7fffffffu) bq = (unsigned)(y + 1) * (unsigned)(y + 1) + (unsigned)x * (unsigned)x;
            }
            best[i] = bq;
        }
        const size_t i0 = plane + (size_t)y * W + xg;
        if (vec) {
            if (d2_out) *reinterpret_cast<int4*>(d2_out + i0) = make_int4((int)best[0], (int)best[1], (int)best[2], (int)best[3]);
            if (edt_out) {
                *reinterpret_cast<double2*>(edt_out + i0) = make_double2(sqrt((double)best[0]), sqrt((double)best[1]));
                *reinterpret_cast<double2*>(edt_out + i0 + 2) = make_double2(sqrt((double)best[2]), sqrt((double)best[3]));
            }
        } else {
            for (int i = 0; i < 4 && xg + i < W; ++i) {
                if (d2_out) d2_out[i0 + i] = (int)best[i];
                if (edt_out) edt_out[i0 + i] = sqrt((double)best[i]);
            }
        }
    }
}

// AMT_EDT_FUSED=0: the two-pass transform through a uint16 plane of row distances (A/B switch; identical results)
static bool edt_fused_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_EDT_FUSED");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

extern "C" int amt_edt(amt_ctx* ctx, const uint8_t* mask, int32_t* d2_out, double* edt_out, int nplanes, int H,
                       int W) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(mask && (d2_out || edt_out) && nplanes >= 0 && H > 0 && W > 0, "edt: bad arguments");
    AMT_REQUIRE(H <= 32768 && W <= 32768, "edt: image larger than 32768 pixels per side");
    if (nplanes == 0) return AMT_OK;
    const size_t n = (size_t)H * W;
    const size_t tiles64 = (size_t)((W + 63) / 64) * ((H + 63) / 64) * nplanes;
    if (edt_fused_enabled()) {
        const int WW = (W + 63) / 64;
        const size_t nwords = (size_t)nplanes * H * WW;
        AMT_TRY(amt_arena_begin(ctx, amt_align(nwords * 8)));
        unsigned long long* zw = arena_take_t<unsigned long long>(ctx, nwords);
        const int fast = W % 16 == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
        hipLaunchKernelGGL(edt_zero_words_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, ctx->stream, mask, zw, H,
                           W, WW, nwords, fast);
        AMT_LAUNCH_CHECK();
        if (tiles64 >= 4096 && !edt_out)
            hipLaunchKernelGGL((edt_bits_kernel<64, 16>), dim3(WW, (H + 63) / 64, nplanes), dim3(256), 0, ctx->stream, zw, d2_out,
                               edt_out, H, W, WW);
        else
            hipLaunchKernelGGL((edt_bits_kernel<32, 24>), dim3(WW, (H + 31) / 32, nplanes), dim3(256), 0, ctx->stream, zw, d2_out,
                               edt_out, H, W, WW);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    AMT_TRY(amt_arena_begin(ctx, amt_align((size_t)nplanes * n * 2)));
    unsigned short* g = arena_take_t<unsigned short>(ctx, (size_t)nplanes * n);
    if ((W & 7) == 0 && (reinterpret_cast<uintptr_t>(mask) & 7) == 0)  // g comes from the arena: 256-byte aligned
        hipLaunchKernelGGL(edt_rows8_kernel, dim3(H, nplanes), dim3(256), (size_t)((W + 63) / 64) * 8, ctx->stream, mask,
                           g, H, W);
    else
        hipLaunchKernelGGL(edt_rows_kernel, dim3(H, nplanes), dim3(256), (size_t)((W + 63) / 64) * 8, ctx->stream, mask,
                           g, H, W);
    AMT_LAUNCH_CHECK();
    if (tiles64 >= 4096 && !edt_out)  // with the float64 output (sqrt + 8-byte stores) the shorter tiles measured faster
        hipLaunchKernelGGL((edt_cols_kernel<64, 16>), dim3((W + 63) / 64, (H + 63) / 64, nplanes), dim3(256), 0,
                           ctx->stream, g, d2_out, edt_out, H, W);
    else
        hipLaunchKernelGGL((edt_cols_kernel<32, 24>), dim3((W + 63) / 64, (H + 31) / 32, nplanes), dim3(256), 0,
                           ctx->stream, g, d2_out, edt_out, H, W);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---- peak mask ---------------------------------------------------------------------------------
// peaks = (d2 == maximum_filter(d2, size=2m+1, mode='constant' (0))) & mask & (d2 > 0), border m cleared
// (comparing the integer d2 is equivalent to comparing sqrt(d2)).
// "d2 equals the window maximum" == "nothing in the window exceeds d2", so no maximum image is built: a pixel
// is first compared with its 8 neighbours (which rejects all but a few per nucleus), and only the survivors look
// at the full (2m+1)^2 window.
constexpr int PK_MAXM = 16;
// No LDS: a wave slides a three-row register window down a strip of PKR_IN columns (+ one halo
// column on either side); the 8-neighbour pre-test uses DPP wave shifts, and the few survivors (a handful per
// nucleus) are checked one at a time by the WHOLE wave: 64 lanes read the (2m+1)^2 window straight from global
// memory (L2 hits: the strip has just streamed those rows) and vote.  The host clears `peaks` first; only peak
// pixels are written.
constexpr int PKR_ROWS = 64, PKR_IN = 62, PKR_BATCH = 22, PKR_NBATCH = (PKR_ROWS + 2) / PKR_BATCH;
static_assert(PKR_BATCH * PKR_NBATCH == PKR_ROWS + 2, "row batches must tile the strip");

__global__ void __launch_bounds__(256) peaks_rows_kernel(const int* __restrict__ d2, const uint8_t* __restrict__ mask,
                                                         uint8_t* __restrict__ peaks, int H, int W, int m) {
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip * PKR_IN >= W) return;  // whole wave; the kernel has no barrier
    const size_t base = (size_t)blockIdx.z * H * W;
    const int* D = d2 + base;
    const int x = strip * PKR_IN - 1 + lane;
    const int y0 = blockIdx.y * PKR_ROWS;
    const bool xin = x >= 0 && x < W;
    const bool xok = xin && lane >= 1 && lane <= PKR_IN && x >= m && x < W - m;  // output lane inside the cleared frame
    const int side = 2 * m + 1, wsize = side * side;

    // UNCONDITIONAL loads from clamped coordinates; positions outside the image become the constant 0 where the
    // value is used (a select next to the load would be turned back into a branch around it, and the wait that
    // comes with it serialises the loads)
    const int xc = x < 0 ? 0 : (x < W ? x : W - 1);
    auto load_row = [&](int y) -> int {
        const int yc = y < 0 ? 0 : (y < H ? y : H - 1);
        return D[(size_t)yc * W + xc];
    };
    int cur[PKR_BATCH], nxt[PKR_BATCH];
#pragma unroll
    for (int j = 0; j < PKR_BATCH; ++j) cur[j] = load_row(y0 - 1 + j);
    // window: v1 = row r-1 (the centre row of this step) with its neighbours, h2 = 3-wide maximum of row r-2
    int v1 = 0, v1l = 0, v1r = 0, h1 = 0, h2 = 0;
    for (int b = 0; b < PKR_NBATCH; ++b) {
        if (b + 1 < PKR_NBATCH) {
#pragma unroll
            for (int j = 0; j < PKR_BATCH; ++j) nxt[j] = load_row(y0 - 1 + (b + 1) * PKR_BATCH + j);
        }
#pragma unroll
        for (int j = 0; j < PKR_BATCH; ++j) {
            const int r = y0 - 1 + b * PKR_BATCH + j;  // row of v0
            const int v0 = (xin && r >= 0 && r < H) ? cur[j] : 0;
            if (__ballot(v0 > 0 || v1 > 0) == 0ull) {  // uniform: no candidate in row r-1, nothing to carry from row r
                h2 = h1, h1 = 0, v1 = 0, v1l = 0, v1r = 0;  // (non-positive values never beat a candidate)
                continue;
            }
            const int v0l = amt_lane_left(v0), v0r = amt_lane_right(v0);
            int h0 = v0 > v0l ? v0 : v0l;
            h0 = h0 > v0r ? h0 : v0r;
            const int yc = r - 1;  // centre row
            bool cand = xok && v1 > 0 && yc >= y0 && yc < y0 + PKR_ROWS && yc >= m && yc < H - m;
            if (m >= 1) {
                int nb = h0 > h2 ? h0 : h2;
                nb = nb > v1l ? nb : v1l;
                nb = nb > v1r ? nb : v1r;
                cand = cand && !(nb > v1);
            }
            unsigned long long todo = __ballot(cand);
            while (todo) {  // uniform: one survivor at a time, the whole wave reads its window
                const int sl = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int sv = __shfl(v1, sl);
                const int sx = strip * PKR_IN - 1 + sl;
                bool gt = false;
                for (int k = lane; k < wsize; k += 64) {
                    const int dy = k / side, dx = k - dy * side;
                    gt |= D[(size_t)(yc - m + dy) * W + (sx - m + dx)] > sv;  // inside the image: the frame is cleared
                }
                if (__ballot(gt) == 0ull && lane == sl && mask[base + (size_t)yc * W + sx])
                    peaks[base + (size_t)yc * W + sx] = 1;
            }
            h2 = h1, h1 = h0;
            v1 = v0, v1l = v0l, v1r = v0r;
        }
#pragma unroll
        for (int j = 0; j < PKR_BATCH; ++j) cur[j] = nxt[j];
    }
}

// peaks[list[k]] = 0 for the listed pixels of every plane
// (status[plane] < 0: that run's list overflowed and does not hold every peak -- the whole plane is cleared)
__global__ void __launch_bounds__(256) peaks_unwrite_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                                            const int* __restrict__ status, uint8_t* __restrict__ peaks,
                                                            size_t n, int cap) {
    uint8_t* o = peaks + (size_t)blockIdx.y * n;
    if (status[blockIdx.y] < 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = 0;
        return;
    }
    const int K = count[blockIdx.y] < cap ? count[blockIdx.y] : cap;
    const int* lst = list + (size_t)blockIdx.y * cap;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) o[lst[k]] = 0;
}

static int peak_mask_impl(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H, int W,
                          int min_distance, const int32_t* prev_list, const int32_t* prev_count, int capacity,
                          const int32_t* prev_status);

extern "C" int amt_peak_mask(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H,
                             int W, int min_distance) {
    return peak_mask_impl(ctx, d2, mask, peaks, nplanes, H, W, min_distance, nullptr, nullptr, 0, nullptr);
}

extern "C" int amt_peak_mask_reuse(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes,
                                   int H, int W, int min_distance, const int32_t* prev_list, const int32_t* prev_count,
                                   int capacity, const int32_t* prev_status) {
    AMT_REQUIRE(prev_list && prev_count && prev_status && capacity >= 1,
                "peak_mask_reuse: the previous run's peak lists, counts and label counts are required");
    return peak_mask_impl(ctx, d2, mask, peaks, nplanes, H, W, min_distance, prev_list, prev_count, capacity, prev_status);
}

static int peak_mask_impl(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H, int W,
                          int min_distance, const int32_t* prev_list, const int32_t* prev_count, int capacity,
                          const int32_t* prev_status) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(d2 && mask && peaks && nplanes >= 0 && H > 0 && W > 0, "peak_mask: bad arguments");
    AMT_REQUIRE(min_distance >= 0 && min_distance <= PK_MAXM, "peak_mask: min_distance %d out of range 0..%d",
                min_distance, PK_MAXM);
    if (nplanes == 0) return AMT_OK;
    const int m = min_distance;
    if (prev_list) {
        // the plane is zero except at the previous run's peaks, which amt_label_sparse_reuse kept as a list
        hipLaunchKernelGGL(peaks_unwrite_kernel, dim3(amt_grid_for((size_t)capacity, 256, 64), nplanes), dim3(256), 0,
                           ctx->stream, prev_list, prev_count, prev_status, peaks, (size_t)H * W, capacity);
        AMT_LAUNCH_CHECK();
    } else {
        AMT_HIP_CHECK(hipMemsetAsync(peaks, 0, (size_t)nplanes * H * W, ctx->stream));
    }
    const int nstrips = (W + PKR_IN - 1) / PKR_IN;
    dim3 grid((nstrips + 3) / 4, (H + PKR_ROWS - 1) / PKR_ROWS, nplanes);
    hipLaunchKernelGGL(peaks_rows_kernel, grid, dim3(256), 0, ctx->stream, d2, mask, peaks, H, W, m);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
