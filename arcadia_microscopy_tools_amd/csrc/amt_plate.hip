// Per-plate feature rows: the dense per-FOV tables of one rank compacted to one row per cell, the block a rank
// contributes to the plate's all-gather (row counts first, then rows).
#include "amt_common.h"

// One workgroup handles `ROWS` table rows of one field of view.  Its first output row = the number of cells of all
// earlier fields of view, summed by the block itself (B is a few hundred int32 words that sit in L2).
template <int ROWS>
__global__ void __launch_bounds__(256) pack_rows_kernel(const double* __restrict__ table, const double* __restrict__ itable,
                                                        const int* __restrict__ ncells, int B, int K, int C,
                                                        const int* __restrict__ fov_index, int fov_index0,
                                                        double* __restrict__ rows, long long* __restrict__ nrows) {
    __shared__ int s_part[4];
    __shared__ int s_bad;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int before = 0, bad = 0;
    // the last block of the grid also adds the fields of view behind it: it publishes the total
    const bool is_last = (b == B - 1) && (blockIdx.y == gridDim.y - 1);
    const int upto = is_last ? B : b;
    for (int j = threadIdx.x; j < upto; j += 256) {
        const int n = ncells[j];
        if (n < 0 || n > K) bad = 1;
        else if (j < b) before += n;
    }
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if (bad) atomicOr(&s_bad, 1);
    if (lane == 0) s_part[wave] = before;
    __syncthreads();
    const int row0 = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    int n = ncells[b];
    const bool own_bad = n < 0 || n > K;
    if (own_bad) n = 0;  // an overflowed field of view contributes no rows; the total is flagged below
    if (is_last && threadIdx.x == 0) *nrows = (s_bad || own_bad) ? -1ll : (long long)row0 + n;
    const int ncols = 2 + AMT_RP_NCOLS + 4 * C;
    const int r_lo = blockIdx.y * ROWS;
    const int r_hi = min(n, r_lo + ROWS);
    if (r_lo >= r_hi) return;
    const double fov = (double)(fov_index ? fov_index[b] : fov_index0 + b);
    const int total = (r_hi - r_lo) * ncols;
    const double* t = table + ((size_t)b * K + r_lo) * AMT_RP_NCOLS;
    const double* it = itable ? itable + ((size_t)b * K + r_lo) * C * 4 : nullptr;
    double* o = rows + ((size_t)row0 + r_lo) * ncols;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        double v;
        if (c == 0) v = fov;
        else if (c == 1) v = (double)(r_lo + r + 1);
        else if (c < 2 + AMT_RP_NCOLS) v = t[(size_t)r * AMT_RP_NCOLS + (c - 2)];
        else v = it ? it[(size_t)r * C * 4 + (c - 2 - AMT_RP_NCOLS)] : 0.0;
        o[e] = v;
    }
}

extern "C" int amt_pack_plate_rows(amt_ctx* ctx, const double* table_dev, const double* itable_dev,
                                   const int32_t* ncells_dev, int B, int K, int C, const int32_t* fov_index_dev,
                                   int fov_index0, double* rows_dev, size_t rows_cap, int64_t* nrows_dev) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(table_dev && ncells_dev && rows_dev && nrows_dev && B >= 0 && K >= 1 && C >= 0,
                "pack_plate_rows: bad arguments");
    AMT_REQUIRE(C == 0 || itable_dev, "pack_plate_rows: C > 0 needs the intensity table");
    AMT_REQUIRE(rows_cap >= (size_t)B * K, "pack_plate_rows: rows_cap %zu < B * K = %zu (every field of view may be full)",
                rows_cap, (size_t)B * K);
    if (B == 0) {
        AMT_HIP_CHECK(hipMemsetAsync(nrows_dev, 0, sizeof(int64_t), ctx->stream));
        return AMT_OK;
    }
    constexpr int ROWS = 64;
    hipLaunchKernelGGL((pack_rows_kernel<ROWS>), dim3(B, (K + ROWS - 1) / ROWS), dim3(256), 0, ctx->stream, table_dev,
                       C ? itable_dev : nullptr, ncells_dev, B, K, C, fov_index_dev, fov_index0, rows_dev,
                       (long long*)nrows_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
