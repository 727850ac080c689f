// Internal helpers shared by the HIP translation units of libamt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/amt_hip.h"

struct amt_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    // grow-only scratch arena; reset at the start of every public op
    char* arena;
    size_t arena_cap;
    size_t arena_off;
    // pinned host ring for small parameter tables (weights, footprint offsets, rank requests): the
    // caller's pointer is consumed before the public function returns, the DMA reads the pinned copy.
    char* mailbox;
    size_t mailbox_cap;
    size_t mailbox_off;
    int num_cus;
    // auxiliary streams + events for fork/join of independent latency-bound kernels inside one op
    hipStream_t aux[3];      // the streams an op's independent kernels are launched on (auxiliary ones, or `stream`)
    hipStream_t aux_own[3];  // the auxiliary streams this context created
    hipEvent_t ev[4];
    bool aux_ready;
    int fork;  // amt_ctx_set_fork: how many auxiliary streams the ops of this context use (0..3)
};

// fork: aux streams wait for everything enqueued so far on the main stream; join: main waits for them
int amt_fork(amt_ctx* ctx);
int amt_join(amt_ctx* ctx);

void amt_set_error(const char* fmt, ...);

#define AMT_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            amt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AMT_EHIP;                                                                     \
        }                                                                                        \
    } while (0)

#define AMT_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            amt_set_error(__VA_ARGS__); \
            return AMT_EINVAL;          \
        }                               \
    } while (0)

#define AMT_LAUNCH_CHECK()                                                                        \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess) {                                                                   \
            amt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return AMT_EHIP;                                                                      \
        }                                                                                         \
    } while (0)

#define AMT_TRY(expr)             \
    do {                          \
        int _rc = (expr);         \
        if (_rc != AMT_OK) return _rc; \
    } while (0)

// ---- arena -----------------------------------------------------------------------------------
// Ops call arena_begin(ctx, total_bytes) once (may reallocate: synchronises the stream first),
// then carve with arena_take.  Memory is only valid until the next op on the same context, which
// is safe because all work of one context is ordered on one stream.
int amt_arena_begin(amt_ctx* ctx, size_t total_bytes);
// copy `bytes` from caller-owned host memory to device memory via the pinned ring (stream ordered)
int amt_param_upload(amt_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
void* amt_arena_take(amt_ctx* ctx, size_t bytes);

__host__ __device__ static inline size_t amt_align(size_t n, size_t a = 256) { return (n + a - 1) / a * a; }

template <typename T>
static inline T* arena_take_t(amt_ctx* ctx, size_t count) {
    return reinterpret_cast<T*>(amt_arena_take(ctx, amt_align(count * sizeof(T))));
}

static inline int amt_set_device(amt_ctx* ctx) {
    if (!ctx) {
        amt_set_error("null context");
        return AMT_EINVAL;
    }
    AMT_HIP_CHECK(hipSetDevice(ctx->device));
    return AMT_OK;
}

static inline unsigned amt_grid_for(size_t work_items, unsigned block, unsigned max_blocks = 8192) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

// ---- device helpers ----------------------------------------------------------------------------
// scipy.ndimage index extension.  Returns -1 for AMT_MODE_CONSTANT out of range.
__device__ __forceinline__ int amt_map_index(int i, int n, int mode) {
    if (i >= 0 && i < n) return i;
    switch (mode) {
        case AMT_MODE_NEAREST:
            return i < 0 ? 0 : n - 1;
        case AMT_MODE_REFLECT: {
            int p = 2 * n;
            int m = i % p;
            if (m < 0) m += p;
            return m < n ? m : p - 1 - m;
        }
        case AMT_MODE_MIRROR: {
            if (n == 1) return 0;
            int p = 2 * n - 2;
            int m = i % p;
            if (m < 0) m += p;
            return m < n ? m : p - m;
        }
        case AMT_MODE_WRAP: {
            int m = i % n;
            if (m < 0) m += n;
            return m;
        }
        default:
            return -1;
    }
}

// order-preserving map float64 <-> uint64 (for atomic min/max and radix select)
__device__ __forceinline__ unsigned long long amt_f64_key(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double amt_key_f64(unsigned long long k) {
    unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
