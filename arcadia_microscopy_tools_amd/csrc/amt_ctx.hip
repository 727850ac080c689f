// Context, memory, timers and error plumbing of libamt_hip.so.
#include "amt_common.h"

#include <cstdlib>

static thread_local char g_err[1024] = "";

void amt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* amt_last_error(void) { return g_err; }
extern "C" const char* amt_version(void) { return "amt_hip 0.1 (gfx950)"; }

extern "C" int amt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int fork_default();

static int ctx_create_common(int device, hipStream_t stream, bool own, amt_ctx** out) {
    AMT_REQUIRE(out != nullptr, "amt_ctx_create: out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        amt_set_error("no HIP device available (%s); libamt_hip requires an MI355X (gfx950)",
                      e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return AMT_ENODEV;
    }
    AMT_REQUIRE(device >= 0 && device < n, "amt_ctx_create: device %d out of range (have %d)", device, n);
    {
        // AMT_SYNC_MODE=yield|block: how a host thread waits in amt_sync (default: the runtime's choice, a spin).  Only
        // takes effect if nothing has initialised the device yet (A/B switch for many worker threads on few cores)
        static bool once = false;
        const char* m = getenv("AMT_SYNC_MODE");
        if (!once && m) {
            once = true;
            (void)hipSetDeviceFlags(m[0] == 'b' ? hipDeviceScheduleBlockingSync : m[0] == 'y' ? hipDeviceScheduleYield : hipDeviceScheduleSpin);
            (void)hipGetLastError();
        }
    }
    AMT_HIP_CHECK(hipSetDevice(device));
    amt_ctx* c = new amt_ctx();
    c->device = device;
    c->own_stream = own;
    c->stream = stream;
    if (own) {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            amt_set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
            return AMT_EHIP;
        }
    }
    c->arena = nullptr;
    c->arena_cap = 0;
    c->arena_off = 0;
    c->mailbox = nullptr;
    c->mailbox_cap = 0;
    c->mailbox_off = 0;
    c->aux_ready = false;
    c->fork = fork_default();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        c->num_cus = prop.multiProcessorCount;
    else
        c->num_cus = 256;
    c->mailbox_cap = 1u << 20;
    if (hipHostMalloc((void**)&c->mailbox, c->mailbox_cap, hipHostMallocDefault) != hipSuccess) {
        c->mailbox = nullptr;
        c->mailbox_cap = 0;
    }
    *out = c;
    return AMT_OK;
}

extern "C" int amt_ctx_create(int device, amt_ctx** out) { return ctx_create_common(device, nullptr, true, out); }

extern "C" int amt_ctx_create_on_stream(int device, void* hip_stream, amt_ctx** out) {
    return ctx_create_common(device, (hipStream_t)hip_stream, false, out);
}

extern "C" int amt_ctx_destroy(amt_ctx* ctx) {
    if (!ctx) return AMT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->aux_ready) {
        for (int i = 0; i < 3; ++i) (void)hipStreamDestroy(ctx->aux_own[i]);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->ev[i]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return AMT_OK;
}

extern "C" int amt_device_name(amt_ctx* ctx, char* buf, int buflen) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(buf && buflen > 0, "amt_device_name: bad buffer");
    hipDeviceProp_t prop;
    AMT_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return AMT_OK;
}

// AMT_DEBUG_POISON=1 (diagnostic): every allocation and, at the start of every op, the whole scratch arena are filled
// with 0xCD bytes, so a kernel that relies on "fresh memory is zero" fails deterministically instead of once in a while.
static bool poison_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_DEBUG_POISON");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

int amt_arena_begin(amt_ctx* ctx, size_t total_bytes) {
    total_bytes = amt_align(total_bytes) + 4096;
    if (total_bytes > ctx->arena_cap) {
        // previous users of the arena are ordered before us on the stream; drain it before freeing
        AMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->arena) AMT_HIP_CHECK(hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_cap = 0;
        size_t cap = total_bytes + total_bytes / 4;
        hipError_t e = hipMalloc((void**)&ctx->arena, cap);
        if (e != hipSuccess) {
            amt_set_error("scratch arena allocation of %zu bytes failed: %s", cap, hipGetErrorString(e));
            return AMT_ENOMEM;
        }
        ctx->arena_cap = cap;
    }
    ctx->arena_off = 0;
    if (poison_enabled()) AMT_HIP_CHECK(hipMemsetAsync(ctx->arena, 0xCD, ctx->arena_cap, ctx->stream));
    return AMT_OK;
}

// AMT_FORK=0 keeps every kernel of an op on the context's single stream (no auxiliary streams); 1..3 = that many
static int fork_default() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AMT_FORK");
        v = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 3;
    }
    return v;
}

// Per context: a process that already runs several contexts side by side (one per HIP stream) gains nothing from the
// auxiliary streams -- their events and the extra hardware queues cost more than the overlap inside one op brings
// (bench.py, 4 contexts: 11.1 k FOV/s without, 10.2 k with) -- while a single context needs them (watershed stage
// 1.80 ms against 2.32 ms per 32 FOVs).
extern "C" int amt_ctx_set_fork(amt_ctx* ctx, int enable) {
    AMT_REQUIRE(ctx != nullptr, "ctx_set_fork: null context");
    ctx->fork = enable < 0 ? 0 : (enable > 3 ? 3 : enable);  // 1 ("on") .. 3 = that many auxiliary streams
    return AMT_OK;
}

int amt_fork(amt_ctx* ctx) {
    const int n = ctx->fork;
    if (n == 0) {
        ctx->aux[0] = ctx->aux[1] = ctx->aux[2] = ctx->stream;
        return AMT_OK;
    }
    if (!ctx->aux_ready) {
        for (int i = 0; i < 3; ++i) AMT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux_own[i], hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) AMT_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming));
        ctx->aux_ready = true;
    }
    for (int i = 0; i < 3; ++i) ctx->aux[i] = ctx->aux_own[i % n];
    AMT_HIP_CHECK(hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < n; ++i) AMT_HIP_CHECK(hipStreamWaitEvent(ctx->aux_own[i], ctx->ev[0], 0));
    return AMT_OK;
}

int amt_join(amt_ctx* ctx) {
    const int n = ctx->fork;
    if (n == 0) return AMT_OK;
    for (int i = 0; i < n; ++i) {
        AMT_HIP_CHECK(hipEventRecord(ctx->ev[1 + i], ctx->aux_own[i]));
        AMT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev[1 + i], 0));
    }
    return AMT_OK;
}

int amt_param_upload(amt_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (bytes == 0) return AMT_OK;
    size_t need = amt_align(bytes, 256);
    if (!ctx->mailbox || need > ctx->mailbox_cap) {
        // too large for the ring: synchronous copy (the caller's buffer is consumed before returning)
        AMT_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        AMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return AMT_OK;
    }
    if (ctx->mailbox_off + need > ctx->mailbox_cap) {
        // wrap: older slots may still be in flight
        AMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->mailbox_off = 0;
    }
    char* slot = ctx->mailbox + ctx->mailbox_off;
    ctx->mailbox_off += need;
    memcpy(slot, host_src, bytes);
    AMT_HIP_CHECK(hipMemcpyAsync(dev_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return AMT_OK;
}

void* amt_arena_take(amt_ctx* ctx, size_t bytes) {
    bytes = amt_align(bytes);
    void* p = ctx->arena + ctx->arena_off;
    ctx->arena_off += bytes;
    return p;
}

extern "C" int amt_malloc(amt_ctx* ctx, size_t bytes, void** dptr) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(dptr != nullptr, "amt_malloc: dptr is null");
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        amt_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return AMT_ENOMEM;
    }
    if (poison_enabled()) {
        AMT_HIP_CHECK(hipMemsetAsync(*dptr, 0xCD, bytes, ctx->stream));
        AMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return AMT_OK;
}

extern "C" int amt_free(amt_ctx* ctx, void* dptr) {
    AMT_TRY(amt_set_device(ctx));
    if (dptr) AMT_HIP_CHECK(hipFree(dptr));
    return AMT_OK;
}

extern "C" int amt_memcpy_h2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    AMT_TRY(amt_set_device(ctx));
    if (bytes) AMT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_memcpy_d2h(amt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    AMT_TRY(amt_set_device(ctx));
    if (bytes) AMT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_memcpy_d2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    AMT_TRY(amt_set_device(ctx));
    if (bytes) AMT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_memset(amt_ctx* ctx, void* dst, int value, size_t bytes) {
    AMT_TRY(amt_set_device(ctx));
    if (bytes) AMT_HIP_CHECK(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_sync(amt_ctx* ctx) {
    AMT_TRY(amt_set_device(ctx));
    AMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return AMT_OK;
}

extern "C" int amt_ctx_stream(amt_ctx* ctx, void** hip_stream) {
    AMT_REQUIRE(ctx && hip_stream, "amt_ctx_stream: null argument");
    *hip_stream = (void*)ctx->stream;
    return AMT_OK;
}

extern "C" int amt_stream_wait(amt_ctx* ctx, amt_ctx* other) {
    // everything enqueued so far on `other`'s stream happens-before what is enqueued next on `ctx`'s stream
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(other != nullptr, "amt_stream_wait: other is null");
    AMT_REQUIRE(other->device == ctx->device, "amt_stream_wait: contexts on different devices");
    hipEvent_t ev;
    AMT_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    AMT_HIP_CHECK(hipEventRecord(ev, other->stream));
    AMT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
    AMT_HIP_CHECK(hipEventDestroy(ev));  // destruction is deferred until the event has completed
    return AMT_OK;
}

// ---- events: ordering between contexts at a RECORDED point (amt_stream_wait orders after everything so far) ----
extern "C" int amt_event_create(amt_ctx* ctx, void** event) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(event != nullptr, "amt_event_create: event is null");
    hipEvent_t ev;
    AMT_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *event = (void*)ev;
    return AMT_OK;
}
extern "C" int amt_event_record(amt_ctx* ctx, void* event) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(event != nullptr, "amt_event_record: event is null");
    AMT_HIP_CHECK(hipEventRecord((hipEvent_t)event, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_event_wait(amt_ctx* ctx, void* event) {
    // what is enqueued next on ctx's stream happens after the event's last record (no-op if never recorded)
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(event != nullptr, "amt_event_wait: event is null");
    AMT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return AMT_OK;
}
extern "C" int amt_event_sync(amt_ctx* ctx, void* event) {
    // the HOST waits for the event's last record: a chunked download hands finished chunks to host threads
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(event != nullptr, "amt_event_sync: event is null");
    AMT_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return AMT_OK;
}
extern "C" int amt_event_destroy(amt_ctx* ctx, void* event) {
    AMT_TRY(amt_set_device(ctx));
    if (event) AMT_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return AMT_OK;
}

extern "C" int amt_host_alloc(size_t bytes, void** hptr) {
    AMT_REQUIRE(hptr != nullptr, "amt_host_alloc: hptr is null");
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        amt_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return AMT_ENOMEM;
    }
    return AMT_OK;
}
extern "C" int amt_host_free(void* hptr) {
    if (hptr) AMT_HIP_CHECK(hipHostFree(hptr));
    return AMT_OK;
}

// Host memcpy with streaming stores: a staging block is written once and next read by the DMA engine, so pulling its
// lines into the cache first (what an ordinary store does) only doubles the destination's traffic.  Plain host code.
extern "C" int amt_host_copy(void* dst, const void* src, size_t bytes) {
    AMT_REQUIRE((dst && src) || bytes == 0, "amt_host_copy: null pointer");
    typedef long long v2di __attribute__((vector_size(16)));
    unsigned char* d = (unsigned char*)dst;
    const unsigned char* s = (const unsigned char*)src;
    size_t head = ((uintptr_t)d & 15) ? 16 - ((uintptr_t)d & 15) : 0;
    if (head > bytes) head = bytes;
    memcpy(d, s, head);
    d += head, s += head, bytes -= head;
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        v2di a, b, c, e;
        memcpy(&a, s + i, 16);
        memcpy(&b, s + i + 16, 16);
        memcpy(&c, s + i + 32, 16);
        memcpy(&e, s + i + 48, 16);
        __builtin_nontemporal_store(a, (v2di*)(d + i));
        __builtin_nontemporal_store(b, (v2di*)(d + i + 16));
        __builtin_nontemporal_store(c, (v2di*)(d + i + 32));
        __builtin_nontemporal_store(e, (v2di*)(d + i + 48));
    }
    __atomic_thread_fence(__ATOMIC_SEQ_CST);  // streaming stores are weakly ordered: drain them before returning
    memcpy(d + i, s + i, bytes - i);
    return AMT_OK;
}

// Extrema of an integer host array in ONE foreign call (the Python layer's worker threads otherwise make dozens of
// short numpy calls per label image and queue for the interpreter lock between them).  itemsize 1 / 2 / 4 / 8.
template <typename T>
static void host_minmax(const T* p, size_t n, int64_t out[2]) {
    T lo[4] = {p[0], p[0], p[0], p[0]}, hi[4] = {p[0], p[0], p[0], p[0]};
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int k = 0; k < 4; ++k) {
            const T v = p[i + k];
            lo[k] = v < lo[k] ? v : lo[k];
            hi[k] = v > hi[k] ? v : hi[k];
        }
    for (; i < n; ++i) {
        lo[0] = p[i] < lo[0] ? p[i] : lo[0];
        hi[0] = p[i] > hi[0] ? p[i] : hi[0];
    }
    for (int k = 1; k < 4; ++k) {
        lo[0] = lo[k] < lo[0] ? lo[k] : lo[0];
        hi[0] = hi[k] > hi[0] ? hi[k] : hi[0];
    }
    out[0] = (int64_t)lo[0];
    out[1] = (int64_t)hi[0];
}

extern "C" int amt_host_minmax_int(const void* src, int itemsize, int is_signed, size_t n, int64_t* out) {
    AMT_REQUIRE(src && out && n > 0, "amt_host_minmax_int: empty input");
    switch (itemsize * 2 + (is_signed ? 1 : 0)) {
        case 2: host_minmax((const uint8_t*)src, n, out); break;
        case 3: host_minmax((const int8_t*)src, n, out); break;
        case 4: host_minmax((const uint16_t*)src, n, out); break;
        case 5: host_minmax((const int16_t*)src, n, out); break;
        case 8: host_minmax((const uint32_t*)src, n, out); break;
        case 9: host_minmax((const int32_t*)src, n, out); break;
        case 17: host_minmax((const int64_t*)src, n, out); break;
        default: AMT_REQUIRE(false, "amt_host_minmax_int: unsupported element type (itemsize %d, signed %d)", itemsize, is_signed);
    }
    return AMT_OK;
}

// int64 -> int32 (label images travel as int32) with streaming stores, and the extrema of the SOURCE values in the
// same pass when `minmax` is given.  Values are truncated like a C cast; callers check the range from the extrema.
extern "C" int amt_host_narrow_i64_i32(int32_t* dst, const int64_t* src, size_t n, int64_t* minmax) {
    AMT_REQUIRE((dst && src) || n == 0, "amt_host_narrow_i64_i32: null pointer");
    int64_t lo = n ? src[0] : 0, hi = lo;
    size_t i = 0;
    typedef int v4si __attribute__((vector_size(16)));
    if (((uintptr_t)dst & 15) == 0) {
        for (; i + 4 <= n; i += 4) {
            const int64_t a = src[i], b = src[i + 1], c = src[i + 2], d = src[i + 3];
            if (minmax) {
                const int64_t l1 = a < b ? a : b, l2 = c < d ? c : d, h1 = a > b ? a : b, h2 = c > d ? c : d;
                const int64_t l = l1 < l2 ? l1 : l2, h = h1 > h2 ? h1 : h2;
                lo = l < lo ? l : lo;
                hi = h > hi ? h : hi;
            }
            const v4si v = {(int)a, (int)b, (int)c, (int)d};
            __builtin_nontemporal_store(v, (v4si*)(dst + i));
        }
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
    }
    for (; i < n; ++i) {
        const int64_t a = src[i];
        if (minmax) {
            lo = a < lo ? a : lo;
            hi = a > hi ? a : hi;
        }
        dst[i] = (int32_t)a;
    }
    if (minmax) {
        minmax[0] = lo;
        minmax[1] = hi;
    }
    return AMT_OK;
}

struct amt_timer {
    hipEvent_t start, stop;
};

extern "C" int amt_timer_create(amt_ctx* ctx, void** timer) {
    AMT_TRY(amt_set_device(ctx));
    AMT_REQUIRE(timer != nullptr, "amt_timer_create: null");
    amt_timer* t = new amt_timer();
    AMT_HIP_CHECK(hipEventCreate(&t->start));
    AMT_HIP_CHECK(hipEventCreate(&t->stop));
    *timer = t;
    return AMT_OK;
}
extern "C" int amt_timer_start(amt_ctx* ctx, void* timer) {
    AMT_TRY(amt_set_device(ctx));
    AMT_HIP_CHECK(hipEventRecord(((amt_timer*)timer)->start, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_timer_stop(amt_ctx* ctx, void* timer) {
    AMT_TRY(amt_set_device(ctx));
    AMT_HIP_CHECK(hipEventRecord(((amt_timer*)timer)->stop, ctx->stream));
    return AMT_OK;
}
extern "C" int amt_timer_elapsed_ms(amt_ctx* ctx, void* timer, float* ms) {
    AMT_TRY(amt_set_device(ctx));
    amt_timer* t = (amt_timer*)timer;
    AMT_HIP_CHECK(hipEventSynchronize(t->stop));
    AMT_HIP_CHECK(hipEventElapsedTime(ms, t->start, t->stop));
    return AMT_OK;
}
extern "C" int amt_timer_destroy(amt_ctx* ctx, void* timer) {
    AMT_TRY(amt_set_device(ctx));
    amt_timer* t = (amt_timer*)timer;
    if (t) {
        (void)hipEventDestroy(t->start);
        (void)hipEventDestroy(t->stop);
        delete t;
    }
    return AMT_OK;
}
