// Does a launch with hipExtAnyOrderLaunch START while the previous kernel of the same stream is still running?
// A: 32 workgroups that spin for ~300 us (a few long chains, most of the GPU idle); B: 4096 short workgroups (~40 us of
// work for the whole GPU).  In order: t(A) + t(B).  Overlapped: ~t(A).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/anyorder_probe2.hip -o tools/probes/anyorder_probe2.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void __launch_bounds__(64) spin_kernel(unsigned* out, long long cycles) {
    extern __shared__ unsigned lds[];
    const long long t0 = clock64();
    unsigned p = threadIdx.x;
    while (clock64() - t0 < cycles) p = p * 1664525u + 1013904223u;
    lds[threadIdx.x] = p;
    out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x] | 1u;
}

int main() {
    unsigned *a, *b;
    hipMalloc(&a, 4096 * 64 * 4);
    hipMalloc(&b, 4096 * 64 * 4);
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, ev;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long longc = 30000, shortc = 1000;  // clock64 ticks at 100 MHz: 300 us / 10 us per workgroup
    for (int ldsA : {256, 159744}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipStreamSynchronize(s);
                hipEventRecord(e0, s);
                void* argsA[] = {(void*)&a, (void*)&longc};
                void* argsB[] = {(void*)&b, (void*)&shortc};
                hipExtLaunchKernel((const void*)spin_kernel, dim3(32), dim3(64), argsA, ldsA, s, nullptr, nullptr, 0);
                if (mode == 2) {
                    hipEventRecord(ev, s2);
                    hipExtLaunchKernel((const void*)spin_kernel, dim3(4096), dim3(64), argsB, 256, s2, nullptr, nullptr, 0);
                    hipEventRecord(ev, s2);
                    hipStreamWaitEvent(s, ev, 0);
                } else {
                    hipExtLaunchKernel((const void*)spin_kernel, dim3(4096), dim3(64), argsB, 256, s, nullptr, nullptr,
                                       mode == 1 ? hipExtAnyOrderLaunch : 0);
                }
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("A lds %6d, mode %d (%s) rep %d: %.3f ms\n", ldsA, mode,
                       mode == 0 ? "in order" : mode == 1 ? "any-order flag" : "second stream", rep, ms);
            }
        }
    }
    return 0;
}
