// Does hipExtAnyOrderLaunch let latency-bound kernels of ONE stream overlap on gfx950?
// Four "flood-like" kernels (few waves, tens of KB of LDS each, ~200 us of dependent LDS work) are launched
//   (a) back to back with ordinary launches, (b) the 2nd..4th with hipExtAnyOrderLaunch, (c) on four streams,
// followed by an ordinary kernel that checks every result (ordering of the FOLLOWING launch must hold).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) spin_kernel(unsigned* out, int iters, int words) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 64) lds[i] = i * 2654435761u;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned p = threadIdx.x;
    for (int i = 0; i < iters; ++i) p = lds[(p ^ i) % words] + 1;  // dependent LDS chain
    out[blockIdx.x * 64 + threadIdx.x] = p | 1u;
}

__global__ void check_kernel(const unsigned* a, int n, int* bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] == 0) atomicAdd(bad, 1);
}

int main() {
    const int NK = 4, WG = 512, iters = 3000;
    const size_t lds_bytes[NK] = {15360, 27648, 55296, 159744};
    unsigned* out[NK];
    int* bad;
    hipMalloc(&bad, 4);
    for (int k = 0; k < NK; ++k) hipMalloc(&out[k], WG * 64 * 4);
    hipStream_t s[NK];
    for (int k = 0; k < NK; ++k) hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking);
    hipEvent_t e0, e1, ev[NK];
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < NK; ++k) hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    for (int k = 0; k < NK; ++k)
        hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            for (int k = 0; k < NK; ++k) hipMemsetAsync(out[k], 0, WG * 64 * 4, s[0]);
            hipMemsetAsync(bad, 0, 4, s[0]);
            hipStreamSynchronize(s[0]);
            hipEventRecord(e0, s[0]);
            for (int k = 0; k < NK; ++k) {
                const int words = (int)(lds_bytes[k] / 4);
                const int wg = k == 3 ? 64 : WG;
                if (mode == 0) {
                    hipLaunchKernelGGL(spin_kernel, dim3(wg), dim3(64), lds_bytes[k], s[0], out[k], iters, words);
                } else if (mode == 1) {
                    hipExtLaunchKernelGGL(spin_kernel, dim3(wg), dim3(64), (unsigned)lds_bytes[k], s[0], nullptr, nullptr,
                                          k == 0 ? 0u : (unsigned)hipExtAnyOrderLaunch, out[k], iters, words);
                } else {
                    if (k > 0) {
                        hipEventRecord(ev[0], s[0]);
                        hipStreamWaitEvent(s[k], ev[0], 0);
                    }
                    hipLaunchKernelGGL(spin_kernel, dim3(wg), dim3(64), lds_bytes[k], s[k], out[k], iters, words);
                    if (k > 0) {
                        hipEventRecord(ev[k], s[k]);
                    }
                }
            }
            if (mode == 2)
                for (int k = 1; k < NK; ++k) hipStreamWaitEvent(s[0], ev[k], 0);
            for (int k = 0; k < NK; ++k)
                hipLaunchKernelGGL(check_kernel, dim3(WG * 64 / 256), dim3(256), 0, s[0], out[k], (k == 3 ? 64 : WG) * 64, bad);
            hipEventRecord(e1, s[0]);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            int hb;
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("mode %d (%s) rep %d: %.3f ms, unwritten results seen by the following kernel: %d, err=%s\n", mode,
                   mode == 0 ? "in order" : mode == 1 ? "any-order flag" : "four streams", rep, ms, hb,
                   hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
