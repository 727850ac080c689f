// fp64 VALU issue probe for gfx950: dependent vs independent v_add_f64 / v_mul_f64 chains at 1, 2, 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/fp64_probe.hip -o gpurun_out/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ void __launch_bounds__(256) chains(double* out, int iters, double w) {
    double a[NCH], b[NCH];
    for (int k = 0; k < NCH; ++k) { a[k] = threadIdx.x + k; b[k] = 1.0 + k; }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) { double t = a[k] + b[k]; t = t * w; b[k] = b[k] + t; }  // add -> mul -> add, dependent
    }
    long long t1 = clock64();
    double s = 0;
    for (int k = 0; k < NCH; ++k) s += b[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int NCH>
static void run(int waves_per_simd, const char* name) {
    double* d; hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    dim3 block(256), grid(256 * waves_per_simd);  // 256 CUs x 4 SIMDs: one block = one wave per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chains<NCH><<<grid, block>>>(d, 10, 0.999);
    hipEventRecord(e0);
    chains<NCH><<<grid, block>>>(d, iters, 0.999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    const double ops = 3.0 * NCH * iters;  // fp64 instructions per wave
    // per SIMD: waves_per_simd waves, each `ops` instructions
    printf("%s chains=%d waves/SIMD=%d: %.3f ms, %.2f ns per wave-instr per SIMD => %.2f cycles@2.4GHz; s_memtime ticks/instr(wave0) %.2f\n", name,
           NCH, waves_per_simd, ms, ms * 1e6 / (ops * waves_per_simd), ms * 1e6 / (ops * waves_per_simd) * 2.4, cyc / ops);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) { run<1>(w, "dep"); run<2>(w, "dep"); run<4>(w, "dep"); run<8>(w, "dep"); }
    return 0;
}
