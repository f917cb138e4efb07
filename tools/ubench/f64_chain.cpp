// Microbenchmark (tuning aid): throughput of dependent f64 add/mul chains on gfx950 as a function of the number of independent
// chains per lane (ILP) and of waves per SIMD.  Prints wave-instructions per SIMD-cycle (peak for f64 = 0.25).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/f64_chain.cpp -o /tmp/f64_chain && /tmp/f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k_chain(double* out, double f, int iters, long long* cyc) {
    double p[ILP];
    for (int i = 0; i < ILP; ++i) p[i] = 0.1 * (i + 1) + threadIdx.x * 1e-3;
    double x = 0.5;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) { double d = x - p[i]; d = f * d; p[i] = p[i] + d; }   // sub, mul, add: 3 dependent f64 ops per chain step
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < ILP; ++i) s += p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int ILP> void run(int waves_per_simd) {
    const int n_cu = 256, iters = 20000;
    const int blocks = n_cu * 4 * waves_per_simd;   // 64-thread blocks: one wave each
    double* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_chain<ILP><<<blocks, 64>>>(out, 0.05, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_chain<ILP><<<blocks, 64>>>(out, 0.05, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = (double)iters * 8 * ILP * 3;
    const double per_simd = instr_per_wave * waves_per_simd;
    printf("ILP %d waves/SIMD %d: %.3f ms, wave0 %lld cycles (%.2f GHz), %.3f wave-instr per SIMD-cycle (f64 peak 0.25), %.2f cycles per instr per wave\n",
           ILP, waves_per_simd, ms, c, c / (ms * 1e6), per_simd / (double)c, (double)c / instr_per_wave);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 3, 4, 6, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
