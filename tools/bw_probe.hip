// What the memory system gives a bare streaming kernel with a given mix of read and write streams (the chain phase is measured against this).
// hipcc --offload-arch=gfx950 -O3 -o bw_probe tools/bw_probe.hip && ./bw_probe   (the read-only rows are optimised away: look at the mixed ones)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// read NS streams (one 16-byte load each), write NW streams: pure traffic, like the chain's mix (7 layers in, 2.67x one layer out)
template <int NS, int NW>
__global__ __launch_bounds__(256) void k(const u4* const* in, u4* const* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u4 a = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s) { const u4 v = in[s][i]; a.x ^= v.x; a.y += v.y; a.z ^= v.z; a.w += v.w; }
#pragma unroll
        for (int w = 0; w < NW; ++w) { u4 o = a; o.x += w; out[w][i] = o; }
    }
}
template <int NS, int NW> void run(const char* name) {
    const size_t n = (size_t)16 << 20;   // 16 M x 16 B = 256 MB per stream
    std::vector<u4*> hin(NS), hout(NW);
    for (auto& p : hin) { hipMalloc(&p, n * 16); hipMemset(p, 1, n * 16); }
    for (auto& p : hout) hipMalloc(&p, n * 16);
    u4** din; u4** dout; hipMalloc(&din, 64); hipMalloc(&dout, 64);
    hipMemcpy(din, hin.data(), NS * 8, hipMemcpyHostToDevice); if (NW) hipMemcpy(dout, hout.data(), NW * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 65536}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k<NS, NW>), dim3(grid), dim3(256), 0, 0, (const u4* const*)din, (u4* const*)dout, n);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s grid %6d: %.3f ms  %.2f TB/s\n", name, grid, best, (double)(NS + NW) * n * 16 / best / 1e9);
    }
    for (auto p : hin) hipFree(p); for (auto p : hout) hipFree(p);
}
int main() { run<1, 0>("read 1      "); run<8, 0>("read 8      "); run<1, 1>("copy 1->1   "); run<7, 3>("read 7 wr 3 "); run<3, 1>("read 3 wr 1 "); return 0; }
