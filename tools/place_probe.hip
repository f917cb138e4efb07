// How does the dispatcher place the workgroups of one launch?  Every block records HW_ID / XCC_ID and spins for a while (so that all blocks are resident together);
// the host counts blocks per CU and waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/place_probe.hip -o /tmp/place_probe && /tmp/place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ void k(uint32_t* out, long spin) {
    extern __shared__ float lds[];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long t0 = __builtin_amdgcn_s_memtime();
    double a = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < spin) { a = a * 1.0000001 + 1e-9; }
    if (a == 12345.678) lds[threadIdx.x] = (float)a;
    if ((threadIdx.x & 63) == 0) { const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64; out[2 * w] = hw; out[2 * w + 1] = xcc; }
}
static void run(const char* what, dim3 grid, int threads, size_t lds, long spin) {
    const uint32_t waves = grid.x * grid.y * (threads / 64);
    uint32_t* d; hipMalloc(&d, waves * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, waves * 8);
        hipLaunchKernelGGL(k, grid, dim3(threads), lds, 0, d, spin);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(2 * waves); hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> per_simd, per_cu;
        for (uint32_t w = 0; w < waves; ++w) {
            const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 15;
            const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const uint32_t cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
            per_cu[cuid]++; per_simd[cuid * 4 + simd]++;
        }
        std::map<int, int> hist_simd, hist_cu;
        for (auto& kv : per_simd) hist_simd[kv.second]++;
        for (auto& kv : per_cu) hist_cu[kv.second]++;
        printf("%-46s rep %d: %u waves on %zu CUs / %zu SIMDs; waves per used SIMD:", what, rep, waves, per_cu.size(), per_simd.size());
        for (auto& kv : hist_simd) printf(" %d x%d", kv.first, kv.second);
        printf(" | waves per used CU:");
        for (auto& kv : hist_cu) printf(" %d x%d", kv.first, kv.second);
        printf("\n");
    }
    hipFree(d);
}
int main() {
    const long spin = 20000000;   // ~ 10 ms of s_memtime ticks at 100 MHz?  long enough for every block to be resident at once
    run("1024 blocks x 64 threads, 16 KB LDS (EqThree, old)", dim3(1024), 64, 16384, spin / 100);
    run("256 blocks x 256 threads, 64 KB LDS (EqThree, new)", dim3(256), 256, 65536, spin / 100);
    run("2048 blocks x 64 threads, 16 KB LDS", dim3(2048), 64, 16384, spin / 100);
    run("4096 blocks x 64 threads, 8 KB LDS", dim3(4096), 64, 8192, spin / 100);
    run("1280 blocks x 256 threads, 8 KB LDS (k_resample_ps)", dim3(5, 256), 256, 8192, spin / 100);
    run("1024 blocks x 256 threads, 39 KB LDS (k_fir-like)", dim3(4, 256), 256, 39936, spin / 100);
    return 0;
}
