// FETCH_SIZE calibration on the speculative EqThree kernel's own read pattern (MI355X_MICROARCH.md: "calibrate on a
// known byte count in your own access pattern before trusting an absolute").
//
// k_eq_three_spec_tiled reads its source as 64 chunk ROWS per wave, C samples apart, a SUPER-BLOCK of SB samples per row per step through
// global_load_lds_dwordx4: SB = 16 moves 64-byte HALF lines (4 lanes x 16 B per row), and the other half of every 128-byte line is asked
// for one super-block of compute later.  This probe moves a KNOWN byte count in exactly that shape -- same rows, same DMA instruction, same
// one-wave workgroups, 4 waves per SIMD, a spin of dependent f64 work per super-block standing in for the recurrence -- so that, under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_probe
// FETCH_SIZE per dispatch / bytes per dispatch is the counter's factor for: a wide coalesced read (k_wide: the documented 1/2), half lines
// whose partner follows at once (k_rows<16>, spin 0), half lines whose partner follows late (k_rows<16>, spin > 0: does the line survive in
// the 4 MiB L2 of an XCD that 512 such waves stream through?), and whole lines (k_rows<32>).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe tools/fetch_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef const float __attribute__((address_space(1)))* gfp1;
typedef float __attribute__((address_space(3)))* lfp3;
typedef float __attribute__((ext_vector_type(4))) f4v;

__global__ __launch_bounds__(256) void k_wide(const f4v* __restrict__ in, float* __restrict__ sink, size_t n4) {
    f4v a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const f4v v = __builtin_nontemporal_load(in + i); a += v; }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[0] = a.x;
}

template <int SB>
__global__ __launch_bounds__(64, 4) void k_rows(const float* __restrict__ in, float* __restrict__ sink, uint32_t C, int spin) {
    extern __shared__ __attribute__((aligned(16))) float tiles[];   // [2][64 * SB]
    constexpr int S = SB / 4, ROWS = 64 / S, TILE = 64 * SB;
    const int lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    size_t base[S];
#pragma unroll
    for (int k = 0; k < S; ++k) base[k] = (row0 + (size_t)(ROWS * k + lane / S)) * C + (size_t)(4 * (lane % S));
    auto issue = [&](float* buf, uint32_t so) {
#pragma unroll
        for (int k = 0; k < S; ++k) __builtin_amdgcn_global_load_lds((gfp1)(in + base[k] + so), (lfp3)(buf + k * 256), 16, 0, 0);
    };
    const int n_sb = (int)(C / SB);
    double acc = 1.0 + lane * 1e-3, mul = 0.999999;
    asm volatile("" : "+v"(mul));
    issue(tiles, 0);
    for (int g = 0; g < n_sb; ++g) {
        if (g + 1 < n_sb) {
            issue(tiles + ((g + 1) & 1) * TILE, (uint32_t)(g + 1) * SB);
            if (S == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f4v* row = reinterpret_cast<const f4v*>(tiles + (g & 1) * TILE + lane * SB);
#pragma unroll
        for (int p = 0; p < S; ++p) { const f4v v = row[p]; acc += (double)(v.x + v.y + v.z + v.w); }
        for (int i = 0; i < spin; ++i) acc = __builtin_fma(acc, mul, 1e-9);     // the recurrence's stand-in: a dependent f64 chain
    }
    if (acc == 12345.678) sink[blockIdx.x] = (float)acc;
}

int main(int argc, char** argv) {
    const uint32_t strips = 1024, chunks = 256, C = 6400;          // bench.py's plan: 1024 strips x 256 chunks of 6400 samples (T = 2048 ticks @ 48 kHz)
    const size_t n = (size_t)strips * chunks * C;                   // 1.68 G samples = 6.7 GB
    const int spin = argc > 1 ? atoi(argv[1]) : 250;                // ~16 samples x 64 instructions of f64 per super-block
    float* in; float* sink;
    if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 0, n * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto timed = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-28s %8.3f ms  %6.2f TB/s  (%.0f bytes per dispatch)\n", name, ms, (double)n * 4 / ms / 1e9, (double)n * 4);
    };
    const unsigned waves = strips * chunks / 64;
    timed("k_wide", [&] { hipLaunchKernelGGL(k_wide, dim3(8192), dim3(256), 0, 0, (const f4v*)in, sink, n / 4); });
    timed("k_rows<16> spin 0", [&] { hipLaunchKernelGGL(k_rows<16>, dim3(waves), dim3(64), 2 * 64 * 16 * 4, 0, in, sink, C, 0); });
    timed("k_rows<16> spin N", [&] { hipLaunchKernelGGL(k_rows<16>, dim3(waves), dim3(64), 2 * 64 * 16 * 4, 0, in, sink, C, spin); });
    timed("k_rows<32> spin 0", [&] { hipLaunchKernelGGL(k_rows<32>, dim3(waves), dim3(64), 2 * 64 * 32 * 4, 0, in, sink, C, 0); });
    timed("k_rows<32> spin 2N", [&] { hipLaunchKernelGGL(k_rows<32>, dim3(waves), dim3(64), 2 * 64 * 32 * 4, 0, in, sink, C, 2 * spin); });
    return 0;
}
