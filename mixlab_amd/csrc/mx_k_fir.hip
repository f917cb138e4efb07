// mx_k_fir.hip -- BUILD-SPECIFIED modules with no reference counterpart (BASELINE.json configs[2];
// the reference only has a `TODO implement resampling`, src/icecast/mod.rs:94-97):
//   Fir       128-tap (any length) FIR "reverb" on an interleaved stereo stream
//   Resample  rational polyphase resampler (44.1 -> 48 kHz is up 160 / down 147)
// Arithmetic follows the reference's own convention for audio (mixer.rs:62, eq_three.rs:85,
// amplifier.rs:56): widen f32 to f64, accumulate in f64 in ASCENDING tap index with separate
// multiply and add (no FMA), round once to f32.  Bit-exact against the oracle; parity unpinned.
//
// Every output sample is independent given the input history, so both kernels are plain
// data-parallel: a 256-lane block stages its input window in LDS (coalesced), taps are wave-uniform.
#include "mx_dev.hpp"

namespace mx {

#define FIR_BLOCK 256
// out[n] = (f32) sum_k h[k] * (f64) x[n-k] per channel; x[m<0] comes from the carried history
__global__ __launch_bounds__(FIR_BLOCK) void k_fir(const FirDesc* __restrict__ descs, size_t frames) {
    const FirDesc d = descs[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* win = reinterpret_cast<float2*>(smem);           // [FIR_BLOCK + n_taps - 1] stereo frames
    const int K = (int)d.n_taps;
    for (size_t blk = (size_t)blockIdx.x * FIR_BLOCK; blk < frames; blk += (size_t)gridDim.x * FIR_BLOCK) {
        // window frames blk-(K-1) .. blk+FIR_BLOCK-1
        for (int w = threadIdx.x; w < FIR_BLOCK + K - 1; w += FIR_BLOCK) {
            const long long f = (long long)blk + w - (K - 1);
            float2 v = make_float2(0.f, 0.f);
            if (f >= 0) { if ((size_t)f < frames && d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long h = (long long)(K - 1) + f; if (h >= 0) v = d.hist[h]; }   // hist[j] = x[j - (K-1)]
            win[w] = v;
        }
        __syncthreads();
        const size_t n = blk + threadIdx.x;
        if (n < frames) {
            double al = 0.0, ar = 0.0;
            const float2* x = win + threadIdx.x + (K - 1);     // x[0] = current frame
            for (int k = 0; k < K; ++k) {
                const double h = d.taps[k];
                const float2 v = x[-k];
                al = al + h * (double)v.x;
                ar = ar + h * (double)v.y;
            }
            reinterpret_cast<float2*>(d.out)[n] = make_float2((float)al, (float)ar);
        }
        __syncthreads();
    }
}
// new history = the last K-1 input frames of (old history ++ input)
__global__ __launch_bounds__(256) void k_fir_history(const FirDesc* __restrict__ descs, size_t frames) {
    const FirDesc d = descs[blockIdx.x];
    const int H = (int)d.n_taps - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tmp = reinterpret_cast<float2*>(smem);
    for (int j = threadIdx.x; j < H; j += 256) {
        const long long f = (long long)frames - H + j;           // index into the input stream
        float2 v = make_float2(0.f, 0.f);
        if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
        else { const long long h = (long long)H + f; if (h >= 0) v = d.hist[h]; }
        tmp[j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += 256) d.hist[j] = tmp[j];
}
void launch_fir(const FirDesc* d, uint32_t n, uint32_t max_taps, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    const size_t lds = (FIR_BLOCK + max_taps) * sizeof(float2);
    dim3 grid(grid_x(frames, FIR_BLOCK, 1024), n);
    hipLaunchKernelGGL(k_fir, grid, dim3(FIR_BLOCK), lds, s, d, frames);
    hipLaunchKernelGGL(k_fir_history, dim3(n), dim3(256), max_taps * sizeof(float2), s, d, frames);
}

// Rational resampler.  Output sample m (absolute index M = out_base + m):
//   n = floor(M * down / up), phase = (M * down) mod up
//   y[m] = (f32) sum_{k < P} h[phase][k] * (f64) x[n - k]      (x indexed absolutely; x before the run from history)
__global__ __launch_bounds__(256) void k_resample(const ResampleDesc* __restrict__ descs, size_t out_frames,
                                                  uint64_t out_base, uint64_t in_base) {
    const ResampleDesc d = descs[blockIdx.y];
    const int P = (int)d.taps_per_phase, H = P - 1;
    for (size_t m = (size_t)blockIdx.x * 256 + threadIdx.x; m < out_frames; m += (size_t)gridDim.x * 256) {
        const uint64_t M = out_base + m;
        const uint64_t num = M * d.down;
        const uint64_t n_abs = num / d.up;
        const uint32_t phase = (uint32_t)(num - n_abs * d.up);
        const long long n = (long long)(n_abs - in_base);            // index into this run's input
        const double* __restrict__ h = d.taps + (size_t)phase * P;
        double al = 0.0, ar = 0.0;
        for (int k = 0; k < P; ++k) {
            const long long f = n - k;
            float2 v = make_float2(0.f, 0.f);
            if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long hh = (long long)H + f; if (hh >= 0) v = d.hist[hh]; }
            al = al + h[k] * (double)v.x;
            ar = ar + h[k] * (double)v.y;
        }
        reinterpret_cast<float2*>(d.out)[m] = make_float2((float)al, (float)ar);
    }
}
__global__ __launch_bounds__(256) void k_resample_history(const ResampleDesc* __restrict__ descs, size_t in_frames) {
    const ResampleDesc d = descs[blockIdx.x];
    const int H = (int)d.taps_per_phase - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tmp = reinterpret_cast<float2*>(smem);
    for (int j = threadIdx.x; j < H; j += 256) {
        const long long f = (long long)in_frames - H + j;
        float2 v = make_float2(0.f, 0.f);
        if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
        else { const long long h = (long long)H + f; if (h >= 0) v = d.hist[h]; }
        tmp[j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += 256) d.hist[j] = tmp[j];
}
void launch_resample(const ResampleDesc* d, uint32_t n, uint32_t max_taps, size_t in_frames, size_t out_frames,
                     uint64_t in_base, uint64_t out_base, hipStream_t s) {
    if (!n || !out_frames) return;
    dim3 grid(grid_x(out_frames, 256, 1024), n);
    hipLaunchKernelGGL(k_resample, grid, dim3(256), 0, s, d, out_frames, out_base, in_base);
    hipLaunchKernelGGL(k_resample_history, dim3(n), dim3(256), (max_taps + 1) * sizeof(float2), s, d, in_frames);
}

}  // namespace mx
