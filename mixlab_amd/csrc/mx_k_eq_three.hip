// mx_k_eq_three.hip -- EqThree: exact sequential kernel and the time-parallel chunked scan.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written,
// never fused; parity with it is bit-exact only if v_fma_f64 is not substituted for mul+add.
//
// Layout: every port buffer is a flat f32 stream of `frames` mono samples (or 2*frames interleaved
// L,R) -- n_ticks consecutive 735/800-sample tick buffers back to back -- 256-byte aligned.
// Instances of one module kind are batched into one launch.
#include "mx_dev.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// EqThree, exact order (src/module/eq_three.rs:58-89,117-124): one lane per instance walks its
// stream sequentially; bit-exact against the reference's golden pair.  f64-VALU/latency bound.
// ---------------------------------------------------------------------------------------------
#define MX_VSA (1.0 / 4294967295.0)   /* eq_three.rs:11 */

// Fused epilogue (see EqDesc): what StereoPanner (stereo_panner.rs:35-38) and Amplifier
// (amplifier.rs:52-57,71-73) would do to the f32 sample y the EQ just produced.
__device__ __forceinline__ void eq_emit(const EqDesc& d, size_t i, float y) {
    if (d.epi == 0u) { d.out[i] = y; return; }
    float v = y;
    if (d.epi == 2u) {
        const double m = d.ctl ? (double)d.ctl[i] : 1.0;          // amplifier.rs:54 (mono control, one value per frame)
        const double depth = d.amp_one_minus + d.amp_mod_depth * m;   // amplifier.rs:71-73
        v = (float)((double)y * depth * d.amp_amplitude);           // amplifier.rs:56
    }
    reinterpret_cast<float2*>(d.out)[i] = make_float2(v, v);
}

__device__ __forceinline__ double pump(const double f, double (&p)[4], const double sample) {
    p[0] += f * (sample - p[0]) + MX_VSA;
    p[1] += f * (p[0] - p[1]);
    p[2] += f * (p[1] - p[2]);
    p[3] += f * (p[2] - p[3]);
    return p[3];
}

__global__ __launch_bounds__(64) void k_eq_three_exact(const EqDesc* __restrict__ descs, EqState* __restrict__ states,
                                                        uint32_t n_inst, size_t frames, double lo_f, double hi_f) {
    const uint32_t inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= n_inst) return;
    const EqDesc d = descs[inst];
    EqState st = states[inst];
    double lo[4] = {st.lo[0], st.lo[1], st.lo[2], st.lo[3]};
    double hi[4] = {st.hi[0], st.hi[1], st.hi[2], st.hi[3]};
    double h0 = st.history[0], h1 = st.history[1], h2 = st.history[2];
    for (size_t i = 0; i < frames; ++i) {
        const double sample = d.in ? (double)d.in[i] : 0.0;
        const double l = pump(lo_f, lo, sample);
        const double h = h0 - pump(hi_f, hi, sample);
        const double mid = h0 - (h + l);
        h0 = h1; h1 = h2; h2 = sample;
        eq_emit(d, i, (float)(l * d.gain_lo + mid * d.gain_mid + h * d.gain_hi));
    }
    for (int k = 0; k < 4; ++k) { st.lo[k] = lo[k]; st.hi[k] = hi[k]; }
    st.history[0] = h0; st.history[1] = h1; st.history[2] = h2;
    states[inst] = st;
}
void launch_eq_three_exact(const EqDesc* d, EqState* st, uint32_t n, size_t frames, double lo_f, double hi_f, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_eq_three_exact, dim3((n + 63) / 64), dim3(64), 0, s, d, st, n, frames, lo_f, hi_f);
}

// ---------------------------------------------------------------------------------------------
// EqThree, time-parallel (default).  The two 4-pole cascades are affine recurrences
//   s[n+1] = A s[n] + b x[n] + c ,  A = lower-triangular Toeplitz with first column f^k (1-f)
// so a stream can be cut into chunks that are processed concurrently:
//   one 256-thread workgroup per instance walks its stream in segments of 256 chunks x L samples,
//   staged through LDS with coalesced loads (lane stride L+1 words => conflict-free ds_read_b32);
//   phase A: every lane runs the EXACT recurrence over its chunk from a zero state  -> z_j
//   scan:    S_j = A^(L j) S_seg + sum_{i<j} A^(L (j-1-i)) z_i   (Hillis-Steele over the wave with
//            f64 shuffles and host-precomputed Toeplitz powers, then a 4-entry hop across waves)
//   phase C: every lane re-runs the EXACT recurrence from its true initial state and emits samples.
// Only the chunk-initial states differ from the sequential order, by ~1e-16 relative; the f32
// outputs stay within 1 ULP of the reference order (measured: tests/test_gpu_audio_parity.py).
// Work is ~1.7x the sequential op count but spread over 256 lanes per instance.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pump_state(const double f, double (&p)[4], const double sample) {
    p[0] += f * (sample - p[0]) + MX_VSA;
    p[1] += f * (p[0] - p[1]);
    p[2] += f * (p[1] - p[2]);
    p[3] += f * (p[2] - p[3]);
}
// y = T(c) v for a lower-triangular Toeplitz matrix with first column c
__device__ __forceinline__ void toep_apply(const double (&c)[4], const double (&v)[4], double (&y)[4]) {
    y[0] = c[0] * v[0];
    y[1] = c[0] * v[1] + c[1] * v[0];
    y[2] = c[0] * v[2] + c[1] * v[1] + c[2] * v[0];
    y[3] = c[0] * v[3] + c[1] * v[2] + c[2] * v[1] + c[3] * v[0];
}

template <int LOG2L>
__global__ __launch_bounds__(256) void k_eq_three_scan(const EqDesc* __restrict__ descs, EqState* __restrict__ states,
                                                        size_t frames, double lo_f, double hi_f,
                                                        const EqScanTab* __restrict__ tab) {
    constexpr int L = 1 << LOG2L;
    constexpr int SEG = 256 * L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                                  // 256 * (L + 1) floats
    double* wtot = reinterpret_cast<double*>(smem + 256 * (L + 1) * sizeof(float)); // [4 waves][8]
    double* carry = wtot + 32;                                                      // [11]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const EqDesc d = descs[blockIdx.x];
    double s_lo[4], s_hi[4], hist[3];
    {
        const EqState st = states[blockIdx.x];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_lo[k] = st.lo[k]; s_hi[k] = st.hi[k]; }
        hist[0] = st.history[0]; hist[1] = st.history[1]; hist[2] = st.history[2];
    }
    double pl_lo[4], pl_hi[4], p64_lo[4], p64_hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pl_lo[k] = tab->pw[0][lane][k]; pl_hi[k] = tab->pw[1][lane][k];
        p64_lo[k] = tab->pw[0][64][k]; p64_hi[k] = tab->pw[1][64][k];
    }

    for (size_t base = 0; base < frames; base += SEG) {
        const size_t rem = frames - base;
        const int nv = rem < (size_t)SEG ? (int)rem : SEG;
        // coalesced stage-in: element e of the segment -> tile[e + e / L]
#pragma unroll 4
        for (int k = 0; k < L; ++k) {
            const int e = tid + 256 * k;
            float v = 0.f;
            if (e < nv && d.in) v = d.in[base + e];
            tile[e + (e >> LOG2L)] = v;
        }
        __syncthreads();

        const int start = tid << LOG2L;
        const int my_n = nv - start >= L ? L : (nv - start > 0 ? nv - start : 0);
        const float* mine = tile + start + tid;   // (start + i) + (start + i) / L == start + tid + i

        // phase A: zero-state response of a full chunk
        double zl[4] = {0.0, 0.0, 0.0, 0.0}, zh[4] = {0.0, 0.0, 0.0, 0.0};
        if (my_n == L) {
#pragma unroll 4
            for (int i = 0; i < L; ++i) {
                const double x = (double)mine[i];
                pump_state(lo_f, zl, x);
                pump_state(hi_f, zh, x);
            }
        }
        // the three samples before my chunk (the EQ's 3-sample delay line), read before anyone overwrites the tile
        double h0, h1, h2;
        if (tid == 0) { h0 = hist[0]; h1 = hist[1]; h2 = hist[2]; }
        else { const float* prev = tile + (start - L) + (tid - 1); h0 = (double)prev[L - 3]; h1 = (double)prev[L - 2]; h2 = (double)prev[L - 1]; }

        // inclusive scan over the wave: E_j = sum_{i<=j} P^(j-i) z_i
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int dd = 1 << k;
            double ul[4], uh[4], tl[4], th[4], cl[4], ch[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { ul[q] = __shfl_up(zl[q], dd); uh[q] = __shfl_up(zh[q], dd); cl[q] = tab->p2[0][k][q]; ch[q] = tab->p2[1][k][q]; }
            toep_apply(cl, ul, tl);
            toep_apply(ch, uh, th);
            if (lane >= dd) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { zl[q] += tl[q]; zh[q] += th[q]; }
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { wtot[wave * 8 + q] = zl[q]; wtot[wave * 8 + 4 + q] = zh[q]; }
        }
        __syncthreads();
        // state entering my wave: C_w = P^64 C_{w-1} + W_{w-1}, C_0 = segment-in state
        double cl_[4] = {s_lo[0], s_lo[1], s_lo[2], s_lo[3]}, ch_[4] = {s_hi[0], s_hi[1], s_hi[2], s_hi[3]};
        for (int w = 0; w < wave; ++w) {
            double tl[4], th[4];
            toep_apply(p64_lo, cl_, tl);
            toep_apply(p64_hi, ch_, th);
#pragma unroll
            for (int q = 0; q < 4; ++q) { cl_[q] = tl[q] + wtot[w * 8 + q]; ch_[q] = th[q] + wtot[w * 8 + 4 + q]; }
        }
        // my chunk's true initial state: S = P^lane C_w + E_{lane-1}
        double lo[4], hi[4];
        {
            double el[4], eh[4], tl[4], th[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { el[q] = __shfl_up(zl[q], 1); eh[q] = __shfl_up(zh[q], 1); }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { el[q] = 0.0; eh[q] = 0.0; }
            }
            toep_apply(pl_lo, cl_, tl);
            toep_apply(pl_hi, ch_, th);
#pragma unroll
            for (int q = 0; q < 4; ++q) { lo[q] = tl[q] + el[q]; hi[q] = th[q] + eh[q]; }
        }
        // phase C: exact recurrence from the true state, outputs overwrite my chunk of the tile
        float* mine_w = tile + start + tid;
#pragma unroll 4
        for (int i = 0; i < L; ++i) {
            if (i < my_n) {
                const double sample = (double)mine_w[i];
                const double l = pump(lo_f, lo, sample);
                const double h = h0 - pump(hi_f, hi, sample);
                const double mid = h0 - (h + l);
                h0 = h1; h1 = h2; h2 = sample;
                mine_w[i] = (float)(l * d.gain_lo + mid * d.gain_mid + h * d.gain_hi);
            }
        }
        // the chunk holding the segment's last valid sample publishes the carried state
        const int jl = (nv - 1) >> LOG2L;
        if (tid == jl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { carry[q] = lo[q]; carry[4 + q] = hi[q]; }
            carry[8] = h0; carry[9] = h1; carry[10] = h2;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { s_lo[q] = carry[q]; s_hi[q] = carry[4 + q]; }
        hist[0] = carry[8]; hist[1] = carry[9]; hist[2] = carry[10];
        // coalesced stage-out
#pragma unroll 4
        for (int k = 0; k < L; ++k) {
            const int e = tid + 256 * k;
            if (e < nv) eq_emit(d, base + e, tile[e + (e >> LOG2L)]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        EqState st;
#pragma unroll
        for (int k = 0; k < 4; ++k) { st.lo[k] = s_lo[k]; st.hi[k] = s_hi[k]; }
        st.history[0] = hist[0]; st.history[1] = hist[1]; st.history[2] = hist[2]; st.pad = 0.0;
        states[blockIdx.x] = st;
    }
}

int eq_scan_log2l(size_t frames) {
    if (frames <= 256 * 4) return 2;
    if (frames <= 256 * 8) return 3;
    if (frames <= 256 * 16) return 4;
    return 5;
}

void launch_eq_three_scan(const EqDesc* d, EqState* st, uint32_t n, size_t frames, double lo_f, double hi_f,
                          const EqScanTab* tabs /* indexed by log2L - 2 */, hipStream_t s) {
    if (!n || !frames) return;
    const int l2 = eq_scan_log2l(frames);
    const size_t lds = (size_t)256 * ((1u << l2) + 1) * sizeof(float) + (32 + 12) * sizeof(double);
    const EqScanTab* tab = tabs + (l2 - 2);
    switch (l2) {
    case 2: hipLaunchKernelGGL(k_eq_three_scan<2>, dim3(n), dim3(256), lds, s, d, st, frames, lo_f, hi_f, tab); break;
    case 3: hipLaunchKernelGGL(k_eq_three_scan<3>, dim3(n), dim3(256), lds, s, d, st, frames, lo_f, hi_f, tab); break;
    case 4: hipLaunchKernelGGL(k_eq_three_scan<4>, dim3(n), dim3(256), lds, s, d, st, frames, lo_f, hi_f, tab); break;
    default: hipLaunchKernelGGL(k_eq_three_scan<5>, dim3(n), dim3(256), lds, s, d, st, frames, lo_f, hi_f, tab); break;
    }
}

}  // namespace mx
