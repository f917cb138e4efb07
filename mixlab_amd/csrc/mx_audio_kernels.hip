// mx_audio_kernels.hip -- hand-written gfx950 kernels for mixlab's audio modules.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written,
// never fused; parity with it is bit-exact only if v_fma_f64 is not substituted for mul+add.
//
// Layout: every port buffer is a flat f32 stream of `frames` mono samples (or 2*frames interleaved
// L,R) -- n_ticks consecutive 735/800-sample tick buffers back to back -- 256-byte aligned, so the
// streaming kernels move 16 B per lane (one float4 quad) and a wave moves 1 KiB per instruction.
// Instances of one module kind are batched into one launch: blockIdx.y = instance for streaming
// kernels, one wave per instance for the envelope scan, one lane per instance for the exact EQ.
#include "mx_kernels.hpp"

#include <cstdlib>

namespace mx {

// ---------------------------------------------------------------------------------------------
// guarded quad access: full quads are one dwordx4, the (single) partial tail quad goes scalar
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, size_t q, size_t n) {
    const size_t b = q * 4;
    if (b + 4 <= n) return reinterpret_cast<const float4*>(p)[q];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < n) r.x = p[b];
    if (b + 1 < n) r.y = p[b + 1];
    if (b + 2 < n) r.z = p[b + 2];
    return r;
}
__device__ __forceinline__ void st4(float* __restrict__ p, size_t q, size_t n, float4 v) {
    const size_t b = q * 4;
    if (b + 4 <= n) { reinterpret_cast<float4*>(p)[q] = v; return; }
    if (b < n) p[b] = v.x;
    if (b + 1 < n) p[b + 1] = v.y;
    if (b + 2 < n) p[b + 2] = v.z;
}
__device__ __forceinline__ float2 ld2(const float* __restrict__ p, size_t h, size_t n) {
    const size_t b = h * 2;
    if (b + 2 <= n) return reinterpret_cast<const float2*>(p)[h];
    float2 r = make_float2(0.f, 0.f);
    if (b < n) r.x = p[b];
    return r;
}

static inline unsigned grid_x(size_t items, unsigned block, unsigned cap) {
    size_t b = (items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// Amplifier (src/module/amplifier.rs:38-60,71-73)
//   out[i] = (in[i] as f64 * (1.0 - d + d * mod[i/2]) * amplitude) as f32
// algorithmic bytes per frame: 8 (in) + 4 (ctl) + 8 (out) = 20
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_amplifier(const AmpDesc* __restrict__ descs, size_t n /* stereo floats */) {
    const AmpDesc d = descs[blockIdx.y];
    const size_t nq = (n + 3) >> 2;
    const double md = d.mod_depth, amp = d.amplitude;
    const double one_minus = 1.0 - md;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 v = ld4(d.in, q, n);
        double m0 = 1.0, m1 = 1.0;
        if (d.ctl) {  // block-uniform
            const float2 c = ld2(d.ctl, q, n >> 1);   // stereo floats 4q..4q+3 <-> mono 2q, 2q+1
            m0 = (double)c.x; m1 = (double)c.y;
        }
        const double dep0 = one_minus + md * m0;      // depth(), amplifier.rs:71-73
        const double dep1 = one_minus + md * m1;
        float4 o;
        o.x = (float)((double)v.x * dep0 * amp);
        o.y = (float)((double)v.y * dep0 * amp);
        o.z = (float)((double)v.z * dep1 * amp);
        o.w = (float)((double)v.w * dep1 * amp);
        st4(d.out, q, n, o);
    }
}
void launch_amplifier(const AmpDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    const size_t ns = frames * 2;
    dim3 grid(grid_x((ns + 3) / 4, 256, 4096), n);
    hipLaunchKernelGGL(k_amplifier, grid, dim3(256), 0, s, d, ns);
}

// ---------------------------------------------------------------------------------------------
// Mixer (src/module/mixer.rs:46-71).  The f32 accumulation order over channels IS the result
// (order-sensitive far beyond 1 ULP), so each output element owns one strictly sequential add chain;
// parallelism comes from the sample axis, memory-level parallelism from U independent channel loads
// in flight per lane ahead of the chain.
//   master[i] += (in[ch][i] as f64 * gain[ch]) as f32 ;  if cue[ch] { cue[i] += in[ch][i] }
// algorithmic bytes per mixer per frame: 8 * (n_ch + 2)
// ---------------------------------------------------------------------------------------------
template <int W> struct VecF;
template <> struct VecF<1> { typedef float T; };
template <> struct VecF<2> { typedef float2 T; };
template <> struct VecF<4> { typedef float4 T; };

template <int W>
__device__ __forceinline__ void ldw(const float* __restrict__ p, size_t idx, size_t n, float (&v)[W]) {
    const size_t b = idx * W;
    if (b + W <= n) {
        const typename VecF<W>::T t = reinterpret_cast<const typename VecF<W>::T*>(p)[idx];
        const float* tf = reinterpret_cast<const float*>(&t);
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = tf[k];
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = (b + k < n) ? p[b + k] : 0.f;
    }
}
template <int W>
__device__ __forceinline__ void stw(float* __restrict__ p, size_t idx, size_t n, const float (&v)[W]) {
    const size_t b = idx * W;
    if (b + W <= n) {
        typename VecF<W>::T t;
        float* tf = reinterpret_cast<float*>(&t);
#pragma unroll
        for (int k = 0; k < W; ++k) tf[k] = v[k];
        reinterpret_cast<typename VecF<W>::T*>(p)[idx] = t;
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) if (b + k < n) p[b + k] = v[k];
    }
}

// W floats per lane, U channel loads in flight per lane, one wave per block so that small sample
// counts still spread over all 256 CUs (a CU's HBM pull is capped at ~10 B/clk).
template <int W, int U>
__global__ __launch_bounds__(64) void k_mixer(const MixDesc* __restrict__ descs, size_t n /* stereo floats */) {
    const MixDesc m = descs[blockIdx.y];
    const MixChan* __restrict__ ch = m.chans;
    const size_t items = (n + W - 1) / W;
    for (size_t q = (size_t)blockIdx.x * 64 + threadIdx.x; q < items; q += (size_t)gridDim.x * 64) {
        float acc[W], cac[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { acc[k] = 0.f; cac[k] = 0.f; }   // util::zero(master/cue), mixer.rs:54-55
        uint32_t c = 0;
        for (; c + U <= m.n_ch; c += U) {
            float v[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u) ldw<W>(ch[c + u].in, q, n, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double g = ch[c + u].gain;
                const bool cue = ch[c + u].cue != 0;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    acc[k] += (float)((double)v[u][k] * g);   // mixer.rs:62
                    if (cue) cac[k] += v[u][k];              // mixer.rs:64-66
                }
            }
        }
        for (; c < m.n_ch; ++c) {
            float v[W];
            ldw<W>(ch[c].in, q, n, v);
            const double g = ch[c].gain;
            const bool cue = ch[c].cue != 0;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                acc[k] += (float)((double)v[k] * g);
                if (cue) cac[k] += v[k];
            }
        }
        stw<W>(m.master, q, n, acc);
        stw<W>(m.cue, q, n, cac);
    }
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

void launch_mixer(const MixDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    const size_t ns = frames * 2;
    // widest lane vector that still yields >= ~4 waves per CU; tuning override for experiments
    static const int force_w = env_int("MX_MIXER_W", 0);
    int w = force_w;
    if (w != 1 && w != 2 && w != 4) {
        const size_t want_lanes = (size_t)64 * 1024 / (n ? n : 1);
        w = (ns / 4 >= want_lanes) ? 4 : (ns / 2 >= want_lanes ? 2 : 1);
    }
    const size_t items = (ns + w - 1) / w;
    dim3 grid(grid_x(items, 64, 16384), n);
    switch (w) {
    case 4: hipLaunchKernelGGL((k_mixer<4, 8>), grid, dim3(64), 0, s, d, ns); break;
    case 2: hipLaunchKernelGGL((k_mixer<2, 16>), grid, dim3(64), 0, s, d, ns); break;
    default: hipLaunchKernelGGL((k_mixer<1, 16>), grid, dim3(64), 0, s, d, ns); break;
    }
}

// ---------------------------------------------------------------------------------------------
// EqThree, exact order (src/module/eq_three.rs:58-89,117-124): one lane per instance walks its
// stream sequentially; bit-exact against the reference's golden pair.  f64-VALU/latency bound.
// ---------------------------------------------------------------------------------------------
#define MX_VSA (1.0 / 4294967295.0)   /* eq_three.rs:11 */

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
        d.out[i] = (float)(l * d.gain_lo + mid * d.gain_mid + h * d.gain_hi);
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
            if (e < nv) d.out[base + e] = tile[e + (e >> LOG2L)];
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

// ---------------------------------------------------------------------------------------------
// Envelope (src/module/envelope.rs:34-58,91-120).  The reference is a per-sample state machine, but
// both non-identity inputs are constant maps on {not-on, on}: gate == 1.0 forces "on", gate == 0.0
// forces "not-on".  So the state bit after sample i is the value of the last marker at or before i,
// edges are where that bit flips, and (state, since-when) follows from the last rising / falling
// edge.  One wave per instance: 64 samples per step, edges found with ballots + clz (no shuffles,
// no LDS), closed-form amplitude per lane, state carried in SGPR-uniform registers across steps.
// ---------------------------------------------------------------------------------------------
// (last - first) as f64 / SAMPLE_RATE * 1000.0 (envelope.rs:16-18) with the IEEE quotient obtained
// by Markstein's correction instead of the 20-instruction division expansion: q = a*y, r = fma(-q,b,a),
// q' = fma(r,y,q) with y = RN(1/b) from the host.  tests/test_fastdiv.py checks q' == a/b bit-for-bit
// for every a in [0, 2^32) at 44.1 and 48 kHz; larger spans (> 24 h) take the true division.
__device__ __forceinline__ double seq_ms(uint64_t first, uint64_t last, double sr, double rsr) {
    const uint64_t dt = last - first;
    if (dt >> 32) return (double)dt / sr * 1000.0;
    const double a = (double)(uint32_t)dt;
    double q = a * rsr;
    const double r = fma(-q, sr, a);
    q = fma(r, rsr, q);
    return q * 1000.0;
}
__device__ __forceinline__ double clamp01(double x) { return x > 1.0 ? 1.0 : (x < 0.0 ? 0.0 : x); }  // envelope.rs:20-28
__device__ __forceinline__ double amp_on_ms(const EnvDesc& p, double ms) {                            // envelope.rs:37-49
    const double attack = p.inv_attack * ms;
    const double since_decay = ms - p.attack_ms;
    const double decay_amplitude = 1.0 - clamp01(p.inv_decay * since_decay);
    const double decay = p.sustain + (p.one_minus_sustain * decay_amplitude);
    return ms < p.attack_ms ? attack : decay;
}
__device__ __forceinline__ double amp_off_ms(const EnvDesc& p, double off_amp, double ms) {           // envelope.rs:51-56
    const double release_amplitude = 1.0 - clamp01(p.inv_release * ms);
    return off_amp * release_amplitude;
}
__device__ __forceinline__ int top_bit(uint64_t m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint64_t read_lane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

template <int K>
__global__ __launch_bounds__(256) void k_envelope(const EnvDesc* __restrict__ descs, EnvState* __restrict__ states,
                                                   uint32_t n_inst, size_t frames, uint64_t t0, double sr, double rsr) {
    const int lane = threadIdx.x & 63;
    const uint32_t inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= n_inst) return;  // wave-uniform
    const EnvDesc p = descs[inst];
    // carried EnvelopeState: wave-uniform (kept in SGPRs via readlane)
    uint32_t tag = states[inst].tag;
    uint64_t seq = states[inst].seq;
    double off_amp = states[inst].off_amplitude;
    tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)tag);
    seq = read_lane_u64(seq, 0);
    off_amp = __longlong_as_double((long long)read_lane_u64((uint64_t)__double_as_longlong(off_amp), 0));

    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t le = lt | (1ull << lane);

    for (size_t base = 0; base < frames; base += 64 * K) {
        float xs[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {                       // K independent loads in flight
            const size_t i = base + 64 * k + lane;
            xs[k] = (i < frames && p.gate) ? p.gate[i] : 0.0f;   // Disconnected => ZERO_BUFFER_MONO
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t tbase = base + 64 * k;
            if (tbase >= frames) break;                     // wave-uniform
            const size_t i = tbase + lane;
            const bool valid = i < frames;
            const float x = xs[k];
            const uint64_t m1 = __ballot(valid && x == 1.0f);        // envelope.rs:102
            const uint64_t m0 = __ballot(valid && x == 0.0f);        // envelope.rs:107
            const uint64_t mk = m0 | m1;
            const uint64_t below = mk & lt;
            const bool carry_on = (tag == 1u);
            const bool b_prev = below ? (((m1 >> top_bit(below)) & 1ull) != 0) : carry_on;
            const bool b_cur = ((mk >> lane) & 1ull) ? (((m1 >> lane) & 1ull) != 0) : b_prev;
            const uint64_t R = __ballot(valid && !b_prev && b_cur);  // Initial|Off -> On
            const uint64_t F = __ballot(valid && b_prev && !b_cur);  // On -> Off
            const uint64_t tb = t0 + tbase;

            uint32_t my_tag = tag; uint64_t my_seq = seq; double my_off = off_amp;   // carried Initial / TriggerOff
            const uint64_t Rle = R & le, Fle = F & le;
            if (b_cur) {
                my_tag = 1u;
                if (Rle) my_seq = tb + (uint64_t)top_bit(Rle);
            } else if (Fle) {                                         // rare: a falling edge inside this tile
                const int fl = top_bit(Fle);
                const uint64_t off = tb + (uint64_t)fl;
                const uint64_t Rb = R & ((1ull << fl) - 1ull);
                const uint64_t on = Rb ? tb + (uint64_t)top_bit(Rb) : seq;
                my_tag = 2u; my_seq = off;
                my_off = amp_on_ms(p, seq_ms(on, off, sr, rsr));      // envelope.rs:108-111
            }
            const double ms = seq_ms(my_seq, tb + (uint64_t)lane, sr, rsr);
            const double a_on = amp_on_ms(p, ms);
            const double a_off = amp_off_ms(p, my_off, ms);
            const double a = my_tag == 1u ? a_on : (my_tag == 2u ? a_off : 0.0);   // envelope.rs:36
            if (valid) p.out[i] = (float)a;

            const size_t rem = frames - tbase;
            const int last = rem >= 64 ? 63 : (int)rem - 1;
            tag = (uint32_t)__builtin_amdgcn_readlane((int)my_tag, last);
            seq = read_lane_u64(my_seq, last);
            off_amp = __longlong_as_double((long long)read_lane_u64((uint64_t)__double_as_longlong(my_off), last));
        }
    }
    if (lane == 0) { states[inst].tag = tag; states[inst].seq = seq; states[inst].off_amplitude = off_amp; }
}
void launch_envelope(const EnvDesc* d, EnvState* st, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s) {
    if (!n || !frames) return;
    const double rsr = 1.0 / sample_rate;
    if (frames > 64 * 4) hipLaunchKernelGGL(k_envelope<8>, dim3((n + 3) / 4), dim3(256), 0, s, d, st, n, frames, t0, sample_rate, rsr);
    else hipLaunchKernelGGL(k_envelope<2>, dim3((n + 3) / 4), dim3(256), 0, s, d, st, n, frames, t0, sample_rate, rsr);
}

// ---------------------------------------------------------------------------------------------
// Oscillator (src/module/oscillator.rs:15-37,65-92) and FmSine (src/module/fm_sine.rs:37-56).
// f64 sin = ocml's; the reference's is the host libm.  Both are sub-ULP f64 routines; after the
// f32 cast the results differ in at most 1 f32 ULP, rarely (measured in tests).
// ---------------------------------------------------------------------------------------------
#define MX_PI 3.14159265358979323846264338327950288

__device__ __forceinline__ double osc_saw(double n) { return 2.0 * (n - floor(0.5 + n)); }

__global__ __launch_bounds__(256) void k_oscillator(const OscDesc* __restrict__ descs, size_t frames, uint64_t t0, double sr) {
    const OscDesc d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += (size_t)gridDim.x * 256) {
        const double tt = (double)(t0 + (uint64_t)i) / sr;
        const double n = tt * d.freq;
        double v;
        switch (d.waveform) {
        case 2: v = sin(n * 2.0 * MX_PI); break;                                      // Sine
        case 3: { const double sv = sin(n * 2.0 * MX_PI); v = signbit(sv) ? -1.0 : 1.0; break; }  // Square: sign by sign bit (oscillator.rs:15-23)
        case 5: v = osc_saw(n); break;                                                // Saw
        case 4: v = 2.0 * fabs(osc_saw(n)) - 1.0; break;                              // Triangle
        case 0: v = 1.0; break;                                                       // On
        default: v = 0.0; break;                                                      // Off
        }
        const float sm = (float)v;
        d.mono[i] = sm;
        reinterpret_cast<float2*>(d.stereo)[i] = make_float2(sm, sm);
    }
}
void launch_oscillator(const OscDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x(frames, 256, 4096), n);
    hipLaunchKernelGGL(k_oscillator, grid, dim3(256), 0, s, d, frames, t0, sample_rate);
}

__global__ __launch_bounds__(256) void k_fm_sine(const FmDesc* __restrict__ descs, size_t frames, uint64_t t0, double sr) {
    const FmDesc d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += (size_t)gridDim.x * 256) {
        const double tt = (double)(t0 + (uint64_t)i) / sr;
        const double xin = d.in ? (double)d.in[i] : 0.0;
        const double co = (d.freq_mid + d.freq_amp * xin) * 2.0 * MX_PI;
        const float x = (float)sin(co * tt);
        reinterpret_cast<float2*>(d.out)[i] = make_float2(x, x);
    }
}
void launch_fm_sine(const FmDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x(frames, 256, 4096), n);
    hipLaunchKernelGGL(k_fm_sine, grid, dim3(256), 0, s, d, frames, t0, sample_rate);
}

// ---------------------------------------------------------------------------------------------
// Trigger / StereoPanner / StereoSplitter (trigger.rs:35-48, stereo_panner.rs:30-41,
// stereo_splitter.rs:33-47): fills and layout shuffles, 16 B per lane on the wide side.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trigger(const TrigDesc* __restrict__ descs, size_t frames) {
    const TrigDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;
    const float4 v = make_float4(d.value, d.value, d.value, d.value);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) st4(d.out, q, frames, v);
}
void launch_trigger(const TrigDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x((frames + 3) / 4, 256, 2048), n);
    hipLaunchKernelGGL(k_trigger, grid, dim3(256), 0, s, d, frames);
}

__global__ __launch_bounds__(256) void k_panner(const PanDesc* __restrict__ descs, size_t frames) {
    const PanDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;   // quads of frames -> two stereo quads
    const size_t ns = frames * 2;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 l = d.l ? ld4(d.l, q, frames) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r = d.r ? ld4(d.r, q, frames) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(d.out, 2 * q, ns, make_float4(l.x, r.x, l.y, r.y));
        if ((2 * q + 1) * 4 < ns) st4(d.out, 2 * q + 1, ns, make_float4(l.z, r.z, l.w, r.w));
    }
}
void launch_panner(const PanDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x((frames + 3) / 4, 256, 4096), n);
    hipLaunchKernelGGL(k_panner, grid, dim3(256), 0, s, d, frames);
}

__global__ __launch_bounds__(256) void k_splitter(const SplitDesc* __restrict__ descs, size_t frames) {
    const SplitDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;
    const size_t ns = frames * 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 a = d.in ? ld4(d.in, 2 * q, ns) : z;
        const float4 b = (d.in && (2 * q + 1) * 4 < ns) ? ld4(d.in, 2 * q + 1, ns) : z;
        st4(d.l, q, frames, make_float4(a.x, a.z, b.x, b.z));
        st4(d.r, q, frames, make_float4(a.y, a.w, b.y, b.w));
    }
}
void launch_splitter(const SplitDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x((frames + 3) / 4, 256, 4096), n);
    hipLaunchKernelGGL(k_splitter, grid, dim3(256), 0, s, d, frames);
}

// Plotter (plotter.rs:37-56): de-interleave one tick per job into the indication staging area.
__global__ __launch_bounds__(256) void k_plotter(const PlotJob* __restrict__ jobs, size_t spt) {
    const PlotJob j = jobs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < spt; i += (size_t)gridDim.x * 256) {
        const float2 v = reinterpret_cast<const float2*>(j.in)[i];
        j.left[i] = v.x; j.right[i] = v.y;
    }
}
void launch_plotter(const PlotJob* d, uint32_t n, size_t spt, hipStream_t s) {
    if (!n || !spt) return;
    dim3 grid(grid_x(spt, 256, 64), n);
    hipLaunchKernelGGL(k_plotter, grid, dim3(256), 0, s, d, spt);
}

}  // namespace mx
