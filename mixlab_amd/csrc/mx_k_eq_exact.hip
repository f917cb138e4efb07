// mx_k_eq_exact.hip -- EqThree in the reference's exact order (reference src/module/eq_three.rs:58-89,100-125), the default:
//
//   k_env_ticks        per-tick states of the Envelopes folded into the EQ epilogue (one wave per instance, ballots + clz)
//   k_eq_three_exact   one lane per instance walks its stream (short streams: a tick at a time)
//   k_eq_three_spec    SPECULATIVE time-parallel form for long streams, bit-exact by verification:
//   k_eq_three_repair  ... the pass that proves (or restores) exactness
//
// Why speculation can be exact.  The two 4-pole cascades are contractions: two trajectories driven by the same input from
// different states approach each other by (1 - f) per sample and, because every step rounds to the same f64 grid, they
// COALESCE bit for bit once their distance is below half an ulp -- after a warm-up of W samples (eq_warm_len: the
// cascade's k^3 p^k envelope below 2^-72) a filter started from zeros is in the very state the sequential filter is in,
// for every live signal (noise, tones, music, DC; measured, DESIGN.md "EqThree").  So the stream of an instance is cut into
// chunks of C samples, one LANE per chunk: the lane runs the exact recurrence over the W samples before its chunk from a
// zero state, records the state it reaches (start), runs its chunk in the exact order -- outputs through the fused epilogue
// -- and records its end state.  Chunk 0 starts from the carried state.  k_eq_three_repair then walks the chunks of an
// instance in order: chunk j is PROVEN exact when its recorded start state equals, bit for bit, the proven end state of
// chunk j-1 (induction from chunk 0).  Where the bits differ (it happens when the input was exactly constant for a long
// time -- digital silence after a signal: the poles stall a few ulps from the fixed point, on the side they came from) the
// pass re-runs the chunk from the proven state beside the speculative trajectory, rewrites the output samples whose f32
// differs, and stops as soon as the two trajectories coalesce.  The result is the sequential order's, always; only the time
// it takes depends on the input.
#include <algorithm>
#include <cmath>

#include "mx_k_eq_common.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// per-tick Envelope states (EnvTick) for Envelopes whose gate is a Trigger constant per tick
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int top_bit64(uint64_t m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), l));
}

// One wave per instance, 64 ticks per step: the gate of tick c is bit c, so "state after the first sample of tick c" follows
// from the last rising / falling edge at or before c exactly as in k_envelope (envelope.rs:99-115 with one marker per tick).
template <bool FC>
__global__ __launch_bounds__(256) void k_env_ticks(const EnvTickDesc* __restrict__ descs, uint32_t n_inst, GateBits gates, uint32_t n_calls,
                                                    size_t fpc, uint64_t t0, double sr, double rsr, EnvTick* __restrict__ ticks) {
    const int lane = threadIdx.x & 63;
    const uint32_t inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= n_inst) return;   // wave-uniform
    const EnvTickDesc d = descs[inst];
    if (!d.state) return;   // this instance folds no Envelope (wave-uniform)
    uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.state->tag);
    uint64_t seq = readlane_u64(d.state->seq, 0);
    double off_amp = readlane_f64(d.state->off_amplitude, 0);
    const uint64_t lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
    for (uint32_t c0 = 0; c0 < n_calls; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        const bool valid = c < n_calls;
        const bool b_cur = valid && gate_bit(gates, inst, c);
        const uint64_t m1 = __ballot(b_cur);
        const bool carry_on = tag == 1u;
        const bool b_prev = lane ? ((m1 >> (lane - 1)) & 1ull) != 0 : carry_on;
        const uint64_t R = __ballot(valid && !b_prev && b_cur);   // Initial | Off -> On  (envelope.rs:101-105)
        const uint64_t F = __ballot(valid && b_prev && !b_cur);   // On -> Off             (envelope.rs:106-113)
        auto t_of = [&](int l) { return t0 + (uint64_t)(c0 + (uint32_t)l) * (uint64_t)fpc; };
        uint32_t my_tag = tag; uint64_t my_seq = seq; double my_off = off_amp;
        const uint64_t Rle = R & le, Fle = F & le;
        if (b_cur) {
            my_tag = 1u;
            if (Rle) my_seq = t_of(top_bit64(Rle));
        } else if (Fle) {
            const int fl = top_bit64(Fle);
            const uint64_t off = t_of(fl);
            const uint64_t Rb = R & ((1ull << fl) - 1ull);
            const uint64_t on = Rb ? t_of(top_bit64(Rb)) : seq;
            my_tag = 2u; my_seq = off;
            my_off = amp_on_ms<FC>(d.p, seq_ms(on, off, sr, rsr));   // envelope.rs:108-111
        }
        if (valid) {
            const uint64_t t = t_of(lane);
            EnvTick k;
            k.seq = my_seq; k.off_amp = my_off; k.tag = my_tag;
            k.flat = env_saturated(d.p, my_tag, my_seq, t, sr, rsr) ? 1u : 0u;
            const float cc = (float)env_amplitude<FC>(d.p, my_tag, my_seq, my_off, t, sr, rsr);
            k.depth = amp_depth<FC>(d.amp_one_minus, d.amp_mod_depth, (double)cc);
            ticks[(size_t)inst * n_calls + c] = k;
        }
        const int last = n_calls - c0 >= 64u ? 63 : (int)(n_calls - c0) - 1;
        tag = (uint32_t)__builtin_amdgcn_readlane((int)my_tag, last);
        seq = readlane_u64(my_seq, last);
        off_amp = readlane_f64(my_off, last);
    }
    if (lane == 0) { d.state->tag = tag; d.state->seq = seq; d.state->off_amplitude = off_amp; }
}
void launch_env_ticks(const EnvTickDesc* d, uint32_t n, const GateBits& gates, uint32_t n_calls, size_t fpc, uint64_t t0, double sample_rate, EnvTick* ticks, hipStream_t s, bool fc) {
    if (!n || !n_calls) return;
    if (fc) hipLaunchKernelGGL(k_env_ticks<true>, dim3((n + 3) / 4), dim3(256), 0, s, d, n, gates, n_calls, fpc, t0, sample_rate, 1.0 / sample_rate, ticks);
    else hipLaunchKernelGGL(k_env_ticks<false>, dim3((n + 3) / 4), dim3(256), 0, s, d, n, gates, n_calls, fpc, t0, sample_rate, 1.0 / sample_rate, ticks);
}

// ---------------------------------------------------------------------------------------------
// speculative time-parallel exact form
// ---------------------------------------------------------------------------------------------
// what a chunk's lane leaves for the repair pass
struct EqChunkRec {
    double start[8];        // state the warm-up reached at the chunk's first sample (chunk 0: the carried state)
    double end[8];          // state after the chunk's last sample
    uint32_t xmin, xmax;    // min / max of the chunk's input bit patterns: equal => the input was constant
    uint32_t pad[2];
};

typedef float __attribute__((ext_vector_type(4))) f4v;
typedef float __attribute__((ext_vector_type(2))) f2v;
__device__ __forceinline__ f4v ld_stream4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); }   // the source is streamed once

// epilogue modes of the chunk loop (wave-uniform: one instance per wave)
enum { EQM_PLAIN = 0, EQM_AMP_CONST = 1, EQM_AMP_CTL = 2, EQM_AMP_ENV = 3 };

constexpr int EQ_BLK = 16;   // samples per block: four 16-byte loads in flight per lane while the previous block is computed

// The inline Envelope of one tick for one lane, reduced to per-lane coefficients so that the per-sample code has NO branch
// (lanes of a wave sit in different ticks and Envelope phases; branches would also fence the recurrence's instruction stream):
//   On  (envelope.rs:37-49):  ms < attack_ms ? inv_attack * ms : sustain + (1 - sustain) * (1 - clamp(inv_decay * (ms - attack_ms)))
//   Off (envelope.rs:51-56):  off_amplitude * (1 - clamp(inv_release * ms))
//   Initial (envelope.rs:36): 0.0
// All three are  A + B * (1 - min(k * (ms - m0), 1))  with (A, B, k, m0) = (sustain, 1 - sustain, inv_decay, attack_ms),
// (0, off_amplitude, inv_release, 0), (0, 0, 0, 0):
//   * Off / Initial: the extra operations are exact identities (ms - 0.0 == ms; 0.0 + p == p for every p but -0.0, which
//     off_amplitude * [0, 1] cannot be when off_amplitude >= +0.0);
//   * clamp()'s lower bound (envelope.rs:23-24) never acts where the value is used: with k >= 0, k * (ms - m0) < 0 only for an On
//     lane before attack_ms, where the attack ramp is selected instead;
//   * a FLAT lane (sustain reached, release finished, never triggered) needs no special case: saturation means min() returns 1.0
//     and the formula yields the same constant the reference computes, operation for operation.
// Preconditions (else the general form, env_depth): finite parameters with inv_decay, inv_release >= 0 (env_params_nice),
// off_amplitude >= +0.0, and a sample distance that stays in 32 bits.  ms = (dt as f64) / SR * 1000, dt = dt0 + sample index.
struct EnvLane { double A, B, k, m0, depth; uint32_t dt0; uint32_t on; uint32_t flat; uint32_t general;
                 uint32_t k0 = 0; uint64_t t_chunk = 0; /* tiled kernel: chunk-relative index of the tick's first sample; absolute time of the chunk's */ };
__device__ __forceinline__ bool env_params_nice(const EnvParams& p) {
    auto fin = [](double v) { return v == v && fabs(v) < 1.0e300; };
    return fin(p.attack_ms) && fin(p.inv_attack) && fin(p.inv_decay) && fin(p.sustain) && fin(p.one_minus_sustain) && fin(p.inv_release) &&
           p.inv_decay >= 0.0 && p.inv_release >= 0.0;
}
__device__ __forceinline__ EnvLane env_lane_coeffs(const EnvParams& p, const EnvTick& c, uint64_t t_begin, size_t n, bool nice) {
    EnvLane e;
    e.depth = c.depth; e.flat = c.flat; e.on = c.tag == 1u ? 1u : 0u;
    const bool off = c.tag == 2u;
    e.A = e.on ? p.sustain : 0.0; e.B = e.on ? p.one_minus_sustain : (off ? c.off_amp : 0.0);
    e.k = e.on ? p.inv_decay : (off ? p.inv_release : 0.0); e.m0 = e.on ? p.attack_ms : 0.0;
    const uint64_t d0 = t_begin - c.seq;
    e.dt0 = c.tag != 0u ? (uint32_t)d0 : 0u;
    const bool off_ok = !(c.off_amp < 0.0) && (c.off_amp == c.off_amp) && !signbit(c.off_amp) && fabs(c.off_amp) < 1.0e300;
    e.general = (!nice || (c.tag != 0u && ((((d0 + n) >> 32) != 0) || (off && !off_ok)))) ? 1u : 0u;
    return e;
}
// amplifier depth() for sample k of the span (branch-free form; `e.general` lanes are handled by the caller)
template <bool FC>
__device__ __forceinline__ double env_lane_depth(const EnvParams& p, const EnvLane& e, uint32_t k, double one_minus, double mod_depth, double sr, double rsr) {
    const double ms = ms_of_u32(e.dt0 + k, sr, rsr);
    const double tt = e.k * (ms - e.m0);
    const double c = __builtin_fmin(tt, 1.0);                  // clamp()'s upper bound (envelope.rs:21-22); no NaN here, see above
    const double val = mul_add<FC>(e.B, 1.0 - c, e.A);
    const double att = p.inv_attack * ms;
    const double a = (e.on && ms < p.attack_ms) ? att : val;
    const float cc = (float)a;                                 // Envelope stores f32 (envelope.rs:117)
    return amp_depth<FC>(one_minus, mod_depth, (double)cc);    // amplifier.rs:71-73
}

// A run of `n` consecutive samples of one chunk through the recurrence and the epilogue: blocks of EQ_BLK samples with the next
// block's loads in flight, then the ragged tail.  Everything the epilogue needs is in registers before the loop starts -- no
// load sits inside a per-sample condition (its join would cost an s_waitcnt vmcnt(0), i.e. the prefetch).
// ENVK: 0 no inline Envelope; 1 every lane of the wave is flat this tick (constant depth); 2 branch-free closed form;
//       3 general form (env_depth: 64-bit distances, negative off_amplitude).
template <int MODE, bool STEREO, int ENVK, bool FC>
__device__ __forceinline__ void eq_spec_span(const EqDesc& d, const EqRun& r, const float* __restrict__ in, float* __restrict__ outm,
                                             const float* __restrict__ ctl, const size_t n, const EnvTick& cur, const EnvLane& el, uint64_t t,
                                             EqPoles& s, uint32_t& xmin, uint32_t& xmax) {
    const double g_lo = d.gain_lo, g_mid = d.gain_mid, g_hi = d.gain_hi, lo_f = r.lo_f, hi_f = r.hi_f;
    const double one_minus = d.amp_one_minus, mod_depth = d.amp_mod_depth, amplitude = d.amp_amplitude;
    const double depth_const = one_minus + mod_depth * 1.0;                    // Disconnected control: mod value 1.0 (amplifier.rs:54)
    const double rsr = r.rsr;
    uint32_t kk = 0;
    auto fold = [&](float y, float c) -> float {
        if (MODE == EQM_PLAIN) return y;
        double depth;
        if (MODE == EQM_AMP_CONST) depth = depth_const;
        else if (MODE == EQM_AMP_CTL) depth = amp_depth<FC>(one_minus, mod_depth, (double)c);   // amplifier.rs:71-73
        else if (ENVK == 1) depth = el.depth;
        else if (ENVK == 2) { depth = env_lane_depth<FC>(d.env, el, kk, one_minus, mod_depth, r.sr, rsr); ++kk; }
        else { depth = env_depth<FC>(d.env, cur, one_minus, mod_depth, t, r.sr, rsr); ++t; }
        return amp_apply(y, depth, amplitude);
    };
    auto track = [&](float x) { const uint32_t b = __float_as_uint(x); xmin = b < xmin ? b : xmin; xmax = b > xmax ? b : xmax; };
    auto put4 = [&](size_t i, const float (&v)[4]) {
        if (STEREO) {
            f4v a = {v[0], v[0], v[1], v[1]}, b = {v[2], v[2], v[3], v[3]};       // stereo_panner.rs:35-38
            __builtin_nontemporal_store(a, reinterpret_cast<f4v*>(outm + 2 * i));
            __builtin_nontemporal_store(b, reinterpret_cast<f4v*>(outm + 2 * i + 4));
        } else {
            f4v a = {v[0], v[1], v[2], v[3]};
            __builtin_nontemporal_store(a, reinterpret_cast<f4v*>(outm + i));
        }
    };
    const size_t n_blk = n / EQ_BLK;
    f4v xa[EQ_BLK / 4], ca[EQ_BLK / 4];
    if (n_blk) {
#pragma unroll
        for (int q = 0; q < EQ_BLK / 4; ++q) { xa[q] = ld_stream4(in + 4 * q); if (MODE == EQM_AMP_CTL) ca[q] = ld_stream4(ctl + 4 * q); }
    }
    for (size_t b = 0; b < n_blk; ++b) {
        f4v xb[EQ_BLK / 4], cb[EQ_BLK / 4];
        const size_t nb = (b + 1 < n_blk ? b + 1 : b) * EQ_BLK;                 // the last block re-reads itself (in bounds, unused)
#pragma unroll
        for (int q = 0; q < EQ_BLK / 4; ++q) { xb[q] = ld_stream4(in + nb + 4 * q); if (MODE == EQM_AMP_CTL) cb[q] = ld_stream4(ctl + nb + 4 * q); }
#pragma unroll
        for (int q = 0; q < EQ_BLK / 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = xa[q][e];
                track(x);
                v[e] = fold(eq_step<FC>(s, lo_f, hi_f, g_lo, g_mid, g_hi, x), MODE == EQM_AMP_CTL ? ca[q][e] : 0.f);
            }
            put4(b * EQ_BLK + 4 * q, v);
            __builtin_amdgcn_sched_barrier(0);   // four samples at a time: interleaving all sixteen epilogues costs more registers than it hides latency
        }
#pragma unroll
        for (int q = 0; q < EQ_BLK / 4; ++q) { xa[q] = xb[q]; if (MODE == EQM_AMP_CTL) ca[q] = cb[q]; }
    }
    // ragged tail (a tick of 735 samples, the stream's last chunk): at most EQ_BLK - 1 samples, four at a time, then one at a time
    size_t i = n_blk * EQ_BLK;
#pragma unroll 1
    for (; i + 4 <= n; i += 4) {
        const f4v x4 = ld_stream4(in + i);
        f4v c4 = {0.f, 0.f, 0.f, 0.f};
        if (MODE == EQM_AMP_CTL) c4 = ld_stream4(ctl + i);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { track(x4[e]); v[e] = fold(eq_step<FC>(s, lo_f, hi_f, g_lo, g_mid, g_hi, x4[e]), c4[e]); }
        put4(i, v);
    }
#pragma unroll 1
    for (; i < n; ++i) {
        const float x = in[i];
        track(x);
        const float v = fold(eq_step<FC>(s, lo_f, hi_f, g_lo, g_mid, g_hi, x), MODE == EQM_AMP_CTL ? ctl[i] : 0.f);
        if (STEREO) reinterpret_cast<float2*>(outm)[i] = make_float2(v, v); else outm[i] = v;
    }
}

// my chunk: [begin, begin + len) of the instance's stream.  With an inline Envelope the chunk is a whole number of ticks
// (eq_plan_spec) and is walked tick by tick: the tick's Envelope state is loaded once per tick and turned into per-lane coefficients.
template <int MODE, bool STEREO, bool FC>
__device__ __forceinline__ void eq_spec_chunk(const EqDesc& d, const EqRun& r, const EnvTick* __restrict__ ticks, const size_t begin, const size_t len,
                                              EqPoles& s, uint32_t& xmin, uint32_t& xmax) {
    const float* __restrict__ in = d.in + begin;
    float* __restrict__ outm = d.out + (STEREO ? 2 * begin : begin);
    const float* __restrict__ ctl = MODE == EQM_AMP_CTL ? d.ctl + begin : nullptr;
    if (MODE != EQM_AMP_ENV) {
        const EnvTick none{}; const EnvLane nl{};
        eq_spec_span<MODE, STEREO, 0, FC>(d, r, in, outm, ctl, len, none, nl, 0, s, xmin, xmax);
        return;
    }
    const size_t fpc = r.fpc;
    const bool nice = env_params_nice(d.env);
    uint32_t call = (uint32_t)(begin / fpc);
    size_t off = begin % fpc;                                                  // 0 when chunks are whole ticks; the walk is general
    for (size_t i = 0; i < len;) {
        const size_t n = (fpc - off) < (len - i) ? (fpc - off) : (len - i);
        const EnvTick cur = ticks[call];                                       // one 32-byte load per tick of 800 samples
        const uint64_t t = r.t0 + begin + i;
        const EnvLane el = env_lane_coeffs(d.env, cur, t, n, nice);
        float* const o = outm + (STEREO ? 2 * i : i);
        // wave-level choice of the span's form (lanes that left the loop already do not vote)
        if (__ballot(el.general != 0u) != 0ull) eq_spec_span<MODE, STEREO, 3, FC>(d, r, in + i, o, nullptr, n, cur, el, t, s, xmin, xmax);
        else if (__ballot(el.flat == 0u) == 0ull) eq_spec_span<MODE, STEREO, 1, FC>(d, r, in + i, o, nullptr, n, cur, el, t, s, xmin, xmax);
        else eq_spec_span<MODE, STEREO, 2, FC>(d, r, in + i, o, nullptr, n, cur, el, t, s, xmin, xmax);
        i += n; off = 0; ++call;
    }
}

// ---------------------------------------------------------------------------------------------
// one lane per instance, strictly sequential: short streams (the real-time regime: a tick at a time).  The lane walks its stream
// with the chunk loop above -- sixteen samples of loads in flight while the previous sixteen are computed, outputs stored four at
// a time; lanes of a wave may belong to instances with different epilogues (the switch is per lane).
// ---------------------------------------------------------------------------------------------
template <bool FC>
__global__ __launch_bounds__(64) void k_eq_three_exact(const EqDesc* __restrict__ descs, EqState* __restrict__ states, uint32_t n_inst, EqRun r) {
    const uint32_t inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= n_inst) return;
    const EqDesc d = descs[inst];
    EqState st = states[inst];
    EqPoles s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.lo[k] = st.lo[k]; s.hi[k] = st.hi[k]; }
    s.h0 = st.history[0]; s.h1 = st.history[1]; s.h2 = st.history[2];
    uint32_t xmin = 0xffffffffu, xmax = 0u;
    const EnvTick* ticks = r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr;
    const bool stereo = !(d.epi == 0u || (d.flags & MX_EQF_MONO_DUP));
    const int mode = d.epi != 2u ? EQM_PLAIN : ((d.flags & MX_EQF_ENV) ? EQM_AMP_ENV : (d.ctl ? EQM_AMP_CTL : EQM_AMP_CONST));
#define MX_EQ_CASE(M) case M: if (stereo) eq_spec_chunk<M, true, FC>(d, r, ticks, 0, r.frames, s, xmin, xmax); else eq_spec_chunk<M, false, FC>(d, r, ticks, 0, r.frames, s, xmin, xmax); break
    switch (mode) { MX_EQ_CASE(EQM_PLAIN); MX_EQ_CASE(EQM_AMP_CONST); MX_EQ_CASE(EQM_AMP_CTL); default: MX_EQ_CASE(EQM_AMP_ENV); }
#undef MX_EQ_CASE
#pragma unroll
    for (int k = 0; k < 4; ++k) { st.lo[k] = s.lo[k]; st.hi[k] = s.hi[k]; }
    st.history[0] = s.h0; st.history[1] = s.h1; st.history[2] = s.h2;
    states[inst] = st;
}
// ---------------------------------------------------------------------------------------------
// Short streams, few instances (the real-time regime: a tick at a time): the serial part is cut to what IS serial.  The two cascades of an
// instance are independent filters of the same input (eq_three.rs:68-74), so they run in two LANES -- identical code, own coefficient and
// poles -- and leave their last pole per sample (l, the high cascade's p[3]) in a scratch stream; everything after that (band mix
// eq_three.rs:76-88, Panner, Amplifier, Envelope) has no recurrence and runs as a second, sample-parallel kernel.  14 dependent-enough
// instructions per sample and lane instead of 52-72; same operations on the same operands, bit for bit.
// (A systolic variant -- the eight poles in eight lanes, one DPP shift per step -- was also built and measured: bit-exact, no faster than
// one lane per instance: its step is a chain of five dependent f64 instructions plus LDS round trips.  Removed.)
// ---------------------------------------------------------------------------------------------
struct EqPolesScratch { double* p3; double* hist_old; };   // [n][2][frames] last pole per cascade and sample; [n][3] the delay line before the run
template <bool FC>
__global__ __launch_bounds__(64) void k_eq_three_poles(const EqDesc* __restrict__ descs, EqState* __restrict__ states, uint32_t n_inst, EqRun r, EqPolesScratch sc) {
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    if (id >= 2u * n_inst) return;
    const uint32_t inst = id >> 1, c = id & 1u;
    typedef const float __attribute__((address_space(1)))* gfp;
    const gfp in = (gfp)descs[inst].in;
    EqState& st = states[inst];
    double p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = c ? st.hi[k] : st.lo[k];
    const double f = c ? r.hi_f : r.lo_f;
    const size_t N = r.frames;
    double* __restrict__ out = sc.p3 + ((size_t)inst * 2 + c) * N;
    if (c == 0) {   // the delay line: what the epilogue kernel needs from before the run, and what the next run needs from this one
#pragma unroll
        for (int k = 0; k < 3; ++k) sc.hist_old[(size_t)inst * 3 + k] = st.history[k];
    }
    size_t i = 0;
    f4v xa[4];
    if (N >= 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = *reinterpret_cast<const f4v __attribute__((address_space(1)))*>(in + 4 * q);
    }
    for (; i + 16 <= N; i += 16) {                      // sixteen samples of loads in flight while the previous sixteen are computed
        f4v xb[4];
        const size_t nb = i + 32 <= N ? i + 16 : i;
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[q] = *reinterpret_cast<const f4v __attribute__((address_space(1)))*>(in + nb + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pump<FC>(f, p, (double)xa[q][e]);
            typedef double __attribute__((ext_vector_type(2))) d2v;
            __builtin_nontemporal_store(d2v{o[0], o[1]}, reinterpret_cast<d2v*>(out + i + 4 * q));
            __builtin_nontemporal_store(d2v{o[2], o[3]}, reinterpret_cast<d2v*>(out + i + 4 * q + 2));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = xb[q];
    }
    for (; i < N; ++i) out[i] = pump<FC>(f, p, (double)in[i]);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (c) st.hi[k] = p[k]; else st.lo[k] = p[k]; }
    if (c == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // x[N-3 .. N-1]; a run shorter than the delay line keeps part of the old one
            const long long j = (long long)N - 3 + k;
            st.history[k] = j >= 0 ? (double)in[j] : sc.hist_old[(size_t)inst * 3 + (size_t)(j + 3)];
        }
    }
}
// sample-parallel: the band mix and the folded modules for sample i of instance blockIdx.y
template <bool FC>
__global__ __launch_bounds__(256) void k_eq_three_emit(const EqDesc* __restrict__ descs, uint32_t n_inst, EqRun r, EqPolesScratch sc) {
    const uint32_t inst = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (inst >= n_inst || i >= r.frames) return;
    const EqDesc& d = descs[inst];
    const double l = sc.p3[((size_t)inst * 2) * r.frames + i], hp = sc.p3[((size_t)inst * 2 + 1) * r.frames + i];
    const double h0 = i >= 3 ? (double)d.in[i - 3] : sc.hist_old[(size_t)inst * 3 + i];   // eq_three.rs:66,80-83: the input three samples back
    const double h = h0 - hp;
    const double mid = h0 - (h + l);
    const float y = band_mix<FC>(l, mid, h, d.gain_lo, d.gain_mid, d.gain_hi);          // eq_three.rs:76-88
    EqSeqEmit<FC> em;
    em.E = eq_epi_of(d, r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr);
    em.sr = r.sr; em.rsr = r.rsr; em.t0 = r.t0; em.fpc = r.fpc;
    if ((em.E.flags & MX_EQF_ENV) && em.E.epi == 2u) {   // EqSeqEmit::seek with 32-bit arithmetic (the launcher keeps frames below 2^31: a 64-bit division per sample costs more than the sample)
        const uint32_t fpc = (uint32_t)r.fpc, call = (uint32_t)i / fpc;
        em.call = call; em.left = fpc - ((uint32_t)i - call * fpc); em.cur = em.E.ticks[call];
    }
    em.emit(i, y);
}
size_t eq_poles_scratch_bytes(uint32_t n, size_t frames) { return ((size_t)n * 2 * frames + (size_t)n * 3) * sizeof(double); }
bool eq_use_poles_split(uint32_t n, size_t frames) {   // few instances (the chip is mostly idle under one lane per instance), scratch within reason
    static const int below = env_int("MX_EQ_POLES_BELOW", 4097);        // instances; 0 = never (measured: 1 024 strips x 1 tick 77 -> 44 us; 10 240: 104 -> 315 us)
    return (int)n < below && frames >= 1 && frames < ((size_t)1 << 31) && eq_poles_scratch_bytes(n, frames) <= ((size_t)768 << 20);
}
void launch_eq_three_exact(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r, void* scratch, hipStream_t s) {
    if (!n || !r.frames) return;
    if (scratch) {
        EqPolesScratch sc{(double*)scratch, (double*)scratch + (size_t)n * 2 * r.frames};
        if (r.fc) {
            hipLaunchKernelGGL(k_eq_three_poles<true>, dim3((2 * n + 63) / 64), dim3(64), 0, s, d, st, n, r, sc);
            hipLaunchKernelGGL(k_eq_three_emit<true>, dim3((unsigned)((r.frames + 255) / 256), n), dim3(256), 0, s, d, n, r, sc);
        } else {
            hipLaunchKernelGGL(k_eq_three_poles<false>, dim3((2 * n + 63) / 64), dim3(64), 0, s, d, st, n, r, sc);
            hipLaunchKernelGGL(k_eq_three_emit<false>, dim3((unsigned)((r.frames + 255) / 256), n), dim3(256), 0, s, d, n, r, sc);
        }
        return;
    }
    if (r.fc) hipLaunchKernelGGL(k_eq_three_exact<true>, dim3((n + 63) / 64), dim3(64), 0, s, d, st, n, r);
    else hipLaunchKernelGGL(k_eq_three_exact<false>, dim3((n + 63) / 64), dim3(64), 0, s, d, st, n, r);
}

// Where a speculative lane's warm-up starts from.  Zeros are as good as any bounded guess on live input (the warm-up forgets it).  The one
// input class the speculation fails on is a CONSTANT one (a muted strip, DC): the true state stands still a few ulps from the fixed point,
// on the side it came from, and no warm-up from zeros ever arrives there -- every chunk of a muted strip failed its proof and went through
// the repair pass (691 200 of 3 145 728 chunks of bench.py's `material.daw`).  But the state such a strip CARRIES into the run is that very
// standing state: if one more sample of the value my warm-up window begins with leaves the carried state where it is, the lane starts from
// it -- exact as long as the input stays that constant up to my chunk, and a guess like any other if it does not (the proof decides).
template <bool FC>
__device__ __forceinline__ void eq_spec_guess(const EqState& st, const double lo_f, const double hi_f, const float x_first, EqPoles& s) {
    double lo[4], hi[4], nlo[4], nhi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo[k] = nlo[k] = st.lo[k]; hi[k] = nhi[k] = st.hi[k]; }
    pump<FC>(lo_f, nlo, (double)x_first); pump<FC>(hi_f, nhi, (double)x_first);
    bool still = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) still = still && __double_as_longlong(nlo[k]) == __double_as_longlong(lo[k]) && __double_as_longlong(nhi[k]) == __double_as_longlong(hi[k]);
    if (still) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s.lo[k] = lo[k]; s.hi[k] = hi[k]; }
    }
}

// KMODE / KSTEREO >= 0: every instance of the launch has that epilogue (the usual case: a bank of equal strips) and the kernel is
// compiled for it alone -- its own register budget, no dead variants; -1: decided per wave.
template <int KMODE, int KSTEREO, bool FC>
__global__ __launch_bounds__(64, (KMODE == EQM_AMP_ENV || KMODE < 0) ? 3 : 4) void k_eq_three_spec(const EqDesc* __restrict__ descs, const EqState* __restrict__ states, EqRun r, EqSpecPlan plan,
                                                       uint32_t waves_per_inst, EqChunkRec* __restrict__ recs) {
    const uint32_t inst = blockIdx.x / waves_per_inst;                         // wave-uniform: one instance per wave, descriptor in SGPRs
    const uint32_t j = (blockIdx.x % waves_per_inst) * 64u + threadIdx.x;      // my chunk
    if (j >= plan.n_chunks) return;
    const EqDesc& d = descs[inst];
    const size_t C = plan.chunk, W = plan.warm;
    const size_t begin = (size_t)j * C;
    const size_t len = r.frames - begin < C ? r.frames - begin : C;
    EqPoles s;
    // A chunk whose warm-up window would reach the stream's start warms up from THERE, from the carried state: exact, not a guess
    // (chunks may be shorter than the warm-up, eq_plan_spec).  Chunk 0 is that case with nothing to walk.
    const bool from_start = begin <= W;
    const size_t wlen = from_start ? begin : W;
    if (from_start) {
        const EqState& st = states[inst];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s.lo[k] = st.lo[k]; s.hi[k] = st.hi[k]; }
        s.h0 = st.history[0]; s.h1 = st.history[1]; s.h2 = st.history[2];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s.lo[k] = 0.0; s.hi[k] = 0.0; }
        eq_spec_guess<FC>(states[inst], r.lo_f, r.hi_f, d.in[begin - W], s);
    }
    if (wlen) {
        // the exact recurrence over the wlen samples before my chunk (a multiple of EQ_BLK: chunks are multiples of 32 samples)
        const float* __restrict__ in = d.in + (begin - wlen);
        f4v xa[EQ_BLK / 4];
#pragma unroll
        for (int q = 0; q < EQ_BLK / 4; ++q) xa[q] = ld_stream4(in + 4 * q);
        const size_t n_blk = wlen / EQ_BLK;
        for (size_t b = 0; b < n_blk; ++b) {
            f4v xb[EQ_BLK / 4];
            const size_t nb = (b + 1 < n_blk ? b + 1 : b) * EQ_BLK;
#pragma unroll
            for (int q = 0; q < EQ_BLK / 4; ++q) xb[q] = ld_stream4(in + nb + 4 * q);
#pragma unroll
            for (int q = 0; q < EQ_BLK / 4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double x = (double)xa[q][e]; pump<FC>(r.lo_f, s.lo, x); pump<FC>(r.hi_f, s.hi, x); }
            }
#pragma unroll
            for (int q = 0; q < EQ_BLK / 4; ++q) xa[q] = xb[q];
        }
        // the EQ's 3-sample delay line is the input itself: exact
        s.h0 = (double)d.in[begin - 3]; s.h1 = (double)d.in[begin - 2]; s.h2 = (double)d.in[begin - 1];
    }
    EqChunkRec* rec = recs + (size_t)inst * plan.n_chunks + j;
#pragma unroll
    for (int k = 0; k < 4; ++k) { rec->start[k] = s.lo[k]; rec->start[4 + k] = s.hi[k]; }
    uint32_t xmin = 0xffffffffu, xmax = 0u;
    const EnvTick* ticks = r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr;
    const bool stereo = !(d.epi == 0u || (d.flags & MX_EQF_MONO_DUP));
    const int mode = d.epi != 2u ? EQM_PLAIN : ((d.flags & MX_EQF_ENV) ? EQM_AMP_ENV : (d.ctl ? EQM_AMP_CTL : EQM_AMP_CONST));
    if constexpr (KMODE >= 0) {
        eq_spec_chunk<KMODE, KSTEREO != 0, FC>(d, r, ticks, begin, len, s, xmin, xmax);
    } else {
#define MX_EQ_CASE(M) case M: if (stereo) eq_spec_chunk<M, true, FC>(d, r, ticks, begin, len, s, xmin, xmax); else eq_spec_chunk<M, false, FC>(d, r, ticks, begin, len, s, xmin, xmax); break
        switch (mode) { MX_EQ_CASE(EQM_PLAIN); MX_EQ_CASE(EQM_AMP_CONST); MX_EQ_CASE(EQM_AMP_CTL); default: MX_EQ_CASE(EQM_AMP_ENV); }
#undef MX_EQ_CASE
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { rec->end[k] = s.lo[k]; rec->end[4 + k] = s.hi[k]; }
    rec->xmin = xmin; rec->xmax = xmax;
}

// ---------------------------------------------------------------------------------------------
// TILED form of the speculative kernel: the same arithmetic, but the wave moves its 64 chunk streams through LDS.
//
// Why: with one lane per chunk the lanes of a wave are C samples apart in memory.  Loading / storing 16 bytes per lane directly
// (the form above) makes every wave instruction touch 64 different cache lines, 16 bytes each: PMC on MI355X showed one memory
// write request per lane per store (2.3x the algorithmic write bytes), 4 L2 read requests per line, the L1 miss queue full for
// 97 % of the kernel's cycles and the VALU active for 9 % -- 25.7 ms for 1024 strips x 2048 ticks.  Here a SUPER-BLOCK of 32
// samples per lane (one 128-byte line per chunk) is moved per step:
//   stage-in   8 x global_load_lds_dwordx4: instruction k brings the lines of chunks 8k .. 8k+7, eight lanes per line (whole
//              lines, no VGPRs), straight into the tile buffer of the NEXT super-block while this one is computed;
//   tile       [64 chunks][8 slots of 16 B]; chunk row c keeps piece p in slot p ^ ((c >> 1) & 7): the walk of lane j over its own
//              row (ds_read_b128 / ds_write_b128 of slot p ^ ((j >> 1) & 7)) is conflict-free, and rows stay contiguous for the
//              DMA, whose destination is lane-linear -- so the swizzle is applied to the SOURCE address (inverse) and to the READ;
//   compute    lane j walks its row: recurrence + epilogue, results written back in place;
//   stage-out  the tile is read back linearly (lane l of step k holds slot l & 7 of chunk 8k + (l >> 3)) and stored with eight
//              lanes per line: whole lines again.
// The wave needs no barrier (a wave per workgroup; its own vmcnt orders the DMA before its ds_reads).  16 KiB of LDS per wave.
// Conditions (launch_eq_three_spec): frames % 4 == 0, chunk % 32 == 0, frames < 2^30 (a control BUFFER, EQM_AMP_CTL, takes a second tile; before round 4 it kept the
// direct form), and with an inline Envelope samples-per-tick % 32 == 0 (48 kHz: 800).
// ---------------------------------------------------------------------------------------------
typedef const float __attribute__((address_space(1)))* mx_gfp1;
typedef float __attribute__((address_space(3)))* mx_lfp3;
// SB = samples per lane per super-block: 32 (one 128-byte line per chunk, 16 KiB of LDS per wave: 10 waves per CU) or 16 (half
// lines, 8 KiB per wave: the register file, not the LDS, bounds the occupancy).  S = SB / 4 sixteen-byte slots per row; row c
// keeps piece p in slot p ^ sw(c), sw(c) = (c >> log2(16 / S)) & (S - 1): conflict-free ds_read_b128 for either S.
template <int SB> struct EqTileGeo {
    static constexpr int S = SB / 4;                       // slots per row = lanes per row in a DMA instruction
    static constexpr int ROWS = 64 / S;                    // rows per DMA instruction (1 KiB)
    static constexpr int N_INSTR = S;                      // DMA instructions per tile
    static constexpr int SHIFT = S == 8 ? 1 : 2;
    static constexpr int TILE = 64 * SB;                   // floats per tile buffer
    static __device__ __forceinline__ int sw(int row) { return (row >> SHIFT) & (S - 1); }
};

struct EqTileCtx {
    const float* in; float* out;          // stream bases of the instance
    uint32_t chunk0, n_chunks, C, F;      // first chunk of the wave, chunks per instance, chunk length, stream length (samples; < 2^30, launcher)
    int lane;
    // loop-invariant part of the stream index instruction k of a stage-in / stage-out moves: chunk_k * C + 4 * piece_k (the super-block
    // offset `so` is the only thing that changes from step to step: one 32-bit add and one clamp per DMA instruction)
    int base[8];
};

// ALIGNED ROWS (the RT instantiations).  A chunk of 5 880 samples (eight ticks of 735) begins 0 / 96 / 64 / 32 bytes into a 128-byte line: a row of the tile that
// starts AT the chunk's begin straddles two lines in every super-block, and the other half of each is wanted one super-block later, when the L2 has turned over --
// 12.9 GB fetched per launch for 7.4 needed (PMC).  So a row's grid starts at the LINE its chunk begins in: the chunk's sample i sits at tile position i + shift,
// shift = (chunk * C) mod 32 samples (a multiple of 4: whole pieces); every DMA row is a whole line again, and what differs per lane is only where its chunk starts and
// ends inside the first and the last super-block, and where inside a super-block its ticks end (tile positions differ by the shifts: multiples of 4 up to 28 -- 0 / 24 / 16 / 8 for chunks of 5 880 samples).
__device__ __forceinline__ int eq_row_shift(uint32_t chunk, uint32_t C) { return (int)(((chunk & 31u) * (C & 31u)) & 31u); }

// stage-in of the super-block whose first sample sits `so` samples from each chunk's begin (negative during the warm-up)
template <int SB, bool ALIGN_ROWS = false>
__device__ __forceinline__ void eq_tile_bases(EqTileCtx& c) {
    typedef EqTileGeo<SB> G;
    const int s = c.lane % G::S;
#pragma unroll
    for (int k = 0; k < G::N_INSTR; ++k) {
        const int cj = G::ROWS * k + c.lane / G::S;
        const int pce = s ^ G::sw(cj);                                // the piece that belongs in slot s of row cj
        const long long b = (long long)(c.chunk0 + (uint32_t)cj) * (long long)c.C + 4 * pce - (ALIGN_ROWS ? eq_row_shift(c.chunk0 + (uint32_t)cj, c.C) : 0);
        c.base[k] = (int)(b > 0x3fffffffLL ? 0x3fffffffLL : b);       // lanes beyond the stream: any legal value (clamped again below, never used)
    }
}
template <int SB>
__device__ __forceinline__ void eq_tile_issue(const EqTileCtx& c, float* buf, int so, const float* src = nullptr /* another stream of the instance with the input's indexing: the Amplifier's control */) {
    typedef EqTileGeo<SB> G;
    const float* from = src ? src : c.in;
#pragma unroll
    for (int k = 0; k < G::N_INSTR; ++k) {
        // before the stream (chunk 0's warm-up) / past it (lanes beyond the last chunk): never used, keep the address legal
        const int idx = min(max(c.base[k] + so, 0), (int)c.F - 4);    // one v_med3_i32
        __builtin_amdgcn_global_load_lds((mx_gfp1)(from + (size_t)(uint32_t)idx), (mx_lfp3)(buf + k * 256), 16, 0, 0);
    }
}

// The wave-uniform constants of the inner loops, pinned in SGPRs: read through the descriptor reference the compiler
// re-loads them (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS queue) in the middle of the per-sample code.
struct EqK { double lo_f, hi_f, g_lo, g_mid, g_hi, one_minus, mod_depth, amplitude, sr, rsr; EnvParams env; };
__device__ __forceinline__ double pin_sgpr(double v) { asm volatile("; pinned %0" : "+s"(v)); return v; }
__device__ __forceinline__ EqK eq_constants(const EqDesc& d, const EqRun& r) {
    EqK k;
    k.lo_f = pin_sgpr(r.lo_f); k.hi_f = pin_sgpr(r.hi_f);
    k.g_lo = pin_sgpr(d.gain_lo); k.g_mid = pin_sgpr(d.gain_mid); k.g_hi = pin_sgpr(d.gain_hi);
    k.one_minus = pin_sgpr(d.amp_one_minus); k.mod_depth = pin_sgpr(d.amp_mod_depth); k.amplitude = pin_sgpr(d.amp_amplitude);
    k.sr = pin_sgpr(r.sr); k.rsr = pin_sgpr(r.rsr);
    k.env.attack_ms = pin_sgpr(d.env.attack_ms); k.env.inv_attack = pin_sgpr(d.env.inv_attack); k.env.inv_decay = pin_sgpr(d.env.inv_decay);
    k.env.sustain = pin_sgpr(d.env.sustain); k.env.one_minus_sustain = pin_sgpr(d.env.one_minus_sustain); k.env.inv_release = pin_sgpr(d.env.inv_release);
    return k;
}

// compute phase over my row of the tile: ENVK as in eq_spec_span (0: no inline Envelope)
template <int SB, int MODE, int ENVK, bool WARM, bool FC, bool LO_ONLY = false>   // WARM: `len` is the (negative) chunk-relative index of the lane's first warm-up sample; LO_ONLY: the early part of a warm-up, where only the slow cascade runs
__device__ __forceinline__ void eq_tile_compute(const EqK& K, float* buf, const int lane, const int so, const int len,
                                                const EnvTick& cur, const EnvLane& el, EqPoles& s, uint32_t& xmin, uint32_t& xmax, const float* cbuf = nullptr /* EQM_AMP_CTL: the control's tile */) {
    const double g_lo = K.g_lo, g_mid = K.g_mid, g_hi = K.g_hi, lo_f = K.lo_f, hi_f = K.hi_f;
    const double one_minus = K.one_minus, mod_depth = K.mod_depth, amplitude = K.amplitude;
    const double depth_const = one_minus + mod_depth * 1.0;           // Disconnected control: mod value 1.0 (amplifier.rs:54)
    typedef EqTileGeo<SB> G;
    const int sw = G::sw(lane);
    f4v* row = reinterpret_cast<f4v*>(buf + lane * SB);
    f4v x4 = row[0 ^ sw];
#pragma unroll 1   // (unrolling the row fully, or by two, was measured: no faster -- the moves at the back edge go, the code grows 4x)
    for (int pce = 0; pce < G::S; ++pce) {
        const f4v xn = row[((pce + 1) & (G::S - 1)) ^ sw];            // next piece travels while this one is computed
        if (WARM ? (so + 4 * pce >= len) : (so + 4 * pce < len)) {
            // the delay line is "the last three inputs" (s.h0, s.h1, s.h2 = x[i-3], x[i-2], x[i-1]): the four samples of a piece read
            // s.h0, s.h1, s.h2 and the piece's own first input, then the piece's last three inputs become the delay line -- no shifts
            const double dx[4] = {(double)x4[0], (double)x4[1], (double)x4[2], (double)x4[3]};
            if (WARM) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { pump<FC>(lo_f, s.lo, dx[e]); if (!LO_ONLY) pump<FC>(hi_f, s.hi, dx[e]); }
            } else {
                const double hh[4] = {s.h0, s.h1, s.h2, dx[0]};
                f4v v, c4 = {0.f, 0.f, 0.f, 0.f};
                if (MODE == EQM_AMP_CTL) c4 = reinterpret_cast<const f4v*>(cbuf + lane * SB)[pce ^ sw];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t b = __float_as_uint(x4[e]); xmin = b < xmin ? b : xmin; xmax = b > xmax ? b : xmax;
                    const float y = eq_step_h<FC>(s, lo_f, hi_f, g_lo, g_mid, g_hi, dx[e], hh[e]);
                    if (MODE == EQM_PLAIN) v[e] = y;
                    else {
                        double depth;
                        const uint32_t kk = (uint32_t)(so + 4 * pce + e);      // sample index inside the chunk; el.dt0 counts from the tick's first sample
                        if (MODE == EQM_AMP_CONST) depth = depth_const;
                        else if (MODE == EQM_AMP_CTL) depth = amp_depth<FC>(one_minus, mod_depth, (double)c4[e]);   // amplifier.rs:54,71-73
                        else if (ENVK == 1) depth = el.depth;
                        else if (ENVK == 2) depth = env_lane_depth<FC>(K.env, el, kk - el.k0, one_minus, mod_depth, K.sr, K.rsr);
                        else depth = env_depth<FC>(K.env, cur, one_minus, mod_depth, el.t_chunk + kk, K.sr, K.rsr);
                        v[e] = amp_apply(y, depth, amplitude);
                    }
                }
                row[pce ^ sw] = v;
            }
            s.h0 = dx[1]; s.h1 = dx[2]; s.h2 = dx[3];
        }
        x4 = xn;
    }
}

// The samples of a super-block whose chunk-relative index lies in [i_lo, i_hi), one at a time, in the general Envelope form: the first super-block of a chunk (the
// samples before index 0 are still warm-up: OUT = false, the recurrence only), and the one or two super-blocks in which a tick ends (the samples before the boundary
// with the old tick's state, the rest with the new one's).  `so` is the chunk-relative index of the lane's first tile sample -- per lane with aligned rows; the bounds
// are wave-uniform.  Samples outside the range keep what the tile holds: outputs already made, or inputs still to come.
template <int SB, int MODE, bool FC, bool OUT = true>
__device__ __forceinline__ void eq_tile_compute_range(const EqK& K, float* buf, const int lane, const int so, const int len, const int i_lo, const int i_hi,
                                                      const EnvTick& cur, const uint64_t t_chunk, EqPoles& s, uint32_t& xmin, uint32_t& xmax) {
    typedef EqTileGeo<SB> G;
    const int sw = G::sw(lane);
    f4v* row = reinterpret_cast<f4v*>(buf + lane * SB);
#pragma unroll 1
    for (int pce = 0; pce < G::S; ++pce) {
        const int base = so + 4 * pce;
        if (base + 4 > i_lo && base < i_hi && base < len) {           // len is a multiple of 4
            f4v v = row[pce ^ sw];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (base + e >= i_lo && base + e < i_hi) {
                    if (OUT) {
                        const uint32_t b = __float_as_uint(v[e]); xmin = b < xmin ? b : xmin; xmax = b > xmax ? b : xmax;
                        const float y = eq_step<FC>(s, K.lo_f, K.hi_f, K.g_lo, K.g_mid, K.g_hi, v[e]);
                        const double depth = env_depth<FC>(K.env, cur, K.one_minus, K.mod_depth, t_chunk + (uint64_t)(base + e), K.sr, K.rsr);
                        v[e] = MODE == EQM_AMP_ENV ? amp_apply(y, depth, K.amplitude) : y;
                    } else {
                        const double x = (double)v[e];
                        pump<FC>(K.lo_f, s.lo, x); pump<FC>(K.hi_f, s.hi, x);
                        s.h0 = s.h1; s.h1 = s.h2; s.h2 = x;
                    }
                }
            }
            if (OUT) row[pce ^ sw] = v;
        }
    }
}

// stage-out of a computed tile: whole lines, eight lanes per chunk line
// RAGGED (aligned rows, eq_row_shift): a row holds its chunk's samples at tile positions shift .. shift + C - 1; what lies before and behind them in the first and the
// last super-block is the neighbouring chunks', whose lanes write it
template <int SB, bool STEREO, bool RAGGED = false>
__device__ __forceinline__ void eq_tile_store(const EqTileCtx& c, const float* buf, int so) {
    typedef EqTileGeo<SB> G;
#pragma unroll
    for (int k = 0; k < G::N_INSTR; ++k) {
        const f4v o = *reinterpret_cast<const f4v*>(buf + k * 256 + c.lane * 4);
        const int cj = G::ROWS * k + c.lane / G::S;
        const uint32_t chunk = c.chunk0 + (uint32_t)cj;
        const int idx = c.base[k] + so;                               // >= 0: stage-out only happens from the chunk's first super-block on
        bool mine = true;
        if (RAGGED) {
            const int i = so + 4 * ((c.lane % G::S) ^ G::sw(cj)) - eq_row_shift(chunk, c.C);   // chunk-relative index of this piece
            mine = i >= 0 && i + 4 <= (int)c.C;
        }
        if (mine && chunk < c.n_chunks && idx + 4 <= (int)c.F) {
            if (STEREO) {
                f4v a = {o[0], o[0], o[1], o[1]}, b = {o[2], o[2], o[3], o[3]};   // stereo_panner.rs:35-38
                __builtin_nontemporal_store(a, reinterpret_cast<f4v*>(c.out + 2 * idx));
                __builtin_nontemporal_store(b, reinterpret_cast<f4v*>(c.out + 2 * idx + 4));
            } else {
                __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(c.out + idx));
            }
        }
    }
}

// RT ("ragged ticks", inline Envelope only): ticks that are not whole super-blocks -- 735 samples at 44.1 kHz, the reference's own rate -- in chunks of whole ticks that
// are multiples of 4 samples (2 940 = 4 ticks).  An instantiation of its own: the 48 kHz kernels keep their code and their registers.
// A WORKGROUP IS FOUR WAVES (round 5) that never meet: no barrier, each wave has its own tiles in the group's LDS.  The four waves of a 256-thread group go to the four
// SIMDs of one CU by construction.  As 64-thread groups the dispatcher placed the waves one by one, and launches of one or two waves per SIMD came out uneven -- 128 strips x
// 2048 ticks in 1 024 waves: 60 SIMDs with two waves and 60 with none (tools/wave_times.py reads every wave's HW_ID and its time in the launch out of the chunk records);
// the launch lasts as long as its most crowded SIMD: 0.99 ms where the same work spread evenly takes 0.71.
enum { EQ_WPB = 4 };
template <int SB, int KMODE, int KSTEREO, bool FC, int NBUF = 2, bool RT = false>
__global__ __launch_bounds__(64 * EQ_WPB, 4) void k_eq_three_spec_tiled(const EqDesc* __restrict__ descs, const EqState* __restrict__ states, EqRun r, EqSpecPlan plan,
                                                                         uint32_t waves_per_inst, EqChunkRec* __restrict__ recs, uint32_t n_waves) {
    extern __shared__ __attribute__((aligned(16))) float eq_tiles_all[];   // [EQ_WPB][tiles per wave][TILE]
    constexpr int EQ_SB = SB, EQ_TILE = EqTileGeo<SB>::TILE;
    constexpr int EQ_TILES_PER_WAVE = (NBUF == 2 || KMODE == EQM_AMP_CTL) ? 2 : 1;
    const uint32_t t_enter = (uint32_t)__builtin_amdgcn_s_memtime();   // (a wave's life inside the launch, kept in its records' padding: mx_graph_debug_eq_records, tools/wave_times.py)
    // (Graph's tail gate: workgroups are placed in order, so when the last one runs every other one of this launch has its place)
    if (r.started && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(r.started, r.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave in the group: wave-uniform, so that everything derived from it stays in SGPRs
    const uint32_t wave_id = blockIdx.x * EQ_WPB + wib;
    if (wave_id >= n_waves) return;
    float* const eq_tiles = eq_tiles_all + wib * (EQ_TILES_PER_WAVE * EQ_TILE);
    const uint32_t inst = wave_id / waves_per_inst;
    const EqDesc& d = descs[inst];
    const EqK K = eq_constants(d, r);
    EqTileCtx c;
    c.in = d.in; c.out = d.out;
    c.chunk0 = (wave_id % waves_per_inst) * 64u; c.n_chunks = plan.n_chunks; c.C = plan.chunk; c.F = (uint32_t)r.frames;
    c.lane = (int)(threadIdx.x & 63u);
    eq_tile_bases<SB, RT>(c);
    const uint32_t j = c.chunk0 + (uint32_t)c.lane;
    const bool active = j < plan.n_chunks;
    const long long begin = (long long)j * c.C;
    const int len = active ? (int)((long long)c.F - begin < (long long)c.C ? (long long)c.F - begin : (long long)c.C) : 0;
    const int len0 = (int)((long long)c.F - (long long)c.chunk0 * c.C < (long long)c.C ? (long long)c.F - (long long)c.chunk0 * c.C : (long long)c.C);   // the wave's longest chunk (wave-uniform)
    const int row_shift = RT ? eq_row_shift(j, c.C) : 0;   // aligned rows: my chunk's sample i sits at tile position i + row_shift
    const int n_warm = (int)(plan.warm / EQ_SB), n_main = (len0 + (RT ? EQ_SB - 4 : 0) + EQ_SB - 1) / EQ_SB;
    const int total = n_warm + n_main;

    EqPoles s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.lo[k] = 0.0; s.hi[k] = 0.0; }
    s.h0 = s.h1 = s.h2 = 0.0;
    // A chunk whose warm-up window would reach the stream's start warms up from THERE, from the carried state (exact); chunk 0 is that
    // case with nothing to walk.  warm_from: chunk-relative index of my first warm-up sample (a multiple of 4: chunks are multiples of 32).
    const bool from_start = begin <= (long long)plan.warm;
    const int warm_from = from_start ? -(int)begin : -(int)plan.warm;
    if (from_start) {
        const EqState& st = states[inst];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s.lo[k] = st.lo[k]; s.hi[k] = st.hi[k]; }
        s.h0 = st.history[0]; s.h1 = st.history[1]; s.h2 = st.history[2];
    } else if (active) {
        eq_spec_guess<FC>(states[inst], K.lo_f, K.hi_f, d.in[begin - (long long)plan.warm], s);
    }
    uint32_t xmin = 0xffffffffu, xmax = 0u;
    const EnvTick* ticks = r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr;
    EnvTick cur{}; EnvLane el{};
    const bool nice = env_params_nice(d.env);
    EqChunkRec* rec = recs + (size_t)inst * plan.n_chunks + (active ? j : 0);

    // One loop per PHASE, not one loop that picks its phase every step: the warm-up, and then -- per tick -- the one epilogue form the tick
    // needs.  With a single loop over the super-blocks that dispatched to the four compute variants, the compiler gave each variant its
    // own register assignment for the carried state (eight f64 poles, the delay line, min / max) and moved it back to a common one at the
    // back edge: 44 v_mov_b64 per super-block, 3 instructions per sample that no sample needed (ISA; PMC 79 -> 77 per output sample with
    // the index arithmetic of the DMA, see DESIGN.md 5.2 ledger).  Here a variant's loop keeps the state where it is; moves happen once per tick.
    // NBUF == 1 (with SB = 32): ONE tile of WHOLE 128-byte lines.  Half lines (SB = 16) cost the memory system: the L2 fetches whole lines, the
    // other half is asked for a super-block of compute later, and by then 44 % of the lines have left the 4 MiB L2 that 512 such waves stream
    // through (tools/fetch_probe.hip: 1.44x the bytes on the fabric; this kernel's FETCH_SIZE said the same).  Whole lines need 8 KiB per tile
    // and a second tile would halve the waves a CU holds -- so there is none: a super-block's lines are asked for when the one before is
    // stored, and the round trip is hidden by the other three waves of the SIMD instead of by a buffer.
    if constexpr (NBUF == 2) eq_tile_issue<SB>(c, eq_tiles, -(int)plan.warm);
    auto begin_sb = [&](int g) -> float* {
        const int so = (g - n_warm) * EQ_SB;
        if constexpr (NBUF == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage-out's reads of the tile are done before the DMA may land on it
            eq_tile_issue<SB>(c, eq_tiles, so);
            if constexpr (KMODE == EQM_AMP_CTL) { if (so >= 0) eq_tile_issue<SB>(c, eq_tiles + EQ_TILE, so, d.ctl); }   // the control of the same samples, into a tile of its own
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return eq_tiles;
        }
        if (g + 1 < total) eq_tile_issue<SB>(c, eq_tiles + ((g + 1) & 1) * EQ_TILE, so + EQ_SB);   // its previous tenant was stored one step ago
        // everything but the DMA just issued has landed: this super-block's tile, and the stores of the one before
        if (g + 1 < total) { if (SB == 32) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return eq_tiles + (g & 1) * EQ_TILE;
    };
    // The two cascades are independent filters of one input (eq_three.rs:68-74) and forget at their own rates: the high one (2 700 Hz, p = 0.65 at 48 kHz) has
    // coalesced after plan.warm_hi samples (eq_warm_len(hi_f): 256 where the low one, 420 Hz, needs 1 280), so a lane that speculates runs it over the LAST warm_hi
    // samples of its window only -- 13 of the warm-up's 26 f64 operations per sample saved over four fifths of it.  A lane that warms up from the stream's start
    // runs both from there (it starts from the exact carried state: nothing may be skipped); the wave's early, low-only super-blocks end where its first lane needs both.
    const int hi_from = (active && from_start) ? warm_from : -(int)plan.warm_hi;
    int hi_min = hi_from;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) hi_min = min(hi_min, __shfl_xor(hi_min, m, 64));
    const int n_lo = max(0, min(n_warm, n_warm + (hi_min - (EQ_SB - 1)) / EQ_SB));   // super-blocks that end at or before hi_min (hi_min <= 0: C division rounds towards zero, hence the bias)
    int g = 0;
    for (; g < n_lo; ++g) {
        float* buf = begin_sb(g);
        eq_tile_compute<SB, EQM_PLAIN, 0, true, FC, true>(K, buf, c.lane, (g - n_warm) * EQ_SB - row_shift, warm_from, cur, el, s, xmin, xmax);
    }
    for (; g < n_warm; ++g) {
        float* buf = begin_sb(g);
        eq_tile_compute<SB, EQM_PLAIN, 0, true, FC>(K, buf, c.lane, (g - n_warm) * EQ_SB - row_shift, warm_from, cur, el, s, xmin, xmax);
    }
    if (!RT && active) {   // first sample of my chunk: record where the warm-up took me (chunks that started at the stream's start: the exact state)
#pragma unroll
        for (int k = 0; k < 4; ++k) { rec->start[k] = s.lo[k]; rec->start[4 + k] = s.hi[k]; }
    }
    if constexpr (KMODE == EQM_AMP_ENV && RT) {
        // Chunks are whole ticks (launcher), so every lane is at the same place inside its tick -- in chunk-relative samples; its tile position is that plus its row's
        // shift (aligned rows, eq_row_shift: 0 .. 28).  A super-block whose samples are one tick's for every lane runs in that tick's form as at 48 kHz; the first
        // super-block of the chunk (the samples before the chunk's begin are the end of the warm-up) and the one or two in which a tick ends for some lane are walked
        // sample by sample in the general form (eq_tile_compute_range), the old tick's state before the boundary and the new one's behind it.
        const int fpc = (int)r.fpc;
        int tick_i = 0, so0 = 0;
        auto tick_at = [&](int ti) { const size_t tk = ((size_t)begin + (size_t)ti * (size_t)fpc) / r.fpc; return ticks[tk < r.n_calls ? tk : r.n_calls - 1]; };
        // (a tick's EnvTick is fetched where it is used, not carried: eight registers that the per-sample loops of the other forms would have to leave alone)
        const uint64_t t_chunk = r.t0 + (uint64_t)begin;
        auto load_tick = [&]() {                     // the coefficients of tick tick_i of every lane's chunk, per lane
            so0 = tick_i * fpc;
            const EnvTick ct = tick_at(tick_i);
            el = env_lane_coeffs(K.env, ct, t_chunk + (uint64_t)so0, r.fpc, nice);
            el.k0 = (uint32_t)so0; el.t_chunk = t_chunk;
        };
        load_tick();
        {   // the chunk's first super-block
            float* buf = begin_sb(g);
            const EnvTick cur0 = tick_at(0);
            // (the warm-up part has no upper bound of its own: `len` only has to admit the negative indices -- a lane without a chunk walks them too, harmlessly)
            eq_tile_compute_range<SB, KMODE, FC, false>(K, buf, c.lane, -row_shift, len > 0 ? len : 4, -EQ_SB, 0, cur0, t_chunk, s, xmin, xmax);
            if (active) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { rec->start[k] = s.lo[k]; rec->start[4 + k] = s.hi[k]; }
            }
            eq_tile_compute_range<SB, KMODE, FC>(K, buf, c.lane, -row_shift, len, 0, 0x3fffffff, cur0, t_chunk, s, xmin, xmax);
            eq_tile_store<SB, KSTEREO != 0, true>(c, buf, 0);
            ++g;
        }
        while (g < total) {
            int so_g = (g - n_warm) * EQ_SB;
            const int t_end = so0 + fpc;             // chunk-relative index of the next tick's first sample
            if (so_g + EQ_SB <= t_end) {
                const int envk = __ballot(active && so0 < len && el.general != 0u) != 0ull ? 3 : (__ballot(active && so0 < len && el.flat == 0u) == 0ull ? 1 : 2);
                const int g_int = n_warm + t_end / EQ_SB;                      // super-blocks that end at or before t_end for the unshifted rows (the shifted ones end earlier)
                const int g_end = g_int < total ? g_int : total;
                if (envk == 1) {
                    for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 1, false, FC>(K, buf, c.lane, so - row_shift, len, cur, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0, true>(c, buf, so); }
                } else if (envk == 2) {
                    for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 2, false, FC>(K, buf, c.lane, so - row_shift, len, cur, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0, true>(c, buf, so); }
                } else {
                    const EnvTick cur3 = tick_at(tick_i);
                    for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 3, false, FC>(K, buf, c.lane, so - row_shift, len, cur3, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0, true>(c, buf, so); }
                }
            } else {
                // the tick ends inside this super-block for the rows with the smallest shift, up to 28 samples later for the others
                const EnvTick now = tick_at(tick_i), nxt = tick_at(tick_i + 1);
                for (; g < total && so_g - (EQ_SB - 4) < t_end; ++g, so_g += EQ_SB) {
                    float* buf = begin_sb(g);
                    eq_tile_compute_range<SB, KMODE, FC>(K, buf, c.lane, so_g - row_shift, len, -0x3fffffff, t_end, now, t_chunk, s, xmin, xmax);
                    eq_tile_compute_range<SB, KMODE, FC>(K, buf, c.lane, so_g - row_shift, len, t_end, 0x3fffffff, nxt, t_chunk, s, xmin, xmax);
                    eq_tile_store<SB, KSTEREO != 0, true>(c, buf, so_g);
                }
                ++tick_i; load_tick();
            }
        }
    } else if constexpr (KMODE == EQM_AMP_ENV) {
        const int sb_per_tick = (int)(r.fpc / EQ_SB);   // chunks are whole ticks, ticks whole super-blocks (launcher)
        while (g < total) {
            // a new tick (wave-uniform): its Envelope state, per lane
            const int so0 = (g - n_warm) * EQ_SB;
            const size_t tk = ((size_t)begin + (size_t)so0) / r.fpc;
            cur = ticks[tk < r.n_calls ? tk : r.n_calls - 1];
            const uint64_t t = r.t0 + (uint64_t)begin + (uint64_t)so0;
            el = env_lane_coeffs(K.env, cur, t, r.fpc, nice);
            el.k0 = (uint32_t)so0; el.t_chunk = r.t0 + (uint64_t)begin;
            const int envk = __ballot(active && so0 < len && el.general != 0u) != 0ull ? 3 : (__ballot(active && so0 < len && el.flat == 0u) == 0ull ? 1 : 2);
            const int g_end = g + sb_per_tick < total ? g + sb_per_tick : total;
            if (envk == 1) {
                for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 1, false, FC>(K, buf, c.lane, so, len, cur, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0>(c, buf, so); }
            } else if (envk == 2) {
                for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 2, false, FC>(K, buf, c.lane, so, len, cur, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0>(c, buf, so); }
            } else {
                for (; g < g_end; ++g) { float* buf = begin_sb(g); const int so = (g - n_warm) * EQ_SB; eq_tile_compute<SB, KMODE, 3, false, FC>(K, buf, c.lane, so, len, cur, el, s, xmin, xmax); eq_tile_store<SB, KSTEREO != 0>(c, buf, so); }
            }
        }
    } else {
        for (; g < total; ++g) {
            float* buf = begin_sb(g);
            const int so = (g - n_warm) * EQ_SB;
            eq_tile_compute<SB, KMODE, 0, false, FC>(K, buf, c.lane, so, len, cur, el, s, xmin, xmax, KMODE == EQM_AMP_CTL ? buf + EQ_TILE : nullptr);
            eq_tile_store<SB, KSTEREO != 0>(c, buf, so);
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { rec->end[k] = s.lo[k]; rec->end[4 + k] = s.hi[k]; }
        rec->xmin = xmin; rec->xmax = xmax;
        // lane 0: when the wave entered; lane 1: where it ran (HW_ID: wave slot, SIMD, CU, SE); lane 2: its XCD; every lane: when it left
        uint32_t hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec->pad[0] = c.lane == 0 ? t_enter : (c.lane == 1 ? hw : (c.lane == 2 ? xcc : 0u));
        rec->pad[1] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
}

// ---- repair ----
// The repair pass works in GROUPS of four consecutive lanes.  A walk runs two trajectories side by side -- A from the true state (the
// sequential order) and B, the one the chunk's speculative lane ran -- and each is two independent cascades of the same input
// (eq_three.rs:68-74): four recurrences of identical code, one per lane (role 0: A low, 1: A high, 2: B low, 3: B high), thirteen dependent
// f64 instructions per sample instead of the ~110 one lane needed for all of it (the wave pays an instruction's issue slot whether one lane
// or sixty-four run it).  What has no recurrence -- the band mix, the folded Panner / Amplifier / Envelope, the stores -- is done by the
// four lanes together, sixteen samples at a time, from the last poles the A lanes leave in LDS.  Every lane of a group takes the same
// branches: decisions are made on values shared through ballots and shuffles inside the group.  Groups are independent of each other
// (up to sixteen islands of a stream side by side).
__device__ __forceinline__ bool same4(const double (&a)[4], const double* b) {
    bool eq = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) eq = eq && (__double_as_longlong(a[k]) == __double_as_longlong(b[k]));
    return eq;
}
__device__ __forceinline__ bool same8(const double (&a)[8], const double* b) {
    bool eq = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) eq = eq && (__double_as_longlong(a[k]) == __double_as_longlong(b[k]));
    return eq;
}
__device__ __forceinline__ bool all_nan4(const double (&a)[4]) { return a[0] != a[0] && a[1] != a[1] && a[2] != a[2] && a[3] != a[3]; }
__device__ __forceinline__ bool eq_all_nan(const double (&E)[8]) {
    bool a = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) a = a && (E[k] != E[k]);
    return a;
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)u, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(u >> 32), src, 64);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// true when the condition holds in all four lanes of the group that starts at lane g0 (all four execute this together)
__device__ __forceinline__ bool grp_all(bool c, int g0) { return ((__ballot(c) >> g0) & 0xFull) == 0xFull; }
// LDS executes a wave's own accesses in order, so what one lane wrote is there for another lane's later read without any wait; this only
// keeps the compiler from moving either across the hand-over (a fence would also drain the wave's global loads: the input prefetch)
__device__ __forceinline__ void lds_handover() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// the EQ's f32 (eq_three.rs:70-85) from the two cascades' last poles AFTER the sample was pumped in, and the input three samples back
template <bool FC>
__device__ __forceinline__ float eq_out_of(double l, double hp, double h0, double g_lo, double g_mid, double g_hi) {
    const double h = h0 - hp;
    const double mid = h0 - (h + l);
    return band_mix<FC>(l, mid, h, g_lo, g_mid, g_hi);
}
// four samples from index i of a stream of n: one 16-byte load where all four exist, the ragged tail element by element
__device__ __forceinline__ f4v eq_ld4(const float* __restrict__ p, size_t i, size_t n) {
    if (i + 4 <= n) return *reinterpret_cast<const f4v*>(p + i);
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) v[0] = p[i];
    if (i + 1 < n) v[1] = p[i + 1];
    if (i + 2 < n) v[2] = p[i + 2];
    return v;
}

// The folded epilogue for a lane that emits four consecutive samples out of every sixteen (EqSeqEmit's arithmetic, with a tick cursor that
// can step over the samples its neighbours emit): `call` is the tick of the cursor's sample, `left` what remains of that tick counting it.
template <bool FC>
struct EqBlkEmit {
    EqEpi E; double sr, rsr; uint64_t t0; size_t fpc;
    size_t left = 1; uint32_t call = 0, cur_call = 0xffffffffu; EnvTick cur{}; bool env = false;
    __device__ __forceinline__ void init(const EqDesc& d, const EqRun& r, uint32_t inst) {
        E = eq_epi_of(d, r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr);
        sr = r.sr; rsr = r.rsr; t0 = r.t0; fpc = r.fpc ? r.fpc : 1;
        env = (E.flags & MX_EQF_ENV) && E.epi == 2u && E.ticks;
    }
    __device__ __forceinline__ void seek(size_t i) { if (env) { call = (uint32_t)(i / fpc); left = fpc - i % fpc; } }
    __device__ __forceinline__ void skip(size_t n) {
        if (!env) return;
        while (n >= left) { n -= left; ++call; left = fpc; }
        left -= n;
    }
    __device__ __forceinline__ void emit(size_t i, float y) {   // the cursor is on sample i; it moves to i + 1
        float v = y;
        if (E.epi == 2u) {
            double depth;
            if (env) {
                if (call != cur_call) { cur = E.ticks[call]; cur_call = call; }
                depth = env_depth<FC>(E.env, cur, E.amp_one_minus, E.amp_mod_depth, t0 + i, sr, rsr);
                if (--left == 0) { ++call; left = fpc; }
            } else {
                const double m = E.ctl ? (double)E.ctl[i] : 1.0;                       // amplifier.rs:54
                depth = amp_depth<FC>(E.amp_one_minus, E.amp_mod_depth, m);           // amplifier.rs:71-73
            }
            v = amp_apply(y, depth, E.amp_amplitude);
        }
        if (eq_mono_out(E)) E.out[i] = v;                                             // one float per frame
        else reinterpret_cast<float2*>(E.out)[i] = make_float2(v, v);                 // stereo_panner.rs:35-38
    }
};

// ---- the walk: chunks [j0, j_limit) in order with the TRUE state E at the start of chunk j0 in hand; called by the four lanes of a group ----
// A chunk whose recorded start equals the true state is exact as the speculative lane left it (E moves to its recorded end).  Any other:
//   * as long as the input stays what the chunk's first sample is and both trajectories stand still under it, the f32 outputs are a
//     function of (state, delay-line value) only -- compared for the delay-line values in question; equal => what the lane wrote there is
//     the sequential order's and the true state stays where it is.  A chunk whose whole input is that constant (record: min == max; every
//     chunk deep inside a silence) is settled by that comparison alone, O(1); otherwise the input is scanned for its first change (the chunk
//     in which programme returns after a silence) and the walk starts there;
//   * from there both trajectories are re-run, EQ_RB samples per step, and every output whose f32 differs from what the lane wrote is
//     replaced, until
//       - the trajectories coalesce: from there on the lane's run IS the sequential order (E = its recorded end), or
//       - the input is constant to the chunk's end and both stand still: the rest is compared as above and, if it differs, filled from A, or
//       - the chunk ends (E = A).
// On return E is the true state at the start of chunk j_end; j_end = j_limit unless the walk stopped in front of an ABSORBING state (all eight
// poles NaN: no later chunk can ever match -- the caller fills the rest of the stream with the whole wave instead of walking it).  Chunks
// below `force_until` are rewritten sample for sample whatever is recorded about them (a fallback after an island that started from a wrong
// assumption: memory there may hold that island's rewrites, not the speculative outputs).
struct EqWalk { uint32_t j_end; uint32_t repaired; uint32_t settled /* chunks the O(1) comparison settled */, steps /* walk steps of EQ_RB samples */, fills /* fill steps */; };
constexpr int EQ_RB = 16;   // samples per step of a walk
template <bool FC>
__device__ __forceinline__ EqWalk eq_repair_walk(const EqDesc& d, const EqRun& r, const EqSpecPlan& plan, const EqChunkRec* __restrict__ rc, uint32_t inst,
                                                 uint32_t j0, uint32_t j_limit, uint32_t force_until, double (&E)[8], const int lane,
                                                 double* __restrict__ p3g /* this group's [4][EQ_RB]: the four cascades' last poles of a step */) {
    const int role = lane & 3, g0 = lane & ~3, half = role & 1;
    const bool is_a = role < 2;
    const double f = half ? r.hi_f : r.lo_f;
    const double g_lo = d.gain_lo, g_mid = d.gain_mid, g_hi = d.gain_hi;
    const size_t C = plan.chunk;
    EqWalk w{j_limit, 0u, 0u, 0u, 0u};
    double Q[4];                                   // the cascade this lane runs: A's (roles 0, 1: always the true state's half), B's inside a walk
#pragma unroll
    for (int k = 0; k < 4; ++k) Q[k] = E[4 * half + k];
    EqBlkEmit<FC> em; em.init(d, r, inst);
    // the four last poles of the group after one more sample: A low, A high, B low, B high
    auto outs_differ = [&](const double q3, const double hv) {
        const double a_lo = shfl_f64(q3, g0), a_hi = shfl_f64(q3, g0 + 1), b_lo = shfl_f64(q3, g0 + 2), b_hi = shfl_f64(q3, g0 + 3);
        return __float_as_uint(eq_out_of<FC>(a_lo, a_hi, hv, g_lo, g_mid, g_hi)) != __float_as_uint(eq_out_of<FC>(b_lo, b_hi, hv, g_lo, g_mid, g_hi));
    };
    for (uint32_t j = j0; j < j_limit; ++j) {
        const EqChunkRec& R = rc[j];
        const double* rs = R.start + 4 * half; const double* re = R.end + 4 * half;
        const bool force = j < force_until;
        if (!force && grp_all(!is_a || same4(Q, rs), g0)) {   // the speculative chunk started from the true state: everything it wrote is exact
            if (is_a) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Q[k] = re[k];
            }
            continue;
        }
        if (grp_all(!is_a || all_nan4(Q), g0)) { w.j_end = j; break; }
        ++w.repaired;
        const size_t begin = (size_t)j * C;
        const size_t len = r.frames - begin < C ? r.frames - begin : C;
        const float* __restrict__ xin = d.in + begin;     // begin >= one chunk: xin[-3 .. -1] exist (chunk 0 is never repaired)
        if (!is_a) {
#pragma unroll
            for (int k = 0; k < 4; ++k) Q[k] = rs[k];     // B: the trajectory the chunk's lane ran
        }
        const bool konst = R.xmin == R.xmax;
        // does one more sample of x leave my cascade where it is?  (then every further sample of x does, too)
        auto stands_still = [&](const double x, double& q3_after) {
            double n[4] = {Q[0], Q[1], Q[2], Q[3]};
            pump<FC>(f, n, x);
            q3_after = n[3];
            return same4(n, Q);
        };
        size_t i0 = 0;
        if (!force) {
            const float x0f = konst ? __uint_as_float(R.xmin) : xin[0];
            const double x0 = (double)x0f;
            double q3;
            if (grp_all(stands_still(x0, q3), g0)) {
                const double hv = role < 3 ? (double)xin[role - 3] : x0;          // the four delay-line values in question: x[-3], x[-2], x[-1], x0
                if (grp_all(!outs_differ(q3, hv), g0) && !(plan.pad & 1u)) {   // (plan.pad: MX_EQ_REPAIR_TEST, see launch_eq_three_spec)
                    if (konst) { ++w.settled; continue; }                         // E unchanged: the state stands still through the chunk
                    // where does the input change?  128 samples per round trip to memory: the four lanes take four samples each of eight steps
                    const uint32_t b0 = __float_as_uint(x0f);
                    size_t is = 0;
                    int first = 8;
                    for (; is < len; is += 8 * EQ_RB) {
                        f4v v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = eq_ld4(xin, is + (size_t)(EQ_RB * u + 4 * role), len);
                        first = 8;
#pragma unroll
                        for (int u = 7; u >= 0; --u) {
                            bool df = false;
#pragma unroll
                            for (int e = 0; e < 4; ++e) df = df || (is + (size_t)(EQ_RB * u + 4 * role + e) < len && __float_as_uint(v[u][e]) != b0);
                            if (df) first = u;
                        }
                        first = min(first, __shfl_xor(first, 1, 64)); first = min(first, __shfl_xor(first, 2, 64));
                        if (first < 8) break;
                    }
                    if (first >= 8) { ++w.settled; continue; }                    // (the record said the input changes; it does not: settled like a constant chunk)
                    i0 = is + (size_t)(EQ_RB * first);                            // nothing moves and nothing differs before the step the change is in
                }
            }
        }
        // both trajectories, EQ_RB samples per step; the input travels a step ahead of the walk
        em.seek(begin + i0 + (size_t)(4 * role));
        bool coalesced = false;
        const double xc = (double)__uint_as_float(R.xmin);
        f4v xa[4], xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = eq_ld4(xin, i0 + (size_t)(4 * q), len);
        for (; i0 < len; i0 += EQ_RB) {
            ++w.steps;
#pragma unroll
            for (int q = 0; q < 4; ++q) xb[q] = eq_ld4(xin, i0 + EQ_RB + (size_t)(4 * q), len);
            float xh[4];                               // the inputs three samples back of the four samples this lane emits
#pragma unroll
            for (int k = 0; k < 4; ++k) { const size_t i = i0 + (size_t)(4 * role + k); xh[k] = i < len ? xin[(long long)i - 3] : 0.f; }
            const int nb = len - i0 < (size_t)EQ_RB ? (int)(len - i0) : EQ_RB;
            if (nb == EQ_RB) {
#pragma unroll
                for (int e = 0; e < EQ_RB; ++e) { pump<FC>(f, Q, (double)xa[e >> 2][e & 3]); p3g[role * EQ_RB + e] = Q[3]; }
            } else {
#pragma unroll
                for (int e = 0; e < EQ_RB; ++e) if (e < nb) { pump<FC>(f, Q, (double)xa[e >> 2][e & 3]); p3g[role * EQ_RB + e] = Q[3]; }
            }
            lds_handover();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * role + k;
                if (e < nb) {
                    const float ya = eq_out_of<FC>(p3g[e], p3g[EQ_RB + e], (double)xh[k], g_lo, g_mid, g_hi);
                    // what the chunk's lane wrote here went through the same epilogue: only an EQ sample that differs has to be emitted again
                    if (force || __float_as_uint(ya) != __float_as_uint(eq_out_of<FC>(p3g[2 * EQ_RB + e], p3g[3 * EQ_RB + e], (double)xh[k], g_lo, g_mid, g_hi)))
                        em.emit(begin + i0 + (size_t)e, ya);
                    else em.skip(1);
                }
            }
            lds_handover();
            if (nb < EQ_RB) { i0 = len; break; }       // the stream's ragged end
            em.skip(EQ_RB - 4);
            if (!force) {
                double P2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) P2[k] = shfl_f64(Q[k], lane ^ 2);
                if (grp_all(same4(Q, P2), g0)) { coalesced = true; break; }               // from here on the lane's run IS the sequential order
                if (konst) {
                    double q3;
                    if (grp_all(stands_still(xc, q3), g0)) {                              // neither trajectory moves again in this chunk
                        i0 += EQ_RB;
                        if (outs_differ(q3, xc) || (plan.pad & 1u)) {                     // the same in all four lanes (every delay-line value from here on is xc)
                            const double a_lo = shfl_f64(q3, g0), a_hi = shfl_f64(q3, g0 + 1);
                            const float yc = eq_out_of<FC>(a_lo, a_hi, xc, g_lo, g_mid, g_hi);
                            for (; i0 < len; i0 += EQ_RB) {
                                ++w.fills;
#pragma unroll
                                for (int k = 0; k < 4; ++k) { const size_t i = i0 + (size_t)(4 * role + k); if (i < len) em.emit(begin + i, yc); }
                                em.skip(EQ_RB - 4);
                            }
                        }
                        break;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[q] = xb[q];
        }
        if (coalesced && is_a) {
#pragma unroll
            for (int k = 0; k < 4; ++k) Q[k] = re[k];
        }
        // else: roles 0, 1 hold A's state at the chunk's end (walked to the end, or standing still until it)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { E[k] = shfl_f64(Q[k], g0); E[4 + k] = shfl_f64(Q[k], g0 + 1); }
    return w;
}
// the rest of the stream from chunk j on under an absorbing (all-NaN) state, all 64 lanes: every output is the EQ's f32 for (that
// state pumped once with the sample's input, the sample's delay-line value) through the folded epilogue, sample by sample independently
template <bool FC>
__device__ __forceinline__ void eq_nan_fill(const EqDesc& d, const EqRun& r, const EqSpecPlan& plan, uint32_t inst, uint32_t j, const double (&E)[8], int lane) {
    const size_t begin = (size_t)j * plan.chunk;
    const size_t unit = r.fpc ? r.fpc : 1024;                      // a lane takes whole ticks: its Envelope cursor starts at a tick's first sample
    const size_t n_units = (r.frames - begin + unit - 1) / unit;   // begin is a multiple of a tick when an Envelope is folded in (eq_plan_spec)
    for (size_t u = (size_t)lane; u < n_units; u += 64) {
        const size_t a = begin + u * unit, b = a + unit < r.frames ? a + unit : r.frames;
        EqSeqEmit<FC> em;
        em.E = eq_epi_of(d, r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr);
        em.sr = r.sr; em.rsr = r.rsr; em.t0 = r.t0; em.fpc = r.fpc;
        em.seek(a);
        double h0 = (double)d.in[a - 3], h1 = (double)d.in[a - 2], h2 = (double)d.in[a - 1];
        for (size_t i0 = a; i0 < b; i0 += 16) {                                 // sixteen samples per request
            f4v xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = eq_ld4(d.in, i0 + (size_t)(4 * q), b);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t i = i0 + (size_t)e;
                if (i < b) {
                    const double x = (double)xv[e >> 2][e & 3];
                    double lo[4] = {E[0], E[1], E[2], E[3]}, hi[4] = {E[4], E[5], E[6], E[7]};
                    pump<FC>(r.lo_f, lo, x); pump<FC>(r.hi_f, hi, x);
                    em.emit(i, eq_out_of<FC>(lo[3], hi[3], h0, d.gain_lo, d.gain_mid, d.gain_hi));
                    h0 = h1; h1 = h2; h2 = x;
                }
            }
        }
    }
}

// One wave per instance.  First every boundary is checked in parallel against the SPECULATIVE end state of the chunk before
// it: if all of them match, induction from chunk 0 (which started from the carried state) proves the whole stream -- the only case
// on live signals.  Where a boundary fails (the input was exactly CONSTANT for long: digital silence or DC after a signal -- the
// poles stall a few ulps from the fixed point, on the side they came from, and trajectories stalled at different values never meet;
// or a NaN) the stream is walked on from there with the true state in hand (eq_repair_walk).
// ISLANDS.  Programme that falls silent and comes back -- a desk's normal material -- fails a few boundaries after every onset of
// silence, seconds apart.  Each run of failures starts at a HEAD: a failing boundary whose predecessor matched.  If the chunk before a
// head is exact, the true state at the head is that chunk's recorded end -- which every island can ASSUME and walk at once, one GROUP of
// four lanes per head (up to 16 per round), each up to the next head.  The assumptions are then checked in stream order: island i held if
// island i - 1 ended on the recorded end of its last chunk.  Where one did not, everything from there to the end of the round is walked
// again in order from the true state, rewriting every sample (memory there may hold a wrong island's rewrites).  An absorbing all-NaN
// state is never walked: the rest of the stream is filled by the whole wave (eq_nan_fill).
constexpr int EQ_ISLANDS = 16;
template <bool FC>
__global__ __launch_bounds__(64) void k_eq_three_repair(const EqDesc* __restrict__ descs, EqState* __restrict__ states, EqRun r, EqSpecPlan plan,
                                                         const EqChunkRec* __restrict__ recs, unsigned long long* __restrict__ stats, uint32_t tail) {
    const uint32_t inst = blockIdx.x;
    const int lane = threadIdx.x, grp = lane >> 2;
    const EqDesc& d = descs[inst];
    const EqChunkRec* rc = recs + (size_t)inst * plan.n_chunks;
    const uint32_t NC = plan.n_chunks;
    __shared__ uint32_t heads[EQ_ISLANDS + 1];
    __shared__ double p3s[EQ_ISLANDS * 4 * EQ_RB];
    double* p3g = p3s + grp * 4 * EQ_RB;
    auto fails = [&](uint32_t j) {                 // boundary j (1 <= j < NC): chunk j's recorded start differs from chunk j - 1's recorded end
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) ok = ok && __double_as_longlong(rc[j].start[k]) == __double_as_longlong(rc[j - 1].end[k]);
        return !ok;
    };
    double E[8];                                   // wave-uniform.  have_E: the true state at the start of chunk `pos`, and NOT chunk pos - 1's recorded end
#pragma unroll
    for (int k = 0; k < 8; ++k) E[k] = 0.0;
    bool have_E = false;
    uint32_t pos = 1, force_until = 0;             // every chunk before `pos` is proven
    unsigned long long repaired = 0, settled = 0, steps = 0, fills = 0, rounds = 0, in_order = 0, nan_fills = 0;
    while (pos < NC) {
        if (have_E) {
            if (eq_all_nan(E)) { repaired += NC - pos; ++nan_fills; eq_nan_fill<FC>(d, r, plan, inst, pos, E, lane); pos = NC; break; }
            // in order from the true state, by the first group: to the end of the round that went wrong (rewriting everything), else one chunk
            // at a time -- as soon as a chunk ends on its recorded end the islands behind it are searched (and walked side by side) again
            const uint32_t lim = force_until > pos ? force_until : pos + 1;
            EqWalk w{lim, 0u, 0u, 0u, 0u};
            ++in_order;
            if (lane < 4) w = eq_repair_walk<FC>(d, r, plan, rc, inst, pos, lim, force_until, E, lane, p3g);
            w.j_end = (uint32_t)__shfl((int)w.j_end, 0, 64);
            repaired += (unsigned long long)(uint32_t)__shfl((int)w.repaired, 0, 64);
            settled += (unsigned long long)(uint32_t)__shfl((int)w.settled, 0, 64); steps += (unsigned long long)(uint32_t)__shfl((int)w.steps, 0, 64);
            fills += (unsigned long long)(uint32_t)__shfl((int)w.fills, 0, 64);
#pragma unroll
            for (int k = 0; k < 8; ++k) E[k] = shfl_f64(E[k], 0);
            pos = w.j_end;
            have_E = pos >= NC || !same8(E, rc[pos - 1].end);   // at the end of the stream E stays the carried state
            if (pos >= NC) { have_E = true; break; }
            continue;
        }
        // the next (up to EQ_ISLANDS) heads at or after pos
        uint32_t n_heads = 0, scan_end = NC;
        for (uint32_t j0 = pos; j0 < NC; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const bool head = j < NC && fails(j) && (j == pos || !fails(j - 1));
            uint64_t m = __ballot(head);
            while (m && n_heads < (uint32_t)EQ_ISLANDS) { const int b = __builtin_ctzll(m); if (lane == 0) heads[n_heads] = j0 + (uint32_t)b; ++n_heads; m &= m - 1; }
            if (m) { scan_end = j0 + (uint32_t)__builtin_ctzll(m); break; }   // one head too many: this round ends in front of it
        }
        if (n_heads == 0) { pos = NC; break; }     // every boundary from pos on matches: proven to the end
        if (lane == 0) heads[n_heads] = scan_end;
        __syncthreads();
        // one group per island, all at once, each from its predecessor chunk's recorded end, each up to the next head
        double El[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) El[k] = 0.0;
        EqWalk w{0u, 0u, 0u, 0u, 0u};
        ++rounds;
        bool in_sync = true;
        if ((uint32_t)grp < n_heads) {
            const uint32_t h = heads[grp], lim = heads[grp + 1];
#pragma unroll
            for (int k = 0; k < 8; ++k) El[k] = rc[h - 1].end[k];
            w = eq_repair_walk<FC>(d, r, plan, rc, inst, h, lim, 0u, El, lane, p3g);
            in_sync = w.j_end == lim && same8(El, rc[lim - 1].end) && !(plan.pad & 2u);
        }
        // the assumptions, in stream order
        uint32_t i = 0;
        bool apart = false;
        for (; i < n_heads; ++i) {
            repaired += (unsigned long long)(uint32_t)__shfl((int)w.repaired, (int)(4 * i), 64);
            settled += (unsigned long long)(uint32_t)__shfl((int)w.settled, (int)(4 * i), 64); steps += (unsigned long long)(uint32_t)__shfl((int)w.steps, (int)(4 * i), 64);
            fills += (unsigned long long)(uint32_t)__shfl((int)w.fills, (int)(4 * i), 64);
            if (!__shfl((int)in_sync, (int)(4 * i), 64)) { apart = true; break; }
        }
        __syncthreads();                           // heads[] is rewritten by the next round
        if (!apart) { pos = scan_end; continue; }
        // island i ended apart, at chunk j_end with the true state in its group: go on from there in order
        pos = (uint32_t)__shfl((int)w.j_end, (int)(4 * i), 64);
#pragma unroll
        for (int k = 0; k < 8; ++k) E[k] = shfl_f64(El[k], (int)(4 * i));
        have_E = true;
        force_until = i + 1 < n_heads ? scan_end : pos;   // the islands behind it in this round started from an assumption that did not hold
    }
    if (lane != 0) return;
    // the carried state (eq_three.rs:17-22): poles after the last sample, delay line = the last three inputs
    EqState st = states[inst];
    if (!have_E) {
#pragma unroll
        for (int k = 0; k < 8; ++k) E[k] = rc[NC - 1].end[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { st.lo[k] = E[k]; st.hi[k] = E[4 + k]; }
    size_t F = r.frames;
    if (tail) {   // the one to three samples behind the last whole piece (launcher), from the exact state, through the same epilogue
        EqBlkEmit<FC> em; em.init(d, r, inst); em.seek(F);
        double lo[4] = {E[0], E[1], E[2], E[3]}, hi[4] = {E[4], E[5], E[6], E[7]};
        for (uint32_t k = 0; k < tail; ++k) {
            const double x = (double)d.in[F + k];
            pump<FC>(r.lo_f, lo, x); pump<FC>(r.hi_f, hi, x);
            em.emit(F + k, eq_out_of<FC>(lo[3], hi[3], (double)d.in[F + k - 3], d.gain_lo, d.gain_mid, d.gain_hi));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { st.lo[k] = lo[k]; st.hi[k] = hi[k]; }
        F += tail;
    }
    st.history[0] = (double)d.in[F - 3]; st.history[1] = (double)d.in[F - 2]; st.history[2] = (double)d.in[F - 1];   // F >= 2 warm-ups >= 3 samples
    states[inst] = st;
    if (stats) {   // [0] chunks run, [1] not proven by their recorded start, [2] of those settled by the O(1) comparison, [3] walk steps (EQ_RB samples), [4] fill steps,
                   // [5] rounds of islands, [6] in-order walks after an island that ended apart, [7] streams finished by the NaN fill
        atomicAdd(&stats[0], (unsigned long long)NC);
        if (repaired) {
            atomicAdd(&stats[1], repaired); atomicAdd(&stats[2], settled); atomicAdd(&stats[3], steps); atomicAdd(&stats[4], fills);
            atomicAdd(&stats[5], rounds); atomicAdd(&stats[6], in_order); atomicAdd(&stats[7], nan_fills);
        }
    }
}

// warm-up length: the 4-pole cascade's response to a unit difference k samples back is at most C(k+3,3) p^k (p = 1 - f); W is
// the first multiple of 128 where that is below 2^-72 -- twenty bits under the f64 ulp of a full-scale state, so that two
// trajectories are within rounding of each other and coalesce (measured: 0 mismatching boundaries from W = 1024 at 48 kHz
// on every live signal tried, all of them at W = 768).  A longer W only costs time; a shorter one only costs repairs.
static size_t eq_warm_len(double f) {
    const long double p = 1.0L - (long double)f;
    if (!(p > 0.0L) || !(p < 1.0L)) return 128;
    const long double lim = ldexpl(1.0L, -72);
    for (size_t K = 128; K <= ((size_t)1 << 22); K += 128) {
        const long double c = (long double)(K + 3) * (long double)(K + 2) * (long double)(K + 1) / 6.0L;
        if (c * expl((long double)K * logl(p)) < lim) return K;
    }
    return (size_t)-1;
}

bool eq_plan_spec(uint32_t n, size_t frames, size_t fpc, double lo_f, double hi_f, EqSpecPlan& plan, bool whole_ticks, bool two_tiles) {
    plan = EqSpecPlan{1u, 0u, 0u, 0u, 0u};
    if (!n || !frames) return false;
    size_t W = std::max(eq_warm_len(lo_f), eq_warm_len(hi_f));
    size_t W_hi = eq_warm_len(hi_f) == (size_t)-1 ? W : (eq_warm_len(hi_f) + 31) / 32 * 32;   // the high cascade's own forgetting length (whole super-blocks)
    const int force_w = env_int("MX_EQ_SPEC_WARM", 0);       // tests: a short warm-up makes every boundary fail and the repair pass do all the work
    if (force_w > 0) { W = ((size_t)force_w + 15) / 16 * 16; W_hi = W; }
    if (env_int("MX_EQ_SPEC_WARM_HI_FULL", 0)) W_hi = W;      // A/B: both cascades over the whole window (rounds 2 - 3)
    if (W == (size_t)-1) return false;
    const int force_c = env_int("MX_EQ_SPEC_CHUNKS", 0);     // tuning / tests: chunks per instance (1 = never speculate)
    if (force_c == 1) return false;
    if (frames < 2 * W || frames < 64) return false;          // a stream shorter than two warm-ups: one lane per instance
    // Chunk lengths are whole ticks when that is a multiple of 16 samples (an inline Envelope's state is read once per tick and every
    // lane of a wave crosses its tick boundaries at the same step), else multiples of 32 samples.
    // With an inline Envelope (whole_ticks) chunks are whole ticks at ANY rate, in multiples that keep them multiples of 4 samples: 735 -> 2 940 (44.1 kHz), 800 -> 800.
    size_t unit = (fpc && fpc % 16 == 0 && fpc <= frames / 2) ? fpc : 32;
    if (!whole_ticks && unit % 32) unit = 32;                 // (ticks of 16 (mod 32) samples: only the inline Envelope needs whole ticks, everybody else whole lines)
    if (whole_ticks && fpc >= 32 && fpc % 16 != 0) {
        size_t u = fpc; while (u % 4) u += fpc;
        if (u <= frames / 2) unit = u;
    }
    auto chunk_of = [&](size_t nc) { return ((frames + nc - 1) / nc + unit - 1) / unit * unit; };
    // Chunks may be SHORTER than the warm-up (a chunk whose window would reach the stream's start warms up from there, from the carried
    // state): short submissions get a wave of 64 chunks per strip where C >= W allowed 40.  Not below 256 samples.
    const size_t c_min = std::max<size_t>(unit, 256);
    const size_t nc_max = frames / c_min;
    size_t best;
    if (force_c > 1) best = std::min<size_t>((size_t)force_c, nc_max);
    else {
        // Cost of a plan with nc chunks per instance: the samples every lane walks, warm-ups included, over how full the chip is.  A wave
        // is 64 chunks of one instance; every tiled variant of the kernel now fits FOUR waves per SIMD (the Envelope variant went from 141
        // to 109 VGPRs when its loop was split per phase) and the fourth pays: 1024 strips x 2048 ticks, 192 chunks (3 waves per SIMD)
        // 5.83 ms, 256 chunks (4 waves) 4.55 ms.  Waves beyond one round of 4 per SIMD wait for a second round.  Few instances => more,
        // shorter chunks, down to one warm-up (the rank of an 8-GPU job).  Below one wave per SIMD a wave runs at its own pace whether 1
        // or 64 of its lanes work, so the chunk count only shortens it: as many lanes of the one wave as fit.
        // MEASURED table (round 5, tools/q_grid3.sh on MI355X after the workgroups became four waves: 1024 strips x 64 .. 2048 ticks and 128 .. 1024 strips x 2048 ticks,
        // every whole-tick chunk length): the time of a launch is what a SIMD needs for its w waves, each walking its chunk and its warm-up, and a wave's pace depends on
        // how many share the SIMD:
        //   per chunk sample    0.20 / 0.16 / 0.153 / 0.154 us per wave at w = 1 / 2 / 3 / 4 (a SIMD turns out 5.0 / 6.25 / 6.5 / 6.5 samples per us: one wave alone is
        //                       latency-bound at 77 % of what four reach, two are at 96 %)
        //   per warm-up         120 / 140 / 120 / 95 us per wave (1 280 samples at 48 kHz)
        // The model this replaces charged occupancy linearly (half the waves = twice the time) and took one-tick chunks at four waves per SIMD wherever it could:
        // 1024 strips x 256 ticks 0.95 ms (two-tick chunks: 0.80), x 512 ticks 1.41 (1.33), 256 strips x 2048 ticks 1.48 (1.40), 128 x 2048 0.99 (0.83-0.85).
        auto cost = [&](size_t nc) {
            const double waves = (double)n * (double)((nc + 63) / 64);
            const double Cn = (double)chunk_of(nc), wscale = (double)W / 1280.0;
            if (two_tiles) {   // a control tile beside the input's (16 KiB of LDS per wave: ten waves per CU): the round-4 model, measured on that shape
                const double resident = 2560.0, rounds = std::ceil(waves / resident), occ = waves / (rounds * resident);
                return ((double)nc * (Cn + (double)W)) / occ * (1.0 + 0.15 * (rounds - 1.0));
            }
            static const double m_us[5] = {0.0, 0.20, 0.16, 0.153, 0.1542}, warm_us[5] = {0.0, 120.0, 140.0, 120.0, 95.0};
            const double per_simd = waves / 1024.0;
            if (per_simd <= 4.0) {
                const int w = std::max(1, (int)std::ceil(per_simd - 1e-9));
                return (double)w * (Cn * m_us[w] + warm_us[w] * wscale);
            }
            return per_simd * (Cn * m_us[4] + warm_us[4] * wscale) * 0.94;   // further rounds of four start as slots free up: a little better than whole rounds (measured 0.94)
        };
        best = std::min<size_t>(64, nc_max);
        double best_cost = cost(best);
        for (size_t nc = 128; nc <= nc_max && nc <= 8192; nc += 64) {
            const double c = cost(nc);
            if (c < best_cost * 0.98) { best_cost = c; best = nc; }
        }
        // one lane per instance (no speculation) costs `frames` dependent steps of ~110 cycles, 64 instances per wave; a chunk lane
        // (C + W) steps of ~260 cycles when its wave has a SIMD to itself, ~190 per resident wave when the pipes are shared
        const double seq = (double)frames * 110.0 * std::ceil((double)((n + 63) / 64) / 1024.0);
        const double per_simd = std::max(1.0, (double)n * (double)((best + 63) / 64) / 1024.0);
        // (260: 1024 strips x 16 ticks, one wave per SIMD walking 800 + 1 280 samples, 1 024 of them low cascade only: 0.244 ms; it was 460 before the warm-up split)
        const double spec = (double)(chunk_of(best) + W) * (per_simd <= 1.0 ? 260.0 : 190.0 * per_simd);
        if (best < 2 || spec >= seq) return false;
    }
    if (best < 2) return false;
    size_t C = chunk_of(best);
    if (C < c_min) C = (c_min + unit - 1) / unit * unit;
    plan.chunk = (uint32_t)C; plan.warm = (uint32_t)W; plan.warm_hi = (uint32_t)std::min(W, W_hi);
    plan.n_chunks = (uint32_t)((frames + C - 1) / C);
    return plan.n_chunks >= 2;
}

__global__ __launch_bounds__(64) void k_tail_gate(const uint32_t* flag, uint32_t seq, uint32_t limit_ticks) {
    if (threadIdx.x) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();                            // the constant 100 MHz counter (s_memtime runs with the shader clock)
    for (;;) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(v - seq) >= 0) break;
        if ((uint64_t)__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) break;  // an ordering hint, never a dependency: bounded
        __builtin_amdgcn_s_sleep(8);
    }
}
void launch_tail_gate(const uint32_t* flag, uint32_t seq, uint32_t limit_us, hipStream_t s) {
    hipLaunchKernelGGL(k_tail_gate, dim3(1), dim3(64), 0, s, flag, seq, limit_us * 100u);   // s_memtime: 100 MHz
}

size_t eq_spec_scratch_bytes(uint32_t n, const EqSpecPlan& plan) { return (size_t)n * plan.n_chunks * sizeof(EqChunkRec); }

int eq_epilogue_mode(uint32_t epi, uint32_t flags, bool has_ctl) {
    const int mode = epi != 2u ? EQM_PLAIN : ((flags & MX_EQF_ENV) ? EQM_AMP_ENV : (has_ctl ? EQM_AMP_CTL : EQM_AMP_CONST));
    const int stereo = !(epi == 0u || (flags & MX_EQF_MONO_DUP));
    return mode * 2 + stereo;
}

bool launch_eq_three_spec(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r_in, const EqSpecPlan& plan_in, int uniform_mode, void* scratch, uint64_t* stats, hipStream_t s) {
    if (!n || !r_in.frames) return false;
    // A stream that is not whole pieces of four samples (735 t frames at 44.1 kHz with t not a multiple of 4): the tiled kernel runs the frames up to the last
    // multiple of four, and the proof / repair kernel -- which ends up holding the exact state there -- walks the one to three samples left (`tail`).
    EqRun r = r_in; EqSpecPlan plan = plan_in;
    uint32_t tail = (uint32_t)(r_in.frames & 3u);
    if (tail) {
        r.frames = r_in.frames - tail;
        plan.n_chunks = (uint32_t)((r.frames + plan.chunk - 1) / plan.chunk);
        if (plan.n_chunks < 2) { r = r_in; plan = plan_in; tail = 0; }
    }
    uint32_t wpi = (plan.n_chunks + 63) / 64;
    EqChunkRec* recs = (EqChunkRec*)scratch;
    // MX_EQ_REPAIR_TEST (tests only; never changes a result bit, only which path produces it): 1 = a standing pair of trajectories is treated as if its
    // outputs differed (the comparison that settles a constant chunk is skipped, the rest of the chunk is FILLED from the true state); 2 = every island is
    // treated as having ended apart from the speculative run (the in-order fallback that rewrites everything behind it).  Both paths are otherwise reached
    // only by inputs whose f32 outputs differ between two states a few f64 ulps apart.
    EqSpecPlan plan_t = plan; plan_t.pad = (uint32_t)env_int("MX_EQ_REPAIR_TEST", 0);
    const int no_tiles = env_int("MX_EQ_SPEC_DIRECT", 0);   // A/B: the direct (16 bytes per lane) form everywhere (read per launch: tests switch it inside one process)
    const int um = uniform_mode;
    // 321: whole 128-byte lines, ONE tile of 32 samples per row (8 KiB of LDS per wave); 16: half lines, two tiles (round 3's default:
    // 1.44x the source bytes on the fabric, tools/fetch_probe.hip); 32: whole lines, two tiles (16 KiB: ten waves per CU).  Measured, 1024 strips x
    // 2048 ticks: 4.46 / 4.56 ms exact, 3.85 / 4.09 ms contracted (321 / 16).  A/B knob; the contracted order is compiled for 321 and 16.
    // Unset: ONE tile while three or four waves share a SIMD (they hide each other's tile round trips and a second tile would cost a wave), TWO where at most two do
    // (short submissions: a wave alone on its SIMD waits out every round trip itself; 1024 strips x 64 ticks 0.294 -> 0.276 ms, x 128 ticks 0.470 -> 0.458 ms).
    const int sb_env = env_int("MX_EQ_SPEC_SB", 0);
    // (round 5, four-wave workgroups, tools/q_fc2.sh: two tiles pay only while a wave has its SIMD to itself -- 1024 strips x 64 / 128 ticks exact 0.294 / 0.458 ms against
    // 0.311 / 0.476 with one tile; from two waves per SIMD on one whole-line tile is the faster form: 1024 x 2048 in 128 chunks 4.28 against 4.36 ms exact, 3.85 against 4.14
    // contracted -- the contracted order's two-tile form is the half-line one and pays its 1.44x source bytes there)
    const int sb_auto = (size_t)n * wpi <= 1024 ? (r.fc ? 16 : 32) : 321;
    const int sb = sb_env == 0 ? sb_auto : (!r.fc && sb_env == 32) ? 32 : (sb_env == 16 ? 16 : 321);   // samples per lane per super-block
    // ragged ticks (RT instantiations): an inline Envelope at a rate whose tick is not whole super-blocks (44.1 kHz: 735) -- chunks of whole ticks, multiples of 4 samples
    // (ticks of at least 64 samples: the RT kernel's boundary walk looks at the current and the next tick only -- with a super-block of 32 samples and row shifts up to 28
    // a shorter tick could end twice inside one block; shorter ragged ticks take the direct form)
    const bool rt = !no_tiles && (um == 6 || um == 7) && r.fpc >= 64 && r.fpc % 32 != 0 && plan.chunk % r.fpc == 0 && plan.chunk % 4 == 0 && r.frames % 4 == 0 &&
                    plan.warm % 32 == 0 && r.frames < (1ull << 30) && r.frames >= 4;
    // a control BUFFER (um 4 / 5: an Amplifier modulated by another module's output -- an LFO, an Envelope that is not folded in): the control of a super-block travels
    // through a second tile beside the input's (one tile each, 16 KiB per wave: ten waves per CU); the direct form it had until round 4: 33 ms per step where this takes 6
    const bool ctl_tiled = env_int("MX_EQ_CTL_DIRECT", 0) == 0;
    const bool tiled = rt || (!no_tiles && um >= 0 && (ctl_tiled || (um != 4 && um != 5)) && r.frames % 4 == 0 && plan.chunk % 32 == 0 && plan.warm % 32 == 0 &&
                       r.frames < (1ull << 30) && r.frames >= 4 && ((um != 6 && um != 7) || (r.fpc % 32 == 0 && plan.chunk % r.fpc == 0)));
    if (!tiled && tail) { r = r_in; plan = plan_in; tail = 0; wpi = (plan.n_chunks + 63) / 64; }   // the direct form takes any length itself
    if (tiled) {
        static const int lds_pad = env_int("MX_EQ_SPEC_LDS", 0);   // A/B: bytes of LDS requested per wave (occupancy shaping)
        const uint32_t nwv = n * wpi, nwg = (nwv + EQ_WPB - 1) / EQ_WPB;   // waves, and workgroups of EQ_WPB waves
        const size_t lds = std::max<size_t>((sb == 321 ? 1 : 2) * 64 * (size_t)(sb == 321 ? 32 : sb) * sizeof(float), (size_t)lds_pad);
        // the contracted order (MX_FLAG_FP_CONTRACT) is compiled for the 16-sample super-block only (the faster of the two)
#define MX_GT(M, S) { if (sb == 321 && r.fc) hipLaunchKernelGGL((k_eq_three_spec_tiled<32, M, S, true, 1>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds, s, d, (const EqState*)st, r, plan, wpi, recs, nwv); \
                      else if (sb == 321) hipLaunchKernelGGL((k_eq_three_spec_tiled<32, M, S, false, 1>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds, s, d, (const EqState*)st, r, plan, wpi, recs, nwv); \
                      else if (r.fc) hipLaunchKernelGGL((k_eq_three_spec_tiled<16, M, S, true>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds, s, d, (const EqState*)st, r, plan, wpi, recs, nwv); \
                      else if (sb == 32) hipLaunchKernelGGL((k_eq_three_spec_tiled<32, M, S, false>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds, s, d, (const EqState*)st, r, plan, wpi, recs, nwv); \
                      else hipLaunchKernelGGL((k_eq_three_spec_tiled<16, M, S, false>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds, s, d, (const EqState*)st, r, plan, wpi, recs, nwv); }
        if (um == 4 || um == 5) {
            const size_t lds2 = (size_t)2 * 64 * 32 * sizeof(float);
#define MX_GCT(S, F) hipLaunchKernelGGL((k_eq_three_spec_tiled<32, EQM_AMP_CTL, S, F, 1>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds2, s, d, (const EqState*)st, r, plan, wpi, recs, nwv)
            if (um == 4) { if (r.fc) MX_GCT(0, true); else MX_GCT(0, false); } else { if (r.fc) MX_GCT(1, true); else MX_GCT(1, false); }
#undef MX_GCT
        } else
        if (rt) {   // one tile of whole 16-byte rows (the rows of such chunks are not line-aligned either way)
            const size_t lds1 = (size_t)64 * 32 * sizeof(float);
#define MX_GRT(S, F) hipLaunchKernelGGL((k_eq_three_spec_tiled<32, EQM_AMP_ENV, S, F, 1, true>), dim3(nwg), dim3(64 * EQ_WPB), EQ_WPB * lds1, s, d, (const EqState*)st, r, plan, wpi, recs, nwv)
            if (um == 6) { if (r.fc) MX_GRT(0, true); else MX_GRT(0, false); } else { if (r.fc) MX_GRT(1, true); else MX_GRT(1, false); }
#undef MX_GRT
        } else
        switch (um) {
        case 0: MX_GT(EQM_PLAIN, 0); break;     case 1: MX_GT(EQM_PLAIN, 1); break;
        case 2: MX_GT(EQM_AMP_CONST, 0); break; case 3: MX_GT(EQM_AMP_CONST, 1); break;
        case 6: MX_GT(EQM_AMP_ENV, 0); break;   default: MX_GT(EQM_AMP_ENV, 1); break;
        }
#undef MX_GT
    } else {
#define MX_GO(M, S) { if (r.fc) hipLaunchKernelGGL((k_eq_three_spec<M, S, true>), dim3(n * wpi), dim3(64), 0, s, d, (const EqState*)st, r, plan, wpi, recs); \
                      else hipLaunchKernelGGL((k_eq_three_spec<M, S, false>), dim3(n * wpi), dim3(64), 0, s, d, (const EqState*)st, r, plan, wpi, recs); }
        switch (uniform_mode) {
        case 0: MX_GO(EQM_PLAIN, 0); break;     case 1: MX_GO(EQM_PLAIN, 1); break;
        case 2: MX_GO(EQM_AMP_CONST, 0); break; case 3: MX_GO(EQM_AMP_CONST, 1); break;
        case 4: MX_GO(EQM_AMP_CTL, 0); break;   case 5: MX_GO(EQM_AMP_CTL, 1); break;
        case 6: MX_GO(EQM_AMP_ENV, 0); break;   case 7: MX_GO(EQM_AMP_ENV, 1); break;
        default: MX_GO(-1, -1); break;
        }
#undef MX_GO
    }
    // the proof (and, where a boundary fails, the repair) runs the same order the chunks ran
    plan_t.n_chunks = plan.n_chunks;
    if (r.fc) hipLaunchKernelGGL(k_eq_three_repair<true>, dim3(n), dim3(64), 0, s, d, st, r, plan_t, (const EqChunkRec*)recs, (unsigned long long*)stats, tail);
    else hipLaunchKernelGGL(k_eq_three_repair<false>, dim3(n), dim3(64), 0, s, d, st, r, plan_t, (const EqChunkRec*)recs, (unsigned long long*)stats, tail);
    return tiled;   // only the tiled kernel stores r.started (the flag k_tail_gate waits for)
}

}  // namespace mx
