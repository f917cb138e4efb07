// mx_k_eq_common.hpp -- what the EqThree kernels share (internal, device only): the exact pole update, the fused epilogue
// (StereoPanner / Amplifier / inline Envelope) and the sequential walker the exact-order kernels use.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written, never fused; the exact
// recurrence must not become v_fma_f64.
#pragma once
#include "mx_dev.hpp"
#include "mx_env_math.hpp"

namespace mx {

#define MX_VSA (1.0 / 4294967295.0)   /* eq_three.rs:11 */

// LowPass::pump, eq_three.rs:117-124 -- exact order
__device__ __forceinline__ double pump(const double f, double (&p)[4], const double sample) {
    p[0] += f * (sample - p[0]) + MX_VSA;
    p[1] += f * (p[0] - p[1]);
    p[2] += f * (p[1] - p[2]);
    p[3] += f * (p[2] - p[3]);
    return p[3];
}

// one EqThree::run_tick sample (eq_three.rs:66-88): the f32 the EQ stores
struct EqPoles { double lo[4], hi[4]; double h0, h1, h2; };
__device__ __forceinline__ float eq_step(EqPoles& s, const double lo_f, const double hi_f, const double g_lo, const double g_mid, const double g_hi, const float x) {
    const double sample = (double)x;
    const double l = pump(lo_f, s.lo, sample);
    const double h = s.h0 - pump(hi_f, s.hi, sample);
    const double mid = s.h0 - (h + l);
    s.h0 = s.h1; s.h1 = s.h2; s.h2 = sample;
    return (float)(l * g_lo + mid * g_mid + h * g_hi);
}

// the same sample with the delay-line value handed in (h0 = the input three samples back, eq_three.rs:66,80-83): callers that walk
// four samples at a time keep the delay line as "the last three inputs" and never shift registers
__device__ __forceinline__ float eq_step_h(EqPoles& s, const double lo_f, const double hi_f, const double g_lo, const double g_mid, const double g_hi,
                                           const double sample, const double h0) {
    const double l = pump(lo_f, s.lo, sample);
    const double h = h0 - pump(hi_f, s.hi, sample);
    const double mid = h0 - (h + l);
    return (float)(l * g_lo + mid * g_mid + h * g_hi);
}

// Fused epilogue (see EqDesc): what StereoPanner (stereo_panner.rs:35-38), Amplifier (amplifier.rs:52-57,71-73) and -- with
// MX_EQF_ENV -- the Envelope feeding its control (envelope.rs:34-58,117) would do to the f32 sample y the EQ just produced.
struct EqEpi {
    float* out; const float* ctl; double amp_one_minus, amp_mod_depth, amp_amplitude;
    EnvParams env; const EnvTick* ticks /* this instance's row of the per-tick table */; uint32_t epi, flags;
};
__device__ __forceinline__ EqEpi eq_epi_of(const EqDesc& d, const EnvTick* ticks_row) {
    EqEpi e;
    e.out = d.out; e.ctl = d.ctl; e.amp_one_minus = d.amp_one_minus; e.amp_mod_depth = d.amp_mod_depth; e.amp_amplitude = d.amp_amplitude;
    e.env = d.env; e.ticks = ticks_row; e.epi = d.epi; e.flags = d.flags;
    return e;
}
__device__ __forceinline__ bool eq_mono_out(const EqEpi& e) { return e.epi == 0u || (e.flags & MX_EQF_MONO_DUP); }

// Amplifier depth() for the Envelope sample at absolute time t (amplifier.rs:71-73 over envelope.rs:117's f32)
__device__ __forceinline__ double env_depth(const EnvParams& p, const EnvTick& k, double one_minus, double mod_depth, uint64_t t, double sr, double rsr) {
    if (k.flat) return k.depth;
    const float cc = (float)env_amplitude(p, k.tag, k.seq, k.off_amp, t, sr, rsr);   // Envelope stores f32 (envelope.rs:117)
    return one_minus + mod_depth * (double)cc;
}
__device__ __forceinline__ float amp_apply(float y, double depth, double amplitude) { return (float)((double)y * depth * amplitude); }   // amplifier.rs:56

// One instance walked sample by sample in time order (the exact kernel, the repair pass): epilogue with a running tick cursor.
struct EqSeqEmit {
    EqEpi E; double sr, rsr; uint64_t t0; size_t fpc;
    size_t left = 0; uint32_t call = 0; EnvTick cur{};
    __device__ __forceinline__ void seek(size_t i) {   // position the cursor on sample i of the run
        if (!(E.flags & MX_EQF_ENV) || E.epi != 2u) return;
        call = (uint32_t)(i / fpc); left = fpc - i % fpc; cur = E.ticks[call];
    }
    // sample i (cursor must be there): store what the folded modules make of y
    __device__ __forceinline__ void emit(size_t i, float y) {
        float v = y;
        if (E.epi == 2u) {
            double depth;
            if (E.flags & MX_EQF_ENV) {
                if (left == 0) { ++call; cur = E.ticks[call]; left = fpc; }
                --left;
                depth = env_depth(E.env, cur, E.amp_one_minus, E.amp_mod_depth, t0 + i, sr, rsr);
            } else {
                const double m = E.ctl ? (double)E.ctl[i] : 1.0;                       // amplifier.rs:54
                depth = E.amp_one_minus + E.amp_mod_depth * m;                        // amplifier.rs:71-73
            }
            v = amp_apply(y, depth, E.amp_amplitude);
        }
        if (eq_mono_out(E)) E.out[i] = v;                                             // one float per frame
        else reinterpret_cast<float2*>(E.out)[i] = make_float2(v, v);                 // stereo_panner.rs:35-38
    }
};

}  // namespace mx
