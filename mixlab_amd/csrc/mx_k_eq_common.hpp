// mx_k_eq_common.hpp -- what the EqThree kernels share (internal, device only): the exact pole update, the fused epilogue
// (StereoPanner / Amplifier / inline Envelope) and the sequential walker the exact-order kernels use.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written, never fused; the exact
// recurrence must not become v_fma_f64.  The template parameter FC (MX_FLAG_FP_CONTRACT) selects the CONTRACTED order instead:
// the same source expressions with each multiply fused into the add that consumes it (mul_add<FC>, mx_env_math.hpp) -- 26 f64
// instructions per EqThree sample instead of 36, every f32 output within 1 ULP of the exact order's (tests: equal to the oracle's
// contract mode bit for bit, and within 1 ULP of its exact mode with the differing samples counted).
#pragma once
#include "mx_dev.hpp"
#include "mx_env_math.hpp"

namespace mx {

#define MX_VSA (1.0 / 4294967295.0)   /* eq_three.rs:11 */

// LowPass::pump, eq_three.rs:117-124 -- exact order (FC: p0 += fma(f, x - p0, VSA); p_k = fma(f, p_{k-1} - p_k, p_k))
template <bool FC = false>
__device__ __forceinline__ double pump(const double f, double (&p)[4], const double sample) {
    p[0] += mul_add<FC>(f, sample - p[0], MX_VSA);
    p[1] = mul_add<FC>(f, p[0] - p[1], p[1]);
    p[2] = mul_add<FC>(f, p[1] - p[2], p[2]);
    p[3] = mul_add<FC>(f, p[2] - p[3], p[3]);
    return p[3];
}
// the band mix, eq_three.rs:76-88: (lo * gain_lo + mid * gain_mid) + hi * gain_hi  (FC: fma(hi, g_hi, fma(mid, g_mid, lo * g_lo)))
template <bool FC = false>
__device__ __forceinline__ float band_mix(double l, double mid, double h, double g_lo, double g_mid, double g_hi) {
    return (float)mul_add<FC>(h, g_hi, mul_add<FC>(mid, g_mid, l * g_lo));
}

// one EqThree::run_tick sample (eq_three.rs:66-88): the f32 the EQ stores
struct EqPoles { double lo[4], hi[4]; double h0, h1, h2; };
template <bool FC = false>
__device__ __forceinline__ float eq_step(EqPoles& s, const double lo_f, const double hi_f, const double g_lo, const double g_mid, const double g_hi, const float x) {
    const double sample = (double)x;
    const double l = pump<FC>(lo_f, s.lo, sample);
    const double h = s.h0 - pump<FC>(hi_f, s.hi, sample);
    const double mid = s.h0 - (h + l);
    s.h0 = s.h1; s.h1 = s.h2; s.h2 = sample;
    return band_mix<FC>(l, mid, h, g_lo, g_mid, g_hi);
}

// the same sample with the delay-line value handed in (h0 = the input three samples back, eq_three.rs:66,80-83): callers that walk
// four samples at a time keep the delay line as "the last three inputs" and never shift registers
template <bool FC = false>
__device__ __forceinline__ float eq_step_h(EqPoles& s, const double lo_f, const double hi_f, const double g_lo, const double g_mid, const double g_hi,
                                           const double sample, const double h0) {
    const double l = pump<FC>(lo_f, s.lo, sample);
    const double h = h0 - pump<FC>(hi_f, s.hi, sample);
    const double mid = h0 - (h + l);
    return band_mix<FC>(l, mid, h, g_lo, g_mid, g_hi);
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

// Amplifier depth(), amplifier.rs:71-73: (1.0 - mod_depth) + mod_depth * mod_value, `one_minus` from the host
template <bool FC = false>
__device__ __forceinline__ double amp_depth(double one_minus, double mod_depth, double mod_value) { return mul_add<FC>(mod_depth, mod_value, one_minus); }
// ... for the Envelope sample at absolute time t (over envelope.rs:117's f32)
template <bool FC = false>
__device__ __forceinline__ double env_depth(const EnvParams& p, const EnvTick& k, double one_minus, double mod_depth, uint64_t t, double sr, double rsr) {
    if (k.flat) return k.depth;
    const float cc = (float)env_amplitude<FC>(p, k.tag, k.seq, k.off_amp, t, sr, rsr);   // Envelope stores f32 (envelope.rs:117)
    return amp_depth<FC>(one_minus, mod_depth, (double)cc);
}
__device__ __forceinline__ float amp_apply(float y, double depth, double amplitude) { return (float)((double)y * depth * amplitude); }   // amplifier.rs:56

// One instance walked sample by sample in time order (the exact kernel, the repair pass): epilogue with a running tick cursor.
template <bool FC = false>
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
                depth = env_depth<FC>(E.env, cur, E.amp_one_minus, E.amp_mod_depth, t0 + i, sr, rsr);
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

}  // namespace mx
