// mx_env_math.hpp -- the Envelope's closed-form amplitude (reference src/module/envelope.rs:16-58),
// shared by the envelope kernel and the EqThree kernel's fused epilogue (internal, device only).
#pragma once
#include "mx_kernels.hpp"

namespace mx {

// (last - first) as f64 / SAMPLE_RATE * 1000.0 (envelope.rs:16-18) with the IEEE quotient obtained
// by Markstein's correction instead of the ~20-instruction division expansion: q = a*y,
// r = fma(-q,b,a), q' = fma(r,y,q) with y = RN(1/b) from the host.  tests/test_fastdiv.py checks
// q' == a/b bit-for-bit for every a in [0, 2^32) at 44.1 and 48 kHz; larger spans (> 24 h) take
// the true division.
__device__ __forceinline__ double ms_of_u32(uint32_t dt, double sr, double rsr) {
    const double a = (double)dt;
    double q = a * rsr;
    const double r = fma(-q, sr, a);
    q = fma(r, rsr, q);
    return q * 1000.0;
}
__device__ __forceinline__ double seq_ms(uint64_t first, uint64_t last, double sr, double rsr) {
    const uint64_t dt = last - first;
    if (dt >> 32) return (double)dt / sr * 1000.0;
    return ms_of_u32((uint32_t)dt, sr, rsr);
}
__device__ __forceinline__ double clamp01(double x) { return x > 1.0 ? 1.0 : (x < 0.0 ? 0.0 : x); }  // envelope.rs:20-28

// FC (MX_FLAG_FP_CONTRACT, mixlab_gpu.h): the same expressions with every multiply whose only consumer is an add fused into it
// -- what `-ffp-contract=fast` would make of the reference's source; one rounding instead of two, results within 1 ULP of the
// f32 the exact order stores.  The fusions are spelled out (never left to the compiler) and restated by the oracle's contract
// mode, so the contracted order is as reproducible as the exact one.  mul_add<FC>(a, b, c) = a * b + c.
template <bool FC> __device__ __forceinline__ double mul_add(double a, double b, double c) {
    if constexpr (FC) return __builtin_fma(a, b, c); else return a * b + c;
}
template <bool FC = false>
__device__ __forceinline__ double amp_on_ms(const EnvParams& p, double ms) {                          // envelope.rs:37-49
    const double attack = p.inv_attack * ms;
    const double since_decay = ms - p.attack_ms;
    const double decay_amplitude = 1.0 - clamp01(p.inv_decay * since_decay);
    const double decay = mul_add<FC>(p.one_minus_sustain, decay_amplitude, p.sustain);
    return ms < p.attack_ms ? attack : decay;
}
__device__ __forceinline__ double amp_off_ms(const EnvParams& p, double off_amp, double ms) {         // envelope.rs:51-56
    const double release_amplitude = 1.0 - clamp01(p.inv_release * ms);
    return off_amp * release_amplitude;
}

// One Envelope::run_tick step of the state machine for a CONSTANT gate over a whole run starting at
// t0 (envelope.rs:99-115): only the first sample can change the state.
template <bool FC = false>
__device__ __forceinline__ void env_const_gate_step(const EnvParams& p, float gate, uint64_t t0, double sr, double rsr,
                                                    uint32_t& tag, uint64_t& seq, double& off_amp) {
    if (tag != 1u) { if (gate == 1.0f) { tag = 1u; seq = t0; } }
    else if (gate == 0.0f) { off_amp = amp_on_ms<FC>(p, seq_ms(seq, t0, sr, rsr)); tag = 2u; seq = t0; }
}
// amplitude at absolute sample time t for a state that no longer changes during the run
template <bool FC = false>
__device__ __forceinline__ double env_amplitude(const EnvParams& p, uint32_t tag, uint64_t seq, double off_amp,
                                                uint64_t t, double sr, double rsr) {
    if (tag == 0u) return 0.0;                                   // envelope.rs:36
    const double ms = seq_ms(seq, t, sr, rsr);
    return tag == 1u ? amp_on_ms<FC>(p, ms) : amp_off_ms(p, off_amp, ms);
}

// Has the amplitude stopped changing at time t (and therefore for every later t of the run)?  ms is
// non-decreasing in t, so once the decay clamp (envelope.rs:45) or the release clamp (:54) saturates it
// stays saturated: sustain + (1 - sustain) * 0.0 and off_amp * 0.0 are then the same f64 for all later
// samples.  Needs positive decay / release times (otherwise the products are not monotone).
__device__ __forceinline__ bool env_saturated(const EnvParams& p, uint32_t tag, uint64_t seq, uint64_t t, double sr, double rsr) {
    if (tag == 0u) return true;
    const double ms = seq_ms(seq, t, sr, rsr);
    if (tag == 1u) return p.inv_decay > 0.0 && !(ms < p.attack_ms) && p.inv_decay * (ms - p.attack_ms) >= 1.0;
    return p.inv_release > 0.0 && p.inv_release * ms >= 1.0;
}

}  // namespace mx
