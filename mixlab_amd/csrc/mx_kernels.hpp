// mx_kernels.hpp -- per-kind device descriptors and launcher prototypes (internal).
//
// One launch handles every instance of one module kind at one dependency level: instance = blockIdx.y
// (streaming kernels) or one wave / one lane (recurrences).  Descriptors are plain structs in device
// memory; all port buffers are raw device pointers into the graph's HBM slab (or caller-bound memory).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mx {

// src/module/amplifier.rs:38-60
struct AmpDesc { const float* in; const float* ctl; /* nullptr => Disconnected => mod 1.0 */ float* out; double amplitude; double mod_depth; };

// src/module/envelope.rs:34-58,91-120.  Reciprocals are loop-invariant subexpressions of the
// reference (`1.0 / params.attack_ms * ms`), evaluated once on the host with the same IEEE division.
struct EnvParams { double attack_ms, inv_attack, inv_decay, sustain, one_minus_sustain, inv_release; };
struct EnvDesc {
    const float* gate; float* out;
    float gate_const; uint32_t use_const;   // 1: gate is a Trigger fused in: constant 1.0 / 0.0, no buffer (trigger.rs:38-41)
                                            // 2: ... whose params change at tick boundaries inside this run: one bit per tick (GateBits)
    EnvParams p;
};
struct EnvState { uint32_t tag; uint32_t pad; uint64_t seq; double off_amplitude; };  // EnvelopeState, envelope.rs:8-13

// Per-tick parameter schedule of a Trigger (Engine::client_update between two ticks, src/engine.rs:192-214,277-398, applied
// inside one submission): bit c of an instance's row = gate_open during tick c of the run.  Rows are `words` u32 long,
// instance i of a launch group reads row i; `call_off` = first tick of this launch inside the run.
struct GateBits { const uint32_t* bits; uint32_t words; uint32_t call_off; };
__host__ __device__ inline bool gate_bit(const GateBits& g, uint32_t inst, uint32_t call) {
    const uint32_t c = g.call_off + call;
    return (g.bits[(size_t)inst * g.words + (c >> 5)] >> (c & 31)) & 1u;
}

// State of an Envelope whose gate is such a Trigger, as it ENTERS each tick of the run (k_env_ticks): with a gate that is
// constant over a tick only the tick's first sample can change the state (envelope.rs:99-115), so the EqThree kernels'
// fused epilogue reads one entry per tick and evaluates the closed form (envelope.rs:34-58) per sample from it.
//   flat: the amplitude no longer changes from this tick's first sample on (Initial, sustain reached, release finished):
//   `depth` = the Amplifier's depth() for that constant control (amplifier.rs:71-73), hoisted.
struct EnvTick { uint64_t seq; double off_amp; double depth; uint32_t tag; uint32_t flat; };
struct EnvTickDesc { EnvParams p; double amp_one_minus, amp_mod_depth; EnvState* state; };

// src/module/eq_three.rs:58-89
// epi: fused epilogue chosen by the graph compiler (mx_engine.cpp plan_fusion):
//   0  out[i] = y                                  (plain EqThree)
//   1  out[2i] = out[2i+1] = y                     (EqThree -> StereoPanner with L = R = this EQ)
//   2  out[2i] = out[2i+1] = amp(y, ctl[i])        (... -> Amplifier input; ctl nullptr => 1.0)
// y is the f32 the EQ would have stored, so the fused result is bit-identical to the three modules.
// MX_EQF_MONO_DUP: the stereo result has L == R by construction and every consumer is a Mixer input
//   that understands it, so ONE float per frame is stored (half the write, half the mixer's read).
// MX_EQF_ENV: the Amplifier's control is an Envelope whose gate is a Trigger constant and that feeds
//   nothing else: its closed-form amplitude (envelope.rs:34-58) is evaluated in the epilogue and the
//   control buffer never exists; env_state holds that Envelope's carried state.
enum { MX_EQF_MONO_DUP = 1u, MX_EQF_ENV = 2u };
struct EqDesc {
    const float* in; float* out; double gain_lo, gain_mid, gain_hi;
    const float* ctl; double amp_one_minus, amp_mod_depth, amp_amplitude; uint32_t epi; uint32_t flags;
    EnvParams env;   // MX_EQF_ENV: the folded Envelope's parameters; its per-tick states come from k_env_ticks (EnvTick table)
};
struct EqState { double lo[4]; double hi[4]; double history[3]; double pad; };       // eq_three.rs:13-26,100-103

// time-parallel EqThree: Toeplitz powers of the one-sample pole matrix, per chunk length L (host-computed)
//   pw[f][j] = first column of A_f^(L*j), j = 0..64 ; p2[f][k] = first column of A_f^(L * 2^k), k = 0..5
//   h[f][m]  = A_f^m b_f, m = 0..31 (impulse response of the 4 poles: phase A is 8 dot products against it)
//   cz[f]    = (sum_{m<L} A_f^m) c_f (what the VSA constant alone leaves in a zero-initialised filter after L samples)
struct EqScanTab { double pw[2][65][4]; double p2[2][6][4]; double h[2][32][4]; double cz[2][4]; };

// time-split plan of the scan kernel (see k_eq_three_scan MODE 1/2): wave-uniform kernel arguments
// warm / l2_pre: the pre-pass runs over the last `warm` samples of a span only (whole segments of 256 << l2_pre); see eq_plan_split
struct EqSplit { uint32_t n_split; uint32_t l2_pre; uint32_t stream_out /* outputs too large to stay cached: non-temporal stores */; uint32_t pad; size_t span; size_t warm; double* zbuf /* [n][n_split][8] */; double* bound /* [n][12] */; EnvState* env_snap /* [n] */; };
struct EqSpanPow { double lo[4], hi[4]; };   // first column of A^span per filter (host, long double)

// src/module/fm_sine.rs:37-56
struct FmDesc { const float* in; float* out; double freq_mid, freq_amp; };

// src/module/mixer.rs:46-71
struct MixChan { const float* in; double gain; /* fader * 10^(dB/20), mixer.rs:59 */ uint32_t cue; uint32_t dup; /* input stored as one float per frame (L == R) */ };
struct MixDesc { const MixChan* chans; uint32_t n_ch; uint32_t pad; float* master; float* cue; };

// src/module/oscillator.rs:65-92
struct OscDesc { float* mono; float* stereo; double freq; uint32_t waveform; uint32_t pad; };

// src/module/stereo_panner.rs:30-41 / stereo_splitter.rs:33-47
struct PanDesc { const float* l; const float* r; float* out; };
struct SplitDesc { const float* in; float* l; float* r; };

// src/module/trigger.rs:35-48
struct TrigDesc { float* out; float value; uint32_t pad; };   // a scheduled run reads GateBits instead of `value`

// src/module/plotter.rs:37-56: de-interleave the fired ticks into a staging area
struct PlotJob { const float* in; float* left; float* right; };

// build-specified FIR / rational resampler (mx_k_fir.hip); taps live in device memory, hist = carried input frames
struct FirDesc { const float* in; float* out; const double* taps; float2* hist; uint32_t n_taps; uint32_t pad; };
struct ResampleDesc { const float* in; float* out; const double* taps /* [up][taps_per_phase] */; float2* hist;
                      uint32_t up, down, taps_per_phase, pad; };

// Launchers.  `frames` = mono samples in this run (= n_ticks * SPT); stereo buffers hold 2*frames.
// fc (here and below): MX_FLAG_FP_CONTRACT -- the kernel instantiated for the contracted order (mul_add<true>, mx_env_math.hpp)
void launch_amplifier(const AmpDesc* d, uint32_t n, size_t frames, hipStream_t s, bool fc = false);
void launch_envelope(const EnvDesc* d, EnvState* st, uint32_t n, size_t frames, size_t fpc, const GateBits& gates, uint64_t t0, double sample_rate, hipStream_t s, bool fc = false,
                     void* scratch = nullptr, size_t scratch_bytes = 0 /* envelope_scratch_bytes(): segments of long streams (mx_k_envelope.hip); without it one wave per instance */);
size_t envelope_scratch_bytes(uint32_t n, size_t frames);
// per-tick Envelope states of the folded Envelopes of an EqThree group: ticks[inst][call], `n_calls` ticks of `fpc` samples from t0
void launch_env_ticks(const EnvTickDesc* d, uint32_t n, const GateBits& gates, uint32_t n_calls, size_t fpc, uint64_t t0, double sample_rate, EnvTick* ticks, hipStream_t s, bool fc = false);
// what every EqThree launch needs beyond the descriptors: the per-tick Envelope table (null when no instance folds one)
struct EqRun { size_t frames; size_t fpc /* samples per tick (call) */; uint32_t n_calls; uint32_t fc /* MX_FLAG_FP_CONTRACT: the contracted order */; uint64_t t0; double sr, rsr /* RN(1 / sr), host */, lo_f, hi_f; const EnvTick* ticks /* [n][n_calls] */;
               uint32_t* started = nullptr; uint32_t started_seq = 0; /* tiled speculative kernel: its last workgroup stores started_seq there when it starts (every earlier one has been placed by then): Graph's tail gate */ };
// scratch != nullptr: the split-cascade form for few instances (eq_use_poles_split; eq_poles_scratch_bytes of scratch); else one lane per instance
void launch_eq_three_exact(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r, void* scratch, hipStream_t s);
bool eq_use_poles_split(uint32_t n, size_t frames);
size_t eq_poles_scratch_bytes(uint32_t n, size_t frames);
int eq_scan_log2l(size_t frames);
void launch_eq_three_scan(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r,
                          const EqScanTab* tabs /* 4 tables: L = 4, 8, 16, 32 */, const EqSplit& split, const EqSpanPow& pp, hipStream_t s);
// speculative time-parallel EXACT mode (k_eq_three_spec + k_eq_three_repair): see mx_k_eq_three.hip
struct EqSpecPlan { uint32_t n_chunks; uint32_t chunk; uint32_t warm; uint32_t pad; uint32_t warm_hi /* the high cascade runs over the last warm_hi samples of a warm-up only (tiled kernel) */; };
bool eq_plan_spec(uint32_t n, size_t frames, size_t fpc, double lo_f, double hi_f, EqSpecPlan& plan, bool whole_ticks = false /* an inline Envelope: chunks of whole ticks at any rate */,
                  bool two_tiles = false /* a control buffer: input and control tile per wave, ten waves per CU */);   // false: one lane per instance (launch_eq_three_exact)
size_t eq_spec_scratch_bytes(uint32_t n, const EqSpecPlan& plan);
// one wave that leaves when *flag has reached seq (or after limit_us): a launch queued behind it on its stream starts once the kernel that stores the flag has been placed
void launch_tail_gate(const uint32_t* flag, uint32_t seq, uint32_t limit_us, hipStream_t s);
int eq_epilogue_mode(uint32_t epi, uint32_t flags, bool has_ctl);   // 0..7: (epilogue kind) * 2 + (stereo store); the specialisation key
bool launch_eq_three_spec(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r, const EqSpecPlan& plan, int uniform_mode /* 0..7, or -1: mixed */,
                          void* scratch, uint64_t* stats /* [2]: chunks run, chunks repaired */, hipStream_t s);
void eq_plan_split(uint32_t n, size_t frames, double lo_f, double hi_f, EqSplit& sp);
void launch_fm_sine(const FmDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s, int sin_mode /* mx_k_stream.hip SIN_MODE */);
void launch_mixer(const MixDesc* d, uint32_t n, uint32_t max_ch /* most channels of any mixer in the group */, size_t frames,
                  int dup_mode /* 0 none, 1 all, 2 mixed */, hipStream_t s);
void launch_oscillator(const OscDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s, int sin_mode);
void launch_panner(const PanDesc* d, uint32_t n, size_t frames, hipStream_t s);
void launch_splitter(const SplitDesc* d, uint32_t n, size_t frames, hipStream_t s);
void launch_trigger(const TrigDesc* d, uint32_t n, size_t frames, size_t fpc, const GateBits* gates /* null: constant per instance */, hipStream_t s);
void launch_plotter(const PlotJob* d, uint32_t n, size_t spt, hipStream_t s);
void launch_f32_to_i16(const float* in, int16_t* out, size_t n, int dup, hipStream_t s);
void launch_i16_to_f32(const int16_t* in, float* out, size_t n, hipStream_t s);
struct CopyJob { void* dst; const void* src; size_t bytes; };
void launch_copy_jobs(const CopyJob* device_jobs, uint32_t n, hipStream_t s);   // one block per job
// bytes out of page-locked host memory into device memory BY A KERNEL (the device reads the host buffer): ordered by the queue's own barrier packets like any launch
void launch_upload(void* dst_device, const void* src_pinned_host, size_t bytes, hipStream_t s);
void launch_fir(const FirDesc* d, uint32_t n, uint32_t max_taps, size_t frames, hipStream_t s, bool fc = false);
void launch_resample(const ResampleDesc* d, uint32_t n, uint32_t max_taps, uint32_t tab_doubles /* max up * taps_per_phase */,
                     uint32_t win_frames /* max 255 * down / up + 2 + taps_per_phase */, size_t in_frames, size_t out_frames,
                     uint64_t in_base, uint64_t out_base, hipStream_t s, uint32_t common_up = 0 /* every channel's `up` when they all agree, else 0 */, bool fc = false,
                     uint32_t common_taps = 0, uint32_t common_down = 0 /* likewise taps_per_phase and `down` */);

}  // namespace mx
