// mx_k_envelope.hip -- Envelope: ADSR driven by a gate signal (reference src/module/envelope.rs:8-58,91-120).
//
// Build with -ffp-contract=off (see mx_k_stream.hip); the fma() calls below are explicit.
//
// The reference is a per-sample state machine, but both non-identity inputs are constant maps on
// {not-on, on}: gate == 1.0 forces "on", gate == 0.0 forces "not-on".  So the state bit after
// sample i is the value of the last marker at or before i, edges are where that bit flips, and
// (state, since-when) follows from the last rising / falling edge.  One wave per instance walks the
// stream 64 samples per tile: edges are found with ballots + clz (no shuffles, no LDS), amplitude
// is the closed form per lane, and the carried EnvelopeState lives in SGPRs (v_readlane).
//
// K tiles are loaded up-front per step (K independent loads in flight), and a step whose K tiles
// contain no marker that could flip the carried state -- the overwhelmingly common case: gates are
// Trigger outputs, constant for ticks on end -- takes a branch-free path of ~25 VALU ops per tile.
//
// algorithmic bytes per frame: 4 (gate) + 4 (out).
#include "mx_dev.hpp"
#include "mx_env_math.hpp"

namespace mx {

__device__ __forceinline__ int top_bit(uint64_t m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint64_t read_lane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double read_lane_f64(double v, int l) {
    return __longlong_as_double((long long)read_lane_u64((uint64_t)__double_as_longlong(v), l));
}

template <int K, bool FC>   // FC: the contracted order (MX_FLAG_FP_CONTRACT): amp_on_ms's decay as one fma
__global__ __launch_bounds__(256) void k_envelope(const EnvDesc* __restrict__ descs, EnvState* __restrict__ states,
                                                   uint32_t n_inst, size_t frames, size_t fpc, GateBits gates, uint64_t t0, double sr, double rsr) {
    const int lane = threadIdx.x & 63;
    const uint32_t inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= n_inst) return;  // wave-uniform
    const EnvDesc p = descs[inst];
    // carried EnvelopeState, wave-uniform
    uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)states[inst].tag);
    uint64_t seq = read_lane_u64(states[inst].seq, 0);
    double off_amp = read_lane_f64(states[inst].off_amplitude, 0);

    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t le = lt | (1ull << lane);
    // a Trigger whose params change at tick boundaries inside the run (use_const == 2): the gate of sample i is the bit of tick
    // i / fpc; the lane's tick index advances with its sample index (64 per tile)
    uint32_t g_call = (uint32_t)((size_t)lane / fpc); size_t g_rem = (size_t)lane % fpc;

    for (size_t base = 0; base < frames; base += 64 * K) {
        float xs[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {                       // K independent loads in flight
            const size_t i = base + 64 * k + lane;
            if (p.use_const == 2u) {
                xs[k] = (i < frames && gate_bit(gates, inst, g_call)) ? 1.0f : 0.0f;   // trigger.rs:38-41 with this tick's params
                g_rem += 64;
                while (g_rem >= fpc) { g_rem -= fpc; ++g_call; }
            } else
            xs[k] = p.use_const ? p.gate_const : ((i < frames && p.gate) ? p.gate[i] : 0.0f);   // Disconnected => ZERO_BUFFER_MONO
        }
        // can any marker in these K tiles flip the carried state?  On: only a 0.0; Initial/Off: only a 1.0
        bool quiet = true;
        const float flip = (tag == 1u) ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t i = base + 64 * k + lane;
            quiet = quiet && (__ballot(i < frames && xs[k] == flip) == 0ull);
        }
        const uint64_t span_end = t0 + base + 64 * K;       // one past the last sample time of the step
        if (quiet && (tag == 0u || ((span_end - seq) >> 32) == 0)) {
            // ---- fast path: the carried state holds for the whole step ----
            const uint32_t d0 = (uint32_t)(t0 + base - seq);   // only meaningful when tag != 0
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t i = base + 64 * k + lane;
                double a = 0.0;                                                   // envelope.rs:36
                if (tag != 0u) {                                                  // uniform
                    const double ms = ms_of_u32(d0 + (uint32_t)(64 * k + lane), sr, rsr);
                    a = (tag == 1u) ? amp_on_ms<FC>(p.p, ms) : amp_off_ms(p.p, off_amp, ms);
                }
                if (i < frames) p.out[i] = (float)a;
            }
            continue;
        }
        // ---- general path: tile by tile with edge detection ----
#pragma unroll   // static xs[k] indices: a rolled loop would push the tile registers to scratch
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
            } else if (Fle) {                                         // a falling edge at or before this lane
                const int fl = top_bit(Fle);
                const uint64_t off = tb + (uint64_t)fl;
                const uint64_t Rb = R & ((1ull << fl) - 1ull);
                const uint64_t on = Rb ? tb + (uint64_t)top_bit(Rb) : seq;
                my_tag = 2u; my_seq = off;
                my_off = amp_on_ms<FC>(p.p, seq_ms(on, off, sr, rsr));      // envelope.rs:108-111
            }
            const double ms = seq_ms(my_seq, tb + (uint64_t)lane, sr, rsr);
            const double a_on = amp_on_ms<FC>(p.p, ms);
            const double a_off = amp_off_ms(p.p, my_off, ms);
            const double a = my_tag == 1u ? a_on : (my_tag == 2u ? a_off : 0.0);   // envelope.rs:36
            if (valid) p.out[i] = (float)a;

            const size_t rem = frames - tbase;
            const int last = rem >= 64 ? 63 : (int)rem - 1;
            tag = (uint32_t)__builtin_amdgcn_readlane((int)my_tag, last);
            seq = read_lane_u64(my_seq, last);
            off_amp = read_lane_f64(my_off, last);
        }
    }
    if (lane == 0) { states[inst].tag = tag; states[inst].seq = seq; states[inst].off_amplitude = off_amp; }
}

void launch_envelope(const EnvDesc* d, EnvState* st, uint32_t n, size_t frames, size_t fpc, const GateBits& gates, uint64_t t0, double sample_rate, hipStream_t s, bool fc) {
    if (!n || !frames) return;
    const double rsr = 1.0 / sample_rate;
    if (!fpc) fpc = frames;
#define MX_ENV_GO(K, F) hipLaunchKernelGGL((k_envelope<K, F>), dim3((n + 3) / 4), dim3(256), 0, s, d, st, n, frames, fpc, gates, t0, sample_rate, rsr)
    if (frames > 64 * 4) { if (fc) MX_ENV_GO(8, true); else MX_ENV_GO(8, false); }
    else { if (fc) MX_ENV_GO(2, true); else MX_ENV_GO(2, false); }
#undef MX_ENV_GO
}

}  // namespace mx
