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
#include <algorithm>

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

// The edges of one tile of 64 samples and what they make of the carried state, per lane: (my_tag, my_seq, my_off) is the EnvelopeState AFTER the lane's sample
// (envelope.rs:100-116).  Shared by the kernel that writes the amplitudes and by the one that only follows the state (k_env_resolve): one piece of code, one result.
template <bool FC>
__device__ __forceinline__ void env_tile_edges(const EnvParams& pp, const float x, const bool valid, const int lane, const uint64_t tb, const double sr, const double rsr,
                                               const uint32_t tag, const uint64_t seq, const double off_amp, uint32_t& my_tag, uint64_t& my_seq, double& my_off) {
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t le = lt | (1ull << lane);
    const uint64_t m1 = __ballot(valid && x == 1.0f);        // envelope.rs:102
    const uint64_t m0 = __ballot(valid && x == 0.0f);        // envelope.rs:107
    const uint64_t mk = m0 | m1;
    const uint64_t below = mk & lt;
    const bool carry_on = (tag == 1u);
    const bool b_prev = below ? (((m1 >> top_bit(below)) & 1ull) != 0) : carry_on;
    const bool b_cur = ((mk >> lane) & 1ull) ? (((m1 >> lane) & 1ull) != 0) : b_prev;
    const uint64_t R = __ballot(valid && !b_prev && b_cur);  // Initial|Off -> On
    const uint64_t F = __ballot(valid && b_prev && !b_cur);  // On -> Off
    my_tag = tag; my_seq = seq; my_off = off_amp;            // carried Initial / TriggerOff
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
        my_off = amp_on_ms<FC>(pp, seq_ms(on, off, sr, rsr));      // envelope.rs:108-111
    }
}
// the gate of sample i as the Envelope sees it: a buffer (Disconnected => ZERO_BUFFER_MONO), a folded Trigger's constant, or its per-tick bits (trigger.rs:38-41)
__device__ __forceinline__ float env_gate_at(const EnvDesc& p, const GateBits& gates, uint32_t inst, size_t i, size_t frames, size_t fpc) {
    if (i >= frames) return 0.0f;
    if (p.use_const == 2u) return gate_bit(gates, inst, (uint32_t)(i / fpc)) ? 1.0f : 0.0f;
    return p.use_const ? p.gate_const : (p.gate ? p.gate[i] : 0.0f);
}

// SEGMENTS (long streams, few instances).  An Envelope is a state machine over its samples, and one wave per instance walking 25 600 tiles one after the other is
// latency, not work: 15 ms for 1 024 Envelopes x 2 048 ticks on a chip that is idle beside them.  But the state changes only where the gate holds a marker of the
// OTHER kind (a 0.0 while on, a 1.0 while not): k_env_flags notes per tile which kinds of marker it holds (two bits per tile, every tile at once); k_env_resolve --
// one wave per instance -- steps from candidate tile to candidate tile through those bits, runs env_tile_edges on just those tiles, and leaves the state at the
// start of every segment; k_envelope then runs one wave per (instance, segment) from that state.  The state a segment starts from is made by the same code, on the
// same samples, that the one-wave walk would have run: the amplitudes are its bits.
__global__ __launch_bounds__(256) void k_env_flags(const EnvDesc* __restrict__ descs, uint32_t n_inst, size_t frames, size_t fpc, GateBits gates, uint32_t words,
                                                   uint64_t* __restrict__ has1, uint64_t* __restrict__ has0) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), inst = blockIdx.y;
    if (w >= words || inst >= n_inst) return;                // wave-uniform
    const EnvDesc p = descs[inst];
    bool my1 = false, my0 = false;
    for (int t0_ = 0; t0_ < 64; t0_ += 8) {
        float xs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[k] = env_gate_at(p, gates, inst, ((size_t)w * 64 + t0_ + k) * 64 + lane, frames, fpc);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t i = ((size_t)w * 64 + t0_ + k) * 64 + lane;
            const bool a1 = __ballot(i < frames && xs[k] == 1.0f) != 0ull, a0 = __ballot(i < frames && xs[k] == 0.0f) != 0ull;
            if (lane == t0_ + k) { my1 = a1; my0 = a0; }
        }
    }
    const uint64_t b1 = __ballot(my1), b0 = __ballot(my0);
    if (lane == 0) { has1[(size_t)inst * words + w] = b1; has0[(size_t)inst * words + w] = b0; }
}
template <bool FC>
__global__ __launch_bounds__(256) void k_env_resolve(const EnvDesc* __restrict__ descs, EnvState* __restrict__ states, uint32_t n_inst, size_t frames, size_t fpc, GateBits gates,
                                                     uint64_t t0, double sr, double rsr, uint32_t n_seg, size_t seg_len, uint32_t words,
                                                     const uint64_t* __restrict__ has1, const uint64_t* __restrict__ has0, EnvState* __restrict__ seg_state) {
    const int lane = threadIdx.x & 63;
    const uint32_t inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= n_inst) return;  // wave-uniform
    const EnvDesc p = descs[inst];
    uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)states[inst].tag);
    uint64_t seq = read_lane_u64(states[inst].seq, 0);
    double off_amp = read_lane_f64(states[inst].off_amplitude, 0);
    const size_t n_tiles = (frames + 63) / 64, seg_tiles = seg_len / 64;
    for (uint32_t sg = 0; sg < n_seg; ++sg) {
        if (lane == 0) { EnvState e; e.tag = tag; e.pad = 0; e.seq = seq; e.off_amplitude = off_amp; seg_state[(size_t)inst * n_seg + sg] = e; }
        const size_t ta = (size_t)sg * seg_tiles, tb_ = ta + seg_tiles < n_tiles ? ta + seg_tiles : n_tiles;
        size_t t = ta;
        while (t < tb_) {
            // the next tile at or after t that holds a marker of the other kind
            const size_t wd = t / 64;
            uint64_t cand = (tag == 1u ? has0 : has1)[(size_t)inst * words + wd] & (~0ull << (t % 64));
            if (!cand) { t = (wd + 1) * 64; continue; }
            t = wd * 64 + (size_t)__builtin_ctzll(cand);
            if (t >= tb_) break;
            const size_t i = t * 64 + lane;
            const float x = env_gate_at(p, gates, inst, i, frames, fpc);
            uint32_t my_tag; uint64_t my_seq; double my_off;
            env_tile_edges<FC>(p.p, x, i < frames, lane, t0 + t * 64, sr, rsr, tag, seq, off_amp, my_tag, my_seq, my_off);
            const size_t rem = frames - t * 64;
            const int last = rem >= 64 ? 63 : (int)rem - 1;
            tag = (uint32_t)__builtin_amdgcn_readlane((int)my_tag, last);
            seq = read_lane_u64(my_seq, last);
            off_amp = read_lane_f64(my_off, last);
            ++t;
        }
    }
    if (lane == 0) { states[inst].tag = tag; states[inst].seq = seq; states[inst].off_amplitude = off_amp; }
}

template <int K, bool FC>   // FC: the contracted order (MX_FLAG_FP_CONTRACT): amp_on_ms's decay as one fma
__global__ __launch_bounds__(256) void k_envelope(const EnvDesc* __restrict__ descs, EnvState* __restrict__ states,
                                                   uint32_t n_inst, size_t frames_all, size_t fpc, GateBits gates, uint64_t t0, double sr, double rsr,
                                                   uint32_t n_seg, size_t seg_len, const EnvState* __restrict__ seg_state) {
    const int lane = threadIdx.x & 63;
    const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t inst = wv / n_seg, sg = wv - inst * n_seg;
    if (inst >= n_inst) return;  // wave-uniform
    const EnvDesc p = descs[inst];
    // the EnvelopeState this wave starts from, wave-uniform: carried, or (segments) what k_env_resolve left for this segment
    const EnvState* st0 = n_seg > 1 ? seg_state + (size_t)inst * n_seg + sg : states + inst;
    uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)st0->tag);
    uint64_t seq = read_lane_u64(st0->seq, 0);
    double off_amp = read_lane_f64(st0->off_amplitude, 0);
    const size_t start = n_seg > 1 ? (size_t)sg * seg_len : 0;
    const size_t frames = n_seg > 1 ? (start + seg_len < frames_all ? start + seg_len : frames_all) : frames_all;   // this wave's samples: [start, frames)
    if (start >= frames) return;

    // a Trigger whose params change at tick boundaries inside the run (use_const == 2): the gate of sample i is the bit of tick
    // i / fpc; the lane's tick index advances with its sample index (64 per tile)
    uint32_t g_call = (uint32_t)((start + (size_t)lane) / fpc); size_t g_rem = (start + (size_t)lane) % fpc;

    for (size_t base = start; base < frames; base += 64 * K) {
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
            const uint64_t tb = t0 + tbase;
            uint32_t my_tag; uint64_t my_seq; double my_off;
            env_tile_edges<FC>(p.p, x, valid, lane, tb, sr, rsr, tag, seq, off_amp, my_tag, my_seq, my_off);
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
    if (n_seg <= 1 && lane == 0) { states[inst].tag = tag; states[inst].seq = seq; states[inst].off_amplitude = off_amp; }   // (segments: k_env_resolve stored it)
}

// Segments: for streams of at least 64 Ki samples while the instances alone do not fill the chip (one wave each).  Scratch: two marker bitmaps and the segments' states.
static uint32_t env_segments(uint32_t n, size_t frames) {
    const int forced = env_int("MX_ENV_SEGMENTS", -1);             // tests / A/B: 0 or 1 = never, n > 1 = that many (of at least 1 Ki samples)
    if (forced >= 0) return (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)forced, frames / 1024));
    if (frames < (64u << 10)) return 1;
    const size_t by_len = frames / (8u << 10);                     // at least 8 Ki samples per segment
    const size_t want = (4096 + n - 1) / n * 16;                   // (measured, 1 024 Envelopes x 2 048 ticks: 8 segments 4.5 ms, 64 3.7, 128 3.0 -- one wave per instance: 15.1)
    return (uint32_t)std::max<size_t>(1, std::min(by_len, want));
}
size_t envelope_scratch_bytes(uint32_t n, size_t frames) {
    const uint32_t S = env_segments(n, frames);
    if (S <= 1) return 0;
    const size_t words = ((frames + 63) / 64 + 63) / 64;
    return (size_t)n * S * sizeof(EnvState) + 2 * (size_t)n * words * sizeof(uint64_t);
}

void launch_envelope(const EnvDesc* d, EnvState* st, uint32_t n, size_t frames, size_t fpc, const GateBits& gates, uint64_t t0, double sample_rate, hipStream_t s, bool fc,
                     void* scratch, size_t scratch_bytes) {
    if (!n || !frames) return;
    const double rsr = 1.0 / sample_rate;
    if (!fpc) fpc = frames;
    const uint32_t S_plan = env_segments(n, frames);                // (MX_ENV_SEGMENTS is read per launch on purpose -- the tests switch it inside one process -- but once per launch)
    uint32_t S = S_plan;
    if (S > 1) {
        const size_t words0 = ((frames + 63) / 64 + 63) / 64;
        if (!scratch || scratch_bytes < (size_t)n * S * sizeof(EnvState) + 2 * (size_t)n * words0 * sizeof(uint64_t)) S = 1;
    }
    size_t seg_len = frames;
    const EnvState* seg_state = nullptr;
    if (S > 1) {
        seg_len = ((frames + S - 1) / S + 511) / 512 * 512;        // whole steps of K = 8 tiles
        S = (uint32_t)((frames + seg_len - 1) / seg_len);
    }
    if (S > 1) {
        const uint32_t words = (uint32_t)(((frames + 63) / 64 + 63) / 64);
        EnvState* ss = (EnvState*)scratch;
        uint64_t* has1 = (uint64_t*)(ss + (size_t)n * S_plan);
        uint64_t* has0 = has1 + (size_t)n * words;
        hipLaunchKernelGGL(k_env_flags, dim3((words + 3) / 4, n), dim3(256), 0, s, d, n, frames, fpc, gates, words, has1, has0);
        if (fc) hipLaunchKernelGGL(k_env_resolve<true>, dim3((n + 3) / 4), dim3(256), 0, s, d, st, n, frames, fpc, gates, t0, sample_rate, rsr, S, seg_len, words, has1, has0, ss);
        else hipLaunchKernelGGL(k_env_resolve<false>, dim3((n + 3) / 4), dim3(256), 0, s, d, st, n, frames, fpc, gates, t0, sample_rate, rsr, S, seg_len, words, has1, has0, ss);
        seg_state = ss;
    }
    const uint32_t waves = n * S;
#define MX_ENV_GO(K, F) hipLaunchKernelGGL((k_envelope<K, F>), dim3((waves + 3) / 4), dim3(256), 0, s, d, st, n, frames, fpc, gates, t0, sample_rate, rsr, S, seg_len, seg_state)
    if (frames > 64 * 4) { if (fc) MX_ENV_GO(8, true); else MX_ENV_GO(8, false); }
    else { if (fc) MX_ENV_GO(2, true); else MX_ENV_GO(2, false); }
#undef MX_ENV_GO
}

}  // namespace mx
