// mx_k_mixer.hip -- Mixer: order-preserving f32 mix bus (reference src/module/mixer.rs:46-71).
//
// Build with -ffp-contract=off (see mx_k_stream.hip).
//
//   master[i] += (in[ch][i] as f64 * gain[ch]) as f32 ;  if cue[ch] { cue[i] += in[ch][i] }
//
// The f32 accumulation order over channels IS the result (reordering a 1024-way sum moves it by
// tens of ULPs), so every output element owns one strictly sequential add chain and nothing is
// split across lanes, waves or MFMA.  What is left to engineer is memory-level parallelism:
//   * one wave per block, W floats per lane: even a single tick (1 600 outputs) spreads over CUs;
//   * a ring of R channel loads stays in flight per lane ahead of the add chain: the load for
//     channel c+R is issued in the slot channel c just vacated, so the chain never drains;
//   * channel descriptors are fetched 64 at a time, one per lane (a coalesced vector load), and
//     broadcast with v_readlane into SGPRs -- no dependent scalar-memory round trip per channel.
// algorithmic bytes per mixer per frame: 8 * (n_ch + 2).
#include "mx_dev.hpp"

namespace mx {

// Native vector types so the loads can be written against address_space(1) directly.
template <int W> struct VecF;
template <> struct VecF<1> { typedef float T; };
template <> struct VecF<2> { typedef float __attribute__((ext_vector_type(2))) T; };
template <> struct VecF<4> { typedef float __attribute__((ext_vector_type(4))) T; };

// Channel pointers travel through SGPRs as integers (v_readlane), so the address space must be
// restated: a plain float* built from an integer is a FLAT pointer, and flat loads count against
// both vmcnt and lgkmcnt, which serialises the ring.  address_space(1) + a 32-bit lane offset
// selects `global_load_* v, v_off, s[base:base+1]`: one VGPR of address per lane for all channels.
// No tail handling: the launcher only picks W that divides the stream length.
template <int W>
__device__ __forceinline__ void ldw(uint64_t base, uint32_t byte_off, float (&v)[W]) {
    typedef typename VecF<W>::T VT;
    const char __attribute__((address_space(1)))* p = (const char __attribute__((address_space(1)))*)base;
    const VT t = __builtin_nontemporal_load((const VT __attribute__((address_space(1)))*)(p + byte_off));   // streamed once
    if constexpr (W == 1) v[0] = t;
    else {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = t[k];
    }
}
// input stored as one float per frame (a stereo port whose L == R by construction, see EqDesc
// MX_EQF_MONO_DUP): fetch W/2 floats from half the byte offset and duplicate them in registers
template <int W>
__device__ __forceinline__ void ldw_dup(uint64_t base, uint32_t byte_off, float (&v)[W]) {
    float h[W / 2];
    ldw<W / 2>(base, byte_off >> 1, h);
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = h[k >> 1];
}
// DUP: 0 = no input is mono-dup, 1 = every input is, 2 = mixed (per-channel branch; the branch costs the
// compiler its vmcnt bookkeeping across the ring, so the homogeneous cases get their own instantiation)
template <int W, int DUP>
__device__ __forceinline__ void ldw_any(uint64_t base, uint32_t byte_off, bool dup, float (&v)[W]) {
    if constexpr (DUP == 0) ldw<W>(base, byte_off, v);
    else if constexpr (DUP == 1) ldw_dup<W>(base, byte_off, v);
    else { if (dup) ldw_dup<W>(base, byte_off, v); else ldw<W>(base, byte_off, v); }   // wave-uniform
}
template <int W>
__device__ __forceinline__ void stw(float* __restrict__ pf, uint32_t byte_off, const float (&v)[W]) {
    typedef typename VecF<W>::T VT;
    char __attribute__((address_space(1)))* p = (char __attribute__((address_space(1)))*)(uint64_t)pf;
    VT t;
    if constexpr (W == 1) t = v[0];
    else {
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = v[k];
    }
    *(VT __attribute__((address_space(1)))*)(p + byte_off) = t;
}

__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

template <int W>
__device__ __forceinline__ void mix_one(float (&acc)[W], float (&cac)[W], const float (&x)[W], double g, bool cue) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
        acc[k] += (float)((double)x[k] * g);   // mixer.rs:62
        if (cue) cac[k] += x[k];               // mixer.rs:64-66
    }
}

// W floats per lane, ring of R loads in flight per lane, one wave per block.
template <int W, int R, int DUP>
__global__ __launch_bounds__(64) void k_mixer(const MixDesc* __restrict__ descs, size_t n /* stereo floats */) {
    static_assert(64 % R == 0, "ring must divide the descriptor block");
    const MixDesc m = descs[blockIdx.y];
    const MixChan* __restrict__ ch = m.chans;
    const int lane = threadIdx.x;
    const size_t items = n / W;   // W divides n (launcher)
    const uint32_t n_full = m.n_ch & ~63u;   // channels covered by full 64-channel descriptor blocks

    for (size_t qb = (size_t)blockIdx.x * 64; qb < items; qb += (size_t)gridDim.x * 64) {
        // the whole wave runs the channel loop together (v_readlane broadcasts need every lane's
        // descriptor registers): lanes past the end are clamped to the last item and not stored
        const bool live = qb + lane < items;
        const uint32_t off = (uint32_t)((live ? qb + lane : items - 1) * (W * sizeof(float)));   // < 4 GiB per buffer (launcher)
        float acc[W], cac[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { acc[k] = 0.f; cac[k] = 0.f; }   // util::zero(master/cue), mixer.rs:54-55

        if (n_full) {
            // descriptors of the current and the next 64-channel block, one channel per lane
            uint64_t p_cur = (uint64_t)ch[lane].in, p_nxt = 0;
            uint64_t g_cur = (uint64_t)__double_as_longlong(ch[lane].gain), g_nxt = 0;
            uint64_t cue_cur = __ballot(ch[lane].cue != 0), cue_nxt = 0;
            uint64_t dup_cur = __ballot(ch[lane].dup != 0), dup_nxt = 0;
            float v[R][W];
#pragma unroll
            for (int u = 0; u < R; ++u) ldw_any<W, DUP>(bcast_u64(p_cur, u), off, ((dup_cur >> u) & 1ull) != 0, v[u]);   // prologue: fill the ring

            for (uint32_t c0 = 0; c0 < n_full; c0 += 64) {
                const bool have_next = c0 + 64 < n_full;   // uniform
                if (have_next) {
                    p_nxt = (uint64_t)ch[c0 + 64 + lane].in;
                    g_nxt = (uint64_t)__double_as_longlong(ch[c0 + 64 + lane].gain);
                    cue_nxt = __ballot(ch[c0 + 64 + lane].cue != 0);
                    dup_nxt = __ballot(ch[c0 + 64 + lane].dup != 0);
                }
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    float x[W];
#pragma unroll
                    for (int k = 0; k < W; ++k) x[k] = v[u % R][k];
                    // refill the slot with channel c0 + u + R
                    if (u + R < 64) {
                        ldw_any<W, DUP>(bcast_u64(p_cur, (u + R) & 63), off, ((dup_cur >> ((u + R) & 63)) & 1ull) != 0, v[u % R]);
                    } else if (have_next) {
                        ldw_any<W, DUP>(bcast_u64(p_nxt, (u + R) & 63), off, ((dup_nxt >> ((u + R) & 63)) & 1ull) != 0, v[u % R]);
                    }
                    const double g = __longlong_as_double((long long)bcast_u64(g_cur, u));
                    mix_one<W>(acc, cac, x, g, ((cue_cur >> u) & 1ull) != 0);
                }
                p_cur = p_nxt; g_cur = g_nxt; cue_cur = cue_nxt; dup_cur = dup_nxt;
            }
        }
        // remaining (< 64) channels: plain scalar-descriptor path
        for (uint32_t c = n_full; c < m.n_ch; ++c) {
            float x[W];
            ldw_any<W, DUP>((uint64_t)ch[c].in, off, ch[c].dup != 0, x);
            mix_one<W>(acc, cac, x, ch[c].gain, ch[c].cue != 0);
        }
        if (live) {
            stw<W>(m.master, off, acc);
            stw<W>(m.cue, off, cac);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Short streams (the real-time mode: one tick = 800 frames): k_mixer is then a handful of waves, each walking all
// channels with at most 32 loads in flight -- pure load latency (61 us for 1024 channels, whatever the chip).
// k_mixer_coop keeps the single ordered add chain per output but lets SIXTEEN waves work for it: a block owns 64
// frames; per batch of channels every wave loads its share, forms the products (in as f64 * gain) as f32 and the cue
// term (x or +0.0) and lands {product, cue term} in LDS; wave 0 then only walks the batch in channel order with one
// packed f32 add per channel -- the ordered sums -- while the next batch is already in flight.
// The cue term as a select is exact: the cue bus never holds -0.0 (it starts +0.0 and +0.0 + -0.0 = +0.0), so adding
// +0.0 for a non-cue channel changes nothing, NaN and infinity included.
// ---------------------------------------------------------------------------------------------
#define MC_WAVES 16                     // wave 0 only adds; waves 1..15 load, multiply and land
#define MC_PROD (MC_WAVES - 1)
template <int DUP> struct McCfg { static constexpr int FW = DUP ? 1 : 2; static constexpr int PER = DUP ? 8 : 4; static constexpr int BATCH = PER * MC_PROD; };

template <int DUP>   // 1: every input stored mono (L == R), 0: every input interleaved stereo
__global__ __launch_bounds__(64 * MC_WAVES) void k_mixer_coop(const MixDesc* __restrict__ descs, size_t frames) {
    constexpr int FW = McCfg<DUP>::FW, BATCH = McCfg<DUP>::BATCH, PER = McCfg<DUP>::PER;
    typedef float __attribute__((ext_vector_type(2 * FW))) Slot;      // {products[FW], cue terms[FW]} of one frame and channel
    const MixDesc m = descs[blockIdx.y];
    const MixChan* __restrict__ ch = m.chans;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ __attribute__((aligned(16))) char mc_smem[];
    Slot* lds = reinterpret_cast<Slot*>(mc_smem);                     // [2][BATCH][64]
    const size_t f0 = (size_t)blockIdx.x * 64;
    const bool live = f0 + lane < frames;
    const uint32_t foff = (uint32_t)((live ? f0 + lane : frames - 1) * (FW * sizeof(float)));   // lanes past the end re-read the last frame, never stored
    const uint32_t n_ch = m.n_ch, n_batch = (n_ch + BATCH - 1) / BATCH;

    // descriptors of my share of a batch: one channel per lane (lanes < PER), broadcast with v_readlane at use; channels past
    // the end repeat the last one with gain 0 and no cue (+0.0 terms).  Disconnected inputs point at the graph's zero buffer.
    const int prod = wave - 1;                                        // producer index of waves 1..15 (wave 0: -1, loads nothing)
    struct Desc { uint64_t p, g; uint32_t cue; bool valid; };
    struct Inflight { float v[PER][FW]; uint64_t g; uint32_t cue; };  // loads of one batch in flight + what landing them needs
    // Nothing in the producers' steady state may sit under a condition or use a value the moment it is requested: the compiler's wait
    // counters are exact only along straight-line code, and a conservative vmcnt(0) before every landing made each step one full
    // memory latency long (measured: 1.7 us per batch of 120 channels whatever the prefetch depth; PMC: the waves waited 62 % of
    // their cycles).  So: descriptors are requested a step before the loads that need them, three batches of samples are in flight,
    // every request is unconditional (past the end the last channel is re-read and lands as +0.0), and the loop is unrolled by three
    // so that register sets are named, not indexed.
    auto fetch = [&](Desc& dsc, uint32_t b) {
        const uint32_t c = b * BATCH + (uint32_t)prod * PER + (lane < PER ? lane : 0);
        const MixChan* mc = ch + (c < n_ch ? c : n_ch - 1);
        dsc.p = (uint64_t)mc->in;
        dsc.g = (uint64_t)__double_as_longlong(mc->gain);
        dsc.cue = mc->cue;
        dsc.valid = c < n_ch;
    };
    auto issue = [&](Inflight& st, const Desc& dsc) {
        st.g = dsc.valid ? dsc.g : (uint64_t)__double_as_longlong(0.0);
        st.cue = (uint32_t)__ballot(dsc.valid && dsc.cue != 0);
        const uint64_t pl = dsc.p;
#pragma unroll
        for (int u = 0; u < PER; ++u) ldw<FW>(bcast_u64(pl, u), foff, st.v[u]);
    };
    auto land = [&](const Inflight& st, uint32_t b) {
        const uint32_t c0 = b * BATCH + (uint32_t)prod * PER;
        const uint32_t valid = c0 >= n_ch ? 0u : (n_ch - c0 < (uint32_t)PER ? n_ch - c0 : (uint32_t)PER);   // channels of mine that exist
        Slot* dst = lds + ((size_t)(b & 1) * BATCH + (size_t)prod * PER) * 64 + lane;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const double g = __longlong_as_double((long long)bcast_u64(st.g, u));
            const bool cue = ((st.cue >> u) & 1u) != 0;
            Slot sl;
#pragma unroll
            for (int k = 0; k < FW; ++k) {
                const float x = (uint32_t)u < valid ? st.v[u][k] : 0.0f;             // a channel past the end contributes +0.0 + +0.0
                sl[k] = (float)((double)x * g);                                      // mixer.rs:62 (the product)
                sl[FW + k] = cue ? x : 0.0f;                                         // mixer.rs:64-66 (what the cue bus gets)
            }
            dst[(size_t)u * 64] = sl;
        }
    };

    if (wave != 0) {
        // ---- producers ----
        if (n_ch) {   // a Mixer with no channels (params_len 0) sharing the launch: nothing to fetch, its buses stay +0.0 (mixer.rs:54-55)
            // ring of three: batch k travels in register set k % 3 (more sets were measured: no faster)
            Desc D0, D1, D2; Inflight S0, S1, S2;
            fetch(D0, 0); fetch(D1, 1); fetch(D2, 2);
            issue(S0, D0); issue(S1, D1); issue(S2, D2);
            fetch(D0, 3);
            land(S0, 0);
            __syncthreads();
            // step s: descriptors of batch s + 4, samples of batch s + 3 (descriptors a step old), landing of batch s + 1 (samples two steps old)
            auto step = [&](uint32_t s_, Inflight& sx, const Desc& dx, Desc& dy, const Inflight& sn) {
                fetch(dy, s_ + 4);
                issue(sx, dx);
                if (s_ + 1 < n_batch) { land(sn, s_ + 1); __syncthreads(); }          // the other LDS buffer: wave 0 finished with it one barrier ago
            };
            for (uint32_t b = 0; b < n_batch; b += 3) {
                step(b, S0, D0, D1, S1);
                step(b + 1, S1, D1, D2, S2);
                step(b + 2, S2, D2, D0, S0);
            }
        } else {
            __syncthreads();
        }
        return;
    }

    // ---- wave 0: the ordered sums ----
    __syncthreads();
    Slot acc;                                                         // {master[FW], cue[FW]}: both ordered sums in one packed add per channel (scalar add chains were measured: no faster)
#pragma unroll
    for (int k = 0; k < 2 * FW; ++k) acc[k] = 0.f;                    // util::zero(master/cue), mixer.rs:54-55
    auto add_slot = [&](const Slot& t) { acc += t; };
    for (uint32_t b = 0; b < n_batch; ++b) {
        const Slot* src = lds + (size_t)(b & 1) * BATCH * 64 + lane;
        const uint32_t c0 = b * BATCH, cnt = n_ch - c0 < (uint32_t)BATCH ? n_ch - c0 : (uint32_t)BATCH;
        uint32_t u = 0;
        if (cnt >= 8) {                                               // groups of 8, the next group's LDS reads in flight under this group's adds
            Slot t[8], nxt8[8];
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) t[jx] = src[(size_t)jx * 64];
            for (; u + 16 <= cnt; u += 8) {
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) nxt8[jx] = src[(size_t)(u + 8 + jx) * 64];
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) add_slot(t[jx]);       // channel order
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) t[jx] = nxt8[jx];
            }
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) add_slot(t[jx]);
            u += 8;
        }
        for (; u < cnt; ++u) add_slot(src[(size_t)u * 64]);
        if (b + 1 < n_batch) __syncthreads();
    }
    if (live) {
        float2* om = reinterpret_cast<float2*>(m.master) + f0 + lane;
        float2* oc = reinterpret_cast<float2*>(m.cue) + f0 + lane;
        if constexpr (DUP) { *om = make_float2(acc[0], acc[0]); *oc = make_float2(acc[1], acc[1]); }   // L == R inputs: one chain is both
        else { *om = make_float2(acc[0], acc[1]); *oc = make_float2(acc[2], acc[3]); }
    }
}

void launch_mixer(const MixDesc* d, uint32_t n, uint32_t max_ch, size_t frames, int dup_mode, hipStream_t s) {
    if (!n || !frames) return;
    const size_t ns = frames * 2;
    // widest lane vector that still yields enough waves to cover the chip; tuning override for experiments
    static const int force_w = env_int("MX_MIXER_W", 0);
    int w = force_w;
    if (w != 2 && w != 4) {
        const size_t want_lanes = (size_t)64 * 1024 / (n ? n : 1);
        w = (ns / 4 >= want_lanes) ? 4 : 2;   // one stereo frame per lane is the narrowest form
    }
    if (w == 4 && (ns & 3)) w = 2;                 // ns = 2 * frames is always even
    if (ns * sizeof(float) >= (1ull << 32)) return;  // unreachable: the engine caps a port buffer below 4 GiB
    // short streams with many channels: the cooperative form (see k_mixer_coop)
    const int coop_max_blocks = env_int("MX_MIXER_COOP_BLOCKS", 512);   // read per call: tests force either kernel
    const size_t coop_blocks = (frames + 63) / 64 * n;
    if (dup_mode != 2 && max_ch >= 128 && coop_blocks <= (size_t)coop_max_blocks) {
        const size_t lds = (size_t)2 * (dup_mode ? McCfg<1>::BATCH * McCfg<1>::FW : McCfg<0>::BATCH * McCfg<0>::FW) * 64 * 2 * sizeof(float);
        const dim3 g((unsigned)((frames + 63) / 64), n);
        if (dup_mode) hipLaunchKernelGGL(k_mixer_coop<1>, g, dim3(64 * MC_WAVES), lds, s, d, frames);
        else hipLaunchKernelGGL(k_mixer_coop<0>, g, dim3(64 * MC_WAVES), lds, s, d, frames);
        return;
    }
    const size_t items = ns / w;
    // (round 5: a ring of 64 loads per lane for launches of fewer waves than SIMDs -- 1024 strips x 64 ticks is 800 waves, 6.5 MB in flight, 1.9 - 2.5 TB/s -- was built
    // and measured: 0.084 -> 0.449 ms.  Every refill then comes from the NEXT descriptor block, whose descriptor loads are a step old at best: the first refill waits for
    // them with vmcnt(0) and the ring drains.  With 32 the first half of a block refills from descriptors already in registers.)
    dim3 grid(grid_x(items, 64, 16384), n);
#define MX_MIX_LAUNCH(W, R, D) hipLaunchKernelGGL((k_mixer<W, R, D>), grid, dim3(64), 0, s, d, ns)
    if (w == 4) { if (dup_mode == 0) MX_MIX_LAUNCH(4, 16, 0); else if (dup_mode == 1) MX_MIX_LAUNCH(4, 16, 1); else MX_MIX_LAUNCH(4, 16, 2); }
    else        { if (dup_mode == 0) MX_MIX_LAUNCH(2, 32, 0); else if (dup_mode == 1) MX_MIX_LAUNCH(2, 32, 1); else MX_MIX_LAUNCH(2, 32, 2); }
#undef MX_MIX_LAUNCH
}

}  // namespace mx
