// mx_k_eq_three.hip -- EqThree, the opt-in FAST mode (MX_FLAG_EQ_FAST): time-parallel chunked scan with the graph compiler's
// fused epilogue.  NOT bit-exact: f32 outputs within 1 ULP of the reference order (src/module/eq_three.rs:58-89), about one
// sample in 20 000 differs.  The default, exact-order kernels are in mx_k_eq_exact.hip.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written,
// never fused; the recurrence in phase C must not become v_fma_f64.  Explicit fma() calls appear
// only in the scan's helper arithmetic, whose rounding is free by construction.
#include <algorithm>
#include <cmath>

#include "mx_k_eq_common.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// time-parallel.  The two 4-pole cascades are affine recurrences
//   s[n+1] = A s[n] + b x[n] + c ,  A = lower-triangular Toeplitz with first column f^k (1-f),
//   b = (f, f^2, f^3, f^4), c = VSA (1, f, f^2, f^3)
// so a stream can be cut into chunks that are processed concurrently:
//   one 256-thread workgroup per instance walks its stream in segments of 256 chunks x L samples,
//   staged through LDS with coalesced burst loads (lane stride L+1 words => conflict-free ds_read_b32);
//   phase A: z_j = state a zero-initialised filter reaches after chunk j = sum_m (A^m b) x[L-1-m] + const,
//            8 independent f64 FMA chains against host tables held in SGPRs (wave-uniform);
//   scan:    S_j = A^(L j) S_seg + sum_{i<j} A^(L (j-1-i)) z_i  (Hillis-Steele over the wave with f64
//            shuffles and Toeplitz powers from LDS, then a 4-entry hop across waves);
//   phase C: every lane runs the EXACT recurrence (pump) from its true initial state and emits.
// Only chunk-initial states differ from the sequential order, by ~1e-16 relative; f32 outputs stay
// within 1 ULP of the reference order (tests/test_gpu_audio_parity.py counts the mismatches).
// ---------------------------------------------------------------------------------------------
// LDS tile layout: element e of a segment lives at float position swz(e) = e with its low 5 bits XORed
// by bits 5..9 -- a bijection inside every 32-float row.  It makes lane j's walk over its own chunk
// (elements j*L .. j*L+L-1) conflict-free for ds_read_b32 / ds_write_b32 for every L in {4,8,16,32}
// AND keeps rows contiguous, which is what the LDS-DMA stage-in needs: `global_load_lds_dword` writes
// 64 consecutive floats per wave instruction straight from HBM into LDS, no VGPRs, so the swizzle is
// applied to the SOURCE address (lane at LDS position p fetches element swz(p); swz is an involution).
__device__ __forceinline__ int swz(int e) { return (e & ~31) | ((e & 31) ^ ((e >> 5) & 31)); }
typedef const float __attribute__((address_space(1)))* mx_gfp;
typedef float __attribute__((address_space(3)))* mx_lfp;

// y = T(c) v for a lower-triangular Toeplitz matrix with first column c
__device__ __forceinline__ void toep_apply(const double* c, const double (&v)[4], double (&y)[4]) {
    y[0] = c[0] * v[0];
    y[1] = fma(c[1], v[0], c[0] * v[1]);
    y[2] = fma(c[2], v[0], fma(c[1], v[1], c[0] * v[2]));
    y[3] = fma(c[3], v[0], fma(c[2], v[1], fma(c[1], v[2], c[0] * v[3])));
}

// MODE 0: one workgroup walks the whole stream of its instance (enough instances to fill the chip).
// Few instances, long streams (a sharded rank, a small graph): the stream is cut into n_split spans handled by
// different workgroups --
//   MODE 1 (pre-pass, spans 0 .. n_split-2): stage-in + phase A + scan only; leaves the state a zero-initialised
//          filter reaches at the end of the span (zbuf).  It starts `warm` samples before the span's end: the
//          poles forget -- what older samples add to the end state is below 2^-280 of their magnitude
//          (eq_plan_split picks `warm` from the pole), i.e. below 2^-152 in absolute terms for EVERY finite
//          f32 input: an eighth of the smallest positive f32 subnormal.  Its span-0
//          workgroup also snapshots the carried EqState / EnvelopeState (the main pass overwrites them);
//   MODE 2 (main): like MODE 0 on one span; the prologue folds the snapshot and the zbuf entries of the spans
//          before it into the true state at the span start: S_0 = carried, S_{s+1} = A^span S_s + Z_s.
struct EqSegCtx {
    float* tile; double* wtot; double* carry; double* pw; double* p2; const EqEpi* epi;
    const float* din; double g_lo, g_mid, g_hi, lo_f, hi_f, sr, rsr; uint64_t t0; size_t fpc; bool stream_out;
};

template <int LOG2L>
__device__ __forceinline__ void eq_load_tables(const EqSegCtx& c, const EqScanTab* __restrict__ tabs, int tid) {
    const EqScanTab* tab = tabs + (LOG2L - 2);
    for (int i = tid; i < 2 * 65 * 4; i += 256) c.pw[i] = (&tab->pw[0][0][0])[i];
    if (tid < 2 * 6 * 4) c.p2[tid] = (&tab->p2[0][0][0])[tid];
}

// one segment of up to 64 * NW chunks x L samples starting at sample `base` (nv valid samples); ends on a barrier.
// NW = waves per instance: 4 (a 256-thread workgroup walks one stream) or 1 (short streams: one wave per instance, four
// instances per workgroup, no cross-wave step)
template <int LOG2L, int MODE, int NW = 4>
__device__ __forceinline__ void eq_segment(const EqSegCtx& c, const EqScanTab* __restrict__ tabs, const size_t base, const int nv) {
    constexpr int L = 1 << LOG2L;
    const EqScanTab* __restrict__ tab = tabs + (LOG2L - 2);
    float* const tile = c.tile;
    double* const wtot = c.wtot; double* const carry = c.carry; double* const pw = c.pw; double* const p2 = c.p2;
    const int lane = threadIdx.x & 63;
    const int tid = NW == 4 ? (int)threadIdx.x : lane;        // thread index inside the instance
    const int wave = NW == 4 ? (int)(threadIdx.x >> 6) : 0;   // wave index inside the instance
    constexpr int NT = 64 * NW;                               // threads (= chunks per segment) of an instance
    const float* __restrict__ din = c.din;
    const double lo_f = c.lo_f, hi_f = c.hi_f;

    // stage-in by LDS-DMA: each wave issues L `global_load_lds_dword` (256 B each), no VGPRs involved
    if (din) {
#pragma unroll 4   // DMA needs no data registers, but a full unroll materialises L 64-bit addresses at once
        for (int kk = 0; kk < L; ++kk) {
            const int n = wave + NW * kk;                        // 64-float block of the tile
            const int p = n * 64 + lane;                         // LDS float position this lane fills
            const int e = swz(p);                                // ... with this element of the segment
            const int ec = e < nv ? e : nv - 1;                  // past-the-end positions are never read; keep the address legal
            __builtin_amdgcn_global_load_lds((mx_gfp)(din + base + ec), (mx_lfp)(tile + n * 64), 4, 0, 2);   // aux = nt: the source is streamed once
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll 8
        for (int k = 0; k < L; ++k) tile[tid + NT * k] = 0.f;   // Disconnected input => ZERO_BUFFER_MONO
    }
    __syncthreads();

    const int start = tid << LOG2L;
    const int my_n = nv - start >= L ? L : (nv - start > 0 ? nv - start : 0);
    // my chunk: elements start .. start+L-1 sit in one 32-float row; position of element start+i is pbase + (i ^ xl)
    const int xr = (start >> 5) & 31, xl = xr & (L - 1);
    float* mine = tile + (start & ~31) + ((start & 31) ^ (xr & ~(L - 1)));

    // phase A: zero-state response of a full chunk as 8 dot products (tables wave-uniform)
    double zl[4] = {0.0, 0.0, 0.0, 0.0}, zh[4] = {0.0, 0.0, 0.0, 0.0};
    if (my_n == L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { zl[q] = tab->cz[0][q]; zh[q] = tab->cz[1][q]; }
#pragma unroll 2   // each step pulls 8 table doubles into SGPRs: a deeper unroll overflows the scalar file and spills
        for (int i = 0; i < L; ++i) {
            const double x = (double)mine[i ^ xl];
            const int m = L - 1 - i;
#pragma unroll
            for (int q = 0; q < 4; ++q) { zl[q] = fma(tab->h[0][m][q], x, zl[q]); zh[q] = fma(tab->h[1][m][q], x, zh[q]); }
        }
    }
    // the three samples before my chunk (the EQ's 3-sample delay line), read before anyone overwrites the tile
    double h0, h1, h2;
    if (tid == 0) { h0 = carry[8]; h1 = carry[9]; h2 = carry[10]; }
    else { h0 = (double)tile[swz(start - 3)]; h1 = (double)tile[swz(start - 2)]; h2 = (double)tile[swz(start - 1)]; }

    // inclusive scan over the wave: E_j = sum_{i<=j} P^(j-i) z_i
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int dd = 1 << k;
        double ul[4], uh[4], tl[4], th[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ul[q] = __shfl_up(zl[q], dd); uh[q] = __shfl_up(zh[q], dd); }
        toep_apply(p2 + (0 * 6 + k) * 4, ul, tl);
        toep_apply(p2 + (1 * 6 + k) * 4, uh, th);
        if (lane >= dd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { zl[q] += tl[q]; zh[q] += th[q]; }
        }
    }
    if (NW > 1 && lane == 63) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { wtot[wave * 8 + q] = zl[q]; wtot[wave * 8 + 4 + q] = zh[q]; }
    }
    __syncthreads();
    // state entering my wave: C_w = P^64 C_{w-1} + W_{w-1}, C_0 = segment-in state
    double cl[4] = {carry[0], carry[1], carry[2], carry[3]}, ch[4] = {carry[4], carry[5], carry[6], carry[7]};
    for (int w = 0; w < wave; ++w) {
        double tl[4], th[4];
        toep_apply(pw + (0 * 65 + 64) * 4, cl, tl);
        toep_apply(pw + (1 * 65 + 64) * 4, ch, th);
#pragma unroll
        for (int q = 0; q < 4; ++q) { cl[q] = tl[q] + wtot[w * 8 + q]; ch[q] = th[q] + wtot[w * 8 + 4 + q]; }
    }
    // my chunk's true initial state: S = P^lane C_w + E_{lane-1}   (lane 0: P^0 = I, E = 0 => S = C_w exactly)
    double lo[4], hi[4];
    {
        double el[4], eh[4], tl[4], th[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { el[q] = __shfl_up(zl[q], 1); eh[q] = __shfl_up(zh[q], 1); }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { el[q] = 0.0; eh[q] = 0.0; }
        }
        toep_apply(pw + (0 * 65 + lane) * 4, cl, tl);
        toep_apply(pw + (1 * 65 + lane) * 4, ch, th);
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = tl[q] + el[q]; hi[q] = th[q] + eh[q]; }
    }
    if (MODE == 1) {
        // pre-pass: whole segments only; the state at the segment's end is the end state of chunk 255
        //   = P^(64) C_3 + E_63 (lane 63 of wave 3).  No phase C, nothing is emitted.
        __syncthreads();
        if (tid == NT - 1) {
            double tl[4], th[4];
            toep_apply(pw + (0 * 65 + 64) * 4, cl, tl);
            toep_apply(pw + (1 * 65 + 64) * 4, ch, th);
#pragma unroll
            for (int q = 0; q < 4; ++q) { carry[q] = tl[q] + zl[q]; carry[4 + q] = th[q] + zh[q]; }
        }
        __syncthreads();
        return;
    }
    // phase C: exact recurrence from the true state, outputs overwrite my chunk of the tile
    const double g_lo = c.g_lo, g_mid = c.g_mid, g_hi = c.g_hi;
    if (my_n == L) {
#pragma unroll 4
        for (int i = 0; i < L; ++i) {
            const double sample = (double)mine[i ^ xl];
            const double l = pump(lo_f, lo, sample);
            const double h = h0 - pump(hi_f, hi, sample);
            const double mid = h0 - (h + l);
            h0 = h1; h1 = h2; h2 = sample;
            mine[i ^ xl] = (float)(l * g_lo + mid * g_mid + h * g_hi);
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < my_n; ++i) {
            const double sample = (double)mine[i ^ xl];
            const double l = pump(lo_f, lo, sample);
            const double h = h0 - pump(hi_f, hi, sample);
            const double mid = h0 - (h + l);
            h0 = h1; h1 = h2; h2 = sample;
            mine[i ^ xl] = (float)(l * g_lo + mid * g_mid + h * g_hi);
        }
    }
    __syncthreads();   // every lane has finished reading wtot / carry of this segment
    // the chunk holding the segment's last valid sample publishes the carried state
    const int jl = (nv - 1) >> LOG2L;
    if (tid == jl) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { carry[q] = lo[q]; carry[4 + q] = hi[q]; }
        carry[8] = h0; carry[9] = h1; carry[10] = h2;
    }
    // coalesced stage-out through the fused epilogue
    {
        const EqEpi E = *c.epi;
        const uint64_t t0 = c.t0; const double sr = c.sr, rsr = c.rsr;
        float* const outb = E.out + (((E.epi == 0u) || (E.flags & MX_EQF_MONO_DUP)) ? base : 2 * base);   // segment base: lane offsets stay 32-bit
        const float* const ctlb = E.ctl ? E.ctl + base : nullptr;
        const bool mono = (E.epi == 0u) || (E.flags & MX_EQF_MONO_DUP);
        auto store = [&](int i, float v) {
            typedef float __attribute__((ext_vector_type(2))) f2;
            if (c.stream_out) {                                  // long runs: the strips cannot stay cached until their consumer runs -- do not displace what can
                if (mono) __builtin_nontemporal_store(v, outb + i);
                else { f2 t = {v, v}; __builtin_nontemporal_store(t, reinterpret_cast<f2*>(outb) + i); }
            } else {
                if (mono) outb[i] = v; else reinterpret_cast<float2*>(outb)[i] = make_float2(v, v);   // stereo_panner.rs:35-38
            }
        };
        auto amp = [&](float y, double depth) { return (float)((double)y * depth * E.amp_amplitude); };   // amplifier.rs:56
        if (E.epi != 2u) {                                   // plain EqThree, or EqThree -> StereoPanner
#pragma unroll 4
            for (int k = 0; k < L; ++k) {
                const int e = tid + NT * k;
                if (e < nv) store(e, tile[swz(e)]);
            }
        } else if (E.flags & MX_EQF_ENV) {                   // ... -> Amplifier with the Envelope evaluated inline from its per-tick states
            const size_t fpc = c.fpc;
            const size_t i0 = base + (size_t)tid;            // my first sample of the segment (run-relative)
            uint32_t call = (uint32_t)(i0 / fpc); size_t rem = i0 % fpc;
            uint32_t have = 0xffffffffu; EnvTick cur{};
#pragma unroll 2
            for (int k = 0; k < L; ++k) {
                const int e = tid + NT * k;
                if (e < nv) {
                    if (call != have) { cur = E.ticks[call]; have = call; }
                    store(e, amp(tile[swz(e)], env_depth(E.env, cur, E.amp_one_minus, E.amp_mod_depth, t0 + base + e, sr, rsr)));
                }
                rem += NT;
                while (rem >= fpc) { rem -= fpc; ++call; }
            }
        } else if (E.ctl) {                                  // ... -> Amplifier, control from a buffer: bursts of <= 16 loads first
            constexpr int CB = L > 16 ? 16 : L;
#pragma unroll 1
            for (int k0 = 0; k0 < L; k0 += CB) {
                float cv[CB];
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int e = tid + NT * (k0 + k);
                    cv[k] = (e < nv) ? ctlb[e] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const int e = tid + NT * (k0 + k);
                    if (e < nv) store(e, amp(tile[swz(e)], E.amp_one_minus + E.amp_mod_depth * (double)cv[k]));
                }
            }
        } else {                                             // ... -> Amplifier with a Disconnected control: mod value 1.0 (amplifier.rs:54)
            const double depth = E.amp_one_minus + E.amp_mod_depth * 1.0;
#pragma unroll 4
            for (int k = 0; k < L; ++k) {
                const int e = tid + NT * k;
                if (e < nv) store(e, amp(tile[swz(e)], depth));
            }
        }
    }
    __syncthreads();
}

template <int LOG2L, int MODE>
__global__ __launch_bounds__(256, 4) void k_eq_three_scan(const EqDesc* __restrict__ descs, EqState* __restrict__ states, EqRun r,
                                                           const EqScanTab* __restrict__ tabs /* L = 4, 8, 16, 32 */, EqSplit sp, EqSpanPow pp) {
    const size_t frames = r.frames;
    constexpr int L = 1 << LOG2L;
    constexpr int SEG = 256 * L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EqSegCtx c;
    c.tile = reinterpret_cast<float*>(smem);                                        // 256 * L floats, swizzled (see swz)
    c.wtot = reinterpret_cast<double*>(smem + 256 * L * sizeof(float));             // [4 waves][8]
    c.carry = c.wtot + 32;                                                          // [11] lo[4] hi[4] hist[3] (+1 pad)
    c.pw = c.carry + 12;                                                            // [2][65][4] A^(L j)
    c.p2 = c.pw + 2 * 65 * 4;                                                       // [2][6][4]  A^(L 2^k)
    EqEpi* epi_lds = reinterpret_cast<EqEpi*>(c.p2 + 2 * 6 * 4);                    // fused-epilogue parameters
    c.epi = epi_lds;
    c.lo_f = r.lo_f; c.hi_f = r.hi_f; c.sr = r.sr; c.rsr = r.rsr; c.t0 = r.t0; c.fpc = r.fpc; c.stream_out = sp.stream_out != 0;

    const int tid = threadIdx.x;
    // Only what the inner phases need stays in registers; everything the epilogue needs (Amplifier and inline
    // Envelope parameters, the Envelope's state) is parked in LDS: a 160-byte descriptor held in SGPRs across the
    // segment loop overflowed the scalar file and its spills cost ~35% extra HBM traffic.
    const EqDesc* dp = descs + blockIdx.x;
    c.din = dp->in;
    c.g_lo = dp->gain_lo; c.g_mid = dp->gain_mid; c.g_hi = dp->gain_hi;
    const uint32_t span_idx = MODE == 0 ? 0u : blockIdx.y;
    if (MODE != 1 && tid == 0) *epi_lds = eq_epi_of(*dp, r.ticks ? r.ticks + (size_t)blockIdx.x * r.n_calls : nullptr);
    if (MODE == 1 && span_idx == 0) {   // snapshot of the carried state for the main pass (which overwrites it)
        if (tid < 11) sp.bound[(size_t)blockIdx.x * 12 + tid] = reinterpret_cast<const double*>(&states[blockIdx.x])[tid];
    }
    // tables and the carried state live in LDS, not in registers, across the segment loop
    eq_load_tables<LOG2L>(c, tabs, tid);
    const size_t s_end = MODE == 0 ? frames : ((size_t)(span_idx + 1) * sp.span < frames ? (size_t)(span_idx + 1) * sp.span : frames);
    const size_t s_begin = MODE == 0 ? 0 : (MODE == 1 ? s_end - sp.warm : (size_t)span_idx * sp.span);
    if (MODE == 2 && span_idx > 0) {
        // zero-state end states of the spans before mine, fetched by as many lanes as there are doubles
        double* zs = reinterpret_cast<double*>(c.tile);
        if (tid < (int)span_idx * 8) zs[tid] = sp.zbuf[(size_t)blockIdx.x * sp.n_split * 8 + tid];
        if (tid >= 248 && tid < 251) c.carry[8 + (tid - 248)] = c.din ? (double)c.din[s_begin - 3 + (tid - 248)] : 0.0;   // span >= 1024 > 3
    }
    if (tid < 11) {
        if (MODE == 0) c.carry[tid] = reinterpret_cast<const double*>(&states[blockIdx.x])[tid];   // lo[4] hi[4] history[3]
        else if (MODE == 1) c.carry[tid] = 0.0;
        else if (span_idx == 0 || tid < 8) c.carry[tid] = sp.bound[(size_t)blockIdx.x * 12 + tid];
    }
    __syncthreads();
    if (MODE == 2 && span_idx > 0) {
        if (tid == 0) {
            const double* zs = reinterpret_cast<const double*>(c.tile);
            double lo[4] = {c.carry[0], c.carry[1], c.carry[2], c.carry[3]}, hi[4] = {c.carry[4], c.carry[5], c.carry[6], c.carry[7]};
            for (uint32_t j = 0; j < span_idx; ++j) {
                double tl[4], th[4];
                toep_apply(pp.lo, lo, tl);
                toep_apply(pp.hi, hi, th);
#pragma unroll
                for (int q = 0; q < 4; ++q) { lo[q] = tl[q] + zs[j * 8 + q]; hi[q] = th[q] + zs[j * 8 + 4 + q]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { c.carry[q] = lo[q]; c.carry[4 + q] = hi[q]; }
        }
        __syncthreads();
    }

    size_t base = s_begin;
    if (LOG2L < 5 || MODE == 1) {
        for (; base < s_end; base += SEG) {
            const size_t rem = s_end - base;
            eq_segment<LOG2L, MODE>(c, tabs, base, rem < (size_t)SEG ? (int)rem : SEG);
        }
    } else {
        for (; base + SEG <= s_end; base += SEG) eq_segment<LOG2L, MODE>(c, tabs, base, SEG);
        const size_t rem = s_end - base;
        if (rem) {
            // the remainder goes out as ONE segment of the shortest chunk length that covers it: a segment costs
            // (scan + barriers) + L steps of phases A and C, so 1024 left-over samples as 256 chunks of 4 take about a
            // third of the time the same samples would take as 32 chunks of 32 with 224 lanes idle
            if (rem <= 256 * 4) { eq_load_tables<2>(c, tabs, tid); __syncthreads(); eq_segment<2, MODE>(c, tabs, base, (int)rem); }
            else if (rem <= 256 * 8) { eq_load_tables<3>(c, tabs, tid); __syncthreads(); eq_segment<3, MODE>(c, tabs, base, (int)rem); }
            else if (rem <= 256 * 16) { eq_load_tables<4>(c, tabs, tid); __syncthreads(); eq_segment<4, MODE>(c, tabs, base, (int)rem); }
            else eq_segment<5, MODE>(c, tabs, base, (int)rem);
        }
    }
    if (MODE == 1) {
        if (tid < 8) sp.zbuf[((size_t)blockIdx.x * sp.n_split + span_idx) * 8 + tid] = c.carry[tid];
        return;
    }
    if (MODE == 2 && span_idx + 1 != sp.n_split) return;   // only the last span owns the carried state
    if (tid < 11) reinterpret_cast<double*>(&states[blockIdx.x])[tid] = c.carry[tid];
}

// Short streams (the real-time mode: one tick of 800 samples per instance): a 256-thread workgroup per instance is mostly
// start-up -- tables, descriptor, four barriers for a few hundred samples -- and 10 240 instances queue up ten deep.
// Here one WAVE owns an instance (64 chunks of L samples per segment, the wave scan is the whole scan) and a workgroup
// carries four instances that share the tables.  Same arithmetic, same eq_segment.
template <int LOG2L>
__global__ __launch_bounds__(256, 4) void k_eq_three_wave(const EqDesc* __restrict__ descs, EqState* __restrict__ states, uint32_t n_inst, EqRun r,
                                                           const EqScanTab* __restrict__ tabs) {
    const size_t frames = r.frames;
    constexpr int L = 1 << LOG2L;
    constexpr int SEG = 64 * L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // shared: tables; per wave: tile, carry, epilogue parameters
    double* pw = reinterpret_cast<double*>(smem);
    double* p2 = pw + 2 * 65 * 4;
    constexpr size_t PER = (size_t)SEG * sizeof(float) + 12 * sizeof(double) + sizeof(EqEpi);
    char* mine = smem + (2 * 65 * 4 + 2 * 6 * 4) * sizeof(double) + (size_t)wv * ((PER + 15) & ~(size_t)15);
    EqSegCtx c;
    c.tile = reinterpret_cast<float*>(mine);
    c.carry = reinterpret_cast<double*>(mine + (size_t)SEG * sizeof(float));
    c.wtot = nullptr; c.pw = pw; c.p2 = p2;
    EqEpi* epi_lds = reinterpret_cast<EqEpi*>(c.carry + 12);
    c.epi = epi_lds;
    c.lo_f = r.lo_f; c.hi_f = r.hi_f; c.sr = r.sr; c.rsr = r.rsr; c.t0 = r.t0; c.fpc = r.fpc; c.stream_out = false;
    // a workgroup past the end re-does the last instance (identical values written twice) so that every wave reaches every barrier
    const uint32_t inst = min(blockIdx.x * 4u + (uint32_t)wv, n_inst - 1);
    const EqDesc* dp = descs + inst;
    c.din = dp->in;
    c.g_lo = dp->gain_lo; c.g_mid = dp->gain_mid; c.g_hi = dp->gain_hi;
    if (lane == 0) *epi_lds = eq_epi_of(*dp, r.ticks ? r.ticks + (size_t)inst * r.n_calls : nullptr);
    eq_load_tables<LOG2L>(c, tabs, (int)threadIdx.x);
    if (lane < 11) c.carry[lane] = reinterpret_cast<const double*>(&states[inst])[lane];   // lo[4] hi[4] history[3]
    __syncthreads();
    for (size_t base = 0; base < frames; base += SEG) {
        const size_t rem = frames - base;
        eq_segment<LOG2L, 0, 1>(c, tabs, base, rem < (size_t)SEG ? (int)rem : SEG);
    }
    if (lane < 11) reinterpret_cast<double*>(&states[inst])[lane] = c.carry[lane];
}

int eq_scan_log2l(size_t frames) {
    if (frames <= 256 * 4) return 2;
    if (frames <= 256 * 8) return 3;
    if (frames <= 256 * 16) return 4;
    return 5;
}

void launch_eq_three_scan(const EqDesc* d, EqState* st, uint32_t n, const EqRun& r,
                          const EqScanTab* tabs /* indexed by log2L - 2 */, const EqSplit& split_in, const EqSpanPow& pp, hipStream_t s) {
    const size_t frames = r.frames;
    if (!n || !frames) return;
    EqSplit split = split_in;
    split.stream_out = (uint64_t)n * frames * sizeof(float) > (64ull << 20) ? 1u : 0u;   // beyond what L2 + Infinity Cache keep for the consumer
    int l2 = eq_scan_log2l(split.n_split > 1 ? split.span : frames);
#define MX_EQ_GO(L2, MODE, GRID) hipLaunchKernelGGL((k_eq_three_scan<L2, MODE>), GRID, dim3(256), lds, s, d, st, r, tabs, split, pp)
#define MX_EQ_MODE(MODE, GRID) { \
        const size_t lds = (size_t)256 * (1u << l2) * sizeof(float) + (32 + 12 + 2 * 65 * 4 + 2 * 6 * 4) * sizeof(double) + sizeof(EqEpi); \
        switch (l2) { case 2: MX_EQ_GO(2, MODE, GRID); break; case 3: MX_EQ_GO(3, MODE, GRID); break; case 4: MX_EQ_GO(4, MODE, GRID); break; default: MX_EQ_GO(5, MODE, GRID); break; } }
    static const int no_wave = env_int("MX_EQ_NO_WAVE", 0);
    if (split.n_split <= 1 && frames <= 2048 && n >= 256 && !no_wave) {
        // short streams, many instances: one wave per instance (k_eq_three_wave)
        const int lw = frames <= 256 ? 2 : (frames <= 512 ? 3 : (frames <= 1024 ? 4 : 5));
        const size_t per = (((size_t)64 << lw) * sizeof(float) + 12 * sizeof(double) + sizeof(EqEpi) + 15) & ~(size_t)15;
        const size_t lds = (2 * 65 * 4 + 2 * 6 * 4) * sizeof(double) + 4 * per;
        const dim3 g((n + 3) / 4);
        switch (lw) {
        case 2: hipLaunchKernelGGL(k_eq_three_wave<2>, g, dim3(256), lds, s, d, st, n, r, tabs); break;
        case 3: hipLaunchKernelGGL(k_eq_three_wave<3>, g, dim3(256), lds, s, d, st, n, r, tabs); break;
        case 4: hipLaunchKernelGGL(k_eq_three_wave<4>, g, dim3(256), lds, s, d, st, n, r, tabs); break;
        default: hipLaunchKernelGGL(k_eq_three_wave<5>, g, dim3(256), lds, s, d, st, n, r, tabs); break;
        }
    } else if (split.n_split <= 1) {
        MX_EQ_MODE(0, dim3(n));
    } else {
        const int l2_main = l2;
        l2 = (int)split.l2_pre;                              // the pre-pass has its own (smaller) segment size
        MX_EQ_MODE(1, dim3(n, split.n_split - 1));
        l2 = l2_main;
        MX_EQ_MODE(2, dim3(n, split.n_split));
    }
#undef MX_EQ_MODE
#undef MX_EQ_GO
}

// how many trailing samples of a span decide its zero-state end state: the 4-pole cascade's response to a sample k
// steps back is at most C(k+3,3) p^k (p = 1 - f, the pole; all section gains f <= 1), so everything older than K adds
// less than tail(K) = sum_{k>=K} C(k+3,3) p^k per unit of input.  K is the first multiple of 128 where tail(K) < 2^-280
// (48 kHz: 3968, 44.1 kHz: 3584, 96 kHz: 8064, 192 kHz: 16384 samples).
static size_t eq_forget_len(double f) {
    const long double p = 1.0L - (long double)f;
    if (!(p > 0.0L) || !(p < 1.0L)) return p <= 0.0L ? 8 : (size_t)-1;
    const long double lim = ldexpl(1.0L, -280);
    for (size_t K = 128; K <= ((size_t)1 << 22); K += 128) {
        // tail(K) <= C(K+3,3) p^K * sum_j ((K+4)/(K+1) p)^j : ratio of consecutive terms is (k+4)/(k+1) p, decreasing in k
        const long double r = (long double)(K + 4) / (long double)(K + 1) * p;
        if (r >= 1.0L) continue;
        const long double c = (long double)(K + 3) * (long double)(K + 2) * (long double)(K + 1) / 6.0L;
        const long double t = c * expl((long double)K * logl(p)) / (1.0L - r);
        if (t < lim) return K;
    }
    return (size_t)-1;
}

// span plan for n instances of `frames` samples: enough workgroups to cover the chip
void eq_plan_split(uint32_t n, size_t frames, double lo_f, double hi_f, EqSplit& out) {
    out.n_split = 1; out.span = frames; out.warm = frames; out.l2_pre = 5;
    const int force = env_int("MX_EQ_SPLIT", 0);          // tuning / test override (read per call: tests flip it)
    uint32_t want = force > 0 ? (uint32_t)force : (n >= 1024 ? 1u : 1024u / (n ? n : 1));   // 4 workgroups per CU, one round
    if (want > 32) want = 32;                             // the main pass's prologue folds <= 31 spans with 248 lanes
    if (want < 2) return;
    // pre-pass window: whole segments of the smallest segment size that covers the forgetting length
    const size_t K = env_int("MX_EQ_FULL_PREPASS", 0) ? (size_t)-1 : std::max(eq_forget_len(lo_f), eq_forget_len(hi_f));
    size_t warm = (size_t)-1; int l2p = 5;
    if (K != (size_t)-1) {
        l2p = 3;
        while (l2p < 5 && ((size_t)256 << l2p) < K) ++l2p;
        const size_t sgp = (size_t)256 << l2p;
        warm = (K + sgp - 1) / sgp * sgp;
    }
    // windowed pre-pass: spans are multiples of 1024 samples (the remainder after whole 8192-sample segments goes out
    // as one short-chunk segment); full pre-pass: spans are whole 8192-sample segments
    size_t sp = (frames + want - 1) / want;
    const size_t unit = 1024;
    size_t spw = (sp + unit - 1) / unit * unit;
    if (warm != (size_t)-1 && warm < spw && spw < frames) {
        out.span = spw; out.warm = warm; out.l2_pre = (uint32_t)l2p;
        out.n_split = (uint32_t)((frames + spw - 1) / spw);
        return;
    }
    const size_t seg = (size_t)256 << 5;
    sp = (sp + seg - 1) / seg * seg;
    if (sp >= frames) return;                             // stream too short to cut
    out.span = sp; out.warm = sp; out.l2_pre = 5;
    out.n_split = (uint32_t)((frames + sp - 1) / sp);
}

}  // namespace mx
