// mx_k_fir.hip -- BUILD-SPECIFIED modules with no reference counterpart (BASELINE.json configs[2];
// the reference only has a `TODO implement resampling`, src/icecast/mod.rs:94-97):
//   Fir       128-tap (any length) FIR "reverb" on an interleaved stereo stream
//   Resample  rational polyphase resampler (44.1 -> 48 kHz is up 160 / down 147)
// Arithmetic follows the reference's own convention for audio (mixer.rs:62, eq_three.rs:85,
// amplifier.rs:56): widen f32 to f64, accumulate in f64 in ASCENDING tap index with separate
// multiply and add (no FMA), round once to f32.  Bit-exact against the oracle; parity unpinned.
// FC (MX_FLAG_FP_CONTRACT): the accumulation as acc = fma(h[k], x, acc) in the same ascending order -- half the f64 instructions;
// equal to the oracle's contract mode bit for bit, within 1 ULP of the separate multiply-and-add spec.
//
// Every output sample is independent given the input history, so both kernels are plain
// data-parallel: a 256-lane block stages its input window in LDS (coalesced), taps are wave-uniform.
#include <algorithm>

#include "mx_dev.hpp"
#include "mx_env_math.hpp"   // mul_add<FC>

namespace mx {

#define FIR_BLOCK 256
// out[n] = (f32) sum_k h[k] * (f64) x[n-k] per channel; x[m<0] comes from the carried history.
//
// The prescribed work is 4 f64 operations per tap and stereo frame (2 mul + 2 add, no FMA by spec).  With one output
// per lane every tap step also needs 16 B of LDS per lane, and LDS (128 B/clk/CU) feeds only half of what the four
// SIMDs can multiply -- measured 0.475 of the f64 rate, exactly that roof.  So a lane owns PER consecutive outputs (4, or 8:
// launch_fir): their inputs are a sliding window in registers, one new frame per tap step serves all of them, LDS traffic drops
// PER-fold, and the loop's own instructions (a tap's coefficient read, the LDS offsets) are shared by PER x 4 f64 operations.
// Lanes then read frames PER apart; the window is stored with one pad frame after every PER (p = f + f / PER): a stride of
// PER + 1 frames = 20 (36) banks, and the 16 lanes a ds_read_b128 serves per pass cover all 64 banks exactly once.
// Frames are widened to f64 once while staging (exact); taps sit in LDS (uniform reads broadcast).
template <int PER> __device__ __forceinline__ int fir_pad(int f) { return f + (f / PER); }   // f >= 0; PER a power of two: a shift
template <int PER, bool FC>
__global__ __launch_bounds__(FIR_BLOCK) void k_fir(const FirDesc* __restrict__ descs, size_t frames) {
    constexpr int TILE = FIR_BLOCK * PER;                                 // outputs per block step
    const FirDesc d = descs[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = (int)d.n_taps;
    double* tapl = reinterpret_cast<double*>(smem);                       // [K rounded up to even]
    double2* win = reinterpret_cast<double2*>(tapl + ((K + 1) & ~1));     // padded window of TILE + K - 1 frames
    for (int k = threadIdx.x; k < K; k += FIR_BLOCK) tapl[k] = d.taps[k];
    // PER zero frames in front of the window: the frame that "enters" after the last tap of the first output is
    // never used, and with the pad its read needs no guard -- the tap loop is branch-free
    if ((int)threadIdx.x < PER) win[fir_pad<PER>((int)threadIdx.x)] = make_double2(0.0, 0.0);
    const int WN = TILE + K - 1;
    for (size_t blk = (size_t)blockIdx.x * TILE; blk < frames; blk += (size_t)gridDim.x * TILE) {
        // window frame w <-> stream frame blk - (K-1) + w
        for (int w = threadIdx.x; w < WN; w += FIR_BLOCK) {
            const long long f = (long long)blk + w - (K - 1);
            float2 v = make_float2(0.f, 0.f);
            if (f >= 0) { if ((size_t)f < frames && d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long h = (long long)(K - 1) + f; if (h >= 0) v = d.hist[h]; }   // hist[j] = x[j - (K-1)]
            win[fir_pad<PER>(w + PER)] = make_double2((double)v.x, (double)v.y);
        }
        __syncthreads();
        const int o = PER * (int)threadIdx.x;                             // my first output inside the tile
        if (blk + o < frames) {
            double al[PER], ar[PER];
            double2 w[PER];                                               // w[j] = x[o + j - k] for the current tap k
            // o is a multiple of PER, so fir_pad(o + m) = fir_pad(o) + fir_pad(m): one per-lane base, wave-uniform (scalar) offsets
            const char* lane_base = reinterpret_cast<const char*>(win) + (size_t)fir_pad<PER>(o) * sizeof(double2);
            auto rd = [&](int m) { return *reinterpret_cast<const double2*>(lane_base + (size_t)fir_pad<PER>(m) * sizeof(double2)); };
#pragma unroll
            for (int j = 0; j < PER; ++j) { al[j] = 0.0; ar[j] = 0.0; w[j] = rd(j + (K - 1) + PER); }
            int k = 0;
#pragma unroll 2
            for (; k + PER <= K; k += PER) {
#pragma unroll
                for (int u = 0; u < PER; ++u) {                           // tap k + u: the window has slid u frames; slot names rotate, nothing moves
                    const double h = tapl[k + u];
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        const double2 x = w[(j - u + PER) % PER];
                        al[j] = mul_add<FC>(h, x.x, al[j]);
                        ar[j] = mul_add<FC>(h, x.y, ar[j]);
                    }
                    // frame x[o - 1 - (k + u)] enters; it replaces the slot of x[o + PER - 1 - (k + u)], which no later tap needs
                    w[(PER - 1 - u + PER) % PER] = rd((K - 1) - 1 - (k + u) + PER);
                }
            }
            for (; k < K; ++k) {                                          // K not a multiple of PER: rotate by moving
                const double h = tapl[k];
#pragma unroll
                for (int j = 0; j < PER; ++j) { al[j] = mul_add<FC>(h, w[j].x, al[j]); ar[j] = mul_add<FC>(h, w[j].y, ar[j]); }
#pragma unroll
                for (int j = PER - 1; j > 0; --j) w[j] = w[j - 1];
                w[0] = rd((K - 1) - 1 - k + PER);
            }
            float2* out = reinterpret_cast<float2*>(d.out) + blk + o;
#pragma unroll
            for (int j = 0; j < PER; ++j) if (blk + o + j < frames) out[j] = make_float2((float)al[j], (float)ar[j]);
        }
        __syncthreads();
    }
}
// new history = the last K-1 input frames of (old history ++ input)
__global__ __launch_bounds__(256) void k_fir_history(const FirDesc* __restrict__ descs, size_t frames) {
    const FirDesc d = descs[blockIdx.x];
    const int H = (int)d.n_taps - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tmp = reinterpret_cast<float2*>(smem);
    for (int j = threadIdx.x; j < H; j += 256) {
        const long long f = (long long)frames - H + j;           // index into the input stream
        float2 v = make_float2(0.f, 0.f);
        if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
        else { const long long h = (long long)H + f; if (h >= 0) v = d.hist[h]; }
        tmp[j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += 256) d.hist[j] = tmp[j];
}
void launch_fir(const FirDesc* d, uint32_t n, uint32_t max_taps, size_t frames, hipStream_t s, bool fc) {
    if (!n || !frames) return;
    // outputs per lane: 8 where the streams are long enough to fill the chip with tiles of 2048 outputs (and the window fits LDS), else 4
    static const int force_per = env_int("MX_FIR_PER", 0);
    const int per = force_per == 4 || force_per == 8 ? force_per : ((frames >= 8192 && max_taps <= 1024) ? 8 : 4);
    const size_t tile = (size_t)FIR_BLOCK * per;
    const size_t wn = tile + max_taps + per;
    const size_t lds = (size_t)((max_taps + 1) & ~1u) * sizeof(double) + (wn + wn / per + 2) * sizeof(double2);
    dim3 grid(grid_x(frames, (unsigned)tile, 1024), n);
#define MX_FIR_GO(P, F) hipLaunchKernelGGL((k_fir<P, F>), grid, dim3(FIR_BLOCK), lds, s, d, frames)
    if (per == 8) { if (fc) MX_FIR_GO(8, true); else MX_FIR_GO(8, false); }
    else { if (fc) MX_FIR_GO(4, true); else MX_FIR_GO(4, false); }
#undef MX_FIR_GO
    hipLaunchKernelGGL(k_fir_history, dim3(n), dim3(256), max_taps * sizeof(float2), s, d, frames);
}

// Rational resampler.  Output sample m (absolute index M = out_base + m):
//   n = floor(M * down / up), phase = (M * down) mod up
//   y[m] = (f32) sum_{k < P} h[phase][k] * (f64) x[n - k]      (x indexed absolutely; x before the run from history)
//
// k_resample: a block owns 256 consecutive outputs.  Their inputs are one contiguous window (256 * down / up + P
// frames), staged in LDS with coalesced loads; the polyphase table sits in LDS TRANSPOSED ([k][phase]: consecutive
// outputs walk distinct phases, so a tap step reads 64 different doubles, at most 2 per bank pair, instead of 64
// rows of one 128-byte-strided column); the 64-bit division M * down / up is done once per block, each lane only
// divides a 32-bit offset.  A block walks many 256-output groups so the table is loaded once per ~50 groups.
// k_resample_gather is the plain form for tables that do not fit LDS.
// UPC: the interpolation factor when the launcher knows every channel of the launch has that one (160: the 44.1 -> 48 kHz ratio), else 0.  With it the
// table's row stride is a constant and a tap's coefficient read carries its row as an instruction immediate; at run-time stride every tap costs an
// address add per lane (16 of a lane's ~124 VALU instructions per output).
template <int UPC, bool FC>
__global__ __launch_bounds__(256) void k_resample(const ResampleDesc* __restrict__ descs, size_t out_frames,
                                                  uint64_t out_base, uint64_t in_base, uint32_t win_cap) {
    const ResampleDesc d = descs[blockIdx.y];
    const int P = (int)d.taps_per_phase, H = P - 1;
    const uint32_t up = UPC ? (uint32_t)UPC : d.up, down = d.down;
    // floor(q / up) for the lanes' 32-bit offsets by a reciprocal made once per block: mulhi gives the quotient or one less (q M / 2^32 > q / up - 1)
    const uint32_t up_magic = up > 1u ? (uint32_t)((1ull << 32) / up) : 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);                       // [P][up]
    // The window as TWO planes of doubles (left, right), widened once while staging.  As interleaved double2 a tap step is one
    // ds_read_b128 per lane, and that instruction serves the wave in the lane groups {0-3, 12-15, 20-27}, ...: with lanes on
    // (nearly) consecutive frames two lanes of a group meet on a bank and every read costs twice (profiles/r03: SQ_LDS_BANK_CONFLICT
    // 2x the LDS-active cycles).  ds_read_b64 serves lanes 0-31 and 32-63 as they are: 32 (nearly) consecutive doubles cover the 64
    // banks at most once -- two conflict-free 2-cycle reads instead of one 8-cycle one.
    const uint32_t cap = (win_cap + 1u) & ~1u;
    double* xl = tab + (size_t)P * up;                                   // [cap]
    double* xr = xl + cap;                                               // [cap]
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)P * up; i += 256) {
        const uint32_t ph = i / (uint32_t)P, k = i - ph * (uint32_t)P;
        tab[(size_t)k * up + ph] = d.taps[i];
    }
    const bool small = (uint64_t)255 * down + up < (1ull << 32);          // lane offsets fit 32 bits (always, for audio ratios)
    // geometry of the group of 256 outputs from `blk` on: the window's first input frame, its length, the first output's remainder
    struct Grp { long long f0; uint32_t cnt, r0; };
    auto group_of = [&](size_t blk) {
        Grp g;
        const uint64_t num0 = (out_base + blk) * down;
        const uint64_t n0_abs = num0 / up;                                // block-uniform
        g.r0 = (uint32_t)(num0 - n0_abs * up);
        const uint32_t last = (uint32_t)min((size_t)255, out_frames - 1 - blk);
        const uint32_t span = (uint32_t)((g.r0 + (uint64_t)last * down) / up);   // n of the last output relative to n0
        g.f0 = (long long)(n0_abs - in_base) - H;                         // input index of window frame 0
        g.cnt = min(span + 1u + (uint32_t)H, win_cap);
        return g;
    };
    // a window is at most 256 * down / up + P frames: up to `per` frames per lane (1 for upsampling ratios and P <= 256 * (1 - down / up))
    auto fetch = [&](const Grp& g, uint32_t w) {
        const long long f = g.f0 + w;
        float2 v = make_float2(0.f, 0.f);
        if (w < g.cnt) {
            if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long hh = (long long)H + f; if (hh >= 0) v = d.hist[hh]; }
        }
        return v;
    };
    const bool one = win_cap <= 256u;                                     // the whole window is one frame per lane: prefetched in a register
    size_t blk = (size_t)blockIdx.x * 256;
    Grp g = blk < out_frames ? group_of(blk) : Grp{0, 0u, 0u};
    float2 pre = (one && blk < out_frames) ? fetch(g, tid) : make_float2(0.f, 0.f);
    for (; blk < out_frames; blk += (size_t)gridDim.x * 256) {
        if (one) { if ((uint32_t)tid < g.cnt) { xl[tid] = (double)pre.x; xr[tid] = (double)pre.y; } }
        else for (uint32_t w = tid; w < g.cnt; w += 256) { const float2 v = fetch(g, w); xl[w] = (double)v.x; xr[w] = (double)v.y; }
        __syncthreads();
        const uint32_t r0 = g.r0;
        // the next group's frames are requested now and arrive while this group's taps run: a block pays the round trip to memory once, not per group
        const size_t nxt = blk + (size_t)gridDim.x * 256;
        if (nxt < out_frames) { g = group_of(nxt); if (one) pre = fetch(g, tid); }
        if (blk + tid < out_frames) {
            uint32_t dn, phase;
            if (small) {
                const uint32_t q = r0 + (uint32_t)tid * down;
                dn = __umulhi(q, up_magic); phase = q - dn * up;
                if (phase >= up) { ++dn; phase -= up; }
            }
            else { const uint64_t q = r0 + (uint64_t)tid * down; dn = (uint32_t)(q / up); phase = (uint32_t)(q - (uint64_t)dn * up); }
            // volatile: each read stays a ds_read_b64 of its own (merged into ds_read2_b64 pairs they are served 16 lanes at a time)
            typedef const volatile __attribute__((address_space(3))) double* LdsD;
            const LdsD l = (LdsD)(xl + H + dn);                           // l[0] = frame n of this output
            const LdsD r = (LdsD)(xr + H + dn);
            const LdsD h = (LdsD)(tab + phase);
            double al = 0.0, ar = 0.0;
            for (int k = 0; k < P; ++k) {
                const double c = UPC ? h[(size_t)k * UPC] : h[(size_t)k * up];
                const double vl = l[-k], vr = r[-k];
                al = mul_add<FC>(c, vl, al);
                ar = mul_add<FC>(c, vr, ar);
            }
            reinterpret_cast<float2*>(d.out)[blk + tid] = make_float2((float)al, (float)ar);
        }
        __syncthreads();
    }
}
// k_resample_ps ("phase-stationary"): the form for a launch whose channels all share the ratio UP / down and P taps per phase (config 3: 160 / 147,
// 16 taps).  Output M + UP has the phase of output M, so when a block's groups of 256 outputs are a multiple of UP outputs apart (the launcher
// makes the blocks per channel a multiple of UP / gcd(UP, 256)) every lane keeps ONE phase for the whole launch: its P coefficients are loaded
// into registers once -- no table in LDS (20 KiB per block in k_resample) and no coefficient read per tap: two 8-byte LDS reads per four f64
// operations instead of three (profiles/r03: the kernel waited on LDS issue and latency, at 0.245 of the f64 rate and 0.29 of HBM).  The offset of a
// lane's first input frame inside the window is loop-invariant for the same reason.  With 8 KiB of LDS per block the window is double-buffered:
// one barrier per group instead of two, the next group's frames in flight while this group's taps run.
template <int UP, int P, bool FC>
__global__ __launch_bounds__(256) void k_resample_ps(const ResampleDesc* __restrict__ descs, size_t out_frames, uint64_t out_base, uint64_t in_base) {
    const ResampleDesc d = descs[blockIdx.y];
    constexpr int H = P - 1;
    constexpr uint32_t CAP = 256;                                         // window frames per group: 255 * down / UP + 2 + P <= 256 (launcher)
    const uint32_t down = d.down;
    __shared__ double win[2][2][CAP];                                     // [buffer][left, right][frame], widened once while staging
    const int tid = threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;                        // a multiple of UP outputs: phases and offsets repeat
    size_t blk = (size_t)blockIdx.x * 256;
    if (blk >= out_frames) return;
    // the block's first output: n0 = floor(M0 * down / UP), r0 = (M0 * down) mod UP -- r0 is the same for every group of this block
    const uint64_t num0 = (out_base + blk) * down;
    const uint64_t n0_abs = num0 / UP;
    const uint32_t r0 = (uint32_t)(num0 - n0_abs * UP);
    const uint32_t q = r0 + (uint32_t)tid * down;                         // < UP + 255 * down: 32 bits for every audio ratio (launcher)
    const uint32_t dn = q / UP, phase = q - dn * UP;                      // my input frame relative to the group's first, my phase: loop-invariant
    double c[P];
#pragma unroll
    for (int k = 0; k < P; ++k) c[k] = d.taps[(size_t)phase * P + k];
    const uint64_t in_step = (uint64_t)(stride / UP) * down;              // input frames between two groups of this block (exact: stride % UP == 0)
    long long f0 = (long long)(n0_abs - in_base) - H;                    // input index of window frame 0 of the current group
    const uint32_t cnt_full = min((uint32_t)((r0 + 255ull * down) / UP) + 1u + (uint32_t)H, CAP);   // window frames of a full group (loop-invariant)
    const float2* __restrict__ in2 = reinterpret_cast<const float2*>(d.in);
    auto fetch = [&](long long base_f, size_t g_blk) {
        // an interior group (block-uniform test): every frame of its window exists in this run's input -- one guarded load, no geometry
        if (base_f >= 0 && out_frames - g_blk >= 256 && in2) return (uint32_t)tid < cnt_full ? (in2 + base_f)[tid] : make_float2(0.f, 0.f);
        uint32_t cnt = cnt_full;
        if (out_frames - g_blk < 256) cnt = min((uint32_t)((r0 + (uint64_t)(out_frames - 1 - g_blk) * down) / UP) + 1u + (uint32_t)H, CAP);   // the stream's last, partial group
        const long long f = base_f + tid;
        float2 v = make_float2(0.f, 0.f);
        if ((uint32_t)tid < cnt) {
            if (f >= 0) { if (in2) v = in2[f]; }
            else { const long long hh = (long long)H + f; if (hh >= 0) v = d.hist[hh]; }
        }
        return v;
    };
    // the frames of the next TWO groups are in flight while this group's taps run: one group's taps (a few hundred cycles) do not cover a round trip to memory
    float2 pre = fetch(f0, blk), pre2 = make_float2(0.f, 0.f);
    if (blk + stride < out_frames) pre2 = fetch(f0 + (long long)in_step, blk + stride);
    f0 += 2 * (long long)in_step;                                         // window start of the group two ahead
    int buf = 0;
    for (; blk < out_frames; blk += stride, buf ^= 1) {
        win[buf][0][tid] = (double)pre.x; win[buf][1][tid] = (double)pre.y;
        __syncthreads();      // (the one barrier: a wave writes buffer b again two groups later, after every wave has passed the barrier in between)
        pre = pre2;
        const size_t nxt2 = blk + 2 * stride;
        if (nxt2 < out_frames) pre2 = fetch(f0, nxt2);
        f0 += (long long)in_step;
        if (blk + tid < out_frames) {
            // volatile: each read stays a ds_read_b64 of its own (merged into ds_read2_b64 pairs they are served 16 lanes at a time)
            typedef const volatile __attribute__((address_space(3))) double* LdsD;
            const LdsD l = (LdsD)(&win[buf][0][H + dn]);                  // l[0] = frame n of this output
            const LdsD r = (LdsD)(&win[buf][1][H + dn]);
            double al = 0.0, ar = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const double vl = l[-k], vr = r[-k];
                al = mul_add<FC>(c[k], vl, al);
                ar = mul_add<FC>(c[k], vr, ar);
            }
            reinterpret_cast<float2*>(d.out)[blk + tid] = make_float2((float)al, (float)ar);
        }
    }
}
template <bool FC>
__global__ __launch_bounds__(256) void k_resample_gather(const ResampleDesc* __restrict__ descs, size_t out_frames,
                                                         uint64_t out_base, uint64_t in_base) {
    const ResampleDesc d = descs[blockIdx.y];
    const int P = (int)d.taps_per_phase, H = P - 1;
    for (size_t m = (size_t)blockIdx.x * 256 + threadIdx.x; m < out_frames; m += (size_t)gridDim.x * 256) {
        const uint64_t M = out_base + m;
        const uint64_t num = M * d.down;
        const uint64_t n_abs = num / d.up;
        const uint32_t phase = (uint32_t)(num - n_abs * d.up);
        const long long n = (long long)(n_abs - in_base);            // index into this run's input
        const double* __restrict__ h = d.taps + (size_t)phase * P;
        double al = 0.0, ar = 0.0;
        for (int k = 0; k < P; ++k) {
            const long long f = n - k;
            float2 v = make_float2(0.f, 0.f);
            if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long hh = (long long)H + f; if (hh >= 0) v = d.hist[hh]; }
            al = mul_add<FC>(h[k], (double)v.x, al);
            ar = mul_add<FC>(h[k], (double)v.y, ar);
        }
        reinterpret_cast<float2*>(d.out)[m] = make_float2((float)al, (float)ar);
    }
}
__global__ __launch_bounds__(256) void k_resample_history(const ResampleDesc* __restrict__ descs, size_t in_frames) {
    const ResampleDesc d = descs[blockIdx.x];
    const int H = (int)d.taps_per_phase - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tmp = reinterpret_cast<float2*>(smem);
    for (int j = threadIdx.x; j < H; j += 256) {
        const long long f = (long long)in_frames - H + j;
        float2 v = make_float2(0.f, 0.f);
        if (f >= 0) { if (d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
        else { const long long h = (long long)H + f; if (h >= 0) v = d.hist[h]; }
        tmp[j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += 256) d.hist[j] = tmp[j];
}
void launch_resample(const ResampleDesc* d, uint32_t n, uint32_t max_taps, uint32_t tab_doubles, uint32_t win_frames,
                     size_t in_frames, size_t out_frames, uint64_t in_base, uint64_t out_base, hipStream_t s, uint32_t common_up, bool fc, uint32_t common_taps, uint32_t common_down) {
    if (!n || !out_frames) return;
    static const uint32_t cus = [] { int dev = 0, n_cu = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256; return (uint32_t)std::max(n_cu, 1); }();
    static const int no_ps = env_int("MX_RESAMPLE_PS", 1) == 0;           // A/B: the staged kernel everywhere
    // every channel 160 / down with 16 taps per phase, a window that fits 256 frames: lanes keep their phase (k_resample_ps)
    if (!no_ps && common_up == 160u && common_taps == 16u && common_down && (uint64_t)255 * common_down / 160u + 2u + 16u <= 256u && (uint64_t)160u + 255ull * common_down < (1ull << 32)) {
        const uint32_t m = 5;                                             // 160 / gcd(160, 256): blocks per channel come in multiples of it
        const size_t groups = (out_frames + 255) / 256;
        const uint32_t resident = cus * 8u;                               // 8 KiB of LDS and 4 waves per block: the 32-wave limit decides
        uint32_t per_ch = std::max<uint32_t>(1u, resident / n) / m * m;
        if (per_ch == 0) per_ch = m;
        per_ch = (uint32_t)std::min<size_t>(per_ch, (groups + m - 1) / m * m);
        // (the barrier-free form -- every wave staging the window of its own 64 outputs, workgroups of one wave -- was built twice, rounds 4 and 5, bit-exact, and lost
        // both times: 0.21 / 0.17-0.19 ms against 0.15.  The waves' 55 % of parked cycles are not the barrier's: profiles/r05/fir_sq3.txt)
        if (fc) hipLaunchKernelGGL((k_resample_ps<160, 16, true>), dim3(per_ch, n), dim3(256), 0, s, d, out_frames, out_base, in_base);
        else hipLaunchKernelGGL((k_resample_ps<160, 16, false>), dim3(per_ch, n), dim3(256), 0, s, d, out_frames, out_base, in_base);
        hipLaunchKernelGGL(k_resample_history, dim3(n), dim3(256), (max_taps + 1) * sizeof(float2), s, d, in_frames);
        return;
    }
    const size_t lds = (size_t)tab_doubles * sizeof(double) + (size_t)((win_frames + 1u) & ~1u) * 2 * sizeof(double);
    if (lds <= 60 * 1024) {
        // few blocks per channel, each walking many 256-output groups (the table is loaded once per block): as many blocks as the chip
        // holds at once, so they all start together, do the same work and end together -- no partly filled last round
        const uint32_t resident = cus * (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
        const uint32_t per_ch = (uint32_t)std::max<size_t>(1, std::min<size_t>((out_frames + 255) / 256, std::max<uint32_t>(1u, resident / n)));
#define MX_RS_GO(U, F) hipLaunchKernelGGL((k_resample<U, F>), dim3(per_ch, n), dim3(256), lds, s, d, out_frames, out_base, in_base, win_frames)
        if (common_up == 160u) { if (fc) MX_RS_GO(160, true); else MX_RS_GO(160, false); }
        else { if (fc) MX_RS_GO(0, true); else MX_RS_GO(0, false); }
#undef MX_RS_GO
    } else {
        if (fc) hipLaunchKernelGGL(k_resample_gather<true>, dim3(grid_x(out_frames, 256, 1024), n), dim3(256), 0, s, d, out_frames, out_base, in_base);
        else hipLaunchKernelGGL(k_resample_gather<false>, dim3(grid_x(out_frames, 256, 1024), n), dim3(256), 0, s, d, out_frames, out_base, in_base);
    }
    hipLaunchKernelGGL(k_resample_history, dim3(n), dim3(256), (max_taps + 1) * sizeof(float2), s, d, in_frames);
}

}  // namespace mx
