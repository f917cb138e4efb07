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
#define FIR_PER 4                          // consecutive outputs per lane
#define FIR_TILE (FIR_BLOCK * FIR_PER)     // outputs per block step
// out[n] = (f32) sum_k h[k] * (f64) x[n-k] per channel; x[m<0] comes from the carried history.
//
// The prescribed work is 4 f64 operations per tap and stereo frame (2 mul + 2 add, no FMA by spec).  With one output
// per lane every tap step also needs 16 B of LDS per lane, and LDS (128 B/clk/CU) feeds only half of what the four
// SIMDs can multiply -- measured 0.475 of the f64 rate, exactly that roof.  So a lane owns FOUR consecutive outputs:
// their inputs are a sliding window in registers, one new frame per tap step serves all four, LDS traffic drops 4x.
// Lanes then read frames 4 apart; the window is stored with one pad frame after every 4 (p = f + f/4): a stride of
// 5 frames = 20 banks, and the 16 lanes a ds_read_b128 serves per pass cover all 64 banks exactly once.
// Frames are widened to f64 once while staging (exact); taps sit in LDS (uniform reads broadcast).
__device__ __forceinline__ int fir_pad(int f) { return f + (f >> 2); }
template <bool FC>
__global__ __launch_bounds__(FIR_BLOCK) void k_fir(const FirDesc* __restrict__ descs, size_t frames) {
    const FirDesc d = descs[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = (int)d.n_taps;
    double* tapl = reinterpret_cast<double*>(smem);                       // [K rounded up to even]
    double2* win = reinterpret_cast<double2*>(tapl + ((K + 1) & ~1));     // padded window of FIR_TILE + K - 1 frames
    for (int k = threadIdx.x; k < K; k += FIR_BLOCK) tapl[k] = d.taps[k];
    // FIR_PER zero frames in front of the window: the frame that "enters" after the last tap of the first output is
    // never used, and with the pad its read needs no guard -- the tap loop is branch-free
    if (threadIdx.x < FIR_PER) win[fir_pad((int)threadIdx.x)] = make_double2(0.0, 0.0);
    const int WN = FIR_TILE + K - 1;
    for (size_t blk = (size_t)blockIdx.x * FIR_TILE; blk < frames; blk += (size_t)gridDim.x * FIR_TILE) {
        // window frame w <-> stream frame blk - (K-1) + w
        for (int w = threadIdx.x; w < WN; w += FIR_BLOCK) {
            const long long f = (long long)blk + w - (K - 1);
            float2 v = make_float2(0.f, 0.f);
            if (f >= 0) { if ((size_t)f < frames && d.in) v = reinterpret_cast<const float2*>(d.in)[f]; }
            else { const long long h = (long long)(K - 1) + f; if (h >= 0) v = d.hist[h]; }   // hist[j] = x[j - (K-1)]
            win[fir_pad(w + FIR_PER)] = make_double2((double)v.x, (double)v.y);
        }
        __syncthreads();
        const int o = FIR_PER * (int)threadIdx.x;                         // my first output inside the tile
        if (blk + o < frames) {
            double al[FIR_PER], ar[FIR_PER];
            double2 w[FIR_PER];                                           // w[j] = x[o + j - k] for the current tap k
            // o is a multiple of 4, so fir_pad(o + m) = fir_pad(o) + fir_pad(m): one per-lane base, wave-uniform (scalar) offsets
            const char* lane_base = reinterpret_cast<const char*>(win) + (size_t)fir_pad(o) * sizeof(double2);
            auto rd = [&](int m) { return *reinterpret_cast<const double2*>(lane_base + (size_t)fir_pad(m) * sizeof(double2)); };
#pragma unroll
            for (int j = 0; j < FIR_PER; ++j) { al[j] = 0.0; ar[j] = 0.0; w[j] = rd(j + (K - 1) + FIR_PER); }
            int k = 0;
#pragma unroll 2
            for (; k + FIR_PER <= K; k += FIR_PER) {
#pragma unroll
                for (int u = 0; u < FIR_PER; ++u) {                       // tap k + u: the window has slid u frames; slot names rotate, nothing moves
                    const double h = tapl[k + u];
#pragma unroll
                    for (int j = 0; j < FIR_PER; ++j) {
                        const double2 x = w[(j - u + FIR_PER) % FIR_PER];
                        al[j] = mul_add<FC>(h, x.x, al[j]);
                        ar[j] = mul_add<FC>(h, x.y, ar[j]);
                    }
                    // frame x[o - 1 - (k + u)] enters; it replaces the slot of x[o + 3 - (k + u)], which no later tap needs
                    w[(FIR_PER - 1 - u + FIR_PER) % FIR_PER] = rd((K - 1) - 1 - (k + u) + FIR_PER);
                }
            }
            for (; k < K; ++k) {                                          // K not a multiple of 4: rotate by moving
                const double h = tapl[k];
#pragma unroll
                for (int j = 0; j < FIR_PER; ++j) { al[j] = mul_add<FC>(h, w[j].x, al[j]); ar[j] = mul_add<FC>(h, w[j].y, ar[j]); }
#pragma unroll
                for (int j = FIR_PER - 1; j > 0; --j) w[j] = w[j - 1];
                w[0] = rd((K - 1) - 1 - k + FIR_PER);
            }
            float2* out = reinterpret_cast<float2*>(d.out) + blk + o;
#pragma unroll
            for (int j = 0; j < FIR_PER; ++j) if (blk + o + j < frames) out[j] = make_float2((float)al[j], (float)ar[j]);
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
    const size_t wn = (size_t)FIR_TILE + max_taps + FIR_PER;
    const size_t lds = (size_t)((max_taps + 1) & ~1u) * sizeof(double) + (wn + wn / 4 + 2) * sizeof(double2);
    dim3 grid(grid_x(frames, FIR_TILE, 1024), n);
    if (fc) hipLaunchKernelGGL(k_fir<true>, grid, dim3(FIR_BLOCK), lds, s, d, frames);
    else hipLaunchKernelGGL(k_fir<false>, grid, dim3(FIR_BLOCK), lds, s, d, frames);
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
                     size_t in_frames, size_t out_frames, uint64_t in_base, uint64_t out_base, hipStream_t s, uint32_t common_up, bool fc) {
    if (!n || !out_frames) return;
    const size_t lds = (size_t)tab_doubles * sizeof(double) + (size_t)((win_frames + 1u) & ~1u) * 2 * sizeof(double);
    if (lds <= 60 * 1024) {
        // few blocks per channel, each walking many 256-output groups (the table is loaded once per block): as many blocks as the chip
        // holds at once, so they all start together, do the same work and end together -- no partly filled last round
        static const uint32_t cus = [] { int dev = 0, n_cu = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256; return (uint32_t)std::max(n_cu, 1); }();
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
