// mx_k_stream.hip -- streaming kernels: Amplifier, Oscillator, FmSine, Trigger, StereoPanner, StereoSplitter, Plotter.
//
// Build with -ffp-contract=off: the reference (Rust) evaluates every f64 expression as written,
// never fused; parity with it is bit-exact only if v_fma_f64 is not substituted for mul+add.
//
// Layout: every port buffer is a flat f32 stream of `frames` mono samples (or 2*frames interleaved
// L,R) -- n_ticks consecutive 735/800-sample tick buffers back to back -- 256-byte aligned.
// Instances of one module kind are batched into one launch.
#include "mx_dev.hpp"
#include "mx_env_math.hpp"
#include "mx_sin_f32.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// Amplifier (src/module/amplifier.rs:38-60,71-73)
//   out[i] = (in[i] as f64 * (1.0 - d + d * mod[i/2]) * amplitude) as f32
// algorithmic bytes per frame: 8 (in) + 4 (ctl) + 8 (out) = 20
// ---------------------------------------------------------------------------------------------
template <bool FC>   // FC: depth() as one fma (MX_FLAG_FP_CONTRACT, mul_add<> in mx_env_math.hpp)
__global__ __launch_bounds__(256) void k_amplifier(const AmpDesc* __restrict__ descs, size_t n /* stereo floats */) {
    const AmpDesc d = descs[blockIdx.y];
    const size_t nq = (n + 3) >> 2;
    const double md = d.mod_depth, amp = d.amplitude;
    const double one_minus = 1.0 - md;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 v = ld4(d.in, q, n);
        double m0 = 1.0, m1 = 1.0;
        if (d.ctl) {  // block-uniform
            const float2 c = ld2(d.ctl, q, n >> 1);   // stereo floats 4q..4q+3 <-> mono 2q, 2q+1
            m0 = (double)c.x; m1 = (double)c.y;
        }
        const double dep0 = mul_add<FC>(md, m0, one_minus);   // depth(), amplifier.rs:71-73
        const double dep1 = mul_add<FC>(md, m1, one_minus);
        float4 o;
        o.x = (float)((double)v.x * dep0 * amp);
        o.y = (float)((double)v.y * dep0 * amp);
        o.z = (float)((double)v.z * dep1 * amp);
        o.w = (float)((double)v.w * dep1 * amp);
        st4(d.out, q, n, o);
    }
}
void launch_amplifier(const AmpDesc* d, uint32_t n, size_t frames, hipStream_t s, bool fc) {
    if (!n || !frames) return;
    const size_t ns = frames * 2;
    dim3 grid(grid_x((ns + 3) / 4, 256, 4096), n);
    if (fc) hipLaunchKernelGGL(k_amplifier<true>, grid, dim3(256), 0, s, d, ns);
    else hipLaunchKernelGGL(k_amplifier<false>, grid, dim3(256), 0, s, d, ns);
}

// ---------------------------------------------------------------------------------------------
// Oscillator (src/module/oscillator.rs:15-37,65-92) and FmSine (src/module/fm_sine.rs:37-56).
// The reference's f64 sin is the host libm's, the device's is ocml's: both good f64 routines whose f32 casts differ only where the real sine lies within their
// last-bit errors of an f32 rounding boundary (counted: DESIGN.md "Sine").  module_sin_f32 (mx_sin_f32.hpp) takes the device's sine unless that is the case and
// otherwise rounds the real sine, computed in double-double arithmetic: the float (float)glibc_sin(x) gives, save ~2 samples in 10^9.  Square is exact (below).
// ---------------------------------------------------------------------------------------------
// SIN_MODE (MX_SIN_MODE, read once per graph): 0 the default above; 1 the plain cast of the device's sine (the A/B of that count); 2 the double-double path for
// every sample (tests)
#define MX_SIN_DEVICE_ULPS 8.6   // ocml's f64 sin error bound (2 ulp) taken four times over + glibc's 0.55: the slow path then runs on ~3 samples in 10^8
__device__ __forceinline__ float module_sin_f32(double x, int mode) {
    if (mode == 2) return (x == x && fabs(x) < 1099511627776.0) ? dd_to_f32(sin_dd(x)) : (float)sin(x);
    const double y = sin(x);
    return mode == 1 ? (float)y : sin_f32_from(x, y, MX_SIN_DEVICE_ULPS);
}
#define MX_PI 3.14159265358979323846264338327950288

__device__ __forceinline__ double osc_saw(double n) { return 2.0 * (n - floor(0.5 + n)); }

// Square = sign(sin(x)) by the sign BIT (oscillator.rs:15-23,80).  libm's sin never gets the sign of a non-zero result
// wrong, so the reference's value is the sign of the real number sin(x) for the f64 x it formed -- computed here EXACTLY,
// not through the device's sin (whose last-bit differences flip the sign next to a zero crossing):
//   m = rint(x / pi);  r = x - m pi  with pi = PI_HI + PI_MID + PI_LO (161 bits);  sign(sin x) = sign(r) * (-1)^m.
// fma(-m, PI_HI, x) is exact (the difference is a multiple of ulp(PI_HI) below 2 in magnitude), the PI_MID product is
// split exactly (two-product) and added with a two-sum; what is left out is below m * 1e-49.  No f64 below 2^40 comes
// closer than ~1e-20 to a multiple of pi, so the sign is that of the real number.  |x| >= 2^40 (days of audio at ultrasonic
// frequencies), infinities and NaN keep the sign bit of the device's sin.  An m that is off by one (x / pi next to a
// half-integer) leaves |r| < pi, where the identity still holds.
__device__ __forceinline__ bool sin_is_negative(double x) {
    if (x == 0.0) return signbit(x);                               // sin(+-0) = +-0: the sign bit decides (oscillator.rs:16-22)
    if (!(fabs(x) < 1099511627776.0)) return signbit(sin(x));      // 2^40 and beyond, inf, NaN
    const double PI_HI = 0x1.921fb54442d18p+1, PI_MID = 0x1.1a62633145c07p-53, PI_LO = -0x1.f1976b7ed8fbcp-109;
    const double m = rint(x * 0x1.45f306dc9c883p-2);
    const double r1 = fma(-m, PI_HI, x);                           // exact
    const double p2 = m * PI_MID, e2 = fma(m, PI_MID, -p2);        // m * PI_MID = p2 + e2 exactly
    const double sd = r1 - p2, bb = sd - r1;
    const double t = (r1 - (sd - bb)) + (-p2 - bb);                // r1 - p2 = sd + t exactly (two-sum)
    const double r = sd + ((t - e2) - m * PI_LO);
    const bool m_odd = ((long long)m & 1LL) != 0;
    return (r < 0.0) != m_odd;
}

__global__ __launch_bounds__(256) void k_oscillator(const OscDesc* __restrict__ descs, size_t frames, uint64_t t0, double sr, int sin_mode) {
    const OscDesc d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += (size_t)gridDim.x * 256) {
        const double tt = (double)(t0 + (uint64_t)i) / sr;
        const double n = tt * d.freq;
        double v;
        if (d.waveform == 2) {                                                        // Sine: `(n * 2.0 * PI).sin() as f32` (oscillator.rs:25-27,87)
            const float sm = module_sin_f32(n * 2.0 * MX_PI, sin_mode);
            d.mono[i] = sm;
            reinterpret_cast<float2*>(d.stereo)[i] = make_float2(sm, sm);
            continue;
        }
        switch (d.waveform) {
        case 3: v = sin_is_negative(n * 2.0 * MX_PI) ? -1.0 : 1.0; break;                // Square: exact sign of sin (see sin_is_negative)
        case 5: v = osc_saw(n); break;                                                // Saw
        case 4: v = 2.0 * fabs(osc_saw(n)) - 1.0; break;                              // Triangle
        case 0: v = 1.0; break;                                                       // On
        default: v = 0.0; break;                                                      // Off
        }
        const float sm = (float)v;
        d.mono[i] = sm;
        reinterpret_cast<float2*>(d.stereo)[i] = make_float2(sm, sm);
    }
}
void launch_oscillator(const OscDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s, int sin_mode) {
    if (!n || !frames) return;
    dim3 grid(grid_x(frames, 256, 4096), n);
    hipLaunchKernelGGL(k_oscillator, grid, dim3(256), 0, s, d, frames, t0, sample_rate, sin_mode);
}

__global__ __launch_bounds__(256) void k_fm_sine(const FmDesc* __restrict__ descs, size_t frames, uint64_t t0, double sr, int sin_mode) {
    const FmDesc d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += (size_t)gridDim.x * 256) {
        const double tt = (double)(t0 + (uint64_t)i) / sr;
        const double xin = d.in ? (double)d.in[i] : 0.0;
        const double co = (d.freq_mid + d.freq_amp * xin) * 2.0 * MX_PI;
        const float x = module_sin_f32(co * tt, sin_mode);                              // `(co * t).sin() as f32` (fm_sine.rs:50-52)
        reinterpret_cast<float2*>(d.out)[i] = make_float2(x, x);
    }
}
void launch_fm_sine(const FmDesc* d, uint32_t n, size_t frames, uint64_t t0, double sample_rate, hipStream_t s, int sin_mode) {
    if (!n || !frames) return;
    dim3 grid(grid_x(frames, 256, 4096), n);
    hipLaunchKernelGGL(k_fm_sine, grid, dim3(256), 0, s, d, frames, t0, sample_rate, sin_mode);
}

// ---------------------------------------------------------------------------------------------
// Trigger / StereoPanner / StereoSplitter (trigger.rs:35-48, stereo_panner.rs:30-41,
// stereo_splitter.rs:33-47): fills and layout shuffles, 16 B per lane on the wide side.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trigger(const TrigDesc* __restrict__ descs, size_t frames) {
    const TrigDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;
    const float4 v = make_float4(d.value, d.value, d.value, d.value);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) st4(d.out, q, frames, v);
}
// a Trigger whose params change at tick boundaries inside the run (Engine::client_update between two ticks, src/engine.rs:192-214):
// sample i carries the gate of tick i / fpc
__global__ __launch_bounds__(256) void k_trigger_sched(const TrigDesc* __restrict__ descs, size_t frames, size_t fpc, GateBits gates) {
    const TrigDesc d = descs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += (size_t)gridDim.x * 256)
        d.out[i] = gate_bit(gates, blockIdx.y, (uint32_t)(i / fpc)) ? 1.0f : 0.0f;   // trigger.rs:38-41
}
void launch_trigger(const TrigDesc* d, uint32_t n, size_t frames, size_t fpc, const GateBits* gates, hipStream_t s) {
    if (!n || !frames) return;
    if (gates) {
        hipLaunchKernelGGL(k_trigger_sched, dim3(grid_x(frames, 256, 2048), n), dim3(256), 0, s, d, frames, fpc ? fpc : frames, *gates);
        return;
    }
    dim3 grid(grid_x((frames + 3) / 4, 256, 2048), n);
    hipLaunchKernelGGL(k_trigger, grid, dim3(256), 0, s, d, frames);
}

__global__ __launch_bounds__(256) void k_panner(const PanDesc* __restrict__ descs, size_t frames) {
    const PanDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;   // quads of frames -> two stereo quads
    const size_t ns = frames * 2;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 l = d.l ? ld4(d.l, q, frames) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r = d.r ? ld4(d.r, q, frames) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(d.out, 2 * q, ns, make_float4(l.x, r.x, l.y, r.y));
        if ((2 * q + 1) * 4 < ns) st4(d.out, 2 * q + 1, ns, make_float4(l.z, r.z, l.w, r.w));
    }
}
void launch_panner(const PanDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x((frames + 3) / 4, 256, 4096), n);
    hipLaunchKernelGGL(k_panner, grid, dim3(256), 0, s, d, frames);
}

__global__ __launch_bounds__(256) void k_splitter(const SplitDesc* __restrict__ descs, size_t frames) {
    const SplitDesc d = descs[blockIdx.y];
    const size_t nq = (frames + 3) >> 2;
    const size_t ns = frames * 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const float4 a = d.in ? ld4(d.in, 2 * q, ns) : z;
        const float4 b = (d.in && (2 * q + 1) * 4 < ns) ? ld4(d.in, 2 * q + 1, ns) : z;
        st4(d.l, q, frames, make_float4(a.x, a.z, b.x, b.z));
        st4(d.r, q, frames, make_float4(a.y, a.w, b.y, b.w));
    }
}
void launch_splitter(const SplitDesc* d, uint32_t n, size_t frames, hipStream_t s) {
    if (!n || !frames) return;
    dim3 grid(grid_x((frames + 3) / 4, 256, 4096), n);
    hipLaunchKernelGGL(k_splitter, grid, dim3(256), 0, s, d, frames);
}

// Plotter (plotter.rs:37-56): de-interleave one tick per job into the indication staging area.
__global__ __launch_bounds__(256) void k_plotter(const PlotJob* __restrict__ jobs, size_t spt) {
    const PlotJob j = jobs[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < spt; i += (size_t)gridDim.x * 256) {
        const float2 v = reinterpret_cast<const float2*>(j.in)[i];
        j.left[i] = v.x; j.right[i] = v.y;
    }
}
void launch_plotter(const PlotJob* d, uint32_t n, size_t spt, hipStream_t s) {
    if (!n || !spt) return;
    dim3 grid(grid_x(spt, 256, 64), n);
    hipLaunchKernelGGL(k_plotter, grid, dim3(256), 0, s, d, spt);
}

// ---------------------------------------------------------------------------------------------
// Sink / ingest sample formats (the data formats either side of the path):
//   f32 -> i16: clamp to [-1, 1], * i16::MAX as f32, `as i16` (saturating, truncating, NaN -> 0)
//               (src/video/encode.rs:183-195) -- done on the device so a sink reads back half the bytes
//   i16 -> f32: sample as f32 / 32768.0 (src/module/stream_input.rs:167-173)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_f32_to_i16(const float* __restrict__ in, int16_t* __restrict__ out, size_t n, int dup) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = in[dup ? (i >> 1) : i];                    // dup: port stored as one float per frame (L == R)
        s = s > 1.0f ? 1.0f : (s < -1.0f ? -1.0f : s);       // NaN stays NaN, as in the reference's comparisons
        const float v = s * 32767.0f;
        int r;
        if (!(v == v)) r = 0;                                // Rust `as i16`: NaN -> 0
        else if (v >= 32767.0f) r = 32767;
        else if (v <= -32768.0f) r = -32768;
        else r = (int)v;                                     // truncation toward zero
        out[i] = (int16_t)r;
    }
}
void launch_f32_to_i16(const float* in, int16_t* out, size_t n, int dup, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_f32_to_i16, dim3(grid_x(n, 256, 4096)), dim3(256), 0, s, in, out, n, dup);
}
__global__ __launch_bounds__(256) void k_i16_to_f32(const int16_t* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (float)in[i] / 32768.0f;
}
void launch_i16_to_f32(const int16_t* in, float* out, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_i16_to_f32, dim3(grid_x(n, 256, 4096)), dim3(256), 0, s, in, out, n);
}

// a list of small device-to-device copies as ONE launch (a topology edit carries thousands of modules' states over, mx_graph_adopt_state):
// block b copies job b; 4-byte words when both ends and the length allow, bytes otherwise
// Staged host -> device uploads of a run (gate bits, descriptors, parameter events: a few KB to a few MB) as a KERNEL that reads the page-locked staging buffer.
// hipMemcpyAsync puts them on an SDMA engine, and an SDMA copy cannot wait for a barrier packet of a compute queue: when the stream has a pending hipStreamWaitEvent on
// another stream's event (the tail stream of an overlapped graph), the RUNTIME waits for that event ON THE HOST before it queues the copy -- measured (round 5, rocprofv3
// --hip-runtime-trace): hipMemcpyAsync calls of 8 - 10 ms every few runs, the queue drained each time, 1024 strips x 256 ticks 0.90 -> 1.40 ms per run.  A launch is
// ordered on the device.
__global__ __launch_bounds__(256) void k_upload(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t bytes, int aligned) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if (!aligned) { for (size_t i = t; i < bytes; i += step) dst[i] = src[i]; return; }     // (a destination inside a buffer: plot jobs)
    const size_t n16 = bytes / 16;
    for (size_t i = t; i < n16; i += step) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
}
void launch_upload(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    const int aligned = ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) ? 1 : 0;
    const size_t items = aligned ? bytes / 16 + 1 : bytes;
    const size_t blocks = std::min<size_t>(1024, (items + 255) / 256);
    hipLaunchKernelGGL(k_upload, dim3((unsigned)blocks), dim3(256), 0, s, (uint8_t*)dst, (const uint8_t*)src, bytes, aligned);
}

__global__ __launch_bounds__(64) void k_copy_jobs(const CopyJob* __restrict__ jobs) {
    const CopyJob j = jobs[blockIdx.x];
    const bool words = ((((uintptr_t)j.dst) | ((uintptr_t)j.src) | j.bytes) & 3u) == 0;
    if (words) {
        const uint32_t* s = (const uint32_t*)j.src; uint32_t* d = (uint32_t*)j.dst;
        for (size_t i = threadIdx.x; i < j.bytes / 4; i += 64) d[i] = s[i];
    } else {
        const uint8_t* s = (const uint8_t*)j.src; uint8_t* d = (uint8_t*)j.dst;
        for (size_t i = threadIdx.x; i < j.bytes; i += 64) d[i] = s[i];
    }
}
void launch_copy_jobs(const CopyJob* device_jobs, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_copy_jobs, dim3(n), dim3(64), 0, s, device_jobs);
}

}  // namespace mx
