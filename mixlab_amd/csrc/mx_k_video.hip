// mx_k_video.hip -- pixel kernels: VideoMixer cross-fade (+ blank), bicubic plane scaler, YUV420P->RGBA.
//
// Integer work throughout: results are bit-exact against the CPU oracle.
// Frames are planar yuv420p, 8 bit, resident in HBM; every plane row starts 64-byte aligned
// (stride % 64 == 0), so a lane moves 16 pixels (one dwordx4) and a wave 1 KiB per instruction.
#include <algorithm>

#include "mx_dev.hpp"
#include "mx_video.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// Cross-fade (reference src/module/video_mixer.rs:151-239, fade_line :211-235):
//   out = ((a * fade + b * (255 - fade)) / 255) as u8   in u16 lanes, integer division
// A missing A or B reads the blank output plane itself (video_mixer.rs:180-188), i.e. the constant
// the blank fill wrote (Y = 0x00, U = V = 0x80, codec/src/ffmpeg/frame.rs:128-132): the blank pass
// is folded into this kernel instead of costing a separate frame-sized write.
// x / 255 == (x + 1 + (x >> 8)) >> 8 for 0 <= x <= 65534; here x <= 255 * 255 (checked exhaustively in tests).
// Like fade_line, rows are processed in 32-byte blocks up to ceil(width / 32) * 32.
// algorithmic bytes per output frame: 3 F (2 F read + F written).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fade4(uint32_t a4, uint32_t b4, uint32_t fa, uint32_t fb) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t a = (a4 >> (8 * k)) & 0xffu, b = (b4 >> (8 * k)) & 0xffu;
        const uint32_t x = a * fa + b * fb;              // <= 255 * 255, fits u16 like the reference's lanes
        r |= (((x + 1u + (x >> 8)) >> 8) & 0xffu) << (8 * k);
    }
    return r;
}

__global__ __launch_bounds__(256) void k_crossfade(FadeArgs args) {
    // flat index over (plane, row, 16-byte chunk)
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = gid;
    int plane = 0;
    if (idx >= args.chunks[0]) { idx -= args.chunks[0]; plane = 1; if (idx >= args.chunks[1]) { idx -= args.chunks[1]; plane = 2; } }
    if (plane == 2 && idx >= args.chunks[2]) return;
    const uint32_t cpr = args.chunks_per_row[plane];
    const uint32_t row = idx / cpr, col = (idx - row * cpr) * 16u;
    const uint32_t fa = args.fade, fb = 255u - args.fade;
    const uint32_t blank = plane ? 0x80808080u : 0u;
    uint4 a = make_uint4(blank, blank, blank, blank), b = a;
    if (args.a[plane]) a = *reinterpret_cast<const uint4*>(args.a[plane] + (size_t)row * args.a_stride[plane] + col);
    if (args.b[plane]) b = *reinterpret_cast<const uint4*>(args.b[plane] + (size_t)row * args.b_stride[plane] + col);
    uint4 o;
    o.x = fade4(a.x, b.x, fa, fb); o.y = fade4(a.y, b.y, fa, fb); o.z = fade4(a.z, b.z, fa, fb); o.w = fade4(a.w, b.w, fa, fb);
    *reinterpret_cast<uint4*>(args.out[plane] + (size_t)row * args.out_stride[plane] + col) = o;
}

void launch_crossfade(const FadeArgs& a, hipStream_t s) {
    flush_scales(s);
    const uint32_t total = a.chunks[0] + a.chunks[1] + a.chunks[2];
    if (!total) return;
    hipLaunchKernelGGL(k_crossfade, dim3((total + 255) / 256), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// Cross-fade chain: a cascade of VideoMixers (each `out = fade(A, B)` truncated to u8) evaluated in
// one pass.  All layer loads of a lane are issued up-front (<= 8 dwordx4 in flight), the chain runs in
// registers, one dwordx4 store.  algorithmic bytes per output frame: (n_src + 1) F instead of 3 F per
// mixer (8 layers: 9 F = 28 MB instead of 65 MB).
// ---------------------------------------------------------------------------------------------
// Packed arithmetic: the reference's u16 lanes (packed_simd u16x32, video_mixer.rs:215-227) map onto
// v_pk_mul_lo_u16 / v_pk_add_u16: two pixels per VALU op.  A dword of 4 pixels is split into its even
// and odd bytes (two u16x2), and the running composite stays in that form across all steps of a chain.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct Px4 { u16x2 e, o; };   // bytes 0,2 and bytes 1,3 of a dword, each widened to u16
__device__ __forceinline__ Px4 px4_unpack(uint32_t w) {
    Px4 r;
    r.e = __builtin_bit_cast(u16x2, w & 0x00ff00ffu);
    r.o = __builtin_bit_cast(u16x2, (w >> 8) & 0x00ff00ffu);
    return r;
}
__device__ __forceinline__ uint32_t px4_pack(Px4 v) {
    return __builtin_bit_cast(uint32_t, v.e) | (__builtin_bit_cast(uint32_t, v.o) << 8);
}
__device__ __forceinline__ u16x2 fade_pk(u16x2 a, u16x2 b, u16x2 fa, u16x2 fb) {
    const u16x2 x = a * fa + b * fb;                          // <= 255 * 255: no u16 overflow, as in the reference
    const u16x2 one = {1, 1};
    return (u16x2)((x + one + (x >> 8)) >> 8);              // x / 255 for x <= 65534; the sum stays below 65536
}
__device__ __forceinline__ Px4 fade_px4(Px4 a, Px4 b, u16x2 fa, u16x2 fb) {
    Px4 r; r.e = fade_pk(a.e, b.e, fa, fb); r.o = fade_pk(a.o, b.o, fa, fb); return r;
}

template <int NW>   // NW dwords per lane and plane
__device__ __forceinline__ void chain_eval(uint32_t (&v)[NW], const uint32_t (*L)[NW], uint32_t n_src,
                                           const uint32_t* fade, const uint32_t* v_is_a) {
    Px4 acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = px4_unpack(L[0][w]);
#pragma unroll
    for (int k = 1; k < MX_CHAIN_MAX_SRC; ++k) {
        if (k < (int)n_src) {
            // fade(a, b; f, 255 - f) with the running value as B is fade(running, other; 255 - f, f): which side the
            // running value sits on only swaps the two (wave-uniform) factors, never the per-pixel code
            const unsigned short f = (unsigned short)(v_is_a[k - 1] ? fade[k - 1] : 255u - fade[k - 1]), g = (unsigned short)(255u - f);
            const u16x2 fa = {f, f}, fb = {g, g};
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[w] = fade_px4(acc[w], px4_unpack(L[k][w]), fa, fb);
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = px4_pack(acc[w]);
}

__global__ __launch_bounds__(256) void k_fade_chain(ChainArgs args) {
    uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    int plane = 0;
    if (idx >= args.chunks[0]) { idx -= args.chunks[0]; plane = 1; if (idx >= args.chunks[1]) { idx -= args.chunks[1]; plane = 2; } }
    if (plane == 2 && idx >= args.chunks[2]) return;
    const uint32_t cpr = args.chunks_per_row[plane];
    const uint32_t row = idx / cpr, col = (idx - row * cpr) * 16u;
    const uint32_t blank = plane ? 0x80808080u : 0u;
    uint32_t L[MX_CHAIN_MAX_SRC][4];
#pragma unroll
    for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
        uint4 t = make_uint4(blank, blank, blank, blank);
        if (k < (int)args.n_src && args.src[k].p[plane])
            t = *reinterpret_cast<const uint4*>(args.src[k].p[plane] + (size_t)row * args.src[k].stride[plane] + col);
        L[k][0] = t.x; L[k][1] = t.y; L[k][2] = t.z; L[k][3] = t.w;
    }
    uint32_t v[4];
    chain_eval<4>(v, L, args.n_src, args.fade, args.v_is_a);
    *reinterpret_cast<uint4*>(args.out[plane] + (size_t)row * args.out_stride[plane] + col) = make_uint4(v[0], v[1], v[2], v[3]);
}
void launch_fade_chain(const ChainArgs& a, hipStream_t s) {
    flush_scales(s);   // queued scaler output may be among the layers
    const uint32_t total = a.chunks[0] + a.chunks[1] + a.chunks[2];
    if (!total) return;
    hipLaunchKernelGGL(k_fade_chain, dim3((total + 255) / 256), dim3(256), 0, s, a);
}

// the same chain feeding YUV420P -> RGBA (+ matrix): one lane owns 16 luma pixels x 2 rows and their
// 8 + 8 chroma samples, so the composite never exists as a YUV frame.  algorithmic bytes per frame:
// n_src F + 4 w h.
__device__ __forceinline__ int clip8c(int v) { return min(max(v, 0), 255); }
// 24-bit multiplies (v_mad_i32_i24, full rate; a 32-bit v_mul_lo_u32 is quarter rate): R, G, B are 8-bit and the
// launcher takes this path only when every matrix entry fits 24 bits signed, so the products are the int32 ones
__device__ __forceinline__ int mx_row24(const int* m, int R, int G, int B) {
    return __mul24(m[0], R) + __mul24(m[1], G) + __mul24(m[2], B) + m[3] + 2048;
}
// pixel assembly: {clip8(r >> sh), clip8(g >> sh), clip8(b >> sh), 255} with gfx950's v_ashr_pk_u8_i32 (two
// shift + saturate + pack per instruction).  The builtin is used on purpose: the instruction writes only the low 16
// bits of its destination, the builtin's u16 result makes the compiler select that half -- while the compiler's own
// pattern match of clip8(x >> s) | clip8(y >> s) << 8 (ROCm 7.2) ORs the whole register into the pixel and
// leaves stale bits in the B byte (caught by tests/test_gpu_video_graph.py).
__device__ __forceinline__ uint32_t pack_rgba(int r, int g, int b, int sh) {
    const uint32_t rg = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(r, g, sh);
    const uint32_t ba = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(b, 255 << sh, sh);
    return rg | (ba << 16);
}
template <int MM>   // matrix mode: 0 none, 1 full 32-bit products, 2 24-bit products
__device__ __forceinline__ uint32_t yuv_px(const int* m, int Y, int U, int V) {
    const int C = Y - 16, D = U - 128, E = V - 128;
    const int rs = 298 * C + 459 * E + 128, gs = 298 * C - 55 * D - 136 * E + 128, bs = 298 * C + 541 * D + 128;
    if (MM == 0) return pack_rgba(rs, gs, bs, 8);
    const int R = clip8c(rs >> 8), G = clip8c(gs >> 8), B = clip8c(bs >> 8);
    if (MM == 2) return pack_rgba(mx_row24(m, R, G, B), mx_row24(m + 4, R, G, B), mx_row24(m + 8, R, G, B), 12);
    return pack_rgba(m[0] * R + m[1] * G + m[2] * B + m[3] + 2048, m[4] * R + m[5] * G + m[6] * B + m[7] + 2048,
                     m[8] * R + m[9] * G + m[10] * B + m[11] + 2048, 12);
}

template <int MM>
__global__ __launch_bounds__(256) void k_fade_chain_rgba(ChainRgbaArgs a) {
    const uint32_t xb = blockIdx.x * 64 + (threadIdx.x & 63);     // 8-pixel column block
    const uint32_t yb = blockIdx.y * 4 + (threadIdx.x >> 6);       // row pair
    if (xb * 8 >= a.width || yb * 2 >= a.height) return;
    // per source: 2 dwords of Y for each of the two rows, one dword of U, one of V  (6 dwords)
    uint32_t L[MX_CHAIN_MAX_SRC][6];
#pragma unroll
    for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
        L[k][0] = L[k][1] = L[k][2] = L[k][3] = 0u; L[k][4] = L[k][5] = 0x80808080u;
        if (k < (int)a.n_src) {
            const ChainSrc& s = a.src[k];
            if (s.p[0]) {
                const uint2 r0 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)(2 * yb) * s.stride[0] + xb * 8);
                const uint2 r1 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)(2 * yb + 1) * s.stride[0] + xb * 8);
                L[k][0] = r0.x; L[k][1] = r0.y; L[k][2] = r1.x; L[k][3] = r1.y;
            }
            if (s.p[1]) L[k][4] = *reinterpret_cast<const uint32_t*>(s.p[1] + (size_t)yb * s.stride[1] + xb * 4);
            if (s.p[2]) L[k][5] = *reinterpret_cast<const uint32_t*>(s.p[2] + (size_t)yb * s.stride[2] + xb * 4);
        }
    }
    uint32_t v[6];
    chain_eval<6>(v, L, a.n_src, a.fade, a.v_is_a);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t yrow = 2 * yb + r;
        if (yrow >= a.height) break;
        uint8_t* o = a.rgba + (size_t)yrow * a.rgba_stride + (size_t)xb * 32;
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {       // 2 groups of 4 pixels
            const uint32_t yw = v[2 * r + g4];
            const uint32_t cu = (v[4] >> (16 * g4)) & 0xffffu, cv = (v[5] >> (16 * g4)) & 0xffffu;
            uint32_t px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                px[k] = yuv_px<MM>(a.m, (int)((yw >> (8 * k)) & 0xff), (int)((cu >> (8 * (k >> 1))) & 0xff), (int)((cv >> (8 * (k >> 1))) & 0xff));
            const uint32_t x = xb * 8 + g4 * 4;
            if (x + 4 <= a.width) *reinterpret_cast<uint4*>(o + g4 * 16) = make_uint4(px[0], px[1], px[2], px[3]);
            else for (uint32_t k = 0; x + k < a.width; ++k) reinterpret_cast<uint32_t*>(o + g4 * 16)[k] = px[k];
        }
    }
}
void launch_fade_chain_rgba(const ChainRgbaArgs& a0, hipStream_t s) {
    flush_scales(s);
    if (!a0.width || !a0.height) return;
    ChainRgbaArgs a = a0;
    if (a.use_matrix) {
        bool fits = true;
        for (int k = 0; k < 12; ++k) if ((k & 3) != 3 && (a.m[k] < -(1 << 23) || a.m[k] >= (1 << 23))) fits = false;
        a.use_matrix = fits ? 2 : 1;
    }
    const dim3 grid(((a.width + 7) / 8 + 63) / 64, ((a.height + 1) / 2 + 3) / 4);
    if (a.use_matrix == 2) hipLaunchKernelGGL(k_fade_chain_rgba<2>, grid, dim3(256), 0, s, a);
    else if (a.use_matrix) hipLaunchKernelGGL(k_fade_chain_rgba<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_fade_chain_rgba<0>, grid, dim3(256), 0, s, a);
}

// Blank fill (codec/src/ffmpeg/frame.rs:76-138): Y = 0, U = V = 0x80 over the whole allocation of each plane.
__global__ __launch_bounds__(256) void k_blank(uint8_t* y, size_t y_bytes, uint8_t* u, size_t u_bytes, uint8_t* v, size_t v_bytes) {
    const size_t yq = y_bytes / 16, uq = u_bytes / 16, vq = v_bytes / 16;
    const uint4 zy = make_uint4(0, 0, 0, 0), zc = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < yq + uq + vq; i += (size_t)gridDim.x * 256) {
        if (i < yq) reinterpret_cast<uint4*>(y)[i] = zy;
        else if (i < yq + uq) reinterpret_cast<uint4*>(u)[i - yq] = zc;
        else reinterpret_cast<uint4*>(v)[i - yq - uq] = zc;
    }
}
void launch_blank(uint8_t* y, size_t yb, uint8_t* u, size_t ub, uint8_t* v, size_t vb, hipStream_t s) {
    flush_scales(s);
    const size_t q = (yb + ub + vb) / 16;
    if (!q) return;
    hipLaunchKernelGGL(k_blank, dim3(grid_x(q, 256, 2048)), dim3(256), 0, s, y, yb, u, ub, v, vb);
}

// ---------------------------------------------------------------------------------------------
// Bicubic plane scaler -- BUILD-SPECIFIED stand-in for sws_scale(SWS_BICUBIC)
// (codec/src/ffmpeg/scale.rs:23-27,68; libswscale is third-party C outside the reference tree).
// Spec in DESIGN.md "Scaler": host-computed Q14 tap tables (first tap index + 4 coefficients per
// output column / row); H pass t = (sum hc*S + 64) >> 7, V pass D = clip8((sum vc*t + 2^20) >> 21).
// One lane per output pixel; the 4x4 source neighbourhood is served by L1/L2 (neighbouring lanes
// share 3 of 4 columns).  All three planes in one launch.
// ---------------------------------------------------------------------------------------------
// Tiled two-pass version of the same arithmetic (bit-identical): a 256-thread block produces a
// 128 x 32 output tile.  The source window is staged in LDS once (aligned dwords; edge replication
// happens there), the H pass filters each needed source row exactly once per output column, the V
// pass reads four H-filtered rows per pixel group from LDS as ds_read_b128 and stores one dword of 4
// pixels.  The kernel is latency-bound (a 1080p target is ~800 tiles per plane set), so the window
// origin is COMPUTED from the tap spec instead of fetched -- the source loads do not wait for a table
// round trip -- and every coefficient a lane will need is requested in the same burst.
// Used when the window fits 64 KB of LDS (scale ratio <= 2); LDS is sized per launch.
#define SC_TW 128
#define SC_TH 32
// first tap of output o (DESIGN.md "Scaler"): ((floor((2o+1) * src * 65536 / (2 dst)) - 32768) >> 16) - 1, evaluated in f64:
// numerator and denominator are exact (< 2^49), the quotient is corrected with an exact fma remainder, and the rest
// are exact operations on integers below 2^53 -- a 64-bit integer division costs several hundred VALU cycles here.
// Scaler::retarget checks it against the integer tap tables for every tile origin it can be asked for.
__host__ __device__ __forceinline__ int sc_first_tap(uint32_t o, uint32_t src, uint32_t dst) {
    const double n = (double)(2u * o + 1u) * ((double)src * 65536.0), d = 2.0 * (double)dst;
    double q = floor(n / d);
    const double r = fma(-q, d, n);          // exact: |r| < 2 d
    q += (r >= d) ? 1.0 : ((r < 0.0) ? -1.0 : 0.0);
    return (int)floor((q - 32768.0) * (1.0 / 65536.0)) - 1;
}
bool scale_tile_origins_match(uint32_t src, uint32_t dst, const int32_t* first /* host copy of the tap table */) {
    for (uint32_t o = 0; o < dst; ++o)
        if ((o % SC_TW == 0 || o % SC_TH == 0 || o + 1 == dst || (o + 1) % SC_TW == 0 || (o + 1) % SC_TH == 0) && sc_first_tap(o, src, dst) != first[o]) return false;
    return true;
}
__global__ __launch_bounds__(256) void k_scale_bicubic_tiled(ScaleBatchArgs a) {
    int plane = 0;
#pragma unroll
    for (int k = 1; k < MX_SCALE_BATCH_PLANES; ++k) if (k < (int)a.n && blockIdx.x >= a.tile_start[k]) plane = k;   // scalar search
    const ScalePlane p = a.p[plane];
    const uint32_t tile = blockIdx.x - a.tile_start[plane];
    const int ox0 = (int)(tile % a.tiles_x[plane]) * SC_TW, oy0 = (int)(tile / a.tiles_x[plane]) * SC_TH;
    extern __shared__ __attribute__((aligned(16))) uint8_t sc_smem[];
    uint8_t* const S = sc_smem;                                                       // [s_rows][s_stride] source window
    int* const T = reinterpret_cast<int*>(sc_smem + (size_t)a.s_rows * a.s_stride);   // [s_rows][SC_TW] H-filtered rows
    const int tid = threadIdx.x;
    const int oxl = min(ox0 + SC_TW - 1, (int)p.dw - 1), oyl = min(oy0 + SC_TH - 1, (int)p.dh - 1);
    // window origin / extent from the tap spec (tap tables are monotone); columns start on a dword of the source row
    const int cx0 = sc_first_tap((uint32_t)ox0, p.sw, p.dw), cxl = sc_first_tap((uint32_t)oxl, p.sw, p.dw);
    const int ry0 = sc_first_tap((uint32_t)oy0, p.sh, p.dh), ryl = sc_first_tap((uint32_t)oyl, p.sh, p.dh);
    const int cxa = cx0 & ~3;
    const int nc4 = min((cxl + 4 - cxa + 3) >> 2, (int)a.s_stride >> 2), nr = min(ryl + 4 - ry0, (int)a.s_rows);
    // every table entry this lane will need, in one burst
    const int oxi = tid & (SC_TW - 1), ox = min(ox0 + oxi, (int)p.dw - 1);
    const int4 hc = reinterpret_cast<const int4*>(p.hcoef)[ox];
    const int hf = p.hfirst[ox] - cxa;
    const int cg = tid & 31, oyr = tid >> 5;                  // V pass: pixel group (4 columns) and first row; rows oyr + 8k
    int4 vc[4]; int vf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oy = min(oy0 + oyr + 8 * k, (int)p.dh - 1);
        vc[k] = reinterpret_cast<const int4*>(p.vcoef)[oy];
        vf[k] = p.vfirst[oy] - ry0;
    }
    const int sw1 = (int)p.sw - 1, sh1 = (int)p.sh - 1;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.src) | p.src_stride) & 3u) == 0;
    for (int r = tid >> 5; r < nr; r += 8) {                 // 32 lanes per window row, one dword each (no index division)
        const uint8_t* row = p.src + (size_t)min(max(ry0 + r, 0), sh1) * p.src_stride;
        for (int c4 = tid & 31; c4 < nc4; c4 += 32) {
            const int x = cxa + 4 * c4;
            uint32_t w;
            if (aligned && x >= 0 && x + 3 <= sw1) w = *reinterpret_cast<const uint32_t*>(row + x);
            else w = (uint32_t)row[min(max(x, 0), sw1)] | ((uint32_t)row[min(max(x + 1, 0), sw1)] << 8) |
                     ((uint32_t)row[min(max(x + 2, 0), sw1)] << 16) | ((uint32_t)row[min(max(x + 3, 0), sw1)] << 24);   // edge replication
            *reinterpret_cast<uint32_t*>(S + (size_t)r * a.s_stride + 4 * c4) = w;
        }
    }
    __syncthreads();
    if (ox0 + oxi < (int)p.dw) {   // H pass: t = (sum hc * S + 64) >> 7; the 4 taps come out of two aligned dwords
        const int hb = hf & ~3;
        for (int r = tid >> 7; r < nr; r += 2) {
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(S + (size_t)r * a.s_stride + hb);
            const uint32_t w = __builtin_amdgcn_alignbyte(sp[1], sp[0], (uint32_t)(hf & 3));   // bytes hf .. hf+3
            // 24-bit products (full rate; v_mul_lo_u32 is quarter rate): Q14 coefficients x bytes
            const int acc = __mul24(hc.x, (int)(w & 0xffu)) + __mul24(hc.y, (int)((w >> 8) & 0xffu)) + __mul24(hc.z, (int)((w >> 16) & 0xffu)) + __mul24(hc.w, (int)(w >> 24));
            T[r * SC_TW + oxi] = (acc + 64) >> 7;
        }
    }
    __syncthreads();
    // V pass: D = clip8((sum vc * t + 2^20) >> 21), four pixels per lane and row
    const int oxg = cg * 4;
    if (ox0 + oxg < (int)p.dw) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oyk = oy0 + oyr + 8 * k;
            if (oyk >= (int)p.dh) break;
            const int4 t0 = *reinterpret_cast<const int4*>(&T[(vf[k] + 0) * SC_TW + oxg]), t1 = *reinterpret_cast<const int4*>(&T[(vf[k] + 1) * SC_TW + oxg]);
            const int4 t2 = *reinterpret_cast<const int4*>(&T[(vf[k] + 2) * SC_TW + oxg]), t3 = *reinterpret_cast<const int4*>(&T[(vf[k] + 3) * SC_TW + oxg]);
            const int4 c = vc[k];
            // Q14 coefficients x H-filtered values (|t| < 2^16): 24-bit products are the int32 ones
            auto col = [&](int a0, int a1, int a2, int a3) { return __mul24(c.x, a0) + __mul24(c.y, a1) + __mul24(c.z, a2) + __mul24(c.w, a3) + (1 << 20); };
            // clip8(sum >> 21) x 4 -> one dword: two v_ashr_pk_u8_i32 (explicit builtin, see pack_rgba)
            const uint32_t lo = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(col(t0.x, t1.x, t2.x, t3.x), col(t0.y, t1.y, t2.y, t3.y), 21);
            const uint32_t hi = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(col(t0.z, t1.z, t2.z, t3.z), col(t0.w, t1.w, t2.w, t3.w), 21);
            const uint32_t quad = lo | (hi << 16);
            uint8_t* o = p.dst + (size_t)oyk * p.dst_stride + ox0 + oxg;
            if (ox0 + oxg + 4 <= (int)p.dw && (reinterpret_cast<uintptr_t>(o) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = quad;
            else for (int j = 0; j < 4 && ox0 + oxg + j < (int)p.dw; ++j) o[j] = (uint8_t)(quad >> (8 * j));
        }
    }
}

__global__ __launch_bounds__(256) void k_scale_bicubic_batch(ScaleBatchArgs a) {   // simple gather form, any ratio
    const ScalePlane p = a.p[blockIdx.z];
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.dw || y >= p.dh) return;
    const int4 hc = reinterpret_cast<const int4*>(p.hcoef)[x];
    const int4 vc = reinterpret_cast<const int4*>(p.vcoef)[y];
    const int hf = p.hfirst[x], vf = p.vfirst[y];
    const int sw1 = (int)p.sw - 1, sh1 = (int)p.sh - 1;
    const int x0 = min(max(hf, 0), sw1), x1 = min(max(hf + 1, 0), sw1), x2 = min(max(hf + 2, 0), sw1), x3 = min(max(hf + 3, 0), sw1);
    int t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint8_t* row = p.src + (size_t)min(max(vf + k, 0), sh1) * p.src_stride;
        t[k] = (hc.x * (int)row[x0] + hc.y * (int)row[x1] + hc.z * (int)row[x2] + hc.w * (int)row[x3] + 64) >> 7;
    }
    const int v = (vc.x * t[0] + vc.y * t[1] + vc.z * t[2] + vc.w * t[3] + (1 << 20)) >> 21;
    p.dst[(size_t)y * p.dst_stride + x] = (uint8_t)min(max(v, 0), 255);
}

// All planes of all queued scale jobs of a stream go out as ONE launch (grid.z = plane): these
// kernels are latency-bound, so N jobs cost about as much as one.
void launch_scale_batch(const ScaleBatchArgs& a, hipStream_t s) {
    uint32_t mw = 0, mh = 0, s_rows = 0, s_stride = 0;
    for (uint32_t i = 0; i < a.n; ++i) {
        const ScalePlane& p = a.p[i];
        mw = p.dw > mw ? p.dw : mw; mh = p.dh > mh ? p.dh : mh;
        if (!p.dw || !p.dh) continue;
        // window of a 128 x 32 tile: taps of its first and last output + 4, columns widened to whole dwords
        const uint32_t rows = (uint32_t)(((uint64_t)SC_TH * p.sh + p.dh - 1) / p.dh) + 6;
        const uint32_t cols = (uint32_t)(((uint64_t)SC_TW * p.sw + p.dw - 1) / p.dw) + 6 + 3 + 4;   // + the H pass's second dword
        s_rows = rows > s_rows ? rows : s_rows; s_stride = cols > s_stride ? cols : s_stride;
    }
    if (!a.n || !mw || !mh) return;
    s_stride = (s_stride + 15u) & ~15u;               // rows * stride stays a multiple of 16: T is read as int4
    const size_t lds = (size_t)s_rows * s_stride + (size_t)s_rows * SC_TW * sizeof(int);
    static const int force_simple = env_int("MX_SCALE_SIMPLE", 0);
    if (lds <= 64 * 1024 && !force_simple) {
        ScaleBatchArgs b = a;
        uint32_t total = 0;
        for (uint32_t i = 0; i < a.n; ++i) {
            b.tile_start[i] = total;
            b.tiles_x[i] = (a.p[i].dw + SC_TW - 1) / SC_TW;
            total += b.tiles_x[i] * ((a.p[i].dh + SC_TH - 1) / SC_TH);
        }
        for (uint32_t i = a.n; i <= MX_SCALE_BATCH_PLANES; ++i) { b.tile_start[i] = total; if (i < MX_SCALE_BATCH_PLANES) b.tiles_x[i] = 1; }
        b.s_rows = s_rows; b.s_stride = s_stride;
        if (total) hipLaunchKernelGGL(k_scale_bicubic_tiled, dim3(total), dim3(256), lds, s, b);
    } else {
        hipLaunchKernelGGL(k_scale_bicubic_batch, dim3((mw + 63) / 64, (mh + 3) / 4, a.n), dim3(256), 0, s, a);
    }
}
// Downscaling: the kernel widens with the scale factor (hn / vn taps, DESIGN.md "Scaler").  Two plain passes through
// ScalePlane::tmp -- the H pass filters every source row once, the V pass reads vn of those rows per pixel -- instead of
// hn * vn source reads per pixel.  Same arithmetic as the 4-tap kernels; outputs are monitor-sized, this is not a hot path.
__global__ __launch_bounds__(256) void k_scale_wide_h(ScaleArgs a) {
    const ScalePlane p = a.p[blockIdx.z];
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = p.h_row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.dw || y >= p.h_row0 + p.h_rows) return;
    const uint8_t* row = p.src + (ptrdiff_t)y * (ptrdiff_t)p.src_stride;   // a band's `src` is the slice's base minus src_row0 rows: only rows of the slice are touched
    const int32_t* c = p.hcoef + (size_t)x * p.hn;
    const int f = p.hfirst[x], sw1 = (int)p.sw - 1;
    int acc = 0;
    for (uint32_t k = 0; k < p.hn; ++k) acc += c[k] * (int)row[min(max(f + (int)k, 0), sw1)];
    p.tmp[(size_t)y * p.dw + x] = (acc + 64) >> 7;
}
__global__ __launch_bounds__(256) void k_scale_wide_v(ScaleArgs a) {
    const ScalePlane p = a.p[blockIdx.z];
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.dw || y >= p.dh) return;
    const int32_t* c = p.vcoef + (size_t)y * p.vn;
    const int f = p.vfirst[y], sh1 = (int)p.sh - 1;
    int acc = 0;
    for (uint32_t k = 0; k < p.vn; ++k) acc += c[k] * p.tmp[(size_t)min(max(f + (int)k, 0), sh1) * p.dw + x];
    p.dst[(size_t)y * p.dst_stride + x] = (uint8_t)min(max((acc + (1 << 20)) >> 21, 0), 255);
}
void launch_scale_wide(const ScaleArgs& a, hipStream_t s) {
    uint32_t mw = 0, msh = 0, mdh = 0;
    for (int i = 0; i < 3; ++i) { mw = std::max(mw, a.p[i].dw); msh = std::max(msh, a.p[i].h_rows); mdh = std::max(mdh, a.p[i].dh); }
    if (!mw || !msh || !mdh) return;
    hipLaunchKernelGGL(k_scale_wide_h, dim3((mw + 63) / 64, (msh + 3) / 4, 3), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_scale_wide_v, dim3((mw + 63) / 64, (mdh + 3) / 4, 3), dim3(256), 0, s, a);
}

void launch_scale_bicubic(const ScaleArgs& a, hipStream_t s) {
    ScaleBatchArgs b;
    b.n = 3;
    for (int i = 0; i < 3; ++i) b.p[i] = a.p[i];
    launch_scale_batch(b, s);
}

// plane copy (identity "scale" into a differently-strided frame, and frame clones)
__global__ __launch_bounds__(256) void k_copy_planes(CopyArgs a) {
    const int plane = blockIdx.z;
    const uint32_t row = blockIdx.y;
    if (row >= a.rows[plane]) return;
    const uint32_t chunks = (a.row_bytes[plane] + 15) / 16;
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < chunks; c += gridDim.x * 256)
        reinterpret_cast<uint4*>(a.dst[plane] + (size_t)row * a.dst_stride[plane])[c] =
            reinterpret_cast<const uint4*>(a.src[plane] + (size_t)row * a.src_stride[plane])[c];
}
void launch_copy_planes(const CopyArgs& a, hipStream_t s) {
    flush_scales(s);
    uint32_t mr = a.rows[0] > a.rows[1] ? a.rows[0] : a.rows[1];
    uint32_t mb = a.row_bytes[0];
    if (!mr || !mb) return;
    hipLaunchKernelGGL(k_copy_planes, dim3(((mb + 15) / 16 + 255) / 256, mr, 3), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// YUV420P -> RGBA8 -- BUILD-SPECIFIED (no reference counterpart): BT.709 limited range, integer,
// nearest chroma, optional Q12 3x4 colour matrix (DESIGN.md "Colour").  4 pixels per lane: one
// dword of Y, one ushort of U and V, one dwordx4 of RGBA out.
// algorithmic bytes per frame: F + 4 * w * h.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int clip8(int v) { return min(max(v, 0), 255); }

__global__ __launch_bounds__(256) void k_yuv420_to_rgba(RgbaArgs a) {
    const uint32_t xq = blockIdx.x * 64 + (threadIdx.x & 63);   // group of 4 pixels
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xq * 4 >= a.width || y >= a.height) return;
    const uint32_t y4 = *reinterpret_cast<const uint32_t*>(a.y + (size_t)y * a.y_stride + xq * 4);
    const uint16_t u2 = *reinterpret_cast<const uint16_t*>(a.u + (size_t)(y >> 1) * a.u_stride + xq * 2);
    const uint16_t v2 = *reinterpret_cast<const uint16_t*>(a.v + (size_t)(y >> 1) * a.v_stride + xq * 2);
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int Y = (int)((y4 >> (8 * k)) & 0xff), U = (int)((u2 >> (8 * (k >> 1))) & 0xff), V = (int)((v2 >> (8 * (k >> 1))) & 0xff);
        px[k] = a.use_matrix ? yuv_px<1>(a.m, Y, U, V) : yuv_px<0>(a.m, Y, U, V);
    }
    uint8_t* o = a.rgba + (size_t)y * a.rgba_stride + (size_t)xq * 16;
    if (xq * 4 + 4 <= a.width) *reinterpret_cast<uint4*>(o) = make_uint4(px[0], px[1], px[2], px[3]);
    else for (uint32_t k = 0; xq * 4 + k < a.width; ++k) reinterpret_cast<uint32_t*>(o)[k] = px[k];
}
void launch_yuv420_to_rgba(const RgbaArgs& a, hipStream_t s) {
    flush_scales(s);
    if (!a.width || !a.height) return;
    hipLaunchKernelGGL(k_yuv420_to_rgba, dim3(((a.width + 3) / 4 + 63) / 64, (a.height + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace mx
