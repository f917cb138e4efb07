// mx_k_video.hip -- pixel kernels: VideoMixer cross-fade (+ blank), bicubic plane scaler, YUV420P->RGBA.
//
// Integer work throughout: results are bit-exact against the CPU oracle.
// Frames are planar yuv420p, 8 bit, resident in HBM; every plane row starts 64-byte aligned
// (stride % 64 == 0), so a lane moves 16 pixels (one dwordx4) and a wave 1 KiB per instruction.
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <map>
#include <vector>
#include <mutex>

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

// Workgroup b of a launch is observed to run on XCD b % 8, each XCD with an L2 of its own (MI355X_MICROARCH.md "Workgroup dispatch":
// a speed assumption, never a correctness one).  xcd_run hands every XCD a CONTIGUOUS run of the n work items, so that neighbouring
// tiles -- which share cache lines (the two halves of a chroma line, the halo of a scaler window) -- meet in one L2 instead of each
// fetching its own copy over the fabric.  A bijection of [0, n).
__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, j = b >> 3;
    return x * q + min(x, r) + j;
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
    r.o = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w, 0x0c030c01u));   // (w >> 8) & 0x00ff00ff in one v_perm_b32
    return r;
}
__device__ __forceinline__ uint32_t px4_pack(Px4 v) {
    return __builtin_bit_cast(uint32_t, v.e) | (__builtin_bit_cast(uint32_t, v.o) << 8);
}
__device__ __forceinline__ u16x2 fade_pk(u16x2 a, u16x2 b, u16x2 fa, u16x2 fb) {
    // x = a fa + b fb <= 255 * 255: no u16 overflow, as in the reference.  x / 255 == (x + 1 + (x >> 8)) >> 8 for x <= 65534, and with
    // x1 = x + 1 that is (x1 + (x1 >> 8)) >> 8: the two differ only where x1 is a multiple of 256, and there both sums have the same
    // high byte (checked for every x <= 65025 on the host, tests/test_fastdiv.py) -- the + 1 rides on the multiply-add: 2 v_pk_mad + 3.
    const u16x2 one = {1, 1};
    uint32_t t = __builtin_bit_cast(uint32_t, (u16x2)(b * fb + one));
    asm volatile("" : "+v"(t));                               // keeps the + 1 inside the first v_pk_mad_u16 (the optimiser would peel it off again)
    const u16x2 x1 = a * fa + __builtin_bit_cast(u16x2, t);
    return (u16x2)((x1 + (x1 >> 8)) >> 8);
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

// the same with the per-step factors already packed by the launcher: fa_pk[k] = {f, f}, fb_pk[k] = {255 - f, 255 - f} as u16 x 2, f taken
// from the side the running composite sits on -- no select, no subtract in the kernel, the factors are SGPR operands of the packed ops
template <int NW>
__device__ __forceinline__ void chain_eval_pk(uint32_t (&v)[NW], const uint32_t (*L)[NW], uint32_t n_src, const uint32_t* fa_pk, const uint32_t* fb_pk) {
    Px4 acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = px4_unpack(L[0][w]);
#pragma unroll
    for (int k = 1; k < MX_CHAIN_MAX_SRC; ++k) {
        if (k < (int)n_src) {
            const u16x2 fa = __builtin_bit_cast(u16x2, fa_pk[k - 1]), fb = __builtin_bit_cast(u16x2, fb_pk[k - 1]);
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[w] = fade_px4(acc[w], px4_unpack(L[k][w]), fa, fb);
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = px4_pack(acc[w]);
}

// ---- per-pixel alpha (BUILD-SPECIFIED, mx_video.hpp ChainAlpha): the same steps with a per-sample factor pair ----
// x / 255 for x <= 255 * 255 in packed u16 (the identity fade_pk uses, tests/test_fastdiv.py)
__device__ __forceinline__ u16x2 div255_pk(u16x2 x) {
    const u16x2 one = {1, 1};
    const u16x2 x1 = x + one;
    return (u16x2)((x1 + (x1 >> 8)) >> 8);
}
// One step with coverage.  v: the running composite (opaque unless it is still the bare base layer: av != nullptr), o / ao: the other layer and its coverage
// (ao == nullptr: opaque), F = (fader * 255) as u8, via: the running composite sits on input A.
//   wa = (aA F) / 255;  wb = (aB (255 - wa)) / 255;  out = (A (255 - wb) + B wb) / 255
template <int NW>   // (arrays by reference and flags, never pointers: a register array whose address is taken lives in scratch memory)
__device__ __forceinline__ void chain_step_alpha(Px4 (&acc)[NW], const uint32_t (&o)[NW], const uint32_t (&ao_)[NW], const bool ao, const uint32_t (&av_)[NW], const bool av,
                                                 const uint32_t F, const bool via) {
    const u16x2 k255 = {255, 255};
    const unsigned short f = (unsigned short)(via ? F : 255u - F), g = (unsigned short)(255u - f);   // nominal factors of the running composite and of the other layer
    const u16x2 gp = {g, g}, Fp = {(unsigned short)F, (unsigned short)F};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        u16x2 wo_e, wo_o;
        if (!av) {   // opaque running composite: the other layer's factor is its nominal one scaled by its coverage, the composite takes the rest
            const Px4 a = px4_unpack(ao_[w]);
            wo_e = div255_pk(a.e * gp); wo_o = div255_pk(a.o * gp);   // (x + 1 rides on the multiply: one v_pk_mad_u16)
        } else {
            const Px4 a_v = px4_unpack(av_[w]);
            Px4 a_o; a_o.e = k255; a_o.o = k255;
            if (ao) a_o = px4_unpack(ao_[w]);
            if (via) {   // A = running composite, B = other
                const u16x2 wa_e = div255_pk(a_v.e * Fp), wa_o = div255_pk(a_v.o * Fp);
                wo_e = div255_pk(a_o.e * (k255 - wa_e)); wo_o = div255_pk(a_o.o * (k255 - wa_o));
            } else {     // A = other, B = running composite
                const u16x2 wa_e = div255_pk(a_o.e * Fp), wa_o = div255_pk(a_o.o * Fp);
                wo_e = k255 - div255_pk(a_v.e * (k255 - wa_e)); wo_o = k255 - div255_pk(a_v.o * (k255 - wa_o));
            }
        }
        const Px4 ov = px4_unpack(o[w]);
        acc[w].e = fade_pk(acc[w].e, ov.e, k255 - wo_e, wo_e);
        acc[w].o = fade_pk(acc[w].o, ov.o, k255 - wo_o, wo_o);
    }
}
// the chain with coverage planes: A[k] holds layer k's coverage in the layout of L[k] (only for layers whose bit is set in `mask`)
template <int NW>
__device__ __forceinline__ void chain_eval_alpha(uint32_t (&v)[NW], const uint32_t (*L)[NW], const uint32_t (*A)[NW], uint32_t n_src,
                                                 const uint32_t* fade, const uint32_t* v_is_a, const uint32_t mask) {
    Px4 acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = px4_unpack(L[0][w]);
#pragma unroll
    for (int k = 1; k < MX_CHAIN_MAX_SRC; ++k) {
        if (k < (int)n_src) {
            const uint32_t F = fade[k - 1] & 0xffu; const bool via = v_is_a[k - 1] != 0u;
            const bool oa = (mask >> k) & 1u, va = k == 1 && (mask & 1u);     // wave-uniform: the mask is a kernel argument
            if (!oa && !va) {
                const unsigned short f = (unsigned short)(via ? F : 255u - F), g = (unsigned short)(255u - f);
                const u16x2 fa = {f, f}, fb = {g, g};
#pragma unroll
                for (int w = 0; w < NW; ++w) acc[w] = fade_px4(acc[w], px4_unpack(L[k][w]), fa, fb);
            } else {
                chain_step_alpha<NW>(acc, L[k], A[k], oa, A[0], va, F, via);
            }
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = px4_pack(acc[w]);
}

template <bool AL>
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
    if constexpr (!AL) {
        chain_eval<4>(v, L, args.n_src, args.fade, args.v_is_a);
    } else {
        // the coverage of this chunk's 16 samples: luma -- 16 bytes of the plane's row; chroma -- the co-sited luma samples (2x, 2y): the even bytes of 32
        uint32_t A[MX_CHAIN_MAX_SRC][4];
#pragma unroll
        for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
            A[k][0] = A[k][1] = A[k][2] = A[k][3] = 0xffffffffu;
            if (k < (int)args.n_src && ((args.alpha_mask >> k) & 1u)) {
                if (plane == 0) {
                    const uint4 t = *reinterpret_cast<const uint4*>(args.al[k].p + (size_t)row * args.al[k].stride + col);
                    A[k][0] = t.x; A[k][1] = t.y; A[k][2] = t.z; A[k][3] = t.w;
                } else {
                    const uint8_t* ar = args.al[k].p + (size_t)(2u * row) * args.al[k].stride + 2u * col;
                    const uint4 a = *reinterpret_cast<const uint4*>(ar), b = *reinterpret_cast<const uint4*>(ar + 16);
                    A[k][0] = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u); A[k][1] = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
                    A[k][2] = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u); A[k][3] = __builtin_amdgcn_perm(b.w, b.z, 0x06040200u);
                }
            }
        }
        chain_eval_alpha<4>(v, L, A, args.n_src, args.fade, args.v_is_a, args.alpha_mask);
    }
    *reinterpret_cast<uint4*>(args.out[plane] + (size_t)row * args.out_stride[plane] + col) = make_uint4(v[0], v[1], v[2], v[3]);
}
void launch_fade_chain(const ChainArgs& a, hipStream_t s) {
    flush_scales(s);   // queued scaler output may be among the layers
    const uint32_t total = a.chunks[0] + a.chunks[1] + a.chunks[2];
    if (!total) return;
    if (a.alpha_mask) hipLaunchKernelGGL(k_fade_chain<true>, dim3((total + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_fade_chain<false>, dim3((total + 255) / 256), dim3(256), 0, s, a);
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
// Matrix mode 3: the 3 x 4 matrix in f32, two pixels per v_pk_fma_f32.  EXACT, not approximate: R, G, B are integers 0 .. 255, the
// coefficients are m / 4096 (a power-of-two scaling: no rounding) and the launcher takes this mode only when, per row,
// 2 * 255 * (|m0| + |m1| + |m2|) + |2 (m3 + 2048) - 4095| < 2^24 -- every partial sum is then an integer multiple of 2^-13 below 2^11
// in magnitude, representable in f32's 24 bits, so each fused multiply-add returns its exact result.  The constant term is
// (m3 + 2048) / 4096 - 1/2 + 2^-13: v_cvt_pk_u8_f32 rounds to nearest-even and saturates to 0 .. 255 (probed on the chip), and a
// multiple of 2^-12 minus 1/2 plus 2^-13 is never a tie and rounds to the floor -- the integer path's clip8(sum >> 12).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void yuv_px_pair_f32(const float* mf, int Y0, int Y1, int U, int V, uint32_t& p0, uint32_t& p1) {
    const int D = U - 128, E = V - 128;
    const int cr = 459 * E + 128, cg = -55 * D - 136 * E + 128, cb = 541 * D + 128;
    const int y0 = 298 * (Y0 - 16), y1 = 298 * (Y1 - 16);
    const uint32_t rg0 = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y0 + cr, y0 + cg, 8);
    const uint32_t rg1 = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y1 + cr, y1 + cg, 8);
    const uint32_t bb = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y0 + cb, y1 + cb, 8);
    const f32x2 R = {(float)(rg0 & 0xffu), (float)(rg1 & 0xffu)}, G = {(float)((rg0 >> 8) & 0xffu), (float)((rg1 >> 8) & 0xffu)};
    const f32x2 B = {(float)(bb & 0xffu), (float)((bb >> 8) & 0xffu)};
    f32x2 o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x2 c = {mf[4 * i + 3], mf[4 * i + 3]}, m0 = {mf[4 * i], mf[4 * i]}, m1 = {mf[4 * i + 1], mf[4 * i + 1]}, m2 = {mf[4 * i + 2], mf[4 * i + 2]};
        o[i] = __builtin_elementwise_fma(m0, R, __builtin_elementwise_fma(m1, G, __builtin_elementwise_fma(m2, B, c)));
    }
    p0 = __builtin_amdgcn_cvt_pk_u8_f32(o[2].x, 2u, __builtin_amdgcn_cvt_pk_u8_f32(o[1].x, 1u, __builtin_amdgcn_cvt_pk_u8_f32(o[0].x, 0u, 0xff000000u)));
    p1 = __builtin_amdgcn_cvt_pk_u8_f32(o[2].y, 2u, __builtin_amdgcn_cvt_pk_u8_f32(o[1].y, 1u, __builtin_amdgcn_cvt_pk_u8_f32(o[0].y, 0u, 0xff000000u)));
}
// Matrix mode 4 (EXPERIMENT, MX_VIDEO_MFMA_MATRIX=1; DESIGN.md "Colour matrix on the matrix cores"): the Q12 3 x 4 matrix as v_mfma_i32_4x4x4_16b_i8 -- sixteen independent
// 4 x 4 x 4 i8 products per instruction, a block = four neighbouring lanes.  Lane (block, j) supplies column j of B -- ITS OWN pixel as the signed bytes
// (R - 128, G - 128, B - 128, 0) -- and row j of A -- row j of the coefficient matrix -- and receives D[0..3][j]: the three output channels of its own pixel
// (layout probed on the chip, tools/mfma_probe.hip).  Integer, hence exact: m = 256 mh + ml with ml in [-128, 127] and mh an i8 (the launcher takes this mode only
// when every coefficient is below 2^15 - 128 in magnitude): sum m v = 256 sum mh x + sum ml x + 128 sum m with x = v - 128; the constant (m3 + 2048 + 128 sum m) enters
// through the accumulator.  Per pixel: 2 MFMA + 3 shift-adds + the same clip-and-pack as the integer modes.
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct MfmaMatrix { int a_hi, a_lo, c0, c1, c2; };   // this lane's rows of mh / ml (bytes k = 0 .. 2, byte 3 zero) and the three constants
__device__ __forceinline__ uint32_t mx_px_mfma(const MfmaMatrix& q, uint32_t rgb_s /* (R, G, B, 0) - 128 as bytes */) {
    const i32x4 z = {0, 0, 0, 0};
    const i32x4 h = __builtin_amdgcn_mfma_i32_4x4x4i8(q.a_hi, (int)rgb_s, z, 0, 0, 0);
    const i32x4 c = {(h.x << 8) + q.c0, (h.y << 8) + q.c1, (h.z << 8) + q.c2, 0};
    const i32x4 d = __builtin_amdgcn_mfma_i32_4x4x4i8(q.a_lo, (int)rgb_s, c, 0, 0, 0);
    return pack_rgba(d.x, d.y, d.z, 12);
}
__device__ __forceinline__ void yuv_px_pair_mfma(const MfmaMatrix& q, int Y0, int Y1, int U, int V, uint32_t& p0, uint32_t& p1) {
    const int D = U - 128, E = V - 128;
    const int cr = 459 * E + 128, cg = -55 * D - 136 * E + 128, cb = 541 * D + 128;
    const int y0 = 298 * (Y0 - 16), y1 = 298 * (Y1 - 16);
    const uint32_t rg0 = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y0 + cr, y0 + cg, 8);
    const uint32_t rg1 = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y1 + cr, y1 + cg, 8);
    const uint32_t bb = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(y0 + cb, y1 + cb, 8);
    p0 = mx_px_mfma(q, __builtin_amdgcn_perm(bb, rg0, 0x0c040100u) ^ 0x00808080u);    // (R0, G0, B0, 0)
    p1 = mx_px_mfma(q, __builtin_amdgcn_perm(bb, rg1, 0x0c050100u) ^ 0x00808080u);    // (R1, G1, B1, 0)
}
template <int MM>   // matrix mode: 0 none, 1 full 32-bit products, 2 24-bit products (3: yuv_px_pair_f32, 4: yuv_px_pair_mfma)
__device__ __forceinline__ uint32_t yuv_px(const int* m, int Y, int U, int V) {
    const int C = Y - 16, D = U - 128, E = V - 128;
    const int rs = 298 * C + 459 * E + 128, gs = 298 * C - 55 * D - 136 * E + 128, bs = 298 * C + 541 * D + 128;
    if (MM == 0) return pack_rgba(rs, gs, bs, 8);
    const int R = clip8c(rs >> 8), G = clip8c(gs >> 8), B = clip8c(bs >> 8);
    if (MM == 2) return pack_rgba(mx_row24(m, R, G, B), mx_row24(m + 4, R, G, B), mx_row24(m + 8, R, G, B), 12);
    return pack_rgba(m[0] * R + m[1] * G + m[2] * B + m[3] + 2048, m[4] * R + m[5] * G + m[6] * B + m[7] + 2048,
                     m[8] * R + m[9] * G + m[10] * B + m[11] + 2048, 12);
}

#define SC_TW 128
#define SC_TH 32
// ---- the 4-tap two-pass resampler on a 128 x 32 tile (shared by the stand-alone tiled scaler and the chain kernel's inline layers) ----
// Arithmetic of DESIGN.md "Scaler":  H pass t = (sum hc*S + 64) >> 7,  V pass D = clip8((sum vc*t + 2^20) >> 21), Q14 taps that sum to
// 16384.  The kernels evaluate the SAME integers with gfx950's packed dot products (a wave64 VALU instruction occupies the 16-lane
// SIMD for four cycles: the instruction count is what these latency-sized kernels pay for):
//   H  the window is staged as s' = S - 128 (bytes ^ 0x80, signed); a tap c = 256 ch + cl with cl in [-128, 127] (host-packed, i8 x 4):
//        sum c S = 256 dot4(s', ch) + dot4(s', cl) + 128 * 16384    -- two v_dot4_i32_i8 instead of four unpack + multiply-add pairs;
//      the H-filtered value is kept as t' = t - 16384, an i16 (the host checks the range from the actual taps).
//   V  rows of t' are stored in PAIRS: T2[p] holds (t'[p], t'[p+1]) per column, for every p, so the four taps of an output row whose
//      first tap row is vf are the two dwords T2[vf], T2[vf+2]:  sum vc t = dot2(T2[vf], vc01) + dot2(T2[vf+2], vc23) + 16384 * 16384
//      -- two v_dot2_i32_i16 per pixel.  Every intermediate stays below 2^31 (|vc| < 2^15, |t'| < 2^15, four terms).
// Host side: ScaleTables::lean holds the packed taps; contexts whose taps do not fit use the gather kernel.
// first tap of output o (DESIGN.md "Scaler"): ((floor((2o+1) src 65536 / (2 dst)) - 32768) >> 16) - 1  ==  floor(((2o+1) src + dst) / (2 dst)) - 2
// (the -32768 is half a source sample; adding 2 dst keeps the numerator positive).  32-bit: frames are at most 16384 wide.  The quotient
// is an f32 estimate (relative error < 2^-22, quotient < 2^15: off by at most one) corrected with the exact remainder.
__host__ __device__ __forceinline__ int sc_first_tap(uint32_t o, uint32_t src, uint32_t dst, float rcp2d /* 1.0f / (2 dst), IEEE */) {
    const uint32_t n = (2u * o + 1u) * src + dst, d = 2u * dst;
    uint32_t q = (uint32_t)((float)n * rcp2d);
    const int32_t r = (int32_t)(n - q * d);
    q += (r >= (int32_t)d) ? 1u : 0u;
    q -= (r < 0) ? 1u : 0u;
    return (int)q - 2;
}
bool scale_tile_origins_match(uint32_t src, uint32_t dst, const int32_t* first /* host copy of the tap table */) {
    const float rcp = 1.0f / (float)(2u * dst);
    for (uint32_t o = 0; o < dst; ++o)     // every output: the inline resampler's windows start wherever the letterbox offset puts them
        if (sc_first_tap(o, src, dst, rcp) != first[o]) return false;
    return true;
}
struct ScWin { int cxa, nc4, ry0, nr; };   // source window of a tile part: columns [cxa, cxa + 4 nc4) (cxa on a 16-byte boundary), rows [ry0, ry0 + nr)
__device__ __forceinline__ ScWin sc_window(int oxa, int oxb, int oya, int oyb, uint32_t sw, uint32_t dw, uint32_t sh, uint32_t dh) {   // outputs [oxa, oxb) x [oya, oyb)
    const float rh = 1.0f / (float)(2u * dw), rv = 1.0f / (float)(2u * dh);
    const int cx0 = sc_first_tap((uint32_t)oxa, sw, dw, rh), cxl = sc_first_tap((uint32_t)(oxb - 1), sw, dw, rh);
    const int ry0 = sc_first_tap((uint32_t)oya, sh, dh, rv), ryl = sc_first_tap((uint32_t)(oyb - 1), sh, dh, rv);
    ScWin w; w.cxa = cx0 & ~15; w.nc4 = (cxl + 4 - w.cxa + 3) >> 2; w.ry0 = ry0; w.nr = ryl + 4 - ry0;
    return w;
}
// stage the window as signed bytes s' = S - 128 (edge replication here).  16 lanes per window row, 16 bytes each: slot q of a lane is
// row (tid >> 4) + 16 q, chunk tid & 15; the window starts on a 16-byte boundary of the source row (ScWin::cxa).  All interior
// chunks of a lane are requested back to back (no control flow between the loads: a loop that loads and stores per iteration pays one
// memory round trip per iteration); border chunks -- only in tiles at the picture's edges -- are assembled from clamped byte loads
// afterwards.
template <int Q>
__device__ __forceinline__ void sc_stage(const uint8_t* src, uint32_t src_stride, int sw, int sh, const ScWin& g, uint8_t* S, int s_stride, int tid,
                                         const uint32_t sxs = 0u /* samples are 1 + sxs bytes apart ... */, const uint32_t sxo = 0u /* ... from byte sxo of a row (nv12 chroma) */,
                                         const int row_lo = 0, const int row_hi = 0x7fffffff /* rows that exist behind `src` (a row band's slice); the slots load Q * 16 rows whether the tile needs them or not */) {
    const int sw1 = sw - 1, sh1 = min(sh - 1, row_hi), sh0 = max(0, row_lo);
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | src_stride) & 15u) == 0 && sw >= 16;
    const int c16 = tid & 15, x = g.cxa + 16 * c16;
    const bool col_ok = 4 * c16 < g.nc4, interior = aligned && x >= 0 && x + 15 <= sw1;
    const int xs = aligned ? (min(max(x, 0), sw1 - 15) & ~15) : 0;                // always a readable chunk of the row when `aligned`
    uint4 w[Q];
    if (sxs == 0u) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = (tid >> 4) + 16 * q;
            const uint8_t* row = src + (size_t)min(max(g.ry0 + r, sh0), sh1) * src_stride;
            w[q] = aligned ? *reinterpret_cast<const uint4*>(row + xs) : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {   // interleaved samples: 16 of them are the even (sxo = 0) or odd (1) bytes of 32 source bytes -- two loads and four byte permutes
        const uint32_t sel = sxo ? 0x07050301u : 0x06040200u;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = (tid >> 4) + 16 * q;
            const uint8_t* row = src + (size_t)min(max(g.ry0 + r, sh0), sh1) * src_stride;
            uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
            if (aligned) { a = *reinterpret_cast<const uint4*>(row + 2 * xs); b = *reinterpret_cast<const uint4*>(row + 2 * xs + 16); }
            w[q] = make_uint4(__builtin_amdgcn_perm(a.y, a.x, sel), __builtin_amdgcn_perm(a.w, a.z, sel), __builtin_amdgcn_perm(b.y, b.x, sel), __builtin_amdgcn_perm(b.w, b.z, sel));
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int r = (tid >> 4) + 16 * q;
        if (r < g.nr && col_ok) {
            if (!interior) {                                                    // edge replication
                const uint8_t* row = src + (size_t)min(max(g.ry0 + r, sh0), sh1) * src_stride + sxo;
                const int st = 1 + (int)sxs;
                uint32_t d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    d[k] = (uint32_t)row[st * min(max(x + 4 * k, 0), sw1)] | ((uint32_t)row[st * min(max(x + 4 * k + 1, 0), sw1)] << 8) |
                           ((uint32_t)row[st * min(max(x + 4 * k + 2, 0), sw1)] << 16) | ((uint32_t)row[st * min(max(x + 4 * k + 3, 0), sw1)] << 24);
                w[q] = make_uint4(d[0], d[1], d[2], d[3]);
            }
            *reinterpret_cast<uint4*>(S + (size_t)r * s_stride + 16 * c16) =
                make_uint4(w[q].x ^ 0x80808080u, w[q].y ^ 0x80808080u, w[q].z ^ 0x80808080u, w[q].w ^ 0x80808080u);
        }
    }
}
// H pass of tile column i over window rows: T2[p][i] = (t'[p], t'[p+1]) for p in [p0, p1) -- a thread walks its rows in order, so every
// H-filtered value costs one dot-product group and one LDS store of the pair it closes.
// One H-filtered value t' = t - 16384 from the four staged bytes w (low 16 bits are what the callers keep): (256 hi + lo + 64) >> 7 == 2 hi + ((lo + 64) >> 7) exactly
// -- 256 hi is a multiple of 128 -- so the two dot products do not wait for each other, and there is no shift-and-or between them.  `clamp` is never reached
// (|hi|, |lo| < 2^17): it is there because only the three-operand v_dot4_i32_i8 has it -- the compiler otherwise selects the accumulating v_dot4c with a
// v_mov of the addend in front (one more VALU instruction per value; the tile is issue-bound).
__device__ __forceinline__ uint32_t sc_h_value(const uint32_t w, const uint2 hpk) {
    const int hi = __builtin_amdgcn_sdot4((int)w, (int)hpk.x, 0, true);                        // sum ch s'
    const int lo = __builtin_amdgcn_sdot4((int)w, (int)hpk.y, 64, true);                       // sum cl s' + 64
    return (uint32_t)(2 * hi + (lo >> 7));
}
__device__ __forceinline__ uint32_t sc_hdot(const uint8_t* Srow, int hb, uint32_t sh8, const uint2 hpk) {   // t - 16384, low 16 bits valid
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(Srow + hb);
    return sc_h_value(__builtin_amdgcn_alignbyte(sp[1], sp[0], sh8), hpk);                     // s' of taps hf .. hf + 3
}
__device__ __forceinline__ void sc_hcol(const uint8_t* S, int s_stride, int hf /* first tap - cxa */, const uint2 hpk, uint32_t* T2, int t_cols, int i, int p0, int p1) {
    if (p0 >= p1) return;
    const int hb = hf & ~3; const uint32_t sh8 = (uint32_t)(hf & 3);
    const uint8_t* Srow = S + (size_t)p0 * s_stride;
    uint32_t* out = T2 + p0 * t_cols + i;
    uint32_t prev = sc_hdot(Srow, hb, sh8, hpk);
#pragma unroll 2
    for (int p = p0; p < p1; ++p) {
        Srow += s_stride;
        const uint32_t cur = sc_hdot(Srow, hb, sh8, hpk);
        *out = __builtin_amdgcn_perm(cur, prev, 0x05040100u);                                  // (t'[p] & 0xffff) | (t'[p+1] << 16)
        out += t_cols; prev = cur;
    }
}
typedef short sc_s2 __attribute__((ext_vector_type(2)));
// V pass: four pixels at tile columns col .. col + 3 of the output row whose first tap row (relative to the window) is vf
__device__ __forceinline__ uint32_t sc_vquad(const uint32_t* T2, int t_cols, int vf, int col, const uint2 vpk) {
    const uint32_t* P = T2 + vf * t_cols + col;
    const uint4 p0 = *reinterpret_cast<const uint4*>(P), p1 = *reinterpret_cast<const uint4*>(P + 2 * t_cols);
    const sc_s2 c01 = __builtin_bit_cast(sc_s2, vpk.x), c23 = __builtin_bit_cast(sc_s2, vpk.y);
    const int K = 16384 * 16384 + (1 << 20);
    // (clamp: never reached -- the host bounds the sum below 2^31 -- and only the three-operand v_dot2_i32_i16 has it: no v_mov of K per pixel, see sc_h_value)
    auto px = [&](uint32_t a, uint32_t b) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(sc_s2, b), c23, __builtin_amdgcn_sdot2(__builtin_bit_cast(sc_s2, a), c01, K, true), true); };
    // clip8(sum >> 21) x 4 -> one dword: two v_ashr_pk_u8_i32 (explicit builtin, see pack_rgba)
    const uint32_t lo = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(px(p0.x, p1.x), px(p0.y, p1.y), 21);
    const uint32_t hi = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(px(p0.z, p1.z), px(p0.w, p1.w), 21);
    return lo | (hi << 16);
}

// ---- inline resampling of chain layers (ChainScale) ----
// A block of the RGBA chain owns a 128 x 32 luma tile (64 x 16 chroma).  Per scaled layer: the source windows of the three planes are
// staged in LDS, the H pass filters every window row once per tile column, the V pass leaves each lane's own pixels in registers in
// the layout the chain evaluates (L[k][0..5]).  Pixels outside the scaled picture (letterbox bars) are the blank constant
// (encode.rs:382-396).
#define CS_SY_STRIDE 160
#define CS_SY_ROWS 36
#define CS_SC_STRIDE 96
#define CS_SC_ROWS 20
#define CS_S_BYTES (CS_SY_ROWS * CS_SY_STRIDE + 2 * CS_SC_ROWS * CS_SC_STRIDE)              /* 9600 per scaled layer */
#define CS_TY_DW (CS_SY_ROWS * 128)                                                        /* row pairs (p, p + 1), dwords */
#define CS_TC_DW (CS_SC_ROWS * 64)
#define CS_T_BYTES ((CS_TY_DW + 2 * CS_TC_DW) * 4)                                          /* 28672, shared by the layers */
struct CsGeo { int xa, xb, ya, yb; ScWin w; };   // in-tile part of the scaled picture (plane coordinates) and its source window
__device__ __forceinline__ CsGeo cs_geo(const ChainScale& s, int c, int X0, int Y0, int TW, int TH) {
    CsGeo g;
    const int lx = (int)s.lx[c], ly = (int)s.ly[c];
    g.xa = max(X0, lx); g.xb = min(X0 + TW, lx + (int)s.dw[c]);
    g.ya = max(Y0, ly); g.yb = min(Y0 + TH, ly + (int)s.dh[c]);
    g.w.cxa = 0; g.w.nc4 = 0; g.w.ry0 = 0; g.w.nr = 0;
    if (g.xa < g.xb && g.ya < g.yb) g.w = sc_window(g.xa - lx, g.xb - lx, g.ya - ly, g.yb - ly, s.sw[c], s.dw[c], s.sh[c], s.dh[c]);
    return g;
}
__device__ __forceinline__ uint32_t cs_bytes_below(int n) { return n <= 0 ? 0u : (n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u)); }
// keep the bytes of `quad` whose pixel x0 + b lies in [xa, xb); the others are `blank`
__device__ __forceinline__ uint32_t cs_mask_quad(uint32_t quad, uint32_t blank, int x0, int xa, int xb) {
    const uint32_t m = cs_bytes_below(xb - x0) & ~cs_bytes_below(xa - x0);
    return (quad & m) | (blank & ~m);
}

template <int MM, bool SC, bool AL, class ArgsRef>   // ArgsRef: ChainRgbaArgs in kernel arguments, or in the constant address space (k_video_batch); AL: layers with coverage planes
__device__ __forceinline__ void chain_rgba_tile(ArgsRef& a, const int bx, const int by) {
    static_assert(!(SC && AL), "a chain with coverage planes takes its scaled layers materialised");
    // a lane owns a UNIT of 8 pixels x 2 rows and their 4 + 4 chroma samples.  With inline-scaled layers (SC) a block is a tile of
    // 128 x 32 luma pixels (the scaler windows in LDS are per tile).  Without them nothing ties a block to a rectangle, and a block
    // takes 256 CONSECUTIVE units in row-major order (bx = the block's index, by unused): a wave's loads are 512 contiguous bytes of
    // one luma row (256 of a chroma row) instead of 128 (64) bytes of four rows each -- whole DRAM bursts and cache lines per request,
    // no half-used chroma lines, and no idle lanes at the right edge whatever the width.
    const int tid = threadIdx.x;
    const int cb = tid & 15, rp = tid >> 4;
    const int X0 = SC ? bx * 128 : 0, Y0 = SC ? by * 32 : 0;
    uint32_t xb, yb;
    if constexpr (SC) {
        xb = (uint32_t)(X0 / 8 + cb);     // 8-pixel column block
        yb = (uint32_t)(Y0 / 2 + rp);     // row pair
    } else {
        const uint32_t upr = (a.width + 7u) >> 3, idx = (uint32_t)bx * 256u + (uint32_t)tid;
        uint32_t q = __umulhi(idx, a.upr_magic);      // idx / upr or one less (upr_magic = floor(2^32 / upr), idx < 2^32 / upr)
        q += (idx - q * upr >= upr) ? 1u : 0u;
        yb = q; xb = idx - q * upr;
    }
    const bool valid = xb * 8 < a.width && yb * 2 < a.height;
    // Coverage planes FIRST (AL): which layers carry one is only known at run time, so these loads sit in (wave-uniform) branches -- issued before the layers'
    // unconditional burst they cost their issue slots and nothing else: memory returns in order, so by the time the first layer dword is waited for (the
    // counted waits below count only what was issued AFTER it) every coverage dword is there.  Raw row dwords here; the chroma selection happens at the use.
    uint32_t A[AL ? MX_CHAIN_MAX_SRC : 1][6];
    if constexpr (AL) {
        const uint32_t mask = a.alpha_mask;
        const uint32_t xc = min(xb, (a.width - 1u) >> 3), yc = min(yb, (a.height - 1u) >> 1);
#pragma unroll
        for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
#pragma unroll
            for (int w = 0; w < 6; ++w) A[k][w] = 0xffffffffu;
            if ((mask >> k) & 1u) {
                const auto& al = a.al[k];
                const uint32_t o0 = __umul24(2u * yc, al.stride) + xc * 8u;
                const uint2 r0 = *reinterpret_cast<const uint2*>(al.p + (size_t)o0);
                const uint2 r1 = *reinterpret_cast<const uint2*>(al.p + (size_t)(o0 + al.stride));
                A[k][0] = r0.x; A[k][1] = r0.y; A[k][2] = r1.x; A[k][3] = r1.y;
            }
        }
    }
    // per source: 2 dwords of Y for each of the two rows, one dword of U, one of V  (6 dwords)
    uint32_t L[MX_CHAIN_MAX_SRC][6];
    if constexpr (!SC) {
        // Every one of the 8 x 4 loads is issued unconditionally, back to back, with no control flow in between: the launcher points
        // missing layers (None inputs, sources beyond n_src) at a blank row with stride 0 (chain_prepare), lanes outside the picture
        // read the last column block / row pair.  The chain then consumes the layers in the order they were requested, so the waits
        // count DOWN (s_waitcnt vmcnt(28), (24), ...) and a wave fades layers 0 .. k while k + 1 .. 7 are still on their way.
        const uint32_t xc = min(xb, (a.width - 1u) >> 3), yc = min(yb, (a.height - 1u) >> 1);
#pragma unroll
        for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
            // 32-bit offsets from the (scalar) plane base: one full-rate 24-bit multiply-add per plane (rows and strides are far below
            // 2^24, a frame far below 4 GB): the loads take the saddr + voffset form
            const auto& s = a.src[k];
            const uint32_t o0 = __umul24(2u * yc, s.stride[0]) + xc * 8u;
            const uint2 r0 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)o0);
            const uint2 r1 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)(o0 + s.stride[0]));
            L[k][0] = r0.x; L[k][1] = r0.y; L[k][2] = r1.x; L[k][3] = r1.y;
            L[k][4] = *reinterpret_cast<const uint32_t*>(s.p[1] + (size_t)(__umul24(yc, s.stride[1]) + xc * 4u));
            L[k][5] = *reinterpret_cast<const uint32_t*>(s.p[2] + (size_t)(__umul24(yc, s.stride[2]) + xc * 4u));
        }
    } else {
#pragma unroll
    for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) {
        L[k][0] = L[k][1] = L[k][2] = L[k][3] = 0u; L[k][4] = L[k][5] = 0x80808080u;
        if (k < (int)a.n_src && valid) {
            const auto& s = a.src[k];
            if (s.p[0]) {
                const uint2 r0 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)(2 * yb) * s.stride[0] + xb * 8);
                const uint2 r1 = *reinterpret_cast<const uint2*>(s.p[0] + (size_t)(2 * yb + 1) * s.stride[0] + xb * 8);
                L[k][0] = r0.x; L[k][1] = r0.y; L[k][2] = r1.x; L[k][3] = r1.y;
            }
            if (s.p[1]) L[k][4] = *reinterpret_cast<const uint32_t*>(s.p[1] + (size_t)yb * s.stride[1] + xb * 4);
            if (s.p[2]) L[k][5] = *reinterpret_cast<const uint32_t*>(s.p[2] + (size_t)yb * s.stride[2] + xb * 4);
        }
    }
    }
    if constexpr (SC) {
        extern __shared__ __attribute__((aligned(16))) uint8_t cs_smem[];
        uint32_t* const Ty = reinterpret_cast<uint32_t*>(cs_smem);       // [36][128] pairs of H-filtered rows (p, p + 1)
        uint32_t* const Tu = Ty + CS_TY_DW;                               // [20][64]
        uint32_t* const Tv = Tu + CS_TC_DW;
        uint8_t* const S0 = cs_smem + CS_T_BYTES;
        CsGeo gy[MX_CHAIN_MAX_SCALED], gc[MX_CHAIN_MAX_SCALED];
        // every table entry a lane will need, requested before anything waits
        uint2 hpY[MX_CHAIN_MAX_SCALED], hpC[MX_CHAIN_MAX_SCALED], vpY[MX_CHAIN_MAX_SCALED][2], vpC[MX_CHAIN_MAX_SCALED];
        int hfY[MX_CHAIN_MAX_SCALED], hfC[MX_CHAIN_MAX_SCALED], vfY[MX_CHAIN_MAX_SCALED][2], vfC[MX_CHAIN_MAX_SCALED];
#pragma unroll
        for (int j = 0; j < MX_CHAIN_MAX_SCALED; ++j) {
            if (j < (int)a.n_scaled) {
                const ChainScale& s = a.sc[j];
                const int oxY = min(max(X0 + (tid & 127) - (int)s.lx[0], 0), (int)s.dw[0] - 1), oxC = min(max(X0 / 2 + (tid & 63) - (int)s.lx[1], 0), (int)s.dw[1] - 1);
                hpY[j] = s.hpk[0][oxY]; hfY[j] = s.hfirst[0][oxY];
                hpC[j] = s.hpk[1][oxC]; hfC[j] = s.hfirst[1][oxC];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int oy = min(max(Y0 + 2 * rp + r - (int)s.ly[0], 0), (int)s.dh[0] - 1);
                    vpY[j][r] = s.vpk[0][oy]; vfY[j][r] = s.vfirst[0][oy];
                }
                const int oyC = min(max(Y0 / 2 + rp - (int)s.ly[1], 0), (int)s.dh[1] - 1);
                vpC[j] = s.vpk[1][oyC]; vfC[j] = s.vfirst[1][oyC];
            }
        }
#pragma unroll
        for (int j = 0; j < MX_CHAIN_MAX_SCALED; ++j) {
            if (j < (int)a.n_scaled) {
                const ChainScale& s = a.sc[j];
                gy[j] = cs_geo(s, 0, X0, Y0, 128, 32);
                gc[j] = cs_geo(s, 1, X0 / 2, Y0 / 2, 64, 16);
                uint8_t* Sy = S0 + j * CS_S_BYTES; uint8_t* Su = Sy + CS_SY_ROWS * CS_SY_STRIDE; uint8_t* Sv = Su + CS_SC_ROWS * CS_SC_STRIDE;
                sc_stage<3>(s.src[0], s.src_stride[0], (int)s.sw[0], (int)s.sh[0], gy[j].w, Sy, CS_SY_STRIDE, tid);
                sc_stage<2>(s.src[1], s.src_stride[1], (int)s.sw[1], (int)s.sh[1], gc[j].w, Su, CS_SC_STRIDE, tid);
                sc_stage<2>(s.src[2], s.src_stride[2], (int)s.sw[1], (int)s.sh[1], gc[j].w, Sv, CS_SC_STRIDE, tid);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MX_CHAIN_MAX_SCALED; ++j) {
            if (j < (int)a.n_scaled) {
                const CsGeo Gy = gy[j], Gc = gc[j];
                const uint8_t* Sy = S0 + j * CS_S_BYTES; const uint8_t* Su = Sy + CS_SY_ROWS * CS_SY_STRIDE; const uint8_t* Sv = Su + CS_SC_ROWS * CS_SC_STRIDE;
                {   // H pass, luma: column i of the tile; the two threads of a column take half of the row pairs each
                    const int i = tid & 127, x = X0 + i, np = Gy.w.nr - 1, half = (np + 1) >> 1, part = tid >> 7;
                    if (x >= Gy.xa && x < Gy.xb)
                        sc_hcol(Sy, CS_SY_STRIDE, hfY[j] - Gy.w.cxa, hpY[j], Ty, 128, i, part * half, min(part * half + half, np));
                }
                {   // H pass, chroma (U and V share the taps): column i; threads 0-127 U, 128-255 V, two per column
                    const int i = tid & 63, x = X0 / 2 + i, np = Gc.w.nr - 1, half = (np + 1) >> 1, part = (tid >> 6) & 1;
                    if (x >= Gc.xa && x < Gc.xb)
                        sc_hcol((tid & 128) ? Sv : Su, CS_SC_STRIDE, hfC[j] - Gc.w.cxa, hpC[j], (tid & 128) ? Tv : Tu, 64, i, part * half, min(part * half + half, np));
                }
                __syncthreads();
                uint32_t R[6] = {0u, 0u, 0u, 0u, 0x80808080u, 0x80808080u};
                const bool edge_y = Gy.xa > X0 || Gy.xb < X0 + 128, edge_c = Gc.xa > X0 / 2 || Gc.xb < X0 / 2 + 64;   // block-uniform: a letterbox edge crosses the tile
#pragma unroll
                for (int r = 0; r < 2; ++r) {   // V pass, luma
                    const int y = Y0 + 2 * rp + r;
                    if (Gy.xa < Gy.xb && y >= Gy.ya && y < Gy.yb) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            uint32_t quad = sc_vquad(Ty, 128, vfY[j][r] - Gy.w.ry0, cb * 8 + 4 * q, vpY[j][r]);
                            if (edge_y) quad = cs_mask_quad(quad, 0u, X0 + cb * 8 + 4 * q, Gy.xa, Gy.xb);
                            R[2 * r + q] = quad;
                        }
                    }
                }
                {   // V pass, chroma
                    const int y = Y0 / 2 + rp;
                    if (Gc.xa < Gc.xb && y >= Gc.ya && y < Gc.yb) {
                        uint32_t qu = sc_vquad(Tu, 64, vfC[j] - Gc.w.ry0, cb * 4, vpC[j]), qv = sc_vquad(Tv, 64, vfC[j] - Gc.w.ry0, cb * 4, vpC[j]);
                        if (edge_c) { qu = cs_mask_quad(qu, 0x80808080u, X0 / 2 + cb * 4, Gc.xa, Gc.xb); qv = cs_mask_quad(qv, 0x80808080u, X0 / 2 + cb * 4, Gc.xa, Gc.xb); }
                        R[4] = qu; R[5] = qv;
                    }
                }
                const uint32_t at = a.scaled_src[j];
#pragma unroll
                for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k)
                    if ((uint32_t)k == at) {
#pragma unroll
                        for (int w = 0; w < 6; ++w) L[k][w] = R[w];
                    }
                if (j + 1 < (int)a.n_scaled) __syncthreads();   // T is reused by the next layer
            }
        }
    }
    uint32_t v[6];
    uint32_t fa[MX_CHAIN_MAX_SRC - 1], fb[MX_CHAIN_MAX_SRC - 1]; int mtx[12]; float mtf[12];   // wave-uniform copies: SGPRs
#pragma unroll
    for (int k = 0; k < MX_CHAIN_MAX_SRC - 1; ++k) { fa[k] = a.fa_pk[k]; fb[k] = a.fb_pk[k]; }
#pragma unroll
    for (int k = 0; k < 12; ++k) { if (MM == 3) mtf[k] = a.mf[k]; else mtx[k] = a.m[k]; }
    MfmaMatrix mq{};
    if constexpr (MM == 4) {   // the launcher left the packed rows in mf[]: [0..3] rows of mh, [4..7] rows of ml, [8..10] the constants; this lane's row is lane & 3
        const int j = tid & 3;
        const int h0 = __builtin_bit_cast(int, a.mf[0]), h1 = __builtin_bit_cast(int, a.mf[1]), h2 = __builtin_bit_cast(int, a.mf[2]);
        const int l0 = __builtin_bit_cast(int, a.mf[4]), l1 = __builtin_bit_cast(int, a.mf[5]), l2 = __builtin_bit_cast(int, a.mf[6]);
        mq.a_hi = j == 0 ? h0 : (j == 1 ? h1 : (j == 2 ? h2 : 0));
        mq.a_lo = j == 0 ? l0 : (j == 1 ? l1 : (j == 2 ? l2 : 0));
        mq.c0 = __builtin_bit_cast(int, a.mf[8]); mq.c1 = __builtin_bit_cast(int, a.mf[9]); mq.c2 = __builtin_bit_cast(int, a.mf[10]);
    }
    if constexpr (AL) {
        // coverage of the unit's 8 x 2 luma samples (the layout of L[k][0..3]) and of its 4 chroma samples -- the co-sited luma samples (2x, 2y): the even
        // bytes of the upper row -- for U and V alike.  Only layers that carry a plane are read (the mask is wave-uniform).
        const uint32_t mask = a.alpha_mask;
        uint32_t fd[MX_CHAIN_MAX_SRC - 1], via[MX_CHAIN_MAX_SRC - 1];
#pragma unroll
        for (int k = 0; k < MX_CHAIN_MAX_SRC - 1; ++k) { fd[k] = a.fade[k]; via[k] = a.v_is_a[k]; }
        // the coverage of the 4 chroma samples -- the co-sited luma samples (2x, 2y): the even bytes of the upper row -- for U and V alike
#pragma unroll
        for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) A[k][4] = A[k][5] = __builtin_amdgcn_perm(A[k][1], A[k][0], 0x06040200u);
        chain_eval_alpha<6>(v, L, A, a.n_src, fd, via, mask);
    } else {
        chain_eval_pk<6>(v, L, a.n_src, fa, fb);
    }
    if constexpr (MM != 4) { if (!valid) return; }   // (mode 4: an MFMA is executed by the whole wave -- lanes outside the picture compute their clamped unit and store nothing)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t yrow = 2 * yb + r;
        if constexpr (MM != 4) { if (yrow >= a.height) break; }
        uint8_t* o = a.rgba + (size_t)yrow * a.rgba_stride + (size_t)xb * 32;
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {       // 2 groups of 4 pixels
            const uint32_t yw = v[2 * r + g4];
            const uint32_t cu = (v[4] >> (16 * g4)) & 0xffffu, cv = (v[5] >> (16 * g4)) & 0xffffu;
            uint32_t px[4];
            if constexpr (MM == 3) {
#pragma unroll
                for (int k = 0; k < 4; k += 2)
                    yuv_px_pair_f32(mtf, (int)((yw >> (8 * k)) & 0xff), (int)((yw >> (8 * k + 8)) & 0xff), (int)((cu >> (4 * k)) & 0xff), (int)((cv >> (4 * k)) & 0xff), px[k], px[k + 1]);
            } else if constexpr (MM == 4) {
#pragma unroll
                for (int k = 0; k < 4; k += 2)
                    yuv_px_pair_mfma(mq, (int)((yw >> (8 * k)) & 0xff), (int)((yw >> (8 * k + 8)) & 0xff), (int)((cu >> (4 * k)) & 0xff), (int)((cv >> (4 * k)) & 0xff), px[k], px[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    px[k] = yuv_px<MM>(mtx, (int)((yw >> (8 * k)) & 0xff), (int)((cu >> (8 * (k >> 1))) & 0xff), (int)((cv >> (8 * (k >> 1))) & 0xff));
            }
            const uint32_t x = xb * 8 + g4 * 4;
            if constexpr (MM == 4) { if (!valid || yrow >= a.height) continue; }
            if (x + 4 <= a.width) *reinterpret_cast<uint4*>(o + g4 * 16) = make_uint4(px[0], px[1], px[2], px[3]);
            else for (uint32_t k = 0; x + k < a.width; ++k) reinterpret_cast<uint32_t*>(o + g4 * 16)[k] = px[k];
        }
    }
}
template <int MM, bool SC, bool AL = false>
__global__ __launch_bounds__(256) void k_fade_chain_rgba(ChainRgbaArgs a, uint32_t tx, uint32_t n_tiles) {
    const uint32_t t = xcd_run(blockIdx.x, n_tiles);
    if constexpr (SC) chain_rgba_tile<MM, SC, false>(a, (int)(t % tx), (int)(t / tx));
    else chain_rgba_tile<MM, SC, AL>(a, (int)t, 0);
}

// A blank row per device (Y = 0x00, U = V = 0x80: what AvFrame::blank writes, frame.rs:128-132), 32 KB each, read with stride 0 by the
// branch-free chain tiles in place of a layer that is not there (video_mixer.rs:180-188 reads the blank output plane).  Never freed.
static const uint8_t* blank_row(int plane) {
    static std::mutex mu;
    static std::map<int, uint8_t*> rows;
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    std::lock_guard<std::mutex> lk(mu);
    auto it = rows.find(dev);
    if (it == rows.end()) {
        uint8_t* p = nullptr;
        hip_check(hipMalloc((void**)&p, 65536), "hipMalloc(blank rows)");
        hip_check(hipMemset(p, 0x00, 32768), "hipMemset");
        hip_check(hipMemset(p + 32768, 0x80, 32768), "hipMemset");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        it = rows.emplace(dev, p).first;
    }
    return it->second + (plane ? 32768 : 0);
}
static void chain_blank_planes(ChainRgbaArgs& a) {   // kernels without inline-scaled layers: every plane pointer is readable
    for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k)
        for (int p = 0; p < 3; ++p)
            if (k >= (int)a.n_src || !a.src[k].p[p]) { a.src[k].p[p] = blank_row(p); a.src[k].stride[p] = 0; }
    for (int k = (int)a.n_src; k < MX_CHAIN_MAX_SRC; ++k) { a.al[k].p = nullptr; a.al[k].stride = 0; a.al[k]._pad = 0; a.alpha_mask &= ~(1u << k); }
}
// what the launcher derives from the arguments once per launch: the matrix mode (24-bit products when every entry fits) and the
// per-step cross-fade factors as packed u16 pairs (chain_eval_pk)
static uint32_t chain_strip_blocks(const ChainRgbaArgs& a) {   // blocks of 256 units (8 pixels x 2 rows) in row-major order
    const uint64_t units = (uint64_t)((a.width + 7u) >> 3) * ((a.height + 1u) >> 1);
    return (uint32_t)((units + 255u) / 256u);
}
static int chain_matrix_mode(ChainRgbaArgs& a) {
    { const uint32_t upr = std::max(1u, (a.width + 7u) >> 3); a.upr_magic = (uint32_t)std::min<uint64_t>(0xffffffffull, (1ull << 32) / upr); a._pad0 = 0; }
    // A step whose factor for the running composite is 255 returns it unchanged -- (255 v + 0 o) / 255 = v exactly -- and one whose factor
    // is 0 returns the other layer exactly: a fader at either end of its travel (where a fader usually rests).  Such steps are dropped
    // here, bit for bit the same picture: the first kind leaves its layer unread, the second restarts the chain at its layer.
    // With coverage planes (ChainAlpha): a step whose nominal factor for the OTHER layer is 0 still leaves the running composite unchanged (w_other = (a 0) / 255
    // = 0) unless the composite is the bare base layer WITH coverage (its own coverage then decides); a step whose nominal factor for the other layer is 255
    // returns that layer exactly only when the layer is opaque.
    if (!a.n_scaled && a.n_src >= 2) {
        ChainSrc src[MX_CHAIN_MAX_SRC]; ChainAlpha al[MX_CHAIN_MAX_SRC]; uint32_t fade[MX_CHAIN_MAX_SRC - 1], via[MX_CHAIN_MAX_SRC - 1]; uint32_t n = 1;
        src[0] = a.src[0]; al[0] = (a.alpha_mask & 1u) ? a.al[0] : ChainAlpha{nullptr, 0, 0};
        for (uint32_t k = 1; k < a.n_src; ++k) {
            const uint32_t f = (a.v_is_a[k - 1] ? a.fade[k - 1] : 255u - a.fade[k - 1]) & 0xffu;   // the running composite's factor
            const bool other_alpha = ((a.alpha_mask >> k) & 1u) != 0, base_alpha = n == 1 && al[0].p != nullptr;
            if (f == 255u && !base_alpha) continue;
            if (f == 0u && !other_alpha) { src[0] = a.src[k]; al[0] = ChainAlpha{nullptr, 0, 0}; n = 1; continue; }
            src[n] = a.src[k]; al[n] = other_alpha ? a.al[k] : ChainAlpha{nullptr, 0, 0}; fade[n - 1] = a.fade[k - 1]; via[n - 1] = a.v_is_a[k - 1]; ++n;
        }
        a.alpha_mask = 0;
        for (uint32_t k = 0; k < n; ++k) { a.src[k] = src[k]; a.al[k] = al[k]; if (al[k].p) a.alpha_mask |= 1u << k; }
        for (uint32_t k = 0; k + 1 < n; ++k) { a.fade[k] = fade[k]; a.v_is_a[k] = via[k]; }
        for (uint32_t k = n; k < MX_CHAIN_MAX_SRC; ++k) { for (int pl = 0; pl < 3; ++pl) { a.src[k].p[pl] = nullptr; a.src[k].stride[pl] = 0; } a.al[k] = ChainAlpha{nullptr, 0, 0}; }
        for (uint32_t k = n - 1; k < MX_CHAIN_MAX_SRC - 1; ++k) { a.fade[k] = 0; a.v_is_a[k] = 1; }
        a.n_src = n;
    }
    for (int k = 0; k < MX_CHAIN_MAX_SRC - 1; ++k) {
        const uint32_t f = (a.v_is_a[k] ? a.fade[k] : 255u - a.fade[k]) & 0xffu, g = 255u - f;
        a.fa_pk[k] = f * 0x10001u; a.fb_pk[k] = g * 0x10001u;
    }
    if (!a.use_matrix) return 0;
    bool fits = true;
    for (int k = 0; k < 12; ++k) if ((k & 3) != 3 && (a.m[k] < -(1 << 23) || a.m[k] >= (1 << 23))) fits = false;
    a.use_matrix = fits ? 2 : 1;
    static const int no_f32 = env_int("MX_VIDEO_NO_F32_MATRIX", 0);
    bool small = fits && !no_f32;
    for (int i = 0; i < 3 && small; ++i) {
        const int64_t mag = 2 * 255 * (std::llabs((long long)a.m[4 * i]) + std::llabs((long long)a.m[4 * i + 1]) + std::llabs((long long)a.m[4 * i + 2]))
                            + std::llabs(2 * ((long long)a.m[4 * i + 3] + 2048) - 4095);
        if (mag >= (1ll << 24)) small = false;
    }
    if (small) {
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < 3; ++k) a.mf[4 * i + k] = (float)a.m[4 * i + k] / 4096.0f;
            a.mf[4 * i + 3] = (float)(2 * ((long long)a.m[4 * i + 3] + 2048) - 4095) / 8192.0f;
        }
        a.use_matrix = 3;
    }
    // the experiment: the matrix on the matrix cores (yuv_px_pair_mfma); read per call so one process can run both forms
    const char* const mf_env = getenv("MX_VIDEO_MFMA_MATRIX");
    if (mf_env && atoi(mf_env) != 0 && !a.alpha_mask && !a.n_scaled) {
        bool ok = true;
        for (int k = 0; k < 12; ++k) if ((k & 3) != 3 && (a.m[k] <= -(32768 - 128) || a.m[k] >= 32768 - 128)) ok = false;
        for (int i = 0; i < 3 && ok; ++i) {   // the accumulator stays in i32: |sum| <= 255 * 3 * 2^15 + |constant|
            const long long c = (long long)a.m[4 * i + 3] + 2048 + 128ll * ((long long)a.m[4 * i] + a.m[4 * i + 1] + a.m[4 * i + 2]);
            if (c > 0x3fffffffll || c < -0x3fffffffll) ok = false;
        }
        if (ok) {
            uint32_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0}; int32_t c0[4] = {0, 0, 0, 0};
            for (int i = 0; i < 3; ++i) {
                for (int k = 0; k < 3; ++k) {
                    const int32_t mm = a.m[4 * i + k], mh = (mm + 128) >> 8, ml = mm - 256 * mh;
                    hi[i] |= (uint32_t)(mh & 0xff) << (8 * k); lo[i] |= (uint32_t)(ml & 0xff) << (8 * k);
                }
                c0[i] = (int32_t)((long long)a.m[4 * i + 3] + 2048 + 128ll * ((long long)a.m[4 * i] + a.m[4 * i + 1] + a.m[4 * i + 2]));
            }
            for (int i = 0; i < 4; ++i) { std::memcpy(&a.mf[i], &hi[i], 4); std::memcpy(&a.mf[4 + i], &lo[i], 4); std::memcpy(&a.mf[8 + i], &c0[i], 4); }
            a.use_matrix = 4;
        }
    }
    return a.use_matrix;
}
void launch_fade_chain_rgba(const ChainRgbaArgs& a0, hipStream_t s) {
    flush_scales(s);
    if (!a0.width || !a0.height) return;
    ChainRgbaArgs a = a0;
    chain_matrix_mode(a);
    const uint32_t tx = (a.width + 127) / 128, n_tiles = tx * ((a.height + 31) / 32);
    const dim3 grid(n_tiles);
    if (a.n_scaled) {
        if (a.alpha_mask) throw Error(MX_ERR_INTERNAL, "a chain with coverage planes reached the inline-scaling kernel");
        const size_t lds = CS_T_BYTES + (size_t)a.n_scaled * CS_S_BYTES;
        if (a.use_matrix == 3) hipLaunchKernelGGL((k_fade_chain_rgba<3, true>), grid, dim3(256), lds, s, a, tx, n_tiles);
        else if (a.use_matrix == 2) hipLaunchKernelGGL((k_fade_chain_rgba<2, true>), grid, dim3(256), lds, s, a, tx, n_tiles);
        else if (a.use_matrix) hipLaunchKernelGGL((k_fade_chain_rgba<1, true>), grid, dim3(256), lds, s, a, tx, n_tiles);
        else hipLaunchKernelGGL((k_fade_chain_rgba<0, true>), grid, dim3(256), lds, s, a, tx, n_tiles);
        return;
    }
    chain_blank_planes(a);
    const uint32_t nb = chain_strip_blocks(a);   // the strip form: tx = 0 marks it for the kernel
    if (a.alpha_mask) {
        if (a.use_matrix == 3) hipLaunchKernelGGL((k_fade_chain_rgba<3, false, true>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
        else if (a.use_matrix == 2) hipLaunchKernelGGL((k_fade_chain_rgba<2, false, true>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
        else if (a.use_matrix) hipLaunchKernelGGL((k_fade_chain_rgba<1, false, true>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
        else hipLaunchKernelGGL((k_fade_chain_rgba<0, false, true>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
        return;
    }
    if (a.use_matrix == 4) hipLaunchKernelGGL((k_fade_chain_rgba<4, false>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
    else if (a.use_matrix == 3) hipLaunchKernelGGL((k_fade_chain_rgba<3, false>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
    else if (a.use_matrix == 2) hipLaunchKernelGGL((k_fade_chain_rgba<2, false>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
    else if (a.use_matrix) hipLaunchKernelGGL((k_fade_chain_rgba<1, false>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
    else hipLaunchKernelGGL((k_fade_chain_rgba<0, false>), dim3(nb), dim3(256), 0, s, a, 0u, nb);
}

// Blank fill (codec/src/ffmpeg/frame.rs:76-138): Y = 0, U = V = 0x80 over the whole allocation of each plane.
__global__ __launch_bounds__(256) void k_blank(uint8_t* y, size_t y_bytes, uint8_t* u, size_t u_bytes, uint8_t* v, size_t v_bytes, uint32_t cfill) {
    const size_t yq = y_bytes / 16, uq = u_bytes / 16, vq = v_bytes / 16;
    const uint4 zy = make_uint4(0, 0, 0, 0), zc = make_uint4(cfill, cfill, cfill, cfill);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < yq + uq + vq; i += (size_t)gridDim.x * 256) {
        if (i < yq) reinterpret_cast<uint4*>(y)[i] = zy;
        else if (i < yq + uq) reinterpret_cast<uint4*>(u)[i - yq] = zc;
        else reinterpret_cast<uint4*>(v)[i - yq - uq] = zc;
    }
}
void launch_blank(uint8_t* y, size_t yb, uint8_t* u, size_t ub, uint8_t* v, size_t vb, hipStream_t s, uint32_t cfill) {
    flush_scales(s);
    const size_t q = (yb + ub + vb) / 16;
    if (!q) return;
    hipLaunchKernelGGL(k_blank, dim3(grid_x(q, 256, 2048)), dim3(256), 0, s, y, yb, u, ub, v, vb, cfill);
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
// Window origin in integers (same value as sc_first_tap; checked against the tap table for every output on the host,
// scale_tile_origins_match_m): q = floor(n / d), n = (2o+1) src + dst < 2^30, d = 2 dst, from M = floor(2^32 / d): mulhi(n, M) is q or
// q - 1 (n M / 2^32 > n / d - 1/4), the remainder decides.  Every operand is wave-uniform where the tile kernel calls it: SALU only.
__host__ __device__ __forceinline__ int sc_first_tap_m(uint32_t o, uint32_t src, uint32_t dst, uint32_t M) {
    const uint32_t n = (2u * o + 1u) * src + dst, d = 2u * dst;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t q = __umulhi(n, M);
#else
    uint32_t q = (uint32_t)(((uint64_t)n * M) >> 32);
#endif
    q += (n - q * d >= d) ? 1u : 0u;
    return (int)q - 2;
}
uint32_t scale_origin_magic(uint32_t dst) { return dst ? (uint32_t)((1ull << 32) / (2ull * dst)) : 0u; }
bool scale_tile_origins_match_m(uint32_t src, uint32_t dst, const int32_t* first) {
    if (!dst || src > 16384u || dst > 16384u) return false;
    const uint32_t M = scale_origin_magic(dst);
    for (uint32_t o = 0; o < dst; ++o)
        if (sc_first_tap_m(o, src, dst, M) != first[o]) return false;
    return true;
}

// H pass of one tile column with compile-time LDS strides (SS bytes per window row, SC_TW pairs per T2 row) over a FIXED number of rows
// per thread: T2[p][i] = (t'[p], t'[p+1]) for the HR rows from p0 on.  The rows go in groups of eight: nine 8-byte LDS reads are
// issued, then eight H values are computed and written -- the LDS round trip is paid once per group, not once per row (a loop that reads
// and filters row by row waits lgkmcnt(0) in every iteration).  Rows beyond the tile's window are filtered too (stale LDS bytes in,
// pairs out that the V pass never reads): no bounds, no exec masks, every address an immediate.
template <int SS, int HR>
__device__ __forceinline__ void sc_hcol_fixed(const uint8_t* S, int hf /* first tap - cxa */, const uint2 hpk, uint32_t* T2, int i, int p0) {
    static_assert(HR % 8 == 0, "rows per thread go in groups of eight");
    const int hb = hf & ~3; const uint32_t sh8 = (uint32_t)(hf & 3);
    const uint8_t* Srow = S + p0 * SS + hb;
    uint32_t* out = T2 + p0 * SC_TW + i;
    uint32_t prev;
    { const uint32_t* sp = reinterpret_cast<const uint32_t*>(Srow);
      const uint32_t w = __builtin_amdgcn_alignbyte(sp[1], sp[0], sh8);
      prev = sc_h_value(w, hpk); }
#pragma unroll
    for (int g = 0; g < HR / 8; ++g) {
        uint32_t lo[8], hi[8];
        // one base address per group of eight rows, hidden from the optimiser: the eight 8-byte reads then carry their row offsets as
        // instruction immediates (ds_read2_b32 reaches 1020 bytes) instead of one address computation each from a base rows away
        uint32_t base = (uint32_t)(uintptr_t)(Srow + (8 * g + 1) * SS);
        asm volatile("" : "+v"(base));
        const __attribute__((address_space(3))) uint8_t* Sg = (const __attribute__((address_space(3))) uint8_t*)(uintptr_t)base;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __attribute__((address_space(3))) uint32_t* sp = reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(Sg + j * SS);
            lo[j] = sp[0]; hi[j] = sp[1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = __builtin_amdgcn_alignbyte(hi[j], lo[j], sh8);                      // s' of taps hf .. hf + 3
            const uint32_t cur = sc_h_value(w, hpk);                                               // t' of this row
            out[(8 * g + j) * SC_TW] = __builtin_amdgcn_perm(cur, prev, 0x05040100u);             // (t'[p] & 0xffff) | (t'[p+1] << 16)
            prev = cur;
        }
    }
}

// The tile body.  CPR = 16-byte chunks staged per window row (8: windows up to 128 bytes wide -- every upscale by >= 1.27; 16: up to
// 256), Q = staging passes of 256 / CPR rows each.  Interior tiles (the whole window inside the source row, 16-byte aligned planes)
// stage with ONE predicated dwordx4 load per lane and pass -- nothing is fetched that the tile does not filter; tiles at the picture's
// left / right edge, unaligned or interleaved (nv12 chroma) sources take sc_stage's general path into the same window.
// TH = output rows per tile (a multiple of 8): 32, or 40 where 40 output rows still need no more than 32 window rows (upscales by >= ~1.45):
// the H pass filters a fixed 2 HR rows whatever the window holds, so more output rows per tile mean fewer H values per output.
template <int CPR, int Q, int HR, int TH, class PlaneRef>
__device__ __forceinline__ void scale_tile_body(PlaneRef& p, const uint32_t tile, const uint32_t tiles_x, const int s_rows) {
    constexpr int SS = CPR * 16, RPP = 256 / CPR;
    constexpr int ROWS = 2 * HR;               // T2 rows in LDS (the launcher's s_rows <= ROWS); the window has one row more
    const int ty = (int)(tile / tiles_x), tx = (int)(tile - (uint32_t)ty * tiles_x);
    const int ox0 = tx * SC_TW, oy0 = ty * TH;
    extern __shared__ __attribute__((aligned(16))) uint8_t sc_smem[];
    uint32_t* const T2 = reinterpret_cast<uint32_t*>(sc_smem);                 // [ROWS][SC_TW] pairs (p, p + 1) of H-filtered rows
    uint8_t* const S = sc_smem + (size_t)ROWS * SC_TW * 4;                     // [ROWS + 1][SS] source window, signed bytes
    const int tid = threadIdx.x;
    const int dw = (int)p.dw, dh = (int)p.dh, sw = (int)p.sw, sh = (int)p.sh;
    const int oxe = min(ox0 + SC_TW, dw), oye = min(oy0 + TH, dh);
    // source window of the tile: wave-uniform integer arithmetic
    const uint32_t dhf = p.dh_full ? p.dh_full : p.dh;
    const int cx0 = sc_first_tap_m((uint32_t)ox0, p.sw, p.dw, p.mh), cxl = sc_first_tap_m((uint32_t)(oxe - 1), p.sw, p.dw, p.mh);
    const int ry0 = sc_first_tap_m((uint32_t)oy0 + p.oy_base, p.sh, dhf, p.mv), ryl = sc_first_tap_m((uint32_t)(oye - 1) + p.oy_base, p.sh, dhf, p.mv);
    ScWin w;
    w.cxa = cx0 & ~15; w.nc4 = min((cxl + 4 - w.cxa + 3) >> 2, SS / 4); w.ry0 = ry0; w.nr = min(ryl + 4 - ry0, s_rows);
    // every table entry this lane will need, one 16-byte load per axis entry, in one burst
    const int oxi = tid & (SC_TW - 1);
    const uint4 hx = p.hx[min(ox0 + oxi, dw - 1)];
    const int cg = tid & 31, oyr = tid >> 5;                  // V pass: pixel group (4 columns) and first row; rows oyr + 8k
    uint4 vx[TH / 8];
#pragma unroll
    for (int k = 0; k < TH / 8; ++k) vx[k] = p.vx[min(oy0 + oyr + 8 * k, dh - 1)];
    // ---- stage ----
    const int row_lo = max(0, (int)p.h_row0), row_hi = min(sh - 1, (int)(p.h_row0 + p.h_rows) - 1);
    const int n_chunks = (w.nc4 + 3) >> 2;
    const bool fast = p.sxs == 0u && ((reinterpret_cast<uintptr_t>(p.src) | p.src_stride) & 15u) == 0 && w.cxa >= 0 && w.cxa + 16 * n_chunks <= sw;
    if (fast) {
        const int c = tid % CPR, r0 = tid / CPR;
        const uint8_t* const col = p.src + w.cxa + 16 * c;
        uint4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = r0 + q * RPP;
            const int row = min(max(ry0 + r, row_lo), row_hi);
            v[q] = make_uint4(0u, 0u, 0u, 0u);
            if (c < n_chunks && r < w.nr) v[q] = *reinterpret_cast<const uint4*>(col + (size_t)((uint32_t)row * p.src_stride));
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = r0 + q * RPP;
            if (c < n_chunks && r < w.nr)
                *reinterpret_cast<uint4*>(S + r * SS + 16 * c) = make_uint4(v[q].x ^ 0x80808080u, v[q].y ^ 0x80808080u, v[q].z ^ 0x80808080u, v[q].w ^ 0x80808080u);
        }
    } else {
        sc_stage<3>(p.src, p.src_stride, sw, sh, w, S, SS, tid, p.sxs, p.sxo, (int)p.h_row0, (int)(p.h_row0 + p.h_rows) - 1);   // launcher: s_rows <= 48
    }
    __syncthreads();
    // ---- H pass: the two threads of a column take HR row pairs each (columns beyond the picture repeat the last one) ----
    sc_hcol_fixed<SS, HR>(S, (int)hx.z - w.cxa, make_uint2(hx.x, hx.y), T2, oxi, (tid >> 7) * HR);
    __syncthreads();
    // ---- V pass ----
    const int oxg = cg * 4;
    if (ox0 + oxg < dw) {
        const uint32_t off0 = (uint32_t)(oy0 + oyr) * p.dst_stride + (uint32_t)(ox0 + oxg);   // 32-bit offsets from the (scalar) plane base
        const bool whole = ox0 + oxg + 4 <= dw && ((reinterpret_cast<uintptr_t>(p.dst) | p.dst_stride) & 3u) == 0;   // ox0 + oxg is a multiple of 4
        // every row's quad is computed (rows below the picture use the last row's entry: their reads stay inside T2), only the stores are
        // conditional: the TH / 4 LDS reads go out in one burst instead of one round trip per row
        uint32_t quad[TH / 8];
#pragma unroll
        for (int k = 0; k < TH / 8; ++k) quad[k] = sc_vquad(T2, SC_TW, (int)vx[k].z - ry0, oxg, make_uint2(vx[k].x, vx[k].y));
#pragma unroll
        for (int k = 0; k < TH / 8; ++k) {
            if (oy0 + oyr + 8 * k < dh) {
                uint8_t* o = p.dst + (size_t)(off0 + (uint32_t)(8 * k) * p.dst_stride);
                if (whole) *reinterpret_cast<uint32_t*>(o) = quad[k];
                else for (int j = 0; j < 4 && ox0 + oxg + j < dw; ++j) o[j] = (uint8_t)(quad[k] >> (8 * j));
            }
        }
    }
}
// variant (launcher-chosen, wave-uniform): 0 = windows <= 128 bytes x 32 rows, 1 = 128 x 48, 2 = 256 x 48 (tiles of 128 x 32 outputs);
// 3 = 128 bytes x 32 rows under tiles of 128 x 40 outputs (the 720p -> 1080p class)
template <class PlaneRef>
__device__ __forceinline__ void scale_tile_variant(const uint32_t variant, PlaneRef& p, const uint32_t tile, const uint32_t tiles_x, const int s_rows) {
    if (variant == 3u) scale_tile_body<8, 1, 16, 40>(p, tile, tiles_x, s_rows);
    else if (variant == 0u) scale_tile_body<8, 1, 16, 32>(p, tile, tiles_x, s_rows);
    else if (variant == 1u) scale_tile_body<8, 2, 24, 32>(p, tile, tiles_x, s_rows);
    else scale_tile_body<16, 3, 24, 32>(p, tile, tiles_x, s_rows);
}
__device__ __forceinline__ void scale_tile(const ScaleBatchArgs& a, const uint32_t block) {
    int plane = 0;
#pragma unroll
    for (int k = 1; k < MX_SCALE_BATCH_PLANES; ++k) if (k < (int)a.n && block >= a.tile_start[k]) plane = k;   // scalar search
    scale_tile_variant(a.variant, a.p[plane], block - a.tile_start[plane], a.tiles_x[plane], (int)a.s_rows);
}
// LDS of a tile body: [2 HR][128] row pairs + [2 HR + 1][SS] window
static size_t scale_tile_lds(uint32_t variant) {
    const size_t hr = (variant == 0u || variant == 3u) ? 16 : 24, ss = variant == 2u ? 256 : 128;
    return 2 * hr * SC_TW * 4 + (2 * hr + 1) * ss;
}

__global__ __launch_bounds__(256) void k_scale_bicubic_tiled(ScaleBatchArgs a, uint32_t n_tiles) { scale_tile(a, xcd_run(blockIdx.x, n_tiles)); }

// ONE launch for the video work of several ticks of a batched run (Graph defers the RGBA sink inside a run): the chain tiles of ticks
// k .. k + K - 1 and the scaler tiles of the layers ticks k + K .. k + 2K - 1 will composite -- nothing in the launch reads what
// something else in it writes.  What this buys is not the launch floor alone: a frame's chain is ~2 000 waves that all fit on the chip
// at once, so alone they all request their bytes together, then all compute together, then all store; with several frames' chains and
// scaler tiles in one grid the chip holds waves in every phase at once and the memory system and the VALUs work side by side
// (measured: 14.3 us per composited 1080p frame at two frames per launch, 10.0 at sixteen).
// Grid: blockIdx.y = which chain / scale job (chains first: they wait for bytes while scaler tiles, dispatched behind them, compute),
// blockIdx.x = tile of it (rows shorter than the grid's width end at once); the width is a multiple of 8, so blockIdx.x % 8 is the
// XCD in every row and xcd_run hands each XCD a contiguous run of a row's tiles.  The descriptor lives in device memory (it is far
// beyond the 4 KB of kernel arguments) and is read through the constant address space: scalar loads.
typedef const __attribute__((address_space(4))) VideoBatchDesc* VbDescPtr;
__device__ __forceinline__ uint32_t pin_s32(uint32_t v) { asm volatile("" : "+s"(v)); return v; }
template <class T> __device__ __forceinline__ T* pin_sptr(T* v) { uint64_t u = (uint64_t)(uintptr_t)v; asm volatile("" : "+s"(u)); return (T*)(uintptr_t)u; }
template <int MM, bool AL = false>
__global__ __launch_bounds__(256) void k_video_batch(const VideoBatchDesc* desc, const VbRows rows) {
    typedef const __attribute__((address_space(4))) uint8_t* BytePtr;
    const uint32_t x = blockIdx.x;
    VbRow r = rows.r[blockIdx.y];                        // kernel arguments: the whole record requested at once, waited for once
    asm volatile("" : "+s"(r.is_job), "+s"(r.n_tiles), "+s"(r.off), "+s"(r.ts1), "+s"(r.ts2), "+s"(r.tx0), "+s"(r.tx1), "+s"(r.tx2), "+s"(r.variant), "+s"(r.s_rows), "+s"(r.prio));
    const uint32_t n = r.n_tiles;
    if (x >= n) return;
    // wave priority per tile kind (launcher; MX_VIDEO_PRIO): bits 0..1 the chain tiles', bits 2..3 the scaler tiles'
    { const uint32_t pr = r.is_job ? (r.prio >> 2) & 3u : r.prio & 3u;
      if (pr == 1u) __builtin_amdgcn_s_setprio(1); else if (pr == 2u) __builtin_amdgcn_s_setprio(2); else if (pr == 3u) __builtin_amdgcn_s_setprio(3); }
    if (!r.is_job) {
        typedef const __attribute__((address_space(4))) ChainRgbaArgs* ChainPtr;
        chain_rgba_tile<MM, false, AL>(*(ChainPtr)((BytePtr)desc + r.off), (int)xcd_run(x, n), 0);
        return;
    }
    const uint32_t t = xcd_run(x, n);
    const int plane = t >= r.ts2 ? 2 : (t >= r.ts1 ? 1 : 0);
    const uint32_t ts = plane == 2 ? r.ts2 : (plane == 1 ? r.ts1 : 0u), tiles_x = plane == 2 ? r.tx2 : (plane == 1 ? r.tx1 : r.tx0);
    typedef const __attribute__((address_space(4))) ScaleJob* JobPtr;
    const auto& src = ((JobPtr)((BytePtr)desc + r.off))->p[plane];
    // the plane's record in one go: every field the tile body reads is requested here, together (read field by field where it is used, the loads
    // sat behind one another's waits), and waited for once
    ScalePlane pl;
    pl.src = src.src; pl.dst = src.dst; pl.src_stride = src.src_stride; pl.dst_stride = src.dst_stride;
    pl.sw = src.sw; pl.sh = src.sh; pl.dw = src.dw; pl.dh = src.dh;
    pl.hfirst = nullptr; pl.hcoef = nullptr; pl.vfirst = nullptr; pl.vcoef = nullptr; pl.hn = 0; pl.vn = 0; pl.tmp = nullptr; pl.hpk = nullptr; pl.vpk = nullptr;
    pl.h_row0 = src.h_row0; pl.h_rows = src.h_rows; pl.sxs = src.sxs; pl.sxo = src.sxo; pl.oy_base = src.oy_base; pl.dh_full = src.dh_full;
    pl.hx = src.hx; pl.vx = src.vx; pl.mh = src.mh; pl.mv = src.mv;
    {   // ONE statement that needs them all: the loads above are issued together and waited for once
        uint64_t p0 = (uint64_t)(uintptr_t)pl.src, p1 = (uint64_t)(uintptr_t)pl.dst, p2 = (uint64_t)(uintptr_t)pl.hx, p3 = (uint64_t)(uintptr_t)pl.vx;
        asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2), "+s"(p3), "+s"(pl.src_stride), "+s"(pl.dst_stride), "+s"(pl.sw), "+s"(pl.sh), "+s"(pl.dw), "+s"(pl.dh),
                     "+s"(pl.h_row0), "+s"(pl.h_rows), "+s"(pl.sxs), "+s"(pl.sxo), "+s"(pl.oy_base), "+s"(pl.dh_full), "+s"(pl.mh), "+s"(pl.mv));
        pl.src = (const uint8_t*)(uintptr_t)p0; pl.dst = (uint8_t*)(uintptr_t)p1; pl.hx = (const uint4*)(uintptr_t)p2; pl.vx = (const uint4*)(uintptr_t)p3;
    }
    scale_tile_variant(r.variant, pl, t - ts, tiles_x, (int)r.s_rows);
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
        const int st = 1 + (int)p.sxs; const uint8_t* rs = row + p.sxo;   // nv12 chroma: interleaved samples
        t[k] = (hc.x * (int)rs[st * x0] + hc.y * (int)rs[st * x1] + hc.z * (int)rs[st * x2] + hc.w * (int)rs[st * x3] + 64) >> 7;
    }
    const int v = (vc.x * t[0] + vc.y * t[1] + vc.z * t[2] + vc.w * t[3] + (1 << 20)) >> 21;
    p.dst[(size_t)y * p.dst_stride + x] = (uint8_t)min(max(v, 0), 255);
}

// All planes of all queued scale jobs of a stream go out as ONE launch (grid.z = plane): these
// kernels are latency-bound, so N jobs cost about as much as one.
// tile plan of a batch for the tiled kernel: false = this batch needs the gather kernel
static bool plan_scale_tiles(const ScaleBatchArgs& a, ScaleBatchArgs& b, uint32_t& total, size_t& lds, uint32_t& mw, uint32_t& mh) {
    uint32_t s_rows = 0, s_stride = 0;
    mw = mh = 0; total = 0; lds = 0;
    for (uint32_t i = 0; i < a.n; ++i) {
        const ScalePlane& p = a.p[i];
        mw = p.dw > mw ? p.dw : mw; mh = p.dh > mh ? p.dh : mh;
        if (!p.dw || !p.dh) continue;
        // window of a 128 x 32 tile: taps of its first and last output + 4, columns widened to whole dwords
        const uint32_t dhf = p.dh_full ? p.dh_full : p.dh;   // a row band scales with the ratio of the full plane
        const uint32_t rows = (uint32_t)(((uint64_t)SC_TH * p.sh + dhf - 1) / dhf) + 6;
        const uint32_t cols = (uint32_t)(((uint64_t)SC_TW * p.sw + p.dw - 1) / p.dw) + 6 + 15 + 4;   // 16-byte window start + the H pass's second dword
        s_rows = rows > s_rows ? rows : s_rows; s_stride = cols > s_stride ? cols : s_stride;
    }
    if (!a.n || !mw || !mh) return false;
    s_stride = (s_stride + 15u) & ~15u;               // rows * stride stays a multiple of 16: T is read as int4
    lds = (size_t)s_rows * s_stride + (size_t)s_rows * SC_TW * sizeof(int);   // both pair layouts of t' = one int per (row, column)
    static const int force_simple = env_int("MX_SCALE_SIMPLE", 0);
    bool lean = true;                                 // packed taps exist for every plane (ScaleTables::lean)
    for (uint32_t i = 0; i < a.n; ++i) if (a.p[i].dw && a.p[i].dh && (!a.p[i].hpk || !a.p[i].vpk || !a.p[i].hx || !a.p[i].vx)) lean = false;
    if (!(s_rows <= 48 && s_stride <= 256 && !force_simple && lean)) return false;   // the staging slots of sc_stage<3>
    b = a;
    // staging shape of the tile body (scale_tile2): windows up to 128 bytes wide stage 8 chunks per row, 32 rows per pass
    b.variant = s_stride <= 128 ? (s_rows <= 32 ? 0u : 1u) : 2u;
    s_stride = s_stride <= 128 ? 128u : 256u;
    // tiles of 40 output rows where their windows (exactly, from the tap spec) still fit the 32 rows one staging pass holds
    static const int no_th40 = env_int("MX_SCALE_NO_TH40", 0);
    uint32_t th = SC_TH;
    if (b.variant == 0u && !no_th40) {
        bool fits = true;
        for (uint32_t i = 0; i < a.n && fits; ++i) {
            const ScalePlane& p = a.p[i];
            if (!p.dw || !p.dh) continue;
            const uint32_t dhf = p.dh_full ? p.dh_full : p.dh;
            for (uint32_t oy0 = 0; oy0 < p.dh && fits; oy0 += 40u) {
                const uint32_t oye = std::min(oy0 + 40u, p.dh);
                const int ry0 = sc_first_tap_m(oy0 + p.oy_base, p.sh, dhf, p.mv), ryl = sc_first_tap_m(oye - 1u + p.oy_base, p.sh, dhf, p.mv);
                if (ryl + 4 - ry0 > 32) fits = false;
            }
        }
        if (fits) { b.variant = 3u; th = 40u; s_rows = 32u; }
    }
    lds = scale_tile_lds(b.variant);
    for (uint32_t i = 0; i < a.n; ++i) {
        b.tile_start[i] = total;
        b.tiles_x[i] = (a.p[i].dw + SC_TW - 1) / SC_TW;
        total += b.tiles_x[i] * ((a.p[i].dh + th - 1) / th);
    }
    for (uint32_t i = a.n; i <= MX_SCALE_BATCH_PLANES; ++i) { b.tile_start[i] = total; if (i < MX_SCALE_BATCH_PLANES) b.tiles_x[i] = 1; }
    b.s_rows = s_rows; b.s_stride = s_stride;
    return true;
}
// All planes of all queued scale jobs of a stream go out as ONE launch: these
// kernels are latency-bound, so N jobs cost about as much as one.
void launch_scale_batch(const ScaleBatchArgs& a, hipStream_t s) {
    ScaleBatchArgs b; uint32_t total, mw, mh; size_t lds;
    if (plan_scale_tiles(a, b, total, lds, mw, mh)) {
        if (total) hipLaunchKernelGGL(k_scale_bicubic_tiled, dim3(total), dim3(256), lds, s, b, total);
    } else if (a.n && mw && mh) {
        hipLaunchKernelGGL(k_scale_bicubic_batch, dim3((mw + 63) / 64, (mh + 3) / 4, a.n), dim3(256), 0, s, a);
    }
}
// ---- several ticks' scale jobs and RGBA chains as one launch (k_video_batch) ----
// tile plan of ONE job (the three planes of a scaled frame) for the tiled kernel; false = it needs the gather kernel
static bool plan_scale_job(const ScaleArgs& a, ScaleJob& j) {
    ScaleBatchArgs in{}, out{};
    in.n = 3;
    for (int i = 0; i < 3; ++i) in.p[i] = a.p[i];
    uint32_t total = 0, mw = 0, mh = 0; size_t lds = 0;
    if (!plan_scale_tiles(in, out, total, lds, mw, mh) || !total) return false;
    for (int i = 0; i < 3; ++i) { j.p[i] = out.p[i]; j.tile_start[i] = out.tile_start[i]; j.tiles_x[i] = out.tiles_x[i]; }
    j.tile_start[3] = total; j.variant = out.variant; j.s_rows = out.s_rows;
    return true;
}
// Descriptor slots per (device, stream): page-locked staging + its device copy + the bytes it holds.  A stream of pictures cycles through rings of
// frames (a decoder's pool, a Scaler's 2K output frames, the sink's K RGBA buffers), so the descriptor of a launch is, byte for byte, one that was
// uploaded a few launches ago: a launch whose descriptor equals a slot's content uses that slot as it is -- no copy, no event, nothing between two
// launches but the launch itself (the upload + cross-stream wait cost ~11 us per launch: 0.7 us per frame at sixteen frames per launch).  A miss
// takes the least recently used slot: it is rewritten only after the last launch that read it has finished.
namespace {
struct DescSlot { uint8_t* host = nullptr; uint8_t* dev = nullptr; size_t bytes = 0; uint64_t hash = 0, used_at = 0; hipEvent_t done = nullptr, copied = nullptr; bool used = false; };
struct DescRing { DescSlot slot[16]; uint64_t clock = 0; hipStream_t copy = nullptr; std::vector<uint8_t> build; std::mutex mu; };   // `copy`: the uploads' own stream -- they run beside the previous launch
std::mutex g_desc_mu;
std::map<std::pair<int, hipStream_t>, DescRing> g_desc;
constexpr size_t VB_HEADER = offsetof(VideoBatchDesc, c);
constexpr size_t VB_BYTES = sizeof(VideoBatchDesc);
// (a filter in front of the memcmp, eight bytes per step: a descriptor is ~30 KB and a launch must not cost the host tens of microseconds)
uint64_t desc_hash(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 1099511628211ull; h ^= h >> 29; }
    for (; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
}  // namespace
// A stream is about to be destroyed (Graph::~Graph, after it synchronised): its descriptor slots -- page-locked staging, device copies, events, the
// upload stream -- go with it.  (A later stream may get the same handle value; it starts with an empty ring.)
void video_stream_retired(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_desc_mu);
    for (auto it = g_desc.begin(); it != g_desc.end();) {
        if (it->first.second != s) { ++it; continue; }
        DescRing& ring = it->second;
        for (DescSlot& c : ring.slot) {
            if (c.host) (void)hipHostFree(c.host);
            if (c.dev) (void)hipFree(c.dev);
            if (c.done) (void)hipEventDestroy(c.done);
            if (c.copied) (void)hipEventDestroy(c.copied);
        }
        if (ring.copy) { (void)hipStreamSynchronize(ring.copy); (void)hipStreamDestroy(ring.copy); }
        it = g_desc.erase(it);
    }
}
static void launch_separately(const ScaleArgs* jobs, int n_jobs, const ChainRgbaArgs* chains, int n_chains, hipStream_t s) {
    for (int i = 0; i < n_jobs; i += 4) {
        ScaleBatchArgs b{};
        for (int k = i; k < n_jobs && k < i + 4; ++k) for (int pl = 0; pl < 3; ++pl) b.p[b.n++] = jobs[k].p[pl];
        launch_scale_batch(b, s);
    }
    for (int k = 0; k < n_chains; ++k) launch_fade_chain_rgba(chains[k], s);
}
void launch_video_batch(const ScaleArgs* jobs, int n_jobs, const ChainRgbaArgs* chains, int n_chains, hipStream_t s) {
    if (n_jobs <= 0 && n_chains <= 0) return;
    static const int no_fuse = env_int("MX_VIDEO_NO_LAUNCH_FUSION", 0);
    bool one = !no_fuse && n_jobs <= MX_VB_MAX_JOBS && n_chains <= MX_VB_MAX_CHAINS;
    int mm = -1;
    bool any_alpha = false;   // a launch whose chains include one with coverage planes runs the kernel that knows them (the others take its plain steps: the mask is per chain)
    for (int k = 0; k < n_chains && one; ++k) {
        ChainRgbaArgs c = chains[k];
        if (c.n_scaled || !c.width || !c.height) one = false;
        const int m = chain_matrix_mode(c);
        if (mm >= 0 && m != mm) one = false;
        mm = m;
        any_alpha = any_alpha || c.alpha_mask != 0;
    }
    if (!one) { launch_separately(jobs, n_jobs, chains, n_chains, s); return; }
    // the descriptor: built in host memory, then looked up among the slots already on the device
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    DescRing* ring_p;
    { std::lock_guard<std::mutex> lk(g_desc_mu); ring_p = &g_desc[{dev, s}]; }   // map nodes stay where they are; a ring is erased only when its stream is retired
    DescRing& ring = *ring_p;
    std::lock_guard<std::mutex> lk(ring.mu);     // a ring belongs to one stream, i.e. to one engine thread: uncontended, and other streams' launches do not wait here
    if (!ring.copy) hip_check(hipStreamCreateWithFlags(&ring.copy, hipStreamNonBlocking), "hipStreamCreate(descriptor uploads)");
    if (ring.build.size() < VB_BYTES) ring.build.assign(VB_BYTES, 0);
    VideoBatchDesc* d = reinterpret_cast<VideoBatchDesc*>(ring.build.data());
    uint32_t gx = 0; size_t lds = 0;
    std::memset(d, 0, VB_HEADER);
    d->n_chains = (uint32_t)n_chains; d->n_jobs = (uint32_t)n_jobs; d->jobs_off = (uint32_t)(VB_HEADER + (size_t)n_chains * sizeof(ChainRgbaArgs)); d->_pad = 0;
    for (int k = 0; k < n_chains; ++k) {
        ChainRgbaArgs& c = d->c[k];
        std::memset(&c, 0, sizeof c);             // padding bytes take part in the comparison
        c = chains[k];
        chain_matrix_mode(c); chain_blank_planes(c);
        d->chain_tx[k] = 0;
        d->chain_tiles[k] = chain_strip_blocks(c);
        gx = std::max(gx, d->chain_tiles[k]);
    }
    ScaleJob* dj = reinterpret_cast<ScaleJob*>(ring.build.data() + d->jobs_off);
    for (int k = 0; k < n_jobs; ++k) {
        std::memset(&dj[k], 0, sizeof dj[k]);
        if (!plan_scale_job(jobs[k], dj[k])) { launch_separately(jobs, n_jobs, chains, n_chains, s); return; }
        gx = std::max(gx, dj[k].tile_start[3]); lds = std::max(lds, scale_tile_lds(dj[k].variant));
    }
    // row order (MX_VIDEO_ORDER): 0 the chains' rows first, the jobs' behind them; 1 interleaved in proportion; 2 the jobs' first.
    // Round 5: interleaved rows WITH the scaler tiles' waves one priority level above the chain tiles' -- the VALU-bound waves issue whenever they can, the memory-bound
    // ones fill the gaps: 9.54 -> 9.28 us per frame (8 layers read: 10.09 -> 9.60); interleaved without the priority it is 9.94, with the chains' waves on top 10.1,
    // with only the chain tiles' load requests on top (priority dropped before their arithmetic) 9.37 against 8.95 on that box (tools/q_vprio.sh).  A launch with coverage planes (their scale jobs double the job rows) keeps the chains first: 13.2 against 13.5.
    {
        static const int order_env = env_int("MX_VIDEO_ORDER", -1);
        const int order = order_env >= 0 ? order_env : (any_alpha ? 0 : 1);
        const int nr = n_chains + n_jobs;
        int ci = 0, ji = 0;
        for (int r = 0; r < nr; ++r) {
            bool take_chain;
            if (order == 2) take_chain = ji >= n_jobs;
            else if (order == 1) take_chain = ci < n_chains && (ji >= n_jobs || (int64_t)ci * n_jobs <= (int64_t)ji * n_chains);
            else take_chain = ci < n_chains;
            d->row_of[r] = take_chain ? (uint8_t)ci++ : (uint8_t)(128 + ji++);
        }
    }
    const size_t bytes = d->jobs_off + (size_t)n_jobs * sizeof(ScaleJob);
    const uint64_t h = desc_hash(ring.build.data(), bytes);
    DescSlot* sl = nullptr;
    for (DescSlot& c : ring.slot) if (c.used && c.bytes == bytes && c.hash == h && std::memcmp(c.host, ring.build.data(), bytes) == 0) { sl = &c; break; }
    const bool hit = sl != nullptr;
    if (!hit) {
        sl = &ring.slot[0];
        for (DescSlot& c : ring.slot) { if (!c.used) { sl = &c; break; } if (c.used_at < sl->used_at) sl = &c; }
        if (!sl->host) {
            hip_check(hipHostMalloc((void**)&sl->host, VB_BYTES, hipHostMallocDefault), "hipHostMalloc(video batch descriptor)");
            hip_check(hipMalloc((void**)&sl->dev, VB_BYTES), "hipMalloc(video batch descriptor)");
            hip_check(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&sl->copied, hipEventDisableTiming), "hipEventCreate");
        }
        if (sl->used) hip_check(hipEventSynchronize(sl->done), "hipEventSynchronize(video batch descriptor)");   // the last launch that read it
        std::memcpy(sl->host, ring.build.data(), bytes);
        sl->bytes = bytes; sl->hash = h;
        // the upload goes on its own stream (the host runs ahead of the device: it executes while the previous launch still runs); the
        // launch stream only waits for its event
        hip_check(hipMemcpyAsync(sl->dev, sl->host, bytes, hipMemcpyHostToDevice, ring.copy), "hipMemcpyAsync(video batch descriptor)");
        hip_check(hipEventRecord(sl->copied, ring.copy), "hipEventRecord");
        hip_check(hipStreamWaitEvent(s, sl->copied, 0), "hipStreamWaitEvent");
    }
    sl->used_at = ++ring.clock;
    const dim3 grid((gx + 7u) & ~7u, (uint32_t)(n_chains + n_jobs));
    const VideoBatchDesc* dd = reinterpret_cast<const VideoBatchDesc*>(sl->dev);
    VbRows rows;
    std::memset(&rows, 0, sizeof rows);
    static const int vprio_env = env_int("MX_VIDEO_PRIO", -1);
    const int vprio = vprio_env >= 0 ? vprio_env : (any_alpha ? 0 : 4);
    for (int y = 0; y < n_chains + n_jobs; ++y) {
        VbRow& r = rows.r[y];
        r.prio = (uint32_t)vprio;
        const uint32_t ro = d->row_of[y];
        if (ro < 128u) { r.is_job = 0u; r.n_tiles = d->chain_tiles[ro]; r.off = (uint32_t)(VB_HEADER + (size_t)ro * sizeof(ChainRgbaArgs)); }
        else {
            const ScaleJob& jb = dj[ro - 128u];
            r.is_job = 1u; r.n_tiles = jb.tile_start[3]; r.off = d->jobs_off + (uint32_t)((size_t)(ro - 128u) * sizeof(ScaleJob));
            r.ts1 = jb.tile_start[1]; r.ts2 = jb.tile_start[2]; r.tx0 = jb.tiles_x[0]; r.tx1 = jb.tiles_x[1]; r.tx2 = jb.tiles_x[2];
            r.variant = jb.variant; r.s_rows = jb.s_rows;
        }
    }
    if (any_alpha) {
        if (mm == 3) hipLaunchKernelGGL((k_video_batch<3, true>), grid, dim3(256), lds, s, dd, rows);
        else if (mm == 2) hipLaunchKernelGGL((k_video_batch<2, true>), grid, dim3(256), lds, s, dd, rows);
        else if (mm == 1) hipLaunchKernelGGL((k_video_batch<1, true>), grid, dim3(256), lds, s, dd, rows);
        else hipLaunchKernelGGL((k_video_batch<0, true>), grid, dim3(256), lds, s, dd, rows);
    } else
    if (mm == 4) hipLaunchKernelGGL(k_video_batch<4>, grid, dim3(256), lds, s, dd, rows);
    else if (mm == 3) hipLaunchKernelGGL(k_video_batch<3>, grid, dim3(256), lds, s, dd, rows);
    else if (mm == 2) hipLaunchKernelGGL(k_video_batch<2>, grid, dim3(256), lds, s, dd, rows);
    else if (mm == 1) hipLaunchKernelGGL(k_video_batch<1>, grid, dim3(256), lds, s, dd, rows);
    else hipLaunchKernelGGL(k_video_batch<0>, grid, dim3(256), lds, s, dd, rows);
    hip_check(hipEventRecord(sl->done, s), "hipEventRecord");
    sl->used = true;
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
    const int st = 1 + (int)p.sxs;   // nv12 chroma: interleaved samples
    for (uint32_t k = 0; k < p.hn; ++k) acc += c[k] * (int)row[p.sxo + st * min(max(f + (int)k, 0), sw1)];
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

// n frame allocations of q_per_frame 16-byte words each, copied back to back into dst (blockIdx.y = frame; src[k] null = skip)
__global__ __launch_bounds__(256) void k_gather_frames(GatherArgs a) {
    const uint4* src = a.src[blockIdx.y];
    if (!src) return;
    uint4* dst = a.dst + (size_t)blockIdx.y * a.q_per_frame;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.q_per_frame; i += gridDim.x * 256) dst[i] = src[i];
}
void launch_gather_frames(const GatherArgs& a, hipStream_t s) {
    flush_scales(s);
    if (!a.n || !a.q_per_frame) return;
    hipLaunchKernelGGL(k_gather_frames, dim3(grid_x(a.q_per_frame, 256, 64), a.n), dim3(256), 0, s, a);
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
// packed RGB -> yuv444p (BUILD-SPECIFIED, DESIGN.md "Pixel formats"): what a packed scaler input stands for.  One lane per pixel; an ingest
// format conversion, not a hot path.
__global__ __launch_bounds__(256) void k_rgb_to_yuv444(const uint8_t* __restrict__ src, uint32_t src_stride, uint32_t w, uint32_t h, uint32_t bpp, uint32_t ri, uint32_t gi, uint32_t bi,
                                                       uint8_t* __restrict__ dy, uint8_t* __restrict__ du, uint8_t* __restrict__ dv, uint32_t sy, uint32_t su, uint32_t sv,
                                                       uint8_t* __restrict__ da, uint32_t sa, uint32_t ai) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t* px = src + (size_t)y * src_stride + (size_t)x * bpp;
    const int R = px[ri], G = px[gi], B = px[bi];
    if (da) da[(size_t)y * sa + x] = px[ai];   // the A byte: the pixel's coverage (straight alpha), as it is
    dy[(size_t)y * sy + x] = (uint8_t)(((47 * R + 157 * G + 16 * B + 128) >> 8) + 16);
    du[(size_t)y * su + x] = (uint8_t)(((-26 * R - 87 * G + 112 * B + 128) >> 8) + 128);
    dv[(size_t)y * sv + x] = (uint8_t)(((112 * R - 102 * G - 10 * B + 128) >> 8) + 128);
}
void launch_rgb_to_yuv444(const uint8_t* src, uint32_t src_stride, uint32_t w, uint32_t h, uint32_t bpp, uint32_t ri, uint32_t gi, uint32_t bi, uint8_t* const dst[3], const uint32_t dst_stride[3], hipStream_t s,
                          uint8_t* alpha_dst, uint32_t alpha_stride, uint32_t ai) {
    flush_scales(s);
    if (!w || !h) return;
    if (bpp != 4) alpha_dst = nullptr;
    hipLaunchKernelGGL(k_rgb_to_yuv444, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, src, src_stride, w, h, bpp, ri, gi, bi, dst[0], dst[1], dst[2], dst_stride[0], dst_stride[1], dst_stride[2],
                       alpha_dst, alpha_stride, ai);
}
// b-bit samples (10, 12, 16) in 16-bit words -> the 8-bit frame of the same layout (BUILD-SPECIFIED, include/mixlab_gpu.h mx_pixfmt):
// min(255, (v + 2^(b-9)) >> (b - 8)) with v = (word >> shift) & (2^b - 1).  Four samples per lane, all three planes in one launch (blockIdx.z); an ingest
// format conversion like the RGB one.
__global__ __launch_bounds__(256) void k_deep_to_8(DeepArgs a) {
    const uint32_t p = blockIdx.z;
    const uint32_t x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= a.w[p] || y >= a.h[p]) return;
    const uint16_t* row = reinterpret_cast<const uint16_t*>(a.src[p] + (size_t)y * a.src_stride[p]) + a.xoff[p];
    uint8_t* out = a.dst[p] + (size_t)y * a.dst_stride[p] + x4;
    const uint32_t n = min(4u, a.w[p] - x4);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t v = ((uint32_t)row[(size_t)(x4 + k) * a.xstep[p]] >> a.shift) & ((1u << a.bits) - 1u);
        out[k] = (uint8_t)min(255u, (v + (1u << (a.bits - 9u))) >> (a.bits - 8u));
    }
}
void launch_deep_to_8(const DeepArgs& a, hipStream_t s) {
    flush_scales(s);
    uint32_t w = 0, h = 0;
    for (int p = 0; p < 3; ++p) { w = std::max(w, a.w[p]); h = std::max(h, a.h[p]); }
    if (!w || !h) return;
    hipLaunchKernelGGL(k_deep_to_8, dim3((w + 255) / 256, (h + 3) / 4, 3), dim3(256), 0, s, a);
}
// packed 4:2:2 -> yuv422p: one lane per pixel PAIR (4 source bytes -> 2 luma, 1 U, 1 V); an ingest format conversion
__global__ __launch_bounds__(256) void k_yuyv_to_422p(const uint8_t* __restrict__ src, uint32_t src_stride, uint32_t w2 /* pixel pairs per row */, uint32_t h, uint32_t y_first,
                                                      uint8_t* __restrict__ dy, uint8_t* __restrict__ du, uint8_t* __restrict__ dv, uint32_t sy, uint32_t su, uint32_t sv) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w2 || y >= h) return;
    const uint32_t q = *reinterpret_cast<const uint32_t*>(src + (size_t)y * src_stride + 4 * (size_t)x);   // rows are 64-byte aligned (alloc_planes)
    const uint32_t b0 = q & 0xffu, b1 = (q >> 8) & 0xffu, b2 = (q >> 16) & 0xffu, b3 = q >> 24;
    const uint32_t y0 = y_first ? b0 : b1, u = y_first ? b1 : b0, y1 = y_first ? b2 : b3, v = y_first ? b3 : b2;
    *reinterpret_cast<uint16_t*>(dy + (size_t)y * sy + 2 * (size_t)x) = (uint16_t)(y0 | (y1 << 8));
    du[(size_t)y * su + x] = (uint8_t)u; dv[(size_t)y * sv + x] = (uint8_t)v;
}
void launch_yuyv_to_422p(const uint8_t* src, uint32_t src_stride, uint32_t w, uint32_t h, uint32_t y_first, uint8_t* const dst[3], const uint32_t dst_stride[3], hipStream_t s) {
    flush_scales(s);
    if (!w || !h) return;
    hipLaunchKernelGGL(k_yuyv_to_422p, dim3((w / 2 + 63) / 64, (h + 3) / 4), dim3(256), 0, s, src, src_stride, w / 2, h, y_first, dst[0], dst[1], dst[2], dst_stride[0], dst_stride[1], dst_stride[2]);
}
void launch_yuv420_to_rgba(const RgbaArgs& a, hipStream_t s) {
    flush_scales(s);
    if (!a.width || !a.height) return;
    hipLaunchKernelGGL(k_yuv420_to_rgba, dim3(((a.width + 3) / 4 + 63) / 64, (a.height + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace mx
