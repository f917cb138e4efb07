// mx_video.hpp -- device-resident yuv420p frames, the DynamicScaler and the VideoMixer (internal).
//
// Host logic mirrors reference src/module/video_mixer.rs:70-297 and src/video/encode.rs:311-398;
// pixel work is in mx_k_video.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <vector>

#include "mx_common.hpp"

namespace mx {

// ---- kernel argument blocks (passed by value) ----
struct FadeArgs {
    uint8_t* out[3]; const uint8_t* a[3]; const uint8_t* b[3];   // a/b nullptr => blank constant
    uint32_t out_stride[3], a_stride[3], b_stride[3];
    uint32_t chunks[3];          // rows * chunks_per_row per plane
    uint32_t chunks_per_row[3];  // ceil(width / 32) * 2 sixteen-byte chunks (fade_line's 32-byte blocks)
    uint32_t fade;               // (fader * 255.0) as u8, video_mixer.rs:168
};
struct ScalePlane {
    const uint8_t* src; uint8_t* dst;
    uint32_t src_stride, dst_stride, sw, sh, dw, dh;
    const int32_t* hfirst; const int32_t* hcoef;   // [dw], [dw][hn]
    const int32_t* vfirst; const int32_t* vcoef;   // [dh], [dh][vn]
    uint32_t hn, vn;                               // taps per output sample: 4, or 2*ceil(2*src/dst)+2 on a downscaled axis
    int32_t* tmp;                                  // wide path only: [sh][dw] H-filtered rows
    uint32_t h_row0, h_rows;                       // wide path only: the source rows the H pass filters (a row band reads a slice of the plane)
    const uint2* hpk; const uint2* vpk;            // tiled path: packed taps (ScaleTables::lean), nullptr = not available
    uint32_t sxs, sxo;                             // source samples are (1 + sxs) bytes apart starting at byte sxo of a row (nv12 chroma: 1, 0 | 1)
    uint32_t oy_base, dh_full;                     // tiled path, row bands: index of dst's first row in rows of the scaled plane, and that plane's full height (0 = dh)
    // tiled path: one 16-byte entry per output column / row -- {packed taps (hpk / vpk), first tap index, 0} -- so a lane fetches what it needs
    // of an axis with ONE load; and floor(2^32 / (2 dst)) per axis: the window origin of a tile is integer arithmetic on wave-uniform values (SALU)
    const uint4* hx; const uint4* vx;
    uint32_t mh, mv;
};
struct ScaleArgs { ScalePlane p[3]; };
enum { MX_SCALE_BATCH_PLANES = 12 };   // up to 4 frames per launch
struct ScaleBatchArgs {
    ScalePlane p[MX_SCALE_BATCH_PLANES]; uint32_t n;
    uint32_t tile_start[MX_SCALE_BATCH_PLANES + 1];   // launcher-filled: flat block index -> (plane, tile), no empty blocks
    uint32_t tiles_x[MX_SCALE_BATCH_PLANES];
    uint32_t s_rows, s_stride;                        // launcher-filled: LDS window geometry of the tiled kernel
    uint32_t variant;                                 // launcher-filled: staging shape of the tile body (mx_k_video.hip scale_tile)
};
struct CopyArgs { const uint8_t* src[3]; uint8_t* dst[3]; uint32_t src_stride[3], dst_stride[3], rows[3], row_bytes[3]; };
struct RgbaArgs {
    const uint8_t* y; const uint8_t* u; const uint8_t* v; uint8_t* rgba;
    uint32_t y_stride, u_stride, v_stride, rgba_stride, width, height;
    int32_t use_matrix; int32_t m[12];
    float mf[12];   // launcher-filled when every row of m is small enough for exact f32 sums (use_matrix == 3): m / 4096, the constant carrying the rounding
};

void launch_crossfade(const FadeArgs& a, hipStream_t s);
void launch_scale_wide(const ScaleArgs& a, hipStream_t s);   // two passes through ScalePlane::tmp, any tap counts
bool scale_tile_origins_match(uint32_t src, uint32_t dst, const int32_t* first);
uint32_t scale_origin_magic(uint32_t dst);   // floor(2^32 / (2 dst)): ScalePlane::mh / mv
bool scale_tile_origins_match_m(uint32_t src, uint32_t dst, const int32_t* first);   // the integer form the tiled kernel uses == tap table, every output   // f64 window-origin formula of the tiled / inline scalers == tap table, every output

// A chain of cross-fades evaluated per pixel in registers:  v = src[0];  for k >= 1:
//   v = v_is_a[k-1] ? fade(v, src[k]) : fade(src[k], v)   with fade(a,b) = (a*f + b*(255-f)) / 255, f = fade[k-1]
// -- each step truncates to u8 exactly like one VideoMixer (video_mixer.rs:211-235), so a cascade of
// mixers collapses into ONE pass over the layers without changing a single bit.  src[k] planes may be
// nullptr (the blank constant, video_mixer.rs:180-188).
enum { MX_CHAIN_MAX_SRC = 8 };
struct ChainSrc { const uint8_t* p[3]; uint32_t stride[3]; };
// BUILD-SPECIFIED per-pixel alpha (DESIGN.md "Per-pixel alpha"; the reference's only alpha is the global fader, video_mixer.rs:168): a layer may carry a
// coverage plane, one byte per LUMA sample (p = nullptr: opaque).  A step then weighs its two layers per sample:
//   wa = (aA * fade) / 255;  wb = (aB * (255 - wa)) / 255;  out = (A * (255 - wb) + B * wb) / 255     (u16, truncating -- fade_line's form)
// with aA / aB = 255 where a layer carries none -- then wa = fade, wb = 255 - fade: the reference's cross-fade bit for bit.  A chroma sample uses
// the coverage of its co-sited luma sample (2x, 2y).  The running composite of a chain is opaque (a VideoMixer produces yuv420p without alpha), so
// from the second step on only the OTHER layer's plane matters: w_other = (a_other * g) / 255, g its nominal factor.
struct ChainAlpha { const uint8_t* p; uint32_t stride, _pad; };
struct ChainArgs {
    ChainSrc src[MX_CHAIN_MAX_SRC]; uint32_t n_src;
    uint32_t fade[MX_CHAIN_MAX_SRC - 1]; uint32_t v_is_a[MX_CHAIN_MAX_SRC - 1];
    uint8_t* out[3]; uint32_t out_stride[3];
    uint32_t chunks[3], chunks_per_row[3];
    ChainAlpha al[MX_CHAIN_MAX_SRC]; uint32_t alpha_mask, _pad1;   // bit k: src[k] carries a coverage plane
};
// A chain layer that is the DynamicScaler's letterboxed output of a smaller (or equal-sized) picture, resampled INSIDE the chain kernel
// (4-tap axes only): the scaled frame is never written.  Index [0] = luma, [1] = both chroma planes.
enum { MX_CHAIN_MAX_SCALED = 2 };
struct ChainScale {
    const uint8_t* src[3]; uint32_t src_stride[3];
    uint32_t sw[2], sh[2];                         // source plane size
    uint32_t dw[2], dh[2];                         // scaled size (encode.rs:354-364)
    uint32_t lx[2], ly[2];                         // letterbox offset of the scaled picture in the output plane (encode.rs:366-374)
    const int32_t* hfirst[2]; const uint2* hpk[2]; const int32_t* vfirst[2]; const uint2* vpk[2];   // packed taps, ScaleTables::lean
};
struct ChainRgbaArgs {   // the same chain feeding the build-specified YUV420P -> RGBA (+ matrix) directly: no YUV frame is written
    ChainSrc src[MX_CHAIN_MAX_SRC]; uint32_t n_src;
    uint32_t fade[MX_CHAIN_MAX_SRC - 1]; uint32_t v_is_a[MX_CHAIN_MAX_SRC - 1];
    uint32_t fa_pk[MX_CHAIN_MAX_SRC - 1], fb_pk[MX_CHAIN_MAX_SRC - 1];   // launcher-filled from fade / v_is_a: the step's two factors as u16 x 2
    uint8_t* rgba; uint32_t rgba_stride, width, height;
    uint32_t upr_magic, _pad0;   // launcher-filled: floor(2^32 / units per row pair), units of 8 pixels (the strip form of the tile walks units in row-major order)
    int32_t use_matrix; int32_t m[12];
    float mf[12];   // launcher-filled when every row of m is small enough for exact f32 sums (use_matrix == 3): m / 4096, the constant carrying the rounding
    uint32_t n_scaled; uint32_t scaled_src[MX_CHAIN_MAX_SCALED];   // chain position of each inline-scaled layer (its ChainSrc planes are nullptr)
    ChainScale sc[MX_CHAIN_MAX_SCALED];
    ChainAlpha al[MX_CHAIN_MAX_SRC]; uint32_t alpha_mask, _pad1;   // coverage planes (ChainArgs); never together with inline-scaled layers
};
// Several ticks' video work as ONE launch (mx_k_video.hip k_video_batch): up to MX_VB_MAX_CHAINS RGBA chains and MX_VB_MAX_JOBS scale jobs
// (a job = the three planes of one scaled frame).  The descriptor is uploaded to device memory per launch.
enum { MX_VB_MAX_CHAINS = 16, MX_VB_MAX_JOBS = 48 };   // jobs: two scaled layers per tick of 16 ticks, and room for the coverage-plane companions of layers that carry one
struct ScaleJob { ScalePlane p[3]; uint32_t tile_start[4]; uint32_t tiles_x[3]; uint32_t variant, s_rows; };
struct VideoBatchDesc {   // header | c[n_chains] | ScaleJob[n_jobs] at byte jobs_off: only the used part is uploaded
    uint32_t n_chains, n_jobs, jobs_off, _pad;
    uint32_t chain_tiles[MX_VB_MAX_CHAINS], chain_tx[MX_VB_MAX_CHAINS];
    uint8_t row_of[MX_VB_MAX_CHAINS + MX_VB_MAX_JOBS];   // blockIdx.y -> chain k (value k) or scale job j (value 128 + j): the order rows are dispatched in
    ChainRgbaArgs c[MX_VB_MAX_CHAINS];
    ScaleJob j[MX_VB_MAX_JOBS];   // capacity only: the jobs follow the chains actually present
};
// What a block needs to find its work, per grid row, passed in the KERNEL ARGUMENTS (one scalar load off the kernarg pointer; read out of the descriptor in
// device memory it was a chain of dependent loads -- row_of[y] (a byte: a VECTOR load and a readfirstlane), then the row's tile count, then the job's
// tile_start[3], [2], [1] one after the other, then tiles_x and the variant -- eight round trips before a scaler block requested its first pixel).
struct VbRow {
    uint32_t is_job, n_tiles;         // chain row or scale-job row; blocks of the row that have work
    uint32_t off;                     // byte offset in the descriptor of the row's ChainRgbaArgs / of its ScaleJob
    uint32_t ts1, ts2;                // jobs: first tile of planes 1 and 2 (plane 0 starts at 0)
    uint32_t tx0, tx1, tx2;           // jobs: tiles per tile row of each plane
    uint32_t variant, s_rows;         // jobs: tile body variant and window rows
    uint32_t prio;                    // s_setprio per tile kind: bits 0..1 chain rows, bits 2..3 job rows (launch_video_batch)
};
struct VbRows { VbRow r[MX_VB_MAX_CHAINS + MX_VB_MAX_JOBS]; };
void launch_video_batch(const ScaleArgs* jobs, int n_jobs, const ChainRgbaArgs* chains, int n_chains, hipStream_t s);
void video_stream_retired(hipStream_t s);   // the stream is going away: free what launch_video_batch keeps for it
// K: how many ticks' RGBA chains (and the scale jobs of the K ticks after them) share one launch inside a batched run (MX_VIDEO_BATCH, default 16, 1..16)
uint32_t video_batch_ticks();
void launch_fade_chain(const ChainArgs& a, hipStream_t s);
void launch_fade_chain_rgba(const ChainRgbaArgs& a, hipStream_t s);
void launch_blank(uint8_t* y, size_t yb, uint8_t* u, size_t ub, uint8_t* v, size_t vb, hipStream_t s, uint32_t cfill = 0x80808080u);   // cfill: the chroma planes' 32-bit fill pattern
void launch_scale_bicubic(const ScaleArgs& a, hipStream_t s);
void launch_scale_batch(const ScaleBatchArgs& a, hipStream_t s);
// deferred scaling: Scaler::scale queues its planes per stream; every reader of frame pixels flushes first
struct FrameRef;
struct ScaleTables;
void queue_scale(const ScaleArgs& a, hipStream_t s, const FrameRef& src, const FrameRef& dst, std::shared_ptr<const ScaleTables> tabs = nullptr,
                 bool companion = false);   // companion: the coverage-plane job of the frame pair just queued (other planes of the same frames: no hazard, no flush)
void flush_scales(hipStream_t s);
void launch_chains_rgba_after_queued_scales(const ChainRgbaArgs* chains, int n_chains, hipStream_t s);
void launch_copy_planes(const CopyArgs& a, hipStream_t s);
// whole frame allocations (equal size, 16-byte multiples) gathered back to back: the packed read-back of a sink's kept frames
struct GatherArgs { const uint4* src[224]; uint4* dst; uint32_t q_per_frame, n; };
void launch_gather_frames(const GatherArgs& a, hipStream_t s);
void launch_yuv420_to_rgba(const RgbaArgs& a, hipStream_t s);
// packed RGB (bpp bytes per pixel; R, G, B at byte ri, gi, bi of a pixel) -> yuv444p planes, BUILD-SPECIFIED BT.709 limited range (DESIGN.md "Pixel formats")
struct DeepArgs {   // per plane: source words xstep apart from word xoff of a row, w x h samples
    const uint8_t* src[3]; uint8_t* dst[3]; uint32_t src_stride[3], dst_stride[3], w[3], h[3], xstep[3], xoff[3]; uint32_t shift, bits;
};
void launch_deep_to_8(const DeepArgs& a, hipStream_t s);
// packed 4:2:2 (yuyv: y_first 1, uyvy: 0) -> yuv422p planes: a byte shuffle
void launch_yuyv_to_422p(const uint8_t* src, uint32_t src_stride, uint32_t w, uint32_t h, uint32_t y_first, uint8_t* const dst[3], const uint32_t dst_stride[3], hipStream_t s);
void launch_rgb_to_yuv444(const uint8_t* src, uint32_t src_stride, uint32_t w, uint32_t h, uint32_t bpp, uint32_t ri, uint32_t gi, uint32_t bi, uint8_t* const dst[3], const uint32_t dst_stride[3], hipStream_t s,
                          uint8_t* alpha_dst = nullptr, uint32_t alpha_stride = 0, uint32_t ai = 0);   // alpha_dst: the A byte (byte ai of a four-byte pixel) goes to that plane

// ---- exact rationals: MediaTime / MediaDuration (util/src/time.rs:9-75, num_rational::Ratio<i64>) ----
struct Rational {
    int64_t num = 0, den = 1;
    static Rational make(int64_t n, int64_t d);
    Rational operator+(const Rational& o) const;
    Rational operator-(const Rational& o) const;
    bool operator>=(const Rational& o) const;
    bool operator<(const Rational& o) const { return !(*this >= o); }
    bool operator>(const Rational& o) const { return o < *this; }
};

// ---- device frame: the AvFrame<Video> stand-in (codec/src/ffmpeg/frame.rs) ----
// Reference-counted like an AVFrame (frame.rs:351-361 clone = av_frame_clone): the VideoMixer keeps
// inputs past the call by retaining them, never by copying pixels.
struct DFrame;
struct FrameRef {   // intrusive handle
    DFrame* f = nullptr;
    FrameRef() = default;
    explicit FrameRef(DFrame* p, bool add_ref);
    FrameRef(const FrameRef& o);
    FrameRef(FrameRef&& o) noexcept : f(o.f) { o.f = nullptr; }
    FrameRef& operator=(FrameRef o) noexcept { DFrame* t = f; f = o.f; o.f = t; return *this; }
    ~FrameRef();
    explicit operator bool() const { return f != nullptr; }
    DFrame* operator->() const { return f; }
};
struct ScaleGeometry { uint32_t scaled_w, scaled_h, letterbox_x, letterbox_y; };
// A frame whose pixels have not been computed yet: the cross-fade chain that defines them.  Created
// only for VideoMixer program outputs whose single consumer is another video node of the same graph
// (decided at graph build), and always evaluated inside the tick that created it.
struct LazyChain {
    struct Step { FrameRef other; uint8_t fade; bool v_is_a; };
    FrameRef base;                 // empty = blank
    std::vector<Step> steps;
};

// tap tables + geometry of one (input settings -> output settings) scaler context; shared by the Scaler and the frames it defines
struct ScaleTables {
    uint32_t in_w = 0, in_h = 0, out_w = 0, out_h = 0;
    uint32_t in_cw = 1, in_ch = 1;       // the input format's chroma subsampling (the output is yuv420p)
    ScaleGeometry geo{};
    DevBuf tabs;                         // tap tables for luma and chroma
    uint32_t taps[2][2] = {{4, 4}, {4, 4}};   // [luma/chroma][h, v]
    const int32_t* tab[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // [luma/chroma][hfirst,hcoef,vfirst,vcoef]
    // 4-tap contexts whose taps fit the packed-dot-product form of the tiled / inline kernels (mx_k_video.hip):
    //   H: {ch0..3 as i8 x 4, cl0..3 as i8 x 4} with c = 256 ch + cl;  V: {(c0, c1), (c2, c3)} as i16 x 2
    bool four_tap = false;
    const uint2* hpk[2] = {nullptr, nullptr}; const uint2* vpk[2] = {nullptr, nullptr};
    const uint4* hx[2] = {nullptr, nullptr}; const uint4* vx[2] = {nullptr, nullptr};   // {packed taps, first tap, 0} per output (ScalePlane::hx / vx)
    uint32_t mh[2] = {0, 0}, mv[2] = {0, 0};
};
// A frame that is the DynamicScaler's output of `src` and has not been computed: chain kernels resample it inline; anyone else
// materialises it into the scaler's own output frame (`target`, encode.rs:382-396 -- the reference reuses that frame too).
struct LazyScale {
    FrameRef src; std::shared_ptr<ScaleTables> t; FrameRef target;   // only 4-tap contexts are deferred (ScaleTables::four_tap)
};

struct DFrame {
    std::atomic<int> rc{1};
    std::shared_ptr<LazyChain> lazy;     // non-null => data[] are nullptr until ensure_pixels()
    std::shared_ptr<LazyScale> lazy_scale;   // non-null => likewise; after ensure_pixels() data[] alias `alias`'s planes
    FrameRef alias;
    static DFrame* create_lazy_scale(uint32_t w, uint32_t h, std::shared_ptr<LazyScale> sc);
    static DFrame* create_lazy(uint32_t w, uint32_t h, std::shared_ptr<LazyChain> c);
    void ensure_pixels(hipStream_t s);   // materialise a lazy frame (one fused launch)
    uint32_t width = 0, height = 0;      // luma size (a multiple of the chroma subsampling)
    uint8_t fmt = MX_PIXFMT_YUV420P;     // mx_pixfmt
    static constexpr uint8_t kLastFmt = MX_PIXFMT_ABGR;
    // packed 8-bit RGB: bytes per pixel and the byte index of R, G, B inside a pixel (0 bytes per pixel: not an RGB format)
    struct Rgb { uint8_t bpp, r, g, b; };
    static Rgb rgb_of(uint8_t f) {
        switch (f) {
        case MX_PIXFMT_RGB24: return {3, 0, 1, 2}; case MX_PIXFMT_BGR24: return {3, 2, 1, 0};
        case MX_PIXFMT_BGRA: return {4, 2, 1, 0};  case MX_PIXFMT_RGBA: return {4, 0, 1, 2};
        case MX_PIXFMT_ARGB: return {4, 1, 2, 3};  case MX_PIXFMT_ABGR: return {4, 3, 2, 1};
        default: return {0, 0, 0, 0};
        }
    }
    bool yuyv() const { return fmt == MX_PIXFMT_YUYV422 || fmt == MX_PIXFMT_UYVY422; }   // packed 4:2:2: a scaler input only, de-interleaved into the yuv422p frame it stands for
    // samples deeper than 8 bits in 16-bit words: a scaler input only, turned into the 8-bit frame of the same layout it stands for (Scaler::planar_of)
    struct Deep { uint8_t layout /* the 8-bit format of the layout */, bits, shift /* of the value inside a word */, semi; };
    static const Deep* deep_of(uint8_t f) {
        static constexpr Deep k[] = {
            {MX_PIXFMT_YUV420P, 10, 0, 0}, {MX_PIXFMT_YUV422P, 10, 0, 0}, {MX_PIXFMT_YUV444P, 10, 0, 0}, {MX_PIXFMT_YUV420P, 10, 6, 1},   // 10 .. 13
            {MX_PIXFMT_YUV420P, 12, 0, 0}, {MX_PIXFMT_YUV422P, 12, 0, 0}, {MX_PIXFMT_YUV444P, 12, 0, 0},                                  // 14 .. 16
            {MX_PIXFMT_YUV420P, 16, 0, 0}, {MX_PIXFMT_YUV422P, 16, 0, 0}, {MX_PIXFMT_YUV444P, 16, 0, 0}, {MX_PIXFMT_YUV420P, 16, 0, 1}};  // 17 .. 20
        return (f >= MX_PIXFMT_YUV420P10 && f <= MX_PIXFMT_P016) ? &k[f - MX_PIXFMT_YUV420P10] : nullptr;
    }
    bool deep() const { return deep_of(fmt) != nullptr; }
    uint32_t bps() const { return deep() ? 2u : 1u; }                       // bytes per stored sample
    uint32_t blank_chroma() const {                                         // mid-scale chroma as the format stores it, two words
        const Deep* d = deep_of(fmt);
        if (!d) return 0x80808080u;
        const uint32_t w = (1u << (d->bits - 1)) << d->shift;
        return w | (w << 16);
    }
    bool packed() const { return rgb_of(fmt).bpp != 0 || fmt == MX_PIXFMT_GRAY8 || yuyv(); }   // ONE stored plane (3 / 4 / 1 bytes per pixel): a scaler input only, turned into the yuv444p frame it stands for
    uint32_t bpp() const { return rgb_of(fmt).bpp ? rgb_of(fmt).bpp : (yuyv() ? 2u : 1u); }
    static uint32_t fmt_cw(uint8_t f) { if (const Deep* d = deep_of(f)) f = d->layout; return (f == MX_PIXFMT_YUV444P || f == MX_PIXFMT_YUV440P || rgb_of(f).bpp != 0 || f == MX_PIXFMT_GRAY8) ? 0u : ((f == MX_PIXFMT_YUV410P || f == MX_PIXFMT_YUV411P) ? 2u : 1u); }   // log2_chroma_w, pixfmt.rs:97-100
    static uint32_t fmt_ch(uint8_t f) { if (const Deep* d = deep_of(f)) f = d->layout; return (f == MX_PIXFMT_YUV420P || f == MX_PIXFMT_NV12 || f == MX_PIXFMT_YUV440P) ? 1u : (f == MX_PIXFMT_YUV410P ? 2u : 0u); }   // log2_chroma_h, pixfmt.rs:102-105
    uint32_t cw() const { return fmt_cw(fmt); }
    uint32_t chs() const { return fmt_ch(fmt); }
    // nv12: the two chroma "planes" are the even / odd bytes of ONE stored plane (data[1]; data[2] aliases it): samples xstep bytes apart from xoff
    bool semi() const { const Deep* d = deep_of(fmt); return fmt == MX_PIXFMT_NV12 || (d && d->semi); }
    uint32_t xstep(int p) const { return (semi() && p) ? 2u : 1u; }
    uint32_t xoff(int p) const { return (semi() && p == 2) ? 1u : 0u; }
    int stored_planes() const { return packed() ? 1 : (semi() ? 2 : 3); }
    uint32_t stored_row_bytes(int p) const { return packed() ? width * bpp() : ((semi() && p == 1) ? width : pw(p)) * bps(); }
    uint8_t* data[3] = {nullptr, nullptr, nullptr};
    uint32_t stride[3] = {0, 0, 0};
    size_t plane_bytes[3] = {0, 0, 0};
    // BUILD-SPECIFIED coverage plane (MX_PIXFMT_YUVA420P; the planar frame a four-byte packed RGB input stands for): width x height bytes, 255 = opaque.
    // The public format YUVA420P is (fmt = YUV420P, with_alpha); a frame without pixels yet (lazy scale) knows with_alpha before `alpha` exists.
    bool with_alpha = false;
    uint8_t* alpha = nullptr;
    uint32_t alpha_stride = 0;
    size_t alpha_bytes = 0;
    bool carries_alpha() const { return with_alpha || rgb_of(fmt).bpp == 4; }
    DevBuf mem;
    static DFrame* create(uint32_t w, uint32_t h, hipStream_t s, uint8_t fmt = MX_PIXFMT_YUV420P, bool alpha = false);   // blank-filled (frame.rs:76-138); alpha: opaque
    static DFrame* create_unfilled(uint32_t w, uint32_t h, uint8_t fmt, bool alpha = false);   // planes allocated, contents undefined: for a caller that overwrites every byte (FrameStager)
    size_t plane_offset(int p) const { return (size_t)(data[p] - (uint8_t*)mem.p); }
    void retain() { rc.fetch_add(1, std::memory_order_relaxed); }
    void release() { if (rc.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this; }
    uint32_t pw(int p) const { return p ? width >> cw() : width; }
    uint32_t ph(int p) const { return p ? height >> chs() : height; }
};
inline FrameRef::FrameRef(DFrame* p, bool add_ref) : f(p) { if (f && add_ref) f->retain(); }
inline FrameRef::FrameRef(const FrameRef& o) : f(o.f) { if (f) f->retain(); }
inline FrameRef::~FrameRef() { if (f) f->release(); }

// flatten (a, b, fade) into a chain: extends a's (or b's) chain when that side is lazy
std::shared_ptr<LazyChain> make_chain(const FrameRef& a, const FrameRef& b, uint8_t fade, hipStream_t s);
void fill_chain_sources(const LazyChain& c, ChainSrc (&src)[MX_CHAIN_MAX_SRC], uint32_t& n_src,
                        uint32_t (&fade)[MX_CHAIN_MAX_SRC - 1], uint32_t (&v_is_a)[MX_CHAIN_MAX_SRC - 1],
                        ChainAlpha (&al)[MX_CHAIN_MAX_SRC], uint32_t& alpha_mask);
// the RGBA sink's form: layers that are unevaluated scaler outputs are handed over as ChainScale (up to MX_CHAIN_MAX_SCALED; the rest
// and every layer the inline resampler cannot take are materialised on `s` first)
void fill_chain_rgba_sources(const LazyChain& c, ChainRgbaArgs& a, hipStream_t s);

ScaleGeometry scaler_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h);   // encode.rs:354-374
void unify_picture_settings(uint32_t aw, uint32_t ah, uint32_t bw, uint32_t bh, uint32_t& w, uint32_t& h);   // video_mixer.rs:276-297
uint8_t crossfade_factor(double fader);   // video_mixer.rs:168
uint32_t scaler_tap_count(uint32_t src, uint32_t dst);
void scaler_taps(uint32_t src, uint32_t dst, std::vector<int32_t>& first, std::vector<int32_t>& coef);   // DESIGN.md "Scaler"
// Row band of DynamicScaler::scale (multi-GPU row bands): luma rows [row0, row0 + out->height) of the (full_w x full_h)
// letterboxed result from a slice holding luma rows [src_row0, src_row0 + slice->height) of a source in_full_h rows high.  Synchronous.
void scale_band(const DFrame* slice, uint32_t in_full_h, uint32_t src_row0, DFrame* out, uint32_t full_w, uint32_t full_h, uint32_t row0, hipStream_t s);

// The same band, planned once and run asynchronously every tick (a rank of a row-band sharded job scales its halo slice of every smaller
// layer per tick).  One run at a time per object (the H-filtered rows live in its buffer): runs are ordered by the stream.
class BandScaler {
public:
    BandScaler(uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows, uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows);
    void run(const DFrame* slice, DFrame* out, hipStream_t s);
    uint32_t full_w() const { return full_w_; }
    uint32_t band_rows() const { return band_rows_; }
private:
    uint32_t in_w_, src_row0_, slice_rows_, full_w_, band_rows_;
    ScaleGeometry geo_{};
    DevBuf tabs_, tmp_;
    ScaleArgs plan_{};
    uint32_t dst_row_[3] = {0, 0, 0};
    bool any_ = false, tiled_ = false, needs_blank_ = true;
};

// DynamicScaler (src/video/encode.rs:311-398)
class Scaler {
public:
    Scaler(uint32_t out_w, uint32_t out_h, hipStream_t s) : out_w_(out_w), out_h_(out_h), stream_(s) {}
    ~Scaler() { flush_scales(stream_); }   // queued jobs reference this scaler's tap tables
    uint32_t out_w() const { return out_w_; }
    uint32_t out_h() const { return out_h_; }
    // returns the frame itself when its settings equal the output's (encode.rs:342-345), else the
    // scaler's own letterboxed output frame (encode.rs:386-396)
    // may_defer: the caller's consumers are chain kernels, which resample a 4-tap scale inline -- the result may be a frame without pixels
    FrameRef scale(const FrameRef& in, bool may_defer = false);
    // the same picture in a frame nobody else holds, so that a sink can keep every tick's picture of a batch (Monitor)
    FrameRef scale_keep(const FrameRef& in);
    // a topology edit moved the owning VideoMixer to another graph: queued work leaves on the old stream first
    void rebind(hipStream_t s) { flush_scales(stream_); stream_ = s; }
private:
    void retarget(uint32_t in_w, uint32_t in_h, uint8_t in_fmt, bool in_alpha);
    uint32_t out_w_, out_h_;
    hipStream_t stream_;
    uint32_t in_w_ = 0, in_h_ = 0;   // settings the cached context was built for (encode.rs:347-352)
    uint8_t in_fmt_ = MX_PIXFMT_YUV420P;
    bool in_alpha_ = false;          // the input carries a coverage plane: so do the output frames
    FrameRef frame_;                 // cached blank output frame (encode.rs:382) -- the one the latest scale() wrote
    std::vector<FrameRef> keep_pool_; // scale_keep's outputs: blank frames of this context's letterbox geometry, reused once released
    std::vector<FrameRef> rgb_pool_; // yuv444p frames a packed RGB input is converted into before it is resampled
    FrameRef planar_of(const FrameRef& in);
    std::vector<FrameRef> ring_;     // 2 * video_batch_ticks() of them, used in turn: the RGBA chains that read the last K may be launched together with the next K scales
    uint32_t ring_pos_ = 0;
    std::shared_ptr<ScaleTables> t_;
    DevBuf tmp_;                     // downscaling: H-filtered rows of the three planes
    int32_t* tmp_plane_[3] = {nullptr, nullptr, nullptr};
};
// the scale of `src` into `target` under tables `t`, queued (4-tap) or launched (widened) on s
void scale_into(const FrameRef& src, const std::shared_ptr<const ScaleTables>& t, const FrameRef& target, int32_t* const tmp_plane[3], hipStream_t s);

struct VideoInput { DFrame* frame = nullptr; Rational duration_hint; Rational tick_offset; };   // engine::VideoFrame, io.rs:12-17

// VideoMixer (src/module/video_mixer.rs)
class VideoMixer {
public:
    VideoMixer(const mx_video_mixer_params& p, uint32_t sample_rate, hipStream_t s);
    void update(const mx_video_mixer_params& p) { params_ = p; }   // video_mixer.rs:65-68
    bool owns_stream() const { return own_stream_; }
    // program output as an unevaluated chain (graph compiler: single in-graph video consumer)
    void set_lazy_program(bool on, uint32_t ticks_per_second) { lazy_program_ = on; tps_ = ticks_per_second ? ticks_per_second : 60; }
    // returns program / A / B frames (null FrameRef = None)
    void run_tick(uint64_t t, const VideoInput in[4], FrameRef& out, FrameRef& out_a, FrameRef& out_b);
    hipStream_t stream() const { return stream_; }
    // topology edit (Engine::client_update, src/engine.rs:277-398: the module persists, its connections change): the
    // mixer -- stored frames, expiry times, per-channel scalers -- moves to the edited graph and from now on launches on
    // THAT graph's stream, with that graph's fusion decision and tick rate
    void rebind(hipStream_t s, bool lazy_program, uint32_t ticks_per_second);
    int param_a() const { return params_.a; }
    int param_b() const { return params_.b; }
private:
    struct Stored { Rational active_until; FrameRef frame; };
    struct Channel { bool has_stored = false; Stored stored; std::unique_ptr<Scaler> scaler; };
    void rescale(Channel& ch, uint32_t tw, uint32_t th);   // video_mixer.rs:261-274
    mx_video_mixer_params params_;
    uint32_t sample_rate_;
    hipStream_t stream_;
    bool own_stream_ = false;
    bool lazy_program_ = false;
    uint32_t tps_ = 60;
    Channel ch_[4];
    // output frames are fresh each tick in the reference; here a small ring recycles them once the
    // caller has dropped its references
    std::vector<FrameRef> pool_;
    FrameRef fresh_output(uint32_t w, uint32_t h);
public:
    ~VideoMixer();
};

}  // namespace mx
