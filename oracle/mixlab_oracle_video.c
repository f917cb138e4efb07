/*
 * mixlab_oracle_video.c -- CPU restatement of the reference's pixel path.
 * TEST INFRASTRUCTURE ONLY (see mixlab_oracle.h).
 *
 * Reference-following: blank fill, cross-fade, picture-settings unification, scaler geometry,
 * rational frame-expiry arithmetic.  Build-specified (no reference arithmetic to follow):
 * the bicubic plane scaler (libswscale is third-party C outside the reference tree) and
 * YUV420P->RGBA.  Integer work throughout: results are compared bit-exactly.
 */
#include "mixlab_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* AvFrame::blank, codec/src/ffmpeg/frame.rs:76-138.
 * size = stride * (height - 1) + step * width  (frame.rs:126), byte = 0x80 for chroma else 0. */
void orc_frame_blank(orc_frame* f) {
    for (int plane = 0; plane < 3; plane++) {
        int is_chroma = plane > 0;                                  /* frame.rs:103-106 */
        size_t width = is_chroma ? (f->width >> orc_fmt_cw(f->fmt)) : f->width;      /* frame.rs:108-112, log2_chroma_w */
        size_t height = is_chroma ? (f->height >> orc_fmt_ch(f->fmt)) : f->height;   /* frame.rs:114-118 */
        size_t stride = (size_t)f->stride[plane];
        size_t size = stride * (height ? height - 1 : 0) + width;   /* saturating_sub(1) */
        memset(f->data[plane], is_chroma ? 0x80 : 0x00, size);
    }
}

/* src/module/video_mixer.rs:168  `(self.params.fader * 255.0) as u8` -- Rust float->int casts
 * saturate and truncate toward zero; NaN -> 0. */
uint8_t orc_crossfade_factor(double fader) {
    double v = fader * 255.0;
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

/* fade_line, src/module/video_mixer.rs:211-235: 32-byte blocks, u16 lanes,
 * out = (a*fade + b*(255-fade)) / 255 (integer division), runs while out < out+len. */
static void orc_fade_line(uint8_t* out, const uint8_t* a, const uint8_t* b, size_t len, uint8_t fade) {
    uint16_t a_fade = fade, b_fade = (uint16_t)(255 - fade);
    uint8_t* end = out + len;
    while (out < end) {
        for (int k = 0; k < 32; k++) {
            uint16_t a_comp = (uint16_t)((uint16_t)a[k] * a_fade);
            uint16_t b_comp = (uint16_t)((uint16_t)b[k] * b_fade);
            out[k] = (uint8_t)((uint16_t)(a_comp + b_comp) / 255);
        }
        a += 32; b += 32; out += 32;
    }
}

/* BUILD-SPECIFIED per-pixel alpha (mixlab_oracle.h orc_video_crossfade): fade_line with a per-sample factor pair.  aa / ab: the coverage of A / B
 * at each sample of this row, `astep` bytes apart (1 for luma, 1 << log2_chroma_w for chroma: the co-sited luma sample); NULL = 255 everywhere.
 * Same 32-sample blocks, same u16 lanes, same truncating divisions as orc_fade_line. */
static void orc_fade_line_alpha(uint8_t* out, const uint8_t* a, const uint8_t* b, size_t len, uint8_t fade, const uint8_t* aa, const uint8_t* ab, size_t astep, size_t alen) {
    uint8_t* end = out + len;
    size_t x = 0;
    while (out < end) {
        for (int k = 0; k < 32; k++, x++) {
            /* samples of the block beyond the picture's width (fade_line runs to the end of its 32-byte block) have no coverage sample: opaque */
            const uint16_t al_a = (aa && x * astep < alen) ? aa[x * astep] : 255, al_b = (ab && x * astep < alen) ? ab[x * astep] : 255;
            const uint16_t wa = (uint16_t)((uint16_t)(al_a * (uint16_t)fade) / 255);
            const uint16_t wb = (uint16_t)((uint16_t)(al_b * (uint16_t)(255 - wa)) / 255);
            const uint16_t a_comp = (uint16_t)((uint16_t)a[k] * (uint16_t)(255 - wb));
            const uint16_t b_comp = (uint16_t)((uint16_t)b[k] * wb);
            out[k] = (uint8_t)((uint16_t)(a_comp + b_comp) / 255);
        }
        a += 32; b += 32; out += 32;
    }
}

/* src/module/video_mixer.rs:151-239 */
void orc_video_crossfade(orc_frame* out, const orc_frame* a, const orc_frame* b, uint8_t fade) {
    if ((a && a->alpha) || (b && b->alpha)) {   /* BUILD-SPECIFIED: a layer with a coverage plane */
        for (int plane = 0; plane < 3; plane++) {
            const uint32_t cw = plane ? 1u : 0u, chs = plane ? 1u : 0u;   /* the composite is yuv420p (video_mixer.rs:282-283) */
            size_t width = out->width >> cw, height = out->height >> chs;
            const uint8_t* a_ptr = a ? a->data[plane] : out->data[plane];
            size_t a_ls = a ? (size_t)a->stride[plane] : (size_t)out->stride[plane];
            const uint8_t* b_ptr = b ? b->data[plane] : out->data[plane];
            size_t b_ls = b ? (size_t)b->stride[plane] : (size_t)out->stride[plane];
            for (size_t y = 0; y < height; y++)
                orc_fade_line_alpha(out->data[plane] + y * (size_t)out->stride[plane], a_ptr + y * a_ls, b_ptr + y * b_ls, width, fade,
                                    (a && a->alpha) ? a->alpha + (y << chs) * (size_t)a->alpha_stride : NULL,
                                    (b && b->alpha) ? b->alpha + (y << chs) * (size_t)b->alpha_stride : NULL, (size_t)1 << cw, out->width);
        }
        return;
    }
    for (int plane = 0; plane < 3; plane++) {
        size_t width = plane ? (out->width >> 1) : out->width;     /* video_mixer.rs:176 */
        size_t height = plane ? (out->height >> 1) : out->height;  /* video_mixer.rs:177 */
        const uint8_t* a_ptr = a ? a->data[plane] : out->data[plane];           /* :180-183 */
        size_t a_ls = a ? (size_t)a->stride[plane] : (size_t)out->stride[plane];
        const uint8_t* b_ptr = b ? b->data[plane] : out->data[plane];           /* :185-188 */
        size_t b_ls = b ? (size_t)b->stride[plane] : (size_t)out->stride[plane];
        uint8_t* o_ptr = out->data[plane];
        size_t o_ls = (size_t)out->stride[plane];
        for (size_t y = 0; y < height; y++)
            orc_fade_line(o_ptr + y * o_ls, a_ptr + y * a_ls, b_ptr + y * b_ls, width, fade);
    }
}

/* unify_picture_settings, src/module/video_mixer.rs:276-297 (yuv420p: both chroma shifts = 1) */
void orc_unify_picture_settings(uint32_t aw, uint32_t ah, uint32_t bw, uint32_t bh, uint32_t* w, uint32_t* h) {
    uint32_t width = aw > bw ? aw : bw, height = ah > bh ? ah : bh;
    *w = (width + 1u) & ~1u;
    *h = (height + 1u) & ~1u;
}

/* DynamicScaler::scale geometry, src/video/encode.rs:354-374.
 * Ratio<usize> comparisons are exact; `(scale_factor * n).to_integer()` truncates. */
void orc_scaler_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h, orc_scale_geometry* g) {
    /* min(out_w/in_w, out_h/in_h) by cross-multiplication */
    uint64_t num, den;
    if ((uint64_t)out_w * in_h <= (uint64_t)out_h * in_w) { num = out_w; den = in_w; }
    else { num = out_h; den = in_h; }
    uint32_t sw = (uint32_t)((num * in_w) / den) & ~1u;   /* align_horizontal, pixfmt.rs:104-106 */
    uint32_t sh = (uint32_t)((num * in_h) / den) & ~1u;   /* align_vertical, pixfmt.rs:108-110 */
    g->scaled_w = sw; g->scaled_h = sh;
    g->letterbox_x = ((out_w - sw) / 2) & ~1u;            /* encode.rs:373 */
    g->letterbox_y = ((out_h - sh) / 2) & ~1u;            /* encode.rs:374 */
}

/* ------------------------------------------------------------------------------------------ */
/* BUILD-SPECIFIED bicubic (stand-in for libswscale SWS_BICUBIC, codec/src/ffmpeg/scale.rs:23-27).
 * Spec (DESIGN.md "Scaler"):
 *   pos_q16(o) = floor(((2o+1) * src * 65536) / (2 * dst)) - 32768 ; first tap = (pos>>16) - 1 ;
 *   d = pos & 0xffff ; taps at distances 1+d, d, 1-d, 2-d of the cubic with B=0, C=0.6:
 *     |x|<1 : (7x^3 - 12x^2 + 5)/5         1<=|x|<2 : (-3x^3 + 15x^2 - 24x + 12)/5
 *   coefficients rounded to Q14 (round-half-up, floor division), the larger of taps 1/2 (tie: 1)
 *   absorbs the residual so each set sums to 16384; source taps clamp to the plane edge.
 *   H pass: t = (sum_k hc[k]*S[..] + 64) >> 7 (arithmetic shift, int32)
 *   V pass: D = clip_u8((sum_k vc[k]*t[..] + (1<<20)) >> 21)
 */
static int64_t floordiv(int64_t a, int64_t b) { /* b > 0 */
    int64_t q = a / b, r = a % b;
    return (r != 0 && r < 0) ? q - 1 : q;
}
static int32_t cubic_q14(int64_t X /* |x| in Q16, 0..131072 */) {
    int64_t num;
    if (X < 65536) num = 7 * X * X * X - 12 * 65536 * X * X + 5 * ((int64_t)1 << 48);
    else if (X < 131072) num = -3 * X * X * X + 15 * 65536 * X * X - 24 * ((int64_t)1 << 32) * X + 12 * ((int64_t)1 << 48);
    else return 0;
    int64_t den = 5 * ((int64_t)1 << 34);
    return (int32_t)floordiv(2 * num + den, 2 * den);
}
static void bicubic_taps(uint32_t o, uint32_t src, uint32_t dst, int32_t* first, int32_t c[4]) {
    int64_t pos = floordiv((int64_t)(2 * (int64_t)o + 1) * src * 65536, 2 * (int64_t)dst) - 32768;
    int64_t ip = pos >> 16; /* arithmetic shift = floor */
    int64_t d = pos & 0xffff;
    *first = (int32_t)ip - 1;
    c[0] = cubic_q14(65536 + d);
    c[1] = cubic_q14(d);
    c[2] = cubic_q14(65536 - d);
    c[3] = cubic_q14(131072 - d);
    int32_t resid = 16384 - (c[0] + c[1] + c[2] + c[3]);
    if (c[2] > c[1]) c[2] += resid; else c[1] += resid;
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Downscaling (src > dst on an axis) widens the kernel with the scale factor, as libswscale does for SWS_BICUBIC: the
 * same cubic stretched by s = src/dst, so every source sample between two output centres is weighted instead of skipped.
 *   N = 2*ceil(2*src/dst) + 2 taps ; first = (pos>>16) - N/2 + 1 ; tap k sits at i = first + k ;
 *   raw_k = cubic_q14( floor(|i*65536 - pos| * dst / src) ) ; coefficients = round-half-up(raw_k * 16384 / sum raw),
 *   the first largest coefficient absorbs the residual so each set sums to 16384.  Upscaling and 1:1 keep the 4-tap
 *   form above bit for bit.  Passes as before. */
uint32_t orc_bicubic_tap_count(uint32_t src, uint32_t dst) {
    if (src <= dst) return 4;
    return 2 * (uint32_t)((2 * (uint64_t)src + dst - 1) / dst) + 2;
}
void orc_bicubic_taps_n(uint32_t o, uint32_t src, uint32_t dst, int32_t* first, int32_t* c /* [orc_bicubic_tap_count] */) {
    const uint32_t n = orc_bicubic_tap_count(src, dst);
    if (n == 4) { bicubic_taps(o, src, dst, first, c); return; }
    int64_t pos = floordiv((int64_t)(2 * (int64_t)o + 1) * src * 65536, 2 * (int64_t)dst) - 32768;
    int64_t ip = pos >> 16;
    *first = (int32_t)ip - (int32_t)(n / 2) + 1;
    int64_t sum = 0;
    for (uint32_t k = 0; k < n; k++) {
        int64_t dist = ((int64_t)*first + k) * 65536 - pos;
        if (dist < 0) dist = -dist;
        c[k] = cubic_q14(floordiv(dist * dst, src));
        sum += c[k];
    }
    int64_t tot = 0; uint32_t best = 0;
    for (uint32_t k = 0; k < n; k++) {
        c[k] = (int32_t)floordiv(2 * (int64_t)c[k] * 16384 + sum, 2 * sum);
        tot += c[k];
        if (c[k] > c[best]) best = k;
    }
    c[best] += (int32_t)(16384 - tot);
}

void orc_scale_plane_bicubic(const uint8_t* src, int32_t src_stride, uint32_t sw, uint32_t sh,
                             uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh) {
    orc_scale_plane_bicubic_step(src, src_stride, 1, sw, sh, dst, dst_stride, dw, dh);
}
/* the same with source samples `xstep` bytes apart (the interleaved chroma plane of nv12: U at even, V at odd bytes) */
void orc_scale_plane_bicubic_step(const uint8_t* src, int32_t src_stride, uint32_t xstep, uint32_t sw, uint32_t sh,
                                  uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh) {
    const uint32_t hn = orc_bicubic_tap_count(sw, dw), vn = orc_bicubic_tap_count(sh, dh);
    int32_t* hfirst = (int32_t*)malloc(sizeof(int32_t) * dw);
    int32_t* hc = (int32_t*)malloc(sizeof(int32_t) * hn * dw);
    for (uint32_t x = 0; x < dw; x++) orc_bicubic_taps_n(x, sw, dw, &hfirst[x], &hc[(size_t)hn * x]);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw * sh);
    for (uint32_t y = 0; y < sh; y++) {
        const uint8_t* row = src + (size_t)y * src_stride;
        for (uint32_t x = 0; x < dw; x++) {
            int32_t acc = 0;
            for (uint32_t k = 0; k < hn; k++) acc += hc[(size_t)hn * x + k] * (int32_t)row[(size_t)xstep * (size_t)clampi(hfirst[x] + (int32_t)k, 0, (int32_t)sw - 1)];
            tmp[(size_t)y * dw + x] = (acc + 64) >> 7;
        }
    }
    int32_t* vc = (int32_t*)malloc(sizeof(int32_t) * vn);
    for (uint32_t y = 0; y < dh; y++) {
        int32_t vfirst;
        orc_bicubic_taps_n(y, sh, dh, &vfirst, vc);
        uint8_t* drow = dst + (size_t)y * dst_stride;
        for (uint32_t x = 0; x < dw; x++) {
            int32_t acc = 0;
            for (uint32_t k = 0; k < vn; k++) acc += vc[k] * tmp[(size_t)clampi(vfirst + (int32_t)k, 0, (int32_t)sh - 1) * dw + x];
            drow[x] = (uint8_t)clampi((acc + (1 << 20)) >> 21, 0, 255);
        }
    }
    free(hfirst); free(hc); free(tmp); free(vc);
}

/* BUILD-SPECIFIED (multi-GPU row bands): output rows [row0, row0 + rows) of the same scale, reading the
 * source through a SLICE that holds source rows [src_row0, src_row0 + src_rows) of a plane that is `sh` rows high -- tap indices
 * clamp against the FULL plane (as the unsharded scale does), then must fall inside the slice (the caller's halo).  Returns -1
 * if a needed row lies outside the slice.  dst points at output row `row0`. */
int orc_scale_plane_bicubic_rows(const uint8_t* slice, int32_t src_stride, uint32_t sw, uint32_t sh, uint32_t src_row0, uint32_t src_rows,
                                 uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh, uint32_t row0, uint32_t rows) {
    const uint32_t hn = orc_bicubic_tap_count(sw, dw), vn = orc_bicubic_tap_count(sh, dh);
    int32_t* hfirst = (int32_t*)malloc(sizeof(int32_t) * dw);
    int32_t* hc = (int32_t*)malloc(sizeof(int32_t) * hn * dw);
    for (uint32_t x = 0; x < dw; x++) orc_bicubic_taps_n(x, sw, dw, &hfirst[x], &hc[(size_t)hn * x]);
    int32_t* vc = (int32_t*)malloc(sizeof(int32_t) * vn);
    int32_t* trow = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw * vn);
    int rc = 0;
    for (uint32_t y = row0; y < row0 + rows && y < dh && rc == 0; y++) {
        int32_t vfirst;
        orc_bicubic_taps_n(y, sh, dh, &vfirst, vc);
        for (uint32_t k = 0; k < vn; k++) {             /* the H-filtered source rows this output row reads */
            int32_t sy = clampi(vfirst + (int32_t)k, 0, (int32_t)sh - 1);
            if (sy < (int32_t)src_row0 || sy >= (int32_t)(src_row0 + src_rows)) { rc = -1; break; }
            const uint8_t* row = slice + (size_t)(sy - (int32_t)src_row0) * src_stride;
            for (uint32_t x = 0; x < dw; x++) {
                int32_t acc = 0;
                for (uint32_t j = 0; j < hn; j++) acc += hc[(size_t)hn * x + j] * (int32_t)row[clampi(hfirst[x] + (int32_t)j, 0, (int32_t)sw - 1)];
                trow[(size_t)k * dw + x] = (acc + 64) >> 7;
            }
        }
        if (rc) break;
        uint8_t* drow = dst + (size_t)(y - row0) * dst_stride;
        for (uint32_t x = 0; x < dw; x++) {
            int32_t acc = 0;
            for (uint32_t k = 0; k < vn; k++) acc += vc[k] * trow[(size_t)k * dw + x];
            drow[x] = (uint8_t)clampi((acc + (1 << 20)) >> 21, 0, 255);
        }
    }
    free(hfirst); free(hc); free(vc); free(trow);
    return rc;
}

/* The band of DynamicScaler::scale (encode.rs:338-397) a rank of a row-band sharded job computes: luma rows [row0, row0 + out->height)
 * of the (full_w x full_h) letterboxed result, from a source slice holding luma rows [src_row0, src_row0 + in_slice->height) of a frame
 * that is in_full_h rows high (chroma: halves).  out->width == full_w.  Returns -1 when the slice lacks a row the band needs. */
int orc_dynamic_scale_band(const orc_frame* in_slice, uint32_t in_full_h, uint32_t src_row0, orc_frame* out, uint32_t full_w, uint32_t full_h, uint32_t row0) {
    orc_scale_geometry g;
    orc_scaler_geometry(in_slice->width, in_full_h, full_w, full_h, &g);
    orc_frame_blank(out);
    int rc = 0;
    for (int p = 0; p < 3 && rc == 0; p++) {
        const uint32_t c = p ? 1 : 0;
        const uint32_t b0 = row0 >> c, b1 = (row0 + out->height) >> c;             /* the band in this plane's rows */
        const uint32_t s0 = g.letterbox_y >> c, s1 = (g.letterbox_y + g.scaled_h) >> c; /* the scaled picture's rows */
        const uint32_t a = b0 > s0 ? b0 : s0, b = b1 < s1 ? b1 : s1;
        if (a >= b) continue;                                                       /* the band lies in the letterbox bars */
        rc = orc_scale_plane_bicubic_rows(in_slice->data[p], in_slice->stride[p], in_slice->width >> c, in_full_h >> c, src_row0 >> c, in_slice->height >> c,
                                          out->data[p] + (size_t)(a - b0) * out->stride[p] + (g.letterbox_x >> c), out->stride[p],
                                          g.scaled_w >> c, g.scaled_h >> c, a - s0, b - a);
    }
    return rc;
}

/* DynamicScaler::scale, src/video/encode.rs:338-397: equal settings => the frame itself (here: a
 * copy of the visible area); otherwise blank output (encode.rs:382) and scale into the letterboxed
 * sub-frame (encode.rs:386-392; sub-frame plane offsets, frame.rs:253-278). */
void orc_dynamic_scale(const orc_frame* in, orc_frame* out) {
    /* equal picture settings -- size AND pixel format (encode.rs:342-345); the output is always yuv420p */
    if (in->width == out->width && in->height == out->height && in->fmt == 0) {
        for (int p = 0; p < 3; p++) {
            uint32_t w = p ? in->width >> 1 : in->width, h = p ? in->height >> 1 : in->height;
            for (uint32_t y = 0; y < h; y++)
                memcpy(out->data[p] + (size_t)y * out->stride[p], in->data[p] + (size_t)y * in->stride[p], w);
        }
        if (out->alpha)
            for (uint32_t y = 0; y < in->height; y++) {
                if (in->alpha) memcpy(out->alpha + (size_t)y * out->alpha_stride, in->alpha + (size_t)y * in->alpha_stride, in->width);
                else memset(out->alpha + (size_t)y * out->alpha_stride, 255, in->width);
            }
        return;
    }
    orc_scale_geometry g;
    orc_scaler_geometry(in->width, in->height, out->width, out->height, &g);
    orc_frame_blank(out);
    /* BUILD-SPECIFIED: the coverage plane is resampled like the luma plane (same taps, same passes); the letterbox bars are opaque (the blank
     * frame they come from is, encode.rs:382) */
    if (out->alpha) {
        for (uint32_t y = 0; y < out->height; y++) memset(out->alpha + (size_t)y * out->alpha_stride, 255, out->width);
        if (in->alpha && g.scaled_w && g.scaled_h)
            orc_scale_plane_bicubic(in->alpha, in->alpha_stride, in->width, in->height,
                                    out->alpha + (size_t)g.letterbox_y * out->alpha_stride + g.letterbox_x, out->alpha_stride, g.scaled_w, g.scaled_h);
    }
    /* A picture so thin that its aligned scaled size has no rows or columns: the reference hands sws_getContext a zero dimension, gets
     * NULL and panics (codec/src/ffmpeg/scale.rs:22-33).  BUILD-SPECIFIED instead of a panic: the blank letterbox frame. */
    if (g.scaled_w == 0 || g.scaled_h == 0) return;
    for (int p = 0; p < 3; p++) {
        uint32_t sh_ = p ? 1 : 0;
        uint8_t* dst = out->data[p] + (size_t)(g.letterbox_y >> sh_) * out->stride[p] + (g.letterbox_x >> sh_);
        /* BUILD-SPECIFIED format conversion: every plane is resampled from ITS size in the input format to its size in the yuv420p output */
        uint32_t sw = p ? in->width >> orc_fmt_cw(in->fmt) : in->width, sh = p ? in->height >> orc_fmt_ch(in->fmt) : in->height;
        if (in->fmt == 3 && p)   /* nv12: both chroma planes live interleaved in data[1] (U even bytes, V odd bytes) */
            orc_scale_plane_bicubic_step(in->data[1] + (p - 1), in->stride[1], 2, sw, sh, dst, out->stride[p], g.scaled_w >> sh_, g.scaled_h >> sh_);
        else
            orc_scale_plane_bicubic(in->data[p], in->stride[p], sw, sh, dst, out->stride[p], g.scaled_w >> sh_, g.scaled_h >> sh_);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* BUILD-SPECIFIED: BT.709 limited-range YUV420P -> RGBA8 (nearest chroma), optional Q12 3x4 matrix.
 *   C=Y-16 D=U-128 E=V-128
 *   R=clip((298C + 459E + 128)>>8) G=clip((298C - 55D - 136E + 128)>>8) B=clip((298C + 541D + 128)>>8) A=255
 *   matrix: out_c = clip((m[c][0]R + m[c][1]G + m[c][2]B + m[c][3] + 2048) >> 12) */
void orc_yuv420_to_rgba(const orc_frame* in, uint8_t* rgba, int32_t rgba_stride, const int32_t* m) {
    for (uint32_t y = 0; y < in->height; y++) {
        const uint8_t* yr = in->data[0] + (size_t)y * in->stride[0];
        const uint8_t* ur = in->data[1] + (size_t)(y >> 1) * in->stride[1];
        const uint8_t* vr = in->data[2] + (size_t)(y >> 1) * in->stride[2];
        uint8_t* o = rgba + (size_t)y * rgba_stride;
        for (uint32_t x = 0; x < in->width; x++) {
            int32_t C = (int32_t)yr[x] - 16, D = (int32_t)ur[x >> 1] - 128, E = (int32_t)vr[x >> 1] - 128;
            int32_t R = clampi((298 * C + 459 * E + 128) >> 8, 0, 255);
            int32_t G = clampi((298 * C - 55 * D - 136 * E + 128) >> 8, 0, 255);
            int32_t B = clampi((298 * C + 541 * D + 128) >> 8, 0, 255);
            if (m) {
                int32_t r2 = clampi((m[0] * R + m[1] * G + m[2] * B + m[3] + 2048) >> 12, 0, 255);
                int32_t g2 = clampi((m[4] * R + m[5] * G + m[6] * B + m[7] + 2048) >> 12, 0, 255);
                int32_t b2 = clampi((m[8] * R + m[9] * G + m[10] * B + m[11] + 2048) >> 12, 0, 255);
                R = r2; G = g2; B = b2;
            }
            o[4 * x + 0] = (uint8_t)R; o[4 * x + 1] = (uint8_t)G; o[4 * x + 2] = (uint8_t)B; o[4 * x + 3] = 255;
        }
    }
}

/* BUILD-SPECIFIED (DESIGN.md "Pixel formats"; the reference's scaler context takes ANY AVPixelFormat, codec/src/ffmpeg/scale.rs:16-39,
 * pixfmt.rs:51-111 -- libswscale's own RGB input conversion is unknown here: parity unpinned).  A packed RGB scaler input stands for the
 * yuv444p frame of its per-pixel conversion, BT.709 limited range with the usual 8-bit integer coefficients (round(219/255 K 256),
 * round(224/255 K' 256): tests/golden/make_rgb_matrix.py derives them from exact rationals):
 *   Y = ((47 R + 157 G + 16 B + 128) >> 8) + 16,  U = ((-26 R - 87 G + 112 B + 128) >> 8) + 128,  V = ((112 R - 102 G - 10 B + 128) >> 8) + 128
 * fmt 4 = rgb24 (R, G, B bytes), 5 = bgra (B, G, R, A bytes; alpha ignored), 23 bgr24, 24 rgba, 25 argb, 26 abgr.  dst: a yuv444p frame of the same size. */
void orc_packed_rgb_to_yuv444(const uint8_t* src, int32_t src_stride, uint32_t w, uint32_t h, int fmt, orc_frame* dst) {
    /* 4 rgb24, 5 bgra, 23 bgr24, 24 rgba, 25 argb, 26 abgr (include/mixlab_gpu.h mx_pixfmt) */
    const int bpp = (fmt == 4 || fmt == 23) ? 3 : 4;
    const int ri = fmt == 4 || fmt == 24 ? 0 : (fmt == 25 ? 1 : (fmt == 26 ? 3 : 2));
    const int gi = fmt == 25 || fmt == 26 ? 2 : 1;
    const int bi = fmt == 4 || fmt == 24 ? 2 : (fmt == 25 ? 3 : (fmt == 26 ? 1 : 0));
    const int ai = (fmt == 25 || fmt == 26) ? 0 : 3;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = src + (size_t)y * src_stride;
        for (uint32_t x = 0; x < w; x++) {
            const int32_t R = row[bpp * x + ri], G = row[bpp * x + gi], B = row[bpp * x + bi];
            dst->data[0][(size_t)y * dst->stride[0] + x] = (uint8_t)(((47 * R + 157 * G + 16 * B + 128) >> 8) + 16);
            dst->data[1][(size_t)y * dst->stride[1] + x] = (uint8_t)(((-26 * R - 87 * G + 112 * B + 128) >> 8) + 128);
            dst->data[2][(size_t)y * dst->stride[2] + x] = (uint8_t)(((112 * R - 102 * G - 10 * B + 128) >> 8) + 128);
            if (dst->alpha) dst->alpha[(size_t)y * dst->alpha_stride + x] = bpp == 4 ? row[bpp * x + ai] : 255;   /* the A byte is the pixel's coverage (straight, not premultiplied) */
        }
    }
}

/* BUILD-SPECIFIED (include/mixlab_gpu.h mx_pixfmt 10 - 20; the reference hands any AVPixelFormat to libswscale, scale.rs:16-39, and reads the bit depth off the
 * descriptor, pixfmt.rs:107-111 -- what libswscale does with the extra bits is unknown here: parity unpinned).  A scaler input deeper than 8 bits stands for the
 * 8-bit frame of the same layout: sample = min(255, (v + 2^(b-9)) >> (b - 8)), v = (word >> shift) & (2^b - 1).
 * fmt: 10 / 11 / 12 yuv420p10 / 422p10 / 444p10 (b 10, shift 0), 13 p010 (b 10, shift 6), 14 / 15 / 16 the 12-bit planar ones, 17 / 18 / 19 the 16-bit ones, 20 p016.
 * planes[p]: 16-bit little-endian words, stride in BYTES; p010 / p016: plane 1 = interleaved U, V words, planes[2] ignored.
 * dst: the 8-bit frame of the layout (fmt 0 / 1 / 2) and the same size. */
void orc_deep_to_8(const uint8_t* const planes[3], const int32_t strides[3], uint32_t w, uint32_t h, int fmt, orc_frame* dst) {
    const uint32_t bits = fmt <= 13 ? 10u : (fmt <= 16 ? 12u : 16u);
    const uint32_t shift = fmt == 13 ? 6u : 0u;
    const int is_semi = fmt == 13 || fmt == 20;
    for (int p = 0; p < 3; p++) {
        const uint32_t pw = p ? w >> orc_fmt_cw(dst->fmt) : w, ph = p ? h >> orc_fmt_ch(dst->fmt) : h;
        const int semi = is_semi && p;
        const uint8_t* base = semi ? planes[1] : planes[p];
        const int32_t stride = semi ? strides[1] : strides[p];
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t* row = base + (size_t)y * stride;
            for (uint32_t x = 0; x < pw; x++) {
                const size_t word = semi ? 2u * (size_t)x + (p == 2 ? 1u : 0u) : (size_t)x;
                const uint32_t v = (((uint32_t)row[2 * word] | ((uint32_t)row[2 * word + 1] << 8)) >> shift) & ((1u << bits) - 1u);
                const uint32_t o = (v + (1u << (bits - 9u))) >> (bits - 8u);
                dst->data[p][(size_t)y * dst->stride[p] + x] = (uint8_t)(o > 255u ? 255u : o);
            }
        }
    }
}

/* Packed 4:2:2 (include/mixlab_gpu.h mx_pixfmt 21 yuyv422: Y0 U Y1 V; 22 uyvy422: U Y0 V Y1) -> the yuv422p frame with the same samples: a byte shuffle. */
void orc_yuyv_to_422p(const uint8_t* src, int32_t src_stride, uint32_t w, uint32_t h, int fmt, orc_frame* dst) {
    const int yo = fmt == 21 ? 0 : 1, co = 1 - yo;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = src + (size_t)y * src_stride;
        for (uint32_t x = 0; x < w; x++) dst->data[0][(size_t)y * dst->stride[0] + x] = row[2 * x + yo];
        for (uint32_t x = 0; x < w / 2; x++) {
            dst->data[1][(size_t)y * dst->stride[1] + x] = row[4 * x + co];
            dst->data[2][(size_t)y * dst->stride[2] + x] = row[4 * x + co + 2];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Rational64 arithmetic as used by MediaTime / MediaDuration (util/src/time.rs:9-75): always kept
 * reduced with a positive denominator, like num_rational::Ratio::new. */
static int64_t gcd64(int64_t a, int64_t b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { int64_t t = a % b; a = b; b = t; } return a ? a : 1; }
orc_rational orc_rational_new(int64_t num, int64_t den) {
    if (den < 0) { num = -num; den = -den; }
    int64_t g = gcd64(num, den);
    orc_rational r = { num / g, den / g };
    return r;
}
orc_rational orc_rational_add(orc_rational a, orc_rational b) {
    int64_t g = gcd64(a.den, b.den);
    int64_t lcm = a.den / g * b.den;
    return orc_rational_new(a.num * (lcm / a.den) + b.num * (lcm / b.den), lcm);
}
int orc_rational_cmp(orc_rational a, orc_rational b) {
    __int128 l = (__int128)a.num * b.den, r = (__int128)b.num * a.den;
    return l < r ? -1 : (l > r ? 1 : 0);
}

/* ------------------------------------------------------------------------------------------ */
/* VideoMixer::run_tick, src/module/video_mixer.rs:70-250, as a plain state machine over owned
 * frame copies (an AVFrame refcount clone of an immutable frame is observationally a copy). */
static void vm_frame_alloc(orc_frame* f, uint32_t w, uint32_t h, int with_alpha) {
    f->width = w; f->height = h; f->fmt = 0;   /* yuv420p, video_mixer.rs:282-283 */
    for (int p = 0; p < 3; ++p) {
        uint32_t pw = p ? w >> 1 : w, ph = p ? h >> 1 : h;
        f->stride[p] = (int32_t)((pw + 63u) & ~63u);
        f->data[p] = (uint8_t*)malloc((size_t)f->stride[p] * (ph ? ph : 1));
    }
    f->alpha = NULL; f->alpha_stride = 0;
    if (with_alpha) { f->alpha_stride = (int32_t)((w + 63u) & ~63u); f->alpha = (uint8_t*)malloc((size_t)f->alpha_stride * (h ? h : 1)); }   /* build-specified: the layer's coverage travels with it */
}
static void vm_frame_free(orc_frame* f) { for (int p = 0; p < 3; ++p) { free(f->data[p]); f->data[p] = NULL; } free(f->alpha); f->alpha = NULL; f->width = f->height = 0; }
static void vm_frame_copy_from(orc_frame* dst, const orc_frame* src) {   /* dst freshly allocated with src's size */
    for (int p = 0; p < 3; ++p) {
        uint32_t pw = p ? src->width >> 1 : src->width, ph = p ? src->height >> 1 : src->height;
        for (uint32_t y = 0; y < ph; ++y) memcpy(dst->data[p] + (size_t)y * dst->stride[p], src->data[p] + (size_t)y * src->stride[p], pw);
    }
}

void orc_video_mixer_init(orc_video_mixer* m, int32_t a, int32_t b, double fader, uint32_t sample_rate) {
    memset(m, 0, sizeof *m);
    m->a = a; m->b = b; m->fader = fader; m->sample_rate = sample_rate ? sample_rate : 44100;
}
void orc_video_mixer_free(orc_video_mixer* m) {
    for (int i = 0; i < 4; ++i) if (m->has_stored[i]) { vm_frame_free(&m->stored[i]); m->has_stored[i] = 0; }
}

/* Channel::rescale, video_mixer.rs:261-274 */
static void vm_rescale(orc_video_mixer* m, int i, uint32_t tw, uint32_t th) {
    if (!m->has_scaler[i] || m->scaler_w[i] != tw || m->scaler_h[i] != th) {
        m->has_scaler[i] = 1; m->scaler_w[i] = tw; m->scaler_h[i] = th;
        if (m->has_stored[i]) {
            orc_frame scaled; vm_frame_alloc(&scaled, tw, th, m->stored[i].alpha != NULL);
            orc_dynamic_scale(&m->stored[i], &scaled);
            vm_frame_free(&m->stored[i]);
            m->stored[i] = scaled;
        }
    }
}

int orc_video_mixer_run_tick(orc_video_mixer* m, uint64_t t, const orc_video_input in[4], orc_frame* out, int* out_present) {
    *out_present = 0;
    orc_rational now = orc_rational_new((int64_t)t, (int64_t)m->sample_rate);           /* :92 */
    for (int i = 0; i < 4; ++i)                                                          /* expire, :94-101 */
        if (m->has_stored[i] && orc_rational_cmp(now, m->active_until[i]) >= 0) { vm_frame_free(&m->stored[i]); m->has_stored[i] = 0; }
    int have = 0; uint32_t tw = 0, th = 0;                                               /* target, :104-119 */
    for (int i = 0; i < 4; ++i) {
        uint32_t w, h;
        if (in[i].frame) { w = in[i].frame->width; h = in[i].frame->height; }
        else if (m->has_stored[i]) { w = m->stored[i].width; h = m->stored[i].height; }
        else continue;
        if (!have) { tw = w; th = h; have = 1; } else orc_unify_picture_settings(tw, th, w, h, &tw, &th);
    }
    if (!have) return 0;
    for (int i = 0; i < 4; ++i) {                                                        /* new inputs, :122-148 */
        if (in[i].frame) {
            if (m->has_stored[i]) { vm_frame_free(&m->stored[i]); m->has_stored[i] = 0; }
            vm_rescale(m, i, tw, th);
            vm_frame_alloc(&m->stored[i], tw, th, in[i].frame->alpha != NULL);
            orc_dynamic_scale(in[i].frame, &m->stored[i]);
            m->active_until[i] = orc_rational_add(orc_rational_add(now, in[i].tick_offset), in[i].duration_hint);
            m->has_stored[i] = 1;
        } else {
            vm_rescale(m, i, tw, th);
        }
    }
    /* compose, :150-239 (out must have room for tw x th with its own strides) */
    out->width = tw; out->height = th;
    orc_frame_blank(out);
    const orc_frame* fa = (m->a >= 0 && m->a < 4 && m->has_stored[m->a]) ? &m->stored[m->a] : NULL;
    const orc_frame* fb = (m->b >= 0 && m->b < 4 && m->has_stored[m->b]) ? &m->stored[m->b] : NULL;
    orc_video_crossfade(out, fa, fb, orc_crossfade_factor(m->fader));
    *out_present = 1;
    (void)vm_frame_copy_from;
    return 0;
}
