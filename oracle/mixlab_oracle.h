/*
 * mixlab_oracle.h -- CPU restatement of haileys/mixlab's per-tick module-graph hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline.  Nothing under mixlab_amd/ links, imports or calls it.
 *
 * The reference is Rust and cannot be compiled in this image (no rustc/cargo), so every function
 * below restates one reference function and cites the file:line it follows (paths relative to the
 * reference checkout).  Build flags that matter: -O2 -ffp-contract=off, no fast-math -- Rust
 * evaluates f64 expressions exactly as written with no FMA contraction.
 *
 * Parity pinning:
 *   - orc_eq_three_run is pinned bit-exactly by the reference's own golden pair
 *     fixtures/module/eq_three/chronos{,-eq}.f32.raw (src/module/eq_three.rs:150-167); a causal
 *     prefix of that pair is committed under tests/golden/.
 *   - every other audio module and the VideoMixer cross-fade have NO reference test: the source
 *     text is the only spec ("parity unpinned by reference tests").
 *   - the bicubic scaler stands in for libswscale (third-party C, ffmpeg-dev 0.3.8 git rev
 *     372167ae..., absent from the reference tree): build-specified, parity unpinned.
 *   - YUV420->RGBA has no reference counterpart at all: build-specified, parity unpinned.
 */
#ifndef MIXLAB_ORACLE_H
#define MIXLAB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- module kinds (same numbering as include/mixlab_gpu.h so tests can share descriptions) ---- */
enum {
    ORC_KIND_AMPLIFIER = 0,
    ORC_KIND_ENVELOPE = 1,
    ORC_KIND_EQ_THREE = 2,
    ORC_KIND_FM_SINE = 3,
    ORC_KIND_MIXER = 4,
    ORC_KIND_OSCILLATOR = 5,
    ORC_KIND_PLOTTER = 6,
    ORC_KIND_STEREO_PANNER = 7,
    ORC_KIND_STEREO_SPLITTER = 8,
    ORC_KIND_TRIGGER = 9,
    ORC_KIND_VIDEO_MIXER = 10,
    ORC_KIND_SOURCE_MONO = 11,   /* host-fed port: stands in for StreamInput / MediaSource audio */
    ORC_KIND_SOURCE_STEREO = 12,
    ORC_KIND_SOURCE_VIDEO = 13, ORC_KIND_VIDEO_TO_RGBA = 14,   /* not handled by the audio graph runner */
    ORC_KIND_FIR = 15,           /* build-specified (no reference module) */
    ORC_KIND_RESAMPLE = 16,      /* build-specified (no reference module) */
    ORC_KIND_COUNT = 17
};

/* protocol/src/lib.rs:233-241 (bincode variant order) */
enum { ORC_WAVE_ON = 0, ORC_WAVE_OFF = 1, ORC_WAVE_SINE = 2, ORC_WAVE_SQUARE = 3, ORC_WAVE_TRIANGLE = 4, ORC_WAVE_SAW = 5 };

/* ---- parameter structs: C mirrors of protocol/src/lib.rs ---- */
typedef struct { double gain_db; double fader; uint8_t cue; uint8_t _pad[7]; } orc_mixer_channel_params; /* :342-347 */
typedef struct { double gain_lo_db, gain_mid_db, gain_hi_db; } orc_eq_three_params;                    /* :285-290 */
typedef struct { double attack_ms, decay_ms, sustain_amplitude, release_ms; } orc_envelope_params;     /* :310-316 */
typedef struct { double amplitude, mod_depth; } orc_amplifier_params;                                  /* :298-302 */
typedef struct { double freq; uint32_t waveform; uint32_t _pad; } orc_oscillator_params;               /* :243-247 */
typedef struct { double freq_lo, freq_hi; } orc_fm_sine_params;                                        /* :292-296 */
typedef struct { uint32_t gate_open; } orc_trigger_params;                                             /* :304-308 */

/* ---- contract mode: the checker for MX_FLAG_FP_CONTRACT (see mixlab_oracle.c).  0 (default): the reference's order, never fused;
 * 1: the same expressions with every multiply fused into the add that consumes it, as explicit fma().  Process-global. ---- */
void orc_set_fp_contract(int on);
int orc_get_fp_contract(void);

/* ---- stateless helpers ---- */
double orc_decibel_to_linear(double db);                  /* protocol/src/lib.rs:469-471 */
double orc_lowpass_coeff(double freq, double sample_rate); /* src/module/eq_three.rs:113-115 */

/* ---- audio modules, one call == one ModuleT::run_tick body ---- */

/* src/module/mixer.rs:46-71. inputs[ch] may be NULL (Disconnected => zero buffer, io.rs:45-52). */
void orc_mixer_run(const orc_mixer_channel_params* ch, size_t n_ch, const float* const* inputs,
                   float* master, float* cue, size_t len);

/* src/module/eq_three.rs:13-26,58-89,100-125 */
typedef struct { double lo_f, hi_f; double lo[4], hi[4]; double history[3]; } orc_eq_three;
void orc_eq_three_init(orc_eq_three* s, double sample_rate);
void orc_eq_three_run(orc_eq_three* s, const orc_eq_three_params* p, const float* in, float* out, size_t n);

/* src/module/envelope.rs:8-58,91-120 */
typedef struct { uint32_t tag; /* 0 Initial, 1 TriggerOn, 2 TriggerOff */ uint32_t _pad; uint64_t seq; double off_amplitude; } orc_envelope;
void orc_envelope_init(orc_envelope* s);
void orc_envelope_run(orc_envelope* s, const orc_envelope_params* p, double sample_rate, uint64_t t,
                      const float* gate, float* out, size_t n);

/* src/module/amplifier.rs:38-60,71-73. control == NULL means Disconnected (mod value 1.0). */
void orc_amplifier_run(const orc_amplifier_params* p, const float* in_stereo, const float* control,
                       float* out_stereo, size_t stereo_len);

/* src/module/oscillator.rs:15-37,65-92 */
void orc_oscillator_run(const orc_oscillator_params* p, double sample_rate, uint64_t t,
                        float* mono, float* stereo, size_t n);

/* src/module/fm_sine.rs:37-56 */
void orc_fm_sine_run(const orc_fm_sine_params* p, double sample_rate, uint64_t t,
                     const float* in_mono, float* out_stereo, size_t n);

/* src/module/trigger.rs:35-48 */
void orc_trigger_run(const orc_trigger_params* p, float* out, size_t n);

/* src/module/stereo_panner.rs:30-41, stereo_splitter.rs:33-47 */
void orc_stereo_panner_run(const float* l, const float* r, float* out_stereo, size_t n);
void orc_stereo_splitter_run(const float* in_stereo, float* l, float* r, size_t n);

/* src/module/plotter.rs:37-56. Returns 1 and fills left/right when an indication fires. */
typedef struct { uint64_t count; } orc_plotter;
int orc_plotter_run(orc_plotter* s, const float* in_stereo /* NULL = disconnected */, float* left, float* right, size_t n);

/* ---- sink / ingest sample formats ---- */
void orc_f32_to_i16(const float* in, int16_t* out, size_t n);   /* src/video/encode.rs:183-195 */
void orc_i16_to_f32(const int16_t* in, float* out, size_t n);   /* src/module/stream_input.rs:167-173 */

/* ---- BUILD-SPECIFIED audio extras (no reference counterpart; DESIGN.md "FIR and resampler") ----
 * f32 widened to f64, accumulated in f64 in ascending tap index with separate multiply and add, rounded once.
 * hist: the (n_taps - 1) [resp. (P - 1)] stereo input frames before the call, interleaved; updated on return. */
void orc_fir_run(const double* taps, uint32_t n_taps, float* hist, const float* in_stereo, float* out_stereo, size_t frames);
void orc_resample_run(const double* taps /* [up][P] */, uint32_t up, uint32_t down, uint32_t P, float* hist,
                      uint64_t in_base, uint64_t out_base, const float* in_stereo, size_t in_frames,
                      float* out_stereo, size_t out_frames);

/* ---- graph runner: restates Engine::run_tick (src/engine.rs:400-510) ---- */
typedef struct { uint32_t kind; uint32_t params_len; const void* params; } orc_node;
typedef struct { uint32_t src_node, src_port, dst_node, dst_port; } orc_edge;
typedef struct orc_graph orc_graph;

/* sample_rate/ticks_per_second are compile-time 44100/60 in the reference (src/engine.rs:52-55);
 * they are parameters here only so the 48 kHz performance configuration has a CPU baseline. */
orc_graph* orc_graph_build(const orc_node* nodes, size_t n_nodes, const orc_edge* edges, size_t n_edges,
                           uint32_t sample_rate, uint32_t ticks_per_second);
void orc_graph_destroy(orc_graph* g);
size_t orc_graph_samples_per_tick(const orc_graph* g);
/* ModuleT::update (src/module/mod.rs:16): replace a node's params between ticks (same length) */
int orc_graph_update_params(orc_graph* g, uint32_t node, const void* params, uint32_t params_len);
/* host-fed source ports: pointer is read during run_tick (SPT mono / 2*SPT stereo floats) */
int orc_graph_set_source(orc_graph* g, uint32_t node, const float* samples);
/* the source replays a resident buffer of ring_ticks ticks: tick t reads block (t mod ring_ticks) */
int orc_graph_set_source_ring(orc_graph* g, uint32_t node, const float* samples, uint32_t ring_ticks);
/* one Engine::run_tick: fresh zeroed outputs, topological order, t = tick * SPT */
int orc_graph_run_tick(orc_graph* g, uint64_t tick);
/* n consecutive ticks with the sources left as they are (one foreign call: lets a threaded timing harness run free of the caller's interpreter lock) */
int orc_graph_run_ticks(orc_graph* g, uint64_t first_tick, uint32_t n);
/* borrow the output buffer a port produced in the last tick (NULL if bad ids) */
const float* orc_graph_output(const orc_graph* g, uint32_t node, uint32_t port, size_t* len);
/* plotter indication of the last tick: returns 1 if it fired, copies SPT floats to each */
int orc_graph_plotter_indication(const orc_graph* g, uint32_t node, float* left, float* right);
/* number of nodes in run order; order[i] filled */
size_t orc_graph_run_order(const orc_graph* g, uint32_t* order, size_t cap);

/* rational time helpers used by VideoMixer frame expiry (util/src/time.rs:9-75) */
typedef struct { int64_t num, den; } orc_rational;

/* ---- video: planar yuv420p 8-bit ---- */
typedef struct {
    uint32_t width, height;     /* luma dimensions (codec/src/ffmpeg/frame.rs:180-186) */
    uint8_t* data[3];           /* Y, U, V plane bases (frame.rs:188-197) */
    int32_t stride[3];          /* bytes per row, multiple of 32 (video_mixer.rs:196-201) */
    uint32_t fmt;               /* 0 = yuv420p (everything the VideoMixer produces, video_mixer.rs:282-283), 1 = yuv422p, 2 = yuv444p, 3 = nv12 (data[1] = interleaved UV, data[2] unused):
                                   formats a scaler INPUT may have (codec/src/ffmpeg/scale.rs:16-39 carries the input pixel format) */
    /* BUILD-SPECIFIED per-pixel coverage (no reference counterpart: the reference's only "alpha" is the global fader, video_mixer.rs:168): one byte per
     * LUMA sample, 255 = opaque; NULL = the layer carries none (opaque everywhere).  See orc_video_crossfade. */
    uint8_t* alpha;
    int32_t alpha_stride;
} orc_frame;
/* chroma subsampling of a format (codec/src/ffmpeg/pixfmt.rs:97-105) */
/* fmt: 0 yuv420p, 1 yuv422p, 2 yuv444p, 3 nv12, (4, 5: packed RGB -- orc_packed_rgb_to_yuv444), 6 yuv410p, 7 yuv411p, 8 yuv440p */
static inline uint32_t orc_fmt_cw(uint32_t fmt) { return (fmt == 2 || fmt == 8) ? 0u : ((fmt == 6 || fmt == 7) ? 2u : 1u); }
static inline uint32_t orc_fmt_ch(uint32_t fmt) { return (fmt == 0 || fmt == 3 || fmt == 8) ? 1u : (fmt == 6 ? 2u : 0u); }

/* codec/src/ffmpeg/frame.rs:76-138: Y=0x00, U=V=0x80 over stride*(h-1)+w bytes of each plane */
void orc_frame_blank(orc_frame* f);
/* src/module/video_mixer.rs:168 */
uint8_t orc_crossfade_factor(double fader);
/* src/module/video_mixer.rs:151-239: a/b NULL => read the (blank) output plane itself.
 * BUILD-SPECIFIED when a or b carries an alpha plane (per-pixel alpha composite, DESIGN.md "Per-pixel alpha"): per sample of every plane, with
 * aA / aB the layers' coverage there (255 where a layer carries none; a chroma sample takes the coverage of its co-sited luma sample,
 * (x << log2_chroma_w, y << log2_chroma_h)) and all arithmetic in u16 with truncating division, as fade_line's:
 *     wa  = (aA * fade) / 255                 -- the fader scales A's coverage by the reference's own truncation rule
 *     wb  = (aB * (255 - wa)) / 255           -- B takes what A leaves, as far as B covers the sample
 *     out = (A * (255 - wb) + B * wb) / 255   -- fade_line's form with the per-sample pair (255 - wb, wb)
 * aA = aB = 255 gives wa = fade, wb = 255 - fade: fade_line bit for bit.  The output carries no alpha (the VideoMixer produces opaque yuv420p). */
void orc_video_crossfade(orc_frame* out, const orc_frame* a, const orc_frame* b, uint8_t fade);
/* src/module/video_mixer.rs:276-297 */
void orc_unify_picture_settings(uint32_t aw, uint32_t ah, uint32_t bw, uint32_t bh, uint32_t* w, uint32_t* h);
/* src/video/encode.rs:354-374 */
typedef struct { uint32_t scaled_w, scaled_h, letterbox_x, letterbox_y; } orc_scale_geometry;
void orc_scaler_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h, orc_scale_geometry* g);
/* BUILD-SPECIFIED stand-in for sws_scale(SWS_BICUBIC) (codec/src/ffmpeg/scale.rs:16-39,49-70):
 * separable Catmull-Rom-family cubic (B=0,C=0.6), 14-bit coefficients, see DESIGN.md.  Scales one
 * plane. parity unpinned. */
/* taps per output sample on one axis: 4, or 2*ceil(2*src/dst)+2 when downscaling (widened kernel) */
uint32_t orc_bicubic_tap_count(uint32_t src, uint32_t dst);
void orc_bicubic_taps_n(uint32_t o, uint32_t src, uint32_t dst, int32_t* first, int32_t* coef);
void orc_scale_plane_bicubic_step(const uint8_t* src, int32_t src_stride, uint32_t xstep, uint32_t sw, uint32_t sh,
                                  uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh);
void orc_scale_plane_bicubic(const uint8_t* src, int32_t src_stride, uint32_t sw, uint32_t sh,
                             uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh);
/* BUILD-SPECIFIED row-band forms (multi-GPU sharding of one picture): rows [row0, row0 + rows) of the same
 * scale from a source SLICE (rows [src_row0, src_row0 + src_rows) of the plane); -1 when the slice lacks a needed row */
int orc_scale_plane_bicubic_rows(const uint8_t* slice, int32_t src_stride, uint32_t sw, uint32_t sh, uint32_t src_row0, uint32_t src_rows,
                                 uint8_t* dst, int32_t dst_stride, uint32_t dw, uint32_t dh, uint32_t row0, uint32_t rows);
void orc_deep_to_8(const uint8_t* const planes[3], const int32_t strides[3], uint32_t w, uint32_t h, int fmt /* 10 - 20 */, orc_frame* dst);   /* build-specified: 10- / 12- / 16-bit words -> the 8-bit frame of the layout */
void orc_yuyv_to_422p(const uint8_t* src, int32_t src_stride, uint32_t w, uint32_t h, int fmt /* 21 yuyv422, 22 uyvy422 */, orc_frame* dst);   /* a byte shuffle into yuv422p */
void orc_packed_rgb_to_yuv444(const uint8_t* src, int32_t src_stride, uint32_t w, uint32_t h, int fmt /* 4 rgb24, 5 bgra, 23 bgr24, 24 rgba, 25 argb, 26 abgr */, orc_frame* dst);   /* build-specified; dst->alpha (when not NULL) receives the A byte of the four-byte formats, 255 for the three-byte ones */
int orc_dynamic_scale_band(const orc_frame* in_slice, uint32_t in_full_h, uint32_t src_row0, orc_frame* out, uint32_t full_w, uint32_t full_h, uint32_t row0);
/* src/video/encode.rs:338-397: identity when sizes match (copies), else blank + scale into letterbox */
void orc_dynamic_scale(const orc_frame* in, orc_frame* out);
/* BUILD-SPECIFIED (no reference counterpart): BT.709 limited-range integer YUV420P -> RGBA8,
 * nearest chroma, then optional 3x4 colour matrix in Q12. parity unpinned. */
void orc_yuv420_to_rgba(const orc_frame* in, uint8_t* rgba, int32_t rgba_stride, const int32_t* matrix_q12 /* 12 or NULL */);

/* VideoMixer::run_tick (src/module/video_mixer.rs:70-250) as a state machine over owned frame copies */
typedef struct {
    int32_t a, b; double fader; uint32_t sample_rate;
    int has_stored[4]; orc_frame stored[4]; orc_rational active_until[4];
    int has_scaler[4]; uint32_t scaler_w[4], scaler_h[4];
} orc_video_mixer;
typedef struct { const orc_frame* frame; orc_rational duration_hint, tick_offset; } orc_video_input;
void orc_video_mixer_init(orc_video_mixer* m, int32_t a, int32_t b, double fader, uint32_t sample_rate);
void orc_video_mixer_free(orc_video_mixer* m);
/* out: caller-allocated planes large enough for the unified size; width/height are set on return */
int orc_video_mixer_run_tick(orc_video_mixer* m, uint64_t t, const orc_video_input in[4], orc_frame* out, int* out_present);

orc_rational orc_rational_new(int64_t num, int64_t den);
orc_rational orc_rational_add(orc_rational a, orc_rational b);
int orc_rational_cmp(orc_rational a, orc_rational b);

#ifdef __cplusplus
}
#endif
/* ---- timed ingest (mixlab_oracle_ingest.c): which frame leaves on which tick ---- */
typedef struct { int64_t frame_id; orc_rational time; orc_rational duration_hint; } orc_timed_frame;       /* media_source.rs:151-155 */
typedef struct { int64_t frame_id; /* 0 = None */ orc_rational duration_hint, tick_offset; } orc_tick_video;   /* engine::VideoFrame, io.rs:12-17 */
typedef struct orc_media_source orc_media_source;
orc_media_source* orc_media_source_new(uint32_t sample_rate, uint32_t ticks_per_second);
void orc_media_source_free(orc_media_source* m);
void orc_media_source_set_media(orc_media_source* m, int present);
int orc_media_source_send(orc_media_source* m, int64_t frame_id, orc_rational pts, orc_rational duration_hint);   /* 1 sent, 0 would block, -1 no receiver */
orc_tick_video orc_media_source_run_tick(orc_media_source* m, uint64_t t);
typedef struct orc_stream_input orc_stream_input;
orc_stream_input* orc_stream_input_new(uint32_t sample_rate);
void orc_stream_input_free(orc_stream_input* s);
void orc_stream_input_listen(orc_stream_input* s, int listening);
int orc_stream_input_write_audio(orc_stream_input* s, uint64_t source_id, orc_rational source_time, const int16_t* data, size_t n);
int orc_stream_input_write_video(orc_stream_input* s, uint64_t source_id, orc_rational source_time, int64_t frame_id, orc_rational duration_hint);
orc_tick_video orc_stream_input_run_tick(orc_stream_input* s, uint64_t t, int16_t* audio_out, size_t n_out, size_t* zero_filled);

#endif
