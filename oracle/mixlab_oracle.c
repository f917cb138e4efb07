/*
 * mixlab_oracle.c -- CPU restatement of the reference's audio modules and graph scheduler.
 * TEST INFRASTRUCTURE ONLY (see mixlab_oracle.h).  Compile with -O2 -ffp-contract=off.
 *
 * Rust semantics preserved: `f64 as f32` = round-to-nearest-even (C cast under the default
 * rounding mode); f64 expressions evaluated in source order, never contracted to FMA;
 * f64::sin / f64::powf = the platform libm (glibc here), as Rust's std does on Linux.
 */
#include "mixlab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* CONTRACT MODE (test infrastructure for MX_FLAG_FP_CONTRACT, include/mixlab_gpu.h).  Off (the default) every expression is
 * evaluated as the reference writes it, never fused.  On, the SAME expressions are evaluated with each multiply fused into the add
 * that consumes it -- spelled out with fma(), never left to the compiler (this file is built with -ffp-contract=off either way):
 *   eq_three.rs:117-124  p0 += fma(f, x - p0, VSA);  p_k = fma(f, p_{k-1} - p_k, p_k)
 *   eq_three.rs:76-88    out = fma(hi, g_hi, fma(mid, g_mid, lo * g_lo))
 *   envelope.rs:46-47    sustain + (1 - sustain) * decay  = fma(1 - sustain, decay, sustain)
 *   amplifier.rs:71-73   (1 - d) + d * mod                = fma(d, mod, 1 - d)
 *   Fir / Resample       acc = fma(h[k], x, acc), ascending k
 * It is what the device's contracted kernels must reproduce bit for bit; tests bound its distance from the exact mode (<= 1 ULP
 * of every f32 output).  The flag is process-global and read once per call. */
static int g_fp_contract = 0;
void orc_set_fp_contract(int on) { g_fp_contract = on ? 1 : 0; }
int orc_get_fp_contract(void) { return g_fp_contract; }
#define ORC_INLINE static inline __attribute__((always_inline))
ORC_INLINE double orc_mul_add(const int fc, double a, double b, double c) { return fc ? fma(a, b, c) : a * b + c; }

/* std::f64::consts::PI */
#define ORC_PI 3.14159265358979323846264338327950288

/* ------------------------------------------------------------------------------------------ */
/* protocol/src/lib.rs:469-471  Decibel::to_linear = f64::powf(10.0, self.0 / 20.0) */
double orc_decibel_to_linear(double db) { return pow(10.0, db / 20.0); }

/* src/module/eq_three.rs:113-115  freq = 2.0 * sin(PI * freq / SAMPLE_RATE) */
double orc_lowpass_coeff(double freq, double sample_rate) { return 2.0 * sin(ORC_PI * freq / sample_rate); }

/* ------------------------------------------------------------------------------------------ */
/* src/module/mixer.rs:46-71 */
void orc_mixer_run(const orc_mixer_channel_params* ch, size_t n_ch, const float* const* inputs,
                   float* master, float* cue, size_t len) {
    /* util::zero(master); util::zero(cue);  (mixer.rs:54-55, util.rs:26-30) */
    for (size_t i = 0; i < len; i++) master[i] = 0.0f;
    for (size_t i = 0; i < len; i++) cue[i] = 0.0f;

    for (size_t c = 0; c < n_ch; c++) {
        const float* input = inputs[c];
        /* channel.fader * channel.gain.to_linear()  (mixer.rs:59) */
        double channel_gain = ch[c].fader * orc_decibel_to_linear(ch[c].gain_db);
        for (size_t i = 0; i < len; i++) {
            float x = input ? input[i] : 0.0f; /* Disconnected => ZERO_BUFFER_STEREO (io.rs:45-52) */
            master[i] += (float)((double)x * channel_gain); /* mixer.rs:62 */
            if (ch[c].cue) cue[i] += x;                      /* mixer.rs:64-66 */
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/module/eq_three.rs */
#define ORC_FREQ_LO 420.0                 /* eq_three.rs:8 */
#define ORC_FREQ_HI 2700.0                /* eq_three.rs:9 */
#define ORC_VSA (1.0 / 4294967295.0)      /* eq_three.rs:11 */

void orc_eq_three_init(orc_eq_three* s, double sample_rate) {
    memset(s, 0, sizeof *s);
    s->lo_f = orc_lowpass_coeff(ORC_FREQ_LO, sample_rate); /* eq_three.rs:34,107-115 */
    s->hi_f = orc_lowpass_coeff(ORC_FREQ_HI, sample_rate); /* eq_three.rs:35 */
}

/* LowPass::pump, eq_three.rs:117-124 */
ORC_INLINE double orc_pump(const int fc, double f, double* p, double sample) {
    if (fc) {
        p[0] += fma(f, sample - p[0], ORC_VSA);
        p[1] = fma(f, p[0] - p[1], p[1]);
        p[2] = fma(f, p[1] - p[2], p[2]);
        p[3] = fma(f, p[2] - p[3], p[3]);
        return p[3];
    }
    p[0] += f * (sample - p[0]) + ORC_VSA;
    p[1] += f * (p[0] - p[1]);
    p[2] += f * (p[1] - p[2]);
    p[3] += f * (p[2] - p[3]);
    return p[3];
}

/* EqThree::run_tick, eq_three.rs:58-89 */
ORC_INLINE void eq_three_run_impl(const int fc, orc_eq_three* s, const orc_eq_three_params* p, const float* in, float* out, size_t n) {
    double gain_lo = orc_decibel_to_linear(p->gain_lo_db);
    double gain_mid = orc_decibel_to_linear(p->gain_mid_db);
    double gain_hi = orc_decibel_to_linear(p->gain_hi_db);
    for (size_t i = 0; i < n; i++) {
        double sample = (double)(in ? in[i] : 0.0f);
        double lo = orc_pump(fc, s->lo_f, s->lo, sample);
        double hi = s->history[0] - orc_pump(fc, s->hi_f, s->hi, sample);
        double mid = s->history[0] - (hi + lo);
        s->history[0] = s->history[1];
        s->history[1] = s->history[2];
        s->history[2] = sample;
        if (fc) { out[i] = (float)fma(hi, gain_hi, fma(mid, gain_mid, lo * gain_lo)); continue; }
        lo = lo * gain_lo;
        mid = mid * gain_mid;
        hi = hi * gain_hi;
        out[i] = (float)(lo + mid + hi);
    }
}
void orc_eq_three_run(orc_eq_three* s, const orc_eq_three_params* p, const float* in, float* out, size_t n) {
    if (g_fp_contract) eq_three_run_impl(1, s, p, in, out, n); else eq_three_run_impl(0, s, p, in, out, n);
}

/* ------------------------------------------------------------------------------------------ */
/* src/module/envelope.rs */
void orc_envelope_init(orc_envelope* s) { memset(s, 0, sizeof *s); }

/* envelope.rs:16-18 */
static inline double orc_seq_ms(uint64_t first, uint64_t last, double sample_rate) {
    return (double)(last - first) / sample_rate * 1000.0;
}
/* envelope.rs:20-28 */
static inline double orc_clamp01(double x) { return x > 1.0 ? 1.0 : (x < 0.0 ? 0.0 : x); }

/* envelope.rs:34-58 */
static double orc_env_amplitude(const orc_envelope_params* p, const orc_envelope* s, double sr, uint64_t t) {
    switch (s->tag) {
    default:
    case 0: return 0.0;
    case 1: {
        double ms_since_on = orc_seq_ms(s->seq, t, sr);
        if (ms_since_on < p->attack_ms) {
            return 1.0 / p->attack_ms * ms_since_on;
        } else {
            double ms_since_decay_started = ms_since_on - p->attack_ms;
            double decay_amplitude = 1.0 - orc_clamp01(1.0 / p->decay_ms * ms_since_decay_started);
            return orc_mul_add(g_fp_contract, 1.0 - p->sustain_amplitude, decay_amplitude, p->sustain_amplitude);
        }
    }
    case 2: {
        double ms_since_off = orc_seq_ms(s->seq, t, sr);
        double release_amplitude = 1.0 - orc_clamp01(1.0 / p->release_ms * ms_since_off);
        return s->off_amplitude * release_amplitude;
    }
    }
}

/* Envelope::run_tick, envelope.rs:91-120 */
void orc_envelope_run(orc_envelope* s, const orc_envelope_params* p, double sample_rate, uint64_t t,
                      const float* gate, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint64_t sample_seq = t + (uint64_t)i;
        float g = gate ? gate[i] : 0.0f;
        if (s->tag == 0 || s->tag == 2) {
            if (g == 1.0f) { s->tag = 1; s->seq = sample_seq; }
        } else {
            if (g == 0.0f) {
                double amp = orc_env_amplitude(p, s, sample_rate, sample_seq);
                s->tag = 2; s->seq = sample_seq; s->off_amplitude = amp;
            }
        }
        out[i] = (float)orc_env_amplitude(p, s, sample_rate, sample_seq);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/module/amplifier.rs:38-60,71-73 */
void orc_amplifier_run(const orc_amplifier_params* p, const float* in_stereo, const float* control,
                       float* out_stereo, size_t stereo_len) {
    double mod_depth = p->mod_depth, amplitude = p->amplitude;
    const int fc = g_fp_contract;
    for (size_t i = 0; i < stereo_len; i++) {
        double mod_value = control ? (double)control[i / 2] : 1.0; /* amplifier.rs:54 */
        double depth = orc_mul_add(fc, mod_depth, mod_value, 1.0 - mod_depth);    /* amplifier.rs:71-73: 1.0 - mod_depth + mod_depth * mod_value */
        float x = in_stereo ? in_stereo[i] : 0.0f;
        out_stereo[i] = (float)((double)x * depth * amplitude);    /* amplifier.rs:56 */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/module/oscillator.rs:15-37 */
static inline double orc_sign(double n) { return signbit(n) ? -1.0 : 1.0; }
static inline double orc_sine(double n) { return sin(n * 2.0 * ORC_PI); }
static inline double orc_saw(double n) { return 2.0 * (n - floor(0.5 + n)); }
static inline double orc_triangle(double n) { return 2.0 * fabs(orc_saw(n)) - 1.0; }

/* Oscillator::run_tick, oscillator.rs:65-92 */
void orc_oscillator_run(const orc_oscillator_params* p, double sample_rate, uint64_t t,
                        float* mono, float* stereo, size_t n) {
    for (size_t i = 0; i < n; i++) {
        double t0 = (double)(t + (uint64_t)i) / sample_rate;
        double nn = t0 * p->freq;
        double v;
        switch (p->waveform) {
        case ORC_WAVE_SINE: v = orc_sine(nn); break;
        case ORC_WAVE_SQUARE: v = orc_sign(orc_sine(nn)); break;
        case ORC_WAVE_SAW: v = orc_saw(nn); break;
        case ORC_WAVE_TRIANGLE: v = orc_triangle(nn); break;
        case ORC_WAVE_ON: v = 1.0; break;
        default: v = 0.0; break;
        }
        float sample = (float)v;
        if (mono) mono[i] = sample;
        if (stereo) { stereo[i * 2 + 0] = sample; stereo[i * 2 + 1] = sample; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* FmSine::run_tick, src/module/fm_sine.rs:37-56 */
void orc_fm_sine_run(const orc_fm_sine_params* p, double sample_rate, uint64_t t,
                     const float* in_mono, float* out_stereo, size_t n) {
    double freq_amp = (p->freq_hi - p->freq_lo) / 2.0;
    double freq_mid = p->freq_lo + freq_amp;
    for (size_t i = 0; i < n; i++) {
        double tt = (double)(t + (uint64_t)i) / sample_rate;
        double x_in = (double)(in_mono ? in_mono[i] : 0.0f);
        double co = (freq_mid + freq_amp * x_in) * 2.0 * ORC_PI;
        double x = sin(co * tt);
        out_stereo[i * 2 + 0] = (float)x;
        out_stereo[i * 2 + 1] = (float)x;
    }
}

/* Trigger::run_tick, src/module/trigger.rs:35-48 */
void orc_trigger_run(const orc_trigger_params* p, float* out, size_t n) {
    float v = p->gate_open ? 1.0f : 0.0f;
    for (size_t i = 0; i < n; i++) out[i] = v;
}

/* StereoPanner::run_tick, src/module/stereo_panner.rs:30-41 */
void orc_stereo_panner_run(const float* l, const float* r, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i * 2 + 0] = l ? l[i] : 0.0f;
        out[i * 2 + 1] = r ? r[i] : 0.0f;
    }
}

/* StereoSplitter::run_tick, src/module/stereo_splitter.rs:33-47 */
void orc_stereo_splitter_run(const float* in, float* l, float* r, size_t n) {
    for (size_t i = 0; i < n; i++) {
        l[i] = in ? in[i * 2 + 0] : 0.0f;
        r[i] = in ? in[i * 2 + 1] : 0.0f;
    }
}

/* Plotter::run_tick, src/module/plotter.rs:37-56 */
int orc_plotter_run(orc_plotter* s, const float* in, float* left, float* right, size_t n) {
    s->count += 1;
    if (s->count % 6 == 0 && in != NULL) {
        for (size_t i = 0; i < n; i++) { left[i] = in[i * 2]; right[i] = in[i * 2 + 1]; }
        return 1;
    }
    return 0;
}

/* AudioCtx::send_audio, src/video/encode.rs:183-195: clamp, * i16::max_value() as f32, `as i16`
 * (Rust float->int casts saturate, truncate toward zero, and map NaN to 0) */
void orc_f32_to_i16(const float* in, int16_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float sample = in[i];
        if (sample > 1.0f) sample = 1.0f; else if (sample < -1.0f) sample = -1.0f;
        float v = sample * 32767.0f;
        int r;
        if (!(v == v)) r = 0; else if (v >= 32767.0f) r = 32767; else if (v <= -32768.0f) r = -32768; else r = (int)v;
        out[i] = (int16_t)r;
    }
}
/* convert_sample, src/module/stream_input.rs:167-173 */
void orc_i16_to_f32(const int16_t* in, float* out, size_t n) {
    float divisor = -(float)INT16_MIN;
    for (size_t i = 0; i < n; i++) out[i] = (float)in[i] / divisor;
}

/* ------------------------------------------------------------------------------------------ */
/* BUILD-SPECIFIED: K-tap FIR on an interleaved stereo stream.  x[m < 0] = hist[(K-1) + m]. */
static inline float fir_x(const float* hist, uint32_t H, const float* in, long long f, int ch) {
    if (f >= 0) return in ? in[2 * f + ch] : 0.0f;
    long long h = (long long)H + f;
    return h >= 0 ? hist[2 * h + ch] : 0.0f;
}
static void update_hist(float* hist, uint32_t H, const float* in, size_t frames) {
    float* tmp = (float*)malloc(sizeof(float) * 2 * (H ? H : 1));
    for (uint32_t j = 0; j < H; j++) {
        long long f = (long long)frames - H + j;
        tmp[2 * j] = fir_x(hist, H, in, f, 0); tmp[2 * j + 1] = fir_x(hist, H, in, f, 1);
    }
    memcpy(hist, tmp, sizeof(float) * 2 * H);
    free(tmp);
}
void orc_fir_run(const double* taps, uint32_t n_taps, float* hist, const float* in, float* out, size_t frames) {
    const uint32_t H = n_taps - 1;
    const int fc = g_fp_contract;
    for (size_t n = 0; n < frames; n++) {
        double al = 0.0, ar = 0.0;
        for (uint32_t k = 0; k < n_taps; k++) {
            al = orc_mul_add(fc, taps[k], (double)fir_x(hist, H, in, (long long)n - k, 0), al);
            ar = orc_mul_add(fc, taps[k], (double)fir_x(hist, H, in, (long long)n - k, 1), ar);
        }
        out[2 * n] = (float)al; out[2 * n + 1] = (float)ar;
    }
    update_hist(hist, H, in, frames);
}
/* BUILD-SPECIFIED: rational polyphase resampler; absolute output index M: n = floor(M*down/up), phase = (M*down) mod up */
void orc_resample_run(const double* taps, uint32_t up, uint32_t down, uint32_t P, float* hist,
                      uint64_t in_base, uint64_t out_base, const float* in, size_t in_frames, float* out, size_t out_frames) {
    const uint32_t H = P - 1;
    const int fc = g_fp_contract;
    for (size_t m = 0; m < out_frames; m++) {
        uint64_t num = (out_base + m) * (uint64_t)down;
        uint64_t n_abs = num / up;
        uint32_t phase = (uint32_t)(num - n_abs * up);
        long long n = (long long)(n_abs - in_base);
        const double* h = taps + (size_t)phase * P;
        double al = 0.0, ar = 0.0;
        for (uint32_t k = 0; k < P; k++) {
            al = orc_mul_add(fc, h[k], (double)fir_x(hist, H, in, n - k, 0), al);
            ar = orc_mul_add(fc, h[k], (double)fir_x(hist, H, in, n - k, 1), ar);
        }
        out[2 * m] = (float)al; out[2 * m + 1] = (float)ar;
    }
    update_hist(hist, H, in, in_frames);
}

/* ------------------------------------------------------------------------------------------ */
/* Graph runner: Engine::run_tick, src/engine.rs:400-510 */

enum { LT_MONO = 1, LT_STEREO = 2, LT_VIDEO = 3 };

typedef struct {
    uint32_t kind;
    uint32_t n_in, n_out;
    uint8_t* in_type;   /* line types, module inputs()/outputs() */
    uint8_t* out_type;
    int64_t* in_src_node;  /* connection per input terminal (-1 = none), workspace.connections */
    uint32_t* in_src_port;
    float** out_buf;       /* this tick's output buffers (src/engine.rs:470-472) */
    void* params;          /* owned copy */
    uint32_t params_len;
    /* per-kind state */
    orc_eq_three eq;
    orc_envelope env;
    orc_plotter plot;
    int plot_fired;
    float* plot_l; float* plot_r;
    const float* source;   /* host-fed */
    uint32_t source_ring;  /* 0: `source` is this tick's block; R > 0: `source` holds R ticks and tick t reads block t mod R */
    uint32_t dom_num, dom_den, in_dom_num, in_dom_den;   /* sample-rate domain (Resample changes it) */
    float* hist;           /* Fir / Resample carried input frames */
    int ran;               /* produced output this tick (back-edges read Disconnected) */
} onode;

struct orc_graph {
    onode* nodes; size_t n_nodes;
    uint32_t* order; size_t n_order;
    double sample_rate; size_t spt;
};


static size_t node_frames(const orc_graph* g, const onode* n) { return g->spt * n->dom_num / n->dom_den; }
static size_t node_len(const orc_graph* g, const onode* n, uint8_t lt) { return (lt == LT_MONO ? 1 : (lt == LT_STEREO ? 2 : 0)) * node_frames(g, n); }

static int node_ports(onode* n) {
    static const uint8_t none[1] = {0};
    (void)none;
    uint32_t ni = 0, no = 0; uint8_t it[8], ot[8];
    switch (n->kind) {
    case ORC_KIND_AMPLIFIER: ni = 2; it[0] = LT_STEREO; it[1] = LT_MONO; no = 1; ot[0] = LT_STEREO; break; /* amplifier.rs:21-25 */
    case ORC_KIND_ENVELOPE: case ORC_KIND_EQ_THREE: ni = 1; it[0] = LT_MONO; no = 1; ot[0] = LT_MONO; break;
    case ORC_KIND_FM_SINE: ni = 1; it[0] = LT_MONO; no = 1; ot[0] = LT_STEREO; break;                   /* fm_sine.rs:23-24 */
    case ORC_KIND_OSCILLATOR: ni = 0; no = 2; ot[0] = LT_MONO; ot[1] = LT_STEREO; break;                /* oscillator.rs:47-51 */
    case ORC_KIND_PLOTTER: ni = 1; it[0] = LT_STEREO; no = 0; break;                                    /* plotter.rs:23-24 */
    case ORC_KIND_STEREO_PANNER: ni = 2; it[0] = it[1] = LT_MONO; no = 1; ot[0] = LT_STEREO; break;
    case ORC_KIND_STEREO_SPLITTER: ni = 1; it[0] = LT_STEREO; no = 2; ot[0] = ot[1] = LT_MONO; break;
    case ORC_KIND_TRIGGER: ni = 0; no = 1; ot[0] = LT_MONO; break;
    case ORC_KIND_SOURCE_MONO: ni = 0; no = 1; ot[0] = LT_MONO; break;
    case ORC_KIND_SOURCE_STEREO: ni = 0; no = 1; ot[0] = LT_STEREO; break;
    case ORC_KIND_FIR: case ORC_KIND_RESAMPLE: ni = 1; it[0] = LT_STEREO; no = 1; ot[0] = LT_STEREO; break;
    case ORC_KIND_MIXER: {
        uint32_t nch = n->params_len / (uint32_t)sizeof(orc_mixer_channel_params);              /* mixer.rs:22-28 */
        n->n_in = nch; n->n_out = 2;
        n->in_type = (uint8_t*)malloc(nch ? nch : 1); memset(n->in_type, LT_STEREO, nch);
        n->out_type = (uint8_t*)malloc(2); n->out_type[0] = n->out_type[1] = LT_STEREO;
        return 0;
    }
    default: return -1; /* video modules are exercised through the orc_video_* entry points */
    }
    n->n_in = ni; n->n_out = no;
    n->in_type = (uint8_t*)malloc(ni ? ni : 1); memcpy(n->in_type, it, ni);
    n->out_type = (uint8_t*)malloc(no ? no : 1); memcpy(n->out_type, ot, no);
    return 0;
}

/* traverse(), src/engine.rs:439-457 */
static void traverse(orc_graph* g, uint32_t id, uint8_t* seen) {
    if (seen[id]) return;
    seen[id] = 1;
    onode* n = &g->nodes[id];
    for (uint32_t i = 0; i < n->n_in; i++)
        if (n->in_src_node[i] >= 0) traverse(g, (uint32_t)n->in_src_node[i], seen);
    g->order[g->n_order++] = id;
}

orc_graph* orc_graph_build(const orc_node* nodes, size_t n_nodes, const orc_edge* edges, size_t n_edges,
                           uint32_t sample_rate, uint32_t ticks_per_second) {
    orc_graph* g = (orc_graph*)calloc(1, sizeof *g);
    g->sample_rate = (double)sample_rate;
    g->spt = sample_rate / ticks_per_second; /* src/engine.rs:55 */
    g->n_nodes = n_nodes;
    g->nodes = (onode*)calloc(n_nodes ? n_nodes : 1, sizeof(onode));
    for (size_t i = 0; i < n_nodes; i++) {
        onode* n = &g->nodes[i];
        n->kind = nodes[i].kind;
        n->params_len = nodes[i].params_len;
        n->params = malloc(n->params_len ? n->params_len : 1);
        if (n->params_len) memcpy(n->params, nodes[i].params, n->params_len);
        if (node_ports(n) != 0) { orc_graph_destroy(g); return NULL; }
        n->in_src_node = (int64_t*)malloc(sizeof(int64_t) * (n->n_in ? n->n_in : 1));
        n->in_src_port = (uint32_t*)calloc(n->n_in ? n->n_in : 1, sizeof(uint32_t));
        for (uint32_t k = 0; k < n->n_in; k++) n->in_src_node[k] = -1;
        n->out_buf = (float**)calloc(n->n_out ? n->n_out : 1, sizeof(float*));
        n->dom_num = n->dom_den = n->in_dom_num = n->in_dom_den = 1;   /* buffers are sized after domain propagation */
        orc_eq_three_init(&n->eq, g->sample_rate);
        orc_envelope_init(&n->env);
        n->plot.count = 0;
        n->plot_l = (float*)calloc(g->spt, sizeof(float));
        n->plot_r = (float*)calloc(g->spt, sizeof(float));
    }
    for (size_t e = 0; e < n_edges; e++) {
        const orc_edge* ed = &edges[e];
        if (ed->src_node >= n_nodes || ed->dst_node >= n_nodes) { orc_graph_destroy(g); return NULL; }
        onode* s = &g->nodes[ed->src_node]; onode* d = &g->nodes[ed->dst_node];
        if (ed->src_port >= s->n_out || ed->dst_port >= d->n_in) { orc_graph_destroy(g); return NULL; }
        /* type-checked connect, src/engine/workspace.rs:97-114 */
        if (s->out_type[ed->src_port] != d->in_type[ed->dst_port]) { orc_graph_destroy(g); return NULL; }
        d->in_src_node[ed->dst_port] = ed->src_node;
        d->in_src_port[ed->dst_port] = ed->src_port;
    }
    /* terminal modules = modules whose outputs feed nothing (engine.rs:408-416); DFS from each in
     * ascending id order (the reference iterates a HashSet, i.e. unspecified order; any order
     * yields the same result on an acyclic graph). */
    g->order = (uint32_t*)malloc(sizeof(uint32_t) * (n_nodes ? n_nodes : 1));
    uint8_t* feeds = (uint8_t*)calloc(n_nodes ? n_nodes : 1, 1);
    uint8_t* seen = (uint8_t*)calloc(n_nodes ? n_nodes : 1, 1);
    for (size_t e = 0; e < n_edges; e++) feeds[edges[e].src_node] = 1;
    for (size_t i = 0; i < n_nodes; i++) if (!feeds[i]) traverse(g, (uint32_t)i, seen);
    free(feeds); free(seen);
    /* sample-rate domains (build-specified Resample nodes change them), then the port buffers */
    for (size_t oi = 0; oi < g->n_order; oi++) {
        onode* n = &g->nodes[g->order[oi]];
        for (uint32_t k = 0; k < n->n_in; k++) if (n->in_src_node[k] >= 0) {
            const onode* sn = &g->nodes[n->in_src_node[k]];
            n->in_dom_num = sn->dom_num; n->in_dom_den = sn->dom_den;
        }
        n->dom_num = n->in_dom_num; n->dom_den = n->in_dom_den;
        if (n->kind == ORC_KIND_RESAMPLE) {
            const uint32_t* h = (const uint32_t*)n->params;   /* up, down, taps_per_phase, pad */
            uint64_t a = (uint64_t)n->in_dom_num * h[0], b = (uint64_t)n->in_dom_den * h[1], x = a, y = b;
            while (y) { uint64_t t = x % y; x = y; y = t; }
            n->dom_num = (uint32_t)(a / x); n->dom_den = (uint32_t)(b / x);
            n->hist = (float*)calloc(2 * (size_t)h[2], sizeof(float));
        }
        if (n->kind == ORC_KIND_FIR) n->hist = (float*)calloc(2 * (size_t)((const uint32_t*)n->params)[0], sizeof(float));
    }
    for (size_t i = 0; i < n_nodes; i++) {
        onode* n = &g->nodes[i];
        for (uint32_t k = 0; k < n->n_out; k++) n->out_buf[k] = (float*)calloc(node_len(g, n, n->out_type[k]) + 1, sizeof(float));
    }
    return g;
}

void orc_graph_destroy(orc_graph* g) {
    if (!g) return;
    for (size_t i = 0; i < g->n_nodes; i++) {
        onode* n = &g->nodes[i];
        if (n->out_buf) for (uint32_t k = 0; k < n->n_out; k++) free(n->out_buf[k]);
        free(n->out_buf); free(n->in_type); free(n->out_type); free(n->in_src_node); free(n->in_src_port);
        free(n->params); free(n->plot_l); free(n->plot_r); free(n->hist);
    }
    free(g->nodes); free(g->order); free(g);
}

size_t orc_graph_samples_per_tick(const orc_graph* g) { return g->spt; }

int orc_graph_update_params(orc_graph* g, uint32_t node, const void* params, uint32_t params_len) {
    if (node >= g->n_nodes || g->nodes[node].params_len != params_len) return -1;
    if (params_len) memcpy(g->nodes[node].params, params, params_len);
    return 0;
}

int orc_graph_set_source(orc_graph* g, uint32_t node, const float* samples) {
    if (node >= g->n_nodes) return -1;
    onode* n = &g->nodes[node];
    if (n->kind != ORC_KIND_SOURCE_MONO && n->kind != ORC_KIND_SOURCE_STEREO) return -1;
    n->source = samples;
    n->source_ring = 0;
    return 0;
}

/* A source that replays a resident buffer of `ring_ticks` ticks: tick t reads block (t mod ring_ticks).  This is what a device-resident
 * synthetic source re-read by every submission looks like to the modules behind it (bench.py's sources; nothing in the reference: its
 * sources are live, src/source.rs). */
int orc_graph_set_source_ring(orc_graph* g, uint32_t node, const float* samples, uint32_t ring_ticks) {
    if (orc_graph_set_source(g, node, samples) != 0 || ring_ticks == 0) return -1;
    g->nodes[node].source_ring = ring_ticks;
    return 0;
}

/* resolve an input terminal to this tick's producer buffer, or NULL for Disconnected
 * (engine.rs:475-484: connections.get(..).and_then(|o| buffers.get(o)) -- a producer that has not
 * run yet this tick, i.e. a cycle's back-edge, also reads as Disconnected) */
static const float* in_buf(const orc_graph* g, const onode* n, uint32_t port) {
    if (n->in_src_node[port] < 0) return NULL;
    const onode* s = &g->nodes[n->in_src_node[port]];
    if (!s->ran) return NULL;
    return s->out_buf[n->in_src_port[port]];
}

int orc_graph_run_tick(orc_graph* g, uint64_t tick) {
    uint64_t t = tick * (uint64_t)g->spt; /* engine.rs:490 */
    for (size_t i = 0; i < g->n_nodes; i++) { g->nodes[i].ran = 0; g->nodes[i].plot_fired = 0; }
    for (size_t oi = 0; oi < g->n_order; oi++) {
        onode* n = &g->nodes[g->order[oi]];
        /* Output::from_line_type: fresh zero-filled buffers every tick (io.rs:71-77) */
        for (uint32_t k = 0; k < n->n_out; k++) memset(n->out_buf[k], 0, node_len(g, n, n->out_type[k]) * sizeof(float));
        const size_t spt = node_frames(g, n);   /* this node's frames per tick (its sample-rate domain) */
        switch (n->kind) {
        case ORC_KIND_AMPLIFIER:
            orc_amplifier_run((const orc_amplifier_params*)n->params, in_buf(g, n, 0),
                              n->in_src_node[1] >= 0 && g->nodes[n->in_src_node[1]].ran ? in_buf(g, n, 1) : NULL,
                              n->out_buf[0], 2 * spt);
            break;
        case ORC_KIND_ENVELOPE:
            orc_envelope_run(&n->env, (const orc_envelope_params*)n->params, g->sample_rate, t, in_buf(g, n, 0), n->out_buf[0], spt);
            break;
        case ORC_KIND_EQ_THREE:
            orc_eq_three_run(&n->eq, (const orc_eq_three_params*)n->params, in_buf(g, n, 0), n->out_buf[0], spt);
            break;
        case ORC_KIND_FM_SINE:
            orc_fm_sine_run((const orc_fm_sine_params*)n->params, g->sample_rate, t, in_buf(g, n, 0), n->out_buf[0], spt);
            break;
        case ORC_KIND_MIXER: {
            const float** ins = (const float**)malloc(sizeof(float*) * (n->n_in ? n->n_in : 1));
            for (uint32_t k = 0; k < n->n_in; k++) ins[k] = in_buf(g, n, k);
            orc_mixer_run((const orc_mixer_channel_params*)n->params, n->n_in, ins, n->out_buf[0], n->out_buf[1], 2 * spt);
            free(ins);
            break;
        }
        case ORC_KIND_OSCILLATOR:
            orc_oscillator_run((const orc_oscillator_params*)n->params, g->sample_rate, t, n->out_buf[0], n->out_buf[1], spt);
            break;
        case ORC_KIND_PLOTTER:
            n->plot_fired = orc_plotter_run(&n->plot, in_buf(g, n, 0), n->plot_l, n->plot_r, spt);
            break;
        case ORC_KIND_STEREO_PANNER:
            orc_stereo_panner_run(in_buf(g, n, 0), in_buf(g, n, 1), n->out_buf[0], spt);
            break;
        case ORC_KIND_STEREO_SPLITTER:
            orc_stereo_splitter_run(in_buf(g, n, 0), n->out_buf[0], n->out_buf[1], spt);
            break;
        case ORC_KIND_TRIGGER:
            orc_trigger_run((const orc_trigger_params*)n->params, n->out_buf[0], spt);
            break;
        case ORC_KIND_FIR: {
            const uint32_t* h = (const uint32_t*)n->params;
            orc_fir_run((const double*)((const char*)n->params + 8), h[0], n->hist, in_buf(g, n, 0), n->out_buf[0], spt);
            break;
        }
        case ORC_KIND_RESAMPLE: {
            const uint32_t* h = (const uint32_t*)n->params;
            const size_t in_frames = g->spt * n->in_dom_num / n->in_dom_den;
            orc_resample_run((const double*)((const char*)n->params + 16), h[0], h[1], h[2], n->hist,
                             tick * (uint64_t)in_frames, tick * (uint64_t)spt, in_buf(g, n, 0), in_frames, n->out_buf[0], spt);
            break;
        }
        case ORC_KIND_SOURCE_MONO:
            if (n->source) memcpy(n->out_buf[0], n->source + (n->source_ring ? (tick % n->source_ring) * spt : 0), spt * sizeof(float));
            break;
        case ORC_KIND_SOURCE_STEREO:
            if (n->source) memcpy(n->out_buf[0], n->source + (n->source_ring ? (tick % n->source_ring) * 2 * spt : 0), 2 * spt * sizeof(float));
            break;
        default: return -1;
        }
        n->ran = 1;
    }
    return 0;
}

int orc_graph_run_ticks(orc_graph* g, uint64_t first_tick, uint32_t n) {
    for (uint32_t k = 0; k < n; ++k) {
        const int rc = orc_graph_run_tick(g, first_tick + k);
        if (rc) return rc;
    }
    return 0;
}

const float* orc_graph_output(const orc_graph* g, uint32_t node, uint32_t port, size_t* len) {
    if (node >= g->n_nodes || port >= g->nodes[node].n_out) return NULL;
    if (len) *len = node_len(g, &g->nodes[node], g->nodes[node].out_type[port]);
    return g->nodes[node].out_buf[port];
}

int orc_graph_plotter_indication(const orc_graph* g, uint32_t node, float* left, float* right) {
    if (node >= g->n_nodes || g->nodes[node].kind != ORC_KIND_PLOTTER) return -1;
    const onode* n = &g->nodes[node];
    if (!n->plot_fired) return 0;
    memcpy(left, n->plot_l, g->spt * sizeof(float));
    memcpy(right, n->plot_r, g->spt * sizeof(float));
    return 1;
}

size_t orc_graph_run_order(const orc_graph* g, uint32_t* order, size_t cap) {
    for (size_t i = 0; i < g->n_order && i < cap; i++) order[i] = g->order[i];
    return g->n_order;
}
