/*
 * mixlab_gpu.h -- C ABI of the MI355X-native tick engine for haileys/mixlab's per-tick module graph.
 *
 * The reference has no C ABI: modules are Rust types behind `trait ModuleT`
 * (src/module/mod.rs:7-19) dispatched through `DynModuleHostT` (src/engine/module.rs:88-94) by
 * `Engine::run_tick` (src/engine.rs:400-510).  This header is the thin extern-"C" surface a Rust
 * `GpuModule: ModuleT` adapter (or a replacement for Engine::run_tick's inner loop) binds with an
 * `extern "C"` block -- see INTEGRATION.md for the Rust side.  Plain pointers and sizes only.
 *
 * Error convention follows the reference's own extern-"C" boundary (FFmpeg I/O callbacks,
 * codec/src/ffmpeg.rs:25-26, codec/src/ffmpeg/ioctx.rs:51-67,136-152): never unwind across the
 * boundary, return a negative int sentinel, stash the message, let the caller fetch it
 * (mx_last_error) and re-raise on its side.
 *
 * Threading: like Engine::run_tick (one engine thread, src/engine.rs:78-93) a graph/module handle
 * is not re-entrant; different handles may be used from different threads.
 */
#ifndef MIXLAB_GPU_H
#define MIXLAB_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MX_ABI_VERSION 4u   /* 2: mx_exchange_*, mx_monitor_tick.dropped, mx_monitor_params_ex, packed RGB pixel formats; 3: MX_FLAG_FP_CONTRACT;
                              * 4: mx_graph_read_output_window, per-pixel alpha (MX_PIXFMT_YUVA420P, mx_dframe_*_alpha; the A byte of packed RGBA honoured) */

/* ---- status codes (0 ok, <0 error; cf. MIXLAB_IOCTX_ERROR / MIXLAB_IOCTX_PANIC) ---- */
enum {
    MX_OK = 0,
    MX_ERR_INVALID = -1,  /* bad argument / id out of range */
    MX_ERR_TYPE = -2,     /* port line-type mismatch: the reference panics (src/engine/io.rs:40-41,49-50) or refuses the connection (src/engine/workspace.rs:97-114) */
    MX_ERR_DEVICE = -3,   /* HIP runtime error */
    MX_ERR_NOMEM = -4,
    MX_ERR_INTERNAL = -5, /* C++ exception caught at the boundary ("panic") */
    MX_ERR_FULL = -6      /* a bounded ingest queue is full: where the reference's sender blocks (sync_channel) or gets Err (ring push) */
};

/* ---- line types: protocol LineType, src/engine/io.rs:19-24 ---- */
typedef enum { MX_DISCONNECTED = 0, MX_MONO = 1, MX_STEREO = 2, MX_VIDEO = 3 } mx_line;

/* ---- module kinds: the DSP members of enumerate_modules! (src/module/mod.rs:27-49) ---- */
enum {
    MX_KIND_AMPLIFIER = 0,        /* src/module/amplifier.rs       in: Stereo, Mono(control)  out: Stereo */
    MX_KIND_ENVELOPE = 1,         /* src/module/envelope.rs        in: Mono(gate)             out: Mono */
    MX_KIND_EQ_THREE = 2,         /* src/module/eq_three.rs        in: Mono                   out: Mono */
    MX_KIND_FM_SINE = 3,          /* src/module/fm_sine.rs         in: Mono                   out: Stereo */
    MX_KIND_MIXER = 4,            /* src/module/mixer.rs           in: Stereo x N             out: Stereo Master, Stereo Cue */
    MX_KIND_OSCILLATOR = 5,       /* src/module/oscillator.rs      in: -                      out: Mono, Stereo */
    MX_KIND_PLOTTER = 6,          /* src/module/plotter.rs         in: Stereo                 out: - (indication) */
    MX_KIND_STEREO_PANNER = 7,    /* src/module/stereo_panner.rs   in: Mono L, Mono R         out: Stereo */
    MX_KIND_STEREO_SPLITTER = 8,  /* src/module/stereo_splitter.rs in: Stereo                 out: Mono L, Mono R */
    MX_KIND_TRIGGER = 9,          /* src/module/trigger.rs         in: -                      out: Mono */
    MX_KIND_VIDEO_MIXER = 10,     /* src/module/video_mixer.rs     in: Video x 4              out: Video Output, A, B */
    MX_KIND_SOURCE_MONO = 11,     /* host/device-fed port; stands in for the audio outputs of the I/O modules */
    MX_KIND_SOURCE_STEREO = 12,   /*   (StreamInput src/module/stream_input.rs:72-147, MediaSource media_source.rs:93-126) */
    MX_KIND_SOURCE_VIDEO = 13,    /* frame-fed port; stands in for MediaSource / StreamInput video (media_source.rs:93-126) */
    MX_KIND_VIDEO_TO_RGBA = 14,   /* BUILD-SPECIFIED sink (no reference module): YUV420P -> RGBA8 (+ Q12 3x4 matrix)  in: Video */
    MX_KIND_FIR = 15,             /* BUILD-SPECIFIED (BASELINE configs[2]; no reference module): K-tap FIR on a stereo stream  in: Stereo  out: Stereo */
    MX_KIND_RESAMPLE = 16,        /* BUILD-SPECIFIED rational polyphase resampler (44.1 -> 48 kHz = up 160 / down 147; the reference has only
                                     `TODO implement resampling`, src/icecast/mod.rs:94-97)  in: Stereo  out: Stereo at rate * up / down */
    MX_KIND_MONITOR = 17,         /* the hand-off of Monitor / StreamOutput (src/module/monitor.rs:113-139, stream_output.rs:154-158) and the first steps of
                                     their codec threads (monitor.rs:226-236; encode.rs:183-195 f32->i16, :287-295 DynamicScaler to the
                                     encoder's picture): keeps every tick's program frame, scaled, and serves the mix as i16
                                     in: Video, Stereo  out: --   params: mx_monitor_params */
    MX_KIND_COUNT = 18
};

/* protocol/src/lib.rs:233-241, bincode variant order */
enum { MX_WAVE_ON = 0, MX_WAVE_OFF = 1, MX_WAVE_SINE = 2, MX_WAVE_SQUARE = 3, MX_WAVE_TRIANGLE = 4, MX_WAVE_SAW = 5 };

/* ---- parameter structs: #[repr(C)] mirrors of protocol/src/lib.rs ---- */
typedef struct { double gain_db; double fader; uint8_t cue; uint8_t _pad[7]; } mx_mixer_channel_params; /* MixerChannelParams :342-347; MixerParams = array of these (params_len / sizeof) :329-332 */
typedef struct { double gain_lo_db, gain_mid_db, gain_hi_db; } mx_eq_three_params;                     /* EqThreeParams :285-290 (Decibel = f64 dB :455-456) */
typedef struct { double attack_ms, decay_ms, sustain_amplitude, release_ms; } mx_envelope_params;      /* EnvelopeParams :310-316 */
typedef struct { double amplitude, mod_depth; } mx_amplifier_params;                                   /* AmplifierParams :298-302 */
typedef struct { double freq; uint32_t waveform; uint32_t _pad; } mx_oscillator_params;                /* OscillatorParams :243-247 */
typedef struct { double freq_lo, freq_hi; } mx_fm_sine_params;                                         /* FmSineParams :292-296 */
typedef struct { uint32_t gate_open; } mx_trigger_params;                                              /* GateState :304-308 */
typedef struct { int32_t a, b; /* -1 = None */ double fader; } mx_video_mixer_params;                  /* VideoMixerParams :405-410 */
typedef struct { int32_t use_matrix; int32_t matrix_q12[12]; } mx_video_to_rgba_params;                /* build-specified, DESIGN.md "Colour" */
typedef struct { uint32_t width, height; } mx_monitor_params;   /* the encoder's picture: 560 x 350 (monitor.rs:21-22), 1120 x 700 (stream_output.rs:23-24); even */
/* The same with the reference's back-pressure (opt-in; params_len selects the form).  Monitor / StreamOutput hand every tick to their codec
 * thread with try_send on a channel of TWO and DROP the tick when it is full (monitor.rs:163-177, stream_output.rs:316-320).  queue_depth > 0:
 * the node holds at most that many ticks the consumer has not taken (mx_graph_monitor_consume); a tick that finds the queue full is dropped --
 * no picture is scaled or kept, mx_monitor_tick.dropped = 1 -- exactly what the codec thread would never see.  queue_depth = 0 (and the short
 * form): every tick of a submission is kept (one scaled frame per tick stays on the device until the next run: max_ticks_per_run x frame bytes). */
typedef struct { uint32_t width, height, queue_depth, _pad; } mx_monitor_params_ex;
/* Build-specified audio extras (DESIGN.md "FIR and resampler").  Both are variable-length blobs: the header below
 * followed by the f64 coefficients.  Arithmetic: f32 widened to f64, accumulated in f64 in ascending tap index with
 * separate multiply and add, rounded once to f32 -- the reference's own convention (mixer.rs:62, amplifier.rs:56). */
typedef struct { uint32_t n_taps; uint32_t _pad; /* double taps[n_taps] */ } mx_fir_params;               /* y[n] = sum_k taps[k] x[n-k] per channel */
typedef struct { uint32_t up, down, taps_per_phase, _pad; /* double taps[up][taps_per_phase] */ } mx_resample_params;
/*   output sample M (absolute): n = floor(M * down / up), phase = (M * down) mod up, y[M] = sum_k taps[phase][k] x[n-k].
 *   A node's output lives in the sample-rate domain (rate * up / down); modules whose arithmetic depends on the sample
 *   rate or on t (EqThree, Envelope, Oscillator, FmSine) are only accepted in the base domain. */

/* ---- graph description ---- */
typedef struct { uint32_t kind; uint32_t params_len; const void* params; } mx_node;
typedef struct { uint32_t src_node, src_port, dst_node, dst_port; } mx_edge;  /* workspace.connections: InputId -> OutputId */

/* EqThree arithmetic.  DEFAULT: bit-exact with the reference's sequential order (src/module/eq_three.rs:58-89) -- the
 * reference's own module test asserts exact equality with its golden file (eq_three.rs:150-167) and the default passes it.
 * MX_FLAG_EQ_FAST opts into the time-parallel chunked scan: NOT bit-exact -- every f32 output is within 1 ULP of the
 * reference order and about 1 sample in 20 000 differs (DESIGN.md "EqThree").  MX_FLAG_EQ_EXACT is the default's old
 * name, kept as a no-op; it wins if both are given. */
#define MX_FLAG_EQ_EXACT 1u
#define MX_FLAG_EQ_FAST 4u
/* MX_FLAG_FP_CONTRACT: the CONTRACTED order -- the reference's f64 expressions with every multiply fused into the add that consumes
 * it (what -ffp-contract=fast makes of the same source; Rust itself never contracts).  It applies to EqThree
 * (p += f * (in - p) as fma(f, in - p, p), pole 0's + VSA as fma(f, in - p, VSA), the band mix as fma(hi, g_hi, fma(mid, g_mid, lo * g_lo)),
 * eq_three.rs:76-88,117-124), Envelope (sustain + (1 - sustain) * decay as one fma, envelope.rs:46-47), Amplifier (depth() as one fma,
 * amplifier.rs:71-73) and the build-specified Fir / Resample (acc = fma(h[k], x, acc), ascending k).  NOT bit-exact with the
 * reference: every f32 output is within 1 ULP of the exact order's (26 instead of 36 f64 instructions per EqThree sample).  The
 * contracted order is as deterministic as the exact one -- speculation, proof and repair work on it unchanged -- and the oracle's
 * contract mode (orc_set_fp_contract) restates it, so it is tested bit for bit.  Not with MX_FLAG_EQ_FAST (the scan has its own
 * arithmetic): mx_graph_build fails with MX_ERR_INVALID. */
#define MX_FLAG_FP_CONTRACT 16u

#define MX_FLAG_OVERLAP_TAIL 8u /* throughput mode for batched runs: the Mixer groups at the END of the launch order (a bank, or a bank and the
                                  group / master buses above it) run on a second stream beside the NEXT run's earlier groups (an HBM-bound kernel beside a VALU-bound one); the ports it
                                  reads are double-buffered and alternate per run.  Results are unchanged bit for bit; mx_graph_sync, every
                                  read-back and mx_graph_run_ticks' own ordering cover both streams.  mx_graph_output_device_ptr of a port
                                  the tail READS names the buffer of the last run only (it alternates); the tail's own outputs do not move.
                                  The bank's launch of run k is HELD BACK until run k + 1 has queued its EqThree launch (it then starts once that launch's
                                  workgroups are placed) or until something joins the two streams: mx_graph_sync, any read-back, an exchange's submit, a run cut
                                  by a scheduled update, and mx_graph_tail_stream() itself.  A consumer on another stream of the tail's outputs therefore calls
                                  mx_graph_tail_stream() AFTER the run it wants and orders itself after the stream it returns.
                                  WITHOUT the flag the library takes this mode on its own for graphs with at least 64 EqThree instances built for submissions of
                                  at least 16 ticks (runs of fewer ticks stay on one stream), while the second buffers fit (MX_OVERLAP_AUTO_MAX_GB, default 32, and
                                  a quarter of the free device memory) and only while no HOST holds a raw pointer the mode would not keep fresh:
                                  mx_graph_output_device_ptr of one of the tail's outputs, or of a port the tail reads (double-buffered in this mode), ends the
                                  automatism for that graph for good -- everything outstanding completes, the last run's data is where the pointer says, and from
                                  then on the graph is a one-stream graph (stream-ordered consumers see one-stream ordering as before).  An mx_exchange over a bus
                                  does NOT end it: the exchange orders itself behind the bank on whichever stream it runs.  1024 strips x 2048 ticks: 5.3 -> 4.8 ms per run.
                                  MX_OVERLAP_AUTO=0 (environment) turns it off. */
#define MX_FLAG_NO_FUSE 2u   /* materialise every port.  By default the graph compiler folds EqThree -> StereoPanner(L = R)
                                [-> Amplifier [<- Envelope <- Trigger]] into the EQ kernel, a single-consumer Trigger into
                                its Envelope, and stores an L == R stereo result that only Mixers read as one float per
                                frame.  Folded ports are per-tick temporaries (src/engine.rs:461,504-506) and cannot be
                                read back.  Results on every remaining port are bit-identical either way. */

typedef struct {
    uint32_t sample_rate;        /* 0 => 44100 (src/engine.rs:53) */
    uint32_t ticks_per_second;   /* 0 => 60    (src/engine.rs:54) */
    uint32_t max_ticks_per_run;  /* 0 => 1; port buffers hold this many consecutive ticks */
    uint32_t flags;              /* MX_FLAG_* */
    int32_t device;              /* HIP device ordinal, -1 => current */
    int32_t _pad;
    void* stream;                /* hipStream_t to launch on; NULL => the graph creates its own */
} mx_graph_opts;

typedef struct mx_graph mx_graph;

/* Thread-local message for the last failing call on this thread (ioctx "stash then report"). */
const char* mx_last_error(void);
uint32_t mx_abi_version(void);
/* Number of visible HIP devices, or <0. */
int mx_device_count(void);

/* Freeze a topology (the state Engine::run_tick reads from Workspace, src/engine/workspace.rs:13-19)
 * and allocate every port buffer in one HBM slab.  Connections are type-checked like
 * Workspace::connect (workspace.rs:97-114): mismatch => MX_ERR_TYPE. */
int mx_graph_build(const mx_node* nodes, size_t n_nodes, const mx_edge* edges, size_t n_edges,
                   const mx_graph_opts* opts, mx_graph** out);
void mx_graph_destroy(mx_graph* g);

int mx_graph_samples_per_tick(const mx_graph* g, size_t* spt);                 /* SAMPLES_PER_TICK, src/engine.rs:55 */
int mx_graph_run_order(const mx_graph* g, uint32_t* order, size_t cap, size_t* n); /* DFS order of src/engine.rs:421-457 */

/* The hipStream_t the graph launches on (the one given in mx_graph_opts, or its own): for ordering other device work against a run. */
int mx_graph_stream(mx_graph* g, void** stream);
/* MX_FLAG_OVERLAP_TAIL (or the automatic mode): the stream the last launch group runs on (NULL when the mode is off or the graph has no such group).
 * Releases a launch of that group the library was holding back for the next run (see the flag): call it after the run whose result is wanted. */
int mx_graph_tail_stream(mx_graph* g, void** stream);

/* ModuleT::update (src/module/mod.rs:16): replace one node's params between ticks. */
int mx_graph_update_params(mx_graph* g, uint32_t node, const void* params, size_t params_len);

/* Engine::client_update BETWEEN two ticks of one submission (src/engine.rs:192-214 drains the command queue after every tick;
 * :277-398 applies ModuleT::update): `params` replace `node`'s at the boundary before tick `tick_in_run` (0-based) of the NEXT
 * mx_graph_run_ticks, which must cover that tick (else that run fails with MX_ERR_INVALID and drops its schedule).  Updates for
 * the same node and tick apply in submission order; after the run the node holds the last one.  A Trigger's updates cost nothing
 * (one gate bit per tick read by the kernels that folded it in); an update of any other kind cuts the run into separately
 * launched spans at its tick.  The _batch form queues many at once (one foreign call per run). */
typedef struct { uint32_t node; uint32_t tick_in_run; const void* params; size_t params_len; } mx_param_event;
int mx_graph_schedule_params(mx_graph* g, uint32_t node, uint32_t tick_in_run, const void* params, size_t params_len);
int mx_graph_schedule_params_batch(mx_graph* g, const mx_param_event* events, size_t n_events);

/* Counters of the default EqThree path on long streams (speculative time-parallel form proven bit-exact chunk by chunk,
 * DESIGN.md "EqThree"), accumulated since the graph was built: stream chunks run, and chunks whose start state the
 * verification pass found different from the sequential order's and re-ran (exactly-constant input after a signal).
 * Synchronises the graph's stream. */
int mx_graph_eq_spec_stats(mx_graph* g, uint64_t* chunks_run, uint64_t* chunks_repaired);
/* The same counters with what the proof / repair pass did about them: out[0] chunks run, [1] chunks not proven by their recorded start
 * state, [2] of those settled by comparing outputs under a constant input (O(1)), [3] walk steps of 16 samples (both trajectories re-run
 * side by side, outputs rewritten), [4] fill steps of 16 samples (constant input, standing state, outputs differ), [5] rounds of islands
 * walked side by side, [6] in-order walks after an island that ended apart from the speculative run, [7] streams finished by the NaN fill. */
int mx_graph_eq_repair_stats(mx_graph* g, uint64_t out[8]);
/* DEBUG / profiling aid: the chunk records of the first EqThree launch group's last speculative launch (device memory owned by the graph, valid until the next run;
 * synchronises).  144 bytes per chunk: start[8], end[8] (f64 poles), min / max input bits, and in the padding -- written by the tiled kernel -- lane 0 of a wave: the
 * shader clock (low 32 bits) when the wave entered, lane 1: HW_ID, lane 2: XCC_ID, every lane: the clock when it left.  tools/wave_times.py reads it. */
int mx_graph_debug_eq_records(mx_graph* g, void** device_records, size_t* bytes);
/* DEBUG: how the Mixer banks of the second-stream mode (MX_FLAG_OVERLAP_TAIL / automatic) went out since the graph was built: behind the gate that the next run's EqThree
 * launch opens, or at once (a join released them, or the next run had no such launch).  Tests use it to know which path they exercised. */
int mx_graph_debug_tail_releases(mx_graph* g, uint64_t* gated, uint64_t* at_once);

/* Feed a SOURCE_* node: n_ticks consecutive tick buffers (SPT mono / 2*SPT interleaved stereo f32). */
int mx_graph_write_source(mx_graph* g, uint32_t node, const float* host_samples, size_t n_ticks);
/* Or bind an external device buffer holding max_ticks_per_run tick buffers (read-only, caller-owned). */
int mx_graph_bind_source_device(mx_graph* g, uint32_t node, const void* device_ptr);

/* n_ticks consecutive Engine::run_tick calls (src/engine.rs:400-510) in one submission:
 * tick k of the run uses t = (first_tick + k) * SPT (src/engine.rs:490).  Asynchronous on the
 * graph's stream.  n_ticks <= max_ticks_per_run. */
int mx_graph_run_ticks(mx_graph* g, uint64_t first_tick, uint32_t n_ticks);
int mx_graph_sync(mx_graph* g);

/* Copy an output port's buffers of the last run to the host (synchronises the stream). */
int mx_graph_read_output(mx_graph* g, uint32_t node, uint32_t port, float* host_samples, size_t n_ticks);
/* The same for ticks [first_tick_in_run, first_tick_in_run + n_ticks) of the last run only: a consumer that wants the tail of a long
 * submission (or a checker that samples it) does not pay PCIe for the rest. */
int mx_graph_read_output_window(mx_graph* g, uint32_t node, uint32_t port, float* host_samples, size_t first_tick_in_run, size_t n_ticks);
/* The data formats either side of the path, converted on the device so PCIe carries 2 bytes per sample:
 *   sinks (Monitor / StreamOutput, src/video/encode.rs:183-195): clamp to [-1, 1], * 32767.0, `as i16` (saturating, truncating);
 *   ingest (StreamInput, src/module/stream_input.rs:167-173):    sample as f32 / 32768.0. */
int mx_graph_read_output_i16(mx_graph* g, uint32_t node, uint32_t port, int16_t* host_samples, size_t n_ticks);
int mx_graph_write_source_i16(mx_graph* g, uint32_t node, const int16_t* host_samples, size_t n_ticks);

/* Device pointer + per-tick length (floats) of an output port (for zero-copy consumers / RCCL). */
int mx_graph_output_device_ptr(mx_graph* g, uint32_t node, uint32_t port, void** device_ptr, size_t* floats_per_tick);

/* Plotter indication (src/module/plotter.rs:37-56) for tick `tick_in_run` of the last run:
 * *fired = 1 and SPT floats in each of left/right when it fired (every 6th call, input connected). */
int mx_graph_read_plotter(mx_graph* g, uint32_t node, uint32_t tick_in_run, float* left, float* right, int* fired);

/* Per-kind device time of the last profiled run (the PerformanceInfo analogue,
 * src/engine/timing.rs:86-94): run once with hipEvents around every launch group. */
int mx_graph_profile_run(mx_graph* g, uint64_t first_tick, uint32_t n_ticks, float* ms_by_kind /* MX_KIND_COUNT */, float* ms_total);
/* Same, accumulated over many asynchronous runs: while enabled every mx_graph_run_ticks records a
 * hipEvent before/after each launch group on the graph's stream (no synchronisation);
 * collect synchronises and returns the sums (ms) and the number of runs they cover. */
int mx_graph_profile_enable(mx_graph* g, int on);
int mx_graph_profile_collect(mx_graph* g, float* ms_by_kind /* MX_KIND_COUNT */, float* ms_total, uint32_t* n_runs);

/* The reference's performance panel (PerformanceInfo, protocol/src/lib.rs:32-59, filled by EngineStat::report,
 * src/engine/timing.rs:46-60) from the most recent profiled run (mx_graph_profile_run, or profile_enable + collect):
 *   realtime       the tick finished inside its budget (timing.rs:33)
 *   lag            0 none, 1 Recent (< 5 s ago), 2 Active (< 100 ms ago): a tick ran over its budget (util.rs:47-60)
 *   tick_budget_us 1e6 / ticks_per_second (timing.rs:9)
 *   module_us[i]   PerformanceAccount::Module(i), "last" metric, per tick: the time of the launch that served the module,
 *                  split evenly over the modules (and folded modules) that launch served
 *   engine_us      PerformanceAccount::Engine = tick time - sum of module accounts (timing.rs:43) */
typedef struct mx_performance_info {
    int32_t realtime; int32_t lag; uint32_t tick_rate; uint32_t n_modules;
    uint64_t tick_budget_us; uint64_t engine_us;
} mx_performance_info;
int mx_graph_performance_info(mx_graph* g, mx_performance_info* info, uint64_t* module_us /* [cap >= n nodes] or NULL */, size_t cap);

/* Topology edit (Engine::client_update, src/engine.rs:277-398): modules persist while connections are added and removed.
 * Build the edited graph with mx_graph_build, then let it take over the state of every module that survives:
 * old_node_of_new[i] = index in `old_graph` of the module that is node i of `new_graph`, or -1 for a new module
 * (EqThree poles and delay line, Envelope state, FIR / resampler history, Plotter count, VideoMixer stored frames and
 * scalers, pending video sources).  Kinds must match (MX_ERR_TYPE).  VideoMixers are MOVED: destroy `old_graph`
 * afterwards, do not run it. */
int mx_graph_adopt_state(mx_graph* new_graph, mx_graph* old_graph, const int32_t* old_node_of_new, size_t n);

/* Ingest re-blocking (StreamInput::run_tick, src/module/stream_input.rs:92-124): decoded audio arrives as i16 frames of
 * any length; every tick takes exactly 2 * SPT interleaved samples from the queue, a partly consumed frame stays queued,
 * and what the queue cannot fill is zeroed.  mx_pcm_ring_feed re-blocks n_ticks ticks into a SOURCE_STEREO node
 * (H2D as i16, sample / 32768 on the device, stream_input.rs:167-173); *zero_filled = samples it had to zero. */
typedef struct mx_pcm_ring mx_pcm_ring;
int mx_pcm_ring_create(mx_pcm_ring** out);
void mx_pcm_ring_destroy(mx_pcm_ring* r);
int mx_pcm_ring_push_i16(mx_pcm_ring* r, const int16_t* samples, size_t n_samples);
int mx_pcm_ring_queued(const mx_pcm_ring* r, size_t* n_samples);
int mx_pcm_ring_feed(mx_pcm_ring* r, mx_graph* g, uint32_t node, uint32_t n_ticks, size_t* zero_filled);

/* ---------------------------------------------------------------------------------------------- */
/* pixel path: device-resident yuv420p frames, VideoMixer, scaler, colour                          */
/* ---------------------------------------------------------------------------------------------- */

/* Host view of a frame: planar yuv420p, 8 bit (always, src/module/video_mixer.rs:282-283). */
typedef struct {
    uint32_t width, height;          /* luma size */
    uint8_t* data[3];                /* AVFrame.data   (codec/src/ffmpeg/frame.rs:188-197) */
    int32_t stride[3];               /* AVFrame.linesize */
    int64_t dur_num, dur_den;        /* video::Frame.duration_hint (src/video.rs:8-14) */
    int64_t off_num, off_den;        /* VideoFrame.tick_offset (src/engine/io.rs:12-17) */
} mx_frame;

/* Pixel formats a device frame can hold (codec/src/ffmpeg/pixfmt.rs:8-36 wraps AVPixelFormat; planar 8-bit YUV here).  Everything the
 * VideoMixer PRODUCES is yuv420p (src/module/video_mixer.rs:282-283); its INPUTS carry their own format in their picture settings and
 * the DynamicScaler's context converts implicitly (codec/src/ffmpeg/scale.rs:16-39, src/video/encode.rs:342-352: the settings compare
 * unequal when only the format differs).  Here that conversion is the build-specified scaler applied per plane: each chroma plane is
 * resampled from ITS size to the output's chroma size (DESIGN.md "Scaler" -- parity unpinned, like the scaler itself). */
typedef enum { MX_PIXFMT_YUV420P = 0, MX_PIXFMT_YUV422P = 1, MX_PIXFMT_YUV444P = 2,
               MX_PIXFMT_NV12 = 3 /* semi-planar 4:2:0, what hardware decoders deliver: plane 1 = interleaved U,V rows of `width` bytes, no plane 2
                                     (mx_frame.data[2] is ignored, mx_dframe_planes reports it NULL) */,
               /* packed RGB, one plane (data[0]; data[1], data[2] NULL): scaler INPUTS only (a screen capture, an image file).  BUILD-SPECIFIED: the
                * frame stands for the yuv444p frame of its per-pixel BT.709 limited-range conversion (DESIGN.md "Pixel formats"), which is then
                * resampled like any 4:4:4 input -- libswscale's own RGB path is unknown here: parity unpinned, like the scaler */
               MX_PIXFMT_RGB24 = 4 /* R, G, B bytes */, MX_PIXFMT_BGRA = 5 /* B, G, R, A bytes; A = the pixel's coverage, see MX_PIXFMT_YUVA420P */,
               /* the other planar 8-bit YUV layouts an AVPixelFormat descriptor can carry (pixfmt.rs:97-111: log2_chroma_w / log2_chroma_h of 0, 1 or 2): scaler
                * inputs like yuv422p / yuv444p -- every plane resampled from its own size; width / height multiples of the subsampling */
               MX_PIXFMT_YUV410P = 6 /* chroma 1/4 x 1/4 */, MX_PIXFMT_YUV411P = 7 /* chroma 1/4 x 1 */, MX_PIXFMT_YUV440P = 8 /* chroma 1 x 1/2 */,
               MX_PIXFMT_GRAY8 = 9 /* one plane of luma: stands for the yuv444p frame with U = V = 0x80 (scaler input only, like packed RGB) */,
               /* YUV deeper than 8 bits, as decoders of 10- / 12-bit streams deliver it (pixfmt.rs:107-111: bits per component from the descriptor): samples are 16-bit
                * little-endian words, mx_frame strides in BYTES as ever.  Scaler INPUTS only.  BUILD-SPECIFIED: the frame stands for the 8-bit frame of the same layout
                * whose samples are min(255, (v + 2^(b-9)) >> (b - 8)), v the b-bit value -- rounded to nearest, ties up: (v + 2) >> 2 for ten bits -- which is then
                * resampled like any 8-bit input (libswscale keeps the extra bits through its filters and dithers on the way out: unknown here, parity unpinned like the scaler) */
               MX_PIXFMT_YUV420P10 = 10, MX_PIXFMT_YUV422P10 = 11, MX_PIXFMT_YUV444P10 = 12 /* three planes, the value in the LOW ten bits of a word (the upper six ignored) */,
               MX_PIXFMT_P010 = 13 /* semi-planar 4:2:0 like nv12: luma plane + one plane of interleaved U,V words, the value in the HIGH ten bits of a word (the lower six ignored) */,
               MX_PIXFMT_YUV420P12 = 14, MX_PIXFMT_YUV422P12 = 15, MX_PIXFMT_YUV444P12 = 16 /* the value in the low twelve bits (the upper four ignored) */,
               MX_PIXFMT_YUV420P16 = 17, MX_PIXFMT_YUV422P16 = 18, MX_PIXFMT_YUV444P16 = 19 /* all sixteen bits */,
               MX_PIXFMT_P016 = 20 /* semi-planar 4:2:0, all sixteen bits (p012 is this layout with the low four bits zero: the same rounding applies) */,
               /* packed 4:2:2, one plane of 2 bytes per pixel (what capture devices deliver): scaler INPUTS only; the frame stands for the yuv422p frame with the same
                * samples (a byte shuffle, nothing to specify), which is then resampled like any yuv422p input; width even */
               MX_PIXFMT_YUYV422 = 21 /* Y0 U Y1 V */, MX_PIXFMT_UYVY422 = 22 /* U Y0 V Y1 */,
               /* the other byte orders of packed 8-bit RGB: the same build-specified conversion as rgb24 / bgra */
               MX_PIXFMT_BGR24 = 23 /* B, G, R */, MX_PIXFMT_RGBA = 24 /* R, G, B, A */, MX_PIXFMT_ARGB = 25 /* A, R, G, B */, MX_PIXFMT_ABGR = 26 /* A, B, G, R */,
               /* PER-PIXEL ALPHA (BUILD-SPECIFIED: the reference's only "alpha" is the VideoMixer's global fader, video_mixer.rs:168).  A layer may carry a COVERAGE
                * plane -- yuva420p: yuv420p plus a fourth plane of width x height bytes, 255 = opaque (mx_frame holds the three YUV planes; the fourth travels through
                * mx_dframe_upload_alpha / _download_alpha) -- and the A byte of a four-byte packed RGB input IS that plane (straight, not premultiplied).  The scaler
                * resamples it like luma (letterbox bars opaque).  A VideoMixer step then weighs its layers per sample, in fade_line's own u16 arithmetic
                * (video_mixer.rs:211-235), aA / aB = the coverage of A / B there (255 where a layer carries none; chroma samples use the co-sited luma sample's):
                *     wa = (aA * fade) / 255;   wb = (aB * (255 - wa)) / 255;   out = (A * (255 - wb) + B * wb) / 255
                * Opaque layers give wa = fade, wb = 255 - fade: the reference's cross-fade bit for bit.  The composite itself is opaque yuv420p (DESIGN.md "Per-pixel alpha"). */
               MX_PIXFMT_YUVA420P = 27 } mx_pixfmt;

/* Device frame: the AvFrame<Video> stand-in.  Reference-counted like an AVFrame (clone =
 * av_frame_clone, codec/src/ffmpeg/frame.rs:351-361): create returns one reference; the VideoMixer
 * keeps inputs past the call by retaining them.  Frames are immutable once handed to a mixer.
 * All stateless pixel calls take a hipStream_t (`stream`, NULL = the library's default video
 * stream) and are asynchronous on it unless stated. */
typedef struct mx_dframe mx_dframe;
int mx_dframe_create(uint32_t width, uint32_t height, void* stream, mx_dframe** out);  /* yuv420p; blank: Y=0 U=V=0x80 (frame.rs:76-138) */
/* any mx_pixfmt: width / height must be multiples of the chroma subsampling (pixfmt.rs:97-111); mx_frame planes follow the device frame's format */
int mx_dframe_create_fmt(uint32_t width, uint32_t height, mx_pixfmt fmt, void* stream, mx_dframe** out);
int mx_dframe_format(const mx_dframe* f, mx_pixfmt* fmt);
int mx_dframe_retain(mx_dframe* f);
void mx_dframe_release(mx_dframe* f);
int mx_dframe_upload(mx_dframe* f, const mx_frame* host, void* stream);       /* visible area; synchronous */
int mx_dframe_download(const mx_dframe* f, mx_frame* host, void* stream);     /* visible area; synchronous */
int mx_dframe_planes(const mx_dframe* f, uint32_t* width, uint32_t* height, void* device_data[3], int32_t stride[3]);
/* The coverage plane of a yuva420p frame (MX_PIXFMT_YUVA420P): width x height bytes, `stride` bytes per host row.  Synchronous.  MX_ERR_INVALID for a frame
 * without one.  alpha_plane: its device address (NULL: none) for producers that write it on the device. */
int mx_dframe_upload_alpha(mx_dframe* f, const uint8_t* host_alpha, int32_t stride, void* stream);
int mx_dframe_download_alpha(const mx_dframe* f, uint8_t* host_alpha, int32_t stride, void* stream);
int mx_dframe_alpha_plane(const mx_dframe* f, void** device_alpha, int32_t* stride);

/* AvFrame::blank (frame.rs:76-138) */
int mx_video_blank(mx_dframe* f, void* stream);
/* The compose step of VideoMixer::run_tick (src/module/video_mixer.rs:150-239): out = cross-fade of
 * a and b by `(fader * 255.0) as u8`; a/b NULL reads the blank plane (video_mixer.rs:180-188). */
int mx_video_crossfade(mx_dframe* out, const mx_dframe* a, const mx_dframe* b, double fader, void* stream);
/* DynamicScaler::scale (src/video/encode.rs:338-397) of `in` into `out`'s size: identity copy when
 * equal, else blank + aspect-preserving letterboxed bicubic.  The bicubic arithmetic is
 * BUILD-SPECIFIED (libswscale is outside the reference tree): see DESIGN.md "Scaler". */
int mx_video_scale(const mx_dframe* in, mx_dframe* out, void* stream);
/* DynamicScaler (src/video/encode.rs:338-397): keeps its context (tap tables, blank letterboxed output frame) while the
 * input settings stay the same -- what Monitor / StreamOutput call once per tick to shrink the program frame before it
 * leaves the device (monitor.rs:21-22, stream_output.rs:23-24, encode.rs:287-295).  *out holds one reference
 * (mx_dframe_release); it is the input itself when the sizes are equal (encode.rs:342-345), else the scaler's own frame,
 * overwritten by the next call.  Asynchronous on the scaler's stream. */
typedef struct mx_video_scaler mx_video_scaler;
int mx_video_scaler_create(uint32_t out_width, uint32_t out_height, void* stream, mx_video_scaler** out);
int mx_video_scaler_scale(mx_video_scaler* sc, const mx_dframe* in, mx_dframe** out);
void mx_video_scaler_destroy(mx_video_scaler* sc);
int mx_video_scale_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                            uint32_t* scaled_w, uint32_t* scaled_h, uint32_t* letterbox_x, uint32_t* letterbox_y);   /* encode.rs:354-374 */
/* Row band of DynamicScaler::scale for a picture composited over several GPUs by rows (mixlab_amd/shard.py):
 * luma rows [row0, row0 + out_band height) of the (full_w x full_h) letterboxed result, computed from a SLICE that holds luma rows
 * [src_row0, src_row0 + slice height) of a source in_full_h rows high.  Tap indices clamp against the full source plane, exactly as
 * the unsharded scale does, so stitched bands are the unsharded frame bit for bit; a slice that lacks a row the band's vertical
 * taps reach is MX_ERR_INVALID.  All row counts even (whole chroma rows); out_band is as wide as the full picture.  Synchronous. */
int mx_video_scale_band(const mx_dframe* in_slice, uint32_t in_full_h, uint32_t src_row0, mx_dframe* out_band,
                        uint32_t full_w, uint32_t full_h, uint32_t row0, void* stream);
/* The scaler's tap table for one axis (DESIGN.md "Scaler"): *n_taps coefficients (Q14, summing to 16384) per output sample and the
 * index of the first source sample each set applies to.  first: [dst], coef: [dst][*n_taps] with room for mx_video_scaler_tap_count()
 * entries per sample.  Host-only (no device needed): what the tests pin against tests/golden/bicubic_taps_*.json. */
uint32_t mx_video_scaler_tap_count(uint32_t src, uint32_t dst);
int mx_video_scaler_taps(uint32_t src, uint32_t dst, int32_t* first, int32_t* coef, uint32_t* n_taps);
/* BUILD-SPECIFIED (no reference counterpart): BT.709 limited-range YUV420P -> RGBA8 (+ optional Q12 3x4 matrix). */
int mx_video_to_rgba(const mx_dframe* in, void* device_rgba, int32_t rgba_stride, const int32_t* matrix_q12 /* 12 or NULL */, void* stream);
int mx_video_sync(void* stream);
/* A caller-owned hipStream_t that graphs / scalers / mixers launched pictures on is about to be destroyed: release what the library keeps per (device, stream) for its
 * batched video launches (page-locked descriptor staging, device copies, events, an upload stream).  The library frees this itself for streams it created; for a caller's
 * stream it cannot know when the stream dies -- and a later stream may be given the same handle value.  Call it after the last launch on the stream has finished
 * (it synchronises the upload stream, not `stream`). */
int mx_stream_retired(void* stream);

/* VideoMixer (src/module/video_mixer.rs): 4 video inputs, program + A + B outputs. */
typedef struct { mx_dframe* frame; /* NULL = no frame this tick */ int64_t dur_num, dur_den, off_num, off_den; } mx_video_input;
typedef struct mx_video_mixer mx_video_mixer;
int mx_video_mixer_create(const mx_video_mixer_params* params, uint32_t sample_rate /* 0 => 44100 */, void* stream, mx_video_mixer** out);
int mx_video_mixer_update(mx_video_mixer* m, const mx_video_mixer_params* params);
/* One VideoMixer::run_tick (video_mixer.rs:70-250).  Returned frames carry one reference for the
 * caller (mx_dframe_release when done); NULL = None.  The program frame has duration 1/60 and
 * tick_offset 0 (video_mixer.rs:241-247).  The pixels are produced asynchronously on the mixer's stream (mx_video_mixer_create; NULL = the
 * library's default video stream, the one every entry point with a NULL stream uses): consume them there, or mx_video_mixer_sync first. */
int mx_video_mixer_run_tick(mx_video_mixer* m, uint64_t t, const mx_video_input inputs[4],
                            mx_dframe** out_program, mx_dframe** out_a, mx_dframe** out_b);
int mx_video_mixer_sync(mx_video_mixer* m);
void mx_video_mixer_destroy(mx_video_mixer* m);

/* Video nodes inside a graph (MX_KIND_VIDEO_MIXER / SOURCE_VIDEO / VIDEO_TO_RGBA): every tick of
 * mx_graph_run_ticks runs the video sub-graph in run order, all launches on the graph's stream.
 * set_video_source: the frame the source emits -- on the next tick only (repeat = 0, like one
 * decoded frame arriving, media_source.rs:93-126) or as a new frame on every tick (repeat = 1,
 * synthetic 60 fps input).  frame NULL clears it.  The graph retains the frame. */
int mx_graph_set_video_source(mx_graph* g, uint32_t node, mx_dframe* frame, int64_t dur_num, int64_t dur_den,
                              int64_t off_num, int64_t off_den, int repeat);
/* A source that delivers a NEW frame on every tick, cycling through `n` frames (a decoder feeding the graph: MediaSource emits
 * at most one VideoFrame per tick, media_source.rs:93-126): tick k of the graph's life emits frames[k mod n] with the given
 * duration hint and offset.  The graph retains the frames.  n = 0 clears the source. */
int mx_graph_set_video_source_ring(mx_graph* g, uint32_t node, mx_dframe* const* frames, size_t n, int64_t dur_num, int64_t dur_den,
                                   int64_t off_num, int64_t off_den);
/* Row-band sharding of one composited picture over ranks (mixlab_amd/shard.py): a rank's graph is the cascade at (full_w x
 * band_rows).  Layers of the picture's own size are fed as their band rows; a SMALLER layer is fed as the halo slice its band needs
 * (luma rows [src_row0, src_row0 + slice_rows) of an in_w x in_full_h yuv420p source, shard.band_source_rows) and this call makes
 * the source node deliver, on every tick it has a frame, luma rows [row0, row0 + band_rows) of that layer's letterboxed scale into
 * (full_w x full_h) -- what DynamicScaler::scale (encode.rs:338-397) gives the unsharded VideoMixer, cut to the band; asynchronous,
 * on the graph's stream.  band_rows = 0 removes the transform. */
int mx_graph_set_video_source_band(mx_graph* g, uint32_t node, uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows,
                                   uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows);
/* One frame due on one tick of a SOURCE_VIDEO node: tick `tick` (absolute, as in mx_graph_run_ticks' first_tick + k) emits `frame` with the
 * given duration hint and tick offset; ticks without an entry emit None.  Entries are queued in ascending tick order, one per tick
 * (MX_ERR_INVALID otherwise), and while any is queued they take the place of mx_graph_set_video_source[_ring].  What
 * mx_media_source_feed / mx_stream_input_feed use; the graph retains the frame until its tick has run. */
int mx_graph_queue_video_source(mx_graph* g, uint32_t node, uint64_t tick, mx_dframe* frame, int64_t dur_num, int64_t dur_den,
                                int64_t off_num, int64_t off_den);

/* ---------------------------------------------------------------------------------------------- */
/* timed ingest: decoded frames with rational timestamps enter the engine  */
/* ---------------------------------------------------------------------------------------------- */

/* MediaSource::run_tick (src/module/media_source.rs:93-126): the decode thread sends (frame, pts, duration) over a channel of TWO
 * (sync_channel(2), :140); every tick takes at most one frame off the channel, moves its pts by the epoch -- the engine time of the
 * tick the FIRST frame was received on (:104-107) -- and emits the oldest buffered frame once its pts lies before the end of the tick
 * (:113-121), with tick_offset = pts - start of tick (negative for a late frame).  Timestamps are exact rationals in seconds
 * (util/src/time.rs:10-75); sample_rate / ticks_per_second 0 => 44100 / 60 (src/engine.rs:53-54). */
typedef struct mx_media_source mx_media_source;
int mx_media_source_create(uint32_t sample_rate, uint32_t ticks_per_second, mx_media_source** out);
void mx_media_source_destroy(mx_media_source* m);
/* MediaSourceEvent::SetMedia (:85-91): present = 1 installs a fresh OpenMedia (empty channel, no epoch, empty buffer, :140-147), 0 = None. */
int mx_media_source_set_media(mx_media_source* m, int present);
/* the decode thread's tx.send (:271): MX_ERR_FULL while two frames wait (the reference blocks), MX_ERR_INVALID without media (the
 * reference's thread ends, :272-276).  The source retains the frame.  Safe from another thread than the run_tick caller. */
int mx_media_source_send(mx_media_source* m, mx_dframe* frame, int64_t pts_num, int64_t pts_den, int64_t dur_num, int64_t dur_den);
/* One run_tick at engine time t (samples).  out->frame NULL = None; else it carries one reference for the caller. */
int mx_media_source_run_tick(mx_media_source* m, uint64_t t, mx_video_input* out);
/* n_ticks run_tick calls for ticks first_tick .. first_tick + n_ticks - 1, their frames queued on SOURCE_VIDEO `node`
 * (mx_graph_queue_video_source) for the mx_graph_run_ticks(first_tick, n_ticks) that follows. */
int mx_media_source_feed(mx_media_source* m, mx_graph* g, uint32_t node, uint64_t first_tick, uint32_t n_ticks);

/* StreamInput::run_tick (src/module/stream_input.rs:72-147) with both of its rings (src/source.rs:97-98, 65536 frames each): audio
 * frames (source id, source time, i16 samples of any length) are re-blocked to ticks exactly as mx_pcm_ring does; a frame from a source
 * id other than the one the tick started with re-bases the epoch = engine time - its source time (:100-106); the next video frame is due
 * at tick_offset = source time + epoch - engine time (< 0 or no source yet => 0, :127-133) and is held back while that lies beyond the
 * tick (:135-138). */
typedef struct mx_stream_input mx_stream_input;
int mx_stream_input_create(uint32_t sample_rate, mx_stream_input** out);
void mx_stream_input_destroy(mx_stream_input* s);
/* StreamInput::update with another mountpoint (:57-70): listening = 1 replaces both rings by empty ones, 0 leaves none (every write is
 * MX_ERR_FULL, every tick reads silence); the held frames and the source timing stay. */
int mx_stream_input_listen(mx_stream_input* s, int listening);
/* SourceSend::write_audio / write_video (src/source.rs:158-190); source_id != 0 (NonZeroUsize, :35).  MX_ERR_FULL = Err(()).
 * Safe from another thread than the run_tick caller. */
int mx_stream_input_write_audio(mx_stream_input* s, uint64_t source_id, int64_t ts_num, int64_t ts_den, const int16_t* interleaved, size_t n_samples);
int mx_stream_input_write_video(mx_stream_input* s, uint64_t source_id, int64_t ts_num, int64_t ts_den, mx_dframe* frame, int64_t dur_num, int64_t dur_den);
/* One run_tick at engine time t: audio_out[n_out] (n_out = 2 * SPT) receives the tick's interleaved i16 samples -- convert_sample's
 * input (:167-173) -- zero where the queue ran dry (*zero_filled samples); video_out as in mx_media_source_run_tick.  Host only. */
int mx_stream_input_run_tick(mx_stream_input* s, uint64_t t, int16_t* audio_out, size_t n_out, mx_video_input* video_out, size_t* zero_filled);
/* n_ticks run_tick calls: audio into SOURCE_STEREO `audio_node` (H2D as i16 from page-locked memory, / 32768 on the device), frames
 * queued on SOURCE_VIDEO `video_node` (UINT32_MAX = the video output is not connected). */
int mx_stream_input_feed(mx_stream_input* s, mx_graph* g, uint32_t audio_node, uint32_t video_node, uint64_t first_tick, uint32_t n_ticks,
                         size_t* zero_filled);

/* H2D staging ring for decoded frames (the reference moves frames between threads through rings, src/source.rs:97-98): `upload` packs the
 * host planes into the next page-locked slot, laid out like the device frame, and sends it as ONE asynchronous copy on the stager's own
 * stream; *out is a device frame (one reference for the caller) from a pool -- a frame nobody holds any more is written again, after
 * what the last fenced stream has queued.  A consumer stream must be fenced before it reads frames uploaded since the last fence.
 * A slot is reused only once its copy has completed (upload blocks until then): `slots` bounds the frames in flight. */
typedef struct mx_frame_stager mx_frame_stager;
int mx_frame_stager_create(uint32_t slots, mx_frame_stager** out);
void mx_frame_stager_destroy(mx_frame_stager* st);
int mx_frame_stager_upload(mx_frame_stager* st, const mx_frame* host, mx_pixfmt fmt, mx_dframe** out);
/* The copy-free form, for a decoder that can be given its picture buffers (AVCodecContext.get_buffer2): `acquire` hands out the plane
 * pointers and strides of a page-locked slot (rows 64-byte aligned, padding already blank), the decoder writes the picture there, `commit`
 * sends the slot as one asynchronous copy and returns the device frame.  Several slots may be held at once (reference pictures);
 * MX_ERR_FULL when all are.  A slot's memory is the caller's from acquire until commit. */
int mx_frame_stager_acquire(mx_frame_stager* st, uint32_t width, uint32_t height, mx_pixfmt fmt, mx_frame* host, uint32_t* ticket);
int mx_frame_stager_commit(mx_frame_stager* st, uint32_t ticket, mx_dframe** out);
int mx_frame_stager_fence(mx_frame_stager* st, void* stream);
int mx_frame_stager_fence_graph(mx_frame_stager* st, mx_graph* g);
int mx_frame_stager_sync(mx_frame_stager* st);

/* Output port of a video node after the last tick: one reference for the caller, NULL = None. */
int mx_graph_video_output(mx_graph* g, uint32_t node, uint32_t port, mx_dframe** out);
/* RGBA8 device buffer a VIDEO_TO_RGBA node wrote on the last tick (width/height 0 = no frame).  Written asynchronously on the graph's
 * stream: read it there, or after mx_graph_sync. */
int mx_graph_rgba_output(mx_graph* g, uint32_t node, void** device_rgba, int32_t* stride, uint32_t* width, uint32_t* height);

/* MX_KIND_MONITOR after a run.  Tick `tick_in_run` of the last mx_graph_run_ticks as the codec thread would see it:
 * ts = the tick's timestamp relative to the node's epoch -- the first tick it ever ran (monitor.rs:121-123); when the Video input carried
 * a frame: *frame = that picture through the node's DynamicScaler (the input itself when it already has the encoder's size,
 * encode.rs:342-345; one reference for the caller, still on the device: download it or hand it to a device encoder), frame_ts = ts +
 * tick_offset (monitor.rs:229), dur = its duration hint; else video_present = 0 and *frame = NULL.  The reference DROPS a tick when its
 * codec thread lags (try_send on a channel of two, monitor.rs:163-177): a node built with mx_monitor_params_ex.queue_depth > 0 does the same
 * (`dropped`, below: nothing is scaled for such a tick, until mx_graph_monitor_consume frees slots); the short parameter form keeps every tick.
 * Device memory inside a batched run (MX_VIDEO_BATCH = K ticks per launch, default 16): every scaled layer's Scaler holds a ring of 2K output
 * frames (1080p yuv420p: 32 x 3.1 MB = 100 MB per scaled layer) and every VIDEO_TO_RGBA node K RGBA buffers (16 x 8.3 MB = 133 MB). */
typedef struct { int32_t video_present; int64_t ts_num, ts_den, frame_ts_num, frame_ts_den, dur_num, dur_den;
                 int32_t dropped; /* mx_monitor_params_ex.queue_depth > 0: the queue was full, the codec thread never gets this tick (nor its audio) */ int32_t _pad; } mx_monitor_tick;
int mx_graph_read_monitor_tick(mx_graph* g, uint32_t node, uint32_t tick_in_run, mx_monitor_tick* info, mx_dframe** frame);
/* queue_depth > 0: the consumer (the codec thread's rx.recv, monitor.rs:226) has taken n_ticks ticks off the node's queue: that many slots are free
 * for the ticks of the next submissions.  More than are queued empties the queue. */
int mx_graph_monitor_consume(mx_graph* g, uint32_t node, uint32_t n_ticks);
/* All kept pictures of ticks [first_tick, first_tick + n_ticks) of the last run in ONE read-back: one gather launch on the device, one
 * D2H copy.  frames[n_ticks * frame_bytes]: slot k holds tick first_tick + k's picture in the layout mx_graph_monitor_layout reports (rows
 * 64-byte aligned: an AVFrame.linesize an encoder takes as is); present[k] = 0 leaves slot k untouched.  A page-locked `frames` avoids
 * the runtime's staging copy. */
typedef struct { uint32_t width, height; size_t frame_bytes; size_t plane_offset[3]; int32_t stride[3]; } mx_monitor_layout;
int mx_graph_monitor_layout(mx_graph* g, uint32_t node, mx_monitor_layout* out);
int mx_graph_read_monitor_video(mx_graph* g, uint32_t node, uint32_t first_tick, uint32_t n_ticks, uint8_t* frames, uint8_t* present);
/* The mix the node received over the first n_ticks ticks of the last run, as the encoder's PCM: clamp to [-1, 1], * 32767, truncate
 * (encode.rs:183-195), converted on the device: audio[n_ticks * 2 * SPT].  A Disconnected input reads zeros (io.rs:56-57). */
int mx_graph_read_monitor_audio_i16(mx_graph* g, uint32_t node, int16_t* audio, uint32_t n_ticks);

/* ---------------------------------------------------------------------------------------------- */
/* multi-GPU: the bus exchange of a strip-sharded job                       */
/* ---------------------------------------------------------------------------------------------- */

/* One process per GPU, the reference's single engine thread (src/engine.rs:78-96) in each.  Strips are partitioned contiguously
 * over the ranks; the sharded job is DEFINED as the reference-expressible graph  N x Mixer(strips / N) -> Mixer(N, unity gains):
 * every sample of the whole Master / Cue bus is the f32 sum of the N partial buses in rank order 0 .. N-1 (Mixer::run_tick,
 * src/module/mixer.rs:57-68, applied to the partials), and every rank ends with it.  The exchange runs on its own stream,
 * pipelined against the next step's compute: two steps may be in flight.
 *   MX_EXCHANGE_ALLGATHER  one ncclAllGather of the [master | cue] partials, then the rank-ordered sum: (N - 1) bus lengths received
 *   MX_EXCHANGE_SLICES     ordered reduce-scatter + all-gather: the step's ticks are cut into N time slices, rank j receives slice j
 *                          of every partial (grouped ncclSend / ncclRecv), sums it in rank order, an all-gather distributes the
 *                          finished slices: 2 (N - 1) / N bus lengths received.  Bit-identical to ALLGATHER.  n_ticks % N == 0.
 *   MX_EXCHANGE_ALLREDUCE  ncclAllReduce(sum): NOT the summation order of any graph the reference can express (non-parity mode)
 *   MX_EXCHANGE_AUTO       SLICES when N >= 4 and the ticks divide, else ALLGATHER */
enum { MX_EXCHANGE_AUTO = 0, MX_EXCHANGE_ALLGATHER = 1, MX_EXCHANGE_SLICES = 2, MX_EXCHANGE_ALLREDUCE = 3 };
#define MX_EXCHANGE_ID_BYTES 128   /* sizeof(ncclUniqueId) */
typedef struct mx_exchange mx_exchange;
typedef struct mx_loopback_group mx_loopback_group;

/* ncclGetUniqueId: rank 0 makes the job's id and hands the 128 bytes to the other ranks by any means it has (the reference's
 * hosts already talk over sockets); every rank passes them to mx_exchange_create, which is collective (ncclCommInitRank).
 * librccl is bound on first use (dlopen of librccl.so.1): a host without it still loads this library and gets MX_ERR_DEVICE here.
 * MX_RCCL_LIB (environment) names another library to bind in its place -- the test suite's stand-in for RCCL between processes that
 * share one GPU (tests/helpers/fake_rccl.c), through which the RCCL transport's own code runs with 2 and 4 real peers on a one-GPU box. */
int mx_exchange_unique_id(void* id_out /* MX_EXCHANGE_ID_BYTES */);
/* In-process transport for `world` exchanges of ONE process (virtual ranks on one GPU, or one thread driving several GPUs):
 * device-to-device copies stand in for the collectives; buffers, combine and pipelining are the RCCL path's.  Every member
 * submits step k before any member submits step k + 1.  Destroy the group after its exchanges. */
int mx_loopback_group_create(uint32_t world, mx_loopback_group** out);
void mx_loopback_group_destroy(mx_loopback_group* grp);

/* Exchange of Mixer `mixer_node`'s two output buses of `g` over steps of `n_ticks` ticks (<= max_ticks_per_run).  Exactly one of
 * nccl_unique_id / loopback is given.  `g` must outlive the exchange.  Over a graph whose Mixer bank runs on the second stream (MX_FLAG_OVERLAP_TAIL or the automatic mode,
 * which an exchange does NOT end) an RCCL exchange's pack and collectives of step k go out behind the bank when the graph releases it -- with run k + 1, or when the step's
 * result is asked for (wait / result / read_result / elapsed_ms / sync) -- and run k + 2 starts after them; the loopback transport joins the streams at the submit. */
int mx_exchange_create(mx_graph* g, uint32_t mixer_node, uint32_t n_ticks, uint32_t rank, uint32_t world,
                       const void* nccl_unique_id, mx_loopback_group* loopback, uint32_t mode, mx_exchange** out);
void mx_exchange_destroy(mx_exchange* x);
/* After mx_graph_run_ticks of step `step` (any increasing numbering): pack the partial buses on the graph's stream and queue the
 * exchange + combine behind them on the exchange's stream.  Asynchronous.  Steps `step` and `step - 1` stay readable. */
int mx_exchange_submit(mx_exchange* x, uint64_t step);
/* Make `stream` (NULL = the graph's stream) wait for step `step`'s combined bus. */
int mx_exchange_wait(mx_exchange* x, uint64_t step, void* stream);
/* Device pointers of the combined Master / Cue of step `step` (n_ticks tick buffers each, interleaved stereo): valid on a stream
 * that waited (mx_exchange_wait), until step + 2 is submitted -- or, when the consumer reads them on a stream of its own, until
 * the point it marks with mx_exchange_release (the exchange orders its next write of those buffers after that point). */
int mx_exchange_result(mx_exchange* x, uint64_t step, void** master_device, void** cue_device, size_t* floats_per_bus);
int mx_exchange_release(mx_exchange* x, uint64_t step, void* stream);
/* The same to host memory, synchronously (either pointer may be NULL). */
int mx_exchange_read_result(mx_exchange* x, uint64_t step, float* master, float* cue);
/* Device time of step `step`'s exchange on its own stream (collectives + combine), ms; waits for it. */
int mx_exchange_elapsed_ms(mx_exchange* x, uint64_t step, float* ms);
int mx_exchange_sync(mx_exchange* x);
typedef struct { uint32_t mode, rank, world, loopback; uint64_t floats_per_bus, bytes_received_per_step; } mx_exchange_info;
int mx_exchange_get_info(const mx_exchange* x, mx_exchange_info* out);

/* plain device memory for consumers of mx_video_to_rgba (tests, bench) */
int mx_device_alloc(size_t bytes, void** device_ptr);
void mx_device_free(void* device_ptr);
int mx_device_download(void* host, const void* device_ptr, size_t bytes, void* stream);   /* synchronous */
/* page-locked host memory for the buffers that cross PCIe every submission (mx_graph_read_monitor_video's `frames`, PCM, sources):
 * copies from / to it are one DMA, without the runtime's staging through its own pinned bounce buffers */
int mx_host_alloc(size_t bytes, void** host_ptr);
void mx_host_free(void* host_ptr);

/* ---- per-module compatibility path: one ModuleT instance, host pointers in and out ---- */

typedef struct { mx_line kind; const float* samples; size_t len; const mx_frame* video; } mx_input;                 /* InputRef, io.rs:19-24 */
typedef struct { mx_line kind; float* samples; size_t len; mx_frame* video; int video_present; } mx_output;         /* OutputRef, io.rs:96-100 */

typedef struct mx_module mx_module;

/* ModuleT::create (src/module/mod.rs:12) at the reference's compile-time 44100 Hz / 60 ticks. */
int mx_module_create(uint32_t kind, const void* params, size_t params_len, mx_module** out);
/* Same with explicit rates/flags (only sample_rate, ticks_per_second, flags, device are read). */
int mx_module_create_ex(uint32_t kind, const void* params, size_t params_len, const mx_graph_opts* opts, mx_module** out);
/* ModuleT::update */
int mx_module_update(mx_module* m, const void* params, size_t params_len);
/* ModuleT::run_tick (src/module/mod.rs:17): inputs/outputs are host buffers owned by the caller
 * for the duration of the call; outputs are fully overwritten.  A MX_DISCONNECTED input reads the
 * zero buffer.  Any input length is accepted (the reference's one test feeds 355 285 samples in
 * one call, src/module/eq_three.rs:150-167) as long as all ports agree.  Plotter: *indication_len is IN/OUT --
 * on entry the capacity of `indication` in bytes, on return the bytes written (0 = None): left[frames] then
 * right[frames] f32 for the call's buffer length; a capacity below 2 * frames floats is MX_ERR_INVALID.
 * EqThree always runs in the reference's exact order on this path unless MX_FLAG_EQ_FAST was given at creation. */
int mx_module_run_tick(mx_module* m, uint64_t t, const mx_input* inputs, size_t n_inputs,
                       mx_output* outputs, size_t n_outputs, void* indication, size_t* indication_len);
void mx_module_destroy(mx_module* m);

#ifdef __cplusplus
}
#endif
#endif /* MIXLAB_GPU_H */
