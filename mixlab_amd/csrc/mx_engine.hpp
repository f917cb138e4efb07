// mx_engine.hpp -- device-resident graph executor (internal C++ API behind include/mixlab_gpu.h).
//
// Mirrors what Engine::run_tick (reference src/engine.rs:400-510) does per tick -- topological
// order, fresh outputs, Disconnected => zeros, t = tick * SPT -- but freezes the topology once,
// keeps every port buffer resident in one HBM slab, batches all instances of a module kind at one
// dependency level into one launch, and runs n_ticks ticks per submission.
#pragma once
#include <algorithm>
#include <functional>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <deque>
#include <vector>

#include "mx_common.hpp"
#include "mx_kernels.hpp"
#include "mx_video.hpp"

namespace mx {

struct PortRef { int32_t node = -1; uint32_t port = 0; };

struct Node {
    uint32_t kind = 0;
    std::vector<uint8_t> params;
    std::vector<uint8_t> in_type, out_type;   // mx_line per terminal (ModuleT::inputs()/outputs())
    std::vector<PortRef> in_src;              // workspace.connections (back-edges already cut)
    std::vector<PortRef> in_src_orig;         // in_src before set_input_enabled() toggles
    std::vector<size_t> out_off;              // float offset of each output port in the slab
    std::vector<size_t> out_off2;             // MX_FLAG_OVERLAP_TAIL: second buffer of a port the tail group reads (SIZE_MAX: none)
    const float* bound = nullptr;             // SOURCE_*: caller-bound device buffer
    int level = 0;
    uint32_t dom_num = 1, dom_den = 1;        // sample-rate domain of the OUTPUT ports relative to the graph's rate (Resample changes it)
    uint32_t in_dom_num = 1, in_dom_den = 1;  // ... of the input ports
    uint32_t slot = 0;                        // index inside its (level, kind) group
    int group = -1;
    int sub_key = 0;                          // EQ_THREE: 1 + the epilogue mode the graph compiler gave the node (launch groups are of one mode); 0 otherwise
    // parameter updates queued for ticks inside the next run (Engine::client_update between two ticks, src/engine.rs:192-214)
    struct SchedEv { uint32_t tick; std::vector<uint8_t> params; };
    std::vector<SchedEv> sched;
    std::vector<std::pair<uint32_t, uint32_t>> gate_sched;   // Trigger: (tick, gate_open) -- a bench step queues ~70 000 of these, no allocation each
    // plotter
    uint64_t plot_count = 0;
    std::vector<uint8_t> plot_fired;          // per call of the last run
    std::vector<int32_t> plot_slot;           // staging slot per call (-1 = not fired)
    // graph-compiler fusion (Graph::plan_fusion)
    bool elided = false;                      // never launched: its work is folded into `owner`'s kernel
    int32_t owner = -1;                       // node whose descriptors carry this node's params
    int32_t fuse_pan = -1, fuse_amp = -1;     // EQ_THREE: StereoPanner / Amplifier folded into the epilogue
    int32_t fuse_trigger = -1;                // ENVELOPE: Trigger folded in as a constant gate
    int32_t fuse_env = -1;                    // EQ_THREE: constant-gate Envelope evaluated inline as the fused Amplifier's control
    std::vector<uint8_t> out_elided;          // per output port: buffer not materialised
    std::vector<uint8_t> out_dup;             // per output port: stereo with L == R stored as ONE float per frame
    // video nodes (run tick by tick inside Graph::run)
    struct VOut { FrameRef frame; Rational dur, off; };
    std::unique_ptr<VideoMixer> vmixer;       // VIDEO_MIXER
    bool vlazy = false;                       // VIDEO_MIXER: program output handed over as an unevaluated cross-fade chain (graph compiler)
    std::vector<VOut> vout;                   // this tick's video outputs (empty FrameRef = None)
    FrameRef vsrc; Rational vsrc_dur, vsrc_off; bool vsrc_repeat = false, vsrc_pending = false;   // SOURCE_VIDEO
    std::vector<FrameRef> vsrc_ring; size_t vsrc_ring_pos = 0;                                       // SOURCE_VIDEO: a new frame every tick, cycling
    struct VSched { uint64_t tick; FrameRef frame; Rational dur, off; };
    std::deque<VSched> vsrc_sched;                                                                   // SOURCE_VIDEO: frames due on given ticks (MediaSource / StreamInput pacing), oldest first
    std::shared_ptr<BandScaler> vband; std::vector<FrameRef> vband_pool;                             // SOURCE_VIDEO: frames are halo slices, delivered as this rank's row band of the scaled picture
    std::vector<DevBuf> rgba; uint32_t rgba_cur = 0, rgba_w = 0, rgba_h = 0; int32_t rgba_stride = 0;   // VIDEO_TO_RGBA: video_batch_ticks() buffers, written in turn (that many ticks' chains may share a launch); rgba_cur = the last tick's
    struct PendingRgba { ChainRgbaArgs args; std::shared_ptr<LazyChain> keep; };
    std::vector<PendingRgba> rgba_pending;     // VIDEO_TO_RGBA: chains of the last ticks, not launched yet, oldest first (run_video_tick)
    uint32_t rgba_calls = 0;                   // sink calls that queued a chain in this run
    // MONITOR: what the sink kept of every tick of the last run
    struct MonTick { bool present = false, dropped = false; FrameRef frame; Rational ts, frame_ts, dur; };
    uint32_t mon_depth = 0, mon_queued = 0;    // mx_monitor_params_ex.queue_depth (0 = keep every tick) and the ticks the consumer has not taken yet
    std::vector<MonTick> mon_ticks;
    bool mon_has_epoch = false; Rational mon_epoch;
    std::shared_ptr<Scaler> mon_scaler;
    DevBuf mon_pack;                           // the packed read-back's device staging
};

struct Group {
    int level = 0;
    uint32_t kind = 0;
    int sub_key = 0;
    uint32_t dom_num = 1, dom_den = 1, in_dom_num = 1, in_dom_den = 1;   // every node of a group shares one rate domain
    uint32_t max_taps = 0;   // Fir / Resample: most taps; Mixer: most channels
    uint32_t rs_common_taps = 0, rs_common_down = 0;   // ... and their taps per phase / decimation factor, likewise
    uint32_t rs_common_up = 0;   // Resample: the nodes' interpolation factor when they all have the same (the staged kernel's table stride becomes a constant)
    uint32_t rs_tab_doubles = 0, rs_win_frames = 0;   // Resample: LDS plan of the staged kernel (largest table / input window of the group)
    std::vector<uint32_t> nodes;
    DevBuf desc;     // kind-specific descriptor array
    DevBuf desc_alt, extra_alt;   // MX_FLAG_OVERLAP_TAIL: the same for the other parity of the double-buffered ports (extra_alt: Mixer only)
    DevBuf state;    // EnvState[] / EqState[]
    DevBuf extra;    // Mixer: MixChan arrays
    DevBuf state2;   // EqThree: EnvState[] of Envelopes folded into the epilogue
    int dup_mode = 0; // Mixer: 0 no input stored mono-dup, 1 all, 2 mixed
    // Trigger gates per tick of the run (GateBits rows, one per node of the group): Trigger groups, Envelope groups with a folded
    // Trigger, EqThree groups with a folded Envelope.  Re-uploaded only when a gate, a schedule or the run length changed.
    bool has_gates = false;
    DevBuf gates; uint32_t gate_words = 0; uint64_t gates_version = ~0ull; uint32_t gates_calls = 0;
    DevBuf tick_desc;   // EqThree: EnvTickDesc[] of the folded Envelopes
    DevBuf env_ticks;   // EqThree: EnvTick[n][n_calls] of the current launch
    DevBuf spec;        // EqThree: chunk records of the speculative exact kernel; Envelope: marker bitmaps and per-segment states of a long stream (mx_k_envelope.hip)
    int eq_mode = -1;   // EqThree: the one epilogue every instance has (eq_epilogue_mode), or -1 when they differ
};

class Graph {
public:
    Graph(const mx_node* nodes, size_t n_nodes, const mx_edge* edges, size_t n_edges, const mx_graph_opts& opts,
          size_t cap_frames_override = 0);
    ~Graph();

    size_t spt() const { return spt_; }
    double sample_rate() const { return sample_rate_; }
    size_t cap_frames() const { return cap_frames_; }
    const std::vector<uint32_t>& run_order() const { return order_; }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    uint32_t ticks_per_second() const { return tps_; }
    hipStream_t tail_stream() { if (tail_gi_ >= 0) flush_deferred_tail(false); return tail_gi_ >= 0 ? tail_stream_ : nullptr; }   // MX_FLAG_OVERLAP_TAIL; asking for it releases a held tail launch: what the caller orders after the stream then includes the last run's
    // debug: the chunk records of the first EqThree group's last speculative launch (device pointer, bytes; nullptr when there is none)
    void* debug_eq_records(size_t* bytes) const;
    void join_tail() { if (tail_gi_ >= 0) wait_tail(-1); }   // the graph's stream waits for a Mixer bank still running on the tail stream (consumers that read the buses on stream())
    size_t n_nodes() const { return nodes_.size(); }
    bool eq_exact() const { return (flags_ & MX_FLAG_EQ_EXACT) || !(flags_ & MX_FLAG_EQ_FAST); }   // the default is the reference's order
    bool fp_contract() const { return (flags_ & MX_FLAG_FP_CONTRACT) != 0; }                       // the contracted order (mixlab_gpu.h)
    const Node& node(uint32_t i) const { return nodes_.at(i); }

    void update_params(uint32_t node, const void* params, size_t len);
    // ModuleT::update at the boundary before tick `tick_in_run` of the NEXT run (client_update between ticks, src/engine.rs:192-214)
    void schedule_params(uint32_t node, uint32_t tick_in_run, const void* params, size_t len);
    void check_schedule(uint32_t node, const void* params, size_t len) const;   // schedule_params' validation alone
    void drop_schedules();                                                      // forget every queued update (a run that failed must not leave them for the next)
    // counters of the speculative exact EqThree kernel since the graph was built: [0] chunks run, [1] chunks the repair pass had to re-run
    void eq_spec_stats(uint64_t out[8]);   // see k_eq_three_repair
    void write_source(uint32_t node, const float* host, size_t frames);
    void bind_source(uint32_t node, const void* dev);
    // n_calls ModuleT::run_tick calls of frames_per_call mono samples each, back to back
    void run(uint64_t t0, size_t frames_per_call, uint32_t n_calls, float* ms_by_kind = nullptr, float* ms_total = nullptr);
    void sync();
    // accumulate per-group hipEvent timings across runs without synchronising (bench: kernel time over the timed region)
    void profile_enable(bool on);
    uint32_t profile_collect(float* ms_by_kind, float* ms_total);   // syncs; returns number of runs collected
    void read_output(uint32_t node, uint32_t port, float* host, size_t frames, size_t first_frame = 0);   // frames [first_frame, first_frame + frames) of the last run
    void read_output_i16(uint32_t node, uint32_t port, int16_t* host, size_t frames);   // sink hand-off format
    void write_source_i16(uint32_t node, const int16_t* host, size_t frames);            // ingest format
    float* output_ptr(uint32_t node, uint32_t port, size_t* floats_per_tick, bool stream_ordered_consumer = true /* false: a caller inside the library that orders itself
                      after the tail stream (mx_exchange): the automatic second-stream mode stays on */);
    // A Mixer bank of the last run that is being held back for the next run's EqThree launch (flush_deferred_tail)?  `hook` then runs ONCE, right after that launch has been
    // queued, with the stream it was queued on -- what mx_exchange uses to pack the buses behind the bank instead of joining the streams.
    bool tail_held() const { return deferred_.pending; }
    void tail_releases(uint64_t* gated, uint64_t* at_once) const { if (gated) *gated = n_gated_; if (at_once) *at_once = n_at_once_; }
    void set_tail_hook(std::function<void(hipStream_t)> hook) { tail_hook_ = std::move(hook); }
    // The NEXT run's first launch waits for `ev` (once).  What mx_exchange asks for its collectives and its combine of step k, which went out behind the bank that run k + 1
    // released: they normally end inside run k + 1's EqThree launch; when the bank outlasts that launch they would run into run k + 2's EqThree workgroups being placed --
    // thousands of small workgroups around which the dispatcher places those unevenly (every other launch 6.1 instead of 4.7 ms, rocprofv3 --kernel-trace).
    void wait_before_next_run(hipEvent_t ev) { head_waits_.push_back(ev); }
    void forget_wait_before_next_run(hipEvent_t ev) { head_waits_.erase(std::remove(head_waits_.begin(), head_waits_.end(), ev), head_waits_.end()); }   // (its owner is going away; other owners' waits stay)
    int read_plotter(uint32_t node, uint32_t call, float* left, float* right);
    void ensure_capacity(size_t frames);   // module compat path: grow the slab (state is kept)
    // topology edit (client_update, src/engine.rs:277-398): modules persist while the connection set changes.
    // Takes over the carried state of every surviving module: old_of_new[i] = node of `old` that is node i here, or -1.
    void adopt_state(Graph& old, const int32_t* old_of_new, size_t n);
    // per-module time of the last profiled run in the reference's PerformanceInfo shape (src/engine/timing.rs:46-60)
    struct Perf { bool realtime; int lag; uint32_t tick_rate; uint64_t tick_budget_us, engine_us; };
    Perf performance_info(uint64_t* module_us, size_t cap);
    // module compat path: an InputRef may be Disconnected on one call and connected on the next
    void set_input_enabled(uint32_t node, uint32_t port, bool enabled);
    // video nodes
    void set_video_source(uint32_t node, DFrame* frame, Rational dur, Rational off, bool repeat);
    void set_video_source_band(uint32_t node, uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows, uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows);
    void set_video_source_ring(uint32_t node, DFrame* const* frames, size_t n, Rational dur, Rational off);
    void queue_video_source(uint32_t node, uint64_t tick, DFrame* frame, Rational dur, Rational off);
    // what a feed checks BEFORE it takes frames out of a pacing state machine (a rejected feed must not lose media):
    void check_video_queue(uint32_t node, uint64_t first_tick) const;   // queue_video_source(node, first_tick, ...) would be accepted
    void check_source_write(uint32_t node, size_t frames) const;        // write_source[_i16](node, ..., frames) would be accepted
    const Node::MonTick& monitor_tick(uint32_t node, uint32_t tick_in_run);
    void monitor_consume(uint32_t node, uint32_t n_ticks);
    void read_monitor_audio_i16(uint32_t node, int16_t* host, uint32_t n_ticks);
    struct MonitorLayout { uint32_t width, height; size_t frame_bytes, plane_offset[3]; uint32_t stride[3]; };
    MonitorLayout monitor_layout(uint32_t node);
    void read_monitor_video(uint32_t node, uint32_t first_tick, uint32_t n_ticks, uint8_t* frames, uint8_t* present);
    FrameRef video_output(uint32_t node, uint32_t port);
    void rgba_output(uint32_t node, void** dev, int32_t* stride, uint32_t* w, uint32_t* h);

private:
    struct StateLoc { void* p; size_t bytes; };
    std::vector<StateLoc> state_locs(uint32_t node) const;
    void plan_fusion();
    void layout_slab();
    void build_descriptors();
    void upload_group(Group& g);          // descriptors of one group (both parities under MX_FLAG_OVERLAP_TAIL)
    void upload_group_one(Group& g);
    void run_video_tick(uint64_t t);
    void launch_pending_rgba(Node& n, size_t count, bool with_queued_scales);   // the `count` oldest pending chains of a sink
    // one launch sequence over ticks [call_off, call_off + n_calls) of the current run
    void run_span(uint64_t t0, size_t fpc, uint32_t call_off, uint32_t n_calls, uint32_t run_calls);
    void apply_params(uint32_t node, const void* params, size_t len);   // update_params without the synchronisation
    void refresh_gates(Group& g, uint32_t run_calls);
    uint32_t trigger_of_row(const Group& g, uint32_t row) const;        // node id of the Trigger behind row `row` of a gated group, or ~0u
    void stage_upload(void* dst, const void* src, size_t bytes);         // H2D on the graph's stream through page-locked staging
    const float* in_ptr(const Node& n, uint32_t port, bool null_if_disconnected) const;
    float* out_ptr(const Node& n, uint32_t port) const;

    std::vector<Node> nodes_;
    std::vector<uint32_t> order_;
    std::vector<Group> groups_;   // sorted by (level, kind)
    uint32_t flags_ = 0;
    bool has_video_ = false;
    uint32_t tps_ = 60;
    double sample_rate_ = 44100.0;
    size_t spt_ = 735;
    size_t cap_frames_ = 735;
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    DevBuf slab_;
    // MX_FLAG_OVERLAP_TAIL (see mixlab_gpu.h): the last launch group on a second stream, beside the next run's earlier groups
    int tail_gi_ = -1;                    // index of the FIRST tail group in groups_ (every group from it on is a Mixer group), -1 = mode off
    bool eq_mode_warned_ = false;         // the grouping / descriptor mode mismatch was reported (build_descriptors)
    bool tail_auto_ = false;              // the mode was chosen by the library (short submissions), not asked for with MX_FLAG_OVERLAP_TAIL
    int sin_mode_ = 0;                    // MX_SIN_MODE at build time (0: the reference's float through Ziv's strategy)
    uint32_t parity_ = 0;                 // which buffer of the double-buffered ports the current / last run uses
    bool building_alt_ = false;           // upload_group is filling desc_alt / extra_alt
    bool building_main_ = false;          // ... desc / extra (first buffers whatever the current parity is)
    hipStream_t tail_stream_ = nullptr;
    hipEvent_t ev_head_done_ = nullptr;   // recorded on stream_ when a run's earlier groups are queued
    hipEvent_t ev_tail_done_[2] = {nullptr, nullptr};   // recorded on tail_stream_ after the tail of a run (by parity)
    bool tail_pending_[2] = {false, false};
    bool overlap_this_run_ = false;
    // The tail launch of run k is HELD BACK until run k + 1 has queued its speculative EqThree launch, and goes behind a gate (k_tail_gate, one wave) that opens when that
    // launch's last workgroup has started: the next run's k_env_ticks runs alone (beside a Mixer bank it took 140 us instead of 9 and the EqThree launch behind it started
    // when the bank was nearly done: no overlap at all), the EqThree workgroups are placed on an empty chip, and the Mixer's waves fill what is left.  Every join
    // (mx_graph_sync, read-backs, mx_graph_tail_stream, an exchange's submit, a cut run) releases a held launch at once.  MX_TAIL_GATE=0: launched at once as in round 4.
    struct TailLaunch { const void* desc = nullptr; uint32_t n = 0, max_ch = 0; size_t frames = 0; int dup_mode = 0; hipEvent_t prof_ev = nullptr; };
    struct DeferredTail { bool pending = false; std::vector<TailLaunch> items; uint32_t parity = 0; hipEvent_t prof_begin = nullptr; } deferred_;   // the tail: every Mixer group from tail_gi_ on (a bank, or a bank and the buses above it), in order
    std::function<void(hipStream_t)> tail_hook_;
    std::vector<hipEvent_t> head_waits_;
    uint64_t n_gated_ = 0, n_at_once_ = 0;
    bool tail_held_this_span_ = false; std::vector<bool> prof_runs_held_;   // parallel to prof_runs_: that run's tail launch was held back (its events sit on the tail stream)
    DevBuf gate_flag_; uint32_t gate_seq_ = 0; bool gate_armed_ = false; int tail_gate_ = -1;
    void flush_deferred_tail(bool gated);
    void end_auto_tail();                 // the library-chosen second-stream mode ends for good (a host took a raw pointer that mode would not keep fresh)
    void wait_tail(int parity_or_all);    // stream_ waits for the tail launches that have not been waited for (-1: both)
    DevBuf& desc_buf(Group& g) { return building_alt_ ? g.desc_alt : g.desc; }
    DevBuf& extra_buf(Group& g) { return (building_alt_ && g.kind == MX_KIND_MIXER) ? g.extra_alt : g.extra; }
    const void* desc_of(const Group& g) const { return (parity_ && g.desc_alt.p) ? g.desc_alt.p : g.desc.p; }
    size_t run_off_frames_ = 0;   // base-rate frames before the span being launched (a run cut at scheduled parameter updates)
    uint64_t gates_version_ = 0;  // bumped whenever a Trigger's params or schedule change
    DevBuf eq_stats_;             // [8] u64 counters of the speculative EqThree kernel's proof / repair pass
    struct Stage { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; };
    Stage stage_[4]; uint32_t stage_next_ = 0;
    uint32_t prof_runs_count_ = 0;
    std::vector<uint32_t> sched_nodes_;     // nodes with a pending schedule (a 60 000-node graph must not be walked per tick)
    std::vector<uint32_t> plotter_nodes_;   // launched Plotter nodes
    std::vector<uint32_t> video_order_;     // the video nodes of order_, in run order
    bool prof_this_run_ = false;
    size_t plot_job_off_ = 0;
    size_t zero_off_ = 0;
    size_t slab_floats_ = 0;
    double lo_f_ = 0, hi_f_ = 0;
    DevBuf eq_tabs_;              // EqScanTab[4] for L = 4, 8, 16, 32
    // plotter staging
    DevBuf plot_stage_, plot_jobs_;
    DevBuf conv_stage_;           // i16 staging for sink / ingest conversions
    size_t last_frames_per_call_ = 0;
    bool prof_on_ = false;
    std::vector<std::vector<hipEvent_t>> prof_runs_;   // one event list (groups+1) per recorded run
    std::vector<std::vector<hipEvent_t>> prof_pool_;
    uint32_t last_calls_ = 0;
    // PerformanceInfo bookkeeping (last profiled run)
    std::vector<float> perf_group_ms_;   // per group (+1: video section) of the last collected run
    float perf_total_ms_ = 0.f; uint32_t perf_calls_ = 0;
    double perf_last_lag_s_ = -1.0;      // steady-clock seconds of the last over-budget tick; < 0: never
};

}  // namespace mx
