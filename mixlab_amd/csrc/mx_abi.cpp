// mx_abi.cpp -- the extern "C" boundary declared in include/mixlab_gpu.h.
//
// Convention (mirrors the reference's own FFI fencing, codec/src/ffmpeg/ioctx.rs:51-67,136-152):
// nothing unwinds across the boundary; every entry point catches, stashes the message in a
// thread-local, and returns a negative status.
#include <cstring>
#include <memory>
#include <string>

#include "mx_engine.hpp"

using mx::Error;
using mx::Graph;

struct mx_graph { std::unique_ptr<Graph> g; };

struct mx_module {
    uint32_t kind = 0;
    std::unique_ptr<Graph> g;   // node 0 = the module, node 1+i = SOURCE feeding input terminal i
    std::unique_ptr<mx::VideoMixer> vm;   // MX_KIND_VIDEO_MIXER: host frames in, host frames out
    uint32_t tps = 60;                    // TICKS_PER_SECOND (src/engine.rs:54) the module was created for
    std::vector<uint8_t> in_type, out_type;
};

static thread_local std::string t_last_error;
void mx_set_last_error(const std::string& s) { t_last_error = s; }   // shared with mx_abi_video.cpp

template <class F>
static int guard(F&& f) noexcept {
    try {
        f();
        return MX_OK;
    } catch (const Error& e) {
        t_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        t_last_error = "host allocation failed";
        return MX_ERR_NOMEM;
    } catch (const std::exception& e) {
        t_last_error = std::string("internal error: ") + e.what();
        return MX_ERR_INTERNAL;
    } catch (...) {
        t_last_error = "internal error: unknown exception";
        return MX_ERR_INTERNAL;
    }
}

#define REQUIRE(cond, msg) do { if (!(cond)) throw Error(MX_ERR_INVALID, msg); } while (0)

extern "C" {

const char* mx_last_error(void) { return t_last_error.c_str(); }
uint32_t mx_abi_version(void) { return MX_ABI_VERSION; }

int mx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { t_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); (void)hipGetLastError(); return MX_ERR_DEVICE; }
    return n;
}

int mx_graph_build(const mx_node* nodes, size_t n_nodes, const mx_edge* edges, size_t n_edges,
                   const mx_graph_opts* opts, mx_graph** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        mx_graph_opts o{};
        o.device = -1;
        if (opts) o = *opts;
        auto h = std::make_unique<mx_graph>();
        h->g = std::make_unique<Graph>(nodes, n_nodes, edges, n_edges, o);
        *out = h.release();
    });
}

void mx_graph_destroy(mx_graph* g) {
    (void)guard([&] { delete g; });
}

int mx_graph_samples_per_tick(const mx_graph* g, size_t* spt) {
    return guard([&] { REQUIRE(g && spt, "NULL argument"); *spt = g->g->spt(); });
}

int mx_graph_run_order(const mx_graph* g, uint32_t* order, size_t cap, size_t* n) {
    return guard([&] {
        REQUIRE(g && n, "NULL argument");
        const auto& o = g->g->run_order();
        *n = o.size();
        for (size_t i = 0; i < o.size() && i < cap && order; ++i) order[i] = o[i];
    });
}

int mx_graph_stream(mx_graph* g, void** stream) {
    return guard([&] { REQUIRE(g && stream, "NULL argument"); *stream = (void*)g->g->stream(); });
}
int mx_graph_tail_stream(mx_graph* g, void** stream) {
    return guard([&] { REQUIRE(g && stream, "NULL argument"); *stream = (void*)g->g->tail_stream(); });
}
int mx_graph_update_params(mx_graph* g, uint32_t node, const void* params, size_t params_len) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->update_params(node, params, params_len); });
}

int mx_graph_schedule_params(mx_graph* g, uint32_t node, uint32_t tick_in_run, const void* params, size_t params_len) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->schedule_params(node, tick_in_run, params, params_len); });
}

int mx_graph_schedule_params_batch(mx_graph* g, const mx_param_event* events, size_t n_events) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        REQUIRE(events || !n_events, "events is NULL");
        // all or nothing: every event is validated before the first one is queued
        for (size_t i = 0; i < n_events; ++i) g->g->check_schedule(events[i].node, events[i].params, events[i].params_len);
        for (size_t i = 0; i < n_events; ++i) g->g->schedule_params(events[i].node, events[i].tick_in_run, events[i].params, events[i].params_len);
    });
}

int mx_graph_eq_spec_stats(mx_graph* g, uint64_t* chunks_run, uint64_t* chunks_repaired) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        uint64_t v[8];
        g->g->eq_spec_stats(v);
        if (chunks_run) *chunks_run = v[0];
        if (chunks_repaired) *chunks_repaired = v[1];
    });
}

int mx_graph_debug_eq_records(mx_graph* g, void** device_records, size_t* bytes) {
    return guard([&] {
        REQUIRE(g && device_records, "NULL argument");
        g->g->sync();
        *device_records = g->g->debug_eq_records(bytes);
    });
}
int mx_graph_debug_tail_releases(mx_graph* g, uint64_t* gated, uint64_t* at_once) {
    return guard([&] { REQUIRE(g, "NULL argument"); g->g->tail_releases(gated, at_once); });
}

int mx_graph_eq_repair_stats(mx_graph* g, uint64_t out[8]) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        REQUIRE(out, "out is NULL");
        g->g->eq_spec_stats(out);
    });
}

int mx_graph_write_source(mx_graph* g, uint32_t node, const float* host_samples, size_t n_ticks) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->write_source(node, host_samples, n_ticks * g->g->spt()); });
}

int mx_graph_bind_source_device(mx_graph* g, uint32_t node, const void* device_ptr) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->bind_source(node, device_ptr); });
}

int mx_graph_run_ticks(mx_graph* g, uint64_t first_tick, uint32_t n_ticks) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        try {
            g->g->run(first_tick * (uint64_t)g->g->spt(), g->g->spt(), n_ticks);   // t = tick * SPT, src/engine.rs:490
        } catch (...) {
            g->g->drop_schedules();   // a run that failed part-way must not leave its updates queued for the next one
            throw;
        }
    });
}

int mx_graph_sync(mx_graph* g) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->sync(); });
}

int mx_graph_read_output(mx_graph* g, uint32_t node, uint32_t port, float* host_samples, size_t n_ticks) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->read_output(node, port, host_samples, n_ticks * g->g->spt()); });
}

int mx_graph_read_output_window(mx_graph* g, uint32_t node, uint32_t port, float* host_samples, size_t first_tick_in_run, size_t n_ticks) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        g->g->read_output(node, port, host_samples, n_ticks * g->g->spt(), first_tick_in_run * g->g->spt());
    });
}

int mx_graph_read_output_i16(mx_graph* g, uint32_t node, uint32_t port, int16_t* host_samples, size_t n_ticks) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->read_output_i16(node, port, host_samples, n_ticks * g->g->spt()); });
}

int mx_graph_write_source_i16(mx_graph* g, uint32_t node, const int16_t* host_samples, size_t n_ticks) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->write_source_i16(node, host_samples, n_ticks * g->g->spt()); });
}

int mx_graph_output_device_ptr(mx_graph* g, uint32_t node, uint32_t port, void** device_ptr, size_t* floats_per_tick) {
    return guard([&] {
        REQUIRE(g && device_ptr, "NULL argument");
        size_t fpf = 0;
        *device_ptr = g->g->output_ptr(node, port, &fpf);
        if (floats_per_tick) *floats_per_tick = fpf;
    });
}

int mx_graph_read_plotter(mx_graph* g, uint32_t node, uint32_t tick_in_run, float* left, float* right, int* fired) {
    return guard([&] {
        REQUIRE(g && left && right && fired, "NULL argument");
        *fired = g->g->read_plotter(node, tick_in_run, left, right);
    });
}

int mx_graph_profile_run(mx_graph* g, uint64_t first_tick, uint32_t n_ticks, float* ms_by_kind, float* ms_total) {
    return guard([&] {
        REQUIRE(g && ms_by_kind, "NULL argument");
        g->g->run(first_tick * (uint64_t)g->g->spt(), g->g->spt(), n_ticks, ms_by_kind, ms_total);
    });
}

int mx_graph_profile_enable(mx_graph* g, int on) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->profile_enable(on != 0); });
}

int mx_graph_profile_collect(mx_graph* g, float* ms_by_kind, float* ms_total, uint32_t* n_runs) {
    return guard([&] {
        REQUIRE(g && ms_by_kind, "NULL argument");
        const uint32_t n = g->g->profile_collect(ms_by_kind, ms_total);
        if (n_runs) *n_runs = n;
    });
}

int mx_graph_adopt_state(mx_graph* new_graph, mx_graph* old_graph, const int32_t* old_node_of_new, size_t n) {
    return guard([&] {
        REQUIRE(new_graph && old_graph, "graph is NULL");
        REQUIRE(new_graph != old_graph, "a graph cannot adopt its own state");
        new_graph->g->adopt_state(*old_graph->g, old_node_of_new, n);
    });
}

int mx_graph_performance_info(mx_graph* g, mx_performance_info* info, uint64_t* module_us, size_t cap) {
    return guard([&] {
        REQUIRE(g && info, "NULL argument");
        const Graph::Perf p = g->g->performance_info(module_us, cap);
        info->realtime = p.realtime ? 1 : 0; info->lag = p.lag; info->tick_rate = p.tick_rate;
        info->tick_budget_us = p.tick_budget_us; info->engine_us = p.engine_us; info->n_modules = (uint32_t)g->g->n_nodes();
    });
}

/* ---------------------------------------------------------------------------------------------- */
/* PCM ingest ring: variable-length decoded frames re-blocked to ticks (stream_input.rs:92-124)     */
/* ---------------------------------------------------------------------------------------------- */
struct mx_pcm_ring {
    std::vector<int16_t> q;     // interleaved samples not yet consumed, oldest first
    size_t head = 0;
    int16_t* stage = nullptr;   // page-locked staging for the H2D copy (the reference stages through ring buffers too, src/source.rs:97-98)
    size_t stage_cap = 0;
    ~mx_pcm_ring() { if (stage) (void)hipHostFree(stage); }
};

int mx_pcm_ring_create(mx_pcm_ring** out) {
    return guard([&] { REQUIRE(out, "out is NULL"); *out = new mx_pcm_ring(); });
}
void mx_pcm_ring_destroy(mx_pcm_ring* r) { (void)guard([&] { delete r; }); }

int mx_pcm_ring_push_i16(mx_pcm_ring* r, const int16_t* samples, size_t n) {
    return guard([&] {
        REQUIRE(r && (samples || !n), "NULL argument");
        if (r->head && r->head * 2 >= r->q.size()) { r->q.erase(r->q.begin(), r->q.begin() + (ptrdiff_t)r->head); r->head = 0; }   // compact
        r->q.insert(r->q.end(), samples, samples + n);
    });
}

int mx_pcm_ring_queued(const mx_pcm_ring* r, size_t* n) {
    return guard([&] { REQUIRE(r && n, "NULL argument"); *n = r->q.size() - r->head; });
}

int mx_pcm_ring_feed(mx_pcm_ring* r, mx_graph* g, uint32_t node, uint32_t n_ticks, size_t* zero_filled) {
    return guard([&] {
        REQUIRE(r && g, "NULL argument");
        REQUIRE(node < g->g->n_nodes() && g->g->node(node).kind == MX_KIND_SOURCE_STEREO, "node is not a SOURCE_STEREO");
        const size_t per_tick = 2 * g->g->spt();            // audio_out.len(), stream_input.rs:80
        const size_t need = per_tick * (size_t)n_ticks;
        if (need > r->stage_cap) {
            if (r->stage) { (void)hipHostFree(r->stage); r->stage = nullptr; r->stage_cap = 0; }
            mx::hip_check(hipHostMalloc((void**)&r->stage, need * sizeof(int16_t), hipHostMallocDefault), "hipHostMalloc(pcm staging)");
            r->stage_cap = need;
        }
        std::memset(r->stage, 0, need * sizeof(int16_t));   // util::zero for whatever the queue cannot fill (stream_input.rs:120-122)
        size_t missing = 0;
        for (uint32_t t = 0; t < n_ticks; ++t) {
            const size_t have = r->q.size() - r->head, take = have < per_tick ? have : per_tick;
            std::memcpy(r->stage + (size_t)t * per_tick, r->q.data() + r->head, take * sizeof(int16_t));   // partial frames stay queued (:113-116)
            r->head += take;
            missing += per_tick - take;
        }
        if (r->head == r->q.size()) { r->q.clear(); r->head = 0; }
        g->g->write_source_i16(node, r->stage, (size_t)n_ticks * g->g->spt());   // H2D as i16 from pinned memory, /32768 on the device (:167-173)
        if (zero_filled) *zero_filled = missing;
    });
}

/* ---------------------------------------------------------------------------------------------- */
/* per-module compatibility path                                                                    */
/* ---------------------------------------------------------------------------------------------- */

static void module_ports(uint32_t kind, size_t params_len, std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
    // one throw-away 1-node graph would also tell us, but keep this host-only: mirror of kind_ports()
    switch (kind) {
    case MX_KIND_AMPLIFIER: in = {MX_STEREO, MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_ENVELOPE: case MX_KIND_EQ_THREE: in = {MX_MONO}; out = {MX_MONO}; break;
    case MX_KIND_FM_SINE: in = {MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_MIXER: in.assign(params_len / sizeof(mx_mixer_channel_params), MX_STEREO); out = {MX_STEREO, MX_STEREO}; break;
    case MX_KIND_OSCILLATOR: in = {}; out = {MX_MONO, MX_STEREO}; break;
    case MX_KIND_PLOTTER: in = {MX_STEREO}; out = {}; break;
    case MX_KIND_STEREO_PANNER: in = {MX_MONO, MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_STEREO_SPLITTER: in = {MX_STEREO}; out = {MX_MONO, MX_MONO}; break;
    case MX_KIND_TRIGGER: in = {}; out = {MX_MONO}; break;
    case MX_KIND_FIR: in = {MX_STEREO}; out = {MX_STEREO}; break;
    default: throw Error(MX_ERR_INVALID, "kind has no per-module audio path");
    }
}

int mx_module_create_ex(uint32_t kind, const void* params, size_t params_len, const mx_graph_opts* opts, mx_module** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        auto m = std::make_unique<mx_module>();
        m->kind = kind;
        if (kind == MX_KIND_VIDEO_MIXER) {   // src/module/video_mixer.rs: 4 video inputs, program + A + B outputs
            REQUIRE(params && params_len == sizeof(mx_video_mixer_params), "params_len does not match mx_video_mixer_params");
            mx_video_mixer_params p; std::memcpy(&p, params, sizeof p);
            if (opts && opts->device >= 0) mx::hip_check(hipSetDevice(opts->device), "hipSetDevice");
            m->in_type.assign(4, MX_VIDEO); m->out_type.assign(3, MX_VIDEO);
            m->vm = std::make_unique<mx::VideoMixer>(p, opts ? opts->sample_rate : 0u, nullptr);
            m->tps = (opts && opts->ticks_per_second) ? opts->ticks_per_second : 60u;
            m->vm->set_lazy_program(false, m->tps);   // frame expiry and the program frame's duration follow the configured tick rate
            *out = m.release();
            return;
        }
        module_ports(kind, params_len, m->in_type, m->out_type);
        std::vector<mx_node> nodes(1 + m->in_type.size());
        std::vector<mx_edge> edges(m->in_type.size());
        nodes[0] = mx_node{kind, (uint32_t)params_len, params};
        for (size_t i = 0; i < m->in_type.size(); ++i) {
            nodes[1 + i] = mx_node{m->in_type[i] == MX_MONO ? (uint32_t)MX_KIND_SOURCE_MONO : (uint32_t)MX_KIND_SOURCE_STEREO, 0u, nullptr};
            edges[i] = mx_edge{(uint32_t)(1 + i), 0u, 0u, (uint32_t)i};
        }
        mx_graph_opts o{};
        o.device = -1;
        if (opts) { o.sample_rate = opts->sample_rate; o.ticks_per_second = opts->ticks_per_second; o.flags = opts->flags; o.device = opts->device; }
        o.max_ticks_per_run = 1;
        m->g = std::make_unique<Graph>(nodes.data(), nodes.size(), edges.data(), edges.size(), o);
        *out = m.release();
    });
}

int mx_module_create(uint32_t kind, const void* params, size_t params_len, mx_module** out) {
    return mx_module_create_ex(kind, params, params_len, nullptr, out);
}

int mx_module_update(mx_module* m, const void* params, size_t params_len) {
    return guard([&] {
        REQUIRE(m, "module is NULL");
        if (m->vm) {
            REQUIRE(params && params_len == sizeof(mx_video_mixer_params), "params_len does not match mx_video_mixer_params");
            mx_video_mixer_params p; std::memcpy(&p, params, sizeof p);
            m->vm->update(p);
            return;
        }
        m->g->update_params(0, params, params_len);
    });
}

int mx_module_run_tick(mx_module* m, uint64_t t, const mx_input* inputs, size_t n_inputs,
                       mx_output* outputs, size_t n_outputs, void* indication, size_t* indication_len) {
    return guard([&] {
        REQUIRE(m, "module is NULL");
        REQUIRE(n_inputs == m->in_type.size(), "wrong number of inputs for this module kind");
        REQUIRE(n_outputs == m->out_type.size(), "wrong number of outputs for this module kind");
        REQUIRE(!n_inputs || inputs, "inputs is NULL");
        REQUIRE(!n_outputs || outputs, "outputs is NULL");
        const size_t indication_cap = indication_len ? *indication_len : 0;   // in: capacity of `indication` in bytes
        if (indication_len) *indication_len = 0;
        if (m->vm) {   // VideoMixer::run_tick on host frames: upload, run on the device, download what the caller has room for
            hipStream_t st = m->vm->stream();
            mx::VideoInput vin[4];
            mx::FrameRef keep[4];
            for (size_t i = 0; i < 4; ++i) {
                if (inputs[i].kind == MX_DISCONNECTED || !inputs[i].video) continue;          // Disconnected / Video(None), io.rs:54-61
                if (inputs[i].kind != MX_VIDEO) throw Error(MX_ERR_TYPE, "input line type mismatch");
                const mx_frame* hf = inputs[i].video;
                keep[i] = mx::FrameRef(mx::DFrame::create(hf->width, hf->height, st), false);
                for (int p = 0; p < 3; ++p) {
                    REQUIRE(hf->data[p] && hf->stride[p] >= (int32_t)keep[i]->pw(p), "host frame plane is NULL or its stride is smaller than the width");
                    mx::hip_check(hipMemcpy2DAsync(keep[i]->data[p], keep[i]->stride[p], hf->data[p], (size_t)hf->stride[p], keep[i]->pw(p), keep[i]->ph(p),
                                                   hipMemcpyHostToDevice, st), "hipMemcpy2DAsync(H2D frame)");
                }
                vin[i].frame = keep[i].f;
                vin[i].duration_hint = mx::Rational::make(hf->dur_num, hf->dur_den ? hf->dur_den : 1);
                vin[i].tick_offset = mx::Rational::make(hf->off_num, hf->off_den ? hf->off_den : 1);
            }
            mx::FrameRef res[3];
            m->vm->run_tick(t, vin, res[0], res[1], res[2]);
            mx::flush_scales(st);
            for (size_t i = 0; i < 3; ++i) {
                if ((uint8_t)outputs[i].kind != MX_VIDEO) throw Error(MX_ERR_TYPE, "output line type mismatch");
                outputs[i].video_present = 0;
                if (!res[i] || !outputs[i].video) continue;
                mx_frame* hf = outputs[i].video;      // on entry width/height = capacity of the caller's planes
                if (hf->width < res[i]->width || hf->height < res[i]->height) throw Error(MX_ERR_INVALID, "output frame buffer is smaller than the composed picture");
                res[i]->ensure_pixels(st);
                for (int p = 0; p < 3; ++p) {
                    REQUIRE(hf->data[p] && hf->stride[p] >= (int32_t)res[i]->pw(p), "host frame plane is NULL or its stride is smaller than the width");
                    mx::hip_check(hipMemcpy2DAsync(hf->data[p], (size_t)hf->stride[p], res[i]->data[p], res[i]->stride[p], res[i]->pw(p), res[i]->ph(p),
                                                   hipMemcpyDeviceToHost, st), "hipMemcpy2DAsync(D2H frame)");
                }
                hf->width = res[i]->width; hf->height = res[i]->height;
                if (i == 0) { hf->dur_num = 1; hf->dur_den = (int64_t)m->tps; hf->off_num = 0; hf->off_den = 1; }     // video_mixer.rs:241-247 (1 / TICKS_PER_SECOND)
                else { const int src = (i == 1) ? m->vm->param_a() : m->vm->param_b();                      // clone of the input VideoFrame (:80-90)
                       if (src >= 0 && src < 4 && inputs[src].video) { hf->dur_num = inputs[src].video->dur_num; hf->dur_den = inputs[src].video->dur_den;
                                                                       hf->off_num = inputs[src].video->off_num; hf->off_den = inputs[src].video->off_den; } }
                outputs[i].video_present = 1;
            }
            mx::hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
            return;
        }

        // all terminals must agree on the number of frames in this call
        size_t frames = 0; bool have = false;
        auto note = [&](size_t len, uint8_t lt) {
            if (lt == MX_STEREO) { if (len & 1) throw Error(MX_ERR_INVALID, "stereo buffer length is odd"); len >>= 1; }
            if (have && len != frames) throw Error(MX_ERR_INVALID, "terminals disagree on buffer length");
            frames = len; have = true;
        };
        for (size_t i = 0; i < n_inputs; ++i) {
            if (inputs[i].kind == MX_DISCONNECTED) continue;
            // expect_mono()/expect_stereo() panic on the wrong line type (src/engine/io.rs:40-41,49-50)
            if ((uint8_t)inputs[i].kind != m->in_type[i]) throw Error(MX_ERR_TYPE, "input line type mismatch");
            REQUIRE(inputs[i].samples || !inputs[i].len, "input samples is NULL");
            note(inputs[i].len, m->in_type[i]);
        }
        for (size_t i = 0; i < n_outputs; ++i) {
            if ((uint8_t)outputs[i].kind != m->out_type[i]) throw Error(MX_ERR_TYPE, "output line type mismatch");
            REQUIRE(outputs[i].samples || !outputs[i].len, "output samples is NULL");
            note(outputs[i].len, m->out_type[i]);
        }
        Graph& g = *m->g;
        if (!have) frames = g.spt();
        if (frames == 0) return;

        g.ensure_capacity(frames);
        for (size_t i = 0; i < n_inputs; ++i) {
            const bool conn = inputs[i].kind != MX_DISCONNECTED;
            g.set_input_enabled(0, (uint32_t)i, conn);
            if (conn) g.write_source((uint32_t)(1 + i), inputs[i].samples, frames);
        }
        g.run(t, frames, 1);
        for (size_t i = 0; i < n_outputs; ++i) g.read_output(0, (uint32_t)i, outputs[i].samples, frames);
        if (m->kind == MX_KIND_PLOTTER) {
            g.sync();
            if (indication && indication_len) {
                // PlotterIndication { inputs: [left, right] } (plotter.rs:44-52): `frames` floats each, never more than the caller has room for
                if (indication_cap < 2 * frames * sizeof(float)) throw Error(MX_ERR_INVALID, "indication buffer is smaller than 2 * frames floats");
                float* l = (float*)indication;
                if (g.read_plotter(0, 0, l, l + frames)) *indication_len = 2 * frames * sizeof(float);
            }
        }
        g.sync();
    });
}

void mx_module_destroy(mx_module* m) {
    (void)guard([&] { delete m; });
}

}  // extern "C"
