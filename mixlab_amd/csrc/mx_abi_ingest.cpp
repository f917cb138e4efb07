// mx_abi_ingest.cpp -- extern "C" entry points of timed ingest (include/mixlab_gpu.h, "timed ingest").
// Same fencing convention as mx_abi.cpp: catch everything, stash the message, return a status.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mx_engine.hpp"
#include "mx_ingest.hpp"

using mx::DFrame;
using mx::Error;
using mx::Rational;

static inline DFrame* D(mx_dframe* p) { return reinterpret_cast<DFrame*>(p); }
static inline mx_dframe* H(DFrame* p) { return reinterpret_cast<mx_dframe*>(p); }

struct mx_graph { std::unique_ptr<mx::Graph> g; };   // same layout as in mx_abi.cpp
struct mx_media_source { mx::MediaSource m; mx_media_source(uint32_t sr, uint32_t tps) : m(sr, tps) {} };
struct mx_stream_input {
    mx::StreamInput s; uint32_t sr;
    int16_t* stage = nullptr; size_t stage_cap = 0;      // page-locked, like mx_pcm_ring's
    explicit mx_stream_input(uint32_t rate) : s(rate), sr(rate ? rate : 44100u) {}
    ~mx_stream_input() { if (stage) (void)hipHostFree(stage); }
};
struct mx_frame_stager { mx::FrameStager st; explicit mx_frame_stager(uint32_t slots) : st(slots) {} };

void mx_set_last_error(const std::string& s);   // mx_abi.cpp

template <class F>
static int guard(F&& f) noexcept {
    try {
        f();
        return MX_OK;
    } catch (const Error& e) {
        mx_set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        mx_set_last_error("host allocation failed");
        return MX_ERR_NOMEM;
    } catch (const std::exception& e) {
        mx_set_last_error(std::string("internal error: ") + e.what());
        return MX_ERR_INTERNAL;
    } catch (...) {
        mx_set_last_error("internal error: unknown exception");
        return MX_ERR_INTERNAL;
    }
}
#define REQUIRE(cond, msg) do { if (!(cond)) throw Error(MX_ERR_INVALID, msg); } while (0)

static Rational R(int64_t n, int64_t d) { return Rational::make(n, d ? d : 1); }
static void put(mx_video_input* out, mx::TickVideo& v) {
    out->frame = nullptr; out->dur_num = 0; out->dur_den = 1; out->off_num = 0; out->off_den = 1;
    if (!v.frame) return;
    v.frame->retain();
    out->frame = H(v.frame.f);
    out->dur_num = v.duration_hint.num; out->dur_den = v.duration_hint.den;
    out->off_num = v.tick_offset.num; out->off_den = v.tick_offset.den;
}

extern "C" {

int mx_graph_queue_video_source(mx_graph* g, uint32_t node, uint64_t tick, mx_dframe* frame, int64_t dur_num, int64_t dur_den,
                                int64_t off_num, int64_t off_den) {
    return guard([&] {
        REQUIRE(g && frame, "NULL argument");
        g->g->queue_video_source(node, tick, D(frame), R(dur_num, dur_den), R(off_num, off_den));
    });
}

/* ---- MediaSource ---- */
int mx_media_source_create(uint32_t sample_rate, uint32_t ticks_per_second, mx_media_source** out) {
    return guard([&] { REQUIRE(out, "out is NULL"); *out = nullptr; *out = new mx_media_source(sample_rate, ticks_per_second); });
}
void mx_media_source_destroy(mx_media_source* m) { (void)guard([&] { delete m; }); }
int mx_media_source_set_media(mx_media_source* m, int present) {
    return guard([&] { REQUIRE(m, "NULL argument"); m->m.set_media(present != 0); });
}
int mx_media_source_send(mx_media_source* m, mx_dframe* frame, int64_t pts_num, int64_t pts_den, int64_t dur_num, int64_t dur_den) {
    int full = 0;
    const int rc = guard([&] {
        REQUIRE(m && frame, "NULL argument");
        if (!m->m.send(D(frame), R(pts_num, pts_den), R(dur_num, dur_den))) full = 1;
    });
    if (rc == MX_OK && full) { mx_set_last_error("the decode channel holds two frames (sync_channel(2))"); return MX_ERR_FULL; }
    return rc;
}
int mx_media_source_run_tick(mx_media_source* m, uint64_t t, mx_video_input* out) {
    return guard([&] {
        REQUIRE(m && out, "NULL argument");
        mx::TickVideo v = m->m.run_tick(t);
        put(out, v);
    });
}
int mx_media_source_feed(mx_media_source* m, mx_graph* g, uint32_t node, uint64_t first_tick, uint32_t n_ticks) {
    return guard([&] {
        REQUIRE(m && g, "NULL argument");
        REQUIRE(node < g->g->n_nodes() && g->g->node(node).kind == MX_KIND_SOURCE_VIDEO, "node is not a SOURCE_VIDEO");
        // everything that can fail is checked BEFORE a frame leaves the pacing state machine: a rejected feed loses no media
        REQUIRE((double)m->m.sample_rate() == g->g->sample_rate() && m->m.ticks_per_second() == g->g->ticks_per_second(),
                "the media source and the graph run at different sample / tick rates");
        g->g->check_video_queue(node, first_tick);
        for (uint32_t k = 0; k < n_ticks; ++k) {
            mx::TickVideo v = m->m.run_tick((first_tick + k) * (uint64_t)g->g->spt());   // t of tick k, src/engine.rs:490
            if (v.frame) g->g->queue_video_source(node, first_tick + k, v.frame.f, v.duration_hint, v.tick_offset);
        }
    });
}

/* ---- StreamInput ---- */
int mx_stream_input_create(uint32_t sample_rate, mx_stream_input** out) {
    return guard([&] { REQUIRE(out, "out is NULL"); *out = nullptr; *out = new mx_stream_input(sample_rate); });
}
void mx_stream_input_destroy(mx_stream_input* s) { (void)guard([&] { delete s; }); }
int mx_stream_input_listen(mx_stream_input* s, int listening) {
    return guard([&] { REQUIRE(s, "NULL argument"); s->s.listen(listening != 0); });
}
int mx_stream_input_write_audio(mx_stream_input* s, uint64_t source_id, int64_t ts_num, int64_t ts_den, const int16_t* interleaved, size_t n_samples) {
    int full = 0;
    const int rc = guard([&] {
        REQUIRE(s, "NULL argument");
        if (!s->s.write_audio(source_id, R(ts_num, ts_den), interleaved, n_samples)) full = 1;
    });
    if (rc == MX_OK && full) { mx_set_last_error("the audio ring is full or nobody listens (src/source.rs:158-174)"); return MX_ERR_FULL; }
    return rc;
}
int mx_stream_input_write_video(mx_stream_input* s, uint64_t source_id, int64_t ts_num, int64_t ts_den, mx_dframe* frame, int64_t dur_num, int64_t dur_den) {
    int full = 0;
    const int rc = guard([&] {
        REQUIRE(s && frame, "NULL argument");
        if (!s->s.write_video(source_id, R(ts_num, ts_den), D(frame), R(dur_num, dur_den))) full = 1;
    });
    if (rc == MX_OK && full) { mx_set_last_error("the video ring is full or nobody listens (src/source.rs:176-190)"); return MX_ERR_FULL; }
    return rc;
}
int mx_stream_input_run_tick(mx_stream_input* s, uint64_t t, int16_t* audio_out, size_t n_out, mx_video_input* video_out, size_t* zero_filled) {
    return guard([&] {
        REQUIRE(s && video_out && (audio_out || !n_out), "NULL argument");
        REQUIRE(n_out % 2 == 0, "audio_out holds interleaved stereo: an even number of samples");
        mx::TickVideo v = s->s.run_tick(t, audio_out, n_out, zero_filled);
        put(video_out, v);
    });
}
int mx_stream_input_feed(mx_stream_input* s, mx_graph* g, uint32_t audio_node, uint32_t video_node, uint64_t first_tick, uint32_t n_ticks,
                         size_t* zero_filled) {
    return guard([&] {
        REQUIRE(s && g, "NULL argument");
        REQUIRE(audio_node < g->g->n_nodes() && g->g->node(audio_node).kind == MX_KIND_SOURCE_STEREO, "audio node is not a SOURCE_STEREO");
        const bool with_video = video_node != UINT32_MAX;
        if (with_video) REQUIRE(video_node < g->g->n_nodes() && g->g->node(video_node).kind == MX_KIND_SOURCE_VIDEO, "video node is not a SOURCE_VIDEO");
        REQUIRE((double)s->sr == g->g->sample_rate(), "the stream input and the graph run at different sample rates");
        // everything that can fail is checked BEFORE the rings are drained: a rejected feed loses no media
        g->g->check_source_write(audio_node, (size_t)n_ticks * g->g->spt());
        if (with_video) g->g->check_video_queue(video_node, first_tick);
        const size_t per_tick = 2 * g->g->spt(), need = per_tick * (size_t)n_ticks;
        if (need > s->stage_cap) {
            if (s->stage) { (void)hipHostFree(s->stage); s->stage = nullptr; s->stage_cap = 0; }
            mx::hip_check(hipHostMalloc((void**)&s->stage, need * sizeof(int16_t), hipHostMallocDefault), "hipHostMalloc(pcm staging)");
            s->stage_cap = need;
        }
        size_t missing = 0;
        for (uint32_t k = 0; k < n_ticks; ++k) {
            size_t z = 0;
            mx::TickVideo v = s->s.run_tick((first_tick + k) * (uint64_t)g->g->spt(), s->stage + (size_t)k * per_tick, per_tick, &z);
            missing += z;
            if (v.frame && with_video) g->g->queue_video_source(video_node, first_tick + k, v.frame.f, v.duration_hint, v.tick_offset);
        }
        g->g->write_source_i16(audio_node, s->stage, (size_t)n_ticks * g->g->spt());   // H2D as i16, /32768 on the device (stream_input.rs:167-173)
        if (zero_filled) *zero_filled = missing;
    });
}

/* ---- Monitor / StreamOutput hand-off ---- */
int mx_graph_read_monitor_tick(mx_graph* g, uint32_t node, uint32_t tick_in_run, mx_monitor_tick* info, mx_dframe** frame) {
    return guard([&] {
        REQUIRE(g && info && frame, "NULL argument");
        *frame = nullptr; std::memset(info, 0, sizeof *info);
        info->ts_den = info->frame_ts_den = info->dur_den = 1;
        const mx::Node::MonTick& m = g->g->monitor_tick(node, tick_in_run);
        info->ts_num = m.ts.num; info->ts_den = m.ts.den;
        info->dropped = m.dropped ? 1 : 0;
        if (!m.present) return;
        info->video_present = 1;
        info->frame_ts_num = m.frame_ts.num; info->frame_ts_den = m.frame_ts.den;
        info->dur_num = m.dur.num; info->dur_den = m.dur.den;
        m.frame->retain();
        *frame = H(m.frame.f);
    });
}
int mx_graph_monitor_consume(mx_graph* g, uint32_t node, uint32_t n_ticks) {
    return guard([&] { REQUIRE(g, "NULL argument"); g->g->monitor_consume(node, n_ticks); });
}
int mx_graph_read_monitor_audio_i16(mx_graph* g, uint32_t node, int16_t* audio, uint32_t n_ticks) {
    return guard([&] { REQUIRE(g, "NULL argument"); g->g->read_monitor_audio_i16(node, audio, n_ticks); });
}

int mx_graph_monitor_layout(mx_graph* g, uint32_t node, mx_monitor_layout* out) {
    return guard([&] {
        REQUIRE(g && out, "NULL argument");
        const mx::Graph::MonitorLayout l = g->g->monitor_layout(node);
        out->width = l.width; out->height = l.height; out->frame_bytes = l.frame_bytes;
        for (int p = 0; p < 3; ++p) { out->plane_offset[p] = l.plane_offset[p]; out->stride[p] = (int32_t)l.stride[p]; }
    });
}
int mx_graph_read_monitor_video(mx_graph* g, uint32_t node, uint32_t first_tick, uint32_t n_ticks, uint8_t* frames, uint8_t* present) {
    return guard([&] { REQUIRE(g, "NULL argument"); g->g->read_monitor_video(node, first_tick, n_ticks, frames, present); });
}

int mx_host_alloc(size_t bytes, void** host_ptr) {
    return guard([&] {
        REQUIRE(host_ptr, "host_ptr is NULL");
        *host_ptr = nullptr;
        hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault);
        if (e == hipErrorOutOfMemory) { *host_ptr = nullptr; throw Error(MX_ERR_NOMEM, "hipHostMalloc: out of page-locked memory"); }
        mx::hip_check(e, "hipHostMalloc");
    });
}
void mx_host_free(void* host_ptr) { (void)guard([&] { if (host_ptr) mx::hip_check(hipHostFree(host_ptr), "hipHostFree"); }); }

/* ---- frame staging ---- */
int mx_frame_stager_create(uint32_t slots, mx_frame_stager** out) {
    return guard([&] { REQUIRE(out, "out is NULL"); *out = nullptr; *out = new mx_frame_stager(slots); });
}
void mx_frame_stager_destroy(mx_frame_stager* st) { (void)guard([&] { delete st; }); }
int mx_frame_stager_upload(mx_frame_stager* st, const mx_frame* host, mx_pixfmt fmt, mx_dframe** out) {
    return guard([&] {
        REQUIRE(st && host && out, "NULL argument");
        *out = nullptr;
        REQUIRE((int)fmt >= 0 && (int)fmt <= (int)mx::DFrame::kLastFmt, "unknown pixel format");
        *out = H(st->st.upload(host->width, host->height, (uint8_t)fmt, host->data, host->stride));
    });
}
int mx_frame_stager_acquire(mx_frame_stager* st, uint32_t width, uint32_t height, mx_pixfmt fmt, mx_frame* host, uint32_t* ticket) {
    return guard([&] {
        REQUIRE(st && host && ticket, "NULL argument");
        REQUIRE((int)fmt >= 0 && (int)fmt <= (int)mx::DFrame::kLastFmt, "unknown pixel format");
        *ticket = 0;
        std::memset(host, 0, sizeof *host);
        host->dur_den = 1; host->off_den = 1;
        *ticket = st->st.acquire(width, height, (uint8_t)fmt, host->data, host->stride);
        host->width = width; host->height = height;
    });
}
int mx_frame_stager_commit(mx_frame_stager* st, uint32_t ticket, mx_dframe** out) {
    return guard([&] { REQUIRE(st && out, "NULL argument"); *out = nullptr; *out = H(st->st.commit(ticket)); });
}
int mx_frame_stager_fence(mx_frame_stager* st, void* stream) {
    return guard([&] { REQUIRE(st, "NULL argument"); st->st.fence((hipStream_t)stream); });
}
int mx_frame_stager_fence_graph(mx_frame_stager* st, mx_graph* g) {
    return guard([&] { REQUIRE(st && g, "NULL argument"); st->st.fence(g->g->stream()); });
}
int mx_frame_stager_sync(mx_frame_stager* st) {
    return guard([&] { REQUIRE(st, "NULL argument"); st->st.sync(); });
}

}  // extern "C"
