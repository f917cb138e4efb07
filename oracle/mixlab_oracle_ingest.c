/* mixlab_oracle_ingest.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * Restates, call by call, how the reference's two ingest modules decide WHICH frame leaves on WHICH tick and with what offset:
 *   MediaSource::run_tick   src/module/media_source.rs:93-126  (channel: mpsc::sync_channel(2), :140)
 *   StreamInput::run_tick   src/module/stream_input.rs:72-147  (rings: RingBuffer::new(65536), src/source.rs:97-98)
 * Frames are opaque here: a positive id stands for the video::Frame, pixels never enter the decision.  Times are exact rationals
 * (util/src/time.rs:10-75, num_rational::Rational64).
 * Parity: unpinned by reference tests (the reference has none for these modules); the source text is the spec. */
#include <stdlib.h>
#include <string.h>

#include "mixlab_oracle.h"

static orc_rational rat_sub(orc_rational a, orc_rational b) { b.num = -b.num; return orc_rational_add(a, b); }

/* ------------------------------------------------------------------------------------------- MediaSource */
#define ORC_MS_BUF 4096
struct orc_media_source {
    uint32_t sample_rate, ticks_per_second;
    int present;                                     /* self.media.is_some() */
    int chan_n; orc_timed_frame chan[2];             /* what the decode thread has sent and run_tick has not received yet */
    int has_epoch; orc_rational epoch;               /* OpenMedia.epoch */
    int buf_head, buf_n; orc_timed_frame buf[ORC_MS_BUF];   /* OpenMedia.video_buffer (VecDeque) */
};

orc_media_source* orc_media_source_new(uint32_t sample_rate, uint32_t ticks_per_second) {
    orc_media_source* m = (orc_media_source*)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->sample_rate = sample_rate; m->ticks_per_second = ticks_per_second;
    return m;
}
void orc_media_source_free(orc_media_source* m) { free(m); }

void orc_media_source_set_media(orc_media_source* m, int present) {   /* :85-91 + :140-147 */
    m->present = present; m->chan_n = 0; m->has_epoch = 0; m->buf_head = 0; m->buf_n = 0;
}

int orc_media_source_send(orc_media_source* m, int64_t frame_id, orc_rational pts, orc_rational duration_hint) {   /* tx.send, :271 */
    if (!m->present) return -1;          /* receiver disconnected */
    if (m->chan_n == 2) return 0;        /* would block */
    m->chan[m->chan_n].frame_id = frame_id; m->chan[m->chan_n].time = pts; m->chan[m->chan_n].duration_hint = duration_hint;
    m->chan_n++;
    return 1;
}

orc_tick_video orc_media_source_run_tick(orc_media_source* m, uint64_t t) {
    orc_tick_video out; memset(&out, 0, sizeof out); out.duration_hint.den = 1; out.tick_offset.den = 1;
    const orc_rational start_of_frame = orc_rational_new((int64_t)t, (int64_t)m->sample_rate);
    const orc_rational end_of_frame = orc_rational_add(start_of_frame, orc_rational_new(1, (int64_t)m->ticks_per_second));
    if (!m->present) return out;
    if (m->chan_n > 0) {                                                   /* Ok(frame) => ... :102-110 */
        orc_timed_frame f = m->chan[0];
        m->chan[0] = m->chan[1]; m->chan_n--;
        if (!m->has_epoch) { m->has_epoch = 1; m->epoch = start_of_frame; }
        f.time = orc_rational_add(f.time, m->epoch);
        if (m->buf_n < ORC_MS_BUF) { m->buf[(m->buf_head + m->buf_n) % ORC_MS_BUF] = f; m->buf_n++; }
    }
    if (m->buf_n > 0) {                                                    /* :113-123 */
        const orc_timed_frame* front = &m->buf[m->buf_head];
        if (orc_rational_cmp(front->time, end_of_frame) < 0) {
            out.frame_id = front->frame_id; out.duration_hint = front->duration_hint;
            out.tick_offset = rat_sub(front->time, start_of_frame);
            m->buf_head = (m->buf_head + 1) % ORC_MS_BUF; m->buf_n--;
        }
    }
    return out;
}

/* ------------------------------------------------------------------------------------------- StreamInput */
typedef struct { uint64_t source_id; orc_rational source_time; int16_t* data; size_t len, head; } orc_si_audio;
typedef struct { uint64_t source_id; orc_rational source_time; int64_t frame_id; orc_rational duration_hint; } orc_si_video;
struct orc_stream_input {
    uint32_t sample_rate;
    int listening;                                    /* self.recv.is_some() */
    orc_si_audio* arx; size_t a_head, a_n, a_cap;     /* SourceRecv.audio_rx */
    orc_si_video* vrx; size_t v_head, v_n, v_cap;     /* SourceRecv.video_rx */
    int has_audio_frame; orc_si_audio audio_frame;    /* self.audio_frame */
    int has_video_frame; orc_si_video video_frame;    /* self.video_frame */
    int has_source; uint64_t source_id; orc_rational source_epoch;   /* self.source */
};
#define ORC_SI_RING 65536

orc_stream_input* orc_stream_input_new(uint32_t sample_rate) {
    orc_stream_input* s = (orc_stream_input*)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->sample_rate = sample_rate; s->listening = 1;
    s->a_cap = s->v_cap = ORC_SI_RING;
    s->arx = (orc_si_audio*)calloc(s->a_cap, sizeof *s->arx);
    s->vrx = (orc_si_video*)calloc(s->v_cap, sizeof *s->vrx);
    if (!s->arx || !s->vrx) { free(s->arx); free(s->vrx); free(s); return NULL; }
    return s;
}
static void si_drop_rings(orc_stream_input* s) {
    for (size_t i = 0; i < s->a_n; ++i) free(s->arx[(s->a_head + i) % s->a_cap].data);
    s->a_head = s->a_n = 0; s->v_head = s->v_n = 0;
}
void orc_stream_input_free(orc_stream_input* s) {
    if (!s) return;
    si_drop_rings(s);
    if (s->has_audio_frame) free(s->audio_frame.data);
    free(s->arx); free(s->vrx); free(s);
}
void orc_stream_input_listen(orc_stream_input* s, int listening) { si_drop_rings(s); s->listening = listening; }   /* :57-70 */

int orc_stream_input_write_audio(orc_stream_input* s, uint64_t source_id, orc_rational source_time, const int16_t* data, size_t n) {
    if (!s->listening || s->a_n == s->a_cap) return 0;
    orc_si_audio* f = &s->arx[(s->a_head + s->a_n) % s->a_cap];
    f->source_id = source_id; f->source_time = source_time; f->len = n; f->head = 0;
    f->data = (int16_t*)malloc((n ? n : 1) * sizeof(int16_t));
    if (!f->data) return 0;
    memcpy(f->data, data, n * sizeof(int16_t));
    s->a_n++;
    return 1;
}
int orc_stream_input_write_video(orc_stream_input* s, uint64_t source_id, orc_rational source_time, int64_t frame_id, orc_rational duration_hint) {
    if (!s->listening || s->v_n == s->v_cap) return 0;
    orc_si_video* f = &s->vrx[(s->v_head + s->v_n) % s->v_cap];
    f->source_id = source_id; f->source_time = source_time; f->frame_id = frame_id; f->duration_hint = duration_hint;
    s->v_n++;
    return 1;
}

orc_tick_video orc_stream_input_run_tick(orc_stream_input* s, uint64_t t, int16_t* audio_out, size_t n_out, size_t* zero_filled) {
    orc_tick_video out; memset(&out, 0, sizeof out); out.duration_hint.den = 1; out.tick_offset.den = 1;
    const orc_rational engine_time = orc_rational_new((int64_t)t, (int64_t)s->sample_rate);
    const orc_rational tick_duration = orc_rational_new((int64_t)(n_out / 2), (int64_t)s->sample_rate);
    /* :82-86 */
    int has_video = 0; orc_si_video video; memset(&video, 0, sizeof video);
    if (s->has_video_frame) { video = s->video_frame; has_video = 1; s->has_video_frame = 0; }
    else if (s->v_n) { video = s->vrx[s->v_head]; s->v_head = (s->v_head + 1) % s->v_cap; s->v_n--; has_video = 1; }
    const int existed = s->has_source; const uint64_t existing_source_id = s->source_id;   /* :88 */
    size_t filled = 0, zeroed = 0;
    while (filled < n_out) {                                                               /* :92-124 */
        orc_si_audio frame; int got = 0;
        if (s->has_audio_frame) { frame = s->audio_frame; s->has_audio_frame = 0; got = 1; }
        else if (s->a_n) { frame = s->arx[s->a_head]; s->a_head = (s->a_head + 1) % s->a_cap; s->a_n--; got = 1; }
        if (!got) {
            for (size_t i = filled; i < n_out; ++i) audio_out[i] = 0;
            zeroed = n_out - filled;
            break;
        }
        if (!(existed && existing_source_id == frame.source_id)) {
            s->has_source = 1; s->source_id = frame.source_id;
            s->source_epoch = rat_sub(engine_time, frame.source_time);
        }
        const size_t remaining = frame.len - frame.head;
        const size_t len = remaining < n_out - filled ? remaining : n_out - filled;
        for (size_t i = 0; i < len; ++i) audio_out[filled + i] = frame.data[frame.head + i];
        filled += len;
        if (len < remaining) { frame.head += len; s->audio_frame = frame; s->has_audio_frame = 1; }
        else free(frame.data);
    }
    if (zero_filled) *zero_filled = zeroed;
    if (has_video) {                                                                       /* :126-146 */
        orc_rational tick_offset = orc_rational_new(0, 1);
        if (s->has_source) {
            const orc_rational d = rat_sub(orc_rational_add(video.source_time, s->source_epoch), engine_time);
            if (orc_rational_cmp(d, orc_rational_new(0, 1)) >= 0) tick_offset = d;
        }
        if (orc_rational_cmp(tick_offset, tick_duration) > 0) { s->video_frame = video; s->has_video_frame = 1; }
        else { out.frame_id = video.frame_id; out.duration_hint = video.duration_hint; out.tick_offset = tick_offset; }
    }
    return out;
}
