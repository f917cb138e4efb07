// mx_ingest.cpp -- MediaSource / StreamInput pacing and the H2D staging ring (see mx_ingest.hpp).
#include "mx_ingest.hpp"

#include <cstring>

namespace mx {

// ---------------------------------------------------------------------------------------------
// MediaSource (src/module/media_source.rs)
// ---------------------------------------------------------------------------------------------
MediaSource::MediaSource(uint32_t sample_rate, uint32_t ticks_per_second) : sr_(sample_rate ? sample_rate : 44100u), tps_(ticks_per_second ? ticks_per_second : 60u) {}

void MediaSource::set_media(bool present) {
    std::lock_guard<std::mutex> lk(mu_);
    present_ = present;
    chan_.clear(); buffer_.clear(); have_epoch_ = false;
}

bool MediaSource::send(DFrame* frame, Rational pts, Rational duration_hint) {
    if (!frame) throw Error(MX_ERR_INVALID, "frame is NULL");
    std::lock_guard<std::mutex> lk(mu_);
    if (!present_) throw Error(MX_ERR_INVALID, "no media is open (the receiver is gone: the reference's decode thread ends here, media_source.rs:271-276)");
    if (chan_.size() >= 2) return false;
    chan_.push_back(Timed{FrameRef(frame, true), pts, duration_hint});
    return true;
}

TickVideo MediaSource::run_tick(uint64_t t) {
    TickVideo out;
    const Rational start_of_frame = Rational::make((int64_t)t, (int64_t)sr_);          // :94
    const Rational end_of_frame = start_of_frame + Rational::make(1, (int64_t)tps_);   // :95
    if (!present_) return out;                                                         // :97
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!chan_.empty()) {                                                          // try_recv: at most one frame per tick (:98-111)
            Timed f = std::move(chan_.front()); chan_.pop_front();
            if (!have_epoch_) { epoch_ = start_of_frame; have_epoch_ = true; }         // get_or_insert (:104)
            f.pts = f.pts + epoch_;                                                    // add_epoch (:107)
            buffer_.push_back(std::move(f));
        }
    }
    if (!buffer_.empty() && buffer_.front().pts < end_of_frame) {                      // :113-114
        Timed& f = buffer_.front();
        out.frame = f.frame; out.duration_hint = f.dur;
        out.tick_offset = f.pts - start_of_frame;                                      // :117 (negative for a late frame)
        buffer_.pop_front();
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// StreamInput (src/module/stream_input.rs)
// ---------------------------------------------------------------------------------------------
StreamInput::StreamInput(uint32_t sample_rate) : sr_(sample_rate ? sample_rate : 44100u) {}

bool StreamInput::write_audio(uint64_t source_id, Rational source_time, const int16_t* interleaved, size_t n) {
    if (!source_id) throw Error(MX_ERR_INVALID, "source id 0 (SourceId is a NonZeroUsize, src/source.rs:35)");
    if (!interleaved && n) throw Error(MX_ERR_INVALID, "samples is NULL");
    std::lock_guard<std::mutex> lk(mu_);
    if (!listening_ || audio_rx_.size() >= RING_FRAMES) return false;
    AudioFrame f; f.source_id = source_id; f.source_time = source_time; f.data.assign(interleaved, interleaved + n);
    audio_rx_.push_back(std::move(f));
    return true;
}

bool StreamInput::write_video(uint64_t source_id, Rational source_time, DFrame* frame, Rational duration_hint) {
    if (!source_id) throw Error(MX_ERR_INVALID, "source id 0 (SourceId is a NonZeroUsize, src/source.rs:35)");
    if (!frame) throw Error(MX_ERR_INVALID, "frame is NULL");
    std::lock_guard<std::mutex> lk(mu_);
    if (!listening_ || video_rx_.size() >= RING_FRAMES) return false;
    video_rx_.push_back(VideoFrameIn{source_id, source_time, FrameRef(frame, true), duration_hint});
    return true;
}

void StreamInput::listen(bool listening) {
    std::lock_guard<std::mutex> lk(mu_);
    listening_ = listening;
    audio_rx_.clear(); video_rx_.clear();
}

TickVideo StreamInput::run_tick(uint64_t t, int16_t* audio_out, size_t n_out, size_t* zero_filled) {
    const Rational engine_time = Rational::make((int64_t)t, (int64_t)sr_);                       // :73
    const Rational tick_duration = Rational::make((int64_t)(n_out / 2), (int64_t)sr_);           // :80
    // :82-86  the held video frame, else the next one of the ring
    bool have_video = have_video_frame_;
    VideoFrameIn video = std::move(video_frame_);
    have_video_frame_ = false; video_frame_ = VideoFrameIn{};
    if (!have_video) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!video_rx_.empty()) { video = std::move(video_rx_.front()); video_rx_.pop_front(); have_video = true; }
    }
    const bool had_source = have_source_; const uint64_t existing_source_id = source_id_;        // :88
    size_t pos = 0, zeroed = 0;
    while (pos < n_out) {                                                                        // :92
        bool have = have_audio_frame_;
        AudioFrame frame = std::move(audio_frame_);
        have_audio_frame_ = false; audio_frame_ = AudioFrame{};
        if (!have) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!audio_rx_.empty()) { frame = std::move(audio_rx_.front()); audio_rx_.pop_front(); have = true; }
        }
        if (!have) {                                                                             // :120-123
            std::memset(audio_out + pos, 0, (n_out - pos) * sizeof(int16_t));
            zeroed = n_out - pos;
            break;
        }
        if (!had_source || existing_source_id != frame.source_id) {                              // :100-106  (compared with the id the tick STARTED with)
            have_source_ = true; source_id_ = frame.source_id;
            source_epoch_ = engine_time - frame.source_time;                                     // remove_epoch
        }
        const size_t avail = frame.data.size() - frame.head, len = avail < n_out - pos ? avail : n_out - pos;   // :108
        std::memcpy(audio_out + pos, frame.data.data() + frame.head, len * sizeof(int16_t));
        pos += len;
        if (len < avail) { frame.head += len; audio_frame_ = std::move(frame); have_audio_frame_ = true; }   // :116-119
    }
    if (zero_filled) *zero_filled = zeroed;
    TickVideo out;
    if (have_video) {                                                                            // :126-146
        Rational tick_offset = Rational::make(0, 1);
        if (have_source_) {
            const Rational d = (video.source_time + source_epoch_) - engine_time;
            if (d >= Rational::make(0, 1)) tick_offset = d;
        }
        if (tick_offset > tick_duration) {       // not due in this tick: put it back
            video_frame_ = std::move(video); have_video_frame_ = true;
        } else {
            out.frame = video.frame; out.duration_hint = video.dur; out.tick_offset = tick_offset;
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// FrameStager
// ---------------------------------------------------------------------------------------------
FrameStager::FrameStager(uint32_t slots) {
    if (slots == 0 || slots > 64) throw Error(MX_ERR_INVALID, "stager slots must be 1..64");
    // HIP's current device is per THREAD (default 0): the decode thread that acquires / commits is not the one that created the
    // stager, so every entry point selects the creator's device before it allocates, records or copies
    hip_check(hipGetDevice(&device_), "hipGetDevice");
    hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate(stager)");
    hip_check(hipEventCreateWithFlags(&last_, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&reuse_, hipEventDisableTiming), "hipEventCreate");
    slots_.resize(slots);
    for (Slot& s : slots_) hip_check(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
}

FrameStager::~FrameStager() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (Slot& s : slots_) { if (s.host) (void)hipHostFree(s.host); if (s.done) (void)hipEventDestroy(s.done); }
    if (last_) (void)hipEventDestroy(last_);
    if (reuse_) (void)hipEventDestroy(reuse_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

FrameRef FrameStager::take_frame(uint32_t w, uint32_t h, uint8_t fmt) {
    if (have_consumer_)   // without a known consumer stream nothing orders a rewrite after the last reader: allocate instead
        for (FrameRef& c : pool_)
            if (c->width == w && c->height == h && c->fmt == fmt && c->rc.load(std::memory_order_acquire) == 1) {   // only the pool holds it
                hip_check(hipEventRecord(reuse_, consumer_), "hipEventRecord");                 // its last reader has been queued there already
                hip_check(hipStreamWaitEvent(stream_, reuse_, 0), "hipStreamWaitEvent(stager reuse)");
                return c;
            }
    FrameRef f(DFrame::create_unfilled(w, h, fmt), false);
    if (pool_.size() >= 32) pool_.erase(pool_.begin());
    pool_.push_back(f);
    return f;
}

uint32_t FrameStager::acquire(uint32_t w, uint32_t h, uint8_t fmt, uint8_t* data[3], int32_t stride[3]) {
    std::lock_guard<std::mutex> lk(mu_);
    return acquire_locked(w, h, fmt, data, stride);
}

uint32_t FrameStager::acquire_locked(uint32_t w, uint32_t h, uint8_t fmt, uint8_t* data[3], int32_t stride[3]) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    FrameRef f = take_frame(w, h, fmt);
    uint32_t k = next_, tries = 0;
    while (slots_[k].held) {                       // slots a decoder still writes are skipped
        k = (k + 1) % (uint32_t)slots_.size();
        if (++tries == slots_.size()) throw Error(MX_ERR_FULL, "every staging slot is held by an uncommitted acquire");
    }
    next_ = (k + 1) % (uint32_t)slots_.size();
    Slot& s = slots_[k];
    if (s.in_flight) { hip_check(hipEventSynchronize(s.done), "hipEventSynchronize(stager slot)"); s.in_flight = false; }
    const size_t total = f->mem.bytes;
    if (s.cap < total) {
        if (s.host) { (void)hipHostFree(s.host); s.host = nullptr; s.cap = 0; }
        hip_check(hipHostMalloc((void**)&s.host, total, hipHostMallocDefault), "hipHostMalloc(frame staging)");
        s.cap = total; s.padded_for = 0;
    }
    if (!(s.padded_for && s.pw == w && s.ph == h && s.pfmt == fmt)) {   // the padding a device frame is created with (frame.rs:76-138): written once per layout
        for (int p = 0; p < f->stored_planes(); ++p) std::memset(s.host + f->plane_offset(p), p ? 0x80 : 0x00, f->plane_bytes[p]);
        s.padded_for = 1; s.pw = w; s.ph = h; s.pfmt = fmt;
    }
    for (int p = 0; p < 3; ++p) {
        const bool stored = p < f->stored_planes();
        data[p] = stored ? s.host + f->plane_offset(p) : nullptr;
        stride[p] = stored ? (int32_t)f->stride[p] : 0;
    }
    s.target = f; s.held = true;
    return k + 1;
}

DFrame* FrameStager::commit(uint32_t ticket) {
    std::lock_guard<std::mutex> lk(mu_);
    return commit_locked(ticket);
}

DFrame* FrameStager::commit_locked(uint32_t ticket) {
    if (ticket == 0 || ticket > slots_.size() || !slots_[ticket - 1].held) throw Error(MX_ERR_INVALID, "not a ticket of an acquired slot");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    Slot& s = slots_[ticket - 1];
    FrameRef f = s.target;
    s.held = false; s.target = FrameRef();
    hip_check(hipMemcpyAsync(f->mem.p, s.host, f->mem.bytes, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D frame)");
    hip_check(hipEventRecord(s.done, stream_), "hipEventRecord");
    hip_check(hipEventRecord(last_, stream_), "hipEventRecord");
    s.in_flight = true; any_ = true;
    f->retain();
    return f.f;
}

DFrame* FrameStager::upload(uint32_t w, uint32_t h, uint8_t fmt, const uint8_t* const data[3], const int32_t stride[3]) {
    std::lock_guard<std::mutex> lk(mu_);
    uint8_t* dst[3]; int32_t dst_stride[3];
    const uint32_t ticket = acquire_locked(w, h, fmt, dst, dst_stride);
    const DFrame* f = slots_[ticket - 1].target.f;
    try {
        for (int p = 0; p < f->stored_planes(); ++p) {
            if (!data[p]) throw Error(MX_ERR_INVALID, "host plane pointer is NULL");
            if (stride[p] < (int32_t)f->stored_row_bytes(p)) throw Error(MX_ERR_INVALID, "host stride smaller than the plane width");
            const uint32_t rb = f->stored_row_bytes(p), rows = f->ph(p);
            if ((uint32_t)stride[p] == rb && (uint32_t)dst_stride[p] == rb) std::memcpy(dst[p], data[p], (size_t)rb * rows);
            else for (uint32_t y = 0; y < rows; ++y) std::memcpy(dst[p] + (size_t)y * dst_stride[p], data[p] + (size_t)y * (size_t)stride[p], rb);
        }
    } catch (...) {
        slots_[ticket - 1].held = false; slots_[ticket - 1].target = FrameRef();
        throw;
    }
    return commit_locked(ticket);
}

void FrameStager::fence(hipStream_t consumer) {
    std::lock_guard<std::mutex> lk(mu_);
    hip_check(hipSetDevice(device_), "hipSetDevice");
    consumer_ = consumer; have_consumer_ = true;
    if (any_) hip_check(hipStreamWaitEvent(consumer, last_, 0), "hipStreamWaitEvent(stager)");
}

void FrameStager::sync() { hip_check(hipSetDevice(device_), "hipSetDevice"); hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize(stager)"); }

}  // namespace mx
