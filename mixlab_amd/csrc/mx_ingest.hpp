// mx_ingest.hpp -- the two modules that bring timed media into the engine, as host state machines over device frames:
// MediaSource::run_tick (src/module/media_source.rs:93-126) and StreamInput::run_tick (src/module/stream_input.rs:72-147),
// plus the page-locked staging ring decoded frames cross PCIe through (the reference hands them over through rings too,
// src/source.rs:97-98).  No pixel is touched here: frames are retained and released, timestamps are exact rationals
// (util/src/time.rs:10-75).
#pragma once
#include <deque>
#include <mutex>
#include <vector>

#include "mx_video.hpp"

namespace mx {

struct TickVideo { FrameRef frame; Rational duration_hint; Rational tick_offset; };   // engine::VideoFrame (io.rs:12-17); frame empty = None

class MediaSource {
public:
    MediaSource(uint32_t sample_rate, uint32_t ticks_per_second);
    // receive_event(SetMedia(..)) (:85-91): open_media's fresh OpenMedia -- empty channel, no epoch, empty buffer (:140-147) -- or None
    void set_media(bool present);
    // the decode thread's tx.send on sync_channel(2) (:140, :271).  false = two frames are waiting (the reference blocks there)
    bool send(DFrame* frame, Rational pts, Rational duration_hint);
    TickVideo run_tick(uint64_t t);
    size_t buffered() const { return buffer_.size(); }
    uint32_t sample_rate() const { return sr_; }
    uint32_t ticks_per_second() const { return tps_; }
private:
    struct Timed { FrameRef frame; Rational pts, dur; };
    uint32_t sr_, tps_;
    std::mutex mu_;                      // the channel is the only state the decode thread shares
    bool present_ = false;
    std::deque<Timed> chan_;
    bool have_epoch_ = false; Rational epoch_;
    std::deque<Timed> buffer_;
};

class StreamInput {
public:
    static constexpr size_t RING_FRAMES = 65536;   // RingBuffer::new(65536), src/source.rs:97-98
    explicit StreamInput(uint32_t sample_rate);
    // SourceSend::write_audio / write_video (src/source.rs:158-190): false = the ring is full or nobody listens
    bool write_audio(uint64_t source_id, Rational source_time, const int16_t* interleaved, size_t n_samples);
    bool write_video(uint64_t source_id, Rational source_time, DFrame* frame, Rational duration_hint);
    // StreamInput::update re-listening on a mountpoint change (stream_input.rs:57-70): the rings are replaced (or gone, listening =
    // false), the held audio / video frame and the source timing stay
    void listen(bool listening);
    // one run_tick: audio_out[n_out] interleaved i16 (what convert_sample is applied to, :167-173), zero where the queue ran dry
    TickVideo run_tick(uint64_t t, int16_t* audio_out, size_t n_out, size_t* zero_filled);
private:
    struct AudioFrame { uint64_t source_id; Rational source_time; std::vector<int16_t> data; size_t head = 0; };
    struct VideoFrameIn { uint64_t source_id; Rational source_time; FrameRef frame; Rational dur; };
    uint32_t sr_;
    std::mutex mu_;
    bool listening_ = true;
    std::deque<AudioFrame> audio_rx_; std::deque<VideoFrameIn> video_rx_;
    bool have_audio_frame_ = false; AudioFrame audio_frame_;   // self.audio_frame
    bool have_video_frame_ = false; VideoFrameIn video_frame_; // self.video_frame
    bool have_source_ = false; uint64_t source_id_ = 0; Rational source_epoch_;   // self.source (SourceTiming)
};

// Page-locked staging ring for decoded frames: rows are packed into a slot laid out like the device frame (strides, blank padding),
// and the slot crosses PCIe as ONE asynchronous copy on the stager's own stream, beside whatever the consumer stream is doing.
// Thread-safe: a decode thread may acquire / commit / upload while the engine thread fences.
class FrameStager {
public:
    explicit FrameStager(uint32_t slots);
    ~FrameStager();
    DFrame* upload(uint32_t w, uint32_t h, uint8_t fmt, const uint8_t* const data[3], const int32_t stride[3]);   // one reference for the caller
    // The copy-free form: a decoder writes its picture straight into a slot (acquire hands out the slot's plane pointers, laid out like
    // the device frame), commit sends it.  Several slots may be held at once (reference pictures); each commit is one async copy.
    uint32_t acquire(uint32_t w, uint32_t h, uint8_t fmt, uint8_t* data[3], int32_t stride[3]);   // -> ticket
    DFrame* commit(uint32_t ticket);
    void fence(hipStream_t consumer);   // `consumer` waits for every upload issued so far
    void sync();
private:
    struct Slot { uint8_t* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool in_flight = false, held = false; FrameRef target; size_t padded_for = 0; uint32_t pw = 0, ph = 0; uint8_t pfmt = 0; };
    FrameRef take_frame(uint32_t w, uint32_t h, uint8_t fmt);
    uint32_t acquire_locked(uint32_t w, uint32_t h, uint8_t fmt, uint8_t* data[3], int32_t stride[3]);
    DFrame* commit_locked(uint32_t ticket);
    int device_ = 0;                    // the creator's HIP device (the current device is per thread)
    std::mutex mu_;                     // the decode thread acquires / commits, the engine thread fences
    std::vector<Slot> slots_; uint32_t next_ = 0;
    std::vector<FrameRef> pool_;        // device frames handed out before: one nobody else holds any more is written again
    hipStream_t stream_ = nullptr; hipEvent_t last_ = nullptr; bool any_ = false;
    hipStream_t consumer_ = nullptr; bool have_consumer_ = false; hipEvent_t reuse_ = nullptr;   // the stream fence() was last called for: a pooled frame is rewritten only after what that stream has queued
};

}  // namespace mx
