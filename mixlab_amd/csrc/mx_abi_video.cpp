// mx_abi_video.cpp -- extern "C" entry points of the pixel path (include/mixlab_gpu.h, second half).
// Same fencing convention as mx_abi.cpp: catch everything, stash the message, return a status.
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "mx_engine.hpp"
#include "mx_video.hpp"

using mx::DFrame;
using mx::Error;
using mx::FrameRef;

// mx_dframe is opaque: an mx_dframe* is a DFrame* in disguise
static inline DFrame* D(mx_dframe* p) { return reinterpret_cast<DFrame*>(p); }
static inline const DFrame* D(const mx_dframe* p) { return reinterpret_cast<const DFrame*>(p); }
static inline mx_dframe* H(DFrame* p) { return reinterpret_cast<mx_dframe*>(p); }

struct mx_video_mixer { std::unique_ptr<mx::VideoMixer> m; };
struct mx_graph { std::unique_ptr<mx::Graph> g; };   // same layout as in mx_abi.cpp

extern "C" const char* mx_last_error(void);
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

static hipStream_t default_video_stream() {
    static std::once_flag once;
    static hipStream_t s = nullptr;
    std::call_once(once, [] { mx::hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); });
    return s;
}
static hipStream_t S(void* stream) {
    hipStream_t s = stream ? (hipStream_t)stream : default_video_stream();
    mx::flush_scales(s);   // every stateless entry point starts from a stream with no deferred scaler work
    return s;
}

extern "C" {

int mx_dframe_create(uint32_t width, uint32_t height, void* stream, mx_dframe** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        *out = H(DFrame::create(width, height, S(stream)));
    });
}
int mx_dframe_create_fmt(uint32_t width, uint32_t height, mx_pixfmt fmt, void* stream, mx_dframe** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        if (fmt == MX_PIXFMT_YUVA420P) { *out = H(DFrame::create(width, height, S(stream), MX_PIXFMT_YUV420P, true)); return; }   // yuv420p + a coverage plane (opaque until uploaded)
        REQUIRE((int)fmt >= 0 && (int)fmt <= (int)mx::DFrame::kLastFmt, "unknown pixel format");
        *out = H(DFrame::create(width, height, S(stream), (uint8_t)fmt));
    });
}
int mx_dframe_format(const mx_dframe* f, mx_pixfmt* fmt) {
    return guard([&] {
        REQUIRE(f && fmt, "NULL argument");
        *fmt = (D(f)->fmt == MX_PIXFMT_YUV420P && D(f)->with_alpha) ? MX_PIXFMT_YUVA420P : (mx_pixfmt)D(f)->fmt;
    });
}
static DFrame* alpha_frame(const mx_dframe* f) {
    REQUIRE(f, "frame is NULL");
    DFrame* d = const_cast<DFrame*>(D(f));
    if (!d->with_alpha) throw Error(MX_ERR_INVALID, "the frame has no coverage plane (create it as MX_PIXFMT_YUVA420P; packed RGBA carries its alpha in the pixels)");
    return d;
}
int mx_dframe_upload_alpha(mx_dframe* f, const uint8_t* host_alpha, int32_t stride, void* stream) {
    return guard([&] {
        DFrame* d = alpha_frame(f);
        REQUIRE(host_alpha && stride >= (int32_t)d->width, "host alpha plane is NULL or its stride smaller than the width");
        hipStream_t s = S(stream);
        d->ensure_pixels(s);
        mx::hip_check(hipMemcpy2DAsync(d->alpha, d->alpha_stride, host_alpha, (size_t)stride, d->width, d->height, hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(H2D alpha)");
        mx::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}
int mx_dframe_download_alpha(const mx_dframe* f, uint8_t* host_alpha, int32_t stride, void* stream) {
    return guard([&] {
        DFrame* d = alpha_frame(f);
        REQUIRE(host_alpha && stride >= (int32_t)d->width, "host alpha plane is NULL or its stride smaller than the width");
        hipStream_t s = S(stream);
        d->ensure_pixels(s);
        mx::flush_scales(s);
        mx::hip_check(hipMemcpy2DAsync(host_alpha, (size_t)stride, d->alpha, d->alpha_stride, d->width, d->height, hipMemcpyDeviceToHost, s), "hipMemcpy2DAsync(D2H alpha)");
        mx::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    });
}
int mx_dframe_alpha_plane(const mx_dframe* f, void** device_alpha, int32_t* stride) {
    return guard([&] {
        REQUIRE(f && device_alpha, "NULL argument");
        const DFrame* d = D(f);
        *device_alpha = d->alpha;                       // NULL: the frame carries none (or has no pixels yet)
        if (stride) *stride = (int32_t)d->alpha_stride;
    });
}
int mx_dframe_retain(mx_dframe* f) {
    return guard([&] { REQUIRE(f, "frame is NULL"); D(f)->retain(); });
}
void mx_dframe_release(mx_dframe* f) {
    (void)guard([&] { if (f) D(f)->release(); });
}

static void check_host_frame(const DFrame* d, const mx_frame* h) {
    REQUIRE(d && h, "NULL argument");
    if (h->width != d->width || h->height != d->height) throw Error(MX_ERR_INVALID, "host frame size differs from the device frame");
    for (int p = 0; p < d->stored_planes(); ++p) {
        REQUIRE(h->data[p], "host plane pointer is NULL");
        if (h->stride[p] < (int32_t)d->stored_row_bytes(p)) throw Error(MX_ERR_INVALID, "host stride smaller than the plane width");
    }
}
int mx_dframe_upload(mx_dframe* f, const mx_frame* host, void* stream) {
    return guard([&] {
        check_host_frame(D(f), host);
        DFrame* d = D(f);
        for (int p = 0; p < d->stored_planes(); ++p)
            mx::hip_check(hipMemcpy2DAsync(d->data[p], d->stride[p], host->data[p], (size_t)host->stride[p], d->stored_row_bytes(p), d->ph(p),
                                           hipMemcpyHostToDevice, S(stream)), "hipMemcpy2DAsync(H2D frame)");
        mx::hip_check(hipStreamSynchronize(S(stream)), "hipStreamSynchronize");
    });
}
int mx_dframe_download(const mx_dframe* f, mx_frame* host, void* stream) {
    return guard([&] {
        check_host_frame(D(f), host);
        const DFrame* d = D(f);
        for (int p = 0; p < d->stored_planes(); ++p)
            mx::hip_check(hipMemcpy2DAsync(host->data[p], (size_t)host->stride[p], d->data[p], d->stride[p], d->stored_row_bytes(p), d->ph(p),
                                           hipMemcpyDeviceToHost, S(stream)), "hipMemcpy2DAsync(D2H frame)");
        mx::hip_check(hipStreamSynchronize(S(stream)), "hipStreamSynchronize");
    });
}
int mx_dframe_planes(const mx_dframe* f, uint32_t* width, uint32_t* height, void* device_data[3], int32_t stride[3]) {
    return guard([&] {
        REQUIRE(f, "frame is NULL");
        const DFrame* d = D(f);
        if (width) *width = d->width;
        if (height) *height = d->height;
        for (int p = 0; p < 3; ++p) {
            const bool stored = p < d->stored_planes();
            if (device_data) device_data[p] = stored ? d->data[p] : nullptr;
            if (stride) stride[p] = stored ? (int32_t)d->stride[p] : 0;
        }
    });
}

int mx_video_blank(mx_dframe* f, void* stream) {
    return guard([&] {
        REQUIRE(f, "frame is NULL");
        DFrame* d = D(f);
        mx::launch_blank(d->data[0], d->plane_bytes[0], d->data[1], d->plane_bytes[1], d->data[2], d->plane_bytes[2], S(stream), d->blank_chroma());
        mx::hip_check(hipGetLastError(), "blank launch");
    });
}

int mx_video_crossfade(mx_dframe* out, const mx_dframe* a, const mx_dframe* b, double fader, void* stream) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        DFrame* o = D(out);
        const DFrame* fa = a ? D(a) : nullptr;
        const DFrame* fb = b ? D(b) : nullptr;
        for (const DFrame* x : {fa, fb, (const DFrame*)o})
            if (x && (x->width != o->width || x->height != o->height || x->fmt != MX_PIXFMT_YUV420P))
                throw Error(MX_ERR_INVALID, "cross-fade inputs must have the output's size, all yuv420p (video_mixer.rs:282-283)");
        FrameRef ra(const_cast<DFrame*>(fa), fa != nullptr), rb(const_cast<DFrame*>(fb), fb != nullptr);
        if (ra) ra->ensure_pixels(S(stream));
        if (rb) rb->ensure_pixels(S(stream));
        auto chain = mx::make_chain(ra, rb, mx::crossfade_factor(fader), S(stream));
        mx::ChainArgs ar;
        mx::fill_chain_sources(*chain, ar.src, ar.n_src, ar.fade, ar.v_is_a, ar.al, ar.alpha_mask);
        ar._pad1 = 0;
        for (int p = 0; p < 3; ++p) {
            ar.out[p] = o->data[p]; ar.out_stride[p] = o->stride[p];
            ar.chunks_per_row[p] = ((o->pw(p) + 31u) / 32u) * 2u;
            ar.chunks[p] = ar.chunks_per_row[p] * o->ph(p);
        }
        mx::launch_fade_chain(ar, S(stream));
        mx::hip_check(hipGetLastError(), "cross-fade launch");
    });
}

int mx_video_scale_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                            uint32_t* scaled_w, uint32_t* scaled_h, uint32_t* letterbox_x, uint32_t* letterbox_y) {
    return guard([&] {
        REQUIRE(in_w && in_h && out_w && out_h, "zero dimension");
        const mx::ScaleGeometry g = mx::scaler_geometry(in_w, in_h, out_w, out_h);
        if (scaled_w) *scaled_w = g.scaled_w;
        if (scaled_h) *scaled_h = g.scaled_h;
        if (letterbox_x) *letterbox_x = g.letterbox_x;
        if (letterbox_y) *letterbox_y = g.letterbox_y;
    });
}

int mx_video_scale(const mx_dframe* in, mx_dframe* out, void* stream) {
    return guard([&] {
        REQUIRE(in && out, "NULL argument");
        DFrame* o = D(out);
        REQUIRE(o->fmt == MX_PIXFMT_YUV420P, "the scaler's output picture is yuv420p");
        hipStream_t s = S(stream);
        mx::Scaler sc(o->width, o->height, s);
        FrameRef src(const_cast<DFrame*>(D(in)), true);
        FrameRef res = sc.scale(src);
        mx::CopyArgs c;
        for (int p = 0; p < 3; ++p) {
            c.src[p] = res->data[p]; c.dst[p] = o->data[p];
            c.src_stride[p] = res->stride[p]; c.dst_stride[p] = o->stride[p];
            c.rows[p] = o->ph(p); c.row_bytes[p] = o->pw(p);
        }
        mx::launch_copy_planes(c, s);
        if (o->alpha) {   // the output asked for the coverage plane too: the scaled one, or opaque when the input carries none
            if (res->alpha) mx::hip_check(hipMemcpy2DAsync(o->alpha, o->alpha_stride, res->alpha, res->alpha_stride, o->width, o->height, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync(alpha)");
            else mx::hip_check(hipMemsetAsync(o->alpha, 0xff, o->alpha_bytes, s), "hipMemsetAsync(alpha)");
        }
        mx::hip_check(hipGetLastError(), "scale launch");
        mx::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");   // the temporary scaler's frame dies here
    });
}

/* DynamicScaler (src/video/encode.rs:338-397): a scaler that keeps its context (tap tables, blank letterboxed output
 * frame) while the input settings stay the same -- the per-tick rescale of Monitor / StreamOutput (encode.rs:287-295). */
struct mx_video_scaler { std::unique_ptr<mx::Scaler> s; hipStream_t stream; };

int mx_video_scaler_create(uint32_t out_width, uint32_t out_height, void* stream, mx_video_scaler** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        REQUIRE(out_width && out_height && !(out_width & 1) && !(out_height & 1), "output size must be even and non-zero (yuv420p)");
        auto* h = new mx_video_scaler();
        h->stream = S(stream);
        h->s.reset(new mx::Scaler(out_width, out_height, h->stream));
        *out = h;
    });
}
int mx_video_scaler_scale(mx_video_scaler* sc, const mx_dframe* in, mx_dframe** out) {
    return guard([&] {
        REQUIRE(sc && in && out, "NULL argument");
        FrameRef src(const_cast<DFrame*>(D(in)), true);
        FrameRef res = sc->s->scale(src);           // the input itself when the settings are equal (encode.rs:342-345)
        mx::flush_scales(sc->stream);
        mx::hip_check(hipGetLastError(), "scale launch");
        res->retain();
        *out = H(res.f);
    });
}
void mx_video_scaler_destroy(mx_video_scaler* sc) {
    (void)guard([&] { if (sc) { (void)hipStreamSynchronize(sc->stream); delete sc; } });
}

int mx_stream_retired(void* stream) {
    return guard([&] {
        REQUIRE(stream, "stream is NULL (the library's own default video stream is never retired)");
        mx::flush_scales((hipStream_t)stream);
        mx::video_stream_retired((hipStream_t)stream);
    });
}

int mx_video_to_rgba(const mx_dframe* in, void* device_rgba, int32_t rgba_stride, const int32_t* matrix_q12, void* stream) {
    return guard([&] {
        REQUIRE(in && device_rgba, "NULL argument");
        const DFrame* d = D(in);
        REQUIRE(d->fmt == MX_PIXFMT_YUV420P, "YUV -> RGBA takes yuv420p (scale a frame of another format first)");
        if (rgba_stride < (int32_t)(d->width * 4) || (rgba_stride & 15) || ((uintptr_t)device_rgba & 15))
            throw Error(MX_ERR_INVALID, "rgba buffer must be 16-byte aligned with stride >= 4 * width, stride % 16 == 0");
        mx::RgbaArgs a;
        a.y = d->data[0]; a.u = d->data[1]; a.v = d->data[2]; a.rgba = (uint8_t*)device_rgba;
        a.y_stride = d->stride[0]; a.u_stride = d->stride[1]; a.v_stride = d->stride[2]; a.rgba_stride = (uint32_t)rgba_stride;
        a.width = d->width; a.height = d->height;
        a.use_matrix = matrix_q12 ? 1 : 0;
        for (int k = 0; k < 12; ++k) a.m[k] = matrix_q12 ? matrix_q12[k] : 0;
        mx::launch_yuv420_to_rgba(a, S(stream));
        mx::hip_check(hipGetLastError(), "yuv->rgba launch");
    });
}

int mx_video_sync(void* stream) {
    return guard([&] { mx::hip_check(hipStreamSynchronize(S(stream)), "hipStreamSynchronize"); });
}

int mx_video_mixer_create(const mx_video_mixer_params* params, uint32_t sample_rate, void* stream, mx_video_mixer** out) {
    return guard([&] {
        REQUIRE(params && out, "NULL argument");
        *out = nullptr;
        auto h = std::make_unique<mx_video_mixer>();
        h->m = std::make_unique<mx::VideoMixer>(*params, sample_rate, S(stream));   // NULL = the library's default video stream, like every stateless entry point: ordered with mx_dframe_upload / download on NULL
        *out = h.release();
    });
}
int mx_video_mixer_update(mx_video_mixer* m, const mx_video_mixer_params* params) {
    return guard([&] { REQUIRE(m && params, "NULL argument"); m->m->update(*params); });
}
int mx_video_mixer_run_tick(mx_video_mixer* m, uint64_t t, const mx_video_input inputs[4],
                            mx_dframe** out_program, mx_dframe** out_a, mx_dframe** out_b) {
    return guard([&] {
        REQUIRE(m && inputs, "NULL argument");
        mx::VideoInput in[4];
        for (int i = 0; i < 4; ++i) {
            in[i].frame = inputs[i].frame ? D(inputs[i].frame) : nullptr;
            if (in[i].frame) {
                in[i].duration_hint = mx::Rational::make(inputs[i].dur_num, inputs[i].dur_den);
                in[i].tick_offset = mx::Rational::make(inputs[i].off_num, inputs[i].off_den);
            }
        }
        FrameRef o, a, b;
        m->m->run_tick(t, in, o, a, b);
        mx::flush_scales(m->m->stream());
        // a mixer created without a stream works on one of its own, which no caller can order against: its frames are complete on return
        if (m->m->owns_stream()) mx::hip_check(hipStreamSynchronize(m->m->stream()), "hipStreamSynchronize");
        auto give = [](FrameRef& r, mx_dframe** dst) {
            if (!dst) return;
            *dst = nullptr;
            if (r) { r->retain(); *dst = H(r.f); }
        };
        give(o, out_program); give(a, out_a); give(b, out_b);
    });
}
int mx_video_mixer_sync(mx_video_mixer* m) {
    return guard([&] { REQUIRE(m, "mixer is NULL"); mx::flush_scales(m->m->stream()); mx::hip_check(hipStreamSynchronize(m->m->stream()), "hipStreamSynchronize"); });
}
void mx_video_mixer_destroy(mx_video_mixer* m) {
    (void)guard([&] { delete m; });
}

int mx_video_scale_band(const mx_dframe* in_slice, uint32_t in_full_h, uint32_t src_row0, mx_dframe* out_band,
                        uint32_t full_w, uint32_t full_h, uint32_t row0, void* stream) {
    return guard([&] {
        REQUIRE(in_slice && out_band, "NULL argument");
        mx::scale_band(D(in_slice), in_full_h, src_row0, D(out_band), full_w, full_h, row0, S(stream));
    });
}
uint32_t mx_video_scaler_tap_count(uint32_t src, uint32_t dst) { return (src && dst) ? mx::scaler_tap_count(src, dst) : 0u; }
int mx_video_scaler_taps(uint32_t src, uint32_t dst, int32_t* first, int32_t* coef, uint32_t* n_taps) {
    return guard([&] {
        REQUIRE(src && dst && first && coef && n_taps, "NULL or zero argument");
        REQUIRE(src <= 16384 && dst <= 16384, "plane too large");
        std::vector<int32_t> f, c;
        mx::scaler_taps(src, dst, f, c);
        *n_taps = mx::scaler_tap_count(src, dst);
        std::memcpy(first, f.data(), f.size() * sizeof(int32_t));
        std::memcpy(coef, c.data(), c.size() * sizeof(int32_t));
    });
}

int mx_graph_set_video_source(mx_graph* g, uint32_t node, mx_dframe* frame, int64_t dur_num, int64_t dur_den,
                              int64_t off_num, int64_t off_den, int repeat) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        g->g->set_video_source(node, frame ? D(frame) : nullptr, mx::Rational::make(dur_num, dur_den ? dur_den : 1),
                               mx::Rational::make(off_num, off_den ? off_den : 1), repeat != 0);
    });
}
int mx_graph_set_video_source_ring(mx_graph* g, uint32_t node, mx_dframe* const* frames, size_t n, int64_t dur_num, int64_t dur_den,
                                   int64_t off_num, int64_t off_den) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        REQUIRE(frames || !n, "frames is NULL");
        std::vector<mx::DFrame*> v(n);
        for (size_t i = 0; i < n; ++i) { REQUIRE(frames[i], "a frame of the ring is NULL"); v[i] = D(frames[i]); }
        g->g->set_video_source_ring(node, v.data(), n, mx::Rational::make(dur_num, dur_den ? dur_den : 1), mx::Rational::make(off_num, off_den ? off_den : 1));
    });
}
int mx_graph_set_video_source_band(mx_graph* g, uint32_t node, uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows,
                                   uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows) {
    return guard([&] {
        REQUIRE(g, "graph is NULL");
        g->g->set_video_source_band(node, in_w, in_full_h, src_row0, slice_rows, full_w, full_h, row0, band_rows);
    });
}
int mx_graph_video_output(mx_graph* g, uint32_t node, uint32_t port, mx_dframe** out) {
    return guard([&] {
        REQUIRE(g && out, "NULL argument");
        *out = nullptr;
        FrameRef r = g->g->video_output(node, port);
        if (r) { r->retain(); *out = H(r.f); }
    });
}
int mx_graph_rgba_output(mx_graph* g, uint32_t node, void** device_rgba, int32_t* stride, uint32_t* width, uint32_t* height) {
    return guard([&] { REQUIRE(g, "graph is NULL"); g->g->rgba_output(node, device_rgba, stride, width, height); });
}

int mx_device_alloc(size_t bytes, void** device_ptr) {
    return guard([&] {
        REQUIRE(device_ptr, "device_ptr is NULL");
        *device_ptr = nullptr;
        hipError_t e = hipMalloc(device_ptr, bytes ? bytes : 16);
        if (e == hipErrorOutOfMemory) throw Error(MX_ERR_NOMEM, "hipMalloc: out of device memory");
        mx::hip_check(e, "hipMalloc");
    });
}
void mx_device_free(void* device_ptr) {
    if (device_ptr) (void)hipFree(device_ptr);
}
int mx_device_download(void* host, const void* device_ptr, size_t bytes, void* stream) {
    return guard([&] {
        REQUIRE(host && device_ptr, "NULL argument");
        mx::hip_check(hipMemcpyAsync(host, device_ptr, bytes, hipMemcpyDeviceToHost, S(stream)), "hipMemcpyAsync(D2H)");
        mx::hip_check(hipStreamSynchronize(S(stream)), "hipStreamSynchronize");
    });
}

}  // extern "C"
