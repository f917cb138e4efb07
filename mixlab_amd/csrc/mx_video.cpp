// mx_video.cpp -- host logic of the pixel path: frames, DynamicScaler, VideoMixer.  See mx_video.hpp.
#include "mx_video.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

namespace mx {

// ---------------------------------------------------------------------------------------------
// Rational64 as used by MediaTime / MediaDuration (util/src/time.rs): reduced, positive denominator
// ---------------------------------------------------------------------------------------------
static int64_t gcd_i64(int64_t a, int64_t b) {
    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    return a ? a : 1;
}
Rational Rational::make(int64_t n, int64_t d) {
    if (d == 0) throw Error(MX_ERR_INVALID, "rational with zero denominator");
    if (d < 0) { n = -n; d = -d; }
    const int64_t g = gcd_i64(n, d);
    Rational r; r.num = n / g; r.den = d / g;
    return r;
}
Rational Rational::operator+(const Rational& o) const {
    const int64_t g = gcd_i64(den, o.den);
    const int64_t lcm = den / g * o.den;
    return make(num * (lcm / den) + o.num * (lcm / o.den), lcm);
}
Rational Rational::operator-(const Rational& o) const {
    Rational n = o; n.num = -n.num;
    return *this + n;
}
bool Rational::operator>=(const Rational& o) const {
    return (__int128)num * o.den >= (__int128)o.num * den;
}

// ---------------------------------------------------------------------------------------------
// deferred scaling: jobs queue per stream and leave as one batched launch when pixels are needed
// ---------------------------------------------------------------------------------------------
uint32_t video_batch_ticks() {
    static const uint32_t k = [] { const char* e = getenv("MX_VIDEO_BATCH"); const int v = e ? atoi(e) : 16; return (uint32_t)std::min(std::max(v, 1), (int)MX_VB_MAX_CHAINS); }();
    return k;
}
namespace {
struct PendingScales { std::vector<ScaleArgs> jobs; std::vector<FrameRef> keep; std::vector<std::shared_ptr<const ScaleTables>> keep_tabs; };
std::mutex g_scale_mu;
std::map<hipStream_t, PendingScales> g_scale_q;
void flush_locked(hipStream_t s, PendingScales& q) {
    if (!q.jobs.empty()) { launch_video_batch(q.jobs.data(), (int)q.jobs.size(), nullptr, 0, s); q.jobs.clear(); }
    q.keep.clear(); q.keep_tabs.clear();
}
}  // namespace
void queue_scale(const ScaleArgs& a, hipStream_t s, const FrameRef& src, const FrameRef& dst, std::shared_ptr<const ScaleTables> tabs, bool companion) {
    std::lock_guard<std::mutex> lk(g_scale_mu);
    PendingScales& q = g_scale_q[s];
    bool conflict = q.jobs.size() >= MX_VB_MAX_JOBS;
    if (!companion) for (const FrameRef& k : q.keep) if (k.f == src.f || k.f == dst.f) conflict = true;   // no ordering inside one launch
    if (conflict) flush_locked(s, q);
    q.jobs.push_back(a);
    q.keep.push_back(src); q.keep.push_back(dst);
    if (tabs) q.keep_tabs.push_back(std::move(tabs));   // the job reads these tables when it leaves
}
void flush_scales(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_scale_mu);
    auto it = g_scale_q.find(s);
    if (it != g_scale_q.end()) flush_locked(s, it->second);
}
// The queued scales of the stream leave TOGETHER with RGBA chains that do not read them (Graph::run_video_tick: the chains of earlier
// ticks) -- one launch instead of dependent ones (mx_k_video.hip k_video_batch).
void launch_chains_rgba_after_queued_scales(const ChainRgbaArgs* chains, int n_chains, hipStream_t s) {
    PendingScales taken;
    {
        std::lock_guard<std::mutex> lk(g_scale_mu);
        auto it = g_scale_q.find(s);
        if (it != g_scale_q.end()) { taken = std::move(it->second); it->second = PendingScales{}; }
    }
    launch_video_batch(taken.jobs.data(), (int)taken.jobs.size(), chains, n_chains, s);   // `taken` keeps the frames and tables until the launch is queued (a later free synchronises)
}

// ---------------------------------------------------------------------------------------------
// frames
// ---------------------------------------------------------------------------------------------
static void alloc_planes(DFrame* f) {
    size_t off[3], total = 0;
    const int np = f->stored_planes();
    for (int p = 0; p < np; ++p) {
        const uint32_t rb = f->stored_row_bytes(p), ph = f->ph(p);
        f->stride[p] = (rb + 63u) & ~63u;               // rows 64-byte aligned (the reference asserts 32, video_mixer.rs:196-201)
        f->plane_bytes[p] = (size_t)f->stride[p] * ph;
        off[p] = total;
        total += (f->plane_bytes[p] + 255) & ~(size_t)255;
    }
    size_t alpha_off = 0;
    if (f->with_alpha) {   // the coverage plane: one byte per luma sample, rows 64-byte aligned like the others
        f->alpha_stride = (f->width + 63u) & ~63u;
        f->alpha_bytes = (size_t)f->alpha_stride * f->height;
        alpha_off = total;
        total += (f->alpha_bytes + 255) & ~(size_t)255;
    }
    f->mem.alloc(total);
    for (int p = 0; p < np; ++p) f->data[p] = (uint8_t*)f->mem.p + off[p];
    if (f->with_alpha) f->alpha = (uint8_t*)f->mem.p + alpha_off;
    if (np == 2) { f->data[2] = f->data[1]; f->stride[2] = f->stride[1]; f->plane_bytes[2] = 0; }   // nv12: V = the odd bytes of the UV plane
}

DFrame* DFrame::create_unfilled(uint32_t w, uint32_t h, uint8_t fmt, bool alpha) {
    if (fmt > DFrame::kLastFmt) throw Error(MX_ERR_INVALID, "unknown pixel format");
    std::unique_ptr<DFrame> f(new DFrame());
    f->fmt = fmt;
    f->with_alpha = alpha;
    if (alpha && (f->packed() || f->deep() || f->semi())) throw Error(MX_ERR_INVALID, "a coverage plane goes with planar 8-bit YUV (yuva420p); four-byte packed RGB carries its own");
    if (w == 0 || h == 0 || (w & ((1u << f->cw()) - 1u)) || (h & ((1u << f->chs()) - 1u)))
        throw Error(MX_ERR_INVALID, "frame size must be non-zero and a multiple of the chroma subsampling (yuv420p: even)");
    if (w > 16384 || h > 16384) throw Error(MX_ERR_INVALID, "frame too large");
    f->width = w; f->height = h;
    alloc_planes(f.get());
    return f.release();
}

DFrame* DFrame::create(uint32_t w, uint32_t h, hipStream_t s, uint8_t fmt, bool alpha) {
    std::unique_ptr<DFrame> f(create_unfilled(w, h, fmt, alpha));
    launch_blank(f->data[0], f->plane_bytes[0], f->data[1], f->plane_bytes[1], f->data[2], f->plane_bytes[2], s, f->blank_chroma());
    if (f->alpha) hip_check(hipMemsetAsync(f->alpha, 0xff, f->alpha_bytes, s), "hipMemsetAsync(alpha)");   // a blank frame is opaque
    return f.release();
}

DFrame* DFrame::create_lazy(uint32_t w, uint32_t h, std::shared_ptr<LazyChain> c) {
    std::unique_ptr<DFrame> f(new DFrame());
    f->width = w; f->height = h;
    f->lazy = std::move(c);
    return f.release();
}

void fill_chain_sources(const LazyChain& c, ChainSrc (&src)[MX_CHAIN_MAX_SRC], uint32_t& n_src,
                        uint32_t (&fade)[MX_CHAIN_MAX_SRC - 1], uint32_t (&v_is_a)[MX_CHAIN_MAX_SRC - 1],
                        ChainAlpha (&al)[MX_CHAIN_MAX_SRC], uint32_t& alpha_mask) {
    alpha_mask = 0;
    auto put = [&](int k, const FrameRef& f) {
        for (int p = 0; p < 3; ++p) { src[k].p[p] = f ? f->data[p] : nullptr; src[k].stride[p] = f ? f->stride[p] : 0; }
        al[k].p = (f && f->alpha) ? f->alpha : nullptr; al[k].stride = al[k].p ? f->alpha_stride : 0; al[k]._pad = 0;
        if (f && f->with_alpha && !f->alpha) throw Error(MX_ERR_INTERNAL, "a frame with coverage reached a chain without its pixels");
        if (al[k].p) alpha_mask |= 1u << k;
    };
    if (c.steps.size() + 1 > MX_CHAIN_MAX_SRC) throw Error(MX_ERR_INTERNAL, "cross-fade chain too long");
    for (int k = 0; k < MX_CHAIN_MAX_SRC; ++k) put(k, FrameRef());
    put(0, c.base);
    n_src = 1 + (uint32_t)c.steps.size();
    for (size_t k = 0; k < c.steps.size(); ++k) {
        put((int)k + 1, c.steps[k].other);
        fade[k] = c.steps[k].fade; v_is_a[k] = c.steps[k].v_is_a ? 1u : 0u;
    }
    for (size_t k = c.steps.size(); k < MX_CHAIN_MAX_SRC - 1; ++k) { fade[k] = 0; v_is_a[k] = 1; }
}

DFrame* DFrame::create_lazy_scale(uint32_t w, uint32_t h, std::shared_ptr<LazyScale> sc) {
    std::unique_ptr<DFrame> f(new DFrame());
    f->width = w; f->height = h;
    f->with_alpha = sc->src && sc->src->carries_alpha() && sc->target && sc->target->alpha;   // known before the pixels exist
    f->lazy_scale = std::move(sc);
    return f.release();
}

static void chain_layers_need_pixels(const LazyChain& c, hipStream_t s) {   // k_fade_chain reads planes: unevaluated scaler outputs are computed first
    if (c.base && c.base->lazy_scale) c.base->ensure_pixels(s);
    for (const auto& st : c.steps) if (st.other && st.other->lazy_scale) st.other->ensure_pixels(s);
}

static void launch_chain_into(const LazyChain& c, DFrame* o, hipStream_t s) {
    chain_layers_need_pixels(c, s);
    ChainArgs a;
    fill_chain_sources(c, a.src, a.n_src, a.fade, a.v_is_a, a.al, a.alpha_mask);
    a._pad1 = 0;
    for (int p = 0; p < 3; ++p) {
        a.out[p] = o->data[p]; a.out_stride[p] = o->stride[p];
        a.chunks_per_row[p] = ((o->pw(p) + 31u) / 32u) * 2u;     // fade_line's 32-byte blocks (video_mixer.rs:219-234)
        a.chunks[p] = a.chunks_per_row[p] * o->ph(p);
    }
    launch_fade_chain(a, s);
    hip_check(hipGetLastError(), "cross-fade launch");
}

void DFrame::ensure_pixels(hipStream_t s) {
    if (lazy_scale) {   // the scaler's own output frame receives the pixels (encode.rs:386-396) and this frame becomes a view of it
        const std::shared_ptr<LazyScale> sc = std::move(lazy_scale);
        lazy_scale.reset();
        scale_into(sc->src, sc->t, sc->target, nullptr, s);
        alias = sc->target;
        for (int p = 0; p < 3; ++p) { data[p] = alias->data[p]; stride[p] = alias->stride[p]; plane_bytes[p] = alias->plane_bytes[p]; }
        if (with_alpha) { alpha = alias->alpha; alpha_stride = alias->alpha_stride; alpha_bytes = alias->alpha_bytes; }
        return;
    }
    if (!lazy) return;
    alloc_planes(this);
    launch_chain_into(*lazy, this, s);
    lazy.reset();
}

static void chain_scale_of(const LazyScale& sc, ChainScale& o) {
    const ScaleTables& t = *sc.t;
    for (int p = 0; p < 3; ++p) { o.src[p] = sc.src->data[p]; o.src_stride[p] = sc.src->stride[p]; }
    for (int c = 0; c < 2; ++c) {
        o.sw[c] = c ? t.in_w >> t.in_cw : t.in_w; o.sh[c] = c ? t.in_h >> t.in_ch : t.in_h;
        o.dw[c] = t.geo.scaled_w >> c; o.dh[c] = t.geo.scaled_h >> c;
        o.lx[c] = t.geo.letterbox_x >> c; o.ly[c] = t.geo.letterbox_y >> c;
        o.hfirst[c] = t.tab[c][0]; o.hpk[c] = t.hpk[c]; o.vfirst[c] = t.tab[c][2]; o.vpk[c] = t.vpk[c];
    }
}

void fill_chain_rgba_sources(const LazyChain& c, ChainRgbaArgs& a, hipStream_t s) {
    a.n_scaled = 0;
    for (uint32_t j = 0; j < MX_CHAIN_MAX_SCALED; ++j) { a.scaled_src[j] = 0xffffffffu; std::memset(&a.sc[j], 0, sizeof a.sc[j]); }
    // Measured: the resampling VALU work is the same wherever it runs and the fused kernel holds 2 waves per SIMD, so
    // resampling inside the chain kernel is SLOWER on MI355X (28.7 us per 1080p frame against 12.1 + 10.5 us for scaler + chain) -- kept
    // as an opt-in (MX_SCALE_INLINE=1, read per call so tests can switch it), bit-exact either way.
    const char* const inl = getenv("MX_SCALE_INLINE");
    const bool no_inline = !(inl && atoi(inl) != 0);
    bool any_alpha = c.base && c.base->with_alpha;
    for (const auto& st : c.steps) any_alpha = any_alpha || (st.other && st.other->with_alpha);
    auto take = [&](const FrameRef& f, uint32_t k) {
        if (!f || !f->lazy_scale) return;
        if (no_inline || any_alpha || a.n_scaled == MX_CHAIN_MAX_SCALED || !f->lazy_scale->t->four_tap || !f->lazy_scale->src->data[0] || f->lazy_scale->src->semi()) { f->ensure_pixels(s); return; }
        chain_scale_of(*f->lazy_scale, a.sc[a.n_scaled]);
        a.scaled_src[a.n_scaled++] = k;
    };
    take(c.base, 0);
    for (size_t k = 0; k < c.steps.size(); ++k) take(c.steps[k].other, (uint32_t)k + 1);
    fill_chain_sources(c, a.src, a.n_src, a.fade, a.v_is_a, a.al, a.alpha_mask);   // an inline-scaled layer has no planes: nullptr here, as a blank one
    a._pad1 = 0;
}

// (a, b, fade) -> chain; a lazy operand's own chain is extended instead of being evaluated
std::shared_ptr<LazyChain> make_chain(const FrameRef& a, const FrameRef& b, uint8_t fade, hipStream_t s) {
    auto c = std::make_shared<LazyChain>();
    if (a && a->lazy && b && b->lazy) b->ensure_pixels(s);                       // only one side may stay symbolic
    if (a && a->lazy && a->lazy->steps.size() + 2 > MX_CHAIN_MAX_SRC) a->ensure_pixels(s);
    if (b && b->lazy && b->lazy->steps.size() + 2 > MX_CHAIN_MAX_SRC) b->ensure_pixels(s);
    if (a && a->lazy) { *c = *a->lazy; c->steps.push_back({b, fade, true}); }          // v (the running composite) is A
    else if (b && b->lazy) { *c = *b->lazy; c->steps.push_back({a, fade, false}); }    // v is B
    else { c->base = a; c->steps.push_back({b, fade, true}); }
    return c;
}

// video_mixer.rs:276-297: max of each dimension, rounded UP to the chroma grid (yuv420p: even)
void unify_picture_settings(uint32_t aw, uint32_t ah, uint32_t bw, uint32_t bh, uint32_t& w, uint32_t& h) {
    const uint32_t width = std::max(aw, bw), height = std::max(ah, bh);
    w = (width + 1u) & ~1u;
    h = (height + 1u) & ~1u;
}

// encode.rs:354-374: min of the two exact ratios, scaled size truncated then aligned DOWN to even,
// letterbox offsets halved then aligned down to even
ScaleGeometry scaler_geometry(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h) {
    uint64_t num, den;
    if ((uint64_t)out_w * in_h <= (uint64_t)out_h * in_w) { num = out_w; den = in_w; } else { num = out_h; den = in_h; }
    ScaleGeometry g;
    g.scaled_w = (uint32_t)(num * in_w / den) & ~1u;
    g.scaled_h = (uint32_t)(num * in_h / den) & ~1u;
    g.letterbox_x = ((out_w - g.scaled_w) / 2) & ~1u;
    g.letterbox_y = ((out_h - g.scaled_h) / 2) & ~1u;
    return g;
}

// video_mixer.rs:168: `(fader * 255.0) as u8` -- saturating, truncating, NaN -> 0
uint8_t crossfade_factor(double fader) {
    const double v = fader * 255.0;
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------
// bicubic tap tables (build-specified scaler, DESIGN.md "Scaler")
// ---------------------------------------------------------------------------------------------
static int64_t floor_div(int64_t a, int64_t b) {   // b > 0
    int64_t q = a / b;
    if ((a % b) < 0) --q;
    return q;
}
static int32_t cubic_weight_q14(int64_t X) {   // |x| in Q16
    const int64_t one48 = (int64_t)1 << 48;
    int64_t num;
    if (X < 65536) num = 7 * X * X * X - 12 * 65536 * X * X + 5 * one48;                                             // (7x^3 - 12x^2 + 5) / 5
    else if (X < 131072) num = -3 * X * X * X + 15 * 65536 * X * X - 24 * ((int64_t)1 << 32) * X + 12 * one48;    // (-3x^3 + 15x^2 - 24x + 12) / 5
    else return 0;
    const int64_t den = 5 * ((int64_t)1 << 34);
    return (int32_t)floor_div(2 * num + den, 2 * den);
}
static uint32_t tap_count(uint32_t src, uint32_t dst) {   // DESIGN.md "Scaler": the kernel widens with the scale factor when downscaling
    if (src <= dst) return 4;
    return 2 * (uint32_t)((2 * (uint64_t)src + dst - 1) / dst) + 2;
}
static void make_taps(uint32_t src, uint32_t dst, std::vector<int32_t>& first, std::vector<int32_t>& coef) {
    const uint32_t n = tap_count(src, dst);
    first.resize(dst); coef.resize((size_t)dst * n);
    for (uint32_t o = 0; o < dst; ++o) {
        const int64_t pos = floor_div((2 * (int64_t)o + 1) * (int64_t)src * 65536, 2 * (int64_t)dst) - 32768;
        const int64_t ip = pos >> 16, d = pos & 0xffff;
        int32_t* c = &coef[(size_t)o * n];
        if (n == 4) {
            c[0] = cubic_weight_q14(65536 + d); c[1] = cubic_weight_q14(d); c[2] = cubic_weight_q14(65536 - d); c[3] = cubic_weight_q14(131072 - d);
            const int32_t resid = 16384 - (c[0] + c[1] + c[2] + c[3]);
            if (c[2] > c[1]) c[2] += resid; else c[1] += resid;
            first[o] = (int32_t)ip - 1;
            continue;
        }
        first[o] = (int32_t)ip - (int32_t)(n / 2) + 1;
        int64_t sum = 0;
        for (uint32_t k = 0; k < n; ++k) {
            int64_t dist = ((int64_t)first[o] + k) * 65536 - pos;
            if (dist < 0) dist = -dist;
            c[k] = cubic_weight_q14(floor_div(dist * (int64_t)dst, (int64_t)src));
            sum += c[k];
        }
        int64_t tot = 0; uint32_t best = 0;
        for (uint32_t k = 0; k < n; ++k) {
            c[k] = (int32_t)floor_div(2 * (int64_t)c[k] * 16384 + sum, 2 * sum);
            tot += c[k];
            if (c[k] > c[best]) best = k;
        }
        c[best] += (int32_t)(16384 - tot);
    }
}

uint32_t scaler_tap_count(uint32_t src, uint32_t dst) { return tap_count(src, dst); }
void scaler_taps(uint32_t src, uint32_t dst, std::vector<int32_t>& first, std::vector<int32_t>& coef) { make_taps(src, dst, first, coef); }

// Packed taps of the tiled / inline kernels (mx_k_video.hip): H {ch0..3 as i8 x 4, cl0..3 as i8 x 4} with c = 256 ch + cl; V {(c0, c1), (c2, c3)}
// as i16 x 2.  Every bound that arithmetic relies on is checked on the actual tables; false = these taps do not fit.
static bool pack_lean_taps(const std::vector<int32_t>& hc, const std::vector<int32_t>& vc, std::vector<int32_t>& hp, std::vector<int32_t>& vp) {
    bool ok = true;
    for (size_t o = 0; o < hc.size() / 4; ++o) {
        uint32_t hi = 0, lo = 0; int64_t pos = 0, neg = 0;
        for (int k = 0; k < 4; ++k) {
            const int32_t cc = hc[4 * o + k], ch = (cc + 128) >> 8, cl = cc - 256 * ch;
            if (ch < -128 || ch > 127) ok = false;
            hi |= (uint32_t)(ch & 0xff) << (8 * k); lo |= (uint32_t)(cl & 0xff) << (8 * k);
            (cc > 0 ? pos : neg) += cc;
        }
        if (((pos * 255 + 64) >> 7) - 16384 > 32767 || ((neg * 255 + 64) >> 7) - 16384 < -32768) ok = false;   // t - 16384 is an i16
        hp.push_back((int32_t)hi); hp.push_back((int32_t)lo);
    }
    for (size_t o = 0; o < vc.size() / 4; ++o) {
        int64_t mag = 0;
        for (int k = 0; k < 4; ++k) { const int32_t cc = vc[4 * o + k]; if (cc < -32768 || cc > 32767) ok = false; mag += cc < 0 ? -cc : cc; }
        if (mag * 32768 + ((int64_t)16384 * 16384 + (1 << 20)) > 0x7fffffffLL) ok = false;                    // the V accumulator stays in i32
        vp.push_back((int32_t)(((uint32_t)vc[4 * o] & 0xffffu) | ((uint32_t)vc[4 * o + 1] << 16)));
        vp.push_back((int32_t)(((uint32_t)vc[4 * o + 2] & 0xffffu) | ((uint32_t)vc[4 * o + 3] << 16)));
    }
    return ok;
}

// {packed taps, first tap index, 0} per output sample: what a lane of the tiled kernel fetches with one 16-byte load (ScalePlane::hx / vx)
static std::vector<int32_t> lean_entries(const std::vector<int32_t>& first, const std::vector<int32_t>& pk) {
    std::vector<int32_t> x(4 * first.size());
    for (size_t o = 0; o < first.size(); ++o) { x[4 * o] = pk[2 * o]; x[4 * o + 1] = pk[2 * o + 1]; x[4 * o + 2] = first[o]; x[4 * o + 3] = 0; }
    return x;
}

// Row band of DynamicScaler::scale.  The plan (tap tables, which rows of which plane) depends only on the geometry; run() binds frames.
BandScaler::BandScaler(uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows, uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows)
    : in_w_(in_w), src_row0_(src_row0), slice_rows_(slice_rows), full_w_(full_w), band_rows_(band_rows) {
    if (!in_w || !in_full_h || (in_w & 1) || (row0 & 1) || (src_row0 & 1) || (in_full_h & 1) || (full_h & 1) || (band_rows & 1) || (slice_rows & 1) || !band_rows || !slice_rows ||
        row0 + band_rows > full_h || src_row0 + slice_rows > in_full_h)
        throw Error(MX_ERR_INVALID, "band / slice rows must be whole chroma rows inside their pictures");
    geo_ = scaler_geometry(in_w, in_full_h, full_w, full_h);
    if (!geo_.scaled_w || !geo_.scaled_h) return;
    std::vector<int32_t> blob; size_t offs[2][4], pk_off[2][2] = {{0, 0}, {0, 0}}, x_off[2][2] = {{0, 0}, {0, 0}}; uint32_t taps[2][2];
    std::vector<int32_t> vf_host[2];
    tiled_ = true;                                  // every axis 4 taps and the taps fit the packed form: the batched tiled kernel takes the band
    for (int c = 0; c < 2; ++c) {
        std::vector<int32_t> hf, hc, vf, vc;
        make_taps(in_w >> c, geo_.scaled_w >> c, hf, hc);
        make_taps(in_full_h >> c, geo_.scaled_h >> c, vf, vc);
        taps[c][0] = tap_count(in_w >> c, geo_.scaled_w >> c); taps[c][1] = tap_count(in_full_h >> c, geo_.scaled_h >> c);
        vf_host[c] = vf;
        auto put = [&](const std::vector<int32_t>& v) { while (blob.size() & 3) blob.push_back(0); size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return o; };
        offs[c][0] = put(hf); offs[c][1] = put(hc); offs[c][2] = put(vf); offs[c][3] = put(vc);
        std::vector<int32_t> hp, vp;
        if (taps[c][0] != 4 || taps[c][1] != 4 || !pack_lean_taps(hc, vc, hp, vp) ||
            !scale_tile_origins_match(in_w >> c, geo_.scaled_w >> c, hf.data()) || !scale_tile_origins_match(in_full_h >> c, geo_.scaled_h >> c, vf.data()) ||
            !scale_tile_origins_match_m(in_w >> c, geo_.scaled_w >> c, hf.data()) || !scale_tile_origins_match_m(in_full_h >> c, geo_.scaled_h >> c, vf.data())) { tiled_ = false; continue; }
        pk_off[c][0] = put(hp); pk_off[c][1] = put(vp);
        x_off[c][0] = put(lean_entries(hf, hp)); x_off[c][1] = put(lean_entries(vf, vp));
    }
    needs_blank_ = geo_.scaled_w != full_w;         // letterbox bars inside the band (encode.rs:382): the blank fill is only paid for when there are any
    tabs_.alloc(blob.size() * sizeof(int32_t));
    hip_check(hipMemcpy(tabs_.p, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(band taps)");
    size_t tmp_off[3], tmp_total = 0;
    for (int p = 0; p < 3; ++p) { const int c = p ? 1 : 0; tmp_off[p] = tmp_total; tmp_total += (size_t)(geo_.scaled_w >> c) * (in_full_h >> c); }
    tmp_.alloc(tmp_total * sizeof(int32_t));
    for (int p = 0; p < 3; ++p) {
        const int c = p ? 1 : 0;
        ScalePlane& sp = plan_.p[p];
        sp = ScalePlane{};
        const uint32_t b0 = row0 >> c, b1 = (row0 + band_rows) >> c;                          // the band in this plane's rows
        const uint32_t s0 = geo_.letterbox_y >> c, s1 = (geo_.letterbox_y + geo_.scaled_h) >> c;   // the scaled picture's rows
        const uint32_t ra = std::max(b0, s0), rb = std::min(b1, s1);
        if (ra > b0 || rb < b1) needs_blank_ = true;
        if (ra >= rb) continue;                                                              // this band lies in the letterbox bars
        const uint32_t sh = in_full_h >> c, vn = taps[c][1];
        const int32_t lo = std::min<int32_t>(std::max<int32_t>(vf_host[c][ra - s0], 0), (int32_t)sh - 1);
        const int32_t hi = std::min<int32_t>(std::max<int32_t>(vf_host[c][rb - 1 - s0] + (int32_t)vn - 1, 0), (int32_t)sh - 1);
        const uint32_t sl0 = src_row0 >> c, sl1 = (src_row0 + slice_rows) >> c;
        if ((uint32_t)lo < sl0 || (uint32_t)hi >= sl1) throw Error(MX_ERR_INVALID, "the source slice lacks a row the band's vertical taps reach (see shard.band_source_rows)");
        sp.sw = in_w >> c; sp.sh = sh; sp.dw = geo_.scaled_w >> c; sp.dh = rb - ra;
        sp.hfirst = (const int32_t*)tabs_.p + offs[c][0]; sp.hcoef = (const int32_t*)tabs_.p + offs[c][1];
        sp.vfirst = (const int32_t*)tabs_.p + offs[c][2] + (ra - s0); sp.vcoef = (const int32_t*)tabs_.p + offs[c][3] + (size_t)(ra - s0) * vn;
        sp.hn = taps[c][0]; sp.vn = vn; sp.tmp = (int32_t*)tmp_.p + tmp_off[p];
        sp.h_row0 = (uint32_t)lo; sp.h_rows = (uint32_t)(hi - lo + 1);
        sp.dh_full = geo_.scaled_h >> c; sp.oy_base = ra - s0;                                                                // the band's first row of this plane, in rows of the scaled picture
        if (tiled_) {
            sp.hpk = reinterpret_cast<const uint2*>((const int32_t*)tabs_.p + pk_off[c][0]);
            sp.vpk = reinterpret_cast<const uint2*>((const int32_t*)tabs_.p + pk_off[c][1]) + (ra - s0);
            sp.hx = reinterpret_cast<const uint4*>((const int32_t*)tabs_.p + x_off[c][0]);
            sp.vx = reinterpret_cast<const uint4*>((const int32_t*)tabs_.p + x_off[c][1]) + (ra - s0);
            sp.mh = scale_origin_magic(geo_.scaled_w >> c); sp.mv = scale_origin_magic(geo_.scaled_h >> c);
        }
        dst_row_[p] = ra - b0;
        any_ = true;
    }
}

void BandScaler::run(const DFrame* slice, DFrame* out, hipStream_t s) {
    if (!slice || !out) throw Error(MX_ERR_INVALID, "NULL frame");
    if (slice->fmt != MX_PIXFMT_YUV420P || out->fmt != MX_PIXFMT_YUV420P) throw Error(MX_ERR_INVALID, "row bands are cut from yuv420p pictures");
    if (slice->with_alpha) throw Error(MX_ERR_INVALID, "a band-scaled layer cannot carry a coverage plane (feed layers with alpha at the picture's own size, cut to the band)");
    if (slice->width != in_w_ || slice->height != slice_rows_ || out->width != full_w_ || out->height != band_rows_ || !slice->data[0] || !out->data[0])
        throw Error(MX_ERR_INVALID, "frames do not have the sizes this band scaler was planned for");
    if (needs_blank_ || !any_) launch_blank(out->data[0], out->plane_bytes[0], out->data[1], out->plane_bytes[1], out->data[2], out->plane_bytes[2], s);   // AvFrame::blank: the letterbox bars
    if (!any_) return;
    ScaleArgs a = plan_;
    for (int p = 0; p < 3; ++p) {
        const int c = p ? 1 : 0;
        ScalePlane& sp = a.p[p];
        if (!sp.dw || !sp.dh) continue;
        sp.src = slice->data[p] - (ptrdiff_t)(src_row0_ >> c) * (ptrdiff_t)slice->stride[p];   // virtual base of the full plane: only rows of the slice are read
        sp.src_stride = slice->stride[p];
        sp.dst = out->data[p] + (size_t)dst_row_[p] * out->stride[p] + (geo_.letterbox_x >> c);
        sp.dst_stride = out->stride[p];
    }
    if (tiled_) {              // leaves with the other scales of this tick as one launch (flushed by whoever reads the pixels)
        queue_scale(a, s, FrameRef(const_cast<DFrame*>(slice), true), FrameRef(out, true));
        return;
    }
    launch_scale_wide(a, s);   // two passes through the row buffer, launched in stream order: one run at a time per BandScaler
    hip_check(hipGetLastError(), "band scale launch");
}

void scale_band(const DFrame* slice, uint32_t in_full_h, uint32_t src_row0, DFrame* out, uint32_t full_w, uint32_t full_h, uint32_t row0, hipStream_t s) {
    if (!slice || !out) throw Error(MX_ERR_INVALID, "NULL frame");
    if (out->width != full_w) throw Error(MX_ERR_INVALID, "a band frame is as wide as the full picture");
    BandScaler bs(slice->width, in_full_h, src_row0, slice->height, full_w, full_h, row0, out->height);
    bs.run(slice, out, s);
    flush_scales(s);
    hip_check(hipGetLastError(), "band scale launch");
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");   // the tables and the row buffer die here
}

void Scaler::retarget(uint32_t in_w, uint32_t in_h, uint8_t in_fmt, bool in_alpha) {
    in_w_ = in_w; in_h_ = in_h; in_fmt_ = in_fmt; in_alpha_ = in_alpha;
    auto t = std::make_shared<ScaleTables>();
    t->in_w = in_w; t->in_h = in_h; t->out_w = out_w_; t->out_h = out_h_;
    t->in_cw = DFrame::fmt_cw(in_fmt); t->in_ch = DFrame::fmt_ch(in_fmt);
    const uint32_t src_w[2] = {in_w, in_w >> t->in_cw}, src_h[2] = {in_h, in_h >> t->in_ch};   // the planes of the input format
    t->geo = scaler_geometry(in_w, in_h, out_w_, out_h_);
    const ScaleGeometry& geo = t->geo;
    flush_scales(stream_);             // queued jobs write the old output frames
    ring_.assign(2 * (size_t)video_batch_ticks(), FrameRef());
    for (auto& f : ring_) f = FrameRef(DFrame::create(out_w_, out_h_, stream_, MX_PIXFMT_YUV420P, in_alpha), false);   // AvFrame::blank(output_picture), encode.rs:382 (a coverage plane goes along with the picture)
    ring_pos_ = 0; frame_ = ring_[0];
    keep_pool_.clear();                // their borders belong to the old letterbox
    if (geo.scaled_w == 0 || geo.scaled_h == 0) {   // a picture so thin that its aligned scaled size has no rows or columns: scale() hands out the blank frame
        // (the reference passes the zero dimension to sws_getContext, gets NULL and panics, codec/src/ffmpeg/scale.rs:22-33)
        tmp_plane_[0] = tmp_plane_[1] = tmp_plane_[2] = nullptr;
        t_ = std::move(t);
        return;
    }
    // tap tables: [luma h, luma v, chroma h, chroma v]
    std::vector<int32_t> blob;
    size_t offs[2][4];
    bool four = geo.scaled_w && geo.scaled_h;
    size_t pk_off[2][2] = {{0, 0}, {0, 0}}, x_off[2][2] = {{0, 0}, {0, 0}};
    for (int c = 0; c < 2; ++c) {
        std::vector<int32_t> hf, hc, vf, vc;
        make_taps(src_w[c], geo.scaled_w >> c, hf, hc);
        make_taps(src_h[c], geo.scaled_h >> c, vf, vc);
        t->taps[c][0] = tap_count(src_w[c], geo.scaled_w >> c); t->taps[c][1] = tap_count(src_h[c], geo.scaled_h >> c);
        auto put = [&](const std::vector<int32_t>& v) { while (blob.size() & 3) blob.push_back(0); size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return o; };
        offs[c][0] = put(hf); offs[c][1] = put(hc); offs[c][2] = put(vf); offs[c][3] = put(vc);
        if (t->taps[c][0] != 4 || t->taps[c][1] != 4) { four = false; continue; }
        if (!scale_tile_origins_match(src_w[c], geo.scaled_w >> c, hf.data()) || !scale_tile_origins_match(src_h[c], geo.scaled_h >> c, vf.data()) ||
            !scale_tile_origins_match_m(src_w[c], geo.scaled_w >> c, hf.data()) || !scale_tile_origins_match_m(src_h[c], geo.scaled_h >> c, vf.data()))
            throw Error(MX_ERR_INTERNAL, "scaler: window-origin formula disagrees with the tap table");
        std::vector<int32_t> hp, vp;
        if (!pack_lean_taps(hc, vc, hp, vp)) four = false;
        pk_off[c][0] = put(hp); pk_off[c][1] = put(vp);
        x_off[c][0] = put(lean_entries(hf, hp)); x_off[c][1] = put(lean_entries(vf, vp));
    }
    t->four_tap = four;
    t->tabs.alloc(blob.size() * sizeof(int32_t));   // the previous tables stay alive with whoever still holds them (frames, queued jobs)
    hip_check(hipMemcpy(t->tabs.p, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(taps)");
    for (int c = 0; c < 2; ++c) for (int k = 0; k < 4; ++k) t->tab[c][k] = (const int32_t*)t->tabs.p + offs[c][k];
    if (four) for (int c = 0; c < 2; ++c) {
        t->hpk[c] = reinterpret_cast<const uint2*>((const int32_t*)t->tabs.p + pk_off[c][0]);
        t->vpk[c] = reinterpret_cast<const uint2*>((const int32_t*)t->tabs.p + pk_off[c][1]);
        t->hx[c] = reinterpret_cast<const uint4*>((const int32_t*)t->tabs.p + x_off[c][0]);
        t->vx[c] = reinterpret_cast<const uint4*>((const int32_t*)t->tabs.p + x_off[c][1]);
        t->mh[c] = scale_origin_magic(geo.scaled_w >> c); t->mv[c] = scale_origin_magic(geo.scaled_h >> c);
    }
    // downscaling on any axis: the two-pass path keeps the H-filtered rows of each plane in device memory
    const bool wide = t->taps[0][0] > 4 || t->taps[0][1] > 4 || t->taps[1][0] > 4 || t->taps[1][1] > 4;
    tmp_plane_[0] = tmp_plane_[1] = tmp_plane_[2] = nullptr;
    if (wide) {
        hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");   // a previous widened scale may still use the row buffer
        size_t off[3], total = 0;
        for (int p = 0; p < 3; ++p) { const int c = p ? 1 : 0; off[p] = total; total += (size_t)(geo.scaled_w >> c) * src_h[c]; }
        tmp_.alloc(total * sizeof(int32_t));
        for (int p = 0; p < 3; ++p) tmp_plane_[p] = (int32_t*)tmp_.p + off[p];
    }
    t_ = std::move(t);
}

void scale_into(const FrameRef& in, const std::shared_ptr<const ScaleTables>& tp, const FrameRef& target, int32_t* const tmp_plane[3], hipStream_t s) {
    const ScaleTables& t = *tp;
    ScaleArgs a{};
    for (int p = 0; p < 3; ++p) {
        const int c = p ? 1 : 0;
        ScalePlane& sp = a.p[p];
        sp.src = in->data[p]; sp.src_stride = in->stride[p]; sp.sw = in->pw(p); sp.sh = in->ph(p);
        // sub-frame view at the letterbox offset (frame.rs:253-278)
        sp.dst = target->data[p] + (size_t)(t.geo.letterbox_y >> c) * target->stride[p] + (t.geo.letterbox_x >> c);
        sp.dst_stride = target->stride[p]; sp.dw = t.geo.scaled_w >> c; sp.dh = t.geo.scaled_h >> c;
        sp.hfirst = t.tab[c][0]; sp.hcoef = t.tab[c][1]; sp.vfirst = t.tab[c][2]; sp.vcoef = t.tab[c][3];
        sp.hn = t.taps[c][0]; sp.vn = t.taps[c][1]; sp.tmp = tmp_plane ? tmp_plane[p] : nullptr;
        sp.h_row0 = 0; sp.h_rows = sp.sh;
        sp.hpk = t.hpk[c]; sp.vpk = t.vpk[c]; sp.oy_base = 0; sp.dh_full = 0;
        sp.hx = t.hx[c]; sp.vx = t.vx[c]; sp.mh = t.mh[c]; sp.mv = t.mv[c];
        sp.sxs = in->xstep(p) - 1u; sp.sxo = in->xoff(p);
    }
    // BUILD-SPECIFIED: the coverage plane is resampled like the luma plane (its tables, its passes) into the same letterboxed rectangle; the bars keep the
    // 255 the blank target was created with.  A companion job whose chroma slots are empty.
    ScaleArgs al{};
    const bool with_alpha = in->alpha && target->alpha;
    if (with_alpha) {
        al.p[0] = a.p[0];
        al.p[0].src = in->alpha; al.p[0].src_stride = in->alpha_stride;
        al.p[0].dst = target->alpha + (size_t)t.geo.letterbox_y * target->alpha_stride + t.geo.letterbox_x; al.p[0].dst_stride = target->alpha_stride;
        al.p[0].sxs = 0; al.p[0].sxo = 0;
    }
    if (tmp_plane && tmp_plane[0]) {        // widened kernel (downscale): two passes, launched in stream order
        flush_scales(s);
        launch_scale_wide(a, s);
        if (with_alpha) launch_scale_wide(al, s);   // after the luma passes in stream order: the row buffer is free again
        return;
    }
    queue_scale(a, s, in, target, tp);      // leaves with the other scales of this tick as one launch
    if (with_alpha) queue_scale(al, s, in, target, tp, true);
}

// a packed RGB (or gray8) input is first turned into the yuv444p frame it stands for, a 10-bit one into the 8-bit frame of its layout (build-specified
// conversions), into a frame nobody else holds
FrameRef Scaler::planar_of(const FrameRef& in) {
    if (!in->packed() && !in->deep()) return in;
    const DFrame::Deep* deep = DFrame::deep_of(in->fmt);
    const uint8_t as_fmt = deep ? deep->layout : (in->yuyv() ? (uint8_t)MX_PIXFMT_YUV422P : (uint8_t)MX_PIXFMT_YUV444P);
    const bool want_alpha = DFrame::rgb_of(in->fmt).bpp == 4;   // the A byte is the pixel's coverage (straight alpha): it travels as the planar frame's coverage plane
    FrameRef out;
    for (auto& f : rgb_pool_) if (f->width == in->width && f->height == in->height && f->fmt == as_fmt && f->with_alpha == want_alpha && f->rc.load(std::memory_order_acquire) == 1) { out = f; break; }
    if (!out) {
        if (rgb_pool_.size() >= 2 * (size_t)video_batch_ticks() + 2) rgb_pool_.erase(rgb_pool_.begin());
        rgb_pool_.push_back(FrameRef(DFrame::create(in->width, in->height, stream_, as_fmt, want_alpha), false));
        out = rgb_pool_.back();
    }
    if (deep) {
        DeepArgs a{};
        for (int p = 0; p < 3; ++p) {
            a.src[p] = in->data[p]; a.dst[p] = out->data[p]; a.src_stride[p] = in->stride[p]; a.dst_stride[p] = out->stride[p];
            a.w[p] = out->pw(p); a.h[p] = out->ph(p); a.xstep[p] = in->xstep(p); a.xoff[p] = in->xoff(p);
        }
        a.shift = deep->shift; a.bits = deep->bits;
        launch_deep_to_8(a, stream_);
        return out;
    }
    if (in->yuyv()) {
        launch_yuyv_to_422p(in->data[0], in->stride[0], in->width, in->height, in->fmt == MX_PIXFMT_YUYV422 ? 1u : 0u, out->data, out->stride, stream_);
        return out;
    }
    if (in->fmt == MX_PIXFMT_GRAY8) {   // luma as it is, U = V = 0x80 (what swscale's gray -> yuv gives; build-specified like the RGB matrix)
        hip_check(hipMemcpy2DAsync(out->data[0], out->stride[0], in->data[0], in->stride[0], in->width, in->height, hipMemcpyDeviceToDevice, stream_), "hipMemcpy2DAsync(gray8)");
        // the pool frame may last have held a packed-RGB picture of this size (same yuv444p class): its chroma planes are that picture's, not 0x80
        for (int p = 1; p < 3; ++p)
            hip_check(hipMemset2DAsync(out->data[p], out->stride[p], 0x80, out->pw(p), out->ph(p), stream_), "hipMemset2DAsync(gray8 chroma)");
        return out;
    }
    const DFrame::Rgb rgb = DFrame::rgb_of(in->fmt);
    launch_rgb_to_yuv444(in->data[0], in->stride[0], in->width, in->height, rgb.bpp, rgb.r, rgb.g, rgb.b, out->data, out->stride, stream_,
                         out->alpha, out->alpha_stride, 6u - rgb.r - rgb.g - rgb.b);   // the byte of a four-byte pixel that is none of R, G, B
    return out;
}

FrameRef Scaler::scale(const FrameRef& in0, bool may_defer) {
    if (in0->width == out_w_ && in0->height == out_h_ && in0->fmt == MX_PIXFMT_YUV420P) return in0;   // equal picture settings, encode.rs:342-345
    in0->ensure_pixels(stream_);                                                 // a symbolic frame must exist before it can be resampled
    const FrameRef in = planar_of(in0);
    if (in.f != in0.f && in->width == out_w_ && in->height == out_h_ && in->fmt == MX_PIXFMT_YUV420P) return in;   // a 10-bit 4:2:0 picture of the output's size: its 8-bit frame is the result
    if (!frame_ || in_w_ != in->width || in_h_ != in->height || in_fmt_ != in->fmt || in_alpha_ != in->with_alpha) retarget(in->width, in->height, in->fmt, in->with_alpha);   // encode.rs:347-384
    if (t_->geo.scaled_w == 0 || t_->geo.scaled_h == 0) return frame_;
    ring_pos_ = (ring_pos_ + 1) % (uint32_t)ring_.size(); frame_ = ring_[ring_pos_];   // not a frame the last 2K - 1 calls wrote: the RGBA chains that read those may be launched AFTER this scale (Graph defers them)
    if (may_defer && t_->four_tap) {
        auto sc = std::make_shared<LazyScale>();
        sc->src = in; sc->t = t_; sc->target = frame_;
        return FrameRef(DFrame::create_lazy_scale(out_w_, out_h_, std::move(sc)), false);
    }
    scale_into(in, t_, frame_, tmp_plane_, stream_);
    return frame_;
}

FrameRef Scaler::scale_keep(const FrameRef& in0) {
    if (in0->width == out_w_ && in0->height == out_h_ && in0->fmt == MX_PIXFMT_YUV420P) { in0->ensure_pixels(stream_); return in0; }
    in0->ensure_pixels(stream_);
    const FrameRef in = planar_of(in0);
    if (in.f != in0.f && in->width == out_w_ && in->height == out_h_ && in->fmt == MX_PIXFMT_YUV420P) return in;
    if (!frame_ || in_w_ != in->width || in_h_ != in->height || in_fmt_ != in->fmt || in_alpha_ != in->with_alpha) retarget(in->width, in->height, in->fmt, in->with_alpha);
    FrameRef out;
    // (a layer that carries coverage keeps it through this path too: pool frames are made like retarget()'s ring frames, with a coverage plane when the input has one)
    for (auto& f : keep_pool_) if (f->with_alpha == in_alpha_ && f->rc.load(std::memory_order_acquire) == 1) { out = f; break; }   // only the pool holds it
    if (!out) { keep_pool_.push_back(FrameRef(DFrame::create(out_w_, out_h_, stream_, MX_PIXFMT_YUV420P, in_alpha_), false)); out = keep_pool_.back(); }
    if (t_->geo.scaled_w == 0 || t_->geo.scaled_h == 0) return out;
    scale_into(in, t_, out, tmp_plane_, stream_);
    return out;
}

// ---------------------------------------------------------------------------------------------
// VideoMixer::run_tick (src/module/video_mixer.rs:70-250)
// ---------------------------------------------------------------------------------------------
VideoMixer::VideoMixer(const mx_video_mixer_params& p, uint32_t sample_rate, hipStream_t s)
    : params_(p), sample_rate_(sample_rate ? sample_rate : 44100u), stream_(s) {
    if (!stream_) { hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate"); own_stream_ = true; }
}
VideoMixer::~VideoMixer() {
    flush_scales(stream_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& c : ch_) { c.stored.frame = FrameRef(); c.scaler.reset(); }
    pool_.clear();
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

void VideoMixer::rebind(hipStream_t s, bool lazy_program, uint32_t ticks_per_second) {
    if (!s) throw Error(MX_ERR_INVALID, "VideoMixer::rebind needs a stream");
    for (auto& c : ch_) if (c.has_stored && c.stored.frame) c.stored.frame->ensure_pixels(stream_);   // a symbolic frame of the old graph must exist before that graph goes
    flush_scales(stream_);
    if (stream_) hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");   // everything queued on the old stream has run
    for (auto& c : ch_) if (c.scaler) c.scaler->rebind(s);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
    own_stream_ = false;
    stream_ = s;
    set_lazy_program(lazy_program, ticks_per_second);
}

FrameRef VideoMixer::fresh_output(uint32_t w, uint32_t h) {
    for (auto& f : pool_)
        if (f->width == w && f->height == h && f->rc.load(std::memory_order_acquire) == 1) return f;   // only the pool holds it
    if (pool_.size() >= 8) pool_.erase(pool_.begin());
    pool_.push_back(FrameRef(DFrame::create(w, h, stream_), false));
    return pool_.back();
}

void VideoMixer::rescale(Channel& ch, uint32_t tw, uint32_t th) {   // Channel::rescale, video_mixer.rs:261-274
    if (!ch.scaler || ch.scaler->out_w() != tw || ch.scaler->out_h() != th) {
        ch.scaler.reset(new Scaler(tw, th, stream_));
        if (ch.has_stored) ch.stored.frame = ch.scaler->scale(ch.stored.frame, lazy_program_);
    }
}

void VideoMixer::run_tick(uint64_t t, const VideoInput in[4], FrameRef& out, FrameRef& out_a, FrameRef& out_b) {
    out = FrameRef(); out_a = FrameRef(); out_b = FrameRef();
    // channel-specific pass-through outputs (video_mixer.rs:80-90)
    if (params_.a >= 0 && params_.a < 4 && in[params_.a].frame) out_a = FrameRef(in[params_.a].frame, true);
    if (params_.b >= 0 && params_.b < 4 && in[params_.b].frame) out_b = FrameRef(in[params_.b].frame, true);

    const Rational now = Rational::make((int64_t)t, (int64_t)sample_rate_);   // video_mixer.rs:92
    for (auto& c : ch_)                                                         // expire stored frames, :94-101
        if (c.has_stored && now >= c.stored.active_until) { c.has_stored = false; c.stored.frame = FrameRef(); }

    // compatible output picture settings (:104-119)
    bool have = false; uint32_t tw = 0, th = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t w, h;
        if (in[i].frame) { w = in[i].frame->width; h = in[i].frame->height; }
        else if (ch_[i].has_stored) { w = ch_[i].stored.frame->width; h = ch_[i].stored.frame->height; }
        else continue;
        if (!have) { tw = w; th = h; have = true; }            // fold1: a single item passes through un-unified
        else unify_picture_settings(tw, th, w, h, tw, th);
    }
    if (!have) return;                                         // :113-119, output stays None

    // receive new input frames (:122-148)
    for (int i = 0; i < 4; ++i) {
        Channel& c = ch_[i];
        if (in[i].frame) {
            c.has_stored = false; c.stored.frame = FrameRef();
            rescale(c, tw, th);
            FrameRef f(in[i].frame, true);                     // video.data.decoded.clone()
            if (f->lazy) {   // a symbolic frame is only valid inside this tick: keep it symbolic only if it expires by the next one
                const Rational life = in[i].tick_offset + in[i].duration_hint, one_tick = Rational::make(1, (int64_t)tps_);
                if (!(one_tick >= life)) f->ensure_pixels(stream_);
            }
            c.stored.frame = c.scaler->scale(f, lazy_program_);
            c.stored.active_until = now + in[i].tick_offset + in[i].duration_hint;
            c.has_stored = true;
        } else {
            rescale(c, tw, th);
        }
    }

    // compose (:150-239); the blank fill (:151) is folded into the cross-fade kernel
    FrameRef fa = (params_.a >= 0 && params_.a < 4 && ch_[params_.a].has_stored) ? ch_[params_.a].stored.frame : FrameRef();
    FrameRef fb = (params_.b >= 0 && params_.b < 4 && ch_[params_.b].has_stored) ? ch_[params_.b].stored.frame : FrameRef();
    std::shared_ptr<LazyChain> chain = make_chain(fa, fb, crossfade_factor(params_.fader), stream_);
    FrameRef o;
    if (lazy_program_) {
        o = FrameRef(DFrame::create_lazy(tw, th, chain), false);   // pixels are computed by the consumer, fused with its own work
    } else {
        o = fresh_output(tw, th);
        launch_chain_into(*chain, o.f, stream_);
    }
    out = o;
}

}  // namespace mx
