// mx_engine.cpp -- device-resident graph executor.  See mx_engine.hpp.
#include <chrono>

#include "mx_engine.hpp"

#include <cstdio>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mx {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(MX_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

void DevBuf::alloc(size_t n) {
    free_();
    if (n == 0) n = 256;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipErrorOutOfMemory) { p = nullptr; throw Error(MX_ERR_NOMEM, "hipMalloc: out of device memory"); }
    hip_check(e, "hipMalloc");
    bytes = n;
}
void DevBuf::free_() {
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
}

// ModuleT::inputs()/outputs() of each kind (reference file:line in include/mixlab_gpu.h)
static void kind_ports(uint32_t kind, size_t params_len, std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
    auto need = [&](size_t sz, const char* name) {
        if (params_len != sz) throw Error(MX_ERR_INVALID, std::string("params_len does not match ") + name);
    };
    switch (kind) {
    case MX_KIND_AMPLIFIER: need(sizeof(mx_amplifier_params), "mx_amplifier_params"); in = {MX_STEREO, MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_ENVELOPE: need(sizeof(mx_envelope_params), "mx_envelope_params"); in = {MX_MONO}; out = {MX_MONO}; break;
    case MX_KIND_EQ_THREE: need(sizeof(mx_eq_three_params), "mx_eq_three_params"); in = {MX_MONO}; out = {MX_MONO}; break;
    case MX_KIND_FM_SINE: need(sizeof(mx_fm_sine_params), "mx_fm_sine_params"); in = {MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_MIXER: {
        if (params_len % sizeof(mx_mixer_channel_params)) throw Error(MX_ERR_INVALID, "mixer params_len is not a multiple of mx_mixer_channel_params");
        in.assign(params_len / sizeof(mx_mixer_channel_params), MX_STEREO); out = {MX_STEREO, MX_STEREO}; break;
    }
    case MX_KIND_OSCILLATOR: need(sizeof(mx_oscillator_params), "mx_oscillator_params"); in = {}; out = {MX_MONO, MX_STEREO}; break;
    case MX_KIND_PLOTTER: need(0, "() (Plotter has no params)"); in = {MX_STEREO}; out = {}; break;
    case MX_KIND_STEREO_PANNER: need(0, "()"); in = {MX_MONO, MX_MONO}; out = {MX_STEREO}; break;
    case MX_KIND_STEREO_SPLITTER: need(0, "()"); in = {MX_STEREO}; out = {MX_MONO, MX_MONO}; break;
    case MX_KIND_TRIGGER: need(sizeof(mx_trigger_params), "mx_trigger_params"); in = {}; out = {MX_MONO}; break;
    case MX_KIND_SOURCE_MONO: in = {}; out = {MX_MONO}; break;
    case MX_KIND_SOURCE_STEREO: in = {}; out = {MX_STEREO}; break;
    case MX_KIND_VIDEO_MIXER: need(sizeof(mx_video_mixer_params), "mx_video_mixer_params"); in.assign(4, MX_VIDEO); out.assign(3, MX_VIDEO); break;  // video_mixer.rs:42-49
    case MX_KIND_SOURCE_VIDEO: in = {}; out = {MX_VIDEO}; break;
    case MX_KIND_FIR: in = {MX_STEREO}; out = {MX_STEREO}; break;        // blob checked in the constructor
    case MX_KIND_RESAMPLE: in = {MX_STEREO}; out = {MX_STEREO}; break;
    case MX_KIND_VIDEO_TO_RGBA: need(sizeof(mx_video_to_rgba_params), "mx_video_to_rgba_params"); in = {MX_VIDEO}; out = {}; break;
    case MX_KIND_MONITOR:   // monitor.rs:99-102
        if (params_len != sizeof(mx_monitor_params_ex)) need(sizeof(mx_monitor_params), "mx_monitor_params (or mx_monitor_params_ex)");
        in = {MX_VIDEO, MX_STEREO}; out = {}; break;
    default: throw Error(MX_ERR_INVALID, "unknown module kind");
    }
}

// first column of A^n for the 4-pole cascade's one-sample matrix A (lower-triangular Toeplitz,
// first column f^k (1-f)): binary exponentiation on polynomials mod x^4, in long double
static void toeplitz_pow_ld(long double f, uint64_t n, long double res[4]) {
    long double base[4] = {1.0L - f, f * (1.0L - f), f * f * (1.0L - f), f * f * f * (1.0L - f)};
    res[0] = 1.0L; res[1] = res[2] = res[3] = 0.0L;
    auto mul = [](const long double a[4], const long double b[4], long double c[4]) {
        long double t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) for (int j = 0; i + j < 4; ++j) t[i + j] += a[i] * b[j];
        for (int i = 0; i < 4; ++i) c[i] = t[i];
    };
    while (n) {
        if (n & 1) mul(res, base, res);
        mul(base, base, base);
        n >>= 1;
    }
}
static void toeplitz_pow(long double f, uint64_t n, double out[4]) {
    long double r[4];
    toeplitz_pow_ld(f, n, r);
    for (int i = 0; i < 4; ++i) out[i] = (double)r[i];
}
// y = T(a) v for the lower-triangular Toeplitz matrix with first column a
static void toeplitz_apply_ld(const long double a[4], const long double v[4], long double y[4]) {
    for (int i = 0; i < 4; ++i) { y[i] = 0.0L; for (int k = 0; k <= i; ++k) y[i] += a[k] * v[i - k]; }
}

static inline size_t floats_per_frame(uint8_t lt) { return lt == MX_MONO ? 1 : (lt == MX_STEREO ? 2 : 0); }

Graph::Graph(const mx_node* nodes, size_t n_nodes, const mx_edge* edges, size_t n_edges, const mx_graph_opts& o,
             size_t cap_frames_override) {
    if (n_nodes && !nodes) throw Error(MX_ERR_INVALID, "nodes is NULL");
    if (n_edges && !edges) throw Error(MX_ERR_INVALID, "edges is NULL");
    const uint32_t sr = o.sample_rate ? o.sample_rate : 44100u;          // src/engine.rs:53
    const uint32_t tps = o.ticks_per_second ? o.ticks_per_second : 60u;  // src/engine.rs:54
    if (sr % tps) throw Error(MX_ERR_INVALID, "sample_rate must be a multiple of ticks_per_second");
    sample_rate_ = (double)sr;
    spt_ = sr / tps;                                                     // src/engine.rs:55
    tps_ = tps;
    const uint32_t max_ticks = o.max_ticks_per_run ? o.max_ticks_per_run : 1u;
    cap_frames_ = cap_frames_override ? cap_frames_override : spt_ * (size_t)max_ticks;
    flags_ = o.flags;
    if ((flags_ & MX_FLAG_FP_CONTRACT) && (flags_ & MX_FLAG_EQ_FAST) && !(flags_ & MX_FLAG_EQ_EXACT))
        throw Error(MX_ERR_INVALID, "MX_FLAG_FP_CONTRACT is a mode of the exact-order kernels: not with MX_FLAG_EQ_FAST");

    if (o.device >= 0) { hip_check(hipSetDevice(o.device), "hipSetDevice"); device_ = o.device; }
    else hip_check(hipGetDevice(&device_), "hipGetDevice");
    if (o.stream) { stream_ = (hipStream_t)o.stream; own_stream_ = false; }
    else { hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate"); own_stream_ = true; }

    // host-side coefficients: same libm as the reference on this host (eq_three.rs:113-115)
    lo_f_ = 2.0 * std::sin(3.14159265358979323846264338327950288 * 420.0 / sample_rate_);
    hi_f_ = 2.0 * std::sin(3.14159265358979323846264338327950288 * 2700.0 / sample_rate_);

    nodes_.resize(n_nodes);
    for (size_t i = 0; i < n_nodes; ++i) {
        Node& n = nodes_[i];
        n.kind = nodes[i].kind;
        if (nodes[i].params_len && !nodes[i].params) throw Error(MX_ERR_INVALID, "node params is NULL");
        n.params.assign((const uint8_t*)nodes[i].params, (const uint8_t*)nodes[i].params + nodes[i].params_len);
        kind_ports(n.kind, n.params.size(), n.in_type, n.out_type);
        n.in_src.assign(n.in_type.size(), PortRef{});
        n.out_off.assign(n.out_type.size(), 0);
        n.out_off2.assign(n.out_type.size(), SIZE_MAX);
    }
    for (size_t e = 0; e < n_edges; ++e) {
        const mx_edge& ed = edges[e];
        if (ed.src_node >= n_nodes || ed.dst_node >= n_nodes) throw Error(MX_ERR_INVALID, "edge references a node out of range");
        Node& s = nodes_[ed.src_node];
        Node& d = nodes_[ed.dst_node];
        if (ed.src_port >= s.out_type.size() || ed.dst_port >= d.in_type.size()) throw Error(MX_ERR_INVALID, "edge references a terminal out of range");
        if (s.out_type[ed.src_port] != d.in_type[ed.dst_port])   // Workspace::connect, workspace.rs:97-114
            throw Error(MX_ERR_TYPE, "line type mismatch on connection");
        d.in_src[ed.dst_port] = PortRef{(int32_t)ed.src_node, ed.src_port};
    }

    // Run order: DFS through inputs from terminal modules (src/engine.rs:408-457).  The reference
    // iterates a HashSet (unspecified order); ascending id is one of its valid orders.
    std::vector<uint8_t> feeds(n_nodes, 0), seen(n_nodes, 0);
    for (size_t e = 0; e < n_edges; ++e) feeds[edges[e].src_node] = 1;
    // iterative DFS (graphs with 1e5 nodes must not blow the stack)
    struct Frame { uint32_t id; uint32_t next; };
    std::vector<Frame> stack;
    for (uint32_t root = 0; root < n_nodes; ++root) {
        if (feeds[root] || seen[root]) continue;
        seen[root] = 1; stack.push_back({root, 0});
        while (!stack.empty()) {
            Frame& f = stack.back();
            Node& n = nodes_[f.id];
            if (f.next < n.in_src.size()) {
                const PortRef pr = n.in_src[f.next++];
                if (pr.node >= 0 && !seen[pr.node]) { seen[pr.node] = 1; stack.push_back({(uint32_t)pr.node, 0}); }
            } else {
                order_.push_back(f.id);
                stack.pop_back();
            }
        }
    }
    // A cycle with no terminal module is never reached from a terminal, so the reference never runs
    // it (engine.rs:428-430); we mirror that: such nodes stay out of order_.

    // dependency levels; an input whose producer runs later this tick is a back-edge and reads
    // Disconnected (engine.rs:479-482: buffers.get(output_id) is None)
    std::vector<int64_t> pos(n_nodes, -1);
    for (size_t i = 0; i < order_.size(); ++i) pos[order_[i]] = (int64_t)i;
    for (uint32_t id : order_) {
        Node& n = nodes_[id];
        int lvl = 0;
        for (PortRef& pr : n.in_src) {
            if (pr.node < 0) continue;
            if (pos[pr.node] < 0 || pos[pr.node] >= pos[id]) { pr.node = -1; continue; }  // back-edge => Disconnected
            lvl = std::max(lvl, nodes_[pr.node].level + 1);
        }
        n.level = lvl;
        n.in_src_orig = n.in_src;
    }
    // variable-length blobs of the build-specified modules
    for (const Node& n : nodes_) {
        if (n.kind == MX_KIND_FIR) {
            mx_fir_params h{}; if (n.params.size() >= sizeof h) std::memcpy(&h, n.params.data(), sizeof h);
            if (n.params.size() < sizeof h || h.n_taps == 0 || h.n_taps > 16384 || n.params.size() != sizeof h + (size_t)h.n_taps * sizeof(double))
                throw Error(MX_ERR_INVALID, "mx_fir_params: params_len must be 8 + 8 * n_taps (1 <= n_taps <= 16384)");
        } else if (n.kind == MX_KIND_RESAMPLE) {
            mx_resample_params h{}; if (n.params.size() >= sizeof h) std::memcpy(&h, n.params.data(), sizeof h);
            if (n.params.size() < sizeof h || !h.up || !h.down || !h.taps_per_phase || h.taps_per_phase > 4096 ||
                n.params.size() != sizeof h + (size_t)h.up * h.taps_per_phase * sizeof(double))
                throw Error(MX_ERR_INVALID, "mx_resample_params: params_len must be 16 + 8 * up * taps_per_phase");
        }
    }
    // sample-rate domains: every input of a node must live in one domain; Resample multiplies it by up / down
    auto gcd_u = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a ? a : 1; };
    for (uint32_t id : order_) {
        Node& n = nodes_[id];
        bool have = false; uint32_t dn = 1, dd = 1;
        for (const PortRef& pr : n.in_src) {
            if (pr.node < 0 || n.in_type[&pr - n.in_src.data()] == MX_VIDEO) continue;
            const Node& sn = nodes_[pr.node];
            if (have && (sn.dom_num != dn || sn.dom_den != dd)) throw Error(MX_ERR_TYPE, "inputs of a module live in different sample-rate domains");
            dn = sn.dom_num; dd = sn.dom_den; have = true;
        }
        n.in_dom_num = dn; n.in_dom_den = dd; n.dom_num = dn; n.dom_den = dd;
        if (n.kind == MX_KIND_RESAMPLE) {
            mx_resample_params h; std::memcpy(&h, n.params.data(), sizeof h);
            const uint64_t a = (uint64_t)dn * h.up, b = (uint64_t)dd * h.down, g = gcd_u(a, b);
            n.dom_num = (uint32_t)(a / g); n.dom_den = (uint32_t)(b / g);
        }
        if ((spt_ * n.dom_num) % n.dom_den) throw Error(MX_ERR_INVALID, "resampling ratio does not give a whole number of samples per tick");
        const bool rate_dependent = n.kind == MX_KIND_EQ_THREE || n.kind == MX_KIND_ENVELOPE || n.kind == MX_KIND_OSCILLATOR || n.kind == MX_KIND_FM_SINE;
        if (n.kind == MX_KIND_PLOTTER && (n.in_dom_num != 1 || n.in_dom_den != 1))
            throw Error(MX_ERR_INVALID, "Plotter is only accepted in the base sample-rate domain");
        if (rate_dependent && (n.dom_num != 1 || n.dom_den != 1))
            throw Error(MX_ERR_INVALID, "EqThree / Envelope / Oscillator / FmSine depend on the sample rate and are only accepted in the base domain");
    }
    for (Node& n : nodes_) { n.out_elided.assign(n.out_type.size(), 0); n.out_dup.assign(n.out_type.size(), 0); }
    if (!(flags_ & MX_FLAG_NO_FUSE)) plan_fusion();
    // groups: (level, kind); nodes folded into another node's kernel are never launched.  EqThree nodes also by the epilogue the compiler gave them (plain, -> Panner,
    // -> Amplifier with a constant / a buffer / an inline Envelope as control; stereo or one float per frame): a launch group of ONE mode takes the kernel
    // specialised for it -- sixteen strips of another mode among a thousand put the whole group on the general direct-load form (29 ms against 5, tools/ctl_probe.py)
    auto eq_key = [&](const Node& nd) -> int {
        if (nd.kind != MX_KIND_EQ_THREE) return 0;
        const uint32_t epi = nd.fuse_amp >= 0 ? 2u : (nd.fuse_pan >= 0 ? 1u : 0u);
        uint32_t fl = 0;
        if (nd.fuse_amp >= 0 ? nodes_[nd.fuse_amp].out_dup[0] : (nd.fuse_pan >= 0 && nodes_[nd.fuse_pan].out_dup[0])) fl |= MX_EQF_MONO_DUP;
        if (nd.fuse_env >= 0) fl |= MX_EQF_ENV;
        const bool has_ctl = nd.fuse_amp >= 0 && nd.fuse_env < 0 && nodes_[nd.fuse_amp].in_src[1].node >= 0;
        return 1 + eq_epilogue_mode(epi, fl, has_ctl);
    };
    for (Node& n : nodes_) n.sub_key = eq_key(n);
    std::vector<uint32_t> sorted;
    for (uint32_t id : order_) if (!nodes_[id].elided) sorted.push_back(id);
    std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) {
        if (nodes_[a].level != nodes_[b].level) return nodes_[a].level < nodes_[b].level;
        if (nodes_[a].kind != nodes_[b].kind) return nodes_[a].kind < nodes_[b].kind;
        if (nodes_[a].sub_key != nodes_[b].sub_key) return nodes_[a].sub_key < nodes_[b].sub_key;
        const uint64_t da = ((uint64_t)nodes_[a].dom_num << 32) | nodes_[a].dom_den, db = ((uint64_t)nodes_[b].dom_num << 32) | nodes_[b].dom_den;
        return da < db;
    });
    for (uint32_t id : sorted) {
        Node& n = nodes_[id];
        if (groups_.empty() || groups_.back().level != n.level || groups_.back().kind != n.kind || groups_.back().sub_key != n.sub_key ||
            groups_.back().dom_num != n.dom_num || groups_.back().dom_den != n.dom_den ||
            groups_.back().in_dom_num != n.in_dom_num || groups_.back().in_dom_den != n.in_dom_den) {
            Group g; g.level = n.level; g.kind = n.kind; g.sub_key = n.sub_key;
            g.dom_num = n.dom_num; g.dom_den = n.dom_den; g.in_dom_num = n.in_dom_num; g.in_dom_den = n.in_dom_den;
            groups_.push_back(std::move(g));
        }
        n.group = (int)groups_.size() - 1;
        n.slot = (uint32_t)groups_.back().nodes.size();
        groups_.back().nodes.push_back(id);
    }
    for (uint32_t id = 0; id < nodes_.size(); ++id) if (nodes_[id].kind == MX_KIND_PLOTTER && nodes_[id].group >= 0) plotter_nodes_.push_back(id);
    for (uint32_t id : order_) if (nodes_[id].kind == MX_KIND_VIDEO_MIXER || nodes_[id].kind == MX_KIND_SOURCE_VIDEO || nodes_[id].kind == MX_KIND_VIDEO_TO_RGBA || nodes_[id].kind == MX_KIND_MONITOR) video_order_.push_back(id);
    // video nodes: per-node state lives on the host, pixels on the graph's stream
    for (Node& n : nodes_) {
        if (n.kind == MX_KIND_VIDEO_MIXER) {
            mx_video_mixer_params p; std::memcpy(&p, n.params.data(), sizeof p);
            n.vmixer.reset(new VideoMixer(p, sr, stream_));
        }
        if (n.kind == MX_KIND_MONITOR) {
            mx_monitor_params p; std::memcpy(&p, n.params.data(), sizeof p);
            if (p.width == 0 || p.height == 0 || (p.width & 1) || (p.height & 1) || p.width > 16384 || p.height > 16384)
                throw Error(MX_ERR_INVALID, "monitor picture must be non-zero, even and at most 16384 a side");
            n.mon_scaler = std::make_shared<Scaler>(p.width, p.height, stream_);
            if (n.params.size() == sizeof(mx_monitor_params_ex)) { mx_monitor_params_ex px; std::memcpy(&px, n.params.data(), sizeof px); n.mon_depth = px.queue_depth; }
            // without a queue depth every tick of a submission keeps its scaled picture on the device until the next run
            const uint64_t kept = (uint64_t)(n.mon_depth ? n.mon_depth : std::max<uint32_t>(1u, o.max_ticks_per_run)) * (((uint64_t)p.width + 63) & ~63ull) * p.height * 3 / 2;
            if (kept > (64ull << 30)) throw Error(MX_ERR_NOMEM, "a Monitor that keeps every tick of a submission would hold more than 64 GiB of pictures: give it a queue depth (mx_monitor_params_ex) or fewer ticks per run");
        }
        if (n.kind == MX_KIND_VIDEO_MIXER || n.kind == MX_KIND_SOURCE_VIDEO || n.kind == MX_KIND_VIDEO_TO_RGBA || n.kind == MX_KIND_MONITOR) has_video_ = true;
        n.vout.resize(n.out_type.size());
    }
    // a VideoMixer whose program output feeds exactly one video node of this graph hands it over as an
    // unevaluated cross-fade chain: a cascade of mixers becomes one fused pass (mx_k_video.hip k_fade_chain*)
    if (!(flags_ & MX_FLAG_NO_FUSE)) {
        std::vector<uint32_t> n_cons(nodes_.size(), 0); std::vector<uint8_t> ok(nodes_.size(), 1);
        for (const Node& n : nodes_)
            for (const PortRef& pr : n.in_src)
                if (pr.node >= 0 && nodes_[pr.node].kind == MX_KIND_VIDEO_MIXER && pr.port == 0) {
                    n_cons[pr.node]++;
                    if (n.kind != MX_KIND_VIDEO_MIXER && n.kind != MX_KIND_VIDEO_TO_RGBA) ok[pr.node] = 0;
                }
        for (size_t i = 0; i < nodes_.size(); ++i)
            if (nodes_[i].vmixer) { nodes_[i].vlazy = n_cons[i] == 1 && ok[i]; nodes_[i].vmixer->set_lazy_program(nodes_[i].vlazy, tps); }
    }
    // time-parallel EqThree tables (unused in MX_FLAG_EQ_EXACT mode)
    {
        std::vector<EqScanTab> tabs(4);
        const long double fs[2] = {(long double)lo_f_, (long double)hi_f_};
        for (int v = 0; v < 4; ++v) {
            const uint64_t L = 4ull << v;
            for (int f = 0; f < 2; ++f) {
                for (int j = 0; j <= 64; ++j) toeplitz_pow(fs[f], L * (uint64_t)j, tabs[v].pw[f][j]);
                for (int k = 0; k < 6; ++k) toeplitz_pow(fs[f], L << k, tabs[v].p2[f][k]);
                // phase-A tables: s' = A s + b x + c with b = (f, f^2, f^3, f^4), c = VSA (1, f, f^2, f^3) (eq_three.rs:117-124)
                const long double ff = fs[f], vsa = 1.0L / 4294967295.0L;
                const long double b[4] = {ff, ff * ff, ff * ff * ff, ff * ff * ff * ff};
                const long double c[4] = {vsa, vsa * ff, vsa * ff * ff, vsa * ff * ff * ff};
                long double sum[4] = {0, 0, 0, 0};
                for (uint64_t m = 0; m < 32; ++m) {
                    long double am[4], hb[4];
                    toeplitz_pow_ld(ff, m, am);
                    toeplitz_apply_ld(am, b, hb);
                    for (int q = 0; q < 4; ++q) tabs[v].h[f][m][q] = (double)hb[q];
                    if (m < L) for (int q = 0; q < 4; ++q) sum[q] += am[q];
                }
                long double cz[4];
                toeplitz_apply_ld(sum, c, cz);
                for (int q = 0; q < 4; ++q) tabs[v].cz[f][q] = (double)cz[q];
            }
        }
        eq_tabs_.alloc(tabs.size() * sizeof(EqScanTab));
        hip_check(hipMemcpy(eq_tabs_.p, tabs.data(), tabs.size() * sizeof(EqScanTab), hipMemcpyHostToDevice), "hipMemcpy(eq tabs)");
    }
    // MX_FLAG_OVERLAP_TAIL: the last launch group, if it is a Mixer bank alone on the highest level of an audio-only graph, may run beside
    // the next run's earlier groups; every port it reads gets a second buffer (the next run must not overwrite what it is still reading)
    // AUTOMATIC (round 5) for graphs with at least 64 EqThree instances and submissions of at least 16 ticks, while the second buffers stay below MX_OVERLAP_AUTO_MAX_GB
    // (default 32) and a quarter of the free device memory: the bank's launch is held back until the next run's EqThree launch has been placed (flush_deferred_tail,
    // k_tail_gate) and then shares the SIMDs with it -- 1024 strips x 2048 ticks 5.42 -> 4.84 ms, x 256 ticks 0.915 -> 0.860, 128 strips x 2048 ticks 0.934 -> 0.875
    // (tools/q_gate.sh).  Launched at once instead (round 4, MX_TAIL_GATE=0) the same mode LOST from 256 ticks up: the next run's k_env_ticks ran beside the bank (140 us
    // instead of 9) with the EqThree launch waiting behind it.  MX_OVERLAP_AUTO=0 turns the automatism off; results are bit-identical either way.
    { const char* const sm = getenv("MX_SIN_MODE"); sin_mode_ = sm ? atoi(sm) : 0; }   // (A/B and tests: mx_k_stream.hip SIN_MODE)
    bool overlap_auto = false;
    {
        const char* const ae = getenv("MX_OVERLAP_AUTO");   // read per graph: tests build both kinds in one process
        const bool auto_on = !(ae && atoi(ae) == 0);
        const size_t max_ticks = spt_ ? cap_frames_ / spt_ : 0;
        size_t n_eq = 0;
        for (const Group& g : groups_) if (g.kind == MX_KIND_EQ_THREE) n_eq = std::max(n_eq, g.nodes.size());
        overlap_auto = auto_on && eq_exact() && n_eq >= 64 && max_ticks >= 16;
    }
    // The tail is every Mixer group at the END of the launch order: a bank, or a bank and the buses above it (group buses -> master).  What its groups read from the
    // groups before them is double-buffered; what they read from each other is not (they run in order on the one tail stream).
    size_t t_first = groups_.size();
    while (t_first > 0 && groups_[t_first - 1].kind == MX_KIND_MIXER) --t_first;
    if (((flags_ & MX_FLAG_OVERLAP_TAIL) || overlap_auto) && !has_video_ && groups_.size() >= 2 && t_first >= 1 && t_first < groups_.size() &&
        groups_[t_first - 1].level < groups_[t_first].level && plotter_nodes_.empty()) {
        bool ok = true;
        std::vector<std::pair<uint32_t, uint32_t>> ports;
        for (size_t tg = t_first; tg < groups_.size() && ok; ++tg)
            for (uint32_t id : groups_[tg].nodes) {
                for (const PortRef& pr : nodes_[id].in_src) {
                    if (pr.node < 0) continue;
                    const Node& sn = nodes_[pr.node];
                    if (sn.group >= (int)t_first && sn.group < (int)tg) continue;   // another tail group's output: same stream, in order
                    if (sn.kind == MX_KIND_SOURCE_MONO || sn.kind == MX_KIND_SOURCE_STEREO || sn.group >= (int)tg) { ok = false; break; }   // a source is rewritten by the caller while the tail may still read it
                    if (std::find(ports.begin(), ports.end(), std::make_pair((uint32_t)pr.node, (uint32_t)pr.port)) == ports.end()) ports.push_back({(uint32_t)pr.node, pr.port});
                }
                if (!ok) break;
            }
        if (ok && overlap_auto && !(flags_ & MX_FLAG_OVERLAP_TAIL)) {   // the second buffers must be affordable (an upper bound: every port as interleaved stereo)
            const char* const ge = getenv("MX_OVERLAP_AUTO_MAX_GB");
            const double max_gb = ge ? atof(ge) : 32.0;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
            const double extra = (double)ports.size() * (double)cap_frames_ * 8.0;
            if (extra > max_gb * 1073741824.0 || extra > (double)free_b / 4.0) ok = false;
        }
        if (ok) {
            tail_gi_ = (int)t_first;
            tail_auto_ = !(flags_ & MX_FLAG_OVERLAP_TAIL);
            for (auto& pp : ports) nodes_[pp.first].out_off2[pp.second] = 0;   // marked; layout_slab gives it its offset
            hip_check(hipStreamCreateWithFlags(&tail_stream_, hipStreamNonBlocking), "hipStreamCreate(tail)");
            hip_check(hipEventCreateWithFlags(&ev_head_done_, hipEventDisableTiming), "hipEventCreate");
            for (auto& e : ev_tail_done_) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        }
    }
    layout_slab();
    build_descriptors();
}

void Graph::flush_deferred_tail(bool gated) {
    if (!deferred_.pending) return;
    deferred_.pending = false;
    hip_check(hipStreamWaitEvent(tail_stream_, ev_head_done_, 0), "hipStreamWaitEvent");
    if (gated && gate_armed_) { launch_tail_gate((const uint32_t*)gate_flag_.p, gate_seq_, 300u, tail_stream_); ++n_gated_; } else ++n_at_once_;
    if (deferred_.prof_begin) hip_check(hipEventRecord(deferred_.prof_begin, tail_stream_), "hipEventRecord");
    for (const TailLaunch& t : deferred_.items) {
        launch_mixer((const MixDesc*)t.desc, t.n, t.max_ch, t.frames, t.dup_mode, tail_stream_);
        if (t.prof_ev) hip_check(hipEventRecord(t.prof_ev, tail_stream_), "hipEventRecord");
    }
    deferred_.items.clear();
    if (tail_hook_) { auto hook = std::move(tail_hook_); tail_hook_ = nullptr; hook(tail_stream_); }   // (mx_exchange: pack + exchange of that run's buses, behind the bank)
    // recorded AFTER the hook: whoever waits for this tail (wait_tail) is then also ordered behind the hook's reads of the buses on the tail stream -- a later run's Mixer on
    // stream_ must not overwrite them under a pack that is still copying
    hip_check(hipEventRecord(ev_tail_done_[deferred_.parity], tail_stream_), "hipEventRecord");
    tail_pending_[deferred_.parity] = true;
}

void Graph::wait_tail(int parity_or_all) {
    if (deferred_.pending && (parity_or_all < 0 || parity_or_all == (int)deferred_.parity)) flush_deferred_tail(false);
    for (int p = 0; p < 2; ++p)
        if ((parity_or_all < 0 || parity_or_all == p) && tail_pending_[p]) {
            hip_check(hipStreamWaitEvent(stream_, ev_tail_done_[p], 0), "hipStreamWaitEvent");
            tail_pending_[p] = false;
        }
}

// The automatic second-stream mode ends, for good.  Everything outstanding completes; the double-buffered ports go back to their FIRST buffer (the last run's data moves
// there when it sits in the second), the second set of descriptors is dropped and the first rebuilt -- from here on the graph is a one-stream graph in every respect
// (update_params, cut runs and ensure_capacity touch the only descriptors there are).
void Graph::end_auto_tail() {
    if (tail_gi_ < 0) return;
    sync();
    for (Node& n : nodes_)
        for (size_t k = 0; k < n.out_type.size(); ++k) {
            if (n.out_off2[k] == SIZE_MAX) continue;
            if (parity_ && n.out_off[k] != SIZE_MAX) {
                const size_t fl = (n.out_dup[k] ? 1 : floats_per_frame(n.out_type[k])) * (cap_frames_ * n.dom_num / n.dom_den + 1);
                hip_check(hipMemcpyAsync((float*)slab_.p + n.out_off[k], (const float*)slab_.p + n.out_off2[k], fl * sizeof(float), hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync(D2D)");
            }
            n.out_off2[k] = SIZE_MAX;
        }
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    tail_gi_ = -1;
    parity_ = 0;
    overlap_this_run_ = false;
    for (Group& g : groups_) { g.desc_alt.free_(); g.extra_alt.free_(); }
    build_descriptors();
}

Graph::~Graph() {
    flush_scales(stream_);
    try { flush_deferred_tail(false); } catch (...) {}
    if (tail_stream_) (void)hipStreamSynchronize(tail_stream_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (tail_stream_) { (void)hipStreamDestroy(tail_stream_); (void)hipEventDestroy(ev_head_done_); for (auto& e : ev_tail_done_) (void)hipEventDestroy(e); }
    for (Node& n : nodes_) { n.vmixer.reset(); n.vout.clear(); n.vsrc = FrameRef(); n.vsrc_ring.clear(); n.vsrc_sched.clear(); }
    for (auto& v : prof_runs_) for (auto& e : v) (void)hipEventDestroy(e);
    for (auto& v : prof_pool_) for (auto& e : v) (void)hipEventDestroy(e);
    for (Stage& st : stage_) { if (st.done) (void)hipEventDestroy(st.done); if (st.host) (void)hipHostFree(st.host); }
    // the descriptor ring launch_video_batch keeps per stream goes with a stream this graph OWNS; a caller's stream may be shared with other
    // graphs / scalers that are launching on it right now (their ring must not be freed under them): its ring lives as long as the process
    if (own_stream_ && stream_) { video_stream_retired(stream_); (void)hipStreamDestroy(stream_); }
}

// Graph-compiler fusion.  A port buffer is a per-tick temporary of Engine::run_tick
// (src/engine.rs:461,504-506), observable only through connections, so a port whose only consumers
// are folded into its producer's kernel need not exist.  Two patterns, both bit-identical to the
// separate modules because the folded arithmetic is applied to the very f32 the producer stores:
//   F1  EqThree -> StereoPanner(L = R = that EQ) [-> Amplifier input]   => epilogue of the EQ kernel
//   F2  Trigger (single consumer) -> Envelope gate                       => constant gate, no buffer
void Graph::plan_fusion() {
    const size_t N = nodes_.size();
    std::vector<std::vector<std::vector<std::pair<uint32_t, uint32_t>>>> cons(N);
    for (size_t i = 0; i < N; ++i) cons[i].resize(nodes_[i].out_type.size());
    for (uint32_t id : order_)
        for (uint32_t k = 0; k < nodes_[id].in_src.size(); ++k) {
            const PortRef pr = nodes_[id].in_src[k];
            if (pr.node >= 0) cons[pr.node][pr.port].push_back({id, k});
        }
    // does `from` (transitively, through its inputs) depend on `target`?  bounded search, conservative
    auto depends_on = [&](int32_t from, uint32_t target) {
        std::vector<int32_t> st{from};
        size_t visited = 0;
        while (!st.empty()) {
            const int32_t x = st.back(); st.pop_back();
            if ((uint32_t)x == target) return true;
            if (++visited > 256) return true;
            for (const PortRef& pr : nodes_[x].in_src) if (pr.node >= 0) st.push_back(pr.node);
        }
        return false;
    };
    for (uint32_t e : order_) {
        Node& E = nodes_[e];
        if (E.kind != MX_KIND_EQ_THREE) continue;
        const auto& c = cons[e][0];
        if (c.size() != 2 || c[0].first != c[1].first || c[0].second == c[1].second) continue;
        const uint32_t p = c[0].first;
        Node& P = nodes_[p];
        if (P.kind != MX_KIND_STEREO_PANNER || P.elided) continue;
        E.fuse_pan = (int32_t)p; E.out_elided[0] = 1;
        P.elided = true; P.owner = (int32_t)e;
        const auto& pc = cons[p][0];
        if (pc.size() == 1 && nodes_[pc[0].first].kind == MX_KIND_AMPLIFIER && pc[0].second == 0) {
            const uint32_t a = pc[0].first;
            Node& A = nodes_[a];
            const PortRef ctl = A.in_src[1];
            if (ctl.node < 0 || !depends_on(ctl.node, e)) {
                E.fuse_amp = (int32_t)a; P.out_elided[0] = 1;
                A.elided = true; A.owner = (int32_t)e;
                if (ctl.node >= 0) E.level = std::max(E.level, nodes_[ctl.node].level + 1);   // the fused kernel reads the control port
            }
        }
    }
    for (uint32_t v : order_) {
        Node& V = nodes_[v];
        if (V.kind != MX_KIND_ENVELOPE) continue;
        const PortRef g = V.in_src[0];
        if (g.node < 0 || nodes_[g.node].kind != MX_KIND_TRIGGER || cons[g.node][0].size() != 1) continue;
        Node& G = nodes_[g.node];
        V.fuse_trigger = g.node;
        G.elided = true; G.owner = (int32_t)v; G.out_elided[0] = 1;
    }
    for (uint32_t e : order_) {
        Node& E = nodes_[e];
        if (E.kind != MX_KIND_EQ_THREE || E.fuse_pan < 0) continue;
        // F3: the fused Amplifier's control is a constant-gate Envelope that feeds nothing else: its closed form
        // (envelope.rs:34-58) is evaluated in the epilogue; for a constant gate only the run's first sample can
        // change the Envelope's state, so no per-sample state machine is needed
        if (E.fuse_amp >= 0) {
            const PortRef ctl = nodes_[E.fuse_amp].in_src[1];
            if (ctl.node >= 0 && nodes_[ctl.node].kind == MX_KIND_ENVELOPE && nodes_[ctl.node].fuse_trigger >= 0 &&
                cons[ctl.node][0].size() == 1) {
                Node& V = nodes_[ctl.node];
                E.fuse_env = ctl.node;
                V.elided = true; V.owner = (int32_t)e; V.out_elided[0] = 1;
            }
        }
        // F4: the fused result is stereo with L == R by construction; if only Mixers read it, store one float per frame
        const uint32_t x = (uint32_t)(E.fuse_amp >= 0 ? E.fuse_amp : E.fuse_pan);
        const auto& xc = cons[x][0];
        bool only_mixers = !xc.empty();
        for (const auto& c : xc) only_mixers = only_mixers && nodes_[c.first].kind == MX_KIND_MIXER;
        if (only_mixers) nodes_[x].out_dup[0] = 1;
    }
}

void Graph::layout_slab() {
    size_t off = 0;
    auto bump = [&](size_t floats) { size_t o = off; off += (floats + 63) & ~(size_t)63; return o; };  // 256-byte aligned
    // Disconnected inputs read this region (io.rs:8-9); a node behind a Resample runs cap_frames * up / down frames per
    // launch, so the region is sized for the largest sample-rate domain of the graph
    size_t zero_frames = cap_frames_;
    for (const Node& n : nodes_) {
        zero_frames = std::max(zero_frames, cap_frames_ * n.dom_num / n.dom_den + 1);
        zero_frames = std::max(zero_frames, cap_frames_ * n.in_dom_num / n.in_dom_den + 1);
    }
    zero_off_ = bump(2 * zero_frames);
    for (Node& n : nodes_)
        for (size_t k = 0; k < n.out_type.size(); ++k)
        {
            const size_t fl = (n.out_dup[k] ? 1 : floats_per_frame(n.out_type[k])) * (cap_frames_ * n.dom_num / n.dom_den + 1);
            n.out_off[k] = n.out_elided[k] ? SIZE_MAX : bump(fl);
            if (n.out_off2[k] != SIZE_MAX) n.out_off2[k] = n.out_elided[k] ? SIZE_MAX : bump(fl);
        }
    slab_floats_ = off;
    slab_.alloc(off * sizeof(float));
    hip_check(hipMemsetAsync(slab_.p, 0, off * sizeof(float), stream_), "hipMemsetAsync(slab)");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

float* Graph::out_ptr(const Node& n, uint32_t port) const {
    // run_off_frames_: a run that is cut at scheduled parameter updates launches span by span; a span's buffers start that many
    // (base-rate) frames into the port
    const size_t fpf = n.out_dup[port] ? 1 : floats_per_frame(n.out_type[port]);
    const size_t off = fpf * (run_off_frames_ * n.dom_num / n.dom_den);
    if (n.bound && port == 0) return const_cast<float*>(n.bound) + off;
    const bool alt = (building_alt_ || (parity_ && !building_main_)) && n.out_off2[port] != SIZE_MAX;
    return (float*)slab_.p + (alt ? n.out_off2[port] : n.out_off[port]) + off;
}
const float* Graph::in_ptr(const Node& n, uint32_t port, bool null_if_disconnected) const {
    const PortRef pr = n.in_src[port];
    if (pr.node < 0) return null_if_disconnected ? nullptr : (const float*)slab_.p + zero_off_;
    return out_ptr(nodes_[pr.node], pr.port);
}

static double db_to_linear(double db) { return std::pow(10.0, db / 20.0); }   // protocol/src/lib.rs:469-471

void Graph::upload_group_one(Group& g) {
    const size_t n = g.nodes.size();
    auto up = [&](DevBuf& b, const void* src, size_t bytes) {
        if (b.bytes < bytes || !b.p) b.alloc(bytes);
        if (bytes) hip_check(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(desc)");
    };
    switch (g.kind) {
    case MX_KIND_AMPLIFIER: {
        std::vector<AmpDesc> d(n);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_amplifier_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            d[i] = AmpDesc{in_ptr(nd, 0, false), in_ptr(nd, 1, true), out_ptr(nd, 0), p.amplitude, p.mod_depth};
        }
        up(desc_buf(g), d.data(), n * sizeof(AmpDesc));
        break;
    }
    case MX_KIND_ENVELOPE: {
        std::vector<EnvDesc> d(n);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_envelope_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            float gate_const = 0.f; uint32_t use_const = 0;
            if (nd.fuse_trigger >= 0) {   // Trigger folded in: its per-tick constant (trigger.rs:38-41) replaces the gate buffer (GateBits row i)
                mx_trigger_params tp; std::memcpy(&tp, nodes_[nd.fuse_trigger].params.data(), sizeof tp);
                gate_const = tp.gate_open ? 1.0f : 0.0f; use_const = 2; g.has_gates = true;
            }
            d[i] = EnvDesc{use_const ? nullptr : in_ptr(nd, 0, false), out_ptr(nd, 0), gate_const, use_const,
                           EnvParams{p.attack_ms, 1.0 / p.attack_ms, 1.0 / p.decay_ms,
                                     p.sustain_amplitude, 1.0 - p.sustain_amplitude, 1.0 / p.release_ms}};
        }
        up(desc_buf(g), d.data(), n * sizeof(EnvDesc));
        if (!g.state.p) { g.state.alloc(n * sizeof(EnvState)); hip_check(hipMemset(g.state.p, 0, n * sizeof(EnvState)), "hipMemset"); }
        break;
    }
    case MX_KIND_EQ_THREE: {
        std::vector<EqDesc> d(n);
        std::vector<EnvTickDesc> td(n);
        bool any_env = false;
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_eq_three_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            EqDesc e{};
            e.in = in_ptr(nd, 0, false);
            e.gain_lo = db_to_linear(p.gain_lo_db); e.gain_mid = db_to_linear(p.gain_mid_db); e.gain_hi = db_to_linear(p.gain_hi_db);
            if (nd.fuse_amp >= 0) {          // EqThree -> StereoPanner(L = R) -> Amplifier in one kernel
                const Node& amp = nodes_[nd.fuse_amp];
                if (amp.out_dup[0]) e.flags |= MX_EQF_MONO_DUP;
                mx_amplifier_params ap; std::memcpy(&ap, amp.params.data(), sizeof ap);
                e.out = out_ptr(amp, 0); e.ctl = in_ptr(amp, 1, true);
                e.amp_one_minus = 1.0 - ap.mod_depth; e.amp_mod_depth = ap.mod_depth; e.amp_amplitude = ap.amplitude; e.epi = 2;
            } else if (nd.fuse_pan >= 0) {   // EqThree -> StereoPanner(L = R)
                e.out = out_ptr(nodes_[nd.fuse_pan], 0); e.epi = 1;
                if (nodes_[nd.fuse_pan].out_dup[0]) e.flags |= MX_EQF_MONO_DUP;
            } else {
                e.out = out_ptr(nd, 0); e.epi = 0;
            }
            if (nd.fuse_env >= 0) {          // Envelope (constant gate) evaluated inline as the control
                const Node& env = nodes_[nd.fuse_env];
                mx_envelope_params ep; std::memcpy(&ep, env.params.data(), sizeof ep);
                e.flags |= MX_EQF_ENV; e.ctl = nullptr;
                e.env = EnvParams{ep.attack_ms, 1.0 / ep.attack_ms, 1.0 / ep.decay_ms, ep.sustain_amplitude, 1.0 - ep.sustain_amplitude, 1.0 / ep.release_ms};
                if (!g.state2.p) { g.state2.alloc(n * sizeof(EnvState)); hip_check(hipMemset(g.state2.p, 0, n * sizeof(EnvState)), "hipMemset"); }
                // its per-tick states are laid out by k_env_ticks before the EQ kernel runs (gate = the folded Trigger, GateBits row i)
                td[i] = EnvTickDesc{e.env, e.amp_one_minus, e.amp_mod_depth, (EnvState*)g.state2.p + i};
                any_env = true; g.has_gates = true;
            }
            d[i] = e;
            const int em = eq_epilogue_mode(e.epi, e.flags, e.ctl != nullptr);
            g.eq_mode = i == 0 ? em : (g.eq_mode == em ? em : -1);
            // the groups were cut by eq_key()'s idea of this mode (from the fusion plan); the descriptor's is the one the kernels see.  If they ever disagree a group
            // silently falls to the general direct-load kernel (29 ms against 5): refuse loudly instead
            if (nd.sub_key != 0 && nd.sub_key != 1 + em && !eq_mode_warned_) {   // (results are right either way: say it, once per graph, and go on)
                eq_mode_warned_ = true;
                fprintf(stderr, "mixlab_gpu: EqThree node %u was grouped for epilogue mode %d but its descriptor says %d: the group runs the general (slower) kernel\n",
                        (unsigned)g.nodes[i], nd.sub_key - 1, em);
            }
        }
        up(desc_buf(g), d.data(), n * sizeof(EqDesc));
        if (any_env) up(g.tick_desc, td.data(), n * sizeof(EnvTickDesc));
        if (!g.state.p) { g.state.alloc(n * sizeof(EqState)); hip_check(hipMemset(g.state.p, 0, n * sizeof(EqState)), "hipMemset"); }
        break;
    }
    case MX_KIND_FM_SINE: {
        std::vector<FmDesc> d(n);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_fm_sine_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            const double freq_amp = (p.freq_hi - p.freq_lo) / 2.0;    // fm_sine.rs:42
            const double freq_mid = p.freq_lo + freq_amp;            // fm_sine.rs:43
            d[i] = FmDesc{in_ptr(nd, 0, false), out_ptr(nd, 0), freq_mid, freq_amp};
        }
        up(desc_buf(g), d.data(), n * sizeof(FmDesc));
        break;
    }
    case MX_KIND_MIXER: {
        size_t total = 0;
        for (uint32_t id : g.nodes) total += nodes_[id].in_type.size();
        std::vector<MixChan> ch(total ? total : 1);
        DevBuf& xb = extra_buf(g);
        if (xb.bytes < ch.size() * sizeof(MixChan) || !xb.p) xb.alloc(ch.size() * sizeof(MixChan));
        std::vector<MixDesc> d(n);
        size_t o = 0, n_dup = 0;
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            const size_t nch = nd.in_type.size();
            const mx_mixer_channel_params* p = (const mx_mixer_channel_params*)nd.params.data();
            for (size_t c = 0; c < nch; ++c) {
                mx_mixer_channel_params cp; std::memcpy(&cp, p + c, sizeof cp);
                const PortRef src = nd.in_src[c];
                const uint32_t dup = (src.node >= 0 && nodes_[src.node].out_dup[src.port]) ? 1u : 0u;
                n_dup += dup;
                ch[o + c] = MixChan{in_ptr(nd, (uint32_t)c, false), cp.fader * db_to_linear(cp.gain_db), cp.cue ? 1u : 0u, dup};  // mixer.rs:59
            }
            d[i] = MixDesc{(const MixChan*)xb.p + o, (uint32_t)nch, 0u, out_ptr(nd, 0), out_ptr(nd, 1)};
            o += nch;
        }
        g.dup_mode = n_dup == 0 ? 0 : (n_dup == total ? 1 : 2);
        g.max_taps = 0;
        for (uint32_t id : g.nodes) g.max_taps = std::max<uint32_t>(g.max_taps, (uint32_t)nodes_[id].in_type.size());   // Mixer: most channels
        hip_check(hipMemcpy(xb.p, ch.data(), ch.size() * sizeof(MixChan), hipMemcpyHostToDevice), "hipMemcpy(mixchan)");
        up(desc_buf(g), d.data(), n * sizeof(MixDesc));
        break;
    }
    case MX_KIND_OSCILLATOR: {
        std::vector<OscDesc> d(n);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_oscillator_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            d[i] = OscDesc{out_ptr(nd, 0), out_ptr(nd, 1), p.freq, p.waveform, 0u};
        }
        up(desc_buf(g), d.data(), n * sizeof(OscDesc));
        break;
    }
    case MX_KIND_STEREO_PANNER: {
        std::vector<PanDesc> d(n);
        for (size_t i = 0; i < n; ++i) { const Node& nd = nodes_[g.nodes[i]]; d[i] = PanDesc{in_ptr(nd, 0, false), in_ptr(nd, 1, false), out_ptr(nd, 0)}; }
        up(desc_buf(g), d.data(), n * sizeof(PanDesc));
        break;
    }
    case MX_KIND_STEREO_SPLITTER: {
        std::vector<SplitDesc> d(n);
        for (size_t i = 0; i < n; ++i) { const Node& nd = nodes_[g.nodes[i]]; d[i] = SplitDesc{in_ptr(nd, 0, false), out_ptr(nd, 0), out_ptr(nd, 1)}; }
        up(desc_buf(g), d.data(), n * sizeof(SplitDesc));
        break;
    }
    case MX_KIND_TRIGGER: {
        std::vector<TrigDesc> d(n);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_trigger_params p; std::memcpy(&p, nd.params.data(), sizeof p);
            d[i] = TrigDesc{out_ptr(nd, 0), p.gate_open ? 1.0f : 0.0f, 0u};   // trigger.rs:38-41
        }
        g.has_gates = true;
        up(desc_buf(g), d.data(), n * sizeof(TrigDesc));
        break;
    }
    case MX_KIND_FIR: {
        size_t tt = 0, th = 0; g.max_taps = 0;
        for (uint32_t id : g.nodes) { mx_fir_params h; std::memcpy(&h, nodes_[id].params.data(), sizeof h); tt += h.n_taps; th += h.n_taps; g.max_taps = std::max(g.max_taps, h.n_taps); }
        std::vector<double> taps(tt);
        if (g.extra.bytes < tt * sizeof(double) || !g.extra.p) g.extra.alloc(tt * sizeof(double));
        if (!g.state.p) { g.state.alloc(th * sizeof(float2)); hip_check(hipMemset(g.state.p, 0, th * sizeof(float2)), "hipMemset"); }
        std::vector<FirDesc> d(n);
        size_t o = 0;
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_fir_params h; std::memcpy(&h, nd.params.data(), sizeof h);
            std::memcpy(&taps[o], nd.params.data() + sizeof h, (size_t)h.n_taps * sizeof(double));
            d[i] = FirDesc{in_ptr(nd, 0, false), out_ptr(nd, 0), (const double*)g.extra.p + o, (float2*)g.state.p + o, h.n_taps, 0u};
            o += h.n_taps;
        }
        hip_check(hipMemcpy(g.extra.p, taps.data(), tt * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(fir taps)");
        up(desc_buf(g), d.data(), n * sizeof(FirDesc));
        break;
    }
    case MX_KIND_RESAMPLE: {
        size_t tt = 0, th = 0; g.max_taps = 0;
        g.rs_tab_doubles = 0; g.rs_win_frames = 0; g.rs_common_up = 0; g.rs_common_taps = 0; g.rs_common_down = 0; bool rs_first = true;
        for (uint32_t id : g.nodes) {
            mx_resample_params h; std::memcpy(&h, nodes_[id].params.data(), sizeof h);
            tt += (size_t)h.up * h.taps_per_phase; th += h.taps_per_phase; g.max_taps = std::max(g.max_taps, h.taps_per_phase);
            g.rs_tab_doubles = (uint32_t)std::min<uint64_t>(0xffffffffu, std::max<uint64_t>(g.rs_tab_doubles, (uint64_t)h.up * h.taps_per_phase));
            g.rs_win_frames = (uint32_t)std::min<uint64_t>(0xffffffffu, std::max<uint64_t>(g.rs_win_frames, (uint64_t)255 * h.down / h.up + 2 + h.taps_per_phase));
            if (rs_first) { g.rs_common_up = h.up; g.rs_common_taps = h.taps_per_phase; g.rs_common_down = h.down; rs_first = false; }   // every node's ratio and taps per phase, where they agree
            else { if (g.rs_common_up != h.up) g.rs_common_up = 0; if (g.rs_common_taps != h.taps_per_phase) g.rs_common_taps = 0; if (g.rs_common_down != h.down) g.rs_common_down = 0; }
        }
        std::vector<double> taps(tt);
        if (g.extra.bytes < tt * sizeof(double) || !g.extra.p) g.extra.alloc(tt * sizeof(double));
        if (!g.state.p) { g.state.alloc(th * sizeof(float2)); hip_check(hipMemset(g.state.p, 0, th * sizeof(float2)), "hipMemset"); }
        std::vector<ResampleDesc> d(n);
        size_t o = 0, oh = 0;
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[g.nodes[i]];
            mx_resample_params h; std::memcpy(&h, nd.params.data(), sizeof h);
            const size_t cnt = (size_t)h.up * h.taps_per_phase;
            std::memcpy(&taps[o], nd.params.data() + sizeof h, cnt * sizeof(double));
            d[i] = ResampleDesc{in_ptr(nd, 0, false), out_ptr(nd, 0), (const double*)g.extra.p + o, (float2*)g.state.p + oh, h.up, h.down, h.taps_per_phase, 0u};
            o += cnt; oh += h.taps_per_phase;
        }
        hip_check(hipMemcpy(g.extra.p, taps.data(), tt * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(resample taps)");
        up(desc_buf(g), d.data(), n * sizeof(ResampleDesc));
        break;
    }
    default: break;  // PLOTTER (jobs built per run), SOURCE_* (no launch)
    }
}

void Graph::upload_group(Group& g) {
    building_main_ = true;
    upload_group_one(g);
    building_main_ = false;
    if (tail_gi_ >= 0) {   // the same descriptors with the double-buffered ports at their second buffer
        building_alt_ = true;
        upload_group_one(g);
        building_alt_ = false;
    }
}

void Graph::build_descriptors() {
    for (Group& g : groups_) upload_group(g);
}

void Graph::update_params(uint32_t node, const void* params, size_t len) {
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    if (len != nodes_[node].params.size()) throw Error(MX_ERR_INVALID, "params_len differs from the node's params (terminal count is frozen with the topology)");
    if (len && !params) throw Error(MX_ERR_INVALID, "params is NULL");
    // a Trigger's params live in the GateBits rows the next run uploads: nothing on the device reads them, nothing has to be waited for (and a Mixer bank that is being
    // held back for the next run stays held)
    if (nodes_[node].kind != MX_KIND_TRIGGER) sync();
    apply_params(node, params, len);
}

// ModuleT::update (src/module/mod.rs:16) on a quiescent stream: new params into the node, descriptors of every launch that
// reads them re-uploaded.  A Trigger only lives in the GateBits rows, which the next run refreshes.
void Graph::apply_params(uint32_t node, const void* params, size_t len) {
    Node& n = nodes_[node];
    if (len) std::memcpy(n.params.data(), params, len);
    if (n.kind == MX_KIND_TRIGGER) { ++gates_version_; return; }
    if (n.kind == MX_KIND_VIDEO_MIXER && n.vmixer) {
        mx_video_mixer_params p; std::memcpy(&p, n.params.data(), sizeof p);
        n.vmixer->update(p);
    }
    if (n.group >= 0) upload_group(groups_[n.group]);
    int32_t o = (int32_t)node;
    while (nodes_[o].elided && nodes_[o].owner >= 0) o = nodes_[o].owner;   // Envelope -> EqThree chains
    if (o != (int32_t)node && nodes_[o].group >= 0) upload_group(groups_[nodes_[o].group]);
}

void Graph::check_schedule(uint32_t node, const void* params, size_t len) const {
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    const Node& n = nodes_[node];
    if (len != n.params.size()) throw Error(MX_ERR_INVALID, "params_len differs from the node's params (terminal count is frozen with the topology)");
    if (len && !params) throw Error(MX_ERR_INVALID, "params is NULL");
}

void Graph::drop_schedules() {
    for (uint32_t id : sched_nodes_) { nodes_[id].sched.clear(); nodes_[id].gate_sched.clear(); }
    if (!sched_nodes_.empty()) ++gates_version_;
    sched_nodes_.clear();
}

void Graph::schedule_params(uint32_t node, uint32_t tick, const void* params, size_t len) {
    check_schedule(node, params, len);
    Node& n = nodes_[node];
    if (n.sched.empty() && n.gate_sched.empty()) sched_nodes_.push_back(node);
    if (n.kind == MX_KIND_TRIGGER) {
        mx_trigger_params tp; std::memcpy(&tp, params, sizeof tp);
        n.gate_sched.emplace_back(tick, tp.gate_open ? 1u : 0u);
        ++gates_version_;
        return;
    }
    Node::SchedEv ev; ev.tick = tick;
    ev.params.assign((const uint8_t*)params, (const uint8_t*)params + len);
    n.sched.push_back(std::move(ev));
}

// H2D copy on the graph's stream out of page-locked staging: the caller's buffer is free on return, nothing waits for the device
void Graph::stage_upload(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    Stage& st = stage_[stage_next_];
    stage_next_ = (stage_next_ + 1) % 4;
    if (st.pending) { hip_check(hipEventSynchronize(st.done), "hipEventSynchronize"); st.pending = false; }
    if (st.cap < bytes) {
        if (st.host) (void)hipHostFree(st.host);
        st.host = nullptr; st.cap = 0;
        hip_check(hipHostMalloc(&st.host, bytes, hipHostMallocDefault), "hipHostMalloc(staging)");
        st.cap = bytes;
    }
    if (!st.done) hip_check(hipEventCreateWithFlags(&st.done, hipEventDisableTiming), "hipEventCreate");
    std::memcpy(st.host, src, bytes);
    // a graph with a second stream: by a kernel, not an SDMA copy -- a copy queued behind a cross-stream wait makes the HOST wait for that event (k_upload); everybody else
    // keeps the copy engine (a tick at a time it is the shorter path: 74 against 84 us per tick of 1024 strips)
    if (tail_gi_ >= 0) launch_upload(dst, st.host, bytes, stream_);
    else hip_check(hipMemcpyAsync(dst, st.host, bytes, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D staged)");
    hip_check(hipEventRecord(st.done, stream_), "hipEventRecord");
    st.pending = true;
}

uint32_t Graph::trigger_of_row(const Group& g, uint32_t row) const {
    const Node& nd = nodes_[g.nodes[row]];
    if (g.kind == MX_KIND_TRIGGER) return g.nodes[row];
    if (g.kind == MX_KIND_ENVELOPE) return nd.fuse_trigger >= 0 ? (uint32_t)nd.fuse_trigger : ~0u;
    if (g.kind == MX_KIND_EQ_THREE) return nd.fuse_env >= 0 && nodes_[nd.fuse_env].fuse_trigger >= 0 ? (uint32_t)nodes_[nd.fuse_env].fuse_trigger : ~0u;
    return ~0u;
}

// GateBits rows of a group for the coming run: bit c of row i = gate_open of row i's Trigger during tick c -- its current params,
// then every scheduled update from its tick on (the reference applies client_update between two ticks, src/engine.rs:192-214)
void Graph::refresh_gates(Group& g, uint32_t run_calls) {
    if (!g.has_gates) return;
    if (g.gates.p && g.gates_version == gates_version_ && g.gates_calls == run_calls) return;
    const uint32_t words = (run_calls + 31) / 32;
    const size_t n = g.nodes.size();
    std::vector<uint32_t> bits(n * words, 0u);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t tr = trigger_of_row(g, (uint32_t)i);
        if (tr == ~0u) continue;
        const Node& T = nodes_[tr];
        mx_trigger_params tp; std::memcpy(&tp, T.params.data(), sizeof tp);
        bool open = tp.gate_open != 0;
        uint32_t c = 0;
        uint32_t* row = bits.data() + i * words;
        auto fill_to = [&](uint32_t end) {   // bits [c, end) = open, a word at a time
            if (open && end > c) {
                uint32_t a = c, b = end;
                while (a < b) {
                    const uint32_t w = a >> 5, lo = a & 31, hi = std::min<uint32_t>(32, lo + (b - a));
                    const uint32_t mask = (hi == 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                    row[w] |= mask;
                    a += hi - lo;
                }
            }
            c = end;
        };
        for (const auto& ev : T.gate_sched) {               // stable by tick: submission order decides among updates for one tick
            const uint32_t at = ev.first < run_calls ? ev.first : run_calls;
            if (at > c) fill_to(at);
            open = ev.second != 0;
        }
        fill_to(run_calls);
    }
    if (g.gates.bytes < bits.size() * sizeof(uint32_t) || !g.gates.p) { sync(); g.gates.alloc(std::max<size_t>(bits.size() * sizeof(uint32_t), 256)); }
    stage_upload(g.gates.p, bits.data(), bits.size() * sizeof(uint32_t));
    g.gate_words = words; g.gates_version = gates_version_; g.gates_calls = run_calls;
}

void Graph::eq_spec_stats(uint64_t out[8]) {
    for (int k = 0; k < 8; ++k) out[k] = 0;
    if (!eq_stats_.p) return;
    sync();
    hip_check(hipMemcpy(out, eq_stats_.p, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy(eq stats)");
}

void Graph::write_source(uint32_t node, const float* host, size_t frames) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    Node& n = nodes_[node];
    if (n.kind != MX_KIND_SOURCE_MONO && n.kind != MX_KIND_SOURCE_STEREO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_*");
    if (n.bound) throw Error(MX_ERR_INVALID, "source is bound to a caller device buffer");
    if (frames > cap_frames_) throw Error(MX_ERR_INVALID, "more ticks than max_ticks_per_run");
    if (frames && !host) throw Error(MX_ERR_INVALID, "host_samples is NULL");
    const size_t fl = floats_per_frame(n.out_type[0]) * frames;
    hip_check(hipMemcpyAsync(out_ptr(n, 0), host, fl * sizeof(float), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D source)");
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");  // host buffer is the caller's again on return
}

void Graph::bind_source(uint32_t node, const void* dev) {
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    Node& n = nodes_[node];
    if (n.kind != MX_KIND_SOURCE_MONO && n.kind != MX_KIND_SOURCE_STEREO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_*");
    if (dev && ((uintptr_t)dev & 15)) throw Error(MX_ERR_INVALID, "bound device buffer must be 16-byte aligned");
    sync();
    n.bound = (const float*)dev;
    build_descriptors();
}

void Graph::set_input_enabled(uint32_t node, uint32_t port, bool enabled) {
    if (node >= nodes_.size() || port >= nodes_[node].in_src.size()) throw Error(MX_ERR_INVALID, "input terminal out of range");
    Node& n = nodes_[node];
    if (n.in_src_orig.size() != n.in_src.size()) return;  // node is outside the run order
    const PortRef want = enabled ? n.in_src_orig[port] : PortRef{};
    if (want.node == n.in_src[port].node && want.port == n.in_src[port].port) return;
    sync();
    n.in_src[port] = want;
    if (n.group >= 0) upload_group(groups_[n.group]);
}

void Graph::sync() {
    hip_check(hipSetDevice(device_), "hipSetDevice");   // the current device is per thread: a graph may be driven from another thread than its creator's
    flush_scales(stream_);
    flush_deferred_tail(false);
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    if (tail_stream_) { hip_check(hipStreamSynchronize(tail_stream_), "hipStreamSynchronize"); tail_pending_[0] = tail_pending_[1] = false; }
}

void Graph::ensure_capacity(size_t frames) {
    if (frames <= cap_frames_) return;
    sync();
    cap_frames_ = frames;
    layout_slab();
    build_descriptors();
}

static bool group_launches(const Group& g);

void Graph::run(uint64_t t0, size_t fpc, uint32_t n_calls, float* ms_by_kind, float* ms_total) {
    const size_t frames = fpc * (size_t)n_calls;
    auto drop_schedules = [&] { for (uint32_t id : sched_nodes_) { nodes_[id].sched.clear(); nodes_[id].gate_sched.clear(); } sched_nodes_.clear(); };
    if (frames > cap_frames_) { drop_schedules(); throw Error(MX_ERR_INVALID, "n_ticks exceeds max_ticks_per_run"); }
    if (n_calls == 0 || fpc == 0) { drop_schedules(); last_calls_ = n_calls; last_frames_per_call_ = fpc; return; }
    hip_check(hipSetDevice(device_), "hipSetDevice");

    // ---- scheduled parameter updates (Engine::client_update between two ticks, src/engine.rs:192-214,277-398) ----
    // Trigger updates travel as one gate bit per tick and cost nothing.  Any other module's update cuts the run into spans:
    // the update is applied (ModuleT::update) between the span that ends before its tick and the span that starts with it.
    std::vector<uint32_t> cuts;   // span starts > 0
    bool any_sched = false;
    const char* beyond = "a scheduled parameter update lies beyond the run (tick_in_run >= n_ticks)";
    for (uint32_t sid : sched_nodes_) {
        Node& n = nodes_[sid];
        if (!n.gate_sched.empty()) {
            any_sched = true;
            bool sorted = true;
            for (size_t i = 1; i < n.gate_sched.size(); ++i) sorted = sorted && n.gate_sched[i - 1].first <= n.gate_sched[i].first;
            if (!sorted) std::stable_sort(n.gate_sched.begin(), n.gate_sched.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            if (n.gate_sched.back().first >= n_calls) { drop_schedules(); throw Error(MX_ERR_INVALID, beyond); }
        }
        if (n.sched.empty()) continue;
        any_sched = true;
        std::stable_sort(n.sched.begin(), n.sched.end(), [](const Node::SchedEv& x, const Node::SchedEv& y) { return x.tick < y.tick; });
        if (n.sched.back().tick >= n_calls) { drop_schedules(); throw Error(MX_ERR_INVALID, beyond); }
        for (const Node::SchedEv& ev : n.sched) if (ev.tick) cuts.push_back(ev.tick);
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    auto apply_at = [&](uint32_t tick) {   // every non-Trigger update scheduled for `tick`, in submission order
        bool any = false;
        for (uint32_t id : sched_nodes_) {
            Node& n = nodes_[id];
            if (n.kind == MX_KIND_TRIGGER) continue;
            for (const Node::SchedEv& ev : n.sched) if (ev.tick == tick) { if (!any) { sync(); any = true; } apply_params(id, ev.params.data(), ev.params.size()); }
        }
    };
    if (any_sched) apply_at(0);
    for (uint32_t id : video_order_) if (nodes_[id].kind == MX_KIND_MONITOR) nodes_[id].mon_ticks.clear();

    // Plotter bookkeeping is host logic (plotter.rs:37-40): count += 1 per call, fire on every 6th
    size_t total_fired = 0;
    for (uint32_t pid : plotter_nodes_) {
        Node& n = nodes_[pid];
        n.plot_fired.assign(n_calls, 0);
        n.plot_slot.assign(n_calls, -1);
        const bool connected = n.in_src[0].node >= 0;
        for (uint32_t c = 0; c < n_calls; ++c) {
            n.plot_count += 1;
            if (n.plot_count % 6 == 0 && connected) { n.plot_fired[c] = 1; n.plot_slot[c] = (int32_t)total_fired++; }
        }
    }
    if (total_fired) {
        const size_t need = total_fired * 2 * fpc * sizeof(float);
        if (plot_stage_.bytes < need) { sync(); plot_stage_.alloc(need); }
        if (plot_jobs_.bytes < total_fired * sizeof(PlotJob)) { sync(); plot_jobs_.alloc(total_fired * sizeof(PlotJob)); }
    }
    plot_job_off_ = 0;
    for (Group& g : groups_) refresh_gates(g, n_calls);

    const bool prof = ms_by_kind != nullptr || prof_on_;
    prof_this_run_ = prof;
    // MX_FLAG_OVERLAP_TAIL: an uncut run alternates the double-buffered ports and leaves its tail on the second stream; before its
    // earlier groups overwrite a buffer, the tail that last read THAT buffer (two runs ago) must be done -- not the previous run's
    // (automatic mode: a run of a tick or a few on a graph built for long submissions stays on one stream -- two cross-stream events cost 13 us of an 75 us tick)
    for (hipEvent_t e : head_waits_) hip_check(hipStreamWaitEvent(stream_, e, 0), "hipStreamWaitEvent");
    head_waits_.clear();
    overlap_this_run_ = tail_gi_ >= 0 && cuts.empty() && (!tail_auto_ || n_calls >= 16);
    if (overlap_this_run_) { parity_ ^= 1u; wait_tail((int)parity_); }
    else if (tail_gi_ >= 0) wait_tail(-1);
    if (cuts.empty()) {
        run_span(t0, fpc, 0, n_calls, n_calls);
    } else {
        cuts.push_back(n_calls);
        uint32_t from = 0;
        for (uint32_t to : cuts) {
            if (from) {
                apply_at(from);
                sync();
                run_off_frames_ = (size_t)from * fpc;
                build_descriptors();                       // every port pointer moves to the span's first tick
            }
            run_span(t0 + (uint64_t)from * fpc, fpc, from, to - from, n_calls);
            from = to;
        }
        sync();
        run_off_frames_ = 0;
        build_descriptors();
    }
    // the modules keep the last scheduled params (a Trigger's are read by the next run's GateBits)
    if (any_sched) {
        for (uint32_t sid : sched_nodes_) {
            Node& n = nodes_[sid];
            if (!n.gate_sched.empty()) {
                mx_trigger_params tp{}; tp.gate_open = n.gate_sched.back().second;
                std::memcpy(n.params.data(), &tp, sizeof tp);
                ++gates_version_;
                n.gate_sched.clear();
            }
            n.sched.clear();
        }
        sched_nodes_.clear();
    }
    hip_check(hipGetLastError(), "kernel launch");
    last_calls_ = n_calls;
    last_frames_per_call_ = fpc;
    if (prof) ++prof_runs_count_;
    if (ms_by_kind) (void)profile_collect(ms_by_kind, ms_total);
}

// ticks [call_off, call_off + n_calls) of the current run: one launch per (level, kind, domain) group, then the video section
void Graph::run_span(uint64_t t0, size_t fpc, uint32_t call_off, uint32_t n_calls, uint32_t run_calls) {
    const size_t frames = fpc * (size_t)n_calls;
    const bool prof = prof_this_run_;
    std::vector<hipEvent_t> ev;
    if (prof) {
        if (!prof_pool_.empty()) { ev = std::move(prof_pool_.back()); prof_pool_.pop_back(); }
        else {
            ev.resize(groups_.size() + 3);   // one slot per launch group + the per-tick video section + the begin of a tail launch that was held back (its own stream)
            for (auto& e : ev) hip_check(hipEventCreate(&e), "hipEventCreate");
        }
        hip_check(hipEventRecord(ev[0], stream_), "hipEventRecord");
    }
    std::vector<PlotJob> jobs;
    size_t gi = 0;
    for (Group& g : groups_) {
        const uint32_t n = (uint32_t)g.nodes.size();
        const size_t gf = frames * g.dom_num / g.dom_den;   // frames of this group's sample-rate domain
        const size_t gfpc = fpc * g.dom_num / g.dom_den;    // ... per tick
        const GateBits gates{(const uint32_t*)g.gates.p, g.gate_words, call_off};
        switch (g.kind) {
        case MX_KIND_AMPLIFIER: launch_amplifier((const AmpDesc*)desc_of(g), n, gf, stream_, fp_contract()); break;
        case MX_KIND_ENVELOPE: {
            const size_t need = envelope_scratch_bytes(n, gf);       // long streams: marker bitmaps and per-segment states (mx_k_envelope.hip)
            if (need && (g.spec.bytes < need || !g.spec.p)) { sync(); g.spec.alloc(need); }
            launch_envelope((const EnvDesc*)desc_of(g), (EnvState*)g.state.p, n, gf, gfpc, gates, t0, sample_rate_, stream_, fp_contract(), need ? g.spec.p : nullptr, need ? g.spec.bytes : 0);
            break;
        }
        case MX_KIND_EQ_THREE: {
            EqRun r{gf, gfpc, n_calls, fp_contract() ? 1u : 0u, t0, sample_rate_, 1.0 / sample_rate_, lo_f_, hi_f_, nullptr};
            if (g.state2.p) {   // Envelopes folded into the epilogue: their state entering every tick of this span
                const size_t need = (size_t)n * n_calls * sizeof(EnvTick);
                if (g.env_ticks.bytes < need || !g.env_ticks.p) { sync(); g.env_ticks.alloc(need); }
                launch_env_ticks((const EnvTickDesc*)g.tick_desc.p, n, gates, n_calls, gfpc, t0, sample_rate_, (EnvTick*)g.env_ticks.p, stream_, fp_contract());
                r.ticks = (const EnvTick*)g.env_ticks.p;
            }
            if (eq_exact()) {
                EqSpecPlan plan;
                if (eq_plan_spec(n, gf, gfpc, lo_f_, hi_f_, plan, g.eq_mode == 6 || g.eq_mode == 7, g.eq_mode == 4 || g.eq_mode == 5)) {   // long streams: speculative time-parallel form, verified bit-exact
                    const size_t need = eq_spec_scratch_bytes(n, plan);
                    if (g.spec.bytes < need || !g.spec.p) { sync(); g.spec.alloc(need); }
                    if (!eq_stats_.p) { eq_stats_.alloc(8 * sizeof(uint64_t)); hip_check(hipMemset(eq_stats_.p, 0, 8 * sizeof(uint64_t)), "hipMemset"); }
                    gate_armed_ = false;
                    if (deferred_.pending) {
                        if (!gate_flag_.p) { gate_flag_.alloc(64); hip_check(hipMemset(gate_flag_.p, 0, 64), "hipMemset"); }
                        r.started = (uint32_t*)gate_flag_.p; r.started_seq = ++gate_seq_; gate_armed_ = true;
                    }
                    const bool opens_gate = launch_eq_three_spec((const EqDesc*)desc_of(g), (EqState*)g.state.p, n, r, plan, g.eq_mode, g.spec.p, (uint64_t*)eq_stats_.p, stream_);
                    if (!opens_gate && !(getenv("MX_TAIL_GATE_TEST") && atoi(getenv("MX_TAIL_GATE_TEST")))) gate_armed_ = false;   // the direct form never stores the flag: a gate would spin to its time limit before the bank starts (MX_TAIL_GATE_TEST: tests of that bounded spin)
                    if (deferred_.pending) flush_deferred_tail(true);     // run k's Mixer bank: behind the gate this launch opens
                } else {
                    void* scratch = nullptr;
                    if (eq_use_poles_split(n, gf)) {     // short streams, few instances: two lanes per instance + a sample-parallel epilogue kernel
                        const size_t need = eq_poles_scratch_bytes(n, gf);
                        if (g.spec.bytes < need || !g.spec.p) { sync(); g.spec.alloc(need); }
                        scratch = g.spec.p;
                    }
                    launch_eq_three_exact((const EqDesc*)desc_of(g), (EqState*)g.state.p, n, r, scratch, stream_);
                }
            } else {
                EqSplit sp{1u, 5u, 0u, 0u, gf, gf, nullptr, nullptr, nullptr};
                EqSpanPow pp{};
                eq_plan_split(n, gf, lo_f_, hi_f_, sp);
                if (sp.n_split > 1) {   // few instances, long streams: cut each stream into spans for different workgroups
                    const size_t need = (size_t)n * sp.n_split * 8 * sizeof(double) + (size_t)n * 12 * sizeof(double);
                    if (g.extra.bytes < need || !g.extra.p) { sync(); g.extra.alloc(need); }
                    sp.zbuf = (double*)g.extra.p;                         // [n][n_split][8] zero-state span end states
                    sp.bound = sp.zbuf + (size_t)n * sp.n_split * 8;      // [n][12] snapshot of the carried EqState
                    toeplitz_pow((long double)lo_f_, sp.span, pp.lo);
                    toeplitz_pow((long double)hi_f_, sp.span, pp.hi);
                }
                launch_eq_three_scan((const EqDesc*)desc_of(g), (EqState*)g.state.p, n, r, (const EqScanTab*)eq_tabs_.p, sp, pp, stream_);
            }
            break;
        }
        case MX_KIND_FM_SINE: launch_fm_sine((const FmDesc*)desc_of(g), n, gf, t0, sample_rate_, stream_, sin_mode_); break;
        case MX_KIND_MIXER:
            if (overlap_this_run_ && (int)gi >= tail_gi_) {   // beside the next run's earlier groups (HBM-bound beside VALU-bound)
                const bool first = (int)gi == tail_gi_, last = gi + 1 == groups_.size();
                if (first) {
                    if (deferred_.pending) flush_deferred_tail(false);   // (a run whose earlier groups had no speculative EqThree launch: nothing opened a gate)
                    hip_check(hipEventRecord(ev_head_done_, stream_), "hipEventRecord");
                    if (tail_gate_ < 0) { const char* e = getenv("MX_TAIL_GATE"); tail_gate_ = e && atoi(e) == 0 ? 0 : 1; }   // A/B: 0 = launched at once (round 4's form)
                    if (tail_gate_) { deferred_.items.clear(); deferred_.parity = parity_; deferred_.prof_begin = prof ? ev[groups_.size() + 2] : nullptr; tail_held_this_span_ = true; }
                    else hip_check(hipStreamWaitEvent(tail_stream_, ev_head_done_, 0), "hipStreamWaitEvent");
                }
                if (tail_gate_) {
                    deferred_.items.push_back(TailLaunch{desc_of(g), n, g.max_taps, gf, g.dup_mode, prof ? ev[gi + 1] : nullptr});
                    if (last) deferred_.pending = true;
                } else {
                    launch_mixer((const MixDesc*)desc_of(g), n, g.max_taps, gf, g.dup_mode, tail_stream_);
                    if (prof) hip_check(hipEventRecord(ev[gi + 1], tail_stream_), "hipEventRecord");
                    if (last) { hip_check(hipEventRecord(ev_tail_done_[parity_], tail_stream_), "hipEventRecord"); tail_pending_[parity_] = true; }
                }
                ++gi;
                continue;
            }
            launch_mixer((const MixDesc*)desc_of(g), n, g.max_taps /* = most channels */, gf, g.dup_mode, stream_);
            break;
        case MX_KIND_OSCILLATOR: launch_oscillator((const OscDesc*)desc_of(g), n, gf, t0, sample_rate_, stream_, sin_mode_); break;
        case MX_KIND_STEREO_PANNER: launch_panner((const PanDesc*)desc_of(g), n, gf, stream_); break;
        case MX_KIND_STEREO_SPLITTER: launch_splitter((const SplitDesc*)desc_of(g), n, gf, stream_); break;
        case MX_KIND_TRIGGER: launch_trigger((const TrigDesc*)desc_of(g), n, gf, gfpc, &gates, stream_); break;
        case MX_KIND_FIR: launch_fir((const FirDesc*)desc_of(g), n, g.max_taps, gf, stream_, fp_contract()); break;
        case MX_KIND_RESAMPLE:
            launch_resample((const ResampleDesc*)desc_of(g), n, g.max_taps, g.rs_tab_doubles, g.rs_win_frames, frames * g.in_dom_num / g.in_dom_den, gf,
                            t0 * g.in_dom_num / g.in_dom_den, t0 * g.dom_num / g.dom_den, stream_, g.rs_common_up, fp_contract(), g.rs_common_taps, g.rs_common_down);
            break;
        case MX_KIND_PLOTTER: {
            jobs.clear();
            for (uint32_t id : g.nodes) {
                const Node& nd = nodes_[id];
                for (uint32_t c = 0; c < n_calls; ++c) {
                    if (!nd.plot_fired[call_off + c]) continue;
                    float* stage = (float*)plot_stage_.p + (size_t)nd.plot_slot[call_off + c] * 2 * fpc;
                    jobs.push_back(PlotJob{in_ptr(nd, 0, false) + (size_t)c * 2 * fpc, stage, stage + fpc});
                }
            }
            if (!jobs.empty()) {
                PlotJob* dst = (PlotJob*)plot_jobs_.p + plot_job_off_;
                stage_upload(dst, jobs.data(), jobs.size() * sizeof(PlotJob));
                launch_plotter(dst, (uint32_t)jobs.size(), fpc, stream_);
                plot_job_off_ += jobs.size();
            }
            break;
        }
        default: break;
        }
        // an event costs ~5 us of stream time: none for groups that launch nothing (sources, video kinds)
        if (prof && group_launches(g)) hip_check(hipEventRecord(ev[gi + 1], stream_), "hipEventRecord");
        ++gi;
    }
    (void)run_calls;
    // video sub-graph: tick by tick (frames arrive per tick; nothing to batch over time)
    if (has_video_) {
        for (uint32_t c = 0; c < n_calls; ++c) run_video_tick(t0 + (uint64_t)c * fpc);
        flush_scales(stream_);
        for (uint32_t id : video_order_) { Node& vn = nodes_[id]; if (!vn.rgba_pending.empty()) launch_pending_rgba(vn, vn.rgba_pending.size(), false); vn.rgba_calls = 0; }   // the last ticks' sinks
    }
    if (prof && has_video_) hip_check(hipEventRecord(ev[groups_.size() + 1], stream_), "hipEventRecord");
    if (prof) { prof_runs_.push_back(std::move(ev)); prof_runs_held_.push_back(tail_held_this_span_); }
    tail_held_this_span_ = false;
}

static bool group_launches(const Group& g) {
    switch (g.kind) {
    case MX_KIND_SOURCE_MONO: case MX_KIND_SOURCE_STEREO: case MX_KIND_SOURCE_VIDEO:
    case MX_KIND_VIDEO_MIXER: case MX_KIND_VIDEO_TO_RGBA: case MX_KIND_MONITOR: return false;   // bound buffers / the per-tick video section
    default: return true;
    }
}

void Graph::profile_enable(bool on) { prof_on_ = on; }

uint32_t Graph::profile_collect(float* ms_by_kind, float* ms_total) {
    sync();
    if (ms_by_kind) for (int k = 0; k < MX_KIND_COUNT; ++k) ms_by_kind[k] = 0.f;
    if (ms_total) *ms_total = 0.f;
    const uint32_t n = prof_runs_count_;   // run() calls; a run cut into spans recorded one event list per span
    prof_runs_count_ = 0;
    size_t run_i = 0;
    for (auto& ev : prof_runs_) {
        const bool held = run_i < prof_runs_held_.size() && prof_runs_held_[run_i]; ++run_i;
        perf_group_ms_.assign(groups_.size() + 1, 0.f);
        size_t last = 0;   // index of the latest event that was recorded in this run
        for (size_t i = 0; i + 1 < ev.size() && i <= groups_.size(); ++i) {
            const bool recorded = i < groups_.size() ? group_launches(groups_[i]) : has_video_;
            if (!recorded) continue;
            float ms = 0.f;
            if (held && i < groups_.size() && (int)i >= tail_gi_) {
                // a tail launch that was held back ran on its own stream, inside the NEXT run's window: its own begin (the tail's, or the tail group's before it) and end;
                // the run's total ends where its stream's work did
                hip_check(hipEventElapsedTime(&ms, (int)i == tail_gi_ ? ev[groups_.size() + 2] : ev[i], ev[i + 1]), "hipEventElapsedTime");
                if (ms_by_kind) ms_by_kind[groups_[i].kind] += ms;
                perf_group_ms_[i] = ms;
                continue;
            }
            hip_check(hipEventElapsedTime(&ms, ev[last], ev[i + 1]), "hipEventElapsedTime");
            if (ms_by_kind) ms_by_kind[i < groups_.size() ? groups_[i].kind : (uint32_t)MX_KIND_VIDEO_MIXER] += ms;
            perf_group_ms_[i] = ms;
            last = i + 1;
        }
        { float ms = 0.f; hip_check(hipEventElapsedTime(&ms, ev.front(), ev[last]), "hipEventElapsedTime"); if (ms_total) *ms_total += ms; perf_total_ms_ = ms; }
        perf_calls_ = last_calls_;
        if (perf_calls_ && perf_total_ms_ * 1000.0 / perf_calls_ > 1e6 / (double)tps_)   // timing.rs:37-40: the tick ran over its budget
            perf_last_lag_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        prof_pool_.push_back(std::move(ev));
    }
    prof_runs_.clear(); prof_runs_held_.clear();
    return n;
}

// -------------------------------------------------------------------------------------------------
// state that a module carries from tick to tick, as device regions in a canonical order per kind
// -------------------------------------------------------------------------------------------------
std::vector<Graph::StateLoc> Graph::state_locs(uint32_t id) const {
    std::vector<StateLoc> out;
    const Node& n = nodes_[id];
    switch (n.kind) {
    case MX_KIND_EQ_THREE:
        if (n.group >= 0 && groups_[n.group].state.p) out.push_back({(EqState*)groups_[n.group].state.p + n.slot, sizeof(EqState)});
        break;
    case MX_KIND_ENVELOPE:
        if (n.elided && n.owner >= 0 && nodes_[n.owner].kind == MX_KIND_EQ_THREE) {   // evaluated inline by the EqThree kernel
            const Node& e = nodes_[n.owner];
            if (e.group >= 0 && groups_[e.group].state2.p) out.push_back({(EnvState*)groups_[e.group].state2.p + e.slot, sizeof(EnvState)});
        } else if (n.group >= 0 && groups_[n.group].state.p) {
            out.push_back({(EnvState*)groups_[n.group].state.p + n.slot, sizeof(EnvState)});
        }
        break;
    case MX_KIND_FIR: case MX_KIND_RESAMPLE:
        if (n.group >= 0 && groups_[n.group].state.p) {
            size_t off = 0, mine = 0;
            for (uint32_t other : groups_[n.group].nodes) {
                size_t h;
                if (n.kind == MX_KIND_FIR) { mx_fir_params q; std::memcpy(&q, nodes_[other].params.data(), sizeof q); h = q.n_taps; }
                else { mx_resample_params q; std::memcpy(&q, nodes_[other].params.data(), sizeof q); h = q.taps_per_phase; }
                if (other == id) { mine = h; break; }
                off += h;
            }
            out.push_back({(float2*)groups_[n.group].state.p + off, mine * sizeof(float2)});
        }
        break;
    default: break;
    }
    return out;
}

void Graph::adopt_state(Graph& old, const int32_t* old_of_new, size_t n) {
    if (n != nodes_.size()) throw Error(MX_ERR_INVALID, "old_node_of_new must have one entry per node of the new graph");
    if (n && !old_of_new) throw Error(MX_ERR_INVALID, "old_node_of_new is NULL");
    for (size_t i = 0; i < n; ++i) {
        const int32_t j = old_of_new[i];
        if (j < 0) continue;
        if ((size_t)j >= old.nodes_.size()) throw Error(MX_ERR_INVALID, "old node out of range");
        if (old.nodes_[j].kind != nodes_[i].kind) throw Error(MX_ERR_TYPE, "a module cannot change kind across a topology edit");
    }
    old.sync(); sync();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    std::vector<CopyJob> jobs;   // thousands of small states (a 1024-strip graph: 2048 of 24 - 136 bytes): one launch, not one copy each
    for (size_t i = 0; i < n; ++i) {
        const int32_t j = old_of_new[i];
        if (j < 0) continue;
        const auto a = state_locs((uint32_t)i), b = old.state_locs((uint32_t)j);
        for (size_t k = 0; k < a.size() && k < b.size(); ++k)
            if (a[k].bytes == b[k].bytes && a[k].bytes)   // a filter whose length changed starts from silence
                jobs.push_back(CopyJob{a[k].p, b[k].p, a[k].bytes});
    }
    DevBuf job_buf;
    if (!jobs.empty()) {
        job_buf.alloc(jobs.size() * sizeof(CopyJob));
        hip_check(hipMemcpyAsync(job_buf.p, jobs.data(), jobs.size() * sizeof(CopyJob), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(adopt jobs)");
        launch_copy_jobs((const CopyJob*)job_buf.p, (uint32_t)jobs.size(), stream_);
        sync();                   // `jobs` and job_buf go out of scope
    }
    for (size_t i = 0; i < n; ++i) {
        const int32_t j = old_of_new[i];
        if (j < 0) continue;
        Node& nn = nodes_[i]; Node& on = old.nodes_[j];
        if (nn.kind == MX_KIND_PLOTTER) nn.plot_count = on.plot_count;          // plotter.rs:37-40
        if (nn.kind == MX_KIND_MONITOR) { nn.mon_has_epoch = on.mon_has_epoch; nn.mon_epoch = on.mon_epoch; nn.mon_queued = on.mon_queued; }   // Monitor.epoch (monitor.rs:122)
        if (nn.kind == MX_KIND_VIDEO_MIXER && on.vmixer) {                       // stored frames, scalers, expiry times
            mx_video_mixer_params p; std::memcpy(&p, nn.params.data(), sizeof p);
            nn.vmixer = std::move(on.vmixer);
            nn.vmixer->rebind(stream_, nn.vlazy, tps_);   // the old graph's stream may be gone after this call; this graph's fusion plan and tick rate apply
            nn.vmixer->update(p);
        }
        if (nn.kind == MX_KIND_SOURCE_VIDEO) { nn.vsrc_ring = on.vsrc_ring; nn.vsrc_ring_pos = on.vsrc_ring_pos; nn.vsrc_sched = on.vsrc_sched; nn.vband = on.vband; nn.vband_pool = on.vband_pool; nn.vsrc = on.vsrc; nn.vsrc_dur = on.vsrc_dur; nn.vsrc_off = on.vsrc_off; nn.vsrc_repeat = on.vsrc_repeat; nn.vsrc_pending = on.vsrc_pending; }
    }
    hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

Graph::Perf Graph::performance_info(uint64_t* module_us, size_t cap) {
    if (cap < nodes_.size() && module_us) throw Error(MX_ERR_INVALID, "module_us is shorter than the node count");
    Perf pf{};
    pf.tick_rate = tps_;
    pf.tick_budget_us = 1000000ull / tps_;                                          // timing.rs:9
    const double calls = perf_calls_ ? (double)perf_calls_ : 1.0;
    const double tick_us = perf_total_ms_ * 1000.0 / calls;
    pf.realtime = perf_calls_ != 0 && tick_us < (double)pf.tick_budget_us;         // timing.rs:33
    if (perf_last_lag_s_ >= 0.0) {                                                  // util.rs:47-60
        const double since = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - perf_last_lag_s_;
        pf.lag = since < 0.1 ? 2 : (since < 5.0 ? 1 : 0);
    }
    if (module_us) for (size_t i = 0; i < nodes_.size(); ++i) module_us[i] = 0;
    double accounted = 0.0;
    for (size_t gi = 0; gi < groups_.size() && gi < perf_group_ms_.size(); ++gi) {
        // one launch serves every module of the group and the modules folded into them: the launch's time per tick is
        // split evenly over all of those (PerformanceAccount::Module, timing.rs:86-94)
        std::vector<uint32_t> members;
        for (uint32_t id : groups_[gi].nodes) {
            members.push_back(id);
            const Node& nd = nodes_[id];
            for (int32_t f : {nd.fuse_pan, nd.fuse_amp, nd.fuse_env, nd.fuse_trigger}) if (f >= 0) members.push_back((uint32_t)f);
            if (nd.fuse_env >= 0 && nodes_[nd.fuse_env].fuse_trigger >= 0) members.push_back((uint32_t)nodes_[nd.fuse_env].fuse_trigger);
        }
        if (members.empty()) continue;
        const double us = perf_group_ms_[gi] * 1000.0 / calls;
        accounted += us;
        if (module_us) for (uint32_t id : members) module_us[id] += (uint64_t)(us / (double)members.size() + 0.5);
    }
    if (has_video_ && perf_group_ms_.size() > groups_.size()) {   // the per-tick video section
        std::vector<uint32_t> members;
        for (size_t i = 0; i < nodes_.size(); ++i) if (nodes_[i].kind == MX_KIND_VIDEO_MIXER || nodes_[i].kind == MX_KIND_VIDEO_TO_RGBA || nodes_[i].kind == MX_KIND_MONITOR) members.push_back((uint32_t)i);
        const double us = perf_group_ms_[groups_.size()] * 1000.0 / calls;
        accounted += us;
        if (module_us) for (uint32_t id : members) module_us[id] += (uint64_t)(us / (double)members.size() + 0.5);
    }
    pf.engine_us = (uint64_t)std::max(0.0, tick_us - accounted + 0.5);             // PerformanceAccount::Engine = tick - modules (timing.rs:43)
    return pf;
}

void Graph::read_output(uint32_t node, uint32_t port, float* host, size_t frames, size_t first_frame) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    wait_tail(-1);
    if (node >= nodes_.size() || port >= nodes_[node].out_type.size()) throw Error(MX_ERR_INVALID, "output terminal out of range");
    if (first_frame > cap_frames_ || frames > cap_frames_ - first_frame) throw Error(MX_ERR_INVALID, "more ticks than max_ticks_per_run");
    if (frames && !host) throw Error(MX_ERR_INVALID, "host_samples is NULL");
    const Node& n = nodes_[node];
    if (n.out_elided[port]) throw Error(MX_ERR_INVALID, "port is not materialised: it only feeds a fused consumer (build with MX_FLAG_NO_FUSE to observe it)");
    // the port's own sample-rate domain; a window starts and ends where whole ticks do
    const size_t f0 = first_frame * n.dom_num / n.dom_den;
    frames = (first_frame + frames) * n.dom_num / n.dom_den - f0;
    if (n.out_dup[port]) {   // stored as one float per frame (L == R): expand for the caller
        hip_check(hipMemcpyAsync(host, out_ptr(n, port) + f0, frames * sizeof(float), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
        sync();
        for (size_t i = frames; i-- > 0;) { const float v = host[i]; host[2 * i] = v; host[2 * i + 1] = v; }
        return;
    }
    const size_t fpf = floats_per_frame(n.out_type[port]);
    hip_check(hipMemcpyAsync(host, out_ptr(n, port) + fpf * f0, fpf * frames * sizeof(float), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
    sync();
}

void Graph::read_output_i16(uint32_t node, uint32_t port, int16_t* host, size_t frames) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    wait_tail(-1);
    if (node >= nodes_.size() || port >= nodes_[node].out_type.size()) throw Error(MX_ERR_INVALID, "output terminal out of range");
    if (frames > cap_frames_) throw Error(MX_ERR_INVALID, "more ticks than max_ticks_per_run");
    if (frames && !host) throw Error(MX_ERR_INVALID, "host_samples is NULL");
    const Node& n = nodes_[node];
    if (n.out_elided[port]) throw Error(MX_ERR_INVALID, "port is not materialised: it only feeds a fused consumer (build with MX_FLAG_NO_FUSE to observe it)");
    const size_t cnt = floats_per_frame(n.out_type[port]) * (frames * n.dom_num / n.dom_den);
    if (!cnt) return;
    if (conv_stage_.bytes < cnt * sizeof(int16_t)) { sync(); conv_stage_.alloc(cnt * sizeof(int16_t)); }
    launch_f32_to_i16(out_ptr(n, port), (int16_t*)conv_stage_.p, cnt, n.out_dup[port] ? 1 : 0, stream_);
    hip_check(hipMemcpyAsync(host, conv_stage_.p, cnt * sizeof(int16_t), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H i16)");
    sync();
}

void Graph::write_source_i16(uint32_t node, const int16_t* host, size_t frames) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    Node& n = nodes_[node];
    if (n.kind != MX_KIND_SOURCE_MONO && n.kind != MX_KIND_SOURCE_STEREO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_*");
    if (n.bound) throw Error(MX_ERR_INVALID, "source is bound to a caller device buffer");
    if (frames > cap_frames_) throw Error(MX_ERR_INVALID, "more ticks than max_ticks_per_run");
    if (frames && !host) throw Error(MX_ERR_INVALID, "host_samples is NULL");
    const size_t cnt = floats_per_frame(n.out_type[0]) * frames;
    if (!cnt) return;
    if (conv_stage_.bytes < cnt * sizeof(int16_t)) { sync(); conv_stage_.alloc(cnt * sizeof(int16_t)); }
    hip_check(hipMemcpyAsync(conv_stage_.p, host, cnt * sizeof(int16_t), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D i16)");
    launch_i16_to_f32((const int16_t*)conv_stage_.p, out_ptr(n, 0), cnt, stream_);
    sync();   // host buffer is the caller's again on return
}

void* Graph::debug_eq_records(size_t* bytes) const {
    for (const Group& g : groups_)
        if (g.kind == MX_KIND_EQ_THREE && g.spec.p) { if (bytes) *bytes = g.spec.bytes; return g.spec.p; }
    if (bytes) *bytes = 0;
    return nullptr;
}

float* Graph::output_ptr(uint32_t node, uint32_t port, size_t* fpf, bool stream_ordered_consumer) {
    if (node >= nodes_.size() || port >= nodes_[node].out_type.size()) throw Error(MX_ERR_INVALID, "output terminal out of range");
    if (nodes_[node].out_elided[port]) throw Error(MX_ERR_INVALID, "port is not materialised: it only feeds a fused consumer (build with MX_FLAG_NO_FUSE to observe it)");
    if (nodes_[node].out_dup[port]) throw Error(MX_ERR_INVALID, "port is stored as one float per frame (L == R fused result): use mx_graph_read_output, or build with MX_FLAG_NO_FUSE");
    if (fpf) *fpf = floats_per_frame(nodes_[node].out_type[port]) * (spt_ * nodes_[node].dom_num / nodes_[node].dom_den);   // floats per TICK in the port's own rate domain
    // A consumer that takes the raw pointer of a bus reads it in stream order on stream(): a Mixer bank the library moved to the second stream ON ITS OWN (short submissions,
    // MX_OVERLAP_AUTO) would not be ordered before it -- so the automatism ends here, for good.  (A host that asked for MX_FLAG_OVERLAP_TAIL knows about mx_graph_tail_stream.)
    // The same holds for a port the tail READS: it is double-buffered while the mode is on, and a raw pointer would see fresh data only every other run.
    if (stream_ordered_consumer && tail_gi_ >= 0 && tail_auto_ && (nodes_[node].group >= tail_gi_ || nodes_[node].out_off2[port] != SIZE_MAX)) end_auto_tail();
    return out_ptr(nodes_[node], port);
}

int Graph::read_plotter(uint32_t node, uint32_t call, float* left, float* right) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_PLOTTER) throw Error(MX_ERR_INVALID, "node is not a Plotter");
    const Node& n = nodes_[node];
    if (call >= n.plot_fired.size()) throw Error(MX_ERR_INVALID, "tick_in_run is outside the last run");
    if (!n.plot_fired[call]) return 0;
    const size_t fpc = last_frames_per_call_;
    const float* stage = (const float*)plot_stage_.p + (size_t)n.plot_slot[call] * 2 * fpc;
    hip_check(hipMemcpyAsync(left, stage, fpc * sizeof(float), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
    hip_check(hipMemcpyAsync(right, stage + fpc, fpc * sizeof(float), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
    sync();
    return 1;
}

// ---------------------------------------------------------------------------------------------
// video sub-graph: one Engine::run_tick pass over the video nodes in run order
// ---------------------------------------------------------------------------------------------
void Graph::run_video_tick(uint64_t t) {
    for (uint32_t id : video_order_) {
        Node& n = nodes_[id];
        switch (n.kind) {
        case MX_KIND_SOURCE_VIDEO: {
            // Output::from_line_type(Video) = None every tick (io.rs:76); the source fills it when a frame is due
            n.vout[0] = Node::VOut{};
            const uint64_t tick = t / spt_;
            while (!n.vsrc_sched.empty() && n.vsrc_sched.front().tick < tick) n.vsrc_sched.pop_front();   // a tick that was never run
            if (!n.vsrc_sched.empty()) {   // paced ingest: the tick a MediaSource / StreamInput emitted the frame on, or None
                if (n.vsrc_sched.front().tick == tick) {
                    Node::VSched& e = n.vsrc_sched.front();
                    n.vout[0].frame = e.frame; n.vout[0].dur = e.dur; n.vout[0].off = e.off;
                    n.vsrc_sched.pop_front();
                }
            } else if (!n.vsrc_ring.empty()) {   // a decoder's stream: the next frame of the ring, every tick
                n.vout[0].frame = n.vsrc_ring[n.vsrc_ring_pos]; n.vout[0].dur = n.vsrc_dur; n.vout[0].off = n.vsrc_off;
                n.vsrc_ring_pos = (n.vsrc_ring_pos + 1) % n.vsrc_ring.size();
            } else if (n.vsrc && (n.vsrc_repeat || n.vsrc_pending)) {
                n.vout[0].frame = n.vsrc; n.vout[0].dur = n.vsrc_dur; n.vout[0].off = n.vsrc_off;
                n.vsrc_pending = false;
            }
            if (n.vband && n.vout[0].frame) {   // row-band sharding: the frame is a halo slice of a smaller layer; deliver this rank's band of its scale
                FrameRef o;
                for (auto& f : n.vband_pool) if (f->rc.load(std::memory_order_acquire) == 1) { o = f; break; }   // only the pool holds it
                if (!o) {
                    if (n.vband_pool.size() >= 8) n.vband_pool.erase(n.vband_pool.begin());
                    n.vband_pool.push_back(FrameRef(DFrame::create(n.vband->full_w(), n.vband->band_rows(), stream_), false));
                    o = n.vband_pool.back();
                }
                n.vband->run(n.vout[0].frame.f, o.f, stream_);
                n.vout[0].frame = o;
            }
            break;
        }
        case MX_KIND_VIDEO_MIXER: {
            VideoInput in[4];
            for (int i = 0; i < 4; ++i) {
                const PortRef pr = n.in_src[i];
                if (pr.node < 0) continue;                                   // Disconnected => None (io.rs:56-57)
                const Node::VOut& v = nodes_[pr.node].vout[pr.port];
                if (!v.frame) continue;
                in[i].frame = v.frame.f; in[i].duration_hint = v.dur; in[i].tick_offset = v.off;
            }
            FrameRef o, a, b;
            n.vmixer->run_tick(t, in, o, a, b);
            n.vout[0] = Node::VOut{o, Rational::make(1, (int64_t)tps_), Rational::make(0, 1)};   // video_mixer.rs:241-247
            // A / B are clones of the input VideoFrames, duration and offset included (video_mixer.rs:80-90)
            mx_video_mixer_params p; std::memcpy(&p, n.params.data(), sizeof p);
            n.vout[1] = Node::VOut{}; n.vout[2] = Node::VOut{};
            if (a && p.a >= 0 && p.a < 4) n.vout[1] = Node::VOut{a, in[p.a].duration_hint, in[p.a].tick_offset};
            if (b && p.b >= 0 && p.b < 4) n.vout[2] = Node::VOut{b, in[p.b].duration_hint, in[p.b].tick_offset};
            break;
        }
        case MX_KIND_MONITOR: {
            // Monitor::run_tick (monitor.rs:113-139) + the codec thread's use of the Tick (monitor.rs:226-236)
            Node::MonTick mt;
            const Rational absolute = Rational::make((int64_t)t, (int64_t)sample_rate_);
            if (!n.mon_has_epoch) { n.mon_epoch = absolute; n.mon_has_epoch = true; }       // epoch.get_or_insert
            mt.ts = absolute - n.mon_epoch;                                                  // remove_epoch
            if (n.mon_depth && n.mon_queued >= n.mon_depth) {                                // try_send on a full channel: the tick is dropped (monitor.rs:163-177)
                mt.dropped = true;
                n.mon_ticks.push_back(std::move(mt));
                break;
            }
            if (n.mon_depth) ++n.mon_queued;
            const PortRef pr = n.in_src[0];
            const Node::VOut* vp = pr.node < 0 ? nullptr : &nodes_[pr.node].vout[pr.port];   // Disconnected => None (io.rs:56-57)
            if (vp && vp->frame) {
                mt.present = true;
                mt.frame_ts = mt.ts + vp->off;                                               // tick.timestamp + tick_offset
                mt.dur = vp->dur;
                mt.frame = n.mon_scaler->scale_keep(vp->frame);                              // VideoCtx::send_frame -> DynamicScaler::scale (encode.rs:287-295)
            }
            n.mon_ticks.push_back(std::move(mt));
            break;
        }
        case MX_KIND_VIDEO_TO_RGBA: {
            const PortRef pr = n.in_src[0];
            n.rgba_w = n.rgba_h = 0;
            const Node::VOut* vp = pr.node < 0 ? nullptr : &nodes_[pr.node].vout[pr.port];
            if (!vp || !vp->frame) {   // no picture this tick: nothing will follow the pending chains soon -- let them go (with whatever scales are queued)
                if (!n.rgba_pending.empty()) { flush_scales(stream_); launch_pending_rgba(n, n.rgba_pending.size(), false); }
                break;
            }
            DFrame* d = vp->frame.f;
            const int32_t stride = (int32_t)(((size_t)d->width * 4 + 15) & ~(size_t)15);
            const size_t need = (size_t)stride * d->height;
            const uint32_t K = video_batch_ticks();   // ticks whose chains share one launch inside a batched run
            if (n.rgba.size() != K || n.rgba[0].bytes < need) {
                if (!n.rgba_pending.empty()) { flush_scales(stream_); launch_pending_rgba(n, n.rgba_pending.size(), false); }
                sync();
                n.rgba.clear(); n.rgba.resize(K);
                for (DevBuf& b : n.rgba) b.alloc(need);
                n.rgba_cur = 0;
            }
            mx_video_to_rgba_params p; std::memcpy(&p, n.params.data(), sizeof p);
            n.rgba_cur = (n.rgba_cur + 1u) % K;
            uint8_t* const out = (uint8_t*)n.rgba[n.rgba_cur].p;
            if (d->lazy) {   // the composite only exists as a cross-fade chain: evaluate it straight into RGBA
                ChainRgbaArgs c{};
                fill_chain_rgba_sources(*d->lazy, c, stream_);   // layers that are unevaluated scaler outputs are resampled inside the kernel
                c.rgba = out; c.rgba_stride = (uint32_t)stride; c.width = d->width; c.height = d->height;
                c.use_matrix = p.use_matrix;
                for (int k = 0; k < 12; ++k) c.m[k] = p.matrix_q12[k];
                // Inside a batched run the sink runs LATE and K ticks at a time (K = video_batch_ticks(), default 16): the chains of ticks k .. k + K - 1
                // leave in ONE launch together with the scaler tiles ticks k + K .. k + 2K - 1 queued (mx_k_video.hip k_video_batch) -- the
                // chip then holds waves of several frames in every phase at once instead of marching through load / compute / store in step.
                // Every K-th sink call is an event: it takes ALL queued scales along (so the scales a chain needs left at its own event
                // or an earlier one) and, once 2K chains are pending, the K oldest -- always at least K calls old.  A Scaler writes 2K output
                // frames in turn and the sink K RGBA buffers, so nothing in one launch writes what something else in it reads or writes.
                // What is still pending when the run ends is launched then.
                n.rgba_pending.push_back(Node::PendingRgba{c, d->lazy});
                if ((++n.rgba_calls % K) == 0) {
                    if (n.rgba_pending.size() >= 2 * (size_t)K) launch_pending_rgba(n, K, true);
                    else flush_scales(stream_);
                }
                n.rgba_w = d->width; n.rgba_h = d->height; n.rgba_stride = stride;
                break;
            }
            if (!n.rgba_pending.empty()) { flush_scales(stream_); launch_pending_rgba(n, n.rgba_pending.size(), false); }
            d->ensure_pixels(stream_);
            if (d->fmt != MX_PIXFMT_YUV420P) throw Error(MX_ERR_INVALID, "VIDEO_TO_RGBA takes yuv420p (a VideoMixer output); put a VideoMixer in front of a source of another format");
            RgbaArgs a;
            a.y = d->data[0]; a.u = d->data[1]; a.v = d->data[2]; a.rgba = out;
            a.y_stride = d->stride[0]; a.u_stride = d->stride[1]; a.v_stride = d->stride[2]; a.rgba_stride = (uint32_t)stride;
            a.width = d->width; a.height = d->height; a.use_matrix = p.use_matrix;
            for (int k = 0; k < 12; ++k) a.m[k] = p.matrix_q12[k];
            launch_yuv420_to_rgba(a, stream_);
            n.rgba_w = d->width; n.rgba_h = d->height; n.rgba_stride = stride;
            break;
        }
        default: break;
        }
    }
}

void Graph::launch_pending_rgba(Node& n, size_t count, bool with_queued_scales) {
    count = std::min(count, n.rgba_pending.size());
    const size_t K = std::min<size_t>(video_batch_ticks(), MX_VB_MAX_CHAINS);   // the sink's K RGBA buffers are written in turn: at most K chains per launch
    std::vector<ChainRgbaArgs> c(K);
    size_t i = 0;
    while (i < count) {
        const int m = (int)std::min(K, count - i);
        for (int k = 0; k < m; ++k) c[k] = n.rgba_pending[i + k].args;
        if (with_queued_scales && i == 0) launch_chains_rgba_after_queued_scales(c.data(), m, stream_);
        else launch_video_batch(nullptr, 0, c.data(), m, stream_);
        i += m;
    }
    n.rgba_pending.erase(n.rgba_pending.begin(), n.rgba_pending.begin() + count);
}

void Graph::set_video_source(uint32_t node, DFrame* frame, Rational dur, Rational off, bool repeat) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_SOURCE_VIDEO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_VIDEO");
    Node& n = nodes_[node];
    n.vsrc = frame ? FrameRef(frame, true) : FrameRef();
    n.vsrc_dur = dur; n.vsrc_off = off; n.vsrc_repeat = repeat; n.vsrc_pending = frame != nullptr;
}

void Graph::set_video_source_band(uint32_t node, uint32_t in_w, uint32_t in_full_h, uint32_t src_row0, uint32_t slice_rows,
                                  uint32_t full_w, uint32_t full_h, uint32_t row0, uint32_t band_rows) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_SOURCE_VIDEO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_VIDEO");
    Node& nd = nodes_[node];
    sync();                                     // a previous band scaler's row buffer may be in use
    nd.vband.reset(); nd.vband_pool.clear();
    if (band_rows) nd.vband = std::make_shared<BandScaler>(in_w, in_full_h, src_row0, slice_rows, full_w, full_h, row0, band_rows);
}

void Graph::queue_video_source(uint32_t node, uint64_t tick, DFrame* frame, Rational dur, Rational off) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_SOURCE_VIDEO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_VIDEO");
    if (!frame) throw Error(MX_ERR_INVALID, "frame is NULL");
    Node& nd = nodes_[node];
    if (!nd.vsrc_sched.empty() && nd.vsrc_sched.back().tick >= tick) throw Error(MX_ERR_INVALID, "video source frames must be queued in tick order, one per tick");
    nd.vsrc_sched.push_back(Node::VSched{tick, FrameRef(frame, true), dur, off});
}

void Graph::check_video_queue(uint32_t node, uint64_t first_tick) const {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_SOURCE_VIDEO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_VIDEO");
    const Node& nd = nodes_[node];
    if (!nd.vsrc_sched.empty() && nd.vsrc_sched.back().tick >= first_tick) throw Error(MX_ERR_INVALID, "video source frames must be queued in tick order, one per tick");
}

void Graph::check_source_write(uint32_t node, size_t frames) const {
    if (node >= nodes_.size()) throw Error(MX_ERR_INVALID, "node out of range");
    const Node& n = nodes_[node];
    if (n.kind != MX_KIND_SOURCE_MONO && n.kind != MX_KIND_SOURCE_STEREO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_*");
    if (n.bound) throw Error(MX_ERR_INVALID, "source is bound to a caller device buffer");
    if (frames > cap_frames_) throw Error(MX_ERR_INVALID, "more ticks than max_ticks_per_run");
}

void Graph::monitor_consume(uint32_t node, uint32_t n_ticks) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_MONITOR) throw Error(MX_ERR_INVALID, "node is not a MONITOR");
    Node& n = nodes_[node];
    n.mon_queued -= std::min(n.mon_queued, n_ticks);
}

const Node::MonTick& Graph::monitor_tick(uint32_t node, uint32_t tick_in_run) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_MONITOR) throw Error(MX_ERR_INVALID, "node is not a MONITOR");
    if (tick_in_run >= nodes_[node].mon_ticks.size()) throw Error(MX_ERR_INVALID, "tick_in_run beyond the last run");
    flush_scales(stream_);
    sync();          // the caller reads the frame on streams of its own
    return nodes_[node].mon_ticks[tick_in_run];
}

Graph::MonitorLayout Graph::monitor_layout(uint32_t node) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_MONITOR) throw Error(MX_ERR_INVALID, "node is not a MONITOR");
    const Scaler& sc = *nodes_[node].mon_scaler;
    MonitorLayout l{};
    l.width = sc.out_w(); l.height = sc.out_h();
    size_t total = 0;                                   // DFrame's own layout for yuv420p (mx_video.cpp alloc_planes): what every kept frame has
    for (int p = 0; p < 3; ++p) {
        const uint32_t rb = p ? l.width >> 1 : l.width, rows = p ? l.height >> 1 : l.height;
        l.stride[p] = (rb + 63u) & ~63u;
        l.plane_offset[p] = total;
        total += ((size_t)l.stride[p] * rows + 255) & ~(size_t)255;
    }
    l.frame_bytes = total;
    return l;
}

void Graph::read_monitor_video(uint32_t node, uint32_t first_tick, uint32_t n_ticks, uint8_t* frames, uint8_t* present) {
    const MonitorLayout l = monitor_layout(node);
    Node& n = nodes_[node];
    if ((size_t)first_tick + n_ticks > n.mon_ticks.size()) throw Error(MX_ERR_INVALID, "ticks beyond the last run");
    if (n_ticks && (!frames || !present)) throw Error(MX_ERR_INVALID, "NULL argument");
    if (!n_ticks) return;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const size_t need = (size_t)n_ticks * l.frame_bytes;
    if (n.mon_pack.bytes < need) { sync(); n.mon_pack.alloc(need); }
    uint32_t any = 0;
    for (uint32_t k0 = 0; k0 < n_ticks; k0 += 224) {
        GatherArgs a{};
        a.n = std::min<uint32_t>(224u, n_ticks - k0);
        a.dst = reinterpret_cast<uint4*>((uint8_t*)n.mon_pack.p + (size_t)k0 * l.frame_bytes);
        a.q_per_frame = (uint32_t)(l.frame_bytes / 16);
        for (uint32_t k = 0; k < a.n; ++k) {
            const Node::MonTick& mt = n.mon_ticks[first_tick + k0 + k];
            present[k0 + k] = mt.present ? 1 : 0;
            a.src[k] = nullptr;
            if (!mt.present) continue;
            const DFrame* f = mt.frame.f;
            if (f->mem.bytes != l.frame_bytes || f->plane_offset(1) != l.plane_offset[1] || f->plane_offset(2) != l.plane_offset[2])
                throw Error(MX_ERR_INTERNAL, "a kept monitor frame does not have the monitor layout");
            a.src[k] = reinterpret_cast<const uint4*>(f->mem.p);
            ++any;
        }
        launch_gather_frames(a, stream_);
    }
    if (any) hip_check(hipMemcpyAsync(frames, n.mon_pack.p, need, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H monitor frames)");
    sync();
}

void Graph::read_monitor_audio_i16(uint32_t node, int16_t* host, uint32_t n_ticks) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_MONITOR) throw Error(MX_ERR_INVALID, "node is not a MONITOR");
    if (n_ticks > last_calls_) throw Error(MX_ERR_INVALID, "more ticks than the last run had");
    if (n_ticks && !host) throw Error(MX_ERR_INVALID, "audio is NULL");
    const PortRef pr = nodes_[node].in_src[1];
    const size_t per_tick = 2 * last_frames_per_call_;
    if (pr.node < 0) { std::memset(host, 0, (size_t)n_ticks * per_tick * sizeof(int16_t)); return; }   // InputRef::Disconnected: the zero buffer
    read_output_i16((uint32_t)pr.node, (uint32_t)pr.port, host, (size_t)n_ticks * last_frames_per_call_);
}

void Graph::set_video_source_ring(uint32_t node, DFrame* const* frames, size_t n, Rational dur, Rational off) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_SOURCE_VIDEO) throw Error(MX_ERR_INVALID, "node is not a SOURCE_VIDEO");
    Node& nd = nodes_[node];
    nd.vsrc_ring.clear(); nd.vsrc_ring_pos = 0;
    for (size_t i = 0; i < n; ++i) nd.vsrc_ring.push_back(FrameRef(frames[i], true));
    nd.vsrc_dur = dur; nd.vsrc_off = off;
    if (n) { nd.vsrc = FrameRef(); nd.vsrc_pending = false; nd.vsrc_repeat = false; }
}

FrameRef Graph::video_output(uint32_t node, uint32_t port) {
    if (node >= nodes_.size() || port >= nodes_[node].vout.size() || nodes_[node].out_type[port] != MX_VIDEO)
        throw Error(MX_ERR_INVALID, "not a video output terminal");
    FrameRef r = nodes_[node].vout[port].frame;
    if (r) {   // a frame crossing the ABI must have pixels -- and they must be there: the caller reads them on whatever stream it likes
        r->ensure_pixels(stream_);
        flush_scales(stream_);
        sync();
    }
    return r;
}

void Graph::rgba_output(uint32_t node, void** dev, int32_t* stride, uint32_t* w, uint32_t* h) {
    if (node >= nodes_.size() || nodes_[node].kind != MX_KIND_VIDEO_TO_RGBA) throw Error(MX_ERR_INVALID, "node is not a VIDEO_TO_RGBA");
    const Node& n = nodes_[node];
    if (dev) *dev = n.rgba_w ? n.rgba[n.rgba_cur].p : nullptr;
    if (stride) *stride = n.rgba_stride;
    if (w) *w = n.rgba_w;
    if (h) *h = n.rgba_h;
}

}  // namespace mx
