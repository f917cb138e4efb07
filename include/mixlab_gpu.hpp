// mixlab_gpu.hpp -- C++17 host-side mirror of the reference's module interface, header-only, over the C ABI
// (include/mixlab_gpu.h).  Names and argument meaning follow the Rust trait the scheduler calls:
//
//   trait ModuleT { fn create(params, ..) ; fn update(&mut self, params) ; fn run_tick(&mut self, t, inputs, outputs) ;
//                   fn inputs(&self) -> &[Terminal] ; fn outputs(&self) -> &[Terminal] }        src/module/mod.rs:7-19
//   enum InputRef  { Disconnected, Mono(&[f32]), Stereo(&[f32]), Video(..) }                     src/engine/io.rs:19-24
//   enum OutputRef { Mono(&mut [f32]), Stereo(&mut [f32]), Video(..) }                           src/engine/io.rs:96-100
//   Engine::run_tick over a frozen Workspace (modules + connections)                             src/engine.rs:400-510
//
// Errors: the C ABI never unwinds and returns MX_ERR_*; here they become mixlab::Error (the Rust side would re-raise the
// same way, codec/src/ffmpeg/ioctx.rs:51-67).  Port-type mismatch -- a panic! in the reference (io.rs:40-41) -- is MX_ERR_TYPE.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mixlab_gpu.h"

namespace mixlab {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) { if (rc != MX_OK) throw Error(rc, mx_last_error() ? mx_last_error() : "mixlab_gpu error"); }

// io.rs:19-24 -- a borrowed input buffer, or Disconnected (the engine then reads its static zero buffer, io.rs:8-9)
struct InputRef {
    int kind = MX_DISCONNECTED; const float* samples = nullptr; size_t len = 0;
    static InputRef Disconnected() { return {}; }
    static InputRef Mono(const float* p, size_t n) { return {MX_MONO, p, n}; }
    static InputRef Stereo(const float* p, size_t n) { return {MX_STEREO, p, n}; }
};
// io.rs:96-100
struct OutputRef {
    int kind = MX_MONO; float* samples = nullptr; size_t len = 0;
    static OutputRef Mono(float* p, size_t n) { return {MX_MONO, p, n}; }
    static OutputRef Stereo(float* p, size_t n) { return {MX_STEREO, p, n}; }
};

// One module instance behind the ModuleT surface (host buffers in, host buffers out: the per-module compat path).
template <class Params>
class ModuleT {
public:
    // ModuleT::create(params, ..)
    static ModuleT create(uint32_t kind, const Params& params, uint32_t sample_rate = 44100, uint32_t flags = 0) {
        return create_raw(kind, &params, sizeof(Params), sample_rate, flags);
    }
    // MixerParams and the build-specified FIR / resampler blobs are variable-length
    static ModuleT create_raw(uint32_t kind, const void* params, size_t len, uint32_t sample_rate = 44100, uint32_t flags = 0) {
        mx_graph_opts o{}; o.sample_rate = sample_rate; o.ticks_per_second = 60; o.max_ticks_per_run = 1; o.flags = flags; o.device = -1;
        ModuleT m; check(mx_module_create_ex(kind, params, len, &o, &m.h_)); return m;
    }
    ModuleT(ModuleT&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    ModuleT& operator=(ModuleT&& o) noexcept { if (this != &o) { reset(); h_ = std::exchange(o.h_, nullptr); } return *this; }
    ModuleT(const ModuleT&) = delete; ModuleT& operator=(const ModuleT&) = delete;
    ~ModuleT() { reset(); }

    // ModuleT::update(&mut self, new_params)
    void update(const Params& p) { check(mx_module_update(h_, &p, sizeof(Params))); }
    // ModuleT::run_tick(&mut self, t, inputs, outputs); t is the absolute sample clock tick * SPT (engine.rs:490)
    void run_tick(uint64_t t, const std::vector<InputRef>& inputs, std::vector<OutputRef>& outputs) {
        std::vector<mx_input> in(inputs.size()); std::vector<mx_output> out(outputs.size());
        for (size_t i = 0; i < inputs.size(); ++i) { in[i] = mx_input{}; in[i].kind = (mx_line)inputs[i].kind; in[i].samples = inputs[i].samples; in[i].len = inputs[i].len; }
        for (size_t i = 0; i < outputs.size(); ++i) { out[i] = mx_output{}; out[i].kind = (mx_line)outputs[i].kind; out[i].samples = outputs[i].samples; out[i].len = outputs[i].len; }
        check(mx_module_run_tick(h_, t, in.data(), in.size(), out.data(), out.size(), nullptr, nullptr));
    }
private:
    ModuleT() = default;
    void reset() { if (h_) { mx_module_destroy(h_); h_ = nullptr; } }
    mx_module* h_ = nullptr;
};

// The frozen part of Workspace (src/engine/workspace.rs:13-19) + the inner loop of Engine::run_tick on the device.
class Workspace {
public:
    uint32_t add(uint32_t kind, const void* params = nullptr, size_t len = 0) {
        blobs_.emplace_back((const uint8_t*)params, (const uint8_t*)params + len);
        kinds_.push_back(kind);
        return (uint32_t)kinds_.size() - 1;
    }
    template <class P> uint32_t add(uint32_t kind, const P& p) { return add(kind, &p, sizeof(P)); }
    // Workspace::connect(InputId, OutputId): a later connect to the same input replaces the earlier one (workspace.rs:110)
    void connect(uint32_t src, uint32_t src_port, uint32_t dst, uint32_t dst_port) {
        for (auto& e : edges_) if (e.dst_node == dst && e.dst_port == dst_port) { e.src_node = src; e.src_port = src_port; return; }
        edges_.push_back(mx_edge{src, src_port, dst, dst_port});
    }
    friend class Engine;
private:
    std::vector<uint32_t> kinds_; std::vector<std::vector<uint8_t>> blobs_; std::vector<mx_edge> edges_;
};

class Engine {
public:
    explicit Engine(const Workspace& ws, uint32_t sample_rate = 44100, uint32_t max_ticks_per_run = 1, uint32_t flags = 0) {
        std::vector<mx_node> nodes(ws.kinds_.size());
        for (size_t i = 0; i < nodes.size(); ++i) nodes[i] = mx_node{ws.kinds_[i], (uint32_t)ws.blobs_[i].size(), ws.blobs_[i].empty() ? nullptr : ws.blobs_[i].data()};
        mx_graph_opts o{}; o.sample_rate = sample_rate; o.ticks_per_second = 60; o.max_ticks_per_run = max_ticks_per_run; o.flags = flags; o.device = -1;
        check(mx_graph_build(nodes.data(), nodes.size(), ws.edges_.data(), ws.edges_.size(), &o, &g_));
    }
    Engine(const Engine&) = delete; Engine& operator=(const Engine&) = delete;
    ~Engine() { if (g_) mx_graph_destroy(g_); }
    // n consecutive Engine::run_tick calls in one submission (engine.rs:400-510); asynchronous
    void run_tick(uint64_t tick, uint32_t n = 1) { check(mx_graph_run_ticks(g_, tick, n)); }
    void write_source(uint32_t node, const float* samples, size_t n_ticks = 1) { check(mx_graph_write_source(g_, node, samples, n_ticks)); }
    void read_output(uint32_t node, uint32_t port, float* samples, size_t n_ticks = 1) { check(mx_graph_read_output(g_, node, port, samples, n_ticks)); }
    template <class P> void update(uint32_t node, const P& p) { check(mx_graph_update_params(g_, node, &p, sizeof(P))); }   // ModuleT::update through the engine (engine.rs:312)
    size_t samples_per_tick() const { size_t s = 0; check(mx_graph_samples_per_tick(g_, &s)); return s; }
    mx_graph* handle() const { return g_; }
private:
    mx_graph* g_ = nullptr;
};

}  // namespace mixlab
