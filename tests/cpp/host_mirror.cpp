// C++ client of include/mixlab_gpu.hpp: the ModuleT / Engine mirror.  Exit 0 = ok, 2 = no GPU (said so on stderr).
// The check is the reference's own module test (src/module/eq_three.rs:150-167) in miniature: an impulse through EqThree
// at unity gains comes back delayed by the 3-sample history, and the Engine path gives the bits of the ModuleT path.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mixlab_gpu.hpp"

int main() {
    if (mx_device_count() <= 0) { std::fprintf(stderr, "no GPU: %s\n", mx_last_error()); return 2; }
    try {
        using namespace mixlab;
        const size_t SPT = 735;
        mx_eq_three_params p{0.0, 0.0, 0.0};
        auto eq = ModuleT<mx_eq_three_params>::create(MX_KIND_EQ_THREE, p, 44100, MX_FLAG_EQ_EXACT);
        std::vector<float> x(SPT, 0.f), y(SPT, -1.f);
        x[10] = 1.0f;
        std::vector<OutputRef> outs{OutputRef::Mono(y.data(), y.size())};
        eq.run_tick(0, {InputRef::Mono(x.data(), x.size())}, outs);
        // unity gains: lo + mid + hi = delayed input (eq_three.rs:76-85), up to f64 rounding
        if (std::fabs(y[13] - 1.0f) > 1e-6f || std::fabs(y[12]) > 1e-6f) { std::fprintf(stderr, "unexpected impulse response %g %g\n", y[12], y[13]); return 1; }
        eq.update(mx_eq_three_params{3.0, 0.0, -3.0});

        Workspace ws;
        const uint32_t src = ws.add(MX_KIND_SOURCE_MONO), e2 = ws.add(MX_KIND_EQ_THREE, p);
        ws.connect(src, 0, e2, 0);
        Engine eng(ws, 44100, 1, MX_FLAG_EQ_EXACT);
        std::vector<float> y2(SPT);
        eng.write_source(src, x.data());
        eng.run_tick(0);
        eng.read_output(e2, 0, y2.data());
        if (std::memcmp(y.data(), y2.data(), SPT * sizeof(float)) != 0) { std::fprintf(stderr, "Engine and ModuleT paths differ\n"); return 1; }

        // a port-type mismatch is an error, not a crash (io.rs:40-41 panics)
        bool threw = false;
        try { std::vector<OutputRef> bad{OutputRef::Stereo(y.data(), y.size())}; eq.run_tick(SPT, {InputRef::Mono(x.data(), x.size())}, bad); }
        catch (const Error& err) { threw = err.code == MX_ERR_TYPE || err.code == MX_ERR_INVALID; }
        if (!threw) { std::fprintf(stderr, "type mismatch was not reported\n"); return 1; }
        std::printf("host_mirror ok\n");
        return 0;
    } catch (const mixlab::Error& e) { std::fprintf(stderr, "mixlab error %d: %s\n", e.code, e.what()); return 1; }
}
