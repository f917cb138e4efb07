// mx_abi_exchange.cpp -- extern "C" entry points of the multi-GPU bus exchange (include/mixlab_gpu.h, "multi-GPU").
// Same fencing convention as mx_abi.cpp: catch everything, stash the message, return a status.
#include <memory>
#include <string>

#include "mx_exchange.hpp"

using mx::Error;

struct mx_graph { std::unique_ptr<mx::Graph> g; };   // same layout as in mx_abi.cpp
struct mx_loopback_group { mx::LoopbackGroup grp; explicit mx_loopback_group(uint32_t w) : grp(w) {} };
struct mx_exchange { std::unique_ptr<mx::Exchange> x; };

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

extern "C" {

int mx_exchange_unique_id(void* id_out) {
    return guard([&] { REQUIRE(id_out, "id_out is NULL"); mx::exchange_unique_id(id_out); });
}

int mx_loopback_group_create(uint32_t world, mx_loopback_group** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        REQUIRE(world >= 1 && world <= 4096, "world out of range");
        *out = new mx_loopback_group(world);
    });
}

void mx_loopback_group_destroy(mx_loopback_group* grp) {
    (void)guard([&] { delete grp; });
}

int mx_exchange_create(mx_graph* g, uint32_t mixer_node, uint32_t n_ticks, uint32_t rank, uint32_t world,
                       const void* nccl_unique_id, mx_loopback_group* loopback, uint32_t mode, mx_exchange** out) {
    return guard([&] {
        REQUIRE(out, "out is NULL");
        *out = nullptr;
        REQUIRE(g, "graph is NULL");
        auto h = std::make_unique<mx_exchange>();
        h->x = std::make_unique<mx::Exchange>(*g->g, mixer_node, n_ticks, rank, world, nccl_unique_id, loopback ? &loopback->grp : nullptr, mode);
        *out = h.release();
    });
}

void mx_exchange_destroy(mx_exchange* x) {
    (void)guard([&] { delete x; });
}

int mx_exchange_submit(mx_exchange* x, uint64_t step) {
    return guard([&] { REQUIRE(x, "exchange is NULL"); x->x->submit(step); });
}

int mx_exchange_wait(mx_exchange* x, uint64_t step, void* stream) {
    return guard([&] { REQUIRE(x, "exchange is NULL"); x->x->wait(step, (hipStream_t)stream); });
}

int mx_exchange_result(mx_exchange* x, uint64_t step, void** master_device, void** cue_device, size_t* floats_per_bus) {
    return guard([&] {
        REQUIRE(x, "exchange is NULL");
        float *m = nullptr, *c = nullptr; size_t n = 0;
        x->x->result(step, &m, &c, &n);
        if (master_device) *master_device = m;
        if (cue_device) *cue_device = c;
        if (floats_per_bus) *floats_per_bus = n;
    });
}

int mx_exchange_release(mx_exchange* x, uint64_t step, void* stream) {
    return guard([&] { REQUIRE(x, "exchange is NULL"); x->x->release(step, (hipStream_t)stream); });
}

int mx_exchange_read_result(mx_exchange* x, uint64_t step, float* master, float* cue) {
    return guard([&] { REQUIRE(x, "exchange is NULL"); x->x->read_result(step, master, cue); });
}

int mx_exchange_elapsed_ms(mx_exchange* x, uint64_t step, float* ms) {
    return guard([&] { REQUIRE(x && ms, "NULL argument"); *ms = x->x->elapsed_ms(step); });
}

int mx_exchange_sync(mx_exchange* x) {
    return guard([&] { REQUIRE(x, "exchange is NULL"); x->x->sync(); });
}

int mx_exchange_get_info(const mx_exchange* x, mx_exchange_info* out) {
    return guard([&] {
        REQUIRE(x && out, "NULL argument");
        out->mode = x->x->mode(); out->rank = x->x->rank(); out->world = x->x->world(); out->loopback = x->x->is_loopback() ? 1u : 0u;
        out->floats_per_bus = x->x->floats_per_bus(); out->bytes_received_per_step = x->x->bytes_received_per_step();
    });
}

}  // extern "C"
