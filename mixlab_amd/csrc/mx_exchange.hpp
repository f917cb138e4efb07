// mx_exchange.hpp -- the one exchange step of the sharded audio job, behind the C ABI (internal).
//
// N ranks each hold the partial Master / Cue buses of their strip shard (the Mixer(strips / N) of their graph); every rank
// ends with the whole bus.  The sharded graph is DEFINED as the reference-expressible hierarchy
//     N x Mixer(strips / N)  ->  Mixer(N, unity gains)
// so every output sample is the f32 sum of the N partials in rank order 0 .. N-1 (src/module/mixer.rs:57-68 applied to the
// partial buses), computed with the ordinary Mixer kernel on a small combine graph.
//
// Two transports: RCCL over xGMI (one process per GPU, a communicator made from an ncclUniqueId the host distributes), and
// an in-process LOOPBACK group (W exchanges of one process, e.g. W virtual ranks on one GPU: device-to-device copies in
// place of the collectives, same buffers, same combine, same pipelining) -- what the single-GPU tests of configs[4] run.
#pragma once
#include <memory>
#include <vector>

#include "mx_engine.hpp"

struct ncclComm;

namespace mx {

class Exchange;

struct LoopbackGroup {
    explicit LoopbackGroup(uint32_t w) : world(w), members(w, nullptr), arrived(w, -1) {}
    uint32_t world;
    std::vector<Exchange*> members;   // by rank
    std::vector<int64_t> arrived;     // the step each member submitted last (-1: none)
    int64_t completed = -1;           // the last step whose copies and combines were queued for every member
};

class Exchange {
public:
    Exchange(Graph& g, uint32_t mix, uint32_t n_ticks, uint32_t rank, uint32_t world, const void* nccl_unique_id,
             LoopbackGroup* loopback, uint32_t mode);
    ~Exchange();
    Exchange(const Exchange&) = delete;
    Exchange& operator=(const Exchange&) = delete;

    void submit(uint64_t step);
    void wait(uint64_t step, hipStream_t consumer);        // consumer stream waits for the step's combined bus
    void release(uint64_t step, hipStream_t consumer);     // the consumer's reads of that bus end here (slot reuse waits for it)
    void result(uint64_t step, float** master, float** cue, size_t* floats_per_bus);
    void read_result(uint64_t step, float* master, float* cue);   // synchronous D2H
    float elapsed_ms(uint64_t step);                        // device time of the exchange on its own stream; synchronises it
    void sync();

    uint32_t mode() const { return mode_; }
    uint32_t world() const { return world_; }
    uint32_t rank() const { return rank_; }
    size_t floats_per_bus() const { return n_fl_; }
    bool is_loopback() const { return lb_ != nullptr; }
    uint64_t bytes_received_per_step() const;

private:
    struct Slot {
        hipEvent_t packed = nullptr, begin = nullptr, fin = nullptr, done = nullptr, consumed = nullptr;
        bool used = false, consumed_pending = false, queued = false;
        int64_t step = -1;
        DevBuf part;       // [master | cue] of this rank, packed on the compute stream (allreduce: reduced in place)
        DevBuf gathered;   // allgather: [rank][master | cue]
        DevBuf recv;       // slices: [peer][master slice | cue slice], each padded to a cache line
        DevBuf final_;     // slices: [master | cue] of the whole step, finished slices in rank order
        std::unique_ptr<Graph> cg;   // Mixer(world, unity) x 2 on the exchange stream: the rank-ordered sum
        uint32_t fm = 0, fc = 0;     // its two Mixer nodes
        float *fm_out = nullptr, *fc_out = nullptr;
    };
    Slot& slot_of(uint64_t step, const char* what);
    void build_combine(Slot& sl, uint32_t ticks, const float* base, size_t peer_stride, size_t cue_off);
    void collective_rccl(Slot& sl);
    static void loopback_round(LoopbackGroup& grp, uint64_t step);
    void do_submit(uint64_t step, hipStream_t pack_stream);   // pack the buses on pack_stream (the graph's stream, or its tail stream behind a released Mixer bank) + the exchange behind them
    void ensure_submitted();           // a submit that waits for the graph to release its held-back Mixer bank: release it now
    bool held_ = false; uint64_t held_step_ = 0;
    void destroy() noexcept;           // what the destructor does; also the constructor's exit by exception
    bool joined_ = true;

    LoopbackGroup* lb_ = nullptr;
    ncclComm* comm_ = nullptr;
    int device_ = 0;
    uint32_t rank_ = 0, world_ = 1, mode_ = 0, T_ = 0, t_slice_ = 0, tps_ = 60;
    size_t fpt_ = 0;        // floats per tick of one bus
    size_t n_fl_ = 0;       // floats per bus per step
    size_t n_flp_ = 0;      // ... padded to 64 floats: where the cue part starts inside a [master | cue] buffer
    size_t L_ = 0, Lp_ = 0; // slices: floats per time slice (and padded)
    float *m_ptr_ = nullptr, *c_ptr_ = nullptr;
    hipStream_t compute_ = nullptr, cs_ = nullptr;
    Graph* graph_ = nullptr;   // the graph whose buses this exchange packs (outlives the exchange: it owns the buffers)
    Slot slots_[2];
};

void exchange_unique_id(void* out128);

}  // namespace mx
