// mx_exchange.cpp -- bus exchange of the sharded audio job: RCCL over xGMI, or the in-process loopback group (mx_exchange.hpp).
//
// Pipelining (both transports): the partial buses are packed device-to-device into one of two slots on the graph's (compute)
// stream; the exchange's own stream waits for that, runs the collectives and the rank-ordered combine and records `done`;
// the compute stream only waits for `done` of the slot it is about to pack again.  Steady-state step = max(compute, exchange).
//
//   allgather   one all-gather of the whole [master | cue] partials, then Mixer(N, unity) over them: (N - 1) bus lengths received.
//   slices      the ORDERED form of reduce-scatter + all-gather: the step's time axis is cut into N slices of whole ticks, rank j
//               receives slice j of every partial (grouped send / recv), adds them in rank order (the same Mixer kernel) and an
//               all-gather distributes the finished slices: 2 (N - 1) / N bus lengths received.  Bit-identical to allgather.
//   allreduce   ncclAllReduce(sum), what the north-star names.  NOT the sum order of any graph the reference can express (the
//               ring's order differs per chunk): explicitly non-parity, RCCL only.
#include "mx_exchange.hpp"
#include "mx_kernels.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded on first use (below)

#include <cstring>
#include <mutex>
#include <string>

namespace mx {

// RCCL is bound LAZILY (dlopen on the first RCCL exchange / mx_exchange_unique_id): single-GPU audio and video use and the loopback
// transport never call it, so a host without librccl can still build and load libmixlab_gpu.so; asking for the RCCL transport there
// fails with MX_ERR_DEVICE and a message that names the library.  The collectives of the data path are still this library's own calls.
namespace {
struct Rccl {
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string error;   // why it is not available (empty: loaded)
};
const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        // MX_RCCL_LIB: the library to bind instead (tests: tests/helpers/fake_rccl.c, a test double that runs these collectives between processes sharing one GPU)
        const char* const over = getenv("MX_RCCL_LIB");
        if (over && *over) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        else for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { const char* e = dlerror(); r.error = std::string(over && *over ? over : "librccl.so") + std::string(" could not be loaded (") + (e ? e : "?") + "): the RCCL transport is not available on this host"; return; }
        auto sym = [&](auto& fp, const char* n) { fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, n)); if (!fp && r.error.empty()) r.error = std::string("librccl.so has no ") + n; };
        sym(r.GetErrorString, "ncclGetErrorString"); sym(r.GetUniqueId, "ncclGetUniqueId"); sym(r.CommInitRank, "ncclCommInitRank");
        sym(r.CommDestroy, "ncclCommDestroy"); sym(r.AllGather, "ncclAllGather"); sym(r.AllReduce, "ncclAllReduce");
        sym(r.Send, "ncclSend"); sym(r.Recv, "ncclRecv"); sym(r.GroupStart, "ncclGroupStart"); sym(r.GroupEnd, "ncclGroupEnd");
    });
    if (!r.error.empty()) throw Error(MX_ERR_DEVICE, r.error);
    return r;
}
}  // namespace

static void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(MX_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(r));
}

void exchange_unique_id(void* out128) {
    static_assert(sizeof(ncclUniqueId) == MX_EXCHANGE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof id);
}

static size_t pad64(size_t n) { return (n + 63) & ~(size_t)63; }

static hipEvent_t make_event(bool timing) {
    hipEvent_t e = nullptr;
    hip_check(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming), "hipEventCreate");
    return e;
}

Exchange::Exchange(Graph& g, uint32_t mix, uint32_t n_ticks, uint32_t rank, uint32_t world, const void* uid, LoopbackGroup* lb, uint32_t mode)
    : lb_(lb), rank_(rank), world_(world), T_(n_ticks) {
  try {
    if (world == 0 || rank >= world) throw Error(MX_ERR_INVALID, "rank must be below world");
    if (n_ticks == 0) throw Error(MX_ERR_INVALID, "n_ticks is 0");
    if (mode > MX_EXCHANGE_ALLREDUCE) throw Error(MX_ERR_INVALID, "unknown exchange mode");
    if (!lb && !uid) throw Error(MX_ERR_INVALID, "an RCCL exchange needs the job's ncclUniqueId (mx_exchange_unique_id on rank 0), a loopback one its group");
    if (lb && uid) throw Error(MX_ERR_INVALID, "give either an ncclUniqueId or a loopback group, not both");
    if (lb && lb->world != world) throw Error(MX_ERR_INVALID, "the loopback group was made for another world size");
    if (lb && lb->members[rank]) throw Error(MX_ERR_INVALID, "this rank of the loopback group is taken");
    graph_ = &g;   // (a graph whose Mixer bank runs on a second stream -- MX_FLAG_OVERLAP_TAIL, or automatically for short submissions -- is waited for before the buses are packed)
    if (mix >= g.n_nodes() || g.node(mix).kind != MX_KIND_MIXER) throw Error(MX_ERR_INVALID, "node is not a Mixer");
    device_ = g.device();
    tps_ = g.ticks_per_second();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    size_t fpt = 0, fpt_c = 0;
    m_ptr_ = g.output_ptr(mix, 0, &fpt, false);   // (this consumer orders itself after the Mixer bank wherever it runs: the graph keeps its automatic second-stream mode)
    c_ptr_ = g.output_ptr(mix, 1, &fpt_c, false);
    fpt_ = fpt;
    if ((size_t)n_ticks * g.spt() > g.cap_frames())
        throw Error(MX_ERR_INVALID, "n_ticks exceeds the graph's max_ticks_per_run");
    n_fl_ = fpt * n_ticks;
    n_flp_ = pad64(n_fl_);
    if (mode == MX_EXCHANGE_AUTO) mode = (world >= 4 && n_ticks % world == 0) ? MX_EXCHANGE_SLICES : MX_EXCHANGE_ALLGATHER;
    if (mode == MX_EXCHANGE_SLICES && n_ticks % world) throw Error(MX_ERR_INVALID, "the ticks of a step do not divide into one time slice per rank");
    if (mode == MX_EXCHANGE_ALLREDUCE && lb) throw Error(MX_ERR_INVALID, "allreduce is RCCL's own summation order: not available on the loopback transport");
    mode_ = mode;
    compute_ = g.stream();
    hip_check(hipStreamCreateWithFlags(&cs_, hipStreamNonBlocking), "hipStreamCreate");
    if (mode == MX_EXCHANGE_SLICES) { t_slice_ = n_ticks / world; L_ = fpt * t_slice_; Lp_ = pad64(L_); }
    for (Slot& sl : slots_) {
        sl.packed = make_event(false); sl.fin = make_event(false); sl.consumed = make_event(false);
        sl.begin = make_event(true); sl.done = make_event(true);
        sl.part.alloc(2 * n_flp_ * sizeof(float));
        hip_check(hipMemsetAsync(sl.part.p, 0, sl.part.bytes, cs_), "hipMemsetAsync");
        if (mode == MX_EXCHANGE_ALLGATHER) {
            sl.gathered.alloc((size_t)world * 2 * n_flp_ * sizeof(float));
            build_combine(sl, n_ticks, (const float*)sl.gathered.p, 2 * n_flp_, n_flp_);
        } else if (mode == MX_EXCHANGE_SLICES) {
            sl.recv.alloc((size_t)world * 2 * Lp_ * sizeof(float));
            sl.final_.alloc(2 * n_flp_ * sizeof(float));
            build_combine(sl, t_slice_, (const float*)sl.recv.p, 2 * Lp_, Lp_);
        }
    }
    hip_check(hipStreamSynchronize(cs_), "hipStreamSynchronize");
    if (!lb) {
        ncclUniqueId id;
        std::memcpy(&id, uid, sizeof id);
        ncclComm_t c = nullptr;
        nccl_check(rccl().CommInitRank(&c, (int)world, id, (int)rank), "ncclCommInitRank");
        comm_ = c;
    } else {
        lb->members[rank] = this;
        lb->arrived[rank] = lb->completed;
    }
  } catch (...) {      // a constructor that throws runs no destructor: the stream, the events and a half-made communicator are released here
    joined_ = false;   // (never entered in the loopback group's member list)
    destroy();
    throw;
  }
}

void Exchange::destroy() noexcept {
    (void)hipSetDevice(device_);
    if (held_) { try { ensure_submitted(); } catch (...) {} }
    if (graph_) { graph_->set_tail_hook(nullptr); for (Slot& sl : slots_) if (sl.done) graph_->forget_wait_before_next_run(sl.done); }
    if (cs_) (void)hipStreamSynchronize(cs_);
    if (lb_ && joined_ && rank_ < lb_->members.size() && lb_->members[rank_] == this) {
        // peers may still be reading this member's buffers
        for (Exchange* q : lb_->members) if (q && q != this && q->cs_) (void)hipStreamSynchronize(q->cs_);
        lb_->members[rank_] = nullptr;
    }
    if (comm_) { try { (void)rccl().CommDestroy(comm_); } catch (...) {} comm_ = nullptr; }
    for (Slot& sl : slots_) {
        sl.cg.reset();
        for (hipEvent_t* e : {&sl.packed, &sl.begin, &sl.fin, &sl.done, &sl.consumed}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    }
    if (cs_) { (void)hipStreamDestroy(cs_); cs_ = nullptr; }
}

Exchange::~Exchange() { destroy(); }

// Mixer(world, unity) for Master and for Cue over bound device buffers: partial r's master at base + r * peer_stride, its cue
// cue_off floats further.  Unity = 0 dB, fader 1.0: each term is (x as f64 * 1.0) as f32 = x, so an output sample is the f32 sum
// of the partials in channel (= rank) order, exactly what Mixer::run_tick does (src/module/mixer.rs:57-68).
void Exchange::build_combine(Slot& sl, uint32_t ticks, const float* base, size_t peer_stride, size_t cue_off) {
    const uint32_t W = world_;
    std::vector<mx_mixer_channel_params> unity(W);
    for (auto& c : unity) { std::memset(&c, 0, sizeof c); c.gain_db = 0.0; c.fader = 1.0; c.cue = 0; }
    std::vector<mx_node> nodes(2 * W + 2);
    for (uint32_t r = 0; r < 2 * W; ++r) nodes[r] = mx_node{MX_KIND_SOURCE_STEREO, 0, nullptr};
    sl.fm = 2 * W; sl.fc = 2 * W + 1;
    nodes[sl.fm] = mx_node{MX_KIND_MIXER, (uint32_t)(W * sizeof(mx_mixer_channel_params)), unity.data()};
    nodes[sl.fc] = nodes[sl.fm];
    std::vector<mx_edge> edges;
    for (uint32_t r = 0; r < W; ++r) {
        edges.push_back(mx_edge{r, 0, sl.fm, r});
        edges.push_back(mx_edge{W + r, 0, sl.fc, r});
    }
    mx_graph_opts o{};
    o.sample_rate = (uint32_t)((fpt_ / 2) * tps_);
    o.ticks_per_second = tps_;
    o.max_ticks_per_run = ticks;
    o.device = device_;
    o.stream = cs_;
    sl.cg = std::make_unique<Graph>(nodes.data(), nodes.size(), edges.data(), edges.size(), o);
    if (sl.cg->spt() * 2 != fpt_) throw Error(MX_ERR_INTERNAL, "combine graph: samples per tick differ from the bus");
    for (uint32_t r = 0; r < W; ++r) {
        sl.cg->bind_source(r, base + (size_t)r * peer_stride);
        sl.cg->bind_source(W + r, base + (size_t)r * peer_stride + cue_off);
    }
    size_t f = 0;
    sl.fm_out = sl.cg->output_ptr(sl.fm, 0, &f);
    sl.fc_out = sl.cg->output_ptr(sl.fc, 0, &f);
}

uint64_t Exchange::bytes_received_per_step() const {
    const uint64_t bus = 2ull * n_fl_ * sizeof(float), w = world_;
    if (mode_ == MX_EXCHANGE_ALLGATHER) return (w - 1) * bus;
    return 2 * (w - 1) * bus / w;   // slices, and a ring all-reduce
}

Exchange::Slot& Exchange::slot_of(uint64_t step, const char* what) {
    Slot& sl = slots_[step & 1];
    if (!sl.used || sl.step != (int64_t)step) throw Error(MX_ERR_INVALID, std::string(what) + ": that step is not in flight (only the last two submitted steps are kept)");
    if (!sl.queued) throw Error(MX_ERR_INVALID, std::string(what) + ": not every rank of the loopback group has submitted that step yet");
    return sl;
}

void Exchange::submit(uint64_t step) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (lb_) {   // everything that can be refused is refused BEFORE the slot or the group's bookkeeping is touched: a refused submit leaves the group usable
        if (lb_->arrived[rank_] != lb_->completed) throw Error(MX_ERR_INVALID, "loopback: this rank already submitted a step the other ranks have not submitted yet");
        for (Exchange* q : lb_->members) if (!q) throw Error(MX_ERR_INVALID, "loopback: not every rank of the group has been created");
        for (uint32_t r = 0; r < world_; ++r)
            if (r != rank_ && lb_->arrived[r] != lb_->completed && lb_->arrived[r] != (int64_t)step)
                throw Error(MX_ERR_INVALID, "loopback: the ranks of a group submit the same step numbers in the same order");
    }
    // RCCL transport over a graph that holds its Mixer bank back for the next run's EqThree launch (the second-stream mode): the pack and the exchange go out when the
    // graph releases that launch -- behind it, on its stream -- instead of joining the streams now (which would put the bank in front of the next run again).  Whoever asks
    // for this step's result first (wait / result / read_result / elapsed_ms / sync) releases it.  (The loopback transport keeps the join: its members' submits are counted
    // when they are made.)
    if (!lb_) {
        ensure_submitted();
        if (graph_->tail_held()) {
            held_ = true; held_step_ = step;
            graph_->set_tail_hook([this](hipStream_t ts) { if (held_) { held_ = false; do_submit(held_step_, ts); } });
            return;
        }
    }
    graph_->join_tail();   // the buses are packed on the graph's stream: a Mixer bank still running on the tail stream finishes first
    do_submit(step, compute_);
}

void Exchange::ensure_submitted() {
    if (held_) graph_->join_tail();      // releases the bank; the hook runs do_submit
    if (held_) { held_ = false; graph_->set_tail_hook(nullptr); do_submit(held_step_, compute_); }   // (the graph had nothing held after all)
}

void Exchange::do_submit(uint64_t step, hipStream_t ps) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    Slot& sl = slots_[step & 1];
    // the exchange that last used this slot has finished with the packed partials (loopback: every peer that read them, too)
    if (sl.used) {
        if (lb_) { for (Exchange* q : lb_->members) hip_check(hipStreamWaitEvent(ps, q->slots_[step & 1].done, 0), "hipStreamWaitEvent"); }
        else hip_check(hipStreamWaitEvent(ps, sl.done, 0), "hipStreamWaitEvent");
    }
    // all-reduce runs in place: its RESULT is `part`, which the pack below overwrites -- on the compute stream, so the consumer's
    // release (an event on ITS stream) has to order the pack, not only the next exchange (the other modes' results are written on cs_)
    if (mode_ == MX_EXCHANGE_ALLREDUCE && sl.consumed_pending) hip_check(hipStreamWaitEvent(ps, sl.consumed, 0), "hipStreamWaitEvent");
    float* part = (float*)sl.part.p;
    // (by a kernel: a copy-engine copy queued behind a cross-stream wait makes the host wait for that event -- k_upload, mx_k_stream.hip)
    if (c_ptr_ == m_ptr_ + n_fl_ && n_flp_ == n_fl_) {   // Master and Cue are neighbours in the graph's slab: one copy packs both
        launch_upload(part, m_ptr_, 2 * n_fl_ * sizeof(float), ps);
    } else {
        launch_upload(part, m_ptr_, n_fl_ * sizeof(float), ps);
        launch_upload(part + n_flp_, c_ptr_, n_fl_ * sizeof(float), ps);
    }
    hip_check(hipEventRecord(sl.packed, ps), "hipEventRecord");
    sl.step = (int64_t)step;
    sl.queued = false;
    if (!lb_) {
        hip_check(hipStreamWaitEvent(cs_, sl.packed, 0), "hipStreamWaitEvent");
        if (sl.consumed_pending) { hip_check(hipStreamWaitEvent(cs_, sl.consumed, 0), "hipStreamWaitEvent"); sl.consumed_pending = false; }
        hip_check(hipEventRecord(sl.begin, cs_), "hipEventRecord");
        collective_rccl(sl);
        hip_check(hipEventRecord(sl.done, cs_), "hipEventRecord");
        sl.used = true; sl.queued = true;
        if (ps != compute_) graph_->wait_before_next_run(sl.done);   // packed behind a released Mixer bank: see Graph::wait_before_next_run
        return;
    }
    lb_->arrived[rank_] = (int64_t)step;
    bool all = true;
    for (int64_t a : lb_->arrived) all = all && a == (int64_t)step;
    for (Exchange* q : lb_->members) all = all && q->slots_[step & 1].step == (int64_t)step;
    if (all) loopback_round(*lb_, step);
}

void Exchange::collective_rccl(Slot& sl) {
    ncclComm_t comm = comm_;
    const Rccl& R = rccl();
    float* part = (float*)sl.part.p;
    const int W = (int)world_;
    if (mode_ == MX_EXCHANGE_ALLGATHER) {
        nccl_check(R.AllGather(part, sl.gathered.p, 2 * n_flp_, ncclFloat, comm, cs_), "ncclAllGather");   // ONE all-gather per step
        sl.cg->run(0, fpt_ / 2, T_);                                                                          // rank-ordered f32 sum
    } else if (mode_ == MX_EXCHANGE_SLICES) {
        float* recv = (float*)sl.recv.p;
        nccl_check(R.GroupStart(), "ncclGroupStart");                  // slice j of every rank's partial buses -> rank j
        for (int q = 0; q < W; ++q) {
            nccl_check(R.Send(part + (size_t)q * L_, L_, ncclFloat, q, comm, cs_), "ncclSend");
            nccl_check(R.Send(part + n_flp_ + (size_t)q * L_, L_, ncclFloat, q, comm, cs_), "ncclSend");
            nccl_check(R.Recv(recv + (size_t)q * 2 * Lp_, L_, ncclFloat, q, comm, cs_), "ncclRecv");
            nccl_check(R.Recv(recv + (size_t)q * 2 * Lp_ + Lp_, L_, ncclFloat, q, comm, cs_), "ncclRecv");
        }
        nccl_check(R.GroupEnd(), "ncclGroupEnd");
        sl.cg->run(0, fpt_ / 2, t_slice_);                              // rank-ordered f32 sum of my slice
        float* fin = (float*)sl.final_.p;
        nccl_check(R.GroupStart(), "ncclGroupStart");                  // every rank ends with the whole Master and Cue
        nccl_check(R.AllGather(sl.fm_out, fin, L_, ncclFloat, comm, cs_), "ncclAllGather");
        nccl_check(R.AllGather(sl.fc_out, fin + n_flp_, L_, ncclFloat, comm, cs_), "ncclAllGather");
        nccl_check(R.GroupEnd(), "ncclGroupEnd");
    } else {
        nccl_check(R.AllReduce(part, part, 2 * n_flp_, ncclFloat, ncclSum, comm, cs_), "ncclAllReduce");   // non-parity: the ring decides the order
    }
}

// Loopback transport: every member has packed step `step`; queue, for every member, what the collectives would have
// delivered (plain device-to-device copies on the member's exchange stream), the same combine, the same events.
void Exchange::loopback_round(LoopbackGroup& grp, uint64_t step) {
    const uint32_t W = grp.world;
    const int s = (int)(step & 1);
    auto cp = [](void* dst, const void* src, size_t floats, hipStream_t st) {
        hip_check(hipMemcpyAsync(dst, src, floats * sizeof(float), hipMemcpyDefault, st), "hipMemcpyAsync(loopback)");
    };
    // before any event of this slot is recorded again: a member's combine outputs and receive buffers of step - 2 may still be
    // read by its peers' copies -- its stream waits for every member's `done` of that step
    for (Exchange* x : grp.members) {
        hip_check(hipSetDevice(x->device_), "hipSetDevice");
        Slot& sl = x->slots_[s];
        if (sl.used) for (Exchange* q : grp.members) hip_check(hipStreamWaitEvent(x->cs_, q->slots_[s].done, 0), "hipStreamWaitEvent");
        if (sl.consumed_pending) { hip_check(hipStreamWaitEvent(x->cs_, sl.consumed, 0), "hipStreamWaitEvent"); sl.consumed_pending = false; }
    }
    for (Exchange* x : grp.members) {
        hip_check(hipSetDevice(x->device_), "hipSetDevice");
        Slot& sl = x->slots_[s];
        for (Exchange* q : grp.members) hip_check(hipStreamWaitEvent(x->cs_, q->slots_[s].packed, 0), "hipStreamWaitEvent");
        hip_check(hipEventRecord(sl.begin, x->cs_), "hipEventRecord");
        if (x->mode_ == MX_EXCHANGE_ALLGATHER) {
            for (uint32_t q = 0; q < W; ++q)
                cp((float*)sl.gathered.p + (size_t)q * 2 * x->n_flp_, grp.members[q]->slots_[s].part.p, 2 * x->n_flp_, x->cs_);
            sl.cg->run(0, x->fpt_ / 2, x->T_);
            hip_check(hipEventRecord(sl.done, x->cs_), "hipEventRecord");
        } else {
            float* recv = (float*)sl.recv.p;
            for (uint32_t q = 0; q < W; ++q) {
                const float* qp = (const float*)grp.members[q]->slots_[s].part.p;
                cp(recv + (size_t)q * 2 * x->Lp_, qp + (size_t)x->rank_ * x->L_, x->L_, x->cs_);
                cp(recv + (size_t)q * 2 * x->Lp_ + x->Lp_, qp + x->n_flp_ + (size_t)x->rank_ * x->L_, x->L_, x->cs_);
            }
            sl.cg->run(0, x->fpt_ / 2, x->t_slice_);
            hip_check(hipEventRecord(sl.fin, x->cs_), "hipEventRecord");
        }
    }
    for (Exchange* x : grp.members) {
        if (x->mode_ != MX_EXCHANGE_SLICES) continue;
        hip_check(hipSetDevice(x->device_), "hipSetDevice");
        Slot& sl = x->slots_[s];
        float* fin = (float*)sl.final_.p;
        for (uint32_t q = 0; q < W; ++q) {
            Slot& qs = grp.members[q]->slots_[s];
            hip_check(hipStreamWaitEvent(x->cs_, qs.fin, 0), "hipStreamWaitEvent");
            cp(fin + (size_t)q * x->L_, qs.fm_out, x->L_, x->cs_);
            cp(fin + x->n_flp_ + (size_t)q * x->L_, qs.fc_out, x->L_, x->cs_);
        }
        hip_check(hipEventRecord(sl.done, x->cs_), "hipEventRecord");
    }
    for (Exchange* x : grp.members) { x->slots_[s].used = true; x->slots_[s].queued = true; }
    grp.completed = (int64_t)step;
}

void Exchange::wait(uint64_t step, hipStream_t consumer) {
    ensure_submitted();
    Slot& sl = slot_of(step, "mx_exchange_wait");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamWaitEvent(consumer ? consumer : compute_, sl.done, 0), "hipStreamWaitEvent");
}

void Exchange::release(uint64_t step, hipStream_t consumer) {
    ensure_submitted();
    Slot& sl = slot_of(step, "mx_exchange_release");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipEventRecord(sl.consumed, consumer ? consumer : compute_), "hipEventRecord");
    sl.consumed_pending = true;
}

void Exchange::result(uint64_t step, float** master, float** cue, size_t* floats) {
    ensure_submitted();
    Slot& sl = slot_of(step, "mx_exchange_result");
    float *m, *c;
    if (mode_ == MX_EXCHANGE_ALLGATHER) { m = sl.fm_out; c = sl.fc_out; }
    else if (mode_ == MX_EXCHANGE_SLICES) { m = (float*)sl.final_.p; c = m + n_flp_; }
    else { m = (float*)sl.part.p; c = m + n_flp_; }
    if (master) *master = m;
    if (cue) *cue = c;
    if (floats) *floats = n_fl_;
}

void Exchange::read_result(uint64_t step, float* master, float* cue) {
    ensure_submitted();
    Slot& sl = slot_of(step, "mx_exchange_read_result");
    float *m, *c; size_t n;
    result(step, &m, &c, &n);
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipEventSynchronize(sl.done), "hipEventSynchronize");
    if (master) hip_check(hipMemcpy(master, m, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    if (cue) hip_check(hipMemcpy(cue, c, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
}

float Exchange::elapsed_ms(uint64_t step) {
    ensure_submitted();
    Slot& sl = slot_of(step, "mx_exchange_elapsed_ms");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipEventSynchronize(sl.done), "hipEventSynchronize");
    float ms = 0.f;
    hip_check(hipEventElapsedTime(&ms, sl.begin, sl.done), "hipEventElapsedTime");
    return ms;
}

void Exchange::sync() {
    ensure_submitted();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(cs_), "hipStreamSynchronize");
}

}  // namespace mx
