/* fake_rccl.c -- a TEST DOUBLE of the ten RCCL entry points mx_exchange.cpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce,
 * ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), between PROCESSES that share one GPU -- so that the product's own collective_rccl code (the
 * grouped send / recv offsets, the all-gathers, the strides q * 2 * Lp_ and n_flp_) runs with W = 2 and W = 4 on a one-GPU box.  libmixlab_gpu.so loads it instead of
 * librccl.so when MX_RCCL_LIB names it (tests only).  Test infrastructure: never shipped, never on a product path.
 *
 * Transport: a POSIX shared-memory segment named by the ncclUniqueId.  Every call (or group of calls) runs synchronously: wait for the stream, copy what this rank sends
 * device-to-host into its staging area and list it in its mailbox, barrier, copy what it receives host-to-device out of the senders' areas (the k-th receive from rank q
 * matches q's k-th send to this rank, as in NCCL), barrier.  Host staging instead of hipIpc handles: nothing here depends on the driver's IPC mode, and what is under
 * test is the caller's pointer arithmetic, not the wire.  Only ncclFloat.  Barriers time out (120 s) instead of hanging a GPU box.
 *
 *   gcc -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o libfake_rccl.so fake_rccl.c -L/opt/rocm/lib -lamdhip64 -lrt
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#define MAX_RANKS 16
#define MAX_OPS 256
#define MAGIC 0x52434c46u

typedef struct { int32_t peer; uint32_t _pad; uint64_t off, count; } desc_t;   /* peer: destination rank, -1 all-gather, -2 all-reduce */
typedef struct { uint32_t n, _pad; desc_t d[MAX_OPS]; } box_t;
typedef struct {
    volatile uint32_t ready, world;
    volatile uint32_t bar_count, bar_sense;
    uint64_t cap;
    box_t box[MAX_RANKS];
} hdr_t;

struct ncclComm { hdr_t* h; uint8_t* data; int rank, world; size_t map_bytes; uint32_t sense; };

enum { OP_SEND, OP_RECV, OP_ALLGATHER, OP_ALLREDUCE };
typedef struct { int kind, peer; const void* send; void* recv; size_t count; ncclComm_t comm; hipStream_t stream; } op_t;
static __thread int g_depth = 0;
static __thread op_t g_ops[MAX_OPS];
static __thread int g_nops = 0;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static int barrier(ncclComm_t c) {   /* sense-reversing, across processes, bounded */
    hdr_t* h = c->h;
    c->sense ^= 1u;
    if (__atomic_add_fetch(&h->bar_count, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
        __atomic_store_n(&h->bar_count, 0u, __ATOMIC_RELEASE);
        __atomic_store_n(&h->bar_sense, c->sense, __ATOMIC_RELEASE);
        return 0;
    }
    const double t0 = now_s();
    while (__atomic_load_n(&h->bar_sense, __ATOMIC_ACQUIRE) != c->sense) {
        if (now_s() - t0 > 120.0) { fprintf(stderr, "fake_rccl: rank %d waited 120 s at a barrier\n", c->rank); return -1; }
        usleep(50);
    }
    return 0;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "fake_rccl: invalid argument" : "fake_rccl: error"); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static unsigned counter = 0;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%u_%lx", (int)getpid(), ++counter, (unsigned long)(now_s() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    const char* e = getenv("FAKE_RCCL_CAP_MB");
    const size_t cap = (size_t)(e ? atoi(e) : 32) << 20;
    const size_t hdr = (sizeof(hdr_t) + 4095) & ~(size_t)4095, bytes = hdr + (size_t)world * cap;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { perror("fake_rccl: shm_open / ftruncate"); return ncclSystemError; }
    } else {
        const double t0 = now_s();
        while ((fd = shm_open(id.internal, O_RDWR, 0600)) < 0) { if (now_s() - t0 > 120.0) return ncclSystemError; usleep(200); }
        for (;;) { off_t sz = lseek(fd, 0, SEEK_END); if (sz >= (off_t)bytes) break; if (now_s() - t0 > 120.0) return ncclSystemError; usleep(200); }
    }
    void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
    c->h = (hdr_t*)p; c->data = (uint8_t*)p + hdr; c->rank = rank; c->world = world; c->map_bytes = bytes; c->sense = 0;
    if (rank == 0) { c->h->world = (uint32_t)world; c->h->cap = cap; c->h->bar_count = 0; c->h->bar_sense = 0; __atomic_store_n(&c->h->ready, MAGIC, __ATOMIC_RELEASE); }
    else { const double t0 = now_s(); while (__atomic_load_n(&c->h->ready, __ATOMIC_ACQUIRE) != MAGIC) { if (now_s() - t0 > 120.0) return ncclSystemError; usleep(100); } }
    if (barrier(c)) return ncclSystemError;
    if (rank == 0) shm_unlink(id.internal);    /* everyone has it mapped: the name can go */
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    munmap((void*)c->h, c->map_bytes);
    free(c);
    return ncclSuccess;
}

static const desc_t* nth(const box_t* b, int peer, int k) {
    for (uint32_t i = 0; i < b->n; ++i) if (b->d[i].peer == peer && k-- == 0) return &b->d[i];
    return NULL;
}

static ncclResult_t run_ops(op_t* ops, int n) {
    if (n == 0) return ncclSuccess;
    ncclComm_t c = ops[0].comm;
    for (int i = 0; i < n; ++i) if (ops[i].comm != c) return ncclInvalidArgument;   /* one communicator per group is all the caller does */
    for (int i = 0; i < n; ++i) if (hipStreamSynchronize(ops[i].stream) != hipSuccess) return ncclUnhandledCudaError;
    hdr_t* h = c->h;
    box_t* mine = &h->box[c->rank];
    uint8_t* area = c->data + (size_t)c->rank * h->cap;
    size_t off = 0;
    mine->n = 0;
    for (int i = 0; i < n; ++i) {
        if (ops[i].kind == OP_RECV) continue;
        const size_t bytes = ops[i].count * sizeof(float);
        if (off + bytes > h->cap || mine->n >= MAX_OPS) { fprintf(stderr, "fake_rccl: staging area too small (FAKE_RCCL_CAP_MB)\n"); return ncclInternalError; }
        if (hipMemcpy(area + off, ops[i].send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        desc_t* d = &mine->d[mine->n++];
        d->peer = ops[i].kind == OP_SEND ? ops[i].peer : (ops[i].kind == OP_ALLGATHER ? -1 : -2);
        d->off = off; d->count = ops[i].count;
        off += (bytes + 255) & ~(size_t)255;
    }
    if (barrier(c)) return ncclSystemError;
    int n_recv_from[MAX_RANKS] = {0}, n_ag = 0, n_ar = 0;
    ncclResult_t rc = ncclSuccess;
    for (int i = 0; i < n && rc == ncclSuccess; ++i) {
        const op_t* o = &ops[i];
        if (o->kind == OP_RECV) {
            const desc_t* d = nth(&h->box[o->peer], c->rank, n_recv_from[o->peer]++);
            if (!d || d->count != o->count) { fprintf(stderr, "fake_rccl: rank %d: receive %d from rank %d has no matching send of %zu floats\n", c->rank, n_recv_from[o->peer] - 1, o->peer, o->count); rc = ncclInvalidUsage; break; }
            if (hipMemcpy(o->recv, c->data + (size_t)o->peer * h->cap + d->off, o->count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        } else if (o->kind == OP_ALLGATHER) {
            for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
                const desc_t* d = nth(&h->box[r], -1, n_ag);
                if (!d || d->count != o->count) { rc = ncclInvalidUsage; break; }
                if (hipMemcpy((float*)o->recv + (size_t)r * o->count, c->data + (size_t)r * h->cap + d->off, o->count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
            }
            ++n_ag;
        } else if (o->kind == OP_ALLREDUCE) {
            float* acc = (float*)calloc(o->count, sizeof(float));
            for (int r = 0; r < c->world; ++r) {
                const desc_t* d = nth(&h->box[r], -2, n_ar);
                if (!d || d->count != o->count) { rc = ncclInvalidUsage; break; }
                const float* src = (const float*)(c->data + (size_t)r * h->cap + d->off);
                for (size_t k = 0; k < o->count; ++k) acc[k] += src[k];
            }
            if (rc == ncclSuccess && hipMemcpy(o->recv, acc, o->count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
            free(acc);
            ++n_ar;
        }
    }
    if (barrier(c)) return ncclSystemError;    /* nobody restages before everybody has read */
    return rc;
}

static ncclResult_t push(op_t o) {
    if (g_depth == 0) return run_ops(&o, 1);
    if (g_nops >= MAX_OPS) return ncclInternalError;
    g_ops[g_nops++] = o;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { if (g_depth++ == 0) g_nops = 0; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    const ncclResult_t r = run_ops(g_ops, g_nops);
    g_nops = 0;
    return r;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (t != ncclFloat || !c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    op_t o = {OP_SEND, peer, buf, NULL, count, c, s};
    return push(o);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (t != ncclFloat || !c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    op_t o = {OP_RECV, peer, NULL, buf, count, c, s};
    return push(o);
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    if (t != ncclFloat || !c) return ncclInvalidArgument;
    op_t o = {OP_ALLGATHER, -1, send, recv, count, c, s};
    return push(o);
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (t != ncclFloat || op != ncclSum || !c) return ncclInvalidArgument;
    op_t o = {OP_ALLREDUCE, -2, send, recv, count, c, s};
    return push(o);
}
