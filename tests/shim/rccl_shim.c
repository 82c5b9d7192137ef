/*
 * rccl_shim.c -- TEST INFRASTRUCTURE, never shipped: a stand-in for librccl that lets several
 * processes on ONE GPU execute the library's multi-rank exchanges for real.
 *
 * RCCL refuses two ranks of a communicator on the same device, and the boxes this is developed on
 * have one MI355X.  csrc/host/shard_rccl.c binds RCCL with dlopen; C21CM_RCCL_LIB points it at this
 * library instead, and the twelve entry points it binds are implemented here over a POSIX shared
 * memory segment with hipMemcpy staging:
 *
 *   - one channel per ordered pair (src -> dst): a single slot of SLOT bytes, a produced and a
 *     consumed counter, and the byte count of the message in flight;
 *   - ncclSend / ncclRecv with RCCL's semantics: inside ncclGroupStart .. ncclGroupEnd the calls
 *     are only recorded; ncclGroupEnd progresses all of them together, chunk by chunk and without
 *     blocking on any single one, so "every rank sends to every peer, then receives" completes --
 *     and outside a group a call is a group of one: a send returns only after the peer has
 *     received ALL of it (rendezvous, as RCCL: an unmatched send hangs -- here it times out);
 *   - messages between a pair match in posting order; a receive whose byte count differs from the
 *     matching send is an error (in RCCL it is undefined behaviour), as are a send to oneself, a
 *     peer out of range and a collective inside a group;
 *   - ncclReduce / ncclAllReduce / ncclBroadcast through the same channels (gather onto the root,
 *     combined on the host in rank order, sent back out): max of uint8 / int32 / uint64, sum of
 *     float64 -- what shard_rccl.c uses, anything else is ncclInvalidArgument;
 *   - every wait has a deadline (RCCL_SHIM_TIMEOUT_S, default 120 s): a mismatched exchange is a
 *     failed test, not a hung GPU box.
 *
 * Stream semantics: the stream is synchronised before the transfers and the copies are blocking,
 * which is a legal (stronger) ordering of what RCCL enqueues on the stream.
 *
 * Nothing here is algorithmic: the bytes arrive as sent.  What the tests check with it is the
 * caller -- the pairing of sends and receives inside one group, the offsets and strides of every
 * message, the status agreement -- at worlds 2 and 3 in real processes (tests/test_gpu_shard_shim.py).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

enum { MAX_WORLD = 16, MAX_OPS = 4096 };

typedef struct {
    volatile uint64_t produced, consumed; /* chunks written / read, ever */
    volatile uint64_t msg_bytes;          /* size of the message whose chunks are in flight */
    volatile uint64_t chunk_bytes;        /* size of the chunk in the slot */
    char pad[32];
} channel_ctl;

typedef struct {
    volatile uint32_t arrived, departed, world, magic;
    volatile uint64_t slot_bytes;
    char pad[40];
    channel_ctl ch[MAX_WORLD * MAX_WORLD];
} shim_header;

struct shim_comm {
    int rank, world;
    size_t slot, map_bytes;
    shim_header *h;
    unsigned char *slots; /* world * world slots */
    char name[128];
    /* statistics the tests read back (rccl_shim_stats) */
    uint64_t n_send, n_recv, n_groups, bytes_sent, bytes_recv, n_coll;
};

typedef struct {
    int is_send, peer;
    const void *sbuf;
    void *rbuf;
    size_t bytes, done;
    int started; /* send: header published; recv: header checked */
    struct shim_comm *comm;
    hipStream_t stream;
} p2p_op;

static __thread int g_depth;
static __thread int g_nops;
static __thread p2p_op g_ops[MAX_OPS];
static char g_err[256];
static struct shim_comm *g_last_comm;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double timeout_s(void) {
    const char *e = getenv("RCCL_SHIM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.;
    return v > 0. ? v : 120.;
}
static ncclResult_t fail(ncclResult_t rc, const char *fmt, const char *a, long b, long c) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    if (getenv("RCCL_SHIM_VERBOSE")) fprintf(stderr, "rccl_shim: %s\n", g_err);
    return rc;
}
static size_t dtype_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
/* RCCL_SHIM_HOST=1: plain memcpy and no stream synchronisation -- the shim's own protocol tests run on
 * hosts without a GPU (tests/test_rccl_shim.py), with host buffers */
static int host_only(void) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("RCCL_SHIM_HOST"); v = (e && e[0] == '1'); }
    return v;
}
static int copy_bytes(void *dst, const void *src, size_t n) {
    if (!n) return 0;
    if (host_only()) { memcpy(dst, src, n); return 0; }
    return hipMemcpy(dst, src, n, hipMemcpyDefault) != hipSuccess;
}
static int sync_stream(hipStream_t s) { return host_only() ? 0 : hipStreamSynchronize(s) != hipSuccess; }
static channel_ctl *chan(struct shim_comm *c, int src, int dst) { return &c->h->ch[src * c->world + dst]; }
static unsigned char *slot_of(struct shim_comm *c, int src, int dst) {
    return c->slots + (size_t)(src * c->world + dst) * c->slot;
}

/* one non-blocking step of an operation; returns 1 when it moved something, 0 when it has to wait,
 * < 0 on error */
static int step(p2p_op *o) {
    struct shim_comm *c = o->comm;
    if (o->is_send) {
        channel_ctl *ch = chan(c, c->rank, o->peer);
        if (o->done == o->bytes && o->started) return 0;         /* waiting for the last consume */
        if (ch->produced != ch->consumed) return 0;              /* slot busy */
        const size_t n = o->bytes - o->done < c->slot ? o->bytes - o->done : c->slot;
        if (copy_bytes(slot_of(c, c->rank, o->peer), (const char *)o->sbuf + o->done, n))
            return -1;
        ch->msg_bytes = o->bytes;
        ch->chunk_bytes = n;
        __sync_synchronize();
        ch->produced = ch->produced + 1;
        o->done += n;
        o->started = 1;
        return 1;
    }
    channel_ctl *ch = chan(c, o->peer, c->rank);
    if (ch->produced == ch->consumed) return 0; /* nothing there yet */
    __sync_synchronize();
    if (ch->msg_bytes != o->bytes) {
        fail(ncclInvalidArgument, "%s: a receive of %ld bytes met a send of %ld bytes", "ncclRecv", (long)o->bytes,
             (long)ch->msg_bytes);
        return -2;
    }
    const size_t n = ch->chunk_bytes;
    if (o->done + n > o->bytes) return -2;
    if (copy_bytes((char *)o->rbuf + o->done, slot_of(c, o->peer, c->rank), n))
        return -1;
    __sync_synchronize();
    ch->consumed = ch->consumed + 1;
    o->done += n;
    o->started = 1;
    return 1;
}
static int op_complete(p2p_op *o) {
    if (!o->started || o->done != o->bytes) return 0;
    if (!o->is_send) return 1;
    channel_ctl *ch = chan(o->comm, o->comm->rank, o->peer);
    return ch->produced == ch->consumed; /* the peer has taken the last chunk */
}

/* progress the recorded operations together.  Operations on the same channel complete in posting
 * order: only the first incomplete one of a (peer, direction) is stepped. */
static ncclResult_t run_ops(p2p_op *ops, int nops) {
    /* data written by earlier work on the streams must be complete before it is staged */
    for (int i = 0; i < nops; i++) {
        int seen = 0;
        for (int j = 0; j < i; j++) seen |= (ops[j].stream == ops[i].stream);
        if (!seen && sync_stream(ops[i].stream))
            return fail(ncclUnhandledCudaError, "%s: hipStreamSynchronize failed", "group", 0, 0);
    }
    const double deadline = now_s() + timeout_s();
    int remaining = nops;
    char *finished = (char *)calloc((size_t)nops + 1, 1);
    if (!finished) return ncclSystemError;
    while (remaining) {
        int moved = 0;
        for (int i = 0; i < nops; i++) {
            if (finished[i]) continue;
            int blocked = 0; /* an earlier unfinished operation on the same channel */
            for (int j = 0; j < i && !blocked; j++)
                blocked = !finished[j] && ops[j].is_send == ops[i].is_send && ops[j].peer == ops[i].peer &&
                          ops[j].comm == ops[i].comm;
            if (blocked) continue;
            const int s = step(&ops[i]);
            if (s < 0) {
                free(finished);
                return s == -2 ? ncclInvalidArgument
                               : fail(ncclUnhandledCudaError, "%s: hipMemcpy failed", "transfer", 0, 0);
            }
            moved |= s;
            if (op_complete(&ops[i])) {
                finished[i] = 1;
                remaining--;
                moved = 1;
            }
        }
        if (!moved) {
            if (now_s() > deadline) {
                int i = 0;
                while (i < nops && finished[i]) i++;
                free(finished);
                return fail(ncclSystemError, "%s with peer %ld timed out (%ld bytes): unmatched on the other side",
                            ops[i].is_send ? "ncclSend" : "ncclRecv", ops[i].peer, (long)ops[i].bytes);
            }
            usleep(50);
        }
    }
    free(finished);
    return ncclSuccess;
}

static ncclResult_t post(int is_send, const void *sbuf, void *rbuf, size_t count, ncclDataType_t t, int peer,
                         ncclComm_t comm, hipStream_t stream) {
    struct shim_comm *c = (struct shim_comm *)comm;
    const size_t w = dtype_size(t);
    if (!c || !w) return fail(ncclInvalidArgument, "%s: bad communicator or datatype", is_send ? "ncclSend" : "ncclRecv", 0, 0);
    if (peer < 0 || peer >= c->world)
        return fail(ncclInvalidArgument, "%s: peer %ld outside a world of %ld", is_send ? "ncclSend" : "ncclRecv", peer, c->world);
    if (peer == c->rank)
        return fail(ncclInvalidArgument, "%s: peer %ld is this rank", is_send ? "ncclSend" : "ncclRecv", peer, 0);
    if (count && !(is_send ? sbuf : rbuf))
        return fail(ncclInvalidArgument, "%s: NULL buffer", is_send ? "ncclSend" : "ncclRecv", 0, 0);
    if (g_nops >= MAX_OPS) return fail(ncclInternalError, "%s: too many operations in one group", "group", 0, 0);
    p2p_op *o = &g_ops[g_nops++];
    memset(o, 0, sizeof(*o));
    o->is_send = is_send;
    o->peer = peer;
    o->sbuf = sbuf;
    o->rbuf = rbuf;
    o->bytes = count * w;
    o->comm = c;
    o->stream = stream;
    if (is_send) c->n_send++, c->bytes_sent += o->bytes;
    else c->n_recv++, c->bytes_recv += o->bytes;
    if (g_depth) return ncclSuccess;
    const ncclResult_t rc = run_ops(g_ops, g_nops);
    g_nops = 0;
    return rc;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return post(1, buf, NULL, count, t, peer, comm, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return post(0, NULL, buf, count, t, peer, comm, s);
}
ncclResult_t ncclGroupStart(void) {
    g_depth++;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return fail(ncclInvalidUsage, "%s without ncclGroupStart", "ncclGroupEnd", 0, 0);
    if (--g_depth) return ncclSuccess;
    if (g_last_comm) g_last_comm->n_groups++;
    const ncclResult_t rc = g_nops ? run_ops(g_ops, g_nops) : ncclSuccess;
    g_nops = 0;
    return rc;
}

/* ---- communicator ---------------------------------------------------------------------------- */
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/c21shim_%d_%lx", (int)getpid(), (unsigned long)t.tv_nsec ^ ((unsigned long)t.tv_sec << 20));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world)
        return fail(ncclInvalidArgument, "%s: rank %ld / world %ld", "ncclCommInitRank", rank, world);
    if (id.internal[0] != '/' || memchr(id.internal, 0, sizeof(id.internal)) == NULL)
        return fail(ncclInvalidArgument, "%s: not an id of ncclGetUniqueId", "ncclCommInitRank", 0, 0);
    const char *e = getenv("RCCL_SHIM_SLOT_KB");
    const size_t slot = ((e && atol(e) > 0) ? (size_t)atol(e) : 1024) * 1024;
    struct shim_comm *c = (struct shim_comm *)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->world = world;
    c->slot = slot;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = sizeof(shim_header) + (size_t)world * world * slot;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
        if (fd >= 0) close(fd);
        free(c);
        return fail(ncclSystemError, "%s: shm_open / ftruncate failed (errno %ld)", "ncclCommInitRank", errno, 0);
    }
    void *m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        free(c);
        return fail(ncclSystemError, "%s: mmap failed (errno %ld)", "ncclCommInitRank", errno, 0);
    }
    c->h = (shim_header *)m;
    c->slots = (unsigned char *)m + sizeof(shim_header);
    c->h->world = (uint32_t)world; /* (every rank writes the same values into the zero-filled segment) */
    c->h->slot_bytes = slot;
    __sync_fetch_and_add(&c->h->arrived, 1);
    const double deadline = now_s() + timeout_s();
    while (c->h->arrived < (uint32_t)world) {
        if (now_s() > deadline) {
            munmap(m, c->map_bytes);
            if (rank == 0) shm_unlink(c->name);
            free(c);
            return fail(ncclSystemError, "%s: the other ranks never arrived (%ld of %ld)", "ncclCommInitRank", 0, world);
        }
        usleep(100);
    }
    /* everybody has the segment mapped: the name can go (the mapping stays) */
    if (__sync_add_and_fetch(&c->h->departed, 1) == (uint32_t)world) shm_unlink(c->name);
    *comm = (ncclComm_t)c;
    g_last_comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    struct shim_comm *c = (struct shim_comm *)comm;
    if (!c) return ncclSuccess;
    if (g_last_comm == c) g_last_comm = NULL;
    munmap((void *)c->h, c->map_bytes);
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((const struct shim_comm *)comm)->world;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t rc) {
    if (rc == ncclSuccess) return "no error";
    return g_err[0] ? g_err : "rccl_shim error";
}

/* ---- collectives over the same channels ------------------------------------------------------- */
static int combine(void *acc, const void *in, size_t count, ncclDataType_t t, ncclRedOp_t op) {
#define LOOP(T, EXPR) do { T *a = (T *)acc; const T *b = (const T *)in; for (size_t i = 0; i < count; i++) a[i] = (EXPR); } while (0)
    if (op == ncclMax && t == ncclUint8) LOOP(uint8_t, a[i] > b[i] ? a[i] : b[i]);
    else if (op == ncclMax && t == ncclInt32) LOOP(int32_t, a[i] > b[i] ? a[i] : b[i]);
    else if (op == ncclMax && t == ncclUint64) LOOP(uint64_t, a[i] > b[i] ? a[i] : b[i]);
    else if (op == ncclSum && t == ncclFloat64) LOOP(double, a[i] + b[i]);
    else if (op == ncclSum && t == ncclInt32) LOOP(int32_t, a[i] + b[i]);
    else return 1;
#undef LOOP
    return 0;
}
static ncclResult_t coll_guard(const char *what, struct shim_comm *c) {
    if (!c) return fail(ncclInvalidArgument, "%s: NULL communicator", what, 0, 0);
    if (g_depth) return fail(ncclInvalidUsage, "%s inside a group is not supported by the shim", what, 0, 0);
    c->n_coll++;
    return ncclSuccess;
}
static ncclResult_t one(int is_send, const void *sbuf, void *rbuf, size_t bytes, int peer, struct shim_comm *c, hipStream_t s) {
    p2p_op o;
    memset(&o, 0, sizeof(o));
    o.is_send = is_send, o.peer = peer, o.sbuf = sbuf, o.rbuf = rbuf, o.bytes = bytes, o.comm = c, o.stream = s;
    return run_ops(&o, 1);
}
ncclResult_t ncclReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream) {
    struct shim_comm *c = (struct shim_comm *)comm;
    ncclResult_t rc = coll_guard("ncclReduce", c);
    if (rc) return rc;
    const size_t bytes = count * dtype_size(t);
    if (!dtype_size(t) || root < 0 || root >= c->world) return fail(ncclInvalidArgument, "%s: bad datatype or root", "ncclReduce", 0, 0);
    if (c->rank != root) return one(1, send, NULL, bytes, root, c, stream);
    if (sync_stream(stream)) return ncclUnhandledCudaError;
    char *acc = (char *)malloc(bytes ? bytes : 1), *in = (char *)malloc(bytes ? bytes : 1);
    if (!acc || !in) { free(acc); free(in); return ncclSystemError; }
    /* rank order, the root's own contribution at its place */
    int first = 1;
    for (int r = 0; r < c->world && !rc; r++) {
        char *dst = first ? acc : in;
        if (r == root) {
            if (copy_bytes(dst, send, bytes)) rc = ncclUnhandledCudaError;
        } else {
            rc = one(0, NULL, dst, bytes, r, c, stream);
        }
        if (!rc && !first && combine(acc, in, count, t, op))
            rc = fail(ncclInvalidArgument, "%s: datatype / operation not implemented by the shim", "ncclReduce", 0, 0);
        first = 0;
    }
    if (!rc && copy_bytes(recv, acc, bytes)) rc = ncclUnhandledCudaError;
    free(acc);
    free(in);
    return rc;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm,
                           hipStream_t stream) {
    struct shim_comm *c = (struct shim_comm *)comm;
    ncclResult_t rc = coll_guard("ncclBroadcast", c);
    if (rc) return rc;
    const size_t bytes = count * dtype_size(t);
    if (!dtype_size(t) || root < 0 || root >= c->world) return fail(ncclInvalidArgument, "%s: bad datatype or root", "ncclBroadcast", 0, 0);
    if (c->rank != root) return one(0, NULL, recv, bytes, root, c, stream);
    if (sync_stream(stream)) return ncclUnhandledCudaError;
    p2p_op *ops = (p2p_op *)calloc((size_t)c->world, sizeof(p2p_op));
    if (!ops) return ncclSystemError;
    int n = 0;
    for (int r = 0; r < c->world; r++) {
        if (r == root) continue;
        ops[n].is_send = 1, ops[n].peer = r, ops[n].sbuf = send, ops[n].bytes = bytes, ops[n].comm = c, ops[n].stream = stream;
        n++;
    }
    rc = n ? run_ops(ops, n) : ncclSuccess;
    free(ops);
    if (!rc && recv != send && copy_bytes(recv, send, bytes)) rc = ncclUnhandledCudaError;
    return rc;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    ncclResult_t rc = ncclReduce(send, recv, count, t, op, 0, comm, stream);
    if (rc) return rc;
    return ncclBroadcast(recv, recv, count, t, 0, comm, stream);
}

/* what went through this process's (last) communicator: {sends, receives, groups, bytes sent, bytes
 * received, collectives} -- the tests assert that the exchanges really ran */
int rccl_shim_stats(uint64_t out[6]) {
    if (!g_last_comm) return 1;
    out[0] = g_last_comm->n_send, out[1] = g_last_comm->n_recv, out[2] = g_last_comm->n_groups;
    out[3] = g_last_comm->bytes_sent, out[4] = g_last_comm->bytes_recv, out[5] = g_last_comm->n_coll;
    return 0;
}
