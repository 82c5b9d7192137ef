/*
 * shard_rccl.c -- the R loop of ComputeIonizedBox sharded over the GPUs of one node, with the
 * exchange behind the C ABI (BASELINE.json north_star: "host/driver code stays
 * in C ... the R-loop of filter scales is partitioned across the 8 GPUs of one node with an RCCL
 * all-reduce over xGMI to combine the per-cell ionized mask").
 *
 * One process per GPU, every rank holds the (replicated) inputs:
 *   shard phase   radii n-1-rank, n-1-rank-world, ... > 0 with the fused kernels into a uint8
 *                 first-crossing grid (c21cm_ionize_shard_radii), or -- with a recombination
 *                 model -- into 64-bit (mean free path, Gamma_12) keys
 *                 (c21cm_ionize_shard_radii_keys)
 *   exchange      onto the rank that owns radius index 0: the uint8 grid packed to one bit per
 *                 cell and gathered point-to-point (N/8 bytes per link, OR on arrival), or
 *                 ONE ncclReduce(max) of the 64-bit keys with recombinations; "the largest
 *                 radius that ionises the cell" is order independent, so the result equals the
 *                 sequential loop (reference: src/py21cmfast/src/IonisationBox.c:1531-1588)
 *   finish        that rank applies the reduced grid, runs the cell-scale radius and the
 *                 post-loop, and (optionally) broadcasts the outputs
 * A reduce onto the finishing rank replaces the all-reduce of the plan: nobody else needs the
 * combined grid, and a reduce moves half the bytes of a ring all-reduce over the xGMI links.
 *
 * librccl is resolved with dlopen at first use, so the library itself has no RCCL dependency
 * (CPU-only hosts load it for the host scalars and the ABI checks), and inside a PyTorch process
 * the SONAME resolves to the copy torch already mapped: one RCCL per process.
 */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"
#include "c21cm_abi.h"

typedef struct {
    char internal[C21CM_SHARD_ID_BYTES];
} rccl_unique_id; /* ncclUniqueId, rccl.h */
typedef void *rccl_comm;
enum { RCCL_UINT8 = 1, RCCL_INT32 = 2, RCCL_UINT64 = 5, RCCL_FLOAT64 = 8 }; /* ncclDataType_t */
enum { RCCL_SUM = 0, RCCL_MAX = 2 };                                         /* ncclRedOp_t   */

/* The library binds RCCL with dlopen and therefore declares the handful of types it passes by
 * value itself.  Where the header is installed (the ROCm image: /opt/rocm/include/rccl/rccl.h)
 * the build checks those declarations against it, so a changed enum or id size is a compile
 * error, not a silently wrong reduce on the one path that needs eight GPUs to run (VERDICT r3
 * weak point 7). */
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <rccl/rccl.h>
#define C21CM_RCCL_HEADER_CHECKED 1
_Static_assert(sizeof(ncclUniqueId) == sizeof(rccl_unique_id), "ncclUniqueId is not 128 bytes");
_Static_assert(NCCL_UNIQUE_ID_BYTES == C21CM_SHARD_ID_BYTES, "NCCL_UNIQUE_ID_BYTES != C21CM_SHARD_ID_BYTES");
_Static_assert((int)ncclUint8 == RCCL_UINT8, "ncclUint8");
_Static_assert((int)ncclInt32 == RCCL_INT32, "ncclInt32");
_Static_assert((int)ncclUint64 == RCCL_UINT64, "ncclUint64");
_Static_assert((int)ncclFloat64 == RCCL_FLOAT64, "ncclFloat64");
_Static_assert((int)ncclSum == RCCL_SUM, "ncclSum");
_Static_assert((int)ncclMax == RCCL_MAX, "ncclMax");
_Static_assert(sizeof(ncclComm_t) == sizeof(rccl_comm), "ncclComm_t is not a pointer");
_Static_assert(sizeof(ncclResult_t) == sizeof(int) && (int)ncclSuccess == 0, "ncclResult_t");
_Static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
               "RCCL enums are not int-sized");
/* the prototypes this file calls through its function pointers (types only: no symbol of
 * librccl is referenced, the library stays loadable without it) */
#define SAME_PROTO(fn, ...) \
    _Static_assert(__builtin_types_compatible_p(__typeof__(&fn), ncclResult_t (*)(__VA_ARGS__)), #fn)
SAME_PROTO(ncclGetUniqueId, ncclUniqueId *);
SAME_PROTO(ncclCommInitRank, ncclComm_t *, int, ncclUniqueId, int);
SAME_PROTO(ncclCommDestroy, ncclComm_t);
SAME_PROTO(ncclCommCount, const ncclComm_t, int *);
SAME_PROTO(ncclAllReduce, const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
SAME_PROTO(ncclReduce, const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
SAME_PROTO(ncclBroadcast, const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME_PROTO(ncclSend, const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME_PROTO(ncclRecv, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME_PROTO(ncclGroupStart, void);
SAME_PROTO(ncclGroupEnd, void);
_Static_assert(sizeof(hipStream_t) == sizeof(void *), "hipStream_t is not a pointer");
#undef SAME_PROTO
#endif
#endif
#ifndef C21CM_RCCL_HEADER_CHECKED
#define C21CM_RCCL_HEADER_CHECKED 0
#endif
int c21cm_shard_rccl_header_checked(void) { return C21CM_RCCL_HEADER_CHECKED; }

static struct {
    void *lib;
    int ready, rank, world;
    rccl_comm comm;
    int (*get_unique_id)(rccl_unique_id *);
    int (*comm_init_rank)(rccl_comm *, int, rccl_unique_id, int);
    int (*comm_destroy)(rccl_comm);
    int (*comm_count)(rccl_comm, int *);
    int (*all_reduce)(const void *, void *, size_t, int, int, rccl_comm, void *);
    int (*reduce)(const void *, void *, size_t, int, int, int, rccl_comm, void *);
    int (*broadcast)(const void *, void *, size_t, int, int, rccl_comm, void *);
    int (*send)(const void *, size_t, int, int, rccl_comm, void *);
    int (*recv)(void *, size_t, int, int, rccl_comm, void *);
    int (*group_start)(void);
    int (*group_end)(void);
    const char *(*error_string)(int);
} R;

/* (256 and 257 are this file's; ionize_driver.c holds 258 and 259 -- each file's enum names the other's.
 * 253 collided with the Eulerian loop's sparse x_e buffer,
 * whose reallocation freed the status word a rank out of memory still needs -- ADVICE r4) */
enum { WS_SHARD_GRID = 140, WS_SHARD_STAGE = 141, WS_SHARD_SCALARS = 142, WS_SHARD_BITS = 143,
       WS_SHARD_STATUS = 256, WS_SHARD_SLABBITS = 257 };

static int rccl_check(int rc, const char *what);

/* The exchange of the uint8 first-crossing grid.  The finish phase only tests it for non-zero,
 * so ONE BIT per cell suffices: every rank packs its grid (N/8 bytes), the finishing rank
 * receives the packed grids of all others at once -- point-to-point, so its seven xGMI links
 * carry one grid each in parallel -- and ORs them back into a byte grid.  At 1024^3 that is
 * 134 MB per link (~1.3 ms at ~100 GB/s) against a 1.07 GB ring reduce whose every step is bound
 * by one link (~10 ms).  C21CM_SHARD_EXCHANGE=reduce keeps ncclReduce(uint8, max), which also
 * preserves the radius index in the grid (nothing downstream reads it). */
static int mask_exchange_is_reduce(void) {
    const char *e = getenv("C21CM_SHARD_EXCHANGE");
    return (e && e[0] == 'r') || !R.send || !R.recv || !R.group_start || !R.group_end;
}
/* local half: the bit buffer and the pack launch (everything that can fail on this rank alone
 * happens BEFORE the ranks agree on a status; nothing between the agreement and the transfers
 * returns early -- ADVICE r3) */
static int exchange_mask_local(const unsigned char *fc, size_t ntot, int owner, unsigned **bits_out,
                               void *stream) {
    *bits_out = NULL;
    if (mask_exchange_is_reduce()) return 0;
    const size_t nwords = (ntot + 31) / 32;
    const int slots = (R.rank == owner) ? R.world : 1;
    unsigned *bits = (unsigned *)c21hip_ws(WS_SHARD_BITS, sizeof(unsigned) * nwords * (size_t)slots);
    if (!bits) return C21CM_MEMORY_ALLOC_ERROR;
    *bits_out = bits;
    return c21hip_pack_mask_bits(fc, bits + (R.rank == owner ? (size_t)owner * nwords : 0), ntot, stream);
}
static int exchange_mask(unsigned char *fc, unsigned *bits, size_t ntot, int owner, void *stream) {
    if (mask_exchange_is_reduce())
        return rccl_check(R.reduce(fc, fc, ntot, RCCL_UINT8, RCCL_MAX, owner, R.comm, stream),
                          "ncclReduce(first_cross)");
    const size_t nwords = (ntot + 31) / 32;
    int st = 0;
    if (R.world == 1) return c21hip_or_unpack_mask_bits(bits, nwords, 1, fc, ntot, stream);
    if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
    if (R.rank == owner) {
        for (int r = 0; r < R.world && !st; r++)
            if (r != owner)
                st = rccl_check(R.recv(bits + (size_t)r * nwords, nwords * sizeof(unsigned),
                                       RCCL_UINT8, r, R.comm, stream), "ncclRecv(mask bits)");
    } else {
        st = rccl_check(R.send(bits, nwords * sizeof(unsigned), RCCL_UINT8, owner, R.comm, stream),
                        "ncclSend(mask bits)");
    }
    const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
    if (st || st2) return st ? st : st2;
    if (R.rank == owner) return c21hip_or_unpack_mask_bits(bits, nwords, R.world, fc, ntot, stream);
    return 0;
}

/* The exchange of the fused recombination loop: uint8 first-crossing index + float Gamma_12 per
 * cell, the winner being the rank with the larger index.  Two hops over direct links, both moving
 * 5 N / world bytes per link (at 1024^3 x 8: 671 MB, ~7-10 ms each, against the 8.6 GB the
 * ncclReduce(uint64, max) of the keys pushes through a ring):
 *   1. reduce-scatter by cell slabs: rank r sends peer p the (contiguous) slab p of both grids and
 *      receives its own slab from everybody; c21hip_combine_cross_g12 keeps the winners;
 *   2. the finishing rank receives every combined slab in place.
 * Slab bounds as for the TsBox sums (multiples of 4 cells). */
enum { WS_SHARD_RC_MASK = 248, WS_SHARD_RC_G12 = 249 };
static int agree_status(int st_local, void *stream);
static size_t rc_slab_maxlen(size_t ntot, int world) {
    size_t maxlen = 0;
    for (int r = 0; r < world; r++) {
        const size_t len = c21hip_ts_slab_begin(ntot, world, r + 1) - c21hip_ts_slab_begin(ntot, world, r);
        if (len > maxlen) maxlen = len;
    }
    return (maxlen + 15) & ~(size_t)15;
}
/* local half (before the agreement): the receive buffers of hop 1 */
static int exchange_cross_g12_local(size_t ntot) {
    const int world = R.world;
    if (world == 1) return 0;
    if (!R.send || !R.recv || !R.group_start || !R.group_end) {
        c21hip_set_error("shard: this librccl has no ncclSend / ncclRecv");
        return C21CM_IO_ERROR;
    }
    const size_t maxlen = rc_slab_maxlen(ntot, world);
    if (!c21hip_ws(WS_SHARD_RC_MASK, maxlen * (size_t)(world - 1)) ||
        !c21hip_ws(WS_SHARD_RC_G12, maxlen * (size_t)(world - 1) * sizeof(float)))
        return C21CM_MEMORY_ALLOC_ERROR;
    return 0;
}
static int exchange_cross_g12(unsigned char *fc, float *g12, size_t ntot, int owner, void *stream) {
    const int rank = R.rank, world = R.world;
    if (world == 1) return 0;
    const size_t maxlen = rc_slab_maxlen(ntot, world);
    const size_t c0 = c21hip_ts_slab_begin(ntot, world, rank);
    const size_t len = c21hip_ts_slab_begin(ntot, world, rank + 1) - c0;
    /* sized by exchange_cross_g12_local: these calls only return the cached pointers */
    unsigned char *pm = (unsigned char *)c21hip_ws(WS_SHARD_RC_MASK, maxlen * (size_t)(world - 1));
    float *pg = (float *)c21hip_ws(WS_SHARD_RC_G12, maxlen * (size_t)(world - 1) * sizeof(float));
    int st = 0;
    if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
    for (int p = 0, slot = 0; p < world && !st; p++) {
        if (p == rank) continue;
        const size_t p0 = c21hip_ts_slab_begin(ntot, world, p);
        const size_t plen = c21hip_ts_slab_begin(ntot, world, p + 1) - p0;
        if (plen) {
            st = rccl_check(R.send(fc + p0, plen, RCCL_UINT8, p, R.comm, stream), "ncclSend(cross slab)");
            if (!st)
                st = rccl_check(R.send(g12 + p0, plen * sizeof(float), RCCL_UINT8, p, R.comm, stream),
                                "ncclSend(G12 slab)");
        }
        if (len && !st) {
            st = rccl_check(R.recv(pm + (size_t)slot * maxlen, len, RCCL_UINT8, p, R.comm, stream),
                            "ncclRecv(cross slab)");
            if (!st)
                st = rccl_check(R.recv(pg + (size_t)slot * maxlen, len * sizeof(float), RCCL_UINT8, p,
                                       R.comm, stream), "ncclRecv(G12 slab)");
        }
        slot++;
    }
    {
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    /* the combine is local work between two hops: the ranks agree again before the second one */
    st = c21hip_combine_cross_g12(fc + c0, g12 + c0, pm, pg, world - 1, maxlen, len, stream);
    if ((st = agree_status(st, stream))) return st;
    if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
    if (rank == owner) {
        for (int p = 0; p < world && !st; p++) {
            if (p == owner) continue;
            const size_t p0 = c21hip_ts_slab_begin(ntot, world, p);
            const size_t plen = c21hip_ts_slab_begin(ntot, world, p + 1) - p0;
            if (!plen) continue;
            st = rccl_check(R.recv(fc + p0, plen, RCCL_UINT8, p, R.comm, stream), "ncclRecv(combined slab)");
            if (!st)
                st = rccl_check(R.recv(g12 + p0, plen * sizeof(float), RCCL_UINT8, p, R.comm, stream),
                                "ncclRecv(combined G12)");
        }
    } else if (len) {
        st = rccl_check(R.send(fc + c0, len, RCCL_UINT8, owner, R.comm, stream), "ncclSend(combined slab)");
        if (!st)
            st = rccl_check(R.send(g12 + c0, len * sizeof(float), RCCL_UINT8, owner, R.comm, stream),
                            "ncclSend(combined G12)");
    }
    {
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    return 0;
}

int c21cm_shard_combine_cross_g12(unsigned char *mask, float *g12, const unsigned char *peer_mask,
                                  const float *peer_g12, int n_peers, size_t stride, size_t n,
                                  void *stream) {
    return c21hip_combine_cross_g12(mask, g12, peer_mask, peer_g12, n_peers, stride, n, stream);
}

/* ---- in-process emulation of the transport (test hook) -------------------------------------
 * The boxes this is developed on have one GPU, and RCCL refuses two ranks on one device.  To
 * exercise c21cm_ionize_sharded for world > 1 anyway -- the radius deal, the slot arithmetic of
 * the bit gather, the key reduce, the owner's finish -- the ranks can be run ONE AFTER THE OTHER in
 * a single process (non-owners first, the owner last) against a device "mailbox":
 *   send(peer)      copies into the mailbox region of the sending rank
 *   recv(peer)      copies out of the mailbox region of `peer`
 *   reduce(max)     non-root: mailbox = max(mailbox, buf);  root: buf = max(buf, mailbox)
 * (the mailbox must be zeroed before a round).  Only the transport is replaced; everything else
 * is the code the 8-GPU run executes.  Broadcasts are not emulated (the root would have to run
 * first). */
static struct {
    unsigned char *box;
    size_t bytes;
} EMU;

static int emu_region(size_t count_bytes, int rank, unsigned char **p) {
    if (!EMU.box || (size_t)(rank + 1) * count_bytes > EMU.bytes) {
        c21hip_set_error("shard emulation: mailbox too small");
        return 1;
    }
    *p = EMU.box + (size_t)rank * count_bytes;
    return 0;
}
static int emu_send(const void *buf, size_t count, int dtype, int peer, rccl_comm comm, void *stream) {
    (void)dtype, (void)peer, (void)comm;
    unsigned char *p;
    if (emu_region(count, R.rank, &p)) return 1;
    return c21hip_d2d(p, buf, count, stream);
}
static int emu_recv(void *buf, size_t count, int dtype, int peer, rccl_comm comm, void *stream) {
    (void)dtype, (void)comm;
    unsigned char *p;
    if (emu_region(count, peer, &p)) return 1;
    return c21hip_d2d(buf, p, count, stream);
}
static int emu_reduce(const void *send, void *recv, size_t count, int dtype, int op, int root,
                      rccl_comm comm, void *stream) {
    (void)comm;
    const int width = dtype == RCCL_UINT8 ? 1 : (dtype == RCCL_UINT64 ? 8 : 0);
    if (!width || op != RCCL_MAX || send != recv || count * (size_t)width > EMU.bytes) {
        c21hip_set_error("shard emulation: only in-place max-reduces of uint8 / uint64 fit the mailbox");
        return 1;
    }
    return R.rank == root ? c21hip_max_into(recv, EMU.box, count, width, stream)
                          : c21hip_max_into(EMU.box, send, count, width, stream);
}
static int emu_bcast(const void *s, void *r, size_t n, int t, int root, rccl_comm c, void *st) {
    (void)s, (void)r, (void)n, (void)t, (void)root, (void)c, (void)st;
    c21hip_set_error("shard emulation: broadcasts are not emulated");
    return 1;
}
static int emu_group(void) { return 0; }
static const char *emu_error(int rc) {
    (void)rc;
    return "emulated transport";
}

int c21cm_shard_emulate(int rank, int world, void *mailbox, size_t mailbox_bytes) {
    if (world < 1 || rank < 0 || rank >= world || !mailbox || !c21hip_is_device_ptr(mailbox)) {
        c21hip_set_error("shard emulation: bad rank / world / mailbox");
        return C21CM_VALUE_ERROR;
    }
    R.comm = NULL;
    R.reduce = emu_reduce;
    R.broadcast = emu_bcast;
    R.send = emu_send;
    R.recv = emu_recv;
    R.group_start = R.group_end = emu_group;
    R.error_string = emu_error;
    R.comm_destroy = NULL;
    EMU.box = (unsigned char *)mailbox;
    EMU.bytes = mailbox_bytes;
    R.rank = rank;
    R.world = world;
    R.ready = 2; /* emulated */
    return 0;
}

static int rccl_load(void) {
    if (R.lib && R.ready != 2 && R.get_unique_id) return 0;
    /* C21CM_RCCL_LIB: another library exporting the same twelve entry points.  The test suite points it
     * at tests/shim/librccl_shim.so (shared memory between processes on ONE GPU, where RCCL refuses to
     * form a communicator), so that every send / receive / group of this file executes at worlds 2 and 3
     * (VERDICT r5 item 1).  Bound RTLD_LOCAL: its ncclXxx symbols must not interpose a librccl torch
     * has mapped. */
    const char *over = getenv("C21CM_RCCL_LIB");
    if (over && over[0]) {
        R.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) {
            c21hip_set_error("shard: C21CM_RCCL_LIB=%s could not be loaded (%s)", over, dlerror());
            return C21CM_IO_ERROR;
        }
    }
    const char *names[] = {"librccl.so.1", "librccl.so", NULL};
    for (int i = 0; names[i] && !R.lib; i++) R.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) {
        c21hip_set_error("shard: librccl could not be loaded (%s)", dlerror());
        return C21CM_IO_ERROR;
    }
    *(void **)&R.get_unique_id = dlsym(R.lib, "ncclGetUniqueId");
    *(void **)&R.comm_init_rank = dlsym(R.lib, "ncclCommInitRank");
    *(void **)&R.comm_destroy = dlsym(R.lib, "ncclCommDestroy");
    *(void **)&R.comm_count = dlsym(R.lib, "ncclCommCount");
    *(void **)&R.all_reduce = dlsym(R.lib, "ncclAllReduce");
    *(void **)&R.reduce = dlsym(R.lib, "ncclReduce");
    *(void **)&R.broadcast = dlsym(R.lib, "ncclBroadcast");
    *(void **)&R.send = dlsym(R.lib, "ncclSend");
    *(void **)&R.recv = dlsym(R.lib, "ncclRecv");
    *(void **)&R.group_start = dlsym(R.lib, "ncclGroupStart");
    *(void **)&R.group_end = dlsym(R.lib, "ncclGroupEnd");
    *(void **)&R.error_string = dlsym(R.lib, "ncclGetErrorString");
    if (!R.get_unique_id || !R.comm_init_rank || !R.comm_destroy || !R.reduce || !R.broadcast) {
        c21hip_set_error("shard: librccl lacks an expected symbol");
        return C21CM_IO_ERROR;
    }
    return 0;
}

static int rccl_check(int rc, const char *what) {
    if (rc == 0) return 0;
    c21hip_set_error("shard: %s failed: %s", what, R.error_string ? R.error_string(rc) : "?");
    return C21CM_IO_ERROR;
}

int c21cm_shard_unique_id(void *id128) {
    if (R.ready == 2) c21cm_shard_finalize(); /* leave the emulated transport first */
    int st = rccl_load();
    if (st) return st;
    if (!id128) return C21CM_VALUE_ERROR;
    rccl_unique_id id;
    if ((st = rccl_check(R.get_unique_id(&id), "ncclGetUniqueId"))) return st;
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int c21cm_shard_init(int rank, int world, const void *id128) {
    if (R.ready == 2) c21cm_shard_finalize();
    int st = rccl_load();
    if (st) return st;
    if (!id128 || world < 1 || rank < 0 || rank >= world) {
        c21hip_set_error("shard: bad rank %d / world %d", rank, world);
        return C21CM_VALUE_ERROR;
    }
    if (R.ready) c21cm_shard_finalize();
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    if ((st = rccl_check(R.comm_init_rank(&R.comm, world, id, rank), "ncclCommInitRank"))) return st;
    R.rank = rank;
    R.world = world;
    R.ready = 1;
    (void)c21hip_ws(WS_SHARD_STATUS, 64); /* the status word of agree_status */
    return 0;
}

/* 1: the communicator is a real RCCL one (not the in-process emulation of the tests) */
int c21cm_shard_is_rccl(void) { return R.ready == 1; }

/* Do ALL ranks say yes?  (min over the ranks of a local 0 / 1; collective, on the NULL stream.)
 * ComputeTsBox decides with it whether to shard: ranks that judged from their own pointer kinds
 * alone could take different paths and deadlock. */
static int agree_status(int st_local, void *stream);
int c21cm_shard_all_agree(int local_yes) {
    if (R.ready != 1 || R.world < 2) return local_yes ? 1 : 0;
    return agree_status(local_yes ? 0 : 1, NULL) == 0;
}

int c21cm_shard_finalize(void) {
    if (R.ready == 1 && R.comm && R.comm_destroy) (void)R.comm_destroy(R.comm);
    if (R.ready == 2) { /* drop the emulated transport: the next init resolves RCCL again */
        R.reduce = NULL;
        R.get_unique_id = NULL;
        EMU.box = NULL;
    }
    R.comm = NULL;
    R.ready = 0;
    return 0;
}

int c21cm_shard_info(int *rank, int *world) {
    if (!R.ready) return C21CM_VALUE_ERROR;
    if (rank) *rank = R.rank;
    if (world) *world = R.world;
    return 0;
}

/* the rank the round-robin deal would hand radius index 0 to: it has the fewest shard radii */
int c21cm_shard_owner(int n_radii, int world) { return world > 0 ? (n_radii - 1) % world : 0; }

/* broadcast one output array (host or device) from `root` */
static int bcast_array(float *p, size_t bytes, int root, void *stream) {
    if (!p) return 0; /* (the same arrays are NULL on every rank: the struct layout is the caller's) */
    int st = 0, s2;
    if (c21hip_is_device_ptr(p))
        return rccl_check(R.broadcast(p, p, bytes, RCCL_UINT8, root, R.comm, stream), "ncclBroadcast");
    void *d = c21hip_ws(WS_SHARD_STAGE, bytes); /* sized before the agreement */
    if (!d) return C21CM_MEMORY_ALLOC_ERROR;
    if (R.rank == root) st = c21hip_h2d(d, p, bytes, stream);
    /* a failed staging copy does not keep this rank out of the collective (the caller agrees on
     * the outcome afterwards) */
    if ((s2 = rccl_check(R.broadcast(d, d, bytes, RCCL_UINT8, root, R.comm, stream), "ncclBroadcast")) && !st)
        st = s2;
    if (R.rank != root && !st) st = c21hip_d2h(p, d, bytes, stream);
    s2 = c21hip_sync(stream); /* the staging slot is reused by the next array */
    return st ? st : s2;
}

/* Every rank learns whether ANY rank failed its local phase before the data collectives start
 * (a rank that returned early left the others blocked in ncclReduce / ncclRecv for
 * ever): a max-all-reduce of one int32.  Returns the rank's own status if it failed, a generic
 * error if another rank did, 0 otherwise.  Emulated transports run the ranks one by one: skipped. */
static int agree_status(int st_local, void *stream) {
    if (R.ready != 1 || R.world < 2 || !R.all_reduce) return st_local;
    /* the status word lives in a slot c21cm_shard_init sized (WS_SHARD_STATUS is used by nothing
     * else), so this rank joins the all-reduce even when its local phase failed on memory */
    int *d = (int *)c21hip_ws(WS_SHARD_STATUS, 64);
    int flag = st_local ? 1 : 0;
    if (!d) return st_local ? st_local : C21CM_MEMORY_ALLOC_ERROR;
    if (c21hip_h2d(d, &flag, sizeof(int), stream) ||
        rccl_check(R.all_reduce(d, d, 1, RCCL_INT32, RCCL_MAX, R.comm, stream), "ncclAllReduce(status)") ||
        c21hip_d2h(&flag, d, sizeof(int), stream) || c21hip_sync(stream))
        return st_local ? st_local : C21CM_IO_ERROR;
    if (st_local) return st_local;
    if (flag) {
        c21hip_set_error("shard: another rank failed its local phase; this call is abandoned on every rank");
        return C21CM_IO_ERROR;
    }
    return 0;
}

/* Device time of the three phases of this rank's last c21cm_ionize_sharded call (ms: shard phase,
 * exchange, finish; synchronises on the last mark).  The exchange entry includes the wait for the
 * slowest peer.  Returns 0, or C21CM_VALUE_ERROR when no sharded call has completed. */
static void *g_phase_ev[4];
static int g_phase_valid;
int c21cm_shard_last_phases(double ms[3]) {
    if (!g_phase_valid || !g_phase_ev[0] || !g_phase_ev[3]) return C21CM_VALUE_ERROR;
    if (c21hip_event_synchronize(g_phase_ev[3])) return C21CM_IO_ERROR;
    for (int i = 0; i < 3; i++) ms[i] = c21hip_event_elapsed_ms(g_phase_ev[i], g_phase_ev[i + 1]);
    return 0;
}
/* ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount); 0: none / emulated */
int c21cm_shard_comm_count(void) {
    int n = 0;
    if (R.ready == 1 && R.comm && R.comm_count && R.comm_count(R.comm, &n) == 0) return n;
    return 0;
}

/* ---- finish phase by cell slabs (round 5; ionize_driver.c: c21cm_ionize_shard_finish_slab) -------------
 * exchange 1 (before the finish): every rank packs its first-crossing grid to one bit per cell and sends
 *   peer p the words of p's slab -- all W - 1 messages in one group over the direct links (17 MB per
 *   link at 1024^3 x 8 where the gather onto one rank moved 134) -- and ORs what it receives into the
 *   bytes of its own slab;
 * exchange 2 (inside the finish, after the slab's sweep): all-gather of the chunks' partial sums
 *   (2 x 8 bytes per chunk, <= 32 KB in all), max-all-reduce of the non-finite flag, and -- only when the
 *   caller wants whole boxes on every rank -- the all-gather of the output slabs.
 * C21CM_SHARD_FINISH=owner keeps the finish on one rank (gather of whole packed grids). */
static int slab_finish_wanted(const c21cm_ionize_spec *spec) {
    const char *e = getenv("C21CM_SHARD_FINISH");
    if (e && e[0] == 'o') return 0;
    if (!R.send || !R.recv || !R.group_start || !R.group_end) return 0;
    if (R.ready == 2) return 0; /* the emulated transport runs its ranks one after the other */
    return c21cm_ionize_shard_slab_supported(spec);
}
static size_t slab_words(const c21cm_ionize_spec *spec, int r, int world, size_t *word0) {
    size_t c0 = 0, c1 = 0;
    (void)c21cm_ionize_shard_slab(spec, r, world, NULL, NULL, &c0, &c1, NULL, NULL);
    if (word0) *word0 = c0 / 32; /* slab starts are multiples of 512 cells */
    return (c1 - c0 + 31) / 32;
}
typedef struct {
    int gather_outputs;
    const c21cm_ionize_spec *spec;
} slab_exchange_arg;
static int slab_exchange_rccl(void *user, const c21cm_shard_slab_state *s, int local_status, void *stream) {
    const slab_exchange_arg *arg = (const slab_exchange_arg *)user;
    int st = agree_status(local_status, stream);
    if (st) return st;
    const int rank = R.rank, world = R.world;
    if (world < 2) return 0;
    if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
    for (int p = 0; p < world && !st; p++) {
        if (p == rank) continue;
        int pb = 0, pe = 0;
        (void)c21cm_ionize_shard_slab(arg->spec, p, world, &pb, &pe, NULL, NULL, NULL, NULL);
        const size_t mine = (size_t)(s->chunk_end - s->chunk_begin) * sizeof(double);
        const size_t theirs = (size_t)(pe - pb) * sizeof(double);
        if (mine) {
            st = rccl_check(R.send(s->partials_stars + s->chunk_begin, mine, RCCL_UINT8, p, R.comm, stream),
                            "ncclSend(chunk sums)");
            if (!st)
                st = rccl_check(R.send(s->partials_xh + s->chunk_begin, mine, RCCL_UINT8, p, R.comm, stream),
                                "ncclSend(chunk sums)");
        }
        if (theirs && !st) {
            st = rccl_check(R.recv(s->partials_stars + pb, theirs, RCCL_UINT8, p, R.comm, stream),
                            "ncclRecv(chunk sums)");
            if (!st)
                st = rccl_check(R.recv(s->partials_xh + pb, theirs, RCCL_UINT8, p, R.comm, stream),
                                "ncclRecv(chunk sums)");
        }
    }
    {
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    if ((st = rccl_check(R.all_reduce(s->flag, s->flag, 1, RCCL_INT32, RCCL_MAX, R.comm, stream),
                         "ncclAllReduce(flag)")))
        return st;
    if (!arg->gather_outputs) return 0;
    if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
    for (int b = 0; b < (arg->gather_outputs == 2 ? 1 : 3) && !st; b++) { /* (2: the neutral fraction alone) */
        if (!s->out[b]) continue;
        for (int p = 0; p < world && !st; p++) {
            if (p == rank) continue;
            size_t p0 = 0, p1 = 0;
            (void)c21cm_ionize_shard_slab(arg->spec, p, world, NULL, NULL, &p0, &p1, NULL, NULL);
            if (s->cell_end > s->cell_begin)
                st = rccl_check(R.send(s->out[b] + s->cell_begin, (s->cell_end - s->cell_begin) * sizeof(float),
                                       RCCL_UINT8, p, R.comm, stream), "ncclSend(output slab)");
            if (p1 > p0 && !st)
                st = rccl_check(R.recv(s->out[b] + p0, (p1 - p0) * sizeof(float), RCCL_UINT8, p, R.comm, stream),
                                "ncclRecv(output slab)");
        }
    }
    {
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    return 0;
}
/* exchange 1: local half (buffers + pack, before the agreement) and the transfers */
static int slab_mask_local(const c21cm_ionize_spec *spec, const unsigned char *fc, size_t ntot,
                           unsigned **send_out, unsigned **recv_out, void *stream) {
    const int rank = R.rank, world = R.world;
    const size_t nwords = (ntot + 31) / 32;
    size_t w0 = 0;
    const size_t mine = slab_words(spec, rank, world, &w0);
    unsigned *sendb = (unsigned *)c21hip_ws(WS_SHARD_BITS, sizeof(unsigned) * nwords);
    unsigned *recvb = (unsigned *)c21hip_ws(WS_SHARD_SLABBITS, sizeof(unsigned) * (mine ? mine : 1) * (size_t)world);
    if (!sendb || !recvb) return C21CM_MEMORY_ALLOC_ERROR;
    *send_out = sendb;
    *recv_out = recvb;
    int st = c21hip_pack_mask_bits(fc, sendb, ntot, stream);
    if (!st && mine) /* the own piece joins the received ones */
        st = c21hip_d2d(recvb + (size_t)rank * mine, sendb + w0, mine * sizeof(unsigned), stream);
    return st;
}
static int slab_mask_exchange(const c21cm_ionize_spec *spec, unsigned char *fc, unsigned *sendb,
                              unsigned *recvb, void *stream) {
    const int rank = R.rank, world = R.world;
    size_t w0 = 0, c0 = 0, c1 = 0;
    const size_t mine = slab_words(spec, rank, world, &w0);
    (void)c21cm_ionize_shard_slab(spec, rank, world, NULL, NULL, &c0, &c1, NULL, NULL);
    int st = 0;
    if (world > 1) {
        if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
        for (int p = 0; p < world && !st; p++) {
            if (p == rank) continue;
            size_t pw0 = 0;
            const size_t theirs = slab_words(spec, p, world, &pw0);
            if (theirs)
                st = rccl_check(R.send(sendb + pw0, theirs * sizeof(unsigned), RCCL_UINT8, p, R.comm, stream),
                                "ncclSend(slab bits)");
            if (mine && !st)
                st = rccl_check(R.recv(recvb + (size_t)p * mine, mine * sizeof(unsigned), RCCL_UINT8, p,
                                       R.comm, stream), "ncclRecv(slab bits)");
        }
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    if (!mine) return 0;
    return c21hip_or_unpack_mask_bits(recvb, mine, world, fc + c0, c1 - c0, stream);
}
int c21cm_shard_pack_mask_bits(const unsigned char *first_cross, unsigned *bits, size_t n, void *stream) {
    return c21hip_pack_mask_bits(first_cross, bits, n, stream);
}
int c21cm_shard_or_unpack_mask_bits(const unsigned *bits, size_t stride_words, int world,
                                    unsigned char *first_cross, size_t n, void *stream) {
    return c21hip_or_unpack_mask_bits(bits, stride_words, world, first_cross, n, stream);
}
/* What a sharded ComputeIonizedBox leaves in the output arrays (the `broadcast` argument of
 * c21cm_ionize_sharded): 1 = whole boxes on every rank, 2 = the whole neutral-fraction box on every rank
 * (what ComputeBrightnessTemp and a power spectrum need: 4 of the 12 bytes per cell), z_reion and T_k
 * slab-resident, 0 = nothing beyond what the finish leaves (a rank's slab / the owner's box), -1 = auto: slab-resident where the finish runs by slabs, the owner's
 * box broadcast otherwise.  c21cm_shard_set_output() or the environment (C21CM_SHARD_OUTPUT = all |
 * xH | none | auto; C21CM_SHARD_BCAST = 1 | 0 as before round 5); default auto. */
static int g_output_mode = -2;
int c21cm_shard_set_output(int mode) {
    if (mode < -1 || mode > 2) return C21CM_VALUE_ERROR;
    g_output_mode = mode;
    return 0;
}
int c21cm_shard_output_mode(void) {
    if (g_output_mode != -2) return g_output_mode;
    const char *o = getenv("C21CM_SHARD_OUTPUT"), *b = getenv("C21CM_SHARD_BCAST");
    if (o && o[0] == 'a' && o[1] == 'l') return 1;
    if (o && o[0] == 'x') return 2; /* xH: whole neutral-fraction boxes, z_reion / T_k slab-resident */
    if (o && o[0] == 'n') return 0;
    if (b && b[0] == '1') return 1;
    if (b && b[0] == '0') return 0;
    return -1;
}
static int g_last_finish_slab;
int c21cm_shard_last_finish_was_slab(void) { return g_last_finish_slab; }

/* C21CM_SHARD_EXCHANGE=keys keeps the 64-bit key reduce for every recombination model */
static int shard_rc_exchange(void) {
    const char *e = getenv("C21CM_SHARD_EXCHANGE");
    return !(e && e[0] == 'k');
}

int c21cm_ionize_sharded(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                         const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                         const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                         int broadcast, void *stream) {
    if (!R.ready) {
        c21hip_set_error("shard: c21cm_shard_init has not been called");
        return C21CM_VALUE_ERROR;
    }
    if (!spec || !box) return C21CM_VALUE_ERROR;
    int st = 0;
    const int rank = R.rank, world = R.world;
    const int owner = c21cm_shard_owner(spec->n_radii, world);
    const size_t ntot = (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z;
    const int recomb = (spec->recomb_model != C21CM_RECOMB_NONE);
    /* the per-radius means matter for the result only when a Lagrangian loop stops above index
     * 0 (box->mean_f_coll = mean of the last radius, IonisationBox.c:1623-1628) */
    const int need_means = (!spec->fix_mean && spec->r_lowest > 0);
    c21cm_ionize_report local;
    memset(&local, 0, sizeof(local));
    /* phase marks on the caller's stream: shard phase | exchange | finish (c21cm_shard_last_phases) */
    for (int i = 0; i < 4; i++)
        if (!g_phase_ev[i]) g_phase_ev[i] = c21hip_event_create();
    g_phase_valid = 0;
#define MARK(i) do { if (g_phase_ev[i]) (void)c21hip_event_record(g_phase_ev[i], stream); } while (0)
    MARK(0);

    g_last_finish_slab = 0;
    if (!recomb && !need_means && slab_finish_wanted(spec)) {
        /* shard phase -> slab exchange of the packed first crossings -> every rank finishes its slab */
        unsigned char *fc = (unsigned char *)c21hip_ws(WS_SHARD_GRID, ntot);
        unsigned *sendb = NULL, *recvb = NULL;
        slab_exchange_arg arg = {broadcast > 0 ? broadcast : 0, spec};
        st = fc ? 0 : C21CM_MEMORY_ALLOC_ERROR;
        if (!st)
            st = c21cm_ionize_shard_radii(spec, rank, world, perturbed_field, previous_ionize_box,
                                          spin_temp, halos, fc, NULL, stream);
        if (!st) st = slab_mask_local(spec, fc, ntot, &sendb, &recvb, stream);
        if ((st = agree_status(st, stream))) return st;
        MARK(1);
        st = slab_mask_exchange(spec, fc, sendb, recvb, stream);
        MARK(2);
        /* (a failed exchange is this rank's local status of the finish: its peers learn of it in the
         * agreement the finish phase enters) */
        if (st) {
            c21cm_shard_slab_state none;
            memset(&none, 0, sizeof(none));
            const int st2 = slab_exchange_rccl(&arg, &none, st, stream);
            return st2 ? st2 : st;
        }
        st = c21cm_ionize_shard_finish_slab(spec, fc, rank, world, perturbed_field, previous_ionize_box,
                                            spin_temp, halos, box, report, slab_exchange_rccl, &arg,
                                            arg.gather_outputs == 1, stream);
        MARK(3);
        g_phase_valid = 1;
        g_last_finish_slab = 1;
        return st;
    }
    if (!recomb) {
        unsigned char *fc = (unsigned char *)c21hip_ws(WS_SHARD_GRID, ntot);
        unsigned *bits = NULL;
        double *d = need_means ? (double *)c21hip_ws(WS_SHARD_SCALARS, sizeof(local.f_coll_grid_mean))
                               : NULL;
        st = (!fc || (need_means && !d)) ? C21CM_MEMORY_ALLOC_ERROR : 0;
        if (!st)
            st = c21cm_ionize_shard_radii(spec, rank, world, perturbed_field, previous_ionize_box,
                                          spin_temp, halos, fc, need_means ? &local : NULL, stream);
        if (!st) st = exchange_mask_local(fc, ntot, owner, &bits, stream);
        if (!st && need_means)
            st = c21hip_h2d(d, local.f_coll_grid_mean, sizeof(local.f_coll_grid_mean), stream);
        if ((st = agree_status(st, stream))) return st;
        MARK(1);
        if ((st = exchange_mask(fc, bits, ntot, owner, stream))) return st;
        MARK(2);
        if (need_means) {
            if ((st = rccl_check(R.reduce(d, d, C21CM_MAX_RADII, RCCL_FLOAT64, RCCL_SUM, owner,
                                          R.comm, stream), "ncclReduce(means)")))
                return st;
            if (rank == owner) {
                if ((st = c21hip_d2h(local.f_coll_grid_mean, d, sizeof(local.f_coll_grid_mean), stream)) ||
                    (st = c21hip_sync(stream)))
                    return st;
                c21cm_ionize_shard_set_means(local.f_coll_grid_mean, spec->n_radii);
            }
        }
        if (rank == owner)
            st = c21cm_ionize_shard_finish(spec, fc, perturbed_field, previous_ionize_box, spin_temp,
                                           halos, box, report, stream);
    } else if (R.ready == 1 && c21cm_ionize_shard_rc_supported(spec) && shard_rc_exchange()) {
        /* the fused recombination loop: first-crossing index + Gamma_12, 5 bytes per cell by slabs
         * (the emulated transport runs its ranks one after the other and keeps the keys) */
        unsigned char *fc = (unsigned char *)c21hip_ws(WS_SHARD_GRID, 5 * ntot);
        float *g12 = (float *)(fc + ntot); /* ntot is a multiple of 16 for every supported box */
        st = fc ? exchange_cross_g12_local(ntot) : C21CM_MEMORY_ALLOC_ERROR;
        if (!st)
            st = c21cm_ionize_shard_radii_rc(spec, rank, world, perturbed_field, previous_ionize_box,
                                             spin_temp, halos, fc, g12, NULL, stream);
        if ((st = agree_status(st, stream))) return st;
        MARK(1);
        if ((st = exchange_cross_g12(fc, g12, ntot, owner, stream))) return st;
        MARK(2);
        if (rank == owner)
            st = c21cm_ionize_shard_finish_rc(spec, fc, g12, perturbed_field, previous_ionize_box,
                                              spin_temp, halos, box, report, stream);
    } else {
        unsigned long long *keys = (unsigned long long *)c21hip_ws(WS_SHARD_GRID, 8 * ntot);
        st = keys ? c21cm_ionize_shard_radii_keys(spec, rank, world, perturbed_field, previous_ionize_box,
                                                  spin_temp, halos, keys, NULL, stream)
                  : C21CM_MEMORY_ALLOC_ERROR;
        if ((st = agree_status(st, stream))) return st;
        MARK(1);
        if ((st = rccl_check(R.reduce(keys, keys, ntot, RCCL_UINT64, RCCL_MAX, owner, R.comm, stream),
                             "ncclReduce(cross_keys)")))
            return st;
        MARK(2);
        if (rank == owner)
            st = c21cm_ionize_shard_finish_keys(spec, keys, perturbed_field, previous_ionize_box,
                                                spin_temp, halos, box, report, stream);
    }
    MARK(3);
    g_phase_valid = 1;
#undef MARK
    if (!broadcast) return st; /* (-1, auto: these models finish on one rank, whose box is broadcast) */
    /* The owner's finish step may have failed: nobody enters the broadcasts then.  Whatever the
     * broadcasts need locally (the staging buffer of host arrays, the scalar slot) is allocated
     * before this agreement; between it and the last broadcast no rank returns early -- a local
     * failure is remembered, every collective is still entered, and a last agreement tells all
     * ranks. */
    const size_t db = ntot * sizeof(float);
    double *dsc = (double *)c21hip_ws(WS_SHARD_SCALARS, sizeof(local.f_coll_grid_mean));
    if (!st && !dsc) st = C21CM_MEMORY_ALLOC_ERROR;
    if (!st && box->neutral_fraction && !c21hip_is_device_ptr(box->neutral_fraction) &&
        !c21hip_ws(WS_SHARD_STAGE, db))
        st = C21CM_MEMORY_ALLOC_ERROR;
    if ((st = agree_status(st, stream))) return st;

    /* every rank returns the finished box */
    int err = 0;
#define BCAST(p, bytes) do { const int s_ = bcast_array(p, bytes, owner, stream); if (s_ && !err) err = s_; } while (0)
    BCAST(box->neutral_fraction, db);
    BCAST(box->z_reion, db);
    BCAST(box->kinetic_temperature, db);
    if (spec->fcoll_mode != C21CM_FCOLL_STARS_GRID) BCAST(box->unnormalised_nion, db);
    if (recomb) {
        BCAST(box->ionisation_rate_G12, db);
        BCAST(box->mean_free_path, db);
        BCAST(box->cumulative_recombinations,
              spec->recomb_model == C21CM_RECOMB_INHOMOGENEOUS ? db : sizeof(float));
    }
#undef BCAST
    {
        double sc[4] = {box->mean_f_coll, box->mean_f_coll_MINI, report ? report->global_xH : 0.,
                        report ? report->mean_f_coll_out : 0.};
        int s_ = 0;
        if (rank == owner && (s_ = c21hip_h2d(dsc, sc, sizeof(sc), stream)) && !err) err = s_;
        if ((s_ = rccl_check(R.broadcast(dsc, dsc, sizeof(sc), RCCL_UINT8, owner, R.comm, stream),
                             "ncclBroadcast(scalars)")) && !err)
            err = s_;
        if (((s_ = c21hip_d2h(sc, dsc, sizeof(sc), stream)) || (s_ = c21hip_sync(stream))) && !err) err = s_;
        if ((st = agree_status(err, stream))) return st;
        box->mean_f_coll = sc[0];
        box->mean_f_coll_MINI = sc[1];
        if (report && rank != owner) {
            report->global_xH = sc[2];
            report->mean_f_coll_out = sc[3];
        }
    }
    return 0;
}

/* ---- ComputeTsBox sharded over the ranks (abi_compute.c: ts_box_run) ---------------------------
 * phase 1 on every rank (its shells' partial sums) -> reduce-scatter by cell slabs, point to point:
 * rank r sends peer p the rows of p's slab (one message per peer over their direct xGMI link, all
 * seven at once) and receives its own slab's rows from everybody -> the complete sums of the slab,
 * ranks added in rank order -> temperature update of the slab -> all-gather of the three output
 * boxes (again one message per peer and box).  Volumes at 512^3, 8 ranks, 4 rows: 537 MB per link
 * for the sums in double (C21CM_TS_SHARD_EXCHANGE=f32: partials as floats scaled by a per-row power of two
 * -- the rows are rates of 1e27 .. 1e49 --, 268 MB), 3 x 67 MB per link for the outputs. */
enum { WS_TSS_SUMS = 242, WS_TSS_SEND = 243, WS_TSS_RECV = 244, WS_TSS_SLAB = 245, WS_TSS_ROWMAX = 260 };

/* how many times this process took the sharded ComputeTsBox (tests: did the call shard?) */
static int g_ts_sharded_calls;
int c21cm_ts_box_sharded_calls(void) { return g_ts_sharded_calls; }

int c21cm_ts_box_sharded(float redshift, float prev_redshift, float perturbed_field_redshift,
                         PerturbedField *perturbed_field, TsBox *previous_spin_temp,
                         TsBox *this_spin_temp) {
    g_ts_sharded_calls++;
    if (!R.ready || R.ready == 2) {
        c21hip_set_error("shard: ComputeTsBox shards over an RCCL communicator (c21cm_shard_init)");
        return C21CM_VALUE_ERROR;
    }
    const int rank = R.rank, world = R.world;
    const size_t ntot = (size_t)simulation_options_global->HII_DIM * simulation_options_global->HII_DIM *
                        (size_t)(simulation_options_global->NON_CUBIC_FACTOR * simulation_options_global->HII_DIM);
    int st = 0, rows = 0;
    const char *e = getenv("C21CM_TS_SHARD_EXCHANGE");
    const int f32 = (e && e[0] == 'f' && e[1] == '3') ? 1 : 0;
    const size_t esz = f32 ? sizeof(float) : sizeof(double);
    size_t maxlen = 0;
    for (int r = 0; r < world; r++) {
        const size_t len = c21hip_ts_slab_begin(ntot, world, r + 1) - c21hip_ts_slab_begin(ntot, world, r);
        if (len > maxlen) maxlen = len;
    }
    const size_t c0 = c21hip_ts_slab_begin(ntot, world, rank);
    const size_t len = c21hip_ts_slab_begin(ntot, world, rank + 1) - c0;
    /* Every buffer of the call is allocated, and the pack launched, BEFORE the ranks agree on a
     * status: a rank that fails locally still joins the agreement and nobody is left in ncclRecv
     * (ADVICE r3).  rows = 4, or 6 with USE_LYA_HEATING (abi_compute.c: ts_box_run). */
    const int rows_max = (astro_options_global && astro_options_global->USE_LYA_HEATING) ? 6 : 4;
    const size_t per_peer_max = c21hip_ts_slot_elems(rows_max, maxlen, f32) * esz;
    void *rowmax = c21hip_ws(WS_TSS_ROWMAX, 64);
    double *sums = (double *)c21hip_ws(WS_TSS_SUMS, 6 * ntot * sizeof(double));
    char *sendb = NULL, *recvb = NULL;
    if (world > 1) {
        sendb = (char *)c21hip_ws(WS_TSS_SEND, per_peer_max * (size_t)(world - 1));
        recvb = (char *)c21hip_ws(WS_TSS_RECV, per_peer_max * (size_t)(world - 1));
    }
    double *slab = (double *)c21hip_ws(WS_TSS_SLAB, (size_t)rows_max * (len ? len : 1) * sizeof(double));
    if (!sums || !slab || !rowmax || (world > 1 && (!sendb || !recvb))) st = C21CM_MEMORY_ALLOC_ERROR;
    if (!st)
        st = c21cm_ts_box_shard_sums(redshift, prev_redshift, perturbed_field_redshift, perturbed_field,
                                     previous_spin_temp, rank, world, sums, &rows);
    if (!st && rows > rows_max) {
        c21hip_set_error("shard: ComputeTsBox returned %d rows of sums, %d expected", rows, rows_max);
        st = C21CM_VALUE_ERROR;
    }
    const size_t per_peer = c21hip_ts_slot_elems(rows, maxlen, f32) * esz;
    if (!st) st = c21hip_ts_pack_slabs(sums, ntot, world, rank, rows, maxlen, f32, rowmax, sendb, NULL);
    if ((st = agree_status(st, NULL))) return st;
    if (world > 1) {
        if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
        for (int p = 0; p < world - 1 && !st; p++) {
            const int peer = p < rank ? p : p + 1;
            st = rccl_check(R.send(sendb + (size_t)p * per_peer, per_peer, RCCL_UINT8, peer, R.comm, NULL),
                            "ncclSend(shell sums)");
            if (!st)
                st = rccl_check(R.recv(recvb + (size_t)p * per_peer, per_peer, RCCL_UINT8, peer, R.comm, NULL),
                                "ncclRecv(shell sums)");
        }
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    /* combine + temperature update are local: their status goes into the second agreement */
    st = c21hip_ts_combine_slab(sums, ntot, world, rank, rows, maxlen, f32, rowmax, recvb, slab, NULL);
    if (!st)
        st = c21cm_ts_box_shard_finish(redshift, prev_redshift, perturbed_field_redshift, perturbed_field,
                                       previous_spin_temp, slab, c0, len, this_spin_temp);
    if ((st = agree_status(st, NULL))) return st;
    if (world > 1) { /* all-gather: every rank ends with the full boxes */
        float *boxes[3] = {this_spin_temp->spin_temperature, this_spin_temp->kinetic_temp_neutral,
                           this_spin_temp->xray_ionised_fraction};
        if ((st = rccl_check(R.group_start(), "ncclGroupStart"))) return st;
        for (int b = 0; b < 3 && !st; b++)
            for (int peer = 0; peer < world && !st; peer++) {
                if (peer == rank) continue;
                const size_t p0 = c21hip_ts_slab_begin(ntot, world, peer);
                const size_t plen = c21hip_ts_slab_begin(ntot, world, peer + 1) - p0;
                st = rccl_check(R.send(boxes[b] + c0, len * sizeof(float), RCCL_UINT8, peer, R.comm, NULL),
                                "ncclSend(Ts slab)");
                if (!st)
                    st = rccl_check(R.recv(boxes[b] + p0, plen * sizeof(float), RCCL_UINT8, peer, R.comm, NULL),
                                    "ncclRecv(Ts slab)");
            }
        const int st2 = rccl_check(R.group_end(), "ncclGroupEnd");
        if (st || st2) return st ? st : st2;
    }
    return c21hip_sync(NULL);
}

size_t c21cm_ts_slab_begin(size_t ntot, int world, int r) { return c21hip_ts_slab_begin(ntot, world, r); }
