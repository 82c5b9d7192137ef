// runtime.hip -- device plumbing behind c21hip.h: pointer residency, cached HBM
// scratch ("workspace slots"), copies, events, last-error string.
//
// The reference allocates and frees its FFTW grids inside every Compute* call
// (src/py21cmfast/src/IonisationBox.c:229-319).  On MI355X a hipMalloc/hipFree
// pair of several GB costs milliseconds and 288 GB of HBM is plentiful, so
// scratch is kept in numbered slots that only ever grow and are reused by the
// next call; c21cm_release_device_cache() returns them.
#include <hip/hip_runtime.h>

#include <dirent.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "c21hip.h"
#include "c21cm_abi.h"

namespace {
constexpr int kMaxSlots = 288;  // ids 256.. : shard_rccl.c (status word, slab exchange)
struct Slot {
    void *ptr = nullptr;
    size_t bytes = 0;
    // scattered placement (C21CM_WS_ALLOC=scatter): a reserved address range backed by 2 MB chunks mapped in a
    // shuffled order
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    size_t va_bytes = 0;
};
Slot g_slots[kMaxSlots];
std::mutex g_mutex;

void slot_free(Slot &s) {
    if (!s.ptr) return;
    if (s.va_bytes) {
        (void)hipMemUnmap(s.ptr, s.va_bytes);
        for (auto h : s.chunks) (void)hipMemRelease(h);
        (void)hipMemAddressFree(s.ptr, s.va_bytes);
        s.chunks.clear();
        s.va_bytes = 0;
    } else {
        (void)hipFree(s.ptr);
    }
    s.ptr = nullptr;
    s.bytes = 0;
}

// Large workspace buffers whose 2 MB pages are NOT in physical order.  On a GPU whose memory is unfragmented
// hipMalloc hands out physically contiguous ranges, and the line passes that walk rows at a power-of-two pitch
// (pass Y: 128-byte pieces every 2 / 4 KB) then run 8-20 % slower than on a box whose memory has been churned
// (profiles/r05_placement_study.txt): the regular address-to-channel map serves such a walk from a subset of the
// HBM channels at a time.  Chunks of the allocation granularity are created in order and mapped at a
// pseudo-randomly permuted position of a reserved address range.
void *scatter_alloc(Slot &s, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
        (void)hipGetLastError();
        return nullptr;
    }
    size_t want = (size_t)2 << 20;  // C21CM_WS_SCATTER_MB: chunk size of the experiment (default 2 MB)
    if (const char *e = getenv("C21CM_WS_SCATTER_MB")) want = (size_t)atol(e) << 20;
    const size_t chunk = (want + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::vector<hipMemGenericAllocationHandle_t> hs(n);
    size_t made = 0;
    bool ok = true;
    for (; made < n; made++)
        if (hipMemCreate(&hs[made], chunk, &prop, 0) != hipSuccess) {
            ok = false;
            break;
        }
    // position of chunk i: a fixed odd multiplier modulo a power of two >= n, cycled into range (a permutation)
    size_t m = 1;
    while (m < n) m <<= 1;
    size_t mapped = 0;
    if (ok) {
        size_t pos = 0;
        for (size_t i = 0; i < m && ok; i++) {
            const size_t j = (i * 0x9E3779B1ull + 12345u) & (m - 1);  // odd multiplier: a bijection on [0, m)
            if (j >= n) continue;
            if (hipMemMap((char *)va + j * chunk, chunk, 0, hs[pos++], 0) != hipSuccess) ok = false;
            else mapped++;
        }
        ok = ok && mapped == n;
    }
    if (ok) {
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        if (mapped) (void)hipMemUnmap(va, total);
        for (size_t i = 0; i < made; i++) (void)hipMemRelease(hs[i]);
        (void)hipMemAddressFree(va, total);
        return nullptr;
    }
    s.chunks.swap(hs);
    s.va_bytes = total;
    return va;
}
thread_local char g_error[512] = "";
}  // namespace

extern "C" void c21hip_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    if (getenv("C21CM_VERBOSE")) fprintf(stderr, "[21cmfast_hip] %s\n", g_error);
}

extern "C" const char *c21hip_get_error(void) { return g_error; }

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            c21hip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                   \
            return (e_ == hipErrorOutOfMemory) ? C21CM_MEMORY_ALLOC_ERROR : C21CM_IO_ERROR; \
        }                                                                                 \
    } while (0)

// The current device is a per-thread setting: a helper thread must adopt its creator's.
extern "C" int c21hip_current_device(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}

extern "C" int c21hip_use_device(int device) {
    if (device < 0) return 0;
    if (hipSetDevice(device) != hipSuccess) {
        c21hip_set_error("hipSetDevice(%d) failed", device);
        (void)hipGetLastError();
        return C21CM_IO_ERROR;
    }
    return 0;
}

extern "C" int c21hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int c21hip_is_device_ptr(const void *p) {
    if (!p) return 0;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd host memory: not an error for us
        return 0;
    }
    return attr.type == hipMemoryTypeDevice ? 1 : 0;
}

extern "C" void *c21hip_ws(int slot, size_t bytes) {
    if (slot < 0 || slot >= kMaxSlots) return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    Slot &s = g_slots[slot];
    if (s.bytes >= bytes && s.ptr) return s.ptr;
    slot_free(s);
    void *p = nullptr;
    // C21CM_WS_ALLOC=contig: large buffers physically contiguous (hipDeviceMallocContiguous) -- placement study
    // of the two speeds of the line passes (DESIGN Appendix B.0); anything else: plain hipMalloc
    static int contig = -1;
    if (contig < 0) {
        const char *e = getenv("C21CM_WS_ALLOC");
        contig = (e && e[0] == 'c') ? 1 : ((e && e[0] == 's') ? 2 : 0);
    }
    hipError_t e = hipErrorUnknown;
    if (contig == 2 && bytes >= ((size_t)64 << 20)) {
        p = scatter_alloc(s, bytes);
        if (p) e = hipSuccess;
    }
    if (contig == 1 && bytes >= ((size_t)64 << 20)) {
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
        }
    }
    if (e != hipSuccess) e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c21hip_set_error("hipMalloc of %zu bytes for workspace slot %d failed: %s", bytes, slot,
                         hipGetErrorString(e));
        return nullptr;
    }
    s.ptr = p;
    s.bytes = bytes;
    {
        static int trace = -1;  // C21CM_WS_TRACE=1: where the workspace buffers land (placement studies)
        if (trace < 0) trace = getenv("C21CM_WS_TRACE") ? 1 : 0;
        if (trace) fprintf(stderr, "[c21hip_ws] slot %d bytes %zu ptr %p\n", slot, bytes, p);
    }
    return p;
}

// What a slot holds now (nullptr / 0: nothing), and a buffer allocated elsewhere (plain hipMalloc) handed to a
// slot that is empty or smaller: the workspace owns it from then on (placement shopping, ionize_driver.c).
extern "C" void *c21hip_ws_peek(int slot, size_t *bytes) {
    if (slot < 0 || slot >= kMaxSlots) return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (bytes) *bytes = g_slots[slot].bytes;
    return g_slots[slot].ptr;
}
extern "C" int c21hip_ws_adopt(int slot, void *ptr, size_t bytes) {
    if (slot < 0 || slot >= kMaxSlots || !ptr) return C21CM_VALUE_ERROR;
    std::lock_guard<std::mutex> lock(g_mutex);
    slot_free(g_slots[slot]);
    g_slots[slot].ptr = ptr;
    g_slots[slot].bytes = bytes;
    return 0;
}
// ---- physical memory without (much of) a mapping: the placement walk's chunks (csrc/host/placement.c) ----------
// hipMalloc of 16 GB costs ~0.6 s on this driver (38-45 ms per GB, profiles/r06_placement_vmm.txt section 1) and the walk holds
// up to eight of them; hipMemCreate hands out the physical range for nothing, and only the head the probe
// launches touch is mapped.  A chunk is {handle, reserved range, mapped bytes}; the workspace can adopt one whose
// physical size equals its mapping (slot_free knows how to undo it).
struct VmmChunk {
    hipMemGenericAllocationHandle_t handle;
    void *va;
    size_t map_bytes, phys_bytes;
};
extern "C" void *c21hip_vmm_chunk(size_t phys_bytes, size_t map_bytes, void **va_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
    phys_bytes = (phys_bytes + gran - 1) / gran * gran;
    map_bytes = (map_bytes + gran - 1) / gran * gran;
    if (map_bytes > phys_bytes) map_bytes = phys_bytes;
    VmmChunk *c = new VmmChunk{};
    c->phys_bytes = phys_bytes;
    c->map_bytes = map_bytes;
    if (hipMemCreate(&c->handle, phys_bytes, &prop, 0) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return nullptr;
    }
    bool ok = hipMemAddressReserve(&c->va, map_bytes, gran, nullptr, 0) == hipSuccess;
    bool mapped = false;
    if (ok) {
        mapped = hipMemMap(c->va, map_bytes, 0, c->handle, 0) == hipSuccess;
        ok = mapped;
    }
    if (ok) {
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(c->va, map_bytes, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        if (mapped) (void)hipMemUnmap(c->va, map_bytes);
        if (c->va) (void)hipMemAddressFree(c->va, map_bytes);
        (void)hipMemRelease(c->handle);
        delete c;
        return nullptr;
    }
    if (va_out) *va_out = c->va;
    return c;
}
extern "C" void c21hip_vmm_chunk_free(void *chunk) {
    VmmChunk *c = (VmmChunk *)chunk;
    if (!c) return;
    (void)hipMemUnmap(c->va, c->map_bytes);
    (void)hipMemAddressFree(c->va, c->map_bytes);
    (void)hipMemRelease(c->handle);
    delete c;
}
// the slot takes the chunk over (its mapping must cover the physical range: nothing is held that is not used)
extern "C" int c21hip_ws_adopt_vmm(int slot, void *chunk, size_t bytes) {
    VmmChunk *c = (VmmChunk *)chunk;
    if (slot < 0 || slot >= kMaxSlots || !c || c->map_bytes != c->phys_bytes || c->map_bytes < bytes)
        return C21CM_VALUE_ERROR;
    std::lock_guard<std::mutex> lock(g_mutex);
    slot_free(g_slots[slot]);
    g_slots[slot].ptr = c->va;
    g_slots[slot].bytes = bytes;
    g_slots[slot].chunks.assign(1, c->handle);
    g_slots[slot].va_bytes = c->map_bytes;
    delete c;
    return 0;
}

extern "C" void *c21hip_raw_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" void c21hip_raw_free(void *p) {
    if (p) (void)hipFree(p);
}
extern "C" size_t c21hip_free_bytes(void) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return fr;
}

extern "C" size_t c21hip_total_bytes(void) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return tot;
}

// How many processes hold memory on the current device (this one included), from the KFD's per-process
// accounting: /sys/class/kfd/kfd/proc/<pid>/vram_<gpu_id>, the gpu_id found by matching the device's PCI
// location against /sys/class/kfd/kfd/topology/nodes/*/properties.  -1: unknown (no sysfs, container without
// it, ...).  The placement walk (csrc/host/placement.c) holds a large part of the free memory for some
// milliseconds: it only does so when this says the process has the device to itself.
extern "C" int c21hip_device_tenants(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    const long want_loc = ((long)prop.pciBusID << 8) | ((long)prop.pciDeviceID << 3);
    const long want_dom = prop.pciDomainID;
    long gpu_id = -1;
    int n_gpu_nodes = 0;
    long only_gpu_id = -1;
    for (int node = 0; node < 64 && gpu_id < 0; node++) {
        char path[160];
        snprintf(path, sizeof(path), "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", node);
        FILE *f = fopen(path, "r");
        if (!f) break;
        long id = 0;
        const int ok = fscanf(f, "%ld", &id) == 1;
        fclose(f);
        if (!ok || id == 0) continue;  // (CPU nodes, and GPUs this container may not see, read 0)
        n_gpu_nodes++;
        only_gpu_id = id;
        snprintf(path, sizeof(path), "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
        f = fopen(path, "r");
        if (!f) continue;
        char key[64];
        long val = 0, loc = -1, dom = 0;
        while (fscanf(f, "%63s %ld", key, &val) == 2) {
            if (!strcmp(key, "location_id")) loc = val;
            if (!strcmp(key, "domain")) dom = val;
        }
        fclose(f);
        if ((loc & ~7L) == want_loc && dom == want_dom) gpu_id = id;
    }
    if (gpu_id < 0 && n_gpu_nodes == 1) gpu_id = only_gpu_id;  // one visible GPU: it is this one
    if (gpu_id < 0) return -1;
    DIR *d = opendir("/sys/class/kfd/kfd/proc");
    if (!d) return -1;
    int tenants = 0;
    struct dirent *e;
    while ((e = readdir(d)) != NULL) {
        if (e->d_name[0] < '0' || e->d_name[0] > '9') continue;
        char path[200];
        snprintf(path, sizeof(path), "/sys/class/kfd/kfd/proc/%s/vram_%ld", e->d_name, gpu_id);
        FILE *f = fopen(path, "r");
        if (!f) continue;
        unsigned long long bytes = 0;
        if (fscanf(f, "%llu", &bytes) == 1 && bytes > (64ull << 20)) tenants++;  // (a bare context holds a few MB)
        fclose(f);
    }
    closedir(d);
    return tenants;
}

static unsigned long g_ws_generation = 0;
// incremented by every release of the workspace: owners of sticky state inside a slot compare it
extern "C" unsigned long c21hip_ws_generation(void) { return g_ws_generation; }

extern "C" void c21hip_ws_release(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_ws_generation++;
    for (auto &s : g_slots) {
        slot_free(s);
    }
}

extern "C" int c21hip_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int c21hip_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
extern "C" int c21hip_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int c21hip_memset(void *dst, int byte, size_t bytes, void *stream) {
    HIP_TRY(hipMemsetAsync(dst, byte, bytes, (hipStream_t)stream));
    return 0;
}
extern "C" int c21hip_sync(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
extern "C" int c21hip_device_sync(void) {
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" void *c21hip_event_create(void) {
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return (void *)ev;
}
extern "C" void c21hip_event_destroy(void *ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
extern "C" int c21hip_event_record(void *ev, void *stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
// A small library-owned pinned host buffer (grown on demand, kept until the device cache is
// released): asynchronous copies to or from pageable memory may block the calling thread on
// the stream, which defeats host/device overlap.
extern "C" void *c21hip_pinned_host(size_t bytes) {
    static void *buf = nullptr;
    static size_t cap = 0;
    if (bytes > cap) {
        if (buf) (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
        if (hipHostMalloc(&buf, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        cap = bytes;
    }
    return buf;
}
// pinned host blocks and extra streams for the helper threads of the host drivers (the IC random
// stream stages its words chunk by chunk while it is still being drawn)
extern "C" void *c21hip_pinned_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" void c21hip_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}
extern "C" void *c21hip_stream_create(void) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
extern "C" void c21hip_stream_destroy(void *s) {
    if (s) (void)hipStreamDestroy((hipStream_t)s);
}
extern "C" int c21hip_event_synchronize(void *ev) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)ev));
    return 0;
}
extern "C" int c21hip_stream_wait_event(void *stream, void *ev) {
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}
// A library-owned non-blocking side stream (created once) for work that depends on no grid
// data of the caller's stream, e.g. the window tables of the next filter radius.
extern "C" void *c21hip_aux_stream(void) {
    static hipStream_t aux = nullptr;
    if (!aux) {
        // C21CM_AUX_PRIO=low: lowest stream priority, so that what runs here (window tables of
        // the coming radii) only fills in behind the caller's kernels.  Measured no difference at
        // 512^3, so the default stays a plain non-blocking stream.
        int least = 0, greatest = 0;
        const char *e = getenv("C21CM_AUX_PRIO");
        const bool low = e && e[0] == 'l' &&
                         hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        const hipError_t st = low ? hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, least)
                                  : hipStreamCreateWithFlags(&aux, hipStreamNonBlocking);
        if (st != hipSuccess) aux = nullptr;
    }
    return aux;
}
extern "C" float c21hip_event_elapsed_ms(void *start, void *stop) {
    float ms = -1.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}
