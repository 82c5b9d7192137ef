// Diagnostic (GPU box only): can pass Y hand an x-plane of both work spectra to the fused pass Z
// through the XCD's L2 instead of HBM?  
//
// Geometry of the 512^3 R loop: an x-plane of one split spectrum is 512 rows (y) x 256 float2 (k_z)
// = 1 MB, 2.1 MB for the two grids.  Pass Y works on tiles "all 512 rows x 16 k_z columns" (a
// 128-byte segment per row, rows 2 KB apart), 32 tiles per plane of both grids; pass Z works on whole
// rows, 2 x 16 rows per workgroup.  The probe moves exactly these bytes with trivial arithmetic:
//
//   two_kernels   "Y": tile in from `src`, + 1, tile out to a full-size work buffer; "Z": rows of the
//                 work buffer in, a checksum out.  What the R loop does today (through HBM).
//   ring          ONE persistent kernel, 256 workgroups of 512 threads.  The workgroups of an XCD
//                 (blockIdx % 8, checked against HW_REG_XCC_ID) share a 2.1 MB ring slot: per plane
//                 every one writes its tile into the slot with PLAIN stores (they stay in this XCD's
//                 L2), drains (vmcnt(0)), arrives on a per-XCD counter; waits for the 32 arrivals;
//                 reads its 2 x 16 rows with sc1 loads (served by the L2, never by the stale per-CU
//                 L1); arrives on a second counter that the next plane's tile stores wait for.
//                 No agent-scope release: nothing has to leave the XCD.  If the placement assumption
//                 fails the data is stale, the checksum shows it, and the probe says so.
//   ring_full     the same kernel with the "slot" = the plane's own place in a full-size buffer
//                 (no address reuse): reads can hit the L2, the writes go to HBM eventually.
//
// Reported: time per launch, us per plane and XCD, checksum agreement with two_kernels.  HBM traffic
// of each variant: run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/l2_probe.sh).
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_plane_probe.hip -o tools/bin/l2_plane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); fflush(stdout); return 1; } } while (0)

constexpr int NX = 512, NY = 512, ROW_F4 = 128;  // a row = 256 float2 = 128 float4 = 2 KB
constexpr int NG = 2;                             // grids
constexpr int TILE_F4 = 8;                        // 16 float2 columns = 8 float4 = 128 B
constexpr int TILES = ROW_F4 / TILE_F4;           // 16 tiles per grid and plane
constexpr int WG = 512;
constexpr int GROUP = 32;                         // workgroups per XCD
constexpr size_t PLANE_F4 = (size_t)NY * ROW_F4;  // float4 per plane and grid

typedef unsigned int __attribute__((address_space(1))) gu32;

// Loads with a cache policy the compiler can see (and count in vmcnt): raw buffer loads through a
// resource descriptor (wave-uniform base, 32-bit per-lane byte offsets); aux 16 = sc1 (bypasses the
// per-CU L1, served by the XCD's L2), aux 2 = nt.  (Inline-asm `global_load ... sc1` hides from the
// compiler WHEN the value arrives: it gave wrong sums in one instantiation and a fault in another.)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float4 ld_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ unsigned long long mix(float4 v) {
    return (unsigned long long)__float_as_uint(v.x) + 3ull * __float_as_uint(v.y) +
           5ull * __float_as_uint(v.z) + 7ull * __float_as_uint(v.w);
}

// tile task k of a plane: grid k / 16, tile k % 16.  A thread owns 8 (row, float4) slots: row =
// r * 64 + tid / 8, float4 = tid % 8 -> every 8 lanes read one 128-byte segment.
template <bool NT_IN>
__device__ __forceinline__ void tile_in(const float4 *__restrict__ src, int x, int k, float4 (&v)[8]) {
    const int g = k / TILES, t = k % TILES;
    const size_t off = ((size_t)g * NX + x) * PLANE_F4 + (size_t)t * TILE_F4;
    const float4 *base = src + off;
    const int row0 = threadIdx.x >> 3, c = threadIdx.x & 7;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, 0xfffffff0u);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const size_t e = (size_t)(r * 64 + row0) * ROW_F4 + c;
        v[r] = NT_IN ? ld_buf<2>(rs, (unsigned)((off + e) * 16)) : base[e];
    }
}
__device__ __forceinline__ void tile_out(float4 *__restrict__ dst_plane0, int k, const float4 (&v)[8]) {
    // dst_plane0: start of the plane of grid 0 in the destination; grid 1 follows at +slot_grid_stride
    const int t = k % TILES;
    float4 *base = dst_plane0 + (size_t)t * TILE_F4;
    const int row0 = threadIdx.x >> 3, c = threadIdx.x & 7;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        float4 o = v[r];
        o.x += 1.f;
        base[(size_t)(r * 64 + row0) * ROW_F4 + c] = o;
    }
}
// rows task k of a plane: rows [16 k, 16 k + 16) of BOTH grids: 2 x 32 KB contiguous.
template <bool SC1>
__device__ __forceinline__ unsigned long long rows_in(const float4 *plane_g0, const float4 *plane_g1, int k) {
    unsigned long long acc = 0;
    float4 v[8];
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(plane_g0, (unsigned)(PLANE_F4 * 16));
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(plane_g1, (unsigned)(PLANE_F4 * 16));
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t e = (size_t)k * 16 * ROW_F4 + (size_t)(i & 3) * WG + threadIdx.x;
        v[i] = SC1 ? ld_buf<16>(i < 4 ? r0 : r1, (unsigned)(e * 16)) : (i < 4 ? plane_g0 : plane_g1)[e];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc += mix(v[i]) * (unsigned long long)(i + 1);
    return acc;
}
__device__ __forceinline__ void block_sum_store(unsigned long long acc, unsigned long long *out) {
    __shared__ unsigned long long part[WG / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < WG / 64; w++) s += part[w];
        *out = s;
    }
    __syncthreads();
}

// ---- today's structure: two kernels through a full-size work buffer
__global__ void __launch_bounds__(WG) y_kernel(const float4 *__restrict__ src, float4 *__restrict__ work) {
    for (int item = blockIdx.x; item < NX * 2 * TILES; item += gridDim.x) {
        const int x = item / (2 * TILES), k = item % (2 * TILES);
        float4 v[8];
        tile_in<false>(src, x, k, v);
        tile_out(work + ((size_t)(k / TILES) * NX + x) * PLANE_F4, k, v);
    }
}
__global__ void __launch_bounds__(WG) z_kernel(const float4 *__restrict__ work, unsigned long long *__restrict__ sums) {
    for (int item = blockIdx.x; item < NX * GROUP; item += gridDim.x) {
        const int x = item / GROUP, k = item % GROUP;
        const unsigned long long a = rows_in<false>(work + (size_t)x * PLANE_F4, work + ((size_t)NX + x) * PLANE_F4, k);
        block_sum_store(a, sums + item);
    }
}

// ---- one persistent kernel, per-XCD hand-off
struct Sync {
    unsigned y_done[8][32];  // one counter per XCD, each on its own 128-byte line
    unsigned z_done[8][32];
    unsigned xcc_mismatch, timeout;
};
// LOCAL: no atomics and no trip to the memory side.  The 32 workgroups of an XCD own one word each
// of a 128-byte line; arriving = ONE plain store of the epoch into the own word (it stays in the
// XCD's L2, like the payload), waiting = lanes 0..31 of the first wave read the 32 words with sc1
// loads (L1 bypassed, served by that L2) until every word has reached the epoch.  Coherent WITHIN an
// XCD only -- the same placement assumption as the data path.
template <bool LOCAL>
__device__ __forceinline__ bool wait_ge(unsigned *cnt, unsigned target, unsigned *timeout_flag) {
    if (LOCAL) {
        if (threadIdx.x < 64) {
            const unsigned epoch = target / GROUP;
            unsigned spins = 0;
            const int lane = threadIdx.x & 31;
            for (;;) {
                const unsigned v = __hip_atomic_load((gu32 *)(cnt + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= epoch)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 15)) {
                    if (threadIdx.x == 0)
                        __hip_atomic_store((gu32 *)timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        return true;
    }
    // ONE lane polls, then the workgroup barrier releases the rest
    if (threadIdx.x == 0 &&
        !__hip_atomic_load((gu32 *)timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        unsigned spins = 0;
        while (__hip_atomic_load((gu32 *)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 15)) {
                __hip_atomic_store((gu32 *)timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return true;
}
template <bool LOCAL>
__device__ __forceinline__ void arrive(unsigned *cnt, int k, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores / loads have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        if (LOCAL) {
            *(volatile unsigned *)(cnt + k) = epoch;  // plain store: stays in this XCD's L2
        } else {
            __hip_atomic_fetch_add((gu32 *)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <bool FULL, bool NT_IN, bool PREFETCH, bool LOCAL = false>
__global__ void __launch_bounds__(WG) ring_kernel(const float4 *__restrict__ src, float4 *__restrict__ ring,
                                                  unsigned long long *__restrict__ sums, Sync *sync) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;  // k: this workgroup's task slot in its XCD
    {
        // XCC_ID: s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits [3:0]
        const unsigned id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
        if (threadIdx.x == 0 && (int)id != xcd) atomicAdd(&sync->xcc_mismatch, 1u);
    }
    unsigned *ycnt = &sync->y_done[xcd][0], *zcnt = &sync->z_done[xcd][0];
    // the slot of this XCD: [grid][plane]; FULL: the plane's own place in a full-size buffer
    float4 v[8];
    int n_planes = NX / 8;
    if (PREFETCH) tile_in<NT_IN>(src, xcd, k, v);
    for (int i = 0; i < n_planes; i++) {
        const int x = xcd + 8 * i;
        float4 *slot0 = FULL ? ring + (size_t)x * PLANE_F4 : ring + (size_t)xcd * 2 * PLANE_F4;
        float4 *slot1 = FULL ? ring + ((size_t)NX + x) * PLANE_F4 : slot0 + PLANE_F4;
        if (!PREFETCH) tile_in<NT_IN>(src, x, k, v);
        // the slot may be overwritten once every workgroup of the XCD has read plane i - 1
        if (!FULL && i > 0) wait_ge<LOCAL>(zcnt, (unsigned)GROUP * i, &sync->timeout);
        tile_out(k < TILES ? slot0 : slot1, k, v);
        arrive<LOCAL>(ycnt, k, (unsigned)(i + 1));
        if (PREFETCH && i + 1 < n_planes) tile_in<NT_IN>(src, x + 8, k, v);  // next plane's loads fly during the wait
        wait_ge<LOCAL>(ycnt, (unsigned)GROUP * (i + 1), &sync->timeout);
        const unsigned long long a = rows_in<true>(slot0, slot1, k);
        if (!FULL) arrive<LOCAL>(zcnt, k, (unsigned)(i + 1));
        block_sum_store(a, sums + (size_t)x * GROUP + k);
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const int mask = argc > 2 ? atoi(argv[2]) : 127;  // which fused variants run
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t grid_f4 = (size_t)NX * PLANE_F4;
    float4 *src, *work, *ring;
    unsigned long long *sums_a, *sums_b;
    Sync *sync;
    CK(hipMalloc(&src, NG * grid_f4 * sizeof(float4)));
    CK(hipMalloc(&work, NG * grid_f4 * sizeof(float4)));
    CK(hipMalloc(&ring, 8 * 2 * PLANE_F4 * sizeof(float4)));
    CK(hipMalloc(&sums_a, (size_t)NX * GROUP * 8));
    CK(hipMalloc(&sums_b, (size_t)NX * GROUP * 8));
    CK(hipMalloc(&sync, sizeof(Sync)));
    {
        std::vector<float> h(NG * grid_f4 * 4);
        unsigned s = 12345u;
        for (auto &f : h) {
            s = s * 1664525u + 1013904223u;
            f = (float)(s >> 8) * (1.0f / 16777216.0f);
        }
        CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t nsum = (size_t)NX * GROUP;
    std::vector<unsigned long long> ha(nsum), hb(nsum);
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };

    // reference
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(y_kernel, dim3(256), dim3(WG), 0, 0, src, work);
        hipLaunchKernelGGL(z_kernel, dim3(256), dim3(WG), 0, 0, work, sums_a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms);
    }
    CK(hipMemcpy(ha.data(), sums_a, nsum * 8, hipMemcpyDeviceToHost));
    const float t_ref = med(ts);
    printf("%-34s %8.1f us per launch pair  (%.2f us per plane and XCD)  alg bytes 2 x (2.1 + 2.1) GB -> %.2f TB/s\n",
           "two_kernels (through HBM)", t_ref * 1e3, t_ref * 1e3 / 64, 3.0 * NG * grid_f4 * 16 / t_ref / 1e9);

    auto run = [&](const char *name, auto kernel, float4 *buf) -> int {
        std::vector<float> tt;
        for (int r = 0; r < reps; r++) {
            CK(hipMemsetAsync(sync, 0, sizeof(Sync), 0));
            CK(hipMemsetAsync(sums_b, 0, nsum * 8, 0));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(256), dim3(WG), 0, 0, (const float4 *)src, buf, sums_b, sync);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            tt.push_back(ms);
        }
        CK(hipMemcpy(hb.data(), sums_b, nsum * 8, hipMemcpyDeviceToHost));
        Sync hs;
        CK(hipMemcpy(&hs, sync, sizeof(Sync), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < nsum; i++) bad += ha[i] != hb[i];
        const float t = med(tt);
        printf("%-34s %8.1f us per launch       (%.2f us per plane and XCD)  x%.2f vs two kernels; wrong checksums %zu / %zu, XCC mismatch %u, timeout %u\n",
               name, t * 1e3, t * 1e3 / 64, t_ref / t, bad, nsum, hs.xcc_mismatch, hs.timeout);
        return 0;
    };
    if ((mask & 1) && run("ring (2.1 MB slot per XCD)", ring_kernel<false, false, false>, ring)) return 1;
    if ((mask & 2) && run("ring, next tile prefetched", ring_kernel<false, false, true>, ring)) return 1;
    if ((mask & 4) && run("ring, prefetched, nt input loads", ring_kernel<false, true, true>, ring)) return 1;
    if ((mask & 8) && run("ring_full (own place, no reuse)", ring_kernel<true, false, false>, work)) return 1;
    if ((mask & 16) && run("ring_full, prefetched", ring_kernel<true, false, true>, work)) return 1;
    if ((mask & 32) && run("ring, prefetched, L2-local counters", ring_kernel<false, false, true, true>, ring)) return 1;
    if ((mask & 64) && run("ring, prefetched, nt, L2-local", ring_kernel<false, true, true, true>, ring)) return 1;
    return 0;
}
