// plane_yz.hip -- pass Y and the fused pass Z of one filter radius in ONE persistent kernel, the
// x-plane between them handed over through the XCD's L2 instead of HBM (512^3 boxes, G = 2).
//
// reference loops being replaced (per radius, both filtered grids): the y- and z-parts of
// dft_c2r_cube (src/py21cmfast/src/dft.c:18-44) + calculate_fcoll_grid + find_ionised_regions
// (src/py21cmfast/src/IonisationBox.c:773-962,1008-1159) -- the arithmetic is that of
// line_pass_kernel<512, +1, 0> and zw_ionise_kernel<16, false> (fft_native.hip), instruction for
// instruction (same DFT code from fft_device.h, same butterflies per thread, same lane-to-cell map
// in the barrier loop, same order of the f_coll partial sums), so x_HI, z_reion and the f_coll
// means are bit-identical to the two-kernel sequence.
//
// Why: after pass X the two work spectra make three more trips through HBM per radius -- pass Y
// reads and writes them (16 N bytes), pass Z reads them again (8 N).  An x-plane of both spectra is
// 2 x 512 x 256 float2 = 2.1 MB and both passes are local to it: pass Y's tiles are "all 512 rows
// (y) x 16 k_z columns" of the plane, pass Z's lines are its rows.  An XCD's L2 holds 4 MB.
//
// How: 256 workgroups of 512 threads, one per CU.  The 32 workgroups of an XCD (blockIdx % 8; the
// dispatcher deals workgroups round robin over the XCDs, checked against HW_REG_XCC_ID at run
// time, see "placement" below) own the planes x = xcd, xcd + 8, ... and ONE 2.1 MB slot of a
// scratch ring.  Per plane each workgroup
//   Y   transforms one tile (its k_z tile of one grid: 32 tiles per plane), first and last radix-8
//       stage on the registers, one stage in LDS, and stores it into the slot with PLAIN stores:
//       they stay in this XCD's L2;
//       drains its stores (vmcnt(0)), arrives on the XCD's y-counter, waits for the 32 arrivals;
//   Z   reads its 16 rows of both grids from the slot with sc1 loads (they bypass the per-CU L1,
//       which is never refreshed by other CUs' stores, and are served by the XCD's L2), arrives on
//       the z-counter (the next plane's stores into the slot wait for it), and runs the wave-level
//       complex-to-real transforms, the f_coll partial sum and the barrier test on the mask rows.
// The next plane's tile is requested from HBM (nt: streamed, it should not push the slot out of the
// L2) while the current one is in its LDS stage.  Measured with the bytes alone
// (tools/l2_plane_probe.hip): 434 against 619 us per radius-equivalent, HBM traffic 1.43 against
// 3.07 GB (the slot's lines are written back only where the streaming reads evict them).
//
// Placement.  Results must not depend on where workgroups run.  The counters are agent-scope
// atomics (correct anywhere).  The PAYLOAD takes the L2 short cut only if every workgroup finds
// itself on XCD blockIdx % 8: each reads HW_REG_XCC_ID, reports a mismatch, and all meet at one
// start barrier; if any mismatch was reported, every workgroup switches to write-through stores
// and system-scope loads (sc0 sc1 both sides: correct for any placement, at HBM speed).  Spins are
// bounded: a workgroup that waits too long raises `timeout`, everybody stops waiting, and the host
// reports the failure (c21hip_plane_yz_status) instead of hanging.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "c21hip.h"
#include "c21cm_abi.h"
#include "c21cm_grid.h"

// EXPERIMENTAL (round 5): the two kernels of this file are bit-identical to the separate passes and were
// measured at parity or slower in rounds 4 and 5 (DESIGN.md, appendix of dead ends), so the default
// library does not carry them: build with EXTRA=-DC21CM_EXPERIMENTAL (tools/build_variant.sh) to get
// C21CM_YZ=1|2 back.  Without it c21hip_plane_yz_supported() is 0 and the driver never comes here.
#ifndef C21CM_EXPERIMENTAL
extern "C" int c21hip_plane_yz_supported(int, int, int) { return 0; }
extern "C" int c21hip_plane_yz_ionise(const float *, const float *, unsigned char *, double *, int, int, int, int,
                                      double, double, int, double, int, void *) {
    c21hip_set_error("plane-fused pass Y + Z: not in this build (EXTRA=-DC21CM_EXPERIMENTAL)");
    return C21CM_VALUE_ERROR;
}
extern "C" int c21hip_plane_yz_status(void *) { return 0; }
extern "C" int c21hip_plane_yz_profile(unsigned long long *) { return -1; }
#else

namespace {
#include "fft_device.h"

constexpr int N = 512;           // line length in x, y, z
constexpr int H = N / 2;         // complex points of a z-line (split layout, Nyquist apart)
constexpr int TZ = 16;           // k_z columns per pass-Y tile
constexpr int WG = 512;          // threads per workgroup
constexpr int GROUP = 32;        // workgroups per XCD = tiles per plane (2 grids x 16)
constexpr int A = 16, P = 16;    // wave-level c2r: 16 lanes per line, 16 points per lane
constexpr int LINE_LDS = A * (P + 1) + 4;
constexpr size_t PLANE = (size_t)N * H;  // float2 per plane and grid

typedef unsigned int __attribute__((address_space(1))) gu32;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

struct YzSync {                // zeroed by the host before every launch
    unsigned y_done[8][32];    // one counter per XCD, each on its own 128-byte line
    unsigned z_done[8][32];
    unsigned start[32];        // start barrier
    unsigned mismatch, timeout, pad[30];
};

struct YzArgs {
    const float2 *main[2];     // work spectra after pass X: [x][y][H] (density, emissivity)
    const float2 *nyq[2];      // their Nyquist planes [x][y] AFTER the y-transform
    float2 *ring;              // 8 slots x 2 grids x PLANE
    unsigned char *first_cross;
    double *partials;          // nx * ny / 16
    double rhocrit_omb, ion_eff, f_limit;
    int mass_dep_zeta, r_index, store_all;
    unsigned long long *prof;  // C21X_YZ_PROF builds: [256][8] ticks (100 MHz) per phase
    int force_safe;            // test hook (C21CM_YZ_FORCE_SAFE=1): take the write-through path
    YzSync *sync;
    YzSync *status;            // sticky copy of mismatch / timeout (never zeroed by a launch)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int AUX>
__device__ __forceinline__ float2 ld8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <int AUX>
__device__ __forceinline__ void st16(float4 f, __amdgpu_buffer_rsrc_t r, unsigned off) {
    v4u v;
    v.x = __float_as_uint(f.x), v.y = __float_as_uint(f.y), v.z = __float_as_uint(f.z), v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
}

// bounded spin of ONE lane on an agent-scope counter, then the workgroup barrier; `dead` (LDS) is set
// once anything timed out, after which nobody waits any more
__device__ __forceinline__ void wait_ge(unsigned *cnt, unsigned target, YzSync *sync, int *dead) {
    if (threadIdx.x == 0 && !*dead) {
        unsigned spins = 0;
        while (__hip_atomic_load((gu32 *)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 &&
                (spins > (1u << 21) ||
                 __hip_atomic_load((gu32 *)&sync->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store((gu32 *)&sync->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *dead = 1;
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void arrive(unsigned *cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores / loads have completed
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add((gu32 *)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef C21X_YZ_PROF
#define C21X_YZ_PROF 0  // 1: thread 0 of every workgroup accumulates s_memrealtime ticks per phase
#endif
#if C21X_YZ_PROF
#define YZ_TICK(slot)                                      \
    do {                                                   \
        if (threadIdx.x == 0) {                            \
            const unsigned long long t_ = wall_clock64();  \
            prof_acc[slot] += t_ - prof_t;                 \
            prof_t = t_;                                   \
        }                                                  \
    } while (0)
#else
#define YZ_TICK(slot) do { } while (0)
#endif

__global__ void __launch_bounds__(WG)
plane_yz_kernel(YzArgs a, const float2 *__restrict__ tw_global, const float2 *__restrict__ twH_global,
                const float2 *__restrict__ twN_global) {
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);  // [N][TZ]
    float2 *tw = tile + N * TZ;                          // [N]   exp(-2 pi i t / 512)
    float2 *twH = tw + N;                                // [H]   exp(-2 pi i t / 256)
    float2 *twN = twH + H;                               // [H]   exp(-2 pi i t / 512), t < 256
    float2 *lines = twN + H;                             // 32 line regions of the wave-level c2r
    double *red = reinterpret_cast<double *>(lines + 32 * LINE_LDS);  // [4]
    int *dead = reinterpret_cast<int *>(red + 4);
    for (int t = threadIdx.x; t < N; t += WG) tw[t] = tw_global[t];
    for (int t = threadIdx.x; t < H; t += WG) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    if (threadIdx.x == 0) *dead = 0;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    YzSync *sync = a.sync;

    // ---- placement: every workgroup on the XCD its index implies?  One start barrier decides.
    if (threadIdx.x == 0) {
        const unsigned id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;  // HW_REG_XCC_ID
        if ((int)id != xcd || a.force_safe) {
            __hip_atomic_store((gu32 *)&sync->mismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32 *)&a.status->mismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (the mismatch store is ordered before the arrival: same address space, same lane, and the
        // arrival below is issued after a vmcnt(0))
    }
    __syncthreads();
    arrive(&sync->start[0]);
    wait_ge(&sync->start[0], gridDim.x, sync, dead);
    if (threadIdx.x == 0)
        dead[1] = (int)__hip_atomic_load((gu32 *)&sync->mismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool safe = dead[1] != 0;

    // ---- pass Y roles: thread (r0, c4) owns rows r0 + 64 u of the column pair c4
    const int r0 = threadIdx.x >> 3, c4 = threadIdx.x & 7;
    float2 twd1[8];
    {
        twd1[0] = make_float2(1.f, 0.f);
        twd1[1] = tw[r0];
        twd1[2] = tw[2 * r0];
        twd1[4] = tw[4 * r0];
        twd1[3] = cmul(twd1[1], twd1[2]);
        twd1[5] = cmul(twd1[1], twd1[4]);
        twd1[6] = cmul(twd1[2], twd1[4]);
        twd1[7] = cmul(twd1[3], twd1[4]);
#pragma unroll
        for (int j = 1; j < 8; j++) twd1[j].y = -twd1[j].y;
    }
    const int yg = k >> 4, yct = k & 15;  // this workgroup's tile: grid, column tile
    const __amdgpu_buffer_rsrc_t src_rs = make_rsrc(yg ? a.main[1] : a.main[0], (unsigned)(N * PLANE * sizeof(float2)));
    const unsigned src_lane = (unsigned)(((size_t)r0 * H + yct * TZ + 2 * c4) * sizeof(float2));
    float2 *slot0 = a.ring + (size_t)xcd * 2 * PLANE;  // density plane, the emissivity plane follows
    const __amdgpu_buffer_rsrc_t slot_rs = make_rsrc(slot0, (unsigned)(2 * PLANE * sizeof(float2)));
    const unsigned slot_store = (unsigned)(((size_t)yg * PLANE + (size_t)r0 * H + yct * TZ + 2 * c4) * sizeof(float2));

    // ---- pass Z roles: 32 groups of 16 lanes; groups 0..15 (waves 0-3) transform the density line of
    // row 16 k + lw, groups 16..31 (waves 4-7) the emissivity line of the same row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gi = wave * 4 + lane / P, b = lane % P;
    const int lw = gi & 15, role = gi >> 4;
    float2 *L = lines + gi * LINE_LDS;
    const float2 *Lpartner = lines + (gi + 16) * LINE_LDS;  // role 0 reads its row's emissivity here
    const unsigned slot_load = (unsigned)(((size_t)role * PLANE + (size_t)(16 * k + lw) * H + b) * sizeof(float2));

    const double floor_lhs = a.f_limit * a.ion_eff;
    const bool floor_ionises = a.mass_dep_zeta && (floor_lhs > 1.);
    const float dmin = (float)(-1. + 1e-7);  // IonisationBox.c:803

    unsigned *ycnt = &sync->y_done[xcd][0], *zcnt = &sync->z_done[xcd][0];
    constexpr int n_planes = N / 8;
#if C21X_YZ_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = wall_clock64();
#endif
    float4 reg[8];
    {
        const unsigned base = (unsigned)((size_t)xcd * PLANE * sizeof(float2)) + src_lane;
#pragma unroll
        for (int u = 0; u < 8; u++) reg[u] = ld16<2>(src_rs, base + (unsigned)(u * 64 * H * sizeof(float2)));
    }
    __syncthreads();

    // ================= pass Y of this workgroup's tile of plane xcd + 8 i (line_pass_kernel<512, +1, 0>,
    // F512): registers -> outv; the tile of plane i + 1 is requested once `reg` has gone to LDS
    float4 outv[8];
    auto y_compute = [&](int i) {
        {
            float2 c0[8], c1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0[u] = make_float2(reg[u].x, reg[u].y);
                c1[u] = make_float2(reg[u].z, reg[u].w);
            }
            Dft<8, +1>::run(c0);
            Dft<8, +1>::run(c1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float2 o0 = c0[j], o1 = c1[j];
                if (j > 0) {
                    o0 = cmul(o0, twd1[j]);
                    o1 = cmul(o1, twd1[j]);
                }
                *reinterpret_cast<float4 *>(tile + (8 * r0 + j) * TZ + 2 * c4) = make_float4(o0.x, o0.y, o1.x, o1.y);
            }
        }
        __syncthreads();
        {
            const int xn = xcd + 8 * ((i + 1 < n_planes) ? i + 1 : i);  // (past the end: the same tile again, in bounds)
            const unsigned base = (unsigned)((size_t)xn * PLANE * sizeof(float2)) + src_lane;
#pragma unroll
            for (int u = 0; u < 8; u++) reg[u] = ld16<2>(src_rs, base + (unsigned)(u * 64 * H * sizeof(float2)));
        }
        {
            const int obase = (r0 & 7) + 64 * (r0 >> 3);
            float2 s0[8], s1[8];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const float4 t = *reinterpret_cast<const float4 *>(tile + (r0 + 64 * kk) * TZ + 2 * c4);
                s0[kk] = make_float2(t.x, t.y);
                s1[kk] = make_float2(t.z, t.w);
            }
            __syncthreads();
            float2 w2[8];
            {
                const int ps = r0 & ~7;
                w2[1] = tw[ps];
                w2[2] = tw[2 * ps];
                w2[4] = tw[4 * ps];
                w2[3] = cmul(w2[1], w2[2]);
                w2[5] = cmul(w2[1], w2[4]);
                w2[6] = cmul(w2[2], w2[4]);
                w2[7] = cmul(w2[3], w2[4]);
#pragma unroll
                for (int j = 1; j < 8; j++) w2[j].y = -w2[j].y;
            }
            Dft<8, +1>::run(s0);
            Dft<8, +1>::run(s1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float2 o0 = s0[j], o1 = s1[j];
                if (j > 0) {
                    o0 = cmul(o0, w2[j]);
                    o1 = cmul(o1, w2[j]);
                }
                *reinterpret_cast<float4 *>(tile + (obase + 8 * j) * TZ + 2 * c4) = make_float4(o0.x, o0.y, o1.x, o1.y);
            }
            __syncthreads();
        }
        {
            float2 c0[8], c1[8];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const float4 t = *reinterpret_cast<const float4 *>(tile + (r0 + 64 * kk) * TZ + 2 * c4);
                c0[kk] = make_float2(t.x, t.y);
                c1[kk] = make_float2(t.z, t.w);
            }
            Dft<8, +1>::run(c0);
            Dft<8, +1>::run(c1);
#pragma unroll
            for (int j = 0; j < 8; j++) outv[j] = make_float4(c0[j].x, c0[j].y, c1[j].x, c1[j].y);
        }
        __syncthreads();  // the tile buffer is free for the next y_compute
    };
    // outv -> the XCD's slot (plain stores: they stay in this L2), then the arrival on the y-counter
    auto y_publish = [&]() {
        if (!safe) {
#pragma unroll
            for (int j = 0; j < 8; j++) st16<0>(outv[j], slot_rs, slot_store + (unsigned)(j * 64 * H * sizeof(float2)));
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) st16<17>(outv[j], slot_rs, slot_store + (unsigned)(j * 64 * H * sizeof(float2)));
        }
        arrive(ycnt);
    };

    // Order of a workgroup's phases: Y(0) publish(0) | Y(1) Z(0) publish(1) | Y(2) Z(1) publish(2) | ...
    // Every wait then follows a whole compute phase that started after this workgroup's own arrival:
    // "plane i complete" is awaited after the pass Y of tile i + 1, "slot free" after the pass Z
    // compute of plane i -- by then the other 31 workgroups have long arrived.
    y_compute(0);
    YZ_TICK(0);
    y_publish();
    YZ_TICK(2);
    for (int i = 0; i < n_planes; i++) {
        const int x = xcd + 8 * i;
        if (i + 1 < n_planes) y_compute(i + 1);
        YZ_TICK(0);  // pass Y compute (incl. the wait for the prefetched tile)

        // ================= pass Z of rows 16 k .. 16 k + 15 of plane x (zw_ionise_kernel<16, false>)
        const long lline = (long)x * N + 16 * k + lw;
        unsigned char *mrow = a.first_cross + lline * N;
        uchar2 old[A];
        float nyq_re;
        if (role == 0) {
#pragma unroll
            for (int q = 0; q < A; q++) old[q] = reinterpret_cast<const uchar2 *>(mrow)[b + A * q];
        }
        nyq_re = (role ? a.nyq[1] : a.nyq[0])[lline].x;
        wait_ge(ycnt, (unsigned)GROUP * (i + 1), sync, dead);
        YZ_TICK(3);  // wait: plane complete
        float2 xv[A];
        if (!safe) {
#pragma unroll
            for (int q = 0; q < A; q++) xv[q] = ld8<16>(slot_rs, slot_load + (unsigned)(P * q * sizeof(float2)));
        } else {
#pragma unroll
            for (int q = 0; q < A; q++) xv[q] = ld8<17>(slot_rs, slot_load + (unsigned)(P * q * sizeof(float2)));
        }
        arrive(zcnt);  // (its vmcnt(0) is the wait for the rows; the slot is free once all 32 have arrived)
        YZ_TICK(4);  // row loads from the slot + drain + arrival
        wave_c2r<A, P>(xv, nyq_re, L, twH, twN, b);
        double acc = 0.;
        if (role == 1) {
            // emissivity: the sum of max(s, 0) in zw_ionise_kernel's order, the cells to the partner
            wave_fence();
#pragma unroll
            for (int q = 0; q < A; q++) {
                L[q * P + b] = xv[q];
                const float s0 = fmaxf(xv[q].x, 0.f), s1 = fmaxf(xv[q].y, 0.f);
                acc += (double)s0;
                acc += (double)s1;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0) red[wave - 4] = acc;
        }
        __syncthreads();
        if (role == 0) {
#pragma unroll
            for (int q = 0; q < A; q++) {
                const int j = b + A * q;  // cells (2 j, 2 j + 1) of the row
                const float2 xs = Lpartner[q * P + b];
                const float s0 = fmaxf(xs.x, 0.f), s1 = fmaxf(xs.y, 0.f);
                const double D0 = a.rhocrit_omb * (1. + (double)fmaxf(xv[q].x, dmin));
                const double D1 = a.rhocrit_omb * (1. + (double)fmaxf(xv[q].y, dmin));
                const bool i0 = floor_ionises || ((double)s0 * a.ion_eff > D0);
                const bool i1 = floor_ionises || ((double)s1 * a.ion_eff > D1);
                uchar2 m = old[q];
                const bool n0 = i0 && m.x == 0, n1 = i1 && m.y == 0;
                if (n0) m.x = (unsigned char)a.r_index;
                if (n1) m.y = (unsigned char)a.r_index;
                if (n0 || n1 || a.store_all) reinterpret_cast<uchar2 *>(mrow)[j] = m;
            }
        }
        if (threadIdx.x == 256) {
            double sum = 0.;
#pragma unroll
            for (int w = 0; w < 4; w++) sum += red[w];
            a.partials[(long)x * GROUP + k] = sum;
        }
        YZ_TICK(5);  // pass Z compute + mask
        if (i + 1 < n_planes) {
            // the slot may be overwritten once every workgroup of the XCD has read plane i
            wait_ge(zcnt, (unsigned)GROUP * (i + 1), sync, dead);
            YZ_TICK(1);  // wait: slot free
            y_publish();
            YZ_TICK(2);  // slot stores + drain + arrival
        }
    }
#if C21X_YZ_PROF
    if (threadIdx.x == 0 && a.prof)
        for (int t = 0; t < 8; t++) a.prof[blockIdx.x * 8 + t] = prof_acc[t];
#endif
    if (threadIdx.x == 0 && *dead)
        __hip_atomic_store((gu32 *)&a.status->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the same plane protocol with the two passes on SEPARATE WAVES of one workgroup ---------------
// plane_yz_kernel runs pass Y and pass Z of a plane one after the other in all eight waves, and a
// plane costs the SUM of their instruction streams plus every wait (13.4 us).  Here a
// workgroup is 1024 threads: waves 0-7 are the Y role (the tile pipeline above, one tile per plane),
// waves 8-15 the Z role (rows of the planes behind it), each role with its own loop, linked only
// through the XCD's two counters exactly like two workgroups would be -- the waits of one role lie
// under the arithmetic of the other and four waves per SIMD hide the LDS round trips of the wave-level
// transform.
// No s_barrier after the prologue (it would tie the roles together): the eight waves of a role meet at
// software barriers on LDS counters (monotonic epochs; the LDS operations of a wave execute in issue
// order, so its counter increment follows its data writes), the last wave to drain its stores / loads
// arrives on the XCD's counter for the workgroup, one lane polls the other counter and publishes it.
// In the Z role every wave carries BOTH grids of two rows (lanes 0-31 density, 32-63 emissivity), so the
// two values of a cell meet inside the wave and no Z wave ever waits for another.
// Measured (MI355X, tools/time_yz.py): 0.79-0.84 ms per radius, the separate passes 0.47 + 0.34 (+ 0.03
// for the Nyquist planes' pass Y, which this kernel needs too): at parity, bit-identical.  Per plane and
// CU (C21X_YZ_PROF): Y role 5.3 us of transform (2.7 us when it has the SIMDs to itself), 1.3 us of
// stores, the rest waiting; Z role 0.6 us row loads, 2.9 us transform, 1.8 us barrier test, the rest
// waiting.  Without the tile reads from HBM (timing experiment) 0.68 ms = 9.3 us per plane, of which
// ~8 us are the two instruction streams sharing the four SIMDs: the kernel is bound by instruction
// issue, not by the hand-off, and cannot reach the ~0.45 ms its HBM bytes would allow.  Also measured:
// polling through the scalar data path (slower: 0.97 ms), wave priorities (no change), and one HBM effect
// -- tile columns whose 128-byte lines lie at 384 (mod 1024) of the address arrive ~2.5 us later than the
// others, whichever workgroup reads them (it moves with the base address) and gate every plane.
constexpr int WG2 = 1024;
#ifndef C21X_YZ_PRIO
#define C21X_YZ_PRIO 1
#endif
#ifndef C21X_YZ_SRC_AUX
#define C21X_YZ_SRC_AUX 2  // cache policy of the tile reads: 2 = nt (streamed)
#endif
#ifndef C21X_YZ_SCALAR_POLL
#define C21X_YZ_SCALAR_POLL 0
#endif
#if C21X_YZ_PROF
#define YZ2_TICK(slot)                                     \
    do {                                                   \
        if (threadIdx.x == 0 || threadIdx.x == 512) { \
            const unsigned long long t_ = wall_clock64();  \
            prof_acc[slot] += t_ - prof_t;                 \
            prof_t = t_;                                   \
        }                                                  \
    } while (0)
#else
#define YZ2_TICK(slot) do { } while (0)
#endif
struct Roles {            // LDS words of the software synchronisation, all monotonic
    int y_bar;            // Y role: stage barriers (8 arrivals per barrier)
    int y_go;             // Y role: epoch for which "slot free" has been seen by the polling lane
    int y_stored;         // Y role: waves whose slot stores have drained
    int z_go;             // Z role: epoch for which "plane complete" has been seen
    int z_loaded;         // Z role: waves whose row loads have landed
    int z_red;            // Z role: waves whose per-lane f_coll sums of the plane are in LDS
    int dead;             // a wait timed out somewhere: nobody waits any more
    int safe;
};

// (LDS pointers carry their address space: a poll through a generic pointer is a FLAT load, which
//  counts on vmcnt as well and so waits for every global load in flight -- the prefetched tile)
typedef int __attribute__((address_space(3))) lds_int;
__device__ __forceinline__ int lds_peek(int *p) {
    return __hip_atomic_load((lds_int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_poke(int *p, int v) {
    __hip_atomic_store((lds_int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_bump(int *p) {
    return __hip_atomic_fetch_add((lds_int *)p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool lds_wait_ge(int *cnt, int target, int *dead) {
    unsigned spins = 0;
    while (lds_peek(cnt) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0 && (lds_peek(dead) || spins > (1u << 24))) {
            lds_poke(dead, 1);
            return false;
        }
    }
    return true;
}
// barrier among the `n` waves of a role: every lane waits, lane 0 of each wave arrives
__device__ __forceinline__ void role_barrier(int *cnt, int target, int *dead) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if ((threadIdx.x & 63) == 0) (void)lds_bump(cnt);
    (void)lds_wait_ge(cnt, target, dead);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void role_arrive(int *cnt) {  // split barrier: arrival ...
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if ((threadIdx.x & 63) == 0) (void)lds_bump(cnt);
}
__device__ __forceinline__ void role_wait(int *cnt, int target, int *dead) {  // ... and the wait
    (void)lds_wait_ge(cnt, target, dead);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// A counter of the XCD read through the SCALAR data path (glc: past the scalar cache, from the L2).  A
// vector load would queue in the CU's texture path behind the tile prefetch and the slot stores of all
// sixteen waves and come back microseconds later; the scalar path is idle here.
__device__ __forceinline__ unsigned scalar_peek(const unsigned *p) {
#if C21X_YZ_SCALAR_POLL
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
#else
    return __hip_atomic_load((gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void global_wait_ge(unsigned *cnt, unsigned target, YzSync *sync, int *dead) {
    unsigned spins = 0;
    while (scalar_peek(cnt) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0 &&
            (lds_peek(dead) || spins > (1u << 21) ||
             __hip_atomic_load((gu32 *)&sync->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store((gu32 *)&sync->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_poke(dead, 1);
            break;
        }
    }
}

__global__ void __launch_bounds__(WG2)
plane_yz2_kernel(YzArgs a, const float2 *__restrict__ tw_global, const float2 *__restrict__ twH_global,
                 const float2 *__restrict__ twN_global) {
    extern __shared__ float4 lds_raw[];
    float2 *tile = reinterpret_cast<float2 *>(lds_raw);  // [N][TZ]
    float2 *tw = tile + N * TZ;
    float2 *twH = tw + N;
    float2 *twN = twH + H;
    float2 *lines = twN + H;
    Roles *R = reinterpret_cast<Roles *>(lines + 32 * LINE_LDS);
    for (int t = threadIdx.x; t < N; t += WG2) tw[t] = tw_global[t];
    for (int t = threadIdx.x; t < H; t += WG2) {
        twH[t] = twH_global[t];
        twN[t] = twN_global[t];
    }
    if (threadIdx.x < (int)(sizeof(Roles) / sizeof(int))) reinterpret_cast<int *>(R)[threadIdx.x] = 0;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    YzSync *sync = a.sync;
    // ---- placement check and start barrier (the only s_barriers of the kernel)
    if (threadIdx.x == 0) {
        const unsigned id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;  // HW_REG_XCC_ID
        if ((int)id != xcd || a.force_safe) {
            __hip_atomic_store((gu32 *)&sync->mismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32 *)&a.status->mismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add((gu32 *)&sync->start[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        global_wait_ge(&sync->start[0], gridDim.x, sync, &R->dead);
        lds_poke(&R->safe, (int)__hip_atomic_load((gu32 *)&sync->mismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    const bool safe = lds_peek(&R->safe) != 0;
    unsigned *ycnt = &sync->y_done[xcd][0], *zcnt = &sync->z_done[xcd][0];
    constexpr int n_planes = N / 8;
    float2 *slot0 = a.ring + (size_t)xcd * 2 * PLANE;
    const __amdgpu_buffer_rsrc_t slot_rs = make_rsrc(slot0, (unsigned)(2 * PLANE * sizeof(float2)));
#if C21X_YZ_PROF
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = wall_clock64();  // Y role: slots 0-3, Z role: 4-7
#endif

    if (threadIdx.x < 512) {
        // ===================== Y role: one tile per plane (line_pass_kernel<512, +1, 0>, F512)
        const int r0 = threadIdx.x >> 3, c4 = threadIdx.x & 7;
        const bool poller = threadIdx.x == 0;
        const int yg = k >> 4, yct = k & 15;
        const __amdgpu_buffer_rsrc_t src_rs = make_rsrc(yg ? a.main[1] : a.main[0], (unsigned)(N * PLANE * sizeof(float2)));
        const unsigned src_lane = (unsigned)(((size_t)r0 * H + yct * TZ + 2 * c4) * sizeof(float2));
        const unsigned slot_store = (unsigned)(((size_t)yg * PLANE + (size_t)r0 * H + yct * TZ + 2 * c4) * sizeof(float2));
        float4 reg[8];
        {
            const unsigned base = (unsigned)((size_t)xcd * PLANE * sizeof(float2)) + src_lane;
#pragma unroll
            for (int u = 0; u < 8; u++) reg[u] = ld16<C21X_YZ_SRC_AUX>(src_rs, base + (unsigned)(u * 64 * H * sizeof(float2)));
        }
        int bar = 0;  // barriers of this role so far
        for (int i = 0; i < n_planes; i++) {
            {
                float2 c0[8], c1[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    c0[u] = make_float2(reg[u].x, reg[u].y);
                    c1[u] = make_float2(reg[u].z, reg[u].w);
                }
                Dft<8, +1>::run(c0);
                Dft<8, +1>::run(c1);
                if (i > 0) role_wait(&R->y_bar, 8 * bar, &R->dead);  // every wave has read stage 3 of the last tile
                float2 twd1[8];  // (rebuilt per tile: 128 registers per lane at four waves per SIMD)
                twd1[1] = tw[r0];
                twd1[2] = tw[2 * r0];
                twd1[4] = tw[4 * r0];
                twd1[3] = cmul(twd1[1], twd1[2]);
                twd1[5] = cmul(twd1[1], twd1[4]);
                twd1[6] = cmul(twd1[2], twd1[4]);
                twd1[7] = cmul(twd1[3], twd1[4]);
#pragma unroll
                for (int j = 1; j < 8; j++) twd1[j].y = -twd1[j].y;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float2 o0 = c0[j], o1 = c1[j];
                    if (j > 0) {
                        o0 = cmul(o0, twd1[j]);
                        o1 = cmul(o1, twd1[j]);
                    }
                    *reinterpret_cast<float4 *>(tile + (8 * r0 + j) * TZ + 2 * c4) = make_float4(o0.x, o0.y, o1.x, o1.y);
                }
            }
            role_barrier(&R->y_bar, 8 * ++bar, &R->dead);
            {
                const int xn = xcd + 8 * ((i + 1 < n_planes) ? i + 1 : i);
                const unsigned base = (unsigned)((size_t)xn * PLANE * sizeof(float2)) + src_lane;
#pragma unroll
                for (int u = 0; u < 8; u++) reg[u] = ld16<C21X_YZ_SRC_AUX>(src_rs, base + (unsigned)(u * 64 * H * sizeof(float2)));
            }
            float4 outv[8];
            {
                const int obase = (r0 & 7) + 64 * (r0 >> 3);
                float2 s0[8], s1[8];
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const float4 t = *reinterpret_cast<const float4 *>(tile + (r0 + 64 * kk) * TZ + 2 * c4);
                    s0[kk] = make_float2(t.x, t.y);
                    s1[kk] = make_float2(t.z, t.w);
                }
                role_barrier(&R->y_bar, 8 * ++bar, &R->dead);
                float2 w2[8];
                {
                    const int ps = r0 & ~7;
                    w2[1] = tw[ps];
                    w2[2] = tw[2 * ps];
                    w2[4] = tw[4 * ps];
                    w2[3] = cmul(w2[1], w2[2]);
                    w2[5] = cmul(w2[1], w2[4]);
                    w2[6] = cmul(w2[2], w2[4]);
                    w2[7] = cmul(w2[3], w2[4]);
#pragma unroll
                    for (int j = 1; j < 8; j++) w2[j].y = -w2[j].y;
                }
                Dft<8, +1>::run(s0);
                Dft<8, +1>::run(s1);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float2 o0 = s0[j], o1 = s1[j];
                    if (j > 0) {
                        o0 = cmul(o0, w2[j]);
                        o1 = cmul(o1, w2[j]);
                    }
                    *reinterpret_cast<float4 *>(tile + (obase + 8 * j) * TZ + 2 * c4) = make_float4(o0.x, o0.y, o1.x, o1.y);
                }
                role_barrier(&R->y_bar, 8 * ++bar, &R->dead);
            }
            {
                float2 c0[8], c1[8];
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const float4 t = *reinterpret_cast<const float4 *>(tile + (r0 + 64 * kk) * TZ + 2 * c4);
                    c0[kk] = make_float2(t.x, t.y);
                    c1[kk] = make_float2(t.z, t.w);
                }
                role_arrive(&R->y_bar);  // (split barrier: the wait is before the next tile's stage-1 writes)
                ++bar;
                Dft<8, +1>::run(c0);
                Dft<8, +1>::run(c1);
#pragma unroll
                for (int j = 0; j < 8; j++) outv[j] = make_float4(c0[j].x, c0[j].y, c1[j].x, c1[j].y);
            }
            YZ2_TICK(0);  // Y: the three stages of the tile
            // slot free?  (every workgroup of the XCD has read plane i - 1)
            if (poller) {
                if (i > 0) global_wait_ge(zcnt, (unsigned)GROUP * i, sync, &R->dead);
                lds_poke(&R->y_go, i + 1);
            }
            (void)lds_wait_ge(&R->y_go, i + 1, &R->dead);
            YZ2_TICK(1);  // Y: wait, slot free
            if (!safe) {
#pragma unroll
                for (int j = 0; j < 8; j++) st16<0>(outv[j], slot_rs, slot_store + (unsigned)(j * 64 * H * sizeof(float2)));
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) st16<17>(outv[j], slot_rs, slot_store + (unsigned)(j * 64 * H * sizeof(float2)));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores (and the prefetch) have completed
            // the last of the eight waves to get here arrives for the workgroup (no wave can be a plane
            // ahead: plane i + 1's stores wait for the z-counter, which waits for this arrival)
            if ((threadIdx.x & 63) == 0 &&
                lds_bump(&R->y_stored) == 8 * (i + 1) - 1)
                __hip_atomic_fetch_add((gu32 *)ycnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            YZ2_TICK(2);  // Y: slot stores, drain, arrival
        }
    } else {
        // ===================== Z role: rows 16 k .. 16 k + 15 of every plane (zw_ionise_kernel<16, false>)
        // Every wave carries BOTH grids of two rows: lanes 0-31 the density lines of rows 2 w, 2 w + 1,
        // lanes 32-63 the emissivity lines of the same rows, so the two values of a cell meet inside the
        // wave (through its own LDS regions, no flag between waves) and each half tests half of the
        // row's cells: the density lanes cells of q = 0..7, the emissivity lanes those of q = 8..15.
        const int zt = threadIdx.x - 512;
        const int lane = zt & 63, wave = zt >> 6;
        const int grp = lane >> 4, b = lane & 15;
        const int role = grp >> 1, lw = 2 * wave + (grp & 1);
        const bool poller = zt == 0;
#if C21X_YZ_PRIO
        __builtin_amdgcn_s_setprio(2);  // (the Z role is the critical chain of a plane; the Y role has slack)
#endif
        float2 *L = lines + (wave * 4 + grp) * LINE_LDS;
        const float2 *Lpartner = lines + (wave * 4 + (grp ^ 2)) * LINE_LDS;
        const unsigned slot_load = (unsigned)(((size_t)role * PLANE + (size_t)(16 * k + lw) * H + b) * sizeof(float2));
        const double floor_lhs = a.f_limit * a.ion_eff;
        const bool floor_ionises = a.mass_dep_zeta && (floor_lhs > 1.);
        const float dmin = (float)(-1. + 1e-7);
        // f_coll partial sum of the block in zw_ionise_kernel's order: there emissivity wave v holds rows
        // 4 v .. 4 v + 3, lane (row % 4) * 16 + b the sum of its 32 cells, a shuffle tree over the 64 lanes,
        // then the four waves in order.  Here the per-lane sums go to `accs` in that arrangement and the
        // last wave to arrive runs the four trees.
        double *accs = reinterpret_cast<double *>(R + 1);  // [2 (plane parity)][4][64]
        const int acc_slot = (lw >> 2) * 64 + (lw & 3) * 16 + b;
        // (nothing from HBM may be outstanding at the poll: its vmcnt(0) would wait for that too -- the
        //  Nyquist coefficient of a line is fetched one plane ahead, the mask rows after the arrival)
        float nyq_next = (role ? a.nyq[1] : a.nyq[0])[(long)xcd * N + 16 * k + lw].x;
        for (int i = 0; i < n_planes; i++) {
            const int x = xcd + 8 * i;
            const long lline = (long)x * N + 16 * k + lw;
            uchar2 *mrow = reinterpret_cast<uchar2 *>(a.first_cross + lline * N) + b + A * 8 * role;
            uchar2 old[A / 2];
            const float nyq_re = nyq_next;
            if (poller) {
                global_wait_ge(ycnt, (unsigned)GROUP * (i + 1), sync, &R->dead);
                lds_poke(&R->z_go, i + 1);
            }
            (void)lds_wait_ge(&R->z_go, i + 1, &R->dead);
            YZ2_TICK(0);  // Z: wait, plane complete
            float2 xv[A];
            if (!safe) {
#pragma unroll
                for (int q = 0; q < A; q++) xv[q] = ld8<16>(slot_rs, slot_load + (unsigned)(P * q * sizeof(float2)));
            } else {
#pragma unroll
                for (int q = 0; q < A; q++) xv[q] = ld8<17>(slot_rs, slot_load + (unsigned)(P * q * sizeof(float2)));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have landed
            if (lane == 0 && lds_bump(&R->z_loaded) == 8 * (i + 1) - 1)
                __hip_atomic_fetch_add((gu32 *)zcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            YZ2_TICK(1);  // Z: row loads from the slot, arrival
#pragma unroll
            for (int q = 0; q < A / 2; q++) old[q] = mrow[A * q];  // (their latency lies under the transform)
            nyq_next = (role ? a.nyq[1] : a.nyq[0])[lline + (i + 1 < n_planes ? 8 * N : 0)].x;
            // (the lane index made opaque per plane: otherwise the ~40 LDS addresses of the transform's
            //  exchange steps are hoisted out of the plane loop, and at 128 registers per lane spilled)
            int bq = b;
            asm volatile("" : "+v"(bq));
            wave_c2r<A, P>(xv, nyq_re, L, twH, twN, bq);
            YZ2_TICK(2);  // Z: wave-level c2r
            double *accp = accs + 256 * (i & 1);  // (two planes apart is safe: a wave cannot load plane i + 2
                                                  //  before every wave of this workgroup has loaded i + 1)
            wave_fence();
            if (role == 1) {
                double acc = 0.;
#pragma unroll
                for (int q = 0; q < A; q++) {
                    const float s0 = fmaxf(xv[q].x, 0.f), s1 = fmaxf(xv[q].y, 0.f);
                    acc += (double)s0;
                    acc += (double)s1;
                }
                accp[acc_slot] = acc;
            }
            // the half of its line a lane does not test goes to the partner lane
#pragma unroll
            for (int q = 0; q < A / 2; q++) {
                // (selects of VALUES: `role ? xv[q] : xv[q + 8]` on the array elements themselves is a
                //  select of addresses and sends the whole array to scratch)
                const float2 lo = xv[q], hi = xv[q + A / 2];
                L[q * P + bq] = make_float2(role ? lo.x : hi.x, role ? lo.y : hi.y);
            }
            wave_fence();
#pragma unroll
            for (int q = 0; q < A / 2; q++) {
                const float2 lo = xv[q], hi = xv[q + A / 2], other = Lpartner[q * P + bq];
                const float2 own = make_float2(role ? hi.x : lo.x, role ? hi.y : lo.y);
                const float2 dn = make_float2(role ? other.x : own.x, role ? other.y : own.y);
                const float2 xs = make_float2(role ? own.x : other.x, role ? own.y : other.y);
                const float s0 = fmaxf(xs.x, 0.f), s1 = fmaxf(xs.y, 0.f);
                const double D0 = a.rhocrit_omb * (1. + (double)fmaxf(dn.x, dmin));
                const double D1 = a.rhocrit_omb * (1. + (double)fmaxf(dn.y, dmin));
                const bool i0 = floor_ionises || ((double)s0 * a.ion_eff > D0);
                const bool i1 = floor_ionises || ((double)s1 * a.ion_eff > D1);
                uchar2 m = old[q];
                const bool n0 = i0 && m.x == 0, n1 = i1 && m.y == 0;
                if (n0) m.x = (unsigned char)a.r_index;
                if (n1) m.y = (unsigned char)a.r_index;
                if (n0 || n1 || a.store_all) mrow[A * q] = m;
            }
            wave_fence();  // (the regions are the next transform's)
            // the block's partial sum: by the last wave to get here
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            int arrived = 0;
            if (lane == 0) arrived = lds_bump(&R->z_red);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            if (arrived == 8 * (i + 1) - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                double sum = 0.;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    double t = accp[v * 64 + lane];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
                    sum += t;
                }
                if (lane == 0) a.partials[(long)x * GROUP + k] = sum;
            }
            YZ2_TICK(3);  // Z: exchange, barrier test, mask stores, partial sum
        }
    }
#if C21X_YZ_PROF
    if ((threadIdx.x == 0 || threadIdx.x == 512) && a.prof)
        for (int t = 0; t < 4; t++) a.prof[blockIdx.x * 8 + (threadIdx.x >> 9) * 4 + t] = prof_acc[t];
#endif
    if ((threadIdx.x & 511) == 0 && lds_peek(&R->dead))
        __hip_atomic_store((gu32 *)&a.status->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct YzState {
    float2 *ring = nullptr;
    YzSync *sync = nullptr, *status = nullptr;
    int attr_done = 0, attr2_done = 0;
    int usable = -1, resident = -1;
    unsigned long gen = 0;
} g_yz;
}  // namespace

extern "C" const void *c21hip_twiddles_dev(int n);
extern "C" void *c21hip_ktime_begin(int kind, void *stream);
extern "C" void c21hip_ktime_end(void *scope);

// 1: the plane-fused pass Y + Z is selected (C21CM_YZ=1) and serves this box (512^3 on a device with
// 8 XCDs x 32 CUs).  OFF by default: measured on the MI355X it is bit-identical to the separate passes
// and moves 0.47 of their HBM bytes, but takes 0.90 ms per radius against 0.47 + 0.34 ms -- the wave-level
// c2r + barrier loop of a plane's rows is ~3000 wave-instructions per SIMD (5.7 us per plane at two
// waves per SIMD: VALU-issue bound), pass Y's butterflies 2.7 us, and one workgroup per CU runs them
// one after the other, where the separate pass Z overlaps four workgroups per CU.
extern "C" int c21hip_plane_yz_supported(int nx, int ny, int nz) {
    const char *e = getenv("C21CM_YZ");
    if (!(e && (e[0] == '1' || e[0] == '2'))) return 0;  // 2: the role-split kernel
    if (nx != N || ny != N || nz != N) return 0;
    if (g_yz.usable < 0) {
        int dev = 0, cus = 0;
        g_yz.usable = (hipGetDevice(&dev) == hipSuccess &&
                       hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                       cus == 8 * GROUP) ? 1 : 0;
    }
    return g_yz.usable;
}

// Pass Y + fused pass Z of one radius on work spectra that have been through pass X (main blocks)
// and whose Nyquist planes have ALSO been through their y-transform (c21hip_split_y_nyq).
// partials: nx * ny / 16 doubles, as c21hip_split_z_ionise_stars leaves them.
extern "C" int c21hip_plane_yz_ionise(const float *delta_work, const float *stars_work,
                                      unsigned char *first_cross, double *partials, int nx, int ny, int nz,
                                      int r_index, double rhocrit_omb, double ion_eff, int mass_dep_zeta,
                                      double f_limit, int store_all, void *stream_) {
    if (!c21hip_plane_yz_supported(nx, ny, nz)) {
        c21hip_set_error("plane-fused pass Y + Z: unsupported box %d x %d x %d", nx, ny, nz);
        return C21CM_VALUE_ERROR;
    }
    hipStream_t stream = (hipStream_t)stream_;
    {   // workspace slots (c21cm_release_device_cache frees them: ask every call)
        g_yz.ring = (float2 *)c21hip_ws(250, 8 * 2 * PLANE * sizeof(float2));
        YzSync *sy = (YzSync *)c21hip_ws(251, 2 * sizeof(YzSync) + 8192);
        if (!g_yz.ring || !sy) return C21CM_MEMORY_ALLOC_ERROR;
        // (a released and re-allocated slot can come back at the same address: the release generation of
        //  the workspace decides, not the pointer -- ADVICE r4)
        const unsigned long gen = c21hip_ws_generation();
        if (sy != g_yz.sync || gen != g_yz.gen) {  // new allocation: the sticky status starts clean
            g_yz.gen = gen;
            g_yz.sync = sy;
            g_yz.status = (YzSync *)((char *)sy + sizeof(YzSync) + 4096);
            if (hipMemsetAsync(g_yz.status, 0, sizeof(YzSync), stream) != hipSuccess) return C21CM_IO_ERROR;
        }
    }
    const float2 *tw = (const float2 *)c21hip_twiddles_dev(N);
    const float2 *twH = (const float2 *)c21hip_twiddles_dev(H);
    if (!tw || !twH) return C21CM_MEMORY_ALLOC_ERROR;
    if (g_yz.resident < 0) {
        // the plane protocol needs all 256 workgroups resident at once: one per CU must fit (a CU
        // mask or a shared device can still defeat it, which the bounded spins report as a timeout)
        int nb = 0;
        const size_t l1 = sizeof(float2) * ((size_t)N * TZ + N + 2 * H + 32 * LINE_LDS) + sizeof(Roles) +
                          2 * 4 * 64 * sizeof(double);
        (void)hipFuncSetAttribute((const void *)plane_yz2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
        g_yz.resident = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, plane_yz2_kernel, WG2, l1) == hipSuccess &&
                         nb >= 1) ? 1 : 0;
    }
    if (!g_yz.resident) {
        c21hip_set_error("plane-fused pass Y + Z: a workgroup does not fit a CU on this device");
        return C21CM_VALUE_ERROR;
    }
    YzArgs a{};
    const long nlines = (long)nx * ny;
    a.main[0] = reinterpret_cast<const float2 *>(delta_work);
    a.main[1] = reinterpret_cast<const float2 *>(stars_work);
    a.nyq[0] = a.main[0] + nlines * H;
    a.nyq[1] = a.main[1] + nlines * H;
    a.ring = g_yz.ring;
    a.first_cross = first_cross;
    a.partials = partials;
    a.rhocrit_omb = rhocrit_omb;
    a.ion_eff = ion_eff;
    a.f_limit = f_limit;
    a.mass_dep_zeta = mass_dep_zeta;
    a.r_index = r_index;
    if (store_all < 0) {  // as the separate pass Z: C21CM_MASK_STORE_ALL=1 writes every mask byte back
        const char *e = getenv("C21CM_MASK_STORE_ALL");
        store_all = (e && e[0] == '1') ? 1 : 0;
    }
    a.store_all = store_all;
    {
        const char *e = getenv("C21CM_YZ_FORCE_SAFE");
        a.force_safe = (e && e[0] == '1') ? 1 : 0;
    }
    a.sync = g_yz.sync;
    a.status = g_yz.status;
#if C21X_YZ_PROF
    a.prof = (unsigned long long *)c21hip_ws(252, 256 * 8 * sizeof(unsigned long long));
#endif
    const size_t lds = sizeof(float2) * ((size_t)N * TZ + N + 2 * H + 32 * LINE_LDS) + 4 * sizeof(double) + 16;
    if (!g_yz.attr_done) {
        if (hipFuncSetAttribute((const void *)plane_yz_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            c21hip_set_error("plane-fused pass Y + Z: %zu bytes of LDS refused", lds);
            return C21CM_IO_ERROR;
        }
        g_yz.attr_done = 1;
    }
    const size_t lds2 = sizeof(float2) * ((size_t)N * TZ + N + 2 * H + 32 * LINE_LDS) + sizeof(Roles) +
                        2 * 4 * 64 * sizeof(double);
    const char *mode = getenv("C21CM_YZ");
    const bool split = mode && mode[0] == '2';
    if (split && !g_yz.attr2_done) {
        if (hipFuncSetAttribute((const void *)plane_yz2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds2) != hipSuccess) {
            c21hip_set_error("plane-fused pass Y + Z: %zu bytes of LDS refused", lds2);
            return C21CM_IO_ERROR;
        }
        g_yz.attr2_done = 1;
    }
    if (hipMemsetAsync(a.sync, 0, sizeof(YzSync), stream) != hipSuccess) return C21CM_IO_ERROR;
    void *kt = c21hip_ktime_begin(11, stream);
    if (split)
        hipLaunchKernelGGL(plane_yz2_kernel, dim3(8 * GROUP), dim3(WG2), lds2, stream, a, tw, twH, tw);
    else
        hipLaunchKernelGGL(plane_yz_kernel, dim3(8 * GROUP), dim3(WG), lds, stream, a, tw, twH, tw);
    c21hip_ktime_end(kt);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) {
        c21hip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__);
        return C21CM_IO_ERROR;
    }
    return 0;
}

// After a synchronisation: bit 0 = some launch found a workgroup off its XCD (the payload then went
// write-through: slower, still correct), bit 1 = a wait timed out (results invalid).  Clears the flags.
extern "C" int c21hip_plane_yz_status(void *stream_) {
    if (!g_yz.status) return 0;
    YzSync h;
    if (hipMemcpyAsync(&h, g_yz.status, sizeof(YzSync), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream_) != hipSuccess)
        return -1;
    const int flags = (h.mismatch ? 1 : 0) | (h.timeout ? 2 : 0);
    if (flags && hipMemsetAsync(g_yz.status, 0, sizeof(YzSync), (hipStream_t)stream_) != hipSuccess) return -1;
    return flags;
}

// C21X_YZ_PROF builds: the per-phase ticks (100 MHz) of the last launch, [256 workgroups][8]; else -1
extern "C" int c21hip_plane_yz_profile(unsigned long long *out2048) {
#if C21X_YZ_PROF
    void *p = c21hip_ws(252, 256 * 8 * sizeof(unsigned long long));
    if (!p || hipMemcpy(out2048, p, 256 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return 0;
#else
    (void)out2048;
    return -1;
#endif
}
#endif  // C21CM_EXPERIMENTAL
