// perturb_kernels.hip -- ComputePerturbedField sweeps on MI355X.
//
// reference loops being replaced:
//   move_grid_masses + do_cic_interpolation_double   src/py21cmfast/src/map_mass.c:23-60,146-208
//   double -> padded float, normalise_delta_grid      src/py21cmfast/src/PerturbedField.c:115-128,180-210
//   /N, clip, copy-out                                PerturbedField.c:251-276,450-464
//   compute_perturbed_velocities k-space multiply     PerturbedField.c:320-350
//   resample_index gathers                            PerturbedField.c:162-177,367-383
//
// The mass deposit is the only non-streaming kernel of the whole path.  One thread moves
// one hi-res particle: it gathers the displacement from the (coarser or equal) velocity
// grids at the nearest index, and deposits 1 + delta*D_i onto the 8 neighbouring cells
// of the output grid with fp64 atomics, like the reference's `#pragma omp atomic` on a
// double grid.  MI355X executes `global_atomic_add_f64` in the L2, so no CAS loop is
// involved (unsafeAtomicAdd).  Threads of a wavefront walk z fastest: their 8 targets
// fall into a handful of cache lines of the output grid, which keeps the atomic traffic
// inside L2; summation order is the only non-determinism (1e-16 relative).
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>
#include <cstdlib>

#include "c21hip.h"
#include "c21cm_abi.h"
#include "c21cm_grid.h"
#include "fcoll_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(size_t work_items) {
    size_t b = (work_items + kBlock - 1) / kBlock;
    if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

#define LAUNCH_CHECK()                                                                  \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            c21hip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                       \
            return C21CM_IO_ERROR;                                                      \
        }                                                                               \
    } while (0)

__device__ __forceinline__ int wrap_idx(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

struct CicParams {
    int dens_dim[3], vel_dim[3], out_dim[3];
    double dim_ratio_vel, dim_ratio_out;
    double vdf[3], vdf2[3];  // displacement factors, map_mass.c:165-173
    double init_growth;
    int lpt2;
};

__global__ void __launch_bounds__(kBlock)
cic_scatter_kernel(CicParams p, const float *__restrict__ dens, const float *__restrict__ vx,
                   const float *__restrict__ vy, const float *__restrict__ vz,
                   const float *__restrict__ v2x, const float *__restrict__ v2y,
                   const float *__restrict__ v2z, double *__restrict__ out) {
    const size_t total = (size_t)p.dens_dim[0] * p.dens_dim[1] * p.dens_dim[2];
    const size_t plane = (size_t)p.dens_dim[1] * p.dens_dim[2];
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const int i = (int)(t / plane);
        const size_t rem = t - (size_t)i * plane;
        const int j = (int)(rem / (size_t)p.dens_dim[2]);
        const int k = (int)(rem - (size_t)j * p.dens_dim[2]);
        const int src[3] = {i, j, k};
        // resample_index + wrap_coord (indexing.h:110-114)
        int ip[3];
#pragma unroll
        for (int a = 0; a < 3; a++)
            ip[a] = wrap_idx((int)((double)src[a] * p.dim_ratio_vel + 0.5), p.vel_dim[a]);
        const size_t vi =
            (size_t)ip[2] + (size_t)p.vel_dim[2] * ((size_t)ip[1] + (size_t)p.vel_dim[1] * ip[0]);
        const float v[3] = {vx[vi], vy[vi], vz[vi]};
        float v2[3] = {0.f, 0.f, 0.f};
        if (p.lpt2) {
            v2[0] = v2x[vi];
            v2[1] = v2y[vi];
            v2[2] = v2z[vi];
        }
        int i0[3], i1[3];
        double w0[3], w1[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            double pos = (double)src[a];
            pos += (double)v[a] * p.vdf[a];
            if (p.lpt2) pos -= (double)v2[a] * p.vdf2[a];
            pos *= p.dim_ratio_out;
            const double fl = floor(pos);
            const int ipos = (int)fl;
            const double dist = pos - (double)ipos;
            i0[a] = wrap_idx(ipos, p.out_dim[a]);
            i1[a] = wrap_idx(ipos + 1, p.out_dim[a]);
            w0[a] = 1. - dist;
            w1[a] = dist;
        }
        const double mass = 1.0 + (double)dens[t] * p.init_growth;
        const size_t sy = (size_t)p.out_dim[2], sx = (size_t)p.out_dim[1] * p.out_dim[2];
        const size_t bx[2] = {(size_t)i0[0] * sx, (size_t)i1[0] * sx};
        const size_t by[2] = {(size_t)i0[1] * sy, (size_t)i1[1] * sy};
        const size_t bz[2] = {(size_t)i0[2], (size_t)i1[2]};
        const double wx[2] = {w0[0], w1[0]}, wy[2] = {w0[1], w1[1]}, wz[2] = {w0[2], w1[2]};
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int a = 0; a < 2; a++)
                    unsafeAtomicAdd(out + bx[a] + by[b] + bz[c], mass * (wx[a] * wy[b] * wz[c]));
    }
}

// The same deposit with an LDS accumulation tile.  A workgroup owns a brick of SB[0] x SB[1] x
// SB[2] source cells; their displaced positions mostly land within `halo` output cells of the
// brick's own footprint, so the 8 weights of a particle go to an fp64 tile in LDS (ds_add_f64)
// and the tile is flushed once with one global atomic per touched cell: with DIM = 2 HII_DIM
// that is ~30x fewer global atomics, which at 38 G/s were 97 % of ComputePerturbedField.
// Particles displaced beyond the halo take the direct global path, so the result does not
// depend on the halo (fp64 summation order aside, as before).
struct CicTileParams {
    CicParams c;
    int sb[3];       // brick size in source cells
    int nb[3];       // bricks per axis
    int td[3];       // tile extent in output cells (incl. halo and the +1 CIC neighbour)
    int halo;
    int zstride4;    // lanes four source cells apart along z (see the kernel)
    int sb_shift[2]; // log2 of sb[1] * sb[2] and of sb[2] when both are powers of two, else -1
    // NV >= 2 (ComputeHaloBox, map_mass.c:214-344): per-cell values from NV ln-tables of
    // delta = density * growth; table_dev = [NV][C21CM_NDELTA_TABLE] floats (ln N_ion, ln SFRD
    // and, with USE_TS_FLUCT, ln X-ray emissivity)
    double growth, tab_min, tab_width, pref[3];
};

// NV = 1: the deposited value is the particle mass 1 + delta * D_init (PerturbedField).
// NV = 2, 3: that many values per source cell, exp(lerp(table_v, delta * D)) * prefactor_v
// (HaloBox n_ion, halo_sfr and halo_xray), accumulated in NV tiles / NV output grids.
template <int NV>
__global__ void __launch_bounds__(kBlock)
cic_scatter_tiled_kernel(CicTileParams q, const float *__restrict__ dens,
                         const float *__restrict__ vx, const float *__restrict__ vy,
                         const float *__restrict__ vz, const float *__restrict__ v2x,
                         const float *__restrict__ v2y, const float *__restrict__ v2z,
                         const float *__restrict__ table_dev, double *__restrict__ out,
                         double *__restrict__ out_b, double *__restrict__ out_c) {
    extern __shared__ double tile[];
    const CicParams &p = q.c;
    const int tcells = q.td[0] * q.td[1] * q.td[2];
    float *tab = reinterpret_cast<float *>(tile + NV * tcells);  // NV >= 2: [NV][NDELTA]
    if (NV >= 2) {
        for (int t = threadIdx.x; t < NV * C21CM_NDELTA_TABLE; t += kBlock) tab[t] = table_dev[t];
    }
    double *const outs[3] = {out, out_b, out_c};
    const size_t sy = (size_t)p.out_dim[2], sx = (size_t)p.out_dim[1] * p.out_dim[2];
    const int n_bricks = q.nb[0] * q.nb[1] * q.nb[2];
    const int per_brick = q.sb[0] * q.sb[1] * q.sb[2];
    for (int brick = blockIdx.x; brick < n_bricks; brick += gridDim.x) {
        const int b0 = brick / (q.nb[1] * q.nb[2]);
        const int b1 = (brick / q.nb[2]) % q.nb[1];
        const int b2 = brick % q.nb[2];
        const int s0[3] = {b0 * q.sb[0], b1 * q.sb[1], b2 * q.sb[2]};
        int t0[3];  // output coordinate (unwrapped) of tile cell 0
#pragma unroll
        for (int a = 0; a < 3; a++) t0[a] = (int)floor((double)s0[a] * p.dim_ratio_out) - q.halo;
        for (int c = threadIdx.x; c < NV * tcells; c += kBlock) tile[c] = 0.;
        __syncthreads();
        // C21CM_CIC_STRIDE=4 (experiment, off): lanes of one instruction take particles FOUR source
        // cells apart along z (a thread walks its four neighbours one after the other), so that
        // neighbouring particles -- which deposit into the same output cells -- are not in the same
        // LDS atomic instruction.  Not faster: the atomics are not what bounds the kernel.
        const int zs = (q.zstride4 && q.sb[2] % 4 == 0 && per_brick % (4 * kBlock) == 0) ? 4 : 1;
        for (int e4 = threadIdx.x; e4 < per_brick; e4 += kBlock) {
            // e4 = (group g, member j) with j slowest inside a sweep of the workgroup
            const int sweep = e4 / kBlock;              // kBlock particles per sweep
            const int j = sweep % zs;
            const int g = (sweep / zs) * kBlock + (int)threadIdx.x;  // group index among per_brick / zs
            const int e = (zs == 1) ? e4 : (g / (q.sb[2] / 4)) * q.sb[2] + (g % (q.sb[2] / 4)) * 4 + j;
            if (e >= per_brick) continue;
            // (the kernel is bound by its instruction count, not by the LDS atomics -- ds_add_f64 runs
            // at 7 per clock and CU, tools/scratch/lds_atomic_rate.hip -- so the index arithmetic
            // avoids runtime integer divisions where the brick is a power of two)
            int l0, l1, l2;
            if (q.sb_shift[0] >= 0) {
                l0 = e >> q.sb_shift[0];
                l1 = (e >> q.sb_shift[1]) & (q.sb[1] - 1);
                l2 = e & (q.sb[2] - 1);
            } else {
                l0 = e / (q.sb[1] * q.sb[2]);
                l1 = (e / q.sb[2]) % q.sb[1];
                l2 = e % q.sb[2];
            }
            const int src[3] = {s0[0] + l0, s0[1] + l1, s0[2] + l2};
            if (src[0] >= p.dens_dim[0] || src[1] >= p.dens_dim[1] || src[2] >= p.dens_dim[2])
                continue;  // ragged last brick
            const size_t t = (size_t)src[2] +
                             (size_t)p.dens_dim[2] * ((size_t)src[1] + (size_t)p.dens_dim[1] * src[0]);
            int ip[3];
#pragma unroll
            for (int a = 0; a < 3; a++)
            {  // 0 <= src < dens_dim, so the rounded index is in [0, vel_dim]: one conditional wraps it
                ip[a] = (int)((double)src[a] * p.dim_ratio_vel + 0.5);
                if (ip[a] >= p.vel_dim[a]) ip[a] -= p.vel_dim[a];
            }
            const size_t vi = (size_t)ip[2] +
                              (size_t)p.vel_dim[2] * ((size_t)ip[1] + (size_t)p.vel_dim[1] * ip[0]);
            const float v[3] = {vx[vi], vy[vi], vz[vi]};
            float v2[3] = {0.f, 0.f, 0.f};
            if (p.lpt2) {
                v2[0] = v2x[vi];
                v2[1] = v2y[vi];
                v2[2] = v2z[vi];
            }
            int ipos[3];
            double w0[3], w1[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double pos = (double)src[a];
                pos += (double)v[a] * p.vdf[a];
                if (p.lpt2) pos -= (double)v2[a] * p.vdf2[a];
                pos *= p.dim_ratio_out;
                const double fl = floor(pos);
                ipos[a] = (int)fl;
                const double dist = pos - (double)ipos[a];
                w0[a] = 1. - dist;
                w1[a] = dist;
                const int rel = ipos[a] - t0[a];
                inside = inside && rel >= 0 && rel + 1 < q.td[a];
            }
            double val[NV];
            if (NV == 1) {
                val[0] = 1.0 + (double)dens[t] * p.init_growth;
            } else {
                const double curr_dens = (double)dens[t] * q.growth;  // map_mass.c:283
#pragma unroll
                for (int v = 0; v < NV; v++)
                    val[v] = exp(eval_table_f(curr_dens, q.tab_min, q.tab_width,
                                              tab + v * C21CM_NDELTA_TABLE)) * q.pref[v];
            }
            const double wx[2] = {w0[0], w1[0]}, wy[2] = {w0[1], w1[1]}, wz[2] = {w0[2], w1[2]};
            if (inside) {
                const int base = ((ipos[0] - t0[0]) * q.td[1] + (ipos[1] - t0[1])) * q.td[2] +
                                 (ipos[2] - t0[2]);
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const double w = wx[a] * wy[b] * wz[c];
#pragma unroll
                            for (int v = 0; v < NV; v++)
                                __hip_atomic_fetch_add(
                                    &tile[v * tcells + base + (a * q.td[1] + b) * q.td[2] + c],
                                    val[v] * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
            } else {
                size_t bx[2], by[2], bz[2];
                bx[0] = (size_t)wrap_idx(ipos[0], p.out_dim[0]) * sx;
                bx[1] = (size_t)wrap_idx(ipos[0] + 1, p.out_dim[0]) * sx;
                by[0] = (size_t)wrap_idx(ipos[1], p.out_dim[1]) * sy;
                by[1] = (size_t)wrap_idx(ipos[1] + 1, p.out_dim[1]) * sy;
                bz[0] = (size_t)wrap_idx(ipos[2], p.out_dim[2]);
                bz[1] = (size_t)wrap_idx(ipos[2] + 1, p.out_dim[2]);
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            const double w = wx[a] * wy[b] * wz[c];
#pragma unroll
                            for (int v = 0; v < NV; v++)
                                unsafeAtomicAdd(outs[v] + bx[a] + by[b] + bz[c], val[v] * w);
                        }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < tcells; c += kBlock) {
            double tv[NV];
            bool any = false;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                tv[v] = tile[v * tcells + c];
                any = any || tv[v] != 0.;
            }
            if (any) {
                const int c2 = c % q.td[2];
                const int c1 = (c / q.td[2]) % q.td[1];
                const int c0 = c / (q.td[1] * q.td[2]);
                const size_t o = (size_t)wrap_idx(t0[0] + c0, p.out_dim[0]) * sx +
                                 (size_t)wrap_idx(t0[1] + c1, p.out_dim[1]) * sy +
                                 (size_t)wrap_idx(t0[2] + c2, p.out_dim[2]);
#pragma unroll
                for (int v = 0; v < NV; v++) unsafeAtomicAdd(outs[v] + o, tv[v]);
            }
        }
        __syncthreads();
    }
}

// ---- the deposit per VELOCITY cell (round 4) --------------------------------------------------
// With DIM = F * HII_DIM the F^3 particles whose resample_index is velocity cell m -- sources
// F m + lo + j, j < F, lo = -(F / 2): (int)(i / F + 0.5) == m, checked on the host for every i --
// share ONE displacement.  A thread owns a velocity cell: three (six) velocity loads and one
// displacement product per axis instead of F^3 of each, F positions / floors per axis instead of
// F^3, the F^3 densities requested together, and lanes of one ds_add_f64 are neighbouring
// velocity cells, whose targets are DIFFERENT tile cells (neighbouring particles, half an output
// cell apart, collide on the same address).  Positions, floors and weights are the reference's
// doubles (map_mass.c:23-33,180-196); the cell's 8 F^3 terms curr_dens * wx * wy * wz are summed
// in registers onto the 27 tile cells they reach before they go to LDS, i.e. in another order and
// grouping than the reference's atomics (which have no defined order either): 1e-16 relative on a
// double grid that is rounded to float next.  Cells with a particle beyond the tile are queued
// in LDS and leave through global atomics afterwards, so a wavefront with one far-flung lane no
// longer drags all its lanes through the slow path.
struct CicCellParams {
    CicParams c;
    int lo;           // first source offset of a velocity cell: src = F m + lo + j
    int vb[3];        // brick in velocity cells (powers of two)
    int vb_shift[2];  // log2(vb[1] * vb[2]), log2(vb[2])
    int nb[3];        // bricks per axis
    int td[3];        // tile extent in output cells
    int halo, lead;   // lead = 1 when lo < 0: the brick's first particles sit in output cell m0 - 1
};

// FIXED (round 5): the masses accumulate as 64-bit FIXED-POINT integers (scale 2^44) in the LDS tile and in
// the grid -- integer additions commute, so the deposit no longer depends on the order in which the
// hardware serves the atomics: two calls give the same bits (the fp64 atomics of rounds 1-4 did not:
// about one ComputePerturbedField in twelve differed from the previous one in the last bit of a cell,
// which was enough to move a filtered extremum and with it a whole f_coll table).  A term is rounded to
// 2^-44 = 5.7e-14 of the mean particle mass, a cell collects a few hundred of them: 1e-12 relative on a
// grid that is rounded to float (6e-8) next, i.e. the float differs from the fp64 sum's in about one cell
// in 1e5 (at 2^36 it was one in 500, which a 40^3 test noticed).  A cell holds up to 2^19 = 524288
// particle masses before the 63 bits overflow.  The reference's OpenMP atomics have no defined order
// either (map_mass.c:197-206).
// (the integers wrap silently where the fp64 atomics carried NaN / Inf through to the output's non-finite
//  checks.  A non-finite particle mass, or a cell's masses beyond what 63 bits hold, raise this flag; the launcher
//  clears it, c21hip_cic_fixed_status reads it back and the driver turns it into C21CM_INFINITY_OR_NAN_ERROR.)
__device__ int g_cic_fixed_bad;
constexpr double kFixScale = 17592186044416.;       // 2^44
constexpr double kFixMagic = 6755399441055744.;     // 1.5 * 2^52: x + magic holds rint(x) in its low bits
constexpr long long kFixMagicBits = 0x4338000000000000LL;
constexpr double kFixFastLimit = 128.;              // |term| 2^44 < 2^51: the magic-number conversion is exact
__device__ __forceinline__ long long to_fixed_fast(double term) {  // one fma and an integer subtraction
    return __double_as_longlong(fma(term, kFixScale, kFixMagic)) - kFixMagicBits;
}
__device__ __forceinline__ long long to_fixed(double term) {
    if (fabs(term) < kFixFastLimit) return to_fixed_fast(term);
    return __double2ll_rn(term * kFixScale);
}

template <int F, bool LPT2, int DIAG, bool FIXED>
__global__ void __launch_bounds__(kBlock)
cic_cell_kernel(CicCellParams q, const float *__restrict__ dens, const float *__restrict__ vx,
                const float *__restrict__ vy, const float *__restrict__ vz,
                const float *__restrict__ v2x, const float *__restrict__ v2y,
                const float *__restrict__ v2z, double *__restrict__ out) {
    extern __shared__ double tile[];
    __shared__ int nq;
    const CicParams &p = q.c;
    const int tcells = q.td[0] * q.td[1] * q.td[2];
    const int per_brick = q.vb[0] * q.vb[1] * q.vb[2];
    int *queue = reinterpret_cast<int *>(tile + tcells);  // per_brick entries
    const size_t sy = (size_t)p.out_dim[2], sx = (size_t)p.out_dim[1] * p.out_dim[2];
    const size_t d2 = (size_t)p.dens_dim[2], d1 = (size_t)p.dens_dim[1];
    const int n_bricks = q.nb[0] * q.nb[1] * q.nb[2];
    constexpr int F3 = F * F * F;
    for (int brick = blockIdx.x; brick < n_bricks; brick += gridDim.x) {
        const int b0 = brick / (q.nb[1] * q.nb[2]);
        const int b1 = (brick / q.nb[2]) % q.nb[1];
        const int b2 = brick % q.nb[2];
        const int m0[3] = {b0 * q.vb[0], b1 * q.vb[1], b2 * q.vb[2]};
        const int t0[3] = {m0[0] - q.lead - q.halo, m0[1] - q.lead - q.halo,
                           m0[2] - q.lead - q.halo};
        for (int c = threadIdx.x; c < tcells; c += kBlock) tile[c] = 0.;
        if (threadIdx.x == 0) nq = 0;
        __syncthreads();
        for (int e = threadIdx.x; e < per_brick; e += kBlock) {
            const int m[3] = {m0[0] + (e >> q.vb_shift[0]),
                              m0[1] + ((e >> q.vb_shift[1]) & (q.vb[1] - 1)),
                              m0[2] + (e & (q.vb[2] - 1))};
            if (m[0] >= p.vel_dim[0] || m[1] >= p.vel_dim[1] || m[2] >= p.vel_dim[2]) continue;
            const size_t vi = (size_t)m[2] +
                              (size_t)p.vel_dim[2] * ((size_t)m[1] + (size_t)p.vel_dim[1] * m[0]);
            const float v[3] = {vx[vi], vy[vi], vz[vi]};
            float v2[3] = {0.f, 0.f, 0.f};
            if (LPT2) {
                v2[0] = v2x[vi];
                v2[1] = v2y[vi];
                v2[2] = v2z[vi];
            }
            int sw[3][F];  // wrapped source index (m = 0 owns the last source planes)
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int j = 0; j < F; j++) {
                    const int s = F * m[a] + q.lo + j;
                    sw[a][j] = s < 0 ? s + p.dens_dim[a] : s;
                }
            float dn[F][F][F];
#pragma unroll
            for (int jx = 0; jx < F; jx++)
#pragma unroll
                for (int jy = 0; jy < F; jy++) {
                    const size_t row = ((size_t)sw[0][jx] * d1 + (size_t)sw[1][jy]) * d2;
#pragma unroll
                    for (int jz = 0; jz < F; jz++) dn[jx][jy][jz] = dens[row + sw[2][jz]];
                }
            int rel[3][F];
            double w1[3][F];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double disp = (double)v[a] * p.vdf[a];
                const double disp2 = LPT2 ? (double)v2[a] * p.vdf2[a] : 0.;
#pragma unroll
                for (int j = 0; j < F; j++) {
                    double pos = (double)sw[a][j];  // map_mass.c:180-196
                    pos += disp;
                    if (LPT2) pos -= disp2;
                    pos *= p.dim_ratio_out;
                    const double fl = floor(pos);
                    const int ip = (int)fl;
                    w1[a][j] = pos - (double)ip;
                    int r = ip - t0[a];
                    if (r >= p.out_dim[a]) r -= p.out_dim[a];
                    else if (r < 0) r += p.out_dim[a];
                    rel[a][j] = r;
                    inside = inside && r >= 0 && r + 1 < q.td[a] &&
                             (unsigned)(r - rel[a][0]) <= 1u;
                }
            }
            if (!inside) {
                queue[atomicAdd(&nq, 1)] = e;
                continue;
            }
            // The F particles of an axis are 1 / F output cells apart: rel[a][j] - rel[a][0] is 0 or
            // 1 (part of `inside`), so the cell's 8 F^3 terms land in 3 x 3 x 3 tile cells.  Per
            // axis a particle's weights become a 3-vector (w0, w1, 0) or (0, w0, w1), and the sums
            // are formed z first, then y, then x: 27 LDS atomics per velocity cell instead of
            // 8 F^3 (the atomics were 3.4 of the kernel's 5.9 ms at DIM = 1024).
            double W[3][F][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int j = 0; j < F; j++) {
                    const double wlo = 1. - w1[a][j], whi = w1[a][j];
                    if (j == 0) {
                        W[a][j][0] = wlo;
                        W[a][j][1] = whi;
                        W[a][j][2] = 0.;
                    } else {
                        const bool sh = rel[a][j] != rel[a][0];
                        W[a][j][0] = sh ? 0. : wlo;
                        W[a][j][1] = sh ? wlo : whi;
                        W[a][j][2] = sh ? whi : 0.;
                    }
                }
            double acc[3][3][3];
            float dmax = 0.f;  // FIXED: bound of the cell's particle masses
            if constexpr (FIXED) {
#pragma unroll
                for (int jx = 0; jx < F; jx++)
#pragma unroll
                    for (int jy = 0; jy < F; jy++)
#pragma unroll
                        for (int jz = 0; jz < F; jz++) dmax = fmaxf(dmax, fabsf(dn[jx][jy][jz]));
                float dsum = 0.f;  // (fmaxf drops a NaN, a sum keeps it; Inf stays Inf)
#pragma unroll
                for (int jx = 0; jx < F; jx++)
#pragma unroll
                    for (int jy = 0; jy < F; jy++)
#pragma unroll
                        for (int jz = 0; jz < F; jz++) dsum += fabsf(dn[jx][jy][jz]);
                // 2^18 mean particle masses in ONE velocity cell's particles: no physical field gets near it
                if (!(dsum < 3.0e38f) || (1.0 + (double)dmax * fabs(p.init_growth)) * (double)F3 > 262144.)
                    atomicOr(&g_cic_fixed_bad, 1);
            }
            const bool small = (1.0 + (double)dmax * fabs(p.init_growth)) * (double)F3 < kFixFastLimit;
#pragma unroll
            for (int jx = 0; jx < F; jx++) {
                double Y[3][3];
#pragma unroll
                for (int jy = 0; jy < F; jy++) {
                    double Z[3];
#pragma unroll
                    for (int jz = 0; jz < F; jz++) {
                        const double mass = 1.0 + (double)dn[jx][jy][jz] * p.init_growth;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (jz == 0) Z[c] = (c == 2) ? 0. : mass * W[2][0][c];
                            else Z[c] = fma(mass, W[2][jz][c], Z[c]);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < 3; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (jy == 0) Y[b][c] = (b == 2) ? 0. : W[1][0][b] * Z[c];
                            else Y[b][c] = fma(W[1][jy][b], Z[c], Y[b][c]);
                        }
                }
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (jx == 0) acc[a][b][c] = (a == 2) ? 0. : W[0][0][a] * Y[b][c];
                            else acc[a][b][c] = fma(W[0][jx][a], Y[b][c], acc[a][b][c]);
                        }
            }
            const int base = (rel[0][0] * q.td[1] + rel[1][0]) * q.td[2] + rel[2][0];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        if (F == 1 && (a == 2 || b == 2 || c == 2)) continue;
                        const double term = acc[a][b][c];
                        if (DIAG & 1) {  // diagnostic build only: no LDS atomics
                            if (term == 1.2345e300) tile[0] = term;
                        } else if (term != 0.) {  // a third of the 27 are empty: -0.35 ms of 3.7
                            if constexpr (FIXED)
                                // (a term is at most the F^3 particle masses of the cell: `small` decided once
                                //  per cell; the slow conversion is there for IC cells whose F^3 masses exceed 128 only)
                                __hip_atomic_fetch_add(
                                    reinterpret_cast<long long *>(tile) + base + (a * q.td[1] + b) * q.td[2] + c,
                                    small ? to_fixed_fast(term) : to_fixed(term), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_WORKGROUP);
                            else
                                __hip_atomic_fetch_add(&tile[base + (a * q.td[1] + b) * q.td[2] + c], term,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
        }
        __syncthreads();
        // queued cells: one particle per thread through the global path
        const int nqueued = nq;
        for (int t = threadIdx.x; t < nqueued * F3; t += kBlock) {
            const int e = queue[t / F3];
            const int jj = t % F3;
            const int j[3] = {jj / (F * F), (jj / F) % F, jj % F};
            const int m[3] = {m0[0] + (e >> q.vb_shift[0]),
                              m0[1] + ((e >> q.vb_shift[1]) & (q.vb[1] - 1)),
                              m0[2] + (e & (q.vb[2] - 1))};
            const size_t vi = (size_t)m[2] +
                              (size_t)p.vel_dim[2] * ((size_t)m[1] + (size_t)p.vel_dim[1] * m[0]);
            const float v[3] = {vx[vi], vy[vi], vz[vi]};
            float v2[3] = {0.f, 0.f, 0.f};
            if (LPT2) {
                v2[0] = v2x[vi];
                v2[1] = v2y[vi];
                v2[2] = v2z[vi];
            }
            int s[3];
            size_t bo[3][2];
            double w[3][2];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[a] = F * m[a] + q.lo + j[a];
                if (s[a] < 0) s[a] += p.dens_dim[a];
                double pos = (double)s[a];
                pos += (double)v[a] * p.vdf[a];
                if (LPT2) pos -= (double)v2[a] * p.vdf2[a];
                pos *= p.dim_ratio_out;
                const double fl = floor(pos);
                const int ip = (int)fl;
                const double dist = pos - (double)ip;
                w[a][0] = 1. - dist;
                w[a][1] = dist;
                const size_t stride = a == 0 ? sx : (a == 1 ? sy : (size_t)1);
                bo[a][0] = (size_t)wrap_idx(ip, p.out_dim[a]) * stride;
                bo[a][1] = (size_t)wrap_idx(ip + 1, p.out_dim[a]) * stride;
            }
            const double mass =
                1.0 + (double)dens[((size_t)s[0] * d1 + (size_t)s[1]) * d2 + s[2]] * p.init_growth;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const double term = mass * ((w[0][a] * w[1][b]) * w[2][c]);
                        if constexpr (FIXED)
                            atomicAdd(reinterpret_cast<unsigned long long *>(out) + bo[0][a] + bo[1][b] + bo[2][c],
                                      (unsigned long long)to_fixed(term));
                        else
                            unsafeAtomicAdd(out + bo[0][a] + bo[1][b] + bo[2][c], term);
                    }
        }
        // flush: rows of the tile are runs of td[2] cells along z
        if (!(DIAG & 2)) {
            const int rows = q.td[0] * q.td[1];
            for (int c = threadIdx.x; c < tcells; c += kBlock) {
                const double tv = tile[c];  // (FIXED: the integer's bit pattern; 0 is 0 either way)
                if (__double_as_longlong(tv) != 0) {
                    const int row = c / q.td[2];
                    const int c2 = c - row * q.td[2];
                    const int c0 = row / q.td[1];
                    const int c1 = row - c0 * q.td[1];
                    int o0 = t0[0] + c0, o1 = t0[1] + c1, o2 = t0[2] + c2;
                    if (o0 < 0) o0 += p.out_dim[0]; else if (o0 >= p.out_dim[0]) o0 -= p.out_dim[0];
                    if (o1 < 0) o1 += p.out_dim[1]; else if (o1 >= p.out_dim[1]) o1 -= p.out_dim[1];
                    if (o2 < 0) o2 += p.out_dim[2]; else if (o2 >= p.out_dim[2]) o2 -= p.out_dim[2];
                    const size_t o = (size_t)o0 * sx + (size_t)o1 * sy + (size_t)o2;
                    if constexpr (FIXED)
                        atomicAdd(reinterpret_cast<unsigned long long *>(out) + o,
                                  (unsigned long long)__double_as_longlong(tv));
                    else
                        unsafeAtomicAdd(out + o, tv);
                }
            }
            (void)rows;
        }
        __syncthreads();
    }
}

// double grid -> padded float, then (optionally) *= mass_factor; -= 1
__global__ void __launch_bounds__(kBlock)
widen_normalise_kernel(const double *__restrict__ in, float *__restrict__ padded, size_t nlines,
                       int nz, int zpad, int normalise, double mass_factor, int fixed) {
    const size_t total = nlines * (size_t)zpad;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)zpad;
        const int k = (int)(i - line * (size_t)zpad);
        float v = 0.f;
        if (k < nz) {
            const double acc = in[line * (size_t)nz + k];
            // (fixed: the deposit's 2^44 fixed-point integer; exact in a double below 2^53 = 512 particle masses,
            //  rounded to double above: the float next to it does not see that)
            v = (float)(fixed ? (double)__double_as_longlong(acc) * (1. / kFixScale) : acc);  // PerturbedField.c:124
            if (normalise) {
                v = (float)((double)v * mass_factor);  // :201
                v = v - 1.f;                           // :202
            }
        }
        padded[i] = v;
    }
}

// padded real -> (optionally /divisor, clip at -1 + 1e-7) -> dense, with nearest-index
// resampling from a (possibly finer) source grid.  Covers PerturbedField.c:162-177 (hi->lo,
// divide), :251-276 + :450-464 (divide, clip, copy out) and :367-383 (velocity gather).
struct GatherParams {
    int lo_dim[3], src_dim[3];
    int src_zpad;
    double dim_ratio;
    float divisor;  // 0: no division
    int clip;
    int dst_zstride;  // row length of the destination (dense: lo_dim[2], padded: zpad)
};

__global__ void __launch_bounds__(kBlock)
gather_kernel(GatherParams p, const float *__restrict__ src, float *__restrict__ dst) {
    const size_t total = (size_t)p.lo_dim[0] * p.lo_dim[1] * p.lo_dim[2];
    const size_t plane = (size_t)p.lo_dim[1] * p.lo_dim[2];
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const int i = (int)(t / plane);
        const size_t rem = t - (size_t)i * plane;
        const int j = (int)(rem / (size_t)p.lo_dim[2]);
        const int k = (int)(rem - (size_t)j * p.lo_dim[2]);
        const int hi = (int)((double)i * p.dim_ratio + 0.5);
        const int hj = (int)((double)j * p.dim_ratio + 0.5);
        const int hk = (int)((double)k * p.dim_ratio + 0.5);
        float v = src[(size_t)hk + (size_t)p.src_zpad * ((size_t)hj + (size_t)p.src_dim[1] * hi)];
        if (p.divisor != 0.f) v = __fdiv_rn(v, p.divisor);
        if (p.clip && (double)v < -1.0 + 1e-7) v = (float)(-1.0 + 1e-7);
        dst[(size_t)k + (size_t)p.dst_zstride * ((size_t)j + (size_t)p.lo_dim[1] * i)] = v;
    }
}

// padded = (float)(factor * dense): the LINEAR branch, PerturbedField.c:64-80
__global__ void __launch_bounds__(kBlock)
scale_pack_kernel(const float *__restrict__ dense, float *__restrict__ padded, size_t nlines,
                  int nz, int zpad, double factor) {
    const size_t total = nlines * (size_t)zpad;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)zpad;
        const int k = (int)(i - line * (size_t)zpad);
        padded[i] = (k < nz) ? (float)(factor * (double)dense[line * (size_t)nz + k]) : 0.f;
    }
}

// index_to_k: indexing.h:116-120
__device__ __forceinline__ float index_to_k(int idx, double len, int dim) {
    const double buf = (idx <= dim / 2) ? (double)idx : (double)(idx - dim);
    return (float)(buf * 2. * M_PI / len);
}

// v_k = delta_k * (dD/dt / D) * i k_axis / k^2 / N, DC := 0.  PerturbedField.c:320-350
__global__ void __launch_bounds__(kBlock)
velocity_kernel(const float2 *__restrict__ saved, float2 *__restrict__ grid, int nx, int ny,
                int nz, double len_x, double len_y, double len_z, int axis, double dDdt_over_D) {
    const int nzc = nz / 2 + 1;
    const size_t total = (size_t)nx * ny * nzc;
    const double n_r_pixels = (double)((size_t)nx * ny * nz);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (size_t)gridDim.x * kBlock) {
        const size_t line = i / (size_t)nzc;
        const int n_z = (int)(i - line * (size_t)nzc);
        const int n_x = (int)(line / (size_t)ny);
        const int n_y = (int)(line - (size_t)n_x * ny);
        float kvec[3];
        kvec[0] = index_to_k(n_x, len_x, nx);
        kvec[1] = index_to_k(n_y, len_y, ny);
        kvec[2] = index_to_k(n_z, len_z, nz);
        const float k_sq = __fadd_rn(
            __fadd_rn(__fmul_rn(kvec[0], kvec[0]), __fmul_rn(kvec[1], kvec[1])),
            __fmul_rn(kvec[2], kvec[2]));
        float2 v = make_float2(0.f, 0.f);
        if (i != 0) {
            const float2 s = saved[i];
            const double c = dDdt_over_D * (double)kvec[axis] / (double)k_sq / n_r_pixels;
            v.x = (float)(-(double)s.y * c);
            v.y = (float)((double)s.x * c);
        }
        grid[i] = v;
    }
}
}  // namespace

namespace {
// double accumulation grid -> the caller's float grid (optionally a second, scaled copy)
__global__ void __launch_bounds__(kBlock)
narrow_kernel(const double *__restrict__ in, float *__restrict__ out, float *__restrict__ out_scaled,
              double scale, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const float v = (float)in[i];
        out[i] = v;
        if (out_scaled) out_scaled[i] = (float)((double)v * scale);
    }
}

// min and max of a dense float array, seeded with 0 like HaloBox.c:303-304,357-366 when
// seed_zero, else with the first element; one (min, max) pair per workgroup
__global__ void __launch_bounds__(kBlock)
minmax_dense_kernel(const float *__restrict__ a, size_t n, double *__restrict__ pmin,
                    double *__restrict__ pmax) {
    double lo = (double)a[0], hi = lo;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const double v = (double)a[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    __shared__ double slo[kBlock / 64], shi[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, 64));
        hi = fmax(hi, __shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) {
            lo = fmin(lo, slo[w]);
            hi = fmax(hi, shi[w]);
        }
        pmin[blockIdx.x] = lo;
        pmax[blockIdx.x] = hi;
    }
}
}  // namespace

extern "C" int c21hip_narrow(const double *in, float *out, float *out_scaled, double scale,
                             size_t n, void *stream) {
    hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, in,
                       out, out_scaled, scale, n);
    LAUNCH_CHECK();
    return 0;
}

// partials: 2 * 2048 doubles; out2 (device): {min, max}
extern "C" int c21hip_minmax_dense(const float *a, size_t n, double *partials, double *out2,
                                   void *stream) {
    const int blocks = grid_for(n);
    hipLaunchKernelGGL(minmax_dense_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, a,
                       n, partials, partials + kMaxBlocks);
    LAUNCH_CHECK();
    int st = c21hip_reduce_op(partials, blocks, 1, nullptr, out2, stream);
    if (st) return st;
    return c21hip_reduce_op(partials + kMaxBlocks, blocks, 2, nullptr, out2 + 1, stream);
}

namespace {
// brick / tile geometry of the LDS-tiled deposit: bricks of ~8 x 8 x 16 OUTPUT cells worth of
// sources (16 x 16 x 32 source cells at DIM = 2 HII_DIM), 2 output cells of halo
bool tile_setup(const CicParams &p, int nv, CicTileParams &q, size_t *lds, int *blocks) {
    q.c = p;
    q.halo = 2;
    {
        static const int strided = [] {  // measured: 13.4 ms against 12.8 ms consecutive (DIM 1024 -> 512)
            const char *e = getenv("C21CM_CIC_STRIDE");
            return (e && e[0] == '4') ? 1 : 0;
        }();
        q.zstride4 = strided;
    }
    q.sb_shift[0] = q.sb_shift[1] = -1;
    const int ob[3] = {8, 8, 16};
    size_t tcells = 1;
    long n_bricks = 1;
    for (int a = 0; a < 3; a++) {
        int sb = (int)floor(ob[a] / p.dim_ratio_out + 0.5);
        if (sb < 1) sb = 1;
        q.sb[a] = sb < p.dens_dim[a] ? sb : p.dens_dim[a];
        q.nb[a] = (p.dens_dim[a] + q.sb[a] - 1) / q.sb[a];
        q.td[a] = (int)ceil(q.sb[a] * p.dim_ratio_out) + 2 + 2 * q.halo;
        tcells *= (size_t)q.td[a];
        n_bricks *= q.nb[a];
    }
    if (!(q.sb[1] & (q.sb[1] - 1)) && !(q.sb[2] & (q.sb[2] - 1))) {
        q.sb_shift[1] = __builtin_ctz(q.sb[2]);
        q.sb_shift[0] = q.sb_shift[1] + __builtin_ctz(q.sb[1]);
    }
    *lds = tcells * sizeof(double) * nv + (nv >= 2 ? nv * C21CM_NDELTA_TABLE * sizeof(float) : 0);
    *blocks = (int)(n_bricks < 256 * 8 ? n_bricks : 256 * 8);
    return *lds <= (nv == 3 ? 144 : 96) * 1024;  // three tiles need 108 KB of the CU's 160 KB
}

void fill_cic_params(CicParams &p, const int dens_dim[3], const int vel_dim[3], const int out_dim[3],
                     double box_len, double box_len_z, double growth, double init_growth, int lpt2) {
    const double box_size[3] = {box_len, box_len, box_len_z};
    const double d2 = -(3.0 / 7.0) * growth * growth;
    const double id2 = -(3.0 / 7.0) * init_growth * init_growth;
    for (int a = 0; a < 3; a++) {
        p.dens_dim[a] = dens_dim[a];
        p.vel_dim[a] = vel_dim[a];
        p.out_dim[a] = out_dim[a];
        p.vdf[a] = (growth - init_growth) / box_size[a] * dens_dim[a];
        p.vdf2[a] = (d2 - id2) / box_size[a] * dens_dim[a];
    }
    p.dim_ratio_vel = (double)vel_dim[0] / (double)dens_dim[0];
    p.dim_ratio_out = (double)out_dim[0] / (double)dens_dim[0];
    p.init_growth = init_growth;
    p.lpt2 = lpt2;
}
}  // namespace

// ComputeHaloBox deposit (map_mass.c:214-344): two or three values per source cell from the
// ln-tables (out_xray != NULL: three), CIC-deposited at the displaced positions into double grids.
extern "C" int c21hip_halobox_scatter(const float *src_density, const int dens_dim[3],
                                      const float *const vel[3], const float *const vel2[3],
                                      const int vel_dim[3], double *out_nion, double *out_sfr,
                                      double *out_xray, const int out_dim[3], double box_len,
                                      double box_len_z, double growth, double init_growth, int lpt2,
                                      const float *tables_dev, double tab_min, double tab_width,
                                      double pref_nion, double pref_sfr, double pref_xray,
                                      void *stream) {
    CicParams p;
    fill_cic_params(p, dens_dim, vel_dim, out_dim, box_len, box_len_z, growth, init_growth, lpt2);
    CicTileParams q;
    size_t lds;
    int blocks;
    const int nv = out_xray ? 3 : 2;
    if (!tile_setup(p, nv, q, &lds, &blocks)) {
        c21hip_set_error("halobox deposit: tile of %zu bytes does not fit the LDS", lds);
        return C21CM_VALUE_ERROR;
    }
    q.growth = growth;
    q.tab_min = tab_min;
    q.tab_width = tab_width;
    q.pref[0] = pref_nion;
    q.pref[1] = pref_sfr;
    q.pref[2] = pref_xray;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)cic_scatter_tiled_kernel<2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void *)cic_scatter_tiled_kernel<3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        attr_done = true;
    }
    if (nv == 3)
        hipLaunchKernelGGL(cic_scatter_tiled_kernel<3>, dim3(blocks), dim3(kBlock), lds,
                           (hipStream_t)stream, q, src_density, vel[0], vel[1], vel[2],
                           lpt2 ? vel2[0] : nullptr, lpt2 ? vel2[1] : nullptr,
                           lpt2 ? vel2[2] : nullptr, tables_dev, out_nion, out_sfr, out_xray);
    else
        hipLaunchKernelGGL(cic_scatter_tiled_kernel<2>, dim3(blocks), dim3(kBlock), lds,
                           (hipStream_t)stream, q, src_density, vel[0], vel[1], vel[2],
                           lpt2 ? vel2[0] : nullptr, lpt2 ? vel2[1] : nullptr,
                           lpt2 ? vel2[2] : nullptr, tables_dev, out_nion, out_sfr,
                           (double *)nullptr);
    LAUNCH_CHECK();
    return 0;
}

namespace {
// geometry of the per-velocity-cell deposit; false when the grids do not have its structure
// (DIM = F * velocity grid = F * output grid, F <= 4, every source's resample_index as assumed)
bool cell_setup(const CicParams &p, CicCellParams &q, size_t *lds, int *blocks, int *f_out) {
    const int f = p.dens_dim[0] / (p.vel_dim[0] > 0 ? p.vel_dim[0] : 1);
    if (f < 1 || f > 4) return false;
    q.c = p;
    q.lo = -(f / 2);
    q.lead = q.lo < 0 ? 1 : 0;
    // 3 output cells of halo: at a displacement rms of 0.8 cells per axis (z ~ 7 on 1.5 Mpc cells)
    // as fast as 2 (3.75 against 3.70 ms at DIM = 1024 -> 512), at 1.5 cells 5.0 against 12.9 ms
    q.halo = 3;
    if (const char *e = getenv("C21CM_CIC_HALO")) {  // experiment
        const int h = atoi(e);
        if (h >= 1 && h <= 8) q.halo = h;
    }
    int vb[3] = {8, 8, 16};
    if (const char *e = getenv("C21CM_CIC_BRICK")) {  // experiment: "x,y,z" (powers of two)
        int a, b, c;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0 && !(a & (a - 1)) &&
            !(b & (b - 1)) && !(c & (c - 1))) {
            vb[0] = a;
            vb[1] = b;
            vb[2] = c;
        }
    }
    size_t tcells = 1;
    long n_bricks = 1, per_brick = 1;
    for (int a = 0; a < 3; a++) {
        if (p.dens_dim[a] != f * p.vel_dim[a] || p.out_dim[a] != p.vel_dim[a]) return false;
        for (int i = 0; i < p.dens_dim[a]; i++) {  // resample_index + wrap_coord, indexing.h:110-114
            int ip = (int)((double)i * p.dim_ratio_vel + 0.5) % p.vel_dim[a];
            if (ip != ((i - q.lo) / f) % p.vel_dim[a]) return false;
        }
        q.vb[a] = vb[a];
        q.nb[a] = (p.vel_dim[a] + vb[a] - 1) / vb[a];
        q.td[a] = vb[a] + q.lead + 1 + 2 * q.halo;
        if (p.out_dim[a] < q.td[a]) return false;
        tcells *= (size_t)q.td[a];
        n_bricks *= q.nb[a];
        per_brick *= vb[a];
    }
    q.vb_shift[1] = __builtin_ctz(vb[2]);
    q.vb_shift[0] = q.vb_shift[1] + __builtin_ctz(vb[1]);
    *lds = tcells * sizeof(double) + (size_t)per_brick * sizeof(int);
    *blocks = (int)(n_bricks < 256 * 8 ? n_bricks : 256 * 8);
    *f_out = f;
    return *lds <= 96 * 1024;
}

template <int F, int DIAG, bool FIXED = false>
void launch_cell(const CicCellParams &q, size_t lds, int blocks, int lpt2, const float *dens,
                 const float *const vel[3], const float *const vel2[3], double *out,
                 hipStream_t stream) {
    if (lpt2) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)cic_cell_kernel<F, true, DIAG, FIXED>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL((cic_cell_kernel<F, true, DIAG, FIXED>), dim3(blocks), dim3(kBlock), lds, stream,
                           q, dens, vel[0], vel[1], vel[2], vel2[0], vel2[1], vel2[2], out);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)cic_cell_kernel<F, false, DIAG, FIXED>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL((cic_cell_kernel<F, false, DIAG, FIXED>), dim3(blocks), dim3(kBlock), lds,
                           stream, q, dens, vel[0], vel[1], vel[2], (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, out);
    }
}
}  // namespace

// 1: the last fixed-point deposit met a non-finite or absurdly large particle mass (synchronises `stream`)
extern "C" int c21hip_cic_fixed_status(int *bad, void *stream) {
    int v = 0;
    if (hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_cic_fixed_bad), sizeof(int), 0, hipMemcpyDeviceToHost,
                                 (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return C21CM_IO_ERROR;
    }
    *bad = v;
    return 0;
}

extern "C" int c21hip_cic_scatter(const float *hires_density, const int dens_dim[3],
                                  const float *const vel[3], const float *const vel2[3],
                                  const int vel_dim[3], double *out, const int out_dim[3],
                                  double box_len, double box_len_z, double growth,
                                  double init_growth, int lpt2, int *fixed_out, void *stream) {
    CicParams p;
    fill_cic_params(p, dens_dim, vel_dim, out_dim, box_len, box_len_z, growth, init_growth, lpt2);
    const size_t total = (size_t)dens_dim[0] * dens_dim[1] * dens_dim[2];
    if (fixed_out) *fixed_out = 0;
    {
        // C21CM_CIC = cell (default: per velocity cell, LDS tile) | tiled (per particle, LDS tile;
        // rounds 1-3) | direct (global atomics); tiny grids take the direct kernel
        const char *e = getenv("C21CM_CIC");
        const int direct = (e && e[0] == 'd') ? 1 : 0;
        const int tiled = (e && e[0] == 't') ? 1 : 0;
        if (!direct && !tiled && total >= (size_t)1 << 15) {
            CicCellParams q;
            size_t lds;
            int blocks, f;
            if (cell_setup(p, q, &lds, &blocks, &f)) {
#ifdef C21X_CIC_DIAG
                const char *dg = getenv("C21CM_CIC_DIAG");
                const int diag = dg ? atoi(dg) : 0;
                if (f == 2 && diag == 1)
                    launch_cell<2, 1>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else if (f == 2 && diag == 2)
                    launch_cell<2, 2>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else if (f == 2 && diag == 3)
                    launch_cell<2, 3>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else
#endif
                // C21CM_CIC_ACC = fixed (default where the caller can read it: deterministic 64-bit
                // fixed-point accumulation) | double (fp64 atomics, rounds 1-4)
                const char *ea = getenv("C21CM_CIC_ACC");
                if (fixed_out && !(ea && ea[0] == 'd')) {
                    *fixed_out = 1;
                    {
                        const int zero = 0;  // (c21hip_cic_fixed_status reads it back after the deposit)
                        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cic_fixed_bad), &zero, sizeof(int), 0,
                                                     hipMemcpyHostToDevice, (hipStream_t)stream);
                    }
                    if (f == 1)
                        launch_cell<1, 0, true>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                    else if (f == 2)
                        launch_cell<2, 0, true>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                    else if (f == 3)
                        launch_cell<3, 0, true>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                    else
                        launch_cell<4, 0, true>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                    LAUNCH_CHECK();
                    return 0;
                }
                if (f == 1)
                    launch_cell<1, 0>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else if (f == 2)
                    launch_cell<2, 0>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else if (f == 3)
                    launch_cell<3, 0>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                else
                    launch_cell<4, 0>(q, lds, blocks, lpt2, hires_density, vel, vel2, out, (hipStream_t)stream);
                LAUNCH_CHECK();
                return 0;
            }
        }
        if (!direct && total >= (size_t)1 << 15) {
            CicTileParams q;
            size_t lds;
            int blocks;
            if (tile_setup(p, 1, q, &lds, &blocks)) {
                hipLaunchKernelGGL(cic_scatter_tiled_kernel<1>, dim3(blocks), dim3(kBlock), lds,
                                   (hipStream_t)stream, q, hires_density, vel[0], vel[1], vel[2],
                                   lpt2 ? vel2[0] : nullptr, lpt2 ? vel2[1] : nullptr,
                                   lpt2 ? vel2[2] : nullptr, (const float *)nullptr, out,
                                   (double *)nullptr, (double *)nullptr);
                LAUNCH_CHECK();
                return 0;
            }
        }
    }
    hipLaunchKernelGGL(cic_scatter_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, p, hires_density, vel[0], vel[1], vel[2],
                       lpt2 ? vel2[0] : nullptr, lpt2 ? vel2[1] : nullptr,
                       lpt2 ? vel2[2] : nullptr, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_widen_normalise(const double *in, float *padded, int nx, int ny, int nz,
                                      int normalise, double mass_factor, int fixed, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(widen_normalise_kernel, dim3(grid_for(nlines * zpad)), dim3(kBlock), 0,
                       (hipStream_t)stream, in, padded, nlines, nz, zpad, normalise, mass_factor, fixed);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_scale_pack(const float *dense, float *padded, int nx, int ny, int nz,
                                 double factor, void *stream) {
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nlines = (size_t)nx * ny;
    hipLaunchKernelGGL(scale_pack_kernel, dim3(grid_for(nlines * zpad)), dim3(kBlock), 0,
                       (hipStream_t)stream, dense, padded, nlines, nz, zpad, factor);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_gather(const float *src_padded, const int src_dim[3], float *dst,
                             const int lo_dim[3], int dst_padded, float divisor, int clip,
                             void *stream) {
    GatherParams p;
    for (int a = 0; a < 3; a++) {
        p.lo_dim[a] = lo_dim[a];
        p.src_dim[a] = src_dim[a];
    }
    p.src_zpad = 2 * (src_dim[2] / 2 + 1);
    p.dim_ratio = src_dim[0] / (double)lo_dim[0];
    p.divisor = divisor;
    p.clip = clip;
    p.dst_zstride = dst_padded ? 2 * (lo_dim[2] / 2 + 1) : lo_dim[2];
    const size_t total = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       p, src_padded, dst);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_velocity_kspace(const float *saved_c, float *grid_c, int nx, int ny, int nz,
                                      double box_len, double box_len_z, int axis,
                                      double dDdt_over_D, void *stream) {
    const size_t total = (size_t)nx * ny * (nz / 2 + 1);
    hipLaunchKernelGGL(velocity_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const float2 *)saved_c, (float2 *)grid_c, nx, ny, nz,
                       box_len, box_len, box_len_z, axis, dDdt_over_D);
    LAUNCH_CHECK();
    return 0;
}

// ======================================================================================
// ComputeHaloBox with USE_MINI_HALOS (HaloBox.c:245-283,465-516, map_mass.c:285-321)
// ======================================================================================
namespace {
struct TurnoverParams {
    size_t ntot;
    int n_chunks, below_z_heat_max;
    float z;
    double mturn_a_nofb, m_turn, vcb_const, A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb;
};

// get_log10_turnovers.  Upstream's atomic turnover is a running maximum within each OpenMP
// thread's share of the cells (HaloBox.c:481,497), so one workgroup walks one share (libgomp's
// static schedule: the first N mod T shares are one cell longer) 256 cells at a time with a
// wave-level inclusive max-scan and a carry.
__global__ void __launch_bounds__(kBlock)
halobox_turnover_kernel(TurnoverParams m, const float *__restrict__ prev_G12,
                        const float *__restrict__ prev_z_reion, const float *__restrict__ J_21_LW,
                        const float *__restrict__ vcb, float *__restrict__ out_a,
                        float *__restrict__ out_m, double *__restrict__ sums) {
    __shared__ double wmax[kBlock / 64];
    __shared__ double red[2][kBlock / 64];
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t q = m.ntot / m.n_chunks, r = m.ntot % m.n_chunks;
    const size_t start = q * t + ((size_t)t < r ? (size_t)t : r);
    const size_t len = q + ((size_t)t < r ? 1 : 0);
    const double zp1 = 1. + (double)m.z;
    const double mcrit_noLW = 3.314e7 * pow(zp1, -1.5);
    double carry = m.mturn_a_nofb, acc_a = 0., acc_m = 0.;
    for (size_t base = start; base < start + len; base += kBlock) {
        const size_t i = base + threadIdx.x;
        const bool valid = i < start + len;
        double v = -1., M_turn_m = 1.;
        if (valid) {
            float j = 0.f, g = 0.f, zin = 0.f;
            if (m.below_z_heat_max) {
                j = J_21_LW[i];
                g = prev_G12[i];
                zin = prev_z_reion[i];
            }
            const float vc = vcb ? vcb[i] : (float)m.vcb_const;
            M_turn_m = mcrit_noLW * (1.0 + m.A_LW * pow((double)j, m.BETA_LW)) *
                       pow(1.0 + m.A_VCB * (double)vc / m.sigma_vcb, m.BETA_VCB);
            double M_turn_r = 1e-40;
            if (!((double)zin <= 1e-19))
                M_turn_r = 3e9 * pow(2.0 * (double)g, 0.17) * pow(zp1 / 10, -2.1) *
                           pow(1 - pow(zp1 / (1. + (double)zin), 2.0), 2.5);
            v = fmax(M_turn_r, m.m_turn);
            M_turn_m = fmax(M_turn_m, v);
        }
        double s = v;  // inclusive max-scan within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(s, off, 64);
            if (lane >= off) s = fmax(s, o);
        }
        if (lane == 63) wmax[wave] = s;
        __syncthreads();
        double before = carry;
        for (int w = 0; w < wave; w++) before = fmax(before, wmax[w]);
        const double M_turn_a = fmax(before, s);
        double block_max = carry;
        for (int w = 0; w < kBlock / 64; w++) block_max = fmax(block_max, wmax[w]);
        carry = block_max;
        __syncthreads();
        if (valid) {
            const double la = log10(M_turn_a), lm = log10(M_turn_m);
            out_a[i] = (float)la;
            out_m[i] = (float)lm;
            acc_a += la;
            acc_m += lm;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc_a += __shfl_down(acc_a, off, 64);
        acc_m += __shfl_down(acc_m, off, 64);
    }
    if (lane == 0) red[0][wave] = acc_a, red[1][wave] = acc_m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0., b = 0.;
        for (int w = 0; w < kBlock / 64; w++) a += red[0][w], b += red[1][w];
        unsafeAtomicAdd(sums, a);
        unsafeAtomicAdd(sums + 1, b);
    }
}

__device__ __forceinline__ double hb_table_2d(double x, double y, double x_min, double x_width,
                                              double y_min, double y_width,
                                              const float *__restrict__ z_arr) {
    const int x_idx = (int)floor((x - x_min) / x_width);
    const int y_idx = (int)floor((y - y_min) / y_width);
    const double px = (x - (x_min + x_width * (double)x_idx)) / x_width;
    const double py = (y - (y_min + y_width * (double)y_idx)) / y_width;
    const float *r0 = z_arr + (size_t)x_idx * C21CM_NMTURN_TABLE + y_idx;
    const float *r1 = r0 + C21CM_NMTURN_TABLE;
    const double left_edge = (double)r0[0] * (1 - py) + (double)r0[1] * py;
    const double right_edge = (double)r1[0] * (1 - py) + (double)r1[1] * py;
    return left_edge * (1 - px) + right_edge * px;
}

struct HaloMiniParams {
    CicParams c;
    double growth, tab_min, tab_width, mta_min, mta_width, mtm_min, mtm_width, mtf_min, mtf_width;
    double pref_nion, pref_nion_mini, pref_sfr, pref_sfr_mini, pref_xray;
};

// move_grid_galprops with USE_MINI_HALOS: four values per source cell, CIC-deposited with plain
// fp64 global atomics (this branch is bound by its table lookups and is not the tuned path)
__global__ void __launch_bounds__(kBlock)
halobox_scatter_mini_kernel(HaloMiniParams h, const float *__restrict__ dens,
                            const float *__restrict__ vx, const float *__restrict__ vy,
                            const float *__restrict__ vz, const float *__restrict__ v2x,
                            const float *__restrict__ v2y, const float *__restrict__ v2z,
                            const float *__restrict__ mturn_a, const float *__restrict__ mturn_m,
                            const float *__restrict__ tab_sfrd,    // [NDELTA]
                            const float *__restrict__ tab_nion_a,  // [NDELTA][NMTURN] ...
                            const float *__restrict__ tab_nion_m, const float *__restrict__ tab_sfrd_m,
                            const float *__restrict__ tab_xray, double *__restrict__ out_nion,
                            double *__restrict__ out_sfr, double *__restrict__ out_sfr_mini,
                            double *__restrict__ out_xray) {
    const CicParams &p = h.c;
    const size_t total = (size_t)p.dens_dim[0] * p.dens_dim[1] * p.dens_dim[2];
    const size_t plane = (size_t)p.dens_dim[1] * p.dens_dim[2];
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const int i = (int)(t / plane);
        const size_t rem = t - (size_t)i * plane;
        const int j = (int)(rem / (size_t)p.dens_dim[2]);
        const int k = (int)(rem - (size_t)j * p.dens_dim[2]);
        const int src[3] = {i, j, k};
        const float v[3] = {vx[t], vy[t], vz[t]};  // velocities on the density grid (low-res sources)
        float v2[3] = {0.f, 0.f, 0.f};
        if (p.lpt2) v2[0] = v2x[t], v2[1] = v2y[t], v2[2] = v2z[t];
        int i0[3], i1[3];
        double w0[3], w1[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            double pos = (double)src[a];
            pos += (double)v[a] * p.vdf[a];
            if (p.lpt2) pos -= (double)v2[a] * p.vdf2[a];
            pos *= p.dim_ratio_out;
            const int ipos = (int)floor(pos);
            const double dist = pos - (double)ipos;
            i0[a] = wrap_idx(ipos, p.out_dim[a]);
            i1[a] = wrap_idx(ipos + 1, p.out_dim[a]);
            w0[a] = 1. - dist;
            w1[a] = dist;
        }
        const double curr_dens = (double)dens[t] * h.growth;
        const double la = (double)mturn_a[t], lm = (double)mturn_m[t];
        const int idx = (int)floor((curr_dens - h.tab_min) / h.tab_width);
        const double ip1 = (curr_dens - (h.tab_min + h.tab_width * (double)(float)idx)) / h.tab_width;
        const double sfrd = exp((double)tab_sfrd[idx] * (1 - ip1) + (double)tab_sfrd[idx + 1] * ip1);
        const double nion_a = exp(hb_table_2d(curr_dens, la, h.tab_min, h.tab_width, h.mta_min, h.mta_width, tab_nion_a));
        const double nion_m = exp(hb_table_2d(curr_dens, lm, h.tab_min, h.tab_width, h.mtm_min, h.mtm_width, tab_nion_m));
        const double sfrd_m = exp(hb_table_2d(curr_dens, lm, h.tab_min, h.tab_width, h.mtf_min, h.mtf_width, tab_sfrd_m));
        const double val[4] = {nion_a * h.pref_nion + nion_m * h.pref_nion_mini, sfrd * h.pref_sfr,
                               sfrd_m * h.pref_sfr_mini,
                               out_xray ? exp(hb_table_2d(curr_dens, lm, h.tab_min, h.tab_width, h.mtf_min,
                                                          h.mtf_width, tab_xray)) * h.pref_xray
                                        : 0.};
        double *outs[4] = {out_nion, out_sfr, out_sfr_mini, out_xray};
        const size_t sy = (size_t)p.out_dim[2], sx = (size_t)p.out_dim[1] * p.out_dim[2];
        const size_t bx[2] = {(size_t)i0[0] * sx, (size_t)i1[0] * sx};
        const size_t by[2] = {(size_t)i0[1] * sy, (size_t)i1[1] * sy};
        const size_t bz[2] = {(size_t)i0[2], (size_t)i1[2]};
        const double wx[2] = {w0[0], w1[0]}, wy[2] = {w0[1], w1[1]}, wz[2] = {w0[2], w1[2]};
        for (int g = 0; g < 4; g++) {
            if (!outs[g]) continue;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int a = 0; a < 2; a++)
                        unsafeAtomicAdd(outs[g] + bx[a] + by[b] + bz[c], val[g] * (wx[a] * wy[b] * wz[c]));
        }
    }
}
}  // namespace

extern "C" int c21hip_halobox_turnovers(size_t ntot, int n_chunks, int below_z_heat_max,
                                        double redshift, double mturn_a_nofb, double m_turn,
                                        double vcb_const, double A_LW, double BETA_LW, double A_VCB,
                                        double BETA_VCB, double sigma_vcb, const float *prev_G12,
                                        const float *prev_z_reion, const float *J_21_LW,
                                        const float *vcb, float *out_a, float *out_m,
                                        double *sums_dev, void *stream) {
    TurnoverParams m;
    m.ntot = ntot;
    m.n_chunks = n_chunks < 1 ? 1 : n_chunks;
    m.below_z_heat_max = below_z_heat_max;
    m.z = (float)redshift;
    m.mturn_a_nofb = mturn_a_nofb;
    m.m_turn = m_turn;
    m.vcb_const = vcb_const;
    m.A_LW = A_LW, m.BETA_LW = BETA_LW, m.A_VCB = A_VCB, m.BETA_VCB = BETA_VCB;
    m.sigma_vcb = sigma_vcb;
    hipLaunchKernelGGL(halobox_turnover_kernel, dim3(m.n_chunks), dim3(kBlock), 0,
                       (hipStream_t)stream, m, prev_G12, prev_z_reion, J_21_LW, vcb, out_a, out_m,
                       sums_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_halobox_scatter_mini(const float *src_density, const int dim[3],
                                           const float *const vel[3], const float *const vel2[3],
                                           const float *mturn_a, const float *mturn_m,
                                           double *out_nion, double *out_sfr, double *out_sfr_mini,
                                           double *out_xray, double box_len, double box_len_z,
                                           double growth, double init_growth, int lpt2,
                                           const float *tab_sfrd, const float *tab_nion_a,
                                           const float *tab_nion_m, const float *tab_sfrd_m,
                                           const float *tab_xray, const double *ranges,
                                           const double *prefactors, void *stream) {
    HaloMiniParams h;
    fill_cic_params(h.c, dim, dim, dim, box_len, box_len_z, growth, init_growth, lpt2);
    h.growth = growth;
    h.tab_min = ranges[0], h.tab_width = ranges[1];
    h.mta_min = ranges[2], h.mta_width = ranges[3];
    h.mtm_min = ranges[4], h.mtm_width = ranges[5];
    h.mtf_min = ranges[6], h.mtf_width = ranges[7];
    h.pref_nion = prefactors[0], h.pref_nion_mini = prefactors[1], h.pref_sfr = prefactors[2];
    h.pref_sfr_mini = prefactors[3], h.pref_xray = prefactors[4];
    const size_t total = (size_t)dim[0] * dim[1] * dim[2];
    hipLaunchKernelGGL(halobox_scatter_mini_kernel, dim3(grid_for(total)), dim3(kBlock), 0,
                       (hipStream_t)stream, h, src_density, vel[0], vel[1], vel[2], vel2[0], vel2[1],
                       vel2[2], mturn_a, mturn_m, tab_sfrd, tab_nion_a, tab_nion_m, tab_sfrd_m,
                       tab_xray, out_nion, out_sfr, out_sfr_mini, out_xray);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Halo catalogue -> HaloBox grids: move_halo_galprops (map_mass.c:346-476) with
// set_halo_properties (HaloBox.c:62-102) and the per-halo scaling relations
// (scaling_relations.c:277-283,331-500).  One thread per halo: displace the halo with the
// velocities of its Lagrangian cell, evaluate the relations in double, deposit up to five values
// into eight cells each with fp64 atomics (per unit cell volume: upstream scales the float grids
// afterwards).  Upstream adds into FLOAT grids in thread order; the double accumulation here
// agrees with its N_THREADS = 1 result to float rounding of the per-cell sums.
namespace {
constexpr double kSecPerYr = 31556925.9747;  // physconst.s_per_yr

struct HaloDepositParams {
    c21cm_halo_consts c;
    unsigned long long n_halos;
    int vel_dim[3], out_dim[3];
    double cell_size_inv_v;  // vel_dim[0] / BOX_LEN for every axis (:355)
    double vdf, vdf2;        // D - D_i and the 2LPT analogue: Mpc per velocity unit (:368-370)
    double box_size[3];      // pos * out_dim / box_size (:409-411)
    double cell_vol_inv;
    // logarithms of constants of the scaling relations, taken once on the host
    double ln_pivot_upper, ln_tstar_th, ln_s_per_yr, ln_m0_norm;
    double ln_z_norm;  // ln(1.23 * 10^(-0.056 z + 0.064) / 0.05): metallicity over the L_X pivot
    int lpt2;
};

__device__ __forceinline__ double cic_read(const float *__restrict__ box, const size_t idx[8],
                                            const double w[8]) {
    double sum = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) sum += w[c] * (double)box[idx[c]];
    return sum;
}

struct HaloArrays {
    const float *masses, *coords, *star_rng, *sfr_rng, *xray_rng;
    const float *vx, *vy, *vz, *v2x, *v2y, *v2z;
    const float *mturn_a, *mturn_m;
};
struct HaloOutputs {
    double *g[5];  // n_ion, SFR, SFR_mini, L_X, f_esc-weighted SFR; NULL = not wanted
};

struct HaloProps {  // HaloProperties of HaloBox.h, per halo (not per volume)
    double stars, stars_mini, sfr, sfr_mini, xray, n_ion, wsfr, metallicity;
};

// set_halo_properties (HaloBox.c:62-102) with the scaling relations of scaling_relations.c:331-500
__device__ __forceinline__ HaloProps halo_relations(const HaloDepositParams &h, double hmass,
                                                    double r_star, double r_sfr, double r_xray,
                                                    double M_turn_a, double M_turn_m) {
    const c21cm_halo_consts &c = h.c;
    // The power laws below are evaluated as exp(index * ln x) on shared logarithms (one log of
    // the halo mass, one of the stellar mass, one of the SFR) instead of one pow() each: the
    // kernel is bound by its fp64 transcendentals, and the rounding difference (1e-14) is far
    // below the float grids' resolution.
    const double lnM = log(hmass), l10 = lnM - 10. * M_LN10;
    // get_halo_stellarmass (scaling_relations.c:331-400)
    const double adj_star = c.scaling_median ? 0. : c.sigma_star * c.sigma_star / 2.;
    const double s_rng = r_star;
    double f_sample;
    if (c.upper_stellar_turnover && c.alpha_star > c.alpha_upper) {
        const double lp = lnM - h.ln_pivot_upper;
        f_sample = c.fstar_10 * (c.upper_pivot_ratio / (exp(-c.alpha_star * lp) + exp(-c.alpha_upper * lp))) *
                   exp(-M_turn_a / hmass + s_rng * c.sigma_star - adj_star);
    } else {
        f_sample = c.fstar_10 * exp(c.alpha_star * l10 - M_turn_a / hmass + s_rng * c.sigma_star - adj_star);
    }
    if (f_sample > 1.) f_sample = 1.;
    const double stars = f_sample * hmass * c.baryon_ratio;
    double stars_mini = 0.;
    if (c.use_mini_halos) {
        double f_mini = c.fstar_7 * exp(c.alpha_star_mini * (lnM - 7. * M_LN10) - M_turn_m / hmass -
                                        hmass / c.acg_thresh + s_rng * c.sigma_star - adj_star);
        if (f_mini > 1.) f_mini = 1.;
        stars_mini = f_mini * hmass * c.baryon_ratio;
    }
    const double stars_t = stars + stars_mini;
    const double ln_stars = log(stars_t);  // -inf for an empty halo: handled where it is used
    // get_halo_sfr (:402-444): the scatter widens towards low stellar masses
    // A halo so far below its turnover that exp(-M_turn / M) underflows has no stars: upstream
    // then multiplies 0 by exp(r * inf - inf) and writes NaN for deviates r >= 0 (0 for r < 0);
    // here such halos (M* < 1e-290 Msun, where the CPU's product underflows to 0 anyway) get SFR 0.
    double sigma_sfr = 0.;
    if (c.sigma_sfr_lim > 0.) {
        sigma_sfr = c.sigma_sfr_idx * ((ln_stars - 10. * M_LN10) * M_LOG10E) + c.sigma_sfr_lim;
        if (!(sigma_sfr >= c.sigma_sfr_lim) || !(stars_t > 1e-290)) sigma_sfr = c.sigma_sfr_lim;
    }
    const bool starless = !(stars_t > 1e-290);
    const double adj_sfr = c.scaling_median ? 0. : sigma_sfr * sigma_sfr / 2.;
    const double ln_sfr_fac = r_sfr * sigma_sfr - adj_sfr;
    const double sfr_fac = starless ? 0. : exp(ln_sfr_fac);
    const double sfr = stars / (c.t_star * c.t_h) * sfr_fac;
    const double sfr_mini = c.use_mini_halos ? stars_mini / (c.t_star * c.t_h) * sfr_fac : 0.;
    // get_halo_metallicity, get_halo_xray (:446-500)
    double xray = 0., metallicity = 0.;
    if (c.use_xray) {
        const double sfr_t = sfr + sfr_mini;
        double ln_stellar_term = 0.;
        if (stars_t > 0 && sfr_t > 0.) {
            // M0 = 1.28825e10 (SFR s_per_yr)^0.56;  (1 + (M*/M0)^-2.1)^-0.148
            const double ln_sfr_yr = ln_stars - h.ln_tstar_th + ln_sfr_fac + h.ln_s_per_yr;
            const double ln_ratio = ln_stars - (h.ln_m0_norm + 0.56 * ln_sfr_yr);
            ln_stellar_term = -0.148 * log1p(exp(-2.1 * ln_ratio));
        }
        metallicity = 0.05 * exp(h.ln_z_norm + ln_stellar_term);
        double lx_a = c.l_x, lx_m = c.l_x_mini;
        if (c.upper_stellar_turnover) {  // double power law in Z, flat below Z = 0.05 (:277-283)
            const double dpl = 1. / (1. + exp(0.64 * (h.ln_z_norm + ln_stellar_term)));
            lx_a *= dpl, lx_m *= dpl;
        }
        double mu_x = lx_a * (sfr * kSecPerYr);
        if (c.use_mini_halos) mu_x += lx_m * (sfr_mini * kSecPerYr);
        const double adj_x = c.scaling_median ? 0. : c.sigma_xray * c.sigma_xray / 2.;
        xray = mu_x * exp(r_xray * c.sigma_xray - adj_x);
    }
    const double fesc = fmin(c.fesc_10 * exp(c.alpha_esc * l10), 1.);
    const double fesc_mini = c.use_mini_halos ? fmin(c.fesc_7 * exp(c.alpha_esc * (lnM - 7. * M_LN10)), 1.) : 0.;
    const double n_ion = stars * c.pop2_ion * fesc + stars_mini * c.pop3_ion * fesc_mini;
    const double wsfr = sfr * c.pop2_ion * fesc + sfr_mini * c.pop3_ion * fesc_mini;
    HaloProps p = {stars, stars_mini, sfr, sfr_mini, xray, n_ion, wsfr, metallicity};
    return p;
}

// One halo: displaced position (ipos_out: its unwrapped output cell; idx / w: the eight cells and
// CIC weights) and the five values per unit cell volume.  false: the halo was cut (mass 0).
__device__ __forceinline__ bool halo_eval(const HaloDepositParams &h, const HaloArrays &A,
                                          unsigned long long t, int ipos_out[3], size_t idx[8],
                                          double w[8], double val[5]) {
    const c21cm_halo_consts &c = h.c;
    const double hmass = (double)A.masses[t];
    if (hmass == 0.) return false;  // halos cut from the catalogue (:388-390)
    double pos[3] = {(double)A.coords[3 * t], (double)A.coords[3 * t + 1], (double)A.coords[3 * t + 2]};
    int ip[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
        ip[a] = wrap_idx((int)(pos[a] * h.cell_size_inv_v + 0.5), h.vel_dim[a]);
    const size_t vi =
        (size_t)ip[2] + (size_t)h.vel_dim[2] * ((size_t)ip[1] + (size_t)h.vel_dim[1] * ip[0]);
    const float v[3] = {A.vx[vi], A.vy[vi], A.vz[vi]};
    float v2[3] = {0.f, 0.f, 0.f};
    if (h.lpt2) v2[0] = A.v2x[vi], v2[1] = A.v2y[vi], v2[2] = A.v2z[vi];
    {
        int i0[3], i1[3];
        double d[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            pos[a] += (double)v[a] * h.vdf;
            if (h.lpt2) pos[a] -= (double)v2[a] * h.vdf2;
            pos[a] = pos[a] * h.out_dim[a] / h.box_size[a];
            const int ipos = (int)floor(pos[a]);
            ipos_out[a] = ipos;
            d[a] = pos[a] - (double)ipos;
            i0[a] = wrap_idx(ipos, h.out_dim[a]);
            i1[a] = wrap_idx(ipos + 1, h.out_dim[a]);
        }
        const size_t sy = (size_t)h.out_dim[2], sx = (size_t)h.out_dim[1] * h.out_dim[2];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            idx[k] = (size_t)((k & 1) ? i1[0] : i0[0]) * sx + (size_t)((k & 2) ? i1[1] : i0[1]) * sy +
                     (size_t)((k & 4) ? i1[2] : i0[2]);
            w[k] = ((k & 1) ? d[0] : 1. - d[0]) * ((k & 2) ? d[1] : 1. - d[1]) *
                   ((k & 4) ? d[2] : 1. - d[2]);
        }
    }
    double M_turn_a = c.mturn_a_nofb, M_turn_m = c.mturn_m_nofb;
    if (c.use_mini_halos) {  // the turnover grids, CIC-read at the halo (:413-416)
        M_turn_a = pow(10., cic_read(A.mturn_a, idx, w));
        M_turn_m = pow(10., cic_read(A.mturn_m, idx, w));
    }
    const HaloProps p = halo_relations(h, hmass, (double)A.star_rng[t], (double)A.sfr_rng[t],
                                       c.use_xray ? (double)A.xray_rng[t] : 0., M_turn_a, M_turn_m);
    const double n_ion = p.n_ion, sfr = p.sfr, sfr_mini = p.sfr_mini, xray = p.xray, wsfr = p.wsfr;
    val[0] = n_ion * h.cell_vol_inv;
    val[1] = sfr * h.cell_vol_inv;
    val[2] = sfr_mini * h.cell_vol_inv;
    val[3] = xray * h.cell_vol_inv;
    val[4] = wsfr * h.cell_vol_inv;
    return true;
}

// direct path: eight global fp64 atomics per value and halo (~35 G atomics/s on the MI355X)
__global__ void __launch_bounds__(kBlock)
halo_deposit_kernel(HaloDepositParams h, HaloArrays A, HaloOutputs O) {
    for (unsigned long long t = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; t < h.n_halos;
         t += (unsigned long long)gridDim.x * kBlock) {
        int ipos[3];
        size_t idx[8];
        double w[8], val[5];
        if (!halo_eval(h, A, t, ipos, idx, w, val)) continue;
        for (int g = 0; g < 5; g++) {
            if (!O.g[g]) continue;
#pragma unroll
            for (int k = 0; k < 8; k++) unsafeAtomicAdd(O.g[g] + idx[k], val[g] * w[k]);
        }
    }
}

// ---- LDS-tiled path -------------------------------------------------------------------------
// The halos are binned by the brick of kHaloBrick^3 OUTPUT cells their catalogue position lies in
// (count, scan, fill: two int atomics per halo, aggregated over runs of equal bricks within a
// wavefront, so that a catalogue in cell order costs one atomic per run); a workgroup then owns a
// brick, deposits its halos into fp64 tiles in LDS (brick + kHaloMargin cells of margin for the
// displacement) and flushes each touched cell with one global atomic per value.  Halos displaced
// beyond the margin take the direct path, so the result does not depend on the geometry.
constexpr int kHaloBrick = 8, kHaloMargin = 2, kHaloTile = kHaloBrick + 2 * kHaloMargin + 1;
constexpr int kHaloTileCells = kHaloTile * kHaloTile * kHaloTile;

struct HaloBrickGeom {
    int nb[3];
    int n_bricks;
};

// brick of a halo from its catalogue position (-1: cut halo); ucell: the wrapped output cell,
// ufloor: the unwrapped one
__device__ __forceinline__ int halo_brick(const HaloDepositParams &h, const HaloBrickGeom &g,
                                          const HaloArrays &A, unsigned long long t, int ucell[3],
                                          int ufloor[3]) {
    if (A.masses[t] == 0.f) return -1;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double pos = (double)A.coords[3 * t + a] * h.out_dim[a] / h.box_size[a];
        ufloor[a] = (int)floor(pos);
        ucell[a] = wrap_idx(ufloor[a], h.out_dim[a]);
    }
    return ((ucell[0] / kHaloBrick) * g.nb[1] + ucell[1] / kHaloBrick) * g.nb[2] + ucell[2] / kHaloBrick;
}

// atomicAdd(&counters[key], 1) for every lane with key >= 0, one atomic per run of equal keys in
// the wavefront; returns the lane's slot.  All 64 lanes must call it.
__device__ __forceinline__ int wave_run_increment(int *counters, int key) {
    const int lane = (int)__lane_id();
    const int prev = __shfl_up(key, 1);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const int start = 63 - __clzll((long long)(heads & upto));
    const unsigned long long above = heads & ~upto;
    const int end = above ? __ffsll((long long)above) - 1 : 64;
    int base = 0;
    if (head && key >= 0) base = atomicAdd(&counters[key], end - start);
    base = __shfl(base, start);
    return base + (lane - start);
}

__global__ void __launch_bounds__(kBlock)
halo_brick_count_kernel(HaloDepositParams h, HaloBrickGeom g, HaloArrays A, int *counts) {
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < h.n_halos;
         base += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long t = base + threadIdx.x;
        int uc[3], uf[3];
        const int key = t < h.n_halos ? halo_brick(h, g, A, t, uc, uf) : -1;
        (void)wave_run_increment(counts, key);
    }
}

// exclusive scan of n counts into offsets[n + 1] and a working copy cursor[n]; one workgroup
__global__ void __launch_bounds__(1024)
halo_brick_scan_kernel(const int *__restrict__ counts, int *__restrict__ offsets,
                       int *__restrict__ cursor, int n) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, chunk = (n + 1023) / 1024;
    const int lo = tid * chunk < n ? tid * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    int s = 0;
    for (int i = lo; i < hi; i++) s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; i++) {
        offsets[i] = run;
        cursor[i] = run;
        run += counts[i];
    }
    if (tid == 1023) offsets[n] = part[1023];
}

__global__ void __launch_bounds__(kBlock)
halo_brick_fill_kernel(HaloDepositParams h, HaloBrickGeom g, HaloArrays A, int *cursor,
                       unsigned int *__restrict__ order) {
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < h.n_halos;
         base += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long t = base + threadIdx.x;
        int uc[3], uf[3];
        const int key = t < h.n_halos ? halo_brick(h, g, A, t, uc, uf) : -1;
        const int slot = wave_run_increment(cursor, key);
        if (key >= 0) order[slot] = (unsigned int)t;
    }
}

__global__ void __launch_bounds__(kBlock)
halo_deposit_tiled_kernel(HaloDepositParams h, HaloBrickGeom g, HaloArrays A, HaloOutputs O,
                          const int *__restrict__ offsets, const unsigned int *__restrict__ order) {
    extern __shared__ double tile[];  // [nv][kHaloTileCells]
    int slot_of[5], nv = 0;
#pragma unroll
    for (int v = 0; v < 5; v++) slot_of[v] = O.g[v] ? nv++ : -1;
    const size_t sy = (size_t)h.out_dim[2], sx = (size_t)h.out_dim[1] * h.out_dim[2];
    for (int brick = blockIdx.x; brick < g.n_bricks; brick += gridDim.x) {
        const int lo = offsets[brick], hi = offsets[brick + 1];
        if (lo == hi) continue;  // uniform over the workgroup
        const int b0 = brick / (g.nb[1] * g.nb[2]), b1 = (brick / g.nb[2]) % g.nb[1], b2 = brick % g.nb[2];
        const int t0[3] = {b0 * kHaloBrick - kHaloMargin, b1 * kHaloBrick - kHaloMargin,
                           b2 * kHaloBrick - kHaloMargin};
        for (int c = threadIdx.x; c < nv * kHaloTileCells; c += kBlock) tile[c] = 0.;
        __syncthreads();
        for (int e = lo + (int)threadIdx.x; e < hi; e += kBlock) {
            const unsigned long long t = order[e];
            int ipos[3], uc[3], uf[3];
            size_t idx[8];
            double w[8], val[5];
            if (!halo_eval(h, A, t, ipos, idx, w, val)) continue;
            (void)halo_brick(h, g, A, t, uc, uf);
            int rel[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                rel[a] = (ipos[a] - uf[a]) + (uc[a] - t0[a]);
                inside = inside && rel[a] >= 0 && rel[a] + 1 < kHaloTile;
            }
            if (inside) {
                const int base = (rel[0] * kHaloTile + rel[1]) * kHaloTile + rel[2];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int cell = base + (((k & 1) ? kHaloTile : 0) + ((k & 2) ? 1 : 0)) * kHaloTile +
                                     ((k & 4) ? 1 : 0);
#pragma unroll
                    for (int v = 0; v < 5; v++)
                        if (slot_of[v] >= 0)
                            __hip_atomic_fetch_add(&tile[slot_of[v] * kHaloTileCells + cell], val[v] * w[k],
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 5; v++) {
                    if (!O.g[v]) continue;
#pragma unroll
                    for (int k = 0; k < 8; k++) unsafeAtomicAdd(O.g[v] + idx[k], val[v] * w[k]);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < kHaloTileCells; c += kBlock) {
            const int c2 = c % kHaloTile, c1 = (c / kHaloTile) % kHaloTile, c0 = c / (kHaloTile * kHaloTile);
            const size_t o = (size_t)wrap_idx(t0[0] + c0, h.out_dim[0]) * sx +
                             (size_t)wrap_idx(t0[1] + c1, h.out_dim[1]) * sy +
                             (size_t)wrap_idx(t0[2] + c2, h.out_dim[2]);
#pragma unroll
            for (int v = 0; v < 5; v++) {
                if (slot_of[v] < 0) continue;
                const double tv = tile[slot_of[v] * kHaloTileCells + c];
                if (tv != 0.) unsafeAtomicAdd(O.g[v] + o, tv);
            }
        }
        __syncthreads();
    }
}
}  // namespace

// ints of scratch the tiled path wants: counts, offsets (+1), cursors per brick and one slot per halo
extern "C" size_t c21hip_halo_deposit_scratch_ints(unsigned long long n_halos, const int out_dim[3]) {
    size_t n_bricks = 1;
    for (int a = 0; a < 3; a++) n_bricks *= (size_t)((out_dim[a] + kHaloBrick - 1) / kHaloBrick);
    return 3 * (n_bricks + 1) + (size_t)n_halos;
}

extern "C" int c21hip_halo_deposit(const c21cm_halo_consts *consts, unsigned long long n_halos,
                                   const float *masses, const float *coords, const float *star_rng,
                                   const float *sfr_rng, const float *xray_rng,
                                   const float *const vel[3], const float *const vel2[3],
                                   const int vel_dim[3], const int out_dim[3], double box_len,
                                   double box_len_z, double growth, double init_growth, int lpt2,
                                   const float *mturn_a, const float *mturn_m, double *out_nion,
                                   double *out_sfr, double *out_sfr_mini, double *out_xray,
                                   double *out_wsfr, int *scratch, void *stream) {
    if (!n_halos) return 0;
    HaloDepositParams h;
    h.c = *consts;
    h.n_halos = n_halos;
    const double box[3] = {box_len, box_len, box_len_z};
    for (int a = 0; a < 3; a++) {
        h.vel_dim[a] = vel_dim[a];
        h.out_dim[a] = out_dim[a];
        h.box_size[a] = box[a];
    }
    h.cell_size_inv_v = vel_dim[0] / box_len;
    h.vdf = growth - init_growth;
    h.vdf2 = -(3.0 / 7.0) * growth * growth - (-(3.0 / 7.0) * init_growth * init_growth);
    const double cell_size_inv_o = out_dim[0] / box_len;
    h.cell_vol_inv = cell_size_inv_o * cell_size_inv_o * cell_size_inv_o;
    h.lpt2 = lpt2;
    h.ln_pivot_upper = log(consts->pivot_upper);
    h.ln_tstar_th = log(consts->t_star * consts->t_h);
    h.ln_s_per_yr = log(kSecPerYr);
    h.ln_m0_norm = log(1.28825e10);
    h.ln_z_norm = log(1.23 * pow(10., -0.056 * consts->redshift + 0.064) / 0.05);
    const HaloArrays A = {masses, coords, star_rng, sfr_rng, xray_rng, vel[0], vel[1], vel[2],
                          vel2[0], vel2[1], vel2[2], mturn_a, mturn_m};
    const HaloOutputs O = {{out_nion, out_sfr, out_sfr_mini, out_xray, out_wsfr}};
    hipStream_t st = (hipStream_t)stream;
    HaloBrickGeom g;
    size_t n_bricks = 1;
    for (int a = 0; a < 3; a++) {
        g.nb[a] = (out_dim[a] + kHaloBrick - 1) / kHaloBrick;
        n_bricks *= (size_t)g.nb[a];
    }
    static std::once_flag once;
    static int direct = 0;    // C21CM_HALO_DEPOSIT=direct: eight global atomics per value and halo
    static size_t lds_cap = 0;  // dynamic LDS the tiled kernel may ask for on this device
    std::call_once(once, [] {
        const char *e = getenv("C21CM_HALO_DEPOSIT");
        direct = e && e[0] == 'd';
        int dev = 0, max_lds = 0;
        const int want = 5 * kHaloTileCells * (int)sizeof(double);
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess) {
            const int ask = want < max_lds ? want : max_lds;
            if (hipFuncSetAttribute((const void *)halo_deposit_tiled_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess)
                lds_cap = (size_t)ask;
        }
        (void)hipGetLastError();
    });
    int nv_tiles = 0;
    for (int v = 0; v < 5; v++) nv_tiles += O.g[v] != nullptr;
    // (a device with less LDS than the tiles of this call need -- 88 KB for five grids -- takes the
    //  direct deposit instead of failing the launch: ADVICE r2)
    const bool tiles_fit = (size_t)nv_tiles * kHaloTileCells * sizeof(double) <= lds_cap;
    if (direct || !tiles_fit || !scratch || n_halos >= (1ull << 31) || n_bricks >= (1ull << 30)) {
        hipLaunchKernelGGL(halo_deposit_kernel, dim3(grid_for((size_t)n_halos)), dim3(kBlock), 0, st,
                           h, A, O);
        LAUNCH_CHECK();
        return 0;
    }
    g.n_bricks = (int)n_bricks;
    int *counts = scratch, *offsets = counts + n_bricks + 1, *cursor = offsets + n_bricks + 1;
    unsigned int *order = (unsigned int *)(cursor + n_bricks + 1);
    int status = c21hip_memset(counts, 0, n_bricks * sizeof(int), stream);
    if (status) return status;
    hipLaunchKernelGGL(halo_brick_count_kernel, dim3(grid_for((size_t)n_halos)), dim3(kBlock), 0, st,
                       h, g, A, counts);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(halo_brick_scan_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, cursor,
                       g.n_bricks);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(halo_brick_fill_kernel, dim3(grid_for((size_t)n_halos)), dim3(kBlock), 0, st,
                       h, g, A, cursor, order);
    LAUNCH_CHECK();
    int nv = 0;
    for (int v = 0; v < 5; v++) nv += O.g[v] != nullptr;
    const size_t lds = (size_t)nv * kHaloTileCells * sizeof(double);
    const int blocks = (int)(n_bricks < 256 * 16 ? n_bricks : 256 * 16);
    hipLaunchKernelGGL(halo_deposit_tiled_kernel, dim3(blocks), dim3(kBlock), lds, st, h, g, A, O,
                       offsets, order);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// test_halo_props (HaloBox.c:658-779): the twelve properties of every halo of a catalogue, with
// the turnover masses of the cell the halo sits in (no displacement, no CIC read).
namespace {
struct HaloPropsParams {
    HaloDepositParams h;
    int dim[3];
    double cell_length;
    float z;
    int below_z_heat_max, vcb_flucts;
    double m_turn, vcb_const, A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb;
};

__global__ void __launch_bounds__(kBlock)
halo_props_kernel(HaloPropsParams q, const float *__restrict__ masses,
                  const float *__restrict__ coords, const float *__restrict__ star_rng,
                  const float *__restrict__ sfr_rng, const float *__restrict__ xray_rng,
                  const float *__restrict__ vcb, const float *__restrict__ J21,
                  const float *__restrict__ z_re, const float *__restrict__ G12,
                  float *__restrict__ out) {
    const c21cm_halo_consts &c = q.h.c;
    for (unsigned long long t = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; t < q.h.n_halos;
         t += (unsigned long long)gridDim.x * kBlock) {
        const double m = (double)masses[t];
        if (m == 0.) continue;  // :693-695
        double M_turn_a = c.mturn_a_nofb, M_turn_m = c.mturn_m_nofb, M_turn_r = 0.;
        if (c.use_mini_halos) {
            int cell[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double pos = (double)coords[3 * t + a] / q.cell_length;
                if (pos == (double)(float)q.dim[0]) pos = (double)((float)q.dim[0] - 0.1);  // :700-703
                cell[a] = wrap_idx((int)pos, q.dim[a]);
            }
            const size_t i = (size_t)cell[2] + (size_t)q.dim[2] * ((size_t)cell[1] + (size_t)q.dim[1] * cell[0]);
            const double vc = q.vcb_flucts ? (double)vcb[i] : (double)(float)q.vcb_const;  // float arguments
            double j = 0., g = 0., zin = 0.;
            if (q.below_z_heat_max) j = (double)J21[i], g = (double)G12[i], zin = (double)z_re[i];
            const double zp1 = 1. + (double)q.z;
            M_turn_m = 3.314e7 * pow(zp1, -1.5) * (1.0 + q.A_LW * pow(j, q.BETA_LW)) *
                       pow(1.0 + q.A_VCB * vc / q.sigma_vcb, q.BETA_VCB);
            M_turn_r = 1e-40;
            if (!(zin <= 1e-19))
                M_turn_r = 3e9 * pow(2.0 * g, 0.17) * pow(zp1 / 10, -2.1) *
                           pow(1 - pow(zp1 / (1. + zin), 2.0), 2.5);
            M_turn_a = fmax(M_turn_a, fmax(M_turn_r, q.m_turn));
            M_turn_m = fmax(M_turn_m, fmax(M_turn_r, q.m_turn));
        }
        const HaloProps p = halo_relations(q.h, m, (double)star_rng[t], (double)sfr_rng[t],
                                           (double)xray_rng[t], M_turn_a, M_turn_m);
        float *o = out + 12 * t;
        o[0] = (float)m, o[1] = (float)p.stars, o[2] = (float)p.sfr, o[3] = (float)p.xray;
        o[4] = (float)p.n_ion, o[5] = (float)p.wsfr, o[6] = (float)p.stars_mini, o[7] = (float)p.sfr_mini;
        o[8] = (float)M_turn_a, o[9] = (float)M_turn_m, o[10] = (float)M_turn_r, o[11] = (float)p.metallicity;
    }
}

void fill_relation_constants(HaloDepositParams &h, const c21cm_halo_consts *consts) {
    h.c = *consts;
    h.ln_pivot_upper = log(consts->pivot_upper);
    h.ln_tstar_th = log(consts->t_star * consts->t_h);
    h.ln_s_per_yr = log(kSecPerYr);
    h.ln_m0_norm = log(1.28825e10);
    h.ln_z_norm = log(1.23 * pow(10., -0.056 * consts->redshift + 0.064) / 0.05);
}
}  // namespace

// lw = {A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb, vcb_const, M_TURN}; grids may be NULL without
// mini-halos (vcb only read with vcb_flucts, the other three only below Z_HEAT_MAX)
extern "C" int c21hip_halo_props(const c21cm_halo_consts *consts, unsigned long long n_halos,
                                 const float *masses, const float *coords, const float *star_rng,
                                 const float *sfr_rng, const float *xray_rng, const int dim[3],
                                 double cell_length, double redshift, int below_z_heat_max,
                                 int vcb_flucts, const double lw[7], const float *vcb,
                                 const float *J21, const float *z_re, const float *G12, float *out,
                                 void *stream) {
    if (!n_halos) return 0;
    HaloPropsParams q;
    fill_relation_constants(q.h, consts);
    q.h.n_halos = n_halos;
    q.h.cell_vol_inv = 1.;
    for (int a = 0; a < 3; a++) q.dim[a] = dim[a];
    q.cell_length = cell_length;
    q.z = (float)redshift;
    q.below_z_heat_max = below_z_heat_max;
    q.vcb_flucts = vcb_flucts;
    q.A_LW = lw[0], q.BETA_LW = lw[1], q.A_VCB = lw[2], q.BETA_VCB = lw[3], q.sigma_vcb = lw[4];
    q.vcb_const = lw[5], q.m_turn = lw[6];
    hipLaunchKernelGGL(halo_props_kernel, dim3(grid_for((size_t)n_halos)), dim3(kBlock), 0,
                       (hipStream_t)stream, q, masses, coords, star_rng, sfr_rng, xray_rng, vcb, J21,
                       z_re, G12, out);
    LAUNCH_CHECK();
    return 0;
}
