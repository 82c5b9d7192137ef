// ionize_kernels.hip -- the per-radius real-space sweeps of ComputeIonizedBox.
//
// reference loops being replaced (src/py21cmfast/src/IonisationBox.c):
//   clip_and_get_extrema   :668-699    -> clip_minmax_kernel
//   calculate_fcoll_grid   :773-962    -> fcoll_eulerian_kernel / fused into ionise_stars_kernel
//   find_ionised_regions   :1008-1201  -> ionise_stars_kernel / ionise_eulerian_kernel
//   set_ionized_temperatures + sum(xH) :1203-1256,1597-1608 -> finalize_kernel
//
// Design (HBM-bound, no MFMA):
//   * the reference writes the clipped filtered grids back and re-reads them in the
//     next loop; here the clips happen in registers and the filtered grids are only
//     ever READ by these kernels -- for Lagrangian source grids the f_coll sum, the
//     barrier test and the partial-ionisation branch are one sweep;
//   * each thread owns two z-neighbours (float2 from the padded grids, whose rows are
//     only 8-byte aligned, and float2 from the dense grids);
//   * global sums are two-stage and deterministic: wavefront shuffle -> LDS -> one
//     partial per workgroup -> a single-workgroup finishing kernel that adds the
//     partials in index order (no float atomics, bit-reproducible run to run);
//   * outputs (xH, z_reion, T_k) are written only where the reference writes them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "c21hip.h"
#include "c21cm_abi.h"
#include "c21cm_grid.h"
#include "fcoll_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;
constexpr double kTiny = 1e-30;          // reference: Constants.h TINY
constexpr double kMinDensityLowLimit = 9e-8;  // reference: thermochem.c:16

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

// ---- reductions -----------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// one partial per workgroup
__device__ __forceinline__ void block_sum_to(double v, double *partials) {
    __shared__ double lds[kBlock / 64];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) s += lds[w];
        partials[blockIdx.x] = s;
    }
}

// op: 0 sum, 1 min, 2 max.  Single workgroup, fixed summation order.
__global__ void __launch_bounds__(kBlock)
finish_reduce_kernel(const double *__restrict__ partials, int n, int op, double *out) {
    __shared__ double lds[kBlock];
    double acc = (op == 0) ? 0. : partials[0];
    for (int i = threadIdx.x; i < n; i += kBlock) {
        double p = partials[i];
        acc = (op == 0) ? acc + p : (op == 1 ? fmin(acc, p) : fmax(acc, p));
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            double a = lds[threadIdx.x], b = lds[threadIdx.x + s];
            lds[threadIdx.x] = (op == 0) ? a + b : (op == 1 ? fmin(a, b) : fmax(a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = lds[0];
}

// mean = sum/N with the clamp of IonisationBox.c:1566-1576
__global__ void finish_mean_kernel(const double *sum, double ntot, int mass_dep_zeta,
                                   double f_limit, double *mean_out) {
    double m = *sum / ntot;
    if (mass_dep_zeta) {
        if (m <= f_limit) m = f_limit;
    } else {
        if (m <= kFractFloatErr) m = kFractFloatErr;
    }
    *mean_out = m;
}

// ---- per-cell physics -------------------------------------------------------------
// reference: thermochem.c:58-63 (all three arguments arrive as float)
__device__ __forceinline__ float partially_ionized_T(float T_HI, float res_xH, float T_re) {
    if (res_xH <= 0.f) return T_re;
    if (res_xH >= 1.f) return T_HI;
    return (float)((double)__fmul_rn(T_HI, res_xH) + (double)T_re * (1. - (double)res_xH));
}

// x^a for x > 0 to float accuracy (~3e-7 relative for |a| <= 3.4; the result is stored as float
// anyway): x = m 2^e, log2 x = e + log2(m) with the hardware log2 on m in [0.5, 1) (absolute error
// ~1e-7, kept small by taking the exponent out first), the product split into integer and
// fraction, hardware exp2.  A libm fp64 pow costs ~200 instructions per call and made the final
// sweep ALU-bound (85 % of the cells of the benchmark box take one).
__device__ __forceinline__ float pow_f32acc(double x, double a) {
    if (!(x > 0.) || !(x < 1e300)) return (float)pow(x, a);
    int e;
    const double m = frexp(x, &e);
    const double t = a * ((double)e + (double)__builtin_amdgcn_logf((float)m));
    const double n = floor(t);
    return ldexpf(__builtin_amdgcn_exp2f((float)(t - n)), (int)n);
}

// reference: thermochem.c:31-56.  pow_Tre = pow(T_re, 1.7) and pow_z = pow(1e4*((1+z)/4), 1.7)
// do not depend on the cell and are passed in (same double values the reference computes).
__device__ float fully_ionized_T(float z_re, float z, float delta, double pow_Tre, double pow_z) {
    float result, delta_re;
    if (fabs((double)(z - z_re)) < 1e-4) {
        result = 1.f;
    } else {
        delta_re = (float)((double)delta * (1. + (double)z) / (1. + (double)z_re));
        if (delta_re <= -1.f) delta_re = (float)(-1. + kMinDensityLowLimit);
        if (delta <= -1.f) delta = (float)(-1. + kMinDensityLowLimit);
        result = pow_f32acc((1. + (double)delta) / (1. + (double)delta_re), 1.1333);
        result = (float)((double)result * (double)pow_f32acc((1. + (double)z) / (1. + (double)z_re), 3.4));
        result = __fmul_rn(result, expf((float)((double)pow_f32acc((1. + (double)z) / 7.1, 2.5) -
                                                (double)pow_f32acc((1. + (double)z_re) / 7.1, 2.5))));
    }
    result = (float)((double)result * pow_Tre);
    result = (float)((double)result + pow_z * (double)(1.f + delta));
    result = pow_f32acc((double)result, 0.5882);
    return result;
}

__device__ __forceinline__ float clip_xe(float v) { return fminf(fmaxf(v, 0.f), 0.999f); }

// ---- index helper: item i -> (padded element index, dense element index), VEC cells each
template <int VEC>
struct CellIndex {
    size_t padded;  // in units of VEC floats
    size_t dense;   // in units of VEC floats
};
template <int VEC>
__device__ __forceinline__ CellIndex<VEC> cell_index(size_t i, int nz_items, int zpad_items) {
    const size_t line = i / (size_t)nz_items;
    const int k = (int)(i - line * (size_t)nz_items);
    return CellIndex<VEC>{line * (size_t)zpad_items + k, i};
}

// rows of a power-of-two number of items: the 64-bit division becomes a shift (sh = log2(nz_items), -1: divide)
template <int VEC>
__device__ __forceinline__ CellIndex<VEC> cell_index_sh(size_t i, int nz_items, int zpad_items, int sh) {
    if (sh < 0) return cell_index<VEC>(i, nz_items, zpad_items);
    const size_t line = i >> sh;
    const int k = (int)(i & (size_t)(nz_items - 1));
    return CellIndex<VEC>{line * (size_t)zpad_items + k, i};
}
__device__ __forceinline__ int row_shift(int nz_items) {
    return (nz_items > 0 && (nz_items & (nz_items - 1)) == 0) ? __builtin_ctz(nz_items) : -1;
}

template <int VEC>
struct Pack;
template <>
struct Pack<1> {
    float v[1];
    __device__ __forceinline__ static Pack load(const float *p, size_t i) { return Pack{{p[i]}}; }
    __device__ __forceinline__ void store(float *p, size_t i) const { p[i] = v[0]; }
};
template <>
struct Pack<2> {
    float v[2];
    __device__ __forceinline__ static Pack load(const float *p, size_t i) {
        float2 t = reinterpret_cast<const float2 *>(p)[i];
        return Pack{{t.x, t.y}};
    }
    __device__ __forceinline__ void store(float *p, size_t i) const {
        reinterpret_cast<float2 *>(p)[i] = make_float2(v[0], v[1]);
    }
};

template <>
struct Pack<4> {
    float v[4];
    __device__ __forceinline__ static Pack load(const float *p, size_t i) {
        float4 t = reinterpret_cast<const float4 *>(p)[i];
        return Pack{{t.x, t.y, t.z, t.w}};
    }
    __device__ __forceinline__ void store(float *p, size_t i) const {
        reinterpret_cast<float4 *>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
};

// ---- clip_and_get_extrema ------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(kBlock)
minmax_kernel(const float *__restrict__ delta_fil, size_t nitems, int nz_items, int zpad_items,
              double *__restrict__ pmin, double *__restrict__ pmax) {
    // the reference seeds min/max with cell 0 (IonisationBox.c:672-673)
    double lo = (double)delta_fil[0], hi = lo;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, nz_items, zpad_items);
        const auto d = Pack<VEC>::load(delta_fil, ci.padded);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            lo = fmin(lo, (double)d.v[e]);
            hi = fmax(hi, (double)d.v[e]);
        }
    }
    __shared__ double lds_lo[kBlock / 64], lds_hi[kBlock / 64];
    lo = wave_min(lo);
    hi = wave_max(hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds_lo[wave] = lo;
        lds_hi[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) {
            lo = fmin(lo, lds_lo[w]);
            hi = fmax(hi, lds_hi[w]);
        }
        pmin[blockIdx.x] = lo;
        pmax[blockIdx.x] = hi;
    }
}

// ---- calculate_fcoll_grid, Eulerian source models -------------------------------------
struct FcollParams {
    int mode;  // enum c21cm_fcoll_mode
    float growthf;
    double sig;  // sqrt(sig_small^2 - sig_large^2) from the float sigmas; <0: equal sigmas
    double delta_c;
    double tab_min, tab_width;
};

template <int VEC>
__global__ void __launch_bounds__(kBlock)
fcoll_eulerian_kernel(const float *__restrict__ delta_fil, float *__restrict__ nion_dense,
                      size_t nitems, int nz_items, int zpad_items, FcollParams fp,
                      const float *__restrict__ table, double *__restrict__ partials) {
    __shared__ float tab[C21CM_NDELTA_TABLE];
    if (fp.mode >= C21CM_FCOLL_TABLE_LINEAR) {
        for (int t = threadIdx.x; t < C21CM_NDELTA_TABLE; t += kBlock) tab[t] = table[t];
        __syncthreads();
    }
    double acc = 0.;
    const double inv_w = 1. / fp.tab_width;
    const int sh = row_shift(nz_items);
    constexpr int U = 4;  // items per thread and trip, loads issued before the arithmetic
    for (size_t i0 = (size_t)blockIdx.x * kBlock * U + threadIdx.x; i0 < nitems;
         i0 += (size_t)gridDim.x * kBlock * U) {
        Pack<VEC> d[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + (size_t)u * kBlock;
            if (i < nitems) d[u] = Pack<VEC>::load(delta_fil, cell_index_sh<VEC>(i, nz_items, zpad_items, sh).padded);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + (size_t)u * kBlock;
            if (i >= nitems) continue;
            Pack<VEC> out;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const float dens = clip_delta_eulerian(d[u].v[e]);
                double f;
                if (fp.mode == C21CM_FCOLL_ERFC) {
                    f = (fp.sig < 0) ? 0. : fgtrm_bias_fast(fp.growthf, dens, fp.sig, fp.delta_c);
                } else if (fp.mode == C21CM_FCOLL_TABLE_LINEAR) {
                    f = eval_table_f_inv((double)dens, fp.tab_min, fp.tab_width, inv_w, tab);
                } else {
                    f = exp_f32acc(eval_table_f_inv((double)dens, fp.tab_min, fp.tab_width, inv_w, tab));
                }
                out.v[e] = (float)f;  // box->unnormalised_nion is float (IonisationBox.c:951)
                acc += f;
            }
            out.store(nion_dense, i);
        }
    }
    block_sum_to(acc, partials);
}

// fcoll_eulerian_kernel<2> of a table mode with the radius' barrier decided in the same sweep wherever
// it does not depend on the exact box mean (the banded barrier of the closed-form loop, pass Z EPI 7 in
// fft_native.hip, for the loops whose f_coll comes from a per-radius table): band[0] / band[1] = the two
// float thresholds of this radius (eul_band_step), cells in between get the marker 255 in first_cross
// and their f_coll in f_pend (a sparse write: NOT a complete grid); r_prev >= 0: the radius whose
// markers are outstanding, *thr_prev its exact threshold.  The dense f_coll grid is never written and
// eulerian_mask_kernel not launched: 4 N + 1 N bytes read instead of (4 + 4) N + 6 N per radius.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
fcoll_eulerian_band_kernel(const float *__restrict__ delta_fil, float *__restrict__ f_pend,
                           unsigned char *__restrict__ first_cross, size_t nitems, int nz_items,
                           int zpad_items, FcollParams fp, const float *__restrict__ table,
                           const double *__restrict__ band, const double *__restrict__ thr_prev,
                           int r_index, int r_prev, double *__restrict__ partials) {
    __shared__ float tab[C21CM_NDELTA_TABLE];
    for (int t = threadIdx.x; t < C21CM_NDELTA_TABLE; t += kBlock) tab[t] = table[t];
    __syncthreads();
    const float t_sure = (float)band[0], t_maybe = (float)band[1];
    const float t_prev = r_prev >= 0 ? (float)*thr_prev : 0.f;
    double acc = 0.;
    const double inv_w = 1. / fp.tab_width;
    const int sh = row_shift(nz_items);
    constexpr int U = 4;
    for (size_t i0 = (size_t)blockIdx.x * kBlock * U + threadIdx.x; i0 < nitems;
         i0 += (size_t)gridDim.x * kBlock * U) {
        Pack<VEC> d[U];
        unsigned char mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + (size_t)u * kBlock;
            if (i < nitems) {
                d[u] = Pack<VEC>::load(delta_fil, cell_index_sh<VEC>(i, nz_items, zpad_items, sh).padded);
                if constexpr (VEC == 4) {
                    const uchar4 m4 = reinterpret_cast<const uchar4 *>(first_cross)[i];
                    mk[u][0] = m4.x, mk[u][1] = m4.y, mk[u][2 % VEC] = m4.z, mk[u][3 % VEC] = m4.w;
                } else {
                    const uchar2 m2 = reinterpret_cast<const uchar2 *>(first_cross)[i];
                    mk[u][0] = m2.x, mk[u][1] = m2.y;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + (size_t)u * kBlock;
            if (i >= nitems) continue;
            unsigned char mv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e++) mv[e] = mk[u][e];
            bool ch = false;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const float dens = clip_delta_eulerian(d[u].v[e]);
                double f;
                if (fp.mode == C21CM_FCOLL_TABLE_LINEAR)
                    f = eval_table_f_inv((double)dens, fp.tab_min, fp.tab_width, inv_w, tab);
                else
                    f = exp_f32acc(eval_table_f_inv((double)dens, fp.tab_min, fp.tab_width, inv_w, tab));
                acc += f;
                const float g = (float)f;  // what the dense grid would hold
                if (mv[e] == 255) {        // the previous radius' undecided cell
                    mv[e] = (f_pend[VEC * i + e] >= t_prev) ? (unsigned char)r_prev : (unsigned char)0;
                    ch = true;
                }
                if (mv[e] == 0 && g >= t_maybe) {
                    const bool sure = g >= t_sure;
                    mv[e] = sure ? (unsigned char)r_index : (unsigned char)255;
                    if (!sure) f_pend[VEC * i + e] = g;
                    ch = true;
                }
            }
            if (ch) {
                if constexpr (VEC == 4)
                    reinterpret_cast<uchar4 *>(first_cross)[i] = make_uchar4(mv[0], mv[1], mv[2 % VEC], mv[3 % VEC]);
                else
                    reinterpret_cast<uchar2 *>(first_cross)[i] = make_uchar2(mv[0], mv[1]);
            }
        }
    }
    block_sum_to(acc, partials);
}

// E-INTEGRAL without interpolation tables (C21CM_FCOLL_NODES): Nion_ConditionalM of every cell
// (IonisationBox.c:889-893 -> hmf.c:1106-1140, Gauss-Legendre) from the radius' node data in LDS:
// one exp and a handful of multiply-adds per (cell, node), all in double like the host integral.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
fcoll_nodes_kernel(const float *__restrict__ delta_fil, float *__restrict__ nion_dense, size_t nitems,
                   int nz_items, int zpad_items, const double *__restrict__ nodes_dev,
                   double *__restrict__ partials) {
    __shared__ double nd[C21CM_NODE_DOUBLES];
    for (int t = threadIdx.x; t < C21CM_NODE_DOUBLES; t += kBlock) nd[t] = nodes_dev[t];
    __syncthreads();
    const int n = (int)nd[0];
    const bool st = nd[1] != 0.;
    const double inv_growth = 1. / nd[2], limit = nd[3], collapsed = nd[4];
    const bool empty = nd[5] != 0.;
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nitems;
         i += (size_t)gridDim.x * kBlock) {
        const Pack<VEC> d = Pack<VEC>::load(delta_fil, cell_index<VEC>(i, nz_items, zpad_items).padded);
        Pack<VEC> out;
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const double delta = (double)clip_delta_eulerian(d.v[e]);
            double f = 0.;
            if (!empty) {
                if (delta > limit) {
                    f = collapsed;
                } else {
                    // (delta / growthf as the host writes it: a division, not a reciprocal product,
                    //  would cost a double division per cell; the 1-ulp difference of d0 is far below the
                    //  float the result is stored in)
                    const double d0 = delta * inv_growth;
                    const double del = (1.686 - delta) * inv_growth;
                    for (int k = 0; k < n; k++) {
                        const double w = nd[8 + 4 * k];
                        if (w == 0.) continue;
                        const double sdi = nd[8 + 4 * k + 3];
                        double cmf;
                        if (st) {
                            const double B = nd[8 + 4 * k + 2] - d0;
                            cmf = (nd[8 + 4 * k + 1] - d0) * exp(-B * B * 0.5 * sdi);
                        } else {
                            cmf = del * exp(-del * del * 0.5 * sdi);
                        }
                        f += w * cmf;
                    }
                }
            }
            out.v[e] = (float)f;  // box->unnormalised_nion is float (IonisationBox.c:951)
            acc += f;
        }
        out.store(nion_dense, i);
    }
    block_sum_to(acc, partials);
}

// ---- find_ionised_regions ---------------------------------------------------------------
struct IoniseParams {
    c21hip_ionize_args a;
    size_t nitems;
    int nz_items, zpad_items;
};

// Lagrangian source grids: f_coll sum + barrier + partial ionisation in ONE sweep.
template <int VEC, bool LAST, bool TS>
__global__ void __launch_bounds__(kBlock)
ionise_stars_kernel(IoniseParams p, const float *__restrict__ delta_fil,
                    const float *__restrict__ stars_fil, const float *__restrict__ xe_fil,
                    const float *__restrict__ density, const float *__restrict__ prev_z_reion,
                    const float *__restrict__ Tneutral, float *__restrict__ xH,
                    float *__restrict__ z_reion, float *__restrict__ Tk,
                    unsigned char *__restrict__ first_cross, double *__restrict__ partials) {
    const c21hip_ionize_args &a = p.a;
    double acc = 0.;
    const float z_now = (float)a.redshift;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < p.nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, p.nz_items, p.zpad_items);
        const auto st = Pack<VEC>::load(stars_fil, ci.padded);
        Pack<VEC> dl, xe, de;
        if (!LAST) dl = Pack<VEC>::load(delta_fil, ci.padded);
        if (TS) xe = Pack<VEC>::load(xe_fil, ci.padded);
        if (LAST) de = Pack<VEC>::load(density, ci.dense);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t idx = ci.dense * VEC + e;
            const float stars = fmaxf(st.v[e], 0.f);  // IonisationBox.c:822-823
            acc += (double)stars;
            // IonisationBox.c:1048-1052
            const double curr_dens = LAST ? (double)de.v[e] * a.photoncons_factor
                                          : (double)clip_delta(dl.v[e]);
            double curr_fcoll = (double)stars;
            curr_fcoll *= 1 / (a.rhocrit_omb * (1 + curr_dens));  // :1066-1067
            if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;  // :1077
            const double x_e = TS ? (double)clip_xe(xe.v[e]) : 0.;
            if (curr_fcoll * a.ion_eff_factor > (1. - x_e)) {  // :1118 (rec = 0)
                if (first_cross) {
                    if (first_cross[idx] == 0) first_cross[idx] = (unsigned char)a.r_index;
                } else {
                    const float pz = a.first_snapshot ? -1.f : prev_z_reion[idx];
                    z_reion[idx] = (pz < 0.f) ? z_now : pz;  // :1143-1147
                    xH[idx] = 0.f;                           // :1151
                }
            } else if (LAST) {
                if ((double)xH[idx] > kTiny) {  // :1161
                    double res_xH = 1. - curr_fcoll * a.ion_eff_factor;
                    if (!a.minimize_memory) {
                        const float T_HI =
                            TS ? Tneutral[idx]
                               : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)de.v[e]));
                        Tk[idx] = partially_ionized_T(T_HI, (float)res_xH, (float)a.T_re);
                    }
                    res_xH -= x_e;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    xH[idx] = (float)res_xH;
                }
            }
        }
    }
    block_sum_to(acc, partials);
}

// Eulerian source models: f_coll comes from the dense unnormalised_nion grid.
template <int VEC, bool LAST, bool TS>
__global__ void __launch_bounds__(kBlock)
ionise_eulerian_kernel(IoniseParams p, const float *__restrict__ nion_dense,
                       const float *__restrict__ xe_fil, const float *__restrict__ density,
                       const float *__restrict__ prev_z_reion,
                       const float *__restrict__ Tneutral, const double *__restrict__ mean_dev,
                       float *__restrict__ xH, float *__restrict__ z_reion,
                       float *__restrict__ Tk, unsigned char *__restrict__ first_cross) {
    const c21hip_ionize_args &a = p.a;
    const double mean_fix = a.fix_mean ? a.mean_f_coll / *mean_dev : 1.;  // :1022-1023
    const float z_now = (float)a.redshift;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < p.nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, p.nz_items, p.zpad_items);
        const auto fc = Pack<VEC>::load(nion_dense, ci.dense);
        Pack<VEC> xe, de;
        if (TS) xe = Pack<VEC>::load(xe_fil, ci.padded);
        if (LAST) de = Pack<VEC>::load(density, ci.dense);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t idx = ci.dense * VEC + e;
            double curr_fcoll = mean_fix * (double)fc.v[e];
            if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;
            const double x_e = TS ? (double)clip_xe(xe.v[e]) : 0.;
            if (curr_fcoll * a.ion_eff_factor > (1. - x_e)) {
                if (first_cross) {
                    if (first_cross[idx] == 0) first_cross[idx] = (unsigned char)a.r_index;
                } else {
                    const float pz = a.first_snapshot ? -1.f : prev_z_reion[idx];
                    z_reion[idx] = (pz < 0.f) ? z_now : pz;
                    xH[idx] = 0.f;
                }
            } else if (LAST) {
                if ((double)xH[idx] > kTiny) {
                    double res_xH = 1. - curr_fcoll * a.ion_eff_factor;
                    if (!a.minimize_memory) {
                        const float T_HI =
                            TS ? Tneutral[idx]
                               : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)de.v[e]));
                        Tk[idx] = partially_ionized_T(T_HI, (float)res_xH, (float)a.T_re);
                    }
                    res_xH -= x_e;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    xH[idx] = (float)res_xH;
                }
            }
        }
    }
}

// Eulerian source models, radius index > 0, no x_e grid: the barrier test of
// find_ionised_regions (IonisationBox.c:1022-1027,1077,1118) on the dense f_coll grid, recording
// only the first crossing (uint8 mask) like the fused Lagrangian path.  4 cells per thread.
// xe_dense (spin-temperature runs): the filtered x_e grid as dense rows; the barrier then is
// f zeta > 1 - x_e with x_e clipped to [0, 0.999] (:811-813,1118).
__global__ void __launch_bounds__(kBlock)
eulerian_mask_kernel(c21hip_ionize_args a, const float *__restrict__ nion_dense,
                     const float *__restrict__ xe_dense, const double *__restrict__ mean_dev,
                     unsigned char *__restrict__ first_cross, size_t ntot) {
    const double mean_fix = a.fix_mean ? a.mean_f_coll / *mean_dev : 1.;
    const size_t n4 = ntot / 4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * kBlock) {
        const float4 f = reinterpret_cast<const float4 *>(nion_dense)[i];
        uchar4 m = reinterpret_cast<const uchar4 *>(first_cross)[i];
        const float fv[4] = {f.x, f.y, f.z, f.w};
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (xe_dense) {
            const float4 x = reinterpret_cast<const float4 *>(xe_dense)[i];
            xv[0] = x.x, xv[1] = x.y, xv[2] = x.z, xv[3] = x.w;
        }
        unsigned char mv[4] = {m.x, m.y, m.z, m.w};
        bool changed = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            double curr_fcoll = mean_fix * (double)fv[e];
            if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;
            const double x_e = xe_dense ? (double)clip_xe(xv[e]) : 0.;
            if (curr_fcoll * a.ion_eff_factor > (1. - x_e) && mv[e] == 0) {
                mv[e] = (unsigned char)a.r_index;
                changed = true;
            }
        }
        if (changed)
            reinterpret_cast<uchar4 *>(first_cross)[i] = make_uchar4(mv[0], mv[1], mv[2], mv[3]);
    }
    // ragged tail (ntot not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        double curr_fcoll = mean_fix * (double)nion_dense[i];
        if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;
        const double x_e = xe_dense ? (double)clip_xe(xe_dense[i]) : 0.;
        if (curr_fcoll * a.ion_eff_factor > (1. - x_e) && first_cross[i] == 0)
            first_cross[i] = (unsigned char)a.r_index;
    }
}

// ---- closed-form Eulerian loop, banded barrier (pass Z EPI 7, fft_native.hip) --------------------
// eulerian_mask_kernel's test on one cell: f (the float the dense grid holds) against the barrier with
// the mean fix `mf`.
__device__ __forceinline__ bool eul_barrier(double mf, float f, int mass_dep_zeta, double f_limit,
                                            double ion_eff) {
    double c = mf * (double)f;
    if (mass_dep_zeta && c < f_limit) c = f_limit;
    return c * ion_eff > (1. - 0.);
}
// The test is monotone in f >= 0 (rounded products of non-negative factors are), so it IS a threshold:
// the smallest non-negative float that passes, found by bisection on the bit pattern (+inf: none does;
// 0: all do).  A sweep then decides a cell with one float comparison instead of the fp64 sequence.
__device__ float eul_threshold(double mf, int mass_dep_zeta, double f_limit, double ion_eff) {
    if (eul_barrier(mf, 0.f, mass_dep_zeta, f_limit, ion_eff)) return 0.f;
    unsigned lo = 0u, hi = 0x7f800000u;  // barrier(lo) false; hi = +inf stands for "none"
    if (!eul_barrier(mf, __uint_as_float(hi), mass_dep_zeta, f_limit, ion_eff)) return __uint_as_float(hi);
    while (hi - lo > 1u) {
        const unsigned mid = lo + (hi - lo) / 2u;
        if (eul_barrier(mf, __uint_as_float(mid), mass_dep_zeta, f_limit, ion_eff))
            hi = mid;
        else
            lo = mid;
    }
    return __uint_as_float(hi);
}
// finish_mean_kernel for the radius r_cur just summed, then (i) thr[r_cur] = the exact threshold of its
// barrier (mean fix mean_f_coll / mean); if r_cur was itself decided on a band: does the exact threshold
// lie between the band's two?  If not the definite decisions of its sweep cannot be trusted: *fail = the
// largest such radius index, and the host reruns the loop from that radius on with the dense sweeps
// (eul_rewind_kernel restores the grid to what it was before that radius); (ii) the band of the next
// radius r_next: ln(mean) extrapolated in ln R from the last two (three, where equally spaced) means,
// widened by min_rel of itself and by 8 x the relative error the same rule made for r_cur; band[2 r] =
// threshold at the lower end of mean_f_coll / mean (cells at or above it cross whatever the exact mean
// turns out to be), band[2 r + 1] = threshold at the upper end (cells below it do not).
struct EulBandArgs {
    double ntot, f_limit, t_cur, t_next, mean_f_coll, ion_eff, min_rel, shift;
    int mass_dep_zeta, r_cur, r_p1, r_p2, r_next, cur_banded, fix_mean;
    int mf_space;  // 1: band[] / thr[] hold the mean fix itself, not thresholds (barriers with an x_e grid)
    int quad;      // 1: the last three radii and the next are equally spaced in ln R: quadratic extrapolation
    double *pred;  // [radius][2]: the mean the linear / the quadratic rule predicted for it (0: none); then
                   // [radius]: the relative error measure in force after it
    double *means, *band, *thr;
    int *fail;
};
__device__ void eul_band_step(double sum_value, const EulBandArgs &a) {
    const double ntot = a.ntot, f_limit = a.f_limit, t_cur = a.t_cur, t_next = a.t_next;
    const double mean_f_coll = a.mean_f_coll, ion_eff = a.ion_eff, min_rel = a.min_rel, shift = a.shift;
    const int mass_dep_zeta = a.mass_dep_zeta, r_cur = a.r_cur, r_p1 = a.r_p1, r_p2 = a.r_p2;
    const int r_next = a.r_next, cur_banded = a.cur_banded, fix_mean = a.fix_mean;
    double *means = a.means, *band = a.band, *thr = a.thr;
    int *fail = a.fail;
    double m = sum_value / ntot;
    if (mass_dep_zeta) {
        if (m <= f_limit) m = f_limit;
    } else {
        if (m <= kFractFloatErr) m = kFractFloatErr;
    }
    means[r_cur] = m;
    if (a.mf_space) {  // the barrier also depends on the cell's x_e: no single threshold; the band is one of mf
        const double mf = fix_mean ? mean_f_coll / m : 1.;
        thr[r_cur] = mf;
        if (cur_banded && !(band[2 * r_cur] <= mf && mf <= band[2 * r_cur + 1]) && *fail < r_cur) *fail = r_cur;
    } else {
        const float t_exact = eul_threshold(fix_mean ? mean_f_coll / m : 1., mass_dep_zeta, f_limit, ion_eff);
        thr[r_cur] = (double)t_exact;
        if (cur_banded) {
            if (!((float)band[2 * r_cur + 1] <= t_exact && t_exact <= (float)band[2 * r_cur]) && *fail < r_cur)
                *fail = r_cur;
        }
    }
    if (r_next < 0) return;
    double lo = 1., hi = 1.;
    if (fix_mean) {
        // ln <f> against ln R: the mean spans decades over the ladder at high redshift (512^3, z = 15: a
        // factor of 50) and bends.  Two rules -- the straight line through the last two points and, where
        // three radii and the next are equally spaced in ln R, the parabola through the last three -- are
        // both evaluated for every radius; the one that predicted the CURRENT radius better is believed for
        // the next (smooth, curved ladders: the parabola; small noisy boxes: the line), and what it was off
        // by is the measure of the band.  (means are > 0: clamped above)
        const double m1 = r_p1 >= 0 ? means[r_p1] : m;
        const double l0 = log(m), l1 = log(m1);
        const double lpl = l0 + (l0 - l1) * t_next;
        const bool have_q = a.quad && r_p2 >= 0;
        const double lpq = have_q ? 3. * l0 - 3. * l1 + log(means[r_p2]) : lpl;
        const double pl = a.pred[2 * r_cur], pq = a.pred[2 * r_cur + 1];
        const double el = pl > 0. ? fabs(m - pl) / m : (double)INFINITY;
        const double eq = pq > 0. ? fabs(m - pq) / m : (double)INFINITY;
        const bool use_q = have_q && eq < el;
        double err = use_q ? eq : el;
        if (!(err < (double)INFINITY)) err = fabs(m - m1) / m;  // no prediction yet: the step itself
        // the curve is not equally smooth everywhere (at high redshift the mean hangs on a few peaks and
        // jumps by several per cent between radii), and an error that happens to be small says little
        // about the next: the measure is the largest recent error, forgotten at 20 % a radius
        if (r_p1 >= 0) err = fmax(err, 0.8 * a.pred[2 * C21CM_MAX_RADII + r_p1]);
        a.pred[2 * C21CM_MAX_RADII + r_cur] = err;
        const double pred = exp(use_q ? lpq : lpl) * (1. + shift);
        a.pred[2 * r_next] = exp(lpl);
        a.pred[2 * r_next + 1] = have_q ? exp(lpq) : 0.;
        err *= fabs(pred);
        const double w = fmax(min_rel * fabs(pred), 8. * err);
        const double mean_lo = pred - w, mean_hi = pred + w;
        lo = mean_f_coll / mean_hi;
        hi = mean_lo > 0. ? mean_f_coll / mean_lo : (double)INFINITY;
    }
    if (a.mf_space) {
        band[2 * r_next] = lo;
        band[2 * r_next + 1] = hi;
        return;
    }
    band[2 * r_next] = (double)eul_threshold(lo, mass_dep_zeta, f_limit, ion_eff);
    band[2 * r_next + 1] = (double)eul_threshold(hi, mass_dep_zeta, f_limit, ion_eff);
}
// c21hip_reduce_sum's arithmetic (1024-partial chunks summed like chunk_reduce_kernel, one workgroup per
// chunk; the chunk sums like finish_reduce_kernel, by the workgroup that arrives last: the same additions
// in the same order, so the same double) followed by eul_band_step: one launch per radius instead of
// three.  n <= 4096: one workgroup, finish_reduce_kernel's order over the partials themselves.
// stage: >= gridDim.x doubles; *counter: 0 on entry, 0 again on exit.
__global__ void __launch_bounds__(kBlock)
eul_sum_band_kernel(const double *__restrict__ partials, int n, double *stage, unsigned *counter,
                    double *sum_out, EulBandArgs a) {
    __shared__ double lds[kBlock];
    __shared__ int is_last;
    const int chunk = 1024;
    const int two_level = gridDim.x > 1;
    if (n > 0) {
        const int lo = two_level ? blockIdx.x * chunk : 0;
        const int hi = two_level ? min(n, lo + chunk) : n;
        double acc = 0.;
        for (int i = lo + threadIdx.x; i < hi; i += kBlock) acc = acc + partials[i];
        lds[threadIdx.x] = acc;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + s];
            __syncthreads();
        }
    }
    if (two_level) {
        if (threadIdx.x == 0) {
            stage[blockIdx.x] = lds[0];
            __threadfence();
            is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
        }
        __syncthreads();
        if (!is_last) return;
        __threadfence();
        double acc = 0.;  // finish_reduce_kernel over the chunk sums
        for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock)
            acc = acc + __builtin_nontemporal_load(stage + i);
        __syncthreads();
        lds[threadIdx.x] = acc;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) lds[threadIdx.x] = lds[threadIdx.x] + lds[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) *counter = 0u;
    }
    if (threadIdx.x == 0) {
        const double total = (n == 0) ? *sum_out : lds[0];
        *sum_out = total;
        eul_band_step(total, a);
    }
}

// First-crossing grid as it was before radius index r_fail of a descending loop: crossings of that and
// of later (smaller) radii and outstanding markers cleared.
__global__ void __launch_bounds__(kBlock)
eul_rewind_kernel(unsigned char *__restrict__ first_cross, int r_fail, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) {
        uint4 w = reinterpret_cast<const uint4 *>(first_cross)[i];
        unsigned wv[4] = {w.x, w.y, w.z, w.w};
        bool ch = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned v = wv[e];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned m = (v >> (8 * k)) & 0xffu;
                if (m != 0 && (m <= (unsigned)r_fail || m == 255u)) {
                    v &= ~(0xffu << (8 * k));
                    ch = true;
                }
            }
            wv[e] = v;
        }
        if (ch) reinterpret_cast<uint4 *>(first_cross)[i] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

// The markers (255) a banded sweep left in the first-crossing grid and no later sweep settled:
// eulerian_mask_kernel's test on those cells with the exact mean of radius r_index, as the threshold
// *thr_dev that test amounts to (eul_band_kernel).
__global__ void __launch_bounds__(kBlock)
eul_resolve_pending_kernel(int r_index, const float *__restrict__ f_pend,
                           const double *__restrict__ thr_dev, unsigned char *__restrict__ first_cross,
                           size_t n16) {
    const float t = (float)*thr_dev;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) {
        uint4 w = reinterpret_cast<const uint4 *>(first_cross)[i];
        unsigned wv[4] = {w.x, w.y, w.z, w.w};
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned v = wv[e];
            if ((v & 0xffu) == 0xffu || (v & 0xff00u) == 0xff00u || (v & 0xff0000u) == 0xff0000u ||
                (v & 0xff000000u) == 0xff000000u)
                any = true;
        }
        if (!any) continue;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned v = wv[e];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (((v >> (8 * k)) & 0xffu) != 0xffu) continue;
                const unsigned r = (f_pend[i * 16 + e * 4 + k] >= t) ? (unsigned)r_index : 0u;
                v = (v & ~(0xffu << (8 * k))) | (r << (8 * k));
            }
            wv[e] = v;
        }
        reinterpret_cast<uint4 *>(first_cross)[i] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

// ... with an x_e grid: the undecided cells left (f_coll, clipped x_e); eulerian_mask_kernel's statements
// with the exact mean fix *mf_dev of radius r_index.
__global__ void __launch_bounds__(kBlock)
eul_resolve_pending_xe_kernel(int r_index, const float *__restrict__ f_pend,
                              const float *__restrict__ xe_pend, const double *__restrict__ mf_dev,
                              int mass_dep_zeta, double f_limit, double ion_eff,
                              unsigned char *__restrict__ first_cross, size_t n16) {
    const double mf = *mf_dev;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) {
        uint4 w = reinterpret_cast<const uint4 *>(first_cross)[i];
        unsigned wv[4] = {w.x, w.y, w.z, w.w};
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned v = wv[e];
            if ((v & 0xffu) == 0xffu || (v & 0xff00u) == 0xff00u || (v & 0xff0000u) == 0xff0000u ||
                (v & 0xff000000u) == 0xff000000u)
                any = true;
        }
        if (!any) continue;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned v = wv[e];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (((v >> (8 * k)) & 0xffu) != 0xffu) continue;
                const size_t cell = i * 16 + e * 4 + k;
                double c = mf * (double)f_pend[cell];
                if (mass_dep_zeta && c < f_limit) c = f_limit;
                const unsigned r = (c * ion_eff > (1. - (double)xe_pend[cell])) ? (unsigned)r_index : 0u;
                v = (v & ~(0xffu << (8 * k))) | (r << (8 * k));
            }
            wv[e] = v;
        }
        reinterpret_cast<uint4 *>(first_cross)[i] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

// ---- find_ionised_regions with a recombination model -------------------------------------------
// IonisationBox.c:1031-1200 with RECOMB_MODEL != none: recombinations per baryon enter the barrier,
//   f zeta > (1 - x_e)(1 + rec),   rec = N_rec / (1 + delta_R)
// with N_rec the previous snapshot's cumulative_recombinations of the cell (CELL_RECOMB; one
// number for the homogeneous model) or that grid filtered at R (:1084-1099), and the FIRST
// crossing of a cell (largest R; neutral_fraction still > 1e-7) records Gamma_12 and the mean free
// path (:1124-1140).  One general kernel for both source-grid kinds (uniform branches; the
// recombination models run the unfused per-radius sequence, so this sweep reads padded real grids).
struct RecombParams {
    IoniseParams ip;
    int lagrangian, inhomo, cell_recomb, ts;
    double R, gamma_prefactor;
};

template <int VEC>
__global__ void __launch_bounds__(kBlock)
ionise_recomb_kernel(RecombParams q, const float *__restrict__ delta_fil,
                     const float *__restrict__ src_grid,   // stars_fil (padded) | nion_dense
                     const float *__restrict__ sfr_fil,    // Lagrangian: filtered whalo_sfr
                     const float *__restrict__ xe_fil, const float *__restrict__ nrec_fil,
                     const float *__restrict__ prev_nrec,  // CELL_RECOMB: dense previous N_rec
                     const float *__restrict__ density, const float *__restrict__ prev_z_reion,
                     const float *__restrict__ Tneutral, const double *__restrict__ mean_dev,
                     float *__restrict__ xH, float *__restrict__ z_reion, float *__restrict__ Tk,
                     float *__restrict__ G12, float *__restrict__ mfp,
                     double *__restrict__ partials) {
    const c21hip_ionize_args &a = q.ip.a;
    const bool LAST = (a.r_index == 0);
    const double mean_fix = (!q.lagrangian && a.fix_mean) ? a.mean_f_coll / *mean_dev : 1.;
    const float z_now = (float)a.redshift;
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < q.ip.nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, q.ip.nz_items, q.ip.zpad_items);
        const auto src = Pack<VEC>::load(src_grid, q.lagrangian ? ci.padded : ci.dense);
        Pack<VEC> dl, xe, de, sf, nr;
        if (!LAST) dl = Pack<VEC>::load(delta_fil, ci.padded);
        if (LAST || !a.minimize_memory) de = Pack<VEC>::load(density, ci.dense);
        if (q.ts) xe = Pack<VEC>::load(xe_fil, ci.padded);
        if (q.lagrangian) sf = Pack<VEC>::load(sfr_fil, ci.padded);
        if (!q.cell_recomb)
            nr = Pack<VEC>::load(nrec_fil, ci.padded);
        else if (q.inhomo)
            nr = Pack<VEC>::load(prev_nrec, ci.dense);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t idx = ci.dense * VEC + e;
            const double curr_dens =
                LAST ? (double)de.v[e] * a.photoncons_factor
                     : (double)(q.lagrangian ? clip_delta(dl.v[e]) : clip_delta_eulerian(dl.v[e]));
            double curr_fcoll;
            if (q.lagrangian) {
                const float stars = fmaxf(src.v[e], 0.f);
                acc += (double)stars;
                curr_fcoll = (double)stars * (1 / (a.rhocrit_omb * (1 + curr_dens)));
            } else {
                curr_fcoll = mean_fix * (double)src.v[e];
            }
            if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;
            double rec;
            if (!q.cell_recomb)
                rec = (double)fmaxf(nr.v[e], 0.f);  // :806-809
            else
                rec = q.inhomo ? (double)nr.v[e] : (double)prev_nrec[0];
            rec /= (1. + curr_dens);
            const double x_e = q.ts ? (double)clip_xe(xe.v[e]) : 0.;
            if (curr_fcoll * a.ion_eff_factor > (1. - x_e) * (1.0 + rec)) {
                if ((double)xH[idx] > kFractFloatErr) {  // first crossing, :1124-1140
                    double g;
                    if (q.lagrangian)
                        g = q.R * q.gamma_prefactor / (1 + curr_dens) * (double)fmaxf(sf.v[e], 0.f);
                    else
                        g = q.R * (q.gamma_prefactor * curr_fcoll);
                    G12[idx] = (float)g;
                    if (mfp) mfp[idx] = (float)q.R;
                }
                const float pz = a.first_snapshot ? -1.f : prev_z_reion[idx];
                z_reion[idx] = (pz < 0.f) ? z_now : pz;
                xH[idx] = 0.f;
            } else if (LAST) {
                if ((double)xH[idx] > kTiny) {
                    double res_xH = 1. - curr_fcoll * a.ion_eff_factor;
                    if (!a.minimize_memory) {
                        const float T_HI =
                            q.ts ? Tneutral[idx]
                                 : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)de.v[e]));
                        Tk[idx] = partially_ionized_T(T_HI, (float)res_xH, (float)a.T_re);
                    }
                    res_xH -= x_e;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    xH[idx] = (float)res_xH;
                }
            }
        }
    }
    block_sum_to(acc, partials);
}

// splined_recombination_rate (recombinations.c:64-92): row z_ct of the table, natural cubic spline
// in ln Gamma evaluated like gsl_interp_cspline (b, d from the c coefficients, Horner in delta)
__device__ __forceinline__ double rr_spline(const double *__restrict__ rr_y,
                                            const double *__restrict__ rr_c, double z_eff,
                                            double gamma12_bg) {
    int z_ct = z_eff > 0 ? (int)(fmin(z_eff, 1e6) / C21CM_RR_DZ + 0.5) : 0;  // NaN -> row 0
    double lnGamma = log(gamma12_bg);
    z_ct = max(0, min(z_ct, C21CM_RR_NZ - 1));
    const double top = C21CM_RR_LNGAMMA_MIN + C21CM_RR_DLNGAMMA * (C21CM_RR_NGAMMA - 1);
    if (lnGamma < C21CM_RR_LNGAMMA_MIN) return 0.;
    if (lnGamma >= top) lnGamma = top - kFractFloatErr;
    const double *y = rr_y + (size_t)z_ct * C21CM_RR_NGAMMA, *c = rr_c + (size_t)z_ct * C21CM_RR_NGAMMA;
    auto knot = [](int g) { return C21CM_RR_LNGAMMA_MIN + g * C21CM_RR_DLNGAMMA; };
    int i = (int)((lnGamma - C21CM_RR_LNGAMMA_MIN) / C21CM_RR_DLNGAMMA);
    i = min(i, C21CM_RR_NGAMMA - 2);
    while (i > 0 && lnGamma < knot(i)) i--;
    while (i < C21CM_RR_NGAMMA - 2 && lnGamma >= knot(i + 1)) i++;
    const double x_lo = knot(i), dx = knot(i + 1) - x_lo, dy = y[i + 1] - y[i];
    const double b = (dy / dx) - dx * (c[i + 1] + 2.0 * c[i]) / 3.0;
    const double d = (c[i + 1] - c[i]) / (3.0 * dx);
    const double delx = lnGamma - x_lo;
    return y[i] + delx * (b + delx * (c[i] + delx * d));
}

// set_recombination_rates, inhomogeneous model (IonisationBox.c:1277-1339): one sweep
__global__ void __launch_bounds__(kBlock)
recomb_rates_kernel(const float *__restrict__ density, const float *__restrict__ G12,
                    const float *__restrict__ xH, const float *__restrict__ prev_nrec,
                    float *__restrict__ nrec, size_t ntot, double stored_redshift, double rate_scale,
                    const double *__restrict__ rr_y, const double *__restrict__ rr_c,
                    int *__restrict__ flag) {
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const double curr_dens = 1.0 + (double)density[i];
        double z_eff = pow(curr_dens, 1.0 / 3.0);
        z_eff *= (1 + stored_redshift);
        const double dNrec =
            rr_spline(rr_y, rr_c, z_eff - 1., (double)G12[i]) * rate_scale * (1. - (double)xH[i]);
        if (!isfinite(dNrec)) bad = 1;
        nrec[i] = (float)((double)prev_nrec[i] + dNrec);
    }
    if (bad) atomicOr(flag, 1);
}

__global__ void __launch_bounds__(kBlock)
sum_float_kernel(const float *__restrict__ v, size_t n, double *__restrict__ partials) {
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        acc += (double)v[i];
    block_sum_to(acc, partials);
}

// ---- post-loop: ionised temperatures, sum(xH), non-finite flag -------------------------
__global__ void __launch_bounds__(kBlock)
finalize_kernel(c21hip_ionize_args a, float stored_z, const float *__restrict__ density,
                const float *__restrict__ Tneutral, const float *__restrict__ xH,
                const float *__restrict__ z_reion, float *__restrict__ Tk, size_t ntot,
                double *__restrict__ partials, int *__restrict__ flag) {
    double acc = 0.;
    int bad = 0;
    const double pow_Tre = pow((double)(float)a.T_re, 1.7);
    const double pow_z = pow(1e4 * ((1. + (double)stored_z) / 4.), 1.7);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const float x = xH[i];
        acc += (double)x;
        if (!a.minimize_memory) {
            const float zr = z_reion[i];
            float T = Tk[i];
            if (zr > 0.f && (double)x < kTiny) {  // IonisationBox.c:1218
                const float dens = density[i];
                T = fully_ionized_T(zr, stored_z, dens, pow_Tre, pow_z);
                const float floorT =
                    a.use_ts_fluct ? Tneutral[i]
                                   : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)dens));
                if (T < floorT) T = floorT;
                Tk[i] = T;
            }
            if (!isfinite(T)) bad = 1;  // :1245
        }
    }
    if (bad) atomicOr(flag, 1);
    block_sum_to(acc, partials);
}

// ---- last step of the fused Lagrangian path: mask of the radii > 0, the cell-scale radius
// (index 0, IonisationBox.c:1031-1200 with LAST_FILTER_STEP) and the post-loop sweep
// (:1203-1256, 1597-1608) in ONE pass over the cells.  Equivalent to apply_first_cross_kernel
// -> ionise_stars_kernel<VEC, true, false> -> finalize_kernel, which move xH / z_reion / T_k
// through HBM three times.  z_reion is written for every cell (-1 where never ionised, the
// value IonisationBox.c:1372-1378 initialises it to), so the caller need not pre-fill it.
// DIRECT: `stars_fil` is the dense emissivity INPUT and the clip of prepare_box_for_filtering
// (0 .. 1e20, IonisationBox.c:1497-1500) is applied on load: at radius index 0 no window is
// applied (:606), so the reference's filtered grid is c2r(r2c(clipped input)) / N -- the
// clipped input itself up to the rounding of the transform pair (~1e-7 relative).  All grids
// are then dense, so rows need no padding arithmetic and VEC = 4 (16-byte accesses) is legal.
// EUL (round 4, late): the same sweep for the Eulerian source models -- stars_fil then is the dense
// f_coll grid of radius index 0 and *mean_dev its box mean: curr_fcoll = f mean_f_coll / mean
// (IonisationBox.c:1022-1027) instead of stars / (rho (1 + delta)); apply_first_cross_kernel +
// ionise_eulerian_kernel<LAST> + finalize_kernel in one pass over the cells.
template <int VEC, bool DIRECT, bool EUL = false>
__global__ void __launch_bounds__(kBlock)
final_sweep_kernel(IoniseParams p, float stored_z, const unsigned char *__restrict__ first_cross,
                   const float *__restrict__ stars_fil, const float *__restrict__ density,
                   const float *__restrict__ prev_z_reion, const float *__restrict__ xe_dense,
                   const float *__restrict__ Tneutral, float *__restrict__ xH,
                   float *__restrict__ z_reion, float *__restrict__ Tk,
                   double *__restrict__ partials_stars, double *__restrict__ partials_xh,
                   int *__restrict__ flag, size_t chunk_items, int chunk0,
                   const double *__restrict__ mean_dev = nullptr) {
    static_assert(!EUL || DIRECT, "the Eulerian sweep reads dense grids");
    double mean_fix = 1.;
    if constexpr (EUL) mean_fix = p.a.fix_mean ? p.a.mean_f_coll / *mean_dev : 1.;  // :1022-1023
    // a.use_ts_fluct (DIRECT only): the x_e input (clipped like its filtered grid, :1504-1507
    // and :1091-1094) enters the barrier and the partial ionisation, the neutral-gas
    // temperature replaces the adiabatic one
    constexpr int U = 2;  // items per thread and trip, loads issued before any arithmetic
    const c21hip_ionize_args &a = p.a;
    const float z_now = (float)a.redshift;
    const double pow_Tre = pow((double)(float)a.T_re, 1.7);
    const double pow_z = pow(1e4 * ((1. + (double)stored_z) / 4.), 1.7);
    double acc_s = 0., acc_x = 0.;
    int bad = 0;
    // Workgroup b sweeps the CONTIGUOUS item range [chunk b, chunk b + 1) (round 5): a chunk's partial
    // sums then depend on the chunk alone, not on how many workgroups the launch has -- a rank that
    // sweeps only its slab of chunks (c21cm_ionize_shard_finish_slab) leaves the very partials the
    // single pass leaves there, and the fixed-order reduce over all chunks gives the same sums to the
    // last bit whoever computed which chunk.
    const int chunk = chunk0 + (int)blockIdx.x;
    const size_t c_begin = (size_t)chunk * chunk_items;
    const size_t c_end = (c_begin + chunk_items < p.nitems) ? c_begin + chunk_items : p.nitems;
    for (size_t i0 = c_begin + threadIdx.x; i0 < c_end; i0 += (size_t)kBlock * U) {
        Pack<VEC> st[U], de[U], x0[U], T0[U], pz[U], xe[U], Tn[U];
        unsigned char m[U][VEC];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + (size_t)u * kBlock;
            ok[u] = i < c_end;
            if (!ok[u]) continue;
            const auto ci = DIRECT ? CellIndex<VEC>{i, i} : cell_index<VEC>(i, p.nz_items, p.zpad_items);
            st[u] = Pack<VEC>::load(stars_fil, ci.padded);
            de[u] = Pack<VEC>::load(density, ci.dense);
            x0[u] = Pack<VEC>::load(xH, ci.dense);
            if (!a.minimize_memory) T0[u] = Pack<VEC>::load(Tk, ci.dense);
            if (!a.first_snapshot) pz[u] = Pack<VEC>::load(prev_z_reion, ci.dense);
            if (DIRECT && a.use_ts_fluct) {
                xe[u] = Pack<VEC>::load(xe_dense, ci.dense);
                if (!a.minimize_memory) Tn[u] = Pack<VEC>::load(Tneutral, ci.dense);
            }
            if (VEC == 4) {
                const uchar4 m4 = reinterpret_cast<const uchar4 *>(first_cross)[ci.dense];
                m[u][0] = m4.x;
                m[u][1] = m4.y;
                m[u][2 % VEC] = m4.z;
                m[u][3 % VEC] = m4.w;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; e++) m[u][e] = first_cross[ci.dense * VEC + e];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!ok[u]) continue;
            const size_t i = i0 + (size_t)u * kBlock;
            Pack<VEC> xo, zo, To;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const float dens = de[u].v[e];
                double curr_fcoll;
                if constexpr (EUL) {
                    curr_fcoll = mean_fix * (double)st[u].v[e];
                } else {
                    float stars = fmaxf(st[u].v[e], 0.f);  // IonisationBox.c:822-823
                    if (DIRECT) stars = fminf(stars, 1e20f);
                    acc_s += (double)stars;
                    const double curr_dens = (double)dens * a.photoncons_factor;  // :1048
                    curr_fcoll = (double)stars;
                    curr_fcoll *= 1 / (a.rhocrit_omb * (1 + curr_dens));  // :1066-1067
                }
                if (a.mass_dep_zeta && curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;  // :1077
                float x = x0[u].v[e];
                float T = a.minimize_memory ? 0.f : T0[u].v[e];
                float zr = -1.f;
                const bool ts = DIRECT && a.use_ts_fluct;
                const double x_e = ts ? (double)clip_xe(xe[u].v[e]) : 0.;
                const float T_neutral =
                    (ts && !a.minimize_memory)
                        ? Tn[u].v[e]
                        : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)dens));
                const bool ionised =
                    m[u][e] != 0 || (curr_fcoll * a.ion_eff_factor > (1. - x_e));  // :1118
                if (ionised) {
                    const float pzv = a.first_snapshot ? -1.f : pz[u].v[e];
                    zr = (pzv < 0.f) ? z_now : pzv;  // :1143-1147
                    x = 0.f;                         // :1151
                } else if ((double)x > kTiny) {      // :1161
                    double res_xH = 1. - curr_fcoll * a.ion_eff_factor;
                    if (!a.minimize_memory)
                        T = partially_ionized_T(T_neutral, (float)res_xH, (float)a.T_re);
                    res_xH -= x_e;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    x = (float)res_xH;
                }
                acc_x += (double)x;
                if (!a.minimize_memory) {
                    if (zr > 0.f && (double)x < kTiny) {  // :1218
                        T = fully_ionized_T(zr, stored_z, dens, pow_Tre, pow_z);
                        if (T < T_neutral) T = T_neutral;
                    }
                    if (!isfinite(T)) bad = 1;  // :1245
                }
                xo.v[e] = x;
                zo.v[e] = zr;
                To.v[e] = T;
            }
            xo.store(xH, i);
            zo.store(z_reion, i);
            if (!a.minimize_memory) To.store(Tk, i);
        }
    }
    if (bad) atomicOr(flag, 1);
    block_sum_to(acc_s, partials_stars + chunk0);
    __syncthreads();
    block_sum_to(acc_x, partials_xh + chunk0);
}

__global__ void __launch_bounds__(kBlock)
apply_first_cross_kernel(const unsigned char *__restrict__ fc,
                         const float *__restrict__ prev_z_reion, int first_snapshot, float z_now,
                         float *__restrict__ xH, float *__restrict__ z_reion, size_t ntot) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        if (fc[i]) {
            const float pz = first_snapshot ? -1.f : prev_z_reion[i];
            z_reion[i] = (pz < 0.f) ? z_now : pz;
            xH[i] = 0.f;
        }
    }
}

// fused recombination loop: the mask holds the radius index of the first crossing, Gamma_12 was
// written by the fused pass Z; what is left is x_HI, z_reion and the mean free path R[index]
__global__ void __launch_bounds__(kBlock)
apply_first_cross_recomb_kernel(const unsigned char *__restrict__ fc, const float *__restrict__ R_dev,
                                const float *__restrict__ prev_z_reion, int first_snapshot,
                                float z_now, float *__restrict__ xH, float *__restrict__ z_reion,
                                float *__restrict__ mfp, size_t ntot) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const unsigned char r = fc[i];
        if (r) {
            const float pz = first_snapshot ? -1.f : prev_z_reion[i];
            z_reion[i] = (pz < 0.f) ? z_now : pz;
            xH[i] = 0.f;
            if (mfp) mfp[i] = R_dev[r];
        }
    }
}

// R-loop sharding with a recombination model: a rank's first crossing of a cell is the pair
// (mean free path = R of the crossing, Gamma_12 at it).  Both are non-negative floats, whose IEEE
// bit patterns order like the values, so key = bits(mfp) << 32 | bits(G12) and ONE max-reduce of
// 64-bit keys over the ranks selects the largest radius that ionised the cell together with ITS
// Gamma_12 (a radius belongs to exactly one rank: no ties).  key = 0: never crossed.
__global__ void __launch_bounds__(kBlock)
pack_cross_keys_kernel(const float *__restrict__ mfp, const float *__restrict__ G12,
                       unsigned long long *__restrict__ keys, size_t ntot) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const float m = mfp[i];
        keys[i] = (m > 0.f) ? (((unsigned long long)__float_as_uint(m) << 32) |
                               (unsigned long long)__float_as_uint(fmaxf(G12[i], 0.f)))
                            : 0ull;
    }
}

// the reduced keys -> the state find_ionised_regions leaves after the radii > 0
// (IonisationBox.c:1124-1151): crossed cells are ionised, carry z_reion, Gamma_12, mean free path
__global__ void __launch_bounds__(kBlock)
apply_cross_keys_kernel(const unsigned long long *__restrict__ keys,
                        const float *__restrict__ prev_z_reion, int first_snapshot, float z_now,
                        float *__restrict__ xH, float *__restrict__ z_reion,
                        float *__restrict__ G12, float *__restrict__ mfp, size_t ntot) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const unsigned long long k = keys[i];
        if (k) {
            const float pz = first_snapshot ? -1.f : prev_z_reion[i];
            z_reion[i] = (pz < 0.f) ? z_now : pz;
            xH[i] = 0.f;
            G12[i] = __uint_as_float((unsigned)(k & 0xffffffffull));
            if (mfp) mfp[i] = __uint_as_float((unsigned)(k >> 32));
        }
    }
}

// The exchange of a sharded run without recombinations only needs "did any radius > 0 ionise the
// cell" (the finish phase tests the grid for non-zero): one BIT per cell.  pack: 32 cells -> one
// word (ballot of a half wave would do the same; a word per thread keeps it launch-shape free);
// or_unpack: the OR of `world` packed grids back into the uint8 grid the finish phase reads.
__global__ void __launch_bounds__(kBlock)
pack_mask_bits_kernel(const unsigned char *__restrict__ fc, unsigned *__restrict__ bits,
                      size_t nwords, size_t ntot) {
    for (size_t w = (size_t)blockIdx.x * kBlock + threadIdx.x; w < nwords;
         w += (size_t)gridDim.x * kBlock) {
        const size_t base = w * 32;
        unsigned v = 0;
        if (base + 32 <= ntot) {
            const uint4 a = reinterpret_cast<const uint4 *>(fc + base)[0];
            const uint4 b = reinterpret_cast<const uint4 *>(fc + base)[1];
            const unsigned q[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) v |= ((q[i] >> (8 * j)) & 0xffu) ? (1u << (4 * i + j)) : 0u;
        } else {
            for (int i = 0; i < 32 && base + i < ntot; i++) v |= fc[base + i] ? (1u << i) : 0u;
        }
        bits[w] = v;
    }
}

__global__ void __launch_bounds__(kBlock)
or_unpack_mask_bits_kernel(const unsigned *__restrict__ bits, size_t stride_words, int world,
                           unsigned char *__restrict__ fc, size_t nwords, size_t ntot) {
    for (size_t w = (size_t)blockIdx.x * kBlock + threadIdx.x; w < nwords;
         w += (size_t)gridDim.x * kBlock) {
        unsigned v = 0;
        for (int r = 0; r < world; r++) v |= bits[(size_t)r * stride_words + w];
        const size_t base = w * 32;
        if (base + 32 <= ntot) {
            unsigned q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                q[i] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) q[i] |= ((v >> (4 * i + j)) & 1u) << (8 * j);
            }
            reinterpret_cast<uint4 *>(fc + base)[0] = make_uint4(q[0], q[1], q[2], q[3]);
            reinterpret_cast<uint4 *>(fc + base)[1] = make_uint4(q[4], q[5], q[6], q[7]);
        } else {
            for (int i = 0; i < 32 && base + i < ntot; i++) fc[base + i] = (v >> i) & 1u;
        }
    }
}

// set_fully_neutral_box: IonisationBox.c:531-565
__global__ void __launch_bounds__(kBlock)
neutral_box_kernel(const float *__restrict__ density, const float *__restrict__ xe,
                   const float *__restrict__ Tneutral, float *__restrict__ xH,
                   float *__restrict__ Tk, size_t ntot, int ts, double global_xH, double TK,
                   double adia) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        if (ts) {
            xH[i] = (float)(1. - (double)xe[i]);
            if (Tk) Tk[i] = Tneutral[i];
        } else {
            xH[i] = (float)global_xH;
            if (Tk) Tk[i] = (float)(TK * (1.0 + adia * (double)density[i]));
        }
    }
}

__global__ void __launch_bounds__(kBlock)
any_nonzero_kernel(const float *__restrict__ a, size_t n, int *flag) {
    int found = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock)
        if (a[i] != 0.f) found = 1;
    if (found) atomicOr(flag, 1);
}

// ---- ComputeBrightnessTemp: BrightnessTemperatureBox.c:58-87 --------------------------------
__global__ void __launch_bounds__(kBlock)
brightness_kernel(const float *__restrict__ density, const float *__restrict__ xH,
                  const float *__restrict__ Ts, float *__restrict__ bt, float *__restrict__ tau,
                  size_t n, float const_factor, float T_rad, double redshift, int use_ts,
                  double *__restrict__ partials) {
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        // float arithmetic, left to right: (const_factor * x_HI) * (1 + delta)
        float v = __fmul_rn(__fmul_rn(const_factor, xH[i]), __fadd_rn(1.f, density[i]));
        if (use_ts) {
            const float ts = Ts[i];
            v = (float)((double)v * ((1. + redshift) / (1000. * (double)ts)));
            tau[i] = v;
            v = (float)((1. - exp(-(double)v)) * 1000. * (double)__fsub_rn(ts, T_rad) /
                        (1. + redshift));
        }
        bt[i] = v;
        acc += (double)v;
    }
    block_sum_to(acc, partials);
}

IoniseParams make_params(const c21hip_ionize_args *a, int vec) {
    IoniseParams p;
    p.a = *a;
    const int zpad = 2 * (a->nz / 2 + 1);
    p.nz_items = a->nz / vec;
    p.zpad_items = zpad / vec;
    p.nitems = (size_t)a->nx * a->ny * p.nz_items;
    return p;
}
}  // namespace

extern "C" int c21hip_clip_minmax(float *delta_fil, int nx, int ny, int nz, double *partials,
                                  double *minmax_out, void *stream) {
    const int vec = (nz % 2 == 0) ? 2 : 1;
    const int zpad = 2 * (nz / 2 + 1);
    const size_t nitems = (size_t)nx * ny * (nz / vec);
    const int blocks = grid_for(nitems);
    double *pmin = partials, *pmax = partials + kMaxBlocks;
    if (vec == 2)
        hipLaunchKernelGGL(minmax_kernel<2>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           delta_fil, nitems, nz / 2, zpad / 2, pmin, pmax);
    else
        hipLaunchKernelGGL(minmax_kernel<1>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           delta_fil, nitems, nz, zpad, pmin, pmax);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, pmin,
                       blocks, 1, minmax_out);
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, pmax,
                       blocks, 2, minmax_out + 1);
    LAUNCH_CHECK();
    return 0;
}

// zstride: row length of delta_fil in floats -- 2 (nz / 2 + 1) (padded rows, the transforms' layout) or nz
// (dense rows: with nz a multiple of 4 the sweep then moves 16-byte pieces; the table loop writes its
// delta_R that way, round 5).  c21hip_fcoll_eulerian: padded rows.
extern "C" int c21hip_fcoll_eulerian_zs(const float *delta_fil, long zstride, float *nion_dense, int nx, int ny,
                                        int nz, int mode, double growthf, double sigma_min,
                                        double sigma_max, double delta_c, double tab_min,
                                        double tab_width, const float *table_dev, double *partials,
                                        double *sum_out, void *stream);
extern "C" int c21hip_fcoll_eulerian(const float *delta_fil, float *nion_dense, int nx, int ny,
                                     int nz, int mode, double growthf, double sigma_min,
                                     double sigma_max, double delta_c, double tab_min,
                                     double tab_width, const float *table_dev, double *partials,
                                     double *sum_out, void *stream) {
    return c21hip_fcoll_eulerian_zs(delta_fil, 2 * (long)(nz / 2 + 1), nion_dense, nx, ny, nz, mode, growthf,
                                    sigma_min, sigma_max, delta_c, tab_min, tab_width, table_dev, partials,
                                    sum_out, stream);
}
extern "C" int c21hip_fcoll_eulerian_zs(const float *delta_fil, long zstride, float *nion_dense, int nx, int ny,
                                        int nz, int mode, double growthf, double sigma_min,
                                        double sigma_max, double delta_c, double tab_min,
                                        double tab_width, const float *table_dev, double *partials,
                                        double *sum_out, void *stream) {
    FcollParams fp;
    fp.mode = mode;
    fp.growthf = (float)growthf;
    fp.delta_c = delta_c;
    fp.tab_min = tab_min;
    fp.tab_width = tab_width;
    fp.sig = -1.;
    if (mode == C21CM_FCOLL_ERFC) {
        // hmf.c:1221-1232: float sigmas, float products, double sqrt
        const float ss = (float)sigma_min, sl = (float)sigma_max;
        if (sl > ss) {
            c21hip_set_error("FgtrM requested in a region where M_min > M_max (sigma %g > %g)",
                             (double)sl, (double)ss);
            return C21CM_VALUE_ERROR;
        }
        if (sl != ss) {
            const float d = ss * ss - sl * sl;
            fp.sig = sqrt((double)d);
        }
    }
    const bool dense4 = (zstride == (long)nz && nz % 4 == 0 && mode != C21CM_FCOLL_NODES);
    if (zstride != (long)nz && zstride != 2 * (long)(nz / 2 + 1)) {
        c21hip_set_error("f_coll sweep: rows of %ld floats for nz = %d", zstride, nz);
        return C21CM_VALUE_ERROR;
    }
    const int vec = dense4 ? 4 : ((nz % 2 == 0) ? 2 : 1);
    const int zpad = (int)zstride;
    const size_t nitems = (size_t)nx * ny * (nz / vec);
    if (mode == C21CM_FCOLL_NODES) {  // table_dev: C21CM_NODE_DOUBLES doubles of the radius
        const int nb = grid_for(nitems);
        const double *nodes = reinterpret_cast<const double *>(table_dev);
        if (vec == 2)
            hipLaunchKernelGGL(fcoll_nodes_kernel<2>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                               delta_fil, nion_dense, nitems, nz / 2, zpad / 2, nodes, partials);
        else
            hipLaunchKernelGGL(fcoll_nodes_kernel<1>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                               delta_fil, nion_dense, nitems, nz, zpad, nodes, partials);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                           partials, nb, 0, sum_out);
        LAUNCH_CHECK();
        return 0;
    }
    const int blocks = grid_for((nitems + 3) / 4);
    if (vec == 4)
        hipLaunchKernelGGL(fcoll_eulerian_kernel<4>, dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, delta_fil, nion_dense, nitems, nz / 4, zpad / 4, fp,
                           table_dev, partials);
    else if (vec == 2)
        hipLaunchKernelGGL(fcoll_eulerian_kernel<2>, dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, delta_fil, nion_dense, nitems, nz / 2, zpad / 2, fp,
                           table_dev, partials);
    else
        hipLaunchKernelGGL(fcoll_eulerian_kernel<1>, dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, delta_fil, nion_dense, nitems, nz, zpad, fp,
                           table_dev, partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sum_out);
    LAUNCH_CHECK();
    return 0;
}

// Table modes with the banded barrier (fcoll_eulerian_band_kernel): *n_partials_out partial sums are left
// in `partials` for c21hip_eul_band (the order of c21hip_fcoll_eulerian's own reduction).
extern "C" int c21hip_fcoll_eulerian_band(const float *delta_fil, long zstride, float *f_pend,
                                          unsigned char *first_cross,
                                          int nx, int ny, int nz, int mode, double tab_min,
                                          double tab_width, const float *table_dev, const double *band_dev,
                                          const double *thr_prev_dev, int r_index, int r_prev,
                                          double *partials, int *n_partials_out, void *stream) {
    if ((zstride != (long)nz && zstride != 2 * (long)(nz / 2 + 1)) || nz % 2 || (mode != C21CM_FCOLL_TABLE_LINEAR && mode != C21CM_FCOLL_TABLE_EXP) || r_index <= 0 ||
        r_index >= 255 || r_prev >= 255) {
        c21hip_set_error("banded barrier of a table mode: unsupported box, mode or radius index");
        return C21CM_VALUE_ERROR;
    }
    FcollParams fp;
    fp.mode = mode;
    fp.growthf = 0.f;
    fp.delta_c = 0.;
    fp.tab_min = tab_min;
    fp.tab_width = tab_width;
    fp.sig = -1.;
    // (the same items-to-threads map as c21hip_fcoll_eulerian_zs on the same rows: the partial sums of the
    //  banded and the dense sweep of a radius add up in the same order)
    const bool dense4 = (zstride == (long)nz && nz % 4 == 0);
    const int vec = dense4 ? 4 : 2;
    const int zpad = (int)zstride;
    const size_t nitems = (size_t)nx * ny * (nz / vec);
    const int blocks = grid_for((nitems + 3) / 4);
    if (dense4)
        hipLaunchKernelGGL(fcoll_eulerian_band_kernel<4>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           delta_fil, f_pend, first_cross, nitems, nz / 4, zpad / 4, fp, table_dev, band_dev,
                           thr_prev_dev, r_index, r_prev, partials);
    else
        hipLaunchKernelGGL(fcoll_eulerian_band_kernel<2>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           delta_fil, f_pend, first_cross, nitems, nz / 2, zpad / 2, fp, table_dev, band_dev,
                           thr_prev_dev, r_index, r_prev, partials);
    LAUNCH_CHECK();
    *n_partials_out = blocks;
    return 0;
}

// chunked first level for long partial arrays: block b sums partials[b*chunk .. ), fixed order
__global__ void __launch_bounds__(kBlock)
chunk_reduce_kernel(const double *__restrict__ partials, int n, int chunk, int op,
                    double *__restrict__ out) {
    __shared__ double lds[kBlock];
    const int lo = blockIdx.x * chunk;
    const int hi = min(n, lo + chunk);
    double acc = (op == 0) ? 0. : partials[lo];
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
        const double p = partials[i];
        acc = (op == 0) ? acc + p : (op == 1 ? fmin(acc, p) : fmax(acc, p));
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double a = lds[threadIdx.x], b = lds[threadIdx.x + s];
            lds[threadIdx.x] = (op == 0) ? a + b : (op == 1 ? fmin(a, b) : fmax(a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

// Deterministic sum of n doubles.  Long inputs are reduced in place in two levels (the
// first level overwrites partials[0 .. n/1024)).
extern "C" int c21hip_reduce_sum(const double *partials, int n, double *out, void *stream) {
    if (n > 4096) {
        const int chunk = 1024;
        const int nb = (n + chunk - 1) / chunk;
        // level 1 writes into the tail-safe front of the same buffer: block b only reads
        // indices >= b*1024 and writes index b <= its own first read, after all reads of
        // lower blocks' ranges are irrelevant to it -- but to stay race-free regardless of
        // scheduling, stage through the second half of the scratch instead
        double *stage = const_cast<double *>(partials) + n;
        hipLaunchKernelGGL(chunk_reduce_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           partials, n, chunk, 0, stage);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                           stage, nb, 0, out);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, n, 0, out);
    LAUNCH_CHECK();
    return 0;
}

// op 1 = min, 2 = max over n doubles; `stage` (>= n/1024 + 1 doubles) is used for long inputs
extern "C" int c21hip_reduce_op(const double *partials, int n, int op, double *stage, double *out,
                                void *stream) {
    if (n > 4096 && stage) {
        const int chunk = 1024;
        const int nb = (n + chunk - 1) / chunk;
        hipLaunchKernelGGL(chunk_reduce_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           partials, n, chunk, op, stage);
        LAUNCH_CHECK();
        partials = stage;
        n = nb;
    }
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, n, op, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_finish_mean(const double *sum_dev, double ntot, int mass_dep_zeta,
                                  double f_limit, double *mean_dev, void *stream) {
    hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sum_dev, ntot,
                       mass_dep_zeta, f_limit, mean_dev);
    LAUNCH_CHECK();
    return 0;
}

#define DISPATCH_IONISE(KERNEL, ...)                                                          \
    do {                                                                                      \
        const bool last = (a->r_index == 0);                                                  \
        const bool ts = a->use_ts_fluct != 0;                                                 \
        if (vec == 2) {                                                                       \
            if (last && ts)                                                                   \
                hipLaunchKernelGGL((KERNEL<2, true, true>), dim3(blocks), dim3(kBlock), 0,    \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else if (last)                                                                    \
                hipLaunchKernelGGL((KERNEL<2, true, false>), dim3(blocks), dim3(kBlock), 0,   \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else if (ts)                                                                      \
                hipLaunchKernelGGL((KERNEL<2, false, true>), dim3(blocks), dim3(kBlock), 0,   \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else                                                                              \
                hipLaunchKernelGGL((KERNEL<2, false, false>), dim3(blocks), dim3(kBlock), 0,  \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
        } else {                                                                              \
            if (last && ts)                                                                   \
                hipLaunchKernelGGL((KERNEL<1, true, true>), dim3(blocks), dim3(kBlock), 0,    \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else if (last)                                                                    \
                hipLaunchKernelGGL((KERNEL<1, true, false>), dim3(blocks), dim3(kBlock), 0,   \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else if (ts)                                                                      \
                hipLaunchKernelGGL((KERNEL<1, false, true>), dim3(blocks), dim3(kBlock), 0,   \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
            else                                                                              \
                hipLaunchKernelGGL((KERNEL<1, false, false>), dim3(blocks), dim3(kBlock), 0,  \
                                   (hipStream_t)stream, __VA_ARGS__);                         \
        }                                                                                     \
    } while (0)

extern "C" int c21hip_ionise_stars(const c21hip_ionize_args *a, const float *delta_fil,
                                   const float *stars_fil, const float *xe_fil,
                                   const float *density, const float *prev_z_reion,
                                   const float *kinetic_temp_neutral, float *xH, float *z_reion,
                                   float *kinetic_temperature, unsigned char *first_cross,
                                   double *partials, double *sum_out, void *stream) {
    const int vec = (a->nz % 2 == 0) ? 2 : 1;
    const IoniseParams p = make_params(a, vec);
    const int blocks = grid_for(p.nitems);
    DISPATCH_IONISE(ionise_stars_kernel, p, delta_fil, stars_fil, xe_fil, density, prev_z_reion,
                    kinetic_temp_neutral, xH, z_reion, kinetic_temperature, first_cross, partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sum_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ionise_eulerian(const c21hip_ionize_args *a, const float *nion_dense,
                                      const float *xe_fil, const float *density,
                                      const float *prev_z_reion,
                                      const float *kinetic_temp_neutral, const double *mean_dev,
                                      float *xH, float *z_reion, float *kinetic_temperature,
                                      unsigned char *first_cross, void *stream) {
    const int vec = (a->nz % 2 == 0) ? 2 : 1;
    const IoniseParams p = make_params(a, vec);
    const int blocks = grid_for(p.nitems);
    DISPATCH_IONISE(ionise_eulerian_kernel, p, nion_dense, xe_fil, density, prev_z_reion,
                    kinetic_temp_neutral, mean_dev, xH, z_reion, kinetic_temperature,
                    first_cross);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ionise_recomb(const c21hip_ionize_args *a, int lagrangian, int inhomo,
                                    int cell_recomb, double R, double gamma_prefactor,
                                    const float *delta_fil, const float *src_grid,
                                    const float *sfr_fil, const float *xe_fil,
                                    const float *nrec_fil, const float *prev_nrec,
                                    const float *density, const float *prev_z_reion,
                                    const float *kinetic_temp_neutral, const double *mean_dev,
                                    float *xH, float *z_reion, float *kinetic_temperature,
                                    float *G12, float *mfp, double *partials, double *sum_out,
                                    void *stream) {
    const int vec = (a->nz % 2 == 0) ? 2 : 1;
    RecombParams q;
    q.ip = make_params(a, vec);
    q.lagrangian = lagrangian;
    q.inhomo = inhomo;
    q.cell_recomb = cell_recomb;
    q.ts = a->use_ts_fluct;
    q.R = R;
    q.gamma_prefactor = gamma_prefactor;
    const int blocks = grid_for(q.ip.nitems);
    if (vec == 2)
        hipLaunchKernelGGL((ionise_recomb_kernel<2>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, q, delta_fil, src_grid, sfr_fil, xe_fil, nrec_fil,
                           prev_nrec, density, prev_z_reion, kinetic_temp_neutral, mean_dev, xH,
                           z_reion, kinetic_temperature, G12, mfp, partials);
    else
        hipLaunchKernelGGL((ionise_recomb_kernel<1>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, q, delta_fil, src_grid, sfr_fil, xe_fil, nrec_fil,
                           prev_nrec, density, prev_z_reion, kinetic_temp_neutral, mean_dev, xH,
                           z_reion, kinetic_temperature, G12, mfp, partials);
    LAUNCH_CHECK();
    if (sum_out) {
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                           partials, blocks, 0, sum_out);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int c21hip_recomb_rates(const float *density, const float *G12, const float *xH,
                                   const float *prev_nrec, float *nrec, size_t ntot,
                                   double stored_redshift, double rate_scale, const double *rr_y_dev,
                                   const double *rr_c_dev, int *flag_dev, void *stream) {
    hipLaunchKernelGGL(recomb_rates_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, density, G12, xH, prev_nrec, nrec, ntot,
                       stored_redshift, rate_scale, rr_y_dev, rr_c_dev, flag_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_sum_float(const float *v, size_t n, double *partials, double *sum_out,
                                void *stream) {
    const int blocks = grid_for(n);
    hipLaunchKernelGGL(sum_float_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, v, n,
                       partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sum_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_eulerian_mask(const c21hip_ionize_args *a, const float *nion_dense,
                                    const float *xe_dense, const double *mean_dev,
                                    unsigned char *first_cross, void *stream) {
    const size_t ntot = (size_t)a->nx * a->ny * a->nz;
    hipLaunchKernelGGL(eulerian_mask_kernel, dim3(grid_for(ntot / 4 + 1)), dim3(kBlock), 0,
                       (hipStream_t)stream, *a, nion_dense, xe_dense, mean_dev, first_cross, ntot);
    LAUNCH_CHECK();
    return 0;
}

// partials != NULL: the n per-workgroup partial sums of the radius' pass Z are reduced here (the
// arithmetic of c21hip_reduce_sum; its staging area is partials + n, like there) into *sum_dev first;
// NULL: *sum_dev already holds the sum.  counter_dev: a zeroed word the launches share.
extern "C" int c21hip_eul_band(const double *partials, int n, double *sum_dev, double ntot,
                               int mass_dep_zeta, double f_limit, double *means_dev, int r_cur, int r_p1,
                               int r_p2, double t_cur, double t_next, int r_next, int cur_banded,
                               int fix_mean, double mean_f_coll, double ion_eff, double min_rel,
                               double shift, double *band_dev, double *thr_dev, int *fail_dev,
                               unsigned *counter_dev, int mf_space, int quad, double *pred_dev,
                               void *stream) {
    EulBandArgs a;
    a.mf_space = mf_space;
    a.quad = quad;
    a.pred = pred_dev;
    a.ntot = ntot, a.f_limit = f_limit, a.t_cur = t_cur, a.t_next = t_next, a.mean_f_coll = mean_f_coll;
    a.ion_eff = ion_eff, a.min_rel = min_rel, a.shift = shift, a.mass_dep_zeta = mass_dep_zeta;
    a.r_cur = r_cur, a.r_p1 = r_p1, a.r_p2 = r_p2, a.r_next = r_next, a.cur_banded = cur_banded;
    a.fix_mean = fix_mean, a.means = means_dev, a.band = band_dev, a.thr = thr_dev, a.fail = fail_dev;
    const int nn = partials ? n : 0;
    const int nb = nn > 4096 ? (nn + 1023) / 1024 : 1;
    hipLaunchKernelGGL(eul_sum_band_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, partials, nn,
                       const_cast<double *>(partials) + nn, counter_dev, sum_dev, a);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_eul_rewind(unsigned char *first_cross, int r_fail, size_t ntot, void *stream) {
    if (ntot % 16) {
        c21hip_set_error("banded barrier: the box is not a multiple of 16 cells");
        return C21CM_VALUE_ERROR;
    }
    hipLaunchKernelGGL(eul_rewind_kernel, dim3(grid_for(ntot / 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       first_cross, r_fail, ntot / 16);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_eul_resolve_pending(int r_index, const float *f_pend, const double *thr_dev,
                                          unsigned char *first_cross, size_t ntot, void *stream) {
    if (ntot % 16) {
        c21hip_set_error("banded barrier: the box is not a multiple of 16 cells");
        return C21CM_VALUE_ERROR;
    }
    hipLaunchKernelGGL(eul_resolve_pending_kernel, dim3(grid_for(ntot / 16)), dim3(kBlock), 0,
                       (hipStream_t)stream, r_index, f_pend, thr_dev, first_cross, ntot / 16);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_eul_resolve_pending_xe(int r_index, const float *f_pend, const float *xe_pend,
                                             const double *mf_dev, int mass_dep_zeta, double f_limit,
                                             double ion_eff, unsigned char *first_cross, size_t ntot,
                                             void *stream) {
    if (ntot % 16) {
        c21hip_set_error("banded barrier: the box is not a multiple of 16 cells");
        return C21CM_VALUE_ERROR;
    }
    hipLaunchKernelGGL(eul_resolve_pending_xe_kernel, dim3(grid_for(ntot / 16)), dim3(kBlock), 0,
                       (hipStream_t)stream, r_index, f_pend, xe_pend, mf_dev, mass_dep_zeta, f_limit, ion_eff,
                       first_cross, ntot / 16);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_finalize(const c21hip_ionize_args *a, double stored_redshift,
                               const float *density, const float *kinetic_temp_neutral,
                               const float *xH, const float *z_reion, float *kinetic_temperature,
                               size_t ntot, double *partials, double *sum_out, int *flag_out,
                               void *stream) {
    const int blocks = grid_for(ntot);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, *a,
                       (float)stored_redshift, density, kinetic_temp_neutral, xH, z_reion,
                       kinetic_temperature, ntot, partials, flag_out);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sum_out);
    LAUNCH_CHECK();
    return 0;
}

// The final sweep's chunks (round 5): workgroup b owns the contiguous items [b, b + 1) * chunk_items.
// The chunking depends on the box alone, so a sharded finish (every rank sweeps a slab of whole chunks,
// c21cm_ionize_shard_finish_slab) leaves chunk partials identical to the single pass', and one
// fixed-order reduce over all of them gives the same sums whoever swept which chunk.
namespace {
struct FinalChunks {
    size_t chunk_items;
    int n_chunks, vec;
    size_t nitems;
};
FinalChunks final_chunks(const c21hip_ionize_args *a, int dense) {
    FinalChunks c;
    const size_t ntot = (size_t)a->nx * a->ny * a->nz;
    c.vec = dense ? ((ntot % 4 == 0) ? 4 : 1) : ((a->nz % 2 == 0) ? 2 : 1);
    c.nitems = dense ? ntot / c.vec : (size_t)a->nx * a->ny * (a->nz / c.vec);
    const int blocks = grid_for((c.nitems + 1) / 2);
    const size_t q = (size_t)kBlock * 2;  // one trip of a workgroup (U = 2)
    c.chunk_items = ((c.nitems + blocks - 1) / blocks + q - 1) / q * q;
    c.n_chunks = (int)((c.nitems + c.chunk_items - 1) / c.chunk_items);
    return c;
}
}  // namespace

// chunk count and cells per chunk of the final sweep of this box (dense: the sweep reads dense grids --
// the direct Lagrangian sweep and the Eulerian one); the last chunk may be short
extern "C" int c21hip_final_sweep_chunks(const c21hip_ionize_args *a, int dense, int *n_chunks,
                                         size_t *chunk_cells) {
    const FinalChunks c = final_chunks(a, dense);
    if (n_chunks) *n_chunks = c.n_chunks;
    if (chunk_cells) *chunk_cells = c.chunk_items * (size_t)c.vec;
    return 0;
}

// the fixed-order reduce over ALL chunk partials of a final sweep (stars: Lagrangian sweeps only)
extern "C" int c21hip_final_sweep_reduce(const c21hip_ionize_args *a, int dense, double *partials,
                                         double *sum_stars_out, double *sum_xh_out, void *stream) {
    const FinalChunks c = final_chunks(a, dense);
    double *ps = partials, *px = partials + kMaxBlocks;
    if (sum_stars_out)
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, ps,
                           c.n_chunks, 0, sum_stars_out);
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, px,
                       c.n_chunks, 0, sum_xh_out);
    LAUNCH_CHECK();
    return 0;
}

// chunks [chunk_begin, chunk_end) of the final sweep (chunk_end < 0: all); leaves their partials in
// partials[chunk] (stars) and partials[kMaxBlocks + chunk] (x_HI); no reduce
extern "C" int c21hip_final_sweep_range(const c21hip_ionize_args *a, double stored_redshift,
                                        const unsigned char *first_cross, const float *stars_fil,
                                        const float *density, const float *prev_z_reion, float *xH,
                                        float *z_reion, float *kinetic_temperature, double *partials,
                                        int *flag_out, int stars_direct, const float *xe_dense,
                                        const float *kinetic_temp_neutral, int chunk_begin, int chunk_end,
                                        void *stream) {
    if (a->use_ts_fluct && (!stars_direct || !xe_dense || (!a->minimize_memory && !kinetic_temp_neutral))) {
        c21hip_set_error("final sweep: the x_e path needs the dense inputs (stars_direct)");
        return C21CM_VALUE_ERROR;
    }
    const FinalChunks c = final_chunks(a, stars_direct);
    const int vec = c.vec;
    IoniseParams p = make_params(a, vec == 4 ? 2 : vec);
    if (stars_direct) p.nitems = c.nitems;  // dense grids: the box is one long row
    if (chunk_end < 0) chunk_begin = 0, chunk_end = c.n_chunks;
    if (chunk_begin < 0 || chunk_end > c.n_chunks || chunk_begin > chunk_end) {
        c21hip_set_error("final sweep: chunk range [%d, %d) outside [0, %d)", chunk_begin, chunk_end, c.n_chunks);
        return C21CM_VALUE_ERROR;
    }
    const int blocks = chunk_end - chunk_begin;
    if (blocks == 0) return 0;
    double *ps = partials, *px = partials + kMaxBlocks;
#define LAUNCH_FINAL(V, D)                                                                       \
    hipLaunchKernelGGL((final_sweep_kernel<V, D>), dim3(blocks), dim3(kBlock), 0,                \
                       (hipStream_t)stream, p, (float)stored_redshift, first_cross, stars_fil,   \
                       density, prev_z_reion, xe_dense, kinetic_temp_neutral, xH, z_reion,       \
                       kinetic_temperature, ps, px, flag_out, c.chunk_items, chunk_begin)
    if (stars_direct && vec == 4)
        LAUNCH_FINAL(4, true);
    else if (stars_direct)
        LAUNCH_FINAL(1, true);
    else if (vec == 2)
        LAUNCH_FINAL(2, false);
    else
        LAUNCH_FINAL(1, false);
#undef LAUNCH_FINAL
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_final_sweep(const c21hip_ionize_args *a, double stored_redshift,
                                  const unsigned char *first_cross, const float *stars_fil,
                                  const float *density, const float *prev_z_reion, float *xH,
                                  float *z_reion, float *kinetic_temperature, double *partials,
                                  double *sum_stars_out, double *sum_xh_out, int *flag_out,
                                  int stars_direct, const float *xe_dense,
                                  const float *kinetic_temp_neutral, void *stream) {
    const int st = c21hip_final_sweep_range(a, stored_redshift, first_cross, stars_fil, density, prev_z_reion,
                                            xH, z_reion, kinetic_temperature, partials, flag_out, stars_direct,
                                            xe_dense, kinetic_temp_neutral, 0, -1, stream);
    if (st) return st;
    return c21hip_final_sweep_reduce(a, stars_direct, partials, sum_stars_out, sum_xh_out, stream);
}

// The cell-scale radius and the post-loop of the Eulerian source models in one sweep
// (final_sweep_kernel<V, true, true>): first crossings of the larger radii (first_cross), the barrier and
// the partial ionisation at radius index 0 from its dense f_coll grid and box mean, z_reion, T_k, the sum
// of x_HI.  xe_dense / kinetic_temp_neutral: the inputs of a spin-temperature run (NULL otherwise).
// chunk_begin / chunk_end as c21hip_final_sweep_range (no reduce); c21hip_final_sweep_eulerian: all + reduce.
extern "C" int c21hip_final_sweep_eulerian_range(const c21hip_ionize_args *a, double stored_redshift,
                                                 const unsigned char *first_cross, const float *nion_dense,
                                                 const double *mean_dev, const float *density,
                                                 const float *prev_z_reion, float *xH, float *z_reion,
                                                 float *kinetic_temperature, double *partials, int *flag_out,
                                                 const float *xe_dense, const float *kinetic_temp_neutral,
                                                 int chunk_begin, int chunk_end, void *stream) {
    if (a->use_ts_fluct && (!xe_dense || (!a->minimize_memory && !kinetic_temp_neutral))) {
        c21hip_set_error("final sweep: the x_e path needs the dense inputs");
        return C21CM_VALUE_ERROR;
    }
    const FinalChunks c = final_chunks(a, 1);
    const int vec = c.vec;
    IoniseParams p = make_params(a, vec == 4 ? 2 : vec);
    p.nitems = c.nitems;  // dense grids: the box is one long row
    if (chunk_end < 0) chunk_begin = 0, chunk_end = c.n_chunks;
    if (chunk_begin < 0 || chunk_end > c.n_chunks || chunk_begin > chunk_end) {
        c21hip_set_error("final sweep: chunk range [%d, %d) outside [0, %d)", chunk_begin, chunk_end, c.n_chunks);
        return C21CM_VALUE_ERROR;
    }
    const int blocks = chunk_end - chunk_begin;
    if (blocks == 0) return 0;
    double *ps = partials, *px = partials + kMaxBlocks;
    if (vec == 4)
        hipLaunchKernelGGL((final_sweep_kernel<4, true, true>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, p, (float)stored_redshift, first_cross, nion_dense, density,
                           prev_z_reion, xe_dense, kinetic_temp_neutral, xH, z_reion, kinetic_temperature, ps,
                           px, flag_out, c.chunk_items, chunk_begin, mean_dev);
    else
        hipLaunchKernelGGL((final_sweep_kernel<1, true, true>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, p, (float)stored_redshift, first_cross, nion_dense, density,
                           prev_z_reion, xe_dense, kinetic_temp_neutral, xH, z_reion, kinetic_temperature, ps,
                           px, flag_out, c.chunk_items, chunk_begin, mean_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_final_sweep_eulerian(const c21hip_ionize_args *a, double stored_redshift,
                                           const unsigned char *first_cross, const float *nion_dense,
                                           const double *mean_dev, const float *density,
                                           const float *prev_z_reion, float *xH, float *z_reion,
                                           float *kinetic_temperature, double *partials, double *sum_xh_out,
                                           int *flag_out, const float *xe_dense,
                                           const float *kinetic_temp_neutral, void *stream) {
    const int st = c21hip_final_sweep_eulerian_range(a, stored_redshift, first_cross, nion_dense, mean_dev,
                                                     density, prev_z_reion, xH, z_reion, kinetic_temperature,
                                                     partials, flag_out, xe_dense, kinetic_temp_neutral, 0, -1,
                                                     stream);
    if (st) return st;
    return c21hip_final_sweep_reduce(a, 1, partials, nullptr, sum_xh_out, stream);
}

extern "C" int c21hip_brightness_temp(const float *density, const float *xH, const float *Ts,
                                      float *bt, float *tau, size_t n, float const_factor,
                                      float T_rad, double redshift, int use_ts, double *partials,
                                      double *sum_out, void *stream) {
    const int blocks = grid_for(n);
    hipLaunchKernelGGL(brightness_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                       density, xH, Ts, bt, tau, n, const_factor, T_rad, redshift, use_ts,
                       partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sum_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_apply_first_cross(const unsigned char *first_cross,
                                        const float *prev_z_reion, int first_snapshot,
                                        double redshift, float *xH, float *z_reion, size_t ntot,
                                        void *stream) {
    hipLaunchKernelGGL(apply_first_cross_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, first_cross, prev_z_reion, first_snapshot,
                       (float)redshift, xH, z_reion, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_apply_first_cross_recomb(const unsigned char *first_cross, const float *R_dev,
                                               const float *prev_z_reion, int first_snapshot,
                                               double redshift, float *xH, float *z_reion,
                                               float *mfp, size_t ntot, void *stream) {
    hipLaunchKernelGGL(apply_first_cross_recomb_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, first_cross, R_dev, prev_z_reion, first_snapshot,
                       (float)redshift, xH, z_reion, mfp, ntot);
    LAUNCH_CHECK();
    return 0;
}

// element-wise max of two device arrays (uint8 or uint64), dst = max(dst, src): the in-process
// stand-in for ncclReduce(max) of the sharding emulation hook (c21cm_shard_emulate)
template <typename T>
__global__ void __launch_bounds__(kBlock)
max_into_kernel(T *__restrict__ dst, const T *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        dst[i] = src[i] > dst[i] ? src[i] : dst[i];
}

extern "C" int c21hip_max_into(void *dst, const void *src, size_t count, int bytes_per_element,
                               void *stream) {
    if (bytes_per_element == 1)
        hipLaunchKernelGGL((max_into_kernel<unsigned char>), dim3(grid_for(count)), dim3(kBlock), 0,
                           (hipStream_t)stream, (unsigned char *)dst, (const unsigned char *)src,
                           count);
    else if (bytes_per_element == 8)
        hipLaunchKernelGGL((max_into_kernel<unsigned long long>), dim3(grid_for(count)),
                           dim3(kBlock), 0, (hipStream_t)stream, (unsigned long long *)dst,
                           (const unsigned long long *)src, count);
    else
        return C21CM_VALUE_ERROR;
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_pack_mask_bits(const unsigned char *fc, unsigned *bits, size_t ntot,
                                     void *stream) {
    const size_t nwords = (ntot + 31) / 32;
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(grid_for(nwords)), dim3(kBlock), 0,
                       (hipStream_t)stream, fc, bits, nwords, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_or_unpack_mask_bits(const unsigned *bits, size_t stride_words, int world,
                                          unsigned char *fc, size_t ntot, void *stream) {
    const size_t nwords = (ntot + 31) / 32;
    hipLaunchKernelGGL(or_unpack_mask_bits_kernel, dim3(grid_for(nwords)), dim3(kBlock), 0,
                       (hipStream_t)stream, bits, stride_words, world, fc, nwords, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_pack_cross_keys(const float *mfp, const float *G12, unsigned long long *keys,
                                      size_t ntot, void *stream) {
    hipLaunchKernelGGL(pack_cross_keys_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, mfp, G12, keys, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_apply_cross_keys(const unsigned long long *keys, const float *prev_z_reion,
                                       int first_snapshot, double redshift, float *xH,
                                       float *z_reion, float *G12, float *mfp, size_t ntot,
                                       void *stream) {
    hipLaunchKernelGGL(apply_cross_keys_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, keys, prev_z_reion, first_snapshot, (float)redshift, xH,
                       z_reion, G12, mfp, ntot);
    LAUNCH_CHECK();
    return 0;
}

// Sharded fused recombination loop: the winner of a cell is the rank with the larger
// first-crossing index (an index > 0 belongs to one rank), together with ITS Gamma_12.  The own
// slab (mask / g12, in place) against `n_peers` received slabs of `stride` cells each; four cells
// per thread (slab bounds are multiples of 4 cells).
__global__ void __launch_bounds__(kBlock)
combine_cross_g12_kernel(unsigned char *__restrict__ mask, float *__restrict__ g12,
                         const unsigned char *__restrict__ peer_mask, const float *__restrict__ peer_g12,
                         int n_peers, size_t stride, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        uchar4 m = reinterpret_cast<uchar4 *>(mask)[i];
        float4 g = reinterpret_cast<float4 *>(g12)[i];
        bool touched = false;
        for (int q = 0; q < n_peers; q++) {
            const uchar4 pm = reinterpret_cast<const uchar4 *>(peer_mask + q * stride)[i];
            if (pm.x > m.x || pm.y > m.y || pm.z > m.z || pm.w > m.w) {
                const float4 pg = reinterpret_cast<const float4 *>(peer_g12 + q * stride)[i];
                if (pm.x > m.x) { m.x = pm.x; g.x = pg.x; }
                if (pm.y > m.y) { m.y = pm.y; g.y = pg.y; }
                if (pm.z > m.z) { m.z = pm.z; g.z = pg.z; }
                if (pm.w > m.w) { m.w = pm.w; g.w = pg.w; }
                touched = true;
            }
        }
        if (touched) {
            reinterpret_cast<uchar4 *>(mask)[i] = m;
            reinterpret_cast<float4 *>(g12)[i] = g;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail of the last slab
        const size_t i = n4 * 4 + threadIdx.x;
        unsigned char m = mask[i];
        float g = g12[i];
        for (int q = 0; q < n_peers; q++)
            if (peer_mask[q * stride + i] > m) {
                m = peer_mask[q * stride + i];
                g = peer_g12[q * stride + i];
            }
        mask[i] = m;
        g12[i] = g;
    }
}

extern "C" int c21hip_combine_cross_g12(unsigned char *mask, float *g12, const unsigned char *peer_mask,
                                        const float *peer_g12, int n_peers, size_t stride, size_t n,
                                        void *stream) {
    if (n_peers < 1 || n == 0) return 0;
    if (((uintptr_t)mask & 3) || ((uintptr_t)g12 & 15) || ((uintptr_t)peer_mask & 3) ||
        ((uintptr_t)peer_g12 & 15) || (stride & 3)) {
        c21hip_set_error("combine_cross_g12: slabs must start at multiples of 4 cells");
        return C21CM_VALUE_ERROR;
    }
    hipLaunchKernelGGL(combine_cross_g12_kernel, dim3(grid_for(n / 4 + 1)), dim3(kBlock), 0,
                       (hipStream_t)stream, mask, g12, peer_mask, peer_g12, n_peers, stride, n);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_neutral_box(const float *density, const float *xe, const float *Tneutral,
                                  float *xH, float *Tk, size_t ntot, int ts, double global_xH,
                                  double TK, double adia, void *stream) {
    hipLaunchKernelGGL(neutral_box_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0,
                       (hipStream_t)stream, density, xe, Tneutral, xH, Tk, ntot, ts, global_xH, TK,
                       adia);
    LAUNCH_CHECK();
    return 0;
}

// *flag_host = 1 if any element of the device array is non-zero (synchronises)
extern "C" int c21hip_any_nonzero(const float *a, size_t n, int *flag_host, void *stream) {
    int *flag = (int *)c21hip_ws(47, sizeof(int));
    if (!flag) return C21CM_MEMORY_ALLOC_ERROR;
    int st = c21hip_memset(flag, 0, sizeof(int), stream);
    if (st) return st;
    hipLaunchKernelGGL(any_nonzero_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       a, n, flag);
    LAUNCH_CHECK();
    if ((st = c21hip_d2h(flag_host, flag, sizeof(int), stream))) return st;
    return c21hip_sync(stream);
}

// ======================================================================================
// USE_MINI_HALOS (E-INTEGRAL): turnover-mass boxes, two-population f_coll with the per-radius
// history, two-population barrier.  reference: IonisationBox.c:403-457, 838-936, 1068-1200.
// The branch is bound by its four filtered grids per radius, not by these sweeps; they are
// written for parity first (double exp, one cell per thread item, no LDS tables: the four
// 400 x 50 tables are 320 kB and stay in L2).
// ======================================================================================
namespace {
struct MturnParams {
    size_t ntot;
    int first_snapshot;
    float z;
    double mturn_a_nofb, mturn_m_nofb, vcb_const;
    double A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb;
};

// calculate_mcrit_boxes (:403-457) with thermochem.c:281-311 inlined
__global__ void __launch_bounds__(kBlock)
mturn_kernel(MturnParams m, const float *__restrict__ prev_G12,
             const float *__restrict__ prev_z_reion, const float *__restrict__ J_21_LW,
             const float *__restrict__ vcb, float *__restrict__ out_a, float *__restrict__ out_m,
             double *__restrict__ partials_a, double *__restrict__ partials_m,
             int *__restrict__ flag) {
    double acc_a = 0., acc_m = 0.;
    int bad = 0;
    const double zp1 = 1. + (double)m.z;
    const double mcrit_noLW = 3.314e7 * pow(zp1, -1.5);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m.ntot;
         i += (size_t)gridDim.x * kBlock) {
        const float z_IN = m.first_snapshot ? -1.f : prev_z_reion[i];
        double Mcrit_RE = 1e-40;
        if (!((double)z_IN <= 1e-19))
            Mcrit_RE = 3e9 * pow(2.0 * (double)prev_G12[i], 0.17) * pow(zp1 / 10, -2.1) *
                       pow(1 - pow(zp1 / (1. + (double)z_IN), 2.0), 2.5);
        const float v = vcb ? vcb[i] : (float)m.vcb_const;
        const double f_LW = 1.0 + m.A_LW * pow((double)J_21_LW[i], m.BETA_LW);
        const double f_vcb = pow(1.0 + m.A_VCB * (double)v / m.sigma_vcb, m.BETA_VCB);
        const double Mcrit_LW = mcrit_noLW * f_LW * f_vcb;
        if (Mcrit_LW != Mcrit_LW || Mcrit_LW == 0) bad = 1;
        const double curr_Mt = log10(fmax(Mcrit_RE, m.mturn_a_nofb));
        const double curr_Mt_MINI = log10(fmax(Mcrit_RE, fmax(Mcrit_LW, m.mturn_m_nofb)));
        out_a[i] = (float)curr_Mt;
        out_m[i] = (float)curr_Mt_MINI;
        acc_a += curr_Mt;
        acc_m += curr_Mt_MINI;
    }
    if (bad) atomicOr(flag, 1);
    block_sum_to(acc_a, partials_a);
    __syncthreads();
    block_sum_to(acc_m, partials_m);
}

// interpolation.c:133-157
__device__ __forceinline__ double eval_table2d_f(double x, double y, double x_min, double x_width,
                                                 double y_min, double y_width,
                                                 const float *__restrict__ z_arr) {
    const int x_idx = (int)floor((x - x_min) / x_width);
    const int y_idx = (int)floor((y - y_min) / y_width);
    const double x_table = x_min + x_width * (double)x_idx;
    const double y_table = y_min + y_width * (double)y_idx;
    const double px = (x - x_table) / x_width, py = (y - y_table) / y_width;
    const float *r0 = z_arr + (size_t)x_idx * C21CM_NMTURN_TABLE + y_idx;
    const float *r1 = r0 + C21CM_NMTURN_TABLE;
    const double left_edge = (double)r0[0] * (1 - py) + (double)r0[1] * py;
    const double right_edge = (double)r1[0] * (1 - py) + (double)r1[1] * py;
    return left_edge * (1 - px) + right_edge * px;
}

struct MiniFcollParams {
    size_t nitems;
    int nz_items, zpad_items;
    int need_prev;
    double tab_min, tab_width, ptab_min, ptab_width;
    double mta_min, mta_width, mtm_min, mtm_width;
};

__device__ __forceinline__ double clamp_fcoll(double f) {  // :904-907
    if (f > 1.) f = 1.;
    if (f < 0.) f = 1e-40;
    return f;
}

// calculate_fcoll_grid with need_minihalo_nion (:838-936): tables = acg | mcg | prev acg | prev mcg
template <int VEC>
__global__ void __launch_bounds__(kBlock)
fcoll_mini_kernel(MiniFcollParams p, const float *__restrict__ delta_fil,
                  const float *__restrict__ pdelta_fil, const float *__restrict__ mta_fil,
                  const float *__restrict__ mtm_fil, const float *__restrict__ tables,
                  const float *__restrict__ prev_nion, const float *__restrict__ prev_mini,
                  float *__restrict__ nion_out, float *__restrict__ mini_out,
                  double *__restrict__ partials_a, double *__restrict__ partials_m) {
    constexpr size_t t2 = (size_t)C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
    double acc_a = 0., acc_m = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < p.nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, p.nz_items, p.zpad_items);
        const auto dl = Pack<VEC>::load(delta_fil, ci.padded);
        const auto ma = Pack<VEC>::load(mta_fil, ci.padded);
        const auto mm = Pack<VEC>::load(mtm_fil, ci.padded);
        const auto hn = Pack<VEC>::load(prev_nion, ci.dense);
        const auto hm = Pack<VEC>::load(prev_mini, ci.dense);
        Pack<VEC> pd, oa, om;
        if (p.need_prev) pd = Pack<VEC>::load(pdelta_fil, ci.padded);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const double curr_dens = (double)clip_delta_eulerian(dl.v[e]);
            // clip_and_get_extrema(., 0, LOG10_MTURN_MAX): :722-725
            const double l10a = (double)fmaxf((float)fmin((double)ma.v[e], 10.), 0.f);
            const double l10m = (double)fmaxf((float)fmin((double)mm.v[e], 10.), 0.f);
            double f_m = exp(eval_table2d_f(curr_dens, l10m, p.tab_min, p.tab_width, p.mtm_min,
                                            p.mtm_width, tables + t2));
            double pf_a = 0., pf_m = 0.;
            if (p.need_prev) {
                const double prev_dens = (double)fmaxf((float)fmin((double)pd.v[e], 1e6), -1.f);
                pf_a = exp(eval_table2d_f(prev_dens, l10a, p.ptab_min, p.ptab_width, p.mta_min,
                                          p.mta_width, tables + 2 * t2));
                pf_m = exp(eval_table2d_f(prev_dens, l10m, p.ptab_min, p.ptab_width, p.mtm_min,
                                          p.mtm_width, tables + 3 * t2));
            }
            double f_a = exp(eval_table2d_f(curr_dens, l10a, p.tab_min, p.tab_width, p.mta_min,
                                            p.mta_width, tables));
            f_a = clamp_fcoll(f_a);
            pf_a = clamp_fcoll(pf_a);
            float va = (float)((double)hn.v[e] + f_a - pf_a);
            if ((double)va > 1.) va = 1.f;
            f_m = clamp_fcoll(f_m);
            pf_m = clamp_fcoll(pf_m);
            float vm = (float)((double)hm.v[e] + f_m - pf_m);
            if ((double)vm > 1.) vm = 1.f;
            oa.v[e] = va;
            om.v[e] = vm;
            acc_a += (double)va;
            acc_m += (double)vm;
        }
        oa.store(nion_out, ci.dense);
        om.store(mini_out, ci.dense);
    }
    block_sum_to(acc_a, partials_a);
    __syncthreads();
    block_sum_to(acc_m, partials_m);
}

struct MiniIoniseParams {
    IoniseParams ip;
    int lagrangian;  // source grid = filtered HaloBox.n_ion (mini-halos inside): only the floor
                     // f_limit_mcg of the second population enters (:1068-1082)
    int recomb, inhomo, cell_recomb, ts;
    double R, gamma_prefactor, gamma_prefactor_mini;
    double ion_eff_mini, f_limit_mcg, mean_f_coll_mini;
};

// find_ionised_regions with both populations (:1031-1200); recombinations optional
template <int VEC>
__global__ void __launch_bounds__(kBlock)
ionise_mini_kernel(MiniIoniseParams q, const float *__restrict__ delta_fil,
                   const float *__restrict__ nion_dense,  // Lagrangian: stars_fil (padded rows)
                   const float *__restrict__ mini_dense,  // Lagrangian: sfr_fil (padded) or NULL
                   const float *__restrict__ xe_fil, const float *__restrict__ nrec_fil,
                   const float *__restrict__ prev_nrec, const float *__restrict__ density,
                   const float *__restrict__ prev_z_reion, const float *__restrict__ Tneutral,
                   const double *__restrict__ mean_a_dev, const double *__restrict__ mean_m_dev,
                   float *__restrict__ xH, float *__restrict__ z_reion, float *__restrict__ Tk,
                   float *__restrict__ G12, float *__restrict__ mfp,
                   double *__restrict__ partials) {
    const c21hip_ionize_args &a = q.ip.a;
    const bool LAST = (a.r_index == 0);
    const bool LAG = q.lagrangian;
    const double fix_a = (!LAG && a.fix_mean) ? a.mean_f_coll / *mean_a_dev : 1.;
    const double fix_m = (!LAG && a.fix_mean) ? q.mean_f_coll_mini / *mean_m_dev : 1.;
    const float z_now = (float)a.redshift;
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < q.ip.nitems;
         i += (size_t)gridDim.x * kBlock) {
        const auto ci = cell_index<VEC>(i, q.ip.nz_items, q.ip.zpad_items);
        const auto fa = Pack<VEC>::load(nion_dense, LAG ? ci.padded : ci.dense);
        Pack<VEC> fm, sf;
        if (!LAG) fm = Pack<VEC>::load(mini_dense, ci.dense);
        if (LAG && q.recomb) sf = Pack<VEC>::load(mini_dense, ci.padded);
        Pack<VEC> dl, xe, de, nr;
        if (!LAST) dl = Pack<VEC>::load(delta_fil, ci.padded);
        if (LAST || !a.minimize_memory) de = Pack<VEC>::load(density, ci.dense);
        if (q.ts) xe = Pack<VEC>::load(xe_fil, ci.padded);
        if (q.recomb) {
            if (!q.cell_recomb)
                nr = Pack<VEC>::load(nrec_fil, ci.padded);
            else if (q.inhomo)
                nr = Pack<VEC>::load(prev_nrec, ci.dense);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t idx = ci.dense * VEC + e;
            const double curr_dens =
                LAST ? (double)de.v[e] * a.photoncons_factor
                     : (double)(LAG ? clip_delta(dl.v[e]) : clip_delta_eulerian(dl.v[e]));
            double curr_fcoll, curr_fcoll_mini;
            if (LAG) {
                const float stars = fmaxf(fa.v[e], 0.f);
                acc += (double)stars;
                curr_fcoll = (double)stars * (1 / (a.rhocrit_omb * (1 + curr_dens)));
                curr_fcoll_mini = 0.;
            } else {
                curr_fcoll = fix_a * (double)fa.v[e];
                curr_fcoll_mini = fix_m * (double)fm.v[e];
            }
            if (a.mass_dep_zeta) {
                if (curr_fcoll < a.f_limit) curr_fcoll = a.f_limit;
                if (curr_fcoll_mini < q.f_limit_mcg) curr_fcoll_mini = q.f_limit_mcg;
            }
            double rec = 0.;
            if (q.recomb) {
                if (!q.cell_recomb)
                    rec = (double)fmaxf(nr.v[e], 0.f);
                else
                    rec = q.inhomo ? (double)nr.v[e] : (double)prev_nrec[0];
                rec /= (1. + curr_dens);
            }
            const double x_e = q.ts ? (double)clip_xe(xe.v[e]) : 0.;
            if (curr_fcoll * a.ion_eff_factor + curr_fcoll_mini * q.ion_eff_mini >
                (1. - x_e) * (1.0 + rec)) {
                if (q.recomb && (double)xH[idx] > kFractFloatErr) {  // first crossing
                    if (LAG)
                        G12[idx] = (float)(q.R * q.gamma_prefactor / (1 + curr_dens) *
                                           (double)fmaxf(sf.v[e], 0.f));
                    else
                        G12[idx] = (float)(q.R * (q.gamma_prefactor * curr_fcoll +
                                                  q.gamma_prefactor_mini * curr_fcoll_mini));
                    if (mfp) mfp[idx] = (float)q.R;
                }
                const float pz = a.first_snapshot ? -1.f : prev_z_reion[idx];
                z_reion[idx] = (pz < 0.f) ? z_now : pz;
                xH[idx] = 0.f;
            } else if (LAST) {
                if ((double)xH[idx] > kTiny) {
                    double res_xH =
                        1. - curr_fcoll * a.ion_eff_factor - curr_fcoll_mini * q.ion_eff_mini;
                    if (!a.minimize_memory) {
                        const float T_HI =
                            q.ts ? Tneutral[idx]
                                 : (float)(a.TK_nofluct * (1 + a.adia_TK_term * (double)de.v[e]));
                        Tk[idx] = partially_ionized_T(T_HI, (float)res_xH, (float)a.T_re);
                    }
                    res_xH -= x_e;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    xH[idx] = (float)res_xH;
                }
            }
        }
    }
    if (LAG) block_sum_to(acc, partials);
}
}  // namespace

extern "C" int c21hip_mturn_grids(size_t ntot, int first_snapshot, double redshift,
                                  double mturn_a_nofb, double mturn_m_nofb, double vcb_const,
                                  double A_LW, double BETA_LW, double A_VCB, double BETA_VCB,
                                  double sigma_vcb, const float *prev_G12,
                                  const float *prev_z_reion, const float *J_21_LW,
                                  const float *vcb, float *out_a, float *out_m, double *partials,
                                  double *sums_out, int *flag_dev, void *stream) {
    MturnParams m;
    m.ntot = ntot;
    m.first_snapshot = first_snapshot;
    m.z = (float)redshift;  // consts->redshift reaches both thresholds as a float argument
    m.mturn_a_nofb = mturn_a_nofb;
    m.mturn_m_nofb = mturn_m_nofb;
    m.vcb_const = vcb_const;
    m.A_LW = A_LW;
    m.BETA_LW = BETA_LW;
    m.A_VCB = A_VCB;
    m.BETA_VCB = BETA_VCB;
    m.sigma_vcb = sigma_vcb;
    const int blocks = grid_for(ntot);
    hipLaunchKernelGGL(mturn_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, m,
                       prev_G12, prev_z_reion, J_21_LW, vcb, out_a, out_m, partials,
                       partials + kMaxBlocks, flag_dev);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sums_out);
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials + kMaxBlocks, blocks, 0, sums_out + 1);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_fcoll_mini(int nx, int ny, int nz, int need_prev, const double *ranges,
                                 const float *delta_fil, const float *pdelta_fil,
                                 const float *mta_fil, const float *mtm_fil,
                                 const float *tables_dev, const float *prev_nion,
                                 const float *prev_mini, float *nion_out, float *mini_out,
                                 double *partials, double *sums_out, void *stream) {
    const int vec = (nz % 2 == 0) ? 2 : 1;
    const int zpad = 2 * (nz / 2 + 1);
    MiniFcollParams p;
    p.nz_items = nz / vec;
    p.zpad_items = zpad / vec;
    p.nitems = (size_t)nx * ny * p.nz_items;
    p.need_prev = need_prev;
    p.tab_min = ranges[0], p.tab_width = ranges[1];
    p.ptab_min = ranges[2], p.ptab_width = ranges[3];
    p.mta_min = ranges[4], p.mta_width = ranges[5];
    p.mtm_min = ranges[6], p.mtm_width = ranges[7];
    const int blocks = grid_for(p.nitems);
    if (vec == 2)
        hipLaunchKernelGGL(fcoll_mini_kernel<2>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           p, delta_fil, pdelta_fil, mta_fil, mtm_fil, tables_dev, prev_nion,
                           prev_mini, nion_out, mini_out, partials, partials + kMaxBlocks);
    else
        hipLaunchKernelGGL(fcoll_mini_kernel<1>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream,
                           p, delta_fil, pdelta_fil, mta_fil, mtm_fil, tables_dev, prev_nion,
                           prev_mini, nion_out, mini_out, partials, partials + kMaxBlocks);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, blocks, 0, sums_out);
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                       partials + kMaxBlocks, blocks, 0, sums_out + 1);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ionise_mini(const c21hip_ionize_args *a, int lagrangian, int recomb,
                                  int inhomo, int cell_recomb, double R, double gamma_prefactor,
                                  double gamma_prefactor_mini, double ion_eff_mini,
                                  double f_limit_mcg, double mean_f_coll_mini,
                                  const float *delta_fil, const float *nion_dense,
                                  const float *mini_dense, const float *xe_fil,
                                  const float *nrec_fil, const float *prev_nrec,
                                  const float *density, const float *prev_z_reion,
                                  const float *kinetic_temp_neutral, const double *mean_a_dev,
                                  const double *mean_m_dev, float *xH, float *z_reion,
                                  float *kinetic_temperature, float *G12, float *mfp,
                                  double *partials, double *sum_out, void *stream) {
    const int vec = (a->nz % 2 == 0) ? 2 : 1;
    MiniIoniseParams q;
    q.ip = make_params(a, vec);
    q.lagrangian = lagrangian;
    q.recomb = recomb;
    q.inhomo = inhomo;
    q.cell_recomb = cell_recomb;
    q.ts = a->use_ts_fluct;
    q.R = R;
    q.gamma_prefactor = gamma_prefactor;
    q.gamma_prefactor_mini = gamma_prefactor_mini;
    q.ion_eff_mini = ion_eff_mini;
    q.f_limit_mcg = f_limit_mcg;
    q.mean_f_coll_mini = mean_f_coll_mini;
    const int blocks = grid_for(q.ip.nitems);
    if (vec == 2)
        hipLaunchKernelGGL((ionise_mini_kernel<2>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, q, delta_fil, nion_dense, mini_dense, xe_fil,
                           nrec_fil, prev_nrec, density, prev_z_reion, kinetic_temp_neutral,
                           mean_a_dev, mean_m_dev, xH, z_reion, kinetic_temperature, G12, mfp,
                           partials);
    else
        hipLaunchKernelGGL((ionise_mini_kernel<1>), dim3(blocks), dim3(kBlock), 0,
                           (hipStream_t)stream, q, delta_fil, nion_dense, mini_dense, xe_fil,
                           nrec_fil, prev_nrec, density, prev_z_reion, kinetic_temp_neutral,
                           mean_a_dev, mean_m_dev, xH, z_reion, kinetic_temperature, G12, mfp,
                           partials);
    LAUNCH_CHECK();
    if (lagrangian && sum_out) {  // sum of the filtered source grid (f_coll_grid_mean, :946-961)
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream,
                           partials, blocks, 0, sum_out);
        LAUNCH_CHECK();
    }
    return 0;
}

// ======================================================================================
// IONISE_ENTIRE_SPHERE (IonisationBox.c:1150-1158, bubble_helper_progs.c:262-418): a cell that
// crosses the barrier at radius R flags every cell closer than R as ionised,
//   x_HI(x) = 0  for  |x - c|^2 (nearest periodic image, in cells) < Rsq(R)   [strict, floats]
// (update_in_sphere's inner cube lies inside that sphere).  A cell's larger radii contain its
// smaller ones, so the first-crossing mask (largest radius index per cell) describes the whole
// union.  One wavefront per crossing cell: lanes take the (dx, dy) columns of the bounding
// square and store the run of z cells inside the sphere.  Stores of 0.f need no atomics.
// ======================================================================================
namespace {
__global__ void __launch_bounds__(kBlock)
paint_spheres_kernel(const unsigned char *__restrict__ first_cross,
                     const float *__restrict__ rsq_of_index,  // [n_radii] Rsq in cells^2
                     float *__restrict__ xH, int nx, int ny, int nz) {
    const size_t ntot = (size_t)nx * ny * nz;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    for (size_t base = wave * 64; base < ntot; base += n_waves * 64) {
        const size_t mine = base + lane;
        const int r = mine < ntot ? (int)first_cross[mine] : 0;
        unsigned long long todo = __ballot(r > 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const size_t c = base + src;
            const int rc = __shfl(r, src, 64);
            const float Rsq = rsq_of_index[rc];
            const int cz = (int)(c % nz), cy = (int)((c / nz) % ny), cx = (int)(c / ((size_t)nz * ny));
            const int Ri = (int)ceilf(sqrtf(Rsq));  // bounding half-width (>= ceil(R dim))
            const int side = 2 * Ri + 1;
            for (int t = lane; t < side * side; t += 64) {
                const int dx = t / side - Ri, dy = t % side - Ri;
                const float rem = Rsq - (float)(dx * dx + dy * dy);  // exact: small integers
                if (!(rem > 0.f)) continue;                          // Rsq > d^2 needs dz^2 < rem
                int dzmax = (int)sqrtf(rem);
                while ((float)(dzmax * dzmax) >= rem) dzmax--;        // strict inequality
                while ((float)((dzmax + 1) * (dzmax + 1)) < rem) dzmax++;
                int x = (cx + dx) % nx, y = (cy + dy) % ny;
                if (x < 0) x += nx;
                if (y < 0) y += ny;
                float *line = xH + ((size_t)x * ny + y) * nz;
                if (2 * dzmax + 1 >= nz) {
                    for (int z = 0; z < nz; z++) line[z] = 0.f;
                } else {
                    for (int dz = -dzmax; dz <= dzmax; dz++) {
                        int z = (cz + dz) % nz;
                        if (z < 0) z += nz;
                        line[z] = 0.f;
                    }
                }
            }
        }
    }
}
}  // namespace

extern "C" int c21hip_paint_spheres(const unsigned char *first_cross, const float *rsq_dev,
                                    float *xH, int nx, int ny, int nz, void *stream) {
    const size_t ntot = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(paint_spheres_kernel, dim3(grid_for((ntot + 63) / 64 * 64)), dim3(kBlock), 0,
                       (hipStream_t)stream, first_cross, rsq_dev, xH, nx, ny, nz);
    LAUNCH_CHECK();
    return 0;
}
