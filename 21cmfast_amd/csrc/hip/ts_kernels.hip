// ts_kernels.hip -- the per-cell part of ComputeTsBox on the MI355X.
//
// reference: src/py21cmfast/src/SpinTemperatureBox.c
//   :892-927    init_first_Ts                         -> ts_first_kernel
//   :1010-1086  calculate_sfrd_from_grid (E-INTEGRAL)  -> sfrd_sum_kernel (+ the table lookup
//                                                         repeated inside ts_cell_kernel)
//   :1499-1522  x_e index / weight of a cell           \  ts_accumulate_kernel: one sweep over
//   :1541-1784  the R loop (largest shell first)        /  the shells' grids, six sums per cell
//   :1794-1848  prefactors + get_Ts_fast (:1210-1383)  -> ts_cell_kernel
//
// The reference walks the 40 shells in an outer loop and keeps six double boxes of partial
// sums between them (6 x 8 B x N read + written 40 times).  Here a cell is a thread: it walks
// its own 40 shell values (coalesced across the wavefront: grids are [R][N]), keeps the sums in
// registers and writes them once; a second, short sweep runs the temperature update.  Bytes:
// 8 B (GRIDS) or 4 B (SFRD / f_coll tables) per cell and shell, read once (twice with tables).
// The table modes are NOT HBM-bound: the loop costs ~140 fp64-heavy instruction slots per cell
// and shell (table lookup, exp, three interpolated frequency integrals), 19 ms at 512^3 where
// the bytes would take 5.  Per-shell scalars and the 3 x 14 x n_step frequency-integral tables
// sit in LDS (16 KB for 40 shells).  The table modes need the box mean of the table values of
// every shell before the sums (avg_fix_term), hence one extra sweep over the filtered densities
// (sfrd_sum_kernel, blockIdx.y = shell).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "c21cm_grid.h"
#include "c21cm_kappa_tables.h"
#include "c21hip.h"
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

// C21CM_TS_LOOP=v1: the round-2 shell loop (fp64 lookups) instead of the folded fp32 one
inline bool ts_loop_v1() {
    const char *e = getenv("C21CM_TS_LOOP");
    return e && e[0] == 'v' && e[1] == '1';
}

__constant__ double kKappaHH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_HH_VALUES;
__constant__ double kKappaPH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_PH_VALUES;
__constant__ double kKappaEH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_EH_VALUES;
__constant__ float kXHII[C21CM_X_INT_NXHII] = C21CM_X_INT_XHII;

// heating_helper_progs.c:366-643: ln kappa on a regular ln T grid, linear between knots
__device__ inline double kappa_interior(const double *y, double width, double lnT) {
    int idx = (int)floor(lnT * (1. / width));
    idx = min(idx, C21CM_KAPPA_NPTS - 2);
    return y[idx] + (lnT - width * (double)idx) * (y[idx + 1] - y[idx]) * (1. / width);
}

__device__ inline double kappa_10(double lnT) {
    double ans;
    if (lnT < 0.)
        ans = kKappaHH[0];
    else if (lnT > C21CM_KAPPA_HH_LNT_MAX)  // power law T^0.381 above the table
        ans = log(exp(kKappaHH[C21CM_KAPPA_NPTS - 1]) *
                  pow(exp(lnT) / exp(C21CM_KAPPA_HH_LNT_MAX), 0.381));
    else
        ans = kappa_interior(kKappaHH, C21CM_KAPPA_HH_BINWIDTH, lnT);
    return exp(ans);
}

__device__ inline double kappa_linear_tail(const double *y, double width, double lnT_max,
                                           double lnT) {
    double ans;
    if (lnT < 0.)
        ans = y[0];
    else if (lnT > lnT_max)
        ans = y[C21CM_KAPPA_NPTS - 1] + (y[C21CM_KAPPA_NPTS - 1] - y[C21CM_KAPPA_NPTS - 2]) /
                                            (lnT_max - width * (C21CM_KAPPA_NPTS - 2)) *
                                            (lnT - lnT_max);
    else
        ans = kappa_interior(y, width, lnT);
    return exp(ans);
}

// thermochem.c:66-75 (Abel et al. 1997), Horner form of the same polynomial in ln(T / 1 eV)
__device__ inline double alpha_A(double T) {
    const double x = log(T / 1.1604505e4);
    const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x, x6 = x3 * x3, x7 = x6 * x,
                 x8 = x4 * x4, x9 = x8 * x;
    return exp(-28.6130338 - 0.72411256 * x - 2.02604473e-2 * x2 - 2.38086188e-3 * x3 -
               3.21260521e-4 * x4 - 1.42150291e-5 * x5 + 4.98910892e-6 * x6 +
               5.75561414e-7 * x7 - 1.85676704e-8 * x8 - 3.07113524e-9 * x9);
}

// heating_helper_progs.c:1210-1313
__device__ inline int nearest_point(double lo, double hi, int n, double value) {
    const double dn = (hi - lo) / (n - 1);
    if (value <= (lo + dn)) return 0;
    if (value >= hi) return n - 2;
    return (int)floor((value - lo) / dn);
}

__device__ double lya_heating_efficiency(double tk, double ts, double taugp,
                                         const double *__restrict__ arrE) {
    const double T_min = -1., T_max = 3., g_min = 1., g_max = 7.;
    const int nT = C21CM_LYA_NT, ngp = C21CM_LYA_NGP;
    tk = fmin(fmax(log10(tk), T_min), T_max);
    ts = fmin(fmax(log10(ts), T_min), T_max);
    taugp = fmin(fmax(log10(taugp), g_min), g_max);
    const int itk = nearest_point(T_min, T_max, nT, tk), its = nearest_point(T_min, T_max, nT, ts),
              igp = nearest_point(g_min, g_max, ngp, taugp);
    const double x0 = T_min + itk * (T_max - T_min) / (nT - 1),
                 x1 = T_min + (itk + 1) * (T_max - T_min) / (nT - 1);
    const double y0 = T_min + its * (T_max - T_min) / (nT - 1),
                 y1 = T_min + (its + 1) * (T_max - T_min) / (nT - 1);
    const double z0 = g_min + igp * (g_max - g_min) / (ngp - 1),
                 z1 = g_min + (igp + 1) * (g_max - g_min) / (ngp - 1);
    const double xd = (tk - x0) / (x1 - x0), yd = (ts - y0) / (y1 - y0),
                 zd = (taugp - z0) / (z1 - z0);
    auto at = [&](int a, int b, int c) { return arrE[((size_t)a * nT + b) * ngp + c]; };
    const double c00 = at(itk, its, igp) * (1. - xd) + at(itk + 1, its, igp) * xd;
    const double c01 = at(itk, its, igp + 1) * (1. - xd) + at(itk + 1, its, igp + 1) * xd;
    const double c10 = at(itk, its + 1, igp) * (1. - xd) + at(itk + 1, its + 1, igp) * xd;
    const double c11 = at(itk, its + 1, igp + 1) * (1. - xd) + at(itk + 1, its + 1, igp + 1) * xd;
    const double c0 = c00 * (1. - yd) + c10 * yd, c1 = c01 * (1. - yd) + c11 * yd;
    return c0 * (1. - zd) + c1 * zd;
}

// interpolation.c:123-131 with the two divisions by x_width replaced by its reciprocal (one
// rounding apart; the sweep is fp64-ALU bound on these lookups, not HBM bound)
__device__ inline double table_1d(double x, double x_min, double x_width, double inv_width,
                                  const float *__restrict__ y) {
    const int idx = (int)floor((x - x_min) * inv_width);
    const double table_val = x_min + x_width * (float)idx;
    const double interp_point = (x - table_val) * inv_width;
    return y[idx] * (1 - interp_point) + y[idx + 1] * interp_point;
}

__device__ inline double block_sum(double v, double *lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    const double r = lds[0];
    __syncthreads();
    return r;
}

// shell layout of the device table buffer (doubles), n = n_step:
//   [0..9n)  z_edge_factor, xray_R_factor, starlya, lya_cont, lya_inj, zpp_growth, tab_min,
//            tab_width, avg_fix_term (written by sfrd_finish_kernel; 1 for GRIDS)
//   [9n..)   freq_int_heat[14][n], freq_int_ion[14][n], freq_int_lya[14][n]
enum { SH_ZEDGE = 0, SH_XRAY_R, SH_STARLYA, SH_CONT, SH_INJ, SH_GROWTH, SH_TABMIN, SH_TABWIDTH,
       SH_AVGFIX, SH_TABINVW, SH_COUNT };
static_assert(SH_COUNT == C21HIP_TS_SHELL_ROWS, "shell rows of the device table buffer");

// box sum of the SFRD table values of one shell (blockIdx.y)
__global__ void __launch_bounds__(kBlock)
sfrd_sum_kernel(const float *__restrict__ filtered_density, const float *__restrict__ tables,
                int table_exp, const double *__restrict__ shell, int n_step, size_t ntot,
                double *__restrict__ partials) {
    __shared__ double lds[kBlock];
    const int R = blockIdx.y;
    const float *dens = filtered_density + (size_t)R * ntot;
    const float *tab = tables + (size_t)R * C21CM_NDELTA_TABLE;
    const double growth = shell[SH_GROWTH * n_step + R], tab_min = shell[SH_TABMIN * n_step + R],
                 tab_width = shell[SH_TABWIDTH * n_step + R], inv_w = shell[SH_TABINVW * n_step + R];
    double acc = 0.;
    if ((ntot & 3) == 0) {  // four cells per 16-byte load
        const float4 *d4 = reinterpret_cast<const float4 *>(dens);
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot / 4;
             i += (size_t)gridDim.x * kBlock) {
            const float4 t = d4[i];
            const float c[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double v = table_1d((double)c[e] * growth, tab_min, tab_width, inv_w, tab);
                acc += table_exp ? exp_f32acc(v) : v;
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
             i += (size_t)gridDim.x * kBlock) {
            const double v = table_1d((double)dens[i] * growth, tab_min, tab_width, inv_w, tab);
            // ln SFRD table (E-INTEGRAL; the value ends up in a float grid upstream: exp to float
            // accuracy, fcoll_device.h) | f_coll table (CONST-ION-EFF)
            acc += table_exp ? exp_f32acc(v) : v;
        }
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[(size_t)R * gridDim.x + blockIdx.x] = acc;
}

// avg_fix_term = mean_sfr_zpp / (sum / N) per shell (:1624); ave_out keeps the means
__global__ void __launch_bounds__(kBlock)
sfrd_finish_kernel(const double *__restrict__ partials, int nblocks,
                   const double *__restrict__ mean_sfr_zpp, double ntot, int n_step,
                   double *__restrict__ shell, double *__restrict__ ave_out) {
    __shared__ double lds[kBlock];
    const int R = blockIdx.x;
    double acc = 0.;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) acc += partials[(size_t)R * nblocks + i];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) {
        const double ave = acc / ntot;
        ave_out[R] = ave;
        shell[SH_AVGFIX * n_step + R] = mean_sfr_zpp[R] / ave;
    }
}

// One cell's epilogue: prefactors (:1794-1848) and get_Ts_fast (:1210-1383).
struct CellSums {
    double heat, ion, lya, starlya, cont, inj;
};
struct CellOut {
    double Ts, Tk, xe, J_alpha, xheat, xion;
};

__device__ CellOut ts_cell_epilogue(const c21hip_ts_args &a, const CellSums &q, float dens,
                                    float pTs_f, float pTk_f, float pXe_f,
                                    const double *__restrict__ lya_dEC,
                                    const double *__restrict__ lya_dEI) {
    double delta = (double)dens * a.growth_ratio;
    if (delta <= -1) delta = -1 + kFractFloatErr;
    const double dxheat_dt = a.use_xray_heating ? q.heat * a.xray_prefactor * a.volunit_inv : 0.;
    const double dxion_dt = q.ion * a.xray_prefactor * a.volunit_inv;
    const double dxlya_dt = q.lya * a.xray_prefactor * a.volunit_inv * a.Nb_zp * (1 + delta);
    const double dstarlya_dt = q.starlya * a.lya_star_prefactor * a.volunit_inv;
    const double pTs = pTs_f, pTk = pTk_f, pXe = pXe_f;

    const double zp = a.redshift, dzp = a.dzp;
    const double tau21 = (3 * a.h_p * a.A10 * a.c_cms * a.lambda_21 * a.lambda_21 / 32. / M_PI /
                          a.k_B) *
                         ((1 - pXe) * a.N_zp) / pTs / a.hubble_zp;
    double xCMB;
    if (tau21 > 1e-8)
        xCMB = (1. - exp(-tau21)) / tau21;
    else
        xCMB = 1. - tau21 / 2 * (1 - tau21 / 3 * (1 - tau21 / 4));
    const double dxion_sink_dt =
        alpha_A(pTk) * a.clumping_factor * pXe * pXe * a.h_frac * a.Nb_zp * (1. + delta);
    const double dxe_dzp = a.dt_dzp * (dxion_dt - dxion_sink_dt);
    double dadia_dzp = 3 / (1.0 + zp);
    if (fabs(delta) > kFractFloatErr)
        dadia_dzp += a.dgrowth_dzp / (a.growth_zp * (1.0 / delta + 1.0));
    dadia_dzp *= (2.0 / 3.0) * pTk;
    const double dspec_dzp = -dxe_dzp * pTk / (1 + pXe);
    const double dcomp_dzp =
        a.dcomp_dzp_prefactor * (pXe / (1.0 + pXe + a.he_frac)) * (a.Trad - pTk);
    double dxheat_dzp = 0.;
    if (a.use_xray_heating) dxheat_dzp = dxheat_dt * a.dt_dzp * 2.0 / 3.0 / a.k_B / (1.0 + pXe);
    double dCMBheat_dzp = 0.;
    if (a.use_cmb_heating) {
        const double eps_CMB = (3. / 4.) * (a.Trad / a.T_21) * a.A10 * a.h_frac *
                               (a.h_p * a.h_p / a.lambda_21 / a.lambda_21 / a.m_p) *
                               (1. + 2. * pTk / a.T_21);
        dCMBheat_dzp = -eps_CMB * (2. / 3. / a.k_B / (1. + pXe)) / a.hubble_zp / (1. + zp);
    }
    double eps_Lya_cont = 0., eps_Lya_inj = 0.;
    if (a.use_lya_heating) {
        const double tgp = 1.342881e-7 / a.hubble_zp * a.No * pow(1 + zp, 3) * (1.0 + delta) *
                           (1.0 - pXe);
        double E_continuum = lya_heating_efficiency(pTk, pTs, tgp, lya_dEC);
        double E_injected = lya_heating_efficiency(pTk, pTs, tgp, lya_dEI);
        if (isnan(E_continuum) || isinf(E_continuum)) E_continuum = 0.;
        if (isnan(E_injected) || isinf(E_injected)) E_injected = 0.;
        const double cont_dt = q.cont * a.lya_star_prefactor * a.volunit_inv;
        const double inj_dt = q.inj * a.lya_star_prefactor * a.volunit_inv;
        const double Ndot_alpha_cont = (4. * M_PI * a.nu_Ly_alpha) / (a.Nb_zp * (1. + delta)) /
                                       (1. + zp) / a.c_cms * cont_dt;
        const double Ndot_alpha_inj = (4. * M_PI * a.nu_Ly_alpha) / (a.Nb_zp * (1. + delta)) /
                                      (1. + zp) / a.c_cms * inj_dt;
        eps_Lya_cont = -Ndot_alpha_cont * E_continuum * (2. / 3. / a.k_B / (1. + pXe));
        eps_Lya_inj = -Ndot_alpha_inj * E_injected * (2. / 3. / a.k_B / (1. + pXe));
    }
    double x_e = pXe + (dxe_dzp * dzp);
    if (x_e > 1)
        x_e = 1 - kFractFloatErr;
    else if (x_e < 0)
        x_e = 0;
    double Tk = pTk;
    if (Tk < (double)(float)C21CM_TS_MAX_TK)
        Tk += (dxheat_dzp + dcomp_dzp + dspec_dzp + dadia_dzp + dCMBheat_dzp + eps_Lya_cont +
               eps_Lya_inj) *
              dzp;
    if (Tk < 0) Tk = a.Trad;

    const double J_alpha_tot = dstarlya_dt + dxlya_dt;
    const double T_inv = 1 / Tk, T_inv_sq = T_inv * T_inv;
    const double lnTk = log(Tk);
    const double xc_fast =
        (1.0 + delta) * a.xc_inverse *
        ((1.0 - x_e) * a.No * kappa_10(lnTk) +
         x_e * a.N_b0 *
             kappa_linear_tail(kKappaEH, C21CM_KAPPA_EH_BINWIDTH, C21CM_KAPPA_EH_LNT_MAX, lnTk) +
         x_e * a.No *
             kappa_linear_tail(kKappaPH, C21CM_KAPPA_PH_BINWIDTH, C21CM_KAPPA_PH_LNT_MAX, lnTk));
    const double xi_power = a.Ts_prefactor * cbrt((1.0 + delta) * (1.0 - x_e) * T_inv_sq);
    const double xa_arg = a.xa_tilde_prefactor * J_alpha_tot /
                          (1.0 + 2.98394 * xi_power + 1.53583 * xi_power * xi_power +
                           3.85289 * xi_power * xi_power * xi_power);
    const double Trad_inv = 1.0 / a.Trad;
    double TS;
    if (J_alpha_tot > 1.0e-20) {
        double TSold = 0.0;
        TS = a.Trad;
        int guard = 0;  // the fixed point converges in a handful of steps; NaNs end the loop
        while (fabs(TS - TSold) / TS > 1.0e-3 && guard++ < 10000) {
            TSold = TS;
            const double TS_inv = 1. / TS;
            const double xa = (1.0 - 0.0631789 * T_inv + 0.115995 * T_inv_sq -
                               0.401403 * T_inv * TS_inv + 0.336463 * T_inv_sq * TS_inv) *
                              xa_arg;
            TS = (xCMB + xa + xc_fast) /
                 (xCMB * Trad_inv + xa * (T_inv + 0.405535 * T_inv * TS_inv - 0.405535 * T_inv_sq) +
                  xc_fast * T_inv);
        }
    } else {
        TS = (xCMB + xc_fast) / (xCMB * Trad_inv + xc_fast * T_inv);
    }
    CellOut o;
    o.Ts = fabs(TS);
    o.Tk = Tk;
    o.xe = x_e;
    o.J_alpha = dxlya_dt + dstarlya_dt;
    o.xheat = dxheat_dt;
    o.xion = dxion_dt;
    return o;
}

template <int VEC>
struct FloatVec;
template <>
struct FloatVec<1> {
    float v[1];
    __device__ static FloatVec load(const float *p, size_t item) {
        FloatVec r;
        r.v[0] = p[item];
        return r;
    }
};
template <>
struct FloatVec<2> {
    float v[2];
    __device__ static FloatVec load(const float *p, size_t item) {
        const float2 t = reinterpret_cast<const float2 *>(p)[item];
        FloatVec r;
        r.v[0] = t.x, r.v[1] = t.y;
        return r;
    }
};
// Sweep 1 of 2 -- the R loop.  VEC cells per thread; the next shell's values are requested before
// the current shell's arithmetic (40 dependent loads per cell around fp64 code are latency-bound
// otherwise: 53 ms at 512^3 before, 19 ms now).  The six sums of a cell go to `sums` ([6][ntot]
// doubles): keeping the temperature update in the same kernel costs 250 VGPRs (2 waves per SIMD)
// for no gain, the round trip moves 48 B per cell.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
ts_accumulate_kernel(c21hip_ts_args a, const float *__restrict__ prev_xe,
                     const float *__restrict__ grid_a,  // sfr | delNL0
                     const float *__restrict__ grid_b,  // xray
                     const float *__restrict__ tables, const double *__restrict__ dev_tab,
                     double *__restrict__ sums, size_t ntot) {
    extern __shared__ double sh[];  // (SH_COUNT + 3 * NXHII) * n_step doubles
    const int n = a.n_step;
    const int n_tab = (SH_COUNT + 3 * C21CM_X_INT_NXHII) * n;
    for (int i = threadIdx.x; i < n_tab; i += kBlock) sh[i] = dev_tab[i];
    __syncthreads();
    const double *fheat = sh + SH_COUNT * n, *fion = fheat + C21CM_X_INT_NXHII * n,
                 *flya = fion + C21CM_X_INT_NXHII * n;
    const size_t nitems = ntot / VEC;  // ntot % VEC == 0 (launcher)
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < nitems;
         it += (size_t)gridDim.x * kBlock) {
        const auto pxe = FloatVec<VEC>::load(prev_xe, it);
        int m[VEC];
        double ival[VEC];
        CellSums q[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            // :1499-1514, float arithmetic as upstream
            float xHII_call = pxe.v[e];
            if (xHII_call > kXHII[C21CM_X_INT_NXHII - 1] * 0.999)
                xHII_call = (float)(kXHII[C21CM_X_INT_NXHII - 1] * 0.999);
            else if (xHII_call < kXHII[0])
                xHII_call = (float)(1.001 * kXHII[0]);
            int mm = C21CM_X_INT_NXHII - 1;
            while (xHII_call < kXHII[mm]) mm--;
            const float inv_diff = (float)(1. / (kXHII[mm + 1] - kXHII[mm]));
            m[e] = mm;
            ival[e] = (double)((xHII_call - kXHII[mm]) * inv_diff);
            q[e] = CellSums{0., 0., 0., 0., 0., 0.};
        }
        FloatVec<VEC> ga = FloatVec<VEC>::load(grid_a, (size_t)(n - 1) * nitems + it), gb = ga;
        if (a.lagrangian) gb = FloatVec<VEC>::load(grid_b, (size_t)(n - 1) * nitems + it);
        for (int R = n; R--;) {
            const FloatVec<VEC> ca = ga, cb = gb;
            if (R > 0) {  // request the next (smaller) shell now
                ga = FloatVec<VEC>::load(grid_a, (size_t)(R - 1) * nitems + it);
                if (a.lagrangian) gb = FloatVec<VEC>::load(grid_b, (size_t)(R - 1) * nitems + it);
            }
            const double z_edge = sh[SH_ZEDGE * n + R], xray_R = sh[SH_XRAY_R * n + R];
            const double starlya = sh[SH_STARLYA * n + R];
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                double sfr_term, xray_sfr;
                if (a.lagrangian) {
                    sfr_term = (double)ca.v[e] * z_edge;
                    xray_sfr = (double)cb.v[e] * z_edge * xray_R * 1e38;
                } else {
                    const double curr_dens = (double)ca.v[e] * sh[SH_GROWTH * n + R];
                    double fcoll = table_1d(curr_dens, sh[SH_TABMIN * n + R], sh[SH_TABWIDTH * n + R],
                                            sh[SH_TABINVW * n + R],
                                            tables + (size_t)R * C21CM_NDELTA_TABLE);
                    if (a.table_exp) fcoll = exp_f32acc(fcoll);  // else: the dfcoll/dz table itself
                    const float sfrd = (float)((1. + curr_dens) * fcoll);  // del_fcoll_Rct is float
                    sfr_term = (double)sfrd * z_edge * sh[SH_AVGFIX * n + R] * a.sfr_scale;
                    xray_sfr = sfr_term * a.xray_scale * xray_R;
                }
                const int lo = m[e] * n + R, hi = lo + n;
                if (a.use_xray_heating)
                    q[e].heat += xray_sfr * ((fheat[hi] - fheat[lo]) * ival[e] + fheat[lo]);
                q[e].ion += xray_sfr * ((fion[hi] - fion[lo]) * ival[e] + fion[lo]);
                q[e].lya += xray_sfr * ((flya[hi] - flya[lo]) * ival[e] + flya[lo]);
                q[e].starlya += sfr_term * starlya;
                if (a.use_lya_heating) {
                    q[e].cont += sfr_term * sh[SH_CONT * n + R];
                    q[e].inj += sfr_term * sh[SH_INJ * n + R];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t i = it * VEC + e;
            sums[i] = q[e].heat;
            sums[ntot + i] = q[e].ion;
            sums[2 * ntot + i] = q[e].lya;
            sums[3 * ntot + i] = q[e].starlya;
            if (a.use_lya_heating) {
                sums[4 * ntot + i] = q[e].cont;
                sums[5 * ntot + i] = q[e].inj;
            }
        }
    }
}

// ------------------------------------------------------------------ the shell loop, round 3
// The loop above costs ~140 instruction slots per cell and shell, nearly all of it fp64: the table
// lookup + exp, five dependent double products per source term, three two-point interpolations of
// the frequency integrals with per-shell scalars fetched from LDS one by one.  Everything that does
// not depend on the cell is folded per shell ONCE per workgroup instead:
//   c1[R] = z_edge avg_fix sfr_scale                      (Lagrangian grids: z_edge)
//   c2[R] = c1[R] xray_scale xray_R                       (Lagrangian: z_edge xray_R 1e38)
//   W[R][k][m] = c2[R] freq_int_k[m][R],  S[R][j] = c1[R] {starlya, lya_cont, lya_inj}[R]
// and, the interpolation weight of a cell being the same for every shell,
//   sum_R x_R ((f[m+1][R] - f[m][R]) ival + f[m][R]) = lo + ival (hi - lo),
//   lo = sum_R x_R W[R][k][m],  hi = sum_R x_R W[R][k][m+1]
// so a cell and shell cost seven fp64 FMAs.  The source term itself is a float upstream
// (del_fcoll_Rct, :1040-1079) and is evaluated to float accuracy in fp32: bin index and weight
// from one FMA, the table difference times the weight added to an EXACT float knot inside a
// two-term base-2 reduction (so the argument of the hardware exp2 carries ~5e-8, not the 1e-6 a
// float holding ln SFRD ~ -20 would), (1 + delta) exp(.) in fp32.  Against the double evaluation
// rounded to float once that is <= 3e-7 per term, random in sign; x_e and T_k move by a few 1e-8
// (parity bound 2e-6, tests/test_gpu_ts.py).  ~45 slots per cell and shell: the sweep is within
// reach of the 4 B it reads (19.2 -> see DESIGN for the measured figure).  C21CM_TS_LOOP=v1
// selects the kernels above for A/B runs.
struct ShellLookup {
    float gw, off;  // t = delta * gw + off,  gw = growth / width, off = -tab_min / width
    float growth;
};

// value of the per-shell table at the cell (E-INTEGRAL: exp of the ln SFRD table) in fp32
template <bool EXP>
__device__ __forceinline__ float shell_table_f32(float dens, const ShellLookup &L,
                                                 const float *__restrict__ y, float *curr_dens) {
    const float x = __fmul_rn(dens, L.growth);
    *curr_dens = x;
    const float t = __fmaf_rn(dens, L.gw, L.off);
    // the cells that define the table range sit on its first / last knot: keep the bin inside
    // [0, NDELTA - 2] whatever the last bit of t says (upstream reads y[idx + 1] with weight 0 there)
    const int idx = min(max((int)floorf(t), 0), C21CM_NDELTA_TABLE - 2);
    const float ip = t - (float)idx;
    const float y0 = y[idx], y1 = y[idx + 1];
    const float r = __fmul_rn(ip, y1 - y0);
    if (!EXP) return y0 + r;
    // exp(y0 + r) = 2^n 2^f,  n = rint(y0 log2e),  f = (y0 L_hi - n) + y0 L_lo + r L_hi
    const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-8f;
    const float n = rintf(__fmul_rn(y0, L_hi));
    const float f = __fmaf_rn(y0, L_hi, -n) + __fmaf_rn(y0, L_lo, __fmul_rn(r, L_hi));
    // (n < -126 - 24: the float result is zero anyway; ldexpf handles the subnormal range)
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)fmaxf(n, -200.f));
}

// box sum of the table values of one shell (blockIdx.y), fp32 lookup, fp64 accumulation
template <bool EXP>
__global__ void __launch_bounds__(kBlock)
sfrd_sum2_kernel(const float *__restrict__ filtered_density, const float *__restrict__ tables,
                 const double *__restrict__ shell, int n_step, size_t ntot,
                 double *__restrict__ partials) {
    __shared__ double lds[kBlock];
    __shared__ float tab[C21CM_NDELTA_TABLE];  // (round 6: the shell's table in LDS -- the per-lane gathers went to L1)
    const int R = blockIdx.y;
    for (int t = threadIdx.x; t < C21CM_NDELTA_TABLE; t += kBlock) tab[t] = tables[(size_t)R * C21CM_NDELTA_TABLE + t];
    __syncthreads();
    const double inv_w = shell[SH_TABINVW * n_step + R];
    ShellLookup L;
    L.growth = (float)shell[SH_GROWTH * n_step + R];
    L.gw = (float)(shell[SH_GROWTH * n_step + R] * inv_w);
    L.off = (float)(-shell[SH_TABMIN * n_step + R] * inv_w);
    const float4 *d4 = reinterpret_cast<const float4 *>(filtered_density + (size_t)R * ntot);
    double acc = 0.;
    const size_t n4 = ntot / 4;  // ntot % 4 == 0 (launcher)
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * kBlock) {
        const float4 c = d4[i];
        float cd;
        const float a0 = shell_table_f32<EXP>(c.x, L, tab, &cd);
        const float a1 = shell_table_f32<EXP>(c.y, L, tab, &cd);
        const float a2 = shell_table_f32<EXP>(c.z, L, tab, &cd);
        const float a3 = shell_table_f32<EXP>(c.w, L, tab, &cd);
        // pairwise in fp32 (four values of similar size), then into the double
        acc += (double)((a0 + a1) + (a2 + a3));
    }
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[(size_t)R * gridDim.x + blockIdx.x] = acc;
}

// LDS image of the folded per-shell constants: [n][NW] doubles then [n] ShellLookup
constexpr int kW_STAR = 3 * C21CM_X_INT_NXHII;  // W rows: heat[14], ion[14], lya[14], then S
constexpr int kNW = kW_STAR + 3;                 // + starlya, cont, inj

template <int VEC, int MODE>  // MODE 0: Lagrangian grids, 1: ln SFRD tables, 2: dfcoll/dz tables
__global__ void __launch_bounds__(kBlock)
ts_accumulate2_kernel(c21hip_ts_args a, const float *__restrict__ prev_xe,
                      const float *__restrict__ grid_a, const float *__restrict__ grid_b,
                      const float *__restrict__ tables, const double *__restrict__ dev_tab,
                      double *__restrict__ sums, size_t ntot) {
    extern __shared__ double sh[];
    const int n = a.n_step;
    double *W = sh;                                                  // [n][kNW]
    ShellLookup *LK = reinterpret_cast<ShellLookup *>(sh + n * kNW);  // [n]
    {
        const double *fheat = dev_tab + SH_COUNT * n, *fion = fheat + C21CM_X_INT_NXHII * n,
                     *flya = fion + C21CM_X_INT_NXHII * n;
        for (int i = threadIdx.x; i < n * kNW; i += kBlock) {
            const int R = i / kNW, j = i - R * kNW;
            const double z_edge = dev_tab[SH_ZEDGE * n + R], xray_R = dev_tab[SH_XRAY_R * n + R];
            double c1, c2;
            if (MODE == 0) {
                c1 = z_edge;
                c2 = z_edge * xray_R * 1e38;
            } else {
                c1 = z_edge * dev_tab[SH_AVGFIX * n + R] * a.sfr_scale;
                c2 = c1 * a.xray_scale * xray_R;
            }
            double v;
            if (j < kW_STAR) {
                const int k = j / C21CM_X_INT_NXHII, m = j - k * C21CM_X_INT_NXHII;
                const double *f = k == 0 ? fheat : (k == 1 ? fion : flya);
                v = c2 * f[m * n + R];
            } else {
                const int row = j == kW_STAR ? SH_STARLYA : (j == kW_STAR + 1 ? SH_CONT : SH_INJ);
                v = c1 * dev_tab[row * n + R];
            }
            W[i] = v;
        }
        for (int R = threadIdx.x; R < n; R += kBlock) {
            const double inv_w = dev_tab[SH_TABINVW * n + R];
            LK[R].growth = (float)dev_tab[SH_GROWTH * n + R];
            LK[R].gw = (float)(dev_tab[SH_GROWTH * n + R] * inv_w);
            LK[R].off = (float)(-dev_tab[SH_TABMIN * n + R] * inv_w);
        }
    }
    __syncthreads();
    const size_t nitems = ntot / VEC;  // ntot % VEC == 0 (launcher)
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < nitems;
         it += (size_t)gridDim.x * kBlock) {
        const auto pxe = FloatVec<VEC>::load(prev_xe, it);
        int m[VEC];
        double ival[VEC];
        double lo[VEC][3], hi[VEC][3], star[VEC], cont[VEC], inj[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            // :1499-1514, float arithmetic as upstream
            float xHII_call = pxe.v[e];
            if (xHII_call > kXHII[C21CM_X_INT_NXHII - 1] * 0.999)
                xHII_call = (float)(kXHII[C21CM_X_INT_NXHII - 1] * 0.999);
            else if (xHII_call < kXHII[0])
                xHII_call = (float)(1.001 * kXHII[0]);
            int mm = C21CM_X_INT_NXHII - 1;
            while (xHII_call < kXHII[mm]) mm--;
            const float inv_diff = (float)(1. / (kXHII[mm + 1] - kXHII[mm]));
            m[e] = mm;
            ival[e] = (double)((xHII_call - kXHII[mm]) * inv_diff);
#pragma unroll
            for (int k = 0; k < 3; k++) lo[e][k] = hi[e][k] = 0.;
            star[e] = cont[e] = inj[e] = 0.;
        }
        FloatVec<VEC> ga = FloatVec<VEC>::load(grid_a, (size_t)(n - 1) * nitems + it), gb = ga;
        if (MODE == 0) gb = FloatVec<VEC>::load(grid_b, (size_t)(n - 1) * nitems + it);
        for (int R = n; R--;) {
            const FloatVec<VEC> ca = ga, cb = gb;
            if (R > 0) {  // request the next (smaller) shell now
                ga = FloatVec<VEC>::load(grid_a, (size_t)(R - 1) * nitems + it);
                if (MODE == 0) gb = FloatVec<VEC>::load(grid_b, (size_t)(R - 1) * nitems + it);
            }
            const double *Wr = W + R * kNW;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                double xs, xx;  // the star-formation term / the X-ray term before their constants
                if (MODE == 0) {
                    xs = (double)ca.v[e];
                    xx = (double)cb.v[e];
                } else {
                    float cd;
                    const float tv = shell_table_f32<MODE == 1>(
                        ca.v[e], LK[R], tables + (size_t)R * C21CM_NDELTA_TABLE, &cd);
                    xs = xx = (double)__fmul_rn(1.f + cd, tv);  // del_fcoll_Rct is a float
                }
                const double *w = Wr + m[e];
                if (a.use_xray_heating) {
                    lo[e][0] = fma(xx, w[0], lo[e][0]);
                    hi[e][0] = fma(xx, w[1], hi[e][0]);
                }
                lo[e][1] = fma(xx, w[C21CM_X_INT_NXHII], lo[e][1]);
                hi[e][1] = fma(xx, w[C21CM_X_INT_NXHII + 1], hi[e][1]);
                lo[e][2] = fma(xx, w[2 * C21CM_X_INT_NXHII], lo[e][2]);
                hi[e][2] = fma(xx, w[2 * C21CM_X_INT_NXHII + 1], hi[e][2]);
                star[e] = fma(xs, Wr[kW_STAR], star[e]);
                if (a.use_lya_heating) {
                    cont[e] = fma(xs, Wr[kW_STAR + 1], cont[e]);
                    inj[e] = fma(xs, Wr[kW_STAR + 2], inj[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const size_t i = it * VEC + e;
            sums[i] = lo[e][0] + ival[e] * (hi[e][0] - lo[e][0]);
            sums[ntot + i] = lo[e][1] + ival[e] * (hi[e][1] - lo[e][1]);
            sums[2 * ntot + i] = lo[e][2] + ival[e] * (hi[e][2] - lo[e][2]);
            sums[3 * ntot + i] = star[e];
            if (a.use_lya_heating) {
                sums[4 * ntot + i] = cont[e];
                sums[5 * ntot + i] = inj[e];
            }
        }
    }
}

// ---- table modes, third version: everything a shell needs sits in LDS -- the 400-knot tables of
// all shells (64 KB for 40), the folded frequency-integral weights as (w[m], w[m+1]) pairs (one
// 16-byte read per integral) -- so a workgroup is 1024 threads, one per CU (16 waves: the loop is
// issue-bound, not occupancy-bound, at ~40 slots per cell and shell); two cells per thread with the
// lookup written on 2-vectors so that the compiler emits packed fp32 instructions for both cells.
#ifndef C21X_TS_PACKED
#define C21X_TS_PACKED 0
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kAccBlock = 1024;
constexpr int kWP = 3 * C21CM_X_INT_NXHII;  // (w[m], w[m+1]) pairs per shell: heat, ion, lya

template <int MODE>  // 1: ln SFRD tables (exp), 2: dfcoll/dz tables
__global__ void __launch_bounds__(kAccBlock)
ts_accumulate3_kernel(c21hip_ts_args a, const float *__restrict__ prev_xe,
                      const float *__restrict__ grid_a, const float *__restrict__ tables,
                      const double *__restrict__ dev_tab, double *__restrict__ sums, size_t ntot) {
    extern __shared__ double sh[];
    const int n = a.n_step;
    double2 *WP = reinterpret_cast<double2 *>(sh);                    // [n][kWP]
    double *SS = sh + 2 * (size_t)n * kWP;                            // [n][3]: starlya, cont, inj
    ShellLookup *LK = reinterpret_cast<ShellLookup *>(SS + 3 * n);    // [n]
    float *TAB = reinterpret_cast<float *>(LK + n);                   // [n][NDELTA + 1]
    constexpr int TS = C21CM_NDELTA_TABLE + 1;  // odd row pitch; one slack float per table
    {
        const double *fheat = dev_tab + SH_COUNT * n, *fion = fheat + C21CM_X_INT_NXHII * n,
                     *flya = fion + C21CM_X_INT_NXHII * n;
        for (int i = threadIdx.x; i < n * kWP; i += kAccBlock) {
            const int R = i / kWP, j = i - R * kWP;
            const double c1 = dev_tab[SH_ZEDGE * n + R] * dev_tab[SH_AVGFIX * n + R] * a.sfr_scale;
            const double c2 = c1 * a.xray_scale * dev_tab[SH_XRAY_R * n + R];
            const int k = j / C21CM_X_INT_NXHII, m = j - k * C21CM_X_INT_NXHII;
            const double *f = k == 0 ? fheat : (k == 1 ? fion : flya);
            const int m1 = min(m + 1, C21CM_X_INT_NXHII - 1);  // (m <= NXHII - 2 for every cell)
            WP[i] = make_double2(c2 * f[m * n + R], c2 * f[m1 * n + R]);
        }
        for (int i = threadIdx.x; i < 3 * n; i += kAccBlock) {
            const int R = i / 3, j = i - 3 * R;
            const double c1 = dev_tab[SH_ZEDGE * n + R] * dev_tab[SH_AVGFIX * n + R] * a.sfr_scale;
            const int row = j == 0 ? SH_STARLYA : (j == 1 ? SH_CONT : SH_INJ);
            SS[i] = c1 * dev_tab[row * n + R];
        }
        for (int R = threadIdx.x; R < n; R += kAccBlock) {
            const double inv_w = dev_tab[SH_TABINVW * n + R];
            LK[R].growth = (float)dev_tab[SH_GROWTH * n + R];
            LK[R].gw = (float)(dev_tab[SH_GROWTH * n + R] * inv_w);
            LK[R].off = (float)(-dev_tab[SH_TABMIN * n + R] * inv_w);
        }
        for (int i = threadIdx.x; i < n * TS; i += kAccBlock) {
            const int R = i / TS, j = i - R * TS;
            TAB[i] = j < C21CM_NDELTA_TABLE ? tables[(size_t)R * C21CM_NDELTA_TABLE + j] : 0.f;
        }
    }
    __syncthreads();
    const size_t nitems = ntot / 2;  // ntot even (launcher)
    const float2 *ga2 = reinterpret_cast<const float2 *>(grid_a);
    for (size_t it = (size_t)blockIdx.x * kAccBlock + threadIdx.x; it < nitems;
         it += (size_t)gridDim.x * kAccBlock) {
        const float2 pxe = reinterpret_cast<const float2 *>(prev_xe)[it];
        const float pxv[2] = {pxe.x, pxe.y};
        int m[2];
        double ival[2];
        double lo[2][3], hi[2][3], star[2], cont[2], inj[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            // :1499-1514, float arithmetic as upstream
            float xHII_call = pxv[e];
            if (xHII_call > kXHII[C21CM_X_INT_NXHII - 1] * 0.999)
                xHII_call = (float)(kXHII[C21CM_X_INT_NXHII - 1] * 0.999);
            else if (xHII_call < kXHII[0])
                xHII_call = (float)(1.001 * kXHII[0]);
            int mm = C21CM_X_INT_NXHII - 1;
            while (xHII_call < kXHII[mm]) mm--;
            const float inv_diff = (float)(1. / (kXHII[mm + 1] - kXHII[mm]));
            m[e] = mm;
            ival[e] = (double)((xHII_call - kXHII[mm]) * inv_diff);
#pragma unroll
            for (int k = 0; k < 3; k++) lo[e][k] = hi[e][k] = 0.;
            star[e] = cont[e] = inj[e] = 0.;
        }
        float2 g = ga2[(size_t)(n - 1) * nitems + it];
        for (int R = n; R--;) {
            const float2 c = g;
            if (R > 0) g = ga2[(size_t)(R - 1) * nitems + it];  // the next (smaller) shell
            const ShellLookup L = LK[R];
            const float *y = TAB + R * TS;
#if C21X_TS_PACKED  // (the round-3 form: 2-vectors so that the compiler emits packed fp32 instructions)
            const v2f d = {c.x, c.y};
            const v2f x = d * L.growth;
            const v2f t = __builtin_elementwise_fma(d, (v2f){L.gw, L.gw}, (v2f){L.off, L.off});
            const int i0 = min(max((int)floorf(t.x), 0), C21CM_NDELTA_TABLE - 2);
            const int i1 = min(max((int)floorf(t.y), 0), C21CM_NDELTA_TABLE - 2);
            const v2f ip = t - (v2f){(float)i0, (float)i1};
            const v2f y0 = {y[i0], y[i1]}, y1 = {y[i0 + 1], y[i1 + 1]};
            const v2f r = ip * (y1 - y0);
            v2f tv;
            if (MODE == 1) {
                const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-8f;
                const v2f w = y0 * L_hi;
                const v2f nn = {rintf(w.x), rintf(w.y)};
                const v2f f = __builtin_elementwise_fma(y0, (v2f){L_hi, L_hi}, -nn) +
                              __builtin_elementwise_fma(y0, (v2f){L_lo, L_lo}, r * L_hi);
                tv.x = ldexpf(__builtin_amdgcn_exp2f(f.x), (int)fmaxf(nn.x, -200.f));
                tv.y = ldexpf(__builtin_amdgcn_exp2f(f.y), (int)fmaxf(nn.y, -200.f));
            } else {
                tv = y0 + r;
            }
            const v2f sf = (x + 1.0f) * tv;  // del_fcoll_Rct is a float upstream
#else
            // the same arithmetic on scalars (round 5: a packed fp32 instruction issues in 8 cycles per wave on
            // gfx950, a plain one in 2.7 -- tools/valu_rate_probe.hip; the operations and their order are those of
            // the packed form, so the sums are the same bits)
            float2 sf;
            {
                const float cc[2] = {c.x, c.y};
                float sv[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float dd = cc[e];
                    const float xg = __fmul_rn(dd, L.growth);
                    const float tt = __fmaf_rn(dd, L.gw, L.off);
                    const int ii = min(max((int)floorf(tt), 0), C21CM_NDELTA_TABLE - 2);
                    const float ipf = __fsub_rn(tt, (float)ii);
                    const float ya = y[ii], yb = y[ii + 1];
                    const float rr = __fmul_rn(ipf, __fsub_rn(yb, ya));
                    float tvv;
                    if (MODE == 1) {
                        const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-8f;
                        const float nn = rintf(__fmul_rn(ya, L_hi));
                        const float ff = __fadd_rn(__fmaf_rn(ya, L_hi, -nn), __fmaf_rn(ya, L_lo, __fmul_rn(rr, L_hi)));
                        tvv = ldexpf(__builtin_amdgcn_exp2f(ff), (int)fmaxf(nn, -200.f));
                    } else {
                        tvv = __fadd_rn(ya, rr);
                    }
                    sv[e] = __fmul_rn(__fadd_rn(xg, 1.0f), tvv);  // del_fcoll_Rct is a float upstream
                }
                sf = make_float2(sv[0], sv[1]);
            }
#endif
            const double xs[2] = {(double)sf.x, (double)sf.y};
            const double2 *wr = WP + R * kWP;
            const double *sr = SS + 3 * R;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (a.use_xray_heating) {
                    const double2 w0 = wr[m[e]];
                    lo[e][0] = fma(xs[e], w0.x, lo[e][0]);
                    hi[e][0] = fma(xs[e], w0.y, hi[e][0]);
                }
                const double2 w1 = wr[C21CM_X_INT_NXHII + m[e]], w2 = wr[2 * C21CM_X_INT_NXHII + m[e]];
                lo[e][1] = fma(xs[e], w1.x, lo[e][1]);
                hi[e][1] = fma(xs[e], w1.y, hi[e][1]);
                lo[e][2] = fma(xs[e], w2.x, lo[e][2]);
                hi[e][2] = fma(xs[e], w2.y, hi[e][2]);
                star[e] = fma(xs[e], sr[0], star[e]);
                if (a.use_lya_heating) {
                    cont[e] = fma(xs[e], sr[1], cont[e]);
                    inj[e] = fma(xs[e], sr[2], inj[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const size_t i = it * 2 + e;
            sums[i] = lo[e][0] + ival[e] * (hi[e][0] - lo[e][0]);
            sums[ntot + i] = lo[e][1] + ival[e] * (hi[e][1] - lo[e][1]);
            sums[2 * ntot + i] = lo[e][2] + ival[e] * (hi[e][2] - lo[e][2]);
            sums[3 * ntot + i] = star[e];
            if (a.use_lya_heating) {
                sums[4 * ntot + i] = cont[e];
                sums[5 * ntot + i] = inj[e];
            }
        }
    }
}
inline size_t ts_acc3_lds(int n) {
    return (size_t)n * (kWP * sizeof(double2) + 3 * sizeof(double) + sizeof(ShellLookup) +
                        (C21CM_NDELTA_TABLE + 1) * sizeof(float));
}

// Sweep 2 of 2 -- prefactors and get_Ts_fast per cell; `sums` NULL: nothing has formed yet.
__global__ void __launch_bounds__(kBlock)
ts_cell_kernel(c21hip_ts_args a, const float *__restrict__ density,
               const float *__restrict__ prev_Ts, const float *__restrict__ prev_Tk,
               const float *__restrict__ prev_xe, const double *__restrict__ sums,
               const double *__restrict__ lya_dEC, const double *__restrict__ lya_dEI,
               float *__restrict__ Ts_out, float *__restrict__ Tk_out, float *__restrict__ xe_out,
               size_t ntot, double *__restrict__ partials, int *__restrict__ flag) {
    __shared__ double red[kBlock];
    double s_Ts = 0, s_Tk = 0, s_xe = 0, s_Ja = 0, s_heat = 0, s_ion = 0;
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        CellSums q{0., 0., 0., 0., 0., 0.};
        if (sums) {
            q.heat = sums[i], q.ion = sums[ntot + i], q.lya = sums[2 * ntot + i];
            q.starlya = sums[3 * ntot + i];
            if (a.use_lya_heating) q.cont = sums[4 * ntot + i], q.inj = sums[5 * ntot + i];
        }
        const CellOut o = ts_cell_epilogue(a, q, density[i], prev_Ts[i], prev_Tk[i], prev_xe[i],
                                           lya_dEC, lya_dEI);
        const float Ts_f = (float)o.Ts;
        Ts_out[i] = Ts_f;
        Tk_out[i] = (float)o.Tk;
        xe_out[i] = (float)o.xe;
        if (!isfinite(Ts_f)) bad = 1;
        s_Ts += o.Ts, s_Tk += o.Tk, s_xe += o.xe, s_Ja += o.J_alpha, s_heat += o.xheat, s_ion += o.xion;
    }
    if (bad) atomicOr(flag, 1);
    double sums6[6] = {s_Ts, s_Tk, s_xe, s_Ja, s_heat, s_ion};
    for (int k = 0; k < 6; k++) {
        const double r = block_sum(sums6[k], red);
        if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = r;
    }
}

// the six box sums of ts_cell_kernel's per-block partials
__global__ void __launch_bounds__(kBlock)
ts_finish_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out) {
    __shared__ double lds[kBlock];
    const int k = blockIdx.x;
    double acc = 0.;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) acc += partials[(size_t)k * nblocks + i];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) out[k] = acc;
}

// init_first_Ts (:892-927): T_k = TK (1 + cT_ad delta), x_e, collisional T_s at the mean TK
__global__ void __launch_bounds__(kBlock)
ts_first_kernel(const float *__restrict__ density, float inv_growth_z, float growth_zp, double TK,
                double xe_d, double cT_ad, double xc_per_density, float TKf, double Trad,
                float *__restrict__ Ts_out, float *__restrict__ Tk_out, float *__restrict__ xe_out,
                size_t ntot) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const double gdens = (double)(density[i] * inv_growth_z * growth_zp);  // float products
        Tk_out[i] = (float)(TK * (1.0 + cT_ad * gdens));
        xe_out[i] = (float)xe_d;
        const float delta = (float)gdens;
        const double xc = xc_per_density * (1.0 + delta);
        Ts_out[i] = (float)((1.0 + xc) / (1.0 / Trad + xc / TKf));
    }
}
}  // namespace

extern "C" size_t c21hip_ts_table_doubles(int n_step) {
    return (size_t)(SH_COUNT + 3 * C21CM_X_INT_NXHII) * n_step;
}

extern "C" int c21hip_ts_sfrd_means(const float *filtered_density, const float *tables_dev,
                                    int table_exp, double *dev_tab,
                                    const double *mean_sfr_zpp_dev, int n_step,
                                    size_t ntot, double *partials, double *ave_out_dev,
                                    void *stream) {
    int bx = grid_for(ntot);
    if (bx > 512) bx = 512;  // n_step rows of blocks fill the chip
    const bool v2 = !ts_loop_v1() && (ntot & 3) == 0 && ((size_t)filtered_density & 15) == 0;
    if (v2 && table_exp)
        hipLaunchKernelGGL((sfrd_sum2_kernel<true>), dim3(bx, n_step), dim3(kBlock), 0,
                           (hipStream_t)stream, filtered_density, tables_dev, dev_tab, n_step, ntot,
                           partials);
    else if (v2)
        hipLaunchKernelGGL((sfrd_sum2_kernel<false>), dim3(bx, n_step), dim3(kBlock), 0,
                           (hipStream_t)stream, filtered_density, tables_dev, dev_tab, n_step, ntot,
                           partials);
    else
        hipLaunchKernelGGL(sfrd_sum_kernel, dim3(bx, n_step), dim3(kBlock), 0, (hipStream_t)stream,
                           filtered_density, tables_dev, table_exp, dev_tab, n_step, ntot, partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(sfrd_finish_kernel, dim3(n_step), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, bx, mean_sfr_zpp_dev, (double)ntot, n_step, dev_tab, ave_out_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ts_cells(const c21hip_ts_args *a, const float *density, const float *prev_Ts,
                               const float *prev_Tk, const float *prev_xe, const float *grid_a,
                               const float *grid_b, const float *tables_dev, const double *dev_tab,
                               const double *lya_dEC_dev, const double *lya_dEI_dev, float *Ts_out,
                               float *Tk_out, float *xe_out, size_t ntot, double *sums_ws,
                               double *partials, double *sums_out_dev, int *flag_dev, void *stream) {
    const size_t lds = c21hip_ts_table_doubles(a->n_step) * sizeof(double);
    if (lds > 64 * 1024) {
        c21hip_set_error("spin temperature: %d shells do not fit the table cache", a->n_step);
        return C21CM_VALUE_ERROR;
    }
    if (!a->no_light && !a->sums_ready) {
        int st = c21hip_ts_shell_loop(a, prev_xe, grid_a, grid_b, tables_dev, dev_tab, sums_ws, ntot,
                                      stream);
        if (st) return st;
    }
    const int blocks = grid_for(ntot);
    hipLaunchKernelGGL(ts_cell_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, *a, density,
                       prev_Ts, prev_Tk, prev_xe, a->no_light ? nullptr : sums_ws, lya_dEC_dev,
                       lya_dEI_dev, Ts_out, Tk_out, xe_out, ntot, partials, flag_dev);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ts_finish_kernel, dim3(6), dim3(kBlock), 0, (hipStream_t)stream, partials,
                       blocks, sums_out_dev);
    LAUNCH_CHECK();
    return 0;
}

// The shell loop alone: the six sums of every cell into sums_ws ([6][ntot] doubles)
extern "C" int c21hip_ts_shell_loop(const c21hip_ts_args *a, const float *prev_xe,
                                    const float *grid_a, const float *grid_b,
                                    const float *tables_dev, const double *dev_tab, double *sums_ws,
                                    size_t ntot, void *stream) {
    const size_t lds = c21hip_ts_table_doubles(a->n_step) * sizeof(double);
    {
        // two cells per thread (8-byte loads) when the arrays allow it.  Measured at 512^3, 40
        // shells, SFRD tables: 19.1 ms with one or two cells per thread (78 / 102 VGPRs), 25.6 ms
        // with four (170 VGPRs): the loop is bound by its ~140 instruction slots per cell and shell
        // (table lookup, exp, three interpolated frequency integrals in fp64), not by HBM.
        auto aligned8 = [](const void *p) { return ((size_t)p & 7) == 0; };
        const bool vec2 = (ntot & 1) == 0 && aligned8(prev_xe) && aligned8(grid_a) && aligned8(grid_b);
        const int blocks = grid_for(vec2 ? ntot / 2 : ntot);
        const size_t lds2 = (size_t)a->n_step * (kNW * sizeof(double) + sizeof(ShellLookup));
        const char *loop_env = getenv("C21CM_TS_LOOP");
        const bool v3 = !ts_loop_v1() && !(loop_env && loop_env[0] == 'v' && loop_env[1] == '2') &&
                        vec2 && !a->lagrangian && ts_acc3_lds(a->n_step) <= 160 * 1024;
        if (v3) {
            const size_t lds3 = ts_acc3_lds(a->n_step);
            static size_t attr3 = 0;
            if (lds3 > attr3) {
                (void)hipFuncSetAttribute((const void *)ts_accumulate3_kernel<1>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                (void)hipFuncSetAttribute((const void *)ts_accumulate3_kernel<2>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                attr3 = lds3;
            }
            int nb = (int)((ntot / 2 + kAccBlock - 1) / kAccBlock);
            if (nb > 256) nb = 256;  // one 1024-thread workgroup per CU
            if (a->table_exp)
                hipLaunchKernelGGL((ts_accumulate3_kernel<1>), dim3(nb), dim3(kAccBlock), lds3,
                                   (hipStream_t)stream, *a, prev_xe, grid_a, tables_dev, dev_tab, sums_ws, ntot);
            else
                hipLaunchKernelGGL((ts_accumulate3_kernel<2>), dim3(nb), dim3(kAccBlock), lds3,
                                   (hipStream_t)stream, *a, prev_xe, grid_a, tables_dev, dev_tab, sums_ws, ntot);
        } else if (!ts_loop_v1()) {
            const int mode = a->lagrangian ? 0 : (a->table_exp ? 1 : 2);
#define TS_ACC2(V, M)                                                                              \
    hipLaunchKernelGGL((ts_accumulate2_kernel<V, M>), dim3(blocks), dim3(kBlock), lds2,            \
                       (hipStream_t)stream, *a, prev_xe, grid_a, grid_b, tables_dev, dev_tab,      \
                       sums_ws, ntot)
            if (vec2) {
                if (mode == 0) TS_ACC2(2, 0);
                else if (mode == 1) TS_ACC2(2, 1);
                else TS_ACC2(2, 2);
            } else {
                if (mode == 0) TS_ACC2(1, 0);
                else if (mode == 1) TS_ACC2(1, 1);
                else TS_ACC2(1, 2);
            }
#undef TS_ACC2
        } else if (vec2)
            hipLaunchKernelGGL((ts_accumulate_kernel<2>), dim3(blocks), dim3(kBlock), lds,
                               (hipStream_t)stream, *a, prev_xe, grid_a, grid_b, tables_dev, dev_tab,
                               sums_ws, ntot);
        else
            hipLaunchKernelGGL((ts_accumulate_kernel<1>), dim3(blocks), dim3(kBlock), lds,
                               (hipStream_t)stream, *a, prev_xe, grid_a, grid_b, tables_dev, dev_tab,
                               sums_ws, ntot);
        LAUNCH_CHECK();
    }
    return 0;
}

// host-side collision rates at the mean temperature (get_Ts: float z, TK, xe)
static double host_kappa(const double *y, double width, double lnT_max, double lnT, int power_law) {
    double ans;
    if (lnT < 0.)
        ans = y[0];
    else if (lnT > lnT_max) {
        if (power_law)
            ans = log(exp(y[C21CM_KAPPA_NPTS - 1]) * pow(exp(lnT) / exp(lnT_max), 0.381));
        else
            ans = y[C21CM_KAPPA_NPTS - 1] + (y[C21CM_KAPPA_NPTS - 1] - y[C21CM_KAPPA_NPTS - 2]) /
                                                (lnT_max - width * (C21CM_KAPPA_NPTS - 2)) *
                                                (lnT - lnT_max);
    } else {
        int idx = (int)floor(lnT * (1. / width));
        if (idx > C21CM_KAPPA_NPTS - 2) idx = C21CM_KAPPA_NPTS - 2;
        ans = y[idx] + (lnT - width * (double)idx) * (y[idx + 1] - y[idx]) * (1. / width);
    }
    return exp(ans);
}

extern "C" void c21hip_kappa_rates(double T, double *k_HH, double *k_eH, double *k_pH) {
    static const double hh[C21CM_KAPPA_NPTS] = C21CM_KAPPA_HH_VALUES;
    static const double ph[C21CM_KAPPA_NPTS] = C21CM_KAPPA_PH_VALUES;
    static const double eh[C21CM_KAPPA_NPTS] = C21CM_KAPPA_EH_VALUES;
    const double lnT = log(T);
    *k_HH = host_kappa(hh, C21CM_KAPPA_HH_BINWIDTH, C21CM_KAPPA_HH_LNT_MAX, lnT, 1);
    *k_eH = host_kappa(eh, C21CM_KAPPA_EH_BINWIDTH, C21CM_KAPPA_EH_LNT_MAX, lnT, 0);
    *k_pH = host_kappa(ph, C21CM_KAPPA_PH_BINWIDTH, C21CM_KAPPA_PH_LNT_MAX, lnT, 0);
}

extern "C" int c21hip_ts_first(const c21cm_ts_first_spec *s, const float *density, float *Ts_out,
                               float *Tk_out, float *xe_out, size_t ntot, void *stream) {
    const float z = (float)s->perturbed_redshift, xe = (float)s->xe, TK = (float)s->TK;
    const double Trad = s->T_cmb * (1.0 + z);
    double k_HH, k_eH, k_pH;
    c21hip_kappa_rates(TK, &k_HH, &k_eH, &k_pH);
    // xcoll (heating_helper_progs.c:695-728) without its (1 + delta) factor
    const double cube = pow(1.0 + z, 3.0);
    const double xc_per_density = s->T_21 / Trad * ((1.0 - xe) * s->No * cube) * k_HH / s->A10 +
                                  s->T_21 / Trad * (xe * s->N_b0 * cube) * k_eH / s->A10 +
                                  s->T_21 / Trad * (xe * s->No * cube) * k_pH / s->A10;
    hipLaunchKernelGGL(ts_first_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0, (hipStream_t)stream,
                       density, s->inverse_growth_factor_z, s->growth_factor_zp, s->TK, s->xe,
                       s->cT_ad, xc_per_density, TK, Trad, Ts_out, Tk_out, xe_out, ntot);
    LAUNCH_CHECK();
    return 0;
}


// ======================================================================================
// USE_MINI_HALOS (E-INTEGRAL): the molecularly cooled population in the shell loop
// (SpinTemperatureBox.c:1011-1075, 1642-1733, 1843-1845).  Separate kernels: the tuned
// one-population loop above keeps its register budget; this branch adds a 2-D table lookup and a
// second exponential per cell and shell, and is written for parity first.
// ======================================================================================
namespace {
// rows of the mini shell buffer (doubles, n = n_step)
enum { MS_STARLYA = 0, MS_CONT, MS_INJ, MS_LW, MS_LW_MINI, MS_AVGFIX, MS_COUNT };
static_assert(MS_COUNT == C21HIP_TS_MINI_ROWS, "rows of the mini shell buffer");

// interpolation.c:133-157
__device__ __forceinline__ double table_2d(double x, double y, double x_min, double x_width,
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

// box sum of the mini SFRD table values of one shell (blockIdx.y)
__global__ void __launch_bounds__(kBlock)
sfrd_sum_mini_kernel(const float *__restrict__ filtered_density,
                     const float *__restrict__ filtered_mcrit, const float *__restrict__ tables2,
                     const double *__restrict__ shell, int n_step, size_t ntot, double mt_min,
                     double mt_width, double *__restrict__ partials) {
    __shared__ double lds[kBlock];
    const int R = blockIdx.y;
    const float *dens = filtered_density + (size_t)R * ntot;
    const float *mcrit = filtered_mcrit + (size_t)R * ntot;
    const float *tab = tables2 + (size_t)R * C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
    const double growth = shell[SH_GROWTH * n_step + R], tab_min = shell[SH_TABMIN * n_step + R],
                 tab_width = shell[SH_TABWIDTH * n_step + R];
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock)
        acc += exp(table_2d((double)dens[i] * growth, (double)mcrit[i], tab_min, tab_width, mt_min,
                            mt_width, tab));
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[(size_t)R * gridDim.x + blockIdx.x] = acc;
}

// avg_fix_term_MINI = mean_sfr_zpp_mini / (sum / N) per shell (:1617-1618)
__global__ void __launch_bounds__(kBlock)
sfrd_finish_mini_kernel(const double *__restrict__ partials, int nblocks,
                        const double *__restrict__ mean_sfr_zpp_mini, double ntot, int n_step,
                        double *__restrict__ mini_shell, double *__restrict__ ave_out) {
    __shared__ double lds[kBlock];
    const int R = blockIdx.x;
    double acc = 0.;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) acc += partials[(size_t)R * nblocks + i];
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) {
        const double ave = acc / ntot;
        ave_out[R] = ave;
        mini_shell[MS_AVGFIX * n_step + R] = mean_sfr_zpp_mini[R] / ave;
    }
}

struct MiniScalars {
    double sfr_scale_mini, xray_scale_mini, mt_min, mt_width, lw_scale;
};

// the shell loop with both populations: six sums to `sums`, the Lyman-Werner background to J_21_LW
__global__ void __launch_bounds__(kBlock)
ts_accumulate_mini_kernel(c21hip_ts_args a, MiniScalars ms, const float *__restrict__ prev_xe,
                          const float *__restrict__ delNL0, const float *__restrict__ mcrit,
                          const float *__restrict__ tables, const float *__restrict__ tables2,
                          const double *__restrict__ dev_tab, const double *__restrict__ mini_shell,
                          double *__restrict__ sums, float *__restrict__ J_21_LW, size_t ntot) {
    extern __shared__ double sh[];  // the one-population table buffer, then the mini rows
    const int n = a.n_step;
    const int n_tab = (SH_COUNT + 3 * C21CM_X_INT_NXHII) * n;
    for (int i = threadIdx.x; i < n_tab; i += kBlock) sh[i] = dev_tab[i];
    double *shm = sh + n_tab;
    for (int i = threadIdx.x; i < MS_COUNT * n; i += kBlock) shm[i] = mini_shell[i];
    __syncthreads();
    const double *fheat = sh + SH_COUNT * n, *fion = fheat + C21CM_X_INT_NXHII * n,
                 *flya = fion + C21CM_X_INT_NXHII * n;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        float xHII_call = prev_xe[i];  // :1499-1514
        if (xHII_call > kXHII[C21CM_X_INT_NXHII - 1] * 0.999)
            xHII_call = (float)(kXHII[C21CM_X_INT_NXHII - 1] * 0.999);
        else if (xHII_call < kXHII[0])
            xHII_call = (float)(1.001 * kXHII[0]);
        int mm = C21CM_X_INT_NXHII - 1;
        while (xHII_call < kXHII[mm]) mm--;
        const float inv_diff = (float)(1. / (kXHII[mm + 1] - kXHII[mm]));
        const double ival = (double)((xHII_call - kXHII[mm]) * inv_diff);
        CellSums q{0., 0., 0., 0., 0., 0.};
        double lw = 0.;
        for (int R = n; R--;) {
            const double z_edge = sh[SH_ZEDGE * n + R], xray_R = sh[SH_XRAY_R * n + R];
            const double curr_dens = (double)delNL0[(size_t)R * ntot + i] * sh[SH_GROWTH * n + R];
            const double tab_min = sh[SH_TABMIN * n + R], tab_width = sh[SH_TABWIDTH * n + R];
            const double fcoll = exp_f32acc(table_1d(curr_dens, tab_min, tab_width,
                                                     sh[SH_TABINVW * n + R],
                                                     tables + (size_t)R * C21CM_NDELTA_TABLE));
            const float sfrd = (float)((1. + curr_dens) * fcoll);
            const double sfr_term = (double)sfrd * z_edge * sh[SH_AVGFIX * n + R] * a.sfr_scale;
            const double fcoll_mini =
                exp(table_2d(curr_dens, (double)mcrit[(size_t)R * ntot + i], tab_min, tab_width,
                             ms.mt_min, ms.mt_width,
                             tables2 + (size_t)R * C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE));
            const float sfrd_mini = (float)((1. + curr_dens) * fcoll_mini);
            const double sfr_term_mini =
                (double)sfrd_mini * z_edge * shm[MS_AVGFIX * n + R] * ms.sfr_scale_mini;
            double xray_sfr = sfr_term * a.xray_scale * xray_R;
            xray_sfr += sfr_term_mini * ms.xray_scale_mini * xray_R;
            lw += sfr_term * shm[MS_LW * n + R] + sfr_term_mini * shm[MS_LW_MINI * n + R];
            const int lo = mm * n + R, hi = lo + n;
            if (a.use_xray_heating) q.heat += xray_sfr * ((fheat[hi] - fheat[lo]) * ival + fheat[lo]);
            q.ion += xray_sfr * ((fion[hi] - fion[lo]) * ival + fion[lo]);
            q.lya += xray_sfr * ((flya[hi] - flya[lo]) * ival + flya[lo]);
            q.starlya += sfr_term * sh[SH_STARLYA * n + R] + sfr_term_mini * shm[MS_STARLYA * n + R];
            if (a.use_lya_heating) {
                q.cont += sfr_term * sh[SH_CONT * n + R] + sfr_term_mini * shm[MS_CONT * n + R];
                q.inj += sfr_term * sh[SH_INJ * n + R] + sfr_term_mini * shm[MS_INJ * n + R];
            }
        }
        sums[i] = q.heat;
        sums[ntot + i] = q.ion;
        sums[2 * ntot + i] = q.lya;
        sums[3 * ntot + i] = q.starlya;
        if (a.use_lya_heating) {
            sums[4 * ntot + i] = q.cont;
            sums[5 * ntot + i] = q.inj;
        }
        J_21_LW[i] = (float)(lw * ms.lw_scale);  // :1843-1845
    }
}

// Lagrangian source grids with USE_MINI_HALOS (:1657-1701): filtered_xray already holds both
// populations; the molecularly cooled star formation adds its Lyman-alpha terms, and the
// Lyman-Werner sums read the straight-line copies when those exist (LYA_MULTIPLE_SCATTERING)
__global__ void __launch_bounds__(kBlock)
ts_accumulate_grids_mini_kernel(c21hip_ts_args a, double lw_scale,
                                const float *__restrict__ prev_xe, const float *__restrict__ sfr,
                                const float *__restrict__ xray, const float *__restrict__ sfr_mini,
                                const float *__restrict__ sfr_lw,
                                const float *__restrict__ sfr_mini_lw,
                                const double *__restrict__ dev_tab,
                                const double *__restrict__ mini_shell, double *__restrict__ sums,
                                float *__restrict__ J_21_LW, size_t ntot) {
    extern __shared__ double sh[];
    const int n = a.n_step;
    const int n_tab = (SH_COUNT + 3 * C21CM_X_INT_NXHII) * n;
    for (int i = threadIdx.x; i < n_tab; i += kBlock) sh[i] = dev_tab[i];
    double *shm = sh + n_tab;
    for (int i = threadIdx.x; i < MS_COUNT * n; i += kBlock) shm[i] = mini_shell[i];
    __syncthreads();
    const double *fheat = sh + SH_COUNT * n, *fion = fheat + C21CM_X_INT_NXHII * n,
                 *flya = fion + C21CM_X_INT_NXHII * n;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        float xHII_call = prev_xe[i];
        if (xHII_call > kXHII[C21CM_X_INT_NXHII - 1] * 0.999)
            xHII_call = (float)(kXHII[C21CM_X_INT_NXHII - 1] * 0.999);
        else if (xHII_call < kXHII[0])
            xHII_call = (float)(1.001 * kXHII[0]);
        int mm = C21CM_X_INT_NXHII - 1;
        while (xHII_call < kXHII[mm]) mm--;
        const float inv_diff = (float)(1. / (kXHII[mm + 1] - kXHII[mm]));
        const double ival = (double)((xHII_call - kXHII[mm]) * inv_diff);
        CellSums q{0., 0., 0., 0., 0., 0.};
        double lw = 0.;
        for (int R = n; R--;) {
            const size_t o = (size_t)R * ntot + i;
            const double z_edge = sh[SH_ZEDGE * n + R];
            const double sfr_term = (double)sfr[o] * z_edge;
            const double xray_sfr = (double)xray[o] * z_edge * sh[SH_XRAY_R * n + R] * 1e38;
            const double sfr_term_mini = (double)sfr_mini[o] * z_edge;
            const double sfr_term_lw = sfr_lw ? (double)sfr_lw[o] * z_edge : sfr_term;
            const double sfr_term_mini_lw = sfr_lw ? (double)sfr_mini_lw[o] * z_edge : sfr_term_mini;
            lw += sfr_term_lw * shm[MS_LW * n + R] + sfr_term_mini_lw * shm[MS_LW_MINI * n + R];
            const int lo = mm * n + R, hi = lo + n;
            if (a.use_xray_heating) q.heat += xray_sfr * ((fheat[hi] - fheat[lo]) * ival + fheat[lo]);
            q.ion += xray_sfr * ((fion[hi] - fion[lo]) * ival + fion[lo]);
            q.lya += xray_sfr * ((flya[hi] - flya[lo]) * ival + flya[lo]);
            q.starlya += sfr_term * sh[SH_STARLYA * n + R] + sfr_term_mini * shm[MS_STARLYA * n + R];
            if (a.use_lya_heating) {
                q.cont += sfr_term * sh[SH_CONT * n + R] + sfr_term_mini * shm[MS_CONT * n + R];
                q.inj += sfr_term * sh[SH_INJ * n + R] + sfr_term_mini * shm[MS_INJ * n + R];
            }
        }
        sums[i] = q.heat;
        sums[ntot + i] = q.ion;
        sums[2 * ntot + i] = q.lya;
        sums[3 * ntot + i] = q.starlya;
        if (a.use_lya_heating) {
            sums[4 * ntot + i] = q.cont;
            sums[5 * ntot + i] = q.inj;
        }
        J_21_LW[i] = (float)(lw * lw_scale);
    }
}

// prepare_filter_boxes with USE_MINI_HALOS (:535-565)
__global__ void __launch_bounds__(kBlock)
ts_mcrit_kernel(const float *__restrict__ J_21_LW, const float *__restrict__ vcb, float vcb_const,
                float z, double A_LW, double BETA_LW, double A_VCB, double BETA_VCB,
                double sigma_vcb, double m_turn, float *__restrict__ out, size_t ntot) {
    const double mcrit_noLW = 3.314e7 * pow(1. + (double)z, -1.5);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot;
         i += (size_t)gridDim.x * kBlock) {
        const float v = vcb ? vcb[i] : vcb_const;
        const double f_LW = 1.0 + A_LW * pow((double)J_21_LW[i], BETA_LW);
        const double f_vcb = pow(1.0 + A_VCB * (double)v / sigma_vcb, BETA_VCB);
        out[i] = (float)log10(fmax(mcrit_noLW * f_LW * f_vcb, m_turn));
    }
}
}  // namespace

extern "C" int c21hip_ts_mcrit_grid(const float *J_21_LW, const float *vcb, double vcb_const,
                                    double redshift, double A_LW, double BETA_LW, double A_VCB,
                                    double BETA_VCB, double sigma_vcb, double m_turn, float *out,
                                    size_t ntot, void *stream) {
    hipLaunchKernelGGL(ts_mcrit_kernel, dim3(grid_for(ntot)), dim3(kBlock), 0, (hipStream_t)stream,
                       J_21_LW, vcb, (float)vcb_const, (float)redshift, A_LW, BETA_LW, A_VCB,
                       BETA_VCB, sigma_vcb, m_turn, out, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ts_sfrd_means_mini(const float *filtered_density, const float *filtered_mcrit,
                                         const float *tables2_dev, const double *dev_tab,
                                         double *mini_shell_dev, const double *mean_sfr_mini_dev,
                                         int n_step, size_t ntot, double mt_min, double mt_width,
                                         double *partials, double *ave_out_dev, void *stream) {
    int bx = grid_for(ntot);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(sfrd_sum_mini_kernel, dim3(bx, n_step), dim3(kBlock), 0, (hipStream_t)stream,
                       filtered_density, filtered_mcrit, tables2_dev, dev_tab, n_step, ntot, mt_min,
                       mt_width, partials);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(sfrd_finish_mini_kernel, dim3(n_step), dim3(kBlock), 0, (hipStream_t)stream,
                       partials, bx, mean_sfr_mini_dev, (double)ntot, n_step, mini_shell_dev,
                       ave_out_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ts_accumulate_mini(const c21hip_ts_args *a, double sfr_scale_mini,
                                         double xray_scale_mini, double mt_min, double mt_width,
                                         const float *prev_xe, const float *delNL0,
                                         const float *mcrit, const float *tables_dev,
                                         const float *tables2_dev, const double *dev_tab,
                                         const double *mini_shell_dev, double *sums_ws,
                                         float *J_21_LW, size_t ntot, void *stream) {
    const size_t lds =
        (c21hip_ts_table_doubles(a->n_step) + (size_t)MS_COUNT * a->n_step) * sizeof(double);
    if (lds > 64 * 1024) {
        c21hip_set_error("spin temperature: %d shells do not fit the table cache", a->n_step);
        return C21CM_VALUE_ERROR;
    }
    MiniScalars ms;
    ms.sfr_scale_mini = sfr_scale_mini;
    ms.xray_scale_mini = xray_scale_mini;
    ms.mt_min = mt_min;
    ms.mt_width = mt_width;
    ms.lw_scale = a->lya_star_prefactor * a->volunit_inv * a->h_p * 1e21;
    hipLaunchKernelGGL(ts_accumulate_mini_kernel, dim3(grid_for(ntot)), dim3(kBlock), lds,
                       (hipStream_t)stream, *a, ms, prev_xe, delNL0, mcrit, tables_dev, tables2_dev,
                       dev_tab, mini_shell_dev, sums_ws, J_21_LW, ntot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int c21hip_ts_accumulate_grids_mini(const c21hip_ts_args *a, const float *prev_xe,
                                               const float *sfr, const float *xray,
                                               const float *sfr_mini, const float *sfr_lw,
                                               const float *sfr_mini_lw, const double *dev_tab,
                                               const double *mini_shell_dev, double *sums_ws,
                                               float *J_21_LW, size_t ntot, void *stream) {
    const size_t lds =
        (c21hip_ts_table_doubles(a->n_step) + (size_t)MS_COUNT * a->n_step) * sizeof(double);
    if (lds > 64 * 1024) {
        c21hip_set_error("spin temperature: %d shells do not fit the table cache", a->n_step);
        return C21CM_VALUE_ERROR;
    }
    const double lw_scale = a->lya_star_prefactor * a->volunit_inv * a->h_p * 1e21;
    hipLaunchKernelGGL(ts_accumulate_grids_mini_kernel, dim3(grid_for(ntot)), dim3(kBlock), lds,
                       (hipStream_t)stream, *a, lw_scale, prev_xe, sfr, xray, sfr_mini, sfr_lw,
                       sfr_mini_lw, dev_tab, mini_shell_dev, sums_ws, J_21_LW, ntot);
    LAUNCH_CHECK();
    return 0;
}

// ---- sharded shell sums: slabs of the partial sums to / from the exchange buffers ---------------
// pack: out[p * slot + k * maxlen + i] = sums[k * ntot + c0(p) + i] for every peer p != rank
// (slot = c21hip_ts_slot_elems elements); combine: the complete sums of this rank's slab, ranks added in
// rank order (own partial from `sums`, the others from the receive buffer, same layout as pack).
// As floats (C21CM_TS_SHARD_EXCHANGE=f32) the rows travel SCALED: the sums are physical rates of 1e27 ..
// 1e49 (and 1e-12) -- two of the four rows do not fit a float at all, which the first run of this exchange
// with two real ranks showed (round 6; every float came out inf) -- so each rank divides row k by
// 2^e(k), e(k) = exponent of the largest |value| of ITS partial row (exact), and the sixteen trailing
// elements of every slot carry the e(k) to the receiver, which multiplies back in double.
namespace {
enum { kTsExpTail = 16 };  // floats at the end of a float slot: the row exponents as int32
__device__ __forceinline__ size_t slab_begin(size_t ntot, int world, int r) {
    return (ntot / 4 * (size_t)r / (size_t)world) * 4;  // multiples of 4 cells; slab world ends at ntot
}
__device__ __forceinline__ int row_exponent(unsigned long long absmax_bits) {
    const int e = (int)((absmax_bits >> 52) & 0x7ffull);
    return e == 0 ? 0 : e - 1023;  // (zero / subnormal rows: no scaling)
}
__global__ void __launch_bounds__(kBlock)
ts_row_absmax_kernel(const double *__restrict__ sums, size_t ntot, unsigned long long *__restrict__ mx) {
    const int k = blockIdx.y;
    unsigned long long m = 0;  // |x| as bits: ordered like the values for finite non-negative doubles
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ntot; i += (size_t)gridDim.x * kBlock) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(sums[(size_t)k * ntot + i]));
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(mx + k, m);
}
template <typename T>
__global__ void __launch_bounds__(kBlock)
ts_pack_slabs_kernel(const double *__restrict__ sums, size_t ntot, int world, int rank, int rows,
                     size_t maxlen, size_t slot, const unsigned long long *__restrict__ mx, T *__restrict__ out) {
    const int p = blockIdx.y;  // peer slot: ranks other than `rank` in ascending order
    const int peer = p < rank ? p : p + 1;
    const size_t c0 = slab_begin(ntot, world, peer);
    const size_t len = (peer + 1 == world ? ntot : slab_begin(ntot, world, peer + 1)) - c0;
    for (int k = 0; k < rows; k++) {
        const int e = sizeof(T) == 4 ? row_exponent(mx[k]) : 0;
        if (sizeof(T) == 4 && blockIdx.x == 0 && threadIdx.x == 0)
            reinterpret_cast<int *>(out + (size_t)p * slot + (size_t)rows * maxlen)[k] = e;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (size_t)gridDim.x * kBlock) {
            const double v = sums[(size_t)k * ntot + c0 + i];
            out[(size_t)p * slot + (size_t)k * maxlen + i] = sizeof(T) == 4 ? (T)scalbn(v, -e) : (T)v;
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(kBlock)
ts_combine_slab_kernel(const double *__restrict__ sums, size_t ntot, int world, int rank, int rows,
                       size_t maxlen, size_t slot, const unsigned long long *__restrict__ mx,
                       const T *__restrict__ recv, double *__restrict__ out) {
    const size_t c0 = slab_begin(ntot, world, rank);
    const size_t len = (rank + 1 == world ? ntot : slab_begin(ntot, world, rank + 1)) - c0;
    for (int k = 0; k < rows; k++) {
        const int e_own = sizeof(T) == 4 ? row_exponent(mx[k]) : 0;
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (size_t)gridDim.x * kBlock) {
            double acc = 0.;
            for (int r = 0; r < world; r++) {
                if (r == rank) {
                    const double own = sums[(size_t)k * ntot + c0 + i];
                    // every partial in one precision
                    acc += sizeof(T) == 4 ? scalbn((double)(float)scalbn(own, -e_own), e_own) : own;
                } else {
                    const int p = r < rank ? r : r - 1;
                    const double v = (double)recv[(size_t)p * slot + (size_t)k * maxlen + i];
                    acc += sizeof(T) == 4
                               ? scalbn(v, reinterpret_cast<const int *>(recv + (size_t)p * slot + (size_t)rows * maxlen)[k])
                               : v;
                }
            }
            out[(size_t)k * len + i] = acc;
        }
    }
}
}  // namespace

// elements of one peer's slot of the exchange buffers (floats carry the row exponents at the end)
extern "C" size_t c21hip_ts_slot_elems(int rows, size_t maxlen, int as_float) {
    return (size_t)rows * maxlen + (as_float ? (size_t)kTsExpTail : 0);
}
extern "C" size_t c21hip_ts_slab_begin(size_t ntot, int world, int r) {
    return r >= world ? ntot : (ntot / 4 * (size_t)r / (size_t)world) * 4;
}
// `rowmax`: 8 unsigned long long of device scratch (the caller's; filled by the pack, read by the combine)
extern "C" int c21hip_ts_pack_slabs(const double *sums, size_t ntot, int world, int rank, int rows,
                                    size_t maxlen, int as_float, void *rowmax, void *out, void *stream) {
    if (rows > 8) return C21CM_VALUE_ERROR;
    const size_t slot = c21hip_ts_slot_elems(rows, maxlen, as_float);
    if (as_float) {
        if (hipMemsetAsync(rowmax, 0, 8 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
            return C21CM_MEMORY_ALLOC_ERROR;
        hipLaunchKernelGGL(ts_row_absmax_kernel, dim3(512, (unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream,
                           sums, ntot, (unsigned long long *)rowmax);
        LAUNCH_CHECK();
    }
    if (world < 2) return 0;
    const dim3 grid(512, (unsigned)(world - 1));
    if (as_float)
        hipLaunchKernelGGL((ts_pack_slabs_kernel<float>), grid, dim3(kBlock), 0, (hipStream_t)stream, sums,
                           ntot, world, rank, rows, maxlen, slot, (const unsigned long long *)rowmax, (float *)out);
    else
        hipLaunchKernelGGL((ts_pack_slabs_kernel<double>), grid, dim3(kBlock), 0, (hipStream_t)stream, sums,
                           ntot, world, rank, rows, maxlen, slot, (const unsigned long long *)rowmax, (double *)out);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int c21hip_ts_combine_slab(const double *sums, size_t ntot, int world, int rank, int rows,
                                      size_t maxlen, int as_float, const void *rowmax, const void *recv,
                                      double *out, void *stream) {
    const size_t slot = c21hip_ts_slot_elems(rows, maxlen, as_float);
    if (as_float)
        hipLaunchKernelGGL((ts_combine_slab_kernel<float>), dim3(1024), dim3(kBlock), 0,
                           (hipStream_t)stream, sums, ntot, world, rank, rows, maxlen, slot,
                           (const unsigned long long *)rowmax, (const float *)recv, out);
    else
        hipLaunchKernelGGL((ts_combine_slab_kernel<double>), dim3(1024), dim3(kBlock), 0,
                           (hipStream_t)stream, sums, ntot, world, rank, rows, maxlen, slot,
                           (const unsigned long long *)rowmax, (const double *)recv, out);
    LAUNCH_CHECK();
    return 0;
}
