/*
 * cosmology.c -- host-side scalar cosmology needed by the drop-in Compute* entry points:
 * matter power spectrum, sigma(M), growth factor, mass/radius conversions, the mass
 * function integrals that normalise the excursion set, and the RECFAST temperature table.
 *
 * These are O(10-1000) scalar evaluations per Compute* call -- host work, as in the
 * reference -- whose RESULTS steer the device kernels (P(k) table for the mode sampler,
 * D(z) for the displacements, zeta/f_limit/sigma for the ionisation barrier).
 *
 * The reference evaluates the same published formulae with GSL quadrature
 * (src/py21cmfast/src/cosmology.c, hmf.c).  GSL is a third-party dependency that is absent
 * here, so the integrals use an own adaptive Gauss-Kronrod (7,15) rule; where the reference
 * asks QAG for rel. tolerance 1e-6 (sigma) or 1e-3 (mass-function integrals) this code
 * converges to 1e-8 / 1e-6, i.e. results agree within the reference's own tolerance but
 * are not bit-identical ("parity unpinned" for these scalars).
 *
 * Exported names are the ones py21cmfast's cfuncs layer binds
 * (src/py21cmfast/src/_functionprototypes_wrapper.h:137-141,83): init_ps, free_ps, dicke,
 * sigma_z0, dsigmasqdm_z0, power_in_k.
 */
#include "cosmology.h"
#include "c21cm_grid.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"

/* physical constants: src/py21cmfast/src/Constants.c:4-43 */
#define PC_G 6.6743e-8
#define PC_CM_PER_MPC 3.08567758e24
#define PC_MSUN 1.989e33
#define PC_T_CMB 2.7255
#define PC_M_P 1.67262192369e-24
#define PC_SIGMA_HI 6.3e-18
#define DELTA_C_SPH 1.686
#define FRACT_FLOAT_ERR 1e-7 /* Constants.h */
#define N_NU 1.0 /* heavy neutrino species in the EH99 fit (cosmology.c:22) */

static struct {
    double sound_horizon, alpha_nu, beta_c, omhh, f_nu, f_baryon, theta_cmb, sigma_norm;
    int ready;
    unsigned generation; /* bumped by every init_ps: derived tables key on it */
} cc;

/* ---------------------------------------------------------------- adaptive quadrature */
typedef double (*integrand_fn)(double x, void *ctx);
double c21_qag61(integrand_fn f, void *ctx, double a, double b, double epsrel, double *abserr, int *status); /* heating.c */

static const double gk_x[8] = {0.991455371120812639206854697526329, 0.949107912342758524526189684047851,
                               0.864864423359769072789712788640926, 0.741531185599394439863864773280788,
                               0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
                               0.207784955007898467600689403773245, 0.0};
static const double gk_wk[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
static const double gk_wg[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                                0.381830050505118944950369775488975, 0.417959183673469387755102040816327};

static double gk15(integrand_fn f, void *ctx, double a, double b, double *err) {
    const double c = 0.5 * (a + b), h = 0.5 * (b - a);
    const double fc = f(c, ctx);
    double rk = fc * gk_wk[7], rg = fc * gk_wg[3];
    for (int j = 0; j < 7; j++) {
        const double dx = h * gk_x[j];
        const double f1 = f(c - dx, ctx), f2 = f(c + dx, ctx);
        rk += gk_wk[j] * (f1 + f2);
        if (j & 1) rg += gk_wg[j / 2] * (f1 + f2);
    }
    *err = fabs((rk - rg) * h);
    return rk * h;
}

static double adapt(integrand_fn f, void *ctx, double a, double b, double whole, double err,
                    double abs_tol, int depth) {
    if (err <= abs_tol || depth > 40) return whole;
    const double c = 0.5 * (a + b);
    double e1, e2;
    const double left = gk15(f, ctx, a, c, &e1), right = gk15(f, ctx, c, b, &e2);
    return adapt(f, ctx, a, c, left, e1, 0.5 * abs_tol, depth + 1) +
           adapt(f, ctx, c, b, right, e2, 0.5 * abs_tol, depth + 1);
}

double c21_integrate(integrand_fn f, void *ctx, double a, double b, double rel_tol) {
    /* two sweeps: a coarse one fixes the scale of the absolute tolerance */
    const int panels = 16;
    double total = 0., errs[16], vals[16];
    for (int p = 0; p < panels; p++) {
        const double lo = a + (b - a) * p / panels, hi = a + (b - a) * (p + 1) / panels;
        vals[p] = gk15(f, ctx, lo, hi, &errs[p]);
        total += vals[p];
    }
    const double abs_tol = fabs(total) * rel_tol / panels + 1e-300;
    double result = 0.;
    for (int p = 0; p < panels; p++) {
        const double lo = a + (b - a) * p / panels, hi = a + (b - a) * (p + 1) / panels;
        result += adapt(f, ctx, lo, hi, vals[p], errs[p], abs_tol, 0);
    }
    return result;
}

/* ---------------------------------------------------------------- basic quantities */
double c21_hubble0(void) { return (double)(cosmo_params_global->hlittle * 3.2407e-18); }

/* critical density today in Msun / Mpc^3: Constants.h RHOcrit */
double c21_rhocrit(void) {
    const double Ho = c21_hubble0();
    return (3.0 * Ho * Ho / (8.0 * M_PI * PC_G)) * (PC_CM_PER_MPC * PC_CM_PER_MPC * PC_CM_PER_MPC) /
           PC_MSUN;
}

/* present-day baryon number density H + He, Constants.h No / He_No / N_b0 */
double c21_nb0(void) {
    const double Ho = c21_hubble0();
    const double rhocrit_cgs = 3.0 * Ho * Ho / (8.0 * M_PI * PC_G);
    const double no = rhocrit_cgs * cosmo_params_global->OMb * (1 - cosmo_params_global->Y_He) / PC_M_P;
    const double he = rhocrit_cgs * cosmo_params_global->OMb * cosmo_params_global->Y_He / (4.0 * PC_M_P);
    return no + he;
}

/* cosmology.c:593-616 */
double c21_MtoR(double M) {
    const double rho_m = cosmo_params_global->OMm * c21_rhocrit();
    if (matter_options_global->FILTER == C21CM_FILTER_GAUSSIAN)
        return pow(M / (pow(2 * M_PI, 1.5) * rho_m), 1.0 / 3.0);
    return pow(3 * M / (4 * M_PI * rho_m), 1.0 / 3.0);
}
double c21_RtoM(double R) {
    const double rho_m = cosmo_params_global->OMm * c21_rhocrit();
    if (matter_options_global->FILTER == C21CM_FILTER_GAUSSIAN) return pow(2 * M_PI, 1.5) * rho_m * pow(R, 3);
    return (4.0 / 3.0) * M_PI * pow(R, 3) * rho_m;
}

static double omega_mz(float z) {
    const CosmoParams *c = cosmo_params_global;
    return c->OMm * pow(1 + z, 3) /
           (c->OMm * pow(1 + z, 3) + c->OMl + c->OMr * pow(1 + z, 4) + c->OMk * pow(1 + z, 2));
}

/* Barkana & Loeb 2001 virial temperature -> mass; cosmology.c:642-648 */
double c21_TtoM(double z, double T, double mu) {
    const double d = omega_mz((float)z) - 1.0;
    const double deltac_nl = 18 * M_PI * M_PI + 82 * d - 39 * d * d; /* Bryan & Norman 1998 */
    return 7030.97 / (cosmo_params_global->hlittle) *
           sqrt(omega_mz((float)z) / (cosmo_params_global->OMm * deltac_nl)) *
           pow(T / (mu * (1 + z)), 1.5);
}

double c21_hubble(float z) {
    const CosmoParams *c = cosmo_params_global;
    return c21_hubble0() * sqrt(c->OMm * pow(1 + z, 3) + c->OMr * pow(1 + z, 4) + c->OMl);
}

/* Growth factor, Liddle et al. 1996 fit for flat LCDM(+radiation): cosmology.c:670-709 */
double dicke(double z) {
    const CosmoParams *c = cosmo_params_global;
    const double tiny = 1e-4;
    if (fabs(c->OMm - 1.0) < tiny) return 1.0 / (1.0 + z);
    if ((c->OMl > (-tiny)) && (fabs(c->OMl + c->OMm + c->OMr - 1.0) < 0.01) &&
        (fabs(c->wl + 1.0) < tiny)) {
        const double omz = c->OMm * pow(1 + z, 3) /
                           (c->OMl + c->OMm * pow(1 + z, 3) + c->OMr * pow(1 + z, 4));
        const double dz = 2.5 * omz / (1.0 / 70.0 + omz * (209 - omz) / 140.0 + pow(omz, 4.0 / 7.0));
        const double d0 =
            2.5 * c->OMm / (1.0 / 70.0 + c->OMm * (209 - c->OMm) / 140.0 + pow(c->OMm, 4.0 / 7.0));
        return dz / (d0 * (1.0 + z));
    }
    if ((c->OMtot < (1 + tiny)) && (fabs(c->OMl) < tiny)) { /* open, no lambda (Peebles p.53) */
        const double x0 = 1.0 / (c->OMm + 0.0) - 1.0;
        const double d0 = 1 + 3.0 / x0 + 3 * log(sqrt(1 + x0) - sqrt(x0)) * sqrt(1 + x0) / pow(x0, 1.5);
        const double x = fabs(1.0 / (c->OMm + 0.0) - 1.0) / (1 + z);
        const double dz = 1 + 3.0 / x + 3 * log(sqrt(1 + x) - sqrt(x)) * sqrt(1 + x) / pow(x, 1.5);
        return dz / d0;
    }
    c21hip_set_error("dicke: no growth function for this cosmology");
    return NAN;
}

/* dt/dz in seconds: cosmology.c:711-722 */
double c21_dtdz(float z) {
    const CosmoParams *c = cosmo_params_global;
    const double x = sqrt(c->OMl / c->OMm) * pow(1 + z, -3.0 / 2.0);
    const double dxdz = sqrt(c->OMl / c->OMm) * pow(1 + z, -5.0 / 2.0) * (-3.0 / 2.0);
    const double const1 = 2 * sqrt(1 + c->OMm / c->OMl) / (3.0 * c21_hubble0());
    const double numer = dxdz * (1 + x * pow(pow(x, 2) + 1, -0.5));
    const double denom = x + sqrt(pow(x, 2) + 1);
    return const1 * numer / denom;
}

/* dD/dt by the reference's one-sided difference with a float step: cosmology.c:725-731 */
double c21_ddickedt(double z) {
    const float dz = 1e-10;
    return (dicke(z + dz) - dicke(z)) / dz / c21_dtdz((float)z);
}

/* ---------------------------------------------------------------- power spectrum */
/* Eisenstein & Hu 1999 (ApJ 511, 5) fitting constants: cosmology.c:455-503 */
static void set_eh_parameters(void) {
    const double f_nu = cc.f_nu, f_b = cc.f_baryon, omhh = cc.omhh, th = cc.theta_cmb;
    const double obhh = cosmo_params_global->OMb * cosmo_params_global->hlittle * cosmo_params_global->hlittle;
    const double z_eq = 25000 * omhh * pow(th, -4) - 1.0;
    const double k_eq = 0.0746 * omhh / (th * th);
    double z_drag = 0.313 * pow(omhh, -0.419) * (1 + 0.607 * pow(omhh, 0.674));
    z_drag = 1 + z_drag * pow(obhh, 0.238 * pow(omhh, 0.223));
    z_drag *= 1291 * pow(omhh, 0.251) / (1 + 0.659 * pow(omhh, 0.828));
    const double y_d = (1 + z_eq) / (1.0 + z_drag);
    const double R_drag = 31.5 * obhh * pow(th, -4) * 1000 / (1.0 + z_drag);
    const double R_eq = 31.5 * obhh * pow(th, -4) * 1000 / (1.0 + z_eq);
    cc.sound_horizon = 2.0 / 3.0 / k_eq * sqrt(6.0 / R_eq) *
                       log((sqrt(1 + R_drag) + sqrt(R_drag + R_eq)) / (1.0 + sqrt(R_eq)));
    const double p_c = -(5 - sqrt(1 + 24 * (1 - f_nu - f_b))) / 4.0;
    const double p_cb = -(5 - sqrt(1 + 24 * (1 - f_nu))) / 4.0;
    const double f_c = 1 - f_nu - f_b, f_cb = 1 - f_nu, f_nub = f_nu + f_b;
    double a = (f_c / f_cb) * (2 * (p_c + p_cb) + 5) / (4 * p_cb + 5.0);
    a *= 1 - 0.553 * f_nub + 0.126 * pow(f_nub, 3);
    a /= 1 - 0.193 * sqrt(f_nu) + 0.169 * f_nu;
    a *= pow(1 + y_d, p_c - p_cb);
    a *= 1 + (p_cb - p_c) / 2.0 * (1.0 + 1.0 / (4.0 * p_c + 3.0) / (4.0 * p_cb + 7.0)) / (1.0 + y_d);
    cc.alpha_nu = a;
    cc.beta_c = 1.0 / (1.0 - 0.949 * f_nub);
}

/* cosmology.c:52-75 */
static double transfer_eh(double k) {
    const double q = k * pow(cc.theta_cmb, 2) / cc.omhh;
    const double sa = sqrt(cc.alpha_nu);
    const double gamma_eff = sa + (1.0 - sa) / (1.0 + pow(0.43 * k * cc.sound_horizon, 4));
    const double q_eff = q / gamma_eff;
    double tf = log(M_E + 1.84 * cc.beta_c * sa * q_eff);
    tf /= tf + pow(q_eff, 2) * (14.4 + 325.0 / (1.0 + 60.5 * pow(q_eff, 1.11)));
    const double q_nu = 3.92 * q / sqrt(cc.f_nu / N_NU);
    tf *= 1.0 + (1.2 * pow(cc.f_nu, 0.64) * pow(N_NU, 0.3 + 0.6 * cc.f_nu)) /
                    (pow(q_nu, -1.6) + pow(q_nu, 0.8));
    return tf;
}

/* the other analytic fits offered by POWER_SPECTRUM: cosmology.c:79-129 */
static double transfer_other(double k, int which) {
    const CosmoParams *c = cosmo_params_global;
    const double h = c->hlittle;
    if (which == C21CM_PS_BBKS) { /* Bardeen et al. 1986 + Sugiyama 1995 */
        const double gamma = c->OMm * h * exp(-(c->OMb) - (c->OMb / c->OMm));
        const double q = k / (h * gamma);
        return (log(1.0 + 2.34 * q) / (2.34 * q)) *
               pow(1.0 + 3.89 * q + pow(16.1 * q, 2) + pow(5.46 * q, 3) + pow(6.71 * q, 4), -0.25);
    }
    if (which == C21CM_PS_EFSTATHIOU) { /* Efstathiou et al. 1992 */
        const double gamma = c->OMm * h * h;
        const double aa = 6.4 / gamma, bb = 3.0 / gamma, ccc = 1.7 / gamma, nu = 1.13;
        return pow(1 + pow(aa * k + pow(bb * k, 1.5) + pow(ccc * k, 2), nu), -1. / nu);
    }
    if (which == C21CM_PS_PEEBLES) { /* Peebles 1980 */
        const double gamma = c->OMm * h * exp(-(c->OMb) - (c->OMb / c->OMm));
        return 1 + (8.0 / (h * gamma)) * k + (4.7 / pow(h * gamma, 2)) * k * k;
    }
    /* White / Davies et al. 1985 */
    const double gamma = c->OMm * h * h * exp(-(c->OMb) - (c->OMb / c->OMm));
    return 139.284 / (1 + (1.7 / gamma) * k + (9.0 / pow(gamma, 1.5)) * pow(k, 1.5) +
                      (1.0 / pow(gamma, 2)) * k * k);
}

/* POWER_SPECTRUM = CLASS: the tabulated z = 0 transfer functions of CosmoTables (density, and
 * the DM-baryon relative velocity at kinematic decoupling), natural cubic splines in k like
 * gsl_interp_cspline; above the last tabulated k the density one continues with the EH shape
 * and the velocity one with a log-log line through the last two points: cosmology.c:130-225 */
static struct {
    int n_d, n_v; /* 0: not set up */
    const double *k, *Td, *Tv;
    double *cd, *cv; /* spline c coefficients (y'' / 2) */
    double eh_ratio_at_kmax;
} cls;

static void class_spline_c(int n, const double *x, const double *y, double *c) {
    double *g = (double *)malloc(sizeof(double) * (size_t)n), *dg = (double *)malloc(sizeof(double) * (size_t)n),
           *of = (double *)malloc(sizeof(double) * (size_t)n);
    c[0] = c[n - 1] = 0.;
    const int m = n - 2;
    for (int i = 0; i < m; i++) {
        const double h0 = x[i + 1] - x[i], h1 = x[i + 2] - x[i + 1];
        of[i] = h1;
        dg[i] = 2.0 * (h1 + h0);
        g[i] = 3.0 * ((y[i + 2] - y[i + 1]) / h1 - (y[i + 1] - y[i]) / h0);
    }
    for (int i = 1; i < m; i++) {
        const double w = of[i - 1] / dg[i - 1];
        dg[i] -= w * of[i - 1];
        g[i] -= w * g[i - 1];
    }
    for (int i = m - 1; i >= 0; i--) c[i + 1] = (g[i] - (i + 1 < m ? of[i] * c[i + 2] : 0.)) / dg[i];
    free(g);
    free(dg);
    free(of);
}

static double class_spline_eval(int n, const double *x, const double *y, const double *c, double v) {
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (hi + lo) >> 1;
        if (x[mid] > v)
            hi = mid;
        else
            lo = mid;
    }
    const double dx = x[lo + 1] - x[lo], dy = y[lo + 1] - y[lo];
    const double b = dy / dx - dx * (c[lo + 1] + 2.0 * c[lo]) / 3.0;
    const double d = (c[lo + 1] - c[lo]) / (3.0 * dx);
    const double t = v - x[lo];
    return y[lo] + t * (b + t * (c[lo] + t * d));
}

static void class_free(void) {
    free(cls.cd);
    free(cls.cv);
    memset(&cls, 0, sizeof(cls));
}

static int class_setup(void) { /* transfer_function_CLASS(., 0, .) */
    class_free();
    const Table1D *td = cosmo_tables_global ? cosmo_tables_global->transfer_density : NULL;
    if (!td || td->size < 3 || !td->x_values || !td->y_values) {
        c21hip_set_error("POWER_SPECTRUM = CLASS needs CosmoTables.transfer_density (>= 3 points)");
        return C21CM_VALUE_ERROR;
    }
    cls.n_d = td->size;
    cls.k = td->x_values;
    cls.Td = td->y_values;
    cls.cd = (double *)malloc(sizeof(double) * (size_t)td->size);
    if (!cls.cd) return C21CM_MEMORY_ALLOC_ERROR;
    class_spline_c(td->size, cls.k, cls.Td, cls.cd);
    const double kmax = cls.k[td->size - 1];
    cls.eh_ratio_at_kmax = cls.Td[td->size - 1] / kmax / kmax / transfer_eh(kmax);
    if (matter_options_global->V_CB_MODEL == C21CM_VCB_FLUCTS) {
        const Table1D *tv = cosmo_tables_global->transfer_vcb;
        if (!tv || tv->size != td->size || !tv->y_values) {
            c21hip_set_error("V_CB_MODEL = FLUCTS needs CosmoTables.transfer_vcb on the k grid of "
                             "transfer_density");
            return C21CM_VALUE_ERROR;
        }
        cls.n_v = tv->size;
        cls.Tv = tv->y_values;
        cls.cv = (double *)malloc(sizeof(double) * (size_t)tv->size);
        if (!cls.cv) return C21CM_MEMORY_ALLOC_ERROR;
        class_spline_c(tv->size, cls.k, cls.Tv, cls.cv);
    }
    return 0;
}

/* flag_dv 0: density, 1: relative velocity */
static double transfer_class(double k, int flag_dv) {
    if (flag_dv == 0) {
        if (!cls.n_d) return NAN;
        if (k > cls.k[cls.n_d - 1]) return cls.eh_ratio_at_kmax * transfer_eh(k) * k * k;
        return class_spline_eval(cls.n_d, cls.k, cls.Td, cls.cd, k);
    }
    if (!cls.n_v) return NAN;
    const int n = cls.n_v;
    if (k > cls.k[n - 1])
        return exp(log(cls.Tv[n - 1]) + (log(cls.Tv[n - 1]) - log(cls.Tv[n - 2])) /
                                            (log(cls.k[n - 1]) - log(cls.k[n - 2])) *
                                            (log(k) - log(cls.k[n - 1])));
    return class_spline_eval(n, cls.k, cls.Tv, cls.cv, k);
}

/* z = 0 linear matter power spectrum in Mpc^3: cosmology.c:278-308 */
double power_in_k(double k) {
    if (k == 0.) return 0.;
    const int which = matter_options_global->POWER_SPECTRUM;
    double T;
    if (which == C21CM_PS_CLASS) {
        T = transfer_class(k, 0);
    } else {
        T = (which == C21CM_PS_EH) ? transfer_eh(k) : transfer_other(k, which);
        T *= k * k; /* analytic fits tend to 1 as k -> 0; convert to the CLASS convention */
    }
    const double primordial =
        cosmo_tables_global->ps_norm * pow(k / 0.05, cosmo_params_global->POWER_INDEX - 1.);
    double p = cc.sigma_norm * primordial * T * T / pow(k, 3);
    if (which == C21CM_PS_CLASS && matter_options_global->V_CB_MODEL != C21CM_VCB_NONE)
        /* average suppression by the streaming velocity (:295-300; A 0.24, k_p 300 / Mpc, sigma 0.9) */
        p *= 1.0 - 0.24 * exp(-pow(log(k / 300.0), 2.0) / (2.0 * 0.9 * 0.9));
    return p;
}

/* power spectrum of the DM-baryon relative velocity at kinematic decoupling: cosmology.c:310-333 */
double power_in_vcb(double k) {
    if (matter_options_global->POWER_SPECTRUM != C21CM_PS_CLASS) return NAN;
    if (k == 0.) return 0.;
    const double T = transfer_class(k, 1);
    const double primordial =
        cosmo_tables_global->ps_norm * pow(k / 0.05, cosmo_params_global->POWER_INDEX - 1.);
    return cc.sigma_norm * primordial * T * T / pow(k, 3);
}

/* window of FILTER for sigma(M): filtering.c:18-46 */
static double sigma_window(double kR, int filter) {
    if (filter == C21CM_FILTER_GAUSSIAN) return exp(-0.643 * 0.643 * kR * kR / 2.);
    if (filter == C21CM_FILTER_SHARPK) return (kR * 0.413566994 > 1) ? 0. : 1.;
    if (kR < 1e-4) return 1 - kR * kR / 10;
    return 3.0 * pow(kR, -3) * (sin(kR) - cos(kR) * kR);
}

struct sigma_ctx {
    double R;
    int filter;
};

/* integrand in ln k: k^3 P W^2 / (2 pi^2), cosmology.c:354-367 */
static double dsigma_dlnk(double lnk, void *ctx) {
    const struct sigma_ctx *s = (const struct sigma_ctx *)ctx;
    const double k = exp(lnk);
    const double w = sigma_window(k * s->R, s->filter);
    return k * k * k * power_in_k(k) * w * w / (2.0 * M_PI * M_PI);
}

/* cosmology.c:369-408; the reference integrates k in [1e-99/R, 350/R] */
double sigma_z0(double M) {
    struct sigma_ctx s = {c21_MtoR(M), matter_options_global->FILTER};
    const double res = c21_integrate(dsigma_dlnk, &s, log(1e-7 / s.R), log(350.0 / s.R), 1e-9);
    return sqrt(res);
}

/* d(W^2)/dM: filtering.c:49-78 */
static double dw2dm(double k, double R, int filter) {
    const double kR = k * R;
    const double rho_m = cosmo_params_global->OMm * c21_rhocrit();
    double w, dwdr, drdm;
    if (filter == C21CM_FILTER_GAUSSIAN) {
        w = exp(-kR * kR / 2.0);
        dwdr = -k * kR * w;
        drdm = 1.0 / (pow(2 * M_PI, 1.5) * rho_m * 3 * R * R);
    } else {
        w = (kR < 1.0e-4) ? 1.0 : 3.0 * (sin(kR) / pow(kR, 3) - cos(kR) / pow(kR, 2));
        dwdr = (kR < 1.0e-10)
                   ? 0
                   : 9 * cos(kR) * k / pow(kR, 3) + 3 * sin(kR) * (1 - 3 / (kR * kR)) / (kR * R);
        drdm = 1.0 / (4.0 * M_PI * rho_m * R * R);
    }
    return 2 * w * dwdr * drdm;
}

static double dsigmasq_dlnk(double lnk, void *ctx) {
    const struct sigma_ctx *s = (const struct sigma_ctx *)ctx;
    const double k = exp(lnk);
    return k * k * k * power_in_k(k) * dw2dm(k, s->R, s->filter) / (2.0 * M_PI * M_PI);
}

/* cosmology.c:421-453 */
double dsigmasqdm_z0(double M) {
    struct sigma_ctx s = {c21_MtoR(M), matter_options_global->FILTER};
    return c21_integrate(dsigmasq_dlnk, &s, log(1e-7 / s.R), log(350.0 / s.R), 1e-8);
}

/* cosmology.c:507-558 */
void init_ps(void) {
    const CosmoParams *c = cosmo_params_global;
    cc.ready = 0;
    cc.omhh = c->OMm * c->hlittle * c->hlittle;
    cc.theta_cmb = PC_T_CMB / 2.7;
    cc.f_nu = fmax(c->OMn / c->OMm, 1e-10);
    cc.f_baryon = fmax(c->OMb / c->OMm, 1e-10);
    set_eh_parameters();
    if (matter_options_global->POWER_SPECTRUM == C21CM_PS_CLASS) {
        if (class_setup()) return; /* cc.ready stays 0: the Compute* entry points report the error */
    } else {
        class_free();
    }
    if (cosmo_tables_global->USE_SIGMA_8) {
        cc.sigma_norm = 1;
        const double R8 = 8.0 / c->hlittle;
        cc.sigma_norm = pow(cosmo_tables_global->ps_norm / sigma_z0(c21_RtoM(R8)), 2);
    } else {
        cc.sigma_norm = 2.0 * M_PI * M_PI;
    }
    cc.generation++;
    cc.ready = 1;
}

void free_ps(void) { cc.ready = 0; }

int c21_ps_ready(void) { return cc.ready; }

/* ---------------------------------------------------------------- sigma(M) table
 * The mass-function integrals need sigma and d sigma^2/dM at hundreds of masses, each a
 * quadrature over k.  Like the reference (interp_tables.c:1135-1170, Sigma_InterpTable) they
 * are tabulated once per power-spectrum normalisation on a ln M grid; unlike the reference's
 * linear float table the lookup is a natural cubic spline of ln sigma, ln(-d sigma^2/dM). */
#define SIG_N 416
#define SIG_LNM_MIN 6.9   /* ~1e3 Msun  */
#define SIG_LNM_MAX 48.5  /* ~1e21 Msun: RtoM of the outermost spin-temperature shells (R_MAX_TS = 500 Mpc -> 2e19) */
static struct {
    int ready;
    double norm_tag; /* sigma_norm the table was built for */
    unsigned generation; /* init_ps call it was built after (shape parameters may differ at equal norm) */
    int filter, ps;
    double lnM[SIG_N], lns[SIG_N], lnd[SIG_N], lns2[SIG_N], lnd2[SIG_N];
} st;

static void spline_setup(int n, const double *x, const double *y, double *y2);
static double spline_eval(int n, const double *x, const double *y, const double *y2, double v);

static void sigma_table_build(void) {
    if (st.ready && st.generation == cc.generation && st.norm_tag == cc.sigma_norm && st.filter == matter_options_global->FILTER &&
        st.ps == matter_options_global->POWER_SPECTRUM)
        return;
    const int n_thr = simulation_options_global && simulation_options_global->N_THREADS > 1
                          ? simulation_options_global->N_THREADS : 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_thr) /* interp_tables.c:1147-1156 */
    for (int i = 0; i < SIG_N; i++) {
        st.lnM[i] = SIG_LNM_MIN + (SIG_LNM_MAX - SIG_LNM_MIN) * i / (SIG_N - 1.0);
        const double M = exp(st.lnM[i]);
        st.lns[i] = log(sigma_z0(M));
        st.lnd[i] = log(-dsigmasqdm_z0(M));
    }
    spline_setup(SIG_N, st.lnM, st.lns, st.lns2);
    spline_setup(SIG_N, st.lnM, st.lnd, st.lnd2);
    st.norm_tag = cc.sigma_norm;
    st.generation = cc.generation;
    st.filter = matter_options_global->FILTER;
    st.ps = matter_options_global->POWER_SPECTRUM;
    st.ready = 1;
}

/* The reference's own table, restated (C21CM_SIGMA_TABLE=reference): 300 FLOAT entries of
 * sigma_z0 and log10(-dsigmasqdm_z0) at float masses on a uniform ln M grid between the float
 * arguments 5e2 and 1e20 of _global_initialization.py:131-134, looked up by LINEAR interpolation
 * (interp_tables.c:32,1135-1180; interpolation.c:123-131).  Its interpolation error (up to a few
 * 1e-4 of sigma between nodes) is part of every mass-function integral of the reference, so the
 * x_HI / dT_b fields of the reference's fixtures are reproduced more closely with it than with the
 * converged spline above.  The entries are this file's converged quadratures rounded to float; the
 * reference's come from one QAG(61-point) call at epsrel 1e-6, i.e. agree to better than the float
 * rounding except for an occasional last-bit flip. */
#define REF_SIG_N 300
static struct {
    int ready;
    double norm_tag;
    unsigned generation;
    int filter, ps;
    double x_min, x_width;
    float sig[REF_SIG_N], l10d[REF_SIG_N];
} rt;

/* C21CM_HOST_MODE=reference: the host quadratures as the reference STOPS them, not converged --
 * this float sigma(M) table, and gsl_integration_qag(61-point rule, epsrel 1e-3) for the
 * unconditional mass-function integrals (hmf.c:612-655,896-900: Fcoll_General, Nion_General,
 * Nion_General_MINI all go through IntegratedNdM with method 0).  The normalisation of the
 * excursion set (set_mean_fcoll, IonisationBox.c:468-529) inherits that quadrature's error.
 * C21CM_HOST_MODE=converged: every quadrature to 1e-6 or better. */
int c21_host_reference_mode(void) {
    const char *e = getenv("C21CM_HOST_MODE");
    if (e && e[0] == 'r') return 1;
    if (e && e[0] == 'c') return 0;
    return 0;
}
static int sigma_reference_mode(void) {
    const char *e = getenv("C21CM_SIGMA_TABLE"); /* finer switch for diagnostics: r / c */
    if (e && (e[0] == 'r' || e[0] == 'c')) return e[0] == 'r';
    return c21_host_reference_mode();
}
static int mf_quad_reference_mode(void) {
    const char *e = getenv("C21CM_MF_QUAD");
    if (e && (e[0] == 'r' || e[0] == 'c')) return e[0] == 'r';
    return c21_host_reference_mode();
}

static void sigma_reference_build(void) {
    if (rt.ready && rt.generation == cc.generation && rt.norm_tag == cc.sigma_norm &&
        rt.filter == matter_options_global->FILTER && rt.ps == matter_options_global->POWER_SPECTRUM)
        return;
    const float M_min = 5e2f, M_max = 1e20f;
    rt.x_min = log(M_min);
    rt.x_width = (log(M_max) - log(M_min)) / (REF_SIG_N - 1.);
    const int n_thr = simulation_options_global && simulation_options_global->N_THREADS > 1
                          ? simulation_options_global->N_THREADS : 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_thr)
    for (int i = 0; i < REF_SIG_N; i++) {
        const float Mass = (float)exp(rt.x_min + i * rt.x_width);
        rt.sig[i] = (float)sigma_z0(Mass);
        rt.l10d[i] = (float)log10(-dsigmasqdm_z0(Mass));
    }
    rt.norm_tag = cc.sigma_norm;
    rt.generation = cc.generation;
    rt.filter = matter_options_global->FILTER;
    rt.ps = matter_options_global->POWER_SPECTRUM;
    rt.ready = 1;
}

/* EvaluateRGTable1D_f (interpolation.c:123-131) */
static double reference_lookup(const float *y, double x) {
    int idx = (int)floor((x - rt.x_min) / rt.x_width);
    if (idx > REF_SIG_N - 2) idx = REF_SIG_N - 2; /* the quotient can round up to N - 1 at the upper edge */
    if (idx < 0) idx = 0;
    const double table_val = rt.x_min + rt.x_width * (float)idx;
    const double t = (x - table_val) / rt.x_width;
    return y[idx] * (1 - t) + y[idx + 1] * t;
}

static int reference_in_range(double lnM) {
    return lnM >= rt.x_min && lnM < rt.x_min + rt.x_width * (REF_SIG_N - 1);
}

double c21_sigma_fast(double M) {
    const double lnM = log(M);
    if (sigma_reference_mode()) {
        sigma_reference_build();
        if (reference_in_range(lnM)) return reference_lookup(rt.sig, lnM);
        return sigma_z0(M); /* (the reference reads past its table there) */
    }
    if (lnM < SIG_LNM_MIN || lnM > SIG_LNM_MAX) return sigma_z0(M);
    sigma_table_build();
    return exp(spline_eval(SIG_N, st.lnM, st.lns, st.lns2, lnM));
}

static double dsigmasqdm_fast(double M) {
    const double lnM = log(M);
    if (sigma_reference_mode()) {
        sigma_reference_build();
        if (reference_in_range(lnM)) return -pow(10., reference_lookup(rt.l10d, lnM));
        return dsigmasqdm_z0(M);
    }
    if (lnM < SIG_LNM_MIN || lnM > SIG_LNM_MAX) return dsigmasqdm_z0(M);
    sigma_table_build();
    return -exp(spline_eval(SIG_N, st.lnM, st.lnd, st.lnd2, lnM));
}

/* ---------------------------------------------------------------- mass function integrals */
#define SHETH_a 0.73 /* hmf.c:58-60 (Jenkins et al. 2001 values) */
#define SHETH_p 0.175
#define SHETH_A 0.353

/* (1/rho_m) dn/dlnM, i.e. f(nu) |dln sigma / dM|: hmf.c:301-313 (ST), PS analogue */
static double unconditional_mf(double growthf, double lnM, int hmf) {
    const double M = exp(lnM);
    double sigma = c21_sigma_fast(M) * growthf;
    const double dsigmadm = dsigmasqdm_fast(M) * (growthf * growthf / (2. * sigma));
    if (hmf == C21CM_HMF_PS) {
        return -(dsigmadm / sigma) * sqrt(2. / M_PI) * (DELTA_C_SPH / sigma) *
               exp(-(DELTA_C_SPH * DELTA_C_SPH) / (2 * sigma * sigma));
    }
    const double nuhat = sqrt(SHETH_a) * DELTA_C_SPH / sigma;
    return -(dsigmadm / sigma) * sqrt(2. / M_PI) * SHETH_A * (1 + pow(nuhat, -2 * SHETH_p)) * nuhat *
           exp(-nuhat * nuhat / 2.0);
}

struct mf_ctx {
    double growthf;
    int hmf;
    int kind; /* 0: M * mf (collapsed fraction), 1: nion_fraction * mf */
    double ln_fstar_norm, alpha_star, ln_Mlim_star;
    double ln_fesc_norm, alpha_esc, ln_Mlim_esc;
    double Mturn;
};

/* scaling_relations.c:211-231 */
static double log_pl_limit(double lnM, double ln_norm, double alpha, double ln_pivot, double ln_limit) {
    if ((alpha > 0. && lnM > ln_limit) || (alpha < 0. && lnM < ln_limit)) return -ln_norm;
    return alpha * (lnM - ln_pivot);
}

static double mf_integrand(double lnM, void *ctx) {
    const struct mf_ctx *p = (const struct mf_ctx *)ctx;
    const double mf = unconditional_mf(p->growthf, lnM, p->hmf);
    if (p->kind == 0) return exp(lnM) * mf; /* hmf.c:591-593 */
    /* hmf.c:462-468 */
    const double Fstar = log_pl_limit(lnM, p->ln_fstar_norm, p->alpha_star, 10 * M_LN10, p->ln_Mlim_star);
    const double Fesc = log_pl_limit(lnM, p->ln_fesc_norm, p->alpha_esc, 10 * M_LN10, p->ln_Mlim_esc);
    return exp(Fstar + Fesc - p->Mturn / exp(lnM) + lnM) * mf;
}

/* IntegratedNdM with method 0 (hmf.c:612-655): converged, or stopped as the reference stops it */
static double mf_integral(integrand_fn f, void *ctx, double lnM_min, double lnM_max) {
    if (mf_quad_reference_mode()) {
        (void)c21_sigma_fast(1e10); /* the table is built outside the integrand */
        return c21_qag61(f, ctx, lnM_min, lnM_max, 1e-3, NULL, NULL);
    }
    return c21_integrate(f, ctx, lnM_min, lnM_max, 1e-6);
}

static int supported_hmf(void) {
    const int h = matter_options_global->HMF;
    return h == C21CM_HMF_PS || h == C21CM_HMF_ST;
}

/* hmf.c:1187-1241: erfcc (a 1.2e-7 fit) and the conditional collapsed fraction in terms of
 * sigmas; float arguments and float return of erfcc as upstream.  NaN for sig_large > sig_small
 * (the reference throws ValueError there). */
static float erfcc_f(float x) {
    const double q = fabs(x), t = 1.0 / (1.0 + 0.5 * q);
    const double ans =
        t * exp(-q * q - 1.2655122 +
                t * (1.0000237 +
                     t * (0.374092 +
                          t * (0.0967842 +
                               t * (-0.1862881 +
                                    t * (0.2788681 +
                                         t * (-1.13520398 +
                                              t * (1.4885159 +
                                                   t * (-0.82215223 + t * 0.17087277)))))))));
    return x >= 0.0 ? ans : 2.0 - ans;
}

double c21_FgtrM_bias_fast(float growthf, float del_bias, float sig_small, float sig_large) {
    if (sig_large > sig_small) return NAN;
    if (sig_large == sig_small) return 0.;
    const double sig = sqrt(sig_small * sig_small - sig_large * sig_large);
    const double del = (DELTA_C_SPH - del_bias) / growthf;
    const double x = del / (sqrt(2) * sig);
    return x < 0 ? 1.0 : erfcc_f(x);
}

/* hmf.c:1253-1264: central difference over dz = 0.001, float arguments and float result */
float c21_dfcoll_dz(float z, float sigma_min, float del_bias, float sig_bias) {
    const double dz = 0.001, z1 = z + dz, z2 = z - dz;
    const double fc1 = c21_FgtrM_bias_fast(dicke(z1), del_bias, sigma_min, sig_bias);
    const double fc2 = c21_FgtrM_bias_fast(dicke(z2), del_bias, sigma_min, sig_bias);
    return (fc1 - fc2) / (2.0 * dz);
}

/* hmf.c:945-953 */
double c21_Fcoll_General(double z, double lnM_min, double lnM_max) {
    if (!supported_hmf()) return NAN;
    struct mf_ctx p;
    memset(&p, 0, sizeof(p));
    p.growthf = dicke(z);
    p.hmf = matter_options_global->HMF;
    p.kind = 0;
    return mf_integral(mf_integrand, &p, lnM_min, lnM_max);
}

/* hmf.c:955-971 */
double c21_Nion_General(double z, double lnM_min, double lnM_max, double Mturn,
                        const c21_scaling_consts *sc) {
    if (!supported_hmf()) return NAN;
    struct mf_ctx p;
    memset(&p, 0, sizeof(p));
    p.growthf = dicke(z);
    p.hmf = matter_options_global->HMF;
    p.kind = 1;
    p.ln_fstar_norm = log(sc->fstar_10);
    p.alpha_star = sc->alpha_star;
    p.ln_Mlim_star = log(sc->Mlim_Fstar);
    p.ln_fesc_norm = log(sc->fesc_10);
    p.alpha_esc = sc->alpha_esc;
    p.ln_Mlim_esc = log(sc->Mlim_Fesc);
    p.Mturn = Mturn;
    return mf_integral(mf_integrand, &p, lnM_min, lnM_max);
}

/* hmf.c:470-477: n_ion per halo of the molecularly cooled population: pivot 1e7 Msun, lower
 * turnover exp(-M_turn/M), upper turnover exp(-M/M_acg) at the atomic cooling threshold */
static double nion_weight_mini(double lnM, double Mturn, const c21_scaling_consts *sc) {
    const double Fstar = log_pl_limit(lnM, log(sc->fstar_7), sc->alpha_star_mini, 7 * M_LN10,
                                      log(sc->Mlim_Fstar_mini));
    const double Fesc =
        log_pl_limit(lnM, log(sc->fesc_7), sc->alpha_esc, 7 * M_LN10, log(sc->Mlim_Fesc_mini));
    const double M = exp(lnM);
    return exp(Fstar + Fesc - M / sc->acg_thresh - Mturn / M + lnM);
}

struct mf_mini_ctx {
    double growthf, Mturn;
    int hmf;
    const c21_scaling_consts *sc;
};
static double mf_mini_integrand(double lnM, void *ctx) {
    const struct mf_mini_ctx *p = (const struct mf_mini_ctx *)ctx;
    return nion_weight_mini(lnM, p->Mturn, p->sc) * unconditional_mf(p->growthf, lnM, p->hmf);
}

/* hmf.c:973-990 */
double c21_Nion_General_MINI(double z, double lnM_min, double lnM_max, double Mturn,
                             const c21_scaling_consts *sc) {
    if (!supported_hmf()) return NAN;
    struct mf_mini_ctx p = {dicke(z), Mturn, matter_options_global->HMF, sc};
    return mf_integral(mf_mini_integrand, &p, lnM_min, lnM_max);
}

/* ---------------------------------------------------------------- conditional mass function
 * The E-INTEGRAL source model evaluates, per filter radius, N_ion(delta | M_cond = M(R)) on a
 * grid of 400 overdensities (interp_tables.c:291-405) as a Gauss-Legendre sum over ln M of
 * n_ion(M) x conditional mass function (hmf.c:1106-1140, 703-730). */
#define JENKINS_a 0.73 /* hmf.c:48-50 */
#define JENKINS_b 0.34
#define JENKINS_c 0.81
#define MAX_DELTAC_FRAC ((float)0.99) /* hmf.h:8 */
#define NGL_INT 100                   /* hmf.c:87 */

/* hmf.c:151-154 */
static double sheth_delc_fixed(double del, double sig) {
    return sqrt(JENKINS_a) * del * (1. + JENKINS_b * pow(sig * sig / (JENKINS_a * del * del), JENKINS_c));
}

/* hmf.c:166-171 (Delos not supported here) */
static double get_delta_crit(int hmf, double sigma, double growthf) {
    if (hmf == C21CM_HMF_ST) return sheth_delc_fixed(DELTA_C_SPH / growthf, sigma) * growthf;
    return DELTA_C_SPH;
}

/* hmf.c:234-267: Taylor expansion of the moving barrier about sigma_cond */
static double st_taylor_factor(double sig, double sig_cond, double growthf, double *zeroth_order) {
    const double a = JENKINS_a, alpha = JENKINS_c, beta = JENKINS_b;
    const double del = DELTA_C_SPH / growthf;
    const double sigsq = sig * sig, sigsq_inv = 1. / sigsq, sigcsq = sig_cond * sig_cond;
    const double sigdiff = sig == sig_cond ? 1e-6 : sigsq - sigcsq;
    double t_array[6];
    t_array[0] = 1.;
    for (int i = 1; i < 6; i++)
        t_array[i] = t_array[i - 1] * (-sigdiff) / i * (alpha - i + 1) * sigsq_inv;
    double result = 0.;
    for (int i = 5; i >= 0; i--) result += t_array[i];
    const double prefactor_1 = sqrt(a) * del;
    const double prefactor_2 = beta * pow(sigsq_inv * (a * del * del), -alpha);
    result = prefactor_1 * (1 + prefactor_2 * result);
    *zeroth_order = prefactor_1 * (1 + prefactor_2);
    return result;
}

/* hmf.c:270-285 (Sheth-Mo-Tormen) and :317-330 (extended Press-Schechter) */
static double conditional_mf(double growthf, double lnM, double delta_cond, double sigma_cond,
                             int hmf) {
    const double M = exp(lnM);
    const double sigma1 = c21_sigma_fast(M);
    const double dsigmasqdm = dsigmasqdm_fast(M);
    if (sigma1 < sigma_cond) return 0.;
    const double sigdiff_inv =
        sigma1 == sigma_cond ? 1e6 : 1 / (sigma1 * sigma1 - sigma_cond * sigma_cond);
    if (hmf == C21CM_HMF_ST) {
        double Barrier;
        const double delta_0 = delta_cond / growthf;
        const double factor = st_taylor_factor(sigma1, sigma_cond, growthf, &Barrier) - delta_0;
        return -dsigmasqdm * factor * pow(sigdiff_inv, 1.5) *
               exp(-(Barrier - delta_0) * (Barrier - delta_0) * 0.5 * sigdiff_inv) / sqrt(2. * M_PI);
    }
    const double del = (DELTA_C_SPH - delta_cond) / growthf;
    return -del * dsigmasqdm * pow(sigdiff_inv, 1.5) * exp(-del * del * 0.5 * sigdiff_inv) /
           sqrt(2. * M_PI);
}

/* hmf.c:664-700 (gauleg, Numerical-Recipes form, 1-based) */
static void gauleg(double x1, double x2, double *x, double *w, int n) {
    const int m = (n + 1) / 2;
    const double xm = 0.5 * (x2 + x1), xl = 0.5 * (x2 - x1);
    for (int i = 1; i <= m; i++) {
        double z = cos(3.141592654 * (i - 0.25) / (n + 0.5)), z1, pp;
        do {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; j++) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        } while (fabs(z - z1) > 3.0e-11);
        x[i] = xm - xl * z;
        x[n + 1 - i] = xm + xl * z;
        w[i] = 2.0 * xl / ((1.0 - z * z) * pp * pp);
        w[n + 1 - i] = w[i];
    }
}

static struct {
    double lo, hi;
    int ready;
    double x[NGL_INT + 1], w[NGL_INT + 1];
} gl;

static void initialise_GL(double lnM_min, double lnM_max) { /* hmf.c:703-711 */
    if (gl.ready && lnM_min == gl.lo && lnM_max == gl.hi) return;
    gauleg(lnM_min, lnM_max, gl.x, gl.w, NGL_INT);
    gl.lo = lnM_min;
    gl.hi = lnM_max;
    gl.ready = 1;
}

struct cmf_ctx {
    struct mf_ctx m;
    double delta, sigma_cond;
};

static double cnion_integrand(double lnM, void *ctx) { /* hmf.c:541-543 */
    const struct cmf_ctx *c = (const struct cmf_ctx *)ctx;
    const struct mf_ctx *p = &c->m;
    const double Fstar = log_pl_limit(lnM, p->ln_fstar_norm, p->alpha_star, 10 * M_LN10, p->ln_Mlim_star);
    const double Fesc = log_pl_limit(lnM, p->ln_fesc_norm, p->alpha_esc, 10 * M_LN10, p->ln_Mlim_esc);
    return exp(Fstar + Fesc - p->Mturn / exp(lnM) + lnM) *
           conditional_mf(p->growthf, lnM, c->delta, c->sigma_cond, p->hmf);
}

/* hmf.c:1106-1140.  method: 0 adaptive quadrature (the reference: GSL QAG, rel 1e-3), 1
 * Gauss-Legendre with NGL_INT points over [lnM1, lnM2] (the default).  The Gamma-function
 * approximation (method 2) is not provided. */
double c21_Nion_ConditionalM(double growthf, double lnM1, double lnM2, double lnM_cond,
                             double sigma2, double delta2, double Mturn,
                             const c21_scaling_consts *sc, int method) {
    struct cmf_ctx c;
    memset(&c, 0, sizeof(c));
    c.m.growthf = growthf;
    c.m.hmf = matter_options_global->HMF;
    c.m.kind = 1;
    c.m.ln_fstar_norm = log(sc->fstar_10);
    c.m.alpha_star = sc->alpha_star;
    c.m.ln_Mlim_star = log(sc->Mlim_Fstar);
    c.m.ln_fesc_norm = log(sc->fesc_10);
    c.m.alpha_esc = sc->alpha_esc;
    c.m.ln_Mlim_esc = log(sc->Mlim_Fesc);
    c.m.Mturn = Mturn;
    c.delta = delta2;
    c.sigma_cond = sigma2;
    if (lnM1 >= lnM_cond) return 0.;
    if (delta2 > MAX_DELTAC_FRAC * get_delta_crit(c.m.hmf, sigma2, growthf)) {
        /* one halo at the condition mass when the cell itself has collapsed */
        if (lnM_cond * (1 - FRACT_FLOAT_ERR) <= lnM2) {
            const double Fstar = log_pl_limit(lnM_cond, c.m.ln_fstar_norm, c.m.alpha_star,
                                              10 * M_LN10, c.m.ln_Mlim_star);
            const double Fesc = log_pl_limit(lnM_cond, c.m.ln_fesc_norm, c.m.alpha_esc, 10 * M_LN10,
                                             c.m.ln_Mlim_esc);
            return exp(Fstar + Fesc - Mturn / exp(lnM_cond) + lnM_cond) / exp(lnM_cond);
        }
        return 0.;
    }
    if (c.m.hmf != C21CM_HMF_PS && c.m.hmf != C21CM_HMF_ST) c.m.hmf = C21CM_HMF_PS;
    if (method == 1) {
        initialise_GL(lnM1, lnM2);
        double integral = 0;
        for (int i = 1; i < NGL_INT + 1; i++) integral += gl.w[i] * cnion_integrand(gl.x[i], &c);
        return integral;
    }
    return c21_integrate(cnion_integrand, &c, lnM1, lnM2, 1e-4);
}

/* interp_tables.c:291-405 (1-D case; ln_floor = -40) and :415-494 (SFRD table: the same integral
 * with f_esc = 1, ln_floor = -50): table[i] = max(ln N_ion(delta_i | M_cond), ln_floor) on
 * n_delta overdensities from dmin to dmax.  With the Gauss-Legendre method everything that does
 * not depend on delta (sigma, d sigma^2/dM, n_ion(M), the barrier expansion) is evaluated once per
 * node, which leaves one exp per (node, delta): the values equal c21_Nion_ConditionalM's to
 * rounding (1e-15). */
#define S_PER_YR 31556925.9747 /* Constants.c:16 */
/* per-mass weights of the two table families */
typedef double (*mass_weight_fn)(double lnM, double Mturn, const c21_scaling_consts *sc);

static double nion_weight(double lnM, double Mturn, const c21_scaling_consts *sc) {
    const double Fstar = log_pl_limit(lnM, log(sc->fstar_10), sc->alpha_star, 10 * M_LN10, log(sc->Mlim_Fstar));
    const double Fesc = log_pl_limit(lnM, log(sc->fesc_10), sc->alpha_esc, 10 * M_LN10, log(sc->Mlim_Fesc));
    return exp(Fstar + Fesc - Mturn / exp(lnM) + lnM);
}

/* scaling_relations.c:446-467 (Eqs. 14, 15 of arXiv:2504.17254) */
static double halo_metallicity(double sfr, double stellar, double redshift) {
    const double redshift_scaling = pow(10, -0.056 * redshift + 0.064);
    double stellar_term = 1.;
    if (stellar > 0 && sfr > 0.) {
        const double M0 = 1.28825e10 * pow(sfr * S_PER_YR, 0.56);
        stellar_term = pow(1 + pow(stellar / M0, -2.1), -0.148);
    }
    return 1.23 * stellar_term * redshift_scaling;
}

/* scaling_relations.c:236-240,277-283,315-325 */
static double lx_on_sfr(double metallicity, double lx_constant) {
    if (!astro_options_global->USE_UPPER_STELLAR_TURNOVER) return lx_constant;
    const double hi_z_index = -0.64, lo_z_index = 0., z_pivot = 0.05;
    return lx_constant * 1. /
           (pow(metallicity / z_pivot, -lo_z_index) + pow(metallicity / z_pivot, -hi_z_index));
}

/* hmf.c:482-509 with USE_MINI_HALOS off */
double c21_xray_fraction(double lnM, double Mturn, const c21_scaling_consts *sc) {
    const double M = exp(lnM);
    const double ln_norm = log(sc->fstar_10);
    const double Fstar =
        exp(log_pl_limit(lnM, ln_norm, sc->alpha_star, 10 * M_LN10, log(sc->Mlim_Fstar)) - Mturn / M + ln_norm);
    const double stars = M * Fstar * cosmo_params_global->OMb / cosmo_params_global->OMm;
    const double sfr = stars / (sc->t_star * sc->t_h);
    const double metallicity = halo_metallicity(sfr, stars, sc->redshift);
    return S_PER_YR * (sfr * lx_on_sfr(metallicity, sc->l_x));
}

/* hmf.c:482-509 with USE_MINI_HALOS: both populations' X-ray luminosity of a halo; `Mturn` is the
 * molecular turnover, the atomic one is sc->mturn_a_nofb (Xray_ConditionalM :1142-1165) */
static double xray_fraction_mini(double lnM, double Mturn, const c21_scaling_consts *sc) {
    const double M = exp(lnM);
    const double ln_norm = log(sc->fstar_10), ln_norm_m = log(sc->fstar_7);
    const double Fstar = exp(log_pl_limit(lnM, ln_norm, sc->alpha_star, 10 * M_LN10, log(sc->Mlim_Fstar)) -
                             sc->mturn_a_nofb / M + ln_norm);
    const double Fstar_mini =
        exp(log_pl_limit(lnM, ln_norm_m, sc->alpha_star_mini, 7 * M_LN10, log(sc->Mlim_Fstar_mini)) -
            Mturn / M - M / sc->acg_thresh + ln_norm_m);
    const double b = cosmo_params_global->OMb / cosmo_params_global->OMm;
    const double stars = M * Fstar * b, stars_mini = M * Fstar_mini * b;
    const double sfr = stars / (sc->t_star * sc->t_h), sfr_mini = stars_mini / (sc->t_star * sc->t_h);
    const double metallicity = halo_metallicity(sfr + sfr_mini, stars + stars_mini, sc->redshift);
    return S_PER_YR * (sfr * lx_on_sfr(metallicity, sc->l_x) + sfr_mini * lx_on_sfr(metallicity, sc->l_x_mini));
}

static int conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                             double sigma_cond, double dmin, double dmax, double Mturn,
                             const c21_scaling_consts *sc, int method, double ln_floor,
                             mass_weight_fn weight, float *table, int n_delta);

int c21_Nion_Conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double dmin, double dmax, double Mturn,
                               const c21_scaling_consts *sc, int method, double ln_floor,
                               float *table, int n_delta) {
    return conditional_table(growthf, lnMmin, lnMmax, lnMcond, sigma_cond, dmin, dmax, Mturn, sc,
                             method, ln_floor, nion_weight, table, n_delta);
}

int c21_Xray_Conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double dmin, double dmax, double Mturn,
                               const c21_scaling_consts *sc, int method, float *table,
                               int n_delta) {
    return conditional_table(growthf, lnMmin, lnMmax, lnMcond, sigma_cond, dmin, dmax, Mturn, sc,
                             method, -50., c21_xray_fraction, table, n_delta);
}

/* One weight integrated against the conditional mass function for one overdensity
 * (hmf.c:1106-1140 / 1142-1177 for weights other than n_ion) */
struct wcmf_ctx {
    mass_weight_fn weight;
    const c21_scaling_consts *sc;
    double Mturn, growthf, delta, sigma_cond;
    int hmf;
};
static double wcmf_integrand(double lnM, void *ctx) {
    const struct wcmf_ctx *c = (const struct wcmf_ctx *)ctx;
    return c->weight(lnM, c->Mturn, c->sc) *
           conditional_mf(c->growthf, lnM, c->delta, c->sigma_cond, c->hmf);
}
static double weighted_ConditionalM(mass_weight_fn weight, double growthf, double lnM1, double lnM2,
                                    double lnM_cond, double sigma2, double delta2, double Mturn,
                                    const c21_scaling_consts *sc, int method) {
    struct wcmf_ctx c = {weight, sc, Mturn, growthf, delta2, sigma2, matter_options_global->HMF};
    if (lnM1 >= lnM_cond) return 0.;
    if (delta2 > MAX_DELTAC_FRAC * get_delta_crit(c.hmf, sigma2, growthf)) {
        if (lnM_cond * (1 - FRACT_FLOAT_ERR) <= lnM2) return weight(lnM_cond, Mturn, sc) / exp(lnM_cond);
        return 0.;
    }
    if (c.hmf != C21CM_HMF_PS && c.hmf != C21CM_HMF_ST) c.hmf = C21CM_HMF_PS;
    if (method == 1) {
        initialise_GL(lnM1, lnM2);
        double integral = 0;
        for (int i = 1; i < NGL_INT + 1; i++) integral += gl.w[i] * wcmf_integrand(gl.x[i], &c);
        return integral;
    }
    return c21_integrate(wcmf_integrand, &c, lnM1, lnM2, 1e-4);
}

static int conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                             double sigma_cond, double dmin, double dmax, double Mturn,
                             const c21_scaling_consts *sc, int method, double ln_floor,
                             mass_weight_fn weight, float *table, int n_delta) {
    int hmf = matter_options_global->HMF;
    const int fast = (method == 1) && lnMmin < lnMcond;
    double node_pref[NGL_INT + 1], node_factor[NGL_INT + 1], node_barrier[NGL_INT + 1],
        node_sdi[NGL_INT + 1];
    if (hmf != C21CM_HMF_PS && hmf != C21CM_HMF_ST) hmf = C21CM_HMF_PS;
    if (fast) {
        initialise_GL(lnMmin, lnMmax);
        for (int i = 1; i < NGL_INT + 1; i++) {
            const double lnM = gl.x[i], M = exp(lnM);
            const double sigma1 = c21_sigma_fast(M), dsigmasqdm = dsigmasqdm_fast(M);
            const double nion = weight(lnM, Mturn, sc);
            if (sigma1 < sigma_cond) { /* conditional_mf returns 0 */
                node_pref[i] = 0.;
                node_factor[i] = node_barrier[i] = node_sdi[i] = 0.;
                continue;
            }
            const double sdi = sigma1 == sigma_cond ? 1e6 : 1 / (sigma1 * sigma1 - sigma_cond * sigma_cond);
            node_sdi[i] = sdi;
            /* n_ion(M) d sigma^2/dM (sigma1^2 - sigma_c^2)^-3/2 / sqrt(2 pi): everything of the
             * integrand that does not depend on delta (the sign follows below) */
            node_pref[i] = nion * dsigmasqdm * pow(sdi, 1.5) / sqrt(2. * M_PI);
            if (hmf == C21CM_HMF_ST)
                node_factor[i] = st_taylor_factor(sigma1, sigma_cond, growthf, &node_barrier[i]);
            else
                node_factor[i] = node_barrier[i] = 0.;
        }
    }
    /* the overdensities are independent: N_THREADS host threads, as upstream (interp_tables.c:340) */
    int bad = 0;
    const int n_thr = simulation_options_global && simulation_options_global->N_THREADS > 1
                          ? simulation_options_global->N_THREADS : 1;
    if (!fast && method == 1) initialise_GL(lnMmin, lnMmax); /* before the threads share it */
#pragma omp parallel for schedule(static) num_threads(n_thr) reduction(| : bad)
    for (int k = 0; k < n_delta; k++) {
        const double delta = dmin + (float)k / ((float)n_delta - 1.) * (dmax - dmin);
        double v;
        if (!fast || delta > MAX_DELTAC_FRAC * get_delta_crit(hmf, sigma_cond, growthf)) {
            v = weighted_ConditionalM(weight, growthf, lnMmin, lnMmax, lnMcond, sigma_cond, delta,
                                      Mturn, sc, method);
        } else {
            double integral = 0;
            for (int i = 1; i < NGL_INT + 1; i++) {
                if (node_pref[i] == 0.) {
                    integral += gl.w[i] * 0.;
                    continue;
                }
                const double sdi = node_sdi[i];
                double cmf;
                if (hmf == C21CM_HMF_ST) {
                    const double delta_0 = delta / growthf;
                    const double factor = node_factor[i] - delta_0;
                    const double B = node_barrier[i];
                    cmf = factor * exp(-(B - delta_0) * (B - delta_0) * 0.5 * sdi);
                } else {
                    const double del = (DELTA_C_SPH - delta) / growthf;
                    cmf = del * exp(-del * del * 0.5 * sdi);
                }
                integral += gl.w[i] * (-node_pref[i] * cmf);
            }
            v = integral;
        }
        double lv = log(v);
        if (lv < ln_floor) lv = ln_floor;
        if (!isfinite(lv)) bad |= 1;
        table[k] = (float)lv;
    }
    return bad ? C21CM_TABLE_GENERATION_ERROR : 0;
}

/* E-INTEGRAL without interpolation tables: Nion_ConditionalM of EVERY cell (IonisationBox.c:889-893,
 * hmf.c:1106-1140).  With the Gauss-Legendre method the integrand factorises exactly as in
 * conditional_table() above; this fills the delta-independent node data of one radius
 * (include/c21cm_grid.h: C21CM_FCOLL_NODES) and the device sums over the nodes per cell.  The
 * per-cell values equal c21_Nion_ConditionalM(..., method = 1) to rounding. */
int c21_Nion_Conditional_nodes(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double Mturn, const c21_scaling_consts *sc,
                               double *nodes) {
    int hmf = matter_options_global->HMF;
    if (hmf != C21CM_HMF_PS && hmf != C21CM_HMF_ST) hmf = C21CM_HMF_PS;
    memset(nodes, 0, sizeof(double) * C21CM_NODE_DOUBLES);
    nodes[0] = NGL_INT;
    nodes[1] = hmf == C21CM_HMF_ST ? 1. : 0.;
    nodes[2] = growthf;
    nodes[3] = MAX_DELTAC_FRAC * get_delta_crit(matter_options_global->HMF, sigma_cond, growthf);
    nodes[4] = (lnMcond * (1 - FRACT_FLOAT_ERR) <= lnMmax) ? nion_weight(lnMcond, Mturn, sc) / exp(lnMcond) : 0.;
    nodes[5] = (lnMmin >= lnMcond) ? 1. : 0.;
    if (nodes[5] != 0.) return 0;
    initialise_GL(lnMmin, lnMmax);
    for (int i = 1; i < NGL_INT + 1; i++) {
        double *nd = nodes + 8 + 4 * (i - 1);
        const double lnM = gl.x[i], M = exp(lnM);
        const double sigma1 = c21_sigma_fast(M), dsigmasqdm = dsigmasqdm_fast(M);
        if (sigma1 < sigma_cond) continue; /* conditional_mf returns 0: the node's weight stays 0 */
        const double sdi = sigma1 == sigma_cond ? 1e6 : 1 / (sigma1 * sigma1 - sigma_cond * sigma_cond);
        const double pref = nion_weight(lnM, Mturn, sc) * dsigmasqdm * pow(sdi, 1.5) / sqrt(2. * M_PI);
        nd[0] = gl.w[i] * (-pref);
        if (hmf == C21CM_HMF_ST) nd[1] = st_taylor_factor(sigma1, sigma_cond, growthf, &nd[2]);
        nd[3] = sdi;
        if (!isfinite(nd[0]) || !isfinite(nd[1]) || !isfinite(nd[2])) return C21CM_TABLE_GENERATION_ERROR;
    }
    return 0;
}

/* The redshift tables of the molecularly cooled population for the spin temperature
 * (initialise_Nion_Ts_spline / initialise_SFRD_spline with USE_MINI_HALOS, interp_tables.c:96-232):
 *   nion[i * n_mturn + j] = Nion_General_MINI(z_i, ln M_min, ln M_max, 10^(l10_min + j l10_width), sc_z)
 *   sfrd[...]             = the same with f_esc = 1 (evolve_scaling_constants_sfr)
 * with sc_z the constants evolved to z_i (only the upper turnover exp(-M / M_acg(z)) matters).
 * Upstream: 2 x n_z x n_mturn adaptive integrals.  Here one 100-point Gauss-Legendre rule in
 * ln M over [M_min, 200 M_acg(z_min)] shared by all of them: sigma(M) and the two scaling-relation
 * factors per node once, exp(-M_turn_j / M) per (node, j) once, the mass function and the upper
 * turnover per (z, node) -- then a (n_z x nodes) x (nodes x n_mturn) product.  Agreement with the
 * adaptive integrals: tests/test_host_minihalos.py. */
int c21_Nion_z_tables_mini(int n_z, double z_min, double z_width, double lnMmin,
                           const c21_scaling_consts *sc, int n_mturn, double l10_min,
                           double l10_width, double *nion, double *sfrd) {
    enum { NQ = 100 };
    if (!supported_hmf()) return C21CM_VALUE_ERROR;
    if (n_mturn < 1 || n_mturn > 256) return C21CM_VALUE_ERROR;
    const int hmf = matter_options_global->HMF;
    const double acg_lo_z = c21_TtoM((float)z_min, 1e4, 0.59); /* the largest threshold of the range */
    double lnMmax = log(200. * acg_lo_z);
    if (lnMmax > log(1e16)) lnMmax = log(1e16);
    double x[NQ + 1], w[NQ + 1], sig[NQ + 1], dsig[NQ + 1], base_n[NQ + 1], base_s[NQ + 1];
    double *E = (double *)malloc(sizeof(double) * (size_t)(NQ + 1) * n_mturn);
    if (!E) return C21CM_MEMORY_ALLOC_ERROR;
    gauleg(lnMmin, lnMmax, x, w, NQ);
    const c21_scaling_consts sc_s = c21_scaling_consts_sfr(sc);
    for (int i = 1; i <= NQ; i++) {
        const double lnM = x[i], M = exp(lnM);
        sig[i] = c21_sigma_fast(M);
        dsig[i] = dsigmasqdm_fast(M);
        /* nion_weight_mini without its two turnover factors */
        base_n[i] = exp(log_pl_limit(lnM, log(sc->fstar_7), sc->alpha_star_mini, 7 * M_LN10,
                                     log(sc->Mlim_Fstar_mini)) +
                        log_pl_limit(lnM, log(sc->fesc_7), sc->alpha_esc, 7 * M_LN10,
                                     log(sc->Mlim_Fesc_mini)) + lnM);
        base_s[i] = exp(log_pl_limit(lnM, log(sc_s.fstar_7), sc_s.alpha_star_mini, 7 * M_LN10,
                                     log(sc_s.Mlim_Fstar_mini)) +
                        log_pl_limit(lnM, log(sc_s.fesc_7), sc_s.alpha_esc, 7 * M_LN10,
                                     log(sc_s.Mlim_Fesc_mini)) + lnM);
        for (int j = 0; j < n_mturn; j++)
            E[(size_t)i * n_mturn + j] = exp(-pow(10, l10_min + j * l10_width) / M);
    }
    const int n_thr = simulation_options_global && simulation_options_global->N_THREADS > 1
                          ? simulation_options_global->N_THREADS : 1;
    int bad = 0;
#pragma omp parallel for schedule(static) num_threads(n_thr) reduction(| : bad)
    for (int k = 0; k < n_z; k++) {
        const double z = z_min + k * z_width, growthf = dicke(z);
        const double acg = c21_TtoM((float)z, 1e4, 0.59);
        double *rn = nion + (size_t)k * n_mturn, *rs = sfrd + (size_t)k * n_mturn;
        for (int j = 0; j < n_mturn; j++) rn[j] = rs[j] = 0.;
        for (int i = 1; i <= NQ; i++) {
            const double sigma = sig[i] * growthf;
            const double dsigmadm = dsig[i] * (growthf * growthf / (2. * sigma));
            double mf;
            if (hmf == C21CM_HMF_PS) {
                mf = -(dsigmadm / sigma) * sqrt(2. / M_PI) * (DELTA_C_SPH / sigma) *
                     exp(-(DELTA_C_SPH * DELTA_C_SPH) / (2 * sigma * sigma));
            } else {
                const double nuhat = sqrt(SHETH_a) * DELTA_C_SPH / sigma;
                mf = -(dsigmadm / sigma) * sqrt(2. / M_PI) * SHETH_A * (1 + pow(nuhat, -2 * SHETH_p)) *
                     nuhat * exp(-nuhat * nuhat / 2.0);
            }
            const double g = w[i] * mf * exp(-exp(x[i]) / acg);
            const double gn = g * base_n[i], gs = g * base_s[i];
            const double *e = E + (size_t)i * n_mturn;
            for (int j = 0; j < n_mturn; j++) {
                rn[j] += gn * e[j];
                rs[j] += gs * e[j];
            }
        }
        for (int j = 0; j < n_mturn; j++)
            if (!isfinite(rn[j]) || !isfinite(rs[j])) bad |= 1;
    }
    free(E);
    return bad ? C21CM_TABLE_GENERATION_ERROR : 0;
}

/* hmf.c:1066-1104 */
double c21_Nion_ConditionalM_MINI(double growthf, double lnM1, double lnM2, double lnM_cond,
                                  double sigma2, double delta2, double Mturn,
                                  const c21_scaling_consts *sc, int method) {
    return weighted_ConditionalM(nion_weight_mini, growthf, lnM1, lnM2, lnM_cond, sigma2, delta2,
                                 Mturn, sc, method);
}

/* interp_tables.c:291-405 with USE_MINI_HALOS.  With the Gauss-Legendre method the integrand
 * factorises per node into (what depends on delta) x (what depends on M_turn), so one table
 * costs n_delta x NGL exponentials plus n_delta x NGL x n_mturn multiply-adds instead of
 * n_delta x n_mturn quadratures; values equal the per-entry integrals to rounding. */
int c21_Nion_Conditional_table2d(double growthf, double lnMmin, double lnMmax, double lnMcond,
                                 double sigma_cond, double dmin, double dmax, double l10mt_min,
                                 double l10mt_max, const c21_scaling_consts *sc, int mini,
                                 int method, double ln_floor, int float_mturn, float *table,
                                 int n_delta, int n_mturn) {
    int hmf = matter_options_global->HMF;
    /* mini: 0 atomically cooled N_ion, 1 molecularly cooled N_ion, 2 X-ray luminosity of both
     * populations over the molecular turnover (interp_tables.c:497-560) */
    const mass_weight_fn weight = mini == 2 ? xray_fraction_mini : (mini ? nion_weight_mini : nion_weight);
    const int fast = (method == 1) && lnMmin < lnMcond;
    if (n_mturn < 2 || n_mturn > 256) return C21CM_VALUE_ERROR;
    if (hmf != C21CM_HMF_PS && hmf != C21CM_HMF_ST) hmf = C21CM_HMF_PS;
    double *mturn = (double *)malloc(sizeof(double) * (size_t)n_mturn);
    double *wk = (double *)malloc(sizeof(double) * (size_t)(NGL_INT + 1) * (size_t)n_mturn);
    double node_pref[NGL_INT + 1], node_factor[NGL_INT + 1], node_barrier[NGL_INT + 1],
        node_sdi[NGL_INT + 1];
    if (!mturn || !wk) {
        free(mturn);
        free(wk);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    for (int j = 0; j < n_mturn; j++) /* float arithmetic of the ratio as upstream (:353-358) */
        mturn[j] = pow(10., l10mt_min + (float)j / ((float)n_mturn - 1.) * (l10mt_max - l10mt_min));
    if (float_mturn) /* the SFRD tables keep the turnover masses in a float array (:431-435) */
        for (int j = 0; j < n_mturn; j++) mturn[j] = (double)(float)mturn[j];
    if (method == 1) initialise_GL(lnMmin, lnMmax);
    if (fast) {
        for (int i = 1; i < NGL_INT + 1; i++) {
            const double lnM = gl.x[i], M = exp(lnM);
            const double sigma1 = c21_sigma_fast(M), dsigmasqdm = dsigmasqdm_fast(M);
            if (sigma1 < sigma_cond) { /* conditional_mf returns 0 */
                node_pref[i] = node_factor[i] = node_barrier[i] = node_sdi[i] = 0.;
                for (int j = 0; j < n_mturn; j++) wk[(size_t)i * n_mturn + j] = 0.;
                continue;
            }
            const double sdi =
                sigma1 == sigma_cond ? 1e6 : 1 / (sigma1 * sigma1 - sigma_cond * sigma_cond);
            node_sdi[i] = sdi;
            node_pref[i] = dsigmasqdm * pow(sdi, 1.5) / sqrt(2. * M_PI);
            if (hmf == C21CM_HMF_ST)
                node_factor[i] = st_taylor_factor(sigma1, sigma_cond, growthf, &node_barrier[i]);
            else
                node_factor[i] = node_barrier[i] = 0.;
            for (int j = 0; j < n_mturn; j++) wk[(size_t)i * n_mturn + j] = weight(lnM, mturn[j], sc);
        }
    }
    int bad = 0;
    const int n_thr = simulation_options_global && simulation_options_global->N_THREADS > 1
                          ? simulation_options_global->N_THREADS : 1;
#pragma omp parallel for schedule(static) num_threads(n_thr) reduction(| : bad)
    for (int k = 0; k < n_delta; k++) {
        const double delta = dmin + (float)k / ((float)n_delta - 1.) * (dmax - dmin);
        float *row = table + (size_t)k * n_mturn;
        if (!fast || delta > MAX_DELTAC_FRAC * get_delta_crit(hmf, sigma_cond, growthf)) {
            for (int j = 0; j < n_mturn; j++) {
                double lv = log(weighted_ConditionalM(weight, growthf, lnMmin, lnMmax, lnMcond,
                                                      sigma_cond, delta, mturn[j], sc, method));
                if (lv < ln_floor) lv = ln_floor;
                if (!isfinite(lv)) bad |= 1;
                row[j] = (float)lv;
            }
            continue;
        }
        double acc[256];
        for (int j = 0; j < n_mturn; j++) acc[j] = 0.;
        for (int i = 1; i < NGL_INT + 1; i++) {
            if (node_pref[i] == 0.) continue;
            const double sdi = node_sdi[i];
            double cmf;
            if (hmf == C21CM_HMF_ST) {
                const double delta_0 = delta / growthf;
                const double B = node_barrier[i];
                cmf = (node_factor[i] - delta_0) * exp(-(B - delta_0) * (B - delta_0) * 0.5 * sdi);
            } else {
                const double del = (DELTA_C_SPH - delta) / growthf;
                cmf = del * exp(-del * del * 0.5 * sdi);
            }
            const double node = gl.w[i] * (-node_pref[i] * cmf);
            const double *w = wk + (size_t)i * n_mturn;
            for (int j = 0; j < n_mturn; j++) acc[j] += node * w[j];
        }
        for (int j = 0; j < n_mturn; j++) {
            double lv = log(acc[j]);
            if (lv < ln_floor) lv = ln_floor;
            if (!isfinite(lv)) bad |= 1;
            row[j] = (float)lv;
        }
    }
    free(mturn);
    free(wk);
    return bad ? C21CM_TABLE_GENERATION_ERROR : 0;
}

/* hmf.c:1268-1316: mass beyond which F = FRAC (M/1e10)^PL would exceed 1 (float bisection) */
static float mass_limit(float logM, float PL, float FRAC) { return FRAC * pow(pow(10., logM) / 1e10, PL); }

static float mass_limit_bisection(float Mmin, float Mmax, float PL, float FRAC, int *status) {
    int iter = 0;
    const int max_iter = 200;
    const float rel_tol = 0.001;
    float lo = log10(Mmin), hi = log10(Mmax), x, x1;
    if (PL < 0.) {
        if (mass_limit(lo, PL, FRAC) <= 1.) return Mmin;
    } else if (PL > 0.) {
        if (mass_limit(hi, PL, FRAC) <= 1.) return Mmax;
    } else
        return 0;
    x = (lo + hi) / 2.;
    ++iter;
    do {
        if ((mass_limit(lo, PL, FRAC) - 1.) * (mass_limit(x, PL, FRAC) - 1.) < 0.)
            hi = x;
        else
            lo = x;
        x1 = (lo + hi) / 2.;
        ++iter;
        if (fabs(x1 - x) < rel_tol) return pow(10., x1);
        x = x1;
    } while (iter < max_iter);
    *status = C21CM_MASSDEPZETA_ERROR;
    return 0.f;
}

/* scaling_relations.c:36-119 (fields used by the ionisation path) */
size_t c21_scaling_consts_size(void) { return sizeof(c21_scaling_consts); }

int c21_set_scaling_constants(double redshift, c21_scaling_consts *sc) {
    int status = 0;
    const AstroParams *ap = astro_params_global;
    memset(sc, 0, sizeof(*sc));
    sc->fstar_10 = ap->F_STAR10;
    sc->alpha_star = ap->ALPHA_STAR;
    sc->fstar_7 = ap->F_STAR7_MINI;
    sc->t_h = 1.0 / c21_hubble((float)redshift);
    sc->t_star = ap->t_STAR;
    sc->alpha_esc = ap->ALPHA_ESC;
    sc->fesc_10 = ap->F_ESC10;
    sc->fesc_7 = ap->F_ESC7_MINI;
    sc->pop2_ion = ap->POP2_ION;
    sc->pop3_ion = ap->POP3_ION;
    sc->acg_thresh = c21_TtoM((float)redshift, 1e4, 0.59); /* thermochem.c:277 */
    sc->mturn_a_nofb = ap->M_TURN;
    sc->l_x = ap->L_X * 1e-38; /* scaling_relations.c:63 */
    sc->redshift = redshift;
    sc->Mlim_Fstar = mass_limit_bisection(1e5, 1e16, sc->alpha_star, sc->fstar_10, &status);
    sc->Mlim_Fesc = mass_limit_bisection(1e5, 1e16, sc->alpha_esc, sc->fesc_10, &status);
    if (astro_options_global->USE_MINI_HALOS) { /* scaling_relations.c:53-54,64,87-118 */
        sc->alpha_star_mini = ap->ALPHA_STAR_MINI;
        sc->l_x_mini = ap->L_X_MINI * 1e-38;
        sc->mturn_a_nofb = fmax(sc->acg_thresh, sc->mturn_a_nofb);
        switch (matter_options_global->V_CB_MODEL) {
            case C21CM_VCB_AVG_AUTO:
                sc->vcb_const = cosmo_tables_global->V_CB_AVG;
                break;
            case C21CM_VCB_AVG_DEBUG:
                sc->vcb_const = ap->V_CB_AVG_DEBUG;
                break;
            default: /* none, or the per-cell field */
                sc->vcb_const = 0.;
        }
        sc->mturn_m_nofb = c21_lyman_werner_threshold((float)redshift, 0.f, (float)sc->vcb_const);
        sc->Mlim_Fstar_mini =
            mass_limit_bisection(1e5, 1e16, sc->alpha_star_mini,
                                 sc->fstar_7 * pow(1e3, sc->alpha_star_mini), &status);
        sc->Mlim_Fesc_mini = mass_limit_bisection(1e5, 1e16, sc->alpha_esc,
                                                  sc->fesc_7 * pow(1e3, sc->alpha_esc), &status);
    }
    return status;
}

/* scaling_relations.c:132-165 (without the photon-conservation fits) */
c21_scaling_consts c21_scaling_consts_at_z(double redshift, const c21_scaling_consts *sc) {
    c21_scaling_consts sc_z = *sc;
    sc_z.redshift = redshift;
    sc_z.t_h = 1.0 / c21_hubble((float)redshift);
    sc_z.acg_thresh = c21_TtoM((float)redshift, 1e4, 0.59);
    sc_z.mturn_a_nofb = astro_params_global->M_TURN;
    sc_z.mturn_m_nofb = 0.;
    if (astro_options_global->USE_MINI_HALOS) {
        sc_z.mturn_a_nofb = fmax(sc_z.acg_thresh, sc_z.mturn_a_nofb);
        sc_z.mturn_m_nofb = c21_lyman_werner_threshold((float)redshift, 0.f, (float)sc_z.vcb_const);
    }
    return sc_z;
}

/* scaling_relations.c:121-130 */
c21_scaling_consts c21_scaling_consts_sfr(const c21_scaling_consts *sc) {
    c21_scaling_consts sc_sfrd = *sc;
    sc_sfrd.fesc_10 = 1.;
    sc_sfrd.fesc_7 = 1.;
    sc_sfrd.alpha_esc = 0.;
    sc_sfrd.Mlim_Fesc = 0.;
    sc_sfrd.Mlim_Fesc_mini = 0.;
    return sc_sfrd;
}

/* mimic_scatter_in_consts (scaling_relations.c:170-197): HALO_SCALING_RELATIONS_MEDIAN -- the
 * means of the log-normal relations sit above their medians, which the integrated grids mimic by
 * raised normalisations (and re-derived f_* mass limits) */
int c21_scaling_consts_mimic_scatter(c21_scaling_consts *sc) {
    int status = 0;
    const AstroParams *ap = astro_params_global;
    sc->fstar_10 *= exp(0.5 * pow(ap->SIGMA_STAR, 2));
    sc->fstar_7 *= exp(0.5 * pow(ap->SIGMA_STAR, 2));
    sc->l_x *= exp(0.5 * pow(ap->SIGMA_LX, 2));
    sc->l_x_mini *= exp(0.5 * pow(ap->SIGMA_LX, 2));
    sc->t_star /= exp(0.5 * pow(ap->SIGMA_SFR_LIM, 2));
    sc->Mlim_Fstar = mass_limit_bisection(1e5, 1e16, sc->alpha_star, sc->fstar_10, &status);
    if (astro_options_global->USE_MINI_HALOS)
        sc->Mlim_Fstar_mini = mass_limit_bisection(1e5, 1e16, sc->alpha_star_mini,
                                                   sc->fstar_7 * pow(1e3, sc->alpha_star_mini), &status);
    return status;
}

/* thermochem.c:281-304: Lyman-Werner + streaming-velocity threshold of molecular cooling */
double c21_lyman_werner_threshold(float z, float J_21_LW, float vcb) {
    const AstroParams *ap = astro_params_global;
    const double mcrit_noLW = 3.314e7 * pow(1. + z, -1.5);
    const double f_LW = 1.0 + ap->A_LW * pow(J_21_LW, ap->BETA_LW);
    const double sigma_vcb = cosmo_tables_global->V_CB_AVG * sqrt(3 * M_PI / 8);
    const double f_vcb = pow(1.0 + ap->A_VCB * vcb / sigma_vcb, ap->BETA_VCB);
    return mcrit_noLW * f_LW * f_vcb;
}

/* thermochem.c:306-311 (Sobacchi & Mesinger 2013) */
double c21_reionization_feedback(float z, float Gamma_halo_HII, float z_IN) {
    if (z_IN <= 1e-19) return 1e-40;
    return 3e9 * pow(2.0 * Gamma_halo_HII, 0.17) * pow((1. + z) / 10, -2.1) *
           pow(1 - pow((1. + z) / (1. + z_IN), 2.0), 2.5);
}

/* hmf.c:1319-1348 */
double c21_minimum_source_mass(double redshift) {
    const int mass_dep = matter_options_global->SOURCE_MODEL != C21CM_SOURCE_CONST_ION_EFF;
    const double min_factor = (mass_dep && !astro_options_global->USE_MINI_HALOS) ? 50. : 1.;
    double Mmin;
    if (astro_options_global->USE_MINI_HALOS) {
        Mmin = 1e5;
    } else if (astro_options_global->M_MIN_in_Mass) {
        Mmin = astro_params_global->M_TURN;
    } else {
        const double t_vir_min = astro_params_global->ION_Tvir_MIN;
        const double mu = t_vir_min < 9.99999e3 ? 1.22 : 0.6;
        Mmin = c21_TtoM(redshift, t_vir_min, mu);
    }
    return Mmin / min_factor;
}

/* ---------------------------------------------------------------- RECFAST table */
#define RECFAST_NPTS 501 /* heating_helper_progs.h */
static struct {
    int loaded;
    int n;
    double z[RECFAST_NPTS], tk[RECFAST_NPTS], xe[RECFAST_NPTS];
    double tk2[RECFAST_NPTS], xe2[RECFAST_NPTS]; /* natural-spline second derivatives */
    char path[600];
} rf;

/* natural cubic spline (what gsl_interp_cspline provides) */
static void spline_setup(int n, const double *x, const double *y, double *y2) {
    double *u = (double *)malloc(sizeof(double) * (size_t)n);
    y2[0] = u[0] = 0.;
    for (int i = 1; i < n - 1; i++) {
        const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
        const double p = sig * y2[i - 1] + 2.0;
        y2[i] = (sig - 1.0) / p;
        u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
    }
    y2[n - 1] = 0.;
    for (int k = n - 2; k >= 0; k--) y2[k] = y2[k] * y2[k + 1] + u[k];
    free(u);
}

static double spline_eval(int n, const double *x, const double *y, const double *y2, double v) {
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (hi + lo) >> 1;
        if (x[mid] > v)
            hi = mid;
        else
            lo = mid;
    }
    const double h = x[hi] - x[lo];
    const double a = (x[hi] - v) / h, b = (v - x[lo]) / h;
    return a * y[lo] + b * y[hi] + ((a * a * a - a) * y2[lo] + (b * b * b - b) * y2[hi]) * (h * h) / 6.0;
}

/* heating_helper_progs.c:94-190: columns z, x_e, (unused), T_K; rows from high z to low z */
int c21_recfast_load(void) {
    char filename[600];
    if (!config_settings.external_table_path) {
        c21hip_set_error("RECFAST: config_settings.external_table_path is not set");
        return C21CM_IO_ERROR;
    }
    snprintf(filename, sizeof(filename), "%s/recfast_LCDM.dat", config_settings.external_table_path);
    if (rf.loaded && strcmp(filename, rf.path) == 0) return 0;
    FILE *F = fopen(filename, "r");
    if (!F) {
        c21hip_set_error("RECFAST: unable to open %s", filename);
        return C21CM_IO_ERROR;
    }
    double zt[RECFAST_NPTS], xe[RECFAST_NPTS], tk[RECFAST_NPTS];
    int n = 0;
    float cz, cx, trash, ct;
    while (n < RECFAST_NPTS && fscanf(F, "%f %E %E %E", &cz, &cx, &trash, &ct) == 4) {
        zt[n] = cz;
        xe[n] = cx;
        tk[n] = ct;
        n++;
    }
    fclose(F);
    if (n < 4) {
        c21hip_set_error("RECFAST: %s holds only %d rows", filename, n);
        return C21CM_IO_ERROR;
    }
    for (int i = 0; i < n; i++) { /* ascending redshift */
        rf.z[i] = zt[n - 1 - i];
        rf.xe[i] = xe[n - 1 - i];
        rf.tk[i] = tk[n - 1 - i];
    }
    rf.n = n;
    spline_setup(n, rf.z, rf.tk, rf.tk2);
    spline_setup(n, rf.z, rf.xe, rf.xe2);
    strncpy(rf.path, filename, sizeof(rf.path) - 1);
    rf.loaded = 1;
    return 0;
}

double c21_T_RECFAST(float z) { return spline_eval(rf.n, rf.z, rf.tk, rf.tk2, z); }
double c21_xion_RECFAST(float z) { return spline_eval(rf.n, rf.z, rf.xe, rf.xe2, z); }
/* heating_helper_progs.c:197 */
float c21_cT_approx(float z) { return 0.58 - 0.006 * (z - 10.0); }
