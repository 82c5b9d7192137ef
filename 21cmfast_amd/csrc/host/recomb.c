/*
 * recomb.c -- host tables of the inhomogeneous-recombination model (Sobacchi & Mesinger 2014 on
 * the Miralda-Escude, Haehnelt & Rees 2000 density PDF with Rahmati+ 2013 self-shielding).
 *
 * reference: src/py21cmfast/src/recombinations.c
 *   :64-92    splined_recombination_rate  -> evaluated on the device (ionize_kernels.hip) from
 *                                            the table built here
 *   :94-122   init_MHR                    -> c21_rr_tables / init_MHR
 *   :143-148  Gamma_SS   (Rahmati+ 2013 fit, alpha_UVB = 5)
 *   :155-176  MHR_rr     integrand in ln Delta
 *   :181-215  recombination_rate: integral over ln Delta in [ln 0.01, ln 200]
 *   :217-283  A_MHR: normalisation of the PDF, 1 / int P(Delta) dDelta at z = 2..61, splined
 *   :285-370  C_MHR / beta_MHR: MHR00 table values, splined in z
 * and src/py21cmfast/src/thermochem.c:65-110 (alpha_A, alpha_B, neutral_fraction).
 *
 * The reference integrates with GSL QAG (61-point rule) at relative tolerances 1e-2 (the rate)
 * and 1e-3 (the normalisation); here the same integrals converge to 1e-6 with the adaptive
 * Gauss-Kronrod of cosmology.c, i.e. the two agree to the reference's own tolerance.  Splines are
 * natural cubic splines held as gsl_interp_cspline holds them (c_i = y''_i / 2).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"
#include "cosmology.h"

#define PC_M_P 1.67262192369e-24
#define PC_G 6.6743e-8
#define ALPHA_B_10K 2.59e-13 /* Constants.c:38 */
#define TINY 1e-30

/* natural cubic spline: c[i] = y''(x_i) / 2 (tridiagonal solve, as gsl cspline_init) */
static void cspline_c(int n, const double *x, const double *y, double *c) {
    double *g = (double *)malloc(sizeof(double) * (size_t)n), *diag = (double *)malloc(sizeof(double) * (size_t)n),
           *off = (double *)malloc(sizeof(double) * (size_t)n);
    c[0] = c[n - 1] = 0.;
    const int m = n - 2; /* interior unknowns c[1..n-2] */
    for (int i = 0; i < m; i++) {
        const double h_i = x[i + 1] - x[i], h_ip1 = x[i + 2] - x[i + 1];
        const double ydiff_i = y[i + 1] - y[i], ydiff_ip1 = y[i + 2] - y[i + 1];
        off[i] = h_ip1;
        diag[i] = 2.0 * (h_ip1 + h_i);
        g[i] = 3.0 * (ydiff_ip1 / h_ip1 - ydiff_i / h_i);
    }
    /* symmetric tridiagonal system: diag on the diagonal, off[i] between rows i and i+1 */
    for (int i = 1; i < m; i++) {
        const double w = off[i - 1] / diag[i - 1];
        diag[i] -= w * off[i - 1];
        g[i] -= w * g[i - 1];
    }
    for (int i = m - 1; i >= 0; i--) {
        double v = g[i];
        if (i + 1 < m) v -= off[i] * c[i + 2];
        c[i + 1] = v / diag[i];
    }
    free(g);
    free(diag);
    free(off);
}

static double cspline_eval(int n, const double *x, const double *y, const double *c, double v) {
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

/* ---- MHR00 parameters ---------------------------------------------------------------------- */
#define A_NPTS 60
#define C_NPTS 12
#define B_NPTS 5
static struct {
    int ready;
    double Ax[A_NPTS], Ay[A_NPTS], Ac[A_NPTS];
    double Cx[C_NPTS], Cy[C_NPTS], Cc[C_NPTS];
    double Bx[B_NPTS], By[B_NPTS], Bc[B_NPTS];
    double *rr_y, *rr_c;
    float y_he; /* the table depends on Y_He, OMb, hlittle through No and corr_He */
    float omb, h;
} M;

static double C_MHR(double z) {
    if (z >= 13.0) return 1.0;
    if (z <= 2.0) return 0.558;
    return cspline_eval(C_NPTS, M.Cx, M.Cy, M.Cc, z);
}
static double beta_MHR(double z) {
    if (z >= 6.0) return -2.50;
    if (z <= 2.0) return -2.23;
    return cspline_eval(B_NPTS, M.Bx, M.By, M.Bc, z);
}
static double A_MHR(double z) {
    if (z >= 2.0 + (float)A_NPTS) z = 2.0 + (float)A_NPTS;
    if (z <= 2.0) z = 2.0;
    /* (gsl_spline_eval beyond the last knot extrapolates the last interval; z = 62 is one past
     * the last knot 61: the same polynomial is used here) */
    if (z > M.Ax[A_NPTS - 1]) {
        const int lo = A_NPTS - 2;
        const double dx = M.Ax[lo + 1] - M.Ax[lo], dy = M.Ay[lo + 1] - M.Ay[lo];
        const double b = dy / dx - dx * (M.Ac[lo + 1] + 2.0 * M.Ac[lo]) / 3.0;
        const double d = (M.Ac[lo + 1] - M.Ac[lo]) / (3.0 * dx);
        const double t = z - M.Ax[lo];
        return M.Ay[lo] + t * (b + t * (M.Ac[lo] + t * d));
    }
    return cspline_eval(A_NPTS, M.Ax, M.Ay, M.Ac, z);
}

struct aux_ctx {
    double C, beta, sig;
};
/* aux_function (:217-227) in ln Delta: P(Delta) dDelta = P Delta dlnDelta */
static double aux_lnD(double lnD, void *p) {
    const struct aux_ctx *a = (const struct aux_ctx *)p;
    const double del = exp(lnD), u = pow(del, -2.0 / 3.0) - a->C;
    return exp(-u * u / (2.0 * a->sig * a->sig)) * pow(del, a->beta) * del;
}

/* ---- thermochem.c:78-110 ---------------------------------------------------------------------- */
static double alpha_B(double T) { return ALPHA_B_10K * pow(T / 1.0e4, -0.75); }

static double neutral_fraction_caseB(double density, double T4, double gamma) {
    const double corr_He = 1.0 / (4.0 / cosmo_params_global->Y_He - 3);
    const double alpha = alpha_B(T4 * 1e4);
    gamma *= 1e-12;
    double chi = (1 + corr_He) * density * alpha / gamma;
    if (chi < TINY) return 0;
    if (chi < 1e-5) return chi;
    const double b = -2 - gamma / (density * (1 + corr_He) * alpha);
    return (-b - sqrt(b * b - 4)) / 2.0;
}

static double Gamma_SS(double Gamma_bg, double del, double T_4, double z) {
    const double D_ss = 26.7 * pow(T_4, 0.17) * pow((1 + z) / 10.0, -3) * pow(Gamma_bg, 2.0 / 3.0);
    return Gamma_bg *
           (0.98 * pow((1.0 + pow(del / D_ss, 1.64)), -2.28) + 0.02 * pow(1.0 + del / D_ss, -0.84));
}

struct rr_ctx {
    double z, gamma12_bg, T4, A, C_0, beta, avenH;
};
static double MHR_rr(double lnD, void *params) {
    const struct rr_ctx *p = (const struct rr_ctx *)params;
    const double del = exp(lnD);
    const double gamma = Gamma_SS(p->gamma12_bg, del, p->T4, p->z);
    const double n_H = p->avenH * del;
    const double x_e = 1.0 - neutral_fraction_caseB(n_H, p->T4, gamma);
    const double u = (pow(del, -2.0 / 3.0) - p->C_0) / ((2.0 * 7.61 / (3.0 * (1.0 + p->z))));
    const double PDelta = p->A * exp(-0.5 * u * u) * pow(del, p->beta);
    return 1e15 * n_H * PDelta * alpha_B(p->T4 * 1e4) * x_e * x_e * del * del;
}

static double No_today(void) { /* Constants.h:98-101 */
    const double Ho = c21_hubble0();
    const double rhocrit_cgs = 3.0 * Ho * Ho / (8.0 * M_PI * PC_G);
    return rhocrit_cgs * cosmo_params_global->OMb * (1 - cosmo_params_global->Y_He) / PC_M_P;
}

/* 96-point Gauss-Legendre rule on [ln 0.01, ln 200]: the integrand is smooth in ln Delta (the
 * self-shielding transition spans a decade), and 75 000 table entries want a fixed rule; the
 * rule agrees with the adaptive integrator to < 1e-7 (tests/test_host_scalars.py) */
#define RR_NGL 96
static double gl_x[RR_NGL], gl_w[RR_NGL];
static void gl_init(void) {
    static int done;
    if (done) return;
    const double a = log(0.01), b = log(200);
    const double xm = 0.5 * (b + a), xl = 0.5 * (b - a);
    for (int i = 0; i < (RR_NGL + 1) / 2; i++) {
        double zz = cos(M_PI * (i + 0.75) / (RR_NGL + 0.5)), pp, z1;
        do { /* Newton on P_n */
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= RR_NGL; j++) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * zz * p2 - (j - 1.0) * p3) / j;
            }
            pp = RR_NGL * (zz * p1 - p2) / (zz * zz - 1.0);
            z1 = zz;
            zz = z1 - p1 / pp;
        } while (fabs(zz - z1) > 1e-15);
        gl_x[i] = xm - xl * zz;
        gl_x[RR_NGL - 1 - i] = xm + xl * zz;
        gl_w[i] = gl_w[RR_NGL - 1 - i] = 2.0 * xl / ((1.0 - zz * zz) * pp * pp);
    }
    done = 1;
}

/* recombinations.c:181-215, T4 = 1, case B */
double c21_recombination_rate(double z, double gamma12_bg) {
    struct rr_ctx p = {z, gamma12_bg, 1.0, A_MHR(z), C_MHR(z), beta_MHR(z),
                       No_today() * pow(1 + z, 3)};
    gl_init();
    double sum = 0.;
    for (int i = 0; i < RR_NGL; i++) sum += gl_w[i] * MHR_rr(gl_x[i], &p);
    return sum;
}

/* the same integral with the adaptive rule (test hook) */
double c21_recombination_rate_adaptive(double z, double gamma12_bg) {
    struct rr_ctx p = {z, gamma12_bg, 1.0, A_MHR(z), C_MHR(z), beta_MHR(z),
                       No_today() * pow(1 + z, 3)};
    return c21_integrate(MHR_rr, &p, log(0.01), log(200), 1e-9);
}

static void mhr_params_init(void) {
    for (int i = 0; i < C_NPTS; i++) M.Cx[i] = (float)i + 2.0;
    static const double ct[C_NPTS] = {0.558, 0.599, 0.611, 0.769, 0.868, 0.930,
                                      0.964, 0.983, 0.993, 0.998, 0.999, 1.00};
    memcpy(M.Cy, ct, sizeof(ct));
    cspline_c(C_NPTS, M.Cx, M.Cy, M.Cc);
    for (int i = 0; i < B_NPTS; i++) M.Bx[i] = (float)i + 2.0;
    static const double bt[B_NPTS] = {-2.23, -2.35, -2.48, -2.49, -2.50};
    memcpy(M.By, bt, sizeof(bt));
    cspline_c(B_NPTS, M.Bx, M.By, M.Bc);
    for (int i = 0; i < A_NPTS; i++) { /* init_A_MHR :270-283 */
        const double z = 2.0 + (float)i;
        struct aux_ctx a = {C_MHR(z), beta_MHR(z), 2.0 * 7.61 / (3.0 * (1.0 + z))};
        /* the reference integrates over Delta in [1e-25, 1e25]; the integrand is below 1e-300
         * outside ln Delta in [-12, 58] */
        M.Ax[i] = z;
        M.Ay[i] = 1.0 / c21_integrate(aux_lnD, &a, -12.0, 58.0, 1e-9);
    }
    cspline_c(A_NPTS, M.Ax, M.Ay, M.Ac);
}

/* The [RR_NZ][RR_NGAMMA] table and its spline coefficients; built once per (Y_He, OMb, h). */
int c21_rr_tables(const double **y_out, const double **c_out) {
    if (!cosmo_params_global) {
        c21hip_set_error("init_MHR: Broadcast_struct_global_all has not been called");
        return C21CM_VALUE_ERROR;
    }
    const CosmoParams *cp = cosmo_params_global;
    if (!(M.ready && M.y_he == cp->Y_He && M.omb == cp->OMb && M.h == cp->hlittle)) {
        mhr_params_init();
        const size_t n = (size_t)C21CM_RR_NZ * C21CM_RR_NGAMMA;
        if (!M.rr_y) M.rr_y = (double *)malloc(sizeof(double) * n);
        if (!M.rr_c) M.rr_c = (double *)malloc(sizeof(double) * n);
        if (!M.rr_y || !M.rr_c) return C21CM_MEMORY_ALLOC_ERROR;
        double lnG[C21CM_RR_NGAMMA];
        for (int g = 0; g < C21CM_RR_NGAMMA; g++) lnG[g] = C21CM_RR_LNGAMMA_MIN + g * C21CM_RR_DLNGAMMA;
        for (int z_ct = 0; z_ct < C21CM_RR_NZ; z_ct++) {
            const float z = z_ct * C21CM_RR_DZ; /* a float upstream (:100) */
            double *row = M.rr_y + (size_t)z_ct * C21CM_RR_NGAMMA;
            for (int g = 0; g < C21CM_RR_NGAMMA; g++) {
                const float gamma = (float)exp(lnG[g]); /* float gamma (:96,106) */
                row[g] = c21_recombination_rate(z, gamma);
            }
            cspline_c(C21CM_RR_NGAMMA, lnG, row, M.rr_c + (size_t)z_ct * C21CM_RR_NGAMMA);
        }
        M.y_he = cp->Y_He;
        M.omb = cp->OMb;
        M.h = cp->hlittle;
        M.ready = 1;
    }
    if (y_out) *y_out = M.rr_y;
    if (c_out) *c_out = M.rr_c;
    return 0;
}

/* exported with the reference's names (_functionprototypes_wrapper.h: init_MHR / free_MHR) */
void init_MHR(void) { (void)c21_rr_tables(NULL, NULL); }
void free_MHR(void) { /* the tables are kept: they depend on three cosmological parameters only */ }

/* splined_recombination_rate (recombinations.c:64-92) on given tables: the homogeneous model's
 * one evaluation per call happens on the host */
double c21_rr_eval(const double *y, const double *c, double z_eff, double gamma12_bg) {
    int z_ct = z_eff > 0 ? (int)(fmin(z_eff, 1e6) / C21CM_RR_DZ + 0.5) : 0; /* NaN -> row 0 */
    double lnGamma = log(gamma12_bg);
    if (z_ct < 0) z_ct = 0;
    if (z_ct >= C21CM_RR_NZ) z_ct = C21CM_RR_NZ - 1;
    const double top = C21CM_RR_LNGAMMA_MIN + C21CM_RR_DLNGAMMA * (C21CM_RR_NGAMMA - 1);
    if (lnGamma < C21CM_RR_LNGAMMA_MIN) return 0;
    if (lnGamma >= top) lnGamma = top - 1e-7;
    double lnG[C21CM_RR_NGAMMA];
    for (int g = 0; g < C21CM_RR_NGAMMA; g++) lnG[g] = C21CM_RR_LNGAMMA_MIN + g * C21CM_RR_DLNGAMMA;
    return cspline_eval(C21CM_RR_NGAMMA, lnG, y + (size_t)z_ct * C21CM_RR_NGAMMA,
                        c + (size_t)z_ct * C21CM_RR_NGAMMA, lnGamma);
}

double c21_splined_recombination_rate(double z_eff, double gamma12_bg) {
    const double *y, *c;
    if (c21_rr_tables(&y, &c)) return NAN;
    return c21_rr_eval(y, c, z_eff, gamma12_bg);
}
