/*
 * heating.c -- host scalars and tables of the spin-temperature calculation: everything
 * ComputeTsBox prepares before (and between) its cell loops.
 *
 * reference: src/py21cmfast/src/heating_helper_progs.c
 *   :58-90      init_heat / destruct_heat
 *   :195-264    frecycle                    :268-353  spectral_emissivity (stellar_spectra.dat)
 *   :356-362    nu_n                        :1193-1198 zmax
 *   :767-829    the three frequency integrands
 *   :831-858    integrate_over_nu           (gsl_integration_qag, 15-point rule, epsrel 1e-2)
 *   :862-872    species_weighted_x_ray_cross_section
 *   :943-1059   tauX_integrand, tauX        (qag, 15-point rule, epsrel 5e-3)
 *   :1135-1190  nu_tau_one                  (gsl_root_fsolver_brent, interval test at 2 %)
 *   :1316-1365  Energy_Lya_heating          (Lyman_alpha_heating_table.dat)
 * src/py21cmfast/src/elec_interp.c:39-423 (x_int tables, bilinear float lookups),
 * src/py21cmfast/src/thermochem.c:104-146 (photo-ionisation cross sections),
 * src/py21cmfast/src/SpinTemperatureBox.c:312-361 setup_z_edges, :364-499
 * calculate_spectral_factors, :810-889 fill_freqint_tables, :930-1008 global_reion_properties,
 * :1098-1184 set_zp_consts, and src/py21cmfast/src/interp_tables.c:96-232,889-944 (the N_ion(z) and
 * SFRD(z) tables: 400 points, linear interpolation).
 *
 * GSL is a third-party dependency of the reference that is absent here; the two
 * GSL routines the frequency integrals depend on at their loose tolerances are restated from
 * their published algorithms so that the SAME subdivisions and iterates are produced:
 *   gsl_integration_qag = QUADPACK QAG (Piessens et al. 1983): bisect the sub-interval with the
 *     largest error estimate until sum(errors) <= epsrel |sum(results)|; 15-point Gauss-Kronrod
 *     rule with QUADPACK's error rescaling (200 err / resasc)^1.5;
 *   gsl_root_fsolver_brent = Brent's zero (1973) in GSL's bookkeeping, stopped by
 *     gsl_root_test_interval(x_lo, x_hi, 0, 0.02).
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"
#include "cosmology.h"
#include "heating.h"

/* Constants.c:4-45 */
#define PC_C_CMS 2.99792458e10
#define PC_H_P 6.62607015e-27
#define PC_K_B 1.380649e-16
#define PC_M_P 1.67262192369e-24
#define PC_M_E 9.1093837015e-28
#define PC_E_CHARGE 4.803204712570263e-10
#define PC_VAC_PERM 8.8541878128e-12
#define PC_MSUN 1.989e33
#define PC_S_PER_YR 31556925.9747
#define PC_CM_PER_MPC 3.08567758e24
#define PC_EV_TO_HZ 2.417989e14
#define PC_NU_ION_HI 3.288465e15
#define PC_NU_LW_THRESH 2.70331197e15 /* Constants.c:23 */
#define PC_NU_ION_HEI 5.945836e15
#define PC_NU_ION_HEII 1.3153862e16
#define PC_NU_LY_ALPHA 2.46606727e15
#define PC_T_CMB 2.7255
#define PC_T_21 0.0682
#define PC_LAMBDA_21 21.106114054160
#define PC_LAMBDA_LY_ALPHA 1215.67
#define PC_A10 2.85e-15
#define PC_F_ALPHA 0.4162
#define PC_L_FACTOR 0.620350491
#define TINY 1e-30
#define FRACT_FLOAT_ERR 1e-7
#define NSPEC_MAX 23
#define X_INT_NENERGY 258
#define M_MAX_INTEGRAL 1e16
#define ZPP_INTERP_POINTS 400

/* ================================================================ QUADPACK QAG, 15-point rule */
static const double xgk15[8] = {0.991455371120812639206854697526329, 0.949107912342758524526189684047851,
                                0.864864423359769072789712788640926, 0.741531185599394439863864773280788,
                                0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
                                0.207784955007898467600689403773245, 0.000000000000000000000000000000000};
static const double wgk15[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
static const double wg7[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                              0.381830050505118944950369775488975, 0.417959183673469387755102040816327};

static double rescale_error(double err, double result_abs, double result_asc) {
    err = fabs(err);
    if (result_asc != 0 && err != 0) {
        const double scale = pow((200 * err / result_asc), 1.5);
        err = scale < 1 ? result_asc * scale : result_asc;
    }
    if (result_abs > DBL_MIN / (50 * DBL_EPSILON)) {
        const double min_err = 50 * DBL_EPSILON * result_abs;
        if (min_err > err) err = min_err;
    }
    return err;
}

#include "gk61_tables.h" /* xgk61, wgk61, wg30: the 61-point rule (GSL_INTEG_GAUSS61) */

/* gsl_integration_qk: n = number of Kronrod abscissae on [0, 1] including the centre (8 for the
 * 15-point rule, 31 for the 61-point rule) */
static void qk_rule(int n, const double *xgk, const double *wg, const double *wgk, c21_fn f, void *ctx,
                    double a, double b, double *result, double *abserr, double *resabs,
                    double *resasc) {
    double fv1[31], fv2[31];
    const double center = 0.5 * (a + b), half_length = 0.5 * (b - a);
    const double abs_half_length = fabs(half_length);
    const double f_center = f(center, ctx);
    double result_gauss = (n % 2 == 0) ? f_center * wg[n / 2 - 1] : 0.;
    double result_kronrod = f_center * wgk[n - 1];
    double result_abs = fabs(result_kronrod);
    for (int j = 0; j < (n - 1) / 2; j++) {
        const int jtw = j * 2 + 1;
        const double abscissa = half_length * xgk[jtw];
        const double fval1 = f(center - abscissa, ctx), fval2 = f(center + abscissa, ctx);
        const double fsum = fval1 + fval2;
        fv1[jtw] = fval1;
        fv2[jtw] = fval2;
        result_gauss += wg[j] * fsum;
        result_kronrod += wgk[jtw] * fsum;
        result_abs += wgk[jtw] * (fabs(fval1) + fabs(fval2));
    }
    for (int j = 0; j < n / 2; j++) {
        const int jtwm1 = j * 2;
        const double abscissa = half_length * xgk[jtwm1];
        const double fval1 = f(center - abscissa, ctx), fval2 = f(center + abscissa, ctx);
        fv1[jtwm1] = fval1;
        fv2[jtwm1] = fval2;
        result_kronrod += wgk[jtwm1] * (fval1 + fval2);
        result_abs += wgk[jtwm1] * (fabs(fval1) + fabs(fval2));
    }
    const double mean = result_kronrod * 0.5;
    double result_asc = wgk[n - 1] * fabs(f_center - mean);
    for (int j = 0; j < n - 1; j++)
        result_asc += wgk[j] * (fabs(fv1[j] - mean) + fabs(fv2[j] - mean));
    const double err = (result_kronrod - result_gauss) * half_length;
    result_kronrod *= half_length;
    result_abs *= abs_half_length;
    result_asc *= abs_half_length;
    *result = result_kronrod;
    *resabs = result_abs;
    *resasc = result_asc;
    *abserr = rescale_error(err, result_abs, result_asc);
}

static _Thread_local int qag_key61; /* rule of the running c21_qag call */
static void qk15(c21_fn f, void *ctx, double a, double b, double *result, double *abserr,
                 double *resabs, double *resasc) {
    if (qag_key61)
        qk_rule(31, xgk61, wg30, wgk61, f, ctx, a, b, result, abserr, resabs, resasc);
    else
        qk_rule(8, xgk15, wg7, wgk15, f, ctx, a, b, result, abserr, resabs, resasc);
}

#define QAG_LIMIT 1000
static double qag_run(c21_fn f, void *ctx, double a, double b, double epsrel, double *abserr_out,
                      int *status_out);
double c21_qag15(c21_fn f, void *ctx, double a, double b, double epsrel, double *abserr_out,
                 int *status_out) {
    const int saved = qag_key61;
    qag_key61 = 0;
    const double r = qag_run(f, ctx, a, b, epsrel, abserr_out, status_out);
    qag_key61 = saved;
    return r;
}
/* gsl_integration_qag(..., epsabs 0, epsrel, limit 1000, GSL_INTEG_GAUSS61, ...) */
double c21_qag61(c21_fn f, void *ctx, double a, double b, double epsrel, double *abserr_out,
                 int *status_out) {
    const int saved = qag_key61;
    qag_key61 = 1;
    const double r = qag_run(f, ctx, a, b, epsrel, abserr_out, status_out);
    qag_key61 = saved;
    return r;
}
static double qag_run(c21_fn f, void *ctx, double a, double b, double epsrel, double *abserr_out,
                      int *status_out) {
    static _Thread_local double al[QAG_LIMIT], bl[QAG_LIMIT], rl[QAG_LIMIT], el[QAG_LIMIT];
    double result0, abserr0, resabs0, resasc0;
    int status = 0;
    qk15(f, ctx, a, b, &result0, &abserr0, &resabs0, &resasc0);
    al[0] = a, bl[0] = b, rl[0] = result0, el[0] = abserr0;
    int size = 1;
    double tolerance = epsrel * fabs(result0);
    const double round_off = 50 * DBL_EPSILON * resabs0;
    double area = result0, errsum = abserr0;
    if (abserr0 <= round_off && abserr0 > tolerance) {
        status = 1; /* GSL_EROUND on the first attempt */
    } else if ((abserr0 <= tolerance && abserr0 != resasc0) || abserr0 == 0.0) {
        /* converged on the first rule */
    } else {
        int roundoff_type1 = 0, roundoff_type2 = 0, iteration = 1;
        do {
            int i_max = 0; /* the sub-interval with the largest error estimate */
            for (int i = 1; i < size; i++)
                if (el[i] > el[i_max]) i_max = i;
            const double a_i = al[i_max], b_i = bl[i_max], r_i = rl[i_max], e_i = el[i_max];
            const double a1 = a_i, b1 = 0.5 * (a_i + b_i), a2 = b1, b2 = b_i;
            double area1, area2, error1, error2, resabs1, resabs2, resasc1, resasc2;
            qk15(f, ctx, a1, b1, &area1, &error1, &resabs1, &resasc1);
            qk15(f, ctx, a2, b2, &area2, &error2, &resabs2, &resasc2);
            const double area12 = area1 + area2, error12 = error1 + error2;
            errsum += (error12 - e_i);
            area += area12 - r_i;
            if (resasc1 != error1 && resasc2 != error2) {
                const double delta = r_i - area12;
                if (fabs(delta) <= 1.0e-5 * fabs(area12) && error12 >= 0.99 * e_i) roundoff_type1++;
                if (iteration >= 10 && error12 > e_i) roundoff_type2++;
            }
            tolerance = epsrel * fabs(area);
            if (errsum > tolerance) {
                if (roundoff_type1 >= 6 || roundoff_type2 >= 20) status = 2;
                const double tmp = (1 + 100 * DBL_EPSILON) * (fabs(a2) + 1000 * DBL_MIN);
                if (fabs(a1) <= tmp && fabs(b2) <= tmp) status = 3;
            }
            /* the larger error keeps the slot, the other half is appended (workspace update) */
            if (error2 > error1) {
                al[i_max] = a2, bl[i_max] = b2, rl[i_max] = area2, el[i_max] = error2;
                al[size] = a1, bl[size] = b1, rl[size] = area1, el[size] = error1;
            } else {
                al[i_max] = a1, bl[i_max] = b1, rl[i_max] = area1, el[i_max] = error1;
                al[size] = a2, bl[size] = b2, rl[size] = area2, el[size] = error2;
            }
            size++;
            iteration++;
        } while (iteration < QAG_LIMIT && !status && errsum > tolerance);
        if (!status && errsum > tolerance) status = 4; /* GSL_EMAXITER */
    }
    double result = 0;
    for (int i = 0; i < size; i++) result += rl[i];
    if (abserr_out) *abserr_out = errsum;
    if (status_out) *status_out = status;
    return result;
}

/* ================================================================ Brent's zero, GSL bookkeeping */
double c21_brent_root(c21_fn f, void *ctx, double x_lower, double x_upper, double epsrel,
                      int max_iter, int *status_out) {
    double a = x_lower, b = x_upper, fa = f(a, ctx), fb = f(b, ctx);
    double c = x_upper, fc = fb, d = x_upper - x_lower, e = x_upper - x_lower;
    double root = 0.5 * (x_lower + x_upper), x_lo = x_lower, x_hi = x_upper;
    int status = 0;
    if ((fa < 0.0 && fb < 0.0) || (fa > 0.0 && fb > 0.0)) {
        if (status_out) *status_out = 1; /* endpoints do not straddle y = 0 */
        return root;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        int ac_equal = 0;
        if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) {
            ac_equal = 1;
            c = a, fc = fa, d = b - a, e = b - a;
        }
        if (fabs(fc) < fabs(fb)) {
            ac_equal = 1;
            a = b, b = c, c = a;
            fa = fb, fb = fc, fc = fa;
        }
        const double tol = 0.5 * DBL_EPSILON * fabs(b), m = 0.5 * (c - b);
        if (fb == 0) {
            root = x_lo = x_hi = b;
        } else if (fabs(m) <= tol) {
            root = b;
            x_lo = b < c ? b : c;
            x_hi = b < c ? c : b;
        } else {
            if (fabs(e) < tol || fabs(fa) <= fabs(fb)) {
                d = m, e = m; /* bisection */
            } else {
                double p, q;
                const double s = fb / fa;
                if (ac_equal) {
                    p = 2 * m * s;
                    q = 1 - s;
                } else {
                    const double qq = fa / fc, r = fb / fc;
                    p = s * (2 * m * qq * (qq - r) - (b - a) * (r - 1));
                    q = (qq - 1) * (r - 1) * (s - 1);
                }
                if (p > 0)
                    q = -q;
                else
                    p = -p;
                const double lim1 = 3 * m * q - fabs(tol * q), lim2 = fabs(e * q);
                if (2 * p < (lim1 < lim2 ? lim1 : lim2)) {
                    e = d;
                    d = p / q;
                } else {
                    d = m, e = m; /* interpolation failed */
                }
            }
            a = b, fa = fb;
            if (fabs(d) > tol)
                b += d;
            else
                b += (m > 0 ? +tol : -tol);
            fb = f(b, ctx);
            root = b;
            if ((fb < 0 && fc < 0) || (fb > 0 && fc > 0)) c = a;
            x_lo = b < c ? b : c;
            x_hi = b < c ? c : b;
        }
        /* gsl_root_test_interval(x_lo, x_hi, 0, epsrel) */
        const double abs_lo = fabs(x_lo), abs_hi = fabs(x_hi);
        double min_abs;
        if ((x_lo > 0.0 && x_hi > 0.0) || (x_lo < 0.0 && x_hi < 0.0))
            min_abs = abs_lo < abs_hi ? abs_lo : abs_hi;
        else
            min_abs = 0;
        if (fabs(x_hi - x_lo) < epsrel * min_abs) break;
        if (iter == max_iter - 1) status = 2;
    }
    if (status_out) *status_out = status;
    return root;
}

/* ================================================================ data tables */
static struct {
    int loaded;
    char path[600];
    /* spectral_emissivity */
    int n[NSPEC_MAX];
    float nu_n[NSPEC_MAX], alpha_S_2[NSPEC_MAX], alpha_S_3[NSPEC_MAX], N0_2[NSPEC_MAX], N0_3[NSPEC_MAX];
    float pop2_ion, pop3_ion; /* the normalisation is baked in at load time (:342-347) */
    /* elec_interp */
    float x_int_XHII[C21CM_X_INT_NXHII];
    float x_int_Energy[X_INT_NENERGY];
    float x_int_fheat[C21CM_X_INT_NXHII][X_INT_NENERGY];
    float x_int_n_Lya[C21CM_X_INT_NXHII][X_INT_NENERGY];
    float x_int_nion_HI[C21CM_X_INT_NXHII][X_INT_NENERGY];
    float x_int_nion_HeI[C21CM_X_INT_NXHII][X_INT_NENERGY];
    float x_int_nion_HeII[C21CM_X_INT_NXHII][X_INT_NENERGY];
    /* Lyman-alpha heating */
    int lya_loaded;
    double *dEC, *dEI;
} H;

static const float kXHII[C21CM_X_INT_NXHII] = C21CM_X_INT_XHII;

static int skipline(FILE *fl, int n) {
    for (int i = 0; i < n; i++) {
        int ch;
        do {
            ch = fgetc(fl);
            if (ch == EOF) return 1;
        } while (ch != '\n');
    }
    return 0;
}

static int load_x_int(const char *dir) {
    char name[800];
    memcpy(H.x_int_XHII, kXHII, sizeof(kXHII));
    for (int n_ion = 0; n_ion < C21CM_X_INT_NXHII; n_ion++) {
        if (H.x_int_XHII[n_ion] < 0.3)
            snprintf(name, sizeof(name), "%s/x_int_tables/log_xi_%1.1f.dat", dir, log10(H.x_int_XHII[n_ion]));
        else
            snprintf(name, sizeof(name), "%s/x_int_tables/xi_%1.3f.dat", dir, H.x_int_XHII[n_ion]);
        FILE *F = fopen(name, "r");
        if (!F) {
            c21hip_set_error("init_heat: can't open input file %s", name);
            return C21CM_IO_ERROR;
        }
        float xHI, xHeI, xHeII, z, T, trash;
        int bad = skipline(F, 1);
        bad |= fscanf(F, "%g %g %g %g %g", &xHI, &xHeI, &xHeII, &z, &T) != 5;
        bad |= skipline(F, 2);
        for (int i = 0; i < X_INT_NENERGY && !bad; i++)
            bad |= fscanf(F, "%g %g %g %g %g %g %g %g %g", &H.x_int_Energy[i], &trash,
                          &H.x_int_fheat[n_ion][i], &trash, &H.x_int_n_Lya[n_ion][i],
                          &H.x_int_nion_HI[n_ion][i], &H.x_int_nion_HeI[n_ion][i],
                          &H.x_int_nion_HeII[n_ion][i], &trash) != 9;
        fclose(F);
        if (bad) {
            c21hip_set_error("init_heat: %s is not an x_int table of %d energies", name, X_INT_NENERGY);
            return C21CM_IO_ERROR;
        }
    }
    return 0;
}

static int load_spectra(const char *dir) {
    char name[800];
    snprintf(name, sizeof(name), "%s/stellar_spectra.dat", dir);
    FILE *F = fopen(name, "r");
    if (!F) {
        c21hip_set_error("spectral_emissivity: unable to open %s", name);
        return C21CM_IO_ERROR;
    }
    int bad = 0;
    for (int i = 1; i < NSPEC_MAX && !bad; i++)
        bad |= fscanf(F, "%i %e %e %e %e", &H.n[i], &H.N0_2[i], &H.alpha_S_2[i], &H.N0_3[i],
                      &H.alpha_S_3[i]) != 5;
    fclose(F);
    if (bad) {
        c21hip_set_error("spectral_emissivity: %s does not hold %d spectral lines", name, NSPEC_MAX - 1);
        return C21CM_IO_ERROR;
    }
    for (int i = 1; i < NSPEC_MAX; i++) H.nu_n[i] = 4.0 / 3.0 * (1.0 - 1.0 / pow(H.n[i], 2.0));
    H.pop2_ion = astro_params_global->POP2_ION;
    H.pop3_ion = astro_params_global->POP3_ION;
    for (int i = 1; i < (NSPEC_MAX - 1); i++) {
        double n0_fac = (pow(H.nu_n[i + 1], H.alpha_S_2[i] + 1) - pow(H.nu_n[i], H.alpha_S_2[i] + 1));
        H.N0_2[i] *= (H.alpha_S_2[i] + 1) / n0_fac * astro_params_global->POP2_ION;
        n0_fac = (pow(H.nu_n[i + 1], H.alpha_S_3[i] + 1) - pow(H.nu_n[i], H.alpha_S_3[i] + 1));
        H.N0_3[i] *= (H.alpha_S_3[i] + 1) / n0_fac * astro_params_global->POP3_ION;
    }
    return 0;
}

static int load_lya_heating(const char *dir) {
    char name[800];
    snprintf(name, sizeof(name), "%s/Lyman_alpha_heating_table.dat", dir);
    FILE *F = fopen(name, "r");
    if (!F) {
        c21hip_set_error("Energy_Lya_heating: unable to open %s (USE_LYA_HEATING needs it)", name);
        return C21CM_IO_ERROR;
    }
    const size_t n = (size_t)C21CM_LYA_NT * C21CM_LYA_NT * C21CM_LYA_NGP;
    if (!H.dEC) H.dEC = (double *)malloc(n * sizeof(double));
    if (!H.dEI) H.dEI = (double *)malloc(n * sizeof(double));
    if (!H.dEC || !H.dEI) {
        fclose(F);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    int bad = 0;
    for (size_t i = 0; i < n && !bad; i++) bad |= fscanf(F, "%lf %lf", &H.dEC[i], &H.dEI[i]) != 2;
    fclose(F);
    if (bad) {
        c21hip_set_error("Energy_Lya_heating: %s is shorter than %zu rows", name, n);
        return C21CM_IO_ERROR;
    }
    H.lya_loaded = 1;
    return 0;
}

int c21_heat_load(void) {
    if (!astro_params_global || !astro_options_global) {
        c21hip_set_error("init_heat: Broadcast_struct_global_all has not been called");
        return C21CM_VALUE_ERROR;
    }
    const char *dir = config_settings.external_table_path;
    if (!dir) {
        c21hip_set_error("init_heat: config_settings.external_table_path is not set");
        return C21CM_IO_ERROR;
    }
    int st;
    const int same = H.loaded && strcmp(dir, H.path) == 0 &&
                     H.pop2_ion == astro_params_global->POP2_ION &&
                     H.pop3_ion == astro_params_global->POP3_ION;
    if (!same) {
        H.loaded = 0;
        H.lya_loaded = 0;
        if ((st = c21_recfast_load())) return st;
        if ((st = load_spectra(dir))) return st;
        if ((st = load_x_int(dir))) return st;
        snprintf(H.path, sizeof(H.path), "%s", dir);
        H.loaded = 1;
    }
    if (astro_options_global->USE_LYA_HEATING && !H.lya_loaded)
        if ((st = load_lya_heating(dir))) return st;
    return 0;
}

/* the reference's names (_functionprototypes_wrapper.h): init_heat returns 0 or a negative code */
int init_heat(void) { return c21_heat_load() ? -1 : 0; }
void destruct_heat(void) {}

const double *c21_lya_table(int which) { return which ? H.dEI : H.dEC; }

/* ================================================================ spectra */
double c21_frecycle(int n) { /* :195-264 (Pritchard & Furlanetto 2006) */
    static const double v[31] = {1, 1, 1, 0, 0.2609, 0.3078, 0.3259, 0.3353, 0.3410, 0.3448, 0.3476,
                                 0.3496, 0.3512, 0.3524, 0.3535, 0.3543, 0.3550, 0.3556, 0.3561,
                                 0.3565, 0.3569, 0.3572, 0.3575, 0.3578, 0.3580, 0.3582, 0.3584,
                                 0.3586, 0.3587, 0.3589, 0.3590};
    return (n >= 0 && n <= 30) ? v[n] : 0;
}

double c21_nu_n(int n) { return (1.0 - pow(n, -2.0)) / 0.75; }

float c21_zmax(float z, int n) {
    const double num = 1 - pow(n + 1, -2), denom = 1 - pow(n, -2);
    return (1 + z) * num / denom - 1;
}

/* spectral_emissivity(nu_norm, 0, Population) :332-352 */
double c21_spectral_emissivity(double nu_norm, int pop) {
    const float *N0 = pop == 2 ? H.N0_2 : H.N0_3, *al = pop == 2 ? H.alpha_S_2 : H.alpha_S_3;
    for (int i = 1; i < (NSPEC_MAX - 1); i++)
        if ((nu_norm >= H.nu_n[i]) && (nu_norm < H.nu_n[i + 1]))
            return N0[i] * pow(nu_norm, al[i]) / PC_NU_LY_ALPHA;
    const int i = NSPEC_MAX - 1;
    return N0[i] * pow(nu_norm, al[i]) / PC_NU_LY_ALPHA;
}

/* spectral_emissivity(nu_norm, 2, Population) :284-302: photons between nu_norm and the next
 * Lyman line, (1 - F_H2_SHIELD) applied by the caller; outside the tabulated bands the reference
 * falls through to its (re)initialisation branch and returns 0 */
double c21_spectral_emissivity_lw(double nu_norm, int pop) {
    const float *N0 = pop == 2 ? H.N0_2 : H.N0_3, *al = pop == 2 ? H.alpha_S_2 : H.alpha_S_3;
    for (int i = 1; i < (NSPEC_MAX - 1); i++)
        if ((nu_norm >= H.nu_n[i]) && (nu_norm < H.nu_n[i + 1])) {
            const double result =
                N0[i] / (al[i] + 1) * (pow(H.nu_n[i + 1], al[i] + 1) - pow(nu_norm, al[i] + 1));
            return result > 0 ? result : 1e-40;
        }
    return 0.0;
}

/* ================================================================ elec_interp.c */
static int locate_energy_index(float En) {
    if (En < 1008.88) return (int)(log(En / 10.0) / 1.98026273e-2);
    return 232 + (int)(log(En / 1008.88) / 9.53101798e-2);
}

static int locate_xHII_index(float xHII_call) {
    int m = C21CM_X_INT_NXHII - 1;
    while (xHII_call < H.x_int_XHII[m]) m--;
    return m;
}

/* the five interp_* functions share this body; below_value: what an energy under the table gives */
static float x_int_interp(const float tab[][X_INT_NENERGY], float below_value, float En,
                          float xHII_call) {
    if (En > 0.999 * H.x_int_Energy[X_INT_NENERGY - 1])
        En = H.x_int_Energy[X_INT_NENERGY - 1] * 0.999;
    else if (En < H.x_int_Energy[0])
        return below_value;
    if (xHII_call > H.x_int_XHII[C21CM_X_INT_NXHII - 1] * 0.999)
        xHII_call = H.x_int_XHII[C21CM_X_INT_NXHII - 1] * 0.999;
    else if (xHII_call < H.x_int_XHII[0])
        xHII_call = 1.001 * H.x_int_XHII[0];
    const int n_low = locate_energy_index(En), n_high = n_low + 1;
    const int m_low = locate_xHII_index(xHII_call), m_high = m_low + 1;
    float elow_result, ehigh_result, final_result;
    elow_result = ((tab[m_low][n_high] - tab[m_low][n_low]) / (H.x_int_Energy[n_high] - H.x_int_Energy[n_low]));
    elow_result *= (En - H.x_int_Energy[n_low]);
    elow_result += tab[m_low][n_low];
    ehigh_result = ((tab[m_high][n_high] - tab[m_high][n_low]) / (H.x_int_Energy[n_high] - H.x_int_Energy[n_low]));
    ehigh_result *= (En - H.x_int_Energy[n_low]);
    ehigh_result += tab[m_high][n_low];
    final_result = (ehigh_result - elow_result) / (H.x_int_XHII[m_high] - H.x_int_XHII[m_low]);
    final_result *= (xHII_call - H.x_int_XHII[m_low]);
    final_result += elow_result;
    return final_result;
}

float c21_interp_fheat(float En, float x) { return x_int_interp(H.x_int_fheat, 1.0f, En, x); }
float c21_interp_n_Lya(float En, float x) { return x_int_interp(H.x_int_n_Lya, 0.0f, En, x); }
float c21_interp_nion_HI(float En, float x) { return x_int_interp(H.x_int_nion_HI, 0.0f, En, x); }
float c21_interp_nion_HeI(float En, float x) { return x_int_interp(H.x_int_nion_HeI, 0.0f, En, x); }
float c21_interp_nion_HeII(float En, float x) { return x_int_interp(H.x_int_nion_HeII, 0.0f, En, x); }

/* ================================================================ thermochem.c:104-146 */
double c21_HeI_ion_crosssec(double nu) {
    if (nu < PC_NU_ION_HEI) return 0;
    const double x = nu / PC_EV_TO_HZ / 13.61 - 0.4434;
    const double y = sqrt(x * x + pow(2.136, 2));
    return 9.492e-16 * ((x - 1) * (x - 1) + 2.039 * 2.039) * pow(y, (0.5 * 3.188 - 5.5)) *
           pow(1.0 + sqrt(y / 1.469), -3.188);
}

static double hydrogenic_crosssec(double nu, double nu_ion, double Z) {
    if (nu < nu_ion) return 0;
    if (nu == nu_ion) nu += TINY;
    const double epsilon = sqrt(nu / nu_ion - 1);
    return (6.3e-18) / Z / Z * pow(nu_ion / nu, 4) * exp(4 - (4 * atan(epsilon) / epsilon)) /
           (1 - exp(-2 * M_PI / epsilon));
}
double c21_HI_ion_crosssec(double nu) { return hydrogenic_crosssec(nu, PC_NU_ION_HI, 1); }
double c21_HeII_ion_crosssec(double nu) { return hydrogenic_crosssec(nu, PC_NU_ION_HEII, 2); }

static double h_frac(void) { return (1. - cosmo_params_global->Y_He) / (1. - 3. * cosmo_params_global->Y_He / 4.); }
static double he_frac(void) { return (cosmo_params_global->Y_He / 4.) / (1. - 3. * cosmo_params_global->Y_He / 4.); }
static double number_density_H(void) { /* No, Constants.h:98-101 */
    const double Ho = c21_hubble0();
    const double rhocrit_cgs = 3.0 * Ho * Ho / (8.0 * M_PI * 6.6743e-8);
    return rhocrit_cgs * cosmo_params_global->OMb * (1 - cosmo_params_global->Y_He) / PC_M_P;
}

/* ================================================================ frequency integrals */
static double nu_integrand(double nu, void *params, int flag) {
    const float x_e = *(double *)params; /* rounded to float upstream (:769,790,815) */
    const double thr = astro_params_global->NU_X_THRESH * PC_EV_TO_HZ;
    const double HF = h_frac(), HEF = he_frac();
    double species_sum;
    if (flag == 0) {
        species_sum = c21_interp_fheat((nu - PC_NU_ION_HI) / PC_EV_TO_HZ, x_e) * PC_H_P * (nu - PC_NU_ION_HI) *
                      HF * (1 - x_e) * c21_HI_ion_crosssec(nu);
        species_sum += c21_interp_fheat((nu - PC_NU_ION_HEI) / PC_EV_TO_HZ, x_e) * PC_H_P *
                       (nu - PC_NU_ION_HEI) * HEF * (1 - x_e) * c21_HeI_ion_crosssec(nu);
        species_sum += c21_interp_fheat((nu - PC_NU_ION_HEII) / PC_EV_TO_HZ, x_e) * PC_H_P *
                       (nu - PC_NU_ION_HEII) * HEF * x_e * c21_HeII_ion_crosssec(nu);
    } else if (flag == 1) {
        const double edges[3] = {PC_NU_ION_HI, PC_NU_ION_HEI, PC_NU_ION_HEII};
        double F[3];
        for (int s = 0; s < 3; s++) {
            const float E = (nu - edges[s]) / PC_EV_TO_HZ;
            F[s] = c21_interp_nion_HI(E, x_e) + c21_interp_nion_HeI(E, x_e) + c21_interp_nion_HeII(E, x_e) + 1;
        }
        species_sum = F[0] * HF * (1 - x_e) * c21_HI_ion_crosssec(nu);
        species_sum += F[1] * HEF * (1 - x_e) * c21_HeI_ion_crosssec(nu);
        species_sum += F[2] * HEF * x_e * c21_HeII_ion_crosssec(nu);
    } else {
        species_sum = c21_interp_n_Lya((nu - PC_NU_ION_HI) / PC_EV_TO_HZ, x_e) * HF * (double)(1 - x_e) *
                      c21_HI_ion_crosssec(nu);
        species_sum += c21_interp_n_Lya((nu - PC_NU_ION_HEI) / PC_EV_TO_HZ, x_e) * HEF * (double)(1 - x_e) *
                       c21_HeI_ion_crosssec(nu);
        species_sum += c21_interp_n_Lya((nu - PC_NU_ION_HEII) / PC_EV_TO_HZ, x_e) * HEF * (double)x_e *
                       c21_HeII_ion_crosssec(nu);
    }
    return species_sum * pow(nu / thr, -(astro_params_global->X_RAY_SPEC_INDEX) - 1);
}
static double nu_heat(double nu, void *p) { return nu_integrand(nu, p, 0); }
static double nu_ion(double nu, void *p) { return nu_integrand(nu, p, 1); }
static double nu_lya(double nu, void *p) { return nu_integrand(nu, p, 2); }

double c21_nu_integrand(double nu, double x_e, int flag) { return nu_integrand(nu, &x_e, flag); }

double c21_integrate_over_nu(double zp, double local_x_e, double lower_int_limit, int flag) {
    c21_fn fn = flag == 0 ? nu_heat : (flag == 1 ? nu_ion : nu_lya);
    const double result = c21_qag15(fn, &local_x_e, lower_int_limit,
                                    astro_params_global->NU_X_MAX * PC_EV_TO_HZ, 0.01, NULL, NULL);
    if (flag == 2) return result * PC_C_CMS / (4.0 * M_PI) / PC_NU_LY_ALPHA / c21_hubble(zp);
    return result;
}

double c21_weighted_xray_cross_section(double nu, double x_e) {
    return h_frac() * (1 - x_e) * c21_HI_ion_crosssec(nu) + he_frac() * (1 - x_e) * c21_HeI_ion_crosssec(nu) +
           he_frac() * x_e * c21_HeII_ion_crosssec(nu);
}

/* the host loops below run on N_THREADS threads, as the reference's do */
static int host_threads(void) {
    const int n = simulation_options_global ? simulation_options_global->N_THREADS : 1;
    return n < 1 ? 1 : n;
}

/* ================================================================ N_ion(z), SFRD(z) tables */
#define ZT_MAX 2048
static struct {
    int ready, n;
    double x_min, x_width;
    double nion[ZT_MAX], sfrd[ZT_MAX];
    /* USE_MINI_HALOS: [ZPP_INTERP_POINTS][C21_NMTURN] on the fixed turnover grid */
    double *nion_mini, *sfrd_mini;
    double y_min, y_width;
} zt;

static double minimum_source_mass_xray(double redshift) { /* hmf.c:1319-1348 with xray = true */
    const int mass_dep = matter_options_global->SOURCE_MODEL != C21CM_SOURCE_CONST_ION_EFF;
    const double min_factor = (mass_dep && !astro_options_global->USE_MINI_HALOS) ? 50. : 1.;
    double Mmin;
    if (astro_options_global->USE_MINI_HALOS) {
        Mmin = 1e5;
    } else if (astro_options_global->M_MIN_in_Mass) {
        Mmin = astro_params_global->M_TURN;
    } else {
        const double t_vir_min = astro_params_global->X_RAY_Tvir_MIN;
        Mmin = c21_TtoM(redshift, t_vir_min, t_vir_min < 9.99999e3 ? 1.22 : 0.6);
    }
    return Mmin / min_factor;
}
double c21_minimum_source_mass_xray(double redshift) { return minimum_source_mass_xray(redshift); }

/* interpolation.c:112-121 */
static double table_1d(double x, double x_min, double x_width, const double *y) {
    const int idx = (int)floor((x - x_min) / x_width);
    const double table_val = x_min + x_width * (double)idx;
    const double interp_point = (x - table_val) / x_width;
    return y[idx] * (1 - interp_point) + y[idx + 1] * interp_point;
}

/* initialise_Nion_Ts_spline + initialise_SFRD_spline (interp_tables.c:96-232); float limits */
static int build_z_tables(float zmin, float zmax, const c21_scaling_consts *sc) {
    const double lnMmax = log(M_MAX_INTEGRAL);
    zt.ready = 0;
    zt.x_min = zmin;
    zt.x_width = (zmax - zmin) / ((double)ZPP_INTERP_POINTS - 1.);
    const c21_scaling_consts sc_sfrd = c21_scaling_consts_sfr(sc);
    (void)c21_sigma_fast(1e10); /* build the sigma(M) spline before the threads read it */
    int bad = 0;
    const int mini = astro_options_global->USE_MINI_HALOS;
#pragma omp parallel for schedule(dynamic, 8) num_threads(host_threads()) reduction(| : bad)
    for (int i = 0; i < ZPP_INTERP_POINTS; i++) {
        const double z_val = zt.x_min + i * zt.x_width;
        const double lnMmin = log(minimum_source_mass_xray(z_val));
        /* evolve_scaling_constants_to_redshift changes t_h, which these integrals ignore, and with
         * mini-halos the atomic turnover max(M_acg(z), M_TURN) */
        const double mturn_a = mini ? c21_scaling_consts_at_z(z_val, sc).mturn_a_nofb : sc->mturn_a_nofb;
        zt.nion[i] = c21_Nion_General(z_val, lnMmin, lnMmax, mturn_a, sc);
        zt.sfrd[i] = c21_Nion_General(z_val, lnMmin, lnMmax, mturn_a, &sc_sfrd);
        if (!isfinite(zt.nion[i]) || !isfinite(zt.sfrd[i])) bad |= 1;
    }
    if (mini && !bad) { /* Nion_z_table_MINI / SFRD_z_table_MINI, interp_tables.c:105-116,174-195 */
        if (!zt.nion_mini) {
            zt.nion_mini = (double *)malloc(sizeof(double) * 2 * ZPP_INTERP_POINTS * C21_NMTURN);
            if (!zt.nion_mini) return C21CM_MEMORY_ALLOC_ERROR;
            zt.sfrd_mini = zt.nion_mini + (size_t)ZPP_INTERP_POINTS * C21_NMTURN;
        }
        zt.y_min = C21_LOG10_MTURN_MIN;
        zt.y_width = (C21_LOG10_MTURN_MAX - C21_LOG10_MTURN_MIN) / ((double)C21_NMTURN - 1.);
        int st = c21_Nion_z_tables_mini(ZPP_INTERP_POINTS, zt.x_min, zt.x_width,
                                        log(minimum_source_mass_xray(zt.x_min)), sc, C21_NMTURN,
                                        zt.y_min, zt.y_width, zt.nion_mini, zt.sfrd_mini);
        if (st) {
            c21hip_set_error("spin temperature: the mini-halo N_ion(z) / SFRD(z) tables failed (%d)", st);
            return st;
        }
    }
    if (bad) {
        c21hip_set_error("spin temperature: infinite or NaN value in the N_ion(z) / SFRD(z) tables");
        return C21CM_TABLE_GENERATION_ERROR;
    }
    zt.ready = 1;
    return 0;
}
/* init_FcollTable (interp_tables.c:252-284): CONST-ION-EFF keeps ONE table, the collapsed fraction
 * above the minimum source mass every 0.1 in redshift; N_ion and the "SFRD" both read it
 * (interp_tables.c:889-895,923-928) */
static int build_fcoll_z_table(double zmin, double zmax) {
    zt.ready = 0;
    zt.x_min = zmin;
    zt.x_width = 0.1;
    const int n_z = (int)ceil((zmax - zmin) / zt.x_width) + 1;
    if (n_z < 2 || n_z > ZT_MAX) {
        c21hip_set_error("spin temperature: %d redshifts in the collapsed-fraction table", n_z);
        return C21CM_TABLE_GENERATION_ERROR;
    }
    (void)c21_sigma_fast(1e10);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(host_threads()) reduction(| : bad)
    for (int i = 0; i < n_z; i++) {
        const double z_val = zt.x_min + i * zt.x_width;
        const double M_min = minimum_source_mass_xray(z_val);
        const double v = c21_Fcoll_General(z_val, log(M_min), log(fmax(M_MAX_INTEGRAL, M_min * 100)));
        zt.nion[i] = zt.sfrd[i] = v;
        if (!isfinite(v)) bad |= 1;
    }
    if (bad) {
        c21hip_set_error("spin temperature: infinite or NaN value in the collapsed-fraction table");
        return C21CM_TABLE_GENERATION_ERROR;
    }
    zt.n = n_z;
    zt.ready = 1;
    return 0;
}
double c21_EvaluateNionTs(double z) { return table_1d(z, zt.x_min, zt.x_width, zt.nion); }
double c21_EvaluateSFRD(double z) { return table_1d(z, zt.x_min, zt.x_width, zt.sfrd); }
/* interpolation.c:96-120 */
static double table_2d(double x, double y, const double *z_arr) {
    const int x_idx = (int)floor((x - zt.x_min) / zt.x_width);
    const int y_idx = (int)floor((y - zt.y_min) / zt.y_width);
    const double x_table = zt.x_min + zt.x_width * (double)x_idx;
    const double y_table = zt.y_min + zt.y_width * (double)y_idx;
    const double px = (x - x_table) / zt.x_width, py = (y - y_table) / zt.y_width;
    const double *r0 = z_arr + (size_t)x_idx * C21_NMTURN + y_idx, *r1 = r0 + C21_NMTURN;
    const double left_edge = r0[0] * (1 - py) + r0[1] * py;
    const double right_edge = r1[0] * (1 - py) + r1[1] * py;
    return left_edge * (1 - px) + right_edge * px;
}
double c21_EvaluateNionTs_MINI(double z, double l10mt) { return table_2d(z, l10mt, zt.nion_mini); }
double c21_EvaluateSFRD_MINI(double z, double l10mt) { return table_2d(z, l10mt, zt.sfrd_mini); }

/* ================================================================ tauX, nu_tau_one */
typedef struct {
    double nu_0, x_e, x_e_ave, ion_eff;
    int mini; /* tauX_integrand_MINI (:901-941) */
    double ion_eff_mini, log10_mturn_mini;
} taux_params;

static double tauX_integrand(double zhat, void *params) { /* :943-975 */
    const taux_params *p = (const taux_params *)params;
    const double drpropdz = PC_C_CMS * c21_dtdz(zhat);
    const double n = c21_nb0() * pow(1 + zhat, 3);
    const double nuhat = p->nu_0 * (1 + zhat);
    double fcoll, fcoll_mini = 0.;
    if (simulation_options_global->HII_DIM == 1 &&
        p->x_e_ave < simulation_options_global->MIN_XE_FOR_FCOLL_IN_TAUX) {
        fcoll = 0.;
    } else {
        fcoll = c21_EvaluateNionTs(zhat);
        if (p->mini) fcoll_mini = c21_EvaluateNionTs_MINI(zhat, p->log10_mturn_mini);
    }
    double HI_filling_factor_zhat;
    if (fcoll < 1e-20 && (!p->mini || fcoll_mini < 1e-20))
        HI_filling_factor_zhat = 1;
    else
        HI_filling_factor_zhat =
            1 - (p->ion_eff * fcoll + (p->mini ? p->ion_eff_mini * fcoll_mini : 0.)) / (1.0 - p->x_e_ave);
    if (HI_filling_factor_zhat < 1e-4) HI_filling_factor_zhat = 1e-4;
    return drpropdz * n * HI_filling_factor_zhat * c21_weighted_xray_cross_section(nuhat, p->x_e);
}

static double tauX_both(double nu, double x_e, double x_e_ave, double zp, double zpp, double ion_eff,
                        int mini, double ion_eff_mini, double log10_mturn_mini) {
    taux_params p = {nu / (1 + zp), x_e, x_e_ave, ion_eff, mini, ion_eff_mini, log10_mturn_mini};
    return c21_qag15(tauX_integrand, &p, zpp, zp, 0.005, NULL, NULL);
}
double c21_tauX(double nu, double x_e, double x_e_ave, double zp, double zpp, double ion_eff) {
    return tauX_both(nu, x_e, x_e_ave, zp, zpp, ion_eff, 0, 0., 0.);
}

typedef struct {
    double x_e, zp, zpp, ion_eff;
    int mini;
    double ion_eff_mini, log10_mturn_mini;
} tau_one_params;
static double nu_tau_one_helper(double nu, void *params) {
    const tau_one_params *p = (const tau_one_params *)params;
    return tauX_both(nu, p->x_e, p->x_e, p->zp, p->zpp, p->ion_eff, p->mini, p->ion_eff_mini,
                     p->log10_mturn_mini) - 1;
}

static double nu_tau_one_both(double zp, double zpp, double x_e, double ion_eff, int mini,
                              double ion_eff_mini, double log10_mturn_mini, int *status);
double c21_nu_tau_one(double zp, double zpp, double x_e, double ion_eff, int *status) { /* :1135-1190 */
    return nu_tau_one_both(zp, zpp, x_e, ion_eff, 0, 0., 0., status);
}
/* nu_tau_one_MINI (:1094-1160): the same root with both populations in the filling factor */
double c21_nu_tau_one_MINI(double zp, double zpp, double x_e, double ion_eff, double ion_eff_mini,
                           double log10_mturn_mini, int *status) {
    return nu_tau_one_both(zp, zpp, x_e, ion_eff, 1, ion_eff_mini, log10_mturn_mini, status);
}
static double nu_tau_one_both(double zp, double zpp, double x_e, double ion_eff, int mini,
                              double ion_eff_mini, double log10_mturn_mini, int *status) {
    if (status) *status = 0;
    if (x_e > 0.9999) return astro_params_global->NU_X_THRESH; /* sic: eV, not Hz (:1146-1149) */
    if (tauX_both(PC_NU_ION_HEI, x_e, x_e, zp, zpp, ion_eff, mini, ion_eff_mini, log10_mturn_mini) < 1)
        return PC_NU_ION_HEI;
    tau_one_params p = {x_e, zp, zpp, ion_eff, mini, ion_eff_mini, log10_mturn_mini};
    int st = 0;
    const double r = c21_brent_root(nu_tau_one_helper, &p, PC_NU_ION_HEI, 1e6 * PC_EV_TO_HZ, 0.02, 100, &st);
    if (!isfinite(r) || st == 1) {
        c21hip_set_error("nu_tau_one: no root of tau_X = 1 between the HeI edge and 1 MeV");
        if (status) *status = C21CM_INFINITY_OR_NAN_ERROR;
    }
    return r;
}

/* ================================================================ the per-snapshot tables */
void c21_ts_tables_free(c21_ts_tables *t) {
    if (!t) return;
    free(t->freq);
    free(t->sfrd_tables);
    free(t->fcoll_tables);
    free(t->dfcoll_tables);
    free(t->sfrd_tables_mini);
    memset(t, 0, sizeof(*t));
}

/* setup_z_edges (:312-361) */
static void setup_z_edges(double zp, c21_ts_tables *t) {
    const int n = t->n_step;
    double R;
    if (simulation_options_global->HII_DIM == 1)
        R = PC_L_FACTOR * 1.5;
    else
        R = PC_L_FACTOR * simulation_options_global->BOX_LEN / (float)simulation_options_global->HII_DIM;
    const double R_factor = pow(astro_params_global->R_MAX_TS / R, 1 / ((float)n));
    double prev_zpp, prev_R;
    for (int R_ct = 0; R_ct < n; R_ct++) {
        t->R_values[R_ct] = R;
        if (R_ct == 0) {
            prev_zpp = zp;
            prev_R = 0;
        } else {
            prev_zpp = t->zpp_edge[R_ct - 1];
            prev_R = t->R_values[R_ct - 1];
        }
        /* drdz(float z) = (1 + z) c dtdz(z), cosmology.c:778-779 */
        const float pz = (float)prev_zpp;
        const double drdz = (1.0 + pz) * PC_C_CMS * c21_dtdz(pz);
        t->zpp_edge[R_ct] = prev_zpp - (t->R_values[R_ct] - prev_R) * PC_CM_PER_MPC / drdz;
        const double zpp = (t->zpp_edge[R_ct] + prev_zpp) * 0.5;
        t->zpp[R_ct] = zpp;
        t->dzpp[R_ct] = R_ct == 0 ? zp - t->zpp_edge[0] : t->zpp_edge[R_ct - 1] - t->zpp_edge[R_ct];
        t->zpp_growth[R_ct] = dicke(zpp);
        t->dtdz[R_ct] = c21_dtdz(zpp);
        t->M_min_R[R_ct] = minimum_source_mass_xray(zpp);
        t->M_max_R[R_ct] = c21_RtoM(t->R_values[R_ct]);
        R *= R_factor;
    }
}

/* calculate_spectral_factors (:364-499), Pop II only (no mini-halos) */
static void spectral_factors(double zp, c21_ts_tables *t) {
    int first_radii = 1, first_zero = 1;
    const int n_pts_radii = 1000;
    double weight = 0., sum_lyn_prev = 0., sum_ly2_prev = 0., sum_lynto2_prev = 0., prev_zpp = 0;
    double sum_lyn_prev_MINI = 0., sum_ly2_prev_MINI = 0., sum_lynto2_prev_MINI = 0.;
    const int mini = astro_options_global->USE_MINI_HALOS;
    const double lw_edge = PC_NU_LW_THRESH / PC_NU_ION_HI;
    const double shield = 1. - astro_params_global->F_H2_SHIELD;
    for (int R_ct = 0; R_ct < t->n_step; R_ct++) {
        const double zpp = t->zpp[R_ct];
        double sum_lynto2_val = 0., sum_ly2_val = 0.;
        double sum_lynto2_val_MINI = 0., sum_ly2_val_MINI = 0., sum_lyLW_val = 0., sum_lyLW_val_MINI = 0.;
        double nuprime = c21_nu_n(2) * (1. + zpp) / (1. + zp);
        if (zpp < c21_zmax(zp, 2)) {
            sum_ly2_val = c21_frecycle(2) * c21_spectral_emissivity(nuprime, 2);
            if (mini) { /* :397-410 (nuprime < nu_n(3) holds by the definition of zmax) */
                sum_ly2_val_MINI = c21_frecycle(2) * c21_spectral_emissivity(nuprime, 3);
                if (nuprime < lw_edge) nuprime = lw_edge;
                if (nuprime < c21_nu_n(3)) {
                    sum_lyLW_val += shield * c21_spectral_emissivity_lw(nuprime, 2);
                    sum_lyLW_val_MINI += shield * c21_spectral_emissivity_lw(nuprime, 3);
                }
            }
        }
        for (int n_ct = NSPEC_MAX; n_ct >= 3; n_ct--) {
            if (zpp > c21_zmax(zp, n_ct)) continue;
            nuprime = c21_nu_n(n_ct) * (1 + zpp) / (1.0 + zp);
            sum_lynto2_val += c21_frecycle(n_ct) * c21_spectral_emissivity(nuprime, 2);
            if (mini) { /* :418-429 */
                sum_lynto2_val_MINI += c21_frecycle(n_ct) * c21_spectral_emissivity(nuprime, 3);
                if (nuprime < lw_edge) nuprime = lw_edge;
                if (nuprime >= c21_nu_n(n_ct + 1)) continue;
                sum_lyLW_val += shield * c21_spectral_emissivity_lw(nuprime, 2);
                sum_lyLW_val_MINI += shield * c21_spectral_emissivity_lw(nuprime, 3);
            }
        }
        double sum_lyn_val = sum_ly2_val + sum_lynto2_val;
        double sum_lyn_val_MINI = sum_ly2_val_MINI + sum_lynto2_val_MINI;
        if (R_ct > 1 && sum_lyn_val == 0.0 && sum_lyn_prev > 0. && first_radii) {
            for (int ii = 0; ii < n_pts_radii; ii++) {
                const double trial_zpp = prev_zpp + (zpp - prev_zpp) * (float)ii / ((float)n_pts_radii - 1.);
                int counter = 0;
                for (int n_ct = NSPEC_MAX; n_ct >= 2; n_ct--) {
                    if (trial_zpp > c21_zmax(zp, n_ct)) continue;
                    counter += 1;
                }
                if (counter == 0 && first_zero) {
                    first_zero = 0;
                    weight = (float)ii / (float)n_pts_radii;
                }
            }
            sum_lyn_val = weight * sum_lyn_prev;
            sum_ly2_val = weight * sum_ly2_prev;
            sum_lynto2_val = weight * sum_lynto2_prev;
            if (mini) {
                sum_lyn_val_MINI = weight * sum_lyn_prev_MINI;
                sum_ly2_val_MINI = weight * sum_ly2_prev_MINI;
                sum_lynto2_val_MINI = weight * sum_lynto2_prev_MINI;
            }
            first_radii = 0;
        }
        const double zpp_integrand = (pow(1 + zp, 2) * (1 + zpp));
        t->starlya_prefactor[R_ct] = zpp_integrand * sum_lyn_val;
        t->lya_cont_prefactor[R_ct] = zpp_integrand * sum_ly2_val;
        t->lya_inj_prefactor[R_ct] = zpp_integrand * sum_lynto2_val;
        if (mini) { /* :475-482 */
            t->starlya_prefactor_mini[R_ct] = zpp_integrand * sum_lyn_val_MINI;
            t->lw_prefactor[R_ct] = zpp_integrand * sum_lyLW_val;
            t->lw_prefactor_mini[R_ct] = zpp_integrand * sum_lyLW_val_MINI;
            t->lya_cont_prefactor_mini[R_ct] = zpp_integrand * sum_ly2_val_MINI;
            t->lya_inj_prefactor_mini[R_ct] = zpp_integrand * sum_lynto2_val_MINI;
        }
        sum_lyn_prev = sum_lyn_val;
        sum_ly2_prev = sum_ly2_val;
        sum_lynto2_prev = sum_lynto2_val;
        sum_lyn_prev_MINI = sum_lyn_val_MINI;
        sum_ly2_prev_MINI = sum_ly2_val_MINI;
        sum_lynto2_prev_MINI = sum_lynto2_val_MINI;
        prev_zpp = zpp;
    }
}

/* set_zp_consts (:1098-1184) into the spec */
static void set_zp_consts(double zp, int lagrangian, c21cm_ts_spec *s) {
    const AstroParams *ap = astro_params_global;
    const double Ho = c21_hubble0(), No = number_density_H(), N_b0 = c21_nb0();
    double lum;
    if (fabs(ap->X_RAY_SPEC_INDEX - 1.0) < 1e-6) {
        lum = (ap->NU_X_THRESH) * PC_EV_TO_HZ * log(ap->NU_X_BAND_MAX / (ap->NU_X_THRESH));
        lum = 1. / lum;
    } else {
        lum = pow((ap->NU_X_BAND_MAX) * PC_EV_TO_HZ, 1. - (ap->X_RAY_SPEC_INDEX)) -
              pow((ap->NU_X_THRESH) * PC_EV_TO_HZ, 1. - (ap->X_RAY_SPEC_INDEX));
        lum = 1. / lum;
        lum *= pow((ap->NU_X_THRESH) * PC_EV_TO_HZ, -(ap->X_RAY_SPEC_INDEX)) * (1 - (ap->X_RAY_SPEC_INDEX));
    }
    lum /= PC_H_P;
    s->growth_zp = dicke(zp);
    s->hubble_zp = c21_hubble(zp);
    {
        const float dz = 1e-10; /* ddicke_dz, cosmology.c:586-590 */
        s->dgrowth_dzp = (dicke(zp + dz) - dicke(zp)) / dz;
    }
    s->dt_dzp = c21_dtdz(zp);
    s->xray_prefactor = lum / ((ap->NU_X_THRESH) * PC_EV_TO_HZ) * PC_C_CMS * pow(1 + zp, ap->X_RAY_SPEC_INDEX + 3);
    s->Trad = PC_T_CMB * (1.0 + zp);
    s->Ts_prefactor = pow(1.0e-7 * (1.342881e-7 / s->hubble_zp) * No * pow(1 + zp, 3), 1. / 3.);
    double gamma_alpha = PC_F_ALPHA * pow(PC_NU_LY_ALPHA * PC_E_CHARGE / (PC_C_CMS / 10.), 2.);
    gamma_alpha /= 6. * (PC_M_E / 1000.) * pow(PC_C_CMS / 100., 3.) * PC_VAC_PERM;
    s->xa_tilde_prefactor = 8. * M_PI * pow(PC_LAMBDA_LY_ALPHA * 1.e-8, 2.) * gamma_alpha * PC_T_21;
    s->xa_tilde_prefactor /= 9. * PC_A10 * s->Trad;
    s->xc_inverse = pow(1.0 + zp, 3.0) * PC_T_21 / (s->Trad * PC_A10);
    s->dcomp_dzp_prefactor = (-1.51e-4) / (s->hubble_zp / Ho) / (cosmo_params_global->hlittle) *
                             pow(s->Trad, 4.0) / (1.0 + zp);
    s->Nb_zp = N_b0 * (1 + zp) * (1 + zp) * (1 + zp);
    s->N_zp = No * (1 + zp) * (1 + zp) * (1 + zp);
    s->lya_star_prefactor = PC_C_CMS / (4.0 * M_PI) * PC_MSUN / PC_M_P * (1 - 0.75 * cosmo_params_global->Y_He);
    if (lagrangian)
        s->volunit_inv = pow(PC_CM_PER_MPC, -3);
    else
        s->volunit_inv = cosmo_params_global->OMb * c21_rhocrit() * pow(PC_CM_PER_MPC, -3);
    s->No = No;
    s->N_b0 = N_b0;
    s->h_frac = h_frac();
    s->he_frac = he_frac();
    s->k_B = PC_K_B, s->h_p = PC_H_P, s->m_p = PC_M_P, s->c_cms = PC_C_CMS, s->A10 = PC_A10;
    s->T_21 = PC_T_21, s->lambda_21 = PC_LAMBDA_21, s->nu_Ly_alpha = PC_NU_LY_ALPHA;
    s->clumping_factor = ap->CLUMPING_FACTOR;
}

/* Everything ts_main computes on the host up to (not including) the conditional SFRD tables:
 * shells, spectral factors, z' constants, the global N_ion / SFRD tables, Q_HI, NO_LIGHT, the
 * mean SFRD per shell and the frequency-integral tables.  x_e_ave: box mean of the previous
 * x_e box. */
int c21_ts_prepare(float redshift, float prev_redshift, float perturbed_field_redshift,
                   double x_e_ave, c21cm_ts_spec *s, c21_ts_tables *t) {
    const int status = c21_ts_prepare_shells(redshift, prev_redshift, perturbed_field_redshift, s, t);
    return status ? status : c21_ts_prepare_tables(x_e_ave, s, t);
}

/* Part 1, cheap: options, data tables, shells, spectral factors, z' constants.  The density filter
 * loop only needs the shells' radii, so ComputeTsBox runs part 2 beside it. */
int c21_ts_prepare_shells(float redshift, float prev_redshift, float perturbed_field_redshift,
                          c21cm_ts_spec *s, c21_ts_tables *t) {
    int status;
    const AstroParams *ap = astro_params_global;
    const AstroOptions *ao = astro_options_global;
    const int model = matter_options_global->SOURCE_MODEL;
    if (ao->USE_MINI_HALOS && (model == C21CM_SOURCE_CONST_ION_EFF || ao->INTEGRATION_METHOD_MINI > 1)) {
        c21hip_set_error("ComputeTsBox: USE_MINI_HALOS needs a mass-dependent SOURCE_MODEL and the "
                         "Gauss-Legendre or adaptive integrals");
        return C21CM_VALUE_ERROR;
    }
    if (matter_options_global->USE_INTERPOLATION_TABLES == 0) {
        c21hip_set_error("ComputeTsBox: USE_INTERPOLATION_TABLES = no-interpolation is not supported");
        return C21CM_VALUE_ERROR;
    }
    const int n = ap->N_STEP_TS;
    if (n < 1 || n > C21CM_MAX_TS_RADII) {
        c21hip_set_error("ComputeTsBox: N_STEP_TS = %d (supported: 1..%d)", n, C21CM_MAX_TS_RADII);
        return C21CM_VALUE_ERROR;
    }
    if ((status = c21_heat_load())) return status;
    const int const_zeta = model == C21CM_SOURCE_CONST_ION_EFF;
    const int lagrangian = model != C21CM_SOURCE_E_INTEGRAL && !const_zeta;
    memset(s, 0, sizeof(*s));
    c21_ts_tables_free(t);
    t->n_step = n;
    s->hii_dim = simulation_options_global->HII_DIM;
    s->hii_dim_z = (int)(simulation_options_global->NON_CUBIC_FACTOR * simulation_options_global->HII_DIM);
    s->n_step = n;
    s->source_mode = lagrangian ? C21CM_TS_SRC_GRIDS
                                : (const_zeta ? C21CM_TS_SRC_FCOLL_TABLES : C21CM_TS_SRC_SFRD_TABLE);
    s->use_xray_heating = ao->USE_X_RAY_HEATING;
    s->use_cmb_heating = ao->USE_CMB_HEATING;
    s->use_lya_heating = ao->USE_LYA_HEATING;
    s->redshift = redshift;
    s->dzp = (float)((double)(redshift - prev_redshift)); /* get_Ts_fast takes it as a float */
    s->growth_ratio = dicke(redshift) * (1. / dicke(perturbed_field_redshift));
    const double zp = redshift;

    setup_z_edges(zp, t);
    spectral_factors(zp, t);
    set_zp_consts(zp, lagrangian, s);
    for (int i = 0; i < n; i++) {
        s->starlya_prefactor[i] = t->starlya_prefactor[i];
        s->lya_cont_prefactor[i] = t->lya_cont_prefactor[i];
        s->lya_inj_prefactor[i] = t->lya_inj_prefactor[i];
        s->zpp_growth[i] = t->zpp_growth[i];
        const double zpp = t->zpp[i];
        if (const_zeta) /* :1546-1553: the source is dfcoll/dz */
            s->z_edge_factor[i] = t->dzpp[i];
        else if (lagrangian)
            s->z_edge_factor[i] = fabs(t->dzpp[i] * t->dtdz[i]);
        else
            s->z_edge_factor[i] = fabs(t->dzpp[i] * t->dtdz[i]) * c21_hubble(zpp) / ap->t_STAR;
        t->sigma_min[i] = c21_sigma_fast(t->M_min_R[i]); /* :1418-1421 */
        t->sigma_max[i] = c21_sigma_fast(t->M_max_R[i]);
        s->xray_R_factor[i] = pow(1 + zpp, -(ap->X_RAY_SPEC_INDEX));
    }
    s->sfr_scale = ap->F_STAR10;
    s->xray_scale = ap->L_X * PC_S_PER_YR;
    if (ao->USE_MINI_HALOS) {
        s->use_mini_halos = 1;
        s->sfr_scale_mini = ap->F_STAR7_MINI;
        s->xray_scale_mini = ap->L_X_MINI * PC_S_PER_YR;
        s->mturn_tab_min = C21_LOG10_MTURN_MIN;
        s->mturn_tab_width = (C21_LOG10_MTURN_MAX - C21_LOG10_MTURN_MIN) / (C21_NMTURN - 1.);
        for (int i = 0; i < n; i++) {
            s->starlya_prefactor_mini[i] = t->starlya_prefactor_mini[i];
            s->lya_cont_prefactor_mini[i] = t->lya_cont_prefactor_mini[i];
            s->lya_inj_prefactor_mini[i] = t->lya_inj_prefactor_mini[i];
            s->lw_prefactor[i] = t->lw_prefactor[i];
            s->lw_prefactor_mini[i] = t->lw_prefactor_mini[i];
        }
    }
    if (s->use_lya_heating) {
        s->lya_dEC = H.dEC;
        s->lya_dEI = H.dEI;
    }
    return 0;
}

/* Part 2: the global N_ion / SFRD tables, Q_HI, NO_LIGHT, the mean SFRD per shell, the tau_X = 1
 * frequencies and the frequency-integral tables (global_reion_properties :930-1008 with
 * fill_freqint_tables :810-889).  x_e_ave: box mean of the previous x_e box. */
int c21_ts_prepare_tables(double x_e_ave, c21cm_ts_spec *s, c21_ts_tables *t) {
    int status;
    const AstroParams *ap = astro_params_global;
    const int const_zeta = matter_options_global->SOURCE_MODEL == C21CM_SOURCE_CONST_ION_EFF;
    const int n = t->n_step;
    const double zp = s->redshift;

    c21_scaling_consts sc;
    if ((status = c21_set_scaling_constants(zp, &sc))) return status;
    {
        const double determine_zpp_min = zp * 0.999;
        const double determine_zpp_max = t->zpp[n - 1] * 1.001;
        if (const_zeta)
            status = build_fcoll_z_table(determine_zpp_min, determine_zpp_max);
        else
            status = build_z_tables((float)determine_zpp_min, (float)determine_zpp_max, &sc);
        if (status) return status;
    }
    const int mini = astro_options_global->USE_MINI_HALOS;
    const double sum_nion = c21_EvaluateNionTs(zp);
    /* global_reion_properties (:973-1007): both populations in Q_HI and in NO_LIGHT */
    const double sum_nion_mini = mini ? c21_EvaluateNionTs_MINI(zp, t->ave_log10_mturn[0]) : 0.;
    const double ion_eff = const_zeta ? (double)ap->HII_EFF_FACTOR : ap->F_STAR10 * ap->F_ESC10 * ap->POP2_ION;
    const double ion_eff_mini = mini ? ap->F_STAR7_MINI * ap->F_ESC7_MINI * ap->POP3_ION : 0.;
    t->Q_HI = 1 - (ion_eff * sum_nion + ion_eff_mini * sum_nion_mini) / (1.0 - x_e_ave);
    t->no_light = sum_nion + sum_nion_mini > 1e-15 ? 0 : 1;
    s->no_light = t->no_light;
    for (int i = 0; i < n; i++) {
        t->mean_sfr_zpp[i] = c21_EvaluateSFRD(t->zpp[i]);
        s->mean_sfr_zpp[i] = t->mean_sfr_zpp[i];
        if (mini) {
            t->mean_sfr_zpp_mini[i] = c21_EvaluateSFRD_MINI(t->zpp[i], t->ave_log10_mturn[i]);
            s->mean_sfr_zpp_mini[i] = t->mean_sfr_zpp_mini[i];
        }
    }

    /* ---- fill_freqint_tables (:810-889); tauX's efficiency: pop2_ion fstar_10 fesc_10 (:1027-1029) */
    const size_t fn = (size_t)C21CM_X_INT_NXHII * n;
    t->freq = (double *)calloc(3 * fn, sizeof(double));
    if (!t->freq) return C21CM_MEMORY_ALLOC_ERROR;
    double tau_ion_eff = sc.pop2_ion * sc.fstar_10 * sc.fesc_10;
    if (const_zeta) { /* tauX :1030-1040: the efficiency implied by the filling factor at z' */
        static double PS_ION_EFF; /* kept across calls for the post-reionisation regime, as upstream */
        if (t->Q_HI > FRACT_FLOAT_ERR) {
            const double fcoll = c21_EvaluateNionTs(zp);
            PS_ION_EFF = (1.0 - t->Q_HI) / fcoll * (1.0 - x_e_ave);
        }
        tau_ion_eff = PS_ION_EFF;
    }
    int root_failed = 0, table_bad = 0;
    /* the shells are independent (:822-863); a rank of a sharded ComputeTsBox only does its own (shell_mask).
     * Two loops -- the tau_X = 1 roots per shell, then the (shell, x_e) pairs of the integrals -- so that five
     * shells still fill the host threads (the integrand sequence of an integral is unchanged) */
    int todo[C21CM_MAX_TS_RADII], n_todo = 0;
    double lower_lim[C21CM_MAX_TS_RADII];
    for (int R_ct = 0; R_ct < n; R_ct++)
        if (!t->shell_mask || t->shell_mask[R_ct]) todo[n_todo++] = R_ct;
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads()) reduction(| : root_failed)
    for (int k = 0; k < n_todo; k++) {
        const int R_ct = todo[k];
        int st = 0;
        /* :833-842: with mini-halos the root uses each shell's mean turnover mass */
        const double nu1 =
            mini ? c21_nu_tau_one_MINI(zp, t->zpp[R_ct], x_e_ave, tau_ion_eff,
                                       sc.pop3_ion * sc.fstar_7 * sc.fesc_7, t->ave_log10_mturn[R_ct], &st)
                 : c21_nu_tau_one(zp, t->zpp[R_ct], x_e_ave, tau_ion_eff, &st);
        if (st) root_failed |= 1;
        lower_lim[R_ct] = fmax(nu1, (ap->NU_X_THRESH) * PC_EV_TO_HZ);
        t->nu_tau_one[R_ct] = nu1;
    }
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(host_threads()) reduction(| : table_bad)
    for (int k = 0; k < n_todo; k++) {
        for (int x_e_ct = 0; x_e_ct < C21CM_X_INT_NXHII; x_e_ct++) {
            const int R_ct = todo[k];
            for (int flag = 0; flag < 3; flag++) {
                const double v = c21_integrate_over_nu(zp, H.x_int_XHII[x_e_ct], lower_lim[R_ct], flag);
                if (!isfinite(v)) table_bad |= 1;
                t->freq[flag * fn + (size_t)x_e_ct * n + R_ct] = v;
            }
        }
    }
    if (root_failed) return C21CM_INFINITY_OR_NAN_ERROR;
    if (table_bad) {
        c21hip_set_error("One of the frequency interpolation tables has an infinity or a NaN");
        return C21CM_TABLE_GENERATION_ERROR;
    }
    s->freq_int_heat = t->freq;
    s->freq_int_ion = t->freq + fn;
    s->freq_int_lya = t->freq + 2 * fn;
    return 0;
}

/* calculate_sfrd_from_grid's tables (:1016-1036 with interp_tables.c:415-494): one 400-point
 * ln SFRD(delta) table per shell over [min, 1.001 max] of the shell's filtered density */
int c21_ts_sfrd_tables(const double *min_densities, const double *max_densities, c21cm_ts_spec *s,
                       c21_ts_tables *t) {
    const int n = t->n_step;
    free(t->sfrd_tables);
    t->sfrd_tables = (float *)malloc((size_t)n * C21CM_NDELTA_TABLE * sizeof(float));
    if (!t->sfrd_tables) return C21CM_MEMORY_ALLOC_ERROR;
    const int mini = astro_options_global->USE_MINI_HALOS;
    if (mini) {
        free(t->sfrd_tables_mini);
        t->sfrd_tables_mini =
            (float *)malloc((size_t)n * C21CM_NDELTA_TABLE * C21_NMTURN * sizeof(float));
        if (!t->sfrd_tables_mini) return C21CM_MEMORY_ALLOC_ERROR;
    }
    for (int R_ct = 0; R_ct < n; R_ct++) {
        c21_scaling_consts sc; /* set_scaling_constants(zpp) (:1558), then the SFRD variant */
        int status = c21_set_scaling_constants(t->zpp[R_ct], &sc);
        if (status) return status;
        sc = c21_scaling_consts_sfr(&sc);
        const double g = t->zpp_growth[R_ct];
        const double dmin = min_densities[R_ct] * g, dmax = max_densities[R_ct] * g * 1.001;
        const double lnMcond = log(t->M_max_R[R_ct]);
        status = c21_Nion_Conditional_table(dicke(t->zpp[R_ct]), log(t->M_min_R[R_ct]), lnMcond, lnMcond,
                                            c21_sigma_fast(t->M_max_R[R_ct]), dmin, dmax, sc.mturn_a_nofb,
                                            &sc, astro_options_global->INTEGRATION_METHOD_ATOMIC, -50.,
                                            t->sfrd_tables + (size_t)R_ct * C21CM_NDELTA_TABLE,
                                            C21CM_NDELTA_TABLE);
        if (status) return status;
        if (mini) { /* SFRD_conditional_table_MINI on the fixed turnover grid (:440-475) */
            status = c21_Nion_Conditional_table2d(
                dicke(t->zpp[R_ct]), log(t->M_min_R[R_ct]), lnMcond, lnMcond,
                (float)c21_sigma_fast(t->M_max_R[R_ct]), dmin, dmax, C21_LOG10_MTURN_MIN,
                C21_LOG10_MTURN_MAX, &sc, 1, astro_options_global->INTEGRATION_METHOD_MINI, -50., 1,
                t->sfrd_tables_mini + (size_t)R_ct * C21CM_NDELTA_TABLE * C21_NMTURN,
                C21CM_NDELTA_TABLE, C21_NMTURN);
            if (status) return status;
        }
        s->tab_min[R_ct] = dmin;
        s->tab_width[R_ct] = (dmax - dmin) / (C21CM_NDELTA_TABLE - 1.);
    }
    s->ln_sfrd_tables = t->sfrd_tables;
    if (mini) s->ln_sfrd_tables_mini = t->sfrd_tables_mini;
    return 0;
}

/* initialise_FgtrM_delta_table (interp_tables.c:226-250) for every shell: the conditional collapsed
 * fraction and its redshift derivative over [min, max] of the shell's filtered density (:1029-1032:
 * no margin at the upper end) */
int c21_ts_fcoll_tables(const double *min_densities, const double *max_densities, c21cm_ts_spec *s,
                        c21_ts_tables *t) {
    const int n = t->n_step;
    free(t->fcoll_tables);
    free(t->dfcoll_tables);
    t->fcoll_tables = (float *)malloc((size_t)n * C21CM_NDELTA_TABLE * sizeof(float));
    t->dfcoll_tables = (float *)malloc((size_t)n * C21CM_NDELTA_TABLE * sizeof(float));
    if (!t->fcoll_tables || !t->dfcoll_tables) return C21CM_MEMORY_ALLOC_ERROR;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads()) reduction(| : bad)
    for (int R_ct = 0; R_ct < n; R_ct++) {
        const double g = t->zpp_growth[R_ct];
        const double lo = min_densities[R_ct] * g, hi = max_densities[R_ct] * g;
        const double width = (hi - lo) / (C21CM_NDELTA_TABLE - 1.);
        for (int i = 0; i < C21CM_NDELTA_TABLE; i++) {
            const double dens = lo + i * width;
            const double f = c21_FgtrM_bias_fast(g, dens, t->sigma_min[R_ct], t->sigma_max[R_ct]);
            if (isnan(f)) bad |= 1;
            t->fcoll_tables[(size_t)R_ct * C21CM_NDELTA_TABLE + i] = f;
            t->dfcoll_tables[(size_t)R_ct * C21CM_NDELTA_TABLE + i] =
                c21_dfcoll_dz(t->zpp[R_ct], t->sigma_min[R_ct], dens, t->sigma_max[R_ct]);
        }
        s->tab_min[R_ct] = lo;
        s->tab_width[R_ct] = width;
    }
    if (bad) {
        c21hip_set_error("Trying to compute FgtrM in region where M_min > M_max");
        return C21CM_VALUE_ERROR;
    }
    s->fcoll_tables = t->fcoll_tables;
    s->dfcoll_tables = t->dfcoll_tables;
    return 0;
}
