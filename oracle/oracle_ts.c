/*
 * oracle_ts.c -- CPU restatement of the per-cell part of ComputeTsBox.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain loops in the reference's own order and
 * precision; all cosmology / spectra / frequency integrals arrive as the scalars and tables of
 * c21cm_ts_spec, the same ones the library kernel gets.
 *
 * reference: src/py21cmfast/src/SpinTemperatureBox.c
 *   :892-927    init_first_Ts
 *   :1010-1086  calculate_sfrd_from_grid  (E-INTEGRAL: table of ln SFRD(delta), float grid)
 *   :1210-1383  get_Ts_fast
 *   :1499-1522  x_e index and interpolation weight of every cell (float arithmetic)
 *   :1541-1784  the R loop, largest shell first
 *   :1794-1848  prefactors, get_Ts_fast, outputs; :1884-1904 the finiteness check
 * src/py21cmfast/src/heating_helper_progs.c
 *   :366-643    kappa_10, kappa_10_pH, kappa_10_elec (linear interpolation in ln T)
 *   :650-653    taugp
 *   :695-728    xcoll_HI / xcoll_elec / xcoll_prot
 *   :736-760    get_Ts (collisions-only branch, used by init_first_Ts)
 *   :1227-1313  interpolate_heating_efficiencies (tri-linear, clamped)
 * src/py21cmfast/src/thermochem.c:66-75 alpha_A; src/py21cmfast/src/interpolation.c:123-131.
 *   :535-565,1011-1075,1642-1733,1843-1845  USE_MINI_HALOS (E-INTEGRAL): turnover grid, 2-D SFRD
 *               tables, both populations in the shell loop, J_21_LW.  PARITY UNPINNED: no
 *               reproducible reference vector (the `mini` fixtures need CLASS transfer tables);
 *               tied to the pinned one-population path by tests/test_oracle_ts_minihalos.py.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "c21cm_kappa_tables.h"
#include "oracle.h"

#define FRACT_FLOAT_ERR 1e-7

static const double KAPPA_HH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_HH_VALUES;
static const double KAPPA_PH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_PH_VALUES;
static const double KAPPA_EH[C21CM_KAPPA_NPTS] = C21CM_KAPPA_EH_VALUES;

/* the interior branch the three functions share (the `(float)idx` of the source is exact) */
static double kappa_interior(const double *y, double width, double lnT) {
    int idx = (int)floor(lnT * (1. / width));
    if (idx > C21CM_KAPPA_NPTS - 2) idx = C21CM_KAPPA_NPTS - 2; /* lnT == last knot */
    return y[idx] + (lnT - width * (double)idx) * (y[idx + 1] - y[idx]) * (1. / width);
}

double oracle_kappa_10(double TK) { /* :366-455 */
    const double lnT = log(TK);
    double ans;
    if (lnT < 0.)
        ans = KAPPA_HH[0];
    else if (lnT > C21CM_KAPPA_HH_LNT_MAX)
        ans = log(exp(KAPPA_HH[C21CM_KAPPA_NPTS - 1]) *
                  pow(exp(lnT) / exp(C21CM_KAPPA_HH_LNT_MAX), 0.381));
    else
        ans = kappa_interior(KAPPA_HH, C21CM_KAPPA_HH_BINWIDTH, lnT);
    return exp(ans);
}

static double kappa_linear_tail(const double *y, double width, double lnT_max, double lnT) {
    double ans;
    if (lnT < 0.)
        ans = y[0];
    else if (lnT > lnT_max) /* the last segment's slope, with the printed knot positions */
        ans = y[C21CM_KAPPA_NPTS - 1] + (y[C21CM_KAPPA_NPTS - 1] - y[C21CM_KAPPA_NPTS - 2]) /
                                            (lnT_max - width * (C21CM_KAPPA_NPTS - 2)) *
                                            (lnT - lnT_max);
    else
        ans = kappa_interior(y, width, lnT);
    return exp(ans);
}

double oracle_kappa_10_pH(double T) { /* :457-549 */
    return kappa_linear_tail(KAPPA_PH, C21CM_KAPPA_PH_BINWIDTH, C21CM_KAPPA_PH_LNT_MAX, log(T));
}

double oracle_kappa_10_elec(double T) { /* :551-643 */
    return kappa_linear_tail(KAPPA_EH, C21CM_KAPPA_EH_BINWIDTH, C21CM_KAPPA_EH_LNT_MAX, log(T));
}

double oracle_alpha_A(double T) { /* thermochem.c:66-75 (Abel et al. 1997) */
    const double logT = log(T / 1.1604505e4);
    return exp(-28.6130338 - 0.72411256 * logT - 2.02604473e-2 * pow(logT, 2) -
               2.38086188e-3 * pow(logT, 3) - 3.21260521e-4 * pow(logT, 4) -
               1.42150291e-5 * pow(logT, 5) + 4.98910892e-6 * pow(logT, 6) +
               5.75561414e-7 * pow(logT, 7) - 1.85676704e-8 * pow(logT, 8) -
               3.07113524e-9 * pow(logT, 9));
}

/* heating_helper_progs.c:1210-1222 */
static int nearest_point(double min, double max, int n, double value) {
    const double dn = (max - min) / (n - 1);
    if (value <= (min + dn)) return 0;
    if (value >= max) return n - 2;
    return (int)floor((value - min) / dn);
}

/* :1234-1313: tri-linear in (log10 Tk, log10 Ts, log10 tau_GP), arguments clamped to the table */
double oracle_lya_heating_efficiency(double tk, double ts, double taugp, const double *arrE) {
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
    const double xd = (tk - x0) / (x1 - x0), yd = (ts - y0) / (y1 - y0), zd = (taugp - z0) / (z1 - z0);
#define AT(a, b, c) arrE[((size_t)(a) * nT + (b)) * ngp + (c)]
    const double c00 = AT(itk, its, igp) * (1. - xd) + AT(itk + 1, its, igp) * xd;
    const double c01 = AT(itk, its, igp + 1) * (1. - xd) + AT(itk + 1, its, igp + 1) * xd;
    const double c10 = AT(itk, its + 1, igp) * (1. - xd) + AT(itk + 1, its + 1, igp) * xd;
    const double c11 = AT(itk, its + 1, igp + 1) * (1. - xd) + AT(itk + 1, its + 1, igp + 1) * xd;
#undef AT
    const double c0 = c00 * (1. - yd) + c10 * yd, c1 = c01 * (1. - yd) + c11 * yd;
    return c0 * (1. - zd) + c1 * zd;
}

/* interpolation.c:123-131 */
static double table_1d(double x, double x_min, double x_width, const float *y) {
    const int idx = (int)floor((x - x_min) / x_width);
    const double table_val = x_min + x_width * (float)idx;
    const double interp_point = (x - table_val) / x_width;
    return y[idx] * (1 - interp_point) + y[idx + 1] * interp_point;
}

typedef struct {
    double dxion_dt, dxheat_dt, dxlya_dt, dstarlya_dt, dstarlya_cont_dt, dstarlya_inj_dt, delta;
    double prev_Ts, prev_Tk, prev_xe;
} rad_terms;

/* get_Ts_fast (:1210-1383); zp and dzp are floats upstream, the spec holds them rounded */
static void ts_cell(const c21cm_ts_spec *s, const rad_terms *rad, double *Ts_out, double *Tk_out,
                    double *xe_out) {
    const double zp = s->redshift, dzp = s->dzp;
    const double tau21 = (3 * s->h_p * s->A10 * s->c_cms * s->lambda_21 * s->lambda_21 / 32. / M_PI /
                          s->k_B) *
                         ((1 - rad->prev_xe) * s->N_zp) / rad->prev_Ts / s->hubble_zp;
    double xCMB;
    if (tau21 > 1e-8)
        xCMB = (1. - exp(-tau21)) / tau21;
    else
        xCMB = 1. - tau21 / 2 * (1 - tau21 / 3 * (1 - tau21 / 4));

    const double dxion_sink_dt = oracle_alpha_A(rad->prev_Tk) * s->clumping_factor * rad->prev_xe *
                                 rad->prev_xe * s->h_frac * s->Nb_zp * (1. + rad->delta);
    const double dxe_dzp = s->dt_dzp * (rad->dxion_dt - dxion_sink_dt);

    double dadia_dzp = 3 / (1.0 + zp);
    if (fabs(rad->delta) > FRACT_FLOAT_ERR)
        dadia_dzp += s->dgrowth_dzp / (s->growth_zp * (1.0 / rad->delta + 1.0));
    dadia_dzp *= (2.0 / 3.0) * rad->prev_Tk;

    const double dspec_dzp = -dxe_dzp * rad->prev_Tk / (1 + rad->prev_xe);
    const double dcomp_dzp = s->dcomp_dzp_prefactor *
                             (rad->prev_xe / (1.0 + rad->prev_xe + s->he_frac)) *
                             (s->Trad - rad->prev_Tk);
    double dxheat_dzp = 0.;
    if (s->use_xray_heating)
        dxheat_dzp = rad->dxheat_dt * s->dt_dzp * 2.0 / 3.0 / s->k_B / (1.0 + rad->prev_xe);
    double dCMBheat_dzp = 0.;
    if (s->use_cmb_heating) { /* Meiksin et al. 2021 */
        const double eps_CMB = (3. / 4.) * (s->Trad / s->T_21) * s->A10 * s->h_frac *
                               (s->h_p * s->h_p / s->lambda_21 / s->lambda_21 / s->m_p) *
                               (1. + 2. * rad->prev_Tk / s->T_21);
        dCMBheat_dzp =
            -eps_CMB * (2. / 3. / s->k_B / (1. + rad->prev_xe)) / s->hubble_zp / (1. + zp);
    }
    double eps_Lya_cont = 0., eps_Lya_inj = 0.;
    if (s->use_lya_heating) {
        /* taugp (:650-653) with hubble(zp) */
        const double tgp = 1.342881e-7 / s->hubble_zp * s->No * pow(1 + zp, 3) * (1.0 + rad->delta) *
                           (1.0 - rad->prev_xe);
        double E_continuum = oracle_lya_heating_efficiency(rad->prev_Tk, rad->prev_Ts, tgp, s->lya_dEC);
        double E_injected = oracle_lya_heating_efficiency(rad->prev_Tk, rad->prev_Ts, tgp, s->lya_dEI);
        if (isnan(E_continuum) || isinf(E_continuum)) E_continuum = 0.;
        if (isnan(E_injected) || isinf(E_injected)) E_injected = 0.;
        const double Ndot_alpha_cont = (4. * M_PI * s->nu_Ly_alpha) / (s->Nb_zp * (1. + rad->delta)) /
                                       (1. + zp) / s->c_cms * rad->dstarlya_cont_dt;
        const double Ndot_alpha_inj = (4. * M_PI * s->nu_Ly_alpha) / (s->Nb_zp * (1. + rad->delta)) /
                                      (1. + zp) / s->c_cms * rad->dstarlya_inj_dt;
        eps_Lya_cont = -Ndot_alpha_cont * E_continuum * (2. / 3. / s->k_B / (1. + rad->prev_xe));
        eps_Lya_inj = -Ndot_alpha_inj * E_injected * (2. / 3. / s->k_B / (1. + rad->prev_xe));
    }

    double x_e = rad->prev_xe + (dxe_dzp * dzp);
    if (x_e > 1)
        x_e = 1 - FRACT_FLOAT_ERR;
    else if (x_e < 0)
        x_e = 0;
    double Tk = rad->prev_Tk;
    if (Tk < (float)C21CM_TS_MAX_TK)
        Tk += (dxheat_dzp + dcomp_dzp + dspec_dzp + dadia_dzp + dCMBheat_dzp + eps_Lya_cont +
               eps_Lya_inj) *
              dzp;
    if (Tk < 0) Tk = s->Trad;

    const double J_alpha_tot = rad->dstarlya_dt + rad->dxlya_dt;
    const double T_inv = 1 / Tk, T_inv_sq = T_inv * T_inv;
    const double xc_fast = (1.0 + rad->delta) * s->xc_inverse *
                           ((1.0 - x_e) * s->No * oracle_kappa_10(Tk) +
                            x_e * s->N_b0 * oracle_kappa_10_elec(Tk) +
                            x_e * s->No * oracle_kappa_10_pH(Tk));
    const double xi_power = s->Ts_prefactor * cbrt((1.0 + rad->delta) * (1.0 - x_e) * T_inv_sq);
    const double xa_tilde_fast_arg =
        s->xa_tilde_prefactor * J_alpha_tot *
        pow(1.0 + 2.98394 * xi_power + 1.53583 * xi_power * xi_power +
                3.85289 * xi_power * xi_power * xi_power,
            -1.);
    const double Trad_inv = 1.0 / s->Trad;
    double TS_fast;
    if (J_alpha_tot > 1.0e-20) {
        double TSold_fast = 0.0, xa_tilde_fast;
        TS_fast = s->Trad;
        while (fabs(TS_fast - TSold_fast) / TS_fast > 1.0e-3) {
            TSold_fast = TS_fast;
            xa_tilde_fast = (1.0 - 0.0631789 * T_inv + 0.115995 * T_inv_sq -
                             0.401403 * T_inv * pow(TS_fast, -1.) +
                             0.336463 * T_inv_sq * pow(TS_fast, -1.)) *
                            xa_tilde_fast_arg;
            TS_fast = (xCMB + xa_tilde_fast + xc_fast) *
                      pow(xCMB * Trad_inv +
                              xa_tilde_fast * (T_inv + 0.405535 * T_inv * pow(TS_fast, -1.) -
                                               0.405535 * T_inv_sq) +
                              xc_fast * T_inv,
                          -1.);
        }
    } else {
        TS_fast = (xCMB + xc_fast) / (xCMB * Trad_inv + xc_fast * T_inv);
    }
    *Ts_out = fabs(TS_fast);
    *Tk_out = Tk;
    *xe_out = x_e;
}

static int ts_check(const c21cm_ts_spec *s) {
    if (!s || s->hii_dim < 1 || s->hii_dim_z < 1 || s->n_step < 1 || s->n_step > C21CM_MAX_TS_RADII)
        return C21CM_VALUE_ERROR;
    if (s->source_mode != C21CM_TS_SRC_GRIDS && s->source_mode != C21CM_TS_SRC_SFRD_TABLE &&
        s->source_mode != C21CM_TS_SRC_FCOLL_TABLES)
        return C21CM_VALUE_ERROR;
    if (s->source_mode == C21CM_TS_SRC_FCOLL_TABLES && !s->no_light &&
        (!s->fcoll_tables || !s->dfcoll_tables))
        return C21CM_VALUE_ERROR;
    if (!s->freq_int_heat || !s->freq_int_ion || !s->freq_int_lya) return C21CM_VALUE_ERROR;
    if (s->use_lya_heating && (!s->lya_dEC || !s->lya_dEI)) return C21CM_VALUE_ERROR;
    if (s->source_mode == C21CM_TS_SRC_SFRD_TABLE && !s->no_light && !s->ln_sfrd_tables)
        return C21CM_VALUE_ERROR;
    return 0;
}

/* interpolation.c:133-157 */
static double table_2d_f(double x, double y, double x_min, double x_width, double y_min,
                         double y_width, const float *z_arr, int ny) {
    int x_idx = (int)floor((x - x_min) / x_width);
    int y_idx = (int)floor((y - y_min) / y_width);
    double x_table = x_min + x_width * (double)x_idx;
    double y_table = y_min + y_width * (double)y_idx;
    double interp_point_x = (x - x_table) / x_width;
    double interp_point_y = (y - y_table) / y_width;
    double left_edge = z_arr[(size_t)x_idx * ny + y_idx] * (1 - interp_point_y) +
                       z_arr[(size_t)x_idx * ny + y_idx + 1] * (interp_point_y);
    double right_edge = z_arr[(size_t)(x_idx + 1) * ny + y_idx] * (1 - interp_point_y) +
                        z_arr[(size_t)(x_idx + 1) * ny + y_idx + 1] * (interp_point_y);
    return left_edge * (1 - interp_point_x) + right_edge * (interp_point_x);
}

/* prepare_filter_boxes with USE_MINI_HALOS (SpinTemperatureBox.c:535-565) */
int oracle_ts_mcrit_grid(const c21cm_mturn_spec *m, double m_turn, const float *J_21_LW,
                         const float *vcb, float *log10_mcrit) {
    const long ntot = (long)m->hii_dim * m->hii_dim * m->hii_dim_z;
    const float z = (float)m->redshift;
#pragma omp parallel for schedule(static)
    for (long ct = 0; ct < ntot; ct++) {
        const float curr_vcb = vcb ? vcb[ct] : (float)m->vcb_const;
        const float curr_j21 = J_21_LW[ct];
        /* lyman_werner_threshold(float z, float J_21_LW, float vcb): thermochem.c:281-304 */
        double mcrit_noLW = 3.314e7 * pow(1. + z, -1.5);
        double f_LW = 1.0 + m->A_LW * pow(curr_j21, m->BETA_LW);
        double f_vcb = pow(1.0 + m->A_VCB * curr_vcb / m->sigma_vcb, m->BETA_VCB);
        double M_buf = mcrit_noLW * f_LW * f_vcb;
        M_buf = fmax(M_buf, m_turn);
        log10_mcrit[ct] = log10(M_buf);
    }
    return C21CM_OK;
}

int oracle_ts_grids(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                    const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                    c21cm_ts_report *report) {
    int status = ts_check(s);
    if (status) return status;
    if (!density || !previous || !previous->spin_temperature || !previous->kinetic_temp_neutral ||
        !previous->xray_ionised_fraction || !out || !out->spin_temperature ||
        !out->kinetic_temp_neutral || !out->xray_ionised_fraction)
        return C21CM_VALUE_ERROR;
    const int lagrangian = s->source_mode == C21CM_TS_SRC_GRIDS;
    if (lagrangian && !s->no_light &&
        (!source_box || !source_box->filtered_sfr || !source_box->filtered_xray))
        return C21CM_VALUE_ERROR;
    if (!lagrangian && !s->no_light && !filtered_density) return C21CM_VALUE_ERROR;
    const int mini = s->use_mini_halos;
    const int mini_tab = mini && !lagrangian; /* E-INTEGRAL: 2-D tables; GRIDS: the mini grids */
    if (mini && (s->source_mode == C21CM_TS_SRC_FCOLL_TABLES || !out->J_21_LW)) return C21CM_VALUE_ERROR;
    if (mini_tab && !s->no_light && (!s->ln_sfrd_tables_mini || !s->filtered_log10_mcrit))
        return C21CM_VALUE_ERROR;
    if (mini && lagrangian && !s->no_light && !source_box->filtered_sfr_mini) return C21CM_VALUE_ERROR;
    /* LYA_MULTIPLE_SCATTERING: the Lyman-Werner sums read the straight-line copies (:1667-1697) */
    const int lw_grids = mini && lagrangian && source_box && source_box->filtered_sfr_lw &&
                         source_box->filtered_sfr_mini_lw;

    const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
    const int nR = s->n_step;
    static const float X[C21CM_X_INT_NXHII] = C21CM_X_INT_XHII;
    float inverse_diff[C21CM_X_INT_NXHII]; /* :868-870 */
    for (int i = 0; i < C21CM_X_INT_NXHII - 1; i++) inverse_diff[i] = 1. / (X[i + 1] - X[i]);

    int *m_xHII_low_box = (int *)malloc(ntot * sizeof(int));
    float *inverse_val_box = (float *)malloc(ntot * sizeof(float));
    double *acc = (double *)calloc(7 * ntot, sizeof(double));
    float *del_fcoll_Rct = lagrangian ? NULL : (float *)malloc(ntot * sizeof(float));
    float *del_fcoll_Rct_MINI = mini_tab ? (float *)malloc(ntot * sizeof(float)) : NULL;
    if (!m_xHII_low_box || !inverse_val_box || !acc || (!lagrangian && !del_fcoll_Rct) ||
        (mini_tab && !del_fcoll_Rct_MINI)) {
        status = C21CM_MEMORY_ALLOC_ERROR;
        goto done;
    }
    double *dxheat_dt_box = acc, *dxion_source_dt_box = acc + ntot, *dxlya_dt_box = acc + 2 * ntot,
           *dstarlya_dt_box = acc + 3 * ntot, *dstarlya_cont_dt_box = acc + 4 * ntot,
           *dstarlya_inj_dt_box = acc + 5 * ntot, *dstarlyLW_dt_box = acc + 6 * ntot;

#pragma omp parallel for schedule(static)
    for (long ct = 0; ct < (long)ntot; ct++) { /* :1499-1522 */
        float xHII_call = previous->xray_ionised_fraction[ct];
        if (xHII_call > X[C21CM_X_INT_NXHII - 1] * 0.999)
            xHII_call = X[C21CM_X_INT_NXHII - 1] * 0.999;
        else if (xHII_call < X[0])
            xHII_call = 1.001 * X[0];
        int m = C21CM_X_INT_NXHII - 1; /* locate_xHII_index, elec_interp.c:415-423 */
        while (xHII_call < X[m]) m--;
        m_xHII_low_box[ct] = m;
        inverse_val_box[ct] = (xHII_call - X[m]) * inverse_diff[m];
    }

    if (!s->no_light) {
        for (int R_ct = nR; R_ct--;) {
            const double z_edge_factor = s->z_edge_factor[R_ct];
            const double xray_R_factor = s->xray_R_factor[R_ct];
            double avg_fix_term = 1., avg_fix_term_MINI = 1.;
            if (mini_tab) { /* calculate_sfrd_from_grid, the molecularly cooled term (:1048-1073) */
                const float *dens_R = filtered_density + (size_t)R_ct * ntot;
                const float *mcrit_R = s->filtered_log10_mcrit + (size_t)R_ct * ntot;
                const float *tab2 = s->ln_sfrd_tables_mini +
                                    (size_t)R_ct * C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
                double ave_m = 0;
#pragma omp parallel for schedule(static) reduction(+ : ave_m)
                for (long ct = 0; ct < (long)ntot; ct++) {
                    const double curr_dens = dens_R[ct] * s->zpp_growth[R_ct];
                    const double curr_mcrit = mcrit_R[ct];
                    const double fcoll_MINI =
                        exp(table_2d_f(curr_dens, curr_mcrit, s->tab_min[R_ct], s->tab_width[R_ct],
                                       s->mturn_tab_min, s->mturn_tab_width, tab2,
                                       C21CM_NMTURN_TABLE));
                    del_fcoll_Rct_MINI[ct] = (1. + curr_dens) * fcoll_MINI;
                    ave_m += fcoll_MINI;
                }
                ave_m /= ntot;
                if (report) report->ave_sfrd_mini[R_ct] = ave_m;
                avg_fix_term_MINI = s->mean_sfr_zpp_mini[R_ct] / ave_m; /* :1617-1618 */
            }
            if (!lagrangian) { /* calculate_sfrd_from_grid with tables (:1040-1079) */
                const float *dens_R = filtered_density + (size_t)R_ct * ntot;
                const int e_integral = s->source_mode == C21CM_TS_SRC_SFRD_TABLE;
                const float *tab = (e_integral ? s->ln_sfrd_tables : s->fcoll_tables) +
                                   (size_t)R_ct * C21CM_NDELTA_TABLE;
                const float *dtab = e_integral ? NULL : s->dfcoll_tables + (size_t)R_ct * C21CM_NDELTA_TABLE;
                double ave = 0;
#pragma omp parallel for schedule(static) reduction(+ : ave)
                for (long ct = 0; ct < (long)ntot; ct++) {
                    const double curr_dens = dens_R[ct] * s->zpp_growth[R_ct];
                    double fcoll = table_1d(curr_dens, s->tab_min[R_ct], s->tab_width[R_ct], tab);
                    if (e_integral) {
                        fcoll = exp(fcoll);
                        del_fcoll_Rct[ct] = (1. + curr_dens) * fcoll;
                    } else { /* CONST-ION-EFF: the source is dfcoll/dz, the mean is fcoll's */
                        const double dfcoll =
                            table_1d(curr_dens, s->tab_min[R_ct], s->tab_width[R_ct], dtab);
                        del_fcoll_Rct[ct] = (1. + curr_dens) * dfcoll;
                    }
                    ave += fcoll;
                }
                ave /= ntot;
                if (report) report->ave_sfrd[R_ct] = ave;
                avg_fix_term = s->mean_sfr_zpp[R_ct] / ave;
            }
            const float *sfr_R = lagrangian ? source_box->filtered_sfr + (size_t)R_ct * ntot : NULL;
            const float *xray_R = lagrangian ? source_box->filtered_xray + (size_t)R_ct * ntot : NULL;
#pragma omp parallel for schedule(static)
            for (long ct = 0; ct < (long)ntot; ct++) {
                double sfr_term, xray_sfr;
                if (lagrangian) {
                    sfr_term = sfr_R[ct] * z_edge_factor;
                    xray_sfr = xray_R[ct] * z_edge_factor * xray_R_factor * 1e38;
                } else {
                    sfr_term = del_fcoll_Rct[ct] * z_edge_factor * avg_fix_term * s->sfr_scale;
                    xray_sfr = sfr_term * s->xray_scale * xray_R_factor;
                }
                double sfr_term_mini = 0;
                if (mini && lagrangian) { /* :1692-1701: mini-halo X-rays are already in filtered_xray */
                    const size_t o = (size_t)R_ct * ntot + ct;
                    sfr_term_mini = source_box->filtered_sfr_mini[o] * z_edge_factor;
                    const double sfr_term_lw =
                        lw_grids ? source_box->filtered_sfr_lw[o] * z_edge_factor : sfr_term;
                    const double sfr_term_mini_lw =
                        lw_grids ? source_box->filtered_sfr_mini_lw[o] * z_edge_factor : sfr_term_mini;
                    dstarlyLW_dt_box[ct] += sfr_term_lw * s->lw_prefactor[R_ct] +
                                            sfr_term_mini_lw * s->lw_prefactor_mini[R_ct];
                } else if (mini) { /* :1702-1716 */
                    sfr_term_mini =
                        del_fcoll_Rct_MINI[ct] * z_edge_factor * avg_fix_term_MINI * s->sfr_scale_mini;
                    xray_sfr += sfr_term_mini * s->xray_scale_mini * xray_R_factor;
                    dstarlyLW_dt_box[ct] += sfr_term * s->lw_prefactor[R_ct] +
                                            sfr_term_mini * s->lw_prefactor_mini[R_ct];
                }
                const double starlya_factor_mini = mini ? s->starlya_prefactor_mini[R_ct] : 0.;
                const double lyacont_factor_mini =
                    (mini && s->use_lya_heating) ? s->lya_cont_prefactor_mini[R_ct] : 0.;
                const double lyainj_factor_mini =
                    (mini && s->use_lya_heating) ? s->lya_inj_prefactor_mini[R_ct] : 0.;
                const int xidx = m_xHII_low_box[ct];
                const double ival = inverse_val_box[ct];
#define FREQ(tbl) \
    ((s->tbl[(xidx + 1) * nR + R_ct] - s->tbl[xidx * nR + R_ct]) * ival + s->tbl[xidx * nR + R_ct])
                if (s->use_xray_heating) dxheat_dt_box[ct] += xray_sfr * FREQ(freq_int_heat);
                dxion_source_dt_box[ct] += xray_sfr * FREQ(freq_int_ion);
                dxlya_dt_box[ct] += xray_sfr * FREQ(freq_int_lya);
#undef FREQ
                dstarlya_dt_box[ct] +=
                    sfr_term * s->starlya_prefactor[R_ct] + sfr_term_mini * starlya_factor_mini;
                if (s->use_lya_heating) {
                    dstarlya_cont_dt_box[ct] += sfr_term * s->lya_cont_prefactor[R_ct] +
                                                sfr_term_mini * lyacont_factor_mini;
                    dstarlya_inj_dt_box[ct] += sfr_term * s->lya_inj_prefactor[R_ct] +
                                               sfr_term_mini * lyainj_factor_mini;
                }
            }
        }
    }

    double J_alpha_ave = 0, xheat_ave = 0, xion_ave = 0, Ts_ave = 0, Tk_ave = 0, x_e_ave = 0;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : J_alpha_ave, xheat_ave, xion_ave, Ts_ave, Tk_ave, x_e_ave) reduction(| : bad)
    for (long ct = 0; ct < (long)ntot; ct++) { /* :1794-1848 */
        double curr_delta = density[ct] * s->growth_ratio;
        if (curr_delta <= -1) curr_delta = -1 + FRACT_FLOAT_ERR;
        rad_terms rad;
        memset(&rad, 0, sizeof(rad));
        if (s->use_xray_heating) rad.dxheat_dt = dxheat_dt_box[ct] * s->xray_prefactor * s->volunit_inv;
        rad.dxion_dt = dxion_source_dt_box[ct] * s->xray_prefactor * s->volunit_inv;
        rad.dxlya_dt =
            dxlya_dt_box[ct] * s->xray_prefactor * s->volunit_inv * s->Nb_zp * (1 + curr_delta);
        rad.dstarlya_dt = dstarlya_dt_box[ct] * s->lya_star_prefactor * s->volunit_inv;
        rad.delta = curr_delta;
        if (s->use_lya_heating) {
            rad.dstarlya_cont_dt = dstarlya_cont_dt_box[ct] * s->lya_star_prefactor * s->volunit_inv;
            rad.dstarlya_inj_dt = dstarlya_inj_dt_box[ct] * s->lya_star_prefactor * s->volunit_inv;
        }
        rad.prev_Ts = previous->spin_temperature[ct];
        rad.prev_Tk = previous->kinetic_temp_neutral[ct];
        rad.prev_xe = previous->xray_ionised_fraction[ct];
        double Ts, Tk, xe;
        ts_cell(s, &rad, &Ts, &Tk, &xe);
        out->spin_temperature[ct] = Ts;
        out->kinetic_temp_neutral[ct] = Tk;
        out->xray_ionised_fraction[ct] = xe;
        if (mini) /* :1843-1845, 1324 */
            out->J_21_LW[ct] =
                dstarlyLW_dt_box[ct] * s->lya_star_prefactor * s->volunit_inv * s->h_p * 1e21;
        if (isfinite(out->spin_temperature[ct]) == 0) bad |= 1;
        J_alpha_ave += rad.dxlya_dt + rad.dstarlya_dt;
        xheat_ave += rad.dxheat_dt;
        xion_ave += rad.dxion_dt;
        Ts_ave += Ts;
        Tk_ave += Tk;
        x_e_ave += xe;
    }
    if (report) {
        report->Ts_ave = Ts_ave / (double)ntot;
        report->Tk_ave = Tk_ave / (double)ntot;
        report->x_e_ave = x_e_ave / (double)ntot;
        report->J_alpha_ave = J_alpha_ave / (double)ntot;
        report->xheat_ave = xheat_ave / (double)ntot;
        report->xion_ave = xion_ave / (double)ntot;
    }
    if (bad) status = C21CM_INFINITY_OR_NAN_ERROR;
done:
    free(m_xHII_low_box);
    free(inverse_val_box);
    free(acc);
    free(del_fcoll_Rct);
    free(del_fcoll_Rct_MINI);
    return status;
}

/* init_first_Ts (:892-927) with get_Ts's collisions-only branch (heating_helper_progs.c:736-760);
 * the rates are taken at the mean temperature TK and the redshift of the perturbed field */
int oracle_ts_first_grids(const c21cm_ts_first_spec *s, const float *density, TsBox *out) {
    if (!s || !density || !out || !out->spin_temperature || !out->kinetic_temp_neutral ||
        !out->xray_ionised_fraction)
        return C21CM_VALUE_ERROR;
    const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
    const float z = (float)s->perturbed_redshift; /* get_Ts(float z, float delta, float TK, float xe) */
    const float xe = (float)s->xe, TK = (float)s->TK;
    const double Trad = s->T_cmb * (1.0 + z);
    const double k_HI = oracle_kappa_10(TK), k_e = oracle_kappa_10_elec(TK), k_p = oracle_kappa_10_pH(TK);
#pragma omp parallel for schedule(static)
    for (long ct = 0; ct < (long)ntot; ct++) {
        const double gdens = density[ct] * s->inverse_growth_factor_z * s->growth_factor_zp;
        out->kinetic_temp_neutral[ct] = s->TK * (1.0 + s->cT_ad * gdens);
        out->xray_ionised_fraction[ct] = s->xe;
        const float delta = gdens; /* get_Ts takes floats */
        const double nH = (1.0 - xe) * s->No * pow(1.0 + z, 3.0) * (1.0 + delta);
        const double ne = xe * s->N_b0 * pow(1.0 + z, 3.0) * (1.0 + delta);
        const double np = xe * s->No * pow(1.0 + z, 3.0) * (1.0 + delta);
        const double xc = s->T_21 / Trad * nH * k_HI / s->A10 + s->T_21 / Trad * ne * k_e / s->A10 +
                          s->T_21 / Trad * np * k_p / s->A10;
        out->spin_temperature[ct] = (float)((1.0 + xc) / (1.0 / Trad + xc / TK));
    }
    return 0;
}
