/*
 * oracle_halobox.c -- CPU restatement of ComputeHaloBox's integrated branch.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  reference:
 *   src/py21cmfast/src/map_mass.c:214-344   move_grid_galprops (positions, prefactors, deposit)
 *   src/py21cmfast/src/map_mass.c:62-98     do_cic_interpolation_float (float boxes, atomic adds)
 *   src/py21cmfast/src/HaloBox.c:244-262    get_cell_integrals (no mini-halos / X-rays)
 *   src/py21cmfast/src/interp_tables.c:960-1001 + interpolation.c:123-131  exp(lerp(ln-table))
 *   src/py21cmfast/src/map_mass.c:346-476   move_halo_galprops (halo catalogue -> grids)
 *   src/py21cmfast/src/HaloBox.c:62-102     set_halo_properties
 *   src/py21cmfast/src/scaling_relations.c:277-283,315-325,331-500  per-halo scaling relations
 * The two ln-tables come in through the spec (the host quadrature that fills them is pinned
 * separately, tests/test_host_scalars.py).  The halo-catalogue branch runs in catalogue order
 * (upstream with N_THREADS = 1; its float adds depend on the thread order otherwise) and is
 * PARITY UNPINNED: the reference holds no HaloBox vector.
 */
#include <math.h>
#include <stddef.h>

#include "oracle.h"

static double table_1d_f(double x, double x_min, double x_width, const float *y_arr) {
    const int idx = (int)floor((x - x_min) / x_width);
    const double table_val = x_min + x_width * (double)(float)idx;
    const double interp_point = (x - table_val) / x_width;
    return y_arr[idx] * (1 - interp_point) + y_arr[idx + 1] * interp_point;
}

/* interpolation.c:133-157 */
static double table_2d_f(double x, double y, double x_min, double x_width, double y_min,
                         double y_width, const float *z_arr) {
    const int ny = C21CM_NMTURN_TABLE;
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

static int wrapi(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

static void cic_float(float *box, const double pos[3], const int dim[3], double value) {
    int ipos[3], iposp1[3];
    double dist[3];
    for (int a = 0; a < 3; a++) {
        ipos[a] = (int)floor(pos[a]);
        iposp1[a] = ipos[a] + 1;
        dist[a] = pos[a] - ipos[a];
        ipos[a] = wrapi(ipos[a], dim[a]);
        iposp1[a] = wrapi(iposp1[a], dim[a]);
    }
    for (int c = 0; c < 8; c++) {
        const int ix = (c & 1) ? iposp1[0] : ipos[0];
        const int iy = (c & 2) ? iposp1[1] : ipos[1];
        const int iz = (c & 4) ? iposp1[2] : ipos[2];
        const double w = ((c & 1) ? dist[0] : 1. - dist[0]) * ((c & 2) ? dist[1] : 1. - dist[1]) *
                         ((c & 4) ? dist[2] : 1. - dist[2]);
        const size_t idx = (size_t)iz + (size_t)dim[2] * ((size_t)iy + (size_t)dim[1] * ix);
#pragma omp atomic update
        box[idx] += value * w;
    }
}

static double cic_read_f(const float *box, const double pos[3], const int dim[3]) { /* map_mass.c:102-139 */
    int ipos[3], iposp1[3];
    double dist[3], sum = 0;
    for (int a = 0; a < 3; a++) {
        ipos[a] = (int)floor(pos[a]);
        iposp1[a] = ipos[a] + 1;
        dist[a] = pos[a] - ipos[a];
        ipos[a] = wrapi(ipos[a], dim[a]);
        iposp1[a] = wrapi(iposp1[a], dim[a]);
    }
    for (int c = 0; c < 8; c++) {
        const int ix = (c & 1) ? iposp1[0] : ipos[0];
        const int iy = (c & 2) ? iposp1[1] : ipos[1];
        const int iz = (c & 4) ? iposp1[2] : ipos[2];
        const double w = ((c & 1) ? dist[0] : 1. - dist[0]) * ((c & 2) ? dist[1] : 1. - dist[1]) *
                         ((c & 4) ? dist[2] : 1. - dist[2]);
        sum += w * box[(size_t)iz + (size_t)dim[2] * ((size_t)iy + (size_t)dim[1] * ix)];
    }
    return sum;
}

static double lx_on_sfr(double metallicity, double lx_constant, int upper) { /* scaling_relations.c:277-283,315-325 */
    if (!upper) return lx_constant;
    const double hi_z_index = -0.64, lo_z_index = 0., z_pivot = 0.05;
    return lx_constant * (1. / (pow(metallicity / z_pivot, -lo_z_index) + pow(metallicity / z_pivot, -hi_z_index)));
}

typedef struct {
    double n_ion, sfr, sfr_mini, xray, wsfr, stars, stars_mini, metallicity;
} halo_props;

/* set_halo_properties (HaloBox.c:62-102) */
static halo_props halo_properties(double M, double M_turn_a, double M_turn_m, const c21cm_halo_consts *c,
                                  const double rng[3]) {
    const double s_per_yr = 31556925.9747; /* Constants.c:16 */
    /* get_halo_stellarmass (scaling_relations.c:331-400) */
    const double adj_star = c->scaling_median ? 0 : c->sigma_star * c->sigma_star / 2.;
    double mu_fstar;
    if (c->upper_stellar_turnover && c->alpha_star > c->alpha_upper)
        mu_fstar = c->fstar_10 * (c->upper_pivot_ratio / (pow(M / c->pivot_upper, -c->alpha_star) +
                                                          pow(M / c->pivot_upper, -c->alpha_upper)));
    else
        mu_fstar = c->fstar_10 * pow(M / 1e10, c->alpha_star);
    double f = mu_fstar * exp(-M_turn_a / M + rng[0] * c->sigma_star - adj_star);
    if (f > 1.) f = 1.;
    const double stars = f * M * c->baryon_ratio;
    double stars_mini = 0.;
    if (c->use_mini_halos) {
        double fm = c->fstar_7 * pow(M / 1e7, c->alpha_star_mini) *
                    exp(-M_turn_m / M - M / c->acg_thresh + rng[0] * c->sigma_star - adj_star);
        if (fm > 1.) fm = 1.;
        stars_mini = fm * M * c->baryon_ratio;
    }
    /* get_halo_sfr (:402-444) */
    double sigma_sfr = 0.;
    if (c->sigma_sfr_lim > 0.) {
        sigma_sfr = c->sigma_sfr_idx * log10((stars + stars_mini) / 1e10) + c->sigma_sfr_lim;
        if (sigma_sfr < c->sigma_sfr_lim) sigma_sfr = c->sigma_sfr_lim;
    }
    const double adj_sfr = c->scaling_median ? 0 : sigma_sfr * sigma_sfr / 2.;
    halo_props p;
    p.sfr = stars / (c->t_star * c->t_h) * exp(rng[1] * sigma_sfr - adj_sfr);
    p.sfr_mini = c->use_mini_halos ? stars_mini / (c->t_star * c->t_h) * exp(rng[1] * sigma_sfr - adj_sfr) : 0.;
    p.xray = 0., p.metallicity = 0.;
    p.stars = stars, p.stars_mini = stars_mini;
    if (c->use_xray) { /* get_halo_metallicity, get_halo_xray (:446-500) */
        const double sfr_t = p.sfr + p.sfr_mini, stars_t = stars + stars_mini;
        double stellar_term = 1.;
        if (stars_t > 0 && sfr_t > 0.) {
            const double M0 = 1.28825e10 * pow(sfr_t * s_per_yr, 0.56);
            stellar_term = pow(1 + pow(stars_t / M0, -2.1), -0.148);
        }
        const double Z = 1.23 * stellar_term * pow(10, -0.056 * c->redshift + 0.064);
        p.metallicity = Z;
        double mu_x = lx_on_sfr(Z, c->l_x, c->upper_stellar_turnover) * (p.sfr * s_per_yr);
        if (c->use_mini_halos) mu_x += lx_on_sfr(Z, c->l_x_mini, c->upper_stellar_turnover) * (p.sfr_mini * s_per_yr);
        const double adj_x = c->scaling_median ? 0 : c->sigma_xray * c->sigma_xray / 2.;
        p.xray = mu_x * exp(rng[2] * c->sigma_xray - adj_x);
    }
    const double fesc = fmin(c->fesc_10 * pow(M / 1e10, c->alpha_esc), 1);
    const double fesc_mini = c->use_mini_halos ? fmin(c->fesc_7 * pow(M / 1e7, c->alpha_esc), 1) : 0.;
    p.n_ion = stars * c->pop2_ion * fesc + stars_mini * c->pop3_ion * fesc_mini;
    p.wsfr = p.sfr * c->pop2_ion * fesc + p.sfr_mini * c->pop3_ion * fesc_mini;
    return p;
}

/* move_halo_galprops (map_mass.c:346-476) */
static void deposit_halos(const c21cm_halobox_spec *s, const float *const vel[3], const float *const vel2[3],
                          const int vel_dim[3], const int out_dim[3], HaloBox *grids, int xray) {
    const HaloCatalog *h = s->halos;
    const c21cm_halo_consts *c = s->halo_consts;
    const double box_size[3] = {s->box_len, s->box_len, s->box_len_z};
    const double cell_size_inv_v = vel_dim[0] / s->box_len, cell_size_inv_o = out_dim[0] / s->box_len;
    const double cell_vol_inv = cell_size_inv_o * cell_size_inv_o * cell_size_inv_o;
    const double D = s->growth_factor, Di = s->init_growth_factor;
    const double vdf = D - Di, vdf2 = -(3.0 / 7.0) * D * D - (-(3.0 / 7.0) * Di * Di);
    for (unsigned long long i = 0; i < h->n_halos; i++) {
        const double hmass = h->halo_masses[i];
        if (hmass == 0.) continue;
        double pos[3] = {h->halo_coords[3 * i], h->halo_coords[3 * i + 1], h->halo_coords[3 * i + 2]};
        int ip[3];
        for (int a = 0; a < 3; a++) ip[a] = wrapi((int)(pos[a] * cell_size_inv_v + 0.5), vel_dim[a]);
        const size_t vi = (size_t)ip[2] + (size_t)vel_dim[2] * ((size_t)ip[1] + (size_t)vel_dim[1] * ip[0]);
        for (int a = 0; a < 3; a++) {
            pos[a] += vel[a][vi] * vdf;
            if (s->lpt2) pos[a] -= vel2[a][vi] * vdf2;
        }
        for (int a = 0; a < 3; a++) pos[a] = pos[a] * out_dim[a] / box_size[a];
        double M_turn_a = c->mturn_a_nofb, M_turn_m = c->mturn_m_nofb;
        if (c->use_mini_halos) {
            M_turn_a = pow(10, cic_read_f(s->log10_mturn_acg, pos, out_dim));
            M_turn_m = pow(10, cic_read_f(s->log10_mturn_mcg, pos, out_dim));
        }
        const double rng[3] = {h->star_rng[i], h->sfr_rng[i], xray ? h->xray_rng[i] : 0.};
        const halo_props p = halo_properties(hmass, M_turn_a, M_turn_m, c, rng);
        cic_float(grids->halo_sfr, pos, out_dim, p.sfr);
        cic_float(grids->n_ion, pos, out_dim, p.n_ion);
        if (c->use_mini_halos) cic_float(grids->halo_sfr_mini, pos, out_dim, p.sfr_mini);
        if (xray) cic_float(grids->halo_xray, pos, out_dim, p.xray);
        if (grids->whalo_sfr) cic_float(grids->whalo_sfr, pos, out_dim, p.wsfr);
    }
    const size_t n_out = (size_t)out_dim[0] * out_dim[1] * out_dim[2];
    for (size_t i = 0; i < n_out; i++) {
        grids->n_ion[i] *= cell_vol_inv;
        grids->halo_sfr[i] *= cell_vol_inv;
        if (xray) grids->halo_xray[i] *= cell_vol_inv;
        if (grids->whalo_sfr) grids->whalo_sfr[i] *= cell_vol_inv;
        if (c->use_mini_halos) grids->halo_sfr_mini[i] *= cell_vol_inv;
    }
}

int oracle_halobox_grids(const c21cm_halobox_spec *s, const InitialConditions *ics,
                         HaloBox *grids) {
    if (!s || !ics || !grids || !grids->n_ion || !grids->halo_sfr) return C21CM_VALUE_ERROR;
    const int hires = s->perturb_on_high_res;
    const int dens_dim[3] = {hires ? s->dim : s->hii_dim, hires ? s->dim : s->hii_dim,
                             hires ? s->dim_z : s->hii_dim_z};
    const int out_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const float *dens = hires ? ics->hires_density : ics->lowres_density;
    const float *vel[3] = {hires ? ics->hires_vx : ics->lowres_vx, hires ? ics->hires_vy : ics->lowres_vy,
                           hires ? ics->hires_vz : ics->lowres_vz};
    const float *vel2[3] = {hires ? ics->hires_vx_2LPT : ics->lowres_vx_2LPT,
                            hires ? ics->hires_vy_2LPT : ics->lowres_vy_2LPT,
                            hires ? ics->hires_vz_2LPT : ics->lowres_vz_2LPT};
    if (!dens || !vel[0] || !vel[1] || !vel[2]) return C21CM_VALUE_ERROR;
    if (s->lpt2 && (!vel2[0] || !vel2[1] || !vel2[2])) return C21CM_VALUE_ERROR;
    const size_t n_out = (size_t)out_dim[0] * out_dim[1] * out_dim[2];
    for (size_t i = 0; i < n_out; i++) { /* HaloBox.c:583-586 */
        grids->n_ion[i] = 0.f;
        grids->halo_sfr[i] = 0.f;
    }
    const int mini = s->use_mini_halos; /* HaloBox.c:271-277, map_mass.c:289-293,312-315 */
    const int integral = !s->skip_integral;
    if (mini && (hires || !s->log10_mturn_acg || !s->log10_mturn_mcg || !grids->halo_sfr_mini ||
                 (integral && (!s->ln_nion_table2d || !s->ln_nion_mini_table2d || !s->ln_sfrd_mini_table2d))))
        return C21CM_VALUE_ERROR;
    if (mini)
        for (size_t i = 0; i < n_out; i++) grids->halo_sfr_mini[i] = 0.f;
    const int xray = grids->halo_xray && (integral ? (mini ? s->ln_xray_table2d : s->ln_xray_table) != NULL
                                                    : (s->halo_consts && s->halo_consts->use_xray)); /* USE_TS_FLUCT, HaloBox.c:279-283 */
    if (xray)
        for (size_t i = 0; i < n_out; i++) grids->halo_xray[i] = 0.f;
    if (grids->whalo_sfr)
        for (size_t i = 0; i < n_out; i++) grids->whalo_sfr[i] = 0.f;
    if (s->halos && s->halos->n_halos) { /* HaloBox.c:622-625 */
        if (!s->halo_consts) return C21CM_VALUE_ERROR;
        deposit_halos(s, vel, vel2, dens_dim, out_dim, grids, xray);
    }
    if (!integral) return 0; /* :635 */
    const double box_size[3] = {s->box_len, s->box_len, s->box_len_z};
    const double dim_ratio_out = (double)out_dim[0] / (double)dens_dim[0];
    const double D = s->growth_factor, Di = s->init_growth_factor;
    const double d2 = -(3.0 / 7.0) * D * D, d2i = -(3.0 / 7.0) * Di * Di;
    double vdf[3], vdf2[3];
    for (int a = 0; a < 3; a++) {
        vdf[a] = (D - Di) / box_size[a] * dens_dim[a];
        vdf2[a] = (d2 - d2i) / box_size[a] * dens_dim[a];
    }
#pragma omp parallel for collapse(2)
    for (int i = 0; i < dens_dim[0]; i++) {
        for (int j = 0; j < dens_dim[1]; j++) {
            for (int k = 0; k < dens_dim[2]; k++) {
                /* velocities live on the same grid as the density here (dim_ratio_vel = 1) */
                const size_t idx = (size_t)k + (size_t)dens_dim[2] * ((size_t)j + (size_t)dens_dim[1] * i);
                double pos[3] = {i, j, k};
                for (int a = 0; a < 3; a++) {
                    pos[a] += vel[a][idx] * vdf[a];
                    if (s->lpt2) pos[a] -= vel2[a][idx] * vdf2[a];
                    pos[a] *= dim_ratio_out;
                }
                const double curr_dens = dens[idx] * D;
                const double sfrd = exp(table_1d_f(curr_dens, s->tab_min, s->tab_width, s->ln_sfrd_table));
                if (mini) { /* get_cell_integrals with the cell's turnover masses */
                    const double l10_mturn_a = s->log10_mturn_acg[idx], l10_mturn_m = s->log10_mturn_mcg[idx];
                    const double nion_a = exp(table_2d_f(curr_dens, l10_mturn_a, s->tab_min, s->tab_width,
                                                         s->mta_min, s->mta_width, s->ln_nion_table2d));
                    const double nion_m = exp(table_2d_f(curr_dens, l10_mturn_m, s->tab_min, s->tab_width,
                                                         s->mtm_min, s->mtm_width, s->ln_nion_mini_table2d));
                    const double sfrd_m =
                        exp(table_2d_f(curr_dens, l10_mturn_m, s->tab_min, s->tab_width, s->mt_fixed_min,
                                       s->mt_fixed_width, s->ln_sfrd_mini_table2d));
                    cic_float(grids->halo_sfr, pos, out_dim, sfrd * s->prefactor_sfr);
                    cic_float(grids->n_ion, pos, out_dim,
                              nion_a * s->prefactor_nion + nion_m * s->prefactor_nion_mini);
                    cic_float(grids->halo_sfr_mini, pos, out_dim, sfrd_m * s->prefactor_sfr_mini);
                    if (xray) {
                        const double lx =
                            exp(table_2d_f(curr_dens, l10_mturn_m, s->tab_min, s->tab_width, s->mt_fixed_min,
                                           s->mt_fixed_width, s->ln_xray_table2d));
                        cic_float(grids->halo_xray, pos, out_dim, lx * s->prefactor_xray);
                    }
                    continue;
                }
                const double nion = exp(table_1d_f(curr_dens, s->tab_min, s->tab_width, s->ln_nion_table));
                cic_float(grids->halo_sfr, pos, out_dim, sfrd * s->prefactor_sfr);
                cic_float(grids->n_ion, pos, out_dim, nion * s->prefactor_nion);
                if (xray) { /* map_mass.c:316-319 */
                    const double lx =
                        exp(table_1d_f(curr_dens, s->tab_min, s->tab_width, s->ln_xray_table));
                    cic_float(grids->halo_xray, pos, out_dim, lx * s->prefactor_xray);
                }
            }
        }
    }
    if (grids->whalo_sfr)
        for (size_t i = 0; i < n_out; i++) grids->whalo_sfr[i] = grids->n_ion[i] * s->prefactor_wsfr;
    return 0;
}

/* get_log10_turnovers (HaloBox.c:465-516).  M_turn_a is declared once per OpenMP thread (:481) and
 * updated with fmax(M_turn_a, ...) (:497): within a thread's share of the cells it is a running
 * maximum, so the grid depends on how the loop is divided.  `n_threads` contiguous shares as
 * libgomp's static schedule deals them (the first N mod T threads get one cell more).
 * below_z_heat_max: :488-492 (above it J_21_LW, Gamma_12 and z_reion count as zero). */
int oracle_halobox_turnovers(const c21cm_mturn_spec *m, double m_turn, int below_z_heat_max,
                             int n_threads, const float *prev_G12, const float *prev_z_reion,
                             const float *J_21_LW, const float *vcb, float *mturn_a_grid,
                             float *mturn_m_grid, double averages[2]) {
    const long ntot = (long)m->hii_dim * m->hii_dim * m->hii_dim_z;
    if (n_threads < 1) n_threads = 1;
    double log10_mturn_m_avg = 0., log10_mturn_a_avg = 0.;
    const float z = (float)m->redshift;
    const long q = ntot / n_threads, r = ntot % n_threads;
    long start = 0;
    for (int t = 0; t < n_threads; t++) {
        const long len = q + (t < r ? 1 : 0);
        double M_turn_a = m->mturn_a_nofb;
        for (long i = start; i < start + len; i++) {
            const double curr_vcb = vcb ? vcb[i] : m->vcb_const;
            double J21_val = 0., Gamma12_val = 0., zre_val = 0.;
            if (below_z_heat_max) {
                J21_val = J_21_LW[i];
                Gamma12_val = prev_G12[i];
                zre_val = prev_z_reion[i];
            }
            /* lyman_werner_threshold(float, float, float), reionization_feedback(float x 3) */
            const float j = (float)J21_val, v = (float)curr_vcb, g = (float)Gamma12_val, zin = (float)zre_val;
            double M_turn_m = 3.314e7 * pow(1. + z, -1.5) * (1.0 + m->A_LW * pow(j, m->BETA_LW)) *
                              pow(1.0 + m->A_VCB * v / m->sigma_vcb, m->BETA_VCB);
            double M_turn_r = 1e-40;
            if (!(zin <= 1e-19))
                M_turn_r = 3e9 * pow(2.0 * g, 0.17) * pow((1. + z) / 10, -2.1) *
                           pow(1 - pow((1. + z) / (1. + zin), 2.0), 2.5);
            M_turn_a = fmax(M_turn_a, fmax(M_turn_r, m_turn));
            M_turn_m = fmax(M_turn_m, fmax(M_turn_r, m_turn));
            mturn_a_grid[i] = log10(M_turn_a);
            log10_mturn_a_avg += log10(M_turn_a);
            mturn_m_grid[i] = log10(M_turn_m);
            log10_mturn_m_avg += log10(M_turn_m);
        }
        start += len;
    }
    averages[0] = log10_mturn_a_avg / ntot;
    averages[1] = log10_mturn_m_avg / ntot;
    return 0;
}

/* test_halo_props (HaloBox.c:658-779).  lw = {A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb, vcb_const,
 * M_TURN}; lyman_werner_threshold / reionization_feedback as in thermochem.c (float arguments). */
int oracle_halo_props(const c21cm_halo_consts *c, unsigned long long n_halos, const float *masses,
                      const float *coords, const float *star_rng, const float *sfr_rng,
                      const float *xray_rng, const int dim[3], double cell_length, double redshift,
                      int below_z_heat_max, int vcb_flucts, const double lw[7], const float *vcb,
                      const float *J21, const float *z_re, const float *G12, float *out) {
    for (unsigned long long i = 0; i < n_halos; i++) {
        const double m = masses[i];
        if (m == 0.) continue;
        double M_turn_a = c->mturn_a_nofb, M_turn_m = c->mturn_m_nofb, M_turn_r = 0.;
        if (c->use_mini_halos) {
            int cell[3];
            for (int a = 0; a < 3; a++) {
                double pos = coords[a + 3 * i] / cell_length;
                if (pos == (float)dim[0]) pos = (float)dim[0] - 0.1;
                cell[a] = (int)pos;
            }
            const size_t ic = (size_t)cell[2] + (size_t)dim[2] * ((size_t)cell[1] + (size_t)dim[1] * cell[0]);
            const float vc = vcb_flucts ? vcb[ic] : (float)lw[5];
            float j = 0.f, g = 0.f, zin = 0.f;
            if (below_z_heat_max) j = J21[ic], g = G12[ic], zin = z_re[ic];
            const float z = (float)redshift;
            M_turn_m = 3.314e7 * pow(1. + z, -1.5) * (1.0 + lw[0] * pow(j, lw[1])) *
                       pow(1.0 + lw[2] * vc / lw[4], lw[3]);
            M_turn_r = zin <= 1e-19 ? 1e-40
                                    : 3e9 * pow(2.0 * g, 0.17) * pow((1. + z) / 10, -2.1) *
                                          pow(1 - pow((1. + z) / (1. + zin), 2.0), 2.5);
            M_turn_a = fmax(M_turn_a, fmax(M_turn_r, lw[6]));
            M_turn_m = fmax(M_turn_m, fmax(M_turn_r, lw[6]));
        }
        const double rng[3] = {star_rng[i], sfr_rng[i], xray_rng[i]};
        const halo_props p = halo_properties(m, M_turn_a, M_turn_m, c, rng);
        float *o = out + 12 * i;
        o[0] = m, o[1] = p.stars, o[2] = p.sfr, o[3] = p.xray, o[4] = p.n_ion, o[5] = p.wsfr;
        o[6] = p.stars_mini, o[7] = p.sfr_mini, o[8] = M_turn_a, o[9] = M_turn_m, o[10] = M_turn_r;
        o[11] = p.metallicity;
    }
    return 0;
}
