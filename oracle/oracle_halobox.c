/*
 * oracle_halobox.c -- CPU restatement of ComputeHaloBox's integrated branch.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  reference:
 *   src/py21cmfast/src/map_mass.c:214-344   move_grid_galprops (positions, prefactors, deposit)
 *   src/py21cmfast/src/map_mass.c:62-98     do_cic_interpolation_float (float boxes, atomic adds)
 *   src/py21cmfast/src/HaloBox.c:244-262    get_cell_integrals (no mini-halos / X-rays)
 *   src/py21cmfast/src/interp_tables.c:960-1001 + interpolation.c:123-131  exp(lerp(ln-table))
 * The two ln-tables come in through the spec (the host quadrature that fills them is pinned
 * separately, tests/test_host_scalars.py).
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
    if (mini && (hires || !s->log10_mturn_acg || !s->log10_mturn_mcg || !s->ln_nion_table2d ||
                 !s->ln_nion_mini_table2d || !s->ln_sfrd_mini_table2d || !grids->halo_sfr_mini))
        return C21CM_VALUE_ERROR;
    if (mini)
        for (size_t i = 0; i < n_out; i++) grids->halo_sfr_mini[i] = 0.f;
    const int xray = (mini ? s->ln_xray_table2d : s->ln_xray_table) && grids->halo_xray; /* USE_TS_FLUCT, HaloBox.c:279-283 */
    if (xray)
        for (size_t i = 0; i < n_out; i++) grids->halo_xray[i] = 0.f;
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
