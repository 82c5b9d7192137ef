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
    const int xray = s->ln_xray_table && grids->halo_xray; /* USE_TS_FLUCT, HaloBox.c:279-283 */
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
                const double nion = exp(table_1d_f(curr_dens, s->tab_min, s->tab_width, s->ln_nion_table));
                const double sfrd = exp(table_1d_f(curr_dens, s->tab_min, s->tab_width, s->ln_sfrd_table));
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
