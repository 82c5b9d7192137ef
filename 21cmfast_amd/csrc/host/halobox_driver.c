/*
 * halobox_driver.c -- C host driver of ComputeHaloBox's integrated ("fixed grid") branch:
 * set_fixed_grids + move_grid_galprops (src/py21cmfast/src/HaloBox.c:302-436,
 * src/py21cmfast/src/map_mass.c:214-344) without mini-halos, X-rays or the extra fields.
 * The per-cell integrals are table lookups (the two ln-tables arrive in the spec), the move to
 * Eulerian positions is the LDS-tiled CIC deposit shared with ComputePerturbedField.
 * It shares PerturbedField's staging slots (the two never run at the same time).
 */
#include <math.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

enum {
    WS_HB_ACC0 = 23, /* double accumulation grids (WS_PT_RESAMPLED and the next free slot) */
    WS_HB_IN0 = 24,  /* .. +6 staged IC arrays (WS_PT_IN0 ..) */
    WS_HB_OUT0 = 32, /* .. +2 staged outputs */
    WS_HB_ACC1 = 35,
    WS_HB_ACC2 = 87,
    WS_HB_OUT3 = 88,
    WS_HB_TABLES = 38,
    WS_HB_PART = 39
};

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

static const float *hb_in(int slot, const float *p, size_t bytes, void *stream, int *status) {
    if (!p || *status || c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    *status = c21hip_h2d(d, p, bytes, stream);
    return (const float *)d;
}

int c21cm_grid_minmax(const float *values, size_t n, double out_minmax[2], void *stream) {
    int status = 0;
    if (!values || !n || !out_minmax) return C21CM_VALUE_ERROR;
    const float *d = hb_in(WS_HB_IN0, values, n * sizeof(float), stream, &status);
    if (status) return status;
    double *part = (double *)c21hip_ws(WS_HB_PART, (2 * 2048 + 2) * sizeof(double));
    if (!part) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_minmax_dense(d, n, part, part + 2 * 2048, stream));
    TRY(c21hip_d2h(out_minmax, part + 2 * 2048, 2 * sizeof(double), stream));
    TRY(c21hip_sync(stream));
done:
    return status;
}

/* Halo-catalogue branch: sum_halos_onto_grid (HaloBox.c:518-560) -> move_halo_galprops
 * (map_mass.c:346-476) into the zeroed accumulation grids, before the integrated deposit adds the
 * sources below the catalogue's mass limit.  acc = {n_ion, halo_sfr, halo_sfr_mini, halo_xray,
 * whalo_sfr}; entries may be NULL. */
enum { WS_HC_MASS = 231, WS_HC_COORD, WS_HC_RNG0, WS_HC_RNG1, WS_HC_RNG2, WS_HC_WSFR, WS_HC_BINS };

static int deposit_halos(const c21cm_halobox_spec *s, const float *const vel[3],
                         const float *const vel2[3], const int vel_dim[3], const float *mta,
                         const float *mtm, double *const acc[5], void *stream) {
    const HaloCatalog *h = s->halos;
    if (!h || !h->n_halos) return 0;
    const c21cm_halo_consts *c = s->halo_consts;
    if (!c || !h->halo_masses || !h->halo_coords || !h->star_rng || !h->sfr_rng ||
        (c->use_xray && !h->xray_rng) || (c->use_mini_halos && (!mta || !mtm))) {
        c21hip_set_error("halobox: the halo catalogue needs its constants, masses, coordinates and the "
                         "random deviates of the scaling relations");
        return C21CM_VALUE_ERROR;
    }
    int status = 0;
    const size_t nh = (size_t)h->n_halos, fb = nh * sizeof(float);
    const float *mass = hb_in(WS_HC_MASS, h->halo_masses, fb, stream, &status);
    const float *coord = hb_in(WS_HC_COORD, h->halo_coords, 3 * fb, stream, &status);
    const float *r0 = hb_in(WS_HC_RNG0, h->star_rng, fb, stream, &status);
    const float *r1 = hb_in(WS_HC_RNG1, h->sfr_rng, fb, stream, &status);
    const float *r2 = c->use_xray ? hb_in(WS_HC_RNG2, h->xray_rng, fb, stream, &status) : NULL;
    if (status) return status;
    const int out_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    int *bins = (int *)c21hip_ws(WS_HC_BINS, c21hip_halo_deposit_scratch_ints(h->n_halos, out_dim) * sizeof(int));
    if (!bins) return C21CM_MEMORY_ALLOC_ERROR;
    return c21hip_halo_deposit(c, h->n_halos, mass, coord, r0, r1, r2, vel, vel2, vel_dim, out_dim,
                               s->box_len, s->box_len_z, s->growth_factor, s->init_growth_factor,
                               s->lpt2, mta, mtm, acc[0], acc[1], acc[2], acc[3], acc[4], bins, stream);
}

/* USE_MINI_HALOS (HaloBox.c:245-283, map_mass.c:285-321) */
enum { WS_HBM_MTA = 217, WS_HBM_MTM, WS_HBM_TAB, WS_HBM_ACC3, WS_HBM_OUT4, WS_HBM_G12, WS_HBM_ZRE,
       WS_HBM_J21, WS_HBM_VCB, WS_HBM_OUTA, WS_HBM_OUTM, WS_HBM_SUMS };

static int halobox_grids_mini(const c21cm_halobox_spec *s, const InitialConditions *ics,
                              HaloBox *grids, void *stream) {
    int status = 0;
    if (s->perturb_on_high_res) {
        c21hip_set_error("halobox: USE_MINI_HALOS with PERTURB_ON_HIGH_RES is not built (upstream "
                         "indexes the low-resolution turnover grids with the high-resolution cell index)");
        return C21CM_VALUE_ERROR;
    }
    const int integral = !s->skip_integral;
    if (!s->log10_mturn_acg || !s->log10_mturn_mcg || !grids->halo_sfr_mini ||
        (integral && (!s->ln_sfrd_table || !s->ln_nion_table2d || !s->ln_nion_mini_table2d ||
                      !s->ln_sfrd_mini_table2d || !(s->tab_width > 0)))) {
        c21hip_set_error("halobox: USE_MINI_HALOS needs the turnover grids, the 1-D SFRD table, the "
                         "three 2-D tables and HaloBox.halo_sfr_mini");
        return C21CM_VALUE_ERROR;
    }
    const int dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const size_t n = (size_t)dim[0] * dim[1] * dim[2], fb = n * sizeof(float);
    const float *vel_h[3] = {ics->lowres_vx, ics->lowres_vy, ics->lowres_vz};
    const float *vel2_h[3] = {ics->lowres_vx_2LPT, ics->lowres_vy_2LPT, ics->lowres_vz_2LPT};
    if (!ics->lowres_density || !vel_h[0] || !vel_h[1] || !vel_h[2] ||
        (s->lpt2 && (!vel2_h[0] || !vel2_h[1] || !vel2_h[2]))) {
        c21hip_set_error("halobox: required InitialConditions arrays are missing");
        return C21CM_VALUE_ERROR;
    }
    const float *dens = hb_in(WS_HB_IN0, ics->lowres_density, fb, stream, &status);
    const float *vel[3], *vel2[3] = {NULL, NULL, NULL};
    for (int a = 0; a < 3; a++) {
        vel[a] = hb_in(WS_HB_IN0 + 1 + a, vel_h[a], fb, stream, &status);
        if (s->lpt2) vel2[a] = hb_in(WS_HB_IN0 + 4 + a, vel2_h[a], fb, stream, &status);
    }
    const float *mta = hb_in(WS_HBM_MTA, s->log10_mturn_acg, fb, stream, &status);
    const float *mtm = hb_in(WS_HBM_MTM, s->log10_mturn_mcg, fb, stream, &status);
    if (status) return status;
    const int xray = grids->halo_xray && (integral ? s->ln_xray_table2d != NULL
                                                    : (s->halo_consts && s->halo_consts->use_xray));
    const size_t t1 = C21CM_NDELTA_TABLE, t2 = (size_t)C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
    /* one float of slack per table row set: the lookups read [idx + 1] with weight 0 on the last knot */
    float *tab = (float *)c21hip_ws(WS_HBM_TAB, (t1 + 4 * t2 + 2 * C21CM_NMTURN_TABLE + 16) * sizeof(float));
    double *acc[4] = {(double *)c21hip_ws(WS_HB_ACC0, n * sizeof(double)),
                      (double *)c21hip_ws(WS_HB_ACC1, n * sizeof(double)),
                      (double *)c21hip_ws(WS_HBM_ACC3, n * sizeof(double)),
                      xray ? (double *)c21hip_ws(WS_HB_ACC2, n * sizeof(double)) : NULL};
    if (!tab || !acc[0] || !acc[1] || !acc[2] || (xray && !acc[3])) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_memset(tab, 0, (t1 + 4 * t2 + 2 * C21CM_NMTURN_TABLE + 16) * sizeof(float), stream));
    float *tab_sfrd = tab + 4 * t2 + C21CM_NMTURN_TABLE, *tab_na = tab, *tab_nm = tab + t2,
          *tab_sm = tab + 2 * t2, *tab_x = tab + 3 * t2;
    if (integral) {
        TRY(c21hip_h2d(tab_sfrd, s->ln_sfrd_table, t1 * sizeof(float), stream));
        TRY(c21hip_h2d(tab_na, s->ln_nion_table2d, t2 * sizeof(float), stream));
        TRY(c21hip_h2d(tab_nm, s->ln_nion_mini_table2d, t2 * sizeof(float), stream));
        TRY(c21hip_h2d(tab_sm, s->ln_sfrd_mini_table2d, t2 * sizeof(float), stream));
        if (xray) TRY(c21hip_h2d(tab_x, s->ln_xray_table2d, t2 * sizeof(float), stream));
    }
    for (int g = 0; g < 4; g++)
        if (acc[g]) TRY(c21hip_memset(acc[g], 0, n * sizeof(double), stream));
    double *acc_w = NULL; /* whalo_sfr of the halos: only kept without the integrated part */
    if (s->halos) {
        if (!integral && grids->whalo_sfr) {
            if (!(acc_w = (double *)c21hip_ws(WS_HC_WSFR, n * sizeof(double)))) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_memset(acc_w, 0, n * sizeof(double), stream));
        }
        double *const hacc[5] = {acc[0], acc[1], acc[2], acc[3], acc_w};
        TRY(deposit_halos(s, vel, vel2, dim, mta, mtm, hacc, stream));
    }
    const double ranges[8] = {s->tab_min, s->tab_width, s->mta_min, s->mta_width,
                              s->mtm_min, s->mtm_width, s->mt_fixed_min, s->mt_fixed_width};
    const double pref[5] = {s->prefactor_nion, s->prefactor_nion_mini, s->prefactor_sfr,
                            s->prefactor_sfr_mini, s->prefactor_xray};
    if (integral)
        TRY(c21hip_halobox_scatter_mini(dens, dim, vel, vel2, mta, mtm, acc[0], acc[1], acc[2], acc[3],
                                        s->box_len, s->box_len_z, s->growth_factor,
                                        s->init_growth_factor, s->lpt2, tab_sfrd, tab_na, tab_nm, tab_sm,
                                        xray ? tab_x : NULL, ranges, pref, stream));
    {
        float *targets[5] = {grids->n_ion, grids->whalo_sfr, grids->halo_sfr,
                             xray ? grids->halo_xray : NULL, grids->halo_sfr_mini};
        const int slots[5] = {WS_HB_OUT0, WS_HB_OUT0 + 1, WS_HB_OUT0 + 2, WS_HB_OUT3, WS_HBM_OUT4};
        float *dev[5] = {NULL, NULL, NULL, NULL, NULL};
        for (int t = 0; t < 5; t++) {
            if (!targets[t]) continue;
            dev[t] = c21hip_is_device_ptr(targets[t]) ? targets[t] : (float *)c21hip_ws(slots[t], fb);
            if (!dev[t]) return C21CM_MEMORY_ALLOC_ERROR;
        }
        TRY(c21hip_narrow(acc[0], dev[0], acc_w ? NULL : dev[1], s->prefactor_wsfr, n, stream));
        if (acc_w) TRY(c21hip_narrow(acc_w, dev[1], NULL, 0., n, stream));
        TRY(c21hip_narrow(acc[1], dev[2], NULL, 0., n, stream));
        TRY(c21hip_narrow(acc[2], dev[4], NULL, 0., n, stream));
        if (xray) TRY(c21hip_narrow(acc[3], dev[3], NULL, 0., n, stream));
        for (int t = 0; t < 5; t++)
            if (targets[t] && dev[t] != targets[t]) TRY(c21hip_d2h(targets[t], dev[t], fb, stream));
    }
    TRY(c21hip_sync(stream));
done:
    return status;
}

/* get_log10_turnovers (HaloBox.c:465-516); see c21hip_halobox_turnovers for n_threads */
int c21cm_halobox_turnovers(const c21cm_mturn_spec *m, double m_turn, int below_z_heat_max,
                            int n_threads, const float *prev_G12, const float *prev_z_reion,
                            const float *J_21_LW, const float *vcb, float *log10_mturn_acg,
                            float *log10_mturn_mcg, double averages[2], void *stream) {
    int status = 0;
    if (!m || !log10_mturn_acg || !log10_mturn_mcg ||
        (below_z_heat_max && (!prev_G12 || !prev_z_reion || !J_21_LW))) {
        c21hip_set_error("halobox turnovers: output grids and, below Z_HEAT_MAX, the previous "
                         "Gamma_12 / z_reion / J_21_LW grids are required");
        return C21CM_VALUE_ERROR;
    }
    const size_t n = (size_t)m->hii_dim * m->hii_dim * m->hii_dim_z, fb = n * sizeof(float);
    const float *g12 = below_z_heat_max ? hb_in(WS_HBM_G12, prev_G12, fb, stream, &status) : NULL;
    const float *zre = below_z_heat_max ? hb_in(WS_HBM_ZRE, prev_z_reion, fb, stream, &status) : NULL;
    const float *j21 = below_z_heat_max ? hb_in(WS_HBM_J21, J_21_LW, fb, stream, &status) : NULL;
    const float *v = vcb ? hb_in(WS_HBM_VCB, vcb, fb, stream, &status) : NULL;
    if (status) return status;
    float *oa = c21hip_is_device_ptr(log10_mturn_acg) ? log10_mturn_acg : (float *)c21hip_ws(WS_HBM_OUTA, fb);
    float *om = c21hip_is_device_ptr(log10_mturn_mcg) ? log10_mturn_mcg : (float *)c21hip_ws(WS_HBM_OUTM, fb);
    double *sums = (double *)c21hip_ws(WS_HBM_SUMS, 2 * sizeof(double));
    if (!oa || !om || !sums) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_memset(sums, 0, 2 * sizeof(double), stream));
    TRY(c21hip_halobox_turnovers(n, n_threads, below_z_heat_max, m->redshift, m->mturn_a_nofb, m_turn,
                                 m->vcb_const, m->A_LW, m->BETA_LW, m->A_VCB, m->BETA_VCB,
                                 m->sigma_vcb, g12, zre, j21, v, oa, om, sums, stream));
    double host_sums[2];
    TRY(c21hip_d2h(host_sums, sums, sizeof(host_sums), stream));
    if (oa != log10_mturn_acg) TRY(c21hip_d2h(log10_mturn_acg, oa, fb, stream));
    if (om != log10_mturn_mcg) TRY(c21hip_d2h(log10_mturn_mcg, om, fb, stream));
    TRY(c21hip_sync(stream));
    if (averages) averages[0] = host_sums[0] / n, averages[1] = host_sums[1] / n;
done:
    return status;
}

int c21cm_halobox_grids(const c21cm_halobox_spec *s, const InitialConditions *ics, HaloBox *grids,
                        void *stream) {
    int status = 0;
    if (s && s->use_mini_halos) {
        if (!ics || !grids || !grids->n_ion || !grids->halo_sfr) {
            c21hip_set_error("halobox: NULL spec / ics / n_ion / halo_sfr");
            return C21CM_VALUE_ERROR;
        }
        return halobox_grids_mini(s, ics, grids, stream);
    }
    if (!s || !ics || !grids || !grids->n_ion || !grids->halo_sfr) {
        c21hip_set_error("halobox: NULL spec / ics / n_ion / halo_sfr");
        return C21CM_VALUE_ERROR;
    }
    const int integral = !s->skip_integral;
    if (integral && (!s->ln_nion_table || !s->ln_sfrd_table || !(s->tab_width > 0))) {
        c21hip_set_error("halobox: the two ln-tables and a positive bin width are required");
        return C21CM_VALUE_ERROR;
    }
    const int hires = s->perturb_on_high_res;
    const int src_dim[3] = {hires ? s->dim : s->hii_dim, hires ? s->dim : s->hii_dim,
                            hires ? s->dim_z : s->hii_dim_z};
    const int out_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const size_t n_src = (size_t)src_dim[0] * src_dim[1] * src_dim[2];
    const size_t n_out = (size_t)out_dim[0] * out_dim[1] * out_dim[2];
    const float *dens_h = hires ? ics->hires_density : ics->lowres_density;
    const float *vel_h[3] = {hires ? ics->hires_vx : ics->lowres_vx,
                             hires ? ics->hires_vy : ics->lowres_vy,
                             hires ? ics->hires_vz : ics->lowres_vz};
    const float *vel2_h[3] = {hires ? ics->hires_vx_2LPT : ics->lowres_vx_2LPT,
                              hires ? ics->hires_vy_2LPT : ics->lowres_vy_2LPT,
                              hires ? ics->hires_vz_2LPT : ics->lowres_vz_2LPT};
    if (!dens_h || !vel_h[0] || !vel_h[1] || !vel_h[2] ||
        (s->lpt2 && (!vel2_h[0] || !vel2_h[1] || !vel2_h[2]))) {
        c21hip_set_error("halobox: required InitialConditions arrays are missing");
        return C21CM_VALUE_ERROR;
    }
    const float *dens = hb_in(WS_HB_IN0, dens_h, n_src * sizeof(float), stream, &status);
    const float *vel[3], *vel2[3] = {NULL, NULL, NULL};
    for (int a = 0; a < 3; a++) {
        vel[a] = hb_in(WS_HB_IN0 + 1 + a, vel_h[a], n_src * sizeof(float), stream, &status);
        if (s->lpt2) vel2[a] = hb_in(WS_HB_IN0 + 4 + a, vel2_h[a], n_src * sizeof(float), stream, &status);
    }
    if (status) return status;
    const int xray = grids->halo_xray && (integral ? s->ln_xray_table != NULL
                                                    : (s->halo_consts && s->halo_consts->use_xray));
    double *acc0 = (double *)c21hip_ws(WS_HB_ACC0, n_out * sizeof(double));
    double *acc1 = (double *)c21hip_ws(WS_HB_ACC1, n_out * sizeof(double));
    double *acc2 = xray ? (double *)c21hip_ws(WS_HB_ACC2, n_out * sizeof(double)) : NULL;
    float *tables = (float *)c21hip_ws(WS_HB_TABLES, 3 * C21CM_NDELTA_TABLE * sizeof(float));
    if (!acc0 || !acc1 || !tables || (xray && !acc2)) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_memset(acc0, 0, n_out * sizeof(double), stream));
    TRY(c21hip_memset(acc1, 0, n_out * sizeof(double), stream));
    if (xray) TRY(c21hip_memset(acc2, 0, n_out * sizeof(double), stream));
    double *acc_w = NULL; /* whalo_sfr of the halos: only kept without the integrated part */
    if (s->halos) {
        if (!integral && grids->whalo_sfr) {
            if (!(acc_w = (double *)c21hip_ws(WS_HC_WSFR, n_out * sizeof(double)))) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_memset(acc_w, 0, n_out * sizeof(double), stream));
        }
        double *const hacc[5] = {acc0, acc1, NULL, acc2, acc_w};
        TRY(deposit_halos(s, vel, vel2, src_dim, NULL, NULL, hacc, stream));
    }
    if (integral) {
        if (xray)
            TRY(c21hip_h2d(tables + 2 * C21CM_NDELTA_TABLE, s->ln_xray_table,
                           C21CM_NDELTA_TABLE * sizeof(float), stream));
        TRY(c21hip_h2d(tables, s->ln_nion_table, C21CM_NDELTA_TABLE * sizeof(float), stream));
        TRY(c21hip_h2d(tables + C21CM_NDELTA_TABLE, s->ln_sfrd_table,
                       C21CM_NDELTA_TABLE * sizeof(float), stream));
        TRY(c21hip_halobox_scatter(dens, src_dim, vel, vel2, src_dim, acc0, acc1, acc2, out_dim,
                                   s->box_len, s->box_len_z, s->growth_factor, s->init_growth_factor,
                                   s->lpt2, tables, s->tab_min, s->tab_width, s->prefactor_nion,
                                   s->prefactor_sfr, s->prefactor_xray, stream));
    }
    /* narrow into the caller's float grids (staged when they are host arrays) */
    {
        float *targets[4] = {grids->n_ion, grids->whalo_sfr, grids->halo_sfr,
                             xray ? grids->halo_xray : NULL};
        float *dev[4] = {NULL, NULL, NULL, NULL};
        for (int t = 0; t < 4; t++) {
            if (!targets[t]) continue;
            dev[t] = c21hip_is_device_ptr(targets[t])
                         ? targets[t]
                         : (float *)c21hip_ws(t < 3 ? WS_HB_OUT0 + t : WS_HB_OUT3, n_out * sizeof(float));
            if (!dev[t]) return C21CM_MEMORY_ALLOC_ERROR;
        }
        /* whalo_sfr = n_ion / t_h / t_star (map_mass.c:340-346) */
        TRY(c21hip_narrow(acc0, dev[0], acc_w ? NULL : dev[1], s->prefactor_wsfr, n_out, stream));
        if (acc_w) TRY(c21hip_narrow(acc_w, dev[1], NULL, 0., n_out, stream));
        TRY(c21hip_narrow(acc1, dev[2], NULL, 0., n_out, stream));
        if (xray) TRY(c21hip_narrow(acc2, dev[3], NULL, 0., n_out, stream));
        for (int t = 0; t < 4; t++)
            if (targets[t] && dev[t] != targets[t])
                TRY(c21hip_d2h(targets[t], dev[t], n_out * sizeof(float), stream));
    }
    TRY(c21hip_sync(stream));
done:
    return status;
}
