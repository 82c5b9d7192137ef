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

int c21cm_halobox_grids(const c21cm_halobox_spec *s, const InitialConditions *ics, HaloBox *grids,
                        void *stream) {
    int status = 0;
    if (!s || !ics || !grids || !grids->n_ion || !grids->halo_sfr) {
        c21hip_set_error("halobox: NULL spec / ics / n_ion / halo_sfr");
        return C21CM_VALUE_ERROR;
    }
    if (!s->ln_nion_table || !s->ln_sfrd_table || !(s->tab_width > 0)) {
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
    const int xray = s->ln_xray_table && grids->halo_xray;
    double *acc0 = (double *)c21hip_ws(WS_HB_ACC0, n_out * sizeof(double));
    double *acc1 = (double *)c21hip_ws(WS_HB_ACC1, n_out * sizeof(double));
    double *acc2 = xray ? (double *)c21hip_ws(WS_HB_ACC2, n_out * sizeof(double)) : NULL;
    float *tables = (float *)c21hip_ws(WS_HB_TABLES, 3 * C21CM_NDELTA_TABLE * sizeof(float));
    if (!acc0 || !acc1 || !tables || (xray && !acc2)) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_memset(acc0, 0, n_out * sizeof(double), stream));
    TRY(c21hip_memset(acc1, 0, n_out * sizeof(double), stream));
    if (xray) {
        TRY(c21hip_memset(acc2, 0, n_out * sizeof(double), stream));
        TRY(c21hip_h2d(tables + 2 * C21CM_NDELTA_TABLE, s->ln_xray_table,
                       C21CM_NDELTA_TABLE * sizeof(float), stream));
    }
    TRY(c21hip_h2d(tables, s->ln_nion_table, C21CM_NDELTA_TABLE * sizeof(float), stream));
    TRY(c21hip_h2d(tables + C21CM_NDELTA_TABLE, s->ln_sfrd_table,
                   C21CM_NDELTA_TABLE * sizeof(float), stream));
    TRY(c21hip_halobox_scatter(dens, src_dim, vel, vel2, src_dim, acc0, acc1, acc2, out_dim,
                               s->box_len, s->box_len_z, s->growth_factor, s->init_growth_factor,
                               s->lpt2, tables, s->tab_min, s->tab_width, s->prefactor_nion,
                               s->prefactor_sfr, s->prefactor_xray, stream));
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
        TRY(c21hip_narrow(acc0, dev[0], dev[1], s->prefactor_wsfr, n_out, stream));
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
