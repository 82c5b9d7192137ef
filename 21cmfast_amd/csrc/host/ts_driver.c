/*
 * ts_driver.c -- C host driver of the per-cell part of ComputeTsBox.
 *
 * reference: src/py21cmfast/src/SpinTemperatureBox.c
 *   :892-927    init_first_Ts                       -> c21cm_ts_first_grids
 *   :1387-1946  ts_main from the first cell loop on -> c21cm_ts_grids
 *
 * The host side only moves tables: the per-shell scalars and the frequency-integral tables
 * of the spec go to one small device buffer, the SFRD tables / Lyman-alpha heating tables
 * to their own, and two launches (three with SFRD tables) do the rest (ts_kernels.hip).
 * Host arrays are staged through workspace slots, device arrays are used in place; the big
 * [n_step][N] source grids are the only large transfer and are read exactly once.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

/* slots 150-169 (0-143 belong to the other drivers) */
enum {
    WS_TS_DENS = 150, WS_TS_PTS, WS_TS_PTK, WS_TS_PXE, WS_TS_GRID_A, WS_TS_GRID_B, WS_TS_TAB,
    WS_TS_SFRDTAB, WS_TS_LYA_C, WS_TS_LYA_I, WS_TS_OTS, WS_TS_OTK, WS_TS_OXE, WS_TS_PART,
    WS_TS_SMALL, WS_TS_MEANSFR, WS_TS_SFRDTAB2, WS_TS_SUMS
};
/* USE_MINI_HALOS: 2-D tables, filtered turnover grids, mini shell rows, J_21_LW staging */
enum { WS_TS_MINI_TAB = 208, WS_TS_MINI_MCRIT, WS_TS_MINI_SHELL, WS_TS_MINI_J21, WS_TS_MINI_MEAN,
       WS_TS_MCRIT_J21, WS_TS_MCRIT_VCB, WS_TS_MCRIT_OUT };

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

static const void *stage_in(int slot, const void *p, size_t bytes, void *stream, int *status) {
    if (*status || !p || c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        c21hip_set_error("spin temperature: out of device memory staging %zu bytes", bytes);
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    *status = c21hip_h2d(d, p, bytes, stream);
    return d;
}

/* a per-shell table set: copied with one float of slack, because upstream's lookup reads y[idx + 1]
 * with weight 0 when a cell sits exactly on the last knot (interpolation.c:123-131) */
static const float *stage_tables(int slot, const float *p, size_t bytes, void *stream, int *status) {
    if (*status || !p) return p;
    float *d = (float *)c21hip_ws(slot, bytes + 16);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    *status = c21hip_memset((char *)d + bytes, 0, 16, stream);
    if (!*status)
        *status = c21hip_is_device_ptr(p) ? c21hip_d2d(d, p, bytes, stream) : c21hip_h2d(d, p, bytes, stream);
    return d;
}

static float *stage_out(int slot, float *p, size_t bytes, int *status) {
    if (*status || c21hip_is_device_ptr(p)) return p;
    float *d = (float *)c21hip_ws(slot, bytes);
    if (!d) *status = C21CM_MEMORY_ALLOC_ERROR;
    return d;
}

static int check_boxes(const char *who, const TsBox *b) {
    if (!b || !b->spin_temperature || !b->kinetic_temp_neutral || !b->xray_ionised_fraction) {
        c21hip_set_error("spin temperature: %s needs spin_temperature, kinetic_temp_neutral and "
                         "xray_ionised_fraction", who);
        return C21CM_VALUE_ERROR;
    }
    return 0;
}

/* mode 0: the whole cell part.  Sharded runs (the shells dealt over ranks, ts_shard.c):
 * mode 1 = "shell sums only": box means + shell loop over THIS spec's shells, the six sums of every
 * cell copied to `sums_io` ([6][ntot] doubles, device), no temperature update;
 * mode 2 = "cells from sums": the temperature update of cells [cell0, cell0 + ncell) from the
 * combined sums in `sums_io` ([6][ncell] doubles, device); grids must be device arrays. */
static int ts_grids_impl(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                         const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                         c21cm_ts_report *report, void *stream, int mode, double *sums_io,
                         size_t cell0, size_t ncell);

int c21cm_ts_grids(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                   const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                   c21cm_ts_report *report, void *stream) {
    return ts_grids_impl(s, density, previous, source_box, filtered_density, out, report, stream, 0,
                         NULL, 0, 0);
}

int c21cm_ts_shell_sums(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                        const XraySourceBox *source_box, const float *filtered_density,
                        double *sums_dev, void *stream) {
    if (!sums_dev || !c21hip_is_device_ptr(sums_dev)) {
        c21hip_set_error("spin temperature: the shell sums go to a device buffer of 6 N doubles");
        return C21CM_VALUE_ERROR;
    }
    TsBox dummy = *previous; /* outputs are not written in this mode */
    return ts_grids_impl(s, density, previous, source_box, filtered_density, &dummy, NULL, stream, 1,
                         sums_dev, 0, 0);
}

int c21cm_ts_cells_from_sums(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                             const double *sums_dev, size_t cell0, size_t ncell, TsBox *out,
                             void *stream) {
    if (!sums_dev || !c21hip_is_device_ptr(sums_dev) || ncell < 1) {
        c21hip_set_error("spin temperature: cells_from_sums needs device sums and a cell range");
        return C21CM_VALUE_ERROR;
    }
    return ts_grids_impl(s, density, previous, NULL, NULL, out, NULL, stream, 2, (double *)sums_dev,
                         cell0, ncell);
}

static int ts_grids_impl(const c21cm_ts_spec *s, const float *density, const TsBox *previous,
                         const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                         c21cm_ts_report *report, void *stream, int mode, double *sums_io,
                         size_t cell0, size_t ncell) {
    int status = 0;
    if (!s || !density) {
        c21hip_set_error("spin temperature: spec and density are required");
        return C21CM_VALUE_ERROR;
    }
    if (s->hii_dim < 1 || s->hii_dim_z < 1 || s->n_step < 1 || s->n_step > C21CM_MAX_TS_RADII) {
        c21hip_set_error("spin temperature: bad geometry %d x %d x %d, %d shells (<= %d)",
                         s->hii_dim, s->hii_dim, s->hii_dim_z, s->n_step, C21CM_MAX_TS_RADII);
        return C21CM_VALUE_ERROR;
    }
    if ((status = check_boxes("the previous box", previous))) return status;
    if ((status = check_boxes("the output box", out))) return status;
    const int lagrangian = s->source_mode == C21CM_TS_SRC_GRIDS;
    const int fcoll_mode = s->source_mode == C21CM_TS_SRC_FCOLL_TABLES;
    if (!lagrangian && !fcoll_mode && s->source_mode != C21CM_TS_SRC_SFRD_TABLE) {
        c21hip_set_error("spin temperature: unknown source_mode %d", s->source_mode);
        return C21CM_VALUE_ERROR;
    }
    if (!s->freq_int_heat || !s->freq_int_ion || !s->freq_int_lya) {
        c21hip_set_error("spin temperature: the three frequency-integral tables are required");
        return C21CM_VALUE_ERROR;
    }
    if (s->use_lya_heating && (!s->lya_dEC || !s->lya_dEI)) {
        c21hip_set_error("spin temperature: USE_LYA_HEATING needs the two heating-efficiency tables");
        return C21CM_VALUE_ERROR;
    }
    if (mode == 2) {
        if (!c21hip_is_device_ptr(density) || !c21hip_is_device_ptr(previous->spin_temperature) ||
            !c21hip_is_device_ptr(out->spin_temperature) || s->use_mini_halos) {
            c21hip_set_error("spin temperature: cells_from_sums takes device arrays (no mini-halos)");
            return C21CM_VALUE_ERROR;
        }
    } else if (mode == 1 && (s->use_mini_halos || s->no_light)) {
        c21hip_set_error("spin temperature: shell_sums without mini-halos, after the first sources");
        return C21CM_VALUE_ERROR;
    }
    if (mode != 2 && lagrangian && !s->no_light &&
        (!source_box || !source_box->filtered_sfr || !source_box->filtered_xray)) {
        c21hip_set_error("spin temperature: Lagrangian sources need XraySourceBox.filtered_sfr and "
                         "filtered_xray");
        return C21CM_VALUE_ERROR;
    }
    if (mode != 2 && !lagrangian && !s->no_light &&
        (!filtered_density || (fcoll_mode ? (!s->fcoll_tables || !s->dfcoll_tables) : !s->ln_sfrd_tables))) {
        c21hip_set_error("spin temperature: Eulerian sources need the filtered densities and the "
                         "SFRD tables");
        return C21CM_VALUE_ERROR;
    }
    const int mini = s->use_mini_halos;
    if (mini && (fcoll_mode || !out->J_21_LW ||
                 (!lagrangian && !s->no_light && (!s->ln_sfrd_tables_mini || !s->filtered_log10_mcrit)) ||
                 (lagrangian && !s->no_light && !source_box->filtered_sfr_mini))) {
        c21hip_set_error("spin temperature: USE_MINI_HALOS needs TsBox.J_21_LW and either the "
                         "E-INTEGRAL inputs (2-D SFRD tables, filtered log10 M_crit grids) or, with "
                         "source grids, XraySourceBox.filtered_sfr_mini");
        return C21CM_VALUE_ERROR;
    }
    if (c21hip_device_count() < 1) {
        c21hip_set_error("spin temperature: no MI355X device visible");
        return C21CM_IO_ERROR;
    }

    const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
    const size_t bytes = ntot * sizeof(float);
    const int n = s->n_step;

    /* ---- small tables -> one device buffer */
    const size_t n_tab = c21hip_ts_table_doubles(n);
    double *host_tab = (double *)malloc(n_tab * sizeof(double));
    if (!host_tab) return C21CM_MEMORY_ALLOC_ERROR;
    {
        const double *rows[C21HIP_TS_SHELL_ROWS] = {
            s->z_edge_factor, s->xray_R_factor, s->starlya_prefactor, s->lya_cont_prefactor,
            s->lya_inj_prefactor, s->zpp_growth, s->tab_min, s->tab_width, NULL, NULL};
        for (int r = 0; r < C21HIP_TS_SHELL_ROWS; r++)
            for (int i = 0; i < n; i++) host_tab[r * n + i] = rows[r] ? rows[r][i] : 1.;
        for (int i = 0; i < n; i++) /* row 9: reciprocal table spacing (0 for unused shells) */
            host_tab[9 * n + i] = s->tab_width[i] != 0 ? 1. / s->tab_width[i] : 0.;
        double *f = host_tab + (size_t)C21HIP_TS_SHELL_ROWS * n;
        const size_t fn = (size_t)C21CM_X_INT_NXHII * n;
        memcpy(f, s->freq_int_heat, fn * sizeof(double));
        memcpy(f + fn, s->freq_int_ion, fn * sizeof(double));
        memcpy(f + 2 * fn, s->freq_int_lya, fn * sizeof(double));
    }
    double *dev_tab = (double *)c21hip_ws(WS_TS_TAB, n_tab * sizeof(double));
    double *small = (double *)c21hip_ws(WS_TS_SMALL, (8 + (size_t)n) * sizeof(double) + 64);
    double *partials = (double *)c21hip_ws(WS_TS_PART, (size_t)(512 * n + 6 * 2048) * sizeof(double));
    if (!dev_tab || !small || !partials) {
        status = C21CM_MEMORY_ALLOC_ERROR;
        goto done;
    }
    double *sums_dev = small, *ave_dev = small + 8;
    int *flag_dev = (int *)(small + 8 + n);
    TRY(c21hip_h2d(dev_tab, host_tab, n_tab * sizeof(double), stream));
    TRY(c21hip_memset(flag_dev, 0, sizeof(int), stream));

    const float *tables_dev = NULL, *mean_tables_dev = NULL; /* the cell sweep's | the box mean's */
    const double *lya_c = NULL, *lya_i = NULL;
    if (mode != 2 && !lagrangian && !s->no_light) {
        const size_t tb = (size_t)n * C21CM_NDELTA_TABLE * sizeof(float);
        if (fcoll_mode) {
            tables_dev = stage_tables(WS_TS_SFRDTAB, s->dfcoll_tables, tb, stream, &status);
            mean_tables_dev = stage_tables(WS_TS_SFRDTAB2, s->fcoll_tables, tb, stream, &status);
        } else {
            tables_dev = mean_tables_dev = stage_tables(WS_TS_SFRDTAB, s->ln_sfrd_tables, tb, stream, &status);
        }
    }
    if (s->use_lya_heating) {
        const size_t lb = (size_t)C21CM_LYA_NT * C21CM_LYA_NT * C21CM_LYA_NGP * sizeof(double);
        lya_c = (const double *)stage_in(WS_TS_LYA_C, s->lya_dEC, lb, stream, &status);
        lya_i = (const double *)stage_in(WS_TS_LYA_I, s->lya_dEI, lb, stream, &status);
    }
    if (status) goto done;

    /* ---- grids */
    const float *d_dens = (const float *)stage_in(WS_TS_DENS, density, bytes, stream, &status);
    const float *d_pts = (const float *)stage_in(WS_TS_PTS, previous->spin_temperature, bytes, stream, &status);
    const float *d_ptk = (const float *)stage_in(WS_TS_PTK, previous->kinetic_temp_neutral, bytes, stream, &status);
    const float *d_pxe = (const float *)stage_in(WS_TS_PXE, previous->xray_ionised_fraction, bytes, stream, &status);
    if (mode == 2) { /* the slab [cell0, cell0 + ncell) of device arrays */
        d_dens += cell0, d_pts += cell0, d_ptk += cell0, d_pxe += cell0;
    }
    const float *grid_a = NULL, *grid_b = NULL;
    if (mode != 2 && !s->no_light) {
        if (lagrangian) {
            grid_a = (const float *)stage_in(WS_TS_GRID_A, source_box->filtered_sfr, bytes * n, stream, &status);
            grid_b = (const float *)stage_in(WS_TS_GRID_B, source_box->filtered_xray, bytes * n, stream, &status);
        } else {
            grid_a = (const float *)stage_in(WS_TS_GRID_A, filtered_density, bytes * n, stream, &status);
        }
    }
    float *o_ts = stage_out(WS_TS_OTS, out->spin_temperature, bytes, &status);
    float *o_tk = stage_out(WS_TS_OTK, out->kinetic_temp_neutral, bytes, &status);
    float *o_xe = stage_out(WS_TS_OXE, out->xray_ionised_fraction, bytes, &status);
    if (status) goto done;
    if (mode == 2) o_ts += cell0, o_tk += cell0, o_xe += cell0;

    c21hip_ts_args a;
    memset(&a, 0, sizeof(a));
    a.n_step = n;
    a.lagrangian = lagrangian;
    a.use_xray_heating = s->use_xray_heating;
    a.use_cmb_heating = s->use_cmb_heating;
    a.use_lya_heating = s->use_lya_heating;
    a.no_light = s->no_light;
    a.table_exp = !fcoll_mode;
#define CP(f) a.f = s->f
    CP(redshift); CP(dzp); CP(growth_ratio); CP(No); CP(N_b0); CP(h_frac); CP(he_frac); CP(k_B);
    CP(h_p); CP(m_p); CP(c_cms); CP(A10); CP(T_21); CP(lambda_21); CP(nu_Ly_alpha);
    CP(clumping_factor); CP(xray_prefactor); CP(Trad); CP(Ts_prefactor); CP(xa_tilde_prefactor);
    CP(xc_inverse); CP(dcomp_dzp_prefactor); CP(Nb_zp); CP(N_zp); CP(lya_star_prefactor);
    CP(volunit_inv); CP(hubble_zp); CP(growth_zp); CP(dgrowth_dzp); CP(dt_dzp); CP(sfr_scale);
    CP(xray_scale);
#undef CP

    if (mode != 2 && !lagrangian && !s->no_light) { /* avg_fix_term of every shell (:1621-1626) */
        double *mean_dev = (double *)c21hip_ws(WS_TS_MEANSFR, (size_t)n * sizeof(double));
        if (!mean_dev) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        TRY(c21hip_h2d(mean_dev, s->mean_sfr_zpp, (size_t)n * sizeof(double), stream));
        TRY(c21hip_ts_sfrd_means(grid_a, mean_tables_dev, !fcoll_mode, dev_tab, mean_dev, n, ntot,
                                 partials, ave_dev, stream));
    }
    double *sums_ws = NULL; /* the six sums of every cell between the two sweeps */
    if (mode == 2) {
        sums_ws = sums_io; /* combined over the ranks' shells by the caller: [6][ncell] */
        a.sums_ready = 1;
    } else if (mode == 1) {
        sums_ws = sums_io; /* straight into the caller's buffer */
    } else if (!s->no_light) {
        sums_ws = (double *)c21hip_ws(WS_TS_SUMS, 6 * ntot * sizeof(double));
        if (!sums_ws) {
            c21hip_set_error("spin temperature: out of device memory for the shell sums (%zu bytes)",
                             6 * ntot * sizeof(double));
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
    }
    float *o_j21 = NULL;
    double *ave_mini_dev = NULL;
    if (mini) { /* the shell loop with both populations, J_21_LW (:1011-1075,1642-1733,1843) */
        o_j21 = stage_out(WS_TS_MINI_J21, out->J_21_LW, bytes, &status);
        if (status) goto done;
        if (s->no_light) {
            TRY(c21hip_memset(o_j21, 0, bytes, stream));
        } else if (lagrangian) { /* source grids: :1657-1701 */
            double *mini_shell = (double *)c21hip_ws(
                WS_TS_MINI_SHELL, (size_t)(C21HIP_TS_MINI_ROWS + 2) * n * sizeof(double));
            if (!mini_shell) {
                status = C21CM_MEMORY_ALLOC_ERROR;
                goto done;
            }
            double rows[C21HIP_TS_MINI_ROWS * C21CM_MAX_TS_RADII];
            const double *src[C21HIP_TS_MINI_ROWS] = {
                s->starlya_prefactor_mini, s->lya_cont_prefactor_mini, s->lya_inj_prefactor_mini,
                s->lw_prefactor, s->lw_prefactor_mini, NULL};
            for (int r = 0; r < C21HIP_TS_MINI_ROWS; r++)
                for (int i = 0; i < n; i++) rows[r * n + i] = src[r] ? src[r][i] : 1.;
            TRY(c21hip_h2d(mini_shell, rows, (size_t)C21HIP_TS_MINI_ROWS * n * sizeof(double), stream));
            TRY(c21hip_sync(stream)); /* `rows` is a stack buffer */
            const float *g_mini = (const float *)stage_in(WS_TS_MINI_MCRIT, source_box->filtered_sfr_mini,
                                                          bytes * n, stream, &status);
            const float *g_lw = NULL, *g_mini_lw = NULL;
            if (source_box->filtered_sfr_lw && source_box->filtered_sfr_mini_lw) {
                g_lw = (const float *)stage_in(WS_TS_MINI_TAB, source_box->filtered_sfr_lw, bytes * n,
                                               stream, &status);
                g_mini_lw = (const float *)stage_in(WS_TS_MINI_MEAN, source_box->filtered_sfr_mini_lw,
                                                    bytes * n, stream, &status);
            }
            if (status) goto done;
            TRY(c21hip_ts_accumulate_grids_mini(&a, d_pxe, grid_a, grid_b, g_mini, g_lw, g_mini_lw,
                                                dev_tab, mini_shell, sums_ws, o_j21, ntot, stream));
            a.sums_ready = 1;
        } else {
            const size_t t2b = (size_t)n * C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE * sizeof(float);
            /* (slack: one more row of the last table for the weight-0 reads on the last knots) */
            float *tab2 = (float *)c21hip_ws(WS_TS_MINI_TAB, t2b + 2 * C21CM_NMTURN_TABLE * sizeof(float));
            double *mini_shell = (double *)c21hip_ws(
                WS_TS_MINI_SHELL, (size_t)(C21HIP_TS_MINI_ROWS + 2) * n * sizeof(double));
            if (!tab2 || !mini_shell) {
                status = C21CM_MEMORY_ALLOC_ERROR;
                goto done;
            }
            TRY(c21hip_memset((char *)tab2 + t2b, 0, 2 * C21CM_NMTURN_TABLE * sizeof(float), stream));
            if (c21hip_is_device_ptr(s->ln_sfrd_tables_mini))
                TRY(c21hip_d2d(tab2, s->ln_sfrd_tables_mini, t2b, stream));
            else
                TRY(c21hip_h2d(tab2, s->ln_sfrd_tables_mini, t2b, stream));
            const float *d_mcrit = (const float *)stage_in(WS_TS_MINI_MCRIT, s->filtered_log10_mcrit,
                                                           bytes * n, stream, &status);
            if (status) goto done;
            double rows[(C21HIP_TS_MINI_ROWS + 2) * C21CM_MAX_TS_RADII];
            const double *src[C21HIP_TS_MINI_ROWS] = {
                s->starlya_prefactor_mini, s->lya_cont_prefactor_mini, s->lya_inj_prefactor_mini,
                s->lw_prefactor, s->lw_prefactor_mini, NULL};
            for (int r = 0; r < C21HIP_TS_MINI_ROWS; r++)
                for (int i = 0; i < n; i++) rows[r * n + i] = src[r] ? src[r][i] : 1.;
            for (int i = 0; i < n; i++) rows[C21HIP_TS_MINI_ROWS * n + i] = s->mean_sfr_zpp_mini[i];
            TRY(c21hip_h2d(mini_shell, rows, (size_t)(C21HIP_TS_MINI_ROWS + 1) * n * sizeof(double),
                           stream));
            TRY(c21hip_sync(stream)); /* `rows` is a stack buffer */
            ave_mini_dev = mini_shell + (size_t)(C21HIP_TS_MINI_ROWS + 1) * n;
            TRY(c21hip_ts_sfrd_means_mini(grid_a, d_mcrit, tab2, dev_tab, mini_shell,
                                          mini_shell + (size_t)C21HIP_TS_MINI_ROWS * n, n, ntot,
                                          s->mturn_tab_min, s->mturn_tab_width, partials,
                                          ave_mini_dev, stream));
            TRY(c21hip_ts_accumulate_mini(&a, s->sfr_scale_mini, s->xray_scale_mini,
                                          s->mturn_tab_min, s->mturn_tab_width, d_pxe, grid_a,
                                          d_mcrit, tables_dev, tab2, dev_tab, mini_shell, sums_ws,
                                          o_j21, ntot, stream));
            a.sums_ready = 1;
        }
    }
    if (mode == 1) { /* the shell loop only: its sums are the result */
        TRY(c21hip_ts_shell_loop(&a, d_pxe, grid_a, grid_b, tables_dev, dev_tab, sums_ws, ntot, stream));
        TRY(c21hip_sync(stream));
        goto done;
    }
    TRY(c21hip_ts_cells(&a, d_dens, d_pts, d_ptk, d_pxe, grid_a, grid_b, tables_dev, dev_tab, lya_c,
                        lya_i, o_ts, o_tk, o_xe, mode == 2 ? ncell : ntot, sums_ws,
                        partials + (size_t)512 * n, sums_dev, flag_dev, stream));

    if (mode != 2) { /* (a slab of device arrays was written in place) */
        if (o_ts != out->spin_temperature) TRY(c21hip_d2h(out->spin_temperature, o_ts, bytes, stream));
        if (o_tk != out->kinetic_temp_neutral) TRY(c21hip_d2h(out->kinetic_temp_neutral, o_tk, bytes, stream));
        if (o_xe != out->xray_ionised_fraction) TRY(c21hip_d2h(out->xray_ionised_fraction, o_xe, bytes, stream));
    }
    if (mini && o_j21 != out->J_21_LW) TRY(c21hip_d2h(out->J_21_LW, o_j21, bytes, stream));
    double back_mini[C21CM_MAX_TS_RADII];
    if (ave_mini_dev) TRY(c21hip_d2h(back_mini, ave_mini_dev, (size_t)n * sizeof(double), stream));
    {
        double back[8 + C21CM_MAX_TS_RADII + 8];
        TRY(c21hip_d2h(back, small, (8 + (size_t)n) * sizeof(double) + sizeof(int), stream));
        TRY(c21hip_sync(stream));
        int flag;
        memcpy(&flag, back + 8 + n, sizeof(int));
        if (report) {
            memset(report, 0, sizeof(*report));
            report->Ts_ave = back[0] / (double)ntot;
            report->Tk_ave = back[1] / (double)ntot;
            report->x_e_ave = back[2] / (double)ntot;
            report->J_alpha_ave = back[3] / (double)ntot;
            report->xheat_ave = back[4] / (double)ntot;
            report->xion_ave = back[5] / (double)ntot;
            if (!lagrangian && !s->no_light)
                for (int i = 0; i < n; i++) report->ave_sfrd[i] = back[8 + i];
            if (ave_mini_dev)
                for (int i = 0; i < n; i++) report->ave_sfrd_mini[i] = back_mini[i];
        }
        if (flag) {
            c21hip_set_error("Estimated spin temperature is either infinite or NaN");
            status = C21CM_INFINITY_OR_NAN_ERROR;
        }
    }
done:
    free(host_tab);
    return status;
}

int c21cm_ts_first_grids(const c21cm_ts_first_spec *s, const float *density, TsBox *out,
                         void *stream) {
    int status = 0;
    if (!s || !density || s->hii_dim < 1 || s->hii_dim_z < 1) {
        c21hip_set_error("init_first_Ts: spec, density and a grid geometry are required");
        return C21CM_VALUE_ERROR;
    }
    if ((status = check_boxes("the output box", out))) return status;
    if (c21hip_device_count() < 1) {
        c21hip_set_error("init_first_Ts: no MI355X device visible");
        return C21CM_IO_ERROR;
    }
    const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z, bytes = ntot * sizeof(float);
    const float *d_dens = (const float *)stage_in(WS_TS_DENS, density, bytes, stream, &status);
    float *o_ts = stage_out(WS_TS_OTS, out->spin_temperature, bytes, &status);
    float *o_tk = stage_out(WS_TS_OTK, out->kinetic_temp_neutral, bytes, &status);
    float *o_xe = stage_out(WS_TS_OXE, out->xray_ionised_fraction, bytes, &status);
    if (status) goto done;
    TRY(c21hip_ts_first(s, d_dens, o_ts, o_tk, o_xe, ntot, stream));
    if (o_ts != out->spin_temperature) TRY(c21hip_d2h(out->spin_temperature, o_ts, bytes, stream));
    if (o_tk != out->kinetic_temp_neutral) TRY(c21hip_d2h(out->kinetic_temp_neutral, o_tk, bytes, stream));
    if (o_xe != out->xray_ionised_fraction) TRY(c21hip_d2h(out->xray_ionised_fraction, o_xe, bytes, stream));
    TRY(c21hip_sync(stream));
done:
    return status;
}

/* prepare_filter_boxes with USE_MINI_HALOS (SpinTemperatureBox.c:535-565) */
int c21cm_ts_mcrit_grid(const c21cm_mturn_spec *spec, double m_turn, const float *J_21_LW,
                        const float *vcb, float *log10_mcrit, void *stream) {
    int status = 0;
    if (!spec || !J_21_LW || !log10_mcrit || spec->hii_dim < 1 || spec->hii_dim_z < 1) {
        c21hip_set_error("ts_mcrit_grid: spec, J_21_LW and the output grid are required");
        return C21CM_VALUE_ERROR;
    }
    const size_t ntot = (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z;
    const size_t bytes = ntot * sizeof(float);
    const float *j21 = (const float *)stage_in(WS_TS_MCRIT_J21, J_21_LW, bytes, stream, &status);
    const float *v = vcb ? (const float *)stage_in(WS_TS_MCRIT_VCB, vcb, bytes, stream, &status) : NULL;
    float *o = stage_out(WS_TS_MCRIT_OUT, log10_mcrit, bytes, &status);
    if (status) goto done;
    TRY(c21hip_ts_mcrit_grid(j21, v, spec->vcb_const, spec->redshift, spec->A_LW, spec->BETA_LW,
                             spec->A_VCB, spec->BETA_VCB, spec->sigma_vcb, m_turn, o, ntot, stream));
    if (o != log10_mcrit) {
        TRY(c21hip_d2h(log10_mcrit, o, bytes, stream));
        TRY(c21hip_sync(stream));
    }
done:
    return status;
}
