/*
 * tsfilter_driver.c -- C host driver of the filtering stage of the spin-temperature
 * calculation: fill_Rbox_table (SpinTemperatureBox.c:560-636, with the
 * transform of prepare_filter_boxes :502-520) and one_annular_filter (:642-742).
 *
 * Both are "one forward transform, then per window: multiply, transform back, floor, scale,
 * store, statistics".  On the native sizes the spectrum lives in the split layout, the window
 * is applied inside pass X and the floor/scale/statistics are the store epilogue of pass Z;
 * other sizes go through rocFFT and two streaming sweeps.
 * Host arrays are staged through workspace slots, device arrays are used in place.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

/* slots 0-63 belong to the other drivers */
enum { WS_TF_IN = 64, WS_TF_UNF, WS_TF_WORK, WS_TF_OUT, WS_TF_PART, WS_TF_UNF2, WS_TF_WORK2, WS_TF_IN2, WS_TF_OUT2, WS_TF_RPART };

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

typedef struct {
    int nx, ny, nz, native;
    size_t ntot, npad;
    double box_len, box_len_z;
    float *unf, *work;
    double *partials, *stats; /* stats: 3 doubles per window, device */
    void *stream;
} tf_ctx;

static int tf_setup(tf_ctx *c, int hii_dim, int hii_dim_z, double box_len, double box_len_z,
                    int n_stats, void *stream) {
    memset(c, 0, sizeof(*c));
    if (hii_dim < 2 || hii_dim_z < 2 || !(box_len > 0) || !(box_len_z > 0)) {
        c21hip_set_error("ts filter: bad grid geometry %d x %d x %d", hii_dim, hii_dim, hii_dim_z);
        return C21CM_VALUE_ERROR;
    }
    c->nx = c->ny = hii_dim;
    c->nz = hii_dim_z;
    c->native = c21hip_fft_is_native(c->nx, c->ny, c->nz);
    c->ntot = (size_t)c->nx * c->ny * c->nz;
    c->npad = (size_t)c->nx * c->ny * 2 * (size_t)(c->nz / 2 + 1);
    c->box_len = box_len;
    c->box_len_z = box_len_z;
    c->stream = stream;
    const size_t sbytes = (c->native ? c21hip_split_floats(c->nx, c->ny, c->nz) : c->npad) * sizeof(float);
    c->unf = (float *)c21hip_ws(WS_TF_UNF, sbytes);
    c->work = (float *)c21hip_ws(WS_TF_WORK, sbytes);
    const size_t nlines = (size_t)c->nx * c->ny;
    size_t npart = 3 * (nlines / 16 + 1) + 2 * (nlines / 16384 + 2);
    if (npart < 3 * (size_t)C21HIP_PARTIALS) npart = 3 * (size_t)C21HIP_PARTIALS;
    c->partials = (double *)c21hip_ws(WS_TF_PART, (npart + 3 * (size_t)n_stats) * sizeof(double));
    if (!c->unf || !c->work || !c->partials) return C21CM_MEMORY_ALLOC_ERROR;
    c->stats = c->partials + npart;
    return 0;
}

/* dense real input -> unfiltered spectrum / N   (prepare_filter_boxes, :511-530) */
static int tf_forward(tf_ctx *c, const float *d_in) {
    int st;
    if (c->native)
        /* 1 / N is exact for the power-of-two sizes of the native transform */
        return c21hip_split_r2c(d_in, c->nz, c->unf, c->nx, c->ny, c->nz, 1.0, 1., -1.,
                                (float)(1.0 / (double)c->ntot), c->stream);
    if ((st = c21hip_pack_clip(d_in, c->unf, c->nx, c->ny, c->nz, 1.0, -1e300, 1e300, c->stream)))
        return st;
    if ((st = c21hip_fft_r2c(c->unf, c->nx, c->ny, c->nz, c->stream))) return st;
    return c21hip_divide_inplace(c->unf, c->npad, (float)c->ntot, c->stream);
}

/* one window: spectrum x W -> real space -> max(v, min_value) * const_factor -> d_out,
 * statistics {min, max, sum} to stats3 (device) */
static int tf_window(tf_ctx *c, int filter_type, float R, float R_param, int apply,
                     double min_value, double const_factor, float *d_out, double *stats3) {
    int st;
    if (c->native) {
        if ((st = c21hip_split_filter_xy(c->unf, c->work, c->nx, c->ny, c->nz, c->box_len,
                                         c->box_len_z, filter_type, R, R_param, apply, c->stream)))
            return st;
        return c21hip_split_z_c2r_stats(c->work, d_out, c->nz, c->nx, c->ny, c->nz, min_value,
                                        const_factor, c->partials, stats3, c->stream);
    }
    if ((st = c21hip_copy_filter(c->unf, c->work, c->nx, c->ny, c->nz, c->box_len, c->box_len_z,
                                 filter_type, R, R_param, apply, c->stream)))
        return st;
    if ((st = c21hip_fft_c2r(c->work, c->nx, c->ny, c->nz, c->stream))) return st;
    return c21hip_floor_scale_stats(c->work, 2 * (long)(c->nz / 2 + 1), d_out, c->nx, c->ny, c->nz,
                                    min_value, const_factor, c->partials, stats3, c->stream);
}

static const float *tf_stage_in(const float *p, size_t bytes, void *stream, int *status) {
    if (*status || c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(WS_TF_IN, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    *status = c21hip_h2d(d, p, bytes, stream);
    return (const float *)d;
}

/* Native sizes: the W(kR) table of radius r+1 is built on the library's side stream while
 * radius r runs its passes (two table buffers, events both ways, as in the excursion-set loop),
 * and the min / max / sum partials of all radii are reduced by one launch at the end. */
static struct {
    int init, ok;
    void *aux, *ev_table[2], *ev_used[2], *ev_sync;
} tf_tab;

static void tf_tab_init(void) {
    if (tf_tab.init) return;
    tf_tab.init = 1;
    const char *e = getenv("C21CM_ASYNC_TABLES");
    if (e && e[0] == '0') return;
    tf_tab.aux = c21hip_aux_stream();
    tf_tab.ev_sync = c21hip_event_create();
    for (int b = 0; b < 2; b++) {
        tf_tab.ev_table[b] = c21hip_event_create();
        tf_tab.ev_used[b] = c21hip_event_create();
    }
    tf_tab.ok = tf_tab.aux && tf_tab.ev_sync && tf_tab.ev_table[0] && tf_tab.ev_table[1] &&
                tf_tab.ev_used[0] && tf_tab.ev_used[1];
}

static int fill_Rbox_native(tf_ctx *c, const c21cm_rbox_spec *s, float *result, int host_out,
                            float *stage_out) {
    int status = 0;
    const size_t bytes = c->ntot * sizeof(float);
    const long nb = (long)c->nx * c->ny / 16, stride = 3 * nb;
    double *partials =
        (double *)c21hip_ws(WS_TF_RPART, (size_t)s->n_R * (size_t)stride * sizeof(double));
    if (!partials) return C21CM_MEMORY_ALLOC_ERROR;
    tf_tab_init();
    int filtered[C21CM_MAX_TS_RADII], n_f = 0; /* radii that need a table, in order */
    for (int r = 0; r < s->n_R; r++)
        if (s->R[r] > s->cell_radius) filtered[n_f++] = r;
    int next = 0; /* index into filtered[] of the next table to build */
    /* Top-hat / sharp-k windows on 256- and 512-point lines: evaluated inside pass X from node
     * tables of W(kR) (fft_native.hip: c21hip_wev_prepare) -- no 3-D window table is built or read,
     * and two radii share one pass-X sweep of the spectrum (read once, windowed and transformed
     * twice into work / work2), as in the excursion-set loop. */
    int wev = 0;
    if (n_f > 0) {
        float radii[C21CM_MAX_TS_RADII];
        for (int k = 0; k < n_f; k++) radii[k] = (float)s->R[filtered[k]];
        const int pair_ok = c21hip_pair_sweep_supported(c->nx);
        TRY(c21hip_wev_prepare(s->filter_type, 0.f, s->filter_type, 0.f, 1, radii, n_f, c->nx, c->ny,
                               c->nz, c->box_len, c->box_len_z, pair_ok, &wev, c->stream));
    }
    float *work2 = NULL;
    if (wev && c21hip_pair_sweep_supported(c->nx)) {
        static int pair = -1;
        if (pair < 0) {
            const char *e = getenv("C21CM_PAIR_RADII");
            pair = (e && e[0] == '0') ? 0 : 1;
        }
        if (pair)
            work2 = (float *)c21hip_ws(WS_TF_WORK2, c21hip_split_floats(c->nx, c->ny, c->nz) * sizeof(float));
    }
    const int tab_async = !wev && tf_tab.ok && c->ntot >= ((size_t)1 << 26); /* see ionize_driver.c */
    if (tab_async && n_f > 0) {
        TRY(c21hip_event_record(tf_tab.ev_sync, c->stream));
        TRY(c21hip_stream_wait_event(tf_tab.aux, tf_tab.ev_sync));
    }
    for (int r = 0; r < s->n_R; r++) {
        const int apply = s->R[r] > s->cell_radius;
        const float R = (float)s->R[r];
        float *d_out = host_out ? stage_out : result + (size_t)r * c->ntot;
        if (apply && work2 && r + 1 < s->n_R && s->R[r + 1] > s->cell_radius) {
            /* radii r and r + 1 out of one pass-X sweep; pass Y and pass Z per radius */
            const float R2 = (float)s->R[r + 1];
            TRY(c21hip_split_filter_xy_shared_pair(c->unf, c->work, work2, s->filter_type, c->nx, c->ny,
                                                   c->nz, c->box_len, c->box_len_z, R, R2, 0, 1, 2 | 4,
                                                   c->stream));
            TRY(c21hip_split_z_c2r_stats(c->work, d_out, c->nz, c->nx, c->ny, c->nz, s->min_value,
                                         s->const_factor, partials + (size_t)r * stride, NULL,
                                         c->stream));
            if (host_out) TRY(c21hip_d2h(result + (size_t)r * c->ntot, stage_out, bytes, c->stream));
            r++;
            d_out = host_out ? stage_out : result + (size_t)r * c->ntot;
            TRY(c21hip_split_filter_xy_shared_pair(c->unf, c->work, work2, s->filter_type, c->nx, c->ny,
                                                   c->nz, c->box_len, c->box_len_z, R, R2, 0, 1, 8,
                                                   c->stream));
            TRY(c21hip_split_z_c2r_stats(work2, d_out, c->nz, c->nx, c->ny, c->nz, s->min_value,
                                         s->const_factor, partials + (size_t)r * stride, NULL,
                                         c->stream));
            if (host_out) TRY(c21hip_d2h(result + (size_t)r * c->ntot, stage_out, bytes, c->stream));
            continue;
        }
        if (apply && tab_async) {
            const int k = next; /* this radius is filtered[k] */
            const int buf = k & 1;
            if (k == 0) {
                TRY(c21hip_window_tables(buf, s->filter_type, 0.f, s->filter_type, 0.f, c->nx, c->ny,
                                         c->nz, c->box_len, c->box_len_z, R, tf_tab.aux));
                TRY(c21hip_event_record(tf_tab.ev_table[buf], tf_tab.aux));
            }
            if (k + 1 < n_f) { /* the next filtered radius' table, into the other buffer */
                const int nb2 = buf ^ 1;
                if (k >= 1) TRY(c21hip_stream_wait_event(tf_tab.aux, tf_tab.ev_used[nb2]));
                TRY(c21hip_window_tables(nb2, s->filter_type, 0.f, s->filter_type, 0.f, c->nx,
                                         c->ny, c->nz, c->box_len, c->box_len_z,
                                         (float)s->R[filtered[k + 1]], tf_tab.aux));
                TRY(c21hip_event_record(tf_tab.ev_table[nb2], tf_tab.aux));
            }
            TRY(c21hip_stream_wait_event(c->stream, tf_tab.ev_table[buf]));
            TRY(c21hip_split_filter_xy_shared(c->unf, c->work, s->filter_type, c->nx, c->ny, c->nz,
                                              c->box_len, c->box_len_z, R, 1, buf, c->stream));
            TRY(c21hip_event_record(tf_tab.ev_used[buf], c->stream));
            next++;
        } else {
            TRY(c21hip_split_filter_xy(c->unf, c->work, c->nx, c->ny, c->nz, c->box_len,
                                       c->box_len_z, s->filter_type, R, 0.f, apply, c->stream));
        }
        TRY(c21hip_split_z_c2r_stats(c->work, d_out, c->nz, c->nx, c->ny, c->nz, s->min_value,
                                     s->const_factor, partials + (size_t)r * stride, NULL,
                                     c->stream));
        if (host_out) TRY(c21hip_d2h(result + (size_t)r * c->ntot, stage_out, bytes, c->stream));
    }
    TRY(c21hip_batched_stats(partials, stride, (int)nb, s->n_R, c->stats, c->stream));
done:
    c21hip_wev_release();
    return status;
}

int c21cm_fill_Rbox_grids(const c21cm_rbox_spec *s, const float *input, float *result,
                          double *min_arr, double *average_arr, double *max_arr, void *stream) {
    int status = 0;
    if (!s || !input || !result || s->n_R < 1 || s->n_R > C21CM_MAX_TS_RADII) {
        c21hip_set_error("fill_Rbox: input, result and 1 <= n_R <= %d are required",
                         C21CM_MAX_TS_RADII);
        return C21CM_VALUE_ERROR;
    }
    if (s->filter_type < 0 || s->filter_type > 2) {
        c21hip_set_error("fill_Rbox: HEAT_FILTER must be 0 (top-hat), 1 (sharp-k) or 2 (Gaussian)");
        return C21CM_VALUE_ERROR;
    }
    tf_ctx c;
    TRY(tf_setup(&c, s->hii_dim, s->hii_dim_z, s->box_len, s->box_len_z, s->n_R, stream));
    const size_t bytes = c.ntot * sizeof(float);
    const float *d_in = tf_stage_in(input, bytes, stream, &status);
    if (status) return status;
    const int host_out = !c21hip_is_device_ptr(result);
    float *stage_out = host_out ? (float *)c21hip_ws(WS_TF_OUT, bytes) : NULL;
    if (host_out && !stage_out) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(tf_forward(&c, d_in));
    if (c.native) {
        TRY(fill_Rbox_native(&c, s, result, host_out, stage_out));
    } else {
        for (int r = 0; r < s->n_R; r++) {
            const double R = s->R[r];
            float *d_out = host_out ? stage_out : result + (size_t)r * c.ntot;
            TRY(tf_window(&c, s->filter_type, (float)R, 0.f, R > s->cell_radius, s->min_value,
                          s->const_factor, d_out, c.stats + 3 * r));
            /* stream order keeps the staging buffer safe: the next store follows the copy */
            if (host_out) TRY(c21hip_d2h(result + (size_t)r * c.ntot, stage_out, bytes, stream));
        }
    }
    {
        double host_stats[3 * C21CM_MAX_TS_RADII];
        TRY(c21hip_d2h(host_stats, c.stats, 3 * (size_t)s->n_R * sizeof(double), stream));
        TRY(c21hip_sync(stream));
        for (int r = 0; r < s->n_R; r++) {
            if (min_arr) min_arr[r] = host_stats[3 * r];
            if (max_arr) max_arr[r] = host_stats[3 * r + 1];
            if (average_arr) average_arr[r] = host_stats[3 * r + 2] / (double)c.ntot;
        }
    }
done:
    return status;
}

int c21cm_annular_filter_grids(const c21cm_annular_spec *s, const float *const *inputs,
                               float *const *outputs, double *u_avg, double *f_avg,
                               void *stream) {
    int status = 0;
    if (!s || !inputs || !outputs || s->n_grids < 1 || s->n_grids > C21CM_MAX_ANNULAR_GRIDS) {
        c21hip_set_error("annular filter: 1 <= n_grids <= %d input/output grids are required",
                         C21CM_MAX_ANNULAR_GRIDS);
        return C21CM_VALUE_ERROR;
    }
    for (int g = 0; g < s->n_grids; g++) {
        if (!inputs[g] || !outputs[g]) {
            c21hip_set_error("annular filter: grid %d is missing", g);
            return C21CM_VALUE_ERROR;
        }
        if (s->filter_type[g] != 4 && s->filter_type[g] != 5) {
            c21hip_set_error("annular filter: window type %d (4 = spherical shell, 5 = multiple "
                             "scattering)", s->filter_type[g]);
            return C21CM_VALUE_ERROR;
        }
    }
    tf_ctx c;
    TRY(tf_setup(&c, s->hii_dim, s->hii_dim_z, s->box_len, s->box_len_z, 2 * s->n_grids, stream));
    const size_t bytes = c.ntot * sizeof(float);
    const size_t sbytes = (c.native ? c21hip_split_floats(c.nx, c.ny, c.nz) : c.npad) * sizeof(float);
    /* filter_box takes float radii (filtering.c:308); the cell-scale shell is not filtered */
    const float Ri = (float)s->R_inner, Ro = (float)s->R_outer, Rs = (float)s->R_star;
    const int apply = s->R_inner > 0;
    /* grids go through in pairs: one pass X / pass Y launch filters two spectra */
    for (int g0 = 0; g0 < s->n_grids; g0 += 2) {
        const int np = (s->n_grids - g0 >= 2) ? 2 : 1;
        float *unf[2] = {c.unf, NULL}, *work[2] = {c.work, NULL}, *d_out[2] = {NULL, NULL};
        int host_out[2] = {0, 0};
        if (np == 2) {
            unf[1] = (float *)c21hip_ws(WS_TF_UNF2, sbytes);
            work[1] = c21_place_work_partner(WS_TF_WORK, WS_TF_WORK2, sbytes, c.nx, c.ny, c.nz, stream);
            if (!unf[1] || !work[1]) return C21CM_MEMORY_ALLOC_ERROR;
        }
        for (int p = 0; p < np; p++) {
            const int g = g0 + p;
            const float *d_in = inputs[g];
            if (!c21hip_is_device_ptr(d_in)) {
                float *st_in = (float *)c21hip_ws(p ? WS_TF_IN2 : WS_TF_IN, bytes);
                if (!st_in) return C21CM_MEMORY_ALLOC_ERROR;
                TRY(c21hip_h2d(st_in, d_in, bytes, stream));
                d_in = st_in;
            }
            host_out[p] = !c21hip_is_device_ptr(outputs[g]);
            d_out[p] = host_out[p] ? (float *)c21hip_ws(p ? WS_TF_OUT2 : WS_TF_OUT, bytes) : outputs[g];
            if (!d_out[p]) return C21CM_MEMORY_ALLOC_ERROR;
            /* box average of the input (:660-675) */
            TRY(c21hip_floor_scale_stats(d_in, c.nz, NULL, c.nx, c.ny, c.nz, -INFINITY, 1.0,
                                         c.partials, c.stats + 6 * g, stream));
            c.unf = unf[p];
            TRY(tf_forward(&c, d_in));
        }
        if (c.native) {
            TRY(c21hip_split_filter_shell(unf[0], work[0], s->filter_type[g0], unf[1], work[1],
                                          np == 2 ? s->filter_type[g0 + 1] : 0, np, c.nx, c.ny, c.nz,
                                          c.box_len, c.box_len_z, Ri, Ro, Rs, apply, stream));
            for (int p = 0; p < np; p++)
                TRY(c21hip_split_z_c2r_stats(work[p], d_out[p], c.nz, c.nx, c.ny, c.nz, 0., 1.0,
                                             c.partials, c.stats + 6 * (g0 + p) + 3, stream));
        } else {
            for (int p = 0; p < np; p++) {
                TRY(c21hip_copy_filter_star(unf[p], work[p], c.nx, c.ny, c.nz, c.box_len, c.box_len_z,
                                            s->filter_type[g0 + p], Ri, Ro, Rs, apply, stream));
                TRY(c21hip_fft_c2r(work[p], c.nx, c.ny, c.nz, stream));
                TRY(c21hip_floor_scale_stats(work[p], 2 * (long)(c.nz / 2 + 1), d_out[p], c.nx, c.ny,
                                             c.nz, 0., 1.0, c.partials,
                                             c.stats + 6 * (g0 + p) + 3, stream));
            }
        }
        for (int p = 0; p < np; p++)
            if (host_out[p]) TRY(c21hip_d2h(outputs[g0 + p], d_out[p], bytes, stream));
        c.unf = unf[0];
    }
    {
        double host_stats[6 * C21CM_MAX_ANNULAR_GRIDS];
        TRY(c21hip_d2h(host_stats, c.stats, 6 * (size_t)s->n_grids * sizeof(double), stream));
        TRY(c21hip_sync(stream));
        for (int g = 0; g < s->n_grids; g++) {
            if (u_avg) u_avg[g] = host_stats[6 * g + 2] / (double)c.ntot;
            if (f_avg) f_avg[g] = host_stats[6 * g + 5] / (double)c.ntot;
        }
    }
done:
    return status;
}
