/*
 * ics_driver.c -- C host driver of the ComputeInitialConditions grid algorithm.
 *
 * Order of operations = the reference's ComputeInitialConditions
 * (src/py21cmfast/src/InitialConditions.c:547-772): delta_k (sampled, or r2c of the
 * supplied hires_density), hi-res density, filtered + subsampled low-res density,
 * first-order velocities (3 gradients), 2LPT source (6 second derivatives, products,
 * r2c) and its 3 gradients: 14 hi-res c2r + 1-2 r2c transforms and ~25 streaming sweeps,
 * all resident in HBM.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

#define L_FACTOR 0.620350491 /* reference: src/py21cmfast/src/Constants.c:41 */

enum {
    WS_IC_BOX = 50,
    WS_IC_SAVED,
    WS_IC_PHI,
    WS_IC_DIAG0, /* +1, +2 */
    WS_IC_PK = 55,
    WS_IC_IN = 56,
    WS_IC_OUT0 = 57, /* staged outputs, reused one at a time */
    WS_IC_DEVIATES = 58,
    WS_IC_VCBTAB = 59
};

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

/* C21CM_RNG_GSL: the reference's random stream (gsl_stream.c) drawn on the host in the
 * reference's order and staged in HBM for sample_modes_kernel; NULL for the Philox stream. */
int c21_gsl_mode_deviates(unsigned long long seed, int n_threads, int nx, int ny, int nzc,
                          double *ab);
int c21_gsl_mode_deviates_device(unsigned long long seed, int n_threads, int nx, int ny, int nzc,
                                 double *dev_ab, void *stream);
static int stream_deviates(const c21cm_ics_spec *s, void *stream, const double **dev_ab) {
    *dev_ab = NULL;
    if (s->rng_stream == C21CM_RNG_PHILOX) return 0;
    if (s->rng_stream != C21CM_RNG_GSL) {
        c21hip_set_error("ics: unknown rng_stream %d", s->rng_stream);
        return C21CM_VALUE_ERROR;
    }
    const int nzc = s->dim_z / 2 + 1;
    const size_t bytes = 2 * sizeof(double) * (size_t)s->dim * s->dim * nzc;
    {
        /* default: raw words staged chunk by chunk while the streams are drawn, ln / sqrt on the
         * device (C21CM_GSL_DEVIATES=host: the deviates computed on the host, one blocking copy) */
        const char *e = getenv("C21CM_GSL_DEVIATES");
        if (!(e && e[0] == 'h')) {
            double *dev = (double *)c21hip_ws(WS_IC_DEVIATES, bytes);
            if (!dev) return C21CM_MEMORY_ALLOC_ERROR;
            *dev_ab = dev;
            return c21_gsl_mode_deviates_device(s->seed, s->rng_threads > 0 ? s->rng_threads : 1, s->dim,
                                                s->dim, nzc, dev, stream);
        }
    }
    double *host = (double *)malloc(bytes);
    double *dev = (double *)c21hip_ws(WS_IC_DEVIATES, bytes);
    if (!host || !dev) {
        free(host);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    int st = c21_gsl_mode_deviates(s->seed, s->rng_threads > 0 ? s->rng_threads : 1, s->dim, s->dim,
                                   nzc, host);
    if (!st) st = c21hip_h2d(dev, host, bytes, stream);
    if (!st) st = c21hip_sync(stream); /* `host` is freed below */
    free(host);
    *dev_ab = dev;
    return st;
}

/* gather a padded hi-res grid into a (host or device) dense output array */
static int emit(const float *box, const int hi_dim[3], float *target, const int dim[3],
                float divisor, void *stream) {
    if (!target) return 0;
    const size_t n = (size_t)dim[0] * dim[1] * dim[2];
    float *d_out = target;
    const int host = !c21hip_is_device_ptr(target);
    if (host) {
        d_out = (float *)c21hip_ws(WS_IC_OUT0, n * sizeof(float));
        if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
    }
    int st = c21hip_gather(box, hi_dim, d_out, dim, 0, divisor, 0, stream);
    if (st) return st;
    if (host) {
        if ((st = c21hip_d2h(target, d_out, n * sizeof(float), stream))) return st;
        /* the staging slot is reused by the next field */
        if ((st = c21hip_sync(stream))) return st;
    }
    return 0;
}


/* ---- split-layout pipeline -----------------------------------------------------------------
 * For grids the native transform covers: the spectra stay in the split layout (no padded <->
 * split conversion per transform), pass Z stores dense outputs with the "/ VOLUME" folded in (no
 * gather sweep), the six 2LPT second derivatives meet in one sweep, and every LOW-RESOLUTION
 * output (lowres_density, lowres_v*, lowres_v*_2LPT: 7 of the 14 inverse transforms of the
 * default configuration) is the small transform of the FOLDED spectrum (c21hip_split_fold)
 * instead of a DIM^3 transform followed by subsampling.  C21CM_ICS=padded selects the older
 * padded-layout pipeline below. */
enum {
    WS_IS_SAVED = 100, WS_IS_FILT, WS_IS_WORK, WS_IS_LO, WS_IS_LOWORK, WS_IS_BOX,
    WS_IS_D0 = 106, /* 107, 108 */
    WS_IS_O0 = 109, /* 110, 111 */
    WS_IS_OUT = 112, WS_IS_IN = 113, WS_IS_PK2 = 114,
    WS_IS_LO1 = 115 /* 116, 117: the folded spectra of one fused fold (lo_fields) */
};

typedef struct {
    const c21cm_ics_spec *s;
    int hi[3], lo[3], f;
    size_t ntot, sfl_hi, sfl_lo;
    float *saved, *filt, *work, *lo_k, *lo_work;
    float *pk2;   /* spectrum / k^2 of `saved` (lazily built): operand of the fused operators */
    int pk2_valid;
    void *stream;
} is_ctx;

/* dense output array (host or device) written by pass Z: returns the device pointer to store to */
static float *out_target(float *target, size_t n, int *is_host) {
    *is_host = !c21hip_is_device_ptr(target);
    return *is_host ? (float *)c21hip_ws(WS_IS_OUT, n * sizeof(float)) : target;
}
static int out_finish(float *target, const float *dev, size_t n, int is_host, void *stream) {
    if (!is_host) return 0;
    int st = c21hip_d2h(target, dev, n * sizeof(float), stream);
    if (st) return st;
    return c21hip_sync(stream); /* the staging slot is reused by the next field */
}

/* full-resolution field: op(spectrum) -> dense real / divisor */
static int hi_field(is_ctx *c, const float *spec, int axis0, int axis1, float *target,
                    float divisor) {
    if (!target) return 0;
    int status = 0, is_host;
    float *d_out = out_target(target, c->ntot, &is_host);
    if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
    const c21cm_ics_spec *s = c->s;
    if (axis0 >= 0) {
        /* op(delta_k) = separable factor x (delta_k / k^2): the division once per spectrum, the
         * rest inside pass X */
        if (!c->pk2_valid) {
            c->pk2 = (float *)c21hip_ws(WS_IS_PK2, c->sfl_hi * sizeof(float));
            if (!c->pk2) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_split_kop(spec, c->pk2, c->hi[0], c->hi[1], c->hi[2], s->box_len,
                                 s->box_len_z, -2, -1, c->stream));
            c->pk2_valid = 1;
        }
        TRY(c21hip_split_sepop_xy(c->pk2, c->work, c->hi[0], c->hi[1], c->hi[2], s->box_len,
                                  s->box_len_z, axis0, axis1, c->stream));
    } else {
        TRY(c21hip_split_filter_xy(spec, c->work, c->hi[0], c->hi[1], c->hi[2], s->box_len,
                                   s->box_len_z, 0, 0.f, 0.f, 0, c->stream));
    }
    TRY(c21hip_split_z_c2r_div(c->work, d_out, c->hi[2], c->hi[0], c->hi[1], c->hi[2], divisor,
                               c->stream));
    TRY(out_finish(target, d_out, c->ntot, is_host, c->stream));
done:
    return status;
}

/* low-resolution field: fold(op(filtered spectrum)) -> small transform -> dense real / divisor */
static int lo_field(is_ctx *c, const float *filt_spec, int axis0, int axis1, float *target,
                    float divisor) {
    if (!target) return 0;
    int status = 0, is_host;
    const size_t nlo = (size_t)c->lo[0] * c->lo[1] * c->lo[2];
    float *d_out = out_target(target, nlo, &is_host);
    if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
    const c21cm_ics_spec *s = c->s;
    TRY(c21hip_split_fold(filt_spec, c->lo_k, c->hi[0], c->hi[1], c->hi[2], c->f, s->box_len,
                          s->box_len_z, axis0, axis1, c->stream));
    /* the folded spectrum lives on the lo grid of the SAME box: lengths unchanged */
    TRY(c21hip_split_filter_xy(c->lo_k, c->lo_work, c->lo[0], c->lo[1], c->lo[2], s->box_len,
                               s->box_len_z, 0, 0.f, 0.f, 0, c->stream));
    TRY(c21hip_split_z_c2r_div(c->lo_work, d_out, c->lo[2], c->lo[0], c->lo[1], c->lo[2], divisor,
                               c->stream));
    TRY(out_finish(target, d_out, nlo, is_host, c->stream));
done:
    return status;
}

/* Several low-resolution fields of ONE filtered spectrum (the density and / or the three velocity components):
 * one fold launch reads the DIM^3 spectrum for all of them (round 6; C21CM_ICS_FOLD=single: one launch each). */
static int lo_fields(is_ctx *c, const float *filt_spec, int n, const int ops[4], float *const targets[4],
                     float divisor, float tophat_R) {
    int status = 0;
    const c21cm_ics_spec *s = c->s;
    const char *e = getenv("C21CM_ICS_FOLD");
    int m = 0, use_ops[4];
    float *use_t[4], *lo_k[4];
    for (int k = 0; k < n; k++)
        if (targets[k]) use_ops[m] = ops[k], use_t[m] = targets[k], m++;
    if (m == 0) return 0;
    if (m == 1 || (e && e[0] == 's')) {
        if (tophat_R > 0.f) { /* the separate sweep of round 5: the filtered spectrum written, then read per field */
            TRY(c21hip_copy_filter_split(filt_spec, c->filt, c->hi[0], c->hi[1], c->hi[2], s->box_len, s->box_len_z, 0,
                                         tophat_R, 0.f, 1, c->stream));
            filt_spec = c->filt;
        }
        for (int k = 0; k < m; k++) TRY(lo_field(c, filt_spec, use_ops[k], -1, use_t[k], divisor));
        return 0;
    }
    lo_k[0] = c->lo_k;
    for (int k = 1; k < m; k++)
        if (!(lo_k[k] = (float *)c21hip_ws(WS_IS_LO1 + k - 1, c->sfl_lo * sizeof(float)))) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_split_fold_multi(filt_spec, lo_k, use_ops, m, c->hi[0], c->hi[1], c->hi[2], c->f, s->box_len,
                                s->box_len_z, tophat_R, c->stream));
    const size_t nlo = (size_t)c->lo[0] * c->lo[1] * c->lo[2];
    for (int k = 0; k < m; k++) {
        int is_host;
        float *d_out = out_target(use_t[k], nlo, &is_host);
        if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_split_filter_xy(lo_k[k], c->lo_work, c->lo[0], c->lo[1], c->lo[2], s->box_len, s->box_len_z, 0,
                                   0.f, 0.f, 0, c->stream));
        TRY(c21hip_split_z_c2r_div(c->lo_work, d_out, c->lo[2], c->lo[0], c->lo[1], c->lo[2], divisor, c->stream));
        TRY(out_finish(use_t[k], d_out, nlo, is_host, c->stream));
    }
done:
    return status;
}

static int ics_split_supported(const c21cm_ics_spec *s) {
    const char *e = getenv("C21CM_ICS");
    if (e && e[0] == 'p') return 0; /* C21CM_ICS=padded */
    if (s->vcb_by_m) return 0; /* relative velocities work on the padded spectrum (rare option) */
    /* (DIM >= 1024: the main blocks are x-blocked; the element-wise kernels of this pipeline map
     * memory lines to wavenumbers through split_layout.h -- round 4, DIM = 1536 / HII_DIM = 512) */
    if (!c21hip_fft_is_native(s->dim, s->dim, s->dim_z)) return 0;
    if (s->dim == s->hii_dim && s->dim_z == s->hii_dim_z) return 1;
    if (s->perturb_on_high_res && s->dim % s->hii_dim) return 0;
    const int f = s->dim / s->hii_dim;
    if ((f != 2 && f != 3 && f != 4) || s->hii_dim * f != s->dim || s->hii_dim_z * f != s->dim_z) return 0;
    return c21hip_fft_is_native(s->hii_dim, s->hii_dim, s->hii_dim_z);
}

static int ics_grids_split(const c21cm_ics_spec *s, InitialConditions *ics, float *const vel[3],
                           float *const vel2[3], void *stream) {
    int status = 0;
    is_ctx c;
    memset(&c, 0, sizeof(c));
    c.s = s;
    c.stream = stream;
    c.hi[0] = c.hi[1] = s->dim;
    c.hi[2] = s->dim_z;
    c.lo[0] = c.lo[1] = s->hii_dim;
    c.lo[2] = s->hii_dim_z;
    c.f = s->dim / s->hii_dim;
    c.ntot = (size_t)c.hi[0] * c.hi[1] * c.hi[2];
    c.sfl_hi = c21hip_split_floats(c.hi[0], c.hi[1], c.hi[2]);
    c.sfl_lo = c21hip_split_floats(c.lo[0], c.lo[1], c.lo[2]);
    const int hires = s->perturb_on_high_res;
    const int need_filter = (s->dim != s->hii_dim);
    const int lpt2 = (s->perturb_algorithm == C21CM_PERTURB_2LPT);
    const float VOLUME = s->volume;
    const float R_lo = (float)(L_FACTOR * s->box_len / (s->hii_dim + 0.0));
    c.saved = (float *)c21hip_ws(WS_IS_SAVED, c.sfl_hi * sizeof(float));
    c.work = (float *)c21hip_ws(WS_IS_WORK, c.sfl_hi * sizeof(float));
    c.filt = need_filter ? (float *)c21hip_ws(WS_IS_FILT, c.sfl_hi * sizeof(float)) : c.saved;
    c.lo_k = (float *)c21hip_ws(WS_IS_LO, c.sfl_lo * sizeof(float));
    c.lo_work = (float *)c21hip_ws(WS_IS_LOWORK, c.sfl_lo * sizeof(float));
    if (!c.saved || !c.work || !c.filt || !c.lo_k || !c.lo_work) return C21CM_MEMORY_ALLOC_ERROR;

    if (s->density_is_input) {
        /* InitialConditions.c:636-663: delta_k = r2c(hires_density * VOLUME / N) */
        const float *d_in = ics->hires_density;
        if (!c21hip_is_device_ptr(d_in)) {
            float *tmp = (float *)c21hip_ws(WS_IS_IN, c.ntot * sizeof(float));
            if (!tmp) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_h2d(tmp, ics->hires_density, c.ntot * sizeof(float), stream));
            d_in = tmp;
        }
        /* the float product / quotient of :650-651 is done on load in double and rounded once */
        TRY(c21hip_split_r2c(d_in, c.hi[2], c.saved, c.hi[0], c.hi[1], c.hi[2],
                             (double)VOLUME / (double)(float)c.ntot, 1., -1., 1.0f, stream));
    } else {
        /* InitialConditions.c:664-692 */
        const int n_m = 3 * (s->dim / 2) * (s->dim / 2) + 1;
        if (!s->pk_by_m || s->n_m < n_m || s->dim != s->dim_z || s->box_len != s->box_len_z) {
            c21hip_set_error("ics: mode sampling needs a cubic grid and pk_by_m[0..3(DIM/2)^2]");
            return C21CM_VALUE_ERROR;
        }
        double *pk_dev = (double *)c21hip_ws(WS_IC_PK, (size_t)n_m * sizeof(double));
        if (!pk_dev) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_h2d(pk_dev, s->pk_by_m, (size_t)n_m * sizeof(double), stream));
        const double *dev_ab = NULL;
        TRY(stream_deviates(s, stream, &dev_ab));
        /* (round 6: the sampler writes the split layout itself -- no padded copy, no conversion sweep) */
        TRY(c21hip_sample_modes_split(c.saved, c.hi[0], c.hi[1], c.hi[2], pk_dev, VOLUME, s->seed, dev_ab, stream));
        TRY(hi_field(&c, c.saved, -1, -1, ics->hires_density, VOLUME));
    }
    /* the top-hat at the low-resolution cell scale is common to every low-resolution output and
     * commutes with the k-space operators: applied once (InitialConditions.c:700-703,330-333) */
    if (need_filter && hires)
        TRY(c21hip_copy_filter_split(c.saved, c.filt, c.hi[0], c.hi[1], c.hi[2], s->box_len,
                                     s->box_len_z, 0, R_lo, 0.f, 1, stream));
    if (need_filter && !hires) {
        /* lowres_density and the three first-order velocities (InitialConditions.c:299-364): folds of one spectrum,
         * the top-hat applied to the aliases on the way (round 6: no filtered copy of the DIM^3 spectrum) */
        const int ops[4] = {-1, 0, 1, 2};
        float *const tg[4] = {ics->lowres_density, vel[0], vel[1], vel[2]};
        TRY(lo_fields(&c, c.saved, 4, ops, tg, VOLUME, R_lo));
    } else {
        if (need_filter)
            TRY(lo_field(&c, c.filt, -1, -1, ics->lowres_density, VOLUME));
        else
            TRY(hi_field(&c, c.saved, -1, -1, ics->lowres_density, VOLUME));
        /* first-order velocities: InitialConditions.c:299-364 */
        for (int ii = 0; ii < 3; ii++) TRY(hi_field(&c, c.saved, ii, -1, vel[ii], VOLUME));
    }

    if (lpt2) {
        /* InitialConditions.c:366-545 */
        float *d[3], *o[3];
        for (int k = 0; k < 3; k++) {
            d[k] = (float *)c21hip_ws(WS_IS_D0 + k, c.ntot * sizeof(float));
            o[k] = (float *)c21hip_ws(WS_IS_O0 + k, c.ntot * sizeof(float));
            if (!d[k] || !o[k]) return C21CM_MEMORY_ALLOC_ERROR;
        }
        float *box = (float *)c21hip_ws(WS_IS_BOX, (c.ntot + 2 * (size_t)c.hi[0] * c.hi[1]) * sizeof(float));
        if (!box) return C21CM_MEMORY_ALLOC_ERROR;
        static const int dirs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int k = 0; k < 3; k++) {
            TRY(hi_field(&c, c.saved, k, k, d[k], 0.f));
            TRY(hi_field(&c, c.saved, dirs[k][0], dirs[k][1], o[k], 0.f));
        }
        /* /= VOLUME*VOLUME*TOT_NUM_PIXELS: float * float * (ull -> float), :493 */
        const float norm = VOLUME * VOLUME * (float)c.ntot;
        {
            const float *dc[3] = {d[0], d[1], d[2]}, *oc[3] = {o[0], o[1], o[2]};
            TRY(c21hip_lpt2_source(dc, oc, box, c.ntot, norm, stream));
        }
        TRY(c21hip_split_r2c(box, c.hi[2], c.saved, c.hi[0], c.hi[1], c.hi[2], 1.0, 1., -1., 1.0f,
                             stream));
        c.pk2_valid = 0; /* `saved` now holds the 2LPT source */
        if (hires || !need_filter) {
            for (int ii = 0; ii < 3; ii++) TRY(hi_field(&c, c.saved, ii, -1, vel2[ii], 0.f));
        } else {
            const int ops[4] = {0, 1, 2, -1};
            float *const tg[4] = {vel2[0], vel2[1], vel2[2], NULL};
            TRY(lo_fields(&c, c.saved, 3, ops, tg, 0.f, R_lo));
        }
    }
    TRY(c21hip_sync(stream));
done:
    return status;
}

int c21cm_ics_grids(const c21cm_ics_spec *s, InitialConditions *ics, void *stream) {
    int status = 0;
    if (!s || !ics || !ics->hires_density || !ics->lowres_density) {
        c21hip_set_error("ics: NULL spec / hires_density / lowres_density");
        return C21CM_VALUE_ERROR;
    }
    const int hi_dim[3] = {s->dim, s->dim, s->dim_z};
    const int lo_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const int hires = s->perturb_on_high_res;
    const int *pt_dim = hires ? hi_dim : lo_dim;
    const size_t npad = (size_t)hi_dim[0] * hi_dim[1] * 2 * (size_t)(hi_dim[2] / 2 + 1);
    const size_t ntot = (size_t)hi_dim[0] * hi_dim[1] * hi_dim[2];
    const float VOLUME = s->volume;
    const float R_lo = (float)(L_FACTOR * s->box_len / (s->hii_dim + 0.0));
    const int need_filter = (s->dim != s->hii_dim);
    const int lpt2 = (s->perturb_algorithm == C21CM_PERTURB_2LPT);
    float *vel[3], *vel2[3];
    if (hires) {
        vel[0] = ics->hires_vx; vel[1] = ics->hires_vy; vel[2] = ics->hires_vz;
        vel2[0] = ics->hires_vx_2LPT; vel2[1] = ics->hires_vy_2LPT; vel2[2] = ics->hires_vz_2LPT;
    } else {
        vel[0] = ics->lowres_vx; vel[1] = ics->lowres_vy; vel[2] = ics->lowres_vz;
        vel2[0] = ics->lowres_vx_2LPT; vel2[1] = ics->lowres_vy_2LPT; vel2[2] = ics->lowres_vz_2LPT;
    }
    if (!vel[0] || !vel[1] || !vel[2] || (lpt2 && (!vel2[0] || !vel2[1] || !vel2[2]))) {
        c21hip_set_error("ics: velocity output arrays are missing");
        return C21CM_VALUE_ERROR;
    }
    if (ics_split_supported(s)) return ics_grids_split(s, ics, vel, vel2, stream);

    float *box = (float *)c21hip_ws(WS_IC_BOX, npad * sizeof(float));
    float *saved = (float *)c21hip_ws(WS_IC_SAVED, npad * sizeof(float));
    if (!box || !saved) return C21CM_MEMORY_ALLOC_ERROR;

    if (s->density_is_input) {
        /* InitialConditions.c:636-663 */
        const float *d_in = ics->hires_density;
        if (!c21hip_is_device_ptr(d_in)) {
            float *tmp = (float *)c21hip_ws(WS_IC_IN, ntot * sizeof(float));
            if (!tmp) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_h2d(tmp, ics->hires_density, ntot * sizeof(float), stream));
            d_in = tmp;
        }
        TRY(c21hip_pack_density(d_in, box, hi_dim[0], hi_dim[1], hi_dim[2], VOLUME, stream));
        TRY(c21hip_fft_r2c(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(c21hip_d2d(saved, box, npad * sizeof(float), stream));
    } else {
        /* InitialConditions.c:664-692 */
        const int n_m = 3 * (s->dim / 2) * (s->dim / 2) + 1;
        if (!s->pk_by_m || s->n_m < n_m || s->dim != s->dim_z || s->box_len != s->box_len_z) {
            c21hip_set_error("ics: mode sampling needs a cubic grid and pk_by_m[0..3(DIM/2)^2]");
            return C21CM_VALUE_ERROR;
        }
        double *pk_dev = (double *)c21hip_ws(WS_IC_PK, (size_t)n_m * sizeof(double));
        if (!pk_dev) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_h2d(pk_dev, s->pk_by_m, (size_t)n_m * sizeof(double), stream));
        const double *dev_ab = NULL;
        TRY(stream_deviates(s, stream, &dev_ab));
        TRY(c21hip_sample_modes(saved, hi_dim[0], hi_dim[1], hi_dim[2], pk_dev, VOLUME, s->seed,
                                dev_ab, stream));
        TRY(c21hip_d2d(box, saved, npad * sizeof(float), stream));
        TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(emit(box, hi_dim, ics->hires_density, hi_dim, VOLUME, stream));
    }
    /* low-res density: InitialConditions.c:694-730 (fused copy x top-hat) */
    TRY(c21hip_copy_filter(saved, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z, 0,
                           R_lo, 0.f, need_filter, stream));
    TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
    TRY(emit(box, hi_dim, ics->lowres_density, lo_dim, VOLUME, stream));

    if (s->vcb_by_m) { /* compute_relative_velocities: InitialConditions.c:141-238,733 */
        const int n_m = 3 * (s->dim / 2) * (s->dim / 2) + 1;
        if (!ics->lowres_vcb || s->dim != s->dim_z || s->n_m < n_m) {
            c21hip_set_error("ics: relative velocities need lowres_vcb, a cubic grid and vcb_by_m[0..3(DIM/2)^2]");
            return C21CM_VALUE_ERROR;
        }
        double *h_dev = (double *)c21hip_ws(WS_IC_VCBTAB, (size_t)n_m * sizeof(double));
        const size_t nlo = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
        float *d_vcb = ics->lowres_vcb;
        const int vcb_host = !c21hip_is_device_ptr(d_vcb);
        if (vcb_host) d_vcb = (float *)c21hip_ws(WS_IC_OUT0, nlo * sizeof(float));
        if (!h_dev || !d_vcb) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_h2d(h_dev, s->vcb_by_m, (size_t)n_m * sizeof(double), stream));
        for (int ii = 0; ii < 3; ii++) {
            TRY(c21hip_vcb_op(saved, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z,
                              ii, h_dev, stream));
            if (need_filter) /* "we only care about the lowres vcb box, so we filter it directly" */
                TRY(c21hip_copy_filter(box, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                       s->box_len_z, 0, R_lo, 0.f, 1, stream));
            TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
            TRY(c21hip_vcb_accumulate(box, hi_dim, d_vcb, lo_dim, ii == 0, ii == 2, VOLUME, stream));
        }
        if (vcb_host) {
            TRY(c21hip_d2h(ics->lowres_vcb, d_vcb, nlo * sizeof(float), stream));
            TRY(c21hip_sync(stream));
        }
    }

    /* first-order velocities: InitialConditions.c:299-364 */
    for (int ii = 0; ii < 3; ii++) {
        TRY(c21hip_kspace_op(saved, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z,
                             ii, -1, stream));
        if (!hires && need_filter)
            TRY(c21hip_copy_filter(box, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                   s->box_len_z, 0, R_lo, 0.f, 1, stream));
        TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(emit(box, hi_dim, vel[ii], pt_dim, VOLUME, stream));
    }

    if (lpt2) {
        /* InitialConditions.c:366-545 */
        float *phi = (float *)c21hip_ws(WS_IC_PHI, npad * sizeof(float));
        float *diag[3];
        for (int c = 0; c < 3; c++) diag[c] = (float *)c21hip_ws(WS_IC_DIAG0 + c, ntot * sizeof(float));
        if (!phi || !diag[0] || !diag[1] || !diag[2]) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_memset(box, 0, npad * sizeof(float), stream));
        for (int c = 0; c < 3; c++) {
            TRY(c21hip_kspace_op(saved, phi, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                 s->box_len_z, c, c, stream));
            TRY(c21hip_fft_c2r(phi, hi_dim[0], hi_dim[1], hi_dim[2], stream));
            TRY(c21hip_gather(phi, hi_dim, diag[c], hi_dim, 0, 0.f, 0, stream));
        }
        static const int dirs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int c = 0; c < 3; c++) {
            TRY(c21hip_kspace_op(saved, phi, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                 s->box_len_z, dirs[c][0], dirs[c][1], stream));
            TRY(c21hip_fft_c2r(phi, hi_dim[0], hi_dim[1], hi_dim[2], stream));
            TRY(c21hip_lpt2_accumulate(box, phi, diag[dirs[c][0]], diag[dirs[c][1]], hi_dim[0],
                                       hi_dim[1], hi_dim[2], stream));
        }
        /* /= VOLUME*VOLUME*TOT_NUM_PIXELS: float * float * (ull -> float), :493 */
        const float norm = VOLUME * VOLUME * (float)ntot;
        TRY(c21hip_divide_inplace(box, npad, norm, stream));
        TRY(c21hip_fft_r2c(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(c21hip_d2d(saved, box, npad * sizeof(float), stream));
        for (int ii = 0; ii < 3; ii++) {
            TRY(c21hip_kspace_op(saved, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                 s->box_len_z, ii, -1, stream));
            if (!hires && need_filter)
                TRY(c21hip_copy_filter(box, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                                       s->box_len_z, 0, R_lo, 0.f, 1, stream));
            TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
            TRY(emit(box, hi_dim, vel2[ii], pt_dim, 0.f, stream));
        }
    }
    TRY(c21hip_sync(stream));
done:
    return status;
}
