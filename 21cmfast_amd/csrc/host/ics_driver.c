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
    WS_IC_OUT0 = 57 /* staged outputs, reused one at a time */
};

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

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
        TRY(c21hip_sample_modes(saved, hi_dim[0], hi_dim[1], hi_dim[2], pk_dev, VOLUME, s->seed,
                                stream));
        TRY(c21hip_d2d(box, saved, npad * sizeof(float), stream));
        TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(emit(box, hi_dim, ics->hires_density, hi_dim, VOLUME, stream));
    }
    /* low-res density: InitialConditions.c:694-730 (fused copy x top-hat) */
    TRY(c21hip_copy_filter(saved, box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z, 0,
                           R_lo, 0.f, need_filter, stream));
    TRY(c21hip_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2], stream));
    TRY(emit(box, hi_dim, ics->lowres_density, lo_dim, VOLUME, stream));

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
