/*
 * perturb_driver.c -- C host driver of the ComputePerturbedField grid algorithm.
 *
 * Order of operations = the reference's ComputePerturbedField
 * (src/py21cmfast/src/PerturbedField.c:389-496):
 *   make_density_grid        CIC mass deposit (fp64 atomics) or linear scaling   :24-135
 *   [assign_to_lowres_grid]  hi-res perturbation: r2c, top-hat, c2r, subsample   :137-178
 *   normalise_delta_grid                                                         :180-210
 *   smooth_and_clip_density  r2c, [Gaussian], keep delta_k, c2r, /N, clip        :212-282
 *   compute_perturbed_velocities  per axis: k-space multiply, c2r, gather        :284-387
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

#define L_FACTOR 0.620350491 /* reference: src/py21cmfast/src/Constants.c:41 */

enum {
    WS_PT_LOW = 20,
    WS_PT_HIGH,
    WS_PT_SAVED,
    WS_PT_RESAMPLED,
    WS_PT_IN0, /* .. +7 staged IC arrays */
    WS_PT_OUT0 = 32, /* .. +3 staged outputs */
    WS_PT_SPLIT = 89 /* .. +2 split-layout spectra */
};

/* C21CM_PT=padded keeps the k-space part of the low-resolution branch on the padded layout */
static int pt_split_supported(const int lo_dim[3]) {
    const char *e = getenv("C21CM_PT");
    if (e && e[0] == 'p') return 0;
    return c21hip_fft_is_native(lo_dim[0], lo_dim[1], lo_dim[2]) &&
           !c21hip_split_xblock_log2(lo_dim[0]);
}

#define TRY(expr)         \
    do {                  \
        int st_ = (expr); \
        if (st_) {        \
            status = st_; \
            goto done;    \
        }                 \
    } while (0)

static const float *stage_in(int slot, const float *p, size_t bytes, void *stream, int *status) {
    if (!p || *status) return NULL;
    if (c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    int st = c21hip_h2d(d, p, bytes, stream);
    if (st) *status = st;
    return (const float *)d;
}

int c21cm_perturb_grids(const c21cm_perturb_spec *s, const InitialConditions *ics,
                        PerturbedField *pf, void *stream) {
    int status = 0;
    if (!s || !ics || !pf || !pf->density) {
        c21hip_set_error("perturb: NULL spec / ics / density");
        return C21CM_VALUE_ERROR;
    }
    if (s->hii_dim < 2 || s->dim < s->hii_dim) {
        c21hip_set_error("perturb: bad dimensions DIM=%d HII_DIM=%d", s->dim, s->hii_dim);
        return C21CM_VALUE_ERROR;
    }
    const int lo_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const int hi_dim[3] = {s->dim, s->dim, s->dim_z};
    const int hires = s->perturb_on_high_res;
    const int *box_dim = hires ? hi_dim : lo_dim;
    const int linear = (s->perturb_algorithm == C21CM_PERTURB_LINEAR);
    const int lpt2 = (s->perturb_algorithm == C21CM_PERTURB_2LPT);
    const size_t lo_tot = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
    const size_t hi_tot = (size_t)hi_dim[0] * hi_dim[1] * hi_dim[2];
    const size_t lo_npad = (size_t)lo_dim[0] * lo_dim[1] * 2 * (size_t)(lo_dim[2] / 2 + 1);
    const size_t hi_npad = (size_t)hi_dim[0] * hi_dim[1] * 2 * (size_t)(hi_dim[2] / 2 + 1);
    const size_t b_tot = hires ? hi_tot : lo_tot;
    const size_t b_npad = hires ? hi_npad : lo_npad;

    const float *vel_h[3], *vel2_h[3], *dens_box_h;
    if (hires) {
        vel_h[0] = ics->hires_vx; vel_h[1] = ics->hires_vy; vel_h[2] = ics->hires_vz;
        vel2_h[0] = ics->hires_vx_2LPT; vel2_h[1] = ics->hires_vy_2LPT; vel2_h[2] = ics->hires_vz_2LPT;
        dens_box_h = ics->hires_density;
    } else {
        vel_h[0] = ics->lowres_vx; vel_h[1] = ics->lowres_vy; vel_h[2] = ics->lowres_vz;
        vel2_h[0] = ics->lowres_vx_2LPT; vel2_h[1] = ics->lowres_vy_2LPT; vel2_h[2] = ics->lowres_vz_2LPT;
        dens_box_h = ics->lowres_density;
    }
    if (linear ? !dens_box_h : (!ics->hires_density || !vel_h[0] || !vel_h[1] || !vel_h[2] ||
                                (lpt2 && (!vel2_h[0] || !vel2_h[1] || !vel2_h[2])))) {
        c21hip_set_error("perturb: required InitialConditions arrays are missing");
        return C21CM_VALUE_ERROR;
    }
    if (s->keep_3d_velocities && (!pf->velocity_x || !pf->velocity_y)) {
        c21hip_set_error("perturb: KEEP_3D_VELOCITIES needs velocity_x / velocity_y");
        return C21CM_VALUE_ERROR;
    }

    float *lowres = (float *)c21hip_ws(WS_PT_LOW, lo_npad * sizeof(float));
    float *highres = hires ? (float *)c21hip_ws(WS_PT_HIGH, hi_npad * sizeof(float)) : NULL;
    float *saved = (float *)c21hip_ws(WS_PT_SAVED, b_npad * sizeof(float));
    if (!lowres || !saved || (hires && !highres)) return C21CM_MEMORY_ALLOC_ERROR;
    float *grid = hires ? highres : lowres;

    /* ---- make_density_grid */
    if (linear) {
        const float *d = stage_in(WS_PT_IN0, dens_box_h, b_tot * sizeof(float), stream, &status);
        if (status) return status;
        TRY(c21hip_scale_pack(d, grid, box_dim[0], box_dim[1], box_dim[2], s->growth_factor,
                              stream));
    } else {
        const float *d_dens =
            stage_in(WS_PT_IN0, ics->hires_density, hi_tot * sizeof(float), stream, &status);
        const float *vel[3], *vel2[3] = {NULL, NULL, NULL};
        for (int a = 0; a < 3; a++) {
            vel[a] = stage_in(WS_PT_IN0 + 1 + a, vel_h[a], b_tot * sizeof(float), stream, &status);
            if (lpt2)
                vel2[a] = stage_in(WS_PT_IN0 + 4 + a, vel2_h[a], b_tot * sizeof(float), stream,
                                   &status);
        }
        if (status) return status;
        double *resampled = (double *)c21hip_ws(WS_PT_RESAMPLED, b_tot * sizeof(double));
        if (!resampled) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_memset(resampled, 0, b_tot * sizeof(double), stream));
        int fixed = 0; /* 1: the deposit left 2^44 fixed-point integers (order-independent sums) */
        TRY(c21hip_cic_scatter(d_dens, hi_dim, vel, vel2, box_dim, resampled, box_dim, s->box_len,
                               s->box_len_z, s->growth_factor, s->init_growth_factor, lpt2, &fixed,
                               stream));
        if (fixed) { /* the integers would wrap silently where the fp64 sums carried NaN / Inf to the caller */
            int bad = 0;
            TRY(c21hip_cic_fixed_status(&bad, stream));
            if (bad) {
                c21hip_set_error("perturb: non-finite (or absurdly large) hi-res density / displacement in the mass deposit");
                return C21CM_INFINITY_OR_NAN_ERROR;
            }
        }
        /* widen (+ normalise when the deposit already happened on the output grid) */
        const double mass_factor = lo_tot / (double)hi_tot;
        TRY(c21hip_widen_normalise(resampled, grid, box_dim[0], box_dim[1], box_dim[2],
                                   !hires /* normalise_delta_grid */, mass_factor, fixed, stream));
    }
    if (hires) {
        /* ---- assign_to_lowres_grid */
        TRY(c21hip_fft_r2c(highres, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(c21hip_d2d(saved, highres, hi_npad * sizeof(float), stream));
        TRY(c21hip_copy_filter(saved, highres, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len,
                               s->box_len_z, 0, (float)(L_FACTOR * s->box_len / (lo_dim[0] + 0.0)),
                               0.f, 1, stream));
        TRY(c21hip_fft_c2r(highres, hi_dim[0], hi_dim[1], hi_dim[2], stream));
        TRY(c21hip_gather(highres, hi_dim, lowres, lo_dim, 1, (float)hi_tot, 0, stream));
        if (!linear) {
            /* normalise_delta_grid with mass_factor = 1 (PerturbedField.c:188-190): v*1 - 1 */
            TRY(c21hip_add_scalar(lowres, lo_npad, -1.0f, stream));
        }
    }
    if (!hires && pt_split_supported(lo_dim)) {
        /* k-space part on the split layout of the native transform (as the IC pipeline): the
         * forward transform writes it, the smoothing window is one sweep over it, pass Z stores
         * the dense outputs with "/ N" and the density floor folded in, and the velocity
         * operator i k_a / k^2 is applied inside pass X on the spectrum divided by k^2 once. */
        const size_t sfl = c21hip_split_floats(lo_dim[0], lo_dim[1], lo_dim[2]) * sizeof(float);
        float *spec = (float *)c21hip_ws(WS_PT_SPLIT, sfl);
        float *work = (float *)c21hip_ws(WS_PT_SPLIT + 1, sfl);
        float *pk2 = (float *)c21hip_ws(WS_PT_SPLIT + 2, sfl);
        if (!spec || !work || !pk2) return C21CM_MEMORY_ALLOC_ERROR;
        const long zs = 2 * (long)(lo_dim[2] / 2 + 1);
        TRY(c21hip_split_r2c(lowres, zs, spec, lo_dim[0], lo_dim[1], lo_dim[2], 1.0, 1., -1., 1.0f,
                             stream));
        if (s->smooth_evolved_density)
            TRY(c21hip_copy_filter_split(spec, spec, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len,
                                         s->box_len_z, 2, (float)s->density_smooth_radius_mpc, 0.f,
                                         1, stream));
        float *targets[4] = {pf->density, pf->velocity_x, pf->velocity_y, pf->velocity_z};
        int pk2_done = 0;
        for (int t = 0; t < 4; t++) {
            if (!targets[t]) continue;
            if (t >= 1 && (s->hii_dim <= 1 || (!s->keep_3d_velocities && t < 3))) continue;
            float *d_out = targets[t];
            const int host_out = !c21hip_is_device_ptr(targets[t]);
            if (host_out) d_out = (float *)c21hip_ws(WS_PT_OUT0 + t, lo_tot * sizeof(float));
            if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
            if (t == 0) { /* /N, clip (PerturbedField.c:251-276,450-464) */
                TRY(c21hip_split_filter_xy(spec, work, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len,
                                           s->box_len_z, 0, 0.f, 0.f, 0, stream));
                TRY(c21hip_split_z_c2r_out(work, d_out, lo_dim[2], lo_dim[0], lo_dim[1], lo_dim[2],
                                           1.0f, (float)lo_tot, 1, stream));
            } else { /* v = c2r(delta_k dD/dt/D i k_a / k^2 / N), PerturbedField.c:320-383 */
                if (!pk2_done) {
                    TRY(c21hip_split_kop(spec, pk2, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len,
                                         s->box_len_z, -2, -1, stream));
                    pk2_done = 1;
                }
                TRY(c21hip_split_sepop_xy(pk2, work, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len,
                                          s->box_len_z, t - 1, -1, stream));
                TRY(c21hip_split_z_c2r_out(work, d_out, lo_dim[2], lo_dim[0], lo_dim[1], lo_dim[2],
                                           (float)(s->dDdt_over_D / (double)lo_tot), 0.f, 0,
                                           stream));
            }
            if (host_out) TRY(c21hip_d2h(targets[t], d_out, lo_tot * sizeof(float), stream));
        }
        TRY(c21hip_sync(stream));
        goto done;
    }
    /* ---- smooth_and_clip_density */
    TRY(c21hip_fft_r2c(lowres, lo_dim[0], lo_dim[1], lo_dim[2], stream));
    if (s->smooth_evolved_density) {
        /* in-place filter: src == dst */
        TRY(c21hip_copy_filter(lowres, lowres, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len,
                               s->box_len_z, 2, (float)s->density_smooth_radius_mpc, 0.f, 1,
                               stream));
    }
    if (!hires) TRY(c21hip_d2d(saved, lowres, lo_npad * sizeof(float), stream));
    TRY(c21hip_fft_c2r(lowres, lo_dim[0], lo_dim[1], lo_dim[2], stream));
    {
        /* /N, clip, copy out (PerturbedField.c:251-276,450-464) */
        float *d_out = pf->density;
        const int host_out = !c21hip_is_device_ptr(pf->density);
        if (host_out) d_out = (float *)c21hip_ws(WS_PT_OUT0, lo_tot * sizeof(float));
        if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
        TRY(c21hip_gather(lowres, lo_dim, d_out, lo_dim, 0, (float)lo_tot, 1, stream));
        if (host_out) TRY(c21hip_d2h(pf->density, d_out, lo_tot * sizeof(float), stream));
    }
    /* ---- velocities */
    if (s->hii_dim > 1) {
        float *targets[3] = {pf->velocity_x, pf->velocity_y, pf->velocity_z};
        for (int axis = s->keep_3d_velocities ? 0 : 2; axis < 3; axis++) {
            if (!targets[axis]) continue;
            TRY(c21hip_velocity_kspace(saved, grid, box_dim[0], box_dim[1], box_dim[2], s->box_len,
                                       s->box_len_z, axis, s->dDdt_over_D, stream));
            if (hires && s->dim != s->hii_dim)
                TRY(c21hip_copy_filter(grid, grid, box_dim[0], box_dim[1], box_dim[2], s->box_len,
                                       s->box_len_z, 0,
                                       (float)(L_FACTOR * s->box_len / (s->hii_dim + 0.0)), 0.f, 1,
                                       stream));
            TRY(c21hip_fft_c2r(grid, box_dim[0], box_dim[1], box_dim[2], stream));
            float *d_out = targets[axis];
            const int host_out = !c21hip_is_device_ptr(targets[axis]);
            if (host_out) d_out = (float *)c21hip_ws(WS_PT_OUT0 + 1 + axis, lo_tot * sizeof(float));
            if (!d_out) return C21CM_MEMORY_ALLOC_ERROR;
            TRY(c21hip_gather(grid, box_dim, d_out, lo_dim, 0, 0.f, 0, stream));
            if (host_out) TRY(c21hip_d2h(targets[axis], d_out, lo_tot * sizeof(float), stream));
        }
    }
    TRY(c21hip_sync(stream));
done:
    return status;
}
