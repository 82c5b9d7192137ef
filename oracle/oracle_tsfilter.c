/*
 * oracle_tsfilter.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The two filter loops of the spin-temperature calculation.
 * reference: src/py21cmfast/src/SpinTemperatureBox.c:502-530 (prepare_filter_boxes, density
 *            part), :560-636 (fill_Rbox_table), :642-742 (one_annular_filter).
 * Same loop structure as the reference: padded in-place boxes, a fresh copy of the unfiltered
 * spectrum per radius, double accumulators for the statistics.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static float *padded_from_dense(const float *input, int nx, int ny, int nz, double *sum_out) {
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    float *buf = (float *)calloc((size_t)nx * ny * zpad, sizeof(float));
    if (!buf) return NULL;
    double sum = 0.;
#pragma omp parallel for schedule(static) reduction(+ : sum)
    for (long l = 0; l < (long)nx * ny; l++)
        for (int k = 0; k < nz; k++) {
            const float v = input[(size_t)l * nz + k];
            buf[(size_t)l * zpad + k] = v;
            sum += v;
        }
    if (sum_out) *sum_out = sum;
    return buf;
}

/* r2c then division of the float complex cells by (float)N (:518-520, :684-690) */
static void forward_normalised(float *buf, int nx, int ny, int nz) {
    const size_t nk2 = (size_t)nx * ny * 2 * (size_t)(nz / 2 + 1);
    const float ntot = (float)((double)nx * ny * nz);
    oracle_fft_r2c(buf, nx, ny, nz);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)nk2; i++) buf[i] /= ntot;
}

int oracle_fill_Rbox_grids(const c21cm_rbox_spec *s, const float *input, float *result,
                           double *min_arr, double *average_arr, double *max_arr) {
    const int nx = s->hii_dim, ny = s->hii_dim, nz = s->hii_dim_z;
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    const size_t npad = (size_t)nx * ny * zpad, ntot = (size_t)nx * ny * nz;
    float *unf = padded_from_dense(input, nx, ny, nz, NULL);
    float *box = (float *)malloc(sizeof(float) * npad);
    if (!unf || !box) {
        free(unf);
        free(box);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    forward_normalised(unf, nx, ny, nz);
    for (int R_ct = 0; R_ct < s->n_R; R_ct++) {
        const double R = s->R[R_ct];
        double ave = 0, lo = 1e20, hi = -1e20; /* :577-579 */
        memcpy(box, unf, sizeof(float) * npad);
        if (R > s->cell_radius) { /* :585-588 */
            int st = oracle_filter_box(box, nx, ny, nz, s->box_len, s->box_len_z, s->filter_type,
                                       (float)R, 0.f);
            if (st) {
                free(unf);
                free(box);
                return st;
            }
        }
        oracle_fft_c2r(box, nx, ny, nz);
        float *out = result + (size_t)R_ct * ntot;
#pragma omp parallel for schedule(static) reduction(+ : ave) reduction(max : hi) reduction(min : lo)
        for (long l = 0; l < (long)nx * ny; l++)
            for (int k = 0; k < nz; k++) {
                float curr = box[(size_t)l * zpad + k];
                if (curr < s->min_value) curr = s->min_value; /* before the factor, :617-620 */
                curr = curr * s->const_factor;
                ave += curr;
                if (curr < lo) lo = curr;
                if (curr > hi) hi = curr;
                out[(size_t)l * nz + k] = curr;
            }
        if (average_arr) average_arr[R_ct] = ave / (double)ntot;
        if (min_arr) min_arr[R_ct] = lo;
        if (max_arr) max_arr[R_ct] = hi;
    }
    free(unf);
    free(box);
    return C21CM_OK;
}

int oracle_annular_filter_grids(const c21cm_annular_spec *s, const float *const *inputs,
                                float *const *outputs, double *u_avg, double *f_avg) {
    const int nx = s->hii_dim, ny = s->hii_dim, nz = s->hii_dim_z;
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    const size_t ntot = (size_t)nx * ny * nz;
    for (int g = 0; g < s->n_grids; g++) {
        double unfiltered_sum = 0., filtered_sum = 0.;
        float *box = padded_from_dense(inputs[g], nx, ny, nz, &unfiltered_sum);
        if (!box) return C21CM_MEMORY_ALLOC_ERROR;
        forward_normalised(box, nx, ny, nz);
        if (s->R_inner > 0) { /* :698-700 */
            int st = oracle_filter_box_star(box, nx, ny, nz, s->box_len, s->box_len_z,
                                            s->filter_type[g], (float)s->R_inner,
                                            (float)s->R_outer, (float)s->R_star);
            if (st) {
                free(box);
                return st;
            }
        }
        oracle_fft_c2r(box, nx, ny, nz);
        float *out = outputs[g];
#pragma omp parallel for schedule(static) reduction(+ : filtered_sum)
        for (long l = 0; l < (long)nx * ny; l++)
            for (int k = 0; k < nz; k++) {
                float v = box[(size_t)l * zpad + k];
                if (v < 0.) v = 0.; /* aliasing, :726 */
                out[(size_t)l * nz + k] = v;
                filtered_sum += v;
            }
        if (u_avg) u_avg[g] = unfiltered_sum / (double)ntot;
        if (f_avg) f_avg[g] = filtered_sum / (double)ntot;
        free(box);
    }
    return C21CM_OK;
}
