/*
 * oracle_filter.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * k-space window functions and the filter sweep.
 * reference: src/py21cmfast/src/filtering.c:18-32 (top-hat, sharp-k, Gaussian),
 *            :80-104 (exponential-MFP top-hat), :106-117 (spherical shell),
 *            :308-394 (filter_box), :397-445 (test_filter).
 * The multiple-scattering window (type 5, :119-306) belongs to the spin
 * temperature path and is out of scope.
 *
 * Precision notes that matter for parity: the reference holds k_x, k_y, k_z,
 * |k|^2 and (for types 0-2) kR in `float`, evaluates the window in `double`
 * and multiplies the float complex cell by the double window.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

void oracle_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

/* filtering.c:18-22 */
static double w_tophat(double kR) {
    if (kR < 1e-4) return 1 - kR * kR / 10;
    return 3.0 * pow(kR, -3) * (sin(kR) - cos(kR) * kR);
}

/* filtering.c:26-30 */
static double w_sharpk(double kR) {
    if (kR * 0.413566994 > 1) return 0.;
    return 1;
}

/* filtering.c:32 */
static double w_gauss(double kR_squared) { return exp(-0.643 * 0.643 * kR_squared / 2.); }

/* filtering.c:80-104 */
static double w_exp_mfp(double k, double R, double mfp, double exp_term) {
    double kR = k * R;
    double ratio = mfp / R;
    if (kR < 1e-4) {
        double ts_0 =
            6 * pow(ratio, 3) - exp_term * (6 * pow(ratio, 3) + 6 * pow(ratio, 2) + 3 * ratio);
        return ts_0 +
               (exp_term * (2 * pow(ratio, 2) + 0.5 * ratio) - 2 * ts_0 * pow(ratio, 2)) * kR * kR;
    }
    double f = (kR * kR * pow(ratio, 2) + 2 * ratio + 1) * ratio * cos(kR);
    f += (kR * kR * (pow(ratio, 2) - pow(ratio, 3)) + ratio + 1) * sin(kR) / kR;
    f *= exp_term;
    f -= 2 * pow(ratio, 2);
    f *= -3 * ratio / pow(pow(kR * ratio, 2) + 1, 2);
    return f;
}

/* filtering.c:106-117 */
static double w_shell(double k, double R_inner, double R_outer) {
    double kR_inner = k * R_inner;
    double kR_outer = k * R_outer;
    if (kR_outer < 1e-4)
        return 1. - kR_outer * kR_outer / 10 * (pow(R_inner / R_outer, 5) - 1) /
                        (pow(R_inner / R_outer, 3) - 1);
    return 3.0 / (pow(kR_outer, 3) - pow(kR_inner, 3)) *
           (sin(kR_outer) - cos(kR_outer) * kR_outer - sin(kR_inner) + cos(kR_inner) * kR_inner);
}

/* Window value for a mode of magnitude k (exposed for the analytic tests). */
double oracle_filter_window(int filter_type, double k, float R, float R_param) {
    float kmag_sq = (float)(k * k);
    float kR;
    switch (filter_type) {
        case 0:
            kR = sqrt(kmag_sq) * R;
            return w_tophat(kR);
        case 1:
            kR = sqrt(kmag_sq) * R;
            return w_sharpk(kR);
        case 2:
            kR = kmag_sq * R * R;
            return w_gauss(kR);
        case 3:
            return w_exp_mfp(sqrt(kmag_sq), R, R_param, exp(-R / R_param));
        case 4:
            return w_shell(sqrt(kmag_sq), R, R_param);
        default:
            return NAN;
    }
}

/* filtering.c:308-394.  cbox is the half-spectrum float[nx][ny][nz/2+1][2]. */
int oracle_filter_box(float *cbox, int nx, int ny, int nz, double box_len, double box_len_z,
                      int filter_type, float R, float R_param) {
    if (filter_type < 0 || filter_type > 4) return C21CM_VALUE_ERROR;
    const double delta_k[3] = {2.0 * M_PI / box_len, 2.0 * M_PI / box_len,
                               2.0 * M_PI / box_len_z};
    double R_const = 0.;
    if (filter_type == 3) R_const = exp(-R / R_param);
    const int nzc = nz / 2 + 1;

#pragma omp parallel for schedule(static)
    for (int n_x = 0; n_x < nx; n_x++) {
        float k_x, k_y, k_z, k_mag_sq, kR;
        if (n_x > nx / 2)
            k_x = (n_x - nx) * delta_k[0];
        else
            k_x = n_x * delta_k[0];
        for (int n_y = 0; n_y < ny; n_y++) {
            if (n_y > ny / 2)
                k_y = (n_y - ny) * delta_k[1];
            else
                k_y = n_y * delta_k[1];
            float *line = cbox + 2 * (((size_t)n_x * ny + n_y) * nzc);
            for (int n_z = 0; n_z < nzc; n_z++) {
                k_z = n_z * delta_k[2];
                k_mag_sq = k_x * k_x + k_y * k_y + k_z * k_z;
                double w;
                if (filter_type == 0) {
                    kR = sqrt(k_mag_sq) * R;
                    w = w_tophat(kR);
                } else if (filter_type == 1) {
                    kR = sqrt(k_mag_sq) * R;
                    w = w_sharpk(kR);
                } else if (filter_type == 2) {
                    kR = k_mag_sq * R * R;
                    w = w_gauss(kR);
                } else if (filter_type == 3) {
                    w = w_exp_mfp(sqrt(k_mag_sq), R, R_param, R_const);
                } else {
                    w = w_shell(sqrt(k_mag_sq), R, R_param);
                }
                line[2 * n_z] = (float)(line[2 * n_z] * w);
                line[2 * n_z + 1] = (float)(line[2 * n_z + 1] * w);
            }
        }
    }
    return C21CM_OK;
}

/* r2c, /N, window, c2r: dense float in, dense float out (filtering.c:397-445
 * without the final widening to double). */
int oracle_filter_grid(const float *input, float *output, int nx, int ny, int nz, double box_len,
                       double box_len_z, int filter_type, double R, double R_param) {
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    const size_t npad = (size_t)nx * ny * zpad;
    const size_t nk = npad / 2;
    const double ntot = (double)nx * ny * nz;
    float *buf = (float *)malloc(sizeof(float) * npad);
    if (!buf) return C21CM_MEMORY_ALLOC_ERROR;
    memset(buf, 0, sizeof(float) * npad);
#pragma omp parallel for schedule(static)
    for (long l = 0; l < (long)nx * ny; l++)
        memcpy(buf + (size_t)l * zpad, input + (size_t)l * nz, sizeof(float) * nz);
    oracle_fft_r2c(buf, nx, ny, nz);
    /* test_filter divides the float complex by (double)N: filtering.c:422-424 */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)(2 * nk); i++) buf[i] = (float)(buf[i] / ntot);
    int st = oracle_filter_box(buf, nx, ny, nz, box_len, box_len_z, filter_type, (float)R,
                               (float)R_param);
    if (st) {
        free(buf);
        return st;
    }
    oracle_fft_c2r(buf, nx, ny, nz);
#pragma omp parallel for schedule(static)
    for (long l = 0; l < (long)nx * ny; l++)
        memcpy(output + (size_t)l * nz, buf + (size_t)l * zpad, sizeof(float) * nz);
    free(buf);
    return C21CM_OK;
}

/* filtering.c:397-445: same, result widened to double like the exported test hook */
int oracle_test_filter(const float *input, int nx, int ny, int nz, double box_len,
                       double box_len_z, double R, double R_param, int filter_type,
                       double *result) {
    const size_t n = (size_t)nx * ny * nz;
    float *tmp = (float *)malloc(sizeof(float) * n);
    if (!tmp) return C21CM_MEMORY_ALLOC_ERROR;
    int st =
        oracle_filter_grid(input, tmp, nx, ny, nz, box_len, box_len_z, filter_type, R, R_param);
    if (!st)
        for (size_t i = 0; i < n; i++) result[i] = tmp[i];
    free(tmp);
    return st;
}
