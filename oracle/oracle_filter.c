/*
 * oracle_filter.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * k-space window functions and the filter sweep.
 * reference: src/py21cmfast/src/filtering.c:18-32 (top-hat, sharp-k, Gaussian),
 *            :80-104 (exponential-MFP top-hat), :106-117 (spherical shell),
 *            :308-394 (filter_box), :397-445 (test_filter).
 *            :119-306 (multiple-scattering window of the Lyman-alpha shells, type 5;
 *            GSL's reciprocal gamma function is restated with tgamma).
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


/* ---- multiple-scattering window, filtering.c:119-306 (arXiv:2601.14360) ---------------- */
typedef struct {
    double alpha_outer, beta_outer, alpha_inner, beta_inner;
} ms_params;

/* filtering.c:126-142 */
static double ms_mu(double x_em) {
    double zeta_em = log10(x_em);
    if (x_em > 30) return 1. - 1.0478 * pow(x_em, -0.7266);
    if (x_em > 3.)
        return -0.104 * pow(zeta_em, 5) + 0.4867 * pow(zeta_em, 4) - 0.8217 * pow(zeta_em, 3) +
               0.4889 * zeta_em * zeta_em + 0.264 * zeta_em + 0.518;
    if (x_em > 0.2)
        return -0.0285 * pow(zeta_em, 5) + 0.087 * pow(zeta_em, 4) - 0.1205 * pow(zeta_em, 3) -
               0.0456 * zeta_em * zeta_em + 0.3787 * zeta_em + 0.5285;
    return 0.3982 * pow(x_em, 0.1592);
}

/* filtering.c:144-160 */
static double ms_eta(double x_em) {
    double zeta_em = log10(x_em);
    if (x_em > 20.) return 1. - 2.804 * pow(x_em, -1.242);
    if (x_em > 3.)
        return 2.17 * pow(zeta_em, 5) - 8.832 * pow(zeta_em, 4) + 13.579 * pow(zeta_em, 3) -
               10.04 * zeta_em * zeta_em + 4.166 * zeta_em - 0.17;
    if (x_em > 0.2)
        return 0.352 * pow(zeta_em, 5) - 0.0516 * pow(zeta_em, 4) - 0.293 * pow(zeta_em, 3) +
               0.342 * zeta_em * zeta_em + 0.582 * zeta_em + 0.266;
    return 0.4453 * pow(x_em, 1.296);
}

/* filtering.c:162-187 */
static void ms_init(double R_inner, double R_outer, double R_star, ms_params *c) {
    if (R_star == 0.) {
        c->alpha_inner = 1.;
        c->alpha_outer = 1.;
        c->beta_inner = 1.;
        c->beta_outer = 0.;
        return;
    }
    double x_in = R_inner / R_star, x_out = R_outer / R_star;
    double mu_in = ms_mu(x_in), eta_in = ms_eta(x_in);
    double mu_out = ms_mu(x_out), eta_out = ms_eta(x_out);
    c->alpha_inner = (1. / eta_in - 1.) / pow(1. / mu_in - 1., 2);
    c->beta_inner = (1. / eta_in - 1.) / (1. / mu_in - 1.);
    c->alpha_outer = (1. / eta_out - 1.) / pow(1. / mu_out - 1., 2);
    c->beta_outer = (1. / eta_out - 1.) / (1. / mu_out - 1.);
}

/* 1 / Gamma(x), zero at the poles (what gsl_sf_gammainv returns) */
static double gamma_inv(double x) {
    if (x <= 0. && x == floor(x)) return 0.;
    return 1. / tgamma(x);
}

/* filtering.c:189-254 */
static double ms_asymptotic_2F3(double kR, double alpha, double beta) {
    double a1 = (2. + alpha) / 2., a2 = (3. + alpha) / 2., b1 = 5. / 2.;
    double b2 = (2. + alpha + beta) / 2., b3 = (3. + alpha + beta) / 2.;
    double gamma_a1 = tgamma(a1), gamma_a2 = tgamma(a2), gamma_b1 = 3. / 4.;
    double gamma_b2 = tgamma(b2), gamma_b3 = tgamma(b3);
    double g21, g32, d1, d2;
    if (a1 < 20.) {
        g21 = gamma_b2 / gamma_a1;
        g32 = gamma_b3 / gamma_a2;
    } else {
        double y = beta / 2;
        g21 = pow(a1, y) * exp((a1 + y - 0.5) * (y / a1 - y * y / (2. * a1 * a1) +
                                                 y * y * y / (3. * a1 * a1 * a1)) -
                               y);
        g32 = pow(a2, y) * exp((a2 + y - 0.5) * (y / a2 - y * y / (2. * a2 * a1) +
                                                 y * y * y / (3. * a2 * a2 * a2)) -
                               y);
    }
    if (alpha < 10.) {
        d1 = M_PI * gamma_a1 * gamma_inv(b1 - a1) / tgamma(b2 - a1) / tgamma(b3 - a1) /
             pow(kR / 2., alpha + 2.);
        d2 = -2. * M_PI * gamma_a2 * gamma_inv(b1 - a2) * gamma_inv(b2 - a2) / tgamma(b3 - a2) /
             pow(kR / 2., alpha + 3.);
    } else {
        d1 = 0.;
        d2 = 0.;
    }
    double F = (cos(kR - M_PI * (2. + beta) / 2.) -
                (1. + (alpha - 1.) * beta) / kR * sin(kR - M_PI * (2. + beta) / 2.)) /
               pow(kR / 2, beta + 2);
    F += d1 + d2;
    F *= gamma_b1 * g21 * g32;
    return F;
}

/* filtering.c:258-293 */
static double ms_hyper_2F3(double kR, double alpha, double beta) {
    if (beta == 0.) return 3.0 / (pow(kR, 3)) * (sin(kR) - cos(kR) * kR);
    if (kR < 30.) {
        double sum = 0., term = 1.;
        for (int n = 1; n < 1000; n++) {
            sum += term;
            term *= -1. / (1. + beta / (alpha + 2. * n)) / (1. + beta / (alpha + 1 + 2. * n)) * kR *
                    kR / (2. * n) / (2. * n + 3.);
            if (fabs(term) < fabs(sum) * 1e-4) break;
        }
        return sum;
    }
    double F_ms = ms_asymptotic_2F3(kR, alpha, beta);
    double F_sl = 3.0 / (pow(kR, 3)) * (sin(kR) - cos(kR) * kR);
    return (fabs(F_ms) < fabs(F_sl)) ? F_ms : F_sl;
}

/* filtering.c:295-306 */
static double w_multiple_scattering(double k, double R_inner, double R_outer, const ms_params *c) {
    double kR_inner = k * R_inner, kR_outer = k * R_outer;
    double W = pow(R_outer, 3.) * ms_hyper_2F3(kR_outer, c->alpha_outer, c->beta_outer) -
               pow(R_inner, 3.) * ms_hyper_2F3(kR_inner, c->alpha_inner, c->beta_inner);
    W /= pow(R_outer, 3.) - pow(R_inner, 3.);
    return W;
}

/* the three helpers the reference exports for its own tests (tests/test_filtering.py:369-396) */
double oracle_ms_mu(double x_em) { return ms_mu(x_em); }
double oracle_ms_eta(double x_em) { return ms_eta(x_em); }
double oracle_hyper_2F3(double kR, double alpha, double beta) { return ms_hyper_2F3(kR, alpha, beta); }

/* Window value for a mode of magnitude k (exposed for the analytic tests). */
double oracle_filter_window_ms(double k, float R_inner, float R_outer, float R_star) {
    ms_params c;
    ms_init(R_inner, R_outer, R_star, &c);
    float kmag_sq = (float)(k * k);
    return w_multiple_scattering(sqrt(kmag_sq), R_inner, R_outer, &c);
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
    return oracle_filter_box_star(cbox, nx, ny, nz, box_len, box_len_z, filter_type, R, R_param,
                                  0.f);
}

/* the full signature of filter_box (filtering.c:308): R_star only matters for type 5 */
int oracle_filter_box_star(float *cbox, int nx, int ny, int nz, double box_len, double box_len_z,
                           int filter_type, float R, float R_param, float R_star) {
    if (filter_type < 0 || filter_type > 5) return C21CM_VALUE_ERROR;
    ms_params ms = {0};
    if (filter_type == 5) ms_init(R, R_param, R_star, &ms);
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
                } else if (filter_type == 4) {
                    w = w_shell(sqrt(k_mag_sq), R, R_param);
                } else {
                    w = w_multiple_scattering(sqrt(k_mag_sq), R, R_param, &ms);
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
