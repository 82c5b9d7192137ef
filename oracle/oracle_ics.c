/*
 * oracle_ics.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The grid algorithm of ComputeInitialConditions with explicit scalars.
 * reference: src/py21cmfast/src/InitialConditions.c
 *   :26-101   adj_complex_conj            (Hermitian symmetry of the k_z = 0 / Nyquist planes)
 *   :103-139  sample_ic_modes             (delta_k = sqrt(V P(k)/2) (a + i b))
 *   :240-267  compute_f_gradient          (i k_a / k^2)
 *   :269-297  compute_f_laplacian         (-k_a k_b / k^2)
 *   :299-364  compute_velocity_fields
 *   :366-545  compute_velocity_fields_2LPT
 *   :547-772  ComputeInitialConditions
 *
 * RNG: the reference draws (a, b) from per-OpenMP-thread GSL generators
 * (src/py21cmfast/src/rng.c:31-90), i.e. its realisation depends on N_THREADS.  Two streams:
 *   rng_stream = C21CM_RNG_GSL     that stream, restated in oracle_gslrng.c
 *                                  and pinned by the reference's HDF5 fixtures
 *                                  (tests/test_reference_fixtures.py);
 *   rng_stream = C21CM_RNG_PHILOX  a counter-based Philox-4x32-10 keyed by the seed with the
 *                                  mode index as counter + Box-Muller in double (the device's
 *                                  fast generator; a different realisation for the same seed).
 * Everything downstream of delta_k is also pinned through the `density_is_input` path
 * (reference test: tests/test_initial_conditions.py:153-178).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define L_FACTOR 0.620350491

/* ---------------- Philox-4x32-10 (Salmon et al. 2011), counter = mode index ------------- */
static inline void philox4x32_10(uint64_t counter, uint64_t key, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_gaussian_pair(uint64_t counter, uint64_t seed, double *a, double *b) {
    uint32_t x[4];
    philox4x32_10(counter, seed, x);
    /* 53-bit uniforms in (0, 1) */
    const double u1 = ((double)(((uint64_t)(x[0] >> 5) << 26) | (x[2] >> 6)) + 0.5) * 0x1p-53;
    const double u2 = ((double)(((uint64_t)(x[1] >> 5) << 26) | (x[3] >> 6)) + 0.5) * 0x1p-53;
    const double r = sqrt(-2.0 * log(u1));
    *a = r * cos(2.0 * M_PI * u2);
    *b = r * sin(2.0 * M_PI * u2);
}

static inline double index_to_k(int idx, double len, int dim) {
    double buf = (idx <= dim / 2) ? idx : (idx - dim);
    return buf * 2. * M_PI / len;
}

/* The element whose value (conjugated) a Hermitian-constrained element copies, as
 * adj_complex_conj assigns them (:58-100).  Returns 0 when (i,j) is a free element. */
static inline int hermitian_source(int i, int j, int nx, int ny, int *si, int *sj) {
    const int mx = nx / 2, my = ny / 2;
    if (i >= 1 && i < mx) { /* "do entire i except corners" */
        if (j == 0 || j == my) { *si = nx - i; *sj = j; return 1; }
        if (j >= 1 && j < my) { *si = nx - i; *sj = ny - j; return 1; }
        /* j > my: element (i, ny-j') with j' = ny-j in [1,my): = conj(nx-i, j') */
        *si = nx - i; *sj = ny - j; return 1;
    }
    if ((i == 0 || i == mx) && j >= 1 && j < my) { *si = i; *sj = ny - j; return 1; }
    return 0;
}

/* :103-139 + :26-101 */
static void sample_modes(const c21cm_ics_spec *s, float *cbox) {
    const int nx = s->dim, ny = s->dim, nz = s->dim_z, nzc = nz / 2 + 1;
    const int mx = nx / 2, my = ny / 2, mz = nz / 2;
#pragma omp parallel for schedule(static)
    for (int n_x = 0; n_x < nx; n_x++)
        for (int n_y = 0; n_y < ny; n_y++)
            for (int n_z = 0; n_z < nzc; n_z++) {
                int gi = n_x, gj = n_y, conj = 0;
                if (n_z == 0 || n_z == mz) {
                    int si, sj;
                    if (hermitian_source(n_x, n_y, nx, ny, &si, &sj)) {
                        gi = si; gj = sj; conj = 1;
                    }
                }
                const int ax = gi <= mx ? gi : nx - gi, ay = gj <= my ? gj : ny - gj;
                const long m = (long)ax * ax + (long)ay * ay + (long)n_z * n_z;
                const double p = s->pk_by_m[m];
                const uint64_t counter = ((uint64_t)gi * ny + gj) * nzc + n_z;
                double a, b;
                oracle_gaussian_pair(counter, s->seed, &a, &b);
                const double amp = sqrt(s->volume * p / 2.0);
                float re = (float)(amp * a), im = (float)(amp * b);
                if (conj) im = -im;
                /* the 7 self-conjugate corner modes are real, the DC mode is zero (:46-50) */
                const int cx = (n_x == 0 || n_x == mx), cy = (n_y == 0 || n_y == my),
                          cz = (n_z == 0 || n_z == mz);
                if (cx && cy && cz) im = 0.f;
                if (n_x == 0 && n_y == 0 && n_z == 0) re = 0.f;
                float *cell = cbox + 2 * (((size_t)n_x * ny + n_y) * nzc + n_z);
                cell[0] = re;
                cell[1] = im;
            }
}

/* compute_f_gradient :240-267 (axis1 < 0) and compute_f_laplacian :269-297 */
static void kspace_op(const c21cm_ics_spec *s, const float *in, float *out, int axis0, int axis1) {
    const int nx = s->dim, ny = s->dim, nz = s->dim_z, nzc = nz / 2 + 1;
    const double len[3] = {s->box_len, s->box_len, s->box_len_z};
#pragma omp parallel for schedule(static)
    for (int n_x = 0; n_x < nx; n_x++) {
        const double k_x = index_to_k(n_x, len[0], nx);
        for (int n_y = 0; n_y < ny; n_y++) {
            const double k_y = index_to_k(n_y, len[1], ny);
            for (int n_z = 0; n_z < nzc; n_z++) {
                const double k_z = index_to_k(n_z, len[2], nz);
                const double k_sq = k_x * k_x + k_y * k_y + k_z * k_z;
                const double kvec[3] = {k_x, k_y, k_z};
                const size_t idx = 2 * (((size_t)n_x * ny + n_y) * nzc + n_z);
                const double re = in[idx], im = in[idx + 1];
                if (n_x == 0 && n_y == 0 && n_z == 0) {
                    out[idx] = 0.f;
                    out[idx + 1] = 0.f;
                } else if (axis1 < 0) { /* in * k * I / k_sq */
                    out[idx] = (float)(-(im * kvec[axis0]) / k_sq);
                    out[idx + 1] = (float)((re * kvec[axis0]) / k_sq);
                } else { /* -k0 * k1 * in / k_sq */
                    const double f = -kvec[axis0] * kvec[axis1];
                    out[idx] = (float)(f * re / k_sq);
                    out[idx + 1] = (float)(f * im / k_sq);
                }
            }
        }
    }
}

static void subsample(const c21cm_ics_spec *s, const float *hi_padded, float *dst,
                      const int pt_dim[3], float divisor) {
    const int hi_dim[3] = {s->dim, s->dim, s->dim_z};
    const double ratio = hi_dim[0] / (double)pt_dim[0];
    const size_t zpad = 2 * (size_t)(hi_dim[2] / 2 + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < pt_dim[0]; i++)
        for (int j = 0; j < pt_dim[1]; j++)
            for (int k = 0; k < pt_dim[2]; k++) {
                const int hi = (int)(i * ratio + 0.5), hj = (int)(j * ratio + 0.5),
                          hk = (int)(k * ratio + 0.5);
                float v = hi_padded[(size_t)hk + zpad * ((size_t)hj + (size_t)hi_dim[1] * hi)];
                if (divisor != 0.f) v = v / divisor;
                dst[(size_t)k + (size_t)pt_dim[2] * ((size_t)j + (size_t)pt_dim[1] * i)] = v;
            }
}

int oracle_ics_grids(const c21cm_ics_spec *s, InitialConditions *ics) {
    const int hi_dim[3] = {s->dim, s->dim, s->dim_z};
    const int lo_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const int *pt_dim = s->perturb_on_high_res ? hi_dim : lo_dim;
    const size_t zpad = 2 * (size_t)(hi_dim[2] / 2 + 1);
    const size_t npad = (size_t)hi_dim[0] * hi_dim[1] * zpad;
    const size_t ntot = (size_t)hi_dim[0] * hi_dim[1] * hi_dim[2];
    const float VOLUME = s->volume;
    const float R_lo = (float)(L_FACTOR * s->box_len / (s->hii_dim + 0.0));
    const int need_filter = (s->dim != s->hii_dim);

    float *box = (float *)calloc(npad, sizeof(float));
    float *saved = (float *)calloc(npad, sizeof(float));
    if (!box || !saved) return C21CM_MEMORY_ALLOC_ERROR;

    if (s->density_is_input) { /* :636-663 */
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)hi_dim[0] * hi_dim[1]; l++)
            for (int k = 0; k < hi_dim[2]; k++)
                box[(size_t)l * zpad + k] =
                    ics->hires_density[(size_t)l * hi_dim[2] + k] * VOLUME / ntot;
        oracle_fft_r2c(box, hi_dim[0], hi_dim[1], hi_dim[2]);
        memcpy(saved, box, sizeof(float) * npad);
    } else { /* :664-692 */
        if (!s->pk_by_m || s->dim != s->dim_z || s->box_len != s->box_len_z) {
            free(box);
            free(saved);
            return C21CM_VALUE_ERROR;
        }
        if (s->rng_stream == C21CM_RNG_GSL) {
            int gst = oracle_gsl_sample_modes(s, s->rng_threads > 0 ? s->rng_threads : 1, box);
            if (gst) {
                free(box);
                free(saved);
                return gst;
            }
        } else {
            sample_modes(s, box);
        }
        memcpy(saved, box, sizeof(float) * npad);
        oracle_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2]);
        subsample(s, box, ics->hires_density, hi_dim, VOLUME);
    }
    /* low-res density :694-730 */
    memcpy(box, saved, sizeof(float) * npad);
    if (need_filter)
        oracle_filter_box(box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z, 0, R_lo,
                          0.f);
    oracle_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2]);
    subsample(s, box, ics->lowres_density, lo_dim, VOLUME);

    /* compute_relative_velocities :141-238 (V_CB_MODEL = FLUCTS), called at :733 */
    if (s->vcb_by_m) {
        if (!ics->lowres_vcb || s->dim != s->dim_z) {
            free(box);
            free(saved);
            return C21CM_VALUE_ERROR;
        }
        const int nx = hi_dim[0], ny = hi_dim[1], nz = hi_dim[2], nzc = nz / 2 + 1;
        const size_t nlo = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
        const double ratio = hi_dim[0] / (double)lo_dim[0];
        for (size_t t = 0; t < nlo; t++) ics->lowres_vcb[t] = 0.f; /* the caller's zeros */
        for (int ii = 0; ii < 3; ii++) {
#pragma omp parallel for schedule(static)
            for (int n_x = 0; n_x < nx; n_x++) {
                const double k_x = index_to_k(n_x, s->box_len, nx);
                for (int n_y = 0; n_y < ny; n_y++) {
                    const double k_y = index_to_k(n_y, s->box_len, ny);
                    for (int n_z = 0; n_z < nzc; n_z++) {
                        const double k_z = index_to_k(n_z, s->box_len_z, nz);
                        const double kvec[3] = {k_x, k_y, k_z};
                        const int ax = n_x <= nx / 2 ? n_x : nx - n_x, ay = n_y <= ny / 2 ? n_y : ny - n_y;
                        const long m = (long)ax * ax + (long)ay * ay + (long)n_z * n_z;
                        const size_t idx = 2 * (((size_t)n_x * ny + n_y) * nzc + n_z);
                        if (n_x == 0 && n_y == 0 && n_z == 0) {
                            box[0] = box[1] = 0.f;
                        } else { /* saved * I * k_a / |k| * sqrt(p_vcb / p) * c_kms */
                            const double f = kvec[ii] * s->vcb_by_m[m];
                            box[idx] = (float)(-((double)saved[idx + 1] * f));
                            box[idx + 1] = (float)((double)saved[idx] * f);
                        }
                    }
                }
            }
            if (need_filter)
                oracle_filter_box(box, nx, ny, nz, s->box_len, s->box_len_z, 0, R_lo, 0.f);
            oracle_fft_c2r(box, nx, ny, nz);
#pragma omp parallel for schedule(static)
            for (int i = 0; i < lo_dim[0]; i++)
                for (int j = 0; j < lo_dim[1]; j++)
                    for (int k = 0; k < lo_dim[2]; k++) {
                        const int hi = (int)(i * ratio + 0.5), hj = (int)(j * ratio + 0.5),
                                  hk = (int)(k * ratio + 0.5);
                        const double vcb_i = box[(size_t)hk + zpad * ((size_t)hj + (size_t)ny * hi)];
                        ics->lowres_vcb[(size_t)k + (size_t)lo_dim[2] * ((size_t)j + (size_t)lo_dim[1] * i)] +=
                            vcb_i * vcb_i;
                    }
        }
        for (size_t t = 0; t < nlo; t++) ics->lowres_vcb[t] = sqrt(ics->lowres_vcb[t]) / VOLUME;
    }

    /* first-order velocities :299-364 */
    float *vel[3], *vel2[3];
    if (s->perturb_on_high_res) {
        vel[0] = ics->hires_vx; vel[1] = ics->hires_vy; vel[2] = ics->hires_vz;
        vel2[0] = ics->hires_vx_2LPT; vel2[1] = ics->hires_vy_2LPT; vel2[2] = ics->hires_vz_2LPT;
    } else {
        vel[0] = ics->lowres_vx; vel[1] = ics->lowres_vy; vel[2] = ics->lowres_vz;
        vel2[0] = ics->lowres_vx_2LPT; vel2[1] = ics->lowres_vy_2LPT; vel2[2] = ics->lowres_vz_2LPT;
    }
    for (int ii = 0; ii < 3; ii++) {
        kspace_op(s, saved, box, ii, -1);
        if (!s->perturb_on_high_res && need_filter)
            oracle_filter_box(box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z, 0,
                              R_lo, 0.f);
        oracle_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2]);
        subsample(s, box, vel[ii], pt_dim, VOLUME);
    }

    if (s->perturb_algorithm == C21CM_PERTURB_2LPT) { /* :366-545 */
        float *phi_1 = (float *)calloc(npad, sizeof(float));
        float *diag[3];
        for (int c = 0; c < 3; c++) diag[c] = (float *)malloc(sizeof(float) * ntot);
        memset(box, 0, sizeof(float) * npad);
        for (int c = 0; c < 3; c++) {
            kspace_op(s, saved, phi_1, c, c);
            oracle_fft_c2r(phi_1, hi_dim[0], hi_dim[1], hi_dim[2]);
            subsample(s, phi_1, diag[c], hi_dim, 0.f);
        }
        static const int dirs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int c = 0; c < 3; c++) {
            const int pi = dirs[c][0], pj = dirs[c][1];
            kspace_op(s, saved, phi_1, pi, pj);
            oracle_fft_c2r(phi_1, hi_dim[0], hi_dim[1], hi_dim[2]);
#pragma omp parallel for schedule(static)
            for (long l = 0; l < (long)hi_dim[0] * hi_dim[1]; l++)
                for (int k = 0; k < hi_dim[2]; k++) {
                    const size_t ir = (size_t)l * hi_dim[2] + k, iff = (size_t)l * zpad + k;
                    const double cii = diag[pi][ir], cjj = diag[pj][ir], cij = phi_1[iff];
                    box[iff] += (cii * cjj);
                    box[iff] -= (cij * cij);
                }
        }
        const float norm = VOLUME * VOLUME * ntot; /* float * float * (ull -> float) */
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)hi_dim[0] * hi_dim[1]; l++)
            for (int k = 0; k < hi_dim[2]; k++) box[(size_t)l * zpad + k] /= norm;
        oracle_fft_r2c(box, hi_dim[0], hi_dim[1], hi_dim[2]);
        memcpy(saved, box, sizeof(float) * npad);
        for (int ii = 0; ii < 3; ii++) {
            kspace_op(s, saved, box, ii, -1);
            if (!s->perturb_on_high_res && need_filter)
                oracle_filter_box(box, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z,
                                  0, R_lo, 0.f);
            oracle_fft_c2r(box, hi_dim[0], hi_dim[1], hi_dim[2]);
            subsample(s, box, vel2[ii], pt_dim, 0.f);
        }
        free(phi_1);
        for (int c = 0; c < 3; c++) free(diag[c]);
    }
    free(box);
    free(saved);
    return C21CM_OK;
}
