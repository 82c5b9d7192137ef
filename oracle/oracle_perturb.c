/*
 * oracle_perturb.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The grid algorithm of ComputePerturbedField with explicit scalars.
 * reference: src/py21cmfast/src/PerturbedField.c
 *   :24-135   make_density_grid           (linear scaling, or move_grid_masses + widen)
 *   :137-178  assign_to_lowres_grid       (PERTURB_ON_HIGH_RES: r2c, top-hat, c2r, subsample)
 *   :180-210  normalise_delta_grid
 *   :212-282  smooth_and_clip_density
 *   :284-387  compute_perturbed_velocities
 *   :389-496  ComputePerturbedField
 * and src/py21cmfast/src/map_mass.c:23-60 (CIC deposit), :146-208 (move_grid_masses),
 *     src/py21cmfast/src/indexing.h:108-120 (resample_index, index_to_k).
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define FRACT_FLOAT_ERR ((double)1e-7)
#define L_FACTOR 0.620350491 /* Constants.c:41 */

static inline int wrap(int i, int n) {
    /* indexing.c:37-60 (while loops == mathematical modulo) */
    i %= n;
    if (i < 0) i += n;
    return i;
}

/* map_mass.c:23-60 */
static void cic_deposit(double *box, const double pos[3], const int dim[3], double mass) {
    int ipos[3], iposp1[3];
    double dist[3];
    for (int a = 0; a < 3; a++) {
        ipos[a] = (int)floor(pos[a]);
        iposp1[a] = ipos[a] + 1;
        dist[a] = pos[a] - ipos[a];
        ipos[a] = wrap(ipos[a], dim[a]);
        iposp1[a] = wrap(iposp1[a], dim[a]);
    }
    const int ix[2] = {ipos[0], iposp1[0]}, iy[2] = {ipos[1], iposp1[1]},
              iz[2] = {ipos[2], iposp1[2]};
    const double wx[2] = {1. - dist[0], dist[0]}, wy[2] = {1. - dist[1], dist[1]},
                 wz[2] = {1. - dist[2], dist[2]};
    /* same 8 targets and weight products as the reference (x fastest in its list) */
    for (int c = 0; c < 2; c++)
        for (int b = 0; b < 2; b++)
            for (int a = 0; a < 2; a++) {
                size_t idx = (size_t)iz[c] + (size_t)dim[2] * ((size_t)iy[b] + (size_t)dim[1] * ix[a]);
                double w = wx[a] * wy[b] * wz[c];
#pragma omp atomic update
                box[idx] += mass * w;
            }
}

/* map_mass.c:146-208 */
static void move_grid_masses(const c21cm_perturb_spec *s, const float *dens, const int dens_dim[3],
                             float *const vel[3], float *const vel2[3], const int vel_dim[3],
                             double *resampled, const int out_dim[3]) {
    const double box_size[3] = {s->box_len, s->box_len, s->box_len_z};
    const double dim_ratio_vel = (double)vel_dim[0] / (double)dens_dim[0];
    const double dim_ratio_out = (double)out_dim[0] / (double)dens_dim[0];
    const double gf = s->growth_factor, igf = s->init_growth_factor;
    const double d2 = -(3.0 / 7.0) * gf * gf, id2 = -(3.0 / 7.0) * igf * igf;
    double vdf[3], vdf2[3];
    for (int a = 0; a < 3; a++) {
        vdf[a] = (gf - igf) / box_size[a] * dens_dim[a];
        vdf2[a] = (d2 - id2) / box_size[a] * dens_dim[a];
    }
    const int lpt2 = (s->perturb_algorithm == C21CM_PERTURB_2LPT);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < dens_dim[0]; i++)
        for (int j = 0; j < dens_dim[1]; j++)
            for (int k = 0; k < dens_dim[2]; k++) {
                double pos[3] = {i, j, k};
                const int src[3] = {i, j, k};
                int ip[3];
                for (int a = 0; a < 3; a++)
                    ip[a] = wrap((int)(src[a] * dim_ratio_vel + 0.5), vel_dim[a]);
                const size_t vi =
                    (size_t)ip[2] + (size_t)vel_dim[2] * ((size_t)ip[1] + (size_t)vel_dim[1] * ip[0]);
                for (int a = 0; a < 3; a++) {
                    pos[a] += vel[a][vi] * vdf[a];
                    if (lpt2) pos[a] -= vel2[a][vi] * vdf2[a];
                    pos[a] *= dim_ratio_out;
                }
                const size_t di =
                    (size_t)k + (size_t)dens_dim[2] * ((size_t)j + (size_t)dens_dim[1] * i);
                const double curr_dens = 1.0 + dens[di] * igf;
                cic_deposit(resampled, pos, out_dim, curr_dens);
            }
}

/* indexing.h:116-120 */
static inline double index_to_k(int idx, double len, int dim) {
    double buf = (idx <= dim / 2) ? idx : (idx - dim);
    return buf * 2. * M_PI / len;
}

/* PerturbedField.c:284-387 */
static void perturbed_velocity(const c21cm_perturb_spec *s, int axis, const float *saved_c,
                               float *grid, const int box_dim[3], float *velocity) {
    const int nx = box_dim[0], ny = box_dim[1], nz = box_dim[2];
    const int nzc = nz / 2 + 1;
    const size_t nk = (size_t)nx * ny * nzc;
    const size_t n_r_pixels = (size_t)nx * ny * nz;
    const double box_len[3] = {s->box_len, s->box_len, s->box_len_z};
    const int lo_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const double dim_ratio = box_dim[0] / (double)lo_dim[0];
    memcpy(grid, saved_c, sizeof(float) * 2 * nk);
#pragma omp parallel for schedule(static)
    for (int n_x = 0; n_x < nx; n_x++) {
        float kvec[3];
        kvec[0] = index_to_k(n_x, box_len[0], nx);
        for (int n_y = 0; n_y < ny; n_y++) {
            kvec[1] = index_to_k(n_y, box_len[1], ny);
            for (int n_z = 0; n_z < nzc; n_z++) {
                kvec[2] = index_to_k(n_z, box_len[2], nz);
                float *cell = grid + 2 * (((size_t)n_x * ny + n_y) * nzc + n_z);
                const float k_sq = kvec[0] * kvec[0] + kvec[1] * kvec[1] + kvec[2] * kvec[2];
                if (n_x == 0 && n_y == 0 && n_z == 0) {
                    cell[0] = 0.f;
                    cell[1] = 0.f;
                } else {
                    /* cell *= dDdt_over_D * kvec[axis] * I / k_sq / n_r_pixels (complex double) */
                    const double c = s->dDdt_over_D * kvec[axis] / k_sq / n_r_pixels;
                    const double re = cell[0], im = cell[1];
                    cell[0] = (float)(-im * c);
                    cell[1] = (float)(re * c);
                }
            }
        }
    }
    if (s->perturb_on_high_res && s->dim != s->hii_dim)
        oracle_filter_box(grid, nx, ny, nz, s->box_len, s->box_len_z, 0,
                          (float)(L_FACTOR * s->box_len / (s->hii_dim + 0.0)), 0.f);
    oracle_fft_c2r(grid, nx, ny, nz);
    const size_t zpad = 2 * (size_t)nzc;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < lo_dim[0]; i++)
        for (int j = 0; j < lo_dim[1]; j++)
            for (int k = 0; k < lo_dim[2]; k++) {
                const int hi = (int)(i * dim_ratio + 0.5), hj = (int)(j * dim_ratio + 0.5),
                          hk = (int)(k * dim_ratio + 0.5);
                velocity[(size_t)k + (size_t)lo_dim[2] * ((size_t)j + (size_t)lo_dim[1] * i)] =
                    grid[(size_t)hk + zpad * ((size_t)hj + (size_t)ny * hi)];
            }
}

int oracle_perturb_grids(const c21cm_perturb_spec *s, const InitialConditions *ics,
                         PerturbedField *pf) {
    const int lo_dim[3] = {s->hii_dim, s->hii_dim, s->hii_dim_z};
    const int hi_dim[3] = {s->dim, s->dim, s->dim_z};
    const int hires = s->perturb_on_high_res;
    const int *box_dim = hires ? hi_dim : lo_dim;
    const size_t lo_zpad = 2 * (size_t)(lo_dim[2] / 2 + 1);
    const size_t lo_npad = (size_t)lo_dim[0] * lo_dim[1] * lo_zpad;
    const size_t lo_tot = (size_t)lo_dim[0] * lo_dim[1] * lo_dim[2];
    const size_t hi_zpad = 2 * (size_t)(hi_dim[2] / 2 + 1);
    const size_t hi_npad = (size_t)hi_dim[0] * hi_dim[1] * hi_zpad;
    const size_t hi_tot = (size_t)hi_dim[0] * hi_dim[1] * hi_dim[2];
    const size_t b_zpad = hires ? hi_zpad : lo_zpad;
    const size_t b_tot = hires ? hi_tot : lo_tot;

    float *vel[3], *vel2[3];
    const float *dens_box;
    if (hires) {
        vel[0] = ics->hires_vx; vel[1] = ics->hires_vy; vel[2] = ics->hires_vz;
        vel2[0] = ics->hires_vx_2LPT; vel2[1] = ics->hires_vy_2LPT; vel2[2] = ics->hires_vz_2LPT;
        dens_box = ics->hires_density;
    } else {
        vel[0] = ics->lowres_vx; vel[1] = ics->lowres_vy; vel[2] = ics->lowres_vz;
        vel2[0] = ics->lowres_vx_2LPT; vel2[1] = ics->lowres_vy_2LPT; vel2[2] = ics->lowres_vz_2LPT;
        dens_box = ics->lowres_density;
    }

    float *lowres = (float *)calloc(lo_npad, sizeof(float));
    float *hires_grid = hires ? (float *)calloc(hi_npad, sizeof(float)) : NULL;
    float *saved = (float *)calloc(hires ? hi_npad : lo_npad, sizeof(float));
    float *grid = hires ? hires_grid : lowres;

    /* make_density_grid: PerturbedField.c:24-135 */
    if (s->perturb_algorithm == C21CM_PERTURB_LINEAR) {
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)box_dim[0] * box_dim[1]; l++)
            for (int k = 0; k < box_dim[2]; k++)
                grid[(size_t)l * b_zpad + k] = s->growth_factor * dens_box[(size_t)l * box_dim[2] + k];
    } else {
        double *resampled = (double *)calloc(b_tot, sizeof(double));
        move_grid_masses(s, ics->hires_density, hi_dim, vel, vel2, box_dim, resampled, box_dim);
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)box_dim[0] * box_dim[1]; l++)
            for (int k = 0; k < box_dim[2]; k++)
                grid[(size_t)l * b_zpad + k] = resampled[(size_t)l * box_dim[2] + k];
        free(resampled);
    }

    if (hires) {
        /* assign_to_lowres_grid: PerturbedField.c:137-178 */
        const double dim_ratio = hi_dim[0] / (double)lo_dim[0];
        oracle_fft_r2c(hires_grid, hi_dim[0], hi_dim[1], hi_dim[2]);
        memcpy(saved, hires_grid, sizeof(float) * hi_npad);
        oracle_filter_box(hires_grid, hi_dim[0], hi_dim[1], hi_dim[2], s->box_len, s->box_len_z, 0,
                          (float)(L_FACTOR * s->box_len / (lo_dim[0] + 0.0)), 0.f);
        oracle_fft_c2r(hires_grid, hi_dim[0], hi_dim[1], hi_dim[2]);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < lo_dim[0]; i++)
            for (int j = 0; j < lo_dim[1]; j++)
                for (int k = 0; k < lo_dim[2]; k++) {
                    const int hi = (int)(i * dim_ratio + 0.5), hj = (int)(j * dim_ratio + 0.5),
                              hk = (int)(k * dim_ratio + 0.5);
                    lowres[(size_t)k + lo_zpad * ((size_t)j + (size_t)lo_dim[1] * i)] =
                        hires_grid[(size_t)hk + hi_zpad * ((size_t)hj + (size_t)hi_dim[1] * hi)] /
                        (float)hi_tot;
                }
    }
    if (s->perturb_algorithm > C21CM_PERTURB_LINEAR) {
        /* normalise_delta_grid: PerturbedField.c:180-210 */
        const double mass_factor = hires ? 1.0 : lo_tot / (double)hi_tot;
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)lo_dim[0] * lo_dim[1]; l++)
            for (int k = 0; k < lo_dim[2]; k++) {
                float *cell = lowres + (size_t)l * lo_zpad + k;
                *cell *= mass_factor;
                *cell -= 1;
            }
    }
    /* smooth_and_clip_density: PerturbedField.c:212-282 */
    oracle_fft_r2c(lowres, lo_dim[0], lo_dim[1], lo_dim[2]);
    if (s->smooth_evolved_density)
        oracle_filter_box(lowres, lo_dim[0], lo_dim[1], lo_dim[2], s->box_len, s->box_len_z, 2,
                          (float)s->density_smooth_radius_mpc, 0.f);
    if (!hires) memcpy(saved, lowres, sizeof(float) * lo_npad);
    oracle_fft_c2r(lowres, lo_dim[0], lo_dim[1], lo_dim[2]);
#pragma omp parallel for schedule(static)
    for (long l = 0; l < (long)lo_dim[0] * lo_dim[1]; l++)
        for (int k = 0; k < lo_dim[2]; k++) {
            float *cell = lowres + (size_t)l * lo_zpad + k;
            *cell /= (float)lo_tot;
            if (*cell < -1.0 + FRACT_FLOAT_ERR) *cell = -1.0 + FRACT_FLOAT_ERR;
            pf->density[(size_t)l * lo_dim[2] + k] = *cell; /* PerturbedField.c:450-464 */
        }
    /* velocities: PerturbedField.c:466-477 */
    if (s->hii_dim > 1) {
        if (s->keep_3d_velocities) {
            perturbed_velocity(s, 0, saved, grid, box_dim, pf->velocity_x);
            perturbed_velocity(s, 1, saved, grid, box_dim, pf->velocity_y);
        }
        perturbed_velocity(s, 2, saved, grid, box_dim, pf->velocity_z);
    }
    free(lowres);
    free(hires_grid);
    free(saved);
    return C21CM_OK;
}
