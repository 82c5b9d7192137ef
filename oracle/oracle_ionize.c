/*
 * oracle_ionize.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The grid algorithm of ComputeIonizedBox with every physics scalar supplied
 * through c21cm_ionize_spec.
 * reference: src/py21cmfast/src/IonisationBox.c
 *   :323-360   prepare_box_for_filtering  (scale, clip, pack, r2c, /N)
 *   :365-401   setup_first_z_prevbox      (previous z_reion := -1)
 *   :572-664   copy_filter_transform      (memcpy, filter_box, c2r per grid)
 *   :668-699   clip_and_get_extrema
 *   :702-768   setup_integration_tables   (table range = extrema -/+ 0.001)
 *   :773-962   calculate_fcoll_grid
 *   :1008-1201 find_ionised_regions
 *   :1203-1256 set_ionized_temperatures
 *   :1531-1628 the R loop, global sums, returned mean_f_coll
 * plus src/py21cmfast/src/thermochem.c:31-63 (temperatures),
 *      src/py21cmfast/src/hmf.c:1187-1241 (erfcc / FgtrM_bias_fast),
 *      src/py21cmfast/src/interpolation.c:123-131 (EvaluateRGTable1D_f).
 *
 *   :583-663   N_rec / whalo_sfr grids of the recombination models (filtered like delta / n_ion)
 *   :1084-1140 recombinations in the barrier, Gamma_12 and mean free path at first crossing
 *   :1258-1340 set_recombination_rates (+ recombinations.c:64-92 splined_recombination_rate)
 *
 *   :403-457   calculate_mcrit_boxes      (USE_MINI_HALOS: turnover-mass boxes)
 *   :595-603,715-761  previous delta + turnover grids filtered, 2-D table ranges
 *   :838-936   per-radius f_coll history of both populations; :1068-1158 two-population barrier
 *   :1150-1158 IONISE_ENTIRE_SPHERE with bubble_helper_progs.c:262-418 (update_in_sphere)
 *
 * PINS.  The one-population paths are pinned to the reference's fixtures (tests/golden/reference,
 * DESIGN.md section 3).  The USE_MINI_HALOS and IONISE_ENTIRE_SPHERE branches are PARITY
 * UNPINNED: the reference holds no vector for them that this image can reproduce (its `mini`
 * fixtures need CLASS transfer tables; none uses the sphere method).  They are tied to the pinned
 * path by construction tests only (tests/test_oracle_minihalos.py).
 * Not restated (returns C21CM_VALUE_ERROR): IONISE_ENTIRE_SPHERE with a recombination model or
 * mini-halos (thread-order dependent upstream).
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define FRACT_FLOAT_ERR ((double)1e-7) /* reference: Constants.h */
#define TINY ((double)1e-30)
#define MIN_DENSITY_LOW_LIMIT (9e-8) /* reference: thermochem.c:16 */

/* thermochem.c:31-56 */
float oracle_fully_ionized_temperature(float z_re, float z, float delta, float T_re) {
    float result, delta_re;
    if (fabs(z - z_re) < 1e-4)
        result = 1;
    else {
        delta_re = delta * (1. + z) / (1. + z_re);
        if (delta_re <= -1) delta_re = -1. + MIN_DENSITY_LOW_LIMIT;
        if (delta <= -1) delta = -1. + MIN_DENSITY_LOW_LIMIT;
        result = pow((1. + delta) / (1. + delta_re), 1.1333);
        result *= pow((1. + z) / (1. + z_re), 3.4);
        result *= expf(pow((1. + z) / 7.1, 2.5) - pow((1. + z_re) / 7.1, 2.5));
    }
    result *= pow(T_re, 1.7);
    result += pow(1e4 * ((1. + z) / 4.), 1.7) * (1 + delta);
    result = pow(result, 0.5882);
    return result;
}

/* thermochem.c:58-63 */
float oracle_partially_ionized_temperature(float T_HI, float res_xH, float T_re) {
    if (res_xH <= 0.) return T_re;
    if (res_xH >= 1) return T_HI;
    return T_HI * res_xH + T_re * (1. - res_xH);
}

/* hmf.c:1187-1203 */
static float erfcc(float x) {
    double t, q, ans;
    q = fabs(x);
    t = 1.0 / (1.0 + 0.5 * q);
    ans = t * exp(-q * q - 1.2655122 +
                  t * (1.0000237 +
                       t * (0.374092 +
                            t * (0.0967842 +
                                 t * (-0.1862881 +
                                      t * (0.2788681 +
                                           t * (-1.13520398 +
                                                t * (1.4885159 +
                                                     t * (-0.82215223 + t * 0.17087277)))))))));
    return x >= 0.0 ? ans : 2.0 - ans;
}

/* hmf.c:1205-1241; NAN signals the reference's Throw(ValueError) */
double oracle_fgtrm_bias_fast(float growthf, float del_bias, float sig_small, float sig_large,
                              double delta_c) {
    double del, sig;
    if (sig_large > sig_small) return NAN;
    if (sig_large == sig_small) return 0.;
    sig = sqrt(sig_small * sig_small - sig_large * sig_large);
    del = (delta_c - del_bias) / growthf;
    double x = del / (sqrt(2) * sig);
    if (x < 0) return 1.0;
    return erfcc(x);
}

/* interpolation.c:123-131 */
static double eval_table_f(double x, double x_min, double x_width, const float *y_arr) {
    int idx = (int)floor((x - x_min) / x_width);
    double table_val = x_min + x_width * (float)idx;
    double interp_point = (x - table_val) / x_width;
    return y_arr[idx] * (1 - interp_point) + y_arr[idx + 1] * (interp_point);
}

/* interpolation.c:133-157 (a float table read as doubles) */
static double eval_table2d_f(double x, double y, double x_min, double x_width, double y_min,
                             double y_width, const float *z_arr, int ny) {
    int x_idx = (int)floor((x - x_min) / x_width);
    int y_idx = (int)floor((y - y_min) / y_width);
    double x_table = x_min + x_width * (double)x_idx;
    double y_table = y_min + y_width * (double)y_idx;
    double interp_point_x = (x - x_table) / x_width;
    double interp_point_y = (y - y_table) / y_width;
    double left_edge = z_arr[(size_t)x_idx * ny + y_idx] * (1 - interp_point_y) +
                       z_arr[(size_t)x_idx * ny + y_idx + 1] * (interp_point_y);
    double right_edge = z_arr[(size_t)(x_idx + 1) * ny + y_idx] * (1 - interp_point_y) +
                        z_arr[(size_t)(x_idx + 1) * ny + y_idx + 1] * (interp_point_y);
    return left_edge * (1 - interp_point_x) + right_edge * (interp_point_x);
}

/* clip_and_get_extrema: IonisationBox.c:668-699 (extrema of the values before the clip) */
static void clip_extrema(float *grid, int nx, int ny, int nz, double lower_limit,
                         double upper_limit, double *grid_min, double *grid_max) {
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    double min_buf = grid[0], max_buf = grid[0];
#pragma omp parallel for schedule(static) reduction(max : max_buf) reduction(min : min_buf)
    for (long l = 0; l < (long)nx * ny; l++) {
        for (int k = 0; k < nz; k++) {
            float curr = grid[(size_t)l * zpad + k];
            grid[(size_t)l * zpad + k] = fmaxf(fmin(curr, upper_limit), lower_limit);
            if (curr < min_buf) min_buf = curr;
            if (curr > max_buf) max_buf = curr;
        }
    }
    *grid_min = min_buf;
    *grid_max = max_buf;
}

/* bubble_helper_progs.c:262-330 (check_region) and :341-418 (update_in_sphere) */
static void wrap3(int idx[3], const int size[3]) {
    for (int a = 0; a < 3; a++) {
        while (idx[a] >= size[a]) idx[a] -= size[a];
        while (idx[a] < 0) idx[a] += size[a];
    }
}
static void check_region(float *box, int dimensions, int dimensions_ncf, float Rsq_curr_index, int x,
                         int y, int z, int x_min, int x_max, int y_min, int y_max, int z_min,
                         int z_max) {
    const int box_dim[3] = {dimensions, dimensions, dimensions_ncf};
    for (int x_curr = x_min; x_curr <= x_max; x_curr++)
        for (int y_curr = y_min; y_curr <= y_max; y_curr++)
            for (int z_curr = z_min; z_curr <= z_max; z_curr++) {
                int index_arr[3] = {x_curr, y_curr, z_curr};
                wrap3(index_arr, box_dim);
                const size_t index =
                    (size_t)index_arr[2] + (size_t)box_dim[2] * ((size_t)index_arr[1] + (size_t)box_dim[1] * index_arr[0]);
                if (box[index]) { /* all 27 reflections */
                    const float sq[3][3] = {
                        {powf(x - index_arr[0], 2), powf(x - index_arr[0] + dimensions, 2),
                         powf(x - index_arr[0] - dimensions, 2)},
                        {powf(y - index_arr[1], 2), powf(y - index_arr[1] + dimensions, 2),
                         powf(y - index_arr[1] - dimensions, 2)},
                        {powf(z - index_arr[2], 2), powf(z - index_arr[2] + dimensions_ncf, 2),
                         powf(z - index_arr[2] - dimensions_ncf, 2)}};
                    int inside = 0;
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++)
                            for (int c = 0; c < 3; c++)
                                if (Rsq_curr_index > (sq[0][a] + sq[1][b] + sq[2][c])) inside = 1;
                    if (inside) box[index] = 0;
                }
            }
}
static void update_in_sphere(float *box, int dimensions, int dimensions_ncf, float R, float xf,
                             float yf, float zf) {
    const int box_dim[3] = {dimensions, dimensions, dimensions_ncf};
    if (R < 0) return;
    const int x = (int)(xf * dimensions + 0.5), y = (int)(yf * dimensions + 0.5),
              z = (int)(zf * dimensions_ncf + 0.5);
    int R_index = ceil(R / sqrt(3.0) * dimensions) - 1; /* the inner cube is painted outright */
    const int xl_min = x - R_index, xl_max = x + R_index, yl_min = y - R_index,
              yl_max = y + R_index, zl_min = z - R_index, zl_max = z + R_index;
    for (int x_curr = xl_min; x_curr <= xl_max; x_curr++)
        for (int y_curr = yl_min; y_curr <= yl_max; y_curr++)
            for (int z_curr = zl_min; z_curr <= zl_max; z_curr++) {
                int index_arr[3] = {x_curr, y_curr, z_curr};
                wrap3(index_arr, box_dim);
                box[(size_t)index_arr[2] +
                    (size_t)box_dim[2] * ((size_t)index_arr[1] + (size_t)box_dim[1] * index_arr[0])] = 0;
            }
    R_index = ceil(R * dimensions);
    const float Rsq_curr_index = pow(R * dimensions, 2);
    const int xb_min = x - R_index, xb_max = x + R_index, yb_min = y - R_index,
              yb_max = y + R_index, zb_min = z - R_index, zb_max = z + R_index;
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xb_min, xl_min, yb_min,
                 yb_max, zb_min, zb_max);
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xl_max, xb_max, yb_min,
                 yb_max, zb_min, zb_max);
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xb_min, xb_max, yb_min,
                 yl_min, zb_min, zb_max);
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xb_min, xb_max, yl_max,
                 yb_max, zb_min, zb_max);
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xb_min, xb_max, yb_min,
                 yb_max, zb_min, zl_min);
    check_region(box, dimensions, dimensions_ncf, Rsq_curr_index, x, y, z, xb_min, xb_max, yb_min,
                 yb_max, zl_max, zb_max);
}

/* thermochem.c:281-311 */
static double lyman_werner_threshold(const c21cm_mturn_spec *m, float z, float J_21_LW, float vcb) {
    double mcrit_noLW = 3.314e7 * pow(1. + z, -1.5);
    double f_LW = 1.0 + m->A_LW * pow(J_21_LW, m->BETA_LW);
    double f_vcb = pow(1.0 + m->A_VCB * vcb / m->sigma_vcb, m->BETA_VCB);
    return (mcrit_noLW * f_LW * f_vcb);
}
static double reionization_feedback(float z, float Gamma_halo_HII, float z_IN) {
    if (z_IN <= 1e-19) return 1e-40;
    return 3e9 * pow(2.0 * Gamma_halo_HII, 0.17) * pow((1. + z) / 10, -2.1) *
           pow(1 - pow((1. + z) / (1. + z_IN), 2.0), 2.5);
}

/* calculate_mcrit_boxes: IonisationBox.c:403-457 */
int oracle_mturn_grids(const c21cm_mturn_spec *m, const float *prev_G12, const float *prev_z_reion,
                       const float *J_21_LW, const float *vcb, float *log10_mturn_acg,
                       float *log10_mturn_mcg, double *ave_acg, double *ave_mcg) {
    const long ntot = (long)m->hii_dim * m->hii_dim * m->hii_dim_z;
    double ave_a = 0., ave_m = 0.;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : ave_a, ave_m) reduction(| : bad)
    for (long i = 0; i < ntot; i++) {
        double Mcrit_RE = reionization_feedback(m->redshift, prev_G12[i],
                                                m->first_snapshot ? -1.0f : prev_z_reion[i]);
        double Mcrit_LW =
            lyman_werner_threshold(m, m->redshift, J_21_LW[i], vcb ? vcb[i] : (float)m->vcb_const);
        if (Mcrit_LW != Mcrit_LW || Mcrit_LW == 0) bad |= 1;
        double curr_Mt = log10(fmax(Mcrit_RE, m->mturn_a_nofb));
        double curr_Mt_MINI = log10(fmax(Mcrit_RE, fmax(Mcrit_LW, m->mturn_m_nofb)));
        log10_mturn_acg[i] = curr_Mt;
        log10_mturn_mcg[i] = curr_Mt_MINI;
        ave_a += curr_Mt;
        ave_m += curr_Mt_MINI;
    }
    *ave_acg = ave_a / ntot;
    *ave_mcg = ave_m / ntot;
    return bad ? C21CM_VALUE_ERROR : C21CM_OK;
}

/* recombinations.c:64-92: row z_ct of the table, natural cubic spline in ln Gamma evaluated
 * the way gsl_interp_cspline does (b and d from the c coefficients, Horner in delta) */
double oracle_splined_recombination_rate(const double *rr_y, const double *rr_c, double z_eff,
                                         double gamma12_bg) {
    /* a density below -1 (outside PerturbedField's clip) gives a NaN z_eff: row 0, not UB */
    int z_ct = z_eff > 0 ? (int)(fmin(z_eff, 1e6) / C21CM_RR_DZ + 0.5) : 0;
    double lnGamma = log(gamma12_bg);
    if (z_ct < 0)
        z_ct = 0;
    else if (z_ct >= C21CM_RR_NZ)
        z_ct = C21CM_RR_NZ - 1;
    const double top = C21CM_RR_LNGAMMA_MIN + C21CM_RR_DLNGAMMA * (C21CM_RR_NGAMMA - 1);
    if (isnan(lnGamma)) return lnGamma; /* gsl_spline_eval hands a NaN through */
    if (lnGamma < C21CM_RR_LNGAMMA_MIN) return 0;
    if (lnGamma >= top) lnGamma = top - FRACT_FLOAT_ERR;
    const double *y = rr_y + (size_t)z_ct * C21CM_RR_NGAMMA, *c = rr_c + (size_t)z_ct * C21CM_RR_NGAMMA;
#define RRX(g) (C21CM_RR_LNGAMMA_MIN + (g) * C21CM_RR_DLNGAMMA)
    int i = (int)((lnGamma - C21CM_RR_LNGAMMA_MIN) / C21CM_RR_DLNGAMMA);
    if (i > C21CM_RR_NGAMMA - 2) i = C21CM_RR_NGAMMA - 2;
    while (i > 0 && lnGamma < RRX(i)) i--;
    while (i < C21CM_RR_NGAMMA - 2 && lnGamma >= RRX(i + 1)) i++;
    const double x_lo = RRX(i), x_hi = RRX(i + 1), dx = x_hi - x_lo, dy = y[i + 1] - y[i];
#undef RRX
    const double b = (dy / dx) - dx * (c[i + 1] + 2.0 * c[i]) / 3.0;
    const double d = (c[i + 1] - c[i]) / (3.0 * dx);
    const double delx = lnGamma - x_lo;
    return y[i] + delx * (b + delx * (c[i] + delx * d));
}

/* IonisationBox.c:323-360 */
static void prepare_box(const float *input, float *cbox, int nx, int ny, int nz, double factor,
                        double lo, double hi) {
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    const size_t npad = (size_t)nx * ny * zpad;
    memset(cbox, 0, sizeof(float) * npad);
#pragma omp parallel for schedule(static)
    for (long l = 0; l < (long)nx * ny; l++) {
        for (int k = 0; k < nz; k++) {
            double curr_cell = input[(size_t)l * nz + k] * factor;
            cbox[(size_t)l * zpad + k] = fmax(fmin(curr_cell, hi), lo);
        }
    }
    oracle_fft_r2c(cbox, nx, ny, nz);
    const float ntot = (float)((size_t)nx * ny * nz);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)npad; i++) cbox[i] /= ntot;
}

/* one grid of copy_filter_transform: IonisationBox.c:577-663 */
static int copy_filter_c2r(const float *unfiltered, float *filtered, const c21cm_ionize_spec *s,
                           int r_index, int filter_type, double R_param) {
    const int nx = s->hii_dim, ny = s->hii_dim, nz = s->hii_dim_z;
    const size_t npad = (size_t)nx * ny * 2 * (size_t)(nz / 2 + 1);
    memcpy(filtered, unfiltered, sizeof(float) * npad);
    if (r_index > 0) {
        /* filter_box takes float R, float R_param: filtering.c:308 */
        int st = oracle_filter_box(filtered, nx, ny, nz, s->box_len, s->box_len_z, filter_type,
                                   (float)s->R[r_index], (float)R_param);
        if (st) return st;
    }
    oracle_fft_c2r(filtered, nx, ny, nz);
    return C21CM_OK;
}

int oracle_ionize_grids(const c21cm_ionize_spec *s, const PerturbedField *pf,
                        const IonizedBox *prev, const TsBox *ts, const HaloBox *halos,
                        IonizedBox *box, c21cm_ionize_report *report) {
    const int recomb = (s->recomb_model != C21CM_RECOMB_NONE);
    const int inhomo = (s->recomb_model == C21CM_RECOMB_INHOMOGENEOUS);
    const int filter_rec = recomb && !s->cell_recomb; /* IonisationBox.c:156-157 */
    if (recomb && (!s->rr_y || !s->rr_c || !prev || !prev->cumulative_recombinations ||
                   !box->cumulative_recombinations || !box->ionisation_rate_G12))
        return C21CM_VALUE_ERROR;
    if (filter_rec && !inhomo) return C21CM_VALUE_ERROR; /* inputs.py: homogeneous needs CELL_RECOMB */
    if (recomb && s->fcoll_mode == C21CM_FCOLL_STARS_GRID && (!halos || !halos->whalo_sfr))
        return C21CM_VALUE_ERROR;
    if (s->n_radii < 1 || s->n_radii > C21CM_MAX_RADII) return C21CM_VALUE_ERROR;
    const int nx = s->hii_dim, ny = s->hii_dim, nz = s->hii_dim_z;
    const size_t zpad = 2 * (size_t)(nz / 2 + 1);
    const size_t npad = (size_t)nx * ny * zpad;
    const size_t ntot = (size_t)nx * ny * nz;
    const int lagrangian = (s->fcoll_mode == C21CM_FCOLL_STARS_GRID);
    const int use_table =
        (s->fcoll_mode == C21CM_FCOLL_TABLE_LINEAR || s->fcoll_mode == C21CM_FCOLL_TABLE_EXP);
    if (use_table && !s->use_mini_halos && !s->table_fn) return C21CM_VALUE_ERROR;
    if (!lagrangian && !box->unnormalised_nion) return C21CM_VALUE_ERROR;
    if (s->ionise_entire_sphere && (recomb || s->use_mini_halos))
        return C21CM_VALUE_ERROR; /* thread-order dependent upstream: not restated */
    /* USE_MINI_HALOS: Eulerian sources carry their own f_coll history (need_minihalo_nion,
     * IonisationBox.c:30-31); with Lagrangian grids the mini-halos are inside HaloBox.n_ion and only
     * the floor f_limit_mcg enters the barrier (:1068-1082 with ion_eff_factor_mini = 1, :49-50) */
    const int mini_any = s->use_mini_halos;
    const int mini = mini_any && !lagrangian;
    if (mini && (s->fcoll_mode != C21CM_FCOLL_TABLE_EXP || !s->table2d_fn || !s->prev_density ||
                 !s->log10_mturn_acg || !s->log10_mturn_mcg || !box->unnormalised_nion_mini ||
                 !prev || !prev->unnormalised_nion || !prev->unnormalised_nion_mini))
        return C21CM_VALUE_ERROR;
    int status = C21CM_OK;

    /* IonisationBox.c:1372-1378 */
#pragma omp parallel for schedule(static)
    for (long ct = 0; ct < (long)ntot; ct++) box->z_reion[ct] = -1.0;

    /* IonisationBox.c:365-386: on the first snapshot the (caller-zeroed) previous box
     * gets z_reion = -1 written INTO it */
    if (s->first_snapshot && prev && prev->z_reion) {
#pragma omp parallel for schedule(static)
        for (long ct = 0; ct < (long)ntot; ct++) prev->z_reion[ct] = -1.0;
    }

    float *delta_unf = (float *)malloc(sizeof(float) * npad);
    float *delta_fil = (float *)malloc(sizeof(float) * npad);
    float *stars_unf = NULL, *stars_fil = NULL, *xe_unf = NULL, *xe_fil = NULL;
    float *sfr_unf = NULL, *sfr_fil = NULL, *nrec_unf = NULL, *nrec_fil = NULL;
    if (lagrangian) {
        stars_unf = (float *)malloc(sizeof(float) * npad);
        stars_fil = (float *)malloc(sizeof(float) * npad);
        if (recomb) {
            sfr_unf = (float *)malloc(sizeof(float) * npad);
            sfr_fil = (float *)malloc(sizeof(float) * npad);
        }
    }
    if (filter_rec) {
        nrec_unf = (float *)malloc(sizeof(float) * npad);
        nrec_fil = (float *)malloc(sizeof(float) * npad);
    }
    if (s->use_ts_fluct) {
        xe_unf = (float *)malloc(sizeof(float) * npad);
        xe_fil = (float *)malloc(sizeof(float) * npad);
    }
    float table[C21CM_NDELTA_TABLE];
    float *pdelta_unf = NULL, *pdelta_fil = NULL, *mta_unf = NULL, *mta_fil = NULL, *mtm_unf = NULL,
          *mtm_fil = NULL, *tab2d = NULL;
    const size_t t2 = (size_t)C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
    if (mini) {
        pdelta_unf = (float *)malloc(sizeof(float) * npad);
        pdelta_fil = (float *)malloc(sizeof(float) * npad);
        mta_unf = (float *)malloc(sizeof(float) * npad);
        mta_fil = (float *)malloc(sizeof(float) * npad);
        mtm_unf = (float *)malloc(sizeof(float) * npad);
        mtm_fil = (float *)malloc(sizeof(float) * npad);
        tab2d = (float *)malloc(sizeof(float) * 4 * t2); /* acg, mcg, prev acg, prev mcg */
        /* IonisationBox.c:1493-1509: previous delta clipped like delta; the turnover grids are
         * transformed as they are */
        prepare_box(s->prev_density, pdelta_unf, nx, ny, nz, 1., -1, 1e6);
        prepare_box(s->log10_mturn_mcg, mtm_unf, nx, ny, nz, 1., -INFINITY, INFINITY);
        prepare_box(s->log10_mturn_acg, mta_unf, nx, ny, nz, 1., -INFINITY, INFINITY);
    }
    double last_mean_mini = 0.;

    /* IonisationBox.c:1480-1513 */
    prepare_box(pf->density, delta_unf, nx, ny, nz, s->photoncons_adjustment_factor, -1., 1e6);
    if (lagrangian) prepare_box(halos->n_ion, stars_unf, nx, ny, nz, 1., 0., 1e20);
    if (lagrangian && recomb) prepare_box(halos->whalo_sfr, sfr_unf, nx, ny, nz, 1., 0., 1e20);
    if (s->use_ts_fluct) prepare_box(ts->xray_ionised_fraction, xe_unf, nx, ny, nz, 1., 0, 1.);
    if (filter_rec)
        prepare_box(prev->cumulative_recombinations, nrec_unf, nx, ny, nz, 1., 0, 1e20);

    double last_mean = 0.;
    for (int R_ct = s->n_radii; R_ct--;) {
        if (R_ct < s->r_lowest) break; /* IonisationBox.c:1537-1541 */

        status = copy_filter_c2r(delta_unf, delta_fil, s, R_ct, s->hii_filter, 0.);
        if (!status && lagrangian)
            status = copy_filter_c2r(stars_unf, stars_fil, s, R_ct, s->stars_filter,
                                     s->mfp_meandens);
        if (!status && lagrangian && recomb)
            status = copy_filter_c2r(sfr_unf, sfr_fil, s, R_ct, s->stars_filter, s->mfp_meandens);
        if (!status && s->use_ts_fluct)
            status = copy_filter_c2r(xe_unf, xe_fil, s, R_ct, s->hii_filter, 0.);
        if (!status && filter_rec)
            status = copy_filter_c2r(nrec_unf, nrec_fil, s, R_ct, s->hii_filter, 0.);
        if (!status && mini) {
            status = copy_filter_c2r(pdelta_unf, pdelta_fil, s, R_ct, s->hii_filter, 0.);
            if (!status) status = copy_filter_c2r(mtm_unf, mtm_fil, s, R_ct, s->hii_filter, 0.);
            if (!status) status = copy_filter_c2r(mta_unf, mta_fil, s, R_ct, s->hii_filter, 0.);
        }
        if (status) break;

        double tab_min = 0., tab_width = 1.;
        double ptab_min = 0., ptab_width = 1., mta_min = 0., mta_width = 1., mtm_min = 0.,
               mtm_width = 1.;
        if (!lagrangian) {
            /* clip_and_get_extrema(delta_filtered, -1, 1e6): IonisationBox.c:668-699,711-713 */
            double min_buf = delta_fil[0], max_buf = delta_fil[0];
#pragma omp parallel for schedule(static) reduction(max : max_buf) reduction(min : min_buf)
            for (long l = 0; l < (long)nx * ny; l++) {
                for (int k = 0; k < nz; k++) {
                    float curr = delta_fil[(size_t)l * zpad + k];
                    delta_fil[(size_t)l * zpad + k] = fmaxf(fmin(curr, 1e6), -1);
                    if (curr < min_buf) min_buf = curr;
                    if (curr > max_buf) max_buf = curr;
                }
            }
            double min_density = min_buf - 0.001, max_density = max_buf + 0.001;
            if (mini) { /* setup_integration_tables with USE_MINI_HALOS: IonisationBox.c:715-761 */
                double pmin, pmax, amin, amax, mmin, mmax;
                clip_extrema(pdelta_fil, nx, ny, nz, -1, 1e6, &pmin, &pmax);
                clip_extrema(mta_fil, nx, ny, nz, 0., 10., &amin, &amax); /* LOG10_MTURN_MAX */
                clip_extrema(mtm_fil, nx, ny, nz, 0., 10., &mmin, &mmax);
                pmin -= 0.001;
                pmax += 0.001;
                amin = amin * 0.99;
                amax = amax * 1.01;
                mmin = mmin * 0.99;
                mmax = mmax * 1.01;
                status = s->table2d_fn(R_ct, 0, min_density, max_density, amin, amax, mmin, mmax,
                                       tab2d, tab2d + t2, s->table2d_user);
                if (!status && s->need_prev_ion)
                    status = s->table2d_fn(R_ct, 1, pmin, pmax, amin, amax, mmin, mmax,
                                           tab2d + 2 * t2, tab2d + 3 * t2, s->table2d_user);
                if (status) break;
                tab_min = min_density;
                tab_width = (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.);
                ptab_min = pmin;
                ptab_width = (pmax - pmin) / (C21CM_NDELTA_TABLE - 1.);
                mta_min = amin;
                mta_width = (amax - amin) / (C21CM_NMTURN_TABLE - 1.);
                mtm_min = mmin;
                mtm_width = (mmax - mmin) / (C21CM_NMTURN_TABLE - 1.);
            } else if (use_table) {
                status = s->table_fn(R_ct, min_density, max_density, table, s->table_user);
                if (status) break;
                tab_min = min_density;
                tab_width = (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.);
            }
        }

        /* calculate_fcoll_grid: IonisationBox.c:773-962 */
        double f_coll_total = 0., f_coll_MINI_total = 0.;
        int bad = 0;
        const size_t roff = mini ? (size_t)R_ct * ntot : 0; /* fc_r_idx, :783-784 */
#pragma omp parallel for schedule(static) reduction(+ : f_coll_total, f_coll_MINI_total) reduction(| : bad)
        for (long l = 0; l < (long)nx * ny; l++) {
            for (int k = 0; k < nz; k++) {
                const size_t index_f = (size_t)l * zpad + k;
                const size_t index_r = (size_t)l * nz + k;
                delta_fil[index_f] = fmaxf(delta_fil[index_f], -1. + FRACT_FLOAT_ERR);
                if (filter_rec) nrec_fil[index_f] = fmaxf(nrec_fil[index_f], 0.0);
                if (s->use_ts_fluct) {
                    xe_fil[index_f] = fmaxf(xe_fil[index_f], 0.);
                    xe_fil[index_f] = fminf(xe_fil[index_f], 0.999);
                }
                double Splined_Fcoll;
                if (lagrangian) {
                    stars_fil[index_f] = fmaxf(stars_fil[index_f], 0.0);
                    if (recomb) sfr_fil[index_f] = fmaxf(sfr_fil[index_f], 0.0);
                    Splined_Fcoll = stars_fil[index_f];
                } else {
                    double curr_dens = delta_fil[index_f];
                    if (mini) { /* :838-936 */
                        const int ny2 = C21CM_NMTURN_TABLE;
                        double log10_Mturnover = mta_fil[index_f];
                        double log10_Mturnover_MINI = mtm_fil[index_f];
                        double Splined_Fcoll_MINI =
                            exp(eval_table2d_f(curr_dens, log10_Mturnover_MINI, tab_min, tab_width,
                                               mtm_min, mtm_width, tab2d + t2, ny2));
                        double prev_Splined_Fcoll = 0., prev_Splined_Fcoll_MINI = 0.;
                        if (s->need_prev_ion) {
                            double prev_dens = pdelta_fil[index_f];
                            prev_Splined_Fcoll =
                                exp(eval_table2d_f(prev_dens, log10_Mturnover, ptab_min, ptab_width,
                                                   mta_min, mta_width, tab2d + 2 * t2, ny2));
                            prev_Splined_Fcoll_MINI = exp(
                                eval_table2d_f(prev_dens, log10_Mturnover_MINI, ptab_min, ptab_width,
                                               mtm_min, mtm_width, tab2d + 3 * t2, ny2));
                        }
                        Splined_Fcoll = exp(eval_table2d_f(curr_dens, log10_Mturnover, tab_min,
                                                           tab_width, mta_min, mta_width, tab2d, ny2));
                        if (Splined_Fcoll > 1.) Splined_Fcoll = 1.;
                        if (Splined_Fcoll < 0.) Splined_Fcoll = 1e-40;
                        if (prev_Splined_Fcoll > 1.) prev_Splined_Fcoll = 1.;
                        if (prev_Splined_Fcoll < 0.) prev_Splined_Fcoll = 1e-40;
                        box->unnormalised_nion[roff + index_r] =
                            prev->unnormalised_nion[roff + index_r] + Splined_Fcoll -
                            prev_Splined_Fcoll;
                        if (box->unnormalised_nion[roff + index_r] > 1.)
                            box->unnormalised_nion[roff + index_r] = 1.;
                        f_coll_total += box->unnormalised_nion[roff + index_r];
                        if (Splined_Fcoll_MINI > 1.) Splined_Fcoll_MINI = 1.;
                        if (Splined_Fcoll_MINI < 0.) Splined_Fcoll_MINI = 1e-40;
                        if (prev_Splined_Fcoll_MINI > 1.) prev_Splined_Fcoll_MINI = 1.;
                        if (prev_Splined_Fcoll_MINI < 0.) prev_Splined_Fcoll_MINI = 1e-40;
                        box->unnormalised_nion_mini[roff + index_r] =
                            prev->unnormalised_nion_mini[roff + index_r] + Splined_Fcoll_MINI -
                            prev_Splined_Fcoll_MINI;
                        if (box->unnormalised_nion_mini[roff + index_r] > 1.)
                            box->unnormalised_nion_mini[roff + index_r] = 1.;
                        f_coll_MINI_total += box->unnormalised_nion_mini[roff + index_r];
                        continue;
                    }
                    if (s->fcoll_mode == C21CM_FCOLL_ERFC) {
                        Splined_Fcoll =
                            oracle_fgtrm_bias_fast(s->growth_factor, curr_dens, s->sigma_minmass,
                                                   s->sigma_maxmass[R_ct], s->delta_c);
                        if (isnan(Splined_Fcoll)) bad |= 1;
                    } else if (s->fcoll_mode == C21CM_FCOLL_TABLE_LINEAR) {
                        Splined_Fcoll = eval_table_f(curr_dens, tab_min, tab_width, table);
                    } else {
                        Splined_Fcoll = exp(eval_table_f(curr_dens, tab_min, tab_width, table));
                    }
                    box->unnormalised_nion[index_r] = Splined_Fcoll;
                }
                f_coll_total += Splined_Fcoll;
            }
        }
        if (bad) {
            status = C21CM_VALUE_ERROR;
            break;
        }
        /* (checked upstream inside the mini-halo branch only, :914,943) */
        if (mini && (isfinite(f_coll_total) == 0 || isfinite(f_coll_MINI_total) == 0)) {
            status = C21CM_INFINITY_OR_NAN_ERROR;
            break;
        }
        double f_coll_grid_mean = f_coll_total / ntot;
        double f_coll_grid_mean_MINI = f_coll_MINI_total / ntot;
        /* IonisationBox.c:1566-1576 */
        if (s->mass_dep_zeta) {
            if (f_coll_grid_mean <= s->f_limit_acg) f_coll_grid_mean = s->f_limit_acg;
            if (mini_any && f_coll_grid_mean_MINI <= s->f_limit_mcg) f_coll_grid_mean_MINI = s->f_limit_mcg;
        } else {
            if (f_coll_grid_mean <= FRACT_FLOAT_ERR) f_coll_grid_mean = FRACT_FLOAT_ERR;
        }
        if (report) report->f_coll_grid_mean[R_ct] = f_coll_grid_mean;
        if (report && mini_any) report->f_coll_grid_mean_mini[R_ct] = f_coll_grid_mean_MINI;
        last_mean = f_coll_grid_mean;
        last_mean_mini = f_coll_grid_mean_MINI;

        /* find_ionised_regions: IonisationBox.c:1008-1201 */
        double mean_fix_term_acg = 1., mean_fix_term_mcg = 1.;
        if (s->fix_mean) mean_fix_term_acg = s->mean_f_coll / f_coll_grid_mean;
        if (s->fix_mean && mini) mean_fix_term_mcg = s->mean_f_coll_mini / f_coll_grid_mean_MINI;
#pragma omp parallel for schedule(static)
        for (long l = 0; l < (long)nx * ny; l++) {
            for (int k = 0; k < nz; k++) {
                const size_t index_f = (size_t)l * zpad + k;
                const size_t index_r = (size_t)l * nz + k;
                double curr_dens, curr_fcoll, rec = 0., xHII_from_xrays, res_xH;
                if (R_ct == 0)
                    curr_dens = pf->density[index_r] * s->photoncons_adjustment_factor;
                else
                    curr_dens = delta_fil[index_f];
                if (lagrangian)
                    curr_fcoll = stars_fil[index_f];
                else
                    curr_fcoll = box->unnormalised_nion[roff + index_r];
                curr_fcoll = mean_fix_term_acg * curr_fcoll;
                if (lagrangian) curr_fcoll *= 1 / (s->rhocrit_omb * (1 + curr_dens));
                double curr_fcoll_mini = 0.; /* :1068-1074 */
                if (mini)
                    curr_fcoll_mini = mean_fix_term_mcg * box->unnormalised_nion_mini[roff + index_r];
                if (s->mass_dep_zeta) {
                    if (curr_fcoll < s->f_limit_acg) curr_fcoll = s->f_limit_acg;
                    if (mini_any && curr_fcoll_mini < s->f_limit_mcg) curr_fcoll_mini = s->f_limit_mcg;
                }
                const double zeta_m = mini_any ? s->ion_eff_factor_mini : 0.;
                if (recomb) { /* :1084-1099 */
                    if (s->cell_recomb)
                        rec = prev->cumulative_recombinations[inhomo ? index_r : 0];
                    else
                        rec = nrec_fil[index_f];
                    rec /= (1. + curr_dens);
                }
                xHII_from_xrays = s->use_ts_fluct ? xe_fil[index_f] : 0.;

                if (curr_fcoll * s->ion_eff_factor + curr_fcoll_mini * zeta_m >
                    (1. - xHII_from_xrays) * (1.0 + rec)) {
                    /* first crossing (largest R): Gamma_12 and the mean free path, :1124-1140 */
                    if (recomb && (box->neutral_fraction[index_r] > FRACT_FLOAT_ERR)) {
                        if (lagrangian)
                            box->ionisation_rate_G12[index_r] =
                                s->R[R_ct] * s->gamma_prefactor / (1 + curr_dens) * sfr_fil[index_f];
                        else
                            box->ionisation_rate_G12[index_r] =
                                s->R[R_ct] *
                                (s->gamma_prefactor * curr_fcoll +
                                 (mini ? s->gamma_prefactor_mini : 0.) * curr_fcoll_mini);
                        if (!s->minimize_memory && box->mean_free_path)
                            box->mean_free_path[index_r] = s->R[R_ct];
                    }
                    float prev_zre = (s->first_snapshot || !prev || !prev->z_reion)
                                         ? -1.0f
                                         : prev->z_reion[index_r];
                    if (prev_zre < 0)
                        box->z_reion[index_r] = s->redshift;
                    else
                        box->z_reion[index_r] = prev_zre;
                    if (!s->ionise_entire_sphere) /* center method, :1150-1158 */
                        box->neutral_fraction[index_r] = 0;
                    else /* sphere method (the stores of 0 from several threads commute) */
                        update_in_sphere(box->neutral_fraction, nx, nz,
                                         s->R[R_ct] / (double)(float)s->box_len, (l / ny) / (nx + 0.0),
                                         (l % ny) / (nx + 0.0), k / (nz + 0.0));
                } else if (R_ct == 0 && (box->neutral_fraction[index_r] > TINY)) {
                    res_xH = 1. - curr_fcoll * s->ion_eff_factor - curr_fcoll_mini * zeta_m;
                    if (!s->minimize_memory) {
                        if (s->use_ts_fluct) {
                            box->kinetic_temperature[index_r] =
                                oracle_partially_ionized_temperature(
                                    ts->kinetic_temp_neutral[index_r], res_xH, s->T_re);
                        } else {
                            box->kinetic_temperature[index_r] =
                                oracle_partially_ionized_temperature(
                                    s->TK_nofluct *
                                        (1 + s->adia_TK_term * pf->density[index_r]),
                                    res_xH, s->T_re);
                        }
                    }
                    res_xH -= xHII_from_xrays;
                    if (res_xH < 0)
                        res_xH = 0;
                    else if (res_xH > 1)
                        res_xH = 1;
                    box->neutral_fraction[index_r] = res_xH;
                }
            }
        }
    }

    if (!status && !s->minimize_memory) {
        /* set_ionized_temperatures: IonisationBox.c:1203-1256 */
        int nonfinite = 0;
#pragma omp parallel for schedule(static) reduction(| : nonfinite)
        for (long idx = 0; idx < (long)ntot; idx++) {
            if ((box->z_reion[idx] > 0) && (box->neutral_fraction[idx] < TINY)) {
                box->kinetic_temperature[idx] = oracle_fully_ionized_temperature(
                    box->z_reion[idx], s->stored_redshift, pf->density[idx], s->T_re);
                if (s->use_ts_fluct) {
                    if (box->kinetic_temperature[idx] < ts->kinetic_temp_neutral[idx])
                        box->kinetic_temperature[idx] = ts->kinetic_temp_neutral[idx];
                } else {
                    float thistk = s->TK_nofluct * (1 + s->adia_TK_term * pf->density[idx]);
                    if (box->kinetic_temperature[idx] < thistk)
                        box->kinetic_temperature[idx] = thistk;
                }
            }
            if (isfinite(box->kinetic_temperature[idx]) == 0) nonfinite |= 1;
        }
        if (nonfinite) status = C21CM_INFINITY_OR_NAN_ERROR;
    }

    if (!status) {
        /* IonisationBox.c:1594-1615 */
        double global_xH = 0;
#pragma omp parallel for schedule(static) reduction(+ : global_xH)
        for (long ct = 0; ct < (long)ntot; ct++) global_xH += box->neutral_fraction[ct];
        global_xH /= (float)ntot;
        if (isfinite(global_xH) == 0) status = C21CM_INFINITY_OR_NAN_ERROR;
        if (!status && recomb) { /* set_recombination_rates: IonisationBox.c:1258-1340 */
            if (!inhomo) {
                double g12 = 0; /* :1600-1609 (a float reduction upstream) */
#pragma omp parallel for schedule(static) reduction(+ : g12)
                for (long ct = 0; ct < (long)ntot; ct++) g12 += box->ionisation_rate_G12[ct];
                const float global_g12 = (float)(g12 / (float)ntot), global_xHI = (float)global_xH;
                const double dNrec_global =
                    oracle_splined_recombination_rate(s->rr_y, s->rr_c, s->stored_redshift,
                                                      global_g12) *
                    s->fabs_dtdz * s->dz * (1. - global_xHI);
                const double cum = prev->cumulative_recombinations[0] + dNrec_global;
                if (isfinite(cum) == 0) status = C21CM_INFINITY_OR_NAN_ERROR;
                box->cumulative_recombinations[0] = cum;
            } else {
                int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
                for (long idx = 0; idx < (long)ntot; idx++) {
                    const double curr_dens = 1.0 + (pf->density[idx]);
                    double z_eff = pow(curr_dens, 1.0 / 3.0);
                    z_eff *= (1 + s->stored_redshift);
                    const double dNrec =
                        oracle_splined_recombination_rate(s->rr_y, s->rr_c, z_eff - 1.,
                                                          box->ionisation_rate_G12[idx]) *
                        s->fabs_dtdz * s->dz * (1. - box->neutral_fraction[idx]);
                    if (isfinite(dNrec) == 0) bad |= 1;
                    box->cumulative_recombinations[idx] =
                        prev->cumulative_recombinations[idx] + dNrec;
                }
                if (bad) status = C21CM_INFINITY_OR_NAN_ERROR;
            }
        }
        if (report) {
            report->global_xH = global_xH;
            /* IonisationBox.c:1623-1628 */
            report->mean_f_coll_out = s->fix_mean ? s->mean_f_coll : last_mean;
        }
        box->mean_f_coll = s->fix_mean ? s->mean_f_coll : last_mean;
        box->mean_f_coll_MINI = !mini_any ? 0. : (s->fix_mean ? s->mean_f_coll_mini : last_mean_mini);
        if (report && mini_any) report->mean_f_coll_mini_out = box->mean_f_coll_MINI;
    }
    free(pdelta_unf);
    free(pdelta_fil);
    free(mta_unf);
    free(mta_fil);
    free(mtm_unf);
    free(mtm_fil);
    free(tab2d);

    free(delta_unf);
    free(delta_fil);
    free(stars_unf);
    free(stars_fil);
    free(xe_unf);
    free(xe_fil);
    free(sfr_unf);
    free(sfr_fil);
    free(nrec_unf);
    free(nrec_fil);
    return status;
}
