/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP) of the reference's grid algorithms for the
 * IC -> PerturbedField -> IonizedBox hot path.  It exists to CHECK the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load liboracle21.so.  Nothing under 21cmfast_amd/ links, imports or
 * calls it; the product fails loudly when its HIP library is missing.
 *
 * Pinning status: the reference itself cannot be built in this image (FFTW and
 * GSL headers/libraries are absent, Python is 3.10 < 3.12), so
 * the oracle is pinned through what the reference HOLDS:
 *  (1) its HDF5 fixtures (the .h5 files of tests/golden/reference, copied data files of the
 *      reference's tests/test_data): with the reference's GSL random stream
 *      restated (oracle_gslrng.c), seed 12345 gives the reference's universe and
 *      the chain ICs -> PerturbedField -> [HaloBox ->] IonizedBox -> BrightnessTemp,
 *      incl. the recombination models evolved over 18 snapshots, reproduces the
 *      reference's binned power spectra / PDFs at the reference's own tolerances
 *      (tests/test_reference_fixtures*.py);
 *  (2) its analytic known-answer tests (tests/test_filtering.py:111-236,
 *      tests/test_perturb.py:108-135, tests/test_initial_conditions.py:153-178)
 *      restated in tests/test_oracle_*.py.
 * Still unpinned: per-cell xH of a mid-reionisation box (no upstream vector
 * exists; all fixtures are at z = 18).
 *
 * The oracle shares the public struct definitions of include/c21cm_grid.h so
 * that one spec drives both implementations.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "c21cm_grid.h"

#ifdef __cplusplus
extern "C" {
#endif

/* oracle_fft.c -- reference: src/py21cmfast/src/dft.c:18-72 */
void oracle_fft_r2c(float *box, int nx, int ny, int nz);
void oracle_fft_c2r(float *box, int nx, int ny, int nz);

/* oracle_filter.c -- reference: src/py21cmfast/src/filtering.c:18-117,308-445 */
double oracle_filter_window(int filter_type, double k, float R, float R_param);
int oracle_filter_box(float *cbox, int nx, int ny, int nz, double box_len, double box_len_z,
                      int filter_type, float R, float R_param);
double oracle_ms_mu(double x_em);
double oracle_ms_eta(double x_em);
double oracle_hyper_2F3(double kR, double alpha, double beta);
double oracle_filter_window_ms(double k, float R_inner, float R_outer, float R_star);
int oracle_filter_box_star(float *cbox, int nx, int ny, int nz, double box_len, double box_len_z,
                           int filter_type, float R, float R_param, float R_star);
int oracle_filter_grid(const float *input, float *output, int nx, int ny, int nz, double box_len,
                       double box_len_z, int filter_type, double R, double R_param);
int oracle_test_filter(const float *input, int nx, int ny, int nz, double box_len,
                       double box_len_z, double R, double R_param, int filter_type,
                       double *result);

/* oracle_ionize.c -- reference: src/py21cmfast/src/IonisationBox.c:323-360,572-1256,1477-1628 */
int oracle_halobox_turnovers(const c21cm_mturn_spec *m, double m_turn, int below_z_heat_max,
                             int n_threads, const float *prev_G12, const float *prev_z_reion,
                             const float *J_21_LW, const float *vcb, float *mturn_a_grid,
                             float *mturn_m_grid, double averages[2]);
int oracle_mturn_grids(const c21cm_mturn_spec *spec, const float *prev_G12,
                       const float *prev_z_reion, const float *J_21_LW, const float *vcb,
                       float *log10_mturn_acg, float *log10_mturn_mcg, double *ave_acg,
                       double *ave_mcg);
int oracle_ionize_grids(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                        const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                        const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report);
double oracle_splined_recombination_rate(const double *rr_y, const double *rr_c, double z_eff,
                                         double gamma12_bg);
float oracle_fully_ionized_temperature(float z_re, float z, float delta, float T_re);
float oracle_partially_ionized_temperature(float T_HI, float res_xH, float T_re);
double oracle_fgtrm_bias_fast(float growthf, float del_bias, float sig_small, float sig_large,
                              double delta_c);

/* oracle_perturb.c -- reference: src/py21cmfast/src/PerturbedField.c, map_mass.c:23-208 */
int oracle_perturb_grids(const c21cm_perturb_spec *spec, const InitialConditions *ics,
                         PerturbedField *pf);

/* oracle_ics.c -- reference: src/py21cmfast/src/InitialConditions.c */
int oracle_ics_grids(const c21cm_ics_spec *spec, InitialConditions *ics);

/* oracle_gslrng.c -- reference: src/py21cmfast/src/rng.c:31-90, InitialConditions.c:26-139
 * (GSL's mt19937 / gfsr4 / choose / shuffle / polar gaussian restated from their published
 * algorithms) */
struct oracle_gsl_rng;
struct oracle_gsl_rng *oracle_gsl_rng_alloc(int kind, unsigned long seed); /* 0 mt19937, 1 gfsr4 */
void oracle_gsl_rng_free(struct oracle_gsl_rng *r);
unsigned int oracle_gsl_rng_get(struct oracle_gsl_rng *r);
double oracle_gsl_ran_ugaussian(struct oracle_gsl_rng *r);
int oracle_gsl_thread_seeds(unsigned long long seed, int n_threads, unsigned int *seeds);
int oracle_gsl_mode_deviates(unsigned long long seed, int n_threads, int nx, int ny, int nz,
                             double *ab);
int oracle_gsl_sample_modes(const c21cm_ics_spec *s, int n_threads, float *cbox);

/* oracle_brightness.c -- reference: src/py21cmfast/src/BrightnessTemperatureBox.c:22-105 */
int oracle_brightness_grids(const c21cm_brightness_spec *spec, const float *density,
                            const float *neutral_fraction, const float *spin_temperature,
                            float *brightness_temp, float *tau_21, double *mean_out);

/* oracle_halobox.c -- reference: src/py21cmfast/src/HaloBox.c:244-436, map_mass.c:62-98,214-344 */
int oracle_halo_props(const c21cm_halo_consts *c, unsigned long long n_halos, const float *masses,
                      const float *coords, const float *star_rng, const float *sfr_rng,
                      const float *xray_rng, const int dim[3], double cell_length, double redshift,
                      int below_z_heat_max, int vcb_flucts, const double lw[7], const float *vcb,
                      const float *J21, const float *z_re, const float *G12, float *out);
int oracle_halobox_grids(const c21cm_halobox_spec *spec, const InitialConditions *ics,
                         HaloBox *grids);

/* oracle_tsfilter.c -- reference: src/py21cmfast/src/SpinTemperatureBox.c:502-520,560-742 */
int oracle_fill_Rbox_grids(const c21cm_rbox_spec *spec, const float *input, float *result,
                           double *min_arr, double *average_arr, double *max_arr);
int oracle_annular_filter_grids(const c21cm_annular_spec *spec, const float *const *inputs,
                                float *const *outputs, double *u_avg, double *f_avg);

/* oracle_ts.c -- reference: src/py21cmfast/src/SpinTemperatureBox.c:892-927,1010-1086,1210-1383,
 * 1499-1848; heating_helper_progs.c:366-760,1210-1313; thermochem.c:66-75 */
int oracle_ts_mcrit_grid(const c21cm_mturn_spec *spec, double m_turn, const float *J_21_LW,
                         const float *vcb, float *log10_mcrit);
int oracle_ts_grids(const c21cm_ts_spec *spec, const float *density, const TsBox *previous,
                    const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                    c21cm_ts_report *report);
int oracle_ts_first_grids(const c21cm_ts_first_spec *spec, const float *density, TsBox *out);
double oracle_kappa_10(double TK);
double oracle_kappa_10_elec(double T);
double oracle_kappa_10_pH(double T);
double oracle_alpha_A(double T);
double oracle_lya_heating_efficiency(double tk, double ts, double taugp, const double *arrE);

void oracle_set_threads(int n);
void oracle_set_fft_threads(int n); /* 0: follow oracle_set_threads; 1: single-threaded FFTs */

#ifdef __cplusplus
}
#endif
#endif
