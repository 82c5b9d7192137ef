/* heating.h -- host scalars and tables of the spin-temperature calculation (see heating.c). */
#ifndef C21_HEATING_H
#define C21_HEATING_H

#include "c21cm_grid.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef double (*c21_fn)(double x, void *ctx);

/* QUADPACK QAG with the 15-point Gauss-Kronrod rule (= gsl_integration_qag(.., GSL_INTEG_GAUSS15)
 * with epsabs = 0, limit 1000).  status: 0 converged, 1-4 GSL's roundoff / bad-integrand /
 * max-iteration conditions (the best estimate is still returned). */
double c21_qag15(c21_fn f, void *ctx, double a, double b, double epsrel, double *abserr, int *status);
/* Brent's root bracketing in GSL's bookkeeping (gsl_root_fsolver_brent), iterated until
 * gsl_root_test_interval(x_lo, x_hi, 0, epsrel) holds or max_iter; returns the current iterate. */
double c21_brent_root(c21_fn f, void *ctx, double x_lower, double x_upper, double epsrel,
                      int max_iter, int *status);

/* data tables under config_settings.external_table_path (x_int_tables/, stellar_spectra.dat,
 * recfast_LCDM.dat, and Lyman_alpha_heating_table.dat when USE_LYA_HEATING) */
int c21_heat_load(void);
int init_heat(void);      /* reference name; 0 or a negative code */
void destruct_heat(void); /* reference name */
const double *c21_lya_table(int which);

double c21_frecycle(int n);
double c21_nu_n(int n);
float c21_zmax(float z, int n);
double c21_spectral_emissivity(double nu_norm, int pop);
double c21_spectral_emissivity_lw(double nu_norm, int pop); /* flag 2: the Lyman-Werner band integral */
double c21_EvaluateNionTs_MINI(double z, double log10_mturn); /* valid after c21_ts_prepare_tables */
double c21_EvaluateSFRD_MINI(double z, double log10_mturn);
float c21_interp_fheat(float En, float xHII);
float c21_interp_n_Lya(float En, float xHII);
float c21_interp_nion_HI(float En, float xHII);
float c21_interp_nion_HeI(float En, float xHII);
float c21_interp_nion_HeII(float En, float xHII);
double c21_HI_ion_crosssec(double nu);
double c21_HeI_ion_crosssec(double nu);
double c21_HeII_ion_crosssec(double nu);
double c21_weighted_xray_cross_section(double nu, double x_e);
double c21_nu_integrand(double nu, double x_e, int flag);
double c21_integrate_over_nu(double zp, double local_x_e, double lower_int_limit, int flag);
double c21_minimum_source_mass_xray(double redshift);
double c21_EvaluateNionTs(double z); /* valid after c21_ts_prepare */
double c21_EvaluateSFRD(double z);
double c21_tauX(double nu, double x_e, double x_e_ave, double zp, double zpp, double ion_eff);
double c21_nu_tau_one(double zp, double zpp, double x_e, double ion_eff, int *status);

/* per-snapshot host tables; the spec points into them */
typedef struct c21_ts_tables {
    int n_step, no_light;
    double Q_HI;
    double R_values[C21CM_MAX_TS_RADII], zpp_edge[C21CM_MAX_TS_RADII], zpp[C21CM_MAX_TS_RADII];
    double dzpp[C21CM_MAX_TS_RADII], dtdz[C21CM_MAX_TS_RADII], zpp_growth[C21CM_MAX_TS_RADII];
    double M_min_R[C21CM_MAX_TS_RADII], M_max_R[C21CM_MAX_TS_RADII];
    double starlya_prefactor[C21CM_MAX_TS_RADII], lya_cont_prefactor[C21CM_MAX_TS_RADII];
    double lya_inj_prefactor[C21CM_MAX_TS_RADII], mean_sfr_zpp[C21CM_MAX_TS_RADII];
    double nu_tau_one[C21CM_MAX_TS_RADII];
    double *freq;       /* [3][C21CM_X_INT_NXHII][n_step] */
    float *sfrd_tables; /* [n_step][C21CM_NDELTA_TABLE] */
    float *fcoll_tables, *dfcoll_tables; /* CONST-ION-EFF, same shape */
    double sigma_min[C21CM_MAX_TS_RADII], sigma_max[C21CM_MAX_TS_RADII];
    /* USE_MINI_HALOS: the caller fills ave_log10_mturn (box mean of each shell's filtered
     * log10 M_crit,LW, fill_Rbox_table's average_arr) before c21_ts_prepare_tables */
    double ave_log10_mturn[C21CM_MAX_TS_RADII], mean_sfr_zpp_mini[C21CM_MAX_TS_RADII];
    double starlya_prefactor_mini[C21CM_MAX_TS_RADII], lya_cont_prefactor_mini[C21CM_MAX_TS_RADII];
    double lya_inj_prefactor_mini[C21CM_MAX_TS_RADII];
    double lw_prefactor[C21CM_MAX_TS_RADII], lw_prefactor_mini[C21CM_MAX_TS_RADII];
    float *sfrd_tables_mini; /* [n_step][C21CM_NDELTA_TABLE][C21CM_NMTURN_TABLE] */
} c21_ts_tables;
void c21_ts_tables_free(c21_ts_tables *t);
int c21_ts_prepare(float redshift, float prev_redshift, float perturbed_field_redshift,
                   double x_e_ave, c21cm_ts_spec *spec, c21_ts_tables *tables);
/* the two halves of it: shells / spectral factors / z' constants, then the global tables and the
 * frequency integrals (which ComputeTsBox overlaps with the density filter loop on the device) */
int c21_ts_prepare_shells(float redshift, float prev_redshift, float perturbed_field_redshift,
                          c21cm_ts_spec *spec, c21_ts_tables *tables);
int c21_ts_prepare_tables(double x_e_ave, c21cm_ts_spec *spec, c21_ts_tables *tables);
int c21_ts_sfrd_tables(const double *min_densities, const double *max_densities,
                       c21cm_ts_spec *spec, c21_ts_tables *tables);
int c21_ts_fcoll_tables(const double *min_densities, const double *max_densities,
                        c21cm_ts_spec *spec, c21_ts_tables *tables);

#ifdef __cplusplus
}
#endif
#endif
