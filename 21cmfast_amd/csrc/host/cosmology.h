/* cosmology.h -- host-side scalar cosmology (see cosmology.c). */
#ifndef C21_COSMOLOGY_H
#define C21_COSMOLOGY_H

#include <stddef.h>

#include "c21cm_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* subset of the reference's ScalingConstants (scaling_relations.h:12-52) used here */
typedef struct c21_scaling_consts {
    double fstar_10, alpha_star, fstar_7;
    double t_h, t_star;
    double fesc_10, alpha_esc, fesc_7;
    double pop2_ion, pop3_ion;
    double acg_thresh, mturn_a_nofb;
    double Mlim_Fstar, Mlim_Fesc;
    double l_x;      /* L_X * 1e-38 (scaling_relations.c:63) */
    double redshift; /* the halo metallicity relation depends on it */
    /* mini-halos (scaling_relations.c:53-54,64,87-118); zero unless USE_MINI_HALOS */
    double alpha_star_mini, Mlim_Fstar_mini, Mlim_Fesc_mini;
    double mturn_m_nofb, vcb_const, l_x_mini;
} c21_scaling_consts;

/* exported with the reference's names (bound by py21cmfast's cfuncs layer) */
void init_ps(void);
void free_ps(void);
double dicke(double z);
double sigma_z0(double M);
double dsigmasqdm_z0(double M);
double power_in_k(double k);
double power_in_vcb(double k); /* POWER_SPECTRUM = CLASS only (NaN otherwise) */

int c21_ps_ready(void);
double c21_integrate(double (*f)(double, void *), void *ctx, double a, double b, double rel_tol);
double c21_hubble0(void);
double c21_hubble(float z);
double c21_rhocrit(void);
double c21_nb0(void);
double c21_MtoR(double M);
double c21_RtoM(double R);
double c21_TtoM(double z, double T, double mu);
double c21_dtdz(float z);
double c21_ddickedt(double z);
double c21_sigma_fast(double M); /* spline over a cached ln M table */
double c21_Fcoll_General(double z, double lnM_min, double lnM_max);
double c21_FgtrM_bias_fast(float growthf, float del_bias, float sig_small, float sig_large);
float c21_dfcoll_dz(float z, float sigma_min, float del_bias, float sig_bias);
double c21_Nion_General(double z, double lnM_min, double lnM_max, double Mturn,
                        const c21_scaling_consts *sc);
/* N_ion per unit mass of a region of mass exp(lnM_cond), sigma2 and overdensity delta2
 * (hmf.c:1106-1140); method 0 adaptive, 1 Gauss-Legendre (100 points) */
double c21_Nion_ConditionalM(double growthf, double lnM1, double lnM2, double lnM_cond,
                             double sigma2, double delta2, double Mturn,
                             const c21_scaling_consts *sc, int method);
/* ln N_ion(delta | M_cond) on n_delta overdensities in [dmin, dmax], floored at ln_floor
 * (interp_tables.c:291-405 with -40; the SFRD table :415-494 is the same with f_esc = 1, -50) */
int c21_Nion_Conditional_nodes(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double Mturn, const c21_scaling_consts *sc,
                               double *nodes /* C21CM_NODE_DOUBLES */);
int c21_Nion_Conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double dmin, double dmax, double Mturn,
                               const c21_scaling_consts *sc, int method, double ln_floor,
                               float *table, int n_delta);
/* ln of the X-ray emissivity integral over the conditional mass function (ACG only; hmf.c:482-509,
 * 1142-1177 and interp_tables.c:497-560, floor -50): the same quadrature as the N_ion table with
 * the per-mass weight  s_per_yr * SFR(M) * L_X/SFR(Z(SFR, M_*, z)) */
int c21_Xray_Conditional_table(double growthf, double lnMmin, double lnMmax, double lnMcond,
                               double sigma_cond, double dmin, double dmax, double Mturn,
                               const c21_scaling_consts *sc, int method, float *table,
                               int n_delta);
/* the weight itself, per unit ln M (hmf.c:482-509 without mini-halos) */
double c21_xray_fraction(double lnM, double Mturn, const c21_scaling_consts *sc);
int c21_set_scaling_constants(double redshift, c21_scaling_consts *sc);
/* thermochem.c:281-311 */
double c21_lyman_werner_threshold(float z, float J_21_LW, float vcb);
double c21_reionization_feedback(float z, float Gamma_halo_HII, float z_IN);
/* hmf.c:973-990, 1066-1104 (molecularly cooled galaxies: pivot 1e7, exp(-M/M_acg) upper turnover) */
double c21_Nion_General_MINI(double z, double lnM_min, double lnM_max, double Mturn,
                             const c21_scaling_consts *sc);
double c21_Nion_ConditionalM_MINI(double growthf, double lnM1, double lnM2, double lnM_cond,
                                  double sigma2, double delta2, double Mturn,
                                  const c21_scaling_consts *sc, int method);
/* interp_tables.c:291-405, USE_MINI_HALOS: table[i * n_mturn + j] = max(ln N_ion(delta_i | M_cond;
 * M_turn_j), ln_floor) on n_delta overdensities x n_mturn log-spaced turnover masses
 * (10^l10mt_min .. 10^l10mt_max); mini != 0: the molecularly cooled population.  ln_floor = -40
 * for the N_ion tables, -50 and float_mturn = 1 for the SFRD tables (interp_tables.c:415-494) */
#define C21_NMTURN 50           /* interp_tables.c:28 */
#define C21_LOG10_MTURN_MAX 10. /* interp_tables.c:29-30 */
#define C21_LOG10_MTURN_MIN (5. - 9e-8)
int c21_Nion_Conditional_table2d(double growthf, double lnMmin, double lnMmax, double lnMcond,
                                 double sigma_cond, double dmin, double dmax, double l10mt_min,
                                 double l10mt_max, const c21_scaling_consts *sc, int mini,
                                 int method, double ln_floor, int float_mturn, float *table,
                                 int n_delta, int n_mturn);
/* Nion_General_MINI on a (redshift x log10 turnover) grid, with and without the escape fraction
 * (the spin temperature's Nion_z_table_MINI / SFRD_z_table_MINI, interp_tables.c:96-232) */
int c21_Nion_z_tables_mini(int n_z, double z_min, double z_width, double lnMmin,
                           const c21_scaling_consts *sc, int n_mturn, double l10_min,
                           double l10_width, double *nion, double *sfrd);
/* the scaling constants moved to another redshift (evolve_scaling_constants_to_redshift,
 * scaling_relations.c:132-165) and their star-formation variant (f_esc = 1, :121-130) */
c21_scaling_consts c21_scaling_consts_at_z(double redshift, const c21_scaling_consts *sc);
c21_scaling_consts c21_scaling_consts_sfr(const c21_scaling_consts *sc);
/* mimic_scatter_in_consts (scaling_relations.c:170-197), in place */
int c21_scaling_consts_mimic_scatter(c21_scaling_consts *sc);
size_t c21_scaling_consts_size(void); /* for binding layers that mirror the struct */
double c21_minimum_source_mass(double redshift);
int c21_recfast_load(void);
double c21_T_RECFAST(float z);
double c21_xion_RECFAST(float z);
float c21_cT_approx(float z);

#ifdef __cplusplus
}
#endif
#endif
