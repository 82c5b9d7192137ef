/*
 * abi_compute.c -- the three drop-in entry points of the hot path.
 *
 *   int ComputeInitialConditions(unsigned long long seed, InitialConditions *boxes)
 *   int ComputePerturbedField(float z, InitialConditions *boxes, PerturbedField *pf)
 *   int ComputeIonizedBox(float z, float prev_z, PerturbedField*, PerturbedField*, IonizedBox*,
 *                         TsBox*, HaloBox*, InitialConditions*, IonizedBox*)
 * (reference: src/py21cmfast/src/_functionprototypes_wrapper.h:6-9,23-26).
 *
 * Each reads the broadcast parameter structs exactly as the reference does (process
 * globals, src/py21cmfast/src/InputParameters.c:82-90), evaluates the host-side physics
 * scalars (cosmology.c) and hands an explicit-scalar spec to the grid drivers
 * (c21cm_*_grids), which run on the MI355X.  Status codes follow exceptions.h:12-21; no
 * exception crosses the boundary.
 *
 * Option coverage.  Implemented: every PERTURB_ALGORITHM, PERTURB_ON_HIGH_RES,
 * KEEP_3D_VELOCITIES, SMOOTH_EVOLVED_DENSITY_FIELD, analytic POWER_SPECTRUM fits,
 * SOURCE_MODEL = CONST-ION-EFF (closed form or FgtrM table), E-INTEGRAL (conditional
 * mass-function tables per radius, PS / ST, Gauss-Legendre or adaptive quadrature; needs
 * USE_INTERPOLATION_TABLES = hmf-interpolation) and the Lagrangian models (L-INTEGRAL /
 * DEXM-ESF / CHMF-SAMPLER: HaloBox.n_ion supplied), USE_TS_FLUCT, all HII_FILTER types,
 * USE_EXP_FILTER, MINIMIZE_MEMORY, RECOMB_MODEL homogeneous / inhomogeneous with or without
 * CELL_RECOMB, USE_MINI_HALOS with E-INTEGRAL (turnover-mass boxes, 2-D tables, f_coll history)
 * and with the Lagrangian grids (HaloBox.n_ion holds both populations; global means and floors),
 * IONISE_ENTIRE_SPHERE;
 * ComputeBrightnessTemp with or without spin temperatures.
 * Returning ValueError (3) with a message in
 * c21cm_last_error(): E-INTEGRAL without interpolation tables or with the Gamma-function
 * approximation, USE_MINI_HALOS in ComputeHaloBox with PERTURB_ON_HIGH_RES,
 * PHOTON_CONS_TYPE != none, IONISE_ENTIRE_SPHERE together with recombinations or mini-halos.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"
#include "cosmology.h"
#include "heating.h"

#define L_FACTOR 0.620350491 /* Constants.c:41 */
#define FRACT_FLOAT_ERR 1e-7
#define HII_ROUND_ERR 1e-5 /* IonisationBox.c:35 */
#define M_MAX_INTEGRAL 1e16 /* hmf.h:11 */
#define CM_PER_MPC 3.08567758e24
#define SIGMA_HI 6.3e-18

static int require_globals(const char *who, int need_astro) {
    if (!simulation_options_global || !matter_options_global || !cosmo_params_global ||
        !cosmo_tables_global || (need_astro && (!astro_params_global || !astro_options_global))) {
        c21hip_set_error("%s: Broadcast_struct_global_all has not been called", who);
        return C21CM_VALUE_ERROR;
    }
    return 0;
}

/* init_ps on demand; fails (ValueError, message set) when the CLASS tables are unusable */
static int ensure_ps(void) {
    if (!c21_ps_ready()) init_ps();
    return c21_ps_ready() ? 0 : C21CM_VALUE_ERROR;
}

static void geometry(int *dim, int *dim_z, int *hii, int *hii_z, double *len, double *len_z) {
    const SimulationOptions *so = simulation_options_global;
    *dim = so->DIM;
    *dim_z = (int)(so->NON_CUBIC_FACTOR * so->DIM); /* indexing.h D_PARA */
    *hii = so->HII_DIM;
    *hii_z = (int)(so->NON_CUBIC_FACTOR * so->HII_DIM);
    *len = (double)so->BOX_LEN;
    *len_z = (double)(so->BOX_LEN * so->NON_CUBIC_FACTOR); /* float product, filtering.c:313 */
}

/* ------------------------------------------------------------------------------------ */
int ComputePerturbedField(float redshift, InitialConditions *boxes, PerturbedField *pf) {
    int st = require_globals("ComputePerturbedField", 0);
    if (st) return st;
    const SimulationOptions *so = simulation_options_global;
    const MatterOptions *mo = matter_options_global;
    c21cm_perturb_spec s;
    memset(&s, 0, sizeof(s));
    geometry(&s.dim, &s.dim_z, &s.hii_dim, &s.hii_dim_z, &s.box_len, &s.box_len_z);
    s.perturb_algorithm = mo->PERTURB_ALGORITHM;
    s.perturb_on_high_res = mo->PERTURB_ON_HIGH_RES;
    s.keep_3d_velocities = mo->KEEP_3D_VELOCITIES;
    s.smooth_evolved_density = mo->SMOOTH_EVOLVED_DENSITY_FIELD;
    /* PerturbedField.c:222-225: double * float / float */
    s.density_smooth_radius_mpc = so->DENSITY_SMOOTH_RADIUS * so->BOX_LEN / (float)so->HII_DIM;
    s.growth_factor = dicke(redshift);
    s.init_growth_factor = dicke(so->INITIAL_REDSHIFT);
    s.dDdt_over_D = c21_ddickedt(redshift) / dicke(redshift);
    if (!isfinite(s.growth_factor) || !isfinite(s.dDdt_over_D)) return C21CM_VALUE_ERROR;
    return c21cm_perturb_grids(&s, boxes, pf, NULL);
}

/* ------------------------------------------------------------------------------------ */
int ComputeInitialConditions(unsigned long long random_seed, InitialConditions *boxes) {
    int st = require_globals("ComputeInitialConditions", 0);
    if (st) return st;
    if (!boxes || !boxes->hires_density) {
        c21hip_set_error("ComputeInitialConditions: hires_density is required");
        return C21CM_VALUE_ERROR;
    }
    const SimulationOptions *so = simulation_options_global;
    const MatterOptions *mo = matter_options_global;
    const int want_vcb = (mo->V_CB_MODEL == C21CM_VCB_FLUCTS);
    if (want_vcb && mo->POWER_SPECTRUM != C21CM_PS_CLASS) {
        c21hip_set_error("ComputeInitialConditions: V_CB_MODEL=FLUCTS needs POWER_SPECTRUM=CLASS "
                         "(the relative-velocity transfer function)");
        return C21CM_VALUE_ERROR;
    }
    if (want_vcb && !boxes->lowres_vcb) {
        c21hip_set_error("ComputeInitialConditions: V_CB_MODEL=FLUCTS needs lowres_vcb");
        return C21CM_VALUE_ERROR;
    }
    if ((st = ensure_ps())) return st;
    c21cm_ics_spec s;
    memset(&s, 0, sizeof(s));
    geometry(&s.dim, &s.dim_z, &s.hii_dim, &s.hii_dim_z, &s.box_len, &s.box_len_z);
    s.volume = so->BOX_LEN * so->BOX_LEN * so->NON_CUBIC_FACTOR * so->BOX_LEN; /* indexing.h VOLUME */
    s.perturb_algorithm = mo->PERTURB_ALGORITHM;
    s.perturb_on_high_res = mo->PERTURB_ON_HIGH_RES;
    s.seed = random_seed;
    /* Random stream.  Default: the reference's own (seed_rng_threads + gsl_ran_ugaussian in its
     * loop order, rng.c:31-90, InitialConditions.c:103-139), so that the same random_seed and
     * N_THREADS give the same universe as upstream (all five per-thread generators, gsl_stream.c);
     * it is drawn serially on the host, as upstream draws it.  With C21CM_IC_RNG=philox the
     * counter-based device generator is used (a different, equally valid realisation; ~100x faster
     * at DIM = 512).  C21CM_IC_RNG=gsl insists on the former. */
    {
        extern int c21_gsl_stream_supported(int n_threads);
        const char *e = getenv("C21CM_IC_RNG");
        const int n_thr = so->N_THREADS > 0 ? so->N_THREADS : 1;
        s.rng_threads = n_thr;
        if (e && e[0] == 'p')
            s.rng_stream = C21CM_RNG_PHILOX;
        else if (e && e[0] == 'g')
            s.rng_stream = C21CM_RNG_GSL;
        else
            s.rng_stream = c21_gsl_stream_supported(n_thr) ? C21CM_RNG_GSL : C21CM_RNG_PHILOX;
    }

    /* InitialConditions.c:620-634: a non-zero hires_density means "use it as the field" */
    const size_t ntot = (size_t)s.dim * s.dim * s.dim_z;
    if (c21hip_is_device_ptr(boxes->hires_density)) {
        int flag = 0;
        if ((st = c21hip_any_nonzero(boxes->hires_density, ntot, &flag, NULL))) return st;
        s.density_is_input = flag;
    } else {
        for (size_t i = 0; i < ntot; i++)
            if (boxes->hires_density[i]) {
                s.density_is_input = 1;
                break;
            }
    }
    double *pk = NULL;
    if (!s.density_is_input) {
        if (s.dim != s.dim_z) {
            c21hip_set_error("ComputeInitialConditions: mode sampling needs NON_CUBIC_FACTOR = 1");
            return C21CM_VALUE_ERROR;
        }
        s.n_m = 3 * (s.dim / 2) * (s.dim / 2) + 1;
        pk = (double *)malloc(sizeof(double) * (size_t)s.n_m);
        if (!pk) return C21CM_MEMORY_ALLOC_ERROR;
        const double dk = 2.0 * M_PI / s.box_len;
        for (int m = 0; m < s.n_m; m++) pk[m] = power_in_k(dk * sqrt((double)m));
        s.pk_by_m = pk;
    }
    double *vcb = NULL;
    if (want_vcb) { /* compute_relative_velocities, InitialConditions.c:141-238 */
        if (s.dim != s.dim_z) {
            free(pk);
            c21hip_set_error("ComputeInitialConditions: V_CB_MODEL=FLUCTS needs NON_CUBIC_FACTOR = 1");
            return C21CM_VALUE_ERROR;
        }
        const int n_m = 3 * (s.dim / 2) * (s.dim / 2) + 1;
        vcb = (double *)malloc(sizeof(double) * (size_t)n_m);
        if (!vcb) {
            free(pk);
            return C21CM_MEMORY_ALLOC_ERROR;
        }
        const double dk = 2.0 * M_PI / s.box_len;
        vcb[0] = 0.;
        for (int m = 1; m < n_m; m++) { /* sqrt(P_vcb / P) c / |k| (:186-187; c in km/s) */
            const double k = dk * sqrt((double)m);
            vcb[m] = sqrt(power_in_vcb(k) / power_in_k(k)) * 2.99792458e5 / k;
        }
        s.vcb_by_m = vcb;
        s.n_m = n_m;
    }
    st = c21cm_ics_grids(&s, boxes, NULL);
    free(pk);
    free(vcb);
    return st;
}

/* ------------------------------------------------------------------------------------ */
#define fgtrm_bias_fast_host c21_FgtrM_bias_fast /* hmf.c:1221-1241, cosmology.c */

struct fgtrm_table_ctx {
    const c21cm_ionize_spec *spec;
};

/* interp_tables.c:226-250 (initialise_FgtrM_delta_table) */
static int fgtrm_table_fn(int r_index, double dmin, double dmax, float *table, void *user) {
    const c21cm_ionize_spec *s = ((struct fgtrm_table_ctx *)user)->spec;
    const double width = (dmax - dmin) / (C21CM_NDELTA_TABLE - 1.);
    for (int i = 0; i < C21CM_NDELTA_TABLE; i++) {
        const double v = fgtrm_bias_fast_host(s->growth_factor, dmin + i * width, s->sigma_minmass,
                                              s->sigma_maxmass[r_index]);
        if (isnan(v)) return C21CM_VALUE_ERROR;
        table[i] = v;
    }
    return 0;
}

/* E-INTEGRAL: ln N_ion(delta | M(R)) per filter radius, IonisationBox.c:702-765 with
 * interp_tables.c:291-405 (no mini-halos: the 1-D table, turnover mass M_TURN) */
struct nion_table_ctx {
    const c21cm_ionize_spec *spec;
    c21_scaling_consts sc;
    double lnMmin;
    int method;
};

/* C21CM_FCOLL_NODES: the same integrand's node data for the per-cell sums (no interpolation tables) */
static int nion_nodes_fn(int r_index, double dmin, double dmax, float *table, void *user) {
    (void)dmin, (void)dmax;
    const struct nion_table_ctx *t = (const struct nion_table_ctx *)user;
    const c21cm_ionize_spec *s = t->spec;
    const double M_max_R = c21_RtoM(s->R[r_index]);
    return c21_Nion_Conditional_nodes(s->growth_factor, t->lnMmin, log(M_max_R), log(M_max_R),
                                      c21_sigma_fast(M_max_R), t->sc.mturn_a_nofb, &t->sc, (double *)table);
}

static int nion_table_fn(int r_index, double dmin, double dmax, float *table, void *user) {
    const struct nion_table_ctx *t = (const struct nion_table_ctx *)user;
    const c21cm_ionize_spec *s = t->spec;
    const double M_max_R = c21_RtoM(s->R[r_index]);
    return c21_Nion_Conditional_table(s->growth_factor, t->lnMmin, log(M_max_R), log(M_max_R),
                                      c21_sigma_fast(M_max_R), dmin, dmax, t->sc.mturn_a_nofb,
                                      &t->sc, t->method, -40., table, C21CM_NDELTA_TABLE);
}

/* USE_MINI_HALOS: the 2-D tables (overdensity x log10 M_turn) of one radius for both
 * populations, at this redshift or (prev) at the previous snapshot's growth factor with this
 * redshift's scaling constants (IonisationBox.c:744-760, interp_tables.c:291-405) */
struct nion_table2d_ctx {
    const c21cm_ionize_spec *spec;
    c21_scaling_consts sc;
    double lnMmin, prev_growth_factor;
    int method_atomic, method_mini;
};

static int nion_table2d_fn(int r_index, int prev, double dmin, double dmax, double l10mt_min,
                           double l10mt_max, double l10mt_min_mini, double l10mt_max_mini,
                           float *table_acg, float *table_mcg, void *user) {
    const struct nion_table2d_ctx *t = (const struct nion_table2d_ctx *)user;
    const c21cm_ionize_spec *s = t->spec;
    const double M_max_R = c21_RtoM(s->R[r_index]), lnMc = log(M_max_R);
    const double growthf = prev ? t->prev_growth_factor : s->growth_factor;
    const double sigma_c = c21_sigma_fast(M_max_R);
    int st = c21_Nion_Conditional_table2d(growthf, t->lnMmin, lnMc, lnMc, sigma_c, dmin, dmax,
                                          l10mt_min, l10mt_max, &t->sc, 0, t->method_atomic, -40., 0,
                                          table_acg, C21CM_NDELTA_TABLE, C21CM_NMTURN_TABLE);
    if (!st)
        st = c21_Nion_Conditional_table2d(growthf, t->lnMmin, lnMc, lnMc, sigma_c, dmin, dmax,
                                          l10mt_min_mini, l10mt_max_mini, &t->sc, 1,
                                          t->method_mini, -40., 0, table_mcg, C21CM_NDELTA_TABLE,
                                          C21CM_NMTURN_TABLE);
    return st;
}

/* Where the time of the last ComputeIonizedBox of this process went (diagnostic):
 * [0] host ms before the device driver is entered (scalars, sigma(M) and f_coll tables, the spec),
 * [1] device pre-loop, [2] R loop, [3] post-loop (HIP events of c21cm_ionize_grids; 0 on the sharded
 * path), [4] wall ms of the whole call, [5] number of filter radii. */
static double g_ion_timing[6];
int c21cm_last_ionize_timing(double out[6]) {
    if (!out) return C21CM_VALUE_ERROR;
    memcpy(out, g_ion_timing, sizeof(g_ion_timing));
    return 0;
}
static double wall_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int ComputeIonizedBox(float redshift, float prev_redshift, PerturbedField *perturbed_field,
                      PerturbedField *previous_perturbed_field, IonizedBox *previous_ionize_box,
                      TsBox *spin_temp, HaloBox *halos, InitialConditions *ini_boxes,
                      IonizedBox *box) {
    const double t_enter = wall_ms();
    memset(g_ion_timing, 0, sizeof(g_ion_timing));
    int st = require_globals("ComputeIonizedBox", 1);
    if (st) return st;
    const SimulationOptions *so = simulation_options_global;
    const MatterOptions *mo = matter_options_global;
    const AstroParams *ap = astro_params_global;
    const AstroOptions *ao = astro_options_global;
    if (!perturbed_field || !box) return C21CM_VALUE_ERROR;

    const int src = mo->SOURCE_MODEL;
    const int lagrangian = !(src == C21CM_SOURCE_E_INTEGRAL || src == C21CM_SOURCE_CONST_ION_EFF);
    const int mass_dep = src != C21CM_SOURCE_CONST_ION_EFF;
    const char *unsupported = NULL;
    /* E-INTEGRAL without interpolation tables: the per-cell conditional integral runs on the device
     * with the Gauss-Legendre rule (round 4); an adaptive rule per cell does not */
    const int no_tables = src == C21CM_SOURCE_E_INTEGRAL && mo->USE_INTERPOLATION_TABLES != C21CM_INTERP_HMF;
    if (no_tables && ao->INTEGRATION_METHOD_ATOMIC != 1)
        unsupported = "SOURCE_MODEL=E-INTEGRAL without USE_INTERPOLATION_TABLES=hmf-interpolation unless "
                      "INTEGRATION_METHOD_ATOMIC=GAUSS-LEGENDRE";
    if (no_tables && (ao->USE_MINI_HALOS || ao->USE_TS_FLUCT))
        unsupported = "SOURCE_MODEL=E-INTEGRAL without interpolation tables together with USE_MINI_HALOS "
                      "or USE_TS_FLUCT";
    if (src == C21CM_SOURCE_E_INTEGRAL && ao->INTEGRATION_METHOD_ATOMIC > 1)
        unsupported = "INTEGRATION_METHOD_ATOMIC=GAMMA-APPROX";
    /* mini-halos: the Eulerian E-INTEGRAL model (need_minihalo_nion, IonisationBox.c:30-31); the
     * Lagrangian models would take them from a HaloBox that this backend does not fill yet */
    const int mini = ao->USE_MINI_HALOS;
    if (mini && src == C21CM_SOURCE_CONST_ION_EFF)
        unsupported = "USE_MINI_HALOS with SOURCE_MODEL = CONST-ION-EFF";
    if (mini && ao->INTEGRATION_METHOD_MINI > 1) unsupported = "INTEGRATION_METHOD_MINI=GAMMA-APPROX";
    if (ao->PHOTON_CONS_TYPE != C21CM_PHOTONCONS_NONE) unsupported = "PHOTON_CONS_TYPE != none";
    if (ao->IONISE_ENTIRE_SPHERE && (ao->RECOMB_MODEL != C21CM_RECOMB_NONE || mini))
        unsupported = "IONISE_ENTIRE_SPHERE with a recombination model or mini-halos (thread-order "
                      "dependent upstream)";
    if (unsupported) {
        c21hip_set_error("ComputeIonizedBox: %s is not implemented in this backend yet", unsupported);
        return C21CM_VALUE_ERROR;
    }
    if ((st = ensure_ps())) return st;

    c21cm_ionize_spec *s = (c21cm_ionize_spec *)calloc(1, sizeof(*s));
    if (!s) return C21CM_MEMORY_ALLOC_ERROR;
    int dim, dim_z;
    geometry(&dim, &dim_z, &s->hii_dim, &s->hii_dim_z, &s->box_len, &s->box_len_z);

    /* ---- set_ionbox_constants: IonisationBox.c:125-227 */
    c21_scaling_consts sc;
    if ((st = c21_set_scaling_constants(redshift, &sc))) goto done;
    s->redshift = redshift;
    s->stored_redshift = redshift;
    s->photoncons_adjustment_factor = 1.;
    s->dz = (prev_redshift < 1) ? (1. + redshift) * (so->ZPRIME_STEP_FACTOR - 1.)
                                : (double)prev_redshift - redshift;
    s->fabs_dtdz = fabs(c21_dtdz(redshift)) / 1e15;
    s->growth_factor = dicke(redshift);
    s->mass_dep_zeta = mass_dep;
    s->fix_mean = !lagrangian;
    s->hii_filter = ao->HII_FILTER;
    s->stars_filter = ao->USE_EXP_FILTER ? 3 : ao->HII_FILTER;
    s->T_re = ap->T_RE;
    const double ion_eff_factor_gl =
        mass_dep ? sc.pop2_ion * sc.fstar_10 * sc.fesc_10 : (double)ap->HII_EFF_FACTOR;
    s->ion_eff_factor = lagrangian ? 1. : ion_eff_factor_gl;
    s->mfp_meandens = 25.483241248322766 / cosmo_params_global->hlittle;
    const double M_min = c21_minimum_source_mass(redshift);
    const double lnMmin = log(M_min), lnMmax_gl = log(M_MAX_INTEGRAL);
    s->sigma_minmass = c21_sigma_fast(M_min);
    s->delta_c = 1.686;
    s->use_ts_fluct = ao->USE_TS_FLUCT;
    s->minimize_memory = mo->MINIMIZE_MEMORY;
    s->recomb_model = ao->RECOMB_MODEL;
    s->cell_recomb = ao->CELL_RECOMB;
    if (ao->RECOMB_MODEL != C21CM_RECOMB_NONE) { /* init_MHR's tables (recombinations.c:94-122) */
        extern int c21_rr_tables(const double **y_out, const double **c_out);
        if ((st = c21_rr_tables(&s->rr_y, &s->rr_c))) goto done;
    }
    s->first_snapshot = (prev_redshift < 1);
    s->ionise_entire_sphere = ao->IONISE_ENTIRE_SPHERE;
    if (!ao->USE_TS_FLUCT) {
        if ((st = c21_recfast_load())) goto done;
        s->TK_nofluct = c21_T_RECFAST(redshift);
        s->adia_TK_term = c21_cT_approx(redshift);
    }
    const double pixel_length = so->BOX_LEN / (double)so->HII_DIM;
    s->rhocrit_omb = c21_rhocrit() * cosmo_params_global->OMb;
    s->gamma_prefactor = pow(1 + redshift, 2) * CM_PER_MPC * SIGMA_HI * ap->ALPHA_UVB /
                         (ap->ALPHA_UVB + 2.75) * c21_nb0() * s->ion_eff_factor / 1.0e-12;
    if (lagrangian)
        s->gamma_prefactor /= s->rhocrit_omb;
    else
        s->gamma_prefactor /= (sc.t_h * sc.t_star);

    /* ---- setup_radii: IonisationBox.c:964-1006 */
    {
        const double maximum_radius = fmin(ap->R_BUBBLE_MAX, L_FACTOR * so->BOX_LEN);
        double cell_length_factor = L_FACTOR;
        if (lagrangian && !ao->IONISE_ENTIRE_SPHERE && pixel_length < 1) cell_length_factor = 1.;
        const double minimum_radius = fmax(ap->R_BUBBLE_MIN, cell_length_factor * pixel_length);
        int n_radii = (int)(log(maximum_radius / minimum_radius) / log(ap->DELTA_R_HII_FACTOR) + 1);
        if (n_radii < 1 || n_radii > 255) {
            c21hip_set_error("ComputeIonizedBox: %d filter radii (supported: 1..255)", n_radii);
            st = C21CM_VALUE_ERROR;
            goto done;
        }
        for (int i = 0; i < n_radii; i++) {
            double R = minimum_radius * pow(ap->DELTA_R_HII_FACTOR, i);
            if (R > maximum_radius - FRACT_FLOAT_ERR) {
                R = maximum_radius;
                n_radii = i + 1;
            }
            s->R[i] = R;
            s->sigma_maxmass[i] = c21_sigma_fast(c21_RtoM(R));
        }
        s->n_radii = n_radii;
        s->r_lowest = 0;
        for (int r = n_radii; r--;) /* IonisationBox.c:1537-1541 */
            if (M_min > c21_RtoM(s->R[r])) {
                s->r_lowest = r + 1;
                break;
            }
    }

    /* ---- turnover masses and the global mean: IonisationBox.c:1423-1469, :468-529 */
    double Mturn_avg;
    if (lagrangian) {
        if (!halos) {
            c21hip_set_error("ComputeIonizedBox: this SOURCE_MODEL needs a HaloBox");
            st = C21CM_VALUE_ERROR;
            goto done;
        }
        box->log10_Mturnover_ave = halos->log10_Mcrit_ACG_ave;
        box->log10_Mturnover_MINI_ave = halos->log10_Mcrit_MCG_ave;
        Mturn_avg = pow(10., halos->log10_Mcrit_ACG_ave);
        if (mini) { /* the HaloBox holds both populations: only the global means and floors remain */
            if (!previous_ionize_box) {
                c21hip_set_error("ComputeIonizedBox: USE_MINI_HALOS needs the previous IonizedBox "
                                 "(its mean collapsed fractions)");
                st = C21CM_VALUE_ERROR;
                goto done;
            }
            if (s->first_snapshot) { /* setup_first_z_prevbox, :388-400 */
                previous_ionize_box->mean_f_coll = 0.0;
                previous_ionize_box->mean_f_coll_MINI = 0.0;
                if (previous_perturbed_field && previous_perturbed_field->density) {
                    const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
                    float *pd = previous_perturbed_field->density;
                    if (c21hip_is_device_ptr(pd)) {
                        if ((st = c21hip_fill(pd, ntot, -1.5f, NULL))) goto done;
                    } else {
                        for (size_t i = 0; i < ntot; i++) pd[i] = -1.5f;
                    }
                }
            }
            s->use_mini_halos = 1;
        }
    } else if (mini) { /* calculate_mcrit_boxes, :1432-1445 */
        if (!previous_ionize_box || !previous_perturbed_field || !previous_perturbed_field->density ||
            !spin_temp || !spin_temp->J_21_LW || !box->unnormalised_nion_mini ||
            !previous_ionize_box->ionisation_rate_G12 ||
            (mo->V_CB_MODEL == C21CM_VCB_FLUCTS && (!ini_boxes || !ini_boxes->lowres_vcb))) {
            c21hip_set_error("ComputeIonizedBox: USE_MINI_HALOS needs the previous PerturbedField "
                             "and IonizedBox (Gamma_12, z_reion, f_coll histories), TsBox.J_21_LW "
                             "and, with V_CB_MODEL = FLUCTS, lowres_vcb");
            st = C21CM_VALUE_ERROR;
            goto done;
        }
        const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
        if (s->first_snapshot) { /* setup_first_z_prevbox, :388-400 */
            previous_ionize_box->mean_f_coll = 0.0;
            previous_ionize_box->mean_f_coll_MINI = 0.0;
            float *pd = previous_perturbed_field->density; /* "Makes Fcoll == 0." */
            if (c21hip_is_device_ptr(pd)) {
                if ((st = c21hip_fill(pd, ntot, -1.5f, NULL))) goto done;
            } else {
                for (size_t i = 0; i < ntot; i++) pd[i] = -1.5f;
            }
        }
        float *mta = (float *)c21hip_ws(205, ntot * sizeof(float));
        float *mtm = (float *)c21hip_ws(206, ntot * sizeof(float));
        if (!mta || !mtm) {
            st = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        c21cm_mturn_spec ms;
        memset(&ms, 0, sizeof(ms));
        ms.hii_dim = s->hii_dim;
        ms.hii_dim_z = s->hii_dim_z;
        ms.first_snapshot = s->first_snapshot;
        ms.redshift = redshift;
        ms.mturn_a_nofb = sc.mturn_a_nofb;
        ms.mturn_m_nofb = sc.mturn_m_nofb;
        ms.vcb_const = sc.vcb_const;
        ms.A_LW = ap->A_LW;
        ms.BETA_LW = ap->BETA_LW;
        ms.A_VCB = ap->A_VCB;
        ms.BETA_VCB = ap->BETA_VCB;
        ms.sigma_vcb = cosmo_tables_global->V_CB_AVG * sqrt(3 * M_PI / 8);
        if ((st = c21cm_mturn_grids(&ms, previous_ionize_box->ionisation_rate_G12,
                                    previous_ionize_box->z_reion, spin_temp->J_21_LW,
                                    mo->V_CB_MODEL == C21CM_VCB_FLUCTS ? ini_boxes->lowres_vcb : NULL,
                                    mta, mtm, &box->log10_Mturnover_ave,
                                    &box->log10_Mturnover_MINI_ave, NULL)))
            goto done;
        Mturn_avg = pow(10., box->log10_Mturnover_ave);
        s->use_mini_halos = 1;
        s->prev_density = previous_perturbed_field->density;
        s->log10_mturn_acg = mta;
        s->log10_mturn_mcg = mtm;
    } else if (mass_dep) { /* E-INTEGRAL without mini-halos: the turnover mass, :1446-1449 */
        Mturn_avg = ap->M_TURN;
        box->log10_Mturnover_ave = log10(Mturn_avg);
        box->log10_Mturnover_MINI_ave = 0.0;
    } else { /* CONST-ION-EFF: "just store the sharp cutoff mass", :1450-1454 */
        Mturn_avg = M_min;
        box->log10_Mturnover_ave = log10(M_min);
        box->log10_Mturnover_MINI_ave = 0.0;
    }
    if (mass_dep) {
        s->mean_f_coll = c21_Nion_General(redshift, lnMmin, lnMmax_gl, Mturn_avg, &sc);
        s->f_limit_acg = c21_Nion_General(so->Z_HEAT_MAX, lnMmin, lnMmax_gl, Mturn_avg, &sc);
    } else {
        s->mean_f_coll = c21_Fcoll_General(redshift, lnMmin, lnMmax_gl);
        s->f_limit_acg = c21_Fcoll_General(so->Z_HEAT_MAX, lnMmin, lnMmax_gl);
    }
    if (!isfinite(s->mean_f_coll) || s->mean_f_coll < 0) {
        c21hip_set_error("ComputeIonizedBox: mean collapse fraction is invalid (HMF %d)", mo->HMF);
        st = C21CM_INFINITY_OR_NAN_ERROR;
        goto done;
    }
    box->mean_f_coll = s->mean_f_coll;
    box->mean_f_coll_MINI = 0.;
    double exp_global_hii = s->mean_f_coll * ion_eff_factor_gl;
    struct nion_table2d_ctx n2ctx;
    if (mini) { /* set_mean_fcoll with the trapezoidal history, :476-501 */
        const double ion_eff_factor_mini_gl = sc.pop3_ion * sc.fstar_7 * sc.fesc_7;
        const double Mturn_avg_mini = pow(10., box->log10_Mturnover_MINI_ave);
        const double f_coll_curr = s->mean_f_coll;
        if (!(previous_ionize_box->mean_f_coll * ion_eff_factor_gl < 1e-4))
            s->mean_f_coll = previous_ionize_box->mean_f_coll + f_coll_curr -
                             c21_Nion_General(prev_redshift, lnMmin, lnMmax_gl, Mturn_avg, &sc);
        const double f_coll_curr_mini =
            c21_Nion_General_MINI(redshift, lnMmin, lnMmax_gl, Mturn_avg_mini, &sc);
        if (previous_ionize_box->mean_f_coll_MINI * ion_eff_factor_gl < 1e-4) /* (sic: zeta of the ACGs) */
            s->mean_f_coll_mini = f_coll_curr_mini;
        else
            s->mean_f_coll_mini =
                previous_ionize_box->mean_f_coll_MINI + f_coll_curr_mini -
                c21_Nion_General_MINI(prev_redshift, lnMmin, lnMmax_gl, Mturn_avg_mini, &sc);
        s->f_limit_mcg = c21_Nion_General_MINI(so->Z_HEAT_MAX, lnMmin, lnMmax_gl, Mturn_avg_mini, &sc);
        if (!isfinite(s->mean_f_coll) || s->mean_f_coll < 0 || !isfinite(s->mean_f_coll_mini) ||
            s->mean_f_coll_mini < 0) {
            c21hip_set_error("ComputeIonizedBox: mean collapse fraction is invalid (%g, mini-halos %g)",
                             s->mean_f_coll, s->mean_f_coll_mini);
            st = C21CM_INFINITY_OR_NAN_ERROR;
            goto done;
        }
        box->mean_f_coll = s->mean_f_coll;
        box->mean_f_coll_MINI = s->mean_f_coll_mini;
        exp_global_hii = s->mean_f_coll * ion_eff_factor_gl + s->mean_f_coll_mini * ion_eff_factor_mini_gl;
        s->ion_eff_factor_mini = lagrangian ? 1. : ion_eff_factor_mini_gl; /* :49-53 */
        s->gamma_prefactor_mini = s->gamma_prefactor * s->ion_eff_factor_mini / s->ion_eff_factor;
        s->need_prev_ion = previous_ionize_box->mean_f_coll_MINI * ion_eff_factor_mini_gl +
                               previous_ionize_box->mean_f_coll * ion_eff_factor_gl > 1e-4;
        n2ctx.spec = s;
        n2ctx.sc = sc;
        n2ctx.lnMmin = lnMmin;
        n2ctx.prev_growth_factor = dicke(prev_redshift);
        n2ctx.method_atomic = ao->INTEGRATION_METHOD_ATOMIC;
        n2ctx.method_mini = ao->INTEGRATION_METHOD_MINI;
        s->table2d_fn = nion_table2d_fn;
        s->table2d_user = &n2ctx;
    }

    struct fgtrm_table_ctx tctx = {s};
    struct nion_table_ctx nctx;
    if (lagrangian) {
        s->fcoll_mode = C21CM_FCOLL_STARS_GRID;
    } else if (src == C21CM_SOURCE_E_INTEGRAL) {
        nctx.spec = s;
        nctx.sc = sc;
        nctx.lnMmin = lnMmin;
        nctx.method = ao->INTEGRATION_METHOD_ATOMIC;
        s->fcoll_mode = no_tables ? C21CM_FCOLL_NODES : C21CM_FCOLL_TABLE_EXP;
        s->table_fn = no_tables ? nion_nodes_fn : nion_table_fn;
        s->table_user = &nctx;
    } else if (mo->USE_INTERPOLATION_TABLES == C21CM_INTERP_HMF) {
        s->fcoll_mode = C21CM_FCOLL_TABLE_LINEAR;
        s->table_fn = fgtrm_table_fn;
        s->table_user = &tctx;
    } else {
        s->fcoll_mode = C21CM_FCOLL_ERFC;
    }

    if (exp_global_hii < HII_ROUND_ERR) {
        /* set_fully_neutral_box: IonisationBox.c:531-565 */
        const size_t ntot = (size_t)s->hii_dim * s->hii_dim * s->hii_dim_z;
        st = c21cm_neutral_box(s, perturbed_field, spin_temp, box, ntot);
        goto done;
    }
    {
        /* one process per GPU with an initialised communicator (c21cm_shard_init): the R loop
         * is sharded over the ranks (C21CM_SHARD=0: single GPU per process).  What the output arrays
         * hold afterwards: WHOLE BOXES ON EVERY RANK, as the reference's caller expects of a
         * ComputeIonizedBox that returned 0 (round 6, ADVICE r5: anything that reads the box next --
         * ComputeBrightnessTemp, a power spectrum, the wrapper's cache -- saw unwritten cells outside
         * the rank's slab when the slab-resident form was the default).  Where the finish phase runs by
         * cell slabs that is an all-gather of the output slabs (12 bytes per cell, 7/8 of it incoming
         * over seven links); models that finish on one rank broadcast its box.  Slab-resident outputs
         * (a rank's slab of the box + the complete scalars: what the next snapshot's per-cell work
         * needs) are an explicit opt-in: C21CM_SHARD_OUTPUT=none | xH, or c21cm_shard_set_output(). */
        int srank, sworld;
        const char *e = getenv("C21CM_SHARD");
        const int out_mode = c21cm_shard_output_mode() == -1 ? 1 : c21cm_shard_output_mode();
        /* (a USE_MINI_HALOS run keeps one f_coll history slice per radius: every rank runs the
         * whole R loop) */
        /* (C21CM_SHARD=force: also on a one-rank communicator -- the plumbing test of a 1-GPU box) */
        if (c21cm_shard_info(&srank, &sworld) == 0 && (sworld > 1 || (e && e[0] == 'f')) &&
            !(e && e[0] == '0') && !mini)
            st = c21cm_ionize_sharded(s, perturbed_field, previous_ionize_box, spin_temp, halos,
                                      box, NULL, out_mode, NULL);
        else {
            c21cm_ionize_report *rep = (c21cm_ionize_report *)calloc(1, sizeof(*rep));
            g_ion_timing[0] = wall_ms() - t_enter;
            g_ion_timing[5] = s->n_radii;
            st = c21cm_ionize_grids(s, perturbed_field, previous_ionize_box, spin_temp, halos, box,
                                    rep, NULL);
            if (rep) {
                g_ion_timing[1] = rep->ms_preloop;
                g_ion_timing[2] = rep->ms_rloop;
                g_ion_timing[3] = rep->ms_postloop;
                free(rep);
            }
        }
    }
done:
    g_ion_timing[4] = wall_ms() - t_enter;
    free(s);
    return st;
}

/* reference: src/py21cmfast/src/BrightnessTemperatureBox.c:22-105 */
int ComputeBrightnessTemp(float redshift, TsBox *spin_temp, IonizedBox *ionized_box,
                          PerturbedField *perturb_field, BrightnessTemp *box) {
    int st = require_globals("ComputeBrightnessTemp", 1);
    if (st) return st;
    if (!ionized_box || !perturb_field || !box) return C21CM_VALUE_ERROR;
    const SimulationOptions *so = simulation_options_global;
    const CosmoParams *cp = cosmo_params_global;
    c21cm_brightness_spec s;
    memset(&s, 0, sizeof(s));
    s.n_cells = (size_t)so->HII_DIM * so->HII_DIM * (size_t)(so->NON_CUBIC_FACTOR * so->HII_DIM);
    s.redshift = redshift;
    s.use_ts_fluct = astro_options_global->USE_TS_FLUCT;
    s.T_rad = (float)(2.7255 * (1 + redshift)); /* physconst.T_cmb, Constants.c:25 */
    s.const_factor = (float)(27 * (cp->OMb * cp->hlittle * cp->hlittle / 0.023) *
                             sqrt((0.15 / (cp->OMm) / (cp->hlittle) / (cp->hlittle)) *
                                  (1. + redshift) / 10.0));
    return c21cm_brightness_grids(&s, perturb_field->density, ionized_box->neutral_fraction,
                                  spin_temp ? spin_temp->spin_temperature : NULL,
                                  box->brightness_temp, box->tau_21, NULL, NULL);
}

/* reference: src/py21cmfast/src/SpinTemperatureBox.c:748-808 */
int UpdateXraySourceBox(HaloBox *halobox, double R_inner, double R_outer, int R_ct, double R_star,
                        XraySourceBox *source_box) {
    int st = require_globals("UpdateXraySourceBox", 1);
    if (st) return st;
    const AstroOptions *ao = astro_options_global;
    if (!halobox || !source_box || !halobox->halo_sfr || !halobox->halo_xray ||
        !source_box->filtered_sfr || !source_box->filtered_xray || !source_box->mean_sfr) {
        c21hip_set_error("UpdateXraySourceBox: halo_sfr / halo_xray and their filtered grids are required");
        return C21CM_VALUE_ERROR;
    }
    if (R_ct < 0 || R_ct >= astro_params_global->N_STEP_TS) {
        c21hip_set_error("UpdateXraySourceBox: R_ct %d outside [0, N_STEP_TS = %d)", R_ct,
                         astro_params_global->N_STEP_TS);
        return C21CM_VALUE_ERROR;
    }
    c21cm_annular_spec s;
    memset(&s, 0, sizeof(s));
    int dim, dim_z;
    geometry(&dim, &dim_z, &s.hii_dim, &s.hii_dim_z, &s.box_len, &s.box_len_z);
    s.R_inner = R_inner;
    s.R_outer = R_outer;
    s.R_star = R_star;
    const size_t off = (size_t)R_ct * s.hii_dim * s.hii_dim * (size_t)s.hii_dim_z;
    const int lya = ao->LYA_MULTIPLE_SCATTERING ? 5 : 4; /* :755 */
    const float *in[C21CM_MAX_ANNULAR_GRIDS];
    float *out[C21CM_MAX_ANNULAR_GRIDS];
    int n = 0, i_mini = -1;
    in[n] = halobox->halo_sfr, out[n] = source_box->filtered_sfr + off, s.filter_type[n++] = lya;
    in[n] = halobox->halo_xray, out[n] = source_box->filtered_xray + off, s.filter_type[n++] = 4;
    if (ao->USE_MINI_HALOS) {
        if (!halobox->halo_sfr_mini || !source_box->filtered_sfr_mini || !source_box->mean_sfr_mini ||
            !source_box->mean_log10_Mcrit_LW ||
            (ao->LYA_MULTIPLE_SCATTERING &&
             (!source_box->filtered_sfr_lw || !source_box->filtered_sfr_mini_lw))) {
            c21hip_set_error("UpdateXraySourceBox: USE_MINI_HALOS needs the mini-halo grids");
            return C21CM_VALUE_ERROR;
        }
        i_mini = n;
        in[n] = halobox->halo_sfr_mini, out[n] = source_box->filtered_sfr_mini + off, s.filter_type[n++] = lya;
        if (ao->LYA_MULTIPLE_SCATTERING) { /* LW photons travel in straight lines, :786-796 */
            in[n] = halobox->halo_sfr, out[n] = source_box->filtered_sfr_lw + off, s.filter_type[n++] = 4;
            in[n] = halobox->halo_sfr_mini, out[n] = source_box->filtered_sfr_mini_lw + off, s.filter_type[n++] = 4;
        }
    }
    s.n_grids = n;
    double u_avg[C21CM_MAX_ANNULAR_GRIDS], f_avg[C21CM_MAX_ANNULAR_GRIDS];
    st = c21cm_annular_filter_grids(&s, in, out, u_avg, f_avg, NULL);
    if (st) return st;
    source_box->mean_sfr[R_ct] = f_avg[0];
    if (i_mini >= 0) {
        source_box->mean_sfr_mini[R_ct] = f_avg[i_mini];
        source_box->mean_log10_Mcrit_LW[R_ct] = halobox->log10_Mcrit_MCG_ave;
    }
    return 0;
}

/* the density filter loop of ComputeTsBox as a job for a helper thread */
struct filter_job {
    c21cm_rbox_spec spec;
    const float *input;
    float *result;
    double mn[C21CM_MAX_TS_RADII], av[C21CM_MAX_TS_RADII], mx[C21CM_MAX_TS_RADII];
    int device, status;
};

static void *filter_job_run(void *arg) {
    struct filter_job *j = (struct filter_job *)arg;
    j->status = c21hip_use_device(j->device); /* the current device is per thread */
    if (!j->status)
        j->status = c21cm_fill_Rbox_grids(&j->spec, j->input, j->result, j->mn, j->av, j->mx, NULL);
    return NULL;
}

static double wall_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Box mean as ts_main takes it (:1479-1489): double sum, divided by (float)N. */
static int box_mean(const float *v, size_t n, double *mean) {
    if (c21hip_is_device_ptr(v)) {
        double *scratch = (double *)c21hip_ws(171, (C21HIP_PARTIALS + 1) * sizeof(double));
        if (!scratch) return C21CM_MEMORY_ALLOC_ERROR;
        int st = c21hip_sum_float(v, n, scratch + 1, scratch, NULL);
        if (st) return st;
        double sum;
        if ((st = c21hip_d2h(&sum, scratch, sizeof(double), NULL))) return st;
        if ((st = c21hip_sync(NULL))) return st;
        *mean = sum / (float)n;
        return 0;
    }
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += v[i];
    *mean = sum / (float)n;
    return 0;
}

/* reference: src/py21cmfast/src/SpinTemperatureBox.c:87-115 (ComputeTsBox) and :1387-1946 (ts_main).
 * Host: shells, spectral factors, z' constants, N_ion(z) / SFRD(z) tables, tau_X = 1 frequencies
 * and the frequency-integral tables (heating.c); device: the density filter loop of the Eulerian
 * source model (tsfilter_driver.c) and the cell sweep (ts_driver.c / ts_kernels.hip).
 * Supported: the Eulerian source models (E-INTEGRAL: SFRD tables, CONST-ION-EFF: dfcoll/dz tables of
 * the filtered density) and the Lagrangian ones (XraySourceBox grids), with interpolation tables;
 * USE_MINI_HALOS: E-INTEGRAL (2-D tables, the Lyman-Werner grid) and source grids
 * (XraySourceBox.filtered_sfr_mini [+ the LW copies]); J_21_LW out. */
/* ---- sharding of ComputeTsBox over the ranks of a node ----------------------
 * The N_STEP_TS shells are independent until their sums meet in the cell (SpinTemperatureBox.c:
 * 1541-1784 is linear in the shells), so they are dealt round-robin like the radii of the
 * excursion set: rank r takes shells n-1-r, n-1-r-world, ...  Each rank filters the density at
 * ITS shells' radii only (the costly part: one pass X/Y/Z per shell), builds their tables, takes
 * their box means and runs the shell loop over them: six partial sums per cell.  One exchange
 * (reduce-scatter by cell slabs: every rank ends up with the COMPLETE sums of N / world cells), the
 * temperature update on that slab, one all-gather of the three output boxes.  ts_shard_* below are
 * the two compute phases; shard_rccl.c carries the exchange. */
enum { TS_RUN_ALL = 0, TS_RUN_SHARD_SUMS = 1, TS_RUN_SHARD_FINISH = 2 };
typedef struct {
    int mode, rank, world;
    double *sums;        /* SHARD_SUMS: out [6][N]; SHARD_FINISH: in [6][ncell] (device) */
    size_t cell0, ncell; /* SHARD_FINISH */
    int n_rows;          /* out: rows of `sums` in use (4, or 6 with USE_LYA_HEATING) */
} ts_shard_args;

static int ts_box_run(float redshift, float prev_redshift, float perturbed_field_redshift,
                      PerturbedField *perturbed_field, XraySourceBox *source_box,
                      TsBox *previous_spin_temp, InitialConditions *ini_boxes, TsBox *this_spin_temp,
                      ts_shard_args *sh);

/* The host preparation of a snapshot (shells, spectral factors, global tables: 3-8 ms on 16-64 threads) handed
 * from phase 1 of a sharded ComputeTsBox to its phase 2, which used to build all of it again for a slab whose
 * device work is 0.7 ms (round 6).  Keyed on the redshifts, the box mean of the previous x_e and every byte of the
 * installed parameter structs; the spec / tables as they were BEFORE phase 1 compacted them to its shells. */
unsigned long long c21_params_fingerprint(void); /* params.c */
static struct {
    int valid;
    float z, prev_z, pf_z;
    double x_e_ave;
    unsigned long long params;
    c21cm_ts_spec spec;
    c21_ts_tables tab;
    size_t freq_doubles;
    double *freq;
} g_ts_prep;
static int ts_prep_matches(float z, float prev_z, float pf_z, double x_e_ave) {
    return g_ts_prep.valid && g_ts_prep.z == z && g_ts_prep.prev_z == prev_z && g_ts_prep.pf_z == pf_z &&
           g_ts_prep.x_e_ave == x_e_ave && g_ts_prep.params == c21_params_fingerprint();
}
static void ts_prep_store(float z, float prev_z, float pf_z, double x_e_ave, const c21cm_ts_spec *spec,
                          const c21_ts_tables *tab) {
    g_ts_prep.valid = 0;
    const size_t nd = 3 * (size_t)C21CM_X_INT_NXHII * tab->n_step;
    if (!tab->freq || tab->sfrd_tables || tab->sfrd_tables_mini || tab->fcoll_tables || tab->dfcoll_tables) return; /* (only the state right after prepare_tables) */
    double *f = (double *)realloc(g_ts_prep.freq, nd * sizeof(double));
    if (!f) return;
    memcpy(f, tab->freq, nd * sizeof(double));
    g_ts_prep.freq = f, g_ts_prep.freq_doubles = nd;
    g_ts_prep.spec = *spec, g_ts_prep.tab = *tab;
    g_ts_prep.tab.freq = NULL, g_ts_prep.tab.shell_mask = NULL;
    g_ts_prep.z = z, g_ts_prep.prev_z = prev_z, g_ts_prep.pf_z = pf_z, g_ts_prep.x_e_ave = x_e_ave;
    g_ts_prep.params = c21_params_fingerprint();
    g_ts_prep.valid = 1;
}
/* spec / tab <- the stored preparation (tab owns a copy of the frequency integrals); 0: nothing stored for this call */
static int ts_prep_load(float z, float prev_z, float pf_z, double x_e_ave, c21cm_ts_spec *spec, c21_ts_tables *tab) {
    if (!ts_prep_matches(z, prev_z, pf_z, x_e_ave)) return 0;
    double *f = (double *)malloc(g_ts_prep.freq_doubles * sizeof(double));
    if (!f) return 0;
    memcpy(f, g_ts_prep.freq, g_ts_prep.freq_doubles * sizeof(double));
    c21_ts_tables_free(tab);
    *spec = g_ts_prep.spec, *tab = g_ts_prep.tab;
    tab->freq = f;
    const size_t fn = (size_t)C21CM_X_INT_NXHII * tab->n_step;
    spec->freq_int_heat = f, spec->freq_int_ion = f + fn, spec->freq_int_lya = f + 2 * fn;
    return 1;
}

int ComputeTsBox(float redshift, float prev_redshift, float perturbed_field_redshift, short cleanup,
                 PerturbedField *perturbed_field, XraySourceBox *source_box,
                 TsBox *previous_spin_temp, InitialConditions *ini_boxes, TsBox *this_spin_temp) {
    (void)cleanup;
    /* one process per GPU with an RCCL communicator (c21cm_shard_init): shells dealt over the
     * ranks, every rank returns the full boxes (C21CM_SHARD_TS=0 keeps the replicated
     * computation, which is also what an emulated transport gets: c21cm_ts_box_sharded needs real
     * point-to-point transfers).  Whether THIS call can be sharded depends on the caller's arrays
     * (device pointers), so the ranks agree on it before any of them takes the path. */
    int srank = 0, sworld = 1;
    const char *e = getenv("C21CM_SHARD_TS");
    /* (C21CM_SHARD_TS=force: also on a one-rank communicator -- the plumbing test of a 1-GPU box, as
     * C21CM_SHARD=force for ComputeIonizedBox) */
    if (c21cm_shard_info(&srank, &sworld) == 0 && (sworld > 1 || (e && e[0] == 'f')) && c21cm_shard_is_rccl() &&
        !(e && e[0] == '0') &&
        c21cm_shard_all_agree(
            c21cm_ts_shardable(redshift, perturbed_field, previous_spin_temp, this_spin_temp)))
        return c21cm_ts_box_sharded(redshift, prev_redshift, perturbed_field_redshift,
                                    perturbed_field, previous_spin_temp, this_spin_temp);
    ts_shard_args all = {TS_RUN_ALL, 0, 1, NULL, 0, 0, 0};
    return ts_box_run(redshift, prev_redshift, perturbed_field_redshift, perturbed_field, source_box,
                      previous_spin_temp, ini_boxes, this_spin_temp, &all);
}

/* the shells of `rank` (descending), as c21cm_ts_box_shard_sums deals them */
int c21cm_ts_shard_shells(int n_step, int rank, int world, int *idx) {
    int n = 0;
    for (int r = n_step - 1 - rank; r >= 0; r -= world) idx[n++] = r;
    return n;
}

/* 1: this call can be sharded (Eulerian table models on device arrays, below Z_HEAT_MAX, no
 * mini-halos) */
int c21cm_ts_shardable(float redshift, const PerturbedField *pf, const TsBox *prev, const TsBox *out) {
    if (!simulation_options_global || !astro_options_global || !matter_options_global) return 0;
    if (redshift >= simulation_options_global->Z_HEAT_MAX || astro_options_global->USE_MINI_HALOS) return 0;
    const int model = matter_options_global->SOURCE_MODEL;
    if (!(model == C21CM_SOURCE_E_INTEGRAL || model == C21CM_SOURCE_CONST_ION_EFF)) return 0;
    return pf && pf->density && prev && prev->xray_ionised_fraction && out && out->spin_temperature &&
           c21hip_is_device_ptr(pf->density) && c21hip_is_device_ptr(prev->xray_ionised_fraction) &&
           c21hip_is_device_ptr(prev->spin_temperature) && c21hip_is_device_ptr(prev->kinetic_temp_neutral) &&
           c21hip_is_device_ptr(out->spin_temperature) && c21hip_is_device_ptr(out->kinetic_temp_neutral) &&
           c21hip_is_device_ptr(out->xray_ionised_fraction);
}

/* phase 1 of rank `rank` of `world`: partial sums of its shells into sums_dev ([6][N] doubles on the
 * device; *n_rows of them in use) */
int c21cm_ts_box_shard_sums(float redshift, float prev_redshift, float perturbed_field_redshift,
                            PerturbedField *perturbed_field, TsBox *previous_spin_temp, int rank,
                            int world, double *sums_dev, int *n_rows) {
    ts_shard_args sh = {TS_RUN_SHARD_SUMS, rank, world, sums_dev, 0, 0, 0};
    TsBox out = *previous_spin_temp; /* not written in this phase */
    int st = ts_box_run(redshift, prev_redshift, perturbed_field_redshift, perturbed_field, NULL,
                        previous_spin_temp, NULL, &out, &sh);
    if (n_rows) *n_rows = sh.n_rows;
    return st;
}

/* phase 2: the temperature update of cells [cell0, cell0 + ncell) from their complete sums
 * ([n_rows][ncell] doubles on the device, rows as phase 1 wrote them) */
int c21cm_ts_box_shard_finish(float redshift, float prev_redshift, float perturbed_field_redshift,
                              PerturbedField *perturbed_field, TsBox *previous_spin_temp,
                              const double *slab_sums, size_t cell0, size_t ncell,
                              TsBox *this_spin_temp) {
    ts_shard_args sh = {TS_RUN_SHARD_FINISH, 0, 1, (double *)slab_sums, cell0, ncell, 0};
    return ts_box_run(redshift, prev_redshift, perturbed_field_redshift, perturbed_field, NULL,
                      previous_spin_temp, NULL, this_spin_temp, &sh);
}

static int ts_box_run(float redshift, float prev_redshift, float perturbed_field_redshift,
                      PerturbedField *perturbed_field, XraySourceBox *source_box,
                      TsBox *previous_spin_temp, InitialConditions *ini_boxes, TsBox *this_spin_temp,
                      ts_shard_args *sh) {
    int st = require_globals("ComputeTsBox", 1);
    if (st) return st;
    if (!perturbed_field || !perturbed_field->density || !this_spin_temp) {
        c21hip_set_error("ComputeTsBox: the perturbed field and the output box are required");
        return C21CM_VALUE_ERROR;
    }
    if ((st = ensure_ps())) return st;
    const SimulationOptions *so = simulation_options_global;
    const AstroOptions *ao = astro_options_global;
    int dim, dim_z, hii, hii_z;
    double box_len, box_len_z;
    geometry(&dim, &dim_z, &hii, &hii_z, &box_len, &box_len_z);
    const size_t ntot = (size_t)hii * hii * hii_z;

    if (redshift >= so->Z_HEAT_MAX) { /* init_first_Ts :1424-1429 */
        if ((st = c21_recfast_load())) return st;
        c21cm_ts_first_spec f;
        memset(&f, 0, sizeof(f));
        f.hii_dim = hii, f.hii_dim_z = hii_z;
        f.redshift = redshift;
        f.perturbed_redshift = perturbed_field_redshift;
        f.growth_factor_zp = dicke(redshift);
        f.inverse_growth_factor_z = 1 / dicke(perturbed_field_redshift);
        f.xe = c21_xion_RECFAST(redshift);
        f.TK = c21_T_RECFAST(redshift);
        f.cT_ad = ao->USE_ADIABATIC_FLUCTUATIONS ? c21_cT_approx(redshift) : 0.;
        f.N_b0 = c21_nb0();
        f.No = f.N_b0 * (1 - cosmo_params_global->Y_He) / (1 - 0.75 * cosmo_params_global->Y_He);
        f.A10 = 2.85e-15, f.T_21 = 0.0682, f.T_cmb = 2.7255; /* Constants.c:24-33 */
        if (ao->USE_MINI_HALOS && this_spin_temp->J_21_LW) { /* no sources yet: no LW background */
            float *j = this_spin_temp->J_21_LW;
            if (c21hip_is_device_ptr(j)) {
                if ((st = c21hip_fill(j, ntot, 0.f, NULL))) return st;
            } else {
                memset(j, 0, ntot * sizeof(float));
            }
        }
        return c21cm_ts_first_grids(&f, perturbed_field->density, this_spin_temp, NULL);
    }
    if (!previous_spin_temp || !previous_spin_temp->xray_ionised_fraction) {
        c21hip_set_error("ComputeTsBox: below Z_HEAT_MAX the previous spin-temperature box is needed");
        return C21CM_VALUE_ERROR;
    }
    double x_e_ave_p;
    if ((st = box_mean(previous_spin_temp->xray_ionised_fraction, ntot, &x_e_ave_p))) return st;

    c21cm_ts_spec *spec = (c21cm_ts_spec *)calloc(1, sizeof(*spec));
    c21_ts_tables *tab = (c21_ts_tables *)calloc(1, sizeof(*tab));
    if (!spec || !tab) {
        free(spec), free(tab);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    const float *filtered = NULL;
    const int timing = getenv("C21CM_TS_TIMING") != NULL; /* stage wall times to stderr */
    double t_mark = timing ? wall_seconds() : 0., t_prep = 0., t_tables = 0.;
    if (sh->mode == TS_RUN_SHARD_FINISH &&
        ts_prep_load(redshift, prev_redshift, perturbed_field_redshift, x_e_ave_p, spec, tab)) {
        /* phase 1 of this call prepared the snapshot in this process */
        st = c21cm_ts_cells_from_sums(spec, perturbed_field->density, previous_spin_temp, sh->sums, sh->cell0,
                                      sh->ncell, this_spin_temp, NULL);
        if (!st) this_spin_temp->Q_HI = tab->Q_HI;
        goto done;
    }
    if ((st = c21_ts_prepare_shells(redshift, prev_redshift, perturbed_field_redshift, spec, tab)))
        goto done;
    if (sh->mode == TS_RUN_SHARD_FINISH) {
        /* the slab's temperature update from the ranks' combined sums: host tables only -- and of those the
         * global ones (Q_HI, NO_LIGHT, the mean SFRDs): the frequency integrals went into the sums (round 6: an
         * empty shell mask; they were 4-12 ms of a phase whose device work is 0.7 ms) */
        unsigned char no_shells[C21CM_MAX_TS_RADII];
        memset(no_shells, 0, sizeof(no_shells));
        tab->shell_mask = no_shells;
        st = c21_ts_prepare_tables(x_e_ave_p, spec, tab);
        tab->shell_mask = NULL;
        if (st) goto done;
        st = c21cm_ts_cells_from_sums(spec, perturbed_field->density, previous_spin_temp, sh->sums,
                                      sh->cell0, sh->ncell, this_spin_temp, NULL);
        if (!st) this_spin_temp->Q_HI = tab->Q_HI;
        goto done;
    }
    if (spec->source_mode != C21CM_TS_SRC_GRIDS) {
        /* prepare_filter_boxes + fill_Rbox_table (:1453-1463): delNL0[R] stays on the device.  The
         * loop needs only the shells' radii, so it runs on a helper thread while this one builds
         * the global tables and the frequency integrals (host work of about the same length at
         * 256^3).  Before anything has formed (NO_LIGHT) its result is simply not used. */
        struct filter_job job;
        memset(&job, 0, sizeof(job));
        c21cm_rbox_spec *r = &job.spec;
        r->hii_dim = hii, r->hii_dim_z = hii_z, r->box_len = box_len, r->box_len_z = box_len_z;
        r->filter_type = ao->HEAT_FILTER;
        r->n_R = tab->n_step;
        for (int i = 0; i < tab->n_step; i++) r->R[i] = tab->R_values[i];
        int local[C21CM_MAX_TS_RADII], n_local = tab->n_step;
        if (sh->mode == TS_RUN_SHARD_SUMS) { /* this rank's shells only, in ascending radius */
            int desc[C21CM_MAX_TS_RADII];
            n_local = c21cm_ts_shard_shells(tab->n_step, sh->rank, sh->world, desc);
            for (int i = 0; i < n_local; i++) local[i] = desc[n_local - 1 - i];
            r->n_R = n_local;
            for (int i = 0; i < n_local; i++) r->R[i] = tab->R_values[local[i]];
        }
        r->cell_radius = 0.620350491 * so->BOX_LEN / (float)so->HII_DIM; /* physconst.l_factor */
        r->min_value = -1;
        r->const_factor = 1. / dicke(perturbed_field_redshift);
        job.input = perturbed_field->density;
        job.result = (float *)c21hip_ws(170, (size_t)tab->n_step * ntot * sizeof(float));
        if (!job.result) {
            c21hip_set_error("ComputeTsBox: out of device memory for %d filtered density grids", tab->n_step);
            st = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        if (spec->use_mini_halos) {
            /* prepare_filter_boxes + fill_Rbox_table of log10 M_crit,LW (:535-565,1459-1466): the
             * turnover grid from the previous box's J_21_LW, filtered at every shell radius and
             * floored at the threshold without a background; its box means feed the global tables,
             * so this loop runs before them */
            if (!previous_spin_temp->J_21_LW || !this_spin_temp->J_21_LW ||
                (matter_options_global->V_CB_MODEL == C21CM_VCB_FLUCTS &&
                 (!ini_boxes || !ini_boxes->lowres_vcb))) {
                c21hip_set_error("ComputeTsBox: USE_MINI_HALOS needs J_21_LW in the previous and the new "
                                 "box and, with V_CB_MODEL = FLUCTS, lowres_vcb");
                st = C21CM_VALUE_ERROR;
                goto done;
            }
            c21_scaling_consts sc;
            if ((st = c21_set_scaling_constants(redshift, &sc))) goto done;
            c21cm_mturn_spec ms;
            memset(&ms, 0, sizeof(ms));
            ms.hii_dim = hii, ms.hii_dim_z = hii_z;
            ms.redshift = redshift;
            ms.vcb_const = sc.vcb_const;
            ms.A_LW = astro_params_global->A_LW, ms.BETA_LW = astro_params_global->BETA_LW;
            ms.A_VCB = astro_params_global->A_VCB, ms.BETA_VCB = astro_params_global->BETA_VCB;
            ms.sigma_vcb = cosmo_tables_global->V_CB_AVG * sqrt(3 * M_PI / 8);
            float *mcrit = (float *)c21hip_ws(172, ntot * sizeof(float));
            float *mcrit_R = (float *)c21hip_ws(173, (size_t)tab->n_step * ntot * sizeof(float));
            if (!mcrit || !mcrit_R) {
                st = C21CM_MEMORY_ALLOC_ERROR;
                goto done;
            }
            if ((st = c21cm_ts_mcrit_grid(&ms, astro_params_global->M_TURN, previous_spin_temp->J_21_LW,
                                          matter_options_global->V_CB_MODEL == C21CM_VCB_FLUCTS
                                              ? ini_boxes->lowres_vcb : NULL,
                                          mcrit, NULL)))
                goto done;
            c21cm_rbox_spec rm = job.spec;
            /* (MINIMIZE_MEMORY filters one shell at a time upstream, with a floor of 0: :1588-1594) */
            rm.min_value = matter_options_global->MINIMIZE_MEMORY
                               ? 0.
                               : log10(c21_lyman_werner_threshold(redshift, 0.f, 0.f));
            rm.const_factor = 1.;
            double mn[C21CM_MAX_TS_RADII], mx[C21CM_MAX_TS_RADII];
            if ((st = c21cm_fill_Rbox_grids(&rm, mcrit, mcrit_R, mn, tab->ave_log10_mturn, mx, NULL)))
                goto done;
            spec->filtered_log10_mcrit = mcrit_R;
        }
        unsigned char shell_mask[C21CM_MAX_TS_RADII];
        if (sh->mode == TS_RUN_SHARD_SUMS) { /* the frequency integrals of this rank's shells only */
            memset(shell_mask, 0, sizeof(shell_mask));
            for (int i = 0; i < n_local; i++) shell_mask[local[i]] = 1;
            tab->shell_mask = shell_mask;
        }
        job.device = c21hip_current_device();
        pthread_t worker;
        const int threaded = pthread_create(&worker, NULL, filter_job_run, &job) == 0;
        if (!threaded) (void)filter_job_run(&job); /* no thread: one after the other */
        st = c21_ts_prepare_tables(x_e_ave_p, spec, tab);
        if (timing) t_prep = wall_seconds() - t_mark;
        tab->shell_mask = NULL; /* (points into this frame) */
        if (threaded) pthread_join(worker, NULL);
        if (st) goto done;
        if ((st = job.status)) goto done;
        if (timing) t_mark = wall_seconds();
        if (sh->mode == TS_RUN_SHARD_SUMS) {
            ts_prep_store(redshift, prev_redshift, perturbed_field_redshift, x_e_ave_p, spec, tab);
            sh->n_rows = spec->use_lya_heating ? 6 : 4;
            if (spec->no_light) { /* nothing has formed: every partial sum is zero */
                st = c21hip_memset(sh->sums, 0, (size_t)sh->n_rows * ntot * sizeof(double), NULL);
                if (!st) st = c21hip_sync(NULL);
                goto done;
            }
            /* the per-shell members of spec / tab compacted to this rank's shells */
            double *fq = (double *)malloc(3 * (size_t)C21CM_X_INT_NXHII * n_local * sizeof(double));
            if (!fq) {
                st = C21CM_MEMORY_ALLOC_ERROR;
                goto done;
            }
            const int nf = tab->n_step;
            for (int k = 0; k < 3; k++) {
                const double *src = k == 0 ? spec->freq_int_heat : (k == 1 ? spec->freq_int_ion : spec->freq_int_lya);
                for (int m = 0; m < C21CM_X_INT_NXHII; m++)
                    for (int i = 0; i < n_local; i++)
                        fq[((size_t)k * C21CM_X_INT_NXHII + m) * n_local + i] = src[(size_t)m * nf + local[i]];
            }
#define GATHER(arr)                                                \
    do {                                                           \
        double tmp_[C21CM_MAX_TS_RADII];                           \
        for (int i = 0; i < n_local; i++) tmp_[i] = arr[local[i]]; \
        for (int i = 0; i < n_local; i++) arr[i] = tmp_[i];        \
    } while (0)
            GATHER(spec->z_edge_factor); GATHER(spec->xray_R_factor); GATHER(spec->starlya_prefactor);
            GATHER(spec->lya_cont_prefactor); GATHER(spec->lya_inj_prefactor); GATHER(spec->zpp_growth);
            GATHER(spec->mean_sfr_zpp);
            GATHER(tab->zpp); GATHER(tab->zpp_growth); GATHER(tab->M_min_R); GATHER(tab->M_max_R);
            GATHER(tab->sigma_min); GATHER(tab->sigma_max); GATHER(tab->R_values);
#undef GATHER
            free(tab->freq); /* spec's three pointers pointed into it */
            tab->freq = fq;
            spec->freq_int_heat = fq;
            spec->freq_int_ion = fq + (size_t)C21CM_X_INT_NXHII * n_local;
            spec->freq_int_lya = fq + 2 * (size_t)C21CM_X_INT_NXHII * n_local;
            spec->n_step = tab->n_step = n_local;
        }
        if (!spec->no_light) {
            if (spec->source_mode == C21CM_TS_SRC_SFRD_TABLE)
                st = c21_ts_sfrd_tables(job.mn, job.mx, spec, tab);
            else
                st = c21_ts_fcoll_tables(job.mn, job.mx, spec, tab);
            if (st) goto done;
            filtered = job.result;
        }
        if (sh->mode == TS_RUN_SHARD_SUMS) {
            st = c21cm_ts_shell_sums(spec, perturbed_field->density, previous_spin_temp, NULL, filtered,
                                     sh->sums, NULL);
            goto done;
        }
        if (timing) t_tables = wall_seconds() - t_mark, t_mark = wall_seconds();
    } else {
        if (spec->use_mini_halos) { /* :1479-1483: the shells' mean turnover masses come with the grids */
            if (!source_box || !source_box->mean_log10_Mcrit_LW || !source_box->filtered_sfr_mini ||
                !this_spin_temp->J_21_LW) {
                c21hip_set_error("ComputeTsBox: USE_MINI_HALOS with source grids needs "
                                 "XraySourceBox.filtered_sfr_mini, mean_log10_Mcrit_LW and TsBox.J_21_LW");
                st = C21CM_VALUE_ERROR;
                goto done;
            }
            for (int i = 0; i < tab->n_step; i++)
                tab->ave_log10_mturn[i] = source_box->mean_log10_Mcrit_LW[i];
        }
        if ((st = c21_ts_prepare_tables(x_e_ave_p, spec, tab))) goto done;
        if (timing) t_prep = wall_seconds() - t_mark, t_mark = wall_seconds();
    }
    st = c21cm_ts_grids(spec, perturbed_field->density, previous_spin_temp, source_box, filtered,
                        this_spin_temp, NULL, NULL);
    if (!st) this_spin_temp->Q_HI = tab->Q_HI;
    if (timing)
        fprintf(stderr, "ComputeTsBox z=%.3f: host tables beside the density filter loop %.1f ms, SFRD "
                        "tables %.1f ms, cell sweeps (incl. staging) %.1f ms\n",
                redshift, 1e3 * t_prep, 1e3 * t_tables, 1e3 * (wall_seconds() - t_mark));
done:
    c21_ts_tables_free(tab);
    free(tab);
    free(spec);
    return st;
}

/* the ScalingConstants read by set_halo_properties (scaling_relations.c:29-118) */
static void fill_halo_consts(double redshift, const c21_scaling_consts *sc, c21cm_halo_consts *hc) {
    const AstroParams *ap = astro_params_global;
    const AstroOptions *ao = astro_options_global;
    memset(hc, 0, sizeof(*hc));
    hc->redshift = redshift;
    hc->fstar_10 = sc->fstar_10, hc->alpha_star = sc->alpha_star, hc->sigma_star = ap->SIGMA_STAR;
    hc->alpha_upper = ap->UPPER_STELLAR_TURNOVER_INDEX, hc->pivot_upper = ap->UPPER_STELLAR_TURNOVER_MASS;
    hc->upper_pivot_ratio = pow(hc->pivot_upper / 1e10, hc->alpha_star) + pow(hc->pivot_upper / 1e10, hc->alpha_upper);
    hc->fstar_7 = sc->fstar_7, hc->alpha_star_mini = sc->alpha_star_mini, hc->acg_thresh = sc->acg_thresh;
    hc->baryon_ratio = cosmo_params_global->OMb / cosmo_params_global->OMm;
    hc->t_h = sc->t_h, hc->t_star = sc->t_star;
    hc->sigma_sfr_lim = ap->SIGMA_SFR_LIM, hc->sigma_sfr_idx = ap->SIGMA_SFR_INDEX;
    hc->l_x = sc->l_x, hc->l_x_mini = sc->l_x_mini, hc->sigma_xray = ap->SIGMA_LX;
    hc->fesc_10 = sc->fesc_10, hc->fesc_7 = sc->fesc_7, hc->alpha_esc = sc->alpha_esc;
    hc->pop2_ion = sc->pop2_ion, hc->pop3_ion = sc->pop3_ion;
    hc->mturn_a_nofb = sc->mturn_a_nofb, hc->mturn_m_nofb = sc->mturn_m_nofb;
    hc->scaling_median = ao->HALO_SCALING_RELATIONS_MEDIAN;
    hc->upper_stellar_turnover = ao->USE_UPPER_STELLAR_TURNOVER;
    hc->use_mini_halos = ao->USE_MINI_HALOS, hc->use_xray = ao->USE_TS_FLUCT;
}

/* reference: src/py21cmfast/src/HaloBox.c:658-779 (_functionprototypes_wrapper.h:127-130): the twelve
 * properties of every halo (mass, M*, SFR, L_X, n_ion, f_esc-weighted SFR, M*_mini, SFR_mini, the
 * three turnover masses, metallicity); halos of zero mass are left untouched */
int test_halo_props(double redshift, float *vcb_grid, float *J21_LW_grid, float *z_re_grid,
                    float *Gamma12_ion_grid, unsigned long long n_halos, float *halo_masses,
                    float *halo_coords, float *star_rng, float *sfr_rng, float *xray_rng,
                    float *halo_props_out) {
    int st = require_globals("test_halo_props", 1);
    if (st) return st;
    if (!n_halos) return 0;
    const SimulationOptions *so = simulation_options_global;
    const AstroOptions *ao = astro_options_global;
    const int mini = ao->USE_MINI_HALOS, below = redshift < so->Z_HEAT_MAX;
    const int flucts = matter_options_global->V_CB_MODEL == C21CM_VCB_FLUCTS;
    if (!halo_masses || !halo_coords || !star_rng || !sfr_rng || !xray_rng || !halo_props_out ||
        (mini && ((flucts && !vcb_grid) || (below && (!J21_LW_grid || !z_re_grid || !Gamma12_ion_grid))))) {
        c21hip_set_error("test_halo_props: NULL catalogue array or feedback grid");
        return C21CM_VALUE_ERROR;
    }
    if ((st = ensure_ps())) return st;
    c21_scaling_consts sc;
    if ((st = c21_set_scaling_constants(redshift, &sc))) return st;
    c21cm_halo_consts hc;
    fill_halo_consts(redshift, &sc, &hc);
    int dim, dim_z, hii[3];
    double bl, blz;
    geometry(&dim, &dim_z, &hii[0], &hii[2], &bl, &blz);
    hii[1] = hii[0];
    const size_t nh = (size_t)n_halos, fb = nh * sizeof(float);
    const size_t gb = (size_t)hii[0] * hii[1] * hii[2] * sizeof(float);
    enum { WS_HP0 = 238 };
    const float *in[5] = {halo_masses, halo_coords, star_rng, sfr_rng, xray_rng};
    const float *grid_in[4] = {vcb_grid, J21_LW_grid, z_re_grid, Gamma12_ion_grid};
    const float *dev[5], *gdev[4] = {NULL, NULL, NULL, NULL};
    for (int k = 0; k < 5; k++) {
        dev[k] = in[k];
        if (!c21hip_is_device_ptr(in[k])) {
            void *d = c21hip_ws(WS_HP0 + k, k == 1 ? 3 * fb : fb);
            if (!d) return C21CM_MEMORY_ALLOC_ERROR;
            if ((st = c21hip_h2d(d, in[k], k == 1 ? 3 * fb : fb, NULL))) return st;
            dev[k] = (const float *)d;
        }
    }
    for (int k = 0; k < 4 && mini; k++) {
        if (!grid_in[k] || (k == 0 ? !flucts : !below)) continue;
        gdev[k] = grid_in[k];
        if (!c21hip_is_device_ptr(grid_in[k])) {
            void *d = c21hip_ws(WS_HP0 + 5 + k, gb);
            if (!d) return C21CM_MEMORY_ALLOC_ERROR;
            if ((st = c21hip_h2d(d, grid_in[k], gb, NULL))) return st;
            gdev[k] = (const float *)d;
        }
    }
    float *out = halo_props_out;
    if (!c21hip_is_device_ptr(out)) {
        if (!(out = (float *)c21hip_ws(WS_HP0 + 9, 12 * fb))) return C21CM_MEMORY_ALLOC_ERROR;
        if ((st = c21hip_h2d(out, halo_props_out, 12 * fb, NULL))) return st; /* cut halos keep their rows */
    }
    const double lw[7] = {astro_params_global->A_LW, astro_params_global->BETA_LW, astro_params_global->A_VCB,
                          astro_params_global->BETA_VCB, cosmo_tables_global->V_CB_AVG * sqrt(3 * M_PI / 8),
                          sc.vcb_const, astro_params_global->M_TURN};
    if ((st = c21hip_halo_props(&hc, n_halos, dev[0], dev[1], dev[2], dev[3], dev[4], hii, bl / hii[0],
                                redshift, below, flucts, lw, gdev[0], gdev[1], gdev[2], gdev[3], out, NULL)))
        return st;
    if (out != halo_props_out && (st = c21hip_d2h(halo_props_out, out, 12 * fb, NULL))) return st;
    return c21hip_sync(NULL);
}

/* reference: src/py21cmfast/src/HaloBox.c:563-653 with set_fixed_grids :302-436 and
 * sum_halos_onto_grid :518-560, without the extra fields; with USE_TS_FLUCT the X-ray
 * emissivity grid halo_xray is filled as well (the input of UpdateXraySourceBox); with USE_MINI_HALOS
 * the turnover grids (get_log10_turnovers :465-516), the 2-D tables and halo_sfr_mini. */
int ComputeHaloBox(double redshift, InitialConditions *ini_boxes, HaloCatalog *halos,
                   TsBox *previous_spin_temp, IonizedBox *previous_ionize_box, HaloBox *grids) {
    int st = require_globals("ComputeHaloBox", 1);
    if (st) return st;
    if (!ini_boxes || !grids) return C21CM_VALUE_ERROR;
    const SimulationOptions *so = simulation_options_global;
    const MatterOptions *mo = matter_options_global;
    const AstroOptions *ao = astro_options_global;
    const char *unsupported = NULL;
    const int sampled = mo->SOURCE_MODEL == C21CM_SOURCE_DEXM_ESF ||
                        mo->SOURCE_MODEL == C21CM_SOURCE_CHMF_SAMPLER; /* InputParameters.h:76-79 */
    if (mo->SOURCE_MODEL != C21CM_SOURCE_L_INTEGRAL && !sampled)
        unsupported = "an Eulerian SOURCE_MODEL (no HaloBox)";
    if (sampled && !halos) {
        c21hip_set_error("ComputeHaloBox: SOURCE_MODEL = DEXM-ESF / CHMF-SAMPLER needs a halo catalogue");
        return C21CM_VALUE_ERROR;
    }
    if (mo->USE_INTERPOLATION_TABLES != C21CM_INTERP_HMF)
        unsupported = "L-INTEGRAL without USE_INTERPOLATION_TABLES=hmf-interpolation";
    if (ao->USE_MINI_HALOS && mo->PERTURB_ON_HIGH_RES)
        unsupported = "USE_MINI_HALOS with PERTURB_ON_HIGH_RES (upstream indexes the low-resolution "
                      "turnover grids with the high-resolution cell index, map_mass.c:291-292)";
    if (ao->USE_MINI_HALOS && ao->INTEGRATION_METHOD_MINI > 1)
        unsupported = "INTEGRATION_METHOD_MINI=GAMMA-APPROX";
    if (ao->INTEGRATION_METHOD_ATOMIC > 1) unsupported = "INTEGRATION_METHOD_ATOMIC=GAMMA-APPROX";
    if (ao->PHOTON_CONS_TYPE != C21CM_PHOTONCONS_NONE) unsupported = "PHOTON_CONS_TYPE != none";
    if (config_settings.EXTRA_HALOBOX_FIELDS) unsupported = "EXTRA_HALOBOX_FIELDS";
    if (mo->HMF != C21CM_HMF_PS && mo->HMF != C21CM_HMF_ST)
        unsupported = "an HMF without conditional mass function (mean-fixed grids)";
    if (unsupported) {
        c21hip_set_error("ComputeHaloBox: %s is not implemented in this backend yet", unsupported);
        return C21CM_VALUE_ERROR;
    }
    if ((st = ensure_ps())) return st;

    c21cm_halobox_spec s;
    memset(&s, 0, sizeof(s));
    int dim, dim_z;
    geometry(&dim, &dim_z, &s.hii_dim, &s.hii_dim_z, &s.box_len, &s.box_len_z);
    s.dim = dim;
    s.dim_z = dim_z;
    s.perturb_on_high_res = mo->PERTURB_ON_HIGH_RES;
    s.lpt2 = (mo->PERTURB_ALGORITHM == C21CM_PERTURB_2LPT);
    s.growth_factor = dicke(redshift);
    s.init_growth_factor = dicke(so->INITIAL_REDSHIFT);

    c21_scaling_consts sc, sc_sfrd;
    if ((st = c21_set_scaling_constants(redshift, &sc))) return st;
    /* the halos of a catalogue take the unadjusted constants (hbox_consts, :599-600,623-624) */
    c21cm_halo_consts hc;
    fill_halo_consts(redshift, &sc, &hc);
    /* set_fixed_grids :300-308: median relations -> raised normalisations in the sub-grid integrals */
    if (ao->HALO_SCALING_RELATIONS_MEDIAN && (st = c21_scaling_consts_mimic_scatter(&sc))) return st;
    sc_sfrd = c21_scaling_consts_sfr(&sc); /* scaling_relations.c:122-131 */

    /* the integrated part ends where the catalogue begins (:626-634) */
    const double M_min = c21_minimum_source_mass(redshift);
    const double M_max = mo->SOURCE_MODEL == C21CM_SOURCE_CHMF_SAMPLER ? so->SAMPLER_MIN_MASS
                         : mo->SOURCE_MODEL == C21CM_SOURCE_DEXM_ESF
                             ? c21_RtoM(L_FACTOR * so->BOX_LEN / so->DIM)
                             : M_MAX_INTEGRAL;
    const int integral = M_min < M_max; /* :635 */
    if (sampled) s.halos = halos, s.halo_consts = &hc, s.skip_integral = !integral;
    const size_t n_src = s.perturb_on_high_res ? (size_t)dim * dim * dim_z
                                                : (size_t)s.hii_dim * s.hii_dim * s.hii_dim_z;
    const size_t n_out = (size_t)s.hii_dim * s.hii_dim * s.hii_dim_z;
    const double volume = s.box_len * s.box_len * s.box_len_z;
    const double M_cell = c21_rhocrit() * cosmo_params_global->OMm * volume / (double)n_src;
    const double sigma_cell = sigma_z0(M_cell); /* HaloBox.c:45 */
    const float *dens = s.perturb_on_high_res ? ini_boxes->hires_density : ini_boxes->lowres_density;
    if (!dens) {
        c21hip_set_error("ComputeHaloBox: the InitialConditions density grid is missing");
        return C21CM_VALUE_ERROR;
    }
    /* table range: extrema of density * D, seeded with 0, widened by 0.1 % (HaloBox.c:303-381) */
    double mm[2] = {0., 0.};
    if (integral && (st = c21cm_grid_minmax(dens, n_src, mm, NULL))) return st;
    double min_density = fmin(0., mm[0] * s.growth_factor) * 1.001;
    double max_density = fmax(0., mm[1] * s.growth_factor) * 1.001;
    if (!(max_density > min_density)) max_density = min_density + 1e-6;
    static float tab_nion[C21CM_NDELTA_TABLE], tab_sfrd[C21CM_NDELTA_TABLE];
    const int method = ao->INTEGRATION_METHOD_ATOMIC;
    if (integral &&
        (st = c21_Nion_Conditional_table(s.growth_factor, log(M_min), log(M_max), log(M_cell),
                                         sigma_cell, min_density, max_density, sc.mturn_a_nofb, &sc,
                                         method, -40., tab_nion, C21CM_NDELTA_TABLE)))
        return st;
    if (integral &&
        (st = c21_Nion_Conditional_table(s.growth_factor, log(M_min), log(M_max), log(M_cell),
                                         sigma_cell, min_density, max_density, sc_sfrd.mturn_a_nofb,
                                         &sc_sfrd, method, -50., tab_sfrd, C21CM_NDELTA_TABLE)))
        return st;
    s.tab_min = min_density;
    s.tab_width = (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.);
    s.ln_nion_table = tab_nion;
    s.ln_sfrd_table = tab_sfrd;
    if (ao->USE_TS_FLUCT) { /* HaloBox.c:411-414, interp_tables.c:497-560 */
        static float tab_xray[C21CM_NDELTA_TABLE];
        if (!grids->halo_xray) {
            c21hip_set_error("ComputeHaloBox: USE_TS_FLUCT needs HaloBox.halo_xray");
            return C21CM_VALUE_ERROR;
        }
        if (integral &&
            (st = c21_Xray_Conditional_table(s.growth_factor, log(M_min), log(M_max), log(M_cell),
                                             sigma_cell, min_density, max_density, sc.mturn_a_nofb,
                                             &sc, method, tab_xray, C21CM_NDELTA_TABLE)))
            return st;
        if (integral) s.ln_xray_table = tab_xray;
    }
    /* map_mass.c:223-239 */
    const double vol_ratio_out = (double)n_out / (double)n_src;
    const double prefactor_stars = c21_rhocrit() * cosmo_params_global->OMb * sc.fstar_10 * vol_ratio_out;
    s.prefactor_sfr = prefactor_stars / sc.t_star / sc.t_h;
    s.prefactor_nion = prefactor_stars * sc.fesc_10 * sc.pop2_ion;
    s.prefactor_wsfr = 1 / sc.t_h / sc.t_star;
    s.prefactor_xray = c21_rhocrit() * cosmo_params_global->OMm * vol_ratio_out;
    /* get_log10_turnovers without mini-halos (HaloBox.c:467-470) */
    grids->log10_Mcrit_ACG_ave = log10(sc.mturn_a_nofb);
    grids->log10_Mcrit_MCG_ave = log10(sc.mturn_m_nofb);
    static float *tab2[4]; /* N_ion (ACG, MCG), SFRD_MINI, X-ray: [NDELTA + 1][NMTURN] each */
    if (ao->USE_MINI_HALOS) {
        const int below = redshift < so->Z_HEAT_MAX; /* HaloBox.c:488-492 */
        if (!grids->halo_sfr_mini ||
            (below && (!previous_spin_temp || !previous_spin_temp->J_21_LW || !previous_ionize_box ||
                       !previous_ionize_box->ionisation_rate_G12 || !previous_ionize_box->z_reion)) ||
            (mo->V_CB_MODEL == C21CM_VCB_FLUCTS && !ini_boxes->lowres_vcb)) {
            c21hip_set_error("ComputeHaloBox: USE_MINI_HALOS needs HaloBox.halo_sfr_mini and, below "
                             "Z_HEAT_MAX, the previous TsBox.J_21_LW and IonizedBox Gamma_12 / z_reion "
                             "(with V_CB_MODEL = FLUCTS also lowres_vcb)");
            return C21CM_VALUE_ERROR;
        }
        float *mta = (float *)c21hip_ws(229, n_out * sizeof(float));
        float *mtm = (float *)c21hip_ws(230, n_out * sizeof(float));
        if (!mta || !mtm) return C21CM_MEMORY_ALLOC_ERROR;
        c21cm_mturn_spec ms;
        memset(&ms, 0, sizeof(ms));
        ms.hii_dim = s.hii_dim, ms.hii_dim_z = s.hii_dim_z;
        ms.redshift = redshift;
        ms.mturn_a_nofb = sc.mturn_a_nofb;
        ms.vcb_const = sc.vcb_const;
        ms.A_LW = astro_params_global->A_LW, ms.BETA_LW = astro_params_global->BETA_LW;
        ms.A_VCB = astro_params_global->A_VCB, ms.BETA_VCB = astro_params_global->BETA_VCB;
        ms.sigma_vcb = cosmo_tables_global->V_CB_AVG * sqrt(3 * M_PI / 8);
        double ave[2];
        if ((st = c21cm_halobox_turnovers(
                 &ms, astro_params_global->M_TURN, below, so->N_THREADS,
                 below ? previous_ionize_box->ionisation_rate_G12 : NULL,
                 below ? previous_ionize_box->z_reion : NULL, below ? previous_spin_temp->J_21_LW : NULL,
                 mo->V_CB_MODEL == C21CM_VCB_FLUCTS ? ini_boxes->lowres_vcb : NULL, mta, mtm, ave, NULL)))
            return st;
        grids->log10_Mcrit_ACG_ave = ave[0];
        grids->log10_Mcrit_MCG_ave = ave[1];
        /* table ranges of the turnover grids (HaloBox.c:316-319,372-390) */
        double ra[2], rm[2];
        if ((st = c21cm_grid_minmax(mta, n_out, ra, NULL))) return st;
        if ((st = c21cm_grid_minmax(mtm, n_out, rm, NULL))) return st;
        const double l10_max_int = log10(M_MAX_INTEGRAL), l10_mturn = log10(astro_params_global->M_TURN);
        const double a_lo = fmin(l10_max_int, ra[0]) * 0.999, a_hi = fmax(l10_mturn, ra[1]) * 1.001;
        const double m_lo = fmin(l10_max_int, rm[0]) * 0.999, m_hi = fmax(l10_mturn, rm[1]) * 1.001;
        const size_t t2 = (size_t)(C21CM_NDELTA_TABLE + 1) * C21CM_NMTURN_TABLE;
        for (int k = 0; k < 4; k++)
            if (!tab2[k] && !(tab2[k] = (float *)calloc(t2, sizeof(float)))) return C21CM_MEMORY_ALLOC_ERROR;
        const double lnMmin = log(M_min), lnMmax = log(M_max), lnMc = log(M_cell);
        const int m_a = ao->INTEGRATION_METHOD_ATOMIC, m_m = ao->INTEGRATION_METHOD_MINI;
        /* initialise_Nion_Conditional_spline (interp_tables.c:291-405) on the grids' ranges */
        if (integral &&
            (st = c21_Nion_Conditional_table2d(s.growth_factor, lnMmin, lnMmax, lnMc, sigma_cell,
                                               min_density, max_density, a_lo, a_hi, &sc, 0, m_a, -40.,
                                               0, tab2[0], C21CM_NDELTA_TABLE, C21CM_NMTURN_TABLE)))
            return st;
        if (integral &&
            (st = c21_Nion_Conditional_table2d(s.growth_factor, lnMmin, lnMmax, lnMc, sigma_cell,
                                               min_density, max_density, m_lo, m_hi, &sc, 1, m_m, -40.,
                                               0, tab2[1], C21CM_NDELTA_TABLE, C21CM_NMTURN_TABLE)))
            return st;
        /* SFRD_conditional_table_MINI / Xray_conditional_table_2D on the fixed turnover grid
         * (interp_tables.c:440-475,497-560; float condition sigma and turnover masses) */
        if (integral &&
            (st = c21_Nion_Conditional_table2d(s.growth_factor, lnMmin, lnMmax, lnMc, (float)sigma_cell,
                                               min_density, max_density, C21_LOG10_MTURN_MIN,
                                               C21_LOG10_MTURN_MAX, &sc_sfrd, 1, m_m, -50., 1, tab2[2],
                                               C21CM_NDELTA_TABLE, C21CM_NMTURN_TABLE)))
            return st;
        s.use_mini_halos = 1;
        s.log10_mturn_acg = mta, s.log10_mturn_mcg = mtm;
        s.ln_nion_table2d = tab2[0], s.ln_nion_mini_table2d = tab2[1], s.ln_sfrd_mini_table2d = tab2[2];
        s.mta_min = a_lo, s.mta_width = (a_hi - a_lo) / (C21CM_NMTURN_TABLE - 1.);
        s.mtm_min = m_lo, s.mtm_width = (m_hi - m_lo) / (C21CM_NMTURN_TABLE - 1.);
        s.mt_fixed_min = C21_LOG10_MTURN_MIN;
        s.mt_fixed_width = (C21_LOG10_MTURN_MAX - C21_LOG10_MTURN_MIN) / (C21CM_NMTURN_TABLE - 1.);
        if (ao->USE_TS_FLUCT && integral) {
            if ((st = c21_Nion_Conditional_table2d(s.growth_factor, lnMmin, lnMmax, lnMc, sigma_cell,
                                                   min_density, max_density, C21_LOG10_MTURN_MIN,
                                                   C21_LOG10_MTURN_MAX, &sc, 2, m_m, -50., 1, tab2[3],
                                                   C21CM_NDELTA_TABLE, C21CM_NMTURN_TABLE)))
                return st;
            s.ln_xray_table2d = tab2[3];
        }
        const double prefactor_stars_mini =
            c21_rhocrit() * cosmo_params_global->OMb * sc.fstar_7 * vol_ratio_out; /* map_mass.c:228-237 */
        s.prefactor_sfr_mini = prefactor_stars_mini / sc.t_star / sc.t_h;
        s.prefactor_nion_mini = prefactor_stars_mini * sc.fesc_7 * sc.pop3_ion;
    }
    if (!integral && !sampled) { /* :635 -- nothing to integrate: the grids stay zero */
        c21hip_set_error("ComputeHaloBox: M_min >= M_max");
        return C21CM_VALUE_ERROR;
    }
    HaloBox g = *grids;
    if (ao->RECOMB_MODEL == C21CM_RECOMB_NONE) g.whalo_sfr = NULL; /* map_mass.c:342 */
    if (!ao->USE_TS_FLUCT) g.halo_xray = NULL;
    return c21cm_halobox_grids(&s, ini_boxes, &g, NULL);
}
