/*
 * c21cm_abi.h -- the drop-in C ABI of the MI355X-native 21cmFAST hot path.
 *
 * This header is the CONTRACT between `py21cmfast`'s CFFI layer and
 * `lib21cmfast_hip.so`.  py21cmfast feeds three cdef headers to ffi.cdef()
 * (reference: build_cffi.py:163-169) and then calls `lib.ComputeXxx(...)`
 * with cffi-allocated structs (reference: src/py21cmfast/wrapper/outputs.py:447-487).
 * A replacement backend therefore has to agree with those headers on
 *   - the memory layout of every struct that crosses the boundary
 *     (reference: src/py21cmfast/src/_inputparams_wrapper.h:6-202,
 *                 src/py21cmfast/src/_outputstructs_wrapper.h:6-100),
 *   - the exported symbol names and signatures of the hot path
 *     (reference: src/py21cmfast/src/_functionprototypes_wrapper.h:6-26,130-131),
 *   - the integer status-code convention (reference: src/py21cmfast/src/exceptions.h:12-21).
 * Layouts below are written out field by field from that contract; nothing
 * else in this repository is derived from reference source text.
 *
 * Grid conventions (reference: src/py21cmfast/src/indexing.h:84-100): all grids
 * are C-contiguous float32 `[x][y][z]` with z fastest.  Arrays are OWNED BY THE
 * CALLER; a pointer may address host memory (numpy, the py21cmfast case) or
 * MI355X HBM (a torch tensor): the library asks the HIP runtime which it is
 * and only stages host arrays.  Optional arrays that are unused for the
 * selected flags may be NULL.
 */
#ifndef C21CM_ABI_H
#define C21CM_ABI_H

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Status codes returned by every Compute* entry point.               */
/* reference: src/py21cmfast/src/exceptions.h:12-21                   */
/* ------------------------------------------------------------------ */
enum c21cm_status {
    C21CM_OK = 0,
    C21CM_IO_ERROR = 1,
    C21CM_GSL_ERROR = 2,
    C21CM_VALUE_ERROR = 3,
    C21CM_PHOTONCONS_ERROR = 4,
    C21CM_TABLE_GENERATION_ERROR = 5,
    C21CM_TABLE_EVALUATION_ERROR = 6,
    C21CM_INFINITY_OR_NAN_ERROR = 7,
    C21CM_MASSDEPZETA_ERROR = 8,
    C21CM_MEMORY_ALLOC_ERROR = 9
};

/* ------------------------------------------------------------------ */
/* Enumerated option values (the Python side sends the index of the   */
/* chosen string; reference: src/py21cmfast/src/InputParameters.h:10-58). */
/* ------------------------------------------------------------------ */
enum { C21CM_HMF_PS = 0, C21CM_HMF_ST = 1, C21CM_HMF_WATSON = 2, C21CM_HMF_WATSON_Z = 3,
       C21CM_HMF_DELOS = 4 };
enum { C21CM_PS_EH = 0, C21CM_PS_BBKS = 1, C21CM_PS_EFSTATHIOU = 2, C21CM_PS_PEEBLES = 3,
       C21CM_PS_WHITE = 4, C21CM_PS_CLASS = 5 };
enum { C21CM_INTERP_NONE = 0, C21CM_INTERP_SIGMA = 1, C21CM_INTERP_HMF = 2 };
enum { C21CM_FILTER_TOPHAT = 0, C21CM_FILTER_SHARPK = 1, C21CM_FILTER_GAUSSIAN = 2,
       C21CM_FILTER_EXP_MFP = 3, C21CM_FILTER_SHELL = 4, C21CM_FILTER_MULTISCATTER = 5 };
enum { C21CM_PERTURB_LINEAR = 0, C21CM_PERTURB_ZELDOVICH = 1, C21CM_PERTURB_2LPT = 2 };
enum { C21CM_SOURCE_CONST_ION_EFF = 0, C21CM_SOURCE_E_INTEGRAL = 1, C21CM_SOURCE_L_INTEGRAL = 2,
       C21CM_SOURCE_DEXM_ESF = 3, C21CM_SOURCE_CHMF_SAMPLER = 4 };
enum { C21CM_PHOTONCONS_NONE = 0, C21CM_PHOTONCONS_Z = 1, C21CM_PHOTONCONS_ALPHA = 2,
       C21CM_PHOTONCONS_F = 3 };
enum { C21CM_INTEG_QAG = 0, C21CM_INTEG_GAUSS_LEGENDRE = 1, C21CM_INTEG_GAMMA_APPROX = 2 };
enum { C21CM_RECOMB_NONE = 0, C21CM_RECOMB_HOMOGENEOUS = 1, C21CM_RECOMB_INHOMOGENEOUS = 2 };
enum { C21CM_VCB_NONE = 0, C21CM_VCB_AVG_AUTO = 1, C21CM_VCB_FLUCTS = 2, C21CM_VCB_AVG_DEBUG = 3 };

/* ------------------------------------------------------------------ */
/* Input parameter structs                                            */
/* reference: src/py21cmfast/src/_inputparams_wrapper.h:6-202         */
/* ------------------------------------------------------------------ */
typedef int hmf_model_t;
typedef int source_model_t;
typedef int v_cb_model_t;
typedef int integration_method_t;

typedef struct CosmoParams { /* _inputparams_wrapper.h:11-26 */
    float hlittle;
    float OMm;
    float OMl;
    float OMb;
    float POWER_INDEX;
    float OMn;
    float OMk;
    float OMr;
    float OMtot;
    float Y_He;
    float wl;
} CosmoParams;

typedef struct SimulationOptions { /* _inputparams_wrapper.h:28-63 */
    int HII_DIM;            /* cells per side of the low-resolution (output) grids      */
    int DIM;                /* cells per side of the high-resolution (IC) grid          */
    float BOX_LEN;          /* comoving Mpc                                             */
    float NON_CUBIC_FACTOR; /* z-axis length multiplier                                 */
    int N_THREADS;          /* honoured only by host-side sweeps; GPU work ignores it   */
    double Z_HEAT_MAX;
    double ZPRIME_STEP_FACTOR;
    float SAMPLER_MIN_MASS;
    double SAMPLER_BUFFER_FACTOR;
    int N_COND_INTERP;
    int N_PROB_INTERP;
    double MIN_LOGPROB;
    double HALOMASS_CORRECTION;
    double PARKINSON_G0;
    double PARKINSON_y1;
    double PARKINSON_y2;
    float INITIAL_REDSHIFT;
    double DELTA_R_FACTOR;
    double DENSITY_SMOOTH_RADIUS;
    double DEXM_OPTIMIZE_MINMASS;
    double DEXM_R_OVERLAP;
    double CORR_STAR;
    double CORR_SFR;
    double CORR_LX;
    double MIN_XE_FOR_FCOLL_IN_TAUX;
} SimulationOptions;

typedef struct MatterOptions { /* _inputparams_wrapper.h:65-83 */
    bool USE_FFTW_WISDOM; /* accepted and ignored: rocFFT plans are cached per process */
    hmf_model_t HMF;
    v_cb_model_t V_CB_MODEL;
    int POWER_SPECTRUM;
    int USE_INTERPOLATION_TABLES;
    bool PERTURB_ON_HIGH_RES;
    int PERTURB_ALGORITHM;
    bool MINIMIZE_MEMORY;
    bool KEEP_3D_VELOCITIES;
    bool DEXM_OPTIMIZE;
    int FILTER;
    int HALO_FILTER;
    bool SMOOTH_EVOLVED_DENSITY_FIELD;
    source_model_t SOURCE_MODEL;
    int SAMPLE_METHOD;
} MatterOptions;

typedef struct AstroParams { /* _inputparams_wrapper.h:85-143; values arrive LINEAR */
    float HII_EFF_FACTOR;
    float F_STAR10;
    float ALPHA_STAR;
    float ALPHA_STAR_MINI;
    float SIGMA_STAR;
    double UPPER_STELLAR_TURNOVER_MASS;
    double UPPER_STELLAR_TURNOVER_INDEX;
    float F_STAR7_MINI;
    float t_STAR;
    double SIGMA_SFR_INDEX;
    double SIGMA_SFR_LIM;
    double L_X;
    double L_X_MINI;
    double SIGMA_LX;
    float F_ESC10;
    float ALPHA_ESC;
    float F_ESC7_MINI;
    float T_RE;
    float M_TURN;
    float R_BUBBLE_MAX;
    float ION_Tvir_MIN;
    double F_H2_SHIELD;
    float NU_X_THRESH;
    float X_RAY_SPEC_INDEX;
    float X_RAY_Tvir_MIN;
    double A_LW;
    double BETA_LW;
    double A_VCB;
    double BETA_VCB;
    double V_CB_AVG_DEBUG;
    double POP2_ION;
    double POP3_ION;
    double PHOTONCONS_CALIBRATION_END;
    double CLUMPING_FACTOR;
    double ALPHA_UVB;
    float R_MAX_TS;
    int N_STEP_TS;
    double DELTA_R_HII_FACTOR;
    float R_BUBBLE_MIN;
    double MAX_DVDR;
    double NU_X_MAX;
    double NU_X_BAND_MAX;
} AstroParams;

typedef struct AstroOptions { /* _inputparams_wrapper.h:145-165 */
    bool USE_MINI_HALOS;
    bool USE_X_RAY_HEATING;
    bool USE_CMB_HEATING;
    bool USE_LYA_HEATING;
    int RECOMB_MODEL;
    bool USE_TS_FLUCT;
    bool M_MIN_in_Mass;
    bool USE_EXP_FILTER;
    bool CELL_RECOMB;
    bool LYA_MULTIPLE_SCATTERING;
    bool USE_ADIABATIC_FLUCTUATIONS;
    int PHOTON_CONS_TYPE;
    bool USE_UPPER_STELLAR_TURNOVER;
    bool HALO_SCALING_RELATIONS_MEDIAN;
    int HII_FILTER;
    int HEAT_FILTER;
    bool IONISE_ENTIRE_SPHERE;
    integration_method_t INTEGRATION_METHOD_ATOMIC;
    integration_method_t INTEGRATION_METHOD_MINI;
} AstroOptions;

typedef struct Table1D { /* _inputparams_wrapper.h:167-171 */
    int size;
    double *x_values;
    double *y_values;
} Table1D;

typedef struct CosmoTables { /* _inputparams_wrapper.h:173-179 */
    Table1D *transfer_density;
    Table1D *transfer_vcb;
    double ps_norm; /* sigma_8 when USE_SIGMA_8, else A_s */
    bool USE_SIGMA_8;
    double V_CB_AVG;
} CosmoTables;

typedef struct ConfigSettings { /* _inputparams_wrapper.h:181-187 */
    double HALO_CATALOG_MEM_FACTOR;
    bool EXTRA_HALOBOX_FIELDS;
    char *external_table_path;
    char *wisdoms_path;
} ConfigSettings;

/* Process-global parameter pointers, installed by Broadcast_struct_global_all.
 * The memory stays owned by the caller (reference: src/py21cmfast/src/InputParameters.c:82-90). */
extern SimulationOptions *simulation_options_global;
extern MatterOptions *matter_options_global;
extern CosmoParams *cosmo_params_global;
extern AstroParams *astro_params_global;
extern AstroOptions *astro_options_global;
extern CosmoTables *cosmo_tables_global;
extern ConfigSettings config_settings; /* written directly from Python, _cfg.py:52-69 */

/* ------------------------------------------------------------------ */
/* Output structs on the hot path                                     */
/* reference: src/py21cmfast/src/_outputstructs_wrapper.h:6-100       */
/* ------------------------------------------------------------------ */
typedef struct InitialConditions { /* :6-12 ; lowres_* are HII_DIM^3, hires_* are DIM^3 */
    float *lowres_density, *lowres_vx, *lowres_vy, *lowres_vz;
    float *lowres_vx_2LPT, *lowres_vy_2LPT, *lowres_vz_2LPT;
    float *hires_density, *hires_vx, *hires_vy, *hires_vz;
    float *hires_vx_2LPT, *hires_vy_2LPT, *hires_vz_2LPT;
    float *lowres_vcb;
} InitialConditions;

typedef struct PerturbedField { /* :14-16 ; all HII_DIM^3 */
    float *density, *velocity_x, *velocity_y, *velocity_z;
} PerturbedField;

typedef struct HaloBox { /* :44-62 ; only n_ion / whalo_sfr / the two averages are read here */
    float *halo_mass;
    float *halo_stars;
    float *halo_stars_mini;
    float *count;
    float *n_ion;
    float *halo_sfr;
    float *halo_xray;
    float *halo_sfr_mini;
    float *whalo_sfr;
    double log10_Mcrit_ACG_ave;
    double log10_Mcrit_MCG_ave;
} HaloBox;

typedef struct XraySourceBox { /* :67-77 ; [R_ct][HII_TOT_NUM_PIXELS] grids, [N_STEP_TS] means */
    float *filtered_sfr;
    float *filtered_xray;
    float *filtered_sfr_mini;
    float *filtered_sfr_lw;
    float *filtered_sfr_mini_lw;

    double *mean_log10_Mcrit_LW;
    double *mean_sfr;
    double *mean_sfr_mini;
} XraySourceBox;

typedef struct TsBox { /* :78-84 */
    float *spin_temperature;
    float *xray_ionised_fraction;
    float *kinetic_temp_neutral;
    float *J_21_LW;
    double Q_HI;
} TsBox;

typedef struct IonizedBox { /* :87-100 */
    double mean_f_coll;
    double mean_f_coll_MINI;
    double log10_Mturnover_ave;
    double log10_Mturnover_MINI_ave;
    float *neutral_fraction;          /* caller initialises to 1.0 (outputs.py:1524-1527) */
    float *ionisation_rate_G12;       /* caller initialises to 0 */
    float *mean_free_path;            /* caller initialises to 0 */
    float *z_reion;                   /* set to -1 then to the first-crossing redshift */
    float *cumulative_recombinations; /* only with a recombination model */
    float *kinetic_temperature;       /* absent when MINIMIZE_MEMORY */
    float *unnormalised_nion;         /* Eulerian source models only */
    float *unnormalised_nion_mini;
} IonizedBox;

typedef struct BrightnessTemp { /* :102-105 */
    float *brightness_temp;
    float *tau_21; /* only with USE_TS_FLUCT */
} BrightnessTemp;

/* ------------------------------------------------------------------ */
/* Exported entry points                                              */
/* ------------------------------------------------------------------ */

/* reference: src/py21cmfast/src/InputParameters.c:11-54.  Stores the five
 * parameter pointers (no copy) and deep-copies the CosmoTables. */
void Broadcast_struct_global_all(SimulationOptions *simulation_options,
                                 MatterOptions *matter_options, CosmoParams *cosmo_params,
                                 AstroParams *astro_params, AstroOptions *astro_options,
                                 CosmoTables *cosmo_tables);
/* reference: src/py21cmfast/src/InputParameters.c:56-62 */
void Broadcast_struct_global_noastro(SimulationOptions *simulation_options,
                                     MatterOptions *matter_options, CosmoParams *cosmo_params);

/* Power-spectrum normalisation; reference: src/py21cmfast/src/cosmology.c:507-558.
 * Called by @init_c_state(ps=True) before ComputeInitialConditions. */
void init_ps(void);
void free_ps(void);
/* reference: _functionprototypes_wrapper.h:70,86 (recombinations.c:94-140): the MHR00
 * recombination-rate tables; built on first use as well, so calling these is optional */
void init_MHR(void);
void free_MHR(void);

/* reference: src/py21cmfast/src/InitialConditions.c:547 (_functionprototypes_wrapper.h:6) */
int ComputeInitialConditions(unsigned long long random_seed, InitialConditions *boxes);

/* reference: src/py21cmfast/src/PerturbedField.c:389 (_functionprototypes_wrapper.h:8-9) */
int ComputePerturbedField(float redshift, InitialConditions *boxes,
                          PerturbedField *perturbed_field);

/* reference: src/py21cmfast/src/IonisationBox.c:1344 (_functionprototypes_wrapper.h:23-26) */
int ComputeIonizedBox(float redshift, float prev_redshift, PerturbedField *perturbed_field,
                      PerturbedField *previous_perturbed_field, IonizedBox *previous_ionize_box,
                      TsBox *spin_temp, HaloBox *halos, InitialConditions *ini_boxes,
                      IonizedBox *box);

/* reference: src/py21cmfast/src/SpinTemperatureBox.c:87 (_functionprototypes_wrapper.h:19-22).
 * The Eulerian (CONST-ION-EFF, E-INTEGRAL) and the Lagrangian source models with interpolation
 * tables; USE_MINI_HALOS with E-INTEGRAL (previous J_21_LW in) or with source grids
 * (XraySourceBox.filtered_sfr_mini and mean_log10_Mcrit_LW), J_21_LW out.  `cleanup` is accepted and ignored (nothing is cached per call beyond the
 * data tables). */
int ComputeTsBox(float redshift, float prev_redshift, float perturbed_field_redshift, short cleanup,
                 PerturbedField *perturbed_field, XraySourceBox *source_box,
                 TsBox *previous_spin_temp, InitialConditions *ini_boxes, TsBox *this_spin_temp);
/* reference: src/py21cmfast/src/heating_helper_progs.c:58-90 (_functionprototypes_wrapper.h):
 * read the data tables under config_settings.external_table_path */
int init_heat(void);
void destruct_heat(void);

/* reference: src/py21cmfast/src/HaloBox.c:563 (_functionprototypes_wrapper.h:31-32).
 * SOURCE_MODEL = L-INTEGRAL: the integrated branch, `halos` is not read.  SOURCE_MODEL = DEXM-ESF /
 * CHMF-SAMPLER: the halos of a (perturbed-position-free) catalogue get their galaxy properties
 * (set_halo_properties, HaloBox.c:62-102), are moved with the 1LPT/2LPT displacement of their
 * Lagrangian cell and CIC-deposited (move_halo_galprops, map_mass.c:346-476); the sources below
 * the catalogue's mass limit are added by the integrated branch.  The catalogue itself comes from
 * the caller (the halo finder / sampler are not part of this backend).  USE_MINI_HALOS
 * (low-resolution sources): the previous TsBox.J_21_LW and IonizedBox Gamma_12 / z_reion set the
 * turnover masses below Z_HEAT_MAX; halo_sfr_mini is filled, n_ion holds both populations. */
/* reference: src/py21cmfast/src/_outputstructs_wrapper.h:18-28 */
typedef struct HaloCatalog {
    unsigned long long int n_halos;
    unsigned long long int buffer_size;
    float *halo_masses;
    float *halo_coords; /* [3 n_halos], Mpc */
    float *star_rng;    /* standard-normal deviates of the three scaling relations */
    float *sfr_rng;
    float *xray_rng;
} HaloCatalog;
int ComputeHaloBox(double redshift, InitialConditions *ini_boxes, HaloCatalog *halos,
                   TsBox *previous_spin_temp, IonizedBox *previous_ionize_box, HaloBox *grids);

/* reference: src/py21cmfast/src/HaloBox.c:658-779 (_functionprototypes_wrapper.h:127-130; bound by
 * py21cmfast's cfuncs.convert_halo_properties).  Twelve floats per halo: mass, M*, SFR, L_X
 * [1e38 erg/s], n_ion, f_esc-weighted SFR, M*_mini, SFR_mini, M_turn (atomic, molecular,
 * reionisation), metallicity; the feedback grids [HII_DIM^3] are read with USE_MINI_HALOS only
 * (J_21_LW, z_re, Gamma_12 below Z_HEAT_MAX; vcb with V_CB_MODEL = FLUCTS).  Rows of halos with
 * zero mass are left as they are.  Arrays host or device. */
int test_halo_props(double redshift, float *vcb_grid, float *J21_LW_grid, float *z_re_grid,
                    float *Gamma12_ion_grid, unsigned long long n_halos, float *halo_masses,
                    float *halo_coords, float *star_rng, float *sfr_rng, float *xray_rng,
                    float *halo_props_out);

/* reference: src/py21cmfast/src/BrightnessTemperatureBox.c:22 (_functionprototypes_wrapper.h:28-29) */
int ComputeBrightnessTemp(float redshift, TsBox *spin_temp, IonizedBox *ionized_box,
                          PerturbedField *perturb_field, BrightnessTemp *box);

/* reference: src/py21cmfast/src/SpinTemperatureBox.c:748-808 (_functionprototypes_wrapper.h:34-35).
 * One shell of the X-ray / Lyman-alpha source grids: halo_sfr (window 5 with
 * LYA_MULTIPLE_SCATTERING, else 4) and halo_xray (window 4) between R_inner and R_outer,
 * negative cells zeroed, stored at [R_ct * HII_TOT_NUM_PIXELS]; with USE_MINI_HALOS also
 * halo_sfr_mini and, under multiple scattering, the straight-line copies for the LW feedback. */
int UpdateXraySourceBox(HaloBox *halobox, double R_inner, double R_outer, int R_ct, double R_star,
                        XraySourceBox *source_box);

/* reference: src/py21cmfast/src/filtering.c:126-160,258-293 (_functionprototypes_wrapper.h:134-136):
 * the fits mu(x_em), eta(x_em) and the hypergeometric function of the multiple-scattering window */
double compute_mu_for_multiple_scattering(double x_em);
double compute_eta_for_multiple_scattering(double x_em);
double hyper_2F3(double kR, double alpha, double beta);

/* reference: src/py21cmfast/src/filtering.c:397 (_functionprototypes_wrapper.h:130-131).
 * r2c -> /N -> filter_box -> c2r of one HII_DIM^3 box; `result` is float64[N]. */
int test_filter(float *input_box, double R, double R_param, double R_star, int filter_flag,
                double *result);

#ifdef __cplusplus
}
#endif
#endif /* C21CM_ABI_H */
