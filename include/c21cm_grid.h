/*
 * c21cm_grid.h -- explicit-scalar ("grid level") entry points of lib21cmfast_hip.so.
 *
 * The drop-in entry points of c21cm_abi.h read their physics scalars from the
 * process-global parameter structs and from host-side cosmology / HMF
 * integrals.  Everything they do on the GRIDS is delegated to the functions
 * declared here, which take every scalar explicitly.  These are what
 * bench.py and the parity tests call: the grid work can then be compared with
 * the CPU oracle on identical numbers, independent of host-integral numerics
 * ("mean_f_coll, f_limit supplied as fixed scalars").
 *
 * Pointer arguments may address host memory or MI355X HBM; the library detects
 * which (hipPointerGetAttributes) and stages host arrays through device
 * scratch.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Every function returns a c21cm_status code.
 */
#ifndef C21CM_GRID_H
#define C21CM_GRID_H

#include <stddef.h>

#include "c21cm_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define C21CM_MAX_RADII 256
#define C21CM_NDELTA_TABLE 400 /* reference: src/py21cmfast/src/interp_tables.c:27,34 */

/* How the per-cell collapsed fraction is obtained inside the R loop
 * (reference: src/py21cmfast/src/IonisationBox.c:821-881). */
enum c21cm_fcoll_mode {
    C21CM_FCOLL_STARS_GRID = 0,   /* Lagrangian source grids: f = filtered HaloBox.n_ion    */
    C21CM_FCOLL_ERFC = 1,         /* CONST-ION-EFF, no tables: FgtrM_bias_fast closed form  */
    C21CM_FCOLL_TABLE_LINEAR = 2, /* CONST-ION-EFF with tables: lerp(table, delta)          */
    C21CM_FCOLL_TABLE_EXP = 3,    /* E-INTEGRAL: exp(lerp(table, delta))                    */
    C21CM_FCOLL_NODES = 4         /* E-INTEGRAL WITHOUT interpolation tables: the conditional integral
                                   * Nion_ConditionalM per cell (IonisationBox.c:889-893, hmf.c:1106-1140)
                                   * on the Gauss-Legendre nodes of the radius, see below          */
};

/* Host callback used by the two TABLE modes: fill `table[C21CM_NDELTA_TABLE]`
 * (float, like the reference's RGTable1D_f) for filter radius index `r_index`
 * on the regular delta grid x_i = dens_min + i*(dens_max-dens_min)/(NDELTA-1).
 * reference: src/py21cmfast/src/IonisationBox.c:702-768. */
/* Mode C21CM_FCOLL_NODES: table_fn is called as table_fn(r_index, 0, 0, (float *)nodes, user) and fills
 * C21CM_NODE_DOUBLES DOUBLES instead of a float table -- everything of the integrand that does not
 * depend on the cell's overdensity:
 *   nodes[0] number of Gauss-Legendre nodes n (<= 100)   nodes[1] 0: extended Press-Schechter, 1: Sheth-Tormen
 *   nodes[2] growth factor                                nodes[3] delta above which the cell has collapsed
 *   nodes[4] the value returned for such a cell           nodes[5] != 0: the integral is empty (M_min >= M_cond)
 *   nodes[8 + 4 i ..]  (w_i x prefactor_i, barrier-expansion factor_i, barrier_i, 1 / (sigma_i^2 - sigma_c^2))
 * and the device sums  - w_i pref_i (factor_i - d0) exp(-(barrier_i - d0)^2 sdi_i / 2),  d0 = delta / D
 * (ST), or the same with (delta_c - delta) / D for both factors (PS), over the nodes. */
#define C21CM_NODE_DOUBLES 408
typedef int (*c21cm_table_fn)(int r_index, double dens_min, double dens_max, float *table,
                              void *user);

/* USE_MINI_HALOS: the two conditional N_ion tables of one radius are 2-D, overdensity x log10 of
 * the turnover mass (interp_tables.c:291-405): table[i * C21CM_NMTURN_TABLE + j] = ln N_ion at
 * delta_i = dens_min + i (dens_max - dens_min)/(NDELTA-1) and turnover 10^(l10mt_min +
 * j (l10mt_max - l10mt_min)/(NMTURN-1)); `table_acg` for the atomically cooled galaxies (turnover
 * range l10mt_*), `table_mcg` for the molecularly cooled ones (range l10mt_*_mini).  prev != 0:
 * the tables at the previous snapshot's redshift (trapezoidal history, IonisationBox.c:752-760). */
#define C21CM_NMTURN_TABLE 50 /* interp_tables.c:28 */
typedef int (*c21cm_table2d_fn)(int r_index, int prev, double dens_min, double dens_max,
                                double l10mt_min, double l10mt_max, double l10mt_min_mini,
                                double l10mt_max_mini, float *table_acg, float *table_mcg,
                                void *user);

/* All scalars of one ComputeIonizedBox call.
 * reference: struct IonBoxConstants / RadiusSpec, src/py21cmfast/src/IonisationBox.c:38-102. */
typedef struct c21cm_ionize_spec {
    /* geometry */
    int hii_dim;      /* cells along x and y                          */
    int hii_dim_z;    /* cells along z (= NON_CUBIC_FACTOR * HII_DIM) */
    double box_len;   /* Mpc, x and y                                 */
    double box_len_z; /* Mpc, z                                       */

    /* filter radii, ascending; index 0 is the cell-scale radius (setup_radii :964-1006) */
    int n_radii;
    int r_lowest; /* lowest radius index that is processed (0 unless M_min > RtoM(R) breaks :1537) */
    double R[C21CM_MAX_RADII];
    double sigma_maxmass[C21CM_MAX_RADII]; /* sigma_z0(RtoM(R)); ERFC / TABLE_LINEAR modes */

    /* filters (copy_filter_transform :572-664) */
    int hii_filter;      /* delta, x_e, N_rec grids                 */
    int stars_filter;    /* n_ion, whalo_sfr grids (3 = exp-MFP)    */
    double mfp_meandens; /* R_param of filter 3                     */

    /* source model */
    int fcoll_mode;   /* enum c21cm_fcoll_mode                                         */
    int fix_mean;     /* rescale grid f_coll to mean_f_coll (Eulerian models)          */
    int mass_dep_zeta; /* floor f at f_limit_acg (:1077-1082) and mean clamp (:1566)  */
    c21cm_table_fn table_fn;
    void *table_user;

    /* option flags */
    int use_ts_fluct;          /* filter TsBox.xray_ionised_fraction, partial T from Ts    */
    int recomb_model;          /* enum C21CM_RECOMB_*                                      */
    int cell_recomb;           /* AstroOptions.CELL_RECOMB                                 */
    int minimize_memory;       /* skip kinetic_temperature / mean_free_path                */
    int first_snapshot;        /* prev_redshift < 1: previous z_reion := -1 (:365-401)     */

    /* redshift / astro scalars (set_ionbox_constants :125-227) */
    double redshift;        /* written into z_reion on first crossing  */
    double stored_redshift; /* used by the ionised-temperature formula */
    double photoncons_adjustment_factor;
    double ion_eff_factor;  /* zeta applied to the per-cell f_coll     */
    double mean_f_coll;     /* global expectation (fix_mean numerator) */
    double f_limit_acg;
    double gamma_prefactor;
    double rhocrit_omb; /* RHOcrit * OMb, Lagrangian absorber normalisation (:1066) */
    double growth_factor;
    double sigma_minmass;
    double delta_c; /* physconst.delta_c_sph = 1.686 */
    double TK_nofluct;
    double adia_TK_term;
    double T_re;
    double fabs_dtdz;
    double dz;

    /* recombinations (recomb_model != none): the table behind splined_recombination_rate
     * (recombinations.c:64-122): rr_y[z_ct * C21CM_RR_NGAMMA + g] = recombination_rate(z_ct * 0.2f,
     * exp(ln Gamma_g), T4 = 1, case B) with ln Gamma_g = -10 + 0.1f g, and rr_c = the natural cubic
     * spline's c coefficients (half the second derivatives) of each row in ln Gamma, as
     * gsl_interp_cspline holds them.  Host arrays; c21_rr_tables() builds them (init_MHR). */
    const double *rr_y, *rr_c;

    /* mini-halos: E-INTEGRAL with USE_MINI_HALOS (need_minihalo_nion, IonisationBox.c:30-31).
     * Four filtered grids per radius (delta, the previous snapshot's delta and the two log10
     * turnover-mass grids of c21cm_mturn_grids), f_coll of both populations from the 2-D tables,
     * accumulated over snapshots per radius (box->unnormalised_nion[_mini][n_radii][N], :908-936):
     *   f(R) = f_prev_box(R) + f(z, M_turn) - f(z_prev, M_turn)
     * and the barrier f zeta + f_m zeta_m > (1 - x_e)(1 + rec) (:1118-1120). */
    int use_mini_halos;
    int need_prev_ion; /* previous box: mean_f_coll zeta + mean_f_coll_MINI zeta_m > 1e-4 (:1546) */
    double ion_eff_factor_mini;
    double mean_f_coll_mini;
    double f_limit_mcg;
    double gamma_prefactor_mini;
    const float *prev_density;    /* previous PerturbedField.density [N]          */
    const float *log10_mturn_acg; /* [N], calculate_mcrit_boxes (:403-457)        */
    const float *log10_mturn_mcg;
    c21cm_table2d_fn table2d_fn;
    void *table2d_user;

    /* AstroOptions.IONISE_ENTIRE_SPHERE (IonisationBox.c:1150-1158): a crossing cell flags all
     * cells within R as ionised (x_HI only; z_reion stays with the centres).  Without a
     * recombination model and without mini-halos (upstream's result then depends on the order in
     * which its threads paint), and with a cell-scale radius below one cell. */
    int ionise_entire_sphere;
} c21cm_ionize_spec;

#define C21CM_RR_NZ 300       /* recombinations.c:35 RR_Z_NPTS        */
#define C21CM_RR_NGAMMA 250   /* :39 RR_lnGamma_NPTS                  */
#define C21CM_RR_DZ 0.2f      /* :38 RR_DEL_Z      (a float upstream) */
#define C21CM_RR_LNGAMMA_MIN (-10.0) /* :40                            */
#define C21CM_RR_DLNGAMMA 0.1f       /* :41 RR_DEL_lnGamma (float)     */

/* Per-call diagnostics returned by the grid-level ionisation driver. */
typedef struct c21cm_ionize_report {
    double f_coll_grid_mean[C21CM_MAX_RADII]; /* after the clamp of :1566-1576 */
    double global_xH;
    double mean_f_coll_out; /* what ComputeIonizedBox leaves in box->mean_f_coll */
    double ms_preloop, ms_rloop, ms_postloop; /* device timings (hip events)     */
    double f_coll_grid_mean_mini[C21CM_MAX_RADII]; /* USE_MINI_HALOS */
    double mean_f_coll_mini_out;
} c21cm_ionize_report;

/* calculate_mcrit_boxes (IonisationBox.c:403-457): per cell log10 of the turnover masses
 *   M_turn,a = max(M_reion-feedback, mturn_a_nofb),  M_turn,m = max(M_rf, M_LW(J_21_LW, v_cb), mturn_m_nofb)
 * from the previous box's Gamma_12 / z_reion and the TsBox's J_21_LW (vcb == NULL: vcb_const).
 * Arrays on the host or the device ([N], dense); averages of the two log10 grids returned. */
typedef struct c21cm_mturn_spec {
    int hii_dim, hii_dim_z;
    int first_snapshot; /* previous z_reion is -1 everywhere (no feedback)           */
    double redshift;
    double mturn_a_nofb, mturn_m_nofb, vcb_const;
    double A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb; /* sigma_vcb = V_CB_AVG sqrt(3 pi / 8) */
} c21cm_mturn_spec;
int c21cm_mturn_grids(const c21cm_mturn_spec *spec, const float *prev_G12,
                      const float *prev_z_reion, const float *J_21_LW, const float *vcb,
                      float *log10_mturn_acg, float *log10_mturn_mcg, double *ave_acg,
                      double *ave_mcg, void *stream);

/* The whole ComputeIonizedBox grid algorithm (pre-loop r2c, R loop, post-loop).
 * reference: src/py21cmfast/src/IonisationBox.c:1477-1628. */
int c21cm_ionize_grids(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                       const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                       const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                       void *stream);

/* Early exit of ComputeIonizedBox (expected HII fraction < 1e-5): uniform neutral box.
 * reference: src/py21cmfast/src/IonisationBox.c:531-565 */
int c21cm_neutral_box(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                      const TsBox *spin_temp, IonizedBox *box, size_t ntot);

/* R-loop sharding over GPUs.  Each rank runs the radii
 * r = n_radii-1-rank, n_radii-1-rank-world, ... > 0 and records in
 * `first_cross[N]` (uint8, device) the largest 1-based radius index whose
 * barrier the cell crossed (0 = none).  The caller max-reduces first_cross over
 * ranks (RCCL), then the rank that owns the outputs calls the finish step, which
 * applies the mask, runs radius index 0 (partial ionisation) and the post-loop. */
int c21cm_ionize_shard_radii(const c21cm_ionize_spec *spec, int rank, int world,
                             const PerturbedField *perturbed_field,
                             const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                             const HaloBox *halos, unsigned char *first_cross,
                             c21cm_ionize_report *report, void *stream);
/* Per-radius f_coll grid means for the finish step: the element-wise SUM over ranks of the shard
 * phases' report->f_coll_grid_mean (each radius > 0 belongs to one rank).  Optional; consumed by
 * the next c21cm_ionize_shard_finish of this process.  Needed for box->mean_f_coll of Lagrangian
 * models when r_lowest > 0 (IonisationBox.c:1623-1628) and for a complete report. */
int c21cm_ionize_shard_set_means(const double *means, int n_radii);
int c21cm_ionize_shard_finish(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                              const PerturbedField *perturbed_field,
                              const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                              const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                              void *stream);

/* The finish phase split by cell slabs (round 5).  The cell-scale radius and the post-loop are per cell
 * (IonisationBox.c:1031-1256,1597-1608) and every rank holds the replicated inputs, so rank r finishes
 * the cells of ITS slab only -- whole chunks of the final sweep, c21cm_ionize_shard_slab -- from the
 * combined first crossings of that slab (first_cross is read on [cell_begin, cell_end) only: the
 * exchange before it moves 1 / world of each rank's packed grid per link instead of whole grids onto one
 * rank).  The sweep leaves the chunks' partial sums where the single pass leaves them; `exchange` then
 * all-gathers them (and, if the caller wants full boxes on every rank, the output slabs), after which
 * every rank reduces ALL chunks in the single pass' fixed order: global_xH and mean_f_coll are the
 * single pass' to the last bit on every rank, without a scalar broadcast.
 *   exchange(user, state, local_status, stream) is entered exactly once per call whatever happened
 *   locally (local_status != 0: this rank failed before its sweep; `state` may then be incomplete) and
 *   returns the status the ranks agreed on; NULL on a one-rank run.
 *   outputs_gathered != 0: the callback gathered the output slabs, host arrays receive whole grids;
 *   0: the outputs are slab-resident (host arrays receive this rank's slab only). */
typedef struct c21cm_shard_slab_state {
    int rank, world;
    int n_chunks, chunk_begin, chunk_end; /* this rank's chunks of the final sweep */
    size_t chunk_cells, cell_begin, cell_end, ntot;
    double *partials_stars, *partials_xh; /* device, n_chunks each; [chunk_begin, chunk_end) filled */
    int *flag;                            /* device: non-finite flag of this rank's slab (combine: max) */
    float *out[3]; /* device addresses of neutral_fraction, z_reion, kinetic_temperature (NULL: absent) */
} c21cm_shard_slab_state;
typedef int (*c21cm_shard_slab_exchange_fn)(void *user, const c21cm_shard_slab_state *state,
                                            int local_status, void *stream);
int c21cm_ionize_shard_slab_supported(const c21cm_ionize_spec *spec);
int c21cm_ionize_shard_slab(const c21cm_ionize_spec *spec, int rank, int world, int *chunk_begin,
                            int *chunk_end, size_t *cell_begin, size_t *cell_end, int *n_chunks,
                            size_t *chunk_cells);
int c21cm_ionize_shard_finish_slab(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                                   int rank, int world, const PerturbedField *perturbed_field,
                                   const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                   const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                                   c21cm_shard_slab_exchange_fn exchange, void *exchange_user,
                                   int outputs_gathered, void *stream);
/* device helpers of the slab exchange of the packed first crossings (one bit per cell):
 * pack the uint8 grid (n cells, n and the grid's start multiples of 32 cells or the box end);
 * OR `world` packed pieces (`stride_words` apart) into n bytes at `first_cross` */
int c21cm_shard_pack_mask_bits(const unsigned char *first_cross, unsigned *bits, size_t n, void *stream);
int c21cm_shard_or_unpack_mask_bits(const unsigned *bits, size_t stride_words, int world,
                                    unsigned char *first_cross, size_t n, void *stream);

/* The same two phases for a recombination model (spec->recomb_model != none): a first crossing
 * carries the radius (= mean free path) and Gamma_12, so the shard phase leaves 64-bit keys
 * bits(mfp) << 32 | bits(G12) in `cross_keys[N]` (device) -- non-negative floats order like their
 * bit patterns -- the caller max-reduces them as unsigned 64-bit integers, and the finish phase
 * unpacks the winner of every cell (reference: IonisationBox.c:1124-1140). */
int c21cm_ionize_shard_radii_keys(const c21cm_ionize_spec *spec, int rank, int world,
                                  const PerturbedField *perturbed_field,
                                  const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                  const HaloBox *halos, unsigned long long *cross_keys,
                                  c21cm_ionize_report *report, void *stream);
int c21cm_ionize_shard_finish_keys(const c21cm_ionize_spec *spec,
                                   const unsigned long long *cross_keys,
                                   const PerturbedField *perturbed_field,
                                   const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                   const HaloBox *halos, IonizedBox *box,
                                   c21cm_ionize_report *report, void *stream);

/* The fused recombination loop sharded (CELL_RECOMB, no x_e grid, native line lengths with the
 * windows evaluated in pass X: c21cm_ionize_shard_rc_supported != 0): a rank's radii leave the
 * uint8 first-crossing index and Gamma_12 at the crossing -- 5 bytes per cell instead of the
 * 8-byte key; the mean free path is the radius of the index.  Between the phases the caller keeps,
 * per cell, the entry of the rank with the LARGER index (an index > 0 is owned by one rank):
 * c21cm_ionize_sharded does it with a reduce-scatter by cell slabs + a gather (5 N / world bytes
 * per link and hop).  reference: IonisationBox.c:1084-1140,1531-1588 */
int c21cm_ionize_shard_rc_supported(const c21cm_ionize_spec *spec);
int c21cm_ionize_shard_radii_rc(const c21cm_ionize_spec *spec, int rank, int world,
                                const PerturbedField *perturbed_field,
                                const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                const HaloBox *halos, unsigned char *first_cross, float *cross_g12,
                                c21cm_ionize_report *report, void *stream);
int c21cm_ionize_shard_finish_rc(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                                 const float *cross_g12, const PerturbedField *perturbed_field,
                                 const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                 const HaloBox *halos, IonizedBox *box,
                                 c21cm_ionize_report *report, void *stream);
/* device helper of that exchange (own slab in place against n_peers received slabs of `stride`
 * cells; all starts at multiples of 4 cells) */
int c21cm_shard_combine_cross_g12(unsigned char *mask, float *g12, const unsigned char *peer_mask,
                                  const float *peer_g12, int n_peers, size_t stride, size_t n,
                                  void *stream);

/* ---- the sharded R loop behind the C ABI ---------------------------------------------------
 * One process per GPU; every rank calls c21cm_ionize_sharded with the same (replicated) inputs.
 * Collectives go through RCCL (librccl resolved at run time with dlopen: inside a PyTorch
 * process that is torch's own copy, so there is one RCCL per process).  Bootstrap: rank 0
 * obtains 128 bytes from c21cm_shard_unique_id, the launcher distributes them by any means
 * (torch.distributed, MPI, a file), every rank calls c21cm_shard_init.  Once initialised,
 * ComputeIonizedBox itself shards (C21CM_SHARD=0 keeps it single-GPU).
 * The rank that would own radius index 0 finishes and holds the outputs; with broadcast != 0
 * they are broadcast so that every rank returns the full box, as a drop-in caller expects. */
#define C21CM_SHARD_ID_BYTES 128
int c21cm_shard_unique_id(void *id128);
int c21cm_shard_init(int rank, int world, const void *id128);
int c21cm_shard_finalize(void);
int c21cm_shard_info(int *rank, int *world); /* returns 0 and fills them when initialised */
/* device time (ms) of this rank's last c21cm_ionize_sharded call: shard phase, exchange (includes
 * the wait for the slowest peer), finish; and the rank count RCCL itself reports (ncclCommCount) */
int c21cm_shard_last_phases(double ms[3]);
int c21cm_shard_comm_count(void);
/* 1: the communicator is RCCL's (not the in-process emulation the tests use) */
int c21cm_shard_is_rccl(void);
/* collective: 1 when EVERY rank passed a non-zero `local_yes` (ranks agree on a path before they
 * take it: ComputeTsBox shards only if all of them hold device arrays) */
int c21cm_shard_all_agree(int local_yes);
/* 1: shard_rccl.c was compiled against <rccl/rccl.h> and its hand-declared ncclUniqueId /
 * ncclDataType_t / ncclRedOp_t values and the prototypes it calls were checked against it */
int c21cm_shard_rccl_header_checked(void);

/* ---- ComputeTsBox sharded over the same communicator (the N_STEP_TS shells dealt round-robin;
 * reference: SpinTemperatureBox.c:1541-1784 is linear in the shells).  With a communicator in place
 * ComputeTsBox shards by itself for the Eulerian table models on device arrays (C21CM_SHARD_TS=0
 * opts out); the two compute phases are exported for tests and other transports:
 *   c21cm_ts_box_shard_sums    rank's shells: density filter loop, tables, box means, shell loop ->
 *                              partial sums [6][N] doubles (device), *n_rows of them in use
 *   (exchange)                 sum the partials over the ranks; rank r needs cells
 *                              [c21cm_ts_slab_begin(N, world, r), c21cm_ts_slab_begin(N, world, r + 1))
 *   c21cm_ts_box_shard_finish  temperature update of a cell range from its complete sums
 *                              ([n_rows][ncell] doubles, device) into the output boxes
 * Parameters come from the broadcast globals like ComputeTsBox's. */
int c21cm_ts_shardable(float redshift, const PerturbedField *pf, const TsBox *prev, const TsBox *out);
int c21cm_ts_shard_shells(int n_step, int rank, int world, int *idx);
size_t c21cm_ts_slab_begin(size_t ntot, int world, int r);
int c21cm_ts_box_shard_sums(float redshift, float prev_redshift, float perturbed_field_redshift,
                            PerturbedField *perturbed_field, TsBox *previous_spin_temp, int rank,
                            int world, double *sums_dev, int *n_rows);
int c21cm_ts_box_shard_finish(float redshift, float prev_redshift, float perturbed_field_redshift,
                              PerturbedField *perturbed_field, TsBox *previous_spin_temp,
                              const double *slab_sums, size_t cell0, size_t ncell,
                              TsBox *this_spin_temp);
int c21cm_ts_box_sharded(float redshift, float prev_redshift, float perturbed_field_redshift,
                         PerturbedField *perturbed_field, TsBox *previous_spin_temp,
                         TsBox *this_spin_temp);
int c21cm_ts_box_sharded_calls(void); /* calls of c21cm_ts_box_sharded in this process (tests) */
int c21cm_shard_owner(int n_radii, int world);
/* Test hook: replace the transport by an in-process device mailbox so that the ranks of a
 * world > 1 run can be executed one after the other in ONE process (non-owners first, the owner
 * last; mailbox zeroed before each round, >= world * N/8 + 8 N bytes).  See shard_rccl.c. */
int c21cm_shard_emulate(int rank, int world, void *mailbox, size_t mailbox_bytes);
/* Diagnostic: where the time of this process' last ComputeIonizedBox went -- out[0] host ms before the
 * device driver (scalars, tables, spec), [1] device pre-loop, [2] R loop, [3] post-loop, [4] wall ms of
 * the call, [5] number of filter radii. */
int c21cm_last_ionize_timing(double out[6]);
/* diagnostic: which R loop the last ionisation call set up -- 1 fused loop, 2 fused recombination loop,
 * 4 a third spectrum in the barrier kernel, 8 ... which is the filtered N_rec, 16 a fourth spectrum
 * (x_e AND filtered N_rec), 32 two radii per pass-X sweep */
int c21cm_ionize_last_loop_flags(void);

/* What a sharded call leaves in the output arrays -- `broadcast` of c21cm_ionize_sharded, and what the
 * drop-in ComputeIonizedBox passes (c21cm_shard_output_mode: c21cm_shard_set_output, else the environment
 * C21CM_SHARD_OUTPUT = all | none | auto, default auto):
 *    1  whole boxes on every rank (all-gather of the output slabs, or a broadcast of the owner's box)
 *    2  (slab finish) the whole neutral-fraction box on every rank, z_reion and T_k slab-resident:
 *       4 of the 12 bytes per cell of mode 1 (C21CM_SHARD_OUTPUT=xH); device arrays only
 *    0  what the finish leaves: every rank its slab (c21cm_ionize_shard_slab) where the finish phase runs
 *       by cell slabs, the owner's box otherwise; scalars (global_xH, mean_f_coll) complete on every rank
 *       of a slab finish
 *   -1  auto: slab-resident where the finish runs by slabs, the owner's box broadcast otherwise
 * c21cm_shard_last_finish_was_slab(): 1 if this process' last c21cm_ionize_sharded finished by slabs. */
int c21cm_shard_set_output(int mode);
int c21cm_shard_output_mode(void);
int c21cm_shard_last_finish_was_slab(void);
int c21cm_ionize_sharded(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                         const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                         const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                         int broadcast, void *stream);

/* filter_box on an explicit geometry: r2c, /N, W(kR) multiply, c2r.
 * reference: src/py21cmfast/src/filtering.c:308-445 (filter_box / test_filter). */
int c21cm_filter_grid(const float *input, float *output, int nx, int ny, int nz, double box_len,
                      double box_len_z, int filter_type, double R, double R_param, void *stream);

/* In-place padded real FFTs (reference: src/py21cmfast/src/dft.c:18-72).
 * `box` is float[nx][ny][2*(nz/2+1)] on the device. */
int c21cm_fft_r2c(float *box, int nx, int ny, int nz, void *stream);
int c21cm_fft_c2r(float *box, int nx, int ny, int nz, void *stream);

/* Scalars of one ComputePerturbedField call
 * (reference: src/py21cmfast/src/PerturbedField.c:24-135,389-496, map_mass.c:146-208). */
typedef struct c21cm_perturb_spec {
    int dim, dim_z;         /* hi-res particle grid (DIM, D_PARA)                         */
    int hii_dim, hii_dim_z; /* low-res output grid (HII_DIM, HII_D_PARA)                  */
    double box_len, box_len_z;
    int perturb_algorithm;  /* enum C21CM_PERTURB_*                                       */
    int perturb_on_high_res;
    int keep_3d_velocities;
    int smooth_evolved_density;
    double density_smooth_radius_mpc; /* Gaussian R handed to filter_box (PerturbedField.c:222-225) */
    double growth_factor;      /* dicke(z)                                                */
    double init_growth_factor; /* dicke(INITIAL_REDSHIFT)                                 */
    double dDdt_over_D;        /* ddickedt(z)/dicke(z)                                    */
} c21cm_perturb_spec;

int c21cm_perturb_grids(const c21cm_perturb_spec *spec, const InitialConditions *ics,
                        PerturbedField *pf, void *stream);

/* Scalars of one ComputeInitialConditions call
 * (reference: src/py21cmfast/src/InitialConditions.c:547-772). */
typedef struct c21cm_ics_spec {
    int dim, dim_z;         /* DIM, D_PARA                                                   */
    int hii_dim, hii_dim_z; /* HII_DIM, HII_D_PARA                                           */
    double box_len, box_len_z;
    float volume;           /* VOLUME macro: float product BOX_LEN^3 * NON_CUBIC_FACTOR      */
    int perturb_algorithm;  /* 2LPT fields only when == C21CM_PERTURB_2LPT                   */
    int perturb_on_high_res;
    int density_is_input;   /* regenerate everything from boxes->hires_density (:620-663)    */
    /* mode sampling (density_is_input == 0), cubic grids only: P(k) tabulated at
     * k = (2 pi / L) sqrt(m), m = 0 .. 3 (DIM/2)^2, in Mpc^3 (power_in_k, cosmology.c:278) */
    int n_m;
    const double *pk_by_m; /* host array */
    unsigned long long seed;
    /* which random stream turns `seed` into delta_k:
     *   C21CM_RNG_PHILOX (0)  counter-based Philox-4x32-10 + Box-Muller on the device (counter =
     *                         mode index: independent of launch geometry and of N_THREADS)
     *   C21CM_RNG_GSL    (1)  the reference's streams -- seed_rng_threads (rng.c:31-90) and two
     *                         gsl_ran_ugaussian per mode from the generator of the OpenMP thread
     *                         that owns the mode's n_x (InitialConditions.c:103-139) -- for
     *                         N_THREADS = rng_threads (1 or 2): same seed, same universe as upstream.
     *                         The stream is serial by construction and is drawn on the host. */
    int rng_stream;
    int rng_threads;
    /* V_CB_MODEL = FLUCTS (compute_relative_velocities, InitialConditions.c:141-238), cubic grids:
     * sqrt(P_vcb(k) / P(k)) c_kms / k at k = (2 pi / L) sqrt(m), m = 0 .. 3 (DIM/2)^2 (entry 0
     * unused); NULL = no relative velocities.  Output: InitialConditions.lowres_vcb. */
    const double *vcb_by_m;
} c21cm_ics_spec;
enum { C21CM_RNG_PHILOX = 0, C21CM_RNG_GSL = 1 };

int c21cm_ics_grids(const c21cm_ics_spec *spec, InitialConditions *ics, void *stream);

/* ---- ComputeBrightnessTemp grid algorithm (BrightnessTemperatureBox.c:22-105) -----------
 * delta_T = const_factor * x_HI * (1 + delta) [mK]; with spin temperatures additionally the
 * optical depth tau_21 and (1 - exp(-tau)) (T_S - T_rad)/(1+z).  Arrays may be host or device.
 * mean_out (optional): the box average the reference logs and checks for finiteness. */
typedef struct c21cm_brightness_spec {
    size_t n_cells;
    double redshift;
    float const_factor; /* 27 (Ob h^2/0.023) sqrt(0.15/(Om h^2) (1+z)/10), float as in :44-49 */
    float T_rad;        /* T_cmb (1+z), float */
    int use_ts_fluct;
} c21cm_brightness_spec;

int c21cm_brightness_grids(const c21cm_brightness_spec *spec, const float *density,
                           const float *neutral_fraction, const float *spin_temperature,
                           float *brightness_temp, float *tau_21, double *mean_out, void *stream);

/* ---- ComputeHaloBox, integrated ("fixed grid") branch: HaloBox.c:302-436 + map_mass.c:214-344 --
 * Each cell of the Lagrangian grid (`hires_density` with PERTURB_ON_HIGH_RES, else
 * `lowres_density`, times the growth factor) gets its expected emissivity and star-formation
 * rate from two 400-bin conditional-mass-function tables (ln N_ion(delta), ln SFRD(delta));
 * the values are moved with the 1LPT/2LPT displacement and CIC-deposited on the HII_DIM grid. */
typedef struct c21cm_halobox_spec {
    int dim, dim_z;         /* grid of the source cells (DIM or HII_DIM) */
    int hii_dim, hii_dim_z; /* output grid */
    double box_len, box_len_z;
    int perturb_on_high_res; /* sources = hires_density + hires_v*, else lowres_density + lowres_v* */
    int lpt2;                /* PERTURB_ALGORITHM == 2LPT */
    double growth_factor, init_growth_factor;
    /* tables over delta = density * growth_factor in [tab_min, tab_min + 399 tab_width]
     * (interp_tables.c:291-405,415-494), evaluated as exp(lerp) */
    double tab_min, tab_width;
    const float *ln_nion_table; /* host, C21CM_NDELTA_TABLE floats */
    const float *ln_sfrd_table; /* host, C21CM_NDELTA_TABLE floats */
    double prefactor_nion, prefactor_sfr; /* map_mass.c:228-239 */
    double prefactor_wsfr;                /* 1 / t_h / t_star, used when whalo_sfr != NULL (:340-346) */
    /* X-ray emissivity grid (USE_TS_FLUCT; HaloBox.c:279-283, map_mass.c:231,316-319): filled
     * when both the table and HaloBox.halo_xray are given */
    const float *ln_xray_table; /* host, C21CM_NDELTA_TABLE floats, or NULL */
    double prefactor_xray;      /* rho_crit Omega_m x volume ratio */
    /* USE_MINI_HALOS (HaloBox.c:245-283, map_mass.c:285-321): per source cell the turnover masses
     * of the cell (grids at the OUTPUT resolution, indexed with the source-cell index as upstream
     * does, map_mass.c:291-292: low-resolution sources only), N_ion of both populations from 2-D
     * tables over (delta, log10 M_turn) on the grids' own ranges, the molecularly cooled SFRD and
     * the X-ray luminosity from 2-D tables on the fixed turnover grid; halo_sfr_mini is filled and
     * n_ion sums both populations.  Tables: host, [NDELTA][NMTURN] floats (ln values). */
    int use_mini_halos;
    const float *log10_mturn_acg, *log10_mturn_mcg; /* [N out], host or device */
    const float *ln_nion_table2d, *ln_nion_mini_table2d;
    double mta_min, mta_width, mtm_min, mtm_width;
    const float *ln_sfrd_mini_table2d, *ln_xray_table2d; /* the second one NULL without halo_xray */
    double mt_fixed_min, mt_fixed_width;
    double prefactor_nion_mini, prefactor_sfr_mini;
    /* Halo-catalogue branch (sum_halos_onto_grid, HaloBox.c:518-560; move_halo_galprops,
     * map_mass.c:346-476): every halo of non-zero mass is displaced with the velocities of its
     * Lagrangian cell, gets its properties (set_halo_properties, HaloBox.c:62-102; with mini-halos
     * the turnover masses CIC-read at its position) and is CIC-deposited per unit cell volume; the
     * integrated branch above then adds the sources below the catalogue's mass limit unless
     * `skip_integral` (HaloBox.c:635: M_min >= the limit; the tables are not read then and, with a
     * recombination model, whalo_sfr comes from the halos instead of n_ion). */
    const HaloCatalog *halos; /* NULL: integrated branch only; arrays host or device */
    const struct c21cm_halo_consts *halo_consts;
    int skip_integral;
} c21cm_halobox_spec;

/* the ScalingConstants read by set_halo_properties (scaling_relations.c:29-118,331-500) */
typedef struct c21cm_halo_consts {
    double redshift;
    double fstar_10, alpha_star, sigma_star;
    double alpha_upper, pivot_upper, upper_pivot_ratio;
    double fstar_7, alpha_star_mini, acg_thresh;
    double baryon_ratio; /* OMb / OMm */
    double t_h, t_star, sigma_sfr_lim, sigma_sfr_idx;
    double l_x, l_x_mini, sigma_xray; /* L_X in 1e38 erg/s */
    double fesc_10, fesc_7, alpha_esc, pop2_ion, pop3_ion;
    double mturn_a_nofb, mturn_m_nofb; /* the turnovers without mini-halos */
    int scaling_median;                /* HALO_SCALING_RELATIONS_MEDIAN */
    int upper_stellar_turnover;        /* USE_UPPER_STELLAR_TURNOVER */
    int use_mini_halos, use_xray;      /* USE_MINI_HALOS, USE_TS_FLUCT */
} c21cm_halo_consts;

int c21cm_halobox_grids(const c21cm_halobox_spec *spec, const InitialConditions *ics,
                        HaloBox *grids, void *stream);

/* get_log10_turnovers (HaloBox.c:465-516): the two log10 turnover grids of a USE_MINI_HALOS HaloBox
 * and their averages.  Atomic: max(M_acg, M_TURN, reionisation feedback) -- upstream keeps this one
 * as a RUNNING maximum within each OpenMP thread's share of the cells (:481,497), which `n_threads`
 * reproduces for libgomp's static schedule; molecular: max(M_LW, M_TURN, feedback) per cell.
 * The spec's mturn_m_nofb and first_snapshot are not read; below_z_heat_max: :488-492. */
int c21cm_halobox_turnovers(const c21cm_mturn_spec *spec, double m_turn, int below_z_heat_max,
                            int n_threads, const float *prev_G12, const float *prev_z_reion,
                            const float *J_21_LW, const float *vcb, float *log10_mturn_acg,
                            float *log10_mturn_mcg, double averages[2], void *stream);

/* min and max of n floats (host or device array), e.g. the table range of the above */
int c21cm_grid_minmax(const float *values, size_t n, double out_minmax[2], void *stream);

/* ---- spin-temperature filtering stage --------------------------------------
 * The two filter loops of the spin-temperature calculation, on the same transform passes as
 * the excursion-set loop.
 *
 * c21cm_fill_Rbox_grids: prepare_filter_boxes + fill_Rbox_table (SpinTemperatureBox.c:502-520,
 * 560-636).  `input` [N] is transformed once (r2c, / N); for every radius the spectrum is
 * multiplied by window `filter_type` (HEAT_FILTER) at R -- radii not above `cell_radius`
 * (L_FACTOR BOX_LEN / HII_DIM) are left unfiltered (:585-588) -- transformed back, floored at
 * `min_value` (before the constant factor, :617-620), multiplied by `const_factor` and stored
 * to result[r * N ...]; min_arr / average_arr / max_arr [n_R] get the statistics of the stored
 * values.  Arrays may be host or device. */
#define C21CM_MAX_TS_RADII 128
typedef struct c21cm_rbox_spec {
    int hii_dim, hii_dim_z;
    double box_len, box_len_z;
    int filter_type;
    int n_R;
    double R[C21CM_MAX_TS_RADII];
    double cell_radius;
    double min_value, const_factor;
} c21cm_rbox_spec;

int c21cm_fill_Rbox_grids(const c21cm_rbox_spec *spec, const float *input, float *result,
                          double *min_arr, double *average_arr, double *max_arr, void *stream);

/* c21cm_annular_filter_grids: one_annular_filter (SpinTemperatureBox.c:642-742) for n_grids
 * grids of one shell: r2c, / N, window filter_type[g] (4 = spherical shell, 5 = multiple
 * scattering) between R_inner and R_outer unless R_inner <= 0, c2r, negative values (aliasing)
 * set to zero.  u_avg / f_avg [n_grids]: box averages of the input and of the output. */
#define C21CM_MAX_ANNULAR_GRIDS 5
typedef struct c21cm_annular_spec {
    int hii_dim, hii_dim_z;
    double box_len, box_len_z;
    double R_inner, R_outer, R_star;
    int n_grids;
    int filter_type[C21CM_MAX_ANNULAR_GRIDS];
} c21cm_annular_spec;

int c21cm_annular_filter_grids(const c21cm_annular_spec *spec, const float *const *inputs,
                               float *const *outputs, double *u_avg, double *f_avg,
                               void *stream);

/* ---- spin temperature: the per-cell part of ComputeTsBox ---------------------------------------
 * reference: src/py21cmfast/src/SpinTemperatureBox.c
 *   :892-927    init_first_Ts   (redshift >= Z_HEAT_MAX)      -> c21cm_ts_first_grids
 *   :1010-1086  calculate_sfrd_from_grid                      -> source_mode SFRD_TABLE
 *   :1210-1383  get_Ts_fast     (x_e, T_k update and T_s)     -> the cell epilogue
 *   :1499-1522  x_e interpolation index per cell
 *   :1541-1784  the R loop: SFR of every shell -> dxheat / dxion / dxlya / dstarlya sums
 *   :1794-1848  prefactors of the sums, get_Ts_fast, outputs
 * Everything that depends on the cosmology, on the spectra or on the frequency integrals is a
 * scalar or a small table of this struct (the host builds them: heating.c); the library kernel
 * and the oracle evaluate the same formulae on the same numbers.
 *
 * source_mode C21CM_TS_SRC_GRIDS: the shells' star-formation and X-ray grids are given
 *   (XraySourceBox.filtered_sfr / filtered_xray [n_step][N], Lagrangian source models);
 * source_mode C21CM_TS_SRC_SFRD_TABLE: `filtered_density` [n_step][N] (fill_Rbox_table's
 *   delNL0, i.e. linearly extrapolated to z = 0) is turned into an SFRD per shell through
 *   exp(lerp(ln_sfrd_tables[R], delta zpp_growth[R])) (1 + delta zpp_growth[R]), normalised to
 *   mean_sfr_zpp[R] over the box mean of the table values (avg_fix_term, :1624);
 * source_mode C21CM_TS_SRC_FCOLL_TABLES (CONST-ION-EFF): two linear tables per shell, the
 *   conditional collapsed fraction (`fcoll_tables`, its box mean normalises) and its redshift
 *   derivative (`dfcoll_tables`, the source: (1 + delta) dfcoll/dz, :1067-1072). */
#define C21CM_X_INT_NXHII 14 /* elec_interp.h:5 */
#define C21CM_X_INT_XHII                                                                        \
    { 1.0e-4f, 2.318e-4f, 4.677e-4f, 1.0e-3f, 2.318e-3f, 4.677e-3f, 1.0e-2f, 2.318e-2f, 4.677e-2f, \
      1.0e-1f, 0.5f, 0.9f, 0.99f, 0.999f } /* elec_interp.c:57-70 */
#define C21CM_LYA_NT 101  /* heating_helper_progs.c:50-55: log10 T_k, log10 T_s in [-1, 3] */
#define C21CM_LYA_NGP 51  /* log10 tau_GP in [1, 7] */
#define C21CM_TS_MAX_TK 5e4 /* SpinTemperatureBox.c:30 */
enum { C21CM_TS_SRC_GRIDS = 0, C21CM_TS_SRC_SFRD_TABLE = 1, C21CM_TS_SRC_FCOLL_TABLES = 2 };

typedef struct c21cm_ts_spec {
    int hii_dim, hii_dim_z;
    int n_step;      /* N_STEP_TS, <= C21CM_MAX_TS_RADII */
    int source_mode; /* C21CM_TS_SRC_* */
    int use_xray_heating, use_cmb_heating, use_lya_heating;
    int no_light;    /* global_reion_properties: nothing has formed yet, the sums stay zero */
    double redshift; /* z' */
    double dzp;      /* z' - previous z' (negative) */
    double growth_ratio; /* dicke(z') / dicke(perturbed_field_redshift) */
    /* constants (Constants.c / Constants.h:98-112 with the run's cosmology) */
    double No, N_b0, h_frac, he_frac;
    double k_B, h_p, m_p, c_cms, A10, T_21, lambda_21, nu_Ly_alpha;
    double clumping_factor;
    /* set_zp_consts (:1098-1184) */
    double xray_prefactor, Trad, Ts_prefactor, xa_tilde_prefactor, xc_inverse, dcomp_dzp_prefactor;
    double Nb_zp, N_zp, lya_star_prefactor, volunit_inv, hubble_zp, growth_zp, dgrowth_dzp, dt_dzp;
    /* per shell */
    double z_edge_factor[C21CM_MAX_TS_RADII]; /* :1546-1553 */
    double xray_R_factor[C21CM_MAX_TS_RADII]; /* (1 + z'')^-X_RAY_SPEC_INDEX */
    double starlya_prefactor[C21CM_MAX_TS_RADII];  /* dstarlya_dt_prefactor */
    double lya_cont_prefactor[C21CM_MAX_TS_RADII]; /* dstarlya_cont_dt_prefactor */
    double lya_inj_prefactor[C21CM_MAX_TS_RADII];  /* dstarlya_inj_dt_prefactor */
    /* C21CM_TS_SRC_SFRD_TABLE */
    double zpp_growth[C21CM_MAX_TS_RADII];
    double mean_sfr_zpp[C21CM_MAX_TS_RADII];
    double tab_min[C21CM_MAX_TS_RADII], tab_width[C21CM_MAX_TS_RADII];
    const float *ln_sfrd_tables; /* host, [n_step][C21CM_NDELTA_TABLE] */
    const float *fcoll_tables, *dfcoll_tables; /* C21CM_TS_SRC_FCOLL_TABLES, same shape */
    double sfr_scale;            /* F_STAR10 */
    double xray_scale;           /* L_X s_per_yr */
    /* fill_freqint_tables (:810-889): host, [C21CM_X_INT_NXHII][n_step] each */
    const double *freq_int_heat, *freq_int_ion, *freq_int_lya;
    /* Energy_Lya_heating's two tables, host, [NT][NT][NGP]; needed with use_lya_heating */
    const double *lya_dEC, *lya_dEI;
    /* USE_MINI_HALOS with C21CM_TS_SRC_SFRD_TABLE (SpinTemperatureBox.c:1011-1075,1642-1716): the
     * molecularly cooled population adds, per shell, a star-formation term from a 2-D table
     * (overdensity x log10 of the shell-filtered Lyman-Werner turnover mass), its own Lyman-alpha
     * prefactors and X-ray luminosity, and both populations feed the Lyman-Werner background
     * written to TsBox.J_21_LW (:1843-1845). */
    int use_mini_halos;
    double starlya_prefactor_mini[C21CM_MAX_TS_RADII];
    double lya_cont_prefactor_mini[C21CM_MAX_TS_RADII], lya_inj_prefactor_mini[C21CM_MAX_TS_RADII];
    double lw_prefactor[C21CM_MAX_TS_RADII], lw_prefactor_mini[C21CM_MAX_TS_RADII];
    double mean_sfr_zpp_mini[C21CM_MAX_TS_RADII];
    const float *ln_sfrd_tables_mini; /* host, [n_step][C21CM_NDELTA_TABLE][C21CM_NMTURN_TABLE] */
    double mturn_tab_min, mturn_tab_width; /* LOG10_MTURN_MIN and the bin width (interp_tables.c:29-30) */
    double sfr_scale_mini;  /* F_STAR7_MINI   */
    double xray_scale_mini; /* L_X_MINI s_per_yr */
    const float *filtered_log10_mcrit; /* [n_step][N]: fill_Rbox_table of log10 M_crit,LW (:1459-1466) */
} c21cm_ts_spec;

typedef struct c21cm_ts_report { /* box means, as the reference logs them at DEBUG level */
    double Ts_ave, Tk_ave, x_e_ave, J_alpha_ave, xheat_ave, xion_ave;
    double ave_sfrd[C21CM_MAX_TS_RADII]; /* SFRD_TABLE: mean table value per shell */
    double ave_sfrd_mini[C21CM_MAX_TS_RADII]; /* USE_MINI_HALOS */
} c21cm_ts_report;

/* prepare_filter_boxes with USE_MINI_HALOS (SpinTemperatureBox.c:535-565): per cell
 * log10(max(M_LW(z, J_21_LW, v_cb), M_TURN)) from the previous TsBox's J_21_LW (vcb NULL:
 * vcb_const); the spec's reionisation / no-feedback fields are not read. */
int c21cm_ts_mcrit_grid(const c21cm_mturn_spec *spec, double m_turn, const float *J_21_LW,
                        const float *vcb, float *log10_mcrit, void *stream);

/* `density` [N]: PerturbedField.density; `previous` holds the three boxes of the previous
 * snapshot; `source_box` (GRIDS) or `filtered_density` (SFRD_TABLE) as described above; `out`
 * receives spin_temperature, kinetic_temp_neutral, xray_ionised_fraction.  Arrays may be host
 * or device.  A non-finite spin temperature gives C21CM_INFINITY_OR_NAN_ERROR (:1884-1904). */
int c21cm_ts_grids(const c21cm_ts_spec *spec, const float *density, const TsBox *previous,
                   const XraySourceBox *source_box, const float *filtered_density, TsBox *out,
                   c21cm_ts_report *report, void *stream);
/* the cell part alone (ts_driver.c): the shell loop of a spec's shells into sums_dev ([6][N]), and
 * the temperature update of cells [cell0, cell0 + ncell) from sums ([6][ncell]) */
int c21cm_ts_shell_sums(const c21cm_ts_spec *spec, const float *density, const TsBox *previous,
                        const XraySourceBox *source_box, const float *filtered_density,
                        double *sums_dev, void *stream);
int c21cm_ts_cells_from_sums(const c21cm_ts_spec *spec, const float *density, const TsBox *previous,
                             const double *sums_dev, size_t cell0, size_t ncell, TsBox *out,
                             void *stream);

/* init_first_Ts: T_k = TK (1 + cT_ad delta), x_e = xe, T_s from collisions only. */
typedef struct c21cm_ts_first_spec {
    int hii_dim, hii_dim_z;
    double redshift;          /* z' of the requested box */
    double perturbed_redshift; /* get_Ts is called with this one (:923) */
    float inverse_growth_factor_z, growth_factor_zp; /* floats upstream (:896-897) */
    double xe, TK, cT_ad;
    double No, N_b0, A10, T_21, T_cmb;
} c21cm_ts_first_spec;
int c21cm_ts_first_grids(const c21cm_ts_first_spec *spec, const float *density, TsBox *out,
                         void *stream);

/* Library management */
const char *c21cm_version(void);
int c21cm_device_synchronize(void);
void c21cm_release_device_cache(void); /* drop cached rocFFT plans and scratch */
const char *c21cm_last_error(void);

/* Placement of the second work spectrum of a two-grid sweep (csrc/host/placement.c): what the last decision was.
 * out = {outcome (0 placed by timed launches, 1 off / not applicable, 2 other tenants on the device, 3 another
 * process walking, 4 nothing faster within the budget, 5 remembered failure, 6 tenancy unknown + busy device, 7 time budget exhausted),
 * GB held at the peak of the walk, timed probes, ms of the chosen pair, ms of the first candidate, wall ms of the
 * decision, tenants seen (-1 unknown), walks of this process so far}.  C21CM_VALUE_ERROR before any decision. */
int c21cm_placement_report(double out[8]);
/* The walk is opt-in: 1 on, 2 on even with other tenants on the device, 0 off, -1 the environment's C21CM_WS_PLACE
 * (unset: off).  Changing the setting re-opens the decisions taken under the old one. */
int c21cm_placement_set(int mode);

#ifdef __cplusplus
}
#endif
#endif /* C21CM_GRID_H */
