/*
 * ionize_driver.c -- C host driver of the ComputeIonizedBox grid algorithm on MI355X.
 *
 * Orchestrates the HIP launchers of csrc/hip/c21hip.h in the order of the
 * reference's ComputeIonizedBox (src/py21cmfast/src/IonisationBox.c:1477-1628):
 *   pre-loop   pack+clip -> r2c -> /N for every grid that is filtered   (:323-360)
 *   R loop     fused copy x W(kR) -> c2r per grid                       (:572-664)
 *              [Eulerian] min/max -> host table -> f_coll + sum         (:702-962)
 *              barrier / partial-ionisation sweep                       (:1008-1201)
 *   post-loop  ionised temperatures, sum(xH)                            (:1203-1256,1597-1608)
 * Everything stays resident in HBM for the whole call; the only host round trips
 * inside the loop are the two doubles + 400-float table of the TABLE modes, which
 * the reference's design requires (the table range depends on the filtered extrema).
 *
 * Data layout in HBM: 2*G padded k-space grids (unfiltered spectra kept for all
 * radii + one filtered working copy each), the dense inputs/outputs, and a small
 * block of doubles for reductions.  Host arrays (numpy through CFFI) are staged
 * into workspace slots; device arrays (torch) are used in place.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

/* workspace slots */
enum {
    WS_DELTA_UNF = 0,
    WS_DELTA_FIL,
    WS_STARS_UNF,
    WS_STARS_FIL,
    WS_XE_UNF,
    WS_XE_FIL,
    WS_DENSITY,
    WS_NION,
    WS_XE_DENSE,
    WS_TNEUTRAL,
    WS_PREV_ZRE,
    WS_XH,
    WS_ZRE,
    WS_TK,
    WS_NION_DENSE,
    WS_SCALARS,
    WS_TABLE,
    WS_FIRST_CROSS,
    WS_DELTA_WORK,
    WS_STARS_WORK,
    WS_XE_WORK,
    WS_PARTIALS,
    WS_DEF_PARTIALS = 84,
    WS_DELTA_WORK2 = 92, /* second radius of a two-radius sweep */
    WS_STARS_WORK2 = 93,
    WS_XE_WORK2 = 96,
    WS_EUL_DFIL2 = 97, /* Eulerian table loop: second delta_R buffer, two dense x_e(R) buffers */
    WS_EUL_XE0 = 98,
    WS_EUL_XE1 = 99,
    /* recombination models: filtered whalo_sfr and N_rec grids, staged arrays, rate tables */
    WS_SFR_UNF = 120,
    WS_SFR_FIL,
    WS_SFR_WORK,
    WS_NREC_UNF,
    WS_NREC_FIL,
    WS_NREC_WORK,
    WS_WSFR,
    WS_PREV_NREC,
    WS_G12,
    WS_MFP,
    WS_NREC_OUT,
    WS_RR_TABLES,
    /* rank-local state of a sharded run with a recombination model */
    WS_SH_XH,
    WS_SH_ZRE,
    WS_SH_G12,
    WS_SH_MFP,
    /* USE_MINI_HALOS: previous delta and the two turnover-mass grids (spectra, scratch, filtered),
     * staged inputs, 2-D tables, per-radius f_coll history in and out */
    WS_MINI_PD_UNF = 180,
    WS_MINI_PD_WORK,
    WS_MINI_PD_FIL,
    WS_MINI_MTA_UNF,
    WS_MINI_MTA_WORK,
    WS_MINI_MTA_FIL,
    WS_MINI_MTM_UNF,
    WS_MINI_MTM_WORK,
    WS_MINI_MTM_FIL,
    WS_MINI_PDENS,
    WS_MINI_MTA,
    WS_MINI_MTM,
    WS_MINI_TABLES,
    WS_MINI_HIST_A,
    WS_MINI_HIST_M,
    WS_MINI_OUT_A,
    WS_MINI_OUT_M,
    WS_SPHERE_RSQ = 216,
    WS_SFR_WORK2 = 246, /* fused recombination loop: whalo_sfr of the second radius of a sweep */
    WS_R_DEV = 247,     /* float R per radius index (mean free path of a first crossing) */
    WS_EUL_XEPEND = 253, /* banded barrier with an x_e grid: clipped x_e of the undecided cells (sparse) */
    WS_NION_DENSE2 = 254, /* closed-form Eulerian loop: second dense f_coll buffer (deferred barrier) */
    /* (256 and 257 are shard_rccl.c's: WS_SHARD_STATUS, WS_SHARD_SLABBITS) */
    WS_NREC_WORK2 = 259, /* fused recombination loop with x_e AND a filtered N_rec: N_rec of the second radius */
    WS_EUL_WORK3 = 261,   /* Eulerian table loop, two radii per pass-X sweep: the second set of k-space buffers */
    WS_EUL_WORK4 = 262,
    WS_ARENA = 258        /* experiment: the spectra of the two-grid loop out of one allocation (C21CM_ARENA) */
};

#define MAX_COPYBACK 12
typedef struct {
    void *host[MAX_COPYBACK];
    void *dev[MAX_COPYBACK];
    size_t bytes[MAX_COPYBACK];
    int n;
} copyback_list;

/* device scalar block layout (doubles) */
#define SC_SUMS 0
#define SC_MEANS (SC_SUMS + C21CM_MAX_RADII)
#define SC_MINMAX (SC_MEANS + C21CM_MAX_RADII)
#define SC_XHSUM (SC_MINMAX + 4) /* two (min, max) pairs: the table loop is double-buffered */
#define SC_FLAG (SC_XHSUM + 1) /* an int stored in a double-sized cell */
#define SC_G12SUM (SC_FLAG + 1)
/* USE_MINI_HALOS: sums / means of the molecularly cooled f_coll per radius, extrema of the
 * previous delta and the two turnover grids, scratch pair of a two-sum reduction */
#define SC_SUMS_M (SC_G12SUM + 1)
#define SC_MEANS_M (SC_SUMS_M + C21CM_MAX_RADII)
#define SC_MINMAX_M (SC_MEANS_M + C21CM_MAX_RADII)
#define SC_PAIR (SC_MINMAX_M + 6)
/* closed-form Eulerian loop, banded barrier: the two thresholds of a radius' band, failure flag */
#define SC_BAND (SC_PAIR + 2)
#define SC_BANDX (SC_BAND + 2 * C21CM_MAX_RADII) /* exact threshold of every radius */
#define SC_BANDFAIL (SC_BANDX + C21CM_MAX_RADII) /* an int stored in a double-sized cell */
#define SC_BANDCTR (SC_BANDFAIL + 1) /* arrival counter of eul_sum_band_kernel (an unsigned, zero between launches) */
#define SC_BANDP (SC_BANDCTR + 1)    /* the mean the two rules predicted for every radius (0: none), the error measure after it */
#define SC_COUNT (SC_BANDP + 3 * C21CM_MAX_RADII)

#define TRY(expr)                   \
    do {                            \
        int st_ = (expr);           \
        if (st_) {                  \
            status = st_;           \
            goto done;              \
        }                           \
    } while (0)

static const float *stage_in(int slot, const float *p, size_t bytes, void *stream, int *status) {
    if (!p || *status) return NULL;
    if (c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    int st = c21hip_h2d(d, p, bytes, stream);
    if (st) *status = st;
    return (const float *)d;
}

/* in/out array: upload the caller's initial contents, remember to copy back */
static float *stage_inout(int slot, float *p, size_t bytes, int upload, copyback_list *cb,
                          void *stream, int *status) {
    if (!p || *status) return NULL;
    if (c21hip_is_device_ptr(p)) return p;
    void *d = c21hip_ws(slot, bytes);
    if (!d) {
        *status = C21CM_MEMORY_ALLOC_ERROR;
        return NULL;
    }
    if (upload) {
        int st = c21hip_h2d(d, p, bytes, stream);
        if (st) *status = st;
    }
    if (cb->n < MAX_COPYBACK) {
        cb->host[cb->n] = p;
        cb->dev[cb->n] = d;
        cb->bytes[cb->n] = bytes;
        cb->n++;
    }
    return (float *)d;
}

/* (R in cells)^2 as update_in_sphere forms it: float R = rspec.R / BOX_LEN, float product with
 * the grid size, squared in double, stored as float (IonisationBox.c:1153-1158,
 * bubble_helper_progs.c:342,384-385) */
static float sphere_rsq(const c21cm_ionize_spec *s, int r) {
    const float Rf = (float)(s->R[r] / (double)(float)s->box_len);
    const float Rd = Rf * s->hii_dim;
    return (float)pow((double)Rd, 2);
}

static int validate_spec(const c21cm_ionize_spec *s, const PerturbedField *pf,
                         const HaloBox *halos, const TsBox *ts, const IonizedBox *box) {
    if (!s || !pf || !box) {
        c21hip_set_error("ionize: NULL spec / perturbed_field / box");
        return C21CM_VALUE_ERROR;
    }
    if (s->hii_dim < 2 || s->hii_dim_z < 2 || s->n_radii < 1 || s->n_radii > C21CM_MAX_RADII ||
        s->n_radii > 255 || s->r_lowest < 0) {
        c21hip_set_error("ionize: bad geometry or radius count (dim %d/%d, n_radii %d)",
                         s->hii_dim, s->hii_dim_z, s->n_radii);
        return C21CM_VALUE_ERROR;
    }
    if (s->recomb_model != C21CM_RECOMB_NONE) {
        if (s->recomb_model != C21CM_RECOMB_HOMOGENEOUS &&
            s->recomb_model != C21CM_RECOMB_INHOMOGENEOUS) {
            c21hip_set_error("ionize: unknown recomb_model %d", s->recomb_model);
            return C21CM_VALUE_ERROR;
        }
        if (!s->rr_y || !s->rr_c) {
            c21hip_set_error("ionize: a recombination model needs the rr_y / rr_c rate tables");
            return C21CM_VALUE_ERROR;
        }
        if (!box->ionisation_rate_G12 || !box->cumulative_recombinations) {
            c21hip_set_error("ionize: a recombination model needs ionisation_rate_G12 and "
                             "cumulative_recombinations");
            return C21CM_VALUE_ERROR;
        }
        if (s->recomb_model == C21CM_RECOMB_HOMOGENEOUS && !s->cell_recomb) {
            /* the homogeneous N_rec is one number: there is no grid to filter (inputs.py) */
            c21hip_set_error("ionize: RECOMB_MODEL = homogeneous needs CELL_RECOMB");
            return C21CM_VALUE_ERROR;
        }
        if (s->fcoll_mode == C21CM_FCOLL_STARS_GRID && (!halos || !halos->whalo_sfr)) {
            c21hip_set_error("ionize: Lagrangian sources with recombinations need HaloBox.whalo_sfr");
            return C21CM_VALUE_ERROR;
        }
    }
    if (s->fcoll_mode < C21CM_FCOLL_STARS_GRID || s->fcoll_mode > C21CM_FCOLL_NODES) {
        c21hip_set_error("ionize: unknown fcoll_mode %d", s->fcoll_mode);
        return C21CM_VALUE_ERROR;
    }
    if (s->use_mini_halos && s->fcoll_mode == C21CM_FCOLL_STARS_GRID) {
        /* Lagrangian grids: the mini-halos are inside HaloBox.n_ion, only f_limit_mcg enters */
    } else if (s->use_mini_halos) {
        if (s->fcoll_mode != C21CM_FCOLL_TABLE_EXP || !s->table2d_fn) {
            c21hip_set_error("ionize: USE_MINI_HALOS runs on the E-INTEGRAL tables (fcoll_mode "
                             "TABLE_EXP) and needs table2d_fn");
            return C21CM_VALUE_ERROR;
        }
        if (!s->prev_density || !s->log10_mturn_acg || !s->log10_mturn_mcg ||
            !box->unnormalised_nion_mini) {
            c21hip_set_error("ionize: USE_MINI_HALOS needs prev_density, the two log10 M_turn "
                             "grids and IonizedBox.unnormalised_nion_mini");
            return C21CM_VALUE_ERROR;
        }
    } else if (s->fcoll_mode >= C21CM_FCOLL_TABLE_LINEAR && !s->table_fn) {
        c21hip_set_error("ionize: TABLE fcoll_mode needs table_fn");
        return C21CM_VALUE_ERROR;
    }
    if (s->ionise_entire_sphere) {
        if (s->recomb_model != C21CM_RECOMB_NONE || s->use_mini_halos) {
            c21hip_set_error("ionize: IONISE_ENTIRE_SPHERE with a recombination model or mini-halos "
                             "is not built (upstream's result depends on its thread order there)");
            return C21CM_VALUE_ERROR;
        }
        if (sphere_rsq(s, 0) > 1.f) {
            c21hip_set_error("ionize: IONISE_ENTIRE_SPHERE needs a cell-scale radius below one cell");
            return C21CM_VALUE_ERROR;
        }
    }
    if (!pf->density || !box->neutral_fraction || !box->z_reion) {
        c21hip_set_error("ionize: density / neutral_fraction / z_reion arrays are required");
        return C21CM_VALUE_ERROR;
    }
    if (s->fcoll_mode == C21CM_FCOLL_STARS_GRID && (!halos || !halos->n_ion)) {
        c21hip_set_error("ionize: Lagrangian source model needs HaloBox.n_ion");
        return C21CM_VALUE_ERROR;
    }
    if (s->fcoll_mode != C21CM_FCOLL_STARS_GRID && !box->unnormalised_nion) {
        c21hip_set_error("ionize: Eulerian source model needs IonizedBox.unnormalised_nion");
        return C21CM_VALUE_ERROR;
    }
    if (s->use_ts_fluct &&
        (!ts || !ts->xray_ionised_fraction || (!s->minimize_memory && !ts->kinetic_temp_neutral))) {
        c21hip_set_error("ionize: USE_TS_FLUCT needs the TsBox arrays");
        return C21CM_VALUE_ERROR;
    }
    if (!s->minimize_memory && !box->kinetic_temperature) {
        c21hip_set_error("ionize: kinetic_temperature is required unless MINIMIZE_MEMORY");
        return C21CM_VALUE_ERROR;
    }
    return 0;
}

static void fill_args(c21hip_ionize_args *a, const c21cm_ionize_spec *s, int r_index) {
    memset(a, 0, sizeof(*a));
    a->nx = s->hii_dim;
    a->ny = s->hii_dim;
    a->nz = s->hii_dim_z;
    a->r_index = r_index;
    a->lagrangian = (s->fcoll_mode == C21CM_FCOLL_STARS_GRID);
    a->mass_dep_zeta = s->mass_dep_zeta;
    a->use_ts_fluct = s->use_ts_fluct;
    a->minimize_memory = s->minimize_memory;
    a->first_snapshot = s->first_snapshot;
    a->fix_mean = s->fix_mean;
    a->mean_f_coll = s->mean_f_coll;
    a->f_limit = s->f_limit_acg;
    a->ion_eff_factor = s->ion_eff_factor;
    a->rhocrit_omb = s->rhocrit_omb;
    a->photoncons_factor = s->photoncons_adjustment_factor;
    a->redshift = s->redshift;
    a->TK_nofluct = s->TK_nofluct;
    a->adia_TK_term = s->adia_TK_term;
    a->T_re = s->T_re;
}

/* State shared by the single-GPU driver and the two shard phases. */
typedef struct {
    const c21cm_ionize_spec *s;
    void *stream;
    int nx, ny, nz;
    size_t ntot, npad;
    int lagrangian;
    int native; /* hand-written split-layout FFT with the filter fused into pass X */
    /* k-space grids: *_unf unfiltered spectra (padded layout with rocFFT, split layout with
     * the native FFT), *_work split-layout scratch of the native passes, *_fil filtered
     * real-space grids (padded rows) */
    float *delta_unf, *delta_fil, *stars_unf, *stars_fil, *xe_unf, *xe_fil;
    float *delta_work, *stars_work, *xe_work;
    float *delta_work2, *stars_work2, *xe_work2; /* two radii per pass-X sweep (pair_radii) */
    /* recombination models (unfused per-radius sequence) */
    int recomb, inhomo, filter_rec;
    float *sfr_unf, *sfr_fil, *sfr_work, *nrec_unf, *nrec_fil, *nrec_work;
    const float *whalo_sfr, *prev_nrec;
    float *G12, *mfp, *nrec_out;
    double *rr_dev; /* rr_y then rr_c */
    int pair_radii;
    /* dense inputs */
    const float *density, *n_ion, *xe_dense, *Tneutral, *prev_zre;
    /* dense outputs */
    float *xH, *zre, *Tk, *nion_dense;
    double *scalars;
    double *partials; /* >= max(C21HIP_PARTIALS, nx*ny/8) doubles */
    float *table_dev;
    unsigned char *mask; /* internal first-crossing mask of the fused single-GPU path */
    int fused;           /* fused pass Z + barrier available for radius index > 0 */
    /* deferred f_coll sums of the fused radii: partials of radius R at def_partials + R *
     * def_stride, radii def_first, def_first - def_step, ... (def_count of them) */
    double *def_partials;
    long def_stride;
    int def_first, def_step, def_count;
    int tab_seq;         /* fused radii done so far: the window-table buffer alternates */
    int wev;             /* 1: this loop's passes X evaluate their windows in the kernel */
    int fused_rc;        /* fused loop with a recombination model (CELL_RECOMB, no x_e grid) */
    /* closed-form Eulerian loop: the barrier of a radius rides the NEXT radius' pass Z (EPI 6) */
    int eul_pend, eul_pend_buf;      /* radius index whose barrier is still owed (-1: none), its f_coll buffer */
    unsigned char *eul_pend_mask;
    float *nion_dense2;
    /* closed-form Eulerian loop, banded barrier (pass Z EPI 7): the barrier of a radius decided inside
     * its own pass Z from a predicted band of the mean fix; see eul_band_ok() */
    int band_off;        /* 1: a band missed in this call -- the loop reruns on the dense sweeps */
    int band_used;       /* a banded sweep ran: the failure flag is read before the mask is used */
    int band_next;       /* radius index the device holds a band for (-1: none) */
    int band_pend;       /* radius whose markers (255) are outstanding in band_mask (-1: none) */
    int band_h1, band_h2, band_hn; /* the last two radii of this loop with a mean on the device */
    unsigned char *band_mask;
    const unsigned char *r0_mask; /* Eulerian loops: the first crossings of the larger radii, applied by the ONE
                                   * sweep that also does the cell-scale radius and the post-loop (NULL: the
                                   * general kernels) */
    int r0_slab, r0_cb, r0_ce; /* sharded finish by slabs: the one sweep covers chunks [r0_cb, r0_ce) only, no reduce */
    int band_mf;         /* 1: this loop's bands live in mean-fix space (barriers with an x_e grid) */
    int band_skip;       /* radius index that takes the dense sweeps whatever band exists (-1: none) */
    short band_hist[C21CM_MAX_RADII][3]; /* band_h1 / _h2 / _hn as they were BEFORE radius r was processed */
    float *band_xe_pend; /* ... and its undecided cells leave their clipped x_e here */
    const float *cur_xe; /* fused recombination loop with an x_e grid: its work spectrum of the radius in hand */
    /* third HII-window spectrum of the fused loop: the x_e grid, or -- on the fused recombination loop
     * with CELL_RECOMB = false -- the previous snapshot's N_rec (x3_nrec) */
    int x3_on, x3_nrec;
    float *x3_unf, *x3_work, *x3_work2;
    /* ... and a fourth: N_rec when the third is the x_e grid (round 5; barrier kernel with the N_rec
     * transform parked in LDS) */
    int x4_on;
    float *x4_unf, *x4_work, *x4_work2;
    const float *cur_x4;
    float *sfr_work2;
    double rec0;         /* homogeneous model: the one previous N_rec */
    int finalised;       /* the post-loop sweep already ran inside final_step() */
    float *eul_xe[2];    /* dense x_e(R) of the Eulerian mask path (spin-temperature runs) */
    /* Eulerian loops, one filtered grid: pass X serves two radii per sweep where the evaluated windows allow it
     * (round 6, eul_filter_density): the spectrum of radius pairx_R waits in pairx_buf for its pass Y */
    int wev_pair, pairx_R;
    float *pairx_buf;
    int sphere;          /* IONISE_ENTIRE_SPHERE: radii > 0 only record the mask, spheres follow */
    /* USE_MINI_HALOS: Eulerian tables + history (mini) | Lagrangian grids, floor only (lag_mini) */
    int mini, lag_mini;
    float *pd_unf, *pd_work, *pd_fil, *mta_unf, *mta_work, *mta_fil, *mtm_unf, *mtm_work, *mtm_fil;
    const float *prev_density, *mta_dense, *mtm_dense, *hist_a, *hist_m;
    float *mini_tables, *nion_all, *mini_all;
    int eul_mask;        /* Eulerian models on the native passes: radii > 0
                          * run pass Z fused with f_coll (or its extrema) and only update the
                          * first-crossing mask */
    copyback_list cb;
    /* sharded finish by cell slabs: the host copies of staged per-cell outputs cover [cb_cell0, cb_cell0 +
     * cb_ncell) only (cb_ncell = 0: the whole arrays) */
    size_t cb_cell0, cb_ncell;
} ion_ctx;

static int r0_direct(void);

/* which R loop the last context set up (tests, timing tools): 1 fused, 2 fused recombination loop, 4 third
 * spectrum, 8 ... which is N_rec, 16 fourth spectrum (x_e + N_rec), 32 two radii per sweep */
static int g_loop_flags;
int c21cm_ionize_last_loop_flags(void) { return g_loop_flags; }
static int g_single_pass; /* set by c21cm_ionize_grids around its ctx_setup: not a shard phase */
static int g_rc_phase;    /* set by the (first crossing, Gamma_12) shard phases around their ctx_setup */

static int ctx_setup(ion_ctx *c, const c21cm_ionize_spec *s, const PerturbedField *pf,
                     const IonizedBox *prev, const TsBox *ts, const HaloBox *halos,
                     IonizedBox *box, int need_outputs, void *stream) {
    int status = 0;
    memset(c, 0, sizeof(*c));
    c->s = s;
    c->stream = stream;
    c->nx = s->hii_dim;
    c->ny = s->hii_dim;
    c->nz = s->hii_dim_z;
    c->ntot = (size_t)c->nx * c->ny * c->nz;
    c->npad = (size_t)c->nx * c->ny * 2 * (size_t)(c->nz / 2 + 1);
    c->lagrangian = (s->fcoll_mode == C21CM_FCOLL_STARS_GRID);
    const size_t gbytes = c->npad * sizeof(float), dbytes = c->ntot * sizeof(float);

    c->native = c21hip_fft_is_native(c->nx, c->ny, c->nz);
    /* experiment (round 5): the four spectra of the two-grid loop out of ONE allocation, `skew` bytes
     * between consecutive buffers (C21CM_ARENA=skew; default: separate workspace slots) */
    char *arena = NULL;
    size_t astep = 0;
    {
        const char *e = getenv("C21CM_ARENA");
        if (e && c->lagrangian) {
            const size_t skew = (size_t)atol(e);
            astep = ((gbytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) + skew;
            arena = (char *)c21hip_ws(WS_ARENA, 4 * astep);
            if (!arena) return C21CM_MEMORY_ALLOC_ERROR;
        }
    }
    c->delta_unf = arena ? (float *)arena : (float *)c21hip_ws(WS_DELTA_UNF, gbytes);
    c->delta_fil = (float *)c21hip_ws(WS_DELTA_FIL, gbytes);
    if (!c->delta_unf || !c->delta_fil) return C21CM_MEMORY_ALLOC_ERROR;
    if (c->native && !(c->delta_work = arena ? (float *)(arena + 2 * astep) : (float *)c21hip_ws(WS_DELTA_WORK, gbytes)))
        return C21CM_MEMORY_ALLOC_ERROR;
    if (c->lagrangian) {
        c->stars_unf = arena ? (float *)(arena + astep) : (float *)c21hip_ws(WS_STARS_UNF, gbytes);
        c->stars_fil = (float *)c21hip_ws(WS_STARS_FIL, gbytes);
        if (!c->stars_unf || !c->stars_fil) return C21CM_MEMORY_ALLOC_ERROR;
        if (c->native && !(c->stars_work = arena ? (float *)(arena + 3 * astep)
                                                 : c21_place_work_partner(WS_DELTA_WORK, WS_STARS_WORK, gbytes, c->nx, c->ny,
                                                                      c->nz, stream)))
            return C21CM_MEMORY_ALLOC_ERROR;
    }
    if (s->use_ts_fluct) {
        c->xe_unf = (float *)c21hip_ws(WS_XE_UNF, gbytes);
        c->xe_fil = (float *)c21hip_ws(WS_XE_FIL, gbytes);
        if (!c->xe_unf || !c->xe_fil) return C21CM_MEMORY_ALLOC_ERROR;
        /* (Eulerian sources: delta and x_e ride one two-grid sweep -- the x_e work spectrum placed against delta's) */
        if (c->native && !(c->xe_work = !c->lagrangian ? c21_place_work_partner(WS_DELTA_WORK, WS_XE_WORK, gbytes, c->nx,
                                                                                c->ny, c->nz, stream)
                                                       : (float *)c21hip_ws(WS_XE_WORK, gbytes)))
            return C21CM_MEMORY_ALLOC_ERROR;
    }
    {
        size_t np = (size_t)c->nx * c->ny / 4; /* the most partials a pass-Z variant writes */
        np += np / 512 + 16; /* stage areas of the two-level reductions */
        if (np < C21HIP_PARTIALS) np = C21HIP_PARTIALS;
        c->partials = (double *)c21hip_ws(WS_PARTIALS, np * sizeof(double));
        if (!c->partials) return C21CM_MEMORY_ALLOC_ERROR;
    }
    /* Lagrangian grids without the x_e grid: passes Z of both grids, the f_coll sum and the
     * barrier test run as one kernel that only updates a uint8 first-crossing mask */
    /* with an x_e grid (spin-temperature runs) the fused path needs the three-grid pass Z
     * (512/1024-point z-lines) and the dense-input final sweep */
    c->recomb = (s->recomb_model != C21CM_RECOMB_NONE);
    c->inhomo = (s->recomb_model == C21CM_RECOMB_INHOMOGENEOUS);
    c->filter_rec = c->recomb && !s->cell_recomb; /* IonisationBox.c:156-157 */
    /* (recombination models take the unfused sequence: up to five filtered grids per radius and
     * per-cell Gamma_12 / mean-free-path outputs do not fit the fused pass Z) */
    c->fused = c->native && c->lagrangian && !c->recomb &&
               (!s->use_ts_fluct ||
                (c21hip_z_ionise_xe_supported(c->nx, c->ny, c->nz) && r0_direct()));
    /* Recombination models with CELL_RECOMB and no x_e grid ride the fused loop too (round 3): the
     * filtered whalo_sfr is a third spectrum of passes X / Y and of the wave-level pass Z, the
     * barrier gains (1 + N_rec / (1 + delta)), Gamma_12 is written at first crossings and the
     * mean free path follows from the first-crossing index; the cell-scale radius and the
     * post-loop stay on the general kernels.  Needs the evaluated windows (whalo_sfr takes window
     * b alone).  C21CM_RECOMB_FUSED=0: the unfused per-radius sequence. */
    c->fused_rc = 0;
    /* (round 4, late: CELL_RECOMB = false without an x_e grid as well -- N_rec filtered at the radius is
     * the barrier kernel's third line where the x_e grid would be; single pass only.
     * C21CM_RECOMB_FUSED_NREC=0 keeps such runs on the unfused sequence) */
    const char *e_nr = getenv("C21CM_RECOMB_FUSED_NREC");
    const int nrec_ok = s->cell_recomb ||
                        (c->inhomo && (g_single_pass || g_rc_phase) && !(e_nr && e_nr[0] == '0') &&
                         (s->use_ts_fluct ? c21hip_z_ionise_recomb_xe_nrec_supported(c->nx, c->ny, c->nz)
                                          : c21hip_z_ionise_recomb_xe_supported(c->nx, c->ny, c->nz)));
    c->x3_on = c->x3_nrec = 0;
    c->x3_unf = c->x3_work = c->x3_work2 = NULL;
    c->x4_on = 0;
    c->x4_unf = c->x4_work = c->x4_work2 = NULL;
    c->cur_x4 = NULL;
    if (c->native && c->lagrangian && c->recomb && nrec_ok &&
        !s->use_mini_halos && !s->ionise_entire_sphere && s->r_lowest == 0 &&
        (g_single_pass || g_rc_phase)) {
        /* (round 4: with the x_e grid of a spin-temperature run too -- a third line of the barrier
         * kernel; C21CM_RECOMB_FUSED_TS=0 keeps such runs on the unfused sequence) */
        const char *e = getenv("C21CM_RECOMB_FUSED"), *et = getenv("C21CM_RECOMB_FUSED_TS");
        const int ts_ok = !s->use_ts_fluct ||
                          (!(et && et[0] == '0') && r0_direct() &&
                           c21hip_z_ionise_recomb_xe_supported(c->nx, c->ny, c->nz));
        if (!(e && e[0] == '0') && ts_ok && c21hip_z_ionise_recomb_supported(c->nx, c->ny, c->nz) &&
            c21hip_wev_applicable(s->hii_filter, s->stars_filter, 2, c->nx, c->ny, c->nz))
            c->fused_rc = c->fused = 1;
    }
    c->eul_pend = -1;
    c->eul_pend_buf = 0;
    c->eul_pend_mask = NULL;
    c->wev_pair = 0;
    c->pairx_R = -1;
    c->pairx_buf = NULL;
    c->nion_dense2 = NULL;
    c->band_off = c->band_used = 0;
    c->band_next = c->band_pend = c->band_h1 = c->band_h2 = -1;
    c->band_hn = 0;
    c->band_mask = NULL;
    c->band_mf = 0;
    c->r0_mask = NULL;
    c->band_skip = -1;
    c->band_xe_pend = NULL;
    c->sphere = s->ionise_entire_sphere;
    c->mini = s->use_mini_halos && !c->lagrangian;
    c->lag_mini = s->use_mini_halos && c->lagrangian;
    if (c->lag_mini) c->fused = 0; /* the generic sequence carries the extra barrier term */
    c->eul_mask = c->native && !c->lagrangian && !c->recomb && !c->mini;
    if (c->mini) { /* four filtered grids per radius and the 2-D tables: the unfused sequence */
        const int slots[9] = {WS_MINI_PD_UNF, WS_MINI_PD_WORK, WS_MINI_PD_FIL,
                              WS_MINI_MTA_UNF, WS_MINI_MTA_WORK, WS_MINI_MTA_FIL,
                              WS_MINI_MTM_UNF, WS_MINI_MTM_WORK, WS_MINI_MTM_FIL};
        float **dst[9] = {&c->pd_unf, &c->pd_work, &c->pd_fil, &c->mta_unf, &c->mta_work,
                          &c->mta_fil, &c->mtm_unf, &c->mtm_work, &c->mtm_fil};
        for (int i = 0; i < 9; i++) {
            if (!c->native && i % 3 == 1) continue; /* split-layout scratch */
            if (!(*dst[i] = (float *)c21hip_ws(slots[i], gbytes))) return C21CM_MEMORY_ALLOC_ERROR;
        }
        c->mini_tables = (float *)c21hip_ws(
            WS_MINI_TABLES, 4 * sizeof(float) * C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE);
        if (!c->mini_tables) return C21CM_MEMORY_ALLOC_ERROR;
    }
    if (c->filter_rec) {
        c->nrec_unf = (float *)c21hip_ws(WS_NREC_UNF, gbytes);
        c->nrec_fil = (float *)c21hip_ws(WS_NREC_FIL, gbytes);
        if (!c->nrec_unf || !c->nrec_fil) return C21CM_MEMORY_ALLOC_ERROR;
        if (c->native && !(c->nrec_work = (float *)c21hip_ws(WS_NREC_WORK, gbytes)))
            return C21CM_MEMORY_ALLOC_ERROR;
    }
    if (c->recomb && c->lagrangian) {
        c->sfr_unf = (float *)c21hip_ws(WS_SFR_UNF, gbytes);
        c->sfr_fil = (float *)c21hip_ws(WS_SFR_FIL, gbytes);
        if (!c->sfr_unf || !c->sfr_fil) return C21CM_MEMORY_ALLOC_ERROR;
        /* (on the fused loop whalo_sfr shares a two-grid sweep with the x_e / N_rec spectrum: placed against it) */
        const int x3_slot = s->use_ts_fluct ? WS_XE_WORK : (c->filter_rec ? WS_NREC_WORK : -1);
        if (c->native &&
            !(c->sfr_work = (c->fused_rc && x3_slot >= 0)
                                ? c21_place_work_partner(x3_slot, WS_SFR_WORK, gbytes, c->nx, c->ny, c->nz, stream)
                                : (float *)c21hip_ws(WS_SFR_WORK, gbytes)))
            return C21CM_MEMORY_ALLOC_ERROR;
    }
    if (c->recomb) {
        const size_t tb = sizeof(double) * (size_t)C21CM_RR_NZ * C21CM_RR_NGAMMA;
        c->rr_dev = (double *)c21hip_ws(WS_RR_TABLES, 2 * tb);
        if (!c->rr_dev) return C21CM_MEMORY_ALLOC_ERROR;
        status = c21hip_h2d(c->rr_dev, s->rr_y, tb, stream);
        if (!status)
            status = c21hip_h2d(c->rr_dev + (size_t)C21CM_RR_NZ * C21CM_RR_NGAMMA, s->rr_c, tb,
                                stream);
        if (status) return status;
    }
    if (c->fused && c21hip_pair_sweep_supported(c->nx)) {
        /* pass X reads each spectrum tile once for two consecutive radii (C21CM_PAIR_RADII=0:
         * one radius per sweep) */
        const char *e = getenv("C21CM_PAIR_RADII");
        if (!(e && e[0] == '0')) {
            c->delta_work2 = (float *)c21hip_ws(WS_DELTA_WORK2, gbytes);
            c->stars_work2 = c->delta_work2 ? c21_place_work_partner(WS_DELTA_WORK2, WS_STARS_WORK2, gbytes, c->nx, c->ny,
                                                                 c->nz, stream)
                                            : NULL;
            if (!c->delta_work2 || !c->stars_work2) return C21CM_MEMORY_ALLOC_ERROR;
            if ((s->use_ts_fluct || (c->fused_rc && c->filter_rec)) &&
                !(c->xe_work2 = (float *)c21hip_ws(WS_XE_WORK2, gbytes)))
                return C21CM_MEMORY_ALLOC_ERROR; /* (the second work spectrum of N_rec on the fused loop too) */
            if (c->fused_rc && !(c->sfr_work2 = c->xe_work2 ? c21_place_work_partner(WS_XE_WORK2, WS_SFR_WORK2, gbytes, c->nx,
                                                                                  c->ny, c->nz, stream)
                                                             : (float *)c21hip_ws(WS_SFR_WORK2, gbytes)))
                return C21CM_MEMORY_ALLOC_ERROR;
            if (c->fused_rc && c->filter_rec && s->use_ts_fluct &&
                !(c->x4_work2 = (float *)c21hip_ws(WS_NREC_WORK2, gbytes)))
                return C21CM_MEMORY_ALLOC_ERROR;
            c->pair_radii = 1;
        }
    }
    if (c->fused && s->use_ts_fluct) {
        c->x3_on = 1;
        c->x3_unf = c->xe_unf, c->x3_work = c->xe_work, c->x3_work2 = c->xe_work2;
        if (c->fused_rc && c->filter_rec) {
            c->x4_on = 1;
            c->x4_unf = c->nrec_unf, c->x4_work = c->nrec_work;
        }
    } else if (c->fused_rc && c->filter_rec) {
        c->x3_on = c->x3_nrec = 1;
        c->x3_unf = c->nrec_unf, c->x3_work = c->nrec_work, c->x3_work2 = c->xe_work2;
    }
    g_loop_flags = (c->fused ? 1 : 0) | (c->fused_rc ? 2 : 0) | (c->x3_on ? 4 : 0) | (c->x3_nrec ? 8 : 0) |
                   (c->x4_on ? 16 : 0) | (c->pair_radii ? 32 : 0);
    if (c->fused) {
        const char *e = getenv("C21CM_DEFER_SUMS");
        if (!(e && e[0] == '0')) {
            c->def_stride = c21hip_z_ionise_partials(c->nx, c->ny, c->nz);
            c->def_partials = (double *)c21hip_ws(
                WS_DEF_PARTIALS, (size_t)s->n_radii * (size_t)c->def_stride * sizeof(double));
            if (!c->def_partials) return C21CM_MEMORY_ALLOC_ERROR;
        }
    }
    c->scalars = (double *)c21hip_ws(WS_SCALARS, SC_COUNT * sizeof(double));
    /* two float tables (the pipelined table loop), or the node data of C21CM_FCOLL_NODES */
    c->table_dev = (float *)c21hip_ws(WS_TABLE, 2 * C21CM_NDELTA_TABLE * sizeof(float) + C21CM_NODE_DOUBLES * sizeof(double));
    if (!c->scalars || !c->table_dev) return C21CM_MEMORY_ALLOC_ERROR;
    status = c21hip_memset(c->scalars, 0, SC_COUNT * sizeof(double), stream);
    if (status) return status;

    c->density = stage_in(WS_DENSITY, pf->density, dbytes, stream, &status);
    if (c->lagrangian) c->n_ion = stage_in(WS_NION, halos->n_ion, dbytes, stream, &status);
    if (s->use_ts_fluct) {
        c->xe_dense = stage_in(WS_XE_DENSE, ts->xray_ionised_fraction, dbytes, stream, &status);
        if (!s->minimize_memory)
            c->Tneutral = stage_in(WS_TNEUTRAL, ts->kinetic_temp_neutral, dbytes, stream, &status);
    }
    if (!s->first_snapshot && prev && prev->z_reion)
        c->prev_zre = stage_in(WS_PREV_ZRE, prev->z_reion, dbytes, stream, &status);
    if (!s->first_snapshot && !c->prev_zre && !status) {
        c21hip_set_error("ionize: previous z_reion is required after the first snapshot");
        return C21CM_VALUE_ERROR;
    }
    if (need_outputs) {
        c->xH = stage_inout(WS_XH, box->neutral_fraction, dbytes, 1, &c->cb, stream, &status);
        c->zre = stage_inout(WS_ZRE, box->z_reion, dbytes, 0, &c->cb, stream, &status);
        if (!s->minimize_memory)
            c->Tk = stage_inout(WS_TK, box->kinetic_temperature, dbytes, 1, &c->cb, stream,
                                &status);
    }
    if (c->mini) {
        if (!prev || !prev->unnormalised_nion || !prev->unnormalised_nion_mini) {
            c21hip_set_error("ionize: USE_MINI_HALOS needs the previous box's unnormalised_nion "
                             "and unnormalised_nion_mini histories");
            return C21CM_VALUE_ERROR;
        }
        const size_t hbytes = dbytes * (size_t)s->n_radii; /* one grid per radius (:783-784) */
        c->prev_density = stage_in(WS_MINI_PDENS, s->prev_density, dbytes, stream, &status);
        c->mta_dense = stage_in(WS_MINI_MTA, s->log10_mturn_acg, dbytes, stream, &status);
        c->mtm_dense = stage_in(WS_MINI_MTM, s->log10_mturn_mcg, dbytes, stream, &status);
        c->hist_a = stage_in(WS_MINI_HIST_A, prev->unnormalised_nion, hbytes, stream, &status);
        c->hist_m = stage_in(WS_MINI_HIST_M, prev->unnormalised_nion_mini, hbytes, stream, &status);
        c->nion_all = stage_inout(WS_MINI_OUT_A, box->unnormalised_nion, hbytes, 0, &c->cb, stream,
                                  &status);
        c->mini_all = stage_inout(WS_MINI_OUT_M, box->unnormalised_nion_mini, hbytes, 0, &c->cb,
                                  stream, &status);
    } else if (!c->lagrangian)
        c->nion_dense = stage_inout(WS_NION_DENSE, box->unnormalised_nion, dbytes, 0, &c->cb,
                                    stream, &status);
    if (c->recomb) {
        if (!prev || !prev->cumulative_recombinations) {
            c21hip_set_error("ionize: a recombination model needs the previous box's "
                             "cumulative_recombinations");
            return C21CM_VALUE_ERROR;
        }
        const size_t rbytes = c->inhomo ? dbytes : sizeof(float); /* homogeneous: one number */
        c->prev_nrec = stage_in(WS_PREV_NREC, prev->cumulative_recombinations, rbytes, stream,
                                &status);
        if (c->lagrangian)
            c->whalo_sfr = stage_in(WS_WSFR, halos->whalo_sfr, dbytes, stream, &status);
        if (need_outputs) {
            /* Gamma_12 / mean free path keep the caller's values where no barrier is crossed */
            c->G12 = stage_inout(WS_G12, box->ionisation_rate_G12, dbytes, 1, &c->cb, stream,
                                 &status);
            if (!s->minimize_memory && box->mean_free_path)
                c->mfp = stage_inout(WS_MFP, box->mean_free_path, dbytes, 1, &c->cb, stream,
                                     &status);
            if (c->inhomo)
                c->nrec_out = stage_inout(WS_NREC_OUT, box->cumulative_recombinations, dbytes, 0,
                                          &c->cb, stream, &status);
        }
    }
    return status;
}

/* Identity of the inputs whose unfiltered spectra currently sit in the workspace.  Lets
 * c21cm_ionize_shard_finish reuse the spectra left by c21cm_ionize_shard_radii in the same
 * process instead of repeating the pre-loop transforms. */
static struct {
    int valid, nx, nz, ts;
    int stars_r0_ready; /* stars_fil already holds the unfiltered emissivity in real space */
    const void *density, *n_ion, *xe;
    double factor;
} g_spectra;

static int spectra_match(const ion_ctx *c, const PerturbedField *pf, const HaloBox *halos,
                         const TsBox *ts) {
    return g_spectra.valid && g_spectra.nx == c->nx && g_spectra.nz == c->nz &&
           g_spectra.ts == c->s->use_ts_fluct && g_spectra.density == (const void *)pf->density &&
           g_spectra.n_ion == (const void *)(c->lagrangian ? halos->n_ion : NULL) &&
           g_spectra.xe == (const void *)(c->s->use_ts_fluct ? ts->xray_ionised_fraction : NULL) &&
           g_spectra.factor == c->s->photoncons_adjustment_factor;
}

static void spectra_remember(const ion_ctx *c, const PerturbedField *pf, const HaloBox *halos,
                             const TsBox *ts) {
    g_spectra.valid = 1;
    g_spectra.stars_r0_ready = 0;
    g_spectra.nx = c->nx;
    g_spectra.nz = c->nz;
    g_spectra.ts = c->s->use_ts_fluct;
    g_spectra.density = pf->density;
    g_spectra.n_ion = c->lagrangian ? halos->n_ion : NULL;
    g_spectra.xe = c->s->use_ts_fluct ? ts->xray_ionised_fraction : NULL;
    g_spectra.factor = c->s->photoncons_adjustment_factor;
}

/* prepare_box_for_filtering: IonisationBox.c:323-360.  `scratch` is the grid's *_fil buffer:
 * with the native FFT the padded spectrum is built there and re-laid-out into cgrid. */
static int prepare_grid(ion_ctx *c, const float *dense, float *cgrid, float *scratch,
                        double factor, double lo, double hi) {
    int status = 0;
    (void)scratch;
    if (c->native) {
        /* clip+scale on load, r2c in three sweeps, 1/N (exact: N is a power of two) on store */
        TRY(c21hip_split_r2c(dense, c->nz, cgrid, c->nx, c->ny, c->nz, factor, lo, hi,
                             1.0f / (float)c->ntot, c->stream));
        goto done;
    }
    float *padded = cgrid;
    TRY(c21hip_pack_clip(dense, padded, c->nx, c->ny, c->nz, factor, lo, hi, c->stream));
    TRY(c21hip_fft_r2c(padded, c->nx, c->ny, c->nz, c->stream));
    TRY(c21hip_divide_inplace(padded, c->npad, (float)c->ntot, c->stream));
done:
    return status;
}

/* one grid of copy_filter_transform (IonisationBox.c:577-663): unfiltered spectrum ->
 * filtered real-space grid with padded rows */
static int filter_to_real(ion_ctx *c, const float *unf, float *work, float *fil, int filter_type,
                          float R, float R_param, int apply) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    if (c->native) {
        TRY(c21hip_split_filter_c2r(unf, work, fil, 2 * (long)(c->nz / 2 + 1), c->nx, c->ny, c->nz,
                                    s->box_len, s->box_len_z, filter_type, R, R_param, apply,
                                    c->stream));
    } else {
        TRY(c21hip_copy_filter(unf, fil, c->nx, c->ny, c->nz, s->box_len, s->box_len_z,
                               filter_type, R, R_param, apply, c->stream));
        TRY(c21hip_fft_c2r(fil, c->nx, c->ny, c->nz, c->stream));
    }
done:
    return status;
}

static int preloop(ion_ctx *c) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    /* IonisationBox.c:1480-1513 */
    TRY(prepare_grid(c, c->density, c->delta_unf, c->delta_fil, s->photoncons_adjustment_factor,
                     -1., 1e6));
    if (c->lagrangian) TRY(prepare_grid(c, c->n_ion, c->stars_unf, c->stars_fil, 1., 0., 1e20));
    if (s->use_ts_fluct) TRY(prepare_grid(c, c->xe_dense, c->xe_unf, c->xe_fil, 1., 0., 1.));
    if (c->recomb && c->lagrangian)
        TRY(prepare_grid(c, c->whalo_sfr, c->sfr_unf, c->sfr_fil, 1., 0., 1e20));
    if (c->filter_rec) TRY(prepare_grid(c, c->prev_nrec, c->nrec_unf, c->nrec_fil, 1., 0., 1e20));
    if (c->mini) { /* :1493-1509: the turnover grids are transformed unclipped */
        TRY(prepare_grid(c, c->prev_density, c->pd_unf, c->pd_fil, 1., -1., 1e6));
        TRY(prepare_grid(c, c->mtm_dense, c->mtm_unf, c->mtm_fil, 1., -INFINITY, INFINITY));
        TRY(prepare_grid(c, c->mta_dense, c->mta_unf, c->mta_fil, 1., -INFINITY, INFINITY));
    }
done:
    return status;
}

/* The W(kR) tables of a radius depend on no grid data, so those of the NEXT radius are built on
 * a side stream while the current radius runs its passes (the table kernel is fp64-ALU work,
 * the passes are memory/LDS work).  Two table buffers alternate; ev_table[b] = buffer b is
 * filled, ev_used[b] = the pass X that read buffer b has been issued on the caller's stream.
 * C21CM_ASYNC_TABLES=0 builds them inline on the caller's stream instead. */
static struct {
    int init, enabled;
    void *aux, *ev_table[4], *ev_used[4], *ev_sync;
} g_tab;

static void tab_init(void) {
    if (g_tab.init) return;
    g_tab.init = 1;
    const char *e = getenv("C21CM_ASYNC_TABLES");
    if (e && e[0] == '0') return;
    g_tab.aux = c21hip_aux_stream();
    g_tab.ev_sync = c21hip_event_create();
    g_tab.enabled = g_tab.aux && g_tab.ev_sync;
    for (int b = 0; b < 4; b++) {
        g_tab.ev_table[b] = c21hip_event_create();
        g_tab.ev_used[b] = c21hip_event_create();
        g_tab.enabled = g_tab.enabled && g_tab.ev_table[b] && g_tab.ev_used[b];
    }
}

static int tab_build_async(ion_ctx *c, int R_ct, int buf) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    TRY(c21hip_stream_wait_event(g_tab.aux, g_tab.ev_used[buf]));
#ifdef C21CM_DIAG_BUILD /* (make EXTRA=-DC21CM_DIAG_BUILD: timing diagnostics that change results) */
    /* C21CM_DIAG_SKIP_TABLES=1 (WRONG results): no table kernels after the first step -- what the R
     * loop would cost if the windows came for free */
    static int skip = -1;
    if (skip < 0) skip = getenv("C21CM_DIAG_SKIP_TABLES") != NULL;
    if (!(skip && c->tab_seq > 0))
#endif
    TRY(c21hip_window_tables(buf, s->hii_filter, 0.f, s->stars_filter, (float)s->mfp_meandens,
                             c->nx, c->ny, c->nz, s->box_len, s->box_len_z, (float)s->R[R_ct],
                             g_tab.aux));
    TRY(c21hip_event_record(g_tab.ev_table[buf], g_tab.aux));
done:
    return status;
}

/* The barrier the closed-form Eulerian loop still owes (its last radius had no successor to ride on). */
static int eul_flush_pending(ion_ctx *c, int last) {
    if (c->eul_pend < 0) return 0;
    c21hip_ionize_args args;
    fill_args(&args, c->s, c->eul_pend);
    const int R = c->eul_pend;
    c->eul_pend = -1;
    int st = c21hip_eulerian_mask(&args, c->eul_pend_buf ? c->nion_dense2 : c->nion_dense, NULL,
                                  c->scalars + SC_MEANS + R, c->eul_pend_mask, c->stream);
    /* `last`: no further radius writes the f_coll grid, and box->unnormalised_nion is that of the
     * last radius processed (IonisationBox.c:773-962 overwrite it per radius) */
    if (!st && last && c->eul_pend_buf)
        st = c21hip_d2d(c->nion_dense, c->nion_dense2, c->ntot * sizeof(float), c->stream);
    return st;
}

/* The f_coll sums and means of the fused radii processed so far, in one launch. */
static int flush_deferred(ion_ctx *c) {
    {
        const int st_e = eul_flush_pending(c, 1);
        if (st_e) return st_e;
    }
    if (!c->def_partials || c->def_count == 0) return 0;
    const c21cm_ionize_spec *s = c->s;
    int st = c21hip_batched_means(c->def_partials, c->def_stride, (int)c->def_stride, c->def_first,
                                  c->def_count > 1 ? c->def_step : 1, c->def_count,
                                  (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                                  c->scalars + SC_SUMS, c->scalars + SC_MEANS, c->stream);
    c->def_count = 0;
    return st;
}

/* Fused pass Z of one radius (density + emissivity [+ x_e] spectra after passes X, Y): barrier
 * test into the mask, f_coll sum deferred or reduced now. */
static int z_ionise_radius(ion_ctx *c, int R_ct, const float *dwork, const float *swork,
                           const float *xwork, unsigned char *first_cross) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    if (c->fused_rc) { /* xwork: the filtered whalo_sfr; sums always deferred or reduced below */
        double *part = c->def_partials ? c->def_partials + (long)R_ct * c->def_stride : c->partials;
        if (c->def_partials) {
            if (c->def_count == 0)
                c->def_first = R_ct;
            else if (c->def_count == 1)
                c->def_step = c->def_first - R_ct;
            else if (R_ct != c->def_first - c->def_count * c->def_step)
                TRY(flush_deferred(c));
            if (c->def_count == 0) c->def_first = R_ct;
            c->def_count++;
        }
        if (c->x4_on) /* CELL_RECOMB = false with an x_e grid: four spectra */
            TRY(c21hip_split_z_ionise_recomb_xe_nrec(dwork, swork, c->cur_xe, c->cur_x4, c->G12, first_cross,
                                                     part, c->nx, c->ny, c->nz, R_ct, s->rhocrit_omb,
                                                     s->ion_eff_factor, s->mass_dep_zeta, s->f_limit_acg,
                                                     c->stream));
        else if (c->x3_nrec) /* CELL_RECOMB = false: N_rec filtered at the radius is the third line */
            TRY(c21hip_split_z_ionise_recomb_nrec(dwork, swork, c->cur_xe, c->G12, first_cross, part, c->nx,
                                                  c->ny, c->nz, R_ct, s->rhocrit_omb, s->ion_eff_factor,
                                                  s->mass_dep_zeta, s->f_limit_acg, c->stream));
        else
        TRY(c21hip_split_z_ionise_recomb_xe(dwork, swork, s->use_ts_fluct ? c->cur_xe : NULL,
                                            c->inhomo ? c->prev_nrec : NULL, c->rec0, c->G12, first_cross,
                                            part, c->nx, c->ny, c->nz, R_ct, s->rhocrit_omb,
                                            s->ion_eff_factor, s->mass_dep_zeta, s->f_limit_acg, c->stream));
        TRY(c21hip_split_z_sfr_gamma12(xwork, first_cross, c->G12, c->nx, c->ny, c->nz, R_ct,
                                       s->R[R_ct] * s->gamma_prefactor, c->stream));
        if (!c->def_partials) {
            double *sum_dev = c->scalars + SC_SUMS + R_ct;
            TRY(c21hip_reduce_sum(part, c21hip_z_ionise_partials(c->nx, c->ny, c->nz), sum_dev, c->stream));
            TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                                   c->scalars + SC_MEANS + R_ct, c->stream));
        }
        goto done;
    }
    if (c->def_partials) {
        /* no kernel of this loop reads a radius' mean: reduce all of them at the end */
        if (c->def_count == 0)
            c->def_first = R_ct;
        else if (c->def_count == 1)
            c->def_step = c->def_first - R_ct;
        else if (R_ct != c->def_first - c->def_count * c->def_step)
            TRY(flush_deferred(c));
        if (c->def_count == 0) c->def_first = R_ct;
        c->def_count++;
        TRY(c21hip_split_z_ionise_stars_xe(dwork, swork, xwork, first_cross,
                                           c->def_partials + (long)R_ct * c->def_stride, NULL,
                                           c->nx, c->ny, c->nz, R_ct, s->rhocrit_omb,
                                           s->ion_eff_factor, s->mass_dep_zeta, s->f_limit_acg,
                                           c->stream));
        goto done;
    }
    {
        double *sum_dev = c->scalars + SC_SUMS + R_ct;
        TRY(c21hip_split_z_ionise_stars_xe(dwork, swork, xwork, first_cross, c->partials, sum_dev,
                                           c->nx, c->ny, c->nz, R_ct, s->rhocrit_omb,
                                           s->ion_eff_factor, s->mass_dep_zeta, s->f_limit_acg,
                                           c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               c->scalars + SC_MEANS + R_ct, c->stream));
    }
done:
    return status;
}

/* One step of the fused R loop: radius R_a, and with R_b >= 1 also radius R_b out of the same
 * pass-X sweep (each spectrum tile read once, windowed and transformed twice).  next_a / next_b:
 * the radii of the step after this one (-1: none), whose window tables are built ahead on the
 * side stream.  Four table buffers: steps alternate between the pairs (0, 1) and (2, 3). */
static int fused_step(ion_ctx *c, int R_a, int R_b, unsigned char *first_cross, int next_a,
                      int next_b) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    const int set = c->tab_seq & 1;
    const int buf_a = 2 * set, buf_b = 2 * set + 1;
    tab_init();
    /* below ~64 M cells the two cross-stream waits per radius cost more than the table
     * kernel they hide (256^3: 9.5 vs 9.25 ms per call): build the tables inline there */
    const int tab_async = !c->wev && g_tab.enabled && c->ntot >= ((size_t)1 << 26);
    if (tab_async) {
        if (c->tab_seq == 0) {
            /* first fused step of this call: order the side stream after whatever the
             * caller's stream still runs on the table buffers, then build this step's tables */
            TRY(c21hip_event_record(g_tab.ev_sync, c->stream));
            TRY(c21hip_stream_wait_event(g_tab.aux, g_tab.ev_sync));
            TRY(tab_build_async(c, R_a, buf_a));
            if (R_b >= 1) TRY(tab_build_async(c, R_b, buf_b));
        }
        if (next_a >= 1) TRY(tab_build_async(c, next_a, 2 * (set ^ 1)));
        if (next_b >= 1) TRY(tab_build_async(c, next_b, 2 * (set ^ 1) + 1));
        TRY(c21hip_stream_wait_event(c->stream, g_tab.ev_table[buf_a]));
        if (R_b >= 1) TRY(c21hip_stream_wait_event(c->stream, g_tab.ev_table[buf_b]));
    }
    if (R_b >= 1) {
        /* X (both radii) -> Y_a -> Z_a -> Y_b -> Z_b: each fused pass Z directly follows the pass
         * Y that wrote its input and walks the lines backwards, so its first quarter comes out
         * of the Infinity Cache (C21CM_PAIR_ORDER=yy: both passes Y first) */
        static int yy = -1;
        if (yy < 0) {
            const char *e = getenv("C21CM_PAIR_ORDER");
            yy = (e && e[0] == 'y') ? 1 : 0;
        }
        /* third spectrum handed to z_ionise_radius: whalo_sfr on the fused recombination loop (the
         * x_e spectrum of such a run travels in c->cur_xe), else x_e */
        const float *xw[2] = {c->fused_rc ? c->sfr_work : (s->use_ts_fluct ? c->xe_work : NULL),
                              c->fused_rc ? c->sfr_work2 : (s->use_ts_fluct ? c->xe_work2 : NULL)};
        const float *xe_of[2] = {c->x3_on ? c->x3_work : NULL, c->x3_on ? c->x3_work2 : NULL};
        for (int ph = 0; ph < 3; ph++) { /* pass X, pass Y of R_a, pass Y of R_b */
            const int bits = ph == 0 ? (tab_async ? 2 : 3) : (4 << (ph - 1));
            TRY(c21hip_split_filter_xy2_pair(
                c->delta_unf, c->delta_work, c->delta_work2, s->hii_filter, 0.f, c->stars_unf,
                c->stars_work, c->stars_work2, s->stars_filter, (float)s->mfp_meandens, c->nx,
                c->ny, c->nz, s->box_len, s->box_len_z, (float)s->R[R_a], (float)s->R[R_b], buf_a,
                buf_b, bits, c->stream));
            /* x_e (N_rec) shares the density grid's window (IonisationBox.c:1551-1553, :613), whalo_sfr
             * the emissivity's (IonisationBox.c:583-663): together they are a second two-grid sweep
             * with the same window pair (one pass X, one pass Y per radius instead of two each) */
            static int merge34 = -1;
            if (merge34 < 0) {
                const char *e = getenv("C21CM_RECOMB_MERGE34");
                merge34 = (e && e[0] == '0') ? 0 : 1;
            }
            if (c->x3_on && c->fused_rc && c->wev && merge34) {
                TRY(c21hip_split_filter_xy2_pair(
                    c->x3_unf, c->x3_work, c->x3_work2, s->hii_filter, 0.f, c->sfr_unf, c->sfr_work,
                    c->sfr_work2, s->stars_filter, (float)s->mfp_meandens, c->nx, c->ny, c->nz,
                    s->box_len, s->box_len_z, (float)s->R[R_a], (float)s->R[R_b], buf_a, buf_b,
                    bits & ~1, c->stream));
            } else {
            if (c->x3_on)
                TRY(c21hip_split_filter_xy_shared_pair(
                    c->x3_unf, c->x3_work, c->x3_work2, s->hii_filter, c->nx, c->ny, c->nz,
                    s->box_len, s->box_len_z, (float)s->R[R_a], (float)s->R[R_b], buf_a, buf_b,
                    bits & ~1, c->stream));
            if (c->fused_rc)
                TRY(c21hip_split_filter_xy_single_pair(
                    c->sfr_unf, c->sfr_work, c->sfr_work2, s->stars_filter, (float)s->mfp_meandens,
                    c->nx, c->ny, c->nz, s->box_len, s->box_len_z, (float)s->R[R_a],
                    (float)s->R[R_b], bits & ~1, c->stream));
            }
            if (c->x4_on) /* N_rec under the density grid's window too (IonisationBox.c:613) */
                TRY(c21hip_split_filter_xy_shared_pair(
                    c->x4_unf, c->x4_work, c->x4_work2, s->hii_filter, c->nx, c->ny, c->nz,
                    s->box_len, s->box_len_z, (float)s->R[R_a], (float)s->R[R_b], buf_a, buf_b,
                    bits & ~1, c->stream));
            /* (the tables are free after pass X, their only reader; releasing them there lets the
             * next builds run under pass Y, which measured 4 ms per call slower than under pass Z) */
            if (ph == (yy ? 2 : 1) && tab_async) {
                TRY(c21hip_event_record(g_tab.ev_used[buf_a], c->stream));
                TRY(c21hip_event_record(g_tab.ev_used[buf_b], c->stream));
            }
            if (ph == 1 && !yy) {
                c->cur_xe = xe_of[0];
                c->cur_x4 = c->x4_work;
                TRY(z_ionise_radius(c, R_a, c->delta_work, c->stars_work, xw[0], first_cross));
            }
        }
        c->tab_seq++;
        if (yy) {
            c->cur_xe = xe_of[0];
            c->cur_x4 = c->x4_work;
            TRY(z_ionise_radius(c, R_a, c->delta_work, c->stars_work, xw[0], first_cross));
        }
        c->cur_xe = xe_of[1];
        c->cur_x4 = c->x4_work2;
        TRY(z_ionise_radius(c, R_b, c->delta_work2, c->stars_work2, xw[1], first_cross));
        goto done;
    } else {
        TRY(c21hip_split_filter_xy2(c->delta_unf, c->delta_work, s->hii_filter, 0.f, c->stars_unf,
                                    c->stars_work, s->stars_filter, (float)s->mfp_meandens, c->nx,
                                    c->ny, c->nz, s->box_len, s->box_len_z, (float)s->R[R_a], 1,
                                    buf_a, tab_async, c->stream));
        if (c->x3_on) /* x_e (N_rec) shares the density grid's window (IonisationBox.c:1551-1553, :613) */
            TRY(c21hip_split_filter_xy_shared(c->x3_unf, c->x3_work, s->hii_filter, c->nx, c->ny,
                                              c->nz, s->box_len, s->box_len_z, (float)s->R[R_a], 1,
                                              buf_a, c->stream));
        if (c->fused_rc)
            TRY(c21hip_split_filter_xy(c->sfr_unf, c->sfr_work, c->nx, c->ny, c->nz, s->box_len,
                                       s->box_len_z, s->stars_filter, (float)s->R[R_a],
                                       (float)s->mfp_meandens, 1, c->stream));
        if (c->x4_on)
            TRY(c21hip_split_filter_xy_shared(c->x4_unf, c->x4_work, s->hii_filter, c->nx, c->ny,
                                              c->nz, s->box_len, s->box_len_z, (float)s->R[R_a], 1,
                                              buf_a, c->stream));
        if (tab_async) TRY(c21hip_event_record(g_tab.ev_used[buf_a], c->stream));
    }
    c->tab_seq++;
    c->cur_xe = c->x3_on ? c->x3_work : NULL;
    c->cur_x4 = c->x4_work;
    TRY(z_ionise_radius(c, R_a, c->delta_work, c->stars_work,
                        c->fused_rc ? c->sfr_work : (s->use_ts_fluct ? c->xe_work : NULL), first_cross));
done:
    return status;
}

/* The fused R loop over radii first, first - step, ... >= lowest (>= 1), two per sweep where
 * the context allows it. */
static int fused_loop(ion_ctx *c, int first, int step, int lowest, unsigned char *first_cross) {
    int status = 0;
    if (lowest < 1) lowest = 1;
    if (first < lowest) return 0; /* a rank beyond the number of radii has nothing to do */
    /* Windows evaluated inside pass X from node tables of W(kR) (fft_native.hip: c21hip_wev_prepare)
     * where the filter types and the line length allow it: then no 3-D window table is built,
     * written or streamed for these radii (C21CM_WINDOWS=table keeps the tables). */
    c->wev = 0;
    {
        float radii[C21CM_MAX_RADII];
        int n = 0;
        for (int R = first; R >= lowest && n < C21CM_MAX_RADII; R -= step) radii[n++] = (float)c->s->R[R];
        if (n > 0)
            TRY(c21hip_wev_prepare(c->s->hii_filter, 0.f, c->s->stars_filter, (float)c->s->mfp_meandens,
                                   2, radii, n, c->nx, c->ny, c->nz, c->s->box_len, c->s->box_len_z,
                                   c->pair_radii, &c->wev, c->stream));
    }
    if (c->fused_rc) {
        if (!c->wev) {
            c21hip_set_error("ionize: the fused recombination loop needs the evaluated windows");
            status = C21CM_VALUE_ERROR;
            goto done;
        }
        c->rec0 = 0.;
        if (!c->inhomo) { /* the homogeneous model's one number (outputs.py:1526-1537) */
            float r0 = 0.f;
            if (c21hip_is_device_ptr(c->prev_nrec)) {
                TRY(c21hip_d2h(&r0, c->prev_nrec, sizeof(float), c->stream));
                TRY(c21hip_sync(c->stream));
            } else {
                r0 = c->prev_nrec[0];
            }
            c->rec0 = (double)r0;
        }
    }
    int R_a = first;
    while (R_a >= lowest) {
        const int want_b = c->pair_radii && R_a - step >= lowest;
        const int R_b = want_b ? R_a - step : -1;
        const int n_a = (want_b ? R_b : R_a) - step;
        const int next_a = n_a >= lowest ? n_a : -1;
        const int next_b = (c->pair_radii && next_a >= 1 && next_a - step >= lowest) ? next_a - step
                                                                                     : -1;
        TRY(fused_step(c, R_a, R_b, first_cross, next_a, next_b));
        R_a = n_a;
    }
done:
    c21hip_wev_release();
    c->wev = 0;
    return status;
}

/* One filter radius: IonisationBox.c:1546-1580.  first_cross != NULL = shard mode.
 * next_R: the radius index this process handles after R_ct (-1: none / unknown). */
/* the two dense x_e(R) buffers of the Eulerian mask path */
static int eul_xe_buffers(ion_ctx *c) {
    if (c->eul_xe[0] && c->eul_xe[1]) return 0;
    c->eul_xe[0] = (float *)c21hip_ws(WS_EUL_XE0, c->ntot * sizeof(float));
    c->eul_xe[1] = (float *)c21hip_ws(WS_EUL_XE1, c->ntot * sizeof(float));
    return (c->eul_xe[0] && c->eul_xe[1]) ? 0 : C21CM_MEMORY_ALLOC_ERROR;
}

/* One radius with USE_MINI_HALOS (delta, x_e and N_rec are already filtered): the previous
 * delta and the two turnover grids, extrema of all four, the 2-D tables of this (and the
 * previous) redshift from the host callback, both f_coll grids with their history, the
 * two-population barrier.  reference: IonisationBox.c:595-603,715-761,838-936,1068-1200 */
static int mini_radius(ion_ctx *c, int R_ct, const c21hip_ionize_args *args, int apply, float R) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    const size_t t2 = (size_t)C21CM_NDELTA_TABLE * C21CM_NMTURN_TABLE;
    const size_t roff = (size_t)R_ct * c->ntot;
    double mm[8];
    float *tables = (float *)malloc(sizeof(float) * 4 * t2);
    if (!tables) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(filter_to_real(c, c->pd_unf, c->pd_work, c->pd_fil, s->hii_filter, R, 0.f, apply));
    TRY(filter_to_real(c, c->mtm_unf, c->mtm_work, c->mtm_fil, s->hii_filter, R, 0.f, apply));
    TRY(filter_to_real(c, c->mta_unf, c->mta_work, c->mta_fil, s->hii_filter, R, 0.f, apply));
    TRY(c21hip_clip_minmax(c->delta_fil, c->nx, c->ny, c->nz, c->partials, c->scalars + SC_MINMAX,
                           c->stream));
    TRY(c21hip_clip_minmax(c->pd_fil, c->nx, c->ny, c->nz, c->partials, c->scalars + SC_MINMAX_M,
                           c->stream));
    TRY(c21hip_clip_minmax(c->mta_fil, c->nx, c->ny, c->nz, c->partials,
                           c->scalars + SC_MINMAX_M + 2, c->stream));
    TRY(c21hip_clip_minmax(c->mtm_fil, c->nx, c->ny, c->nz, c->partials,
                           c->scalars + SC_MINMAX_M + 4, c->stream));
    TRY(c21hip_d2h(mm, c->scalars + SC_MINMAX, 2 * sizeof(double), c->stream));
    TRY(c21hip_d2h(mm + 2, c->scalars + SC_MINMAX_M, 6 * sizeof(double), c->stream));
    TRY(c21hip_sync(c->stream));
    {
        /* setup_integration_tables: margins of :712-713,735-741 */
        const double dmin = mm[0] - 0.001, dmax = mm[1] + 0.001;
        const double pmin = mm[2] - 0.001, pmax = mm[3] + 0.001;
        const double amin = mm[4] * 0.99, amax = mm[5] * 1.01;
        const double mmin = mm[6] * 0.99, mmax = mm[7] * 1.01;
        int tst = s->table2d_fn(R_ct, 0, dmin, dmax, amin, amax, mmin, mmax, tables, tables + t2,
                                s->table2d_user);
        if (!tst && s->need_prev_ion)
            tst = s->table2d_fn(R_ct, 1, pmin, pmax, amin, amax, mmin, mmax, tables + 2 * t2,
                                tables + 3 * t2, s->table2d_user);
        if (tst) {
            c21hip_set_error("ionize: table2d_fn failed with status %d at radius %d", tst, R_ct);
            status = tst;
            goto done;
        }
        const double ranges[8] = {dmin, (dmax - dmin) / (C21CM_NDELTA_TABLE - 1.),
                                  pmin, (pmax - pmin) / (C21CM_NDELTA_TABLE - 1.),
                                  amin, (amax - amin) / (C21CM_NMTURN_TABLE - 1.),
                                  mmin, (mmax - mmin) / (C21CM_NMTURN_TABLE - 1.)};
        TRY(c21hip_h2d(c->mini_tables, tables, sizeof(float) * (s->need_prev_ion ? 4 : 2) * t2,
                       c->stream));
        TRY(c21hip_sync(c->stream)); /* `tables` is freed below */
        TRY(c21hip_fcoll_mini(c->nx, c->ny, c->nz, s->need_prev_ion, ranges, c->delta_fil,
                              c->pd_fil, c->mta_fil, c->mtm_fil, c->mini_tables, c->hist_a + roff,
                              c->hist_m + roff, c->nion_all + roff, c->mini_all + roff, c->partials,
                              c->scalars + SC_PAIR, c->stream));
    }
    TRY(c21hip_d2d(c->scalars + SC_SUMS + R_ct, c->scalars + SC_PAIR, sizeof(double), c->stream));
    TRY(c21hip_d2d(c->scalars + SC_SUMS_M + R_ct, c->scalars + SC_PAIR + 1, sizeof(double),
                   c->stream));
    TRY(c21hip_finish_mean(c->scalars + SC_SUMS + R_ct, (double)c->ntot, s->mass_dep_zeta,
                           s->f_limit_acg, c->scalars + SC_MEANS + R_ct, c->stream));
    TRY(c21hip_finish_mean(c->scalars + SC_SUMS_M + R_ct, (double)c->ntot, s->mass_dep_zeta,
                           s->f_limit_mcg, c->scalars + SC_MEANS_M + R_ct, c->stream));
    TRY(c21hip_ionise_mini(args, 0, c->recomb, c->inhomo, s->cell_recomb, s->R[R_ct],
                           s->gamma_prefactor, s->gamma_prefactor_mini, s->ion_eff_factor_mini,
                           s->f_limit_mcg, s->mean_f_coll_mini, c->delta_fil,
                           c->nion_all + roff, c->mini_all + roff, c->xe_fil, c->nrec_fil,
                           c->prev_nrec, c->density, c->prev_zre, c->Tneutral,
                           c->scalars + SC_MEANS + R_ct, c->scalars + SC_MEANS_M + R_ct, c->xH,
                           c->zre, c->Tk, c->G12, c->mfp, c->partials, NULL, c->stream));
done:
    free(tables);
    return status;
}

/* C21CM_FCOLL_NODES: node data of radius R_ct from the host callback -> device -> per-cell sums of
 * delta_fil into nion_dense (+ the f_coll sum) */
static int fcoll_nodes(ion_ctx *c, int R_ct, double *partials, double *sum_dev) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    double nodes[C21CM_NODE_DOUBLES];
    const int tst = s->table_fn(R_ct, 0., 0., (float *)nodes, s->table_user);
    if (tst) {
        c21hip_set_error("ionize: table_fn (node data) failed with status %d at radius %d", tst, R_ct);
        return tst;
    }
    double *nodes_dev = (double *)(c->table_dev + 2 * C21CM_NDELTA_TABLE);
    TRY(c21hip_h2d(nodes_dev, nodes, sizeof(nodes), c->stream));
    TRY(c21hip_sync(c->stream)); /* `nodes` is a stack buffer */
    TRY(c21hip_fcoll_eulerian(c->delta_fil, c->nion_dense, c->nx, c->ny, c->nz, C21CM_FCOLL_NODES,
                              s->growth_factor, 0., 0., s->delta_c, 0., 1., (const float *)nodes_dev,
                              partials, sum_dev, c->stream));
done:
    return status;
}

/* C21CM_EUL_DEFER=1: the barrier of a radius rides the next radius' pass Z (EPI 6) instead of its own
 * sweep (eulerian_mask_kernel).  OFF by default: bit-identical, but measured slower -- 44.3 against
 * 42.6 ms per 512^3 x 40-radii call; the extra 6 N bytes and 48 registers cost the pass Z (336 us,
 * not purely instruction-bound after all) more than the 153 us sweep they replace. */
static int eul_defer_ok(ion_ctx *c) {
    const char *e = getenv("C21CM_EUL_DEFER");
    if (!(e && e[0] == '1')) return 0;
    if (!c21hip_z_fcoll_erfc_mask_supported(c->nx, c->ny, c->nz)) return 0;
    if (!c->nion_dense2)
        c->nion_dense2 = (float *)c21hip_ws(WS_NION_DENSE2, c->ntot * sizeof(float));
    return c->nion_dense2 != NULL;
}

/* Banded barrier of the closed-form Eulerian loop (default where the wave-level pass Z serves the
 * z-lines; C21CM_EUL_BAND=0: the dense f_coll grid + eulerian_mask_kernel of every radius).
 * The barrier of a radius needs the box mean of its f_coll grid (IonisationBox.c:1022-1027), which is
 * what forces a second sweep per radius.  But the test  f mean_f_coll / mean zeta > 1  is monotone in
 * the correction, and the mean is a smooth function of ln R: extrapolated from the two radii before it,
 * it is known to a fraction of a per cent BEFORE the sweep.  Pass Z (EPI 7) therefore decides every cell
 * on which both ends of the band agree, and leaves a marker + the cell's f_coll for the others (a few
 * per mille), which the next radius' sweep settles with the exact mean.  eul_band_kernel checks that the
 * exact correction fell inside the band; if it ever does not, the whole loop is rerun on the dense
 * sweeps -- the result is the same first-crossing grid bit for bit either way
 * (test_closed_form_loop_banded_barrier_equals_dense_sweeps).  The first two radii of a loop, radii
 * whose sigmas coincide and the last radius of a loop that stops above index 0 (its dense grid is
 * box->unnormalised_nion) take the dense sweeps. */
static int eul_band_ok(ion_ctx *c) {
    const char *e = getenv("C21CM_EUL_BAND");
    if ((e && e[0] == '0') || c->band_off || c->s->n_radii >= 255) return 0;
    return c21hip_z_fcoll_erfc_mask_supported(c->nx, c->ny, c->nz);
}

/* settle the markers a banded sweep left behind (no later sweep of the loop did) */
static int eul_band_flush(ion_ctx *c) {
    if (c->band_pend < 0) return 0;
    const int R = c->band_pend;
    c->band_pend = -1;
    if (c->band_mf)
        return c21hip_eul_resolve_pending_xe(R, c->nion_dense, c->band_xe_pend, c->scalars + SC_BANDX + R,
                                             c->s->mass_dep_zeta, c->s->f_limit_acg, c->s->ion_eff_factor,
                                             c->band_mask, c->ntot, c->stream);
    return c21hip_eul_resolve_pending(R, c->nion_dense, c->scalars + SC_BANDX + R, c->band_mask, c->ntot,
                                      c->stream);
}

/* End of a loop with banded sweeps: markers settled, then did every band hold?  *redo = the largest
 * radius index whose band missed (0: none): the first-crossing grid is rewound to its state before that
 * radius and the caller runs its radii <= *redo again (band_off is set: dense sweeps). */
static int eul_band_finish(ion_ctx *c, unsigned char *mask, int *redo) {
    *redo = 0;
    int st = eul_band_flush(c);
    if (st || !c->band_used) return st;
    double cell;
    int fail;
    if ((st = c21hip_d2h(&cell, c->scalars + SC_BANDFAIL, sizeof(cell), c->stream))) return st;
    if ((st = c21hip_sync(c->stream))) return st;
    memcpy(&fail, &cell, sizeof(int));
    if (getenv("C21CM_EUL_BAND_DEBUG")) {
        double b[2 * C21CM_MAX_RADII], t[C21CM_MAX_RADII], m[C21CM_MAX_RADII];
        if (!c21hip_d2h(b, c->scalars + SC_BAND, sizeof(b), c->stream) &&
            !c21hip_d2h(t, c->scalars + SC_BANDX, sizeof(t), c->stream) &&
            !c21hip_d2h(m, c->scalars + SC_MEANS, sizeof(m), c->stream) && !c21hip_sync(c->stream))
            for (int r = c->s->n_radii - 1; r >= 1; r--)
                fprintf(stderr, "band r=%3d mean=%.9e threshold=%.9e sure>=%.9e maybe>=%.9e rel=[%+.2e, %+.2e]\n",
                        r, m[r], t[r], b[2 * r], b[2 * r + 1], b[2 * r] / t[r] - 1., b[2 * r + 1] / t[r] - 1.);
        fprintf(stderr, "band fail=%d\n", fail);
    }
    c->band_used = 0;
    c->band_next = c->band_h1 = c->band_h2 = -1;
    c->band_hn = 0;
    if (fail > 0) {
        c->band_off = 1;
        *redo = fail;
        if ((st = c21hip_memset(c->scalars + SC_BANDFAIL, 0, sizeof(double), c->stream))) return st;
        return c21hip_eul_rewind(mask, fail, c->ntot, c->stream);
    }
    return 0;
}

/* Is radius R_ct of a descending loop decided on a band (the device holds one for it)?  */
static int eul_band_this(ion_ctx *c, int R_ct, unsigned char *first_cross, int sig_ok) {
    const c21cm_ionize_spec *s = c->s;
    /* (index 1 sits a step above the cell scale, where the mean leaves the curve the larger radii
     * drew -- 1.6 % at 512^3 against the < 0.1 % of every other step: dense) */
    return c->band_next == R_ct && R_ct != c->band_skip && sig_ok && R_ct >= 2 &&
           (R_ct > s->r_lowest || s->r_lowest == 0) && (c->band_pend < 0 || c->band_mask == first_cross);
}

/* After the sweep of radius R_ct (banded or dense) left n_part partial sums of its f_coll grid: their
 * sum, the mean, its exact threshold, the check of R_ct's band and the band of next_R -- one launch
 * (partials NULL: *sum_dev holds the sum already). */
static int eul_band_after(ion_ctx *c, int R_ct, int next_R, int banded, int sig_ok, const double *partials,
                          int n_part, double *sum_dev) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    const char *e_rel = getenv("C21CM_EUL_BAND_MINREL"), *e_shift = getenv("C21CM_EUL_BAND_SHIFT");
    double min_rel = e_rel ? atof(e_rel) : 0.005;
    if (!(min_rel >= 0.)) min_rel = 0.005;
    const double shift = e_shift ? atof(e_shift) : 0.; /* test hook: a prediction off by this fraction */
    const int h1 = c->band_h1, h2 = c->band_h2;
    c->band_hist[R_ct][0] = (short)h1, c->band_hist[R_ct][1] = (short)h2, c->band_hist[R_ct][2] = (short)c->band_hn;
    /* a band for the next radius only once a prediction has been checked against a radius -- from the
     * fourth radius of a loop on */
    const int will_next = next_R >= 1 && c->band_hn >= 2 && sig_ok;
    const int predict_next = next_R >= 1 && c->band_hn >= 1 && sig_ok; /* the prediction alone: its error is the next band's width */
    double t_cur = 0., t_next = 0.;
    if (h1 >= 0) {
        const double d1 = log(s->R[R_ct]) - log(s->R[h1]);
        if (next_R >= 1) t_next = (log(s->R[next_R]) - log(s->R[R_ct])) / d1;
        if (h2 >= 0) t_cur = d1 / (log(s->R[h1]) - log(s->R[h2]));
    }
    /* three equally spaced points in ln R (the ladder is geometric; a rank's share of it too): quadratic
     * extrapolation -- the E-INTEGRAL mean doubles over the ladder and bends (C21CM_EUL_BAND_QUAD=0: linear) */
    const char *e_q = getenv("C21CM_EUL_BAND_QUAD");
    const int quad = h1 >= 0 && h2 >= 0 && next_R >= 1 && fabs(t_next - 1.) < 1e-6 && fabs(t_cur - 1.) < 1e-6 &&
                     !(e_q && e_q[0] == '0');
    const char *e_sb = getenv("C21CM_EUL_SUMBAND"); /* 0: c21hip_reduce_sum's own launches (A/B) */
    const int split = partials && e_sb && e_sb[0] == '0';
    if (split) TRY(c21hip_reduce_sum(partials, n_part, sum_dev, c->stream));
    TRY(c21hip_eul_band(split ? NULL : partials, n_part, sum_dev, (double)c->ntot, s->mass_dep_zeta,
                        s->f_limit_acg, c->scalars + SC_MEANS, R_ct, h1, h2, t_cur, t_next,
                        predict_next ? next_R : -1, banded, s->fix_mean, s->mean_f_coll, s->ion_eff_factor,
                        min_rel, shift, c->scalars + SC_BAND, c->scalars + SC_BANDX,
                        (int *)(c->scalars + SC_BANDFAIL), (unsigned *)(c->scalars + SC_BANDCTR),
                        c->band_mf, quad, c->scalars + SC_BANDP, c->stream));
    c->band_next = will_next ? next_R : -1;
    if (sig_ok) {
        c->band_h2 = h1;
        c->band_h1 = R_ct;
        c->band_hn++;
    } else { /* a radius without sources: its clamped mean is no point of the curve */
        c->band_h1 = c->band_h2 = -1;
        c->band_hn = 0;
    }
done:
    return status;
}

/* The filtered density spectrum of radius R_ct into `dst` (passes X and Y; Eulerian loops, one grid).
 * Round 6: where the evaluated windows serve two radii per sweep (c->wev_pair), pass X also produces the spectrum
 * of next_R -- the radius this loop takes next -- into dst_next, where it waits for the call that asks for next_R
 * with that buffer and then only runs its pass Y: the unfiltered spectrum is read once for two radii (3 S
 * instead of 4 S per pair; 0.24 -> 0.18 ms per radius at 512^3).  The window of a radius is the same arithmetic
 * whoever its partner is (fft_native.hip: FMODE 6 / 7), so the pairing changes no bit.
 * Reference: IonisationBox.c:577-631 (copy + filter_box per radius), filtering.c:327-391. */
static int eul_filter_density(ion_ctx *c, int R_ct, int next_R, float *dst, float *dst_next) {
    const c21cm_ionize_spec *s = c->s;
    const float R = (float)s->R[R_ct];
    if (c->pairx_R == R_ct && c->pairx_buf == dst) { /* parked by the radius before: its pass Y is left */
        c->pairx_R = -1;
        return c21hip_split_filter_x_pair1(NULL, dst, NULL, s->hii_filter, c->nx, c->ny, c->nz, s->box_len,
                                           s->box_len_z, R, 0.f, 4, c->stream);
    }
    c->pairx_R = -1;
    if (c->wev_pair && R_ct > 0 && next_R > 0 && dst_next && dst_next != dst &&
        !c21hip_split_filter_x_pair1(c->delta_unf, dst, dst_next, s->hii_filter, c->nx, c->ny, c->nz, s->box_len,
                                     s->box_len_z, R, (float)s->R[next_R], 2 | 4, c->stream)) {
        c->pairx_R = next_R;
        c->pairx_buf = dst_next;
        return 0;
    }
    return c21hip_split_filter_xy(c->delta_unf, dst, c->nx, c->ny, c->nz, s->box_len, s->box_len_z, s->hii_filter,
                                  R, 0.f, R_ct > 0, c->stream);
}

/* The same for the two grids of a spin-temperature run's Eulerian loop (delta and x_e under one window): the
 * pair's second radius waits in (dst_next, xdst_next). */
static int eul_filter_density_xe(ion_ctx *c, int R_ct, int next_R, float *dst, float *dst_next, float *xdst,
                                 float *xdst_next) {
    const c21cm_ionize_spec *s = c->s;
    const float R = (float)s->R[R_ct];
    if (c->pairx_R == R_ct && c->pairx_buf == dst) {
        c->pairx_R = -1;
        return c21hip_split_filter_xy2_pair_eval(NULL, dst, NULL, s->hii_filter, NULL, xdst, NULL, s->hii_filter, c->nx,
                                                 c->ny, c->nz, s->box_len, s->box_len_z, R, 0.f, 4, c->stream);
    }
    c->pairx_R = -1;
    if (c->wev_pair && R_ct > 0 && next_R > 0 && dst_next && xdst_next && dst_next != dst && xdst_next != xdst &&
        !c21hip_split_filter_xy2_pair_eval(c->delta_unf, dst, dst_next, s->hii_filter, c->xe_unf, xdst, xdst_next,
                                           s->hii_filter, c->nx, c->ny, c->nz, s->box_len, s->box_len_z, R,
                                           (float)s->R[next_R], 2 | 4, c->stream)) {
        c->pairx_R = next_R;
        c->pairx_buf = dst_next;
        return 0;
    }
    return c21hip_split_filter_xy2(c->delta_unf, dst, s->hii_filter, 0.f, c->xe_unf, xdst, s->hii_filter, 0.f, c->nx,
                                   c->ny, c->nz, s->box_len, s->box_len_z, R, R_ct > 0, 0, 0, c->stream);
}

static int one_radius(ion_ctx *c, int R_ct, unsigned char *first_cross, int next_R) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    const int apply = R_ct > 0; /* copy_filter_transform skips filter_box at R_index 0 (:606) */
    const float R = (float)s->R[R_ct];
    double *partials = c->partials;
    double *sum_dev = c->scalars + SC_SUMS + R_ct;
    double *mean_dev = c->scalars + SC_MEANS + R_ct;
    c21hip_ionize_args args;
    fill_args(&args, s, R_ct);

    if (c->fused && first_cross && R_ct > 0)
        return fused_step(c, R_ct, -1, first_cross, next_R, -1);
    if (c->eul_mask && first_cross && R_ct > 0) {
        const int zs = 2 * (c->nz / 2 + 1);
        const float *xe_dense = NULL;
        float *dw = c->delta_work; /* where this radius' filtered density spectrum is */
        if (s->use_ts_fluct) { /* delta and x_e through one sweep of passes X / Y (same window) */
            TRY(eul_xe_buffers(c));
            TRY(c21hip_split_filter_xy2(c->delta_unf, c->delta_work, s->hii_filter, 0.f, c->xe_unf,
                                        c->xe_work, s->hii_filter, 0.f, c->nx, c->ny, c->nz,
                                        s->box_len, s->box_len_z, R, apply, 0, 0, c->stream));
            TRY(c21hip_split_z_c2r(c->xe_work, c->eul_xe[0], c->nz, c->nx, c->ny, c->nz, c->stream));
            xe_dense = c->eul_xe[0];
        } else {
            /* (two radii per pass-X sweep: this radius' spectrum may be waiting in the other work buffer) */
            if (c->wev_pair && c->pairx_R == R_ct && c->pairx_buf) dw = c->pairx_buf;
            float *park = NULL;
            if (c->wev_pair) {
                if (!c->delta_work2)
                    c->delta_work2 = (float *)c21hip_ws(WS_DELTA_WORK2, c21hip_split_floats(c->nx, c->ny, c->nz) * sizeof(float));
                park = (dw == c->delta_work) ? c->delta_work2 : c->delta_work;
            }
            TRY(eul_filter_density(c, R_ct, next_R, dw, park));
        }
        if (s->fcoll_mode == C21CM_FCOLL_ERFC && !s->use_ts_fluct && eul_defer_ok(c)) {
            /* f_coll of this radius into one of two dense buffers; the barrier of the radius before
             * it (whose mean is known by now) in the same sweep; this radius' barrier follows with
             * the next radius, or with eul_flush_pending() when the mask is needed */
            const int cur = c->eul_pend >= 0 ? (c->eul_pend_buf ^ 1) : 0;
            float *nion_cur = cur ? c->nion_dense2 : c->nion_dense;
            if (c->eul_pend >= 0 && c->eul_pend_mask == first_cross) {
                TRY(c21hip_split_z_fcoll_erfc_mask(
                    dw, nion_cur, c->eul_pend_buf ? c->nion_dense2 : c->nion_dense,
                    c->scalars + SC_MEANS + c->eul_pend, first_cross, c->eul_pend, s->fix_mean,
                    s->mean_f_coll, s->mass_dep_zeta, s->f_limit_acg, s->ion_eff_factor, c->nx, c->ny,
                    c->nz, s->growth_factor, s->sigma_minmass, s->sigma_maxmass[R_ct], s->delta_c,
                    partials, sum_dev, c->stream));
                c->eul_pend = -1;
            } else {
                TRY(eul_flush_pending(c, 0));
                TRY(c21hip_split_z_fcoll_erfc(dw, nion_cur, c->nx, c->ny, c->nz,
                                              s->growth_factor, s->sigma_minmass,
                                              s->sigma_maxmass[R_ct], s->delta_c, partials, sum_dev,
                                              c->stream));
            }
            TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                                   mean_dev, c->stream));
            c->eul_pend = R_ct;
            c->eul_pend_buf = cur;
            c->eul_pend_mask = first_cross;
            goto done;
        }
        if (s->fcoll_mode == C21CM_FCOLL_ERFC && !s->use_ts_fluct && eul_band_ok(c)) {
            const int sig_ok = (float)s->sigma_maxmass[R_ct] != (float)s->sigma_minmass;
            const int banded = eul_band_this(c, R_ct, first_cross, sig_ok);
            if (banded) {
                TRY(c21hip_split_z_fcoll_erfc_band(
                    dw, c->nion_dense, c->scalars + SC_BAND + 2 * R_ct,
                    c->scalars + SC_BANDX + (c->band_pend >= 0 ? c->band_pend : 0), first_cross, R_ct,
                    c->band_pend, c->nx, c->ny, c->nz, s->growth_factor, s->sigma_minmass,
                    s->sigma_maxmass[R_ct], s->delta_c, partials, NULL, c->stream));
                c->band_pend = R_ct;
                c->band_mask = first_cross;
                c->band_used = 1;
            } else {
                TRY(eul_band_flush(c)); /* the dense sweep overwrites the marked cells' f_coll */
                TRY(c21hip_split_z_fcoll_erfc(dw, c->nion_dense, c->nx, c->ny, c->nz,
                                              s->growth_factor, s->sigma_minmass,
                                              s->sigma_maxmass[R_ct], s->delta_c, partials, NULL,
                                              c->stream));
            }
            /* the partial sums of the sweep, the mean, its threshold, the next band: one launch */
            TRY(eul_band_after(c, R_ct, next_R, banded, sig_ok, partials, (int)((long)c->nx * c->ny / 16),
                               sum_dev));
            if (!banded)
                TRY(c21hip_eulerian_mask(&args, c->nion_dense, NULL, mean_dev, first_cross, c->stream));
            goto done;
        }
        if (s->fcoll_mode == C21CM_FCOLL_ERFC) {
            TRY(c21hip_split_z_fcoll_erfc(dw, c->nion_dense, c->nx, c->ny, c->nz,
                                          s->growth_factor, s->sigma_minmass,
                                          s->sigma_maxmass[R_ct], s->delta_c, partials, sum_dev,
                                          c->stream));
        } else if (s->fcoll_mode == C21CM_FCOLL_NODES) {
            /* E-INTEGRAL without interpolation tables: the conditional integral per cell from the
             * radius' Gauss-Legendre node data (no extrema, no table) */
            TRY(c21hip_split_z_c2r(dw, c->delta_fil, zs, c->nx, c->ny, c->nz, c->stream));
            TRY(fcoll_nodes(c, R_ct, partials, sum_dev));
        } else {
            double mm[2];
            float table[C21CM_NDELTA_TABLE];
            TRY(c21hip_split_z_c2r_minmax(dw, c->delta_fil, zs, c->nx, c->ny, c->nz,
                                          partials, c->scalars + SC_MINMAX, c->stream));
            TRY(c21hip_d2h(mm, c->scalars + SC_MINMAX, sizeof(mm), c->stream));
            TRY(c21hip_sync(c->stream));
            const double min_density = mm[0] - 0.001, max_density = mm[1] + 0.001;
            int tst = s->table_fn(R_ct, min_density, max_density, table, s->table_user);
            if (tst) {
                c21hip_set_error("ionize: table_fn failed with status %d at radius %d", tst, R_ct);
                status = tst;
                goto done;
            }
            TRY(c21hip_h2d(c->table_dev, table, sizeof(table), c->stream));
            TRY(c21hip_sync(c->stream)); /* `table` is a stack buffer */
            TRY(c21hip_fcoll_eulerian(c->delta_fil, c->nion_dense, c->nx, c->ny, c->nz,
                                      s->fcoll_mode, s->growth_factor, s->sigma_minmass,
                                      s->sigma_maxmass[R_ct], s->delta_c, min_density,
                                      (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                      c->table_dev, partials, sum_dev, c->stream));
        }
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
        TRY(c21hip_eulerian_mask(&args, c->nion_dense, xe_dense, mean_dev, first_cross, c->stream));
        goto done;
    }
    TRY(filter_to_real(c, c->delta_unf, c->delta_work, c->delta_fil, s->hii_filter, R, 0.f,
                       apply));
    if (c->lagrangian)
        TRY(filter_to_real(c, c->stars_unf, c->stars_work, c->stars_fil, s->stars_filter, R,
                           (float)s->mfp_meandens, apply));
    if (s->use_ts_fluct && !(R_ct == 0 && c->r0_mask)) /* (the fused cell-scale sweep reads the x_e input itself) */
        TRY(filter_to_real(c, c->xe_unf, c->xe_work, c->xe_fil, s->hii_filter, R, 0.f, apply));

    if (c->recomb) {
        if (first_cross) { /* sharded runs use the key variant (c21cm_ionize_shard_radii_keys) */
            c21hip_set_error("ionize: a recombination model shards through the 64-bit key phases");
            status = C21CM_VALUE_ERROR;
            goto done;
        }
        if (c->lagrangian)
            TRY(filter_to_real(c, c->sfr_unf, c->sfr_work, c->sfr_fil, s->stars_filter, R,
                               (float)s->mfp_meandens, apply));
        if (c->filter_rec)
            TRY(filter_to_real(c, c->nrec_unf, c->nrec_work, c->nrec_fil, s->hii_filter, R, 0.f,
                               apply));
    }
    if (c->lag_mini) {
        if (first_cross) {
            c21hip_set_error("ionize: USE_MINI_HALOS does not run through the first-crossing mask");
            status = C21CM_VALUE_ERROR;
            goto done;
        }
        TRY(c21hip_ionise_mini(&args, 1, c->recomb, c->inhomo, s->cell_recomb, s->R[R_ct],
                               s->gamma_prefactor, 0., s->ion_eff_factor_mini, s->f_limit_mcg, 0.,
                               c->delta_fil, c->stars_fil, c->sfr_fil, c->xe_fil, c->nrec_fil,
                               c->prev_nrec, c->density, c->prev_zre, c->Tneutral, NULL, NULL, c->xH,
                               c->zre, c->Tk, c->G12, c->mfp, partials, sum_dev, c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
    } else if (c->recomb && c->lagrangian) {
        TRY(c21hip_ionise_recomb(&args, 1, c->inhomo, s->cell_recomb, s->R[R_ct],
                                 s->gamma_prefactor, c->delta_fil, c->stars_fil, c->sfr_fil,
                                 c->xe_fil, c->nrec_fil, c->prev_nrec, c->density, c->prev_zre,
                                 c->Tneutral, NULL, c->xH, c->zre, c->Tk, c->G12, c->mfp, partials,
                                 sum_dev, c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
    } else if (c->lagrangian) {
        TRY(c21hip_ionise_stars(&args, c->delta_fil, c->stars_fil, c->xe_fil, c->density,
                                c->prev_zre, c->Tneutral, c->xH, c->zre, c->Tk, first_cross,
                                partials, sum_dev, c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
    } else if (c->mini) {
        TRY(mini_radius(c, R_ct, &args, apply, R));
    } else {
        double tab_min = 0., tab_width = 1.;
        if (s->fcoll_mode == C21CM_FCOLL_NODES) {
            TRY(fcoll_nodes(c, R_ct, partials, sum_dev));
            TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                                   mean_dev, c->stream));
            goto nodes_done;
        }
        if (s->fcoll_mode >= C21CM_FCOLL_TABLE_LINEAR) {
            /* setup_integration_tables: IonisationBox.c:702-768 */
            double mm[2];
            float table[C21CM_NDELTA_TABLE];
            TRY(c21hip_clip_minmax(c->delta_fil, c->nx, c->ny, c->nz, partials,
                                   c->scalars + SC_MINMAX, c->stream));
            TRY(c21hip_d2h(mm, c->scalars + SC_MINMAX, sizeof(mm), c->stream));
            TRY(c21hip_sync(c->stream));
            const double min_density = mm[0] - 0.001, max_density = mm[1] + 0.001;
            int tst = s->table_fn(R_ct, min_density, max_density, table, s->table_user);
            if (tst) {
                c21hip_set_error("ionize: table_fn failed with status %d at radius %d", tst, R_ct);
                status = tst;
                goto done;
            }
            tab_min = min_density;
            tab_width = (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.);
            TRY(c21hip_h2d(c->table_dev, table, sizeof(table), c->stream));
            TRY(c21hip_sync(c->stream)); /* `table` is a stack buffer */
        }
        TRY(c21hip_fcoll_eulerian(c->delta_fil, c->nion_dense, c->nx, c->ny, c->nz, s->fcoll_mode,
                                  s->growth_factor, s->sigma_minmass, s->sigma_maxmass[R_ct],
                                  s->delta_c, tab_min, tab_width, c->table_dev, partials, sum_dev,
                                  c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
    nodes_done:
        if (c->recomb)
            TRY(c21hip_ionise_recomb(&args, 0, c->inhomo, s->cell_recomb, s->R[R_ct],
                                     s->gamma_prefactor, c->delta_fil, c->nion_dense, NULL,
                                     c->xe_fil, c->nrec_fil, c->prev_nrec, c->density, c->prev_zre,
                                     c->Tneutral, mean_dev, c->xH, c->zre, c->Tk, c->G12, c->mfp,
                                     partials, NULL, c->stream));
        else if (R_ct == 0 && c->r0_mask) {
            /* Eulerian loops, cell-scale radius: the first crossings of the larger radii, this radius'
             * barrier and partial ionisation and the post-loop sweep in ONE pass over the cells
             * (apply_first_cross + ionise_eulerian<LAST> + finalize: 2.3 -> 0.8 ms at 512^3); with an x_e grid
             * the sweep reads the x_e input itself -- at index 0 no window is applied (IonisationBox.c:606),
             * the filtered grid is the clipped input up to the rounding of a transform pair, as for the
             * emissivity grid of the Lagrangian path (r0_direct) */
            if (c->r0_slab) /* a rank's slab of the sweep; the caller exchanges the chunk sums and reduces them */
                TRY(c21hip_final_sweep_eulerian_range(&args, s->stored_redshift, c->r0_mask, c->nion_dense,
                                                      mean_dev, c->density, c->prev_zre, c->xH, c->zre, c->Tk,
                                                      c->partials, (int *)(c->scalars + SC_FLAG),
                                                      s->use_ts_fluct ? c->xe_dense : NULL, c->Tneutral,
                                                      c->r0_cb, c->r0_ce, c->stream));
            else
            TRY(c21hip_final_sweep_eulerian(&args, s->stored_redshift, c->r0_mask, c->nion_dense, mean_dev,
                                            c->density, c->prev_zre, c->xH, c->zre, c->Tk, c->partials,
                                            c->scalars + SC_XHSUM, (int *)(c->scalars + SC_FLAG),
                                            s->use_ts_fluct ? c->xe_dense : NULL, c->Tneutral, c->stream));
            c->finalised = 1;
        } else
            TRY(c21hip_ionise_eulerian(&args, c->nion_dense, c->xe_fil, c->density, c->prev_zre,
                                       c->Tneutral, mean_dev, c->xH, c->zre, c->Tk, first_cross,
                                       c->stream));
    }
done:
    return status;
}

/* Eulerian source models with a per-radius f_coll table (E-INTEGRAL, CONST-ION-EFF with
 * interpolation tables): the table of radius R needs the extrema of delta_R
 * (IonisationBox.c:702-765), i.e. a device -> host round trip and ~0.5 ms of host quadrature
 * per radius.  The loop is software-pipelined over two delta_R buffers: stage A (window, passes
 * X, Y, Z + extrema, async copy of the two numbers) of the NEXT radius is queued before the
 * host waits for this radius' extrema and builds its table, so the GPU transforms while the
 * host integrates; stage B (f_coll sweep, mean, barrier into the mask) follows in radius
 * order.  Barrier results only depend on the radius order of stage B, which is unchanged. */
/* x_e grids of a spin-temperature run: where the wave-level pass Z serves the line length, the x_e
 * spectrum of a radius stays in k-space (two work buffers, one per pipeline stage) until the
 * radius' f_coll mean is known, and its pass Z applies the barrier itself
 * (c21hip_split_z_xe_mask): x_e(R) is never written or read back as a grid, and
 * eulerian_mask_kernel is not launched (C21CM_XE_MASK_FUSED=0: the dense x_e(R) buffers). */
static int eul_xe_fused(const ion_ctx *c) {
    const char *e = getenv("C21CM_XE_MASK_FUSED");
    return c->s->use_ts_fluct && !(e && e[0] == '0') && c->xe_work2 != NULL &&
           c21hip_z_xe_mask_supported(c->nx, c->ny, c->nz);
}

/* Row length of the table loop's delta_R grids: dense rows (nz floats) where only the f_coll sweeps read
 * them -- they then move 16-byte pieces (round 5: 269 -> ... us per 512^3 sweep) -- padded rows with an x_e
 * grid, whose pass Z reads delta_R next to its own lines */
static long eul_dfil_stride(const ion_ctx *c) {
    const char *e = getenv("C21CM_EUL_DENSE_ROWS");
    if (!c->s->use_ts_fluct && c->nz % 4 == 0 && !(e && e[0] == '0')) return c->nz;
    return 2 * (long)(c->nz / 2 + 1);
}

/* `work`: the k-space buffer this radius' filtered density spectrum goes to; `store` = 0: the extrema alone
 * (the radius is expected to take the fused table sweep, which transforms `work` again) */
static int eul_stage_a(ion_ctx *c, int R_ct, int buf, float *delta_fil, double *mm_host, void *ev, float *work,
                       int store, int next_R, float *work_next, float *xwork, float *xwork_next) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    if (s->use_ts_fluct) { /* x_e shares the density grid's window (IonisationBox.c:1551-1553) */
        const int fused = eul_xe_fused(c);
        float *xe_work = xwork ? xwork : ((fused && buf) ? c->xe_work2 : c->xe_work);
        /* (work: delta_work, or delta_work2 for the second radius of a pass-X pair) */
        TRY(eul_filter_density_xe(c, R_ct, next_R, work, work_next, xe_work, xwork_next));
        if (!fused)
            TRY(c21hip_split_z_c2r(c->xe_work, c->eul_xe[buf], c->nz, c->nx, c->ny, c->nz, c->stream));
    } else {
        /* (pass X for this radius and the next where two radii ride one sweep: work_next != NULL) */
        TRY(eul_filter_density(c, R_ct, next_R, work, work_next));
    }
    if (!store)
        TRY(c21hip_split_z_minmax_only(work, c->nx, c->ny, c->nz, c->partials, c->scalars + SC_MINMAX + 2 * buf,
                                       c->stream));
    else
        TRY(c21hip_split_z_c2r_minmax(work, delta_fil, eul_dfil_stride(c), c->nx, c->ny, c->nz, c->partials,
                                      c->scalars + SC_MINMAX + 2 * buf, c->stream));
    TRY(c21hip_d2h(mm_host, c->scalars + SC_MINMAX + 2 * buf, 2 * sizeof(double), c->stream));
    TRY(c21hip_event_record(ev, c->stream));
done:
    return status;
}

/* A band missed at radius index `fail` and the host knows before the loop is over (the table loop waits
 * for every radius' extrema anyway, and the failure word travels with them): the stream drained, the
 * first-crossing grid rewound to its state before that radius, the band history put back to what it was
 * when that radius was entered; the caller restarts its pipeline AT that radius, which then takes the
 * dense sweeps (band_skip) -- the radii after it are banded again, with the miss in their error term.
 * A miss costs the two or three radii that were in flight, not the rest of the loop. */
static int eul_band_recover(ion_ctx *c, unsigned char *mask, int fail) {
    int status = 0;
    TRY(c21hip_sync(c->stream));
    TRY(c21hip_memset(c->scalars + SC_BANDFAIL, 0, sizeof(double), c->stream));
    TRY(c21hip_eul_rewind(mask, fail, c->ntot, c->stream));
    c->band_h1 = c->band_hist[fail][0], c->band_h2 = c->band_hist[fail][1], c->band_hn = c->band_hist[fail][2];
    c->band_next = -1;
    c->band_pend = -1;
    c->band_skip = fail;
    if (getenv("C21CM_EUL_BAND_DEBUG")) fprintf(stderr, "band miss at r=%d: recovered in the loop\n", fail);
done:
    return status;
}

static int eul_table_loop(ion_ctx *c, const int *radii, int n, unsigned char *mask) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    /* pinned host staging: [2][2] extrema, then [2][NDELTA] table floats */
    double(*mm)[2] = (double(*)[2])c21hip_pinned_host(4 * sizeof(double) +
                                                      2 * C21CM_NDELTA_TABLE * sizeof(float) + 2 * sizeof(double));
    float(*table)[C21CM_NDELTA_TABLE] = mm ? (float(*)[C21CM_NDELTA_TABLE])(mm + 2) : NULL;
    /* the banded barrier's failure word, copied out behind every radius' band step: [2] by stage */
    volatile int *fail_host = table ? (volatile int *)(table + 2) : NULL;
    void *ev[2] = {c21hip_event_create(), c21hip_event_create()};
    float *dfil[2] = {c->delta_fil, NULL};
    if (n <= 0) goto done;
    dfil[1] = (float *)c21hip_ws(WS_EUL_DFIL2, c->npad * sizeof(float));
    if (!ev[0] || !ev[1] || !dfil[1] || !mm) {
        status = C21CM_MEMORY_ALLOC_ERROR;
        goto done;
    }
    if (s->use_ts_fluct) {
        /* the second k-space buffer of the x_e grid (shared with the two-radius sweep's, idle here) */
        if (!c->xe_work2)
            c->xe_work2 = (float *)c21hip_ws(WS_XE_WORK2, c21hip_split_floats(c->nx, c->ny, c->nz) * sizeof(float));
        if (!eul_xe_fused(c)) TRY(eul_xe_buffers(c));
    }
    /* banded barrier: table modes without an x_e grid (its barrier needs x_e(R) of the cell, which only
     * the x_e grid's own pass Z holds).  (nz even: the kernel's two-cell items) */
    const int band_on = c->nz % 2 == 0 && c->ntot % 16 == 0 && s->n_radii < 255 && !c->band_off &&
                        !(getenv("C21CM_EUL_BAND") && getenv("C21CM_EUL_BAND")[0] == '0');
    const int use_band = band_on && !s->use_ts_fluct;
    /* ... and WITH an x_e grid (round 4, late): the x_e grid's pass Z does the table sweep as well and
     * decides the cells on a band of the mean fix itself (the barrier f mf zeta > 1 - x_e is monotone in
     * mf but has no single threshold); undecided cells leave (f, x_e).  Needs the fused x_e sweep. */
    const int use_band_xe = band_on && s->use_ts_fluct && eul_xe_fused(c) &&
                            c21hip_z_xe_fcoll_band_supported(c->nx, c->ny, c->nz);
    if (use_band_xe) {
        c->band_mf = 1;
        c->band_xe_pend = (float *)c21hip_ws(WS_EUL_XEPEND, c->ntot * sizeof(float));
        if (!c->band_xe_pend) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
    }
    /* Round 6: the table sweep + banded barrier as the epilogue of a second pass Z of the radius' spectrum
     * (pass Z EPI 9) -- delta_R is neither written nor read back (13 N -> 9 N bytes behind pass Y per radius;
     * DESIGN 4.3).  The radius' spectrum then has to outlive the stage A of the NEXT radius, which is enqueued
     * first: two k-space buffers, alternating.  Whether a radius will be banded is only known once the radius
     * before it has been swept; stage A predicts it (a band exists from the third radius of a run on) and stores
     * delta_R if it expects the dense sweeps; a wrong "no store" costs that radius one more pass Z.
     * C21CM_EUL_TABLE_FUSED=0: the store pass + sweep of round 5. */
    /* k-space buffers of the loop.  Fused table sweep: a radius' spectrum is transformed a second time AFTER the
     * stage A of the next radius has been enqueued -- two buffers, alternating; with two radii per pass-X sweep the
     * second radius of a pair is written (pass X) one radius early, while the radius before the pair still waits
     * for its table sweep -- four buffers, the pairs alternating between two sets. */
    float *wbuf[4] = {c->delta_work, c->delta_work, c->delta_work, c->delta_work};
    int stored[2] = {1, 1};
    const char *ef = getenv("C21CM_EUL_TABLE_FUSED");
    const int fuse_table = use_band && !(ef && ef[0] == '0') && c21hip_z_table_band_supported(c->nx, c->ny, c->nz);
    /* ... with an x_e grid (banded, fused x_e sweep): delta's spectrum only lives inside stage A (two buffers for the
     * two members of a pair), the x_e spectrum until the radius' sweep (four) */
    const int pair_xe = use_band_xe && c->wev_pair && !(ef && ef[0] == '0');
    float *xbuf[4] = {c->xe_work, c->xe_work2, c->xe_work, c->xe_work2};
    if (pair_xe) {
        const size_t wb = c21hip_split_floats(c->nx, c->ny, c->nz) * sizeof(float);
        if (!c->delta_work2) c->delta_work2 = (float *)c21hip_ws(WS_DELTA_WORK2, wb);
        /* every x_e spectrum shares its in-place pass Y with one of the two delta spectra: placed against it
         * (csrc/host/placement.c; plain allocations where the walk does not apply) */
        if (c->delta_work2) {
            xbuf[1] = c->xe_work2 = c21_place_work_partner(WS_DELTA_WORK2, WS_XE_WORK2, wb, c->nx, c->ny, c->nz, c->stream);
            xbuf[2] = c21_place_work_partner(WS_DELTA_WORK, WS_EUL_WORK3, wb, c->nx, c->ny, c->nz, c->stream);
            xbuf[3] = c21_place_work_partner(WS_DELTA_WORK2, WS_EUL_WORK4, wb, c->nx, c->ny, c->nz, c->stream);
        }
        if (!c->delta_work2 || !xbuf[1] || !xbuf[2] || !xbuf[3]) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        wbuf[1] = wbuf[3] = c->delta_work2; /* (member 1 of either set) */
    }
    const int pair_x = (fuse_table && c->wev_pair) || pair_xe;
    if (fuse_table) {
        const size_t wb = c21hip_split_floats(c->nx, c->ny, c->nz) * sizeof(float);
        if (!c->delta_work2) c->delta_work2 = (float *)c21hip_ws(WS_DELTA_WORK2, wb);
        wbuf[1] = c->delta_work2;
        if (pair_x) {
            wbuf[2] = (float *)c21hip_ws(WS_EUL_WORK3, wb);
            wbuf[3] = (float *)c21hip_ws(WS_EUL_WORK4, wb);
        } else {
            wbuf[2] = wbuf[0], wbuf[3] = wbuf[1];
        }
        if (!wbuf[1] || !wbuf[2] || !wbuf[3]) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
    }
#define EUL_EXPECT_BAND(i_) (fuse_table && (i_) - i0 >= 2 && radii[i_] >= 2 && radii[i_] != c->band_skip && \
                             (radii[i_] > s->r_lowest || s->r_lowest == 0))
    /* buffer of the radius at position i_ of this run (j = i_ - i0: pair j / 2 in set (j / 2) % 2, member j % 2) */
#define EUL_WORK(i_) (pair_x ? wbuf[(((((i_) - i0) >> 1) & 1) << 1) | (((i_) - i0) & 1)] : wbuf[((i_) - i0) & 1])
    /* the partner the pass X of position i_ also serves: the next radius, when i_ opens a pair */
#define EUL_NEXT_R(i_) ((pair_x && !(((i_) - i0) & 1) && (i_) + 1 < n) ? radii[(i_) + 1] : -1)
#define EUL_XWORK(i_) (pair_xe ? xbuf[(((((i_) - i0) >> 1) & 1) << 1) | (((i_) - i0) & 1)] : xbuf[(i_) & 1])
#define EUL_STAGE_A(i_) eul_stage_a(c, radii[i_], (i_) & 1, dfil[(i_) & 1], mm[(i_) & 1], ev[(i_) & 1], EUL_WORK(i_), \
                                    stored[(i_) & 1], EUL_NEXT_R(i_), EUL_NEXT_R(i_) >= 0 ? EUL_WORK((i_) + 1) : NULL, \
                                    s->use_ts_fluct && eul_xe_fused(c) ? EUL_XWORK(i_) : NULL,                           \
                                    (pair_xe && EUL_NEXT_R(i_) >= 0) ? EUL_XWORK((i_) + 1) : NULL)
    int i0 = 0;
    if (fail_host) fail_host[0] = fail_host[2] = 0;
restart:
    c->pairx_R = -1;
    stored[i0 & 1] = !EUL_EXPECT_BAND(i0);
    TRY(EUL_STAGE_A(i0));
    for (int i = i0; i < n; i++) {
        const int b = i & 1, R_ct = radii[i];
        float *const work_i = EUL_WORK(i);
        if (i + 1 < n) {
            stored[b ^ 1] = !EUL_EXPECT_BAND(i + 1);
            TRY(EUL_STAGE_A(i + 1));
        }
        TRY(c21hip_event_synchronize(ev[b]));
        if ((use_band || use_band_xe) && (fail_host[0] > 0 || fail_host[2] > 0)) {
            /* (the word copied behind the band step of radius i - 2 or earlier has arrived with this event) */
            const int fail = fail_host[0] > fail_host[2] ? fail_host[0] : fail_host[2];
            TRY(eul_band_recover(c, mask, fail));
            fail_host[0] = fail_host[2] = 0;
            for (i0 = 0; i0 < n && radii[i0] != fail; i0++) {}
            if (i0 == n) { /* (cannot happen: the word names a radius of this loop) */
                c21hip_set_error("ionize: banded barrier reported a radius outside the loop");
                status = C21CM_VALUE_ERROR;
                goto done;
            }
            goto restart;
        }
        const double min_density = mm[b][0] - 0.001, max_density = mm[b][1] + 0.001;
        int tst = s->table_fn(R_ct, min_density, max_density, table[b], s->table_user);
        if (tst) {
            c21hip_set_error("ionize: table_fn failed with status %d at radius %d", tst, R_ct);
            status = tst;
            goto done;
        }
        float *table_dev = c->table_dev + b * C21CM_NDELTA_TABLE;
        TRY(c21hip_h2d(table_dev, table[b], C21CM_NDELTA_TABLE * sizeof(float), c->stream));
        double *sum_dev = c->scalars + SC_SUMS + R_ct, *mean_dev = c->scalars + SC_MEANS + R_ct;
        c21hip_ionize_args args;
        fill_args(&args, s, R_ct);
        if (use_band_xe) {
            const int banded = eul_band_this(c, R_ct, mask, 1);
            const float *xw = EUL_XWORK(i);
            if (banded) {
                TRY(c21hip_split_z_xe_fcoll_band(
                    xw, dfil[b], eul_dfil_stride(c), c->nion_dense, c->band_xe_pend,
                    c->scalars + SC_BAND + 2 * R_ct,
                    c->scalars + SC_BANDX + (c->band_pend >= 0 ? c->band_pend : 0), mask, R_ct, c->band_pend,
                    s->fcoll_mode, min_density, (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                    table_dev, s->mass_dep_zeta, s->f_limit_acg, s->ion_eff_factor, c->nx, c->ny, c->nz,
                    c->partials, c->stream));
                c->band_pend = R_ct;
                c->band_mask = mask;
                c->band_used = 1;
                TRY(eul_band_after(c, R_ct, i + 1 < n ? radii[i + 1] : -1, 1, 1, c->partials,
                                   (int)((long)c->nx * c->ny / 16), sum_dev));
            } else {
                TRY(eul_band_flush(c)); /* the dense sweep overwrites the marked cells' f_coll */
                TRY(c21hip_fcoll_eulerian_zs(dfil[b], eul_dfil_stride(c), c->nion_dense, c->nx, c->ny, c->nz, s->fcoll_mode,
                                          s->growth_factor, s->sigma_minmass, s->sigma_maxmass[R_ct],
                                          s->delta_c, min_density,
                                          (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                          table_dev, c->partials, sum_dev, c->stream));
                TRY(eul_band_after(c, R_ct, i + 1 < n ? radii[i + 1] : -1, 0, 1, NULL, 0, sum_dev));
                TRY(c21hip_split_z_xe_mask(xw, c->nion_dense, mean_dev, mask, c->nx, c->ny, c->nz, R_ct,
                                           s->mean_f_coll, s->fix_mean, s->mass_dep_zeta, s->f_limit_acg,
                                           s->ion_eff_factor, c->stream));
            }
            TRY(c21hip_d2h((void *)(fail_host + 2 * b), c->scalars + SC_BANDFAIL, sizeof(int), c->stream));
            continue;
        }
        if (use_band) {
            /* banded barrier (see eul_band_ok): the table sweep decides the cells itself, no dense
             * f_coll grid, no barrier sweep */
            const int banded = eul_band_this(c, R_ct, mask, 1);
            int n_part = 0;
            if (banded && !stored[b]) {
                /* the second pass Z of the radius' spectrum: table sweep + banded barrier in its epilogue */
                TRY(c21hip_split_z_fcoll_table_band(work_i, c->nion_dense, c->scalars + SC_BAND + 2 * R_ct,
                                                    c->scalars + SC_BANDX + (c->band_pend >= 0 ? c->band_pend : 0),
                                                    mask, R_ct, c->band_pend, c->nx, c->ny, c->nz, s->fcoll_mode,
                                                    min_density, (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                                    table_dev, c->partials, c->stream));
                c->band_pend = R_ct;
                c->band_mask = mask;
                c->band_used = 1;
                TRY(eul_band_after(c, R_ct, i + 1 < n ? radii[i + 1] : -1, 1, 1, c->partials,
                                   (int)((long)c->nx * c->ny / 16), sum_dev));
                TRY(c21hip_d2h((void *)(fail_host + 2 * b), c->scalars + SC_BANDFAIL, sizeof(int), c->stream));
                continue;
            }
            if (!stored[b]) { /* expected a band, got none: delta_R after all (the spectrum is still in work[b]) */
                TRY(c21hip_split_z_c2r_minmax(work_i, dfil[b], eul_dfil_stride(c), c->nx, c->ny, c->nz, c->partials,
                                              c->scalars + SC_MINMAX + 2 * b, c->stream));
                stored[b] = 1;
            }
            if (banded) {
                TRY(c21hip_fcoll_eulerian_band(dfil[b], eul_dfil_stride(c), c->nion_dense, mask, c->nx, c->ny, c->nz,
                                               s->fcoll_mode, min_density,
                                               (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                               table_dev, c->scalars + SC_BAND + 2 * R_ct,
                                               c->scalars + SC_BANDX + (c->band_pend >= 0 ? c->band_pend : 0),
                                               R_ct, c->band_pend, c->partials, &n_part, c->stream));
                c->band_pend = R_ct;
                c->band_mask = mask;
                c->band_used = 1;
                TRY(eul_band_after(c, R_ct, i + 1 < n ? radii[i + 1] : -1, 1, 1, c->partials, n_part, sum_dev));
            } else {
                TRY(eul_band_flush(c)); /* the dense sweep overwrites the marked cells' f_coll */
                TRY(c21hip_fcoll_eulerian_zs(dfil[b], eul_dfil_stride(c), c->nion_dense, c->nx, c->ny, c->nz, s->fcoll_mode,
                                          s->growth_factor, s->sigma_minmass, s->sigma_maxmass[R_ct],
                                          s->delta_c, min_density,
                                          (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                          table_dev, c->partials, sum_dev, c->stream));
                TRY(eul_band_after(c, R_ct, i + 1 < n ? radii[i + 1] : -1, 0, 1, NULL, 0, sum_dev));
                TRY(c21hip_eulerian_mask(&args, c->nion_dense, NULL, mean_dev, mask, c->stream));
            }
            TRY(c21hip_d2h((void *)(fail_host + 2 * b), c->scalars + SC_BANDFAIL, sizeof(int), c->stream));
            continue;
        }
        TRY(c21hip_fcoll_eulerian_zs(dfil[b], eul_dfil_stride(c), c->nion_dense, c->nx, c->ny, c->nz, s->fcoll_mode,
                                  s->growth_factor, s->sigma_minmass, s->sigma_maxmass[R_ct],
                                  s->delta_c, min_density,
                                  (max_density - min_density) / (C21CM_NDELTA_TABLE - 1.),
                                  table_dev, c->partials, sum_dev, c->stream));
        TRY(c21hip_finish_mean(sum_dev, (double)c->ntot, s->mass_dep_zeta, s->f_limit_acg,
                               mean_dev, c->stream));
        if (eul_xe_fused(c))
            TRY(c21hip_split_z_xe_mask(EUL_XWORK(i), c->nion_dense, mean_dev, mask, c->nx,
                                       c->ny, c->nz, R_ct, s->mean_f_coll, s->fix_mean, s->mass_dep_zeta,
                                       s->f_limit_acg, s->ion_eff_factor, c->stream));
        else
            TRY(c21hip_eulerian_mask(&args, c->nion_dense, s->use_ts_fluct ? c->eul_xe[b] : NULL, mean_dev,
                                     mask, c->stream));
    }
    /* the pinned staging buffer is shared: make sure the last copies have left it */
    TRY(c21hip_sync(c->stream));
#undef EUL_EXPECT_BAND
#undef EUL_WORK
#undef EUL_XWORK
#undef EUL_NEXT_R
#undef EUL_STAGE_A
done:
    c21hip_event_destroy(ev[0]);
    c21hip_event_destroy(ev[1]);
    return status;
}

/* Fused Lagrangian path, radius index 0 reached: the cell-scale step only needs the emissivity
 * grid in real space (the density it tests is the unfiltered one, IonisationBox.c:1048), and
 * the mask of the larger radii, the barrier / partial ionisation at index 0 and the post-loop
 * sweep are one pass over the cells (c21hip_final_sweep). */
static int final_prepare(ion_ctx *c) {
    const c21cm_ionize_spec *s = c->s;
    return c21hip_split_filter_c2r(c->stars_unf, c->stars_work, c->stars_fil,
                                   2 * (long)(c->nz / 2 + 1), c->nx, c->ny, c->nz, s->box_len,
                                   s->box_len_z, s->stars_filter, (float)s->R[0],
                                   (float)s->mfp_meandens, 0, c->stream);
}

/* At radius index 0 no window is applied (IonisationBox.c:606), so the reference's filtered
 * emissivity is the transform round trip of the clipped input: the clipped input itself up to
 * the rounding of an FFT pair (~1e-7 relative, far inside the 1e-4 tolerance of the path).
 * The final sweep therefore reads the input directly; C21CM_R0_ROUNDTRIP=1 restores the
 * round trip through passes X, Y, Z (three extra sweeps of one grid). */
static int r0_direct(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("C21CM_R0_ROUNDTRIP");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v;
}

/* chunks [chunk_begin, chunk_end) of the final sweep (chunk_end < 0: all): their partial sums stay in
 * c->partials; final_step_sums() reduces ALL chunks in a fixed order */
static int final_step_range(ion_ctx *c, const unsigned char *mask, int stars_ready, int chunk_begin,
                            int chunk_end) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    c21hip_ionize_args args;
    fill_args(&args, s, 0);
    const int direct = r0_direct();
    if (!direct && !stars_ready) TRY(final_prepare(c));
    TRY(c21hip_final_sweep_range(&args, s->stored_redshift, mask, direct ? c->n_ion : c->stars_fil,
                                 c->density, c->prev_zre, c->xH, c->zre, c->Tk, c->partials,
                                 (int *)(c->scalars + SC_FLAG), direct, c->xe_dense, c->Tneutral,
                                 chunk_begin, chunk_end, c->stream));
done:
    return status;
}

static int final_step_sums(ion_ctx *c) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    c21hip_ionize_args args;
    fill_args(&args, s, 0);
    TRY(c21hip_final_sweep_reduce(&args, r0_direct(), c->partials, c->scalars + SC_SUMS,
                                  c->scalars + SC_XHSUM, c->stream));
    TRY(c21hip_finish_mean(c->scalars + SC_SUMS, (double)c->ntot, s->mass_dep_zeta,
                           s->f_limit_acg, c->scalars + SC_MEANS, c->stream));
    c->finalised = 1;
done:
    return status;
}

static int final_step(ion_ctx *c, const unsigned char *mask, int stars_ready) {
    int status = 0;
    TRY(final_step_range(c, mask, stars_ready, 0, -1));
    TRY(final_step_sums(c));
done:
    return status;
}

/* post-loop + result collection: IonisationBox.c:1589-1628 */
static int postloop(ion_ctx *c, IonizedBox *box, c21cm_ionize_report *report) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    c21hip_ionize_args args;
    fill_args(&args, s, 0);
    double host_sc[SC_COUNT - SC_SUMS];
    int *flag_dev = (int *)(c->scalars + SC_FLAG);
    if (!c->finalised)
        TRY(c21hip_finalize(&args, s->stored_redshift, c->density, c->Tneutral, c->xH, c->zre,
                            c->Tk, c->ntot, c->partials, c->scalars + SC_XHSUM, flag_dev,
                            c->stream));
    if (c->recomb && c->inhomo) /* set_recombination_rates, IonisationBox.c:1277-1339 */
        TRY(c21hip_recomb_rates(c->density, c->G12, c->xH, c->prev_nrec, c->nrec_out, c->ntot,
                                s->stored_redshift, s->fabs_dtdz * s->dz, c->rr_dev,
                                c->rr_dev + (size_t)C21CM_RR_NZ * C21CM_RR_NGAMMA, flag_dev,
                                c->stream));
    if (c->recomb && !c->inhomo) /* the global Gamma_12 of the homogeneous model, :1600-1609 */
        TRY(c21hip_sum_float(c->G12, c->ntot, c->partials, c->scalars + SC_G12SUM, c->stream));
    TRY(c21hip_d2h(host_sc, c->scalars + SC_SUMS, sizeof(host_sc), c->stream));
    for (int i = 0; i < c->cb.n; i++) {
        if (c->cb_ncell && c->cb.bytes[i] == c->ntot * sizeof(float)) { /* a rank's slab of a per-cell grid */
            TRY(c21hip_d2h((float *)c->cb.host[i] + c->cb_cell0, (float *)c->cb.dev[i] + c->cb_cell0,
                           c->cb_ncell * sizeof(float), c->stream));
            continue;
        }
        TRY(c21hip_d2h(c->cb.host[i], c->cb.dev[i], c->cb.bytes[i], c->stream));
    }
    TRY(c21hip_sync(c->stream));
    {
        const double *means = host_sc + (SC_MEANS - SC_SUMS);
        double global_xH = host_sc[SC_XHSUM - SC_SUMS];
        int flag;
        memcpy(&flag, &host_sc[SC_FLAG - SC_SUMS], sizeof(int));
        global_xH /= (float)c->ntot; /* IonisationBox.c:1607 */
        if (flag || !isfinite(global_xH)) {
            c21hip_set_error("ionize: non-finite %s",
                             flag ? "kinetic temperature or recombination count" : "neutral fraction");
            status = C21CM_INFINITY_OR_NAN_ERROR;
            goto done;
        }
        if (c->recomb && !c->inhomo) { /* IonisationBox.c:1261-1276 */
            double c21_rr_eval(const double *rr_y, const double *rr_c, double z_eff, double g);
            const float global_g12 = (float)(host_sc[SC_G12SUM - SC_SUMS] / (float)c->ntot);
            const float global_xHI = (float)global_xH;
            float prev0; /* (staged or caller-owned: prev_nrec is a device address either way) */
            TRY(c21hip_d2h(&prev0, c->prev_nrec, sizeof(float), c->stream));
            TRY(c21hip_sync(c->stream));
            const double dNrec = c21_rr_eval(s->rr_y, s->rr_c, s->stored_redshift, global_g12) *
                                 s->fabs_dtdz * s->dz * (1. - global_xHI);
            const double cum = (double)prev0 + dNrec;
            if (!isfinite(cum)) {
                c21hip_set_error("ionize: non-finite cumulative recombinations");
                status = C21CM_INFINITY_OR_NAN_ERROR;
                goto done;
            }
            const float cumf = (float)cum;
            if (c21hip_is_device_ptr(box->cumulative_recombinations)) {
                TRY(c21hip_h2d(box->cumulative_recombinations, &cumf, sizeof(float), c->stream));
                TRY(c21hip_sync(c->stream));
            } else {
                box->cumulative_recombinations[0] = cumf;
            }
        }
        const int last = s->r_lowest < s->n_radii ? s->r_lowest : s->n_radii - 1;
        const double mean_out = s->fix_mean ? s->mean_f_coll : means[last];
        box->mean_f_coll = mean_out; /* IonisationBox.c:1623-1628 */
        const double *means_m = host_sc + (SC_MEANS_M - SC_SUMS);
        if (c->mini) /* IonisationBox.c:914,943 */
            for (int r = s->r_lowest; r < s->n_radii; r++)
                if (!isfinite(means[r]) || !isfinite(means_m[r])) {
                    c21hip_set_error("ionize: f_coll is either infinite or NaN at radius %d", r);
                    status = C21CM_INFINITY_OR_NAN_ERROR;
                    goto done;
                }
        /* (Lagrangian grids: the grid mean of the second population is 0, clamped to its floor,
         * IonisationBox.c:1570-1573) */
        const double lag_floor = s->mass_dep_zeta ? s->f_limit_mcg : 0.;
        const double mean_m_out = c->lag_mini ? (s->fix_mean ? s->mean_f_coll_mini : lag_floor)
                                  : !c->mini  ? 0.
                                              : (s->fix_mean ? s->mean_f_coll_mini : means_m[last]);
        box->mean_f_coll_MINI = mean_m_out;
        if (report) {
            for (int r = 0; r < s->n_radii; r++) report->f_coll_grid_mean[r] = means[r];
            report->global_xH = global_xH;
            report->mean_f_coll_out = mean_out;
            if (c->mini)
                for (int r = 0; r < s->n_radii; r++) report->f_coll_grid_mean_mini[r] = means_m[r];
            if (c->lag_mini)
                for (int r = s->r_lowest; r < s->n_radii; r++) report->f_coll_grid_mean_mini[r] = lag_floor;
            report->mean_f_coll_mini_out = mean_m_out;
        }
    }
done:
    return status;
}

static int init_output_grids(ion_ctx *c, const IonizedBox *prev) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    /* IonisationBox.c:1372-1378 (final_step writes every cell, -1 included) */
    /* (... and the slab finish of the Eulerian models: its one sweep writes every z_reion of the slab, the
     * rest of the array is not this rank's) */
    if (!(c->fused && !c->fused_rc && !c->sphere && s->r_lowest == 0) && !c->r0_slab)
        TRY(c21hip_fill(c->zre, c->ntot, -1.0f, c->stream));
    /* IonisationBox.c:365-386: the caller's zeroed previous box receives z_reion = -1 */
    if (s->first_snapshot && prev && prev->z_reion) {
        if (c21hip_is_device_ptr(prev->z_reion)) {
            TRY(c21hip_fill(prev->z_reion, c->ntot, -1.0f, c->stream));
        } else {
            for (size_t i = 0; i < c->ntot; i++) prev->z_reion[i] = -1.0f;
        }
    }
done:
    return status;
}

/* IONISE_ENTIRE_SPHERE: the spheres of all first crossings (IonisationBox.c:1150-1158) */
static int paint_spheres(ion_ctx *c, const unsigned char *mask) {
    int status = 0;
    const c21cm_ionize_spec *s = c->s;
    float rsq[C21CM_MAX_RADII];
    for (int r = 0; r < s->n_radii; r++) rsq[r] = sphere_rsq(s, r);
    float *rsq_dev = (float *)c21hip_ws(WS_SPHERE_RSQ, sizeof(rsq));
    if (!rsq_dev) return C21CM_MEMORY_ALLOC_ERROR;
    TRY(c21hip_h2d(rsq_dev, rsq, sizeof(rsq), c->stream));
    TRY(c21hip_sync(c->stream)); /* `rsq` is a stack buffer */
    TRY(c21hip_paint_spheres(mask, rsq_dev, c->xH, c->nx, c->ny, c->nz, c->stream));
done:
    return status;
}

/* The Eulerian loops' cell-scale radius + post-loop as one sweep (one_radius, R_ct = 0, c->r0_mask)?
 * C21CM_EUL_R0_FUSED=0: the general kernels. */
static int eul_r0_fused(const ion_ctx *c) {
    const char *e = getenv("C21CM_EUL_R0_FUSED");
    return c->eul_mask && !c->sphere && !c->recomb && c->s->r_lowest == 0 && !(e && e[0] == '0') &&
           (!c->s->use_ts_fluct || r0_direct());
}

/* Eulerian source models (and every other loop off the fused path) on the native passes: the density
 * (and x_e) windows of the radii first, first - step, ... evaluated inside pass X (top-hat / sharp-k
 * HII_FILTER), no window tables.  The single pass AND the shard phases call this, so that a rank's
 * radii see the very arithmetic of the single pass: the evaluated windows (quintic Hermite in float)
 * and the table fallback (fp64, rounded) differ by an ulp or two of W, which is enough to flip a cell
 * within float noise of a barrier -- the sharded result is only bit-identical by construction if both
 * use the same one (round 4: the shard phases did not, and test_gpu_config5's comparison failed on about
 * one box in three). */
static int native_wev_prepare(ion_ctx *c, int first, int step, void *stream) {
    const c21cm_ionize_spec *spec = c->s;
    if (!c->native || c->fused) return 0;
    float radii[C21CM_MAX_RADII];
    int n = 0, on = 0;
    for (int R_ct = first; R_ct >= 1 && R_ct >= spec->r_lowest && n < C21CM_MAX_RADII; R_ct -= step)
        radii[n++] = (float)spec->R[R_ct];
    if (n == 0) return 0;
    /* one filtered grid: two radii per pass-X sweep where two line tiles fit the LDS (C21CM_EUL_PAIR=0: one) */
    const char *ep = getenv("C21CM_EUL_PAIR");
    const int pair = c->eul_mask && !(ep && ep[0] == '0') && c21hip_pair_sweep_supported(c->nx);
    const int st = c21hip_wev_prepare(spec->hii_filter, 0.f, spec->hii_filter, 0.f, spec->use_ts_fluct ? 2 : 1,
                                      radii, n, c->nx, c->ny, c->nz, spec->box_len, spec->box_len_z, pair, &on,
                                      stream);
    c->wev_pair = !st && on && pair;
    return st;
}

int c21cm_ionize_grids(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                       const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                       const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                       void *stream) {
    int status = validate_spec(spec, perturbed_field, halos, spin_temp, box);
    if (status) return status;
    ion_ctx c;
    void *ev[4] = {NULL, NULL, NULL, NULL};
    g_spectra.valid = 0;
    g_single_pass = 1;
    status = ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, box, 1, stream);
    g_single_pass = 0;
    if (status) goto done;
    for (int i = 0; i < 4; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    TRY(init_output_grids(&c, previous_ionize_box));
    TRY(preloop(&c));
    TRY(c21hip_event_record(ev[1], stream));
    if (c.fused || c.eul_mask || c.sphere) {
        c.mask = (unsigned char *)c21hip_ws(WS_FIRST_CROSS, c.ntot);
        if (!c.mask) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        TRY(c21hip_memset(c.mask, 0, c.ntot, stream));
    }
    {
        const int use_mask = c.fused || c.eul_mask || c.sphere;
        int mask_pending = use_mask;
        int R_start = spec->n_radii;
        /* (the fused Lagrangian loop prepares its own set; grids with other windows keep their tables) */
        TRY(native_wev_prepare(&c, spec->n_radii - 1, 1, stream));
        if (c.eul_mask && spec->fcoll_mode != C21CM_FCOLL_ERFC && spec->fcoll_mode != C21CM_FCOLL_NODES) {
            int radii[C21CM_MAX_RADII], n = 0;
            for (int R_ct = spec->n_radii - 1; R_ct >= 1 && R_ct >= spec->r_lowest; R_ct--)
                radii[n++] = R_ct;
            TRY(eul_table_loop(&c, radii, n, c.mask));
            {
                int redo = 0; /* banded barrier: markers settled; a missed band reruns its radii on the
                               * dense sweeps, through the pipelined loop again */
                TRY(eul_band_finish(&c, c.mask, &redo));
                if (redo) {
                    int m = 0;
                    for (int i = 0; i < n; i++)
                        if (radii[i] <= redo) radii[m++] = radii[i];
                    TRY(eul_table_loop(&c, radii, m, c.mask));
                }
            }
            R_start = 1; /* only the cell-scale radius is left */
        }
        if (c.fused) { /* radii n-1 .. 1 through the fused steps; index 0 is the final sweep */
            TRY(fused_loop(&c, spec->n_radii - 1, 1, spec->r_lowest, c.mask));
            R_start = 1;
        }
        for (int R_ct = R_start; R_ct--;) {
            if (R_ct < spec->r_lowest) break; /* IonisationBox.c:1537-1541 */
            if (R_ct == 0 && mask_pending) {
                TRY(eul_flush_pending(&c, 0)); /* the last radius' barrier of the closed-form loop */
                {
                    int redo = 0; /* banded barrier: markers settled; a missed band reruns its radii */
                    TRY(eul_band_finish(&c, c.mask, &redo));
                    if (redo) {
                        R_ct = redo + 1;
                        continue;
                    }
                }
                mask_pending = 0;
                if (c.fused && !c.sphere && !c.fused_rc) {
                    TRY(flush_deferred(&c));
                    TRY(final_step(&c, c.mask, 0));
                    break;
                }
                if (c.fused) TRY(flush_deferred(&c));
                if (c.fused_rc) { /* first crossings -> x_HI, z_reion, mean free path (Gamma_12 is in place) */
                    float Rf[C21CM_MAX_RADII];
                    for (int r = 0; r < C21CM_MAX_RADII; r++) Rf[r] = r < spec->n_radii ? (float)spec->R[r] : 0.f;
                    float *R_dev = (float *)c21hip_ws(WS_R_DEV, sizeof(Rf));
                    if (!R_dev) {
                        status = C21CM_MEMORY_ALLOC_ERROR;
                        goto done;
                    }
                    TRY(c21hip_h2d(R_dev, Rf, sizeof(Rf), stream));
                    TRY(c21hip_sync(stream)); /* `Rf` is a stack buffer */
                    TRY(c21hip_apply_first_cross_recomb(c.mask, R_dev, c.prev_zre, spec->first_snapshot,
                                                        spec->redshift, c.xH, c.zre, c.mfp, c.ntot, stream));
                    TRY(one_radius(&c, 0, NULL, -1));
                    continue;
                }
                if (eul_r0_fused(&c)) { /* one sweep: mask + cell-scale radius + post-loop (one_radius) */
                    c.r0_mask = c.mask;
                } else {
                    /* the cell-scale radius tests xH > TINY: materialise the mask first */
                    TRY(c21hip_apply_first_cross(c.mask, c.prev_zre, spec->first_snapshot,
                                                 spec->redshift, c.xH, c.zre, c.ntot, stream));
                    if (c.sphere) TRY(paint_spheres(&c, c.mask));
                }
            }
            TRY(one_radius(&c, R_ct, (R_ct > 0 && use_mask) ? c.mask : NULL,
                           (R_ct - 1 >= spec->r_lowest) ? R_ct - 1 : -1));
        }
        TRY(flush_deferred(&c));
        if (mask_pending) {
            int redo = 0; /* a loop that stopped above index 0 (r_lowest): same check as above */
            TRY(eul_band_finish(&c, c.mask, &redo));
            if (redo) {
                for (int R_ct = redo; R_ct >= spec->r_lowest && R_ct >= 1; R_ct--)
                    TRY(one_radius(&c, R_ct, c.mask, (R_ct - 1 >= spec->r_lowest) ? R_ct - 1 : -1));
                TRY(flush_deferred(&c));
            }
            TRY(c21hip_apply_first_cross(c.mask, c.prev_zre, spec->first_snapshot, spec->redshift,
                                         c.xH, c.zre, c.ntot, stream));
            if (c.sphere) TRY(paint_spheres(&c, c.mask));
        }
    }
    c21hip_wev_release();
    TRY(c21hip_event_record(ev[2], stream));
    TRY(postloop(&c, box, report));
    TRY(c21hip_event_record(ev[3], stream));
    if (report) {
        report->ms_preloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_rloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
        report->ms_postloop = c21hip_event_elapsed_ms(ev[2], ev[3]);
    }
done:
    c21hip_wev_release();
    for (int i = 0; i < 4; i++) c21hip_event_destroy(ev[i]);
    return status;
}

/* ---- R-loop sharding ------------------------------------------------- */
static int no_mini_shards(void) {
    /* every radius of a USE_MINI_HALOS run reads and writes its own slice of the f_coll history,
     * which a rank-local shard would have to exchange as well: not built */
    c21hip_set_error("ionize: the sharded R loop does not take USE_MINI_HALOS");
    return C21CM_VALUE_ERROR;
}

/* Per-radius f_coll grid means of the shard phase.  Each radius > 0 is owned by exactly one rank,
 * so the caller sums the report arrays of all ranks (a 2 KB all-reduce) and hands the result to
 * the finishing rank, whose finish phase starts from a zeroed scalar block.  Without this the
 * Lagrangian box->mean_f_coll (= the mean of the last radius processed, IonisationBox.c:
 * 1623-1628) would be lost whenever that radius is not index 0. */
static struct {
    int valid, n;
    double means[C21CM_MAX_RADII];
} g_shard_means;

int c21cm_ionize_shard_set_means(const double *means, int n_radii) {
    if (!means || n_radii < 0 || n_radii > C21CM_MAX_RADII) {
        g_shard_means.valid = 0;
        return means ? C21CM_VALUE_ERROR : 0;
    }
    memset(g_shard_means.means, 0, sizeof(g_shard_means.means));
    memcpy(g_shard_means.means, means, sizeof(double) * (size_t)n_radii);
    g_shard_means.n = n_radii;
    g_shard_means.valid = 1;
    return 0;
}

int c21cm_ionize_shard_radii(const c21cm_ionize_spec *spec, int rank, int world,
                             const PerturbedField *perturbed_field,
                             const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                             const HaloBox *halos, unsigned char *first_cross,
                             c21cm_ionize_report *report, void *stream) {
    IonizedBox dummy;
    memset(&dummy, 0, sizeof(dummy));
    int status = 0;
    if (!first_cross || !c21hip_is_device_ptr(first_cross) || world < 1 || rank < 0 ||
        rank >= world) {
        c21hip_set_error("ionize shard: first_cross must be a device array, 0 <= rank < world");
        return C21CM_VALUE_ERROR;
    }
    if (spec && spec->fcoll_mode != C21CM_FCOLL_STARS_GRID) {
        /* Eulerian models need unnormalised_nion scratch; give the ctx a slot-backed one */
        dummy.unnormalised_nion = (float *)c21hip_ws(
            WS_NION_DENSE, (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z * sizeof(float));
    }
    /* validate with stand-in outputs: the shard phase never touches xH / z_reion / T_k */
    {
        IonizedBox probe = dummy;
        float sentinel;
        probe.neutral_fraction = &sentinel;
        probe.z_reion = &sentinel;
        probe.kinetic_temperature = &sentinel;
        status = validate_spec(spec, perturbed_field, halos, spin_temp, &probe);
        if (!status && spec->use_mini_halos) status = no_mini_shards();
        if (status) return status;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    TRY(ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, &dummy, 0,
                  stream));
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    TRY(c21hip_memset(first_cross, 0, c.ntot, stream));
    g_spectra.valid = 0;
    TRY(preloop(&c));
    spectra_remember(&c, perturbed_field, halos, spin_temp);
    TRY(c21hip_event_record(ev[1], stream));
    TRY(native_wev_prepare(&c, spec->n_radii - 1, 1, stream)); /* the single pass' node tables: the whole ladder */
    /* radii n-1 .. 1 dealt round-robin, largest first; index 0 belongs to the finish step */
    if (c.eul_mask && spec->fcoll_mode != C21CM_FCOLL_ERFC && spec->fcoll_mode != C21CM_FCOLL_NODES) {
        int radii[C21CM_MAX_RADII], n = 0;
        for (int R_ct = spec->n_radii - 1 - rank; R_ct >= 1 && R_ct >= spec->r_lowest; R_ct -= world)
            radii[n++] = R_ct;
        TRY(eul_table_loop(&c, radii, n, first_cross));
        {
            int redo = 0; /* banded barrier: markers settled; a missed band reruns its radii (dense) */
            TRY(eul_band_finish(&c, first_cross, &redo));
            if (redo) {
                int m = 0;
                for (int i = 0; i < n; i++)
                    if (radii[i] <= redo) radii[m++] = radii[i];
                TRY(eul_table_loop(&c, radii, m, first_cross));
            }
        }
    } else if (c.fused) {
        TRY(fused_loop(&c, spec->n_radii - 1 - rank, world, spec->r_lowest, first_cross));
    } else {
        for (int attempt = 0, from = spec->n_radii; attempt < 2; attempt++) {
            int redo = 0;
            for (int R_ct = spec->n_radii - 1 - rank; R_ct >= 1; R_ct -= world) {
                if (R_ct < spec->r_lowest) break;
                if (R_ct > from) continue; /* second attempt: from the radius whose band missed */
                TRY(one_radius(&c, R_ct, first_cross, R_ct - world));
            }
            TRY(eul_flush_pending(&c, 1));
            TRY(eul_band_finish(&c, first_cross, &redo)); /* banded barrier: markers settled, bands checked */
            if (!redo) break;
            from = redo;
        }
    }
    TRY(flush_deferred(&c));
    /* The rank that will run the finish step has one radius fewer than the busiest ranks: it
     * uses that slack to transform the unfiltered emissivity for the cell-scale step, so that
     * after the reduce only the final sweep remains. */
    if (c.fused && spec->r_lowest == 0 && rank == (spec->n_radii - 1) % world && !r0_direct()) {
        TRY(final_prepare(&c));
        g_spectra.stars_r0_ready = 1;
    }
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        double means[C21CM_MAX_RADII];
        TRY(c21hip_d2h(means, c.scalars + SC_MEANS, sizeof(means), stream));
        TRY(c21hip_sync(stream));
        for (int r = 0; r < spec->n_radii; r++) report->f_coll_grid_mean[r] = means[r];
        /* a single-process run of both phases (world = 1) keeps its own means */
        if (world == 1) c21cm_ionize_shard_set_means(means, spec->n_radii);
        report->ms_preloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_rloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
        report->ms_postloop = 0.;
    }
done:
    c21hip_wev_release();
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

int c21cm_ionize_shard_finish(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                              const PerturbedField *perturbed_field,
                              const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                              const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                              void *stream) {
    int status = validate_spec(spec, perturbed_field, halos, spin_temp, box);
    if (!status && spec->use_mini_halos) status = no_mini_shards();
    if (status) return status;
    if (!first_cross || !c21hip_is_device_ptr(first_cross)) {
        c21hip_set_error("ionize shard: first_cross must be a device array");
        return C21CM_VALUE_ERROR;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    /* When this process ran c21cm_ionize_shard_radii on the same input arrays just before,
     * its unfiltered spectra are still in the workspace and are reused; otherwise the
     * pre-loop transforms are redone. */
    TRY(ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, box, 1,
                  stream));
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    if (g_shard_means.valid && g_shard_means.n == spec->n_radii) /* ctx_setup zeroed the block */
        TRY(c21hip_h2d(c.scalars + SC_MEANS, g_shard_means.means,
                       sizeof(double) * (size_t)spec->n_radii, stream));
    g_shard_means.valid = 0;
    TRY(init_output_grids(&c, previous_ionize_box));
    int stars_ready = 0;
    if (spec->r_lowest == 0) {
        if (c.fused && r0_direct() && !c.sphere)
            stars_ready = 0; /* the final sweep reads the emissivity input itself */
        else if (spectra_match(&c, perturbed_field, halos, spin_temp))
            stars_ready = g_spectra.stars_r0_ready;
        else
            TRY(preloop(&c));
        g_spectra.valid = 0;
    }
    if (c.fused && spec->r_lowest == 0 && !c.sphere) {
        TRY(final_step(&c, first_cross, stars_ready));
    } else if (eul_r0_fused(&c)) { /* as the single pass: one sweep for mask + cell-scale radius + post-loop */
        c.r0_mask = first_cross;
        TRY(one_radius(&c, 0, NULL, -1));
    } else {
        TRY(c21hip_apply_first_cross(first_cross, c.prev_zre, spec->first_snapshot,
                                     spec->redshift, c.xH, c.zre, c.ntot, stream));
        if (c.sphere) TRY(paint_spheres(&c, first_cross));
        if (spec->r_lowest == 0) TRY(one_radius(&c, 0, NULL, -1));
    }
    TRY(c21hip_event_record(ev[1], stream));
    TRY(postloop(&c, box, report));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        report->ms_preloop = 0.;
        report->ms_rloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_postloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
    }
done:
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

/* ---- the finish phase split by cell slabs (round 5) -------------------------------------------
 * The cell-scale radius and the post-loop are per cell (IonisationBox.c:1031-1256), and every rank holds
 * the replicated inputs: rank r sweeps the cells of chunks [n r / W, n (r + 1) / W) of the final sweep
 * -- its slab -- from the combined first crossings of THAT slab alone (so the exchange before it moves
 * 1 / W of each rank's packed grid per link instead of whole grids onto one rank), leaves the chunks'
 * partial sums where the single pass leaves them, the ranks all-gather those (2 x n doubles) and every
 * rank reduces all of them in the single pass' fixed order: sum(x_HI), the f_coll mean of index 0 and
 * with them global_xH / mean_f_coll are the single pass' to the last bit on every rank, with no scalar
 * broadcast.  The outputs stay slab-resident unless the exchange callback gathers them.
 * Supported: the fused Lagrangian loop down to index 0 with the direct cell-scale sweep (no recombination
 * model, no IONISE_ENTIRE_SPHERE, no mini-halos); everything else finishes on the owner rank. */
int c21cm_ionize_shard_slab_supported(const c21cm_ionize_spec *s) {
    if (!s) return 0;
    const int nz = s->hii_dim_z;
    if (s->recomb_model != C21CM_RECOMB_NONE || s->use_mini_halos || s->ionise_entire_sphere ||
        s->r_lowest != 0 || !c21hip_fft_is_native(s->hii_dim, s->hii_dim, nz))
        return 0;
    if (s->fcoll_mode == C21CM_FCOLL_STARS_GRID)
        return r0_direct() && (!s->use_ts_fluct || c21hip_z_ionise_xe_supported(s->hii_dim, s->hii_dim, nz));
    /* Eulerian models: the cell-scale radius' f_coll grid and its box mean are computed by every rank
     * (replicated sweeps of one grid), the ONE sweep that applies mask + barrier + post-loop
     * (eul_r0_fused) runs on the rank's slab */
    {
        const char *e = getenv("C21CM_EUL_R0_FUSED");
        return !(e && e[0] == '0') && (!s->use_ts_fluct || r0_direct());
    }
}

/* chunks and cells of `rank`'s slab (whole chunks of the final sweep, dealt evenly; cell bounds are
 * multiples of 512 cells except the box end) */
int c21cm_ionize_shard_slab(const c21cm_ionize_spec *spec, int rank, int world, int *chunk_begin,
                            int *chunk_end, size_t *cell_begin, size_t *cell_end, int *n_chunks_out,
                            size_t *chunk_cells_out) {
    if (!spec || world < 1 || rank < 0 || rank >= world) return C21CM_VALUE_ERROR;
    c21hip_ionize_args args;
    fill_args(&args, spec, 0);
    int n_chunks = 0;
    size_t chunk_cells = 0;
    c21hip_final_sweep_chunks(&args, 1, &n_chunks, &chunk_cells);
    const size_t ntot = (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z;
    const int b = (int)((long)n_chunks * rank / world), e = (int)((long)n_chunks * (rank + 1) / world);
    size_t cb = (size_t)b * chunk_cells, ce = (size_t)e * chunk_cells;
    if (cb > ntot) cb = ntot;
    if (ce > ntot || e == n_chunks) ce = ntot;
    if (chunk_begin) *chunk_begin = b;
    if (chunk_end) *chunk_end = e;
    if (cell_begin) *cell_begin = cb;
    if (cell_end) *cell_end = ce;
    if (n_chunks_out) *n_chunks_out = n_chunks;
    if (chunk_cells_out) *chunk_cells_out = chunk_cells;
    return 0;
}

int c21cm_ionize_shard_finish_slab(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                                   int rank, int world, const PerturbedField *perturbed_field,
                                   const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                   const HaloBox *halos, IonizedBox *box, c21cm_ionize_report *report,
                                   c21cm_shard_slab_exchange_fn exchange, void *exchange_user,
                                   int outputs_gathered, void *stream) {
    /* `exchange` is entered exactly once whatever happens locally (with the local status), so that a
     * rank which fails here still joins the agreement its peers wait in */
    c21cm_shard_slab_state st;
    memset(&st, 0, sizeof(st));
    st.rank = rank;
    st.world = world;
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    int entered = 0, wev = 0;
    int status = validate_spec(spec, perturbed_field, halos, spin_temp, box);
    if (!status && !c21cm_ionize_shard_slab_supported(spec)) {
        c21hip_set_error("ionize shard: this model does not finish by slabs (c21cm_ionize_shard_slab_supported)");
        status = C21CM_VALUE_ERROR;
    }
    if (!status && (!first_cross || !c21hip_is_device_ptr(first_cross) || world < 1 || rank < 0 ||
                    rank >= world)) {
        c21hip_set_error("ionize shard: first_cross must be a device array, 0 <= rank < world");
        status = C21CM_VALUE_ERROR;
    }
    if (status) goto done;
    TRY(c21cm_ionize_shard_slab(spec, rank, world, &st.chunk_begin, &st.chunk_end, &st.cell_begin,
                                &st.cell_end, &st.n_chunks, &st.chunk_cells));
    TRY(ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, box, 1, stream));
    const int eul = !c.lagrangian;
    if ((!eul && (!c.fused || c.sphere)) || (eul && !eul_r0_fused(&c))) { /* (slab_supported mirrors ctx_setup) */
        c21hip_set_error("ionize shard: the slab finish needs the fused Lagrangian loop or the one-sweep "
                         "cell-scale radius of the Eulerian loops");
        status = C21CM_VALUE_ERROR;
        goto done;
    }
    st.ntot = c.ntot;
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    c.r0_slab = eul;
    TRY(init_output_grids(&c, previous_ionize_box));
    g_shard_means.valid = 0;
    if (eul) {
        /* the cell-scale radius' f_coll grid on every rank (its spectra: left by this process' shard phase,
         * else recomputed), then the rank's slab of the one sweep */
        if (!spectra_match(&c, perturbed_field, halos, spin_temp)) TRY(preloop(&c));
        g_spectra.valid = 0;
        wev = 1; /* (released under done: a failing radius must not leak the node tables -- ADVICE r5) */
        TRY(native_wev_prepare(&c, spec->n_radii - 1, 1, stream));
        c.r0_mask = first_cross;
        c.r0_slab = 1;
        c.r0_cb = st.chunk_begin;
        c.r0_ce = st.chunk_end;
        TRY(one_radius(&c, 0, NULL, -1));
        c21hip_wev_release();
        wev = 0;
    } else {
        g_spectra.valid = 0;
        TRY(final_step_range(&c, first_cross, 0, st.chunk_begin, st.chunk_end));
    }
    st.partials_stars = c.partials;
    st.partials_xh = c.partials + C21HIP_PARTIALS / 2;
    st.flag = (int *)(c.scalars + SC_FLAG);
    st.out[0] = c.xH;
    st.out[1] = c.zre;
    st.out[2] = c.Tk;
    entered = 1;
    if (exchange) TRY(exchange(exchange_user, &st, 0, stream));
    if (eul) {
        c21hip_ionize_args args0;
        fill_args(&args0, spec, 0);
        TRY(c21hip_final_sweep_reduce(&args0, 1, c.partials, NULL, c.scalars + SC_XHSUM, stream));
    } else {
        TRY(final_step_sums(&c));
    }
    TRY(c21hip_event_record(ev[1], stream));
    if (!outputs_gathered) { /* slab-resident outputs: the host copies of staged grids cover the slab */
        c.cb_cell0 = st.cell_begin;
        c.cb_ncell = st.cell_end - st.cell_begin;
        if (!c.cb_ncell) c.cb.n = 0;
    }
    TRY(postloop(&c, box, report));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        report->ms_preloop = 0.;
        report->ms_rloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_postloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
    }
done:
    if (wev) c21hip_wev_release();
    if (!entered && exchange) {
        const int st2 = exchange(exchange_user, &st, status ? status : C21CM_VALUE_ERROR, stream);
        if (!status) status = st2;
    }
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

/* ---- R-loop sharding with a recombination model ----------------------------------------------
 * The first crossing of a cell carries two floats (the radius = mean free path, and Gamma_12), so
 * the uint8 index grid is replaced by 64-bit keys bits(mfp) << 32 | bits(G12): the max over ranks
 * is the crossing at the largest radius together with its own Gamma_12 ("float
 * G12 candidate for that index"), in ONE reduce of 8 N bytes.  Each rank runs the unfused
 * recombination sequence for its radii on rank-local x_HI / Gamma_12 / mean-free-path scratch
 * (the barrier of a radius depends on no other radius: rec comes from the PREVIOUS snapshot). */
int c21cm_ionize_shard_radii_keys(const c21cm_ionize_spec *spec, int rank, int world,
                                  const PerturbedField *perturbed_field,
                                  const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                  const HaloBox *halos, unsigned long long *cross_keys,
                                  c21cm_ionize_report *report, void *stream) {
    IonizedBox dummy;
    memset(&dummy, 0, sizeof(dummy));
    int status = 0;
    if (!cross_keys || !c21hip_is_device_ptr(cross_keys) || world < 1 || rank < 0 || rank >= world) {
        c21hip_set_error("ionize shard: cross_keys must be a device array, 0 <= rank < world");
        return C21CM_VALUE_ERROR;
    }
    if (!spec || spec->recomb_model == C21CM_RECOMB_NONE) {
        c21hip_set_error("ionize shard: the key phases are for recombination models");
        return C21CM_VALUE_ERROR;
    }
    const size_t ntot = (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z;
    if (spec->fcoll_mode != C21CM_FCOLL_STARS_GRID)
        dummy.unnormalised_nion = (float *)c21hip_ws(WS_NION_DENSE, ntot * sizeof(float));
    {
        IonizedBox probe = dummy;
        float sentinel;
        probe.neutral_fraction = probe.z_reion = probe.kinetic_temperature = &sentinel;
        probe.ionisation_rate_G12 = probe.cumulative_recombinations = &sentinel;
        status = validate_spec(spec, perturbed_field, halos, spin_temp, &probe);
        if (!status && spec->use_mini_halos) status = no_mini_shards();
        if (status) return status;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    TRY(ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, &dummy, 0,
                  stream));
    c.xH = (float *)c21hip_ws(WS_SH_XH, ntot * sizeof(float));
    c.zre = (float *)c21hip_ws(WS_SH_ZRE, ntot * sizeof(float));
    c.G12 = (float *)c21hip_ws(WS_SH_G12, ntot * sizeof(float));
    c.mfp = (float *)c21hip_ws(WS_SH_MFP, ntot * sizeof(float));
    if (!c.xH || !c.zre || !c.G12 || !c.mfp) return C21CM_MEMORY_ALLOC_ERROR;
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    TRY(c21hip_fill(c.xH, ntot, 1.0f, stream)); /* a fresh IonizedBox (outputs.py:1524-1527) */
    TRY(c21hip_memset(c.G12, 0, ntot * sizeof(float), stream));
    TRY(c21hip_memset(c.mfp, 0, ntot * sizeof(float), stream));
    g_spectra.valid = 0;
    TRY(preloop(&c));
    spectra_remember(&c, perturbed_field, halos, spin_temp);
    TRY(c21hip_event_record(ev[1], stream));
    TRY(native_wev_prepare(&c, spec->n_radii - 1, 1, stream)); /* the single pass' node tables: the whole ladder */
    for (int R_ct = spec->n_radii - 1 - rank; R_ct >= 1; R_ct -= world) {
        if (R_ct < spec->r_lowest) break;
        TRY(one_radius(&c, R_ct, NULL, R_ct - world));
    }
    TRY(c21hip_pack_cross_keys(c.mfp, c.G12, cross_keys, ntot, stream));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        double means[C21CM_MAX_RADII];
        TRY(c21hip_d2h(means, c.scalars + SC_MEANS, sizeof(means), stream));
        TRY(c21hip_sync(stream));
        for (int r = 0; r < spec->n_radii; r++) report->f_coll_grid_mean[r] = means[r];
        if (world == 1) c21cm_ionize_shard_set_means(means, spec->n_radii);
        report->ms_preloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_rloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
        report->ms_postloop = 0.;
    }
done:
    c21hip_wev_release();
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

int c21cm_ionize_shard_finish_keys(const c21cm_ionize_spec *spec,
                                   const unsigned long long *cross_keys,
                                   const PerturbedField *perturbed_field,
                                   const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                   const HaloBox *halos, IonizedBox *box,
                                   c21cm_ionize_report *report, void *stream) {
    int status = validate_spec(spec, perturbed_field, halos, spin_temp, box);
    if (!status && spec->use_mini_halos) status = no_mini_shards();
    if (status) return status;
    if (!cross_keys || !c21hip_is_device_ptr(cross_keys) ||
        spec->recomb_model == C21CM_RECOMB_NONE) {
        c21hip_set_error("ionize shard: cross_keys must be a device array (recombination models)");
        return C21CM_VALUE_ERROR;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    TRY(ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, box, 1,
                  stream));
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    if (g_shard_means.valid && g_shard_means.n == spec->n_radii)
        TRY(c21hip_h2d(c.scalars + SC_MEANS, g_shard_means.means,
                       sizeof(double) * (size_t)spec->n_radii, stream));
    g_shard_means.valid = 0;
    TRY(init_output_grids(&c, previous_ionize_box));
    TRY(c21hip_apply_cross_keys(cross_keys, c.prev_zre, spec->first_snapshot, spec->redshift, c.xH,
                                c.zre, c.G12, c.mfp, c.ntot, stream));
    if (spec->r_lowest == 0) {
        if (!spectra_match(&c, perturbed_field, halos, spin_temp)) TRY(preloop(&c));
        g_spectra.valid = 0;
        TRY(one_radius(&c, 0, NULL, -1));
    }
    TRY(c21hip_event_record(ev[1], stream));
    TRY(postloop(&c, box, report));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        report->ms_preloop = 0.;
        report->ms_rloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_postloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
    }
done:
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

/* ---- R-loop sharding of the FUSED recombination loop (round 3) ---------------------------------
 * CELL_RECOMB runs without an x_e grid ride the fused loop (ctx_setup: fused_rc), whose state after
 * the radii is the uint8 first-crossing index plus Gamma_12 at the crossing -- the mean free path
 * IS the radius of that index.  A rank's radii therefore leave 5 bytes per cell instead of the
 * 8-byte key, the winner of a cell is the rank with the larger index (indices > 0 are owned by
 * one rank each) together with ITS Gamma_12 (c21hip_combine_cross_g12), and the finish phase is
 * the tail of the single pass: first crossings -> x_HI / z_reion / mean free path, the cell-scale
 * radius, the post-loop.  c21cm_ionize_shard_rc_supported tells whether a spec takes this route;
 * everything else with a recombination model keeps the 64-bit keys above.
 * reference: src/py21cmfast/src/IonisationBox.c:1084-1140,1531-1588 */
int c21cm_ionize_shard_rc_supported(const c21cm_ionize_spec *s) {
    if (!s || s->recomb_model == C21CM_RECOMB_NONE ||
        s->use_mini_halos || s->ionise_entire_sphere || s->r_lowest != 0 ||
        s->fcoll_mode != C21CM_FCOLL_STARS_GRID)
        return 0;
    const char *e = getenv("C21CM_RECOMB_FUSED");
    if (e && e[0] == '0') return 0;
    const int nx = s->hii_dim, ny = s->hii_dim, nz = s->hii_dim_z;
    if (s->use_ts_fluct) { /* round 5: with the x_e grid of a spin-temperature run too (the same conditions
                            * as the single pass: ctx_setup) */
        const char *et = getenv("C21CM_RECOMB_FUSED_TS");
        if ((et && et[0] == '0') || !r0_direct() || !c21hip_z_ionise_recomb_xe_supported(nx, ny, nz)) return 0;
    }
    if (!s->cell_recomb) { /* round 5: the filtered N_rec of the previous snapshot is an input every rank holds */
        const char *en = getenv("C21CM_RECOMB_FUSED_NREC");
        if ((en && en[0] == '0') || s->recomb_model != C21CM_RECOMB_INHOMOGENEOUS) return 0;
        if (!(s->use_ts_fluct ? c21hip_z_ionise_recomb_xe_nrec_supported(nx, ny, nz)
                              : c21hip_z_ionise_recomb_xe_supported(nx, ny, nz)))
            return 0;
    }
    return c21hip_fft_is_native(nx, ny, nz) && c21hip_z_ionise_recomb_supported(nx, ny, nz) &&
           c21hip_wev_applicable(s->hii_filter, s->stars_filter, 2, nx, ny, nz);
}

int c21cm_ionize_shard_radii_rc(const c21cm_ionize_spec *spec, int rank, int world,
                                const PerturbedField *perturbed_field,
                                const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                const HaloBox *halos, unsigned char *first_cross, float *cross_g12,
                                c21cm_ionize_report *report, void *stream) {
    IonizedBox dummy;
    memset(&dummy, 0, sizeof(dummy));
    int status = 0;
    if (!first_cross || !cross_g12 || !c21hip_is_device_ptr(first_cross) ||
        !c21hip_is_device_ptr(cross_g12) || world < 1 || rank < 0 || rank >= world) {
        c21hip_set_error("ionize shard: first_cross / cross_g12 must be device arrays, 0 <= rank < world");
        return C21CM_VALUE_ERROR;
    }
    if (!c21cm_ionize_shard_rc_supported(spec)) {
        c21hip_set_error("ionize shard: this spec does not take the fused recombination loop "
                         "(c21cm_ionize_shard_rc_supported); use the 64-bit key phases");
        return C21CM_VALUE_ERROR;
    }
    {
        IonizedBox probe = dummy;
        float sentinel;
        probe.neutral_fraction = probe.z_reion = probe.kinetic_temperature = &sentinel;
        probe.ionisation_rate_G12 = probe.cumulative_recombinations = &sentinel;
        status = validate_spec(spec, perturbed_field, halos, spin_temp, &probe);
        if (status) return status;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    g_rc_phase = 1;
    status = ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, &dummy, 0,
                       stream);
    g_rc_phase = 0;
    if (status) return status;
    if (!c.fused_rc) {
        c21hip_set_error("ionize shard: the fused recombination loop is not available for this box");
        return C21CM_VALUE_ERROR;
    }
    c.G12 = cross_g12;
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    TRY(c21hip_memset(first_cross, 0, c.ntot, stream));
    TRY(c21hip_memset(cross_g12, 0, c.ntot * sizeof(float), stream));
    g_spectra.valid = 0;
    TRY(preloop(&c));
    spectra_remember(&c, perturbed_field, halos, spin_temp);
    TRY(c21hip_event_record(ev[1], stream));
    TRY(fused_loop(&c, spec->n_radii - 1 - rank, world, 1, first_cross));
    TRY(flush_deferred(&c));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        double means[C21CM_MAX_RADII];
        TRY(c21hip_d2h(means, c.scalars + SC_MEANS, sizeof(means), stream));
        TRY(c21hip_sync(stream));
        for (int r = 0; r < spec->n_radii; r++) report->f_coll_grid_mean[r] = means[r];
        if (world == 1) c21cm_ionize_shard_set_means(means, spec->n_radii);
        report->ms_preloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_rloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
        report->ms_postloop = 0.;
    }
done:
    c21hip_wev_release();
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

int c21cm_ionize_shard_finish_rc(const c21cm_ionize_spec *spec, const unsigned char *first_cross,
                                 const float *cross_g12, const PerturbedField *perturbed_field,
                                 const IonizedBox *previous_ionize_box, const TsBox *spin_temp,
                                 const HaloBox *halos, IonizedBox *box,
                                 c21cm_ionize_report *report, void *stream) {
    int status = validate_spec(spec, perturbed_field, halos, spin_temp, box);
    if (status) return status;
    if (!first_cross || !cross_g12 || !c21hip_is_device_ptr(first_cross) ||
        !c21hip_is_device_ptr(cross_g12) || !c21cm_ionize_shard_rc_supported(spec)) {
        c21hip_set_error("ionize shard: first_cross / cross_g12 must be device arrays of a spec that "
                         "takes the fused recombination loop");
        return C21CM_VALUE_ERROR;
    }
    ion_ctx c;
    void *ev[3] = {NULL, NULL, NULL};
    g_rc_phase = 1;
    status = ctx_setup(&c, spec, perturbed_field, previous_ionize_box, spin_temp, halos, box, 1, stream);
    g_rc_phase = 0;
    if (status) return status;
    for (int i = 0; i < 3; i++) ev[i] = c21hip_event_create();
    TRY(c21hip_event_record(ev[0], stream));
    if (g_shard_means.valid && g_shard_means.n == spec->n_radii)
        TRY(c21hip_h2d(c.scalars + SC_MEANS, g_shard_means.means,
                       sizeof(double) * (size_t)spec->n_radii, stream));
    g_shard_means.valid = 0;
    TRY(init_output_grids(&c, previous_ionize_box));
    TRY(c21hip_d2d(c.G12, cross_g12, c.ntot * sizeof(float), stream));
    {
        float Rf[C21CM_MAX_RADII];
        for (int r = 0; r < C21CM_MAX_RADII; r++) Rf[r] = r < spec->n_radii ? (float)spec->R[r] : 0.f;
        float *R_dev = (float *)c21hip_ws(WS_R_DEV, sizeof(Rf));
        if (!R_dev) {
            status = C21CM_MEMORY_ALLOC_ERROR;
            goto done;
        }
        TRY(c21hip_h2d(R_dev, Rf, sizeof(Rf), stream));
        TRY(c21hip_sync(stream)); /* `Rf` is a stack buffer */
        TRY(c21hip_apply_first_cross_recomb(first_cross, R_dev, c.prev_zre, spec->first_snapshot,
                                            spec->redshift, c.xH, c.zre, c.mfp, c.ntot, stream));
    }
    if (!spectra_match(&c, perturbed_field, halos, spin_temp)) TRY(preloop(&c));
    g_spectra.valid = 0;
    TRY(one_radius(&c, 0, NULL, -1));
    TRY(c21hip_event_record(ev[1], stream));
    TRY(postloop(&c, box, report));
    TRY(c21hip_event_record(ev[2], stream));
    if (report) {
        report->ms_preloop = 0.;
        report->ms_rloop = c21hip_event_elapsed_ms(ev[0], ev[1]);
        report->ms_postloop = c21hip_event_elapsed_ms(ev[1], ev[2]);
    }
done:
    for (int i = 0; i < 3; i++) c21hip_event_destroy(ev[i]);
    return status;
}

/* The early exit of ComputeIonizedBox when the expected HII fraction is negligible.
 * reference: src/py21cmfast/src/IonisationBox.c:531-565,1472-1475 */
int c21cm_neutral_box(const c21cm_ionize_spec *spec, const PerturbedField *perturbed_field,
                      const TsBox *spin_temp, IonizedBox *box, size_t ntot) {
    extern double c21_xion_RECFAST(float z);
    int status = 0;
    const size_t bytes = ntot * sizeof(float);
    copyback_list cb;
    memset(&cb, 0, sizeof(cb));
    void *stream = NULL;
    const int ts = spec->use_ts_fluct;
    const float *density = NULL, *xe = NULL, *Tn = NULL;
    /* the same required arrays as the full path (validate_spec), checked before any launch */
    if (!box || !box->neutral_fraction || !box->z_reion ||
        (!spec->minimize_memory && !box->kinetic_temperature)) {
        c21hip_set_error("ionize (neutral box): neutral_fraction / z_reion%s are required",
                         spec->minimize_memory ? "" : " / kinetic_temperature");
        return C21CM_VALUE_ERROR;
    }
    if (ts && (!spin_temp || !spin_temp->xray_ionised_fraction ||
               (!spec->minimize_memory && !spin_temp->kinetic_temp_neutral))) {
        c21hip_set_error("ionize (neutral box): USE_TS_FLUCT needs the TsBox arrays");
        return C21CM_VALUE_ERROR;
    }
    if (!ts && !spec->minimize_memory && (!perturbed_field || !perturbed_field->density)) {
        c21hip_set_error("ionize (neutral box): PerturbedField.density is required");
        return C21CM_VALUE_ERROR;
    }
    if (ts) {
        xe = stage_in(WS_XE_DENSE, spin_temp->xray_ionised_fraction, bytes, stream, &status);
        if (!spec->minimize_memory)
            Tn = stage_in(WS_TNEUTRAL, spin_temp->kinetic_temp_neutral, bytes, stream, &status);
    } else if (!spec->minimize_memory) {
        density = stage_in(WS_DENSITY, perturbed_field->density, bytes, stream, &status);
    }
    float *xH = stage_inout(WS_XH, box->neutral_fraction, bytes, 0, &cb, stream, &status);
    float *zre = stage_inout(WS_ZRE, box->z_reion, bytes, 0, &cb, stream, &status);
    float *Tk = spec->minimize_memory
                    ? NULL
                    : stage_inout(WS_TK, box->kinetic_temperature, bytes, 0, &cb, stream, &status);
    if (status) return status;
    const double global_xH = ts ? 0. : 1. - c21_xion_RECFAST((float)spec->redshift);
    TRY(c21hip_fill(zre, ntot, -1.0f, stream)); /* IonisationBox.c:1372-1378 */
    TRY(c21hip_neutral_box(density, xe, Tn, xH, Tk, ntot, ts, global_xH, spec->TK_nofluct,
                           spec->adia_TK_term, stream));
    for (int i = 0; i < cb.n; i++) TRY(c21hip_d2h(cb.host[i], cb.dev[i], cb.bytes[i], stream));
    TRY(c21hip_sync(stream));
done:
    return status;
}

/* calculate_mcrit_boxes (IonisationBox.c:403-457): the two log10 turnover-mass grids of a
 * USE_MINI_HALOS run and their box averages */
int c21cm_mturn_grids(const c21cm_mturn_spec *spec, const float *prev_G12,
                      const float *prev_z_reion, const float *J_21_LW, const float *vcb,
                      float *log10_mturn_acg, float *log10_mturn_mcg, double *ave_acg,
                      double *ave_mcg, void *stream) {
    enum { WS_MT_G12 = 197, WS_MT_ZRE, WS_MT_J21, WS_MT_VCB, WS_MT_OUT_A, WS_MT_OUT_M, WS_MT_SC,
           WS_MT_PART };
    int status = 0;
    if (!spec || !prev_G12 || !J_21_LW || !log10_mturn_acg || !log10_mturn_mcg ||
        (!spec->first_snapshot && !prev_z_reion) || spec->hii_dim < 1 || spec->hii_dim_z < 1) {
        c21hip_set_error("mturn_grids: previous Gamma_12 / z_reion, J_21_LW and the two output "
                         "grids are required");
        return C21CM_VALUE_ERROR;
    }
    const size_t ntot = (size_t)spec->hii_dim * spec->hii_dim * spec->hii_dim_z;
    const size_t dbytes = ntot * sizeof(float);
    copyback_list cb;
    cb.n = 0;
    const float *g12 = stage_in(WS_MT_G12, prev_G12, dbytes, stream, &status);
    const float *zre = spec->first_snapshot
                           ? NULL
                           : stage_in(WS_MT_ZRE, prev_z_reion, dbytes, stream, &status);
    const float *j21 = stage_in(WS_MT_J21, J_21_LW, dbytes, stream, &status);
    const float *v = vcb ? stage_in(WS_MT_VCB, vcb, dbytes, stream, &status) : NULL;
    float *out_a = stage_inout(WS_MT_OUT_A, log10_mturn_acg, dbytes, 0, &cb, stream, &status);
    float *out_m = stage_inout(WS_MT_OUT_M, log10_mturn_mcg, dbytes, 0, &cb, stream, &status);
    double *sc = (double *)c21hip_ws(WS_MT_SC, 4 * sizeof(double));
    double *partials = (double *)c21hip_ws(WS_MT_PART, 2 * C21HIP_PARTIALS * sizeof(double));
    if (status) return status;
    if (!sc || !partials) return C21CM_MEMORY_ALLOC_ERROR;
    double host_sc[3];
    TRY(c21hip_memset(sc, 0, 4 * sizeof(double), stream));
    TRY(c21hip_mturn_grids(ntot, spec->first_snapshot, spec->redshift, spec->mturn_a_nofb,
                           spec->mturn_m_nofb, spec->vcb_const, spec->A_LW, spec->BETA_LW,
                           spec->A_VCB, spec->BETA_VCB, spec->sigma_vcb, g12, zre, j21, v, out_a,
                           out_m, partials, sc, (int *)(sc + 2), stream));
    TRY(c21hip_d2h(host_sc, sc, sizeof(host_sc), stream));
    for (int i = 0; i < cb.n; i++) TRY(c21hip_d2h(cb.host[i], cb.dev[i], cb.bytes[i], stream));
    TRY(c21hip_sync(stream));
    {
        int flag;
        memcpy(&flag, &host_sc[2], sizeof(int));
        if (flag) { /* :425-429 */
            c21hip_set_error("mturn_grids: Lyman-Werner threshold is NaN or zero (J_21_LW, v_cb)");
            status = C21CM_VALUE_ERROR;
            goto done;
        }
    }
    if (ave_acg) *ave_acg = host_sc[0] / ntot;
    if (ave_mcg) *ave_mcg = host_sc[1] / ntot;
done:
    return status;
}
