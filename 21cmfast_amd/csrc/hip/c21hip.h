/*
 * c21hip.h -- the thin C layer between the C host drivers (csrc/host/ *.c) and the
 * hand-written CDNA4 kernels (csrc/hip/ *.hip).  Plain pointers and sizes only.
 * Every function enqueues work on `stream` (a hipStream_t carried as void*) and
 * returns 0 or a c21cm_status code; nothing here synchronises unless it says so.
 *
 * Grid layouts (reference: src/py21cmfast/src/indexing.h:84-100):
 *   dense  float[nx][ny][nz]
 *   padded float[nx][ny][2*(nz/2+1)]  == complex float2[nx][ny][nz/2+1]
 */
#ifndef C21HIP_H
#define C21HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime.hip : memory, residency, workspace, timing ---- */
int c21hip_device_count(void);
int c21hip_current_device(void);     /* -1 when there is none */
int c21hip_use_device(int device);   /* per-thread: for helper threads of the host drivers */
int c21hip_is_device_ptr(const void *p);     /* 1 = MI355X HBM, 0 = host          */
void *c21hip_ws(int slot, size_t bytes);     /* cached device scratch, NULL = OOM */
/* placement shopping (ionize_driver.c: place_work_partner): what a slot holds, a buffer handed to a slot, plain
 * allocations outside the workspace, free device memory; the timed two-grid pass Y that tells whether two work
 * spectra sit well together (fft_native.hip) */
void *c21hip_ws_peek(int slot, size_t *bytes);
int c21hip_ws_adopt(int slot, void *ptr, size_t bytes);
void *c21hip_raw_alloc(size_t bytes);
/* physical memory with only its head mapped (placement walk); the workspace adopts a fully mapped one */
void *c21hip_vmm_chunk(size_t phys_bytes, size_t map_bytes, void **va_out);
void c21hip_vmm_chunk_free(void *chunk);
int c21hip_ws_adopt_vmm(int slot, void *chunk, size_t bytes);
void c21hip_raw_free(void *p);
size_t c21hip_free_bytes(void);
size_t c21hip_total_bytes(void);
/* processes holding memory on the current device, this one included (KFD sysfs); -1: unknown */
int c21hip_device_tenants(void);
int c21hip_probe_pass_y2(float *work_a, float *work_b, int nx, int ny, int nz, int reps, float *ms, void *stream);
/* host/placement.c: the workspace slot `slot_new` (bytes) filled with a buffer that sits well with the one in
 * `slot_partner` for two-grid launches; plain c21hip_ws where the walk does not apply */
float *c21_place_work_partner(int slot_partner, int slot_new, size_t bytes, int nx, int ny, int nz, void *stream);
void c21hip_ws_release(void);
unsigned long c21hip_ws_generation(void); /* counts c21hip_ws_release calls */
int c21hip_h2d(void *dst, const void *src, size_t bytes, void *stream);
int c21hip_d2h(void *dst, const void *src, size_t bytes, void *stream);
int c21hip_d2d(void *dst, const void *src, size_t bytes, void *stream);
int c21hip_memset(void *dst, int byte, size_t bytes, void *stream);
int c21hip_sync(void *stream);
int c21hip_device_sync(void);
void *c21hip_event_create(void);
void c21hip_event_destroy(void *ev);
int c21hip_event_record(void *ev, void *stream);
int c21hip_stream_wait_event(void *stream, void *ev);
int c21hip_event_synchronize(void *ev); /* host waits for the event */
void *c21hip_pinned_host(size_t bytes);  /* library-owned pinned staging buffer (one, reused) */
void *c21hip_pinned_alloc(size_t bytes); /* caller-owned pinned blocks (c21hip_pinned_free) */
void c21hip_pinned_free(void *p);
void *c21hip_stream_create(void);        /* caller-owned non-blocking stream */
void c21hip_stream_destroy(void *s);
void *c21hip_aux_stream(void); /* library-owned non-blocking side stream, NULL on failure */
float c21hip_event_elapsed_ms(void *start, void *stop); /* synchronises on stop */
void c21hip_set_error(const char *fmt, ...);
const char *c21hip_get_error(void);

/* ---- fft.hip : in-place padded real 3-D transforms (reference: dft.c:18-72) ---- */
int c21hip_fft_r2c(float *padded, int nx, int ny, int nz, void *stream);
int c21hip_fft_c2r(float *padded, int nx, int ny, int nz, void *stream);
void c21hip_fft_release(void);
/* 1 when the hand-written power-of-two transform is used, 0 for rocFFT */
int c21hip_fft_is_native(int nx, int ny, int nz);
int c21hip_pair_sweep_supported(int nx); /* two line tiles of nx points fit in LDS */

/* ---- fft_native.hip : power-of-two transform on the split k-space layout ----
 * split layout: complex main[nx][ny][nz/2] followed by the Nyquist plane nyq[nx][ny]. */
size_t c21hip_split_floats(int nx, int ny, int nz);
int c21hip_padded_to_split(const float *padded_c, float *split, int nx, int ny, int nz,
                           void *stream);
/* [W(kR) x] inverse transform: split_src -> (split_work) -> real rows of out_zstride floats.
 * Fuses memcpy + filter_box + dft_c2r_cube (IonisationBox.c:577-663). */
int c21hip_split_filter_c2r(const float *split_src, float *split_work, float *real_out,
                            long out_zstride, int nx, int ny, int nz, double box_len,
                            double box_len_z, int filter_type, float R, float R_param, int apply,
                            void *stream);

/* forward transform of real rows into the split layout, scale-and-clip fused into the load
 * (lo > hi disables the clip), result times out_scale: IonisationBox.c:323-360 in 3 sweeps */
int c21hip_split_r2c(const float *real_in, long in_zstride, float *split_out, int nx, int ny,
                     int nz, double factor, double lo, double hi, float out_scale, void *stream);
int c21hip_split_filter_xy(const float *split_src, float *split_work, int nx, int ny, int nz,
                           double box_len, double box_len_z, int filter_type, float R,
                           float R_param, int apply, void *stream);
/* two grids (density, emissivity) of one radius in one sweep, each with its own window;
 * tables_ready: the W(kR) tables were prebuilt into buffer table_slot (0/1) */
int c21hip_split_filter_xy2(const float *src_a, float *work_a, int filter_a, float R_param_a,
                            const float *src_b, float *work_b, int filter_b, float R_param_b,
                            int nx, int ny, int nz, double box_len, double box_len_z, float R,
                            int apply, int table_slot, int tables_ready, void *stream);
/* one or two grids of one shell (windows 4 = spherical shell / 5 = multiple scattering between
 * R_inner and R_outer; SpinTemperatureBox.c:698-700) */
int c21hip_split_filter_shell(const float *src_a, float *work_a, int filter_a, const float *src_b,
                              float *work_b, int filter_b, int n_grids, int nx, int ny, int nz,
                              double box_len, double box_len_z, float R_inner, float R_outer,
                              float R_star, int apply, void *stream);
/* the same for two radii out of one pass-X sweep (nx < 1024); table slots 0..3;
 * phases: 1 build the tables, 2 pass X, 4 / 8 pass Y of the first / second radius */
int c21hip_split_filter_xy2_pair(const float *src_a, float *work_a, float *work_a2, int filter_a,
                                 float R_param_a, const float *src_b, float *work_b,
                                 float *work_b2, int filter_b, float R_param_b, int nx, int ny,
                                 int nz, double box_len, double box_len_z, float R, float R2,
                                 int table_slot, int table_slot2, int phases, void *stream);
/* one grid, two radii, windows evaluated in the kernel (the Eulerian loops' filtered density; phases 2 / 4 / 8 as
 * above, calls without 2 run the pass Y alone); C21CM_VALUE_ERROR when the evaluated set does not cover it */
int c21hip_split_filter_x_pair1(const float *src, float *work, float *work2, int filter_type, int nx, int ny,
                                int nz, double box_len, double box_len_z, float R, float R2, int phases,
                                void *stream);
/* ... two grids (delta and x_e of a spin-temperature run's Eulerian loop), same rules */
int c21hip_split_filter_xy2_pair_eval(const float *src_a, float *work_a, float *work_a2, int filter_a,
                                      const float *src_b, float *work_b, float *work_b2, int filter_b, int nx,
                                      int ny, int nz, double box_len, double box_len_z, float R, float R2,
                                      int phases, void *stream);
/* one grid, two radii, window a of the tables built for the two-grid sweep of those radii */
int c21hip_split_filter_xy_shared_pair(const float *src, float *work, float *work2,
                                       int filter_type, int nx, int ny, int nz, double box_len,
                                       double box_len_z, float R, float R2, int table_slot,
                                       int table_slot2, int phases, void *stream);
/* Node tables of W(x) for the radii of one excursion-set call: when *enabled comes back 1 the
 * passes X of these radii (c21hip_split_filter_xy2[_pair], _xy_shared[_pair]) evaluate their
 * windows in the kernel and c21hip_window_tables need not be called; c21hip_wev_release ends it */
int c21hip_wev_prepare(int filter_a, float R_param_a, int filter_b, float R_param_b, int n_grids,
                       const float *R, int n_R, int nx, int ny, int nz, double box_len,
                       double box_len_z, int pair, int *enabled, void *stream);
void c21hip_wev_release(void);
int c21hip_wev_applicable(int filter_a, int filter_b, int n_grids, int nx, int ny, int nz);
/* one grid, two radii per pass-X sweep, under window a or b of the prepared set (phases: 2 pass X,
 * 4 / 8 pass Y of the first / second radius) */
int c21hip_split_filter_xy_single_pair(const float *src, float *work, float *work2, int filter_type,
                                       float R_param, int nx, int ny, int nz, double box_len,
                                       double box_len_z, float R, float R2, int phases, void *stream);
/* fused pass Z with a recombination model (CELL_RECOMB): barrier (1 + N_rec / (1 + delta)), delta_R
 * of a first crossing left in the Gamma_12 grid */
int c21hip_z_ionise_recomb_xe_supported(int nx, int ny, int nz);
int c21hip_split_z_ionise_recomb_xe(const float *delta_work, const float *stars_work, const float *xe_work,
                                    const float *nrec, double rec0, float *g12, unsigned char *first_cross,
                                    double *partials, int nx, int ny, int nz, int r_index,
                                    double rhocrit_omb, double ion_eff, int mass_dep_zeta, double f_limit,
                                    void *stream);
/* ... with N_rec of the previous snapshot filtered at this radius as the third line (CELL_RECOMB = false) */
int c21hip_split_z_ionise_recomb_nrec(const float *delta_work, const float *stars_work, const float *nrec_work,
                                      float *g12, unsigned char *first_cross, double *partials, int nx,
                                      int ny, int nz, int r_index, double rhocrit_omb, double ion_eff,
                                      int mass_dep_zeta, double f_limit, void *stream);
/* ... and with both: the x_e grid of a spin-temperature run AND the filtered N_rec (four spectra, nz = 512) */
int c21hip_z_ionise_recomb_xe_nrec_supported(int nx, int ny, int nz);
int c21hip_split_z_ionise_recomb_xe_nrec(const float *delta_work, const float *stars_work, const float *xe_work,
                                         const float *nrec_work, float *g12, unsigned char *first_cross,
                                         double *partials, int nx, int ny, int nz, int r_index,
                                         double rhocrit_omb, double ion_eff, int mass_dep_zeta, double f_limit,
                                         void *stream);
int c21hip_split_z_ionise_recomb(const float *delta_work, const float *stars_work, const float *nrec,
                                 double rec0, float *g12, unsigned char *first_cross,
                                 double *partials, int nx, int ny, int nz, int r_index,
                                 double rhocrit_omb, double ion_eff, int mass_dep_zeta,
                                 double f_limit, void *stream);
/* ... followed, per radius, by the pass Z of the filtered whalo_sfr: first crossings of r_index
 * (g12 holds their delta_R) receive Gamma_12 = g12_scale / (1 + delta_R) max(sfr_R, 0) */
int c21hip_split_z_sfr_gamma12(const float *sfr_work, const unsigned char *first_cross, float *g12,
                               int nx, int ny, int nz, int r_index, double g12_scale, void *stream);
int c21hip_z_ionise_recomb_supported(int nx, int ny, int nz);
/* Eulerian source models with an x_e grid: pass Z of the filtered x_e spectrum with the barrier
 * f_coll zeta > 1 - x_e(R) fused in (nion_dense: the radius' dense f_coll, *mean_dev its mean);
 * first crossings go into the mask, x_e(R) is never stored (IonisationBox.c:1091-1118) */
int c21hip_z_xe_mask_supported(int nx, int ny, int nz);
int c21hip_split_z_xe_mask(const float *xe_work, const float *nion_dense, const double *mean_dev,
                           unsigned char *first_cross, int nx, int ny, int nz, int r_index,
                           double mean_f_coll, int fix_mean, int mass_dep_zeta, double f_limit,
                           double ion_eff, void *stream);
/* first-crossing mask + Gamma_12 grid of the fused recombination loop -> x_HI = 0, z_reion,
 * mean free path = R[index] (R_dev: float per radius index) */
int c21hip_apply_first_cross_recomb(const unsigned char *first_cross, const float *R_dev,
                                    const float *prev_z_reion, int first_snapshot, double redshift,
                                    float *xH, float *z_reion, float *mfp, size_t ntot, void *stream);
/* W(kR) tables of one radius for c21hip_split_filter_xy2, on any stream */
int c21hip_window_tables(int table_slot, int filter_a, float R_param_a, int filter_b,
                         float R_param_b, int nx, int ny, int nz, double box_len,
                         double box_len_z, float R, void *stream);
/* pass Z of the filtered density + its extrema on the device (minmax_out[2]);
 * partials: 2 * nx*ny/16 doubles (IonisationBox.c:668-699) */
int c21hip_split_z_c2r_minmax(const float *split_work, float *real_out, long out_zstride, int nx,
                              int ny, int nz, double *partials, double *minmax_out, void *stream);
/* Eulerian table loop without the delta_R round trip (round 6): the extrema alone (nothing stored), then the
 * table sweep + banded barrier as the epilogue of a SECOND pass Z of the same spectrum (nx*ny/16 partial sums of
 * f_coll left in `partials` for c21hip_eul_band) */
int c21hip_z_table_band_supported(int nx, int ny, int nz);
int c21hip_split_z_minmax_only(const float *split_work, int nx, int ny, int nz, double *partials,
                               double *minmax_out, void *stream);
int c21hip_split_z_fcoll_table_band(const float *split_work, float *f_pend, const double *band_dev,
                                    const double *thr_prev_dev, unsigned char *first_cross, int r_index,
                                    int r_prev, int nx, int ny, int nz, int mode, double tab_min, double tab_width,
                                    const float *table_dev, double *partials, void *stream);
/* pass Z fused with the CONST-ION-EFF closed-form f_coll(delta_R): dense f_coll grid + sum
 * (IonisationBox.c:785-961, hmf.c:1205-1241); partials: 2 * nx*ny/16 doubles */
int c21hip_split_z_fcoll_erfc(const float *split_work, float *nion_dense, int nx, int ny, int nz,
                              double growthf, double sigma_min, double sigma_max, double delta_c,
                              double *partials, double *sum_out, void *stream);
/* pass Z with out = max(v, min_value) * const_factor into dense rows and stats_out[3] =
 * {min, max, sum} of the stored values on the device (SpinTemperatureBox.c:606-629);
 * partials: 3 * nx*ny/16 + 2 * (nx*ny/16384 + 2) doubles */
int c21hip_split_z_c2r_stats(const float *split_work, float *real_out, long out_zstride, int nx,
                             int ny, int nz, double min_value, double const_factor,
                             double *partials, double *stats_out, void *stream);
/* stats_out == NULL above defers the reduction: window i keeps its partials (nb = nx*ny/16 minima,
 * maxima, sums) at partials + i * stride and one launch reduces `count` windows */
int c21hip_batched_stats(const double *partials, long stride, int nb, int count,
                         double *stats_out, void *stream);
/* the same store + statistics from real rows of in_zstride floats (generic sizes; out may be
 * NULL for the statistics alone); partials: 3 * C21HIP_PARTIALS doubles */
int c21hip_floor_scale_stats(const float *in, long in_zstride, float *out, int nx, int ny, int nz,
                             double min_value, double const_factor, double *partials,
                             double *stats_out, void *stream);
/* pass Z storing v / divisor (0: no division) */
int c21hip_split_z_c2r_div(const float *split_work, float *real_out, long out_zstride, int nx,
                           int ny, int nz, float divisor, void *stream);
/* pass Z storing v * scale / divisor (0: no division), floored at -1 + 1e-7 when floor_density */
int c21hip_split_z_c2r_out(const float *split_work, float *real_out, long out_zstride, int nx,
                           int ny, int nz, float scale, float divisor, int floor_density,
                           void *stream);
int c21hip_split_xblock_log2(int nx); /* 0: plain split layout */
/* split-layout InitialConditions pipeline (ics_kernels.hip, plain layout only) */
int c21hip_split_kop(const float *in_split, float *out_split, int nx, int ny, int nz,
                     double box_len, double box_len_z, int axis0, int axis1, void *stream);
/* passes X, Y of the inverse transform of op(P), P = spectrum / k^2 (c21hip_split_kop with
 * axis0 = -2): the gradient / second-derivative operator is applied inside pass X */
int c21hip_split_sepop_xy(const float *split_src, float *split_work, int nx, int ny, int nz,
                          double box_len, double box_len_z, int axis0, int axis1, void *stream);
int c21hip_split_fold(const float *hi_split, float *lo_split, int nx, int ny, int nz, int f,
                      double box_len, double box_len_z, int axis0, int axis1, void *stream);
/* ... up to four folded spectra of one hi spectrum in one read: ops[k] = -1 identity, 0 / 1 / 2 gradient axis;
 * tophat_R > 0: the top-hat of that radius applied to the stored elements on the way (the filtered spectrum is
 * never written) */
int c21hip_split_fold_multi(const float *hi_split, float *const lo_split[4], const int ops[4], int n_out, int nx,
                            int ny, int nz, int f, double box_len, double box_len_z, float tophat_R, void *stream);
int c21hip_lpt2_source(const float *const diag[3], const float *const off[3], float *out, size_t n,
                       float norm, void *stream);
int c21hip_copy_filter_split(const float *src_split, float *dst_split, int nx, int ny, int nz,
                             double box_len, double box_len_z, int filter_type, float R,
                             float R_param, int apply, void *stream);
int c21hip_split_z_c2r(const float *split_work, float *real_out, long out_zstride, int nx, int ny,
                       int nz, void *stream);
/* Fused pass Z of delta_R and the filtered emissivity + sum(stars) + ionisation barrier for
 * Lagrangian source grids at radius index > 0; writes only first_cross (uint8 [N]).
 * partials: nx*ny/8 doubles.  reference: IonisationBox.c:634-638,821-837,1054-1151 */
int c21hip_split_z_ionise_stars(const float *delta_work, const float *stars_work,
                                unsigned char *first_cross, double *partials, double *sum_out,
                                int nx, int ny, int nz, int r_index, double rhocrit_omb,
                                double ion_eff, int mass_dep_zeta, double f_limit, void *stream);
/* sum_out == NULL in c21hip_split_z_ionise_stars defers the reduction: the partials of every
 * radius stay at partials + R * stride and ONE launch turns them into sums[R] / means[R] for
 * R = first, first - step, ... (count radii) */
int c21hip_z_ionise_partials(int nx, int ny, int nz);
int c21hip_batched_means(const double *partials, long stride, int n_partials, int first, int step,
                         int count, double ntot, int mass_dep_zeta, double f_limit, double *sums,
                         double *means, void *stream);
/* fused pass Z with the filtered x_e spectrum as a third grid (spin-temperature runs; the
 * barrier becomes f_coll zeta > 1 - x_e, IonisationBox.c:1118); 512/1024-point z-lines */
int c21hip_split_z_ionise_stars_xe(const float *delta_work, const float *stars_work,
                                   const float *xe_work, unsigned char *first_cross,
                                   double *partials, double *sum_out, int nx, int ny, int nz,
                                   int r_index, double rhocrit_omb, double ion_eff,
                                   int mass_dep_zeta, double f_limit, void *stream);
int c21hip_z_ionise_xe_supported(int nx, int ny, int nz);
/* passes X, Y of one grid with window a of the two-grid tables in buffer table_slot */
int c21hip_split_filter_xy_shared(const float *src, float *work, int filter_type, int nx, int ny,
                                  int nz, double box_len, double box_len_z, float R, int apply,
                                  int table_slot, void *stream);
/* time `reps` launches of one pass with HIP events on `stream` (bench.py roofline leg);
 * kind: 0 pass X and 1 pass Y as the excursion-set loop launches them (two grids, windows
 * filter_a / filter_b streamed from the per-radius tables), 2 fused pass Z, 3 plain pass Z,
 * 4 the window-table kernel */
int c21hip_bench_pass(int kind, int n, int filter_a, int filter_b, float R, float R_param_b,
                      double box_len, int reps, void *stream, float *ms_out);
/* deterministic single-workgroup sum of n doubles (ionize_kernels.hip) */
int c21hip_reduce_sum(const double *partials, int n, double *out, void *stream);
/* op 1 = min, 2 = max of n doubles; stage: >= n/1024 + 1 doubles of scratch for long inputs */
int c21hip_reduce_op(const double *partials, int n, int op, double *stage, double *out,
                     void *stream);

/* ---- grid_kernels.hip : generic sweeps ---- */
/* padded[l][k] = clip(dense[l][k] * factor, lo, hi); pad columns zeroed.
 * reference: IonisationBox.c:333-350 */
int c21hip_pack_clip(const float *dense, float *padded, int nx, int ny, int nz, double factor,
                     double lo, double hi, void *stream);
/* dense[l][k] = padded[l][k] * scale (scale applied in float) */
int c21hip_unpack_scale(const float *padded, float *dense, int nx, int ny, int nz, float scale,
                        void *stream);
/* buf[i] /= divisor, float arithmetic (reference: IonisationBox.c:356-359) */
int c21hip_divide_inplace(float *buf, size_t n, float divisor, void *stream);
/* buf[i] = (float)(buf[i] / divisor), double arithmetic (reference: filtering.c:422-424) */
int c21hip_divide_inplace_f64(float *buf, size_t n, double divisor, void *stream);
int c21hip_fill(float *buf, size_t n, float value, void *stream);
int c21hip_add_scalar(float *buf, size_t n, float value, void *stream); /* buf[i] += value */
/* out[i] = (double)in[i] */
int c21hip_widen(const float *in, double *out, size_t n, void *stream);

/* dst = src * W(kR), the fused memcpy + filter_box of copy_filter_transform.
 * reference: IonisationBox.c:577-631 + filtering.c:308-394.  `apply` = 0 copies only. */
int c21hip_copy_filter(const float *src_c, float *dst_c, int nx, int ny, int nz, double box_len,
                       double box_len_z, int filter_type, float R, float R_param, int apply,
                       void *stream);
/* the same with filter_box's full argument list: R_star of the multiple-scattering window 5 */
int c21hip_copy_filter_star(const float *src_c, float *dst_c, int nx, int ny, int nz,
                            double box_len, double box_len_z, int filter_type, float R,
                            float R_param, float R_star, int apply, void *stream);

/* ics_kernels.hip: the raw word pairs of the reference's IC random stream (gsl_stream.c: two accepted
 * 32-bit outputs per deviate, generator kind per x-row) -> gsl_ran_ugaussian deviates, in place */
int c21hip_gsl_words_to_deviates(void *buf, size_t n_deviates, const unsigned char *row_kind_dev,
                                 size_t deviates_per_row, void *stream);

/* ---- perturb_kernels.hip ---- */
/* move_grid_masses: map_mass.c:146-208.  `out` (double[out_dim]) must be zeroed by the caller.
 * fixed_out != NULL: the deposit may leave 64-bit FIXED-POINT integers (scale 2^44) in `out` instead of
 * doubles -- deterministic whatever the order of the atomics -- and says so in *fixed_out;
 * c21hip_widen_normalise(fixed = 1) reads them.  NULL: doubles. */
int c21hip_cic_scatter(const float *hires_density, const int dens_dim[3], const float *const vel[3],
                       const float *const vel2[3], const int vel_dim[3], double *out,
                       const int out_dim[3], double box_len, double box_len_z, double growth,
                       double init_growth, int lpt2, int *fixed_out, void *stream);
/* after a fixed-point deposit: *bad = 1 when a particle mass was non-finite or beyond what the 63 bits hold
 * (synchronises the stream) */
int c21hip_cic_fixed_status(int *bad, void *stream);
/* double grid -> padded float [, *= mass_factor, -= 1]: PerturbedField.c:115-128,180-210 */
/* ComputeHaloBox deposit (map_mass.c:214-344): exp(lerp(ln-table, delta*D)) * prefactor for the
 * two tables in tables_dev[2][NDELTA], CIC-deposited at the displaced positions (double grids) */
int c21hip_halobox_scatter(const float *src_density, const int dens_dim[3],
                           const float *const vel[3], const float *const vel2[3],
                           const int vel_dim[3], double *out_nion, double *out_sfr,
                           double *out_xray /* NULL: two values */, const int out_dim[3],
                           double box_len, double box_len_z, double growth, double init_growth,
                           int lpt2, const float *tables_dev /* [2 or 3][NDELTA] */,
                           double tab_min, double tab_width, double pref_nion, double pref_sfr,
                           double pref_xray, void *stream);
/* ComputeHaloBox with USE_MINI_HALOS.  get_log10_turnovers (HaloBox.c:465-516): n_chunks = the
 * N_THREADS shares over which upstream's running maximum of the atomic turnover restarts;
 * sums_dev[2] (zeroed by the caller) receive the sums of the two log10 grids. */
int c21hip_halobox_turnovers(size_t ntot, int n_chunks, int below_z_heat_max, double redshift,
                             double mturn_a_nofb, double m_turn, double vcb_const, double A_LW,
                             double BETA_LW, double A_VCB, double BETA_VCB, double sigma_vcb,
                             const float *prev_G12, const float *prev_z_reion, const float *J_21_LW,
                             const float *vcb, float *out_a, float *out_m, double *sums_dev,
                             void *stream);
/* move_grid_galprops with mini-halos (map_mass.c:285-321): low-resolution sources; ranges =
 * {delta min, width, log10 M_turn,a min, width, log10 M_turn,m min, width, fixed turnover grid
 * min, width}; prefactors = {nion, nion_mini, sfr, sfr_mini, xray}; out_xray / tab_xray may be
 * NULL; double accumulation grids */
int c21hip_halobox_scatter_mini(const float *src_density, const int dim[3],
                                const float *const vel[3], const float *const vel2[3],
                                const float *mturn_a, const float *mturn_m, double *out_nion,
                                double *out_sfr, double *out_sfr_mini, double *out_xray,
                                double box_len, double box_len_z, double growth, double init_growth,
                                int lpt2, const float *tab_sfrd, const float *tab_nion_a,
                                const float *tab_nion_m, const float *tab_sfrd_m,
                                const float *tab_xray, const double *ranges,
                                const double *prefactors, void *stream);
/* move_halo_galprops (map_mass.c:346-476) for a catalogue in device arrays: fp64 accumulation
 * grids, values per unit output-cell volume; out_sfr_mini / out_xray / out_wsfr and (without
 * mini-halos) the turnover grids may be NULL.  scratch: c21hip_halo_deposit_scratch_ints() ints
 * for the binned, LDS-tiled path (NULL or C21CM_HALO_DEPOSIT=direct: global atomics only). */
struct c21cm_halo_consts;
size_t c21hip_halo_deposit_scratch_ints(unsigned long long n_halos, const int out_dim[3]);
/* test_halo_props (HaloBox.c:658-779): twelve floats per halo; lw = {A_LW, BETA_LW, A_VCB,
 * BETA_VCB, sigma_vcb, vcb_const, M_TURN}; the grids are read with mini-halos only */
int c21hip_halo_props(const struct c21cm_halo_consts *consts, unsigned long long n_halos,
                      const float *masses, const float *coords, const float *star_rng,
                      const float *sfr_rng, const float *xray_rng, const int dim[3],
                      double cell_length, double redshift, int below_z_heat_max, int vcb_flucts,
                      const double lw[7], const float *vcb, const float *J21, const float *z_re,
                      const float *G12, float *out, void *stream);
int c21hip_halo_deposit(const struct c21cm_halo_consts *consts, unsigned long long n_halos,
                        const float *masses, const float *coords, const float *star_rng,
                        const float *sfr_rng, const float *xray_rng, const float *const vel[3],
                        const float *const vel2[3], const int vel_dim[3], const int out_dim[3],
                        double box_len, double box_len_z, double growth, double init_growth, int lpt2,
                        const float *mturn_a, const float *mturn_m, double *out_nion, double *out_sfr,
                        double *out_sfr_mini, double *out_xray, double *out_wsfr, int *scratch,
                        void *stream);
int c21hip_narrow(const double *in, float *out, float *out_scaled, double scale, size_t n,
                  void *stream);
/* {min, max} of n floats into out2 (device); partials: 2 * 2048 doubles */
int c21hip_minmax_dense(const float *a, size_t n, double *partials, double *out2, void *stream);
int c21hip_widen_normalise(const double *in, float *padded, int nx, int ny, int nz, int normalise,
                           double mass_factor, int fixed, void *stream);
/* padded = (float)(factor * dense): PerturbedField.c:64-80 */
int c21hip_scale_pack(const float *dense, float *padded, int nx, int ny, int nz, double factor,
                      void *stream);
/* nearest-index gather from a padded real grid [, /divisor][, clip at -1+1e-7] into a dense
 * (or padded) low-res grid: PerturbedField.c:162-177,251-276,367-383,450-464 */
int c21hip_gather(const float *src_padded, const int src_dim[3], float *dst, const int lo_dim[3],
                  int dst_padded, float divisor, int clip, void *stream);
/* grid = saved * (dD/dt / D) i k_axis / k^2 / N: PerturbedField.c:320-350 */
int c21hip_velocity_kspace(const float *saved_c, float *grid_c, int nx, int ny, int nz,
                           double box_len, double box_len_z, int axis, double dDdt_over_D,
                           void *stream);

/* ---- ics_kernels.hip ---- */
/* delta_k = sqrt(V P/2)(a + ib) with Hermitian planes: InitialConditions.c:26-139 */
/* compute_relative_velocities (InitialConditions.c:141-238): k-space operator of one component,
 * and the subsampled sum of squares (sqrt / VOLUME on the last component) into lowres_vcb */
int c21hip_vcb_op(const float *in_c, float *out_c, int nx, int ny, int nz, double box_len,
                  double box_len_z, int axis, const double *h_by_m_dev, void *stream);
int c21hip_vcb_accumulate(const float *src_padded, const int hi_dim[3], float *dst,
                          const int lo_dim[3], int first, int last, float volume, void *stream);
int c21hip_sample_modes(float *cbox, int nx, int ny, int nz, const double *pk_by_m_dev,
                        float volume, unsigned long long seed, const double *deviates_dev,
                        void *stream);
/* ... straight into the split layout of the native transforms */
int c21hip_sample_modes_split(float *split, int nx, int ny, int nz, const double *pk_by_m_dev, float volume,
                              unsigned long long seed, const double *deviates_dev, void *stream);
/* axis1 < 0: out = in*i*k_axis0/k^2 (:240-267); else out = -k_axis0*k_axis1*in/k^2 (:269-297) */
int c21hip_kspace_op(const float *in_c, float *out_c, int nx, int ny, int nz, double box_len,
                     double box_len_z, int axis0, int axis1, void *stream);
/* padded = dense * VOLUME / N (:637-653) */
int c21hip_pack_density(const float *dense, float *padded, int nx, int ny, int nz, float volume,
                        void *stream);
/* box += phi_ii*phi_jj; box -= phi_ij^2 (:451-482) */
int c21hip_lpt2_accumulate(float *box, const float *phi_ij_padded, const float *diag_i,
                           const float *diag_j, int nx, int ny, int nz, void *stream);

/* ---- ionize_kernels.hip ---- */
typedef struct c21hip_ionize_args {
    int nx, ny, nz;
    int r_index;        /* 0 = last (cell-scale) radius: partial ionisation branch */
    int lagrangian;     /* stars grid divides by rhocrit_omb*(1+delta)             */
    int mass_dep_zeta;  /* floor f at f_limit                                      */
    int use_ts_fluct;
    int minimize_memory;
    int first_snapshot; /* previous z_reion treated as -1                          */
    int fix_mean;
    double mean_f_coll; /* fix_mean numerator                                      */
    double f_limit;
    double ion_eff_factor;
    double rhocrit_omb;
    double photoncons_factor;
    double redshift;
    double TK_nofluct, adia_TK_term, T_re;
} c21hip_ionize_args;

/* All reductions are two-stage and deterministic: `partials` is device scratch of
 * C21HIP_PARTIALS doubles, the result lands in a device double. */
#define C21HIP_PARTIALS 4096

/* min/max of the filtered delta (the clip of :689 is applied in registers by the
 * consumers).  reference: IonisationBox.c:668-699.  minmax_out = {min, max}. */
int c21hip_clip_minmax(float *delta_fil, int nx, int ny, int nz, double *partials,
                       double *minmax_out, void *stream);
/* Eulerian source models: f_coll(delta_R) -> unnormalised_nion (dense) and sum(f_coll).
 * mode: enum c21cm_fcoll_mode (ERFC / TABLE_LINEAR / TABLE_EXP).
 * reference: IonisationBox.c:773-962 */
int c21hip_fcoll_eulerian(const float *delta_fil, float *nion_dense, int nx, int ny, int nz,
                          int mode, double growthf, double sigma_min, double sigma_max,
                          double delta_c, double tab_min, double tab_width,
                          const float *table_dev, double *partials, double *sum_out,
                          void *stream);
/* the same on rows of `zstride` floats: 2 (nz / 2 + 1) (padded) or nz (dense: 16-byte pieces when nz % 4 == 0) */
int c21hip_fcoll_eulerian_zs(const float *delta_fil, long zstride, float *nion_dense, int nx, int ny, int nz,
                             int mode, double growthf, double sigma_min, double sigma_max, double delta_c,
                             double tab_min, double tab_width, const float *table_dev, double *partials,
                             double *sum_out, void *stream);
/* Lagrangian source grids: sum(stars) + barrier test + partial ionisation, one sweep.
 * first_cross != NULL switches to shard mode (records the radius index instead of
 * touching xH / z_reion).  reference: IonisationBox.c:821-837,1008-1201 */
int c21hip_ionise_stars(const c21hip_ionize_args *a, const float *delta_fil,
                        const float *stars_fil, const float *xe_fil, const float *density,
                        const float *prev_z_reion, const float *kinetic_temp_neutral, float *xH,
                        float *z_reion, float *kinetic_temperature, unsigned char *first_cross,
                        double *partials, double *sum_out, void *stream);
/* Eulerian barrier test; mean_dev holds the clamped grid mean (fix_mean denominator).
 * reference: IonisationBox.c:1008-1201 */
int c21hip_ionise_eulerian(const c21hip_ionize_args *a, const float *nion_dense,
                           const float *xe_fil, const float *density, const float *prev_z_reion,
                           const float *kinetic_temp_neutral, const double *mean_dev, float *xH,
                           float *z_reion, float *kinetic_temperature,
                           unsigned char *first_cross, void *stream);
/* mean = clamp(sum/N): reference IonisationBox.c:960,1566-1576 */
int c21hip_finish_mean(const double *sum_dev, double ntot, int mass_dep_zeta, double f_limit,
                       double *mean_dev, void *stream);
/* Ionised-cell temperatures + sum(xH) + NaN flag.  reference: IonisationBox.c:1203-1256,1597-1608 */
int c21hip_finalize(const c21hip_ionize_args *a, double stored_redshift, const float *density,
                    const float *kinetic_temp_neutral, const float *xH, const float *z_reion,
                    float *kinetic_temperature, size_t ntot, double *partials, double *sum_out,
                    int *flag_out, void *stream);
/* set_fully_neutral_box: IonisationBox.c:531-565 */
int c21hip_neutral_box(const float *density, const float *xe, const float *Tneutral, float *xH,
                       float *Tk, size_t ntot, int ts, double global_xH, double TK, double adia,
                       void *stream);
int c21hip_any_nonzero(const float *a, size_t n, int *flag_host, void *stream);
/* xH/z_reion from a max-reduced first_cross mask (multi-GPU tail). */
/* Eulerian barrier test on the dense f_coll grid, first crossing into a uint8 mask
 * (radius index > 0, no x_e grid; IonisationBox.c:1022-1027,1077,1118) */
/* find_ionised_regions with a recombination model (IonisationBox.c:1084-1140): rec in the barrier,
 * Gamma_12 / mean free path at the first crossing.  src_grid = filtered n_ion (padded real,
 * lagrangian) or the dense unnormalised_nion grid; nrec_fil padded (no CELL_RECOMB) or prev_nrec
 * dense (CELL_RECOMB; one value for the homogeneous model).  sum_out (Lagrangian f_coll sum) may
 * be NULL. */
int c21hip_ionise_recomb(const c21hip_ionize_args *a, int lagrangian, int inhomo, int cell_recomb,
                         double R, double gamma_prefactor, const float *delta_fil,
                         const float *src_grid, const float *sfr_fil, const float *xe_fil,
                         const float *nrec_fil, const float *prev_nrec, const float *density,
                         const float *prev_z_reion, const float *kinetic_temp_neutral,
                         const double *mean_dev, float *xH, float *z_reion,
                         float *kinetic_temperature, float *G12, float *mfp, double *partials,
                         double *sum_out, void *stream);
/* USE_MINI_HALOS (E-INTEGRAL).  calculate_mcrit_boxes (IonisationBox.c:403-457): dense [N]
 * device arrays, sums_out[2] = sum of the two log10 grids, *flag_dev set on a NaN/zero threshold.
 * partials: >= 2 * 2048 doubles. */
int c21hip_mturn_grids(size_t ntot, int first_snapshot, double redshift, double mturn_a_nofb,
                       double mturn_m_nofb, double vcb_const, double A_LW, double BETA_LW,
                       double A_VCB, double BETA_VCB, double sigma_vcb, const float *prev_G12,
                       const float *prev_z_reion, const float *J_21_LW, const float *vcb,
                       float *out_a, float *out_m, double *partials, double *sums_out,
                       int *flag_dev, void *stream);
/* calculate_fcoll_grid with both populations and the per-radius history (:838-936).  Filtered
 * grids padded, history / outputs dense; tables_dev = 4 tables of NDELTA x NMTURN floats (acg, mcg,
 * previous-redshift acg, mcg); ranges = {delta min, width, prev delta min, width, log10 M_turn,a
 * min, width, log10 M_turn,m min, width}; sums_out[2] = sum f_a, sum f_m. */
int c21hip_fcoll_mini(int nx, int ny, int nz, int need_prev, const double *ranges,
                      const float *delta_fil, const float *pdelta_fil, const float *mta_fil,
                      const float *mtm_fil, const float *tables_dev, const float *prev_nion,
                      const float *prev_mini, float *nion_out, float *mini_out, double *partials,
                      double *sums_out, void *stream);
/* find_ionised_regions with the two-population barrier (:1068-1200), recombinations optional.
 * lagrangian: nion_dense = stars_fil and mini_dense = sfr_fil (padded rows; sfr_fil only with a
 * recombination model), the second population contributes its floor f_limit_mcg only, and the sum
 * of the source grid goes to sum_out (partials: C21HIP_PARTIALS doubles). */
int c21hip_ionise_mini(const c21hip_ionize_args *a, int lagrangian, int recomb, int inhomo,
                       int cell_recomb,
                       double R, double gamma_prefactor, double gamma_prefactor_mini,
                       double ion_eff_mini, double f_limit_mcg, double mean_f_coll_mini,
                       const float *delta_fil, const float *nion_dense, const float *mini_dense,
                       const float *xe_fil, const float *nrec_fil, const float *prev_nrec,
                       const float *density, const float *prev_z_reion,
                       const float *kinetic_temp_neutral, const double *mean_a_dev,
                       const double *mean_m_dev, float *xH, float *z_reion,
                       float *kinetic_temperature, float *G12, float *mfp, double *partials,
                       double *sum_out, void *stream);
/* IONISE_ENTIRE_SPHERE (IonisationBox.c:1150-1158): every cell of the first-crossing mask flags
 * the cells closer than its radius; rsq_dev[r] = (R_r in cells)^2 as update_in_sphere forms it
 * (float), compared strictly with the integer distances (bubble_helper_progs.c:292-325) */
int c21hip_paint_spheres(const unsigned char *first_cross, const float *rsq_dev, float *xH, int nx,
                         int ny, int nz, void *stream);
/* set_recombination_rates, inhomogeneous model (IonisationBox.c:1277-1339); rate_scale =
 * fabs_dtdz * dz; rr tables on the device */
int c21hip_recomb_rates(const float *density, const float *G12, const float *xH,
                        const float *prev_nrec, float *nrec, size_t ntot, double stored_redshift,
                        double rate_scale, const double *rr_y_dev, const double *rr_c_dev,
                        int *flag_dev, void *stream);
/* sharded R loop with a recombination model: (mean free path, Gamma_12) of a rank's first
 * crossings packed into order-preserving 64-bit keys, and the reduced keys applied to the outputs */
int c21hip_pack_cross_keys(const float *mfp, const float *G12, unsigned long long *keys,
                           size_t ntot, void *stream);
int c21hip_apply_cross_keys(const unsigned long long *keys, const float *prev_z_reion,
                            int first_snapshot, double redshift, float *xH, float *z_reion,
                            float *G12, float *mfp, size_t ntot, void *stream);
/* one bit per cell of a first-crossing grid (non-zero -> 1), and the OR of `world` such packed
 * grids (stride_words apart) expanded back to 0 / 1 bytes: the exchange of a sharded run */
int c21hip_pack_mask_bits(const unsigned char *fc, unsigned *bits, size_t ntot, void *stream);
int c21hip_or_unpack_mask_bits(const unsigned *bits, size_t stride_words, int world,
                               unsigned char *fc, size_t ntot, void *stream);
/* in-loop kernel timing (bench.py): HIP events around every pass launch on its stream while enabled;
 * kinds as in c21hip_bench_pass (12: one-grid pass Z + closed-form f_coll of the Eulerian loop; 1 pass Y, 2 fused pass Z, 7 / 8 pass X / two-radius pass X with
 * evaluated windows, 0 / 6 with streamed tables, 9 forward line passes) */
/* closed-form Eulerian loop: pass Z + f_coll of a radius with the barrier of the PREVIOUS radius of the
 * loop (its dense grid, its mean, its index) applied in the same sweep */
int c21hip_z_fcoll_erfc_mask_supported(int nx, int ny, int nz);
int c21hip_split_z_fcoll_erfc_mask(const float *split_work, float *nion_dense, const float *nion_prev,
                                   const double *mean_prev_dev, unsigned char *first_cross, int r_index_prev,
                                   int fix_mean, double mean_f_coll, int mass_dep_zeta, double f_limit,
                                   double ion_eff, int nx, int ny, int nz, double growthf, double sigma_min,
                                   double sigma_max, double delta_c, double *partials, double *sum_out,
                                   void *stream);
/* closed-form Eulerian loop, banded barrier: the barrier of a radius decided inside its own pass Z
 * wherever it does not depend on the exact box mean.  The barrier test is monotone in the cell's f_coll,
 * i.e. a threshold; band_dev[0] / band_dev[1] = the thresholds at the two ends of the predicted band of
 * mean_f_coll / mean (at or above [0]: crosses for sure; below [1]: does not).  Cells in between carry
 * the marker 255 in first_cross and their f_coll in f_pend until the next radius' sweep (r_prev,
 * thr_prev_dev = its exact threshold) or c21hip_eul_resolve_pending settles them.  c21hip_eul_band =
 * [the sum of the pass-Z partials, c21hip_reduce_sum's arithmetic, when partials != NULL +] finish_mean of r_cur + its exact threshold (thr_dev[r_cur]) + the check of r_cur's band (*fail_dev =
 * the largest radius index whose band missed; c21hip_eul_rewind clears what that radius and the ones
 * after it wrote) + the band of r_next. */
int c21hip_split_z_fcoll_erfc_band(const float *split_work, float *f_pend, const double *band_dev,
                                   const double *thr_prev_dev, unsigned char *first_cross, int r_index,
                                   int r_prev, int nx, int ny, int nz, double growthf, double sigma_min,
                                   double sigma_max, double delta_c, double *partials, double *sum_out,
                                   void *stream);
int c21hip_eul_band(const double *partials, int n, double *sum_dev, double ntot, int mass_dep_zeta,
                    double f_limit, double *means_dev, int r_cur, int r_p1, int r_p2, double t_cur,
                    double t_next, int r_next, int cur_banded, int fix_mean, double mean_f_coll,
                    double ion_eff, double min_rel, double shift, double *band_dev, double *thr_dev,
                    int *fail_dev, unsigned *counter_dev, int mf_space, int quad, double *pred_dev,
                    void *stream);
/* the same for the table models WITH an x_e grid (mf_space = 1: the band and thr_dev hold the mean fix
 * itself -- the barrier f mf zeta > 1 - x_e has no single threshold): pass Z of the filtered x_e + the
 * table sweep of the dense filtered density + the barrier in one launch (512-point z-lines) */
int c21hip_z_xe_fcoll_band_supported(int nx, int ny, int nz);
int c21hip_split_z_xe_fcoll_band(const float *xe_work, const float *delta_fil, long delta_zstride, float *f_pend,
                                 float *xe_pend, const double *band_dev, const double *mf_prev_dev,
                                 unsigned char *first_cross, int r_index, int r_prev, int mode, double tab_min,
                                 double tab_width, const float *table_dev, int mass_dep_zeta, double f_limit,
                                 double ion_eff, int nx, int ny, int nz, double *partials, void *stream);
int c21hip_eul_resolve_pending_xe(int r_index, const float *f_pend, const float *xe_pend, const double *mf_dev,
                                  int mass_dep_zeta, double f_limit, double ion_eff, unsigned char *first_cross,
                                  size_t ntot, void *stream);
int c21hip_eul_rewind(unsigned char *first_cross, int r_fail, size_t ntot, void *stream);
/* the same for the table modes: c21hip_fcoll_eulerian with the barrier decided in the sweep; the dense
 * f_coll grid is not written, *n_partials_out partial sums stay in `partials` for c21hip_eul_band */
int c21hip_fcoll_eulerian_band(const float *delta_fil, long zstride, float *f_pend, unsigned char *first_cross,
                               int nx, int ny, int nz, int mode, double tab_min, double tab_width,
                               const float *table_dev, const double *band_dev, const double *thr_prev_dev,
                               int r_index, int r_prev, double *partials, int *n_partials_out, void *stream);
int c21hip_eul_resolve_pending(int r_index, const float *f_pend, const double *thr_dev,
                               unsigned char *first_cross, size_t ntot, void *stream);
void c21hip_ktime_enable(int on);
int c21hip_ktime_report(int kind, double *ms_total, int *count);
int c21hip_max_into(void *dst, const void *src, size_t count, int bytes_per_element, void *stream);
/* sharded fused recombination loop: own slab (mask, g12; in place) against n_peers received slabs
 * of `stride` cells each -- larger first-crossing index wins, with its Gamma_12 */
int c21hip_combine_cross_g12(unsigned char *mask, float *g12, const unsigned char *peer_mask,
                             const float *peer_g12, int n_peers, size_t stride, size_t n, void *stream);
int c21hip_sum_float(const float *v, size_t n, double *partials, double *sum_out, void *stream);
int c21hip_eulerian_mask(const c21hip_ionize_args *a, const float *nion_dense,
                         const float *xe_dense, const double *mean_dev,
                         unsigned char *first_cross, void *stream);
/* mask of radii > 0 + radius index 0 + post-loop sweep of the fused Lagrangian path in one pass
 * (IonisationBox.c:1031-1256,1597-1608); partials: 2 * 2048 doubles; writes every z_reion.
 * stars_direct = 1: stars_fil is the dense emissivity input (clipped on load) instead of its
 * padded, window-less transform round trip.  a->use_ts_fluct (stars_direct only): xe_dense and
 * kinetic_temp_neutral are the dense x_e / T_k inputs of the spin-temperature run. */
int c21hip_final_sweep(const c21hip_ionize_args *a, double stored_redshift,
                       const unsigned char *first_cross, const float *stars_fil,
                       const float *density, const float *prev_z_reion, float *xH, float *z_reion,
                       float *kinetic_temperature, double *partials, double *sum_stars_out,
                       double *sum_xh_out, int *flag_out, int stars_direct, const float *xe_dense,
                       const float *kinetic_temp_neutral, void *stream);
/* The final sweep walks the box in CHUNKS of contiguous cells, one workgroup each; the chunking depends on
 * the box alone.  _chunks: their number and size (dense: the sweep reads dense grids); _range: chunks
 * [chunk_begin, chunk_end) only (chunk_end < 0: all), partials left at partials[chunk] (stars) and
 * partials[2048 + chunk] (x_HI), no reduce; _reduce: the fixed-order reduce over ALL chunk partials.
 * A rank that sweeps a slab of whole chunks leaves the partials the single pass leaves there (sharded
 * finish phase, c21cm_ionize_shard_finish_slab). */
int c21hip_final_sweep_chunks(const c21hip_ionize_args *a, int dense, int *n_chunks, size_t *chunk_cells);
int c21hip_final_sweep_range(const c21hip_ionize_args *a, double stored_redshift,
                             const unsigned char *first_cross, const float *stars_fil,
                             const float *density, const float *prev_z_reion, float *xH, float *z_reion,
                             float *kinetic_temperature, double *partials, int *flag_out, int stars_direct,
                             const float *xe_dense, const float *kinetic_temp_neutral, int chunk_begin,
                             int chunk_end, void *stream);
int c21hip_final_sweep_reduce(const c21hip_ionize_args *a, int dense, double *partials,
                              double *sum_stars_out, double *sum_xh_out, void *stream);
int c21hip_final_sweep_eulerian_range(const c21hip_ionize_args *a, double stored_redshift,
                                      const unsigned char *first_cross, const float *nion_dense,
                                      const double *mean_dev, const float *density, const float *prev_z_reion,
                                      float *xH, float *z_reion, float *kinetic_temperature, double *partials,
                                      int *flag_out, const float *xe_dense, const float *kinetic_temp_neutral,
                                      int chunk_begin, int chunk_end, void *stream);
/* the same sweep for the Eulerian source models: the dense f_coll grid of radius index 0 and its box mean
 * in place of the emissivity grid (apply_first_cross + ionise_eulerian<LAST> + finalize in one pass) */
int c21hip_final_sweep_eulerian(const c21hip_ionize_args *a, double stored_redshift,
                                const unsigned char *first_cross, const float *nion_dense,
                                const double *mean_dev, const float *density, const float *prev_z_reion,
                                float *xH, float *z_reion, float *kinetic_temperature, double *partials,
                                double *sum_xh_out, int *flag_out, const float *xe_dense,
                                const float *kinetic_temp_neutral, void *stream);
/* delta_T (and tau_21) per cell + their sum (BrightnessTemperatureBox.c:58-87); partials: 2048 */
int c21hip_brightness_temp(const float *density, const float *xH, const float *Ts, float *bt,
                           float *tau, size_t n, float const_factor, float T_rad, double redshift,
                           int use_ts, double *partials, double *sum_out, void *stream);
int c21hip_apply_first_cross(const unsigned char *first_cross, const float *prev_z_reion,
                             int first_snapshot, double redshift, float *xH, float *z_reion,
                             size_t ntot, void *stream);

/* ---- ts_kernels.hip : per-cell part of ComputeTsBox (SpinTemperatureBox.c:892-927,1010-1086,
 * 1210-1383,1499-1848) ---- */
typedef struct c21hip_ts_args { /* the scalars of c21cm_ts_spec, passed by value */
    int n_step, lagrangian, use_xray_heating, use_cmb_heating, use_lya_heating, no_light;
    int table_exp; /* Eulerian: the per-shell table holds ln SFRD (1) or dfcoll/dz itself (0) */
    double redshift, dzp, growth_ratio;
    double No, N_b0, h_frac, he_frac, k_B, h_p, m_p, c_cms, A10, T_21, lambda_21, nu_Ly_alpha;
    double clumping_factor;
    double xray_prefactor, Trad, Ts_prefactor, xa_tilde_prefactor, xc_inverse, dcomp_dzp_prefactor;
    double Nb_zp, N_zp, lya_star_prefactor, volunit_inv, hubble_zp, growth_zp, dgrowth_dzp, dt_dzp;
    double sfr_scale, xray_scale;
    int sums_ready; /* the shell loop already ran (USE_MINI_HALOS: c21hip_ts_accumulate_mini) */
} c21hip_ts_args;
/* USE_MINI_HALOS (E-INTEGRAL).  Mini shell buffer: 6 per-shell rows (starlya, lya_cont, lya_inj of
 * the molecularly cooled population, the Lyman-Werner prefactors of both populations,
 * avg_fix_term_MINI written by c21hip_ts_sfrd_means_mini). */
#define C21HIP_TS_MINI_ROWS 6
int c21hip_ts_mcrit_grid(const float *J_21_LW, const float *vcb, double vcb_const, double redshift,
                         double A_LW, double BETA_LW, double A_VCB, double BETA_VCB,
                         double sigma_vcb, double m_turn, float *out, size_t ntot, void *stream);
int c21hip_ts_sfrd_means_mini(const float *filtered_density, const float *filtered_mcrit,
                              const float *tables2_dev, const double *dev_tab,
                              double *mini_shell_dev, const double *mean_sfr_mini_dev, int n_step,
                              size_t ntot, double mt_min, double mt_width, double *partials,
                              double *ave_out_dev, void *stream);
/* Lagrangian source grids with mini-halos: sfr_lw / sfr_mini_lw NULL unless the straight-line
 * copies exist (LYA_MULTIPLE_SCATTERING); same outputs and follow-up as the next function */
int c21hip_ts_accumulate_grids_mini(const c21hip_ts_args *a, const float *prev_xe, const float *sfr,
                                    const float *xray, const float *sfr_mini, const float *sfr_lw,
                                    const float *sfr_mini_lw, const double *dev_tab,
                                    const double *mini_shell_dev, double *sums_ws, float *J_21_LW,
                                    size_t ntot, void *stream);
/* the shell loop with both populations: sums_ws (6 * ntot doubles) and J_21_LW; follow with
 * c21hip_ts_cells(a with sums_ready = 1, ...) */
int c21hip_ts_accumulate_mini(const c21hip_ts_args *a, double sfr_scale_mini,
                              double xray_scale_mini, double mt_min, double mt_width,
                              const float *prev_xe, const float *delNL0, const float *mcrit,
                              const float *tables_dev, const float *tables2_dev,
                              const double *dev_tab, const double *mini_shell_dev, double *sums_ws,
                              float *J_21_LW, size_t ntot, void *stream);
/* device table buffer: 10 per-shell rows (z_edge_factor, xray_R_factor, starlya, lya_cont, lya_inj,
 * zpp_growth, tab_min, tab_width, avg_fix_term, 1 / tab_width) then the three [14][n_step]
 * frequency tables */
#define C21HIP_TS_SHELL_ROWS 10
size_t c21hip_ts_table_doubles(int n_step);
/* SFRD_TABLE: box mean of exp(table) per shell -> avg_fix_term row of dev_tab, means to ave_out;
 * partials: 512 * n_step doubles */
int c21hip_ts_sfrd_means(const float *filtered_density, const float *tables_dev, int table_exp,
                         double *dev_tab, const double *mean_sfr_zpp_dev, int n_step, size_t ntot,
                         double *partials, double *ave_out_dev, void *stream);
/* the two cell sweeps (R loop into sums_ws = 6 * ntot doubles, then the temperature update);
 * grid_a/grid_b = filtered_sfr/filtered_xray (lagrangian) or delNL0/NULL; partials: 6 * 2048
 * doubles; sums_out_dev: 6 doubles (Ts, Tk, x_e, J_alpha, xheat, xion) */
int c21hip_ts_cells(const c21hip_ts_args *a, const float *density, const float *prev_Ts,
                    const float *prev_Tk, const float *prev_xe, const float *grid_a,
                    const float *grid_b, const float *tables_dev, const double *dev_tab,
                    const double *lya_dEC_dev, const double *lya_dEI_dev, float *Ts_out,
                    float *Tk_out, float *xe_out, size_t ntot, double *sums_ws, double *partials,
                    double *sums_out_dev, int *flag_dev, void *stream);
/* the shell loop alone (sums_ws: [6][ntot] doubles) */
int c21hip_ts_shell_loop(const c21hip_ts_args *a, const float *prev_xe, const float *grid_a,
                         const float *grid_b, const float *tables_dev, const double *dev_tab,
                         double *sums_ws, size_t ntot, void *stream);
/* sharded ComputeTsBox: cell slabs of the ranks (rank r owns [begin(r), begin(r + 1))), the partial
 * sums of every peer's slab packed per peer ([world - 1] slots of c21hip_ts_slot_elems elements, each
 * [rows][maxlen] doubles -- or floats scaled by a per-row power of two, whose exponents close the slot), and the
 * complete sums of this rank's slab from its own partial and the received ones.  rowmax: 64 bytes of device
 * scratch, written by the pack and read by the combine of the same call. */
size_t c21hip_ts_slab_begin(size_t ntot, int world, int r);
size_t c21hip_ts_slot_elems(int rows, size_t maxlen, int as_float);
int c21hip_ts_pack_slabs(const double *sums, size_t ntot, int world, int rank, int rows,
                         size_t maxlen, int as_float, void *rowmax, void *out, void *stream);
int c21hip_ts_combine_slab(const double *sums, size_t ntot, int world, int rank, int rows,
                           size_t maxlen, int as_float, const void *rowmax, const void *recv, double *out,
                           void *stream);
struct c21cm_ts_first_spec;
int c21hip_ts_first(const struct c21cm_ts_first_spec *s, const float *density, float *Ts_out,
                    float *Tk_out, float *xe_out, size_t ntot, void *stream);
/* spin-exchange rate coefficients kappa_10 (H-H, e-H, p-H) at temperature T [K] */
void c21hip_kappa_rates(double T, double *k_HH, double *k_eH, double *k_pH);

#ifdef __cplusplus
}
#endif
#endif
