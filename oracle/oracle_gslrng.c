/*
 * oracle_gslrng.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The random stream of the reference's initial conditions, restated so that the SAME
 * random_seed gives the SAME universe as upstream and the reference's own HDF5 fixtures
 * (tests/test_data/perturb_field_data_*.h5, power_spectra_*.h5) become usable as pins.
 *
 * reference call sites (the only part that lives in /root/reference):
 *   src/py21cmfast/src/rng.c:31-90              seed_rng_threads
 *   src/py21cmfast/src/InitialConditions.c:103-139  sample_ic_modes (two gsl_ran_ugaussian per mode,
 *                                                   `#pragma omp for` over n_x, one generator per thread)
 *   src/py21cmfast/src/InitialConditions.c:26-101   adj_complex_conj
 *
 * The generators themselves are GSL's (third-party, absent from /root/reference and from this
 * image; version unpinned upstream: `conda install gsl`).  They are restated here from their
 * published algorithms:
 *   gsl_rng_mt19937   Matsumoto & Nishimura 1998, with the 2002 initialisation
 *                     s[i] = 1812433253 (s[i-1] ^ (s[i-1] >> 30)) + i, seed 0 -> 4357,
 *                     double = get / 2^32                                   (GSL rng/mt.c)
 *   gsl_rng_gfsr4     Ziff 1998 four-tap shift register (471, 1586, 6988, 9689; 2^14 words),
 *                     seeded bitwise from the LCG x -> 69069 x mod 2^32 with the 32-word
 *                     "orthogonalisation" of the diagonal                  (GSL rng/gfsr4.c)
 *   gsl_rng_uniform_pos / uniform_int                                      (GSL rng/gsl_rng.h)
 *   gsl_ran_choose / gsl_ran_shuffle  sequential selection sampling, Fisher-Yates from the top
 *                                                                          (GSL randist/shuffle.c)
 *   gsl_ran_ugaussian = gsl_ran_gaussian(r, 1): polar Box-Muller, x, y = 2 u_pos - 1 until
 *                     0 < r2 <= 1, returns y sqrt(-2 ln r2 / r2) (second value discarded)
 *                                                                          (GSL randist/gauss.c)
 * Thread t of seed_rng_threads uses generator t mod 5 of (mt19937, gfsr4, cmrg, mrg, taus2).
 * Every pinnable fixture of the reference uses N_THREADS <= 2
 * (tests/produce_integration_test_data.py:62: N_THREADS = 2, :213-220: 1); the other three
 * generators are pinned by GSL's own self-test values (see below).
 *
 * Pinning: mt19937 is checked word for word against numpy's MT19937 bit generator with legacy
 * seeding (the same init_genrand) in tests/test_oracle_gslrng.py; the whole chain
 * (seeding -> stream -> delta_k -> ICs -> PerturbedField -> binned power) is pinned by the
 * reference's own fixtures at the reference's own tolerances in tests/test_reference_fixtures.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ------------------------------------------------------------------ mt19937 */
#define MT_N 624
#define MT_M 397
typedef struct {
    uint32_t mt[MT_N];
    int mti;
} mt_state;

static void mt_set(mt_state *s, unsigned long seed) {
    if (seed == 0) seed = 4357;
    s->mt[0] = (uint32_t)(seed & 0xffffffffUL);
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = (uint32_t)(1812433253UL * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (unsigned long)i);
    s->mti = MT_N;
}

static inline uint32_t mt_get(mt_state *s) {
    uint32_t *const mt = s->mt;
    if (s->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; kk++) {
            const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        const uint32_t y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->mti = 0;
    }
    uint32_t k = mt[s->mti++];
    k ^= (k >> 11);
    k ^= (k << 7) & 0x9d2c5680u;
    k ^= (k << 15) & 0xefc60000u;
    k ^= (k >> 18);
    return k;
}

/* ------------------------------------------------------------------ gfsr4 */
#define GF_A 471
#define GF_B 1586
#define GF_C 6988
#define GF_D 9689
#define GF_M 16383
typedef struct {
    int nd;
    uint32_t ra[GF_M + 1];
} gfsr4_state;

static void gfsr4_set(gfsr4_state *st, unsigned long s) {
    uint32_t msb = 0x80000000u, mask = 0xffffffffu;
    if (s == 0) s = 4357;
    uint32_t x = (uint32_t)s;
    for (int i = 0; i <= GF_M; i++) {
        uint32_t t = 0, bit = 0x80000000u;
        for (int j = 0; j < 32; j++) {
            x = 69069u * x; /* mod 2^32 */
            if (x & 0x80000000u) t |= bit;
            bit >>= 1;
        }
        st->ra[i] = t;
    }
    for (int i = 0; i < 32; ++i) {
        const int k = 7 + i * 3;
        st->ra[k] &= mask; /* turn off bits left of the diagonal */
        st->ra[k] |= msb;  /* turn on the diagonal bit */
        mask >>= 1;
        msb >>= 1;
    }
    st->nd = 32;
}

static inline uint32_t gfsr4_get(gfsr4_state *st) {
    st->nd = (st->nd + 1) & GF_M;
    return st->ra[st->nd] = st->ra[(st->nd + (GF_M + 1 - GF_A)) & GF_M] ^
                            st->ra[(st->nd + (GF_M + 1 - GF_B)) & GF_M] ^
                            st->ra[(st->nd + (GF_M + 1 - GF_C)) & GF_M] ^
                            st->ra[(st->nd + (GF_M + 1 - GF_D)) & GF_M];
}

/* ------------------------------------------------------------------ cmrg, mrg, taus2
 * The three remaining generators of seed_rng_threads (rng.c:66-77), written from their defining
 * recurrences with exact 64-bit modular arithmetic (GSL evaluates the same recurrences in 32-bit
 * longs by Schrage's decomposition, the library under test follows that route):
 *   cmrg (L'Ecuyer 1996):  x_n = (63308 x_{n-2} - 183326 x_{n-3}) mod (2^31 - 1),
 *                          y_n = (86098 y_{n-1} - 539608 y_{n-3}) mod 2145483479,
 *                          output (x_n - y_n) mod (2^31 - 1); uniform = output / 2147483647
 *   mrg (L'Ecuyer, Blouin & Couture 1993):
 *                          x_n = (107374182 x_{n-1} + 104480 x_{n-5}) mod (2^31 - 1)
 *   taus2 (L'Ecuyer 1996, seeding of 1999): three Tausworthe components, output their XOR;
 *                          uniform = output / 2^32
 * all seeded from successive values of the LCG s -> 69069 s (mod 2^32) (seed 0 meaning 1) and
 * warmed up by 7, 6 and 6 steps.  GSL's self-test values pin them (tests/test_oracle_gslrng.py). */
typedef struct {
    int64_t x[3], y[3];
} cmrg_state;
typedef struct {
    int64_t x[5];
} mrg_state;
typedef struct {
    uint32_t s[3];
} taus_state;

static inline int64_t pmod(int64_t v, int64_t m) {
    v %= m;
    return v < 0 ? v + m : v;
}

static uint32_t cmrg_get(cmrg_state *c) {
    const int64_t m1 = 2147483647, m2 = 2145483479;
    const int64_t xn = pmod(63308 * c->x[1] - 183326 * c->x[2], m1);
    c->x[2] = c->x[1], c->x[1] = c->x[0], c->x[0] = xn;
    const int64_t yn = pmod(86098 * c->y[0] - 539608 * c->y[2], m2);
    c->y[2] = c->y[1], c->y[1] = c->y[0], c->y[0] = yn;
    return (uint32_t)(xn < yn ? xn - yn + m1 : xn - yn);
}

static uint32_t mrg_get(mrg_state *g) {
    const int64_t m = 2147483647;
    const int64_t xn = pmod(107374182 * g->x[0] + 104480 * g->x[4], m);
    g->x[4] = g->x[3], g->x[3] = g->x[2], g->x[2] = g->x[1], g->x[1] = g->x[0], g->x[0] = xn;
    return (uint32_t)xn;
}

static uint32_t taus_get(taus_state *t) {
    uint32_t *s = t->s;
    s[0] = ((s[0] & 4294967294u) << 12) ^ (((s[0] << 13) ^ s[0]) >> 19);
    s[1] = ((s[1] & 4294967288u) << 4) ^ (((s[1] << 2) ^ s[1]) >> 25);
    s[2] = ((s[2] & 4294967280u) << 17) ^ (((s[2] << 3) ^ s[2]) >> 11);
    return s[0] ^ s[1] ^ s[2];
}

static inline uint32_t lcg69069(uint32_t s) { return 69069u * s; }

/* ------------------------------------------------------------------ generic front */
enum { OGSL_MT19937 = 0, OGSL_GFSR4 = 1, OGSL_CMRG = 2, OGSL_MRG = 3, OGSL_TAUS2 = 4 };
typedef struct oracle_gsl_rng {
    int kind;
    union {
        mt_state mt;
        gfsr4_state gf;
        cmrg_state cm;
        mrg_state mr;
        taus_state ta;
    } u;
} oracle_gsl_rng;

oracle_gsl_rng *oracle_gsl_rng_alloc(int kind, unsigned long seed) {
    oracle_gsl_rng *r = (oracle_gsl_rng *)malloc(sizeof(*r));
    if (!r) return NULL;
    r->kind = kind;
    if (kind == OGSL_MT19937)
        mt_set(&r->u.mt, seed);
    else if (kind == OGSL_GFSR4)
        gfsr4_set(&r->u.gf, seed);
    else if (kind == OGSL_CMRG || kind == OGSL_MRG || kind == OGSL_TAUS2) {
        uint32_t s = seed == 0 ? 1u : (uint32_t)seed;
        if (kind == OGSL_CMRG) {
            for (int i = 0; i < 3; i++) r->u.cm.x[i] = (s = lcg69069(s)) % 2147483647u;
            for (int i = 0; i < 3; i++) r->u.cm.y[i] = (s = lcg69069(s)) % 2145483479u;
            for (int i = 0; i < 7; i++) (void)cmrg_get(&r->u.cm);
        } else if (kind == OGSL_MRG) {
            for (int i = 0; i < 5; i++) r->u.mr.x[i] = (s = lcg69069(s)) % 2147483647u;
            for (int i = 0; i < 6; i++) (void)mrg_get(&r->u.mr);
        } else {
            const uint32_t floor_[3] = {2u, 8u, 16u};
            for (int i = 0; i < 3; i++) {
                s = lcg69069(s);
                if (s < floor_[i]) s += floor_[i];
                r->u.ta.s[i] = s;
            }
            for (int i = 0; i < 6; i++) (void)taus_get(&r->u.ta);
        }
    } else {
        free(r);
        return NULL;
    }
    return r;
}

void oracle_gsl_rng_free(oracle_gsl_rng *r) { free(r); }

uint32_t oracle_gsl_rng_get(oracle_gsl_rng *r) {
    switch (r->kind) {
        case OGSL_MT19937: return mt_get(&r->u.mt);
        case OGSL_GFSR4: return gfsr4_get(&r->u.gf);
        case OGSL_CMRG: return cmrg_get(&r->u.cm);
        case OGSL_MRG: return mrg_get(&r->u.mr);
        default: return taus_get(&r->u.ta);
    }
}

/* get_double: get / 2^32 for the 32-bit generators, get / (2^31 - 1) for the two modulo that prime */
static inline double rng_uniform(oracle_gsl_rng *r) {
    const double range = (r->kind == OGSL_CMRG || r->kind == OGSL_MRG) ? 2147483647.0 : 4294967296.0;
    return oracle_gsl_rng_get(r) / range;
}

static inline double rng_uniform_pos(oracle_gsl_rng *r) {
    double x;
    do x = rng_uniform(r);
    while (x == 0);
    return x;
}

static unsigned long rng_uniform_int(oracle_gsl_rng *r, unsigned long n) {
    const unsigned long range = 0xffffffffUL, scale = range / n;
    unsigned long k;
    do k = oracle_gsl_rng_get(r) / scale;
    while (k >= n);
    return k;
}

double oracle_gsl_ran_ugaussian(oracle_gsl_rng *r) {
    double x, y, r2;
    do {
        x = -1 + 2 * rng_uniform_pos(r);
        y = -1 + 2 * rng_uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0);
    return 1.0 * y * sqrt(-2.0 * log(r2) / r2);
}

/* rng.c:31-56: N_THREADS seeds chosen out of the integers 0 .. INT_MAX/16 - 1 (src[i] = i, so
 * the 512 MB table of the reference is not needed), then shuffled. */
int oracle_gsl_thread_seeds(unsigned long long seed, int n_threads, unsigned int *seeds) {
    if (n_threads < 1) return C21CM_VALUE_ERROR;
    oracle_gsl_rng *rseed = oracle_gsl_rng_alloc(OGSL_MT19937, (unsigned long)seed);
    if (!rseed) return C21CM_MEMORY_ALLOC_ERROR;
    const size_t n = 2147483647 / 16, k = (size_t)n_threads;
    size_t j = 0;
    for (size_t i = 0; i < n && j < k; i++) /* gsl_ran_choose */
        if ((n - i) * rng_uniform(rseed) < k - j) seeds[j++] = (unsigned int)i;
    for (size_t i = k - 1; i > 0; i--) { /* gsl_ran_shuffle */
        const size_t jj = rng_uniform_int(rseed, i + 1);
        const unsigned int t = seeds[i];
        seeds[i] = seeds[jj];
        seeds[jj] = t;
    }
    oracle_gsl_rng_free(rseed);
    return 0;
}

/* The iterations thread t of n_threads gets from `#pragma omp for` over [0, n) with the default
 * (static, no chunk) schedule of libgomp and libomp: contiguous blocks, the first n % n_threads
 * threads one iteration longer. */
static void omp_static_block(int n, int n_threads, int t, int *lo, int *hi) {
    const int q = n / n_threads, rem = n % n_threads;
    *lo = t * q + (t < rem ? t : rem);
    *hi = *lo + q + (t < rem ? 1 : 0);
}

/* sample_ic_modes (:103-139): the (a, b) deviates of every mode in the order each thread's
 * generator produces them.  ab = double[nx][ny][nz/2+1][2]. */
int oracle_gsl_mode_deviates(unsigned long long seed, int n_threads, int nx, int ny, int nz,
                             double *ab) {
    if (n_threads < 1) return C21CM_VALUE_ERROR;
    unsigned int *seeds = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)n_threads);
    if (!seeds) return C21CM_MEMORY_ALLOC_ERROR;
    int st = oracle_gsl_thread_seeds(seed, n_threads, seeds);
    if (st) {
        free(seeds);
        return st;
    }
    const int nzc = nz / 2 + 1;
    for (int t = 0; t < n_threads; t++) {
        oracle_gsl_rng *r = oracle_gsl_rng_alloc(t % 5, seeds[t]); /* rng.c:58-85 */
        if (!r) {
            free(seeds);
            return C21CM_MEMORY_ALLOC_ERROR;
        }
        int lo, hi;
        omp_static_block(nx, n_threads, t, &lo, &hi);
        double *p = ab + (size_t)lo * ny * nzc * 2;
        for (size_t m = 0, end = (size_t)(hi - lo) * ny * nzc; m < end; m++) {
            p[2 * m] = oracle_gsl_ran_ugaussian(r);
            p[2 * m + 1] = oracle_gsl_ran_ugaussian(r);
        }
        oracle_gsl_rng_free(r);
    }
    free(seeds);
    return 0;
}

/* adj_complex_conj, statement for statement (:26-101) */
static void adj_complex_conj(float *box, int nx, int ny, int nz) {
    const int nzc = nz / 2 + 1;
    const int mid[3] = {nx / 2, ny / 2, nz / 2};
#define CIDX(i, j, k) (2 * (((size_t)(i) * ny + (j)) * nzc + (k)))
    const size_t corners[7] = {CIDX(0, 0, mid[2]),      CIDX(0, mid[1], 0),      CIDX(0, mid[1], mid[2]),
                               CIDX(mid[0], 0, 0),      CIDX(mid[0], 0, mid[2]), CIDX(mid[0], mid[1], 0),
                               CIDX(mid[0], mid[1], mid[2])};
    for (int i = 0; i < 7; i++) box[corners[i] + 1] = 0.f;
    box[CIDX(0, 0, 0)] = 0.f;
    box[CIDX(0, 0, 0) + 1] = 0.f;
    for (int i = 1; i < mid[0]; i++) {
        for (int j = 0; j <= mid[1]; j += mid[1])
            for (int k = 0; k <= mid[2]; k += mid[2]) {
                const size_t a = CIDX(i, j, k), b = CIDX(nx - i, j, k);
                box[a] = box[b];
                box[a + 1] = -box[b + 1];
            }
        for (int j = 1; j < mid[1]; j++)
            for (int k = 0; k <= mid[2]; k += mid[2]) {
                const size_t a = CIDX(i, j, k), rx = CIDX(nx - i, j, k), ry = CIDX(i, ny - j, k),
                             rxy = CIDX(nx - i, ny - j, k);
                box[a] = box[rxy];
                box[a + 1] = -box[rxy + 1];
                box[ry] = box[rx];
                box[ry + 1] = -box[rx + 1];
            }
    }
    for (int i = 0; i <= mid[0]; i += mid[0])
        for (int j = 1; j < mid[1]; j++)
            for (int k = 0; k <= mid[2]; k += mid[2]) {
                const size_t a = CIDX(i, j, k), ry = CIDX(i, ny - j, k);
                box[a] = box[ry];
                box[a + 1] = -box[ry + 1];
            }
#undef CIDX
}

static inline double index_to_k(int idx, double len, int dim) {
    double buf = (idx <= dim / 2) ? idx : (idx - dim);
    return buf * 2. * M_PI / len;
}

/* delta_k of the reference for `seed` and N_THREADS = n_threads: complex float[nx][ny][nz/2+1].
 * P(k) comes from the spec's table (cubic boxes: m = n_x^2 + n_y^2 + n_z^2 in units of dk^2). */
int oracle_gsl_sample_modes(const c21cm_ics_spec *s, int n_threads, float *cbox) {
    const int nx = s->dim, ny = s->dim, nz = s->dim_z, nzc = nz / 2 + 1;
    if (!s->pk_by_m || nx != nz || s->box_len != s->box_len_z) return C21CM_VALUE_ERROR;
    double *ab = (double *)malloc(sizeof(double) * 2 * (size_t)nx * ny * nzc);
    if (!ab) return C21CM_MEMORY_ALLOC_ERROR;
    int st = oracle_gsl_mode_deviates(s->seed, n_threads, nx, ny, nz, ab);
    if (st) {
        free(ab);
        return st;
    }
    (void)index_to_k;
    for (int n_x = 0; n_x < nx; n_x++) {
        const int ax = n_x <= nx / 2 ? n_x : nx - n_x;
        for (int n_y = 0; n_y < ny; n_y++) {
            const int ay = n_y <= ny / 2 ? n_y : ny - n_y;
            for (int n_z = 0; n_z < nzc; n_z++) {
                const long m = (long)ax * ax + (long)ay * ay + (long)n_z * n_z;
                const double p = s->pk_by_m[m];
                const size_t idx = ((size_t)n_x * ny + n_y) * nzc + n_z;
                const double amp = sqrt(s->volume * p / 2.0);
                cbox[2 * idx] = (float)(amp * ab[2 * idx]);
                cbox[2 * idx + 1] = (float)(amp * ab[2 * idx + 1]);
            }
        }
    }
    free(ab);
    adj_complex_conj(cbox, nx, ny, nz);
    return 0;
}
