/*
 * gsl_stream.c -- the reference's initial-condition random stream, for "same seed, same universe".
 *
 * reference call sites:
 *   src/py21cmfast/src/rng.c:31-90                  seed_rng_threads(r, random_seed)
 *   src/py21cmfast/src/InitialConditions.c:103-139  sample_ic_modes: per mode two
 *       gsl_ran_ugaussian(r[omp_get_thread_num()]) inside `#pragma omp for` over n_x
 *
 * The generators are GSL's, a dependency of the reference that is neither in its tree nor in
 * this image; they are written here from their published definitions:
 *   mt19937  Matsumoto & Nishimura (1998) with the 2002 seeding recurrence
 *            x_i = 1812433253 (x_{i-1} ^ (x_{i-1} >> 30)) + i, seed 0 meaning 4357
 *   gfsr4    Ziff (1998): x_n = x_{n-471} ^ x_{n-1586} ^ x_{n-6988} ^ x_{n-9689} on 2^14 words,
 *            filled bit by bit from the top bit of the LCG x -> 69069 x (mod 2^32), 32 words
 *            forced to a unit upper-triangular pattern, first output at position 33
 *   uniform = word / 2^32; uniform_pos rejects 0; uniform_int(n) = word / (0xffffffff / n),
 *            rejecting values >= n
 *   choose   selection sampling: item i of n is taken with probability (k - taken)/(n - i)
 *   shuffle  Fisher-Yates from the last element down, swapping with uniform_int(i + 1)
 *   ugaussian  polar method: (x, y) = 2 u_pos - 1 until 0 < x^2 + y^2 <= 1, value
 *            y sqrt(-2 ln r2 / r2); the pair's other value is discarded
 *   cmrg     L'Ecuyer (1996) combined multiple recursive generator: two order-3 recurrences
 *            modulo 2147483647 and 2145483479 (multipliers 63308, -183326 / 86098, -539608, each
 *            step by Schrage's decomposition), output (x - y) mod m1; seeded from the LCG
 *            x -> 69069 x (mod 2^32), seven warm-up steps; uniform = value / 2147483647
 *   mrg      L'Ecuyer, Blouin & Couture (1993) order-5 recurrence modulo 2147483647
 *            (multipliers 107374182 and 104480), same seeding, six warm-up steps
 *   taus2    L'Ecuyer (1996/1999) three-component Tausworthe generator with the 1999 seeding
 *            (components forced >= 2, 8, 16), six warm-up steps; uniform = value / 2^32
 *   GSL's own self-test values (rng/test.c: the 10000th output for seed 1 is 719452880 for cmrg,
 *   2064828650 for mrg, 2733957125 for taus2) are reproduced (the generator tests under tests/).
 * Thread t of seed_rng_threads cycles through mt19937, gfsr4, cmrg, mrg, taus2 (rng.c:58-85), so
 * any N_THREADS reproduces upstream; the reference's fixtures pin the first two
 * (tests/produce_integration_test_data.py:62,213-220), the other three rest on GSL's self-test
 * values.  A mode belongs to the
 * thread that owns its n_x under the default static OpenMP schedule (contiguous blocks, the
 * first DIM % N_THREADS threads one row longer).
 *
 * Each thread's stream is serial by definition (the number of words a deviate consumes depends
 * on the words themselves), so they are drawn on the host, one OpenMP thread per stream; the
 * device then applies
 * sqrt(V P(k) / 2) and the Hermitian constraints (ics_kernels.hip: sample_modes_kernel).
 * Parity: bit-identical deviates to the CPU oracle's independent restatement
 * (tests/test_gpu_ics.py), which the reference's own HDF5 fixtures pin
 * (tests/test_reference_fixtures.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../hip/c21hip.h"
#include "c21cm_grid.h"

/* ---- word sources: both fill a block of raw 32-bit outputs at a time ------------------- */
typedef struct word_source {
    uint32_t *buf;
    int pos, len;
    void (*refill)(struct word_source *);
    /* mt19937 */
    uint32_t mt[624];
    /* gfsr4 */
    uint32_t *ring;
    int nd;
    /* cmrg (x1..x3, y1..y3), mrg (x1..x5), taus2 (s1..s3) */
    int kind;
    long int lx[6];
    uint32_t ts[3];
} word_source;

/* ---- the three generators without a block structure ------------------------------------ */
static unsigned long cmrg_get(word_source *w) {
    const long int m1 = 2147483647, m2 = 2145483479;
    const long int a2 = 63308, qa2 = 33921, ra2 = 12979, a3 = -183326, qa3 = 11714, ra3 = 2883;
    const long int b1 = 86098, qb1 = 24919, rb1 = 7417, b3 = -539608, qb3 = 3976, rb3 = 2071;
    long int *x = w->lx, *y = w->lx + 3;
    { /* component 1 */
        const long int h3 = x[2] / qa3, h2 = x[1] / qa2;
        long int p3 = -a3 * (x[2] - h3 * qa3) - h3 * ra3;
        long int p2 = a2 * (x[1] - h2 * qa2) - h2 * ra2;
        if (p3 < 0) p3 += m1;
        if (p2 < 0) p2 += m1;
        x[2] = x[1], x[1] = x[0], x[0] = p2 - p3;
        if (x[0] < 0) x[0] += m1;
    }
    { /* component 2 */
        const long int h3 = y[2] / qb3, h1 = y[0] / qb1;
        long int p3 = -b3 * (y[2] - h3 * qb3) - h3 * rb3;
        long int p1 = b1 * (y[0] - h1 * qb1) - h1 * rb1;
        if (p3 < 0) p3 += m2;
        if (p1 < 0) p1 += m2;
        y[2] = y[1], y[1] = y[0], y[0] = p1 - p3;
        if (y[0] < 0) y[0] += m2;
    }
    return x[0] < y[0] ? (unsigned long)(x[0] - y[0] + m1) : (unsigned long)(x[0] - y[0]);
}

static unsigned long mrg_get(word_source *w) {
    const long int m = 2147483647, a1 = 107374182, q1 = 20, r1 = 7, a5 = 104480, q5 = 20554, r5 = 1727;
    long int *x = w->lx;
    const long int h5 = x[4] / q5, h1 = x[0] / q1;
    long int p5 = a5 * (x[4] - h5 * q5) - h5 * r5;
    long int p1 = a1 * (x[0] - h1 * q1) - h1 * r1;
    if (p5 > 0) p5 -= m;
    if (p1 < 0) p1 += m;
    x[4] = x[3], x[3] = x[2], x[2] = x[1], x[1] = x[0], x[0] = p1 + p5;
    if (x[0] < 0) x[0] += m;
    return (unsigned long)x[0];
}

static unsigned long taus2_get(word_source *w) {
#define TAUSWORTHE(s, a, b, c, d) ((((s) & (c)) << (d)) ^ ((((s) << (a)) ^ (s)) >> (b)))
    uint32_t *s = w->ts;
    s[0] = TAUSWORTHE(s[0], 13, 19, 4294967294u, 12);
    s[1] = TAUSWORTHE(s[1], 2, 25, 4294967288u, 4);
    s[2] = TAUSWORTHE(s[2], 3, 11, 4294967280u, 17);
#undef TAUSWORTHE
    return s[0] ^ s[1] ^ s[2];
}



static void mt_refill(word_source *w) {
    uint32_t *x = w->mt;
    for (int i = 0; i < 624; i++) {
        const uint32_t y = (x[i] & 0x80000000u) | (x[(i + 1) % 624] & 0x7fffffffu);
        x[i] = x[(i + 397) % 624] ^ (y >> 1) ^ (0x9908b0dfu & (0u - (y & 1u)));
    }
    for (int i = 0; i < 624; i++) { /* tempering */
        uint32_t k = x[i];
        k ^= k >> 11;
        k ^= (k << 7) & 0x9d2c5680u;
        k ^= (k << 15) & 0xefc60000u;
        k ^= k >> 18;
        w->buf[i] = k;
    }
    w->pos = 0;
    w->len = 624;
}

static void gfsr4_refill(word_source *w) {
    uint32_t *r = w->ring;
    int nd = w->nd;
    for (int i = 0; i < 624; i++) {
        nd = (nd + 1) & 16383;
        r[nd] = r[(nd + 16384 - 471) & 16383] ^ r[(nd + 16384 - 1586) & 16383] ^
                r[(nd + 16384 - 6988) & 16383] ^ r[(nd + 16384 - 9689) & 16383];
        w->buf[i] = r[nd];
    }
    w->nd = nd;
    w->pos = 0;
    w->len = 624;
}

static int source_open(word_source *w, int kind, unsigned long seed) {
    memset(w, 0, sizeof(*w));
    w->buf = (uint32_t *)malloc(624 * sizeof(uint32_t));
    if (!w->buf) return C21CM_MEMORY_ALLOC_ERROR;
    w->kind = kind;
    if (kind >= 2) { /* cmrg, mrg, taus2: seeded through the LCG x -> 69069 x mod 2^32; 0 means 1 */
        uint32_t x = seed == 0 ? 1u : (uint32_t)seed;
#define LCG(n) ((uint32_t)(69069u * (n)))
        if (kind == 2) {
            for (int i = 0; i < 6; i++) {
                x = LCG(x);
                w->lx[i] = (long int)(x % (i < 3 ? 2147483647u : 2145483479u));
            }
            for (int i = 0; i < 7; i++) (void)cmrg_get(w);
        } else if (kind == 3) {
            for (int i = 0; i < 5; i++) {
                x = LCG(x);
                w->lx[i] = (long int)(x % 2147483647u);
            }
            for (int i = 0; i < 6; i++) (void)mrg_get(w);
        } else {
            w->ts[0] = LCG(x);
            if (w->ts[0] < 2) w->ts[0] += 2u;
            w->ts[1] = LCG(w->ts[0]);
            if (w->ts[1] < 8) w->ts[1] += 8u;
            w->ts[2] = LCG(w->ts[1]);
            if (w->ts[2] < 16) w->ts[2] += 16u;
            for (int i = 0; i < 6; i++) (void)taus2_get(w);
        }
#undef LCG
        return 0;
    }
    if (seed == 0) seed = 4357;
    if (kind == 0) {
        w->mt[0] = (uint32_t)seed;
        for (uint32_t i = 1; i < 624; i++)
            w->mt[i] = 1812433253u * (w->mt[i - 1] ^ (w->mt[i - 1] >> 30)) + i;
        w->refill = mt_refill;
    } else {
        w->ring = (uint32_t *)malloc(16384 * sizeof(uint32_t));
        if (!w->ring) return C21CM_MEMORY_ALLOC_ERROR;
        uint32_t x = (uint32_t)seed;
        for (int i = 0; i < 16384; i++) {
            uint32_t word = 0;
            for (int b = 31; b >= 0; b--) {
                x *= 69069u;
                word |= (x >> 31) << b;
            }
            w->ring[i] = word;
        }
        for (int i = 0; i < 32; i++) { /* rows 7, 10, 13, ...: zero left of the diagonal, one on it */
            const uint32_t diag = 0x80000000u >> i;
            w->ring[7 + 3 * i] = (w->ring[7 + 3 * i] & (0xffffffffu >> i)) | diag;
        }
        w->nd = 32;
        w->refill = gfsr4_refill;
    }
    w->pos = w->len = 0;
    return 0;
}

static void source_close(word_source *w) {
    free(w->buf);
    free(w->ring);
    w->buf = w->ring = NULL;
}

static inline uint32_t next_word(word_source *w) {
    if (w->pos == w->len) w->refill(w);
    return w->buf[w->pos++];
}

static inline double next_uniform_pos(word_source *w) {
    if (w->kind >= 2) { /* gsl_rng_uniform_pos: the generator's get_double until it is non-zero */
        double u;
        do {
            if (w->kind == 2)
                u = cmrg_get(w) / 2147483647.0;
            else if (w->kind == 3)
                u = mrg_get(w) / 2147483647.0;
            else
                u = taus2_get(w) / 4294967296.0;
        } while (u == 0);
        return u;
    }
    uint32_t v;
    do v = next_word(w);
    while (v == 0);
    return v * (1.0 / 4294967296.0);
}

static inline double next_ugaussian(word_source *w) {
    double x, y, r2;
    do {
        x = 2 * next_uniform_pos(w) - 1;
        y = 2 * next_uniform_pos(w) - 1;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0);
    return y * sqrt(-2.0 * log(r2) / r2);
}

/* rng.c:31-56 with src[i] = i: the per-thread seeds */
int c21_gsl_thread_seeds(unsigned long long seed, int n_threads, unsigned int *seeds) {
    if (n_threads < 1 || !seeds) return C21CM_VALUE_ERROR;
    word_source w;
    int st = source_open(&w, 0, (unsigned long)seed);
    if (st) return st;
    const uint64_t n = 2147483647 / 16;
    uint64_t taken = 0;
    for (uint64_t i = 0; i < n && taken < (uint64_t)n_threads; i++) {
        const double u = next_word(&w) * (1.0 / 4294967296.0);
        if ((double)(n - i) * u < (double)((uint64_t)n_threads - taken)) seeds[taken++] = (unsigned int)i;
    }
    for (int i = n_threads - 1; i > 0; i--) {
        const uint32_t scale = 0xffffffffu / (uint32_t)(i + 1);
        uint32_t j;
        do j = next_word(&w) / scale;
        while (j >= (uint32_t)(i + 1));
        const unsigned int t = seeds[i];
        seeds[i] = seeds[j];
        seeds[j] = t;
    }
    source_close(&w);
    return 0;
}

int c21_gsl_stream_supported(int n_threads) { return n_threads >= 1 && n_threads <= 4096; }

/* the n-th raw output of generator `kind` (0 mt19937, 1 gfsr4, 2 cmrg, 3 mrg, 4 taus2) after
 * gsl_rng_set(seed): what GSL's own self-test compares (rng/test.c) */
unsigned long c21_gsl_nth_output(int kind, unsigned long seed, int n) {
    word_source w;
    if (kind < 0 || kind > 4 || n < 1 || source_open(&w, kind, seed)) return 0;
    unsigned long v = 0;
    for (int i = 0; i < n; i++)
        v = kind == 2 ? cmrg_get(&w) : kind == 3 ? mrg_get(&w) : kind == 4 ? taus2_get(&w) : next_word(&w);
    source_close(&w);
    return v;
}

/* The (a, b) deviates of all nx * ny * nzc modes in grid order: ab[2 * mode + {0, 1}]. */
int c21_gsl_mode_deviates(unsigned long long seed, int n_threads, int nx, int ny, int nzc,
                          double *ab) {
    if (!c21_gsl_stream_supported(n_threads)) {
        c21hip_set_error("ics: N_THREADS = %d is outside 1..4096", n_threads);
        return C21CM_VALUE_ERROR;
    }
    unsigned int *seeds = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)n_threads);
    if (!seeds) return C21CM_MEMORY_ALLOC_ERROR;
    int st = c21_gsl_thread_seeds(seed, n_threads, seeds);
    if (st) {
        free(seeds);
        return st;
    }
    const int q = nx / n_threads, rem = nx % n_threads;
    /* the threads' streams are independent of each other: draw them concurrently, as upstream's
     * OpenMP loop does (each stream stays serial) */
    int failed = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 64 ? 64 : n_threads) reduction(| : failed)
    for (int t = 0; t < n_threads; t++) {
        const int lo = t * q + (t < rem ? t : rem), rows = q + (t < rem ? 1 : 0);
        word_source w;
        if (source_open(&w, t % 5, seeds[t])) { /* rng.c:58-85: the five kinds in turn */
            failed |= 1;
            continue;
        }
        double *p = ab + 2 * (size_t)lo * ny * nzc;
        const size_t count = 2 * (size_t)rows * ny * nzc;
        for (size_t m = 0; m < count; m++) p[m] = next_ugaussian(&w);
        source_close(&w);
    }
    if (failed) {
        free(seeds);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    free(seeds);
    return 0;
}
