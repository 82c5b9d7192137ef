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
/* (Both recurrences in 64-bit arithmetic: the products stay below 2^59, and the reductions by the
 * constant moduli compile to multiplications.  GSL steps them by Schrage's decomposition in 32-bit
 * longs; the residues are the same numbers -- GSL's self-test values and the oracle's independent
 * restatement pin them -- at a third of the time: 22 -> 7 ns per output, which matters because the
 * slowest of the N_THREADS streams sets the wall time of the IC draw.) */
static unsigned long cmrg_get(word_source *w) {
    const int64_t m1 = 2147483647, m2 = 2145483479;
    long int *x = w->lx, *y = w->lx + 3;
    { /* component 1: x_n = 63308 x_{n-2} - 183326 x_{n-3} (mod m1) */
        int64_t t = (63308 * (int64_t)x[1] - 183326 * (int64_t)x[2]) % m1;
        if (t < 0) t += m1;
        x[2] = x[1], x[1] = x[0], x[0] = (long int)t;
    }
    { /* component 2: y_n = 86098 y_{n-1} - 539608 y_{n-3} (mod m2) */
        int64_t t = (86098 * (int64_t)y[0] - 539608 * (int64_t)y[2]) % m2;
        if (t < 0) t += m2;
        y[2] = y[1], y[1] = y[0], y[0] = (long int)t;
    }
    return x[0] < y[0] ? (unsigned long)(x[0] - y[0] + m1) : (unsigned long)(x[0] - y[0]);
}

static unsigned long mrg_get(word_source *w) {
    /* x_n = 107374182 x_{n-1} + 104480 x_{n-5} (mod 2^31 - 1) */
    const int64_t m = 2147483647;
    long int *x = w->lx;
    const int64_t t = (107374182 * (int64_t)x[0] + 104480 * (int64_t)x[4]) % m;
    x[4] = x[3], x[3] = x[2], x[2] = x[1], x[1] = x[0], x[0] = (long int)t;
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

/* ---- the same stream with the transcendental half on the device (round 4) ------------------------
 * What is serial in a thread's stream is the acceptance loop, not the logarithm: the host draws words
 * until a pair is accepted (0 < x^2 + y^2 <= 1) and keeps the two RAW outputs of the accepted pair --
 * eight bytes per deviate, the size of the deviate itself -- and the device turns them into
 * y sqrt(-2 ln r2 / r2) in place (ics_kernels.hip: gsl_words_kernel).  Every thread stages its words
 * through two pinned chunks on its own stream while it keeps drawing, so the copy over PCIe and the
 * page faults of an 8.6 GB host array (DIM = 1024) are off the critical path as well.  5.8 -> ~1.5 s
 * at DIM = 1024 with 16 streams.  dev_ab: device array of 2 nx ny nzc doubles, filled in grid order. */
static inline uint32_t next_raw_pos(word_source *w) { /* the raw output behind gsl_rng_uniform_pos */
    if (w->kind >= 2) {
        unsigned long v;
        do v = w->kind == 2 ? cmrg_get(w) : w->kind == 3 ? mrg_get(w) : taus2_get(w);
        while (v == 0);
        return (uint32_t)v;
    }
    uint32_t v;
    do v = next_word(w);
    while (v == 0);
    return v;
}
static inline double raw_to_uniform(int kind, uint32_t v) {
    return (kind == 2 || kind == 3) ? v / 2147483647.0 : v * (1.0 / 4294967296.0);
}

int c21_gsl_mode_deviates_device(unsigned long long seed, int n_threads, int nx, int ny, int nzc,
                                 double *dev_ab, void *stream) {
    if (!c21_gsl_stream_supported(n_threads)) {
        c21hip_set_error("ics: N_THREADS = %d is outside 1..4096", n_threads);
        return C21CM_VALUE_ERROR;
    }
    unsigned int *seeds = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)n_threads);
    unsigned char *row_kind = (unsigned char *)malloc((size_t)nx);
    if (!seeds || !row_kind) {
        free(seeds);
        free(row_kind);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    int st = c21_gsl_thread_seeds(seed, n_threads, seeds);
    if (st) {
        free(seeds);
        free(row_kind);
        return st;
    }
    const int q = nx / n_threads, rem = nx % n_threads;
    const size_t per_row = 2 * (size_t)ny * nzc; /* deviates per x-row */
    const int device = c21hip_current_device();
    /* deviates per staging chunk: 32 MB each, two per stream -- but no more than 512 MB of pinned memory
     * over all concurrently drawing streams (64 streams of 2 x 32 MB were 4 GB: ADVICE r4) */
    size_t CH = (size_t)1 << 22;
    {
        const int streams = n_threads > 64 ? 64 : n_threads;
        while (CH > ((size_t)1 << 16) && 2 * CH * sizeof(uint64_t) * (size_t)streams > ((size_t)512 << 20)) CH >>= 1;
    }
    int failed = 0;
    /* everything queued on the caller's stream before this call has to be done with dev_ab */
    if ((st = c21hip_sync(stream))) {
        free(seeds);
        free(row_kind);
        return st;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 64 ? 64 : n_threads) reduction(| : failed)
    for (int t = 0; t < n_threads; t++) {
        const int lo = t * q + (t < rem ? t : rem), rows = q + (t < rem ? 1 : 0);
        for (int r = 0; r < rows; r++) row_kind[lo + r] = (unsigned char)(t % 5);
        if (rows == 0) continue;
        word_source w;
        uint64_t *chunk[2] = {NULL, NULL};
        void *ev[2] = {NULL, NULL}, *cs = NULL;
        int ok = device < 0 || c21hip_use_device(device) == 0;
        if (ok && source_open(&w, t % 5, seeds[t])) ok = 0; /* rng.c:58-85: the five kinds in turn */
        if (!ok) {
            failed |= 1;
            continue;
        }
        const size_t count = (size_t)rows * per_row;
        const size_t ch = count < CH ? count : CH;
        chunk[0] = (uint64_t *)c21hip_pinned_alloc(ch * sizeof(uint64_t));
        chunk[1] = (uint64_t *)c21hip_pinned_alloc(ch * sizeof(uint64_t));
        ev[0] = c21hip_event_create();
        ev[1] = c21hip_event_create();
        cs = c21hip_stream_create();
        if (!chunk[0] || !chunk[1] || !ev[0] || !ev[1] || !cs) {
            failed |= 1;
        } else {
            uint64_t *dst = (uint64_t *)dev_ab + (size_t)lo * per_row;
            const int kind = w.kind;
            int used[2] = {0, 0};
            size_t done = 0;
            for (int b = 0; done < count && !failed; b ^= 1) {
                const size_t n = count - done < ch ? count - done : ch;
                if (used[b] && c21hip_event_synchronize(ev[b])) failed |= 1; /* the chunk's last copy has left it */
                uint64_t *p = chunk[b];
                for (size_t m = 0; m < n; m++) {
                    uint32_t a, c;
                    double x, y, r2;
                    do {
                        a = next_raw_pos(&w);
                        c = next_raw_pos(&w);
                        x = 2 * raw_to_uniform(kind, a) - 1;
                        y = 2 * raw_to_uniform(kind, c) - 1;
                        r2 = x * x + y * y;
                    } while (r2 > 1.0 || r2 == 0);
                    p[m] = (uint64_t)a | ((uint64_t)c << 32);
                }
                if (c21hip_h2d(dst + done, p, n * sizeof(uint64_t), cs) || c21hip_event_record(ev[b], cs))
                    failed |= 1;
                used[b] = 1;
                done += n;
            }
            if (c21hip_sync(cs)) failed |= 1;
        }
        c21hip_pinned_free(chunk[0]);
        c21hip_pinned_free(chunk[1]);
        c21hip_event_destroy(ev[0]);
        c21hip_event_destroy(ev[1]);
        c21hip_stream_destroy(cs);
        source_close(&w);
    }
    free(seeds);
    if (failed) {
        free(row_kind);
        c21hip_set_error("ics: staging the random stream failed (pinned memory / stream / copy)");
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    /* the generator kind of every row, then the conversion in place on the caller's stream */
    unsigned char *kind_dev = (unsigned char *)c21hip_ws(255, (size_t)nx);
    if (!kind_dev) {
        free(row_kind);
        return C21CM_MEMORY_ALLOC_ERROR;
    }
    st = c21hip_h2d(kind_dev, row_kind, (size_t)nx, stream);
    if (!st) st = c21hip_sync(stream); /* `row_kind` is freed below */
    free(row_kind);
    if (st) return st;
    return c21hip_gsl_words_to_deviates(dev_ab, (size_t)nx * per_row, kind_dev, per_row, stream);
}
