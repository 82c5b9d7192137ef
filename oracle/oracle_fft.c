/*
 * oracle_fft.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * In-place, padded, unnormalised real 3-D FFTs with the array format the
 * reference obtains from FFTW: `fftwf_plan_dft_r2c_3d(dim, dim, dim_los)` /
 * `fftwf_plan_dft_c2r_3d` executed in place on a `[nx][ny][2*(nz/2+1)]` float
 * buffer (reference call sites: src/py21cmfast/src/dft.c:38-40 and :66-68;
 * layout: src/py21cmfast/src/indexing.h:90-98).
 *
 * FFTW itself (third-party, unpinned `fftw` conda package, single precision)
 * is not available in this image, so the published definition is restated:
 *   forward  Y[k] = sum_j X[j] exp(-2 pi i j k / n)   (r2c, no scaling)
 *   backward X[j] = sum_k Y[k] exp(+2 pi i j k / n)   (c2r, no scaling; the
 *   imaginary parts of the self-conjugate k_z = 0 and k_z = n/2 entries of each
 *   z-line are ignored, exactly what a c2r transform does).
 * The 1-D engine is a Stockham autosort mixed-radix FFT (any n; radices 4, 2
 * and a generic odd-prime butterfly) working on an `[n][batch]` tile so the
 * innermost loop runs over independent lines and vectorises.
 */
#include "oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float re, im;
} cpx;

#define TILE 32 /* lines transformed together by one thread */

typedef struct {
    int n;
    int nfac;
    int fac[40];
    cpx *tw; /* exp(-2 pi i k / n), k = 0..n-1 */
} fft_plan;

static void plan_init(fft_plan *p, int n) {
    p->n = n;
    p->nfac = 0;
    int m = n;
    while (m % 4 == 0) {
        p->fac[p->nfac++] = 4;
        m /= 4;
    }
    while (m % 2 == 0) {
        p->fac[p->nfac++] = 2;
        m /= 2;
    }
    for (int f = 3; m > 1; f += 2) {
        while (m % f == 0) {
            p->fac[p->nfac++] = f;
            m /= f;
        }
    }
    p->tw = (cpx *)malloc(sizeof(cpx) * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->tw[k].re = (float)cos(a);
        p->tw[k].im = (float)sin(a);
    }
}

static void plan_free(fft_plan *p) { free(p->tw); }

static inline cpx cmul(cpx a, cpx b) {
    cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}

/* One Stockham pass of radix r: x is viewed as [r][m][s], y as [m][r][s].
 * n = r*m is the current sub-transform length, s the stride (product of
 * radices already done times the batch width).  `sign` = +1 conjugates the
 * twiddles (backward transform).  N is the full length (twiddle table size). */
static void stockham_pass(const fft_plan *pl, int r, int m, long s, int sign, const cpx *x,
                          cpx *y) {
    const int N = pl->n;
    const int n = r * m;
    const int tstep = N / n; /* table stride for exp(-2 pi i /n) */
    const cpx *tw = pl->tw;
    if (r == 2) {
        for (int p = 0; p < m; p++) {
            cpx w = tw[(long)p * tstep];
            if (sign > 0) w.im = -w.im;
            const cpx *xa = x + (long)s * p;
            const cpx *xb = x + (long)s * (p + m);
            cpx *y0 = y + (long)s * (2 * p);
            cpx *y1 = y0 + s;
            for (long q = 0; q < s; q++) {
                cpx a = xa[q], b = xb[q];
                y0[q].re = a.re + b.re;
                y0[q].im = a.im + b.im;
                cpx d = {a.re - b.re, a.im - b.im};
                y1[q] = cmul(d, w);
            }
        }
        return;
    }
    if (r == 4) {
        for (int p = 0; p < m; p++) {
            cpx w1 = tw[(long)p * tstep];
            cpx w2 = tw[((long)2 * p * tstep) % N];
            cpx w3 = tw[((long)3 * p * tstep) % N];
            if (sign > 0) {
                w1.im = -w1.im;
                w2.im = -w2.im;
                w3.im = -w3.im;
            }
            const cpx *x0 = x + (long)s * p;
            const cpx *x1 = x + (long)s * (p + m);
            const cpx *x2 = x + (long)s * (p + 2 * m);
            const cpx *x3 = x + (long)s * (p + 3 * m);
            cpx *y0 = y + (long)s * (4 * p);
            cpx *y1 = y0 + s, *y2 = y1 + s, *y3 = y2 + s;
            const float sg = (sign > 0) ? -1.f : 1.f; /* -i (forward) or +i (backward) */
            for (long q = 0; q < s; q++) {
                cpx a = x0[q], b = x1[q], c = x2[q], d = x3[q];
                cpx apc = {a.re + c.re, a.im + c.im};
                cpx amc = {a.re - c.re, a.im - c.im};
                cpx bpd = {b.re + d.re, b.im + d.im};
                cpx bmd = {b.re - d.re, b.im - d.im};
                /* jbmd = (-i)*(b-d) forward, (+i)*(b-d) backward */
                cpx jbmd = {sg * bmd.im, -sg * bmd.re};
                y0[q].re = apc.re + bpd.re;
                y0[q].im = apc.im + bpd.im;
                cpx t1 = {amc.re + jbmd.re, amc.im + jbmd.im};
                cpx t2 = {apc.re - bpd.re, apc.im - bpd.im};
                cpx t3 = {amc.re - jbmd.re, amc.im - jbmd.im};
                y1[q] = cmul(t1, w1);
                y2[q] = cmul(t2, w2);
                y3[q] = cmul(t3, w3);
            }
        }
        return;
    }
    /* generic radix */
    const int rstep = N / r;
    for (int p = 0; p < m; p++) {
        for (int j = 0; j < r; j++) {
            cpx wo = tw[((long)p * j * tstep) % N];
            if (sign > 0) wo.im = -wo.im;
            cpx *yo = y + (long)s * ((long)r * p + j);
            for (long q = 0; q < s; q++) {
                float sr = 0.f, si = 0.f;
                for (int k = 0; k < r; k++) {
                    cpx wk = tw[((long)j * k % r) * rstep];
                    if (sign > 0) wk.im = -wk.im;
                    cpx a = x[(long)s * (p + (long)m * k) + q];
                    sr += a.re * wk.re - a.im * wk.im;
                    si += a.re * wk.im + a.im * wk.re;
                }
                cpx b = {sr, si};
                yo[q] = cmul(b, wo);
            }
        }
    }
}

/* Transform `batch` lines held as buf[n][batch]; result ends up in buf. */
static void fft_tile(const fft_plan *pl, int sign, long batch, cpx *buf, cpx *scratch) {
    int n = pl->n;
    long s = batch;
    cpx *x = buf, *y = scratch;
    for (int f = 0; f < pl->nfac; f++) {
        int r = pl->fac[f];
        int m = n / r;
        stockham_pass(pl, r, m, s, sign, x, y);
        n = m;
        s *= r;
        cpx *t = x;
        x = y;
        y = t;
    }
    if (x != buf) memcpy(buf, x, sizeof(cpx) * (size_t)pl->n * (size_t)batch);
}

/* Threads of the transforms, separately from the sweeps: the reference plans its FFTW transforms
 * without fftwf_plan_with_nthreads taking effect (dft.c:83-85: created per call with
 * FFTW_ESTIMATE, effectively single-threaded), so the CPU baseline is reported both
 * ways: 1 = "faithful", 0 = as many as the sweeps use ("charitable"). */
static int g_fft_threads = 0;
void oracle_set_fft_threads(int n) { g_fft_threads = n; }
static inline int fft_threads(void) { return g_fft_threads > 0 ? g_fft_threads : omp_get_max_threads(); }

/* complex FFT along x and y of the half-spectrum grid c[nx][ny][nzc] */
static void fft_xy(cpx *c, int nx, int ny, int nzc, int sign) {
    fft_plan px, py;
    plan_init(&px, nx);
    plan_init(&py, ny);
    /* y-lines: for each x the sub-array [ny][nzc] is already [n][batch] */
#pragma omp parallel num_threads(fft_threads())
    {
        cpx *buf = (cpx *)malloc(sizeof(cpx) * (size_t)(ny > nx ? ny : nx) * TILE);
        cpx *scr = (cpx *)malloc(sizeof(cpx) * (size_t)(ny > nx ? ny : nx) * TILE);
#pragma omp for collapse(2) schedule(static)
        for (int ix = 0; ix < nx; ix++) {
            for (int z0 = 0; z0 < nzc; z0 += TILE) {
                int w = (nzc - z0 < TILE) ? nzc - z0 : TILE;
                cpx *base = c + ((size_t)ix * ny) * nzc + z0;
                for (int j = 0; j < ny; j++)
                    memcpy(buf + (size_t)j * w, base + (size_t)j * nzc, sizeof(cpx) * w);
                fft_tile(&py, sign, w, buf, scr);
                for (int j = 0; j < ny; j++)
                    memcpy(base + (size_t)j * nzc, buf + (size_t)j * w, sizeof(cpx) * w);
            }
        }
        /* x-lines: the whole array is [nx][ny*nzc] */
        const size_t cols = (size_t)ny * nzc;
#pragma omp for schedule(static)
        for (long c0 = 0; c0 < (long)cols; c0 += TILE) {
            int w = (cols - c0 < TILE) ? (int)(cols - c0) : TILE;
            for (int i = 0; i < nx; i++)
                memcpy(buf + (size_t)i * w, c + (size_t)i * cols + c0, sizeof(cpx) * w);
            fft_tile(&px, sign, w, buf, scr);
            for (int i = 0; i < nx; i++)
                memcpy(c + (size_t)i * cols + c0, buf + (size_t)i * w, sizeof(cpx) * w);
        }
        free(buf);
        free(scr);
    }
    plan_free(&px);
    plan_free(&py);
}

/* reference: src/py21cmfast/src/dft.c:46-72 (dft_r2c_cube) */
void oracle_fft_r2c(float *box, int nx, int ny, int nz) {
    const int nzc = nz / 2 + 1;
    const size_t zpad = 2 * (size_t)nzc;
    fft_plan pz;
    plan_init(&pz, nz);
    const long nlines = (long)nx * ny;
#pragma omp parallel num_threads(fft_threads())
    {
        cpx *buf = (cpx *)malloc(sizeof(cpx) * (size_t)nz * TILE);
        cpx *scr = (cpx *)malloc(sizeof(cpx) * (size_t)nz * TILE);
#pragma omp for schedule(static)
        for (long l0 = 0; l0 < nlines; l0 += TILE) {
            int w = (nlines - l0 < TILE) ? (int)(nlines - l0) : TILE;
            for (int b = 0; b < w; b++) {
                const float *line = box + (size_t)(l0 + b) * zpad;
                for (int k = 0; k < nz; k++) {
                    buf[(size_t)k * w + b].re = line[k];
                    buf[(size_t)k * w + b].im = 0.f;
                }
            }
            fft_tile(&pz, -1, w, buf, scr);
            for (int b = 0; b < w; b++) {
                cpx *line = (cpx *)(box + (size_t)(l0 + b) * zpad);
                for (int k = 0; k < nzc; k++) line[k] = buf[(size_t)k * w + b];
            }
        }
        free(buf);
        free(scr);
    }
    plan_free(&pz);
    fft_xy((cpx *)box, nx, ny, nzc, -1);
}

/* reference: src/py21cmfast/src/dft.c:18-44 (dft_c2r_cube) */
void oracle_fft_c2r(float *box, int nx, int ny, int nz) {
    const int nzc = nz / 2 + 1;
    const size_t zpad = 2 * (size_t)nzc;
    fft_xy((cpx *)box, nx, ny, nzc, +1);
    fft_plan pz;
    plan_init(&pz, nz);
    const long nlines = (long)nx * ny;
#pragma omp parallel num_threads(fft_threads())
    {
        cpx *buf = (cpx *)malloc(sizeof(cpx) * (size_t)nz * TILE);
        cpx *scr = (cpx *)malloc(sizeof(cpx) * (size_t)nz * TILE);
#pragma omp for schedule(static)
        for (long l0 = 0; l0 < nlines; l0 += TILE) {
            int w = (nlines - l0 < TILE) ? (int)(nlines - l0) : TILE;
            for (int b = 0; b < w; b++) {
                const cpx *line = (const cpx *)(box + (size_t)(l0 + b) * zpad);
                for (int k = 0; k < nzc; k++) buf[(size_t)k * w + b] = line[k];
                /* Hermitian completion; self-conjugate entries lose their imaginary part */
                buf[b].im = 0.f;
                if (nz % 2 == 0) buf[(size_t)(nz / 2) * w + b].im = 0.f;
                for (int k = nzc; k < nz; k++) {
                    cpx v = line[nz - k];
                    buf[(size_t)k * w + b].re = v.re;
                    buf[(size_t)k * w + b].im = -v.im;
                }
            }
            fft_tile(&pz, +1, w, buf, scr);
            for (int b = 0; b < w; b++) {
                float *line = box + (size_t)(l0 + b) * zpad;
                for (int k = 0; k < nz; k++) line[k] = buf[(size_t)k * w + b].re;
            }
        }
        free(buf);
        free(scr);
    }
    plan_free(&pz);
}
